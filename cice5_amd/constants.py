"""Physical and numerical constants the EVP path reads.

Values follow drivers/auscom/ice_constants.F90:20-44,136-190 (rhos, rhoi, rhow, gravit,
omega, puny, the p* fractions) and source/ice_dyn_shared.F90:43-61 (eyc, a_min, m_min).
Under AusCOM `dragio`, `cosw`, `sinw` are namelist variables (ice_dyn_shared.F90:66-72);
the defaults below are the non-AusCOM parameter values (ice_constants.F90:38,
ice_dyn_shared.F90:57-58).
"""

rhos = 330.0
rhoi = 917.0
rhow = 1026.0
gravit = 9.80616
omega = 7.292e-5
puny = 1.0e-11
dragio = 0.00536
cosw = 1.0
sinw = 0.0
eyc = 0.36
a_min = 0.001
m_min = 0.01

# Hibler (1979) strength, source/ice_mechred.F90:80-82
Pstar = 2.75e4
Cstar = 20.0

# boundary types (ice_domain.F90 domain_nml: ew_boundary_type / ns_boundary_type)
BND_CYCLIC, BND_OPEN, BND_CLOSED, BND_TRIPOLE = 0, 1, 2, 3
BND_NAMES = {"cyclic": BND_CYCLIC, "open": BND_OPEN, "closed": BND_CLOSED, "tripole": BND_TRIPOLE}

# field_loc_* / field_type_* (ice_constants.F90:198-215)
LOC_CENTER, LOC_NECORNER, LOC_NFACE, LOC_EFACE = 1, 2, 3, 4        # ice_constants.F90: field_loc_*
KIND_SCALAR, KIND_VECTOR = 1, 2
