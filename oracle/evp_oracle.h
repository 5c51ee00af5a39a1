/*
 * oracle/evp_oracle.h -- CPU restatement of the CICE5 EVP dynamics path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (cice5_amd/, include/,
 * fortran/) includes, links or calls this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * PARITY: pinned by the reference's OWN output for the halo updates (orc_halo_r8 / i4 / stress), ice_strength and the
 * block / distribution arithmetic -- those reference routines compile here unmodified (oracle/ref/Makefile) and their
 * outputs are the fixtures tests/golden/ref_*.npz (tests/test_ref_pins.py).  PARITY UNPINNED for stress, stepu, evp_prep1/2,
 * evp_finish, to_ugrid / to_tgrid, horizontal_remap and eap: their Fortran modules `use ice_grid`, which needs the netCDF
 * Fortran module (source/ice_grid.F90:144), absent from the image, and the reference ships no tests or golden vectors
 * (SURVEY.md S4).  For those this restatement is checked by decomposition invariance and analytic properties
 * (tests/test_oracle.py), and tests/golden/evp_*.npz hold ITS OWN outputs (regression pins, not reference vectors).

