"""A SECOND, independent restatement of evp(dt) -- whole-array numpy, written in round 4 straight from the Fortran by a reader who
had not written oracle/evp_oracle.c -- for ONE block that covers the whole domain (cyclic E-W, open N-S).

Why: `stress`, `stepu`, `evp_prep1/2`, `evp_finish` and the grid averages cannot be pinned by reference output in this image (their
modules need netCDF), so the C restatement and the HIP kernels were both checked against ONE reading of the
Fortran.  This file is another reading, in another language and another shape (array slices instead of index lists and
per-cell loops); tests/test_numpy_crosscheck.py compares it with the C oracle bit for bit.  A transcription slip shared by the
C oracle and the kernels would have to be repeated here, independently, to go unseen.  It pins nothing by itself -- test
infrastructure, like oracle/.

Arrays are (ny_block, nx_block) = (ny + 2, nx + 2), index [j, i], Fortran (i, j) -> [j - 1, i - 1]; physical cells
ilo..ihi = 2..nx+1 -> slices 1:-1.  Every expression keeps the Fortran's left-to-right evaluation order (numpy does not
reassociate and has no fused multiply-add), `x**2` is `x * x`.

    evp            source/ice_dyn_evp.F90:68-510         stress       source/ice_dyn_evp.F90:520-849
    evp_prep1      source/ice_dyn_shared.F90:270-365     evp_prep2    source/ice_dyn_shared.F90:377-614
    stepu          source/ice_dyn_shared.F90:623-748     evp_finish   source/ice_dyn_shared.F90:757-844
    to_ugrid / t2ugrid_vector / to_tgrid / u2tgrid_vector   source/ice_grid.F90:1799-1958
    set_evp_parameters  source/ice_dyn_shared.F90:185-259
"""
import numpy as np

# drivers/auscom/ice_constants.F90:22-44, :175-188; ice_dyn_shared.F90:51-61
rhos, rhoi, rhow, gravit = 330.0, 917.0, 1026.0, 9.80616
p5, p25 = 0.5, 0.25
p166, p333, p111, p222 = 1.0 / 6.0, 1.0 / 3.0, 1.0 / 9.0, 2.0 / 9.0
p055 = p111 * p5
p027 = p055 * p5
eyc, a_min, m_min = 0.36, 0.001, 0.01


def set_evp_parameters(dt, ndte, revised_evp, xmin):
    """ice_dyn_shared.F90:185-259"""
    dte = dt / float(ndte)
    dtei = 1.0 / dte
    ecci = p25
    tdamp2 = 2.0 * eyc * dt
    dte2T = dte / tdamp2
    Se, xi = 0.86, 5.5e-3
    gamma = p25 * 1.0e11 * dt
    if revised_evp:
        revp = 1.0
        arlx1i = 2.0 * xi / Se
        brlx = 2.0 * Se * xi * gamma / (xmin * xmin)
    else:
        revp = 0.0
        arlx1i = dte2T
        brlx = dt * dtei
    denom1 = 1.0 / (1.0 + arlx1i)
    return dict(ecci=ecci, revp=revp, arlx1i=arlx1i, brlx=brlx, denom1=denom1)


def halo_cyclic(a):
    """ice_HaloUpdate of a one-block domain, cyclic E-W, open N-S (centre or NE-corner field alike): the ghost rows get the fill
    value 0, then the ghost columns wrap over all rows (mpi/ice_boundary.F90:1409-1416, :1489-1560)"""
    a[0, :] = 0.0
    a[-1, :] = 0.0
    a[:, 0] = a[:, -2]
    a[:, -1] = a[:, 1]


def to_ugrid(w1, tarea, uarea):
    """ice_grid.F90:1838-1866"""
    w2 = np.zeros_like(w1)
    c = (slice(1, -1), slice(1, -1))
    e, n, ne = (slice(1, -1), slice(2, None)), (slice(2, None), slice(1, -1)), (slice(2, None), slice(2, None))
    w2[c] = p25 * (w1[c] * tarea[c] + w1[e] * tarea[e] + w1[n] * tarea[n] + w1[ne] * tarea[ne]) / uarea[c]
    return w2


def to_tgrid(w1, tarea, uarea):
    """ice_grid.F90:1915-1945 (work2 keeps whatever it held outside the physical cells: it is strocnxT itself)"""
    c = (slice(1, -1), slice(1, -1))
    w, s_, sw = (slice(1, -1), slice(0, -2)), (slice(0, -2), slice(1, -1)), (slice(0, -2), slice(0, -2))
    return p25 * (w1[c] * uarea[c] + w1[w] * uarea[w] + w1[s_] * uarea[s_] + w1[sw] * uarea[sw]) / tarea[c]


LOC_CENTER, LOC_NECORNER = 1, 2                                                    # ice_constants.F90: field_loc_*
KIND_SCALAR, KIND_VECTOR = 1, 2                                                    #                    field_type_*


def evp(f, dt, ndte, xmin, revised_evp=False, cosw=1.0, sinw=0.0, nsub=None, eap_tables=None, halo_update=None, stress_fold=None):
    """one call of evp(dt) in place on the dict of 2-D arrays `f` (names as in ice_state / ice_flux / ice_grid);
    eap_tables = (s11r, s12r, s22r, s11s, s12s, s22s): eap(dt) instead (ice_dyn_eap.F90:66-486).
    halo_update(a, field_loc, field_type) replaces the cyclic / open update above (a tripole domain: the caller passes the
    halo routine that IS pinned by reference output); stress_fold(array1, array2) is ice_HaloUpdate_stress
    (ice_dyn_evp.F90:416-481), called as the reference calls it when given"""
    halo = halo_update if halo_update is not None else (lambda a, loc, kind: halo_cyclic(a))
    P = set_evp_parameters(dt, ndte, revised_evp, xmin)
    ecci, revp, arlx1i, brlx, denom1 = P["ecci"], P["revp"], P["arlx1i"], P["brlx"], P["denom1"]
    ph = (slice(1, -1), slice(1, -1))
    tmask, umask = f["tmask"] != 0, f["umask"] != 0
    for n in ("rdg_conv", "rdg_shear", "divu", "shear", "prs_sig"):              # ice_dyn_evp.F90:174-182
        f[n][...] = 0.0
    if eap_tables is not None:                                                   # ice_dyn_eap.F90:171-180
        for n in EAP_HIST:
            f[n][...] = 0.0

    # ---- evp_prep1 ----
    tmass = np.where(tmask, rhoi * f["vice"] + rhos * f["vsno"], 0.0)
    tmphm = tmask & (f["aice"] > a_min) & (tmass > m_min)
    f["strairx"][...] = f["strairxT"]
    f["strairy"][...] = f["strairyT"]
    icetmask = np.zeros(tmass.shape, dtype=np.int32)
    near = np.zeros(tmass.shape, dtype=bool)
    for dj in (-1, 0, 1):
        for di in (-1, 0, 1):
            near[ph] |= tmphm[1 + dj:tmphm.shape[0] - 1 + dj, 1 + di:tmphm.shape[1] - 1 + di]
    icetmask[ph] = np.where(near[ph] & tmask[ph], 1, 0)
    f["tmass"][...] = tmass
    tmp = icetmask.astype(np.float64)
    halo(tmp, LOC_CENTER, KIND_SCALAR)                                           # :210-211
    icetmask = tmp.astype(np.int32)

    # ---- T -> U (:218-241) ----
    umass = to_ugrid(tmass, f["tarea"], f["uarea"])
    aiu = to_ugrid(f["aice_init"], f["tarea"], f["uarea"])
    for n in ("strairx", "strairy"):                                             # t2ugrid_vector
        w1 = f[n].copy()
        halo(w1, LOC_CENTER, KIND_VECTOR)
        f[n][...] = to_ugrid(w1, f["tarea"], f["uarea"])
    f["umass"][...] = umass
    f["aiu"][...] = aiu

    # ---- evp_prep2 ----
    S = [f[f"{k}_{c}"] for k in ("stressp", "stressm", "stress12") for c in (1, 2, 3, 4)]
    for s in S:
        if revp == 1.0:
            s[...] = 0.0
        else:
            s[icetmask == 0] = 0.0
    tcell = np.zeros(tmass.shape, dtype=bool)                                    # the T-cell list: jlo..jhi+1, ilo..ihi+1
    tcell[1:, 1:] = icetmask[1:, 1:] == 1
    old = f["iceumask"] != 0
    ium = np.zeros(tmass.shape, dtype=bool)
    ium[ph] = umask[ph] & (aiu[ph] > a_min) & (umass[ph] > m_min)
    u, v = f["uvel"], f["vvel"]
    new = np.zeros_like(ium)
    new[ph] = ium[ph] & ~old[ph]
    u[new] = f["uocn"][new]
    v[new] = f["vocn"][new]
    gone = np.zeros_like(ium)
    gone[ph] = ~ium[ph]
    for n in ("uvel", "vvel", "strintx", "strinty", "strocnx", "strocny"):
        f[n][gone] = 0.0
    f["iceumask"][ph] = ium[ph].astype(np.int32)
    f["uvel_init"][ph] = u[ph]
    f["vvel_init"][ph] = v[ph]
    umassdti = np.zeros_like(tmass)
    waterx, watery = np.zeros_like(tmass), np.zeros_like(tmass)
    forcex, forcey = np.zeros_like(tmass), np.zeros_like(tmass)
    umassdti[ium] = umass[ium] / dt
    f["fm"][ium] = f["fcor"][ium] * umass[ium]
    sgn = np.copysign(1.0, f["fm"])
    waterx[ium] = (f["uocn"] * cosw - f["vocn"] * sinw * sgn)[ium]
    watery[ium] = (f["vocn"] * cosw + f["uocn"] * sinw * sgn)[ium]
    f["strtltx"][ium] = (-f["fm"] * f["vocn"])[ium]                              # (not coupled / AusCOM without ocean slope)
    f["strtlty"][ium] = (f["fm"] * f["uocn"])[ium]
    forcex[ium] = (f["strairx"] + f["strtltx"])[ium]
    forcey[ium] = (f["strairy"] + f["strtlty"])[ium]
    if eap_tables is not None:                                                   # ice_dyn_eap.F90:284-298: isotropic where there is no ice
        for c in (1, 2, 3, 4):
            f[f"a11_{c}"][icetmask == 0] = p5
            f[f"a12_{c}"][icetmask == 0] = 0.0
    halo(f["strength"], LOC_CENTER, KIND_SCALAR)                                 # :311-315
    halo(u, LOC_NECORNER, KIND_VECTOR)
    halo(v, LOC_NECORNER, KIND_VECTOR)

    # ---- the subcycle loop ----
    g = {n: f[n] for n in ("cxp", "cyp", "cxm", "cym", "dxt", "dyt", "dxhy", "dyhx", "tarear", "tinyarea", "strength")}
    dtei = 1.0 / (dt / float(ndte))
    for ksub in range(1, (nsub or ndte) + 1):
        if eap_tables is None:
            strt = stress(f, g, tcell, S, u, v, ecci, arlx1i, denom1, last=(ksub == ndte))
        else:
            strt = stress_eap(f, g, tcell, S, u, v, arlx1i, denom1, eap_tables, last=(ksub == ndte))
        stepu(f, strt, ium, aiu, umassdti, waterx, watery, forcex, forcey, u, v, brlx, revp, cosw, sinw)
        if eap_tables is not None and ksub % 10 == 1:                            # ice_dyn_eap.F90:411-426
            stepa(f, tcell, S, dtei)
        halo(u, LOC_NECORNER, KIND_VECTOR)
        halo(v, LOC_NECORNER, KIND_VECTOR)

    if stress_fold is not None and eap_tables is None:                           # ice_dyn_evp.F90:416-481
        for k in (0, 4, 8):
            stress_fold(S[k + 0], S[k + 2])
            stress_fold(S[k + 2], S[k + 0])
            stress_fold(S[k + 1], S[k + 3])
            stress_fold(S[k + 3], S[k + 1])

    # ---- evp_finish + u2tgrid_vector ----
    f["strocnxT"][...] = 0.0
    f["strocnyT"][...] = 0.0
    du, dv = f["uocn"] - u, f["vocn"] - v
    vrel = rhow * f["Cdn_ocn"] * np.sqrt(du * du + dv * dv)
    vrel = vrel * aiu
    sx = vrel * (du * cosw - dv * sinw * sgn)
    sy = vrel * (dv * cosw + du * sinw * sgn)
    f["strocnx"][ium] = sx[ium]
    f["strocny"][ium] = sy[ium]
    with np.errstate(divide="ignore", invalid="ignore"):
        f["strocnxT"][ium] = (sx / aiu)[ium]
        f["strocnyT"][ium] = (sy / aiu)[ium]
    for n in ("strocnxT", "strocnyT"):
        w1 = f[n].copy()
        halo(w1, LOC_NECORNER, KIND_VECTOR)
        f[n][ph] = to_tgrid(w1, f["tarea"], f["uarea"])
    f["icetmask"][...] = icetmask
    return int(tcell[1:-1, 1:-1].sum()), int(ium.sum())


def stress(f, g, tcell, S, u, v, ecci, arlx1i, denom1, last):
    """ice_dyn_evp.F90:520-849 on every cell of the T-cell list at once; returns str(:,:,1:8)"""
    sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124 = S
    J, I = np.nonzero(tcell)
    c = (J, I)
    cxp, cyp, cxm, cym = g["cxp"][c], g["cyp"][c], g["cxm"][c], g["cym"][c]
    dxt, dyt, dxhy, dyhx = g["dxt"][c], g["dyt"][c], g["dxhy"][c], g["dyhx"][c]
    u_ij, u_mj, u_im, u_mm = u[J, I], u[J, I - 1], u[J - 1, I], u[J - 1, I - 1]
    v_ij, v_mj, v_im, v_mm = v[J, I], v[J, I - 1], v[J - 1, I], v[J - 1, I - 1]
    divune = cyp * u_ij - dyt * u_mj + cxp * v_ij - dxt * v_im
    divunw = cym * u_mj + dyt * u_ij + cxp * v_mj - dxt * v_mm
    divusw = cym * u_mm + dyt * u_im + cxm * v_mm + dxt * v_mj
    divuse = cyp * u_im - dyt * u_mm + cxm * v_im + dxt * v_ij
    tensionne = -cym * u_ij - dyt * u_mj + cxm * v_ij + dxt * v_im
    tensionnw = -cyp * u_mj + dyt * u_ij + cxm * v_mj + dxt * v_mm
    tensionsw = -cyp * u_mm + dyt * u_im + cxp * v_mm - dxt * v_mj
    tensionse = -cym * u_im - dyt * u_mm + cxp * v_im - dxt * v_ij
    shearne = -cym * v_ij - dyt * v_mj - cxm * u_ij - dxt * u_im
    shearnw = -cyp * v_mj + dyt * v_ij - cxm * u_mj - dxt * u_mm
    shearsw = -cyp * v_mm + dyt * v_im - cxp * u_mm + dxt * u_mj
    shearse = -cym * v_im - dyt * v_mm - cxp * u_im + dxt * u_ij
    Deltane = np.sqrt(divune * divune + ecci * (tensionne * tensionne + shearne * shearne))
    Deltanw = np.sqrt(divunw * divunw + ecci * (tensionnw * tensionnw + shearnw * shearnw))
    Deltase = np.sqrt(divuse * divuse + ecci * (tensionse * tensionse + shearse * shearse))
    Deltasw = np.sqrt(divusw * divusw + ecci * (tensionsw * tensionsw + shearsw * shearsw))
    if last:
        tarear = g["tarear"][c]
        divu = p25 * (divune + divunw + divuse + divusw) * tarear
        tmp = p25 * (Deltane + Deltanw + Deltase + Deltasw) * tarear
        f["divu"][c] = divu
        f["rdg_conv"][c] = -np.minimum(divu, 0.0)
        f["rdg_shear"][c] = p5 * (tmp - np.abs(divu))
        tt = tensionne + tensionnw + tensionse + tensionsw
        ss = shearne + shearnw + shearse + shearsw
        f["shear"][c] = p25 * tarear * np.sqrt(tt * tt + ss * ss)
    strength, tiny = g["strength"][c], g["tinyarea"][c]
    c0ne = strength / np.maximum(Deltane, tiny)
    c0nw = strength / np.maximum(Deltanw, tiny)
    c0sw = strength / np.maximum(Deltasw, tiny)
    c0se = strength / np.maximum(Deltase, tiny)
    f["prs_sig"][c] = c0ne * Deltane
    c1ne, c1nw, c1sw, c1se = c0ne * arlx1i, c0nw * arlx1i, c0sw * arlx1i, c0se * arlx1i
    c0ne, c0nw, c0sw, c0se = c1ne * ecci, c1nw * ecci, c1sw * ecci, c1se * ecci
    a1 = (sp1[c] + c1ne * (divune - Deltane)) * denom1
    a2 = (sp2[c] + c1nw * (divunw - Deltanw)) * denom1
    a3 = (sp3[c] + c1sw * (divusw - Deltasw)) * denom1
    a4 = (sp4[c] + c1se * (divuse - Deltase)) * denom1
    m1 = (sm1[c] + c0ne * tensionne) * denom1
    m2 = (sm2[c] + c0nw * tensionnw) * denom1
    m3 = (sm3[c] + c0sw * tensionsw) * denom1
    m4 = (sm4[c] + c0se * tensionse) * denom1
    t1 = (s121[c] + c0ne * shearne * p5) * denom1
    t2 = (s122[c] + c0nw * shearnw * p5) * denom1
    t3 = (s123[c] + c0sw * shearsw * p5) * denom1
    t4 = (s124[c] + c0se * shearse * p5) * denom1
    sp1[c], sp2[c], sp3[c], sp4[c] = a1, a2, a3, a4
    sm1[c], sm2[c], sm3[c], sm4[c] = m1, m2, m3, m4
    s121[c], s122[c], s123[c], s124[c] = t1, t2, t3, t4
    ssigpn, ssigps, ssigpe, ssigpw = a1 + a2, a3 + a4, a1 + a4, a2 + a3
    ssigp1, ssigp2 = (a1 + a3) * p055, (a2 + a4) * p055
    ssigmn, ssigms, ssigme, ssigmw = m1 + m2, m3 + m4, m1 + m4, m2 + m3
    ssigm1, ssigm2 = (m1 + m3) * p055, (m2 + m4) * p055
    ssig12n, ssig12s, ssig12e, ssig12w = t1 + t2, t3 + t4, t1 + t4, t2 + t3
    ssig121, ssig122 = (t1 + t3) * p111, (t2 + t4) * p111
    csigpne = p111 * a1 + ssigp2 + p027 * a3
    csigpnw = p111 * a2 + ssigp1 + p027 * a4
    csigpsw = p111 * a3 + ssigp2 + p027 * a1
    csigpse = p111 * a4 + ssigp1 + p027 * a2
    csigmne = p111 * m1 + ssigm2 + p027 * m3
    csigmnw = p111 * m2 + ssigm1 + p027 * m4
    csigmsw = p111 * m3 + ssigm2 + p027 * m1
    csigmse = p111 * m4 + ssigm1 + p027 * m2
    csig12ne = p222 * t1 + ssig122 + p055 * t3
    csig12nw = p222 * t2 + ssig121 + p055 * t4
    csig12sw = p222 * t3 + ssig122 + p055 * t1
    csig12se = p222 * t4 + ssig121 + p055 * t2
    str12ew = p5 * dxt * (p333 * ssig12e + p166 * ssig12w)
    str12we = p5 * dxt * (p333 * ssig12w + p166 * ssig12e)
    str12ns = p5 * dyt * (p333 * ssig12n + p166 * ssig12s)
    str12sn = p5 * dyt * (p333 * ssig12s + p166 * ssig12n)
    out = np.zeros((8,) + u.shape)
    strp = p25 * dyt * (p333 * ssigpn + p166 * ssigps)
    strm = p25 * dyt * (p333 * ssigmn + p166 * ssigms)
    out[0][c] = -strp - strm - str12ew + dxhy * (-csigpne + csigmne) + dyhx * csig12ne
    out[1][c] = strp + strm - str12we + dxhy * (-csigpnw + csigmnw) + dyhx * csig12nw
    strp = p25 * dyt * (p333 * ssigps + p166 * ssigpn)
    strm = p25 * dyt * (p333 * ssigms + p166 * ssigmn)
    out[2][c] = -strp - strm + str12ew + dxhy * (-csigpse + csigmse) + dyhx * csig12se
    out[3][c] = strp + strm + str12we + dxhy * (-csigpsw + csigmsw) + dyhx * csig12sw
    strp = p25 * dxt * (p333 * ssigpe + p166 * ssigpw)
    strm = p25 * dxt * (p333 * ssigme + p166 * ssigmw)
    out[4][c] = -strp + strm - str12ns - dyhx * (csigpne + csigmne) + dxhy * csig12ne
    out[5][c] = strp - strm - str12sn - dyhx * (csigpse + csigmse) + dxhy * csig12se
    strp = p25 * dxt * (p333 * ssigpw + p166 * ssigpe)
    strm = p25 * dxt * (p333 * ssigmw + p166 * ssigme)
    out[6][c] = -strp + strm + str12ns - dyhx * (csigpnw + csigmnw) + dxhy * csig12nw
    out[7][c] = strp - strm + str12sn - dyhx * (csigpsw + csigmsw) + dxhy * csig12sw
    return out


def stepu(f, strt, ium, aiu, umassdti, waterx, watery, forcex, forcey, u, v, brlx, revp, cosw, sinw):
    """ice_dyn_shared.F90:623-748 on every cell of the U-cell list at once"""
    J, I = np.nonzero(ium)
    c = (J, I)
    uold, vold = u[c], v[c]
    uocn, vocn, fm = f["uocn"][c], f["vocn"][c], f["fm"][c]
    du, dv = uocn - uold, vocn - vold
    vrel = aiu[c] * rhow * f["Cdn_ocn"][c] * np.sqrt(du * du + dv * dv)
    taux, tauy = vrel * waterx[c], vrel * watery[c]
    cca = (brlx + revp) * umassdti[c] + vrel * cosw
    ccb = fm + np.copysign(1.0, fm) * vrel * sinw
    ab2 = cca * cca + ccb * ccb
    sx = f["uarear"][c] * (strt[0][J, I] + strt[1][J, I + 1] + strt[2][J + 1, I] + strt[3][J + 1, I + 1])
    sy = f["uarear"][c] * (strt[4][J, I] + strt[5][J + 1, I] + strt[6][J, I + 1] + strt[7][J + 1, I + 1])
    f["strintx"][c] = sx
    f["strinty"][c] = sy
    cc1 = sx + forcex[c] + taux + umassdti[c] * (brlx * uold + revp * f["uvel_init"][c])
    cc2 = sy + forcey[c] + tauy + umassdti[c] * (brlx * vold + revp * f["vvel_init"][c])
    u[c] = (cca * cc1 + ccb * cc2) / ab2
    v[c] = (cca * cc2 - ccb * cc1) / ab2
    f["strocnx"][c] = taux
    f["strocny"][c] = tauy


def transport_upwind(f, works, dt):
    """transport_upwind without its state transforms (ice_transport_driver.F90:634-772): cell-edge velocities (:688-701), their halo
    updates (E face / N face vectors: ghost rows 0, ghost columns wrap), upwind_field (:1614-1689) of every plane of
    works[narr, ny_block, nx_block] in place on the physical cells"""
    u, v = f["uvel"], f["vvel"]
    uee, vnn = np.zeros_like(u), np.zeros_like(u)
    ph = (slice(1, -1), slice(1, -1))
    uee[ph] = p5 * (u[ph] + u[0:-2, 1:-1])
    vnn[ph] = p5 * (v[ph] + v[1:-1, 0:-2])
    halo_cyclic(uee)
    halo_cyclic(vnn)
    HTE, HTN, tarea = f["HTE"], f["HTN"], f["tarea"]

    def upwind(y1, y2, a, h):
        return p5 * dt * h * ((a + np.abs(a)) * y1 + (a - np.abs(a)) * y2)

    for n in range(works.shape[0]):
        phi = works[n]
        wa, wb = np.zeros_like(phi), np.zeros_like(phi)
        s_ = (slice(0, -1), slice(0, -1))                      # j = 1..jhi, i = 1..ihi
        wa[s_] = upwind(phi[0:-1, 0:-1], phi[0:-1, 1:], uee[s_], HTE[s_])
        wb[s_] = upwind(phi[0:-1, 0:-1], phi[1:, 0:-1], vnn[s_], HTN[s_])
        phi[ph] = phi[ph] - (wa[ph] - wa[1:-1, 0:-2] + wb[ph] - wb[0:-2, 1:-1]) / tarea[ph]


# ---------------------------------------------------------------------------------------------------------------------------
# eap(dt), kdyn = 2 (source/ice_dyn_eap.F90): stress_eap :1052-1467, update_stress_rdg :1474-1658, stepa :1664-1787,
# calc_ffrac :1795-1864.  sin / cos / atan2 are the C library's (math.*), element by element: comparable with the libm build of
# the C oracle (oracle/libevp_oracle_libm.so), not with the fixed-algorithm build the kernels share.
# ---------------------------------------------------------------------------------------------------------------------------
import math

EAP_HIST = ("e11", "e12", "e22", "s11", "s12", "s22", "yieldstress11", "yieldstress12", "yieldstress22")
pi = 3.14159265358979323846                                   # ice_constants.F90
pih, pi2 = p5 * pi, 2.0 * pi
piq = p5 * pih
puny = 1.0e-11
_sin, _cos, _atan2 = np.vectorize(math.sin), np.vectorize(math.cos), np.vectorize(math.atan2)
# the C compiler turns a sin(x), cos(x) pair into ONE sincos(x) call of libm, whose results differ from the separate functions' in
# the last bit of the odd argument (a handful of cells per run): the cross-check with the libm build of the oracle calls the same
import ctypes as _ct
import ctypes.util as _ctu
_libm = _ct.CDLL(_ctu.find_library("m") or "libm.so.6")
_libm.sincos.argtypes = [_ct.c_double, _ct.POINTER(_ct.c_double), _ct.POINTER(_ct.c_double)]
_libm.sincos.restype = None


def _sincos(x):
    x = np.asarray(x, dtype=np.float64)
    s_, c_ = np.empty_like(x), np.empty_like(x)
    a, b = _ct.c_double(), _ct.c_double()
    xf, sf, cf = x.ravel(), s_.ravel(), c_.ravel()
    for k in range(xf.size):
        _libm.sincos(float(xf[k]), _ct.byref(a), _ct.byref(b))
        sf[k], cf[k] = a.value, b.value
    return s_, c_


def update_stress_rdg(last, divu, tension, shear, a11, a12, strength, tables):
    """:1474-1658 for arrays of corners; returns stressp, stressm, stress12, alphar"""
    s11r, s12r, s22r, s11s, s12s, s22s = tables
    na_yield, ny_yield, nx_yield = s11r.shape
    kfriction = 0.45
    invstressconviso = 1.0 / (1.0 + kfriction * kfriction)
    invsin = 1.0 / math.sin(pi2 / 12.0) * invstressconviso
    a22 = 1.0 - a11
    gamma = p5 * _atan2(2.0 * a12, a11 - a22)
    Q12, Q11 = _sincos(gamma)
    Q11Q11, Q11Q12, Q12Q12 = Q11 * Q11, Q11 * Q12, Q12 * Q12
    atempprime = Q11Q11 * a11 + 2.0 * Q11Q12 * a12 + Q12Q12 * a22
    atempprime = np.maximum(atempprime, 1.0 - atempprime)
    dtemp11 = p5 * (divu + tension)
    dtemp12 = shear * p5
    dtemp22 = p5 * (divu - tension)
    alpha = p5 * _atan2(2.0 * dtemp12, dtemp11 - dtemp22)
    alpha = np.where(alpha > gamma, alpha - pi, alpha)
    alpha = np.where(alpha < gamma - pi, alpha + pi, alpha)
    y = gamma - alpha
    Qd12, Qd11 = _sincos(alpha)
    dtemp1 = Qd11 * (Qd11 * dtemp11 + 2.0 * Qd12 * dtemp12) + Qd12 * Qd12 * dtemp22
    dtemp2 = Qd12 * (Qd12 * dtemp11 - 2.0 * Qd11 * dtemp12) + Qd11 * Qd11 * dtemp22
    nz = (np.abs(dtemp1) > puny) | (np.abs(dtemp2) > puny)
    with np.errstate(divide="ignore", invalid="ignore"):
        invleng = 1.0 / np.sqrt(dtemp1 * dtemp1 + dtemp2 * dtemp2)
        d1 = np.where(nz, dtemp1 * invleng, dtemp1)
        d2 = np.where(nz, dtemp2 * invleng, dtemp2)
    x = np.where(nz, _atan2(np.where(nz, d2, 0.0), np.where(nz, d1, 1.0)), 0.0)
    x = np.where(x < piq, x + pi2, x)
    dx, dy, da = pi / float(nx_yield - 1), pi / float(ny_yield - 1), p5 / float(na_yield - 1)
    invdx, invdy, invda = 1.0 / dx, 1.0 / dy, 1.0 / da
    kx = np.trunc((x - piq - pi) * invdx).astype(np.int64) + 1
    ky = np.trunc(y * invdy).astype(np.int64) + 1
    ka = np.trunc((atempprime - p5) * invda).astype(np.int64) + 1
    ix = (ka - 1, ky - 1, kx - 1)
    t11r, t12r, t22r, t11s, t12s, t22s = s11r[ix], s12r[ix], s22r[ix], s11s[ix], s12s[ix], s22s[ix]
    stressp = strength * (t11r + kfriction * t11s + t22r + kfriction * t22s) * invsin
    stress12 = strength * (t12r + kfriction * t12s) * invsin
    stressm = strength * (t11r + kfriction * t11s - t22r - kfriction * t22s) * invsin
    sig11, sig12, sig22 = p5 * (stressp + stressm), stress12, p5 * (stressp - stressm)
    sgprm11 = Q11Q11 * sig11 + Q12Q12 * sig22 - 2.0 * Q11Q12 * sig12
    sgprm12 = Q11Q12 * sig11 - Q11Q12 * sig22 + (Q11Q11 - Q12Q12) * sig12
    sgprm22 = Q12Q12 * sig11 + Q11Q11 * sig22 + 2.0 * Q11Q12 * sig12
    alphar = None
    if last:
        r11 = Q11Q11 * t11r - 2.0 * Q11Q12 * t12r + Q12Q12 * t22r
        r12 = Q11Q11 * t12r + Q11Q12 * (t11r - t22r) - Q12Q12 * t12r
        r22 = Q12Q12 * t11r + 2.0 * Q11Q12 * t12r + Q11Q11 * t22r
        alphar = r11 * dtemp11 + 2.0 * r12 * dtemp12 + r22 * dtemp22
    return sgprm11 + sgprm22, sgprm11 - sgprm22, sgprm12, alphar


def stress_eap(f, g, tcell, S, u, v, arlx1i, denom1, tables, last):
    """:1052-1467 on every cell of the T-cell list at once; returns strtmp(:,:,1:8)"""
    sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124 = S
    J, I = np.nonzero(tcell)
    c = (J, I)
    cxp, cyp, cxm, cym = g["cxp"][c], g["cyp"][c], g["cxm"][c], g["cym"][c]
    dxt, dyt, dxhy, dyhx = g["dxt"][c], g["dyt"][c], g["dxhy"][c], g["dyhx"][c]
    tarear, strength = g["tarear"][c], g["strength"][c]
    u_ij, u_mj, u_im, u_mm = u[J, I], u[J, I - 1], u[J - 1, I], u[J - 1, I - 1]
    v_ij, v_mj, v_im, v_mm = v[J, I], v[J, I - 1], v[J - 1, I], v[J - 1, I - 1]
    divune = cyp * u_ij - dyt * u_mj + cxp * v_ij - dxt * v_im
    divunw = cym * u_mj + dyt * u_ij + cxp * v_mj - dxt * v_mm
    divusw = cym * u_mm + dyt * u_im + cxm * v_mm + dxt * v_mj
    divuse = cyp * u_im - dyt * u_mm + cxm * v_im + dxt * v_ij
    tensionne = -cym * u_ij - dyt * u_mj + cxm * v_ij + dxt * v_im
    tensionnw = -cyp * u_mj + dyt * u_ij + cxm * v_mj + dxt * v_mm
    tensionsw = -cyp * u_mm + dyt * u_im + cxp * v_mm - dxt * v_mj
    tensionse = -cym * u_im - dyt * u_mm + cxp * v_im - dxt * v_ij
    shearne = -cym * v_ij - dyt * v_mj - cxm * u_ij - dxt * u_im
    shearnw = -cyp * v_mj + dyt * v_ij - cxm * u_mj - dxt * u_mm
    shearsw = -cyp * v_mm + dyt * v_im - cxp * u_mm + dxt * u_mj
    shearse = -cym * v_im - dyt * v_mm - cxp * u_im + dxt * u_ij
    pt1, mt1, tt1, ar1 = update_stress_rdg(last, divune, tensionne, shearne, f["a11_1"][c], f["a12_1"][c], strength, tables)
    pt2, mt2, tt2, ar2 = update_stress_rdg(last, divunw, tensionnw, shearnw, f["a11_2"][c], f["a12_2"][c], strength, tables)
    pt3, mt3, tt3, ar3 = update_stress_rdg(last, divusw, tensionsw, shearsw, f["a11_3"][c], f["a12_3"][c], strength, tables)
    pt4, mt4, tt4, ar4 = update_stress_rdg(last, divuse, tensionse, shearse, f["a11_4"][c], f["a12_4"][c], strength, tables)
    if last:
        tt = tensionne + tensionnw + tensionse + tensionsw
        ss = shearne + shearnw + shearse + shearsw
        f["shear"][c] = p25 * tarear * np.sqrt(tt * tt + ss * ss)
        f["divu"][c] = p25 * (divune + divunw + divuse + divusw) * tarear
        f["rdg_conv"][c] = -np.minimum(p25 * (ar1 + ar2 + ar3 + ar4), 0.0) * tarear
    f["e11"][c] = p5 * p25 * (divune + divunw + divuse + divusw + tensionne + tensionnw + tensionse + tensionsw) * tarear
    f["e12"][c] = p5 * p25 * (shearne + shearnw + shearse + shearsw) * tarear
    f["e22"][c] = p5 * p25 * (divune + divunw + divuse + divusw - tensionne - tensionnw - tensionse - tensionsw) * tarear
    f["prs_sig"][c] = strength
    a1 = (sp1[c] + pt1 * arlx1i) * denom1
    a2 = (sp2[c] + pt2 * arlx1i) * denom1
    a3 = (sp3[c] + pt3 * arlx1i) * denom1
    a4 = (sp4[c] + pt4 * arlx1i) * denom1
    m1 = (sm1[c] + mt1 * arlx1i) * denom1
    m2 = (sm2[c] + mt2 * arlx1i) * denom1
    m3 = (sm3[c] + mt3 * arlx1i) * denom1
    m4 = (sm4[c] + mt4 * arlx1i) * denom1
    t1 = (s121[c] + tt1 * arlx1i) * denom1
    t2 = (s122[c] + tt2 * arlx1i) * denom1
    t3 = (s123[c] + tt3 * arlx1i) * denom1
    t4 = (s124[c] + tt4 * arlx1i) * denom1
    sp1[c], sp2[c], sp3[c], sp4[c] = a1, a2, a3, a4
    sm1[c], sm2[c], sm3[c], sm4[c] = m1, m2, m3, m4
    s121[c], s122[c], s123[c], s124[c] = t1, t2, t3, t4
    f["s11"][c] = p5 * p25 * (a1 + a2 + a3 + a4 + m1 + m2 + m3 + m4)
    f["s22"][c] = p5 * p25 * (a1 + a2 + a3 + a4 - m1 - m2 - m3 - m4)
    f["s12"][c] = p25 * (t1 + t2 + t3 + t4)
    f["yieldstress11"][c] = p5 * p25 * (pt1 + pt2 + pt3 + pt4 + mt1 + mt2 + mt3 + mt4)
    f["yieldstress22"][c] = p5 * p25 * (pt1 + pt2 + pt3 + pt4 - mt1 - mt2 - mt3 - mt4)
    f["yieldstress12"][c] = p25 * (tt1 + tt2 + tt3 + tt4)
    # combinations for the momentum equation (:1322-1463): as stress of evp
    ssigpn, ssigps, ssigpe, ssigpw = a1 + a2, a3 + a4, a1 + a4, a2 + a3
    ssigp1, ssigp2 = (a1 + a3) * p055, (a2 + a4) * p055
    ssigmn, ssigms, ssigme, ssigmw = m1 + m2, m3 + m4, m1 + m4, m2 + m3
    ssigm1, ssigm2 = (m1 + m3) * p055, (m2 + m4) * p055
    ssig12n, ssig12s, ssig12e, ssig12w = t1 + t2, t3 + t4, t1 + t4, t2 + t3
    ssig121, ssig122 = (t1 + t3) * p111, (t2 + t4) * p111
    csigpne = p111 * a1 + ssigp2 + p027 * a3
    csigpnw = p111 * a2 + ssigp1 + p027 * a4
    csigpsw = p111 * a3 + ssigp2 + p027 * a1
    csigpse = p111 * a4 + ssigp1 + p027 * a2
    csigmne = p111 * m1 + ssigm2 + p027 * m3
    csigmnw = p111 * m2 + ssigm1 + p027 * m4
    csigmsw = p111 * m3 + ssigm2 + p027 * m1
    csigmse = p111 * m4 + ssigm1 + p027 * m2
    csig12ne = p222 * t1 + ssig122 + p055 * t3
    csig12nw = p222 * t2 + ssig121 + p055 * t4
    csig12sw = p222 * t3 + ssig122 + p055 * t1
    csig12se = p222 * t4 + ssig121 + p055 * t2
    str12ew = p5 * dxt * (p333 * ssig12e + p166 * ssig12w)
    str12we = p5 * dxt * (p333 * ssig12w + p166 * ssig12e)
    str12ns = p5 * dyt * (p333 * ssig12n + p166 * ssig12s)
    str12sn = p5 * dyt * (p333 * ssig12s + p166 * ssig12n)
    out = np.zeros((8,) + u.shape)
    strp = p25 * dyt * (p333 * ssigpn + p166 * ssigps)
    strm = p25 * dyt * (p333 * ssigmn + p166 * ssigms)
    out[0][c] = -strp - strm - str12ew + dxhy * (-csigpne + csigmne) + dyhx * csig12ne
    out[1][c] = strp + strm - str12we + dxhy * (-csigpnw + csigmnw) + dyhx * csig12nw
    strp = p25 * dyt * (p333 * ssigps + p166 * ssigpn)
    strm = p25 * dyt * (p333 * ssigms + p166 * ssigmn)
    out[2][c] = -strp - strm + str12ew + dxhy * (-csigpse + csigmse) + dyhx * csig12se
    out[3][c] = strp + strm + str12we + dxhy * (-csigpsw + csigmsw) + dyhx * csig12sw
    strp = p25 * dxt * (p333 * ssigpe + p166 * ssigpw)
    strm = p25 * dxt * (p333 * ssigme + p166 * ssigmw)
    out[4][c] = -strp + strm - str12ns - dyhx * (csigpne + csigmne) + dxhy * csig12ne
    out[5][c] = strp - strm - str12sn - dyhx * (csigpse + csigmse) + dxhy * csig12se
    strp = p25 * dxt * (p333 * ssigpw + p166 * ssigpe)
    strm = p25 * dxt * (p333 * ssigmw + p166 * ssigme)
    out[6][c] = -strp + strm + str12ns - dyhx * (csigpnw + csigmnw) + dxhy * csig12nw
    out[7][c] = strp - strm + str12sn - dyhx * (csigpsw + csigmsw) + dxhy * csig12sw
    return out


def calc_ffrac(blockno, stressp, stressm, stress12, a1x):
    """:1795-1864"""
    kfrac, threshold = 0.001, 3.0 * 0.1
    sigma11, sigma12, sigma22 = p5 * (stressp + stressm), stress12, p5 * (stressp - stressm)
    gamma = p5 * _atan2(2.0 * sigma12, sigma11 - sigma22)
    Q12, Q11 = _sincos(gamma)
    Q11Q11, Q11Q12, Q12Q12 = Q11 * Q11, Q11 * Q12, Q12 * Q12
    sigma_1 = Q11Q11 * sigma11 + 2.0 * Q11Q12 * sigma12 + Q12Q12 * sigma22
    sigma_2 = Q12Q12 * sigma11 - 2.0 * Q11Q12 * sigma12 + Q11Q11 * sigma22
    val = kfrac * (a1x - Q12Q12) if blockno == 1 else kfrac * (a1x + Q11Q12)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = sigma_1 / sigma_2
    c1_ = (sigma_1 >= 0.0) & (sigma_2 >= 0.0)
    c2_ = (sigma_1 >= 0.0) & (sigma_2 < 0.0)
    c3_ = sigma_2 == 0.0
    c4_ = (sigma_1 <= 0.0) & (ratio <= threshold)
    return np.where(c1_, 0.0, np.where(c2_, val, np.where(c3_, 0.0, np.where(c4_, val, 0.0))))


def stepa(f, tcell, S, dtei):
    """:1664-1787"""
    sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124 = S
    c = np.nonzero(tcell)
    kth = 0.2 * 0.001
    dteikth = 1.0 / (dtei + kth)
    p5kth = p5 * kth
    for k, (sp, sm, s12) in enumerate(((sp1, sm1, s121), (sp2, sm2, s122), (sp3, sm3, s123), (sp4, sm4, s124)), start=1):
        a11, a12 = f[f"a11_{k}"], f[f"a12_{k}"]
        m11 = calc_ffrac(1, sp[c], sm[c], s12[c], a11[c])
        m12 = calc_ffrac(2, sp[c], sm[c], s12[c], a12[c])
        a11[c] = (a11[c] * dtei + p5kth - m11) * dteikth
        a12[c] = (a12[c] * dtei - m12) * dteikth
    f["a11"][c] = p25 * (f["a11_1"][c] + f["a11_2"][c] + f["a11_3"][c] + f["a11_4"][c])
    f["a12"][c] = p25 * (f["a12_1"][c] + f["a12_2"][c] + f["a12_3"][c] + f["a12_4"][c])
