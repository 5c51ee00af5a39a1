"""A SECOND, independent restatement of horizontal_remap (row f-3) -- numpy, written in round 4 straight from
source/ice_transport_remap.F90 by a reader who had not written oracle/remap_oracle.c -- for ONE block that covers the whole
domain (cyclic E-W, open N-S), l_fixed_area = .false.

Why: like `stress` and `stepu`, the remapping cannot be pinned by reference output in this image (its module needs netCDF through
ice_grid): the C restatement and the HIP kernels were checked against ONE reading of the Fortran.  This is another
one, in another shape: whole arrays for the fields, flat lists of edges for the triangles (the C oracle walks edge by edge, the
kernels thread by thread); tests/test_numpy_crosscheck.py compares it with the C oracle bit for bit.  Test infrastructure.

Arrays are (ny_block, nx_block) = (ny + 2, nx + 2), index [j, i], Fortran (i, j) -> [j - 1, i - 1].  Expressions keep the
Fortran's order of evaluation.

    horizontal_remap      :309-850      make_masks         :867-1015     construct_fields     :1024-1331
    limited_gradient      :1344-1484    departure_points   :1493-1670    locate_triangles     :1680-3047
    triangle_coordinates  :3078-3187    transport_integrals :3199-3509   update_fields        :3517-3729
"""
import numpy as np

puny, eps16 = 1.0e-11, 1.0e-16
p5, p333, p4, p6 = 0.5, 1.0 / 3.0, 0.4, 0.6
p5625m, p52083 = -9.0 / 16.0, 25.0 / 48.0


def halo_cyclic(a):
    """ice_HaloUpdate, one block, cyclic E-W, open N-S: ghost rows <- 0, ghost columns wrap over all rows"""
    a[0, :] = 0.0
    a[-1, :] = 0.0
    a[:, 0] = a[:, -2]
    a[:, -1] = a[:, 1]


def limited_gradient(phi, phimask, cnx, cny):
    """:1344-1484 -- on the physical cells where phimask > puny; zero elsewhere"""
    gx, gy = np.zeros_like(phi), np.zeros_like(phi)
    c = (slice(1, -1), slice(1, -1))

    def sh(dj, di):
        return (slice(1 + dj, phi.shape[0] - 1 + dj), slice(1 + di, phi.shape[1] - 1 + di))

    ph = phi[c]

    def nb(dj, di):
        s = sh(dj, di)
        return phimask[s] * phi[s] + (1.0 - phimask[s]) * ph

    phi_nw, phi_n, phi_ne = nb(1, -1), nb(1, 0), nb(1, 1)
    phi_w, phi_e = nb(0, -1), nb(0, 1)
    phi_sw, phi_s, phi_se = nb(-1, -1), nb(-1, 0), nb(-1, 1)
    gxtmp = (phi_e - phi_w) * p5
    gytmp = (phi_n - phi_s) * p5
    allv = [phi_nw, phi_n, phi_ne, phi_w, ph, phi_e, phi_sw, phi_s, phi_se]
    pmn = np.minimum.reduce(allv) - ph
    pmx = np.maximum.reduce(allv) - ph
    cx, cy = cnx[c], cny[c]
    w1 = (p5 - cx) * gxtmp + (p5 - cy) * gytmp
    w2 = (p5 - cx) * gxtmp - (p5 + cy) * gytmp
    w3 = -(p5 + cx) * gxtmp - (p5 + cy) * gytmp
    w4 = (p5 - cy) * gytmp - (p5 + cx) * gxtmp
    qmn = np.minimum.reduce([w1, w2, w3, w4])
    qmx = np.maximum.reduce([w1, w2, w3, w4])
    with np.errstate(divide="ignore", invalid="ignore"):
        a1 = np.where(np.abs(qmn) > np.abs(pmn), np.maximum(0.0, pmn / qmn), 1.0)
        a2 = np.where(np.abs(qmx) > np.abs(pmx), np.maximum(0.0, pmx / qmx), 1.0)
    lim = np.minimum(a1, a2)
    on = phimask[c] > puny
    gx[c] = np.where(on, lim * gxtmp, 0.0)
    gy[c] = np.where(on, lim * gytmp, 0.0)
    return gx, gy


def construct_fields(mm, hm, mmask, tm=None, tmask=None, ttype=None, depend=None, has=None):
    """:1024-1331 for one category: mc, mx, my (and tc, tx, ty of its tracers); xav = yav = 0, xxav = yyav = 1/12 (init_remap)"""
    xav = np.zeros_like(mm)
    yav = np.zeros_like(mm)
    xxav = np.full_like(mm, 1.0 / 12.0)
    yyav = np.full_like(mm, 1.0 / 12.0)
    mx, my = limited_gradient(mm, hm, xav, yav)
    ice = np.zeros(mm.shape, dtype=bool)
    if tm is None:                                  # category 0: the list of make_masks over ALL cells (ghost cells are halo-updated later)
        ice[...] = mm > puny
    else:
        ice[1:-1, 1:-1] = mm[1:-1, 1:-1] > puny
    mc = np.where(ice, mm, 0.0)
    if tm is None:
        return mc, mx, my
    mxav, myav = np.zeros_like(mm), np.zeros_like(mm)
    with np.errstate(divide="ignore", invalid="ignore"):
        mxav[ice] = ((mx * xxav + mc * xav) / mm)[ice]
        myav[ice] = ((my * yyav + mc * yav) / mm)[ice]
    ntrace = tm.shape[0]
    tc, tx, ty = np.zeros_like(tm), np.zeros_like(tm), np.zeros_like(tm)
    mtxav, mtyav = np.zeros_like(tm), np.zeros_like(tm)
    for nt in range(ntrace):
        if ttype[nt] == 1:
            tx[nt], ty[nt] = limited_gradient(tm[nt], mmask, mxav, myav)
            tc[nt][ice] = (tm[nt] - tx[nt] * mxav - ty[nt] * myav)[ice]
            if has[nt]:
                on = ice & (tmask[nt] > puny)
                w1 = mc * tc[nt]
                w2 = mc * tx[nt] + mx * tc[nt]
                w3 = mc * ty[nt] + my * tc[nt]
                with np.errstate(divide="ignore", invalid="ignore"):
                    w7 = 1.0 / (mm * tm[nt])
                    mtxav[nt][on] = ((w1 * xav + w2 * xxav) * w7)[on]
                    mtyav[nt][on] = ((w1 * yav + w3 * yyav) * w7)[on]
        elif ttype[nt] == 2:
            nt1 = depend[nt] - 1
            tx[nt], ty[nt] = limited_gradient(tm[nt], tmask[nt1], mtxav[nt1], mtyav[nt1])
            tc[nt][ice] = (tm[nt] - tx[nt] * mtxav[nt1] - ty[nt] * mtyav[nt1])[ice]
        else:
            tc[nt][ice] = tm[nt][ice]
    return mc, mx, my, tc, tx, ty


def departure_points(u, v, dxu, dyu, HTN, HTE, dt, midpt):
    """:1493-1670; returns dpx, dpy and the stop flag"""
    dpx, dpy = np.zeros_like(u), np.zeros_like(u)
    c = (slice(1, -1), slice(1, -1))
    e, n = (slice(1, -1), slice(2, None)), (slice(2, None), slice(1, -1))
    dpx[c] = -dt * u[c]
    dpy[c] = -dt * v[c]
    stop = bool(((dpx[c] < -HTN[c]) | (dpx[c] > HTN[e]) | (dpy[c] < -HTE[c]) | (dpy[c] > HTE[n])).any())
    if stop or not midpt:
        return dpx, dpy, stop
    J, I = np.nonzero((u[c] != 0.0) | (v[c] != 0.0))
    J, I = J + 1, I + 1
    dx = dpx[J, I] / dxu[J, I]
    dy = dpy[J, I] / dyu[J, I]
    mpx, mpy = p5 * dx, p5 * dy
    east, north = mpx >= 0.0, mpy >= 0.0
    i2 = np.where(east, I + 1, I)
    j2 = np.where(north, J + 1, J)
    mpxt = np.where(east, mpx - p5, mpx + p5)
    mpyt = np.where(north, mpy - p5, mpy + p5)

    def mid(a):
        return (a[j2 - 1, i2 - 1] * (mpxt - p5) * (mpyt - p5) - a[j2 - 1, i2] * (mpxt + p5) * (mpyt - p5)
                + a[j2, i2] * (mpxt + p5) * (mpyt + p5) - a[j2, i2 - 1] * (mpxt - p5) * (mpyt + p5))

    dpx[J, I] = -dt * mid(u)
    dpy[J, I] = -dt * mid(v)
    return dpx, dpy, False


class Triangles:
    """the six triangle groups of a list of edges: vertices xp[g][v], yp[g][v] (v = 0 the centroid), area, source cell (J2, I2)"""


def locate_triangles(north, dpx, dpy, dxu, dyu, order):
    """:1680-3047 (l_fixed_area = .false.) + triangle_coordinates :3078-3187 for the east (north = False) or north edges"""
    ny, nx = dpx.shape[0] - 2, dpx.shape[1] - 2
    if north:
        jj, ii = np.meshgrid(np.arange(0, ny + 1), np.arange(1, nx + 1), indexing="ij")         # jb = jlo-1 .. jhi, ib = ilo .. ihi
        moving = (dpx[jj, ii - 1] != 0.0) | (dpy[jj, ii - 1] != 0.0) | (dpx[jj, ii] != 0.0) | (dpy[jj, ii] != 0.0)
    else:
        jj, ii = np.meshgrid(np.arange(1, ny + 1), np.arange(0, nx + 1), indexing="ij")         # jb = jlo .. jhi, ib = ilo-1 .. ihi
        moving = (dpx[jj - 1, ii] != 0.0) | (dpy[jj - 1, ii] != 0.0) | (dpx[jj, ii] != 0.0) | (dpy[jj, ii] != 0.0)
    J, I = jj[moving], ii[moving]
    with np.errstate(divide="ignore", invalid="ignore"):
        dx, dy = dpx / dxu, dpy / dyu
    if north:
        sh = dict(tl=(-1, 1), bl=(-1, 0), tr=(1, 1), br=(1, 0), tc=(0, 1), bc=(0, 0))          # (ishift, jshift)
        fl = dxu[J, I - 1] * dyu[J, I - 1]
        fr = dxu[J, I] * dyu[J, I]
    else:
        sh = dict(tl=(1, 1), bl=(0, 1), tr=(1, -1), br=(0, -1), tc=(1, 0), bc=(0, 0))
        fl = dxu[J, I] * dyu[J, I]
        fr = dxu[J - 1, I] * dyu[J - 1, I]
    fc = p5 * (fl + fr)
    ne = J.size
    xcl, ycl, xcr, ycr = np.full(ne, -p5), np.zeros(ne), np.full(ne, p5), np.zeros(ne)
    if north:
        xdl, ydl = xcl + dx[J, I - 1], ycl + dy[J, I - 1]
        xdr, ydr = xcr + dx[J, I], ycr + dy[J, I]
    else:                                            # east edge: the trajectory rotated by pi/2
        xdl, ydl = xcl - dy[J, I], ycl + dx[J, I]
        xdr, ydr = xcr - dy[J - 1, I], ycr + dx[J - 1, I]
    xdm, ydm = p5 * (xdr + xdl), p5 * (ydr + ydl)
    with np.errstate(divide="ignore", invalid="ignore"):
        xil, yil = xcl, (xcl * (ydm - ydl) + xdm * ydl - xdl * ydm) / (xdm - xdl)
        xir, yir = xcr, (xcr * (ydr - ydm) - xdm * ydr + xdr * ydm) / (xdr - xdm)
        md = (ydr - ydl) / (xdr - xdl)
        xic = np.where(np.abs(md) > puny, xdl - ydl / md, 0.0)
    yic = np.zeros(ne)
    xicl, yicl, xicr, yicr = xic, yic, xic, yic

    T = Triangles()
    T.J, T.I, T.ne = J, I, ne
    T.xp = [[np.zeros(ne) for _ in range(4)] for _ in range(6)]
    T.yp = [[np.zeros(ne) for _ in range(4)] for _ in range(6)]
    T.di = [np.zeros(ne, dtype=np.int64) for _ in range(6)]
    T.dj = [np.zeros(ne, dtype=np.int64) for _ in range(6)]
    fact = [np.zeros(ne) for _ in range(6)]

    def tri(on, ng, p1, p2, p3, where, f):
        g = ng - 1
        for v, (px, py) in enumerate((p1, p2, p3), start=1):
            T.xp[g][v] = np.where(on, px, T.xp[g][v])
            T.yp[g][v] = np.where(on, py, T.yp[g][v])
        T.di[g] = np.where(on, sh[where][0], T.di[g])
        T.dj[g] = np.where(on, sh[where][1], T.dj[g])
        fact[g] = np.where(on, f, fact[g])

    CL, CR, DL, DR = (xcl, ycl), (xcr, ycr), (xdl, ydl), (xdr, ydr)
    IL, IR, IC = (xil, yil), (xir, yir), (xic, yic)
    # ---- TL and BL triangles (:2013-2100)
    left = xdl < xcl
    a = (yil > 0.0) & left & (ydl >= 0.0)
    b = ~a & (yil < 0.0) & left & (ydl < 0.0)
    c_ = ~a & ~b & (yil < 0.0) & left & (ydl >= 0.0)
    d_ = ~a & ~b & ~c_ & (yil > 0.0) & left & (ydl < 0.0)
    tri(a, 1, CL, IL, DL, "tl", -fl)
    tri(b, 1, CL, DL, IL, "bl", fl)
    tri(c_, 1, CL, DL, IC, "tl", fl)
    tri(c_, 3, CL, IC, IL, "bl", fl)
    tri(d_, 3, CL, IL, IC, "tl", -fl)
    tri(d_, 1, CL, IC, DL, "bl", -fl)
    # ---- TR and BR triangles (:2106-2196)
    right = xdr >= xcr
    a = (yir > 0.0) & right & (ydr >= 0.0)
    b = ~a & (yir < 0.0) & right & (ydr < 0.0)
    c_ = ~a & ~b & (yir < 0.0) & right & (ydr >= 0.0)
    d_ = ~a & ~b & ~c_ & (yir > 0.0) & right & (ydr < 0.0)
    tri(a, 2, CR, DR, IR, "tr", -fr)
    tri(b, 2, CR, IR, DR, "br", fr)
    tri(c_, 2, CR, IC, DR, "tr", fr)
    tri(c_, 3, CR, IR, IC, "br", fr)
    tri(d_, 3, CR, IC, IR, "tr", -fr)
    tri(d_, 2, CR, DR, IC, "br", -fr)
    # ---- departure points outside the central cells move to the cell sides (:2202-2210)
    mv = xdl < xcl
    xdl, ydl = np.where(mv, xil, xdl), np.where(mv, yil, ydl)
    mv = xdr > xcr
    xdr, ydr = np.where(mv, xir, xdr), np.where(mv, yir, ydr)
    DL, DR, DM = (xdl, ydl), (xdr, ydr), (xdm, ydm)
    ICL, ICR = (xicl, yicl), (xicr, yicr)
    # ---- TC and BC triangles (:2378-2836): twelve exclusive cases, first match wins
    lp, rp, mp, cp = ydl >= 0.0, ydr >= 0.0, ydm >= 0.0, xic >= 0.0
    cases = [
        (lp & rp & mp,          [(4, CL, CR, DL, "tc", -fc), (5, CR, DR, DL, "tc", -fc), (6, DL, DR, DM, "tc", -fc)]),
        (lp & rp & ~mp,         [(4, CL, ICL, DL, "tc", -fc), (5, CR, DR, ICR, "tc", -fc), (6, ICR, ICL, DM, "bc", fc)]),
        (~lp & ~rp & ~mp,       [(4, CL, DL, CR, "bc", fc), (5, CR, DL, DR, "bc", fc), (6, DL, DM, DR, "bc", fc)]),
        (~lp & ~rp & mp,        [(4, CL, DL, ICL, "bc", fc), (5, CR, ICR, DR, "bc", fc), (6, ICL, ICR, DM, "tc", -fc)]),
        (lp & ~rp & cp & mp,    [(4, CL, ICR, DL, "tc", -fc), (5, CR, ICR, DR, "bc", fr), (6, DL, ICR, DM, "tc", -fc)]),
        (lp & ~rp & cp & ~mp,   [(4, CL, ICL, DL, "tc", -fc), (5, CR, ICR, DR, "bc", fr), (6, ICR, ICL, DM, "bc", fc)]),
        (lp & ~rp & ~cp & ~mp,  [(4, CL, ICL, DL, "tc", -fl), (5, CR, ICL, DR, "bc", fc), (6, DR, ICL, DM, "bc", fc)]),
        (lp & ~rp & ~cp & mp,   [(4, CL, ICL, DL, "tc", -fl), (5, CR, ICR, DR, "bc", fc), (6, ICL, ICR, DM, "tc", -fc)]),
        (~lp & rp & ~cp & mp,   [(4, CL, DL, ICL, "bc", fl), (5, CR, DR, ICL, "tc", -fc), (6, ICL, DR, DM, "tc", -fc)]),
        (~lp & rp & ~cp & ~mp,  [(4, CL, DL, ICL, "bc", fl), (5, CR, DR, ICR, "tc", -fc), (6, ICR, ICL, DM, "bc", fc)]),
        (~lp & rp & cp & ~mp,   [(4, CL, DL, ICR, "bc", fc), (5, CR, DR, ICR, "tc", -fr), (6, ICR, DL, DM, "bc", fc)]),
        (~lp & rp & cp & mp,    [(4, CL, DL, ICL, "bc", fc), (5, CR, DR, ICR, "tc", -fr), (6, ICL, ICR, DM, "tc", -fc)]),
    ]
    taken = np.zeros(ne, dtype=bool)
    for cond, tris in cases:
        on = cond & ~taken
        taken |= on
        for (ng, p1, p2, p3, where, f) in tris:
            tri(on, ng, p1, p2, p3, where, f)
    # ---- areas (:2868-2897), then coordinates relative to the source cell (:2943-2975), then triangle_coordinates
    T.area, T.J2, T.I2 = [], [], []
    for g in range(6):
        xp, yp = T.xp[g], T.yp[g]
        ar = p5 * ((xp[2] - xp[1]) * (yp[3] - yp[1]) - (yp[2] - yp[1]) * (xp[3] - xp[1])) * fact[g]
        ar = np.where(np.abs(ar) < eps16 * fc, 0.0, ar)
        T.area.append(ar)
        live = ar != 0.0
        for v in (1, 2, 3):
            if north:
                nxp = xp[v] - 1.0 * T.di[g]
                nyp = yp[v] + p5 - 1.0 * T.dj[g]
            else:
                nxp = yp[v] + p5 - 1.0 * T.di[g]
                nyp = -xp[v] - 1.0 * T.dj[g]
            xp[v], yp[v] = np.where(live, nxp, xp[v]), np.where(live, nyp, yp[v])
        x0 = p333 * (xp[1] + xp[2] + xp[3])
        y0 = p333 * (yp[1] + yp[2] + yp[3])
        xp[0], yp[0] = np.where(live, x0, 0.0), np.where(live, y0, 0.0)
        if order == 2:
            for v in (1, 2, 3):
                xp[v] = np.where(live, p5 * xp[v] + p5 * xp[0], xp[v])
                yp[v] = np.where(live, p5 * yp[v] + p5 * yp[0], yp[v])
        elif order != 1:
            for v in (1, 2, 3):
                xp[v] = np.where(live, p4 * xp[v] + p6 * xp[0], xp[v])
                yp[v] = np.where(live, p4 * yp[v] + p6 * yp[0], yp[v])
        T.J2.append(J + T.dj[g])
        T.I2.append(I + T.di[g])
    return T


def transport_integrals(T, order, mc, mx, my, tc=None, tx=None, ty=None, ttype=None, depend=None):
    """:3199-3509 -- the mass transport across every edge of the list, and mass * tracer transports; returned as planes"""
    shape = mc.shape
    mflx = np.zeros(T.ne)
    ntrace = 0 if tc is None else tc.shape[0]
    mtflx = np.zeros((ntrace, T.ne))
    mtsum, mtxsum, mtysum = np.zeros((ntrace, T.ne)), np.zeros((ntrace, T.ne)), np.zeros((ntrace, T.ne))
    for g in range(6):
        live = T.area[g] != 0.0
        J2, I2 = T.J2[g], T.I2[g]
        xp, yp = T.xp[g], T.yp[g]
        c0, cx, cy = mc[J2, I2], mx[J2, I2], my[J2, I2]
        if order == 1:
            m0 = c0 + xp[0] * cx + yp[0] * cy
            msum = m0
            mxsum = m0 * xp[0]
            mxxsum = mxsum * xp[0]
            mxysum = mxsum * yp[0]
            mysum = m0 * yp[0]
            myysum = mysum * yp[0]
        elif order == 2:
            m1 = p333 * (c0 + xp[1] * cx + yp[1] * cy)
            m2 = p333 * (c0 + xp[2] * cx + yp[2] * cy)
            m3 = p333 * (c0 + xp[3] * cx + yp[3] * cy)
            msum = m1 + m2 + m3
            w1, w2, w3 = m1 * xp[1], m2 * xp[2], m3 * xp[3]
            mxsum = w1 + w2 + w3
            mxxsum = w1 * xp[1] + w2 * xp[2] + w3 * xp[3]
            mxysum = w1 * yp[1] + w2 * yp[2] + w3 * yp[3]
            w1, w2, w3 = m1 * yp[1], m2 * yp[2], m3 * yp[3]
            mysum = w1 + w2 + w3
            myysum = w1 * yp[1] + w2 * yp[2] + w3 * yp[3]
        else:
            m0 = p5625m * (c0 + xp[0] * cx + yp[0] * cy)
            m1 = p52083 * (c0 + xp[1] * cx + yp[1] * cy)
            m2 = p52083 * (c0 + xp[2] * cx + yp[2] * cy)
            m3 = p52083 * (c0 + xp[3] * cx + yp[3] * cy)
            msum = m0 + m1 + m2 + m3
            w0, w1, w2, w3 = m0 * xp[0], m1 * xp[1], m2 * xp[2], m3 * xp[3]
            mxsum = w0 + w1 + w2 + w3
            mxxsum = w0 * xp[0] + w1 * xp[1] + w2 * xp[2] + w3 * xp[3]
            mxysum = w0 * yp[0] + w1 * yp[1] + w2 * yp[2] + w3 * yp[3]
            w0, w1, w2, w3 = m0 * yp[0], m1 * yp[1], m2 * yp[2], m3 * yp[3]
            mysum = w0 + w1 + w2 + w3
            myysum = w0 * yp[0] + w1 * yp[1] + w2 * yp[2] + w3 * yp[3]
        mflx = np.where(live, mflx + T.area[g] * msum, mflx)
        for nt in range(ntrace):
            a, b, c = tc[nt][J2, I2], tx[nt][J2, I2], ty[nt][J2, I2]
            if ttype[nt] == 1:
                s = msum * a + mxsum * b + mysum * c
                mtxsum[nt] = np.where(live, mxsum * a + mxxsum * b + mxysum * c, mtxsum[nt])
                mtysum[nt] = np.where(live, mysum * a + mxysum * b + myysum * c, mtysum[nt])
            elif ttype[nt] == 2:
                n1 = depend[nt] - 1
                s = mtsum[n1] * a + mtxsum[n1] * b + mtysum[n1] * c
            else:
                n1 = depend[nt] - 1
                s = mtsum[n1] * a
            mtsum[nt] = np.where(live, s, mtsum[nt])
            mtflx[nt] = np.where(live, mtflx[nt] + T.area[g] * s, mtflx[nt])
    out = np.zeros(shape)
    out[T.J, T.I] = mflx
    outs = np.zeros((ntrace,) + shape)
    for nt in range(ntrace):
        outs[nt][T.J, T.I] = mtflx[nt]
    return out, outs


def update_fields(mm, fe, fn, tarear, tm=None, tfe=None, tfn=None, ttype=None, depend=None):
    """:3517-3729 for one category, in place; returns True where the reference would abort (new mass < -puny)"""
    c = (slice(1, -1), slice(1, -1))
    w, s = (slice(1, -1), slice(0, -2)), (slice(0, -2), slice(1, -1))
    ntrace = 0 if tm is None else tm.shape[0]
    mtold = []
    for nt in range(ntrace):
        if ttype[nt] == 1:
            mtold.append(mm[c] * tm[nt][c])
        elif ttype[nt] == 2:
            mtold.append(mm[c] * tm[depend[nt] - 1][c] * tm[nt][c])
        else:
            n1 = depend[nt] - 1
            n2 = depend[n1] - 1
            mtold.append(mm[c] * tm[n2][c] * tm[n1][c] * tm[nt][c])
    w1 = fe[c] - fe[w] + fn[c] - fn[s]
    new = mm[c] - w1 * tarear[c]
    stop = bool((new < -puny).any())
    new = np.where(new < 0.0, 0.0, new)
    mm[c] = new
    if stop or tm is None:
        return stop
    pos = new > 0.0
    for nt in range(ntrace):
        w1 = tfe[nt][c] - tfe[nt][w] + tfn[nt][c] - tfn[nt][s]
        with np.errstate(divide="ignore", invalid="ignore"):
            if ttype[nt] == 1:
                val, on = (mtold[nt] - w1 * tarear[c]) / new, pos
            elif ttype[nt] == 2:
                t1 = tm[depend[nt] - 1][c]
                val, on = (mtold[nt] - w1 * tarear[c]) / (new * t1), pos & (np.abs(t1) > 0.0)
            else:
                n1 = depend[nt] - 1
                t1, t2 = tm[n1][c], tm[depend[n1] - 1][c]
                val, on = (mtold[nt] - w1 * tarear[c]) / (new * t2 * t1), pos & (np.abs(t1) > 0.0) & (np.abs(t2) > 0.0)
        tm[nt][c] = np.where(on, val, 0.0)
    return False


LOC_CENTER, LOC_NECORNER, KIND_SCALAR, KIND_VECTOR = 1, 2, 1, 2                   # ice_constants.F90: field_loc_*, field_type_*


def horizontal_remap(g, mm, tm, dt, ttype, depend, has, order=3, midpt=True, halo_update=None):
    """:309-850 on one block: mm (ncat + 1, ny + 2, nx + 2), tm (ncat, ntrace, ny + 2, nx + 2) in place; g: uvel, vvel, dxu, dyu,
    HTN, HTE, hm, tarear.  Returns 0, 1 (departure points out of bounds) or 2 (negative mass).
    halo_update(a, field_loc, field_type) replaces the cyclic / open update (a tripole domain: the caller passes the halo routine
    that IS pinned by reference output) -- the locations and types are those of :564-613"""
    halo = halo_update if halo_update is not None else (lambda a, loc, kind: halo_cyclic(a))
    ncat = mm.shape[0] - 1
    mmask = [(mm[n] > puny).astype(np.float64) for n in range(ncat + 1)]                         # make_masks
    tmask = [[((mm[n] > puny) & (np.abs(tm[n - 1][nt]) > puny)).astype(np.float64) if has[nt] else np.zeros_like(mm[n])
              for nt in range(tm.shape[1])] for n in range(1, ncat + 1)]
    mc, mx, my = [None] * (ncat + 1), [None] * (ncat + 1), [None] * (ncat + 1)
    tc, tx, ty = [None] * (ncat + 1), [None] * (ncat + 1), [None] * (ncat + 1)
    mc[0], mx[0], my[0] = construct_fields(mm[0], g["hm"], mmask[0])
    for n in range(1, ncat + 1):
        mc[n], mx[n], my[n], tc[n], tx[n], ty[n] = construct_fields(mm[n], g["hm"], mmask[n], tm[n - 1], tmask[n - 1], ttype, depend, has)
    dpx, dpy, stop = departure_points(g["uvel"], g["vvel"], g["dxu"], g["dyu"], g["HTN"], g["HTE"], dt, midpt)
    if stop:
        return 1
    for a in (dpx, dpy):
        halo(a, LOC_NECORNER, KIND_VECTOR)
    for a in mc:
        halo(a, LOC_CENTER, KIND_SCALAR)
    for a in mx + my:
        halo(a, LOC_CENTER, KIND_VECTOR)
    for n in range(1, ncat + 1):
        for nt in range(tc[n].shape[0]):
            halo(tc[n][nt], LOC_CENTER, KIND_SCALAR)
            halo(tx[n][nt], LOC_CENTER, KIND_VECTOR)
            halo(ty[n][nt], LOC_CENTER, KIND_VECTOR)
    flux = {}
    for north in (False, True):
        T = locate_triangles(north, dpx, dpy, g["dxu"], g["dyu"], order)
        flux[north] = [transport_integrals(T, order, mc[0], mx[0], my[0])] + \
                      [transport_integrals(T, order, mc[n], mx[n], my[n], tc[n], tx[n], ty[n], ttype, depend) for n in range(1, ncat + 1)]
    if update_fields(mm[0], flux[False][0][0], flux[True][0][0], g["tarear"]):
        return 2
    for n in range(1, ncat + 1):
        if update_fields(mm[n], flux[False][n][0], flux[True][n][0], g["tarear"], tm[n - 1], flux[False][n][1], flux[True][n][1], ttype, depend):
            return 2
    return 0
