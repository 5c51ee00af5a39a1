// evpk_remap.hip -- incremental remapping transport (source/ice_transport_remap.F90: horizontal_remap, :309-850) on the
// velocities resident on the device.  SURVEY.md S8 row f-3.  Included by evpk_api.hip (one translation unit).
//
// Layout: every advected or derived field is a plain plane indexed like the mask planes (mcell), one per (field, category
// [, tracer]); the planes of one call live in a pool addressed through a table of pointers in device memory.
//   mm[n], n = 0..ncat            mean mass (area) per category, 0 = open water     (in / out)
//   tm[n-1][nt]                   mean tracers                                        (in / out)
//   mx, my[n]; tc, tx, ty[n-1][nt]   limited gradients / centre values (construct_fields), ghost ring by k_planes_halo
//   fe, fn[n]; tfe, tfn[n-1][nt]  mass and mass*tracer transports across the east / north edges
// mc (the mass at the cell centre) is mm where mm > puny and 0 elsewhere (:1190, xav = yav = 0), on ghost cells too -- it needs no
// plane.  Kernels: k_remap_dp (departure_points) -> halo -> k_remap_construct (make_masks + construct_fields + limited_gradient,
// one thread per cell and category) -> k_planes_halo -> k_remap_flux (locate_triangles + triangle_coordinates +
// transport_integrals, one thread per edge, all categories and tracers from one set of triangles) -> k_remap_update.
// Same operation order as the Fortran, -ffp-contract=off: bit-comparable with the CPU restatement.
#pragma once

namespace evpk {

constexpr int RM_MAXT = 32;          // tracers per category
constexpr int RM_GROUPS = 6, RM_VERT = 3;
#ifndef RM_FLUX_WAVES
#define RM_FLUX_WAVES 3          // waves per SIMD the flux kernels are compiled for (register budget 512 / waves)
#endif

struct RemapTab {
    int ncat, ntrace, order, midpt;
    signed char type[RM_MAXT], dep[RM_MAXT], has[RM_MAXT];     // tracer_type, depend (1-based, 0 none), has_dependents
    // the tracers in depth-first order of the dependency forest: a type-1 tracer, then each tracer that hangs on it (type 2),
    // each followed by its own dependents (type 3).  Tracers are independent of each other apart from what they read of their
    // parents, so the order of processing is free; in THIS order the parent of a type-2 tracer is always the last type-1 tracer
    // seen and that of a type-3 tracer the last type-2 one: their values are two scalars per thread instead of arrays.
    signed char ord[RM_MAXT];
};
// pointer table: [0 .. ncat] mm, then mx, my, fe, fn (ncat+1 each); then per (n-1)*ntrace+nt: tm, tc, tx, ty, tfe, tfn
struct RemapPlanes {
    double *const *tab;
    int ncp, ntp;
    __device__ double *mm(int n) const { return tab[n]; }
    __device__ double *mx(int n) const { return tab[ncp + n]; }
    __device__ double *my(int n) const { return tab[2 * ncp + n]; }
    __device__ double *fe(int n) const { return tab[3 * ncp + n]; }
    __device__ double *fn(int n) const { return tab[4 * ncp + n]; }
    __device__ double *tm(int p) const { return tab[5 * ncp + p]; }
    __device__ double *tc(int p) const { return tab[5 * ncp + ntp + p]; }
    __device__ double *tx(int p) const { return tab[5 * ncp + 2 * ntp + p]; }
    __device__ double *ty(int p) const { return tab[5 * ncp + 3 * ntp + p]; }
    __device__ double *tfe(int p) const { return tab[5 * ncp + 4 * ntp + p]; }
    __device__ double *tfn(int p) const { return tab[5 * ncp + 5 * ntp + p]; }
};

#define RM_PUNY 1.0e-11
#define RM_EPS16 1.0e-16

// ---- departure_points (:1493-1670): dpx, dpy into two F planes (their halo is the NE-corner vector update of evp) ----
__global__ void k_remap_dp(Slab s, int SB, double dt, const double *dxu, const double *dyu, int midpt, int fdx, int fdy, unsigned *bad) {
    SLAB_IJ_ALL
    double px = 0.0, py = 0.0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {
        const double u = FD(s, SB + S_U, k), v = FD(s, SB + S_V, k);
        px = -dt * u;
        py = -dt * v;
        if (px < -FD(s, F_HTN, k) || px > FD(s, F_HTN, cell(s, i + 1, j)) || py < -FD(s, F_HTE, k) || py > FD(s, F_HTE, cell(s, i, j + 1)))
            atomicOr(bad, 1u);                                                        // :1583-1589
        if (midpt && (u != 0.0 || v != 0.0)) {                                        // :1611-1667
            px = px / dxu[km];
            py = py / dyu[km];
            const double mpx = 0.5 * px, mpy = 0.5 * py;
            int i2, j2;
            double mpxt, mpyt;
            if (mpx >= 0.0 && mpy >= 0.0) { i2 = i + 1; j2 = j + 1; mpxt = mpx - 0.5; mpyt = mpy - 0.5; }
            else if (mpx < 0.0 && mpy < 0.0) { i2 = i; j2 = j; mpxt = mpx + 0.5; mpyt = mpy + 0.5; }
            else if (mpx >= 0.0 && mpy < 0.0) { i2 = i + 1; j2 = j; mpxt = mpx - 0.5; mpyt = mpy + 0.5; }
            else { i2 = i; j2 = j + 1; mpxt = mpx + 0.5; mpyt = mpy - 0.5; }
            const size_t kmm = cell(s, i2 - 1, j2 - 1), kpm = cell(s, i2, j2 - 1), kpp = cell(s, i2, j2), kmp = cell(s, i2 - 1, j2);
            const double ump = FD(s, SB + S_U, kmm) * (mpxt - 0.5) * (mpyt - 0.5) - FD(s, SB + S_U, kpm) * (mpxt + 0.5) * (mpyt - 0.5)
                             + FD(s, SB + S_U, kpp) * (mpxt + 0.5) * (mpyt + 0.5) - FD(s, SB + S_U, kmp) * (mpxt - 0.5) * (mpyt + 0.5);
            const double vmp = FD(s, SB + S_V, kmm) * (mpxt - 0.5) * (mpyt - 0.5) - FD(s, SB + S_V, kpm) * (mpxt + 0.5) * (mpyt - 0.5)
                             + FD(s, SB + S_V, kpp) * (mpxt + 0.5) * (mpyt + 0.5) - FD(s, SB + S_V, kmp) * (mpxt - 0.5) * (mpyt + 0.5);
            px = -dt * ump;
            py = -dt * vmp;
        }
    }
    FD(s, fdx, k) = px;
    FD(s, fdy, k) = py;
}

// ---- limited_gradient (:1344-1484) at one cell from the 3x3 values around it (index (dj+1)*3 + (di+1)) ----
__device__ __forceinline__ void rm_lg9(const double *mk, const double *ph9, double cx, double cy, double &gx, double &gy) {
    const double ph = ph9[4];
#define RM_NB(q) (mk[q] * ph9[q] + (1.0 - mk[q]) * ph)
    const double phi_nw = RM_NB(6), phi_n = RM_NB(7), phi_ne = RM_NB(8);
    const double phi_w = RM_NB(3), phi_e = RM_NB(5);
    const double phi_sw = RM_NB(0), phi_s = RM_NB(1), phi_se = RM_NB(2);
#undef RM_NB
    const double gxtmp = (phi_e - phi_w) * 0.5, gytmp = (phi_n - phi_s) * 0.5;
    double pmn = fmin(fmin(fmin(fmin(fmin(fmin(fmin(fmin(phi_nw, phi_n), phi_ne), phi_w), ph), phi_e), phi_sw), phi_s), phi_se);
    double pmx = fmax(fmax(fmax(fmax(fmax(fmax(fmax(fmax(phi_nw, phi_n), phi_ne), phi_w), ph), phi_e), phi_sw), phi_s), phi_se);
    pmn = pmn - ph;
    pmx = pmx - ph;
    double w1 = (0.5 - cx) * gxtmp + (0.5 - cy) * gytmp;
    double w2 = (0.5 - cx) * gxtmp - (0.5 + cy) * gytmp;
    const double w3 = -(0.5 + cx) * gxtmp - (0.5 + cy) * gytmp;
    const double w4 = (0.5 - cy) * gytmp - (0.5 + cx) * gxtmp;
    const double qmn = fmin(fmin(fmin(w1, w2), w3), w4), qmx = fmax(fmax(fmax(w1, w2), w3), w4);
    if (fabs(qmn) > fabs(pmn)) w1 = fmax(0.0, pmn / qmn); else w1 = 1.0;           // :1459-1468
    if (fabs(qmx) > fabs(pmx)) w2 = fmax(0.0, pmx / qmx); else w2 = 1.0;
    w1 = fmin(w1, w2);
    gx = w1 * gxtmp;
    gy = w1 * gytmp;
}

// ---- make_masks (:867-1015) + construct_fields (:1024-1331): one thread per cell, blockIdx.z = category ----
// The masks are functions of mm and tm at the nine stencil cells, which the thread holds in registers (mm once per category, the
// tmask of the current type-1 tracer for the tracers that hang on it).  Cells without ice of the category (every mask 0, tc = tx =
// ty = 0 in the reference) write NOTHING: the flux kernel takes the zeros from the mass plane's own test (mm <= puny), so most
// of the 3 ntrace ncat planes are never touched where there is no ice.  The nine values of the next tracer are in flight while
// the gradient of this one is computed.
__global__ void __launch_bounds__(256) k_remap_construct(Slab s, RemapTab t, RemapPlanes P, const double *hm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, n = blockIdx.z;
    if (i > s.nxl + 1 || j > s.nyl + 1) return;
    const size_t km = mcell(s, i, j);
    const long pitch = s.pitch;
#define RM_K9(q) (size_t)((long)km + ((q) / 3 - 1) * pitch + ((q) % 3 - 1))          // = mcell(s, i + q%3 - 1, j + q/3 - 1)
    const double xxav = 1.0 / 12.0, yyav = 1.0 / 12.0, xav = 0.0, yav = 0.0;          // init_remap, :249-289
    const double *mm = P.mm(n);
    const bool phys = (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl);
    double m9[9], mk[9];
    double mx = 0.0, my = 0.0;
    const bool ocean = phys && hm[km] > RM_PUNY;
    const bool ice = phys && mm[km] > RM_PUNY;                                        // the cell list of construct_fields
    if (ocean || ice) {
#pragma unroll
        for (int q = 0; q < 9; q++) m9[q] = mm[RM_K9(q)];
    }
    if (ocean) {                                                                      // limited_gradient(mm, hm, xav, yav), :1186-1191
#pragma unroll
        for (int q = 0; q < 9; q++) mk[q] = hm[RM_K9(q)];
        rm_lg9(mk, m9, xav, yav, mx, my);
    }
    P.mx(n)[km] = mx;
    P.my(n)[km] = my;
    if (n == 0 || !ice) return;
    const double mc = m9[4];                                                          // :1198-1202 (xav = yav = 0)
    const double mxav = (mx * xxav + mc * xav) / m9[4];                               // :1212-1215
    const double myav = (my * yyav + mc * yav) / m9[4];
#pragma unroll
    for (int q = 0; q < 9; q++) mk[q] = m9[q] > RM_PUNY ? 1.0 : 0.0;                  // mmask
    double pk[9], t9[9], nx9[9];                                                      // tmask of the parent (type 1); this tracer's and the next one's values
    double ptx = 0.0, pty = 0.0;                                                      // mtxav, mtyav of the parent
    bool pmask = false;
    if (t.ntrace > 0) {
        const double *tm = P.tm((n - 1) * t.ntrace + t.ord[0]);
#pragma unroll
        for (int q = 0; q < 9; q++) nx9[q] = tm[RM_K9(q)];
    }
    for (int q0 = 0; q0 < t.ntrace; q0++) {
        const int nt = t.ord[q0], p = (n - 1) * t.ntrace + nt, ty_ = t.type[nt];
#pragma unroll
        for (int q = 0; q < 9; q++) t9[q] = nx9[q];
        if (q0 + 1 < t.ntrace) {
            const double *tm = P.tm((n - 1) * t.ntrace + t.ord[q0 + 1]);
#pragma unroll
            for (int q = 0; q < 9; q++) nx9[q] = tm[RM_K9(q)];
        }
        double tx = 0.0, ty = 0.0, tc = 0.0;
        if (ty_ == 1) {                                                               // :1219-1273
            rm_lg9(mk, t9, mxav, myav, tx, ty);
            tc = t9[4] - tx * mxav - ty * myav;
            pmask = false; ptx = 0.0; pty = 0.0;
            if (t.has[nt]) {
#pragma unroll
                for (int q = 0; q < 9; q++) pk[q] = (m9[q] > RM_PUNY && fabs(t9[q]) > RM_PUNY) ? 1.0 : 0.0;      // tmask (:971-981)
                pmask = pk[4] > RM_PUNY;
                if (pmask) {
                    const double w1 = mc * tc, w2 = mc * tx + mx * tc, w3 = mc * ty + my * tc;
                    const double w7 = 1.0 / (m9[4] * t9[4]);
                    ptx = (w1 * xav + w2 * xxav) * w7;
                    pty = (w1 * yav + w3 * yyav) * w7;
                }
            }
        } else if (ty_ == 2) {                                                        // :1275-1293
            if (pmask) {
                rm_lg9(pk, t9, ptx, pty, tx, ty);
                tc = t9[4] - tx * ptx - ty * pty;
            } else
                tc = t9[4];                                                           // (- 0 * mtxav - 0 * mtyav)
        } else {                                                                      // :1295-1303
            tc = t9[4];
        }
        P.tc(p)[km] = tc;
        P.tx(p)[km] = tx;
        P.ty(p)[km] = ty;
    }
#undef RM_K9
}

// ---- ghost ring of a list of plain planes, one rank: centre fields, scalar (sgn +1) or vector (-1) ----
// blockIdx.y = plane.  N-S: south fill, north fill or the centre u-fold ghost(i, ny+1) = sgn * P(nx-i+1, ny); then E-W over all
// rows (cyclic wrap or fill).  One launch does both: a thread owning a ghost-row cell of a ghost column computes it from the source.
__global__ void k_planes_halo(Slab s, double *const *planes, const signed char *sgn, int cyclic, int tripole) {
    double *P = planes[blockIdx.y];
    const double sg = (double)sgn[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int ncol = s.nxl + 2, nrow = s.nyl + 2;
    const int nx = s.nxl;                                      // one rank: nxl == nx_global
    auto north = [&](int g) -> double {                         // value of the north ghost row at global column g (1..nx)
        return tripole ? sg * P[mcell(s, nx - g + 1, s.nyl)] : 0.0;
    };
    if (t < ncol) {                                             // south and north ghost rows, all columns
        const int i = t;
        int g = i; if (g < 1) g += nx; if (g > nx) g -= nx;
        const bool inside = (i >= 1 && i <= nx) || cyclic;
        P[mcell(s, i, 0)] = 0.0;
        P[mcell(s, i, s.nyl + 1)] = inside ? north(g) : 0.0;
    } else if (t < ncol + nrow - 2) {                           // west and east ghost columns, physical rows
        const int j = t - ncol + 1;
        P[mcell(s, 0, j)] = cyclic ? P[mcell(s, nx, j)] : 0.0;
        P[mcell(s, nx + 1, j)] = cyclic ? P[mcell(s, 1, j)] : 0.0;
    }
}

// ---- locate_triangles (:1680-3047) + triangle_coordinates (:3078-3187) for one edge; l_fixed_area = .false. ----
struct RmTri { double xp[RM_GROUPS][RM_VERT + 1], yp[RM_GROUPS][RM_VERT + 1], area[RM_GROUPS]; int di[RM_GROUPS], dj[RM_GROUPS]; };

template <bool NORTH, typename FDP>
__device__ __forceinline__ void rm_edge_triangles(int i, int j, FDP dxy /* (ii, jj, &dx, &dy): scaled departure point */,
                                                  double areafac_l, double areafac_r, int order, RmTri &t) {
    // source cells relative to the edge: TL, BL, TR, BR, TC, BC (:1823-1876)
    const int tl0 = NORTH ? -1 : 1, tl1 = 1, bl0 = NORTH ? -1 : 0, bl1 = NORTH ? 0 : 1, tr0 = 1, tr1 = NORTH ? 1 : -1;
    const int br0 = NORTH ? 1 : 0, br1 = NORTH ? 0 : -1, tc0 = NORTH ? 0 : 1, tc1 = NORTH ? 1 : 0, bc0 = 0, bc1 = 0;
    const double areafac_c = 0.5 * (areafac_l + areafac_r);
    double fact[RM_GROUPS];
#pragma unroll
    for (int g = 0; g < RM_GROUPS; g++) {
        fact[g] = 0.0; t.area[g] = 0.0; t.di[g] = 0; t.dj[g] = 0;
#pragma unroll
        for (int v = 0; v <= RM_VERT; v++) { t.xp[g][v] = 0.0; t.yp[g][v] = 0.0; }
    }
#define RM_TRI(NG, X1, Y1, X2, Y2, X3, Y3, S0, S1, FAC) do { t.xp[NG - 1][1] = X1; t.yp[NG - 1][1] = Y1; t.xp[NG - 1][2] = X2; t.yp[NG - 1][2] = Y2; \
        t.xp[NG - 1][3] = X3; t.yp[NG - 1][3] = Y3; t.di[NG - 1] = S0; t.dj[NG - 1] = S1; fact[NG - 1] = FAC; } while (0)
    const double xcl = -0.5, ycl = 0.0, xcr = 0.5, ycr = 0.0;
    double xdl, ydl, xdr, ydr, ax, ay, bx, by;
    if (NORTH) { dxy(i - 1, j, ax, ay); dxy(i, j, bx, by); xdl = xcl + ax; ydl = ycl + ay; xdr = xcr + bx; ydr = ycr + by; }      // :1958-1963
    else { dxy(i, j, ax, ay); dxy(i, j - 1, bx, by); xdl = xcl - ay; ydl = ycl + ax; xdr = xcr - by; ydr = ycr + bx; }           // :1965-1968
    const double xdm = 0.5 * (xdr + xdl), ydm = 0.5 * (ydr + ydl);
    const double xil = xcl, yil = (xcl * (ydm - ydl) + xdm * ydl - xdl * ydm) / (xdm - xdl);
    const double xir = xcr, yir = (xcr * (ydr - ydm) - xdm * ydr + xdr * ydm) / (xdr - xdm);
    const double md = (ydr - ydl) / (xdr - xdl);
    const double xic = fabs(md) > RM_PUNY ? xdl - ydl / md : 0.0, yic = 0.0;
    const double xicl = xic, yicl = yic, xicr = xic, yicr = yic;
    // TL and BL triangles (:2013-2100)
    if (yil > 0.0 && xdl < xcl && ydl >= 0.0) RM_TRI(1, xcl, ycl, xil, yil, xdl, ydl, tl0, tl1, -areafac_l);
    else if (yil < 0.0 && xdl < xcl && ydl < 0.0) RM_TRI(1, xcl, ycl, xdl, ydl, xil, yil, bl0, bl1, areafac_l);
    else if (yil < 0.0 && xdl < xcl && ydl >= 0.0) {
        RM_TRI(1, xcl, ycl, xdl, ydl, xic, yic, tl0, tl1, areafac_l);
        RM_TRI(3, xcl, ycl, xic, yic, xil, yil, bl0, bl1, areafac_l);
    } else if (yil > 0.0 && xdl < xcl && ydl < 0.0) {
        RM_TRI(3, xcl, ycl, xil, yil, xic, yic, tl0, tl1, -areafac_l);
        RM_TRI(1, xcl, ycl, xic, yic, xdl, ydl, bl0, bl1, -areafac_l);
    }
    // TR and BR triangles (:2106-2196)
    if (yir > 0.0 && xdr >= xcr && ydr >= 0.0) RM_TRI(2, xcr, ycr, xdr, ydr, xir, yir, tr0, tr1, -areafac_r);
    else if (yir < 0.0 && xdr >= xcr && ydr < 0.0) RM_TRI(2, xcr, ycr, xir, yir, xdr, ydr, br0, br1, areafac_r);
    else if (yir < 0.0 && xdr >= xcr && ydr >= 0.0) {
        RM_TRI(2, xcr, ycr, xic, yic, xdr, ydr, tr0, tr1, areafac_r);
        RM_TRI(3, xcr, ycr, xir, yir, xic, yic, br0, br1, areafac_r);
    } else if (yir > 0.0 && xdr >= xcr && ydr < 0.0) {
        RM_TRI(3, xcr, ycr, xic, yic, xir, yir, tr0, tr1, -areafac_r);
        RM_TRI(2, xcr, ycr, xdr, ydr, xic, yic, br0, br1, -areafac_r);
    }
    if (xdl < xcl) { xdl = xil; ydl = yil; }                                          // :2202-2210
    if (xdr > xcr) { xdr = xir; ydr = yir; }
    // TC and BC triangles (:2378-2836)
    if (ydl >= 0.0 && ydr >= 0.0 && ydm >= 0.0) {
        RM_TRI(4, xcl, ycl, xcr, ycr, xdl, ydl, tc0, tc1, -areafac_c);
        RM_TRI(5, xcr, ycr, xdr, ydr, xdl, ydl, tc0, tc1, -areafac_c);
        RM_TRI(6, xdl, ydl, xdr, ydr, xdm, ydm, tc0, tc1, -areafac_c);
    } else if (ydl >= 0.0 && ydr >= 0.0 && ydm < 0.0) {
        RM_TRI(4, xcl, ycl, xicl, yicl, xdl, ydl, tc0, tc1, -areafac_c);
        RM_TRI(5, xcr, ycr, xdr, ydr, xicr, yicr, tc0, tc1, -areafac_c);
        RM_TRI(6, xicr, yicr, xicl, yicl, xdm, ydm, bc0, bc1, areafac_c);
    } else if (ydl < 0.0 && ydr < 0.0 && ydm < 0.0) {
        RM_TRI(4, xcl, ycl, xdl, ydl, xcr, ycr, bc0, bc1, areafac_c);
        RM_TRI(5, xcr, ycr, xdl, ydl, xdr, ydr, bc0, bc1, areafac_c);
        RM_TRI(6, xdl, ydl, xdm, ydm, xdr, ydr, bc0, bc1, areafac_c);
    } else if (ydl < 0.0 && ydr < 0.0 && ydm >= 0.0) {
        RM_TRI(4, xcl, ycl, xdl, ydl, xicl, yicl, bc0, bc1, areafac_c);
        RM_TRI(5, xcr, ycr, xicr, yicr, xdr, ydr, bc0, bc1, areafac_c);
        RM_TRI(6, xicl, yicl, xicr, yicr, xdm, ydm, tc0, tc1, -areafac_c);
    } else if (ydl >= 0.0 && ydr < 0.0 && xic >= 0.0 && ydm >= 0.0) {
        RM_TRI(4, xcl, ycl, xicr, yicr, xdl, ydl, tc0, tc1, -areafac_c);
        RM_TRI(5, xcr, ycr, xicr, yicr, xdr, ydr, bc0, bc1, areafac_r);
        RM_TRI(6, xdl, ydl, xicr, yicr, xdm, ydm, tc0, tc1, -areafac_c);
    } else if (ydl >= 0.0 && ydr < 0.0 && xic >= 0.0 && ydm < 0.0) {
        RM_TRI(4, xcl, ycl, xicl, yicl, xdl, ydl, tc0, tc1, -areafac_c);
        RM_TRI(5, xcr, ycr, xicr, yicr, xdr, ydr, bc0, bc1, areafac_r);
        RM_TRI(6, xicr, yicr, xicl, yicl, xdm, ydm, bc0, bc1, areafac_c);
    } else if (ydl >= 0.0 && ydr < 0.0 && xic < 0.0 && ydm < 0.0) {
        RM_TRI(4, xcl, ycl, xicl, yicl, xdl, ydl, tc0, tc1, -areafac_l);
        RM_TRI(5, xcr, ycr, xicl, yicl, xdr, ydr, bc0, bc1, areafac_c);
        RM_TRI(6, xdr, ydr, xicl, yicl, xdm, ydm, bc0, bc1, areafac_c);
    } else if (ydl >= 0.0 && ydr < 0.0 && xic < 0.0 && ydm >= 0.0) {
        RM_TRI(4, xcl, ycl, xicl, yicl, xdl, ydl, tc0, tc1, -areafac_l);
        RM_TRI(5, xcr, ycr, xicr, yicr, xdr, ydr, bc0, bc1, areafac_c);
        RM_TRI(6, xicl, yicl, xicr, yicr, xdm, ydm, tc0, tc1, -areafac_c);
    } else if (ydl < 0.0 && ydr >= 0.0 && xic < 0.0 && ydm >= 0.0) {
        RM_TRI(4, xcl, ycl, xdl, ydl, xicl, yicl, bc0, bc1, areafac_l);
        RM_TRI(5, xcr, ycr, xdr, ydr, xicl, yicl, tc0, tc1, -areafac_c);
        RM_TRI(6, xicl, yicl, xdr, ydr, xdm, ydm, tc0, tc1, -areafac_c);
    } else if (ydl < 0.0 && ydr >= 0.0 && xic < 0.0 && ydm < 0.0) {
        RM_TRI(4, xcl, ycl, xdl, ydl, xicl, yicl, bc0, bc1, areafac_l);
        RM_TRI(5, xcr, ycr, xdr, ydr, xicr, yicr, tc0, tc1, -areafac_c);
        RM_TRI(6, xicr, yicr, xicl, yicl, xdm, ydm, bc0, bc1, areafac_c);
    } else if (ydl < 0.0 && ydr >= 0.0 && xic >= 0.0 && ydm < 0.0) {
        RM_TRI(4, xcl, ycl, xdl, ydl, xicr, yicr, bc0, bc1, areafac_c);
        RM_TRI(5, xcr, ycr, xdr, ydr, xicr, yicr, tc0, tc1, -areafac_r);
        RM_TRI(6, xicr, yicr, xdl, ydl, xdm, ydm, bc0, bc1, areafac_c);
    } else if (ydl < 0.0 && ydr >= 0.0 && xic >= 0.0 && ydm >= 0.0) {
        RM_TRI(4, xcl, ycl, xdl, ydl, xicl, yicl, bc0, bc1, areafac_c);
        RM_TRI(5, xcr, ycr, xdr, ydr, xicr, yicr, tc0, tc1, -areafac_r);
        RM_TRI(6, xicl, yicl, xicr, yicr, xdm, ydm, tc0, tc1, -areafac_c);
    }
#undef RM_TRI
#pragma unroll
    for (int g = 0; g < RM_GROUPS; g++) {
        double a = 0.5 * ((t.xp[g][2] - t.xp[g][1]) * (t.yp[g][3] - t.yp[g][1]) - (t.yp[g][2] - t.yp[g][1]) * (t.xp[g][3] - t.xp[g][1])) * fact[g];
        if (fabs(a) < RM_EPS16 * areafac_c) a = 0.0;                                   // :2890-2893
        t.area[g] = a;
        if (a == 0.0) continue;
#pragma unroll
        for (int v = 1; v <= RM_VERT; v++) {                                          // coordinates relative to the source cell (:2943-2975)
            if (NORTH) {
                t.xp[g][v] = t.xp[g][v] - 1.0 * t.di[g];
                t.yp[g][v] = t.yp[g][v] + 0.5 - 1.0 * t.dj[g];
            } else {
                const double w1 = t.xp[g][v];
                t.xp[g][v] = t.yp[g][v] + 0.5 - 1.0 * t.di[g];
                t.yp[g][v] = -w1 - 1.0 * t.dj[g];
            }
        }
        t.xp[g][0] = (1.0 / 3.0) * (t.xp[g][1] + t.xp[g][2] + t.xp[g][3]);            // triangle_coordinates (:3078-3187)
        t.yp[g][0] = (1.0 / 3.0) * (t.yp[g][1] + t.yp[g][2] + t.yp[g][3]);
        if (order == 2) {
#pragma unroll
            for (int v = 1; v <= RM_VERT; v++) { t.xp[g][v] = 0.5 * t.xp[g][v] + 0.5 * t.xp[g][0]; t.yp[g][v] = 0.5 * t.yp[g][v] + 0.5 * t.yp[g][0]; }
        } else if (order != 1) {
#pragma unroll
            for (int v = 1; v <= RM_VERT; v++) { t.xp[g][v] = 0.4 * t.xp[g][v] + 0.6 * t.xp[g][0]; t.yp[g][v] = 0.4 * t.yp[g][v] + 0.6 * t.yp[g][0]; }
        }
    }
}

// ---- transport_integrals (:3199-3509): one thread per (edge, category); the triangles are rebuilt per category (a few hundred
// operations against the loads of the integrals), the six mass sums of every triangle stay in registers while the tracers go by
// in dependency order (RemapTab::ord), so no per-thread array is indexed at run time ----
// put(-1, v): the mass transport of the edge; put(nt, v): mass * tracer nt.  Every one of them is delivered exactly once.
template <bool NORTH, typename PUT>
__device__ __forceinline__ void rm_flux_edge(const Slab &s, const RemapTab &tb, const RemapPlanes &P, const double *dxu, const double *dyu,
                                             int fdx, int fdy, int i, int j, int n, PUT put) {
    const size_t ka = cell(s, i, j), kb = NORTH ? cell(s, i - 1, j) : cell(s, i, j - 1);
    const bool moving = FD(s, fdx, kb) != 0.0 || FD(s, fdy, kb) != 0.0 || FD(s, fdx, ka) != 0.0 || FD(s, fdy, ka) != 0.0;   // :1911-1929
    if (!moving) {
        put(-1, 0.0);
        if (n >= 1)
            for (int nt = 0; nt < tb.ntrace; nt++) put(nt, 0.0);
        return;
    }
    double area[RM_GROUPS];
    unsigned k2[RM_GROUPS];              // (a plane has fewer than 2^32 cells: evpk_transport_remap checks)
    double ms[RM_GROUPS][6];             // msum, mxsum, mxxsum, mxysum, mysum, myysum of each triangle
    double mflx = 0.0;
    bool anyice = false;                 // does any triangle of this edge draw on a cell with ice of this category?
    bool icg[RM_GROUPS];                 // ... per triangle: tc = tx = ty = 0 where there is none (k_remap_construct does not store those zeros)
    {
        RmTri t;
        auto dxy = [&](int ii, int jj, double &dx, double &dy) {
            const size_t q = cell(s, ii, jj), qm = mcell(s, ii, jj);
            dx = FD(s, fdx, q) / dxu[qm];                                             // :1932-1937
            dy = FD(s, fdy, q) / dyu[qm];
        };
        const size_t kl = NORTH ? mcell(s, i - 1, j) : mcell(s, i, j), kr = NORTH ? mcell(s, i, j) : mcell(s, i, j - 1);
        rm_edge_triangles<NORTH>(i, j, dxy, dxu[kl] * dyu[kl], dxu[kr] * dyu[kr], tb.order, t);
        const double p5625m = -9.0 / 16.0, p52083 = 25.0 / 48.0, p333 = 1.0 / 3.0;
        const double *mm = P.mm(n), *mxp = P.mx(n), *myp = P.my(n);
#pragma unroll
        for (int g = 0; g < RM_GROUPS; g++) {
            area[g] = t.area[g];
            k2[g] = (unsigned)mcell(s, i + t.di[g], j + t.dj[g]);
            icg[g] = false;
            if (area[g] == 0.0) continue;
            const double mmv = mm[k2[g]];
            icg[g] = mmv > RM_PUNY;
            anyice = anyice || icg[g];
            const double mc = mmv > RM_PUNY ? mmv : 0.0, mx = mxp[k2[g]], my = myp[k2[g]];
            const double *xp = t.xp[g], *yp = t.yp[g];
            double msum, mxsum, mxxsum, mxysum, mysum, myysum;
            if (tb.order == 1) {
                const double m0 = mc + xp[0] * mx + yp[0] * my;
                msum = m0;
                mxsum = m0 * xp[0]; mxxsum = mxsum * xp[0]; mxysum = mxsum * yp[0];
                mysum = m0 * yp[0]; myysum = mysum * yp[0];
            } else if (tb.order == 2) {
                const double m1 = p333 * (mc + xp[1] * mx + yp[1] * my), m2 = p333 * (mc + xp[2] * mx + yp[2] * my),
                             m3 = p333 * (mc + xp[3] * mx + yp[3] * my);
                msum = m1 + m2 + m3;
                double w1 = m1 * xp[1], w2 = m2 * xp[2], w3 = m3 * xp[3];
                mxsum = w1 + w2 + w3;
                mxxsum = w1 * xp[1] + w2 * xp[2] + w3 * xp[3];
                mxysum = w1 * yp[1] + w2 * yp[2] + w3 * yp[3];
                w1 = m1 * yp[1]; w2 = m2 * yp[2]; w3 = m3 * yp[3];
                mysum = w1 + w2 + w3;
                myysum = w1 * yp[1] + w2 * yp[2] + w3 * yp[3];
            } else {
                const double m0 = p5625m * (mc + xp[0] * mx + yp[0] * my), m1 = p52083 * (mc + xp[1] * mx + yp[1] * my),
                             m2 = p52083 * (mc + xp[2] * mx + yp[2] * my), m3 = p52083 * (mc + xp[3] * mx + yp[3] * my);
                msum = m0 + m1 + m2 + m3;
                double w0 = m0 * xp[0], w1 = m1 * xp[1], w2 = m2 * xp[2], w3 = m3 * xp[3];
                mxsum = w0 + w1 + w2 + w3;
                mxxsum = w0 * xp[0] + w1 * xp[1] + w2 * xp[2] + w3 * xp[3];
                mxysum = w0 * yp[0] + w1 * yp[1] + w2 * yp[2] + w3 * yp[3];
                w0 = m0 * yp[0]; w1 = m1 * yp[1]; w2 = m2 * yp[2]; w3 = m3 * yp[3];
                mysum = w0 + w1 + w2 + w3;
                myysum = w0 * yp[0] + w1 * yp[1] + w2 * yp[2] + w3 * yp[3];
            }
            ms[g][0] = msum; ms[g][1] = mxsum; ms[g][2] = mxxsum; ms[g][3] = mxysum; ms[g][4] = mysum; ms[g][5] = myysum;
            mflx = mflx + area[g] * msum;
        }
    }
    put(-1, mflx);
    if (n == 0) return;
    if (!anyice) {
        // tc = tx = ty = 0 where mm <= puny (construct_fields): every mtsum is a sum of (finite) * 0 and every flux 0 + (+-0) = +0
        for (int nt = 0; nt < tb.ntrace; nt++) put(nt, 0.0);
        return;
    }
    double p1s[RM_GROUPS], p1x[RM_GROUPS], p1y[RM_GROUPS], p2s[RM_GROUPS];        // mtsum, mtxsum, mtysum of the last type-1 tracer; mtsum of the last type-2
#pragma unroll
    for (int g = 0; g < RM_GROUPS; g++) { p1s[g] = 0.0; p1x[g] = 0.0; p1y[g] = 0.0; p2s[g] = 0.0; }
    // the three central triangles (groups 4, 5, 6) usually draw on ONE cell (TC or BC): its tc, tx, ty are then loaded once per tracer
    const bool cen5 = area[3] != 0.0 && area[4] != 0.0 && k2[4] == k2[3], cen6 = area[3] != 0.0 && area[5] != 0.0 && k2[5] == k2[3];
    for (int q0 = 0; q0 < tb.ntrace; q0++) {
        const int nt = tb.ord[q0], p = (n - 1) * tb.ntrace + nt, ty_ = tb.type[nt];
        const double *tcp = P.tc(p), *txp = P.tx(p), *typ = P.ty(p);
        double flx = 0.0;
        double ctc = 0.0, ctx = 0.0, cty = 0.0;                   // of group 4's cell
        if (ty_ == 1) {                                                               // :3449-3468
            const bool keep = tb.has[nt] != 0;
#pragma unroll
            for (int g = 0; g < RM_GROUPS; g++) {
                if (area[g] == 0.0) continue;
                double tc, tx, ty;
                if ((g == 4 && cen5) || (g == 5 && cen6)) { tc = ctc; tx = ctx; ty = cty; }
                else { tc = icg[g] ? tcp[k2[g]] : 0.0; tx = icg[g] ? txp[k2[g]] : 0.0; ty = icg[g] ? typ[k2[g]] : 0.0; }
                if (g == 3) { ctc = tc; ctx = tx; cty = ty; }
                const double mts = ms[g][0] * tc + ms[g][1] * tx + ms[g][4] * ty;
                flx = flx + area[g] * mts;
                if (keep) {
                    p1s[g] = mts;
                    p1x[g] = ms[g][1] * tc + ms[g][2] * tx + ms[g][3] * ty;
                    p1y[g] = ms[g][4] * tc + ms[g][3] * tx + ms[g][5] * ty;
                }
            }
        } else if (ty_ == 2) {                                                        // :3470-3483
#pragma unroll
            for (int g = 0; g < RM_GROUPS; g++) {
                if (area[g] == 0.0) continue;
                double tc, tx, ty;
                if ((g == 4 && cen5) || (g == 5 && cen6)) { tc = ctc; tx = ctx; ty = cty; }
                else { tc = icg[g] ? tcp[k2[g]] : 0.0; tx = icg[g] ? txp[k2[g]] : 0.0; ty = icg[g] ? typ[k2[g]] : 0.0; }
                if (g == 3) { ctc = tc; ctx = tx; cty = ty; }
                const double mts = p1s[g] * tc + p1x[g] * tx + p1y[g] * ty;
                flx = flx + area[g] * mts;
                p2s[g] = mts;
            }
        } else {                                                                      // :3485-3497
#pragma unroll
            for (int g = 0; g < RM_GROUPS; g++) {
                if (area[g] == 0.0) continue;
                double tc;
                if ((g == 4 && cen5) || (g == 5 && cen6)) tc = ctc;
                else tc = icg[g] ? tcp[k2[g]] : 0.0;
                if (g == 3) ctc = tc;
                const double mts = p2s[g] * tc;
                flx = flx + area[g] * mts;
            }
        }
        put(nt, flx);
    }
}

// one thread per (edge, category), fluxes to the planes fe / fn, tfe / tfn (the unfused path: EVPK_REMAP_FUSED=0, ntrace > 14)
template <bool NORTH>
__global__ void __launch_bounds__(256, RM_FLUX_WAVES) k_remap_flux(Slab s, RemapTab tb, RemapPlanes P, const double *dxu, const double *dyu, int fdx, int fdy) {
    // east edges: i = 0 .. nxl, j = 1 .. nyl;  north edges: i = 1 .. nxl, j = 0 .. nyl   (:1823-1826, :1849-1852)
    const int i = blockIdx.x * blockDim.x + threadIdx.x + (NORTH ? 1 : 0);
    const int j = blockIdx.y * blockDim.y + threadIdx.y + (NORTH ? 0 : 1);
    const int n = blockIdx.z;
    if (i > s.nxl || j > s.nyl) return;
    const size_t km = mcell(s, i, j);
    rm_flux_edge<NORTH>(s, tb, P, dxu, dyu, fdx, fdy, i, j, n, [&](int nt, double v) {
        if (nt < 0) (NORTH ? P.fn(n) : P.fe(n))[km] = v;
        else (NORTH ? P.tfn((n - 1) * tb.ntrace + nt) : P.tfe((n - 1) * tb.ntrace + nt))[km] = v;
    });
}

// ---- fluxes and update of one tile in ONE kernel: the transports across the edges of a 16 x 16 tile never leave the LDS ----
// Thread (tx, ty) of tile (bx, by) stands for cell (i, j) = (15 bx + tx, 15 by + ty): it computes the transports across the EAST
// and then the NORTH edge of its cell for the mass and every tracer of category blockIdx.z into LDS ([1 + ntrace][256] doubles
// per direction), and after one barrier the threads with tx, ty >= 1 update their cells (update_fields, :3517-3729) from their
// own two edges and those of the west and south neighbour threads.  Tiles overlap by one row and one column (15 x 15 of 16 x 16
// cells are updated per tile; the edges on the rim are computed by both neighbours, same bits).  Against k_remap_flux<E>,
// k_remap_flux<N>, k_remap_update: fe, fn, tfe, tfn (2 (ncat + 1) + 2 ncat ntrace planes) are neither written nor read, and
// tc, tx, ty, mm, mx, my are read from HBM once instead of twice.  The new mass goes to the plane of fe(n) (other tiles still
// read the old one); tm is updated in place (the fluxes read tc, tx, ty, never tm).
constexpr int RM_TILE = 16;
// RmOut (direct != 0): the new mm, tm go straight into the caller's block arrays mm(nx_block, ny_block, 0:ncat, nblocks),
// tm(nx_block, ny_block, ntrace, ncat, nblocks) -- physical cells, what k_scatter_planes would deliver -- instead of into planes
// that a scatter pass then copies: one HBM round trip of every field less.  bmap: block column / row -> local block (-1: none).
struct RmOut { double *mm, *tm; const int *bmap; const BlockDesc *bd; int direct, nbxg, bsx, bsy, nxb, nyb; };

__global__ void __launch_bounds__(256, RM_FLUX_WAVES) k_remap_fluxupd(Slab s, RemapTab tb, RemapPlanes P, const double *dxu, const double *dyu, int fdx, int fdy,
                                                                        unsigned *bad, RmOut O) {
    extern __shared__ double rl[];                    // FE[1 + ntrace][256], FN[1 + ntrace][256]
    if (O.direct && (*bad & 1u)) return;              // a departure point out of bounds (k_remap_dp): the caller's arrays stay as they are
    const int tx = threadIdx.x & (RM_TILE - 1), ty = threadIdx.x >> 4, tid = threadIdx.x;
    // XCD-aware order (workgroup b runs on XCD b % 8, each XCD has its own L2): every XCD takes a contiguous run of tiles,
    // x fastest, so that tiles running side by side share the cache lines their 16-cell rows straddle and their rims
    const int nbx = (s.nxl + RM_TILE - 2) / (RM_TILE - 1), nby = (s.nyl + RM_TILE - 2) / (RM_TILE - 1);
    const int chunk = gridDim.x >> 3;
    const int t = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (t >= nbx * nby * (tb.ncat + 1)) return;       // (the whole workgroup: nobody waits at the barrier)
    const int n = t / (nbx * nby), tile = t - n * (nbx * nby);
    const int i = (tile % nbx) * (RM_TILE - 1) + tx, j = (tile / nbx) * (RM_TILE - 1) + ty;
    const int nf = 1 + tb.ntrace;
    double *const FE = rl + tid, *const FN = rl + (size_t)nf * 256 + tid;
    const bool in = (i <= s.nxl && j <= s.nyl);
    if (in && j >= 1) rm_flux_edge<false>(s, tb, P, dxu, dyu, fdx, fdy, i, j, n, [&](int nt, double v) { FE[(size_t)(nt + 1) * 256] = v; });
    if (in && i >= 1) rm_flux_edge<true>(s, tb, P, dxu, dyu, fdx, fdy, i, j, n, [&](int nt, double v) { FN[(size_t)(nt + 1) * 256] = v; });
    __syncthreads();
    if (!(in && tx >= 1 && ty >= 1 && i >= 1 && j >= 1)) return;
    // update_fields: as k_remap_update, the four transports from the LDS (west: thread tid - 1, south: tid - 16)
    const size_t k = mcell(s, i, j);
    const double tarear = FD(s, F_TAREAR, cell(s, i, j));
    const double mold = P.mm(n)[k];
    double w1 = FE[0] - FE[-1] + FN[0] - FN[-RM_TILE];
    double mnew = mold - w1 * tarear;
    if (mnew < -RM_PUNY) atomicOr(bad, 2u);
    else if (mnew < 0.0) mnew = 0.0;
    size_t oc = 0;                                    // offset of this cell inside a (ny_block, nx_block) plane of its block
    int ob = -1;                                      // its local block
    if (O.direct) {
        const int gi = s.i0 + i - 1, gj = s.j0 + j - 1;
        ob = O.bmap[((gj - 1) / O.bsy) * O.nbxg + (gi - 1) / O.bsx];
        if (ob >= 0) {
            const BlockDesc d = O.bd[ob];
            oc = (size_t)(d.jlo + (gj - d.jglob_lo) - 1) * O.nxb + (d.ilo + (gi - d.iglob_lo) - 1);
            O.mm[((size_t)ob * (tb.ncat + 1) + n) * O.nyb * O.nxb + oc] = mnew;
        }
    } else
        P.fe(n)[k] = mnew;                            // (copied over mm(n) when the launch is complete)
    if (n == 0) return;
    double o1 = 0.0, n1 = 0.0, o2 = 0.0, n2 = 0.0;
    for (int q0 = 0; q0 < tb.ntrace; q0++) {
        const int nt = tb.ord[q0], p = (n - 1) * tb.ntrace + nt, ty_ = tb.type[nt];
        double *tmp = P.tm(p);
        double v = 0.0, told = 0.0;
        if (mnew > 0.0) {
            told = tmp[k];
            const size_t o = (size_t)(nt + 1) * 256;
            w1 = FE[o] - FE[o - 1] + FN[o] - FN[o - RM_TILE];
        }
        if (ty_ == 1) {
            if (mnew > 0.0) v = (mold * told - w1 * tarear) / mnew;
            o1 = told; n1 = v;
        } else if (ty_ == 2) {
            if (mnew > 0.0 && fabs(n1) > 0.0) v = (mold * o1 * told - w1 * tarear) / (mnew * n1);
            o2 = told; n2 = v;
        } else {
            if (mnew > 0.0 && fabs(n2) > 0.0 && fabs(n1) > 0.0) v = (mold * o1 * o2 * told - w1 * tarear) / (mnew * n1 * n2);
        }
        if (O.direct) {
            if (ob >= 0) O.tm[(((size_t)ob * tb.ncat + (n - 1)) * tb.ntrace + nt) * O.nyb * O.nxb + oc] = v;
        } else
            tmp[k] = v;
    }
}


// ---- update_fields (:3517-3729): one thread per physical cell, blockIdx.z = category; tracers in dependency order ----
__global__ void __launch_bounds__(256) k_remap_update(Slab s, RemapTab tb, RemapPlanes P, unsigned *bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y * blockDim.y + threadIdx.y + 1, n = blockIdx.z;
    if (i > s.nxl || j > s.nyl) return;
    const size_t k = mcell(s, i, j), kw = mcell(s, i - 1, j), ks = mcell(s, i, j - 1);
    const double tarear = FD(s, F_TAREAR, cell(s, i, j));
    double *mmp = P.mm(n);
    const double mold = mmp[k];
    double w1 = P.fe(n)[k] - P.fe(n)[kw] + P.fn(n)[k] - P.fn(n)[ks];                  // :3605-3620
    double mnew = mold - w1 * tarear;
    if (mnew < -RM_PUNY) atomicOr(bad, 2u);
    else if (mnew < 0.0) mnew = 0.0;
    mmp[k] = mnew;
    if (n == 0) return;
    double o1 = 0.0, n1 = 0.0, o2 = 0.0, n2 = 0.0;       // old / new values of the last type-1 and type-2 tracers
    for (int q0 = 0; q0 < tb.ntrace; q0++) {
        const int nt = tb.ord[q0], p = (n - 1) * tb.ntrace + nt, ty_ = tb.type[nt];
        double *tmp = P.tm(p);
        double v = 0.0, told = 0.0;                                                   // :3658-3662
        if (mnew > 0.0) {
            told = tmp[k];
            w1 = P.tfe(p)[k] - P.tfe(p)[kw] + P.tfn(p)[k] - P.tfn(p)[ks];
        }
        if (ty_ == 1) {                                                               // mtold: :3574-3598
            if (mnew > 0.0) v = (mold * told - w1 * tarear) / mnew;
            o1 = told; n1 = v;
        } else if (ty_ == 2) {
            if (mnew > 0.0 && fabs(n1) > 0.0) v = (mold * o1 * told - w1 * tarear) / (mnew * n1);
            o2 = told; n2 = v;
        } else {
            if (mnew > 0.0 && fabs(n2) > 0.0 && fabs(n1) > 0.0) v = (mold * o1 * o2 * told - w1 * tarear) / (mnew * n1 * n2);
        }
        tmp[k] = v;
    }
}

// ---- the slices of one block array (blocks `bstride` doubles apart, slices `pstride` apart) <-> a list of plain planes, one launch:
// blockIdx.z = slice, blockIdx.y = (block, group of blockDim.y rows).  k_gather_plane / k_scatter_plane for many planes. ----
__global__ void k_gather_planes(Slab s, const BlockDesc *bd, int nxb, int nyb, const double *src, size_t pstride, size_t bstride, double *const *dst, int nrg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int b = blockIdx.y / nrg, j = (blockIdx.y % nrg) * blockDim.y + threadIdx.y + 1;
    if (i > nxb || j > nyb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    if (!gather_take(s, d, i, j, si, sj)) return;
    dst[blockIdx.z][mcell(s, si, sj)] = src[(size_t)blockIdx.z * pstride + (size_t)b * bstride + (size_t)(j - 1) * nxb + (i - 1)];
}
__global__ void k_scatter_planes(Slab s, const BlockDesc *bd, int nxb, int nyb, double *const *src, double *dst, size_t pstride, size_t bstride, int nrg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int b = blockIdx.y / nrg, j = (blockIdx.y % nrg) * blockDim.y + threadIdx.y + 1;
    if (i > nxb || j > nyb) return;
    int si, sj;
    if (!scatter_take(s, bd[b], i, j, MODE_PHYS, si, sj)) return;
    dst[(size_t)blockIdx.z * pstride + (size_t)b * bstride + (size_t)(j - 1) * nxb + (i - 1)] = src[blockIdx.z][mcell(s, si, sj)];
}

// ---- state_to_tracers (ice_transport_driver.F90:789-900) fused with the gather, tracers_to_state (:908-1003) + bound_state
// (ice_state.F90:173-238) fused with the scatter: the caller's aice0, aicen, vicen, vsnon, trcrn instead of aim / trm ----
struct RemapState {
    double *aice0, *aicen, *vicen, *vsnon, *trcrn;      // block arrays: (nb, ny, nx), (nb, ncat, ny, nx) x 3, (nb, ncat, ntrcr_dim, ny, nx)
    int ncat, ntrcr, ntrcr_dim, nt_qsno, nslyr;         // nt_qsno 1-based: tracers nt_qsno .. nt_qsno+nslyr-1 are snow enthalpies
    double shift;                                       // rhos*Lfresh (:866, :988)
};
__global__ void k_state_gather(Slab s, const BlockDesc *bd, int nxb, int nyb, RemapState io, RemapPlanes P, int nrg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int b = blockIdx.y / nrg, j = (blockIdx.y % nrg) * blockDim.y + threadIdx.y + 1, n = blockIdx.z;
    if (i > nxb || j > nyb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    if (!gather_take(s, d, i, j, si, sj)) return;
    const size_t km = mcell(s, si, sj), nblk = (size_t)nxb * nyb, kb = (size_t)(j - 1) * nxb + (i - 1);
    if (n == 0) { P.mm(0)[km] = io.aice0[(size_t)b * nblk + kb]; return; }
    const size_t kc = ((size_t)b * io.ncat + (n - 1)) * nblk + kb;
    const double a = io.aicen[kc];
    P.mm(n)[km] = a;
    const int ntrace = 2 + io.ntrcr, p0 = (n - 1) * ntrace;
    const bool ice = a > RM_PUNY;
    const double w1 = ice ? 1.0 / a : 0.0;
    P.tm(p0)[km] = ice ? io.vicen[kc] * w1 : 0.0;                                     // hice
    P.tm(p0 + 1)[km] = ice ? io.vsnon[kc] * w1 : 0.0;                                 // hsno
    for (int it = 1; it <= io.ntrcr; it++) {
        double v = 0.0;
        if (ice) {
            v = io.trcrn[(((size_t)b * io.ncat + (n - 1)) * io.ntrcr_dim + (it - 1)) * nblk + kb];
            if (it >= io.nt_qsno && it < io.nt_qsno + io.nslyr) v = v + io.shift;
        }
        P.tm(p0 + 2 + it - 1)[km] = v;
    }
}
// every cell of every block (ghost cells too: their slab cell holds the neighbour's new values or, on the slab's ring, what the
// halo update of the planes brought) whose new area is > 0; the others keep what they had, as tracers_to_state leaves them
__global__ void k_state_scatter(Slab s, const BlockDesc *bd, int nxb, int nyb, RemapState io, RemapPlanes P, int nrg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int b = blockIdx.y / nrg, j = (blockIdx.y % nrg) * blockDim.y + threadIdx.y + 1, n = blockIdx.z;
    if (i > nxb || j > nyb) return;
    const BlockDesc d = bd[b];
    int si, sj;
    if (!scatter_take(s, d, i, j, MODE_ALL, si, sj)) return;
    const size_t km = mcell(s, si, sj), nblk = (size_t)nxb * nyb, kb = (size_t)(j - 1) * nxb + (i - 1);
    if (n == 0) {                                                                     // aice0 = aim(:,:,0); not part of bound_state
        if (i >= d.ilo && i <= d.ihi && j >= d.jlo && j <= d.jhi) io.aice0[(size_t)b * nblk + kb] = P.mm(0)[km];
        return;
    }
    const double a = P.mm(n)[km];
    if (!(a > 0.0)) return;
    const size_t kc = ((size_t)b * io.ncat + (n - 1)) * nblk + kb;
    const int ntrace = 2 + io.ntrcr, p0 = (n - 1) * ntrace;
    io.aicen[kc] = a;
    io.vicen[kc] = a * P.tm(p0)[km];
    io.vsnon[kc] = a * P.tm(p0 + 1)[km];
    for (int it = 1; it <= io.ntrcr; it++) {
        double v = P.tm(p0 + 2 + it - 1)[km];
        if (it >= io.nt_qsno && it < io.nt_qsno + io.nslyr) v = v - io.shift;
        io.trcrn[(((size_t)b * io.ncat + (n - 1)) * io.ntrcr_dim + (it - 1)) * nblk + kb] = v;
    }
}

// ---- frame (two outermost rows / columns) of a list of planes <-> consecutive planes of the pair-interleaved slab, where
// the general halo update works (several ranks, forced exchange); back: the ghost ring only ----
__global__ void k_planes_frame(Slab s, double *const *planes, int f0, int nf, int back) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int ncol = s.nxl + 2, nrow = s.nyl + 2;
    int i, j;
    if (t < 4 * ncol) { const int r = t / ncol; i = t % ncol; j = r == 0 ? 0 : r == 1 ? 1 : r == 2 ? s.nyl : s.nyl + 1; }
    else if (t < 4 * ncol + 4 * nrow) { const int u = t - 4 * ncol, q = u / nrow; j = u % nrow; i = q == 0 ? 0 : q == 1 ? 1 : q == 2 ? s.nxl : s.nxl + 1; }
    else return;
    if (back && !(i == 0 || i == s.nxl + 1 || j == 0 || j == s.nyl + 1)) return;
    const size_t k = cell(s, i, j), km = mcell(s, i, j);
    for (int q = 0; q < nf; q++) {
        if (back) planes[q][km] = FD(s, f0 + q, k);
        else FD(s, f0 + q, k) = planes[q][km];
    }
}

}  // namespace evpk
