/*
 * oracle/remap_oracle.c -- CPU restatement of the incremental-remapping transport of CICE5
 * (source/ice_transport_remap.F90: horizontal_remap and its subroutines), plain C99.
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see evp_oracle.h): the reference ships no vectors for this path and cannot
 * be built here; what pins this file are the properties the scheme is built to have (conservation, monotonicity,
 * exactness for uniform fields, tests/test_oracle.py) and decomposition invariance.
 *
 * Every routine follows the operation order of the Fortran it cites (-ffp-contract=off); Fortran evaluates a + b + c as
 * (a + b) + c and a*b*c as (a*b)*c, and so does the C below.  Arrays are (nx, ny) blocks, i fastest, 1-based through IX().
 */
#include "evp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IX(i, j) ((size_t)((j)-1) * (size_t)nx + (size_t)((i)-1))

static const double c0 = 0.0, c1 = 1.0, p5 = 0.5;
static const double puny = 1.0e-11, eps16 = 1.0e-16;
#define P333 (1.0 / 3.0)
static const double p4 = 0.4, p6 = 0.6;
static const double p5625m = -9.0 / 16.0, p52083 = 25.0 / 48.0;      /* ice_transport_remap.F90:47-50 */
enum { NGROUPS = 6, NVERT = 3 };                                       /* :42-45 */

static double min2(double a, double b) { return a < b ? a : b; }
static double max2(double a, double b) { return a > b ? a : b; }

/* ---------------------------------------------------------------------------
 * limited_gradient  (:1344-1484): van Leer limited gradient of phi where phimask > puny; cnx, cny = centre of the
 * weighting field (NULL: zero)
 * ------------------------------------------------------------------------- */
static void limited_gradient(int nx, int ny, int ilo, int ihi, int jlo, int jhi, const double *phi, const double *phimask,
                             const double *cnx, const double *cny, double *gx, double *gy) {
    const size_t nn = (size_t)nx * ny;
    for (size_t k = 0; k < nn; k++) { gx[k] = c0; gy[k] = c0; }
    for (int j = jlo; j <= jhi; j++)                                  /* nghost = 1: jlo-nghost+1 .. jhi+nghost-1 */
        for (int i = ilo; i <= ihi; i++) {
            if (!(phimask[IX(i, j)] > puny)) continue;
            const double ph = phi[IX(i, j)];
#define NB(ii, jj) (phimask[IX(ii, jj)] * phi[IX(ii, jj)] + (c1 - phimask[IX(ii, jj)]) * ph)
            const double phi_nw = NB(i - 1, j + 1), phi_n = NB(i, j + 1), phi_ne = NB(i + 1, j + 1);
            const double phi_w = NB(i - 1, j), phi_e = NB(i + 1, j);
            const double phi_sw = NB(i - 1, j - 1), phi_s = NB(i, j - 1), phi_se = NB(i + 1, j - 1);
#undef NB
            const double gxtmp = (phi_e - phi_w) * p5, gytmp = (phi_n - phi_s) * p5;
            double pmn = min2(min2(min2(min2(min2(min2(min2(min2(phi_nw, phi_n), phi_ne), phi_w), ph), phi_e), phi_sw), phi_s), phi_se);
            double pmx = max2(max2(max2(max2(max2(max2(max2(max2(phi_nw, phi_n), phi_ne), phi_w), ph), phi_e), phi_sw), phi_s), phi_se);
            pmn = pmn - ph;
            pmx = pmx - ph;
            const double cx = cnx ? cnx[IX(i, j)] : c0, cy = cny ? cny[IX(i, j)] : c0;
            double w1 = (p5 - cx) * gxtmp + (p5 - cy) * gytmp;
            double w2 = (p5 - cx) * gxtmp - (p5 + cy) * gytmp;
            const double w3 = -(p5 + cx) * gxtmp - (p5 + cy) * gytmp;
            const double w4 = (p5 - cy) * gytmp - (p5 + cx) * gxtmp;
            const double qmn = min2(min2(min2(w1, w2), w3), w4), qmx = max2(max2(max2(w1, w2), w3), w4);
            if (fabs(qmn) > fabs(pmn)) w1 = max2(c0, pmn / qmn); else w1 = c1;       /* :1459-1468 */
            if (fabs(qmx) > fabs(pmx)) w2 = max2(c0, pmx / qmx); else w2 = c1;
            w1 = min2(w1, w2);
            gx[IX(i, j)] = w1 * gxtmp;
            gy[IX(i, j)] = w1 * gytmp;
        }
}

/* ---------------------------------------------------------------------------
 * make_masks (:867-1015) for one category: mmask, tmask on the whole block (ghost cells included), and the cell test
 * "mm > puny on physical cells" that construct_fields loops over afterwards
 * ------------------------------------------------------------------------- */
static void make_masks(int nx, int ny, int ntrace, const int32_t *has_dependents, const double *mm, double *mmask,
                       const double *const *tm, double *const *tmask) {
    const size_t nn = (size_t)nx * ny;
    for (size_t k = 0; k < nn; k++) mmask[k] = mm[k] > puny ? c1 : c0;
    if (tm)
        for (int nt = 0; nt < ntrace; nt++) {
            for (size_t k = 0; k < nn; k++) tmask[nt][k] = c0;
            if (has_dependents[nt])
                for (size_t k = 0; k < nn; k++)
                    if (mm[k] > puny && fabs(tm[nt][k]) > puny) tmask[nt][k] = c1;
        }
}

/* ---------------------------------------------------------------------------
 * construct_fields (:1024-1331) for one category; xav = yav = 0, xxav = yyav = 1/12 (init_remap, :249-289)
 * tracer indices nt are 0-based here, depend[] holds the reference's 1-based index (0 = none)
 * ------------------------------------------------------------------------- */
static void construct_fields(int nx, int ny, int ilo, int ihi, int jlo, int jhi, int ntrace, const int32_t *tracer_type,
                             const int32_t *depend, const int32_t *has_dependents, const double *hm, const double *mm,
                             double *mc, double *mx, double *my, const double *mmask, const double *const *tm, double *const *tc,
                             double *const *tx, double *const *ty, const double *const *tmask) {
    const size_t nn = (size_t)nx * ny;
    const double xav = c0, yav = c0, xxav = c1 / 12.0, yyav = c1 / 12.0;
    double *mxav = calloc(nn, 8), *myav = calloc(nn, 8);
    double **mtxav = NULL, **mtyav = NULL;
    for (size_t k = 0; k < nn; k++) { mc[k] = c0; mx[k] = c0; my[k] = c0; }
    if (tm) {
        mtxav = calloc((size_t)ntrace, sizeof(double *)); mtyav = calloc((size_t)ntrace, sizeof(double *));
        for (int nt = 0; nt < ntrace; nt++) {
            mtxav[nt] = calloc(nn, 8); mtyav[nt] = calloc(nn, 8);
            for (size_t k = 0; k < nn; k++) { tc[nt][k] = c0; tx[nt][k] = c0; ty[nt][k] = c0; }
        }
    }
    limited_gradient(nx, ny, ilo, ihi, jlo, jhi, mm, hm, NULL, NULL, mx, my);          /* :1186-1191 (xav = yav = 0) */
#define CELLS for (int j = jlo; j <= jhi; j++) for (int i = ilo; i <= ihi; i++) if (mm[IX(i, j)] > puny)
    CELLS mc[IX(i, j)] = mm[IX(i, j)];                                                /* :1198-1202: mc = mm - xav*mx - yav*my */
    if (tm) {
        CELLS {
            const size_t k = IX(i, j);
            mxav[k] = (mx[k] * xxav + mc[k] * xav) / mm[k];                            /* :1212-1215 */
            myav[k] = (my[k] * yyav + mc[k] * yav) / mm[k];
        }
        for (int nt = 0; nt < ntrace; nt++) {
            if (tracer_type[nt] == 1) {                                               /* :1219-1273 */
                limited_gradient(nx, ny, ilo, ihi, jlo, jhi, tm[nt], mmask, mxav, myav, tx[nt], ty[nt]);
                CELLS {
                    const size_t k = IX(i, j);
                    tc[nt][k] = tm[nt][k] - tx[nt][k] * mxav[k] - ty[nt][k] * myav[k];
                    if (has_dependents[nt] && tmask[nt][k] > puny) {
                        const double w1 = mc[k] * tc[nt][k];
                        const double w2 = mc[k] * tx[nt][k] + mx[k] * tc[nt][k];
                        const double w3 = mc[k] * ty[nt][k] + my[k] * tc[nt][k];
                        const double w7 = c1 / (mm[k] * tm[nt][k]);
                        mtxav[nt][k] = (w1 * xav + w2 * xxav) * w7;
                        mtyav[nt][k] = (w1 * yav + w3 * yyav) * w7;
                    }
                }
            } else if (tracer_type[nt] == 2) {                                        /* :1275-1293 */
                const int nt1 = depend[nt] - 1;
                limited_gradient(nx, ny, ilo, ihi, jlo, jhi, tm[nt], tmask[nt1], mtxav[nt1], mtyav[nt1], tx[nt], ty[nt]);
                CELLS {
                    const size_t k = IX(i, j);
                    tc[nt][k] = tm[nt][k] - tx[nt][k] * mtxav[nt1][k] - ty[nt][k] * mtyav[nt1][k];
                }
            } else if (tracer_type[nt] == 3) {                                        /* :1295-1303: upwind approximation */
                CELLS tc[nt][IX(i, j)] = tm[nt][IX(i, j)];
            }
        }
        for (int nt = 0; nt < ntrace; nt++) { free(mtxav[nt]); free(mtyav[nt]); }
        free(mtxav); free(mtyav);
    }
#undef CELLS
    free(mxav); free(myav);
}

/* ---------------------------------------------------------------------------
 * departure_points (:1493-1670); returns 1 if a departure point leaves the neighbouring cells
 * ------------------------------------------------------------------------- */
static int departure_points(int nx, int ny, int ilo, int ihi, int jlo, int jhi, double dt, const double *uvel, const double *vvel,
                            const double *dxu, const double *dyu, const double *HTN, const double *HTE, double *dpx, double *dpy,
                            int l_dp_midpt) {
    const size_t nn = (size_t)nx * ny;
    int l_stop = 0;
    for (size_t k = 0; k < nn; k++) { dpx[k] = c0; dpy[k] = c0; }
    for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++) {
            const size_t k = IX(i, j);
            dpx[k] = -dt * uvel[k];
            dpy[k] = -dt * vvel[k];
            if (dpx[k] < -HTN[k] || dpx[k] > HTN[IX(i + 1, j)] || dpy[k] < -HTE[k] || dpy[k] > HTE[IX(i, j + 1)]) l_stop = 1;   /* :1583-1589 */
        }
    if (l_stop) return 1;
    if (l_dp_midpt)                                                                   /* :1611-1667 */
        for (int j = jlo; j <= jhi; j++)
            for (int i = ilo; i <= ihi; i++) {
                const size_t k = IX(i, j);
                if (uvel[k] != c0 || vvel[k] != c0) {
                    dpx[k] = dpx[k] / dxu[k];
                    dpy[k] = dpy[k] / dyu[k];
                    const double mpx = p5 * dpx[k], mpy = p5 * dpy[k];
                    int i2, j2;
                    double mpxt, mpyt;
                    if (mpx >= c0 && mpy >= c0) { i2 = i + 1; j2 = j + 1; mpxt = mpx - p5; mpyt = mpy - p5; }
                    else if (mpx < c0 && mpy < c0) { i2 = i; j2 = j; mpxt = mpx + p5; mpyt = mpy + p5; }
                    else if (mpx >= c0 && mpy < c0) { i2 = i + 1; j2 = j; mpxt = mpx - p5; mpyt = mpy + p5; }
                    else { i2 = i; j2 = j + 1; mpxt = mpx + p5; mpyt = mpy - p5; }
                    const double ump = uvel[IX(i2 - 1, j2 - 1)] * (mpxt - p5) * (mpyt - p5) - uvel[IX(i2, j2 - 1)] * (mpxt + p5) * (mpyt - p5)
                                     + uvel[IX(i2, j2)] * (mpxt + p5) * (mpyt + p5) - uvel[IX(i2 - 1, j2)] * (mpxt - p5) * (mpyt + p5);
                    const double vmp = vvel[IX(i2 - 1, j2 - 1)] * (mpxt - p5) * (mpyt - p5) - vvel[IX(i2, j2 - 1)] * (mpxt + p5) * (mpyt - p5)
                                     + vvel[IX(i2, j2)] * (mpxt + p5) * (mpyt + p5) - vvel[IX(i2 - 1, j2)] * (mpxt - p5) * (mpyt + p5);
                    dpx[k] = -dt * ump;
                    dpy[k] = -dt * vmp;
                }
            }
    return 0;
}

/* ---------------------------------------------------------------------------
 * locate_triangles (:1680-3047) + triangle_coordinates (:3078-3187) for ONE edge; l_fixed_area = .false. only.
 * north != 0: the north edge of cell (i,j), else its east edge.  Outputs per group ng = 0..5: xp/yp[ng][0..3], the source
 * cell (iflux, jflux) and triarea (0 if below the eps16 threshold, :2890-2893).
 * ------------------------------------------------------------------------- */
typedef struct { double xp[NGROUPS][NVERT + 1], yp[NGROUPS][NVERT + 1], triarea[NGROUPS]; int iflux[NGROUPS], jflux[NGROUPS]; } edge_tri;

static void edge_triangles(int nx, int i, int j, int north, const double *dpx, const double *dpy, const double *dxu, const double *dyu,
                           int integral_order, edge_tri *t) {
    int sh_tl[2], sh_bl[2], sh_tr[2], sh_br[2], sh_tc[2], sh_bc[2];
    double areafac_l, areafac_r;
    if (north) {                                                                      /* :1823-1847 */
        sh_tl[0] = -1; sh_tl[1] = 1; sh_bl[0] = -1; sh_bl[1] = 0; sh_tr[0] = 1; sh_tr[1] = 1; sh_br[0] = 1; sh_br[1] = 0;
        sh_tc[0] = 0; sh_tc[1] = 1; sh_bc[0] = 0; sh_bc[1] = 0;
        areafac_l = dxu[IX(i - 1, j)] * dyu[IX(i - 1, j)];
        areafac_r = dxu[IX(i, j)] * dyu[IX(i, j)];
    } else {                                                                          /* :1849-1876 */
        sh_tl[0] = 1; sh_tl[1] = 1; sh_bl[0] = 0; sh_bl[1] = 1; sh_tr[0] = 1; sh_tr[1] = -1; sh_br[0] = 0; sh_br[1] = -1;
        sh_tc[0] = 1; sh_tc[1] = 0; sh_bc[0] = 0; sh_bc[1] = 0;
        areafac_l = dxu[IX(i, j)] * dyu[IX(i, j)];
        areafac_r = dxu[IX(i, j - 1)] * dyu[IX(i, j - 1)];
    }
    const double areafac_c = p5 * (areafac_l + areafac_r);
    double areafact[NGROUPS];
    for (int ng = 0; ng < NGROUPS; ng++) {
        areafact[ng] = c0; t->triarea[ng] = c0; t->iflux[ng] = i; t->jflux[ng] = j;
        for (int nv = 0; nv <= NVERT; nv++) { t->xp[ng][nv] = c0; t->yp[ng][nv] = c0; }
    }
#define DX(ii, jj) (dpx[IX(ii, jj)] / dxu[IX(ii, jj)])                               /* :1932-1937 */
#define DY(ii, jj) (dpy[IX(ii, jj)] / dyu[IX(ii, jj)])
#define TRI(NG, X1, Y1, X2, Y2, X3, Y3, SH, FAC) do { const int g_ = (NG)-1; t->xp[g_][1] = X1; t->yp[g_][1] = Y1; t->xp[g_][2] = X2; \
        t->yp[g_][2] = Y2; t->xp[g_][3] = X3; t->yp[g_][3] = Y3; t->iflux[g_] = i + SH[0]; t->jflux[g_] = j + SH[1]; areafact[g_] = FAC; } while (0)
    const double xcl = -p5, ycl = c0, xcr = p5, ycr = c0;
    double xdl, ydl, xdr, ydr;
    if (north) { xdl = xcl + DX(i - 1, j); ydl = ycl + DY(i - 1, j); xdr = xcr + DX(i, j); ydr = ycr + DY(i, j); }    /* :1958-1963 */
    else { xdl = xcl - DY(i, j); ydl = ycl + DX(i, j); xdr = xcr - DY(i, j - 1); ydr = ycr + DX(i, j - 1); }           /* :1965-1968 */
    const double xdm = p5 * (xdr + xdl), ydm = p5 * (ydr + ydl);
    const double xil = xcl, yil = (xcl * (ydm - ydl) + xdm * ydl - xdl * ydm) / (xdm - xdl);                           /* :1979-1980 */
    const double xir = xcr, yir = (xcr * (ydr - ydm) - xdm * ydr + xdr * ydm) / (xdr - xdm);
    const double md = (ydr - ydl) / (xdr - xdl);
    double xic;
    if (fabs(md) > puny) xic = xdl - ydl / md; else xic = c0;                                                           /* :1988-1992 */
    const double yic = c0;
    const double xicl = xic, yicl = yic, xicr = xic, yicr = yic;                                                       /* (l_fixed_area = F) */
    /* TL and BL triangles (:2013-2100) */
    if (yil > c0 && xdl < xcl && ydl >= c0) TRI(1, xcl, ycl, xil, yil, xdl, ydl, sh_tl, -areafac_l);
    else if (yil < c0 && xdl < xcl && ydl < c0) TRI(1, xcl, ycl, xdl, ydl, xil, yil, sh_bl, areafac_l);
    else if (yil < c0 && xdl < xcl && ydl >= c0) {
        TRI(1, xcl, ycl, xdl, ydl, xic, yic, sh_tl, areafac_l);
        TRI(3, xcl, ycl, xic, yic, xil, yil, sh_bl, areafac_l);
    } else if (yil > c0 && xdl < xcl && ydl < c0) {
        TRI(3, xcl, ycl, xil, yil, xic, yic, sh_tl, -areafac_l);
        TRI(1, xcl, ycl, xic, yic, xdl, ydl, sh_bl, -areafac_l);
    }
    /* TR and BR triangles (:2106-2196) */
    if (yir > c0 && xdr >= xcr && ydr >= c0) TRI(2, xcr, ycr, xdr, ydr, xir, yir, sh_tr, -areafac_r);
    else if (yir < c0 && xdr >= xcr && ydr < c0) TRI(2, xcr, ycr, xir, yir, xdr, ydr, sh_br, areafac_r);
    else if (yir < c0 && xdr >= xcr && ydr >= c0) {
        TRI(2, xcr, ycr, xic, yic, xdr, ydr, sh_tr, areafac_r);
        TRI(3, xcr, ycr, xir, yir, xic, yic, sh_br, areafac_r);
    } else if (yir > c0 && xdr >= xcr && ydr < c0) {
        TRI(3, xcr, ycr, xic, yic, xir, yir, sh_tr, -areafac_r);
        TRI(2, xcr, ycr, xdr, ydr, xic, yic, sh_br, -areafac_r);
    }
    /* redefine the departure points if not in the central cells (:2202-2210) */
    if (xdl < xcl) { xdl = xil; ydl = yil; }
    if (xdr > xcr) { xdr = xir; ydr = yir; }
    /* TC and BC triangles (:2378-2836) */
    if (ydl >= c0 && ydr >= c0 && ydm >= c0) {
        TRI(4, xcl, ycl, xcr, ycr, xdl, ydl, sh_tc, -areafac_c);
        TRI(5, xcr, ycr, xdr, ydr, xdl, ydl, sh_tc, -areafac_c);
        TRI(6, xdl, ydl, xdr, ydr, xdm, ydm, sh_tc, -areafac_c);
    } else if (ydl >= c0 && ydr >= c0 && ydm < c0) {
        TRI(4, xcl, ycl, xicl, yicl, xdl, ydl, sh_tc, -areafac_c);
        TRI(5, xcr, ycr, xdr, ydr, xicr, yicr, sh_tc, -areafac_c);
        TRI(6, xicr, yicr, xicl, yicl, xdm, ydm, sh_bc, areafac_c);
    } else if (ydl < c0 && ydr < c0 && ydm < c0) {
        TRI(4, xcl, ycl, xdl, ydl, xcr, ycr, sh_bc, areafac_c);
        TRI(5, xcr, ycr, xdl, ydl, xdr, ydr, sh_bc, areafac_c);
        TRI(6, xdl, ydl, xdm, ydm, xdr, ydr, sh_bc, areafac_c);
    } else if (ydl < c0 && ydr < c0 && ydm >= c0) {
        TRI(4, xcl, ycl, xdl, ydl, xicl, yicl, sh_bc, areafac_c);
        TRI(5, xcr, ycr, xicr, yicr, xdr, ydr, sh_bc, areafac_c);
        TRI(6, xicl, yicl, xicr, yicr, xdm, ydm, sh_tc, -areafac_c);
    } else if (ydl >= c0 && ydr < c0 && xic >= c0 && ydm >= c0) {
        TRI(4, xcl, ycl, xicr, yicr, xdl, ydl, sh_tc, -areafac_c);
        TRI(5, xcr, ycr, xicr, yicr, xdr, ydr, sh_bc, areafac_r);
        TRI(6, xdl, ydl, xicr, yicr, xdm, ydm, sh_tc, -areafac_c);
    } else if (ydl >= c0 && ydr < c0 && xic >= c0 && ydm < c0) {
        TRI(4, xcl, ycl, xicl, yicl, xdl, ydl, sh_tc, -areafac_c);
        TRI(5, xcr, ycr, xicr, yicr, xdr, ydr, sh_bc, areafac_r);
        TRI(6, xicr, yicr, xicl, yicl, xdm, ydm, sh_bc, areafac_c);
    } else if (ydl >= c0 && ydr < c0 && xic < c0 && ydm < c0) {
        TRI(4, xcl, ycl, xicl, yicl, xdl, ydl, sh_tc, -areafac_l);
        TRI(5, xcr, ycr, xicl, yicl, xdr, ydr, sh_bc, areafac_c);
        TRI(6, xdr, ydr, xicl, yicl, xdm, ydm, sh_bc, areafac_c);
    } else if (ydl >= c0 && ydr < c0 && xic < c0 && ydm >= c0) {
        TRI(4, xcl, ycl, xicl, yicl, xdl, ydl, sh_tc, -areafac_l);
        TRI(5, xcr, ycr, xicr, yicr, xdr, ydr, sh_bc, areafac_c);
        TRI(6, xicl, yicl, xicr, yicr, xdm, ydm, sh_tc, -areafac_c);
    } else if (ydl < c0 && ydr >= c0 && xic < c0 && ydm >= c0) {
        TRI(4, xcl, ycl, xdl, ydl, xicl, yicl, sh_bc, areafac_l);
        TRI(5, xcr, ycr, xdr, ydr, xicl, yicl, sh_tc, -areafac_c);
        TRI(6, xicl, yicl, xdr, ydr, xdm, ydm, sh_tc, -areafac_c);
    } else if (ydl < c0 && ydr >= c0 && xic < c0 && ydm < c0) {
        TRI(4, xcl, ycl, xdl, ydl, xicl, yicl, sh_bc, areafac_l);
        TRI(5, xcr, ycr, xdr, ydr, xicr, yicr, sh_tc, -areafac_c);
        TRI(6, xicr, yicr, xicl, yicl, xdm, ydm, sh_bc, areafac_c);
    } else if (ydl < c0 && ydr >= c0 && xic >= c0 && ydm < c0) {
        TRI(4, xcl, ycl, xdl, ydl, xicr, yicr, sh_bc, areafac_c);
        TRI(5, xcr, ycr, xdr, ydr, xicr, yicr, sh_tc, -areafac_r);
        TRI(6, xicr, yicr, xdl, ydl, xdm, ydm, sh_bc, areafac_c);
    } else if (ydl < c0 && ydr >= c0 && xic >= c0 && ydm >= c0) {
        TRI(4, xcl, ycl, xdl, ydl, xicl, yicl, sh_bc, areafac_c);
        TRI(5, xcr, ycr, xdr, ydr, xicr, yicr, sh_tc, -areafac_r);
        TRI(6, xicl, yicl, xicr, yicr, xdm, ydm, sh_tc, -areafac_c);
    }
#undef TRI
#undef DX
#undef DY
    /* triangle areas, threshold (:2876-2897) */
    for (int ng = 0; ng < NGROUPS; ng++) {
        double a = p5 * ((t->xp[ng][2] - t->xp[ng][1]) * (t->yp[ng][3] - t->yp[ng][1]) - (t->yp[ng][2] - t->yp[ng][1]) * (t->xp[ng][3] - t->xp[ng][1]))
                 * areafact[ng];
        if (fabs(a) < eps16 * areafac_c) a = c0;
        t->triarea[ng] = a;
        if (a == c0) continue;
        /* coordinates relative to the source cell (:2943-2975) */
        const int ishift = t->iflux[ng] - i, jshift = t->jflux[ng] - j;
        for (int nv = 1; nv <= NVERT; nv++) {
            if (north) {
                t->xp[ng][nv] = t->xp[ng][nv] - c1 * ishift;
                t->yp[ng][nv] = t->yp[ng][nv] + p5 - c1 * jshift;
            } else {
                const double w1 = t->xp[ng][nv];
                t->xp[ng][nv] = t->yp[ng][nv] + p5 - c1 * ishift;
                t->yp[ng][nv] = -w1 - c1 * jshift;
            }
        }
        /* triangle_coordinates (:3078-3187): quadrature points */
        t->xp[ng][0] = P333 * (t->xp[ng][1] + t->xp[ng][2] + t->xp[ng][3]);
        t->yp[ng][0] = P333 * (t->yp[ng][1] + t->yp[ng][2] + t->yp[ng][3]);
        if (integral_order == 2)
            for (int nv = 1; nv <= NVERT; nv++) {
                t->xp[ng][nv] = p5 * t->xp[ng][nv] + p5 * t->xp[ng][0];
                t->yp[ng][nv] = p5 * t->yp[ng][nv] + p5 * t->yp[ng][0];
            }
        else if (integral_order != 1)
            for (int nv = 1; nv <= NVERT; nv++) {
                t->xp[ng][nv] = p4 * t->xp[ng][nv] + p6 * t->xp[ng][0];
                t->yp[ng][nv] = p4 * t->yp[ng][nv] + p6 * t->yp[ng][0];
            }
    }
}

/* ---------------------------------------------------------------------------
 * transport_integrals (:3199-3509) for ONE edge and one category: mass flux and mass*tracer fluxes through the edge
 * ------------------------------------------------------------------------- */
static void edge_integrals(int nx, int ntrace, const int32_t *tracer_type, const int32_t *depend, int integral_order, const edge_tri *t,
                           const double *mc, const double *mx, const double *my, double *mflx, const double *const *tc,
                           const double *const *tx, const double *const *ty, double *mtflx /* [ntrace] or NULL */) {
    double mtsum[64], mtxsum[64], mtysum[64];
    *mflx = c0;
    if (mtflx) for (int nt = 0; nt < ntrace; nt++) mtflx[nt] = c0;
    for (int ng = 0; ng < NGROUPS; ng++) {
        if (t->triarea[ng] == c0) continue;                                          /* the compressed lists hold triarea /= 0 only */
        const size_t k2 = IX(t->iflux[ng], t->jflux[ng]);
        const double *xp = t->xp[ng], *yp = t->yp[ng];
        double msum, mxsum, mxxsum, mxysum, mysum, myysum;
        if (integral_order == 1) {
            const double m0 = mc[k2] + xp[0] * mx[k2] + yp[0] * my[k2];
            msum = m0;
            mxsum = m0 * xp[0]; mxxsum = mxsum * xp[0]; mxysum = mxsum * yp[0];
            mysum = m0 * yp[0]; myysum = mysum * yp[0];
        } else if (integral_order == 2) {
            const double m1 = P333 * (mc[k2] + xp[1] * mx[k2] + yp[1] * my[k2]);
            const double m2 = P333 * (mc[k2] + xp[2] * mx[k2] + yp[2] * my[k2]);
            const double m3 = P333 * (mc[k2] + xp[3] * mx[k2] + yp[3] * my[k2]);
            msum = m1 + m2 + m3;
            double w1 = m1 * xp[1], w2 = m2 * xp[2], w3 = m3 * xp[3];
            mxsum = w1 + w2 + w3;
            mxxsum = w1 * xp[1] + w2 * xp[2] + w3 * xp[3];
            mxysum = w1 * yp[1] + w2 * yp[2] + w3 * yp[3];
            w1 = m1 * yp[1]; w2 = m2 * yp[2]; w3 = m3 * yp[3];
            mysum = w1 + w2 + w3;
            myysum = w1 * yp[1] + w2 * yp[2] + w3 * yp[3];
        } else {
            const double m0 = p5625m * (mc[k2] + xp[0] * mx[k2] + yp[0] * my[k2]);
            const double m1 = p52083 * (mc[k2] + xp[1] * mx[k2] + yp[1] * my[k2]);
            const double m2 = p52083 * (mc[k2] + xp[2] * mx[k2] + yp[2] * my[k2]);
            const double m3 = p52083 * (mc[k2] + xp[3] * mx[k2] + yp[3] * my[k2]);
            msum = m0 + m1 + m2 + m3;
            double w0 = m0 * xp[0], w1 = m1 * xp[1], w2 = m2 * xp[2], w3 = m3 * xp[3];
            mxsum = w0 + w1 + w2 + w3;
            mxxsum = w0 * xp[0] + w1 * xp[1] + w2 * xp[2] + w3 * xp[3];
            mxysum = w0 * yp[0] + w1 * yp[1] + w2 * yp[2] + w3 * yp[3];
            w0 = m0 * yp[0]; w1 = m1 * yp[1]; w2 = m2 * yp[2]; w3 = m3 * yp[3];
            mysum = w0 + w1 + w2 + w3;
            myysum = w0 * yp[0] + w1 * yp[1] + w2 * yp[2] + w3 * yp[3];
        }
        *mflx = *mflx + t->triarea[ng] * msum;
        if (!mtflx) continue;
        for (int nt = 0; nt < ntrace; nt++) {
            if (tracer_type[nt] == 1) {                                               /* :3449-3468 */
                mtsum[nt] = msum * tc[nt][k2] + mxsum * tx[nt][k2] + mysum * ty[nt][k2];
                mtflx[nt] = mtflx[nt] + t->triarea[ng] * mtsum[nt];
                mtxsum[nt] = mxsum * tc[nt][k2] + mxxsum * tx[nt][k2] + mxysum * ty[nt][k2];
                mtysum[nt] = mysum * tc[nt][k2] + mxysum * tx[nt][k2] + myysum * ty[nt][k2];
            } else if (tracer_type[nt] == 2) {                                        /* :3470-3483 */
                const int nt1 = depend[nt] - 1;
                mtsum[nt] = mtsum[nt1] * tc[nt][k2] + mtxsum[nt1] * tx[nt][k2] + mtysum[nt1] * ty[nt][k2];
                mtflx[nt] = mtflx[nt] + t->triarea[ng] * mtsum[nt];
            } else if (tracer_type[nt] == 3) {                                        /* :3485-3497 */
                const int nt1 = depend[nt] - 1;
                mtsum[nt] = mtsum[nt1] * tc[nt][k2];
                mtflx[nt] = mtflx[nt] + t->triarea[ng] * mtsum[nt];
            }
        }
    }
}

/* ---------------------------------------------------------------------------
 * update_fields (:3517-3729) for one category of one block; returns 1 on a negative new mass
 * mflxe/mflxn: (nx, ny) planes; mtflxe/mtflxn: [ntrace] planes
 * ------------------------------------------------------------------------- */
static int update_fields(int nx, int ny, int ilo, int ihi, int jlo, int jhi, int ntrace, const int32_t *tracer_type, const int32_t *depend,
                         const double *tarear, const double *mflxe, const double *mflxn, double *mm, double *const *mtflxe,
                         double *const *mtflxn, double *const *tm) {
    const size_t nn = (size_t)nx * ny;
    int l_stop = 0;
    double **mtold = NULL;
    if (tm) {
        mtold = calloc((size_t)ntrace, sizeof(double *));
        for (int nt = 0; nt < ntrace; nt++) {
            mtold[nt] = calloc(nn, 8);
            for (int j = jlo; j <= jhi; j++)
                for (int i = ilo; i <= ihi; i++) {
                    const size_t k = IX(i, j);
                    if (tracer_type[nt] == 1) mtold[nt][k] = mm[k] * tm[nt][k];                                          /* :3574-3580 */
                    else if (tracer_type[nt] == 2) mtold[nt][k] = mm[k] * tm[depend[nt] - 1][k] * tm[nt][k];
                    else if (tracer_type[nt] == 3) {
                        const int nt1 = depend[nt] - 1, nt2 = depend[nt1] - 1;
                        mtold[nt][k] = mm[k] * tm[nt2][k] * tm[nt1][k] * tm[nt][k];
                    }
                }
        }
    }
    for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++) {                                            /* :3605-3620 */
            const size_t k = IX(i, j);
            const double w1 = mflxe[k] - mflxe[IX(i - 1, j)] + mflxn[k] - mflxn[IX(i, j - 1)];
            mm[k] = mm[k] - w1 * tarear[k];
            if (mm[k] < -puny) l_stop = 1;
            else if (mm[k] < c0) mm[k] = c0;
        }
    if (!l_stop && tm)
        for (int nt = 0; nt < ntrace; nt++)
            for (int j = jlo; j <= jhi; j++)
                for (int i = ilo; i <= ihi; i++) {
                    const size_t k = IX(i, j);
                    const double told = tm[nt][k];
                    (void)told;
                    tm[nt][k] = c0;                                                    /* :3658-3662 */
                    if (!(mm[k] > c0)) continue;
                    const double w1 = mtflxe[nt][k] - mtflxe[nt][IX(i - 1, j)] + mtflxn[nt][k] - mtflxn[nt][IX(i, j - 1)];
                    if (tracer_type[nt] == 1) tm[nt][k] = (mtold[nt][k] - w1 * tarear[k]) / mm[k];
                    else if (tracer_type[nt] == 2) {
                        const int nt1 = depend[nt] - 1;
                        if (fabs(tm[nt1][k]) > c0) tm[nt][k] = (mtold[nt][k] - w1 * tarear[k]) / (mm[k] * tm[nt1][k]);
                    } else if (tracer_type[nt] == 3) {
                        const int nt1 = depend[nt] - 1, nt2 = depend[nt1] - 1;
                        if (fabs(tm[nt1][k]) > c0 && fabs(tm[nt2][k]) > c0)
                            tm[nt][k] = (mtold[nt][k] - w1 * tarear[k]) / (mm[k] * tm[nt2][k] * tm[nt1][k]);
                    }
                }
    if (mtold) { for (int nt = 0; nt < ntrace; nt++) free(mtold[nt]); free(mtold); }
    return l_stop;
}

/* ---------------------------------------------------------------------------
 * horizontal_remap (:309-850).  mm: (nblocks, ncat+1, ny, nx) [category 0 = open water], tm: (nblocks, ncat, ntrace, ny, nx),
 * both advanced in place on physical cells; their ghost cells must be current on entry.  l_fixed_area must be 0.
 * Returns 0, 1 (bad departure points), 2 (negative area) or 3 (unsupported option).
 * ------------------------------------------------------------------------- */
int orc_horizontal_remap(const orc_geom *g, double dt, int ncat, int ntrace, const double *uvel, const double *vvel, double *mm, double *tm,
                         int l_fixed_area, const int32_t *tracer_type, const int32_t *depend, const int32_t *has_dependents,
                         int integral_order, int l_dp_midpt, const double *HTE, const double *HTN, const double *dxu, const double *dyu,
                         const double *tarear, const double *hm) {
    if (l_fixed_area || ntrace > 64) return 3;
    const int nx = g->nx_block, ny = g->ny_block, nb = g->nblocks;
    const size_t nn = (size_t)nx * ny, tot = nn * nb;
    const int ncp = ncat + 1;
    /* fields laid out [plane][block][cell] so that a plane is one halo-updatable block array */
    double *dpx = calloc(tot, 8), *dpy = calloc(tot, 8);
    double *mc = calloc(tot * ncp, 8), *mx = calloc(tot * ncp, 8), *my = calloc(tot * ncp, 8);
    const size_t ntp = (size_t)ncat * ntrace;
    double *tc = calloc(tot * (ntp ? ntp : 1), 8), *tx = calloc(tot * (ntp ? ntp : 1), 8), *ty = calloc(tot * (ntp ? ntp : 1), 8);
    int rc = 0;
#define MM(b, n) (mm + ((size_t)(b) * ncp + (n)) * nn)
#define TM(b, n, nt) (tm + (((size_t)(b) * ncat + ((n)-1)) * ntrace + (nt)) * nn)
#define PL(a, p, b) ((a) + ((size_t)(p) * nb + (b)) * nn)
    for (int b = 0; b < nb && !rc; b++) {                                             /* :455-563 */
        const int ilo = g->ilo[b], ihi = g->ihi[b], jlo = g->jlo[b], jhi = g->jhi[b];
        const size_t o = (size_t)b * nn;
        double *mmask = malloc(nn * 8);
        double **tmk = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *));
        const double **tmp = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *));
        double **tcp = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *)), **txp = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *)),
               **typ = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *));
        for (int nt = 0; nt < ntrace; nt++) tmk[nt] = malloc(nn * 8);
        make_masks(nx, ny, ntrace, has_dependents, MM(b, 0), mmask, NULL, NULL);
        construct_fields(nx, ny, ilo, ihi, jlo, jhi, ntrace, tracer_type, depend, has_dependents, hm + o, MM(b, 0), PL(mc, 0, b), PL(mx, 0, b),
                         PL(my, 0, b), mmask, NULL, NULL, NULL, NULL, NULL);
        for (int n = 1; n <= ncat; n++) {
            for (int nt = 0; nt < ntrace; nt++) {
                tmp[nt] = TM(b, n, nt);
                tcp[nt] = PL(tc, (size_t)(n - 1) * ntrace + nt, b); txp[nt] = PL(tx, (size_t)(n - 1) * ntrace + nt, b);
                typ[nt] = PL(ty, (size_t)(n - 1) * ntrace + nt, b);
            }
            make_masks(nx, ny, ntrace, has_dependents, MM(b, n), mmask, tmp, tmk);
            construct_fields(nx, ny, ilo, ihi, jlo, jhi, ntrace, tracer_type, depend, has_dependents, hm + o, MM(b, n), PL(mc, n, b),
                             PL(mx, n, b), PL(my, n, b), mmask, tmp, tcp, txp, typ, (const double *const *)tmk);
        }
        if (departure_points(nx, ny, ilo, ihi, jlo, jhi, dt, uvel + o, vvel + o, dxu + o, dyu + o, HTN + o, HTE + o, dpx + o, dpy + o, l_dp_midpt))
            rc = 1;
        for (int nt = 0; nt < ntrace; nt++) free(tmk[nt]);
        free(tmk); free(tmp); free(tcp); free(txp); free(typ); free(mmask);
    }
    if (!rc) {                                                                        /* :564-613 (nghost = 1) */
        orc_halo_r8(g, dpx, ORC_LOC_NECORNER, ORC_KIND_VECTOR, 0.0);
        orc_halo_r8(g, dpy, ORC_LOC_NECORNER, ORC_KIND_VECTOR, 0.0);
        for (int n = 0; n < ncp; n++) {
            orc_halo_r8(g, PL(mc, n, 0), ORC_LOC_CENTER, ORC_KIND_SCALAR, 0.0);
            orc_halo_r8(g, PL(mx, n, 0), ORC_LOC_CENTER, ORC_KIND_VECTOR, 0.0);
            orc_halo_r8(g, PL(my, n, 0), ORC_LOC_CENTER, ORC_KIND_VECTOR, 0.0);
        }
        for (size_t p = 0; p < ntp; p++) {
            orc_halo_r8(g, PL(tc, p, 0), ORC_LOC_CENTER, ORC_KIND_SCALAR, 0.0);
            orc_halo_r8(g, PL(tx, p, 0), ORC_LOC_CENTER, ORC_KIND_VECTOR, 0.0);
            orc_halo_r8(g, PL(ty, p, 0), ORC_LOC_CENTER, ORC_KIND_VECTOR, 0.0);
        }
    }
    for (int b = 0; b < nb && !rc; b++) {                                             /* :615-847 */
        const int ilo = g->ilo[b], ihi = g->ihi[b], jlo = g->jlo[b], jhi = g->jhi[b];
        const size_t o = (size_t)b * nn;
        double *mflxe = calloc(nn * ncp, 8), *mflxn = calloc(nn * ncp, 8);
        double *mtflxe = calloc(nn * (ntp ? ntp : 1), 8), *mtflxn = calloc(nn * (ntp ? ntp : 1), 8);
        const double **tcp = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *)), **txp = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *)),
                     **typ = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *));
        double flx[64];
        edge_tri t;
        for (int north = 0; north <= 1; north++) {
            const int ib = north ? ilo : ilo - 1, ie = ihi, jb = north ? jlo - 1 : jlo, je = jhi;          /* :1823-1826, :1849-1852 */
            double *mf = north ? mflxn : mflxe, *mtf = north ? mtflxn : mtflxe;
            for (int j = jb; j <= je; j++)
                for (int i = ib; i <= ie; i++) {
                    /* edges whose departure points are all zero carry no flux (:1911-1929) */
                    const size_t ka = IX(i, j), kb = north ? IX(i - 1, j) : IX(i, j - 1);
                    if (!(dpx[o + kb] != c0 || dpy[o + kb] != c0 || dpx[o + ka] != c0 || dpy[o + ka] != c0)) continue;
                    edge_triangles(nx, i, j, north, dpx + o, dpy + o, dxu + o, dyu + o, integral_order, &t);
                    edge_integrals(nx, ntrace, tracer_type, depend, integral_order, &t, PL(mc, 0, b), PL(mx, 0, b), PL(my, 0, b), &mf[ka], NULL, NULL,
                                   NULL, NULL);
                    for (int n = 1; n <= ncat; n++) {
                        for (int nt = 0; nt < ntrace; nt++) {
                            tcp[nt] = PL(tc, (size_t)(n - 1) * ntrace + nt, b); txp[nt] = PL(tx, (size_t)(n - 1) * ntrace + nt, b);
                            typ[nt] = PL(ty, (size_t)(n - 1) * ntrace + nt, b);
                        }
                        edge_integrals(nx, ntrace, tracer_type, depend, integral_order, &t, PL(mc, n, b), PL(mx, n, b), PL(my, n, b),
                                       &mf[(size_t)n * nn + ka], tcp, txp, typ, flx);
                        for (int nt = 0; nt < ntrace; nt++) mtf[((size_t)(n - 1) * ntrace + nt) * nn + ka] = flx[nt];
                    }
                }
        }
        if (update_fields(nx, ny, ilo, ihi, jlo, jhi, ntrace, tracer_type, depend, tarear + o, mflxe, mflxn, MM(b, 0), NULL, NULL, NULL)) rc = 2;
        double **fe = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *)), **fn = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *)),
               **tmn = calloc((size_t)(ntrace ? ntrace : 1), sizeof(double *));
        for (int n = 1; n <= ncat && !rc; n++) {
            for (int nt = 0; nt < ntrace; nt++) {
                fe[nt] = mtflxe + ((size_t)(n - 1) * ntrace + nt) * nn; fn[nt] = mtflxn + ((size_t)(n - 1) * ntrace + nt) * nn;
                tmn[nt] = TM(b, n, nt);
            }
            if (update_fields(nx, ny, ilo, ihi, jlo, jhi, ntrace, tracer_type, depend, tarear + o, mflxe + (size_t)n * nn, mflxn + (size_t)n * nn,
                              MM(b, n), fe, fn, ntrace ? tmn : NULL))
                rc = 2;
        }
        free(fe); free(fn); free(tmn); free(tcp); free(txp); free(typ);
        free(mflxe); free(mflxn); free(mtflxe); free(mtflxn);
    }
    free(dpx); free(dpy); free(mc); free(mx); free(my); free(tc); free(tx); free(ty);
    return rc;
}

/* ---------------------------------------------------------------------------
 * transport_remap (ice_transport_driver.F90:198-627) with its optional checks off (they are compile-time .false., :255-257):
 * state_to_tracers (:789-900), horizontal_remap, tracers_to_state (:908-1003), bound_state (ice_state.F90: the ghost-cell update of
 * aicen, trcrn, vicen, vsnon; aice0 is not part of it).
 * aice0 (nb, ny, nx); aicen, vicen, vsnon (nb, ncat, ny, nx); trcrn (nb, ncat, ntrcr_dim, ny, nx) of which the first ntrcr tracers are
 * in use (the reference passes trcrn(:,:,1:ntrcr,:,iblk)); nt_qsno 1-based, nslyr: the snow enthalpy tracers, shifted by rhos*Lfresh.
 * tracer_type / depend / has_dependents: (2 + ntrcr), as init_transport sets them.
 * ------------------------------------------------------------------------- */
int orc_transport_remap_state(const orc_geom *g, double dt, int ncat, int ntrcr, int ntrcr_dim, int nt_qsno, int nslyr, double rhos_lfresh,
                              const double *uvel, const double *vvel, double *aice0, double *aicen, double *vicen, double *vsnon, double *trcrn,
                              const int32_t *tracer_type, const int32_t *depend, const int32_t *has_dependents, int integral_order,
                              int l_dp_midpt, const double *HTE, const double *HTN, const double *dxu, const double *dyu, const double *tarear,
                              const double *hm) {
    const int nx = g->nx_block, ny = g->ny_block, nb = g->nblocks, ntrace = 2 + ntrcr, ncp = ncat + 1;
    const size_t nn = (size_t)nx * ny;
    double *aim = calloc((size_t)nb * ncp * nn, 8), *trm = calloc((size_t)nb * ncat * ntrace * nn, 8);
#define AIM(b, n) (aim + ((size_t)(b) * ncp + (n)) * nn)
#define TRM(b, n, nt) (trm + (((size_t)(b) * ncat + ((n)-1)) * ntrace + (nt)) * nn)
#define ST3(a, b, n) ((a) + ((size_t)(b) * ncat + ((n)-1)) * nn)
#define TRC(b, n, it) (trcrn + (((size_t)(b) * ncat + ((n)-1)) * ntrcr_dim + (it)) * nn)
    for (int b = 0; b < nb; b++) {                                                    /* state_to_tracers */
        memcpy(AIM(b, 0), aice0 + (size_t)b * nn, nn * 8);
        for (int n = 1; n <= ncat; n++) {
            const double *an = ST3(aicen, b, n), *vi = ST3(vicen, b, n), *vs = ST3(vsnon, b, n);
            double *am = AIM(b, n);
            for (size_t k = 0; k < nn; k++) {
                am[k] = an[k];
                if (!(am[k] > puny)) continue;
                const double w1 = c1 / am[k];
                TRM(b, n, 0)[k] = vi[k] * w1;
                TRM(b, n, 1)[k] = vs[k] * w1;
                for (int it = 1; it <= ntrcr; it++) {
                    const double v = TRC(b, n, it - 1)[k];
                    TRM(b, n, 2 + it - 1)[k] = (it >= nt_qsno && it < nt_qsno + nslyr) ? v + rhos_lfresh : v;
                }
            }
        }
    }
    const int rc = orc_horizontal_remap(g, dt, ncat, ntrace, uvel, vvel, aim, trm, 0, tracer_type, depend, has_dependents, integral_order,
                                        l_dp_midpt, HTE, HTN, dxu, dyu, tarear, hm);
    if (!rc) {
        for (int b = 0; b < nb; b++) {                                                /* tracers_to_state: every cell of the block with aim > 0 */
            memcpy(aice0 + (size_t)b * nn, AIM(b, 0), nn * 8);
            for (int n = 1; n <= ncat; n++) {
                double *an = ST3(aicen, b, n), *vi = ST3(vicen, b, n), *vs = ST3(vsnon, b, n);
                const double *am = AIM(b, n);
                for (size_t k = 0; k < nn; k++) {
                    if (!(am[k] > c0)) continue;
                    an[k] = am[k];
                    vi[k] = am[k] * TRM(b, n, 0)[k];
                    vs[k] = am[k] * TRM(b, n, 1)[k];
                    for (int it = 1; it <= ntrcr; it++) {
                        const double v = TRM(b, n, 2 + it - 1)[k];
                        TRC(b, n, it - 1)[k] = (it >= nt_qsno && it < nt_qsno + nslyr) ? v - rhos_lfresh : v;
                    }
                }
            }
        }
        /* bound_state: planes of one category / tracer are strided in these arrays, so each is copied out, updated, copied back */
        double *w = malloc((size_t)nb * nn * 8);
        for (int n = 1; n <= ncat; n++)
            for (int q = 0; q < 3 + ntrcr; q++) {
                for (int b = 0; b < nb; b++) {
                    const double *src = q == 0 ? ST3(aicen, b, n) : q == 1 ? ST3(vicen, b, n) : q == 2 ? ST3(vsnon, b, n) : TRC(b, n, q - 3);
                    memcpy(w + (size_t)b * nn, src, nn * 8);
                }
                orc_halo_r8(g, w, ORC_LOC_CENTER, ORC_KIND_SCALAR, 0.0);
                for (int b = 0; b < nb; b++) {
                    double *dst = q == 0 ? ST3(aicen, b, n) : q == 1 ? ST3(vicen, b, n) : q == 2 ? ST3(vsnon, b, n) : TRC(b, n, q - 3);
                    memcpy(dst, w + (size_t)b * nn, nn * 8);
                }
            }
        free(w);
    }
    free(aim); free(trm);
    return rc;
}
