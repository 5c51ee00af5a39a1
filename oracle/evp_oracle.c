/*
 * oracle/evp_oracle.c -- CPU restatement of the CICE5 EVP dynamics path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY.  Parity: halo updates and ice_strength pinned by the reference's own output, the rest UNPINNED (see evp_oracle.h).
 *
 * Every routine follows the operation order of the Fortran it cites, with
 * -ffp-contract=off so that no multiply-add is fused.  Fortran evaluates
 * a + b + c as (a + b) + c and a*b*c as (a*b)*c; the C below is written the same way.
 */
#include "evp_oracle.h"
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* Fortran (i,j) 1-based -> flat index of one (nx,ny) block */
#define IX(i, j) ((size_t)((j)-1) * (size_t)nx + (size_t)((i)-1))

/* ice_constants.F90:136-148 (drivers/auscom/ice_constants.F90:180-188) */
static const double c0 = 0.0, c1 = 1.0, c2 = 2.0, c4 = 4.0;
static const double p25 = 0.25, p5 = 0.5;
#define P166 (1.0 / 6.0)
#define P333 (1.0 / 3.0)
#define P111 (1.0 / 9.0)
#define P055 ((1.0 / 9.0) * 0.5)
#define P027 (((1.0 / 9.0) * 0.5) * 0.5)
#define P222 (2.0 / 9.0)
static const double puny = 1.0e-11;
static const double spval_dbl = 1.0e30;

static double wall(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---------------------------------------------------------------------------
 * set_evp_parameters  (source/ice_dyn_shared.F90:185-259)
 * xmin = min(global_minval(dxt,tmask), global_minval(dyt,tmask)) is supplied.
 * ------------------------------------------------------------------------- */
void orc_set_evp_parameters(double dt, int32_t ndte, int32_t revised_evp, double xmin, orc_params *p) {
    const double eyc = 0.36;                        /* :51 */
    double dte = dt / (double)ndte;                 /* :209 */
    p->dt = dt;
    p->ndte = ndte;
    p->revised_evp = revised_evp;
    p->dtei = c1 / dte;                             /* :210 */
    p->ecci = p25;                                  /* :214 */
    double tdamp2 = c2 * eyc * dt;                  /* :217 */
    p->dte2T = dte / tdamp2;                        /* :218 */
    double Se = 0.86, xi = 5.5e-3;                  /* :226-227 */
    double gamma = p25 * 1.e11 * dt;                /* :228 */
    if (revised_evp) {                              /* :230-233 */
        p->revp = c1;
        p->arlx1i = c2 * xi / Se;
        p->brlx = c2 * Se * xi * gamma / (xmin * xmin);
    } else {                                        /* :239-242 */
        p->revp = c0;
        p->arlx1i = p->dte2T;
        p->brlx = dt * p->dtei;
    }
    p->denom1 = c1 / (c1 + p->arlx1i);              /* :257 */
}

/* ---------------------------------------------------------------------------
 * evp_prep1  (source/ice_dyn_shared.F90:270-365)
 * ------------------------------------------------------------------------- */
void orc_evp_prep1(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                   const double *aice, const double *vice, const double *vsno, const int32_t *tmask,
                   const double *strairxT, const double *strairyT,
                   double *strairx, double *strairy, double *tmass, int32_t *icetmask,
                   const orc_params *p) {
    unsigned char *tmphm = (unsigned char *)malloc((size_t)nx * ny);
    for (int j = 1; j <= ny; j++)
        for (int i = 1; i <= nx; i++) {
            size_t k = IX(i, j);
            if (tmask[k])                                               /* :322-326 */
                tmass[k] = (p->rhoi * vice[k] + p->rhos * vsno[k]);
            else
                tmass[k] = c0;
            tmphm[k] = tmask[k] && (aice[k] > p->a_min) && (tmass[k] > p->m_min); /* :331-332 */
            strairx[k] = strairxT[k];                                   /* :339-340 */
            strairy[k] = strairyT[k];
            icetmask[k] = 0;                                            /* :345 */
        }
    for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++) {                              /* :350-363 */
            if (tmphm[IX(i - 1, j + 1)] || tmphm[IX(i, j + 1)] || tmphm[IX(i + 1, j + 1)] ||
                tmphm[IX(i - 1, j)]     || tmphm[IX(i, j)]     || tmphm[IX(i + 1, j)] ||
                tmphm[IX(i - 1, j - 1)] || tmphm[IX(i, j - 1)] || tmphm[IX(i + 1, j - 1)])
                icetmask[IX(i, j)] = 1;
            if (!tmask[IX(i, j)]) icetmask[IX(i, j)] = 0;
        }
    free(tmphm);
}

/* ---------------------------------------------------------------------------
 * to_ugrid / to_tgrid, one block  (source/ice_grid.F90:1834-1878, 1924-1958)
 * ------------------------------------------------------------------------- */
void orc_to_ugrid_blk(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                      const double *work1, const double *tarea, const double *uarea, double *work2) {
    for (size_t k = 0; k < (size_t)nx * ny; k++) work2[k] = c0;          /* :1853 */
    for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++)                                 /* :1865-1870 */
            work2[IX(i, j)] = p25 *
                (((work1[IX(i, j)] * tarea[IX(i, j)]
                 + work1[IX(i + 1, j)] * tarea[IX(i + 1, j)])
                 + work1[IX(i, j + 1)] * tarea[IX(i, j + 1)])
                 + work1[IX(i + 1, j + 1)] * tarea[IX(i + 1, j + 1)])
                / uarea[IX(i, j)];
}

void orc_to_tgrid_blk(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                      const double *work1, const double *tarea, const double *uarea, double *work2) {
    (void)ny;
    for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++)                                 /* :1946-1952 */
            work2[IX(i, j)] = p25 *
                (((work1[IX(i, j)] * uarea[IX(i, j)]
                 + work1[IX(i - 1, j)] * uarea[IX(i - 1, j)])
                 + work1[IX(i, j - 1)] * uarea[IX(i, j - 1)])
                 + work1[IX(i - 1, j - 1)] * uarea[IX(i - 1, j - 1)])
                / tarea[IX(i, j)];
}

/* ---------------------------------------------------------------------------
 * evp_prep2  (source/ice_dyn_shared.F90:377-614)
 * ------------------------------------------------------------------------- */
void orc_evp_prep2(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                   int32_t *icellt, int32_t *icellu,
                   int32_t *indxti, int32_t *indxtj, int32_t *indxui, int32_t *indxuj,
                   const double *aiu, const double *umass, double *umassdti, const double *fcor,
                   const int32_t *umask, const double *uocn, const double *vocn,
                   const double *strairx, const double *strairy,
                   const double *ss_tltx, const double *ss_tlty,
                   const int32_t *icetmask, int32_t *iceumask, double *fm, double dt,
                   double *strtltx, double *strtlty, double *strocnx, double *strocny,
                   double *strintx, double *strinty, double *waterx, double *watery,
                   double *forcex, double *forcey,
                   double *const stressp[4], double *const stressm[4], double *const stress12[4],
                   double *uvel_init, double *vvel_init, double *uvel, double *vvel,
                   const orc_params *p) {
    for (int j = 1; j <= ny; j++)
        for (int i = 1; i <= nx; i++) {                                  /* :484-520 */
            size_t k = IX(i, j);
            waterx[k] = c0; watery[k] = c0; forcex[k] = c0; forcey[k] = c0; umassdti[k] = c0;
            if (p->revp == 1 || icetmask[k] == 0)
                for (int c = 0; c < 4; c++) { stressp[c][k] = c0; stressm[c][k] = c0; stress12[c][k] = c0; }
        }

    int nt = 0;                                                          /* :528-537 */
    for (int j = jlo; j <= jhi + 1; j++)
        for (int i = ilo; i <= ihi + 1; i++)
            if (icetmask[IX(i, j)] == 1) { indxti[nt] = i; indxtj[nt] = j; nt++; }
    *icellt = nt;

    int nu = 0;                                                          /* :545-577 */
    for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++) {
            size_t k = IX(i, j);
            int old = iceumask[k];
            iceumask[k] = umask[k] && (aiu[k] > p->a_min) && (umass[k] > p->m_min);
            if (iceumask[k]) {
                indxui[nu] = i; indxuj[nu] = j; nu++;
                if (!old) { uvel[k] = uocn[k]; vvel[k] = vocn[k]; }
            } else {
                uvel[k] = c0; vvel[k] = c0;
                strintx[k] = c0; strinty[k] = c0; strocnx[k] = c0; strocny[k] = c0;
            }
            uvel_init[k] = uvel[k];
            vvel_init[k] = vvel[k];
        }
    *icellu = nu;

    for (int ij = 0; ij < nu; ij++) {                                    /* :583-612 */
        size_t k = IX(indxui[ij], indxuj[ij]);
        umassdti[k] = umass[k] / dt;
        fm[k] = fcor[k] * umass[k];
        double sg = copysign(c1, fm[k]);
        waterx[k] = uocn[k] * p->cosw - vocn[k] * p->sinw * sg;
        watery[k] = vocn[k] * p->cosw + uocn[k] * p->sinw * sg;
        if (p->tilt_from_slope) {
            strtltx[k] = -p->gravit * umass[k] * ss_tltx[k];
            strtlty[k] = -p->gravit * umass[k] * ss_tlty[k];
        } else {
            strtltx[k] = -fm[k] * vocn[k];
            strtlty[k] = fm[k] * uocn[k];
        }
        forcex[k] = strairx[k] + strtltx[k];
        forcey[k] = strairy[k] + strtlty[k];
    }
}

/* ---------------------------------------------------------------------------
 * stress  (source/ice_dyn_evp.F90:520-849)
 * str is (nx,ny,8): str[(c-1)*nx*ny + IX(i,j)]
 * ------------------------------------------------------------------------- */
void orc_stress(int nx, int ny, int ksub, int ndte, int icellt,
                const int32_t *indxti, const int32_t *indxtj,
                const double *uvel, const double *vvel,
                const double *dxt, const double *dyt, const double *dxhy, const double *dyhx,
                const double *cxp, const double *cyp, const double *cxm, const double *cym,
                const double *tarear, const double *tinyarea, const double *strength,
                double *const stressp[4], double *const stressm[4], double *const stress12[4],
                double *shear, double *divu, double *prs_sig, double *rdg_conv, double *rdg_shear,
                double *str, const orc_params *p) {
    const size_t nn = (size_t)nx * ny;
    const double ecci = p->ecci, arlx1i = p->arlx1i, denom1 = p->denom1;
    double *stressp_1 = stressp[0], *stressp_2 = stressp[1], *stressp_3 = stressp[2], *stressp_4 = stressp[3];
    double *stressm_1 = stressm[0], *stressm_2 = stressm[1], *stressm_3 = stressm[2], *stressm_4 = stressm[3];
    double *stress12_1 = stress12[0], *stress12_2 = stress12[1], *stress12_3 = stress12[2], *stress12_4 = stress12[3];

    memset(str, 0, 8 * nn * sizeof(double));                              /* :613 */

    for (int ij = 0; ij < icellt; ij++) {
        const int i = indxti[ij], j = indxtj[ij];
        const size_t k = IX(i, j);
        const double u_ij = uvel[IX(i, j)], u_mj = uvel[IX(i - 1, j)];
        const double u_im = uvel[IX(i, j - 1)], u_mm = uvel[IX(i - 1, j - 1)];
        const double v_ij = vvel[IX(i, j)], v_mj = vvel[IX(i - 1, j)];
        const double v_im = vvel[IX(i, j - 1)], v_mm = vvel[IX(i - 1, j - 1)];
        const double Cyp = cyp[k], Cxp = cxp[k], Cym = cym[k], Cxm = cxm[k], Dxt = dxt[k], Dyt = dyt[k];

        /* :627-634 divergence */
        double divune = Cyp * u_ij - Dyt * u_mj + Cxp * v_ij - Dxt * v_im;
        double divunw = Cym * u_mj + Dyt * u_ij + Cxp * v_mj - Dxt * v_mm;
        double divusw = Cym * u_mm + Dyt * u_im + Cxm * v_mm + Dxt * v_mj;
        double divuse = Cyp * u_im - Dyt * u_mm + Cxm * v_im + Dxt * v_ij;
        /* :637-644 tension */
        double tensionne = -Cym * u_ij - Dyt * u_mj + Cxm * v_ij + Dxt * v_im;
        double tensionnw = -Cyp * u_mj + Dyt * u_ij + Cxm * v_mj + Dxt * v_mm;
        double tensionsw = -Cyp * u_mm + Dyt * u_im + Cxp * v_mm - Dxt * v_mj;
        double tensionse = -Cym * u_im - Dyt * u_mm + Cxp * v_im - Dxt * v_ij;
        /* :647-654 shearing */
        double shearne = -Cym * v_ij - Dyt * v_mj - Cxm * u_ij - Dxt * u_im;
        double shearnw = -Cyp * v_mj + Dyt * v_ij - Cxm * u_mj - Dxt * u_mm;
        double shearsw = -Cyp * v_mm + Dyt * v_im - Cxp * u_mm + Dxt * u_mj;
        double shearse = -Cym * v_im - Dyt * v_mm - Cxp * u_im + Dxt * u_ij;
        /* :657-660 Delta */
        double Deltane = sqrt(divune * divune + ecci * (tensionne * tensionne + shearne * shearne));
        double Deltanw = sqrt(divunw * divunw + ecci * (tensionnw * tensionnw + shearnw * shearnw));
        double Deltase = sqrt(divuse * divuse + ecci * (tensionse * tensionse + shearse * shearse));
        double Deltasw = sqrt(divusw * divusw + ecci * (tensionsw * tensionsw + shearsw * shearsw));

        if (ksub == ndte) {                                               /* :665-677 */
            divu[k] = p25 * (divune + divunw + divuse + divusw) * tarear[k];
            double tmp = p25 * (Deltane + Deltanw + Deltase + Deltasw) * tarear[k];
            rdg_conv[k] = -fmin(divu[k], c0);
            rdg_shear[k] = p5 * (tmp - fabs(divu[k]));
            double ts = tensionne + tensionnw + tensionse + tensionsw;
            double ss = shearne + shearnw + shearse + shearsw;
            shear[k] = p25 * tarear[k] * sqrt(ts * ts + ss * ss);
        }

        /* :683-697 replacement pressure */
        double c0ne = strength[k] / fmax(Deltane, tinyarea[k]);
        double c0nw = strength[k] / fmax(Deltanw, tinyarea[k]);
        double c0sw = strength[k] / fmax(Deltasw, tinyarea[k]);
        double c0se = strength[k] / fmax(Deltase, tinyarea[k]);
        prs_sig[k] = c0ne * Deltane;
        double c1ne = c0ne * arlx1i, c1nw = c0nw * arlx1i, c1sw = c0sw * arlx1i, c1se = c0se * arlx1i;
        c0ne = c1ne * ecci; c0nw = c1nw * ecci; c0sw = c1sw * ecci; c0se = c1se * ecci;

        /* :704-721 the stresses */
        stressp_1[k] = (stressp_1[k] + c1ne * (divune - Deltane)) * denom1;
        stressp_2[k] = (stressp_2[k] + c1nw * (divunw - Deltanw)) * denom1;
        stressp_3[k] = (stressp_3[k] + c1sw * (divusw - Deltasw)) * denom1;
        stressp_4[k] = (stressp_4[k] + c1se * (divuse - Deltase)) * denom1;
        stressm_1[k] = (stressm_1[k] + c0ne * tensionne) * denom1;
        stressm_2[k] = (stressm_2[k] + c0nw * tensionnw) * denom1;
        stressm_3[k] = (stressm_3[k] + c0sw * tensionsw) * denom1;
        stressm_4[k] = (stressm_4[k] + c0se * tensionse) * denom1;
        stress12_1[k] = (stress12_1[k] + c0ne * shearne * p5) * denom1;
        stress12_2[k] = (stress12_2[k] + c0nw * shearnw * p5) * denom1;
        stress12_3[k] = (stress12_3[k] + c0sw * shearsw * p5) * denom1;
        stress12_4[k] = (stress12_4[k] + c0se * shearse * p5) * denom1;

        /* :752-771 combinations */
        double ssigpn = stressp_1[k] + stressp_2[k];
        double ssigps = stressp_3[k] + stressp_4[k];
        double ssigpe = stressp_1[k] + stressp_4[k];
        double ssigpw = stressp_2[k] + stressp_3[k];
        double ssigp1 = (stressp_1[k] + stressp_3[k]) * P055;
        double ssigp2 = (stressp_2[k] + stressp_4[k]) * P055;
        double ssigmn = stressm_1[k] + stressm_2[k];
        double ssigms = stressm_3[k] + stressm_4[k];
        double ssigme = stressm_1[k] + stressm_4[k];
        double ssigmw = stressm_2[k] + stressm_3[k];
        double ssigm1 = (stressm_1[k] + stressm_3[k]) * P055;
        double ssigm2 = (stressm_2[k] + stressm_4[k]) * P055;
        double ssig12n = stress12_1[k] + stress12_2[k];
        double ssig12s = stress12_3[k] + stress12_4[k];
        double ssig12e = stress12_1[k] + stress12_4[k];
        double ssig12w = stress12_2[k] + stress12_3[k];
        double ssig121 = (stress12_1[k] + stress12_3[k]) * P111;
        double ssig122 = (stress12_2[k] + stress12_4[k]) * P111;
        /* :773-790 */
        double csigpne = P111 * stressp_1[k] + ssigp2 + P027 * stressp_3[k];
        double csigpnw = P111 * stressp_2[k] + ssigp1 + P027 * stressp_4[k];
        double csigpsw = P111 * stressp_3[k] + ssigp2 + P027 * stressp_1[k];
        double csigpse = P111 * stressp_4[k] + ssigp1 + P027 * stressp_2[k];
        double csigmne = P111 * stressm_1[k] + ssigm2 + P027 * stressm_3[k];
        double csigmnw = P111 * stressm_2[k] + ssigm1 + P027 * stressm_4[k];
        double csigmsw = P111 * stressm_3[k] + ssigm2 + P027 * stressm_1[k];
        double csigmse = P111 * stressm_4[k] + ssigm1 + P027 * stressm_2[k];
        double csig12ne = P222 * stress12_1[k] + ssig122 + P055 * stress12_3[k];
        double csig12nw = P222 * stress12_2[k] + ssig121 + P055 * stress12_4[k];
        double csig12sw = P222 * stress12_3[k] + ssig122 + P055 * stress12_1[k];
        double csig12se = P222 * stress12_4[k] + ssig121 + P055 * stress12_2[k];
        /* :792-795 */
        double str12ew = p5 * Dxt * (P333 * ssig12e + P166 * ssig12w);
        double str12we = p5 * Dxt * (P333 * ssig12w + P166 * ssig12e);
        double str12ns = p5 * Dyt * (P333 * ssig12n + P166 * ssig12s);
        double str12sn = p5 * Dyt * (P333 * ssig12s + P166 * ssig12n);

        const double Dxhy = dxhy[k], Dyhx = dyhx[k];
        /* :800-820 dF/dx */
        double strp_tmp = p25 * Dyt * (P333 * ssigpn + P166 * ssigps);
        double strm_tmp = p25 * Dyt * (P333 * ssigmn + P166 * ssigms);
        str[0 * nn + k] = -strp_tmp - strm_tmp - str12ew + Dxhy * (-csigpne + csigmne) + Dyhx * csig12ne;
        str[1 * nn + k] = strp_tmp + strm_tmp - str12we + Dxhy * (-csigpnw + csigmnw) + Dyhx * csig12nw;
        strp_tmp = p25 * Dyt * (P333 * ssigps + P166 * ssigpn);
        strm_tmp = p25 * Dyt * (P333 * ssigms + P166 * ssigmn);
        str[2 * nn + k] = -strp_tmp - strm_tmp + str12ew + Dxhy * (-csigpse + csigmse) + Dyhx * csig12se;
        str[3 * nn + k] = strp_tmp + strm_tmp + str12we + Dxhy * (-csigpsw + csigmsw) + Dyhx * csig12sw;
        /* :825-845 dF/dy */
        strp_tmp = p25 * Dxt * (P333 * ssigpe + P166 * ssigpw);
        strm_tmp = p25 * Dxt * (P333 * ssigme + P166 * ssigmw);
        str[4 * nn + k] = -strp_tmp + strm_tmp - str12ns - Dyhx * (csigpne + csigmne) + Dxhy * csig12ne;
        str[5 * nn + k] = strp_tmp - strm_tmp - str12sn - Dyhx * (csigpse + csigmse) + Dxhy * csig12se;
        strp_tmp = p25 * Dxt * (P333 * ssigpw + P166 * ssigpe);
        strm_tmp = p25 * Dxt * (P333 * ssigmw + P166 * ssigme);
        str[6 * nn + k] = -strp_tmp + strm_tmp + str12ns - Dyhx * (csigpnw + csigmnw) + Dxhy * csig12nw;
        str[7 * nn + k] = strp_tmp - strm_tmp + str12sn - Dyhx * (csigpsw + csigmsw) + Dxhy * csig12sw;
    }
}

/* ---------------------------------------------------------------------------
 * stepu  (source/ice_dyn_shared.F90:623-748)
 * ------------------------------------------------------------------------- */
void orc_stepu(int nx, int ny, int icellu, const double *Cw,
               const int32_t *indxui, const int32_t *indxuj,
               const double *aiu, const double *str,
               const double *uocn, const double *vocn, const double *waterx, const double *watery,
               const double *forcex, const double *forcey, const double *umassdti, const double *fm,
               const double *uarear, double *strocnx, double *strocny, double *strintx, double *strinty,
               const double *uvel_init, const double *vvel_init, double *uvel, double *vvel,
               const orc_params *p) {
    const size_t nn = (size_t)nx * ny;
    const double brlx = p->brlx, revp = p->revp, cosw = p->cosw, sinw = p->sinw, rhow = p->rhow;
    for (int ij = 0; ij < icellu; ij++) {
        const int i = indxui[ij], j = indxuj[ij];
        const size_t k = IX(i, j);
        double uold = uvel[k], vold = vvel[k];                            /* :704-705 */
        double du = uocn[k] - uold, dv = vocn[k] - vold;
        double vrel = aiu[k] * rhow * Cw[k] * sqrt(du * du + dv * dv);    /* :708-709 */
        double taux = vrel * waterx[k];                                   /* :711-712 */
        double tauy = vrel * watery[k];
        double cca = (brlx + revp) * umassdti[k] + vrel * cosw;           /* :715 */
        double ccb = fm[k] + copysign(c1, fm[k]) * vrel * sinw;           /* :720 */
        double ab2 = cca * cca + ccb * ccb;                               /* :722 */
        strintx[k] = uarear[k] *                                          /* :725-728 */
            (((str[0 * nn + IX(i, j)] + str[1 * nn + IX(i + 1, j)]) + str[2 * nn + IX(i, j + 1)]) + str[3 * nn + IX(i + 1, j + 1)]);
        strinty[k] = uarear[k] *
            (((str[4 * nn + IX(i, j)] + str[5 * nn + IX(i, j + 1)]) + str[6 * nn + IX(i + 1, j)]) + str[7 * nn + IX(i + 1, j + 1)]);
        double cc1 = strintx[k] + forcex[k] + taux + umassdti[k] * (brlx * uold + revp * uvel_init[k]); /* :731-734 */
        double cc2 = strinty[k] + forcey[k] + tauy + umassdti[k] * (brlx * vold + revp * vvel_init[k]);
        uvel[k] = (cca * cc1 + ccb * cc2) / ab2;                          /* :736-737 */
        vvel[k] = (cca * cc2 - ccb * cc1) / ab2;
        strocnx[k] = taux;                                                /* :743-744 */
        strocny[k] = tauy;
    }
}

/* ---------------------------------------------------------------------------
 * evp_finish  (source/ice_dyn_shared.F90:757-844)
 * ------------------------------------------------------------------------- */
void orc_evp_finish(int nx, int ny, int icellu, const double *Cw,
                    const int32_t *indxui, const int32_t *indxuj,
                    const double *uvel, const double *vvel, const double *uocn, const double *vocn,
                    const double *aiu, const double *fm,
                    double *strocnx, double *strocny, double *strocnxT, double *strocnyT,
                    const orc_params *p) {
    for (size_t k = 0; k < (size_t)nx * ny; k++) { strocnxT[k] = c0; strocnyT[k] = c0; }  /* :806-811 */
    for (int ij = 0; ij < icellu; ij++) {
        const size_t k = IX(indxui[ij], indxuj[ij]);
        double du = uocn[k] - uvel[k], dv = vocn[k] - vvel[k];
        double vrel = p->rhow * Cw[k] * sqrt(du * du + dv * dv);          /* :818-819 */
        vrel = vrel * aiu[k];                                             /* :827 */
        double sg = copysign(c1, fm[k]);
        strocnx[k] = vrel * (du * p->cosw - dv * p->sinw * sg);           /* :828-831 */
        strocny[k] = vrel * (dv * p->cosw + du * p->sinw * sg);
        strocnxT[k] = strocnx[k] / aiu[k];                                /* :840-841 */
        strocnyT[k] = strocny[k] / aiu[k];
    }
}

/* principal_stress  (source/ice_dyn_shared.F90:853-893) */
void orc_principal_stress(int nx, int ny, const double *stressp_1, const double *stressm_1,
                          const double *stress12_1, const double *prs_sig, double *sig1, double *sig2) {
    for (size_t k = 0; k < (size_t)nx * ny; k++) {
        if (prs_sig[k] > puny) {
            double r = sqrt(stressm_1[k] * stressm_1[k] + c4 * (stress12_1[k] * stress12_1[k]));
            sig1[k] = (p5 * (stressp_1[k] + r)) / prs_sig[k];
            sig2[k] = (p5 * (stressp_1[k] - r)) / prs_sig[k];
        } else {
            sig1[k] = spval_dbl;
            sig2[k] = spval_dbl;
        }
    }
}

/* ---------------------------------------------------------------------------
 * exp(x): k = nint(x/ln2), r = x - k ln2 in two pieces, exp(r) = 1 + 2r/(R(r^2) - r) with R = 2 - c/r ... written as in
 * fdlibm's e_exp.c (Sun Microsystems, public domain), whose error bound is < 1 ulp.
 * ------------------------------------------------------------------------- */
/* OpenMP team size of the loops below (the runtime may have been initialised by the host process with another default) */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

double orc_exp(double x) {
    static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00,
                        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                        P5 = 4.13813679705723846039e-08;
    const double ax = fabs(x);
    double hi = 0.0, lo = 0.0;
    int k = 0;
    if (ax > 0.34657359027997264) {                 /* |x| > 0.5 ln2 */
        if (ax < 1.0397207708399179) {              /* |x| < 1.5 ln2 */
            k = x < 0.0 ? -1 : 1;
            hi = x - (double)k * ln2HI;
            lo = (double)k * ln2LO;
        } else {
            k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
            const double t = (double)k;
            hi = x - t * ln2HI;                     /* t*ln2HI is exact */
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) {        /* 2^-28 */
        return c1 + x;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return c1 - ((x * c) / (c - c2) - x);
    const double y = c1 - ((lo - (x * c) / (c2 - c)) - hi);
    return ldexp(y, k);
}

/* ---------------------------------------------------------------------------
 * ice_strength, source/ice_mechred.F90:2111-2269.  Rothrock (1975): strength from the potential-energy change of
 * ridging, with the participation function and the ridge thickness distribution of ridge_itd (:936-1285); constants
 * :66-82.  asum_ridging (:758-812) is called by the reference but its result is not used by the strength.
 * ------------------------------------------------------------------------- */
#define ORC_MAXCAT 16
void orc_ice_strength(int nx, int ny, int ilo, int ihi, int jlo, int jhi, int icells,
                      const int32_t *indxi, const int32_t *indxj,
                      const double *aice, const double *vice, const double *aice0,
                      const double *aicen, const double *vicen, double *strength, const orc_params *p) {
    const size_t nn = (size_t)nx * ny;
    const double puny = 1.0e-11, p333 = c1 / 3.0, p15 = 0.15, p05 = 0.05, c25 = 25.0, c20 = 20.0;
    const double Gstar = p15, astar = p05, maxraft = c1, Hstar = c25, Pstar = 2.75e4, Cstar = c20;      /* :72-82 */
    const double Cp = p5 * p->gravit * (p->rhow - p->rhoi) * p->rhoi / p->rhow;                         /* :68 */
    const double Gstari = c1 / Gstar, astari = c1 / astar;                                              /* :1003-1005 */
    const int ncat = p->ncat;
    for (size_t k = 0; k < nn; k++) strength[k] = c0;                                                   /* :2183 */
    if (p->kstrength != 1) {                                                                            /* :2258-2265 */
        for (int j = jlo; j <= jhi; j++)
            for (int i = ilo; i <= ihi; i++)
                strength[IX(i, j)] = Pstar * vice[IX(i, j)] * orc_exp(-Cstar * (c1 - aice[IX(i, j)]));
        return;
    }
    for (int ij = 0; ij < icells; ij++) {
        const size_t k = IX(indxi[ij], indxj[ij]);
        double Gsum[ORC_MAXCAT + 2];        /* Gsum[n+1] = Gsum(ij,n), n = -1..ncat */
        double apartic[ORC_MAXCAT + 1], hrmin[ORC_MAXCAT + 1], hrmax[ORC_MAXCAT + 1], hrexp[ORC_MAXCAT + 1], krdg[ORC_MAXCAT + 1];
        /* ---- ridge_itd ---- */
        Gsum[0] = c0;                                                                                   /* :1021-1025 */
        apartic[0] = c0;
        for (int n = 1; n <= ncat; n++) { apartic[n] = c0; hrmin[n] = c0; hrmax[n] = c0; hrexp[n] = c0; krdg[n] = c1; }
        Gsum[1] = (aice0[k] > puny) ? aice0[k] : Gsum[0];                                               /* :1050-1058 */
        for (int n = 1; n <= ncat; n++) {                                                               /* :1060-1071 */
            const double a = aicen[(size_t)(n - 1) * nn + k];
            Gsum[n + 1] = (a > puny) ? Gsum[n] + a : Gsum[n];
        }
        const double work = c1 / Gsum[ncat + 1];                                                        /* :1076-1083 */
        for (int n = 0; n <= ncat; n++) Gsum[n + 1] = Gsum[n + 1] * work;
        if (p->krdg_partic == 0) {                                                                      /* :1104-1117 */
            for (int n = 0; n <= ncat; n++) {
                const double g1 = Gsum[n + 1], g0 = Gsum[n];
                if (g1 < Gstar) apartic[n] = Gstari * (g1 - g0) * (c2 - (g0 + g1) * Gstari);
                else if (g0 < Gstar) apartic[n] = Gstari * (Gstar - g0) * (c2 - (g0 + Gstar) * Gstari);
            }
        } else {                                                                                        /* :1119-1141 */
            const double xtmp = c1 / (c1 - orc_exp(-astari));
            for (int n = -1; n <= ncat; n++) Gsum[n + 1] = orc_exp(-Gsum[n + 1] * astari) * xtmp;
            for (int n = 0; n <= ncat; n++) apartic[n] = Gsum[n] - Gsum[n + 1];
        }
        if (p->krdg_redist == 0) {                                                                      /* :1169-1190 */
            for (int n = 1; n <= ncat; n++) {
                const double a = aicen[(size_t)(n - 1) * nn + k];
                if (a > puny) {
                    const double hi = vicen[(size_t)(n - 1) * nn + k] / a;
                    hrmin[n] = fmin(c2 * hi, hi + maxraft);
                    hrmax[n] = c2 * sqrt(Hstar * hi);
                    hrmax[n] = fmax(hrmax[n], hrmin[n] + puny);
                    const double hrmean = p5 * (hrmin[n] + hrmax[n]);
                    krdg[n] = hrmean / hi;
                }
            }
        } else {                                                                                        /* :1219-1240 */
            for (int n = 1; n <= ncat; n++) {
                const double a = aicen[(size_t)(n - 1) * nn + k];
                if (a > puny) {
                    double hi = vicen[(size_t)(n - 1) * nn + k] / a;
                    hi = fmax(hi, puny);
                    hrmin[n] = fmin(c2 * hi, hi + maxraft);
                    hrexp[n] = p->mu_rdg * sqrt(hi);
                    krdg[n] = (hrmin[n] + hrexp[n]) / hi;
                }
            }
        }
        double aksum = apartic[0];                                                                      /* :1248-1258 */
        for (int n = 1; n <= ncat; n++) aksum = aksum + apartic[n] * (c1 - c1 / krdg[n]);
        /* ---- ice_strength, Rothrock ---- */
        double s = c0;
        for (int n = 1; n <= ncat; n++) {                                                               /* :2207-2243 */
            const double a = aicen[(size_t)(n - 1) * nn + k];
            if (a > puny && apartic[n] > c0) {
                const double hi = vicen[(size_t)(n - 1) * nn + k] / a;
                double h2rdg;
                if (p->krdg_redist == 0)
                    h2rdg = p333 * (hrmax[n] * hrmax[n] * hrmax[n] - hrmin[n] * hrmin[n] * hrmin[n]) / (hrmax[n] - hrmin[n]);
                else
                    h2rdg = hrmin[n] * hrmin[n] + c2 * hrmin[n] * hrexp[n] + c2 * hrexp[n] * hrexp[n];
                const double dh2rdg = -hi * hi + h2rdg / krdg[n];
                s = s + apartic[n] * dh2rdg;
            }
        }
        strength[k] = p->Cf * Cp * s / aksum;                                                           /* :2250 */
    }
}

/* Hibler (1979) strength, kstrength /= 1  (source/ice_mechred.F90:2258-2265; Pstar,Cstar :80-82) */
void orc_strength_hibler(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                         const double *aice, const double *vice, double *strength) {
    const double Pstar = 2.75e4, Cstar = 20.0;
    (void)ny;
    for (int j = jlo; j <= jhi; j++)
        for (int i = ilo; i <= ihi; i++)
            strength[IX(i, j)] = Pstar * vice[IX(i, j)] * exp(-Cstar * (c1 - aice[IX(i, j)]));
}

/* ---------------------------------------------------------------------------
 * Halo updates.  Semantics of mpi/ice_boundary.F90 (ghost cells are first
 * overwritten with `fill`, :1409-1416, :2625-2632), expressed through a global
 * buffer: physical cells of all local blocks -> global array; every ghost cell
 * then reads its global source.  E-W: cyclic wraps, open/closed have no source
 * (ice_blocks.F90:455-480).  N-S: open/closed no source; tripole (u-fold) follows
 * the copy-in/copy-out lists of serial/ice_boundary.F90:3717-3776 and the
 * per-location offsets of :801-888:
 *   buffer row 2 = physical row ny_global, row 1 = ny_global-1
 *   center  : ghost(i,ny+1) = s*B(nx-i+1, row2)                 (top row untouched)
 *   NEcorner: B(.,row2) symmetrised for i=1..nx/2-1 (:818-824); then
 *             top(i,ny)   = s*B(nx-i, row2),  ghost(i,ny+1) = s*B(nx-i, row1),  index 0 -> nx
 * ------------------------------------------------------------------------- */
typedef struct { double *G; unsigned char *have; } gbuf;

static int wrap_i(const orc_geom *g, int gi) {
    if (gi < 1) return (g->ew_boundary == ORC_BND_CYCLIC) ? gi + g->nx_global : 0;
    if (gi > g->nx_global) return (g->ew_boundary == ORC_BND_CYCLIC) ? gi - g->nx_global : 0;
    return gi;
}

#define GIX(gi, gj) ((size_t)((gj)-1) * (size_t)g->nx_global + (size_t)((gi)-1))

static void halo_generic(const orc_geom *g, double *a, const double *asrc, int loc, int kind, double fill,
                         int stress_mode) {
    const int nx = g->nx_block, ny = g->ny_block;
    const int nxg = g->nx_global, nyg = g->ny_global;
    const size_t nn = (size_t)nx * ny;
    double *G = (double *)malloc(sizeof(double) * (size_t)nxg * nyg);
    unsigned char *have = (unsigned char *)calloc((size_t)nxg * nyg, 1);      /* cells some local block owns */
#pragma omp parallel for schedule(static)
    for (long long k = 0; k < (long long)nxg * nyg; k++) G[k] = fill;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < g->nblocks; b++) {
        const double *ab = asrc + (size_t)b * nn;
        for (int j = g->jlo[b]; j <= g->jhi[b]; j++)
            for (int i = g->ilo[b]; i <= g->ihi[b]; i++) {
                int gi = g->iglob_lo[b] + (i - g->ilo[b]), gj = g->jglob_lo[b] + (j - g->jlo[b]);
                G[GIX(gi, gj)] = ab[IX(i, j)];
                have[GIX(gi, gj)] = 1;
            }
    }
    const int tripole = (g->ns_boundary == ORC_BND_TRIPOLE);
    const double s = (kind == ORC_KIND_VECTOR) ? -1.0 : 1.0;
    double *top = NULL, *north = NULL;   /* 1-based global i */
    if (tripole) {
        top = (double *)malloc(sizeof(double) * (size_t)(nxg + 1));
        north = (double *)malloc(sizeof(double) * (size_t)(nxg + 1));
        double *row2 = (double *)malloc(sizeof(double) * (size_t)(nxg + 1));
        for (int i = 1; i <= nxg; i++) row2[i] = G[GIX(i, nyg)];
        /* offsets and top-row symmetrisation by field location (serial/ice_boundary.F90:801-846, u-fold):
           centre 0,0; NE corner 1,1 with pairs (i, nx-i), i = 1..nx/2-1; E face 1,0; N face 0,1 with pairs (i, nx+1-i), i = 1..nx/2 */
        const int ioff = (loc == ORC_LOC_NECORNER || loc == ORC_LOC_EFACE) ? 1 : 0;
        const int joff = (loc == ORC_LOC_NECORNER || loc == ORC_LOC_NFACE) ? 1 : 0;
        if (stress_mode) {
            for (int i = 1; i <= nxg; i++) north[i] = s * row2[nxg - i + 1];
        } else {
            if (loc == ORC_LOC_NECORNER)
                for (int i = 1; i <= nxg / 2 - 1; i++) {                  /* :818-824 */
                    int id = nxg - i;
                    double x1 = row2[i], x2 = row2[id];
                    double xavg = 0.5 * (x1 + s * x2);
                    row2[i] = xavg;
                    row2[id] = s * xavg;
                }
            if (loc == ORC_LOC_NFACE)
                for (int i = 1; i <= nxg / 2; i++) {                      /* :836-843 */
                    int id = nxg + 1 - i;
                    double x1 = row2[i], x2 = row2[id];
                    double xavg = 0.5 * (x1 + s * x2);
                    row2[i] = xavg;
                    row2[id] = s * xavg;
                }
            for (int i = 1; i <= nxg; i++) {
                int is = nxg - i + 1 - ioff;                              /* :872-876 */
                if (is == 0) is = nxg;
                if (is > nxg) is -= nxg;
                if (joff) { top[i] = s * row2[is]; north[i] = s * G[GIX(is, nyg - 1)]; }
                else north[i] = s * row2[is];
            }
        }
        free(row2);
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < g->nblocks; b++) {
        double *ab = a + (size_t)b * nn;
        const int ilo = g->ilo[b], ihi = g->ihi[b], jlo = g->jlo[b], jhi = g->jhi[b];
        const int top_block = tripole && (g->jglob_lo[b] + (jhi - jlo) == nyg);
        for (int j = 1; j <= ny; j++)
            for (int i = 1; i <= nx; i++) {
                const int phys = (i >= ilo && i <= ihi && j >= jlo && j <= jhi);
                int gi = wrap_i(g, g->iglob_lo[b] + (i - ilo));
                int gj = g->jglob_lo[b] + (j - jlo);
                if (stress_mode) {
                    /* ice_HaloUpdate_stress (serial/ice_boundary.F90:3269-3443): the tripole north ghost row of array1 is
                       written from array2's top row, a ghost cell whose neighbour block is an eliminated land block gets
                       `fill` (:3349-3350), nothing else is touched and nothing is pre-filled.  (Tripole grids are cyclic
                       E-W: with 'open' / 'closed' the reference's copy out of the tripole buffer reads mirrored or
                       out-of-range columns, :3752-3776 with :3420-3424 -- not supported here.) */
                    if (phys || i > ihi + 1 || j > jhi + 1) continue;
                    if (top_block && j == jhi + 1) { if (gi > 0) ab[IX(i, j)] = north[gi]; }
                    else if (gi > 0 && gj >= 1 && gj <= nyg && !have[GIX(gi, gj)]) ab[IX(i, j)] = fill;
                    continue;
                }
                if (phys) {
                    if (top_block && j == jhi && (loc == ORC_LOC_NECORNER || loc == ORC_LOC_NFACE)) ab[IX(i, j)] = top[gi];
                    continue;
                }
                /* a ghost cell takes its neighbour's value (a message of the halo schedule; `fill` if that neighbour is an
                   eliminated land block, serial/ice_boundary.F90:722-723).  A cell no message writes -- beyond an open /
                   closed boundary, or padding of a short block -- keeps its value, except on the outermost nghost rows /
                   columns of the block ARRAY, which the production backend fills first (mpi/ice_boundary.F90:1409-1416).
                   Pinned by tests/golden/ref_*.npz (the reference's serial backend + that one rule). */
                double v = fill;
                int has_src = 0;
                if (!(i > ihi + 1 || j > jhi + 1)) {
                    if (gi > 0) {
                        if (gj < 1) { if (g->ns_boundary == ORC_BND_CYCLIC) { v = G[GIX(gi, gj + nyg)]; has_src = 1; } }
                        else if (gj > nyg) {
                            if (g->ns_boundary == ORC_BND_CYCLIC) { v = G[GIX(gi, gj - nyg)]; has_src = 1; }
                            else if (tripole && gj == nyg + 1) { v = north[gi]; has_src = 1; }
                        } else {
                            v = G[GIX(gi, gj)]; has_src = 1;
                            if (top_block && j == jhi && (loc == ORC_LOC_NECORNER || loc == ORC_LOC_NFACE)) v = top[gi];
                        }
                    }
                }
                if (has_src || i == 1 || i == nx || j == 1 || j == ny) ab[IX(i, j)] = v;
            }
    }
    free(G); free(have); free(top); free(north);
}

/* ---------------------------------------------------------------------------
 * The same update as halo_generic for the call that sits inside the subcycle loop (u, v at the NE corner, once per
 * subcycle), done the way the reference does it: a precomputed schedule of ghost-cell addresses (the `ice_halo` type,
 * mpi/ice_boundary.F90:42-76, built once by ice_HaloCreate) -- copy lists ghost <- physical cell of the owning block,
 * fill lists, and for the tripole fold two global ROWS instead of a whole global array.  Identical results
 * (tests: the goldens and every oracle-vs-oracle comparison run through it); used so that the CPU baseline of bench.py
 * is not charged for gathering an nx_global x ny_global array twice per subcycle.
 * ------------------------------------------------------------------------- */
typedef struct {
    size_t ncopy, nfill, ntop, nnorth, nrow;
    size_t *cdst, *csrc, *fdst;            /* ghost <- physical cell; ghost <- fill */
    size_t *tdst; int *tgi;                /* top physical row / its E-W ghosts <- top[gi]   (tripole, NE corner) */
    size_t *ndst; int *ngi;                /* north ghost row <- north[gi] */
    size_t *r2src, *r1src; int *rgi;       /* global rows ny and ny-1 gathered from the top blocks' physical cells */
    int nxg;
} halo_sched;

static void sched_free(halo_sched *h) {
    free(h->cdst); free(h->csrc); free(h->fdst); free(h->tdst); free(h->tgi); free(h->ndst); free(h->ngi);
    free(h->r2src); free(h->r1src); free(h->rgi);
    memset(h, 0, sizeof(*h));
}

/* schedule of an NE-corner field (loc = ORC_LOC_NECORNER), any kind */
static void sched_build_ne(const orc_geom *g, halo_sched *h) {
    const int nx = g->nx_block, ny = g->ny_block, nxg = g->nx_global, nyg = g->ny_global;
    const size_t nn = (size_t)nx * ny, cap = nn * (size_t)g->nblocks;
    memset(h, 0, sizeof(*h));
    h->nxg = nxg;
    h->cdst = malloc(cap * sizeof(size_t)); h->csrc = malloc(cap * sizeof(size_t)); h->fdst = malloc(cap * sizeof(size_t));
    h->tdst = malloc(cap * sizeof(size_t)); h->tgi = malloc(cap * sizeof(int));
    h->ndst = malloc(cap * sizeof(size_t)); h->ngi = malloc(cap * sizeof(int));
    h->r2src = malloc((size_t)nxg * sizeof(size_t)); h->r1src = malloc((size_t)nxg * sizeof(size_t)); h->rgi = malloc((size_t)nxg * sizeof(int));
    /* owner of every global cell: (block, local index), -1 where no local block covers it (eliminated land blocks) */
    long long *own = malloc(sizeof(long long) * (size_t)nxg * nyg);
    for (size_t k = 0; k < (size_t)nxg * nyg; k++) own[k] = -1;
    for (int b = 0; b < g->nblocks; b++)
        for (int j = g->jlo[b]; j <= g->jhi[b]; j++)
            for (int i = g->ilo[b]; i <= g->ihi[b]; i++) {
                int gi = g->iglob_lo[b] + (i - g->ilo[b]), gj = g->jglob_lo[b] + (j - g->jlo[b]);
                own[GIX(gi, gj)] = (long long)((size_t)b * nn + IX(i, j));
            }
    const int tripole = (g->ns_boundary == ORC_BND_TRIPOLE);
    if (tripole)
        for (int gi = 1; gi <= nxg; gi++)
            if (own[GIX(gi, nyg)] >= 0 && nyg >= 2 && own[GIX(gi, nyg - 1)] >= 0) {
                h->r2src[h->nrow] = (size_t)own[GIX(gi, nyg)]; h->r1src[h->nrow] = (size_t)own[GIX(gi, nyg - 1)]; h->rgi[h->nrow++] = gi;
            }
    for (int b = 0; b < g->nblocks; b++) {
        const int ilo = g->ilo[b], ihi = g->ihi[b], jlo = g->jlo[b], jhi = g->jhi[b];
        const int top_block = tripole && (g->jglob_lo[b] + (jhi - jlo) == nyg);
        for (int j = 1; j <= ny; j++)
            for (int i = 1; i <= nx; i++) {
                const size_t d = (size_t)b * nn + IX(i, j);
                const int phys = (i >= ilo && i <= ihi && j >= jlo && j <= jhi);
                int gi = wrap_i(g, g->iglob_lo[b] + (i - ilo));
                int gj = g->jglob_lo[b] + (j - jlo);
                if (phys) {
                    if (top_block && j == jhi) { h->tdst[h->ntop] = d; h->tgi[h->ntop++] = gi; }
                    continue;
                }
                const int edge = (i == 1 || i == nx || j == 1 || j == ny);     /* cells without a source: see halo_generic */
                if (i > ihi + 1 || j > jhi + 1 || gi <= 0) { if (edge) h->fdst[h->nfill++] = d; continue; }
                if (gj < 1) {
                    if (g->ns_boundary == ORC_BND_CYCLIC && own[GIX(gi, gj + nyg)] >= 0) { h->cdst[h->ncopy] = d; h->csrc[h->ncopy++] = (size_t)own[GIX(gi, gj + nyg)]; }
                    else if (edge) h->fdst[h->nfill++] = d;
                } else if (gj > nyg) {
                    if (g->ns_boundary == ORC_BND_CYCLIC && own[GIX(gi, gj - nyg)] >= 0) { h->cdst[h->ncopy] = d; h->csrc[h->ncopy++] = (size_t)own[GIX(gi, gj - nyg)]; }
                    else if (tripole && gj == nyg + 1) { h->ndst[h->nnorth] = d; h->ngi[h->nnorth++] = gi; }
                    else if (edge) h->fdst[h->nfill++] = d;
                } else if (top_block && j == jhi) { h->tdst[h->ntop] = d; h->tgi[h->ntop++] = gi; }
                else if (own[GIX(gi, gj)] >= 0) { h->cdst[h->ncopy] = d; h->csrc[h->ncopy++] = (size_t)own[GIX(gi, gj)]; }
                else h->fdst[h->nfill++] = d;
            }
    }
    free(own);
}

static void sched_apply_ne(const halo_sched *h, double *a, int kind, double fill) {
    const double s = (kind == ORC_KIND_VECTOR) ? -1.0 : 1.0;
    const int nxg = h->nxg;
    double *top = NULL, *north = NULL;
    if (h->ntop || h->nnorth) {      /* the fold reads the rows BEFORE any ghost or top-row cell is rewritten */
        double *row2 = (double *)malloc(sizeof(double) * (size_t)(nxg + 1)), *row1 = (double *)malloc(sizeof(double) * (size_t)(nxg + 1));
        top = (double *)malloc(sizeof(double) * (size_t)(nxg + 1)); north = (double *)malloc(sizeof(double) * (size_t)(nxg + 1));
        for (int i = 0; i <= nxg; i++) row2[i] = row1[i] = fill;
        for (size_t k = 0; k < h->nrow; k++) { row2[h->rgi[k]] = a[h->r2src[k]]; row1[h->rgi[k]] = a[h->r1src[k]]; }
        for (int i = 1; i <= nxg / 2 - 1; i++) {                          /* serial/ice_boundary.F90:818-824 */
            int id = nxg - i;
            double x1 = row2[i], x2 = row2[id];
            double xavg = 0.5 * (x1 + s * x2);
            row2[i] = xavg;
            row2[id] = s * xavg;
        }
        for (int i = 1; i <= nxg; i++) {
            int is = nxg - i; if (is == 0) is = nxg;                      /* :874-876 */
            top[i] = s * row2[is];
            north[i] = s * row1[is];
        }
        free(row2); free(row1);
    }
#pragma omp parallel
    {
#pragma omp for schedule(static) nowait
        for (long long k = 0; k < (long long)h->ncopy; k++) a[h->cdst[k]] = a[h->csrc[k]];
#pragma omp for schedule(static) nowait
        for (long long k = 0; k < (long long)h->nfill; k++) a[h->fdst[k]] = fill;
#pragma omp for schedule(static) nowait
        for (long long k = 0; k < (long long)h->nnorth; k++) a[h->ndst[k]] = north[h->ngi[k]];
    }
    /* (copies read physical cells only; the top-row rewrite touches physical cells of the top row, which no copy of a
       NE-corner update reads as a source in halo_generic either: its sources on that row take top[gi] -- done last) */
    for (size_t k = 0; k < h->ntop; k++) a[h->tdst[k]] = top[h->tgi[k]];
    free(top); free(north);
}

/* Optional hook for multi-process CPU tests.  Phase 0 runs before the local update (the hook
   can save the rows a tripole fold needs), phase 1 after it (sources on other ranks read as
   `fill` locally; the hook patches the cells whose source lives elsewhere). */
static orc_halo_cb g_halo_cb = NULL;
static void *g_halo_cb_user = NULL;
void orc_set_halo_callback(orc_halo_cb cb, void *user) { g_halo_cb = cb; g_halo_cb_user = user; }

void orc_halo_r8(const orc_geom *g, double *a, int loc, int kind, double fill) {
    if (g_halo_cb) g_halo_cb(a, loc, kind, fill, 0, g_halo_cb_user);   /* phase 0: before the local update */
    halo_generic(g, a, a, loc, kind, fill, 0);
    if (g_halo_cb) g_halo_cb(a, loc, kind, fill, 1, g_halo_cb_user);   /* phase 1: patch remote sources */
}

void orc_halo_stress(const orc_geom *g, double *a1, const double *a2) {
    halo_generic(g, a1, a2, ORC_LOC_CENTER, ORC_KIND_SCALAR, 0.0, 1);
}

void orc_halo_i4(const orc_geom *g, int32_t *a, int32_t fill) {
    /* integers in this path are 0/1 masks: exact in double */
    size_t n = (size_t)g->nx_block * g->ny_block * g->nblocks;
    double *t = (double *)malloc(n * sizeof(double));
    for (size_t k = 0; k < n; k++) t[k] = (double)a[k];
    orc_halo_r8(g, t, ORC_LOC_CENTER, ORC_KIND_SCALAR, (double)fill);
    for (size_t k = 0; k < n; k++) a[k] = (int32_t)t[k];
    free(t);
}

/* ---------------------------------------------------------------------------
 * compute_tracers (source/ice_itd.F90:1359-1501) on ALL cells of one block (work_to_state's list, ice_transport_driver.F90:
 * 1571-1580): trcrn from the advected products atrcrn = (aicen | vicen | vsnon [* parent tracer]) * trcrn.
 * trcrn: (ntrcr_dim, ny, nx) planes of this block and category, atr: (ntrcr, ny, nx); nt_* 1-based, 0 = tracer absent.
 * ------------------------------------------------------------------------- */
void orc_compute_tracers(int nx, int ny, int ntrcr, const int32_t *trcr_depend, int nt_Tsfc, int nt_alvl, int nt_apnd, int nt_fbri,
                         int tr_pond_cesm, int tr_pond_lvl, int tr_pond_topo, double Tocnfrz, const double *atr,
                         const double *aicen, const double *vicen, const double *vsnon, double *trcrn) {
    const size_t nn = (size_t)nx * ny;
    for (int it = 1; it <= ntrcr; it++)
        for (size_t k = 0; k < nn; k++) trcrn[(size_t)(it - 1) * nn + k] = c0;                       /* :1405 */
    for (int it = 1; it <= ntrcr; it++) {
        double *t = trcrn + (size_t)(it - 1) * nn;
        const double *a = atr + (size_t)(it - 1) * nn;
        const int dep = trcr_depend[it - 1];
        for (size_t k = 0; k < nn; k++) {
            if (it == nt_Tsfc) t[k] = aicen[k] > puny ? a[k] / aicen[k] : Tocnfrz;                     /* :1413-1422 */
            else if (dep == 0) t[k] = aicen[k] > puny ? a[k] / aicen[k] : c0;
            else if (dep == 1) {
                if (vicen[k] > c0) t[k] = a[k] / vicen[k];
                else { t[k] = c0; if (it == nt_fbri) t[k] = c1; }
            } else if (dep == 2) t[k] = vsnon[k] > c0 ? a[k] / vsnon[k] : c0;
            else if (nt_alvl > 0 && dep == 2 + nt_alvl) {
                const double d = trcrn[(size_t)(nt_alvl - 1) * nn + k] * aicen[k];
                t[k] = d > c0 ? a[k] / d : c0;
            } else if (nt_apnd > 0 && dep == 2 + nt_apnd && (tr_pond_cesm || tr_pond_topo)) {
                const double d = trcrn[(size_t)(nt_apnd - 1) * nn + k] * aicen[k];
                t[k] = d > c0 ? a[k] / d : c0;
            } else if (nt_apnd > 0 && dep == 2 + nt_apnd && tr_pond_lvl) {
                const double d = trcrn[(size_t)(nt_alvl - 1) * nn + k] * trcrn[(size_t)(nt_apnd - 1) * nn + k] * aicen[k];
                t[k] = d > c0 ? a[k] / d : c0;
            } else if (nt_fbri > 0 && dep == 2 + nt_fbri) {
                const double d = trcrn[(size_t)(nt_fbri - 1) * nn + k] * vicen[k];
                t[k] = d > c0 ? a[k] / d : c0;
            }
        }
    }
}

/* ---------------------------------------------------------------------------
 * transport_upwind WHOLE (ice_transport_driver.F90:634-772): state_to_work (:1382-1513) -> upwind_field -> work_to_state
 * (:1520-1609, compute_tracers) -> bound_state (ice_state.F90:173-238).  aice0 (nb, ny, nx), aicen / vicen / vsnon
 * (nb, ncat, ny, nx), trcrn (nb, ncat, ntrcr_dim, ny, nx), in place, every cell of every block.  Ghost cells current on entry.
 * state_to_work's pond branch is written `a .and. tr_pond_cesm .or. tr_pond_topo` (:1480-1481): with topo ponds it takes every
 * tracer that reaches it -- restated as written.
 * ------------------------------------------------------------------------- */
void orc_transport_upwind_state(const orc_geom *g, double dt, int ncat, int ntrcr, int ntrcr_dim, const int32_t *trcr_depend,
                                int nt_Tsfc, int nt_alvl, int nt_apnd, int nt_fbri, int tr_pond_cesm, int tr_pond_lvl, int tr_pond_topo,
                                double Tocnfrz, const double *uvel, const double *vvel, const double *HTE, const double *HTN,
                                const double *tarea, double *aice0, double *aicen, double *vicen, double *vsnon, double *trcrn) {
    const int nx = g->nx_block, ny = g->ny_block, nb = g->nblocks;
    const size_t nn = (size_t)nx * ny;
    const int narr = 1 + ncat * (3 + ntrcr);
    double *works = (double *)calloc((size_t)nb * narr * nn, sizeof(double));
    for (int b = 0; b < nb; b++) {
        double *w = works + (size_t)b * narr * nn;
        memcpy(w, aice0 + (size_t)b * nn, nn * sizeof(double));
        int na = 1;
        for (int n = 0; n < ncat; n++) {
            const double *a = aicen + ((size_t)b * ncat + n) * nn, *v = vicen + ((size_t)b * ncat + n) * nn, *sn = vsnon + ((size_t)b * ncat + n) * nn;
            const double *t = trcrn + ((size_t)b * ncat + n) * ntrcr_dim * nn;
            memcpy(w + (size_t)(na + 0) * nn, a, nn * 8); memcpy(w + (size_t)(na + 1) * nn, v, nn * 8); memcpy(w + (size_t)(na + 2) * nn, sn, nn * 8);
            na += 3;
            for (int it = 1; it <= ntrcr; it++) {
                double *o = w + (size_t)(na + it - 1) * nn;
                const double *ti = t + (size_t)(it - 1) * nn;
                const int dep = trcr_depend[it - 1];
                for (size_t k = 0; k < nn; k++) {
                    if (dep == 0) o[k] = a[k] * ti[k];
                    else if (dep == 1) o[k] = v[k] * ti[k];
                    else if (dep == 2) o[k] = sn[k] * ti[k];
                    else if (nt_alvl > 0 && dep == 2 + nt_alvl) o[k] = a[k] * t[(size_t)(nt_alvl - 1) * nn + k] * ti[k];
                    else if ((nt_apnd > 0 && dep == 2 + nt_apnd && tr_pond_cesm) || tr_pond_topo) o[k] = a[k] * t[(size_t)(nt_apnd - 1) * nn + k] * ti[k];
                    else if (nt_apnd > 0 && dep == 2 + nt_apnd && tr_pond_lvl) o[k] = a[k] * t[(size_t)(nt_alvl - 1) * nn + k] * t[(size_t)(nt_apnd - 1) * nn + k] * ti[k];
                    else if (nt_fbri > 0 && dep == 2 + nt_fbri) o[k] = v[k] * t[(size_t)(nt_fbri - 1) * nn + k] * ti[k];
                    /* else: works(:,:,narrays+it) is left as allocated (the reference never reads a tracer without a rule back) */
                }
            }
            na += ntrcr;
        }
    }
    orc_transport_upwind(g, dt, narr, uvel, vvel, HTE, HTN, tarea, works);
    for (int b = 0; b < nb; b++) {
        const double *w = works + (size_t)b * narr * nn;
        memcpy(aice0 + (size_t)b * nn, w, nn * sizeof(double));
        int na = 1;
        for (int n = 0; n < ncat; n++) {
            double *a = aicen + ((size_t)b * ncat + n) * nn, *v = vicen + ((size_t)b * ncat + n) * nn, *sn = vsnon + ((size_t)b * ncat + n) * nn;
            memcpy(a, w + (size_t)(na + 0) * nn, nn * 8); memcpy(v, w + (size_t)(na + 1) * nn, nn * 8); memcpy(sn, w + (size_t)(na + 2) * nn, nn * 8);
            na += 3;
            orc_compute_tracers(nx, ny, ntrcr, trcr_depend, nt_Tsfc, nt_alvl, nt_apnd, nt_fbri, tr_pond_cesm, tr_pond_lvl, tr_pond_topo, Tocnfrz,
                                w + (size_t)na * nn, a, v, sn, trcrn + ((size_t)b * ncat + n) * ntrcr_dim * nn);
            na += ntrcr;
        }
    }
    free(works);
    /* bound_state: aicen, trcrn(1:ntrcr), vicen, vsnon -- centre scalars (aice0 has no halo update of its own) */
    double *pl = (double *)malloc((size_t)nb * nn * sizeof(double));
    for (int n = 0; n < ncat; n++)
        for (int q = 0; q < 3 + ntrcr; q++) {
            for (int b = 0; b < nb; b++) {
                const double *src = q == 0 ? aicen + ((size_t)b * ncat + n) * nn : q == 1 ? vicen + ((size_t)b * ncat + n) * nn
                                  : q == 2 ? vsnon + ((size_t)b * ncat + n) * nn : trcrn + (((size_t)b * ncat + n) * ntrcr_dim + (q - 3)) * nn;
                memcpy(pl + (size_t)b * nn, src, nn * 8);
            }
            orc_halo_r8(g, pl, ORC_LOC_CENTER, ORC_KIND_SCALAR, 0.0);
            for (int b = 0; b < nb; b++) {
                double *dst = q == 0 ? aicen + ((size_t)b * ncat + n) * nn : q == 1 ? vicen + ((size_t)b * ncat + n) * nn
                            : q == 2 ? vsnon + ((size_t)b * ncat + n) * nn : trcrn + (((size_t)b * ncat + n) * ntrcr_dim + (q - 3)) * nn;
                memcpy(dst, pl + (size_t)b * nn, nn * 8);
            }
        }
    free(pl);
}

/* ---------------------------------------------------------------------------
 * transport_upwind (source/ice_transport_driver.F90:634-772), see evp_oracle.h
 * ------------------------------------------------------------------------- */
void orc_transport_upwind(const orc_geom *g, double dt, int narr, const double *uvel, const double *vvel,
                          const double *HTE, const double *HTN, const double *tarea, double *works) {
    const int nx = g->nx_block, ny = g->ny_block, nb = g->nblocks;
    const size_t nn = (size_t)nx * ny, tot = nn * nb;
    double *uee = calloc(tot, 8), *vnn = calloc(tot, 8);
    for (int b = 0; b < nb; b++) {                                        /* :688-701 */
        const size_t o = (size_t)b * nn;
        for (int j = g->jlo[b]; j <= g->jhi[b]; j++)
            for (int i = g->ilo[b]; i <= g->ihi[b]; i++) {
                uee[o + IX(i, j)] = p5 * (uvel[o + IX(i, j)] + uvel[o + IX(i, j - 1)]);
                vnn[o + IX(i, j)] = p5 * (vvel[o + IX(i, j)] + vvel[o + IX(i - 1, j)]);
            }
    }
    orc_halo_r8(g, uee, ORC_LOC_EFACE, ORC_KIND_VECTOR, 0.0);             /* :703-708 */
    orc_halo_r8(g, vnn, ORC_LOC_NFACE, ORC_KIND_VECTOR, 0.0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < nb; b++) {
        const size_t o = (size_t)b * nn;
        const int ilo = g->ilo[b], ihi = g->ihi[b], jlo = g->jlo[b], jhi = g->jhi[b];
        double *worka = malloc(nn * 8), *workb = malloc(nn * 8);
        for (int n = 0; n < narr; n++) {                                  /* upwind_field, :1667-1687 */
            double *phi = works + ((size_t)b * narr + n) * nn;
            for (int j = 1; j <= jhi; j++)
                for (int i = 1; i <= ihi; i++) {
                    /* upwind(y1,y2,a,h) = p5*dt*h*((a+abs(a))*y1+(a-abs(a))*y2)   (:1661) */
                    double a = uee[o + IX(i, j)], h = HTE[o + IX(i, j)];
                    worka[IX(i, j)] = p5 * dt * h * ((a + fabs(a)) * phi[IX(i, j)] + (a - fabs(a)) * phi[IX(i + 1, j)]);
                    a = vnn[o + IX(i, j)]; h = HTN[o + IX(i, j)];
                    workb[IX(i, j)] = p5 * dt * h * ((a + fabs(a)) * phi[IX(i, j)] + (a - fabs(a)) * phi[IX(i, j + 1)]);
                }
            for (int j = jlo; j <= jhi; j++)
                for (int i = ilo; i <= ihi; i++)
                    phi[IX(i, j)] = phi[IX(i, j)] - (worka[IX(i, j)] - worka[IX(i - 1, j)]
                                                     + workb[IX(i, j)] - workb[IX(i, j - 1)]) / tarea[o + IX(i, j)];
        }
        free(worka); free(workb);
    }
    free(uee); free(vnn);
}

/* ---------------------------------------------------------------------------
 * evp(dt)  (source/ice_dyn_evp.F90:68-510)
 * ------------------------------------------------------------------------- */
/* evp(dt) (ice_dyn_evp.F90:68-510) and, with e != NULL, eap(dt) (ice_dyn_eap.F90:66-486): the same driver around another stress */
static void dyn_driver(const orc_geom *g, const orc_params *p, orc_fields *f, orc_eap_state *e, int nsub_override,
                       int64_t counts[2], double *loop_seconds) {
    const int nx = g->nx_block, ny = g->ny_block, nb = g->nblocks;
    const size_t nn = (size_t)nx * ny, tot = nn * nb;
    double *waterx = calloc(tot, 8), *watery = calloc(tot, 8), *forcex = calloc(tot, 8), *forcey = calloc(tot, 8);
    double *umassdti = calloc(tot, 8), *work1 = calloc(tot, 8);
    int32_t *indxti = malloc(tot * 4), *indxtj = malloc(tot * 4), *indxui = malloc(tot * 4), *indxuj = malloc(tot * 4);
    int32_t *icellt = calloc(nb, 4), *icellu = calloc(nb, 4);
    double *sp[4], *sm[4], *s12[4];

    for (int b = 0; b < nb; b++) {                                        /* :171-203 */
        size_t o = (size_t)b * nn;
        for (size_t k = 0; k < nn; k++) {
            f->rdg_conv[o + k] = c0; f->rdg_shear[o + k] = c0; f->divu[o + k] = c0;
            f->shear[o + k] = c0; f->prs_sig[o + k] = c0;
            if (e) {                                                      /* ice_dyn_eap.F90:171-180 */
                e->e11[o + k] = c0; e->e12[o + k] = c0; e->e22[o + k] = c0; e->s11[o + k] = c0; e->s12[o + k] = c0; e->s22[o + k] = c0;
                e->yieldstress11[o + k] = c0; e->yieldstress12[o + k] = c0; e->yieldstress22[o + k] = c0;
            }
        }
        orc_evp_prep1(nx, ny, g->ilo[b], g->ihi[b], g->jlo[b], g->jhi[b],
                      f->aice + o, f->vice + o, f->vsno + o, f->tmask + o,
                      f->strairxT + o, f->strairyT + o, f->strairx + o, f->strairy + o,
                      f->tmass + o, f->icetmask + o, p);
    }
    orc_halo_i4(g, f->icetmask, 0);                                       /* :210-211 */

    for (int b = 0; b < nb; b++) {                                        /* :218-219 */
        size_t o = (size_t)b * nn;
        orc_to_ugrid_blk(nx, ny, g->ilo[b], g->ihi[b], g->jlo[b], g->jhi[b], f->tmass + o, f->tarea + o, f->uarea + o, f->umass + o);
        orc_to_ugrid_blk(nx, ny, g->ilo[b], g->ihi[b], g->jlo[b], g->jhi[b], f->aice_init + o, f->tarea + o, f->uarea + o, f->aiu + o);
    }
    if (p->wind_on_ugrid) {                                               /* :226-228 */
        memcpy(f->strairx, f->strax, tot * 8);
        memcpy(f->strairy, f->stray, tot * 8);
    } else {                                                              /* :240-241, ice_grid.F90:1799-1823 */
        double *w[2] = { f->strairx, f->strairy };
        for (int c = 0; c < 2; c++) {
            memcpy(work1, w[c], tot * 8);
            orc_halo_r8(g, work1, ORC_LOC_CENTER, ORC_KIND_VECTOR, 0.0);
            for (int b = 0; b < nb; b++) {
                size_t o = (size_t)b * nn;
                orc_to_ugrid_blk(nx, ny, g->ilo[b], g->ihi[b], g->jlo[b], g->jhi[b], work1 + o, f->tarea + o, f->uarea + o, w[c] + o);
            }
        }
    }

    for (int b = 0; b < nb; b++) {                                        /* :247-308 */
        size_t o = (size_t)b * nn;
        for (int c = 0; c < 4; c++) { sp[c] = f->stressp[c] + o; sm[c] = f->stressm[c] + o; s12[c] = f->stress12[c] + o; }
        orc_evp_prep2(nx, ny, g->ilo[b], g->ihi[b], g->jlo[b], g->jhi[b], &icellt[b], &icellu[b],
                      indxti + o, indxtj + o, indxui + o, indxuj + o,
                      f->aiu + o, f->umass + o, umassdti + o, f->fcor + o, f->umask + o,
                      f->uocn + o, f->vocn + o, f->strairx + o, f->strairy + o, f->ss_tltx + o, f->ss_tlty + o,
                      f->icetmask + o, f->iceumask + o, f->fm + o, p->dt,
                      f->strtltx + o, f->strtlty + o, f->strocnx + o, f->strocny + o,
                      f->strintx + o, f->strinty + o, waterx + o, watery + o, forcex + o, forcey + o,
                      sp, sm, s12, f->uvel_init + o, f->vvel_init + o, f->uvel + o, f->vvel + o, p);
        if (e)                                                            /* structure tensor where there is no ice (ice_dyn_eap.F90:284-298) */
            for (size_t k = 0; k < nn; k++)
                if (f->icetmask[o + k] == 0)
                    for (int c = 0; c < 4; c++) { e->a11[c][o + k] = 0.5; e->a12[c][o + k] = c0; }
        /* ice_strength (:291-301): an input (f->strength already holds it on physical cells) unless strength_mode = 1 */
        if (p->strength_mode)
            orc_ice_strength(nx, ny, g->ilo[b], g->ihi[b], g->jlo[b], g->jhi[b], icellt[b], indxti + o, indxtj + o,
                             f->aice + o, f->vice + o, f->aice0 ? f->aice0 + o : NULL,
                             f->aicen ? f->aicen + o * (size_t)p->ncat : NULL, f->vicen ? f->vicen + o * (size_t)p->ncat : NULL,
                             f->strength + o, p);
    }
    orc_halo_r8(g, f->strength, ORC_LOC_CENTER, ORC_KIND_SCALAR, 0.0);    /* :311-312 */
    orc_halo_r8(g, f->uvel, ORC_LOC_NECORNER, ORC_KIND_VECTOR, 0.0);      /* :314-315 (fld2 = u,v) */
    orc_halo_r8(g, f->vvel, ORC_LOC_NECORNER, ORC_KIND_VECTOR, 0.0);

    int64_t nt = 0, nu = 0;
    for (int b = 0; b < nb; b++) {
        nu += icellu[b];
        /* count physical T cells only (the N/E ghost T cells are redundant work) */
        for (int ij = 0; ij < icellt[b]; ij++) {
            size_t o = (size_t)b * nn + ij;
            if (indxti[o] <= g->ihi[b] && indxtj[o] <= g->jhi[b]) nt++;
        }
    }
    if (counts) { counts[0] = nt; counts[1] = nu; }

    const int nsub = nsub_override > 0 ? nsub_override : p->ndte;
    /* one str(nx,ny,8) work array per thread for the whole loop (the reference's is an automatic array of stress) */
    int nthr = 1;
#ifdef _OPENMP
    nthr = omp_get_max_threads();
#endif
    double **strbuf = (double **)malloc(sizeof(double *) * (size_t)nthr);
    for (int t = 0; t < nthr; t++) strbuf[t] = (double *)malloc(8 * nn * sizeof(double));
    halo_sched hs;                  /* the halo schedule of (u, v): built once, like ice_HaloCreate's */
    sched_build_ne(g, &hs);
    double t0 = wall(), t_halo = 0.0;
    for (int ksub = 1; ksub <= nsub; ksub++) {                            /* :336-410 */
#pragma omp parallel
        {
            int me = 0;
#ifdef _OPENMP
            me = omp_get_thread_num();
#endif
            double *strtmp = strbuf[me];
            double *tsp[4], *tsm[4], *ts12[4];
#pragma omp for schedule(dynamic, 1)
            for (int b = 0; b < nb; b++) {
                size_t o = (size_t)b * nn;
                for (int c = 0; c < 4; c++) { tsp[c] = f->stressp[c] + o; tsm[c] = f->stressm[c] + o; ts12[c] = f->stress12[c] + o; }
                if (e)
                    orc_eap_stress(nx, ny, ksub, p->ndte, icellt[b], indxti + o, indxtj + o, p->arlx1i, p->denom1,
                                   f->uvel + o, f->vvel + o, f->dxt + o, f->dyt + o, f->dxhy + o, f->dyhx + o,
                                   f->cxp + o, f->cyp + o, f->cxm + o, f->cym + o, f->tarear + o, f->strength + o, tsp, tsm, ts12,
                                   f->shear + o, f->divu + o, f->prs_sig + o, f->rdg_conv + o, f->rdg_shear + o, strtmp, e, o);
                else
                orc_stress(nx, ny, ksub, p->ndte, icellt[b], indxti + o, indxtj + o,
                           f->uvel + o, f->vvel + o, f->dxt + o, f->dyt + o, f->dxhy + o, f->dyhx + o,
                           f->cxp + o, f->cyp + o, f->cxm + o, f->cym + o, f->tarear + o, f->tinyarea + o,
                           f->strength + o, tsp, tsm, ts12, f->shear + o, f->divu + o, f->prs_sig + o,
                           f->rdg_conv + o, f->rdg_shear + o, strtmp, p);
                orc_stepu(nx, ny, icellu[b], f->Cdn_ocn + o, indxui + o, indxuj + o, f->aiu + o, strtmp,
                          f->uocn + o, f->vocn + o, waterx + o, watery + o, forcex + o, forcey + o,
                          umassdti + o, f->fm + o, f->uarear + o, f->strocnx + o, f->strocny + o,
                          f->strintx + o, f->strinty + o, f->uvel_init + o, f->vvel_init + o,
                          f->uvel + o, f->vvel + o, p);
                if (e && ksub % 10 == 1)                                  /* ice_dyn_eap.F90:411-426 */
                    orc_eap_stepa(nx, ny, p->dtei, icellt[b], indxti + o, indxtj + o, tsp, tsm, ts12, e, o);
            }
        }
        const double th = wall();
        double *uv[2] = { f->uvel, f->vvel };                             /* :392-400 */
        for (int c = 0; c < 2; c++) {
            if (g_halo_cb) g_halo_cb(uv[c], ORC_LOC_NECORNER, ORC_KIND_VECTOR, 0.0, 0, g_halo_cb_user);
            sched_apply_ne(&hs, uv[c], ORC_KIND_VECTOR, 0.0);
            if (g_halo_cb) g_halo_cb(uv[c], ORC_LOC_NECORNER, ORC_KIND_VECTOR, 0.0, 1, g_halo_cb_user);
        }
        t_halo += wall() - th;
    }
    if (loop_seconds) { loop_seconds[0] = wall() - t0; loop_seconds[1] = t_halo; }
    sched_free(&hs);
    for (int t = 0; t < nthr; t++) free(strbuf[t]);
    free(strbuf);

    if (g->ns_boundary == ORC_BND_TRIPOLE && !e) {                        /* :416-481 (eap has no stress fold) */
        double **S[3] = { f->stressp, f->stressm, f->stress12 };
        for (int t = 0; t < 3; t++) {
            orc_halo_stress(g, S[t][0], S[t][2]);
            orc_halo_stress(g, S[t][2], S[t][0]);
            orc_halo_stress(g, S[t][1], S[t][3]);
            orc_halo_stress(g, S[t][3], S[t][1]);
        }
    }

    for (int b = 0; b < nb; b++) {                                        /* :487-503 */
        size_t o = (size_t)b * nn;
        orc_evp_finish(nx, ny, icellu[b], f->Cdn_ocn + o, indxui + o, indxuj + o, f->uvel + o, f->vvel + o,
                       f->uocn + o, f->vocn + o, f->aiu + o, f->fm + o,
                       f->strocnx + o, f->strocny + o, f->strocnxT + o, f->strocnyT + o, p);
    }
    double *w[2] = { f->strocnxT, f->strocnyT };                          /* :505-506, ice_grid.F90:1886-1910 */
    for (int c = 0; c < 2; c++) {
        memcpy(work1, w[c], tot * 8);
        orc_halo_r8(g, work1, ORC_LOC_NECORNER, ORC_KIND_VECTOR, 0.0);
        for (int b = 0; b < nb; b++) {
            size_t o = (size_t)b * nn;
            orc_to_tgrid_blk(nx, ny, g->ilo[b], g->ihi[b], g->jlo[b], g->jhi[b], work1 + o, f->tarea + o, f->uarea + o, w[c] + o);
        }
    }

    free(waterx); free(watery); free(forcex); free(forcey); free(umassdti); free(work1);
    free(indxti); free(indxtj); free(indxui); free(indxuj); free(icellt); free(icellu);
}

void orc_evp(const orc_geom *g, const orc_params *p, orc_fields *f, int nsub_override, int64_t counts[2], double *loop_seconds) {
    dyn_driver(g, p, f, NULL, nsub_override, counts, loop_seconds);
}

void orc_eap(const orc_geom *g, const orc_params *p, orc_fields *f, orc_eap_state *e, int nsub_override, int64_t counts[2], double *loop_seconds) {
    dyn_driver(g, p, f, e, nsub_override, counts, loop_seconds);
}
