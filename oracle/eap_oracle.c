/*
 * oracle/eap_oracle.c -- CPU restatement of the elastic-anisotropic-plastic rheology of CICE5 (source/ice_dyn_eap.F90:
 * stress_eap, update_stress_rdg, stepa, calc_ffrac), plain C99.  The driver (eap, :66-486) is orc_eap in evp_oracle.c: it
 * shares evp's preparation, stepu and finish as the reference does.
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see evp_oracle.h).  sin, cos and atan2 are the fixed algorithms of
 * cice5_amd/csrc/evpk_fmath.h (within 1-2 ulp of libm): the table indices of update_stress_rdg depend on the last bit of an
 * angle, so no two math libraries agree everywhere -- the checker and the kernels must use the same one.
 *
 * Operation order of the Fortran throughout (-ffp-contract=off); blocks are (nx, ny), i fastest, 1-based through IX().
 */
#include "evp_oracle.h"
#include "../cice5_amd/csrc/evpk_fmath.h"
#include <math.h>
#include <string.h>

#ifdef ORC_EAP_LIBM
/* second build of this file (libevp_oracle_libm.so): the host's libm instead of the fixed algorithms -- what another compiler
 * of the reference would evaluate.  Only tests/test_oracle.py::test_eap_libm_* use it, to STATE the tolerance of kdyn = 2. */
static inline void orc_libm_sincos(double x, double *s, double *c) { *s = sin(x); *c = cos(x); }
#define evpk_sincos orc_libm_sincos
#define evpk_atan2 atan2
#endif

/* counting mode (orc_eap_lookup_counts): next to every table lookup of update_stress_rdg and every branch decision of
 * calc_ffrac the same quantities are evaluated with libm's sin / cos / atan2 on the SAME inputs; counted are the lookups
 * whose index triple (kx, ky, ka) and the decisions whose outcome differ */
static int g_eap_count = 0;
static long long g_eap_n[4] = {0, 0, 0, 0};      /* lookups, lookups that differ, ffrac decisions, decisions that differ */
void orc_eap_lookup_counts(long long out[4], int enable) {
    for (int k = 0; k < 4; k++) { out[k] = g_eap_n[k]; g_eap_n[k] = 0; }
    g_eap_count = enable;
}

#define IX(i, j) ((size_t)((j)-1) * (size_t)nx + (size_t)((i)-1))

static const double c0 = 0.0, c1 = 1.0, c2 = 2.0, c3 = 3.0, p001 = 0.001, p1 = 0.1, p2 = 0.2, p25 = 0.25, p5 = 0.5;
static const double puny = 1.0e-11;
static const double pi = 3.14159265358979323846;       /* drivers/auscom/ice_constants.F90:192 */
#define PI2 (c2 * pi)
#define PIQ (p5 * (p5 * pi))                            /* pih = p5*pi, piq = p5*pih, pi2 = c2*pi (:193-195) */
static const double p111 = 1.0 / 9.0, p166 = 1.0 / 6.0, p222 = 2.0 / 9.0, p333 = 1.0 / 3.0;
#define P055 (p111 * 0.5)
#define P027 (P055 * 0.5)

/* ---------------------------------------------------------------------------
 * update_stress_rdg (:1474-1658): stresses of one corner from the lookup tables
 * tables: Fortran s11r(nx_yield, ny_yield, na_yield) = C [ka-1][ky-1][kx-1]
 * ------------------------------------------------------------------------- */
static void update_stress_rdg(int last, double divu, double tension, double shear, double a11, double a12,
                              double *stressp, double *stressm, double *stress12, double strength, double *alphar, double *alphas,
                              const orc_eap_state *e) {
    const double kfriction = 0.45;
    const int nxy = e->nx_yield, nyy = e->ny_yield, nay = e->na_yield;
    const double invstressconviso = c1 / (c1 + kfriction * kfriction);
    double sn, cs;
    evpk_sincos(PI2 / 12.0, &sn, &cs);
    const double invsin = c1 / sn * invstressconviso;
    /* 1) structure tensor */
    const double a22 = c1 - a11;
    const double gamma = p5 * evpk_atan2((c2 * a12), (a11 - a22));
    double Q11, Q12;
    evpk_sincos(gamma, &Q12, &Q11);
    const double Q11Q11 = Q11 * Q11, Q11Q12 = Q11 * Q12, Q12Q12 = Q12 * Q12;
    double atempprime = Q11Q11 * a11 + c2 * Q11Q12 * a12 + Q12Q12 * a22;
    atempprime = fmax(atempprime, c1 - atempprime);
    /* 2) strain rate */
    const double dtemp11 = p5 * (divu + tension), dtemp12 = shear * p5, dtemp22 = p5 * (divu - tension);
    double alpha = p5 * evpk_atan2((c2 * dtemp12), (dtemp11 - dtemp22));
    if (alpha > gamma) alpha = alpha - pi;
    if (alpha < gamma - pi) alpha = alpha + pi;
    const double y = gamma - alpha;
    double Qd11, Qd12;
    evpk_sincos(alpha, &Qd12, &Qd11);
    double dtemp1 = Qd11 * (Qd11 * dtemp11 + c2 * Qd12 * dtemp12) + Qd12 * Qd12 * dtemp22;
    double dtemp2 = Qd12 * (Qd12 * dtemp11 - c2 * Qd11 * dtemp12) + Qd11 * Qd11 * dtemp22;
    double x = c0;
    if (fabs(dtemp1) > puny || fabs(dtemp2) > puny) {
        const double invleng = c1 / sqrt(dtemp1 * dtemp1 + dtemp2 * dtemp2);
        dtemp1 = dtemp1 * invleng;
        dtemp2 = dtemp2 * invleng;
        x = evpk_atan2(dtemp2, dtemp1);
    }
    if (x < PIQ) x = x + PI2;
    const double dx = pi / (double)(nxy - 1), dy = pi / (double)(nyy - 1), da = p5 / (double)(nay - 1);
    const double invdx = c1 / dx, invdy = c1 / dy, invda = c1 / da;
    int kx = (int)((x - PIQ - pi) * invdx) + 1;
    int ky = (int)(y * invdy) + 1;
    int ka = (int)((atempprime - p5) * invda) + 1;
    /* (the Fortran indexes its tables unchecked; a value on the upper bound would read past them -- clamp, as the kernels do) */
    kx = kx < 1 ? 1 : (kx > nxy ? nxy : kx);
    ky = ky < 1 ? 1 : (ky > nyy ? nyy : ky);
    ka = ka < 1 ? 1 : (ka > nay ? nay : ka);
    if (g_eap_count) {
        /* the same with libm on the same inputs */
        const double g2 = p5 * atan2((c2 * a12), (a11 - a22));
        const double q11 = cos(g2), q12 = sin(g2);
        double ap = q11 * q11 * a11 + c2 * (q11 * q12) * a12 + q12 * q12 * a22;
        ap = fmax(ap, c1 - ap);
        double al = p5 * atan2((c2 * (shear * p5)), (p5 * (divu + tension) - p5 * (divu - tension)));
        if (al > g2) al = al - pi;
        if (al < g2 - pi) al = al + pi;
        const double y2 = g2 - al;
        const double d11 = cos(al), d12 = sin(al);
        const double e11 = p5 * (divu + tension), e12 = shear * p5, e22 = p5 * (divu - tension);
        double t1 = d11 * (d11 * e11 + c2 * d12 * e12) + d12 * d12 * e22;
        double t2 = d12 * (d12 * e11 - c2 * d11 * e12) + d11 * d11 * e22;
        double x2 = c0;
        if (fabs(t1) > puny || fabs(t2) > puny) {
            const double il = c1 / sqrt(t1 * t1 + t2 * t2);
            x2 = atan2(t2 * il, t1 * il);
        }
        if (x2 < PIQ) x2 = x2 + PI2;
        int kx2 = (int)((x2 - PIQ - pi) * invdx) + 1, ky2 = (int)(y2 * invdy) + 1, ka2 = (int)((ap - p5) * invda) + 1;
        kx2 = kx2 < 1 ? 1 : (kx2 > nxy ? nxy : kx2);
        ky2 = ky2 < 1 ? 1 : (ky2 > nyy ? nyy : ky2);
        ka2 = ka2 < 1 ? 1 : (ka2 > nay ? nay : ka2);
#pragma omp atomic
        g_eap_n[0]++;
        if (kx2 != kx || ky2 != ky || ka2 != ka) {
#pragma omp atomic
            g_eap_n[1]++;
        }
    }
    const size_t q = ((size_t)(ka - 1) * nyy + (ky - 1)) * nxy + (kx - 1);
    const double stemp11r = e->s11r[q], stemp12r = e->s12r[q], stemp22r = e->s22r[q];
    const double stemp11s = e->s11s[q], stemp12s = e->s12s[q], stemp22s = e->s22s[q];
    double sp = strength * (stemp11r + kfriction * stemp11s + stemp22r + kfriction * stemp22s) * invsin;
    double s12 = strength * (stemp12r + kfriction * stemp12s) * invsin;
    double sm = strength * (stemp11r + kfriction * stemp11s - stemp22r - kfriction * stemp22s) * invsin;
    const double sig11 = p5 * (sp + sm), sig12 = s12, sig22 = p5 * (sp - sm);
    const double sgprm11 = Q11Q11 * sig11 + Q12Q12 * sig22 - c2 * Q11Q12 * sig12;
    const double sgprm12 = Q11Q12 * sig11 - Q11Q12 * sig22 + (Q11Q11 - Q12Q12) * sig12;
    const double sgprm22 = Q12Q12 * sig11 + Q11Q11 * sig22 + c2 * Q11Q12 * sig12;
    *stressp = sgprm11 + sgprm22;
    *stress12 = sgprm12;
    *stressm = sgprm11 - sgprm22;
    if (last) {                                                                       /* :1628-1656 */
        const double r11 = Q11Q11 * stemp11r - c2 * Q11Q12 * stemp12r + Q12Q12 * stemp22r;
        const double r12 = Q11Q11 * stemp12r + Q11Q12 * (stemp11r - stemp22r) - Q12Q12 * stemp12r;
        const double r22 = Q12Q12 * stemp11r + c2 * Q11Q12 * stemp12r + Q11Q11 * stemp22r;
        const double s11_ = Q11Q11 * stemp11s - c2 * Q11Q12 * stemp12s + Q12Q12 * stemp22s;
        const double s12_ = Q11Q11 * stemp12s + Q11Q12 * (stemp11s - stemp22s) - Q12Q12 * stemp12s;
        const double s22_ = Q12Q12 * stemp11s + c2 * Q11Q12 * stemp12s + Q11Q11 * stemp22s;
        *alphar = r11 * dtemp11 + c2 * r12 * dtemp12 + r22 * dtemp22;
        *alphas = s11_ * dtemp11 + c2 * s12_ * dtemp12 + s22_ * dtemp22;
    }
}

/* ---------------------------------------------------------------------------
 * stress_eap (:1052-1467)
 * ------------------------------------------------------------------------- */
void orc_eap_stress(int nx, int ny, int ksub, int ndte, int icellt, const int32_t *indxti, const int32_t *indxtj, double arlx1i, double denom1,
                    const double *uvel, const double *vvel, const double *dxt, const double *dyt, const double *dxhy, const double *dyhx,
                    const double *cxp, const double *cyp, const double *cxm, const double *cym, const double *tarear, const double *strength,
                    double *const stressp[4], double *const stressm[4], double *const stress12[4], double *shear, double *divu,
                    double *prs_sig, double *rdg_conv, double *rdg_shear, double *str, const orc_eap_state *e, size_t off) {
    const size_t nn = (size_t)nx * ny;
    (void)rdg_shear;
    memset(str, 0, 8 * nn * sizeof(double));
    const int last = (ksub == ndte);
    for (int ij = 0; ij < icellt; ij++) {
        const int i = indxti[ij], j = indxtj[ij];
        const size_t k = IX(i, j), kw = IX(i - 1, j), ks = IX(i, j - 1), ksw = IX(i - 1, j - 1);
        /* strain rates * area (:1130-1160) */
        const double divune = cyp[k] * uvel[k] - dyt[k] * uvel[kw] + cxp[k] * vvel[k] - dxt[k] * vvel[ks];
        const double divunw = cym[k] * uvel[kw] + dyt[k] * uvel[k] + cxp[k] * vvel[kw] - dxt[k] * vvel[ksw];
        const double divusw = cym[k] * uvel[ksw] + dyt[k] * uvel[ks] + cxm[k] * vvel[ksw] + dxt[k] * vvel[kw];
        const double divuse = cyp[k] * uvel[ks] - dyt[k] * uvel[ksw] + cxm[k] * vvel[ks] + dxt[k] * vvel[k];
        const double tensionne = -cym[k] * uvel[k] - dyt[k] * uvel[kw] + cxm[k] * vvel[k] + dxt[k] * vvel[ks];
        const double tensionnw = -cyp[k] * uvel[kw] + dyt[k] * uvel[k] + cxm[k] * vvel[kw] + dxt[k] * vvel[ksw];
        const double tensionsw = -cyp[k] * uvel[ksw] + dyt[k] * uvel[ks] + cxp[k] * vvel[ksw] - dxt[k] * vvel[kw];
        const double tensionse = -cym[k] * uvel[ks] - dyt[k] * uvel[ksw] + cxp[k] * vvel[ks] - dxt[k] * vvel[k];
        const double shearne = -cym[k] * vvel[k] - dyt[k] * vvel[kw] - cxm[k] * uvel[k] - dxt[k] * uvel[ks];
        const double shearnw = -cyp[k] * vvel[kw] + dyt[k] * vvel[k] - cxm[k] * uvel[kw] - dxt[k] * uvel[ksw];
        const double shearsw = -cyp[k] * vvel[ksw] + dyt[k] * vvel[ks] - cxp[k] * uvel[ksw] + dxt[k] * uvel[kw];
        const double shearse = -cym[k] * vvel[ks] - dyt[k] * vvel[ksw] - cxp[k] * uvel[ks] + dxt[k] * uvel[k];
        double sptmp[4], smtmp[4], s12tmp[4], ar[4] = {0, 0, 0, 0}, as[4] = {0, 0, 0, 0};
        const size_t ko = off + k;
        update_stress_rdg(last, divune, tensionne, shearne, e->a11[0][ko], e->a12[0][ko], &sptmp[0], &smtmp[0], &s12tmp[0], strength[k], &ar[0], &as[0], e);
        update_stress_rdg(last, divunw, tensionnw, shearnw, e->a11[1][ko], e->a12[1][ko], &sptmp[1], &smtmp[1], &s12tmp[1], strength[k], &ar[1], &as[1], e);
        update_stress_rdg(last, divusw, tensionsw, shearsw, e->a11[2][ko], e->a12[2][ko], &sptmp[2], &smtmp[2], &s12tmp[2], strength[k], &ar[2], &as[2], e);
        update_stress_rdg(last, divuse, tensionse, shearse, e->a11[3][ko], e->a12[3][ko], &sptmp[3], &smtmp[3], &s12tmp[3], strength[k], &ar[3], &as[3], e);
        if (last) {                                                                   /* :1219-1234 */
            const double tt = tensionne + tensionnw + tensionse + tensionsw, ss = shearne + shearnw + shearse + shearsw;
            shear[k] = p25 * tarear[k] * sqrt(tt * tt + ss * ss);
            divu[k] = p25 * (divune + divunw + divuse + divusw) * tarear[k];
            rdg_conv[k] = -fmin(p25 * (ar[0] + ar[1] + ar[2] + ar[3]), c0) * tarear[k];
        }
        e->e11[ko] = p5 * p25 * (divune + divunw + divuse + divusw + tensionne + tensionnw + tensionse + tensionsw) * tarear[k];
        e->e12[ko] = p5 * p25 * (shearne + shearnw + shearse + shearsw) * tarear[k];
        e->e22[ko] = p5 * p25 * (divune + divunw + divuse + divusw - tensionne - tensionnw - tensionse - tensionsw) * tarear[k];
        prs_sig[k] = strength[k];
        for (int c = 0; c < 4; c++) {                                                 /* elastic relaxation (:1250-1278) */
            stressp[c][k] = (stressp[c][k] + sptmp[c] * arlx1i) * denom1;
            stressm[c][k] = (stressm[c][k] + smtmp[c] * arlx1i) * denom1;
            stress12[c][k] = (stress12[c][k] + s12tmp[c] * arlx1i) * denom1;
        }
        const double sp1 = stressp[0][k], sp2 = stressp[1][k], sp3 = stressp[2][k], sp4 = stressp[3][k];
        const double sm1 = stressm[0][k], sm2 = stressm[1][k], sm3 = stressm[2][k], sm4 = stressm[3][k];
        const double s121 = stress12[0][k], s122 = stress12[1][k], s123 = stress12[2][k], s124 = stress12[3][k];
        e->s11[ko] = p5 * p25 * (sp1 + sp2 + sp3 + sp4 + sm1 + sm2 + sm3 + sm4);
        e->s22[ko] = p5 * p25 * (sp1 + sp2 + sp3 + sp4 - sm1 - sm2 - sm3 - sm4);
        e->s12[ko] = p25 * (s121 + s122 + s123 + s124);
        e->yieldstress11[ko] = p5 * p25 * (sptmp[0] + sptmp[1] + sptmp[2] + sptmp[3] + smtmp[0] + smtmp[1] + smtmp[2] + smtmp[3]);
        e->yieldstress22[ko] = p5 * p25 * (sptmp[0] + sptmp[1] + sptmp[2] + sptmp[3] - smtmp[0] - smtmp[1] - smtmp[2] - smtmp[3]);
        e->yieldstress12[ko] = p25 * (s12tmp[0] + s12tmp[1] + s12tmp[2] + s12tmp[3]);
        /* combinations for the momentum equation (:1322-1463), as in stress of evp */
        const double ssigpn = sp1 + sp2, ssigps = sp3 + sp4, ssigpe = sp1 + sp4, ssigpw = sp2 + sp3;
        const double ssigp1 = (sp1 + sp3) * P055, ssigp2 = (sp2 + sp4) * P055;
        const double ssigmn = sm1 + sm2, ssigms = sm3 + sm4, ssigme = sm1 + sm4, ssigmw = sm2 + sm3;
        const double ssigm1 = (sm1 + sm3) * P055, ssigm2 = (sm2 + sm4) * P055;
        const double ssig12n = s121 + s122, ssig12s = s123 + s124, ssig12e = s121 + s124, ssig12w = s122 + s123;
        const double ssig121 = (s121 + s123) * p111, ssig122 = (s122 + s124) * p111;
        const double csigpne = p111 * sp1 + ssigp2 + P027 * sp3, csigpnw = p111 * sp2 + ssigp1 + P027 * sp4;
        const double csigpsw = p111 * sp3 + ssigp2 + P027 * sp1, csigpse = p111 * sp4 + ssigp1 + P027 * sp2;
        const double csigmne = p111 * sm1 + ssigm2 + P027 * sm3, csigmnw = p111 * sm2 + ssigm1 + P027 * sm4;
        const double csigmsw = p111 * sm3 + ssigm2 + P027 * sm1, csigmse = p111 * sm4 + ssigm1 + P027 * sm2;
        const double csig12ne = p222 * s121 + ssig122 + P055 * s123, csig12nw = p222 * s122 + ssig121 + P055 * s124;
        const double csig12sw = p222 * s123 + ssig122 + P055 * s121, csig12se = p222 * s124 + ssig121 + P055 * s122;
        const double str12ew = p5 * dxt[k] * (p333 * ssig12e + p166 * ssig12w), str12we = p5 * dxt[k] * (p333 * ssig12w + p166 * ssig12e);
        const double str12ns = p5 * dyt[k] * (p333 * ssig12n + p166 * ssig12s), str12sn = p5 * dyt[k] * (p333 * ssig12s + p166 * ssig12n);
        double strp_tmp = p25 * dyt[k] * (p333 * ssigpn + p166 * ssigps), strm_tmp = p25 * dyt[k] * (p333 * ssigmn + p166 * ssigms);
        str[0 * nn + k] = -strp_tmp - strm_tmp - str12ew + dxhy[k] * (-csigpne + csigmne) + dyhx[k] * csig12ne;
        str[1 * nn + k] = strp_tmp + strm_tmp - str12we + dxhy[k] * (-csigpnw + csigmnw) + dyhx[k] * csig12nw;
        strp_tmp = p25 * dyt[k] * (p333 * ssigps + p166 * ssigpn); strm_tmp = p25 * dyt[k] * (p333 * ssigms + p166 * ssigmn);
        str[2 * nn + k] = -strp_tmp - strm_tmp + str12ew + dxhy[k] * (-csigpse + csigmse) + dyhx[k] * csig12se;
        str[3 * nn + k] = strp_tmp + strm_tmp + str12we + dxhy[k] * (-csigpsw + csigmsw) + dyhx[k] * csig12sw;
        strp_tmp = p25 * dxt[k] * (p333 * ssigpe + p166 * ssigpw); strm_tmp = p25 * dxt[k] * (p333 * ssigme + p166 * ssigmw);
        str[4 * nn + k] = -strp_tmp + strm_tmp - str12ns - dyhx[k] * (csigpne + csigmne) + dxhy[k] * csig12ne;
        str[5 * nn + k] = strp_tmp - strm_tmp - str12sn - dyhx[k] * (csigpse + csigmse) + dxhy[k] * csig12se;
        strp_tmp = p25 * dxt[k] * (p333 * ssigpw + p166 * ssigpe); strm_tmp = p25 * dxt[k] * (p333 * ssigmw + p166 * ssigme);
        str[6 * nn + k] = -strp_tmp + strm_tmp + str12ns - dyhx[k] * (csigpnw + csigmnw) + dxhy[k] * csig12nw;
        str[7 * nn + k] = strp_tmp - strm_tmp + str12sn - dyhx[k] * (csigpsw + csigmsw) + dxhy[k] * csig12sw;
    }
}

/* ---------------------------------------------------------------------------
 * calc_ffrac (:1795-1864)
 * ------------------------------------------------------------------------- */
static double calc_ffrac(int blockno, double stressp, double stressm, double stress12, double a1x) {
    const double kfrac = p001, threshold = c3 * p1;
    const double sigma11 = p5 * (stressp + stressm), sigma12 = stress12, sigma22 = p5 * (stressp - stressm);
    const double gamma = p5 * evpk_atan2((c2 * sigma12), (sigma11 - sigma22));
    double Q11, Q12;
    evpk_sincos(gamma, &Q12, &Q11);
    const double Q11Q11 = Q11 * Q11, Q11Q12 = Q11 * Q12, Q12Q12 = Q12 * Q12;
    const double sigma_1 = Q11Q11 * sigma11 + c2 * Q11Q12 * sigma12 + Q12Q12 * sigma22;
    const double sigma_2 = Q12Q12 * sigma11 - c2 * Q11Q12 * sigma12 + Q11Q11 * sigma22;
    const double diffuse = blockno == 1 ? kfrac * (a1x - Q12Q12) : kfrac * (a1x + Q11Q12);
    int branch;                                   /* 0: no fracture term, 1: the diffuse term */
    if (sigma_1 >= c0 && sigma_2 >= c0) branch = 0;
    else if (sigma_1 >= c0 && sigma_2 < c0) branch = 1;
    else if (sigma_2 == c0) branch = 0;
    else if (sigma_1 <= c0 && sigma_1 / sigma_2 <= threshold) branch = 1;
    else branch = 0;
    if (g_eap_count && blockno == 1) {
        const double g2 = p5 * atan2((c2 * sigma12), (sigma11 - sigma22));
        const double q11 = cos(g2), q12 = sin(g2);
        const double s1 = q11 * q11 * sigma11 + c2 * (q11 * q12) * sigma12 + q12 * q12 * sigma22;
        const double s2 = q12 * q12 * sigma11 - c2 * (q11 * q12) * sigma12 + q11 * q11 * sigma22;
        int b2;
        if (s1 >= c0 && s2 >= c0) b2 = 0;
        else if (s1 >= c0 && s2 < c0) b2 = 1;
        else if (s2 == c0) b2 = 0;
        else if (s1 <= c0 && s1 / s2 <= threshold) b2 = 1;
        else b2 = 0;
#pragma omp atomic
        g_eap_n[2]++;
        if (b2 != branch) {
#pragma omp atomic
            g_eap_n[3]++;
        }
    }
    return branch ? diffuse : c0;
}

/* ---------------------------------------------------------------------------
 * stepa (:1664-1787)
 * ------------------------------------------------------------------------- */
void orc_eap_stepa(int nx, int ny, double dtei, int icellt, const int32_t *indxti, const int32_t *indxtj,
                   double *const stressp[4], double *const stressm[4], double *const stress12[4], const orc_eap_state *e, size_t off) {
    const double kth = p2 * p001;
    const double dteikth = c1 / (dtei + kth), p5kth = p5 * kth;
    (void)ny;
    for (int ij = 0; ij < icellt; ij++) {
        const int i = indxti[ij], j = indxtj[ij];
        const size_t k = IX(i, j), ko = off + k;
        for (int c = 0; c < 4; c++) {
            const double m11 = calc_ffrac(1, stressp[c][k], stressm[c][k], stress12[c][k], e->a11[c][ko]);
            const double m12 = calc_ffrac(2, stressp[c][k], stressm[c][k], stress12[c][k], e->a12[c][ko]);
            e->a11[c][ko] = (e->a11[c][ko] * dtei + p5kth - m11) * dteikth;
            e->a12[c][ko] = (e->a12[c][ko] * dtei - m12) * dteikth;
        }
        e->a11m[ko] = p25 * (e->a11[0][ko] + e->a11[1][ko] + e->a11[2][ko] + e->a11[3][ko]);
        e->a12m[ko] = p25 * (e->a12[0][ko] + e->a12[1][ko] + e->a12[2][ko] + e->a12[3][ko]);
    }
}

double orc_fm_sin(double x) { double s, c; evpk_sincos(x, &s, &c); return s; }
double orc_fm_cos(double x) { double s, c; evpk_sincos(x, &s, &c); return c; }
double orc_fm_atan2(double y, double x) { return evpk_atan2(y, x); }
