// evpk_eap.hip -- the elastic-anisotropic-plastic rheology (kdyn = 2, source/ice_dyn_eap.F90) on the device: SURVEY.md S8 row f-4.
// eap(dt) (:66-486) is evp(dt) with another stress: the preparation, stepu, the velocity halo and the finish are the kernels of
// evpk_kernels.hip; here are stress_eap (:1052-1467) with update_stress_rdg (:1474-1658), stepa (:1664-1787) with calc_ffrac
// (:1795-1864), and the momentum step as a kernel of its own.  Included by evpk_api.hip.
//
// Structure: one launch per stage and subcycle (k_eap_stress -> k_eap_stepu -> halo [-> k_eap_stepa every tenth subcycle]), the
// state updated in place in the current state buffer.  stress_eap is arithmetic-bound (per corner three atan2, two sincos, a
// square root and a table lookup: about 2 000 fp64 operations per cell against 500 B of traffic), so the temporal blocking
// that the EVP kernels need for HBM would buy little here; the str(8) hand-over between the two kernels costs 128 B per cell.
// The extra state lives in plain planes (mask-plane indexing): a11_1..4, a12_1..4 (prognostic, restart), a11, a12, e11, e12, e22,
// yieldstress11/12/22, s11, s12, s22 (history), str(8) (work).
// sin / cos / atan2: the fixed algorithms of evpk_fmath.h, shared with the CPU checker (the table indices hang on their last bit).
#pragma once
#define EVPK_HD __host__ __device__ __forceinline__
#include "evpk_fmath.h"

namespace evpk {

constexpr int EAP_NPLANES = 8 + 11 + 8 + 16;      // a11_c, a12_c; history; str; the angles (4 double4 planes)
struct EapDev {
    const double *tab[6];                 // s11r, s12r, s22r, s11s, s12s, s22s: [na][ny][nx]
    int nxy, nyy, nay, pad_;
    double invsin;                        // c1/sin(pi2/c12) * invstressconviso (:1524-1526), evaluated once on the host with the same evpk_sincos
    double *a11[4], *a12[4];
    double4 *ang[4];                      // per corner {gamma, cos gamma, sin gamma, a'}: functions of (a11, a12) only, which change
                                          // every tenth subcycle (stepa) -- kept instead of recomputed in every stress_eap (eap_tensor_angles)
    double *hist[11];                     // a11, a12, e11, e12, e22, yieldstress11, yieldstress12, yieldstress22, s11, s12, s22
    double *str[8];
};
enum { EH_A11 = 0, EH_A12, EH_E11, EH_E12, EH_E22, EH_Y11, EH_Y12, EH_Y22, EH_S11, EH_S12, EH_S22 };

#define EAP_PI 3.14159265358979323846
#define EAP_PI2 (2.0 * EAP_PI)
#define EAP_PIQ (0.5 * (0.5 * EAP_PI))
#define EAP_PUNY 1.0e-11

// c1/sin(pi2/c12) * invstressconviso, invstressconviso = c1/(c1 + kfriction*kfriction) (:1521-1526)
static inline double eap_invsin() {
    const double kfriction = 0.45;
    const double invstressconviso = 1.0 / (1.0 + kfriction * kfriction);
    double sn, cs;
    evpk_sincos(EAP_PI2 / 12.0, &sn, &cs);
    return 1.0 / sn * invstressconviso;
}

// ---- update_stress_rdg (:1474-1658), first part (:1528-1545): the principal axis of the structure tensor ----
__device__ __forceinline__ double4 eap_tensor_angles(double a11, double a12) {
    const double a22 = 1.0 - a11;
    const double gamma = 0.5 * evpk_atan2((2.0 * a12), (a11 - a22));
    double Q11, Q12;
    evpk_sincos(gamma, &Q12, &Q11);
    const double Q11Q11 = Q11 * Q11, Q11Q12 = Q11 * Q12, Q12Q12 = Q12 * Q12;
    double atempprime = Q11Q11 * a11 + 2.0 * Q11Q12 * a12 + Q12Q12 * a22;
    atempprime = fmax(atempprime, 1.0 - atempprime);
    return make_double4(gamma, Q11, Q12, atempprime);
}

// ---- update_stress_rdg (:1474-1658), the rest ----
template <bool LAST>
__device__ __forceinline__ void eap_update_stress_rdg(const EapDev &E, double divu, double tension, double shear, double4 ang, double strength,
                                                      double &stressp, double &stressm, double &stress12, double &alphar) {
    const double kfriction = 0.45;
    const double invsin = E.invsin;
    const double gamma = ang.x, Q11 = ang.y, Q12 = ang.z, atempprime = ang.w;
    const double Q11Q11 = Q11 * Q11, Q11Q12 = Q11 * Q12, Q12Q12 = Q12 * Q12;
    const double dtemp11 = 0.5 * (divu + tension), dtemp12 = shear * 0.5, dtemp22 = 0.5 * (divu - tension);
    double alpha = 0.5 * evpk_atan2((2.0 * dtemp12), (dtemp11 - dtemp22));
    if (alpha > gamma) alpha = alpha - EAP_PI;
    if (alpha < gamma - EAP_PI) alpha = alpha + EAP_PI;
    const double y = gamma - alpha;
    double Qd11, Qd12;
    evpk_sincos(alpha, &Qd12, &Qd11);
    double dtemp1 = Qd11 * (Qd11 * dtemp11 + 2.0 * Qd12 * dtemp12) + Qd12 * Qd12 * dtemp22;
    double dtemp2 = Qd12 * (Qd12 * dtemp11 - 2.0 * Qd11 * dtemp12) + Qd11 * Qd11 * dtemp22;
    double x = 0.0;
    if (fabs(dtemp1) > EAP_PUNY || fabs(dtemp2) > EAP_PUNY) {
        const double invleng = 1.0 / sqrt(dtemp1 * dtemp1 + dtemp2 * dtemp2);
        dtemp1 = dtemp1 * invleng;
        dtemp2 = dtemp2 * invleng;
        x = evpk_atan2(dtemp2, dtemp1);
    }
    if (x < EAP_PIQ) x = x + EAP_PI2;
    const double dx = EAP_PI / (double)(E.nxy - 1), dy = EAP_PI / (double)(E.nyy - 1), da = 0.5 / (double)(E.nay - 1);
    const double invdx = 1.0 / dx, invdy = 1.0 / dy, invda = 1.0 / da;
    int kx = (int)((x - EAP_PIQ - EAP_PI) * invdx) + 1;
    int ky = (int)(y * invdy) + 1;
    int ka = (int)((atempprime - 0.5) * invda) + 1;
    kx = kx < 1 ? 1 : (kx > E.nxy ? E.nxy : kx);      // (the Fortran indexes unchecked; never out of the tables here)
    ky = ky < 1 ? 1 : (ky > E.nyy ? E.nyy : ky);
    ka = ka < 1 ? 1 : (ka > E.nay ? E.nay : ka);
    const size_t q = ((size_t)(ka - 1) * E.nyy + (ky - 1)) * E.nxy + (kx - 1);
    const double stemp11r = E.tab[0][q], stemp12r = E.tab[1][q], stemp22r = E.tab[2][q];
    const double stemp11s = E.tab[3][q], stemp12s = E.tab[4][q], stemp22s = E.tab[5][q];
    const double sp = strength * (stemp11r + kfriction * stemp11s + stemp22r + kfriction * stemp22s) * invsin;
    const double s12 = strength * (stemp12r + kfriction * stemp12s) * invsin;
    const double sm = strength * (stemp11r + kfriction * stemp11s - stemp22r - kfriction * stemp22s) * invsin;
    const double sig11 = 0.5 * (sp + sm), sig12 = s12, sig22 = 0.5 * (sp - sm);
    const double sgprm11 = Q11Q11 * sig11 + Q12Q12 * sig22 - 2.0 * Q11Q12 * sig12;
    const double sgprm12 = Q11Q12 * sig11 - Q11Q12 * sig22 + (Q11Q11 - Q12Q12) * sig12;
    const double sgprm22 = Q12Q12 * sig11 + Q11Q11 * sig22 + 2.0 * Q11Q12 * sig12;
    stressp = sgprm11 + sgprm22;
    stress12 = sgprm12;
    stressm = sgprm11 - sgprm22;
    if (LAST) {                                                                       // :1628-1656 (alphas feeds the rdg_shear the reference leaves commented out)
        const double r11 = Q11Q11 * stemp11r - 2.0 * Q11Q12 * stemp12r + Q12Q12 * stemp22r;
        const double r12 = Q11Q11 * stemp12r + Q11Q12 * (stemp11r - stemp22r) - Q12Q12 * stemp12r;
        const double r22 = Q12Q12 * stemp11r + 2.0 * Q11Q12 * stemp12r + Q11Q11 * stemp22r;
        alphar = r11 * dtemp11 + 2.0 * r12 * dtemp12 + r22 * dtemp22;
    }
}

// ---- start of eap(dt): history fields zero (:171-180); structure tensor isotropic where icetmask = 0 (:284-298) ----
__global__ void k_eap_reset(Slab s, EapDev E) {
    SLAB_IJ_ALL
    (void)k;
    const bool tact = (s.cmask[km] & CM_T) != 0;
    if (!tact) {
        const double4 iso = eap_tensor_angles(0.5, 0.0);
#pragma unroll
        for (int c = 0; c < 4; c++) { E.a11[c][km] = 0.5; E.a12[c][km] = 0.0; E.ang[c][km] = iso; }
#pragma unroll
        for (int h = EH_E11; h <= EH_S22; h++) E.hist[h][km] = 0.0;      // (active cells are rewritten by every k_eap_stress)
    }
}

// ---- stress_eap (:1052-1467): one thread per T cell of the slab (1 .. nxl+1, 1 .. nyl+1: the N / E ghost T cells as the reference) ----
template <bool LAST>
__global__ void __launch_bounds__(256) k_eap_stress(Slab s, EapDev E, int SB, double arlx1i, double denom1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (i > s.nxl + 1 || j > s.nyl + 1) return;
    const size_t km = mcell(s, i, j);
    if (!(s.cmask[km] & CM_T)) return;                  // (str of an inactive cell is never read: k_eap_stepu tests the same mask)
    const size_t k = cell(s, i, j), kw = cell(s, i - 1, j), ks = cell(s, i, j - 1), ksw = cell(s, i - 1, j - 1);
    const double p111 = 1.0 / 9.0, p055 = p111 * 0.5, p027 = p055 * 0.5, p166 = 1.0 / 6.0, p222 = 2.0 / 9.0, p333 = 1.0 / 3.0;
    const double u_ij = FD(s, SB + S_U, k), u_mj = FD(s, SB + S_U, kw), u_im = FD(s, SB + S_U, ks), u_mm = FD(s, SB + S_U, ksw);
    const double v_ij = FD(s, SB + S_V, k), v_mj = FD(s, SB + S_V, kw), v_im = FD(s, SB + S_V, ks), v_mm = FD(s, SB + S_V, ksw);
    const double cxp = FD(s, F_CXP, k), cyp = FD(s, F_CYP, k), cxm = FD(s, F_CXM, k), cym = FD(s, F_CYM, k);
    const double dxt = FD(s, F_DXT, k), dyt = FD(s, F_DYT, k), dxhy = FD(s, F_DXHY, k), dyhx = FD(s, F_DYHX, k);
    const double tarear = FD(s, F_TAREAR, k), strength = FD(s, F_STRENGTH, k);
    // strain rates * area (:1130-1160)
    const double divune = cyp * u_ij - dyt * u_mj + cxp * v_ij - dxt * v_im;
    const double divunw = cym * u_mj + dyt * u_ij + cxp * v_mj - dxt * v_mm;
    const double divusw = cym * u_mm + dyt * u_im + cxm * v_mm + dxt * v_mj;
    const double divuse = cyp * u_im - dyt * u_mm + cxm * v_im + dxt * v_ij;
    const double tensionne = -cym * u_ij - dyt * u_mj + cxm * v_ij + dxt * v_im;
    const double tensionnw = -cyp * u_mj + dyt * u_ij + cxm * v_mj + dxt * v_mm;
    const double tensionsw = -cyp * u_mm + dyt * u_im + cxp * v_mm - dxt * v_mj;
    const double tensionse = -cym * u_im - dyt * u_mm + cxp * v_im - dxt * v_ij;
    const double shearne = -cym * v_ij - dyt * v_mj - cxm * u_ij - dxt * u_im;
    const double shearnw = -cyp * v_mj + dyt * v_ij - cxm * u_mj - dxt * u_mm;
    const double shearsw = -cyp * v_mm + dyt * v_im - cxp * u_mm + dxt * u_mj;
    const double shearse = -cym * v_im - dyt * v_mm - cxp * u_im + dxt * u_ij;
    double spt1, spt2, spt3, spt4, smt1, smt2, smt3, smt4, s12t1, s12t2, s12t3, s12t4, ar1 = 0.0, ar2 = 0.0, ar3 = 0.0, ar4 = 0.0;
    eap_update_stress_rdg<LAST>(E, divune, tensionne, shearne, E.ang[0][km], strength, spt1, smt1, s12t1, ar1);
    eap_update_stress_rdg<LAST>(E, divunw, tensionnw, shearnw, E.ang[1][km], strength, spt2, smt2, s12t2, ar2);
    eap_update_stress_rdg<LAST>(E, divusw, tensionsw, shearsw, E.ang[2][km], strength, spt3, smt3, s12t3, ar3);
    eap_update_stress_rdg<LAST>(E, divuse, tensionse, shearse, E.ang[3][km], strength, spt4, smt4, s12t4, ar4);
    if (LAST) {                                                                       // :1219-1234
        const double tt = tensionne + tensionnw + tensionse + tensionsw, ss = shearne + shearnw + shearse + shearsw;
        FD(s, F_SHEAR, k) = 0.25 * tarear * sqrt(tt * tt + ss * ss);
        FD(s, F_DIVU, k) = 0.25 * (divune + divunw + divuse + divusw) * tarear;
        FD(s, F_RDGCONV, k) = -fmin(0.25 * (ar1 + ar2 + ar3 + ar4), 0.0) * tarear;
    }
    E.hist[EH_E11][km] = 0.5 * 0.25 * (divune + divunw + divuse + divusw + tensionne + tensionnw + tensionse + tensionsw) * tarear;
    E.hist[EH_E12][km] = 0.5 * 0.25 * (shearne + shearnw + shearse + shearsw) * tarear;
    E.hist[EH_E22][km] = 0.5 * 0.25 * (divune + divunw + divuse + divusw - tensionne - tensionnw - tensionse - tensionsw) * tarear;
    FD(s, F_PRSSIG, k) = strength;
    // elastic relaxation (:1250-1278)
    const double sp1 = (FD(s, SB + S_SP + 0, k) + spt1 * arlx1i) * denom1, sp2 = (FD(s, SB + S_SP + 1, k) + spt2 * arlx1i) * denom1;
    const double sp3 = (FD(s, SB + S_SP + 2, k) + spt3 * arlx1i) * denom1, sp4 = (FD(s, SB + S_SP + 3, k) + spt4 * arlx1i) * denom1;
    const double sm1 = (FD(s, SB + S_SM + 0, k) + smt1 * arlx1i) * denom1, sm2 = (FD(s, SB + S_SM + 1, k) + smt2 * arlx1i) * denom1;
    const double sm3 = (FD(s, SB + S_SM + 2, k) + smt3 * arlx1i) * denom1, sm4 = (FD(s, SB + S_SM + 3, k) + smt4 * arlx1i) * denom1;
    const double s121 = (FD(s, SB + S_S12 + 0, k) + s12t1 * arlx1i) * denom1, s122 = (FD(s, SB + S_S12 + 1, k) + s12t2 * arlx1i) * denom1;
    const double s123 = (FD(s, SB + S_S12 + 2, k) + s12t3 * arlx1i) * denom1, s124 = (FD(s, SB + S_S12 + 3, k) + s12t4 * arlx1i) * denom1;
    FD(s, SB + S_SP + 0, k) = sp1; FD(s, SB + S_SP + 1, k) = sp2; FD(s, SB + S_SP + 2, k) = sp3; FD(s, SB + S_SP + 3, k) = sp4;
    FD(s, SB + S_SM + 0, k) = sm1; FD(s, SB + S_SM + 1, k) = sm2; FD(s, SB + S_SM + 2, k) = sm3; FD(s, SB + S_SM + 3, k) = sm4;
    FD(s, SB + S_S12 + 0, k) = s121; FD(s, SB + S_S12 + 1, k) = s122; FD(s, SB + S_S12 + 2, k) = s123; FD(s, SB + S_S12 + 3, k) = s124;
    E.hist[EH_S11][km] = 0.5 * 0.25 * (sp1 + sp2 + sp3 + sp4 + sm1 + sm2 + sm3 + sm4);
    E.hist[EH_S22][km] = 0.5 * 0.25 * (sp1 + sp2 + sp3 + sp4 - sm1 - sm2 - sm3 - sm4);
    E.hist[EH_S12][km] = 0.25 * (s121 + s122 + s123 + s124);
    E.hist[EH_Y11][km] = 0.5 * 0.25 * (spt1 + spt2 + spt3 + spt4 + smt1 + smt2 + smt3 + smt4);
    E.hist[EH_Y22][km] = 0.5 * 0.25 * (spt1 + spt2 + spt3 + spt4 - smt1 - smt2 - smt3 - smt4);
    E.hist[EH_Y12][km] = 0.25 * (s12t1 + s12t2 + s12t3 + s12t4);
    // combinations for the momentum equation (:1322-1463), as in stress of evp
    const double ssigpn = sp1 + sp2, ssigps = sp3 + sp4, ssigpe = sp1 + sp4, ssigpw = sp2 + sp3;
    const double ssigp1 = (sp1 + sp3) * p055, ssigp2 = (sp2 + sp4) * p055;
    const double ssigmn = sm1 + sm2, ssigms = sm3 + sm4, ssigme = sm1 + sm4, ssigmw = sm2 + sm3;
    const double ssigm1 = (sm1 + sm3) * p055, ssigm2 = (sm2 + sm4) * p055;
    const double ssig12n = s121 + s122, ssig12s = s123 + s124, ssig12e = s121 + s124, ssig12w = s122 + s123;
    const double ssig121 = (s121 + s123) * p111, ssig122 = (s122 + s124) * p111;
    const double csigpne = p111 * sp1 + ssigp2 + p027 * sp3, csigpnw = p111 * sp2 + ssigp1 + p027 * sp4;
    const double csigpsw = p111 * sp3 + ssigp2 + p027 * sp1, csigpse = p111 * sp4 + ssigp1 + p027 * sp2;
    const double csigmne = p111 * sm1 + ssigm2 + p027 * sm3, csigmnw = p111 * sm2 + ssigm1 + p027 * sm4;
    const double csigmsw = p111 * sm3 + ssigm2 + p027 * sm1, csigmse = p111 * sm4 + ssigm1 + p027 * sm2;
    const double csig12ne = p222 * s121 + ssig122 + p055 * s123, csig12nw = p222 * s122 + ssig121 + p055 * s124;
    const double csig12sw = p222 * s123 + ssig122 + p055 * s121, csig12se = p222 * s124 + ssig121 + p055 * s122;
    const double str12ew = 0.5 * dxt * (p333 * ssig12e + p166 * ssig12w), str12we = 0.5 * dxt * (p333 * ssig12w + p166 * ssig12e);
    const double str12ns = 0.5 * dyt * (p333 * ssig12n + p166 * ssig12s), str12sn = 0.5 * dyt * (p333 * ssig12s + p166 * ssig12n);
    double strp_tmp = 0.25 * dyt * (p333 * ssigpn + p166 * ssigps), strm_tmp = 0.25 * dyt * (p333 * ssigmn + p166 * ssigms);
    E.str[0][km] = -strp_tmp - strm_tmp - str12ew + dxhy * (-csigpne + csigmne) + dyhx * csig12ne;
    E.str[1][km] = strp_tmp + strm_tmp - str12we + dxhy * (-csigpnw + csigmnw) + dyhx * csig12nw;
    strp_tmp = 0.25 * dyt * (p333 * ssigps + p166 * ssigpn); strm_tmp = 0.25 * dyt * (p333 * ssigms + p166 * ssigmn);
    E.str[2][km] = -strp_tmp - strm_tmp + str12ew + dxhy * (-csigpse + csigmse) + dyhx * csig12se;
    E.str[3][km] = strp_tmp + strm_tmp + str12we + dxhy * (-csigpsw + csigmsw) + dyhx * csig12sw;
    strp_tmp = 0.25 * dxt * (p333 * ssigpe + p166 * ssigpw); strm_tmp = 0.25 * dxt * (p333 * ssigme + p166 * ssigmw);
    E.str[4][km] = -strp_tmp + strm_tmp - str12ns - dyhx * (csigpne + csigmne) + dxhy * csig12ne;
    E.str[5][km] = strp_tmp - strm_tmp - str12sn - dyhx * (csigpse + csigmse) + dxhy * csig12se;
    strp_tmp = 0.25 * dxt * (p333 * ssigpw + p166 * ssigpe); strm_tmp = 0.25 * dxt * (p333 * ssigmw + p166 * ssigme);
    E.str[6][km] = -strp_tmp + strm_tmp + str12ns - dyhx * (csigpnw + csigmnw) + dxhy * csig12nw;
    E.str[7][km] = strp_tmp - strm_tmp + str12sn - dyhx * (csigpsw + csigmsw) + dxhy * csig12sw;
}

// ---- stepu (ice_dyn_shared.F90:623-748) from the str planes: one thread per U cell; str of a T cell without ice is 0 (:1125) ----
__global__ void __launch_bounds__(256) k_eap_stepu(Slab s, EapDev E, DevParams p, int SB) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (i > s.nxl || j > s.nyl) return;
    const size_t km = mcell(s, i, j);
    if (!(s.cmask[km] & CM_U)) return;
    const size_t k = cell(s, i, j), ke = mcell(s, i + 1, j), kn = mcell(s, i, j + 1), kne = mcell(s, i + 1, j + 1);
    const bool t0 = (s.cmask[km] & CM_T) != 0, te = (s.cmask[ke] & CM_T) != 0, tn = (s.cmask[kn] & CM_T) != 0, tne = (s.cmask[kne] & CM_T) != 0;
    const double s1 = t0 ? E.str[0][km] : 0.0, s2 = te ? E.str[1][ke] : 0.0, s3 = tn ? E.str[2][kn] : 0.0, s4 = tne ? E.str[3][kne] : 0.0;
    const double s5 = t0 ? E.str[4][km] : 0.0, s6 = tn ? E.str[5][kn] : 0.0, s7 = te ? E.str[6][ke] : 0.0, s8 = tne ? E.str[7][kne] : 0.0;
    const UStat q{FD(s, F_VRELC, k), FD(s, F_UAREAR, k), FD(s, F_UOCN, k), FD(s, F_VOCN, k),
                  FD(s, F_FORCEX, k), FD(s, F_FORCEY, k), FD(s, F_UMASSDTI, k), FD(s, F_FM, k)};
    const double uold = FD(s, SB + S_U, k), vold = FD(s, SB + S_V, k);
    double ui = 0.0, vi = 0.0;
    if (p.revp == 1.0) { ui = FD(s, F_UVEL_INIT, k); vi = FD(s, F_VVEL_INIT, k); }
    double un, vn, strintx, strinty;
    stepu_cell(q, uold, vold, ui, vi, ((s1 + s2) + s3) + s4, ((s5 + s6) + s7) + s8, p.brlx, p.revp, p.cosw, p.sinw, un, vn, strintx, strinty);
    FD(s, SB + S_U, k) = un;
    FD(s, SB + S_V, k) = vn;
    FD(s, F_STRINTX, k) = strintx;
    FD(s, F_STRINTY, k) = strinty;
}

// ---- calc_ffrac (:1795-1864) ----
__device__ __forceinline__ void eap_ffrac(double stressp, double stressm, double stress12, double a11, double a12, double &m11, double &m12) {
    const double kfrac = 0.001, threshold = 3.0 * 0.1;
    const double sigma11 = 0.5 * (stressp + stressm), sigma12 = stress12, sigma22 = 0.5 * (stressp - stressm);
    const double gamma = 0.5 * evpk_atan2((2.0 * sigma12), (sigma11 - sigma22));
    double Q11, Q12;
    evpk_sincos(gamma, &Q12, &Q11);
    const double Q11Q11 = Q11 * Q11, Q11Q12 = Q11 * Q12, Q12Q12 = Q12 * Q12;
    const double sigma_1 = Q11Q11 * sigma11 + 2.0 * Q11Q12 * sigma12 + Q12Q12 * sigma22;
    const double sigma_2 = Q12Q12 * sigma11 - 2.0 * Q11Q12 * sigma12 + Q11Q11 * sigma22;
    bool diffuse;
    if (sigma_1 >= 0.0 && sigma_2 >= 0.0) diffuse = false;
    else if (sigma_1 >= 0.0 && sigma_2 < 0.0) diffuse = true;
    else if (sigma_2 == 0.0) diffuse = false;
    else diffuse = (sigma_1 <= 0.0 && sigma_1 / sigma_2 <= threshold);
    m11 = diffuse ? kfrac * (a11 - Q12Q12) : 0.0;
    m12 = diffuse ? kfrac * (a12 + Q11Q12) : 0.0;
}

// ---- stepa (:1664-1787): every tenth subcycle, T cells ----
__global__ void __launch_bounds__(256) k_eap_stepa(Slab s, EapDev E, int SB, double dtei) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (i > s.nxl + 1 || j > s.nyl + 1) return;
    const size_t km = mcell(s, i, j);
    if (!(s.cmask[km] & CM_T)) return;
    const size_t k = cell(s, i, j);
    const double kth = 0.2 * 0.001;
    const double dteikth = 1.0 / (dtei + kth), p5kth = 0.5 * kth;
    double a11n[4], a12n[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const double a11 = E.a11[c][km], a12 = E.a12[c][km];
        double m11, m12;
        eap_ffrac(FD(s, SB + S_SP + c, k), FD(s, SB + S_SM + c, k), FD(s, SB + S_S12 + c, k), a11, a12, m11, m12);
        a11n[c] = (a11 * dtei + p5kth - m11) * dteikth;
        a12n[c] = (a12 * dtei - m12) * dteikth;
        E.a11[c][km] = a11n[c];
        E.a12[c][km] = a12n[c];
        E.ang[c][km] = eap_tensor_angles(a11n[c], a12n[c]);
    }
    E.hist[EH_A11][km] = 0.25 * (a11n[0] + a11n[1] + a11n[2] + a11n[3]);
    E.hist[EH_A12][km] = 0.25 * (a12n[0] + a12n[1] + a12n[2] + a12n[3]);
}

// a plain plane -> one block array, cells chosen by `mode` (MODE_NE: the physical cells and the N / E ghost T cells)
__global__ void k_scatter_mplane(Slab s, const BlockDesc *bd, int nxb, int nyb, const double *src, double *dst, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    int si, sj;
    if (!scatter_take(s, bd[b], i, j, mode, si, sj)) return;
    dst[((size_t)b * nyb + (j - 1)) * nxb + (i - 1)] = src[mcell(s, si, sj)];
}

// the angles of every cell from its (a11, a12): after init_eap's fill and after an upload of the structure tensor
__global__ void k_eap_angles(Slab s, EapDev E) {
    SLAB_IJ_ALL
    (void)k;
#pragma unroll
    for (int c = 0; c < 4; c++) E.ang[c][km] = eap_tensor_angles(E.a11[c][km], E.a12[c][km]);
}

__global__ void k_fill_mplane(Slab s, double *P, double v) {
    SLAB_IJ_ALL
    (void)k;
    P[km] = v;
}

}  // namespace evpk
