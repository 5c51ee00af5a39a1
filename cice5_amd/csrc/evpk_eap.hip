// evpk_eap.hip -- the elastic-anisotropic-plastic rheology (kdyn = 2, source/ice_dyn_eap.F90) on the device: SURVEY.md S8 row f-4.
// eap(dt) (:66-486) is evp(dt) with another stress: the preparation, stepu, the velocity halo and the finish are the kernels of
// evpk_kernels.hip; here are stress_eap (:1052-1467) with update_stress_rdg (:1474-1658), stepa (:1664-1787) with calc_ffrac
// (:1795-1864), and the momentum step as a kernel of its own.  Included by evpk_api.hip.
//
// Structure: one launch per subcycle (k_eap_sub: stress_eap + stepu fused, the strip march of evp's k_subcycle; EVPK_EAP_FUSED=0:
// k_eap_stress -> k_eap_stepu with str(8) through memory), then the velocity halo, and k_eap_stepa every tenth subcycle.
// What bounds stress_eap (profiles/r03_v5): not the arithmetic (2 060 VALU instructions per cell after the fma polynomials and
// the 17-entry atan table of evpk_fmath.h, 39 % VALU busy) but the table lookup of update_stress_rdg -- four gathers per cell
// over a 2.3 MB table, one 64-byte entry each since the six tables are interleaved (six cache lines each before: 67 -> 55 ms
// per eap) -- and 1.4 GB of HBM traffic per launch (45 % of peak); the history planes are stored by a call's last subcycle only.
// The extra state lives in plain planes (mask-plane indexing): a11_1..4, a12_1..4 (prognostic, restart), a11, a12, e11, e12, e22,
// yieldstress11/12/22, s11, s12, s22 (history), str(8) (work).
// sin / cos / atan2: the fixed algorithms of evpk_fmath.h, shared with the CPU checker (the table indices hang on their last bit).
#pragma once
#define EVPK_HD __host__ __device__ __forceinline__
#include "evpk_fmath.h"

namespace evpk {

constexpr int EAP_NPLANES = 8 + 11 + 8 + 8;       // a11_c, a12_c; history; str; the angles (4 double2 planes)
struct EapDev {
    // the yield-curve tables s11r, s12r, s22r, s11s, s12s, s22s interleaved: entry [na][ny][nx] = eight doubles (six values + two of
    // padding, 64 bytes) -- a lookup of update_stress_rdg is a gather over a 2.3 MB table, and one cache line per lane and
    // corner instead of six is what the L2 -> L1 path can carry; the planes of the
    // extra state, np elements each (mask-plane indexing), one after the other: a11_1..4, a12_1..4, the eleven history fields,
    // str(8), then per corner a double2 plane of angles.  One base pointer each: forty plane pointers as kernel arguments
    // would not fit the scalar registers.
    const double *tabs;
    double *pool;
    size_t nt, np;
    int nxy, nyy, nay, pad_;
    double invsin;                        // c1/sin(pi2/c12) * invstressconviso (:1524-1526), evaluated once on the host with the same evpk_sincos
    double invdx, invdy, invda;           // 1/dx, 1/dy, 1/da of the table axes (eap_set_steps)
    __host__ __device__ double *a11(int c) const { return pool + (size_t)c * np; }
    __host__ __device__ double *a12(int c) const { return pool + (size_t)(4 + c) * np; }
    // a11, a12, e11, e12, e22, yieldstress11, yieldstress12, yieldstress22, s11, s12, s22
    __host__ __device__ double *hist(int h) const { return pool + (size_t)(8 + h) * np; }
    __host__ __device__ double *str(int k) const { return pool + (size_t)(19 + k) * np; }
    // per corner {gamma, a'}: functions of (a11, a12) only, which change every tenth subcycle (stepa) -- kept instead of recomputed
    // in every stress_eap (eap_tensor_angles: an atan2 and the tensor rotation); cos gamma, sin gamma are NOT kept: one sincos per
    // corner is cheaper than the 64 B per cell they cost now that the kernel is bound by its loads, and gives the same bits
    __host__ __device__ double2 *ang(int c) const { return reinterpret_cast<double2 *>(pool + (size_t)(27 + 2 * c) * np); }
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

// the reciprocal steps of the table axes, the reference's expressions (dx = pi/(nx_yield-1), ..., invdx = c1/dx) in IEEE double
static inline void eap_set_steps(EapDev &E) {
    const double dx = EAP_PI / (double)(E.nxy - 1), dy = EAP_PI / (double)(E.nyy - 1), da = 0.5 / (double)(E.nay - 1);
    E.invdx = 1.0 / dx; E.invdy = 1.0 / dy; E.invda = 1.0 / da;
}

// ---- update_stress_rdg (:1474-1658), first part (:1528-1545): the principal axis of the structure tensor ----
__device__ __forceinline__ double2 eap_tensor_angles(double a11, double a12) {
    const double a22 = 1.0 - a11;
    const double gamma = 0.5 * evpk_atan2((2.0 * a12), (a11 - a22));
    double Q11, Q12;
    evpk_sincos(gamma, &Q12, &Q11);
    const double Q11Q11 = Q11 * Q11, Q11Q12 = Q11 * Q12, Q12Q12 = Q12 * Q12;
    double atempprime = Q11Q11 * a11 + 2.0 * Q11Q12 * a12 + Q12Q12 * a22;
    atempprime = fmax(atempprime, 1.0 - atempprime);
    return make_double2(gamma, atempprime);
}

// ---- update_stress_rdg (:1474-1658), the rest ----
template <bool LAST>
__device__ __forceinline__ void eap_update_stress_rdg(const EapDev &E, double divu, double tension, double shear, double2 ang, double strength,
                                                      double &stressp, double &stressm, double &stress12, double &alphar) {
    const double kfriction = 0.45;
    const double invsin = E.invsin;
    const double gamma = ang.x, atempprime = ang.y;
    double Q11, Q12;
    evpk_sincos(gamma, &Q12, &Q11);                                                   // :1533-1534
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
    // (:1561-1569, as straight-line code: the branch not taken computes with 1/0 at worst and is dropped by the select)
    const bool nz = fabs(dtemp1) > EAP_PUNY || fabs(dtemp2) > EAP_PUNY;
    const double invleng = 1.0 / sqrt(dtemp1 * dtemp1 + dtemp2 * dtemp2);
    dtemp1 = nz ? dtemp1 * invleng : dtemp1;
    dtemp2 = nz ? dtemp2 * invleng : dtemp2;
    const double xa = evpk_atan2(dtemp2, dtemp1);
    double x = nz ? xa : 0.0;
    if (x < EAP_PIQ) x = x + EAP_PI2;
    const double invdx = E.invdx, invdy = E.invdy, invda = E.invda;      // (:1580-1585, wave-uniform: evaluated once on the host, eap_set_steps)
    int kx = (int)((x - EAP_PIQ - EAP_PI) * invdx) + 1;
    int ky = (int)(y * invdy) + 1;
    int ka = (int)((atempprime - 0.5) * invda) + 1;
    kx = kx < 1 ? 1 : (kx > E.nxy ? E.nxy : kx);      // (the Fortran indexes unchecked; never out of the tables here)
    ky = ky < 1 ? 1 : (ky > E.nyy ? E.nyy : ky);
    ka = ka < 1 ? 1 : (ka > E.nay ? E.nay : ka);
    const size_t q = ((size_t)(ka - 1) * E.nyy + (ky - 1)) * E.nxy + (kx - 1);
    // one 64-byte entry holds the six table values of this index: one cache line per lookup instead of six
    const double2 *te = reinterpret_cast<const double2 *>(E.tabs + q * 8);
    const double2 t0 = te[0], t1 = te[1], t2 = te[2];
    const double stemp11r = t0.x, stemp12r = t0.y, stemp22r = t1.x;
    const double stemp11s = t1.y, stemp12s = t2.x, stemp22s = t2.y;
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
        const double2 iso = eap_tensor_angles(0.5, 0.0);
#pragma unroll
        for (int c = 0; c < 4; c++) { E.a11(c)[km] = 0.5; E.a12(c)[km] = 0.0; E.ang(c)[km] = iso; }
#pragma unroll
        for (int h = EH_E11; h <= EH_S22; h++) E.hist(h)[km] = 0.0;      // (active cells are rewritten by every k_eap_stress)
    }
}

// ---- stress_eap (:1052-1467) of one T cell: strain rates, update_stress_rdg at the four corners, elastic relaxation, the
// history fields and the eight stress-divergence terms (str1..8 = o.s1..s8, as stress of evp) ----
struct EapHist { double e11, e12, e22, y11, y12, y22, s11, s12, s22, shear, divu, rdgconv; };
// store_hist: the nine history planes of this cell are written (as soon as they are known: fewer live registers)
// load_sig(): the cell's stresses, fetched only when the relaxation needs them (twelve registers less across update_stress_rdg)
template <bool LAST, class LoadSig>
__device__ __forceinline__ void eap_stress_cell(const EapDev &E, size_t km, const TMet &mt, double u_ij, double u_mj, double u_im, double u_mm,
                                                double v_ij, double v_mj, double v_im, double v_mm, double tarear, double arlx1i, double denom1,
                                                LoadSig load_sig_late, Sig &g, Str8 &o, EapHist &h, bool store_hist) {
    const double p111 = 1.0 / 9.0, p055 = p111 * 0.5, p027 = p055 * 0.5, p166 = 1.0 / 6.0, p222 = 2.0 / 9.0, p333 = 1.0 / 3.0;
    const double cxp = mt.cxp, cyp = mt.cyp, cxm = mt.cxm, cym = mt.cym, dxt = mt.dxt, dyt = mt.dyt, dxhy = mt.dxhy, dyhx = mt.dyhx;
    const double strength = mt.strength;
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
    eap_update_stress_rdg<LAST>(E, divune, tensionne, shearne, E.ang(0)[km], strength, spt1, smt1, s12t1, ar1);
    eap_update_stress_rdg<LAST>(E, divunw, tensionnw, shearnw, E.ang(1)[km], strength, spt2, smt2, s12t2, ar2);
    eap_update_stress_rdg<LAST>(E, divusw, tensionsw, shearsw, E.ang(2)[km], strength, spt3, smt3, s12t3, ar3);
    eap_update_stress_rdg<LAST>(E, divuse, tensionse, shearse, E.ang(3)[km], strength, spt4, smt4, s12t4, ar4);
    h.shear = h.divu = h.rdgconv = 0.0;
    if (LAST) {                                                                       // :1219-1234
        const double tt = tensionne + tensionnw + tensionse + tensionsw, ss = shearne + shearnw + shearse + shearsw;
        h.shear = 0.25 * tarear * sqrt(tt * tt + ss * ss);
        h.divu = 0.25 * (divune + divunw + divuse + divusw) * tarear;
        h.rdgconv = -fmin(0.25 * (ar1 + ar2 + ar3 + ar4), 0.0) * tarear;
    }
    h.e11 = 0.5 * 0.25 * (divune + divunw + divuse + divusw + tensionne + tensionnw + tensionse + tensionsw) * tarear;
    h.e12 = 0.5 * 0.25 * (shearne + shearnw + shearse + shearsw) * tarear;
    h.e22 = 0.5 * 0.25 * (divune + divunw + divuse + divusw - tensionne - tensionnw - tensionse - tensionsw) * tarear;
    if (store_hist) { E.hist(EH_E11)[km] = h.e11; E.hist(EH_E12)[km] = h.e12; E.hist(EH_E22)[km] = h.e22; }
    // elastic relaxation (:1250-1278)
    g = load_sig_late();
    const double sp1 = (g.sp1 + spt1 * arlx1i) * denom1, sp2 = (g.sp2 + spt2 * arlx1i) * denom1;
    const double sp3 = (g.sp3 + spt3 * arlx1i) * denom1, sp4 = (g.sp4 + spt4 * arlx1i) * denom1;
    const double sm1 = (g.sm1 + smt1 * arlx1i) * denom1, sm2 = (g.sm2 + smt2 * arlx1i) * denom1;
    const double sm3 = (g.sm3 + smt3 * arlx1i) * denom1, sm4 = (g.sm4 + smt4 * arlx1i) * denom1;
    const double s121 = (g.s121 + s12t1 * arlx1i) * denom1, s122 = (g.s122 + s12t2 * arlx1i) * denom1;
    const double s123 = (g.s123 + s12t3 * arlx1i) * denom1, s124 = (g.s124 + s12t4 * arlx1i) * denom1;
    g = Sig{sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124};
    h.s11 = 0.5 * 0.25 * (sp1 + sp2 + sp3 + sp4 + sm1 + sm2 + sm3 + sm4);
    h.s22 = 0.5 * 0.25 * (sp1 + sp2 + sp3 + sp4 - sm1 - sm2 - sm3 - sm4);
    h.s12 = 0.25 * (s121 + s122 + s123 + s124);
    h.y11 = 0.5 * 0.25 * (spt1 + spt2 + spt3 + spt4 + smt1 + smt2 + smt3 + smt4);
    h.y22 = 0.5 * 0.25 * (spt1 + spt2 + spt3 + spt4 - smt1 - smt2 - smt3 - smt4);
    h.y12 = 0.25 * (s12t1 + s12t2 + s12t3 + s12t4);
    if (store_hist) {
        E.hist(EH_S11)[km] = h.s11; E.hist(EH_S22)[km] = h.s22; E.hist(EH_S12)[km] = h.s12;
        E.hist(EH_Y11)[km] = h.y11; E.hist(EH_Y22)[km] = h.y22; E.hist(EH_Y12)[km] = h.y12;
    }
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
    o.s1 = -strp_tmp - strm_tmp - str12ew + dxhy * (-csigpne + csigmne) + dyhx * csig12ne;
    o.s2 = strp_tmp + strm_tmp - str12we + dxhy * (-csigpnw + csigmnw) + dyhx * csig12nw;
    strp_tmp = 0.25 * dyt * (p333 * ssigps + p166 * ssigpn); strm_tmp = 0.25 * dyt * (p333 * ssigms + p166 * ssigmn);
    o.s3 = -strp_tmp - strm_tmp + str12ew + dxhy * (-csigpse + csigmse) + dyhx * csig12se;
    o.s4 = strp_tmp + strm_tmp + str12we + dxhy * (-csigpsw + csigmsw) + dyhx * csig12sw;
    strp_tmp = 0.25 * dxt * (p333 * ssigpe + p166 * ssigpw); strm_tmp = 0.25 * dxt * (p333 * ssigme + p166 * ssigmw);
    o.s5 = -strp_tmp + strm_tmp - str12ns - dyhx * (csigpne + csigmne) + dxhy * csig12ne;
    o.s6 = strp_tmp - strm_tmp - str12sn - dyhx * (csigpse + csigmse) + dxhy * csig12se;
    strp_tmp = 0.25 * dxt * (p333 * ssigpw + p166 * ssigpe); strm_tmp = 0.25 * dxt * (p333 * ssigmw + p166 * ssigme);
    o.s7 = -strp_tmp + strm_tmp + str12ns - dyhx * (csigpnw + csigmnw) + dxhy * csig12nw;
    o.s8 = strp_tmp - strm_tmp + str12sn - dyhx * (csigpsw + csigmsw) + dxhy * csig12sw;
}

// ---- stress_eap as a kernel of its own, one thread per T cell of the slab (1 .. nxl+1, 1 .. nyl+1: the N / E ghost T cells as
// the reference), str(8) through memory to k_eap_stepu: the unfused form (EVPK_EAP_FUSED=0), in place in one state buffer ----
template <bool LAST>
__global__ void __launch_bounds__(256) k_eap_stress(Slab s, EapDev E, int SB, double arlx1i, double denom1, int hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (i > s.nxl + 1 || j > s.nyl + 1) return;
    const size_t km = mcell(s, i, j);
    if (!(s.cmask[km] & CM_T)) return;                  // (str of an inactive cell is never read: k_eap_stepu tests the same mask)
    const size_t k = cell(s, i, j), kw = cell(s, i - 1, j), ks = cell(s, i, j - 1), ksw = cell(s, i - 1, j - 1);
    const TMet mt{FD(s, F_CXP, k), FD(s, F_CYP, k), FD(s, F_CXM, k), FD(s, F_CYM, k), FD(s, F_DXT, k), FD(s, F_DYT, k),
                  FD(s, F_DXHY, k), FD(s, F_DYHX, k), 0.0, FD(s, F_STRENGTH, k)};
    auto lsig = [&]() { return Sig{FD(s, SB + S_SP + 0, k), FD(s, SB + S_SP + 1, k), FD(s, SB + S_SP + 2, k), FD(s, SB + S_SP + 3, k),
                                   FD(s, SB + S_SM + 0, k), FD(s, SB + S_SM + 1, k), FD(s, SB + S_SM + 2, k), FD(s, SB + S_SM + 3, k),
                                   FD(s, SB + S_S12 + 0, k), FD(s, SB + S_S12 + 1, k), FD(s, SB + S_S12 + 2, k), FD(s, SB + S_S12 + 3, k)}; };
    Sig g;
    Str8 o;
    EapHist h;
    eap_stress_cell<LAST>(E, km, mt, FD(s, SB + S_U, k), FD(s, SB + S_U, kw), FD(s, SB + S_U, ks), FD(s, SB + S_U, ksw),
                          FD(s, SB + S_V, k), FD(s, SB + S_V, kw), FD(s, SB + S_V, ks), FD(s, SB + S_V, ksw), FD(s, F_TAREAR, k), arlx1i, denom1, lsig, g, o, h, hist != 0);
    if (LAST) { FD(s, F_SHEAR, k) = h.shear; FD(s, F_DIVU, k) = h.divu; FD(s, F_RDGCONV, k) = h.rdgconv; }
    if (hist) FD(s, F_PRSSIG, k) = mt.strength;
    FD(s, SB + S_SP + 0, k) = g.sp1; FD(s, SB + S_SP + 1, k) = g.sp2; FD(s, SB + S_SP + 2, k) = g.sp3; FD(s, SB + S_SP + 3, k) = g.sp4;
    FD(s, SB + S_SM + 0, k) = g.sm1; FD(s, SB + S_SM + 1, k) = g.sm2; FD(s, SB + S_SM + 2, k) = g.sm3; FD(s, SB + S_SM + 3, k) = g.sm4;
    FD(s, SB + S_S12 + 0, k) = g.s121; FD(s, SB + S_S12 + 1, k) = g.s122; FD(s, SB + S_S12 + 2, k) = g.s123; FD(s, SB + S_S12 + 3, k) = g.s124;
    E.str(0)[km] = o.s1; E.str(1)[km] = o.s2; E.str(2)[km] = o.s3; E.str(3)[km] = o.s4;
    E.str(4)[km] = o.s5; E.str(5)[km] = o.s6; E.str(6)[km] = o.s7; E.str(7)[km] = o.s8;
}

// ---- stepu (ice_dyn_shared.F90:623-748) from the str planes: one thread per U cell; str of a T cell without ice is 0 (:1125) ----
__global__ void __launch_bounds__(256) k_eap_stepu(Slab s, EapDev E, DevParams p, int SB) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (i > s.nxl || j > s.nyl) return;
    const size_t km = mcell(s, i, j);
    if (!(s.cmask[km] & CM_U)) return;
    const size_t k = cell(s, i, j), ke = mcell(s, i + 1, j), kn = mcell(s, i, j + 1), kne = mcell(s, i + 1, j + 1);
    const bool t0 = (s.cmask[km] & CM_T) != 0, te = (s.cmask[ke] & CM_T) != 0, tn = (s.cmask[kn] & CM_T) != 0, tne = (s.cmask[kne] & CM_T) != 0;
    const double s1 = t0 ? E.str(0)[km] : 0.0, s2 = te ? E.str(1)[ke] : 0.0, s3 = tn ? E.str(2)[kn] : 0.0, s4 = tne ? E.str(3)[kne] : 0.0;
    const double s5 = t0 ? E.str(4)[km] : 0.0, s6 = tn ? E.str(5)[kn] : 0.0, s7 = te ? E.str(6)[ke] : 0.0, s8 = tne ? E.str(7)[kne] : 0.0;
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

// ---- stepa (:1664-1787): every tenth subcycle, T cells; sig = the stresses this subcycle's stress_eap left ----
__device__ __forceinline__ void eap_stepa_cell(const EapDev &E, size_t km, const Sig &g, double dtei) {
    const double kth = 0.2 * 0.001;
    const double dteikth = 1.0 / (dtei + kth), p5kth = 0.5 * kth;
    const double sp[4] = {g.sp1, g.sp2, g.sp3, g.sp4}, sm[4] = {g.sm1, g.sm2, g.sm3, g.sm4}, s12[4] = {g.s121, g.s122, g.s123, g.s124};
    double a11n[4], a12n[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const double a11 = E.a11(c)[km], a12 = E.a12(c)[km];
        double m11, m12;
        eap_ffrac(sp[c], sm[c], s12[c], a11, a12, m11, m12);
        a11n[c] = (a11 * dtei + p5kth - m11) * dteikth;
        a12n[c] = (a12 * dtei - m12) * dteikth;
        E.a11(c)[km] = a11n[c];
        E.a12(c)[km] = a12n[c];
        E.ang(c)[km] = eap_tensor_angles(a11n[c], a12n[c]);
    }
    E.hist(EH_A11)[km] = 0.25 * (a11n[0] + a11n[1] + a11n[2] + a11n[3]);
    E.hist(EH_A12)[km] = 0.25 * (a12n[0] + a12n[1] + a12n[2] + a12n[3]);
}
__global__ void __launch_bounds__(256) k_eap_stepa(Slab s, EapDev E, int SB, double dtei) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (i > s.nxl + 1 || j > s.nyl + 1) return;
    const size_t km = mcell(s, i, j);
    if (!(s.cmask[km] & CM_T)) return;
    const size_t k = cell(s, i, j);
    const Sig g{FD(s, SB + S_SP + 0, k), FD(s, SB + S_SP + 1, k), FD(s, SB + S_SP + 2, k), FD(s, SB + S_SP + 3, k),
                FD(s, SB + S_SM + 0, k), FD(s, SB + S_SM + 1, k), FD(s, SB + S_SM + 2, k), FD(s, SB + S_SM + 3, k),
                FD(s, SB + S_S12 + 0, k), FD(s, SB + S_S12 + 1, k), FD(s, SB + S_S12 + 2, k), FD(s, SB + S_S12 + 3, k)};
    eap_stepa_cell(E, km, g, dtei);
}

// ------------------------------------------------------------------------------------
// k_eap_sub: one EAP subcycle in ONE launch -- stress_eap and stepu fused, with k_subcycle's
// structure: a wave owns a strip of 63 columns and marches north over R rows, the east cell's str terms come by a DPP wave
// shift, the row below's are carried in registers, so str(8) never goes through memory (128 B per cell of ~560) and the
// momentum step costs no launch of its own.  State double-buffered like evp's (a.sr -> a.sw): a T cell reads the velocities
// of four U cells that other waves update in the same launch.  The history fields (strain rates, yield stresses, mean
// stresses: nine planes, 72 B per cell) and prs_sig are overwritten by every subcycle in the reference; only the values of the
// last subcycle of a call can be observed, so they are stored when a.hist says so (evpk_subcycle's last subcycle).
// Measured at 3600x2700 (profiles/r03_v5): with sin / cos / atan2 and the normalisation of update_stress_rdg written as
// straight-line code (selects instead of branches: 50 -> 18 branches in the loop body, the four corners of a cell interleave)
// the kernel needs 160 VGPRs instead of 196; compiled for four waves per SIMD (128 VGPRs, 10 spilled) it runs 46.6-48.3 ms per
// eap against 48.4-51.0 at three waves and 54.8 before (same box, strips of 8 or 16 rows alike).
// stepa (every tenth subcycle) stays a launch of its own after this one: it rewrites the angles that the redundant row and
// column of the neighbouring strips read in the same launch.
// ------------------------------------------------------------------------------------
struct EapSubArgs { EapDev E; int hist; };
template <bool LAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_eap_sub(SubArgs a, EapSubArgs x) {
    const Slab &s = a.s;
    const EapDev &E = x.E;
    const int lane = threadIdx.x & 63;
    const int chunk = gridDim.x >> 3;                                 // XCD-aware order, as k_subcycle
    const int wg = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int sid = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
    if (sid >= a.nstrips) return;
    const int st = __builtin_amdgcn_readfirstlane(a.strips[sid]);
    const int cx = st % a.ncx, ry = st / a.ncx;
    const int R = a.R;
    const int i = cx * STRIP_W + 1 + lane;
    const int jb = ry * R + 1;
    const bool colT = (i <= s.nxl + 1);
    const bool ownT = colT && (lane < STRIP_W);
    const bool colU = (i <= s.nxl) && (lane < STRIP_W);
    const size_t pp = (size_t)s.pitch * 16;
    const size_t rowb = (size_t)s.rstride * 16;
    const unsigned lo = (unsigned)(C0 + i) * 16u;
    const int SR = a.sr, SW = a.sw;
    char *const base = reinterpret_cast<char *>(s.F);
    const bool revp = (a.revp == 1.0);

    double u_im = 0.0, u_mm = 0.0, v_im = 0.0, v_mm = 0.0;
    if (colT) {
        const char *rb0 = base + (size_t)(jb - 1) * rowb;
        const double2 a0 = ldp(rb0, pp, SR + S_U, lo), a1 = ldp(rb0, pp, SR + S_U, lo - 16u);
        u_im = a0.x; v_im = a0.y; u_mm = a1.x; v_mm = a1.y;
    }
    double s1c = 0.0, s5c = 0.0, s2r = 0.0, s7r = 0.0;
    unsigned char mprev = 0;
    const int pq = a.prio ? max(1, (R + 1 + 3) / 4) : 0;      // progress-based issue priority: SubArgs.prio
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    for (int jj = 0; jj <= R; jj++) {
        const int j = jb + jj;
        if (j > s.nyl + 1) break;
        if (a.prio) {
            if (jj == pq) __builtin_amdgcn_s_setprio(2);
            else if (jj == 2 * pq) __builtin_amdgcn_s_setprio(1);
            else if (jj == 3 * pq) __builtin_amdgcn_s_setprio(0);
        }
        char *const rb = base + (size_t)j * rowb;
        const size_t km = (size_t)j * s.pitch + C0 + i;
        unsigned char m = 0;
        double u_ij = 0.0, u_mj = 0.0, v_ij = 0.0, v_mj = 0.0;
        if (colT) {
            m = s.cmask[km];
            const double2 a0 = ldp(rb, pp, SR + S_U, lo), a1 = ldp(rb, pp, SR + S_U, lo - 16u);
            u_ij = a0.x; v_ij = a0.y; u_mj = a1.x; v_mj = a1.y;
        }
        const bool tact = (m & CM_T) != 0;
        Str8 o{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (__any(tact)) {
            if (tact) {
                const TMet mt = load_tmet(rb, pp, lo);
                Sig g;
                const bool store = ownT && (jj < R);
                const double tarear = *reinterpret_cast<const double *>(rb + (size_t)(F_TAREAR >> 1) * pp + lo + (F_TAREAR & 1) * 8);
                EapHist h;
                eap_stress_cell<LAST>(E, km, mt, u_ij, u_mj, u_im, u_mm, v_ij, v_mj, v_im, v_mm, tarear, a.arlx1i, a.denom1,
                                      [&]() { return load_sig(rb, pp, SR, lo); }, g, o, h, store && x.hist);
                if (store) {
                    store_sig(rb, pp, SW, lo, g);
                    if (LAST) { st1(rb, pp, F_SHEAR, lo, h.shear); st1(rb, pp, F_DIVU, lo, h.divu); st1(rb, pp, F_RDGCONV, lo, h.rdgconv); }
                    if (x.hist) st1(rb, pp, F_PRSSIG, lo, mt.strength);
                }
            }
        }
        const double s2n = shfl_dn1(o.s2), s4n = shfl_dn1(o.s4), s7n = shfl_dn1(o.s7), s8n = shfl_dn1(o.s8);
        if (jj >= 1) {
            const bool uact = colU && ((mprev & CM_U) != 0);
            if (__any(uact)) {
                if (uact) {
                    char *const ru = rb - rowb;
                    const UStat q = load_ustat(ru, pp, lo);
                    double ui = 0.0, vi = 0.0;
                    if (revp) { const double2 iv = ldp(ru, pp, F_UVEL_INIT, lo); ui = iv.x; vi = iv.y; }
                    double un, vn, strintx, strinty;
                    stepu_cell(q, u_im, v_im, ui, vi, ((s1c + s2r) + o.s3) + s4n, ((s5c + o.s6) + s7r) + s8n,
                               a.brlx, a.revp, a.cosw, a.sinw, un, vn, strintx, strinty);
                    stp(ru, pp, SW + S_U, lo, un, vn);
                    if (a.wrap) {
                        if (i == 1) stp(ru, pp, SW + S_U, lo + (unsigned)s.nxl * 16u, un, vn);
                        if (i == s.nxl) stp(ru, pp, SW + S_U, lo - (unsigned)s.nxl * 16u, un, vn);
                    }
                    if (x.hist) { st1(ru, pp, F_STRINTX, lo, strintx); st1(ru, pp, F_STRINTY, lo, strinty); }
                }
            }
        }
        s1c = o.s1; s5c = o.s5; s2r = s2n; s7r = s7n;
        u_im = u_ij; u_mm = u_mj; v_im = v_ij; v_mm = v_mj;
        mprev = m;
    }
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
    for (int c = 0; c < 4; c++) E.ang(c)[km] = eap_tensor_angles(E.a11(c)[km], E.a12(c)[km]);
}

__global__ void k_fill_mplane(Slab s, double *P, double v) {
    SLAB_IJ_ALL
    (void)k;
    P[km] = v;
}

}  // namespace evpk
