/* evpk_fmath.h -- sin / cos / atan2 as FIXED algorithms in IEEE double arithmetic (+, -, *, /, explicit fma; no contraction
 * by the compiler, no libm call beyond fma / rint / floor), so that the HIP kernels of the EAP rheology and their CPU checker produce the same bits: update_stress_rdg and calc_ffrac
 * (source/ice_dyn_eap.F90:1474-1658, :1795-1864) take table indices from angles, where a last-bit difference between two
 * math libraries moves a lookup to the neighbouring entry.  Against glibc sin and cos differ by at most 1 ulp, atan2 by at most 2
 * (tests/test_oracle.py::test_fmath_*): the same footing as the exp() of ice_strength.
 *
 * The constants are derived with exact rational arithmetic by scripts/gen_fmath.py (pi/2 split in three parts for the
 * Cody-Waite reduction, atan(k/16) as hi + lo pairs, Taylor coefficients as correctly rounded 1/k!, 1/k).
 * Compile with -ffp-contract=off: the only fused operations are the fma() calls written out below.  Included by csrc/evpk_eap.hip (device) and oracle/eap_oracle.c (host checker).
 */
#ifndef EVPK_FMATH_H
#define EVPK_FMATH_H
#include <math.h>

#ifndef EVPK_HD
#define EVPK_HD static inline
#endif

/* pi/2 = P1 + P2 + P3: P1, P2 carry 33 bits each (n * P1, n * P2 exact for |n| < 2^20) */
#define EVPK_PIO2_1 1.5707963267341256
#define EVPK_PIO2_2 6.077100506303966e-11
#define EVPK_PIO2_3 2.0222662487959506e-21
#define EVPK_2OPI 0.6366197723675814
#define EVPK_PI_HI 3.141592653589793
#define EVPK_PI_LO 1.2246467991473532e-16
#define EVPK_PIO2_HI 1.5707963267948966
#define EVPK_PIO2_LO 6.123233995736766e-17

/* sin and cos of |x| < 1e5 (the EAP angles lie in [-3 pi/2, pi/2]) */
EVPK_HD void evpk_sincos(double x, double *sn, double *cs) {
    /* sin r = r + r^3 (S0 + r^2 (S1 + ...)), cos r = 1 - (r^2/2 - r^4 (C0 + r^2 (C1 + ...))), |r| <= pi/4: Taylor to r^17, r^16,
     * Horner steps as fused multiply-adds (fma() is correctly rounded wherever it runs: the same bits on the GPU and on the host) */
    const double S0 = -0.16666666666666666, S1 = 0.008333333333333333, S2 = -0.0001984126984126984, S3 = 2.7557319223985893e-06,
                 S4 = -2.505210838544172e-08, S5 = 1.6059043836821613e-10, S6 = -7.647163731819816e-13, S7 = 2.8114572543455206e-15;
    const double C0 = 0.041666666666666664, C1 = -0.001388888888888889, C2 = 2.48015873015873e-05, C3 = -2.755731922398589e-07,
                 C4 = 2.08767569878681e-09, C5 = -1.1470745597729725e-11, C6 = 4.779477332387385e-14;
    const double fn = rint(x * EVPK_2OPI);
    const int n = (int)fn;
    const double r = fma(-fn, EVPK_PIO2_3, fma(-fn, EVPK_PIO2_2, fma(-fn, EVPK_PIO2_1, x)));
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, S7, S6), S5), S4), S3), S2), S1), S0);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1), C0);
    const double s = fma(r * z, ps, r);
    const double c = 1.0 - fma(-(z * z), pc, 0.5 * z);
    /* quadrant n & 3: (s, c), (c, -s), (-s, -c), (-c, s) -- by selects */
    const double sv = (n & 1) ? c : s, cv = (n & 1) ? s : c;
    *sn = (n & 2) ? -sv : sv;
    *cs = ((n + 1) & 2) ? -cv : cv;
}

/* atan2(y, x) for finite arguments; the signed zeros as IEEE 754 / C99 Annex F have them */
EVPK_HD double evpk_atan2(double y, double x) {
    /* atan(k/16), k = 0..16, as {hi, lo} pairs (one 16-byte load on the device: the table sits in constant memory) */
    const double tab[34] = {
        0.0, 0.0, 0.06241880999595735, -1.5490756308295046e-18, 0.12435499454676144, -3.1253241424539383e-18,
        0.18534794999569476, 4.180692268843079e-18, 0.24497866312686414, 1.0698755618734451e-17, 0.3028848683749714, -1.1010827903001369e-17,
        0.35877067027057225, -2.4623815582638635e-17, 0.4124104415973873, -1.587652227770689e-17, 0.4636476090008061, 2.2698777452961687e-17,
        0.5123894603107377, -2.5462781472855804e-17, 0.5585993153435624, -5.4556305485916264e-18, 0.6022873461349642, 2.950430737228402e-17,
        0.6435011087932844, 1.5834785051444286e-17, 0.6823165548747481, 6.943223671560008e-18, 0.7188299996216245, -2.1478388444456983e-17,
        0.7531512809621944, -2.4256934659182068e-17, 0.7853981633974483, 3.061616997868383e-17};
    /* atan u = u + u^3 (A0 + u^2 (A1 + ...)), 0 <= u < 1/16: Taylor to u^13 (the next term is below 2^-59 u) */
    const double A0 = -0.3333333333333333, A1 = 0.2, A2 = -0.14285714285714285, A3 = 0.1111111111111111, A4 = -0.09090909090909091,
                 A5 = 0.07692307692307693;
    const double ax = fabs(x), ay = fabs(y);
    /* straight-line code (selects, no branches): the four corners of a cell are four independent chains the compiler can
     * interleave.  ay = 0 (also x = y = 0) takes a = 0 at the end; what the main path computes then (0/0 at worst) is dropped. */
    const int swap = ay > ax;
    const double t = (swap ? ax : ay) / (swap ? ay : ax);       /* min / max, in [0, 1] */
    const double fk0 = floor(t * 16.0);               /* the breakpoint below t: u >= 0, nothing cancels in hi + (atan u + lo) */
    const double fk = (fk0 >= 0.0 && fk0 <= 16.0) ? fk0 : 0.0;      /* (t is NaN only on the dropped path: keep the index in the table) */
    const int k = (int)fk;
    const double cpt = fk * 0.0625;
    const double u = (t - cpt) / fma(t, cpt, 1.0);
    const double z = u * u;
    const double p = fma(z, fma(z, fma(z, fma(z, fma(z, A5, A4), A3), A2), A1), A0);
    double a = tab[2 * k] + (fma(u * z, p, u) + tab[2 * k + 1]);
    a = swap ? EVPK_PIO2_HI - (a - EVPK_PIO2_LO) : a;
    a = (ay == 0.0) ? 0.0 : a;
    const double ar = EVPK_PI_HI - (a - EVPK_PI_LO);
    const int xneg = (signbit(x) != 0) & ((ay == 0.0) | (x < 0.0));      /* ay = 0: the sign bit of x (-0 too); else x < 0 */
    a = xneg ? ar : a;
    return copysign(a, y);
}

#endif
