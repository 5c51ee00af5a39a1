// evpk_experimental.hip -- kernels that were built, are bit-exact, were measured and were NOT adopted.  Compiled only with
// -DEVPK_EXPERIMENTAL (make -C cice5_amd/csrc exp -> libevpk_exp.so); the product library does not contain them and refuses the
// environment switches that select them (EVPK_PREFETCH=0, EVPK_TRIPLE=1).  Their measurements: docs/NOTEBOOK.md S4,
// profiles/r04_v1/k3_rejected.txt.  Included by evpk_api.hip after evpk_kernels.hip (same translation unit, same helpers).
#include "evpk_internal.h"

namespace evpk {

// ------------------------------------------------------------------------------------
// k_subcycle2: the two-subcycle march WITHOUT the LDS prefetch (round 1; superseded by k_subcycle2p, EVPK_PREFETCH=0 selects it)
// ------------------------------------------------------------------------------------
template <bool REVP, bool LAST2>
__global__ __launch_bounds__(256) void k_subcycle2(SubArgs a) {
    const Slab &s = a.s;
    const int lane = threadIdx.x & 63;
    const int ns = pair_nstrips(a);
    const int chunk = a.nsdev ? (((ns + 3) >> 2) + 7) >> 3 : (int)(gridDim.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk) return;
    const int wg = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int sid = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
    if (sid >= ns) return;
    const int st = __builtin_amdgcn_readfirstlane(a.strips[sid]);
    const int cx = st % a.ncx, ry = st / a.ncx;
    const int R = a.R, nxl = s.nxl, nyl = s.nyl;
    const int G = a.G;
    const int c = cx * STRIP2_W + lane - G;           // unwrapped column of this lane (lane 1 = first owned column)
    const int jb = ry * R + 1;
    const bool cyc = a.wrap != 0;

    // storage columns of (c) and (c-1); outside a non-cyclic domain the lane is dead for that access
    int ci = c, cm1 = c - 1;
    bool okc, okm;
    if (cyc) {
        ci = (c - 1) % nxl; if (ci < 0) ci += nxl; ci += 1;
        cm1 = (c - 2) % nxl; if (cm1 < 0) cm1 += nxl; cm1 += 1;
        okc = okm = true;
    } else {
        // ghost-zone mode: storage has G+2 ghost columns per side (-1-G .. 0 | nxl+1 .. nxl+2+G); on an open / closed
        // boundary they hold zeros and inactive masks, between ranks the neighbour's columns.  The zone loses two
        // valid columns per launch, so it lasts (G+2)/2 launches between exchanges.
        okc = (c >= -1 - G && c <= nxl + 2 + G);
        okm = (cm1 >= -1 - G && cm1 <= nxl + 2 + G);
        if (!okc) ci = 0;
        if (!okm) cm1 = 0;
    }
    const bool tcol = cyc ? true : (c >= -G && c <= nxl + 2 + G);      // column can hold an active T cell
    const bool ucol = cyc ? true : (c >= -G && c <= nxl + 1 + G);      // ... an active U cell
    const bool own = (lane >= 1 && lane <= STRIP2_W && c >= 1 - G && c <= nxl + G);   // columns this lane stores

    const size_t pp = (size_t)s.pitch * 16;
    const size_t rowb = (size_t)s.rstride * 16;
    const unsigned lo = (unsigned)(C0 + ci) * 16u, lom = (unsigned)(C0 + cm1) * 16u;
    const int SR = a.sr;
    const int SW = a.sw;
    char *const base = reinterpret_cast<char *>(s.F);

    // ---- carried state ----
    // first subcycle
    double uo_c = 0.0, vo_c = 0.0, uo_m = 0.0, vo_m = 0.0;        // u_old at (c, r-1), (c-1, r-1)
    double a1c = 0.0, a5c = 0.0, a2r = 0.0, a7r = 0.0;            // str terms of T1(r-1)
    Sig g1p{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                  // sigma after subcycle 1 at row r-1
    TMet mtp{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                       // metrics of row r-1
    unsigned char mp = 0, mpp = 0;                                // cmask of rows r-1, r-2
    // second subcycle
    double u1p_c = 0.0, v1p_c = 0.0, u1p_m = 0.0, v1p_m = 0.0;    // u after subcycle 1 at (c, r-2), (c-1, r-2)
    double b1c = 0.0, b5c = 0.0, b2r = 0.0, b7r = 0.0;            // str terms of T2(r-2)
    UStat qp{0, 0, 0, 0, 0, 0, 0, 0};                             // stepu inputs of row r-2
    double uip = 0.0, vip = 0.0;

    {   // u_old of the row below the first T1 row
        const int r0 = jb - 2;
        if (r0 >= 0) {
            const char *rb0 = base + (size_t)r0 * rowb;
            if (okc) { const double2 t = ldp(rb0, pp, SR + S_U, lo); uo_c = t.x; vo_c = t.y; }
            if (okm) { const double2 t = ldp(rb0, pp, SR + S_U, lom); uo_m = t.x; vo_m = t.y; }
        }
    }

    for (int t = 0; t <= R + 2; t++) {
        const int r = jb - 1 + t;                     // row of T1 in this step
        if (r > nyl + 2) break;
        const bool rowok = (r >= 0 && r <= nyl + 1);  // row exists in storage
        char *const rb = base + (size_t)(rowok ? r : 0) * rowb;

        // ---------------- stage 1: T1(r) ----------------
        unsigned char m = 0;
        double un_c = 0.0, vn_c = 0.0, un_m = 0.0, vn_m = 0.0;    // u_old at (c, r), (c-1, r)
        if (rowok) {
            if (okc) {
                m = s.cmask[(size_t)r * s.pitch + C0 + ci];
                const double2 q = ldp(rb, pp, SR + S_U, lo); un_c = q.x; vn_c = q.y;
            }
            if (okm) { const double2 q = ldp(rb, pp, SR + S_U, lom); un_m = q.x; vn_m = q.y; }
        }
        const bool t1act = tcol && (m & CM_T) != 0;
        Str8 o1{0, 0, 0, 0, 0, 0, 0, 0};
        Sig g1{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        TMet mt{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (__any(t1act)) {
            if (t1act) {
                mt = load_tmet(rb, pp, lo);
                g1 = load_sig(rb, pp, SR, lo);
                Diag dg;
                stress_cell<false>(mt, un_c, un_m, uo_c, uo_m, vn_c, vn_m, vo_c, vo_m, a.ecci, a.arlx1i, a.denom1, 0.0, g1, o1, dg);
            }
        }
        const double a2n = shfl_dn1(o1.s2), a4n = shfl_dn1(o1.s4), a7n = shfl_dn1(o1.s7), a8n = shfl_dn1(o1.s8);

        // ---------------- stage 1: U1(r-1) ----------------
        // velocity after the first subcycle; an inactive cell keeps its value
        double u1_c = uo_c, v1_c = vo_c;
        UStat q1{0, 0, 0, 0, 0, 0, 0, 0};
        double ui1 = 0.0, vi1 = 0.0;
        const bool u1act = (t >= 1) && ucol && (mp & CM_U) != 0 && (r - 1 >= 1) && (r - 1 <= nyl);
        if (__any(u1act)) {
            if (u1act) {
                const char *ru = base + (size_t)(r - 1) * rowb;
                q1 = load_ustat(ru, pp, lo);
                if (REVP) { const double2 iv = ldp(ru, pp, F_UVEL_INIT, lo); ui1 = iv.x; vi1 = iv.y; }
                double sxi, syi;
                stepu_cell(q1, uo_c, vo_c, ui1, vi1, ((a1c + a2r) + o1.s3) + a4n, ((a5c + o1.s6) + a7r) + a8n,
                           a.brlx, a.revp, a.cosw, a.sinw, u1_c, v1_c, sxi, syi);
            }
        }
        const double u1_m = shfl_up1(u1_c), v1_m = shfl_up1(v1_c);      // (c-1, r-1); lane 0 is not used below

        // ---------------- stage 2: T2(r-1) ----------------
        const int q2 = r - 1;
        const bool t2act = (t >= 2) && tcol && (mp & CM_T) != 0 && lane >= 1;
        Str8 o2{0, 0, 0, 0, 0, 0, 0, 0};
        if (__any(t2act)) {
            if (t2act) {
                Sig g2 = g1p;
                Diag dg;
                char *const rq = base + (size_t)q2 * rowb;
                double tarear = 0.0;
                if (LAST2) tarear = *reinterpret_cast<const double *>(rq + (size_t)(F_TAREAR >> 1) * pp + lo + (F_TAREAR & 1) * 8);
                stress_cell<LAST2>(mtp, u1_c, u1_m, u1p_c, u1p_m, v1_c, v1_m, v1p_c, v1p_m, a.ecci, a.arlx1i, a.denom1, tarear, g2, o2, dg);
                if (own && q2 >= jb && q2 < jb + R && q2 <= a.jmax) {
                    store_sig(rq, pp, SW, lo, g2);
                    if (cyc && c == 1) store_sig(rq, pp, SW, lo + (unsigned)nxl * 16u, g2);     // east ghost T column = image of column 1
                    if (LAST2) {    // the second subcycle is the last one of this evp: ridging diagnostics (ice_dyn_evp.F90:665-677)
                        st1(rq, pp, F_DIVU, lo, dg.divu);       st1(rq, pp, F_RDGCONV, lo, dg.rdg_conv);
                        st1(rq, pp, F_RDGSHEAR, lo, dg.rdg_shear); st1(rq, pp, F_SHEAR, lo, dg.shear);
                        st1(rq, pp, F_PRSSIG, lo, dg.prs);
                    }
                }
            }
        }
        const double b2n = shfl_dn1(o2.s2), b4n = shfl_dn1(o2.s4), b7n = shfl_dn1(o2.s7), b8n = shfl_dn1(o2.s8);

        // ---------------- stage 2: U2(r-2) ----------------
        const int q3 = r - 2;
        const bool u2act = (t >= 3) && own && (mpp & CM_U) != 0 && q3 >= jb && q3 < jb + R && q3 <= nyl && q3 <= a.jmax;
        if (__any(u2act)) {
            if (u2act) {
                double un, vn, sxi, syi;
                stepu_cell(qp, u1p_c, v1p_c, uip, vip, ((b1c + b2r) + o2.s3) + b4n, ((b5c + o2.s6) + b7r) + b8n,
                           a.brlx, a.revp, a.cosw, a.sinw, un, vn, sxi, syi);
                char *const ru = base + (size_t)q3 * rowb;
                stp(ru, pp, SW + S_U, lo, un, vn);
                if (cyc) {
                    if (c == 1) stp(ru, pp, SW + S_U, lo + (unsigned)nxl * 16u, un, vn);
                    if (c == nxl) stp(ru, pp, SW + S_U, lo - (unsigned)nxl * 16u, un, vn);
                }
                if (LAST2) { st1(ru, pp, F_STRINTX, lo, sxi); st1(ru, pp, F_STRINTY, lo, syi); }
            }
        }

        // ---------------- rotate ----------------
        b1c = o2.s1; b5c = o2.s5; b2r = b2n; b7r = b7n;
        u1p_c = u1_c; v1p_c = v1_c; u1p_m = u1_m; v1p_m = v1_m;
        qp = q1; uip = ui1; vip = vi1;
        a1c = o1.s1; a5c = o1.s5; a2r = a2n; a7r = a7n;
        g1p = g1; mtp = mt;
        uo_c = un_c; vo_c = vn_c; uo_m = un_m; vo_m = vn_m;
        mpp = mp; mp = m;
    }
}

template __global__ void k_subcycle2<false, false>(SubArgs);
template __global__ void k_subcycle2<true, false>(SubArgs);
template __global__ void k_subcycle2<false, true>(SubArgs);
template __global__ void k_subcycle2<true, true>(SubArgs);

// ---- the same in two parts, for k_subcycle3w: the strain-rate half needs no sigma, so sigma can be fetched from the LDS between
// the halves (24 registers less while the twelve strain rates, four Deltas and four divisions are in flight).  Same expressions in
// the same order as stress_cell<false>: the same bits. ----
struct StrainQ {                        // what the sigma update reads: (div - Delta), tension, shear per corner, c1 and c0 = c1 * ecci
    double dne, dnw, dsw, dse, tne, tnw, tsw, tse, sne, snw, ssw, sse, c1ne, c1nw, c1sw, c1se, c0ne, c0nw, c0sw, c0se;
};
__device__ __forceinline__ void stress_strain(const TMet &m, double u_ij, double u_mj, double u_im, double u_mm,
                                              double v_ij, double v_mj, double v_im, double v_mm,
                                              double ecci, double arlx1i, StrainQ &q) {
    const double cxp = m.cxp, cyp = m.cyp, cxm = m.cxm, cym = m.cym, dxt = m.dxt, dyt = m.dyt;
    // strain rates * area (:627-654)
    const double divune = cyp * u_ij - dyt * u_mj + cxp * v_ij - dxt * v_im;
    const double divunw = cym * u_mj + dyt * u_ij + cxp * v_mj - dxt * v_mm;
    const double divusw = cym * u_mm + dyt * u_im + cxm * v_mm + dxt * v_mj;
    const double divuse = cyp * u_im - dyt * u_mm + cxm * v_im + dxt * v_ij;
    q.tne = -cym * u_ij - dyt * u_mj + cxm * v_ij + dxt * v_im;
    q.tnw = -cyp * u_mj + dyt * u_ij + cxm * v_mj + dxt * v_mm;
    q.tsw = -cyp * u_mm + dyt * u_im + cxp * v_mm - dxt * v_mj;
    q.tse = -cym * u_im - dyt * u_mm + cxp * v_im - dxt * v_ij;
    q.sne = -cym * v_ij - dyt * v_mj - cxm * u_ij - dxt * u_im;
    q.snw = -cyp * v_mj + dyt * v_ij - cxm * u_mj - dxt * u_mm;
    q.ssw = -cyp * v_mm + dyt * v_im - cxp * u_mm + dxt * u_mj;
    q.sse = -cym * v_im - dyt * v_mm - cxp * u_im + dxt * u_ij;
    // Delta (:657-660)
    const double Deltane = sqrt(divune * divune + ecci * (q.tne * q.tne + q.sne * q.sne));
    const double Deltanw = sqrt(divunw * divunw + ecci * (q.tnw * q.tnw + q.snw * q.snw));
    const double Deltase = sqrt(divuse * divuse + ecci * (q.tse * q.tse + q.sse * q.sse));
    const double Deltasw = sqrt(divusw * divusw + ecci * (q.tsw * q.tsw + q.ssw * q.ssw));
    // replacement pressure / Delta (:683-697)
    const double c0ne = m.strength / fmax(Deltane, m.tiny);
    const double c0nw = m.strength / fmax(Deltanw, m.tiny);
    const double c0sw = m.strength / fmax(Deltasw, m.tiny);
    const double c0se = m.strength / fmax(Deltase, m.tiny);
    q.c1ne = c0ne * arlx1i; q.c1nw = c0nw * arlx1i; q.c1sw = c0sw * arlx1i; q.c1se = c0se * arlx1i;
    q.c0ne = q.c1ne * ecci; q.c0nw = q.c1nw * ecci; q.c0sw = q.c1sw * ecci; q.c0se = q.c1se * ecci;
    q.dne = divune - Deltane; q.dnw = divunw - Deltanw; q.dsw = divusw - Deltasw; q.dse = divuse - Deltase;
}
__device__ __forceinline__ void stress_update(const StrainQ &q, double dxt, double dyt, double dxhy, double dyhx, double denom1, Sig &g, Str8 &o) {
    const double p111 = 1.0 / 9.0, p055 = p111 * 0.5, p027 = p055 * 0.5;
    const double p166 = 1.0 / 6.0, p222 = 2.0 / 9.0, p333 = 1.0 / 3.0;
    // the stresses (:704-721)
    const double sp1 = (g.sp1 + q.c1ne * q.dne) * denom1;
    const double sp2 = (g.sp2 + q.c1nw * q.dnw) * denom1;
    const double sp3 = (g.sp3 + q.c1sw * q.dsw) * denom1;
    const double sp4 = (g.sp4 + q.c1se * q.dse) * denom1;
    const double sm1 = (g.sm1 + q.c0ne * q.tne) * denom1;
    const double sm2 = (g.sm2 + q.c0nw * q.tnw) * denom1;
    const double sm3 = (g.sm3 + q.c0sw * q.tsw) * denom1;
    const double sm4 = (g.sm4 + q.c0se * q.tse) * denom1;
    const double s121 = (g.s121 + q.c0ne * q.sne * 0.5) * denom1;
    const double s122 = (g.s122 + q.c0nw * q.snw * 0.5) * denom1;
    const double s123 = (g.s123 + q.c0sw * q.ssw * 0.5) * denom1;
    const double s124 = (g.s124 + q.c0se * q.sse * 0.5) * denom1;
    g = Sig{sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124};

    // combinations for the momentum equation (:752-795)
    const double ssigpn = sp1 + sp2, ssigps = sp3 + sp4, ssigpe = sp1 + sp4, ssigpw = sp2 + sp3;
    const double ssigp1 = (sp1 + sp3) * p055, ssigp2 = (sp2 + sp4) * p055;
    const double ssigmn = sm1 + sm2, ssigms = sm3 + sm4, ssigme = sm1 + sm4, ssigmw = sm2 + sm3;
    const double ssigm1 = (sm1 + sm3) * p055, ssigm2 = (sm2 + sm4) * p055;
    const double ssig12n = s121 + s122, ssig12s = s123 + s124, ssig12e = s121 + s124, ssig12w = s122 + s123;
    const double ssig121 = (s121 + s123) * p111, ssig122 = (s122 + s124) * p111;

    const double csigpne = p111 * sp1 + ssigp2 + p027 * sp3;
    const double csigpnw = p111 * sp2 + ssigp1 + p027 * sp4;
    const double csigpsw = p111 * sp3 + ssigp2 + p027 * sp1;
    const double csigpse = p111 * sp4 + ssigp1 + p027 * sp2;
    const double csigmne = p111 * sm1 + ssigm2 + p027 * sm3;
    const double csigmnw = p111 * sm2 + ssigm1 + p027 * sm4;
    const double csigmsw = p111 * sm3 + ssigm2 + p027 * sm1;
    const double csigmse = p111 * sm4 + ssigm1 + p027 * sm2;
    const double csig12ne = p222 * s121 + ssig122 + p055 * s123;
    const double csig12nw = p222 * s122 + ssig121 + p055 * s124;
    const double csig12sw = p222 * s123 + ssig122 + p055 * s121;
    const double csig12se = p222 * s124 + ssig121 + p055 * s122;

    const double str12ew = 0.5 * dxt * (p333 * ssig12e + p166 * ssig12w);
    const double str12we = 0.5 * dxt * (p333 * ssig12w + p166 * ssig12e);
    const double str12ns = 0.5 * dyt * (p333 * ssig12n + p166 * ssig12s);
    const double str12sn = 0.5 * dyt * (p333 * ssig12s + p166 * ssig12n);

    // dF/dx (:800-820)
    double strp_tmp = 0.25 * dyt * (p333 * ssigpn + p166 * ssigps);
    double strm_tmp = 0.25 * dyt * (p333 * ssigmn + p166 * ssigms);
    o.s1 = -strp_tmp - strm_tmp - str12ew + dxhy * (-csigpne + csigmne) + dyhx * csig12ne;
    o.s2 = strp_tmp + strm_tmp - str12we + dxhy * (-csigpnw + csigmnw) + dyhx * csig12nw;
    strp_tmp = 0.25 * dyt * (p333 * ssigps + p166 * ssigpn);
    strm_tmp = 0.25 * dyt * (p333 * ssigms + p166 * ssigmn);
    o.s3 = -strp_tmp - strm_tmp + str12ew + dxhy * (-csigpse + csigmse) + dyhx * csig12se;
    o.s4 = strp_tmp + strm_tmp + str12we + dxhy * (-csigpsw + csigmsw) + dyhx * csig12sw;
    // dF/dy (:825-845)
    strp_tmp = 0.25 * dxt * (p333 * ssigpe + p166 * ssigpw);
    strm_tmp = 0.25 * dxt * (p333 * ssigme + p166 * ssigmw);
    o.s5 = -strp_tmp + strm_tmp - str12ns - dyhx * (csigpne + csigmne) + dxhy * csig12ne;
    o.s6 = strp_tmp - strm_tmp - str12sn - dyhx * (csigpse + csigmse) + dxhy * csig12se;
    strp_tmp = 0.25 * dxt * (p333 * ssigpw + p166 * ssigpe);
    strm_tmp = 0.25 * dxt * (p333 * ssigmw + p166 * ssigme);
    o.s7 = -strp_tmp + strm_tmp + str12ns - dyhx * (csigpnw + csigmnw) + dxhy * csig12nw;
    o.s8 = strp_tmp - strm_tmp + str12sn - dyhx * (csigpsw + csigmsw) + dxhy * csig12sw;
}


// ------------------------------------------------------------------------------------
// k_subcycle3w: THREE subcycles per launch, ONE WAVE PER SUBCYCLE STAGE (round 4).
// The pair kernel keeps both of its subcycles in one wave's registers: 253 VGPRs, two waves per SIMD, the VALU 60 % busy.  A
// third subcycle does not fit a wave.  Here a workgroup of three waves takes one strip: wave S runs subcycle S + 1 (stress +
// stepu, k_subcycle's working set) two rows and one step behind wave S - 1, all three marching north in lockstep -- one
// s_barrier per row step.  What crosses from one subcycle to the next is sigma (12 doubles per cell) and (u, v); it crosses
// through the LDS:
//   sigma ring  5 slots x 6 KiB   row q: written by wave 0 at step q, rewritten IN PLACE by wave 1 at step q + 2, read by wave 2 at
//                                 step q + 4 (the 2-step lag: T(q) of the next subcycle needs u(q), which needs T(q + 1))
//   u rings     2 x 2 slots x 1 KiB  (u, v) of one row, written at step q + 1, read (lanes l and l - 1) at step q + 2
//   staging     6 KiB             sigma of the NEXT row for wave 0 by direct global->LDS loads (no VGPRs)
// = 40 KiB per workgroup, four workgroups (12 waves, three per SIMD at <= 168 VGPRs) per CU.  Grid metrics, the stepu inputs and
// the mask are read by every wave for its own row from global memory a step ahead into registers -- wave 0 from HBM, waves 1, 2
// out of the L2 two and four steps later -- so sigma, the metrics and the stepu inputs cross HBM ONCE PER THREE subcycles.
//   lanes: T1 64, U1 0..62, T2 1..62, U2 1..61, T3 2..61, U3 2..60: 59 owned columns per strip;
//   rows:  wave S computes T rows jb-2+S .. jb+R+2-S and U rows jb-2+S .. jb+R+1-S; U rows jb .. jb+R-1 of wave 2 are stored.
// Arithmetic, operation order, mask rules (an inactive U cell keeps its velocity, an inactive T cell contributes no str) and the
// column / ghost-zone conventions are k_subcycle2's: bit-identical to three k_subcycle launches.
// ------------------------------------------------------------------------------------
constexpr int STRIP3_W = 58;
constexpr int STRIP3_OWN0 = 3;     // first owned lane
constexpr int K3_LDS_ROWS = 40;      // 1-KiB rows (64 double2): 30 sigma ring, 2 + 2 u rings, 6 staging

template <bool REVP, bool CM>
__global__ __launch_bounds__(192, 3) void k_subcycle3w(SubArgs a) {
    __shared__ double2 smem[K3_LDS_ROWS * 64];
    const Slab &s = a.s;
    const int lane = threadIdx.x & 63;
    const int S = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // this wave's subcycle stage: 0, 1, 2
    // one strip per workgroup.  Every exit below is taken by the whole workgroup: all three waves pass the same number of barriers.
    // The list is sorted by work, longest strips first (k_sort_strips), and the dispatcher hands out workgroups in blockIdx order:
    // block b takes entry b (the XCDs, b mod 8, then get equal shares of every length class).
    const int ns = a.nsdev ? __builtin_amdgcn_readfirstlane(*a.nsdev) : a.nstrips;
    const int sid = (int)blockIdx.x;
    if (sid >= ns) return;
    const int st = __builtin_amdgcn_readfirstlane(a.strips[sid]);
    if (S == 0) dbg_stamp(a, sid, 0);
    const int cx = st % a.ncx, ry = st / a.ncx;
    const int R = a.R, nxl = s.nxl, nyl = s.nyl;
    const int G = a.G;
    const int c = cx * STRIP3_W + lane - (STRIP3_OWN0 - 1) - G;       // unwrapped column of this lane (lane 3 = first owned column)
    const int jb = ry * R + 1;
    const bool cyc = a.wrap != 0;

    // storage column of (c); outside a non-cyclic domain the lane is dead.  Lane 0 only carries the west neighbour's (u, v) and HTE
    // for lane 1: every (c - 1) operand is a one-lane DPP shift, not a second load.
    int ci = c;
    bool okc;
    if (cyc) {
        ci = (c - 1) % nxl; if (ci < 0) ci += nxl; ci += 1;
        okc = true;
    } else {
        okc = (c >= -1 - G && c <= nxl + 2 + G);
        if (!okc) ci = 0;
    }
    const bool tcol = cyc ? true : (c >= -G && c <= nxl + 2 + G);
    const bool ucol = cyc ? true : (c >= -G && c <= nxl + 1 + G);
    const bool own = (lane >= STRIP3_OWN0 && lane < STRIP3_OWN0 + STRIP3_W && c >= 1 - G && c <= nxl + G);
    const bool laneT = tcol && lane >= S + 1 && lane <= 63 - S;                    // lanes whose T cell of this stage has valid inputs
    const bool laneU = ucol && lane >= S + 1 && lane <= 62 - S && (S < 2 || own);  // ... U cell (the last stage: the columns it stores)

    const size_t pp = (size_t)s.pitch * 16;
    const size_t rowb = (size_t)s.rstride * 16;
    const unsigned lo = (unsigned)(C0 + ci) * 16u;
    const int SR = a.sr;
    const int SW = a.sw;
    char *const base = reinterpret_cast<char *>(s.F);

    double2 *const SIG = smem;                        // [5][6][64]
    double2 *const Uin = smem + (S == 1 ? 30 : 32) * 64;      // the u ring this stage reads (S >= 1): [2][64]
    double2 *const Uout = smem + (S == 0 ? 30 : 32) * 64;     // ... and writes (S <= 1)
    double2 *const STG = smem + 34 * 64;              // [6][64], wave 0 only

    // row windows of this stage
    const int tlo = jb - 2 + S, thi = jb + R + 2 - S;          // T rows
    const int ulo = jb - 2 + S, uhi = jb + R + 1 - S;          // U rows
    auto rowok = [&](int r) { return r >= 0 && r <= nyl + 1; };
    auto mask_of = [&](int r) -> unsigned char { return (rowok(r) && okc) ? s.cmask[(size_t)r * s.pitch + C0 + ci] : (unsigned char)0; };

    // ---- operands on their way (global -> registers, one step ahead) ----
    double2 pu_c = make_double2(0.0, 0.0);                                      // stage 0: (u, v) at c of the next T row
    double2 ph = make_double2(0.0, 0.0);                                        // CM: (HTN, HTE) at c
    double2 pm0 = make_double2(0.0, 0.0), pm1 = pm0, pm2 = pm0, pm3 = pm0;      // !CM: the four metric pairs
    double2 pts = make_double2(0.0, 0.0);                                       // (tinyarea, strength)
    double2 pq0 = make_double2(0.0, 0.0), pq1 = pq0, pq2 = pq0, pq3 = pq0, piv = pq0;   // stepu inputs of the next U row
    auto issue = [&](int rn, unsigned char mt_, unsigned char mu_, unsigned char mt_next) {
        if (rowok(rn) && rn >= tlo - 1 && rn <= thi) {
            const char *rbn = base + (size_t)rn * rowb;
            if (S == 0 && okc) pu_c = ldp(rbn, pp, SR + S_U, lo);
            const bool ta = laneT && (mt_ & CM_T) != 0 && rn >= tlo;
            if (CM) {   // HTN of this row is also the south length of the next row, HTE the west length of the east neighbour:
                        // fetch the pair if this cell, the next row's or the east neighbour's is active
                const bool tn = tcol && ((mt_ | mt_next) & CM_T) != 0;
                const bool te = __shfl_down((int)((mt_ & CM_T) != 0), 1) != 0;      // (every lane takes part: not behind `tn ||`)
                const bool th = okc && (tn || te);
                if (__any(th)) { if (th) ph = ldp(rbn, pp, F_HTN, lo); }
            }
            if (__any(ta)) {
                if (ta) {
                    if (!CM) {
                        pm0 = ldp(rbn, pp, F_CXP, lo); pm1 = ldp(rbn, pp, F_CXM, lo);
                        pm2 = ldp(rbn, pp, F_DXT, lo); pm3 = ldp(rbn, pp, F_DXHY, lo);
                    }
                    pts = ldp(rbn, pp, F_TINYAREA, lo);
                }
            }
        }
        const int ru_ = rn - 1;
        const bool ua = laneU && (mu_ & CM_U) != 0 && ru_ >= 1 && ru_ <= nyl && ru_ >= ulo && ru_ <= uhi;
        if (__any(ua)) {
            if (ua) {
                const char *rbu = base + (size_t)ru_ * rowb;
                pq0 = ldp(rbu, pp, F_VRELC, lo); pq1 = ldp(rbu, pp, F_UOCN, lo);
                pq2 = ldp(rbu, pp, F_FORCEX, lo); pq3 = ldp(rbu, pp, F_UMASSDTI, lo);
                if (REVP) piv = ldp(rbu, pp, F_UVEL_INIT, lo);
            }
        }
    };

    // wave 0: sigma of T row rn, global -> LDS staging, no registers (issued once this step's sigma has been read out of the staging)
    auto stage_sigma = [&](int rn, unsigned char mt_) {
        const bool ta = laneT && (mt_ & CM_T) != 0 && rowok(rn) && rn >= tlo && rn <= thi;
        if (__any(ta)) {
            if (ta) {
                const char *rbn = base + (size_t)rn * rowb;
#pragma unroll
                for (int q = 0; q < 6; q++) lds_dma16(rbn + (size_t)((SR + S_SP) / 2 + q) * pp + lo, STG + q * 64);
            }
        }
    };

    // ---- carried from the previous step ----
    double uo_c = 0.0, vo_c = 0.0;                                // velocity entering this stage at (c, r-1)
    double a1c = 0.0, a5c = 0.0, a2r = 0.0, a7r = 0.0;            // str terms of T(r-1)
    double hn_p = 0.0;                                            // CM: HTN(c, r-1)
    unsigned char mp = 0;

    const int r_first = jb - 2 - 2 * S;                           // T row of step 0 (stages 1, 2: below their windows)
    if (S == 0) {   // velocity and south length of the row below the first T1 row
        const int r0 = r_first - 1;
        if (r0 >= 0) {
            const char *rb0 = base + (size_t)r0 * rowb;
            if (okc) { const double2 t = ldp(rb0, pp, SR + S_U, lo); uo_c = t.x; vo_c = t.y; }
        }
    }
    unsigned char m = mask_of(r_first), m_n1 = mask_of(r_first + 1), m_n2 = mask_of(r_first + 2);
    issue(r_first, m, 0, m_n1);
    if (S == 0) stage_sigma(r_first, m);
    if (CM && S == 0) {   // south length of the first T1 row (the later stages pick theirs up on the way to their windows)
        const int r0 = tlo - 1;
        if (r0 >= 0 && okc) hn_p = *reinterpret_cast<const double *>(base + (size_t)r0 * rowb + (size_t)(F_HTN >> 1) * pp + lo);
    }

    const int nstep = min(R, nyl + 1 - jb) + 6;                   // the last step stores U3 of the strip's top row
    for (int t = 0; t <= nstep; t++) {
        const int r = r_first + t;                    // T row of this stage in this step
        const bool rok = rowok(r);
        const int slot = (t + 10 - 2 * S) % 5;        // sigma ring slot of row r

        // ---------------- operands of this step ----------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        double un_c = 0.0, vn_c = 0.0;                // velocity entering this stage at (c, r)
        if (S == 0) {
            if (rok && okc) { un_c = pu_c.x; vn_c = pu_c.y; }
        } else if (r >= ulo - 1 && r <= uhi + 1) {    // rows the previous stage has written: its U window
            const double2 q = Uin[(r & 1) * 64 + lane]; un_c = q.x; vn_c = q.y;
        }
        // the west neighbour's, rows r and r-1 (lane 0 has none and computes nothing)
        const double un_m = shfl_up1(un_c), vn_m = shfl_up1(vn_c), uo_m = shfl_up1(uo_c), vo_m = shfl_up1(vo_c);
        const bool tact = laneT && rok && r >= tlo && r <= thi && (m & CM_T) != 0;
        TMet mt{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        double hn = 0.0;
        double he = 0.0;
        if (CM) {
            const bool tn = tcol && ((m | m_n1) & CM_T) != 0;
            const bool te = __shfl_down((int)((m & CM_T) != 0), 1) != 0;
            const bool th = okc && (tn || te) && rok && r >= tlo - 1 && r <= thi;
            if (__any(th)) { if (th) { hn = ph.x; he = ph.y; } }
        }
        const double hw = shfl_up1(he);               // HTE(c-1, r)
        if (__any(tact)) {
            if (tact) {
                if (CM) mt = tmet_from_lengths(hn, hn_p, he, hw, pts.x, pts.y);
                else mt = TMet{pm0.x, pm0.y, pm1.x, pm1.y, pm2.x, pm2.y, pm3.x, pm3.y, pts.x, pts.y};
            }
        }
        const bool uact = laneU && (mp & CM_U) != 0 && (r - 1 >= 1) && (r - 1 <= nyl) && r - 1 >= ulo && r - 1 <= uhi;
        UStat q1s{0, 0, 0, 0, 0, 0, 0, 0};
        double ui1 = 0.0, vi1 = 0.0;
        if (__any(uact)) {
            if (uact) {
                q1s = UStat{pq0.x, pq0.y, pq1.x, pq1.y, pq2.x, pq2.y, pq3.x, pq3.y};
                if (REVP) { ui1 = piv.x; vi1 = piv.y; }
            }
        }

        // ---------------- loads for the next step (registers), mask three rows ahead ----------------
        const unsigned char m_n3 = mask_of(r + 3);
        issue(r + 1, m_n1, m, m_n2);

        // ---------------- T(r) ----------------
        Str8 o{0, 0, 0, 0, 0, 0, 0, 0};
        const bool anyT = __any(tact);
        if (anyT) {
            if (tact) {
                StrainQ sq;
                stress_strain(mt, un_c, un_m, uo_c, uo_m, vn_c, vn_m, vo_c, vo_m, a.ecci, a.arlx1i, sq);
                asm volatile("" ::: "memory");      // sigma is fetched HERE, not before the strain rates (24 registers)
                const double2 *src = (S == 0) ? STG : SIG + slot * 6 * 64;
                const double2 q0 = src[0 * 64 + lane], q1 = src[1 * 64 + lane], q2 = src[2 * 64 + lane];
                const double2 q3 = src[3 * 64 + lane], q4 = src[4 * 64 + lane], q5 = src[5 * 64 + lane];
                Sig g{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y, q4.x, q4.y, q5.x, q5.y};
                stress_update(sq, mt.dxt, mt.dyt, mt.dxhy, mt.dyhx, a.denom1, g, o);
                if (S < 2) {        // sigma after this subcycle -> ring slot of row r (wave 1: in place)
                    double2 *dst = SIG + slot * 6 * 64 + lane;
                    dst[0 * 64] = make_double2(g.sp1, g.sp2);   dst[1 * 64] = make_double2(g.sp3, g.sp4);
                    dst[2 * 64] = make_double2(g.sm1, g.sm2);   dst[3 * 64] = make_double2(g.sm3, g.sm4);
                    dst[4 * 64] = make_double2(g.s121, g.s122); dst[5 * 64] = make_double2(g.s123, g.s124);
                } else if (own && r >= jb && r < jb + R && r <= a.jmax) {
                    char *const rq = base + (size_t)r * rowb;
                    store_sig(rq, pp, SW, lo, g);
                    if (cyc && c == 1) store_sig(rq, pp, SW, lo + (unsigned)nxl * 16u, g);     // east ghost T column = image of column 1
                }
            }
        }
        if (S == 0) {       // the staging is free again: sigma of the next T1 row, to be there a step from now
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            stage_sigma(r + 1, m_n1);
        }
        const double a2n = shfl_dn1(o.s2), a4n = shfl_dn1(o.s4), a7n = shfl_dn1(o.s7), a8n = shfl_dn1(o.s8);

        // ---------------- U(r-1) ----------------
        double u1_c = uo_c, v1_c = vo_c;              // an inactive cell keeps its velocity
        if (__any(uact)) {
            if (uact) {
                double sxi, syi;
                stepu_cell(q1s, uo_c, vo_c, ui1, vi1, ((a1c + a2r) + o.s3) + a4n, ((a5c + o.s6) + a7r) + a8n,
                           a.brlx, a.revp, a.cosw, a.sinw, u1_c, v1_c, sxi, syi);
                if (S == 2 && r - 1 <= a.jmax) {
                    char *const ru = base + (size_t)(r - 1) * rowb;
                    stp(ru, pp, SW + S_U, lo, u1_c, v1_c);
                    if (cyc) {
                        if (c == 1) stp(ru, pp, SW + S_U, lo + (unsigned)nxl * 16u, u1_c, v1_c);
                        if (c == nxl) stp(ru, pp, SW + S_U, lo - (unsigned)nxl * 16u, u1_c, v1_c);
                    }
                }
            }
        }
        if (S < 2 && r - 1 >= ulo && r - 1 <= uhi) Uout[((r - 1) & 1) * 64 + lane] = make_double2(u1_c, v1_c);

        // ---------------- rotate ----------------
        a1c = o.s1; a5c = o.s5; a2r = a2n; a7r = a7n;
        if (CM) hn_p = hn;
        uo_c = un_c; vo_c = vn_c;
        mp = m; m = m_n1; m_n1 = m_n2; m_n2 = m_n3;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // this step's LDS writes are in place; loads stay in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may be in flight when the wave ends
    if (S == 2) dbg_stamp(a, sid, 1);
}

template __global__ void k_subcycle3w<false, false>(SubArgs);
template __global__ void k_subcycle3w<true, false>(SubArgs);
template __global__ void k_subcycle3w<false, true>(SubArgs);
template __global__ void k_subcycle3w<true, true>(SubArgs);


}  // namespace evpk
