// evpk_kernels.hip -- CDNA4 (gfx950) kernels of the EVP sea-ice dynamics path.
//
// Built with -ffp-contract=off: the arithmetic is the plain IEEE fp64 sequence the
// reference spells out, so results are bit-comparable with a non-FMA CPU build.
// fp64 sqrt and divide lower to correctly rounded sequences on gfx950.
//
// Hot kernel: k_subcycle -- one launch = one EVP subcycle = stress (ice_dyn_evp.F90:520-849)
// fused with stepu (ice_dyn_shared.F90:623-748).  The str(:,:,1:8) work array of the
// reference never reaches HBM: it lives in registers and moves between lanes with DPP/
// bpermute shuffles.  u, v and the twelve sigma planes are double buffered (read buffer
// `cur`, write buffer `cur^1`) so that redundant T cells on strip edges see old values.
#include "evpk_internal.h"

namespace evpk {

// ------------------------------------------------------------------------------------
// gather / scatter between the reference's block layout and the slab
// ------------------------------------------------------------------------------------
enum { MODE_PHYS = 0, MODE_ALL = 1, MODE_NE = 2 };

template <typename T>
__global__ void k_gather(Slab s, const BlockDesc *bd, int nblocks, int nxb, int nyb, const T *src, T *dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;   // 1-based block column
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    const bool phys = (i >= d.ilo && i <= d.ihi && j >= d.jlo && j <= d.jhi);
    const bool ring = (si == 0 || si == s.nxl + 1 || sj == 0 || sj == s.nyl + 1);
    if (!phys && !(ring && i <= d.ihi + 1 && j <= d.jhi + 1)) return;
    dst[cell(s, si, sj)] = src[((size_t)b * nyb + (j - 1)) * nxb + (i - 1)];
}

template <typename T>
__global__ void k_scatter(Slab s, const BlockDesc *bd, int nblocks, int nxb, int nyb, const T *src, T *dst, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    if (i > d.ihi + 1 || j > d.jhi + 1) return;   // padding
    bool take;
    if (mode == MODE_ALL) take = true;
    else if (mode == MODE_NE) take = (i >= d.ilo && j >= d.jlo);
    else take = (i >= d.ilo && i <= d.ihi && j >= d.jlo && j <= d.jhi);
    if (!take) return;
    dst[((size_t)b * nyb + (j - 1)) * nxb + (i - 1)] = src[cell(s, si, sj)];
}

template __global__ void k_gather<double>(Slab, const BlockDesc *, int, int, int, const double *, double *);
template __global__ void k_gather<int32_t>(Slab, const BlockDesc *, int, int, int, const int32_t *, int32_t *);
template __global__ void k_scatter<double>(Slab, const BlockDesc *, int, int, int, const double *, double *, int);
template __global__ void k_scatter<int32_t>(Slab, const BlockDesc *, int, int, int, const int32_t *, int32_t *, int);

// all cells of the slab incl. ring: thread (i,j), i = 0..nxl+1, j = 0..nyl+1
#define SLAB_IJ_ALL                                            \
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       \
    const int j = blockIdx.y * blockDim.y + threadIdx.y;       \
    if (i > s.nxl + 1 || j > s.nyl + 1) return;                \
    const size_t k = cell(s, i, j);

// ------------------------------------------------------------------------------------
// evp_prep1 (ice_dyn_shared.F90:270-365) on the slab
// ------------------------------------------------------------------------------------
__global__ void k_prep1a(Slab s, DevParams p) {
    SLAB_IJ_ALL
    const double vice = plane(s, F_VICE)[k], vsno = plane(s, F_VSNO)[k], aice = plane(s, F_AICE)[k];
    const bool tm = s.tmask[k] != 0;
    double tmass = 0.0;
    if (tm) tmass = (p.rhoi * vice + p.rhos * vsno);                              // :322-326
    plane(s, F_TMASS)[k] = tmass;
    s.tmphm[k] = (tm && (aice > p.a_min) && (tmass > p.m_min)) ? 1 : 0;           // :331-332
    plane(s, F_STRAIRX)[k] = plane(s, F_STRAIRXT)[k];                             // :339-340
    plane(s, F_STRAIRY)[k] = plane(s, F_STRAIRYT)[k];
    // evp(): zero the diagnostics (ice_dyn_evp.F90:174-182)
    plane(s, F_RDGCONV)[k] = 0.0; plane(s, F_RDGSHEAR)[k] = 0.0; plane(s, F_DIVU)[k] = 0.0;
    plane(s, F_SHEAR)[k] = 0.0; plane(s, F_PRSSIG)[k] = 0.0;
}

__global__ void k_prep1b(Slab s) {
    SLAB_IJ_ALL
    double m = 0.0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {                           // :350-363
        const unsigned char *t = s.tmphm;
        bool any = t[cell(s, i - 1, j + 1)] | t[cell(s, i, j + 1)] | t[cell(s, i + 1, j + 1)] |
                   t[cell(s, i - 1, j)] | t[k] | t[cell(s, i + 1, j)] |
                   t[cell(s, i - 1, j - 1)] | t[cell(s, i, j - 1)] | t[cell(s, i + 1, j - 1)];
        if (any) m = 1.0;
        if (!s.tmask[k]) m = 0.0;
    }
    plane(s, F_ICETM)[k] = m;
}

// to_ugrid (ice_grid.F90:1834-1878): dst = 0 outside the physical cells
__global__ void k_to_ugrid(Slab s, int fsrc, int fdst) {
    SLAB_IJ_ALL
    double r = 0.0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {
        const double *w = plane(s, fsrc), *ta = plane(s, F_TAREA);
        const size_t ke = cell(s, i + 1, j), kn = cell(s, i, j + 1), kne = cell(s, i + 1, j + 1);
        r = 0.25 * (((w[k] * ta[k] + w[ke] * ta[ke]) + w[kn] * ta[kn]) + w[kne] * ta[kne]) / plane(s, F_UAREA)[k];
    }
    plane(s, fdst)[k] = r;
}

// to_tgrid (ice_grid.F90:1924-1958): only physical cells of dst are written
__global__ void k_to_tgrid(Slab s, int fsrc, int fdst) {
    SLAB_IJ_ALL
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {
        const double *w = plane(s, fsrc), *ua = plane(s, F_UAREA);
        const size_t kw = cell(s, i - 1, j), ks = cell(s, i, j - 1), ksw = cell(s, i - 1, j - 1);
        plane(s, fdst)[k] = 0.25 * (((w[k] * ua[k] + w[kw] * ua[kw]) + w[ks] * ua[ks]) + w[ksw] * ua[ksw]) / plane(s, F_TAREA)[k];
    }
}

__global__ void k_copy_plane(Slab s, int fsrc, int fdst) {
    SLAB_IJ_ALL
    plane(s, fdst)[k] = plane(s, fsrc)[k];
}

// ------------------------------------------------------------------------------------
// evp_prep2 (ice_dyn_shared.F90:377-614) on the slab.  State lives in buffer 0 on entry;
// both buffers are left identical on the physical cells (the ring is completed by the
// halo update + copy that follow in evpk_prep).
// ------------------------------------------------------------------------------------
__global__ void k_prep2(Slab s, DevParams p) {
    SLAB_IJ_ALL
    const bool icet = plane(s, F_ICETM)[k] == 1.0;
    double *S0 = plane(s, F_STATE0), *S1 = plane(s, F_STATE1);
    double wx = 0.0, wy = 0.0, fx = 0.0, fy = 0.0, umdti = 0.0, vrelc = 0.0;      // :484-490
    if (p.revp == 1.0 || !icet) {                                                  // :492-518
#pragma unroll
        for (int c = S_SP; c < NSTATE; c++) S0[(size_t)c * s.fstride + k] = 0.0;
    }
    unsigned char cm = icet ? CM_T : 0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {                            // :545-577
        const double aiu = plane(s, F_AIU)[k], umass = plane(s, F_UMASS)[k];
        const double uocn = plane(s, F_UOCN)[k], vocn = plane(s, F_VOCN)[k];
        const bool old = s.iceumask[k] != 0;
        const bool ium = (s.umask[k] != 0) && (aiu > p.a_min) && (umass > p.m_min);
        s.iceumask[k] = ium ? 1 : 0;
        double u = S0[(size_t)S_U * s.fstride + k], v = S0[(size_t)S_V * s.fstride + k];
        if (ium) {
            if (!old) { u = uocn; v = vocn; }
            cm |= CM_U;
        } else {
            u = 0.0; v = 0.0;
            plane(s, F_STRINTX)[k] = 0.0; plane(s, F_STRINTY)[k] = 0.0;
            plane(s, F_STROCNX)[k] = 0.0; plane(s, F_STROCNY)[k] = 0.0;
        }
        S0[(size_t)S_U * s.fstride + k] = u;
        S0[(size_t)S_V * s.fstride + k] = v;
        plane(s, F_UVEL_INIT)[k] = u;
        plane(s, F_VVEL_INIT)[k] = v;
        if (ium) {                                                                 // :583-612
            umdti = umass / p.dt;
            const double fm = plane(s, F_FCOR)[k] * umass;
            plane(s, F_FM)[k] = fm;
            const double sg = copysign(1.0, fm);
            wx = uocn * p.cosw - vocn * p.sinw * sg;
            wy = vocn * p.cosw + uocn * p.sinw * sg;
            double tx, ty;
            if (p.tilt_from_slope) {
                tx = -p.gravit * umass * plane(s, F_SSTLTX)[k];
                ty = -p.gravit * umass * plane(s, F_SSTLTY)[k];
            } else {
                tx = -fm * vocn;
                ty = fm * uocn;
            }
            plane(s, F_STRTLTX)[k] = tx;
            plane(s, F_STRTLTY)[k] = ty;
            fx = plane(s, F_STRAIRX)[k] + tx;
            fy = plane(s, F_STRAIRY)[k] + ty;
            // stepu: vrel = aiu*rhow*Cw*sqrt(..) evaluates (aiu*rhow)*Cw first (ice_dyn_shared.F90:708)
            vrelc = aiu * p.rhow * plane(s, F_CW)[k];
        }
    }
    plane(s, F_WATERX)[k] = wx; plane(s, F_WATERY)[k] = wy;
    plane(s, F_FORCEX)[k] = fx; plane(s, F_FORCEY)[k] = fy;
    plane(s, F_UMASSDTI)[k] = umdti; plane(s, F_VRELC)[k] = vrelc;
    s.cmask[k] = cm;
#pragma unroll
    for (int c = 0; c < NSTATE; c++) S1[(size_t)c * s.fstride + k] = S0[(size_t)c * s.fstride + k];
}

// ------------------------------------------------------------------------------------
// ghost ring, single rank.  MPI-backend semantics: ghosts are `fill` where no neighbour
// exists (mpi/ice_boundary.F90:1409-1416).  N-S first (fill, or tripole fold from a
// packed copy of the two top rows), then E-W (cyclic wrap or fill) over all rows.
// ------------------------------------------------------------------------------------
__global__ void k_halo_ns_fill(Slab s, int f, int nf, double fill, int north_too) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > s.nxl + 1) return;
    for (int q = 0; q < nf; q++) {
        double *a = plane(s, f + q);
        a[cell(s, i, 0)] = fill;
        if (north_too) a[cell(s, i, s.nyl + 1)] = fill;
    }
}

// fold buffer layout: fb[(q*2 + r)*nxg + (g-1)], r = 0: row ny-1, r = 1: row ny; g global column
__global__ void k_fold_pack(Slab s, int f, int nf, double *fb, int gofs /* global col of local col 1, minus 1 */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i > s.nxl) return;
    for (int q = 0; q < nf; q++) {
        const double *a = plane(s, f + q);
        fb[((size_t)q * 2 + 0) * s.nxg + (gofs + i - 1)] = a[cell(s, i, s.nyl - 1)];
        fb[((size_t)q * 2 + 1) * s.nxg + (gofs + i - 1)] = a[cell(s, i, s.nyl)];
    }
}

// u-fold copy-out (serial/ice_boundary.F90:801-888, copy lists :3752-3776)
//   center  : ghost(i,ny+1) = sgn*B2(nx-g+1)
//   NEcorner: top(i,ny) = sgn*sym(B2)(nx-g), ghost(i,ny+1) = sgn*B1(nx-g), index 0 -> nx
//   stress  : (ice_HaloUpdate_stress) center rule, no sign, source plane differs from dest
__global__ void k_fold_apply(Slab s, int fdst, int nf, const double *fb, int necorner, double sgn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // local col 0..nxl+1
    if (i > s.nxl + 1) return;
    const int nx = s.nxg;
    int g = s.i0 + i - 1;                                       // global col, wrap
    if (g < 1) g += nx;
    if (g > nx) g -= nx;
    for (int q = 0; q < nf; q++) {
        const double *B1 = fb + ((size_t)q * 2 + 0) * nx - 1;  // 1-based
        const double *B2 = fb + ((size_t)q * 2 + 1) * nx - 1;
        double *a = plane(s, fdst + q);
        if (!necorner) {
            a[cell(s, i, s.nyl + 1)] = sgn * B2[nx - g + 1];
        } else {
            int src = nx - g;
            if (src == 0) src = nx;
            // symmetrised top row at column src (:818-824)
            double v;
            const int h = nx / 2;
            if (src >= 1 && src <= h - 1) {
                v = 0.5 * (B2[src] + sgn * B2[nx - src]);
            } else if (src >= h + 1 && src <= nx - 1) {
                const int ii = nx - src;
                v = sgn * (0.5 * (B2[ii] + sgn * B2[src]));
            } else {
                v = B2[src];
            }
            a[cell(s, i, s.nyl)] = sgn * v;
            a[cell(s, i, s.nyl + 1)] = sgn * B1[src];
        }
    }
}

__global__ void k_halo_ew_local(Slab s, int f, int nf, int cyclic, double fill) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > s.nyl + 1) return;
    for (int q = 0; q < nf; q++) {
        double *a = plane(s, f + q);
        a[cell(s, 0, j)] = cyclic ? a[cell(s, s.nxl, j)] : fill;
        a[cell(s, s.nxl + 1, j)] = cyclic ? a[cell(s, 1, j)] : fill;
    }
}

// multi-rank E-W exchange: pack the two physical edge columns / unpack into the ghost columns.
// buffer layout: buf[(q*rows + j)], rows = nyl+2
__global__ void k_ew_pack(Slab s, int f, int nf, double *sendW, double *sendE) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > s.nyl + 1) return;
    const int rows = s.nyl + 2;
    for (int q = 0; q < nf; q++) {
        const double *a = plane(s, f + q);
        sendW[(size_t)q * rows + j] = a[cell(s, 1, j)];
        sendE[(size_t)q * rows + j] = a[cell(s, s.nxl, j)];
    }
}

__global__ void k_ew_unpack(Slab s, int f, int nf, const double *recvW, const double *recvE, int haveW, int haveE, double fill) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > s.nyl + 1) return;
    const int rows = s.nyl + 2;
    for (int q = 0; q < nf; q++) {
        double *a = plane(s, f + q);
        a[cell(s, 0, j)] = haveW ? recvW[(size_t)q * rows + j] : fill;
        a[cell(s, s.nxl + 1, j)] = haveE ? recvE[(size_t)q * rows + j] : fill;
    }
}

// icetmask plane (double 0/1, after its halo update) -> cmask bit, then strip activity flags
__global__ void k_icetm_to_cmask(Slab s) {
    SLAB_IJ_ALL
    unsigned char cm = s.cmask[k] & CM_U;
    if (plane(s, F_ICETM)[k] == 1.0) cm |= CM_T;
    s.cmask[k] = cm;
}

// one wave per strip: is there any T work (cols cx*63+1..+64, rows jb..jb+R) or U work?
// Also counts active cells: T on physical cells, U.
__global__ void k_strip_flags(Slab s, int ncx, int nry, int R, unsigned char *flags, unsigned long long *counts) {
    const int sid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (sid >= ncx * nry) return;
    const int cx = sid % ncx, ry = sid / ncx;
    const int i = cx * STRIP_W + 1 + lane;
    const int jb = ry * R + 1;
    int any = 0, nt = 0, nu = 0;
    if (i <= s.nxl + 1) {
        for (int jj = 0; jj <= R; jj++) {
            const int j = jb + jj;
            if (j > s.nyl + 1) break;
            const unsigned char m = s.cmask[cell(s, i, j)];
            if (m & CM_T) {
                any = 1;
                if (lane < STRIP_W && jj < R && i <= s.nxl && j <= s.nyl) nt++;
            }
            if ((m & CM_U) && lane < STRIP_W && jj < R) { any = 1; nu++; }
        }
    }
    const unsigned long long b = __ballot(any);
    for (int o = 32; o > 0; o >>= 1) { nt += __shfl_down(nt, o); nu += __shfl_down(nu, o); }
    if (lane == 0) {
        flags[sid] = b ? 1 : 0;
        if (nt) atomicAdd(&counts[0], (unsigned long long)nt);
        if (nu) atomicAdd(&counts[1], (unsigned long long)nu);
    }
}

// ------------------------------------------------------------------------------------
// THE HOT KERNEL: one EVP subcycle, stress + stepu fused.
//
// One wave = one strip of 63 U columns x R U rows.  Lane l holds T column i = cx*63+1+l
// (64 T columns, the 64th is the redundant east neighbour) and marches north: at step j
// it computes the stress of T(i,j), hands the four west-going str terms to lane l-1 by a
// one-lane shuffle, and finishes U(i,j-1) from T(i,j-1), T(i+1,j-1), T(i,j), T(i+1,j).
// Rows and strips with no active cell are skipped wave-uniformly (wavefront predication
// on the ice mask); inactive lanes neither load nor store.
// ------------------------------------------------------------------------------------
struct SubArgs {
    Slab s;
    double ecci, arlx1i, denom1, brlx, revp, cosw, sinw;
    const int *strips;
    int nstrips, ncx, R, cur, wrap;
};

__device__ __forceinline__ double shfl_dn1(double x) { return __shfl_down(x, 1); }

template <bool LAST, bool REVP>
__global__ __launch_bounds__(256) void k_subcycle(SubArgs a) {
    const Slab &s = a.s;
    const int lane = threadIdx.x & 63;
    // XCD-aware order: consecutive strips of the list stay on one XCD (blocks are dealt
    // round-robin over the 8 XCDs), so neighbouring strips share an L2.
    // gridDim.x is a multiple of 8 (host rounds up), so this is a bijection on [0, gridDim.x).
    const int chunk = gridDim.x >> 3;
    const int wg = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int sid = wg * 4 + (threadIdx.x >> 6);
    if (sid >= a.nstrips) return;
    const int st = a.strips[sid];
    const int cx = st % a.ncx, ry = st / a.ncx;
    const int R = a.R;
    const int i = cx * STRIP_W + 1 + lane;            // T column of this lane
    const int jb = ry * R + 1;
    const bool colT = (i <= s.nxl + 1);               // lane has a T column
    const bool ownT = colT && (lane < STRIP_W);       // ... and owns its sigma stores
    const bool colU = (i <= s.nxl) && (lane < STRIP_W);

    const size_t fs = s.fstride;
    const double *Sr = plane(s, a.cur ? F_STATE1 : F_STATE0);
    double *Sw = plane(s, a.cur ? F_STATE0 : F_STATE1);
    const double *ur = Sr + (size_t)S_U * fs, *vr = Sr + (size_t)S_V * fs;
    const double *P = s.F;                            // plane(s,f)[k] = P[f*fs + k]

    const double ecci = a.ecci, arlx1i = a.arlx1i, denom1 = a.denom1;
    const double p111 = 1.0 / 9.0, p055 = p111 * 0.5, p027 = p055 * 0.5;
    const double p166 = 1.0 / 6.0, p222 = 2.0 / 9.0, p333 = 1.0 / 3.0;

    // carried from the previous row (j-1)
    double u_im = 0.0, u_mm = 0.0, v_im = 0.0, v_mm = 0.0;
    if (colT) {
        const size_t k0 = cell(s, i, jb - 1);
        u_im = ur[k0]; u_mm = ur[k0 - 1]; v_im = vr[k0]; v_mm = vr[k0 - 1];
    }
    double s1c = 0.0, s5c = 0.0, s2r = 0.0, s7r = 0.0;
    unsigned char mprev = 0;

    for (int jj = 0; jj <= R; jj++) {
        const int j = jb + jj;
        if (j > s.nyl + 1) break;
        const size_t k = cell(s, i, j);
        unsigned char m = 0;
        double u_ij = 0.0, u_mj = 0.0, v_ij = 0.0, v_mj = 0.0;
        if (colT) {
            m = s.cmask[k];
            u_ij = ur[k]; u_mj = ur[k - 1]; v_ij = vr[k]; v_mj = vr[k - 1];
        }
        const bool tact = (m & CM_T) != 0;
        double str1 = 0.0, str2 = 0.0, str3 = 0.0, str4 = 0.0, str5 = 0.0, str6 = 0.0, str7 = 0.0, str8 = 0.0;

        if (__any(tact)) {
            if (tact) {
                const double cyp = P[(size_t)F_CYP * fs + k], cxp = P[(size_t)F_CXP * fs + k];
                const double cym = P[(size_t)F_CYM * fs + k], cxm = P[(size_t)F_CXM * fs + k];
                const double dxt = P[(size_t)F_DXT * fs + k], dyt = P[(size_t)F_DYT * fs + k];
                const double dxhy = P[(size_t)F_DXHY * fs + k], dyhx = P[(size_t)F_DYHX * fs + k];
                const double tiny = P[(size_t)F_TINYAREA * fs + k], strength = P[(size_t)F_STRENGTH * fs + k];
                double sp1 = Sr[(size_t)(S_SP + 0) * fs + k], sp2 = Sr[(size_t)(S_SP + 1) * fs + k];
                double sp3 = Sr[(size_t)(S_SP + 2) * fs + k], sp4 = Sr[(size_t)(S_SP + 3) * fs + k];
                double sm1 = Sr[(size_t)(S_SM + 0) * fs + k], sm2 = Sr[(size_t)(S_SM + 1) * fs + k];
                double sm3 = Sr[(size_t)(S_SM + 2) * fs + k], sm4 = Sr[(size_t)(S_SM + 3) * fs + k];
                double s121 = Sr[(size_t)(S_S12 + 0) * fs + k], s122 = Sr[(size_t)(S_S12 + 1) * fs + k];
                double s123 = Sr[(size_t)(S_S12 + 2) * fs + k], s124 = Sr[(size_t)(S_S12 + 3) * fs + k];

                // strain rates * area (ice_dyn_evp.F90:627-654)
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

                // Delta (:657-660)
                const double Deltane = sqrt(divune * divune + ecci * (tensionne * tensionne + shearne * shearne));
                const double Deltanw = sqrt(divunw * divunw + ecci * (tensionnw * tensionnw + shearnw * shearnw));
                const double Deltase = sqrt(divuse * divuse + ecci * (tensionse * tensionse + shearse * shearse));
                const double Deltasw = sqrt(divusw * divusw + ecci * (tensionsw * tensionsw + shearsw * shearsw));

                const bool store = ownT && (jj < R);
                if (LAST) {                                                         // :665-677
                    if (store) {
                        const double tarear = P[(size_t)F_TAREAR * fs + k];
                        const double divu = 0.25 * (divune + divunw + divuse + divusw) * tarear;
                        const double tmp = 0.25 * (Deltane + Deltanw + Deltase + Deltasw) * tarear;
                        s.F[(size_t)F_DIVU * fs + k] = divu;
                        s.F[(size_t)F_RDGCONV * fs + k] = -fmin(divu, 0.0);
                        s.F[(size_t)F_RDGSHEAR * fs + k] = 0.5 * (tmp - fabs(divu));
                        const double ts = tensionne + tensionnw + tensionse + tensionsw;
                        const double ss = shearne + shearnw + shearse + shearsw;
                        s.F[(size_t)F_SHEAR * fs + k] = 0.25 * tarear * sqrt(ts * ts + ss * ss);
                    }
                }

                // replacement pressure / Delta (:683-697)
                double c0ne = strength / fmax(Deltane, tiny);
                double c0nw = strength / fmax(Deltanw, tiny);
                double c0sw = strength / fmax(Deltasw, tiny);
                double c0se = strength / fmax(Deltase, tiny);
                if (LAST) { if (store) s.F[(size_t)F_PRSSIG * fs + k] = c0ne * Deltane; }
                const double c1ne = c0ne * arlx1i, c1nw = c0nw * arlx1i, c1sw = c0sw * arlx1i, c1se = c0se * arlx1i;
                c0ne = c1ne * ecci; c0nw = c1nw * ecci; c0sw = c1sw * ecci; c0se = c1se * ecci;

                // the stresses (:704-721)
                sp1 = (sp1 + c1ne * (divune - Deltane)) * denom1;
                sp2 = (sp2 + c1nw * (divunw - Deltanw)) * denom1;
                sp3 = (sp3 + c1sw * (divusw - Deltasw)) * denom1;
                sp4 = (sp4 + c1se * (divuse - Deltase)) * denom1;
                sm1 = (sm1 + c0ne * tensionne) * denom1;
                sm2 = (sm2 + c0nw * tensionnw) * denom1;
                sm3 = (sm3 + c0sw * tensionsw) * denom1;
                sm4 = (sm4 + c0se * tensionse) * denom1;
                s121 = (s121 + c0ne * shearne * 0.5) * denom1;
                s122 = (s122 + c0nw * shearnw * 0.5) * denom1;
                s123 = (s123 + c0sw * shearsw * 0.5) * denom1;
                s124 = (s124 + c0se * shearse * 0.5) * denom1;

                if (store) {
                    Sw[(size_t)(S_SP + 0) * fs + k] = sp1; Sw[(size_t)(S_SP + 1) * fs + k] = sp2;
                    Sw[(size_t)(S_SP + 2) * fs + k] = sp3; Sw[(size_t)(S_SP + 3) * fs + k] = sp4;
                    Sw[(size_t)(S_SM + 0) * fs + k] = sm1; Sw[(size_t)(S_SM + 1) * fs + k] = sm2;
                    Sw[(size_t)(S_SM + 2) * fs + k] = sm3; Sw[(size_t)(S_SM + 3) * fs + k] = sm4;
                    Sw[(size_t)(S_S12 + 0) * fs + k] = s121; Sw[(size_t)(S_S12 + 1) * fs + k] = s122;
                    Sw[(size_t)(S_S12 + 2) * fs + k] = s123; Sw[(size_t)(S_S12 + 3) * fs + k] = s124;
                }

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
                str1 = -strp_tmp - strm_tmp - str12ew + dxhy * (-csigpne + csigmne) + dyhx * csig12ne;
                str2 = strp_tmp + strm_tmp - str12we + dxhy * (-csigpnw + csigmnw) + dyhx * csig12nw;
                strp_tmp = 0.25 * dyt * (p333 * ssigps + p166 * ssigpn);
                strm_tmp = 0.25 * dyt * (p333 * ssigms + p166 * ssigmn);
                str3 = -strp_tmp - strm_tmp + str12ew + dxhy * (-csigpse + csigmse) + dyhx * csig12se;
                str4 = strp_tmp + strm_tmp + str12we + dxhy * (-csigpsw + csigmsw) + dyhx * csig12sw;
                // dF/dy (:825-845)
                strp_tmp = 0.25 * dxt * (p333 * ssigpe + p166 * ssigpw);
                strm_tmp = 0.25 * dxt * (p333 * ssigme + p166 * ssigmw);
                str5 = -strp_tmp + strm_tmp - str12ns - dyhx * (csigpne + csigmne) + dxhy * csig12ne;
                str6 = strp_tmp - strm_tmp - str12sn - dyhx * (csigpse + csigmse) + dxhy * csig12se;
                strp_tmp = 0.25 * dxt * (p333 * ssigpw + p166 * ssigpe);
                strm_tmp = 0.25 * dxt * (p333 * ssigmw + p166 * ssigme);
                str7 = -strp_tmp + strm_tmp + str12ns - dyhx * (csigpnw + csigmnw) + dxhy * csig12nw;
                str8 = strp_tmp - strm_tmp + str12sn - dyhx * (csigpsw + csigmsw) + dxhy * csig12sw;
            }
        }

        // east neighbour's contributions of this T row
        const double s2n = shfl_dn1(str2), s4n = shfl_dn1(str4), s7n = shfl_dn1(str7), s8n = shfl_dn1(str8);

        // stepu for U(i, j-1)  (ice_dyn_shared.F90:700-746)
        if (jj >= 1) {
            const bool uact = colU && ((mprev & CM_U) != 0);
            if (__any(uact)) {
                if (uact) {
                    const size_t ku = k - s.pitch;
                    const double uold = u_im, vold = v_im;
                    const double vrelc = P[(size_t)F_VRELC * fs + ku];
                    const double uocn = P[(size_t)F_UOCN * fs + ku], vocn = P[(size_t)F_VOCN * fs + ku];
                    const double waterx = P[(size_t)F_WATERX * fs + ku], watery = P[(size_t)F_WATERY * fs + ku];
                    const double forcex = P[(size_t)F_FORCEX * fs + ku], forcey = P[(size_t)F_FORCEY * fs + ku];
                    const double umassdti = P[(size_t)F_UMASSDTI * fs + ku], fm = P[(size_t)F_FM * fs + ku];
                    const double uarear = P[(size_t)F_UAREAR * fs + ku];
                    const double du = uocn - uold, dv = vocn - vold;
                    const double vrel = vrelc * sqrt(du * du + dv * dv);            // :708-709
                    const double taux = vrel * waterx, tauy = vrel * watery;        // :711-712
                    const double cca = (a.brlx + a.revp) * umassdti + vrel * a.cosw;   // :715
                    const double ccb = fm + copysign(1.0, fm) * vrel * a.sinw;      // :720
                    const double ab2 = cca * cca + ccb * ccb;
                    const double strintx = uarear * (((s1c + s2r) + str3) + s4n);   // :725-728
                    const double strinty = uarear * (((s5c + str6) + s7r) + s8n);
                    double ui = 0.0, vi = 0.0;
                    if (REVP) { ui = P[(size_t)F_UVEL_INIT * fs + ku]; vi = P[(size_t)F_VVEL_INIT * fs + ku]; }
                    const double cc1 = strintx + forcex + taux + umassdti * (a.brlx * uold + a.revp * ui);   // :731-734
                    const double cc2 = strinty + forcey + tauy + umassdti * (a.brlx * vold + a.revp * vi);
                    const double un = (cca * cc1 + ccb * cc2) / ab2;                // :736-737
                    const double vn = (cca * cc2 - ccb * cc1) / ab2;
                    Sw[(size_t)S_U * fs + ku] = un;
                    Sw[(size_t)S_V * fs + ku] = vn;
                    if (a.wrap) {   // single-rank cyclic E-W: the owner also writes the ghost image
                        if (i == 1) { Sw[(size_t)S_U * fs + ku + s.nxl] = un; Sw[(size_t)S_V * fs + ku + s.nxl] = vn; }
                        if (i == s.nxl) { Sw[(size_t)S_U * fs + ku - s.nxl] = un; Sw[(size_t)S_V * fs + ku - s.nxl] = vn; }
                    }
                    if (LAST) {
                        s.F[(size_t)F_STRINTX * fs + ku] = strintx;
                        s.F[(size_t)F_STRINTY * fs + ku] = strinty;
                    }
                }
            }
        }
        // carry
        s1c = str1; s5c = str5; s2r = s2n; s7r = s7n;
        u_im = u_ij; u_mm = u_mj; v_im = v_ij; v_mm = v_mj;
        mprev = m;
    }
}

template __global__ void k_subcycle<false, false>(SubArgs);
template __global__ void k_subcycle<true, false>(SubArgs);
template __global__ void k_subcycle<false, true>(SubArgs);
template __global__ void k_subcycle<true, true>(SubArgs);

// ------------------------------------------------------------------------------------
// evp_finish (ice_dyn_shared.F90:757-844)
// ------------------------------------------------------------------------------------
__global__ void k_finish(Slab s, DevParams p, int cur) {
    SLAB_IJ_ALL
    double xT = 0.0, yT = 0.0;                                                     // :806-811
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl && s.iceumask[k]) {
        const double *S = plane(s, cur ? F_STATE1 : F_STATE0);
        const double u = S[(size_t)S_U * s.fstride + k], v = S[(size_t)S_V * s.fstride + k];
        const double du = plane(s, F_UOCN)[k] - u, dv = plane(s, F_VOCN)[k] - v;
        const double aiu = plane(s, F_AIU)[k], fm = plane(s, F_FM)[k];
        double vrel = p.rhow * plane(s, F_CW)[k] * sqrt(du * du + dv * dv);        // :818-819
        vrel = vrel * aiu;                                                         // :827
        const double sg = copysign(1.0, fm);
        const double sx = vrel * (du * p.cosw - dv * p.sinw * sg);                 // :828-831
        const double sy = vrel * (dv * p.cosw + du * p.sinw * sg);
        plane(s, F_STROCNX)[k] = sx;
        plane(s, F_STROCNY)[k] = sy;
        xT = sx / aiu;                                                             // :840-841
        yT = sy / aiu;
    }
    plane(s, F_STROCNXT)[k] = xT;
    plane(s, F_STROCNYT)[k] = yT;
}

}  // namespace evpk
