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
// All hot fields are read and written as double2 pairs (16 B per lane, see evpk_internal.h).
#include "evpk_internal.h"

namespace evpk {

// ------------------------------------------------------------------------------------
// gather / scatter between the reference's block layout and the slab
// ------------------------------------------------------------------------------------
enum { MODE_PHYS = 0, MODE_ALL = 1, MODE_NE = 2, MODE_NE_FOLD = 3 };

// Which block cells feed the slab: every physical cell, plus ghost cells that land on the slab's
// ghost ring -- but only from the block whose own columns (rows) the ring cell continues, so a
// neighbouring block's corner ghost (which the reference may leave stale) never competes.
__device__ __forceinline__ bool gather_take(const Slab &s, const BlockDesc &d, int i, int j, int si, int sj) {
    const bool owncol = (i >= d.ilo && i <= d.ihi), ownrow = (j >= d.jlo && j <= d.jhi);
    if (owncol && ownrow) return true;
    if (i > d.ihi + 1 || j > d.jhi + 1) return false;                       // padding
    const bool colok = owncol || si == 0 || si == s.nxl + 1;
    const bool rowok = ownrow || sj == 0 || sj == s.nyl + 1;
    const bool ring = (si == 0 || si == s.nxl + 1 || sj == 0 || sj == s.nyl + 1);
    return ring && colok && rowok;
}

// double fields: destination / source is field f of the pair-interleaved slab
__global__ void k_gather_f(Slab s, const BlockDesc *bd, int nxb, int nyb, const double *src, int f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;   // 1-based block column
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    if (!gather_take(s, d, i, j, si, sj)) return;
    FD(s, f, cell(s, si, sj)) = src[((size_t)b * nyb + (j - 1)) * nxb + (i - 1)];
}

__global__ void k_gather_m(Slab s, const BlockDesc *bd, int nxb, int nyb, const int32_t *src, int32_t *dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    if (!gather_take(s, d, i, j, si, sj)) return;
    dst[mcell(s, si, sj)] = src[((size_t)b * nyb + (j - 1)) * nxb + (i - 1)];
}

__device__ __forceinline__ bool scatter_take(const Slab &s, const BlockDesc &d, int i, int j, int mode, int &si, int &sj) {
    si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return false;
    if (i > d.ihi + 1 || j > d.jhi + 1) return false;   // padding
    if (mode == MODE_ALL) return true;
    if (mode == MODE_NE_FOLD && sj == s.nyl + 1) return true;              // ice_HaloUpdate_stress writes the whole north ghost row
    if (mode == MODE_NE || mode == MODE_NE_FOLD) return (i >= d.ilo && j >= d.jlo);
    return (i >= d.ilo && i <= d.ihi && j >= d.jlo && j <= d.jhi);
}

__global__ void k_scatter_f(Slab s, const BlockDesc *bd, int nxb, int nyb, int f, double *dst, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    int si, sj;
    if (!scatter_take(s, bd[b], i, j, mode, si, sj)) return;
    dst[((size_t)b * nyb + (j - 1)) * nxb + (i - 1)] = FD(s, f, cell(s, si, sj));
}

__global__ void k_scatter_m(Slab s, const BlockDesc *bd, int nxb, int nyb, const int32_t *src, int32_t *dst, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    int si, sj;
    if (!scatter_take(s, bd[b], i, j, mode, si, sj)) return;
    dst[((size_t)b * nyb + (j - 1)) * nxb + (i - 1)] = src[mcell(s, si, sj)];
}

__global__ void k_fill_plane(Slab s, int f, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > s.nxl + 1 || j > s.nyl + 1) return;
    FD(s, f, cell(s, i, j)) = v;
}

// all cells of the slab incl. ring: thread (i,j), i = 0..nxl+1, j = 0..nyl+1
#define SLAB_IJ_ALL                                            \
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       \
    const int j = blockIdx.y * blockDim.y + threadIdx.y;       \
    if (i > s.nxl + 1 || j > s.nyl + 1) return;                \
    const size_t k = cell(s, i, j);                            \
    const size_t km = mcell(s, i, j);                          \
    (void)km;

// ------------------------------------------------------------------------------------
// evp_prep1 (ice_dyn_shared.F90:270-365) on the slab
// ------------------------------------------------------------------------------------
// `fresh` = the state planes were just uploaded from the host (anything may be non-zero anywhere).
// Otherwise cells that were inactive at the previous prep still hold their zeros and are skipped.
__global__ void k_prep1a(Slab s, DevParams p, int fresh) {
    SLAB_IJ_ALL
    const double vice = FD(s, F_VICE, k), vsno = FD(s, F_VSNO, k), aice = FD(s, F_AICE, k);
    const bool tm = s.tmask[km] != 0;
    double tmass = 0.0;
    if (tm) tmass = (p.rhoi * vice + p.rhos * vsno);                              // :322-326
    FD(s, F_TMASS, k) = tmass;
    s.tmphm[km] = (tm && (aice > p.a_min) && (tmass > p.m_min)) ? 1 : 0;           // :331-332
    // :339-340 strairx = strairxT; the T->U average that follows (t2ugrid_vector) reads it from the
    // work planes, so the copy lands there directly; U-grid wind (ACCESS) goes straight to strairx/y
    const int wx = p.wind_on_ugrid ? F_STRAIRX : F_WORK1, wy = p.wind_on_ugrid ? F_STRAIRY : F_WORK2;
    FD(s, wx, k) = FD(s, F_STRAIRXT, k);
    FD(s, wy, k) = FD(s, F_STRAIRYT, k);
    // evp(): zero the diagnostics (ice_dyn_evp.F90:174-182); only T cells active last time can be non-zero
    if (fresh || (s.cmask[km] & CM_T)) {
        FD(s, F_RDGCONV, k) = 0.0; FD(s, F_RDGSHEAR, k) = 0.0; FD(s, F_DIVU, k) = 0.0;
        FD(s, F_SHEAR, k) = 0.0; FD(s, F_PRSSIG, k) = 0.0;
    }
}

__global__ void k_prep1b(Slab s) {
    SLAB_IJ_ALL
    double m = 0.0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {                           // :350-363
        const unsigned char *t = s.tmphm;
        bool any = t[mcell(s, i - 1, j + 1)] | t[mcell(s, i, j + 1)] | t[mcell(s, i + 1, j + 1)] |
                   t[mcell(s, i - 1, j)] | t[km] | t[mcell(s, i + 1, j)] |
                   t[mcell(s, i - 1, j - 1)] | t[mcell(s, i, j - 1)] | t[mcell(s, i + 1, j - 1)];
        if (any) m = 1.0;
        if (!s.tmask[km]) m = 0.0;
    }
    FD(s, F_ICETM, k) = m;
}

// to_ugrid (ice_grid.F90:1834-1878): dst = 0 outside the physical cells
__global__ void k_to_ugrid(Slab s, int fsrc, int fdst) {
    SLAB_IJ_ALL
    double r = 0.0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {
        const size_t ke = cell(s, i + 1, j), kn = cell(s, i, j + 1), kne = cell(s, i + 1, j + 1);
#define W_(q) FD(s, fsrc, q)
#define TA_(q) FD(s, F_TAREA, q)
        r = 0.25 * (((W_(k) * TA_(k) + W_(ke) * TA_(ke)) + W_(kn) * TA_(kn)) + W_(kne) * TA_(kne)) / FD(s, F_UAREA, k);
#undef W_
#undef TA_
    }
    FD(s, fdst, k) = r;
}

// the four T->U averages of evp() in one pass: umass <- tmass, aiu <- aice_init (ice_dyn_evp.F90:218-219) and,
// unless the wind is already on the U grid, strairx/y <- work1/2 (t2ugrid_vector, :240-241)
__global__ void k_to_ugrid4(Slab s, int wind) {
    SLAB_IJ_ALL
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {
        const size_t ke = cell(s, i + 1, j), kn = cell(s, i, j + 1), kne = cell(s, i + 1, j + 1);
        const double t0 = FD(s, F_TAREA, k), t1 = FD(s, F_TAREA, ke), t2 = FD(s, F_TAREA, kn), t3 = FD(s, F_TAREA, kne);
        const double ua = FD(s, F_UAREA, k);
#define UG_(f) (0.25 * (((FD(s, f, k) * t0 + FD(s, f, ke) * t1) + FD(s, f, kn) * t2) + FD(s, f, kne) * t3) / ua)
        r0 = UG_(F_TMASS);
        r1 = UG_(F_AICE_INIT);
        if (wind) { r2 = UG_(F_WORK1); r3 = UG_(F_WORK2); }
#undef UG_
    }
    FD(s, F_UMASS, k) = r0;
    FD(s, F_AIU, k) = r1;
    if (wind) { FD(s, F_STRAIRX, k) = r2; FD(s, F_STRAIRY, k) = r3; }
}

// the two U->T averages of u2tgrid_vector (ice_dyn_evp.F90:505-506): strocnxT/yT <- work1/2, physical cells
__global__ void k_to_tgrid2(Slab s) {
    SLAB_IJ_ALL
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {
        const size_t kw = cell(s, i - 1, j), ks = cell(s, i, j - 1), ksw = cell(s, i - 1, j - 1);
        const double u0 = FD(s, F_UAREA, k), u1 = FD(s, F_UAREA, kw), u2 = FD(s, F_UAREA, ks), u3 = FD(s, F_UAREA, ksw);
        const double ta = FD(s, F_TAREA, k);
#define TG_(f) (0.25 * (((FD(s, f, k) * u0 + FD(s, f, kw) * u1) + FD(s, f, ks) * u2) + FD(s, f, ksw) * u3) / ta)
        FD(s, F_STROCNXT, k) = TG_(F_WORK1);
        FD(s, F_STROCNYT, k) = TG_(F_WORK2);
#undef TG_
    }
}

// ghost ring of nf planes: dst <- src (after a halo update of src)
__global__ void k_ring_copy(Slab s, int fsrc, int fdst, int nf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nrow = s.nyl + 2, ncol = s.nxl + 2;
    int i, j;
    if (t < 2 * ncol) { i = t % ncol; j = (t < ncol) ? 0 : s.nyl + 1; }
    else if (t < 2 * ncol + 2 * nrow) { const int q = t - 2 * ncol; j = q % nrow; i = (q < nrow) ? 0 : s.nxl + 1; }
    else return;
    const size_t k = cell(s, i, j);
    for (int q = 0; q < nf; q++) FD(s, fdst + q, k) = FD(s, fsrc + q, k);
}

// one full row (all columns incl. ghosts) of nf planes: dst <- src
__global__ void k_row_copy(Slab s, int fsrc, int fdst, int nf, int j) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > s.nxl + 1) return;
    const size_t k = cell(s, i, j);
    for (int q = 0; q < nf; q++) FD(s, fdst + q, k) = FD(s, fsrc + q, k);
}

// to_tgrid (ice_grid.F90:1924-1958): only physical cells of dst are written
__global__ void k_to_tgrid(Slab s, int fsrc, int fdst) {
    SLAB_IJ_ALL
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {
        const size_t kw = cell(s, i - 1, j), ks = cell(s, i, j - 1), ksw = cell(s, i - 1, j - 1);
#define W_(q) FD(s, fsrc, q)
#define UA_(q) FD(s, F_UAREA, q)
        FD(s, fdst, k) = 0.25 * (((W_(k) * UA_(k) + W_(kw) * UA_(kw)) + W_(ks) * UA_(ks)) + W_(ksw) * UA_(ksw)) / FD(s, F_TAREA, k);
#undef W_
#undef UA_
    }
}

// profiling aid: copy one pair plane with the hot kernel's access shape (16 B per lane, coalesced);
// moves exactly (nxl+2)*(nyl+2)*16 bytes each way -- a known byte count to calibrate FETCH_SIZE / WRITE_SIZE
__global__ void k_calib_copy_pair(Slab s, int fsrc_even, int fdst_even) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    if (i > s.nxl + 1 || j > s.nyl + 1) return;
    const size_t k = cell(s, i, j);
    const double2 *src = reinterpret_cast<const double2 *>(s.F) + (size_t)(fsrc_even >> 1) * s.pitch + k;
    double2 *dst = reinterpret_cast<double2 *>(s.F) + (size_t)(fdst_even >> 1) * s.pitch + k;
    *dst = *src;
}

__global__ void k_copy_plane(Slab s, int fsrc, int fdst) {
    SLAB_IJ_ALL
    FD(s, fdst, k) = FD(s, fsrc, k);
}

// ------------------------------------------------------------------------------------
// evp_prep2 (ice_dyn_shared.F90:377-614) on the slab.  State lives in buffer 0 on entry; both
// buffers are left identical (the velocity ring is completed by the halo update + ring copy that
// follow in evpk_prep).  Invariant kept for the subcycle kernel, which never touches inactive
// cells: sigma = 0 in both buffers where icetmask == 0, u = v = 0 in both where iceumask is false,
// and the stepu input planes are 0 where iceumask is false.  Unless `fresh`, a cell that was
// inactive at the previous prep already satisfies this and is skipped.
// ------------------------------------------------------------------------------------
__global__ void k_prep2(Slab s, DevParams p, int fresh) {
    SLAB_IJ_ALL
    const bool icet = FD(s, F_ICETM, k) == 1.0;          // after its halo update
    const unsigned char cmold = s.cmask[km];
    const bool prevT = fresh || (cmold & CM_T), prevU = fresh || (cmold & CM_U);
    if (p.revp == 1.0 || !icet) {                                                  // :492-518
        if (icet || prevT) {
#pragma unroll
            for (int c = S_SP; c < NSTATE; c++) { FD(s, F_STATE0 + c, k) = 0.0; FD(s, F_STATE1 + c, k) = 0.0; }
        }
    } else {
#pragma unroll
        for (int c = S_SP; c < NSTATE; c++) FD(s, F_STATE1 + c, k) = FD(s, F_STATE0 + c, k);
    }
    unsigned char cm = icet ? CM_T : 0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {                            // :545-577
        const double aiu = FD(s, F_AIU, k), umass = FD(s, F_UMASS, k);
        const bool old = s.iceumask[km] != 0;
        const bool ium = (s.umask[km] != 0) && (aiu > p.a_min) && (umass > p.m_min);
        s.iceumask[km] = ium ? 1 : 0;
        if (ium) {
            const double uocn = FD(s, F_UOCN, k), vocn = FD(s, F_VOCN, k);
            double u = FD(s, F_STATE0 + S_U, k), v = FD(s, F_STATE0 + S_V, k);
            if (!old) { u = uocn; v = vocn; }
            cm |= CM_U;
            FD(s, F_STATE0 + S_U, k) = u; FD(s, F_STATE0 + S_V, k) = v;
            FD(s, F_STATE1 + S_U, k) = u; FD(s, F_STATE1 + S_V, k) = v;
            FD(s, F_UVEL_INIT, k) = u;    FD(s, F_VVEL_INIT, k) = v;
            const double umdti = umass / p.dt;                                     // :583-612
            const double fm = FD(s, F_FCOR, k) * umass;
            FD(s, F_FM, k) = fm;
            const double sg = copysign(1.0, fm);
            const double wx = uocn * p.cosw - vocn * p.sinw * sg;
            const double wy = vocn * p.cosw + uocn * p.sinw * sg;
            double tx, ty;
            if (p.tilt_from_slope) {
                tx = -p.gravit * umass * FD(s, F_SSTLTX, k);
                ty = -p.gravit * umass * FD(s, F_SSTLTY, k);
            } else {
                tx = -fm * vocn;
                ty = fm * uocn;
            }
            FD(s, F_STRTLTX, k) = tx;
            FD(s, F_STRTLTY, k) = ty;
            FD(s, F_WATERX, k) = wx; FD(s, F_WATERY, k) = wy;
            FD(s, F_FORCEX, k) = FD(s, F_STRAIRX, k) + tx;
            FD(s, F_FORCEY, k) = FD(s, F_STRAIRY, k) + ty;
            FD(s, F_UMASSDTI, k) = umdti;
            // stepu: vrel = aiu*rhow*Cw*sqrt(..) evaluates (aiu*rhow)*Cw first (ice_dyn_shared.F90:708)
            FD(s, F_VRELC, k) = aiu * p.rhow * FD(s, F_CW, k);
        } else if (prevU || old) {
            FD(s, F_STATE0 + S_U, k) = 0.0; FD(s, F_STATE0 + S_V, k) = 0.0;
            FD(s, F_STATE1 + S_U, k) = 0.0; FD(s, F_STATE1 + S_V, k) = 0.0;
            FD(s, F_UVEL_INIT, k) = 0.0;    FD(s, F_VVEL_INIT, k) = 0.0;
            FD(s, F_STRINTX, k) = 0.0; FD(s, F_STRINTY, k) = 0.0;
            FD(s, F_STROCNX, k) = 0.0; FD(s, F_STROCNY, k) = 0.0;
            FD(s, F_WATERX, k) = 0.0; FD(s, F_WATERY, k) = 0.0;
            FD(s, F_FORCEX, k) = 0.0; FD(s, F_FORCEY, k) = 0.0;
            FD(s, F_UMASSDTI, k) = 0.0; FD(s, F_VRELC, k) = 0.0;
        }
    }
    s.cmask[km] = cm;
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
        FD(s, f + q, cell(s, i, 0)) = fill;
        if (north_too) FD(s, f + q, cell(s, i, s.nyl + 1)) = fill;
    }
}

// fold buffer layout: fb[(q*2 + r)*nxg + (g-1)], r = 0: row ny-1, r = 1: row ny; g global column
__global__ void k_fold_pack(Slab s, int f, int nf, double *fb, int gofs /* global col of local col 1, minus 1 */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i > s.nxl) return;
    for (int q = 0; q < nf; q++) {
        fb[((size_t)q * 2 + 0) * s.nxg + (gofs + i - 1)] = FD(s, f + q, cell(s, i, s.nyl - 1));
        fb[((size_t)q * 2 + 1) * s.nxg + (gofs + i - 1)] = FD(s, f + q, cell(s, i, s.nyl));
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
        if (!necorner) {
            FD(s, fdst + q, cell(s, i, s.nyl + 1)) = sgn * B2[nx - g + 1];
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
            FD(s, fdst + q, cell(s, i, s.nyl)) = sgn * v;
            FD(s, fdst + q, cell(s, i, s.nyl + 1)) = sgn * B1[src];
        }
    }
}

__global__ void k_halo_ew_local(Slab s, int f, int nf, int cyclic, double fill) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > s.nyl + 1) return;
    for (int q = 0; q < nf; q++) {
        FD(s, f + q, cell(s, 0, j)) = cyclic ? FD(s, f + q, cell(s, s.nxl, j)) : fill;
        FD(s, f + q, cell(s, s.nxl + 1, j)) = cyclic ? FD(s, f + q, cell(s, 1, j)) : fill;
    }
}

// multi-rank E-W exchange: pack the two physical edge columns / unpack into the ghost columns.
// buffer layout: buf[(q*rows + j)], rows = nyl+2
__global__ void k_ew_pack(Slab s, int f, int nf, double *sendW, double *sendE) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > s.nyl + 1) return;
    const int rows = s.nyl + 2;
    for (int q = 0; q < nf; q++) {
        sendW[(size_t)q * rows + j] = FD(s, f + q, cell(s, 1, j));
        sendE[(size_t)q * rows + j] = FD(s, f + q, cell(s, s.nxl, j));
    }
}

__global__ void k_ew_unpack(Slab s, int f, int nf, const double *recvW, const double *recvE, int haveW, int haveE, double fill) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > s.nyl + 1) return;
    const int rows = s.nyl + 2;
    for (int q = 0; q < nf; q++) {
        FD(s, f + q, cell(s, 0, j)) = haveW ? recvW[(size_t)q * rows + j] : fill;
        FD(s, f + q, cell(s, s.nxl + 1, j)) = haveE ? recvE[(size_t)q * rows + j] : fill;
    }
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
            const unsigned char m = s.cmask[mcell(s, i, j)];
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

// pair-plane access with a wave-uniform row base (SGPR) and a 32-bit lane offset (VGPR):
// rb = byte address of (row j, pair plane 0, column 0); pp = pitch in bytes of one pair plane.
__device__ __forceinline__ double2 ldp(const char *rb, size_t pp, int f_even, unsigned lo) {
    return *reinterpret_cast<const double2 *>(rb + (size_t)(f_even >> 1) * pp + lo);
}
__device__ __forceinline__ void stp(char *rb, size_t pp, int f_even, unsigned lo, double x, double y) {
    *reinterpret_cast<double2 *>(rb + (size_t)(f_even >> 1) * pp + lo) = make_double2(x, y);
}
__device__ __forceinline__ void st1(char *rb, size_t pp, int f, unsigned lo, double x) {
    *reinterpret_cast<double *>(rb + (size_t)(f >> 1) * pp + lo + (f & 1) * 8) = x;
}

template <bool LAST, bool REVP>
__global__ __launch_bounds__(256) void k_subcycle(SubArgs a) {
    const Slab &s = a.s;
    const int lane = threadIdx.x & 63;
    // XCD-aware order: consecutive strips of the list stay on one XCD (blocks are dealt
    // round-robin over the 8 XCDs), so neighbouring strips share an L2.
    // gridDim.x is a multiple of 8 (host rounds up), so this is a bijection on [0, gridDim.x).
    const int chunk = gridDim.x >> 3;
    const int wg = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int sid = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
    if (sid >= a.nstrips) return;
    const int st = __builtin_amdgcn_readfirstlane(a.strips[sid]);
    const int cx = st % a.ncx, ry = st / a.ncx;
    const int R = a.R;
    const int i = cx * STRIP_W + 1 + lane;            // T column of this lane
    const int jb = ry * R + 1;
    const bool colT = (i <= s.nxl + 1);               // lane has a T column
    const bool ownT = colT && (lane < STRIP_W);       // ... and owns its sigma stores
    const bool colU = (i <= s.nxl) && (lane < STRIP_W);

    const size_t pp = (size_t)s.pitch * 16;           // bytes per row of one pair plane
    const size_t rowb = (size_t)s.rstride * 16;       // bytes per row of all planes
    const unsigned lo = (unsigned)(C0 + i) * 16u;     // lane byte offset inside a pair-plane row
    const int SR = a.cur ? F_STATE1 : F_STATE0;       // read buffer
    const int SW = a.cur ? F_STATE0 : F_STATE1;       // write buffer
    char *const base = reinterpret_cast<char *>(s.F);

    const double ecci = a.ecci, arlx1i = a.arlx1i, denom1 = a.denom1;
    const double p111 = 1.0 / 9.0, p055 = p111 * 0.5, p027 = p055 * 0.5;
    const double p166 = 1.0 / 6.0, p222 = 2.0 / 9.0, p333 = 1.0 / 3.0;

    // carried from the previous row (j-1)
    double u_im = 0.0, u_mm = 0.0, v_im = 0.0, v_mm = 0.0;
    if (colT) {
        const char *rb0 = base + (size_t)(jb - 1) * rowb;
        const double2 a0 = ldp(rb0, pp, SR + S_U, lo), a1 = ldp(rb0, pp, SR + S_U, lo - 16u);
        u_im = a0.x; v_im = a0.y; u_mm = a1.x; v_mm = a1.y;
    }
    double s1c = 0.0, s5c = 0.0, s2r = 0.0, s7r = 0.0;
    unsigned char mprev = 0;

    for (int jj = 0; jj <= R; jj++) {
        const int j = jb + jj;
        if (j > s.nyl + 1) break;
        char *const rb = base + (size_t)j * rowb;     // wave-uniform
        unsigned char m = 0;
        double u_ij = 0.0, u_mj = 0.0, v_ij = 0.0, v_mj = 0.0;
        if (colT) {
            m = s.cmask[(size_t)j * s.pitch + C0 + i];
            const double2 a0 = ldp(rb, pp, SR + S_U, lo), a1 = ldp(rb, pp, SR + S_U, lo - 16u);
            u_ij = a0.x; v_ij = a0.y; u_mj = a1.x; v_mj = a1.y;
        }
        const bool tact = (m & CM_T) != 0;
        double str1 = 0.0, str2 = 0.0, str3 = 0.0, str4 = 0.0, str5 = 0.0, str6 = 0.0, str7 = 0.0, str8 = 0.0;

        if (__any(tact)) {
            if (tact) {
                const double2 cp = ldp(rb, pp, F_CXP, lo), cm = ldp(rb, pp, F_CXM, lo);
                const double2 dd = ldp(rb, pp, F_DXT, lo), dh = ldp(rb, pp, F_DXHY, lo);
                const double2 ts = ldp(rb, pp, F_TINYAREA, lo);
                const double2 q0 = ldp(rb, pp, SR + S_SP, lo), q1 = ldp(rb, pp, SR + S_SP + 2, lo);
                const double2 q2 = ldp(rb, pp, SR + S_SM, lo), q3 = ldp(rb, pp, SR + S_SM + 2, lo);
                const double2 q4 = ldp(rb, pp, SR + S_S12, lo), q5 = ldp(rb, pp, SR + S_S12 + 2, lo);
                const double cxp = cp.x, cyp = cp.y, cxm = cm.x, cym = cm.y;
                const double dxt = dd.x, dyt = dd.y, dxhy = dh.x, dyhx = dh.y;
                const double tiny = ts.x, strength = ts.y;
                double sp1 = q0.x, sp2 = q0.y, sp3 = q1.x, sp4 = q1.y;
                double sm1 = q2.x, sm2 = q2.y, sm3 = q3.x, sm4 = q3.y;
                double s121 = q4.x, s122 = q4.y, s123 = q5.x, s124 = q5.y;

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
                        const double tarear = *reinterpret_cast<const double *>(rb + (size_t)(F_TAREAR >> 1) * pp + lo + (F_TAREAR & 1) * 8);
                        const double divu = 0.25 * (divune + divunw + divuse + divusw) * tarear;
                        const double tmp = 0.25 * (Deltane + Deltanw + Deltase + Deltasw) * tarear;
                        st1(rb, pp, F_DIVU, lo, divu);
                        st1(rb, pp, F_RDGCONV, lo, -fmin(divu, 0.0));
                        st1(rb, pp, F_RDGSHEAR, lo, 0.5 * (tmp - fabs(divu)));
                        const double tt = tensionne + tensionnw + tensionse + tensionsw;
                        const double ss = shearne + shearnw + shearse + shearsw;
                        st1(rb, pp, F_SHEAR, lo, 0.25 * tarear * sqrt(tt * tt + ss * ss));
                    }
                }

                // replacement pressure / Delta (:683-697)
                double c0ne = strength / fmax(Deltane, tiny);
                double c0nw = strength / fmax(Deltanw, tiny);
                double c0sw = strength / fmax(Deltasw, tiny);
                double c0se = strength / fmax(Deltase, tiny);
                if (LAST) { if (store) st1(rb, pp, F_PRSSIG, lo, c0ne * Deltane); }
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
                    stp(rb, pp, SW + S_SP, lo, sp1, sp2);    stp(rb, pp, SW + S_SP + 2, lo, sp3, sp4);
                    stp(rb, pp, SW + S_SM, lo, sm1, sm2);    stp(rb, pp, SW + S_SM + 2, lo, sm3, sm4);
                    stp(rb, pp, SW + S_S12, lo, s121, s122); stp(rb, pp, SW + S_S12 + 2, lo, s123, s124);
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
                    char *const ru = rb - rowb;
                    const double uold = u_im, vold = v_im;
                    const double2 va = ldp(ru, pp, F_VRELC, lo), oc = ldp(ru, pp, F_UOCN, lo);
                    const double2 wa = ldp(ru, pp, F_WATERX, lo), fo = ldp(ru, pp, F_FORCEX, lo);
                    const double2 mf = ldp(ru, pp, F_UMASSDTI, lo);
                    const double vrelc = va.x, uarear = va.y, uocn = oc.x, vocn = oc.y;
                    const double waterx = wa.x, watery = wa.y, forcex = fo.x, forcey = fo.y;
                    const double umassdti = mf.x, fm = mf.y;
                    const double du = uocn - uold, dv = vocn - vold;
                    const double vrel = vrelc * sqrt(du * du + dv * dv);            // :708-709
                    const double taux = vrel * waterx, tauy = vrel * watery;        // :711-712
                    const double cca = (a.brlx + a.revp) * umassdti + vrel * a.cosw;   // :715
                    const double ccb = fm + copysign(1.0, fm) * vrel * a.sinw;      // :720
                    const double ab2 = cca * cca + ccb * ccb;
                    const double strintx = uarear * (((s1c + s2r) + str3) + s4n);   // :725-728
                    const double strinty = uarear * (((s5c + str6) + s7r) + s8n);
                    double ui = 0.0, vi = 0.0;
                    if (REVP) { const double2 iv = ldp(ru, pp, F_UVEL_INIT, lo); ui = iv.x; vi = iv.y; }
                    const double cc1 = strintx + forcex + taux + umassdti * (a.brlx * uold + a.revp * ui);   // :731-734
                    const double cc2 = strinty + forcey + tauy + umassdti * (a.brlx * vold + a.revp * vi);
                    const double un = (cca * cc1 + ccb * cc2) / ab2;                // :736-737
                    const double vn = (cca * cc2 - ccb * cc1) / ab2;
                    stp(ru, pp, SW + S_U, lo, un, vn);
                    if (a.wrap) {   // single-rank cyclic E-W: the owner also writes the ghost image
                        if (i == 1) stp(ru, pp, SW + S_U, lo + (unsigned)s.nxl * 16u, un, vn);
                        if (i == s.nxl) stp(ru, pp, SW + S_U, lo - (unsigned)s.nxl * 16u, un, vn);
                    }
                    if (LAST) {
                        st1(ru, pp, F_STRINTX, lo, strintx);
                        st1(ru, pp, F_STRINTY, lo, strinty);
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
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl && s.iceumask[km]) {
        const int SB = cur ? F_STATE1 : F_STATE0;
        const double u = FD(s, SB + S_U, k), v = FD(s, SB + S_V, k);
        const double du = FD(s, F_UOCN, k) - u, dv = FD(s, F_VOCN, k) - v;
        const double aiu = FD(s, F_AIU, k), fm = FD(s, F_FM, k);
        double vrel = p.rhow * FD(s, F_CW, k) * sqrt(du * du + dv * dv);        // :818-819
        vrel = vrel * aiu;                                                         // :827
        const double sg = copysign(1.0, fm);
        const double sx = vrel * (du * p.cosw - dv * p.sinw * sg);                 // :828-831
        const double sy = vrel * (dv * p.cosw + du * p.sinw * sg);
        FD(s, F_STROCNX, k) = sx;
        FD(s, F_STROCNY, k) = sy;
        xT = sx / aiu;                                                             // :840-841
        yT = sy / aiu;
    }
    // strocnxT/yT before the U->T average (u2tgrid_vector works on a copy, ice_grid.F90:1899): work planes
    FD(s, F_WORK1, k) = xT;
    FD(s, F_WORK2, k) = yT;
}

}  // namespace evpk
