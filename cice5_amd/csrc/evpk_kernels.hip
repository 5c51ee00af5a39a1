// evpk_kernels.hip -- CDNA4 (gfx950) kernels of the EVP sea-ice dynamics path.
//
// Built with -ffp-contract=off: the arithmetic is the plain IEEE fp64 sequence the
// reference spells out, so results are bit-comparable with a non-FMA CPU build.
// fp64 sqrt and divide lower to correctly rounded sequences on gfx950.
//
// Hot kernels: k_subcycle2p / k_subcycle2 -- one launch = TWO EVP subcycles, and k_subcycle -- one launch = one EVP
// subcycle = stress (ice_dyn_evp.F90:520-849) fused with stepu (ice_dyn_shared.F90:623-748).  The str(:,:,1:8)
// work array of the reference never reaches HBM: it lives in registers and moves between lanes with DPP wave
// shifts.  u, v and the twelve sigma planes are double buffered (read buffer `cur`, write buffer `cur^1`) so that
// redundant T cells on strip edges see old values.  All hot fields are read and written as double2 pairs (16 B per
// lane, see evpk_internal.h).  Order of the file: block <-> slab transfer, per-evp kernels (prep, strength, T<->U
// averages), halo / fold / ghost-zone kernels, strip bookkeeping, the subcycle kernels, finish.
#include "evpk_internal.h"

namespace evpk {

// ------------------------------------------------------------------------------------
// gather / scatter between the reference's block layout and the slab
// ------------------------------------------------------------------------------------
enum { MODE_PHYS = 0, MODE_ALL = 1, MODE_NE = 2, MODE_NE_FOLD = 3,
       MODE_PHYS_ZG = 4 };   // physical cells delivered, the block's ghost cells zeroed (to_ugrid: work2(:,:,:) = c0, ice_grid.F90:1852)

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
// `act` (optional): tile activity flags -- cells of inactive 64x4-cell tiles are not transferred (sparse I/O: with the
// caller's array page-locked the kernel reads it in place over PCIe, so skipped cells cost no PCIe traffic)
__global__ void k_gather_f(Slab s, const BlockDesc *bd, int nxb, int nyb, const double *src, int f, const unsigned char *act = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;   // 1-based block column
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    if (act && !act[(sj / TILE_Y) * s.ntx + si / TILE_X]) return;
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

// one (nx_block, ny_block) slice per block, blocks `bstride` doubles apart (a category of aicen / vicen), into a plain
// plane with the mask planes' indexing
__global__ void k_gather_plane(Slab s, const BlockDesc *bd, int nxb, int nyb, const double *src, size_t bstride, double *dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    if (!gather_take(s, d, i, j, si, sj)) return;
    dst[mcell(s, si, sj)] = src[(size_t)b * bstride + (size_t)(j - 1) * nxb + (i - 1)];
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

__global__ void k_scatter_f(Slab s, const BlockDesc *bd, int nxb, int nyb, int f, double *dst, int mode, const unsigned char *act = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    int si, sj;
    if (act) {      // sparse download: a cell of a tile that is inactive now and was at the previous evp holds the same zero on both sides
        const BlockDesc d = bd[b];
        const int ti = d.iglob_lo + (i - d.ilo) - s.i0 + 1, tj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
        if (ti >= 0 && ti <= s.nxl + 1 && tj >= 0 && tj <= s.nyl + 1 && !act[(tj / TILE_Y) * s.ntx + ti / TILE_X]) return;
    }
    if (mode == MODE_PHYS_ZG) {
        const BlockDesc d = bd[b];
        const bool phys = (i >= d.ilo && i <= d.ihi && j >= d.jlo && j <= d.jhi);
        double v = 0.0;
        if (phys) { (void)scatter_take(s, d, i, j, MODE_PHYS, si, sj); v = FD(s, f, cell(s, si, sj)); }
        dst[((size_t)b * nyb + (j - 1)) * nxb + (i - 1)] = v;      // (padding cells of a short edge block are ghost cells too)
        return;
    }
    if (!scatter_take(s, bd[b], i, j, mode, si, sj)) return;
    dst[((size_t)b * nyb + (j - 1)) * nxb + (i - 1)] = FD(s, f, cell(s, si, sj));
}

// Several block arrays in ONE launch (round 5): a thread moves its cell of every array of the list, all loads issued before the first
// store -- over PCIe (page-locked caller arrays read / written in place) that keeps up to XFER_MAX reads of a lane in flight instead
// of one per launch, and a dozen launches with their gaps become one.  Same cell rules as k_gather_f / k_scatter_f.
constexpr int XFER_MAX = 12;
struct XferList {
    int n;
    int f[XFER_MAX];             // slab field of each array
    int mode[XFER_MAX];          // scatter: MODE_* of each array
    double *host[XFER_MAX];      // the caller's arrays (device-visible aliases)
};
__global__ void k_gather_multi(Slab s, const BlockDesc *bd, int nxb, int nyb, XferList L, const unsigned char *act) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    if (act && !act[(sj / TILE_Y) * s.ntx + si / TILE_X]) return;
    if (!gather_take(s, d, i, j, si, sj)) return;
    const size_t k = ((size_t)b * nyb + (j - 1)) * nxb + (i - 1);
    const size_t c = cell(s, si, sj);
    double v[XFER_MAX];
#pragma unroll
    for (int q = 0; q < XFER_MAX; q++) if (q < L.n) v[q] = L.host[q][k];
#pragma unroll
    for (int q = 0; q < XFER_MAX; q++) if (q < L.n) FD(s, L.f[q], c) = v[q];
}

__global__ void k_scatter_multi(Slab s, const BlockDesc *bd, int nxb, int nyb, XferList L, const unsigned char *act) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    if (act) {      // sparse download: a cell of a tile that is inactive now and was at the previous evp holds the same zero on both sides
        const int ti = d.iglob_lo + (i - d.ilo) - s.i0 + 1, tj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
        if (ti >= 0 && ti <= s.nxl + 1 && tj >= 0 && tj <= s.nyl + 1 && !act[(tj / TILE_Y) * s.ntx + ti / TILE_X]) return;
    }
    const size_t k = ((size_t)b * nyb + (j - 1)) * nxb + (i - 1);
    double v[XFER_MAX];
    bool take[XFER_MAX];
#pragma unroll
    for (int q = 0; q < XFER_MAX; q++) {
        take[q] = false;
        if (q < L.n) {
            int si, sj;
            if (L.mode[q] == MODE_PHYS_ZG) {       // physical cells delivered, ghost (and padding) cells of the block zeroed
                const bool phys = (i >= d.ilo && i <= d.ihi && j >= d.jlo && j <= d.jhi);
                take[q] = true; v[q] = 0.0;
                if (phys) { (void)scatter_take(s, d, i, j, MODE_PHYS, si, sj); v[q] = FD(s, L.f[q], cell(s, si, sj)); }
            } else if (scatter_take(s, d, i, j, L.mode[q], si, sj)) {
                take[q] = true; v[q] = FD(s, L.f[q], cell(s, si, sj));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < XFER_MAX; q++) if (q < L.n && take[q]) L.host[q][k] = v[q];
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

// ---- evpk_halo_update: one (nx_block, ny_block) slice per block, blocks `bstride` doubles apart, <-> field f of the slab ----
__global__ void k_gather_fs(Slab s, const BlockDesc *bd, int nxb, int nyb, const double *src, size_t bstride, int f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    if (i < d.ilo || i > d.ihi || j < d.jlo || j > d.jhi) return;           // physical cells only: the update rewrites the ring
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    FD(s, f, cell(s, si, sj)) = src[(size_t)b * bstride + (size_t)(j - 1) * nxb + (i - 1)];
}

// 1.0 on the physical cells of the blocks (cells of eliminated land blocks keep the plane's 0)
__global__ void k_cover_f(Slab s, const BlockDesc *bd, int nxb, int nyb, int f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    if (i < d.ilo || i > d.ihi || j < d.jlo || j > d.jhi) return;
    FD(s, f, cell(s, d.iglob_lo + (i - d.ilo) - s.i0 + 1, d.jglob_lo + (j - d.jlo) - s.j0 + 1)) = 1.0;
}

// What ice_HaloUpdate leaves in a block array (production backend, mpi/ice_boundary.F90:1331-1700): a ghost cell with a
// neighbour takes the slab's value (the neighbour's, `fill` where that is an eliminated land block, the fold on the tripole
// row; for NE-corner / N-face fields the fold also rewrites the top physical row); a cell nothing writes -- beyond an open /
// closed boundary, padding of a short block -- keeps the caller's value unless it lies on the outermost row / column of the
// ARRAY, which the reference fills first (:1409-1416).
// stress != 0: ice_HaloUpdate_stress -- only the tripole north ghost row, and 0 next to an eliminated land block (fcov: the
// halo-updated coverage plane, < 0 if every cell is covered)
__global__ void k_scatter_halo(Slab s, const BlockDesc *bd, int nxb, int nyb, int f, double *dst, size_t bstride, double fill,
                               int cyclic, int tripole, int top_row_too, int stress, int fcov) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    double *o = dst + (size_t)b * bstride + (size_t)(j - 1) * nxb + (i - 1);
    const bool edge = (i == 1 || i == nxb || j == 1 || j == nyb);
    const bool phys = (i >= d.ilo && i <= d.ihi && j >= d.jlo && j <= d.jhi);
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1, sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    const int gi = s.i0 + si - 1, gj = s.j0 + sj - 1;
    if (phys) {
        if (top_row_too && !stress && tripole && gj == s.nyg) *o = FD(s, f, cell(s, si, sj));
        return;
    }
    const bool padding = (i > d.ihi + 1 || j > d.jhi + 1);
    const bool has_src = !padding && ((gi >= 1 && gi <= s.nxg) || cyclic) && ((gj >= 1 && gj <= s.nyg) || (tripole && gj == s.nyg + 1));
    if (stress) {
        if (!has_src) return;
        if (tripole && gj == s.nyg + 1) *o = FD(s, f, cell(s, si, sj));
        else if (fcov >= 0 && FD(s, fcov, cell(s, si, sj)) == 0.0) *o = 0.0;
        return;
    }
    if (has_src) *o = FD(s, f, cell(s, si, sj));
    else if (edge) *o = fill;
}

// restart records: physical cells of one field <-> a (ny_global, nx_global) array in global order (the layout
// gather_global / scatter_global give the master task, ice_gather_scatter.F90); f < 0: the iceumask plane as 0 / 1
__global__ void k_slab_to_global(Slab s, int f, double *G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y + 1;
    if (i > s.nxl) return;
    const double v = f >= 0 ? FD(s, f, cell(s, i, j)) : (s.iceumask[mcell(s, i, j)] ? 1.0 : 0.0);
    G[(size_t)(s.j0 + j - 2) * s.nxg + (s.i0 + i - 2)] = v;
}
__global__ void k_global_to_slab(Slab s, int f, const double *G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y + 1;
    if (i > s.nxl) return;
    const double v = G[(size_t)(s.j0 + j - 2) * s.nxg + (s.i0 + i - 2)];
    if (f >= 0) FD(s, f, cell(s, i, j)) = v;
    else s.iceumask[mcell(s, i, j)] = v > 0.5 ? 1 : 0;                     // ice_restart_driver.F90:399-409
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

#define TILE_SKIP(flags) if (!(flags)[blockIdx.y * s.ntx + blockIdx.x]) return;

// ------------------------------------------------------------------------------------
// evp_prep1 (ice_dyn_shared.F90:270-365) on the slab
// ------------------------------------------------------------------------------------
// `fresh` = the state planes were just uploaded from the host (anything may be non-zero anywhere).
// Otherwise cells that were inactive at the previous prep still hold their zeros and are skipped.
// up_dat (k_up_tiles, after every upload): tiles in which the host's inputs hold anything at all.  A tile without, that was not active
// at the previous evp either, is left alone WITHOUT reading its cells (74 % of the bench grid; the inputs only change by uploads).
__global__ void k_up_tiles(Slab s, DevParams p, unsigned char *up_dat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    bool dat = false;
    if (i <= s.nxl + 1 && j <= s.nyl + 1) {
        const size_t k = cell(s, i, j);
        const double vice = FD(s, F_VICE, k), vsno = FD(s, F_VSNO, k), aice = FD(s, F_AICE, k);
        const double tmass = s.tmask[mcell(s, i, j)] != 0 ? (p.rhoi * vice + p.rhos * vsno) : 0.0;
        dat = (tmass != 0.0) || (aice != 0.0) || (FD(s, F_AICE_INIT, k) != 0.0) || (FD(s, F_STRAIRXT, k) != 0.0) || (FD(s, F_STRAIRYT, k) != 0.0);
    }
    const int any = __syncthreads_or(dat ? 1 : 0);
    if (threadIdx.x == 0 && threadIdx.y == 0) up_dat[blockIdx.y * s.ntx + blockIdx.x] = any ? 1 : 0;
}

__global__ void k_prep1a(Slab s, DevParams p, int fresh, const unsigned char *up_dat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    {   // (block-uniform: every thread of the tile leaves before the block-wide votes below)
        const int t0 = blockIdx.y * s.ntx + blockIdx.x;
        const bool edge0 = (blockIdx.x == 0 || blockIdx.x == (unsigned)s.ntx - 1 || blockIdx.y == 0 || blockIdx.y == (unsigned)s.nty - 1);
        if (up_dat && !up_dat[t0] && !edge0 && !fresh && !s.act_any[t0]) {
            if (threadIdx.x == 0 && threadIdx.y == 0) { s.tile_ice[t0] = 0; s.tile_dat[t0] = 0; }
            return;
        }
    }
    const bool in = (i <= s.nxl + 1 && j <= s.nyl + 1);
    const size_t k = in ? cell(s, i, j) : 0, km = in ? mcell(s, i, j) : 0;
    double tmass = 0.0, wx = 0.0, wy = 0.0;
    bool hm = false, dat = false;
    if (in) {
        const double vice = FD(s, F_VICE, k), vsno = FD(s, F_VSNO, k), aice = FD(s, F_AICE, k);
        const bool tm = s.tmask[km] != 0;
        if (tm) tmass = (p.rhoi * vice + p.rhos * vsno);                          // :322-326
        hm = tm && (aice > p.a_min) && (tmass > p.m_min);                         // :331-332
        wx = FD(s, F_STRAIRXT, k); wy = FD(s, F_STRAIRYT, k);
        dat = (tmass != 0.0) || (aice != 0.0) || (FD(s, F_AICE_INIT, k) != 0.0) || (wx != 0.0) || (wy != 0.0);
    }
    // tile activity (block-uniform): nothing in, nothing in last time -> every output of this evp is already zero here
    const int t = blockIdx.y * s.ntx + blockIdx.x;
    const int any_ice = __syncthreads_or(hm ? 1 : 0), any_dat = __syncthreads_or(dat ? 1 : 0);
    const bool edge = (blockIdx.x == 0 || blockIdx.x == (unsigned)s.ntx - 1 || blockIdx.y == 0 || blockIdx.y == (unsigned)s.nty - 1);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        s.tile_ice[t] = (any_ice || edge) ? 1 : 0;          // tiles touching the ghost ring always count as active
        s.tile_dat[t] = (any_dat || edge) ? 1 : 0;
    }
    if (!in) return;
    // s.act_any still holds the previous evp's dilated activity here (k_tile_dilate runs after this kernel): it also
    // covers tiles that only border ice, whose T cells next to the ice carried diagnostics last time
    if (!(any_dat || edge || fresh || s.act_any[t])) return;
    FD(s, F_TMASS, k) = tmass;
    s.tmphm[km] = hm ? 1 : 0;
    // :339-340 strairx = strairxT; the T->U average that follows (t2ugrid_vector) reads it from the
    // work planes, so the copy lands there directly; U-grid wind (ACCESS) goes straight to strairx/y
    const int fx = p.wind_on_ugrid ? F_STRAIRX : F_WORK1, fy = p.wind_on_ugrid ? F_STRAIRY : F_WORK2;
    FD(s, fx, k) = wx;
    FD(s, fy, k) = wy;
    // evp(): zero the diagnostics (ice_dyn_evp.F90:174-182); only T cells active last time can be non-zero
    if (fresh || (s.cmask[km] & CM_T)) {
        FD(s, F_RDGCONV, k) = 0.0; FD(s, F_RDGSHEAR, k) = 0.0; FD(s, F_DIVU, k) = 0.0;
        FD(s, F_SHEAR, k) = 0.0; FD(s, F_PRSSIG, k) = 0.0;
    }
}

// sparse upload: which tiles can matter to this evp -- any cell with ice (the test of evp_prep1, :331-332, or any non-zero
// aice / tmass at all) on the freshly uploaded aice / vice / vsno, or the ghost ring
// (aice_only: vice / vsno have not been uploaded yet -- they are zero wherever aice is, evpk_params.sparse_io = 2)
__global__ void k_io_tiles(Slab s, DevParams p, unsigned char *raw, int aice_only) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    bool hm = false;
    if (i <= s.nxl + 1 && j <= s.nyl + 1) {
        const size_t k = cell(s, i, j);
        if (aice_only) {
            hm = FD(s, F_AICE, k) != 0.0;
        } else {
            const double tmass = s.tmask[mcell(s, i, j)] ? (p.rhoi * FD(s, F_VICE, k) + p.rhos * FD(s, F_VSNO, k)) : 0.0;
            hm = (s.tmask[mcell(s, i, j)] && FD(s, F_AICE, k) > p.a_min && tmass > p.m_min) || tmass != 0.0 || FD(s, F_AICE, k) != 0.0;
        }
    }
    const int any = __syncthreads_or(hm ? 1 : 0);
    const bool edge = (blockIdx.x == 0 || blockIdx.x == (unsigned)s.ntx - 1 || blockIdx.y == 0 || blockIdx.y == (unsigned)s.nty - 1);
    if (threadIdx.x == 0 && threadIdx.y == 0) raw[blockIdx.y * s.ntx + blockIdx.x] = (any || edge) ? 1 : 0;
}
// ... dilated by two tiles (the T->U averages and the 3x3 dilation of icetmask reach one cell further, the tiles of the
// per-evp kernels one tile further), or active at the previous evp (s.act_any still holds that)
__global__ void k_io_tiles_dilate(Slab s, const unsigned char *raw, unsigned char *act) {
    const int tx = blockIdx.x * blockDim.x + threadIdx.x, ty = blockIdx.y;
    if (tx >= s.ntx || ty >= s.nty) return;
    int a = s.act_any[ty * s.ntx + tx];
    for (int dy = -2; dy <= 2 && !a; dy++)
        for (int dx = -2; dx <= 2; dx++) {
            const int x = tx + dx, y = ty + dy;
            if (x < 0 || x >= s.ntx || y < 0 || y >= s.nty) continue;
            a |= raw[y * s.ntx + x];
        }
    act[ty * s.ntx + tx] = a ? 1 : 0;
}

// act = dilate_3x3(new | prev) per tile; `fresh` (state just uploaded): everything is active
__global__ void k_tile_dilate(Slab s, const unsigned char *prev_ice, const unsigned char *prev_dat, int fresh) {
    const int tx = blockIdx.x * blockDim.x + threadIdx.x, ty = blockIdx.y;
    if (tx >= s.ntx || ty >= s.nty) return;
    int ice = fresh, any = fresh;
    for (int dy = -1; dy <= 1 && !ice; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            const int x = tx + dx, y = ty + dy;
            if (x < 0 || x >= s.ntx || y < 0 || y >= s.nty) continue;
            const int q = y * s.ntx + x;
            ice |= s.tile_ice[q] | prev_ice[q];
            any |= s.tile_dat[q] | prev_dat[q];
        }
    if (!ice) {     // the loop above stops early only when ice is set; finish `any` otherwise
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int x = tx + dx, y = ty + dy;
                if (x < 0 || x >= s.ntx || y < 0 || y >= s.nty) continue;
                any |= s.tile_dat[y * s.ntx + x] | prev_dat[y * s.ntx + x];
            }
    }
    s.act_ice[ty * s.ntx + tx] = ice ? 1 : 0;
    s.act_any[ty * s.ntx + tx] = (ice | any) ? 1 : 0;
}

__global__ void k_prep1b(Slab s) {
    TILE_SKIP(s.act_ice)
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

// ------------------------------------------------------------------------------------
// ice_strength (ice_mechred.F90:2111-2269) with ridge_itd (:936-1285), one thread per T cell.
// exp(): Cody-Waite reduction + the degree-5 minimax in r^2 of fdlibm's e_exp.c (< 1 ulp), un-fused: the same operations
// in the same order as the CPU restatement the tests compare with.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double dev_exp(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const double ax = fabs(x);
    double hi = 0.0, lo = 0.0;
    int k = 0;
    if (ax > 0.34657359027997264) {
        if (ax < 1.0397207708399179) {
            k = x < 0.0 ? -1 : 1;
            hi = x - (double)k * ln2HI;
            lo = (double)k * ln2LO;
        } else {
            k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
            const double t = (double)k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) {
        return 1.0 + x;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    return ldexp(y, k);
}

constexpr int MAXCAT = 16;

// itd: planes [0..ncat) aicen, [ncat..2 ncat) vicen, [2 ncat] aice0, each mask_elems(s) doubles, index mcell()
// NC > 0: the number of thickness categories at compile time (NC = 5: the reference's default, ice_domain_size) -- the per-category work
// arrays then live in registers; NC = 0: any ncat <= MAXCAT, the arrays in scratch memory (864 bytes per lane)
template <int NC>
__global__ void k_ice_strength(Slab s, DevParams p, const double *itd) {
    TILE_SKIP(s.act_any)
    SLAB_IJ_ALL
    const double puny = 1.0e-11, c0 = 0.0, c1 = 1.0, c2 = 2.0, p5 = 0.5, p333 = 1.0 / 3.0;
    const double Gstar = 0.15, astar = 0.05, maxraft = 1.0, Hstar = 25.0, Pstar = 2.75e4, Cstar = 20.0;     // :72-82
    double str = c0;                                                                                        // :2183
    if (p.kstrength != 1) {                                                                                 // :2258-2265
        if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl)
            str = Pstar * FD(s, F_VICE, k) * dev_exp(-Cstar * (c1 - FD(s, F_AICE, k)));
    } else if (i >= 1 && j >= 1 && FD(s, F_ICETM, k) == 1.0) {      // the T-cell list of evp_prep2 (ice_dyn_shared.F90:528-537)
        const size_t np = mask_elems(s);
        const int ncat = NC ? NC : p.ncat;
        const double Cp = p5 * p.gravit * (p.rhow - p.rhoi) * p.rhoi / p.rhow;                              // :68
        const double Gstari = c1 / Gstar, astari = c1 / astar;
        constexpr int NA = NC ? NC : MAXCAT;
        double Gsum[NA + 2], apartic[NA + 1], hrmin[NA + 1], hrmax[NA + 1], hrexp[NA + 1], krdg[NA + 1];
        const double a0 = itd[(size_t)(2 * ncat) * np + km];
        Gsum[0] = c0;
        apartic[0] = c0;
        _Pragma("unroll")
        for (int n = 1; n <= ncat; n++) { apartic[n] = c0; hrmin[n] = c0; hrmax[n] = c0; hrexp[n] = c0; krdg[n] = c1; }
        Gsum[1] = (a0 > puny) ? a0 : Gsum[0];
        _Pragma("unroll")
        for (int n = 1; n <= ncat; n++) {
            const double a = itd[(size_t)(n - 1) * np + km];
            Gsum[n + 1] = (a > puny) ? Gsum[n] + a : Gsum[n];
        }
        const double work = c1 / Gsum[ncat + 1];
        _Pragma("unroll")
        for (int n = 0; n <= ncat; n++) Gsum[n + 1] = Gsum[n + 1] * work;
        if (p.krdg_partic == 0) {
            _Pragma("unroll")
            for (int n = 0; n <= ncat; n++) {
                const double g1 = Gsum[n + 1], g0 = Gsum[n];
                if (g1 < Gstar) apartic[n] = Gstari * (g1 - g0) * (c2 - (g0 + g1) * Gstari);
                else if (g0 < Gstar) apartic[n] = Gstari * (Gstar - g0) * (c2 - (g0 + Gstar) * Gstari);
            }
        } else {
            const double xtmp = c1 / (c1 - dev_exp(-astari));
            _Pragma("unroll")
            for (int n = -1; n <= ncat; n++) Gsum[n + 1] = dev_exp(-Gsum[n + 1] * astari) * xtmp;
            _Pragma("unroll")
            for (int n = 0; n <= ncat; n++) apartic[n] = Gsum[n] - Gsum[n + 1];
        }
        _Pragma("unroll")
        for (int n = 1; n <= ncat; n++) {
            const double a = itd[(size_t)(n - 1) * np + km];
            if (a > puny) {
                double hi = itd[(size_t)(ncat + n - 1) * np + km] / a;
                if (p.krdg_redist == 0) {
                    hrmin[n] = fmin(c2 * hi, hi + maxraft);
                    hrmax[n] = c2 * sqrt(Hstar * hi);
                    hrmax[n] = fmax(hrmax[n], hrmin[n] + puny);
                    const double hrmean = p5 * (hrmin[n] + hrmax[n]);
                    krdg[n] = hrmean / hi;
                } else {
                    hi = fmax(hi, puny);
                    hrmin[n] = fmin(c2 * hi, hi + maxraft);
                    hrexp[n] = p.mu_rdg * sqrt(hi);
                    krdg[n] = (hrmin[n] + hrexp[n]) / hi;
                }
            }
        }
        double aksum = apartic[0];
        _Pragma("unroll")
        for (int n = 1; n <= ncat; n++) aksum = aksum + apartic[n] * (c1 - c1 / krdg[n]);
        double sacc = c0;
        _Pragma("unroll")
        for (int n = 1; n <= ncat; n++) {
            const double a = itd[(size_t)(n - 1) * np + km];
            if (a > puny && apartic[n] > c0) {
                const double hi = itd[(size_t)(ncat + n - 1) * np + km] / a;
                double h2rdg;
                if (p.krdg_redist == 0)
                    h2rdg = p333 * (hrmax[n] * hrmax[n] * hrmax[n] - hrmin[n] * hrmin[n] * hrmin[n]) / (hrmax[n] - hrmin[n]);
                else
                    h2rdg = hrmin[n] * hrmin[n] + c2 * hrmin[n] * hrexp[n] + c2 * hrexp[n] * hrexp[n];
                const double dh2rdg = -hi * hi + h2rdg / krdg[n];
                sacc = sacc + apartic[n] * dh2rdg;
            }
        }
        str = p.Cf * Cp * sacc / aksum;
    }
    FD(s, F_STRENGTH, k) = str;
}
template __global__ void k_ice_strength<0>(Slab, DevParams, const double *);
template __global__ void k_ice_strength<5>(Slab, DevParams, const double *);

// the four T->U averages of evp() in one pass: umass <- tmass, aiu <- aice_init (ice_dyn_evp.F90:218-219) and,
// unless the wind is already on the U grid, strairx/y <- work1/2 (t2ugrid_vector, :240-241)
__global__ void k_to_ugrid4(Slab s, int wind) {
    TILE_SKIP(s.act_any)
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
    TILE_SKIP(s.act_any)
    SLAB_IJ_ALL
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {
        const size_t kw = cell(s, i - 1, j), ks = cell(s, i, j - 1), ksw = cell(s, i - 1, j - 1);
        const double u0 = FD(s, F_UAREA, k), u1 = FD(s, F_UAREA, kw), u2 = FD(s, F_UAREA, ks), u3 = FD(s, F_UAREA, ksw);
        const double ta = FD(s, F_TAREA, k);
#define TG_(f) (0.25 * (((FD(s, f, k) * u0 + FD(s, f, kw) * u1) + FD(s, f, ks) * u2) + FD(s, f, ksw) * u3) / ta)
        FD(s, F_STROCNXT, k) = TG_(F_WORK3);
        FD(s, F_STROCNYT, k) = TG_(F_WORK4);
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

// rows j0..j1 (all columns incl. ghosts) of nf planes: dst <- src
__global__ void k_rows_copy(Slab s, int fsrc, int fdst, int nf, int j0, int j1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = j0 + blockIdx.y;
    if (i > s.nxl + 1 || j > j1) return;
    const size_t k = cell(s, i, j);
    for (int q = 0; q < nf; q++) FD(s, fdst + q, k) = FD(s, fsrc + q, k);
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

// profiling aid, second form: a streaming copy between two buffers far larger than the 256 MiB Infinity Cache (16 B per
// lane, coalesced): n double2 read, n double2 written, nothing the caches can absorb
// the same byte count with 8 B per lane (the access shape of the plain planes of horizontal_remap / eap)
__global__ void k_calib_copy_big8(const double *__restrict__ src, double *__restrict__ dst, size_t n) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] = src[k];
}
__global__ void k_calib_copy_big(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] = src[k];
}

// ------------------------------------------------------------------------------------
// evp_prep2 (ice_dyn_shared.F90:377-614) on the slab.  State lives in buffer `cur` on entry; both
// buffers are left identical (the velocity ring is completed by the halo update + ring copy that
// follow in evpk_prep).  Invariant kept for the subcycle kernel, which never touches inactive
// cells: sigma = 0 in both buffers where icetmask == 0, u = v = 0 in both where iceumask is false,
// and the stepu input planes are 0 where iceumask is false.  Unless `fresh`, a cell that was
// inactive at the previous prep already satisfies this and is skipped.
// ------------------------------------------------------------------------------------
__global__ void k_prep2(Slab s, DevParams p, int fresh, int cur) {
    TILE_SKIP(s.act_any)
    SLAB_IJ_ALL
    const int SA = cur ? F_STATE1 : F_STATE0;      // buffer holding the current state
    const int SB = cur ? F_STATE0 : F_STATE1;      // the other one, made identical here
    const bool icet = FD(s, F_ICETM, k) == 1.0;          // after its halo update
    const unsigned char cmold = s.cmask[km];
    const bool prevT = fresh || (cmold & CM_T), prevU = fresh || (cmold & CM_U);
    if (p.revp == 1.0 || !icet) {                                                  // :492-518
        if (icet || prevT) {
#pragma unroll
            for (int c = S_SP; c < NSTATE; c++) { FD(s, SA + c, k) = 0.0; FD(s, SB + c, k) = 0.0; }
        }
    }
    // (an active T cell needs nothing here: the first kernel of the loop reads buffer `cur` and rewrites the other one)
    unsigned char cm = icet ? CM_T : 0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {                            // :545-577
        const double aiu = FD(s, F_AIU, k), umass = FD(s, F_UMASS, k);
        const bool old = s.iceumask[km] != 0;
        const bool ium = (s.umask[km] != 0) && (aiu > p.a_min) && (umass > p.m_min);
        s.iceumask[km] = ium ? 1 : 0;
        if (ium) {
            const double uocn = FD(s, F_UOCN, k), vocn = FD(s, F_VOCN, k);
            double u = FD(s, SA + S_U, k), v = FD(s, SA + S_V, k);
            if (!old) { u = uocn; v = vocn; }
            cm |= CM_U;
            if (!old) { FD(s, SA + S_U, k) = u; FD(s, SA + S_V, k) = v; }   // the other buffer is rewritten by the first kernel
            FD(s, F_UVEL_INIT, k) = u;    FD(s, F_VVEL_INIT, k) = v;
            const double umdti = umass / p.dt;                                     // :583-612
            const double fm = FD(s, F_FCOR, k) * umass;
            FD(s, F_FM, k) = fm;
            double tx, ty;      // waterx/watery (:592-593) are recomputed inside stepu_cell
            if (p.tilt_from_slope) {
                tx = -p.gravit * umass * FD(s, F_SSTLTX, k);
                ty = -p.gravit * umass * FD(s, F_SSTLTY, k);
            } else {
                tx = -fm * vocn;
                ty = fm * uocn;
            }
            FD(s, F_STRTLTX, k) = tx;
            FD(s, F_STRTLTY, k) = ty;
            FD(s, F_FORCEX, k) = FD(s, F_STRAIRX, k) + tx;
            FD(s, F_FORCEY, k) = FD(s, F_STRAIRY, k) + ty;
            FD(s, F_UMASSDTI, k) = umdti;
            // stepu: vrel = aiu*rhow*Cw*sqrt(..) evaluates (aiu*rhow)*Cw first (ice_dyn_shared.F90:708)
            FD(s, F_VRELC, k) = aiu * p.rhow * FD(s, F_CW, k);
        } else if (prevU || old) {
            FD(s, SA + S_U, k) = 0.0; FD(s, SA + S_V, k) = 0.0;
            FD(s, SB + S_U, k) = 0.0; FD(s, SB + S_V, k) = 0.0;
            FD(s, F_UVEL_INIT, k) = 0.0;    FD(s, F_VVEL_INIT, k) = 0.0;
            FD(s, F_STRINTX, k) = 0.0; FD(s, F_STRINTY, k) = 0.0;
            FD(s, F_STROCNX, k) = 0.0; FD(s, F_STROCNY, k) = 0.0;
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
// fprev >= 0 (velocity in the subcycle loop): the planes f were written from the planes fprev by a kernel that leaves
// inactive U cells alone.  The reference keeps ONE array, in which the fold rewrites the whole top row after every
// subcycle -- also an inactive cell whose mirror image is active -- so the value such a cell carries into this update
// is the one the previous update left in fprev.
__global__ void k_fold_pack(Slab s, int f, int nf, double *fb, int gofs /* global col of local col 1, minus 1 */, int fprev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i > s.nxl) return;
    const bool stale = fprev >= 0 && !(s.cmask[mcell(s, i, s.nyl)] & CM_U);
    for (int q = 0; q < nf; q++) {
        fb[((size_t)q * 2 + 0) * s.nxg + (gofs + i - 1)] = FD(s, f + q, cell(s, i, s.nyl - 1));
        fb[((size_t)q * 2 + 1) * s.nxg + (gofs + i - 1)] = FD(s, (stale ? fprev : f) + q, cell(s, i, s.nyl));
    }
}

// x-slabs: the all-gathered segments [rank][nfmax*2][wmax] -> the global fold buffer [nf*2][nxg] (one launch instead of
// nranks x nf x 2 small copies).  i0[r] = first global column of rank r, i0[nranks] = nxg + 1.
__global__ void k_fold_repack(int nf, int nfmax, int nranks, int wmax, int nxg, const int *i0, const double *all, double *fb) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (g > nxg) return;
    int r = 0;
    while (r + 1 < nranks && g >= i0[r + 1]) r++;
    const size_t seg = (size_t)nfmax * 2 * wmax;
    for (int q = 0; q < nf * 2; q++) fb[(size_t)q * nxg + (g - 1)] = all[(size_t)r * seg + (size_t)q * wmax + (g - i0[r])];
}

// ---- the fold between x-slab ranks as a point-to-point exchange with the mirror ranks -------------------------------
// The rows the fold of MY columns reads sit at the mirrored columns nx - g (NE corner) and nx - g + 1 (centre), g = i0 ..
// i1: the slab of the mirror rank P-1-r plus one column of each of its neighbours -- what the reference exchanges too
// (one tripole message per neighbour rank, mpi/ice_boundary.F90:2737-2913, :6128-6168).  Every rank packs the two top rows
// of its own columns once, [nf][2][nxl], straight into the buffer of each rank that needs them (itself included; through
// peer-mapped memory where the transport offers it), and scatters what it receives into the global-row buffer the apply
// kernels read.
struct DstList { int n; double *p[6]; };
struct SrcList { int n; const double *p[6]; int i0[6]; int w[6]; };

// segment layout: seg[(q*2 + r)*nxl + (i-1)], r = 0: row nyl-1, r = 1: row nyl; `stale` as in k_fold_pack
__global__ void k_fold_pack_multi(Slab s, int f, int nf, DstList dl, int fprev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i > s.nxl) return;
    const bool stale = fprev >= 0 && !(s.cmask[mcell(s, i, s.nyl)] & CM_U);
    for (int q = 0; q < nf; q++) {
        const double r0 = FD(s, f + q, cell(s, i, s.nyl - 1));
        const double r1 = FD(s, (stale ? fprev : f) + q, cell(s, i, s.nyl));
        for (int d = 0; d < dl.n; d++) {
            dl.p[d][((size_t)q * 2 + 0) * s.nxl + (i - 1)] = r0;
            dl.p[d][((size_t)q * 2 + 1) * s.nxl + (i - 1)] = r1;
        }
    }
}

// received segments -> the global fold buffer [nf*2][nxg] (columns no segment covers are not read by this rank's apply)
__global__ void k_fold_unpack(int nf, int nxg, SrcList sl, double *fb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int src = blockIdx.y;
    if (src >= sl.n || t >= sl.w[src]) return;
    const double *seg = sl.p[src];
    const int w = sl.w[src], g = sl.i0[src] + t;
    for (int q = 0; q < nf * 2; q++) fb[(size_t)q * nxg + (g - 1)] = seg[(size_t)q * w + t];
}

// ---- peer-mapped transport: completion flags (page-locked host memory shared by the rank processes) ----------------
// k_ipc_signal runs behind the kernel that stored into the peer's buffer (the kernel boundary is the release: the stores
// have landed); k_ipc_wait runs in front of the kernel that reads this rank's buffer (the boundary after it is the
// acquire).  The spin is bounded twice -- a wall-clock limit and an iteration cap -- and reports through *err instead of
// hanging: every wave ends.
__global__ void k_ipc_signal(unsigned *flag, unsigned seq) {
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_ipc_wait(const unsigned *flag, unsigned seq, unsigned *err, unsigned long long limit_ticks /* 100 MHz */) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned it = 0; it < (1u << 26); it++) {
        // (sequence numbers wrap: compare as a signed distance)
        const int ahead = (int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq);
        if (ahead >= 0) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            // every exchange is a swap: the partner can be ONE message ahead (it has mine and went on), never two -- it would
            // need my next one for that.  More means the two ranks disagree about the exchanges they post, and the slot read
            // next (parity of the sequence number) may hold another message: report it like a wait that gave up
            if (ahead > 1) atomicAdd(err, 1u << 16);
            return;
        }
        __builtin_amdgcn_s_sleep(20);
        if (wall_clock64() - t0 > limit_ticks) break;
    }
    atomicAdd(err, 1u);
}

// u-fold copy-out (serial/ice_boundary.F90:801-888, copy lists :3752-3776)
//   center  : ghost(i,ny+1) = sgn*B2(nx-g+1)
//   NEcorner: top(i,ny) = sgn*sym(B2)(nx-g), ghost(i,ny+1) = sgn*B1(nx-g), index 0 -> nx
//   stress  : (ice_HaloUpdate_stress) center rule, no sign, source plane differs from dest
// loc: 0 centre, 1 NE corner, 2 E face, 3 N face -- source column nx - g + 1 - ioffset, ioffset = 1 for NE corner and E face;
// joffset = 1 (NE corner, N face): the top physical row is rewritten from the symmetrised row (pairs (i, nx-i), i < nx/2 for
// the NE corner, :818-824; pairs (i, nx+1-i), i <= nx/2 for the N face, :836-843) and the ghost row mirrors row ny-1
__global__ void k_fold_apply(Slab s, int fdst, int nf, const double *fb, int loc, double sgn, int own_only = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // local col 0..nxl+1
    if (i > s.nxl + 1) return;
    if (own_only && (i < 1 || i > s.nxl)) return;              // x-slab ranks: the ghost columns come with the E-W exchange that follows
    const int nx = s.nxg;
    int g = s.i0 + i - 1;                                       // global col, wrap
    if (g < 1) g += nx;
    if (g > nx) g -= nx;
    const int ioff = (loc == 1 || loc == 2) ? 1 : 0, joff = (loc == 1 || loc == 3) ? 1 : 0;
    int src = nx - g + 1 - ioff;
    if (src == 0) src = nx;
    if (src > nx) src -= nx;
    const int h = nx / 2;
    for (int q = 0; q < nf; q++) {
        const double *B1 = fb + ((size_t)q * 2 + 0) * nx - 1;  // 1-based
        const double *B2 = fb + ((size_t)q * 2 + 1) * nx - 1;
        if (!joff) {
            FD(s, fdst + q, cell(s, i, s.nyl + 1)) = sgn * B2[src];
        } else {
            // symmetrised top row at column src
            double v;
            if (loc == 1) {
                if (src >= 1 && src <= h - 1) v = 0.5 * (B2[src] + sgn * B2[nx - src]);
                else if (src >= h + 1 && src <= nx - 1) { const int ii = nx - src; v = sgn * (0.5 * (B2[ii] + sgn * B2[src])); }
                else v = B2[src];
            } else {
                if (src <= h) v = 0.5 * (B2[src] + sgn * B2[nx + 1 - src]);
                else { const int ii = nx + 1 - src; v = sgn * (0.5 * (B2[ii] + sgn * B2[src])); }
            }
            FD(s, fdst + q, cell(s, i, s.nyl)) = sgn * v;
            FD(s, fdst + q, cell(s, i, s.nyl + 1)) = sgn * B1[src];
        }
    }
}

// The twelve ice_HaloUpdate_stress calls of evp (ice_dyn_evp.F90:454-479) in one launch: the north ghost row of sigma_k
// takes the top physical row of its partner (1 <-> 3, 2 <-> 4 within stressp, stressm, stress12) at the mirrored column,
// center rule, no sign.  The sources are physical rows that none of the twelve updates writes, so their order is immaterial.
// fb: k_fold_pack of the twelve planes.
__global__ void k_fold_apply_stress12(Slab s, int fdst0, const double *fb, int own_only = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // local col 0..nxl+1
    if (i > s.nxl + 1) return;
    if (own_only && (i < 1 || i > s.nxl)) return;
    const int nx = s.nxg;
    int g = s.i0 + i - 1;
    if (g < 1) g += nx;
    if (g > nx) g -= nx;
    for (int q = 0; q < 12; q++) {
        const int src = (q & ~3) | ((q & 3) ^ 2);
        const double *B2 = fb + ((size_t)src * 2 + 1) * nx - 1;  // 1-based top row of the partner plane
        FD(s, fdst0 + q, cell(s, i, s.nyl + 1)) = B2[nx - g + 1];
    }
}

// Whole halo update of an NE-corner vector field on a single-rank tripole grid in ONE launch (the subcycle loop calls
// it twice per launch pair): south ghost row, the u-fold of k_fold_apply straight from the planes, and the E-W ghost
// columns.  The symmetrised top row of column g depends on the old values at g and nx-g only, so one thread owns that
// pair and no staging buffer is needed; every other value read lies in a cell this kernel does not write.
//   blockIdx.y == 0: fold (thread k: columns k and nx-k; k = 0 stands for column nx), rows nyl and nyl+1
//   blockIdx.y == 1: rows 0 .. nyl-1: south fill and the two ghost columns
// ew_from: the E-W ghost columns are refreshed for rows ew_from .. nyl-1 only (the rows below are another launch's business)
__global__ void k_halo_tripole_ne1(Slab s, int f, int nf, int cyclic, double fill, double sgn, int fprev /* see k_fold_pack */, int ew_from) {
    const int nx = s.nxg, h = nx / 2, nyl = s.nyl;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 1) {
        if (t <= s.nxl + 1)
            for (int q = 0; q < nf; q++) FD(s, f + q, cell(s, t, 0)) = fill;
        if (t >= 1 && t >= ew_from && t <= nyl - 1)
            for (int q = 0; q < nf; q++) {
                FD(s, f + q, cell(s, 0, t)) = cyclic ? FD(s, f + q, cell(s, s.nxl, t)) : fill;
                FD(s, f + q, cell(s, s.nxl + 1, t)) = cyclic ? FD(s, f + q, cell(s, 1, t)) : fill;
            }
        return;
    }
    if (t > h) return;
    // new values of column g: top row = sgn * v(src), ghost row = sgn * (row nyl-1)(src), src = nx - g (0 -> nx)
    // (plain macros, not lambdas: the compiler kept the lambdas' closures -- references to f, fprev, nyl -- in scratch memory)
#define TNE_PUT(q, g, topv, ghostv)                                                                                          \
    do {                                                                                                                     \
        FD(s, f + (q), cell(s, (g), nyl)) = (topv);                                                                          \
        FD(s, f + (q), cell(s, (g), nyl + 1)) = (ghostv);                                                                    \
        /* E-W ghost columns of the two rows (k_halo_ew_local ran after the fold): copies of columns nx and 1 */             \
        if ((g) == nx) { FD(s, f + (q), cell(s, 0, nyl)) = cyclic ? (topv) : fill; FD(s, f + (q), cell(s, 0, nyl + 1)) = cyclic ? (ghostv) : fill; }          \
        if ((g) == 1) { FD(s, f + (q), cell(s, nx + 1, nyl)) = cyclic ? (topv) : fill; FD(s, f + (q), cell(s, nx + 1, nyl + 1)) = cyclic ? (ghostv) : fill; } \
    } while (0)
    // old top-row value of column g: an inactive cell carries the previous update's
#define TNE_TOP(q, g) FD(s, ((fprev >= 0 && !(s.cmask[mcell(s, (g), nyl)] & CM_U)) ? fprev : f) + (q), cell(s, (g), nyl))
    for (int q = 0; q < nf; q++) {
        if (t == 0 || t == h) {
            const int g = t == 0 ? nx : h;                  // src == g: the value itself
            const double T = TNE_TOP(q, g), R = FD(s, f + q, cell(s, g, nyl - 1));
            { const double tv = sgn * T, gv = sgn * R; TNE_PUT(q, g, tv, gv); }
        } else {
            const int ga = t, gb = nx - t;                   // ga in 1..h-1, gb in h+1..nx-1
            const double Ta = TNE_TOP(q, ga), Tb = TNE_TOP(q, gb);
            const double Ra = FD(s, f + q, cell(s, ga, nyl - 1)), Rb = FD(s, f + q, cell(s, gb, nyl - 1));
            // column ga: src = gb >= h+1:  v = sgn*(0.5*(B2[nx-src] + sgn*B2[src]));  column gb: src = ga:  v = 0.5*(B2[src] + sgn*B2[nx-src])
            const double va = sgn * (0.5 * (Ta + sgn * Tb));
            const double vb = 0.5 * (Ta + sgn * Tb);
            { const double tv = sgn * va, gv = sgn * Rb; TNE_PUT(q, ga, tv, gv); }
            { const double tv = sgn * vb, gv = sgn * Ra; TNE_PUT(q, gb, tv, gv); }
        }
    }
#undef TNE_PUT
#undef TNE_TOP
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

// ------------------------------------------------------------------------------------
// Ghost zones of the two-subcycle kernel on x-slabs, W = 2..8 columns per side: columns 1..W go to the west neighbour
// (its columns nxl+1..nxl+W), columns nxl-W+1..nxl to the east neighbour (its 1-W..0), for a list of pair planes
// (+ optionally the byte mask as a pseudo plane), all rows or a compacted row list per side.
// ------------------------------------------------------------------------------------
struct PairList { int n; int with_cmask; int p[24]; };

// rows taken by one side of a ghost-zone message: all rows (idx == nullptr, n = nyl+2) or a compacted list
struct RowSet { const int *idx; int n; };
struct ZoneRows { RowSet sW, sE, rE, rW; };     // my W edge -> west, my E edge -> east, east zone <- east, west zone <- west

// message layout: buf[(q*n + jj)*W + k] of double2, q = plane, jj = position in the row set, k = column (ascending)
__global__ void k_cols_pack(Slab s, PairList pl, int W, ZoneRows zr, double2 *sendW, double2 *sendE) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = idx % W, jj = idx / W;
    const double2 *F2 = reinterpret_cast<const double2 *>(s.F);
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const RowSet rs = side ? zr.sE : zr.sW;
        if (jj >= rs.n) continue;
        const int j = rs.idx ? rs.idx[jj] : jj;
        const int col = (side ? s.nxl - W + 1 : 1) + k;
        double2 *dst = side ? sendE : sendW;
        for (int q = 0; q < pl.n; q++)
            dst[((size_t)q * rs.n + jj) * W + k] = F2[(size_t)j * s.rstride + (size_t)pl.p[q] * s.pitch + C0 + col];
        if (pl.with_cmask)
            dst[((size_t)pl.n * rs.n + jj) * W + k] = make_double2((double)s.cmask[(size_t)j * s.pitch + C0 + col], 0.0);
    }
}

__global__ void k_cols_unpack(Slab s, PairList pl, int W, ZoneRows zr, const double2 *recvW, const double2 *recvE, int haveW, int haveE) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = idx % W, jj = idx / W;
    double2 *F2 = reinterpret_cast<double2 *>(s.F);
#pragma unroll
    for (int side = 0; side < 2; side++) {
        if (!(side ? haveE : haveW)) continue;
        const RowSet rs = side ? zr.rE : zr.rW;
        if (jj >= rs.n) continue;
        const int j = rs.idx ? rs.idx[jj] : jj;
        const int col = (side ? s.nxl + 1 : 1 - W) + k;
        const double2 *src = side ? recvE : recvW;
        for (int q = 0; q < pl.n; q++)
            F2[(size_t)j * s.rstride + (size_t)pl.p[q] * s.pitch + C0 + col] = src[((size_t)q * rs.n + jj) * W + k];
        if (pl.with_cmask)
            s.cmask[(size_t)j * s.pitch + C0 + col] = (unsigned char)src[((size_t)pl.n * rs.n + jj) * W + k].x;
    }
}

// rows of the four W-column windows that hold an active cell (flags[4][nyl+2]: send W, send E, recv E, recv W).  A row
// without one is never written by the subcycle kernels, so its zone image stays what the full exchange at prep made it.
__global__ void k_zone_rows(Slab s, int W, unsigned char *flags) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > s.nyl + 1) return;
    const int rows = s.nyl + 2;
    const unsigned char *m = s.cmask + (size_t)j * s.pitch + C0;
    const int c0[4] = {1, s.nxl - W + 1, s.nxl + 1, 1 - W};
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned char any = 0;
        for (int k = 0; k < W; k++) any |= m[c0[w] + k];
        flags[(size_t)w * rows + j] = any ? 1 : 0;
    }
}

// zone columns of nf/2 pair planes, all rows: fdst := fsrc (the other state buffer starts from the same zone image)
__global__ void k_zone_copy(Slab s, int W, int fsrc, int fdst, int npairs) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = idx % (2 * W), j = idx / (2 * W);
    if (j > s.nyl + 1) return;
    const int col = k < W ? 1 - W + k : s.nxl + 1 + (k - W);
    double2 *F2 = reinterpret_cast<double2 *>(s.F);
    for (int q = 0; q < npairs; q++)
        F2[(size_t)j * s.rstride + (size_t)((fdst >> 1) + q) * s.pitch + C0 + col] =
            F2[(size_t)j * s.rstride + (size_t)((fsrc >> 1) + q) * s.pitch + C0 + col];
}

// one wave per strip: is there any T work (cols cx*63+1..+64, rows jb..jb+R) or U work?
// Also counts active cells: T on physical cells, U.
// flags / counts / nact may be null: nact counts the active strips (the strip-height tuner of the one-subcycle kernels)
__global__ void k_strip_flags(Slab s, int ncx, int nry, int R, unsigned char *flags, unsigned long long *counts, unsigned int *nact = nullptr) {
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
            if ((m & CM_U) && lane < STRIP_W && jj < R && i <= s.nxl && j <= s.nyl) { any = 1; nu++; }
        }
    }
    const unsigned long long b = __ballot(any);
    for (int o = 32; o > 0; o >>= 1) { nt += __shfl_down(nt, o); nu += __shfl_down(nu, o); }
    if (lane == 0) {
        if (flags) flags[sid] = b ? 1 : 0;
        if (nact && b) atomicAdd(nact, 1u);
        if (counts) {
            if (nt) atomicAdd(&counts[0], (unsigned long long)nt);
            if (nu) atomicAdd(&counts[1], (unsigned long long)nu);
        }
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
    int nstrips, ncx, R, wrap;
    int sr, sw;        // field ids of the state buffer read / written (F_STATE0, F_STATE1 or F_STATE2)
    int jb0;           // > 0: band launch -- every strip starts at row jb0 (tripole top band), strips[] holds cx only
    int jmax;          // k_subcycle2: rows above are not stored (tripole, single rank: the band launches own rows >= nyl-1)
    int G;             // k_subcycle2 in ghost-zone mode: columns 1-G .. nxl+G are advanced (zones of G+2 columns per side)
    const Slab *xm;    // != nullptr (x-slab ranks on a tripole grid): the mirror slab M of band_pair, a Slab struct in device memory
    __device__ const Slab *xm_slab() const { return xm; }
    const int *nsdev;  // != nullptr: the number of entries of strips[] lives on the device (k_compact_strips wrote list and count; the
                       // host launches for the upper bound nstrips and never waits for the count): one rank, the pair kernels
    int nmir, mjmax;   // > 0 (x-slab ranks on a tripole grid, XM kernels): the nmir strips of the mirror slab *xm (identity list, the
                       // same ncx / R / G) are advanced by workgroups of THIS launch, after the band workgroups and before the slab's
                       // own strips, storing rows <= mjmax -- M's advance between two refreshes costs no launch of its own
    int nband;         // > 0 (tripole, one rank, cyclic E-W): the first nband8 = 8*ceil(nband/8) workgroups of the launch are the
                       // tripole top band (band_pair); the strips follow
    int band_last;     // != 0: the band (and mirror-slab) workgroups are the LAST of the grid instead of the first, so that the strips take
                       // the workgroup slots first and the short band workgroups run in the slots the first strips free (round 5)
    int prio;          // k_subcycle2p: != 0 -- a wave lowers its issue priority as it advances (3, 2, 1, 0 over the quarters of its march), so
                       // that of the two waves of a SIMD the one that is BEHIND wins the VALU arbitration (by default the older wave wins
                       // every time, finishes a third earlier and leaves the younger one alone on the SIMD, where a single dependent
                       // fp64 stream issues at 5.3 us per step instead of 3.4 for two)
    unsigned long long *dbg;   // != nullptr (EVPK_DEBUG_CLOCKS, one launch per evp): per strip of the list {start, end} of its wave in
                               // s_memrealtime ticks (100 MHz) and the XCC / CU it ran on -- the launch's timeline (scripts/k_timeline.py)
};
__global__ void k_dbg_clock(unsigned long long *out) { if (threadIdx.x == 0) *out = wall_clock64(); }
__device__ __forceinline__ void dbg_stamp(const SubArgs &a, int sid, int which) {
    if (a.dbg && (threadIdx.x & 63) == 0 && sid < 65535) {      // (the buffer holds 65536 entries; the last one is k_dbg_clock's)
        a.dbg[(size_t)sid * 4 + which] = wall_clock64();
        if (which == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            a.dbg[(size_t)sid * 4 + 2] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
}

// Neighbour-lane exchange as DPP whole-wave shifts (v_mov_b32_dpp wave_shl:1 / wave_shr:1, VALU only) instead of
// ds_bpermute through the LDS crossbar: no lgkmcnt wait in the dependent stress -> stepu -> stress chain.  The lane
// without a source (63 / 0) keeps its own value, as __shfl_down / __shfl_up do.  Call in wave-uniform control flow.
__device__ __forceinline__ double shfl_dn1(double x) {        // lane i <- lane i+1
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_up1(double x) {        // lane i <- lane i-1
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

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

struct TMet { double cxp, cyp, cxm, cym, dxt, dyt, dxhy, dyhx, tiny, strength; };
struct Sig { double sp1, sp2, sp3, sp4, sm1, sm2, sm3, sm4, s121, s122, s123, s124; };
struct Str8 { double s1, s2, s3, s4, s5, s6, s7, s8; };
struct Diag { double divu, rdg_conv, rdg_shear, shear, prs; };

__device__ __forceinline__ TMet load_tmet(const char *rb, size_t pp, unsigned lo) {
    const double2 cp = ldp(rb, pp, F_CXP, lo), cm = ldp(rb, pp, F_CXM, lo);
    const double2 dd = ldp(rb, pp, F_DXT, lo), dh = ldp(rb, pp, F_DXHY, lo);
    const double2 ts = ldp(rb, pp, F_TINYAREA, lo);
    return TMet{cp.x, cp.y, cm.x, cm.y, dd.x, dd.y, dh.x, dh.y, ts.x, ts.y};
}
__device__ __forceinline__ Sig load_sig(const char *rb, size_t pp, int SB, unsigned lo) {
    const double2 q0 = ldp(rb, pp, SB + S_SP, lo), q1 = ldp(rb, pp, SB + S_SP + 2, lo);
    const double2 q2 = ldp(rb, pp, SB + S_SM, lo), q3 = ldp(rb, pp, SB + S_SM + 2, lo);
    const double2 q4 = ldp(rb, pp, SB + S_S12, lo), q5 = ldp(rb, pp, SB + S_S12 + 2, lo);
    return Sig{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y, q4.x, q4.y, q5.x, q5.y};
}
__device__ __forceinline__ void store_sig(char *rb, size_t pp, int SB, unsigned lo, const Sig &g) {
    stp(rb, pp, SB + S_SP, lo, g.sp1, g.sp2);     stp(rb, pp, SB + S_SP + 2, lo, g.sp3, g.sp4);
    stp(rb, pp, SB + S_SM, lo, g.sm1, g.sm2);     stp(rb, pp, SB + S_SM + 2, lo, g.sm3, g.sm4);
    stp(rb, pp, SB + S_S12, lo, g.s121, g.s122);  stp(rb, pp, SB + S_S12 + 2, lo, g.s123, g.s124);
}

// ---- fp64 sqrt and divide, FOUR AT A TIME (round 4) ----
// hipcc expands sqrt() and `/` of doubles into dependent chains of 17 and 11 instructions (v_rsq_f64 / v_rcp_f64, Newton steps as
// fma, v_div_scale / v_div_fmas / v_div_fixup) and emits the four Deltas and the four divisions of a T cell ONE AFTER THE OTHER: 112
// instructions in which every one waits for the one before it (the profile of k_subcycle2p: one wave alone on its SIMD issues one
// VALU instruction per 8.4 cycles, a pair per 5.5, where the pipe takes one per 4).  Here the same sequences -- instruction for
// instruction what the compiler emits (AMDGPU lowering of fsqrt.f64 / fdiv.f64), hence the same correctly rounded results -- are
// written out for four operands side by side, so that each instruction has three independent neighbours between itself and its
// consumer.  tests: every parity test runs through them (bit-identical with the oracle's sqrt() and `/`).
__device__ __forceinline__ void sqrt4_f64(double &x0, double &x1, double &x2, double &x3) {
    const double tiny = 0x1.0p-767;
    const bool s0 = x0 < tiny, s1 = x1 < tiny, s2 = x2 < tiny, s3 = x3 < tiny;
    const double a0 = __builtin_amdgcn_ldexp(x0, s0 ? 256 : 0), a1 = __builtin_amdgcn_ldexp(x1, s1 ? 256 : 0);
    const double a2 = __builtin_amdgcn_ldexp(x2, s2 ? 256 : 0), a3 = __builtin_amdgcn_ldexp(x3, s3 ? 256 : 0);
    const double y0 = __builtin_amdgcn_rsq(a0), y1 = __builtin_amdgcn_rsq(a1), y2 = __builtin_amdgcn_rsq(a2), y3 = __builtin_amdgcn_rsq(a3);
    double g0 = a0 * y0, g1 = a1 * y1, g2 = a2 * y2, g3 = a3 * y3;
    double h0 = y0 * 0.5, h1 = y1 * 0.5, h2 = y2 * 0.5, h3 = y3 * 0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5), r1 = __builtin_fma(-h1, g1, 0.5), r2 = __builtin_fma(-h2, g2, 0.5), r3 = __builtin_fma(-h3, g3, 0.5);
    g0 = __builtin_fma(g0, r0, g0); g1 = __builtin_fma(g1, r1, g1); g2 = __builtin_fma(g2, r2, g2); g3 = __builtin_fma(g3, r3, g3);
    double d0 = __builtin_fma(-g0, g0, a0), d1 = __builtin_fma(-g1, g1, a1), d2 = __builtin_fma(-g2, g2, a2), d3 = __builtin_fma(-g3, g3, a3);
    h0 = __builtin_fma(h0, r0, h0); h1 = __builtin_fma(h1, r1, h1); h2 = __builtin_fma(h2, r2, h2); h3 = __builtin_fma(h3, r3, h3);
    g0 = __builtin_fma(d0, h0, g0); g1 = __builtin_fma(d1, h1, g1); g2 = __builtin_fma(d2, h2, g2); g3 = __builtin_fma(d3, h3, g3);
    d0 = __builtin_fma(-g0, g0, a0); d1 = __builtin_fma(-g1, g1, a1); d2 = __builtin_fma(-g2, g2, a2); d3 = __builtin_fma(-g3, g3, a3);
    g0 = __builtin_fma(d0, h0, g0); g1 = __builtin_fma(d1, h1, g1); g2 = __builtin_fma(d2, h2, g2); g3 = __builtin_fma(d3, h3, g3);
    g0 = __builtin_amdgcn_ldexp(g0, s0 ? -128 : 0); g1 = __builtin_amdgcn_ldexp(g1, s1 ? -128 : 0);
    g2 = __builtin_amdgcn_ldexp(g2, s2 ? -128 : 0); g3 = __builtin_amdgcn_ldexp(g3, s3 ? -128 : 0);
    const int zi = 0x260;       // +inf | +0 | -0: sqrt(x) = x
    x0 = __builtin_amdgcn_class(a0, zi) ? a0 : g0; x1 = __builtin_amdgcn_class(a1, zi) ? a1 : g1;
    x2 = __builtin_amdgcn_class(a2, zi) ? a2 : g2; x3 = __builtin_amdgcn_class(a3, zi) ? a3 : g3;
}
// q_k = n / d_k
__device__ __forceinline__ void div4_f64(double n, double d0, double d1, double d2, double d3, double &q0, double &q1, double &q2, double &q3) {
    bool f0, f1, f2, f3, u0, u1, u2, u3;
    const double e0 = __builtin_amdgcn_div_scale(n, d0, false, &u0), e1 = __builtin_amdgcn_div_scale(n, d1, false, &u1);
    const double e2 = __builtin_amdgcn_div_scale(n, d2, false, &u2), e3 = __builtin_amdgcn_div_scale(n, d3, false, &u3);
    double r0 = __builtin_amdgcn_rcp(e0), r1 = __builtin_amdgcn_rcp(e1), r2 = __builtin_amdgcn_rcp(e2), r3 = __builtin_amdgcn_rcp(e3);
    double t0 = __builtin_fma(-e0, r0, 1.0), t1 = __builtin_fma(-e1, r1, 1.0), t2 = __builtin_fma(-e2, r2, 1.0), t3 = __builtin_fma(-e3, r3, 1.0);
    r0 = __builtin_fma(r0, t0, r0); r1 = __builtin_fma(r1, t1, r1); r2 = __builtin_fma(r2, t2, r2); r3 = __builtin_fma(r3, t3, r3);
    t0 = __builtin_fma(-e0, r0, 1.0); t1 = __builtin_fma(-e1, r1, 1.0); t2 = __builtin_fma(-e2, r2, 1.0); t3 = __builtin_fma(-e3, r3, 1.0);
    r0 = __builtin_fma(r0, t0, r0); r1 = __builtin_fma(r1, t1, r1); r2 = __builtin_fma(r2, t2, r2); r3 = __builtin_fma(r3, t3, r3);
    const double m0 = __builtin_amdgcn_div_scale(n, d0, true, &f0), m1 = __builtin_amdgcn_div_scale(n, d1, true, &f1);
    const double m2 = __builtin_amdgcn_div_scale(n, d2, true, &f2), m3 = __builtin_amdgcn_div_scale(n, d3, true, &f3);
    const double p0 = m0 * r0, p1 = m1 * r1, p2 = m2 * r2, p3 = m3 * r3;
    t0 = __builtin_fma(-e0, p0, m0); t1 = __builtin_fma(-e1, p1, m1); t2 = __builtin_fma(-e2, p2, m2); t3 = __builtin_fma(-e3, p3, m3);
    q0 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(t0, r0, p0, f0), d0, n);
    q1 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(t1, r1, p1, f1), d1, n);
    q2 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(t2, r2, p2, f2), d2, n);
    q3 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(t3, r3, p3, f3), d3, n);
}
// (q0, q1) = (n0, n1) / d: two quotients by one denominator, side by side (stepu)
__device__ __forceinline__ void div2_f64(double n0, double n1, double d, double &q0, double &q1) {
    bool f0, f1, u0, u1;
    const double e0 = __builtin_amdgcn_div_scale(n0, d, false, &u0), e1 = __builtin_amdgcn_div_scale(n1, d, false, &u1);
    double r0 = __builtin_amdgcn_rcp(e0), r1 = __builtin_amdgcn_rcp(e1);
    double t0 = __builtin_fma(-e0, r0, 1.0), t1 = __builtin_fma(-e1, r1, 1.0);
    r0 = __builtin_fma(r0, t0, r0); r1 = __builtin_fma(r1, t1, r1);
    t0 = __builtin_fma(-e0, r0, 1.0); t1 = __builtin_fma(-e1, r1, 1.0);
    r0 = __builtin_fma(r0, t0, r0); r1 = __builtin_fma(r1, t1, r1);
    const double m0 = __builtin_amdgcn_div_scale(n0, d, true, &f0), m1 = __builtin_amdgcn_div_scale(n1, d, true, &f1);
    const double p0 = m0 * r0, p1 = m1 * r1;
    t0 = __builtin_fma(-e0, p0, m0); t1 = __builtin_fma(-e1, p1, m1);
    q0 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(t0, r0, p0, f0), d, n0);
    q1 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(t1, r1, p1, f1), d, n1);
}

// ---- stress of one T cell (ice_dyn_evp.F90:618-847), the reference's operation order ----
// u/v naming: _ij = (i,j), _mj = (i-1,j), _im = (i,j-1), _mm = (i-1,j-1).  g: sigma in -> sigma out.
// ILP: the four Deltas and the four divisions through sqrt4_f64 / div4_f64 (the kernels with registers to spare for it)
template <bool DIAG, bool ILP = false>
__device__ __forceinline__ void stress_cell(const TMet &m, double u_ij, double u_mj, double u_im, double u_mm,
                                            double v_ij, double v_mj, double v_im, double v_mm,
                                            double ecci, double arlx1i, double denom1, double tarear,
                                            Sig &g, Str8 &o, Diag &dg) {
    const double p111 = 1.0 / 9.0, p055 = p111 * 0.5, p027 = p055 * 0.5;
    const double p166 = 1.0 / 6.0, p222 = 2.0 / 9.0, p333 = 1.0 / 3.0;
    const double cxp = m.cxp, cyp = m.cyp, cxm = m.cxm, cym = m.cym, dxt = m.dxt, dyt = m.dyt, dxhy = m.dxhy, dyhx = m.dyhx;

    // strain rates * area (:627-654)
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
    double Deltane = divune * divune + ecci * (tensionne * tensionne + shearne * shearne);
    double Deltanw = divunw * divunw + ecci * (tensionnw * tensionnw + shearnw * shearnw);
    double Deltase = divuse * divuse + ecci * (tensionse * tensionse + shearse * shearse);
    double Deltasw = divusw * divusw + ecci * (tensionsw * tensionsw + shearsw * shearsw);
    if (ILP) sqrt4_f64(Deltane, Deltanw, Deltase, Deltasw);
    else { Deltane = sqrt(Deltane); Deltanw = sqrt(Deltanw); Deltase = sqrt(Deltase); Deltasw = sqrt(Deltasw); }

    if (DIAG) {                                                                     // :665-677
        dg.divu = 0.25 * (divune + divunw + divuse + divusw) * tarear;
        const double tmp = 0.25 * (Deltane + Deltanw + Deltase + Deltasw) * tarear;
        dg.rdg_conv = -fmin(dg.divu, 0.0);
        dg.rdg_shear = 0.5 * (tmp - fabs(dg.divu));
        const double tt = tensionne + tensionnw + tensionse + tensionsw;
        const double ss = shearne + shearnw + shearse + shearsw;
        dg.shear = 0.25 * tarear * sqrt(tt * tt + ss * ss);
    }

    // replacement pressure / Delta (:683-697)
    double c0ne, c0nw, c0sw, c0se;
    if (ILP) div4_f64(m.strength, fmax(Deltane, m.tiny), fmax(Deltanw, m.tiny), fmax(Deltasw, m.tiny), fmax(Deltase, m.tiny), c0ne, c0nw, c0sw, c0se);
    else {
        c0ne = m.strength / fmax(Deltane, m.tiny);
        c0nw = m.strength / fmax(Deltanw, m.tiny);
        c0sw = m.strength / fmax(Deltasw, m.tiny);
        c0se = m.strength / fmax(Deltase, m.tiny);
    }
    if (DIAG) dg.prs = c0ne * Deltane;
    const double c1ne = c0ne * arlx1i, c1nw = c0nw * arlx1i, c1sw = c0sw * arlx1i, c1se = c0se * arlx1i;
    c0ne = c1ne * ecci; c0nw = c1nw * ecci; c0sw = c1sw * ecci; c0se = c1se * ecci;

    // the stresses (:704-721)
    const double sp1 = (g.sp1 + c1ne * (divune - Deltane)) * denom1;
    const double sp2 = (g.sp2 + c1nw * (divunw - Deltanw)) * denom1;
    const double sp3 = (g.sp3 + c1sw * (divusw - Deltasw)) * denom1;
    const double sp4 = (g.sp4 + c1se * (divuse - Deltase)) * denom1;
    const double sm1 = (g.sm1 + c0ne * tensionne) * denom1;
    const double sm2 = (g.sm2 + c0nw * tensionnw) * denom1;
    const double sm3 = (g.sm3 + c0sw * tensionsw) * denom1;
    const double sm4 = (g.sm4 + c0se * tensionse) * denom1;
    const double s121 = (g.s121 + c0ne * shearne * 0.5) * denom1;
    const double s122 = (g.s122 + c0nw * shearnw * 0.5) * denom1;
    const double s123 = (g.s123 + c0sw * shearsw * 0.5) * denom1;
    const double s124 = (g.s124 + c0se * shearse * 0.5) * denom1;
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

// ---- stepu of one U cell (ice_dyn_shared.F90:700-746) ----
// sx = (((str1(i,j) + str2(i+1,j)) + str3(i,j+1)) + str4(i+1,j+1)), sy likewise with str5,6,7,8 (:725-728)
struct UStat { double vrelc, uarear, uocn, vocn, forcex, forcey, umassdti, fm; };
__device__ __forceinline__ UStat load_ustat(const char *ru, size_t pp, unsigned lo) {
    const double2 va = ldp(ru, pp, F_VRELC, lo), oc = ldp(ru, pp, F_UOCN, lo);
    const double2 fo = ldp(ru, pp, F_FORCEX, lo), mf = ldp(ru, pp, F_UMASSDTI, lo);
    return UStat{va.x, va.y, oc.x, oc.y, fo.x, fo.y, mf.x, mf.y};
}
template <bool ILP = false>
__device__ __forceinline__ void stepu_cell(const UStat &q, double uold, double vold, double ui, double vi,
                                           double sx, double sy, double brlx, double revp, double cosw, double sinw,
                                           double &un, double &vn, double &strintx, double &strinty) {
    // waterx/y of evp_prep2 (:592-593) recomputed instead of loaded: same expression, same bits
    const double sgf = copysign(1.0, q.fm);
    const double waterx = q.uocn * cosw - q.vocn * sinw * sgf;
    const double watery = q.vocn * cosw + q.uocn * sinw * sgf;
    const double du = q.uocn - uold, dv = q.vocn - vold;
    const double vrel = q.vrelc * sqrt(du * du + dv * dv);                        // :708-709
    const double taux = vrel * waterx, tauy = vrel * watery;                        // :711-712
    const double cca = (brlx + revp) * q.umassdti + vrel * cosw;                    // :715
    const double ccb = q.fm + sgf * vrel * sinw;                                    // :720
    const double ab2 = cca * cca + ccb * ccb;
    strintx = q.uarear * sx;                                                        // :725-728
    strinty = q.uarear * sy;
    const double cc1 = strintx + q.forcex + taux + q.umassdti * (brlx * uold + revp * ui);   // :731-734
    const double cc2 = strinty + q.forcey + tauy + q.umassdti * (brlx * vold + revp * vi);
    if (ILP) div2_f64(cca * cc1 + ccb * cc2, cca * cc2 - ccb * cc1, ab2, un, vn);
    else {
        un = (cca * cc1 + ccb * cc2) / ab2;                                         // :736-737
        vn = (cca * cc2 - ccb * cc1) / ab2;
    }
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
    const int jb = a.jb0 > 0 ? a.jb0 : ry * R + 1;
    const bool colT = (i <= s.nxl + 1);               // lane has a T column
    const bool ownT = colT && (lane < STRIP_W);       // ... and owns its sigma stores
    const bool colU = (i <= s.nxl) && (lane < STRIP_W);

    const size_t pp = (size_t)s.pitch * 16;           // bytes per row of one pair plane
    const size_t rowb = (size_t)s.rstride * 16;       // bytes per row of all planes
    const unsigned lo = (unsigned)(C0 + i) * 16u;     // lane byte offset inside a pair-plane row
    const int SR = a.sr;                              // read buffer
    const int SW = a.sw;                              // write buffer
    char *const base = reinterpret_cast<char *>(s.F);

    // carried from the previous row (j-1)
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
        char *const rb = base + (size_t)j * rowb;     // wave-uniform
        unsigned char m = 0;
        double u_ij = 0.0, u_mj = 0.0, v_ij = 0.0, v_mj = 0.0;
        if (colT) {
            m = s.cmask[(size_t)j * s.pitch + C0 + i];
            const double2 a0 = ldp(rb, pp, SR + S_U, lo), a1 = ldp(rb, pp, SR + S_U, lo - 16u);
            u_ij = a0.x; v_ij = a0.y; u_mj = a1.x; v_mj = a1.y;
        }
        const bool tact = (m & CM_T) != 0;
        Str8 o{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};

        if (__any(tact)) {
            if (tact) {
                const TMet mt = load_tmet(rb, pp, lo);
                Sig g = load_sig(rb, pp, SR, lo);
                const bool store = ownT && (jj < R);
                double tarear = 0.0;
                if (LAST) tarear = *reinterpret_cast<const double *>(rb + (size_t)(F_TAREAR >> 1) * pp + lo + (F_TAREAR & 1) * 8);
                Diag dg;
                stress_cell<LAST>(mt, u_ij, u_mj, u_im, u_mm, v_ij, v_mj, v_im, v_mm, a.ecci, a.arlx1i, a.denom1, tarear, g, o, dg);
                if (store) {
                    store_sig(rb, pp, SW, lo, g);
                    if (LAST) {
                        st1(rb, pp, F_DIVU, lo, dg.divu);       st1(rb, pp, F_RDGCONV, lo, dg.rdg_conv);
                        st1(rb, pp, F_RDGSHEAR, lo, dg.rdg_shear); st1(rb, pp, F_SHEAR, lo, dg.shear);
                        st1(rb, pp, F_PRSSIG, lo, dg.prs);
                    }
                }
            }
        }

        // east neighbour's contributions of this T row
        const double s2n = shfl_dn1(o.s2), s4n = shfl_dn1(o.s4), s7n = shfl_dn1(o.s7), s8n = shfl_dn1(o.s8);

        // stepu for U(i, j-1)
        if (jj >= 1) {
            const bool uact = colU && ((mprev & CM_U) != 0);
            if (__any(uact)) {
                if (uact) {
                    char *const ru = rb - rowb;
                    const UStat q = load_ustat(ru, pp, lo);
                    double ui = 0.0, vi = 0.0;
                    if (REVP) { const double2 iv = ldp(ru, pp, F_UVEL_INIT, lo); ui = iv.x; vi = iv.y; }
                    double un, vn, strintx, strinty;
                    stepu_cell(q, u_im, v_im, ui, vi, ((s1c + s2r) + o.s3) + s4n, ((s5c + o.s6) + s7r) + s8n,
                               a.brlx, a.revp, a.cosw, a.sinw, un, vn, strintx, strinty);
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
        s1c = o.s1; s5c = o.s5; s2r = s2n; s7r = s7n;
        u_im = u_ij; u_mm = u_mj; v_im = v_ij; v_mm = v_mj;
        mprev = m;
    }
}

template __global__ void k_subcycle<false, false>(SubArgs);
template __global__ void k_subcycle<true, false>(SubArgs);
template __global__ void k_subcycle<false, true>(SubArgs);
template __global__ void k_subcycle<true, true>(SubArgs);

// ------------------------------------------------------------------------------------
// k_subcycle_t: k_subcycle without the north march, for launches of a few rows (the tripole top-band launches: 3-4 rows
// across the whole slab, on the critical path of every pair of subcycles).  A workgroup of R + 1 waves takes one strip,
// wave w the T row j = jb + w: stress of T(i, j) -> LDS (str3, str6 and the east cell's str4, str8) -> one workgroup barrier
// -> U(i, j) from its own row and the row above.  One row's latency instead of R + 1 march steps; same arithmetic.
// ------------------------------------------------------------------------------------
template <bool LAST, bool REVP>
__global__ __launch_bounds__(512) void k_subcycle_t(SubArgs a) {
    extern __shared__ double tl[];
    const Slab &s = a.s;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sid = blockIdx.x;
    if (sid >= a.nstrips) return;
    const int st = __builtin_amdgcn_readfirstlane(a.strips[sid]);
    const int cx = st % a.ncx, ry = st / a.ncx;
    const int R = a.R;
    const int i = cx * STRIP_W + 1 + lane;
    const int jb = a.jb0 > 0 ? a.jb0 : ry * R + 1;
    const int j = jb + w;                                 // this wave's T row (and U row, for w < R)
    const bool rowok = (j <= s.nyl + 1);
    const bool colT = (i <= s.nxl + 1) && rowok;
    const bool ownT = colT && (lane < STRIP_W);
    const bool colU = (i <= s.nxl) && (lane < STRIP_W) && rowok;
    const size_t pp = (size_t)s.pitch * 16;
    const size_t rowb = (size_t)s.rstride * 16;
    const unsigned lo = (unsigned)(C0 + i) * 16u;
    const int SR = a.sr, SW = a.sw;
    char *const base = reinterpret_cast<char *>(s.F);
    char *const rb = base + (size_t)(rowok ? j : 0) * rowb;
    double *const Xw = tl + (size_t)w * 256 + lane;

    unsigned char m = 0;
    double u_ij = 0.0, u_mj = 0.0, v_ij = 0.0, v_mj = 0.0, u_im = 0.0, u_mm = 0.0, v_im = 0.0, v_mm = 0.0;
    if (colT) {
        m = s.cmask[(size_t)j * s.pitch + C0 + i];
        const double2 a0 = ldp(rb, pp, SR + S_U, lo), a1 = ldp(rb, pp, SR + S_U, lo - 16u);
        u_ij = a0.x; v_ij = a0.y; u_mj = a1.x; v_mj = a1.y;
        const char *rs = rb - rowb;                       // j >= 1: jb >= 1
        const double2 b0 = ldp(rs, pp, SR + S_U, lo), b1 = ldp(rs, pp, SR + S_U, lo - 16u);
        u_im = b0.x; v_im = b0.y; u_mm = b1.x; v_mm = b1.y;
    }
    const bool tact = (m & CM_T) != 0;
    const bool uact = colU && (w < R) && (m & CM_U) != 0;
    Str8 o{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    UStat q{0, 0, 0, 0, 0, 0, 0, 0};
    double ui = 0.0, vi = 0.0;
    if (__any(uact)) {
        if (uact) {
            q = load_ustat(rb, pp, lo);
            if (REVP) { const double2 iv = ldp(rb, pp, F_UVEL_INIT, lo); ui = iv.x; vi = iv.y; }
        }
    }
    if (__any(tact)) {
        if (tact) {
            const TMet mt = load_tmet(rb, pp, lo);
            Sig g = load_sig(rb, pp, SR, lo);
            double tarear = 0.0;
            if (LAST) tarear = *reinterpret_cast<const double *>(rb + (size_t)(F_TAREAR >> 1) * pp + lo + (F_TAREAR & 1) * 8);
            Diag dg;
            stress_cell<LAST>(mt, u_ij, u_mj, u_im, u_mm, v_ij, v_mj, v_im, v_mm, a.ecci, a.arlx1i, a.denom1, tarear, g, o, dg);
            if (ownT && w < R) {
                store_sig(rb, pp, SW, lo, g);
                if (LAST) {
                    st1(rb, pp, F_DIVU, lo, dg.divu);       st1(rb, pp, F_RDGCONV, lo, dg.rdg_conv);
                    st1(rb, pp, F_RDGSHEAR, lo, dg.rdg_shear); st1(rb, pp, F_SHEAR, lo, dg.shear);
                    st1(rb, pp, F_PRSSIG, lo, dg.prs);
                }
            }
        }
    }
    const double s2n = shfl_dn1(o.s2), s4n = shfl_dn1(o.s4), s7n = shfl_dn1(o.s7), s8n = shfl_dn1(o.s8);
    Xw[0] = o.s3; Xw[64] = o.s6; Xw[128] = s4n; Xw[192] = s8n;
    __syncthreads();
    if (__any(uact)) {
        if (uact) {
            const double *Xn = Xw + 256;                  // the T row above
            double un, vn, strintx, strinty;
            stepu_cell(q, u_ij, v_ij, ui, vi, ((o.s1 + s2n) + Xn[0]) + Xn[128], ((o.s5 + Xn[64]) + s7n) + Xn[192],
                       a.brlx, a.revp, a.cosw, a.sinw, un, vn, strintx, strinty);
            stp(rb, pp, SW + S_U, lo, un, vn);
            if (a.wrap) {
                if (i == 1) stp(rb, pp, SW + S_U, lo + (unsigned)s.nxl * 16u, un, vn);
                if (i == s.nxl) stp(rb, pp, SW + S_U, lo - (unsigned)s.nxl * 16u, un, vn);
            }
            if (LAST) {
                st1(rb, pp, F_STRINTX, lo, strintx);
                st1(rb, pp, F_STRINTY, lo, strinty);
            }
        }
    }
}

template __global__ void k_subcycle_t<false, false>(SubArgs);
template __global__ void k_subcycle_t<true, false>(SubArgs);
template __global__ void k_subcycle_t<false, true>(SubArgs);
template __global__ void k_subcycle_t<true, true>(SubArgs);

// ------------------------------------------------------------------------------------
// TWO subcycles per launch (temporal blocking): sigma, the grid metrics and the stepu input planes
// cross HBM once per two subcycles.  Single rank, no tripole fold (the fold needs the mirrored
// columns between the two subcycles).  One wave = 61 U columns x R U rows of the SECOND subcycle:
//   stage 1  T1(c,r) on 64 lanes, U1(c,r-1) on lanes 0..62      (first subcycle, kept in registers)
//   stage 2  T2(c,r-1) on lanes 1..62, U2(c,r-2) on lanes 1..61 (second subcycle, stored)
// marching north with the second stage two rows behind the first.  Columns wrap (cyclic E-W) or fall
// outside the domain where every cell is inactive and u = 0 (open / closed E-W).
// ------------------------------------------------------------------------------------
constexpr int STRIP2_W = 61;

// number of strips of a pair launch: from the launch (the host knows the list) or from the count k_compact_strips left on the
// device (the launch then covers the upper bound and the blockIdx -> strip map takes its chunk from the count)
__device__ __forceinline__ int pair_nstrips(const SubArgs &a) {
    return a.nsdev ? __builtin_amdgcn_readfirstlane(*a.nsdev) : a.nstrips;
}

// (k_subcycle2 itself -- the pair kernel WITHOUT the LDS prefetch, superseded by k_subcycle2p in round 1 -- lives in evpk_experimental.hip,
//  built with -DEVPK_EXPERIMENTAL only)

// ------------------------------------------------------------------------------------
// k_subcycle2p: k_subcycle2 with the next row's planes prefetched through LDS.
// In k_subcycle2 a wave spends half of its life in s_waitcnt (225 VGPRs leave two waves per SIMD to hide HBM
// latency, and register prefetch has no room).  Here every step first issues direct global->LDS loads
// (global_load_lds_dwordx4: no VGPR destination, 1 KiB per wave instruction) for the row it will need in the NEXT
// step -- u/v at the two columns, five metric pairs, six sigma pairs of row r+1 and the stepu input pairs of
// row r -- then computes the current step from registers; the next step reads its operands from LDS.
// One 18 KiB ring slot per wave, 72 KiB per 256-thread workgroup, two workgroups per CU.
// The arithmetic and the results are those of k_subcycle2.
// ------------------------------------------------------------------------------------
// the eight metric planes from the primary grid lengths (ice_grid.F90:356-357, :362-367, :1455, :1533):
//   hn = HTN(i,j), hs = HTN(i,j-1), he = HTE(i,j), hw = HTE(i-1,j)
__device__ __forceinline__ TMet tmet_from_lengths(double hn, double hs, double he, double hw, double tiny, double strength) {
    TMet m;
    m.dxt = 0.5 * (hn + hs);
    m.dyt = 0.5 * (he + hw);
    m.dxhy = 0.5 * (he - hw);
    m.dyhx = 0.5 * (hn - hs);
    m.cyp = (1.5 * he - 0.5 * hw);
    m.cxp = (1.5 * hn - 0.5 * hs);
    m.cym = -(1.5 * hw - 0.5 * he);
    m.cxm = -(1.5 * hs - 0.5 * hn);
    m.tiny = tiny;
    m.strength = strength;
    return m;
}

// do HTN/HTE reproduce the stored metric planes bit for bit on every T cell that can become active?
__global__ void k_verify_metrics(Slab s, unsigned int *mismatch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;     // T cells 1..nxl+1, 1..nyl+1
    const int j = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (i > s.nxl + 1 || j > s.nyl + 1) return;
    if (!s.tmask[mcell(s, i, j)]) return;
    const size_t k = cell(s, i, j);
    const TMet m = tmet_from_lengths(FD(s, F_HTN, k), FD(s, F_HTN, cell(s, i, j - 1)), FD(s, F_HTE, k), FD(s, F_HTE, k - 1), 0.0, 0.0);
    const bool ok = m.dxt == FD(s, F_DXT, k) && m.dyt == FD(s, F_DYT, k) && m.dxhy == FD(s, F_DXHY, k) && m.dyhx == FD(s, F_DYHX, k) &&
                    m.cxp == FD(s, F_CXP, k) && m.cyp == FD(s, F_CYP, k) && m.cxm == FD(s, F_CXM, k) && m.cym == FD(s, F_CYM, k);
    if (!ok) atomicAdd(mismatch, 1u);
}

// ------------------------------------------------------------------------------------
// band_pair: the tripole TOP BAND of a pair of subcycles as workgroups OF THE MAIN LAUNCH (one rank, cyclic E-W).
// The fold (serial/ice_boundary.F90:801-888, :3752-3776) mixes mirrored columns between the two fused subcycles, but only U
// rows >= N-1 and T rows >= N-1 of the pair's result depend on it (N = nyl); the main strips leave those rows alone (jmax =
// N-2).  Until round 3 they were redone by two band launches and two fold launches on a second stream, with two stream
// hand-overs per pair (~17 us, and the band launches queued behind the main launch's resident workgroups).  Here ONE
// workgroup owns a strip A and its mirror image B, so that the fold between the subcycles is a matter of this workgroup's
// LDS -- no launch, no stream, no inter-workgroup synchronisation:
//   strip A of band workgroup k: lane l <-> column 61 k - 1 + l,  strip B: lane l <-> column nx - 61 k - 61 + l   (cyclic);
//   the mirror image of A's lane l is B's lane 62 - l (NE-corner fold: column g <-> nx - g, 0 == nx).
//   phase A  T1 rows N-2 .. N+1 (wave w: row N-2+w, strips A and B in turn), 64 lanes       -> LDS: str terms, sigma_1
//   phase B  U1 rows N-2 .. N, lanes 0..62                                                    -> LDS
//   fold 1   top row symmetrised, ghost row N+1 <- mirrored row N-1 (lanes 0..62)              in LDS
//   phase C  T2 rows N-1 .. N+1, lanes 1..62; sigma stored for lanes 1..61                    -> LDS: str terms
//   phase D  U2 rows N-1, N, lanes 1..61; row N-1 stored                                      -> LDS
//   fold 2   rows N and N+1 of the new state stored (+ the E-W ghost images)
// Lanes 1..61 of the A and B strips of all band workgroups cover every column once or twice (overlaps compute the same
// bits).  Arithmetic, operation order and the rule for inactive top-row cells are those of k_subcycle_t +
// k_halo_tripole_ne1: bit-identical (tests: every tripole case, test_two_subcycle_kernel_equals_single).
// Uses waves 0..3 of the workgroup and BAND_LDS_DOUBLES doubles of LDS; every wave takes part in the barriers.
// ------------------------------------------------------------------------------------
constexpr int BAND_LDS_DOUBLES = 2048 + 1024 + 4608 + 512;      // X, U1, S1, V2 = 64 KiB

// Two forms, one code:
//  * ONE RANK (a.xm == nullptr): strip A and strip B are both strips of the rank's own slab (cyclic wrap), A covering columns
//    0 .. nx/2 over the band workgroups and B their mirror images -- lane l of A <-> column 61 k - 1 + l, of B <-> nx - 61 k - 61 + l.
//  * X-SLAB RANKS (a.xm = the mirror slab M, round 3): strip A is a strip of MY slab (ghost-zone mode: columns 1-G .. w+G), strip
//    B its mirror image, read from and written to M -- rows N-3 .. N+1 of a VIRTUAL slab of my own width w that starts at global
//    column nx - i0 - w + 2 (with equal slab widths and an even number of ranks: the slab of the mirror rank P-1-r; otherwise
//    the columns of up to three ranks, round 4), ghost zones included.  The image of my local column i is M's local column
//    w - i, whatever the rank: lane l of A <-> column cA + l, of B <-> w - cA - 62 + l.  The owners of M's columns compute the
//    same strips (the same bits); rows N-1 .. N+1 of M are therefore kept current HERE, the rows below come with one message
//    per zM pairs (xband_swap) -- instead of four exchanges per pair around band launches on a second stream.
template <bool REVP, bool LAST2>
__device__ __forceinline__ void band_pair(const SubArgs &a, int k, double *lds) {
    const Slab &sA = a.s;
    const Slab &sB = a.xm ? *a.xm_slab() : a.s;
    const bool cross = a.xm != nullptr;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nxg = sA.nxg, h = nxg >> 1;
    const int wloc = sA.nxl;                                       // one rank: nx; x-slabs: the slab width w
    const double sgn = -1.0;                                       // (u, v): a vector
    const int SR = a.sr, SW = a.sw;
    double *const X = lds;                                         // [strip][row 0..3][4][64]
    double *const U1 = X + 2048;                                   // [strip][row 0..3][u, v][64]     rows N-2 .. N+1
    double *const S1 = U1 + 1024;                                  // [strip][row 1..3][12][64]
    double *const V2 = S1 + 4608;                                  // [strip][row 1..2][u, v][64]
    const bool rw = (w < 4);
    const int wr = rw ? w : 0;
    // per strip: slab, rows, addresses
    const Slab *ss[2] = {&sA, &sB};
    size_t pp[2], rowb[2];
    char *base[2], *rb[2];
    int r[2], Nn[2], ci[2], cl[2];
    unsigned lo[2];
    bool okc[2], okm[2], own[2];
    unsigned char m[2] = {0, 0};
    double uc[2] = {0, 0}, vc[2] = {0, 0}, k1[2] = {0, 0}, k2[2] = {0, 0}, k5[2] = {0, 0}, k7[2] = {0, 0};
    auto wrapc = [&](int c, int n) { int q = (c - 1) % n; if (q < 0) q += n; return q + 1; };
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const Slab &S = *ss[t];
        pp[t] = (size_t)S.pitch * 16; rowb[t] = (size_t)S.rstride * 16;
        base[t] = reinterpret_cast<char *>(S.F);
        Nn[t] = S.nyl;
        r[t] = Nn[t] - 2 + wr;
        rb[t] = base[t] + (size_t)r[t] * rowb[t];
        const int cA = cross ? 61 * k - a.G : 61 * k - 1;           // lane 0 of strip A (lane 1 = first stored column)
        const int c = (t ? wloc - cA - 62 : cA) + lane;
        cl[t] = c;
        if (cross) {
            okc[t] = (c >= 1 - ZW_MAX && c <= wloc + ZW_MAX);
            okm[t] = (c - 1 >= 1 - ZW_MAX && c - 1 <= wloc + ZW_MAX);
            ci[t] = okc[t] ? c : 1;
            own[t] = (lane >= 1 && lane <= 61 && c >= 1 - a.G && c <= wloc + a.G);
        } else {
            okc[t] = okm[t] = true;
            ci[t] = wrapc(c, wloc);
            own[t] = (lane >= 1 && lane <= 61);
        }
        lo[t] = (unsigned)(C0 + ci[t]) * 16u;
    }

    // ---------------- phase A: T1(r), both strips ----------------
    if (rw) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const Slab &S = *ss[t];
            const int cm = cross ? (okm[t] ? cl[t] - 1 : 1) : wrapc(ci[t] - 1, wloc);
            const unsigned lom = (unsigned)(C0 + cm) * 16u;
            double2 q0 = make_double2(0, 0), q1 = q0, q2 = q0, q3 = q0;
            const char *rs = rb[t] - rowb[t];
            if (okc[t]) {
                m[t] = S.cmask[(size_t)r[t] * S.pitch + C0 + ci[t]];
                q0 = ldp(rb[t], pp[t], SR + S_U, lo[t]); q2 = ldp(rs, pp[t], SR + S_U, lo[t]);
            }
            if (okm[t]) { q1 = ldp(rb[t], pp[t], SR + S_U, lom); q3 = ldp(rs, pp[t], SR + S_U, lom); }
            uc[t] = q0.x; vc[t] = q0.y;
            const bool tact = (m[t] & CM_T) != 0;
            Str8 o{0, 0, 0, 0, 0, 0, 0, 0};
            Sig g{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (__any(tact)) {
                if (tact) {
                    const TMet mt = load_tmet(rb[t], pp[t], lo[t]);
                    g = load_sig(rb[t], pp[t], SR, lo[t]);
                    Diag dg;
                    stress_cell<false>(mt, q0.x, q1.x, q2.x, q3.x, q0.y, q1.y, q2.y, q3.y, a.ecci, a.arlx1i, a.denom1, 0.0, g, o, dg);
                }
            }
            const double s2n = shfl_dn1(o.s2), s4n = shfl_dn1(o.s4), s7n = shfl_dn1(o.s7), s8n = shfl_dn1(o.s8);
            double *const Xq = X + (size_t)((t * 4 + w) * 4) * 64 + lane;
            Xq[0] = o.s3; Xq[64] = o.s6; Xq[128] = s4n; Xq[192] = s8n;
            k1[t] = o.s1; k2[t] = s2n; k5[t] = o.s5; k7[t] = s7n;
            if (w >= 1) {
                double *const Sq = S1 + (size_t)((t * 3 + (w - 1)) * 12) * 64 + lane;
                Sq[0] = g.sp1; Sq[64] = g.sp2; Sq[128] = g.sp3; Sq[192] = g.sp4; Sq[256] = g.sm1; Sq[320] = g.sm2;
                Sq[384] = g.sm3; Sq[448] = g.sm4; Sq[512] = g.s121; Sq[576] = g.s122; Sq[640] = g.s123; Sq[704] = g.s124;
            }
        }
    }
    __syncthreads();

    // ---------------- phase B: U1(r), rows N-2 .. N ----------------
    if (w <= 2) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            double u1 = uc[t], v1 = vc[t];                          // an inactive cell keeps its velocity
            const bool uact = (m[t] & CM_U) != 0 && r[t] >= 1 && r[t] <= Nn[t];
            if (__any(uact)) {
                if (uact) {
                    const UStat q = load_ustat(rb[t], pp[t], lo[t]);
                    double ui = 0.0, vi = 0.0, sxi, syi;
                    if (REVP) { const double2 iv = ldp(rb[t], pp[t], F_UVEL_INIT, lo[t]); ui = iv.x; vi = iv.y; }
                    const double *Xn = X + (size_t)((t * 4 + w + 1) * 4) * 64 + lane;      // the T row above
                    stepu_cell(q, uc[t], vc[t], ui, vi, ((k1[t] + k2[t]) + Xn[0]) + Xn[128], ((k5[t] + Xn[64]) + k7[t]) + Xn[192],
                               a.brlx, a.revp, a.cosw, a.sinw, u1, v1, sxi, syi);
                }
            }
            double *const Uq = U1 + (size_t)((t * 4 + w) * 2) * 64 + lane;
            Uq[0] = u1; Uq[64] = v1;
        }
    }
    __syncthreads();

    // global column of this lane in either strip (for the roles in the fold).  The folds pick their strip by the WAVE index: every
    // per-strip quantity they use is selected here from two scalars -- an array indexed by a run-time value would live in
    // scratch memory and drag the whole kernel (the main strips included) into spilling
    const int gcolA = wrapc(sA.i0 + cl[0] - 1, nxg), gcolB = wrapc(sB.i0 + cl[1] - 1, nxg);
    const bool w1 = (w == 1);
    const int gcolW = w1 ? gcolB : gcolA;
    const bool ownW = w1 ? own[1] : own[0];
    char *const baseW = w1 ? base[1] : base[0];
    const size_t rowbW = w1 ? rowb[1] : rowb[0], ppW = w1 ? pp[1] : pp[0];
    const int NnW = w1 ? Nn[1] : Nn[0], ciW = w1 ? ci[1] : ci[0];
    const unsigned loW = w1 ? lo[1] : lo[0];

    // ---------------- fold 1 (k_halo_tripole_ne1's arithmetic): waves 0, 1 <-> strips A, B; lanes 0 .. 62 ----------------
    {
        double tu = 0, tv = 0, gu = 0, gv = 0;
        const bool f1 = (w < 2) && lane <= 62;
        if (f1) {
            const int t = w, g = gcolW;
            const double *Ut = U1 + (size_t)((t * 4 + 2) * 2) * 64 + lane;                  // my top row
            const double *Um = U1 + (size_t)(((1 - t) * 4 + 2) * 2) * 64 + (62 - lane);     // the mirror column's top row
            const double *Rm = U1 + (size_t)(((1 - t) * 4 + 1) * 2) * 64 + (62 - lane);     // ... and its row N-1
            const double Tu = Ut[0], Tv = Ut[64], Mu = Um[0], Mv = Um[64];
            if (g == nxg || g == h) { tu = sgn * Tu; tv = sgn * Tv; }
            else if (g < h) { tu = sgn * (sgn * (0.5 * (Tu + sgn * Mu))); tv = sgn * (sgn * (0.5 * (Tv + sgn * Mv))); }
            else { tu = sgn * (0.5 * (Mu + sgn * Tu)); tv = sgn * (0.5 * (Mv + sgn * Tv)); }
            gu = sgn * Rm[0]; gv = sgn * Rm[64];
        }
        __syncthreads();
        if (f1) {
            const int t = w;
            double *const Ut = U1 + (size_t)((t * 4 + 2) * 2) * 64 + lane;
            double *const Ug = U1 + (size_t)((t * 4 + 3) * 2) * 64 + lane;
            Ut[0] = tu; Ut[64] = tv; Ug[0] = gu; Ug[64] = gv;
        }
    }
    __syncthreads();

    // ---------------- phase C: T2(r), rows N-1 .. N+1 (waves 1 .. 3), lanes 1 .. 62 ----------------
    if (w >= 1 && w <= 3) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const bool t2act = (m[t] & CM_T) != 0 && lane >= 1 && lane <= 62;
            Str8 o{0, 0, 0, 0, 0, 0, 0, 0};
            if (__any(t2act)) {
                if (t2act) {
                    const double *Uq = U1 + (size_t)((t * 4 + w) * 2) * 64 + lane, *Us = Uq - 128;
                    const double *Sq = S1 + (size_t)((t * 3 + (w - 1)) * 12) * 64 + lane;
                    Sig g{Sq[0], Sq[64], Sq[128], Sq[192], Sq[256], Sq[320], Sq[384], Sq[448], Sq[512], Sq[576], Sq[640], Sq[704]};
                    const TMet mt = load_tmet(rb[t], pp[t], lo[t]);
                    double tarear = 0.0;
                    if (LAST2) tarear = *reinterpret_cast<const double *>(rb[t] + (size_t)(F_TAREAR >> 1) * pp[t] + lo[t] + (F_TAREAR & 1) * 8);
                    Diag dg;
                    stress_cell<LAST2>(mt, Uq[0], Uq[-1], Us[0], Us[-1], Uq[64], Uq[63], Us[64], Us[63], a.ecci, a.arlx1i, a.denom1, tarear, g, o, dg);
                    if (own[t]) {
                        store_sig(rb[t], pp[t], SW, lo[t], g);
                        if (!cross && ci[t] == 1) store_sig(rb[t], pp[t], SW, lo[t] + (unsigned)wloc * 16u, g);     // east ghost T column = image of column 1
                        if (LAST2) {
                            st1(rb[t], pp[t], F_DIVU, lo[t], dg.divu);       st1(rb[t], pp[t], F_RDGCONV, lo[t], dg.rdg_conv);
                            st1(rb[t], pp[t], F_RDGSHEAR, lo[t], dg.rdg_shear); st1(rb[t], pp[t], F_SHEAR, lo[t], dg.shear);
                            st1(rb[t], pp[t], F_PRSSIG, lo[t], dg.prs);
                        }
                    }
                }
            }
            const double s2n = shfl_dn1(o.s2), s4n = shfl_dn1(o.s4), s7n = shfl_dn1(o.s7), s8n = shfl_dn1(o.s8);
            double *const Xq = X + (size_t)((t * 4 + w) * 4) * 64 + lane;
            Xq[0] = o.s3; Xq[64] = o.s6; Xq[128] = s4n; Xq[192] = s8n;
            k1[t] = o.s1; k2[t] = s2n; k5[t] = o.s5; k7[t] = s7n;
        }
    }
    __syncthreads();

    // ---------------- phase D: U2(r), rows N-1, N (waves 1, 2), lanes 1 .. 61 ----------------
    if (w >= 1 && w <= 2) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const double *Uq = U1 + (size_t)((t * 4 + w) * 2) * 64 + lane;
            const double u1 = Uq[0], v1 = Uq[64];
            double u2 = u1, v2 = v1;
            const bool uact = (m[t] & CM_U) != 0 && lane >= 1 && lane <= 61;       // (r <= N)
            if (__any(uact)) {
                if (uact) {
                    const UStat q = load_ustat(rb[t], pp[t], lo[t]);
                    double ui = 0.0, vi = 0.0, sxi, syi;
                    if (REVP) { const double2 iv = ldp(rb[t], pp[t], F_UVEL_INIT, lo[t]); ui = iv.x; vi = iv.y; }
                    const double *Xn = X + (size_t)((t * 4 + w + 1) * 4) * 64 + lane;
                    stepu_cell(q, u1, v1, ui, vi, ((k1[t] + k2[t]) + Xn[0]) + Xn[128], ((k5[t] + Xn[64]) + k7[t]) + Xn[192],
                               a.brlx, a.revp, a.cosw, a.sinw, u2, v2, sxi, syi);
                    if (w == 1 && own[t]) {                         // row N-1 is final; row N goes through the fold below
                        stp(rb[t], pp[t], SW + S_U, lo[t], u2, v2);
                        if (!cross) {
                            if (ci[t] == 1) stp(rb[t], pp[t], SW + S_U, lo[t] + (unsigned)wloc * 16u, u2, v2);
                            if (ci[t] == wloc) stp(rb[t], pp[t], SW + S_U, lo[t] - (unsigned)wloc * 16u, u2, v2);
                        }
                    }
                    if (LAST2 && own[t]) { st1(rb[t], pp[t], F_STRINTX, lo[t], sxi); st1(rb[t], pp[t], F_STRINTY, lo[t], syi); }
                }
            }
            double *const Vq = V2 + (size_t)((t * 2 + (w - 1)) * 2) * 64 + lane;
            Vq[0] = u2; Vq[64] = v2;
        }
    }
    __syncthreads();

    // ---------------- fold 2: rows N and N+1 of the new state, lanes 1 .. 61 ----------------
    if (w < 2 && lane >= 1 && lane <= 61) {
        const int t = w, g = gcolW;
        const double *Vt = V2 + (size_t)((t * 2 + 1) * 2) * 64 + lane;
        const double *Vm = V2 + (size_t)(((1 - t) * 2 + 1) * 2) * 64 + (62 - lane);
        const double *Rm = V2 + (size_t)(((1 - t) * 2 + 0) * 2) * 64 + (62 - lane);
        const double Tu = Vt[0], Tv = Vt[64], Mu = Vm[0], Mv = Vm[64];
        double tu, tv;
        if (g == nxg || g == h) { tu = sgn * Tu; tv = sgn * Tv; }
        else if (g < h) { tu = sgn * (sgn * (0.5 * (Tu + sgn * Mu))); tv = sgn * (sgn * (0.5 * (Tv + sgn * Mv))); }
        else { tu = sgn * (0.5 * (Mu + sgn * Tu)); tv = sgn * (0.5 * (Mv + sgn * Tv)); }
        const double gu = sgn * Rm[0], gv = sgn * Rm[64];
        if (ownW) {
            char *const rN = baseW + (size_t)NnW * rowbW, *const rG = rN + rowbW;
            stp(rN, ppW, SW + S_U, loW, tu, tv);
            stp(rG, ppW, SW + S_U, loW, gu, gv);
            if (!cross) {
                if (ciW == 1) { stp(rN, ppW, SW + S_U, loW + (unsigned)wloc * 16u, tu, tv); stp(rG, ppW, SW + S_U, loW + (unsigned)wloc * 16u, gu, gv); }
                if (ciW == wloc) { stp(rN, ppW, SW + S_U, loW - (unsigned)wloc * 16u, tu, tv); stp(rG, ppW, SW + S_U, loW - (unsigned)wloc * 16u, gu, gv); }
            }
        }
    }
}

constexpr int PF_SLOTS = 18;     // 0,1: (u,v) at c, c-1; 2..6: metrics; 7..12: sigma; 13..16: stepu inputs; 17: uvel_init

__device__ __forceinline__ void lds_dma16(const char *gsrc, double2 *lds_slot) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_slot, 16, 0, 0);
}

// The same with the address as the hardware takes it -- a scalar base (SGPR pair) + a 32-bit lane offset -- and the LDS byte address of
// the slot in M0: one instruction where the builtin form costs one to three vector instructions per transfer (a 64-bit vector add
// for the address; a vector add and a v_readfirstlane for M0 when the wave index is not known to be uniform).  Used by the tile
// kernels.  In k_subcycle2p it took 54 of the 1 393 vector instructions out of a march step (with the zero fills of lane-private
// state: 1 393 -> 1 314) and changed its time by NOTHING (profiles/r05_v1/diet_ab.txt: 0.2147 against 0.2150 ms per launch,
// alternating on one box) -- that kernel is not bound by instruction issue; it keeps the builtin.  M0 is written behind the
// compiler's back: use this only on paths where every LDS-DMA goes through it.
// (the slot's offset is an immediate of the s_add that makes M0: one SGPR -- the wave's LDS base -- serves all slots; with a value per
// slot the compiler hoisted eighteen of them out of the march, spilled them and read them back with v_readlane)
template <int SLOT>
__device__ __forceinline__ void lds_dma16s(const char *sbase, unsigned voff, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_base), "n"(SLOT * 1024) : "memory", "scc");
}

// CM (compact metrics): slots 2,3 hold (HTN,HTE) at columns c and c-1 instead of the four metric pairs in slots 2..5
// what the pair kernels use of a slab, in scalar registers: the rank's own (kernel arguments) or, for the workgroups that advance
// the mirror slab (XM), the struct in device memory -- read lane-uniformly so that it stays in SGPRs
struct SlabV { double *F; const unsigned char *cmask; int pitch, rstride, nxl, nyl; };
__device__ __forceinline__ unsigned long long uni64(unsigned long long x) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}
__device__ __forceinline__ SlabV slab_view(const SubArgs &a, bool mir) {
    SlabV v{a.s.F, a.s.cmask, a.s.pitch, a.s.rstride, a.s.nxl, a.s.nyl};
    if (mir) {
        const Slab *m = a.xm;
        v.F = reinterpret_cast<double *>(uni64(reinterpret_cast<unsigned long long>(m->F)));
        v.cmask = reinterpret_cast<const unsigned char *>(uni64(reinterpret_cast<unsigned long long>(m->cmask)));
        v.pitch = __builtin_amdgcn_readfirstlane(m->pitch); v.rstride = __builtin_amdgcn_readfirstlane(m->rstride);
        v.nxl = __builtin_amdgcn_readfirstlane(m->nxl); v.nyl = __builtin_amdgcn_readfirstlane(m->nyl);
    }
    return v;
}

#ifndef EVPK_K2P_ILP
#define EVPK_K2P_ILP 0      // measured (round 4, scripts/lib_ab.sh): 255 VGPRs + 2 spilled, 0 ... -1.5 % -- the partner wave already fills the bubbles
#endif
constexpr bool K2P_ILP = EVPK_K2P_ILP != 0;
template <bool REVP, bool LAST2, bool CM, bool XM = false>
__global__ __launch_bounds__(256, 2) void k_subcycle2p(SubArgs a) {      // two workgroups per CU: at most 256 VGPRs
    __shared__ double2 smem[4 * PF_SLOTS * 64];
    static_assert(sizeof(double2) * 4 * PF_SLOTS * 64 >= sizeof(double) * BAND_LDS_DOUBLES, "the band workgroups use the same LDS");
    const int lane = threadIdx.x & 63;
    double2 *const L = smem + (size_t)(threadIdx.x >> 6) * PF_SLOTS * 64;     // this wave's slots
    const int nband8 = (a.nband + 7) & ~7;
    const int nmir8 = XM ? ((((a.nmir + 3) >> 2) + 7) & ~7) : 0;
    // logical workgroup index: band, mirror slab, strips -- in that order from workgroup 0, or (band_last) rotated so that the strips
    // are dispatched first (nband8, nmir8 are multiples of 8: the XCD of a strip's workgroup does not change)
    const int vb = a.band_last ? (int)((blockIdx.x + (unsigned)(nband8 + nmir8)) % gridDim.x) : (int)blockIdx.x;
    if (vb < nband8) {              // tripole top band of this pair of subcycles (one rank): see band_pair
        if (vb < a.nband) band_pair<REVP, LAST2>(a, vb, reinterpret_cast<double *>(smem));
        return;
    }
    // XM (x-slab ranks, tripole): the next workgroups advance the strips of the mirror slab (SubArgs::nmir)
    const bool mir = XM && vb - nband8 < nmir8;
    const SlabV s = slab_view(a, mir);
    const int jmax = mir ? a.mjmax : a.jmax;
    int sid, st;
    if (mir) {
        sid = __builtin_amdgcn_readfirstlane((vb - nband8) * 4 + (int)(threadIdx.x >> 6));
        if (sid >= a.nmir) return;
        st = sid;
    } else {
        const int bidx = vb - nband8 - nmir8;
        const int ns = pair_nstrips(a);
        const int chunk = a.nsdev ? (((ns + 3) >> 2) + 7) >> 3 : ((int)gridDim.x - nband8 - nmir8) >> 3;
        if ((bidx >> 3) >= chunk) return;
        const int wg = (bidx & 7) * chunk + (bidx >> 3);
        sid = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
        if (sid >= ns) return;
        st = __builtin_amdgcn_readfirstlane(a.strips[sid]);
        dbg_stamp(a, sid, 0);
    }
    const int cx = st % a.ncx, ry = st / a.ncx;
    const int R = a.R, nxl = s.nxl, nyl = s.nyl;
    const int G = a.G;
    const int c = cx * STRIP2_W + lane - G;
    const int jb = ry * R + 1;
    const bool cyc = a.wrap != 0;

    int ci = c, cm1 = c - 1;
    bool okc, okm;
    if (cyc) {
        ci = (c - 1) % nxl; if (ci < 0) ci += nxl; ci += 1;
        cm1 = (c - 2) % nxl; if (cm1 < 0) cm1 += nxl; cm1 += 1;
        okc = okm = true;
    } else {
        okc = (c >= -1 - G && c <= nxl + 2 + G);
        okm = (cm1 >= -1 - G && cm1 <= nxl + 2 + G);
        if (!okc) ci = 0;
        if (!okm) cm1 = 0;
    }
    const bool tcol = cyc ? true : (c >= -G && c <= nxl + 2 + G);
    const bool ucol = cyc ? true : (c >= -G && c <= nxl + 1 + G);
    const bool own = (lane >= 1 && lane <= STRIP2_W && c >= 1 - G && c <= nxl + G);

    const size_t pp = (size_t)s.pitch * 16;
    const size_t rowb = (size_t)s.rstride * 16;
    const unsigned lo = (unsigned)(C0 + ci) * 16u, lom = (unsigned)(C0 + cm1) * 16u;
    const int SR = a.sr;
    const int SW = a.sw;
    char *const base = reinterpret_cast<char *>(s.F);

    auto rowok = [&](int r) { return r >= 0 && r <= nyl + 1; };
    auto mask_of = [&](int r) -> unsigned char { return (rowok(r) && okc) ? s.cmask[(size_t)r * s.pitch + C0 + ci] : (unsigned char)0; };
    // prefetch for the step whose T1 row is rn: T planes of row rn, stepu inputs of row rn-1 (mask mu)
    auto issue = [&](int rn, unsigned char mt_, unsigned char mu_, unsigned char mt_next) {
        if (rowok(rn)) {
            const char *rbn = base + (size_t)rn * rowb;
            if (okc) lds_dma16(rbn + (size_t)((SR + S_U) >> 1) * pp + lo, L + 0 * 64);
            if (okm) lds_dma16(rbn + (size_t)((SR + S_U) >> 1) * pp + lom, L + 1 * 64);
            const bool ta = tcol && (mt_ & CM_T) != 0;
            if (CM) {   // HTN of this row is also the south length of the next row: fetch it if either is active
                const bool th = tcol && ((mt_ | mt_next) & CM_T) != 0;
                if (__any(th)) { if (th) lds_dma16(rbn + (size_t)(F_HTN >> 1) * pp + lo, L + 2 * 64); }
            }
            if (__any(ta)) {
                if (ta) {
                    if (CM) {
                        if (okm) lds_dma16(rbn + (size_t)(F_HTN >> 1) * pp + lom, L + 3 * 64);
                    } else {
                        lds_dma16(rbn + (size_t)(F_CXP >> 1) * pp + lo, L + 2 * 64);
                        lds_dma16(rbn + (size_t)(F_CXM >> 1) * pp + lo, L + 3 * 64);
                        lds_dma16(rbn + (size_t)(F_DXT >> 1) * pp + lo, L + 4 * 64);
                        lds_dma16(rbn + (size_t)(F_DXHY >> 1) * pp + lo, L + 5 * 64);
                    }
                    lds_dma16(rbn + (size_t)(F_TINYAREA >> 1) * pp + lo, L + 6 * 64);
#pragma unroll
                    for (int q = 0; q < 6; q++) lds_dma16(rbn + (size_t)((SR + S_SP) / 2 + q) * pp + lo, L + (7 + q) * 64);
                }
            }
        }
        const int ru_ = rn - 1;
        const bool ua = ucol && (mu_ & CM_U) != 0 && ru_ >= 1 && ru_ <= nyl;
        if (__any(ua)) {
            if (ua) {
                const char *rbu = base + (size_t)ru_ * rowb;
                lds_dma16(rbu + (size_t)(F_VRELC >> 1) * pp + lo, L + 13 * 64);
                lds_dma16(rbu + (size_t)(F_UOCN >> 1) * pp + lo, L + 14 * 64);
                lds_dma16(rbu + (size_t)(F_FORCEX >> 1) * pp + lo, L + 15 * 64);
                lds_dma16(rbu + (size_t)(F_UMASSDTI >> 1) * pp + lo, L + 16 * 64);
                if (REVP) lds_dma16(rbu + (size_t)(F_UVEL_INIT >> 1) * pp + lo, L + 17 * 64);
            }
        }
    };

    // ---- carried state (as in k_subcycle2) ----
    double uo_c = 0.0, vo_c = 0.0, uo_m = 0.0, vo_m = 0.0;
    double a1c = 0.0, a5c = 0.0, a2r = 0.0, a7r = 0.0;
    Sig g1p{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    TMet mtp{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                       // !CM: metrics of row r-1
    double hn_p = 0.0, hn_pp = 0.0, he_p = 0.0, hw_p = 0.0, tiny_p = 0.0, str_p = 0.0;   // CM: HTN(c,r-1), HTN(c,r-2), HTE(c,r-1), HTE(c-1,r-1), ...
    unsigned char mp = 0, mpp = 0;
    double u1p_c = 0.0, v1p_c = 0.0, u1p_m = 0.0, v1p_m = 0.0;
    double b1c = 0.0, b5c = 0.0, b2r = 0.0, b7r = 0.0;
    UStat qp{0, 0, 0, 0, 0, 0, 0, 0};
    double uip = 0.0, vip = 0.0;

    {
        const int r0 = jb - 2;
        if (r0 >= 0) {
            const char *rb0 = base + (size_t)r0 * rowb;
            if (okc) { const double2 t = ldp(rb0, pp, SR + S_U, lo); uo_c = t.x; vo_c = t.y; }
            if (okm) { const double2 t = ldp(rb0, pp, SR + S_U, lom); uo_m = t.x; vo_m = t.y; }
        }
    }
    if (CM) {   // south length of the first T1 row
        const int r0 = jb - 2;
        if (r0 >= 0 && okc) hn_p = *reinterpret_cast<const double *>(base + (size_t)r0 * rowb + (size_t)(F_HTN >> 1) * pp + lo);
    }
    // masks of the first three T1 rows; prefetch of the first one (its U row jb-2 is never advanced)
    unsigned char m = mask_of(jb - 1), m_n1 = mask_of(jb), m_n2 = mask_of(jb + 1);
    issue(jb - 1, m, 0, m_n1);

    // (prio 2: the wave in the odd hardware slot -- with two waves per SIMD the YOUNGER one, which loses every tie -- steps down
    // half a quarter later than its partner: at equal progress it wins half of the time instead of never)
    const int pq = a.prio ? max(1, (R + 3) / 4) : 0;
    int psh = 0;
    if (a.prio >= 2) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        psh = (hwid & 1u) ? (pq + 1) / 2 : 0;
    }
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    for (int t = 0; t <= R + 2; t++) {
        const int r = jb - 1 + t;
        if (r > nyl + 2) break;
        const bool rok = rowok(r);
        if (a.prio) {
            if (t == pq + psh) __builtin_amdgcn_s_setprio(2);
            else if (t == 2 * pq + psh) __builtin_amdgcn_s_setprio(1);
            else if (t == 3 * pq + psh) __builtin_amdgcn_s_setprio(0);
        }

        // ---------------- operands of this step: LDS -> registers ----------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        double un_c = 0.0, vn_c = 0.0, un_m = 0.0, vn_m = 0.0;
        if (rok) {
            if (okc) { const double2 q = L[0 * 64 + lane]; un_c = q.x; vn_c = q.y; }
            if (okm) { const double2 q = L[1 * 64 + lane]; un_m = q.x; vn_m = q.y; }
        }
        const bool t1act = tcol && (m & CM_T) != 0;
        Sig g1{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        TMet mt{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        double hn = 0.0, he = 0.0, hw = 0.0, tiny_c = 0.0, str_c = 0.0;
        if (CM) {
            const bool th = tcol && ((m | m_n1) & CM_T) != 0 && rok;
            if (__any(th)) { if (th) { const double2 h = L[2 * 64 + lane]; hn = h.x; he = h.y; } }
        }
        if (__any(t1act)) {
            if (t1act) {
                if (CM) {
                    hw = L[3 * 64 + lane].y;
                    const double2 ts = L[6 * 64 + lane];
                    tiny_c = ts.x; str_c = ts.y;
                    mt = tmet_from_lengths(hn, hn_p, he, hw, tiny_c, str_c);
                } else {
                    const double2 cp = L[2 * 64 + lane], cm = L[3 * 64 + lane], dd = L[4 * 64 + lane], dh = L[5 * 64 + lane], ts = L[6 * 64 + lane];
                    mt = TMet{cp.x, cp.y, cm.x, cm.y, dd.x, dd.y, dh.x, dh.y, ts.x, ts.y};
                }
                const double2 q0 = L[7 * 64 + lane], q1 = L[8 * 64 + lane], q2 = L[9 * 64 + lane];
                const double2 q3 = L[10 * 64 + lane], q4 = L[11 * 64 + lane], q5 = L[12 * 64 + lane];
                g1 = Sig{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y, q4.x, q4.y, q5.x, q5.y};
            }
        }
        const bool u1act = (t >= 1) && ucol && (mp & CM_U) != 0 && (r - 1 >= 1) && (r - 1 <= nyl);
        UStat q1{0, 0, 0, 0, 0, 0, 0, 0};
        double ui1 = 0.0, vi1 = 0.0;
        if (__any(u1act)) {
            if (u1act) {
                const double2 va = L[13 * 64 + lane], oc = L[14 * 64 + lane], fo = L[15 * 64 + lane], mf = L[16 * 64 + lane];
                q1 = UStat{va.x, va.y, oc.x, oc.y, fo.x, fo.y, mf.x, mf.y};
                if (REVP) { const double2 iv = L[17 * 64 + lane]; ui1 = iv.x; vi1 = iv.y; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // all LDS reads back before the slots are refilled
        __builtin_amdgcn_wave_barrier();

        // ---------------- prefetch for the next step, mask three rows ahead ----------------
        const unsigned char m_n3 = mask_of(r + 3);
        issue(r + 1, m_n1, m, m_n2);

        // ---------------- stage 1: T1(r) ----------------
        Str8 o1{0, 0, 0, 0, 0, 0, 0, 0};
        if (__any(t1act)) {
            if (t1act) {
                Diag dg;
                stress_cell<false, K2P_ILP>(mt, un_c, un_m, uo_c, uo_m, vn_c, vn_m, vo_c, vo_m, a.ecci, a.arlx1i, a.denom1, 0.0, g1, o1, dg);
            }
        }
        const double a2n = shfl_dn1(o1.s2), a4n = shfl_dn1(o1.s4), a7n = shfl_dn1(o1.s7), a8n = shfl_dn1(o1.s8);

        // ---------------- stage 1: U1(r-1) ----------------
        double u1_c = uo_c, v1_c = vo_c;
        if (__any(u1act)) {
            if (u1act) {
                double sxi, syi;
                stepu_cell<K2P_ILP>(q1, uo_c, vo_c, ui1, vi1, ((a1c + a2r) + o1.s3) + a4n, ((a5c + o1.s6) + a7r) + a8n,
                           a.brlx, a.revp, a.cosw, a.sinw, u1_c, v1_c, sxi, syi);
            }
        }
        const double u1_m = shfl_up1(u1_c), v1_m = shfl_up1(v1_c);

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
                const TMet mt2 = CM ? tmet_from_lengths(hn_p, hn_pp, he_p, hw_p, tiny_p, str_p) : mtp;
                stress_cell<LAST2, K2P_ILP>(mt2, u1_c, u1_m, u1p_c, u1p_m, v1_c, v1_m, v1p_c, v1p_m, a.ecci, a.arlx1i, a.denom1, tarear, g2, o2, dg);
                if (own && q2 >= jb && q2 < jb + R && q2 <= jmax) {
                    store_sig(rq, pp, SW, lo, g2);
                    if (cyc && c == 1) store_sig(rq, pp, SW, lo + (unsigned)nxl * 16u, g2);
                    if (LAST2) {
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
        const bool u2act = (t >= 3) && own && (mpp & CM_U) != 0 && q3 >= jb && q3 < jb + R && q3 <= nyl && q3 <= jmax;
        if (__any(u2act)) {
            if (u2act) {
                double un, vn, sxi, syi;
                stepu_cell<K2P_ILP>(qp, u1p_c, v1p_c, uip, vip, ((b1c + b2r) + o2.s3) + b4n, ((b5c + o2.s6) + b7r) + b8n,
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
        g1p = g1;
        if (CM) { hn_pp = hn_p; hn_p = hn; he_p = he; hw_p = hw; tiny_p = tiny_c; str_p = str_c; } else { mtp = mt; }
        uo_c = un_c; vo_c = vn_c; uo_m = un_m; vo_m = vn_m;
        mpp = mp; mp = m; m = m_n1; m_n1 = m_n2; m_n2 = m_n3;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may be in flight when the wave ends
    if (!mir) dbg_stamp(a, sid, 1);
}

template __global__ void k_subcycle2p<false, false, false>(SubArgs);
template __global__ void k_subcycle2p<true, false, false>(SubArgs);
template __global__ void k_subcycle2p<false, true, false>(SubArgs);
template __global__ void k_subcycle2p<true, true, false>(SubArgs);
template __global__ void k_subcycle2p<false, false, true>(SubArgs);
template __global__ void k_subcycle2p<true, false, true>(SubArgs);
template __global__ void k_subcycle2p<false, true, true>(SubArgs);
template __global__ void k_subcycle2p<true, true, true>(SubArgs);
template __global__ void k_subcycle2p<false, false, false, true>(SubArgs);      // (the mirror slab is only advanced when another pair follows)
template __global__ void k_subcycle2p<true, false, false, true>(SubArgs);
template __global__ void k_subcycle2p<false, false, true, true>(SubArgs);
template __global__ void k_subcycle2p<true, false, true, true>(SubArgs);

// ------------------------------------------------------------------------------------
// k_subcycle2t: the two fused subcycles of k_subcycle2 WITHOUT the north march -- the small-slab variant.
// k_subcycle2 gives one wave a strip of 61 columns x R rows and lets it march north, R + 3 dependent steps of ~3 us each:
// fine when there are more strips than wave slots, but a 320 x 384 grid (or one eighth of the 3600 x 2700 grid per GPU)
// has fewer, and the launch then lasts as long as ONE wave's serial march whatever the chip could do in parallel.
// Here a workgroup of R + 3 waves takes the same strip and every wave takes ONE row of it, r = jb - 1 + w:
//   phase A  T1(r)      all waves            -> LDS: the four terms the U row below needs (str3, str6, str4/str8 of the east cell)
//   phase B  U1(r)      waves 0 .. R+1       -> LDS: (u, v) after the first subcycle at columns c and c-1
//   phase C  T2(r)      waves 1 .. R+1       -> LDS: the same four terms of the second subcycle; sigma stored for waves 1 .. R
//   phase D  U2(r)      waves 1 .. R         -> (u, v) stored
// three workgroup barriers instead of R + 3 march steps; E-W neighbours still travel by DPP wave shifts.  Same strips,
// same column / ghost-zone / tripole-band (jmax) rules, same arithmetic in the same order: bit-identical to k_subcycle2.
// LDS: two arrays of [waves][4][64] doubles (the first one serves phases A and C).
// ------------------------------------------------------------------------------------
template <bool REVP, bool LAST2, bool XM>
__device__ __forceinline__ void subcycle2t_body(const SubArgs &a) {
    extern __shared__ double tl[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = blockDim.x >> 6;                   // R + 3
    const int nband8 = (a.nband + 7) & ~7;
    const int nmir8 = XM ? ((a.nmir + 7) & ~7) : 0;
    const int vb = a.band_last ? (int)((blockIdx.x + (unsigned)(nband8 + nmir8)) % gridDim.x) : (int)blockIdx.x;      // (see k_subcycle2p)
    if (vb < nband8) {                   // tripole top band of this pair (one rank; the launch asks for the LDS)
        if (vb < a.nband) band_pair<REVP, LAST2>(a, vb, tl);
        return;
    }
    // XM (x-slab ranks, tripole): the next workgroups advance the strips of the mirror slab (SubArgs::nmir), one each
    const bool mir = XM && vb - nband8 < nmir8;
    const SlabV s = slab_view(a, mir);
    const int jmax = mir ? a.mjmax : a.jmax;
    int wg, st;
    if (mir) {
        wg = vb - nband8;
        if (wg >= a.nmir) return;                     // (the whole workgroup leaves: no barrier is left waiting)
        st = wg;
    } else {
        const int bidx = vb - nband8 - nmir8;
        const int ns = pair_nstrips(a);
        const int chunk = a.nsdev ? (ns + 7) >> 3 : ((int)gridDim.x - nband8 - nmir8) >> 3;
        wg = (bidx & 7) * chunk + (bidx >> 3);
        if ((bidx >> 3) >= chunk || wg >= ns) return; // (likewise)
        st = __builtin_amdgcn_readfirstlane(a.strips[wg]);
        if (w == 0) dbg_stamp(a, wg, 0);
    }
    const int cx = st % a.ncx, ry = st / a.ncx;
    const int R = a.R, nxl = s.nxl, nyl = s.nyl;
    const int G = a.G;
    const int c = cx * STRIP2_W + lane - G;
    const int jb = ry * R + 1;
    const bool cyc = a.wrap != 0;
    int ci = c, cm1 = c - 1;
    bool okc, okm;
    if (cyc) {
        ci = (c - 1) % nxl; if (ci < 0) ci += nxl; ci += 1;
        cm1 = (c - 2) % nxl; if (cm1 < 0) cm1 += nxl; cm1 += 1;
        okc = okm = true;
    } else {
        okc = (c >= -1 - G && c <= nxl + 2 + G);
        okm = (cm1 >= -1 - G && cm1 <= nxl + 2 + G);
        if (!okc) ci = 0;
        if (!okm) cm1 = 0;
    }
    const bool tcol = cyc ? true : (c >= -G && c <= nxl + 2 + G);
    const bool ucol = cyc ? true : (c >= -G && c <= nxl + 1 + G);
    const bool own = (lane >= 1 && lane <= STRIP2_W && c >= 1 - G && c <= nxl + G);

    const size_t pp = (size_t)s.pitch * 16;
    const unsigned pp32 = (unsigned)s.pitch * 16u;
    const size_t rowb = (size_t)s.rstride * 16;
    const unsigned lo = (unsigned)(C0 + ci) * 16u, lom = (unsigned)(C0 + cm1) * 16u;
    const int SR = a.sr, SW = a.sw;
    char *const base = reinterpret_cast<char *>(s.F);
    double *const X = tl, *const Y = tl + (size_t)NW * 256;
    double *const Xw = X + (size_t)w * 256 + lane, *const Yw = Y + (size_t)w * 256 + lane;
    // the stepu inputs of this wave's row (four pairs + uvel_init): global -> LDS directly, no registers while the two stresses
    // are computed (round 4: the kernel is compiled for 128 VGPRs and spilled 8 with them live from the start)
    double2 *const Qw = reinterpret_cast<double2 *>(tl + (size_t)NW * 512) + (size_t)w * 5 * 64;
    const unsigned Qb = (unsigned)(size_t)(__attribute__((address_space(3))) void *)Qw;      // (LDS byte address, uniform: w is)

    const int r = jb - 1 + w;                         // this wave's row
    const bool rowok = (r >= 0 && r <= nyl + 1);
    char *const rb = base + (size_t)(rowok ? r : 0) * rowb;

    // ---------------- phase A: T1(r) ----------------
    unsigned char m = 0;
    double un_c = 0.0, vn_c = 0.0, un_m = 0.0, vn_m = 0.0;        // u_old at (c, r), (c-1, r)
    double uo_c = 0.0, vo_c = 0.0, uo_m = 0.0, vo_m = 0.0;        // ... at (c, r-1), (c-1, r-1)
    if (rowok) {
        if (okc) {
            m = s.cmask[(size_t)r * s.pitch + C0 + ci];
            const double2 q = ldp(rb, pp, SR + S_U, lo); un_c = q.x; vn_c = q.y;
        }
        if (okm) { const double2 q = ldp(rb, pp, SR + S_U, lom); un_m = q.x; vn_m = q.y; }
    }
    if (r - 1 >= 0 && r - 1 <= nyl + 1) {
        const char *rs = base + (size_t)(r - 1) * rowb;
        if (okc) { const double2 q = ldp(rs, pp, SR + S_U, lo); uo_c = q.x; vo_c = q.y; }
        if (okm) { const double2 q = ldp(rs, pp, SR + S_U, lom); uo_m = q.x; vo_m = q.y; }
    }
    const bool t1act = tcol && (m & CM_T) != 0;
    const bool u1act = (w <= NW - 2) && ucol && (m & CM_U) != 0 && r >= 1 && r <= nyl;
    Str8 o1{0, 0, 0, 0, 0, 0, 0, 0};
    Sig g1{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    TMet mt{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (__any(t1act)) {
        if (t1act) { mt = load_tmet(rb, pp, lo); g1 = load_sig(rb, pp, SR, lo); }
    }
    if (__any(u1act)) {      // the stepu inputs of this row serve both subcycles; fetched beside the T planes, into the LDS
        if (u1act) {
            lds_dma16s<0>(rb + (size_t)((unsigned)(F_VRELC >> 1) * pp32), lo, Qb);
            lds_dma16s<1>(rb + (size_t)((unsigned)(F_UOCN >> 1) * pp32), lo, Qb);
            lds_dma16s<2>(rb + (size_t)((unsigned)(F_FORCEX >> 1) * pp32), lo, Qb);
            lds_dma16s<3>(rb + (size_t)((unsigned)(F_UMASSDTI >> 1) * pp32), lo, Qb);
            if (REVP) lds_dma16s<4>(rb + (size_t)((unsigned)(F_UVEL_INIT >> 1) * pp32), lo, Qb);
        }
    }
    auto load_q = [&](UStat &q, double &ui, double &vi) {
        const double2 va = Qw[0 * 64 + lane], oc = Qw[1 * 64 + lane], fo = Qw[2 * 64 + lane], mf = Qw[3 * 64 + lane];
        q = UStat{va.x, va.y, oc.x, oc.y, fo.x, fo.y, mf.x, mf.y};
        ui = 0.0; vi = 0.0;
        if (REVP) { const double2 iv = Qw[4 * 64 + lane]; ui = iv.x; vi = iv.y; }
    };
    if (__any(t1act)) {
        if (t1act) {
            Diag dg;
            stress_cell<false>(mt, un_c, un_m, uo_c, uo_m, vn_c, vn_m, vo_c, vo_m, a.ecci, a.arlx1i, a.denom1, 0.0, g1, o1, dg);
        }
    }
    const double a2n = shfl_dn1(o1.s2), a4n = shfl_dn1(o1.s4), a7n = shfl_dn1(o1.s7), a8n = shfl_dn1(o1.s8);
    Xw[0] = o1.s3; Xw[64] = o1.s6; Xw[128] = a4n; Xw[192] = a8n;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (this wave's LDS-DMA has landed)
    __syncthreads();

    // ---------------- phase B: U1(r) ----------------
    double u1_c = un_c, v1_c = vn_c;                  // an inactive cell keeps its velocity
    if (__any(u1act)) {
        if (u1act) {
            const double *Xn = Xw + 256;              // the row above
            double sxi, syi;
            UStat q1; double ui1, vi1;
            load_q(q1, ui1, vi1);
            stepu_cell(q1, un_c, vn_c, ui1, vi1, ((o1.s1 + a2n) + Xn[0]) + Xn[128], ((o1.s5 + Xn[64]) + a7n) + Xn[192],
                       a.brlx, a.revp, a.cosw, a.sinw, u1_c, v1_c, sxi, syi);
        }
    }
    const double u1_m = shfl_up1(u1_c), v1_m = shfl_up1(v1_c);      // (c-1, r); lane 0 is not used below
    Yw[0] = u1_c; Yw[64] = v1_c; Yw[128] = u1_m; Yw[192] = v1_m;
    __syncthreads();

    // ---------------- phase C: T2(r) ----------------
    const bool t2act = (w >= 1) && (w <= NW - 2) && t1act && lane >= 1;
    Str8 o2{0, 0, 0, 0, 0, 0, 0, 0};
    if (__any(t2act)) {
        if (t2act) {
            const double *Ys = Yw - 256;              // the row below, after the first subcycle
            Sig g2 = g1;
            Diag dg;
            double tarear = 0.0;
            if (LAST2) tarear = *reinterpret_cast<const double *>(rb + (size_t)(F_TAREAR >> 1) * pp + lo + (F_TAREAR & 1) * 8);
            stress_cell<LAST2>(mt, u1_c, u1_m, Ys[0], Ys[128], v1_c, v1_m, Ys[64], Ys[192], a.ecci, a.arlx1i, a.denom1, tarear, g2, o2, dg);
            if (own && w <= R && r <= jmax) {
                store_sig(rb, pp, SW, lo, g2);
                if (cyc && c == 1) store_sig(rb, pp, SW, lo + (unsigned)nxl * 16u, g2);     // east ghost T column = image of column 1
                if (LAST2) {
                    st1(rb, pp, F_DIVU, lo, dg.divu);       st1(rb, pp, F_RDGCONV, lo, dg.rdg_conv);
                    st1(rb, pp, F_RDGSHEAR, lo, dg.rdg_shear); st1(rb, pp, F_SHEAR, lo, dg.shear);
                    st1(rb, pp, F_PRSSIG, lo, dg.prs);
                }
            }
        }
    }
    const double b2n = shfl_dn1(o2.s2), b4n = shfl_dn1(o2.s4), b7n = shfl_dn1(o2.s7), b8n = shfl_dn1(o2.s8);
    Xw[0] = o2.s3; Xw[64] = o2.s6; Xw[128] = b4n; Xw[192] = b8n;
    __syncthreads();

    // ---------------- phase D: U2(r) ----------------
    const bool u2act = (w >= 1) && (w <= R) && own && (m & CM_U) != 0 && r <= nyl && r <= jmax;
    if (__any(u2act)) {
        if (u2act) {
            const double *Xn = Xw + 256;
            double un, vn, sxi, syi;
            UStat q1; double ui1, vi1;
            load_q(q1, ui1, vi1);
            stepu_cell(q1, u1_c, v1_c, ui1, vi1, ((o2.s1 + b2n) + Xn[0]) + Xn[128], ((o2.s5 + Xn[64]) + b7n) + Xn[192],
                       a.brlx, a.revp, a.cosw, a.sinw, un, vn, sxi, syi);
            stp(rb, pp, SW + S_U, lo, un, vn);
            if (cyc) {
                if (c == 1) stp(rb, pp, SW + S_U, lo + (unsigned)nxl * 16u, un, vn);
                if (c == nxl) stp(rb, pp, SW + S_U, lo - (unsigned)nxl * 16u, un, vn);
            }
            if (LAST2) { st1(rb, pp, F_STRINTX, lo, sxi); st1(rb, pp, F_STRINTY, lo, syi); }
        }
    }
    if (w == 0 && !mir) dbg_stamp(a, wg, 1);
}

// Workgroups of up to sixteen waves (tile heights up to 13): 128 VGPRs.  The LAST2 variants -- one launch per evp, with the ridging
// diagnostics -- need a dozen more and spilled them to scratch memory; for workgroups of up to eight waves (the tuner's tile height 5)
// they are compiled for 256 VGPRs instead (k_subcycle2t8): NO kernel on the default evp path uses scratch memory since round 5.
// (Both unexplained one-plane differences, rounds 4 and 5, were zeros in a stress plane after the FIRST evp of a context on a small
// grid: the one launch of this kernel's LAST2 variant, the only spilling kernel on that path -- docs/NOTEBOOK.md R5.11.)
template <bool REVP, bool LAST2, bool XM = false>
__global__ __launch_bounds__(1024) void k_subcycle2t(SubArgs a) { subcycle2t_body<REVP, LAST2, XM>(a); }
template <bool REVP>
__global__ __launch_bounds__(512) void k_subcycle2t8(SubArgs a) { subcycle2t_body<REVP, true, false>(a); }

template __global__ void k_subcycle2t<false, false>(SubArgs);
template __global__ void k_subcycle2t<true, false>(SubArgs);
template __global__ void k_subcycle2t<false, true>(SubArgs);
template __global__ void k_subcycle2t<true, true>(SubArgs);
template __global__ void k_subcycle2t<false, false, true>(SubArgs);
template __global__ void k_subcycle2t<true, false, true>(SubArgs);
template __global__ void k_subcycle2t8<false>(SubArgs);
template __global__ void k_subcycle2t8<true>(SubArgs);

// ------------------------------------------------------------------------------------
// k_subcycle2r (round 5): k_subcycle2t that ROLLS north -- the one-row-per-wave tile WITHOUT its redundant rows.
// A tile of k_subcycle2t computes T1 on R + 3 rows, U1 on R + 2, T2 on R + 1 for R owned rows (8, 7, 6 for 5: 1.4 x the stress
// work), and the small slabs it serves are bound by fp64 issue, not by latency (450 x 2700: 17 K wave-rows x 1 450 instructions on
// 1 024 SIMDs = the 42 us the launch takes).  Here a workgroup of NW waves takes a strip of R >= NW - 3 rows and works through it
// in PASSES of the same four phases; a row that could not finish in a pass because the row above it had not started (the top
// row has T1 only, the one below it lacks U2) is KEPT by its wave -- sigma after the first subcycle, the metrics, its str terms
// stay in that wave's registers -- and finishes in the next pass beside NW - 2 new rows:
//   pass p:  A  T1 of the new rows s .. t            (s = 0, t = NW - 1 in the first pass; then s = t' + 1, t = s + NW - 3)
//            B  U1 of rows s - 1 .. t - 1            (needs T1 of the row above: LDS)
//            C  T2 of rows s - 1 .. t - 1            (needs U1 of the row below: LDS; sigma stored)
//            D  U2 of rows s - 2 .. t - 2            (needs T2 of the row above: LDS; (u, v) stored)
//            rows <= t - 2 are complete: their waves take rows + NW
// Row q of the strip lives on wave q mod NW for all four phases, so the LDS rows of k_subcycle2t serve unchanged (the row above /
// below is the next / previous wave, cyclically).  Every pass runs NW - 2 rows through every phase: R + 3 T1 rows, R + 2 U1, R + 1
// T2 per R owned rows of a STRIP instead of a five-row tile.  Four barriers per pass (the fourth keeps a fast wave's next T1 terms
// out of the LDS row a slow wave's U2 still reads).  Same strips, lists, column / ghost-zone / band (jmax) rules and arithmetic as
// k_subcycle2t and k_subcycle2p: bit-identical.
// ------------------------------------------------------------------------------------
constexpr int ROLL_NW = 8;                           // waves per workgroup (two workgroups per CU)
constexpr size_t ROLL_LDS_PER_WAVE = 10 * 1024;      // X 2 KiB + Y 1 + Z 3 + Q 4
template <bool REVP, bool LAST2, bool XM = false>
__global__ __launch_bounds__(512, 4) void k_subcycle2r(SubArgs a) {
    extern __shared__ double tl[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = blockDim.x >> 6;
    const int nband8 = (a.nband + 7) & ~7;
    const int nmir8 = XM ? ((a.nmir + 7) & ~7) : 0;
    const int vb = a.band_last ? (int)((blockIdx.x + (unsigned)(nband8 + nmir8)) % gridDim.x) : (int)blockIdx.x;      // (see k_subcycle2p)
    if (vb < nband8) {                   // tripole top band of this pair (one rank; the launch asks for the LDS)
        if (vb < a.nband) band_pair<REVP, LAST2>(a, vb, tl);
        return;
    }
    const bool mir = XM && vb - nband8 < nmir8;
    const SlabV s = slab_view(a, mir);
    const int jmax = mir ? a.mjmax : a.jmax;
    int wg, st;
    if (mir) {
        wg = vb - nband8;
        if (wg >= a.nmir) return;                     // (the whole workgroup leaves: no barrier is left waiting)
        st = wg;
    } else {
        const int bidx = vb - nband8 - nmir8;
        const int ns = pair_nstrips(a);
        const int chunk = a.nsdev ? (ns + 7) >> 3 : ((int)gridDim.x - nband8 - nmir8) >> 3;
        wg = (bidx & 7) * chunk + (bidx >> 3);
        if ((bidx >> 3) >= chunk || wg >= ns) return; // (likewise)
        st = __builtin_amdgcn_readfirstlane(a.strips[wg]);
        if (w == 0) dbg_stamp(a, wg, 0);
    }
    const int cx = st % a.ncx, ry = st / a.ncx;
    const int R = a.R, nxl = s.nxl, nyl = s.nyl;
    const int G = a.G;
    const int c = cx * STRIP2_W + lane - G;
    const int jb = ry * R + 1;
    const bool cyc = a.wrap != 0;
    int ci = c, cm1 = c - 1;
    bool okc, okm;
    if (cyc) {
        ci = (c - 1) % nxl; if (ci < 0) ci += nxl; ci += 1;
        cm1 = (c - 2) % nxl; if (cm1 < 0) cm1 += nxl; cm1 += 1;
        okc = okm = true;
    } else {
        okc = (c >= -1 - G && c <= nxl + 2 + G);
        okm = (cm1 >= -1 - G && cm1 <= nxl + 2 + G);
        if (!okc) ci = 0;
        if (!okm) cm1 = 0;
    }
    const bool tcol = cyc ? true : (c >= -G && c <= nxl + 2 + G);
    const bool ucol = cyc ? true : (c >= -G && c <= nxl + 1 + G);
    const bool own = (lane >= 1 && lane <= STRIP2_W && c >= 1 - G && c <= nxl + G);

    const size_t pp0 = (size_t)s.pitch * 16;
    const size_t rowb = (size_t)s.rstride * 16;
    const unsigned lo = (unsigned)(C0 + ci) * 16u, lom = (unsigned)(C0 + cm1) * 16u;
    const int SR = a.sr, SW = a.sw;
    char *const base = reinterpret_cast<char *>(s.F);
    // LDS per wave: X 4 x 64 doubles (the str terms the row below needs), Y 2 x 64 ((u, v) after the first subcycle; the column to the
    // west is the lane below), Z 6 x 64 (what the row's own later phases need of its stresses: nothing of a row is carried in
    // registers from one stress to the next except sigma and the metrics), Q 4 x 64 double2 (the stepu inputs): 10 KiB, two
    // workgroups of eight waves per CU
    double *const X = tl, *const Y = tl + (size_t)NW * 256, *const Z = tl + (size_t)NW * 384;
    const int wup = (w + 1 == NW) ? 0 : w + 1, wdn = (w == 0) ? NW - 1 : w - 1;      // the waves of the rows above / below mine
    double *const Xw = X + (size_t)w * 256 + lane, *const Yw = Y + (size_t)w * 128 + lane, *const Zw = Z + (size_t)w * 384 + lane;
    const double *const Xn = X + (size_t)wup * 256 + lane, *const Ys = Y + (size_t)wdn * 128 + lane;
    double2 *const Qw = reinterpret_cast<double2 *>(tl + (size_t)NW * 768) + (size_t)w * 4 * 64;
    const unsigned Qb = (unsigned)(size_t)(__attribute__((address_space(3))) void *)Qw;      // (LDS byte address, uniform: w is)
    auto load_q = [&](UStat &q, double &ui, double &vi, const char *rbq) {
        const double2 va = Qw[0 * 64 + lane], oc = Qw[1 * 64 + lane], fo = Qw[2 * 64 + lane], mf = Qw[3 * 64 + lane];
        q = UStat{va.x, va.y, oc.x, oc.y, fo.x, fo.y, mf.x, mf.y};
        ui = 0.0; vi = 0.0;
        if (REVP) { const double2 iv = ldp(rbq, pp0, F_UVEL_INIT, lo); ui = iv.x; vi = iv.y; }      // (revised EVP only: read where it is used)
    };

    const int qmax = R + 2;                           // rows q = 0 .. R + 2 of the strip: r = jb - 1 + q
    int q = w;                                        // the row this wave holds
    int sp = 0, tp = (NW - 1 < qmax) ? NW - 1 : qmax; // T1 runs on rows sp .. tp in this pass
    // the state of the held row (lives across passes while the row waits for the rows above it)
    Sig g1{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // sigma after the first subcycle: the one thing a row keeps in registers between its stresses

    for (;;) {
        // (the plane offsets inside a row -- some twenty products with the pitch -- are made again in every pass: as loop invariants the
        //  compiler kept them all in scalar registers, spilled 34 of those and fetched them back lane by lane)
        size_t pp = pp0;
        asm volatile("" : "+s"(pp));
        const unsigned pp32 = (unsigned)pp;
        // ---------------- phase A: T1 of the new rows ----------------
        // (the mask byte, the old velocity and the metrics of the held row are re-read where a later phase needs them -- they come out
        //  of the L2 -- instead of living in registers across the passes: the register budget is the tile kernel's)
        const int r = jb - 1 + q;
        const bool rowok = (r >= 0 && r <= nyl + 1);
        char *const rb = base + (size_t)(rowok ? r : 0) * rowb;
        const unsigned char m = (rowok && okc && q <= qmax) ? s.cmask[(size_t)r * s.pitch + C0 + ci] : (unsigned char)0;
        const bool t1act = tcol && (m & CM_T) != 0;
        if (q >= sp && q <= tp) {
            double un_c = 0.0, vn_c = 0.0;
            double un_m = 0.0, vn_m = 0.0, uo_c = 0.0, vo_c = 0.0, uo_m = 0.0, vo_m = 0.0;
            if (rowok) {
                if (okc) { const double2 v = ldp(rb, pp, SR + S_U, lo); un_c = v.x; vn_c = v.y; }
                if (okm) { const double2 v = ldp(rb, pp, SR + S_U, lom); un_m = v.x; vn_m = v.y; }
            }
            if (r - 1 >= 0 && r - 1 <= nyl + 1) {
                const char *rs = base + (size_t)(r - 1) * rowb;
                if (okc) { const double2 v = ldp(rs, pp, SR + S_U, lo); uo_c = v.x; vo_c = v.y; }
                if (okm) { const double2 v = ldp(rs, pp, SR + S_U, lom); uo_m = v.x; vo_m = v.y; }
            }
            const bool u1need = (q <= qmax - 1) && ucol && (m & CM_U) != 0 && r >= 1 && r <= nyl;
            Str8 o1{0, 0, 0, 0, 0, 0, 0, 0};
            g1 = Sig{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            TMet mt{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (__any(t1act)) {
                if (t1act) { mt = load_tmet(rb, pp, lo); g1 = load_sig(rb, pp, SR, lo); }
            }
            if (__any(u1need)) {      // the stepu inputs of this row serve both subcycles: global -> LDS directly
                if (u1need) {
                    lds_dma16s<0>(rb + (size_t)((unsigned)(F_VRELC >> 1) * pp32), lo, Qb);
                    lds_dma16s<1>(rb + (size_t)((unsigned)(F_UOCN >> 1) * pp32), lo, Qb);
                    lds_dma16s<2>(rb + (size_t)((unsigned)(F_FORCEX >> 1) * pp32), lo, Qb);
                    lds_dma16s<3>(rb + (size_t)((unsigned)(F_UMASSDTI >> 1) * pp32), lo, Qb);
                }
            }
            if (__any(t1act)) {
                if (t1act) {
                    Diag dg;
                    stress_cell<false>(mt, un_c, un_m, uo_c, uo_m, vn_c, vn_m, vo_c, vo_m, a.ecci, a.arlx1i, a.denom1, 0.0, g1, o1, dg);
                }
            }
            const double a2n = shfl_dn1(o1.s2), a7n = shfl_dn1(o1.s7), a4n = shfl_dn1(o1.s4), a8n = shfl_dn1(o1.s8);
            Xw[0] = o1.s3; Xw[64] = o1.s6; Xw[128] = a4n; Xw[192] = a8n;
            Zw[0] = o1.s1 + a2n; Zw[64] = o1.s5; Zw[128] = a7n;      // (for this row's U1, this pass or the next)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (this wave's LDS-DMA has landed)
        }
        __syncthreads();

        // ---------------- phase B: U1 of rows sp - 1 .. tp - 1 ----------------
        const bool doB = (q >= sp - 1 && q <= tp - 1 && q >= 0);
        if (doB) {
            const bool u1act = ucol && (m & CM_U) != 0 && r >= 1 && r <= nyl;
            double un_c = 0.0, vn_c = 0.0;
            if (rowok && okc) { const double2 v = ldp(rb, pp, SR + S_U, lo); un_c = v.x; vn_c = v.y; }
            double u1_c = un_c, v1_c = vn_c;          // an inactive cell keeps its velocity
            if (__any(u1act)) {
                if (u1act) {
                    double sxi, syi;
                    UStat q1; double ui1, vi1;
                    load_q(q1, ui1, vi1, rb);
                    stepu_cell(q1, un_c, vn_c, ui1, vi1, (Zw[0] + Xn[0]) + Xn[128], ((Zw[64] + Xn[64]) + Zw[128]) + Xn[192],
                               a.brlx, a.revp, a.cosw, a.sinw, u1_c, v1_c, sxi, syi);
                }
            }
            Yw[0] = u1_c; Yw[64] = v1_c;              // (phases C and D of this row read them back from here; (c-1, r) is the lane below)
        }
        __syncthreads();

        // ---------------- phase C: T2 of rows sp - 1 .. tp - 1 (from row 1 of the strip) ----------------
        const bool doC = doB && q >= 1;
        if (doC) {
            const bool t2act = t1act && lane >= 1;
            Str8 o2{0, 0, 0, 0, 0, 0, 0, 0};
            if (__any(t2act)) {
                if (t2act) {
                    const TMet mt = load_tmet(rb, pp, lo);
                    Sig g2 = g1;
                    Diag dg;
                    double tarear = 0.0;
                    if (LAST2) tarear = *reinterpret_cast<const double *>(rb + (size_t)(F_TAREAR >> 1) * pp + lo + (F_TAREAR & 1) * 8);
                    stress_cell<LAST2>(mt, Yw[0], Yw[-1], Ys[0], Ys[-1], Yw[64], Yw[63], Ys[64], Ys[63], a.ecci, a.arlx1i, a.denom1, tarear, g2, o2, dg);
                    if (own && q <= R && r <= jmax) {
                        store_sig(rb, pp, SW, lo, g2);
                        if (cyc && c == 1) store_sig(rb, pp, SW, lo + (unsigned)nxl * 16u, g2);     // east ghost T column = image of column 1
                        if (LAST2) {
                            st1(rb, pp, F_DIVU, lo, dg.divu);       st1(rb, pp, F_RDGCONV, lo, dg.rdg_conv);
                            st1(rb, pp, F_RDGSHEAR, lo, dg.rdg_shear); st1(rb, pp, F_SHEAR, lo, dg.shear);
                            st1(rb, pp, F_PRSSIG, lo, dg.prs);
                        }
                    }
                }
            }
            const double b2n = shfl_dn1(o2.s2), b7n = shfl_dn1(o2.s7), b4n = shfl_dn1(o2.s4), b8n = shfl_dn1(o2.s8);
            Xw[0] = o2.s3; Xw[64] = o2.s6; Xw[128] = b4n; Xw[192] = b8n;
            Zw[192] = o2.s1 + b2n; Zw[256] = o2.s5; Zw[320] = b7n;      // (for this row's U2)
        }
        __syncthreads();

        // ---------------- phase D: U2 of rows sp - 2 .. tp - 2 (rows 1 .. R of the strip) ----------------
        const bool doD = (q >= sp - 2 && q <= tp - 2 && q >= 1 && q <= R);
        if (doD) {
            const bool u2act = own && (m & CM_U) != 0 && r <= nyl && r <= jmax;
            if (__any(u2act)) {
                if (u2act) {
                    double un, vn, sxi, syi;
                    UStat q1; double ui1, vi1;
                    load_q(q1, ui1, vi1, rb);
                    stepu_cell(q1, Yw[0], Yw[64], ui1, vi1, (Zw[192] + Xn[0]) + Xn[128], ((Zw[256] + Xn[64]) + Zw[320]) + Xn[192],
                               a.brlx, a.revp, a.cosw, a.sinw, un, vn, sxi, syi);
                    stp(rb, pp, SW + S_U, lo, un, vn);
                    if (cyc) {
                        if (c == 1) stp(rb, pp, SW + S_U, lo + (unsigned)nxl * 16u, un, vn);
                        if (c == nxl) stp(rb, pp, SW + S_U, lo - (unsigned)nxl * 16u, un, vn);
                    }
                    if (LAST2) { st1(rb, pp, F_STRINTX, lo, sxi); st1(rb, pp, F_STRINTY, lo, syi); }
                }
            }
        }
        if (tp >= qmax) break;                        // the strip's last row has had its T1: everything finished in this pass
        if (q <= tp - 2) q += NW;                     // my row is complete: the next one of this wave
        sp = tp + 1;
        tp = (sp + NW - 3 < qmax) ? sp + NW - 3 : qmax;
        __syncthreads();                              // (a slow wave's phase D still reads the LDS row a new T1 is about to rewrite)
    }
    if (w == 0 && !mir) dbg_stamp(a, wg, 1);
}

template __global__ void k_subcycle2r<false, false>(SubArgs);
template __global__ void k_subcycle2r<true, false>(SubArgs);
template __global__ void k_subcycle2r<false, true>(SubArgs);
template __global__ void k_subcycle2r<true, true>(SubArgs);
template __global__ void k_subcycle2r<false, false, true>(SubArgs);
template __global__ void k_subcycle2r<true, false, true>(SubArgs);

// strip activity for k_subcycle2: any active T / U cell in the window the strip touches
// (columns c0..c0+63 wrapped, rows jb-1..jb+R+1)
// cells: if given, also counts the active T / U cells on the physical cells each strip owns (icellt, icellu of the rank)
// W, own0, rmar: strip width, first owned lane and the rows read beyond the owned ones -- 61, 1, 1 for the pair kernels, 59, 2, 2
// for k_subcycle3w
__global__ void k_strip_flags2(Slab s, int ncx, int nry, int R, int cyc, int G, unsigned char *flags, unsigned int *count,
                               unsigned long long *cells = nullptr, int W = STRIP2_W, int own0 = 1, int rmar = 1,
                               unsigned char *work = nullptr /* rows of the window with an active cell (<= 255): the strip's run time */) {
    const int sid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (sid >= ncx * nry) return;
    const int cx = sid % ncx, ry = sid / ncx;
    const int c = cx * W + lane - (own0 - 1) - G;
    int ci = c;
    bool ok;
    if (cyc) { ci = (c - 1) % s.nxl; if (ci < 0) ci += s.nxl; ci += 1; ok = true; }
    else ok = (c >= -G && c <= s.nxl + 2 + G);
    const int jb = ry * R + 1;
    int any = 0, nt = 0, nu = 0, nrows = 0;
    const bool owned = (lane >= own0 && lane <= own0 + W - 1 && c >= 1 && c <= s.nxl);
    {   // quick reject: cmask is non-zero only where k_prep2 ran, i.e. in tiles of act_any -- 74 % of the bench grid has none
        // (tiles: TILE_X x TILE_Y cells, the thread blocks of the per-evp kernels; columns beyond the ring count as the edge tile)
        const int jlo = max(jb - rmar, 1), jhi = min(jb + R + rmar, s.nyl + 1);
        const int tx = min(max(ci, 0), s.nxl + 1) / TILE_X;
        bool hit = false;
        if (ok)
            for (int ty = jlo / TILE_Y; ty <= jhi / TILE_Y; ty++) hit = hit || s.act_any[ty * s.ntx + tx] != 0;
        if (!__any(hit)) {
            if (lane == 0) { if (flags) flags[sid] = 0; if (work) work[sid] = 0; }
            return;
        }
    }
    const int rend = jb + R + rmar;
    for (int r0 = jb - rmar; r0 <= rend; r0 += 8) {           // eight rows' mask bytes in flight at a time
        unsigned char mm[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int r = r0 + q;
            mm[q] = (ok && r >= 1 && r <= s.nyl + 1 && r <= rend) ? s.cmask[mcell(s, ci, r)] : (unsigned char)0;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int r = r0 + q;
            const unsigned char m = mm[q];
            if (m) any = 1;
            if (work && __ballot(m != 0)) nrows++;
            if (owned && r >= jb && r < jb + R && r <= s.nyl) { nt += (m & CM_T) ? 1 : 0; nu += (m & CM_U) ? 1 : 0; }
        }
    }
    if (work && lane == 0) work[sid] = (unsigned char)min(nrows, 255);
    const unsigned long long b = __ballot(any);
    if (cells) {
        for (int o = 32; o > 0; o >>= 1) { nt += __shfl_down(nt, o); nu += __shfl_down(nu, o); }
        if (lane == 0) {
            if (nt) atomicAdd(&cells[0], (unsigned long long)nt);
            if (nu) atomicAdd(&cells[1], (unsigned long long)nu);
        }
    }
    if (lane == 0) {
        if (flags) flags[sid] = b ? 1 : 0;
        if (count && b) atomicAdd(count, 1u);
    }
}

// ---- rows of pair planes <-> a message, for the mirror slab M of band_pair (x-slab ranks, tripole): pair planes L.f[0 .. np)
// (even field ids), rows r0 .. r0+nr-1, the columns of the segment list G (up to four runs of columns: a run of the SENDER's
// own columns that maps onto a run of the receiver's mirror slab -- M is a virtual slab of the receiver's width that starts at
// global column nx - i0 - w + 2, so its columns come from the one, two or three ranks whose slabs cover that range, cyclically,
// the way ice_HaloUpdate's tripole messages come from whichever blocks hold the mirrored columns, mpi/ice_boundary.F90:2737-2913),
// then (mask != 0) the cmask bytes of the same rows and columns.  Message layout: double2 [np][nr][G.tot], bytes [nr][G.tot]. ----
struct XbList { int f[16]; int np; };
struct XbSeg { int n, tot; int len[4]; int c0[4]; };          // c0: first local column of the run (pack: in my slab; unpack: in M)
__device__ __forceinline__ int xb_col(const XbSeg &G, int x) {
    int c = G.c0[0] + x;
    for (int k = 1, o = G.len[0]; k < G.n; o += G.len[k], k++) if (x >= o) c = G.c0[k] + (x - o);
    return c;
}
__global__ void k_xband_pack(Slab s, XbList L, XbSeg G, int r0, int nr, int mask, double2 *msg) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, ncol = G.tot;
    if (x >= ncol) return;
    const int y = blockIdx.y, c = xb_col(G, x);
    if (y < L.np * nr) {
        const int p = y / nr, k = y - p * nr;
        msg[(size_t)y * ncol + x] = *reinterpret_cast<const double2 *>(&FD(s, L.f[p], cell(s, c, r0 + k)));
    } else if (mask) {
        const int k = y - L.np * nr;
        reinterpret_cast<unsigned char *>(msg + (size_t)L.np * nr * ncol)[(size_t)k * ncol + x] = s.cmask[mcell(s, c, r0 + k)];
    }
}
__global__ void k_xband_unpack(Slab m, XbList L, XbSeg G, int r0, int nr, int mask, const double2 *msg) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, ncol = G.tot;
    if (x >= ncol) return;
    const int y = blockIdx.y, c = xb_col(G, x);
    if (y < L.np * nr) {
        const int p = y / nr, k = y - p * nr;
        *reinterpret_cast<double2 *>(&FD(m, L.f[p], cell(m, c, r0 + k))) = msg[(size_t)y * ncol + x];
    } else if (mask) {
        const int k = y - L.np * nr;
        m.cmask[mcell(m, c, r0 + k)] = reinterpret_cast<const unsigned char *>(msg + (size_t)L.np * nr * ncol)[(size_t)k * ncol + x];
    }
}

// rows j0 .. j1 of nf fields of the mirror slab, every column it has (ghost zones included): fdst <- fsrc
__global__ void k_xband_rows_copy(Slab m, int fsrc, int fdst, int nf, int j0, int j1) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, ncol = m.nxl + 2 * ZW_MAX;
    const int j = j0 + blockIdx.y;
    if (x >= ncol || j > j1) return;
    const size_t k = cell(m, 1 - ZW_MAX + x, j);
    for (int q = 0; q < nf; q++) FD(m, fdst + q, k) = FD(m, fsrc + q, k);
}

// the list of flagged strips, in order, and its length -- on the device, so that evpk_prep need not wait for the flags, compact
// them on the host and send the list back: one workgroup of 1024 threads walks the flags 1024 at a time (ballot + wave offsets)
__global__ __launch_bounds__(1024) void k_compact_strips(const unsigned char *flags, int n, int *list, int *count) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += 1024) {
        const int k = k0 + (int)threadIdx.x;
        const bool f = k < n && flags[k] != 0;
        const unsigned long long b = __ballot(f);
        if (lane == 0) wsum[w] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int q = 0; q < w; q++) off += wsum[q];
        if (f) list[off + __popcll(b & ((1ull << lane) - 1ull))] = k;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < 16; q++) t += wsum[q]; base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}

// the active strips (work > 0) in order of DESCENDING work -- longest first, so that the workgroups the dispatcher hands out last
// are the short ones and the launch's tail is short (k_subcycle3w runs its strips in two or three rounds of resident workgroups).
// One workgroup: histogram, offsets, scatter (the order inside a class of equal work is arbitrary: it changes no result).
__global__ __launch_bounds__(1024) void k_sort_strips(const unsigned char *work, int n, int *list, int *count) {
    __shared__ int hist[256], cur[256];
    for (int k = threadIdx.x; k < 256; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) if (work[k]) atomicAdd(&hist[work[k]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int off = 0;
        for (int w = 255; w >= 1; w--) { cur[w] = off; off += hist[w]; }
        *count = off;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) if (work[k]) list[atomicAdd(&cur[work[k]], 1)] = k;
}

// ------------------------------------------------------------------------------------
// transport_upwind (source/ice_transport_driver.F90:634-772) on the velocities resident on the device
// ------------------------------------------------------------------------------------
// cell-edge velocities (:688-701): uee = p5*(uvel(i,j)+uvel(i,j-1)), vnn = p5*(vvel(i,j)+vvel(i-1,j)), physical cells
__global__ void k_edge_vel(Slab s, int SB, int fu, int fv) {
    SLAB_IJ_ALL
    double ue = 0.0, vn = 0.0;
    if (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl) {
        ue = 0.5 * (FD(s, SB + S_U, k) + FD(s, SB + S_U, cell(s, i, j - 1)));
        vn = 0.5 * (FD(s, SB + S_V, k) + FD(s, SB + S_V, cell(s, i - 1, j)));
    }
    FD(s, fu, k) = ue;
    FD(s, fv, k) = vn;
}

// upwind_field (:1614-1689) for one array: phi, out are plain planes indexed like the masks (mcell)
__global__ void k_upwind(Slab s, double dt, int fu, int fv, const double *phi, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (i > s.nxl || j > s.nyl) return;
    // upwind(y1,y2,a,h) = p5*dt*h*((a+abs(a))*y1+(a-abs(a))*y2)   (:1661)
    auto up = [&](double y1, double y2, double a, double h) { return 0.5 * dt * h * ((a + fabs(a)) * y1 + (a - fabs(a)) * y2); };
    const size_t k = cell(s, i, j), kw = cell(s, i - 1, j), ks = cell(s, i, j - 1);
    const double p = phi[mcell(s, i, j)];
    const double wa = up(p, phi[mcell(s, i + 1, j)], FD(s, fu, k), FD(s, F_HTE, k));
    const double wam = up(phi[mcell(s, i - 1, j)], p, FD(s, fu, kw), FD(s, F_HTE, kw));
    const double wb = up(p, phi[mcell(s, i, j + 1)], FD(s, fv, k), FD(s, F_HTN, k));
    const double wbm = up(phi[mcell(s, i, j - 1)], p, FD(s, fv, ks), FD(s, F_HTN, ks));
    out[mcell(s, i, j)] = p - (wa - wam + wb - wbm) / FD(s, F_TAREA, k);        // :1680-1682
}

// physical cells of a plain plane -> one (nx_block, ny_block) slice per block, blocks `bstride` doubles apart
// ---- transport_upwind with its state transforms (ice_transport_driver.F90:634-772): state_to_work (:1382-1513) inside the
// gather, work_to_state (:1520-1609) with compute_tracers (ice_itd.F90:1359-1501) and bound_state inside the scatter ----
constexpr int UW_MAXT = 32;
struct UpwState {
    double *aicen, *vicen, *vsnon, *trcrn;      // block arrays (nb, ncat, ny, nx) x 3, (nb, ncat, ntrcr_dim, ny, nx)
    int ncat, ntrcr, ntrcr_dim;
    int nt_Tsfc, nt_fbri;                       // 1-based, 0 = absent
    // per tracer: works = base * trcrn(m1) * trcrn(m2) * trcrn(it) in this order (base 0 aicen, 1 vicen, 2 vsnon, -1: no rule --
    // the reference leaves works as allocated and compute_tracers returns 0); m1, m2 1-based tracers or 0
    signed char base[UW_MAXT], m1[UW_MAXT], m2[UW_MAXT];
    // compute_tracers: rule 0 Tsfc, 1 area, 2 ice volume, 3 snow volume, 4 d = trcrn(d1) * aicen, 5 d = trcrn(d1) * trcrn(d2) * aicen,
    // 6 d = trcrn(d1) * vicen, -1 none
    signed char rule[UW_MAXT], d1[UW_MAXT], d2[UW_MAXT];
    double Tocnfrz;
};

// q = 0, 1, 2: aicen, vicen, vsnon of category n; q = 3 + it - 1: the product of tracer it
__global__ void k_upw_gather(Slab s, const BlockDesc *bd, int nxb, int nyb, UpwState u, int n, int q, double *dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1;
    const int sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    if (si < 0 || si > s.nxl + 1 || sj < 0 || sj > s.nyl + 1) return;
    if (!gather_take(s, d, i, j, si, sj)) return;
    const size_t nn = (size_t)nyb * nxb, o = (size_t)(j - 1) * nxb + (i - 1), bc = ((size_t)b * u.ncat + n) * nn + o;
    double v;
    if (q < 3) v = (q == 0 ? u.aicen : q == 1 ? u.vicen : u.vsnon)[bc];
    else {
        const int it = q - 3;                                       // 0-based tracer
        const double *t = u.trcrn + ((size_t)b * u.ncat + n) * u.ntrcr_dim * nn + o;
        const int bs = u.base[it];
        if (bs < 0) v = 0.0;
        else {
            v = (bs == 0 ? u.aicen : bs == 1 ? u.vicen : u.vsnon)[bc];
            if (u.m1[it]) v = v * t[(size_t)(u.m1[it] - 1) * nn];
            if (u.m2[it]) v = v * t[(size_t)(u.m2[it] - 1) * nn];
            v = v * t[(size_t)it * nn];
        }
    }
    dst[mcell(s, si, sj)] = v;
}

// planes[0..2] = the new aicen, vicen, vsnon of category n, planes[3 + it - 1] = the new products; ghost ring halo-updated.
// Every cell of every block with a source is written (physical cells: work_to_state; ghost cells: bound_state, whose values are
// the neighbour's -- the same arithmetic on the same numbers); a ghost cell without one gets bound_state's fill (0) on the
// outermost row / column of the array and keeps the caller's value elsewhere (as k_scatter_halo).
// fcov >= 0: the halo-updated coverage plane -- a ghost cell whose source lies in an eliminated land block gets the halo fill 0 in
// every state array (bound_state's ice_HaloUpdate, srcBlock == 0), not compute_tracers of an empty cell.
__global__ void k_upw_scatter(Slab s, const BlockDesc *bd, int nxb, int nyb, UpwState u, int n, double *const *planes, int cyclic, int tripole,
                              int fcov) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    const BlockDesc d = bd[b];
    const size_t nn = (size_t)nyb * nxb, o = (size_t)(j - 1) * nxb + (i - 1), bc = ((size_t)b * u.ncat + n) * nn + o;
    double *t = u.trcrn + ((size_t)b * u.ncat + n) * u.ntrcr_dim * nn + o;
    const bool edge = (i == 1 || i == nxb || j == 1 || j == nyb);
    const bool phys = (i >= d.ilo && i <= d.ihi && j >= d.jlo && j <= d.jhi);
    const int si = d.iglob_lo + (i - d.ilo) - s.i0 + 1, sj = d.jglob_lo + (j - d.jlo) - s.j0 + 1;
    const int gi = s.i0 + si - 1, gj = s.j0 + sj - 1;
    const bool padding = (i > d.ihi + 1 || j > d.jhi + 1);
    const bool has_src = phys || (!padding && ((gi >= 1 && gi <= s.nxg) || cyclic) && ((gj >= 1 && gj <= s.nyg) || (tripole && gj == s.nyg + 1)));
    const bool landsrc = has_src && !phys && fcov >= 0 && FD(s, fcov, cell(s, si, sj)) == 0.0;
    if ((!has_src && edge) || landsrc) {                // bound_state's fill on the outermost row / column, or next to a land block
        u.aicen[bc] = 0.0; u.vicen[bc] = 0.0; u.vsnon[bc] = 0.0;
        for (int it = 0; it < u.ntrcr; it++) t[(size_t)it * nn] = 0.0;
        return;
    }
    // a cell nothing reaches (padding, a ghost cell beyond an open / closed boundary inside the array): upwind_field leaves its
    // works alone, so work_to_state hands back compute_tracers(state_to_work(old values)) -- evaluated here from the cell itself
    double a, v, sn, atl[UW_MAXT];
    if (has_src) {
        const size_t km = mcell(s, si, sj);
        a = planes[0][km]; v = planes[1][km]; sn = planes[2][km];
        for (int it = 0; it < u.ntrcr; it++) atl[it] = planes[3 + it][km];
    } else {
        a = u.aicen[bc]; v = u.vicen[bc]; sn = u.vsnon[bc];
        for (int it = 0; it < u.ntrcr; it++) {
            const int bs = u.base[it];
            double w = 0.0;
            if (bs >= 0) {
                w = bs == 0 ? a : bs == 1 ? v : sn;
                if (u.m1[it]) w = w * t[(size_t)(u.m1[it] - 1) * nn];
                if (u.m2[it]) w = w * t[(size_t)(u.m2[it] - 1) * nn];
                w = w * t[(size_t)it * nn];
            }
            atl[it] = w;
        }
    }
    u.aicen[bc] = a; u.vicen[bc] = v; u.vsnon[bc] = sn;
    const double puny = 1.0e-11;
    for (int it = 0; it < u.ntrcr; it++) t[(size_t)it * nn] = 0.0;                   // trcrn(:,:,:) = c0 (ice_itd.F90:1405)
    for (int it = 0; it < u.ntrcr; it++) {
        const double at = atl[it];
        double r = 0.0;
        switch (u.rule[it]) {
        case 0: r = a > puny ? at / a : u.Tocnfrz; break;
        case 1: r = a > puny ? at / a : 0.0; break;
        case 2: r = v > 0.0 ? at / v : ((it + 1 == u.nt_fbri) ? 1.0 : 0.0); break;
        case 3: r = sn > 0.0 ? at / sn : 0.0; break;
        case 4: { const double dd = t[(size_t)(u.d1[it] - 1) * nn] * a; r = dd > 0.0 ? at / dd : 0.0; } break;
        case 5: { const double dd = t[(size_t)(u.d1[it] - 1) * nn] * t[(size_t)(u.d2[it] - 1) * nn] * a; r = dd > 0.0 ? at / dd : 0.0; } break;
        case 6: { const double dd = t[(size_t)(u.d1[it] - 1) * nn] * v; r = dd > 0.0 ? at / dd : 0.0; } break;
        default: r = 0.0;
        }
        t[(size_t)it * nn] = r;
    }
}

__global__ void k_scatter_plane(Slab s, const BlockDesc *bd, int nxb, int nyb, const double *src, double *dst, size_t bstride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    const int b = blockIdx.z;
    if (i > nxb) return;
    int si, sj;
    if (!scatter_take(s, bd[b], i, j, MODE_PHYS, si, sj)) return;
    dst[(size_t)b * bstride + (size_t)(j - 1) * nxb + (i - 1)] = src[mcell(s, si, sj)];
}

// ------------------------------------------------------------------------------------
// evp_finish (ice_dyn_shared.F90:757-844)
// ------------------------------------------------------------------------------------
__global__ void k_finish(Slab s, DevParams p, int cur) {
    TILE_SKIP(s.act_any)
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
    // (planes of their own, never used for anything else: tiles that are skipped keep their zeros)
    FD(s, F_WORK3, k) = xT;
    FD(s, F_WORK4, k) = yT;
}

// evp_finish AND the two U->T averages of u2tgrid_vector after it (ice_dyn_evp.F90:487-506) in one pass -- one rank, no exchange.
// u2tgrid_vector averages strocnxT/yT over the U cells (i, j), (i-1, j), (i, j-1), (i-1, j-1): the west ghost column is the
// cyclic image of column nx (or the fill value 0), the south ghost row the fill value, the north ghost row is not read -- what
// the halo update of the two work planes between k_finish and k_to_tgrid2 delivers can be computed where it is needed.  A tile of
// 64 x 4 threads evaluates evp_finish for its own cells and for the 64 + 4 + 1 cells of its south and west rim into the LDS,
// then averages: the work planes are neither written nor read, one launch instead of six (finish, four halo steps, to_tgrid).
__device__ __forceinline__ void finish_cell(const Slab &s, const DevParams &p, int SB, int i, int j, bool cyclic, bool store, double &xT, double &yT) {
    xT = 0.0; yT = 0.0;
    if (i == 0 && cyclic) i = s.nxl;                                              // (ice_HaloUpdate, E-W: ghost column 0 <- column nx)
    if (i < 1 || i > s.nxl || j < 1 || j > s.nyl) return;                         // fill value of the halo update / not a U cell
    if (!s.iceumask[mcell(s, i, j)]) return;
    const size_t k = cell(s, i, j);
    const double u = FD(s, SB + S_U, k), v = FD(s, SB + S_V, k);
    const double du = FD(s, F_UOCN, k) - u, dv = FD(s, F_VOCN, k) - v;
    const double aiu = FD(s, F_AIU, k), fm = FD(s, F_FM, k);
    double vrel = p.rhow * FD(s, F_CW, k) * sqrt(du * du + dv * dv);              // :818-819
    vrel = vrel * aiu;                                                             // :827
    const double sg = copysign(1.0, fm);
    const double sx = vrel * (du * p.cosw - dv * p.sinw * sg);                     // :828-831
    const double sy = vrel * (dv * p.cosw + du * p.sinw * sg);
    if (store) { FD(s, F_STROCNX, k) = sx; FD(s, F_STROCNY, k) = sy; }
    xT = sx / aiu;                                                                 // :840-841
    yT = sy / aiu;
}
// ... and on a tripole grid the NE-corner update of the work pair also REWRITES the top physical row: U points of row ny lie on
// the fold, the update leaves the mean of a point and the image of its mirror point there, the negated value on the two axis
// columns nx/2 and nx (serial/ice_boundary.F90:818-824 and the copy :3752-3776; k_fold_apply, loc = 1, sgn = -1 for a vector)
__device__ __forceinline__ void finish_cell_folded(const Slab &s, const DevParams &p, int SB, int i, int j, bool cyclic, bool tripole, bool store,
                                                   double &xT, double &yT) {
    finish_cell(s, p, SB, i, j, cyclic, store, xT, yT);
    if (!tripole || j != s.nyl) return;
    const int nx = s.nxl, h = nx / 2;                                               // (one rank: the slab is the whole grid)
    const int g = (i == 0 && cyclic) ? nx : i;
    if (g < 1 || g > nx) return;
    const double sgn = -1.0;
    if (g == h || g == nx) { xT = sgn * xT; yT = sgn * yT; return; }
    double bx, by;
    finish_cell(s, p, SB, nx - g, j, cyclic, false, bx, by);
    if (g < h) { xT = sgn * (sgn * (0.5 * (xT + sgn * bx))); yT = sgn * (sgn * (0.5 * (yT + sgn * by))); }
    else       { xT = sgn * (0.5 * (bx + sgn * xT)); yT = sgn * (0.5 * (by + sgn * yT)); }
}
__global__ void k_finish_tgrid(Slab s, DevParams p, int cur, int cyclic, int tripole) {
    TILE_SKIP(s.act_any)
    __shared__ double X[TILE_Y + 1][TILE_X + 1], Y[TILE_Y + 1][TILE_X + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int i = blockIdx.x * blockDim.x + tx, j = blockIdx.y * blockDim.y + ty;
    const int SB = cur ? F_STATE1 : F_STATE0;
    const bool cyc = cyclic != 0, tri = tripole != 0;
    const bool own = (i >= 1 && i <= s.nxl && j >= 1 && j <= s.nyl);
    double xT, yT;
    finish_cell_folded(s, p, SB, i, j, cyc, tri, own, xT, yT);                                 // (thread i = 0: the west ghost column's value)
    X[ty + 1][tx + 1] = xT; Y[ty + 1][tx + 1] = yT;
    if (tx == 0) { finish_cell_folded(s, p, SB, i - 1, j, cyc, tri, false, xT, yT); X[ty + 1][0] = xT; Y[ty + 1][0] = yT; }
    if (ty == 0) { finish_cell_folded(s, p, SB, i, j - 1, cyc, tri, false, xT, yT); X[0][tx + 1] = xT; Y[0][tx + 1] = yT; }
    if (tx == 0 && ty == 0) { finish_cell_folded(s, p, SB, i - 1, j - 1, cyc, tri, false, xT, yT); X[0][0] = xT; Y[0][0] = yT; }
    __syncthreads();
    if (!own) return;
    const size_t k = cell(s, i, j), kw = cell(s, i - 1, j), ks = cell(s, i, j - 1), ksw = cell(s, i - 1, j - 1);
    const double u0 = FD(s, F_UAREA, k), u1 = FD(s, F_UAREA, kw), u2 = FD(s, F_UAREA, ks), u3 = FD(s, F_UAREA, ksw);
    const double ta = FD(s, F_TAREA, k);
#define TG_(A) (0.25 * (((A[ty + 1][tx + 1] * u0 + A[ty + 1][tx] * u1) + A[ty][tx + 1] * u2) + A[ty][tx] * u3) / ta)
    FD(s, F_STROCNXT, k) = TG_(X);
    FD(s, F_STROCNYT, k) = TG_(Y);
#undef TG_
}

// ------------------------------------------------------------------------------------
// principal_stress (ice_dyn_shared.F90:853-893): normalised principal stresses of the NE corner,
// from the resident sigma_1 planes and prs_sig
// ------------------------------------------------------------------------------------
__global__ void k_principal_stress(Slab s, int SB) {
    SLAB_IJ_ALL
    const double puny = 1.0e-11, spval_dbl = 1.0e30;
    const double sp = FD(s, SB + S_SP, k), sm = FD(s, SB + S_SM, k), s12 = FD(s, SB + S_S12, k), prs = FD(s, F_PRSSIG, k);
    double s1 = spval_dbl, s2 = spval_dbl;
    if (prs > puny) {
        const double r = sqrt(sm * sm + 4.0 * (s12 * s12));
        s1 = (0.5 * (sp + r)) / prs;
        s2 = (0.5 * (sp - r)) / prs;
    }
    FD(s, F_SIG1, k) = s1;
    FD(s, F_SIG2, k) = s2;
}

}  // namespace evpk
