// evpk_internal.h -- device data layout shared by the kernels and the host API of libevpk.
//
// HBM layout: every rank owns ONE slab = the bounding rectangle of its ice_blocks blocks
// (nxl x nyl physical cells) plus one ghost ring.  fp64 fields are stored as PAIR planes
// of double2, interleaved by row:
//     F[row j = 0..nyl+1][pair p = 0..NP-1][col = 0..pitch-1] of double2 {field 2p, field 2p+1}
// so cell (i,j) of field f sits at double index  2*((j*NP + f/2)*pitch + C0 + i) + f%2.
//  * the two members of a pair are always consumed together by the hot kernel (cxp/cyp,
//    u/v, stressp_1/2, ...): one 16-byte load per lane (dwordx4, 1 KiB per wave) fetches both;
//  * all planes of one row are contiguous (~2 MB), so a wave marching north touches one
//    or two pages per row instead of ~50 planes 78 MB apart (TLB reach, DRAM locality);
//  * i = 0 is the west ghost column, i = 1..nxl physical, i = nxl+1 the east ghost column;
//    C0 = 7 puts i = 1 on a 128-byte boundary, pitch is a multiple of 8 cells; columns 1-ZW_MAX .. -1 and
//    nxl+2 .. nxl+ZW_MAX are the deeper ghost zones of the two-subcycle kernel between x-slab neighbours.
// Integer / byte masks are separate planes with the same (pitch, C0) but no interleave.
// The ice_blocks decomposition is the unit of host<->device transfer (gather/scatter
// kernels) and of sharding over GPUs; inside a GPU the blocks are fused into the slab so
// that no intra-device halo copies are needed in the subcycle loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evpk {

enum Field : int {
    // ---- hot planes, in pairs (even, odd) ----
    F_CXP, F_CYP, F_CXM, F_CYM, F_DXT, F_DYT, F_DXHY, F_DYHX, F_TINYAREA, F_STRENGTH,
    F_HTN, F_HTE,        // primary grid lengths (optional): the eight metric planes above are functions of them (ice_grid.F90:338-369)
    // prognostic state, double buffered: u, v, stressp_1..4, stressm_1..4, stress12_1..4
    F_STATE0,
    F_STATE1 = F_STATE0 + 14,
    F_STATE2 = F_STATE1 + 14,       // scratch state of the tripole top band between the two fused subcycles
    F_VRELC = F_STATE2 + 14, F_UAREAR, F_UOCN, F_VOCN, F_FORCEX, F_FORCEY,
    F_UMASSDTI, F_FM, F_UVEL_INIT, F_VVEL_INIT,
    // ---- cold planes ----
    F_TAREAR, F_TAREA, F_UAREA, F_FCOR,
    F_AICE, F_VICE, F_VSNO, F_AICE_INIT, F_STRAIRXT, F_STRAIRYT, F_SSTLTX, F_SSTLTY, F_CW,
    F_TMASS, F_UMASS, F_AIU, F_STRAIRX, F_STRAIRY, F_STRTLTX, F_STRTLTY,
    F_WORK1, F_WORK2, F_WORK3, F_WORK4, F_SIG1, F_SIG2, F_ICETM,
    F_DIVU, F_SHEAR, F_RDGCONV, F_RDGSHEAR, F_PRSSIG, F_STRINTX, F_STRINTY,
    F_STROCNX, F_STROCNY, F_STROCNXT, F_STROCNYT,
    F_COUNT
};
constexpr int NSTATE = 14;           // u, v, stressp_1..4, stressm_1..4, stress12_1..4
constexpr int S_U = 0, S_V = 1, S_SP = 2, S_SM = 6, S_S12 = 10;
constexpr int NP = (F_COUNT + 1) / 2;
static_assert(F_STATE0 % 2 == 0 && F_VRELC % 2 == 0, "pairs must start on even field ids");

constexpr int C0 = 7;                // column offset of the west ghost (cells)
constexpr int ZW_MAX = 8;            // widest ghost zone per side (x-slab neighbours): columns 1-ZW_MAX .. 0 and nxl+1 .. nxl+ZW_MAX
static_assert(C0 >= ZW_MAX - 1, "the west ghost zone must fit in front of column 1");
constexpr int STRIP_W = 63;          // U columns per wave strip (64 T columns)

// cmask bits
constexpr unsigned char CM_T = 1;    // icetmask == 1
constexpr unsigned char CM_U = 2;    // iceumask

struct Slab {
    double *F;            // pair planes, see above
    int nxl, nyl;         // physical cells
    int pitch;            // cells per row of one pair plane
    int rstride;          // cells per row of all pair planes = NP * pitch
    int i0, j0;           // global index of local cell (1,1)
    int nxg, nyg;
    int32_t *tmask, *umask, *iceumask;   // int planes, index mcell()
    unsigned char *cmask, *tmphm;        // byte planes, index mcell()
    // activity of the 64x4-cell tiles the per-evp kernels work in (one thread block each): a tile with no ice / no data
    // now and at the previous evp already holds its zeros and is skipped.  act_* are dilated by one tile.
    unsigned char *tile_ice, *tile_dat;  // written by k_prep1a for this evp (undilated)
    unsigned char *act_ice, *act_any;    // dilate(new | prev): ice-dependent kernels / everything
    int ntx, nty;
};
constexpr int TILE_X = 64, TILE_Y = 4;

// cell index inside pair plane 0; field f adds (f/2)*pitch
__host__ __device__ inline size_t cell(const Slab &s, int i, int j) { return (size_t)j * s.rstride + C0 + i; }
__host__ __device__ inline size_t mcell(const Slab &s, int i, int j) { return (size_t)j * s.pitch + C0 + i; }
__host__ __device__ inline double &FD(const Slab &s, int f, size_t k) {
    return s.F[(((size_t)(f >> 1)) * s.pitch + k) * 2 + (f & 1)];
}
__host__ __device__ inline size_t slab_doubles(const Slab &s) { return ((size_t)s.rstride * (s.nyl + 2) + 128) * 2; }
__host__ __device__ inline size_t mask_elems(const Slab &s) { return (size_t)s.pitch * (s.nyl + 2) + 128; }

struct DevParams {
    double dt, revp, ecci, denom1, arlx1i, brlx, cosw, sinw, rhow, rhoi, rhos, gravit, a_min, m_min;
    int ndte, tilt_from_slope, wind_on_ugrid;
    int kstrength, krdg_partic, krdg_redist, ncat;
    double mu_rdg, Cf;
    int sparse_io;
};

// block descriptors on the device (gather / scatter)
struct BlockDesc {
    int ilo, ihi, jlo, jhi, iglob_lo, jglob_lo;
};

}  // namespace evpk
