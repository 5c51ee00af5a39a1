// evpk_internal.h -- device data layout shared by the kernels and the host API of libevpk.
//
// HBM layout: every rank owns ONE slab = the bounding rectangle of its ice_blocks blocks
// (nxl x nyl physical cells) plus one ghost ring, stored as SoA planes of fp64:
//     F[field][row j = 0..nyl+1][col = 0..pitch-1],   cell (i,j) at  j*pitch + c0 + i
// i = 0 is the west ghost column, i = 1..nxl physical, i = nxl+1 the east ghost column;
// c0 = 15 puts i = 1 on a 128-byte boundary, pitch is a multiple of 16 doubles.
// The ice_blocks decomposition is the unit of host<->device transfer (gather/scatter
// kernels) and of sharding over GPUs; inside a GPU the blocks are fused into the slab so
// that no intra-device halo copies are needed in the subcycle loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evpk {

enum Field : int {
    // grid metrics (time invariant)
    F_DXT, F_DYT, F_DXHY, F_DYHX, F_CXP, F_CYP, F_CXM, F_CYM, F_TINYAREA, F_TAREAR,
    F_TAREA, F_UAREA, F_UAREAR, F_FCOR,
    // per-call inputs
    F_AICE, F_VICE, F_VSNO, F_AICE_INIT, F_STRAIRXT, F_STRAIRYT, F_UOCN, F_VOCN,
    F_SSTLTX, F_SSTLTY, F_CW, F_STRENGTH,
    // work arrays of evp() (ice_dyn_evp.F90:120-128) and derived per-call planes
    F_TMASS, F_UMASS, F_AIU, F_STRAIRX, F_STRAIRY, F_STRTLTX, F_STRTLTY, F_FM,
    F_WATERX, F_WATERY, F_FORCEX, F_FORCEY, F_UMASSDTI, F_VRELC, F_UVEL_INIT, F_VVEL_INIT,
    F_WORK1, F_WORK2, F_ICETM,
    // prognostic state, double buffered (buffer b at F_STATE0 + b*NSTATE)
    F_STATE0,
    F_STATE1 = F_STATE0 + 14,
    // outputs
    F_DIVU = F_STATE1 + 14, F_SHEAR, F_RDGCONV, F_RDGSHEAR, F_PRSSIG, F_STRINTX, F_STRINTY,
    F_STROCNX, F_STROCNY, F_STROCNXT, F_STROCNYT,
    F_COUNT
};
constexpr int NSTATE = 14;           // u, v, stressp_1..4, stressm_1..4, stress12_1..4
constexpr int S_U = 0, S_V = 1, S_SP = 2, S_SM = 6, S_S12 = 10;

constexpr int C0 = 15;               // column offset of the west ghost
constexpr int STRIP_W = 63;          // U columns per wave strip (64 T columns)

// cmask bits
constexpr unsigned char CM_T = 1;    // icetmask == 1
constexpr unsigned char CM_U = 2;    // iceumask

struct Slab {
    double *F;            // F_COUNT planes
    size_t fstride;       // doubles per plane
    int nxl, nyl;         // physical cells
    int pitch;            // doubles per row
    int i0, j0;           // global index of local cell (1,1)
    int nxg, nyg;
    int32_t *tmask, *umask, *iceumask;   // int planes, same indexing
    unsigned char *cmask, *tmphm;        // byte planes, same indexing
};

__host__ __device__ inline size_t cell(const Slab &s, int i, int j) { return (size_t)j * s.pitch + C0 + i; }
__host__ __device__ inline double *plane(const Slab &s, int f) { return s.F + (size_t)f * s.fstride; }

struct DevParams {
    double dt, revp, ecci, denom1, arlx1i, brlx, cosw, sinw, rhow, rhoi, rhos, gravit, a_min, m_min;
    int ndte, tilt_from_slope, wind_on_ugrid;
};

// block descriptors on the device (gather / scatter)
struct BlockDesc {
    int ilo, ihi, jlo, jhi, iglob_lo, jglob_lo;
};

}  // namespace evpk
