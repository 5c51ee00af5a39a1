/*
 * evpk.h -- C ABI of the MI355X EVP sea-ice dynamics solver (libevpk.so).
 *
 * This is the drop-in boundary for ONE path of COSIMA/cice5: the body of
 *     subroutine evp(dt)                      source/ice_dyn_evp.F90:68-510
 * i.e. evp_prep1/evp_prep2 (source/ice_dyn_shared.F90:270,377), the ndte-subcycled
 * stress + stepu loop with its velocity halo update (ice_dyn_evp.F90:336-410,
 * :520-849; ice_dyn_shared.F90:623-748), the tripole stress fold (:416-481),
 * evp_finish (ice_dyn_shared.F90:757) and the T<->U averages around them
 * (source/ice_grid.F90:1799-1958).  The reference has no FFI for this path: the
 * boundary there is a Fortran module procedure plus module-global arrays
 * (SURVEY.md S8b).  The entry points below are what a Fortran `ice_dyn_evp`
 * replacement binds through ISO_C_BINDING (fortran/evpk_mod.F90, fortran/ice_dyn_evp.F90,
 * INTEGRATION.md).
 *
 * Conventions
 *  - plain C, no C++/torch types; every array is a caller-owned host buffer borrowed for
 *    the duration of the call only.
 *  - every field is the reference's block array  real(8) a(nx_block,ny_block,nblocks)
 *    (ice_state.F90:141-147), i fastest, ghost cells included, blocks in local-ID order.
 *    Fortran LOGICAL arrays (tmask, umask, iceumask) are passed as int32 0/1.
 *  - all indices are Fortran 1-based.
 *  - return 0 on success, non-zero on error; evpk_last_error() gives the text.  The
 *    library never aborts the process (the Fortran shim maps non-zero to abort_ice,
 *    mpi/ice_exit.F90:22).
 *  - one host thread per context; a context owns its HIP streams and (nranks > 1) its
 *    RCCL communicator.  There is no CPU fallback: without a usable gfx950 device
 *    evpk_create fails.
 */
#ifndef EVPK_H
#define EVPK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVPK_VERSION 5

/* boundary types: ice_domain.F90 domain_nml ew_boundary_type / ns_boundary_type */
enum { EVPK_BND_CYCLIC = 0, EVPK_BND_OPEN = 1, EVPK_BND_CLOSED = 2, EVPK_BND_TRIPOLE = 3 };

#define EVPK_UNIQUE_ID_BYTES 128

typedef struct evpk_ctx evpk_ctx;

/* Replaces what evp() pulls from ice_blocks / ice_domain / ice_grid:
 *   get_block(blocks_ice(iblk),iblk) -> ilo,ihi,jlo,jhi,i_glob,j_glob  (ice_blocks.F90:22-35, :788)
 *   nblocks, ew/ns boundary types                                       (ice_domain.F90:41-66)
 *   dxt..tinyarea, tarea, uarea, tmask, umask                           (ice_grid.F90:48-77,112-117)
 *   fcor_blk                                                            (ice_dyn_shared.F90:83-84,150)
 * Sharding: ranks own contiguous x-slabs of whole columns (processor_shape slenderX1,
 * ice_distribution.F90:535); rank r's east neighbour is r+1 (cyclic if ew is cyclic). */
typedef struct {
    int32_t nx_global, ny_global;
    int32_t nx_block, ny_block, nblocks;
    int32_t ew_boundary, ns_boundary;
    const int32_t *ilo, *ihi, *jlo, *jhi;      /* [nblocks] */
    const int32_t *iglob_lo, *jglob_lo;        /* [nblocks] this_block%i_glob(ilo), %j_glob(jlo) */
    int32_t rank, nranks;                      /* position in the ring of x-slabs */
    int32_t device;                            /* HIP device ordinal */
    const void *unique_id;                     /* EVPK_UNIQUE_ID_BYTES from evpk_get_unique_id (rank 0), NULL if nranks == 1 */
    const double *dxt, *dyt, *dxhy, *dyhx, *cxp, *cyp, *cxm, *cym;
    const double *tarear, *uarear, *tinyarea, *tarea, *uarea, *fcor;
    const int32_t *tmask, *umask;
    /* optional (NULL allowed): the primary grid lengths HTN, HTE (ice_grid.F90:50-53).  dxt, dyt, dxhy, dyhx, cxp, cyp,
     * cxm, cym are functions of HTN(i,j), HTN(i,j-1), HTE(i,j), HTE(i-1,j) (ice_grid.F90:360-369, :1455, :1533).  If they are
     * given AND reproduce the eight planes above bit for bit on every ocean T cell, the kernels read the two lengths
     * instead of the eight planes (less HBM traffic, identical results); otherwise they are ignored. */
    const double *HTN, *HTE;
} evpk_geom;

/* Replaces the module scalars of ice_dyn_shared.F90:29-81 set by set_evp_parameters
 * (:185-259) and the constants evp reads (ice_constants.F90; a_min, m_min :60-61). */
typedef struct {
    double dt;
    int32_t ndte;
    int32_t revised_evp;
    double revp, ecci, denom1, arlx1i, brlx;
    double cosw, sinw;                         /* AusCOM: namelist variables (ice_dyn_shared.F90:66-72) */
    double rhow, rhoi, rhos, gravit;
    double a_min, m_min;
    int32_t tilt_from_slope;                   /* 1: strtlt = -gravit*umass*ss_tlt (:601-602 / use_ocnslope), 0: geostrophic (:598-599) */
    int32_t wind_on_ugrid;                     /* 1: strairx/y := strax/stray (ice_dyn_evp.F90:226-228), 0: t2ugrid_vector (:240-241) */
    /* ice_strength on the device (used only when evpk_step_in.strength == NULL): the namelist switches of
     * ice_mechred.F90:54-64 (defaults ice_init.F90:273-277: 1, 1, 1, mu_rdg = 3, Cf = 17) and ncat (ice_domain_size) */
    int32_t kstrength, krdg_partic, krdg_redist, ncat;
    double mu_rdg, Cf;
    /* 1: sparse transfers for a host model that keeps the state resident (evpk_upload(in, NULL), page-locked arrays):
     * aice, vice, vsno are uploaded whole, every other input only in the 64x4-cell tiles that hold ice now or held any at
     * the previous call (+ two tiles around), and evpk_download skips tiles that are and were ice-free (they hold the same
     * zeros on both sides).  Identical results provided the T-grid inputs strairxT/yT, aice_init, strength are zero where
     * there is no ice, as CICE leaves them; 0 (default): every cell of every array moves.
     * 2: as 1, and vice, vsno travel in the active tiles too -- only aice is uploaded whole and the tiles are taken from it alone:
     * the host promises vice = vsno = 0 wherever aice = 0 (CICE keeps it so: zap_small_areas, cleanup_itd). */
    int32_t sparse_io;
    int32_t reserved_;
} evpk_params;

/* Per-call inputs: what evp(dt) reads from ice_state / ice_flux / ice_atmo
 * (ice_dyn_evp.F90:70-90).  `strength` is the result of ice_strength (:291-301),
 * physical cells; the library does its halo update (:311).  NULL is allowed for
 * ss_tltx/ss_tlty when tilt_from_slope == 0 and for strax/stray when wind_on_ugrid == 0.
 * strength == NULL: the library evaluates ice_strength itself (ice_mechred.F90:2111-2269) where the reference calls it,
 * from aice, vice and -- for kstrength == 1 -- the thickness distribution aicen, vicen (nx_block, ny_block, ncat,
 * nblocks) and aice0 (ice_state.F90); the host then needs neither evp_prep1 nor the icetmask halo before the call.
 * Its exp() is a fixed < 1 ulp algorithm, so the result can differ from a host intrinsic in the last bit. */
typedef struct {
    const double *aice, *vice, *vsno, *aice_init;
    const double *strairxT, *strairyT, *strax, *stray;
    const double *uocn, *vocn, *ss_tltx, *ss_tlty, *Cdn_ocn;
    const double *strength;
    const double *aicen, *vicen, *aice0;       /* read only when strength == NULL and kstrength == 1 */
} evpk_step_in;

/* In/out prognostic state and outputs: the arrays evp(dt) leaves modified.
 * sigma order: stressp[0..3] = stressp_1..4 etc. (ice_flux.F90:93-97). Any output
 * pointer may be NULL (skipped).  uvel/vvel are written on all cells (ghosts are
 * halo-updated, ice_dyn_evp.F90:392-407); sigma on physical cells and the N/E ghost
 * T-cells the reference computes (ice_dyn_shared.F90:528-537); the rest on physical cells. */
typedef struct {
    double *uvel, *vvel;
    double *stressp[4], *stressm[4], *stress12[4];
    int32_t *iceumask;
    double *divu, *shear, *rdg_conv, *rdg_shear, *prs_sig;
    double *strintx, *strinty, *strocnx, *strocny, *strocnxT, *strocnyT;
    double *strairx, *strairy, *strtltx, *strtlty, *fm;
    double *tmass;                             /* AusCOM: sicemass = tmass (ice_dyn_evp.F90:205-207) */
    double *aiu, *umass, *uvel_init, *vvel_init;
    int32_t *icetmask;
    double *strength;                          /* out, all cells: the strength with its ghost cells halo-updated, as evp leaves it
                                                  (ice_dyn_evp.F90:311-312); may be the array evpk_step_in.strength points to */
} evpk_state;

typedef struct {
    int64_t icellt;          /* active T cells on physical cells of this rank (sum of icellt minus ghost duplicates) */
    int64_t icellu;          /* active U cells of this rank */
    int64_t ncell_slab;      /* physical cells of this rank */
    int32_t nstrips;         /* wave strips launched per subcycle (active) */
    int32_t nstrips_total;
    int32_t subcycles_done;  /* since the last evpk_prep */
    float loop_ms;           /* HIP-event time of the last evpk_subcycle call on the compute stream */
    float kernel_ms;         /* time of the one-subcycle kernel launches (k_subcycle) of that call: HIP events around runs of six
                                consecutive launches (launches 3..8 of every 20 by default; EVPK_TIME_KERNELS=2: every launch, 0: none),
                                span time / launches in the spans x number of launches -- never more than loop_ms */
    int32_t kernel_launches; /* ... and their number */
    float kernel2_ms;        /* the same for the two-subcycle kernel (k_subcycle2) */
    int32_t kernel2_launches;
    int32_t strip_rows, strip_rows2;   /* strip heights in use: k_subcycle, k_subcycle2 (auto-tuned) */
    int32_t nstrips2;
    int32_t zone_cols;                 /* x-slabs: ghost-zone width per side (2 x launches per exchange); 0 = no ghost zones */
    int32_t zone_exchanges;            /* ghost-zone exchanges with the slab neighbours in the last evpk_subcycle call */
    int64_t zone_bytes;                /* bytes this rank sent in them */
    int32_t overlap_split;             /* x-slabs: 1 the exchange launches are split into edge + interior strips on two streams, 0 whole
                                          launches, -1 still trying both (first three evps; EVPK_OVERLAP fixes it) */
    int32_t tile_kernel;               /* 1: the pairs of the last call ran the small-slab tile variant (k_subcycle2t), 0: the marching one */
    int32_t kernel_timed, kernel2_timed;   /* launches of each kind that were actually bracketed by HIP events (kernel_ms / kernel2_ms are
                                          their mean x the number of launches) */
    float bound_ms;                    /* halo / fold / ghost-zone work of the last evpk_subcycle call (pack, transport, unpack): mean of the
                                          HIP-event-timed updates x their number -- what the reference books under timer_bound
                                          (ice_dyn_evp.F90:392-400) */
    int32_t bound_updates;             /* ... and their number */
    int32_t compact_metrics;           /* 1: the kernels rebuild the eight metric planes from HTN / HTE (evpk_geom) */
    int32_t transport;                 /* EVPK_XP_*: what carries the exchanges between ranks */
    int32_t band_row_exchanges;        /* x-slab ranks on a tripole grid: refreshes of the mirror slab in the last evpk_subcycle call -- one message
                                          per rank that owns columns of it, ghost zones included (the mirror rank and its neighbours with equal slabs: up to three ranks), posted
                                          together with the ghost-zone exchange; the fold itself runs inside the pair launches (band_pair) */
    float kernel3_ms;                  /* the same as kernel2_ms for the three-subcycle pipeline kernel (k_subcycle3w) */
    int32_t kernel3_launches, kernel3_timed;
    int32_t strip_rows3, nstrips3;     /* its strip height and active strips */
    int32_t rccl_ranks;                /* ncclCommCount of the context's communicator (0: no RCCL communicator) */
    int32_t device;                    /* hipGetDevice ordinal the context runs on */
    int32_t device_pci;                /* (PCI domain << 16) | (bus << 8) | device of that GPU: distinct per physical device */
    int64_t delivery_checked;          /* EVPK_VERIFY_DELIVERY: values written in place into page-locked caller arrays that were delivered a second
                                          time through the staged path and compared (since evpk_create) ... */
    int64_t delivery_bad;              /* ... and how many of them differed (mode 1: the first one fails the download; mode 2: repaired) */
} evpk_stats;

enum { EVPK_XP_NONE = 0, EVPK_XP_RCCL = 1, EVPK_XP_SHM_RELAY = 2, EVPK_XP_IPC = 3, EVPK_XP_SELF = 4 };

/* rank 0 creates the RCCL id; the host model broadcasts the bytes (MPI_Bcast in CICE,
 * torch.distributed in bench.py) and every rank passes them to evpk_create.
 * Second transport, for ranks that cannot form an RCCL communicator (several ranks on one GPU: tests; or a node whose
 * RCCL bootstrap fails: bench.py's fallback): a unique id that starts with "EVPKSHM:<name>" selects a host-staged relay
 * through the POSIX shared-memory segment /<name> with the same point-to-point / all-gather semantics (correct, slow). */
int evpk_get_unique_id(void *id /* EVPK_UNIQUE_ID_BYTES */);

int evpk_create(const evpk_geom *g, evpk_ctx **out);
/* Two-phase start for nranks > 1: evpk_create with unique_id == NULL allocates and uploads everything but touches no other
 * rank; the host then agrees across ranks (MPI_Allreduce in CICE, gloo in bench.py) that every create succeeded and only
 * then calls evpk_connect, which is collective (communicator / peer mapping, slab starts).  A rank that failed early can
 * therefore not leave the others hanging inside the communicator bootstrap.  evpk_create with a unique_id does both. */
int evpk_connect(evpk_ctx *c, const void *unique_id /* EVPK_UNIQUE_ID_BYTES */);
/* 0 if `device` exists and is a gfx950 part this library can run on (no context needed; error text via evpk_last_error(NULL)) */
int evpk_device_check(int32_t device);
int evpk_set_params(evpk_ctx *c, const evpk_params *p);

/* whole evp(dt): upload + prep + ndte subcycles + finish + download */
int evpk_run(evpk_ctx *c, const evpk_step_in *in, evpk_state *st);

/* the same in stages (bench.py times prep..finish with the data resident in HBM) */
int evpk_upload(evpk_ctx *c, const evpk_step_in *in, const evpk_state *st);  /* st == NULL: inputs only, the state stays resident on the device */
int evpk_prep(evpk_ctx *c);
int evpk_subcycle(evpk_ctx *c, int32_t nsub);   /* advances ksub; the last-subcycle diagnostics fire at ksub == ndte */
int evpk_finish(evpk_ctx *c);
int evpk_download(evpk_ctx *c, evpk_state *st);
int evpk_sync(evpk_ctx *c);

/* On one rank (and on x-slab ranks of a tripole grid) evpk_prep compacts the strip list on the device and returns without
 * waiting for icellt / icellu / nstrips2; evpk_subcycle fills them in when its loop has run.  evpk_get_stats called in between
 * synchronises the compute stream once and returns this evp's counts (never stale ones); it also reports a transport error that
 * an exchange of evpk_prep raised on that path (otherwise reported by evpk_subcycle). */
int evpk_get_stats(evpk_ctx *c, evpk_stats *s);

/* principal_stress (ice_dyn_shared.F90:853-893; called by ice_history for sig1/sig2): normalised principal
 * stresses from the sigma_1 planes and prs_sig resident on the device after evpk_finish / evpk_run.
 * Physical cells of sig1, sig2 (block arrays) are written. */
int evpk_principal_stress(evpk_ctx *c, double *sig1, double *sig2);

/* Row a3 as an entry point of its own: ice_HaloUpdate (mpi/ice_boundary.F90:1331 2DR8, :2451 3DR8) of a block array on the
 * device -- upload, the halo / fold kernels and transports the evp path uses (N-S fill or tripole u-fold, E-W ring between the
 * slab ranks), download.  Collective over the ranks of the context.
 *   a          real(8) (nx_block, ny_block, nblocks) for nz == 0, (nx_block, ny_block, nz, nblocks) for nz >= 1    in / out
 *   field_loc  1 centre, 2 NE corner, 3 N face, 4 E face; field_type 1 scalar, 2 vector, 3 angle (ice_constants.F90:198-220)
 *   fill       the reference's optional fillValue (0 when absent)
 * Written, as by the reference's production (MPI) backend: every ghost cell with a neighbour (its value; `fill` if the
 * neighbour is an eliminated land block), the tripole top row for NE-corner / N-face fields, and the outermost row / column
 * of the block array where nothing else writes (mpi/ice_boundary.F90:1409-1416).  Ghost cells beyond an open / closed
 * boundary that are not on the array's edge (short blocks) and padding cells keep the caller's values.
 * evpk_halo_update_stress: ice_HaloUpdate_stress(array1, array2, halo, field_loc_center, field_type_scalar)
 * (mpi/ice_boundary.F90; ice_dyn_evp.F90:416-481): the tripole north ghost row of a1 from the top row of a2; a ghost cell
 * next to an eliminated land block gets 0; nothing else is touched.
 * Pinned by tests/golden/ref_*.npz (outputs of the reference's own routines). */
/* Both stage through the scratch state planes that evpk_prep primes for the subcycle loop on some paths: call them before
 * evpk_prep or after evpk_finish / evpk_run, never between evpk_prep and the end of the subcycle loop (the reference has no halo
 * update of a foreign field inside evp's loop either, ice_dyn_evp.F90:345-409). */
int evpk_halo_update(evpk_ctx *c, double *a, int32_t nz, int32_t field_loc, int32_t field_type, double fill);
int evpk_halo_update_stress(evpk_ctx *c, double *a1, const double *a2);

/* SURVEY S8 row f-3, first step: transport_upwind (source/ice_transport_driver.F90:634-772) on the velocities the last evp
 * left on the device.  The cell-edge velocities uee, vnn (:688-701) and their halo updates (E face / N face vectors,
 * :703-708) and upwind_field (:1614-1689) run on the GPU; `works` is the work array state_to_work (:1382-1513) fills on the
 * host, real(8) (nx_block, ny_block, narr, nblocks), advected in place on physical cells (ghost cells must be current on
 * entry -- bound_state -- and are left alone; the reference calls bound_state afterwards, :763).  work_to_state /
 * compute_tracers stay with the host's tracer bookkeeping.  Needs HTN and HTE in evpk_geom. */
int evpk_transport_upwind(evpk_ctx *c, double dt, int32_t narr, double *works);

/* transport_upwind WHOLE (ice_transport_driver.F90:634-772): state_to_work (:1382-1513) inside the gather, upwind_field on the
 * device, work_to_state (:1520-1609) with compute_tracers (ice_itd.F90:1359-1501) and bound_state (ice_state.F90:173-238) inside
 * the scatter -- the caller hands over its state arrays as they are:
 *   aice0 (nx_block, ny_block, max_blocks), aicen, vicen, vsnon (nx_block, ny_block, ncat, max_blocks),
 *   trcrn (nx_block, ny_block, ntrcr_dim, ncat, max_blocks) of which tracers 1 .. ntrcr are in use
 *   trcr_depend (ntrcr): 0 area, 1 ice volume, 2 snow volume, 2 + nt: tracer nt (ice_state.F90); nt_Tsfc, nt_alvl, nt_apnd,
 *   nt_fbri: 1-based tracer indices, 0 = not in use; tr_pond_cesm / lvl / topo: the pond scheme flags; Tocnfrz (ice_constants)
 * In / out, every cell of every block: physical cells advected, ghost cells their neighbours' new values (bound_state; aice0 has
 * no halo update in the reference and keeps its ghost cells).  Ghost cells must be current on entry.  Up to 32 tracers. */
int evpk_transport_upwind_state(evpk_ctx *c, double dt, int32_t ncat, int32_t ntrcr, int32_t ntrcr_dim, const int32_t *trcr_depend,
                                int32_t nt_Tsfc, int32_t nt_alvl, int32_t nt_apnd, int32_t nt_fbri, int32_t tr_pond_cesm,
                                int32_t tr_pond_lvl, int32_t tr_pond_topo, double Tocnfrz, double *aice0, double *aicen, double *vicen,
                                double *vsnon, double *trcrn);

/* SURVEY S8 row f-3, second step: horizontal_remap (source/ice_transport_remap.F90:309-850), the incremental remapping of
 * transport_remap (ice_transport_driver.F90:258-626), on the velocities the last evp left on the device.
 *
 * evpk_remap_init (once, after evpk_create / evpk_connect): the grid arrays this path reads beyond evpk_geom's -- dxu, dyu
 * (ice_grid.F90) and hm (the land mask as a real, ice_grid.F90) -- block arrays, ghost cells current.
 *
 * evpk_transport_remap replaces the call at ice_transport_driver.F90:513-517 with the same arguments minus uvel, vvel:
 *   mm  real(8) (nx_block, ny_block, 0:ncat, max_blocks): mean mass (aim; category 0 = open water)      in / out
 *   tm  real(8) (nx_block, ny_block, ntrace, ncat, max_blocks): mean tracers (trm); NULL if ntrace = 0   in / out
 *   tracer_type, depend (1-based, 0 = none), has_dependents: (ntrace), as init_transport sets them (:117-187); a tracer
 *   follows the one it depends on
 *   integral_order 1..3, l_dp_midpt: the module parameters of ice_transport_remap.F90:252-262
 *   l_fixed_area: must be 0 (.false., the reference's setting for the B grid, :489-492)
 * Physical cells of mm and tm are advanced in place; ghost cells are read (they are refreshed on the device first, so they
 * need not be current) and left alone -- the reference calls bound_state afterwards (:573).  Up to EVPK_REMAP_MAX_TRACERS
 * tracers.  make_masks, construct_fields, limited_gradient, departure_points, locate_triangles, triangle_coordinates,
 * transport_integrals and update_fields run on the GPU in the Fortran's operation order; state_to_tracers /
 * tracers_to_state and the conservation / monotonicity diagnostics stay with the host's tracer bookkeeping.
 * Returns 0, EVPK_REMAP_BAD_DEPARTURE (a departure point left the neighbouring cells, :1583-1607: the time step is too
 * long; mm, tm untouched), EVPK_REMAP_NEGATIVE_MASS (:3622-3640; mm, tm undefined: the update writes them as it goes) -- the
 * reference aborts the run in both cases -- or 1 with evpk_last_error.  Needs HTN and HTE in evpk_geom. */
#define EVPK_REMAP_MAX_TRACERS 32
#define EVPK_REMAP_BAD_DEPARTURE 11
#define EVPK_REMAP_NEGATIVE_MASS 12
int evpk_remap_init(evpk_ctx *c, const double *dxu, const double *dyu, const double *hm);
int evpk_transport_remap(evpk_ctx *c, double dt, int32_t ncat, int32_t ntrace, double *mm, double *tm, const int32_t *tracer_type,
                         const int32_t *depend, const int32_t *has_dependents, int32_t integral_order, int32_t l_dp_midpt,
                         int32_t l_fixed_area);

/* The same with the state transforms of transport_remap (ice_transport_driver.F90:198-627) on the device too: state_to_tracers
 * (:789-900) inside the gather, tracers_to_state (:908-1003) and bound_state (ice_state.F90:173-238) inside the scatter, so that
 * the caller hands over its state arrays as they are (the optional conservation / monotonicity checks of the reference are
 * compile-time .false., :255-257, and are not reproduced):
 *   aice0 (nx_block, ny_block, max_blocks), aicen, vicen, vsnon (nx_block, ny_block, ncat, max_blocks),
 *   trcrn (nx_block, ny_block, ntrcr_dim, ncat, max_blocks) of which tracers 1 .. ntrcr are in use (ice_state.F90: max_ntrcr / ntrcr)
 *   nt_qsno (1-based), nslyr: the snow enthalpy tracers, advected as trcrn + rhos*Lfresh (:866, :988); rhos_lfresh = that constant
 *   tracer_type, depend, has_dependents: (2 + ntrcr), as init_transport sets them for hice, hsno and the ntrcr tracers
 * In / out, every cell of every block: physical cells whose new area is > 0 are rewritten (the others keep their values, as
 * tracers_to_state leaves them), ghost cells take their neighbours' new values (bound_state).  Ghost cells must be current on
 * entry.  Return values as evpk_transport_remap. */
int evpk_transport_remap_state(evpk_ctx *c, double dt, int32_t ncat, int32_t ntrcr, int32_t ntrcr_dim, int32_t nt_qsno, int32_t nslyr,
                               double rhos_lfresh, double *aice0, double *aicen, double *vicen, double *vsnon, double *trcrn,
                               const int32_t *tracer_type, const int32_t *depend, const int32_t *has_dependents,
                               int32_t integral_order, int32_t l_dp_midpt);

/* the elastic-anisotropic-plastic rheology, eap(dt) (source/ice_dyn_eap.F90:66-486; kdyn = 2,
 * ice_step_mod.F90:1118).  eap is evp with another stress: evp_prep1/2, stepu, the velocity halo, evp_finish are shared
 * (:79-80), stress_eap (:1052-1467) with update_stress_rdg (:1474-1658) takes the place of stress, stepa (:1664-1787, with
 * calc_ffrac :1795-1864) evolves the structure tensor every tenth subcycle, and there is no stress fold at the end.
 *
 * evpk_eap_init switches the context to EAP: every later evpk_run / evpk_subcycle runs the eap loop.  It takes the six
 * lookup tables init_eap builds on the host (:555-619; module arrays s11r ... s22s(nx_yield, ny_yield, na_yield), :36-39) and
 * sets the structure tensor to isotropic (a11 = 1/2, a12 = 0, :529-551) as init_eap does.
 * evpk_eap_upload moves a11_1..4, a12_1..4 (the prognostic EAP state: restart, read_restart_eap :1908-2010) to the device
 * (members that are NULL stay as they are); evpk_eap_download brings back any non-NULL member: physical cells and the N / E
 * ghost T cells, as the reference computes them.
 * sin, cos and atan2 inside update_stress_rdg / calc_ffrac are fixed algorithms (csrc/evpk_fmath.h, within 1-2 ulp of libm):
 * the table indices depend on the last bit of an angle, so results agree with another math library except where an index
 * falls on the other side of a table cell -- as between any two compilers of the reference. */
typedef struct {
    double *a11_c[4], *a12_c[4];      /* a11_1..4, a12_1..4: structure tensor at the corners ne, nw, sw, se (ice_dyn_eap.F90:40-41) */
    double *a11, *a12;                /* out: cell means (history, :55-56) */
    double *e11, *e12, *e22;          /* out: strain rate tensor (:47-49) */
    double *yieldstress11, *yieldstress12, *yieldstress22;      /* out (:50-52) */
    double *s11, *s12, *s22;          /* out: stress tensor (:53-55) */
} evpk_eap_state;
int evpk_eap_init(evpk_ctx *c, int32_t nx_yield, int32_t ny_yield, int32_t na_yield, const double *s11r, const double *s12r,
                  const double *s22r, const double *s11s, const double *s12s, const double *s22s);
int evpk_eap_upload(evpk_ctx *c, const evpk_eap_state *st);
int evpk_eap_download(evpk_ctx *c, evpk_eap_state *st);

/* The dynamics records of the reference's binary restart (source/ice_restart_driver.F90:122-176 dumpfile, :295-412
 * restartfile; io_binary/ice_restart.F90:641-684): uvel, vvel, strocnxT, strocnyT, stressp_1,3,2,4, stressm_1,3,2,4,
 * stress12_1,3,2,4, iceumask as real 0/1 -- 17 Fortran sequential unformatted records of the (nx_global, ny_global)
 * real*8 array, big-endian under the production flags.  Written from / read into the state resident on the device (one
 * rank; a multi-rank host gathers through its own path as the reference does).  evpk_restart_read fills the ghost
 * cells as restartfile does (halo by field location and type, then the twelve ice_HaloUpdate_stress pairings on
 * tripole grids, :370-395) and leaves the context as after evpk_upload: evpk_upload(in, NULL) may follow.
 * byte_offset: where the first of the 17 records starts in the file (they are consecutive in this library's own files;
 * in a full CICE restart the radiation fields sit between vvel and strocnxT, so that file goes through the host). */
int evpk_restart_write(evpk_ctx *c, const char *path, int32_t append, int32_t big_endian);
int evpk_restart_read(evpk_ctx *c, const char *path, int64_t byte_offset, int32_t big_endian);

/* Profiling aid: nrep device-to-device copies of one scratch pair plane with the hot kernel's access
 * shape (16 B per lane, coalesced).  Each moves exactly (nxl+2)*(nyl+2)*16 bytes each way: a known
 * byte count in the PMC trace to calibrate FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM). */
/* Each call also runs nrep copies of EVPK_CALIB_BIG_BYTES from one buffer to another (kernel k_calib_copy_big): both far
 * larger than the 256 MiB Infinity Cache, the byte count the FETCH_SIZE correction is checked against. */
#define EVPK_CALIB_BIG_BYTES (1ull << 30)
int evpk_calibrate(evpk_ctx *c, int32_t nrep);
int evpk_destroy(evpk_ctx *c);

/* Optional: page-lock a host array the caller keeps for the life of the run (CICE's module arrays of ice_state /
 * ice_flux / ice_grid, ice_state.F90 / ice_flux.F90) and map it into the device address space.  evpk_upload /
 * evpk_download / evpk_run then read and write such arrays in place over PCIe instead of staging them, and a download
 * touches only the cells it delivers.  Arrays that were not registered keep working (staged copies).  Unpin before the
 * memory is freed.  Returns 0 on success; non-zero if the range overlaps a live registration or the runtime refuses.
 * The library moves an array in place ONLY if all of it lies inside a range registered here (and not yet released) or
 * handed out by evpk_host_alloc -- it keeps its own table and never trusts a registration it did not make.  The pages
 * of a registered range are advised MADV_NOHUGEPAGE and mlock'ed (best effort; EVPK_PIN_HARDEN=0 skips both):
 * hipHostRegister mirrors the caller's pages through MMU notifiers, it does not hard-pin them.
 * evpk_host_alloc / evpk_host_free: page-locked memory allocated and pinned BY THE DRIVER (hipHostMalloc), mapped into
 * the device address space -- for a host that can choose where its arrays live (allocatable arrays, c_f_pointer).
 * evpk_host_is_mapped: 1 if [ptr, ptr + bytes) would be moved in place.
 * EVPK_VERIFY_DELIVERY=1 (diagnostic): every plane a download writes in place is delivered a second time through the
 * staged path and compared on the host; a difference fails the call with plane, block, cell and page in
 * evpk_last_error (=2: the caller's array is repaired and the event counted, evpk_stats.delivery_bad). */
/* Arrays that already live in DEVICE memory (a host model whose fields are resident on the GPU: OpenMP target /
 * OpenACC use_device pointers, hipMalloc) may be passed wherever a host array is expected: the library detects them
 * (hipPointerGetAttributes) and gathers / scatters them in place, no PCIe transfer at all. */
int evpk_pin_host(void *ptr, size_t bytes);
int evpk_unpin_host(void *ptr);
int evpk_host_alloc(size_t bytes, void **out);
int evpk_host_free(void *ptr);
int evpk_host_is_mapped(const void *ptr, size_t bytes);
/* 1 if the library was built with -DEVPK_EXPERIMENTAL (it then contains k_subcycle2 and k_subcycle3w -- measured, not adopted -- and
 * honours EVPK_PREFETCH=0 / EVPK_TRIPLE=1; the product build refuses them) */
int evpk_experimental_built(void);
const char *evpk_last_error(const evpk_ctx *c);  /* c may be NULL: error of the last failed evpk_create */

/* Host-only description of this rank's halo exchange (no GPU needed): neighbours in the
 * slab ring and the fold partner layout.  Used by the world_size-2 CPU tests.
 *   out[0] = west rank (-1 none), out[1] = east rank (-1 none),
 *   out[2] = first global column i0, out[3] = last global column i1,
 *   out[4] = columns per rank used for the tripole all-gather (max slab width) */
int evpk_slab_layout(int32_t nx_global, int32_t nranks, int32_t rank, int32_t ew_boundary,
                     int32_t i0, int32_t i1, int32_t out[5]);

#ifdef __cplusplus
}
#endif
#endif
