/*
 * oracle/evp_oracle.h -- CPU restatement of the CICE5 EVP dynamics path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (cice5_amd/, include/,
 * fortran/) includes, links or calls this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * PARITY: pinned by the reference's OWN output for the halo updates (orc_halo_r8 / i4 / stress), ice_strength and the
 * block / distribution arithmetic -- those reference routines compile here unmodified (oracle/ref/Makefile) and their
 * outputs are the fixtures tests/golden/ref_*.npz (tests/test_ref_pins.py).  PARITY UNPINNED for stress, stepu, evp_prep1/2,
 * evp_finish, to_ugrid / to_tgrid, horizontal_remap and eap: their Fortran modules `use ice_grid`, which needs the netCDF
 * Fortran module (source/ice_grid.F90:144), absent from the image, and the reference ships no tests or golden vectors
 * (SURVEY.md S4).  For those this restatement is checked by decomposition invariance and analytic properties
 * (tests/test_oracle.py), and tests/golden/evp_*.npz hold ITS OWN outputs (regression pins, not reference vectors).
 *
 * Array convention: every field is a Fortran-ordered block array
 *   a(nx_block, ny_block, nblocks),  i fastest  (source/ice_state.F90:141-147)
 * Fortran LOGICAL arrays (tmask, umask, iceumask) are int32 0/1 here.
 * All indices in orc_geom are Fortran 1-based.
 */
#ifndef EVP_ORACLE_H
#define EVP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BND_CYCLIC = 0, ORC_BND_OPEN = 1, ORC_BND_CLOSED = 2, ORC_BND_TRIPOLE = 3 };
enum { ORC_LOC_CENTER = 1, ORC_LOC_NECORNER = 2, ORC_LOC_NFACE = 3, ORC_LOC_EFACE = 4 };   /* ice_constants.F90: field_loc_* */
enum { ORC_KIND_SCALAR = 1, ORC_KIND_VECTOR = 2 };        /* ice_constants.F90: field_type_* */

/* block decomposition as seen by one process (source/ice_blocks.F90:22-35) */
typedef struct {
    int32_t nx_global, ny_global;
    int32_t nx_block, ny_block, nblocks;
    int32_t ew_boundary, ns_boundary;
    const int32_t *ilo, *ihi, *jlo, *jhi;   /* [nblocks] physical-domain bounds inside the block */
    const int32_t *iglob_lo, *jglob_lo;     /* [nblocks] global (i,j) of cell (ilo,jlo) == this_block%i_glob(ilo), j_glob(jlo) */
} orc_geom;

/* scalars of source/ice_dyn_shared.F90:29-81 plus the constants evp uses */
typedef struct {
    double dt;
    int32_t ndte;
    int32_t revised_evp;
    double revp, ecci, dtei, dte2T, denom1, arlx1i, brlx;
    double cosw, sinw, dragio, rhow, rhoi, rhos, gravit;
    double a_min, m_min;
    int32_t tilt_from_slope;   /* 1: strtlt = -gravit*umass*ss_tlt (coupled / use_ocnslope), 0: geostrophic */
    int32_t wind_on_ugrid;     /* 1: strairx/y := strax/stray (ACCESS, calc_strair=F), skip t2ugrid_vector */
    /* ice_strength (source/ice_mechred.F90:2111-2269), SURVEY.md S8 row a10 / f-1 */
    int32_t strength_mode;     /* 0: f->strength is an input (computed by the caller); 1: evp calls ice_strength as the reference does (:291-301) */
    int32_t kstrength;         /* 1 Rothrock (1975), else Hibler (1979)            ice_mechred.F90:55-56, ice_init.F90:273 */
    int32_t krdg_partic;       /* 0 Thorndike et al. (1975), 1 exponential          :57-58 */
    int32_t krdg_redist;       /* 0 Hibler (1980), 1 exponential                    :59-60 */
    int32_t ncat;              /* thickness categories (ice_domain_size) */
    int32_t pad_;
    double mu_rdg, Cf;         /* :63-64; namelist defaults 3, 17 (ice_init.F90:276-277) */
} orc_params;

/* module-global state evp(dt) reads and writes */
typedef struct {
    /* grid (ice_grid.F90:48-77,112-117) */
    const double *dxt, *dyt, *dxhy, *dyhx, *cxp, *cyp, *cxm, *cym;
    const double *tarear, *uarear, *tinyarea, *tarea, *uarea, *fcor;
    const int32_t *tmask, *umask;
    /* inputs */
    const double *aice, *vice, *vsno, *aice_init;
    const double *strairxT, *strairyT, *strax, *stray;
    const double *uocn, *vocn, *ss_tltx, *ss_tlty, *Cdn_ocn;
    /* ice thickness distribution, only read when strength_mode = 1 and kstrength = 1:
     * aicen, vicen (nx_block, ny_block, ncat, nblocks), aice0 (nx_block, ny_block, nblocks)   ice_state.F90 */
    const double *aicen, *vicen, *aice0;
    /* in/out */
    double *strength;            /* in (already computed) when strength_mode = 0, else out */
    double *uvel, *vvel;
    double *stressp[4], *stressm[4], *stress12[4];
    int32_t *iceumask;
    /* out */
    double *divu, *shear, *rdg_conv, *rdg_shear, *prs_sig;
    double *strintx, *strinty, *strocnx, *strocny, *strocnxT, *strocnyT;
    double *strairx, *strairy, *strtltx, *strtlty, *fm, *tmass;
    double *aiu, *umass, *uvel_init, *vvel_init;
    int32_t *icetmask;
} orc_fields;

/* EAP (kdyn = 2, source/ice_dyn_eap.F90): lookup tables of init_eap (:555-619), the structure tensor (restart state) and the
 * history fields.  Tables: Fortran s11r(nx_yield, ny_yield, na_yield), i.e. C order [na][ny][nx]. */
typedef struct {
    int32_t nx_yield, ny_yield, na_yield, pad_;
    const double *s11r, *s12r, *s22r, *s11s, *s12s, *s22s;
    double *a11[4], *a12[4];             /* in/out: a11_1..4, a12_1..4 (ne, nw, sw, se) */
    double *a11m, *a12m;                 /* out: a11, a12 (cell means) */
    double *e11, *e12, *e22, *yieldstress11, *yieldstress12, *yieldstress22, *s11, *s12, *s22;   /* out */
} orc_eap_state;

void orc_set_evp_parameters(double dt, int32_t ndte, int32_t revised_evp, double xmin, orc_params *p);

void orc_evp_prep1(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                   const double *aice, const double *vice, const double *vsno, const int32_t *tmask,
                   const double *strairxT, const double *strairyT,
                   double *strairx, double *strairy, double *tmass, int32_t *icetmask,
                   const orc_params *p);

void orc_to_ugrid_blk(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                      const double *work1, const double *tarea, const double *uarea, double *work2);
void orc_to_tgrid_blk(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                      const double *work1, const double *tarea, const double *uarea, double *work2);

void orc_evp_prep2(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                   int32_t *icellt, int32_t *icellu,
                   int32_t *indxti, int32_t *indxtj, int32_t *indxui, int32_t *indxuj,
                   const double *aiu, const double *umass, double *umassdti, const double *fcor,
                   const int32_t *umask, const double *uocn, const double *vocn,
                   const double *strairx, const double *strairy,
                   const double *ss_tltx, const double *ss_tlty,
                   const int32_t *icetmask, int32_t *iceumask, double *fm, double dt,
                   double *strtltx, double *strtlty, double *strocnx, double *strocny,
                   double *strintx, double *strinty, double *waterx, double *watery,
                   double *forcex, double *forcey,
                   double *const stressp[4], double *const stressm[4], double *const stress12[4],
                   double *uvel_init, double *vvel_init, double *uvel, double *vvel,
                   const orc_params *p);

void orc_stress(int nx, int ny, int ksub, int ndte, int icellt,
                const int32_t *indxti, const int32_t *indxtj,
                const double *uvel, const double *vvel,
                const double *dxt, const double *dyt, const double *dxhy, const double *dyhx,
                const double *cxp, const double *cyp, const double *cxm, const double *cym,
                const double *tarear, const double *tinyarea, const double *strength,
                double *const stressp[4], double *const stressm[4], double *const stress12[4],
                double *shear, double *divu, double *prs_sig, double *rdg_conv, double *rdg_shear,
                double *str /* [8][ny][nx] */, const orc_params *p);

void orc_stepu(int nx, int ny, int icellu, const double *Cw,
               const int32_t *indxui, const int32_t *indxuj,
               const double *aiu, const double *str,
               const double *uocn, const double *vocn, const double *waterx, const double *watery,
               const double *forcex, const double *forcey, const double *umassdti, const double *fm,
               const double *uarear, double *strocnx, double *strocny, double *strintx, double *strinty,
               const double *uvel_init, const double *vvel_init, double *uvel, double *vvel,
               const orc_params *p);

void orc_evp_finish(int nx, int ny, int icellu, const double *Cw,
                    const int32_t *indxui, const int32_t *indxuj,
                    const double *uvel, const double *vvel, const double *uocn, const double *vocn,
                    const double *aiu, const double *fm,
                    double *strocnx, double *strocny, double *strocnxT, double *strocnyT,
                    const orc_params *p);

void orc_principal_stress(int nx, int ny, const double *stressp_1, const double *stressm_1,
                          const double *stress12_1, const double *prs_sig, double *sig1, double *sig2);

/* exp() of this restatement: the reference calls the Fortran intrinsic, whose last bit is implementation-defined; the
 * oracle and the HIP kernels both evaluate the classical Cody-Waite reduction + degree-5 minimax in r^2 (the fdlibm
 * scheme, < 1 ulp) in plain un-fused fp64 so that they agree bit for bit.  |x| < 700. */
double orc_exp(double x);
void orc_set_num_threads(int n);

/* ice_strength (ice_mechred.F90:2111-2269) with asum_ridging (:758-812, unused by the strength) and ridge_itd (:936-1285)
 * on one block; aicen / vicen are (ncat, ny, nx) planes of that block */
void orc_ice_strength(int nx, int ny, int ilo, int ihi, int jlo, int jhi, int icells,
                      const int32_t *indxi, const int32_t *indxj,
                      const double *aice, const double *vice, const double *aice0,
                      const double *aicen, const double *vicen, double *strength, const orc_params *p);

void orc_strength_hibler(int nx, int ny, int ilo, int ihi, int jlo, int jhi,
                         const double *aice, const double *vice, double *strength);

/* halo updates, MPI-backend semantics (ghost cells pre-filled with `fill`) */
typedef void (*orc_halo_cb)(double *a, int loc, int kind, double fill, int phase, void *user);
void orc_set_halo_callback(orc_halo_cb cb, void *user);   /* multi-process CPU tests only */
void orc_halo_r8(const orc_geom *g, double *a, int loc, int kind, double fill);
void orc_halo_i4(const orc_geom *g, int32_t *a, int32_t fill);
void orc_halo_stress(const orc_geom *g, double *a1, const double *a2);

/* whole evp(dt): returns total active (icellt, icellu) over blocks through counts[2].
   nsub_override > 0 runs that many subcycles instead of ndte (the "last subcycle"
   diagnostics still fire on ksub == ndte only, as in the reference). */
/* transport_upwind (source/ice_transport_driver.F90:634-772) without its tracer bookkeeping: the edge velocities
 * uee = p5*(uvel(i,j)+uvel(i,j-1)), vnn = p5*(vvel(i,j)+vvel(i-1,j)) (:697-698) with their halo updates (E face / N face
 * vectors, :703-708), then upwind_field (:1614-1689) on each of the `narr` arrays of `works` (nx_block, ny_block, narr,
 * nblocks) -- what state_to_work (:1382) hands it -- in place on physical cells.  The ghost cells of `works` must be
 * current on entry (bound_state) and are not updated (the reference calls bound_state afterwards, :763). */
/* the same with the state transforms around it: state_to_work (:1382-1513), work_to_state (:1520-1609) with compute_tracers
 * (ice_itd.F90:1359-1501), bound_state; see evp_oracle.c */
void orc_compute_tracers(int nx, int ny, int ntrcr, const int32_t *trcr_depend, int nt_Tsfc, int nt_alvl, int nt_apnd, int nt_fbri,
                         int tr_pond_cesm, int tr_pond_lvl, int tr_pond_topo, double Tocnfrz, const double *atr,
                         const double *aicen, const double *vicen, const double *vsnon, double *trcrn);
void orc_transport_upwind_state(const orc_geom *g, double dt, int ncat, int ntrcr, int ntrcr_dim, const int32_t *trcr_depend,
                                int nt_Tsfc, int nt_alvl, int nt_apnd, int nt_fbri, int tr_pond_cesm, int tr_pond_lvl, int tr_pond_topo,
                                double Tocnfrz, const double *uvel, const double *vvel, const double *HTE, const double *HTN,
                                const double *tarea, double *aice0, double *aicen, double *vicen, double *vsnon, double *trcrn);
void orc_transport_upwind(const orc_geom *g, double dt, int narr, const double *uvel, const double *vvel,
                          const double *HTE, const double *HTN, const double *tarea, double *works);

/* horizontal_remap (source/ice_transport_remap.F90:309-850), the incremental remapping transport: oracle/remap_oracle.c.
 * mm (nblocks, ncat+1, ny, nx) [plane 0 = open water], tm (nblocks, ncat, ntrace, ny, nx), both in place on physical cells, ghost
 * cells current on entry; tracer_type / depend (1-based, 0 = none) / has_dependents as init_transport builds them
 * (ice_transport_driver.F90:66-183); l_fixed_area must be 0.  Returns 0, 1 (departure points out of bounds), 2 (negative
 * area) or 3 (unsupported option). */
int orc_horizontal_remap(const orc_geom *g, double dt, int ncat, int ntrace, const double *uvel, const double *vvel, double *mm, double *tm,
                         int l_fixed_area, const int32_t *tracer_type, const int32_t *depend, const int32_t *has_dependents,
                         int integral_order, int l_dp_midpt, const double *HTE, const double *HTN, const double *dxu, const double *dyu,
                         const double *tarear, const double *hm);

void orc_evp(const orc_geom *g, const orc_params *p, orc_fields *f, int nsub_override,
             int64_t counts[2], double *loop_seconds /* [0] wall time of the subcycle loop, [1] the halo updates' share of it; may be NULL */);

#ifdef __cplusplus
}
#endif
/* transport_remap (ice_transport_driver.F90:198-627; remap_oracle.c): state_to_tracers, horizontal_remap, tracers_to_state, bound_state */
int orc_transport_remap_state(const orc_geom *g, double dt, int ncat, int ntrcr, int ntrcr_dim, int nt_qsno, int nslyr, double rhos_lfresh,
                              const double *uvel, const double *vvel, double *aice0, double *aicen, double *vicen, double *vsnon, double *trcrn,
                              const int32_t *tracer_type, const int32_t *depend, const int32_t *has_dependents, int integral_order,
                              int l_dp_midpt, const double *HTE, const double *HTN, const double *dxu, const double *dyu, const double *tarear,
                              const double *hm);

/* eap(dt) (ice_dyn_eap.F90:66-486): evp's driver with stress_eap for stress, stepa every tenth subcycle, no stress fold */
void orc_eap(const orc_geom *g, const orc_params *p, orc_fields *f, orc_eap_state *e, int nsub_override, int64_t counts[2], double *loop_seconds);
void orc_eap_stress(int nx, int ny, int ksub, int ndte, int icellt, const int32_t *indxti, const int32_t *indxtj, double arlx1i, double denom1,
                    const double *uvel, const double *vvel, const double *dxt, const double *dyt, const double *dxhy, const double *dyhx,
                    const double *cxp, const double *cyp, const double *cxm, const double *cym, const double *tarear, const double *strength,
                    double *const stressp[4], double *const stressm[4], double *const stress12[4], double *shear, double *divu,
                    double *prs_sig, double *rdg_conv, double *rdg_shear, double *str, const orc_eap_state *e, size_t off);
void orc_eap_stepa(int nx, int ny, double dtei, int icellt, const int32_t *indxti, const int32_t *indxtj,
                   double *const stressp[4], double *const stressm[4], double *const stress12[4], const orc_eap_state *e, size_t off);
double orc_fm_sin(double x);
double orc_fm_cos(double x);
double orc_fm_atan2(double y, double x);

#endif
