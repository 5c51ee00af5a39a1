!=======================================================================
! ice_dyn_evp -- drop-in replacement of source/ice_dyn_evp.F90 (COSIMA/cice5)
! for MI355X: same module name, same public entry point
!
!     subroutine evp (dt)          (reference: ice_dyn_evp.F90:68)
!
! so that ice_step_mod.F90:1119  `if (kdyn == 1) call evp (dt)`  links unchanged.
! All module-global state keeps living in the reference's own modules
! (ice_state, ice_flux, ice_grid, ice_dyn_shared, ...); this module hands those
! arrays to libevpk (HIP kernels) through the ISO_C_BINDING interface evpk_mod
! and copies nothing else.  What stays on the host, exactly as in the reference:
!   * evp_prep1 + halo of icetmask + the T-cell index list, needed only to call
!     ice_strength with its reference signature (ice_dyn_evp.F90:194-212,291-301;
!     ice_dyn_shared.F90:528-537);
!   * timers (timer_dynamics) and abort_ice on error.
! Everything else of evp() -- to_ugrid, t2ugrid_vector, evp_prep2, the ndte
! subcycles of stress + stepu with their halo updates, the tripole stress fold,
! evp_finish, u2tgrid_vector -- runs on the GPU inside evpk_run.
!
! cpp:  EVPK_USE_MPI   broadcast the RCCL unique id with MPI (mpi/ comm layer)
!       AusCOM, ACCESS, coupled   as in the reference
!=======================================================================

      module ice_dyn_evp

      use ice_kinds_mod
      use ice_dyn_shared ! everything
      use, intrinsic :: iso_c_binding
      use evpk_mod

#ifdef AusCOM
      use cpl_parameters
      use cpl_arrays_setup, only : sicemass
#endif

      implicit none
      private
      public :: evp
      public :: evpk_npinned      ! (diagnostic) host arrays page-locked for in-place PCIe transfers
      public :: evpk_pin_module_arrays
      public :: evpk_resident_state, evpk_state_changed_on_host, evpk_device_strength
      public :: evpk_bound_seconds, evpk_loop_seconds
      public :: evpk_download_all, evpk_sparse_io
      public :: evpk_horizontal_remap, evpk_eap, evpk_transport_remap_core
      save

      ! .true. (default): every array the reference's evp leaves modified comes back every call.  .false. (with
      ! evpk_resident_state): only what the host model reads every step -- uvel, vvel (transport), rdg_conv, rdg_shear
      ! (ridging), divu, shear, strocnxT, strocnyT (coupler) and iceumask; the caller sets it .true. for the steps that
      ! write history or a restart (the stresses, strintx/y, strairx/y, ... are then delivered whole).
      logical (kind=log_kind) :: evpk_download_all = .true.
      ! .true.: sparse transfers (evpk_params%sparse_io): only the 64x4-cell tiles with ice move, aice / vice / vsno apart
      logical (kind=log_kind) :: evpk_sparse_io = .false.

      ! Device time of all evp calls so far (HIP events inside the library): the ndte loop, and inside it the halo / tripole
      ! fold / ghost-zone updates -- the share the reference books under timer_bound (ice_dyn_evp.F90:392-400).  ice_timers
      ! offers no way to add a measured duration to a timer (its accumulators are private, mpi/ice_timers.F90), and the
      ! updates run asynchronously beside the kernels, so the host clock cannot bracket them: a CICE timing report shows
      ! the whole call under timer_dynamics, and these two are there to be printed beside it.
      real (kind=dbl_kind) :: evpk_bound_seconds = 0.0_dbl_kind, evpk_loop_seconds = 0.0_dbl_kind

      ! .true.: ice_strength (ice_mechred.F90:2111) runs on the device inside evpk_run, from aice, vice, aicen, vicen,
      ! aice0 and the namelist switches of ice_mechred; the host then skips evp_prep1, the icetmask halo update and
      ! ice_strength.  The device exp() is a fixed < 1 ulp algorithm, so `strength` may differ from the host intrinsic's
      ! in the last bit -- hence opt-in.
      logical (kind=log_kind) :: evpk_device_strength = .false.

      integer (kind=int_kind) :: evpk_npinned = 0

      ! .true. (default): the host model's module arrays (ice_state / ice_flux: static, they cannot be moved) are registered with
      ! evpk_pin_host at the first call and then read / written IN PLACE over PCIe.  The registration is hardened (MADV_NOHUGEPAGE,
      ! mlock) but hipHostRegister mirrors the pages through MMU notifiers, it does not hard-pin them; the environment variable
      ! EVPK_VERIFY_DELIVERY=1 makes the library deliver every in-place plane a second time through its staging buffer and compare
      ! (evpk_stats%delivery_checked / delivery_bad).  .false.: nothing of the host model is registered -- every array moves
      ! through the library's staging copies (hipMemcpy: slower, no dependence on the pages staying where they are).
      ! The arrays this module itself owns (tmass, aiu, umass, icetmask, the int32 mask copies) live on memory the DRIVER
      ! allocates and pins (evpk_host_alloc = hipHostMalloc) either way.
      logical (kind=log_kind) :: evpk_pin_module_arrays = .true.

      ! uvel, vvel, the twelve stresses and iceumask are written by evp only (ice_dyn_evp.F90:336-410; readers:
      ! transport, history, restart), so after the first call the copy on the device is current and only the inputs
      ! are uploaded.  Whoever writes them elsewhere (a restart read after the first step, ...) sets
      ! evpk_state_changed_on_host = .true.; evpk_resident_state = .false. uploads them every call.
      logical (kind=log_kind) :: evpk_resident_state = .true.
      logical (kind=log_kind) :: evpk_state_changed_on_host = .true.

      type (c_ptr) :: ctx = c_null_ptr          ! libevpk context (one per MPI rank = one GPU)
      logical (kind=log_kind) :: ctx_ready = .false.

      ! LOGICAL arrays are not C-interoperable: int32 copies (ice_kinds_mod.F90:20-21)
      integer (c_int32_t), dimension(:,:,:), pointer :: &
         tmask_i => null(), umask_i => null(), iceumask_i => null()     ! (evpk_host_alloc: page-locked by the driver)

      integer (c_int32_t), dimension(:), allocatable, target :: &
         g_ilo, g_ihi, g_jlo, g_jhi, g_iglob, g_jglob

      ! outputs of the library that the reference keeps as locals of evp (allocated once: page-locked below)
      real (kind=dbl_kind), dimension (:,:,:), pointer :: &
         tmass => null(), aiu => null(), umass => null()                ! (evpk_host_alloc)
      integer (kind=int_kind), dimension (:,:,:), allocatable, target :: &
         icetmask

      logical (kind=log_kind) :: pinned = .false.

      character (kind=c_char), dimension(EVPK_UNIQUE_ID_BYTES), target :: uid

!=======================================================================

      contains

!=======================================================================
! First call: describe the block decomposition and the time-invariant grid
! to the library (replaces nothing in the reference: evp() reads the same
! entities from ice_blocks / ice_domain / ice_grid on every call).

      subroutine evpk_setup

      use ice_blocks, only: block, get_block, nx_block, ny_block
      use ice_communicate, only: my_task, master_task, get_num_procs
      use ice_domain, only: nblocks, blocks_ice, ew_boundary_type, ns_boundary_type
      use ice_domain_size, only: nx_global, ny_global, max_blocks
      use ice_exit, only: abort_ice
      use ice_grid, only: tmask, umask, dxt, dyt, dxhy, dyhx, cxp, cyp, cxm, cym, &
          tarear, uarear, tinyarea, tarea, uarea, HTN, HTE
#ifdef EVPK_USE_MPI
      use ice_communicate, only: MPI_COMM_ICE
      include 'mpif.h'
      integer :: ierr
#endif

      type (evpk_geom) :: g
      type (block) :: this_block
      integer (kind=int_kind) :: iblk, rc, ndev

      allocate (g_ilo(nblocks), g_ihi(nblocks), g_jlo(nblocks), g_jhi(nblocks), &
                g_iglob(nblocks), g_jglob(nblocks))
      do iblk = 1, nblocks
         this_block = get_block(blocks_ice(iblk),iblk)
         g_ilo(iblk) = this_block%ilo
         g_ihi(iblk) = this_block%ihi
         g_jlo(iblk) = this_block%jlo
         g_jhi(iblk) = this_block%jhi
         g_iglob(iblk) = this_block%i_glob(this_block%ilo)
         g_jglob(iblk) = this_block%j_glob(this_block%jlo)
      enddo

      call host_alloc_i4 (tmask_i);  call host_alloc_i4 (umask_i);  call host_alloc_i4 (iceumask_i)
      tmask_i = 0
      umask_i = 0
      where (tmask) tmask_i = 1
      where (umask) umask_i = 1

      g%nx_global = nx_global
      g%ny_global = ny_global
      g%nx_block  = nx_block
      g%ny_block  = ny_block
      g%nblocks   = nblocks
      g%ew_boundary = bnd_code(ew_boundary_type)
      g%ns_boundary = bnd_code(ns_boundary_type)
      g%ilo = c_loc(g_ilo);  g%ihi = c_loc(g_ihi)
      g%jlo = c_loc(g_jlo);  g%jhi = c_loc(g_jhi)
      g%iglob_lo = c_loc(g_iglob);  g%jglob_lo = c_loc(g_jglob)
      g%rank   = my_task
      g%nranks = get_num_procs()
      g%device = 0                       ! one rank per GPU: ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES selects it
      g%unique_id = c_null_ptr
      if (g%nranks > 1) then
#ifdef EVPK_USE_MPI
         if (my_task == master_task) then
            if (evpk_get_unique_id(c_loc(uid)) /= 0) call abort_ice('evp: evpk_get_unique_id failed')
         endif
         call MPI_BCAST(uid, EVPK_UNIQUE_ID_BYTES, MPI_BYTE, master_task, MPI_COMM_ICE, ierr)
         g%unique_id = c_loc(uid)
#else
         call abort_ice('evp: more than one task needs -DEVPK_USE_MPI')
#endif
      endif
      g%dxt = loc_r8(dxt);   g%dyt = loc_r8(dyt);   g%dxhy = loc_r8(dxhy); g%dyhx = loc_r8(dyhx)
      g%cxp = loc_r8(cxp);   g%cyp = loc_r8(cyp);   g%cxm  = loc_r8(cxm);  g%cym  = loc_r8(cym)
      g%tarear = loc_r8(tarear); g%uarear = loc_r8(uarear); g%tinyarea = loc_r8(tinyarea)
      g%tarea  = loc_r8(tarea);  g%uarea  = loc_r8(uarea);  g%fcor = loc_r8(fcor_blk)
      g%tmask  = c_loc(tmask_i); g%umask = c_loc(umask_i)
      g%HTN = loc_r8(HTN);  g%HTE = loc_r8(HTE)      ! lets the library rebuild the eight metric planes on the fly if they match

      rc = evpk_create (g, ctx)
      if (rc /= 0) call abort_ice('evp: evpk_create: '//trim(evpk_error_string(c_null_ptr)))
      ctx_ready = .true.

      end subroutine evpk_setup

!=======================================================================

! The reference declares its module arrays without TARGET (ice_state.F90, ice_flux.F90, ice_grid.F90), so C_LOC cannot be
! applied to them directly.  Passed to an assumed-size TARGET dummy (sequence association: a contiguous whole array is
! passed by address, no copy) their address can be taken; the arrays live as long as the run.

      type (c_ptr) function loc_r8 (a)
      real (kind=dbl_kind), dimension (*), intent(in), target :: a
      loc_r8 = c_loc(a)
      end function loc_r8

      ! a (nx_block, ny_block, max_blocks) array of this module on memory the driver allocates and pins (evpk_host_alloc); ordinary
      ! memory if the library refuses (it is then moved through the staging copies)
      subroutine host_alloc_r8 (a)
      use ice_blocks, only: nx_block, ny_block
      use ice_domain_size, only: max_blocks
      real (kind=dbl_kind), dimension (:,:,:), pointer :: a
      type (c_ptr) :: p
      integer (c_size_t) :: n
      n = int(nx_block, c_size_t) * int(ny_block, c_size_t) * int(max(max_blocks, 1), c_size_t) * 8_c_size_t
      if (evpk_host_alloc (n, p) == 0) then
         call c_f_pointer (p, a, (/ nx_block, ny_block, max_blocks /))
         evpk_npinned = evpk_npinned + 1
      else
         allocate (a(nx_block,ny_block,max_blocks))
      endif
      a = 0.0_dbl_kind
      end subroutine host_alloc_r8

      subroutine host_alloc_i4 (a)
      use ice_blocks, only: nx_block, ny_block
      use ice_domain_size, only: max_blocks
      integer (c_int32_t), dimension (:,:,:), pointer :: a
      type (c_ptr) :: p
      integer (c_size_t) :: n
      n = int(nx_block, c_size_t) * int(ny_block, c_size_t) * int(max(max_blocks, 1), c_size_t) * 4_c_size_t
      if (evpk_host_alloc (n, p) == 0) then
         call c_f_pointer (p, a, (/ nx_block, ny_block, max_blocks /))
         evpk_npinned = evpk_npinned + 1
      else
         allocate (a(nx_block,ny_block,max_blocks))
      endif
      a = 0
      end subroutine host_alloc_i4

      subroutine pin_r8 (a, n)
      real (kind=dbl_kind), dimension (*), intent(in), target :: a
      integer (kind=int_kind), intent(in) :: n      ! elements
      integer (c_int) :: rc
      if (n > 0) then
         rc = evpk_pin_host (c_loc(a), int(n, c_size_t) * 8_c_size_t)
         if (rc == 0) evpk_npinned = evpk_npinned + 1
      endif
      end subroutine pin_r8

!=======================================================================

      integer (c_int32_t) function bnd_code (name)
      use ice_exit, only: abort_ice
      character (*), intent(in) :: name
      select case (trim(name))
      case ('cyclic');   bnd_code = EVPK_BND_CYCLIC
      case ('open');     bnd_code = EVPK_BND_OPEN
      case ('closed');   bnd_code = EVPK_BND_CLOSED
      case ('tripole');  bnd_code = EVPK_BND_TRIPOLE
      case default
         bnd_code = -1
         call abort_ice('evp: boundary type not supported on the GPU path: '//trim(name))
      end select
      end function bnd_code

!=======================================================================
! Elastic-viscous-plastic dynamics driver -- same interface as the reference.

      subroutine evp (dt)

      use ice_atmo, only: Cdn_ocn
      use ice_boundary, only: ice_HaloUpdate
      use ice_blocks, only: block, get_block, nx_block, ny_block
      use ice_constants, only: field_loc_center, field_type_scalar, c0, &
          rhow, rhoi, rhos, gravit, p001, p01
      use ice_domain, only: nblocks, blocks_ice, halo_info
      use ice_domain_size, only: max_blocks
      use ice_exit, only: abort_ice
      use ice_flux, only: rdg_conv, rdg_shear, prs_sig, strairxT, strairyT, &
          strairx, strairy, uocn, vocn, ss_tltx, ss_tlty, iceumask, fm, &
          strtltx, strtlty, strocnx, strocny, strintx, strinty, &
          strocnxT, strocnyT, strax, stray, &
          stressp_1, stressp_2, stressp_3, stressp_4, &
          stressm_1, stressm_2, stressm_3, stressm_4, &
          stress12_1, stress12_2, stress12_3, stress12_4
      use ice_mechred, only: ice_strength, kstrength, krdg_partic, krdg_redist, mu_rdg, Cf
      use ice_domain_size, only: ncat
      use ice_state, only: aice, vice, vsno, uvel, vvel, divu, shear, &
          aice_init, aice0, aicen, vicen, strength
      use ice_timers, only: timer_dynamics, timer_bound, &
          ice_timer_start, ice_timer_stop
#ifdef ACCESS
      use ice_atmo, only: calc_strair
#endif

      real (kind=dbl_kind), intent(in) :: &
         dt      ! time step

      ! local variables

      integer (kind=int_kind) :: &
         iblk, ilo,ihi,jlo,jhi, i, j, icellt, rc

      integer (kind=int_kind), dimension (nx_block*ny_block) :: &
         indxti, indxtj     ! compressed T-cell index list for ice_strength

      type (block) :: this_block
      type (evpk_params)  :: p
      type (evpk_step_in) :: sin
      type (evpk_state)   :: st
      type (evpk_stats)   :: stats

      call ice_timer_start(timer_dynamics) ! dynamics

      if (.not. ctx_ready) call evpk_setup

      if (.not. associated(tmass)) then
         call host_alloc_r8 (tmass);  call host_alloc_r8 (aiu);  call host_alloc_r8 (umass)
         allocate (icetmask(nx_block,ny_block,max_blocks))
      endif

      !-----------------------------------------------------------------
      ! scalars of set_evp_parameters (ice_dyn_shared.F90:185-259), read
      ! every call so that a changed dt / namelist is honoured
      !-----------------------------------------------------------------
      p%dt = dt
      p%ndte = ndte
      p%revised_evp = merge(1, 0, revised_evp)
      p%revp = revp;  p%ecci = ecci;  p%denom1 = denom1
      p%arlx1i = arlx1i;  p%brlx = brlx
      p%cosw = cosw;  p%sinw = sinw
      p%rhow = rhow;  p%rhoi = rhoi;  p%rhos = rhos;  p%gravit = gravit
      p%a_min = p001; p%m_min = p01      ! a_min, m_min are private parameters of ice_dyn_shared (:60-61)
#ifdef coupled
      p%tilt_from_slope = 1
#else
      p%tilt_from_slope = 0
#endif
#ifdef AusCOM
      if (.not. use_ocnslope) p%tilt_from_slope = 0      ! ice_dyn_shared.F90:604-608
#endif
      p%wind_on_ugrid = 0
#ifdef ACCESS
      if (.not. calc_strair) p%wind_on_ugrid = 1         ! ice_dyn_evp.F90:226-228
#endif
      p%kstrength = kstrength;  p%krdg_partic = krdg_partic;  p%krdg_redist = krdg_redist
      p%ncat = ncat;  p%mu_rdg = mu_rdg;  p%Cf = Cf
      p%sparse_io = merge(1, 0, evpk_sparse_io)
      rc = evpk_set_params (ctx, p)
      if (rc /= 0) call abort_ice('evp: evpk_set_params: '//trim(evpk_error_string(ctx)))

      !-----------------------------------------------------------------
      ! ice strength (host, reference routine and signature): needs the
      ! T-cell list of evp_prep2, hence evp_prep1 + halo of icetmask
      !-----------------------------------------------------------------
      if (.not. evpk_device_strength) then
      do iblk = 1, nblocks
         this_block = get_block(blocks_ice(iblk),iblk)
         call evp_prep1 (nx_block,           ny_block,           &
                         this_block%ilo, this_block%ihi, this_block%jlo, this_block%jhi, &
                         aice    (:,:,iblk), vice    (:,:,iblk), &
                         vsno    (:,:,iblk), tmask_l (iblk),     &
                         strairxT(:,:,iblk), strairyT(:,:,iblk), &
                         strairx (:,:,iblk), strairy (:,:,iblk), &
                         tmass   (:,:,iblk), icetmask(:,:,iblk))
      enddo
      call ice_timer_start(timer_bound)
      call ice_HaloUpdate (icetmask,          halo_info, &
                           field_loc_center,  field_type_scalar)
      call ice_timer_stop(timer_bound)

      do iblk = 1, nblocks
         this_block = get_block(blocks_ice(iblk),iblk)
         ilo = this_block%ilo;  ihi = this_block%ihi
         jlo = this_block%jlo;  jhi = this_block%jhi
         icellt = 0
         do j = jlo, jhi+1
         do i = ilo, ihi+1
            if (icetmask(i,j,iblk) == 1) then
               icellt = icellt + 1
               indxti(icellt) = i
               indxtj(icellt) = j
            endif
         enddo
         enddo
         call ice_strength (nx_block, ny_block,   &
                            ilo, ihi, jlo, jhi,   &
                            icellt,               &
                            indxti,   indxtj,     &
                            aice    (:,:,  iblk), &
                            vice    (:,:,  iblk), &
                            aice0   (:,:,  iblk), &
                            aicen   (:,:,:,iblk), &
                            vicen   (:,:,:,iblk), &
                            strength(:,:,  iblk) )
      enddo

      endif   ! host ice_strength

      !-----------------------------------------------------------------
      ! everything else of evp(): one call into the HIP library
      !-----------------------------------------------------------------
      iceumask_i = 0
      where (iceumask) iceumask_i = 1

      sin%aice = loc_r8(aice);  sin%vice = loc_r8(vice);  sin%vsno = loc_r8(vsno)
      sin%aice_init = loc_r8(aice_init)
      sin%strairxT = loc_r8(strairxT);  sin%strairyT = loc_r8(strairyT)
      sin%strax = loc_r8(strax);        sin%stray = loc_r8(stray)
      sin%uocn = loc_r8(uocn);          sin%vocn = loc_r8(vocn)
      sin%ss_tltx = loc_r8(ss_tltx);    sin%ss_tlty = loc_r8(ss_tlty)
      sin%Cdn_ocn = loc_r8(Cdn_ocn)
      sin%strength = loc_r8(strength)
      sin%aicen = c_null_ptr;  sin%vicen = c_null_ptr;  sin%aice0 = c_null_ptr
      st%strength = loc_r8(strength)          ! comes back halo-updated (reference: ice_dyn_evp.F90:311-312)
      if (evpk_device_strength) then
         sin%strength = c_null_ptr
         sin%aicen = loc_r8(aicen);  sin%vicen = loc_r8(vicen);  sin%aice0 = loc_r8(aice0)
      endif

      st%uvel = loc_r8(uvel);  st%vvel = loc_r8(vvel)
      st%stressp(1) = loc_r8(stressp_1);   st%stressp(2) = loc_r8(stressp_2)
      st%stressp(3) = loc_r8(stressp_3);   st%stressp(4) = loc_r8(stressp_4)
      st%stressm(1) = loc_r8(stressm_1);   st%stressm(2) = loc_r8(stressm_2)
      st%stressm(3) = loc_r8(stressm_3);   st%stressm(4) = loc_r8(stressm_4)
      st%stress12(1) = loc_r8(stress12_1); st%stress12(2) = loc_r8(stress12_2)
      st%stress12(3) = loc_r8(stress12_3); st%stress12(4) = loc_r8(stress12_4)
      st%iceumask = c_loc(iceumask_i)
      st%divu = loc_r8(divu);          st%shear = loc_r8(shear)
      st%rdg_conv = loc_r8(rdg_conv);  st%rdg_shear = loc_r8(rdg_shear)
      st%prs_sig = loc_r8(prs_sig)
      st%strintx = loc_r8(strintx);    st%strinty = loc_r8(strinty)
      st%strocnx = loc_r8(strocnx);    st%strocny = loc_r8(strocny)
      st%strocnxT = loc_r8(strocnxT);  st%strocnyT = loc_r8(strocnyT)
      st%strairx = loc_r8(strairx);    st%strairy = loc_r8(strairy)
      st%strtltx = loc_r8(strtltx);    st%strtlty = loc_r8(strtlty)
      st%fm = loc_r8(fm)
      st%tmass = c_loc(tmass)
      st%aiu = c_loc(aiu);  st%umass = c_loc(umass)
      st%uvel_init = loc_r8(uvel_init);  st%vvel_init = loc_r8(vvel_init)
      st%icetmask = c_null_ptr

      if (.not. pinned .and. evpk_pin_module_arrays) then
         ! The arrays handed over live as long as the run: page-lock them once, so that the library moves them in
         ! place over PCIe instead of through staging copies (a refusal leaves the staged path in use).
         call pin_r8 (aice, size(aice));  call pin_r8 (vice, size(vice));  call pin_r8 (vsno, size(vsno));  call pin_r8 (aice_init, size(aice_init))
         call pin_r8 (strairxT, size(strairxT));  call pin_r8 (strairyT, size(strairyT));  call pin_r8 (strax, size(strax));  call pin_r8 (stray, size(stray))
         call pin_r8 (uocn, size(uocn));  call pin_r8 (vocn, size(vocn));  call pin_r8 (ss_tltx, size(ss_tltx));  call pin_r8 (ss_tlty, size(ss_tlty))
         call pin_r8 (Cdn_ocn, size(Cdn_ocn));  call pin_r8 (strength, size(strength))
         call pin_r8 (uvel, size(uvel));  call pin_r8 (vvel, size(vvel))
         call pin_r8 (stressp_1, size(stressp_1));  call pin_r8 (stressp_2, size(stressp_2));  call pin_r8 (stressp_3, size(stressp_3));  call pin_r8 (stressp_4, size(stressp_4))
         call pin_r8 (stressm_1, size(stressm_1));  call pin_r8 (stressm_2, size(stressm_2));  call pin_r8 (stressm_3, size(stressm_3));  call pin_r8 (stressm_4, size(stressm_4))
         call pin_r8 (stress12_1, size(stress12_1)); call pin_r8 (stress12_2, size(stress12_2)); call pin_r8 (stress12_3, size(stress12_3)); call pin_r8 (stress12_4, size(stress12_4))
         call pin_r8 (divu, size(divu));  call pin_r8 (shear, size(shear));  call pin_r8 (rdg_conv, size(rdg_conv));  call pin_r8 (rdg_shear, size(rdg_shear))
         call pin_r8 (prs_sig, size(prs_sig));  call pin_r8 (strintx, size(strintx));  call pin_r8 (strinty, size(strinty))
         call pin_r8 (strocnx, size(strocnx));  call pin_r8 (strocny, size(strocny));  call pin_r8 (strocnxT, size(strocnxT));  call pin_r8 (strocnyT, size(strocnyT))
         call pin_r8 (strairx, size(strairx));  call pin_r8 (strairy, size(strairy));  call pin_r8 (strtltx, size(strtltx));  call pin_r8 (strtlty, size(strtlty))
         call pin_r8 (fm, size(fm))
         call pin_r8 (uvel_init, size(uvel_init));  call pin_r8 (vvel_init, size(vvel_init))
         pinned = .true.
      endif

      if (evpk_resident_state .and. .not. evpk_download_all) then
         ! the every-step set only (anything else stays c_null_ptr: skipped by evpk_download)
         st%stressp = c_null_ptr;  st%stressm = c_null_ptr;  st%stress12 = c_null_ptr
         st%prs_sig = c_null_ptr;  st%strintx = c_null_ptr;  st%strinty = c_null_ptr
         st%strocnx = c_null_ptr;  st%strocny = c_null_ptr
         st%strairx = c_null_ptr;  st%strairy = c_null_ptr;  st%strtltx = c_null_ptr;  st%strtlty = c_null_ptr
         st%fm = c_null_ptr;  st%aiu = c_null_ptr;  st%umass = c_null_ptr
         st%uvel_init = c_null_ptr;  st%vvel_init = c_null_ptr;  st%strength = c_null_ptr
      endif

      if (evpk_resident_state .and. .not. evpk_state_changed_on_host) then
         rc = evpk_upload (ctx, sin, c_null_ptr)
         if (rc == 0) rc = evpk_prep (ctx)
         if (rc == 0) rc = evpk_subcycle (ctx, int(ndte, c_int32_t))
         if (rc == 0) rc = evpk_finish (ctx)
         if (rc == 0) rc = evpk_download (ctx, st)
      else
         rc = evpk_run (ctx, sin, st)
      endif
      if (rc /= 0) call abort_ice('evp: evpk_run: '//trim(evpk_error_string(ctx)))
      evpk_state_changed_on_host = .false.
      if (evpk_get_stats (ctx, stats) == 0) then
         evpk_bound_seconds = evpk_bound_seconds + 1.0e-3_dbl_kind * real(stats%bound_ms, kind=dbl_kind)
         evpk_loop_seconds  = evpk_loop_seconds  + 1.0e-3_dbl_kind * real(stats%loop_ms, kind=dbl_kind)
      endif

      iceumask = (iceumask_i == 1)

#ifdef AusCOM
      sicemass(:,:,:) = tmass(:,:,:)          ! ice_dyn_evp.F90:205-207
#endif


      call ice_timer_stop(timer_dynamics)    ! dynamics

      contains

         ! tmask(:,:,iblk) as the explicit-shape logical argument evp_prep1 expects
         function tmask_l (ib) result (m)
            use ice_grid, only: tmask
            integer (kind=int_kind), intent(in) :: ib
            logical (kind=log_kind) :: m(nx_block,ny_block)
            m = tmask(:,:,ib)
         end function tmask_l

      end subroutine evp

!=======================================================================
! horizontal_remap (reference: ice_transport_remap.F90:309-850) on the velocities the last evp left on the device: same
! argument list, so that ice_transport_driver.F90:216
!     use ice_transport_remap, only: horizontal_remap, make_masks
! becomes
!     use ice_transport_remap, only: make_masks
!     use ice_dyn_evp, only: horizontal_remap => evpk_horizontal_remap
! and the call at :475-481 stays as it is.  uvel, vvel are accepted for that reason only: the library advects with the
! device copies (bit-identical to the host arrays after evp).  SURVEY.md S8 row f-3.

      subroutine evpk_horizontal_remap (dt,                ntrace,     &
                                        uvel,              vvel,       &
                                        mm,                tm,         &
                                        l_fixed_area,                  &
                                        tracer_type,       depend,     &
                                        has_dependents,                &
                                        integral_order,                &
                                        l_dp_midpt)

      use ice_blocks, only: nx_block, ny_block
      use ice_domain_size, only: ncat, max_blocks
      use ice_grid, only: dxu, dyu, hm
      use ice_exit, only: abort_ice

      real (kind=dbl_kind), intent(in) :: dt
      integer (kind=int_kind), intent(in) :: ntrace
      real (kind=dbl_kind), intent(in), dimension(nx_block,ny_block,max_blocks) :: uvel, vvel
      real (kind=dbl_kind), intent(inout), target, dimension (nx_block,ny_block,0:ncat,max_blocks) :: mm
      real (kind=dbl_kind), intent(inout), target, dimension (nx_block,ny_block,ntrace,ncat,max_blocks) :: tm
      logical, intent(in) :: l_fixed_area
      integer (kind=int_kind), dimension (ntrace), intent(in) :: tracer_type, depend
      logical (kind=log_kind), dimension (ntrace), intent(in) :: has_dependents
      integer (kind=int_kind), intent(in) :: integral_order
      logical (kind=log_kind), intent(in) :: l_dp_midpt

      integer (c_int32_t), dimension (max(ntrace,1)), target :: ttype, dep, has
      integer (c_int) :: rc
      logical (kind=log_kind), save :: first = .true.

      if (.not. c_associated(ctx)) call abort_ice('horizontal_remap: no velocities on the device: evp has not run yet')
      if (first) then
         rc = evpk_remap_init (ctx, loc_r8(dxu), loc_r8(dyu), loc_r8(hm))
         if (rc /= 0) call abort_ice('horizontal_remap: evpk_remap_init: '//trim(evpk_error_string(ctx)))
         first = .false.
      endif
      ttype(1:ntrace) = tracer_type(1:ntrace)
      dep(1:ntrace) = depend(1:ntrace)
      has(1:ntrace) = merge(1, 0, has_dependents(1:ntrace))
      rc = evpk_transport_remap (ctx, dt, int(ncat, c_int32_t), int(ntrace, c_int32_t), loc_r8(mm), loc_r8(tm), &
                                 c_loc(ttype), c_loc(dep), c_loc(has), int(integral_order, c_int32_t), &
                                 merge(1_c_int32_t, 0_c_int32_t, l_dp_midpt), merge(1_c_int32_t, 0_c_int32_t, l_fixed_area))
      ! the reference's two l_stop cases (:498-503 departure_points, :822-839 update_fields) and everything else
      if (rc == EVPK_REMAP_BAD_DEPARTURE) call abort_ice('remap transport: bad departure points')
      if (rc == EVPK_REMAP_NEGATIVE_MASS) call abort_ice('remap transport: negative area')
      if (rc /= 0) call abort_ice('horizontal_remap: evpk_transport_remap: '//trim(evpk_error_string(ctx)))

      end subroutine evpk_horizontal_remap

!=======================================================================
! The core of transport_remap (reference: ice_transport_driver.F90:198-627 with its compile-time-off checks): state_to_tracers,
! horizontal_remap, tracers_to_state and bound_state in ONE call on the module arrays of ice_state -- for a host that wants the
! whole advection on the device (its state in device or page-locked memory) instead of the call-compatible
! evpk_horizontal_remap above.  In transport_remap the lines from the state_to_tracers loop (:340-349) through bound_state
! (:500-503) become
!     call evpk_transport_remap_core (dt, ntrace, tracer_type, depend, has_dependents, integral_order, l_dp_midpt)

      subroutine evpk_transport_remap_core (dt, ntrace, tracer_type, depend, has_dependents, integral_order, l_dp_midpt)

      use ice_constants, only: rhos, Lfresh
      use ice_domain_size, only: ncat, nslyr, max_ntrcr
      use ice_grid, only: dxu, dyu, hm
      use ice_state, only: aice0, aicen, vicen, vsnon, trcrn, ntrcr, nt_qsno
      use ice_exit, only: abort_ice

      real (kind=dbl_kind), intent(in) :: dt
      integer (kind=int_kind), intent(in) :: ntrace
      integer (kind=int_kind), dimension (ntrace), intent(in) :: tracer_type, depend
      logical (kind=log_kind), dimension (ntrace), intent(in) :: has_dependents
      integer (kind=int_kind), intent(in) :: integral_order
      logical (kind=log_kind), intent(in) :: l_dp_midpt

      integer (c_int32_t), dimension (max(ntrace,1)), target :: ttype, dep, has
      integer (c_int) :: rc
      logical (kind=log_kind), save :: first = .true.

      if (.not. c_associated(ctx)) call abort_ice('transport_remap: no velocities on the device: evp has not run yet')
      if (ntrace /= ntrcr + 2) call abort_ice('transport_remap: ntrace /= ntrcr + 2')
      if (first) then
         rc = evpk_remap_init (ctx, loc_r8(dxu), loc_r8(dyu), loc_r8(hm))
         if (rc /= 0) call abort_ice('transport_remap: evpk_remap_init: '//trim(evpk_error_string(ctx)))
         first = .false.
      endif
      ttype(1:ntrace) = tracer_type(1:ntrace)
      dep(1:ntrace) = depend(1:ntrace)
      has(1:ntrace) = merge(1, 0, has_dependents(1:ntrace))
      rc = evpk_transport_remap_state (ctx, dt, int(ncat, c_int32_t), int(ntrcr, c_int32_t), int(max_ntrcr, c_int32_t), &
                                       int(nt_qsno, c_int32_t), int(nslyr, c_int32_t), rhos*Lfresh, &
                                       loc_r8(aice0), loc_r8(aicen), loc_r8(vicen), loc_r8(vsnon), loc_r8(trcrn), &
                                       c_loc(ttype), c_loc(dep), c_loc(has), int(integral_order, c_int32_t), &
                                       merge(1_c_int32_t, 0_c_int32_t, l_dp_midpt))
      if (rc == EVPK_REMAP_BAD_DEPARTURE) call abort_ice('remap transport: bad departure points')
      if (rc == EVPK_REMAP_NEGATIVE_MASS) call abort_ice('remap transport: negative area')
      if (rc /= 0) call abort_ice('transport_remap: evpk_transport_remap_state: '//trim(evpk_error_string(ctx)))

      end subroutine evpk_transport_remap_core

!=======================================================================
! eap(dt) (reference: ice_dyn_eap.F90:66-486) on the device.  The lookup tables and the structure tensor are private to the
! reference's module ice_dyn_eap, so the call comes from inside it: the body of its `subroutine eap (dt)` becomes
!
!     use ice_dyn_evp, only: evpk_eap
!     call evpk_eap (dt, nx_yield, ny_yield, na_yield, s11r, s12r, s22r, s11s, s12s, s22s, &
!                    a11_1, a11_2, a11_3, a11_4, a12_1, a12_2, a12_3, a12_4, a11, a12,    &
!                    e11, e12, e22, yieldstress11, yieldstress12, yieldstress22, s11, s12, s22)
!
! (INTEGRATION.md S3; tests/test_ref_interfaces.py compiles the reference's module edited that way).  init_eap, the restart
! routines and the module variables stay the reference's.  Everything evp(dt) of this module does for kdyn = 1 -- inputs up,
! ice_strength on the host, the loop and evp_finish on the device, outputs down -- is shared; the structure tensor lives on the
! device between calls (uploaded at the first call and after a restart read: evpk_state_changed_on_host).

      subroutine evpk_eap (dt, nx_yield, ny_yield, na_yield, s11r, s12r, s22r, s11s, s12s, s22s, &
                           a11_1, a11_2, a11_3, a11_4, a12_1, a12_2, a12_3, a12_4, a11, a12,    &
                           e11, e12, e22, yieldstress11, yieldstress12, yieldstress22, s11, s12, s22)

      use ice_blocks, only: nx_block, ny_block
      use ice_domain_size, only: max_blocks
      use ice_exit, only: abort_ice

      real (kind=dbl_kind), intent(in) :: dt
      integer (kind=int_kind), intent(in) :: nx_yield, ny_yield, na_yield
      real (kind=dbl_kind), dimension (nx_yield,ny_yield,na_yield), intent(in), target :: &
         s11r, s12r, s22r, s11s, s12s, s22s
      real (kind=dbl_kind), dimension (nx_block,ny_block,max_blocks), intent(inout), target :: &
         a11_1, a11_2, a11_3, a11_4, a12_1, a12_2, a12_3, a12_4, a11, a12, &
         e11, e12, e22, yieldstress11, yieldstress12, yieldstress22, s11, s12, s22

      type (evpk_eap_state) :: es
      integer (c_int) :: rc
      logical (kind=log_kind) :: push
      logical (kind=log_kind), save :: first = .true.

      if (.not. ctx_ready) call evpk_setup
      if (first) then
         rc = evpk_eap_init (ctx, int(nx_yield, c_int32_t), int(ny_yield, c_int32_t), int(na_yield, c_int32_t), &
                             loc_r8(s11r), loc_r8(s12r), loc_r8(s22r), loc_r8(s11s), loc_r8(s12s), loc_r8(s22s))
         if (rc /= 0) call abort_ice('eap: evpk_eap_init: '//trim(evpk_error_string(ctx)))
      endif
      push = first .or. evpk_state_changed_on_host       ! (evp below clears the flag)
      first = .false.
      es%a11_c(1) = loc_r8(a11_1);  es%a11_c(2) = loc_r8(a11_2);  es%a11_c(3) = loc_r8(a11_3);  es%a11_c(4) = loc_r8(a11_4)
      es%a12_c(1) = loc_r8(a12_1);  es%a12_c(2) = loc_r8(a12_2);  es%a12_c(3) = loc_r8(a12_3);  es%a12_c(4) = loc_r8(a12_4)
      es%a11 = loc_r8(a11);  es%a12 = loc_r8(a12)
      es%e11 = loc_r8(e11);  es%e12 = loc_r8(e12);  es%e22 = loc_r8(e22)
      es%yieldstress11 = loc_r8(yieldstress11);  es%yieldstress12 = loc_r8(yieldstress12);  es%yieldstress22 = loc_r8(yieldstress22)
      es%s11 = loc_r8(s11);  es%s12 = loc_r8(s12);  es%s22 = loc_r8(s22)
      if (push) then
         rc = evpk_eap_upload (ctx, es)
         if (rc /= 0) call abort_ice('eap: evpk_eap_upload: '//trim(evpk_error_string(ctx)))
      endif

      call evp (dt)          ! the context is in EAP mode since evpk_eap_init: prep, the eap loop, finish

      rc = evpk_eap_download (ctx, es)
      if (rc /= 0) call abort_ice('eap: evpk_eap_download: '//trim(evpk_error_string(ctx)))

      end subroutine evpk_eap

!=======================================================================

      end module ice_dyn_evp

!=======================================================================
