!=======================================================================
! evpk_mod -- ISO_C_BINDING view of libevpk (include/evpk.h): the C ABI of the
! MI355X EVP solver, as a Fortran host model binds it.
!
! The derived types mirror the C structs field for field (bind(C)); all array
! arguments are passed as c_loc of the caller's own block arrays
! real(dbl_kind) a(nx_block,ny_block,max_blocks) -- the library borrows them
! for the duration of a call only.  Fortran LOGICAL arrays are not interoperable:
! the caller converts tmask/umask/iceumask to integer(c_int32_t) 0/1.
!=======================================================================
      module evpk_mod

      use, intrinsic :: iso_c_binding
      implicit none
      private

      integer (c_int), parameter, public :: &
         EVPK_BND_CYCLIC = 0, EVPK_BND_OPEN = 1, EVPK_BND_CLOSED = 2, EVPK_BND_TRIPOLE = 3
      integer, parameter, public :: EVPK_UNIQUE_ID_BYTES = 128

      type, bind(C), public :: evpk_geom
         integer (c_int32_t) :: nx_global, ny_global
         integer (c_int32_t) :: nx_block, ny_block, nblocks
         integer (c_int32_t) :: ew_boundary, ns_boundary
         type (c_ptr) :: ilo, ihi, jlo, jhi
         type (c_ptr) :: iglob_lo, jglob_lo
         integer (c_int32_t) :: rank, nranks
         integer (c_int32_t) :: device
         type (c_ptr) :: unique_id
         type (c_ptr) :: dxt, dyt, dxhy, dyhx, cxp, cyp, cxm, cym
         type (c_ptr) :: tarear, uarear, tinyarea, tarea, uarea, fcor
         type (c_ptr) :: tmask, umask
         type (c_ptr) :: HTN, HTE            ! optional (c_null_ptr): see include/evpk.h
      end type evpk_geom

      type, bind(C), public :: evpk_params
         real (c_double) :: dt
         integer (c_int32_t) :: ndte
         integer (c_int32_t) :: revised_evp
         real (c_double) :: revp, ecci, denom1, arlx1i, brlx
         real (c_double) :: cosw, sinw
         real (c_double) :: rhow, rhoi, rhos, gravit
         real (c_double) :: a_min, m_min
         integer (c_int32_t) :: tilt_from_slope
         integer (c_int32_t) :: wind_on_ugrid
         integer (c_int32_t) :: kstrength, krdg_partic, krdg_redist, ncat    ! ice_strength on the device (strength = c_null_ptr)
         real (c_double) :: mu_rdg, Cf
         integer (c_int32_t) :: sparse_io = 0     ! 1: sparse transfers in resident-state mode (include/evpk.h)
         integer (c_int32_t) :: reserved_ = 0
      end type evpk_params

      type, bind(C), public :: evpk_step_in
         type (c_ptr) :: aice, vice, vsno, aice_init
         type (c_ptr) :: strairxT, strairyT, strax, stray
         type (c_ptr) :: uocn, vocn, ss_tltx, ss_tlty, Cdn_ocn
         type (c_ptr) :: strength            ! c_null_ptr: the library evaluates ice_strength itself, from ...
         type (c_ptr) :: aicen, vicen, aice0 ! ... the thickness distribution (kstrength = 1)
      end type evpk_step_in

      type, bind(C), public :: evpk_state
         type (c_ptr) :: uvel, vvel
         type (c_ptr) :: stressp(4), stressm(4), stress12(4)
         type (c_ptr) :: iceumask
         type (c_ptr) :: divu, shear, rdg_conv, rdg_shear, prs_sig
         type (c_ptr) :: strintx, strinty, strocnx, strocny, strocnxT, strocnyT
         type (c_ptr) :: strairx, strairy, strtltx, strtlty, fm
         type (c_ptr) :: tmass
         type (c_ptr) :: aiu, umass, uvel_init, vvel_init
         type (c_ptr) :: icetmask
         type (c_ptr) :: strength            ! out: the strength the library computed
      end type evpk_state

      type, bind(C), public :: evpk_stats
         integer (c_int64_t) :: icellt, icellu, ncell_slab
         integer (c_int32_t) :: nstrips, nstrips_total, subcycles_done
         real (c_float) :: loop_ms, kernel_ms
         integer (c_int32_t) :: kernel_launches
         real (c_float) :: kernel2_ms
         integer (c_int32_t) :: kernel2_launches
         integer (c_int32_t) :: strip_rows, strip_rows2, nstrips2
         integer (c_int32_t) :: zone_cols, zone_exchanges
         integer (c_int64_t) :: zone_bytes
         integer (c_int32_t) :: overlap_split, tile_kernel, kernel_timed, kernel2_timed
         real (c_float) :: bound_ms
         integer (c_int32_t) :: bound_updates, compact_metrics, transport, band_row_exchanges
         real (c_float) :: kernel3_ms
         integer (c_int32_t) :: kernel3_launches, kernel3_timed, strip_rows3, nstrips3
         integer (c_int32_t) :: rccl_ranks, device, device_pci
         integer (c_int64_t) :: delivery_checked, delivery_bad
      end type evpk_stats

      ! evpk_eap_state (include/evpk.h): the structure tensor at the four corners, its cell means, the EAP history fields
      type, bind(C) :: evpk_eap_state
         type (c_ptr) :: a11_c(4) = c_null_ptr, a12_c(4) = c_null_ptr
         type (c_ptr) :: a11 = c_null_ptr, a12 = c_null_ptr
         type (c_ptr) :: e11 = c_null_ptr, e12 = c_null_ptr, e22 = c_null_ptr
         type (c_ptr) :: yieldstress11 = c_null_ptr, yieldstress12 = c_null_ptr, yieldstress22 = c_null_ptr
         type (c_ptr) :: s11 = c_null_ptr, s12 = c_null_ptr, s22 = c_null_ptr
      end type evpk_eap_state

      public :: evpk_get_unique_id, evpk_create, evpk_set_params, evpk_run, &
                evpk_get_stats, evpk_destroy, evpk_last_error, evpk_error_string, &
                evpk_principal_stress, evpk_pin_host, evpk_unpin_host, evpk_host_alloc, evpk_host_free, evpk_host_is_mapped, &
                evpk_upload, evpk_prep, evpk_subcycle, evpk_finish, evpk_download, &
                evpk_connect, evpk_device_check, evpk_restart_write, evpk_restart_read, &
                evpk_halo_update, evpk_halo_update_stress, evpk_transport_upwind_state, &
                evpk_transport_upwind, evpk_remap_init, evpk_transport_remap, evpk_transport_remap_state, &
                EVPK_REMAP_BAD_DEPARTURE, EVPK_REMAP_NEGATIVE_MASS, &
                evpk_eap_state, evpk_eap_init, evpk_eap_upload, evpk_eap_download

      integer (c_int), parameter :: EVPK_REMAP_BAD_DEPARTURE = 11, EVPK_REMAP_NEGATIVE_MASS = 12     ! include/evpk.h

      interface
         integer (c_int) function evpk_get_unique_id (id) bind(C, name='evpk_get_unique_id')
            import :: c_int, c_ptr
            type (c_ptr), value :: id
         end function
         integer (c_int) function evpk_create (g, ctx) bind(C, name='evpk_create')
            import :: c_int, c_ptr, evpk_geom
            type (evpk_geom), intent(in) :: g
            type (c_ptr), intent(out) :: ctx
         end function
         integer (c_int) function evpk_set_params (ctx, p) bind(C, name='evpk_set_params')
            import :: c_int, c_ptr, evpk_params
            type (c_ptr), value :: ctx
            type (evpk_params), intent(in) :: p
         end function
         integer (c_int) function evpk_run (ctx, sin, st) bind(C, name='evpk_run')
            import :: c_int, c_ptr, evpk_step_in, evpk_state
            type (c_ptr), value :: ctx
            type (evpk_step_in), intent(in) :: sin
            type (evpk_state), intent(in) :: st
         end function
         ! evpk_run in stages.  st = c_null_ptr in evpk_upload: inputs only, the prognostic state (uvel, vvel, the
         ! twelve stresses, iceumask) stays as the previous call left it on the device
         integer (c_int) function evpk_upload (ctx, sin, st) bind(C, name='evpk_upload')
            import :: c_int, c_ptr, evpk_step_in
            type (c_ptr), value :: ctx
            type (evpk_step_in), intent(in) :: sin
            type (c_ptr), value :: st                      ! c_loc of an evpk_state, or c_null_ptr
         end function
         integer (c_int) function evpk_prep (ctx) bind(C, name='evpk_prep')
            import :: c_int, c_ptr
            type (c_ptr), value :: ctx
         end function
         integer (c_int) function evpk_subcycle (ctx, nsub) bind(C, name='evpk_subcycle')
            import :: c_int, c_ptr, c_int32_t
            type (c_ptr), value :: ctx
            integer (c_int32_t), value :: nsub
         end function
         integer (c_int) function evpk_finish (ctx) bind(C, name='evpk_finish')
            import :: c_int, c_ptr
            type (c_ptr), value :: ctx
         end function
         integer (c_int) function evpk_download (ctx, st) bind(C, name='evpk_download')
            import :: c_int, c_ptr, evpk_state
            type (c_ptr), value :: ctx
            type (evpk_state), intent(in) :: st
         end function
         integer (c_int) function evpk_get_stats (ctx, s) bind(C, name='evpk_get_stats')
            import :: c_int, c_ptr, evpk_stats
            type (c_ptr), value :: ctx
            type (evpk_stats), intent(out) :: s
         end function
         ! principal_stress (ice_dyn_shared.F90:853) from the device-resident state of the last evp
         integer (c_int) function evpk_principal_stress (ctx, sig1, sig2) bind(C, name='evpk_principal_stress')
            import :: c_int, c_ptr
            type (c_ptr), value :: ctx, sig1, sig2
         end function
         ! page-lock a host array kept for the life of the run: moved in place over PCIe instead of staged
         integer (c_int) function evpk_pin_host (ptr, bytes) bind(C, name='evpk_pin_host')
            import :: c_int, c_ptr, c_size_t
            type (c_ptr), value :: ptr
            integer (c_size_t), value :: bytes
         end function
         integer (c_int) function evpk_unpin_host (ptr) bind(C, name='evpk_unpin_host')
            import :: c_int, c_ptr
            type (c_ptr), value :: ptr
         end function
         ! page-locked memory allocated and pinned by the driver (hipHostMalloc): c_f_pointer(out, a, shape) makes it an array
         integer (c_int) function evpk_host_alloc (bytes, out) bind(C, name='evpk_host_alloc')
            import :: c_int, c_ptr, c_size_t
            integer (c_size_t), value :: bytes
            type (c_ptr), intent(out) :: out
         end function
         integer (c_int) function evpk_host_free (ptr) bind(C, name='evpk_host_free')
            import :: c_int, c_ptr
            type (c_ptr), value :: ptr
         end function
         integer (c_int) function evpk_host_is_mapped (ptr, bytes) bind(C, name='evpk_host_is_mapped')
            import :: c_int, c_ptr, c_size_t
            type (c_ptr), value :: ptr
            integer (c_size_t), value :: bytes
         end function
         ! two-phase start (nranks > 1): evpk_create with unique_id = c_null_ptr, agree across ranks, then connect
         integer (c_int) function evpk_connect (ctx, id) bind(C, name='evpk_connect')
            import :: c_int, c_ptr
            type (c_ptr), value :: ctx, id
         end function
         integer (c_int) function evpk_device_check (device) bind(C, name='evpk_device_check')
            import :: c_int, c_int32_t
            integer (c_int32_t), value :: device
         end function
         ! the dynamics records of the binary restart (ice_restart_driver.F90:122-176, :295-412) from / into the device state
         integer (c_int) function evpk_restart_write (ctx, path, append, big_endian) bind(C, name='evpk_restart_write')
            import :: c_int, c_ptr, c_char, c_int32_t
            type (c_ptr), value :: ctx
            character (kind=c_char), dimension(*), intent(in) :: path      ! null-terminated
            integer (c_int32_t), value :: append, big_endian
         end function
         integer (c_int) function evpk_restart_read (ctx, path, byte_offset, big_endian) bind(C, name='evpk_restart_read')
            import :: c_int, c_ptr, c_char, c_int32_t, c_int64_t
            type (c_ptr), value :: ctx
            character (kind=c_char), dimension(*), intent(in) :: path
            integer (c_int64_t), value :: byte_offset
            integer (c_int32_t), value :: big_endian
         end function
         ! ice_HaloUpdate / ice_HaloUpdate_stress of a block array on the device (include/evpk.h): a(nx_block,ny_block[,nz],nblocks)
         integer (c_int) function evpk_halo_update (ctx, a, nz, field_loc, field_type, fill) bind(C, name='evpk_halo_update')
            import :: c_int, c_ptr, c_double, c_int32_t
            type (c_ptr), value :: ctx, a
            integer (c_int32_t), value :: nz, field_loc, field_type
            real (c_double), value :: fill
         end function
         integer (c_int) function evpk_halo_update_stress (ctx, a1, a2) bind(C, name='evpk_halo_update_stress')
            import :: c_int, c_ptr
            type (c_ptr), value :: ctx, a1, a2
         end function
         ! transport_upwind (ice_transport_driver.F90:634-772) on the resident velocities: works(nx_block,ny_block,narr,nblocks)
         ! as state_to_work fills it, advected in place
         integer (c_int) function evpk_transport_upwind (ctx, dt, narr, works) bind(C, name='evpk_transport_upwind')
            import :: c_int, c_ptr, c_double, c_int32_t
            type (c_ptr), value :: ctx
            real (c_double), value :: dt
            integer (c_int32_t), value :: narr
            type (c_ptr), value :: works
         end function
         ! transport_upwind whole (state_to_work, upwind_field, work_to_state, bound_state; include/evpk.h)
         integer (c_int) function evpk_transport_upwind_state (ctx, dt, ncat, ntrcr, ntrcr_dim, trcr_depend, nt_Tsfc, nt_alvl, nt_apnd, &
               nt_fbri, tr_pond_cesm, tr_pond_lvl, tr_pond_topo, Tocnfrz, aice0, aicen, vicen, vsnon, trcrn) &
               bind(C, name='evpk_transport_upwind_state')
            import :: c_int, c_ptr, c_double, c_int32_t
            type (c_ptr), value :: ctx, trcr_depend, aice0, aicen, vicen, vsnon, trcrn
            real (c_double), value :: dt, Tocnfrz
            integer (c_int32_t), value :: ncat, ntrcr, ntrcr_dim, nt_Tsfc, nt_alvl, nt_apnd, nt_fbri, tr_pond_cesm, tr_pond_lvl, tr_pond_topo
         end function
         ! horizontal_remap (ice_transport_remap.F90:309-850) on the resident velocities: dxu, dyu, hm once, then
         ! mm(nx_block,ny_block,0:ncat,max_blocks), tm(nx_block,ny_block,ntrace,ncat,max_blocks) advanced in place
         integer (c_int) function evpk_remap_init (ctx, dxu, dyu, hm) bind(C, name='evpk_remap_init')
            import :: c_int, c_ptr
            type (c_ptr), value :: ctx, dxu, dyu, hm
         end function
         integer (c_int) function evpk_transport_remap (ctx, dt, ncat, ntrace, mm, tm, tracer_type, depend, has_dependents, &
                                                        integral_order, l_dp_midpt, l_fixed_area) bind(C, name='evpk_transport_remap')
            import :: c_int, c_ptr, c_double, c_int32_t
            type (c_ptr), value :: ctx
            real (c_double), value :: dt
            integer (c_int32_t), value :: ncat, ntrace
            type (c_ptr), value :: mm, tm, tracer_type, depend, has_dependents
            integer (c_int32_t), value :: integral_order, l_dp_midpt, l_fixed_area
         end function
         ! transport_remap with state_to_tracers / tracers_to_state / bound_state on the device too: the state arrays themselves
         integer (c_int) function evpk_transport_remap_state (ctx, dt, ncat, ntrcr, ntrcr_dim, nt_qsno, nslyr, rhos_lfresh, &
               aice0, aicen, vicen, vsnon, trcrn, tracer_type, depend, has_dependents, integral_order, l_dp_midpt) &
               bind(C, name='evpk_transport_remap_state')
            import :: c_int, c_ptr, c_double, c_int32_t
            type (c_ptr), value :: ctx
            real (c_double), value :: dt, rhos_lfresh
            integer (c_int32_t), value :: ncat, ntrcr, ntrcr_dim, nt_qsno, nslyr
            type (c_ptr), value :: aice0, aicen, vicen, vsnon, trcrn, tracer_type, depend, has_dependents
            integer (c_int32_t), value :: integral_order, l_dp_midpt
         end function
         ! EAP (ice_dyn_eap.F90): tables of init_eap once, then the context runs eap(dt); structure tensor up / everything down
         integer (c_int) function evpk_eap_init (ctx, nx_yield, ny_yield, na_yield, s11r, s12r, s22r, s11s, s12s, s22s) &
               bind(C, name='evpk_eap_init')
            import :: c_int, c_ptr, c_int32_t
            type (c_ptr), value :: ctx
            integer (c_int32_t), value :: nx_yield, ny_yield, na_yield
            type (c_ptr), value :: s11r, s12r, s22r, s11s, s12s, s22s
         end function
         integer (c_int) function evpk_eap_upload (ctx, st) bind(C, name='evpk_eap_upload')
            import :: c_int, c_ptr, evpk_eap_state
            type (c_ptr), value :: ctx
            type (evpk_eap_state), intent(in) :: st
         end function
         integer (c_int) function evpk_eap_download (ctx, st) bind(C, name='evpk_eap_download')
            import :: c_int, c_ptr, evpk_eap_state
            type (c_ptr), value :: ctx
            type (evpk_eap_state), intent(in) :: st
         end function
         integer (c_int) function evpk_destroy (ctx) bind(C, name='evpk_destroy')
            import :: c_int, c_ptr
            type (c_ptr), value :: ctx
         end function
         type (c_ptr) function evpk_last_error (ctx) bind(C, name='evpk_last_error')
            import :: c_ptr
            type (c_ptr), value :: ctx
         end function
      end interface

      contains

      ! C string of evpk_last_error -> Fortran string (for abort_ice)
      function evpk_error_string (ctx) result (msg)
         type (c_ptr), intent(in) :: ctx
         character (len=512) :: msg
         type (c_ptr) :: p
         character (kind=c_char), pointer :: s(:)
         integer :: i
         msg = ' '
         p = evpk_last_error (ctx)
         if (.not. c_associated(p)) return
         call c_f_pointer (p, s, [512])
         do i = 1, 512
            if (s(i) == c_null_char) exit
            msg(i:i) = s(i)
         enddo
      end function evpk_error_string

      end module evpk_mod
