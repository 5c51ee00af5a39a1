!=======================================================================
! TEST DOUBLES, not product code and not reference code: the smallest modules
! with the names and public entities that fortran/ice_dyn_evp.F90 `use`s, so
! that the drop-in shim can be compiled and run against libevpk without a
! CICE build (the reference's own modules need netCDF, which this image lacks).
! Array sizes are run-time here (the reference fixes them with cpp at compile
! time, ice_domain_size.F90:24-66); the shim does not depend on which.
! ice_strength leaves `strength` as loaded from the fixture.
!=======================================================================
      module ice_kinds_mod
      implicit none
      integer, parameter :: char_len = 80, log_kind = kind(.true.), int_kind = selected_int_kind(6), &
                            real_kind = selected_real_kind(6), dbl_kind = selected_real_kind(13)
      end module ice_kinds_mod

      module ice_constants
      use ice_kinds_mod
      implicit none
      real (kind=dbl_kind), parameter :: c0 = 0.0_dbl_kind, c1 = 1.0_dbl_kind, &
         p001 = 0.001_dbl_kind, p01 = 0.01_dbl_kind, &
         Lfresh = 3.34e5_dbl_kind, rhos = 330.0_dbl_kind, rhoi = 917.0_dbl_kind, rhow = 1026.0_dbl_kind, gravit = 9.80616_dbl_kind
      integer (int_kind), parameter :: field_loc_center = 1, field_loc_NEcorner = 2, &
         field_type_scalar = 1, field_type_vector = 2
      end module ice_constants

      module ice_domain_size
      use ice_kinds_mod
      implicit none
      integer (int_kind) :: nx_global, ny_global, max_blocks, ncat = 1, nslyr = 1, max_ntrcr = 1
      end module ice_domain_size

      module ice_exit
      implicit none
      contains
      subroutine abort_ice (msg)
      character (*), intent(in) :: msg
      write (0,*) 'abort_ice: ', trim(msg)
      stop 1
      end subroutine abort_ice
      end module ice_exit

      module ice_communicate
      use ice_kinds_mod
      implicit none
      integer (int_kind) :: my_task = 0, master_task = 0
      contains
      integer function get_num_procs ()
      get_num_procs = 1
      end function get_num_procs
      end module ice_communicate

      module ice_timers
      use ice_kinds_mod
      implicit none
      integer (int_kind) :: timer_dynamics = 1, timer_bound = 2
      contains
      subroutine ice_timer_start (t)
      integer (int_kind), intent(in) :: t
      end subroutine
      subroutine ice_timer_stop (t)
      integer (int_kind), intent(in) :: t
      end subroutine
      end module ice_timers

      module ice_blocks
      use ice_kinds_mod
      implicit none
      type block
         integer (int_kind) :: block_id, local_id, ilo, ihi, jlo, jhi, iblock, jblock
         logical (log_kind) :: tripole, tripoleTFlag
         integer (int_kind), dimension(:), pointer :: i_glob, j_glob
      end type
      integer (int_kind) :: nx_block, ny_block
      type (block), dimension(:), allocatable, target :: all_blocks
      contains
      function get_block (block_id, local_id) result (b)
      integer (int_kind), intent(in) :: block_id, local_id
      type (block) :: b
      b = all_blocks(block_id)
      b%local_id = local_id
      end function get_block
      end module ice_blocks

      module ice_boundary
      use ice_kinds_mod
      implicit none
      type ice_halo
         integer (int_kind) :: dummy = 0
      end type
      interface ice_HaloUpdate
         module procedure halo_i4
      end interface
      contains
      ! integer centre-scalar halo through a global array (cyclic / open / tripole centre fold)
      subroutine halo_i4 (array, halo, fieldLoc, fieldKind)
      use ice_blocks, only: all_blocks, nx_block, ny_block
      use ice_domain_size, only: nx_global, ny_global
      integer (int_kind), dimension(:,:,:), intent(inout) :: array
      type (ice_halo), intent(in) :: halo
      integer (int_kind), intent(in) :: fieldLoc, fieldKind
      integer (int_kind), allocatable :: g(:,:)
      integer (int_kind) :: n, i, j, gi, gj, v
      character (len=16) :: ns
      common /mock_bnd/ ns
      allocate (g(nx_global,ny_global)); g = 0
      do n = 1, size(all_blocks)
         do j = all_blocks(n)%jlo, all_blocks(n)%jhi
         do i = all_blocks(n)%ilo, all_blocks(n)%ihi
            g(all_blocks(n)%i_glob(i), all_blocks(n)%j_glob(j)) = array(i,j,n)
         enddo
         enddo
      enddo
      do n = 1, size(all_blocks)
         do j = 1, ny_block
         do i = 1, nx_block
            if (i >= all_blocks(n)%ilo .and. i <= all_blocks(n)%ihi .and. &
                j >= all_blocks(n)%jlo .and. j <= all_blocks(n)%jhi) cycle
            gi = all_blocks(n)%i_glob(all_blocks(n)%ilo) + (i - all_blocks(n)%ilo)
            gj = all_blocks(n)%j_glob(all_blocks(n)%jlo) + (j - all_blocks(n)%jlo)
            if (gi < 1) gi = gi + nx_global
            if (gi > nx_global) gi = gi - nx_global
            v = 0
            if (gj >= 1 .and. gj <= ny_global) then
               v = g(gi,gj)
            else if (gj == ny_global+1 .and. trim(ns) == 'tripole') then
               v = g(nx_global-gi+1, ny_global)
            endif
            array(i,j,n) = v
         enddo
         enddo
      enddo
      deallocate (g)
      end subroutine halo_i4
      end module ice_boundary

      module ice_domain
      use ice_kinds_mod
      use ice_boundary, only: ice_halo
      implicit none
      integer (int_kind) :: nblocks
      integer (int_kind), dimension(:), allocatable :: blocks_ice
      type (ice_halo) :: halo_info
      character (char_len) :: ew_boundary_type, ns_boundary_type
      logical (log_kind) :: maskhalo_dyn = .false.
      end module ice_domain

      module ice_grid
      use ice_kinds_mod
      implicit none
      real (kind=dbl_kind), dimension(:,:,:), allocatable :: &
         dxt, dyt, dxhy, dyhx, cxp, cyp, cxm, cym, tarear, uarear, tinyarea, tarea, uarea, HTN, HTE, &
         dxu, dyu, hm
      logical (kind=log_kind), dimension(:,:,:), allocatable :: tmask, umask
      character (char_len) :: grid_type = 'rectangular'
      end module ice_grid

      module ice_state
      use ice_kinds_mod
      implicit none
      real (kind=dbl_kind), dimension(:,:,:), allocatable :: &
         aice, vice, vsno, aice_init, aice0, uvel, vvel, divu, shear, strength
      real (kind=dbl_kind), dimension(:,:,:,:), allocatable :: aicen, vicen, vsnon
      real (kind=dbl_kind), dimension(:,:,:,:,:), allocatable :: trcrn
      integer (kind=int_kind) :: ntrcr = 0, nt_qsno = 1
      end module ice_state

      module ice_flux
      use ice_kinds_mod
      implicit none
      real (kind=dbl_kind), dimension(:,:,:), allocatable :: &
         rdg_conv, rdg_shear, prs_sig, strairxT, strairyT, strairx, strairy, uocn, vocn, &
         ss_tltx, ss_tlty, fm, strtltx, strtlty, strocnx, strocny, strintx, strinty, &
         strocnxT, strocnyT, strax, stray, &
         stressp_1, stressp_2, stressp_3, stressp_4, stressm_1, stressm_2, stressm_3, stressm_4, &
         stress12_1, stress12_2, stress12_3, stress12_4
      logical (kind=log_kind), dimension(:,:,:), allocatable :: iceumask
      end module ice_flux

      module ice_atmo
      use ice_kinds_mod
      implicit none
      real (kind=dbl_kind), dimension(:,:,:), allocatable :: Cdn_ocn
      logical (kind=log_kind) :: calc_strair = .true.
      end module ice_atmo

      module ice_mechred
      use ice_kinds_mod
      implicit none
      ! namelist switches (ice_mechred.F90:54-64; defaults ice_init.F90:273-277)
      integer (kind=int_kind) :: kstrength = 1, krdg_partic = 1, krdg_redist = 1
      real (kind=dbl_kind) :: mu_rdg = 3.0_dbl_kind, Cf = 17.0_dbl_kind
      contains
      subroutine ice_strength (nx_block, ny_block, ilo, ihi, jlo, jhi, icells, indxi, indxj, &
                               aice, vice, aice0, aicen, vicen, strength)
      integer (kind=int_kind), intent(in) :: nx_block, ny_block, ilo, ihi, jlo, jhi, icells
      integer (kind=int_kind), dimension (nx_block*ny_block), intent(in) :: indxi, indxj
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(in) :: aice, vice, aice0
      real (kind=dbl_kind), dimension (:,:,:), intent(in) :: aicen, vicen
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(inout) :: strength
      ! test double: strength is part of the fixture
      end subroutine ice_strength
      end module ice_mechred

      ! AusCOM coupling modules (drivers/auscom/cpl_parameters.F90, cpl_arrays_setup.F90): only what evp touches
      module cpl_parameters
      use ice_kinds_mod
      implicit none
      logical (kind=log_kind) :: use_ocnslope = .false., use_umask = .false.
      end module cpl_parameters

      module cpl_arrays_setup
      use ice_kinds_mod
      implicit none
      real (kind=dbl_kind), dimension(:,:,:), allocatable :: sicemass
      end module cpl_arrays_setup

      module ice_dyn_shared
      use ice_kinds_mod
      implicit none
      integer (kind=int_kind) :: kdyn = 1, ndte = 120
      logical (kind=log_kind) :: revised_evp = .false.
      real (kind=dbl_kind) :: cosw = 1.0_dbl_kind, sinw = 0.0_dbl_kind, &
         revp, ecci, dtei, dte2T, denom1, arlx1i, brlx
      real (kind=dbl_kind), allocatable :: fcor_blk(:,:,:), uvel_init(:,:,:), vvel_init(:,:,:)
      contains
      ! own few-line version of the T-cell mask step (3x3 dilation of the ice mask, cleared on land)
      subroutine evp_prep1 (nx_block, ny_block, ilo, ihi, jlo, jhi, aice, vice, vsno, tmask, &
                            strairxT, strairyT, strairx, strairy, tmass, icetmask)
      use ice_constants, only: rhoi, rhos, p001, p01
      integer (kind=int_kind), intent(in) :: nx_block, ny_block, ilo, ihi, jlo, jhi
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(in) :: aice, vice, vsno, strairxT, strairyT
      logical (kind=log_kind), dimension (nx_block,ny_block), intent(in) :: tmask
      real (kind=dbl_kind), dimension (nx_block,ny_block), intent(out) :: strairx, strairy, tmass
      integer (kind=int_kind), dimension (nx_block,ny_block), intent(out) :: icetmask
      logical (kind=log_kind) :: icy(nx_block,ny_block)
      integer (kind=int_kind) :: i, j
      tmass = merge(rhoi*vice + rhos*vsno, 0.0_dbl_kind, tmask)
      icy = tmask .and. aice > p001 .and. tmass > p01
      strairx = strairxT
      strairy = strairyT
      icetmask = 0
      do j = jlo, jhi
      do i = ilo, ihi
         if (any(icy(i-1:i+1,j-1:j+1)) .and. tmask(i,j)) icetmask(i,j) = 1
      enddo
      enddo
      end subroutine evp_prep1
      end module ice_dyn_shared
