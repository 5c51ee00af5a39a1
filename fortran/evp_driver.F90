!=======================================================================
! evp_driver -- stand-alone Fortran host for the drop-in shim: loads a binary
! fixture (written by tests/test_fortran_gpu.py) into the module arrays, calls
!     call evp (dt)
! exactly as ice_step_mod.F90:1119 does, and writes the arrays evp leaves
! modified.  Used on the GPU box to prove the Fortran -> ISO_C_BINDING ->
! libevpk -> HIP path end to end (BASELINE config 1 plumbing).
!   usage: evp_driver <fixture.bin> <out.bin>
!=======================================================================
      program evp_driver

      use ice_kinds_mod
      use ice_blocks
      use ice_domain
      use ice_domain_size
      use ice_grid
      use ice_state
      use ice_flux
      use ice_atmo
      use ice_dyn_shared
      use ice_dyn_evp, only: evp, evpk_npinned, evpk_device_strength, evpk_eap, evpk_horizontal_remap
      use ice_mechred, only: kstrength, krdg_partic, krdg_redist
#ifdef AusCOM
      use cpl_arrays_setup, only: sicemass
#endif

      implicit none
      character (len=512) :: fin, fout
      integer (int_kind) :: hdr(10), trl(6), ios, n, i, nb, ncalls, call_no, ew, ns
      real (dbl_kind) :: sc(8), dt
      integer (int_kind), allocatable :: geo(:,:), itmp(:,:,:)
      ! kdyn = 2: what the reference keeps in module ice_dyn_eap (tables, structure tensor, history fields)
      logical :: run_eap = .false.
      ! transport_remap's horizontal_remap after the last evp: aim, trm and the tables of init_transport
      logical :: run_remap = .false.
      integer (int_kind) :: rm_ntrace, rm_order, rm_midpt
      real (dbl_kind) :: rm_dt(1)
      real (dbl_kind), allocatable :: aim(:,:,:,:), trm(:,:,:,:,:)
      integer (int_kind), allocatable :: rm_type(:), rm_dep(:), rm_hasi(:)
      logical (log_kind), allocatable :: rm_has(:)
      integer (int_kind) :: nxy, nyy, nay
      real (dbl_kind), allocatable, dimension(:,:,:) :: s11r, s12r, s22r, s11s, s12s, s22s, &
         a11_1, a11_2, a11_3, a11_4, a12_1, a12_2, a12_3, a12_4, a11, a12, e11, e12, e22, ys11, ys12, ys22, s11, s12, s22
      character (len=16) :: nsname
      common /mock_bnd/ nsname

      call get_command_argument (1, fin)
      call get_command_argument (2, fout)
      open (10, file=trim(fin), access='stream', form='unformatted', status='old')
      read (10) hdr
      read (10) sc
      nx_global = hdr(1); ny_global = hdr(2); nx_block = hdr(3); ny_block = hdr(4); nb = hdr(5)
      ew = hdr(6); ns = hdr(7); ndte = hdr(8); revised_evp = hdr(9) /= 0; ncalls = hdr(10)
      dt = sc(1); revp = sc(2); ecci = sc(3); denom1 = sc(4); arlx1i = sc(5); brlx = sc(6); cosw = sc(7); sinw = sc(8)
      nblocks = nb; max_blocks = nb
      ew_boundary_type = 'cyclic'; if (ew == 1) ew_boundary_type = 'open'; if (ew == 2) ew_boundary_type = 'closed'
      ns_boundary_type = 'open'; if (ns == 2) ns_boundary_type = 'closed'; if (ns == 3) ns_boundary_type = 'tripole'
      nsname = ns_boundary_type

      allocate (geo(nb,6), all_blocks(nb), blocks_ice(nb))
      read (10) geo
      do n = 1, nb
         blocks_ice(n) = n
         all_blocks(n)%block_id = n
         all_blocks(n)%ilo = geo(n,1); all_blocks(n)%ihi = geo(n,2)
         all_blocks(n)%jlo = geo(n,3); all_blocks(n)%jhi = geo(n,4)
         allocate (all_blocks(n)%i_glob(nx_block), all_blocks(n)%j_glob(ny_block))
         do i = 1, nx_block
            all_blocks(n)%i_glob(i) = geo(n,5) + (i - geo(n,1))
         enddo
         do i = 1, ny_block
            all_blocks(n)%j_glob(i) = geo(n,6) + (i - geo(n,3))
         enddo
         all_blocks(n)%tripole = (ns == 3 .and. all_blocks(n)%j_glob(geo(n,4)) == ny_global)
      enddo

      call rd (dxt); call rd (dyt); call rd (dxhy); call rd (dyhx); call rd (cxp); call rd (cyp)
      call rd (cxm); call rd (cym); call rd (tarear); call rd (uarear); call rd (tinyarea)
      call rd (tarea); call rd (uarea); call rd (fcor_blk); call rd (HTN); call rd (HTE)
      call rd (aice); call rd (vice); call rd (vsno); call rd (aice_init)
      call rd (strairxT); call rd (strairyT); call rd (strax); call rd (stray)
      call rd (uocn); call rd (vocn); call rd (ss_tltx); call rd (ss_tlty); call rd (Cdn_ocn); call rd (strength)
      call rd (uvel); call rd (vvel)
      call rd (stressp_1); call rd (stressp_2); call rd (stressp_3); call rd (stressp_4)
      call rd (stressm_1); call rd (stressm_2); call rd (stressm_3); call rd (stressm_4)
      call rd (stress12_1); call rd (stress12_2); call rd (stress12_3); call rd (stress12_4)
      allocate (itmp(nx_block,ny_block,nb), tmask(nx_block,ny_block,nb), umask(nx_block,ny_block,nb), &
                iceumask(nx_block,ny_block,nb))
      read (10) itmp; tmask = itmp /= 0
      read (10) itmp; umask = itmp /= 0
      read (10) itmp; iceumask = itmp /= 0
      ! optional trailer: ice_strength on the device (thickness distribution + the ice_mechred switches)
      trl = 0
      read (10, iostat=ios) trl
      if (ios /= 0) trl = 0
      call zr (aice0)
      if (trl(1) == 1) then
         evpk_device_strength = .true.
         ncat = trl(2); kstrength = trl(3); krdg_partic = trl(4); krdg_redist = trl(5)
         allocate (aicen(nx_block,ny_block,ncat,nb), vicen(nx_block,ny_block,ncat,nb))
         read (10) aicen
         read (10) vicen
         read (10) aice0
      else
         allocate (aicen(nx_block,ny_block,1,nb), vicen(nx_block,ny_block,1,nb))
         aicen = 0.0_dbl_kind; vicen = 0.0_dbl_kind
      endif
      if (trl(1) == 2) then          ! EAP: table extents, the six tables, a11_1..4, a12_1..4
         run_eap = .true.
         nxy = trl(2); nyy = trl(3); nay = trl(4)
         allocate (s11r(nxy,nyy,nay), s12r(nxy,nyy,nay), s22r(nxy,nyy,nay), s11s(nxy,nyy,nay), s12s(nxy,nyy,nay), s22s(nxy,nyy,nay))
         read (10) s11r, s12r, s22r, s11s, s12s, s22s
         call rd (a11_1); call rd (a11_2); call rd (a11_3); call rd (a11_4)
         call rd (a12_1); call rd (a12_2); call rd (a12_3); call rd (a12_4)
         call zr (a11); call zr (a12); call zr (e11); call zr (e12); call zr (e22)
         call zr (ys11); call zr (ys12); call zr (ys22); call zr (s11); call zr (s12); call zr (s22)
      endif
      if (trl(1) == 3) then          ! horizontal_remap: ncat, ntrace, integral_order, l_dp_midpt; dt; dxu, dyu, hm; tables; aim; trm
         run_remap = .true.
         ncat = trl(2); rm_ntrace = trl(3); rm_order = trl(4); rm_midpt = trl(5)
         read (10) rm_dt
         call rd (dxu); call rd (dyu); call rd (hm)
         allocate (rm_type(rm_ntrace), rm_dep(rm_ntrace), rm_hasi(rm_ntrace), rm_has(rm_ntrace))
         read (10) rm_type, rm_dep, rm_hasi
         rm_has = rm_hasi /= 0
         allocate (aim(nx_block,ny_block,0:ncat,nb), trm(nx_block,ny_block,rm_ntrace,ncat,nb))
         read (10) aim
         read (10) trm
      endif
      close (10)

      call zr (divu); call zr (shear); call zr (rdg_conv); call zr (rdg_shear); call zr (prs_sig)
      call zr (strintx); call zr (strinty); call zr (strocnx); call zr (strocny)
      call zr (strocnxT); call zr (strocnyT); call zr (strairx); call zr (strairy)
      call zr (strtltx); call zr (strtlty); call zr (fm); call zr (uvel_init); call zr (vvel_init)

#ifdef AusCOM
      allocate (sicemass(nx_block,ny_block,nb))     ! drivers/auscom/CICE_InitMod.F90 allocates it in the real model
#endif
      do call_no = 1, ncalls
         if (run_eap) then           ! what the body of the reference's `subroutine eap (dt)` becomes (INTEGRATION.md S3)
            call evpk_eap (dt, nxy, nyy, nay, s11r, s12r, s22r, s11s, s12s, s22s, &
                           a11_1, a11_2, a11_3, a11_4, a12_1, a12_2, a12_3, a12_4, a11, a12, &
                           e11, e12, e22, ys11, ys12, ys22, s11, s12, s22)
         else
            call evp (dt)
         endif
      enddo
#ifdef AusCOM
      write (*,'(a,es12.5)') 'evp_driver: AusCOM sicemass max = ', maxval(sicemass)
#endif
      ! the call of ice_transport_driver.F90:475-481, with the shim's horizontal_remap (INTEGRATION.md S3)
      if (run_remap) call evpk_horizontal_remap (rm_dt(1), rm_ntrace, uvel, vvel, aim, trm, .false., rm_type, rm_dep, rm_has, &
                                                 rm_order, rm_midpt /= 0)

      open (11, file=trim(fout), access='stream', form='unformatted', status='replace')
      write (11) uvel, vvel, stressp_1, stressp_2, stressp_3, stressp_4, stressm_1, stressm_2, stressm_3, stressm_4, &
                 stress12_1, stress12_2, stress12_3, stress12_4, divu, shear, rdg_conv, rdg_shear, prs_sig, &
                 strintx, strinty, strocnx, strocny, strocnxT, strocnyT, strairx, strairy, strtltx, strtlty, fm, &
                 uvel_init, vvel_init
      itmp = 0
      where (iceumask) itmp = 1
      write (11) itmp
      if (evpk_device_strength) write (11) strength
      if (run_remap) write (11) aim, trm
      if (run_eap) write (11) a11_1, a11_2, a11_3, a11_4, a12_1, a12_2, a12_3, a12_4, a11, a12, e11, e12, e22, ys11, ys12, ys22, s11, s12, s22
      close (11)
      write (*,'(a,i0)') 'evp_driver: page-locked host arrays = ', evpk_npinned
      write (*,'(a,i0,a,i0,a,es12.5)') 'evp_driver: ', ncalls, ' call(s) of evp(dt) on ', nb, ' block(s); max |uvel| = ', maxval(abs(uvel))

      contains

      subroutine rd (a)
      real (dbl_kind), allocatable, intent(inout) :: a(:,:,:)
      allocate (a(nx_block,ny_block,nb))
      read (10) a
      end subroutine rd

      subroutine zr (a)
      real (dbl_kind), allocatable, intent(inout) :: a(:,:,:)
      allocate (a(nx_block,ny_block,nb))
      a = 0.0_dbl_kind
      end subroutine zr

      end program evp_driver
