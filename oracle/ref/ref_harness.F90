!=======================================================================
! ref_harness -- test infrastructure (ours, not reference code): a small driver that calls the REFERENCE'S OWN routines,
! compiled unmodified from /root/reference by oracle/ref/Makefile, on inputs read from a binary file and dumps what they
! return.  tests/golden/make_ref_golden.py turns the dumps into the fixtures tests/golden/ref_*.npz that pin the C
! restatement (oracle/) and the HIP kernels.
!
! What it reaches (the part of SURVEY.md S8's path whose module closure compiles here without netCDF / MPI):
!   init_domain_blocks, init_domain_distribution   source/ice_domain.F90:84, :248  (create_blocks ice_blocks.F90:111,
!                                                   create_distribution ice_distribution.F90:535, ice_HaloCreate)
!   ice_HaloUpdate 2DR8 / 3DR8 / 2DI4               serial/ice_boundary.F90:630, :1440, :1170
!   ice_HaloUpdate_stress                           serial/ice_boundary.F90:3269
!   bound_state                                     source/ice_state.F90:173
!   ice_strength (asum_ridging, ridge_itd)          source/ice_mechred.F90:2111
!   create_distribution('cartesian', nprocs > 1)    source/ice_distribution.F90:535
!   global_minval (scalar, masked)                  serial/ice_global_reductions.F90 (set_evp_parameters' xmin, ymin)
!   compute_tracers                                 source/ice_itd.F90:1359 (work_to_state of transport_upwind)
!
! Usage:  ref_harness <in.bin> <out.bin>   with the namelist file cice_in.nml (domain_nml) in the working directory.
! Both files are native-endian streams of int32 / real64; the layout is the sequence of reads / writes below.
!=======================================================================
program ref_harness

   use ice_kinds_mod
   use ice_communicate, only: init_communicate, my_task, master_task
   use ice_fileunits, only: init_fileunits, nu_diag, ice_stdout
   use ice_domain_size, only: nx_global, ny_global, max_blocks, ncat, max_ntrcr
   use ice_blocks, only: block, get_block, nblocks_tot, nblocks_x, nblocks_y, nx_block, ny_block
   use ice_distribution, only: distrb, create_distribution, processor_shape
   use ice_domain, only: init_domain_blocks, init_domain_distribution, nblocks, blocks_ice, distrb_info, halo_info
   use ice_boundary, only: ice_HaloUpdate, ice_HaloUpdate_stress
   use ice_state, only: bound_state, ntrcr, nt_Tsfc, nt_alvl, nt_apnd, nt_fbri, tr_pond_cesm, tr_pond_lvl, tr_pond_topo
   use ice_itd, only: compute_tracers
   use ice_mechred, only: ice_strength, kstrength, krdg_partic, krdg_redist, mu_rdg, Cf
   use ice_global_reductions, only: global_minval
   use ice_constants, only: field_loc_center, Tocnfrz

   implicit none

   integer, parameter :: uin = 201, uout = 202
   character (len=512) :: fin, fout
   integer (int_kind) :: op, n, nz, loc, ftype, hasfill, ifill, np, ic, lenshape, iblk, k
   real (dbl_kind) :: fill, xmin
   real (dbl_kind), allocatable :: kmtg(:,:), ulatg(:,:)
   real (dbl_kind), allocatable :: a2(:,:,:), b2(:,:,:), a3(:,:,:,:)
   integer (int_kind), allocatable :: i2(:,:,:), work(:), indxi(:), indxj(:)
   real (dbl_kind), allocatable :: aicen(:,:,:,:), vicen(:,:,:,:), vsnon(:,:,:,:), trcrn(:,:,:,:,:)
   real (dbl_kind), allocatable :: s_aice(:,:), s_vice(:,:), s_aice0(:,:), s_aicen(:,:,:), s_vicen(:,:,:), s_str(:,:)
   logical (log_kind), allocatable :: lmask(:,:,:)
   integer (int_kind) :: nt, pond(3)
   integer (int_kind), allocatable :: tdep(:)
   real (dbl_kind), allocatable :: atr(:,:), c_a(:,:), c_v(:,:), c_s(:,:), c_t(:,:,:)
   type (block) :: b
   type (distrb) :: d
   character (len=16) :: shape

   call get_command_argument(1, fin)
   call get_command_argument(2, fout)
   open (uin,  file=trim(fin),  access='stream', form='unformatted', status='old')
   open (uout, file=trim(fout), access='stream', form='unformatted', status='replace')

   call init_communicate
   call init_fileunits
   nu_diag = ice_stdout

   ! ---- domain: the reference's own set-up sequence (CICE_InitMod: init_domain_blocks, init_grid1 -> init_domain_distribution)
   call init_domain_blocks
   allocate (kmtg(nx_global,ny_global), ulatg(nx_global,ny_global))
   read (uin) kmtg
   read (uin) ulatg
   call init_domain_distribution(kmtg, ulatg)

   write (uout) nx_global, ny_global, nx_block, ny_block, max_blocks, ncat, max_ntrcr
   write (uout) nblocks_tot, nblocks_x, nblocks_y, nblocks
   do n = 1, nblocks_tot
      b = get_block(n, n)
      write (uout) b%block_id, b%iblock, b%jblock, b%ilo, b%ihi, b%jlo, b%jhi, merge(1, 0, b%tripole)
      write (uout) b%i_glob(1:nx_block)
      write (uout) b%j_glob(1:ny_block)
   enddo
   if (nblocks > 0) write (uout) blocks_ice(1:nblocks)
   write (uout) distrb_info%blockLocation(1:nblocks_tot)
   write (uout) distrb_info%blockLocalID(1:nblocks_tot)

   ! ---- operations
   do
      read (uin) op
      select case (op)
      case (0)
         exit

      case (1)          ! ice_HaloUpdate, real(8), 2-D (nz = 0) or 3-D (nz > 0)
         read (uin) nz, loc, ftype, hasfill, fill
         if (nz == 0) then
            allocate (a2(nx_block,ny_block,nblocks))
            read (uin) a2
            if (hasfill /= 0) then
               call ice_HaloUpdate(a2, halo_info, loc, ftype, fill)
            else
               call ice_HaloUpdate(a2, halo_info, loc, ftype)
            endif
            write (uout) a2
            deallocate (a2)
         else
            allocate (a3(nx_block,ny_block,nz,nblocks))
            read (uin) a3
            if (hasfill /= 0) then
               call ice_HaloUpdate(a3, halo_info, loc, ftype, fill)
            else
               call ice_HaloUpdate(a3, halo_info, loc, ftype)
            endif
            write (uout) a3
            deallocate (a3)
         endif

      case (2)          ! ice_HaloUpdate, integer(4), 2-D
         read (uin) loc, ftype, hasfill, ifill
         allocate (i2(nx_block,ny_block,nblocks))
         read (uin) i2
         if (hasfill /= 0) then
            call ice_HaloUpdate(i2, halo_info, loc, ftype, ifill)
         else
            call ice_HaloUpdate(i2, halo_info, loc, ftype)
         endif
         write (uout) i2
         deallocate (i2)

      case (3)          ! ice_HaloUpdate_stress(array1, array2, ...)
         read (uin) loc, ftype
         allocate (a2(nx_block,ny_block,nblocks), b2(nx_block,ny_block,nblocks))
         read (uin) a2
         read (uin) b2
         call ice_HaloUpdate_stress(a2, b2, halo_info, loc, ftype)
         write (uout) a2
         deallocate (a2, b2)

      case (4)          ! bound_state
         read (uin) ntrcr
         allocate (aicen(nx_block,ny_block,ncat,max_blocks), vicen(nx_block,ny_block,ncat,max_blocks), &
                   vsnon(nx_block,ny_block,ncat,max_blocks), trcrn(nx_block,ny_block,max_ntrcr,ncat,max_blocks))
         read (uin) aicen
         read (uin) vicen
         read (uin) vsnon
         read (uin) trcrn
         call bound_state(aicen, trcrn, vicen, vsnon)
         write (uout) aicen
         write (uout) vicen
         write (uout) vsnon
         write (uout) trcrn
         deallocate (aicen, vicen, vsnon, trcrn)

      case (5)          ! create_distribution('cartesian', nprocs, work_per_block) for another processor count / shape
         read (uin) np, lenshape
         shape = ' '
         read (uin) shape(1:lenshape)
         allocate (work(nblocks_tot))
         read (uin) work
         processor_shape = shape
         d = create_distribution('cartesian', np, work)
         write (uout) d%blockLocation(1:nblocks_tot)
         write (uout) d%blockLocalID(1:nblocks_tot)
         deallocate (work)

      case (6)          ! ice_strength on one block
         read (uin) kstrength, krdg_partic, krdg_redist, mu_rdg, Cf
         allocate (s_aice(nx_block,ny_block), s_vice(nx_block,ny_block), s_aice0(nx_block,ny_block), &
                   s_aicen(nx_block,ny_block,ncat), s_vicen(nx_block,ny_block,ncat), s_str(nx_block,ny_block), &
                   indxi(nx_block*ny_block), indxj(nx_block*ny_block))
         read (uin) b%ilo, b%ihi, b%jlo, b%jhi, ic
         read (uin) indxi
         read (uin) indxj
         read (uin) s_aice
         read (uin) s_vice
         read (uin) s_aice0
         read (uin) s_aicen
         read (uin) s_vicen
         call ice_strength(nx_block, ny_block, b%ilo, b%ihi, b%jlo, b%jhi, ic, indxi, indxj, &
                           s_aice, s_vice, s_aice0, s_aicen, s_vicen, s_str)
         write (uout) s_str
         deallocate (s_aice, s_vice, s_aice0, s_aicen, s_vicen, s_str, indxi, indxj)

      case (7)          ! global_minval(array, distrb_info, lmask): what set_evp_parameters reduces dxt / dyt with
         allocate (a2(nx_block,ny_block,max_blocks), i2(nx_block,ny_block,max_blocks), lmask(nx_block,ny_block,max_blocks))
         a2 = 0; i2 = 0
         read (uin) a2(:,:,1:nblocks)
         read (uin) i2(:,:,1:nblocks)
         lmask = (i2 /= 0)
         xmin = global_minval(a2, distrb_info, lmask)
         write (uout) xmin
         deallocate (a2, i2, lmask)

      case (8)          ! compute_tracers (ice_itd.F90:1359) on one block: work_to_state's inverse of state_to_work
         read (uin) nt
         allocate (tdep(nt))
         read (uin) tdep
         read (uin) nt_Tsfc, nt_alvl, nt_apnd, nt_fbri, pond
         read (uin) Tocnfrz            ! (AusCOM: a namelist variable, drivers/auscom/ice_constants.F90:103)
         tr_pond_cesm = pond(1) /= 0; tr_pond_lvl = pond(2) /= 0; tr_pond_topo = pond(3) /= 0
         read (uin) ic
         allocate (indxi(nx_block*ny_block), indxj(nx_block*ny_block), atr(ic,nt), c_a(nx_block,ny_block), c_v(nx_block,ny_block), &
                   c_s(nx_block,ny_block), c_t(nx_block,ny_block,nt))
         read (uin) indxi
         read (uin) indxj
         read (uin) atr
         read (uin) c_a
         read (uin) c_v
         read (uin) c_s
         call compute_tracers(nx_block, ny_block, ic, indxi, indxj, nt, tdep, atr, c_a, c_v, c_s, c_t)
         write (uout) c_t
         deallocate (tdep, indxi, indxj, atr, c_a, c_v, c_s, c_t)

      case default
         write (*,*) 'ref_harness: unknown op ', op
         stop 2
      end select
   enddo

   close (uin)
   close (uout)

end program ref_harness
