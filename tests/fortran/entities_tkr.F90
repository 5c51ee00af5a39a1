! Type / kind / rank of every entity fortran/ice_dyn_evp.F90 imports from the host model.  Compiled twice by
! tests/test_ref_interfaces.py: against the reference's real modules and against the test doubles of fortran/mock/ --
! if both compile, the doubles declare each entity exactly as the reference does (generic resolution below accepts one
! type, kind and rank per name only), so what the stand-alone driver tests against them carries over to a CICE build.
      module evpk_tkr_checks
      use ice_kinds_mod
      implicit none
      contains
      subroutine r8_0 (a);  real (kind=dbl_kind), intent(in) :: a;  end subroutine
      subroutine r8_3 (a);  real (kind=dbl_kind), dimension(:,:,:), intent(in) :: a;  end subroutine
      subroutine r8_4 (a);  real (kind=dbl_kind), dimension(:,:,:,:), intent(in) :: a;  end subroutine
      subroutine r8_5 (a);  real (kind=dbl_kind), dimension(:,:,:,:,:), intent(in) :: a;  end subroutine
      subroutine l_0 (a);   logical (kind=log_kind), intent(in) :: a;  end subroutine
      subroutine l_3 (a);   logical (kind=log_kind), dimension(:,:,:), intent(in) :: a;  end subroutine
      subroutine i4_0 (a);  integer (kind=int_kind), intent(in) :: a;  end subroutine
      subroutine i4_1 (a);  integer (kind=int_kind), dimension(:), intent(in) :: a;  end subroutine
      subroutine ch_0 (a);  character (*), intent(in) :: a;  end subroutine
      end module evpk_tkr_checks

      subroutine evpk_entities_tkr
      use ice_kinds_mod
      use evpk_tkr_checks
      use ice_atmo, only: Cdn_ocn
#ifdef ACCESS
      use ice_atmo, only: calc_strair
#endif
      use ice_boundary, only: ice_HaloUpdate
      use ice_blocks, only: block, get_block, nx_block, ny_block
      use ice_communicate, only: my_task, master_task, get_num_procs
      use ice_constants, only: field_loc_center, field_type_scalar, c0, rhow, rhoi, rhos, gravit, p001, p01, Lfresh
      use ice_domain, only: nblocks, blocks_ice, halo_info, ew_boundary_type, ns_boundary_type
      use ice_domain_size, only: max_blocks, nx_global, ny_global, ncat, nslyr, max_ntrcr
      use ice_exit, only: abort_ice
      use ice_flux, only: rdg_conv, rdg_shear, prs_sig, strairxT, strairyT, strairx, strairy, uocn, vocn, &
          ss_tltx, ss_tlty, iceumask, fm, strtltx, strtlty, strocnx, strocny, strintx, strinty, strocnxT, strocnyT, &
          strax, stray, stressp_1, stressp_2, stressp_3, stressp_4, stressm_1, stressm_2, stressm_3, stressm_4, &
          stress12_1, stress12_2, stress12_3, stress12_4
      use ice_grid, only: tmask, umask, dxt, dyt, dxhy, dyhx, cxp, cyp, cxm, cym, tarear, uarear, tinyarea, tarea, uarea, HTN, HTE, dxu, dyu, hm
      use ice_mechred, only: ice_strength, kstrength, krdg_partic, krdg_redist, mu_rdg, Cf
      use ice_state, only: aice, vice, vsno, uvel, vvel, divu, shear, aice_init, aice0, aicen, vicen, strength, &
          vsnon, trcrn, ntrcr, nt_qsno
      use ice_timers, only: timer_dynamics, timer_bound, ice_timer_start, ice_timer_stop
      use ice_dyn_shared, only: ndte, revised_evp, revp, ecci, denom1, arlx1i, brlx, cosw, sinw, fcor_blk, &
          uvel_init, vvel_init, evp_prep1, kdyn
#ifdef AusCOM
      use cpl_parameters, only: use_ocnslope
      use cpl_arrays_setup, only: sicemass
#endif
      implicit none
      type (block) :: this_block
      integer (kind=int_kind), dimension (4) :: ix
      integer (kind=int_kind), dimension (:,:,:), allocatable :: icetmask
      real (kind=dbl_kind), dimension (:,:,:), allocatable :: tmass

      this_block = get_block(blocks_ice(1), 1)
      call i4_0 (this_block%ilo);  call i4_0 (this_block%ihi);  call i4_0 (this_block%jlo);  call i4_0 (this_block%jhi)
      call i4_1 (this_block%i_glob);  call i4_1 (this_block%j_glob)
      call i4_0 (nx_block);  call i4_0 (ny_block);  call i4_0 (nblocks);  call i4_1 (blocks_ice)
      call i4_0 (max_blocks);  call i4_0 (nx_global);  call i4_0 (ny_global);  call i4_0 (ncat)
      call i4_0 (my_task);  call i4_0 (master_task);  call i4_0 (get_num_procs())
      call ch_0 (ew_boundary_type);  call ch_0 (ns_boundary_type)
      call i4_0 (field_loc_center);  call i4_0 (field_type_scalar)
      call r8_0 (c0);  call r8_0 (rhow);  call r8_0 (rhoi);  call r8_0 (rhos);  call r8_0 (gravit);  call r8_0 (p001);  call r8_0 (p01)
      call r8_3 (Cdn_ocn)
#ifdef ACCESS
      call l_0 (calc_strair)
#endif
      call r8_3 (rdg_conv);  call r8_3 (rdg_shear);  call r8_3 (prs_sig);  call r8_3 (strairxT);  call r8_3 (strairyT)
      call r8_3 (strairx);  call r8_3 (strairy);  call r8_3 (uocn);  call r8_3 (vocn);  call r8_3 (ss_tltx);  call r8_3 (ss_tlty)
      call l_3 (iceumask);  call r8_3 (fm);  call r8_3 (strtltx);  call r8_3 (strtlty);  call r8_3 (strocnx);  call r8_3 (strocny)
      call r8_3 (strintx);  call r8_3 (strinty);  call r8_3 (strocnxT);  call r8_3 (strocnyT);  call r8_3 (strax);  call r8_3 (stray)
      call r8_3 (stressp_1);  call r8_3 (stressp_2);  call r8_3 (stressp_3);  call r8_3 (stressp_4)
      call r8_3 (stressm_1);  call r8_3 (stressm_2);  call r8_3 (stressm_3);  call r8_3 (stressm_4)
      call r8_3 (stress12_1);  call r8_3 (stress12_2);  call r8_3 (stress12_3);  call r8_3 (stress12_4)
      call l_3 (tmask);  call l_3 (umask)
      call r8_3 (dxt);  call r8_3 (dyt);  call r8_3 (dxhy);  call r8_3 (dyhx);  call r8_3 (cxp);  call r8_3 (cyp);  call r8_3 (cxm);  call r8_3 (cym)
      call r8_3 (tarear);  call r8_3 (uarear);  call r8_3 (tinyarea);  call r8_3 (tarea);  call r8_3 (uarea);  call r8_3 (HTN);  call r8_3 (HTE)
      call r8_3 (dxu);  call r8_3 (dyu);  call r8_3 (hm)
      call i4_0 (kstrength);  call i4_0 (krdg_partic);  call i4_0 (krdg_redist);  call r8_0 (mu_rdg);  call r8_0 (Cf)
      call r8_3 (aice);  call r8_3 (vice);  call r8_3 (vsno);  call r8_3 (uvel);  call r8_3 (vvel);  call r8_3 (divu);  call r8_3 (shear)
      call r8_3 (aice_init);  call r8_3 (aice0);  call r8_4 (aicen);  call r8_4 (vicen);  call r8_3 (strength)
      call r8_4 (vsnon);  call r8_5 (trcrn);  call i4_0 (ntrcr);  call i4_0 (nt_qsno);  call i4_0 (nslyr);  call i4_0 (max_ntrcr);  call r8_0 (Lfresh)
      call i4_0 (timer_dynamics);  call i4_0 (timer_bound)
      call i4_0 (ndte);  call i4_0 (kdyn);  call l_0 (revised_evp)
      call r8_0 (revp);  call r8_0 (ecci);  call r8_0 (denom1);  call r8_0 (arlx1i);  call r8_0 (brlx);  call r8_0 (cosw);  call r8_0 (sinw)
      call r8_3 (fcor_blk);  call r8_3 (uvel_init);  call r8_3 (vvel_init)
#ifdef AusCOM
      call l_0 (use_ocnslope);  call r8_3 (sicemass)
#endif
      ! procedures, with the argument lists the shim uses
      allocate (icetmask(nx_block,ny_block,max_blocks), tmass(nx_block,ny_block,max_blocks))
      call ice_timer_start (timer_bound)
      call ice_HaloUpdate (icetmask, halo_info, field_loc_center, field_type_scalar)
      call ice_timer_stop (timer_bound)
      call evp_prep1 (nx_block, ny_block, 2, 3, 2, 3, aice(:,:,1), vice(:,:,1), vsno(:,:,1), tmask(:,:,1), &
                      strairxT(:,:,1), strairyT(:,:,1), strairx(:,:,1), strairy(:,:,1), tmass(:,:,1), icetmask(:,:,1))
      ix = 1
      call ice_strength (nx_block, ny_block, 2, 3, 2, 3, 1, ix, ix, aice(:,:,1), vice(:,:,1), aice0(:,:,1), &
                         aicen(:,:,:,1), vicen(:,:,:,1), strength(:,:,1))
      call abort_ice ('never called')
      end subroutine evpk_entities_tkr
