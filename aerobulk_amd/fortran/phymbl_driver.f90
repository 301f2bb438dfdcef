! phymbl_driver.f90 -- calls every public function of `mod_phymbl` on columns of numbers read from a file and writes the results.
!
! Own source (it only USEs the public interface of mod_const / mod_phymbl and the two helper functions of mod_blk_ice_an05).  Built twice: against this repository's modules
! (aerobulk_amd/build.py -> phymbl_driver.x: Fortran host -> C ABI -> HIP kernels) and, in the build container, against the
! UNMODIFIED reference modules (oracle/Makefile -> oracle/_ref/ref_phymbl_driver.x), whose output is the golden data of
! tests/test_phymbl.py (tools/gen_phymbl_golden.py).
!
!   phymbl_driver.x <in.bin> <out.bin>
!   in : int32 n ; 34 columns of n doubles (order below)
!   out: records { character(24) name ; int32 m ; m doubles }.  `_s` records: the scalar specific on the first min(n,8) cells.
PROGRAM phymbl_driver

   USE mod_const
   USE mod_phymbl
   USE mod_blk_ice_an05, ONLY: rough_leng_m, rough_leng_tq     ! the two PUBLIC helper functions of the sea-ice module (ice/test_ice.f90 calls them)
   !! the PUBLIC functions of the algorithm modules (src/tests/test_psi_stab.f90:25-28 imports the psi's)
   USE mod_common_coare,  ONLY: psi_m_coare,   psi_h_coare, first_guess_coare
   USE mod_blk_ncar,      ONLY: psi_m_ncar,    psi_h_ncar, cd_n10_ncar, ch_n10_ncar, ce_n10_ncar
   USE mod_blk_ecmwf,     ONLY: psi_m_ecmwf,   psi_h_ecmwf
   USE mod_blk_andreas,   ONLY: psi_m_andreas, psi_h_andreas, u_star_andreas
   USE mod_blk_coare3p0,  ONLY: charn_coare3p0
   USE mod_blk_coare3p6,  ONLY: charn_coare3p6

   IMPLICIT NONE

   INTEGER, PARAMETER :: ncol = 34, ns_max = 8
   REAL(wp), PARAMETER :: pz = 2._wp, pzu = 10._wp
   INTEGER(4) :: n4
   INTEGER :: n, ns, k
   REAL(wp), DIMENSION(:,:,:), ALLOCATABLE :: c
   REAL(wp), DIMENSION(:,:),   ALLOCATABLE :: r1, r2, r3, r4, r5
   REAL(wp), DIMENSION(ns_max) :: s1, s2, s3, s4, s5
   CHARACTER(len=512) :: cfin, cfout
   !! columns
   INTEGER, PARAMETER :: iTa=1, iTs=2, iP=3, iqa=4, iqs=5, iTh=6, iPz=7, ius=8, itst=9, iqst=10, iW=11, iUb=12, iCd=13, iCh=14, &
      &                  iCe=15, ipsi=16, iz0=17, iRib=18, irlw=19, irh=20, idp=21, irho=22, iRer=23, ialp=24, iQd=25, iQlt=26,   &
      &                  iTly=27, iqly=28, iTi=29, inua=30, izeta=31, istab=32, isqcd=33, icharn=34

   CALL GET_COMMAND_ARGUMENT(1, cfin)
   CALL GET_COMMAND_ARGUMENT(2, cfout)
   OPEN(11, FILE=TRIM(cfin), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='OLD')
   READ(11) n4
   n = n4 ; ns = MIN(n, ns_max)
   ALLOCATE( c(n,1,ncol), r1(n,1), r2(n,1), r3(n,1), r4(n,1), r5(n,1) )
   READ(11) c
   CLOSE(11)
   OPEN(12, FILE=TRIM(cfout), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')

   !! ---- potential / absolute / virtual temperature
   r1 = pot_temp( c(:,:,iTa), c(:,:,iPz) )                         ; CALL put('pot_temp', r1)
   r1 = pot_temp( c(:,:,iTa), c(:,:,iPz), pPref=c(:,:,iP) )        ; CALL put('pot_temp_pref', r1)
   r1 = abs_temp( c(:,:,iTh), c(:,:,iPz) )                         ; CALL put('abs_temp', r1)
   r1 = abs_temp( c(:,:,iTh), c(:,:,iPz), pPref=c(:,:,iP) )        ; CALL put('abs_temp_pref', r1)
   r1 = virt_temp( c(:,:,iTa), c(:,:,iqa) )                        ; CALL put('virt_temp', r1)
   DO k = 1, ns
      s1(k) = pot_temp( c(k,1,iTa), c(k,1,iPz) )
      s2(k) = pot_temp( c(k,1,iTa), c(k,1,iPz), pPref=c(k,1,iP) )
      s3(k) = abs_temp( c(k,1,iTh), c(k,1,iPz), pPref=c(k,1,iP) )
      s4(k) = virt_temp( c(k,1,iTa), c(k,1,iqa) )
   END DO
   CALL puts('pot_temp_s', s1) ; CALL puts('pot_temp_pref_s', s2) ; CALL puts('abs_temp_pref_s', s3) ; CALL puts('virt_temp_s', s4)
   !! the scalar versions remember the last pPref they were given (an initialised local is SAVEd)
   s1(1) = pot_temp( c(1,1,iTa), c(1,1,iPz), pPref=99000._wp )
   s1(2) = pot_temp( c(1,1,iTa), c(1,1,iPz) )
   s1(3) = abs_temp( c(1,1,iTh), c(1,1,iPz), pPref=99000._wp )
   s1(4) = abs_temp( c(1,1,iTh), c(1,1,iPz) )
   CALL putk('pref_sticky_s', s1, 4)

   !! ---- pressure and temperature at height
   r1 = Pz_from_P0_tz_qz( pz, c(:,:,iP), c(:,:,iTa), c(:,:,iqa) )  ; CALL put('pz_from_p0', r1)
   r1 = Theta_from_z_P0_T_q( pz, c(:,:,iP), c(:,:,iTa), c(:,:,iqa) ) ; CALL put('theta_from_z', r1)
   r1 = T_from_z_P0_Theta_q( pz, c(:,:,iP), c(:,:,iTh), c(:,:,iqa) ) ; CALL put('t_from_z', r1)
   DO k = 1, ns
      s1(k) = Pz_from_P0_tz_qz( pz, c(k,1,iP), c(k,1,iTa), c(k,1,iqa) )
      s2(k) = Theta_from_z_P0_T_q( pz, c(k,1,iP), c(k,1,iTa), c(k,1,iqa) )
      s3(k) = T_from_z_P0_Theta_q( pz, c(k,1,iP), c(k,1,iTh), c(k,1,iqa) )
   END DO
   CALL puts('pz_from_p0_s', s1) ; CALL puts('theta_from_z_s', s2) ; CALL puts('t_from_z_s', s3)
   r1 = Pz_from_P0_tz_qz( pz, c(:,:,iP), c(:,:,iTi), c(:,:,iqa), l_ice=.TRUE. )  ; CALL put('pz_from_p0_ice', r1)
   r1 = Theta_from_z_P0_T_q( pz, c(:,:,iP), c(:,:,iTi), c(:,:,iqa) )   ; CALL put('theta_from_z_after_ice', r1)   ! l_ice sticks
   r1 = Pz_from_P0_tz_qz( pz, c(:,:,iP), c(:,:,iTa), c(:,:,iqa), l_ice=.FALSE. ) ; CALL put('pz_from_p0_again', r1)

   !! ---- air properties
   r1 = rho_air( c(:,:,iTa), c(:,:,iqa), c(:,:,iP) )               ; CALL put('rho_air', r1)
   r1 = visc_air( c(:,:,iTa) )                                     ; CALL put('visc_air', r1)
   r1 = L_vap( c(:,:,iTs) )                                        ; CALL put('l_vap', r1)
   r1 = cp_air( c(:,:,iqa) )                                       ; CALL put('cp_air', r1)
   r1 = gamma_moist( c(:,:,iTa), c(:,:,iqa) )                      ; CALL put('gamma_moist', r1)
   r1 = rho_air_adv( c(:,:,iTa), c(:,:,iqa), c(:,:,iP) )           ; CALL put('rho_air_adv', r1)
   r1 = dry_static_energy( pz, c(:,:,iTa), c(:,:,iqa) )            ; CALL put('dry_static_energy', r1)
   DO k = 1, ns
      s1(k) = rho_air( c(k,1,iTa), c(k,1,iqa), c(k,1,iP) )
      s2(k) = visc_air( c(k,1,iTa) )
      s3(k) = L_vap( c(k,1,iTs) )
      s4(k) = cp_air( c(k,1,iqa) )
      s5(k) = gamma_moist( c(k,1,iTa), c(k,1,iqa) )
   END DO
   CALL puts('rho_air_s', s1) ; CALL puts('visc_air_s', s2) ; CALL puts('l_vap_s', s3) ; CALL puts('cp_air_s', s4)
   CALL puts('gamma_moist_s', s5)

   !! ---- stability
   r1 = One_on_L( c(:,:,iTh), c(:,:,iqa), c(:,:,ius), c(:,:,itst), c(:,:,iqst) )            ; CALL put('one_on_l', r1)
   r1 = Ri_bulk( pzu, c(:,:,iTs), c(:,:,iTh), c(:,:,iqs), c(:,:,iqa), c(:,:,iUb) )         ; CALL put('ri_bulk', r1)
   DO k = 1, ns
      s1(k) = One_on_L( c(k,1,iTh), c(k,1,iqa), c(k,1,ius), c(k,1,itst), c(k,1,iqst) )
      s2(k) = Ri_bulk( pzu, c(k,1,iTs), c(k,1,iTh), c(k,1,iqs), c(k,1,iqa), c(k,1,iUb) )
   END DO
   CALL puts('one_on_l_s', s1) ; CALL puts('ri_bulk_s', s2)
   !! FIRST_GUESS_COARE (zt = 2 m, zu = 10 m; a Charnock parameter per cell).  Here, BEFORE Ri_bulk is given its layer arguments: the routine calls
   !! Ri_bulk_sclr without them, and the reference's function, once it has had them, reads them for ever (absent: a segmentation fault)
   BLOCK
      REAL(wp), DIMENSION(:,:), ALLOCATABLE :: g1, g2, g3, g4, g5, g6, g7
      ALLOCATE( g1(n,1), g2(n,1), g3(n,1), g4(n,1), g5(n,1), g6(n,1), g7(n,1) )
      CALL first_guess_coare( pz, pzu, c(:,:,iTs), c(:,:,iTh), c(:,:,iqs), c(:,:,iqa), c(:,:,iW), c(:,:,icharn), g1, g2, g3, g4, g5, g6, qz0=g7 )
      CALL put('fg_us', g1) ; CALL put('fg_ts', g2) ; CALL put('fg_qs', g3) ; CALL put('fg_t_zu', g4) ; CALL put('fg_q_zu', g5)
      CALL put('fg_ub', g6) ; CALL put('fg_z0', g7)
      DO k = 1, ns
         CALL first_guess_coare( pz, pzu, c(k,1,iTs), c(k,1,iTh), c(k,1,iqs), c(k,1,iqa), c(k,1,iW), c(k,1,icharn), &
            &                    s1(k), s2(k), s3(k), s4(k), s5(k), g6(k,1) )
      END DO
      CALL puts('fg_us_s', s1) ; CALL puts('fg_t_zu_s', s4)
   END BLOCK
   !! (the layer arguments last: the reference's Ri_bulk never forgets that it once had them)
   r1 = Ri_bulk( pzu, c(:,:,iTs), c(:,:,iTh), c(:,:,iqs), c(:,:,iqa), c(:,:,iUb), pTa_layer=c(:,:,iTly), pqa_layer=c(:,:,iqly) )
   CALL put('ri_bulk_layer', r1)

   !! ---- saturation
   r1 = e_sat( c(:,:,iTa) )                                        ; CALL put('e_sat', r1)
   r1 = e_sat_ice( c(:,:,iTi) )                                    ; CALL put('e_sat_ice', r1)
   r1 = de_sat_dt_ice( c(:,:,iTi) )                                ; CALL put('de_sat_dt_ice', r1)
   r1 = q_sat( c(:,:,iTa), c(:,:,iP) )                             ; CALL put('q_sat', r1)
   r1 = q_sat( c(:,:,iTi), c(:,:,iP), l_ice=.TRUE. )               ; CALL put('q_sat_ice', r1)
   r1 = dq_sat_dt_ice( c(:,:,iTi), c(:,:,iP) )                     ; CALL put('dq_sat_dt_ice', r1)
   r1 = q_air_rh( c(:,:,irh), c(:,:,iTa), c(:,:,iP) )              ; CALL put('q_air_rh', r1)
   r1 = q_air_dp( c(:,:,idp), c(:,:,iP) )                          ; CALL put('q_air_dp', r1)
   r1 = q_sat_crude( c(:,:,iTs), c(:,:,irho) )                     ; CALL put('q_sat_crude', r1)
   r1 = e_air( c(:,:,iqa), c(:,:,iP) )                             ; CALL put('e_air', r1)
   r1 = rh_air( c(:,:,iqa), c(:,:,iTa), c(:,:,iP) )                ; CALL put('rh_air', r1)
   DO k = 1, ns
      s1(k) = e_sat( c(k,1,iTa) )
      s2(k) = e_sat_ice( c(k,1,iTi) )
      s3(k) = de_sat_dt_ice( c(k,1,iTi) )
      s4(k) = q_sat( c(k,1,iTa), c(k,1,iP) )
      s5(k) = dq_sat_dt_ice( c(k,1,iTi), c(k,1,iP) )
   END DO
   CALL puts('e_sat_s', s1) ; CALL puts('e_sat_ice_s', s2) ; CALL puts('de_sat_dt_ice_s', s3) ; CALL puts('q_sat_s', s4)
   CALL puts('dq_sat_dt_ice_s', s5)
   DO k = 1, ns
      s1(k) = q_sat( c(k,1,iTi), c(k,1,iP), l_ice=.TRUE. )
   END DO
   CALL puts('q_sat_ice_s', s1)

   !! ---- fluxes
   CALL UPDATE_QNSOL_TAU( pzu, c(:,:,iTs), c(:,:,iqs), c(:,:,iTh), c(:,:,iqa), c(:,:,ius), c(:,:,itst), c(:,:,iqst), c(:,:,iW), &
      &                   c(:,:,iUb), c(:,:,iP), c(:,:,irlw), r1, r2, Qlat=r3 )
   CALL put('uqt_qns', r1) ; CALL put('uqt_tau', r2) ; CALL put('uqt_qlat', r3)
   CALL BULK_FORMULA( pzu, c(:,:,iTs), c(:,:,iqs), c(:,:,iTh), c(:,:,iqa), c(:,:,iCd), c(:,:,iCh), c(:,:,iCe), c(:,:,iW),     &
      &               c(:,:,iUb), c(:,:,iP), r1, r2, r3, pEvap=r4, prhoa=r5 )
   CALL put('bf_tau', r1) ; CALL put('bf_qsen', r2) ; CALL put('bf_qlat', r3) ; CALL put('bf_evap', r4) ; CALL put('bf_rhoa', r5)
   CALL BULK_FORMULA( pzu, c(:,:,iTi), c(:,:,iqs), c(:,:,iTh), c(:,:,iqa), c(:,:,iCd), c(:,:,iCh), c(:,:,iCe), c(:,:,iW),     &
      &               c(:,:,iUb), c(:,:,iP), r1, r2, r3, pEvap=r4, l_ice=.TRUE. )
   CALL put('bf_ice_qlat', r3) ; CALL put('bf_ice_evap', r4)
   DO k = 1, ns
      CALL UPDATE_QNSOL_TAU( pzu, c(k,1,iTs), c(k,1,iqs), c(k,1,iTh), c(k,1,iqa), c(k,1,ius), c(k,1,itst), c(k,1,iqst), c(k,1,iW), &
         &                   c(k,1,iUb), c(k,1,iP), c(k,1,irlw), s1(k), s2(k), Qlat=s3(k) )
      CALL BULK_FORMULA( pzu, c(k,1,iTs), c(k,1,iqs), c(k,1,iTh), c(k,1,iqa), c(k,1,iCd), c(k,1,iCh), c(k,1,iCe), c(k,1,iW),     &
         &               c(k,1,iUb), c(k,1,iP), s4(k), s5(k), r1(1,1), prhoa=r2(1,1) )
   END DO
   CALL puts('uqt_qns_s', s1) ; CALL puts('uqt_tau_s', s2) ; CALL puts('uqt_qlat_s', s3) ; CALL puts('bf_tau_s', s4)
   CALL puts('bf_qsen_s', s5)
   r1 = alpha_sw( c(:,:,iTs) )                                     ; CALL put('alpha_sw', r1)
   r1 = qlw_net( c(:,:,irlw), c(:,:,iTs) )                         ; CALL put('qlw_net', r1)
   r1 = qlw_net( c(:,:,irlw), c(:,:,iTi), l_ice=.TRUE. )           ; CALL put('qlw_net_ice', r1)
   DO k = 1, ns
      s1(k) = alpha_sw( c(k,1,iTs) )
      s2(k) = qlw_net( c(k,1,irlw), c(k,1,iTs) )
      s3(k) = delta_skin_layer_sclr( c(k,1,ialp), c(k,1,iQd), c(k,1,ius) )
      s4(k) = delta_skin_layer_sclr( c(k,1,ialp), c(k,1,iQd), c(k,1,ius), Qlat=c(k,1,iQlt) )
   END DO
   CALL puts('alpha_sw_s', s1) ; CALL puts('qlw_net_s', s2) ; CALL puts('delta_skin_s', s3) ; CALL puts('delta_skin_qlat_s', s4)

   !! ---- roughness, neutral wind, Louis
   r1 = z0_from_Cd( pzu, c(:,:,iCd) )                              ; CALL put('z0_from_cd', r1)
   r1 = z0_from_Cd( pzu, c(:,:,iCd), ppsi=c(:,:,ipsi) )            ; CALL put('z0_from_cd_psi', r1)
   r1 = z0_from_ustar( pzu, c(:,:,ius), c(:,:,iUb) )               ; CALL put('z0_from_ustar', r1)
   r1 = Cd_from_z0( pzu, c(:,:,iz0) )                              ; CALL put('cd_from_z0', r1)
   r1 = Cd_from_z0( pzu, c(:,:,iz0), ppsi=c(:,:,ipsi) )            ; CALL put('cd_from_z0_psi', r1)
   r1 = f_m_louis( pzu, c(:,:,iRib), c(:,:,iCd), c(:,:,iz0) )      ; CALL put('f_m_louis', r1)
   r1 = f_h_louis( pzu, c(:,:,iRib), c(:,:,iCh), c(:,:,iz0) )      ; CALL put('f_h_louis', r1)
   r1 = UN10_from_ustar( pzu, c(:,:,iUb), c(:,:,ius), c(:,:,ipsi) ) ; CALL put('un10_from_ustar', r1)
   r1 = UN10_from_CDN( pzu, c(:,:,iUb), c(:,:,iCd), c(:,:,ipsi) )  ; CALL put('un10_from_cdn', r1)
   r1 = UN10_from_CD( pzu, c(:,:,iUb), c(:,:,iCd), c(:,:,ipsi) )   ; CALL put('un10_from_cd', r1)
   r1 = z0tq_LKB( 1, c(:,:,iRer), c(:,:,iz0) )                     ; CALL put('z0t_lkb', r1)
   r1 = z0tq_LKB( 2, c(:,:,iRer), c(:,:,iz0) )                     ; CALL put('z0q_lkb', r1)
   DO k = 1, ns
      s1(k) = z0_from_Cd( pzu, c(k,1,iCd), ppsi=c(k,1,ipsi) )
      s2(k) = z0_from_ustar( pzu, c(k,1,ius), c(k,1,iUb) )
      s3(k) = f_m_louis( pzu, c(k,1,iRib), c(k,1,iCd), c(k,1,iz0) )
      s4(k) = f_h_louis( pzu, c(k,1,iRib), c(k,1,iCh), c(k,1,iz0) )
      s5(k) = UN10_from_CD( pzu, c(k,1,iUb), c(k,1,iCd), c(k,1,ipsi) )
   END DO
   CALL puts('z0_from_cd_psi_s', s1) ; CALL puts('z0_from_ustar_s', s2) ; CALL puts('f_m_louis_s', s3) ; CALL puts('f_h_louis_s', s4)
   CALL puts('un10_from_cd_s', s5)

   !! ---- roughness lengths over sea ice (Andreas et al. 2005), functions of u* and the air's viscosity
   BLOCK
      REAL(wp), DIMENSION(:,:,:), ALLOCATABLE :: ztq
      ALLOCATE( ztq(n,1,2) )
      r1 = rough_leng_m( c(:,:,ius), c(:,:,inua) )                 ; CALL put('rough_leng_m', r1)
      ztq = rough_leng_tq( c(:,:,iz0), c(:,:,ius), c(:,:,inua) )
      r3 = ztq(:,:,1)                                              ; CALL put('rough_leng_t', r3)
      r3 = ztq(:,:,2)                                              ; CALL put('rough_leng_q', r3)
   END BLOCK

   !! ---- stability functions of zeta = z/L, Charnock parameters, NCAR's neutral coefficients, ANDREAS' u*
   r1 = psi_m_coare( c(:,:,izeta) )                                ; CALL put('psi_m_coare', r1)
   r1 = psi_h_coare( c(:,:,izeta) )                                ; CALL put('psi_h_coare', r1)
   r1 = psi_m_ncar( c(:,:,izeta) )                                 ; CALL put('psi_m_ncar', r1)
   r1 = psi_h_ncar( c(:,:,izeta) )                                 ; CALL put('psi_h_ncar', r1)
   r1 = psi_m_ecmwf( c(:,:,izeta) )                                ; CALL put('psi_m_ecmwf', r1)
   r1 = psi_h_ecmwf( c(:,:,izeta) )                                ; CALL put('psi_h_ecmwf', r1)
   r1 = psi_m_andreas( c(:,:,izeta) )                              ; CALL put('psi_m_andreas', r1)
   r1 = psi_h_andreas( c(:,:,izeta) )                              ; CALL put('psi_h_andreas', r1)
   r1 = charn_coare3p6( c(:,:,iW) )                                ; CALL put('charn_coare3p6', r1)
   r1 = cd_n10_ncar( c(:,:,iW) )                                   ; CALL put('cd_n10_ncar', r1)
   r3 = ch_n10_ncar( c(:,:,isqcd), c(:,:,istab) )                  ; CALL put('ch_n10_ncar', r3)
   r3 = ce_n10_ncar( c(:,:,isqcd) )                                ; CALL put('ce_n10_ncar', r3)
   r1 = u_star_andreas( c(:,:,iW) )                                ; CALL put('u_star_andreas', r1)
   DO k = 1, n
      r1(k,1) = charn_coare3p0( c(k,1,iW) )                        ! a scalar function in the reference
   END DO
   CALL put('charn_coare3p0', r1)
   DO k = 1, ns
      s1(k) = psi_m_coare( c(k,1,izeta) ) ; s2(k) = psi_h_ecmwf( c(k,1,izeta) ) ; s3(k) = psi_m_ncar( c(k,1,izeta) )
      s4(k) = u_star_andreas( c(k,1,iW) ) ; s5(k) = charn_coare3p6( c(k,1,iW) )
   END DO
   CALL puts('psi_m_coare_s', s1) ; CALL puts('psi_h_ecmwf_s', s2) ; CALL puts('psi_m_ncar_s', s3) ; CALL puts('u_star_andreas_s', s4)
   CALL puts('charn_coare3p6_s', s5)

   !! ---- host-side statistics
   s1(1) = VARIANCE( c(:,1,iTa) ) ; s1(2) = VMEAN( c(:,1,iTa) )
   CALL putk('variance_vmean', s1, 2)
   s1(1) = MERGE( 1._wp, 0._wp, type_of_humidity( c(:,:,iqa), INT(1 + 0*NINT(c(:,:,iqa)),1) ) == 'sh' )
   s1(2) = MERGE( 1._wp, 0._wp, type_of_humidity( c(:,:,idp), INT(1 + 0*NINT(c(:,:,iqa)),1) ) == 'dp' )
   s1(3) = MERGE( 1._wp, 0._wp, type_of_humidity( c(:,:,irh), INT(1 + 0*NINT(c(:,:,iqa)),1) ) == 'rh' )
   CALL putk('type_of_humidity', s1, 3)
   CALL check_unit_consistency( 'sst', c(:,:,iTs) )
   CALL check_unit_consistency( 'slp', c(:,:,iP), mask=INT(1 + 0*NINT(c(:,:,iqa)),1) )
   !! constants a caller reads from mod_const
   s1(1) = grav ; s1(2) = rt0 ; s1(3) = reps0 ; s1(4) = rctv0 ; s1(5) = rcst_cs ; s1(6) = sq_radrw ; s1(7) = rpoiss_dry ; s1(8) = rgamma_dry
   CALL putk('mod_const', s1, 8)
   CLOSE(12)

CONTAINS

   SUBROUTINE put( cname, pr )
      CHARACTER(len=*), INTENT(in) :: cname
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pr
      CHARACTER(len=24) :: c24
      c24 = cname
      WRITE(12) c24, INT(SIZE(pr),4), pr
      FLUSH(12)
   END SUBROUTINE put

   SUBROUTINE putk( cname, ps, km )
      CHARACTER(len=*), INTENT(in) :: cname
      REAL(wp), DIMENSION(:), INTENT(in) :: ps
      INTEGER, INTENT(in) :: km
      CHARACTER(len=24) :: c24
      c24 = cname
      WRITE(12) c24, INT(km,4), ps(1:km)
   END SUBROUTINE putk

   SUBROUTINE puts( cname, ps )
      CHARACTER(len=*), INTENT(in) :: cname
      REAL(wp), DIMENSION(:), INTENT(in) :: ps
      CALL putk( cname, ps, ns )
   END SUBROUTINE puts

END PROGRAM phymbl_driver
