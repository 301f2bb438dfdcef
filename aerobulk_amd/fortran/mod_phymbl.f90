! mod_phymbl.f90 -- the thermodynamic helper functions callers of AeroBulk import with `USE mod_phymbl`, on the MI355X engine.
!
! Source-compatibility module: every PUBLIC generic and specific name of the reference's src/mod_phymbl.f90 (generics :33-139,
! procedures :163-2046) with the same dummy-argument names (callers use keywords: pPref=, l_ice=, ppsi=, Qlat= ...), kinds,
! OPTIONALs and result shapes.  Nothing is computed in Fortran: each procedure hands its arrays to `ab_phymbl` of the C ABI
! (include/aerobulk_amd.h), i.e. to an elementwise HIP kernel of libaerobulk_amd.so built from the device functions the flux kernels
! use themselves (aerobulk_amd/csrc/ab_phymbl.hpp).  A scalar call is an array call of one cell: there is no CPU implementation.
! Exceptions, all host-side bookkeeping with no arithmetic of the flux path in it: VARIANCE, VMEAN, TO_KELVIN_3D,
! check_unit_consistency and type_of_humidity (statistics of a field against the ranges of mod_const).
!
! Behaviours of the reference that are reproduced although they are slips, because a caller can observe them:
!   * initialised locals are SAVEd in Fortran: `zPref = Patm` of pot_temp_sclr / abs_temp_sclr (:179,:221) and `lice = .FALSE.` of
!     Pz_from_P0_tz_qz_sclr / _vctr (:300,:329) keep the value of the LAST call that passed pPref / l_ice;
!     Theta_from_z_P0_T_q and T_from_z_P0_Theta_q inherit that l_ice, and T_from_z_P0_Theta_q leaves abs_temp_sclr's zPref at its pslp;
!   * BULK_FORMULA_VCTR takes the PRESENCE of l_ice for its value (:1236).
! Not reproduced: Ri_bulk's sticky `l_ptqa_l_prvd` (:729,:759), which makes the reference read absent arguments on a later call.
!
! Build with -fdefault-real-8 like the rest (arch/make.macro_GnuLinux:17).

MODULE mod_phymbl

   USE, INTRINSIC :: ISO_C_BINDING, ONLY: C_INT, C_LONG, C_PTR, C_DOUBLE, C_CHAR, C_SIZE_T, C_NULL_PTR, C_LOC, C_ASSOCIATED, C_F_POINTER
   USE mod_const

   IMPLICIT NONE
   PRIVATE :: C_INT, C_LONG, C_PTR, C_DOUBLE, C_CHAR, C_SIZE_T, C_NULL_PTR, C_LOC, C_ASSOCIATED, C_F_POINTER

   INTERFACE pot_temp
      MODULE PROCEDURE pot_temp_vctr, pot_temp_sclr
   END INTERFACE pot_temp
   INTERFACE abs_temp
      MODULE PROCEDURE abs_temp_vctr, abs_temp_sclr
   END INTERFACE abs_temp
   INTERFACE virt_temp
      MODULE PROCEDURE virt_temp_vctr, virt_temp_sclr
   END INTERFACE virt_temp
   INTERFACE Pz_from_P0_tz_qz
      MODULE PROCEDURE Pz_from_P0_tz_qz_vctr, Pz_from_P0_tz_qz_sclr
   END INTERFACE Pz_from_P0_tz_qz
   INTERFACE Theta_from_z_P0_T_q
      MODULE PROCEDURE Theta_from_z_P0_T_q_vctr, Theta_from_z_P0_T_q_sclr
   END INTERFACE Theta_from_z_P0_T_q
   INTERFACE T_from_z_P0_Theta_q
      MODULE PROCEDURE T_from_z_P0_Theta_q_vctr, T_from_z_P0_Theta_q_sclr
   END INTERFACE T_from_z_P0_Theta_q
   INTERFACE visc_air
      MODULE PROCEDURE visc_air_vctr, visc_air_sclr
   END INTERFACE visc_air
   INTERFACE gamma_moist
      MODULE PROCEDURE gamma_moist_vctr, gamma_moist_sclr
   END INTERFACE gamma_moist
   INTERFACE e_sat
      MODULE PROCEDURE e_sat_vctr, e_sat_sclr
   END INTERFACE e_sat
   INTERFACE e_sat_ice
      MODULE PROCEDURE e_sat_ice_vctr, e_sat_ice_sclr
   END INTERFACE e_sat_ice
   INTERFACE de_sat_dt_ice
      MODULE PROCEDURE de_sat_dt_ice_vctr, de_sat_dt_ice_sclr
   END INTERFACE de_sat_dt_ice
   INTERFACE One_on_L
      MODULE PROCEDURE One_on_L_vctr, One_on_L_sclr
   END INTERFACE One_on_L
   INTERFACE Ri_bulk
      MODULE PROCEDURE Ri_bulk_vctr, Ri_bulk_sclr
   END INTERFACE Ri_bulk
   INTERFACE q_sat
      MODULE PROCEDURE q_sat_vctr, q_sat_sclr
   END INTERFACE q_sat
   INTERFACE dq_sat_dt_ice
      MODULE PROCEDURE dq_sat_dt_ice_vctr, dq_sat_dt_ice_sclr
   END INTERFACE dq_sat_dt_ice
   INTERFACE L_vap
      MODULE PROCEDURE L_vap_vctr, L_vap_sclr
   END INTERFACE L_vap
   INTERFACE rho_air
      MODULE PROCEDURE rho_air_vctr, rho_air_sclr
   END INTERFACE rho_air
   INTERFACE cp_air
      MODULE PROCEDURE cp_air_vctr, cp_air_sclr
   END INTERFACE cp_air
   INTERFACE alpha_sw
      MODULE PROCEDURE alpha_sw_vctr, alpha_sw_sclr
   END INTERFACE alpha_sw
   INTERFACE update_qnsol_tau
      MODULE PROCEDURE update_qnsol_tau_vctr, update_qnsol_tau_sclr
   END INTERFACE update_qnsol_tau
   INTERFACE bulk_formula
      MODULE PROCEDURE bulk_formula_vctr, bulk_formula_sclr
   END INTERFACE bulk_formula
   INTERFACE qlw_net
      MODULE PROCEDURE qlw_net_vctr, qlw_net_sclr
   END INTERFACE qlw_net
   INTERFACE z0_from_Cd
      MODULE PROCEDURE z0_from_Cd_vctr, z0_from_Cd_sclr
   END INTERFACE z0_from_Cd
   INTERFACE z0_from_ustar
      MODULE PROCEDURE z0_from_ustar_vctr, z0_from_ustar_sclr
   END INTERFACE z0_from_ustar
   INTERFACE UN10_from_CD
      MODULE PROCEDURE UN10_from_CD_vctr, UN10_from_CD_sclr
   END INTERFACE UN10_from_CD
   INTERFACE f_m_louis
      MODULE PROCEDURE f_m_louis_vctr, f_m_louis_sclr
   END INTERFACE f_m_louis
   INTERFACE f_h_louis
      MODULE PROCEDURE f_h_louis_vctr, f_h_louis_sclr
   END INTERFACE f_h_louis

   !! ---- private: the binding to the engine ---------------------------------------------------------------------------------
   !! function ids: enum ab_phymbl_fn, include/aerobulk_amd.h
   INTEGER(C_INT), PARAMETER, PRIVATE :: &
      &  PH_POT_TEMP = 1, PH_ABS_TEMP = 2, PH_VIRT_TEMP = 3, PH_PZ = 4, PH_THETA = 5, PH_TABS = 6, PH_RHO_AIR = 7, PH_VISC_AIR = 8,   &
      &  PH_L_VAP = 9, PH_CP_AIR = 10, PH_GAMMA_MOIST = 11, PH_ONE_ON_L = 12, PH_RI_BULK = 13, PH_E_SAT = 14, PH_E_SAT_ICE = 15,     &
      &  PH_DE_SAT_DT_ICE = 16, PH_Q_SAT = 17, PH_DQ_SAT_DT_ICE = 18, PH_Q_AIR_RH = 19, PH_Q_AIR_DP = 20, PH_RHO_AIR_ADV = 21,       &
      &  PH_Q_SAT_CRUDE = 22, PH_DRY_STATIC_ENERGY = 23, PH_UPDATE_QNSOL_TAU = 24, PH_BULK_FORMULA = 25, PH_ALPHA_SW = 26,           &
      &  PH_QLW_NET = 27, PH_Z0_FROM_CD = 28, PH_Z0_FROM_USTAR = 29, PH_CD_FROM_Z0 = 30, PH_F_M_LOUIS = 31, PH_F_H_LOUIS = 32,        &
      &  PH_UN10_FROM_USTAR = 33, PH_UN10_FROM_CDN = 34, PH_UN10_FROM_CD = 35, PH_Z0TQ_LKB = 36, PH_E_AIR = 37, PH_RH_AIR = 38,       &
      &  PH_DELTA_SKIN = 39

   INTERFACE
      FUNCTION ab_phymbl( fn, n, pin, n_in, pout, n_out, par, iflag, mem, stream, info ) BIND(C, NAME='ab_phymbl') RESULT(istat)
         IMPORT :: C_INT, C_LONG, C_PTR, C_DOUBLE
         INTEGER(C_INT),  VALUE :: fn, n_in, n_out, iflag, mem
         INTEGER(C_LONG), VALUE :: n
         TYPE(C_PTR), DIMENSION(*), INTENT(in) :: pin, pout
         REAL(C_DOUBLE), DIMENSION(2), INTENT(in) :: par
         TYPE(C_PTR),     VALUE :: stream
         REAL(C_DOUBLE), DIMENSION(2), INTENT(out) :: info
         INTEGER(C_INT) :: istat
      END FUNCTION ab_phymbl
      FUNCTION ph_last_error() BIND(C, NAME='ab_last_error') RESULT(cptr)
         IMPORT :: C_PTR
         TYPE(C_PTR) :: cptr
      END FUNCTION ph_last_error
      FUNCTION ph_strlen(s) BIND(C, NAME='strlen') RESULT(n)
         IMPORT :: C_PTR, C_SIZE_T
         TYPE(C_PTR), VALUE :: s
         INTEGER(C_SIZE_T) :: n
      END FUNCTION ph_strlen
   END INTERFACE
   PRIVATE :: ab_phymbl, ph_last_error, ph_strlen, ph_run, ph_stop, ph_s1

   !! the reference's SAVEd locals (see the header)
   REAL(wp), SAVE, PRIVATE :: zPref_pot_s = Patm, zPref_abs_s = Patm
   LOGICAL,  SAVE, PRIVATE :: lice_pz_s = .FALSE., lice_pz_v = .FALSE.

CONTAINS

   !! =============================================================================================================== engine calls
   SUBROUTINE ph_stop( istat )
      !! an engine failure (no GPU, HIP error, bad argument) ends the program the way ctl_stop does
      INTEGER(C_INT), INTENT(in) :: istat
      TYPE(C_PTR) :: cp
      CHARACTER(KIND=C_CHAR), DIMENSION(:), POINTER :: cs
      CHARACTER(len=512) :: cmsg
      INTEGER :: n, i
      cmsg = '' ; cp = ph_last_error()
      IF( C_ASSOCIATED(cp) ) THEN
         n = MIN( INT(ph_strlen(cp)), 512 )
         CALL C_F_POINTER( cp, cs, (/ n /) )
         DO i = 1, n
            cmsg(i:i) = cs(i)
         END DO
      END IF
      WRITE(6,cform_err)
      WRITE(6,*) ' mod_phymbl (aerobulk_amd): status', INT(istat), ' ', TRIM(cmsg)
      WRITE(6,*) ''
      STOP
   END SUBROUTINE ph_stop

   SUBROUTINE ph_run( fn, n, par1, iflag, o1, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, o2, o3, o4, o5, info, tolerate )
      !! one `ab_phymbl` call on contiguous storage: explicit-shape dummies make the compiler pack a strided actual argument
      INTEGER(C_INT), INTENT(in) :: fn
      INTEGER,        INTENT(in) :: n, iflag
      REAL(wp),       INTENT(in) :: par1
      REAL(wp), DIMENSION(n), INTENT(out), TARGET           :: o1
      REAL(wp), DIMENSION(n), INTENT(in),  TARGET, OPTIONAL :: a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11
      REAL(wp), DIMENSION(n), INTENT(out), TARGET, OPTIONAL :: o2, o3, o4, o5
      REAL(C_DOUBLE), DIMENSION(2), INTENT(out),   OPTIONAL :: info
      INTEGER,        INTENT(in),                  OPTIONAL :: tolerate   ! a status that is the caller's business (AB_ERR_TAU = 8)
      TYPE(C_PTR), DIMENSION(11) :: pin
      TYPE(C_PTR), DIMENSION(5)  :: pout
      REAL(C_DOUBLE), DIMENSION(2) :: par, zinfo
      INTEGER(C_INT) :: istat, nin
      pin(:) = C_NULL_PTR ; pout(:) = C_NULL_PTR ; nin = 0
      IF( PRESENT(a1)  ) THEN ; pin(1)  = C_LOC(a1)  ; nin = 1  ; END IF
      IF( PRESENT(a2)  ) THEN ; pin(2)  = C_LOC(a2)  ; nin = 2  ; END IF
      IF( PRESENT(a3)  ) THEN ; pin(3)  = C_LOC(a3)  ; nin = 3  ; END IF
      IF( PRESENT(a4)  ) THEN ; pin(4)  = C_LOC(a4)  ; nin = 4  ; END IF
      IF( PRESENT(a5)  ) THEN ; pin(5)  = C_LOC(a5)  ; nin = 5  ; END IF
      IF( PRESENT(a6)  ) THEN ; pin(6)  = C_LOC(a6)  ; nin = 6  ; END IF
      IF( PRESENT(a7)  ) THEN ; pin(7)  = C_LOC(a7)  ; nin = 7  ; END IF
      IF( PRESENT(a8)  ) THEN ; pin(8)  = C_LOC(a8)  ; nin = 8  ; END IF
      IF( PRESENT(a9)  ) THEN ; pin(9)  = C_LOC(a9)  ; nin = 9  ; END IF
      IF( PRESENT(a10) ) THEN ; pin(10) = C_LOC(a10) ; nin = 10 ; END IF
      IF( PRESENT(a11) ) THEN ; pin(11) = C_LOC(a11) ; nin = 11 ; END IF
      pout(1) = C_LOC(o1)
      IF( PRESENT(o2) ) pout(2) = C_LOC(o2)
      IF( PRESENT(o3) ) pout(3) = C_LOC(o3)
      IF( PRESENT(o4) ) pout(4) = C_LOC(o4)
      IF( PRESENT(o5) ) pout(5) = C_LOC(o5)
      par(1) = REAL(par1, C_DOUBLE) ; par(2) = 0._C_DOUBLE
      istat = ab_phymbl( fn, INT(n,C_LONG), pin, nin, pout, 5_C_INT, par, INT(iflag,C_INT), 0_C_INT, C_NULL_PTR, zinfo )
      IF( PRESENT(info) ) info = zinfo
      IF( istat /= 0 ) THEN
         IF( PRESENT(tolerate) ) THEN
            IF( INT(istat) == tolerate ) RETURN
         END IF
         CALL ph_stop( istat )
      END IF
   END SUBROUTINE ph_run

   FUNCTION ph_s1( fn, par1, iflag, x1, x2, x3, x4, x5, x6, x7 )
      !! a function of up to seven scalars = the same kernel on arrays of one cell
      INTEGER(C_INT), INTENT(in) :: fn
      REAL(wp),       INTENT(in) :: par1
      INTEGER,        INTENT(in) :: iflag
      REAL(wp),       INTENT(in) :: x1
      REAL(wp),       INTENT(in), OPTIONAL :: x2, x3, x4, x5, x6, x7
      REAL(wp) :: ph_s1
      REAL(wp), DIMENSION(1) :: zo, z1, z2, z3, z4, z5, z6, z7
      z1 = x1 ; z2 = 0._wp ; z3 = 0._wp ; z4 = 0._wp ; z5 = 0._wp ; z6 = 0._wp ; z7 = 0._wp
      IF( PRESENT(x2) ) z2 = x2
      IF( PRESENT(x3) ) z3 = x3
      IF( PRESENT(x4) ) z4 = x4
      IF( PRESENT(x5) ) z5 = x5
      IF( PRESENT(x6) ) z6 = x6
      IF( PRESENT(x7) ) z7 = x7
      IF( PRESENT(x7) ) THEN
         CALL ph_run( fn, 1, par1, iflag, zo, z1, z2, z3, z4, z5, z6, z7 )
      ELSEIF( PRESENT(x5) ) THEN
         CALL ph_run( fn, 1, par1, iflag, zo, z1, z2, z3, z4, z5 )
      ELSEIF( PRESENT(x4) ) THEN
         CALL ph_run( fn, 1, par1, iflag, zo, z1, z2, z3, z4 )
      ELSEIF( PRESENT(x3) ) THEN
         CALL ph_run( fn, 1, par1, iflag, zo, z1, z2, z3 )
      ELSEIF( PRESENT(x2) ) THEN
         CALL ph_run( fn, 1, par1, iflag, zo, z1, z2 )
      ELSE
         CALL ph_run( fn, 1, par1, iflag, zo, z1 )
      END IF
      ph_s1 = zo(1)
   END FUNCTION ph_s1


   !! ===================================================================================================== potential temperature
   FUNCTION pot_temp_sclr( pTa, pPz,  pPref )                                  !! reference :163-186
      REAL(wp), INTENT(in)           :: pTa, pPz
      REAL(wp), INTENT(in), OPTIONAL :: pPref
      REAL(wp)                       :: pot_temp_sclr
      IF( PRESENT(pPref) ) zPref_pot_s = pPref
      pot_temp_sclr = ph_s1( PH_POT_TEMP, zPref_pot_s, 0, pTa, pPz )
   END FUNCTION pot_temp_sclr

   FUNCTION pot_temp_vctr( pTa, pPz,  pPref )                                  !! :189-200
      REAL(wp), DIMENSION(:,:), INTENT(in)           :: pTa, pPz
      REAL(wp), DIMENSION(:,:), INTENT(in), OPTIONAL :: pPref
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2))   :: pot_temp_vctr
      IF( PRESENT(pPref) ) THEN
         CALL ph_run( PH_POT_TEMP, SIZE(pTa), Patm, 0, pot_temp_vctr, pTa, pPz, pPref )
      ELSE
         CALL ph_run( PH_POT_TEMP, SIZE(pTa), Patm, 0, pot_temp_vctr, pTa, pPz )
      END IF
   END FUNCTION pot_temp_vctr

   FUNCTION abs_temp_sclr( pThta, pPz,  pPref )                                !! :205-227
      REAL(wp), INTENT(in)           :: pThta, pPz
      REAL(wp), INTENT(in), OPTIONAL :: pPref
      REAL(wp)                       :: abs_temp_sclr
      IF( PRESENT(pPref) ) zPref_abs_s = pPref
      abs_temp_sclr = ph_s1( PH_ABS_TEMP, zPref_abs_s, 0, pThta, pPz )
   END FUNCTION abs_temp_sclr

   FUNCTION abs_temp_vctr( pThta, pPz,  pPref )                                !! :230-242
      REAL(wp), DIMENSION(:,:), INTENT(in)             :: pThta, pPz
      REAL(wp), DIMENSION(:,:), INTENT(in), OPTIONAL   :: pPref
      REAL(wp), DIMENSION(SIZE(pThta,1),SIZE(pThta,2)) :: abs_temp_vctr
      IF( PRESENT(pPref) ) THEN
         CALL ph_run( PH_ABS_TEMP, SIZE(pThta), Patm, 0, abs_temp_vctr, pThta, pPz, pPref )
      ELSE
         CALL ph_run( PH_ABS_TEMP, SIZE(pThta), Patm, 0, abs_temp_vctr, pThta, pPz )
      END IF
   END FUNCTION abs_temp_vctr

   FUNCTION virt_temp_sclr( pTa, pqa )                                         !! :247-269
      REAL(wp)             :: virt_temp_sclr
      REAL(wp), INTENT(in) :: pTa, pqa
      virt_temp_sclr = ph_s1( PH_VIRT_TEMP, 0._wp, 0, pTa, pqa )
   END FUNCTION virt_temp_sclr

   FUNCTION virt_temp_vctr( pTa, pqa )                                         !! :271-276
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pTa, pqa
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: virt_temp_vctr
      CALL ph_run( PH_VIRT_TEMP, SIZE(pTa), 0._wp, 0, virt_temp_vctr, pTa, pqa )
   END FUNCTION virt_temp_vctr


   !! =========================================================================================== pressure / temperature at height
   FUNCTION Pz_from_P0_tz_qz_sclr( pz, pslp, pTa, pqa,  l_ice )                !! :283-318
      REAL(wp), INTENT(in)           :: pz, pslp, pTa, pqa
      LOGICAL , INTENT(in), OPTIONAL :: l_ice
      REAL(wp)                       :: Pz_from_P0_tz_qz_sclr
      IF( PRESENT(l_ice) ) lice_pz_s = l_ice
      Pz_from_P0_tz_qz_sclr = ph_s1( PH_PZ, pz, MERGE(1,0,lice_pz_s), pslp, pTa, pqa )
   END FUNCTION Pz_from_P0_tz_qz_sclr

   FUNCTION Pz_from_P0_tz_qz_vctr( pz, pslp, pTa, pqa,  l_ice )                !! :320-337
      REAL(wp),                 INTENT(in) :: pz
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pslp, pTa, pqa
      LOGICAL , OPTIONAL      , INTENT(in) :: l_ice
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: Pz_from_P0_tz_qz_vctr
      IF( PRESENT(l_ice) ) lice_pz_v = l_ice
      IF( SIZE(pTa) > 0 ) lice_pz_s = lice_pz_v   ! the reference passes l_ice=lice to the scalar version for every cell
      CALL ph_run( PH_PZ, SIZE(pTa), pz, MERGE(1,0,lice_pz_v), Pz_from_P0_tz_qz_vctr, pslp, pTa, pqa )
   END FUNCTION Pz_from_P0_tz_qz_vctr

   FUNCTION Theta_from_z_P0_T_q_sclr( pz, pslp, pTa, pqa )                     !! :343-365
      REAL(wp), INTENT(in) :: pz, pslp, pTa, pqa
      REAL(wp)             :: Theta_from_z_P0_T_q_sclr
      zPref_pot_s = pslp                                                       ! pot_temp_sclr( ..., pPref=pslp )
      Theta_from_z_P0_T_q_sclr = ph_s1( PH_THETA, pz, MERGE(1,0,lice_pz_s), pslp, pTa, pqa )
   END FUNCTION Theta_from_z_P0_T_q_sclr

   FUNCTION Theta_from_z_P0_T_q_vctr( pz, pslp, pTa, pqa )                     !! :367-375
      REAL(wp),                 INTENT(in) :: pz
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pslp, pTa, pqa
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: Theta_from_z_P0_T_q_vctr
      IF( SIZE(pTa) > 0 ) lice_pz_s = lice_pz_v
      CALL ph_run( PH_THETA, SIZE(pTa), pz, MERGE(1,0,lice_pz_v), Theta_from_z_P0_T_q_vctr, pslp, pTa, pqa )
   END FUNCTION Theta_from_z_P0_T_q_vctr

   FUNCTION T_from_z_P0_Theta_q_sclr( pz, pslp, pThta, pqa )                   !! :380-407
      REAL(wp), INTENT(in) :: pz, pslp, pThta, pqa
      REAL(wp)             :: T_from_z_P0_Theta_q_sclr
      zPref_abs_s = pslp                                                       ! abs_temp_sclr( ..., pPref=pslp )
      T_from_z_P0_Theta_q_sclr = ph_s1( PH_TABS, pz, MERGE(1,0,lice_pz_s), pslp, pThta, pqa )
   END FUNCTION T_from_z_P0_Theta_q_sclr

   FUNCTION T_from_z_P0_Theta_q_vctr( pz, pslp, pThta, pqa )                   !! :409-421
      REAL(wp),                 INTENT(in) :: pz
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pslp, pThta, pqa
      REAL(wp), DIMENSION(SIZE(pslp,1),SIZE(pslp,2)) :: T_from_z_P0_Theta_q_vctr
      IF( SIZE(pslp) > 0 ) zPref_abs_s = pslp(SIZE(pslp,1),SIZE(pslp,2))      ! the cell loop ends on the last cell
      CALL ph_run( PH_TABS, SIZE(pslp), pz, MERGE(1,0,lice_pz_s), T_from_z_P0_Theta_q_vctr, pslp, pThta, pqa )
   END FUNCTION T_from_z_P0_Theta_q_vctr


   !! ================================================================================================================ air properties
   FUNCTION rho_air_sclr( pTa, pqa, pslp )                                     !! :522-537
      REAL(wp), INTENT(in) :: pTa, pqa, pslp
      REAL(wp)             :: rho_air_sclr
      rho_air_sclr = ph_s1( PH_RHO_AIR, 0._wp, 0, pTa, pqa, pslp )
   END FUNCTION rho_air_sclr

   FUNCTION rho_air_vctr( pTa, pqa, pslp )                                     !! :539-546
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pTa, pqa, pslp
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: rho_air_vctr
      CALL ph_run( PH_RHO_AIR, SIZE(pTa), 0._wp, 0, rho_air_vctr, pTa, pqa, pslp )
   END FUNCTION rho_air_vctr

   FUNCTION visc_air_sclr(pTa)                                                 !! :549-563
      REAL(wp)             :: visc_air_sclr
      REAL(wp), INTENT(in) :: pTa
      visc_air_sclr = ph_s1( PH_VISC_AIR, 0._wp, 0, pTa )
   END FUNCTION visc_air_sclr

   FUNCTION visc_air_vctr(pTa)                                                 !! :565-574
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pTa
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: visc_air_vctr
      CALL ph_run( PH_VISC_AIR, SIZE(pTa), 0._wp, 0, visc_air_vctr, pTa )
   END FUNCTION visc_air_vctr

   FUNCTION L_vap_sclr( psst )                                                 !! :579-592
      REAL(wp)             :: L_vap_sclr
      REAL(wp), INTENT(in) :: psst
      L_vap_sclr = ph_s1( PH_L_VAP, 0._wp, 0, psst )
   END FUNCTION L_vap_sclr

   FUNCTION L_vap_vctr( psst )                                                 !! :594-598
      REAL(wp), DIMENSION(:,:), INTENT(in)           :: psst
      REAL(wp), DIMENSION(SIZE(psst,1),SIZE(psst,2)) :: L_vap_vctr
      CALL ph_run( PH_L_VAP, SIZE(psst), 0._wp, 0, L_vap_vctr, psst )
   END FUNCTION L_vap_vctr

   FUNCTION cp_air_sclr( pqa )                                                 !! :603-616
      REAL(wp), INTENT(in) :: pqa
      REAL(wp)             :: cp_air_sclr
      cp_air_sclr = ph_s1( PH_CP_AIR, 0._wp, 0, pqa )
   END FUNCTION cp_air_sclr

   FUNCTION cp_air_vctr( pqa )                                                 !! :618-622
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pqa
      REAL(wp), DIMENSION(SIZE(pqa,1),SIZE(pqa,2)) :: cp_air_vctr
      CALL ph_run( PH_CP_AIR, SIZE(pqa), 0._wp, 0, cp_air_vctr, pqa )
   END FUNCTION cp_air_vctr

   FUNCTION gamma_moist_sclr( pTa, pqa )                                       !! :627-649
      REAL(wp)             :: gamma_moist_sclr
      REAL(wp), INTENT(in) :: pTa, pqa
      gamma_moist_sclr = ph_s1( PH_GAMMA_MOIST, 0._wp, 0, pTa, pqa )
   END FUNCTION gamma_moist_sclr

   FUNCTION gamma_moist_vctr( pTa, pqa )                                       !! :651-661
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pTa, pqa
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: gamma_moist_vctr
      CALL ph_run( PH_GAMMA_MOIST, SIZE(pTa), 0._wp, 0, gamma_moist_vctr, pTa, pqa )
   END FUNCTION gamma_moist_vctr


   !! ====================================================================================================================== stability
   FUNCTION One_on_L_sclr( pThta, pqa, pus, pts, pqs )                         !! :666-693
      REAL(wp)             :: One_on_L_sclr
      REAL(wp), INTENT(in) :: pThta, pqa, pus, pts, pqs
      One_on_L_sclr = ph_s1( PH_ONE_ON_L, 0._wp, 0, pThta, pqa, pus, pts, pqs )
   END FUNCTION One_on_L_sclr

   FUNCTION One_on_L_vctr( pThta, pqa, pus, pts, pqs )                         !! :695-708
      REAL(wp), DIMENSION(:,:), INTENT(in)             :: pThta, pqa, pus, pts, pqs
      REAL(wp), DIMENSION(SIZE(pThta,1),SIZE(pThta,2)) :: One_on_L_vctr
      CALL ph_run( PH_ONE_ON_L, SIZE(pThta), 0._wp, 0, One_on_L_vctr, pThta, pqa, pus, pts, pqs )
   END FUNCTION One_on_L_vctr

   FUNCTION Ri_bulk_sclr( pz, psst, pThta, pssq, pqa, pub,  pTa_layer, pqa_layer )   !! :712-747
      REAL(wp)             :: Ri_bulk_sclr
      REAL(wp), INTENT(in) :: pz, psst, pThta, pssq, pqa, pub
      REAL(wp), INTENT(in), OPTIONAL :: pTa_layer, pqa_layer
      IF( PRESENT(pTa_layer) .AND. PRESENT(pqa_layer) ) THEN
         Ri_bulk_sclr = ph_s1( PH_RI_BULK, pz, 0, psst, pThta, pssq, pqa, pub, pTa_layer, pqa_layer )
      ELSE
         Ri_bulk_sclr = ph_s1( PH_RI_BULK, pz, 0, psst, pThta, pssq, pqa, pub )
      END IF
   END FUNCTION Ri_bulk_sclr

   FUNCTION Ri_bulk_vctr( pz, psst, pThta, pssq, pqa, pub,  pTa_layer, pqa_layer )   !! :749-772
      REAL(wp)                , INTENT(in) :: pz
      REAL(wp), DIMENSION(:,:), INTENT(in) :: psst, pThta, pssq, pqa, pub
      REAL(wp), DIMENSION(:,:), INTENT(in), OPTIONAL :: pTa_layer, pqa_layer
      REAL(wp), DIMENSION(SIZE(psst,1),SIZE(psst,2)) :: Ri_bulk_vctr
      IF( PRESENT(pTa_layer) .AND. PRESENT(pqa_layer) ) THEN
         CALL ph_run( PH_RI_BULK, SIZE(psst), pz, 0, Ri_bulk_vctr, psst, pThta, pssq, pqa, pub, pTa_layer, pqa_layer )
      ELSE
         CALL ph_run( PH_RI_BULK, SIZE(psst), pz, 0, Ri_bulk_vctr, psst, pThta, pssq, pqa, pub )
      END IF
   END FUNCTION Ri_bulk_vctr


   !! =============================================================================================================== saturation
   FUNCTION e_sat_sclr( pTa )                                                  !! :777-800
      REAL(wp)             :: e_sat_sclr
      REAL(wp), INTENT(in) :: pTa
      e_sat_sclr = ph_s1( PH_E_SAT, 0._wp, 0, pTa )
   END FUNCTION e_sat_sclr

   FUNCTION e_sat_vctr(pTa)                                                    !! :802-811
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pTa
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: e_sat_vctr
      CALL ph_run( PH_E_SAT, SIZE(pTa), 0._wp, 0, e_sat_vctr, pTa )
   END FUNCTION e_sat_vctr

   FUNCTION e_sat_ice_sclr(pTa)                                                !! :815-830
      REAL(wp)             :: e_sat_ice_sclr
      REAL(wp), INTENT(in) :: pTa
      e_sat_ice_sclr = ph_s1( PH_E_SAT_ICE, 0._wp, 0, pTa )
   END FUNCTION e_sat_ice_sclr

   FUNCTION e_sat_ice_vctr(pTa)                                                !! :832-843
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pTa
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: e_sat_ice_vctr
      CALL ph_run( PH_E_SAT_ICE, SIZE(pTa), 0._wp, 0, e_sat_ice_vctr, pTa )
   END FUNCTION e_sat_ice_vctr

   FUNCTION de_sat_dt_ice_sclr(pTa)                                            !! :845-861
      REAL(wp)             :: de_sat_dt_ice_sclr
      REAL(wp), INTENT(in) :: pTa
      de_sat_dt_ice_sclr = ph_s1( PH_DE_SAT_DT_ICE, 0._wp, 0, pTa )
   END FUNCTION de_sat_dt_ice_sclr

   FUNCTION de_sat_dt_ice_vctr(pTa)                                            !! :863-875
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pTa
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: de_sat_dt_ice_vctr
      CALL ph_run( PH_DE_SAT_DT_ICE, SIZE(pTa), 0._wp, 0, de_sat_dt_ice_vctr, pTa )
   END FUNCTION de_sat_dt_ice_vctr

   FUNCTION q_sat_sclr( pTa, pslp,  l_ice )                                    !! :881-904
      REAL(wp) :: q_sat_sclr
      REAL(wp), INTENT(in) :: pTa, pslp
      LOGICAL,  INTENT(in), OPTIONAL :: l_ice
      LOGICAL  :: lice
      lice = .FALSE.
      IF( PRESENT(l_ice) ) lice = l_ice
      q_sat_sclr = ph_s1( PH_Q_SAT, 0._wp, MERGE(1,0,lice), pTa, pslp )
   END FUNCTION q_sat_sclr

   FUNCTION q_sat_vctr( pTa, pslp,  l_ice )                                    !! :906-921
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pTa, pslp
      LOGICAL,  INTENT(in), OPTIONAL :: l_ice
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: q_sat_vctr
      LOGICAL  :: lice
      lice = .FALSE.
      IF( PRESENT(l_ice) ) lice = l_ice
      CALL ph_run( PH_Q_SAT, SIZE(pTa), 0._wp, MERGE(1,0,lice), q_sat_vctr, pTa, pslp )
   END FUNCTION q_sat_vctr

   FUNCTION dq_sat_dt_ice_sclr( pTa, pslp )                                    !! :926-945
      REAL(wp) :: dq_sat_dt_ice_sclr
      REAL(wp), INTENT(in) :: pTa, pslp
      dq_sat_dt_ice_sclr = ph_s1( PH_DQ_SAT_DT_ICE, 0._wp, 0, pTa, pslp )
   END FUNCTION dq_sat_dt_ice_sclr

   FUNCTION dq_sat_dt_ice_vctr( pTa, pslp )                                    !! :947-958
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pTa, pslp
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: dq_sat_dt_ice_vctr
      CALL ph_run( PH_DQ_SAT_DT_ICE, SIZE(pTa), 0._wp, 0, dq_sat_dt_ice_vctr, pTa, pslp )
   END FUNCTION dq_sat_dt_ice_vctr

   FUNCTION q_air_rh(prha, pTa, pslp)                                          !! :963-985
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: prha, pTa, pslp
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: q_air_rh
      CALL ph_run( PH_Q_AIR_RH, SIZE(pTa), 0._wp, 0, q_air_rh, prha, pTa, pslp )
   END FUNCTION q_air_rh

   FUNCTION q_air_dp(da, slp)                                                  !! :990-1000
      REAL(wp), DIMENSION(:,:), INTENT(in)       :: da, slp
      REAL(wp), DIMENSION(SIZE(da,1),SIZE(da,2)) :: q_air_dp
      CALL ph_run( PH_Q_AIR_DP, SIZE(da), 0._wp, 0, q_air_dp, da, slp )
   END FUNCTION q_air_dp

   FUNCTION rho_air_adv(pTa, pqa, pslp)                                        !! :1008-1024
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pTa, pqa, pslp
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: rho_air_adv
      CALL ph_run( PH_RHO_AIR_ADV, SIZE(pTa), 0._wp, 0, rho_air_adv, pTa, pqa, pslp )
   END FUNCTION rho_air_adv

   FUNCTION q_sat_crude(pts, prhoa)                                            !! :1029-1038
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pts, prhoa
      REAL(wp), DIMENSION(SIZE(pts,1),SIZE(pts,2)) :: q_sat_crude
      CALL ph_run( PH_Q_SAT_CRUDE, SIZE(pts), 0._wp, 0, q_sat_crude, pts, prhoa )
   END FUNCTION q_sat_crude

   FUNCTION dry_static_energy( pz, pTa, pqa )                                  !! :1043-1054
      REAL(wp)                , INTENT(in) :: pz
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pTa, pqa
      REAL(wp), DIMENSION(SIZE(pTa,1),SIZE(pTa,2)) :: dry_static_energy
      CALL ph_run( PH_DRY_STATIC_ENERGY, SIZE(pTa), pz, 0, dry_static_energy, pTa, pqa )
   END FUNCTION dry_static_energy

   FUNCTION e_air(pqa, pslp)                                                   !! :1706-1736
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pqa, pslp
      REAL(wp), DIMENSION(SIZE(pqa,1),SIZE(pqa,2)) :: e_air
      CALL ph_run( PH_E_AIR, SIZE(pqa), 0._wp, 0, e_air, pqa, pslp )
   END FUNCTION e_air

   FUNCTION rh_air(pqa, pTa, pslp)                                             !! :1741-1753
      REAL(wp), DIMENSION(:,:), INTENT(in)         :: pqa, pTa, pslp
      REAL(wp), DIMENSION(SIZE(pqa,1),SIZE(pqa,2)) :: rh_air
      CALL ph_run( PH_RH_AIR, SIZE(pqa), 0._wp, 0, rh_air, pqa, pTa, pslp )
   END FUNCTION rh_air


   !! ==================================================================================================================== fluxes
   SUBROUTINE UPDATE_QNSOL_TAU_SCLR( pzu, pts, pqs, pThta, pqa, pust, ptst, pqst, pwnd, pUb, pslp, prlw, &
      &                              pQns, pTau,    Qlat )                     !! :1059-1103
      REAL(wp), INTENT(in)  :: pzu, pts, pqs, pThta, pqa, pust, ptst, pqst, pwnd, pUb, pslp, prlw
      REAL(wp), INTENT(out) :: pQns, pTau
      REAL(wp), OPTIONAL, INTENT(out) :: Qlat
      REAL(wp), DIMENSION(1) :: z1, z2, z3
      CALL ph_run( PH_UPDATE_QNSOL_TAU, 1, pzu, 0, z1, (/pts/), (/pqs/), (/pThta/), (/pqa/), (/pust/), (/ptst/), (/pqst/), &
         &         (/pwnd/), (/pUb/), (/pslp/), (/prlw/), o2=z2, o3=z3 )
      pQns = z1(1) ; pTau = z2(1)
      IF( PRESENT(Qlat) ) Qlat = z3(1)
   END SUBROUTINE UPDATE_QNSOL_TAU_SCLR

   SUBROUTINE UPDATE_QNSOL_TAU_VCTR( pzu, pts, pqs, pThta, pqa, pust, ptst, pqst, pwnd, pUb, pslp, prlw, &
      &                              pQns, pTau,    Qlat)                      !! :1105-1144
      REAL(wp),                 INTENT(in)  :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in)  :: pts, pqs, pThta, pqa, pust, ptst, pqst, pwnd, pUb, pslp, prlw
      REAL(wp), DIMENSION(:,:), INTENT(out) :: pQns, pTau
      REAL(wp), DIMENSION(:,:), OPTIONAL, INTENT(out) :: Qlat
      REAL(wp), DIMENSION(:,:), ALLOCATABLE :: zns, ztau, zlat
      ALLOCATE( zns(SIZE(pts,1),SIZE(pts,2)), ztau(SIZE(pts,1),SIZE(pts,2)), zlat(SIZE(pts,1),SIZE(pts,2)) )
      CALL ph_run( PH_UPDATE_QNSOL_TAU, SIZE(pts), pzu, 0, zns, pts, pqs, pThta, pqa, pust, ptst, pqst, pwnd, pUb, pslp, prlw, &
         &         o2=ztau, o3=zlat )
      pQns = zns ; pTau = ztau
      IF( PRESENT(Qlat) ) Qlat = zlat
      DEALLOCATE( zns, ztau, zlat )
   END SUBROUTINE UPDATE_QNSOL_TAU_VCTR

   SUBROUTINE BULK_FORMULA_SCLR( pzu, pts, pqs, pThta, pqa, &
      &                          pCd, pCh, pCe,            &
      &                          pwnd, pUb, pslp,          &
      &                          pTau, pQsen, pQlat,       &
      &                          pEvap, prhoa, l_ice      )                    !! :1149-1203
      REAL(wp), INTENT(in)  :: pzu, pts, pqs, pThta, pqa, pCd, pCh, pCe, pwnd, pUb, pslp
      REAL(wp), INTENT(out) :: pTau, pQsen, pQlat
      REAL(wp), INTENT(out), OPTIONAL :: pEvap, prhoa
      LOGICAL,  INTENT(in),  OPTIONAL :: l_ice
      REAL(wp), DIMENSION(1) :: z1, z2, z3, z4, z5
      LOGICAL  :: lice
      lice = .FALSE.
      IF( PRESENT(l_ice) ) lice = l_ice
      CALL ph_run( PH_BULK_FORMULA, 1, pzu, MERGE(1,0,lice), z1, (/pts/), (/pqs/), (/pThta/), (/pqa/), (/pCd/), (/pCh/), (/pCe/), &
         &         (/pwnd/), (/pUb/), (/pslp/), o2=z2, o3=z3, o4=z4, o5=z5, tolerate=8 )   ! (the scalar version has no stress check)
      pTau = z1(1) ; pQsen = z2(1) ; pQlat = z3(1)
      IF( PRESENT(pEvap) ) pEvap = z4(1)
      IF( PRESENT(prhoa) ) prhoa = z5(1)
   END SUBROUTINE BULK_FORMULA_SCLR

   SUBROUTINE BULK_FORMULA_VCTR( pzu, pts, pqs, pThta, pqa, &
      &                          pCd, pCh, pCe,           &
      &                          pwnd, pUb, pslp,         &
      &                          pTau, pQsen, pQlat,      &
      &                          pEvap, prhoa, l_ice )                         !! :1205-1261
      REAL(wp),                 INTENT(in)  :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in)  :: pts, pqs, pThta, pqa, pCd, pCh, pCe, pwnd, pUb, pslp
      REAL(wp), DIMENSION(:,:), INTENT(out) :: pTau, pQsen, pQlat
      REAL(wp), DIMENSION(:,:), INTENT(out), OPTIONAL :: pEvap, prhoa
      LOGICAL,  INTENT(in),  OPTIONAL :: l_ice
      REAL(wp), DIMENSION(:,:), ALLOCATABLE :: z1, z2, z3, z4, z5
      REAL(C_DOUBLE), DIMENSION(2) :: zinfo
      INTEGER :: ji, jj, nx, ny
      CHARACTER(len=256) :: cmsg
      nx = SIZE(pts,1) ; ny = SIZE(pts,2)
      ALLOCATE( z1(nx,ny), z2(nx,ny), z3(nx,ny), z4(nx,ny), z5(nx,ny) )
      CALL ph_run( PH_BULK_FORMULA, SIZE(pts), pzu, MERGE(1,0,PRESENT(l_ice)), z1, pts, pqs, pThta, pqa, pCd, pCh, pCe, pwnd, pUb, pslp, &
         &         o2=z2, o3=z3, o4=z4, o5=z5, info=zinfo, tolerate=8 )
      pTau = z1 ; pQsen = z2 ; pQlat = z3
      IF( PRESENT(pEvap) ) pEvap = z4
      IF( PRESENT(prhoa) ) prhoa = z5
      DEALLOCATE( z1, z2, z3, z4, z5 )
      IF( zinfo(1) >= 0._C_DOUBLE ) THEN       ! first cell (in memory order, the reference's loop order) beyond ref_tau_max
         jj = INT(zinfo(1)) / nx + 1
         ji = INT(zinfo(1)) - (jj-1)*nx + 1
         WRITE(cmsg,'(" => ",f8.2," N/m^2 ! At ji, jj = ", i4.4,", ",i4.4)') pTau(ji,jj), ji, jj
         CALL ctl_stop( 'BULK_FORMULA_VCTR()@mod_phymbl: wind stress too strong!', cmsg )
      END IF
   END SUBROUTINE BULK_FORMULA_VCTR

   FUNCTION alpha_sw_sclr( psst )                                              !! :1267-1280
      REAL(wp), INTENT(in) :: psst
      REAL(wp)             :: alpha_sw_sclr
      alpha_sw_sclr = ph_s1( PH_ALPHA_SW, 0._wp, 0, psst )
   END FUNCTION alpha_sw_sclr

   FUNCTION alpha_sw_vctr( psst )                                              !! :1282-1286
      REAL(wp), DIMENSION(:,:), INTENT(in)           :: psst
      REAL(wp), DIMENSION(SIZE(psst,1),SIZE(psst,2)) :: alpha_sw_vctr
      CALL ph_run( PH_ALPHA_SW, SIZE(psst), 0._wp, 0, alpha_sw_vctr, psst )
   END FUNCTION alpha_sw_vctr

   FUNCTION qlw_net_sclr( pdwlw, pts,  l_ice )                                 !! :1291-1314
      REAL(wp) :: qlw_net_sclr
      REAL(wp), INTENT(in) :: pdwlw, pts
      LOGICAL,  INTENT(in), OPTIONAL :: l_ice
      LOGICAL  :: lice
      lice = .FALSE.
      IF( PRESENT(l_ice) ) lice = l_ice
      qlw_net_sclr = ph_s1( PH_QLW_NET, 0._wp, MERGE(1,0,lice), pdwlw, pts )
   END FUNCTION qlw_net_sclr

   FUNCTION qlw_net_vctr( pdwlw, pts,  l_ice )                                 !! :1316-1330
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pdwlw, pts
      REAL(wp), DIMENSION(SIZE(pts,1),SIZE(pts,2)) :: qlw_net_vctr
      LOGICAL,  INTENT(in), OPTIONAL :: l_ice
      LOGICAL  :: lice
      lice = .FALSE.
      IF( PRESENT(l_ice) ) lice = l_ice
      CALL ph_run( PH_QLW_NET, SIZE(pts), 0._wp, MERGE(1,0,lice), qlw_net_vctr, pdwlw, pts )
   END FUNCTION qlw_net_vctr


   !! ======================================================================================================== roughness, neutral wind
   FUNCTION z0_from_Cd_sclr( pzu, pCd,  ppsi )                                 !! :1335-1352
      REAL(wp)                       :: z0_from_Cd_sclr
      REAL(wp), INTENT(in)           :: pzu, pCd
      REAL(wp), INTENT(in), OPTIONAL :: ppsi
      IF( PRESENT(ppsi) ) THEN
         z0_from_Cd_sclr = ph_s1( PH_Z0_FROM_CD, pzu, 0, pCd, ppsi )
      ELSE
         z0_from_Cd_sclr = ph_s1( PH_Z0_FROM_CD, pzu, 0, pCd )
      END IF
   END FUNCTION z0_from_Cd_sclr

   FUNCTION z0_from_Cd_vctr( pzu, pCd,  ppsi )                                 !! :1354-1366
      REAL(wp)                , INTENT(in) :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pCd
      REAL(wp), DIMENSION(:,:), INTENT(in), OPTIONAL :: ppsi
      REAL(wp), DIMENSION(SIZE(pCd,1),SIZE(pCd,2)) :: z0_from_Cd_vctr
      IF( PRESENT(ppsi) ) THEN
         CALL ph_run( PH_Z0_FROM_CD, SIZE(pCd), pzu, 0, z0_from_Cd_vctr, pCd, ppsi )
      ELSE
         CALL ph_run( PH_Z0_FROM_CD, SIZE(pCd), pzu, 0, z0_from_Cd_vctr, pCd )
      END IF
   END FUNCTION z0_from_Cd_vctr

   FUNCTION z0_from_ustar_sclr( pzu, pus, puzu )                               !! :1371-1380
      REAL(wp)             :: z0_from_ustar_sclr
      REAL(wp), INTENT(in) :: pzu, pus, puzu
      z0_from_ustar_sclr = ph_s1( PH_Z0_FROM_USTAR, pzu, 0, pus, puzu )
   END FUNCTION z0_from_ustar_sclr

   FUNCTION z0_from_ustar_vctr( pzu, pus, puzu )                               !! :1382-1391
      REAL(wp)                , INTENT(in) :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pus, puzu
      REAL(wp), DIMENSION(SIZE(pus,1),SIZE(pus,2)) :: z0_from_ustar_vctr
      CALL ph_run( PH_Z0_FROM_USTAR, SIZE(pus), pzu, 0, z0_from_ustar_vctr, pus, puzu )
   END FUNCTION z0_from_ustar_vctr

   FUNCTION Cd_from_z0( pzu, pz0,  ppsi )                                      !! :1396-1414
      REAL(wp)                , INTENT(in) :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pz0
      REAL(wp), DIMENSION(:,:), INTENT(in), OPTIONAL :: ppsi
      REAL(wp), DIMENSION(SIZE(pz0,1),SIZE(pz0,2)) :: Cd_from_z0
      IF( PRESENT(ppsi) ) THEN
         CALL ph_run( PH_CD_FROM_Z0, SIZE(pz0), pzu, 0, Cd_from_z0, pz0, ppsi )
      ELSE
         CALL ph_run( PH_CD_FROM_Z0, SIZE(pz0), pzu, 0, Cd_from_z0, pz0 )
      END IF
   END FUNCTION Cd_from_z0

   FUNCTION f_m_louis_sclr( pzu, pRib, pCdn, pz0 )                             !! :1419-1440
      REAL(wp)             :: f_m_louis_sclr
      REAL(wp), INTENT(in) :: pzu, pRib, pCdn, pz0
      f_m_louis_sclr = ph_s1( PH_F_M_LOUIS, pzu, 0, pRib, pCdn, pz0 )
   END FUNCTION f_m_louis_sclr

   FUNCTION f_m_louis_vctr( pzu, pRib, pCdn, pz0 )                             !! :1442-1453
      REAL(wp),                 INTENT(in) :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pRib, pCdn, pz0
      REAL(wp), DIMENSION(SIZE(pz0,1),SIZE(pz0,2)) :: f_m_louis_vctr
      CALL ph_run( PH_F_M_LOUIS, SIZE(pz0), pzu, 0, f_m_louis_vctr, pRib, pCdn, pz0 )
   END FUNCTION f_m_louis_vctr

   FUNCTION f_h_louis_sclr( pzu, pRib, pChn, pz0 )                             !! :1458-1479
      REAL(wp)             :: f_h_louis_sclr
      REAL(wp), INTENT(in) :: pzu, pRib, pChn, pz0
      f_h_louis_sclr = ph_s1( PH_F_H_LOUIS, pzu, 0, pRib, pChn, pz0 )
   END FUNCTION f_h_louis_sclr

   FUNCTION f_h_louis_vctr( pzu, pRib, pChn, pz0 )                             !! :1481-1492
      REAL(wp),                 INTENT(in) :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pRib, pChn, pz0
      REAL(wp), DIMENSION(SIZE(pz0,1),SIZE(pz0,2)) :: f_h_louis_vctr
      CALL ph_run( PH_F_H_LOUIS, SIZE(pz0), pzu, 0, f_h_louis_vctr, pRib, pChn, pz0 )
   END FUNCTION f_h_louis_vctr

   FUNCTION UN10_from_ustar( pzu, pUzu, pus, ppsi )                            !! :1498-1510
      REAL(wp),                 INTENT(in) :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pUzu, pus, ppsi
      REAL(wp), DIMENSION(SIZE(pUzu,1),SIZE(pUzu,2)) :: UN10_from_ustar
      CALL ph_run( PH_UN10_FROM_USTAR, SIZE(pUzu), pzu, 0, UN10_from_ustar, pUzu, pus, ppsi )
   END FUNCTION UN10_from_ustar

   FUNCTION UN10_from_CDN( pzu, pUb, pCdn, ppsi )                              !! :1515-1527
      REAL(wp),                 INTENT(in) :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pUb, pCdn, ppsi
      REAL(wp), DIMENSION(SIZE(pUb,1),SIZE(pUb,2)) :: UN10_from_CDN
      CALL ph_run( PH_UN10_FROM_CDN, SIZE(pUb), pzu, 0, UN10_from_CDN, pUb, pCdn, ppsi )
   END FUNCTION UN10_from_CDN

   FUNCTION UN10_from_CD_sclr( pzu, pUb, pCd, ppsi )                           !! :1532-1547
      REAL(wp)             :: UN10_from_CD_sclr
      REAL(wp), INTENT(in) :: pzu, pUb, pCd, ppsi
      UN10_from_CD_sclr = ph_s1( PH_UN10_FROM_CD, pzu, 0, pUb, pCd, ppsi )
   END FUNCTION UN10_from_CD_sclr

   FUNCTION UN10_from_CD_vctr( pzu, pUb, pCd, ppsi )                           !! :1549-1558
      REAL(wp),                 INTENT(in) :: pzu
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pUb, pCd, ppsi
      REAL(wp), DIMENSION(SIZE(pUb,1),SIZE(pUb,2)) :: UN10_from_CD_vctr
      CALL ph_run( PH_UN10_FROM_CD, SIZE(pUb), pzu, 0, UN10_from_CD_vctr, pUb, pCd, ppsi )
   END FUNCTION UN10_from_CD_vctr

   FUNCTION z0tq_LKB( iflag, pRer, pz0 )                                       !! :1635-1701
      INTEGER,                  INTENT(in) :: iflag
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pRer, pz0
      REAL(wp), DIMENSION(SIZE(pRer,1),SIZE(pRer,2)) :: z0tq_LKB
      CALL ph_run( PH_Z0TQ_LKB, SIZE(pRer), 0._wp, iflag, z0tq_LKB, pRer, pz0 )
   END FUNCTION z0tq_LKB

   FUNCTION delta_skin_layer_sclr( palpha, pQd, pustar_a,  Qlat )              !! :2010-2046
      REAL(wp),           INTENT(in) :: palpha, pQd, pustar_a
      REAL(wp), OPTIONAL, INTENT(in) :: Qlat
      REAL(wp)                       :: delta_skin_layer_sclr
      IF( PRESENT(Qlat) ) THEN
         delta_skin_layer_sclr = ph_s1( PH_DELTA_SKIN, 0._wp, 0, palpha, pQd, pustar_a, Qlat )
      ELSE
         delta_skin_layer_sclr = ph_s1( PH_DELTA_SKIN, 0._wp, 0, palpha, pQd, pustar_a )
      END IF
   END FUNCTION delta_skin_layer_sclr


   !! ================================================================== host-side bookkeeping of the drivers (no flux arithmetic)
   FUNCTION VARIANCE( pvc )                                                    !! :1794-1806 (a standard deviation, as there)
      REAL(4)                            :: VARIANCE
      REAL(wp), DIMENSION(:), INTENT(in) :: pvc
      REAL(wp) :: zm
      zm = SUM(pvc)/SIZE(pvc)
      VARIANCE = REAL( SQRT( SUM( (pvc - zm)*(pvc - zm) ) / SIZE(pvc) ), 4 )
   END FUNCTION VARIANCE

   FUNCTION VMEAN( pvc )                                                       !! :1811-1820
      REAL(4)                            :: VMEAN
      REAL(wp), DIMENSION(:), INTENT(in) :: pvc
      VMEAN = SUM(pvc)/SIZE(pvc)
   END FUNCTION VMEAN

   SUBROUTINE TO_KELVIN_3D( pt, cname )                                        !! :1826-1848
      REAL(wp), DIMENSION(:,:,:), INTENT(inout) :: pt
      CHARACTER(len=*), OPTIONAL, INTENT(in)    :: cname
      CHARACTER(len=32), SAVE :: cvar = '...'
      REAL(wp) :: zm
      IF( PRESENT(cname) ) cvar = TRIM(cname)
      zm = SUM(pt)/REAL(SIZE(pt))
      IF( (zm < 50._wp) .AND. (zm > -80._wp) ) THEN
         PRINT *, ' *** Variable ', TRIM(cvar), ' is in [deg.C] => converting to [K] !!!'
         pt = pt + rt0
      ELSEIF( (zm > 200._wp) .AND. (zm < 320._wp) ) THEN
         PRINT *, ' *** Variable ', TRIM(cvar), ' is already in [K], doing nothing...'
      ELSE
         PRINT *, ' *** PROBLEM: cannot figure out unit of variable ', TRIM(cvar), ' !!!'
         STOP
      END IF
   END SUBROUTINE TO_KELVIN_3D

   SUBROUTINE check_unit_consistency( cfield, Xval,  mask )                    !! :1851-1954
      !! min / max / mean of the unmasked cells against the admissible range of that kind of field (mod_const)
      CHARACTER(len=*),                     INTENT(in) :: cfield
      REAL(wp),   DIMENSION(:,:),           INTENT(in) :: Xval
      INTEGER(1), DIMENSION(:,:), OPTIONAL, INTENT(in) :: mask
      LOGICAL,  DIMENSION(:,:), ALLOCATABLE :: lmask
      REAL(wp), DIMENSION(:,:), ALLOCATABLE :: zw
      REAL(wp) :: zmean, zlo, zhi
      CHARACTER(len=64) :: cunit
      ALLOCATE( lmask(SIZE(Xval,1),SIZE(Xval,2)), zw(SIZE(Xval,1),SIZE(Xval,2)) )
      lmask = .TRUE. ; zw = 1._wp
      IF( PRESENT(mask) ) THEN
         IF( ANY( SHAPE(mask) /= SHAPE(Xval) ) ) THEN
            WRITE(*,'(" *** ERROR (check_unit_consistency@mod_phymbl): shape of `mask` does not agree with array of field ",a," !")') TRIM(cfield)
            STOP
         END IF
         lmask = ( mask /= 0 ) ; zw = REAL(mask,wp)
      END IF
      zmean = SUM( Xval*zw ) / SUM( zw )
      SELECT CASE (TRIM(cfield))
      CASE('sst','SST','Ts')
         zlo = ref_sst_min ; zhi = ref_sst_max ; cunit = 'K'
      CASE('t_air','taa','t2m','T2M')
         zlo = ref_taa_min ; zhi = ref_taa_max ; cunit = 'K'
      CASE('q_air','sh','sha','q2m','Q2M')
         zlo = ref_sha_min ; zhi = ref_sha_max ; cunit = 'kg/kg'
      CASE('rh_air','rh','RH','rlh','rha')
         zlo = ref_rlh_min ; zhi = ref_rlh_max ; cunit = 'kg/kg'
      CASE('dp_air','dp','d2m','D2M')
         zlo = ref_dpt_min ; zhi = ref_dpt_max ; cunit = 'kg/kg'
      CASE('slp','mslp','MSL','msl','P')
         zlo = ref_slp_min ; zhi = ref_slp_max ; cunit = 'Pa'
      CASE('u10','v10')
         zlo = -ref_wnd_max ; zhi = ref_wnd_max ; cunit = 'm/s'
      CASE('wnd','wind','w10','W10')
         zlo = ref_wnd_min ; zhi = ref_wnd_max ; cunit = 'm/s'
      CASE('rad_sw')
         zlo = ref_rsw_min ; zhi = ref_rsw_max ; cunit = 'W/m^2'
      CASE('rad_lw')
         zlo = ref_rlw_min ; zhi = ref_rlw_max ; cunit = 'W/m^2'
      CASE DEFAULT
         WRITE(*,'(" *** ERROR (check_unit_consistency@mod_phymbl): we do not know field `",a,"` !")') TRIM(cfield)
         STOP
      END SELECT
      IF( (MAXVAL(Xval, MASK=lmask) > zhi) .OR. (MINVAL(Xval, MASK=lmask) < zlo) .OR. (zmean < zlo) .OR. (zmean > zhi) ) THEN
         WRITE(*,'(" *** ERROR (check_unit_consistency@mod_phymbl): field `",a,"` does not seem to be in ",a," !")') TRIM(cfield), TRIM(cunit)
         WRITE(*,'(" min value = ", es10.3," max value = ", es10.3," mean value = ", es10.3)') MINVAL(Xval), MAXVAL(Xval), zmean
         STOP
      END IF
      DEALLOCATE( lmask, zw )
   END SUBROUTINE check_unit_consistency

   FUNCTION type_of_humidity( Xval, mask )                                     !! :1957-2007
      !! 'sh', 'dp' or 'rh' from the mean, minimum and maximum of the unmasked cells
      REAL(wp),   DIMENSION(:,:), INTENT(in) :: Xval
      INTEGER(1), DIMENSION(:,:), INTENT(in) :: mask
      CHARACTER(len=2)                       :: type_of_humidity
      LOGICAL, DIMENSION(:,:), ALLOCATABLE :: lmask
      REAL(wp) :: zmean, zlo, zhi
      ALLOCATE( lmask(SIZE(mask,1),SIZE(mask,2)) )
      lmask = ( mask == 1 )
      zmean = SUM( Xval * REAL(mask,wp) ) / SUM( REAL(mask,wp) )
      zlo   = MINVAL( Xval, MASK=lmask )
      zhi   = MAXVAL( Xval, MASK=lmask )
      IF(     (zmean >= ref_sha_min).AND.(zmean <  ref_sha_max).AND.(zlo >= ref_sha_min).AND.(zhi <  ref_sha_max) ) THEN
         type_of_humidity = 'sh'
      ELSEIF( (zmean >= ref_dpt_min).AND.(zmean <  ref_dpt_max).AND.(zlo >= ref_dpt_min).AND.(zhi <  ref_dpt_max) ) THEN
         type_of_humidity = 'dp'
      ELSEIF( (zmean >= ref_rlh_min).AND.(zmean <= ref_rlh_max).AND.(zlo >= ref_rlh_min).AND.(zhi <= ref_rlh_max) ) THEN
         type_of_humidity = 'rh'
      ELSE
         type_of_humidity = '00'
         WRITE(6,*) 'ERROR: type_of_humidity()@mod_aerobulk_compute => un-identified humidity type!'
         WRITE(6,*) '   ==> we could not identify the humidity type based on the mean, min & max of the field:'
         WRITE(6,*) '     * mean =', REAL(zmean,4)
         WRITE(6,*) '     * min  =', REAL(zlo, 4)
         WRITE(6,*) '     * max  =', REAL(zhi, 4)
         STOP
      END IF
      DEALLOCATE( lmask )
   END FUNCTION type_of_humidity

END MODULE mod_phymbl
