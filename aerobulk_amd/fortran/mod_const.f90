! mod_const.f90 -- kinds, physical constants and run-time switches that callers of AeroBulk import with `USE mod_const`.
!
! Source-compatibility module of the MI355X-native engine: the same PUBLIC names, kinds and values as the reference's
! src/mod_const.f90 (kinds :10-12, switches :18-34, constants :38-114, calendar :125-127, sanity ranges :138-149, variable
! names :193-204, name presets :209-235, ctl_stop :238-278), so that a driver written against AeroBulk compiles unchanged
! (src/tests/example_call_aerobulk.f90:6,30 uses wp and rt0; aerobulk_toy.F90, test_cx_vs_wind.f90 ... use many more).
! The engine itself does not read this module: its constants live in aerobulk_amd/csrc/ab_physics.hpp (struct K), and
! tests/test_phymbl.py checks the two sets against each other through the compiled reference.
!
! Build like every Fortran source of AeroBulk with default reals promoted to 8 bytes (-fdefault-real-8,
! arch/make.macro_GnuLinux:17): several literals below carry no kind suffix ON PURPOSE, because the reference's do not, and a
! value such as 9.8 must be the double nearest to "9.8" exactly as it is there.

MODULE mod_const

   IMPLICIT NONE
   PUBLIC
   PRIVATE :: set_names

   INTEGER, PARAMETER :: sp = SELECTED_REAL_KIND( 6, 37), dp = SELECTED_REAL_KIND(12,307), wp = dp

   ! ---- state a caller (or AEROBULK_MODEL) sets at run time ------------------------------------------------------------
   INTEGER, PARAMETER :: jpk = 1, nit000 = 1
   INTEGER,           SAVE :: nitend = 1                     ! last time record; AEROBULK_INIT stores Nt here
   INTEGER,           SAVE :: nb_iter = 5                    ! iterations of the bulk algorithms (`Niter` of AEROBULK_MODEL)
   LOGICAL,           SAVE :: l_use_skin_schemes = .FALSE.   ! cool-skin / warm-layer in use
   CHARACTER(len=2),  SAVE :: ctype_humidity = 'sh'          ! 'sh' [kg/kg] | 'rh' [%] | 'dp' [K]
   REAL(wp), DIMENSION(jpk), SAVE :: gdept_1d = (/ 1._wp /)  ! depth of the bulk SST [m]
   REAL(wp),          SAVE :: rdt = 3600.                    ! time step of the skin schemes [s]
   LOGICAL, PARAMETER :: ldebug_blk_algos = .false.

   ! ---- geometry / planet ---------------------------------------------------------------------------------------------------
   REAL(wp), PARAMETER :: grav = 9.8, rpi = 3.141592653589793_wp, twoPi = 2.*rpi, to_rad = rpi/180.
   REAL(wp), PARAMETER :: R_earth = 6.37E6, rtilt_earth = 23.5, Sol0 = 1366.
   REAL(wp), PARAMETER :: roce_alb0 = 0.066, rice_alb0 = 0.8                 ! default albedo of the open ocean / of sea ice

   ! ---- radiation -------------------------------------------------------------------------------------------------------------
   REAL(wp), PARAMETER :: emiss_w = 0.98_wp, emiss_i = 0.996, stefan = 5.67E-8

   ! ---- water -----------------------------------------------------------------------------------------------------------------
   REAL(wp), PARAMETER :: rt0 = 273.15, rtt0 = 273.16                        ! freezing point of fresh water, triple point [K]
   REAL(wp), PARAMETER :: rCp0_w = 4190., rho0_w = 1025., rnu0_w = 1.e-6, rk0_w = 0.6

   ! ---- air -------------------------------------------------------------------------------------------------------------------
   REAL(wp), PARAMETER :: rCp0_a = 1015.0, rCp_dry = 1005.0, rCp_vap = 1860.0
   REAL(wp), PARAMETER :: R_dry = 287.05, R_vap = 461.495, R_gas = 8.314510
   REAL(wp), PARAMETER :: rmm_dryair = 28.9647e-3, rmm_water = 18.0153e-3, rmm_ratio = rmm_water / rmm_dryair
   REAL(wp), PARAMETER :: rpoiss_dry = R_dry / rCp_dry, rgamma_dry = grav / rCp_dry
   REAL(wp), PARAMETER :: reps0 = R_dry/R_vap, rctv0 = R_vap/R_dry - 1.
   REAL(wp), PARAMETER :: rnu0_air = 1.5E-5
   REAL(wp), PARAMETER :: rLevap = 2.46e+6_wp, rLsub = 2.834e+6_wp
   REAL(wp), PARAMETER :: Tswf = 273.
   REAL(wp), PARAMETER :: Patm = 101000., rho0_a = 1.2

   ! ---- bulk model ------------------------------------------------------------------------------------------------------------
   REAL(wp), PARAMETER :: vkarmn = 0.4_wp, vkarmn2 = 0.4_wp*0.4_wp
   REAL(wp), PARAMETER :: rdct_qsat_salt = 0.98_wp, z0_sea_max = 0.0025_wp, Cx_min = 0.1E-3_wp

   ! ---- skin schemes (Fairall et al. 1996, eq. 14) --------------------------------------------------------------------------
   REAL(wp), PARAMETER :: rcst_cs = -16._wp*9.80665_wp*rho0_w*rCp0_w*rnu0_w*rnu0_w*rnu0_w/(rk0_w*rk0_w)
   REAL(wp), PARAMETER :: radrw = rho0_a/rho0_w, sq_radrw = SQRT(rho0_a/rho0_w)

   ! ---- sea ice ---------------------------------------------------------------------------------------------------------------
   REAL(wp), PARAMETER :: rCd_ice = 1.4e-3_wp, to_mm_p_day = 24._wp*3600._wp, wspd_thrshld_ice = 0.2_wp

   ! ---- calendar --------------------------------------------------------------------------------------------------------------
   INTEGER, DIMENSION(12), PARAMETER :: tdmn = (/ 31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31 /)
   INTEGER, DIMENSION(12), PARAMETER :: tdml = (/ 31, 29, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31 /)

   CHARACTER(len=200), PARAMETER :: cform_err = '(" *** E R R O R :  ")'

   ! ---- admissible ranges of the input fields (AEROBULK_INIT's mask, humidity-type detection) and of the wind stress -----
   REAL(wp), PARAMETER :: ref_sst_min = 270._wp,   ref_sst_max = 320._wp      ! [K]
   REAL(wp), PARAMETER :: ref_taa_min = 180._wp,   ref_taa_max = 330._wp      ! [K]
   REAL(wp), PARAMETER :: ref_sha_min = 0._wp,     ref_sha_max = 0.08_wp      ! [kg/kg]
   REAL(wp), PARAMETER :: ref_dpt_min = 150._wp,   ref_dpt_max = 330._wp      ! [K]
   REAL(wp), PARAMETER :: ref_rlh_min = 0._wp,     ref_rlh_max = 100._wp      ! [%]
   REAL(wp), PARAMETER :: ref_slp_min = 80000._wp, ref_slp_max = 110000._wp   ! [Pa]
   REAL(wp), PARAMETER :: ref_wnd_min = 0._wp,     ref_wnd_max = 50._wp       ! [m/s]
   REAL(wp), PARAMETER :: ref_rsw_min = 0._wp,     ref_rsw_max = 1500.0_wp    ! [W/m^2]
   REAL(wp), PARAMETER :: ref_rlw_min = 0._wp,     ref_rlw_max =  750.0_wp    ! [W/m^2]
   REAL(wp), PARAMETER :: ref_tau_max = 10._wp                                ! [N/m^2]

   ! ---- names of the input variables in the drivers' files --------------------------------------------------------------
   CHARACTER(len=32) :: cv_sst = 'xxx', cv_patm = 'xxx', cv_t_air = 'xxx', cv_q_air = 'xxx', cv_rh_air = 'xxx', &
      &                 cv_dp_air = 'xxx', cv_wndspd = 'xxx', cv_u_wnd = 'xxx', cv_v_wnd = 'xxx', cv_radsw = 'xxx', cv_radlw = 'xxx'

CONTAINS

   SUBROUTINE set_names( ct, cq, crh, cdp )
      CHARACTER(len=*), INTENT(in) :: ct, cq, crh, cdp
      cv_sst = 'sst'       ; cv_patm = 'msl'
      cv_t_air = ct        ; cv_q_air = cq     ; cv_rh_air = crh ; cv_dp_air = cdp
      cv_wndspd = 'wndspd' ; cv_u_wnd = 'u10'  ; cv_v_wnd = 'v10'
      cv_radsw = 'ssrd'    ; cv_radlw = 'strd'
   END SUBROUTINE set_names

   SUBROUTINE set_variable_names_default()
      CALL set_names( 't_air', 'q_air', 'rh_air', 'dp_air' )
   END SUBROUTINE set_variable_names_default

   SUBROUTINE set_variable_names_ecmwf()
      CALL set_names( 't2m', 'q2m', 'rh2m', 'd2m' )
   END SUBROUTINE set_variable_names_ecmwf

   SUBROUTINE ctl_stop( cd1, cd2, cd3, cd4, cd5, cd6, cd7, cd8, cd9, cd10 )
      !! the error model of AeroBulk: banner, the message lines that were passed, a blank line, STOP
      CHARACTER(len=*), INTENT(in), OPTIONAL :: cd1, cd2, cd3, cd4, cd5, cd6, cd7, cd8, cd9, cd10
      WRITE(6,cform_err)
      CALL line( cd1 ) ; CALL line( cd2 ) ; CALL line( cd3 ) ; CALL line( cd4 ) ; CALL line( cd5 )
      CALL line( cd6 ) ; CALL line( cd7 ) ; CALL line( cd8 ) ; CALL line( cd9 ) ; CALL line( cd10 )
      WRITE(6,*) ''
      STOP
   CONTAINS
      SUBROUTINE line( cd )
         CHARACTER(len=*), INTENT(in), OPTIONAL :: cd
         IF( PRESENT(cd) ) WRITE(6,*) TRIM(cd)
      END SUBROUTINE line
   END SUBROUTINE ctl_stop

END MODULE mod_const
