! oce_ice_driver.f90 -- a cell that is part open water (leads), part sea ice: the composition of src/ice/test_aerobulk_oce+ice.f90 on arrays.
!
! The reference has no `aerobulk_compute`-level entry for such cells; its test program composes the pieces by hand (:280-420): saturation
! humidities over water and over ice, TURB_ECMWF over the leads, TURB_ICE_NEMO / AN05 / LG15_IO over the ice, Ri_bulk, the moist lapse rate,
! air density at zu, BULK_FORMULA twice (l_ice for the ice part).  That program cannot run under amdflang (it re-opens unit 6 with RECL=), so
! this driver — own source, only the modules' public interfaces — repeats its sequence of calls on n cells and is built twice like the other
! drivers: against this repository's modules (-> libaerobulk_amd.so -> HIP kernels) and against the unmodified reference
! (oracle/_ref/ref_oce_ice_driver.x: the golden data of tests/test_oce_ice.py, tools/gen_oce_ice_golden.py).
!
!   usage: oce_ice_driver.x <n> <in.bin> <out.bin>        zt = 2 m, zu = 10 m, nb_iter = 20 (the program's :67)
!   in.bin : 7 planes of n float64: sst sit t_zt q_zt W10 frci SLP
!   out.bin: records { character(24) name ; int32 m ; m doubles }
PROGRAM oce_ice_driver
   USE mod_const
   USE mod_phymbl
   USE mod_blk_ecmwf,       ONLY: TURB_ECMWF
   USE mod_blk_ice_nemo
   USE mod_blk_ice_an05
   USE mod_blk_ice_lg15_io
   IMPLICIT NONE
   REAL(wp), PARAMETER :: zt = 2._wp, zu = 10._wp
   INTEGER :: n, ialgo, jq
   CHARACTER(len=512) :: carg, cfin, cfout
   CHARACTER(len=8), DIMENSION(3), PARAMETER :: vca = (/ 'nemo    ', 'an05    ', 'lg15_io ' /)
   REAL(wp), DIMENSION(:,:), ALLOCATABLE :: sst, sit, t_zt, q_zt, W10, frci, SLP, ssq, siq, theta_zt, rgamma, tmp
   REAL(wp), DIMENSION(:,:), ALLOCATABLE :: Cd_w, Ch_w, Ce_w, theta_zu_w, q_zu_w, Ublk_w, Tau_w, QH_w, QL_w, Evap_w, rhoa_w
   REAL(wp), DIMENSION(:,:), ALLOCATABLE :: Cd, Ch, Ce, theta_zu, q_zu, Ublk, zz0, zus, zL, zUN10, t_zu, rho_zu, Tau, QH, QL, Evap, rhoa

   CALL GET_COMMAND_ARGUMENT(1, carg) ; READ(carg,*) n
   CALL GET_COMMAND_ARGUMENT(2, cfin)
   CALL GET_COMMAND_ARGUMENT(3, cfout)
   nb_iter = 20
   ALLOCATE( sst(n,1), sit(n,1), t_zt(n,1), q_zt(n,1), W10(n,1), frci(n,1), SLP(n,1), ssq(n,1), siq(n,1), theta_zt(n,1), rgamma(n,1), tmp(n,1) )
   ALLOCATE( Cd_w(n,1), Ch_w(n,1), Ce_w(n,1), theta_zu_w(n,1), q_zu_w(n,1), Ublk_w(n,1), Tau_w(n,1), QH_w(n,1), QL_w(n,1), Evap_w(n,1), rhoa_w(n,1) )
   ALLOCATE( Cd(n,1), Ch(n,1), Ce(n,1), theta_zu(n,1), q_zu(n,1), Ublk(n,1), zz0(n,1), zus(n,1), zL(n,1), zUN10(n,1), t_zu(n,1), rho_zu(n,1) )
   ALLOCATE( Tau(n,1), QH(n,1), QL(n,1), Evap(n,1), rhoa(n,1) )
   OPEN(11, FILE=TRIM(cfin), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='OLD')
   READ(11) sst, sit, t_zt, q_zt, W10, frci, SLP
   CLOSE(11)
   OPEN(12, FILE=TRIM(cfout), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')

   !! saturation at the two surfaces (:215-216), potential temperature at zt from the moist lapse rate (:250,265)
   ssq = rdct_qsat_salt*q_sat( sst, SLP )                  ; CALL put('ssq', ssq)
   siq =                q_sat( sit, SLP, l_ice=.TRUE. )     ; CALL put('siq', siq)
   rgamma   = gamma_moist( t_zt, q_zt )
   theta_zt = t_zt + rgamma*zt                              ; CALL put('theta_zt', theta_zt)
   tmp = Ri_bulk( zt, sit, theta_zt, siq, q_zt, W10 )        ; CALL put('rib_ice_zt', tmp)
   tmp = Ri_bulk( zt, sst, theta_zt, ssq, q_zt, W10 )        ; CALL put('rib_water_zt', tmp)

   !! ---- over water (the leads): :291-312
   CALL TURB_ECMWF( 1, zt, zu, sst, theta_zt, ssq, q_zt, W10, .FALSE., .FALSE., &
      &             Cd_w, Ch_w, Ce_w, theta_zu_w, q_zu_w, Ublk_w )
   CALL put('w_cd', Cd_w) ; CALL put('w_ch', Ch_w) ; CALL put('w_ce', Ce_w) ; CALL put('w_theta_zu', theta_zu_w)
   CALL put('w_q_zu', q_zu_w) ; CALL put('w_ublk', Ublk_w)
   CALL BULK_FORMULA( zu, sst, ssq, theta_zu_w, q_zu_w, Cd_w, Ch_w, Ce_w, W10, Ublk_w, SLP, Tau_w, QH_w, QL_w, pEvap=Evap_w, prhoa=rhoa_w )
   CALL put('w_tau', Tau_w) ; CALL put('w_qh', QH_w) ; CALL put('w_ql', QL_w) ; CALL put('w_evap', Evap_w)

   !! ---- over the ice, three algorithms: :322-405
   DO ialgo = 1, 3
      zz0 = 0._wp ; zus = 0._wp ; zL = 0._wp ; zUN10 = 0._wp
      SELECT CASE(ialgo)
      CASE(1)
         CALL TURB_ICE_NEMO( zt, zu, sit, theta_zt, siq, q_zt, W10,   &
            &                Cd, Ch, Ce, theta_zu, q_zu, Ublk,         &
            &                xz0=zz0, xu_star=zus, xL=zL, xUN10=zUN10 )
      CASE(2)
         CALL TURB_ICE_AN05( zt, zu, sit, theta_zt, siq, q_zt, W10,   &
            &                Cd, Ch, Ce, theta_zu, q_zu, Ublk,         &
            &                xz0=zz0, xu_star=zus, xL=zL, xUN10=zUN10 )
      CASE(3)
         CALL TURB_ICE_LG15_IO( zt, zu, sit, theta_zt, siq, q_zt, W10, frci,  &
            &                   Cd, Ch, Ce, theta_zu, q_zu, Ublk,             &
            &                   xz0=zz0, xu_star=zus, xL=zL, xUN10=zUN10 )
      END SELECT
      CALL put(TRIM(vca(ialgo))//'_cd', Cd) ; CALL put(TRIM(vca(ialgo))//'_ch', Ch) ; CALL put(TRIM(vca(ialgo))//'_ce', Ce)
      CALL put(TRIM(vca(ialgo))//'_theta_zu', theta_zu) ; CALL put(TRIM(vca(ialgo))//'_q_zu', q_zu) ; CALL put(TRIM(vca(ialgo))//'_ublk', Ublk)
      CALL put(TRIM(vca(ialgo))//'_z0', zz0) ; CALL put(TRIM(vca(ialgo))//'_us', zus) ; CALL put(TRIM(vca(ialgo))//'_un10', zUN10)
      tmp = Ri_bulk( zu, sit, theta_zu, siq, q_zu, Ublk )    ; CALL put(TRIM(vca(ialgo))//'_rib', tmp)
      !! absolute temperature at zu (:363-368), air density there (:388-391)
      t_zu = theta_zu
      DO jq = 1, 4
         rgamma = gamma_moist( 0.5*(t_zu+sit), q_zu )
         t_zu = theta_zu - rgamma*zu
      END DO
      CALL put(TRIM(vca(ialgo))//'_t_zu', t_zu)
      rho_zu = rho_air( t_zu, q_zu, SLP )
      tmp = SLP - rho_zu*grav*zu
      rho_zu = rho_air( t_zu, q_zu, tmp )                    ; CALL put(TRIM(vca(ialgo))//'_rho_zu', rho_zu)
      CALL BULK_FORMULA( zu, sit, siq, theta_zu, q_zu, Cd, Ch, Ce, W10, Ublk, SLP, Tau, QH, QL, pEvap=Evap, prhoa=rhoa, l_ice=.TRUE. )
      CALL put(TRIM(vca(ialgo))//'_tau', Tau) ; CALL put(TRIM(vca(ialgo))//'_qh', QH) ; CALL put(TRIM(vca(ialgo))//'_ql', QL)
      CALL put(TRIM(vca(ialgo))//'_evap', Evap)
      !! the cell as a whole: ice and leads weighted by the ice fraction
      tmp = frci*QH + (1._wp - frci)*QH_w                    ; CALL put(TRIM(vca(ialgo))//'_qh_cell', tmp)
      tmp = frci*QL + (1._wp - frci)*QL_w                    ; CALL put(TRIM(vca(ialgo))//'_ql_cell', tmp)
      tmp = frci*Tau + (1._wp - frci)*Tau_w                  ; CALL put(TRIM(vca(ialgo))//'_tau_cell', tmp)
   END DO
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

END PROGRAM oce_ice_driver
