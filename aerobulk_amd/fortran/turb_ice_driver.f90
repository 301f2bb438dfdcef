! turb_ice_driver.f90 -- driver over the sea-ice bulk algorithms TURB_ICE_NEMO / EASY / AN05 / LU12 / LG15 / LG15_IO.
!
! The call pattern of the reference's ice drivers (src/ice/test_aerobulk_ice.f90:307-330,
! src/ice/test_aerobulk_buoy_series_ice.f90:394-412), with raw float64 files.  Like turb_series_driver.f90 the source only
! uses the public interface of the modules, so the SAME file builds against aerobulk_amd/fortran/mod_blk_ice.f90
! (-> libaerobulk_amd.so -> HIP kernels) and against the unmodified reference (oracle/_ref/ref_ice_driver.x, golden data).
!
!   usage: turb_ice_driver.x <nemo|easy|an05|lu12|lg15|lg15_io> <niter> <zt> <zu> <n> <in.bin> <out.bin>
!   in.bin : 6 planes of n float64: Ts_i theta_zt qs_i q_zt U_zu frice, then CdN ChN CeN (3 float64, read by `easy`)
!   out.bin: 14 planes: Cd Ch Ce t_zu q_zu Ub CdN ChN CeN z0 u_star L UN10 CdN_frm (the last one 0 except for lg15_io)
PROGRAM turb_ice_driver
   USE mod_const, ONLY: wp, nb_iter
   USE mod_blk_ice_nemo
   USE mod_blk_ice_easy
   USE mod_blk_ice_an05
   USE mod_blk_ice_lu12
   USE mod_blk_ice_lg15
   USE mod_blk_ice_lg15_io
   IMPLICIT NONE
   CHARACTER(len=512) :: carg, calgo, cfin, cfout
   INTEGER :: n
   REAL(wp) :: zt, zu, zcxn(3)
   REAL(wp), DIMENSION(:,:), ALLOCATABLE :: Ts, tht, qs, q_zt, W, fri
   REAL(wp), DIMENSION(:,:), ALLOCATABLE :: Cd, Ch, Ce, t_zu, q_zu, Ub, CdN, ChN, CeN, z0, us, xL, UN10, CdNf

   CALL GET_COMMAND_ARGUMENT(1, calgo)
   CALL GET_COMMAND_ARGUMENT(2, carg) ; READ(carg,*) nb_iter
   CALL GET_COMMAND_ARGUMENT(3, carg) ; READ(carg,*) zt
   CALL GET_COMMAND_ARGUMENT(4, carg) ; READ(carg,*) zu
   CALL GET_COMMAND_ARGUMENT(5, carg) ; READ(carg,*) n
   CALL GET_COMMAND_ARGUMENT(6, cfin)
   CALL GET_COMMAND_ARGUMENT(7, cfout)

   ALLOCATE( Ts(n,1), tht(n,1), qs(n,1), q_zt(n,1), W(n,1), fri(n,1) )
   ALLOCATE( Cd(n,1), Ch(n,1), Ce(n,1), t_zu(n,1), q_zu(n,1), Ub(n,1) )
   ALLOCATE( CdN(n,1), ChN(n,1), CeN(n,1), z0(n,1), us(n,1), xL(n,1), UN10(n,1), CdNf(n,1) )
   CdNf = 0._wp
   OPEN(11, FILE=TRIM(cfin), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='OLD')
   READ(11) Ts, tht, qs, q_zt, W, fri, zcxn
   CLOSE(11)

   SELECT CASE( TRIM(calgo) )
   CASE('nemo')
      CALL TURB_ICE_NEMO( zt, zu, Ts, tht, qs, q_zt, W, Cd, Ch, Ce, t_zu, q_zu, Ub, &
         &                CdN=CdN, ChN=ChN, CeN=CeN, xz0=z0, xu_star=us, xL=xL, xUN10=UN10 )
   CASE('easy')
      CdN = zcxn(1) ; ChN = zcxn(2) ; CeN = zcxn(3)
      CALL TURB_ICE_EASY( zt, zu, Ts, tht, qs, q_zt, W, zcxn(1), zcxn(2), zcxn(3), Cd, Ch, Ce, t_zu, q_zu, Ub, &
         &                xz0=z0, xu_star=us, xL=xL, xUN10=UN10 )
   CASE('an05')
      CALL TURB_ICE_AN05( zt, zu, Ts, tht, qs, q_zt, W, Cd, Ch, Ce, t_zu, q_zu, Ub, &
         &                CdN=CdN, ChN=ChN, CeN=CeN, xz0=z0, xu_star=us, xL=xL, xUN10=UN10 )
   CASE('lu12')
      CALL TURB_ICE_LU12( zt, zu, Ts, tht, qs, q_zt, W, fri, Cd, Ch, Ce, t_zu, q_zu, Ub, &
         &                CdN=CdN, ChN=ChN, CeN=CeN, xz0=z0, xu_star=us, xL=xL, xUN10=UN10 )
   CASE('lg15')
      CALL TURB_ICE_LG15( zt, zu, Ts, tht, qs, q_zt, W, fri, Cd, Ch, Ce, t_zu, q_zu, Ub, &
         &                CdN=CdN, ChN=ChN, CeN=CeN, xz0=z0, xu_star=us, xL=xL, xUN10=UN10 )
   CASE('lg15_io')   ! the call of src/ice/test_aerobulk_oce+ice.f90:345-347, plus the form-drag output
      CALL TURB_ICE_LG15_IO( zt, zu, Ts, tht, qs, q_zt, W, fri, Cd, Ch, Ce, t_zu, q_zu, Ub, CdN_frm=CdNf, &
         &                   CdN=CdN, ChN=ChN, CeN=CeN, xz0=z0, xu_star=us, xL=xL, xUN10=UN10 )
   CASE DEFAULT
      STOP 'unknown ice algorithm'
   END SELECT

   OPEN(12, FILE=TRIM(cfout), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
   WRITE(12) Cd, Ch, Ce, t_zu, q_zu, Ub, CdN, ChN, CeN, z0, us, xL, UN10, CdNf
   CLOSE(12)
END PROGRAM turb_ice_driver
