! oracle/ref_turb_driver.f90 -- TEST INFRASTRUCTURE (golden-vector generation only; runs in the build container).
!
! Calls the UNMODIFIED reference's TURB_* routines directly (the modules compiled by oracle/Makefile into oracle/_ref)
! with all their OPTIONAL diagnostic outputs, after the same pre-processing as aerobulk_compute
! (src/mod_aerobulk_compute.f90:99-126), and dumps everything as raw float64.  This is the call pattern of the
! reference's own toy driver (src/tests/aerobulk_toy.F90:324-393), written from scratch.
!
!   usage: ref_turb_driver.x <algo> <skin 0|1> <niter> <zt> <zu> <n> <in.bin> <out.bin>
!   in.bin : 8 planes of n float64: sst t_zt q_zt u v slp rad_sw rad_lw
!   out.bin: 16 planes of n float64: Cd Ch Ce t_zu q_zu Ubzu CdN ChN CeN z0 u_star L UN10 dT_cs dT_wl Hz_wl
PROGRAM ref_turb_driver
   USE mod_const
   USE mod_phymbl
   USE mod_blk_coare3p0
   USE mod_blk_coare3p6
   USE mod_blk_ncar
   USE mod_blk_ecmwf
   USE mod_blk_andreas
   IMPLICIT NONE
   CHARACTER(len=256) :: carg, calgo, cfin, cfout
   INTEGER :: iskin, n, k
   REAL(wp) :: zt, zu
   REAL(wp), DIMENSION(:,:), ALLOCATABLE :: sst, t_zt, q_zt, u, v, slp, rsw, rlw
   REAL(wp), DIMENSION(:,:), ALLOCATABLE :: W, Ts, qs, tht, Cd, Ch, Ce, t_zu, q_zu, Ub
   REAL(wp), DIMENSION(:,:), ALLOCATABLE :: CdN, ChN, CeN, z0, us, xL, UN10, dTcs, dTwl, Hzwl, lon
   LOGICAL :: lskin

   CALL GET_COMMAND_ARGUMENT(1, calgo)
   CALL GET_COMMAND_ARGUMENT(2, carg) ; READ(carg,*) iskin
   CALL GET_COMMAND_ARGUMENT(3, carg) ; READ(carg,*) nb_iter
   CALL GET_COMMAND_ARGUMENT(4, carg) ; READ(carg,*) zt
   CALL GET_COMMAND_ARGUMENT(5, carg) ; READ(carg,*) zu
   CALL GET_COMMAND_ARGUMENT(6, carg) ; READ(carg,*) n
   CALL GET_COMMAND_ARGUMENT(7, cfin)
   CALL GET_COMMAND_ARGUMENT(8, cfout)
   lskin = (iskin == 1)
   nitend = 1

   ALLOCATE( sst(n,1), t_zt(n,1), q_zt(n,1), u(n,1), v(n,1), slp(n,1), rsw(n,1), rlw(n,1) )
   ALLOCATE( W(n,1), Ts(n,1), qs(n,1), tht(n,1), Cd(n,1), Ch(n,1), Ce(n,1), t_zu(n,1), q_zu(n,1), Ub(n,1) )
   ALLOCATE( CdN(n,1), ChN(n,1), CeN(n,1), z0(n,1), us(n,1), xL(n,1), UN10(n,1), dTcs(n,1), dTwl(n,1), Hzwl(n,1), lon(n,1) )
   OPEN(11, FILE=TRIM(cfin), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='OLD')
   READ(11) sst, t_zt, q_zt, u, v, slp, rsw, rlw
   CLOSE(11)

   W   = SQRT( u*u + v*v )
   qs  = rdct_qsat_salt*q_sat( sst, slp )
   tht = Theta_from_z_P0_T_q( zt, slp, t_zt, q_zt )
   Ts  = sst
   lon = 0._wp
   dTcs = 0._wp ; dTwl = 0._wp ; Hzwl = 0._wp

   SELECT CASE( TRIM(calgo) )
   CASE('coare3p0')
      IF( lskin ) THEN
         CALL TURB_COARE3P0( 1, zt, zu, Ts, tht, qs, q_zt, W, .TRUE., .TRUE., Cd, Ch, Ce, t_zu, q_zu, Ub,           &
            &                pQsw=(1._wp - roce_alb0)*rsw, prad_lw=rlw, pslp=slp, pdT_cs=dTcs, isecday_utc=12, plong=lon, &
            &                pdT_wl=dTwl, pHz_wl=Hzwl, pCdN=CdN, pChN=ChN, pCeN=CeN, pz0=z0, pu_star=us, pL=xL, pUN10=UN10 )
      ELSE
         CALL TURB_COARE3P0( 1, zt, zu, Ts, tht, qs, q_zt, W, .FALSE., .FALSE., Cd, Ch, Ce, t_zu, q_zu, Ub, &
            &                pCdN=CdN, pChN=ChN, pCeN=CeN, pz0=z0, pu_star=us, pL=xL, pUN10=UN10 )
      END IF
   CASE('coare3p6')
      IF( lskin ) THEN
         CALL TURB_COARE3P6( 1, zt, zu, Ts, tht, qs, q_zt, W, .TRUE., .TRUE., Cd, Ch, Ce, t_zu, q_zu, Ub,           &
            &                Qsw=(1._wp - roce_alb0)*rsw, rad_lw=rlw, slp=slp, pdT_cs=dTcs, isecday_utc=12, plong=lon, &
            &                pdT_wl=dTwl, pHz_wl=Hzwl, CdN=CdN, ChN=ChN, CeN=CeN, xz0=z0, xu_star=us, xL=xL, xUN10=UN10 )
      ELSE
         CALL TURB_COARE3P6( 1, zt, zu, Ts, tht, qs, q_zt, W, .FALSE., .FALSE., Cd, Ch, Ce, t_zu, q_zu, Ub, &
            &                CdN=CdN, ChN=ChN, CeN=CeN, xz0=z0, xu_star=us, xL=xL, xUN10=UN10 )
      END IF
   CASE('ecmwf')
      IF( lskin ) THEN
         CALL TURB_ECMWF( 1, zt, zu, Ts, tht, qs, q_zt, W, .TRUE., .TRUE., Cd, Ch, Ce, t_zu, q_zu, Ub,   &
            &             pQsw=(1._wp - roce_alb0)*rsw, prad_lw=rlw, pslp=slp, pdT_cs=dTcs, pdT_wl=dTwl, pHz_wl=Hzwl, &
            &             pCdN=CdN, pChN=ChN, pCeN=CeN, pz0=z0, pu_star=us, pL=xL, pUN10=UN10 )
      ELSE
         CALL TURB_ECMWF( 1, zt, zu, Ts, tht, qs, q_zt, W, .FALSE., .FALSE., Cd, Ch, Ce, t_zu, q_zu, Ub, &
            &             pCdN=CdN, pChN=ChN, pCeN=CeN, pz0=z0, pu_star=us, pL=xL, pUN10=UN10 )
      END IF
   CASE('ncar')
      CALL TURB_NCAR( zt, zu, Ts, tht, qs, q_zt, W, Cd, Ch, Ce, t_zu, q_zu, Ub, &
         &            CdN=CdN, ChN=ChN, CeN=CeN, xz0=z0, xu_star=us, xL=xL, xUN10=UN10 )
   CASE('andreas')
      CALL TURB_ANDREAS( zt, zu, Ts, tht, qs, q_zt, W, Cd, Ch, Ce, t_zu, q_zu, Ub, &
         &               pCdN=CdN, pChN=ChN, pCeN=CeN, pz0=z0, pu_star=us, pL=xL, pUN10=UN10 )
   CASE DEFAULT
      STOP 'unknown algo'
   END SELECT

   OPEN(12, FILE=TRIM(cfout), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
   WRITE(12) Cd, Ch, Ce, t_zu, q_zu, Ub, CdN, ChN, CeN, z0, us, xL, UN10, dTcs, dTwl, Hzwl
   CLOSE(12)
END PROGRAM ref_turb_driver
