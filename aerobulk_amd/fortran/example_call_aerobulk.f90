! Example / parity driver for the Fortran host: the call pattern of the reference's own API example
! (reference: src/tests/example_call_aerobulk.f90) -- all 5 algorithms on a 2x1 domain (unstable / stable
! cell), skin schemes on for COARE*/ECMWF -- written against this repository's mod_aerobulk.
! Prints one machine-readable line per algorithm:   RESULT <algo> QH1 QH2 QL1 QL2 E1 E2 Ts1 Ts2 Tx1 Tx2 Ty1 Ty2
PROGRAM example_call_aerobulk
   USE mod_aerobulk
   USE mod_const, ONLY: wp, rt0
   IMPLICIT NONE
   INTEGER, PARAMETER :: nx = 2, ny = 1
   INTEGER :: Nbit = 50, ia
   REAL(wp), PARAMETER :: zt = 2., zu = 10.
   REAL(wp), DIMENSION(nx,ny) :: zsst, zt_zt, zq_zt, zU_zu, zV_zu, zSLP, zRsw, zRlw, zQL, zQH, zTau_x, zTau_y, zE, zTs
   CHARACTER(len=8), DIMENSION(5), PARAMETER :: calgos = (/ 'coare3p0', 'coare3p6', 'ecmwf   ', 'ncar    ', 'andreas ' /)
   CHARACTER(len=16) :: carg

   IF( COMMAND_ARGUMENT_COUNT() >= 1 ) THEN
      CALL GET_COMMAND_ARGUMENT(1, carg)
      READ(carg,*) Nbit
   END IF

   zsst  = 22. + rt0
   zt_zt(1,1) = rt0 + 20.
   zt_zt(2,1) = rt0 + 25.
   zq_zt = 0.012
   zU_zu = 5.
   zV_zu = 0.
   zSLP  = 101000.
   zRsw  = 0.
   zRlw  = 350.

   DO ia = 1, 5
      zTs = zsst
      IF( ia <= 3 ) THEN
         CALL aerobulk_model( 1, 1, TRIM(calgos(ia)), zt, zu, zsst, zt_zt, zq_zt, zU_zu, zV_zu, zSLP, &
            &                 zQL, zQH, zTau_x, zTau_y, zE,                                           &
            &                 Niter=Nbit, l_use_skin=.TRUE., rad_sw=zRsw, rad_lw=zRlw, T_s=zTs )
      ELSE
         CALL aerobulk_model( 1, 1, TRIM(calgos(ia)), zt, zu, zsst, zt_zt, zq_zt, zU_zu, zV_zu, zSLP, &
            &                 zQL, zQH, zTau_x, zTau_y, zE, Niter=Nbit )
      END IF
      WRITE(6,'("RESULT ",a8,12(1x,es24.16))') calgos(ia), zQH(1,1), zQH(2,1), zQL(1,1), zQL(2,1), zE(1,1), zE(2,1), &
         &                                     zTs(1,1), zTs(2,1), zTau_x(1,1), zTau_x(2,1), zTau_y(1,1), zTau_y(2,1)
   END DO
END PROGRAM example_call_aerobulk
