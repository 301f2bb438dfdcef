! neutral10_driver.f90 -- driver over TURB_NEUTRAL_10M (reference: src/mod_blk_neutral_10m.f90:33, called by
! src/tests/test_coef_n10.f90).  One source, two builds: against aerobulk_amd/fortran/mod_blk_turb.f90 (HIP engine) and against
! the unmodified reference (oracle/_ref/ref_neutral10_driver.x, golden data).
!   usage: neutral10_driver.x <algo> <niter> <n> <in.bin> <out.bin>     in: U_N10(n) ; out: CdN10 ChN10 CeN10 z0 (4 planes)
PROGRAM neutral10_driver
   USE mod_const, ONLY: wp, nb_iter
   USE mod_blk_neutral_10m
   IMPLICIT NONE
   CHARACTER(len=512) :: carg, calgo, cfin, cfout
   INTEGER :: n
   REAL(wp), DIMENSION(:,:), ALLOCATABLE :: U, Cd, Ch, Ce, z0
   CALL GET_COMMAND_ARGUMENT(1, calgo)
   CALL GET_COMMAND_ARGUMENT(2, carg) ; READ(carg,*) nb_iter
   CALL GET_COMMAND_ARGUMENT(3, carg) ; READ(carg,*) n
   CALL GET_COMMAND_ARGUMENT(4, cfin)
   CALL GET_COMMAND_ARGUMENT(5, cfout)
   ALLOCATE( U(n,1), Cd(n,1), Ch(n,1), Ce(n,1), z0(n,1) )
   OPEN(11, FILE=TRIM(cfin), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='OLD')
   READ(11) U
   CLOSE(11)
   CALL TURB_NEUTRAL_10M( TRIM(calgo), U, Cd, Ch, Ce, z0 )
   OPEN(12, FILE=TRIM(cfout), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='REPLACE')
   WRITE(12) Cd, Ch, Ce, z0
   CLOSE(12)
END PROGRAM neutral10_driver
