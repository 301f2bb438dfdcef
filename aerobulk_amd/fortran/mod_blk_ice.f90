! mod_blk_ice.f90 -- Fortran hosts of the sea-ice bulk algorithms of the MI355X-native engine.
!
! Drop-in for callers of the reference's src/ice modules:
!    mod_blk_ice_nemo : TURB_ICE_NEMO   (reference: src/ice/mod_blk_ice_nemo.f90:36-38)
!    mod_blk_ice_an05 : TURB_ICE_AN05   (src/ice/mod_blk_ice_an05.f90:41-43)
!    mod_blk_ice_lu12 : TURB_ICE_LU12   (src/ice/mod_blk_ice_lu12.f90:69-71)
!    mod_blk_ice_lg15 : TURB_ICE_LG15   (src/ice/mod_blk_ice_lg15.f90:68-70)
!    mod_blk_ice_lg15_io : TURB_ICE_LG15_IO (src/ice/mod_blk_ice_lg15_io.f90:69-72), over-ice outputs + CdN_frm
!    mod_blk_ice_easy : TURB_ICE_EASY   (src/ice/mod_blk_ice_easy.f90:44-47)
! Same module / routine / dummy-argument names, INTENTs and OPTIONALs; everything goes through ISO_C_BINDING to
! `ab_turb_ice` (include/aerobulk_amd.h), i.e. to ice_kernel (aerobulk_amd/csrc/ab_ice_kernels.hip).  `nb_iter` is read from
! mod_const like the reference does.  Compile after mod_aerobulk.f90 (mod_const), with -fdefault-real-8.

MODULE mod_ab_ice
   USE, INTRINSIC :: ISO_C_BINDING
   USE mod_const, ONLY: wp, nb_iter
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: ab_ice_generic, ab_ice_easy, ab_ice_helper

   !! mirror of `ab_ice_fields` (include/aerobulk_amd.h)
   TYPE, BIND(C) :: ab_ice_fields
      TYPE(C_PTR) :: Ts_i, theta_zt, qs_i, q_zt, U_zu, frice
      TYPE(C_PTR) :: Cd, Ch, Ce, t_zu, q_zu, Ub
      TYPE(C_PTR) :: CdN, ChN, CeN, z0, u_star, L, UN10
      TYPE(C_PTR) :: CdN_frm
   END TYPE ab_ice_fields

   INTERFACE
      FUNCTION ab_turb_ice( ice_algo, zt, zu, niter, f, n, iprecision, mem, stream ) BIND(C, NAME='ab_turb_ice') RESULT(istat)
         IMPORT :: C_INT, C_LONG, C_DOUBLE, C_PTR, ab_ice_fields
         INTEGER(C_INT),  VALUE :: ice_algo, niter, iprecision, mem
         REAL(C_DOUBLE),  VALUE :: zt, zu
         TYPE(ab_ice_fields), INTENT(in) :: f
         INTEGER(C_LONG), VALUE :: n
         TYPE(C_PTR),     VALUE :: stream
         INTEGER(C_INT) :: istat
      END FUNCTION ab_turb_ice
      FUNCTION ab_turb_ice_easy( zt, zu, niter, CdN, ChN, CeN, f, n, iprecision, mem, stream ) BIND(C, NAME='ab_turb_ice_easy') RESULT(istat)
         IMPORT :: C_INT, C_LONG, C_DOUBLE, C_PTR, ab_ice_fields
         INTEGER(C_INT),  VALUE :: niter, iprecision, mem
         REAL(C_DOUBLE),  VALUE :: zt, zu, CdN, ChN, CeN
         TYPE(ab_ice_fields), INTENT(in) :: f
         INTEGER(C_LONG), VALUE :: n
         TYPE(C_PTR),     VALUE :: stream
         INTEGER(C_INT) :: istat
      END FUNCTION ab_turb_ice_easy
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
      FUNCTION ab_last_error() BIND(C, NAME='ab_last_error') RESULT(cptr)
         IMPORT :: C_PTR
         TYPE(C_PTR) :: cptr
      END FUNCTION ab_last_error
      FUNCTION c_strlen(s) BIND(C, NAME='strlen') RESULT(n)
         IMPORT :: C_PTR, C_SIZE_T
         TYPE(C_PTR), VALUE :: s
         INTEGER(C_SIZE_T) :: n
      END FUNCTION c_strlen
   END INTERFACE

CONTAINS

   SUBROUTINE ab_ice_helper( fn, n, a1, a2, o1, a3, o2 )
      !! the PUBLIC helper functions of mod_blk_ice_an05 (rough_leng_m, rough_leng_tq) on the engine: `ab_phymbl` functions 40 / 41
      INTEGER, INTENT(in) :: fn, n
      REAL(wp), DIMENSION(n), INTENT(in),  TARGET           :: a1, a2
      REAL(wp), DIMENSION(n), INTENT(out), TARGET           :: o1
      REAL(wp), DIMENSION(n), INTENT(in),  TARGET, OPTIONAL :: a3
      REAL(wp), DIMENSION(n), INTENT(out), TARGET, OPTIONAL :: o2
      TYPE(C_PTR), DIMENSION(3) :: pin
      TYPE(C_PTR), DIMENSION(2) :: pout
      REAL(C_DOUBLE), DIMENSION(2) :: par, zinfo
      INTEGER(C_INT) :: istat
      pin = (/ C_LOC(a1), C_LOC(a2), C_NULL_PTR /) ; pout = (/ C_LOC(o1), C_NULL_PTR /) ; par = 0._C_DOUBLE
      IF( PRESENT(a3) ) pin(3)  = C_LOC(a3)
      IF( PRESENT(o2) ) pout(2) = C_LOC(o2)
      istat = ab_phymbl( INT(fn,C_INT), INT(n,C_LONG), pin, 3_C_INT, pout, 2_C_INT, par, 0_C_INT, 0_C_INT, C_NULL_PTR, zinfo )
      IF( istat /= 0 ) THEN
         WRITE(6,*) ' *** E R R O R : mod_blk_ice_an05 (aerobulk_amd): ab_phymbl status', INT(istat)
         STOP
      END IF
   END SUBROUTINE ab_ice_helper

   SUBROUTINE stop_with_library_message()
      TYPE(C_PTR) :: cp
      CHARACTER(KIND=C_CHAR), DIMENSION(:), POINTER :: cs
      INTEGER :: n, i
      CHARACTER(len=1024) :: cmsg
      cp = ab_last_error()
      cmsg = ''
      IF( C_ASSOCIATED(cp) ) THEN
         n = MIN( INT(c_strlen(cp)), 1024 )
         CALL C_F_POINTER( cp, cs, (/ n /) )
         DO i = 1, n
            cmsg(i:i) = cs(i)
         END DO
      END IF
      WRITE(6,'(" *** E R R O R :  ")')
      WRITE(6,*) TRIM(cmsg)
      WRITE(6,*) ''
      STOP
   END SUBROUTINE stop_with_library_message

   SUBROUTINE ab_ice_generic( ialgo, zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, Cd, Ch, Ce, t_zu, q_zu, Ub, &
      &                       frice, CdN, ChN, CeN, xz0, xu_star, xL, xUN10, CdN_frm )
      INTEGER,                  INTENT(in)  :: ialgo
      REAL(wp),                 INTENT(in)  :: zt, zu
      REAL(wp), DIMENSION(:,:), INTENT(in)  :: Ts_i, t_zt, qs_i, q_zt, U_zu
      REAL(wp), DIMENSION(:,:), INTENT(out) :: Cd, Ch, Ce, t_zu, q_zu, Ub
      REAL(wp), DIMENSION(:,:), INTENT(in),  OPTIONAL :: frice
      REAL(wp), DIMENSION(:,:), INTENT(out), OPTIONAL :: CdN, ChN, CeN, xz0, xu_star, xL, xUN10, CdN_frm
      !! explicit-shape dummies: the compiler hands over contiguous storage
      CALL ice_contig( SIZE(Ts_i), Ts_i, t_zt, qs_i, q_zt, U_zu, Cd, Ch, Ce, t_zu, q_zu, Ub, frice, &
         &             CdN, ChN, CeN, xz0, xu_star, xL, xUN10, CdN_frm )
   CONTAINS
      SUBROUTINE ice_contig( n, a1, a2, a3, a4, a5, o1, o2, o3, o4, o5, o6, r1, d1, d2, d3, d4, d5, d6, d7, d8 )
         INTEGER, INTENT(in) :: n
         REAL(wp), DIMENSION(n), INTENT(in),  TARGET :: a1, a2, a3, a4, a5
         REAL(wp), DIMENSION(n), INTENT(out), TARGET :: o1, o2, o3, o4, o5, o6
         REAL(wp), DIMENSION(n), INTENT(in),  TARGET, OPTIONAL :: r1
         REAL(wp), DIMENSION(n), INTENT(out), TARGET, OPTIONAL :: d1, d2, d3, d4, d5, d6, d7, d8
         TYPE(ab_ice_fields) :: f
         INTEGER(C_INT) :: istat
         f%Ts_i = C_LOC(a1) ; f%theta_zt = C_LOC(a2) ; f%qs_i = C_LOC(a3) ; f%q_zt = C_LOC(a4) ; f%U_zu = C_LOC(a5)
         f%Cd = C_LOC(o1) ; f%Ch = C_LOC(o2) ; f%Ce = C_LOC(o3) ; f%t_zu = C_LOC(o4) ; f%q_zu = C_LOC(o5) ; f%Ub = C_LOC(o6)
         f%frice = C_NULL_PTR ; f%CdN_frm = C_NULL_PTR
         IF( PRESENT(d8) ) f%CdN_frm = C_LOC(d8)
         f%CdN = C_NULL_PTR ; f%ChN = C_NULL_PTR ; f%CeN = C_NULL_PTR ; f%z0 = C_NULL_PTR
         f%u_star = C_NULL_PTR ; f%L = C_NULL_PTR ; f%UN10 = C_NULL_PTR
         IF( PRESENT(r1) ) f%frice  = C_LOC(r1)
         IF( PRESENT(d1) ) f%CdN    = C_LOC(d1)
         IF( PRESENT(d2) ) f%ChN    = C_LOC(d2)
         IF( PRESENT(d3) ) f%CeN    = C_LOC(d3)
         IF( PRESENT(d4) ) f%z0     = C_LOC(d4)
         IF( PRESENT(d5) ) f%u_star = C_LOC(d5)
         IF( PRESENT(d6) ) f%L      = C_LOC(d6)
         IF( PRESENT(d7) ) f%UN10   = C_LOC(d7)
         istat = ab_turb_ice( INT(ialgo,C_INT), REAL(zt,C_DOUBLE), REAL(zu,C_DOUBLE), INT(nb_iter,C_INT), f, INT(n,C_LONG), &
            &                 0_C_INT, 0_C_INT, C_NULL_PTR )      ! AB_F64, AB_MEM_HOST
         IF( istat /= 0 ) CALL stop_with_library_message()
      END SUBROUTINE ice_contig
   END SUBROUTINE ab_ice_generic

   SUBROUTINE ab_ice_easy( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, CdN, ChN, CeN, Cd, Ch, Ce, t_zu, q_zu, Ub, xz0, xu_star, xL, xUN10 )
      REAL(wp),                 INTENT(in)  :: zt, zu, CdN, ChN, CeN
      REAL(wp), DIMENSION(:,:), INTENT(in)  :: Ts_i, t_zt, qs_i, q_zt, U_zu
      REAL(wp), DIMENSION(:,:), INTENT(out) :: Cd, Ch, Ce, t_zu, q_zu, Ub
      REAL(wp), DIMENSION(:,:), INTENT(out), OPTIONAL :: xz0, xu_star, xL, xUN10
      CALL easy_contig( SIZE(Ts_i), Ts_i, t_zt, qs_i, q_zt, U_zu, Cd, Ch, Ce, t_zu, q_zu, Ub, xz0, xu_star, xL, xUN10 )
   CONTAINS
      SUBROUTINE easy_contig( n, a1, a2, a3, a4, a5, o1, o2, o3, o4, o5, o6, d4, d5, d6, d7 )
         INTEGER, INTENT(in) :: n
         REAL(wp), DIMENSION(n), INTENT(in),  TARGET :: a1, a2, a3, a4, a5
         REAL(wp), DIMENSION(n), INTENT(out), TARGET :: o1, o2, o3, o4, o5, o6
         REAL(wp), DIMENSION(n), INTENT(out), TARGET, OPTIONAL :: d4, d5, d6, d7
         TYPE(ab_ice_fields) :: f
         INTEGER(C_INT) :: istat
         f%Ts_i = C_LOC(a1) ; f%theta_zt = C_LOC(a2) ; f%qs_i = C_LOC(a3) ; f%q_zt = C_LOC(a4) ; f%U_zu = C_LOC(a5)
         f%Cd = C_LOC(o1) ; f%Ch = C_LOC(o2) ; f%Ce = C_LOC(o3) ; f%t_zu = C_LOC(o4) ; f%q_zu = C_LOC(o5) ; f%Ub = C_LOC(o6)
         f%frice = C_NULL_PTR ; f%CdN = C_NULL_PTR ; f%ChN = C_NULL_PTR ; f%CeN = C_NULL_PTR ; f%CdN_frm = C_NULL_PTR
         f%z0 = C_NULL_PTR ; f%u_star = C_NULL_PTR ; f%L = C_NULL_PTR ; f%UN10 = C_NULL_PTR
         IF( PRESENT(d4) ) f%z0     = C_LOC(d4)
         IF( PRESENT(d5) ) f%u_star = C_LOC(d5)
         IF( PRESENT(d6) ) f%L      = C_LOC(d6)
         IF( PRESENT(d7) ) f%UN10   = C_LOC(d7)
         istat = ab_turb_ice_easy( REAL(zt,C_DOUBLE), REAL(zu,C_DOUBLE), INT(nb_iter,C_INT), REAL(CdN,C_DOUBLE), REAL(ChN,C_DOUBLE), &
            &                      REAL(CeN,C_DOUBLE), f, INT(n,C_LONG), 0_C_INT, 0_C_INT, C_NULL_PTR )
         IF( istat /= 0 ) CALL stop_with_library_message()
      END SUBROUTINE easy_contig
   END SUBROUTINE ab_ice_easy

END MODULE mod_ab_ice


MODULE mod_blk_ice_nemo
   USE mod_const, ONLY: wp
   USE mod_ab_ice
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_ICE_NEMO
CONTAINS
   SUBROUTINE TURB_ICE_NEMO( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu,         &
      &                      Cd, Ch, Ce, t_zu, q_zu, Ub,                       &
      &                      CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
      REAL(wp), INTENT(in )                 :: zt, zu
      REAL(wp), INTENT(in ), DIMENSION(:,:) :: Ts_i, t_zt, qs_i, q_zt, U_zu
      REAL(wp), INTENT(out), DIMENSION(:,:) :: Cd, Ch, Ce, t_zu, q_zu, Ub
      REAL(wp), INTENT(out), OPTIONAL, DIMENSION(:,:) :: CdN, ChN, CeN, xz0, xu_star, xL, xUN10
      CALL ab_ice_generic( 1, zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, Cd, Ch, Ce, t_zu, q_zu, Ub, &
         &                 CdN=CdN, ChN=ChN, CeN=CeN, xz0=xz0, xu_star=xu_star, xL=xL, xUN10=xUN10 )
   END SUBROUTINE TURB_ICE_NEMO
END MODULE mod_blk_ice_nemo


MODULE mod_blk_ice_an05
   USE mod_const, ONLY: wp
   USE mod_ab_ice
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_ICE_AN05
   PUBLIC :: rough_leng_m, rough_leng_tq
CONTAINS
   FUNCTION rough_leng_m( pus , pnua )                                        !! reference :232-255 (Andreas et al. 2005, eq. 19)
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pus, pnua
      REAL(wp), DIMENSION(SIZE(pus,1),SIZE(pus,2)) :: rough_leng_m
      CALL ab_ice_helper( 40, SIZE(pus), pus, pnua, rough_leng_m )
   END FUNCTION rough_leng_m

   FUNCTION rough_leng_tq( pz0, pus , pnua )                                  !! :257-312 (eq. 22): (:,:,1) temperature, (:,:,2) humidity
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pz0, pus, pnua
      REAL(wp), DIMENSION(SIZE(pus,1),SIZE(pus,2),2) :: rough_leng_tq
      REAL(wp), DIMENSION(SIZE(pus,1),SIZE(pus,2)) :: zt, zq
      CALL ab_ice_helper( 41, SIZE(pus), pz0, pus, zt, a3=pnua, o2=zq )
      rough_leng_tq(:,:,1) = zt ; rough_leng_tq(:,:,2) = zq
   END FUNCTION rough_leng_tq

   SUBROUTINE TURB_ICE_AN05( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu,         &
      &                      Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu,                   &
      &                      CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
      REAL(wp), INTENT(in )                 :: zt, zu
      REAL(wp), INTENT(in ), DIMENSION(:,:) :: Ts_i, t_zt, qs_i, q_zt, U_zu
      REAL(wp), INTENT(out), DIMENSION(:,:) :: Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu
      REAL(wp), INTENT(out), OPTIONAL, DIMENSION(:,:) :: CdN, ChN, CeN, xz0, xu_star, xL, xUN10
      CALL ab_ice_generic( 2, zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu, &
         &                 CdN=CdN, ChN=ChN, CeN=CeN, xz0=xz0, xu_star=xu_star, xL=xL, xUN10=xUN10 )
   END SUBROUTINE TURB_ICE_AN05
END MODULE mod_blk_ice_an05


MODULE mod_blk_ice_lu12
   USE mod_const, ONLY: wp
   USE mod_ab_ice
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_ICE_LU12
CONTAINS
   SUBROUTINE TURB_ICE_LU12( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, frice, &
      &                      Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu,      &
      &                      CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
      REAL(wp), INTENT(in )                 :: zt, zu
      REAL(wp), INTENT(in ), DIMENSION(:,:) :: Ts_i, t_zt, qs_i, q_zt, U_zu, frice
      REAL(wp), INTENT(out), DIMENSION(:,:) :: Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu
      REAL(wp), INTENT(out), OPTIONAL, DIMENSION(:,:) :: CdN, ChN, CeN, xz0, xu_star, xL, xUN10
      CALL ab_ice_generic( 3, zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu, frice=frice, &
         &                 CdN=CdN, ChN=ChN, CeN=CeN, xz0=xz0, xu_star=xu_star, xL=xL, xUN10=xUN10 )
   END SUBROUTINE TURB_ICE_LU12
END MODULE mod_blk_ice_lu12


MODULE mod_blk_ice_lg15
   USE mod_const, ONLY: wp
   USE mod_ab_ice
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_ICE_LG15
CONTAINS
   SUBROUTINE TURB_ICE_LG15( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, frice, &
      &                      Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu,      &
      &                      CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
      REAL(wp), INTENT(in )                 :: zt, zu
      REAL(wp), INTENT(in ), DIMENSION(:,:) :: Ts_i, t_zt, qs_i, q_zt, U_zu, frice
      REAL(wp), INTENT(out), DIMENSION(:,:) :: Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu
      REAL(wp), INTENT(out), OPTIONAL, DIMENSION(:,:) :: CdN, ChN, CeN, xz0, xu_star, xL, xUN10
      CALL ab_ice_generic( 4, zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu, frice=frice, &
         &                 CdN=CdN, ChN=ChN, CeN=CeN, xz0=xz0, xu_star=xu_star, xL=xL, xUN10=xUN10 )
   END SUBROUTINE TURB_ICE_LG15
END MODULE mod_blk_ice_lg15


MODULE mod_blk_ice_easy
   USE mod_const, ONLY: wp
   USE mod_ab_ice
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_ICE_EASY
CONTAINS
   SUBROUTINE TURB_ICE_EASY( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu,     &
      &                      CdN, ChN, CeN,                            &
      &                      Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu,   &
      &                      xz0, xu_star, xL, xUN10 )
      REAL(wp), INTENT(in )                 :: zt, zu
      REAL(wp), INTENT(in ), DIMENSION(:,:) :: Ts_i, t_zt, qs_i, q_zt, U_zu
      REAL(wp), INTENT(in )                 :: CdN, ChN, CeN
      REAL(wp), INTENT(out), DIMENSION(:,:) :: Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu
      REAL(wp), INTENT(out), OPTIONAL, DIMENSION(:,:) :: xz0, xu_star, xL, xUN10
      CALL ab_ice_easy( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, CdN, ChN, CeN, Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu, &
         &              xz0=xz0, xu_star=xu_star, xL=xL, xUN10=xUN10 )
   END SUBROUTINE TURB_ICE_EASY
END MODULE mod_blk_ice_easy


MODULE mod_blk_ice_lg15_io
   !! TURB_ICE_LG15_IO (src/ice/mod_blk_ice_lg15_io.f90:69-404).  Over ice it is TURB_ICE_LG15 plus the form-drag output
   !! CdN_frm.  Its over-water outputs (Cd_w, Ch_w, Ce_w, t_zu_w, q_zu_w from Ts_w, qs_w) read work arrays the reference never
   !! assigns (zz0_s, zCdN_s, zChN_s (:,:,2): ALLOCATEd at :174-175, used at :292-293): there is no defined result to
   !! reproduce, so asking for them stops with a message; the reference's own caller does not ask either
   !! (src/ice/test_aerobulk_oce+ice.f90:345-347).
   USE mod_const, ONLY: wp
   USE mod_ab_ice
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_ICE_LG15_IO
CONTAINS
   SUBROUTINE TURB_ICE_LG15_IO( zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, frice, &
      &                         Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu,            &
      &                         Ts_w, qs_w, CdN_frm, Cd_w, Ch_w, Ce_w, t_zu_w, q_zu_w,    &
      &                         CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
      REAL(wp), INTENT(in )                 :: zt, zu
      REAL(wp), INTENT(in ), DIMENSION(:,:) :: Ts_i, t_zt, qs_i, q_zt, U_zu, frice
      REAL(wp), INTENT(out), DIMENSION(:,:) :: Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu
      REAL(wp), INTENT(in ), DIMENSION(:,:), OPTIONAL :: Ts_w, qs_w
      REAL(wp), INTENT(out), DIMENSION(:,:), OPTIONAL :: CdN_frm, Cd_w, Ch_w, Ce_w, t_zu_w, q_zu_w
      REAL(wp), INTENT(out), DIMENSION(:,:), OPTIONAL :: CdN, ChN, CeN, xz0, xu_star, xL, xUN10
      IF( PRESENT(Cd_w) .AND. PRESENT(Ch_w) .AND. PRESENT(Ce_w) .AND. PRESENT(t_zu_w) .AND. PRESENT(q_zu_w) ) THEN
         IF( .NOT.(PRESENT(Ts_w)) .OR. .NOT.(PRESENT(qs_w)) ) THEN
            PRINT *, ' ERROR: turb_ice_lg15_io@mod_blk_ice_lg15_io => you must specify "Ts_w" and "qs_w" as input'
            STOP
         END IF
         PRINT *, ' ERROR: turb_ice_lg15_io@mod_blk_ice_lg15_io => the over-water outputs (Cd_w, Ch_w, Ce_w, t_zu_w, q_zu_w)'
         PRINT *, '        are computed by the reference from unassigned work arrays; they are not available here.'
         STOP
      END IF
      CALL ab_ice_generic( 4, zt, zu, Ts_i, t_zt, qs_i, q_zt, U_zu, Cd_i, Ch_i, Ce_i, t_zu_i, q_zu_i, Ubzu, frice=frice, &
         &                 CdN=CdN, ChN=ChN, CeN=CeN, xz0=xz0, xu_star=xu_star, xL=xL, xUN10=xUN10, CdN_frm=CdN_frm )
   END SUBROUTINE TURB_ICE_LG15_IO
END MODULE mod_blk_ice_lg15_io
