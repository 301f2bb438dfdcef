! mod_blk_turb.f90 -- Fortran hosts of the TURB_* routines of the MI355X-native engine.
!
! Drop-in for callers that USE the reference's bulk-algorithm modules directly (NEMO's sbcblk, the reference's station
! drivers src/tests/test_aerobulk_buoy_series_oce.f90:452-487, aerobulk_toy.F90:324-393):
!    mod_blk_coare3p6 : TURB_COARE3P6   (reference: src/mod_blk_coare3p6.f90:123-131)
!    mod_blk_coare3p0 : TURB_COARE3P0   (src/mod_blk_coare3p0.f90:54-59)
!    mod_blk_ecmwf    : TURB_ECMWF      (src/mod_blk_ecmwf.f90:63-67)
!    mod_blk_ncar     : TURB_NCAR       (src/mod_blk_ncar.f90:57-59)
!    mod_blk_andreas  : TURB_ANDREAS    (src/mod_blk_andreas.f90:66-68)
! Same module names, routine names, dummy-argument names (keyword calls keep working), kinds, INTENTs and OPTIONALs.
! Nothing is computed here: every call goes through ISO_C_BINDING to `ab_turb` (include/aerobulk_amd.h), i.e. to the
! HIP kernel turb_kernel (aerobulk_amd/csrc/ab_turb_kernels.hip).  `nb_iter` is read from mod_const like the reference
! does (mod_const.f90:33).  Compile after mod_aerobulk.f90 (which provides mod_const), with -fdefault-real-8.

MODULE mod_ab_turb
   !! shared ISO_C_BINDING plumbing of the five modules below (not part of the reference's interface)
   USE, INTRINSIC :: ISO_C_BINDING
   USE mod_const, ONLY: wp, nb_iter
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: ab_turb_generic, ab_fn_arrays, ab_fn_scalar, ab_fn_pointers, ab_wl_fetch

   !! mirror of `ab_diag` (include/aerobulk_amd.h)
   TYPE, BIND(C) :: ab_diag
      TYPE(C_PTR) :: Cd, Ch, Ce, t_zu, q_zu, Ubzu, CdN, ChN, CeN, z0, u_star, L, UN10, dT_cs, dT_wl, Hz_wl
   END TYPE ab_diag

   INTERFACE
      FUNCTION ab_turb( algo, kt, zt, zu, use_cs, use_wl, niter, isecday_utc, lon, T_s, theta_zt, q_s, q_zt, U_zu,  &
         &              Qsw, rad_lw, slp, Cd, Ch, Ce, t_zu, q_zu, Ubzu, opt, ni, nj ) BIND(C, NAME='ab_turb') RESULT(istat)
         IMPORT :: C_INT, C_LONG, C_DOUBLE, C_PTR, ab_diag
         INTEGER(C_INT),  VALUE :: algo, kt, use_cs, use_wl, niter, isecday_utc
         REAL(C_DOUBLE),  VALUE :: zt, zu
         TYPE(C_PTR),     VALUE :: lon, T_s, theta_zt, q_s, q_zt, U_zu, Qsw, rad_lw, slp, Cd, Ch, Ce, t_zu, q_zu, Ubzu
         TYPE(ab_diag), INTENT(in) :: opt
         INTEGER(C_LONG), VALUE :: ni, nj
         INTEGER(C_INT) :: istat
      END FUNCTION ab_turb
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
      FUNCTION ab_turb_get_wl_state( algo, dT_wl, Hz_wl, Qnt_ac, Tau_ac, n ) BIND(C, NAME='ab_turb_get_wl_state') RESULT(istat)
         IMPORT :: C_INT, C_LONG, C_PTR
         INTEGER(C_INT),  VALUE :: algo
         TYPE(C_PTR),     VALUE :: dT_wl, Hz_wl, Qnt_ac, Tau_ac
         INTEGER(C_LONG), VALUE :: n
         INTEGER(C_INT) :: istat
      END FUNCTION ab_turb_get_wl_state
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

   SUBROUTINE ab_fn_arrays( fn, n, a1, o1, a2 )
      !! the PUBLIC functions of the algorithm modules (psi_m / psi_h of the five algorithms, the Charnock parameters, NCAR's neutral coefficients,
      !! ANDREAS' friction velocity) on the engine: helper functions 42-55 of `ab_phymbl` (include/aerobulk_amd.h), one elementwise HIP kernel each
      INTEGER, INTENT(in) :: fn, n
      REAL(wp), DIMENSION(n), INTENT(in),  TARGET           :: a1
      REAL(wp), DIMENSION(n), INTENT(out), TARGET           :: o1
      REAL(wp), DIMENSION(n), INTENT(in),  TARGET, OPTIONAL :: a2
      TYPE(C_PTR), DIMENSION(2) :: pin
      TYPE(C_PTR), DIMENSION(1) :: pout
      REAL(C_DOUBLE), DIMENSION(2) :: par, zinfo
      INTEGER(C_INT) :: istat
      pin = (/ C_LOC(a1), C_NULL_PTR /) ; pout = (/ C_LOC(o1) /) ; par = 0._C_DOUBLE
      IF( PRESENT(a2) ) pin(2) = C_LOC(a2)
      istat = ab_phymbl( INT(fn,C_INT), INT(n,C_LONG), pin, 2_C_INT, pout, 1_C_INT, par, 0_C_INT, 0_C_INT, C_NULL_PTR, zinfo )
      IF( istat /= 0 ) CALL stop_with_library_message()
   END SUBROUTINE ab_fn_arrays

   SUBROUTINE ab_fn_pointers( fn, n, pin, pout, par1, par2, iflag )
      !! any helper function on arrays handed over as C pointers (FIRST_GUESS_COARE: six in, seven out; the skin schemes)
      INTEGER, INTENT(in) :: fn, n
      TYPE(C_PTR), DIMENSION(:), INTENT(in) :: pin, pout
      REAL(wp), INTENT(in) :: par1, par2
      INTEGER,  INTENT(in), OPTIONAL :: iflag
      REAL(C_DOUBLE), DIMENSION(2) :: par, zinfo
      INTEGER(C_INT) :: istat, kflag
      par = (/ REAL(par1,C_DOUBLE), REAL(par2,C_DOUBLE) /)
      kflag = 0_C_INT
      IF( PRESENT(iflag) ) kflag = INT(iflag,C_INT)
      istat = ab_phymbl( INT(fn,C_INT), INT(n,C_LONG), pin, INT(SIZE(pin),C_INT), pout, INT(SIZE(pout),C_INT), par, kflag, 0_C_INT, &
         &               C_NULL_PTR, zinfo )
      IF( istat /= 0 ) CALL stop_with_library_message()
   END SUBROUTINE ab_fn_pointers

   SUBROUTINE ab_wl_fetch( ialgo, n, p1, p2, p3, p4 )
      !! the warm-layer state TURB_<algo> left in the engine's session -> the caller's (module) arrays; unassociated pointers are skipped
      INTEGER, INTENT(in) :: ialgo, n
      TYPE(C_PTR), INTENT(in) :: p1, p2, p3, p4
      INTEGER(C_INT) :: istat
      istat = ab_turb_get_wl_state( INT(ialgo,C_INT), p1, p2, p3, p4, INT(n,C_LONG) )
      IF( istat /= 0 ) CALL stop_with_library_message()
   END SUBROUTINE ab_wl_fetch

   FUNCTION ab_fn_scalar( fn, x1, x2 )
      !! the scalar specifics: a one-cell array through the same kernel (the engine has no host arithmetic)
      INTEGER,  INTENT(in)           :: fn
      REAL(wp), INTENT(in)           :: x1
      REAL(wp), INTENT(in), OPTIONAL :: x2
      REAL(wp) :: ab_fn_scalar
      REAL(wp), DIMENSION(1) :: a1, a2, o1
      a1(1) = x1
      IF( PRESENT(x2) ) THEN
         a2(1) = x2
         CALL ab_fn_arrays( fn, 1, a1, o1, a2 )
      ELSE
         CALL ab_fn_arrays( fn, 1, a1, o1 )
      END IF
      ab_fn_scalar = o1(1)
   END FUNCTION ab_fn_scalar

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


   SUBROUTINE ab_turb_generic( ialgo, kt, zt, zu, T_s, t_zt, q_s, q_zt, U_zu, l_use_cs, l_use_wl, &
      &                        Cd, Ch, Ce, t_zu, q_zu, Ubzu,                                       &
      &                        Qsw, rad_lw, slp, pdT_cs, isecday_utc, plong, pdT_wl, pHz_wl,       &
      &                        CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
      INTEGER,                  INTENT(in)    :: ialgo, kt
      REAL(wp),                 INTENT(in)    :: zt, zu
      REAL(wp), DIMENSION(:,:), INTENT(inout) :: T_s, q_s
      REAL(wp), DIMENSION(:,:), INTENT(in)    :: t_zt, q_zt, U_zu
      LOGICAL,                  INTENT(in)    :: l_use_cs, l_use_wl
      REAL(wp), DIMENSION(:,:), INTENT(out)   :: Cd, Ch, Ce, t_zu, q_zu, Ubzu
      REAL(wp), DIMENSION(:,:), INTENT(in),  OPTIONAL :: Qsw, rad_lw, slp, plong
      INTEGER,                  INTENT(in),  OPTIONAL :: isecday_utc
      REAL(wp), DIMENSION(:,:), INTENT(out), OPTIONAL :: pdT_cs, pdT_wl, pHz_wl, CdN, ChN, CeN, xz0, xu_star, xL, xUN10
      INTEGER :: Ni, Nj, isd
      Ni = SIZE(T_s,1) ; Nj = SIZE(T_s,2)
      isd = 12
      IF( PRESENT(isecday_utc) ) isd = isecday_utc
      !! explicit-shape dummies: the compiler hands over contiguous storage (copy-in/out only for strided actuals)
      CALL turb_contig( Ni*Nj, T_s, t_zt, q_s, q_zt, U_zu, Cd, Ch, Ce, t_zu, q_zu, Ubzu, Qsw, rad_lw, slp, plong, &
         &              pdT_cs, pdT_wl, pHz_wl, CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
   CONTAINS
      SUBROUTINE turb_contig( n, a1, a2, a3, a4, a5, o1, o2, o3, o4, o5, o6, r1, r2, r3, r4, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10 )
         INTEGER, INTENT(in) :: n
         REAL(wp), DIMENSION(n), INTENT(inout), TARGET :: a1, a3
         REAL(wp), DIMENSION(n), INTENT(in),    TARGET :: a2, a4, a5
         REAL(wp), DIMENSION(n), INTENT(out),   TARGET :: o1, o2, o3, o4, o5, o6
         REAL(wp), DIMENSION(n), INTENT(in),    TARGET, OPTIONAL :: r1, r2, r3, r4
         REAL(wp), DIMENSION(n), INTENT(out),   TARGET, OPTIONAL :: d1, d2, d3, d4, d5, d6, d7, d8, d9, d10
         TYPE(ab_diag) :: opt
         TYPE(C_PTR)   :: c1, c2, c3, c4
         INTEGER(C_INT) :: istat
         c1 = C_NULL_PTR ; c2 = C_NULL_PTR ; c3 = C_NULL_PTR ; c4 = C_NULL_PTR
         IF( PRESENT(r1) ) c1 = C_LOC(r1)
         IF( PRESENT(r2) ) c2 = C_LOC(r2)
         IF( PRESENT(r3) ) c3 = C_LOC(r3)
         IF( PRESENT(r4) ) c4 = C_LOC(r4)
         opt%Cd = C_NULL_PTR ; opt%Ch = C_NULL_PTR ; opt%Ce = C_NULL_PTR
         opt%t_zu = C_NULL_PTR ; opt%q_zu = C_NULL_PTR ; opt%Ubzu = C_NULL_PTR
         opt%dT_cs = C_NULL_PTR ; opt%dT_wl = C_NULL_PTR ; opt%Hz_wl = C_NULL_PTR
         opt%CdN = C_NULL_PTR ; opt%ChN = C_NULL_PTR ; opt%CeN = C_NULL_PTR
         opt%z0 = C_NULL_PTR ; opt%u_star = C_NULL_PTR ; opt%L = C_NULL_PTR ; opt%UN10 = C_NULL_PTR
         !! pdT_cs is only assigned when the cool skin runs, pdT_wl / pHz_wl when the warm layer does
         !! (mod_blk_coare3p6.f90:359,406-407): otherwise the caller's arrays are left untouched
         IF( PRESENT(d1) .AND. l_use_cs ) opt%dT_cs = C_LOC(d1)
         IF( PRESENT(d2) .AND. l_use_wl ) opt%dT_wl = C_LOC(d2)
         IF( PRESENT(d3) .AND. l_use_wl ) opt%Hz_wl = C_LOC(d3)
         IF( PRESENT(d4) ) opt%CdN    = C_LOC(d4)
         IF( PRESENT(d5) ) opt%ChN    = C_LOC(d5)
         IF( PRESENT(d6) ) opt%CeN    = C_LOC(d6)
         IF( PRESENT(d7) ) opt%z0     = C_LOC(d7)
         IF( PRESENT(d8) ) opt%u_star = C_LOC(d8)
         IF( PRESENT(d9) ) opt%L      = C_LOC(d9)
         IF( PRESENT(d10)) opt%UN10   = C_LOC(d10)
         istat = ab_turb( INT(ialgo,C_INT), INT(kt,C_INT), REAL(zt,C_DOUBLE), REAL(zu,C_DOUBLE),            &
            &             MERGE(1_C_INT,0_C_INT,l_use_cs), MERGE(1_C_INT,0_C_INT,l_use_wl), INT(nb_iter,C_INT), &
            &             INT(isd,C_INT), c4, C_LOC(a1), C_LOC(a2), C_LOC(a3), C_LOC(a4), C_LOC(a5), c1, c2, c3, &
            &             C_LOC(o1), C_LOC(o2), C_LOC(o3), C_LOC(o4), C_LOC(o5), C_LOC(o6), opt,             &
            &             INT(Ni,C_LONG), INT(Nj,C_LONG) )
         IF( istat /= 0 ) CALL stop_with_library_message()
      END SUBROUTINE turb_contig
   END SUBROUTINE ab_turb_generic

END MODULE mod_ab_turb


MODULE mod_skin_coare
   !! reference: src/mod_skin_coare.f90 — CS_COARE (:48-93) and WL_COARE (:97-250) standing alone, and the warm layer's state as PUBLIC module
   !! arrays (:31-36).  Each call is ONE cell through the engine's own device functions (helper functions 57 / 59 of `ab_phymbl`,
   !! include/aerobulk_amd.h): the arithmetic TURB_COARE3P0 / 3P6 iterate with.  The module arrays are a MIRROR of the state the engine keeps in
   !! HBM: TURB_COARE3P0 / TURB_COARE3P6 copy it here after every call with l_use_wl (callers such as
   !! src/tests/test_aerobulk_buoy_series_oce.f90:16,463-464 read Qnt_ac / Tau_ac between calls); WL_COARE updates the mirror like the
   !! reference updates its arrays.  What a caller writes into them is NOT carried back into the next TURB_* call (the engine's state lives
   !! on the device); and the arrays stay allocated after kt = nitend, where the reference frees them (mod_blk_coare3p6.f90:110).
   USE, INTRINSIC :: ISO_C_BINDING
   USE mod_const, ONLY: wp
   USE mod_ab_turb
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: CS_COARE, WL_COARE, ab_mirror_wl_coare
   REAL(wp), ALLOCATABLE, SAVE, DIMENSION(:,:), PUBLIC, TARGET :: dT_wl, Hz_wl, Qnt_ac, Tau_ac
   REAL(wp), PARAMETER, PUBLIC :: Hwl_max = 20._wp    !: maximum depth of warm layer (mod_skin_coare.f90:38)
CONTAINS
   SUBROUTINE CS_COARE( pQsw, pQnsol, pustar, pSST, pQlat, pdT_cs )
      REAL(wp), INTENT(in)  :: pQsw, pQnsol, pustar, pSST, pQlat
      REAL(wp), INTENT(out) :: pdT_cs
      REAL(wp), DIMENSION(1), TARGET :: a1, a2, a3, a4, a5, o1
      a1(1) = pQsw ; a2(1) = pQnsol ; a3(1) = pustar ; a4(1) = pSST ; a5(1) = pQlat
      CALL ab_fn_pointers( 57, 1, (/ C_LOC(a1), C_LOC(a2), C_LOC(a3), C_LOC(a4), C_LOC(a5) /), (/ C_LOC(o1) /), 0._wp, 0._wp )
      pdT_cs = o1(1)
   END SUBROUTINE CS_COARE

   SUBROUTINE WL_COARE( ki, kj, pQsw, pQnsol, pTau, pSST, plon, isd, iwait )
      INTEGER , INTENT(in) :: ki, kj
      REAL(wp), INTENT(in) :: pQsw, pQnsol, pTau, pSST, plon
      INTEGER , INTENT(in) :: isd, iwait
      REAL(wp), DIMENSION(1), TARGET :: a1, a2, a3, a4, a5, s1, s2, s3, s4
      IF( .NOT. ALLOCATED(dT_wl) ) THEN
         WRITE(6,*) ' *** E R R O R :  [WL_COARE] => dT_wl, Hz_wl, Qnt_ac & Tau_ac are not allocated (the reference allocates them in COARE3P6_INIT)'
         STOP
      END IF
      a1(1) = pQsw ; a2(1) = pQnsol ; a3(1) = pTau ; a4(1) = pSST ; a5(1) = plon
      s1(1) = dT_wl(ki,kj) ; s2(1) = Hz_wl(ki,kj) ; s3(1) = Qnt_ac(ki,kj) ; s4(1) = Tau_ac(ki,kj)
      CALL ab_fn_pointers( 59, 1, (/ C_LOC(a1), C_LOC(a2), C_LOC(a3), C_LOC(a4), C_LOC(a5), C_LOC(s1), C_LOC(s2), C_LOC(s3), C_LOC(s4) /), &
         &                 (/ C_LOC(s1), C_LOC(s2), C_LOC(s3), C_LOC(s4) /), REAL(isd,wp), 0._wp, iflag=iwait )
      IF( iwait == 0 ) THEN           ! mod_skin_coare.f90:239-248
         dT_wl(ki,kj) = s1(1) ; Hz_wl(ki,kj) = s2(1) ; Qnt_ac(ki,kj) = s3(1) ; Tau_ac(ki,kj) = s4(1)
      END IF
   END SUBROUTINE WL_COARE

   SUBROUTINE ab_mirror_wl_coare( ialgo, nx, ny )
      !! (not part of the reference's interface) called by TURB_COARE3P0 / 3P6 after a call with the warm layer on
      INTEGER, INTENT(in) :: ialgo, nx, ny
      IF( ALLOCATED(dT_wl) ) THEN
         IF( SIZE(dT_wl,1) /= nx .OR. SIZE(dT_wl,2) /= ny ) DEALLOCATE( dT_wl, Hz_wl, Qnt_ac, Tau_ac )
      END IF
      IF( .NOT. ALLOCATED(dT_wl) ) ALLOCATE( dT_wl(nx,ny), Hz_wl(nx,ny), Qnt_ac(nx,ny), Tau_ac(nx,ny) )
      CALL ab_wl_fetch( ialgo, nx*ny, C_LOC(dT_wl), C_LOC(Hz_wl), C_LOC(Qnt_ac), C_LOC(Tau_ac) )
   END SUBROUTINE ab_mirror_wl_coare
END MODULE mod_skin_coare


MODULE mod_skin_ecmwf
   !! reference: src/mod_skin_ecmwf.f90 — CS_ECMWF (:68-110), WL_ECMWF (:113-230) and the warm layer's PUBLIC module arrays (:52-55), on the same
   !! terms as mod_skin_coare above (helper functions 58 / 60; mirror filled by TURB_ECMWF and by ECMWF_INIT's initial values)
   USE, INTRINSIC :: ISO_C_BINDING
   USE mod_const, ONLY: wp
   USE mod_ab_turb
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: CS_ECMWF, WL_ECMWF, ab_mirror_wl_ecmwf, ab_alloc_wl_ecmwf
   REAL(wp), ALLOCATABLE, SAVE, DIMENSION(:,:), PUBLIC, TARGET :: dT_wl, Hz_wl
   REAL(wp), PARAMETER, PUBLIC :: rd0 = 3.            !: depth scale [m] of the warm layer (mod_skin_ecmwf.f90:57)
CONTAINS
   SUBROUTINE CS_ECMWF( pQsw, pQnsol, pustar, pSST, pdT_cs )
      REAL(wp), INTENT(in)  :: pQsw, pQnsol, pustar, pSST
      REAL(wp), INTENT(out) :: pdT_cs
      REAL(wp), DIMENSION(1), TARGET :: a1, a2, a3, a4, o1
      a1(1) = pQsw ; a2(1) = pQnsol ; a3(1) = pustar ; a4(1) = pSST
      CALL ab_fn_pointers( 58, 1, (/ C_LOC(a1), C_LOC(a2), C_LOC(a3), C_LOC(a4) /), (/ C_LOC(o1) /), 0._wp, 0._wp )
      pdT_cs = o1(1)
   END SUBROUTINE CS_ECMWF

   SUBROUTINE WL_ECMWF( ki, kj, pQsw, pQnsol, pustar, pSST,  pustk )
      INTEGER , INTENT(in) :: ki, kj
      REAL(wp), INTENT(in) :: pQsw, pQnsol, pustar, pSST
      REAL(wp), OPTIONAL, INTENT(in) :: pustk
      REAL(wp), DIMENSION(1), TARGET :: a1, a2, a3, a4, a7, s1, s2
      TYPE(C_PTR) :: c7
      IF( .NOT. ALLOCATED(dT_wl) ) THEN
         WRITE(6,*) ' *** E R R O R :  [WL_ECMWF] => dT_wl & Hz_wl are not allocated (the reference allocates them in ECMWF_INIT)'
         STOP
      END IF
      a1(1) = pQsw ; a2(1) = pQnsol ; a3(1) = pustar ; a4(1) = pSST
      s1(1) = dT_wl(ki,kj) ; s2(1) = Hz_wl(ki,kj)
      c7 = C_NULL_PTR
      IF( PRESENT(pustk) ) THEN
         a7(1) = pustk ; c7 = C_LOC(a7)
      END IF
      CALL ab_fn_pointers( 60, 1, (/ C_LOC(a1), C_LOC(a2), C_LOC(a3), C_LOC(a4), C_LOC(s1), C_LOC(s2), c7 /), (/ C_LOC(s1) /), 0._wp, 0._wp )
      dT_wl(ki,kj) = s1(1)             ! every call: mod_skin_ecmwf.f90:228
   END SUBROUTINE WL_ECMWF

   SUBROUTINE ab_alloc_wl_ecmwf( nx, ny )
      !! (not part of the reference's interface) ECMWF_INIT's allocation and initial values, mod_blk_ecmwf.f90:401-404
      INTEGER, INTENT(in) :: nx, ny
      IF( ALLOCATED(dT_wl) ) THEN
         IF( SIZE(dT_wl,1) /= nx .OR. SIZE(dT_wl,2) /= ny ) DEALLOCATE( dT_wl, Hz_wl )
      END IF
      IF( .NOT. ALLOCATED(dT_wl) ) THEN
         ALLOCATE( dT_wl(nx,ny), Hz_wl(nx,ny) )
         dT_wl(:,:) = 0._wp
         Hz_wl(:,:) = rd0
      END IF
   END SUBROUTINE ab_alloc_wl_ecmwf

   SUBROUTINE ab_mirror_wl_ecmwf( nx, ny )
      !! (not part of the reference's interface) called by TURB_ECMWF after a call with the warm layer on
      INTEGER, INTENT(in) :: nx, ny
      CALL ab_alloc_wl_ecmwf( nx, ny )
      CALL ab_wl_fetch( 4, nx*ny, C_LOC(dT_wl), C_LOC(Hz_wl), C_NULL_PTR, C_NULL_PTR )
   END SUBROUTINE ab_mirror_wl_ecmwf
END MODULE mod_skin_ecmwf


MODULE mod_common_coare
   !! the three public names of the reference's src/mod_common_coare.f90: FIRST_GUESS_COARE :33-214 (the first block of the engine's TURB_COARE* /
   !! TURB_ECMWF kernels, here as helper function 56 of `ab_phymbl`), psi_m_coare / psi_h_coare :217-392 (src/tests/test_psi_stab.f90:26)
   USE, INTRINSIC :: ISO_C_BINDING
   USE mod_const, ONLY: wp
   USE mod_ab_turb
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: first_guess_coare, psi_m_coare, psi_h_coare
   INTERFACE first_guess_coare
      MODULE PROCEDURE first_guess_coare_vctr, first_guess_coare_sclr
   END INTERFACE
   INTERFACE psi_m_coare
      MODULE PROCEDURE psi_m_coare_vctr, psi_m_coare_sclr
   END INTERFACE
   INTERFACE psi_h_coare
      MODULE PROCEDURE psi_h_coare_vctr, psi_h_coare_sclr
   END INTERFACE
CONTAINS
   FUNCTION psi_m_coare_vctr( pzeta )                                          !! reference mod_common_coare.f90:217-302
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pzeta
      REAL(wp), DIMENSION(SIZE(pzeta,1),SIZE(pzeta,2)) :: psi_m_coare_vctr
      CALL ab_fn_arrays( 42, SIZE(pzeta), pzeta, psi_m_coare_vctr )
   END FUNCTION psi_m_coare_vctr
   FUNCTION psi_m_coare_sclr( pzeta )
      REAL(wp), INTENT(in) :: pzeta
      REAL(wp) :: psi_m_coare_sclr
      psi_m_coare_sclr = ab_fn_scalar( 42, pzeta )
   END FUNCTION psi_m_coare_sclr
   FUNCTION psi_h_coare_vctr( pzeta )                                          !! reference mod_common_coare.f90:305-392
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pzeta
      REAL(wp), DIMENSION(SIZE(pzeta,1),SIZE(pzeta,2)) :: psi_h_coare_vctr
      CALL ab_fn_arrays( 43, SIZE(pzeta), pzeta, psi_h_coare_vctr )
   END FUNCTION psi_h_coare_vctr
   FUNCTION psi_h_coare_sclr( pzeta )
      REAL(wp), INTENT(in) :: pzeta
      REAL(wp) :: psi_h_coare_sclr
      psi_h_coare_sclr = ab_fn_scalar( 43, pzeta )
   END FUNCTION psi_h_coare_sclr
   SUBROUTINE FIRST_GUESS_COARE_VCTR( zt, zu, psst, t_zt, pssq, q_zt, U_zu, pcharn, &
      &                               pus, pts, pqs, t_zu, q_zu, Ubzu,  qz0 )
      REAL(wp), INTENT(in)                  ::   zt, zu
      REAL(wp), INTENT(in),  DIMENSION(:,:) ::   psst, t_zt, pssq, q_zt, U_zu, pcharn
      REAL(wp), INTENT(out), DIMENSION(:,:) ::   pus, pts, pqs, t_zu, q_zu, Ubzu
      REAL(wp), INTENT(out), DIMENSION(:,:), OPTIONAL :: qz0    ! roughness length [m]
      REAL(wp), DIMENSION(:,:), ALLOCATABLE, TARGET :: zi, zo   ! contiguous copies: (n,6) in, (n,7) out
      INTEGER :: n, k
      TYPE(C_PTR), DIMENSION(6) :: pin
      TYPE(C_PTR), DIMENSION(7) :: pout
      n = SIZE(psst)
      ALLOCATE( zi(n,6), zo(n,7) )
      zi(:,1) = RESHAPE(psst, (/n/)) ; zi(:,2) = RESHAPE(t_zt, (/n/)) ; zi(:,3) = RESHAPE(pssq, (/n/))
      zi(:,4) = RESHAPE(q_zt, (/n/)) ; zi(:,5) = RESHAPE(U_zu, (/n/)) ; zi(:,6) = RESHAPE(pcharn, (/n/))
      DO k = 1, 6
         pin(k) = C_LOC(zi(1,k))
      END DO
      DO k = 1, 7
         pout(k) = C_LOC(zo(1,k))
      END DO
      CALL ab_fn_pointers( 56, n, pin, pout, zt, zu )
      pus  = RESHAPE(zo(:,1), SHAPE(pus))  ; pts  = RESHAPE(zo(:,2), SHAPE(pts))  ; pqs  = RESHAPE(zo(:,3), SHAPE(pqs))
      t_zu = RESHAPE(zo(:,4), SHAPE(t_zu)) ; q_zu = RESHAPE(zo(:,5), SHAPE(q_zu)) ; Ubzu = RESHAPE(zo(:,6), SHAPE(Ubzu))
      IF( PRESENT(qz0) ) qz0 = RESHAPE(zo(:,7), SHAPE(qz0))
      DEALLOCATE( zi, zo )
   END SUBROUTINE FIRST_GUESS_COARE_VCTR

   SUBROUTINE FIRST_GUESS_COARE_SCLR( zt, zu, psst, t_zt, pssq, q_zt, U_zu, pcharn, &
      &                               pus, pts, pqs, t_zu, q_zu, Ubzu,  pz0 )
      REAL(wp), INTENT(in)  ::   zt, zu, psst, t_zt, pssq, q_zt, U_zu, pcharn
      REAL(wp), INTENT(out) ::   pus, pts, pqs, t_zu, q_zu, Ubzu
      REAL(wp), INTENT(out), OPTIONAL :: pz0
      REAL(wp), DIMENSION(1,1) :: a1, a2, a3, a4, a5, a6, o1, o2, o3, o4, o5, o6, o7
      a1 = psst ; a2 = t_zt ; a3 = pssq ; a4 = q_zt ; a5 = U_zu ; a6 = pcharn
      CALL FIRST_GUESS_COARE_VCTR( zt, zu, a1, a2, a3, a4, a5, a6, o1, o2, o3, o4, o5, o6, qz0=o7 )
      pus = o1(1,1) ; pts = o2(1,1) ; pqs = o3(1,1) ; t_zu = o4(1,1) ; q_zu = o5(1,1) ; Ubzu = o6(1,1)
      IF( PRESENT(pz0) ) pz0 = o7(1,1)
   END SUBROUTINE FIRST_GUESS_COARE_SCLR
END MODULE mod_common_coare


MODULE mod_blk_coare3p6
   USE mod_const, ONLY: wp
   USE mod_ab_turb
   USE mod_skin_coare, ONLY: ab_mirror_wl_coare
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_COARE3P6, charn_coare3p6
   INTERFACE charn_coare3p6
      MODULE PROCEDURE charn_coare3p6_vctr, charn_coare3p6_sclr
   END INTERFACE charn_coare3p6
CONTAINS
   FUNCTION charn_coare3p6_vctr( pwnd )                                          !! reference mod_blk_coare3p6.f90:417-445
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pwnd
      REAL(wp), DIMENSION(SIZE(pwnd,1),SIZE(pwnd,2)) :: charn_coare3p6_vctr
      CALL ab_fn_arrays( 51, SIZE(pwnd), pwnd, charn_coare3p6_vctr )
   END FUNCTION charn_coare3p6_vctr
   FUNCTION charn_coare3p6_sclr( pwnd )
      REAL(wp), INTENT(in) :: pwnd
      REAL(wp) :: charn_coare3p6_sclr
      charn_coare3p6_sclr = ab_fn_scalar( 51, pwnd )
   END FUNCTION charn_coare3p6_sclr

   SUBROUTINE TURB_COARE3P6( kt, zt, zu, T_s, t_zt, q_s, q_zt, U_zu, l_use_cs, l_use_wl, &
      &                      Cd, Ch, Ce, t_zu, q_zu, Ubzu,                               &
      &                      Qsw, rad_lw, slp, pdT_cs,                                   &
      &                      isecday_utc, plong, pdT_wl, pHz_wl,                         &
      &                      CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
      INTEGER,  INTENT(in   )                 ::   kt
      REAL(wp), INTENT(in   )                 ::   zt, zu
      REAL(wp), INTENT(inout), DIMENSION(:,:) ::   T_s, q_s
      REAL(wp), INTENT(in   ), DIMENSION(:,:) ::   t_zt, q_zt, U_zu
      LOGICAL , INTENT(in   )                 ::   l_use_cs, l_use_wl
      REAL(wp), INTENT(  out), DIMENSION(:,:) ::   Cd, Ch, Ce, t_zu, q_zu, Ubzu
      REAL(wp), INTENT(in   ), OPTIONAL, DIMENSION(:,:) ::   Qsw, rad_lw, slp
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   pdT_cs
      INTEGER,  INTENT(in   ), OPTIONAL                 ::   isecday_utc
      REAL(wp), INTENT(in   ), OPTIONAL, DIMENSION(:,:) ::   plong
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   pdT_wl, pHz_wl
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   CdN, ChN, CeN, xz0, xu_star, xL, xUN10
      IF( l_use_wl .AND. (.NOT.(PRESENT(isecday_utc) .AND. PRESENT(plong))) ) THEN   ! mod_blk_coare3p6.f90:266-269
         WRITE(6,*) ' *** E R R O R :  [turb_coare3p6] => you need to provide Qsw, rad_lw, slp, isecday_utc & plong to use warm-layer param!'
         STOP
      END IF
      CALL ab_turb_generic( 2, kt, zt, zu, T_s, t_zt, q_s, q_zt, U_zu, l_use_cs, l_use_wl, Cd, Ch, Ce, t_zu, q_zu, Ubzu, &
         &                  Qsw, rad_lw, slp, pdT_cs, isecday_utc, plong, pdT_wl, pHz_wl, CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
      IF( l_use_wl ) CALL ab_mirror_wl_coare( 2, SIZE(T_s,1), SIZE(T_s,2) )     ! mod_skin_coare's PUBLIC state arrays
   END SUBROUTINE TURB_COARE3P6
END MODULE mod_blk_coare3p6


MODULE mod_blk_coare3p0
   USE mod_const, ONLY: wp
   USE mod_ab_turb
   USE mod_skin_coare, ONLY: ab_mirror_wl_coare
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_COARE3P0, charn_coare3p0
CONTAINS
   FUNCTION charn_coare3p0( pwnd )                                             !! reference mod_blk_coare3p0.f90:420-447 (a scalar function there too)
      REAL(wp), INTENT(in) :: pwnd
      REAL(wp) :: charn_coare3p0
      charn_coare3p0 = ab_fn_scalar( 50, pwnd )
   END FUNCTION charn_coare3p0

   SUBROUTINE TURB_COARE3P0( kt, zt, zu, pT_s, pt_zt, pq_s, pq_zt, pU_zu, l_use_cs, l_use_wl, &
      &                      pCd, pCh, pCe, pt_zu, pq_zu, pUbzu,                             &
      &                      pQsw, prad_lw, pslp, pdT_cs,                                    &
      &                      isecday_utc, plong,                                             &
      &                      pdT_wl, pHz_wl,                                                 &
      &                      pCdN, pChN, pCeN, pz0, pu_star, pL, pUN10 )
      INTEGER,  INTENT(in   )                 ::   kt
      REAL(wp), INTENT(in   )                 ::   zt, zu
      REAL(wp), INTENT(inout), DIMENSION(:,:) ::   pT_s, pq_s
      REAL(wp), INTENT(in   ), DIMENSION(:,:) ::   pt_zt, pq_zt, pU_zu
      LOGICAL , INTENT(in   )                 ::   l_use_cs, l_use_wl
      REAL(wp), INTENT(  out), DIMENSION(:,:) ::   pCd, pCh, pCe, pt_zu, pq_zu, pUbzu
      REAL(wp), INTENT(in   ), OPTIONAL, DIMENSION(:,:) ::   pQsw, prad_lw, pslp
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   pdT_cs
      INTEGER,  INTENT(in   ), OPTIONAL                 ::   isecday_utc
      REAL(wp), INTENT(in   ), OPTIONAL, DIMENSION(:,:) ::   plong
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   pdT_wl, pHz_wl
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   pCdN, pChN, pCeN, pz0, pu_star, pL, pUN10
      IF( l_use_wl .AND. (.NOT.(PRESENT(isecday_utc) .AND. PRESENT(plong))) ) THEN   ! mod_blk_coare3p0.f90:201-204
         WRITE(6,*) ' *** E R R O R :  [turb_coare3p0] => you need to provide Qsw, rad_lw, slp, isecday_utc & plong to use warm-layer param!'
         STOP
      END IF
      CALL ab_turb_generic( 1, kt, zt, zu, pT_s, pt_zt, pq_s, pq_zt, pU_zu, l_use_cs, l_use_wl, pCd, pCh, pCe, pt_zu, pq_zu, pUbzu, &
         &                  pQsw, prad_lw, pslp, pdT_cs, isecday_utc, plong, pdT_wl, pHz_wl, pCdN, pChN, pCeN, pz0, pu_star, pL, pUN10 )
      IF( l_use_wl ) CALL ab_mirror_wl_coare( 1, SIZE(pT_s,1), SIZE(pT_s,2) )   ! mod_skin_coare's PUBLIC state arrays
   END SUBROUTINE TURB_COARE3P0
END MODULE mod_blk_coare3p0


MODULE mod_blk_ecmwf
   USE mod_const, ONLY: wp
   USE mod_ab_turb
   USE mod_skin_ecmwf, ONLY: ab_mirror_wl_ecmwf, ab_alloc_wl_ecmwf
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: ECMWF_INIT, TURB_ECMWF, psi_m_ecmwf, psi_h_ecmwf
   INTERFACE psi_m_ecmwf
      MODULE PROCEDURE psi_m_ecmwf_scl, psi_m_ecmwf_vct
   END INTERFACE
   INTERFACE psi_h_ecmwf
      MODULE PROCEDURE psi_h_ecmwf_scl, psi_h_ecmwf_vct
   END INTERFACE
CONTAINS
   SUBROUTINE ECMWF_INIT( nx, ny, l_use_wl )                                   !! reference mod_blk_ecmwf.f90:387-411
      !! The reference allocates the warm layer's module arrays here.  The engine keeps that state in its session, created by the first
      !! TURB_ECMWF call (kt = 1); the module arrays of mod_skin_ecmwf are a mirror of it (see there) and get their initial values here.
      INTEGER, INTENT(in) :: nx, ny
      LOGICAL, INTENT(in) :: l_use_wl
      IF( nx < 1 .OR. ny < 1 ) STOP 'ECMWF_INIT: bad shape'
      IF( l_use_wl ) CALL ab_alloc_wl_ecmwf( nx, ny )       ! mod_skin_ecmwf's PUBLIC arrays with their initial values (0, rd0)
   END SUBROUTINE ECMWF_INIT
   FUNCTION psi_m_ecmwf_vct( pzeta )                                          !! reference mod_blk_ecmwf.f90:441-495
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pzeta
      REAL(wp), DIMENSION(SIZE(pzeta,1),SIZE(pzeta,2)) :: psi_m_ecmwf_vct
      CALL ab_fn_arrays( 46, SIZE(pzeta), pzeta, psi_m_ecmwf_vct )
   END FUNCTION psi_m_ecmwf_vct
   FUNCTION psi_m_ecmwf_scl( pzeta )
      REAL(wp), INTENT(in) :: pzeta
      REAL(wp) :: psi_m_ecmwf_scl
      psi_m_ecmwf_scl = ab_fn_scalar( 46, pzeta )
   END FUNCTION psi_m_ecmwf_scl
   FUNCTION psi_h_ecmwf_vct( pzeta )                                          !! reference mod_blk_ecmwf.f90:498-548
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pzeta
      REAL(wp), DIMENSION(SIZE(pzeta,1),SIZE(pzeta,2)) :: psi_h_ecmwf_vct
      CALL ab_fn_arrays( 47, SIZE(pzeta), pzeta, psi_h_ecmwf_vct )
   END FUNCTION psi_h_ecmwf_vct
   FUNCTION psi_h_ecmwf_scl( pzeta )
      REAL(wp), INTENT(in) :: pzeta
      REAL(wp) :: psi_h_ecmwf_scl
      psi_h_ecmwf_scl = ab_fn_scalar( 47, pzeta )
   END FUNCTION psi_h_ecmwf_scl

   SUBROUTINE TURB_ECMWF(    kt, zt, zu, pT_s, pt_zt, pq_s, pq_zt, pU_zu, l_use_cs, l_use_wl, &
      &                      pCd, pCh, pCe, pt_zu, pq_zu, pUbzu,                             &
      &                      pQsw, prad_lw, pslp, pdT_cs,                                    &
      &                      pdT_wl, pHz_wl,                                                 &
      &                      pCdN, pChN, pCeN, pz0, pu_star, pL, pUN10 )
      INTEGER,  INTENT(in   )                 ::   kt
      REAL(wp), INTENT(in   )                 ::   zt, zu
      REAL(wp), INTENT(inout), DIMENSION(:,:) ::   pT_s, pq_s
      REAL(wp), INTENT(in   ), DIMENSION(:,:) ::   pt_zt, pq_zt, pU_zu
      LOGICAL , INTENT(in   )                 ::   l_use_cs, l_use_wl
      REAL(wp), INTENT(  out), DIMENSION(:,:) ::   pCd, pCh, pCe, pt_zu, pq_zu, pUbzu
      REAL(wp), INTENT(in   ), OPTIONAL, DIMENSION(:,:) ::   pQsw, prad_lw, pslp
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   pdT_cs, pdT_wl, pHz_wl
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   pCdN, pChN, pCeN, pz0, pu_star, pL, pUN10
      CALL ab_turb_generic( 4, kt, zt, zu, pT_s, pt_zt, pq_s, pq_zt, pU_zu, l_use_cs, l_use_wl, pCd, pCh, pCe, pt_zu, pq_zu, pUbzu, &
         &                  Qsw=pQsw, rad_lw=prad_lw, slp=pslp, pdT_cs=pdT_cs, pdT_wl=pdT_wl, pHz_wl=pHz_wl,                          &
         &                  CdN=pCdN, ChN=pChN, CeN=pCeN, xz0=pz0, xu_star=pu_star, xL=pL, xUN10=pUN10 )
      IF( l_use_wl ) CALL ab_mirror_wl_ecmwf( SIZE(pT_s,1), SIZE(pT_s,2) )      ! mod_skin_ecmwf's PUBLIC state arrays
   END SUBROUTINE TURB_ECMWF
END MODULE mod_blk_ecmwf


MODULE mod_blk_ncar
   USE mod_const, ONLY: wp
   USE mod_ab_turb
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_NCAR, cd_n10_ncar, ch_n10_ncar, ce_n10_ncar, psi_m_ncar, psi_h_ncar
   INTERFACE cd_n10_ncar
      MODULE PROCEDURE cd_n10_ncar_vctr, cd_n10_ncar_sclr
   END INTERFACE
   INTERFACE ch_n10_ncar
      MODULE PROCEDURE ch_n10_ncar_vctr, ch_n10_ncar_sclr
   END INTERFACE
   INTERFACE ce_n10_ncar
      MODULE PROCEDURE ce_n10_ncar_vctr, ce_n10_ncar_sclr
   END INTERFACE
   INTERFACE psi_m_ncar
      MODULE PROCEDURE psi_m_ncar_vctr, psi_m_ncar_sclr
   END INTERFACE
   INTERFACE psi_h_ncar
      MODULE PROCEDURE psi_h_ncar_vctr, psi_h_ncar_sclr
   END INTERFACE
CONTAINS
   FUNCTION cd_n10_ncar_vctr( pw10 )                                          !! reference mod_blk_ncar.f90:244-284
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pw10
      REAL(wp), DIMENSION(SIZE(pw10,1),SIZE(pw10,2)) :: cd_n10_ncar_vctr
      CALL ab_fn_arrays( 52, SIZE(pw10), pw10, cd_n10_ncar_vctr )
   END FUNCTION cd_n10_ncar_vctr
   FUNCTION cd_n10_ncar_sclr( pw10 )
      REAL(wp), INTENT(in) :: pw10
      REAL(wp) :: cd_n10_ncar_sclr
      cd_n10_ncar_sclr = ab_fn_scalar( 52, pw10 )
   END FUNCTION cd_n10_ncar_sclr
   FUNCTION ch_n10_ncar_vctr( psqrtcdn10 , pstab )                            !! reference mod_blk_ncar.f90:287-310
      REAL(wp), DIMENSION(:,:), INTENT(in) :: psqrtcdn10, pstab
      REAL(wp), DIMENSION(SIZE(psqrtcdn10,1),SIZE(psqrtcdn10,2)) :: ch_n10_ncar_vctr
      CALL ab_fn_arrays( 53, SIZE(psqrtcdn10), psqrtcdn10, ch_n10_ncar_vctr, pstab )
   END FUNCTION ch_n10_ncar_vctr
   FUNCTION ch_n10_ncar_sclr( psqrtcdn10 , pstab )
      REAL(wp), INTENT(in) :: psqrtcdn10, pstab
      REAL(wp) :: ch_n10_ncar_sclr
      IF( (pstab < -0.00001).OR.(pstab >  1.00001) ) THEN                      ! (the scalar version's own check, :292-296)
         PRINT *, 'ERROR: ch_n10_ncar_sclr@mod_blk_ncar.f90: pstab ='
         PRINT *, pstab
         STOP
      END IF
      ch_n10_ncar_sclr = ab_fn_scalar( 53, psqrtcdn10, pstab )
   END FUNCTION ch_n10_ncar_sclr
   FUNCTION ce_n10_ncar_vctr( psqrtcdn10 )                                          !! reference mod_blk_ncar.f90:313-330
      REAL(wp), DIMENSION(:,:), INTENT(in) :: psqrtcdn10
      REAL(wp), DIMENSION(SIZE(psqrtcdn10,1),SIZE(psqrtcdn10,2)) :: ce_n10_ncar_vctr
      CALL ab_fn_arrays( 54, SIZE(psqrtcdn10), psqrtcdn10, ce_n10_ncar_vctr )
   END FUNCTION ce_n10_ncar_vctr
   FUNCTION ce_n10_ncar_sclr( psqrtcdn10 )
      REAL(wp), INTENT(in) :: psqrtcdn10
      REAL(wp) :: ce_n10_ncar_sclr
      ce_n10_ncar_sclr = ab_fn_scalar( 54, psqrtcdn10 )
   END FUNCTION ce_n10_ncar_sclr
   FUNCTION psi_m_ncar_vctr( pzeta )                                          !! reference mod_blk_ncar.f90:333-376
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pzeta
      REAL(wp), DIMENSION(SIZE(pzeta,1),SIZE(pzeta,2)) :: psi_m_ncar_vctr
      CALL ab_fn_arrays( 44, SIZE(pzeta), pzeta, psi_m_ncar_vctr )
   END FUNCTION psi_m_ncar_vctr
   FUNCTION psi_m_ncar_sclr( pzeta )
      REAL(wp), INTENT(in) :: pzeta
      REAL(wp) :: psi_m_ncar_sclr
      psi_m_ncar_sclr = ab_fn_scalar( 44, pzeta )
   END FUNCTION psi_m_ncar_sclr
   FUNCTION psi_h_ncar_vctr( pzeta )                                          !! reference mod_blk_ncar.f90:379-420
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pzeta
      REAL(wp), DIMENSION(SIZE(pzeta,1),SIZE(pzeta,2)) :: psi_h_ncar_vctr
      CALL ab_fn_arrays( 45, SIZE(pzeta), pzeta, psi_h_ncar_vctr )
   END FUNCTION psi_h_ncar_vctr
   FUNCTION psi_h_ncar_sclr( pzeta )
      REAL(wp), INTENT(in) :: pzeta
      REAL(wp) :: psi_h_ncar_sclr
      psi_h_ncar_sclr = ab_fn_scalar( 45, pzeta )
   END FUNCTION psi_h_ncar_sclr

   SUBROUTINE TURB_NCAR( zt, zu, sst, t_zt, ssq, q_zt, U_zu,   &
      &                        Cd, Ch, Ce, t_zu, q_zu, Ubzu,   &
      &                  CdN, ChN, CeN, xz0, xu_star, xL, xUN10 )
      REAL(wp), INTENT(in   )                 ::   zt, zu
      REAL(wp), INTENT(in   ), DIMENSION(:,:) ::   sst, t_zt, ssq, q_zt, U_zu
      REAL(wp), INTENT(  out), DIMENSION(:,:) ::   Cd, Ch, Ce, t_zu, q_zu, Ubzu
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   CdN, ChN, CeN, xz0, xu_star, xL, xUN10
      REAL(wp), DIMENSION(:,:), ALLOCATABLE :: zTs, zqs   ! sst, ssq are INTENT(in) here; the engine's are INOUT
      ALLOCATE( zTs(SIZE(sst,1),SIZE(sst,2)), zqs(SIZE(sst,1),SIZE(sst,2)) )
      zTs = sst ; zqs = ssq
      CALL ab_turb_generic( 3, 1, zt, zu, zTs, t_zt, zqs, q_zt, U_zu, .FALSE., .FALSE., Cd, Ch, Ce, t_zu, q_zu, Ubzu, &
         &                  CdN=CdN, ChN=ChN, CeN=CeN, xz0=xz0, xu_star=xu_star, xL=xL, xUN10=xUN10 )
      DEALLOCATE( zTs, zqs )
   END SUBROUTINE TURB_NCAR
END MODULE mod_blk_ncar


MODULE mod_blk_andreas
   USE mod_const, ONLY: wp
   USE mod_ab_turb
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_ANDREAS, u_star_andreas, psi_m_andreas, psi_h_andreas
   INTERFACE u_star_andreas
      MODULE PROCEDURE u_star_andreas_vctr, u_star_andreas_sclr
   END INTERFACE
CONTAINS
   FUNCTION u_star_andreas_vctr( pun10 )                                          !! reference mod_blk_andreas.f90:275-305
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pun10
      REAL(wp), DIMENSION(SIZE(pun10,1),SIZE(pun10,2)) :: u_star_andreas_vctr
      CALL ab_fn_arrays( 55, SIZE(pun10), pun10, u_star_andreas_vctr )
   END FUNCTION u_star_andreas_vctr
   FUNCTION u_star_andreas_sclr( pun10 )
      REAL(wp), INTENT(in) :: pun10
      REAL(wp) :: u_star_andreas_sclr
      u_star_andreas_sclr = ab_fn_scalar( 55, pun10 )
   END FUNCTION u_star_andreas_sclr
   FUNCTION psi_m_andreas( pzeta )                                             !! reference mod_blk_andreas.f90:307-360 (array functions there too)
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pzeta
      REAL(wp), DIMENSION(SIZE(pzeta,1),SIZE(pzeta,2)) :: psi_m_andreas
      CALL ab_fn_arrays( 48, SIZE(pzeta), pzeta, psi_m_andreas )
   END FUNCTION psi_m_andreas
   FUNCTION psi_h_andreas( pzeta )                                             !! :363-410
      REAL(wp), DIMENSION(:,:), INTENT(in) :: pzeta
      REAL(wp), DIMENSION(SIZE(pzeta,1),SIZE(pzeta,2)) :: psi_h_andreas
      CALL ab_fn_arrays( 49, SIZE(pzeta), pzeta, psi_h_andreas )
   END FUNCTION psi_h_andreas

   SUBROUTINE TURB_ANDREAS( zt, zu, psst, pt_zt, pssq, pq_zt, pU_zu, &
      &                     pCd, pCh, pCe, pt_zu, pq_zu, pUbzu,       &
      &                    pCdN, pChN, pCeN, pz0, pu_star, pL, pUN10 )
      REAL(wp), INTENT(in   )                 ::   zt, zu
      REAL(wp), INTENT(in   ), DIMENSION(:,:) ::   psst, pt_zt, pssq, pq_zt, pU_zu
      REAL(wp), INTENT(  out), DIMENSION(:,:) ::   pCd, pCh, pCe, pt_zu, pq_zu, pUbzu
      REAL(wp), INTENT(  out), OPTIONAL, DIMENSION(:,:) ::   pCdN, pChN, pCeN, pz0, pu_star, pL, pUN10
      REAL(wp), DIMENSION(:,:), ALLOCATABLE :: zTs, zqs
      ALLOCATE( zTs(SIZE(psst,1),SIZE(psst,2)), zqs(SIZE(psst,1),SIZE(psst,2)) )
      zTs = psst ; zqs = pssq
      CALL ab_turb_generic( 5, 1, zt, zu, zTs, pt_zt, zqs, pq_zt, pU_zu, .FALSE., .FALSE., pCd, pCh, pCe, pt_zu, pq_zu, pUbzu, &
         &                  CdN=pCdN, ChN=pChN, CeN=pCeN, xz0=pz0, xu_star=pu_star, xL=pL, xUN10=pUN10 )
      DEALLOCATE( zTs, zqs )
   END SUBROUTINE TURB_ANDREAS
END MODULE mod_blk_andreas


MODULE mod_blk_neutral_10m
   !! TURB_NEUTRAL_10M( calgo, U_N10, CdN10, ChN10, CeN10, pz0 ) -- reference: src/mod_blk_neutral_10m.f90:33
   USE, INTRINSIC :: ISO_C_BINDING
   USE mod_const, ONLY: wp, nb_iter
   IMPLICIT NONE
   PRIVATE
   PUBLIC :: TURB_NEUTRAL_10M
   INTERFACE
      FUNCTION ab_turb_neutral_10m( algo, niter, U_N10, CdN10, ChN10, CeN10, z0, n, iprecision, mem, stream ) &
         &     BIND(C, NAME='ab_turb_neutral_10m') RESULT(istat)
         IMPORT :: C_INT, C_LONG, C_PTR
         INTEGER(C_INT),  VALUE :: algo, niter, iprecision, mem
         TYPE(C_PTR),     VALUE :: U_N10, CdN10, ChN10, CeN10, z0, stream
         INTEGER(C_LONG), VALUE :: n
         INTEGER(C_INT) :: istat
      END FUNCTION ab_turb_neutral_10m
   END INTERFACE
CONTAINS
   SUBROUTINE TURB_NEUTRAL_10M( calgo, U_N10, CdN10, ChN10, CeN10, pz0 )
      CHARACTER(len=*),         INTENT(in)  :: calgo
      REAL(wp), DIMENSION(:,:), INTENT(in ) :: U_N10
      REAL(wp), DIMENSION(:,:), INTENT(out) :: CdN10, ChN10, CeN10, pz0
      INTEGER :: ialgo
      SELECT CASE( TRIM(calgo) )
      CASE('coare3p0') ; ialgo = 1
      CASE('coare3p6') ; ialgo = 2
      CASE('ncar')     ; ialgo = 3
      CASE('ecmwf')    ; ialgo = 4
      CASE DEFAULT
         PRINT *, 'ERROR: algorithm '//TRIM(calgo)//' is not supported yet!'   ! mod_blk_neutral_10m.f90:195-197 (and :190 for andreas)
         PRINT *, ''
         STOP
      END SELECT
      PRINT *, ' *** Algo = ', TRIM(calgo)
      CALL n10_contig( SIZE(U_N10), U_N10, CdN10, ChN10, CeN10, pz0 )
   CONTAINS
      SUBROUTINE n10_contig( n, a, o1, o2, o3, o4 )
         INTEGER, INTENT(in) :: n
         REAL(wp), DIMENSION(n), INTENT(in),  TARGET :: a
         REAL(wp), DIMENSION(n), INTENT(out), TARGET :: o1, o2, o3, o4
         INTEGER(C_INT) :: istat
         istat = ab_turb_neutral_10m( INT(ialgo,C_INT), INT(nb_iter,C_INT), C_LOC(a), C_LOC(o1), C_LOC(o2), C_LOC(o3), C_LOC(o4), &
            &                         INT(n,C_LONG), 0_C_INT, 0_C_INT, C_NULL_PTR )
         IF( istat /= 0 ) STOP 'TURB_NEUTRAL_10M: engine error'
      END SUBROUTINE n10_contig
   END SUBROUTINE TURB_NEUTRAL_10M
END MODULE mod_blk_neutral_10m
