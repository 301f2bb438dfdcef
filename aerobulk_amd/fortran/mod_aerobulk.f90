! mod_aerobulk.f90 -- Fortran host of the MI355X-native bulk air-sea flux engine.
!
! Drop-in for the public interface of AeroBulk's `mod_aerobulk` (reference: src/mod_aerobulk.f90:20):
!    AEROBULK_INIT, AEROBULK_MODEL, AEROBULK_BYE  -- same names, argument lists, OPTIONALs and kinds
! so that a GCM coupling call site (`USE mod_aerobulk ; CALL aerobulk_model(...)`) recompiles unchanged
! and links libaerobulk_amd.so instead of libaerobulk.a.  Nothing is computed here: every call goes
! through the ISO_C_BINDING interface below to the C ABI of include/aerobulk_amd.h, i.e. to the HIP
! kernels.  The banners and the "message + STOP" error model of the reference are kept on this side.
!
! Build with default reals promoted to 8 bytes, like every build macro of the reference
! (-fdefault-real-8, arch/make.macro_GnuLinux:17):
!    amdflang -O2 -fdefault-real-8 -c mod_aerobulk.f90
!
! `mod_const` (kinds, constants, nb_iter / nitend ...) and `mod_phymbl` (the helper functions) are modules of their own next to this
! file (mod_const.f90, mod_phymbl.f90), like in the reference.

MODULE mod_aerobulk

   USE, INTRINSIC :: ISO_C_BINDING
   USE mod_const, ONLY: wp, nb_iter, nitend, l_use_skin_schemes, ctype_humidity

   IMPLICIT NONE
   PRIVATE

   PUBLIC :: AEROBULK_INIT, AEROBULK_MODEL, AEROBULK_BYE

   !! mirror of `ab_init_report` (include/aerobulk_amd.h)
   TYPE, BIND(C) :: ab_init_report
      INTEGER(C_LONG) :: n_cells, n_masked
      INTEGER(C_INT)  :: hum_type, bad_field
      REAL(C_DOUBLE)  :: bad_min, bad_max, bad_mean
   END TYPE ab_init_report

   INTERFACE
      !! int ab_model(...)  -- AEROBULK_MODEL on the library's process-global session
      FUNCTION ab_model( jt, nt, calgo, calgo_len, zt, zu, sst, t_zt, hum_zt, u_zu, v_zu, slp,   &
         &               ql, qh, tau_x, tau_y, evap, niter, use_skin, rad_sw, rad_lw, t_s,        &
         &               ni, nj, report ) BIND(C, NAME='ab_model') RESULT(istat)
         IMPORT :: C_INT, C_LONG, C_DOUBLE, C_CHAR, C_PTR, ab_init_report
         INTEGER(C_INT),    VALUE :: jt, nt, calgo_len, niter, use_skin
         CHARACTER(KIND=C_CHAR), DIMENSION(*), INTENT(in) :: calgo
         REAL(C_DOUBLE),    VALUE :: zt, zu
         TYPE(C_PTR),       VALUE :: sst, t_zt, hum_zt, u_zu, v_zu, slp
         TYPE(C_PTR),       VALUE :: ql, qh, tau_x, tau_y, evap
         TYPE(C_PTR),       VALUE :: rad_sw, rad_lw, t_s
         INTEGER(C_LONG),   VALUE :: ni, nj
         TYPE(ab_init_report), INTENT(out) :: report
         INTEGER(C_INT) :: istat
      END FUNCTION ab_model
      !! const char* ab_last_error(void)
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

   SUBROUTINE stop_with_library_message()
      !! print the diagnostic the engine recorded, the way ctl_stop does, then STOP
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


   SUBROUTINE print_init_banner( calgo, Ni, Nj, Nt, lskin, rep )
      CHARACTER(len=*),     INTENT(in) :: calgo
      INTEGER,              INTENT(in) :: Ni, Nj, Nt
      LOGICAL,              INTENT(in) :: lskin
      TYPE(ab_init_report), INTENT(in) :: rep
      CHARACTER(len=64) :: chum_ln
      WRITE(6,*)''
      WRITE(6,*)'==================================================================='
      WRITE(6,*)'                   ----- AeroBulk_init -----'
      WRITE(6,*)''
      WRITE(6,*)'    *** Bulk parameterization to be used => "', TRIM(calgo), '"'
      IF( lskin ) THEN
         WRITE(6,*)'       ==> will use the Cool-skin & Warm-layer scheme of `'//TRIM(calgo)//'` !'
      ELSE
         WRITE(6,*)'    *** Cool-skin & Warm-layer schemes will NOT be used!'
      END IF
      WRITE(6,'("     *** Computational domain shape: Ni x Nj = ",i5.5," x ",i5.5)') Ni, Nj
      WRITE(6,*)'    *** Number of time records that will be treated:', Nt
      WRITE(6,*)'    *** Number of iterations in bulk algos: nb_iter  =', INT(nb_iter,1)
      WRITE(6,*)'    *** Filling the `mask` array...'
      IF( rep%n_masked == 0 ) THEN
         WRITE(6,*)'        ==> no points need to be masked! :)'
      ELSE
         WRITE(6,*)'        ==> number of points to mask: ', INT(rep%n_masked), ' (out of ',Ni*Nj,')'
      END IF
      SELECT CASE( rep%hum_type )
      CASE(0)
         chum_ln = 'specific humidity [kg/kg]'
      CASE(2)
         chum_ln = 'relative humidity [%]'
      CASE(1)
         chum_ln = 'dew-point temperature [K]'
      CASE DEFAULT
         chum_ln = 'unknown'
      END SELECT
      WRITE(6,'("     *** Type of prescribed air humidity  `",a,"`")') TRIM(chum_ln)
      WRITE(6,*)'==================================================================='
   END SUBROUTINE print_init_banner


   SUBROUTINE AEROBULK_INIT( Nt, calgo, psst, pta, pha, pU, pV, pslp,  l_use_skin, prsw, prlw )
      !! Same interface as the reference (src/mod_aerobulk.f90:24-46).  AEROBULK_MODEL runs the
      !! initialisation itself at jt==1 (as the reference does); a direct call only validates the
      !! inputs: it computes record 1 into scratch arrays and discards the fluxes.
      INTEGER,                  INTENT(in)  :: Nt
      CHARACTER(len=*),         INTENT(in)  :: calgo
      REAL(wp), DIMENSION(:,:), INTENT(in)  :: psst, pta, pha, pU, pV, pslp
      LOGICAL,                  INTENT(in), OPTIONAL :: l_use_skin
      REAL(wp), DIMENSION(:,:), INTENT(in), OPTIONAL :: prsw, prlw
      REAL(wp), DIMENSION(:,:), ALLOCATABLE :: z1, z2, z3, z4, z5, z6
      INTEGER :: Ni, Nj
      Ni = SIZE(psst,1) ; Nj = SIZE(psst,2)
      ALLOCATE( z1(Ni,Nj), z2(Ni,Nj), z3(Ni,Nj), z4(Ni,Nj), z5(Ni,Nj), z6(Ni,Nj) )
      IF( PRESENT(prsw) .AND. PRESENT(prlw) ) THEN
         CALL AEROBULK_MODEL( 1, Nt, calgo, 2._wp, 10._wp, psst, pta, pha, pU, pV, pslp, z1, z2, z3, z4, z5, &
            &                 l_use_skin=l_use_skin, rad_sw=prsw, rad_lw=prlw, T_s=z6 )
      ELSE
         CALL AEROBULK_MODEL( 1, Nt, calgo, 2._wp, 10._wp, psst, pta, pha, pU, pV, pslp, z1, z2, z3, z4, z5, &
            &                 l_use_skin=l_use_skin )
      END IF
      DEALLOCATE( z1, z2, z3, z4, z5, z6 )
   END SUBROUTINE AEROBULK_INIT


   SUBROUTINE AEROBULK_BYE()
      WRITE(6,*)'==================================================================='
      WRITE(6,*)'                   ----- AeroBulk_bye -----'
      WRITE(6,*)'==================================================================='
      WRITE(6,*)''
   END SUBROUTINE AEROBULK_BYE


   SUBROUTINE AEROBULK_MODEL( jt, Nt, &
      &                       calgo, zt, zu, sst, t_zt,   &
      &                       hum_zt, U_zu, V_zu, slp,    &
      &                       QL, QH, Tau_x, Tau_y, Evap, &
      &                       Niter, l_use_skin, rad_sw, rad_lw, T_s  )
      !! Argument list, kinds, INTENTs and OPTIONALs of the reference (src/mod_aerobulk.f90:176-230).
      INTEGER,                  INTENT(in)  :: jt, Nt
      CHARACTER(len=*),         INTENT(in)  :: calgo
      REAL(wp),                 INTENT(in)  :: zt, zu
      REAL(wp), DIMENSION(:,:), INTENT(in)  :: sst, t_zt, hum_zt, U_zu, V_zu, slp
      REAL(wp), DIMENSION(:,:), INTENT(out) :: QL, QH, Tau_x, Tau_y, Evap
      INTEGER,                  INTENT(in), OPTIONAL :: Niter
      LOGICAL,                  INTENT(in), OPTIONAL :: l_use_skin
      REAL(wp), DIMENSION(:,:), INTENT(in), OPTIONAL :: rad_sw, rad_lw
      REAL(wp), DIMENSION(:,:), INTENT(out),OPTIONAL :: T_s
      !!
      LOGICAL :: lskin, lsrad
      INTEGER :: Ni, Nj
      !!
      IF( PRESENT(Niter) ) nb_iter = Niter   ! sticky module variable of mod_const, like the reference (:236)
      IF( jt == 1 ) nitend = Nt             ! mod_aerobulk.f90:99

      lskin = .FALSE.
      IF( PRESENT(l_use_skin) ) lskin = l_use_skin
      lsrad = ( PRESENT(rad_sw) .AND. PRESENT(rad_lw) )

      Ni = SIZE(sst,1)
      Nj = SIZE(sst,2)
      !! shape agreement (reference: src/mod_aerobulk.f90:87-95)
      IF( ANY(SHAPE(t_zt)  /=(/Ni,Nj/)) ) CALL shape_stop('t_air')
      IF( ANY(SHAPE(hum_zt)/=(/Ni,Nj/)) ) CALL shape_stop('hum_air')
      IF( ANY(SHAPE(U_zu)  /=(/Ni,Nj/)) ) CALL shape_stop('U')
      IF( ANY(SHAPE(V_zu)  /=(/Ni,Nj/)) ) CALL shape_stop('V')
      IF( ANY(SHAPE(slp)   /=(/Ni,Nj/)) ) CALL shape_stop('SLP')
      IF( lsrad ) THEN
         IF( ANY(SHAPE(rad_sw)/=(/Ni,Nj/)) ) CALL shape_stop('Rad_SW')
         IF( ANY(SHAPE(rad_lw)/=(/Ni,Nj/)) ) CALL shape_stop('Rad_LW')
      END IF

      !! explicit-shape dummies below make the compiler hand over contiguous storage (copy-in/out only
      !! for non-contiguous actual arguments)
      IF( lsrad .AND. PRESENT(T_s) ) THEN
         CALL model_contig( Ni*Nj, sst, t_zt, hum_zt, U_zu, V_zu, slp, QL, QH, Tau_x, Tau_y, Evap, rad_sw, rad_lw, T_s )
      ELSEIF( lsrad ) THEN
         CALL model_contig( Ni*Nj, sst, t_zt, hum_zt, U_zu, V_zu, slp, QL, QH, Tau_x, Tau_y, Evap, rad_sw, rad_lw )
      ELSE
         CALL model_contig( Ni*Nj, sst, t_zt, hum_zt, U_zu, V_zu, slp, QL, QH, Tau_x, Tau_y, Evap )
      END IF

      IF( jt==Nt ) CALL AEROBULK_BYE()

   CONTAINS

      SUBROUTINE shape_stop( cname )
         CHARACTER(len=*), INTENT(in) :: cname
         WRITE(6,'(" *** E R R O R :  ")')
         WRITE(6,*) ' AEROBULK_INIT => SST and '//cname//' arrays do not agree in shape!'
         WRITE(6,*) ''
         STOP
      END SUBROUTINE shape_stop

      SUBROUTINE model_contig( n, p1, p2, p3, p4, p5, p6, o1, o2, o3, o4, o5, r1, r2, o6 )
         INTEGER, INTENT(in) :: n
         REAL(wp), DIMENSION(n), INTENT(in),  TARGET :: p1, p2, p3, p4, p5, p6
         REAL(wp), DIMENSION(n), INTENT(out), TARGET :: o1, o2, o3, o4, o5
         REAL(wp), DIMENSION(n), INTENT(in),  TARGET, OPTIONAL :: r1, r2
         REAL(wp), DIMENSION(n), INTENT(out), TARGET, OPTIONAL :: o6
         TYPE(C_PTR) :: cr1, cr2, co6
         TYPE(ab_init_report) :: rep
         INTEGER(C_INT) :: istat
         INTEGER :: k
         CHARACTER(KIND=C_CHAR), DIMENSION(LEN_TRIM(calgo)+1) :: c_algo
         DO k = 1, LEN_TRIM(calgo)
            c_algo(k) = calgo(k:k)
         END DO
         c_algo(LEN_TRIM(calgo)+1) = C_NULL_CHAR
         cr1 = C_NULL_PTR ; cr2 = C_NULL_PTR ; co6 = C_NULL_PTR
         IF( PRESENT(r1) ) cr1 = C_LOC(r1)
         IF( PRESENT(r2) ) cr2 = C_LOC(r2)
         IF( PRESENT(o6) ) co6 = C_LOC(o6)
         istat = ab_model( INT(jt,C_INT), INT(Nt,C_INT), c_algo, INT(LEN_TRIM(calgo),C_INT), REAL(zt,C_DOUBLE), REAL(zu,C_DOUBLE), &
            &              C_LOC(p1), C_LOC(p2), C_LOC(p3), C_LOC(p4), C_LOC(p5), C_LOC(p6),                                    &
            &              C_LOC(o1), C_LOC(o2), C_LOC(o3), C_LOC(o4), C_LOC(o5),                                               &
            &              INT(nb_iter,C_INT), MERGE(1_C_INT, 0_C_INT, lskin), cr1, cr2, co6,                               &
            &              INT(Ni,C_LONG), INT(Nj,C_LONG), rep )
         !! the module variables AEROBULK_INIT sets in the reference (mod_aerobulk.f90:74,127), for callers that read mod_const
         IF( (jt==1) .AND. ((istat==0).OR.(istat==8)) ) THEN
            IF( lskin ) l_use_skin_schemes = .TRUE.        ! (never cleared there either)
            SELECT CASE( rep%hum_type )
            CASE(0)
               ctype_humidity = 'sh'
            CASE(1)
               ctype_humidity = 'dp'
            CASE(2)
               ctype_humidity = 'rh'
            END SELECT
         END IF
         !! banner first (the reference prints it before computing), then a possible STOP
         IF( (jt==1) .AND. ((istat==0).OR.(istat==8)) ) CALL print_init_banner( calgo, Ni, Nj, Nt, lskin, rep )
         IF( istat /= 0 ) CALL stop_with_library_message()
      END SUBROUTINE model_contig

   END SUBROUTINE AEROBULK_MODEL

END MODULE mod_aerobulk
