!
! diaglib (MI355X-native) -- drop-in for the public interface of Molecolab-Pisa/diaglib.
!
! Same module name, same public procedures and argument lists as the reference (reference diaglib.f90:166-167,
! 1483-1539, 171-228, 1855, 558, 1024, 3185, 3481, 3094, 3576, 3052), so an existing CI / augmented-Hessian /
! linear-response caller only re-links.  Everything behind the argument lists is this project's own design:
!
!   * every O(n) operation is a call through ISO_C_BINDING into the HIP engine (include/diaglib_amd.h); the
!     expansion panels live in HBM for the whole solve, only lda x lda matrices visit the host;
!   * the host control flow is written once: a `subspace` object does the block bookkeeping (growth, locking,
!     restart) for all Davidson-type drivers, `davidson_core` serves davidson_driver and gen_david_driver,
!     `lr_core` serves caslr_driver and caslr_eff_driver, and the table / timing printers exist once;
!   * there is no module-level state (the reference keeps LAPACK work arrays and timers in the module,
!     diaglib.f90:155-161): a driver call owns its state on the stack, and the engine context it uses belongs
!     to the calling thread (dla_default_ctx), so two host threads can solve at the same time.
!
! Callbacks keep the reference shape  matvec(n,m,x,ax) / precnd(n,m,fac,x,px)  (reference README.md:34-35).  By
! default they receive HOST arrays (the engine stages blocks through pinned memory); after
! `call diaglib_amd_config(callbacks_on_device=.true.)` they receive DEVICE addresses under the same signature.
!
! Not provided (SURVEY.md 2 row 5): nonsym_driver.
!
module diaglib
  use real_precision
  use iso_c_binding
  implicit none
  private
!
  public :: lobpcg_driver, davidson_driver, gen_david_driver, ortho, b_ortho, ortho_cd, ortho_vs_x, b_ortho_vs_x
  public :: caslr_eff_driver, caslr_driver
  public :: diaglib_amd_config, diaglib_amd_timings
!
  real(dp), parameter :: zero = 0.0_dp, one = 1.0_dp, ten = 10.0_dp
  integer,  parameter :: min_dav = 10          ! smallest basis, in blocks (reference diaglib.f90:1544)
!
! option ids of include/diaglib_amd.h
!
  integer(c_int), parameter :: opt_cb_dev = 1, opt_evec_dev = 2, opt_lr_alg = 7
!
! ---------------------------------------------------------------------------------------
! bookkeeping of a basis that grows block by block (all Davidson-type drivers)
! ---------------------------------------------------------------------------------------
  type :: subspace
    integer  :: blk    = 0     ! block width                          (reference n_max)
    integer  :: want   = 0     ! roots asked for                      (n_targ)
    integer  :: cap    = 0     ! blocks the panels can hold           (dim_dav)
    integer  :: ld     = 0     ! cap * blk, leading dimension of the projected matrices (lda)
    integer  :: nblk   = 0     ! blocks in use                        (m_dim)
    integer  :: cols   = 0     ! columns in use                       (ldu)
    integer  :: head   = 1     ! first column of the newest block     (i_beg)
    integer  :: act    = 0     ! width of the newest block            (n_act)
    integer  :: frozen = 0     ! leading roots that were converged when the newest block was built (n_frozen)
    real(dp) :: tol_rms = zero, tol_max = zero
    logical,        allocatable :: locked(:)     ! (done)
    integer(c_int), allocatable :: mask(:)       ! locked, as the engine wants it
    real(dp),       allocatable :: rnorm(:,:)    ! (1,:) rms, (2,:) max of the residuals
  end type subspace
!
! the reference's four timers, (cpu, wall) pairs (diaglib.f90:160-161)
!
  type :: stopwatch
    real(dp) :: mv(2) = zero, diag(2) = zero, ortho(2) = zero, total(2) = zero, mark(2) = zero
  end type stopwatch
!
! device side of one driver call
!
  type :: solve_env
    type(c_ptr) :: ctx   = c_null_ptr
    type(c_ptr) :: ritz  = c_null_ptr    ! the eigenvector block on the device: the caller's (evec_on_device) or a copy
    logical     :: ritz_is_callers = .false.
    integer     :: rows = 0, width = 0   ! shape of that block
    integer     :: op_cols = 0           ! columns handed to the operator callbacks
    integer     :: restarts = 0
  end type solve_env
!
  interface
    function dla_default_ctx() bind(C,name='dla_default_ctx') result(ctx)
      import :: c_ptr
      type(c_ptr) :: ctx
    end function
    function dla_destroy(ctx) bind(C,name='dla_destroy') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int)     :: st
    end function
    function dla_set_option(ctx,opt,val) bind(C,name='dla_set_option') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int), value :: opt, val
      integer(c_int) :: st
    end function
    function dla_get_option(ctx,opt) bind(C,name='dla_get_option') result(val)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int), value :: opt
      integer(c_int) :: val
    end function
    function dla_last_error(ctx) bind(C,name='dla_last_error') result(msg)
      import :: c_ptr
      type(c_ptr), value :: ctx
      type(c_ptr) :: msg
    end function
    function dla_alloc(ctx,bytes,dev) bind(C,name='dla_alloc') result(st)
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: ctx
      integer(c_size_t), value :: bytes
      type(c_ptr) :: dev
      integer(c_int) :: st
    end function
    function dla_free(ctx,dev) bind(C,name='dla_free') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, dev
      integer(c_int) :: st
    end function
    function dla_trim(ctx,released) bind(C,name='dla_trim') result(st)
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: ctx
      integer(c_size_t) :: released
      integer(c_int) :: st
    end function
    function dla_zero(ctx,dev,bytes) bind(C,name='dla_zero') result(st)
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: ctx, dev
      integer(c_size_t), value :: bytes
      integer(c_int) :: st
    end function
    function dla_upload(ctx,dev,host,bytes) bind(C,name='dla_upload') result(st)
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: ctx, dev, host
      integer(c_size_t), value :: bytes
      integer(c_int) :: st
    end function
    function dla_download(ctx,host,dev,bytes) bind(C,name='dla_download') result(st)
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: ctx, dev, host
      integer(c_size_t), value :: bytes
      integer(c_int) :: st
    end function
    function dla_copy(ctx,dst,src,bytes) bind(C,name='dla_copy') result(st)
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: ctx, dst, src
      integer(c_size_t), value :: bytes
      integer(c_int) :: st
    end function
    function dla_gram(ctx,n,l,x,k,u,c,ldc) bind(C,name='dla_gram') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, x, u
      integer(c_int), value :: n, l, k, ldc
      real(c_double) :: c(*)
      integer(c_int) :: st
    end function
    function dla_gram_lower(ctx,n,l,x,u,c,ldc) bind(C,name='dla_gram_lower') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, x, u
      integer(c_int), value :: n, l, ldc
      real(c_double) :: c(*)
      integer(c_int) :: st
    end function
    function dla_panel_gemm(ctx,n,l,x,k,c,ldc,z) bind(C,name='dla_panel_gemm') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, x, z
      integer(c_int), value :: n, l, k, ldc
      real(c_double) :: c(*)
      integer(c_int) :: st
    end function
    function dla_ritz_residual(ctx,n,l,m,v,av,y,ldy,eig,n_res,skip,evec,r,avy,rnorm) &
             bind(C,name='dla_ritz_residual') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, v, av, evec, r, avy
      integer(c_int), value :: n, l, m, ldy, n_res
      real(c_double) :: y(*), eig(*), rnorm(*)
      integer(c_int) :: skip(*)
      integer(c_int) :: st
    end function
    function dla_ritz_residual_p(ctx,n,l,m,v,av,y,ldy,eig,n_res,skip,evec,r,avy,rnorm,k2,c2,ldc2,p,ap) &
             bind(C,name='dla_ritz_residual_p') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, v, av, evec, r, avy, p, ap
      integer(c_int), value :: n, l, m, ldy, n_res, k2, ldc2
      real(c_double) :: y(*), eig(*), rnorm(*), c2(*)
      integer(c_int) :: skip(*)
      integer(c_int) :: st
    end function
    function dla_ritz_residual2(ctx,n,l,m,v,av,y1,ldy1,y2,ldy2,eig,n_res,skip,e,r,twork,junk,rnorm) &
             bind(C,name='dla_ritz_residual2') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr),    value :: ctx, v, av, e, r, twork, junk
      integer(c_int), value :: n, l, m, ldy1, ldy2, n_res
      real(c_double)        :: y1(*), y2(*), eig(*), rnorm(*)
      integer(c_int)        :: skip(*)
      integer(c_int)        :: st
    end function
    function dla_axpy(ctx,len,alpha,x,y) bind(C,name='dla_axpy') result(st)
      import :: c_ptr, c_int, c_double, c_size_t
      type(c_ptr), value :: ctx, x, y
      integer(c_size_t), value :: len
      real(c_double), value :: alpha
      integer(c_int) :: st
    end function
    function dla_ortho_cd(ctx,n,k,u,growth,ok) bind(C,name='dla_ortho_cd') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, u
      integer(c_int), value :: n, k
      real(c_double) :: growth
      integer(c_int) :: ok
      integer(c_int) :: st
    end function
    function dla_ortho_qr(ctx,n,k,u) bind(C,name='dla_ortho_qr') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, u
      integer(c_int), value :: n, k
      integer(c_int) :: st
    end function
    function dla_ortho_vs_x(ctx,n,m,k,x,u) bind(C,name='dla_ortho_vs_x') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, x, u
      integer(c_int), value :: n, m, k
      integer(c_int) :: st
    end function
    function dla_b_ortho(ctx,n,m,u,bu) bind(C,name='dla_b_ortho') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, u, bu
      integer(c_int), value :: n, m
      integer(c_int) :: st
    end function
    function dla_b_ortho_vs_x(ctx,n,m,k,x,bx,u) bind(C,name='dla_b_ortho_vs_x') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, x, bx, u
      integer(c_int), value :: n, m, k
      integer(c_int) :: st
    end function
    function dla_class_times(ctx,ms) bind(C,name='dla_class_times') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx
      real(c_double) :: ms(*)
      integer(c_int) :: st
    end function
    function dla_reset_stats(ctx) bind(C,name='dla_reset_stats') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int) :: st
    end function
    function dla_begin_solve(ctx) bind(C,name='dla_begin_solve') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int) :: st
    end function
    function dla_check_guess(ctx,n,m,evec) bind(C,name='dla_check_guess') result(st)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, evec
      integer(c_int), value :: n, m
      integer(c_int) :: st
    end function
    function dla_get_coeffs(ctx,len_a,len_u,n_max,n_act,a_red,u_x,u_p) bind(C,name='dla_get_coeffs') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: len_a, len_u, n_max, n_act
      real(c_double) :: a_red(*), u_x(*), u_p(*)
      integer(c_int) :: st
    end function
    function dla_call_matvec(ctx,fn,n,m,x,ax) bind(C,name='dla_call_matvec') result(st)
      import :: c_ptr, c_funptr, c_int
      type(c_ptr), value :: ctx, x, ax
      type(c_funptr), value :: fn
      integer(c_int), value :: n, m
      integer(c_int) :: st
    end function
    function dla_expand_project(ctx,mode,n,m,k,basis,abasis,fn,shift,h,ldh) bind(C,name='dla_expand_project') result(st)
      import :: c_ptr, c_funptr, c_int, c_double
      type(c_ptr), value :: ctx, basis, abasis
      type(c_funptr), value :: fn
      integer(c_int), value :: mode, n, m, k, ldh
      real(c_double), value :: shift
      real(c_double) :: h(*)
      integer(c_int) :: st
    end function
    function dla_pending_block(ctx,m,k,p,ldp,applied) bind(C,name='dla_pending_block') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: m, k, ldp
      real(c_double) :: p(*)
      integer(c_int) :: applied
      integer(c_int) :: st
    end function
    function dla_basis_admit(m,k,p,ldp,applied,hraw,dmat,h,ld) bind(C,name='dla_basis_admit') result(st)
      import :: c_int, c_double
      integer(c_int), value :: m, k, ldp, applied, ld
      real(c_double) :: p(*), hraw(*), dmat(*), h(*)
      integer(c_int) :: st
    end function
    function dla_basis_fold(rows,ncol,dmat,ld,c,ldc) bind(C,name='dla_basis_fold') result(st)
      import :: c_int, c_double
      integer(c_int), value :: rows, ncol, ld, ldc
      real(c_double) :: dmat(*), c(*)
      integer(c_int) :: st
    end function
    function dla_basis_sync(ctx,m,k,dmat,ld) bind(C,name='dla_basis_sync') result(st)
      import :: c_ptr, c_int, c_double
      type(c_ptr),    value :: ctx
      integer(c_int), value :: m, k, ld
      real(c_double) :: dmat(*)
      integer(c_int) :: st
    end function
    function dla_expand_project_metric(ctx,mode,n,m,k,basis,bbasis,abasis,fn,bfn,shift,h,ldh) &
             bind(C,name='dla_expand_project_metric') result(st)
      import :: c_ptr, c_funptr, c_int, c_double
      type(c_ptr),    value :: ctx, basis, bbasis, abasis
      type(c_funptr), value :: fn, bfn
      integer(c_int), value :: mode, n, m, k, ldh
      real(c_double), value :: shift
      real(c_double)        :: h(*)
      integer(c_int)        :: st
    end function
    function dla_call_precnd(ctx,fn,n,m,fac,x,px) bind(C,name='dla_call_precnd') result(st)
      import :: c_ptr, c_funptr, c_int, c_double
      type(c_ptr), value :: ctx, x, px
      type(c_funptr), value :: fn
      integer(c_int), value :: n, m
      real(c_double), value :: fac
      integer(c_int) :: st
    end function
    function dla_call_lrprec(ctx,fn,n,m,fac,xp,xm,yp,ym) bind(C,name='dla_call_lrprec') result(st)
      import :: c_ptr, c_funptr, c_int, c_double
      type(c_ptr), value :: ctx, xp, xm, yp, ym
      type(c_funptr), value :: fn
      integer(c_int), value :: n, m
      real(c_double), value :: fac
      integer(c_int) :: st
    end function
    function dla_potrf_lower(m,a,lda) bind(C,name='dla_potrf_lower') result(info)
      import :: c_int, c_double
      integer(c_int), value :: m, lda
      real(c_double) :: a(*)
      integer(c_int) :: info
    end function
    function dla_trtri_lower(m,a,lda) bind(C,name='dla_trtri_lower') result(info)
      import :: c_int, c_double
      integer(c_int), value :: m, lda
      real(c_double) :: a(*)
      integer(c_int) :: info
    end function
    function dla_syev(uplo,n,a,lda,w) bind(C,name='dla_syev') result(info)
      import :: c_char, c_int, c_double
      character(kind=c_char), value :: uplo
      integer(c_int), value :: n, lda
      real(c_double) :: a(*), w(*)
      integer(c_int) :: info
    end function
    function dla_syev_lowest(uplo,n,a,lda,w,m) bind(C,name='dla_syev_lowest') result(info)
      import :: c_char, c_int, c_double
      character(kind=c_char), value :: uplo
      integer(c_int), value :: n, lda, m
      real(c_double) :: a(*), w(*)
      integer(c_int) :: info
    end function
    subroutine dla_set_solve_info(iters,cols,restarts) bind(C,name='dla_set_solve_info')
      import :: c_int
      integer(c_int), value :: iters, cols, restarts
    end subroutine
    function c_strlen(s) bind(C,name='strlen') result(l)
      import :: c_ptr, c_size_t
      type(c_ptr), value :: s
      integer(c_size_t) :: l
    end function
  end interface
!
contains
!
! ---------------------------------------------------------------------------------------
! configuration (extension; defaults reproduce the reference contract: host callbacks, host eig/evec)
! ---------------------------------------------------------------------------------------
  subroutine diaglib_amd_config(callbacks_on_device, evec_on_device, release_cache, caslr_algorithm, release_context)
    logical, intent(in), optional :: callbacks_on_device, evec_on_device
!   release_cache = .true.: hand the panels the allocator keeps between solves back to the runtime (dla_trim)
    logical, intent(in), optional :: release_cache
!   caslr_algorithm: 0 = the reduced pencil of caslr_driver as one generalised eigenproblem (reference default),
!   1 = the Helmich-Paris route through the singular values of the scaled coupling block (the reference selects it
!   with the harness variable i_alg of its module utils, diaglib.f90:560,675); a setting of the calling thread's context
    integer, intent(in), optional :: caslr_algorithm
!   release_context = .true.: destroy the calling thread's engine context (stream, pinned buffers, cached panels); the next
!   driver call of this thread creates a fresh one.  Contexts of threads that end are released by themselves.
    logical, intent(in), optional :: release_context
    integer(c_size_t) :: released
    type(c_ptr)    :: ctx
    integer(c_int) :: st
    ctx = dla_default_ctx()
    if (present(callbacks_on_device)) st = dla_set_option(ctx, opt_cb_dev, merge(1_c_int,0_c_int,callbacks_on_device))
    if (present(evec_on_device))      st = dla_set_option(ctx, opt_evec_dev, merge(1_c_int,0_c_int,evec_on_device))
    if (present(release_cache)) then
      if (release_cache) st = dla_trim(ctx, released)
    end if
    if (present(caslr_algorithm)) st = dla_set_option(ctx, opt_lr_alg, int(caslr_algorithm,c_int))
    if (present(release_context)) then
      if (release_context) st = dla_destroy(ctx)
    end if
  end subroutine diaglib_amd_config
!
! ---------------------------------------------------------------------------------------
! Device time per kind of work (extension).  The reference prints four buckets -- matvec, diagonalization, orthogonalization, total
! (diaglib.f90:1835-1841) -- and leaves the projection, the Ritz vectors and the residuals un-bucketed; the drivers here print the
! same four (byte-compatible tables).  This routine gives a caller the rest: seconds of HIP-event time on the engine's stream since
! the last reset, per kernel class.
!   call diaglib_amd_timings(profile=.true.)              ! switch the event timing on (about 4 us per launch) and clear the counters
!   call davidson_driver(...)
!   call diaglib_amd_timings(t_proj=tp, t_ritz=tr, ...)    ! read; reset=.true. clears the counters afterwards
! t_proj: every X^T U product (projection :1691, the Gram matrices of the orthogonalisation :3256, :3543); t_update: U -= X C and
! Z = V Y sweeps (:3544); t_trmm: triangular updates (:3327); t_ritz: Ritz vectors + residuals + norms, one fused sweep
! (:1717-1732); t_elem: copies / axpy / fills; t_matvec, t_precnd: the library's OWN device operators (a caller's callbacks are
! timed by the reference's matvec bucket, which the drivers still print).
! ---------------------------------------------------------------------------------------
  subroutine diaglib_amd_timings(profile, reset, t_proj, t_update, t_trmm, t_ritz, t_elem, t_matvec, t_precnd)
    logical,  intent(in),  optional :: profile, reset
    real(dp), intent(out), optional :: t_proj, t_update, t_trmm, t_ritz, t_elem, t_matvec, t_precnd
    type(c_ptr)    :: ctx
    integer(c_int) :: st
    real(c_double) :: ms(8)
    ctx = dla_default_ctx()
    if (present(profile)) then
      st = dla_set_option(ctx, 3_c_int, merge(1_c_int,0_c_int,profile))
      if (profile) st = dla_reset_stats(ctx)
    end if
    ms = 0.0_c_double
    st = dla_class_times(ctx, ms)
    if (present(t_proj))   t_proj   = ms(1)*1.0e-3_dp
    if (present(t_update)) t_update = ms(2)*1.0e-3_dp
    if (present(t_trmm))   t_trmm   = ms(3)*1.0e-3_dp
    if (present(t_ritz))   t_ritz   = ms(4)*1.0e-3_dp
    if (present(t_elem))   t_elem   = ms(5)*1.0e-3_dp
    if (present(t_matvec)) t_matvec = ms(6)*1.0e-3_dp
    if (present(t_precnd)) t_precnd = ms(7)*1.0e-3_dp
    if (present(reset)) then
      if (reset) st = dla_reset_stats(ctx)
    end if
  end subroutine diaglib_amd_timings
!
! ---------------------------------------------------------------------------------------
! small plumbing
! ---------------------------------------------------------------------------------------
!
! address of column j (1-based) of a device panel with leading dimension n
!
  function colp(base,n,j) result(p)
    type(c_ptr), intent(in) :: base
    integer,     intent(in) :: n, j
    type(c_ptr)             :: p
    integer(c_intptr_t)     :: a
    a = transfer(base, a) + 8_c_intptr_t * int(n,c_intptr_t) * int(j-1,c_intptr_t)
    p = transfer(a, p)
  end function colp
!
  function nbytes(n,m) result(b)
    integer, intent(in) :: n, m
    integer(c_size_t)   :: b
    b = 8_c_size_t * int(n,c_size_t) * int(m,c_size_t)
  end function nbytes
!
  function nelem(n,m) result(b)
    integer, intent(in) :: n, m
    integer(c_size_t)   :: b
    b = int(n,c_size_t) * int(m,c_size_t)
  end function nelem
!
! any engine failure is fatal, like the reference's `stop` paths (diaglib.f90:414,3283,3568,3800)
!
  subroutine chk(ctx,st,what)
    type(c_ptr),      intent(in) :: ctx
    integer(c_int),   intent(in) :: st
    character(len=*), intent(in) :: what
    type(c_ptr) :: msg
    character(kind=c_char), pointer :: cm(:)
    integer :: i, ln
    if (st.eq.0) return
    msg = dla_last_error(ctx)
    ln  = int(c_strlen(msg))
    call c_f_pointer(msg, cm, [ln])
    write(6,'(t3,a,a,a,i4)') 'diaglib_amd: ', what, ' failed with status ', st
    write(6,'(t3,200a1)') (cm(i), i = 1, min(ln,200))
!   a non-zero exit status: a plain `stop` would let a solve that died on a GPU error look successful to a launcher
    if (st.eq.5) write(6,'(a)') ' catastrophic failure of ortho_vs_x'
    error stop 1
  end subroutine chk
!
! a device panel of n x m doubles
!
  function dev_panel(ctx,n,m,what) result(p)
    type(c_ptr),      intent(in) :: ctx
    integer,          intent(in) :: n, m
    character(len=*), intent(in) :: what
    type(c_ptr) :: p
    call chk(ctx, dla_alloc(ctx, nbytes(n,m), p), 'allocation of '//what)
  end function dev_panel
!
  subroutine drop_panel(ctx,p)
    type(c_ptr), intent(in)    :: ctx
    type(c_ptr), intent(inout) :: p
    if (c_associated(p)) call chk(ctx, dla_free(ctx, p), 'free')
    p = c_null_ptr
  end subroutine drop_panel
!
! failure of the small symmetric eigensolver: the reference's message and stop (diaglib.f90:412-415)
!
  subroutine need_eigensolver(info)
    integer(c_int), intent(in) :: info
    if (info.eq.0) return
    write(6,'(t3,a,i6)') 'dsyev failed. info = ',info
    error stop 1
  end subroutine need_eigensolver
!
! ---------------------------------------------------------------------------------------
! stopwatch: cpu / wall clock pairs, started and charged explicitly
! ---------------------------------------------------------------------------------------
  subroutine clock_now(t)
    real(dp), intent(out) :: t(2)
    integer(8) :: cnt, rate
    call cpu_time(t(1))
    call system_clock(cnt, rate)
    t(2) = real(cnt,dp)/real(rate,dp)
  end subroutine clock_now
!
  subroutine lap_start(w)
    type(stopwatch), intent(inout) :: w
    call clock_now(w%mark)
  end subroutine lap_start
!
  subroutine lap_charge(w, bucket)
    type(stopwatch), intent(in)    :: w
    real(dp),        intent(inout) :: bucket(2)
    real(dp) :: now(2)
    call clock_now(now)
    bucket = bucket + now - w%mark
  end subroutine lap_charge
!
! ---------------------------------------------------------------------------------------
! device side of a driver call: context, eigenvector block, statistics
! ---------------------------------------------------------------------------------------
  subroutine env_open(e, rows, width, evec)
    type(solve_env),  intent(out) :: e
    integer,          intent(in)  :: rows, width
    real(dp), target, intent(in)  :: evec(rows,width)
    e%ctx   = dla_default_ctx()
    call chk(e%ctx, dla_begin_solve(e%ctx), 'begin_solve')
    e%rows  = rows
    e%width = width
    e%ritz_is_callers = dla_get_option(e%ctx, opt_evec_dev) .ne. 0
    if (e%ritz_is_callers) then
      e%ritz = c_loc(evec)
    else
      e%ritz = dev_panel(e%ctx, rows, width, 'evec')
      call chk(e%ctx, dla_upload(e%ctx, e%ritz, c_loc(evec), nbytes(rows,width)), 'upload of the guess')
    end if
  end subroutine env_open
!
! hand the eigenvector block back (evec holds the current vectors on every exit, like the reference) and report
!
  subroutine env_close(e, evec, iterations)
    type(solve_env),  intent(inout) :: e
    real(dp), target, intent(inout) :: evec(e%rows,e%width)
    integer,          intent(in)    :: iterations
    if (.not.e%ritz_is_callers) then
      call chk(e%ctx, dla_download(e%ctx, c_loc(evec), e%ritz, nbytes(e%rows,e%width)), 'download of evec')
      call drop_panel(e%ctx, e%ritz)
    end if
    call dla_set_solve_info(int(iterations,c_int), int(e%op_cols,c_int), int(e%restarts,c_int))
  end subroutine env_close
!
! ---------------------------------------------------------------------------------------
! subspace bookkeeping
! ---------------------------------------------------------------------------------------
  subroutine sub_setup(s, n_targ, n_max, blocks, tol)
    type(subspace), intent(out) :: s
    integer,        intent(in)  :: n_targ, n_max, blocks
    real(dp),       intent(in)  :: tol
    s%blk  = n_max
    s%want = n_targ
    s%cap  = blocks
    s%ld   = blocks*n_max
    s%tol_rms = tol                  ! reference: rms < tol and max < 10 tol (diaglib.f90:1625-1626, 1741)
    s%tol_max = ten*tol
    allocate (s%locked(n_max), s%mask(n_max), s%rnorm(2,n_max))
    s%locked = .false.
    s%mask   = 0_c_int
    s%rnorm  = zero
    call sub_collapse(s)
  end subroutine sub_setup
!
! back to a single block of full width (start and restart)
!
  subroutine sub_collapse(s)
    type(subspace), intent(inout) :: s
    s%nblk = 1
    s%cols = 0
    s%head = 1
    s%act  = s%blk
  end subroutine sub_collapse
!
! the newest block joins the basis
!
  subroutine sub_admit(s)
    type(subspace), intent(inout) :: s
    s%cols = s%cols + s%act
  end subroutine sub_admit
!
! number of leading wanted roots that are locked
!
  function sub_leading(s) result(k)
    type(subspace), intent(in) :: s
    integer :: k
    k = 0
    do while (k.lt.s%want)
      if (.not.s%locked(k+1)) exit
      k = k + 1
    end do
  end function sub_leading
!
! Locking rule of the reference (diaglib.f90:1737-1746, LOBPCG :446-455): roots lock in order; a root locks when
! both residual measures are under their thresholds and at least one iteration has gone by; the first root that
! does not lock unlocks everything after it.
!
  subroutine sub_lock(s, iteration, upto)
    type(subspace), intent(inout) :: s
    integer,        intent(in)    :: iteration, upto
    integer :: r
    do r = 1, upto
      if (s%locked(r)) cycle
      if (iteration.gt.1 .and. s%rnorm(1,r).lt.s%tol_rms .and. s%rnorm(2,r).lt.s%tol_max) then
        s%locked(r) = .true.
      else
        s%locked(r:s%blk) = .false.
        return
      end if
    end do
  end subroutine sub_lock
!
  function sub_finished(s) result(yes)
    type(subspace), intent(in) :: s
    logical :: yes
    yes = all(s%locked(1:s%want))
  end function sub_finished
!
!
! Can the sweep of this iteration end the solve?  Every wanted root that is still open must come under both thresholds in
! ONE sweep.  A Davidson-Liu sweep shrinks a residual by one to two orders of magnitude; a root whose last residual stood
! more than four orders above its threshold is taken not to make it (when it does, the driver forms the Ritz vectors after
! the sweep instead of inside it).  The first sweep cannot lock anything (:1741).
!
  function sub_may_finish(s, iteration) result(yes)
    type(subspace), intent(in) :: s
    integer,        intent(in) :: iteration
    logical :: yes
    integer :: r
    real(dp), parameter :: reach = 1.0e4_dp
    yes = .false.
    if (iteration.le.1) return
    do r = 1, s%want
      if (s%locked(r)) cycle
      if (s%rnorm(1,r).gt.reach*s%tol_rms .or. s%rnorm(2,r).gt.reach*s%tol_max) return
    end do
    yes = .true.
  end function sub_may_finish
!
  function sub_has_room(s) result(yes)
    type(subspace), intent(in) :: s
    logical :: yes
    yes = s%nblk .lt. s%cap
  end function sub_has_room
!
! open the next block: it gets one column per root that is not locked at the front (diaglib.f90:1765-1785);
! first_root is the first of those roots
!
  subroutine sub_open_block(s, first_root)
    type(subspace), intent(inout) :: s
    integer,        intent(out)   :: first_root
    s%nblk   = s%nblk + 1
    s%head   = s%head + s%act
    s%frozen = sub_leading(s)
    s%act    = s%blk - s%frozen
    first_root = s%frozen + 1
  end subroutine sub_open_block
!
  subroutine sub_refresh_mask(s)
    type(subspace), intent(inout) :: s
    s%mask = merge(1_c_int, 0_c_int, s%locked)
  end subroutine sub_refresh_mask
!
! ---------------------------------------------------------------------------------------
! printers: byte-compatible with the reference's verbose output (formats 1030, 1040, 1050, 1000 of every driver)
! ---------------------------------------------------------------------------------------
  subroutine print_table_head(opening, tol)
    character(len=*), intent(in) :: opening      ! e.g. 'Davidson-Liu iterations (tol='
    real(dp),         intent(in) :: tol
    write(6,'(t5,a,d10.2,a,/,t5,a,/,t7,a,a,/,t5,a)') opening, tol, '):', &
          '------------------------------------------------------------------', &
          '  iter  root              eigenvalue', '         rms         max ok', &
          '------------------------------------------------------------------'
  end subroutine print_table_head
!
  subroutine print_table_rows(s, iteration, values)
    type(subspace), intent(in) :: s
    integer,        intent(in) :: iteration
    real(dp),       intent(in) :: values(:)
    integer :: r
    do r = 1, s%want
      write(6,'(t9,i4,2x,i4,f24.12,2d12.4,l3)') iteration, r, values(r), s%rnorm(:,r), s%locked(r)
    end do
    write(6,*)
  end subroutine print_table_rows
!
  subroutine print_block_report(s)
    type(subspace), intent(in) :: s
    write(6,'(t5,a,/,t7,a,i4,/,t7,a,i4,/,t7,a,i4,/,t5,a)') '----------------------------------------', &
          '# target vectors:    ', s%want, '# new vectors added: ', s%act, '# converged vectors: ', s%frozen, &
          '----------------------------------------'
  end subroutine print_block_report
!
  subroutine print_timings(heading, w)
    character(len=*), intent(in) :: heading      ! e.g. 'timings for davidson (cpu/wall): '
    type(stopwatch),  intent(in) :: w
    write(6,'(t3,a,/,t3,a,2f12.4,/,t3,a,2f12.4,/,t3,a,2f12.4,/,t3,a,a,/,t3,a,2f12.4)') heading, &
          '  matrix-vector multiplications: ', w%mv, '  diagonalization:               ', w%diag, &
          '  orthogonalization:             ', w%ortho, '                                 ', repeat('=',24), &
          '  total:                         ', w%total
  end subroutine print_timings
!
! ---------------------------------------------------------------------------------------
! Davidson-Liu, standard and generalised problem (reference diaglib.f90:1483-1853 and :1855-2250).
!
! One routine serves both public drivers.  With a metric B the basis is kept B-orthonormal, bbasis = B * basis is
! carried along, and the residual of a Ritz pair is A x - theta B x (:2108-2123).
! Deliberate deviation at the restart of the generalised solve (SURVEY 8a A13): the reference zeroes all of its
! bspace right after B-orthonormalising the kept Ritz block (:2196-2200) and never refills it, so its residuals
! after a restart miss theta*B*x of that block; here the kept block's B*x stays in bbasis(:,1:n_max).
! ---------------------------------------------------------------------------------------
  subroutine davidson_core(verbose,n,n_targ,n_max,max_iter,tol,max_dav,shift,op,prec,metric,with_metric,eig,evec,ok)
    logical,          intent(in)    :: verbose, with_metric
    integer,          intent(in)    :: n, n_targ, n_max, max_iter, max_dav
    real(dp),         intent(in)    :: tol, shift
    type(c_funptr),   intent(in)    :: op, prec, metric
    real(dp),         intent(inout) :: eig(n_max)
    real(dp), target, intent(inout) :: evec(n,n_max)
    logical,          intent(inout) :: ok
!
    type(subspace)  :: s
    type(stopwatch) :: w
    type(solve_env) :: e
    type(c_ptr)     :: basis, abasis, bbasis, resid, britz
    real(dp), allocatable :: h(:,:), y(:,:), theta(:), dmat(:,:), hraw(:,:), pblk(:,:)
    logical :: any_pending                  ! some block of the basis is not finished in memory (see admit_pending below)
    real(dp)        :: t_begin(2), t_end(2)
    integer         :: it, sweeps, kept, col, first, j, ritz_cols
    logical         :: patch_kept, projected, mirror_raw
    logical         :: exact_basis          ! the pending blocks are kept on the device as well (dla_expand_project mode 5)
    integer         :: synced               ! columns of dmat the device knows
    integer(c_int)  :: applied
!
    call env_open(e, n, n_max, evec)
    ritz_cols = 0
!
!   the basis holds at least min_dav blocks whatever the caller asks for (reference :1595-1596)
!
    call sub_setup(s, n_targ, n_max, max(min_dav,max_dav), tol)
    basis  = dev_panel(e%ctx, n, s%ld, 'space')
    abasis = dev_panel(e%ctx, n, s%ld, 'aspace')
    resid  = dev_panel(e%ctx, n, n_max, 'r')
    bbasis = c_null_ptr
    britz  = c_null_ptr
    if (with_metric) then
      bbasis = dev_panel(e%ctx, n, s%ld, 'bspace')
      britz  = dev_panel(e%ctx, n, n_max, 'b_evec')
    end if
    allocate (h(s%ld,s%ld), y(s%ld,s%ld), theta(s%ld), dmat(s%ld,s%ld), hraw(s%ld,s%ld), pblk(s%ld,n_max))
!   (the engine decides block by block what it can keep pending: blocks of up to 16 columns while the basis fits its copy of D --
!    BASELINE cfg 4's n_max = 21 gets there once five roots are locked -- everything else is finished in memory, by the device chain
!    while nothing is pending in front of the block and by the host-driven loop with D D^T otherwise)
    exact_basis = .not.with_metric
    call reset_pending()
!
!   The reference zero-fills both n x lda panels (:1632-1633).  On the device no column is read before it has been
!   written (guess copy, operator output, orthogonalisation output), so those two 8*n*lda-byte memsets do not exist;
!   the few columns a restart reads as zeros are zeroed there.
!
    h  = zero
    ok = .false.
    call clock_now(t_begin)
!
!   guess: random if zero, orthonormalised if needed (reference :1644), then the first block of the basis (:1648)
!
    call chk(e%ctx, dla_check_guess(e%ctx, n, n_max, e%ritz), 'check_guess')
    call chk(e%ctx, dla_copy(e%ctx, basis, e%ritz, nbytes(n,n_max)), 'copy')
    if (with_metric) then
      call chk(e%ctx, dla_call_matvec(e%ctx, metric, n, n_max, basis, bbasis), 'bvec')          ! :2033
      call chk(e%ctx, dla_b_ortho(e%ctx, n, n_max, basis, bbasis), 'b_ortho')                   ! :2034
    end if
!
    kept       = 0            ! locked Ritz vectors carried over a restart
    patch_kept = .false.
    projected  = .false.      ! the newest block already has its operator image and its columns of h (dla_expand_project)
    sweeps     = max_iter
    if (verbose) then
      if (with_metric) then
        call print_table_head('Generalized Davidson-Liu iterations (tol=', tol)
      else
        call print_table_head('Davidson-Liu iterations (tol=', tol)
      end if
    end if
!
    do it = 1, max_iter
      call sub_admit(s)
!
!     operator on the new block.  Right after a restart the block starts behind the kept (locked) vectors and runs
!     `kept` zero columns past the Ritz block, exactly like the reference's offsets (:1685, SURVEY App. B 4).
!
      col = s%head + kept
      if (.not.projected) then
        call lap_start(w)
        call chk(e%ctx, dla_call_matvec(e%ctx, op, n, s%act, colp(basis,n,col), colp(abasis,n,col)), 'matvec')
        call lap_charge(w, w%mv)
!
!       new columns of the projected matrix (:1691); the kept roots enter through their Ritz values (:1696-1702)
!
        call chk(e%ctx, dla_gram(e%ctx, n, s%cols, basis, s%act, colp(abasis,n,col), h(1,col), s%ld), 'projection')
      end if
      mirror_raw = .not.projected
      projected = .false.
      e%op_cols = e%op_cols + s%act
      if (patch_kept) then
        do j = 1, kept
          h(j,j) = theta(j)
        end do
        patch_kept = .false.
        kept = 0
      end if
!     (a block that came through the separate calls -- the first one, the one after a restart -- is finished in memory and so is
!      everything in front of it: h is the raw projected matrix, kept for the pending blocks that may follow)
      if (mirror_raw) then
        do j = 1, s%cols
          hraw(1:j,j) = h(1:j,j)
          hraw(j,1:j) = h(1:j,j)
        end do
      end if
!
!     Rayleigh-Ritz: the lowest n_max pairs of the upper triangle (:1703-1708)
!
      do j = 1, s%cols                                  ! (only the upper triangle of the leading block: what the solver reads)
        y(1:j,j) = h(1:j,j)
      end do
      call lap_start(w)
      call need_eigensolver(dla_syev_lowest('u', s%cols, y, s%ld, theta, n_max))
      call lap_charge(w, w%diag)
      eig = theta(1:n_max)
!     (coefficients for the STORED blocks: every product with the panel takes D y)
      if (any_pending) call need_ok(dla_basis_fold(s%cols, n_max, dmat, s%ld, y, s%ld), 'basis fold')
!
!     Ritz vectors, residuals of the wanted roots that are still open, and their norms: one sweep (:1717-1732)
!
      call sub_refresh_mask(s)
      if (with_metric) then
!
!       with a metric the sweep reads B*basis and A*basis (residual A x - theta B x, :2108-2123); the Ritz vectors
!       themselves, basis*y (:2103), are a third panel nobody reads before the solve ends or restarts: they are formed there
!       (ritz_cols remembers the width y belongs to)
!
        call chk(e%ctx, dla_ritz_residual(e%ctx, n, s%cols, n_max, bbasis, abasis, y, s%ld, eig, n_targ, s%mask, &
                                          britz, resid, c_null_ptr, s%rnorm), 'ritz/residual')
        ritz_cols = s%cols
      else
!
!       The Ritz vectors basis*y (:1717) are read by nobody before the solve ends or restarts: the sweep writes them only
!       when that can happen after it -- the last sweep allowed, a full basis, or convergence in sight (sub_may_finish) --
!       and the rare sweep that converges unannounced forms them afterwards.
!
        if (it.eq.max_iter .or. .not.sub_has_room(s) .or. sub_may_finish(s, it)) then
          call chk(e%ctx, dla_ritz_residual(e%ctx, n, s%cols, n_max, basis, abasis, y, s%ld, eig, n_targ, s%mask, &
                                            e%ritz, resid, c_null_ptr, s%rnorm), 'ritz/residual')
          ritz_cols = 0
        else
          call chk(e%ctx, dla_ritz_residual(e%ctx, n, s%cols, n_max, basis, abasis, y, s%ld, eig, n_targ, s%mask, &
                                            c_null_ptr, resid, c_null_ptr, s%rnorm), 'ritz/residual')
          ritz_cols = s%cols
        end if
      end if
!
      call sub_lock(s, it, n_targ)
      if (verbose) call print_table_rows(s, it, eig - shift)     ! the shift is cosmetic here (SURVEY App. B 1)
      if (sub_finished(s)) then
        ok = .true.
        sweeps = it
        exit
      end if
!
      if (sub_has_room(s)) then
!
!       expand with the preconditioned open residuals, orthogonalised against the basis (:1773-1794)
!
        call sub_open_block(s, first)
        call chk(e%ctx, dla_call_precnd(e%ctx, prec, n, s%act, -eig(first), colp(resid,n,first), &
                                        colp(basis,n,s%head)), 'precnd')
        call lap_start(w)
        if (with_metric) then
!
!         with a metric: B-orthogonalisation against the basis (:2170), B on the new block (:2184), b_ortho (:2185) -- and, as for
!         the standard problem, the operator on the finished block and its columns of the projected matrix in the same call
!
          if (it.lt.max_iter) then
            call chk(e%ctx, dla_expand_project_metric(e%ctx, 0_c_int, n, s%cols, s%act, basis, bbasis, abasis, op, metric, zero, &
                                                      h(1,s%head), s%ld), 'b_ortho_vs_x + bvec + b_ortho + matvec + projection')
            projected = .true.
          else
            call chk(e%ctx, dla_b_ortho_vs_x(e%ctx, n, s%cols, s%act, basis, bbasis, colp(basis,n,s%head)), 'b_ortho_vs_x')
            call chk(e%ctx, dla_call_matvec(e%ctx, metric, n, s%act, colp(basis,n,s%head), colp(bbasis,n,s%head)), 'bvec')
            call chk(e%ctx, dla_b_ortho(e%ctx, n, s%act, colp(basis,n,s%head), colp(bbasis,n,s%head)), 'b_ortho')
          end if
          call lap_charge(w, w%ortho)
        else
!
!         standard problem: the new block is orthogonalised against the basis (:1790), and the operator on it (:1685) and
!         its columns of the projected matrix (:1691) -- the head of the next sweep -- follow in the same call, so that
!         the three run back to back on the device (dla_expand_project)
!
          if (it.lt.max_iter) then
!
!           mode 4: a block that the closing pass found orthonormal to 1e-8 keeps its closing projection and its last triangular
!           factor pending (the sweeps of :3543-3544 and :3327 are not run on it): the finished block is [X | U] p for the STORED
!           columns; its columns of the projected matrix come back for the stored block and are corrected here, D^T h_raw D with
!           the upper-triangular D that collects the pending blocks of the whole basis.
!           mode 5: the device holds D as well and projects with X (D D^T) X^T -- what may stay pending (blocks of up to 16
!           columns, up to 320 basis columns) is then bounded by the conditioning of the k x k algebra only; shapes beyond that
!           are finished in memory, exactly against panel*D
!
            if (exact_basis) then
              if (synced.lt.s%cols) then
                call chk(e%ctx, dla_basis_sync(e%ctx, synced, s%cols - synced, dmat, s%ld), 'basis sync')
                synced = s%cols
              end if
              call chk(e%ctx, dla_expand_project(e%ctx, 5_c_int, n, s%cols, s%act, basis, abasis, op, zero, &
                                                 h(1,s%head), s%ld), 'ortho_vs_x + matvec + projection')
            else
!             (with a metric: the block is finished in memory, nothing pending)
              call chk(e%ctx, dla_expand_project(e%ctx, 0_c_int, n, s%cols, s%act, basis, abasis, op, zero, &
                                                 h(1,s%head), s%ld), 'ortho_vs_x + matvec + projection')
            end if
            call chk(e%ctx, dla_pending_block(e%ctx, s%cols, s%act, pblk, s%ld, applied), 'pending block')
            call admit_pending(s%cols, s%act)
            if (exact_basis) then
              call chk(e%ctx, dla_basis_sync(e%ctx, s%cols, s%act, dmat, s%ld), 'basis sync')
              synced = s%cols + s%act
            end if
            projected = .true.
          else
!           (last sweep allowed: nobody will read the operator's image of this block -- the caller's routine is not called)
            call chk(e%ctx, dla_ortho_vs_x(e%ctx, n, s%cols, s%act, basis, colp(basis,n,s%head)), 'ortho_vs_x')
          end if
          call lap_charge(w, w%ortho)
        end if
      else
!
!       basis full: restart from the current Ritz vectors (:1796-1824)
!
        if (verbose) write(6,'(t7,a)') 'Restarting davidson.'
        e%restarts = e%restarts + 1
        if (ritz_cols.gt.0) then
          call chk(e%ctx, dla_panel_gemm(e%ctx, n, ritz_cols, basis, n_max, y, s%ld, e%ritz), 'ritz vectors')
          ritz_cols = 0
        end if
        call chk(e%ctx, dla_copy(e%ctx, basis, e%ritz, nbytes(n,n_max)), 'copy')
        if (with_metric) then
          call chk(e%ctx, dla_copy(e%ctx, bbasis, britz, nbytes(n,n_max)), 'copy')
          call chk(e%ctx, dla_b_ortho(e%ctx, n, n_max, basis, bbasis), 'b_ortho')               ! :2195-2197
        end if
        h = zero
        call reset_pending()
        call sub_collapse(s)
        kept = sub_leading(s)
!
!       what the next sweep reads from the reference's zero-filled panels (:1798,1804): the operator block runs `kept`
!       columns past the Ritz block, and the kept columns of A*basis are zero (their Ritz values go on the diagonal)
!
        if (kept.gt.0) then
          call chk(e%ctx, dla_zero(e%ctx, colp(basis,n,n_max+1), nbytes(n,kept)), 'zero')
          call chk(e%ctx, dla_zero(e%ctx, abasis, nbytes(n,kept)), 'zero')
        end if
        patch_kept = .true.
      end if
      if (verbose) call print_block_report(s)
    end do
!
!   (generalised problem: the Ritz vectors of the last sweep, converged or not)
    if (ritz_cols.gt.0) call chk(e%ctx, dla_panel_gemm(e%ctx, n, ritz_cols, basis, n_max, y, s%ld, e%ritz), 'ritz vectors')
!
    call clock_now(t_end)
    w%total = t_end - t_begin
    if (verbose) call print_timings('timings for davidson (cpu/wall): ', w)
!
    call env_close(e, evec, sweeps)
    call drop_panel(e%ctx, basis)
    call drop_panel(e%ctx, abasis)
    call drop_panel(e%ctx, resid)
    call drop_panel(e%ctx, bbasis)
    call drop_panel(e%ctx, britz)
    deallocate (h, y, theta)
!
  contains
!
!   Blocks the device chain did not finish in memory (dla_expand_project mode 4): block i of the orthonormal basis is
!   [X_stored | U_stored] p_i -- p_i = [E_i ; T_i], the closing projection and the last triangular factor of its orthogonalisation.
!   The orthonormal basis is panel * D with D upper triangular (column block i = p_i, the identity where nothing is pending): the
!   projected matrix is D^T hraw D (hraw: the raw products of the stored columns) and every coefficient block takes D before a
!   product with the panel.  The algebra is host-size (dla_basis_admit / dla_basis_fold).
!
    subroutine reset_pending()
      integer :: j
      dmat = zero
      hraw = zero
      do j = 1, s%ld
        dmat(j,j) = one
      end do
      any_pending = .false.
      synced = 0
      if (exact_basis) call chk(e%ctx, dla_basis_sync(e%ctx, 0_c_int, 0_c_int, dmat, s%ld), 'basis sync')
    end subroutine reset_pending
!
!   a block of k columns has come in behind m stored ones with pblk pending
!
    subroutine admit_pending(m, k)
      integer, intent(in) :: m, k
      integer :: i, j
      integer(c_int) :: st
      logical :: ident
      ident = .true.
      do j = 1, k
        do i = 1, m + k
          if (pblk(i,j).ne.merge(one, zero, i.eq.m+j)) ident = .false.
        end do
      end do
      if (.not.ident) any_pending = .true.
      if (.not.any_pending) then
!       (nothing pending anywhere: h is the projected matrix as it came; the raw copy is kept for later blocks)
        hraw(1:m+k,m+1:m+k) = h(1:m+k,m+1:m+k)
        do j = 1, k
          hraw(m+j,1:m+k) = hraw(1:m+k,m+j)
        end do
        return
      end if
      st = dla_basis_admit(m, k, pblk, s%ld, applied, hraw, dmat, h, s%ld)
      if (st.eq.5_c_int .and. exact_basis) then
!
!       the closing factor of the pending block is not positive definite (nothing bounds the chain's last triangular factor where it
!       ended): the block in memory is what the chain stored, so it is finished there -- orthogonalised against the finished basis
!       panel*D by the host-driven loop, operator and projection repeated on the result (dla_expand_project mode 6) -- and comes
!       in with nothing pending
!
        call chk(e%ctx, dla_expand_project(e%ctx, 6_c_int, n, m, k, basis, abasis, op, zero, h(1,m+1), s%ld), &
                 'ortho_vs_x + matvec + projection (block finished in memory)')
        call chk(e%ctx, dla_pending_block(e%ctx, m, k, pblk, s%ld, applied), 'pending block')
        st = dla_basis_admit(m, k, pblk, s%ld, applied, hraw, dmat, h, s%ld)
      end if
      call need_ok(st, 'basis admit')
    end subroutine admit_pending
!
    subroutine need_ok(st, what)
      integer(c_int),   intent(in) :: st
      character(len=*), intent(in) :: what
      if (st.ne.0) then
        write(6,*) ' diaglib: ', what, ' failed'
        error stop 1
      end if
    end subroutine need_ok
  end subroutine davidson_core
!
  subroutine davidson_driver(verbose,n,n_targ,n_max,max_iter,tol,max_dav,shift,matvec,precnd,eig,evec,ok)
    logical,  intent(in)            :: verbose
    integer,  intent(in)            :: n, n_targ, n_max, max_iter, max_dav
    real(dp), intent(in)            :: tol, shift
    real(dp), intent(inout)         :: eig(n_max)
    real(dp), intent(inout), target :: evec(n,n_max)
    logical,  intent(inout)         :: ok
    external                        :: matvec, precnd
    call davidson_core(verbose, n, n_targ, n_max, max_iter, tol, max_dav, shift, c_funloc(matvec), c_funloc(precnd), &
                       c_null_funptr, .false., eig, evec, ok)
  end subroutine davidson_driver
!
  subroutine gen_david_driver(verbose,n,n_targ,n_max,max_iter,tol,max_dav,shift,matvec,precnd,bvec,eig,evec,ok)
    logical,  intent(in)            :: verbose
    integer,  intent(in)            :: n, n_targ, n_max, max_iter, max_dav
    real(dp), intent(in)            :: tol, shift
    real(dp), intent(inout)         :: eig(n_max)
    real(dp), intent(inout), target :: evec(n,n_max)
    logical,  intent(inout)         :: ok
    external                        :: matvec, precnd, bvec
    call davidson_core(verbose, n, n_targ, n_max, max_iter, tol, max_dav, shift, c_funloc(matvec), c_funloc(precnd), &
                       c_funloc(bvec), .true., eig, evec, ok)
  end subroutine gen_david_driver
!
! ---------------------------------------------------------------------------------------
! LOBPCG (reference diaglib.f90:171-556), standard and generalised (gen_eig) problem.
!
! The reference keeps one basis [X | P | W] per panel and copies the new X (x_new, ax_new, bx_new) and the new P
! (through its evec scratch) into it every iteration (:495-514).  Here every basis panel exists twice: an iteration
! reads the basis from one copy and writes the new X and P blocks straight into the other, then the two swap roles,
! so the four (six) block copies per iteration do not exist.  Only the lower block triangle of S^T A S is formed
! (the eigensolver reads nothing else), and the P coefficients (get_coeffs, :3686-3732) are host-size work.
! ---------------------------------------------------------------------------------------
  subroutine lobpcg_driver(verbose,gen_eig,n,n_targ,n_max,max_iter,tol,shift,matvec,precnd,bvec,eig,evec,ok)
    logical,  intent(in)            :: verbose, gen_eig
    integer,  intent(in)            :: n, n_targ, n_max, max_iter
    real(dp), intent(in)            :: tol, shift
    real(dp), intent(inout)         :: eig(n_max)
    real(dp), intent(inout), target :: evec(n,n_max)
    logical,  intent(inout)         :: ok
    external                        :: matvec, precnd, bvec
!
    type(subspace)  :: s
    type(stopwatch) :: w
    type(solve_env) :: e
    type(c_funptr)  :: op, prec, metric
    type(c_ptr)     :: sp(2), asp(2), bsp(2), resid, latest
    integer         :: rd, wr              ! the copy the basis is read from / the copy that receives the new X and P
    real(dp), allocatable :: h(:,:), theta(:), cx(:,:), cp(:,:), ycp(:,:), seen(:,:,:), pfac(:,:), yf(:,:)
    integer :: tw                  ! > 0: the W block in memory (tw columns) is not finished: the orthonormal block is
                                   ! [X P | W_stored] pfac(1:width,1:tw) (see orthogonalise_w)
    real(dp)        :: t_begin(2), t_end(2)
    integer         :: it, sweeps, width, live, c_x, c_p, c_w, wide, bet
    logical         :: projected           ! the W block already has its operator image and S^T A S is in h (dla_expand_project)
    integer(c_int)  :: applied_w
!
    op     = c_funloc(matvec)
    prec   = c_funloc(precnd)
    metric = c_funloc(bvec)
    call env_open(e, n, n_max, evec)
    call sub_setup(s, n_targ, n_max, 3, tol)
    wide = s%ld
    bsp  = c_null_ptr
    do rd = 1, 2
      sp(rd)  = dev_panel(e%ctx, n, wide, 'space')
      asp(rd) = dev_panel(e%ctx, n, wide, 'aspace')
!     (the reference allocates bspace / bx_new for the standard problem too, :259,270; here only when used)
      if (gen_eig) bsp(rd) = dev_panel(e%ctx, n, wide, 'bspace')
    end do
    resid = dev_panel(e%ctx, n, n_max, 'r')
    rd = 1
    wr = 2
    allocate (h(wide,wide), theta(wide), seen(2,n_max,2), pfac(wide,n_max))
    tw = 0
    h    = zero
    seen = zero
    ok   = .false.
    projected = .false.
    call clock_now(t_begin)
!
!   guess (:295), for the generalised problem made B-orthonormal (:299-302)
!
    call chk(e%ctx, dla_check_guess(e%ctx, n, n_max, e%ritz), 'check_guess')
    if (gen_eig) then
      call chk(e%ctx, dla_call_matvec(e%ctx, metric, n, n_max, e%ritz, bsp(rd)), 'bvec')
      call chk(e%ctx, dla_b_ortho(e%ctx, n, n_max, e%ritz, bsp(rd)), 'b_ortho')
    end if
!
!   Rayleigh-Ritz on the guess alone (:306-325): X, A X [, B X] of the first basis go to the other copy
!
    call chk(e%ctx, dla_copy(e%ctx, sp(rd), e%ritz, nbytes(n,n_max)), 'copy')
    call apply_operator(1, n_max)
    call chk(e%ctx, dla_gram(e%ctx, n, n_max, sp(rd), n_max, asp(rd), h, wide), 'projection')
    call lap_start(w)
    call need_eigensolver(dla_syev_lowest('l', n_max, h, wide, theta, n_max))
    call lap_charge(w, w%diag)
    eig = theta(1:n_max)
    s%mask = 0_c_int
    call ritz_step(n_max)
    call turn_over()
!
!   first W block: preconditioned residuals orthogonalised against X (:350-367)
!
    c_x = 1
    c_w = 1 + n_max
    call chk(e%ctx, dla_call_precnd(e%ctx, prec, n, n_max, shift-eig(c_x), colp(resid,n,c_x), colp(sp(rd),n,c_w)), 'precnd')
    it = 0
    call orthogonalise_w(n_max, n_max)
!
    live   = n_max
    sweeps = max_iter
    if (verbose) call print_table_head('LOBPCG iterations (tol=', tol)
!
    do it = 1, max_iter
!
!     A on the W block (:394-397), then all of S^T A S for S = [X | P | W] (:401-403); no P block in the first sweep
!
      width = n_max + 2*live
      if (it.eq.1) width = 2*n_max
      if (projected) then
        e%op_cols = e%op_cols + live          ! (orthogonalise_w has applied the operator and projected already)
      else
        call apply_operator(c_w, live)
        call chk(e%ctx, dla_gram_lower(e%ctx, n, width, sp(rd), asp(rd), h, wide), 'projection')
      end if
      projected = .false.
      call lap_start(w)
      call need_eigensolver(dla_syev_lowest('l', width, h, wide, theta, n_max))
      call lap_charge(w, w%diag)
      eig = theta(1:n_max)
!
!     new X, A X [, B X], residuals and norms in one sweep (:420-442).  Standard problem: the P block of this sweep
!     (P = S cp, A P = AS cp, :485-503) reads the same two panels, so the sweep forms it as well -- for the number of open
!     roots the previous sweep left (`bet`); the coefficients only need the eigenvectors in h.  When a root locks in this
!     sweep the bet is lost and the block is formed again below, as before.
!
      call sub_refresh_mask(s)
      bet = open_roots_expected()
      allocate (cx(width,n_max), cp(width,bet))
      call chk(e%ctx, dla_get_coeffs(e%ctx, wide, width, n_max, bet, h, cx, cp), 'get_coeffs')
      if (.not.gen_eig) then
        if (tw.gt.0) then
!
!         the W block in memory lacks its closing steps (orthogonalise_w): products with the panel take D = [I E ; 0 T] on
!         their coefficients
!
          allocate (yf(width,n_max))
          yf = h(1:width,1:n_max)
          call fold_pending(yf, n_max)
          call fold_pending(cp, bet)
          call chk(e%ctx, dla_ritz_residual_p(e%ctx, n, width, n_max, sp(rd), asp(rd), yf, width, eig, n_max, s%mask, &
                                              sp(wr), resid, asp(wr), s%rnorm, bet, cp, width, &
                                              colp(sp(wr),n,n_max+1), colp(asp(wr),n,n_max+1)), 'ritz/residual + p block')
          deallocate (yf)
        else
          call chk(e%ctx, dla_ritz_residual_p(e%ctx, n, width, n_max, sp(rd), asp(rd), h, wide, eig, n_max, s%mask, &
                                              sp(wr), resid, asp(wr), s%rnorm, bet, cp, width, &
                                              colp(sp(wr),n,n_max+1), colp(asp(wr),n,n_max+1)), 'ritz/residual + p block')
        end if
      else
!
!       with a metric: the sweep over B S and A S gives B X, A X, the residual A X - eig B X and the B P, A P blocks; X and P
!       themselves come from ONE product S [y | cp] -- every basis panel is read once per iteration (the separate products of
!       :420-424 and :495-503 read each of them twice)
!
        call chk(e%ctx, dla_ritz_residual_p(e%ctx, n, width, n_max, bsp(rd), asp(rd), h, wide, eig, n_max, s%mask, &
                                            bsp(wr), resid, asp(wr), s%rnorm, bet, cp, width, &
                                            colp(bsp(wr),n,n_max+1), colp(asp(wr),n,n_max+1)), 'ritz/residual + p block')
        allocate (ycp(width,n_max+bet))
        ycp(:,1:n_max)  = h(1:width,1:n_max)
        ycp(:,n_max+1:) = cp
        call chk(e%ctx, dla_panel_gemm(e%ctx, n, width, sp(rd), n_max+bet, ycp, width, sp(wr)), 'ritz vectors + p block')
        deallocate (ycp)
      end if
      latest = sp(wr)
      deallocate (cx, cp)
      seen(:,:,1) = seen(:,:,2)
      seen(:,:,2) = s%rnorm(1:2,1:n_max)
!
      call sub_lock(s, it, n_max)                                 ! LOBPCG scans all n_max roots (:446-455)
      if (verbose) call print_table_rows(s, it, eig - shift)      ! the returned eig keeps the shift (:416,461)
      if (sub_finished(s)) then
        ok = .true.
        sweeps = it
        exit
      end if
!
      live = n_max - count(s%locked)
      c_x  = n_max - live + 1
      c_p  = c_x + live
      c_w  = c_p + live
!
!     coefficients of the new P block (:485-488); P = S cp, A P = AS cp [, B P = BS cp] (:495-503) go straight into
!     the P block of the other copy, whose X block already holds the new Ritz vectors (:510-514)
!
      if (bet.ne.live) then
        allocate (cx(width,n_max), cp(width,max(live,1)))
        call chk(e%ctx, dla_get_coeffs(e%ctx, wide, width, n_max, live, h, cx, cp), 'get_coeffs')
        if (tw.gt.0) call fold_pending(cp, live)
        call chk(e%ctx, dla_panel_gemm(e%ctx, n, width, sp(rd),  live, cp, width, colp(sp(wr),n,c_p)), 'p block')
        call chk(e%ctx, dla_panel_gemm(e%ctx, n, width, asp(rd), live, cp, width, colp(asp(wr),n,c_p)), 'ap block')
        if (gen_eig) call chk(e%ctx, dla_panel_gemm(e%ctx, n, width, bsp(rd), live, cp, width, colp(bsp(wr),n,c_p)), 'bp block')
        deallocate (cx, cp)
      end if
      call turn_over()
!
!     new W block: preconditioned open residuals, orthogonalised against [X | P] (:518-528)
!
      call chk(e%ctx, dla_call_precnd(e%ctx, prec, n, live, shift-eig(1), colp(resid,n,c_x), colp(sp(rd),n,c_w)), 'precnd')
      call orthogonalise_w(n_max+live, live)
    end do
!
    call clock_now(t_end)
    w%total = t_end - t_begin
!
!   the current Ritz vectors go back in evec (the reference does this on convergence, :466; on a non-converged
!   exit it leaves its P-block scratch there -- here evec holds the Ritz vectors in both cases)
!
    call chk(e%ctx, dla_copy(e%ctx, e%ritz, latest, nbytes(n,n_max)), 'copy')
    if (verbose) call print_timings('timings for lobpcg (cpu/wall):   ', w)
    call env_close(e, evec, sweeps)
    do rd = 1, 2
      call drop_panel(e%ctx, sp(rd))
      call drop_panel(e%ctx, asp(rd))
      call drop_panel(e%ctx, bsp(rd))
    end do
    call drop_panel(e%ctx, resid)
    deallocate (h, theta)
!
  contains
!
!   A (+ shift) on `cols` columns of the basis starting at column c0 (:306-312, 394-397)
!
    subroutine apply_operator(c0, cols)
      integer, intent(in) :: c0, cols
      call lap_start(w)
      call chk(e%ctx, dla_call_matvec(e%ctx, op, n, cols, colp(sp(rd),n,c0), colp(asp(rd),n,c0)), 'matvec')
      call lap_charge(w, w%mv)
      e%op_cols = e%op_cols + cols
      if (shift.ne.zero) call chk(e%ctx, dla_axpy(e%ctx, nelem(n,cols), shift, colp(sp(rd),n,c0), colp(asp(rd),n,c0)), 'axpy')
    end subroutine apply_operator
!
!   X = S y, A X = AS y, r = A X - eig X (B X instead of X with a metric) and the residual norms; the new blocks are
!   the X block of the copy the next basis will be read from
!
    subroutine ritz_step(cols)
      integer, intent(in) :: cols
      if (gen_eig) then
        call chk(e%ctx, dla_ritz_residual(e%ctx, n, cols, n_max, bsp(rd), asp(rd), h, wide, eig, n_max, s%mask, &
                                          bsp(wr), resid, asp(wr), s%rnorm), 'ritz/residual')
        call chk(e%ctx, dla_panel_gemm(e%ctx, n, cols, sp(rd), n_max, h, wide, sp(wr)), 'ritz vectors')
      else
        call chk(e%ctx, dla_ritz_residual(e%ctx, n, cols, n_max, sp(rd), asp(rd), h, wide, eig, n_max, s%mask, &
                                          sp(wr), resid, asp(wr), s%rnorm), 'ritz/residual')
      end if
      latest = sp(wr)
    end subroutine ritz_step
!
!   how many roots will still be open after the sweep that is about to run: the locking rule (sub_lock) applied to the
!   residual norms extrapolated from the last two sweeps (each root's norms shrink by about the same factor per sweep).
!   Only a bet -- the sweep forms the P block for this many roots, and the driver re-forms it when the bet is lost.
!
    function open_roots_expected() result(open)
      integer  :: open, r, front
      real(dp) :: guess(2), ratio
      front = 0
      do r = 1, n_max
        if (.not.s%locked(r)) exit
        front = front + 1
      end do
      if (it.gt.2) then
        do r = front + 1, n_max
          guess = seen(:,r,2)
          if (seen(1,r,1).gt.zero .and. seen(2,r,1).gt.zero) then
            ratio = min(one, seen(1,r,2)/seen(1,r,1))
            guess(1) = seen(1,r,2)*ratio
            ratio = min(one, seen(2,r,2)/seen(2,r,1))
            guess(2) = seen(2,r,2)*ratio
          end if
          if (guess(1).lt.s%tol_rms .and. guess(2).lt.s%tol_max) then
            front = front + 1
          else
            exit
          end if
        end do
      end if
      open = max(1, n_max - front)
    end function open_roots_expected
!
!   cf <- D cf with D = [I E ; 0 T]: the W rows (the last tw of the current width) of a coefficient block reach every row through
!   the pending block pfac = [E ; T]
!
    subroutine fold_pending(cf, ncol)
      integer,  intent(in)    :: ncol
      real(dp), intent(inout) :: cf(width,ncol)
      real(dp) :: wrows(tw,ncol)
      integer  :: mx
      if (tw.le.0 .or. ncol.le.0) return
      mx = width - tw
      wrows = cf(mx+1:width,1:ncol)
      cf(mx+1:width,1:ncol) = matmul(pfac(mx+1:width,1:tw), wrows)
      if (mx.gt.0) cf(1:mx,1:ncol) = cf(1:mx,1:ncol) + matmul(pfac(1:mx,1:tw), wrows)
    end subroutine fold_pending
!
    subroutine turn_over()
      integer :: keep
      keep = rd
      rd = wr
      wr = keep
    end subroutine turn_over
!
!   the W block (k columns behind m basis columns) against the basis, in the metric if there is one (:358-366, 523-529)
!
    subroutine orthogonalise_w(m, k)
      integer, intent(in) :: m, k
      tw = 0
      call lap_start(w)
      if (gen_eig) then
        if (it.lt.max_iter) then
          call chk(e%ctx, dla_expand_project_metric(e%ctx, 1_c_int, n, m, k, sp(rd), bsp(rd), asp(rd), op, metric, shift, h, wide), &
                   'b_ortho_vs_x + bvec + b_ortho + matvec + projection')
          projected = .true.
        else
          call chk(e%ctx, dla_b_ortho_vs_x(e%ctx, n, m, k, sp(rd), bsp(rd), colp(sp(rd),n,m+1)), 'b_ortho_vs_x')
          call chk(e%ctx, dla_call_matvec(e%ctx, metric, n, k, colp(sp(rd),n,m+1), colp(bsp(rd),n,m+1)), 'bvec')
          call chk(e%ctx, dla_b_ortho(e%ctx, n, k, colp(sp(rd),n,m+1), colp(bsp(rd),n,m+1)), 'b_ortho')
        end if
      else
!       (the operator on the W block and S^T A S -- the head of the next sweep -- in the same call: dla_expand_project)
        if (it.lt.max_iter) then
!
!         mode 3: W is used by one sweep and then rebuilt, so the closing projection and the last triangular factor of its
!         orthogonalisation are not applied to it -- the projection comes back corrected, and the sweep's coefficients take the
!         pending block (fold_pending)
!
          call chk(e%ctx, dla_expand_project(e%ctx, 3_c_int, n, m, k, sp(rd), asp(rd), op, shift, h, wide), &
                   'ortho_vs_x + matvec + projection')
          call chk(e%ctx, dla_pending_block(e%ctx, m, k, pfac, wide, applied_w), 'pending block')
          tw = k
          projected = .true.
        else
!         (last sweep allowed: the operator's image of this block would never be read)
          call chk(e%ctx, dla_ortho_vs_x(e%ctx, n, m, k, sp(rd), colp(sp(rd),n,m+1)), 'ortho_vs_x')
        end if
      end if
      call lap_charge(w, w%ortho)
    end subroutine orthogonalise_w
  end subroutine lobpcg_driver
!
! ---------------------------------------------------------------------------------------
! public orthogonalisation routines on HOST arrays (reference signatures); each uploads, runs the device path and
! downloads.  The drivers never use these wrappers.
! ---------------------------------------------------------------------------------------
  subroutine ortho_cd(n,m,u,growth,ok)
    integer,  intent(in)            :: n, m
    real(dp), intent(inout), target :: u(n,m)
    real(dp), intent(inout)         :: growth
    logical,  intent(inout)         :: ok
    type(c_ptr)    :: ctx, ud
    integer(c_int) :: flag
    ctx = dla_default_ctx()
    ud  = dev_panel(ctx, n, m, 'u')
    call chk(ctx, dla_upload(ctx, ud, c_loc(u), nbytes(n,m)), 'upload')
    call chk(ctx, dla_ortho_cd(ctx, n, m, ud, growth, flag), 'ortho_cd')
    ok = flag .ne. 0
    call chk(ctx, dla_download(ctx, c_loc(u), ud, nbytes(n,m)), 'download')
    call drop_panel(ctx, ud)
  end subroutine ortho_cd
!
  subroutine ortho_vs_x(n,m,k,x,u,ax,au)
    integer,  intent(in)            :: n, m, k
    real(dp), intent(in),    target :: x(n,m)
    real(dp), intent(inout), target :: u(n,k)
    real(dp)                        :: ax(*), au(*)      ! dead in the reference too (SURVEY App. B 5)
    type(c_ptr) :: ctx, xd, ud
    ctx = dla_default_ctx()
    xd  = dev_panel(ctx, n, max(m,1), 'x')
    ud  = dev_panel(ctx, n, k, 'u')
    if (m.gt.0) call chk(ctx, dla_upload(ctx, xd, c_loc(x), nbytes(n,m)), 'upload')
    call chk(ctx, dla_upload(ctx, ud, c_loc(u), nbytes(n,k)), 'upload')
    call chk(ctx, dla_ortho_vs_x(ctx, n, m, k, xd, ud), 'ortho_vs_x')
    call chk(ctx, dla_download(ctx, c_loc(u), ud, nbytes(n,k)), 'download')
    call drop_panel(ctx, xd)
    call drop_panel(ctx, ud)
  end subroutine ortho_vs_x
!
! Householder-QR orthonormalisation U <- U R^-1 (reference diaglib.f90:3052-3092: dgeqrf on a copy, dtrsm with its R).
! R carries the signs LAPACK's reflectors give its diagonal, so columns of the result may be the negatives of what a
! Cholesky-QR returns; dla_ortho_qr reproduces that convention.  The second argument is never touched (nor in the
! reference).
!
  subroutine ortho(n,m,u,w)
    integer,  intent(in)            :: n, m
    real(dp), intent(inout), target :: u(n,m)
    real(dp)                        :: w(*)
    type(c_ptr) :: ctx, ud
    ctx = dla_default_ctx()
    ud  = dev_panel(ctx, n, m, 'u')
    call chk(ctx, dla_upload(ctx, ud, c_loc(u), nbytes(n,m)), 'upload')
    call chk(ctx, dla_ortho_qr(ctx, n, m, ud), 'ortho')
    call chk(ctx, dla_download(ctx, c_loc(u), ud, nbytes(n,m)), 'download')
    call drop_panel(ctx, ud)
  end subroutine ortho
!
  subroutine b_ortho(n,m,u,bu)
    integer,  intent(in)            :: n, m
    real(dp), intent(inout), target :: u(n,m), bu(n,m)
    type(c_ptr) :: ctx, ud, bd
    ctx = dla_default_ctx()
    ud  = dev_panel(ctx, n, m, 'u')
    bd  = dev_panel(ctx, n, m, 'bu')
    call chk(ctx, dla_upload(ctx, ud, c_loc(u), nbytes(n,m)), 'upload')
    call chk(ctx, dla_upload(ctx, bd, c_loc(bu), nbytes(n,m)), 'upload')
    call chk(ctx, dla_b_ortho(ctx, n, m, ud, bd), 'b_ortho')
    call chk(ctx, dla_download(ctx, c_loc(u), ud, nbytes(n,m)), 'download')
    call chk(ctx, dla_download(ctx, c_loc(bu), bd, nbytes(n,m)), 'download')
    call drop_panel(ctx, ud)
    call drop_panel(ctx, bd)
  end subroutine b_ortho
!
  subroutine b_ortho_vs_x(n,m,k,x,bx,u)
    integer,  intent(in)            :: n, m, k
    real(dp), intent(in),    target :: x(n,m), bx(n,m)
    real(dp), intent(inout), target :: u(n,k)
    type(c_ptr) :: ctx, xd, bd, ud
    ctx = dla_default_ctx()
    xd  = dev_panel(ctx, n, max(m,1), 'x')
    bd  = dev_panel(ctx, n, max(m,1), 'bx')
    ud  = dev_panel(ctx, n, k, 'u')
    if (m.gt.0) then
      call chk(ctx, dla_upload(ctx, xd, c_loc(x), nbytes(n,m)), 'upload')
      call chk(ctx, dla_upload(ctx, bd, c_loc(bx), nbytes(n,m)), 'upload')
    end if
    call chk(ctx, dla_upload(ctx, ud, c_loc(u), nbytes(n,k)), 'upload')
    call chk(ctx, dla_b_ortho_vs_x(ctx, n, m, k, xd, bd, ud), 'b_ortho_vs_x')
    call chk(ctx, dla_download(ctx, c_loc(u), ud, nbytes(n,k)), 'download')
    call drop_panel(ctx, xd)
    call drop_panel(ctx, bd)
    call drop_panel(ctx, ud)
  end subroutine b_ortho_vs_x
!
! ---------------------------------------------------------------------------------------
! Linear-response generalised eigenproblem
!
!   / A  B \ / Y \     /  S  D \ / Y \
!   |      | |   | = w |       | |   |        (reference diaglib.f90:558-1022 caslr_driver, :1024-1481 caslr_eff_driver)
!   \ B  A / \ Z /     \ -D -S / \ Z /
!
! solved in the combinations b+ = Y + Z, b- = Y - Z with two expansion spaces vp, vm.  One routine serves both public
! drivers; they differ in what is kept orthonormal and in the reduced problem:
!
!   efficient (caslr_eff_driver): vp is (A+B)-orthonormal, vm is (A-B)-orthonormal; with s = vm^T (S+D) vp the
!     reduced problem is  s^T s u+ = (1/w)^2 u+,  u- = w s u+  (:1052-1060);
!   traditional (caslr_driver): vp, vm are Euclidean-orthonormal; with E+ = vp^T (A+B) vp, E- = vm^T (A-B) vm the
!     reduced problem is the 2 ldu-dimensional pencil  (0 s^T; s 0) u = (1/w) diag(E+,E-) u  (dsygv, :783).
!
! Everything lives on the device: vp, vm, their images under the four operators, the residual blocks.
! Own operation order, none of which changes a result beyond rounding: the reduced matrices grow by their new block
! rows / columns instead of being recomputed from the full panels every iteration (:754-756, 1289), and the Ritz
! vectors (:862-868, 1324-1333) are formed when they are needed (convergence, restart, exit).
! `ok` reports convergence (the reference's caslr_driver hands its `ok` to ortho_cd, :715-716, and so returns .true.
! even when max_iter is exhausted).
! ---------------------------------------------------------------------------------------
  subroutine lr_core(traditional,verbose,n,n2,n_targ,n_max,max_iter,tol,max_dav,apb,amb,spd,smd,prec,eig,evec,ok)
    logical,          intent(in)    :: traditional, verbose
    integer,          intent(in)    :: n, n2, n_targ, n_max, max_iter, max_dav
    real(dp),         intent(in)    :: tol
    type(c_funptr),   intent(in)    :: apb, amb, spd, smd, prec
    real(dp),         intent(inout) :: eig(n_max)
    real(dp), target, intent(inout) :: evec(n2,n_max)
    logical,          intent(inout) :: ok
!
    type(subspace)  :: s
    type(stopwatch) :: w
    type(solve_env) :: e
    type(c_ptr)     :: vp, vm, lvp, lvm, bvp, bvm, rp, rm, bp, bm, tp, tm
    real(dp), allocatable :: smat(:,:), epmat(:,:), emmat(:,:), sts(:,:), lam(:), up(:,:), um(:,:), eye(:,:)
    real(dp), allocatable :: np(:,:), nm(:,:)
    real(dp)        :: t_begin(2), t_end(2), sqrt2, growth, nohmat(1)
    integer         :: it, sweeps, first, r, c, lo
    integer(c_int)  :: flag
    logical         :: vectors_current
!
    call env_open(e, n2, n_max, evec)
    call sub_setup(s, n_targ, n_max, max(min_dav,max_dav), tol)
    sqrt2 = sqrt(2.0_dp)
!
    vp  = dev_panel(e%ctx, n, s%ld, 'vp')
    vm  = dev_panel(e%ctx, n, s%ld, 'vm')
    lvp = dev_panel(e%ctx, n, s%ld, 'lvp')         ! (A+B) vp
    lvm = dev_panel(e%ctx, n, s%ld, 'lvm')         ! (A-B) vm
    bvp = dev_panel(e%ctx, n, s%ld, 'bvp')         ! (S-D) vm
    bvm = dev_panel(e%ctx, n, s%ld, 'bvm')         ! (S+D) vp
    rp  = dev_panel(e%ctx, n, n_max, 'rp')
    rm  = dev_panel(e%ctx, n, n_max, 'rm')
    bp  = dev_panel(e%ctx, n, n_max, 'bp')
    bm  = dev_panel(e%ctx, n, n_max, 'bm')
    tp  = dev_panel(e%ctx, n, n_max, 'tp')
    tm  = dev_panel(e%ctx, n, n_max, 'tm')
    allocate (smat(s%ld,s%ld), up(s%ld,n_max), um(s%ld,n_max), eye(n_max,n_max), np(2,n_max), nm(2,n_max))
    if (traditional) then
      allocate (epmat(s%ld,s%ld), emmat(s%ld,s%ld))
      epmat = zero
      emmat = zero
    else
      allocate (sts(s%ld,s%ld), lam(s%ld))
    end if
    smat = zero
    np   = zero
    nm   = zero
    eye  = zero
    do r = 1, n_max
      eye(r,r) = one
    end do
    ok = .false.
    call clock_now(t_begin)
!
!   the guess in the plus / minus combinations, orthonormal in the sense of the variant (:709-716, 1249-1259)
!
    call lr_split(e%ctx, n, n2, n_max, e%ritz, vp, vm)
    call condition_first_block()
    vectors_current = .false.
    sweeps = max_iter
    if (verbose) call print_table_head('Davidson-Liu iterations (tol=', tol)
!
    do it = 1, max_iter
      call sub_admit(s)
      vectors_current = .false.
      c = s%head
!
!     operator products on the new blocks (:743-746, 1281-1282); the efficient variant already has (A+B) vp and
!     (A-B) vm from making the blocks orthonormal
!
      call lap_start(w)
      if (traditional) then
        call chk(e%ctx, dla_call_matvec(e%ctx, apb, n, s%act, colp(vp,n,c), colp(lvp,n,c)), 'apbmul')
        call chk(e%ctx, dla_call_matvec(e%ctx, amb, n, s%act, colp(vm,n,c), colp(lvm,n,c)), 'ambmul')
        e%op_cols = e%op_cols + 2*s%act
      end if
      call chk(e%ctx, dla_call_matvec(e%ctx, spd, n, s%act, colp(vp,n,c), colp(bvm,n,c)), 'spdmul')
      call chk(e%ctx, dla_call_matvec(e%ctx, smd, n, s%act, colp(vm,n,c), colp(bvp,n,c)), 'smdmul')
      e%op_cols = e%op_cols + 2*s%act
      call lap_charge(w, w%mv)
!
!     reduced matrices: new block columns, and for the non-symmetric s = vm^T (S+D) vp also the new block row
!
      lo = c - 1
      call chk(e%ctx, dla_gram(e%ctx, n, s%cols, vm, s%act, colp(bvm,n,c), smat(1,c), s%ld), 'reduced matrix')
      if (lo.gt.0) call chk(e%ctx, dla_gram(e%ctx, n, s%act, colp(vm,n,c), lo, bvm, smat(c,1), s%ld), 'reduced matrix')
      if (traditional) then
        call chk(e%ctx, dla_gram(e%ctx, n, s%cols, vp, s%act, colp(lvp,n,c), epmat(1,c), s%ld), 'reduced matrix')
        call chk(e%ctx, dla_gram(e%ctx, n, s%cols, vm, s%act, colp(lvm,n,c), emmat(1,c), s%ld), 'reduced matrix')
        if (lo.gt.0) then
          epmat(c:s%cols,1:lo) = transpose(epmat(1:lo,c:s%cols))
          emmat(c:s%cols,1:lo) = transpose(emmat(1:lo,c:s%cols))
        end if
      end if
!
!     reduced eigenproblem: eig = w (traditional) or 1/w (efficient), coefficient blocks up, um
!
      call lap_start(w)
      if (traditional) then
        if (dla_get_option(e%ctx, opt_lr_alg).eq.1) then
          call lr_pairs_helmich_paris(s%cols, s%ld, n_max, epmat, emmat, smat, eig, up, um)
        else
          call lr_reduced_pairs(s%cols, s%ld, n_max, epmat, emmat, smat, eig, up, um)
        end if
      else
!       largest pairs of s^T s (:1293-1311) = lowest of -s^T s; u- = s u+ / eig (:1315-1318)
        sts(1:s%cols,1:s%cols) = -matmul(transpose(smat(1:s%cols,1:s%cols)), smat(1:s%cols,1:s%cols))
        call need_eigensolver(dla_syev_lowest('u', s%cols, sts, s%ld, lam, n_max))
        do r = 1, n_max
          eig(r)          = sqrt(-lam(r))
          up(1:s%cols,r)  = sts(1:s%cols,r)
        end do
        um(1:s%cols,:) = matmul(smat(1:s%cols,1:s%cols), up(1:s%cols,:))
        do r = 1, n_max
          um(1:s%cols,r) = um(1:s%cols,r)/eig(r)
        end do
      end if
      call lap_charge(w, w%diag)
!
!     residuals of the open wanted roots and their norms.  The fused sweep forms  t - eig * b  from two n x n_max blocks:
!       traditional (:872-889):  rp = lvp u+ - eig bvp u-      rm = lvm u- - eig bvm u+
!       efficient   (:1337-1353): rp = bvp u- - eig lvp u+      rm = bvm u+ - eig lvm u-
!
      call sub_refresh_mask(s)
      if (traditional) then
        call residual_pair(bvp, um, lvp, up, rp, np)
        call residual_pair(bvm, up, lvm, um, rm, nm)
      else
        call residual_pair(lvp, up, bvp, um, rp, np)
        call residual_pair(lvm, um, bvm, up, rm, nm)
      end if
      do r = 1, n_targ
        if (s%locked(r)) cycle
        if (traditional) then
          s%rnorm(:,r) = np(:,r) + nm(:,r)
        else
          s%rnorm(1,r) = (np(1,r) + nm(1,r))/(eig(r)*sqrt2)
          s%rnorm(2,r) = (np(2,r) + nm(2,r))/(sqrt2*eig(r))
        end if
      end do
!
      call sub_lock(s, it, n_targ)
      if (verbose) then
        if (traditional) then
          call print_table_rows(s, it, eig)
        else
          call print_table_rows(s, it, one/eig)
        end if
      end if
      if (sub_finished(s)) then
        ok = .true.
        sweeps = it
        exit
      end if
!
      if (sub_has_room(s)) then
!
!       expand both spaces with the preconditioned residuals (:928-955, 1397-1424)
!
        call sub_open_block(s, first)
        call chk(e%ctx, dla_call_lrprec(e%ctx, prec, n, s%act, eig(first), colp(rp,n,first), colp(rm,n,first), &
                                        colp(vp,n,s%head), colp(vm,n,s%head)), 'lrprec')
        call lap_start(w)
        if (traditional) then
          call chk(e%ctx, dla_ortho_vs_x(e%ctx, n, s%cols, s%act, vp, colp(vp,n,s%head)), 'ortho_vs_x')
          call chk(e%ctx, dla_ortho_vs_x(e%ctx, n, s%cols, s%act, vm, colp(vm,n,s%head)), 'ortho_vs_x')
        else
!
!         efficient variant: each new block is orthogonalised against its basis in the metric (:1397-1416), gets its metric
!         image (:1417-1423) and is made orthonormal in that metric (:1420-1424) -- one call per block (mode 2 of
!         dla_expand_project_metric: device chain, operator and the Cholesky step of b_ortho back to back, one host wait)
!
          call chk(e%ctx, dla_expand_project_metric(e%ctx, 2_c_int, n, s%cols, s%act, vp, lvp, c_null_ptr, c_null_funptr, apb, zero, &
                                                    nohmat, 1_c_int), 'b_ortho_vs_x + apbmul + b_ortho')
          call chk(e%ctx, dla_expand_project_metric(e%ctx, 2_c_int, n, s%cols, s%act, vm, lvm, c_null_ptr, c_null_funptr, amb, zero, &
                                                    nohmat, 1_c_int), 'b_ortho_vs_x + ambmul + b_ortho')
          e%op_cols = e%op_cols + 2*s%act
        end if
        call lap_charge(w, w%ortho)
      else
!
!       restart from the current Ritz vectors (:957-991, 1426-1463)
!
        if (verbose) write(6,'(t7,a)') 'Restarting davidson.'
        e%restarts = e%restarts + 1
        call gather_vectors()
        call sub_collapse(s)
        call lr_split(e%ctx, n, n2, n_max, e%ritz, vp, vm)
        call condition_first_block()
        smat = zero
        if (traditional) then
          epmat = zero
          emmat = zero
        end if
      end if
      if (verbose) call print_block_report(s)
    end do
!
!   evec holds the current approximation on every exit, like the reference (:865-868, 1330-1333); the efficient
!   variant iterates on 1/w and hands w back for the converged solve (:1377-1380)
!
    if (.not.vectors_current) call gather_vectors()
    if (ok .and. .not.traditional) eig(1:n_targ) = one/eig(1:n_targ)
    call clock_now(t_end)
    w%total = t_end - t_begin
    if (verbose) then
      if (traditional) then
        call print_timings('timings for caslr (cpu/wall):   ', w)
      else
        call print_timings('timings for caslr_eff (cpu/wall):   ', w)
      end if
    end if
!
    call env_close(e, evec, sweeps)
    call drop_panel(e%ctx, vp);  call drop_panel(e%ctx, vm)
    call drop_panel(e%ctx, lvp); call drop_panel(e%ctx, lvm)
    call drop_panel(e%ctx, bvp); call drop_panel(e%ctx, bvm)
    call drop_panel(e%ctx, rp);  call drop_panel(e%ctx, rm)
    call drop_panel(e%ctx, bp);  call drop_panel(e%ctx, bm)
    call drop_panel(e%ctx, tp);  call drop_panel(e%ctx, tm)
!
  contains
!
!   make the first block of vp, vm orthonormal the way the variant wants it
!
    subroutine condition_first_block()
      if (traditional) then
        call chk(e%ctx, dla_ortho_cd(e%ctx, n, n_max, vp, growth, flag), 'ortho_cd')
        call chk(e%ctx, dla_ortho_cd(e%ctx, n, n_max, vm, growth, flag), 'ortho_cd')
      else
        call metric_blocks(1, n_max)
      end if
    end subroutine condition_first_block
!
!   efficient variant: (A+B) vp and (A-B) vm for a new block, which is then made orthonormal in its metric
!   (:1256-1259, 1420-1424)
!
    subroutine metric_blocks(c0, k)
      integer, intent(in) :: c0, k
      call chk(e%ctx, dla_call_matvec(e%ctx, apb, n, k, colp(vp,n,c0), colp(lvp,n,c0)), 'apbmul')
      call chk(e%ctx, dla_b_ortho(e%ctx, n, k, colp(vp,n,c0), colp(lvp,n,c0)), 'b_ortho')
      call chk(e%ctx, dla_call_matvec(e%ctx, amb, n, k, colp(vm,n,c0), colp(lvm,n,c0)), 'ambmul')
      call chk(e%ctx, dla_b_ortho(e%ctx, n, k, colp(vm,n,c0), colp(lvm,n,c0)), 'b_ortho')
      e%op_cols = e%op_cols + 2*k
    end subroutine metric_blocks
!
!   res = tpanel ct - eig * (bpanel cb) for the open wanted roots, with the residual norms; bb receives bpanel cb
!
    subroutine residual_pair(bpanel, cb, tpanel, ct, res, norms)
      type(c_ptr), intent(in)    :: bpanel, tpanel, res
      real(dp),    intent(in)    :: cb(s%ld,n_max), ct(s%ld,n_max)
      real(dp),    intent(inout) :: norms(2,n_max)
!     (one sweep over the two panels with the two coefficient blocks; the reference: two dgemms, then daxpy / dnrm2 per root)
      call chk(e%ctx, dla_ritz_residual2(e%ctx, n, s%cols, n_max, bpanel, tpanel, cb, s%ld, ct, s%ld, eig, n_targ, s%mask, &
                                         bp, res, tp, tm, norms), 'residual')
    end subroutine residual_pair
!
!   Ritz vectors vp u+ and vm u- (in the scratch blocks bp, bm), then Y = sum, Z = difference (:862-868, 1324-1333)
!
    subroutine gather_vectors()
      integer :: jj
      if (s%cols.le.0) return
      call chk(e%ctx, dla_panel_gemm(e%ctx, n, s%cols, vp, n_max, up, s%ld, bp), 'ritz vectors')
      call chk(e%ctx, dla_panel_gemm(e%ctx, n, s%cols, vm, n_max, um, s%ld, bm), 'ritz vectors')
      do jj = 1, n_max
        call chk(e%ctx, dla_copy(e%ctx, lr_half(e%ritz,n,n2,jj,0), colp(bp,n,jj), nbytes(n,1)), 'copy')
        call chk(e%ctx, dla_axpy(e%ctx, nelem(n,1), one, colp(bm,n,jj), lr_half(e%ritz,n,n2,jj,0)), 'axpy')
        call chk(e%ctx, dla_copy(e%ctx, lr_half(e%ritz,n,n2,jj,1), colp(bp,n,jj), nbytes(n,1)), 'copy')
        call chk(e%ctx, dla_axpy(e%ctx, nelem(n,1), -one, colp(bm,n,jj), lr_half(e%ritz,n,n2,jj,1)), 'axpy')
      end do
      vectors_current = .true.
    end subroutine gather_vectors
  end subroutine lr_core
!
  subroutine caslr_driver(verbose,n,n2,n_targ,n_max,max_iter,tol,max_dav,apbmul,ambmul,spdmul,smdmul,lrprec,eig,evec,ok)
    logical,  intent(in)            :: verbose
    integer,  intent(in)            :: n, n2, n_targ, n_max, max_iter, max_dav
    real(dp), intent(in)            :: tol
    real(dp), intent(inout)         :: eig(n_max)
    real(dp), intent(inout), target :: evec(n2,n_max)
    logical,  intent(inout)         :: ok
    external                        :: apbmul, ambmul, spdmul, smdmul, lrprec
    call lr_core(.true., verbose, n, n2, n_targ, n_max, max_iter, tol, max_dav, c_funloc(apbmul), c_funloc(ambmul), &
                 c_funloc(spdmul), c_funloc(smdmul), c_funloc(lrprec), eig, evec, ok)
  end subroutine caslr_driver
!
  subroutine caslr_eff_driver(verbose,n,n2,n_targ,n_max,max_iter,tol,max_dav,apbmul,ambmul,spdmul,smdmul,lrprec, &
                              eig,evec,ok)
    logical,  intent(in)            :: verbose
    integer,  intent(in)            :: n, n2, n_targ, n_max, max_iter, max_dav
    real(dp), intent(in)            :: tol
    real(dp), intent(inout)         :: eig(n_max)
    real(dp), intent(inout), target :: evec(n2,n_max)
    logical,  intent(inout)         :: ok
    external                        :: apbmul, ambmul, spdmul, smdmul, lrprec
    call lr_core(.false., verbose, n, n2, n_targ, n_max, max_iter, tol, max_dav, c_funloc(apbmul), c_funloc(ambmul), &
                 c_funloc(spdmul), c_funloc(smdmul), c_funloc(lrprec), eig, evec, ok)
  end subroutine caslr_eff_driver
!
! vp = Y + Z, vm = Y - Z for the n_max columns of the 2n x n_max block ev (reference :709-712, 1249-1252)
!
  subroutine lr_split(ctx, n, n2, n_max, ev, vp, vm)
    type(c_ptr), intent(in) :: ctx, ev, vp, vm
    integer,     intent(in) :: n, n2, n_max
    integer :: jj
    do jj = 1, n_max
      call chk(ctx, dla_copy(ctx, colp(vp,n,jj), lr_half(ev,n,n2,jj,0), nbytes(n,1)), 'copy')
      call chk(ctx, dla_axpy(ctx, nelem(n,1), one, lr_half(ev,n,n2,jj,1), colp(vp,n,jj)), 'axpy')
      call chk(ctx, dla_copy(ctx, colp(vm,n,jj), lr_half(ev,n,n2,jj,0), nbytes(n,1)), 'copy')
      call chk(ctx, dla_axpy(ctx, nelem(n,1), -one, lr_half(ev,n,n2,jj,1), colp(vm,n,jj)), 'axpy')
    end do
  end subroutine lr_split
!
! device address of the upper (half = 0) or lower (half = 1) n rows of column j of a 2n x m block
!
  function lr_half(ev, n, n2, j, half) result(p)
    type(c_ptr), intent(in) :: ev
    integer,     intent(in) :: n, n2, j, half
    type(c_ptr)             :: p
    integer(c_intptr_t)     :: a
    a = transfer(ev, a) + 8_c_intptr_t * (int(n2,c_intptr_t) * int(j-1,c_intptr_t) + int(half*n,c_intptr_t))
    p = transfer(a, p)
  end function lr_half
!
! the n_max largest eigenpairs of (0 s^T; s 0) u = (1/w) diag(E+,E-) u (dsygv itype 1, reference :783):
! eig = w, up/um = the two halves of u, normalised u^T diag(E+,E-) u = 1
!
  subroutine lr_reduced_pairs(ldu, lda, n_max, epmat, emmat, smat, eig, up, um)
    integer,  intent(in)    :: ldu, lda, n_max
    real(dp), intent(in)    :: epmat(lda,lda), emmat(lda,lda), smat(lda,lda)
    real(dp), intent(inout) :: eig(n_max), up(lda,n_max), um(lda,n_max)
    real(dp), allocatable   :: lp(:,:), lm(:,:), mm(:,:), cc(:,:), w(:)
    integer                 :: i, j, l2
    integer(c_int)          :: info
    l2 = 2*ldu
    allocate (lp(ldu,ldu), lm(ldu,ldu), mm(ldu,ldu), cc(l2,l2), w(l2))
    lp = epmat(1:ldu,1:ldu)
    lm = emmat(1:ldu,1:ldu)
    info = dla_potrf_lower(ldu, lp, ldu)
    if (info.eq.0) info = dla_potrf_lower(ldu, lm, ldu)
    if (info.ne.0) then
      write(6,'(t3,a)') 'DSYGV failed in caslr_driver'
      error stop 1
    end if
    do j = 2, ldu
      lp(1:j-1,j) = zero
      lm(1:j-1,j) = zero
    end do
    info = dla_trtri_lower(ldu, lp, ldu)
    info = dla_trtri_lower(ldu, lm, ldu)
!   M = L-^-1 s L+^-T
    mm = matmul(lm, matmul(smat(1:ldu,1:ldu), transpose(lp)))
    cc = zero
    cc(ldu+1:l2,1:ldu) = -mm
    cc(1:ldu,ldu+1:l2) = -transpose(mm)
    info = dla_syev_lowest('l', l2, cc, l2, w, n_max)
    if (info.ne.0) then
      write(6,'(t3,a,i6)') 'dsyev failed. info = ',info
      error stop 1
    end if
    do i = 1, n_max
      eig(i)      = -one/w(i)
      up(1:ldu,i) = matmul(transpose(lp), cc(1:ldu,i))
      um(1:ldu,i) = matmul(transpose(lm), cc(ldu+1:l2,i))
    end do
    deallocate (lp, lm, mm, cc, w)
  end subroutine lr_reduced_pairs
!
!
! Helmich-Paris route to the same pairs (reference diaglib.f90:805-860, its i_alg = 1): with s = U1 S1 V1^T,
! Vs = V1 S1^-1/2, Us = U1 S1^-1/2, the scaled blocks E+~ = Vs^T E+ Vs = L+ L+^T and E-~ = Us^T E- Us = L- L-^T, and
! C = L-^T L+ = U2 S2 V2^T, the eigenvalues w are the SMALLEST singular values of C and
!   u+ = Vs L- u2 / (sqrt(2) w),   u- = Us L+ v2 / (sqrt(2) w).
! The singular value decompositions come from the symmetric eigensolver applied to (0 M^T; M 0), whose positive
! eigenpairs are (sigma, (v; u)/sqrt(2)): no LAPACK at run time, full accuracy for small singular values.
!
  subroutine lr_pairs_helmich_paris(ldu, lda, n_max, epmat, emmat, smat, eig, up, um)
    integer,  intent(in)    :: ldu, lda, n_max
    real(dp), intent(in)    :: epmat(lda,lda), emmat(lda,lda), smat(lda,lda)
    real(dp), intent(inout) :: eig(n_max), up(lda,n_max), um(lda,n_max)
    real(dp), allocatable   :: u1(:,:), v1(:,:), s1(:), u2(:,:), v2(:,:), s2(:), lp(:,:), lm(:,:), cm(:,:)
    integer                 :: i, j, pick
    integer(c_int)          :: info
    allocate (u1(ldu,ldu), v1(ldu,ldu), s1(ldu), u2(ldu,ldu), v2(ldu,ldu), s2(ldu), lp(ldu,ldu), lm(ldu,ldu), cm(ldu,ldu))
    call small_svd(ldu, smat(1:ldu,1:ldu), u1, s1, v1)
    do i = 1, ldu
      u1(:,i) = u1(:,i)/sqrt(s1(i))
      v1(:,i) = v1(:,i)/sqrt(s1(i))
    end do
    lp = matmul(transpose(v1), matmul(epmat(1:ldu,1:ldu), v1))
    lm = matmul(transpose(u1), matmul(emmat(1:ldu,1:ldu), u1))
    info = dla_potrf_lower(ldu, lp, ldu)
    if (info.eq.0) info = dla_potrf_lower(ldu, lm, ldu)
    if (info.ne.0) then
      write(6,'(t3,a)') 'Cholesky factorisation failed in caslr_driver (Helmich-Paris route)'
      error stop 1
    end if
    do j = 2, ldu
      lp(1:j-1,j) = zero
      lm(1:j-1,j) = zero
    end do
    cm = matmul(transpose(lm), lp)
    call small_svd(ldu, cm, u2, s2, v2)
    do i = 1, n_max
      pick = ldu - i + 1                                     ! singular values come in descending order
      eig(i)      = s2(pick)
      up(1:ldu,i) = matmul(v1, matmul(lm, u2(:,pick)))/(sqrt(2.0_dp)*s2(pick))
      um(1:ldu,i) = matmul(u1, matmul(lp, v2(:,pick)))/(sqrt(2.0_dp)*s2(pick))
    end do
    deallocate (u1, v1, s1, u2, v2, s2, lp, lm, cm)
  end subroutine lr_pairs_helmich_paris
!
! a = u diag(sv) v^T for a square matrix, singular values in descending order
!
  subroutine small_svd(m, a, u, sv, v)
    integer,  intent(in)  :: m
    real(dp), intent(in)  :: a(m,m)
    real(dp), intent(out) :: u(m,m), sv(m), v(m,m)
    real(dp), allocatable :: jw(:,:), ev(:)
    integer               :: i, col
    integer(c_int)        :: info
    allocate (jw(2*m,2*m), ev(2*m))
    jw = zero
    jw(m+1:2*m,1:m) = a
    jw(1:m,m+1:2*m) = transpose(a)
    info = dla_syev('l', 2*m, jw, 2*m, ev)
    call need_eigensolver(info)
    do i = 1, m
      col   = 2*m - i + 1
      sv(i) = ev(col)
      v(:,i) = jw(1:m,col)*sqrt(2.0_dp)
      u(:,i) = jw(m+1:2*m,col)*sqrt(2.0_dp)
    end do
    deallocate (jw, ev)
  end subroutine small_svd
!
end module diaglib
