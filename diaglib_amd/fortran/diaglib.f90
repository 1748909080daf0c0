!
! diaglib (MI355X-native) -- drop-in for the public interface of Molecolab-Pisa/diaglib.
!
! Same module name, same public procedures and argument lists as the reference
! (reference diaglib.f90:166-167, 1483-1539, 171-228, 3185, 3481, 3094, 3576, 3052), so an
! existing CI / augmented-Hessian caller only re-links.  What differs is where the work
! happens: the drivers below keep the host control flow (iteration, locking, restart,
! the small Rayleigh-Ritz problem) and every O(n) operation is a call through
! ISO_C_BINDING into the HIP engine (include/diaglib_amd.h).  The expansion panels
! space/aspace/r live in HBM for the whole solve; only lda x lda matrices visit the host.
!
! Callbacks keep the reference shape  matvec(n,m,x,ax) / precnd(n,m,fac,x,px)
! (reference README.md:34-35).  By default they receive HOST arrays (the engine stages
! blocks through pinned memory); after  call diaglib_amd_config(callbacks_on_device=.true.)
! they receive DEVICE addresses under the same signature (SURVEY.md 8b).
!
! The generalised problem (metric B through a bvec callback) is served by gen_david_driver and by
! lobpcg_driver with gen_eig=.true. (reference diaglib.f90:1855-2250, 299-302/357-364/523-526).
! The linear-response problem is served by caslr_eff_driver (reference diaglib.f90:1024-1481) and caslr_driver
! (:558-1022, its default algorithm i_alg = 0), built from the same device operations.  Not provided (SURVEY.md 2
! row 5): nonsym_driver, and the Helmich-Paris variant of caslr_driver (i_alg = 1, a switch of the reference
! harness' module utils).
!
module diaglib
  use real_precision
  use iso_c_binding
  implicit none
  private
!
  public :: lobpcg_driver, davidson_driver, gen_david_driver, ortho, b_ortho, ortho_cd, ortho_vs_x, b_ortho_vs_x
  public :: caslr_eff_driver, caslr_driver
  public :: diaglib_amd_config
!
  real(dp), parameter :: zero = 0.0_dp, one = 1.0_dp, ten = 10.0_dp
  integer,  parameter :: min_dav = 10          ! reference diaglib.f90:1544
!
! option ids of include/diaglib_amd.h
!
  integer(c_int), parameter :: opt_cb_dev = 1, opt_evec_dev = 2
!
! timers: (cpu, wall) pairs like the reference's t_mv, t_diag, t_ortho, t_tot
!
  real(dp) :: t_mv(2), t_diag(2), t_ortho(2), t_tot(2), t1(2), t2(2)
!
  interface
    function dla_default_ctx() bind(C,name='dla_default_ctx') result(ctx)
      import :: c_ptr
      type(c_ptr) :: ctx
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
! configuration (extension; defaults reproduce the reference contract: host callbacks,
! host eig/evec)
! ---------------------------------------------------------------------------------------
  subroutine diaglib_amd_config(callbacks_on_device, evec_on_device, release_cache)
    logical, intent(in), optional :: callbacks_on_device, evec_on_device
!   release_cache = .true.: hand the panels the allocator keeps between solves back to the runtime (dla_trim)
    logical, intent(in), optional :: release_cache
    integer(c_size_t) :: released
    type(c_ptr)    :: ctx
    integer(c_int) :: st
    ctx = dla_default_ctx()
    if (present(callbacks_on_device)) st = dla_set_option(ctx, opt_cb_dev, merge(1_c_int,0_c_int,callbacks_on_device))
    if (present(evec_on_device))      st = dla_set_option(ctx, opt_evec_dev, merge(1_c_int,0_c_int,evec_on_device))
    if (present(release_cache)) then
      if (release_cache) st = dla_trim(ctx, released)
    end if
  end subroutine diaglib_amd_config
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
  subroutine get_time(t)
    real(dp), intent(inout) :: t(2)
    integer(8) :: cnt, rate
    call cpu_time(t(1))
    call system_clock(cnt, rate)
    t(2) = real(cnt,dp)/real(rate,dp)
  end subroutine get_time
!
! ---------------------------------------------------------------------------------------
! Davidson-Liu (reference diaglib.f90:1483-1853)
! ---------------------------------------------------------------------------------------
  subroutine davidson_driver(verbose,n,n_targ,n_max,max_iter,tol,max_dav,shift,matvec,precnd,eig,evec,ok)
    logical,                              intent(in)    :: verbose
    integer,                              intent(in)    :: n, n_targ, n_max
    integer,                              intent(in)    :: max_iter, max_dav
    real(dp),                             intent(in)    :: tol, shift
    real(dp), dimension(n_max),           intent(inout) :: eig
    real(dp), dimension(n,n_max), target, intent(inout) :: evec
    logical,                              intent(inout) :: ok
    external                                            :: matvec, precnd
!
    type(c_ptr)    :: ctx, space, aspace, r, evd
    type(c_funptr) :: mv, pc
    integer        :: dim_dav, lda, n_act, ind, i_beg, m_dim, ldu, n_frozen, it, i_eig, n_rst, c0
    integer        :: n_mv, n_restarts
    logical        :: restart, evec_dev
    real(dp)       :: tol_rms, tol_max
    logical,        allocatable :: done(:)
    integer(c_int), allocatable :: skip(:)
    real(dp),       allocatable :: a_red(:,:), a_copy(:,:), e_red(:), r_norm(:,:)
    integer(c_int) :: info
!
    ctx = dla_default_ctx()
    mv  = c_funloc(matvec)
    pc  = c_funloc(precnd)
    evec_dev = dla_get_option(ctx, opt_evec_dev) .ne. 0
!
!   expansion space size: never smaller than 10 blocks (reference :1595-1596)
!
    dim_dav = max(min_dav,max_dav)
    lda     = dim_dav*n_max
!
!   device panels (reference :1607) and host-size matrices (:1612-1617)
!
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), space),  'allocation of space')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), aspace), 'allocation of aspace')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), r),    'allocation of r')
    if (evec_dev) then
      evd = c_loc(evec)
    else
      call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), evd), 'allocation of evec')
      call chk(ctx, dla_upload(ctx, evd, c_loc(evec), nbytes(n,n_max)), 'upload of the guess')
    end if
    allocate (done(n_max), skip(n_max), r_norm(2,n_max), a_red(lda,lda), a_copy(lda,lda), e_red(lda))
!
    tol_rms = tol
    tol_max = ten * tol
    t_diag  = zero
    t_ortho = zero
    t_mv    = zero
    t_tot   = zero
!
!   the reference zero-fills both n x lda panels here (:1632-1633).  On the device no column is
!   ever read before it has been written (guess copy, matvec output, ortho_vs_x output), so the
!   two 8*n*lda-byte memsets are skipped; the columns the restart quirk reads as zeros are
!   zeroed at the restart (below).
!
    a_red   = zero
    r_norm  = zero
    ok      = .false.
    done    = .false.
!
    call get_time(t_tot)
!
!   guess: orthonormalise if needed, random if zero (reference :1644), then copy (:1648)
!
    call chk(ctx, dla_check_guess(ctx, n, n_max, evd), 'check_guess')
    call chk(ctx, dla_copy(ctx, space, evd, nbytes(n,n_max)), 'copy')
!
    n_act = n_max
    ind   = 1
    i_beg = 1
    m_dim = 1
    ldu   = 0
    restart = .false.
    n_rst   = 0
    n_frozen = 0
    n_mv = 0
    n_restarts = 0
!
    1030 format(t5,'Davidson-Liu iterations (tol=',d10.2,'):',/, &
                t5,'------------------------------------------------------------------',/, &
                t7,'  iter  root              eigenvalue','         rms         max ok',/, &
                t5,'------------------------------------------------------------------')
    1040 format(t9,i4,2x,i4,f24.12,2d12.4,l3)
    if (verbose) write(6,1030) tol
!
    do it = 1, max_iter
      ldu = ldu + n_act
      c0  = i_beg + n_rst
!
!     A times the new block (reference :1685)
!
      call get_time(t1)
      call chk(ctx, dla_call_matvec(ctx, mv, n, n_act, colp(space,n,c0), colp(aspace,n,c0)), 'matvec')
      call get_time(t2)
      t_mv = t_mv + t2 - t1
      n_mv = n_mv + n_act
!
!     new columns of the projected matrix (reference :1691)
!
      call chk(ctx, dla_gram(ctx, n, ldu, space, n_act, colp(aspace,n,c0), a_red(1,c0), lda), 'projection')
!
!     after a restart the locked roots enter through their eigenvalues (reference :1696-1702)
!
      if (restart) then
        do i_eig = 1, n_rst
          a_red(i_eig,i_eig) = e_red(i_eig)
        end do
        restart = .false.
        n_rst   = 0
      end if
      a_copy = a_red
!
      call get_time(t1)
      info = dla_syev_lowest('u', ldu, a_copy, lda, e_red, n_max)   ! only a_copy(:,1:n_max) is used below
      call get_time(t2)
      t_diag = t_diag + t2 - t1
      if (info.ne.0) then
        write(6,'(t3,a,i6)') 'dsyev failed. info = ',info
        stop
      end if
      eig = e_red(1:n_max)
!
!     Ritz vectors, residuals and their norms in one sweep (reference :1717-1732)
!
      do i_eig = 1, n_max
        skip(i_eig) = merge(1_c_int, 0_c_int, done(i_eig))
      end do
      call chk(ctx, dla_ritz_residual(ctx, n, ldu, n_max, space, aspace, a_copy, lda, eig, n_targ, skip, &
                                      evd, r, c_null_ptr, r_norm), 'ritz/residual')
!
!     lock the leading converged roots (reference :1737-1746)
!
      do i_eig = 1, n_targ
        if (done(i_eig)) cycle
        done(i_eig) = r_norm(1,i_eig).lt.tol_rms .and. r_norm(2,i_eig).lt.tol_max .and. it.gt.1
        if (.not.done(i_eig)) then
          done(i_eig+1:n_max) = .false.
          exit
        end if
      end do
!
      if (verbose) then
        do i_eig = 1, n_targ
          write(6,1040) it, i_eig, eig(i_eig) - shift, r_norm(:,i_eig), done(i_eig)
        end do
        write(6,*)
      end if
!
      if (all(done(1:n_targ))) then
        ok = .true.
        exit
      end if
!
      if (m_dim .lt. dim_dav) then
!
!       expand: precondition the active residuals, orthogonalise against the space
!       (reference :1773-1794)
!
        m_dim = m_dim + 1
        i_beg = i_beg + n_act
        n_act = n_max
        n_frozen = 0
        do i_eig = 1, n_targ
          if (done(i_eig)) then
            n_act = n_act - 1
            n_frozen = n_frozen + 1
          else
            exit
          end if
        end do
        ind = n_max - n_act + 1
        call chk(ctx, dla_call_precnd(ctx, pc, n, n_act, -eig(ind), colp(r,n,ind), colp(space,n,i_beg)), 'precnd')
        call get_time(t1)
        call chk(ctx, dla_ortho_vs_x(ctx, n, ldu, n_act, space, colp(space,n,i_beg)), 'ortho_vs_x')
        call get_time(t2)
        t_ortho = t_ortho + t2 - t1
      else
!
!       restart from the current Ritz vectors (reference :1796-1824)
!
        if (verbose) write(6,'(t7,a)') 'Restarting davidson.'
        n_restarts = n_restarts + 1
        n_act = n_max
        call chk(ctx, dla_copy(ctx, space, evd, nbytes(n,n_max)), 'copy')
        a_red = zero
        ldu   = 0
        i_beg = 1
        m_dim = 1
        n_rst = 0
        do i_eig = 1, n_targ
          if (done(i_eig)) then
            n_rst = n_rst + 1
          else
            exit
          end if
        end do
!
!       the reference zeroes space and aspace entirely (:1798,1804).  What the next iteration
!       actually reads from those zeros: matvec takes n_max columns starting at column 1+n_rst,
!       i.e. it runs n_rst columns past the Ritz block (they must be zero), and the locked
!       columns 1..n_rst of aspace stay zero (their eigenvalues are patched into a_red, :1696-1702).
!
        if (n_rst.gt.0) then
          call chk(ctx, dla_zero(ctx, colp(space,n,n_max+1), nbytes(n,n_rst)), 'zero')
          call chk(ctx, dla_zero(ctx, aspace, nbytes(n,n_rst)), 'zero')
        end if
        restart = .true.
      end if
      if (verbose) write(6,1050) n_targ, n_act, n_frozen
    end do
!
    call get_time(t2)
    t_tot = t2 - t_tot
    call dla_set_solve_info(int(min(it,max_iter),c_int), int(n_mv,c_int), int(n_restarts,c_int))
!
    1000 format(t3,'timings for davidson (cpu/wall): ',/, &
                t3,'  matrix-vector multiplications: ',2f12.4,/, &
                t3,'  diagonalization:               ',2f12.4,/, &
                t3,'  orthogonalization:             ',2f12.4,/, &
                t3,'                                 ',24('='),/,  &
                t3,'  total:                         ',2f12.4)
    if (verbose) write(6,1000) t_mv, t_diag, t_ortho, t_tot
!
!   hand the Ritz vectors back (evec holds them on every exit, like the reference) and free
!
    if (.not.evec_dev) then
      call chk(ctx, dla_download(ctx, c_loc(evec), evd, nbytes(n,n_max)), 'download of evec')
      call chk(ctx, dla_free(ctx, evd), 'free')
    end if
    call chk(ctx, dla_free(ctx, space), 'free')
    call chk(ctx, dla_free(ctx, aspace), 'free')
    call chk(ctx, dla_free(ctx, r), 'free')
    deallocate (done, skip, r_norm, a_red, a_copy, e_red)
!
    1050 format(t5,'----------------------------------------',/,&
                t7,'# target vectors:    ',i4,/,&
                t7,'# new vectors added: ',i4,/,&
                t7,'# converged vectors: ',i4,/,&
                t5,'----------------------------------------')
    return
  end subroutine davidson_driver
!
! ---------------------------------------------------------------------------------------
! Davidson-Liu with a metric B (reference diaglib.f90:1855-2250): same loop plus bspace = B*space.
! Deliberate deviation at the restart (SURVEY 8a A13): the reference zeroes all of bspace right after
! B-orthonormalising the kept Ritz block (:2196-2200) and never refills it, so its residuals after a
! restart miss theta*B*x of that block; here the kept block's B*x stays in bspace(:,1:n_max).
! ---------------------------------------------------------------------------------------
  subroutine gen_david_driver(verbose,n,n_targ,n_max,max_iter,tol,max_dav,shift,matvec,precnd,bvec,eig,evec,ok)
    logical,                              intent(in)    :: verbose
    integer,                              intent(in)    :: n, n_targ, n_max
    integer,                              intent(in)    :: max_iter, max_dav
    real(dp),                             intent(in)    :: tol, shift
    real(dp), dimension(n_max),           intent(inout) :: eig
    real(dp), dimension(n,n_max), target, intent(inout) :: evec
    logical,                              intent(inout) :: ok
    external                                            :: matvec, precnd, bvec
!
    type(c_ptr)    :: ctx, space, aspace, bspace, b_evec, r, evd
    type(c_funptr) :: mv, pc, bv
    integer        :: dim_dav, lda, n_act, ind, i_beg, m_dim, ldu, n_frozen, it, i_eig, n_rst, c0
    integer        :: n_mv, n_restarts
    logical        :: restart, evec_dev
    real(dp)       :: tol_rms, tol_max
    logical,        allocatable :: done(:)
    integer(c_int), allocatable :: skip(:)
    real(dp),       allocatable :: a_red(:,:), a_copy(:,:), e_red(:), r_norm(:,:)
    integer(c_int) :: info
!
    ctx = dla_default_ctx()
    mv  = c_funloc(matvec)
    pc  = c_funloc(precnd)
    bv  = c_funloc(bvec)
    evec_dev = dla_get_option(ctx, opt_evec_dev) .ne. 0
!
!   expansion space size: never smaller than 10 blocks (reference :1595-1596)
!
    dim_dav = max(min_dav,max_dav)
    lda     = dim_dav*n_max
!
!   device panels (reference :1607) and host-size matrices (:1612-1617)
!
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), space),  'allocation of space')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), aspace), 'allocation of aspace')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), r),    'allocation of r')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), bspace), 'allocation of bspace')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), b_evec), 'allocation of b_evec')
    if (evec_dev) then
      evd = c_loc(evec)
    else
      call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), evd), 'allocation of evec')
      call chk(ctx, dla_upload(ctx, evd, c_loc(evec), nbytes(n,n_max)), 'upload of the guess')
    end if
    allocate (done(n_max), skip(n_max), r_norm(2,n_max), a_red(lda,lda), a_copy(lda,lda), e_red(lda))
!
    tol_rms = tol
    tol_max = ten * tol
    t_diag  = zero
    t_ortho = zero
    t_mv    = zero
    t_tot   = zero
!
!   the reference zero-fills both n x lda panels here (:1632-1633).  On the device no column is
!   ever read before it has been written (guess copy, matvec output, ortho_vs_x output), so the
!   two 8*n*lda-byte memsets are skipped; the columns the restart quirk reads as zeros are
!   zeroed at the restart (below).
!
    a_red   = zero
    r_norm  = zero
    ok      = .false.
    done    = .false.
!
    call get_time(t_tot)
!
!   guess: orthonormalise if needed, random if zero (reference :1644), then copy (:1648)
!
    call chk(ctx, dla_check_guess(ctx, n, n_max, evd), 'check_guess')
    call chk(ctx, dla_copy(ctx, space, evd, nbytes(n,n_max)), 'copy')
!
!   B times the guess, then B-orthonormalise it (reference :2033-2034)
!
    call chk(ctx, dla_call_matvec(ctx, bv, n, n_max, space, bspace), 'bvec')
    call chk(ctx, dla_b_ortho(ctx, n, n_max, space, bspace), 'b_ortho')
!
    n_act = n_max
    ind   = 1
    i_beg = 1
    m_dim = 1
    ldu   = 0
    restart = .false.
    n_rst   = 0
    n_frozen = 0
    n_mv = 0
    n_restarts = 0
!
    1030 format(t5,'Generalized Davidson-Liu iterations (tol=',d10.2,'):',/, &
                t5,'------------------------------------------------------------------',/, &
                t7,'  iter  root              eigenvalue','         rms         max ok',/, &
                t5,'------------------------------------------------------------------')
    1040 format(t9,i4,2x,i4,f24.12,2d12.4,l3)
    if (verbose) write(6,1030) tol
!
    do it = 1, max_iter
      ldu = ldu + n_act
      c0  = i_beg + n_rst
!
!     A times the new block (reference :1685)
!
      call get_time(t1)
      call chk(ctx, dla_call_matvec(ctx, mv, n, n_act, colp(space,n,c0), colp(aspace,n,c0)), 'matvec')
      call get_time(t2)
      t_mv = t_mv + t2 - t1
      n_mv = n_mv + n_act
!
!     new columns of the projected matrix (reference :1691)
!
      call chk(ctx, dla_gram(ctx, n, ldu, space, n_act, colp(aspace,n,c0), a_red(1,c0), lda), 'projection')
!
!     after a restart the locked roots enter through their eigenvalues (reference :1696-1702)
!
      if (restart) then
        do i_eig = 1, n_rst
          a_red(i_eig,i_eig) = e_red(i_eig)
        end do
        restart = .false.
        n_rst   = 0
      end if
      a_copy = a_red
!
      call get_time(t1)
      info = dla_syev_lowest('u', ldu, a_copy, lda, e_red, n_max)   ! only a_copy(:,1:n_max) is used below
      call get_time(t2)
      t_diag = t_diag + t2 - t1
      if (info.ne.0) then
        write(6,'(t3,a,i6)') 'dsyev failed. info = ',info
        stop
      end if
      eig = e_red(1:n_max)
!
!     Ritz vectors, residuals and their norms in one sweep (reference :1717-1732)
!
      do i_eig = 1, n_max
        skip(i_eig) = merge(1_c_int, 0_c_int, done(i_eig))
      end do
!     b_evec = BS y and r = AS y - eig * b_evec in one sweep, then evec = S y (reference :2108-2123)
      call chk(ctx, dla_ritz_residual(ctx, n, ldu, n_max, bspace, aspace, a_copy, lda, eig, n_targ, skip, &
                                      b_evec, r, c_null_ptr, r_norm), 'ritz/residual')
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, space, n_max, a_copy, lda, evd), 'ritz vectors')
!
!     lock the leading converged roots (reference :1737-1746)
!
      do i_eig = 1, n_targ
        if (done(i_eig)) cycle
        done(i_eig) = r_norm(1,i_eig).lt.tol_rms .and. r_norm(2,i_eig).lt.tol_max .and. it.gt.1
        if (.not.done(i_eig)) then
          done(i_eig+1:n_max) = .false.
          exit
        end if
      end do
!
      if (verbose) then
        do i_eig = 1, n_targ
          write(6,1040) it, i_eig, eig(i_eig) - shift, r_norm(:,i_eig), done(i_eig)
        end do
        write(6,*)
      end if
!
      if (all(done(1:n_targ))) then
        ok = .true.
        exit
      end if
!
      if (m_dim .lt. dim_dav) then
!
!       expand: precondition the active residuals, orthogonalise against the space
!       (reference :1773-1794)
!
        m_dim = m_dim + 1
        i_beg = i_beg + n_act
        n_act = n_max
        n_frozen = 0
        do i_eig = 1, n_targ
          if (done(i_eig)) then
            n_act = n_act - 1
            n_frozen = n_frozen + 1
          else
            exit
          end if
        end do
        ind = n_max - n_act + 1
        call chk(ctx, dla_call_precnd(ctx, pc, n, n_act, -eig(ind), colp(r,n,ind), colp(space,n,i_beg)), 'precnd')
        call get_time(t1)
        call chk(ctx, dla_b_ortho_vs_x(ctx, n, ldu, n_act, space, bspace, colp(space,n,i_beg)), 'b_ortho_vs_x')
        call chk(ctx, dla_call_matvec(ctx, bv, n, n_act, colp(space,n,i_beg), colp(bspace,n,i_beg)), 'bvec')
        call chk(ctx, dla_b_ortho(ctx, n, n_act, colp(space,n,i_beg), colp(bspace,n,i_beg)), 'b_ortho')
        call get_time(t2)
        t_ortho = t_ortho + t2 - t1
      else
!
!       restart from the current Ritz vectors (reference :1796-1824)
!
        if (verbose) write(6,'(t7,a)') 'Restarting davidson.'
        n_restarts = n_restarts + 1
        n_act = n_max
        call chk(ctx, dla_copy(ctx, space, evd, nbytes(n,n_max)), 'copy')
        call chk(ctx, dla_copy(ctx, bspace, b_evec, nbytes(n,n_max)), 'copy')
        call chk(ctx, dla_b_ortho(ctx, n, n_max, space, bspace), 'b_ortho')     ! reference :2195-2197
        a_red = zero
        ldu   = 0
        i_beg = 1
        m_dim = 1
        n_rst = 0
        do i_eig = 1, n_targ
          if (done(i_eig)) then
            n_rst = n_rst + 1
          else
            exit
          end if
        end do
!
!       the reference zeroes space and aspace entirely (:1798,1804).  What the next iteration
!       actually reads from those zeros: matvec takes n_max columns starting at column 1+n_rst,
!       i.e. it runs n_rst columns past the Ritz block (they must be zero), and the locked
!       columns 1..n_rst of aspace stay zero (their eigenvalues are patched into a_red, :1696-1702).
!
        if (n_rst.gt.0) then
          call chk(ctx, dla_zero(ctx, colp(space,n,n_max+1), nbytes(n,n_rst)), 'zero')
          call chk(ctx, dla_zero(ctx, aspace, nbytes(n,n_rst)), 'zero')
        end if
        restart = .true.
      end if
      if (verbose) write(6,1050) n_targ, n_act, n_frozen
    end do
!
    call get_time(t2)
    t_tot = t2 - t_tot
    call dla_set_solve_info(int(min(it,max_iter),c_int), int(n_mv,c_int), int(n_restarts,c_int))
!
    1000 format(t3,'timings for davidson (cpu/wall): ',/, &
                t3,'  matrix-vector multiplications: ',2f12.4,/, &
                t3,'  diagonalization:               ',2f12.4,/, &
                t3,'  orthogonalization:             ',2f12.4,/, &
                t3,'                                 ',24('='),/,  &
                t3,'  total:                         ',2f12.4)
    if (verbose) write(6,1000) t_mv, t_diag, t_ortho, t_tot
!
!   hand the Ritz vectors back (evec holds them on every exit, like the reference) and free
!
    if (.not.evec_dev) then
      call chk(ctx, dla_download(ctx, c_loc(evec), evd, nbytes(n,n_max)), 'download of evec')
      call chk(ctx, dla_free(ctx, evd), 'free')
    end if
    call chk(ctx, dla_free(ctx, space), 'free')
    call chk(ctx, dla_free(ctx, aspace), 'free')
    call chk(ctx, dla_free(ctx, r), 'free')
    call chk(ctx, dla_free(ctx, bspace), 'free')
    call chk(ctx, dla_free(ctx, b_evec), 'free')
    deallocate (done, skip, r_norm, a_red, a_copy, e_red)
!
    1050 format(t5,'----------------------------------------',/,&
                t7,'# target vectors:    ',i4,/,&
                t7,'# new vectors added: ',i4,/,&
                t7,'# converged vectors: ',i4,/,&
                t5,'----------------------------------------')
    return
  end subroutine gen_david_driver
!
! ---------------------------------------------------------------------------------------
! LOBPCG (reference diaglib.f90:171-556).  gen_eig=.true. is not on this path yet.
! ---------------------------------------------------------------------------------------
  subroutine lobpcg_driver(verbose,gen_eig,n,n_targ,n_max,max_iter,tol,shift,matvec,precnd,bvec,eig,evec,ok)
    logical,                              intent(in)    :: verbose, gen_eig
    integer,                              intent(in)    :: n, n_targ, n_max, max_iter
    real(dp),                             intent(in)    :: tol, shift
    real(dp), dimension(n_max),           intent(inout) :: eig
    real(dp), dimension(n,n_max), target, intent(inout) :: evec
    logical,                              intent(inout) :: ok
    external                                            :: matvec, precnd, bvec
!
    type(c_ptr)    :: ctx, space, aspace, bspace, r, x_new, ax_new, bx_new, evd, xfin
    type(c_ptr)    :: sp(2), asp(2), bsp(2)
    integer        :: cur, nxt
    type(c_funptr) :: mv, pc, bv
    integer        :: it, i_eig, n_act, ind_x, ind_w, ind_p, len_a, len_u, n_mv
    logical        :: evec_dev
    real(dp)       :: tol_rms, tol_max
    logical,        allocatable :: done(:)
    integer(c_int), allocatable :: skip(:)
    real(dp),       allocatable :: a_red(:,:), e_red(:), r_norm(:,:), u_x(:,:), u_p(:,:)
    integer(c_int) :: info
!
    ctx = dla_default_ctx()
    mv  = c_funloc(matvec)
    pc  = c_funloc(precnd)
    bv  = c_funloc(bvec)
    evec_dev = dla_get_option(ctx, opt_evec_dev) .ne. 0
    bsp    = c_null_ptr
    bspace = c_null_ptr
    bx_new = c_null_ptr
!
!   The reference keeps one basis [X P W] per panel and copies the new X (x_new, ax_new, bx_new) and the new
!   P (via its evec scratch) into it every iteration (:495-514).  Here every panel exists twice: an iteration
!   reads the basis from one copy and writes the new X and P blocks straight into the other, then the two
!   swap roles -- the four (six) block copies per iteration become none.
!
    len_a = 3*n_max
    do cur = 1, 2
      call chk(ctx, dla_alloc(ctx, nbytes(n,len_a), sp(cur)),  'allocation of space')
      call chk(ctx, dla_alloc(ctx, nbytes(n,len_a), asp(cur)), 'allocation of aspace')
!     (the reference allocates bspace/bx_new also for the standard problem, :259,270; here only when used)
      if (gen_eig) call chk(ctx, dla_alloc(ctx, nbytes(n,len_a), bsp(cur)), 'allocation of bspace')
    end do
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), r), 'allocation of r')
    cur = 1
    nxt = 2
    call select_panels()
    if (evec_dev) then
      evd = c_loc(evec)
    else
      call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), evd), 'allocation of evec')
      call chk(ctx, dla_upload(ctx, evd, c_loc(evec), nbytes(n,n_max)), 'upload of the guess')
    end if
    allocate (a_red(len_a,len_a), e_red(len_a), done(n_max), skip(n_max), r_norm(2,n_max))
!
    t_diag  = zero
    t_ortho = zero
    t_mv    = zero
    t_tot   = zero
!   (the reference zero-fills space/aspace/bspace, :284-286; every column is written before
!   it is read on this path, so the device panels are left uninitialised)
    a_red  = zero
    r_norm = zero
    n_mv   = 0
!
    call get_time(t_tot)
    call chk(ctx, dla_check_guess(ctx, n, n_max, evd), 'check_guess')
!
!   generalised problem: B times the guess, then B-orthonormalise it (reference :299-302)
!
    if (gen_eig) then
      call chk(ctx, dla_call_matvec(ctx, bv, n, n_max, evd, bspace), 'bvec')
      call chk(ctx, dla_b_ortho(ctx, n, n_max, evd, bspace), 'b_ortho')
    end if
!
!   first Rayleigh-Ritz step on the guess (reference :306-325)
!
    call chk(ctx, dla_copy(ctx, space, evd, nbytes(n,n_max)), 'copy')
    call get_time(t1)
    call chk(ctx, dla_call_matvec(ctx, mv, n, n_max, space, aspace), 'matvec')
    call get_time(t2)
    t_mv = t_mv + t2 - t1
    n_mv = n_mv + n_max
    if (shift.ne.zero) call chk(ctx, dla_axpy(ctx, int(n,c_size_t)*int(n_max,c_size_t), shift, space, aspace), 'axpy')
    call chk(ctx, dla_gram(ctx, n, n_max, space, n_max, aspace, a_red, len_a), 'projection')
    call get_time(t1)
    info = dla_syev_lowest('l', n_max, a_red, len_a, e_red, n_max)
    call get_time(t2)
    t_diag = t_diag + t2 - t1
    eig = e_red(1:n_max)
!
!   Ritz vectors x, a x and the first residuals r = a x - eig x in one sweep (:322-345)
!
    skip = 0
    if (gen_eig) then
!     b x = BS y and r = AS y - eig * (BS y) in one sweep over bspace/aspace, then x = S y (:322-346)
      call chk(ctx, dla_ritz_residual(ctx, n, n_max, n_max, bspace, aspace, a_red, len_a, eig, n_max, skip, &
                                      bx_new, r, ax_new, r_norm), 'ritz/residual')
      call chk(ctx, dla_panel_gemm(ctx, n, n_max, space, n_max, a_red, len_a, x_new), 'ritz vectors')
    else
      call chk(ctx, dla_ritz_residual(ctx, n, n_max, n_max, space, aspace, a_red, len_a, eig, n_max, skip, &
                                      x_new, r, ax_new, r_norm), 'ritz/residual')
    end if
    xfin = x_new
    call swap_panels()
!
!   first block of preconditioned residuals (:350-367)
!
    ind_x = 1
    ind_w = ind_x + n_max
    call chk(ctx, dla_call_precnd(ctx, pc, n, n_max, shift-eig(ind_x), colp(r,n,ind_x), colp(space,n,ind_w)), 'precnd')
    call get_time(t1)
    if (gen_eig) then
      call chk(ctx, dla_b_ortho_vs_x(ctx, n, n_max, n_max, space, bspace, colp(space,n,ind_w)), 'b_ortho_vs_x')
      call chk(ctx, dla_call_matvec(ctx, bv, n, n_max, colp(space,n,ind_w), colp(bspace,n,ind_w)), 'bvec')
      call chk(ctx, dla_b_ortho(ctx, n, n_max, colp(space,n,ind_w), colp(bspace,n,ind_w)), 'b_ortho')
    else
      call chk(ctx, dla_ortho_vs_x(ctx, n, n_max, n_max, space, colp(space,n,ind_w)), 'ortho_vs_x')
    end if
    call get_time(t2)
    t_ortho = t_ortho + t2 - t1
!
    tol_rms = tol
    tol_max = ten*tol
    ok      = .false.
    done    = .false.
    n_act   = n_max
!
    1030 format(t5,'LOBPCG iterations (tol=',d10.2,'):',/, &
                t5,'------------------------------------------------------------------',/, &
                t7,'  iter  root              eigenvalue','         rms         max ok',/, &
                t5,'------------------------------------------------------------------')
    1040 format(t9,i4,2x,i4,f24.12,2d12.4,l3)
    if (verbose) write(6,1030) tol
!
    do it = 1, max_iter
!
!     A times the W block (:394-397)
!
      call get_time(t1)
      call chk(ctx, dla_call_matvec(ctx, mv, n, n_act, colp(space,n,ind_w), colp(aspace,n,ind_w)), 'matvec')
      call get_time(t2)
      t_mv = t_mv + t2 - t1
      n_mv = n_mv + n_act
      if (shift.ne.zero) call chk(ctx, dla_axpy(ctx, int(n,c_size_t)*int(n_act,c_size_t), shift, &
                                                colp(space,n,ind_w), colp(aspace,n,ind_w)), 'axpy')
!
!     reduced matrix S^T A S, all of it, every iteration (:401-403)
!
      len_u = n_max + 2*n_act
      if (it.eq.1) len_u = 2*n_max
!     (dsyev below reads the lower triangle only, so only that part of the product is formed)
      call chk(ctx, dla_gram_lower(ctx, n, len_u, space, aspace, a_red, len_a), 'projection')
      call get_time(t1)
      info = dla_syev_lowest('l', len_u, a_red, len_a, e_red, n_max)   ! get_coeffs uses a_red(:,1:n_max) only
      call get_time(t2)
      t_diag = t_diag + t2 - t1
      if (info.ne.0) then
        write(6,'(t3,a,i6)') 'dsyev failed. info = ',info
        stop
      end if
      eig = e_red(1:n_max)
!
!     x_new, ax_new, residuals and norms in one sweep (:420-442)
!
      do i_eig = 1, n_max
        skip(i_eig) = merge(1_c_int, 0_c_int, done(i_eig))
      end do
      if (gen_eig) then
        call chk(ctx, dla_ritz_residual(ctx, n, len_u, n_max, bspace, aspace, a_red, len_a, eig, n_max, skip, &
                                        bx_new, r, ax_new, r_norm), 'ritz/residual')
        call chk(ctx, dla_panel_gemm(ctx, n, len_u, space, n_max, a_red, len_a, x_new), 'ritz vectors')
      else
        call chk(ctx, dla_ritz_residual(ctx, n, len_u, n_max, space, aspace, a_red, len_a, eig, n_max, skip, &
                                        x_new, r, ax_new, r_norm), 'ritz/residual')
      end if
      xfin = x_new
!
!     lock the leading converged roots (:446-455)
!
      do i_eig = 1, n_max
        if (done(i_eig)) cycle
        done(i_eig) = r_norm(1,i_eig).lt.tol_rms .and. r_norm(2,i_eig).lt.tol_max .and. it.gt.1
        if (.not.done(i_eig)) then
          done(i_eig+1:n_max) = .false.
          exit
        end if
      end do
!
      if (verbose) then
        do i_eig = 1, n_targ
          write(6,1040) it, i_eig, eig(i_eig) - shift, r_norm(:,i_eig), done(i_eig)
        end do
        write(6,*)
      end if
      if (all(done(1:n_targ))) then
        ok = .true.
        exit
      end if
!
      n_act = n_max - count(done)
      ind_x = n_max - n_act + 1
      ind_p = ind_x + n_act
      ind_w = ind_p + n_act
!
!     coefficients of the new P block (:485-488), then P = S u_p, AP = AS u_p (:495-498)
!
      allocate (u_x(len_u,n_max), u_p(len_u,max(n_act,1)))
      call chk(ctx, dla_get_coeffs(ctx, len_a, len_u, n_max, n_act, a_red, u_x, u_p), 'get_coeffs')
!     P = S u_p, AP = AS u_p [, BP = BS u_p] (:495-503) go straight into the P block of the other copy
      call chk(ctx, dla_panel_gemm(ctx, n, len_u, space,  n_act, u_p, len_u, colp(sp(nxt),n,ind_p)), 'p block')
      call chk(ctx, dla_panel_gemm(ctx, n, len_u, aspace, n_act, u_p, len_u, colp(asp(nxt),n,ind_p)), 'ap block')
      if (gen_eig) call chk(ctx, dla_panel_gemm(ctx, n, len_u, bspace, n_act, u_p, len_u, colp(bsp(nxt),n,ind_p)), 'bp block')
      deallocate (u_x, u_p)
!
!     x_new, ax_new [, bx_new] already are the X block of the other copy (:510-511): it becomes the basis
!
      call swap_panels()
!
!     new W block: preconditioned active residuals, orthogonalised against [X P] (:518-528)
!
      call chk(ctx, dla_call_precnd(ctx, pc, n, n_act, shift-eig(1), colp(r,n,ind_x), colp(space,n,ind_w)), 'precnd')
      call get_time(t1)
      if (gen_eig) then
        call chk(ctx, dla_b_ortho_vs_x(ctx, n, n_max+n_act, n_act, space, bspace, colp(space,n,ind_w)), 'b_ortho_vs_x')
        call chk(ctx, dla_call_matvec(ctx, bv, n, n_act, colp(space,n,ind_w), colp(bspace,n,ind_w)), 'bvec')
        call chk(ctx, dla_b_ortho(ctx, n, n_act, colp(space,n,ind_w), colp(bspace,n,ind_w)), 'b_ortho')
      else
        call chk(ctx, dla_ortho_vs_x(ctx, n, n_max+n_act, n_act, space, colp(space,n,ind_w)), 'ortho_vs_x')
      end if
      call get_time(t2)
      t_ortho = t_ortho + t2 - t1
    end do
!
    call get_time(t2)
    t_tot = t2 - t_tot
    call dla_set_solve_info(int(min(it,max_iter),c_int), int(n_mv,c_int), 0_c_int)
!
!   the current Ritz vectors go back in evec (the reference does this on convergence, :466;
!   on a non-converged exit it leaves its P-block scratch there -- we return x_new in both cases)
!
    call chk(ctx, dla_copy(ctx, evd, xfin, nbytes(n,n_max)), 'copy')
!
    1000 format(t3,'timings for lobpcg (cpu/wall):   ',/, &
                t3,'  matrix-vector multiplications: ',2f12.4,/, &
                t3,'  diagonalization:               ',2f12.4,/, &
                t3,'  orthogonalization:             ',2f12.4,/, &
                t3,'                                 ',24('='),/,  &
                t3,'  total:                         ',2f12.4)
    if (verbose) write(6,1000) t_mv, t_diag, t_ortho, t_tot
!
    if (.not.evec_dev) then
      call chk(ctx, dla_download(ctx, c_loc(evec), evd, nbytes(n,n_max)), 'download of evec')
      call chk(ctx, dla_free(ctx, evd), 'free')
    end if
    do cur = 1, 2
      call chk(ctx, dla_free(ctx, sp(cur)), 'free')
      call chk(ctx, dla_free(ctx, asp(cur)), 'free')
      if (gen_eig) call chk(ctx, dla_free(ctx, bsp(cur)), 'free')
    end do
    call chk(ctx, dla_free(ctx, r), 'free')
    deallocate (a_red, e_red, done, skip, r_norm)
    return
!
  contains
!
    subroutine select_panels()
!     the basis is read from copy cur; the new X block (x_new, ax_new, bx_new) is written into copy nxt
      space  = sp(cur)
      aspace = asp(cur)
      bspace = bsp(cur)
      x_new  = sp(nxt)
      ax_new = asp(nxt)
      bx_new = bsp(nxt)
    end subroutine select_panels
!
    subroutine swap_panels()
      integer :: tmp
      tmp = cur
      cur = nxt
      nxt = tmp
      call select_panels()
    end subroutine swap_panels
  end subroutine lobpcg_driver
!
! ---------------------------------------------------------------------------------------
! public orthogonalisation routines on HOST arrays (reference signatures); each uploads,
! runs the device path and downloads.  The drivers never use these wrappers.
! ---------------------------------------------------------------------------------------
  subroutine ortho_cd(n,m,u,growth,ok)
    integer,                          intent(in)    :: n, m
    real(dp), dimension(n,m), target, intent(inout) :: u
    real(dp),                         intent(inout) :: growth
    logical,                          intent(inout) :: ok
    type(c_ptr)    :: ctx, ud
    integer(c_int) :: iok
    ctx = dla_default_ctx()
    call chk(ctx, dla_alloc(ctx, nbytes(n,m), ud), 'allocation')
    call chk(ctx, dla_upload(ctx, ud, c_loc(u), nbytes(n,m)), 'upload')
    call chk(ctx, dla_ortho_cd(ctx, n, m, ud, growth, iok), 'ortho_cd')
    ok = iok .ne. 0
    call chk(ctx, dla_download(ctx, c_loc(u), ud, nbytes(n,m)), 'download')
    call chk(ctx, dla_free(ctx, ud), 'free')
  end subroutine ortho_cd
!
  subroutine ortho_vs_x(n,m,k,x,u,ax,au)
    integer,                          intent(in)    :: n, m, k
    real(dp), dimension(n,m), target, intent(in)    :: x
    real(dp), dimension(n,k), target, intent(inout) :: u
    real(dp), dimension(*)                          :: ax, au     ! dead in the reference too (SURVEY App. B 5)
    type(c_ptr) :: ctx, xd, ud
    ctx = dla_default_ctx()
    call chk(ctx, dla_alloc(ctx, nbytes(n,max(m,1)), xd), 'allocation')
    call chk(ctx, dla_alloc(ctx, nbytes(n,k), ud), 'allocation')
    if (m.gt.0) call chk(ctx, dla_upload(ctx, xd, c_loc(x), nbytes(n,m)), 'upload')
    call chk(ctx, dla_upload(ctx, ud, c_loc(u), nbytes(n,k)), 'upload')
    call chk(ctx, dla_ortho_vs_x(ctx, n, m, k, xd, ud), 'ortho_vs_x')
    call chk(ctx, dla_download(ctx, c_loc(u), ud, nbytes(n,k)), 'download')
    call chk(ctx, dla_free(ctx, xd), 'free')
    call chk(ctx, dla_free(ctx, ud), 'free')
  end subroutine ortho_vs_x
!
  subroutine ortho(n,m,u,w)
    integer,                          intent(in)    :: n, m
    real(dp), dimension(n,m), target, intent(inout) :: u
    real(dp), dimension(*)                          :: w          ! never touched by the reference either
    real(dp) :: growth
    logical  :: ok
!   the reference orthonormalises by Householder QR here; the device path offers the
!   Cholesky-QR family only, which spans the same space with the same orthonormality
    call ortho_cd(n,m,u,growth,ok)
  end subroutine ortho
!
  subroutine b_ortho(n,m,u,bu)
    integer,                          intent(in)    :: n, m
    real(dp), dimension(n,m), target, intent(inout) :: u, bu
    type(c_ptr) :: ctx, ud, bd
    ctx = dla_default_ctx()
    call chk(ctx, dla_alloc(ctx, nbytes(n,m), ud), 'allocation')
    call chk(ctx, dla_alloc(ctx, nbytes(n,m), bd), 'allocation')
    call chk(ctx, dla_upload(ctx, ud, c_loc(u), nbytes(n,m)), 'upload')
    call chk(ctx, dla_upload(ctx, bd, c_loc(bu), nbytes(n,m)), 'upload')
    call chk(ctx, dla_b_ortho(ctx, n, m, ud, bd), 'b_ortho')
    call chk(ctx, dla_download(ctx, c_loc(u), ud, nbytes(n,m)), 'download')
    call chk(ctx, dla_download(ctx, c_loc(bu), bd, nbytes(n,m)), 'download')
    call chk(ctx, dla_free(ctx, ud), 'free')
    call chk(ctx, dla_free(ctx, bd), 'free')
  end subroutine b_ortho
!
  subroutine b_ortho_vs_x(n,m,k,x,bx,u)
    integer,                          intent(in)    :: n, m, k
    real(dp), dimension(n,m), target, intent(in)    :: x, bx
    real(dp), dimension(n,k), target, intent(inout) :: u
    type(c_ptr) :: ctx, xd, bd, ud
    ctx = dla_default_ctx()
    call chk(ctx, dla_alloc(ctx, nbytes(n,max(m,1)), xd), 'allocation')
    call chk(ctx, dla_alloc(ctx, nbytes(n,max(m,1)), bd), 'allocation')
    call chk(ctx, dla_alloc(ctx, nbytes(n,k), ud), 'allocation')
    if (m.gt.0) then
      call chk(ctx, dla_upload(ctx, xd, c_loc(x), nbytes(n,m)), 'upload')
      call chk(ctx, dla_upload(ctx, bd, c_loc(bx), nbytes(n,m)), 'upload')
    end if
    call chk(ctx, dla_upload(ctx, ud, c_loc(u), nbytes(n,k)), 'upload')
    call chk(ctx, dla_b_ortho_vs_x(ctx, n, m, k, xd, bd, ud), 'b_ortho_vs_x')
    call chk(ctx, dla_download(ctx, c_loc(u), ud, nbytes(n,k)), 'download')
    call chk(ctx, dla_free(ctx, xd), 'free')
    call chk(ctx, dla_free(ctx, bd), 'free')
    call chk(ctx, dla_free(ctx, ud), 'free')
  end subroutine b_ortho_vs_x
!
! ---------------------------------------------------------------------------------------
! caslr_eff_driver: linear-response generalised eigenproblem
!
!   / A  B \ / Y \     /  S  D \ / Y \
!   |      | |   | = w |       | |   |     (reference diaglib.f90:1024-1481)
!   \ B  A / \ Z /     \ -D -S / \ Z /
!
! solved in the symmetric/antisymmetric combinations b+ = Y+Z, b- = Y-Z with the Casida matrix as
! the metric: the expansion spaces vp, vm are (A+B)- and (A-B)-orthonormal, the reduced problem is
! s^T s u+ = (1/w)^2 u+ with s = vm^T (S+D) vp, and u- = w s u+ (reference :1052-1060).
! Device-resident: vp, vm, their images under (A+B), (A-B), (S-D), (S+D), and the residuals.
! Deviations from the reference's operation order, none of which changes a result beyond rounding:
!   - s is extended by its new block row and block column instead of being recomputed (:1289);
!   - the Ritz vectors (:1324-1333) are formed when they are needed (convergence, restart, exit),
!     not in every iteration.
! ---------------------------------------------------------------------------------------
  subroutine caslr_eff_driver(verbose,n,n2,n_targ,n_max,max_iter,tol,max_dav, &
                              apbmul,ambmul,spdmul,smdmul,lrprec,eig,evec,ok)
    logical,                               intent(in)    :: verbose
    integer,                               intent(in)    :: n, n2, n_targ, n_max
    integer,                               intent(in)    :: max_iter, max_dav
    real(dp),                              intent(in)    :: tol
    real(dp), dimension(n_max),            intent(inout) :: eig
    real(dp), dimension(n2,n_max), target, intent(inout) :: evec
    logical,                               intent(inout) :: ok
    external                                             :: apbmul, ambmul, spdmul, smdmul, lrprec
!
    type(c_ptr)    :: ctx, vp, vm, lvp, lvm, bvp, bvm, rp, rm, bp, bm, tp, tm, evd
    type(c_funptr) :: f_apb, f_amb, f_spd, f_smd, f_prec
    integer        :: dim_dav, lda, n_act, ind, i_beg, m_dim, ldu, n_frozen, it, i_eig, j, n_mv, n_restarts
    logical        :: evec_dev, have_evec
    real(dp)       :: tol_rms, tol_max, sqrt2
    logical,        allocatable :: done(:)
    integer(c_int), allocatable :: skip(:)
    real(dp),       allocatable :: smat(:,:), s_copy(:,:), e_red(:), up(:,:), um(:,:), ident(:,:)
    real(dp),       allocatable :: r_norm(:,:), rn_p(:,:), rn_m(:,:)
    integer(c_int) :: info
!
    ctx    = dla_default_ctx()
    f_apb  = c_funloc(apbmul)
    f_amb  = c_funloc(ambmul)
    f_spd  = c_funloc(spdmul)
    f_smd  = c_funloc(smdmul)
    f_prec = c_funloc(lrprec)
    evec_dev = dla_get_option(ctx, opt_evec_dev) .ne. 0
!
    dim_dav = max(min_dav,max_dav)
    lda     = dim_dav*n_max
    sqrt2   = sqrt(2.0_dp)
!
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), vp),  'allocation of vp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), vm),  'allocation of vm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), lvp), 'allocation of lvp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), lvm), 'allocation of lvm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), bvp), 'allocation of bvp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), bvm), 'allocation of bvm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), rp), 'allocation of rp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), rm), 'allocation of rm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), bp), 'allocation of bp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), bm), 'allocation of bm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), tp), 'allocation of tp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), tm), 'allocation of tm')
    if (evec_dev) then
      evd = c_loc(evec)
    else
      call chk(ctx, dla_alloc(ctx, nbytes(n2,n_max), evd), 'allocation of evec')
      call chk(ctx, dla_upload(ctx, evd, c_loc(evec), nbytes(n2,n_max)), 'upload of the guess')
    end if
    allocate (done(n_max), skip(n_max), r_norm(2,n_max), rn_p(2,n_max), rn_m(2,n_max))
    allocate (smat(lda,lda), s_copy(lda,lda), e_red(lda), up(lda,n_max), um(lda,n_max), ident(n_max,n_max))
!
    tol_rms = tol
    tol_max = ten * tol
    t_diag  = zero
    t_ortho = zero
    t_mv    = zero
    t_tot   = zero
    smat    = zero
    r_norm  = zero
    rn_p    = zero
    rn_m    = zero
    ident   = zero
    do j = 1, n_max
      ident(j,j) = one
    end do
    ok      = .false.
    done    = .false.
    n_mv    = 0
    n_restarts = 0
!
    call get_time(t_tot)
!
!   the guess in the plus/minus combinations, orthonormal in the metric (reference :1249-1259)
!
    call split_evec()
    call new_metric_blocks(1, n_max)
!
    n_act = n_max
    ind   = 1
    i_beg = 1
    m_dim = 1
    ldu   = 0
    n_frozen = 0
    have_evec = .false.
!
    1030 format(t5,'Davidson-Liu iterations (tol=',d10.2,'):',/, &
                t5,'------------------------------------------------------------------',/, &
                t7,'  iter  root              eigenvalue','         rms         max ok',/, &
                t5,'------------------------------------------------------------------')
    1040 format(t9,i4,2x,i4,f24.12,2d12.4,l3)
    if (verbose) write(6,1030) tol
!
    do it = 1, max_iter
      ldu = ldu + n_act
      have_evec = .false.
!
!     (S+D) times the new vp block, (S-D) times the new vm block (reference :1281-1282)
!
      call get_time(t1)
      call chk(ctx, dla_call_matvec(ctx, f_spd, n, n_act, colp(vp,n,i_beg), colp(bvm,n,i_beg)), 'spdmul')
      call chk(ctx, dla_call_matvec(ctx, f_smd, n, n_act, colp(vm,n,i_beg), colp(bvp,n,i_beg)), 'smdmul')
      call get_time(t2)
      t_mv = t_mv + t2 - t1
      n_mv = n_mv + 2*n_act
!
!     s = vm^T bvm (reference :1289): the new block column and the new block row
!
      call chk(ctx, dla_gram(ctx, n, ldu, vm, n_act, colp(bvm,n,i_beg), smat(1,i_beg), lda), 'reduced matrix')
      if (i_beg.gt.1) call chk(ctx, dla_gram(ctx, n, n_act, colp(vm,n,i_beg), i_beg-1, bvm, smat(i_beg,1), lda), &
                               'reduced matrix')
!
!     s^T s and its largest eigenpairs (reference :1293-1311): the lowest ones of -s^T s
!
      s_copy(1:ldu,1:ldu) = -matmul(transpose(smat(1:ldu,1:ldu)), smat(1:ldu,1:ldu))
      call get_time(t1)
      info = dla_syev_lowest('u', ldu, s_copy, lda, e_red, n_max)
      call get_time(t2)
      t_diag = t_diag + t2 - t1
      if (info.ne.0) then
        write(6,'(t3,a,i6)') 'dsyev failed. info = ',info
        stop
      end if
      do i_eig = 1, n_max
        eig(i_eig)      = sqrt(-e_red(i_eig))
        up(1:ldu,i_eig) = s_copy(1:ldu,i_eig)
      end do
!
!     u- = s u+ / eig (reference :1315-1318)
!
      um(1:ldu,:) = matmul(smat(1:ldu,1:ldu), up(1:ldu,:))
      do i_eig = 1, n_max
        um(1:ldu,i_eig) = um(1:ldu,i_eig)/eig(i_eig)
      end do
!
!     residuals rp = bvp u- - eig lvp u+, rm = bvm u+ - eig lvm u- and their norms (reference :1337-1353)
!
      do i_eig = 1, n_max
        skip(i_eig) = merge(1_c_int, 0_c_int, done(i_eig))
      end do
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, lvp, n_max, up, lda, bp), 'bp')
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, bvp, n_max, um, lda, tp), 'rp')
      call chk(ctx, dla_ritz_residual(ctx, n, n_max, n_max, bp, tp, ident, n_max, eig, n_targ, skip, &
                                      tm, rp, c_null_ptr, rn_p), 'residual')
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, lvm, n_max, um, lda, bm), 'bm')
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, bvm, n_max, up, lda, tp), 'rm')
      call chk(ctx, dla_ritz_residual(ctx, n, n_max, n_max, bm, tp, ident, n_max, eig, n_targ, skip, &
                                      tm, rm, c_null_ptr, rn_m), 'residual')
      do i_eig = 1, n_targ
        if (done(i_eig)) cycle
        r_norm(1,i_eig) = (rn_p(1,i_eig) + rn_m(1,i_eig))/(eig(i_eig)*sqrt2)
        r_norm(2,i_eig) = (rn_p(2,i_eig) + rn_m(2,i_eig))/(sqrt2*eig(i_eig))
      end do
!
!     lock the leading converged roots (reference :1358-1367)
!
      do i_eig = 1, n_targ
        if (done(i_eig)) cycle
        done(i_eig) = r_norm(1,i_eig).lt.tol_rms .and. r_norm(2,i_eig).lt.tol_max .and. it.gt.1
        if (.not.done(i_eig)) then
          done(i_eig+1:n_max) = .false.
          exit
        end if
      end do
!
      if (verbose) then
        do i_eig = 1, n_targ
          write(6,1040) it, i_eig, one/eig(i_eig), r_norm(:,i_eig), done(i_eig)
        end do
        write(6,*)
      end if
!
      if (all(done(1:n_targ))) then
        ok = .true.
        call merge_evec()
        do i_eig = 1, n_targ
          eig(i_eig) = one/eig(i_eig)
        end do
        exit
      end if
!
      if (m_dim .lt. dim_dav) then
!
!       expand both spaces with the preconditioned residuals (reference :1397-1424)
!
        m_dim = m_dim + 1
        i_beg = i_beg + n_act
        n_act = n_max
        n_frozen = 0
        do i_eig = 1, n_targ
          if (done(i_eig)) then
            n_act = n_act - 1
            n_frozen = n_frozen + 1
          else
            exit
          end if
        end do
        ind = n_max - n_act + 1
        call chk(ctx, dla_call_lrprec(ctx, f_prec, n, n_act, eig(ind), colp(rp,n,ind), colp(rm,n,ind), &
                                      colp(vp,n,i_beg), colp(vm,n,i_beg)), 'lrprec')
        call get_time(t1)
        call chk(ctx, dla_b_ortho_vs_x(ctx, n, ldu, n_act, vp, lvp, colp(vp,n,i_beg)), 'b_ortho_vs_x')
        call chk(ctx, dla_b_ortho_vs_x(ctx, n, ldu, n_act, vm, lvm, colp(vm,n,i_beg)), 'b_ortho_vs_x')
        call new_metric_blocks(i_beg, n_act)
        call get_time(t2)
        t_ortho = t_ortho + t2 - t1
      else
!
!       restart from the current Ritz vectors (reference :1426-1463)
!
        if (verbose) write(6,'(t7,a)') 'Restarting davidson.'
        n_restarts = n_restarts + 1
        call merge_evec()
        ldu   = 0
        i_beg = 1
        m_dim = 1
        n_act = n_max
        call split_evec()
        call new_metric_blocks(1, n_max)
        smat = zero
      end if
      if (verbose) write(6,1050) n_targ, n_act, n_frozen
    end do
!
!   evec holds the current approximation on every exit, like the reference (:1330-1333)
!
    if (.not.have_evec) call merge_evec()
    call get_time(t2)
    t_tot = t2 - t_tot
    call dla_set_solve_info(int(min(it,max_iter),c_int), int(n_mv,c_int), int(n_restarts,c_int))
!
    1000 format(t3,'timings for caslr_eff (cpu/wall):   ',/, &
                t3,'  matrix-vector multiplications: ',2f12.4,/, &
                t3,'  diagonalization:               ',2f12.4,/, &
                t3,'  orthogonalization:             ',2f12.4,/, &
                t3,'                                 ',24('='),/,  &
                t3,'  total:                         ',2f12.4)
    if (verbose) write(6,1000) t_mv, t_diag, t_ortho, t_tot
!
    if (.not.evec_dev) then
      call chk(ctx, dla_download(ctx, c_loc(evec), evd, nbytes(n2,n_max)), 'download of evec')
      call chk(ctx, dla_free(ctx, evd), 'free')
    end if
    call chk(ctx, dla_free(ctx, vp), 'free')
    call chk(ctx, dla_free(ctx, vm), 'free')
    call chk(ctx, dla_free(ctx, lvp), 'free')
    call chk(ctx, dla_free(ctx, lvm), 'free')
    call chk(ctx, dla_free(ctx, bvp), 'free')
    call chk(ctx, dla_free(ctx, bvm), 'free')
    call chk(ctx, dla_free(ctx, rp), 'free')
    call chk(ctx, dla_free(ctx, rm), 'free')
    call chk(ctx, dla_free(ctx, bp), 'free')
    call chk(ctx, dla_free(ctx, bm), 'free')
    call chk(ctx, dla_free(ctx, tp), 'free')
    call chk(ctx, dla_free(ctx, tm), 'free')
    deallocate (done, skip, r_norm, rn_p, rn_m, smat, s_copy, e_red, up, um, ident)
!
    1050 format(t5,'----------------------------------------',/,&
                t7,'# target vectors:    ',i4,/,&
                t7,'# new vectors added: ',i4,/,&
                t7,'# converged vectors: ',i4,/,&
                t5,'----------------------------------------')
    return
!
  contains
!
!   device address of the upper (half = 0) or lower (half = 1) n rows of column j of the n2 x n_max block
!
    function halfp(j, half) result(p)
      integer, intent(in) :: j, half
      type(c_ptr)         :: p
      integer(c_intptr_t) :: a
      a = transfer(evd, a) + 8_c_intptr_t * (int(n2,c_intptr_t) * int(j-1,c_intptr_t) + int(half*n,c_intptr_t))
      p = transfer(a, p)
    end function halfp
!
!   vp = Y + Z, vm = Y - Z for the n_max columns of evec (reference :1249-1252, 1443-1446)
!
    subroutine split_evec()
      integer :: jj
      do jj = 1, n_max
        call chk(ctx, dla_copy(ctx, colp(vp,n,jj), halfp(jj,0), nbytes(n,1)), 'copy')
        call chk(ctx, dla_axpy(ctx, int(n,c_size_t), one, halfp(jj,1), colp(vp,n,jj)), 'axpy')
        call chk(ctx, dla_copy(ctx, colp(vm,n,jj), halfp(jj,0), nbytes(n,1)), 'copy')
        call chk(ctx, dla_axpy(ctx, int(n,c_size_t), -one, halfp(jj,1), colp(vm,n,jj)), 'axpy')
      end do
    end subroutine split_evec
!
!   Ritz vectors in the plus/minus combinations, eigp = vp u+, eigm = vm u- (reference :1324-1325; bp, bm
!   are free here and hold them), then Y = eigp + eigm, Z = eigp - eigm (:1330-1333)
!
    subroutine merge_evec()
      integer :: jj
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, vp, n_max, up, lda, bp), 'ritz vectors')
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, vm, n_max, um, lda, bm), 'ritz vectors')
      do jj = 1, n_max
        call chk(ctx, dla_copy(ctx, halfp(jj,0), colp(bp,n,jj), nbytes(n,1)), 'copy')
        call chk(ctx, dla_axpy(ctx, int(n,c_size_t), one, colp(bm,n,jj), halfp(jj,0)), 'axpy')
        call chk(ctx, dla_copy(ctx, halfp(jj,1), colp(bp,n,jj), nbytes(n,1)), 'copy')
        call chk(ctx, dla_axpy(ctx, int(n,c_size_t), -one, colp(bm,n,jj), halfp(jj,1)), 'axpy')
      end do
      have_evec = .true.
    end subroutine merge_evec
!
!   (A+B) vp and (A-B) vm for a new block, which is then made orthonormal in its metric
!   (reference :1256-1259, 1420-1424)
!
    subroutine new_metric_blocks(c0, k)
      integer, intent(in) :: c0, k
      call chk(ctx, dla_call_matvec(ctx, f_apb, n, k, colp(vp,n,c0), colp(lvp,n,c0)), 'apbmul')
      call chk(ctx, dla_b_ortho(ctx, n, k, colp(vp,n,c0), colp(lvp,n,c0)), 'b_ortho')
      call chk(ctx, dla_call_matvec(ctx, f_amb, n, k, colp(vm,n,c0), colp(lvm,n,c0)), 'ambmul')
      call chk(ctx, dla_b_ortho(ctx, n, k, colp(vm,n,c0), colp(lvm,n,c0)), 'b_ortho')
      n_mv = n_mv + 2*k
    end subroutine new_metric_blocks
  end subroutine caslr_eff_driver
!
! ---------------------------------------------------------------------------------------
! caslr_driver: the traditional solver of the same linear-response problem (reference
! diaglib.f90:558-1022, its default algorithm i_alg = 0).  vp, vm are Euclidean-orthonormal
! (ortho_cd / ortho_vs_x); the reduced problem is the 2 ldu-dimensional generalised one
!
!     /  0   s^T \ / u+ \   1  / E+  0  \ / u+ \          E+ = vp^T (A+B) vp,  E- = vm^T (A-B) vm,
!     |          | |    | = -  |        | |    |          s  = vm^T (S+D) vp
!     \  s    0  / \ u- /   w  \ 0   E- / \ u- /
!
! (dsygv itype 1 at :783).  Here the block-diagonal metric is factored blockwise, E+ = L+ L+^T, E- = L- L-^T,
! which turns the pencil into the symmetric matrix (0 M^T; M 0), M = L-^-1 s L+^-T, whose largest eigenpairs come
! from the partial solver; u = L^-T y has the dsygv normalisation u^T diag(E+,E-) u = 1.  As in caslr_eff_driver
! the reduced matrices grow by their new block rows/columns (:754-756 recompute them), and the Ritz vectors
! (:862-868) are formed at convergence, restart and exit.
! ---------------------------------------------------------------------------------------
  subroutine caslr_driver(verbose,n,n2,n_targ,n_max,max_iter,tol,max_dav, &
                          apbmul,ambmul,spdmul,smdmul,lrprec,eig,evec,ok)
    logical,                               intent(in)    :: verbose
    integer,                               intent(in)    :: n, n2, n_targ, n_max
    integer,                               intent(in)    :: max_iter, max_dav
    real(dp),                              intent(in)    :: tol
    real(dp), dimension(n_max),            intent(inout) :: eig
    real(dp), dimension(n2,n_max), target, intent(inout) :: evec
    logical,                               intent(inout) :: ok
    external                                             :: apbmul, ambmul, spdmul, smdmul, lrprec
!
    type(c_ptr)    :: ctx, vp, vm, lvp, lvm, bvp, bvm, rp, rm, bp, bm, tp, tm, evd
    type(c_funptr) :: f_apb, f_amb, f_spd, f_smd, f_prec
    integer        :: dim_dav, lda, n_act, ind, i_beg, m_dim, ldu, n_frozen, it, i_eig, j, n_mv, n_restarts
    logical        :: evec_dev, have_evec
    real(dp)       :: tol_rms, tol_max, growth
    logical,        allocatable :: done(:)
    integer(c_int), allocatable :: skip(:)
    real(dp),       allocatable :: epmat(:,:), emmat(:,:), smat(:,:), up(:,:), um(:,:), ident(:,:)
    real(dp),       allocatable :: r_norm(:,:), rn_p(:,:), rn_m(:,:)
    integer(c_int) :: okc
!
    ctx    = dla_default_ctx()
    f_apb  = c_funloc(apbmul)
    f_amb  = c_funloc(ambmul)
    f_spd  = c_funloc(spdmul)
    f_smd  = c_funloc(smdmul)
    f_prec = c_funloc(lrprec)
    evec_dev = dla_get_option(ctx, opt_evec_dev) .ne. 0
!
    dim_dav = max(min_dav,max_dav)
    lda     = dim_dav*n_max
!
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), vp),  'allocation of vp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), vm),  'allocation of vm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), lvp), 'allocation of lvp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), lvm), 'allocation of lvm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), bvp), 'allocation of bvp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,lda), bvm), 'allocation of bvm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), rp), 'allocation of rp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), rm), 'allocation of rm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), bp), 'allocation of bp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), bm), 'allocation of bm')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), tp), 'allocation of tp')
    call chk(ctx, dla_alloc(ctx, nbytes(n,n_max), tm), 'allocation of tm')
    if (evec_dev) then
      evd = c_loc(evec)
    else
      call chk(ctx, dla_alloc(ctx, nbytes(n2,n_max), evd), 'allocation of evec')
      call chk(ctx, dla_upload(ctx, evd, c_loc(evec), nbytes(n2,n_max)), 'upload of the guess')
    end if
    allocate (done(n_max), skip(n_max), r_norm(2,n_max), rn_p(2,n_max), rn_m(2,n_max))
    allocate (epmat(lda,lda), emmat(lda,lda), smat(lda,lda), up(lda,n_max), um(lda,n_max), ident(n_max,n_max))
!
    tol_rms = tol
    tol_max = ten * tol
    t_diag  = zero
    t_ortho = zero
    t_mv    = zero
    t_tot   = zero
    epmat   = zero
    emmat   = zero
    smat    = zero
    r_norm  = zero
    rn_p    = zero
    rn_m    = zero
    ident   = zero
    do j = 1, n_max
      ident(j,j) = one
    end do
    ok      = .false.
    done    = .false.
    n_mv    = 0
    n_restarts = 0
!
    call get_time(t_tot)
!
!   the guess in the plus/minus combinations, orthonormalised (reference :709-716)
!
    call lr_split_evec(ctx, n, n2, n_max, evd, vp, vm)
    call chk(ctx, dla_ortho_cd(ctx, n, n_max, vp, growth, okc), 'ortho_cd')
    call chk(ctx, dla_ortho_cd(ctx, n, n_max, vm, growth, okc), 'ortho_cd')
!
    n_act = n_max
    ind   = 1
    i_beg = 1
    m_dim = 1
    ldu   = 0
    n_frozen = 0
    have_evec = .false.
!
    1030 format(t5,'Davidson-Liu iterations (tol=',d10.2,'):',/, &
                t5,'------------------------------------------------------------------',/, &
                t7,'  iter  root              eigenvalue','         rms         max ok',/, &
                t5,'------------------------------------------------------------------')
    1040 format(t9,i4,2x,i4,f24.12,2d12.4,l3)
    if (verbose) write(6,1030) tol
!
    do it = 1, max_iter
      ldu = ldu + n_act
      have_evec = .false.
!
!     the four products on the new blocks (reference :743-746)
!
      call get_time(t1)
      call chk(ctx, dla_call_matvec(ctx, f_apb, n, n_act, colp(vp,n,i_beg), colp(lvp,n,i_beg)), 'apbmul')
      call chk(ctx, dla_call_matvec(ctx, f_amb, n, n_act, colp(vm,n,i_beg), colp(lvm,n,i_beg)), 'ambmul')
      call chk(ctx, dla_call_matvec(ctx, f_spd, n, n_act, colp(vp,n,i_beg), colp(bvm,n,i_beg)), 'spdmul')
      call chk(ctx, dla_call_matvec(ctx, f_smd, n, n_act, colp(vm,n,i_beg), colp(bvp,n,i_beg)), 'smdmul')
      call get_time(t2)
      t_mv = t_mv + t2 - t1
      n_mv = n_mv + 4*n_act
!
!     reduced matrices (reference :754-756): new block columns, and for the non-symmetric s also the new block row
!
      call chk(ctx, dla_gram(ctx, n, ldu, vp, n_act, colp(lvp,n,i_beg), epmat(1,i_beg), lda), 'reduced matrix')
      call chk(ctx, dla_gram(ctx, n, ldu, vm, n_act, colp(lvm,n,i_beg), emmat(1,i_beg), lda), 'reduced matrix')
      call chk(ctx, dla_gram(ctx, n, ldu, vm, n_act, colp(bvm,n,i_beg), smat(1,i_beg), lda), 'reduced matrix')
      if (i_beg.gt.1) then
        epmat(i_beg:ldu,1:i_beg-1) = transpose(epmat(1:i_beg-1,i_beg:ldu))
        emmat(i_beg:ldu,1:i_beg-1) = transpose(emmat(1:i_beg-1,i_beg:ldu))
        call chk(ctx, dla_gram(ctx, n, n_act, colp(vm,n,i_beg), i_beg-1, bvm, smat(i_beg,1), lda), 'reduced matrix')
      end if
!
!     largest eigenpairs of the reduced pencil (reference :775-800)
!
      call get_time(t1)
      call lr_reduced_pairs(ldu, lda, n_max, epmat, emmat, smat, eig, up, um)
      call get_time(t2)
      t_diag = t_diag + t2 - t1
!
!     residuals rp = lvp u+ - eig bvp u-, rm = lvm u- - eig bvm u+ and their norms (reference :872-889)
!
      do i_eig = 1, n_max
        skip(i_eig) = merge(1_c_int, 0_c_int, done(i_eig))
      end do
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, bvp, n_max, um, lda, bp), 'bp')
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, lvp, n_max, up, lda, tp), 'rp')
      call chk(ctx, dla_ritz_residual(ctx, n, n_max, n_max, bp, tp, ident, n_max, eig, n_targ, skip, &
                                      tm, rp, c_null_ptr, rn_p), 'residual')
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, bvm, n_max, up, lda, bm), 'bm')
      call chk(ctx, dla_panel_gemm(ctx, n, ldu, lvm, n_max, um, lda, tp), 'rm')
      call chk(ctx, dla_ritz_residual(ctx, n, n_max, n_max, bm, tp, ident, n_max, eig, n_targ, skip, &
                                      tm, rm, c_null_ptr, rn_m), 'residual')
      do i_eig = 1, n_targ
        if (done(i_eig)) cycle
        r_norm(1,i_eig) = rn_p(1,i_eig) + rn_m(1,i_eig)
        r_norm(2,i_eig) = rn_p(2,i_eig) + rn_m(2,i_eig)
      end do
!
!     lock the leading converged roots (reference :894-903)
!
      do i_eig = 1, n_targ
        if (done(i_eig)) cycle
        done(i_eig) = r_norm(1,i_eig).lt.tol_rms .and. r_norm(2,i_eig).lt.tol_max .and. it.gt.1
        if (.not.done(i_eig)) then
          done(i_eig+1:n_max) = .false.
          exit
        end if
      end do
!
      if (verbose) then
        do i_eig = 1, n_targ
          write(6,1040) it, i_eig, eig(i_eig), r_norm(:,i_eig), done(i_eig)
        end do
        write(6,*)
      end if
!
      if (all(done(1:n_targ))) then
        ok = .true.
        exit
      end if
!
      if (m_dim .lt. dim_dav) then
!
!       expand both spaces with the preconditioned residuals (reference :928-955)
!
        m_dim = m_dim + 1
        i_beg = i_beg + n_act
        n_act = n_max
        n_frozen = 0
        do i_eig = 1, n_targ
          if (done(i_eig)) then
            n_act = n_act - 1
            n_frozen = n_frozen + 1
          else
            exit
          end if
        end do
        ind = n_max - n_act + 1
        call chk(ctx, dla_call_lrprec(ctx, f_prec, n, n_act, eig(ind), colp(rp,n,ind), colp(rm,n,ind), &
                                      colp(vp,n,i_beg), colp(vm,n,i_beg)), 'lrprec')
        call get_time(t1)
        call chk(ctx, dla_ortho_vs_x(ctx, n, ldu, n_act, vp, colp(vp,n,i_beg)), 'ortho_vs_x')
        call chk(ctx, dla_ortho_vs_x(ctx, n, ldu, n_act, vm, colp(vm,n,i_beg)), 'ortho_vs_x')
        call get_time(t2)
        t_ortho = t_ortho + t2 - t1
      else
!
!       restart from the current Ritz vectors (reference :957-991)
!
        if (verbose) write(6,'(t7,a)') 'Restarting davidson.'
        n_restarts = n_restarts + 1
        call lr_merge_evec(ctx, n, n2, n_max, ldu, lda, vp, vm, up, um, bp, bm, evd)
        ldu   = 0
        i_beg = 1
        m_dim = 1
        n_act = n_max
        call lr_split_evec(ctx, n, n2, n_max, evd, vp, vm)
        call chk(ctx, dla_ortho_cd(ctx, n, n_max, vp, growth, okc), 'ortho_cd')
        call chk(ctx, dla_ortho_cd(ctx, n, n_max, vm, growth, okc), 'ortho_cd')
        epmat = zero
        emmat = zero
        smat  = zero
      end if
      if (verbose) write(6,1050) n_targ, n_act, n_frozen
    end do
!
!   evec holds the current approximation on every exit, like the reference (:865-868)
!
    if (ldu.gt.0) call lr_merge_evec(ctx, n, n2, n_max, ldu, lda, vp, vm, up, um, bp, bm, evd)
    call get_time(t2)
    t_tot = t2 - t_tot
    call dla_set_solve_info(int(min(it,max_iter),c_int), int(n_mv,c_int), int(n_restarts,c_int))
!
    1000 format(t3,'timings for caslr (cpu/wall):   ',/, &
                t3,'  matrix-vector multiplications: ',2f12.4,/, &
                t3,'  diagonalization:               ',2f12.4,/, &
                t3,'  orthogonalization:             ',2f12.4,/, &
                t3,'                                 ',24('='),/,  &
                t3,'  total:                         ',2f12.4)
    if (verbose) write(6,1000) t_mv, t_diag, t_ortho, t_tot
!
    if (.not.evec_dev) then
      call chk(ctx, dla_download(ctx, c_loc(evec), evd, nbytes(n2,n_max)), 'download of evec')
      call chk(ctx, dla_free(ctx, evd), 'free')
    end if
    call chk(ctx, dla_free(ctx, vp), 'free')
    call chk(ctx, dla_free(ctx, vm), 'free')
    call chk(ctx, dla_free(ctx, lvp), 'free')
    call chk(ctx, dla_free(ctx, lvm), 'free')
    call chk(ctx, dla_free(ctx, bvp), 'free')
    call chk(ctx, dla_free(ctx, bvm), 'free')
    call chk(ctx, dla_free(ctx, rp), 'free')
    call chk(ctx, dla_free(ctx, rm), 'free')
    call chk(ctx, dla_free(ctx, bp), 'free')
    call chk(ctx, dla_free(ctx, bm), 'free')
    call chk(ctx, dla_free(ctx, tp), 'free')
    call chk(ctx, dla_free(ctx, tm), 'free')
    deallocate (done, skip, r_norm, rn_p, rn_m, epmat, emmat, smat, up, um, ident)
!
    1050 format(t5,'----------------------------------------',/,&
                t7,'# target vectors:    ',i4,/,&
                t7,'# new vectors added: ',i4,/,&
                t7,'# converged vectors: ',i4,/,&
                t5,'----------------------------------------')
    return
  end subroutine caslr_driver
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
      stop
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
      stop
    end if
    do i = 1, n_max
      eig(i)      = -one/w(i)
      up(1:ldu,i) = matmul(transpose(lp), cc(1:ldu,i))
      um(1:ldu,i) = matmul(transpose(lm), cc(ldu+1:l2,i))
    end do
    deallocate (lp, lm, mm, cc, w)
  end subroutine lr_reduced_pairs
!
! vp = Y + Z, vm = Y - Z for the n_max columns of the 2n x n_max block evd (reference :709-712, 1249-1252)
!
  subroutine lr_split_evec(ctx, n, n2, n_max, evd, vp, vm)
    type(c_ptr), intent(in) :: ctx, evd, vp, vm
    integer,     intent(in) :: n, n2, n_max
    integer :: jj
    do jj = 1, n_max
      call chk(ctx, dla_copy(ctx, colp(vp,n,jj), lr_halfp(evd,n,n2,jj,0), nbytes(n,1)), 'copy')
      call chk(ctx, dla_axpy(ctx, int(n,c_size_t), one, lr_halfp(evd,n,n2,jj,1), colp(vp,n,jj)), 'axpy')
      call chk(ctx, dla_copy(ctx, colp(vm,n,jj), lr_halfp(evd,n,n2,jj,0), nbytes(n,1)), 'copy')
      call chk(ctx, dla_axpy(ctx, int(n,c_size_t), -one, lr_halfp(evd,n,n2,jj,1), colp(vm,n,jj)), 'axpy')
    end do
  end subroutine lr_split_evec
!
! Ritz vectors eigp = vp u+, eigm = vm u- (in the scratch blocks bp, bm), then Y = eigp + eigm, Z = eigp - eigm
! (reference :862-868, 1324-1333)
!
  subroutine lr_merge_evec(ctx, n, n2, n_max, ldu, lda, vp, vm, up, um, bp, bm, evd)
    type(c_ptr), intent(in) :: ctx, vp, vm, bp, bm, evd
    integer,     intent(in) :: n, n2, n_max, ldu, lda
    real(dp),    intent(in) :: up(lda,n_max), um(lda,n_max)
    integer :: jj
    call chk(ctx, dla_panel_gemm(ctx, n, ldu, vp, n_max, up, lda, bp), 'ritz vectors')
    call chk(ctx, dla_panel_gemm(ctx, n, ldu, vm, n_max, um, lda, bm), 'ritz vectors')
    do jj = 1, n_max
      call chk(ctx, dla_copy(ctx, lr_halfp(evd,n,n2,jj,0), colp(bp,n,jj), nbytes(n,1)), 'copy')
      call chk(ctx, dla_axpy(ctx, int(n,c_size_t), one, colp(bm,n,jj), lr_halfp(evd,n,n2,jj,0)), 'axpy')
      call chk(ctx, dla_copy(ctx, lr_halfp(evd,n,n2,jj,1), colp(bp,n,jj), nbytes(n,1)), 'copy')
      call chk(ctx, dla_axpy(ctx, int(n,c_size_t), -one, colp(bm,n,jj), lr_halfp(evd,n,n2,jj,1)), 'axpy')
    end do
  end subroutine lr_merge_evec
!
! device address of the upper (half = 0) or lower (half = 1) n rows of column j of a 2n x m block
!
  function lr_halfp(evd, n, n2, j, half) result(p)
    type(c_ptr), intent(in) :: evd
    integer,     intent(in) :: n, n2, j, half
    type(c_ptr)             :: p
    integer(c_intptr_t)     :: a
    a = transfer(evd, a) + 8_c_intptr_t * (int(n2,c_intptr_t) * int(j-1,c_intptr_t) + int(half*n,c_intptr_t))
    p = transfer(a, p)
  end function lr_halfp
!
end module diaglib
