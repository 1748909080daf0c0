!
! examples/fortran_sparse_caller/sparse_caller.f90 -- a Fortran caller whose matrix is SPARSE and lives on the device.
!
! The adapter pattern for device-resident callbacks (SURVEY 8f row 4): the caller assembles its matrix once in CSR form on
! the host, hands it to the library's sample ELLPACK operator (include/diaglib_amd.h: dla_spmm_setup_csr), switches the
! drivers to device callbacks and passes the operator's entry points -- bind(C) routines with the reference's
! matvec(n,m,x,ax) / precnd(n,m,fac,x,px) shapes that take DEVICE addresses -- exactly where a host matvec would go.
! A caller with kernels of its own (hipfort, OpenMP target) writes its two routines the same way: x and ax arrive as
! device addresses of n x m column-major blocks, and the work is enqueued on dla_stream(dla_default_ctx()).
!
! The matrix is the reference's test matrix made sparse (main.f90:311-317): a_ii = i + 1, a_ij = 1/(i+j) for |i-j| <= 6.
!
program sparse_caller
  use real_precision
  use iso_c_binding
  use diaglib, only : davidson_driver, diaglib_amd_config
  implicit none
  interface
    function dla_default_ctx() bind(C,name='dla_default_ctx') result(ctx)
      import :: c_ptr
      type(c_ptr) :: ctx
    end function
    function dla_spmm_setup_csr(ctx,n,rowptr,colind,values) bind(C,name='dla_spmm_setup_csr') result(st)
      import :: c_ptr, c_int, c_long_long, c_double
      type(c_ptr), value   :: ctx
      integer(c_int), value :: n
      integer(c_long_long) :: rowptr(*)
      integer(c_int)       :: colind(*)
      real(c_double)       :: values(*)
      integer(c_int)       :: st
    end function
    subroutine dla_spmm_matvec(n,m,x,ax) bind(C,name='dla_spmm_matvec')
      import :: c_int, c_double
      integer(c_int) :: n, m
      real(c_double) :: x(*), ax(*)
    end subroutine
    subroutine dla_spmm_precnd(n,m,fac,x,px) bind(C,name='dla_spmm_precnd')
      import :: c_int, c_double
      integer(c_int) :: n, m
      real(c_double) :: fac, x(*), px(*)
    end subroutine
  end interface
  integer, parameter  :: n = 50000, half = 6, n_want = 6, itmax = 300, m_max = 20
  real(dp), parameter :: tol = 1.0e-9_dp
  integer  :: n_eig, i, j, nnz
  logical  :: ok
  integer(c_long_long), allocatable :: rowptr(:)
  integer(c_int),       allocatable :: colind(:)
  real(dp),             allocatable :: values(:), eig(:), evec(:,:), ax(:), res(:)
!
! CSR assembly, 0-based indices as the C interface wants them
!
  allocate (rowptr(n+1), colind(n*(2*half+1)), values(n*(2*half+1)))
  nnz = 0
  rowptr(1) = 0
  do i = 1, n
    do j = max(1,i-half), min(n,i+half)
      nnz = nnz + 1
      colind(nnz) = j - 1
      if (j.eq.i) then
        values(nnz) = real(i+1,dp)
      else
        values(nnz) = 1.0_dp/real(i+j,dp)
      end if
    end do
    rowptr(i+1) = nnz
  end do
  if (dla_spmm_setup_csr(dla_default_ctx(), n, rowptr, colind, values).ne.0) stop 'setup failed'
  call diaglib_amd_config(callbacks_on_device=.true., evec_on_device=.false.)
!
  n_eig = min(2*n_want, n_want+5)
  allocate (eig(n_eig), evec(n,n_eig), ax(n), res(n_want))
  call random_number(evec)
  evec = evec - 0.5_dp
  evec(201:,:) = 1.0e-3_dp*evec(201:,:)
  call davidson_driver(.false.,n,n_want,n_eig,itmax,tol,m_max,0.0_dp,dla_spmm_matvec,dla_spmm_precnd,eig,evec,ok)
  write(6,'(a,l2)') 'SPARSE DAVIDSON ok:', ok
  write(6,'(a,6f14.9)') 'SPARSE DAVIDSON eig:', eig(1:n_want)
!
! the caller's own check: || A x - eig x || from the CSR arrays on the host
!
  do j = 1, n_want
    ax = 0.0_dp
    do i = 1, n
      ax(i) = sum(values(rowptr(i)+1:rowptr(i+1))*evec(colind(rowptr(i)+1:rowptr(i+1))+1,j))
    end do
    res(j) = sqrt(sum((ax - eig(j)*evec(:,j))**2))
  end do
  write(6,'(a,es12.4)') 'SPARSE max residual:', maxval(res)
  call diaglib_amd_config(release_cache=.true.)
end program sparse_caller
