!
! examples/fortran_caller/caller.f90 -- a diaglib CALLER, written against the reference's interface:
! it `use`s module diaglib, supplies host-array matvec/precnd callbacks through a module of its own and
! calls lobpcg_driver and davidson_driver with the reference's argument lists (reference
! diaglib.f90:171-228, 1483-1539; callback shapes README.md:34-35).  Nothing in it knows about GPUs.
! Built against diaglib_amd/fortran/{real_precision,diaglib}.f90 + libdiaglib_amd.so it runs on the
! MI355X path unchanged (tests/test_fortran_caller_gpu.py compiles and runs it).
!
! The matrix is the reference harness' dense symmetric test matrix a_ii = i+1, a_ij = 1/(i+j)
! (reference main.f90:311-317), the preconditioner its diagonal shift-and-invert (main.f90:146-171).
!
module caller_data
  use real_precision
  implicit none
  real(dp), allocatable :: a(:,:)
  integer               :: nmult = 0
end module caller_data
!
subroutine my_matvec(n,m,x,ax)
  use caller_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: ax(n,m)
  nmult = nmult + m
  ax = matmul(a,x)
end subroutine my_matvec
!
subroutine my_precnd(n,m,fac,x,px)
  use caller_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: fac, x(n,m)
  real(dp), intent(inout) :: px(n,m)
  integer :: i, j
  do j = 1, m
    do i = 1, n
      if (abs(a(i,i)+fac).gt.1.0e-5_dp) then
        px(i,j) = x(i,j)/(a(i,i)+fac)
      else
        px(i,j) = x(i,j)
      end if
    end do
  end do
end subroutine my_precnd
!
program caller
  use real_precision
  use caller_data
  use diaglib, only : lobpcg_driver, davidson_driver
  implicit none
  integer, parameter :: n = 1000, n_want = 10, itmax = 100, m_max = 20
  real(dp), parameter :: tol = 1.0e-8_dp
  integer  :: n_eig, i, j
  logical  :: ok
  real(dp), allocatable :: eig(:), evec(:,:)
  external :: my_matvec, my_precnd
!
  allocate (a(n,n))
  do i = 1, n
    a(i,i) = real(i,dp) + 1.0_dp
    do j = 1, i-1
      a(j,i) = 1.0_dp/real(i+j,dp)
      a(i,j) = a(j,i)
    end do
  end do
  n_eig = min(2*n_want, n_want+5)
  allocate (eig(n_eig), evec(n,n_eig))
!
! unit-vector guess on the smallest diagonal entries
!
  evec = 0.0_dp
  do i = 1, n_eig
    evec(i,i) = 1.0_dp
  end do
  call lobpcg_driver(.false.,.false.,n,n_want,n_eig,itmax,tol,0.0_dp,my_matvec,my_precnd,my_matvec,eig,evec,ok)
  write(6,'(a,l2,i6)') 'LOBPCG ok/matvec columns:', ok, nmult
  write(6,'(a,10f14.9)') 'LOBPCG eig:', eig(1:n_want)
!
  evec = 0.0_dp
  do i = 1, n_eig
    evec(i,i) = 1.0_dp
  end do
  nmult = 0
  call davidson_driver(.false.,n,n_want,n_eig,itmax,tol,m_max,0.0_dp,my_matvec,my_precnd,eig,evec,ok)
  write(6,'(a,l2,i6)') 'DAVIDSON ok/matvec columns:', ok, nmult
  write(6,'(a,10f14.9)') 'DAVIDSON eig:', eig(1:n_want)
  write(6,'(a,f14.9)') 'DAVIDSON |x1|:', sqrt(sum(evec(:,1)**2))
end program caller
