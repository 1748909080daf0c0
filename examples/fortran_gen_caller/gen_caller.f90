!
! examples/fortran_gen_caller/gen_caller.f90 -- a diaglib CALLER of the generalised and linear-response drivers,
! written against the reference's interface (reference diaglib.f90:1855-1911 gen_david_driver, :171-228 lobpcg_driver with
! gen_eig = .true., :558-640 caslr_driver; callers main.f90:403-526 test_geneig, :528-730 test_caslr): it `use`s module
! diaglib, supplies host-array matvec / precnd / bvec / lrprec callbacks and knows nothing about GPUs.
!
! The matrices come from a file the caller's owner wrote (tests/test_fortran_caller_gpu.py writes the ones the golden
! fixtures were generated with):  unformatted stream  n, S(n,n)  --  nlr, apb, amb, spd, smd (nlr x nlr each).
! A is the reference harness' dense test matrix a_ii = i+1, a_ij = 1/(i+j) (main.f90:311-317).
!
module gen_data
  use real_precision
  implicit none
  real(dp), allocatable :: a(:,:), s(:,:)
  real(dp), allocatable :: apb(:,:), amb(:,:), spd(:,:), smd(:,:)
  integer               :: n_mult = 0, n_bmult = 0
end module gen_data
!
subroutine g_matvec(n,m,x,ax)
  use gen_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: ax(n,m)
  n_mult = n_mult + m
  ax = matmul(a,x)
end subroutine g_matvec
!
subroutine g_bvec(n,m,x,bx)
  use gen_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: bx(n,m)
  n_bmult = n_bmult + m
  bx = matmul(s,x)
end subroutine g_bvec
!
subroutine g_precnd(n,m,fac,x,px)
  use gen_data
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
end subroutine g_precnd
!
subroutine t_apb(n,m,x,y)
  use gen_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: y(n,m)
  y = matmul(apb,x)
end subroutine t_apb
subroutine t_amb(n,m,x,y)
  use gen_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: y(n,m)
  y = matmul(amb,x)
end subroutine t_amb
subroutine t_spd(n,m,x,y)
  use gen_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: y(n,m)
  y = matmul(spd,x)
end subroutine t_spd
subroutine t_smd(n,m,x,y)
  use gen_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: y(n,m)
  y = matmul(smd,x)
end subroutine t_smd
!
! the harness' preconditioner for the traditional driver (main.f90:234-255)
!
subroutine t_prec(n,m,fac,xp,xm,yp,ym)
  use gen_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: fac, xp(n,m), xm(n,m)
  real(dp), intent(inout) :: yp(n,m), ym(n,m)
  integer  :: i, j
  real(dp) :: aii, sii, den
  do j = 1, m
    do i = 1, n
      aii = 0.5_dp*(apb(i,i) + amb(i,i))
      sii = 0.5_dp*(spd(i,i) + smd(i,i))
      den = -1.0_dp/(aii*aii - fac*fac*sii*sii)
      yp(i,j) = den*(aii*xp(i,j) + fac*sii*xm(i,j))
      ym(i,j) = den*(aii*xm(i,j) + fac*sii*xp(i,j))
    end do
  end do
end subroutine t_prec
!
program gen_caller
  use real_precision
  use gen_data
  use diaglib, only : gen_david_driver, lobpcg_driver, caslr_driver
  implicit none
  integer, parameter  :: n_want = 4, itmax = 200, m_max = 20
  real(dp), parameter :: tol = 1.0e-8_dp
  integer  :: n, nlr, n_eig, i, j, k, u
  logical  :: ok
  real(dp), allocatable :: eig(:), evec(:,:), w(:), xy(:,:), ax(:), sx(:)
  real(dp) :: res, resmax, orth
  external :: g_matvec, g_bvec, g_precnd, t_apb, t_amb, t_spd, t_smd, t_prec
!
  open (newunit=u, file='gen_caller.in', access='stream', form='unformatted', status='old')
  read (u) n
  allocate (a(n,n), s(n,n))
  read (u) s
  read (u) nlr
  allocate (apb(nlr,nlr), amb(nlr,nlr), spd(nlr,nlr), smd(nlr,nlr))
  read (u) apb, amb, spd, smd
  close (u)
  do i = 1, n
    a(i,i) = real(i,dp) + 1.0_dp
    do j = 1, i-1
      a(j,i) = 1.0_dp/real(i+j,dp)
      a(i,j) = a(j,i)
    end do
  end do
  n_eig = min(2*n_want, n_want+5)
  allocate (eig(n_eig), evec(n,n_eig), ax(n), sx(n))
!
! A x = lambda S x, Davidson-Liu with the metric (reference diaglib.f90:1855; caller main.f90:403-526)
!
  evec = 0.0_dp
  do i = 1, n_eig
    evec(i,i) = 1.0_dp
  end do
  call gen_david_driver(.false.,n,n_want,n_eig,itmax,tol,m_max,0.0_dp,g_matvec,g_precnd,g_bvec,eig,evec,ok)
  call report('GEN_DAVIDSON')
!
! the same problem with LOBPCG, gen_eig = .true. (reference diaglib.f90:171)
!
  evec = 0.0_dp
  do i = 1, n_eig
    evec(i,i) = 1.0_dp
  end do
  n_mult = 0
  n_bmult = 0
  call lobpcg_driver(.false.,.true.,n,n_want,n_eig,itmax,tol,0.0_dp,g_matvec,g_precnd,g_bvec,eig,evec,ok)
  call report('GEN_LOBPCG')
!
! linear response with the traditional driver (reference diaglib.f90:558; caller main.f90:528-730)
!
  allocate (w(n_eig), xy(2*nlr,n_eig))
  xy = 0.0_dp
  do i = 1, n_eig
    xy(i,i) = 1.0_dp
  end do
  call caslr_driver(.false.,nlr,2*nlr,n_want,n_eig,100,tol,m_max,t_apb,t_amb,t_spd,t_smd,t_prec,w,xy,ok)
  write(6,'(a,l2)') 'CASLR ok:', ok
  write(6,'(a,4es24.15)') 'CASLR eig:', w(1:n_want)
!
contains
  subroutine report(tag)
    character(len=*), intent(in) :: tag
    resmax = 0.0_dp
    orth = 0.0_dp
    do k = 1, n_want
      ax = matmul(a, evec(:,k))
      sx = matmul(s, evec(:,k))
      res = sqrt(sum((ax - eig(k)*sx)**2))/abs(eig(k))
      resmax = max(resmax, res)
      orth = max(orth, abs(dot_product(evec(:,k), sx) - 1.0_dp))
    end do
    write(6,'(a,a,l2,2i6)') tag, ' ok/matvec/bvec columns:', ok, n_mult, n_bmult
    write(6,'(a,a,4es24.15)') tag, ' eig:', eig(1:n_want)
    write(6,'(a,a,2es12.4)') tag, ' max residual, max |x^T S x - 1|:', resmax, orth
  end subroutine report
end program gen_caller
