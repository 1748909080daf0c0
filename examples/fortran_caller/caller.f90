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
! linear-response problem (same roles as the reference harness, main.f90:528-600)
  real(dp), allocatable :: apb(:,:), amb(:,:), spd(:,:), smd(:,:)
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
subroutine lr_apb(n,m,x,y)
  use caller_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: y(n,m)
  y = matmul(apb,x)
end subroutine lr_apb
subroutine lr_amb(n,m,x,y)
  use caller_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: y(n,m)
  y = matmul(amb,x)
end subroutine lr_amb
subroutine lr_spd(n,m,x,y)
  use caller_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: y(n,m)
  y = matmul(spd,x)
end subroutine lr_spd
subroutine lr_smd(n,m,x,y)
  use caller_data
  implicit none
  integer,  intent(in)    :: n, m
  real(dp), intent(in)    :: x(n,m)
  real(dp), intent(inout) :: y(n,m)
  y = matmul(smd,x)
end subroutine lr_smd
!
! the reference harness' preconditioner for the efficient driver (main.f90:257-281)
!
subroutine lr_prec(n,m,fac,xp,xm,yp,ym)
  use caller_data
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
      den = 1.0_dp/(fac*fac*aii*aii - sii*sii)
      yp(i,j) = den*(fac*aii*xp(i,j) + sii*xm(i,j))
      ym(i,j) = den*(fac*aii*xm(i,j) + sii*xp(i,j))
    end do
  end do
end subroutine lr_prec
!
program caller
  use real_precision
  use caller_data
  use diaglib, only : lobpcg_driver, davidson_driver, caslr_eff_driver
  implicit none
  integer, parameter :: n = 1000, n_want = 10, itmax = 100, m_max = 20
  real(dp), parameter :: tol = 1.0e-8_dp
  integer  :: n_eig, i, j
  logical  :: ok
  real(dp), allocatable :: eig(:), evec(:,:)
  external :: my_matvec, my_precnd, lr_apb, lr_amb, lr_spd, lr_smd, lr_prec
  integer, parameter :: nlr = 300, lr_want = 4
  integer  :: lr_eig, k
  real(dp), allocatable :: w(:), xy(:,:), gmat(:,:), y(:), z(:), lhs(:), rhs(:)
  real(dp) :: res, resmax
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
!
! linear response: (A B; B A)(Y Z) = w (S D; -D -S)(Y Z) with caslr_eff_driver (reference diaglib.f90:1024-1481)
!
  allocate (apb(nlr,nlr), amb(nlr,nlr), spd(nlr,nlr), smd(nlr,nlr), gmat(nlr,8))
  do j = 1, 8
    do i = 1, nlr
      gmat(i,j) = 0.5_dp*sin(0.37_dp*real(i*j,dp) + real(j,dp))
    end do
  end do
  amb = 0.0_dp
  do i = 1, nlr
    do j = 1, nlr
      apb(i,j) = 0.2_dp/real(i+j,dp)
      spd(i,j) = (0.5_dp/8.0_dp)*dot_product(gmat(i,:),gmat(j,:))
      smd(i,j) = spd(i,j)
      if (i.lt.j) then
        spd(i,j) = spd(i,j) + 0.05_dp*sin(real(i,dp) + 2.0_dp*real(j,dp))
        smd(i,j) = smd(i,j) - 0.05_dp*sin(real(i,dp) + 2.0_dp*real(j,dp))
      else if (i.gt.j) then
        spd(i,j) = spd(i,j) - 0.05_dp*sin(real(j,dp) + 2.0_dp*real(i,dp))
        smd(i,j) = smd(i,j) + 0.05_dp*sin(real(j,dp) + 2.0_dp*real(i,dp))
      end if
    end do
    apb(i,i) = 5.0_dp + real(i,dp)
    amb(i,i) = 2.0_dp + real(i,dp)
    spd(i,i) = spd(i,i) + 1.0_dp
    smd(i,i) = smd(i,i) + 1.0_dp
  end do
  lr_eig = min(2*lr_want, lr_want+5)
  allocate (w(lr_eig), xy(2*nlr,lr_eig), y(nlr), z(nlr), lhs(2*nlr), rhs(2*nlr))
  xy = 0.0_dp
  do i = 1, lr_eig
    xy(i,i) = 1.0_dp
  end do
  call caslr_eff_driver(.false.,nlr,2*nlr,lr_want,lr_eig,itmax,tol,m_max,lr_apb,lr_amb,lr_spd,lr_smd,lr_prec,w,xy,ok)
  resmax = 0.0_dp
  do k = 1, lr_want
    y = xy(1:nlr,k)
    z = xy(nlr+1:2*nlr,k)
!   A = (apb+amb)/2, B = (apb-amb)/2, S = (spd+smd)/2, D = (spd-smd)/2
    lhs(1:nlr)       = 0.5_dp*(matmul(apb,y+z) + matmul(amb,y-z))
    lhs(nlr+1:2*nlr) = 0.5_dp*(matmul(apb,y+z) - matmul(amb,y-z))
    rhs(1:nlr)       = 0.5_dp*(matmul(spd,y+z) + matmul(smd,y-z))
    rhs(nlr+1:2*nlr) = 0.5_dp*(matmul(smd,y-z) - matmul(spd,y+z))
    res = sqrt(sum((lhs - w(k)*rhs)**2))/sqrt(sum(lhs**2))
    resmax = max(resmax,res)
  end do
  write(6,'(a,l2)') 'CASLR_EFF ok:', ok
  write(6,'(a,4f14.9)') 'CASLR_EFF eig:', w(1:lr_want)
  write(6,'(a,es12.4)') 'CASLR_EFF max residual:', resmax
end program caller
