!
! examples/fortran_device_caller/device_caller.f90 -- the opt-in DEVICE mode from a Fortran caller.
!
! Same module, same driver call as examples/fortran_caller; the only additions are
!   call diaglib_amd_config(callbacks_on_device=.true.)
! and callbacks that work on device addresses.  Here they are the library's own benchmark operator
! (include/diaglib_amd.h: dla_synth_setup / dla_synth_matvec / dla_synth_precnd, bind(C), the reference's
! matvec(n,m,x,ax) / precnd(n,m,fac,x,px) shapes); a production caller would pass hipfort or OpenMP-target
! routines of its own.  eig/evec stay host arrays (evec_on_device=.false.), so nothing else changes.
!
program device_caller
  use real_precision
  use iso_c_binding
  use diaglib, only : davidson_driver, lobpcg_driver, diaglib_amd_config, diaglib_amd_timings
  implicit none
  interface
    function dla_default_ctx() bind(C,name='dla_default_ctx') result(ctx)
      import :: c_ptr
      type(c_ptr) :: ctx
    end function
    function dla_synth_setup(ctx,n_global,row0,n_local,rank_w,sigma) bind(C,name='dla_synth_setup') result(st)
      import :: c_ptr, c_int, c_long_long, c_double
      type(c_ptr), value :: ctx
      integer(c_long_long), value :: n_global, row0
      integer(c_int), value :: n_local, rank_w
      real(c_double), value :: sigma
      integer(c_int) :: st
    end function
    subroutine dla_synth_matvec(n,m,x,ax) bind(C,name='dla_synth_matvec')
      import :: c_int, c_double
      integer(c_int) :: n, m
      real(c_double) :: x(*), ax(*)
    end subroutine
    subroutine dla_synth_precnd(n,m,fac,x,px) bind(C,name='dla_synth_precnd')
      import :: c_int, c_double
      integer(c_int) :: n, m
      real(c_double) :: fac, x(*), px(*)
    end subroutine
  end interface
  integer, parameter  :: n = 200000, n_want = 8, itmax = 100, m_max = 20
  real(dp), parameter :: tol = 1.0e-10_dp
  integer  :: n_eig, i
  logical  :: ok
  real(dp), allocatable :: eig(:), evec(:,:)
  real(dp) :: tt(6)
!
  if (dla_synth_setup(dla_default_ctx(), int(n,c_long_long), 0_c_long_long, n, 4, 0.5_dp).ne.0) stop 'setup failed'
  call diaglib_amd_config(callbacks_on_device=.true., evec_on_device=.false.)
  n_eig = min(2*n_want, n_want+5)
  allocate (eig(n_eig), evec(n,n_eig))
  evec = 0.0_dp
  do i = 1, n_eig
    evec(i,i) = 1.0_dp
  end do
  call diaglib_amd_timings(profile=.true.)         ! device time per kind of work (the reference leaves projection / Ritz / residual un-bucketed)
  call davidson_driver(.false.,n,n_want,n_eig,itmax,tol,m_max,0.0_dp,dla_synth_matvec,dla_synth_precnd,eig,evec,ok)
  call diaglib_amd_timings(profile=.false., t_proj=tt(1), t_update=tt(2), t_trmm=tt(3), t_ritz=tt(4), t_matvec=tt(5), t_precnd=tt(6))
  write(6,'(a,6f12.6)') 'DEVICE DAVIDSON seconds (proj update trmm ritz matvec precnd):', tt
  write(6,'(a,l2)') 'DEVICE DAVIDSON ok:', ok
  write(6,'(a,8f14.9)') 'DEVICE DAVIDSON eig:', eig(1:n_want)
  evec = 0.0_dp
  do i = 1, n_eig
    evec(i,i) = 1.0_dp
  end do
  call lobpcg_driver(.false.,.false.,n,n_want,n_eig,itmax,tol,0.0_dp,dla_synth_matvec,dla_synth_precnd, &
                     dla_synth_matvec,eig,evec,ok)
  write(6,'(a,l2)') 'DEVICE LOBPCG ok:', ok
  write(6,'(a,8f14.9)') 'DEVICE LOBPCG eig:', eig(1:n_want)
  write(6,'(a,f14.9)') 'DEVICE |x1|:', sqrt(sum(evec(:,1)**2))
  call diaglib_amd_config(release_cache=.true.)
end program device_caller
