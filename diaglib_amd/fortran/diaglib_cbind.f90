!
! diaglib_cbind -- bind(C) twins of the two drivers (include/diaglib_amd.h:
! dla_davidson_driver, dla_lobpcg_driver) so that C, Python (ctypes) and the tests can
! call the Fortran drivers with plain C scalars and function pointers.  Argument meaning
! is the reference's (reference diaglib.f90:1483-1539, 171-228); logicals travel as int.
!
module diaglib_cbind
  use iso_c_binding
  use diaglib, only : davidson_driver, lobpcg_driver, gen_david_driver, caslr_eff_driver, caslr_driver
  implicit none
!
  abstract interface
    subroutine mv_iface(n,m,x,ax) bind(C)
      import :: c_int, c_double
      integer(c_int) :: n, m
      real(c_double) :: x(*), ax(*)
    end subroutine mv_iface
    subroutine pc_iface(n,m,fac,x,px) bind(C)
      import :: c_int, c_double
      integer(c_int) :: n, m
      real(c_double) :: fac
      real(c_double) :: x(*), px(*)
    end subroutine pc_iface
    subroutine lrpc_iface(n,m,fac,xp,xm,yp,ym) bind(C)
      import :: c_int, c_double
      integer(c_int) :: n, m
      real(c_double) :: fac
      real(c_double) :: xp(*), xm(*), yp(*), ym(*)
    end subroutine lrpc_iface
  end interface
!
contains
!
  subroutine dla_davidson_driver(verbose,n,n_targ,n_max,max_iter,tol,max_dav,shift, &
                                 matvec,precnd,eig,evec,ok) bind(C,name='dla_davidson_driver')
    integer(c_int), value :: verbose, n, n_targ, n_max, max_iter, max_dav
    real(c_double), value :: tol, shift
    type(c_funptr), value :: matvec, precnd
    real(c_double)        :: eig(n_max), evec(n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface), pointer :: mv
    procedure(pc_iface), pointer :: pc
    logical :: lok
    call c_f_procpointer(matvec, mv)
    call c_f_procpointer(precnd, pc)
    lok = .false.
    call davidson_driver(verbose.ne.0,n,n_targ,n_max,max_iter,tol,max_dav,shift,mv,pc,eig,evec,lok)
    ok = merge(1_c_int, 0_c_int, lok)
  end subroutine dla_davidson_driver
!
  subroutine dla_gen_david_driver(verbose,n,n_targ,n_max,max_iter,tol,max_dav,shift, &
                                  matvec,precnd,bvec,eig,evec,ok) bind(C,name='dla_gen_david_driver')
    integer(c_int), value :: verbose, n, n_targ, n_max, max_iter, max_dav
    real(c_double), value :: tol, shift
    type(c_funptr), value :: matvec, precnd, bvec
    real(c_double)        :: eig(n_max), evec(n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface), pointer :: mv, bv
    procedure(pc_iface), pointer :: pc
    logical :: lok
    call c_f_procpointer(matvec, mv)
    call c_f_procpointer(precnd, pc)
    call c_f_procpointer(bvec, bv)
    lok = .false.
    call gen_david_driver(verbose.ne.0,n,n_targ,n_max,max_iter,tol,max_dav,shift,mv,pc,bv,eig,evec,lok)
    ok = merge(1_c_int, 0_c_int, lok)
  end subroutine dla_gen_david_driver
!
  subroutine dla_lobpcg_driver(verbose,gen_eig,n,n_targ,n_max,max_iter,tol,shift, &
                               matvec,precnd,bvec,eig,evec,ok) bind(C,name='dla_lobpcg_driver')
    integer(c_int), value :: verbose, gen_eig, n, n_targ, n_max, max_iter
    real(c_double), value :: tol, shift
    type(c_funptr), value :: matvec, precnd, bvec
    real(c_double)        :: eig(n_max), evec(n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface), pointer :: mv, bv
    procedure(pc_iface), pointer :: pc
    logical :: lok
    call c_f_procpointer(matvec, mv)
    call c_f_procpointer(precnd, pc)
    call c_f_procpointer(bvec, bv)
    lok = .false.
    call lobpcg_driver(verbose.ne.0,gen_eig.ne.0,n,n_targ,n_max,max_iter,tol,shift,mv,pc,bv,eig,evec,lok)
    ok = merge(1_c_int, 0_c_int, lok)
  end subroutine dla_lobpcg_driver
!
  subroutine dla_caslr_eff_driver(verbose,n,n_targ,n_max,max_iter,tol,max_dav, &
                                  apbmul,ambmul,spdmul,smdmul,lrprec,eig,evec,ok) bind(C,name='dla_caslr_eff_driver')
    integer(c_int), value :: verbose, n, n_targ, n_max, max_iter, max_dav
    real(c_double), value :: tol
    type(c_funptr), value :: apbmul, ambmul, spdmul, smdmul, lrprec
    real(c_double)        :: eig(n_max), evec(2*n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface),   pointer :: f1, f2, f3, f4
    procedure(lrpc_iface), pointer :: f5
    logical :: lok
    call c_f_procpointer(apbmul, f1)
    call c_f_procpointer(ambmul, f2)
    call c_f_procpointer(spdmul, f3)
    call c_f_procpointer(smdmul, f4)
    call c_f_procpointer(lrprec, f5)
    lok = .false.
    call caslr_eff_driver(verbose.ne.0,n,2*n,n_targ,n_max,max_iter,tol,max_dav,f1,f2,f3,f4,f5,eig,evec,lok)
    ok = merge(1_c_int, 0_c_int, lok)
  end subroutine dla_caslr_eff_driver
!
  subroutine dla_caslr_driver(verbose,n,n_targ,n_max,max_iter,tol,max_dav, &
                              apbmul,ambmul,spdmul,smdmul,lrprec,eig,evec,ok) bind(C,name='dla_caslr_driver')
    integer(c_int), value :: verbose, n, n_targ, n_max, max_iter, max_dav
    real(c_double), value :: tol
    type(c_funptr), value :: apbmul, ambmul, spdmul, smdmul, lrprec
    real(c_double)        :: eig(n_max), evec(2*n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface),   pointer :: f1, f2, f3, f4
    procedure(lrpc_iface), pointer :: f5
    logical :: lok
    call c_f_procpointer(apbmul, f1)
    call c_f_procpointer(ambmul, f2)
    call c_f_procpointer(spdmul, f3)
    call c_f_procpointer(smdmul, f4)
    call c_f_procpointer(lrprec, f5)
    lok = .false.
    call caslr_driver(verbose.ne.0,n,2*n,n_targ,n_max,max_iter,tol,max_dav,f1,f2,f3,f4,f5,eig,evec,lok)
    ok = merge(1_c_int, 0_c_int, lok)
  end subroutine dla_caslr_driver
!
end module diaglib_cbind
