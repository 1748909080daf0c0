!
! oracle/ref_cbind.f90 -- TEST INFRASTRUCTURE, not product code.
!
! C-callable doors into the *unmodified* reference library (Molecolab-Pisa/diaglib),
! which oracle/Makefile compiles from /root/reference/*.f90 where those files lie.
! Nothing of the reference is restated here: every routine below only converts
! C scalars/pointers to the F77-style by-reference arguments that the reference's
! public module procedures expect (reference diaglib.f90:166-167 public list) and
! forwards the call.  The callbacks are plain C function pointers with the
! reference's own matvec(n,m,x,ax) / precnd(n,m,shift,x,ax) shape
! (reference README.md:34-35).
!
module ref_cbind
  use iso_c_binding
  use diaglib, only : davidson_driver, lobpcg_driver, gen_david_driver, ortho_cd, ortho_vs_x, &
                      b_ortho, b_ortho_vs_x, caslr_eff_driver, caslr_driver
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
  subroutine ref_caslr(verbose,n,n_targ,n_max,max_iter,tol,max_dav, &
                       apbmul,ambmul,spdmul,smdmul,lrprec,eig,evec,ok) bind(C,name='ref_caslr')
    integer(c_int), value :: verbose, n, n_targ, n_max, max_iter, max_dav
    real(c_double), value :: tol
    type(c_funptr), value :: apbmul, ambmul, spdmul, smdmul, lrprec
    real(c_double)        :: eig(n_max), evec(2*n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface),   pointer :: f1, f2, f3, f4
    procedure(lrpc_iface), pointer :: f5
    logical :: lok, lverb
    call c_f_procpointer(apbmul, f1)
    call c_f_procpointer(ambmul, f2)
    call c_f_procpointer(spdmul, f3)
    call c_f_procpointer(smdmul, f4)
    call c_f_procpointer(lrprec, f5)
    lok = .false.
    lverb = verbose .ne. 0
    call caslr_driver(lverb,n,2*n,n_targ,n_max,max_iter,tol,max_dav,f1,f2,f3,f4,f5,eig,evec,lok)
    ok = 0
    if (lok) ok = 1
  end subroutine ref_caslr
!
  subroutine ref_caslr_eff(verbose,n,n_targ,n_max,max_iter,tol,max_dav, &
                           apbmul,ambmul,spdmul,smdmul,lrprec,eig,evec,ok) bind(C,name='ref_caslr_eff')
    integer(c_int), value :: verbose, n, n_targ, n_max, max_iter, max_dav
    real(c_double), value :: tol
    type(c_funptr), value :: apbmul, ambmul, spdmul, smdmul, lrprec
    real(c_double)        :: eig(n_max), evec(2*n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface),   pointer :: f1, f2, f3, f4
    procedure(lrpc_iface), pointer :: f5
    logical :: lok, lverb
    call c_f_procpointer(apbmul, f1)
    call c_f_procpointer(ambmul, f2)
    call c_f_procpointer(spdmul, f3)
    call c_f_procpointer(smdmul, f4)
    call c_f_procpointer(lrprec, f5)
    lok = .false.
    lverb = verbose .ne. 0
    call caslr_eff_driver(lverb,n,2*n,n_targ,n_max,max_iter,tol,max_dav,f1,f2,f3,f4,f5,eig,evec,lok)
    ok = 0
    if (lok) ok = 1
  end subroutine ref_caslr_eff
!
  subroutine ref_davidson(verbose,n,n_targ,n_max,max_iter,tol,max_dav,shift, &
                          matvec,precnd,eig,evec,ok) bind(C,name='ref_davidson')
    integer(c_int), value :: verbose, n, n_targ, n_max, max_iter, max_dav
    real(c_double), value :: tol, shift
    type(c_funptr), value :: matvec, precnd
    real(c_double)        :: eig(n_max), evec(n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface), pointer :: mv
    procedure(pc_iface), pointer :: pc
    logical :: lok, lverb
    call c_f_procpointer(matvec, mv)
    call c_f_procpointer(precnd, pc)
    lok = .false.
    lverb = verbose .ne. 0
    call davidson_driver(lverb,n,n_targ,n_max,max_iter,tol,max_dav,shift,mv,pc,eig,evec,lok)
    ok = 0
    if (lok) ok = 1
  end subroutine ref_davidson
!
  subroutine ref_lobpcg(verbose,gen_eig,n,n_targ,n_max,max_iter,tol,shift, &
                        matvec,precnd,bvec,eig,evec,ok) bind(C,name='ref_lobpcg')
    integer(c_int), value :: verbose, gen_eig, n, n_targ, n_max, max_iter
    real(c_double), value :: tol, shift
    type(c_funptr), value :: matvec, precnd, bvec
    real(c_double)        :: eig(n_max), evec(n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface), pointer :: mv, bv
    procedure(pc_iface), pointer :: pc
    logical :: lok, lverb, lgen
    call c_f_procpointer(matvec, mv)
    call c_f_procpointer(precnd, pc)
    call c_f_procpointer(bvec, bv)
    lok = .false.
    lverb = verbose .ne. 0
    lgen  = gen_eig .ne. 0
    call lobpcg_driver(lverb,lgen,n,n_targ,n_max,max_iter,tol,shift,mv,pc,bv,eig,evec,lok)
    ok = 0
    if (lok) ok = 1
  end subroutine ref_lobpcg
!
  subroutine ref_gen_david(verbose,n,n_targ,n_max,max_iter,tol,max_dav,shift, &
                           matvec,precnd,bvec,eig,evec,ok) bind(C,name='ref_gen_david')
    integer(c_int), value :: verbose, n, n_targ, n_max, max_iter, max_dav
    real(c_double), value :: tol, shift
    type(c_funptr), value :: matvec, precnd, bvec
    real(c_double)        :: eig(n_max), evec(n,n_max)
    integer(c_int)        :: ok
    procedure(mv_iface), pointer :: mv, bv
    procedure(pc_iface), pointer :: pc
    logical :: lok, lverb
    call c_f_procpointer(matvec, mv)
    call c_f_procpointer(precnd, pc)
    call c_f_procpointer(bvec, bv)
    lok = .false.
    lverb = verbose .ne. 0
    call gen_david_driver(lverb,n,n_targ,n_max,max_iter,tol,max_dav,shift,mv,pc,bv,eig,evec,lok)
    ok = 0
    if (lok) ok = 1
  end subroutine ref_gen_david
!
  subroutine ref_ortho_cd(n,m,u,growth,ok) bind(C,name='ref_ortho_cd')
    integer(c_int), value :: n, m
    real(c_double)        :: u(n,m), growth
    integer(c_int)        :: ok
    logical :: lok
    lok = .false.
    call ortho_cd(n,m,u,growth,lok)
    ok = 0
    if (lok) ok = 1
  end subroutine ref_ortho_cd
!
  subroutine ref_ortho_vs_x(n,m,k,x,u) bind(C,name='ref_ortho_vs_x')
    integer(c_int), value :: n, m, k
    real(c_double)        :: x(n,m), u(n,k)
    real(c_double)        :: xx(1)
    call ortho_vs_x(n,m,k,x,u,xx,xx)
  end subroutine ref_ortho_vs_x
!
  subroutine ref_b_ortho(n,m,u,bu) bind(C,name='ref_b_ortho')
    integer(c_int), value :: n, m
    real(c_double)        :: u(n,m), bu(n,m)
    call b_ortho(n,m,u,bu)
  end subroutine ref_b_ortho
!
  subroutine ref_b_ortho_vs_x(n,m,k,x,bx,u) bind(C,name='ref_b_ortho_vs_x')
    integer(c_int), value :: n, m, k
    real(c_double)        :: x(n,m), bx(n,m), u(n,k)
    call b_ortho_vs_x(n,m,k,x,bx,u)
  end subroutine ref_b_ortho_vs_x
!
! the drivers print on unit 6 through the Fortran runtime's own buffer: callers that capture file descriptor 1
! around a verbose call (bench.py reads the reference's timing table) flush it before they look
  subroutine ref_flush() bind(C,name='ref_flush')
    flush(6)
  end subroutine ref_flush
!
end module ref_cbind
