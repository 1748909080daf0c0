!
! real_precision -- the kind parameter callers obtain through `use real_precision` (the module name and the
! name dp are part of the reference's interface, reference real_precision.f90).  The reference writes the kind as
! the literal 8; here it is tied to the C-ABI's double, which is the same kind with every compiler this builds on
! (checked below at compile time).
!
module real_precision
  use, intrinsic :: iso_c_binding, only : c_double
  implicit none
  private
  public :: dp
  integer, parameter :: dp = c_double
  integer, parameter :: dp_is_the_reference_kind = 1/merge(1, 0, c_double == 8)
end module real_precision
