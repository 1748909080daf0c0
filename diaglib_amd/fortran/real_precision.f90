!
! real_precision -- kind parameter shared with callers (`use real_precision`), kept under the
! name and value the reference uses (reference real_precision.f90:1-4: dp = 8).
!
module real_precision
  implicit none
  integer, parameter :: dp = 8
end module real_precision
