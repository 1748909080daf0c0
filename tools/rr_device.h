// tools/rr_device.h -- interface of the experimental device Rayleigh-Ritz kernel (tools/rr_device.hip, tools/rr_probe.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dla_rr {

constexpr int RR_NMAX = 128;   // largest projected matrix the kernel takes (its working copy lives in LDS)
constexpr int RR_PMAX = 16;    // most eigenpairs per call

struct Args {
  const double* h;     // projected matrix, column-major, ld = ldh; the triangle named by `upper` is read
  int ldh, upper;
  int n, p;            // order, number of lowest eigenpairs wanted (p <= n)
  double* y;           // out: eigenvectors, n x p, ld = ldy (device)
  int ldy;
  double* eig;         // out: p eigenvalues, ascending (device)
  double* y_host;      // optional pinned mirrors (device-visible addresses), same layout; nullptr = none
  double* eig_host;
  double* refl;        // scratch: n * n doubles (device)
  int* status;         // device word: seq = done, -seq = failed (non-finite input), untouched when the launch was predicated off
  int* status_host;    // optional pinned mirror
  int seq;
  const int* go;       // optional predicate: the launch runs only if *go == go_want (nullptr = always)
  int go_want;
  unsigned long long* dbg;   // optional: 8 device time stamps (100 MHz wall clock) of the phases
};

// bytes of dynamic LDS a launch with this order needs
size_t lds_bytes(int n);
// enqueue on `st`; hipErrorInvalidValue when the shape is outside RR_NMAX / RR_PMAX (the caller keeps the host solver)
hipError_t enqueue(hipStream_t st, const Args& a);

}  // namespace dla_rr
