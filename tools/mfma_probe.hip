// Probe of the FP64 MFMA instructions on gfx950 (hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o mfma_probe):
//   1. operand / result lane layout of v_mfma_f64_4x4x4 (one-hot A lane x one-hot B lane -> which D lane lights up);
//   2. issue cost per SIMD of v_mfma_f64_16x16x4 and v_mfma_f64_4x4x4 (4 waves per SIMD, back-to-back independent chains).
// Measured on MI355X: layout  A[blk][i][kk] in lane 16 kk + 4 blk + i,  B[blk][kk][j] in lane 16 kk + 4 blk + j,
// D[blk][i][j] in lane 16 i + 4 blk + j;  ~60 cycles per 16x16x4, ~15 per 4x4x4.  The quarter-tile kernel variants
// (mfma_quarter in diaglib_amd/csrc/hip_engine.hip) rest on both facts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(int la, int lb, double* out)
{
  const int lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  out[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}

template <int KIND> __global__ void rate_kernel(double* out, int iters)
{
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
    } else {
      q0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, q1, 0, 0, 0);
      q2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, q2, 0, 0, 0);
      q3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, q3, 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + q0 + q1 + q2 + q3;
}

int main()
{
  double* d = nullptr;
  if (hipMalloc(&d, 1024 * 256 * sizeof(double)) != hipSuccess) return 1;
  std::vector<double> h(64);
  for (int la = 0; la < 64; ++la) {
    std::printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb) {
      hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, la, lb, d);
      if (hipMemcpy(h.data(), d, 64 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
      for (int l = 0; l < 64; ++l)
        if (h[l] != 0.0) std::printf(" (B%d->D%d)", lb, l);
    }
    std::printf("\n");
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int kind = 0; kind < 2; ++kind)
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(rate_kernel<1>, dim3(1024), dim3(256), 0, 0, d, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
      const int per_iter = kind == 0 ? 2 : 4;
      // 1024 blocks x 4 waves over 256 CUs x 4 SIMDs = 4 waves per SIMD
      const double s_per = ms * 1e-3 / (4.0 * iters * per_iter);
      std::printf("%s: %.3f ms, %.1f ns per instruction per SIMD (~%.0f cycles at 2.1 GHz)\n", kind == 0 ? "16x16x4" : "4x4x4", ms,
                  s_per * 1e9, s_per * 2.1e9);
    }
  (void)hipFree(d);
  return 0;
}
