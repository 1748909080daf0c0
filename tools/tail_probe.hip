// tools/tail_probe.hip -- cycle counts of variants of the k x k factorisation step (ortho_tail16), one wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/tail_probe tools/tail_probe.hip && tools/bin/tail_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rlane(double v, int src)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ v4d mfma16(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <int RR>
__device__ __forceinline__ void step(int j, double d, v4d& a, v4d& x, int c, int g)
{
  const int gj = j & 3;
  double inv = __builtin_amdgcn_rcp(d);
  inv = fma(fma(-d, inv, 1.0), inv, inv);
  inv = fma(fma(-d, inv, 1.0), inv, inv);
  const bool rowj = (g == gj);
  const double u = (rowj && c > j) ? a[RR] : 0.0;
  const double xb = rowj ? x[RR] : 0.0;
  const double ua = -u * inv;
  a = mfma16(ua, u, a);
  x = mfma16(ua, xb, x);
}

// variant 0: dynamic loop with switch (as in the product); 1: fully unrolled 16 steps; 2: loop, MFMAs only (no pivot chain);
// 3: loop, pivot chain only (no MFMA); 4: unrolled, k = 13 steps via early exit
template <int V>
__global__ void probe(const double* gin, double* out, unsigned long long* cyc, int k)
{
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  v4d a, x;
  for (int r = 0; r < 4; ++r) { a[r] = gin[(g + 4 * r) * 16 + c]; x[r] = (g + 4 * r == c) ? 1.0 : 0.0; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  if constexpr (V == 0) {
#pragma unroll 1
    for (int j = 0; j < k; ++j) {
      const int src = 16 * (j & 3) + j;
      double d;
      switch (j >> 2) { case 0: d = rlane(a[0], src); break; case 1: d = rlane(a[1], src); break; case 2: d = rlane(a[2], src); break; default: d = rlane(a[3], src); }
      if (!(d > 0.0)) break;
      switch (j >> 2) { case 0: step<0>(j, d, a, x, c, g); break; case 1: step<1>(j, d, a, x, c, g); break; case 2: step<2>(j, d, a, x, c, g); break; default: step<3>(j, d, a, x, c, g); }
    }
  } else if constexpr (V == 1 || V == 4) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (V == 4 && j >= k) break;
      const int src = 16 * (j & 3) + j;
      const double d = rlane(a[j >> 2], src);
      if (j >> 2 == 0) step<0>(j, d, a, x, c, g); else if (j >> 2 == 1) step<1>(j, d, a, x, c, g); else if (j >> 2 == 2) step<2>(j, d, a, x, c, g); else step<3>(j, d, a, x, c, g);
    }
  } else if constexpr (V == 2) {
#pragma unroll 1
    for (int j = 0; j < k; ++j) { const double u = (g == (j & 3)) ? 1e-3 : 0.0; a = mfma16(-u, u, a); x = mfma16(-u, u, x); }
  } else {
    double acc = 0.0;
#pragma unroll 1
    for (int j = 0; j < k; ++j) {
      const double d = rlane(a[0], j) + 2.0 + acc;
      double inv = __builtin_amdgcn_rcp(d);
      inv = fma(fma(-d, inv, 1.0), inv, inv);
      inv = fma(fma(-d, inv, 1.0), inv, inv);
      acc += inv;
    }
    a[1] += acc;
  }
  double chk = a[0] + a[1] + a[2] + a[3] + x[0] + x[1] + x[2] + x[3];
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[lane] = chk;
  if (lane == 0) cyc[0] = t1 - t0;
}

int main()
{
  double h[256];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 4.0 : 0.0) + 0.01 * ((i * 7 + j * 3) % 5 + (j * 7 + i * 3) % 5);
  double *g, *o; unsigned long long* c;
  hipMalloc(&g, sizeof h); hipMalloc(&o, 64 * 8); hipMalloc(&c, 8);
  hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
  const char* names[5] = {"loop + switch (product)", "fully unrolled, 16 steps", "loop, 2 MFMAs only", "loop, pivot chain only", "unrolled, early exit at k"};
  for (int rep = 0; rep < 2; ++rep)
    for (int v = 0; v < 5; ++v) {
      for (int k : {13, 8}) {
        switch (v) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, g, o, c, k); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, g, o, c, k); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, g, o, c, k); break;
          case 3: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, g, o, c, k); break;
          default: hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, g, o, c, k);
        }
        unsigned long long cy = 0;
        hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        if (rep == 1) printf("%-28s k = %2d: %6llu cycles, %5.0f per step\n", names[v], k, cy, (double)cy / (v == 1 ? 16 : k));
      }
    }
  return 0;
}
