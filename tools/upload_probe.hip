// tools/upload_probe.hip -- how fast can a small host matrix (the Ritz coefficients Y, 13..27 KB) reach the device?
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/upload_probe tools/upload_probe.hip && tools/bin/upload_probe
// Variants, each timed as  host has the data -> a dependent kernel has finished  (stream otherwise idle):
//   A  hipMemcpyAsync from pinned memory (what stage_to_device does; shows up as __amd_rocclr_copyBuffer)
//   B  own copy kernel reading the pinned, device-mapped buffer
//   C  the consumer reads the pinned buffer directly (every block: PCIe reads)
//   D  host stores into fine-grained DEVICE memory through the PCIe BAR (if the platform maps it), no copy at all
// The BAR variant is probed behind a SIGSEGV / SIGBUS handler: a platform that does not map device memory for the host faults there.
#include <hip/hip_runtime.h>
#include <setjmp.h>
#include <signal.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void copy_kernel(double* dst, const double* src, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 < n) { ((double2*)dst)[i] = ((const double2*)src)[i]; }
  else if (2 * i < n) dst[2 * i] = src[2 * i];
}

// stand-in for the consumer: every block reads the whole small matrix and adds it up
__global__ void consume_kernel(const double* y, int n, double* out)
{
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += y[i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// the same consumer; the block that finishes last raises a flag in pinned host memory (the host polls it instead of an event)
__global__ void consume_flag_kernel(const double* y, int n, double* out, unsigned* ticket, volatile unsigned* host_flag, unsigned seq)
{
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += y[i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    out[blockIdx.x] = red[0];
    __threadfence_system();
    const unsigned t = atomicAdd(ticket, 1u);
    if (t == gridDim.x - 1) { *ticket = 0u; __hip_atomic_store((unsigned*)host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}

static sigjmp_buf g_jmp;
static void on_fault(int) { siglongjmp(g_jmp, 1); }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  const int n = 117 * 16 * 2;              // doubles: Y packed for one 16-column tile pair at L = 117 (30 KB)
  const int blocks = 256, reps = 200;
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *h_pin, *h_pin_dev, *d_y, *d_out, *h_out;
  CHK(hipHostMalloc((void**)&h_pin, sizeof(double) * n, hipHostMallocMapped));
  CHK(hipHostGetDevicePointer((void**)&h_pin_dev, h_pin, 0));
  CHK(hipMalloc((void**)&d_y, sizeof(double) * n));
  CHK(hipMalloc((void**)&d_out, sizeof(double) * blocks));
  CHK(hipHostMalloc((void**)&h_out, sizeof(double) * blocks, hipHostMallocMapped));
  std::vector<double> src(n);
  hipEvent_t ev;
  CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  auto wait = [&]() { (void)hipEventRecord(ev, st); while (hipEventQuery(ev) == hipErrorNotReady) {} };
  auto check = [&](int r, const char* what) {
    (void)hipMemcpy(h_out, d_out, sizeof(double) * blocks, hipMemcpyDeviceToHost);
    const double want = (double)n * (r + 1);
    if (h_out[0] != want || h_out[blocks - 1] != want) std::printf("  %s: WRONG sum %.1f (want %.1f)\n", what, h_out[0], want);
  };
  double tA = 0, tB = 0, tC = 0, tD = 0;
  for (int r = 0; r < reps; ++r) {
    for (int i = 0; i < n; ++i) src[i] = r + 1;
    // A
    double t0 = now_us();
    std::memcpy(h_pin, src.data(), sizeof(double) * n);
    (void)hipMemcpyAsync(d_y, h_pin, sizeof(double) * n, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(consume_kernel, dim3(blocks), dim3(256), 0, st, (const double*)d_y, n, d_out);
    wait();
    if (r >= 20) tA += now_us() - t0;
    if (r == reps - 1) check(r, "A");
    // B
    t0 = now_us();
    std::memcpy(h_pin, src.data(), sizeof(double) * n);
    hipLaunchKernelGGL(copy_kernel, dim3((n / 2 + 255) / 256), dim3(256), 0, st, d_y, (const double*)h_pin_dev, n);
    hipLaunchKernelGGL(consume_kernel, dim3(blocks), dim3(256), 0, st, (const double*)d_y, n, d_out);
    wait();
    if (r >= 20) tB += now_us() - t0;
    if (r == reps - 1) check(r, "B");
    // C
    t0 = now_us();
    std::memcpy(h_pin, src.data(), sizeof(double) * n);
    hipLaunchKernelGGL(consume_kernel, dim3(blocks), dim3(256), 0, st, (const double*)h_pin_dev, n, d_out);
    wait();
    if (r >= 20) tC += now_us() - t0;
    if (r == reps - 1) check(r, "C");
  }
  // E: like B, but the host polls a flag the kernel raises in pinned memory instead of an event behind it
  {
    unsigned *d_ticket, *h_flag, *h_flag_dev;
    CHK(hipMalloc((void**)&d_ticket, sizeof(unsigned)));
    CHK(hipMemset(d_ticket, 0, sizeof(unsigned)));
    CHK(hipHostMalloc((void**)&h_flag, sizeof(unsigned), hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void**)&h_flag_dev, h_flag, 0));
    *h_flag = 0;
    double tE = 0;
    for (int r = 0; r < reps; ++r) {
      for (int i = 0; i < n; ++i) src[i] = r + 1;
      const double t0 = now_us();
      std::memcpy(h_pin, src.data(), sizeof(double) * n);
      hipLaunchKernelGGL(copy_kernel, dim3((n / 2 + 255) / 256), dim3(256), 0, st, d_y, (const double*)h_pin_dev, n);
      hipLaunchKernelGGL(consume_flag_kernel, dim3(blocks), dim3(256), 0, st, (const double*)d_y, n, d_out, d_ticket, (volatile unsigned*)h_flag_dev, (unsigned)(r + 1));
      while (*(volatile unsigned*)h_flag != (unsigned)(r + 1)) {}
      if (r >= 20) tE += now_us() - t0;
    }
    (void)hipStreamSynchronize(st);
    std::printf("E like B, host polls a pinned flag   : %7.2f us\n", tE / (reps - 20));
  }
  std::printf("A hipMemcpyAsync + consumer      : %7.2f us\n", tA / (reps - 20));
  std::printf("B copy kernel (mapped) + consumer: %7.2f us\n", tB / (reps - 20));
  std::printf("C consumer reads pinned directly : %7.2f us\n", tC / (reps - 20));
  // D: host stores into device memory
  for (int kind = 0; kind < 2; ++kind) {
    double* d_bar = nullptr;
    hipError_t e = kind == 0 ? hipExtMallocWithFlags((void**)&d_bar, sizeof(double) * n, hipDeviceMallocFinegrained)
                             : hipMalloc((void**)&d_bar, sizeof(double) * n);
    const char* nm = kind == 0 ? "fine-grained device memory" : "plain hipMalloc memory";
    if (e != hipSuccess) { std::printf("D %s: allocation failed (%s)\n", nm, hipGetErrorString(e)); continue; }
    std::fflush(stdout);
    // (a forked child does not inherit the device mappings: probe in-process behind a fault handler)
    struct sigaction sa{}, old_segv{}, old_bus{};
    sa.sa_handler = on_fault;
    sigaction(SIGSEGV, &sa, &old_segv); sigaction(SIGBUS, &sa, &old_bus);
    bool ok = false;
    if (sigsetjmp(g_jmp, 1) == 0) { volatile double* p = d_bar; p[0] = 1.0; p[n - 1] = 2.0; ok = (p[0] == 1.0); }
    sigaction(SIGSEGV, &old_segv, nullptr); sigaction(SIGBUS, &old_bus, nullptr);
    if (!ok) { std::printf("D %s: the host cannot store into it (fault)\n", nm); continue; }
    tD = 0;
    for (int r = 0; r < reps; ++r) {
      for (int i = 0; i < n; ++i) src[i] = r + 1;
      const double t0 = now_us();
      std::memcpy(d_bar, src.data(), sizeof(double) * n);
      __atomic_thread_fence(__ATOMIC_SEQ_CST);
      hipLaunchKernelGGL(consume_kernel, dim3(blocks), dim3(256), 0, st, (const double*)d_bar, n, d_out);
      wait();
      if (r >= 20) tD += now_us() - t0;
      if (r == reps - 1) check(r, "D");
    }
    std::printf("D host stores into %-27s + consumer: %7.2f us\n", nm, tD / (reps - 20));
  }
  return 0;
}
