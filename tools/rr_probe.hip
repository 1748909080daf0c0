// tools/rr_probe.hip -- EXPERIMENT (docs/HISTORY.md 13.4): a device Rayleigh-Ritz kernel (tools/rr_device.hip) against the host solver
// (dla_syev_lowest of the built library): eigenvalues, residuals, orthonormality, and the time per call.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/rr_probe.hip -Ldiaglib_amd/lib -ldiaglib_amd -Wl,-rpath,$PWD/diaglib_amd/lib -o /tmp/rr_probe && /tmp/rr_probe 20
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rr_device.hip"

extern "C" int dla_syev_lowest(char uplo, int n, double* a, int lda, double* w, int m);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

static unsigned long long rs = 88172645463325252ULL;
static double rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) * (1.0 / 9007199254740992.0) - 0.5; }

// kind 0: diag(1..n) + 0.3 * random symmetric (a projected operator of the benchmark's kind); 1: random symmetric;
// 2: clustered (pairs of nearly equal eigenvalues); 3: diagonal + one dense border block (what a Davidson step appends)
static void make(int kind, int n, std::vector<double>& h)
{
  h.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      double v = 0.0;
      if (kind == 0) v = 0.3 * rnd() + (i == j ? 1.0 + i : 0.0);
      else if (kind == 1) v = rnd();
      else if (kind == 2) v = (i == j ? 1.0 + (i / 2) + 1e-9 * (i & 1) : 1e-7 * rnd());
      else v = (i == j ? 2.0 + 0.7 * i : (j >= n - 13 ? 0.2 * rnd() : 0.0));
      h[(size_t)i + (size_t)j * n] = v; h[(size_t)j + (size_t)i * n] = v;
    }
}

int main(int argc, char** argv)
{
  const int reps = argc > 1 ? std::atoi(argv[1]) : 20;
  hipStream_t st; CK(hipStreamCreate(&st));
  double *d_h, *d_y, *d_e, *d_refl; int* d_st; unsigned long long* d_dbg;
  CK(hipMalloc(&d_h, sizeof(double) * 128 * 128)); CK(hipMalloc(&d_y, sizeof(double) * 128 * 16)); CK(hipMalloc(&d_e, sizeof(double) * 16));
  CK(hipMalloc(&d_refl, sizeof(double) * 128 * 128)); CK(hipMalloc(&d_st, sizeof(int))); CK(hipMalloc(&d_dbg, sizeof(unsigned long long) * 8));
  CK(hipMemset(d_refl, 0xff, sizeof(double) * 128 * 128));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int bad = 0;
  std::printf("%5s %3s %4s | %10s %10s %10s | %9s %9s | phases (us): tridiag bisect invit backtr\n", "kind", "p", "n", "eig err", "resid", "orth", "dev us", "host us");
  const int sizes[] = {1, 2, 3, 5, 13, 16, 26, 39, 52, 65, 78, 91, 104, 117, 128};
  for (int kind = 0; kind < 4; ++kind)
    for (int p : {13, 8, 16, 1})
      for (int n : sizes) {
        if (p > n) continue;
        if (kind > 0 && p != 13) continue;
        std::vector<double> h, hh, w(n), y((size_t)n * p), ev(p);
        make(kind, n, h);
        CK(hipMemcpy(d_h, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
        dla_rr::Args a{};
        a.h = d_h; a.ldh = n; a.upper = 1; a.n = n; a.p = p; a.y = d_y; a.ldy = n; a.eig = d_e; a.refl = d_refl; a.status = d_st; a.seq = 1; a.dbg = d_dbg;
        CK(dla_rr::enqueue(st, a));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(dla_rr::enqueue(st, a));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        int stv = 0; unsigned long long dbg[8];
        CK(hipMemcpy(&stv, d_st, sizeof(int), hipMemcpyDeviceToHost));
        CK(hipMemcpy(dbg, d_dbg, sizeof dbg, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y.data(), d_y, sizeof(double) * n * p, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ev.data(), d_e, sizeof(double) * p, hipMemcpyDeviceToHost));
        // host
        double host_us = 0.0;
        for (int r = 0; r < reps + 1; ++r) {
          hh = h;
          auto t0 = std::chrono::steady_clock::now();
          dla_syev_lowest('u', n, hh.data(), n, w.data(), p);
          auto t1 = std::chrono::steady_clock::now();
          if (r > 0) host_us += std::chrono::duration<double, std::micro>(t1 - t0).count();
        }
        host_us /= reps;
        double nrm = 0.0;
        for (double v : h) nrm = std::fmax(nrm, std::fabs(v));
        nrm *= n;
        double eerr = 0.0, res = 0.0, orth = 0.0;
        for (int q = 0; q < p; ++q) {
          eerr = std::fmax(eerr, std::fabs(ev[q] - w[q]));
          for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += h[(size_t)i + (size_t)j * n] * y[(size_t)j + (size_t)q * n];
            res = std::fmax(res, std::fabs(s - ev[q] * y[(size_t)i + (size_t)q * n]));
          }
          for (int q2 = 0; q2 <= q; ++q2) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += y[(size_t)i + (size_t)q * n] * y[(size_t)i + (size_t)q2 * n];
            orth = std::fmax(orth, std::fabs(s - (q == q2 ? 1.0 : 0.0)));
          }
        }
        // sign convention and agreement with the host vectors where the eigenvalue is simple
        const bool fail = stv != 1 || !(eerr <= 1e-13 * nrm) || !(res <= 2e-13 * nrm) || !(orth <= 1e-11);
        if (fail) ++bad;
        const double f = 1.0 / 2400.0;   // (cycle counter at about 2.4 GHz; only the split matters)
        std::printf("%5d %3d %4d | %10.2e %10.2e %10.2e | %9.1f %9.1f | %6.1f %6.1f %6.1f %6.1f %s\n", kind, p, n, eerr, res, orth, 1e3 * ms / reps, host_us,
                    (dbg[1] - dbg[0]) * f, (dbg[2] - dbg[1]) * f, (dbg[3] - dbg[2]) * f, (dbg[4] - dbg[3]) * f, fail ? "FAIL" : "");
      }
  std::printf("%s (%d failures)\n", bad ? "FAILED" : "ok", bad);
  return bad ? 1 : 0;
}
