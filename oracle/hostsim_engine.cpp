// oracle/hostsim_engine.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A stand-in for the HIP engine that keeps "device" panels in host memory and runs the oracle's
// plain-C kernels (oracle.c).  tests/hostsim.py links it with the PRODUCT's host logic
// (diaglib_amd/csrc/host_logic.cpp, smalldense.cpp) and the PRODUCT's Fortran drivers into
// tests/_build/libdiaglib_hostsim.so so that
//   * the host control flow (drivers, ortho_cd / ortho_vs_x loops, callback trampolines) is
//     exercised on machines without a GPU (`pytest -m "not gpu"`), and
//   * the row-sharded multi-rank path (every reduction point of the C-ABI) is covered by
//     world_size-2 gloo tests through the dla_set_allreduce_hook door.
// It is never part of diaglib_amd/lib/libdiaglib_amd.so: the product has no CPU path.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../diaglib_amd/csrc/dla_internal.h"
#include "oracle.h"

namespace {

struct HostSimEngine : dla::Engine {
  std::string nm = "hostsim:oracle-kernels";
  long long syn_row0 = 0; int syn_n = 0, syn_rw = 0; double syn_sigma = 0.0;

  const char* name() const override { return nm.c_str(); }
  void* stream() override { return nullptr; }
  int alloc(size_t bytes, void** dev) override { *dev = std::malloc(bytes ? bytes : 8); return *dev ? 0 : DLA_ERR_ALLOC; }
  int free_(void* dev) override { std::free(dev); return 0; }
  int zero(void* dev, size_t bytes) override { std::memset(dev, 0, bytes); return 0; }
  int h2d(void* d, const void* h, size_t b) override { std::memcpy(d, h, b); return 0; }
  int d2h(void* h, const void* d, size_t b) override { std::memcpy(h, d, b); stats.host_syncs++; return 0; }
  int d2d(void* d, const void* s, size_t b) override { std::memmove(d, s, b); return 0; }
  int sync() override { return 0; }
  int host_alloc(size_t b, void** p) override { *p = std::malloc(b ? b : 8); return *p ? 0 : DLA_ERR_ALLOC; }
  int host_free(void* p) override { std::free(p); return 0; }
  int comm_init(int, int, const char*) override { err = "hostsim: use dla_set_allreduce_hook"; return DLA_ERR_COMM; }

  int reduce(double* buf, int count, int op)
  {
    if (local_only || nranks <= 1) return 0;
    if (!hook) { err = "hostsim: nranks > 1 without a reduction hook"; return DLA_ERR_COMM; }
    stats.allreduces++;
    hook(hook_user, buf, count, op);
    return 0;
  }

  int allreduce_host(double* v, int count, int op) override { return reduce(v, count, op); }

  int gram(int n, int l, const double* x, int k, const double* u, double* c, int ldc) override
  {
    std::vector<double> t((size_t)l * k);
    orc_gemm_tn(n, l, k, x, n, u, n, t.data(), l);
    int st = reduce(t.data(), l * k, 0);
    if (st) return st;
    for (int j = 0; j < k; ++j) std::memcpy(c + (size_t)j * ldc, t.data() + (size_t)j * l, sizeof(double) * l);
    stats.launches[DLA_OP_GRAM]++;
    return 0;
  }
  int gemm(int n, int l, const double* x, int k, const double* c, int ldc, double* z, int mode) override
  {
    if (l == 0) { if (mode == 0) std::memset(z, 0, sizeof(double) * (size_t)n * k); return 0; }
    if (mode == 0 && z >= x && z < x + (size_t)n * l) {
      // the C-ABI lets Z be a column block of X (include/diaglib_amd.h: dla_panel_gemm); the plain-C
      // kernel works column by column, so go through a scratch panel
      std::vector<double> t((size_t)n * k);
      orc_gemm_nn(n, l, k, 1.0, x, n, c, ldc, 0.0, t.data(), n);
      std::memcpy(z, t.data(), sizeof(double) * (size_t)n * k);
    }
    else if (mode == 0) orc_gemm_nn(n, l, k, 1.0, x, n, c, ldc, 0.0, z, n);
    else orc_gemm_nn(n, l, k, -1.0, x, n, c, ldc, 1.0, z, n);
    stats.launches[DLA_OP_GEMM]++;
    return 0;
  }
  int trmm(int n, int k, double* u, const double* w, int ld) override
  {
    std::vector<double> t((size_t)n * k);
    std::memcpy(t.data(), u, sizeof(double) * (size_t)n * k);
    orc_gemm_nn(n, k, k, 1.0, t.data(), n, w, ld, 0.0, u, n);
    stats.launches[DLA_OP_TRMM]++;
    return 0;
  }
  int ritz_residual(int n, int l, int m, const double* v, const double* av, const double* y, int ldy, const double* eig,
                    int n_res, const int* skip, double* evec, double* r, double* avy, double* out) override
  {
    std::vector<double> ev_tmp;
    if (!evec) { ev_tmp.resize((size_t)n * m); evec = ev_tmp.data(); }       // (evec is optional: the caller does not want the vectors)
    orc_gemm_nn(n, l, m, 1.0, v, n, y, ldy, 0.0, evec, n);
    orc_gemm_nn(n, l, m, 1.0, av, n, y, ldy, 0.0, r, n);
    if (avy) std::memcpy(avy, r, sizeof(double) * (size_t)n * m);
    std::vector<double> ss(n_res > 0 ? n_res : 1, 0.0), mx(n_res > 0 ? n_res : 1, 0.0);
    for (int i = 0; i < n_res; ++i) {
      if (skip && skip[i]) continue;
      double s = 0.0, q = 0.0;
      double* ri = r + (size_t)n * i; const double* ei = evec + (size_t)n * i;
      for (int p = 0; p < n; ++p) { ri[p] -= eig[i] * ei[p]; s += ri[p] * ri[p]; q = std::fmax(q, std::fabs(ri[p])); }
      ss[i] = s; mx[i] = q;
    }
    int st = reduce(ss.data(), n_res, 0);
    if (st) return st;
    st = reduce(mx.data(), n_res, 1);
    if (st) return st;
    for (int i = 0; i < n_res; ++i) { out[2 * i] = ss[i]; out[2 * i + 1] = mx[i]; }
    stats.launches[DLA_OP_RITZ]++;
    return 0;
  }
  int axpy(size_t len, double alpha, const double* x, double* y) override
  {
    for (size_t i = 0; i < len; ++i) y[i] += alpha * x[i];
    return 0;
  }
  int sumsq(size_t len, const double* x, double* out) override
  {
    double s = 0.0;
    for (size_t i = 0; i < len; ++i) s += x[i] * x[i];
    int st = reduce(&s, 1, 0);
    *out = s;
    return st;
  }
  int random_fill(int n, int m, double* evec, long long row0, unsigned long long seed, double offset, long long support_rows) override
  {
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < n; ++i)
        evec[(size_t)j * n + i] = (support_rows <= 0 || row0 + i < support_rows)
                                      ? orc_u01(seed, (unsigned long long)(row0 + i + 1), (unsigned long long)(j + 1)) + offset : 0.0;
    return 0;
  }
  int synth_setup(long long n_global, long long row0, int n_local, int rank_w, double sigma) override
  {
    orc_synth_setup(n_global, row0, n_local, rank_w, sigma);
    syn_row0 = row0; syn_n = n_local; syn_rw = rank_w; syn_sigma = sigma;
    return 0;
  }
  int synth_matvec(int n, int m, const double* x, double* ax) override
  {
    const double* w = orc_synth_w();
    std::vector<double> t((size_t)syn_rw * m);
    orc_gemm_tn(n, syn_rw, m, w, n, x, n, t.data(), syn_rw);
    int st = reduce(t.data(), syn_rw * m, 0);     // the operator's only exchange: r x m all-reduce
    if (st) return st;
    for (int c = 0; c < m; ++c)
      for (int i = 0; i < n; ++i) {
        const double d = (double)(syn_row0 + i + 1) + 1.0;
        double s = 0.0;
        for (int q = 0; q < syn_rw; ++q) s += w[(size_t)q * n + i] * t[q + (size_t)c * syn_rw];
        ax[(size_t)c * n + i] = d * x[(size_t)c * n + i] + syn_sigma * s;
      }
    return 0;
  }
  int synth_precnd(int n, int m, double fac, const double* x, double* px) override
  {
    orc_synth_precnd(&n, &m, &fac, x, px);
    return 0;
  }
  // the sample sparse operator on a row shard (dla_spmm_setup_csr_sharded): the halo rows of x travel through the reduction
  // hook exactly as in the HIP engine -- every rank fills its own two slots of a zeroed buffer, the sum gathers
  dla::ShardedEll ell;
  int spmm_setup_csr_sharded(int n, long long row0, long long n_global, const long long* rowptr, const long long* colind,
                             const double* values) override
  {
    int w = 0; long long need = 0;
    std::string lerr;
    const int bad = (values == nullptr) ? DLA_ERR_ARG : dla::sharded_ell_need(n, row0, n_global, rowptr, colind, &w, &need, lerr);
    const int nr = nranks > 1 ? nranks : 1;
    double mx[2] = {(double)need, bad ? 1.0 : 0.0};
    int st = reduce(mx, 2, 1);
    if (st) return st;
    if (mx[1] != 0.0) { err = bad ? lerr : std::string("spmm_setup_csr_sharded: another rank rejected its shard"); return DLA_ERR_ARG; }
    std::vector<double> lay((size_t)2 * nr, 0.0);
    lay[2 * rank] = (double)row0; lay[2 * rank + 1] = (double)n;
    st = reduce(lay.data(), 2 * nr, 0);
    if (st) return st;
    const long long halo = (long long)mx[0];
    long long expect = 0;
    for (int r = 0; r < nr; ++r) {
      if ((long long)lay[2 * r] != expect) { err = "spmm_setup_csr_sharded: the shards are not contiguous in rank order"; return DLA_ERR_ARG; }
      if (halo > (long long)lay[2 * r + 1]) { err = "spmm_setup_csr_sharded: a shard reaches beyond its neighbour (halo wider than a shard)"; return DLA_ERR_ARG; }
      expect += (long long)lay[2 * r + 1];
    }
    if (expect != n_global) { err = "spmm_setup_csr_sharded: the shards do not cover n_global rows"; return DLA_ERR_ARG; }
    if (halo > 4096) { err = "spmm_setup_csr_sharded: halo wider than 4096 rows (not a banded matrix)"; return DLA_ERR_ARG; }
    dla::sharded_ell_build(n, row0, rowptr, colind, values, (int)halo, ell);
    return 0;
  }
  int spmm_matvec(int n, int m, const double* x, double* ax) override
  {
    if (n != ell.n || ell.col.empty()) { err = "spmm_matvec: n differs from setup"; return DLA_ERR_ARG; }
    const int nr = nranks > 1 ? nranks : 1, H = ell.halo;
    std::vector<double> buf((size_t)nr * 2 * m * (H > 0 ? H : 1), 0.0);
    if (H > 0 && nr > 1) {
      for (int side = 0; side < 2; ++side)
        for (int c = 0; c < m; ++c)
          for (int h = 0; h < H; ++h)
            buf[(((size_t)rank * 2 + side) * m + c) * H + h] = x[(size_t)c * n + (side == 0 ? h : n - H + h)];
      int st = reduce(buf.data(), nr * 2 * m * H, 0);
      if (st) return st;
    }
    const double* prev = buf.data() + (size_t)((rank > 0 ? (rank - 1) * 2 + 1 : 0) * m) * H;
    const double* next = buf.data() + (size_t)((rank + 1 < nr ? (rank + 1) * 2 : 0) * m) * H;
    for (int c = 0; c < m; ++c)
      for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int q = 0; q < ell.w; ++q) {
          const int idx = ell.col[(size_t)q * n + i];
          const double v = idx < H ? prev[(size_t)c * H + idx] : (idx < H + n ? x[(size_t)c * n + idx - H] : next[(size_t)c * H + idx - H - n]);
          s += ell.val[(size_t)q * n + i] * v;
        }
        ax[(size_t)c * n + i] = s;
      }
    return 0;
  }
  int spmm_precnd(int n, int m, double fac, const double* x, double* px) override
  {
    if (n != ell.n || ell.diag.empty()) { err = "spmm_precnd: n differs from setup"; return DLA_ERR_ARG; }
    for (int c = 0; c < m; ++c)
      for (int i = 0; i < n; ++i) {
        const double den = ell.diag[i] + fac;
        px[(size_t)c * n + i] = std::fabs(den) > 1.0e-5 ? x[(size_t)c * n + i] / den : x[(size_t)c * n + i];
      }
    return 0;
  }
  // the sample operators of the generalised / linear-response drivers, y = d(i) x + W C W^T x (definitions: SynthKind in
  // hip_engine.hip -- restated here so that the host-memory engine can run those drivers on row shards)
  int synth_apply(int kind, int n, int m, const double* x, double* y) override
  {
    if (kind < 0 || kind > 5) return DLA_ERR_ARG;
    const double* w = orc_synth_w();
    const int rw = syn_rw;
    std::vector<double> t((size_t)rw * m), ct((size_t)rw * m, 0.0);
    orc_gemm_tn(n, rw, m, w, n, x, n, t.data(), rw);
    int st = reduce(t.data(), rw * m, 0);
    if (st) return st;
    double cpl[16] = {0.0};
    const double tau = 0.05;
    for (int q = 0; q < 4; ++q) {
      if (kind == 0 || kind == 1) cpl[q + 4 * q] = syn_sigma;
      else if (kind == 2) cpl[q + 4 * q] = 0.2 * syn_sigma;
      else if (kind == 5) cpl[q + 4 * q] = 0.1;
    }
    if (kind == 3 || kind == 4) {
      const double tj = kind == 3 ? tau : -tau;
      cpl[0 + 4 * 1] = tj; cpl[1 + 4 * 0] = -tj; cpl[2 + 4 * 3] = tj; cpl[3 + 4 * 2] = -tj;
    }
    for (int c = 0; c < m; ++c)
      for (int q = 0; q < rw; ++q) {
        double v = 0.0;
        for (int p = 0; p < rw; ++p) v += cpl[q + 4 * p] * t[p + (size_t)c * rw];
        ct[q + (size_t)c * rw] = v;
      }
    for (int c = 0; c < m; ++c)
      for (int i = 0; i < n; ++i) {
        const unsigned long long gi = (unsigned long long)(syn_row0 + i + 1);
        const double si = 1.0 + 0.5 / (1.0 + (double)(gi % 7ULL));
        const double d = kind == 0 ? (double)gi + 1.0 : kind == 1 ? (double)gi + 5.0 : kind == 2 ? (double)gi + 2.0 : si;
        double s = 0.0;
        for (int q = 0; q < rw; ++q) s += w[(size_t)q * n + i] * ct[q + (size_t)c * rw];
        y[(size_t)c * n + i] = d * x[(size_t)c * n + i] + s;
      }
    return 0;
  }
};

}  // namespace

namespace dla {
Engine* make_engine(int, std::string&) { return new HostSimEngine(); }
int engine_unique_id(char id[128]) { std::memset(id, 0, 128); return DLA_ERR_COMM; }
}  // namespace dla
