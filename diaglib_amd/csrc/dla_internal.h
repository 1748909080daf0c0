// diaglib_amd/csrc/dla_internal.h -- internal seam between the host logic (host_logic.cpp:
// ortho_cd / ortho_vs_x control flow, callback trampolines, statistics) and the device
// engine (hip_engine.hip: HIP kernels, streams, RCCL).  The product library links exactly
// one engine, the HIP one; there is no CPU engine in the product.
#pragma once
#include <cstddef>
#include <string>
#include <vector>
#include "../../include/diaglib_amd.h"

namespace dla {

// Outcome of a device-driven orthogonalisation (Engine::ortho_chain)
struct OrthoReport {
  int handled = 0;      // 0: the engine does not take this shape / mode, run the host-driven loop
  int status = 0;       // 1 done, 2 ortho_cd ran out of iterations, 3 factorisation failed after level shifting,
                        // 4 ortho_vs_x ran out of iterations
  int outer_its = 0, macro_its = 0, shifts = 0;
  double growth = 1.0;
  int clean = 1;        // ortho_chain_finish: 1 when the launches enqueued by ortho_chain_begin were the whole chain
};

// The block operations the orthogonalisation control flow (ortho_cd / ortho_vs_x / ortho in host_logic.cpp) is written
// over.  The device engine implements them on n-length panels in HBM; get_coeffs runs the same control flow on
// host-size coefficient blocks through a plain-loop implementation (CoeffOps in host_logic.cpp) -- it implements these
// few operations and nothing else.  Panels are column-major, ld = n; small matrices are host arrays.
struct BlockOps {
  virtual ~BlockOps() {}
  // C = X^T U (cross-rank reduced when the panels are row shards)
  virtual int gram(int n, int l, const double* x, int k, const double* u, double* c_host, int ldc) = 0;
  // mode 0: Z = X C ; mode 1: Z -= X C (Z read and written)
  virtual int gemm(int n, int l, const double* x, int k, const double* c_host, int ldc, double* z, int mode) = 0;
  // in place U <- U W, W (k x k, host, general) -- used for the triangular update
  virtual int trmm(int n, int k, double* u, const double* w_host, int ld) = 0;
  // fused sweeps: the update plus the Gram matrix of its result.  Default = two sweeps.
  virtual int trmm_gram(int n, int k, double* u, const double* w_host, int ld, double* g_host, int ldg)
  {
    int st = trmm(n, k, u, w_host, ld);
    return st ? st : gram(n, k, u, k, u, g_host, ldg);
  }
  virtual int update_gram(int n, int l, const double* x, int k, const double* c_host, int ldc, double* u, double* g_host, int ldg)
  {
    int st = gemm(n, l, x, k, c_host, ldc, u, 1);
    return st ? st : gram(n, k, u, k, u, g_host, ldg);
  }
  // One sweep over the contiguous panel [X | U] (n x (m+k)):  U <- [X | U] * C'  (C' is (m+k) x k) plus the
  // Gram matrix of the new U.  Used to fold a pending triangular update into the projection step.
  virtual bool can_combo(int /*m*/, int /*k*/) { return false; }
  virtual int combo_gram(int, int, const double*, int, const double*, int, double*, double*, int) { return DLA_ERR_ARG; }
  // ortho_vs_x (m > 0: U is the block that follows X in one panel; bx = X, or B X for the B-metric variant) or plain
  // ortho_cd (m == 0) with the k x k factorisations and the loop decisions on the device: one host wait per call.
  virtual int ortho_chain(int /*n*/, int /*m*/, int /*k*/, const double* /*x*/, const double* /*bx*/, double* /*u*/,
                          OrthoReport* rep) { rep->handled = 0; return 0; }
  // The same in two halves (dla_expand_project): begin enqueues the planned launches and returns with the chain in flight
  // (rep->handled = 1, status 0); finish reads the device's report -- `waited`: the caller has waited for the stream in
  // between -- and completes the chain when it went another way than planned (rep->clean = 0 then: whatever the caller
  // enqueued in between has read an unfinished block).  Nothing but launches on the engine's stream may come in between.
  virtual int ortho_chain_begin(int, int, int, const double*, const double*, double*, OrthoReport* rep) { rep->handled = 0; return 0; }
  virtual int ortho_chain_finish(OrthoReport*, bool /*waited*/) { return DLA_ERR_ARG; }
  // b_ortho (M = U^T BU, L = chol(M), U <- U L^-T, BU <- BU L^-T) enqueued without a host wait, optionally only behind a chain
  // that ended well; *handled = 0: not taken (the caller runs the host-driven b_ortho).  b_ortho_ahead_status() after the
  // caller's next host wait: 1 done, 0 did not run, -1 the metric is not positive definite.
  virtual int b_ortho_ahead(int /*n*/, int /*k*/, double* /*u*/, double* /*bu*/, bool /*behind_chain*/, int* handled) { *handled = 0; return 0; }
  virtual int b_ortho_ahead_status() { return 0; }
  // the top k GLOBAL rows of the n x k block u (row0 = global index of local row 0), k x k to the host
  virtual int top_rows(int n, int k, const double* u, long long row0, double* qt_host) = 0;
  // replace column uj (n local rows, global index of row 0 = row0) by a generated one -- the Householder fallback's answer to
  // a column that lies in the span of its predecessors (gram_schmidt2 in host_logic.cpp).  Default: host memory.
  virtual int fresh_column(int n, double* uj, long long row0, unsigned long long seed)
  {
    for (int i = 0; i < n; ++i) {
      unsigned long long s = (seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(row0 + i + 1));
      s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ULL; s ^= s >> 27; s *= 0x94D049BB133111EBULL; s ^= s >> 31;
      uj[i] = (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
    return 0;
  }
  // set by a caller that B-orthonormalises the block by Cholesky-QR right behind ortho_vs_x (dla_expand_project_metric): a device
  // chain may then end without applying its last pending triangular factor (OrthoTailArgs::drop_final in hip_engine.hip)
  bool drop_final = false;
  // ... and with publish_pending the chain hands what it left undone to the host instead (pending_block): for a caller that folds it
  // into its small matrices and coefficient blocks (dla_expand_project modes 3 and 4).  p = [E ; T] is (m + k) x k: the finished
  // block is [X | U_stored] p -- T the upper-triangular factor that was not applied, E the closing projection that was not run
  // (only the three-pass schedule of the device chain leaves one).  Default: [0 ; I] (nothing is ever pending in the host-driven loops).
  bool publish_pending = false;
  double drop_final_tol = 0.0;     // > 0: the block stays pending only when the closing pass found max |U^T U - I| below it
  double drop_final_stol = 1.0e-4; // ... and max |X^T U| of the stored block below this (three-pass schedule)
  // *applied = 1: the chain's closing sweep HAS applied p to the block in memory (it ran without measuring anything): the block in
  // memory is [X | U_measured] p, and what the caller still owes it is the k x k factor of its Gram matrix I - E^T E
  virtual int pending_block(int m, int k, double* p, int ldp, int* applied)
  {
    for (int j = 0; j < k; ++j)
      for (int i = 0; i < m + k; ++i) p[(size_t)i + (size_t)j * ldp] = (i == m + j) ? 1.0 : 0.0;
    if (applied) *applied = 0;
    return 0;
  }
  // basis_exact (dla_expand_project mode 5): the stored columns X are not a finished basis but X D is, with the caller's upper-
  // triangular D kept on the device block by block (basis_sync; m = columns in front of the block, k <= 0: forget everything).
  // Device chains then project with X (D D^T) X^T, and a block may stay pending however far the stored columns are from orthonormal.
  bool basis_exact = false;
  virtual int basis_sync(int /*m*/, int /*k*/, const double* /*dmat*/, int /*ld*/) { return 0; }
  virtual bool basis_exact_ok() const { return false; }   // the device-driven chain takes one-tile blocks on this context
  // xu <- D D^T xu (m x k, leading dimension ld) while basis_exact is set: what the host-driven loop multiplies X^T U with before a
  // projection, so that it projects with X (D D^T) X^T as the device chain does (a chain that stopped half way -- ortho_cd out of
  // iterations, Householder fallback -- is continued there)
  virtual int basis_dd(int /*m*/, int /*k*/, double* /*xu*/, int /*ld*/) { return 0; }
  // What the engine's copy of the caller's D says about the m columns in front of a block: 0 = D is the identity there (the stored
  // columns are finished: engines that never leave anything pending always answer this), 1 = the copy describes exactly these m
  // columns and is NOT the identity (a projection against the stored columns alone would leave |X_c^T X_c - I| of what it removes),
  // -1 = the copy does not describe them.
  virtual int basis_state(int /*m*/) const { return 0; }
  // basis columns (block included) up to which the device-driven chain can project with the caller's D (0: not at all)
  virtual int basis_capacity() const { return 0; }
  // the device-driven chain must not take the next calls: the host-driven loop runs (dla_expand_project mode 5 on a shape the device
  // cannot project exactly, mode 6)
  bool chain_off = false;
  int ortho_maxit = 10;      // maxit of ortho_cd / ortho_vs_x (diaglib.f90:3224,3521); DLA_OPT_ORTHO_MAXIT lowers it in tests
  std::string err;
};

// What the host logic needs from a device.  All panel pointers are device addresses,
// column-major, ld = n.  Reductions (gram / residual norms / nrm2) return host-visible,
// cross-rank-reduced results and imply a stream synchronisation.
struct Engine : BlockOps {
  virtual const char* name() const = 0;
  virtual void* stream() = 0;

  virtual int alloc(size_t bytes, void** dev) = 0;
  virtual int free_(void* dev) = 0;
  virtual int zero(void* dev, size_t bytes) = 0;
  virtual int trim(size_t* released) { if (released) *released = 0; return 0; }   // release cached free blocks
  virtual int h2d(void* dev, const void* host, size_t bytes) = 0;
  virtual int d2h(void* host, const void* dev, size_t bytes) = 0;
  virtual int d2d(void* dst, const void* src, size_t bytes) = 0;
  virtual int sync() = 0;

  // C = X^T U for two n x l panels when only the lower triangle of C will be read (dsyev 'l'): entries above
  // the 16 x 16 block diagonal may be left zero.  Default = the full product.
  virtual int gram_lower(int n, int l, const double* x, const double* u, double* c_host, int ldc) { return gram(n, l, x, l, u, c_host, ldc); }
  // top rows through the Gram door, E^T U with E = (e_1 .. e_k): works for row-sharded panels with what every engine has
  int top_rows(int n, int k, const double* u, long long row0, double* qt_host) override
  {
    void* ev = nullptr;
    int st = alloc(sizeof(double) * (size_t)n * k, &ev);
    if (st) return st;
    st = zero(ev, sizeof(double) * (size_t)n * k);
    const double unit = 1.0;
    for (int j = 0; j < k && !st; ++j) {
      const long long lr = (long long)j - row0;             // global row j on this shard?
      if (lr >= 0 && lr < n) st = h2d((double*)ev + (size_t)j * n + lr, &unit, sizeof(double));
    }
    if (!st) st = gram(n, k, (const double*)ev, k, u, qt_host, k);
    int stf = free_(ev);
    return st ? st : stf;
  }
  virtual int ritz_residual(int n, int l, int m, const double* v, const double* av, const double* y_host, int ldy,
                            const double* eig, int n_res, const int* skip, double* evec, double* r,
                            double* avy /* optional n x m: uncorrected AV*Y */,
                            double* sumsq_max /* 2*n_res: sum r^2, max|r| */) = 0;
  // the same sweep also forms P = V C2 and AP = AV C2 (k2 columns; LOBPCG's new P block, reference diaglib.f90:495-501,
  // whose products read the same two panels as the Ritz step).  Default: the Ritz step, then two panel products.
  virtual int ritz_residual_p(int n, int l, int m, const double* v, const double* av, const double* y_host, int ldy,
                              const double* eig, int n_res, const int* skip, double* evec, double* r, double* avy,
                              double* sumsq_max, int k2, const double* c2_host, int ldc2, double* p, double* ap)
  {
    int st = ritz_residual(n, l, m, v, av, y_host, ldy, eig, n_res, skip, evec, r, avy, sumsq_max);
    if (!st && k2 > 0) st = gemm(n, l, v, k2, c2_host, ldc2, p, 0);
    if (!st && k2 > 0) st = gemm(n, l, av, k2, c2_host, ldc2, ap, 0);
    return st;
  }
  // e = V Y1, r = AV Y2 - eig e for the first n_res columns (skip as above) with the norms of r: the residual blocks of the
  // linear-response drivers (two panels, two coefficient sets).  Default: two panel products (e, and t = AV Y2 in t_work) and
  // the Ritz sweep on the two n x m results with Y = identity (its copy of e goes to `junk`).
  virtual int ritz_residual2(int n, int l, int m, const double* v, const double* av, const double* y1_host, int ldy1,
                             const double* y2_host, int ldy2, const double* eig, int n_res, const int* skip,
                             double* e, double* r, double* t_work, double* junk, double* sumsq_max)
  {
    int st = gemm(n, l, v, m, y1_host, ldy1, e, 0);
    if (!st) st = gemm(n, l, av, m, y2_host, ldy2, t_work, 0);
    if (st) return st;
    std::string ident_s((size_t)m * m * sizeof(double), '\0');
    double* ident = (double*)&ident_s[0];
    for (int j = 0; j < m; ++j) ident[(size_t)j * m + j] = 1.0;
    return ritz_residual(n, m, m, e, t_work, ident, m, eig, n_res, skip, junk, r, nullptr, sumsq_max);
  }
  virtual int axpy(size_t len, double alpha, const double* x, double* y) = 0;
  virtual int sumsq(size_t len, const double* x, double* out) = 0;
  virtual int stream_triad(size_t /*len*/, int /*reps*/, double* gbps) { *gbps = 0.0; return DLA_ERR_ARG; }
  // evec(i,j) = u01(seed, row0+i+1, j+1) + offset (counter-based generator, global row indices); rows whose global
  // index exceeds support_rows (when > 0) are set to zero
  virtual int random_fill(int n, int m, double* evec, long long row0, unsigned long long seed, double offset,
                          long long support_rows) = 0;
  int fresh_column(int n, double* uj, long long row0, unsigned long long seed) override { return random_fill(n, 1, uj, row0, seed, -0.5, 0); }

  virtual int synth_setup(long long n_global, long long row0, int n_local, int rank_w, double sigma) = 0;
  virtual int synth_matvec(int n, int m, const double* x, double* ax) = 0;
  // the sample operators of the linear-response / generalised drivers around the same W (hip_engine.hip SynthKind)
  virtual int synth_apply(int /*kind*/, int /*n*/, int /*m*/, const double* /*x*/, double* /*y*/) { return DLA_ERR_ARG; }
  virtual int synth_lrprec(int /*variant*/, int /*n*/, int /*m*/, double /*fac*/, const double* /*xp*/, const double* /*xm*/,
                           double* /*yp*/, double* /*ym*/) { return DLA_ERR_ARG; }
  virtual int synth_precnd(int n, int m, double fac, const double* x, double* px) = 0;

  // Ordering of a DEVICE-mode callback against the engine's stream (the user's kernels may run on another stream):
  // mode 0 = stream-ordered through events (the legacy null stream waits for the engine's pending work before the
  // callback, the engine's stream waits for the null stream's work after it; no host wait), 1 = host-synchronise the
  // engine's stream before and the whole device after, 2 = nothing (the callback enqueues on dla_stream itself).
  virtual int callback_begin(int /*mode*/) { return 0; }
  virtual int callback_end(int /*mode*/) { return 0; }

  // sample sparse operator (ELLPACK SpMM + its diagonal preconditioner)
  virtual int spmm_setup_csr(int /*n*/, const long long* /*rowptr*/, const int* /*colind*/, const double* /*values*/) { return DLA_ERR_ARG; }
  // the same operator on a row shard: rows row0 .. row0 + n_local - 1 of A with GLOBAL column indices.  Columns outside the
  // shard must lie within `halo` rows of it (a banded matrix); every rank publishes its first and last `halo` rows of x through
  // the all-reduce of the small-product transport (disjoint slots: a sum that gathers), see sharded_ell_* below.  Collective:
  // every rank calls it (the halo width is agreed by a max-reduction).
  virtual int spmm_setup_csr_sharded(int /*n_local*/, long long /*row0*/, long long /*n_global*/, const long long* /*rowptr*/,
                                     const long long* /*colind*/, const double* /*values*/) { return DLA_ERR_ARG; }
  virtual int spmm_matvec(int /*n*/, int /*m*/, const double* /*x*/, double* /*ax*/) { return DLA_ERR_ARG; }
  virtual int spmm_precnd(int /*n*/, int /*m*/, double /*fac*/, const double* /*x*/, double* /*px*/) { return DLA_ERR_ARG; }

  // Staging pipeline of host-mode callbacks: column chunks of a block travel device -> host on one copy stream, the
  // user's routine works on the chunk that has arrived, finished chunks travel host -> device on a second copy stream
  // (PCIe is full duplex), and the engine's own stream only waits for the last upload -- no host wait at the end.
  //   stage_begin():            the copy streams may start once everything queued on the engine's stream is done
  //   stage_d2h(h, d, b, slot): enqueue a download; stage_wait(slot): host waits for that download
  //   stage_h2d(d, h, b):       enqueue an upload;  stage_end(): the engine's stream waits for all uploads
  // Defaults = plain synchronous copies (host-memory test engine).
  virtual int stage_begin() { return 0; }
  virtual int stage_d2h(void* host, const void* dev, size_t bytes, int /*slot*/) { return d2h(host, dev, bytes); }
  virtual int stage_wait(int /*slot*/) { return 0; }
  virtual int stage_h2d(void* dev, const void* host, size_t bytes) { return h2d(dev, host, bytes); }
  virtual int stage_end() { return 0; }
  // projection of a block that arrives in column chunks (dla_expand_project with host-mode callbacks): chunk c0 .. c0 + kc - 1
  // of C = X^T U (l x k_total) behind the uploads issued so far, no host wait; collect waits once and copies C out
  virtual bool gram_chunks_ok(int, int, int) { return false; }
  virtual int gram_chunk(int, int, const double*, int, int, int, const double*) { return DLA_ERR_ARG; }
  virtual int gram_chunks_collect(int, int, double*, int) { return DLA_ERR_ARG; }

  // pinned host staging for host-mode callbacks
  virtual int host_alloc(size_t bytes, void** p) = 0;
  virtual int host_free(void* p) = 0;

  // collectives
  virtual int comm_init(int nranks, int rank, const char id[128]) = 0;
  virtual int comm_finalize() { nranks = 1; rank = 0; hook = nullptr; return 0; }
  // one-shot peer-to-peer all-reduce over IPC mailboxes: export this rank's mailbox (2 handles of 64 bytes), then
  // attach all ranks' handles (rank order)
  virtual int p2p_export(int /*nranks*/, void* /*handles*/) { return DLA_ERR_COMM; }
  virtual int p2p_attach(int /*nranks*/, int /*rank*/, const void* /*handles*/) { return DLA_ERR_COMM; }
  virtual int p2p_detach() { return DLA_OK; }
  virtual int set_p2p_timeout(int /*ms*/) { return DLA_OK; }      // DLA_OPT_P2P_TIMEOUT_MS
  int nranks = 1, rank = 0;
  // Every rank's shard has an even number of rows (agreed when the shards are announced, dla_set_shard).  Whatever decides which
  // SWEEPS and REDUCTIONS a call consists of must not look at the local row count alone: a rank with an odd shard would take
  // another schedule than its peers, and the exchanges would no longer pair up (found by tools/fuzz_multirank.py: odd n on three
  // and four ranks).  even_rows(n) is the test to use.
  bool peers_even = true;
  bool even_rows(long long n) const { return n % 2 == 0 && peers_even; }
  // all-reduce of a few HOST values over the small-product transport (setup-time agreement between the ranks); collective
  virtual int allreduce_host(double* /*v*/, int /*count*/, int /*op: 0 sum, 1 max*/) { return nranks > 1 ? DLA_ERR_COMM : DLA_OK; }
  virtual bool has_transport() const { return hook != nullptr; }     // something that can carry a cross-rank reduction is attached
  bool local_only = false;   // true while working on data that is replicated on every rank (no reductions)
  dla_allreduce_fn hook = nullptr;
  void* hook_user = nullptr;

  // statistics
  dla_stats stats{};
  bool profile = false;
  virtual void collect_times() {}
  // dla_expand_project's run-ahead: what is launched between begin and end(discard = true) was speculation whose output
  // nobody uses -- it is taken out of the launch / byte / flop statistics (and its events out of the time statistics) again
  virtual void spec_stats_begin() {}
  virtual void spec_stats_end(bool /*discard*/) {}
  virtual int kernel_stats(dla_kernel_stat*, int) { return 0; }
  virtual void reset_kernel_stats() {}
  virtual void set_tune(int, int) {}
  virtual int get_tune(int) { return 0; }
  // a driver call starts: whatever the engine adapts from what earlier chains did is put back to its initial state, so that a solve's
  // sequence of sweeps -- and with it every bit of its result -- is a function of that solve alone
  virtual void begin_solve() {}
};

// $DIAGLIB_AMD_HOSTTIME=1: wall time spent inside each C-ABI entry point (printed by dla_destroy)
struct ApiTimer {
  static bool on();
  static void add(const char* name, double dt);
  static void report();
  static double now();
  static void enter(const char* name, double t);     // charges the time since the previous entry point returned to "<prev> .. <name>"
  static void leave(const char* name, double t);
  const char* name; double t0;
  explicit ApiTimer(const char* n) : name(n), t0(on() ? now() : 0.0) { if (on()) enter(name, t0); }
  ~ApiTimer() { if (on()) { const double t1 = now(); add(name, t1 - t0); leave(name, t1); } }
};
#define DLA_CAT2(a, b) a##b
#define DLA_CAT(a, b) DLA_CAT2(a, b)
#define DLA_T(name) dla::ApiTimer DLA_CAT(dla_api_timer_, __LINE__)(name)

// provided by the engine translation unit linked into the library
// ---- a row shard of a sparse matrix as ELLPACK with a halo (spmm_setup_csr_sharded of both engines)
// Column indices are kept relative to an EXTENDED local vector [halo rows of the previous rank | the shard's n rows | halo rows of
// the next rank]: index = global column - row0 + halo.
struct ShardedEll {
  int n = 0, w = 0, halo = 0;
  long long row0 = 0;
  std::vector<int> col;          // [w][n]
  std::vector<double> val, diag; // [w][n], [n]
};
// widest row and the number of rows of the neighbours this shard's columns reach into; checks the indices
inline int sharded_ell_need(int n, long long row0, long long n_global, const long long* rowptr, const long long* colind, int* w,
                            long long* need, std::string& err)
{
  *w = 0; *need = 0;
  if (n <= 0 || row0 < 0 || row0 + n > n_global || !rowptr || !colind) { err = "spmm_setup_csr_sharded: bad arguments"; return DLA_ERR_ARG; }
  for (int i = 0; i < n; ++i) {
    const long long p0 = rowptr[i], p1 = rowptr[i + 1];
    if (p1 < p0) { err = "spmm_setup_csr_sharded: row pointers not ascending"; return DLA_ERR_ARG; }
    if (p1 - p0 > *w) *w = (int)(p1 - p0);
    for (long long p = p0; p < p1; ++p) {
      const long long c = colind[p];
      if (c < 0 || c >= n_global) { err = "spmm_setup_csr_sharded: column index out of range"; return DLA_ERR_ARG; }
      if (c < row0 && row0 - c > *need) *need = row0 - c;
      if (c >= row0 + n && c - (row0 + n - 1) > *need) *need = c - (row0 + n - 1);
    }
  }
  if (*w <= 0) { err = "spmm_setup_csr_sharded: empty shard"; return DLA_ERR_ARG; }
  return DLA_OK;
}
inline void sharded_ell_build(int n, long long row0, const long long* rowptr, const long long* colind, const double* values,
                              int halo, ShardedEll& e)
{
  int w = 0;
  for (int i = 0; i < n; ++i) if ((int)(rowptr[i + 1] - rowptr[i]) > w) w = (int)(rowptr[i + 1] - rowptr[i]);
  e.n = n; e.w = w; e.halo = halo; e.row0 = row0;
  e.col.assign((size_t)w * n, 0); e.val.assign((size_t)w * n, 0.0); e.diag.assign((size_t)n, 0.0);
  for (int i = 0; i < n; ++i) {
    const long long p0 = rowptr[i], p1 = rowptr[i + 1];
    for (int q = 0; q < w; ++q) {
      const bool in = p0 + q < p1;
      const long long c = in ? colind[p0 + q] : row0 + i;      // (padding: a zero that points at the row itself)
      e.col[(size_t)q * n + i] = (int)(c - row0 + halo);
      e.val[(size_t)q * n + i] = in ? values[p0 + q] : 0.0;
      if (in && c == row0 + i) e.diag[i] += values[p0 + q];
    }
  }
}

Engine* make_engine(int device, std::string& err);
int engine_unique_id(char id[128]);

}  // namespace dla

struct dla_ctx {
  dla::Engine* eng = nullptr;
  int callbacks_on_device = 0;
  int evec_on_device = 0;
  int verbose_ortho = 0;
  int caslr_algorithm = 0;   // DLA_OPT_CASLR_ALGORITHM
  int stage_chunks = 0;      // DLA_OPT_STAGE_CHUNKS: 0 = automatic
  int callback_order = 1;    // DLA_OPT_CALLBACK_ORDER (default: host-synchronised, safe for callbacks on any stream)
  int p2p_timeout_ms = 5000; // DLA_OPT_P2P_TIMEOUT_MS
  int run_ahead = 1;         // DLA_OPT_RUN_AHEAD
  int pending_blocks = 1;    // DLA_OPT_PENDING_BLOCKS
  long long n_global = -1;   // -1: single shard, n_global == n
  long long row0 = 0;
  std::vector<double> pending_p;   // dla_expand_project modes 3 / 4: the block [E ; T] the last call left pending ((m + k) x k, ld m + k)
  int pending_k = 0, pending_m = 0, pending_applied = 0;
  int agree_seq = 0;         // agreements on the shard layout this context has taken part in (agree_on_shards)
  int ref_depth = 0;         // nesting of entry points (reference-schedule flops are counted at the outermost one)
  std::string err;
  // pinned staging buffers for host-mode callbacks
  double* stage_x = nullptr;
  double* stage_y = nullptr;
  size_t stage_bytes = 0;
};
