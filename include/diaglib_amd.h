/*
 * include/diaglib_amd.h -- C-ABI of the MI355X-native diaglib hot path.
 *
 * This is the drop-in boundary.  The reference (Molecolab-Pisa/diaglib)
 * is a Fortran module whose drivers do all O(n) work through BLAS calls on column-major
 * n x k panels (ld = n).  Here the Fortran drivers (diaglib_amd/fortran/diaglib.f90,
 * same public names and argument lists as reference diaglib.f90:166-167) keep only the
 * control flow and reach the device through the entry points below via ISO_C_BINDING.
 * Every entry point names the reference line(s) it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - "dev" pointers are device (HBM) addresses obtained from dla_alloc(); panels are
 *     column-major float64 with leading dimension n (the local row count), columns
 *     contiguous -- exactly the reference layout, so a block of columns can be handed
 *     to a matvec(n,m,x,ax) callback unchanged.
 *   - small matrices (<= lda x lda) live on the host, column-major.
 *   - every function returns 0 on success, a DLA_ERR_* code otherwise; nothing calls
 *     exit().  dla_last_error() gives the message.  The Fortran layer maps failures to
 *     the reference behaviour (ok=.false. / stop with the reference's message).
 *   - with a communicator (dla_comm_init) n is the LOCAL row count of this rank's shard;
 *     every reduction below is summed (or max-ed) over ranks before it is returned, so the
 *     callers are unchanged.
 */
#ifndef DIAGLIB_AMD_H
#define DIAGLIB_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dla_ctx dla_ctx;

enum {
  DLA_OK = 0,
  DLA_ERR_NO_DEVICE = 1,   /* no HIP device / extension not usable: fail loudly      */
  DLA_ERR_ALLOC = 2,       /* reference: check_mem, diaglib.f90:3789-3803            */
  DLA_ERR_ARG = 3,
  DLA_ERR_RUNTIME = 4,     /* a HIP/RCCL call failed                                  */
  DLA_ERR_ORTHO = 5,       /* reference: "catastrophic failure of ortho_vs_x", :3568  */
  DLA_ERR_LAPACK = 6,      /* reference: "dsyev failed", :412-415                     */
  DLA_ERR_COMM = 7
};

/* reference callback shapes, README.md:34-35 (F77 ABI: everything by reference) */
typedef void (*dla_matvec_fn)(const int* n, const int* m, const double* x, double* ax);
typedef void (*dla_precnd_fn)(const int* n, const int* m, const double* fac, const double* x, double* px);
/* linear-response preconditioner lrprec(n,m,fac,xp,xm,yp,ym): reference diaglib.f90:1317, callers main.f90:234-281 */
typedef void (*dla_lrprec_fn)(const int* n, const int* m, const double* fac, const double* xp, const double* xm,
                              double* yp, double* ym);

/* options for dla_set_option */
enum {
  DLA_OPT_CALLBACKS_ON_DEVICE = 1, /* 0 (default, drop-in): callbacks get HOST arrays, staged through pinned
                                      buffers; 1: callbacks get DEVICE addresses (SURVEY 8b "callback residency") */
  DLA_OPT_EVEC_ON_DEVICE = 2,      /* 0 (default): eig/evec of the drivers are host arrays as in the reference;
                                      1: evec is a device address (guess in, Ritz vectors out), eig stays host  */
  DLA_OPT_PROFILE = 3,             /* 1: bracket every kernel launch with HIP events (dla_get_stats)          */
  DLA_OPT_VERBOSE_ORTHO = 4,       /* 1: print ortho_cd/ortho_vs_x pass counts                                 */
  DLA_OPT_CALLBACK_ORDER = 5,      /* ordering contract of DEVICE-mode callbacks against the engine's stream (which is a
                                      non-blocking stream, dla_stream()):
                                      1 (default): host-synchronised -- the engine's stream is drained before the call
                                         and the whole device after it: correct for callbacks on ANY stream (hipfort,
                                         OpenMP target, torch, private streams) that do not synchronise themselves.
                                      0: stream-ordered, no host wait -- before the call the legacy null stream is made
                                         to wait for the engine's pending work, after the call the engine's stream
                                         waits for everything the callback enqueued on the null stream.  For callbacks
                                         that launch on the null stream only.
                                      2: none -- the callback promises to enqueue on dla_stream() only (the built-in
                                         operator does; it is always called this way).                              */
  DLA_OPT_ORTHO_MAXIT = 6,         /* iteration cap of ortho_cd / ortho_vs_x (reference parameter maxit = 10,
                                      diaglib.f90:3224,3521).  TEST knob: a small value makes ortho_cd give up so that
                                      the Householder-QR fallback `ortho` (diaglib.f90:3052-3092, 3534, 3549) runs */
  DLA_OPT_CASLR_ALGORITHM = 7,     /* reduced problem of caslr_driver: 0 (default) the 2 ldu-dimensional pencil (reference
                                      i_alg = 0, dsygv at diaglib.f90:783), 1 the Helmich-Paris route (i_alg = 1, :805-860) */
  DLA_OPT_STAGE_CHUNKS = 8,        /* host-mode callbacks: column chunks a block is cut into for the download | user routine |
                                      upload pipeline.  0 / 1 (default): the caller's routine sees every block once and whole,
                                      like the reference (diaglib.f90:1685, 1786); >= 2: that many chunks (at most 16) = that
                                      many calls per block, whose transfers overlap the routine's work on the neighbouring
                                      chunks -- pays only for an operator without fixed cost per call                       */
  DLA_OPT_P2P_TIMEOUT_MS = 9,      /* peer-to-peer all-reduce (dla_p2p_attach): how long a rank waits INSIDE a kernel for its peers'
                                      contributions, in ms (default 5000; 0 = no limit).  A rank that gives up marks the exchange
                                      as failed in every rank's mailbox: all ranks return DLA_ERR_COMM at their next host wait and
                                      the transport stays down until dla_p2p_detach + a fresh export.  Callers whose ranks can be
                                      further apart than this (host-mode callbacks of unequal length) raise it or use RCCL  */
  DLA_OPT_RUN_AHEAD = 10,          /* dla_expand_project: enqueue the operator and the projection sweep behind the
                                      orthogonalisation chain and read the chain's report at THEIR host wait.  When the chain
                                      takes another route than planned the operator is called a SECOND time on the finished
                                      block and its first output is dropped -- the reference calls matvec exactly once per
                                      block (diaglib.f90:1685, 394-397), so:
                                      1 (default): only for the library's own device operators (dla_synth_*, dla_spmm_matvec),
                                         which are pure functions of their input; a caller's callback is called once, in order;
                                      2: also for the caller's device-mode callbacks (ordering contract 0 or 2) -- the caller
                                         states that the operator keeps no state between calls;
                                      0: never (one call after the other; A/B and debugging)                                  */
  DLA_OPT_PENDING_BLOCKS = 11,     /* dla_expand_project modes 3 / 4 (what the drivers call): 1 (default) the orthogonalisation chain may
                                      leave its closing projection and its last triangular factor to the caller's small matrices
                                      (dla_pending_block); 0: every block is finished in memory (modes 3 / 4 behave like 1 / 0) --
                                      same eigenpairs either way, for A/B runs and tests in one process.  New contexts start
                                      with 0 when $DIAGLIB_AMD_NO_PENDING is set                                              */
  DLA_OPT_TUNE0 = 100              /* 100..107: kernel-shape experiment knobs for the interleaved A/B tools
                                      (tools/tune_*.py, tools/kernel_bench.py); 0 = the shipped default          */
};

/* op classes for statistics */
enum {
  DLA_OP_GRAM = 0,      /* C = X^T U                         */
  DLA_OP_GEMM = 1,      /* Z = X C   /  U -= X C             */
  DLA_OP_TRMM = 2,      /* U <- U L^-T                       */
  DLA_OP_RITZ = 3,      /* fused Ritz vectors + residuals    */
  DLA_OP_ELEM = 4,      /* copy / zero / axpy / fill         */
  DLA_OP_MATVEC = 5,    /* built-in operator                 */
  DLA_OP_PRECND = 6,    /* built-in preconditioner           */
  DLA_OP_COUNT = 7
};

typedef struct {
  long long launches[DLA_OP_COUNT];   /* kernel launches per class                          */
  double    alg_bytes[DLA_OP_COUNT];  /* algorithmic (compulsory) HBM bytes, SURVEY 8d      */
  double    flops[DLA_OP_COUNT];      /* reference-schedule flops, SURVEY 8d                */
  double    ms[DLA_OP_COUNT];         /* HIP-event time (only with DLA_OPT_PROFILE)         */
  long long allreduces;               /* cross-rank reductions issued                       */
  long long host_syncs;               /* stream synchronisations for host-visible results   */
  double    ref_flops;                /* reference-schedule flops of the LOGICAL operations the caller asked for (SURVEY 8d:
                                         what the reference's BLAS calls would execute for them), counted at the entry points --
                                         independent of what was launched, fused, left pending or skipped                 */
} dla_stats;

/* per-kernel statistics: name as rocprofv3 prints it (without namespace and argument list), launches,
 * algorithmic bytes and HIP-event time (DLA_OPT_PROFILE) */
typedef struct {
  char      name[96];
  long long launches;
  double    alg_bytes;
  double    ms;
  double    flops;       /* reference-schedule flops of the launches (what the BLAS calls they replace would execute) */
} dla_kernel_stat;

/* ---------------------------------------------------------------- context */
int  dla_create(dla_ctx** ctx, int device);          /* device < 0: $LOCAL_RANK or 0 */
int  dla_destroy(dla_ctx* ctx);
dla_ctx* dla_default_ctx(void);                      /* the context the Fortran drivers use: one per CALLING THREAD, created on
                                                        first use (own stream, scratch, panel cache, statistics and
                                                        options), so two host threads can solve at the same time.  The
                                                        reference cannot: module-level state, diaglib.f90:155-161 */
int  dla_set_option(dla_ctx* ctx, int option, int value);
/* A driver call starts (the Fortran drivers call it first): adaptive choices of the engine -- which sweep schedule the
 * orthogonalisation chains take, decided from what earlier chains of the SAME solve reported -- go back to their initial state, so
 * that a solve's results do not depend on the solves before it (repeated solves are bit-identical). */
int  dla_begin_solve(dla_ctx* ctx);
int  dla_get_option(dla_ctx* ctx, int option);
const char* dla_last_error(dla_ctx* ctx);
const char* dla_backend_name(dla_ctx* ctx);          /* "hip:gfx950" for the product */
int  dla_get_stats(dla_ctx* ctx, dla_stats* out);
int  dla_reset_stats(dla_ctx* ctx);
/* HIP-event time per kernel class since the last reset, ms[DLA_OP_COUNT] in the order of the DLA_OP_* ids (needs DLA_OPT_PROFILE = 1
 * while the work runs).  What the Fortran module's diaglib_amd_timings returns: the reference prints four buckets (matvec,
 * diagonalization, orthogonalization, total; diaglib.f90:1835-1841) and leaves projection / Ritz vectors / residuals un-bucketed. */
int  dla_class_times(dla_ctx* ctx, double* ms);
int  dla_get_kernel_stats(dla_ctx* ctx, dla_kernel_stat* out, int cap);   /* returns the number of entries */
void* dla_stream(dla_ctx* ctx);                      /* hipStream_t the kernels run on */

/* ---------------------------------------------------------------- multi-GPU */
int  dla_comm_unique_id(char id[128]);                                       /* ncclGetUniqueId          */
int  dla_comm_init(dla_ctx* ctx, int nranks, int rank, const char id[128]);  /* ncclCommInitRank         */
int  dla_comm_finalize(dla_ctx* ctx);                                        /* ncclCommDestroy          */
int  dla_comm_info(dla_ctx* ctx, int* nranks, int* rank);
/* One-shot peer-to-peer all-reduce for the small products, an alternative transport to RCCL for buffers
 * of at most 64 KB: every rank writes its contribution into a slot of every peer's mailbox (fine-grained device memory shared
 * through hipIpc), raises a flag and adds the slots of its own mailbox in rank order -- one xGMI hop, a fixed summation order
 * (bit-identical results on all ranks), no host involvement, and it takes part in the device-driven chains.
 *   1. every rank: dla_p2p_export(ctx, nranks, h)   -> 128 bytes (two hipIpcMemHandle_t)
 *   2. the caller's control plane (MPI, torch.distributed, ...) gathers the nranks x 128 bytes in rank order
 *   3. every rank: dla_p2p_attach(ctx, nranks, rank, all)
 * dla_p2p_detach closes the mailboxes and hands the small products back to the RCCL communicator / the hook (rank count and
 * rank stay); dla_comm_finalize detaches too.  Larger buffers and contexts without mailboxes use the communicator / the hook. */
int  dla_p2p_export(dla_ctx* ctx, int nranks, char handles[128]);
int  dla_p2p_attach(dla_ctx* ctx, int nranks, int rank, const char* all_handles);
int  dla_p2p_detach(dla_ctx* ctx);
/* host-buffer reduction hook (op 0 = sum, 1 = max); lets a caller supply the collective
 * (e.g. MPI or torch.distributed/gloo).  Used when no RCCL communicator is attached. */
typedef void (*dla_allreduce_fn)(void* user, double* buf, int count, int op);
int  dla_set_allreduce_hook(dla_ctx* ctx, dla_allreduce_fn fn, void* user, int nranks, int rank);
/* global rows of this rank: the shards are contiguous in rank order, rank r holds rows row0(r) .. row0(r + 1) - 1.  COLLECTIVE
 * once a transport is attached (and attaching a transport after the shard has been announced is collective in the same way):
 * the ranks exchange their row0 through the transport and agree on what schedule decisions may depend on -- a rank whose shard
 * has an odd number of rows would otherwise pick other sweeps than its peers and the exchanges would not pair up. */
int  dla_set_shard(dla_ctx* ctx, long long n_global, long long row0);

/* ---------------------------------------------------------------- device memory
 * replaces allocate/zero/dcopy of the panels: diaglib.f90:1607-1638,1648,1798-1805, 258-287 */
int  dla_alloc(dla_ctx* ctx, size_t bytes, void** dev);
int  dla_free(dla_ctx* ctx, void* dev);
/* Freed panels are kept for reuse (a driver call allocates the same GiB-sized panels every solve and hipMalloc of such
 * blocks costs tens of milliseconds); dla_trim hands the cached blocks back to the runtime (bytes released in *released,
 * may be NULL).  dla_destroy releases everything. */
int  dla_trim(dla_ctx* ctx, size_t* released);
int  dla_zero(dla_ctx* ctx, void* dev, size_t bytes);
int  dla_upload(dla_ctx* ctx, void* dev, const void* host, size_t bytes);
int  dla_download(dla_ctx* ctx, void* host, const void* dev, size_t bytes);
int  dla_copy(dla_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);   /* dcopy on panels */
int  dla_sync(dla_ctx* ctx);

/* ---------------------------------------------------------------- block algebra */
/* C(l x k) = X(n x l)^T U(n x k), result on the host.  dgemm('t','n') at
 * diaglib.f90:1691 (projection), 3256 (ortho_cd Gram), 3543 (X^T U), 403/313 (LOBPCG S^T AS), 3762. */
int  dla_gram(dla_ctx* ctx, int n, int l, const double* x_dev, int k, const double* u_dev,
              double* c_host, int ldc);
/* Same for two n x l panels when the consumer reads only the LOWER triangle of C (LOBPCG: S^T AS at diaglib.f90:403
 * goes to dsyev('v','l') at :406): entries strictly above the 16 x 16 block diagonal are returned as zero. */
int  dla_gram_lower(dla_ctx* ctx, int n, int l, const double* x_dev, const double* u_dev, double* c_host, int ldc);
/* Z(n x k) = X(n x l) C(l x k).  dgemm('n','n') at diaglib.f90:1717, 420-424, 495-501, 322-324.
 * Z may be a column block of X itself when k <= 48 and 128*l*ceil(k/16) <= 65536 (one output pass, one
 * contraction chunk): every row tile is read completely before it is stored. */
int  dla_panel_gemm(dla_ctx* ctx, int n, int l, const double* x_dev, int k, const double* c_host, int ldc,
                    double* z_dev);
/* U(n x k) -= X(n x l) C(l x k).  dgemm('n','n',-one,...,one) at diaglib.f90:3544, 3633. */
int  dla_panel_update(dla_ctx* ctx, int n, int l, const double* x_dev, int k, const double* c_host, int ldc,
                      double* u_dev);
/* U <- U Linv^T with Linv lower triangular (k x k, host).  dtrmm('r','l','t','n') at diaglib.f90:3327. */
int  dla_trmm_linvt(dla_ctx* ctx, int n, int k, double* u_dev, const double* linv_host, int ld);
/* Fused Ritz step.  evec = V Y(:,1:m), r = AV Y(:,1:m); then for i < n_res with skip[i]==0:
 * r_i -= eig_i evec_i, rnorm[2i] = ||r_i||_2/sqrt(n_global), rnorm[2i+1] = max|r_i|.
 * avy_dev (may be NULL) also receives the uncorrected AV Y (LOBPCG's ax_new).  evec_dev may be NULL: the Ritz vectors are
 * then not written (8 n m bytes less) -- the Davidson driver asks for them only in sweeps after which somebody reads them
 * (convergence in sight, restart, last sweep).
 * diaglib.f90:1717-1732 (Davidson, n_res = n_targ) and 420-442 (LOBPCG, n_res = n_max). */
int  dla_ritz_residual(dla_ctx* ctx, int n, int l, int m, const double* v_dev, const double* av_dev,
                       const double* y_host, int ldy, const double* eig, int n_res, const int* skip,
                       double* evec_dev, double* r_dev, double* avy_dev, double* rnorm);
/* The same sweep with k2 extra products: p_dev = V C2, ap_dev = AV C2 (C2: l x k2, host).  LOBPCG forms its new P block
 * from the same two panels the Ritz step reads (diaglib.f90:495-501: P = S cp, AP = AS cp); with this entry they are read
 * once.  k2 = 0: exactly dla_ritz_residual. */
int  dla_ritz_residual_p(dla_ctx* ctx, int n, int l, int m, const double* v_dev, const double* av_dev,
                         const double* y_host, int ldy, const double* eig, int n_res, const int* skip,
                         double* evec_dev, double* r_dev, double* avy_dev, double* rnorm,
                         int k2, const double* c2_host, int ldc2, double* p_dev, double* ap_dev);
/* The sweep with TWO coefficient sets: e_dev = V Y1, r_dev = AV Y2, then for i < n_res with skip[i]==0: r_i -= eig_i e_i and the
 * norms as above.  The residual blocks of the linear-response drivers are of this shape -- rp = (A+B) vp u+ - w (S-D) vm u-
 * (diaglib.f90:872-889, 1337-1353: two dgemms on two panels with two coefficient blocks, then the daxpy / dnrm2 loop).
 * t_work_dev, junk_dev: n x m scratch blocks (used only for shapes the one-sweep kernel does not take: m > 48). */
int  dla_ritz_residual2(dla_ctx* ctx, int n, int l, int m, const double* v_dev, const double* av_dev,
                        const double* y1_host, int ldy1, const double* y2_host, int ldy2, const double* eig, int n_res,
                        const int* skip, double* e_dev, double* r_dev, double* t_work_dev, double* junk_dev, double* rnorm);
/* Expansion step of the Davidson and LOBPCG drivers on the contiguous panels basis = [X | U] (n x (m+k)) and
 * abasis = [AX | AU]:   ortho_vs_x(X, U)  (diaglib.f90:1790, 358-366, 523-529),  AU = A U [+ shift U]  (the caller's matvec,
 * :1685, 394-397; daxpy :397)  and the projection --
 *   mode 0 (Davidson, :1691):  h(1:m+k, 1:k) = [X | U]^T AU, the new columns of the projected matrix;
 *   mode 1 (LOBPCG, :401-403): h = lower triangle of [X | U]^T [AX | AU], (m+k) x (m+k).
 * Same result as dla_ortho_vs_x + dla_call_matvec (+ dla_axpy) + dla_gram / dla_gram_lower.  With device-mode callbacks
 * that need no host wait (DLA_OPT_CALLBACK_ORDER 0 or 2) the three are enqueued back to back and the chain's report is
 * read at the projection's host wait: one wait per expansion instead of two.  When the chain did not end with the planned
 * launches (first call of a shape, or a block that needs more passes than the previous one of its width) the
 * orthogonalisation is completed and the operator and the projection are simply repeated on the finished block -- the
 * operator must therefore be a pure function of its input (it is called a second time for the same block then). */
int  dla_expand_project(dla_ctx* ctx, int mode, int n, int m, int k, double* basis_dev, double* abasis_dev,
                        dla_matvec_fn matvec, double shift, double* h_host, int ldh);
/* mode 3 = mode 1 for a block that is used once and then rebuilt -- LOBPCG's W block (diaglib.f90:518-529, 394-403).  The device
 * chain ends as soon as it holds S = X^T U and G = U^T U MEASURED on the stored block together with the converged factor T of G
 * (T^T G T = I): the reference's closing pass -- U -= X S (:3543-3544), one macro-iteration of ortho_cd with a near-identity factor
 * (:3256-3327) -- is then a matter of coefficients.  The finished block is [X | U_stored] p with p = [E ; T'] ((m + k) x k), formed on
 * the host from what the chain hands over: E = -S T R, T' = T R with the k x k factor R that makes the projected block orthonormal
 * (its Gram matrix is I - (S T)^T (S T): G was measured before the projection).  h_host is the projection for the FINISHED block
 * (D^T H D with D = [I E ; 0 T']); the caller multiplies every coefficient block by D before it forms products with the panel
 * (S [y | cp] = S_stored D [y | cp]).  dla_pending_block returns p of the last mode-3 / mode-4 call ([0 ; I] when nothing stayed
 * pending: host-driven loops, a chain that had to finish in memory); dla_pending_factor returns its T' part.
 * mode 4 = mode 0 for a block that STAYS in the basis (Davidson): the same, but h_host comes back RAW, for the stored block, and
 * dla_pending_block returns the chain's own [-S T ; T] -- the caller keeps the pending blocks of its whole basis in an upper-
 * triangular D and dla_basis_admit completes the closing pass against the FINISHED basis X_stored D (E = -D D^T S T: the stored
 * columns are not orthonormal, X_stored^T X_stored = (D D^T)^-1), records the block in D and turns the raw columns into those of
 * D^T H D; dla_basis_fold multiplies coefficient rows by D.  A block that stays in the basis must keep later device projections
 * against the stored columns effective (the block that comes out of a first projection can be as ill-conditioned as 1e7: what the
 * projection leaves behind has to stay below its smallest directions): the projection part stays pending only for max |S| < 1e-9,
 * the factor only for max |G - I| < 1e-8 -- otherwise the factor is applied in memory (U <- U T, nothing of X is read) or the chain
 * finishes the block there.  (mode 3: max |S| < 1e-4; the block is rebuilt in the next iteration.)  *applied = 1 (mode 4, for
 * 1e-9 <= max |S| < 1e-5): the chain's closing sweep HAS applied [-S T ; T] to the block in memory, without measuring anything behind
 * it -- the block in memory is what h_host belongs to, and dla_basis_admit owes it only the difference to the exact closing pass and
 * the k x k factor of its Gram matrix I - (S T)^T (S T).
 * mode 5 = mode 4 with the caller's D kept on the DEVICE as well (dla_basis_sync after every block, blocks of at most 16 columns, at
 * most 320 basis columns): every projection of the chain is then X (D D^T) X^T U, exact against the finished basis, and the bounds
 * on what may stay pending are those of the host algebra alone -- max |S| < 0.05 with column sums of squares below 0.02, nothing on
 * G (its factor has converged) -- so the chain ends behind the first sweep that has measured S and G on what it stored.  Where the
 * device-driven chain cannot project exactly (an all-reduce hook, the A/B knobs for the host loop, a block wider than 16 columns,
 * an engine whose LDS limit went down after a refused request -- which can happen between two calls of one solve) the block is
 * finished in memory and dla_pending_block answers [0 ; I]: by mode 0 when the copy of D says the m stored columns are finished
 * (D = I there), by the host-driven loop with every X^T U multiplied by D D^T when they are not -- never by a plain projection
 * against unfinished columns.  The call fails (DLA_ERR_ARG) when the copy does not describe the m columns in front of the block.
 * The engine's HOST copy of D has no width limit: a basis with pending blocks that grows beyond the 320 columns of the device copy
 * goes on with blocks finished in memory by the host-driven loop (slower: one host wait per operation; exact).
 * mode 6 = the block is finished in memory against the finished basis X D whatever the device could do (host-driven loop with D D^T),
 * operator and projection on the result, nothing pending: the way out for a caller whose dla_basis_admit answered DLA_ERR_ORTHO (the
 * closing factor I - F^T F of the pending block was not positive definite) -- called on the same block, which is still what the
 * chain stored.  The operator is called a second time for that block.
 * (Mode 4's bounds cost later blocks a projection each time they remove more than 1e-8 from a block -- every chain of such a basis ends
 * on a measured product below that --, so the drivers use mode 5 where it is available and mode 0 elsewhere; mode 4 stays for callers
 * that keep D on the host only.) */
int  dla_pending_factor(dla_ctx* ctx, int k, double* t_host, int ldt);
int  dla_pending_block(dla_ctx* ctx, int m, int k, double* p_host, int ldp, int* applied);
/* Host-size algebra of a basis with pending blocks (no device work; all arrays column-major, leading dimension ld):
 * dla_basis_admit: a block of k columns came in behind m stored ones with the chain's pending block p (in: [-S T ; T], out: the
 * completed [E ; T']); columns m .. m+k-1 of h hold the raw product [X | U]_stored^T A U_stored (reference :1691 on the stored
 * block).  Records them in hraw (both triangles) and p in the upper-triangular dmat, and replaces the columns of h by those of
 * dmat^T hraw dmat.  dla_basis_fold: c <- dmat(0:rows,0:rows) c. */
int  dla_basis_admit(int m, int k, double* p_host, int ldp, int applied, double* hraw, double* dmat, double* h, int ld);
int  dla_basis_fold(int rows, int ncol, const double* dmat, int ld, double* c, int ldc);
/* The engine's copy of the caller's D for dla_expand_project mode 5 (on the device for the first 320 columns, on the host for all of
 * them): columns m .. m+k-1 of dmat (upper triangular, leading dimension ld; rows 0 .. m+k-1 are read) follow the m columns sent so
 * far -- blocks arrive in order, every block of the basis, identity ones included; k <= 0 forgets everything (new solve, restart).
 * Asynchronous: the host array may be changed when the call returns. */
int  dla_basis_sync(dla_ctx* ctx, int m, int k, const double* dmat, int ld);
/* The expansion step with a metric B (gen_david_driver diaglib.f90:2170-2190, lobpcg_driver with gen_eig :523-529): on
 * basis = [X | U], bbasis = [BX | BU], abasis = [AX | AU]:  b_ortho_vs_x(X, BX, U) (:3576-3663),  BU = B U (the caller's bvec),
 * b_ortho(U, BU) (:3094-3183),  AU = A U [+ shift U],  and the projection as in dla_expand_project (mode 0 / 1).  Same result as
 * the separate entry points.  With device-mode callbacks that may run ahead (DLA_OPT_RUN_AHEAD) all of it is enqueued behind
 * the orthogonalisation chain -- the k x k factorisation of b_ortho runs on the device and only behind a chain that ended
 * well -- and one host wait serves the whole step (three with the separate calls).
 * "Same result": the same block as the separate calls up to rounding -- b_ortho's Cholesky-QR gives the same Q for U and for U W
 * (W upper triangular, positive diagonal), so inside this entry the device chain does not apply its last pending factor
 * (16 n k bytes less); dla_b_ortho_vs_x called on its own does.
 * mode 2: b_ortho_vs_x, bvec, b_ortho and nothing else (abasis, matvec, h_host unused, may be NULL): the expansion of
 * caslr_eff_driver, whose new blocks are orthogonalised against the basis in the metric (A+B) resp. (A-B), get their metric
 * image and are made orthonormal in it (diaglib.f90:1397-1424) -- bvec is apbmul resp. ambmul there. */
int  dla_expand_project_metric(dla_ctx* ctx, int mode, int n, int m, int k, double* basis_dev, double* bbasis_dev, double* abasis_dev,
                               dla_matvec_fn matvec, dla_matvec_fn bvec, double shift, double* h_host, int ldh);
/* y += alpha x over len contiguous doubles.  daxpy at diaglib.f90:312,397 (LOBPCG shift). */
int  dla_axpy(dla_ctx* ctx, size_t len, double alpha, const double* x_dev, double* y_dev);
/* sqrt(sum x^2) over len contiguous doubles (all ranks).  dnrm2 at diaglib.f90:3749, 3268. */
int  dla_nrm2(dla_ctx* ctx, size_t len, const double* x_dev, double* out);
/* measurement aid (SURVEY 8d "measured device-triad GB/s as the practical ceiling"): STREAM triad a = b + s c on three
 * scratch arrays of len doubles with the sweeps' access shape, best of reps repetitions, in GB/s (24 bytes per element) */
int  dla_stream_triad(dla_ctx* ctx, size_t len, int reps, double* gbps);
/* fill evec(n x m) with the documented counter-based uniform [0,1) stream (replaces the
 * compiler-specific random_number at diaglib.f90:3754). */
int  dla_random_fill(dla_ctx* ctx, int n, int m, double* evec_dev);
/* evec(i,j) = u01(seed, global row i, j) - 0.5: the "hard" benchmark guess (SURVEY.md 8d guess (b), seed 2); the role of
 * guess_evec mode 4 in the reference harness (main.f90:1362-1367), with a generator that does not depend on the compiler.
 * support_rows > 0 restricts the random entries to the leading support_rows GLOBAL rows (zero below): on the benchmark
 * operator (diagonal 2..n+1) a guess spread over all of n >= 1e6 rows starts at Rayleigh quotients ~n/2 and needs
 * thousands of iterations (measured: 400 iterations at n = 1e7 leave the lowest Ritz value at 4.4e6), so the large
 * restart/locking runs confine it to as many rows as the reference's own toy problem has (main.f90:14: n = 1000..2000). */
int  dla_fill_guess(dla_ctx* ctx, int n, int m, double* evec_dev, unsigned long long seed, long long support_rows);

/* ---------------------------------------------------------------- orthogonalisation */
/* ortho_cd(n,m,u,growth,ok), diaglib.f90:3185-3341: Cholesky-QR with refinement and the level-shift ladder; growth = product of
 * the norm estimates of the inverse factors, ok = 0 when the iteration cap was reached (the reference prints and returns) */
int  dla_ortho_cd(dla_ctx* ctx, int n, int k, double* u_dev, double* growth, int* ok);
/* Fused sweeps of the orthogonalisation loops (one pass over the panel instead of two), as the engine runs them inside
 * ortho_cd / ortho_vs_x; entry points of their own so that each can be checked against the BLAS pair it replaces:
 *   dla_trmm_gram    U <- U W (k x k, general W; dtrmm at diaglib.f90:3327)          and G = U^T U of the result (:3256)
 *   dla_update_gram  U <- U - X C (dgemm at :3544)                                   and G = U^T U of the result
 *   dla_combo_gram   U <- [X | U] C' (C' is (m+k) x k; U follows X in one panel, k <= 48) and G = U^T U of the result:
 *                    the pending triangular factor folded into the projection, X^T (U W) = (X^T U) W
 * G (k x k, host, ld = ldg) comes back complete (symmetric). */
int  dla_trmm_gram(dla_ctx* ctx, int n, int k, double* u_dev, const double* w_host, int ldw, double* g_host, int ldg);
int  dla_update_gram(dla_ctx* ctx, int n, int l, const double* x_dev, int k, const double* c_host, int ldc, double* u_dev,
                     double* g_host, int ldg);
int  dla_combo_gram(dla_ctx* ctx, int n, int m, const double* x_dev, int k, const double* c_host, int ldc, double* u_dev,
                    double* g_host, int ldg);
/* ortho(n,m,u,w), diaglib.f90:3052-3092: Householder QR of a copy (dgeqrf) and U <- U R^-1 (dtrsm) -- the orthonormal factor
 * with LAPACK's sign convention for diag(R).  The reference calls it when ortho_cd gives up (:3534, :3549).  Device path:
 * column-wise Gram-Schmidt for the factor, the Householder recurrence on an isometric 2k x k host matrix for the signs. */
int  dla_ortho_qr(dla_ctx* ctx, int n, int k, double* u_dev);
int  dla_ortho_vs_x(dla_ctx* ctx, int n, int m, int k, const double* x_dev, double* u_dev); /* diaglib.f90:3481-3574 */
int  dla_b_ortho(dla_ctx* ctx, int n, int m, double* u_dev, double* bu_dev);                /* diaglib.f90:3094-3183 */
int  dla_b_ortho_vs_x(dla_ctx* ctx, int n, int m, int k, const double* x_dev, const double* bx_dev,
                      double* u_dev);                                                       /* diaglib.f90:3576-3663 */
int  dla_check_guess(dla_ctx* ctx, int n, int m, double* evec_dev);                         /* diaglib.f90:3734-3786 */
/* host-size coefficient step of LOBPCG, diaglib.f90:3686-3732 (arrays on the host) */
int  dla_get_coeffs(dla_ctx* ctx, int len_a, int len_u, int n_max, int n_act, const double* a_red,
                    double* u_x, double* u_p);

/* ---------------------------------------------------------------- user callbacks
 * call matvec(n,m,x,ax) / precnd(n,m,fac,x,px) on blocks of device panels
 * (diaglib.f90:1685,1786, 309,352,394,518).  Host-callback mode stages through pinned memory. */
int  dla_call_matvec(dla_ctx* ctx, dla_matvec_fn fn, int n, int m, const double* x_dev, double* ax_dev);
int  dla_call_precnd(dla_ctx* ctx, dla_precnd_fn fn, int n, int m, double fac, const double* x_dev, double* px_dev);
/* call lrprec(n,m,fac,xp,xm,yp,ym) (diaglib.f90:1317) on two device blocks, staging through pinned memory in host mode */
int  dla_call_lrprec(dla_ctx* ctx, dla_lrprec_fn fn, int n, int m, double fac, const double* xp_dev, const double* xm_dev,
                     double* yp_dev, double* ym_dev);

/* ---------------------------------------------------------------- small dense, host (LAPACK in the reference) */
int    dla_syev(char uplo, int n, double* a, int lda, double* w);   /* dsyev('v',uplo): :315,406,1708 */
int    dla_syev_lowest(char uplo, int n, double* a, int lda, double* w, int m); /* same call sites: only the m
                                                                       lowest eigenPAIRS are formed (all the drivers use) */
int    dla_potrf_lower(int m, double* a, int lda);                  /* dpotrf('l'): :3261             */
int    dla_trtri_lower(int m, double* a, int lda);                  /* dtrtri('l','n'): :3310         */
double dla_norm_est(int m, const double* a, int lda);               /* norm_est: :3447-3479           */

/* ---------------------------------------------------------------- built-in device operator (bench / tests)
 * A = diag(i+1) + sigma W W^T, W(i,j) = (2 u01(1,i,j)-1)/sqrt(i), i = 1-based global row;
 * preconditioner = main.f90:146-171 (mprec) semantics.  The two callbacks have the reference's shape
 * and expect DEVICE addresses (use with DLA_OPT_CALLBACKS_ON_DEVICE = 1). */
int  dla_synth_setup(dla_ctx* ctx, long long n_global, long long row0, int n_local, int rank_w, double sigma);
void dla_synth_matvec(const int* n, const int* m, const double* x_dev, double* ax_dev);
void dla_synth_precnd(const int* n, const int* m, const double* fac, const double* x_dev, double* px_dev);
/* Sample operators for the linear-response and generalised drivers around the same W (after dla_synth_setup; device addresses,
 * reference callback shapes: apbmul / ambmul / spdmul / smdmul of diaglib.f90:1024-1025, bvec of :1855, lrprec of :1317):
 *   y = d(i) x + W C W^T x :  A+B: d = i+5, C = sigma I;  A-B: d = i+2, C = 0.2 sigma I (the harness' diagonals, main.f90:563,570);
 *   S+D / S-D: d = s(i) = 1 + 0.5/(1 + i mod 7), C = +-0.05 J with J antisymmetric (D = 0.05 W J W^T);  metric: d = s(i), C = 0.1 I.
 * lrprec1 / lrprec2 = the harness' lrprec_1 / lrprec_2 (main.f90:234-281) on the diagonals of these operators. */
void dla_synth_apbmul(const int* n, const int* m, const double* x_dev, double* y_dev);
void dla_synth_ambmul(const int* n, const int* m, const double* x_dev, double* y_dev);
void dla_synth_spdmul(const int* n, const int* m, const double* x_dev, double* y_dev);
void dla_synth_smdmul(const int* n, const int* m, const double* x_dev, double* y_dev);
void dla_synth_metric(const int* n, const int* m, const double* x_dev, double* y_dev);
void dla_synth_lrprec1(const int* n, const int* m, const double* fac, const double* xp_dev, const double* xm_dev, double* yp_dev, double* ym_dev);
void dla_synth_lrprec2(const int* n, const int* m, const double* fac, const double* xp_dev, const double* xm_dev, double* yp_dev, double* ym_dev);

/* ---------------------------------------------------------------- drivers (Fortran, bind(C) twins of the
 * module procedures davidson_driver / gen_david_driver / lobpcg_driver; argument meaning as reference
 * diaglib.f90:1483-1539 and 171-228; logicals as int 0/1) */
void dla_davidson_driver(int verbose, int n, int n_targ, int n_max, int max_iter, double tol, int max_dav,
                         double shift, dla_matvec_fn matvec, dla_precnd_fn precnd,
                         double* eig, double* evec, int* ok);
void dla_gen_david_driver(int verbose, int n, int n_targ, int n_max, int max_iter, double tol, int max_dav,
                          double shift, dla_matvec_fn matvec, dla_precnd_fn precnd, dla_matvec_fn bvec,
                          double* eig, double* evec, int* ok);      /* reference diaglib.f90:1855-2250 */
void dla_lobpcg_driver(int verbose, int gen_eig, int n, int n_targ, int n_max, int max_iter, double tol,
                       double shift, dla_matvec_fn matvec, dla_precnd_fn precnd, dla_matvec_fn bvec,
                       double* eig, double* evec, int* ok);
/* linear-response problem (A B; B A)(Y Z) = w (S D; -D -S)(Y Z): reference diaglib.f90:1024-1481 (caslr_eff_driver).
 * evec is 2n x n_max (Y on top of Z); the four operators have the matvec shape and apply A+B, A-B, S+D, S-D. */
void dla_caslr_eff_driver(int verbose, int n, int n_targ, int n_max, int max_iter, double tol, int max_dav,
                          dla_matvec_fn apbmul, dla_matvec_fn ambmul, dla_matvec_fn spdmul, dla_matvec_fn smdmul,
                          dla_lrprec_fn lrprec, double* eig, double* evec, int* ok);
/* same problem, traditional solver: reference diaglib.f90:558-1022 (caslr_driver, its default algorithm i_alg = 0) */
void dla_caslr_driver(int verbose, int n, int n_targ, int n_max, int max_iter, double tol, int max_dav,
                      dla_matvec_fn apbmul, dla_matvec_fn ambmul, dla_matvec_fn spdmul, dla_matvec_fn smdmul,
                      dla_lrprec_fn lrprec, double* eig, double* evec, int* ok);
/* iteration report of the last driver call (iterations, matvec columns, restarts) */
void dla_last_solve_info(int* iters, int* matvec_cols, int* restarts);
void dla_set_solve_info(int iters, int matvec_cols, int restarts);   /* used by the Fortran drivers */

/* ---------------------------------------------------------------- sample sparse operator
 * A device-resident operator for callers whose matrix is sparse and symmetric: hand A over once in CSR form (host arrays,
 * 0-based, 64-bit row pointers); it is kept on the device as column-major ELLPACK.  dla_spmm_matvec / dla_spmm_precnd have the
 * reference's callback shapes matvec(n,m,x,ax) / precnd(n,m,fac,x,px) (README.md:34-35, main.f90:72-90, 146-171) and expect
 * DEVICE addresses (DLA_OPT_CALLBACKS_ON_DEVICE = 1); the preconditioner is the harness' x / (a_ii + fac).  They act on the
 * operator the calling thread set up last and enqueue on that context's stream.
 * Row shards (dla_spmm_setup_csr_sharded): every rank hands over ITS rows row0 .. row0 + n_local - 1 with GLOBAL 64-bit column
 * indices.  The columns of a shard may reach at most `halo` rows into the two neighbouring shards (a banded matrix; halo is the
 * largest reach of any rank, agreed at setup, at most 4096 and at most a shard's height); each product then exchanges the first
 * and last `halo` rows of every rank's x block through the SAME transport as the small products -- every rank fills its own two
 * slots of a zeroed nranks x 2 x halo x m buffer and the all-reduce sum gathers them (all-reduce only; 2 halo m
 * doubles per rank and call) -- and a matrix with longer-range couplings is refused.  Collective: every rank calls the setup,
 * with shards contiguous in rank order; with one rank it equals dla_spmm_setup_csr. */
int  dla_spmm_setup_csr(dla_ctx* ctx, int n, const long long* rowptr, const int* colind, const double* values);
int  dla_spmm_setup_csr_sharded(dla_ctx* ctx, int n_local, long long row0, long long n_global, const long long* rowptr,
                                const long long* colind_global, const double* values);
void dla_spmm_matvec(const int* n, const int* m, const double* x_dev, double* ax_dev);
void dla_spmm_precnd(const int* n, const int* m, const double* fac, const double* x_dev, double* px_dev);

#ifdef __cplusplus
}
#endif
#endif
