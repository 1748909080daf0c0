/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the algorithm on diaglib's Davidson-Liu / LOBPCG
 * hot path (reference: Molecolab-Pisa/diaglib, diaglib.f90).  It exists so that the
 * HIP product path can be checked against it; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (diaglib_amd/) never
 * links, imports or calls anything in this directory.
 *
 * Parity status: PINNED.  oracle/Makefile also compiles the unmodified reference
 * (flang + MKL, from /root/reference where it lies) into oracle/_ref/ and
 * tests/golden/make_golden.py records reference outputs as fixtures; tests/test_oracle.py
 * checks every routine below against those fixtures.
 *
 * All matrices are column-major with an explicit leading dimension, as in the
 * reference.
 */
#ifndef DIAGLIB_ORACLE_H
#define DIAGLIB_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* reference callback shapes: README.md:34-35, callers main.f90:72-90,146-171 */
typedef void (*orc_matvec_t)(const int* n, const int* m, const double* x, double* ax);
typedef void (*orc_precnd_t)(const int* n, const int* m, const double* fac, const double* x, double* px);

/* per-iteration trace written by the two drivers (all arrays caller-allocated,
 * sized for max_iter iterations; may be NULL) */
typedef struct {
  int     iters;        /* iterations performed                               */
  int     matvec_cols;  /* total columns handed to matvec                     */
  int     restarts;     /* davidson restarts                                  */
  int*    n_act;        /* [max_iter]   active block width used in iteration  */
  int*    ldu;          /* [max_iter]   subspace size diagonalised            */
  double* eig;          /* [max_iter*n_targ]                                  */
  double* rms;          /* [max_iter*n_targ]                                  */
  double* rmax;         /* [max_iter*n_targ]                                  */
  int*    done;         /* [max_iter*n_targ]                                  */
} orc_trace;

/* ---- small dense (what the reference gets from LAPACK) ---- */
int    orc_potrf_lower(int m, double* a, int lda);          /* dpotrf('l'): 0 ok, j>0 failed at column j */
int    orc_trtri_lower(int m, double* a, int lda);          /* dtrtri('l','n') */
int    orc_syev(char uplo, int n, double* a, int lda, double* w); /* dsyev('v',uplo): vectors overwrite a */
double orc_norm_est(int m, const double* a, int lda);       /* diaglib.f90:3447-3479 */

/* ---- tall-skinny panel algebra (what the reference gets from BLAS) ---- */
void orc_gemm_tn(int n, int l, int k, const double* x, int ldx, const double* u, int ldu, double* c, int ldc);
void orc_gemm_nn(int n, int l, int k, double alpha, const double* x, int ldx, const double* c, int ldc,
                 double beta, double* z, int ldz);

/* ---- orthogonalisation kernels ---- */
void orc_ortho_cd(int n, int m, double* u, double* growth, int* ok, int* n_macro);      /* diaglib.f90:3185-3341 */
void orc_ortho_qr(int n, int m, double* u);                                             /* diaglib.f90:3052-3092 */
int  orc_ortho_vs_x(int n, int m, int k, const double* x, double* u, int* n_outer);     /* diaglib.f90:3481-3574 */
void orc_b_ortho(int n, int m, double* u, double* bu);                                  /* diaglib.f90:3094-3183 */
int  orc_b_ortho_vs_x(int n, int m, int k, const double* x, const double* bx, double* u); /* diaglib.f90:3576-3663 */
void orc_check_guess(int n, int m, double* evec);                                       /* diaglib.f90:3734-3786 */
void orc_get_coeffs(int len_a, int len_u, int n_max, int n_act, const double* a_red,
                    double* u_x, double* u_p);                                          /* diaglib.f90:3686-3732 */

/* ---- drivers ---- */
void orc_davidson(int verbose, int n, int n_targ, int n_max, int max_iter, double tol, int max_dav,
                  double shift, orc_matvec_t matvec, orc_precnd_t precnd,
                  double* eig, double* evec, int* ok, orc_trace* tr);                   /* diaglib.f90:1483-1853 */
void orc_lobpcg(int verbose, int n, int n_targ, int n_max, int max_iter, double tol,
                double shift, orc_matvec_t matvec, orc_precnd_t precnd,
                double* eig, double* evec, int* ok, orc_trace* tr);                     /* diaglib.f90:171-556, gen_eig=.false. */

void orc_lobpcg_gen(int verbose, int n, int n_targ, int n_max, int max_iter, double tol,
                    double shift, orc_matvec_t matvec, orc_precnd_t precnd, orc_matvec_t bvec,
                    double* eig, double* evec, int* ok, orc_trace* tr);                 /* diaglib.f90:171-556, gen_eig=.true. */
void orc_gen_davidson(int verbose, int n, int n_targ, int n_max, int max_iter, double tol, int max_dav,
                      double shift, orc_matvec_t matvec, orc_precnd_t precnd, orc_matvec_t bvec,
                      double* eig, double* evec, int* ok, orc_trace* tr);               /* diaglib.f90:1855-2250 */

/* ---- portable counter-based generator shared by oracle and product ---- */
double orc_u01(unsigned long long seed, unsigned long long i, unsigned long long j);    /* uniform [0,1) */

/* ---- synthetic matrix-free operator A = diag(i+1) + sigma*W*W^T ---- */
void orc_synth_setup(long long n_global, long long row0, int n_local, int rank_w, double sigma);
void orc_synth_matvec(const int* n, const int* m, const double* x, double* ax);
void orc_synth_precnd(const int* n, const int* m, const double* fac, const double* x, double* px);
const double* orc_synth_w(void);     /* n_local x rank_w, column-major */
const double* orc_synth_diag(void);  /* a_ii = d_i + sigma*|W_i|^2 */

/* ---- dense test matrix a_ii=i+1, a_ij=1/(i+j) (main.f90:311-317) ---- */
void orc_dense_setup(int n);
void orc_dense_matvec(const int* n, const int* m, const double* x, double* ax);
void orc_dense_precnd(const int* n, const int* m, const double* fac, const double* x, double* px);
/* dense SPD metric for the generalised problem: S = I + (0.5/8) G G^T, G n x 8, G(i,j) = u01(3,i,j) - 0.5
 * (the reference harness uses S = R^T R with R uniform random, main.f90:429-430: same role, but that one is
 * numerically singular; tests want a metric both solvers can factor) */
void orc_metric_setup(int n);
void orc_metric_matvec(const int* n, const int* m, const double* x, double* sx);
const double* orc_metric(void);

/* ---- linear-response test problem (roles of main.f90:528-600, portable generator) ---- */
typedef void (*orc_lrprec_t)(const int* n, const int* m, const double* fac, const double* xp, const double* xm,
                             double* yp, double* ym);
void orc_lr_setup(int n);
void orc_lr_apb(const int* n, const int* m, const double* x, double* y);   /* (A+B) x */
void orc_lr_amb(const int* n, const int* m, const double* x, double* y);   /* (A-B) x */
void orc_lr_spd(const int* n, const int* m, const double* x, double* y);   /* (S+D) x */
void orc_lr_smd(const int* n, const int* m, const double* x, double* y);   /* (S-D) x */
void orc_lr_prec(const int* n, const int* m, const double* fac, const double* xp, const double* xm, double* yp, double* ym);
void orc_lr_prec1(const int* n, const int* m, const double* fac, const double* xp, const double* xm, double* yp, double* ym);
const double* orc_lr_matrix(int which);                                     /* 0 A+B, 1 A-B, 2 S+D, 3 S-D; column-major */

#ifdef __cplusplus
}
#endif
#endif
