/*
 * oracle/oracle_ops.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The two test operators, written as reference-shaped callbacks
 * matvec(n,m,x,ax) / precnd(n,m,fac,x,px) (reference README.md:34-35):
 *
 *  - dense:  a_ii = i+1, a_ij = 1/(i+j)   (reference main.f90:311-317), applied
 *            column by column like main.f90:72-90 (mmult), preconditioned like
 *            main.f90:146-171 (mprec).
 *  - synth:  A = diag(d) + sigma W W^T, d_i = i+1, W(i,j) = (2 u01(1,i,j) - 1)/sqrt(i),
 *            i the 1-based GLOBAL row.  Matrix-free, rank_w columns.
 *            Preconditioner = mprec semantics on a_ii = d_i + sigma |W_i|^2.
 *
 * State is file-global, like the reference harness' module utils (utils.f90:1-9).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------- dense ---------------- */
static int     g_dn = 0;
static double* g_da = NULL;

void orc_dense_setup(int n)
{
  free(g_da);
  g_dn = n;
  g_da = (double*)malloc(sizeof(double) * (size_t)n * n);
  for (int i = 1; i <= n; ++i) {
    g_da[(size_t)(i - 1) * n + (i - 1)] = (double)i + 1.0;
    for (int j = 1; j < i; ++j) {
      double v = 1.0 / (double)(i + j);
      g_da[(size_t)(i - 1) * n + (j - 1)] = v;
      g_da[(size_t)(j - 1) * n + (i - 1)] = v;
    }
  }
}

void orc_dense_matvec(const int* pn, const int* pm, const double* x, double* ax)
{
  int n = *pn, m = *pm;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const double* row = g_da + (size_t)i * n; /* symmetric: row i == column i */
    for (int c = 0; c < m; ++c) {
      const double* xc = x + (size_t)c * n;
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += row[j] * xc[j];
      ax[(size_t)c * n + i] = s;
    }
  }
}

void orc_dense_precnd(const int* pn, const int* pm, const double* fac, const double* x, double* px)
{
  int n = *pn, m = *pm;
  for (int c = 0; c < m; ++c)
    for (int i = 0; i < n; ++i) {
      double den = g_da[(size_t)i * n + i] + *fac;
      size_t p = (size_t)c * n + i;
      px[p] = (fabs(den) > 1.0e-5) ? x[p] / den : x[p];
    }
}

/* ---------------- dense SPD metric (generalised problem) ---------------- */
static int     g_mn = 0;
static double* g_ms = NULL;

void orc_metric_setup(int n)
{
  free(g_ms);
  g_mn = n;
  g_ms = (double*)calloc((size_t)n * n, sizeof(double));
  const int kw = 8;   /* S = I + (0.5/kw) G G^T with G n x kw: SPD, condition number O(1 + n/kw) */
  double* g = (double*)malloc(sizeof(double) * (size_t)n * kw);
  for (int j = 0; j < kw; ++j)
    for (int i = 0; i < n; ++i) g[(size_t)j * n + i] = orc_u01(3ULL, (unsigned long long)(i + 1), (unsigned long long)(j + 1)) - 0.5;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int q = 0; q < kw; ++q) s += g[(size_t)q * n + i] * g[(size_t)q * n + j];
      g_ms[(size_t)j * n + i] = (0.5 / kw) * s + (i == j ? 1.0 : 0.0);
    }
  free(g);
}

const double* orc_metric(void) { return g_ms; }

void orc_metric_matvec(const int* pn, const int* pm, const double* x, double* sx)
{
  int n = *pn, m = *pm;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const double* row = g_ms + (size_t)i * n;
    for (int c = 0; c < m; ++c) {
      const double* xc = x + (size_t)c * n;
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += row[j] * xc[j];
      sx[(size_t)c * n + i] = s;
    }
  }
}

/* ---------------- linear-response test problem (same roles as main.f90:528-600) ----------------
 * A+B: diag 5+i, off-diagonal 0.2/(i+j);  A-B: diag 2+i;  S = I + (0.5/8) G G^T (the metric above, seed 5);
 * D antisymmetric, D(i,j) = 0.05 (u01(7,i,j) - u01(7,j,i)) for i < j.  The reference harness draws S and D
 * with the compiler's random_number; here they come from the portable generator so that every
 * implementation sees the same matrices.  Preconditioner: main.f90:257-281 (lrprec_2). */
static int     g_ln = 0;
static double *g_apb = NULL, *g_amb = NULL, *g_spd = NULL, *g_smd = NULL, *g_adiag = NULL, *g_sdiag = NULL;

void orc_lr_setup(int n)
{
  free(g_apb); free(g_amb); free(g_spd); free(g_smd); free(g_adiag); free(g_sdiag);
  g_ln = n;
  size_t nn = (size_t)n * n;
  g_apb = (double*)calloc(nn, sizeof(double)); g_amb = (double*)calloc(nn, sizeof(double));
  g_spd = (double*)calloc(nn, sizeof(double)); g_smd = (double*)calloc(nn, sizeof(double));
  g_adiag = (double*)malloc(sizeof(double) * n); g_sdiag = (double*)malloc(sizeof(double) * n);
  const int kw = 8;
  double* g = (double*)malloc(sizeof(double) * (size_t)n * kw);
  for (int j = 0; j < kw; ++j)
    for (int i = 0; i < n; ++i) g[(size_t)j * n + i] = orc_u01(5ULL, (unsigned long long)(i + 1), (unsigned long long)(j + 1)) - 0.5;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      size_t p = (size_t)j * n + i;
      int fi = i + 1, fj = j + 1;
      g_apb[p] = (i == j) ? 5.0 + fi : 0.2 / (double)(fi + fj);
      g_amb[p] = (i == j) ? 2.0 + fi : 0.0;
      double sg = 0.0;
      for (int q = 0; q < kw; ++q) sg += g[(size_t)q * n + i] * g[(size_t)q * n + j];
      sg = (0.5 / kw) * sg + (i == j ? 1.0 : 0.0);
      double dl = 0.0;
      if (i < j) dl = 0.05 * (orc_u01(7ULL, fi, fj) - orc_u01(7ULL, fj, fi));
      if (i > j) dl = -0.05 * (orc_u01(7ULL, fj, fi) - orc_u01(7ULL, fi, fj));
      g_spd[p] = sg + dl;
      g_smd[p] = sg - dl;
      if (i == j) { g_adiag[i] = 0.5 * (g_apb[p] + g_amb[p]); g_sdiag[i] = sg; }
    }
  free(g);
}

static void lr_apply(const double* mat, int n, int m, const double* x, double* y)
{
  /* y = mat x, column by column like main.f90:173-232 (matmul); mat is column-major */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < m; ++c) {
      const double* xc = x + (size_t)c * n;
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += mat[(size_t)j * n + i] * xc[j];
      y[(size_t)c * n + i] = s;
    }
}
void orc_lr_apb(const int* n, const int* m, const double* x, double* y) { lr_apply(g_apb, *n, *m, x, y); }
void orc_lr_amb(const int* n, const int* m, const double* x, double* y) { lr_apply(g_amb, *n, *m, x, y); }
void orc_lr_spd(const int* n, const int* m, const double* x, double* y) { lr_apply(g_spd, *n, *m, x, y); }
void orc_lr_smd(const int* n, const int* m, const double* x, double* y) { lr_apply(g_smd, *n, *m, x, y); }
const double* orc_lr_matrix(int which) { return which == 0 ? g_apb : which == 1 ? g_amb : which == 2 ? g_spd : g_smd; }

void orc_lr_prec(const int* pn, const int* pm, const double* pfac, const double* xp, const double* xm, double* yp, double* ym)
{
  const int n = *pn, m = *pm;
  const double fac = *pfac;
  for (int c = 0; c < m; ++c)
    for (int i = 0; i < n; ++i) {
      const size_t p = (size_t)c * n + i;
      const double den = 1.0 / (fac * fac * g_adiag[i] * g_adiag[i] - g_sdiag[i] * g_sdiag[i]);
      yp[p] = den * (fac * g_adiag[i] * xp[p] + g_sdiag[i] * xm[p]);
      ym[p] = den * (fac * g_adiag[i] * xm[p] + g_sdiag[i] * xp[p]);
    }
}

/* main.f90:234-255 (lrprec_1, used by the harness with caslr_driver) */
void orc_lr_prec1(const int* pn, const int* pm, const double* pfac, const double* xp, const double* xm, double* yp, double* ym)
{
  const int n = *pn, m = *pm;
  const double fac = *pfac;
  for (int c = 0; c < m; ++c)
    for (int i = 0; i < n; ++i) {
      const size_t p = (size_t)c * n + i;
      const double den = -1.0 / (g_adiag[i] * g_adiag[i] - fac * fac * g_sdiag[i] * g_sdiag[i]);
      yp[p] = den * (g_adiag[i] * xp[p] + fac * g_sdiag[i] * xm[p]);
      ym[p] = den * (g_adiag[i] * xm[p] + fac * g_sdiag[i] * xp[p]);
    }
}

/* ---------------- synthetic matrix-free ---------------- */
static long long g_row0 = 0;
static int       g_nl = 0, g_rw = 0;
static double    g_sigma = 0.0;
static double*   g_w = NULL;   /* n_local x rank_w */
static double*   g_diag = NULL;

void orc_synth_setup(long long n_global, long long row0, int n_local, int rank_w, double sigma)
{
  (void)n_global;
  free(g_w); free(g_diag);
  g_row0 = row0; g_nl = n_local; g_rw = rank_w; g_sigma = sigma;
  g_w = (double*)malloc(sizeof(double) * (size_t)n_local * rank_w);
  g_diag = (double*)malloc(sizeof(double) * (size_t)n_local);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_local; ++i) {
    unsigned long long gi = (unsigned long long)(row0 + i + 1);
    double s = 0.0;
    double inv = 1.0 / sqrt((double)gi);
    for (int j = 0; j < rank_w; ++j) {
      double h = 2.0 * orc_u01(1ULL, gi, (unsigned long long)(j + 1)) - 1.0;
      double w = h * inv;
      g_w[(size_t)j * n_local + i] = w;
      s += w * w;
    }
    g_diag[i] = ((double)gi + 1.0) + sigma * s;
  }
}

/* call counters of the synthetic operator: what a driver (the reference's, timed by bench.py's cpu_baseline leg) actually asked of
 * its callbacks -- calls and block columns of matvec and precnd since the last reset */
static long long g_cnt[4] = {0, 0, 0, 0};
void orc_synth_counters(int reset, long long* out4)
{
  if (out4) for (int i = 0; i < 4; ++i) out4[i] = g_cnt[i];
  if (reset) for (int i = 0; i < 4; ++i) g_cnt[i] = 0;
}

const double* orc_synth_w(void) { return g_w; }
const double* orc_synth_diag(void) { return g_diag; }

void orc_synth_matvec(const int* pn, const int* pm, const double* x, double* ax)
{
  int n = *pn, m = *pm;
  g_cnt[0] += 1; g_cnt[1] += m;
  double* t = (double*)malloc(sizeof(double) * (size_t)g_rw * m);
  orc_gemm_tn(n, g_rw, m, g_w, n, x, n, t, g_rw);   /* t = W^T x */
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m; ++c)
    for (int i = 0; i < n; ++i) {
      double d = (double)(g_row0 + i + 1) + 1.0;
      double s = 0.0;
      for (int q = 0; q < g_rw; ++q) s += g_w[(size_t)q * n + i] * t[q + (size_t)c * g_rw];
      ax[(size_t)c * n + i] = d * x[(size_t)c * n + i] + g_sigma * s;
    }
  free(t);
}

/* the product's sample metric (diaglib_amd/csrc/hip_engine.hip SynthKind 5, the operator the generalised multi-rank tests use):
 * y = s(i) x + 0.1 W W^T x with s(i) = 1 + 0.5 / (1 + i mod 7), i the 1-based global row -- symmetric positive definite; the
 * counterpart of the dense SPD metric the reference's harness builds for test_geneig (main.f90:403-526) */
void orc_synth_metric(const int* pn, const int* pm, const double* x, double* bx)
{
  int n = *pn, m = *pm;
  double* t = (double*)malloc(sizeof(double) * (size_t)g_rw * m);
  orc_gemm_tn(n, g_rw, m, g_w, n, x, n, t, g_rw);   /* t = W^T x */
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m; ++c)
    for (int i = 0; i < n; ++i) {
      unsigned long long gi = (unsigned long long)(g_row0 + i + 1);
      double d = 1.0 + 0.5 / (1.0 + (double)(gi % 7ULL));
      double s = 0.0;
      for (int q = 0; q < g_rw; ++q) s += g_w[(size_t)q * n + i] * t[q + (size_t)c * g_rw];
      bx[(size_t)c * n + i] = d * x[(size_t)c * n + i] + 0.1 * s;
    }
  free(t);
}

void orc_synth_precnd(const int* pn, const int* pm, const double* fac, const double* x, double* px)
{
  int n = *pn, m = *pm;
  g_cnt[2] += 1; g_cnt[3] += m;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m; ++c)
    for (int i = 0; i < n; ++i) {
      double den = g_diag[i] + *fac;
      size_t p = (size_t)c * n + i;
      px[p] = (fabs(den) > 1.0e-5) ? x[p] / den : x[p];
    }
}
