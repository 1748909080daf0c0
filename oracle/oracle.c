/*
 * oracle/oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of the Davidson-Liu / LOBPCG hot path of
 * Molecolab-Pisa/diaglib.  Each function cites the reference file:line whose
 * behaviour it follows.  BLAS/LAPACK are un-vendored externals of the reference
 * (Makefile:8 "-lblas -llapack"); their published semantics are restated here with
 * simple textbook algorithms (triple loops, Cholesky, cyclic Jacobi).
 *
 * Parity: PINNED against the compiled reference (oracle/_ref) through the
 * fixtures in tests/golden/ -- see tests/test_oracle.py.
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define A_(a, ld, i, j) ((a)[(size_t)(i) + (size_t)(j) * (size_t)(ld)])

static const double ORC_EPS = DBL_EPSILON;          /* epsilon(one)            */
#define ORC_TOL_ORTHO (2.0 * DBL_EPSILON)           /* diaglib.f90:151         */

/* ------------------------------------------------------------------------- */
/* small dense                                                               */
/* ------------------------------------------------------------------------- */

/* dpotrf('l'): call sites diaglib.f90:3173,3261,3290.  Only the lower triangle is
 * referenced/overwritten; returns the 1-based column at which a non-positive pivot
 * was met (LAPACK info>0), 0 on success. */
int orc_potrf_lower(int m, double* a, int lda)
{
  for (int j = 0; j < m; ++j) {
    double d = A_(a, lda, j, j);
    for (int p = 0; p < j; ++p) d -= A_(a, lda, j, p) * A_(a, lda, j, p);
    if (!(d > 0.0)) return j + 1;
    d = sqrt(d);
    A_(a, lda, j, j) = d;
    for (int i = j + 1; i < m; ++i) {
      double s = A_(a, lda, i, j);
      for (int p = 0; p < j; ++p) s -= A_(a, lda, i, p) * A_(a, lda, j, p);
      A_(a, lda, i, j) = s / d;
    }
  }
  return 0;
}

/* dtrtri('l','n'): call site diaglib.f90:3310.  In-place inverse of the lower
 * triangle; the strict upper triangle is left untouched.  Column j of X = L^-1 is
 * x_jj = 1/l_jj, x_ij = -(sum_{p=j}^{i-1} l_ip x_pj)/l_ii; columns are overwritten left
 * to right, so columns p>j still hold L when column j is formed. */
int orc_trtri_lower(int m, double* a, int lda)
{
  for (int j = 0; j < m; ++j)
    if (A_(a, lda, j, j) == 0.0) return j + 1;
  for (int j = 0; j < m; ++j) A_(a, lda, j, j) = 1.0 / A_(a, lda, j, j);
  double* col = (double*)malloc(sizeof(double) * (size_t)m);
  for (int j = 0; j < m; ++j) {
    col[j] = A_(a, lda, j, j);
    for (int i = j + 1; i < m; ++i) {
      double s = A_(a, lda, i, j) * col[j];
      for (int p = j + 1; p < i; ++p) s += A_(a, lda, i, p) * col[p];
      col[i] = -s * A_(a, lda, i, i);
    }
    for (int i = j; i < m; ++i) A_(a, lda, i, j) = col[i];
  }
  free(col);
  return 0;
}

/* dsyev('v',uplo): call sites diaglib.f90:315,406 ('l') and 1708 ('u').
 * Cyclic Jacobi on the symmetric matrix defined by the named triangle; eigenvalues
 * ascending in w, orthonormal eigenvectors in the columns of a. */
int orc_syev(char uplo, int n, double* a, int lda, double* w)
{
  if (n <= 0) return 0;
  double* s = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* v = (double*)calloc((size_t)n * n, sizeof(double));
  int up = (uplo == 'u' || uplo == 'U');
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      double x = up ? A_(a, lda, i, j) : A_(a, lda, j, i);
      s[i + (size_t)j * n] = x;
      s[j + (size_t)i * n] = x;
    }
  for (int i = 0; i < n; ++i) v[i + (size_t)i * n] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, dg = 0.0;
    for (int j = 0; j < n; ++j) {
      dg += s[j + (size_t)j * n] * s[j + (size_t)j * n];
      for (int i = 0; i < j; ++i) off += s[i + (size_t)j * n] * s[i + (size_t)j * n];
    }
    if (off == 0.0 || off <= 1e-34 * dg) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = s[p + (size_t)q * n];
        if (apq == 0.0) continue;
        double app = s[p + (size_t)p * n], aqq = s[q + (size_t)q * n];
        if (fabs(apq) < 1e-300) continue;
        double theta = (aqq - app) / (2.0 * apq);
        double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; ++k) { /* columns p,q */
          double skp = s[k + (size_t)p * n], skq = s[k + (size_t)q * n];
          s[k + (size_t)p * n] = c * skp - sn * skq;
          s[k + (size_t)q * n] = sn * skp + c * skq;
        }
        for (int k = 0; k < n; ++k) { /* rows p,q */
          double spk = s[p + (size_t)k * n], sqk = s[q + (size_t)k * n];
          s[p + (size_t)k * n] = c * spk - sn * sqk;
          s[q + (size_t)k * n] = sn * spk + c * sqk;
        }
        for (int k = 0; k < n; ++k) {
          double vkp = v[k + (size_t)p * n], vkq = v[k + (size_t)q * n];
          v[k + (size_t)p * n] = c * vkp - sn * vkq;
          v[k + (size_t)q * n] = sn * vkp + c * vkq;
        }
      }
  }
  /* sort ascending (selection sort on indices) */
  int* idx = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  for (int i = 0; i < n - 1; ++i) {
    int b = i;
    for (int j = i + 1; j < n; ++j)
      if (s[idx[j] + (size_t)idx[j] * n] < s[idx[b] + (size_t)idx[b] * n]) b = j;
    int t = idx[i]; idx[i] = idx[b]; idx[b] = t;
  }
  for (int j = 0; j < n; ++j) {
    w[j] = s[idx[j] + (size_t)idx[j] * n];
    for (int i = 0; i < n; ++i) A_(a, lda, i, j) = v[i + (size_t)idx[j] * n];
  }
  free(idx); free(s); free(v);
  return 0;
}

/* diaglib.f90:3447-3479: max|diag| + Frobenius norm of the strict lower triangle */
double orc_norm_est(int m, const double* a, int lda)
{
  double dn = 0.0, on = 0.0;
  for (int i = 0; i < m; ++i) {
    double x = fabs(A_(a, lda, i, i));
    if (x > dn) dn = x;
  }
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < i; ++j) on += A_(a, lda, i, j) * A_(a, lda, i, j);
  return dn + sqrt(on);
}

/* ------------------------------------------------------------------------- */
/* tall-skinny panel algebra                                                 */
/* ------------------------------------------------------------------------- */

/* dgemm('t','n',l,k,n,1,x,ldx,u,ldu,0,c,ldc): C = X^T U  (diaglib.f90:1691,3256,3543).
 * Rows are summed in fixed chunks of ORC_CHUNK so that the result does not depend
 * on the OpenMP thread count. */
#define ORC_CHUNK 2048
void orc_gemm_tn(int n, int l, int k, const double* x, int ldx, const double* u, int ldu, double* c, int ldc)
{
  int nchunk = (n + ORC_CHUNK - 1) / ORC_CHUNK;
  if (nchunk < 1) nchunk = 1;
  size_t lk = (size_t)l * k;
  double* part = (double*)calloc(lk * (size_t)nchunk, sizeof(double));
#pragma omp parallel for schedule(static)
  for (int ch = 0; ch < nchunk; ++ch) {
    int r0 = ch * ORC_CHUNK, r1 = r0 + ORC_CHUNK;
    if (r1 > n) r1 = n;
    double* p = part + lk * (size_t)ch;
    for (int j = 0; j < k; ++j) {
      const double* uj = u + (size_t)j * ldu;
      for (int i = 0; i < l; ++i) {
        const double* xi = x + (size_t)i * ldx;
        double s = 0.0;
        for (int r = r0; r < r1; ++r) s += xi[r] * uj[r];
        p[i + (size_t)j * l] = s;
      }
    }
  }
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < l; ++i) {
      double s = 0.0;
      for (int ch = 0; ch < nchunk; ++ch) s += part[lk * (size_t)ch + i + (size_t)j * l];
      A_(c, ldc, i, j) = s;
    }
  free(part);
}

/* dgemm('n','n',n,k,l,alpha,x,ldx,c,ldc,beta,z,ldz): Z = alpha X C + beta Z
 * (diaglib.f90:1717,1721,3544; beta==0 never reads Z, as in BLAS). */
void orc_gemm_nn(int n, int l, int k, double alpha, const double* x, int ldx, const double* c, int ldc,
                 double beta, double* z, int ldz)
{
#pragma omp parallel for schedule(static)
  for (int ch = 0; ch < (n + ORC_CHUNK - 1) / ORC_CHUNK; ++ch) {
    int r0 = ch * ORC_CHUNK, r1 = r0 + ORC_CHUNK;
    if (r1 > n) r1 = n;
    double acc[ORC_CHUNK];
    for (int j = 0; j < k; ++j) {
      for (int r = r0; r < r1; ++r) acc[r - r0] = 0.0;
      for (int p = 0; p < l; ++p) {
        double cpj = A_(c, ldc, p, j);
        const double* xp = x + (size_t)p * ldx;
        for (int r = r0; r < r1; ++r) acc[r - r0] += xp[r] * cpj;
      }
      double* zj = z + (size_t)j * ldz;
      if (beta == 0.0)
        for (int r = r0; r < r1; ++r) zj[r] = alpha * acc[r - r0];
      else
        for (int r = r0; r < r1; ++r) zj[r] = alpha * acc[r - r0] + beta * zj[r];
    }
  }
}

static double orc_nrm2(size_t len, const double* v)
{
  double s = 0.0;
  for (size_t i = 0; i < len; ++i) s += v[i] * v[i];
  return sqrt(s);
}

/* ------------------------------------------------------------------------- */
/* orthogonalisation                                                         */
/* ------------------------------------------------------------------------- */

/* diaglib.f90:3185-3341.  Cholesky orthonormalisation with iterative refinement.
 *   loop (at most 10 macro-iterations, else ok=false, 3248-3255):
 *     G = U^T U (3256); L = chol(G) (3261); on failure retry on G + shift*I with
 *     shift = max(eps*alpha*||U||_F, 2 eps), alpha = 100,1000,... (3265-3295);
 *     Linv = L^-1 (3310); growth *= norm_est(Linv) (3323); U <- U Linv^T (3327);
 *     done when eps*(norm_est(L)*norm_est(Linv))^2 < 2 eps (3331-3332). */
void orc_ortho_cd(int n, int m, double* u, double* growth, int* ok, int* n_macro)
{
  const int maxit = 10;
  double* metric = (double*)calloc((size_t)m * m, sizeof(double));
  double* msave  = (double*)calloc((size_t)m * m, sizeof(double));
  double* tmp    = (double*)malloc(sizeof(double) * (size_t)n * (size_t)m);
  int it = 0, macro_done = 0;
  *growth = 1.0;
  if (n_macro) *n_macro = 0;
  while (!macro_done) {
    it++;
    if (it > maxit) {
      *ok = 0;
      printf("  ortho_cd failed with the following error: maximum number of iterations reached.\n");
      goto out;
    }
    if (n_macro) *n_macro = it;
    orc_gemm_tn(n, m, m, u, n, u, n, metric, m);
    memcpy(msave, metric, sizeof(double) * (size_t)m * m);
    int info = orc_potrf_lower(m, metric, m);
    if (info != 0) {
      double alpha = 100.0;
      double unorm = orc_nrm2((size_t)n * m, u);
      int it_micro = 0, micro_done = 0;
      while (!micro_done) {
        it_micro++;
        if (it_micro > maxit) {
          *ok = 0;
          printf("  ortho_cd failed with the following error: maximum number of iterations for factorization reached.\n");
          goto out; /* reference: stop (3283) */
        }
        double shift = fmax(ORC_EPS * alpha * unorm, ORC_TOL_ORTHO);
        memcpy(metric, msave, sizeof(double) * (size_t)m * m);
        for (int i = 0; i < m; ++i) A_(metric, m, i, i) += shift; /* diag_shift, 3668-3684 */
        info = orc_potrf_lower(m, metric, m);
        alpha *= 10.0;
        micro_done = (info == 0);
      }
    }
    memcpy(msave, metric, sizeof(double) * (size_t)m * m);
    orc_trtri_lower(m, msave, m);
    double l_norm = orc_norm_est(m, metric, m);
    double linv_norm = orc_norm_est(m, msave, m);
    double rcond = l_norm * linv_norm;
    *growth *= linv_norm;
    /* dtrmm('r','l','t','n'): U <- U * Linv^T, i.e. U(:,j) = sum_{p<=j} U(:,p) Linv(j,p) */
    memcpy(tmp, u, sizeof(double) * (size_t)n * (size_t)m);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < m; ++j) {
      double* uj = u + (size_t)j * n;
      double ljj = A_(msave, m, j, j);
      for (int r = 0; r < n; ++r) uj[r] = tmp[(size_t)j * n + r] * ljj;
      for (int p = 0; p < j; ++p) {
        double ljp = A_(msave, m, j, p);
        const double* tp = tmp + (size_t)p * n;
        for (int r = 0; r < n; ++r) uj[r] += tp[r] * ljp;
      }
    }
    double error = ORC_EPS * rcond * rcond;
    macro_done = error < ORC_TOL_ORTHO;
  }
  *ok = 1;
out:
  free(metric); free(msave); free(tmp);
}

/* diaglib.f90:3052-3092: U <- V R^-1 with U = QR (Householder, LAPACK sign convention
 * r_jj = -sign(x_j)*||x||) and V a copy of the input U. */
void orc_ortho_qr(int n, int m, double* u)
{
  double* v = (double*)malloc(sizeof(double) * (size_t)n * m);
  double* r = (double*)calloc((size_t)m * m, sizeof(double));
  memcpy(v, u, sizeof(double) * (size_t)n * m);
  double* hv = (double*)malloc(sizeof(double) * (size_t)n);
  for (int j = 0; j < m && j < n; ++j) {
    double* cj = u + (size_t)j * n;
    double nrm = 0.0;
    for (int i = j; i < n; ++i) nrm += cj[i] * cj[i];
    nrm = sqrt(nrm);
    double alpha = cj[j];
    double beta = (alpha >= 0.0) ? -nrm : nrm;
    if (nrm == 0.0) { A_(r, m, j, j) = 0.0; continue; }
    /* v = x - beta e1, H = I - 2 v v^T / (v^T v) */
    for (int i = 0; i < n; ++i) hv[i] = (i < j) ? 0.0 : cj[i];
    hv[j] = alpha - beta;
    double vtv = 0.0;
    for (int i = j; i < n; ++i) vtv += hv[i] * hv[i];
    for (int c = j; c < m; ++c) {
      double* cc = u + (size_t)c * n;
      double dot = 0.0;
      for (int i = j; i < n; ++i) dot += hv[i] * cc[i];
      double f = 2.0 * dot / vtv;
      for (int i = j; i < n; ++i) cc[i] -= f * hv[i];
    }
    for (int c = j; c < m; ++c) A_(r, m, j, c) = A_(u, n, j, c);
  }
  /* dtrsm('r','u','n','n'): solve X R = V */
  for (int j = 0; j < m; ++j) {
    double* xj = v + (size_t)j * n;
    for (int p = 0; p < j; ++p) {
      double rpj = A_(r, m, p, j);
      const double* xp = v + (size_t)p * n;
      for (int i = 0; i < n; ++i) xj[i] -= xp[i] * rpj;
    }
    double rjj = A_(r, m, j, j);
    for (int i = 0; i < n; ++i) xj[i] /= rjj;
  }
  memcpy(u, v, sizeof(double) * (size_t)n * m);
  free(v); free(r); free(hv);
}

/* diaglib.f90:3481-3574.  ortho_cd(U); repeat { U -= X (X^T U); ortho_cd(U) } until
 * growth_of_last_ortho_cd * eps < 2 eps (3562-3564); QR fallback when ortho_cd fails
 * (3534,3549, with the explicit ||X^T U|| test 3558-3560); abort after 10 passes (3568).
 * Returns 0, or -1 for the reference's "catastrophic failure" stop. */
int orc_ortho_vs_x(int n, int m, int k, const double* x, double* u, int* n_outer)
{
  const int maxit = 10;
  int ok = 0, done = 0, it = 0;
  double growth = 1.0, xu_norm;
  double* xu = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1) * k);
  orc_ortho_cd(n, k, u, &growth, &ok, NULL);
  if (!ok) orc_ortho_qr(n, k, u);
  while (!done) {
    it++;
    orc_gemm_tn(n, m, k, x, n, u, n, xu, m);
    orc_gemm_nn(n, m, k, -1.0, x, n, xu, m, 1.0, u, n);
    orc_ortho_cd(n, k, u, &growth, &ok, NULL);
    if (!ok) orc_ortho_qr(n, k, u);
    if (!ok) {
      orc_gemm_tn(n, m, k, x, n, u, n, xu, m);
      xu_norm = orc_nrm2((size_t)m * k, xu);
    } else {
      xu_norm = growth * ORC_EPS;
    }
    done = xu_norm < ORC_TOL_ORTHO;
    if (it > maxit) { free(xu); if (n_outer) *n_outer = it; return -1; }
  }
  if (n_outer) *n_outer = it;
  free(xu);
  return 0;
}

/* diaglib.f90:3094-3183 (use_svd=.false. branch): M = U^T BU, L = chol(M) with no
 * failure handling, U <- U L^-T and BU <- BU L^-T by triangular solve (3177-3178). */
void orc_b_ortho(int n, int m, double* u, double* bu)
{
  double* metric = (double*)malloc(sizeof(double) * (size_t)m * m);
  orc_gemm_tn(n, m, m, u, n, bu, n, metric, m);
  orc_potrf_lower(m, metric, m);
  double* p2[2] = { u, bu };
  for (int w = 0; w < 2; ++w) {
    double* a = p2[w];
    /* X L^T = A  ->  X(:,j) = (A(:,j) - sum_{p<j} X(:,p) L(j,p)) / L(j,j) */
    for (int j = 0; j < m; ++j) {
      double* xj = a + (size_t)j * n;
      for (int p = 0; p < j; ++p) {
        double ljp = A_(metric, m, j, p);
        const double* xp = a + (size_t)p * n;
        for (int i = 0; i < n; ++i) xj[i] -= xp[i] * ljp;
      }
      double ljj = A_(metric, m, j, j);
      for (int i = 0; i < n; ++i) xj[i] /= ljj;
    }
  }
  free(metric);
}

/* diaglib.f90:3576-3663: as ortho_vs_x with xu = (BX)^T U (3632) and the Euclidean
 * ortho_cd (3637). */
int orc_b_ortho_vs_x(int n, int m, int k, const double* x, const double* bx, double* u)
{
  const int maxit = 10;
  int ok = 0, done = 0, it = 0;
  double growth = 1.0, xu_norm;
  double* xu = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1) * k);
  orc_ortho_cd(n, k, u, &growth, &ok, NULL);
  if (!ok) orc_ortho_qr(n, k, u);
  while (!done) {
    it++;
    orc_gemm_tn(n, m, k, bx, n, u, n, xu, m);
    orc_gemm_nn(n, m, k, -1.0, x, n, xu, m, 1.0, u, n);
    orc_ortho_cd(n, k, u, &growth, &ok, NULL);
    if (!ok) orc_ortho_qr(n, k, u);
    if (!ok) {
      orc_gemm_tn(n, m, k, bx, n, u, n, xu, m);
      xu_norm = orc_nrm2((size_t)m * k, xu);
    } else {
      xu_norm = growth * ORC_EPS;
    }
    done = xu_norm < ORC_TOL_ORTHO;
    if (it > maxit) { free(xu); return -1; }
  }
  free(xu);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* portable counter-based generator                                          */
/* ------------------------------------------------------------------------- */
/* The reference draws its random guess from the compiler's random_number
 * (diaglib.f90:3754), whose stream is compiler-specific.  Oracle and
 * product share this documented generator instead: splitmix64 finaliser of a linear
 * combination of (seed, i, j); 53 mantissa bits -> uniform [0,1). */
double orc_u01(unsigned long long seed, unsigned long long i, unsigned long long j)
{
  unsigned long long z = seed * 0x9E3779B97F4A7C15ULL + i * 0xBF58476D1CE4E5B9ULL + j * 0x94D049BB133111EBULL;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

/* diaglib.f90:3734-3786.  Zero guess -> random columns (generator above, seed 7,
 * 1-based (row,col)) + ortho_cd; otherwise re-orthonormalise unless the Gram matrix is
 * *exactly* the identity in the reference's sense: sum diag^2 / m == 1 and
 * sum offdiag^2 == 0 (3774). */
void orc_check_guess(int n, int m, double* evec)
{
  double growth; int ok;
  double fac = orc_nrm2((size_t)n * m, evec);
  if (fac == 0.0) {
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < n; ++i)
        A_(evec, n, i, j) = orc_u01(7ULL, (unsigned long long)(i + 1), (unsigned long long)(j + 1));
    orc_ortho_cd(n, m, evec, &growth, &ok, NULL);
  } else {
    double* ov = (double*)malloc(sizeof(double) * (size_t)m * m);
    orc_gemm_tn(n, m, m, evec, n, evec, n, ov, m);
    double dn = 0.0, on = 0.0;
    for (int i = 0; i < m; ++i) {
      dn += A_(ov, m, i, i) * A_(ov, m, i, i);
      for (int j = 0; j < i; ++j) on += A_(ov, m, j, i) * A_(ov, m, j, i);
    }
    dn /= (double)m;
    if (dn != 1.0 || on != 0.0) orc_ortho_cd(n, m, evec, &growth, &ok, NULL);
    free(ov);
  }
}

/* diaglib.f90:3686-3732 */
void orc_get_coeffs(int len_a, int len_u, int n_max, int n_act, const double* a_red, double* u_x, double* u_p)
{
  int off_x = n_max - n_act;
  for (int j = 0; j < n_max; ++j)
    for (int i = 0; i < len_u; ++i) A_(u_x, len_u, i, j) = A_(a_red, len_a, i, j);
  for (int j = 0; j < n_act; ++j)
    for (int i = 0; i < len_u; ++i) A_(u_p, len_u, i, j) = A_(u_x, len_u, i, off_x + j);
  for (int j = 0; j < n_act; ++j) A_(u_p, len_u, off_x + j, j) -= 1.0;
  orc_ortho_vs_x(len_u, n_max, n_act, u_x, u_p, NULL);
}

/* ------------------------------------------------------------------------- */
/* drivers                                                                   */
/* ------------------------------------------------------------------------- */

static void trace_put(orc_trace* tr, int it0, int n_targ, int n_act, int ldu, const double* eig, double shift,
                      const double* rn /* 2 x n_max */, const int* done)
{
  if (!tr) return;
  tr->iters = it0 + 1;
  if (tr->n_act) tr->n_act[it0] = n_act;
  if (tr->ldu) tr->ldu[it0] = ldu;
  for (int i = 0; i < n_targ; ++i) {
    if (tr->eig)  tr->eig[(size_t)it0 * n_targ + i]  = eig[i] - shift;
    if (tr->rms)  tr->rms[(size_t)it0 * n_targ + i]  = rn[2 * i];
    if (tr->rmax) tr->rmax[(size_t)it0 * n_targ + i] = rn[2 * i + 1];
    if (tr->done) tr->done[(size_t)it0 * n_targ + i] = done[i];
  }
}

static void print_iter(int it, int n_targ, const double* eig, double shift, const double* rn, const int* done)
{
  for (int i = 0; i < n_targ; ++i)
    printf("        %4d  %4d%24.12f%12.4E%12.4E  %c\n", it, i + 1, eig[i] - shift, rn[2 * i], rn[2 * i + 1],
           done[i] ? 'T' : 'F');
  printf("\n");
}

/* diaglib.f90:1483-1853.  Column indices below are 0-based; "c1" comments give the
 * reference's 1-based value. */
void orc_davidson(int verbose, int n, int n_targ, int n_max, int max_iter, double tol, int max_dav,
                  double shift, orc_matvec_t matvec, orc_precnd_t precnd,
                  double* eig, double* evec, int* ok, orc_trace* tr)
{
  const int min_dav = 10;                                  /* 1544 */
  int dim_dav = max_dav > min_dav ? max_dav : min_dav;     /* 1595 */
  int lda = dim_dav * n_max;                               /* 1596 */
  size_t nn = (size_t)n;
  double* space  = (double*)calloc(nn * lda, sizeof(double));  /* 1607,1632 */
  double* aspace = (double*)calloc(nn * lda, sizeof(double));
  double* r      = (double*)calloc(nn * n_max, sizeof(double));
  double* a_red  = (double*)calloc((size_t)lda * lda, sizeof(double));
  double* a_copy = (double*)calloc((size_t)lda * lda, sizeof(double));
  double* e_red  = (double*)calloc((size_t)lda, sizeof(double));
  double* rn     = (double*)calloc((size_t)2 * n_max, sizeof(double));
  int*    done   = (int*)calloc((size_t)n_max, sizeof(int));
  double sqrtn = sqrt((double)n), tol_rms = tol, tol_max = 10.0 * tol; /* 1622-1624 */
  *ok = 0;
  if (tr) { tr->iters = 0; tr->matvec_cols = 0; tr->restarts = 0; }

  orc_check_guess(n, n_max, evec);                          /* 1644 */
  memcpy(space, evec, sizeof(double) * nn * n_max);         /* 1648 */

  int n_act = n_max, ind = 0, i_beg = 0, m_dim = 1, ldu = 0, restart = 0, n_rst = 0, n_frozen = 0;
  (void)ind;
  if (verbose) printf("    Davidson-Liu iterations (tol=%10.2E):\n", tol);

  for (int it = 1; it <= max_iter; ++it) {
    ldu += n_act;                                           /* 1680 */
    int c0 = i_beg + n_rst;                                 /* c1: i_beg+n_rst */
    matvec(&n, &n_act, space + nn * c0, aspace + nn * c0);  /* 1685 */
    if (tr) tr->matvec_cols += n_act;
    /* a_red(1:ldu, c0:c0+n_act) = space(:,1:ldu)^T aspace(:,c0:)  (1691) */
    orc_gemm_tn(n, ldu, n_act, space, n, aspace + nn * c0, n, a_red + (size_t)lda * c0, lda);
    if (restart) {                                          /* 1696-1702 */
      for (int i = 0; i < n_rst; ++i) A_(a_red, lda, i, i) = e_red[i];
      restart = 0; n_rst = 0;
    }
    memcpy(a_copy, a_red, sizeof(double) * (size_t)lda * lda);  /* 1703 */
    orc_syev('u', ldu, a_copy, lda, e_red);                 /* 1708 */
    for (int i = 0; i < n_max; ++i) eig[i] = e_red[i];      /* 1715 */
    orc_gemm_nn(n, ldu, n_max, 1.0, space, n, a_copy, lda, 0.0, evec, n);   /* 1717 */
    orc_gemm_nn(n, ldu, n_max, 1.0, aspace, n, a_copy, lda, 0.0, r, n);     /* 1721 */
    for (int i = 0; i < n_targ; ++i) {                      /* 1723-1732 */
      if (done[i]) continue;
      double* ri = r + nn * i; const double* vi = evec + nn * i;
      double s = 0.0, mx = 0.0;
      for (size_t p = 0; p < nn; ++p) {
        ri[p] -= eig[i] * vi[p];
        s += ri[p] * ri[p];
        double a = fabs(ri[p]); if (a > mx) mx = a;
      }
      rn[2 * i] = sqrt(s) / sqrtn; rn[2 * i + 1] = mx;
    }
    for (int i = 0; i < n_targ; ++i) {                      /* 1737-1746 */
      if (done[i]) continue;
      done[i] = (rn[2 * i] < tol_rms) && (rn[2 * i + 1] < tol_max) && (it > 1);
      if (!done[i]) { for (int j = i + 1; j < n_max; ++j) done[j] = 0; break; }
    }
    trace_put(tr, it - 1, n_targ, n_act, ldu, eig, shift, rn, done);
    if (verbose) print_iter(it, n_targ, eig, shift, rn, done);
    int all = 1;
    for (int i = 0; i < n_targ; ++i) all = all && done[i];
    if (all) { *ok = 1; break; }                            /* 1757-1760 */

    if (m_dim < dim_dav) {                                  /* 1765 */
      m_dim++; i_beg += n_act; n_act = n_max; n_frozen = 0; /* 1773-1776 */
      for (int i = 0; i < n_targ; ++i) { if (done[i]) { n_act--; n_frozen++; } else break; }
      int ind0 = n_max - n_act;                             /* c1: ind = n_max-n_act+1 */
      double fac = -eig[ind0];
      precnd(&n, &n_act, &fac, r + nn * ind0, space + nn * i_beg);           /* 1786 */
      if (orc_ortho_vs_x(n, ldu, n_act, space, space + nn * i_beg, NULL)) { /* 1792 */
        printf(" catastrophic failure of ortho_vs_x\n"); break;
      }
    } else {                                                /* restart, 1795-1825 */
      if (verbose) printf("      Restarting davidson.\n");
      if (tr) tr->restarts++;
      n_act = n_max;
      memset(space, 0, sizeof(double) * nn * lda);
      memcpy(space, evec, sizeof(double) * nn * n_max);
      memset(aspace, 0, sizeof(double) * nn * lda);
      memset(a_red, 0, sizeof(double) * (size_t)lda * lda);
      ldu = 0; i_beg = 0; m_dim = 1; n_rst = 0;
      for (int i = 0; i < n_targ; ++i) { if (done[i]) n_rst++; else break; }
      restart = 1;
    }
    (void)n_frozen;
  }
  free(space); free(aspace); free(r); free(a_red); free(a_copy); free(e_red); free(rn); free(done);
}

/* diaglib.f90:171-556.  gen != 0 follows the gen_eig=.true. statements (bvec callback, bspace panel,
 * b_ortho / b_ortho_vs_x); gen == 0 is the standard problem. */
static void lobpcg_impl(int verbose, int gen, int n, int n_targ, int n_max, int max_iter, double tol,
                        double shift, orc_matvec_t matvec, orc_precnd_t precnd, orc_matvec_t bvec,
                        double* eig, double* evec, int* ok, orc_trace* tr)
{
  int len_a = 3 * n_max;                                    /* 258 */
  size_t nn = (size_t)n;
  double* space  = (double*)calloc(nn * len_a, sizeof(double));
  double* aspace = (double*)calloc(nn * len_a, sizeof(double));
  double* r      = (double*)calloc(nn * n_max, sizeof(double));
  double* x_new  = (double*)calloc(nn * n_max, sizeof(double));
  double* ax_new = (double*)calloc(nn * n_max, sizeof(double));
  double* bspace = (double*)calloc(nn * len_a, sizeof(double));   /* allocated also for gen=0 in the reference (259) */
  double* bx_new = (double*)calloc(nn * n_max, sizeof(double));
  double* a_red  = (double*)calloc((size_t)len_a * len_a, sizeof(double));
  double* e_red  = (double*)calloc((size_t)len_a, sizeof(double));
  double* rn     = (double*)calloc((size_t)2 * n_max, sizeof(double));
  int*    done   = (int*)calloc((size_t)n_max, sizeof(int));
  if (tr) { tr->iters = 0; tr->matvec_cols = 0; tr->restarts = 0; }

  orc_check_guess(n, n_max, evec);                          /* 295 */
  if (gen) {                                                /* 299-302 */
    bvec(&n, &n_max, evec, bx_new);
    orc_b_ortho(n, n_max, evec, bx_new);
  }
  memcpy(space, evec, sizeof(double) * nn * n_max);         /* 306 */
  if (gen) memcpy(bspace, bx_new, sizeof(double) * nn * n_max);   /* 307 */
  matvec(&n, &n_max, space, aspace);                        /* 309 */
  if (tr) tr->matvec_cols += n_max;
  if (shift != 0.0)                                         /* 312 */
    for (size_t p = 0; p < nn * n_max; ++p) aspace[p] += shift * space[p];
  orc_gemm_tn(n, n_max, n_max, space, n, aspace, n, a_red, len_a);          /* 313 */
  orc_syev('l', n_max, a_red, len_a, e_red);                /* 315 */
  for (int i = 0; i < n_max; ++i) eig[i] = e_red[i];
  orc_gemm_nn(n, n_max, n_max, 1.0, space, n, a_red, len_a, 0.0, evec, n);  /* 322-323 */
  memcpy(space, evec, sizeof(double) * nn * n_max);
  orc_gemm_nn(n, n_max, n_max, 1.0, aspace, n, a_red, len_a, 0.0, evec, n); /* 324-325 */
  memcpy(aspace, evec, sizeof(double) * nn * n_max);
  if (gen) {                                                /* 329-332 */
    orc_gemm_nn(n, n_max, n_max, 1.0, bspace, n, a_red, len_a, 0.0, evec, n);
    memcpy(bspace, evec, sizeof(double) * nn * n_max);
  }
  memcpy(r, aspace, sizeof(double) * nn * n_max);           /* 337 */
  {
    const double* xb = gen ? bspace : space;                /* 338-346 */
    for (int i = 0; i < n_max; ++i)
      for (size_t p = 0; p < nn; ++p) r[nn * i + p] -= eig[i] * xb[nn * i + p];
  }
  int ind_x = 0, ind_w = n_max, ind_p = 0;                  /* 350-351 */
  {
    double fac = shift - eig[ind_x];
    precnd(&n, &n_max, &fac, r + nn * ind_x, space + nn * ind_w);           /* 352 */
  }
  if (gen) {                                                /* 357-364 */
    orc_b_ortho_vs_x(n, n_max, n_max, space, bspace, space + nn * ind_w);
    bvec(&n, &n_max, space + nn * ind_w, bspace + nn * ind_w);
    orc_b_ortho(n, n_max, space + nn * ind_w, bspace + nn * ind_w);
  } else {
    orc_ortho_vs_x(n, n_max, n_max, space, space + nn * ind_w, NULL);       /* 366 */
  }

  double tol_rms = tol, tol_max = 10.0 * tol, sqrtn = sqrt((double)n);      /* 374-376 */
  *ok = 0;
  int n_act = n_max;
  if (verbose) printf("    LOBPCG iterations (tol=%10.2E):\n", tol);

  for (int it = 1; it <= max_iter; ++it) {
    matvec(&n, &n_act, space + nn * ind_w, aspace + nn * ind_w);            /* 394 */
    if (tr) tr->matvec_cols += n_act;
    if (shift != 0.0)                                       /* 397 */
      for (size_t p = 0; p < nn * n_act; ++p) aspace[nn * ind_w + p] += shift * space[nn * ind_w + p];
    int len_u = n_max + 2 * n_act;                          /* 401-402 */
    if (it == 1) len_u = 2 * n_max;
    orc_gemm_tn(n, len_u, len_u, space, n, aspace, n, a_red, len_a);        /* 403 */
    orc_syev('l', len_u, a_red, len_a, e_red);              /* 406 */
    for (int i = 0; i < n_max; ++i) eig[i] = e_red[i];      /* 416 */
    orc_gemm_nn(n, len_u, n_max, 1.0, space, n, a_red, len_a, 0.0, x_new, n);   /* 420 */
    orc_gemm_nn(n, len_u, n_max, 1.0, aspace, n, a_red, len_a, 0.0, ax_new, n); /* 421 */
    if (gen) orc_gemm_nn(n, len_u, n_max, 1.0, bspace, n, a_red, len_a, 0.0, bx_new, n); /* 423 */
    memcpy(r, ax_new, sizeof(double) * nn * n_max);         /* 428 */
    for (int i = 0; i < n_max; ++i) {                       /* 429-442 */
      if (done[i]) continue;
      double* ri = r + nn * i; const double* xi = (gen ? bx_new : x_new) + nn * i;
      double s = 0.0, mx = 0.0;
      for (size_t p = 0; p < nn; ++p) {
        ri[p] -= eig[i] * xi[p];
        s += ri[p] * ri[p];
        double a = fabs(ri[p]); if (a > mx) mx = a;
      }
      rn[2 * i] = sqrt(s) / sqrtn; rn[2 * i + 1] = mx;
    }
    for (int i = 0; i < n_max; ++i) {                       /* 446-455 */
      if (done[i]) continue;
      done[i] = (rn[2 * i] < tol_rms) && (rn[2 * i + 1] < tol_max) && (it > 1);
      if (!done[i]) { for (int j = i + 1; j < n_max; ++j) done[j] = 0; break; }
    }
    trace_put(tr, it - 1, n_targ, n_act, len_u, eig, shift, rn, done);
    if (verbose) print_iter(it, n_targ, eig, shift, rn, done);
    int all = 1;
    for (int i = 0; i < n_targ; ++i) all = all && done[i];
    if (all) {                                              /* 465-469 */
      memcpy(evec, x_new, sizeof(double) * nn * n_max);
      *ok = 1;
      break;
    }
    int cnt = 0;
    for (int i = 0; i < n_max; ++i) cnt += done[i] ? 1 : 0;
    n_act = n_max - cnt;                                    /* 475-478 */
    ind_x = n_max - n_act;
    ind_p = ind_x + n_act;
    ind_w = ind_p + n_act;
    double* u_x = (double*)malloc(sizeof(double) * (size_t)len_u * n_max);
    double* u_p = (double*)malloc(sizeof(double) * (size_t)len_u * (n_act > 0 ? n_act : 1));
    orc_get_coeffs(len_a, len_u, n_max, n_act, a_red, u_x, u_p);            /* 488 */
    /* p = space*u_p, ap = aspace*u_p, each formed in the evec scratch and then copied
     * into the P block (495-498) */
    orc_gemm_nn(n, len_u, n_act, 1.0, space, n, u_p, len_u, 0.0, evec, n);
    double* ptmp = (double*)malloc(sizeof(double) * nn * (n_act > 0 ? n_act : 1));
    memcpy(ptmp, evec, sizeof(double) * nn * n_act);
    orc_gemm_nn(n, len_u, n_act, 1.0, aspace, n, u_p, len_u, 0.0, evec, n);
    memcpy(space + nn * ind_p, ptmp, sizeof(double) * nn * n_act);
    memcpy(aspace + nn * ind_p, evec, sizeof(double) * nn * n_act);
    if (gen) {                                              /* 500-503 */
      orc_gemm_nn(n, len_u, n_act, 1.0, bspace, n, u_p, len_u, 0.0, evec, n);
      memcpy(bspace + nn * ind_p, evec, sizeof(double) * nn * n_act);
    }
    free(ptmp); free(u_x); free(u_p);
    memcpy(space, x_new, sizeof(double) * nn * n_max);      /* 510 */
    memcpy(aspace, ax_new, sizeof(double) * nn * n_max);    /* 511 */
    if (gen) memcpy(bspace, bx_new, sizeof(double) * nn * n_max);   /* 513 */
    {
      double fac = shift - eig[0];                          /* 518 */
      precnd(&n, &n_act, &fac, r + nn * ind_x, space + nn * ind_w);
    }
    if (gen) {                                              /* 523-526 */
      orc_b_ortho_vs_x(n, n_max + n_act, n_act, space, bspace, space + nn * ind_w);
      bvec(&n, &n_act, space + nn * ind_w, bspace + nn * ind_w);
      orc_b_ortho(n, n_act, space + nn * ind_w, bspace + nn * ind_w);
    } else {
      orc_ortho_vs_x(n, n_max + n_act, n_act, space, space + nn * ind_w, NULL); /* 528 */
    }
  }
  free(space); free(aspace); free(bspace); free(bx_new); free(r); free(x_new); free(ax_new); free(a_red); free(e_red);
  free(rn); free(done);
}

void orc_lobpcg(int verbose, int n, int n_targ, int n_max, int max_iter, double tol,
                double shift, orc_matvec_t matvec, orc_precnd_t precnd,
                double* eig, double* evec, int* ok, orc_trace* tr)
{
  lobpcg_impl(verbose, 0, n, n_targ, n_max, max_iter, tol, shift, matvec, precnd, matvec, eig, evec, ok, tr);
}

void orc_lobpcg_gen(int verbose, int n, int n_targ, int n_max, int max_iter, double tol,
                    double shift, orc_matvec_t matvec, orc_precnd_t precnd, orc_matvec_t bvec,
                    double* eig, double* evec, int* ok, orc_trace* tr)
{
  lobpcg_impl(verbose, 1, n, n_targ, n_max, max_iter, tol, shift, matvec, precnd, bvec, eig, evec, ok, tr);
}

/* diaglib.f90:1855-2250: Davidson-Liu with a metric B (bvec callback).  Same loop as orc_davidson plus the
 * bspace = B*space panel: the guess is B-orthonormalised (2033-2034), residuals are A x - theta B x (2113-2123),
 * new blocks go through b_ortho_vs_x + bvec + b_ortho (2183-2185).
 * DELIBERATE DEVIATION at the restart: the reference zeroes ALL of bspace right after it has
 * B-orthonormalised the kept Ritz block (2196-2200) and never refills it (the bvec call at 2072 is commented
 * out), so every residual after a restart misses theta*B*x of the kept block.  Here the kept block's B*x
 * (already available, b_ortho updates it) stays in bspace(:,1:n_max); everything else follows the reference.
 * Runs without a restart are identical to the reference. */
void orc_gen_davidson(int verbose, int n, int n_targ, int n_max, int max_iter, double tol, int max_dav,
                      double shift, orc_matvec_t matvec, orc_precnd_t precnd, orc_matvec_t bvec,
                      double* eig, double* evec, int* ok, orc_trace* tr)
{
  const int min_dav = 10;
  int dim_dav = max_dav > min_dav ? max_dav : min_dav;
  int lda = dim_dav * n_max;
  size_t nn = (size_t)n;
  double* space  = (double*)calloc(nn * lda, sizeof(double));
  double* aspace = (double*)calloc(nn * lda, sizeof(double));
  double* bspace = (double*)calloc(nn * lda, sizeof(double));
  double* r      = (double*)calloc(nn * n_max, sizeof(double));
  double* b_evec = (double*)calloc(nn * n_max, sizeof(double));
  double* a_red  = (double*)calloc((size_t)lda * lda, sizeof(double));
  double* a_copy = (double*)calloc((size_t)lda * lda, sizeof(double));
  double* e_red  = (double*)calloc((size_t)lda, sizeof(double));
  double* rn     = (double*)calloc((size_t)2 * n_max, sizeof(double));
  int*    done   = (int*)calloc((size_t)n_max, sizeof(int));
  double sqrtn = sqrt((double)n), tol_rms = tol, tol_max = 10.0 * tol;
  *ok = 0;
  if (tr) { tr->iters = 0; tr->matvec_cols = 0; tr->restarts = 0; }

  orc_check_guess(n, n_max, evec);                          /* 2025 */
  memcpy(space, evec, sizeof(double) * nn * n_max);         /* 2029 */
  bvec(&n, &n_max, space, bspace);                          /* 2033 */
  orc_b_ortho(n, n_max, space, bspace);                     /* 2034 */

  int n_act = n_max, i_beg = 0, m_dim = 1, ldu = 0, restart = 0, n_rst = 0;
  if (verbose) printf("    Generalized Davidson-Liu iterations (tol=%10.2E):\n", tol);
  for (int it = 1; it <= max_iter; ++it) {
    ldu += n_act;
    int c0 = i_beg + n_rst;
    matvec(&n, &n_act, space + nn * c0, aspace + nn * c0);  /* 2071 */
    if (tr) tr->matvec_cols += n_act;
    orc_gemm_tn(n, ldu, n_act, space, n, aspace + nn * c0, n, a_red + (size_t)lda * c0, lda);  /* 2079 */
    if (restart) {
      for (int i = 0; i < n_rst; ++i) A_(a_red, lda, i, i) = e_red[i];
      restart = 0; n_rst = 0;
    }
    memcpy(a_copy, a_red, sizeof(double) * (size_t)lda * lda);
    orc_syev('u', ldu, a_copy, lda, e_red);                 /* 2099 */
    for (int i = 0; i < n_max; ++i) eig[i] = e_red[i];
    orc_gemm_nn(n, ldu, n_max, 1.0, space, n, a_copy, lda, 0.0, evec, n);    /* 2108 */
    orc_gemm_nn(n, ldu, n_max, 1.0, aspace, n, a_copy, lda, 0.0, r, n);      /* 2112 */
    orc_gemm_nn(n, ldu, n_max, 1.0, bspace, n, a_copy, lda, 0.0, b_evec, n); /* 2113 */
    for (int i = 0; i < n_targ; ++i) {                      /* 2115-2124 */
      if (done[i]) continue;
      double* ri = r + nn * i; const double* bi = b_evec + nn * i;
      double s = 0.0, mx = 0.0;
      for (size_t p = 0; p < nn; ++p) {
        ri[p] -= eig[i] * bi[p];
        s += ri[p] * ri[p];
        double a = fabs(ri[p]); if (a > mx) mx = a;
      }
      rn[2 * i] = sqrt(s) / sqrtn; rn[2 * i + 1] = mx;
    }
    for (int i = 0; i < n_targ; ++i) {
      if (done[i]) continue;
      done[i] = (rn[2 * i] < tol_rms) && (rn[2 * i + 1] < tol_max) && (it > 1);
      if (!done[i]) { for (int j = i + 1; j < n_max; ++j) done[j] = 0; break; }
    }
    trace_put(tr, it - 1, n_targ, n_act, ldu, eig, shift, rn, done);
    if (verbose) print_iter(it, n_targ, eig, shift, rn, done);
    int all = 1;
    for (int i = 0; i < n_targ; ++i) all = all && done[i];
    if (all) { *ok = 1; break; }
    if (m_dim < dim_dav) {
      m_dim++; i_beg += n_act; n_act = n_max;
      for (int i = 0; i < n_targ; ++i) { if (done[i]) n_act--; else break; }
      int ind0 = n_max - n_act;
      double fac = -eig[ind0];
      precnd(&n, &n_act, &fac, r + nn * ind0, space + nn * i_beg);                    /* 2177 */
      if (orc_b_ortho_vs_x(n, ldu, n_act, space, bspace, space + nn * i_beg)) break;  /* 2183 */
      bvec(&n, &n_act, space + nn * i_beg, bspace + nn * i_beg);                      /* 2184 */
      orc_b_ortho(n, n_act, space + nn * i_beg, bspace + nn * i_beg);                 /* 2185 */
    } else {
      if (verbose) printf("      Restarting davidson.\n");
      if (tr) tr->restarts++;
      n_act = n_max;
      memset(space, 0, sizeof(double) * nn * lda);
      memcpy(space, evec, sizeof(double) * nn * n_max);     /* 2195 */
      memset(bspace, 0, sizeof(double) * nn * lda);
      memcpy(bspace, b_evec, sizeof(double) * nn * n_max);  /* 2196 */
      orc_b_ortho(n, n_max, space, bspace);                 /* 2197; bspace(:,1:n_max) is KEPT, see header */
      memset(aspace, 0, sizeof(double) * nn * lda);
      memset(a_red, 0, sizeof(double) * (size_t)lda * lda);
      ldu = 0; i_beg = 0; m_dim = 1; n_rst = 0;
      for (int i = 0; i < n_targ; ++i) { if (done[i]) n_rst++; else break; }
      restart = 1;
    }
  }
  free(space); free(aspace); free(bspace); free(r); free(b_evec); free(a_red); free(a_copy); free(e_red); free(rn); free(done);
}
