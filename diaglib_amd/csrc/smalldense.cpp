// diaglib_amd/csrc/smalldense.cpp -- host-size dense kernels of the hot path.
//
// The reference obtains these from LAPACK (un-vendored external): dsyev at
// diaglib.f90:315,406,1708; dpotrf at :3173,3261,3290; dtrtri at :3310; plus its own
// norm_est (:3447-3479).  They operate on matrices of order <= lda (a few hundred), i.e.
// O(1) in n, and stay on the host.  Written from the textbook
// algorithms so that the library needs no LAPACK at run time.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../include/diaglib_amd.h"

namespace {

inline double& at(double* a, int ld, int i, int j) { return a[(size_t)i + (size_t)j * ld]; }
inline double at(const double* a, int ld, int i, int j) { return a[(size_t)i + (size_t)j * ld]; }

// Symmetric eigensolver: Householder reduction to tridiagonal form, then implicit-shift QL
// on the tridiagonal with the rotations accumulated into the transformation matrix.
// s: full symmetric n x n (row-major == column-major), destroyed.  On exit zt holds the
// eigenvectors as ROWS (zt[k*n + i] = component i of eigenvector k) so that every plane
// rotation touches two contiguous rows.
int sym_eig(int n, std::vector<double>& s, std::vector<double>& d, std::vector<double>& zt)
{
  std::vector<double> e(n, 0.0), v(n), p(n);
  std::vector<double> hv((size_t)n * n, 0.0);  // Householder vectors, row k = vector of step k
  std::vector<double> hbeta(n, 0.0);
  auto S = [&](int i, int j) -> double& { return s[(size_t)i * n + j]; };

  // --- reduction: for k = 0..n-3 annihilate S[k+2.., k]
  for (int k = 0; k + 2 < n; ++k) {
    int m = n - k - 1;  // length of the column below the diagonal
    double scale = 0.0;
    for (int i = 0; i < m; ++i) scale = std::max(scale, std::fabs(S(k + 1 + i, k)));
    if (scale == 0.0) { hbeta[k] = 0.0; continue; }
    double nrm = 0.0;
    for (int i = 0; i < m; ++i) { v[i] = S(k + 1 + i, k) / scale; nrm += v[i] * v[i]; }
    nrm = std::sqrt(nrm);
    double alpha = (v[0] >= 0.0) ? -nrm : nrm;  // new sub-diagonal entry / scale
    double rest = 0.0;
    for (int i = 1; i < m; ++i) rest += v[i] * v[i];
    if (rest == 0.0) { hbeta[k] = 0.0; continue; }  // already tridiagonal in this column
    v[0] -= alpha;
    double vtv = v[0] * v[0] + rest;
    double beta = 2.0 / vtv;
    // p = beta * S22 v
    for (int i = 0; i < m; ++i) {
      const double* row = &s[(size_t)(k + 1 + i) * n + (k + 1)];
      double acc = 0.0;
      for (int j = 0; j < m; ++j) acc += row[j] * v[j];
      p[i] = beta * acc;
    }
    double pv = 0.0;
    for (int i = 0; i < m; ++i) pv += p[i] * v[i];
    double kk = 0.5 * beta * pv;
    for (int i = 0; i < m; ++i) p[i] -= kk * v[i];  // w
    for (int i = 0; i < m; ++i) {
      double* row = &s[(size_t)(k + 1 + i) * n + (k + 1)];
      double vi = v[i], wi = p[i];
      for (int j = 0; j < m; ++j) row[j] -= vi * p[j] + wi * v[j];
    }
    S(k + 1, k) = alpha * scale;
    S(k, k + 1) = alpha * scale;
    for (int i = 1; i < m; ++i) { S(k + 1 + i, k) = 0.0; S(k, k + 1 + i) = 0.0; }
    hbeta[k] = beta;
    std::memcpy(&hv[(size_t)k * n + (k + 1)], v.data(), sizeof(double) * m);
  }
  for (int i = 0; i < n; ++i) d[i] = S(i, i);
  for (int i = 0; i + 1 < n; ++i) e[i] = S(i + 1, i);

  // --- accumulate Q = H_0 H_1 ... H_{n-3}; we need Q^T rows: zt = Q^T, built by applying the
  // reflectors to the identity from the last to the first: Q = H_0 (H_1 (... I)).
  std::fill(zt.begin(), zt.end(), 0.0);
  for (int i = 0; i < n; ++i) zt[(size_t)i * n + i] = 1.0;
  // zt rows are columns of Q: zt[c*n + r] = Q[r][c].  Applying H_k on the left of Q acts on
  // the r index: for each column c, q_c -= beta v (v^T q_c).
  for (int k = n - 3; k >= 0; --k) {
    double beta = hbeta[k];
    if (beta == 0.0) continue;
    const double* vk = &hv[(size_t)k * n];
    for (int c = k + 1; c < n; ++c) {  // columns <= k of the trailing product are still unit vectors e_c with c<=k: untouched
      double* q = &zt[(size_t)c * n];
      double acc = 0.0;
      for (int r = k + 1; r < n; ++r) acc += vk[r] * q[r];
      acc *= beta;
      for (int r = k + 1; r < n; ++r) q[r] -= acc * vk[r];
    }
  }
  // now zt[c*n + r] = Q[r][c]; eigenvectors of S are Q * (eigenvectors of T).  The QL sweep
  // below post-multiplies Q by rotations acting on column pairs (c, c+1) => row pairs of zt.

  // --- implicit QL with Wilkinson shift
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; ++l) {
    int iter = 0;
    while (true) {
      int m = l;
      for (; m + 1 < n; ++m) {
        double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= eps * dd) break;
      }
      if (m == l) break;
      if (++iter > 60) return l + 1;
      double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
      double r = std::hypot(g, 1.0);
      g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? r : -r));
      double sn = 1.0, cs = 1.0, pp = 0.0;
      int i = m - 1;
      for (; i >= l; --i) {
        double f = sn * e[i], b = cs * e[i];
        r = std::hypot(f, g);
        e[i + 1] = r;
        if (r == 0.0) { d[i + 1] -= pp; e[m] = 0.0; break; }
        sn = f / r; cs = g / r;
        g = d[i + 1] - pp;
        r = (d[i] - g) * sn + 2.0 * cs * b;
        pp = sn * r;
        d[i + 1] = g + pp;
        g = cs * r - b;
        double* zi = &zt[(size_t)i * n];
        double* zj = &zt[(size_t)(i + 1) * n];
        for (int k = 0; k < n; ++k) {
          double fj = zj[k];
          zj[k] = sn * zi[k] + cs * fj;
          zi[k] = cs * zi[k] - sn * fj;
        }
      }
      if (r == 0.0 && i >= l) continue;
      d[l] -= pp; e[l] = g; e[m] = 0.0;
    }
  }
  return 0;
}

}  // namespace

extern "C" {

int dla_syev(char uplo, int n, double* a, int lda, double* w)
{
  if (n <= 0) return 0;
  if (lda < n) return -1;
  bool up = (uplo == 'u' || uplo == 'U');
  std::vector<double> s((size_t)n * n), d(n), zt((size_t)n * n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      double x = up ? at(a, lda, i, j) : at(a, lda, j, i);
      s[(size_t)i * n + j] = x;
      s[(size_t)j * n + i] = x;
    }
  int info = sym_eig(n, s, d, zt);
  if (info != 0) return info;
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int p, int q) { return d[p] < d[q]; });
  for (int j = 0; j < n; ++j) {
    w[j] = d[idx[j]];
    const double* z = &zt[(size_t)idx[j] * n];
    // fix the sign so that the result does not depend on reflector conventions:
    // largest-magnitude component positive
    int imax = 0;
    for (int i = 1; i < n; ++i) if (std::fabs(z[i]) > std::fabs(z[imax])) imax = i;
    double sg = (z[imax] < 0.0) ? -1.0 : 1.0;
    for (int i = 0; i < n; ++i) at(a, lda, i, j) = sg * z[i];
  }
  return 0;
}

int dla_potrf_lower(int m, double* a, int lda)
{
  for (int j = 0; j < m; ++j) {
    double dj = at(a, lda, j, j);
    for (int p = 0; p < j; ++p) dj -= at(a, lda, j, p) * at(a, lda, j, p);
    if (!(dj > 0.0) || !std::isfinite(dj)) return j + 1;
    dj = std::sqrt(dj);
    at(a, lda, j, j) = dj;
    double inv = 1.0 / dj;
    for (int i = j + 1; i < m; ++i) {
      double sij = at(a, lda, i, j);
      for (int p = 0; p < j; ++p) sij -= at(a, lda, i, p) * at(a, lda, j, p);
      at(a, lda, i, j) = sij * inv;
    }
  }
  return 0;
}

int dla_trtri_lower(int m, double* a, int lda)
{
  for (int j = 0; j < m; ++j)
    if (at(a, lda, j, j) == 0.0) return j + 1;
  std::vector<double> x(m);
  // invert column by column, right to left, so that the part of L still needed stays intact:
  // X(:,j) depends on L(:, j..m-1) only; write X(:,j) over L(:,j) after it is complete.
  for (int j = m - 1; j >= 0; --j) {
    x[j] = 1.0 / at(a, lda, j, j);
    for (int i = j + 1; i < m; ++i) {
      // row i of L X = I :  sum_{p=j..i} L(i,p) X(p,j) = 0.  Columns p>j already hold X, not L,
      // so use the recurrence on X instead: X = L^-1 satisfies X(i,j) = -X(i,i..) ... solve by rows of X L = I:
      // (X L)(i,j) = sum_{p=j..i} X(i,p) L(p,j) = 0  ->  X(i,j) = -(sum_{p=j+1..i} X(i,p) L(p,j)) / L(j,j)
      double sacc = 0.0;
      for (int p = j + 1; p <= i; ++p) sacc += at(a, lda, i, p) * at(a, lda, p, j);
      x[i] = -sacc * x[j];
    }
    for (int i = j; i < m; ++i) at(a, lda, i, j) = x[i];
  }
  return 0;
}

double dla_norm_est(int m, const double* a, int lda)
{
  double dn = 0.0, on = 0.0;
  for (int i = 0; i < m; ++i) dn = std::max(dn, std::fabs(at(a, lda, i, i)));
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < i; ++j) on += at(a, lda, i, j) * at(a, lda, i, j);
  return dn + std::sqrt(on);
}

}  // extern "C"
