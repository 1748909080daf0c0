// diaglib_amd/csrc/smalldense.cpp -- host-size dense kernels of the hot path.
//
// The reference obtains these from LAPACK (un-vendored external): dsyev at
// diaglib.f90:315,406,1708; dpotrf at :3173,3261,3290; dtrtri at :3310; plus its own
// norm_est (:3447-3479).  They operate on matrices of order <= lda (a few hundred), i.e.
// O(1) in n, and stay on the host.  Written from the textbook
// algorithms so that the library needs no LAPACK at run time.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/diaglib_amd.h"
#include "dla_internal.h"

// The eigensolvers below are compiled twice -- for AVX2 (every x86 host of an MI355X node has it) and for AVX-512 (EPYC Zen 4 / 5:
// twice the vector width in the reduction and the back-transformation, which run on the critical path of every iteration while
// the GPU waits) -- into two namespaces; the C entry points at the end of this file pick one at run time.
#ifndef SD_NS
#define SD_NS sd_v3
#endif

namespace SD_NS {

// sqrt(a^2+b^2); the matrices handled here are projected operators (|entries| << 1e150), so the
// plain form is safe and several times faster than std::hypot
inline double pythag(double a, double b) { return std::sqrt(a * a + b * b); }

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
      double r = pythag(g, 1.0);
      g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? r : -r));
      double sn = 1.0, cs = 1.0, pp = 0.0;
      int i = m - 1;
      for (; i >= l; --i) {
        double f = sn * e[i], b = cs * e[i];
        r = pythag(f, g);
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


// ---------------------------------------------------------------------------------------
// Partial solver: the drivers only ever use the lowest n_max eigenpairs of the projected
// matrix (reference diaglib.f90:1715-1721 uses a_copy(:,1:n_max), get_coeffs :3712 uses
// a_red(:,1:n_max)).  Householder tridiagonalisation, the m lowest eigenvalues by bisection on
// Sturm counts, their eigenvectors by inverse iteration on the tridiagonal matrix with
// re-orthogonalisation inside clusters, back-transformation with the stored reflectors.
// Cost 4/3 n^3 + O(m n^2) instead of ~9 n^3.  This runs on the critical path of every iteration
// while the GPU waits, so the loops are laid out for the host's SIMD units.
// ---------------------------------------------------------------------------------------
#if defined(__AVX512F__)
constexpr int VW = 8;
typedef double v4 __attribute__((vector_size(64)));     // ("v4": the SIMD vector of this build, 4 or 8 doubles)
inline v4 bc4(double x) { return (v4){x, x, x, x, x, x, x, x}; }
inline double hsum4(v4 a) { return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])); }
#else
constexpr int VW = 4;
typedef double v4 __attribute__((vector_size(32)));
inline v4 bc4(double x) { return (v4){x, x, x, x}; }
inline double hsum4(v4 a) { return (a[0] + a[1]) + (a[2] + a[3]); }
#endif
inline v4 ld4(const double* q) { v4 r; std::memcpy(&r, q, sizeof r); return r; }
inline void st4(double* q, v4 r) { std::memcpy(q, &r, sizeof r); }

struct Tridiag {
  int n;
  std::vector<double> d, e;       // diagonal, sub-diagonal (e[i] couples i and i+1)
  std::vector<double> hv, hbeta;  // reflectors: row k of hv (entries k+1..n-1), beta_k
};

// One pass over the stored part of a row of the trailing matrix (upper triangle, row-major: row[0] is
// the diagonal entry, row[j] couples this index with index + j).  All vectors are offset to this index.
//   UPD: row[j] -= vi * w[j] + wi * v[j]                        (symmetric rank-2 update of step k)
//   MV : returns sum_j row[j] * x[j]; pn[j] += row[j] * x[0], j >= 1   (both halves of A x for step k+1)
template <bool UPD, bool MV>
inline double row_pass(double* row, int len, double vi, double wi, const double* v, const double* w,
                       const double* x, double* pn)
{
  double r0 = row[0];
  if (UPD) { r0 -= vi * w[0] + wi * v[0]; row[0] = r0; }
  double dot = 0.0;
  const double x0 = MV ? x[0] : 0.0;
  if (MV) dot = r0 * x0;
  int j = 1;
  v4 acc = bc4(0.0);
  const v4 vvi = bc4(vi), vwi = bc4(wi), vx0 = bc4(x0);
  for (; j + VW <= len; j += VW) {
    v4 r = ld4(row + j);
    if (UPD) { r -= vvi * ld4(w + j) + vwi * ld4(v + j); st4(row + j, r); }
    if (MV) { acc += r * ld4(x + j); st4(pn + j, ld4(pn + j) + r * vx0); }
  }
  double tail = 0.0;
  for (; j < len; ++j) {
    double r = row[j];
    if (UPD) { r -= vi * w[j] + wi * v[j]; row[j] = r; }
    if (MV) { tail += r * x[j]; pn[j] += r * x0; }
  }
  return dot + (hsum4(acc) + tail);
}

// Two consecutive rows (a: index i, b: index i+1) in one pass: the elements of w, v, x and pn that both rows touch are
// loaded once.  Vectors are offset to index i.  Returns the two row dot products through da / db; same arithmetic per
// element as two row_pass<true, true> calls (row a's contribution to pn is added before row b's).
inline void row_pass2(double* a, double* b, int la, double va, double wa, double vb, double wb, const double* v, const double* w,
                      const double* x, double* pn, double* da, double* db)
{
  // la >= 2: a[0] diagonal of i, a[1] couples (i, i+1); b[0] diagonal of i+1
  double a0 = a[0] - (va * w[0] + wa * v[0]); a[0] = a0;
  double a1 = a[1] - (va * w[1] + wa * v[1]); a[1] = a1;
  double b0 = b[0] - (vb * w[1] + wb * v[1]); b[0] = b0;
  const double x0 = x[0], x1 = x[1];
  double dota = a0 * x0, dotb = b0 * x1;
  double ta = a1 * x1;
  pn[1] += a1 * x0;
  int j = 2;
  v4 acca = bc4(0.0), accb = bc4(0.0);
  const v4 vva = bc4(va), vwa = bc4(wa), vvb = bc4(vb), vwb = bc4(wb), vx0 = bc4(x0), vx1 = bc4(x1);
  for (; j + VW <= la; j += VW) {
    const v4 wj = ld4(w + j), vj = ld4(v + j), xj = ld4(x + j);
    v4 ra = ld4(a + j), rb = ld4(b + j - 1);
    ra -= vva * wj + vwa * vj; st4(a + j, ra);
    rb -= vvb * wj + vwb * vj; st4(b + j - 1, rb);
    acca += ra * xj; accb += rb * xj;
    v4 p = ld4(pn + j);
    p = p + ra * vx0;
    p = p + rb * vx1;
    st4(pn + j, p);
  }
  double taila = 0.0, tailb = 0.0;
  for (; j < la; ++j) {
    double ra = a[j], rb = b[j - 1];
    ra -= va * w[j] + wa * v[j]; a[j] = ra;
    rb -= vb * w[j] + wb * v[j]; b[j - 1] = rb;
    taila += ra * x[j]; tailb += rb * x[j];
    double p = pn[j];
    p = p + ra * x0;
    p = p + rb * x1;
    pn[j] = p;
  }
  *da = dota + ta + (hsum4(acca) + taila);
  *db = dotb + (hsum4(accb) + tailb);
}

// Householder vector of x (length m): on exit v holds the vector, returns beta (0: nothing to annihilate)
// and *sub = the new sub-diagonal entry.
inline double make_reflector(const double* x, int m, double* v, double* sub)
{
  double scale = 0.0;
  for (int i = 0; i < m; ++i) scale = std::max(scale, std::fabs(x[i]));
  *sub = x[0];
  if (scale == 0.0) return 0.0;
  double rest = 0.0;
  for (int i = 0; i < m; ++i) v[i] = x[i] / scale;
  for (int i = 1; i < m; ++i) rest += v[i] * v[i];
  if (rest == 0.0) return 0.0;
  const double nrm = std::sqrt(v[0] * v[0] + rest);
  const double alpha = (v[0] >= 0.0) ? -nrm : nrm;
  v[0] -= alpha;
  *sub = alpha * scale;
  return 2.0 / (v[0] * v[0] + rest);
}

// s: symmetric n x n, row-major; only the upper triangle (j >= i) is read and updated.  The rank-2 update
// of step k and the matrix-vector product of step k+1 share one pass over the trailing matrix.
void tridiagonalize(int n, std::vector<double>& s, Tridiag& t)
{
  t.n = n;
  t.d.assign(n, 0.0); t.e.assign(n, 0.0);
  if (t.hv.size() < (size_t)n * n) t.hv.resize((size_t)n * n);     // (row k is written where it is read: entries k+1 .. n-1, when beta_k != 0)
  t.hbeta.assign(n, 0.0);
  std::vector<double> vbuf[2], pbuf[2], w(n + VW, 0.0);
  for (int q = 0; q < 2; ++q) { vbuf[q].assign(n + VW, 0.0); pbuf[q].assign(n + VW, 0.0); }
  if (n >= 3) {
    // step 0: reflector from row 0, plain product with the trailing matrix
    double sub;
    double beta = make_reflector(&s[1], n - 1, vbuf[0].data(), &sub);
    double* p0 = pbuf[0].data();
    if (beta != 0.0)
      for (int i = 0; i < n - 1; ++i) {
        double* row = &s[(size_t)(1 + i) * n + (1 + i)];
        p0[i] += row_pass<false, true>(row, n - 1 - i, 0.0, 0.0, nullptr, nullptr, vbuf[0].data() + i, p0 + i);
      }
    for (int k = 0; k + 2 < n; ++k) {
      const int m = n - k - 1;
      double* v = vbuf[k & 1].data();
      double* p = pbuf[k & 1].data();
      double* vn = vbuf[(k + 1) & 1].data();
      double* pn = pbuf[(k + 1) & 1].data();
      t.hbeta[k] = beta;
      t.e[k] = sub;
      const bool upd = beta != 0.0;
      if (upd) {
        std::memcpy(&t.hv[(size_t)k * n + (k + 1)], v, sizeof(double) * m);
        double pv = 0.0;
        for (int i = 0; i < m; ++i) { p[i] *= beta; pv += p[i] * v[i]; }
        const double kk = 0.5 * beta * pv;
        for (int i = 0; i < m; ++i) w[i] = p[i] - kk * v[i];
      }
      // first row of the trailing matrix: update, then it defines the next reflector
      double* row0 = &s[(size_t)(k + 1) * n + (k + 1)];
      if (upd) row_pass<true, false>(row0, m, v[0], w[0], v, w.data(), nullptr, nullptr);
      double betan = 0.0, subn = (m >= 2) ? row0[1] : 0.0;
      const bool next = (k + 3 < n);
      if (next) {
        betan = make_reflector(row0 + 1, m - 1, vn, &subn);
        for (int i = 0; i < m - 1; ++i) pn[i] = 0.0;
      }
      const bool mv = next && betan != 0.0;
      int i = 1;
      if (upd && mv) {
        // rows in pairs: the elements of w, v, vn and pn that two consecutive rows share are loaded once (-35 % on the
        // whole reduction at n = 117)
        for (; i + 1 < m; i += 2) {
          double* ra = &s[(size_t)(k + 1 + i) * n + (k + 1 + i)];
          double* rb = &s[(size_t)(k + 2 + i) * n + (k + 2 + i)];
          double da, db;
          row_pass2(ra, rb, m - i, v[i], w[i], v[i + 1], w[i + 1], v + i, w.data() + i, vn + i - 1, pn + i - 1, &da, &db);
          pn[i - 1] += da;
          pn[i] += db;
        }
      }
      for (; i < m; ++i) {
        double* row = &s[(size_t)(k + 1 + i) * n + (k + 1 + i)];
        const int len = m - i;
        if (upd && mv)       pn[i - 1] += row_pass<true, true>(row, len, v[i], w[i], v + i, w.data() + i, vn + i - 1, pn + i - 1);
        else if (upd)        row_pass<true, false>(row, len, v[i], w[i], v + i, w.data() + i, nullptr, nullptr);
        else if (mv)         pn[i - 1] += row_pass<false, true>(row, len, 0.0, 0.0, nullptr, nullptr, vn + i - 1, pn + i - 1);
      }
      beta = betan; sub = subn;
      if (!next) { t.e[k + 1] = subn; }
    }
  } else if (n == 2) {
    t.e[0] = s[1];
  }
  for (int i = 0; i < n; ++i) t.d[i] = s[(size_t)i * n + i];
}

// Inverse iteration for up to 16 eigenvectors AT ONCE.  (T - lam_j I) = P L U by Gaussian elimination with partial
// pivoting on the tridiagonal matrix (tiny pivots are replaced by +-tiny: inverse iteration only needs the direction), one
// shift per SIMD lane: the factorisation and the two triangular solves are first-order recurrences along the matrix index,
// i.e. pure latency for one vector -- sixteen of them advance together at the price of one.  One factorisation serves
// all the iterations of its lane.
constexpr int LW = 16;                                 // lanes = shifts per batch
typedef double v8 __attribute__((vector_size(64)));
typedef long long v8i __attribute__((vector_size(64)));
struct Lanes { v8 a, b; };                             // 16 doubles
inline v8 bc8(double x) { return (v8){x, x, x, x, x, x, x, x}; }
inline v8 abs8(v8 x) { return x < bc8(0.0) ? -x : x; }
inline v8 sel8(v8i m, v8 x, v8 y) { return m ? x : y; }

struct TriLU16 {
  // U: reciprocal diagonal, first and second super-diagonal; multipliers; pivot masks -- [index][lane]
  std::vector<v8> a, b, c, l;
  std::vector<v8i> piv;
  void reserve(int n) { if ((int)a.size() < 2 * n) { a.resize(2 * n); b.resize(2 * n); c.resize(2 * n); l.resize(2 * n); piv.resize(2 * n); } }
  void factor(int n, const std::vector<double>& d, const std::vector<double>& e, const double* lam, double tiny)
  {
    reserve(n);
    for (int h = 0; h < 2; ++h) {
      v8 lm; std::memcpy(&lm, lam + 8 * h, sizeof lm);
      v8 ai = bc8(d[0]) - lm, bi = bc8(n > 1 ? e[0] : 0.0);
      const v8 vt = bc8(tiny);
      for (int i = 0; i + 1 < n; ++i) {
        const v8 sub = bc8(e[i]);                          // T(i+1, i), the same in every lane
        const v8 an = bc8(d[i + 1]) - lm, bn = bc8(i + 2 < n ? e[i + 1] : 0.0);
        const v8i swp = abs8(ai) < abs8(sub);              // swap rows i and i+1
        const v8 a0 = sel8(ai == bc8(0.0), vt, ai);        // (only used where no swap happens)
        const v8 piv_i = sel8(swp, sub, a0);               // the pivot of this step
        const v8 mlt = sel8(swp, ai, sub) / piv_i;         // (one division per step on the dependent chain)
        // no swap: U_i = (a0, bi, 0), next row (an - mlt bi, bn);  swap: U_i = (sub, an, bn), next row (bi - mlt an, -mlt bn)
        a[2 * i + h] = piv_i;
        b[2 * i + h] = sel8(swp, an, bi);
        c[2 * i + h] = sel8(swp, bn, bc8(0.0));
        l[2 * i + h] = mlt;
        piv[2 * i + h] = swp;
        const v8 na = sel8(swp, bi - mlt * an, an - mlt * bi);
        const v8 nb = sel8(swp, -mlt * bn, bn);
        ai = na; bi = nb;
      }
      a[2 * (n - 1) + h] = ai; b[2 * (n - 1) + h] = bc8(0.0); c[2 * (n - 1) + h] = bc8(0.0);
      // tiny pivots are replaced by +-tiny (inverse iteration only needs the direction); the back substitution multiplies by
      // the reciprocals (off the dependent chain here, on it there)
      for (int i = 0; i < n; ++i) {
        const v8 x = a[2 * i + h];
        a[2 * i + h] = bc8(1.0) / sel8(abs8(x) < vt, sel8(x < bc8(0.0), -vt, vt), x);
      }
    }
  }
  // x: [index][lane], 2 vectors per index
  void solve(int n, v8* x) const
  {
    for (int h = 0; h < 2; ++h) {
      v8 xi = x[h];
      for (int i = 0; i + 1 < n; ++i) {
        const v8 xn = x[2 * (i + 1) + h];
        const v8i swp = piv[2 * i + h];
        const v8 top = sel8(swp, xn, xi), bot = sel8(swp, xi, xn);
        x[2 * i + h] = top;
        xi = bot - l[2 * i + h] * top;
      }
      x[2 * (n - 1) + h] = xi;
      v8 x1 = bc8(0.0), x2 = bc8(0.0);
      for (int i = n - 1; i >= 0; --i) {
        const v8 t = (x[2 * i + h] - b[2 * i + h] * x1 - c[2 * i + h] * x2) * a[2 * i + h];
        x[2 * i + h] = t;
        x2 = x1; x1 = t;
      }
    }
  }
};

// The m lowest eigenvalues of the symmetric tridiagonal (d, e) by simultaneous bisection on Sturm
// counts: count(x) = number of eigenvalues < x = number of sign changes in the sequence of leading
// principal minors p_0 = 1, p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}.  The matrix is scaled to unit
// norm and (p_i, p_{i-1}) are rescaled by a power of two every fourth step, so the division-free
// recurrence neither overflows nor underflows; 16 shifts advance together (4 AVX2 vectors).
void bisect_lowest(int n, const std::vector<double>& d, const std::vector<double>& e, int m, double onenrm,
                   std::vector<double>& w)
{
  const double eps = 2.220446049250313e-16;
  const double inv = 1.0 / onenrm;
  std::vector<double> ds(n), e2(n > 0 ? n : 1, 0.0);
  for (int i = 0; i < n; ++i) ds[i] = d[i] * inv;
  for (int i = 0; i + 1 < n; ++i) { const double es = e[i] * inv; e2[i] = es * es; }
  // Gershgorin bounds (scaled)
  double gl = ds[0], gu = ds[0];
  for (int i = 0; i < n; ++i) {
    const double r = (i > 0 ? std::fabs(e[i - 1]) : 0.0) * inv + (i + 1 < n ? std::fabs(e[i]) : 0.0) * inv;
    gl = std::min(gl, ds[i] - r);
    gu = std::max(gu, ds[i] + r);
  }
  const double pad = 4.0 * eps * n;
  gl -= pad; gu += pad;
  constexpr int W = 16, NV = W / VW;
  const int mp = ((m + W - 1) / W) * W;
  std::vector<double> lo(mp, gl), hi(mp, gu), mid(mp, 0.0);
  std::vector<int> cnt(mp, 0);
  const double BIG = 0x1p+300, SMALL = 0x1p-300, ZERO_REPL = 0x1p-900;
  typedef long long v4i __attribute__((vector_size(8 * VW)));
  auto sturm = [&](const double* x, int* c) {   // counts for W shifts at once
    v4 xs[NV], p0[NV], p1[NV];
    v4i neg[NV];
    for (int q = 0; q < NV; ++q) {
      xs[q] = ld4(x + VW * q);
      p0[q] = bc4(1.0);
      p1[q] = bc4(ds[0]) - xs[q];
      neg[q] = v4i{};
    }
    auto account = [&](int q) {
      // an exact zero takes the sign opposite to its predecessor (it counts as a negative pivot)
      v4i isz = (p1[q] == bc4(0.0));
      v4 repl = (p0[q] < bc4(0.0)) ? bc4(ZERO_REPL) : bc4(-ZERO_REPL);
      p1[q] = isz ? repl : p1[q];
      v4i changed = ((p1[q] < bc4(0.0)) != (p0[q] < bc4(0.0)));   // all-ones (-1) where the sign changed
      neg[q] -= changed;
    };
    for (int q = 0; q < NV; ++q) account(q);
    for (int i = 1; i < n; ++i) {
      const v4 di = bc4(ds[i]), ei2 = bc4(e2[i - 1]);
      for (int q = 0; q < NV; ++q) {
        const v4 t = (di - xs[q]) * p1[q] - ei2 * p0[q];
        p0[q] = p1[q];
        p1[q] = t;
        account(q);
      }
      if ((i & 3) == 3) {
        for (int q = 0; q < NV; ++q) {
          v4 a1 = p1[q] < bc4(0.0) ? -p1[q] : p1[q];
          v4 a0 = p0[q] < bc4(0.0) ? -p0[q] : p0[q];
          v4 am = a1 > a0 ? a1 : a0;
          v4 sc = am > bc4(BIG) ? bc4(SMALL) : (am < bc4(SMALL) ? bc4(BIG) : bc4(1.0));
          p0[q] *= sc; p1[q] *= sc;
        }
      }
    }
    for (int q = 0; q < NV; ++q)
      for (int r = 0; r < VW; ++r) c[VW * q + r] = (int)neg[q][r];
  };
  const double tol_abs = eps;   // scaled units: eps * ||T||
  for (int it = 0; it < 120; ++it) {
    bool any = false;
    for (int j = 0; j < mp; ++j) {
      mid[j] = 0.5 * (lo[j] + hi[j]);
      if (j < m && hi[j] - lo[j] > 2.0 * eps * std::max(std::fabs(lo[j]), std::fabs(hi[j])) + tol_abs) any = true;
    }
    if (!any) break;
    for (int j = 0; j < mp; j += W) sturm(&mid[j], &cnt[j]);
    for (int j = 0; j < m; ++j) {
      // eigenvalue j (0-based, ascending) lies in [lo, hi): count(mid) >= j+1  <=>  lambda_j < mid
      if (cnt[j] >= j + 1) hi[j] = mid[j]; else lo[j] = mid[j];
    }
  }
  w.resize(m);
  for (int j = 0; j < m; ++j) w[j] = 0.5 * (lo[j] + hi[j]) * onenrm;
  for (int j = 1; j < m; ++j) if (w[j] < w[j - 1]) w[j] = w[j - 1];   // keep the order monotone
}

// lowest m eigenpairs; zt rows 0..m-1 receive the eigenvectors of the ORIGINAL matrix
int sym_eig_lowest(int n, std::vector<double>& s, int m, std::vector<double>& w_all, std::vector<double>& zt)
{
  static thread_local Tridiag t;             // (work arrays live as long as the thread: the solver runs once per iteration of every driver)
  tridiagonalize(n, s, t);
  double onenrm = 0.0;
  for (int i = 0; i < n; ++i) {
    double r = std::fabs(t.d[i]) + (i > 0 ? std::fabs(t.e[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(t.e[i]) : 0.0);
    onenrm = std::max(onenrm, r);
  }
  if (onenrm == 0.0) {                       // the zero matrix: every vector is an eigenvector
    w_all.assign(m, 0.0);
    std::fill(zt.begin(), zt.end(), 0.0);
    for (int j = 0; j < m; ++j) zt[(size_t)j * n + j] = 1.0;
    return 0;
  }
  bisect_lowest(n, t.d, t.e, m, onenrm, w_all);        // w_all[0..m-1]
  const double eps = 2.220446049250313e-16;
  const double ortol = 1.0e-3 * onenrm;     // cluster criterion (as LAPACK dstein)
  const double sep = 10.0 * eps * onenrm;   // minimal separation of the shifts inside a cluster
  const double tiny = eps * onenrm;
  unsigned long long seed = 0x243F6A8885A308D3ULL;
  auto rnd = [&]() {   // deterministic start vectors
    seed = seed * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((double)(seed >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
  };
  // shifts: the eigenvalues, pulled apart inside clusters (as LAPACK dstein)
  std::vector<double> lam(m);
  std::vector<int> cstart(m);
  {
    int cluster_start = 0;
    double lam_prev = 0.0;
    for (int j = 0; j < m; ++j) {
      double lj = w_all[j];
      if (j > 0 && std::fabs(w_all[j] - w_all[j - 1]) >= ortol) cluster_start = j;
      if (j > cluster_start && lj - lam_prev < sep) lj = lam_prev + sep;
      lam_prev = lj;
      lam[j] = lj; cstart[j] = cluster_start;
    }
  }
  // The iterates stay in the [index][lane] layout of the solver: normalisation and the convergence test are vertical
  // operations on all sixteen lanes; a lane's column is only gathered for the rare re-orthogonalisation inside a cluster and
  // once at the end.  Work arrays live as long as the thread (the solver runs once per iteration of every driver).
  static thread_local TriLU16 lu;
  static thread_local std::vector<v8> xl_store;
  static thread_local std::vector<double> col_store;
  if (xl_store.size() < 2 * (size_t)n) xl_store.resize(2 * (size_t)n);
  if (col_store.size() < (size_t)n) col_store.resize(n);
  v8* xl = xl_store.data();
  double* xs = reinterpret_cast<double*>(xl);          // xs[i * LW + lane]
  double* col = col_store.data();
  for (int j0 = 0; j0 < m; j0 += LW) {
    const int nb = std::min(LW, m - j0);
    double lm[LW], wq[LW], slack[LW];
    bool clustered = false;
    for (int q = 0; q < LW; ++q) {
      const int j = j0 + std::min(q, nb - 1);          // (idle lanes shadow the last shift)
      lm[q] = lam[j]; wq[q] = w_all[j];
      slack[q] = 64.0 * eps * onenrm + 2.0 * std::fabs(lam[j] - w_all[j]);
      if (q < nb && cstart[j] < j) clustered = true;
    }
    for (int i = 0; i < n; ++i)
      for (int q = 0; q < LW; ++q) xs[(size_t)i * LW + q] = rnd();
    lu.factor(n, t.d, t.e, lm, tiny);
    v8 vw[2]; std::memcpy(vw, wq, sizeof vw);
    bool conv[LW];
    for (int q = 0; q < LW; ++q) conv[q] = false;
    for (int it = 0; it < 8; ++it) {
      lu.solve(n, xl);
      if (clustered) {
        // re-orthogonalise against the earlier members of the cluster (modified Gram-Schmidt): finished vectors of earlier
        // batches, then the lower lanes of this one (already orthogonalised in this iteration, not yet normalised: the
        // projection divides by their squared norm); zt rows are still in tridiagonal coordinates here
        for (int q = 0; q < nb; ++q) {
          const int j = j0 + q;
          if (cstart[j] >= j) continue;
          for (int i = 0; i < n; ++i) col[i] = xs[(size_t)i * LW + q];
          for (int p = cstart[j]; p < j; ++p) {
            double dot = 0.0, nn = 1.0;
            if (p < j0) {
              const double* zq = &zt[(size_t)p * n];
              for (int i = 0; i < n; ++i) dot += zq[i] * col[i];
              for (int i = 0; i < n; ++i) col[i] -= dot * zq[i];
            } else {
              const int qp = p - j0;
              nn = 0.0;
              for (int i = 0; i < n; ++i) { const double z = xs[(size_t)i * LW + qp]; dot += z * col[i]; nn += z * z; }
              if (nn > 0.0) { const double f = dot / nn; for (int i = 0; i < n; ++i) col[i] -= f * xs[(size_t)i * LW + qp]; }
            }
          }
          for (int i = 0; i < n; ++i) xs[(size_t)i * LW + q] = col[i];
        }
      }
      // largest entry and 2-norm of every lane
      v8 amax[2] = {bc8(0.0), bc8(0.0)};
      for (int i = 0; i < n; ++i)
        for (int h = 0; h < 2; ++h) { const v8 ax = abs8(xl[2 * i + h]); amax[h] = ax > amax[h] ? ax : amax[h]; }
      double am[LW]; std::memcpy(am, amax, sizeof am);
      bool all = true;
      double sc[LW];
      for (int q = 0; q < LW; ++q) {
        if (!(am[q] > 0.0) || !std::isfinite(am[q])) {
          // a lane that broke down starts again from a fresh vector
          for (int i = 0; i < n; ++i) xs[(size_t)i * LW + q] = rnd();
          sc[q] = 1.0; conv[q] = false; if (q < nb) all = false;
        } else sc[q] = 1.0 / am[q];
      }
      v8 vs[2]; std::memcpy(vs, sc, sizeof vs);
      v8 ss[2] = {bc8(0.0), bc8(0.0)};
      for (int i = 0; i < n; ++i)
        for (int h = 0; h < 2; ++h) { const v8 y = xl[2 * i + h] * vs[h]; xl[2 * i + h] = y; ss[h] += y * y; }
      double s2[LW]; std::memcpy(s2, ss, sizeof s2);
      for (int q = 0; q < LW; ++q) sc[q] = 1.0 / std::sqrt(s2[q]);
      std::memcpy(vs, sc, sizeof vs);
      for (int i = 0; i < n; ++i)
        for (int h = 0; h < 2; ++h) xl[2 * i + h] *= vs[h];
      if (it >= 1) {
        // converged when the eigen-residual is at rounding level (plus the shift perturbation)
        v8 rs[2] = {bc8(0.0), bc8(0.0)};
        for (int i = 0; i < n; ++i)
          for (int h = 0; h < 2; ++h) {
            v8 ti = (bc8(t.d[i]) - vw[h]) * xl[2 * i + h];
            if (i > 0) ti += bc8(t.e[i - 1]) * xl[2 * (i - 1) + h];
            if (i + 1 < n) ti += bc8(t.e[i]) * xl[2 * (i + 1) + h];
            rs[h] += ti * ti;
          }
        double r2[LW]; std::memcpy(r2, rs, sizeof r2);
        for (int q = 0; q < nb; ++q) {
          if (!(am[q] > 0.0) || !std::isfinite(am[q])) continue;
          conv[q] = std::sqrt(r2[q]) <= slack[q];
          all = all && conv[q];
        }
      } else all = false;
      if (it >= 1 && all) break;
    }
    for (int q = 0; q < nb; ++q) {
      double* z = &zt[(size_t)(j0 + q) * n];
      for (int i = 0; i < n; ++i) z[i] = xs[(size_t)i * LW + q];
    }
  }
  // back-transformation: eigenvector of S = H_0 H_1 ... H_{n-3} z  (apply the last reflector first).
  // All m vectors advance together: zz[r][j] = component r of vector j, so both loops run along j.
  const int mp = (m + 7) & ~7;
  std::vector<double> zz((size_t)n * mp, 0.0), acc(mp);
  for (int j = 0; j < m; ++j)
    for (int r = 0; r < n; ++r) zz[(size_t)r * mp + j] = zt[(size_t)j * n + r];
  for (int k = n - 3; k >= 0; --k) {
    const double beta = t.hbeta[k];
    if (beta == 0.0) continue;
    const double* vk = &t.hv[(size_t)k * n];
    for (int j = 0; j < mp; ++j) acc[j] = 0.0;
    for (int r = k + 1; r < n; ++r) {
      const double vr = vk[r];
      const double* zr = &zz[(size_t)r * mp];
      for (int j = 0; j < mp; ++j) acc[j] += vr * zr[j];
    }
    for (int j = 0; j < mp; ++j) acc[j] *= beta;
    for (int r = k + 1; r < n; ++r) {
      const double vr = vk[r];
      double* zr = &zz[(size_t)r * mp];
      for (int j = 0; j < mp; ++j) zr[j] -= acc[j] * vr;
    }
  }
  for (int j = 0; j < m; ++j)
    for (int r = 0; r < n; ++r) zt[(size_t)j * n + r] = zz[(size_t)r * mp + j];
  return 0;
}

}  // namespace SD_NS

#ifndef SD_IMPL_ONLY
#ifdef SD_SINGLE_ISA      // (builds that compile this file once: the AVX2 routines serve every host)
namespace sd_v4 = sd_v3;
#else
namespace sd_v4 {   // the same routines, AVX-512 build of this file
int sym_eig(int n, std::vector<double>& s, std::vector<double>& d, std::vector<double>& zt);
int sym_eig_lowest(int n, std::vector<double>& s, int m, std::vector<double>& w_all, std::vector<double>& zt);
}
#endif
namespace {
bool wide_simd()
{
#if defined(__x86_64__)
  static const bool yes = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
                          std::getenv("DIAGLIB_AMD_NO_AVX512") == nullptr;
  return yes;
#else
  return false;
#endif
}
int sym_eig(int n, std::vector<double>& s, std::vector<double>& d, std::vector<double>& zt)
{
  return wide_simd() ? sd_v4::sym_eig(n, s, d, zt) : sd_v3::sym_eig(n, s, d, zt);
}
int sym_eig_lowest(int n, std::vector<double>& s, int m, std::vector<double>& w_all, std::vector<double>& zt)
{
  return wide_simd() ? sd_v4::sym_eig_lowest(n, s, m, w_all, zt) : sd_v3::sym_eig_lowest(n, s, m, w_all, zt);
}
inline double& at(double* a, int ld, int i, int j) { return a[(size_t)i + (size_t)j * ld]; }
inline double at(const double* a, int ld, int i, int j) { return a[(size_t)i + (size_t)j * ld]; }
}  // namespace

extern "C" {

int dla_syev(char uplo, int n, double* a, int lda, double* w)
{
  DLA_T("dla_syev");
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

// lowest m eigenpairs only: w(1:m) ascending (w(m+1:n) is set to w(m): the drivers never read it, reference
// :1715 eig = e_red(1:n_max), :1698 e_red(1:n_rst)), a(:,1:m) the eigenvectors (columns m+1..n undefined).  Same role as dsyev at diaglib.f90:1708 / :406, whose
// callers use the first n_max eigenpairs only.
int dla_syev_lowest(char uplo, int n, double* a, int lda, double* w, int m)
{
  DLA_T("dla_syev_lowest");
  if (n <= 0) return 0;
  if (lda < n) return -1;
  if (m > n) m = n;
  // (measured: the partial solver wins from n = 21 on whatever the share of wanted pairs -- 230 us against 568 at n = 74, m = 37,
  //  24 against 34 at n = 26, m = 13; below 20 the full one is 20-50 % faster)
  if (n <= 20 || m >= n) return dla_syev(uplo, n, a, lda, w);   // small or all wanted: full solve
  bool up = (uplo == 'u' || uplo == 'U');
  static thread_local std::vector<double> s, wall, zt;
  if (s.size() < (size_t)n * n) s.resize((size_t)n * n);
  if (zt.size() < (size_t)m * n) zt.resize((size_t)m * n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      double x = up ? at(a, lda, i, j) : at(a, lda, j, i);
      s[(size_t)i * n + j] = x;
      s[(size_t)j * n + i] = x;
    }
  int info = sym_eig_lowest(n, s, m, wall, zt);
  if (info != 0) return info;
  for (int j = 0; j < n; ++j) w[j] = wall[j < m ? j : m - 1];
  for (int j = 0; j < m; ++j) {
    const double* z = &zt[(size_t)j * n];
    int imax = 0;
    for (int i = 1; i < n; ++i) if (std::fabs(z[i]) > std::fabs(z[imax])) imax = i;
    double sg = (z[imax] < 0.0) ? -1.0 : 1.0;
    for (int i = 0; i < n; ++i) at(a, lda, i, j) = sg * z[i];
  }
  return 0;
}

int dla_potrf_lower(int m, double* a, int lda)
{
  DLA_T("dla_potrf_lower");
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
  DLA_T("dla_trtri_lower");
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
#endif  // SD_IMPL_ONLY
