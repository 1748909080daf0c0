// tools/rr_device.hip -- EXPERIMENT (not part of the library, see DESIGN.md 11, docs/HISTORY.md 13.4): Rayleigh-Ritz on the device, the lowest p
// eigenpairs of the projected matrix by one workgroup.  Built and timed by tools/rr_probe.hip.
//
// The reference solves the projected problem with LAPACK's dsyev on the host (diaglib.f90:1708) and uses the first n_max
// eigenpairs (:1715-1721).  With the panels resident in HBM that solve is the only reason why the host has to wait in the
// middle of an iteration: projection -> [download, dsyev, upload] -> Ritz sweep.  The round trip costs more than the solve
// itself at these orders (n <= 128), so the solve moves next to the data: ONE workgroup, the working copy of the matrix in LDS.
//
// Same algorithm as the host solver (smalldense.cpp: sym_eig_lowest), re-laid for 256 lanes:
//   A  Householder tridiagonalisation, one column per step; the rank-2 update of step k-1 and the product A v of step k share
//      one pass over the trailing matrix in LDS (lane = column, two lanes per column);
//   B  the p lowest eigenvalues by multisection on Sturm counts: 256 shifts per round (256 / p inside every eigenvalue's
//      interval) instead of one -- the division-free, rescaled recurrence of the host code;
//   C  eigenvectors of the tridiagonal matrix by inverse iteration, one lane per eigenvector (pivoted LU of T - lambda I in
//      LDS, shifts pulled apart and re-orthogonalisation inside clusters as in LAPACK's dstein);
//   D  back-transformation with the reflectors, one 16-lane group per eigenvector, the vector in registers;
//   E  sign convention of dla_syev_lowest (largest component positive), results to device memory and pinned mirrors.
#include "rr_device.h"

namespace dla_rr {
namespace {

constexpr int NT = 256;
constexpr int SMALL_DOUBLES = 1536;          // vectors and scalars in front of the matrix
typedef double __attribute__((may_alias)) lds_f64;

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wmax(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ unsigned long long wall_clock() { return __builtin_readcyclecounter(); }
#define WSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// number of eigenvalues of the (scaled) tridiagonal matrix below x: sign changes of the leading principal minors
// p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}, rescaled by a power of two every fourth step (smalldense.cpp: bisect_lowest)
__device__ int sturm_count(int n, const lds_f64* ds, const lds_f64* e2, double x)
{
  const double BIG = 0x1p+300, SMALL = 0x1p-300, ZERO_REPL = 0x1p-900;
  double p0 = 1.0, p1 = ds[0] - x;
  int neg = 0;
  if (p1 == 0.0) p1 = -ZERO_REPL;
  neg += (p1 < 0.0) ? 1 : 0;
  for (int i = 1; i < n; ++i) {
    const double t = (ds[i] - x) * p1 - e2[i - 1] * p0;
    p0 = p1; p1 = t;
    // an exact zero takes the sign opposite to its predecessor (it counts as a negative pivot)
    if (p1 == 0.0) p1 = (p0 < 0.0) ? ZERO_REPL : -ZERO_REPL;
    neg += ((p1 < 0.0) != (p0 < 0.0)) ? 1 : 0;
    if ((i & 3) == 3) {
      const double am = fmax(fabs(p1), fabs(p0));
      const double sc = am > BIG ? SMALL : (am < SMALL ? BIG : 1.0);
      p0 *= sc; p1 *= sc;
    }
  }
  return neg;
}

__device__ __forceinline__ double section_point(double l, double h, int s, int S)
{
  return l + (h - l) * ((double)(s + 1) / (double)(S + 1));
}

__device__ __forceinline__ double start_value(int i, int q, int salt)
{
  unsigned long long s = ((unsigned long long)(i * 16 + q + 1) + ((unsigned long long)salt << 20)) * 6364136223846793005ULL + 1442695040888963407ULL;
  s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ULL; s ^= s >> 32;
  return ((double)(s >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
}

__global__ __launch_bounds__(256) void rr_lowest_kernel(Args a)
{
  if (a.go != nullptr && *a.go != a.go_want) return;
  extern __shared__ __attribute__((aligned(16))) double sm_raw[];
  lds_f64* sm = (lds_f64*)sm_raw;
  const int n = a.n, p = a.p, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int LDA = n | 1;
  lds_f64* vb0 = sm;                 // Householder vector of the current / previous step (by absolute index)
  lds_f64* vb1 = sm + 128;
  lds_f64* wb  = sm + 256;           // w of the pending rank-2 update
  lds_f64* pp  = sm + 384;           // [2][128] partial products
  lds_f64* dd  = sm + 640;           // diagonal of T
  lds_f64* ee  = sm + 768;           // sub-diagonal of T
  lds_f64* tau = sm + 896;           // beta of every reflector
  lds_f64* ds  = sm + 1024;          // scaled diagonal
  lds_f64* e2  = sm + 1152;          // scaled squared sub-diagonal
  lds_f64* red = sm + 1280;          // [16] reduction slots
  lds_f64* lo  = sm + 1296;          // [16]
  lds_f64* hi  = sm + 1312;          // [16]
  lds_f64* lam = sm + 1328;          // [16] shifts of the inverse iteration
  lds_f64* wv  = sm + 1344;          // [16] eigenvalues
  int* cnt = (int*)(sm_raw + 1360);  // [256] Sturm counts of a round
  int* cst = (int*)(sm_raw + 1488);  // [16] first member of every eigenvalue's cluster, [16] = any cluster
  lds_f64* A = sm + SMALL_DOUBLES;   // n x LDA working copy; later the LU factors, then the reflectors
  if (t == 0 && a.dbg) a.dbg[0] = wall_clock();

  // ---- the matrix, symmetric, into LDS
  for (int idx = t; idx < n * n; idx += NT) {
    const int i = idx / n, j = idx - i * n;
    const int l_ = i < j ? i : j, h_ = i < j ? j : i;
    A[i * LDA + j] = a.upper ? a.h[(size_t)l_ + (size_t)h_ * a.ldh] : a.h[(size_t)h_ + (size_t)l_ * a.ldh];
  }
  __syncthreads();

  // ---- A: tridiagonalisation.  H_k = I - beta v v^T annihilates column k below the sub-diagonal;
  //      A <- H A H = A - v w^T - w v^T with p = beta A v, w = p - (beta p^T v / 2) v.  The update of step k stays pending
  //      (vprev, wb) and is applied by the pass of step k + 1, which forms A v of that step from the updated entries.
  bool have_prev = false;
  for (int k = 0; k + 2 < n; ++k) {
    lds_f64* vcur = (k & 1) ? vb1 : vb0;
    lds_f64* vprev = (k & 1) ? vb0 : vb1;
    const int r = n - k - 1;               // order of the trailing matrix (rows / columns k+1 .. n-1)
    const int i1 = k + 1 + t;              // the row this lane looks after in the vector steps
    double x = 0.0;
    if (t < r) {
      x = A[i1 * LDA + k];
      if (have_prev) x -= vprev[i1] * wb[k] + wb[i1] * vprev[k];
    }
    if (t == 0) {
      double dk = A[k * LDA + k];
      if (have_prev) dk -= 2.0 * vprev[k] * wb[k];
      dd[k] = dk;
      red[4] = x;
    }
    const double s2 = wsum((t >= 1 && t < r) ? x * x : 0.0);
    if (lane == 0) red[wave] = s2;
    __syncthreads();
    const double rest = (red[0] + red[1]) + (red[2] + red[3]);
    const double x0 = red[4];
    double beta = 0.0, sub = x0, v0 = x0;
    if (rest != 0.0) {
      const double nrm = sqrt(x0 * x0 + rest);
      const double alpha = (x0 >= 0.0) ? -nrm : nrm;
      v0 = x0 - alpha;
      sub = alpha;
      beta = 2.0 / (v0 * v0 + rest);
    }
    if (t < r) {
      const double vm = (beta != 0.0) ? ((t == 0) ? v0 : x) : 0.0;
      vcur[i1] = vm;
      a.refl[(size_t)k * n + i1] = vm;
    }
    if (t == 0) { tau[k] = beta; ee[k] = sub; }
    __syncthreads();
    {
      const int c = t & 127, hf = t >> 7;
      if (c < r) {
        const int half = (r + 1) >> 1;
        const int j = k + 1 + c;
        const int ib = k + 1 + hf * half, ie = (ib + half < n) ? ib + half : n;
        double acc0 = 0.0, acc1 = 0.0;
        if (have_prev) {
          const double vpj = vprev[j], wpj = wb[j];
          int i = ib;
          for (; i + 1 < ie; i += 2) {
            double a0 = A[i * LDA + j], a1 = A[(i + 1) * LDA + j];
            a0 -= vprev[i] * wpj + wb[i] * vpj;
            a1 -= vprev[i + 1] * wpj + wb[i + 1] * vpj;
            A[i * LDA + j] = a0; A[(i + 1) * LDA + j] = a1;
            acc0 += a0 * vcur[i]; acc1 += a1 * vcur[i + 1];
          }
          if (i < ie) {
            double a0 = A[i * LDA + j];
            a0 -= vprev[i] * wpj + wb[i] * vpj;
            A[i * LDA + j] = a0;
            acc0 += a0 * vcur[i];
          }
        } else {
          int i = ib;
          for (; i + 1 < ie; i += 2) { acc0 += A[i * LDA + j] * vcur[i]; acc1 += A[(i + 1) * LDA + j] * vcur[i + 1]; }
          if (i < ie) acc0 += A[i * LDA + j] * vcur[i];
        }
        pp[hf * 128 + c] = acc0 + acc1;
      }
    }
    __syncthreads();
    double pj = 0.0, vj = 0.0;
    if (t < r) { pj = beta * (pp[t] + pp[128 + t]); vj = vcur[i1]; }
    const double pv = wsum(pj * vj);
    if (lane == 0) red[8 + wave] = pv;
    __syncthreads();
    const double kk = 0.5 * beta * ((red[8] + red[9]) + (red[10] + red[11]));
    if (t < r) wb[i1] = pj - kk * vj;
    have_prev = (beta != 0.0);
    __syncthreads();
  }
  if (t == 0) {
    if (n == 1) dd[0] = A[0];
    else {
      const int i = n - 2, j = n - 1;
      double dii = A[i * LDA + i], djj = A[j * LDA + j], eij = A[j * LDA + i];
      if (have_prev) {
        const lds_f64* vp = ((n - 3) & 1) ? vb1 : vb0;
        dii -= 2.0 * vp[i] * wb[i];
        djj -= 2.0 * vp[j] * wb[j];
        eij -= vp[j] * wb[i] + wb[j] * vp[i];
      }
      dd[i] = dii; dd[j] = djj; ee[i] = eij;
    }
    if (a.dbg) a.dbg[1] = wall_clock();
  }
  __syncthreads();

  // ---- B: eigenvalues.  Norm, scaling, Gershgorin bounds
  const double eps = 2.220446049250313e-16;
  {
    double rown = 0.0, off = 0.0, di = 0.0;
    if (t < n) {
      off = (t > 0 ? fabs(ee[t - 1]) : 0.0) + (t + 1 < n ? fabs(ee[t]) : 0.0);
      di = dd[t];
      rown = fabs(di) + off;
    }
    const double m1 = wmax(rown);
    if (lane == 0) red[wave] = m1;
    __syncthreads();
    const double onenrm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    if (!(onenrm > 0.0) || !(onenrm < 1.0e300)) {
      // the zero matrix (every vector is an eigenvector: unit vectors), or a matrix with non-finite entries (reported)
      const bool bad = !(onenrm == 0.0);
      for (int idx = t; idx < n * p; idx += NT) {
        const int i = idx % n, q = idx / n;
        const double v = (i == q) ? 1.0 : 0.0;
        a.y[(size_t)i + (size_t)q * a.ldy] = v;
        if (a.y_host) a.y_host[(size_t)i + (size_t)q * a.ldy] = v;
      }
      if (t < p) { a.eig[t] = 0.0; if (a.eig_host) a.eig_host[t] = 0.0; }
      __syncthreads();
      if (t == 0) { __threadfence_system(); *a.status = bad ? -a.seq : a.seq; if (a.status_host) *a.status_host = bad ? -a.seq : a.seq; }
      return;
    }
    const double inv = 1.0 / onenrm;
    if (t < n) { ds[t] = di * inv; if (t + 1 < n) { const double es = ee[t] * inv; e2[t] = es * es; } }
    const double glm = (t < n) ? (di - off) * inv : 1.0e300, gum = (t < n) ? (di + off) * inv : -1.0e300;
    const double g1 = -wmax(-glm), g2 = wmax(gum);
    if (lane == 0) { red[4 + wave] = g1; red[8 + wave] = g2; }
    if (t == 0) red[12] = onenrm;
    __syncthreads();
    const double pad = 4.0 * eps * n;
    const double gl = fmin(fmin(red[4], red[5]), fmin(red[6], red[7])) - pad;
    const double gu = fmax(fmax(red[8], red[9]), fmax(red[10], red[11])) + pad;
    if (t < RR_PMAX) { lo[t] = gl; hi[t] = gu; }
    __syncthreads();
  }
  const double onenrm = red[12];
  {
    const int S = NT / p;                 // shifts per eigenvalue and round
    const int q = t / S, s = t - q * S;
    for (int round = 0; round < 80; ++round) {
      bool any = false;
      for (int j = 0; j < p; ++j) {
        const double l_ = lo[j], h_ = hi[j];
        if (h_ - l_ > 2.0 * eps * fmax(fabs(l_), fabs(h_)) + eps) any = true;
      }
      if (!any) break;
      if (q < p) cnt[t] = sturm_count(n, ds, e2, section_point(lo[q], hi[q], s, S));
      __syncthreads();
      if (t < p) {
        // eigenvalue t lies in [lo, hi): count(x) >= t + 1  <=>  lambda_t < x
        int first_above = S;
        for (int s2 = 0; s2 < S; ++s2)
          if (cnt[t * S + s2] >= t + 1) { first_above = s2; break; }
        const double l_ = lo[t], h_ = hi[t];
        if (first_above > 0) lo[t] = section_point(l_, h_, first_above - 1, S);
        if (first_above < S) hi[t] = section_point(l_, h_, first_above, S);
      }
      __syncthreads();
    }
  }
  // shifts of the inverse iteration: the eigenvalues, pulled apart inside clusters (as LAPACK dstein / the host code)
  const double ortol = 1.0e-3 * onenrm, sep = 10.0 * eps * onenrm, tiny = eps * onenrm;
  if (t == 0) {
    double prev = 0.0;
    for (int j = 0; j < p; ++j) {
      double w = 0.5 * (lo[j] + hi[j]) * onenrm;
      if (j > 0 && w < prev) w = prev;     // keep the order monotone
      wv[j] = w; prev = w;
    }
    int cluster_start = 0, anyc = 0;
    double lam_prev = 0.0;
    for (int j = 0; j < p; ++j) {
      double lj = wv[j];
      if (j > 0 && fabs(wv[j] - wv[j - 1]) >= ortol) cluster_start = j;
      if (j > cluster_start && lj - lam_prev < sep) lj = lam_prev + sep;
      lam_prev = lj;
      lam[j] = lj; cst[j] = cluster_start;
      if (cluster_start < j) anyc = 1;
    }
    cst[16] = anyc;
    if (a.dbg) a.dbg[2] = wall_clock();
  }
  __syncthreads();

  // ---- C: inverse iteration, lane q of wave 0 owns eigenvector q.  [index][16] arrays in the matrix' place.
  lds_f64* La = A;                       // reciprocal pivots
  lds_f64* Lb = A + 16 * n;              // first super-diagonal of U
  lds_f64* Lc = A + 32 * n;              // second super-diagonal of U
  lds_f64* Ll = A + 48 * n;              // multipliers
  lds_f64* X  = A + 64 * n;              // the iterates
  if (wave == 0) {
    const int q = lane;
    const bool act = q < p;
    const double lm = act ? lam[q] : 0.0, wq = act ? wv[q] : 0.0;
    const double slack = 64.0 * eps * onenrm + 2.0 * fabs(lm - wq);
    const bool clustered = cst[16] != 0;
    const int my_start = act ? cst[q] : 0;
    unsigned long long piv0 = 0ULL, piv1 = 0ULL;
    if (act) {
      const double rtiny = 1.0 / tiny;
      double ai = dd[0] - lm, bi = (n > 1) ? ee[0] : 0.0;
      for (int i = 0; i + 1 < n; ++i) {
        const double sub = ee[i];
        const double an = dd[i + 1] - lm, bn = (i + 2 < n) ? ee[i + 1] : 0.0;
        const bool swp = fabs(ai) < fabs(sub);
        const double a0 = (ai == 0.0) ? tiny : ai;
        const double piv = swp ? sub : a0;
        const double rp = 1.0 / piv;
        const double mlt = (swp ? ai : sub) * rp;
        La[16 * i + q] = (fabs(piv) < tiny) ? ((piv < 0.0) ? -rtiny : rtiny) : rp;
        Lb[16 * i + q] = swp ? an : bi;
        Lc[16 * i + q] = swp ? bn : 0.0;
        Ll[16 * i + q] = mlt;
        if (swp) { if (i < 64) piv0 |= 1ULL << i; else piv1 |= 1ULL << (i - 64); }
        const double na = swp ? bi - mlt * an : an - mlt * bi;
        const double nb = swp ? -mlt * bn : bn;
        ai = na; bi = nb;
      }
      La[16 * (n - 1) + q] = (fabs(ai) < tiny) ? ((ai < 0.0) ? -rtiny : rtiny) : 1.0 / ai;
      Lb[16 * (n - 1) + q] = 0.0; Lc[16 * (n - 1) + q] = 0.0;
      for (int i = 0; i < n; ++i) X[16 * i + q] = start_value(i, q, 0);
    }
    WSYNC();
    bool conv = !act;
    for (int it = 0; it < 8; ++it) {
      if (act) {
        // L y = P x (forward, with the row interchanges of the factorisation), U x = y (backward)
        double xi = X[q];
        for (int i = 0; i + 1 < n; ++i) {
          const double xn = X[16 * (i + 1) + q];
          const bool swp = (i < 64) ? ((piv0 >> i) & 1ULL) != 0ULL : ((piv1 >> (i - 64)) & 1ULL) != 0ULL;
          const double top = swp ? xn : xi, bot = swp ? xi : xn;
          X[16 * i + q] = top;
          xi = bot - Ll[16 * i + q] * top;
        }
        X[16 * (n - 1) + q] = xi;
        double x1 = 0.0, x2 = 0.0;
        for (int i = n - 1; i >= 0; --i) {
          const double v = (X[16 * i + q] - Lb[16 * i + q] * x1 - Lc[16 * i + q] * x2) * La[16 * i + q];
          X[16 * i + q] = v;
          x2 = x1; x1 = v;
        }
      }
      WSYNC();
      if (clustered) {
        // modified Gram-Schmidt against the earlier members of the cluster (orthogonalised in this iteration, not yet
        // normalised: the projection divides by their squared norm); lane qq is final when its turn comes
        for (int qq = 0; qq + 1 < p; ++qq) {
          if (act && my_start <= qq && qq < q) {
            double dot = 0.0, nn = 0.0;
            for (int i = 0; i < n; ++i) { const double z = X[16 * i + qq]; dot += z * X[16 * i + q]; nn += z * z; }
            if (nn > 0.0) { const double f = dot / nn; for (int i = 0; i < n; ++i) X[16 * i + q] -= f * X[16 * i + qq]; }
          }
          WSYNC();
        }
      }
      bool ok_lane = true;
      if (act) {
        double am = 0.0;
        for (int i = 0; i < n; ++i) am = fmax(am, fabs(X[16 * i + q]));
        if (!(am > 0.0) || !(am < 1.0e300)) {
          // a lane that broke down starts again from a fresh vector
          for (int i = 0; i < n; ++i) X[16 * i + q] = start_value(i, q, it + 1);
          ok_lane = false; conv = false;
        } else {
          const double sc = 1.0 / am;
          double ss = 0.0;
          for (int i = 0; i < n; ++i) { const double y = X[16 * i + q] * sc; X[16 * i + q] = y; ss += y * y; }
          const double sn = 1.0 / sqrt(ss);
          for (int i = 0; i < n; ++i) X[16 * i + q] *= sn;
          if (it >= 1) {
            // converged when the eigen-residual is at rounding level (plus the shift perturbation)
            double rs = 0.0, xm = 0.0, xc = X[q];
            for (int i = 0; i < n; ++i) {
              const double xp = (i + 1 < n) ? X[16 * (i + 1) + q] : 0.0;
              double ti = (dd[i] - wq) * xc;
              if (i > 0) ti += ee[i - 1] * xm;
              if (i + 1 < n) ti += ee[i] * xp;
              rs += ti * ti;
              xm = xc; xc = xp;
            }
            conv = sqrt(rs) <= slack;
          }
        }
      }
      WSYNC();
      const bool all = __all((it >= 1 && conv && ok_lane) || !act);
      if (all) break;
    }
  }
  if (t == 0 && a.dbg) a.dbg[3] = wall_clock();
  __syncthreads();

  // ---- D: back-transformation  y = H_0 H_1 ... H_{n-3} z  (the last reflector first).  The 16-lane group q holds eigenvector q
  //      in registers, lane g the components g, g + 16, ...
  const int q = t >> 4, g = t & 15;
  double z[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int i = g + 16 * rr;
    z[rr] = (q < p && i < n) ? (double)X[16 * i + q] : 0.0;
  }
  __syncthreads();
  // the reflectors back from scratch memory (written by this workgroup above), row k = reflector k
  for (int idx = t; idx < (n - 2) * n; idx += NT) A[idx] = a.refl[idx];
  __syncthreads();
  for (int k = n - 3; k >= 0; --k) {
    const double beta = tau[k];
    if (beta == 0.0) continue;
    double v[8], s = 0.0;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int i = g + 16 * rr;
      v[rr] = (i > k && i < n) ? (double)A[k * n + i] : 0.0;
      s += v[rr] * z[rr];
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    s *= beta;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) z[rr] -= s * v[rr];
  }
  // ---- E: largest component positive (the first one among equals), results out
  double best = -1.0; int bidx = 1 << 30;
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int i = g + 16 * rr;
    const double av = fabs(z[rr]);
    if (i < n && (av > best || (av == best && i < bidx))) { best = av; bidx = i; }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const double ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  }
  double zbest = 0.0;
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) if (g + 16 * rr == bidx) zbest = z[rr];
  zbest += __shfl_xor(zbest, 1, 64); zbest += __shfl_xor(zbest, 2, 64); zbest += __shfl_xor(zbest, 4, 64); zbest += __shfl_xor(zbest, 8, 64);
  const double sg = (zbest < 0.0) ? -1.0 : 1.0;
  bool finite = true;
  if (q < p) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int i = g + 16 * rr;
      if (i < n) {
        const double v = sg * z[rr];
        if (!(fabs(v) < 1.0e300)) finite = false;
        a.y[(size_t)i + (size_t)q * a.ldy] = v;
        if (a.y_host) a.y_host[(size_t)i + (size_t)q * a.ldy] = v;
      }
    }
  }
  if (t < p) {
    const double w = wv[t];
    if (!(fabs(w) < 1.0e300)) finite = false;
    a.eig[t] = w;
    if (a.eig_host) a.eig_host[t] = w;
  }
  const int okall = __syncthreads_and(finite ? 1 : 0);
  if (t == 0) {
    __threadfence_system();
    const int stv = okall ? a.seq : -a.seq;
    *a.status = stv;
    if (a.status_host) *a.status_host = stv;
    if (a.dbg) a.dbg[4] = wall_clock();
  }
}

}  // namespace

size_t lds_bytes(int n)
{
  const size_t lda = (size_t)(n | 1);
  const size_t mat = (size_t)n * lda, lu = (size_t)80 * n;
  return sizeof(double) * (SMALL_DOUBLES + (mat > lu ? mat : lu));
}

hipError_t enqueue(hipStream_t st, const Args& a)
{
  if (a.n < 1 || a.n > RR_NMAX || a.p < 1 || a.p > RR_PMAX || a.p > a.n || a.ldh < a.n || a.ldy < a.n || !a.h || !a.y || !a.eig ||
      !a.refl || !a.status)
    return hipErrorInvalidValue;
  const size_t lds = lds_bytes(a.n);
  static size_t raised = 0;
  if (lds > 64 * 1024 && lds > raised) {
    hipError_t e = hipFuncSetAttribute((const void*)rr_lowest_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - 512));
    if (e != hipSuccess) return e;
    raised = (size_t)160 * 1024 - 512;
  }
  if (lds > (size_t)160 * 1024 - 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rr_lowest_kernel, dim3(1), dim3(NT), lds, st, a);
  return hipGetLastError();
}

}  // namespace dla_rr
