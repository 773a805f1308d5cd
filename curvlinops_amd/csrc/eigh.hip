// Hand-written pieces of the symmetric eigensolver behind the tridiagonalisation of sytrd.hip (reference call
// sites: computers/_base.py:355-372, kronecker.py:292-300 -- torch.linalg.eigh, i.e. rocSOLVER ssyevd):
//
//   clo_larft_f32          triangular factors T of the block reflectors of the back-transformation
//                          Q = prod_p (I - V_p T_p V_p^T)   (the GEMMs around it run on gemm.hip)
//   clo_tql2_batched_f32   eigen-decomposition of many small symmetric TRIDIAGONAL matrices (the leaves of the
//                          divide & conquer below): one thread per matrix, implicit QL in float64
//   clo_dc_*               the merge step of Cuppen's divide & conquer for the tridiagonal eigenproblem,
//                          batched over the nodes of one tree level: deflation scan, secular equation
//                          (bisection in float64 on the shifted variable), Gu-Eisenstat weights, eigenvector
//                          matrix of the rank-one update, Givens rotations of the deflation
//
// Everything O(n^3) -- the merges Q_children @ M -- is a batched GEMM on gemm.hip.
#include <algorithm>

#include "clo_common.h"
#include "gemm.h"

namespace clo {

// ------------------------------------------------------------------------------------------
// T of a block reflector (LAPACK larft, forward / columnwise) from the Gram matrix G = V^T V of its
// nb <= 64 reflectors:  T[i][i] = tau_i,  T[0:i, i] = -tau_i T[0:i, 0:i] G[0:i, i].
// One wave per block; lane r owns row r of T (kept in registers, static indexing via full unrolling
// would be 64^2/2 registers: rows live in LDS instead, one column per step).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void larft_kernel(const float *__restrict__ G, const float *__restrict__ tau,
                                                   float *__restrict__ T, int nb) {
  __shared__ float s_t[64][65], s_g[64][65];
  const int p = blockIdx.x, lane = threadIdx.x;
  const float *Gp = G + (long)p * nb * nb;
  float *Tp = T + (long)p * nb * nb;
  for (int e = lane; e < nb * nb; e += 64) {
    s_g[e / nb][e % nb] = Gp[e];
    s_t[e / nb][e % nb] = 0.f;
  }
  __syncthreads();
  for (int i = 0; i < nb; ++i) {
    const float ti = tau[(long)p * nb + i];
    float acc = 0.f;
    if (lane < i) {  // row `lane` of T[0:i,0:i] (upper triangular: columns lane .. i-1) times G[0:i, i]
      for (int c = lane; c < i; ++c) acc = fmaf(s_t[lane][c], s_g[c][i], acc);
    }
    __syncthreads();
    if (lane < i) s_t[lane][i] = -ti * acc;
    if (lane == i) s_t[i][i] = ti;
    __syncthreads();
  }
  for (int e = lane; e < nb * nb; e += 64) Tp[e] = s_t[e / nb][e % nb];
}

// ------------------------------------------------------------------------------------------
// Small symmetric tridiagonal eigenproblems, one WAVE each (EISPACK tql2 / LAPACK steqr algebra: implicit QL
// with Wilkinson shifts), float64 inside.  The scalar recurrence is computed redundantly by every lane (uniform
// values), the eigenvector matrix lives in LDS and lane k owns its row k: one rotation = one LDS update per lane.
// d[b][L], e[b][L] (e[0..L-2] sub-diagonal), lam[b][L] ascending, Q[b][L][L] eigenvectors in COLUMNS.  L <= 64.
// ------------------------------------------------------------------------------------------
constexpr int TQL_MAXL = 64;
__global__ __launch_bounds__(64) void tql2_kernel(const float *__restrict__ dd, const float *__restrict__ ee,
                                                  float *__restrict__ lam, float *__restrict__ Q, int L,
                                                  int *__restrict__ status) {
  __shared__ double s_z[TQL_MAXL][TQL_MAXL + 1];
  __shared__ double s_d[TQL_MAXL], s_e[TQL_MAXL];
  const int b = blockIdx.x, k = threadIdx.x;
  if (k < L) {
    s_d[k] = dd[(long)b * L + k];
    s_e[k] = k + 1 < L ? ee[(long)b * L + k] : 0.0;
    for (int j = 0; j < L; ++j) s_z[k][j] = k == j ? 1.0 : 0.0;
  }
  __syncthreads();
  int bad = 0;
  for (int l = 0; l < L; ++l) {
    int iter = 0;
    while (true) {
      int m = l;
      for (; m + 1 < L; ++m) {
        const double tst = fabs(s_d[m]) + fabs(s_d[m + 1]);
        if (fabs(s_e[m]) <= 2.220446049250313e-16 * tst) break;
      }
      if (m == l) break;
      if (++iter > 60) { bad = l + 1; break; }
      const double dl = s_d[l], el = s_e[l];
      double g = (s_d[l + 1] - dl) / (2.0 * el);
      double r = hypot(g, 1.0);
      g = s_d[m] - dl + el / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
      double s = 1.0, c = 1.0, p = 0.0;
      int i = m - 1;
      for (; i >= l; --i) {
        const double ei = s_e[i];
        double f = s * ei;
        const double bb = c * ei;
        r = hypot(f, g);
        __syncthreads();           // (uniform control flow: every lane takes the same path)
        if (k == 0) s_e[i + 1] = r;
        if (r == 0.0) {            // recover from underflow
          if (k == 0) { s_d[i + 1] -= p; s_e[m] = 0.0; }
          break;
        }
        s = f / r;
        c = g / r;
        g = s_d[i + 1] - p;
        r = (s_d[i] - g) * s + 2.0 * c * bb;
        p = s * r;
        __syncthreads();
        if (k == 0) s_d[i + 1] = g + p;
        g = c * r - bb;
        if (k < L) {               // lane k rotates its row of the eigenvector matrix
          f = s_z[k][i + 1];
          const double zi = s_z[k][i];
          s_z[k][i + 1] = s * zi + c * f;
          s_z[k][i] = c * zi - s * f;
        }
      }
      __syncthreads();
      if (r == 0.0 && i >= l) continue;
      if (k == 0) { s_d[l] -= p; s_e[l] = g; s_e[m] = 0.0; }
      __syncthreads();
    }
    if (bad) break;
  }
  __syncthreads();
  // rank of every eigenvalue (ties broken by index) -> ascending order, columns moved along
  if (k < L) {
    const double dk = s_d[k];
    int rank = 0;
    for (int j = 0; j < L; ++j) rank += (s_d[j] < dk || (s_d[j] == dk && j < k)) ? 1 : 0;
    lam[(long)b * L + rank] = (float)dk;
    for (int r2 = 0; r2 < L; ++r2) Q[((long)b * L + r2) * L + rank] = (float)s_z[r2][k];
  }
  if (bad && k == 0) atomicMax(status, bad);
}

// ------------------------------------------------------------------------------------------
// Divide & conquer merge (Cuppen; LAPACK slaed1-4 algebra), batched over the `nodes` of one tree level.
// A node joins two solved halves: D = diag(sorted child eigenvalues) [s], z [s] (rows of the children's
// eigenvector matrices at the cut, ||z|| = 1), rho > 0:  eig(D + rho z z^T).
// ------------------------------------------------------------------------------------------
// (1) deflation scan, one thread per node (slaed2): entries with negligible weight and (nearly) equal poles are
// split off.  type[i] = 1: deflated (eigenvalue D[i], eigenvector = column i after the rotations), 0: survivor.
// rot_p[i] >= 0: a Givens rotation of columns (rot_p[i], i) with (rot_c[i], rot_s[i]) was applied when i joined.
__global__ void dc_deflate_kernel(double *__restrict__ D, double *__restrict__ z, const double *__restrict__ rho,
                                  int *__restrict__ type, int *__restrict__ rot_p, double *__restrict__ rot_c,
                                  double *__restrict__ rot_s, int *__restrict__ K, int s, int nodes, double eps) {
  const int nd = blockIdx.x * blockDim.x + threadIdx.x;
  if (nd >= nodes) return;
  double *Dn = D + (long)nd * s, *zn = z + (long)nd * s;
  int *tn = type + (long)nd * s, *rp = rot_p + (long)nd * s;
  double *rc = rot_c + (long)nd * s, *rs = rot_s + (long)nd * s;
  const double r = rho[nd];
  double dmax = 0.0, zmax = 0.0;
  for (int i = 0; i < s; ++i) {
    dmax = fmax(dmax, fabs(Dn[i]));
    zmax = fmax(zmax, fabs(zn[i]));
    rp[i] = -1;
  }
  const double tol = 8.0 * eps * fmax(dmax, zmax);
  int k = 0;
  if (r * zmax <= tol) {  // the halves do not interact
    for (int i = 0; i < s; ++i) tn[i] = 1;
    K[nd] = 0;
    return;
  }
  int pj = -1;
  for (int i = 0; i < s; ++i) {
    if (r * fabs(zn[i]) <= tol) {
      tn[i] = 1;
      continue;
    }
    if (pj < 0) {
      pj = i;
      continue;
    }
    double sv = zn[pj], cv = zn[i];
    const double tau = hypot(cv, sv);
    const double t = Dn[i] - Dn[pj];
    cv /= tau;
    sv = -sv / tau;
    if (fabs(t * cv * sv) <= tol) {  // poles pj and i coincide numerically: rotate the weight of pj into i
      zn[i] = tau;
      zn[pj] = 0.0;
      rp[i] = pj;
      rc[i] = cv;
      rs[i] = sv;
      const double t2 = Dn[pj] * cv * cv + Dn[i] * sv * sv;
      Dn[i] = Dn[pj] * sv * sv + Dn[i] * cv * cv;
      Dn[pj] = t2;
      tn[pj] = 1;
    } else {
      tn[pj] = 0;
      ++k;
    }
    pj = i;
  }
  if (pj >= 0) {
    tn[pj] = 0;
    ++k;
  }
  K[nd] = k;
}

// (2) secular equation 1 + rho sum_i z_i^2 / (d_i - lam) = 0 for the K survivors (dk ascending, strictly):
// root j in (d_j, d_{j+1}) (last: (d_K-1, d_K-1 + rho sum z^2)).  16 lanes per root, bisection in float64 on
// mu = lam - d_org with the origin at the closer pole, so that d_i - lam = (d_i - d_org) - mu keeps full
// relative accuracy (the eigenvector formula needs exactly these differences).
constexpr int DC_LPR = 16;  // lanes per root
__global__ __launch_bounds__(256) void dc_secular_kernel(const double *__restrict__ dk, const double *__restrict__ zk,
                                                         const double *__restrict__ rho, const int *__restrict__ K,
                                                         int *__restrict__ org, double *__restrict__ mu, int s) {
  const int nd = blockIdx.y;
  const int k = K[nd];
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / DC_LPR, sub = threadIdx.x % DC_LPR;
  if (j >= k) return;  // (whole 16-lane groups leave together)
  const double *d = dk + (long)nd * s, *z = zk + (long)nd * s;
  const double r = rho[nd];
  auto reduce = [&](double v) {
#pragma unroll
    for (int off = DC_LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, DC_LPR);
    return v;
  };
  double lo, hi;
  int o;
  if (j + 1 < k) {
    const double gap = d[j + 1] - d[j], mid = 0.5 * gap;
    // f at the midpoint, poles taken relative to d_j
    double part = 0.0;
    for (int i = sub; i < k; i += DC_LPR) part += z[i] * z[i] / ((d[i] - d[j]) - mid);
    const double fm = 1.0 + r * reduce(part);
    if (fm > 0.0) { o = j; lo = 0.0; hi = mid; }
    else { o = j + 1; lo = -mid; hi = 0.0; }
  } else {
    double part = 0.0;
    for (int i = sub; i < k; i += DC_LPR) part += z[i] * z[i];
    o = j; lo = 0.0; hi = r * reduce(part);
  }
  const double dorg = d[o];
  for (int it = 0; it < 60; ++it) {
    const double m = 0.5 * (lo + hi);
    if (m == lo || m == hi) break;
    double part = 0.0;
    for (int i = sub; i < k; i += DC_LPR) part += z[i] * z[i] / ((d[i] - dorg) - m);
    const double f = 1.0 + r * reduce(part);
    if (f > 0.0) hi = m; else lo = m;   // f increases from -inf to +inf across the interval
  }
  if (sub == 0) {
    org[(long)nd * s + j] = o;
    // keep strictly inside the interval: the differences below must not vanish
    double m = 0.5 * (lo + hi);
    if (m == 0.0) m = (o == j) ? hi : lo;
    mu[(long)nd * s + j] = m;
  }
}

// (3) Gu-Eisenstat weights: zh_i^2 = prod_j (lam_j - d_i) / (rho prod_{j != i} (d_j - d_i)), accumulated as a sum
// of logs of ratios in (0, 1] (lam_j paired with d_j for j < i, lam_{j-1} with d_j for j > i, lam_{K-1} alone).
__global__ __launch_bounds__(256) void dc_zhat_kernel(const double *__restrict__ dk, const double *__restrict__ zk,
                                                      const double *__restrict__ rho, const int *__restrict__ K,
                                                      const int *__restrict__ org, const double *__restrict__ mu,
                                                      double *__restrict__ zh, int s) {
  const int nd = blockIdx.y;
  const int k = K[nd];
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) / DC_LPR, sub = threadIdx.x % DC_LPR;
  if (i >= k) return;
  const double *d = dk + (long)nd * s;
  const int *og = org + (long)nd * s;
  const double *m = mu + (long)nd * s;
  const double di = d[i];
  auto lam_minus_di = [&](int j) { return (d[og[j]] - di) + m[j]; };  // lam_j - d_i
  double acc = 0.0;
  for (int j = sub; j < k; j += DC_LPR) {
    if (j < i) acc += log(lam_minus_di(j) / (d[j] - di));
    else if (j > i) acc += log(lam_minus_di(j - 1) / (d[j] - di));
  }
#pragma unroll
  for (int off = DC_LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, DC_LPR);
  if (sub == 0) {
    const double last = lam_minus_di(k - 1);
    const double v = sqrt(exp(acc) * last / rho[nd]);
    zh[(long)nd * s + i] = zk[(long)nd * s + i] < 0.0 ? -v : v;
  }
}

// (4) eigenvectors of the rank-one update in the sorted basis, TRANSPOSED: MT[node][col][row] (row contiguous).
// One block per (root j, node): u_i = zh_i / (d_i - lam_j), normalised; survivor i sits at sorted position
// spos[i].  Rows of deflated entries stay zero here (dc_unit_kernel sets their unit entries).
__global__ __launch_bounds__(256) void dc_build_kernel(const double *__restrict__ dk, const int *__restrict__ K,
                                                       const int *__restrict__ org, const double *__restrict__ mu,
                                                       const double *__restrict__ zh, const int *__restrict__ spos,
                                                       float *__restrict__ MT, int s) {
  __shared__ double s_red[4];
  const int nd = blockIdx.y, j = blockIdx.x;
  const int k = K[nd];
  if (j >= k) return;
  const double *d = dk + (long)nd * s, *zz = zh + (long)nd * s;
  const double dorg = d[org[(long)nd * s + j]], m = mu[(long)nd * s + j];
  float *row = MT + ((long)nd * s + j) * s;
  const int *sp = spos + (long)nd * s;
  double nrm = 0.0;
  for (int i = threadIdx.x; i < k; i += 256) {
    const double u = zz[i] / ((d[i] - dorg) - m);
    nrm += u * u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nrm += __shfl_xor(nrm, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = nrm;
  __syncthreads();
  const double inv = 1.0 / sqrt(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
  for (int i = threadIdx.x; i < k; i += 256) row[sp[i]] = (float)(zz[i] / ((d[i] - dorg) - m) * inv);
}

// (5) the Givens rotations of the deflation, applied in reverse order to the ROWS (pj, i) of M = columns of MT:
// thread per column c of M (row c of MT).  [x_pj ; x_i] <- [c x_pj - s x_i ; s x_pj + c x_i].
__global__ void dc_rotate_kernel(float *__restrict__ MT, const int *__restrict__ rot_p, const double *__restrict__ rot_c,
                                 const double *__restrict__ rot_s, int s) {
  const int nd = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= s) return;
  float *row = MT + ((long)nd * s + c) * s;
  const int *rp = rot_p + (long)nd * s;
  const double *rc = rot_c + (long)nd * s, *rs = rot_s + (long)nd * s;
  for (int i = s - 1; i >= 0; --i) {
    const int pj = rp[i];
    if (pj < 0) continue;
    const float cc = (float)rc[i], ss = (float)rs[i];
    const float xp = row[pj], xi = row[i];
    row[pj] = cc * xp - ss * xi;
    row[i] = ss * xp + cc * xi;
  }
}

// explicit reflector rows for the back-transformation: Vt[i][k] = v_i[k] (unit entry at k = i + 1, zero left of
// it and in the padding; rows >= n - 1 are zero), from LAPACK's 'L' storage in the rows of `work`
__global__ void ormtr_vt_kernel(const float *__restrict__ work, long ldw, float *__restrict__ Vt, long ldv, int n,
                                int rows) {
  const long total = (long)rows * ldv;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int i = (int)(e / ldv), k = (int)(e - (long)i * ldv);
    float v = 0.f;
    if (i < n - 1 && k < n) v = k >= i + 2 ? work[(long)i * ldw + k] : (k == i + 1 ? 1.f : 0.f);
    Vt[e] = v;
  }
}
__global__ void ormtr_tau_kernel(const float *__restrict__ tau, float *__restrict__ tp, int n, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) tp[i] = i < n - 1 ? tau[i] : 0.f;
}

}  // namespace clo
using namespace clo;

// T[p] (nb x nb, row-major, upper triangular) for `np` blocks of nb <= 64 reflectors from their Gram matrices
// G[p] = V_p^T V_p (nb x nb) and scales tau[p][nb] (a zero scale = identity reflector).
extern "C" int clo_larft_f32(const float *G, const float *tau, float *T, int np, int nb, void *stream) {
  CLO_REQUIRE(G && tau && T && np >= 0 && nb >= 1 && nb <= 64, "clo_larft_f32: bad arguments (1 <= nb <= 64)");
  if (np == 0) return CLO_OK;
  hipLaunchKernelGGL(larft_kernel, dim3(np), dim3(64), 0, (hipStream_t)stream, G, tau, T, nb);
  CLO_CHECK_LAUNCH("larft_kernel");
  return CLO_OK;
}

// Eigen-decomposition of `batch` symmetric tridiagonal matrices of order L <= 64 (d[b][L], e[b][L] with the
// sub-diagonal in e[b][0..L-2]): lam[b][L] ascending, Q[b][L][L] eigenvectors in columns.  status (device int,
// not reset here): > 0 if some matrix did not converge.
extern "C" int clo_tql2_batched_f32(const float *d, const float *e, float *lam, float *Q, int L, int batch,
                                    int *status, void *stream) {
  CLO_REQUIRE(d && e && lam && Q && status && L >= 1 && L <= TQL_MAXL && batch >= 0,
              "clo_tql2_batched_f32: bad arguments (1 <= L <= 64)");
  if (batch == 0) return CLO_OK;
  hipLaunchKernelGGL(tql2_kernel, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, d, e, lam, Q, L, status);
  CLO_CHECK_LAUNCH("tql2_kernel");
  return CLO_OK;
}

// ---- divide & conquer merge steps (see the kernels above); all arrays [nodes][s] unless noted, device memory
extern "C" int clo_dc_deflate(double *D, double *z, const double *rho, int *type, int *rot_p, double *rot_c,
                              double *rot_s, int *K, int s, int nodes, double eps, void *stream) {
  CLO_REQUIRE(D && z && rho && type && rot_p && rot_c && rot_s && K && s >= 1 && nodes >= 1, "clo_dc_deflate: bad arguments");
  hipLaunchKernelGGL(dc_deflate_kernel, dim3((unsigned)cdiv(nodes, 64)), dim3(64), 0, (hipStream_t)stream, D, z, rho,
                     type, rot_p, rot_c, rot_s, K, s, nodes, eps);
  CLO_CHECK_LAUNCH("dc_deflate_kernel");
  return CLO_OK;
}
// kmax: upper bound of K over the nodes (host value; <= s)
extern "C" int clo_dc_secular(const double *dk, const double *zk, const double *rho, const int *K, int *org, double *mu,
                              double *zh, int s, int nodes, int kmax, void *stream) {
  CLO_REQUIRE(dk && zk && rho && K && org && mu && zh && s >= 1 && nodes >= 1 && kmax >= 0 && kmax <= s,
              "clo_dc_secular: bad arguments");
  if (kmax == 0) return CLO_OK;
  const dim3 grid((unsigned)cdiv((long)kmax * DC_LPR, 256), (unsigned)nodes);
  hipLaunchKernelGGL(dc_secular_kernel, grid, dim3(256), 0, (hipStream_t)stream, dk, zk, rho, K, org, mu, s);
  CLO_CHECK_LAUNCH("dc_secular_kernel");
  hipLaunchKernelGGL(dc_zhat_kernel, grid, dim3(256), 0, (hipStream_t)stream, dk, zk, rho, K, org, mu, zh, s);
  CLO_CHECK_LAUNCH("dc_zhat_kernel");
  return CLO_OK;
}
// MT [nodes][s][s] (zeroed by the caller): rows 0..K-1 = eigenvectors of the secular roots in sorted coordinates
extern "C" int clo_dc_build(const double *dk, const int *K, const int *org, const double *mu, const double *zh,
                            const int *spos, float *MT, int s, int nodes, int kmax, void *stream) {
  CLO_REQUIRE(dk && K && org && mu && zh && spos && MT && s >= 1 && nodes >= 1 && kmax >= 0 && kmax <= s,
              "clo_dc_build: bad arguments");
  if (kmax == 0) return CLO_OK;
  hipLaunchKernelGGL(dc_build_kernel, dim3((unsigned)kmax, (unsigned)nodes), dim3(256), 0, (hipStream_t)stream, dk, K,
                     org, mu, zh, spos, MT, s);
  CLO_CHECK_LAUNCH("dc_build_kernel");
  return CLO_OK;
}
extern "C" int clo_dc_rotate(float *MT, const int *rot_p, const double *rot_c, const double *rot_s, int s, int nodes,
                             void *stream) {
  CLO_REQUIRE(MT && rot_p && rot_c && rot_s && s >= 1 && nodes >= 1, "clo_dc_rotate: bad arguments");
  hipLaunchKernelGGL(dc_rotate_kernel, dim3((unsigned)cdiv(s, 64), (unsigned)nodes), dim3(64), 0, (hipStream_t)stream,
                     MT, rot_p, rot_c, rot_s, s);
  CLO_CHECK_LAUNCH("dc_rotate_kernel");
  return CLO_OK;
}

// ---- back-transformation  Z <- Z Q^T  (every ROW of Z [m][ldz >= pad4(n)] times Q = H_0 ... H_{n-2}, the
// reflectors clo_sytrd_f32 left in `work` / `tau`): blocks of 256 reflectors as I - V T^T V^T, three GEMMs each.
static inline long ormtr_pad4(long n) { return (n + 3) & ~3L; }
// Blocks of OB_Q x 64 = 256 reflectors (round 4; 64 before: 72 x 3 skinny products of ~50 us at n = 4609).  The T factor of
// a big block is assembled from its panels' 64 x 64 factors with the compact-WY product rule
//   (I - Va Ta Va^T)(I - Vb Tb Vb^T) = I - [Va Vb] [[Ta, -Ta (Va^T Vb) Tb], [0, Tb]] [Va Vb]^T,
// panel by panel: column block j of T_big above its diagonal block = -(T_big[0:64j, 0:64j] G_big[0:64j, j]) T_j.
constexpr int OB_Q = 4, OB_NB = 64 * OB_Q;
namespace clo {
// T_big[b] (256 x 256, zeroed by the caller) <- diagonal blocks T[4 b + q]
__global__ void ormtr_tbig_diag_kernel(const float *__restrict__ T, float *__restrict__ Tbig, int npan) {
  const int pnl = blockIdx.x, b = pnl / OB_Q, q = pnl - b * OB_Q;
  if (pnl >= npan) return;
  for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
    const int r = e >> 6, c = e & 63;
    Tbig[(long)b * OB_NB * OB_NB + (long)(64 * q + r) * OB_NB + 64 * q + c] = T[(long)pnl * 4096 + e];
  }
}
}  // namespace clo
extern "C" long clo_ormtr_ws_floats(int m, int n) {
  if (n < 3 || m < 1) return 64;
  const long ldv = ormtr_pad4(n), nbig = cdiv(n - 1, OB_NB), R = nbig * OB_NB, npan = nbig * OB_Q;
  return R * ldv + 2 * npan * 4096 + R + 2 * nbig * OB_NB * OB_NB + nbig * (OB_NB - 64) * 64 + 2L * m * OB_NB +
         8L * std::max<long>(m, 64) * 64 + 256;
}
extern "C" int clo_ormtr_f32(const float *work, long ldw, const float *tau, float *Z, long ldz, int m, int n,
                             float *ws, long ws_floats, void *stream) {
  CLO_REQUIRE(work && tau && Z && ws && m >= 1 && n >= 1, "clo_ormtr_f32: bad arguments");
  if (n < 3) return CLO_OK;
  const long ldv = ormtr_pad4(n);
  CLO_REQUIRE(ldz >= ldv && ldz % 4 == 0 && ldw >= n && aligned16(Z) && aligned16(ws),
              "clo_ormtr_f32: Z needs a 16-byte aligned base and a leading dimension >= pad4(n), multiple of 4");
  CLO_REQUIRE(ws_floats >= clo_ormtr_ws_floats(m, n), "clo_ormtr_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // reflector rows padded to whole big blocks: the surplus rows are zero reflectors (tau = 0 -> T = 0)
  const int nbig = (int)cdiv(n - 1, OB_NB), R = nbig * OB_NB, npan = nbig * OB_Q;
  constexpr long TB = (long)OB_NB * OB_NB;
  float *Vt = ws, *G = Vt + (long)R * ldv, *T = G + (long)npan * 4096, *tp = T + (long)npan * 4096;
  float *Gbig = tp + R, *Tbig = Gbig + nbig * TB, *Y = Tbig + nbig * TB;
  float *W = Y + (long)nbig * (OB_NB - 64) * 64, *W2 = W + (long)m * OB_NB, *gws = W2 + (long)m * OB_NB;
  const long gws_floats = ws_floats - (gws - ws);
  hipLaunchKernelGGL(ormtr_vt_kernel, dim3((unsigned)std::min<long>(cdiv((long)R * ldv, 256), 4096)), dim3(256), 0, st,
                     work, ldw, Vt, ldv, n, R);
  CLO_CHECK_LAUNCH("ormtr_vt_kernel");
  hipLaunchKernelGGL(ormtr_tau_kernel, dim3((unsigned)cdiv(R, 256)), dim3(256), 0, st, tau, tp, n, R);
  CLO_CHECK_LAUNCH("ormtr_tau_kernel");
  auto gemm = [&](int M, int N, int K, float alpha, const float *A, long sa_m, long sa_k, long sa_b, const float *B,
                  long sb_k, long sb_n, long sb_b, float beta, float *C, long ldc, long sc_b, int batch) {
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
    g.A = A; g.sa_m = sa_m; g.sa_k = sa_k; g.sa_b = sa_b;
    g.B = B; g.sb_k = sb_k; g.sb_n = sb_n; g.sb_b = sb_b;
    g.C = C; g.ldc = ldc; g.sc_b = sc_b;
    return launch_gemm_auto(g, gws, gws_floats, st, batch);
  };
  // Gram matrices: per panel (64 x 64, for the panels' own T factors) and per big block (256 x 256, cross terms)
  int rc = gemm(64, 64, (int)ldv, 1.f, Vt, ldv, 1, 64 * ldv, Vt, 1, ldv, 64 * ldv, 0.f, G, 64, 4096, npan);
  if (rc != CLO_OK) return rc;
  rc = gemm(OB_NB, OB_NB, (int)ldv, 1.f, Vt, ldv, 1, OB_NB * ldv, Vt, 1, ldv, OB_NB * ldv, 0.f, Gbig, OB_NB, TB, nbig);
  if (rc != CLO_OK) return rc;
  hipLaunchKernelGGL(larft_kernel, dim3(npan), dim3(64), 0, st, G, tp, T, 64);
  CLO_CHECK_LAUNCH("larft_kernel");
  rc = check_hip(hipMemsetAsync(Tbig, 0, nbig * TB * sizeof(float), st), "clo_ormtr_f32: T reset");
  if (rc != CLO_OK) return rc;
  hipLaunchKernelGGL(ormtr_tbig_diag_kernel, dim3(npan), dim3(256), 0, st, T, Tbig, npan);
  CLO_CHECK_LAUNCH("ormtr_tbig_diag_kernel");
  for (int j = 1; j < OB_Q; ++j) {   // all big blocks at once
    const int h = 64 * j;
    rc = gemm(h, 64, h, 1.f, Tbig, OB_NB, 1, TB, Gbig + h, OB_NB, 1, TB, 0.f, Y, 64, (long)(OB_NB - 64) * 64, nbig);   // Y = T[0:h,0:h] G[0:h, j]
    if (rc != CLO_OK) return rc;
    rc = gemm(h, 64, 64, -1.f, Y, 64, 1, (long)(OB_NB - 64) * 64, Tbig + (long)h * OB_NB + h, OB_NB, 1, TB, 0.f,
              Tbig + h, OB_NB, TB, nbig);                                                                                // T[0:h, j] = -Y T_j
    if (rc != CLO_OK) return rc;
  }
  for (int b = nbig - 1; b >= 0; --b) {
    const long c0 = (long)OB_NB * b;
    const int w = (int)(ldv - c0);
    const float *Vp = Vt + c0 * ldv + c0;   // rows c0 .., columns c0 ..
    const float *Tb = Tbig + b * TB;
    float *Zs = Z + c0;
    rc = gemm(m, OB_NB, w, 1.f, Zs, ldz, 1, 0, Vp, 1, ldv, 0, 0.f, W, OB_NB, 0, 1);        // W  = Zs Vp^T
    if (rc != CLO_OK) return rc;
    rc = gemm(m, OB_NB, OB_NB, 1.f, W, OB_NB, 1, 0, Tb, 1, OB_NB, 0, 0.f, W2, OB_NB, 0, 1);   // W2 = W T^T
    if (rc != CLO_OK) return rc;
    rc = gemm(m, w, OB_NB, -1.f, W2, OB_NB, 1, 0, Vp, ldv, 1, 0, 1.f, Zs, ldz, 0, 1);      // Zs -= W2 Vp
    if (rc != CLO_OK) return rc;
  }
  return CLO_OK;
}
