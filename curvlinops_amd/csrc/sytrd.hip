// Householder tridiagonalisation of a dense symmetric matrix, one launch per column.
//
// The eigendecompositions of the Kronecker factors (reference kronecker.py:294 / ekfac.py: torch.linalg.eigh,
// i.e. rocSOLVER ssyevd on this platform) spend 85 % of their time in the reduction to tridiagonal form:
// rocSOLVER's latrd runs ~5 small dependent kernels per column (n = 4609: 107 of 126 ms,
// tools/probe_rocsolver_phases.py).  This file is that reduction with ONE kernel per column:
//
//   * the trailing matrix is kept as a full symmetric array, so a block owns complete rows of the
//     matrix-vector product y = A22 v (no cross-block accumulation);
//   * the scalar that finishes column j of W (gamma_j = -tau/2 w^T v) and the norm / panel dot products that
//     start column j+1 are global reductions; instead of a launch each, every block leaves per-block
//     partial sums and the NEXT launch's prologue adds them up (redundantly per block, in a fixed order):
//     the next column is u = u0 - 2 gamma v with u0 computable before gamma is known, and all panel dot
//     products with v' = s (u0 - 2 gamma v) are linear in quantities summed one launch earlier;
//   * the rank-2nb trailing update runs on the MFMA GEMM engine once per 64-column panel.
//
// Storage is LAPACK's (ssytrd, uplo = 'L' of the column-major matrix == the rows of the row-major
// array): on return row j holds the Householder vector of column j in columns j+2.. (unit entry at
// column j+1 implied), D/E the tridiagonal matrix and tau the reflector scales, so rocSOLVER's
// sstedc / sormtr (or any LAPACK-compatible back-transformation) take over from there.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "clo_common.h"
#include "gemm.h"

namespace clo {
namespace {

constexpr int TD_NB = 64;                  // panel width
constexpr int TD_THREADS = 512;               // one block per CU with the full 256-VGPR budget per wave
constexpr int TD_WAVES = TD_THREADS / 64;
constexpr int TD_GMAX = 128;               // blocks per column launch
constexpr int TD_NPART = 4 * TD_NB + 8;    // PWu PVu PWv PVv [64 each], S_wv S_uv S_vv S_wu S_wv2
constexpr int TD_SC = 4 * TD_NB;          // offset of the scalars
constexpr int TD_VEC = 4;                  // float4 groups per thread in the prologue: n <= 8192
constexpr int TD_NMAX = TD_VEC * TD_THREADS * 4 - 8;

// Sum over the wave with DPP row operations; every lane returns the total.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#ifdef CLO_TD_SHFL
  return wave_sum(v);
#endif
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  // every lane of a 16-lane row now holds its row's sum
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

struct TdArgs {
  float *A;
  long lda;
  int n, j, c, rpw;        // matrix order, column, index of the column inside its panel, rows per wave
  int dbg;                 // CLO_TD_DEBUG: phase-skipping bit mask (timing experiments only)
  float *Vp, *Wp;          // panels [n][TD_NB], row-major
  const float *u0;         // column j before the gamma term, indexed by matrix row (c == 0: row j of A)
  float *u0_next;
  const float *vprev;      // Householder vector of column j-1, indexed by matrix row
  float *vcur;
  const float *part_prev;  // [g_prev][TD_NPART]
  int g_prev;
  float *part_cur;
  float *gam;              // [TD_NB]: gamma_k of the panel's finished columns (W_k = w0_k + gamma_k v_k); the panel
                           // in memory keeps w0_k until the panel ends, every reader adds the gamma term
  float *D, *E, *tau;
};

__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// One column of the reduction.  A wave owns `p.rpw` consecutive matrix rows and streams them RPW at a
// time (RPW x UN 1 KB loads in flight per wave, the first stage issued before the prologue).
template <int RPW>
__global__ __launch_bounds__(TD_THREADS) void sytrd_col_kernel(const TdArgs p) {
  extern __shared__ float smem[];
  if (p.dbg & 32) return;
  constexpr int UN = RPW == 1 ? 4 : (RPW <= 4 ? 2 : 1);   // float4 groups per row and pipeline stage (double-buffered)
  const int n = p.n, j = p.j, c = p.c, cp = p.c - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = j + 1;            // first row of the Householder vector
  const int m0 = r0 & ~3;          // 16-byte aligned origin of the vectors kept in LDS
  const int n4 = (n + 3) & ~3;
  const int nq = (n4 - m0) >> 2;   // float4 groups covering [m0, n4)
  float *s_v = smem;               // v      [m - m0]
  float *s_row = s_v + n4;         // row j+1 of A
  float *s_red = s_row + n4;       // [TD_WAVES][TD_NPART]
  float *s_t1 = s_red + TD_WAVES * TD_NPART;  // [64] W^T v
  float *s_t2 = s_t1 + TD_NB;      // [64] V^T v
  float *s_wj1 = s_t2 + TD_NB;     // [64] W[j+1][:]
  float *s_vj1 = s_wj1 + TD_NB;    // [64] V[j+1][:]
  float *s_gam = s_vj1 + TD_NB;    // [64] gamma_k (this launch's for k = c-1)
  float *s_slotA = s_gam + TD_NB;  // [TD_WAVES][4]
  float *s_slotB = s_slotA + 4 * TD_WAVES;   // [TD_WAVES]
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- every load that does not depend on values computed here, issued up front ----
  // (a) this wave's matrix rows and their panel rows (issued after the partial sums are in registers)
  const int wbase = j + 2 + (blockIdx.x * TD_WAVES + wave) * p.rpw;   // first row of this wave
  const int wend = min(n, wbase + p.rpw);
  const float4 *ar[RPW];
  bool rv[RPW];
  float4 cur[RPW][UN];
  float Vik[RPW], Wik[RPW];
  auto issue_rows = [&](int ibase) {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int i = ibase + rr;
      rv[rr] = i < wend;
      ar[rr] = reinterpret_cast<const float4 *>(p.A + (long)(rv[rr] ? i : r0) * p.lda + m0);
#pragma unroll
      for (int t = 0; t < UN; ++t) {
        const int q = lane + 64 * t;
        cur[rr][t] = (rv[rr] && q < nq) ? ar[rr][q] : zero4;
      }
      Vik[rr] = Wik[rr] = 0.f;
      if (lane < c && rv[rr]) {
        Vik[rr] = p.Vp[(long)i * TD_NB + lane];
        Wik[rr] = p.Wp[(long)i * TD_NB + lane];
      }
    }
  };
  float4 u4[TD_VEC], vp4[TD_VEC], row4[TD_VEC];
  const float *rowj1 = p.A + (long)r0 * p.lda;
  // (b) rows j and j+1 of the panels
  float wk0 = 0.f, wk1 = 0.f, vk0 = 0.f, vk1 = 0.f, gk = 0.f;
  if (tid < c) {
    wk0 = p.Wp[(long)j * TD_NB + tid];    // w0 parts; the gamma terms are added once gamma_{c-1} is known
    wk1 = p.Wp[(long)r0 * TD_NB + tid];
    vk0 = p.Vp[(long)j * TD_NB + tid];
    vk1 = p.Vp[(long)r0 * TD_NB + tid];
    if (tid < cp) gk = p.gam[tid];
  }
  const float tau_prev = c > 0 ? p.tau[j - 1] : 0.f;
  const float ajj = p.A[(long)j * p.lda + j];

  // ---- partial sums of the previous launch, added in a fixed order ----
  // thread (gg, q4): float4 group q4 of the blocks gg, gg + NGG, ...: at most TD_PG loads, all in flight at once
  constexpr int NQ4 = TD_NPART / 4;            // 66 float4 groups per block
  constexpr int NGG = TD_THREADS / NQ4;        // 15 block groups
  constexpr int TD_PG = (TD_GMAX + NGG - 1) / NGG;
  if (c > 0) {
    const int q4 = tid % NQ4, gg = tid / NQ4;
    // panel sums are needed for columns < c-1 only
    const bool need = !(p.dbg & 1) && gg < NGG && ((q4 & 15) * 4 < cp || q4 >= TD_SC / 4);
    float4 x[TD_PG];
#pragma unroll
    for (int t = 0; t < TD_PG; ++t) {
      const int g = gg + t * NGG;
      x[t] = (need && g < p.g_prev) ? reinterpret_cast<const float4 *>(p.part_prev + (long)g * TD_NPART)[q4] : zero4;
    }
    float4 sacc = x[0];
#pragma unroll
    for (int t = 1; t < TD_PG; ++t) {
      sacc.x += x[t].x;
      sacc.y += x[t].y;
      sacc.z += x[t].z;
      sacc.w += x[t].w;
    }
    if (gg < NGG) reinterpret_cast<float4 *>(s_red + gg * TD_NPART)[q4] = sacc;
  }
  // everything else the prologue and the row loop start from: in flight under the prologue
  __builtin_amdgcn_sched_barrier(0);   // not before the partial sums have left their registers
  // (c) the vectors of the reflector: u0, previous v, row j+1 of the matrix
#pragma unroll
  for (int t = 0; t < TD_VEC; ++t) {
    const int q = tid + t * TD_THREADS;
    u4[t] = vp4[t] = row4[t] = zero4;
    if (q < nq) {
      const int m = m0 + 4 * q;
      u4[t] = *reinterpret_cast<const float4 *>(p.u0 + m);
      if (c > 0) vp4[t] = *reinterpret_cast<const float4 *>(p.vprev + m);
      row4[t] = *reinterpret_cast<const float4 *>(rowj1 + m);
    }
  }
  issue_rows(wbase);
  __syncthreads();
  float S_wv = 0.f, S_uv = 0.f, S_vv = 0.f, S_wu = 0.f, S_wv2 = 0.f;
  float pwu = 0.f, pvu = 0.f, pwv = 0.f, pvv = 0.f;
  if (c > 0) {
#pragma unroll
    for (int g = 0; g < NGG; ++g) {
      const float4 sc = *reinterpret_cast<const float4 *>(s_red + g * TD_NPART + TD_SC);
      S_wv += sc.x;
      S_uv += sc.y;
      S_vv += sc.z;
      S_wu += sc.w;
      S_wv2 += s_red[g * TD_NPART + TD_SC + 4];
      if (tid < cp) {
        pwu += s_red[g * TD_NPART + tid];
        pvu += s_red[g * TD_NPART + TD_NB + tid];
        pwv += s_red[g * TD_NPART + 2 * TD_NB + tid];
        pvv += s_red[g * TD_NPART + 3 * TD_NB + tid];
      }
    }
  }
  const float gamma = -0.5f * tau_prev * S_wv;

  // ---- column j:  u = u0 - 2 gamma v_prev,  reflector (beta, tau, v) ----
  float sig = 0.f, alpha_loc = 0.f, su0 = 0.f;
#pragma unroll
  for (int t = 0; t < TD_VEC; ++t) {
    const int q = tid + t * TD_THREADS;
    float *u = reinterpret_cast<float *>(&u4[t]);
    const float *vp = reinterpret_cast<const float *>(&vp4[t]);
    if (q < nq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + 4 * q + e;
        const bool in = m >= r0 && m < n;
        const float x = in ? u[e] - 2.f * gamma * vp[e] : 0.f;
        if (in) su0 += u[e] * u[e];
        u[e] = x;
        if (m == r0) alpha_loc = x;
        if (m > r0) sig += x * x;
      }
    }
  }
  {
    const float a = wave_sum_dpp(sig), b = wave_sum_dpp(alpha_loc), d = wave_sum_dpp(su0);
    if (lane == 0) *reinterpret_cast<float4 *>(s_slotA + 4 * wave) = make_float4(a, b, d, 0.f);
  }
  __syncthreads();
  float sigma = 0.f, alpha = 0.f, su0_tot = 0.f;
#pragma unroll
  for (int w = 0; w < TD_WAVES; ++w) {
    const float4 x = *reinterpret_cast<const float4 *>(s_slotA + 4 * w);
    sigma += x.x;
    alpha += x.y;   // one thread holds it, the rest added 0
    su0_tot += x.z;
  }
  // u = u0 - 2 gamma v_prev lost digits to cancellation: the dot products with v derived from last
  // launch's sums would carry an error of eps * |u0| / |u| (they are divided by |u|), which breaks the
  // consistency of the update with the reflector.  Rare (the columns where the rank of a low-rank
  // factor runs out): recompute them from the panel with the actual v (every block, redundantly).
  const bool careful = c > 0 && su0_tot > 16.f * (alpha * alpha + sigma);
  float beta, tau, s;
  if (sigma == 0.f) {
    beta = alpha;
    tau = 0.f;
    s = 0.f;
  } else {
    beta = -copysignf(sqrtf(alpha * alpha + sigma), alpha);
    tau = (beta - alpha) / beta;
    s = 1.f / (alpha - beta);
  }
  float yj1 = 0.f;
#pragma unroll
  for (int t = 0; t < TD_VEC; ++t) {
    const int q = tid + t * TD_THREADS;
    if (q < nq) {
      float4 v, rz;
      float *ve = reinterpret_cast<float *>(&v), *rze = reinterpret_cast<float *>(&rz);
      const float *u = reinterpret_cast<const float *>(&u4[t]);
      const float *rw = reinterpret_cast<const float *>(&row4[t]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + 4 * q + e;
        ve[e] = (m == r0) ? 1.f : s * u[e];   // u is already 0 outside (r0, n)
        rze[e] = (m >= r0 && m < n) ? rw[e] : 0.f;
        yj1 += rze[e] * ve[e];
      }
      reinterpret_cast<float4 *>(s_v)[q] = v;
      reinterpret_cast<float4 *>(s_row)[q] = rz;
    }
  }
  // ---- panel dot products t1 = W^T v, t2 = V^T v from the sums of the previous launch ----
  if (tid < TD_NB) {
    float t1 = 0.f, t2 = 0.f, wj1 = 0.f, vj1 = 0.f, g = 0.f;
    if (tid < c) {
      g = tid == cp ? gamma : gk;
      wk0 += g * vk0;             // finished W[j][k]
      wj1 = wk1 + g * vk1;        // finished W[j+1][k]
      vj1 = vk1;
    }
    if (tid < cp) {
      t1 = wj1 + s * (pwu - 2.f * gamma * pwv);
      t2 = vk1 + s * (pvu - 2.f * gamma * pvv);
    } else if (tid == cp) {
      t1 = wj1 + s * (S_wu - 2.f * gamma * S_wv2 + gamma * S_uv - 2.f * gamma * gamma * S_vv);
      t2 = vk1 + s * (S_uv - 2.f * gamma * S_vv);
    }
    s_t1[tid] = t1;
    s_t2[tid] = t2;
    s_wj1[tid] = wj1;
    s_vj1[tid] = vj1;
    s_gam[tid] = g;
  }
  yj1 = wave_sum_dpp(yj1);
  if (lane == 0) s_slotB[wave] = yj1;
  __syncthreads();   // publishes s_v, s_row, s_t*, s_slotB
  yj1 = 0.f;
#pragma unroll
  for (int w = 0; w < TD_WAVES; w += 4) {
    const float4 x = *reinterpret_cast<const float4 *>(s_slotB + w);
    yj1 += (x.x + x.y) + (x.z + x.w);
  }
  if (careful) {
    // t1 = W^T v, t2 = V^T v over the trailing rows, with the actual v (lane = panel column, a wave takes every eighth
    // row).  Sixteen rows per step with all 32 loads in flight: one row per step was a chain of ~600 dependent L2 round
    // trips, 50 us per column on well-conditioned matrices, where this path is taken for most columns (sytrd of a
    // 4608 x 4608 Wishart matrix: 214 -> ~140 ms).  The ORDER of the additions is the one of the one-row loop: the
    // results are bit-identical to it (a float4 / four-row-group layout was another 20 % faster but moved the
    // reconstruction error of the rank-deficient 4609 test matrix from 0.8e-4 to 1.3e-4 |A|max -- both are one
    // rounding error of the top eigenvalue 1152, but the bound of the test is 1e-4).
    float a1 = 0.f, a2 = 0.f;
    if (lane < c) {
      constexpr int CU = 16;
      const float gl_ = s_gam[lane];
      for (int i0_ = r0 + wave; i0_ < n; i0_ += TD_WAVES * CU) {
        float Vr[CU], Wr[CU];
#pragma unroll
        for (int t = 0; t < CU; ++t) {
          const int i = min(i0_ + t * TD_WAVES, n - 1);
          Vr[t] = p.Vp[(long)i * TD_NB + lane];
          Wr[t] = p.Wp[(long)i * TD_NB + lane];
        }
#pragma unroll
        for (int t = 0; t < CU; ++t) {
          const int i = i0_ + t * TD_WAVES;
          if (i < n) {
            const float vi = s_v[i - m0];
            const float W = Wr[t] + gl_ * Vr[t];   // finished entries W = w0 + gamma_k v
            a1 += W * vi;
            a2 += Vr[t] * vi;
          }
        }
      }
    }
    s_red[wave * TD_NPART + lane] = a1;
    s_red[wave * TD_NPART + TD_NB + lane] = a2;
    __syncthreads();
    if (tid < TD_NB) {
      float x1 = 0.f, x2 = 0.f;
#pragma unroll
      for (int w = 0; w < TD_WAVES; ++w) {
        x1 += s_red[w * TD_NPART + tid];
        x2 += s_red[w * TD_NPART + TD_NB + tid];
      }
      s_t1[tid] = tid < c ? x1 : 0.f;
      s_t2[tid] = tid < c ? x2 : 0.f;
    }
    __syncthreads();
  }
  const float t1l = s_t1[lane], t2l = s_t2[lane], wj1l = s_wj1[lane], vj1l = s_vj1[lane], gl = s_gam[lane];
  // unfinished w at row j+1 (every wave computes it)
  const float w0j1 = tau * (yj1 - wave_sum_dpp(lane < c ? vj1l * t1l + wj1l * t2l : 0.f));
  if (blockIdx.x == 0 && wave == 0) {
    // d_j = A[j][j] - 2 sum_k V[j][k] W[j][k] with the finished W[j][c-1] = w0 + gamma (V[j][c-1] = 1)
    const float x = wave_sum_dpp(lane < c ? vk0 * wk0 : 0.f);
    if (lane == 0) {
      p.D[j] = ajj - 2.f * x;
      p.E[j] = beta;
      p.tau[j] = tau;
      p.Vp[(long)r0 * TD_NB + c] = 1.f;
      p.Wp[(long)r0 * TD_NB + c] = w0j1;
      p.vcur[r0] = 1.f;
      if (c > 0) p.gam[cp] = gamma;   // read by the launches after this one
    }
  }

  // ---- rows i >= j+2:  y_i = A[i][:] v,  w0_i,  next column's u0_i,  partial sums ----
  float accPW = 0.f, accPV = 0.f, accPWv = 0.f, accPVv = 0.f;       // per lane k < c
  float a_wv = 0.f, a_uv = 0.f, a_vv = 0.f, a_wu = 0.f, a_wv2 = 0.f; // wave-uniform
  const float4 *sv = reinterpret_cast<const float4 *>(s_v);
  if (p.dbg & 16) return;
  for (int ibase = wbase; ibase < ((p.dbg & 2) ? wbase : wend); ibase += RPW) {
    float acc[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) acc[rr] = 0.f;
    // three pipeline stages: the loads of step s + 2 are issued while step s is consumed
    float4 nxt[RPW][UN];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
      for (int t = 0; t < UN; ++t) {
        const int q = 64 * UN + lane + 64 * t;
        nxt[rr][t] = (rv[rr] && q < nq) ? ar[rr][q] : zero4;
      }
    for (int q0 = 0; q0 < ((p.dbg & 4) ? 1 : nq); q0 += 64 * UN) {
      float4 nx2[RPW][UN];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int t = 0; t < UN; ++t) {
          const int q = q0 + 2 * 64 * UN + lane + 64 * t;
          nx2[rr][t] = (rv[rr] && q < nq) ? ar[rr][q] : zero4;
        }
#pragma unroll
      for (int t = 0; t < UN; ++t) {
        const int q = q0 + lane + 64 * t;
        const float4 v = q < nq ? sv[q] : zero4;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) acc[rr] += dot4(cur[rr][t], v);
      }
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int t = 0; t < UN; ++t) {
          cur[rr][t] = nxt[rr][t];
          nxt[rr][t] = nx2[rr][t];
        }
    }
    float y[RPW], pw[RPW], bs[RPW], Vk[RPW], Wk[RPW];
    bool ok[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      Wik[rr] += gl * Vik[rr];   // finished entries W = w0 + gamma_k v (the panel keeps w0)
      Vk[rr] = Vik[rr];
      Wk[rr] = Wik[rr];
      ok[rr] = rv[rr];
      y[rr] = wave_sum_dpp(acc[rr]);
      pw[rr] = wave_sum_dpp(Vik[rr] * t1l + Wik[rr] * t2l);
      bs[rr] = wave_sum_dpp(Vik[rr] * wj1l + Wik[rr] * vj1l);
    }
    if (ibase + RPW < wend) issue_rows(ibase + RPW);   // next group's first stage, under this group's tail
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      if (!ok[rr]) continue;
      const int i = ibase + rr;
      const float vi = s_v[i - m0];
      const float w0 = tau * (y[rr] - pw[rr]);
      const float un = s_row[i - m0] - bs[rr] - vi * w0j1 - w0;
      if (lane == 0) {
        p.Wp[(long)i * TD_NB + c] = w0;
        p.Vp[(long)i * TD_NB + c] = vi;
        p.vcur[i] = vi;
        p.u0_next[i] = un;
      }
      a_wv += w0 * vi;
      if (i >= j + 3) {
        accPW += Wk[rr] * un;
        accPV += Vk[rr] * un;
        accPWv += Wk[rr] * vi;
        accPVv += Vk[rr] * vi;
        a_uv += un * vi;
        a_vv += vi * vi;
        a_wu += w0 * un;
        a_wv2 += w0 * vi;
      }
    }
  }
  if (blockIdx.x == 0 && wave == 0) a_wv += w0j1;   // row j+1: v = 1
  float *mine = s_red + wave * TD_NPART;
  mine[lane] = accPW;
  mine[TD_NB + lane] = accPV;
  mine[2 * TD_NB + lane] = accPWv;
  mine[3 * TD_NB + lane] = accPVv;
  if (lane == 0) {
    mine[TD_SC + 0] = a_wv;
    mine[TD_SC + 1] = a_uv;
    mine[TD_SC + 2] = a_vv;
    mine[TD_SC + 3] = a_wu;
    mine[TD_SC + 4] = a_wv2;
    mine[TD_SC + 5] = mine[TD_SC + 6] = mine[TD_SC + 7] = 0.f;
  }
  __syncthreads();
  if (tid < TD_NPART) {
    float sacc = 0.f;
#pragma unroll
    for (int w = 0; w < TD_WAVES; ++w) sacc += s_red[w * TD_NPART + tid];
    p.part_cur[(long)blockIdx.x * TD_NPART + tid] = sacc;
  }
}

// End of a panel whose last column is jl (index cl inside the panel, panel origin i0): finish W for the rows of
// the trailing update (W_k = w0_k + gamma_k v_k, rows >= jl+1) and move the Householder vectors into the rows
// of A.
__global__ __launch_bounds__(256) void sytrd_panel_end_kernel(float *A, long lda, int n, int i0, int ncol,
                                                              const float *Vp, float *Wp,
                                                              const float *part, int g, const float *tau,
                                                              const float *gam) {
  __shared__ float s_part[256];
  __shared__ float s_g[TD_NB];
  const int jl = i0 + ncol - 1, cl = ncol - 1;
  float sacc = 0.f;
  for (int b = threadIdx.x; b < g; b += 256) sacc += part[(long)b * TD_NPART + TD_SC];
  s_part[threadIdx.x] = sacc;
  if ((int)threadIdx.x < TD_NB) s_g[threadIdx.x] = (int)threadIdx.x < cl ? gam[threadIdx.x] : 0.f;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_part[threadIdx.x] += s_part[threadIdx.x + off];
    __syncthreads();
  }
  const float gamma = -0.5f * tau[jl] * s_part[0];
  const int i = blockIdx.x * 256 + threadIdx.x;   // matrix row
  if (i < n && i > i0) {
    if (i >= jl + 1)
      for (int k = 0; k < ncol; ++k)
        Wp[(long)i * TD_NB + k] += (k == cl ? gamma : s_g[k]) * Vp[(long)i * TD_NB + k];
    // reflector k of the panel lives in rows >= i0+k+2 of column (matrix row) i0+k
    const int kmax = min(ncol, i - i0 - 1);
    for (int k = 0; k < kmax; ++k) A[(long)(i0 + k) * lda + i] = Vp[(long)i * TD_NB + k];
  }
}

// The order-2 block left after the last reflector.
__global__ void sytrd_tail_kernel(const float *A, long lda, int n, float *D, float *E, float *tau) {
  if (threadIdx.x == 0) {
    D[n - 2] = A[(long)(n - 2) * lda + (n - 2)];
    D[n - 1] = A[(long)(n - 1) * lda + (n - 1)];
    E[n - 2] = A[(long)(n - 1) * lda + (n - 2)];
    tau[n - 2] = 0.f;
  }
}

long td_ws_floats(int n) {
  const long n4 = (n + 3) & ~3L;
  return 2L * n * TD_NB + 4 * n4 + 2L * TD_GMAX * TD_NPART + 2L * TD_NB + 64;
}

}  // namespace
}  // namespace clo

using namespace clo;

extern "C" long clo_sytrd_ws_bytes(int n) { return n > 0 ? td_ws_floats(n) * 4 : 0; }

extern "C" int clo_sytrd_f32(float *A, long lda, int n, float *D, float *E, float *tau, float *ws,
                             long ws_bytes, void *stream) {
  CLO_REQUIRE(n >= 3 && n <= TD_NMAX, "clo_sytrd_f32: order %d outside [3, %d]", n, TD_NMAX);
  CLO_REQUIRE(A && D && E && tau && ws, "clo_sytrd_f32: null operand");
  CLO_REQUIRE(lda >= ((n + 3) & ~3) && lda % 4 == 0 && aligned16(A) && aligned16(ws),
              "clo_sytrd_f32: rows must be 16-byte aligned and zero-padded to a multiple of 4 columns (lda %ld)", lda);
  CLO_REQUIRE(ws_bytes >= clo_sytrd_ws_bytes(n), "clo_sytrd_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const long n4 = (n + 3) & ~3L;
  float *Vp = ws, *Wp = Vp + (long)n * TD_NB;
  float *u0[2] = {Wp + (long)n * TD_NB, Wp + (long)n * TD_NB + n4};
  float *vv[2] = {u0[1] + n4, u0[1] + 2 * n4};
  float *part[2] = {vv[1] + n4, vv[1] + n4 + (long)TD_GMAX * TD_NPART};
  float *gam = part[1] + (long)TD_GMAX * TD_NPART;   // [TD_NB]

  const size_t lds = (2 * n4 + TD_WAVES * TD_NPART + 5 * TD_NB + 5 * TD_WAVES + 16) * sizeof(float);
  // several host threads may run reductions at once (linalg_native.eigh_many): the attribute must be in
  // place for every instantiation before any of them launches with the larger size
  static std::mutex lds_mutex;
  static size_t lds_set_dev[64] = {0};   // per device: function attributes are per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  {
  std::lock_guard<std::mutex> lds_lock(lds_mutex);
  size_t &lds_set = lds_set_dev[dev];
  if (lds > lds_set) {
    const void *fns[8] = {reinterpret_cast<const void *>(sytrd_col_kernel<1>),
                          reinterpret_cast<const void *>(sytrd_col_kernel<2>),
                          reinterpret_cast<const void *>(sytrd_col_kernel<3>),
                          reinterpret_cast<const void *>(sytrd_col_kernel<4>),
                          reinterpret_cast<const void *>(sytrd_col_kernel<5>),
                          reinterpret_cast<const void *>(sytrd_col_kernel<6>),
                          reinterpret_cast<const void *>(sytrd_col_kernel<7>),
                          reinterpret_cast<const void *>(sytrd_col_kernel<8>)};
    for (const void *fn : fns) {
      int rc = check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                         "clo_sytrd_f32: LDS attribute");
      if (rc != CLO_OK) return rc;
    }
    lds_set = lds;
  }
  }
  static const int dbg = getenv("CLO_TD_DEBUG") ? atoi(getenv("CLO_TD_DEBUG")) : 0;
  int g_prev = 1, flip = 0;
  for (int i0 = 0; i0 < n - 2; i0 += TD_NB) {
    const int ncol = std::min(TD_NB, n - 2 - i0);
    for (int c = 0; c < ncol; ++c) {
      const int j = i0 + c;
      const int nd = n - j - 2;
      const int g = (int)std::max<long>(1, std::min<long>(TD_GMAX, cdiv(nd, TD_WAVES)));
      const int rpw = (int)cdiv(nd, (long)g * TD_WAVES);
      TdArgs a;
      a.A = A; a.lda = lda; a.n = n; a.j = j; a.c = c;
      a.Vp = Vp; a.Wp = Wp;
      a.u0 = c == 0 ? A + (long)j * lda : u0[flip];
      a.u0_next = u0[flip ^ 1];
      a.vprev = vv[flip];
      a.vcur = vv[flip ^ 1];
      a.part_prev = part[flip]; a.g_prev = g_prev;
      a.part_cur = part[flip ^ 1];
      a.gam = gam;
      a.D = D; a.E = E; a.tau = tau;
      a.rpw = rpw;
      a.dbg = dbg;
      switch (rpw) {   // all rows of a wave in one pass
#define CLO_TD_CASE(R) \
  case R: hipLaunchKernelGGL(sytrd_col_kernel<R>, dim3(g), dim3(TD_THREADS), lds, st, a); break;
        CLO_TD_CASE(1) CLO_TD_CASE(2) CLO_TD_CASE(3) CLO_TD_CASE(4)
        CLO_TD_CASE(5) CLO_TD_CASE(6) CLO_TD_CASE(7)
#undef CLO_TD_CASE
        default: hipLaunchKernelGGL(sytrd_col_kernel<8>, dim3(g), dim3(TD_THREADS), lds, st, a); break;
      }
      g_prev = g;
      flip ^= 1;
    }
    CLO_CHECK_LAUNCH("sytrd_col_kernel");
    hipLaunchKernelGGL(sytrd_panel_end_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, A, lda, n, i0,
                       ncol, Vp, Wp, part[flip], g_prev, tau, gam);
    CLO_CHECK_LAUNCH("sytrd_panel_end_kernel");
    // trailing update A[t:, t:] -= V W^T + W V^T on the MFMA GEMM engine (full square: the column
    // kernel reads complete rows)
    const int t = i0 + ncol, m = n - t;
    const float *Vt = Vp + (long)t * TD_NB, *Wt = Wp + (long)t * TD_NB;
    float *C = A + (long)t * lda + t;
    int rc;
    if (ncol % 32 == 0) {
      // ONE symmetric product over the concatenated panels, C -= [V | W] [W | V]^T (second K segment of the GEMM engine),
      // upper block triangle computed and mirrored: C[i][j] and C[j][i] receive bit-identical updates (two full-square
      // products added the two terms in opposite orders and let the trailing matrix drift from symmetry), half the flops,
      // one pass over C instead of two.
      GemmArgs g{};
      g.M = m; g.N = m; g.K = 2 * ncol; g.K1 = ncol;
      g.alpha = -1.f; g.beta = 1.f;
      g.A = Vt; g.sa_m = TD_NB; g.sa_k = 1;
      g.B = Wt; g.sb_k = 1; g.sb_n = TD_NB;
      g.A2 = Wt; g.B2 = Vt;
      g.C = C; g.ldc = lda;
      g.splitk = 1; g.sym = 1;
      rc = launch_gemm(g, 1, st);
      if (rc != CLO_OK) return rc;
    } else {   // (the last, shorter panel)
      rc = clo_gemm_f32(m, m, ncol, -1.f, Vt, TD_NB, 1, 0, Wt, 1, TD_NB, 0, 1.f, C, lda, 0, 1, 1, nullptr, st);
      if (rc != CLO_OK) return rc;
      rc = clo_gemm_f32(m, m, ncol, -1.f, Wt, TD_NB, 1, 0, Vt, 1, TD_NB, 0, 1.f, C, lda, 0, 1, 1, nullptr, st);
      if (rc != CLO_OK) return rc;
    }
  }
  hipLaunchKernelGGL(sytrd_tail_kernel, dim3(1), dim3(64), 0, st, A, lda, n, D, E, tau);
  CLO_CHECK_LAUNCH("sytrd_tail_kernel");
  return CLO_OK;
}
