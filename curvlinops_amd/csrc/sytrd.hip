// Householder tridiagonalisation of a dense symmetric matrix: ONE persistent launch per 64-column panel (round 4; rounds
// 2-3 ran one launch per column).
//
// The eigendecompositions of the Kronecker factors (reference kronecker.py:294 / ekfac.py: torch.linalg.eigh, i.e.
// rocSOLVER ssyevd on this platform) spend 85 % of their time in the reduction to tridiagonal form: rocSOLVER's latrd runs
// ~5 small dependent kernels per column (n = 4609: 107 of 126 ms, tools/probe_rocsolver_phases.py).  This file is that
// reduction (LAPACK latrd algebra, deferred reductions):
//
//   * the trailing matrix is kept as a full symmetric array, so a workgroup owns complete rows of the matrix-vector
//     product y = A22 v (no cross-workgroup accumulation of y);
//   * the scalar that finishes column j of W (gamma_j = -tau/2 w^T v) and the norm / panel dot products that start column
//     j+1 are global reductions taken ONE synchronisation late: the next column is u = u0 - 2 gamma v with u0 computable
//     before gamma is known, and all panel dot products with v' = s (u0 - 2 gamma v) are linear in quantities summed one
//     column earlier -- so a column needs exactly one grid-wide hand-off (see sytrd_panel_kernel below);
//   * the rank-2nb trailing update runs on the MFMA GEMM engine once per 64-column panel, between two panel launches.
//
// Storage is LAPACK's (ssytrd, uplo = 'L' of the column-major matrix == the rows of the row-major array): on return row j
// holds the Householder vector of column j in columns j+2.. (unit entry at column j+1 implied), D/E the tridiagonal
// matrix and tau the reflector scales, so any LAPACK-compatible divide & conquer / back-transformation can take over
// (csrc/eigh.hip does; tools/_rocsolver.py drives rocSOLVER's for comparisons).
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "clo_common.h"
#include "gemm.h"
#include "persist_gate.h"

namespace clo {
namespace {

constexpr int TD_NB = 64;                  // panel width
constexpr int TD_THREADS = 512;               // one block per CU with the full 256-VGPR budget per wave
constexpr int TD_WAVES = TD_THREADS / 64;
constexpr int TD_GMAX = 256;               // workgroups of a panel launch (one per CU)
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

__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// ------------------------------------------------------------------------------------------------------------------
// Round 4: ONE persistent launch per 64-column panel (the column kernel of rounds 2-3 paid ~3.5 us of dependent dispatch plus a
// cold prologue per column: 25 us x 4609 columns = 117 of the 144 ms of a 4609 x 4609 eigh).  The G workgroups of the
// launch own the same matrix rows for the whole panel and walk its columns in lock step; between two columns they
// exchange what the next reflector needs -- the per-block partial sums, the next column u0, the current reflector v
// and one more column of the panels V / W -- through write-through (sc1) stores, a two-level counter barrier
// (16-workgroup groups, leaders, one top counter) and sc1 loads, exactly the hand-off of csrc/mlp_mega.hip.  The
// algebra and the storage are those of the column kernel; the matrix itself is only read (the
// rank-128 trailing update runs between two panel launches on the GEMM engine), so its rows keep the plain loads.
// The per-workgroup partial sums travel in two levels: the 16 workgroups of a group ADD theirs into the group's record
// with float atomics (0.37 us per round for all 256 workgroups incl. the acknowledgements, tools/ubench/atomic_probe.hip;
// the first version had the group's leader read, add and re-publish 16 private records between the two counters: two
// more fabric round trips per column, 10.9 -> 8.4 us per column at n = 577), and a column's prologue adds the <= 16 group
// records in a fixed order.  Three record buffers rotate (read c % 3, add into (c + 1) % 3, leaders clear (c + 2) % 3).
// The additions inside a group happen in arrival order: two runs agree to rounding, not bit for bit -- and since the map
// A -> T is ill-conditioned for rank-deficient A, T itself may differ visibly past the numerical rank while its
// spectrum and the similarity Q^T A Q = T hold to eps (tests compare spectra).
// Every spin is bounded and traps.  All G <= TD_GMAX (256) workgroups must be resident at once (one per CU: the kernel
// uses the full register budget of eight waves).
// ------------------------------------------------------------------------------------------------------------------
struct TpArgs {
  const float *A;
  long lda;
  int n, i0, ncol, rpb, rpw, G;   // rpb rows per workgroup, rpw of them per wave
  float *ws;                 // exchange workspace: V panel at offset 0
  long ws_floats;
  long o_W, o_u0[2], o_vv[2], o_gpart[3], o_gam;   // float offsets inside ws
  float *D, *E, *tau;
  unsigned *cnt;             // [TD_GMAX / 16 + 1] counters (one per 128-byte line), zero at launch, then {err}
  unsigned *fault;           // host-pinned fault word of the device (or nullptr)
  unsigned spin_limit;
};
constexpr int TP_GROUP = 16;
// group counters [16], release lines [16] (the top level of the barrier, one per group: a leader adds to all of them with
// one instruction, a workgroup polls its own group's -- 16 pollers per line, through the scalar path), error word: one
// 128-byte line each.  (One top line polled by up to 256 workgroups was the slowest hop of the barrier, and through the
// scalar path it is slower still: profiles/r05_c2_scalar_seam.txt.)
constexpr int TP_CNT_WORDS = 32 * (2 * (TD_GMAX / TP_GROUP) + 2);
constexpr unsigned TP_SPIN = 1u << 22;

// The waits are bounded (scalar_wait, clo_common.h); on a timeout (or when another workgroup of the launch has timed out)
// a wait simply ends: the reduction then finishes on garbage, which the eigensolver's verification rejects (float64
// retry) -- clo_common.h, "asynchronous faults".

template <int RPW>
__global__ __launch_bounds__(TD_THREADS) void sytrd_panel_kernel(const TpArgs p) {
  extern __shared__ float smem[];
  constexpr int UN = RPW == 1 ? 4 : (RPW <= 4 ? 2 : 1);   // float4 groups per row and pipeline stage
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int n = p.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n4 = (n + 3) & ~3;
  float *s_v = smem;               // v      [m - m0]
  float *s_row = s_v + n4;         // row j+1 of A
  float *s_red = s_row + n4;       // [TD_WAVES][TD_NPART]
  float *s_t1 = s_red + TD_WAVES * TD_NPART;  // [64] W^T v
  float *s_t2 = s_t1 + TD_NB;      // [64] V^T v
  float *s_wj1 = s_t2 + TD_NB;     // [64] W[j+1][:]
  float *s_vj1 = s_wj1 + TD_NB;    // [64] V[j+1][:]
  float *s_gam = s_vj1 + TD_NB;    // [64] gamma_k (this column's for k = c-1)
  float *s_slotA = s_gam + TD_NB;  // [TD_WAVES][4]
  float *s_slotB = s_slotA + 4 * TD_WAVES;   // [TD_WAVES]
  typedef float v4 __attribute__((ext_vector_type(4)));   // register-resident 4-vectors (component access by constant index)
  const v4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr unsigned OOB = 0xfffffff0u;   // a buffer offset beyond the range: the load returns zeros, no branch, no memory access

  // exchanged data lives in ONE buffer: sc1 loads / stores by float offset
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (int)(4 * p.ws_floats), 0x00020000);
  // every exchange load is UNCONDITIONAL (a predicated load makes hipcc branch and wait behind it): ok == false reads OOB
  auto L1 = [&](bool ok, long off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (unsigned)(off * 4) : OOB, 0, 16)); };
  auto L4 = [&](bool ok, long off) { return __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(off * 4) : OOB, 0, 16)); };
  auto dotv = [](const v4 a, const v4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; };
  auto S1 = [&](long off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (unsigned)(off * 4), 0, 16); };
  auto S4 = [&](long off, v4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)(off * 4), 0, 16); };
  const long oV = 0, oW = p.o_W;

  const int G = p.G, ngroups = (G + TP_GROUP - 1) / TP_GROUP;
  const int grp = blockIdx.x / TP_GROUP, gsize = min(TP_GROUP, G - grp * TP_GROUP);
  unsigned *c_grp = p.cnt + 32 * grp, *c_rel0 = p.cnt + 32 * (TD_GMAX / TP_GROUP), *c_rel = c_rel0 + 32 * grp;
  unsigned *c_err = p.cnt + 32 * (2 * (TD_GMAX / TP_GROUP));

  // this wave's rows, fixed for the panel: the workgroup owns rpb consecutive rows, its waves rpw of them each (the rows
  // are dealt per WORKGROUP: dealing rpw rows per wave over the whole grid left a quarter of the CUs without rows at
  // n = 4609 -- 24 rows each for 192 workgroups -- and the row pass is bound by what one CU's memory pipe ingests)
  const int bbase = p.i0 + 2 + blockIdx.x * p.rpb;
  const int wbase = bbase + wave * p.rpw;
  const int wend = min(min(n, bbase + p.rpb), wbase + p.rpw);

#ifdef CLO_TD_TIMING
  // phase stamps of workgroup 0 (and of the last workgroup), summed over the columns of the panel at i0 == 0
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define TD_STAMP(i) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tlast; tlast = now_; } } while (0)
  if (tid == 0) tlast = wall_clock64();
#else
#define TD_STAMP(i) do { } while (0)
#endif
  for (int c = 0; c < p.ncol; ++c) {
    const int j = p.i0 + c, cp = c - 1;
    const int in = c & 1, out = in ^ 1;
    const int r0 = j + 1;            // first row of the Householder vector
    const int m0 = r0 & ~3;          // 16-byte aligned origin of the vectors kept in LDS
    const int nq = (n4 - m0) >> 2;   // float4 groups covering [m0, n4)

    // ---- every load that does not depend on values computed here, issued up front ----
    int ri[RPW];        // the wave's rows of this pass (row r0 stands in for the inactive ones)
    bool rv[RPW];
    v4 cur[RPW][UN];
    float Vik[RPW], Wik[RPW];
    // group q of row rr: plain global load from a clamped address, zeroed by a select outside the row / the range
    auto ldA = [&](int rr, int q) {
      const v4 x = *reinterpret_cast<const v4 *>(p.A + (long)ri[rr] * p.lda + m0 + 4 * min(q, nq - 1));
      return (rv[rr] && q < nq) ? x : zero4;
    };
    auto issue_rows = [&](int ibase) {
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int i = ibase + rr;
        rv[rr] = i < wend && i >= j + 2;
        ri[rr] = rv[rr] ? i : r0;
#pragma unroll
        for (int t = 0; t < UN; ++t) cur[rr][t] = ldA(rr, lane + 64 * t);
        Vik[rr] = L1(lane < c && rv[rr], oV + (long)i * TD_NB + lane);
        Wik[rr] = L1(lane < c && rv[rr], oW + (long)i * TD_NB + lane);
      }
    };
    v4 u4[TD_VEC], vp4[TD_VEC], row4[TD_VEC];
    const float *rowj1 = p.A + (long)r0 * p.lda;
    // w0 parts; the gamma terms are added once gamma_{c-1} is known
    float wk0 = L1(tid < c, oW + (long)j * TD_NB + tid), wk1 = L1(tid < c, oW + (long)r0 * TD_NB + tid);
    float vk0 = L1(tid < c, oV + (long)j * TD_NB + tid), vk1 = L1(tid < c, oV + (long)r0 * TD_NB + tid);
    const float gk = L1(tid < cp, p.o_gam + tid);
    const float tau_prev = c > 0 ? __hip_atomic_load(p.tau + j - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    const float ajj = p.A[(long)j * p.lda + j];

    // ---- partial sums of the previous column: the <= 16 group records, added in a fixed order ----
    constexpr int NQ4 = TD_NPART / 4;            // 66 float4 groups per record
    constexpr int NGRP = TD_GMAX / TP_GROUP;     // 16 groups at most
    if (c > 0 && tid < NQ4) {
      v4 x[NGRP];   // all loads of a thread in flight at once: one fabric round trip
#pragma unroll
      for (int g = 0; g < NGRP; ++g) x[g] = L4(g < ngroups, p.o_gpart[c % 3] + (long)g * TD_NPART + 4 * tid);
      v4 sacc = x[0];
#pragma unroll
      for (int g = 1; g < NGRP; ++g) sacc += x[g];
      reinterpret_cast<v4 *>(s_red)[tid] = sacc;
    }
    __builtin_amdgcn_sched_barrier(0);   // not before the partial sums have left their registers
#pragma unroll
    for (int t = 0; t < TD_VEC; ++t) {
      const int q = tid + t * TD_THREADS;
      const bool okq = q < nq;
      const int m = m0 + 4 * min(q, nq - 1);
      // c == 0: the column is row j of the matrix itself; later columns: u0 of the exchange (both unconditional)
      const v4 ua = *reinterpret_cast<const v4 *>(p.A + (long)j * p.lda + m);
      const v4 ub = L4(okq && c > 0, p.o_u0[in] + m);
      u4[t] = okq ? (c == 0 ? ua : ub) : zero4;
      vp4[t] = L4(okq && c > 0, p.o_vv[in] + m);
      const v4 rw = *reinterpret_cast<const v4 *>(rowj1 + m);
      row4[t] = okq ? rw : zero4;
    }
    issue_rows(wbase);
    __syncthreads();
    TD_STAMP(0);   // exchange loads landed (partial sums in LDS)
    float S_wv = 0.f, S_uv = 0.f, S_vv = 0.f, S_wu = 0.f, S_wv2 = 0.f;
    float pwu = 0.f, pvu = 0.f, pwv = 0.f, pvv = 0.f;
    if (c > 0) {
      const v4 sc = *reinterpret_cast<const v4 *>(s_red + TD_SC);
      S_wv = sc[0];
      S_uv = sc[1];
      S_vv = sc[2];
      S_wu = sc[3];
      S_wv2 = s_red[TD_SC + 4];
      if (tid < cp) {
        pwu = s_red[tid];
        pvu = s_red[TD_NB + tid];
        pwv = s_red[2 * TD_NB + tid];
        pvv = s_red[3 * TD_NB + tid];
      }
    }
    const float gamma = -0.5f * tau_prev * S_wv;

    // ---- column j:  u = u0 - 2 gamma v_prev,  reflector (beta, tau, v) ----
    float sig = 0.f, alpha_loc = 0.f, su0 = 0.f;
#pragma unroll
    for (int t = 0; t < TD_VEC; ++t) {
      const int q = tid + t * TD_THREADS;
      if (q < nq) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + 4 * q + e;
          const bool inr = m >= r0 && m < n;
          const float x = inr ? u4[t][e] - 2.f * gamma * vp4[t][e] : 0.f;
          if (inr) su0 += u4[t][e] * u4[t][e];
          u4[t][e] = x;
          if (m == r0) alpha_loc = x;
          if (m > r0) sig += x * x;
        }
      }
    }
    {
      const float a = wave_sum_dpp(sig), b = wave_sum_dpp(alpha_loc), d = wave_sum_dpp(su0);
      if (lane == 0) *reinterpret_cast<v4 *>(s_slotA + 4 * wave) = v4{a, b, d, 0.f};
    }
    __syncthreads();
    float sigma = 0.f, alpha = 0.f, su0_tot = 0.f;
#pragma unroll
    for (int w = 0; w < TD_WAVES; ++w) {
      const v4 x = *reinterpret_cast<const v4 *>(s_slotA + 4 * w);
      sigma += x[0];
      alpha += x[1];   // one thread holds it, the rest added 0
      su0_tot += x[2];
    }
    // cancellation guard: the deferred formulas below take W^T v', V^T v' from sums over u0 and v_prev taken one column
    // earlier; when u = u0 - 2 gamma v_prev cancels (|u0|^2 >> |u|^2: rank-deficient factors once the numerical rank is
    // exhausted) those sums carry the rounding error of the LARGE terms, so the products are then formed from the actual v
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
        v4 v, rz;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + 4 * q + e;
          v[e] = (m == r0) ? 1.f : s * u4[t][e];   // u is already 0 outside (r0, n)
          rz[e] = (m >= r0 && m < n) ? row4[t][e] : 0.f;
          yj1 += rz[e] * v[e];
        }
        reinterpret_cast<v4 *>(s_v)[q] = v;
        reinterpret_cast<v4 *>(s_row)[q] = rz;
      }
    }
    // ---- panel dot products t1 = W^T v, t2 = V^T v from the sums of the previous column ----
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
    TD_STAMP(1);   // reflector
    yj1 = 0.f;
#pragma unroll
    for (int w = 0; w < TD_WAVES; w += 4) {
      const v4 x = *reinterpret_cast<const v4 *>(s_slotB + w);
      yj1 += (x[0] + x[1]) + (x[2] + x[3]);
    }
    if (careful) {   // t1, t2 over the trailing rows with the actual v (every workgroup, redundantly)
      float a1 = 0.f, a2 = 0.f;
      if (lane < c) {
        constexpr int CU = 16;
        const float gl_ = s_gam[lane];
        for (int i0_ = r0 + wave; i0_ < n; i0_ += TD_WAVES * CU) {
          float Vr[CU], Wr[CU];
#pragma unroll
          for (int t = 0; t < CU; ++t) {
            const int i = min(i0_ + t * TD_WAVES, n - 1);
            Vr[t] = L1(true, oV + (long)i * TD_NB + lane);
            Wr[t] = L1(true, oW + (long)i * TD_NB + lane);
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
    TD_STAMP(2);   // careful path
    const float t1l = s_t1[lane], t2l = s_t2[lane], wj1l = s_wj1[lane], vj1l = s_vj1[lane], gl = s_gam[lane];
    const float w0j1 = tau * (yj1 - wave_sum_dpp(lane < c ? vj1l * t1l + wj1l * t2l : 0.f));
    if (blockIdx.x == 0 && wave == 0) {
      const float x = wave_sum_dpp(lane < c ? vk0 * wk0 : 0.f);
      if (lane == 0) {
        p.D[j] = ajj - 2.f * x;
        p.E[j] = beta;
        __hip_atomic_store(p.tau + j, tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by the next column
        S1(oV + (long)r0 * TD_NB + c, 1.f);
        S1(oW + (long)r0 * TD_NB + c, w0j1);
        S1(p.o_vv[out] + r0, 1.f);
        if (c > 0) S1(p.o_gam + cp, gamma);
      }
    }

    // ---- rows i >= j+2:  y_i = A[i][:] v,  w0_i,  next column's u0_i,  partial sums ----
    float accPW = 0.f, accPV = 0.f, accPWv = 0.f, accPVv = 0.f;       // per lane k < c
    float a_wv = 0.f, a_uv = 0.f, a_vv = 0.f, a_wu = 0.f, a_wv2 = 0.f; // wave-uniform
    const v4 *sv = reinterpret_cast<const v4 *>(s_v);
    for (int ibase = wbase; ibase < wend; ibase += RPW) {
      float acc[RPW];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) acc[rr] = 0.f;
      v4 nxt[RPW][UN];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int t = 0; t < UN; ++t) nxt[rr][t] = ldA(rr, 64 * UN + lane + 64 * t);
      for (int q0 = 0; q0 < nq; q0 += 64 * UN) {
        v4 nx2[RPW][UN];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
          for (int t = 0; t < UN; ++t) nx2[rr][t] = ldA(rr, q0 + 2 * 64 * UN + lane + 64 * t);
#pragma unroll
        for (int t = 0; t < UN; ++t) {
          const int q = q0 + lane + 64 * t;
          const v4 v = sv[min(q, nq - 1)];   // (rows are zero beyond the range: the clamped v drops out)
#pragma unroll
          for (int rr = 0; rr < RPW; ++rr) acc[rr] += dotv(cur[rr][t], v);
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
          S1(oW + (long)i * TD_NB + c, w0);
          S1(oV + (long)i * TD_NB + c, vi);
          S1(p.o_vv[out] + i, vi);
          S1(p.o_u0[out] + i, un);
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
    TD_STAMP(3);   // row pass
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
      // the group's record of the NEXT column: float atomics (the 16 workgroups of a group add in arrival order; the
      // prologue then adds the <= 16 group records in a fixed order)
      __hip_atomic_fetch_add(p.ws + p.o_gpart[(c + 1) % 3] + (long)grp * TD_NPART + tid, sacc, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x % TP_GROUP == 0 && tid < NQ4)   // the record this group adds into at the end of column c + 1
      S4(p.o_gpart[(c + 2) % 3] + (long)grp * TD_NPART + 4 * tid, zero4);
    if (c + 1 == p.ncol) break;
    // ---- hand-off to the next column: every store of this workgroup acknowledged, then the two-level barrier ----
    TD_STAMP(4);   // partial sums
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TD_STAMP(5);   // stores acknowledged
    const unsigned epoch = (unsigned)(c + 1);
    if (tid == 0) __hip_atomic_fetch_add(c_grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
      if (blockIdx.x % TP_GROUP == 0) {   // the group's leader: everybody of the group has arrived -> every release line
        scalar_wait(c_grp, (unsigned)gsize * epoch, c_err, p.fault, p.spin_limit, lane);
        if (lane < ngroups) __hip_atomic_fetch_add(c_rel0 + 32 * lane, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      scalar_wait(c_rel, (unsigned)ngroups * epoch, c_err, p.fault, p.spin_limit, lane);
    }
    __syncthreads();
    TD_STAMP(6);   // barrier
  }
#ifdef CLO_TD_TIMING
  if (tid == 0 && p.i0 == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.cnt + TP_CNT_WORDS) + (blockIdx.x == 0 ? 0 : 8);
    for (int i = 0; i < 8; ++i) dst[i] = tacc[i];
  }
  if (tid == 0 && p.i0 == 0)   // row pass and barrier time of EVERY workgroup
    reinterpret_cast<unsigned long long *>(p.cnt + TP_CNT_WORDS)[16 + blockIdx.x] = (tacc[3] << 32) | (tacc[6] & 0xffffffffull);
#endif
}

// End of a panel whose last column is jl (index cl inside the panel, panel origin i0): finish W for the rows of
// the trailing update (W_k = w0_k + gamma_k v_k, rows >= jl+1) and move the Householder vectors into the rows
// of A.
__global__ __launch_bounds__(256) void sytrd_panel_end_kernel(float *A, long lda, int n, int i0, int ncol,
                                                              const float *Vp, float *Wp,
                                                              const float *part, int g, const float *tau,
                                                              const float *gam, const unsigned *err, float *D) {
  // a wait of the panel launch was cut short (clo_common.h, "asynchronous faults"): the reduction is garbage and says so --
  // NaN in the first diagonal entry fails every acceptance test of the eigensolver instead of passing as plausible numbers
  if (blockIdx.x == 0 && threadIdx.x == 0 && *err != 0u) D[0] = __builtin_nanf("");
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
  return 2L * n * TD_NB + 4 * n4 + 2L * TD_NB + 64 + 3L * (TD_GMAX / TP_GROUP) * TD_NPART + TP_CNT_WORDS + 64 + 2 * TD_GMAX;   // (+ timing slots)
}

}  // namespace
}  // namespace clo

using namespace clo;

extern "C" long clo_sytrd_ws_bytes(int n) { return n > 0 ? td_ws_floats(n) * 4 : 0; }

extern "C" int clo_sytrd_f32(float *A, long lda, int n, float *D, float *E, float *tau, float *ws,
                             long ws_bytes, int max_blocks, void *stream) {
  CLO_REQUIRE(n >= 3 && n <= TD_NMAX, "clo_sytrd_f32: order %d outside [3, %d]", n, TD_NMAX);
  CLO_REQUIRE(A && D && E && tau && ws, "clo_sytrd_f32: null operand");
  CLO_REQUIRE(lda >= ((n + 3) & ~3) && lda % 4 == 0 && aligned16(A) && aligned16(ws),
              "clo_sytrd_f32: rows must be 16-byte aligned and zero-padded to a multiple of 4 columns (lda %ld)", lda);
  CLO_REQUIRE(ws_bytes >= clo_sytrd_ws_bytes(n), "clo_sytrd_f32: workspace too small");
  CLO_REQUIRE(max_blocks >= 0, "clo_sytrd_f32: negative max_blocks");
  hipStream_t st = (hipStream_t)stream;
  const long n4 = (n + 3) & ~3L;
  float *Vp = ws, *Wp = Vp + (long)n * TD_NB;
  float *u0[2] = {Wp + (long)n * TD_NB, Wp + (long)n * TD_NB + n4};
  float *vv[2] = {u0[1] + n4, u0[1] + 2 * n4};
  float *gam = vv[1] + n4;   // [TD_NB]
  // three rotating buffers of group records (column c reads c % 3, adds into (c + 1) % 3, clears (c + 2) % 3), then the
  // counters: ONE memset per panel clears both
  constexpr long GREC = (long)(TD_GMAX / TP_GROUP) * TD_NPART;
  float *gpart[3] = {gam + 2 * TD_NB + 64, gam + 2 * TD_NB + 64 + GREC, gam + 2 * TD_NB + 64 + 2 * GREC};
  unsigned *cnt = reinterpret_cast<unsigned *>(gpart[2] + GREC);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  // a panel grid spins on its own workgroups: never more of them than can be resident at once (one per CU of THIS
  // device -- a partition or a CU-masked queue has fewer than 256), whatever the caller asked for
  int gmax = std::min(max_blocks > 0 ? std::min(max_blocks, TD_GMAX) : TD_GMAX, device_cu_count(dev));
  if (fault_take(dev, FAULT_SYTRD)) {
    set_error("clo_sytrd_f32: an EARLIER reduction on device %d timed out waiting for its own workgroups (GPU shared with "
              "another process or CU-masked); its result is invalid.  Panel launches use at most a quarter of the compute "
              "units on this device from now on -- repeat the call.", dev);
    return CLO_EASYNC;
  }
  if (fault_disabled(dev, FAULT_SYTRD) || !fault_words_device(dev)) gmax = std::max(1, std::min(gmax, device_cu_count(dev) / 4));

  const size_t lds = (2 * n4 + TD_WAVES * TD_NPART + 5 * TD_NB + 5 * TD_WAVES + 16) * sizeof(float);
  // several host threads may run reductions at once (linalg_native.eigh_many): the attribute must be in
  // place for every instantiation before any of them launches with the larger size
  static std::mutex lds_mutex;
  static size_t lds_set_dev[64] = {0};   // per device: function attributes are per device
  {
  std::lock_guard<std::mutex> lds_lock(lds_mutex);
  size_t &lds_set = lds_set_dev[dev];
  if (lds > lds_set) {
    const void *fns[8] = {reinterpret_cast<const void *>(sytrd_panel_kernel<1>),
                          reinterpret_cast<const void *>(sytrd_panel_kernel<2>),
                          reinterpret_cast<const void *>(sytrd_panel_kernel<3>),
                          reinterpret_cast<const void *>(sytrd_panel_kernel<4>),
                          reinterpret_cast<const void *>(sytrd_panel_kernel<5>),
                          reinterpret_cast<const void *>(sytrd_panel_kernel<6>),
                          reinterpret_cast<const void *>(sytrd_panel_kernel<7>),
                          reinterpret_cast<const void *>(sytrd_panel_kernel<8>)};
    for (const void *fn : fns) {
      int rc = check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                         "clo_sytrd_f32: LDS attribute");
      if (rc != CLO_OK) return rc;
    }
    lds_set = lds;
  }
  }
  for (int i0 = 0; i0 < n - 2; i0 += TD_NB) {
    const int ncol = std::min(TD_NB, n - 2 - i0);
    // the panel's row owners: every wave of the G workgroups keeps `rpw` consecutive rows >= i0 + 2 for all columns
    const int nd = n - i0 - 2;
    const int Gmax = (int)std::max<long>(1, std::min<long>(gmax, cdiv(nd, TD_WAVES)));
    const int rpb = (int)cdiv(nd, Gmax);                 // rows per workgroup ...
    const int G = (int)cdiv(nd, rpb);                    // ... on as few workgroups as that needs
    const int rpw = (int)cdiv(rpb, TD_WAVES);
    int rc = check_hip(hipMemsetAsync(gpart[0], 0, (3 * GREC + TP_CNT_WORDS) * sizeof(float), st),
                       "clo_sytrd_f32: group records / counter reset");
    if (rc != CLO_OK) return rc;
    TpArgs a;
    a.A = A; a.lda = lda; a.n = n; a.i0 = i0; a.ncol = ncol; a.rpb = rpb; a.rpw = rpw; a.G = G;
    a.ws = ws; a.ws_floats = td_ws_floats(n);
    a.o_W = Wp - ws;
    a.o_u0[0] = u0[0] - ws; a.o_u0[1] = u0[1] - ws;
    a.o_vv[0] = vv[0] - ws; a.o_vv[1] = vv[1] - ws;
    a.o_gpart[0] = gpart[0] - ws; a.o_gpart[1] = gpart[1] - ws; a.o_gpart[2] = gpart[2] - ws;
    a.o_gam = gam - ws;
    a.D = D; a.E = E; a.tau = tau; a.cnt = cnt;
    a.fault = fault_words_device(dev);
    if (a.fault) a.fault += FAULT_SYTRD;
    a.spin_limit = spin_limit();
    // (no partly resident persistent grids side by side: csrc/persist_gate.h)
    PersistGate &gate = PersistGate::of(dev);
    rc = gate.admit(st, G);
    if (rc != CLO_OK) return rc;
    switch (std::min(rpw, 8)) {   // rows of a wave per pass (more rows per wave: several passes)
#define CLO_TP_CASE(R) \
  case R: hipLaunchKernelGGL(sytrd_panel_kernel<R>, dim3(G), dim3(TD_THREADS), lds, st, a); break;
      CLO_TP_CASE(1) CLO_TP_CASE(2) CLO_TP_CASE(3) CLO_TP_CASE(4)
      CLO_TP_CASE(5) CLO_TP_CASE(6) CLO_TP_CASE(7)
#undef CLO_TP_CASE
      default: hipLaunchKernelGGL(sytrd_panel_kernel<8>, dim3(G), dim3(TD_THREADS), lds, st, a); break;
    }
    const int flip = ncol % 3;   // the buffer the last column of the panel added its partial sums into
    const int g_prev = (int)cdiv(G, TP_GROUP);   // group records
    rc = check_hip(hipGetLastError(), "sytrd_panel_kernel");
    if (rc != CLO_OK) {
      gate.abort();
      return rc;
    }
    rc = gate.done(st);
    if (rc != CLO_OK) return rc;
    hipLaunchKernelGGL(sytrd_panel_end_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, A, lda, n, i0,
                       ncol, Vp, Wp, gpart[flip], g_prev, tau, gam,
                       cnt + 32 * (2 * (TD_GMAX / TP_GROUP)), D);
    CLO_CHECK_LAUNCH("sytrd_panel_end_kernel");
    // trailing update A[t:, t:] -= V W^T + W V^T on the MFMA GEMM engine (full square: the column
    // kernel reads complete rows)
    const int t = i0 + ncol, m = n - t;
    const float *Vt = Vp + (long)t * TD_NB, *Wt = Wp + (long)t * TD_NB;
    float *C = A + (long)t * lda + t;
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
