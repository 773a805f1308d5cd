// Curvature-vector products of fully-connected nets (reference ggn.py:17-168, hessian.py:13-69,
// gradient_moments.py:15-87, jacobian.py:14-358) without autograd.  Entry points at the end of the file:
//
//   clo_mlp_ggn_matvec      GGN / empirical Fisher / MC-GGN, one probe vector
//     N <= 8 rows  : six streaming launches that read W and V once (HBM-bound regime)
//       fwd_mfma_first_kernel  layer 1 forward + JVP, in-block split-K (no slabs)
//       fwd_mfma_kernel        other layers, 16x16x4 f32 MFMA tiles, split-K slabs
//       head_fwd_kernel        slab sum + partial products of a narrow last layer
//       head_bwd_kernel        loss Hessian + backward through the last layer
//       bwd_fused_kernel       data chain delta_{l-1} = phi' * (delta_l W_l), read-only sweep
//       outer_all_kernel       all layers' out_W = beta out_W + delta^T a in one write-only launch
//       (fwd_jvp_kernel, fwd_finish_kernel, bwd_finish_kernel, loss_hessian_kernel: unaligned
//        operands, wide heads, deeper nets)
//     N > 8 rows   : GEMM engine of gemm.hip (fused forward gemm_fwd3_kernel, epilogue-fused
//                    backward GEMMs) + head_rows_* / small_outer_* for narrow layers and biases
//   clo_mlp_ggn_matmat      K probe columns in the reference's K-trailing layout
//       kfwd_stream_kernel, kouter_stream_kernel, loss_cols_kernel, pack_at_multi_kernel
//   clo_mlp_hessian_matvec  exact Hessian by the R-operator (hess_combine_kernel + GEMMs)
//   clo_mlp_jvp / clo_mlp_vjp  the two halves on their own (Jacobian operators)
//   clo_mlp_fwd_jvp_layer / clo_mlp_bwd_layer / clo_loss_hessian_apply  per-layer building blocks
#include "clo_common.h"
#include "gemm.h"
#include "mlp_loss.h"
#include "persist_gate.h"

#include <mutex>

namespace clo {

constexpr int NB = 8;        // batch rows per skinny pass
constexpr int QN = 1;        // 256-float sub-slices per k-chunk
constexpr int KC = 256 * QN; // k-chunk of the forward kernel: 64 lanes x QN x float4
constexpr int FWD_WAVES = 8;
constexpr int FWD_R = 2;     // output features per wave
constexpr int FWD_ROWS = FWD_WAVES * FWD_R;
constexpr int CW = 256;      // column chunk of the backward kernel: 64 lanes x float4
constexpr int BWD_WAVES = 8;

// acc + a.b as a chain of 4 FMAs (no separate multiply / add)
__device__ __forceinline__ float fma4(const float4 &a, const float4 &b, float acc) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc))));
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// streamed-once data: non-temporal policy (global_load_dwordx4 ... nt)
__device__ __forceinline__ float4 ld4nt(const float *p) {
  typedef float __attribute__((ext_vector_type(4))) v4;
  const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st4nt(float *p, const float4 &v) {
  typedef float __attribute__((ext_vector_type(4))) v4;
  __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, reinterpret_cast<v4 *>(p));
}
#ifdef CLO_NT_WEIGHTS
#define CLO_LDW ld4nt
#else
#define CLO_LDW ld4
#endif
// the result stream is written once and not read again by the matvec: nt stores (+1 % measured)
#ifdef CLO_TEMPORAL_STORES
#define CLO_STW st4
#else
#define CLO_STW st4nt
#endif

// Load the lane's 4 floats of 256-wide sub-slice q of a row (k offset k0).
// VEC: lane owns 4 consecutive floats (one 16-byte load); scalar: 4 loads strided by 64.
// All loads are UNCONDITIONAL (out-of-range lanes read a clamped, valid address and the
// value is zeroed by a select): a predicated load makes hipcc branch around it and drain
// vmcnt per element, which serialises the whole prefetch.
template <bool VEC>
__device__ __forceinline__ float4 load_row4(const float *__restrict__ row, int k0, int lane, int q,
                                            int d_in) {
  float4 v;
  if (VEC) {
    const int k = k0 + q * 256 + lane * 4;
    const bool ok = k < d_in;
    v = ld4(row + (ok ? k : 0));
    if (!ok) v = zero4();
  } else {
    float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + q * 256 + e * 64 + lane;
      const bool ok = k < d_in;
      const float x = row[ok ? k : 0];
      pv[e] = ok ? x : 0.f;
    }
  }
  return v;
}

// Same addressing, but out-of-range lanes keep whatever the clamped address holds (the
// consumer multiplies it by a zeroed operand).  No select -> no wait right behind the load.
template <bool VEC>
__device__ __forceinline__ float4 load_row4_raw(const float *__restrict__ row, int k0, int lane,
                                                int q, int d_in) {
  float4 v;
  if (VEC) {
    const int k = k0 + q * 256 + lane * 4;
    v = ld4(row + (k < d_in ? k : 0));
  } else {
    float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + q * 256 + e * 64 + lane;
      pv[e] = row[k < d_in ? k : 0];
    }
  }
  return v;
}
// Zero the lanes of a staged 4-vector that lie beyond d_in (or the whole vector if !row_ok).
// Bitwise AND with an all-ones / all-zeros mask: branch-free on purpose (a uniform `if` here
// splits the k-loop body into several basic blocks and the scheduler then drains the
// prefetch before the FMAs).
__device__ __forceinline__ float and_mask(float x, unsigned m) {
  return __uint_as_float(__float_as_uint(x) & m);
}
template <bool VEC>
__device__ __forceinline__ float4 mask_row4(float4 v, int k0, int lane, int q, int d_in, bool row_ok) {
  if (VEC) {
    const unsigned m = (row_ok && k0 + q * 256 + lane * 4 < d_in) ? 0xFFFFFFFFu : 0u;
    v.x = and_mask(v.x, m); v.y = and_mask(v.y, m); v.z = and_mask(v.z, m); v.w = and_mask(v.w, m);
  } else {
    float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned m = (row_ok && k0 + q * 256 + e * 64 + lane < d_in) ? 0xFFFFFFFFu : 0u;
      pv[e] = and_mask(pv[e], m);
    }
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// Fused forward + JVP through one Linear layer, N <= 8 batch rows.
// grid = (ceil(d_out / 16), ksplit); block = 8 waves.
// ------------------------------------------------------------------------------------------
template <bool VEC, bool HAS_V, bool HAS_DA>
__global__ __launch_bounds__(512, QN == 1 ? 4 : 2) void fwd_jvp_kernel(
    const float *__restrict__ W, const float *__restrict__ b, const float *__restrict__ VW,
    const float *__restrict__ Vb, const float *__restrict__ a_in,
    const float *__restrict__ da_in, float *__restrict__ a_out, float *__restrict__ da_out,
    float *__restrict__ dphi_out, float *__restrict__ part, int N, int d_in, int d_out, int act,
    int chunks_per_split) {
  constexpr bool TANGENT = HAS_V || HAS_DA;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // [2 buffers][NB][KC] for a, then the same for da
  float *s_a = smem;
  float *s_da = smem + 2 * NB * KC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> SGPR addressing
  const int j0 = blockIdx.x * FWD_ROWS + wave * FWD_R;
  const int nchunks_total = (d_in + KC - 1) / KC;
  const int c_begin = blockIdx.y * chunks_per_split;
  const int c_end = min(nchunks_total, c_begin + chunks_per_split);

  const float *wrow[FWD_R], *vrow[FWD_R];
#pragma unroll
  for (int r = 0; r < FWD_R; ++r) {
    const bool rok = j0 + r < d_out;
    wrow[r] = W + (long)(rok ? j0 + r : 0) * d_in;
    vrow[r] = HAS_V ? VW + (long)(rok ? j0 + r : 0) * d_in : nullptr;
  }

  float z[FWD_R][NB], dz[FWD_R][NB];
#pragma unroll
  for (int r = 0; r < FWD_R; ++r)
#pragma unroll
    for (int n = 0; n < NB; ++n) { z[r][n] = 0.f; dz[r][n] = 0.f; }

  // activation staging: wave w loads batch row n = w (8 waves <-> 8 rows), both sub-slices
  float4 st_a[QN], st_da[QN];
  float4 wn[FWD_R][QN], vn[FWD_R][QN];

  auto issue_loads = [&](int c) {
    const int k0 = c * KC;
    const long noff = (long)(wave < N ? wave : 0) * d_in;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      st_a[q] = load_row4_raw<VEC>(a_in + noff, k0, lane, q, d_in);
      if (HAS_DA) st_da[q] = load_row4_raw<VEC>(da_in + noff, k0, lane, q, d_in);
    }
    // weights: out-of-range k lanes / rows >= d_out read valid dummy data; the matching
    // activations are zeroed in stage_store (k) or the result is never written (rows)
#pragma unroll
    for (int r = 0; r < FWD_R; ++r)
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        wn[r][q] = load_row4_raw<VEC>(wrow[r], k0, lane, q, d_in);
        if (HAS_V) vn[r][q] = load_row4_raw<VEC>(vrow[r], k0, lane, q, d_in);
      }
  };
  auto stage_store = [&](int buf, int c) {
    float *pa = s_a + (buf * NB + wave) * KC;
    float *pd = s_da + (buf * NB + wave) * KC;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      st_a[q] = mask_row4<VEC>(st_a[q], c * KC, lane, q, d_in, wave < N);
      if (HAS_DA) st_da[q] = mask_row4<VEC>(st_da[q], c * KC, lane, q, d_in, wave < N);
      if (VEC) {
        *reinterpret_cast<float4 *>(pa + q * 256 + lane * 4) = st_a[q];
        if (HAS_DA) *reinterpret_cast<float4 *>(pd + q * 256 + lane * 4) = st_da[q];
      } else {
        const float *xa = reinterpret_cast<const float *>(&st_a[q]);
        const float *xd = reinterpret_cast<const float *>(&st_da[q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pa[q * 256 + e * 64 + lane] = xa[e];
          if (HAS_DA) pd[q * 256 + e * 64 + lane] = xd[e];
        }
      }
    }
  };

  if (c_begin < c_end) {
    issue_loads(c_begin);
    stage_store(0, c_begin);
  }
  __syncthreads();

  // Branch-free loop body (one basic block, so the sched_barriers below hold): the last
  // iteration harmlessly re-loads its own chunk instead of testing for "more".
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    float4 wc[FWD_R][QN], vc[FWD_R][QN];
#pragma unroll
    for (int r = 0; r < FWD_R; ++r)
#pragma unroll
      for (int q = 0; q < QN; ++q) { wc[r][q] = wn[r][q]; vc[r][q] = vn[r][q]; }
    const int cn = min(c + 1, c_end - 1);
    issue_loads(cn);  // 8 KiB of weights per wave in flight during the FMAs
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int q = 0; q < QN; ++q)
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float4 a4, d4;
        const float *pa = s_a + (buf * NB + n) * KC;
        const float *pd = s_da + (buf * NB + n) * KC;
        if (VEC) {
          a4 = ld4(pa + q * 256 + lane * 4);
          if (HAS_DA) d4 = ld4(pd + q * 256 + lane * 4);
        } else {
          a4 = make_float4(pa[q * 256 + lane], pa[q * 256 + 64 + lane], pa[q * 256 + 128 + lane],
                           pa[q * 256 + 192 + lane]);
          if (HAS_DA)
            d4 = make_float4(pd[q * 256 + lane], pd[q * 256 + 64 + lane],
                             pd[q * 256 + 128 + lane], pd[q * 256 + 192 + lane]);
        }
#pragma unroll
        for (int r = 0; r < FWD_R; ++r) {
          z[r][n] = fma4(wc[r][q], a4, z[r][n]);
          if (HAS_V) dz[r][n] = fma4(vc[r][q], a4, dz[r][n]);
          if (HAS_DA) dz[r][n] = fma4(wc[r][q], d4, dz[r][n]);
        }
        // keep at most two batch rows of LDS operands live (register budget)
        if (n & 1) __builtin_amdgcn_sched_barrier(0);
      }
    __builtin_amdgcn_sched_barrier(0);  // the wait for chunk c+1 must stay behind the FMAs
    stage_store(buf ^ 1, cn);
    __syncthreads();
  }

  // Transposing butterfly reduction: 32 partial sums (16 z + 16 dz) x 64 lanes -> lane l
  // ends up with the full sum of value (l >> 1): lanes 0..31 hold z, lanes 32..63 hold dz.
  float v[2 * FWD_R * NB];
#pragma unroll
  for (int r = 0; r < FWD_R; ++r)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      v[r * NB + n] = z[r][n];
      v[FWD_R * NB + r * NB + n] = dz[r][n];
    }
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int half = 16 >> s, off = 32 >> s;
    const bool upper = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const float send = upper ? v[i] : v[i + half];
      const float keep = upper ? v[i + half] : v[i];
      v[i] = keep + __shfl_xor(send, off, 64);
    }
  }
  const float mine = v[0] + __shfl_xor(v[0], 1, 64);
  const float other = __shfl(mine, (lane + 32) & 63, 64);  // lane < 32: the matching dz
  if (lane < 32 && !(lane & 1)) {
    const int idx = lane >> 1, r = idx / NB, n = idx % NB;
    const int j = j0 + r;
    if (j < d_out && n < N) {
      if (gridDim.y > 1) {  // raw partial sums: part[split][2][NB][d_out]
        float *pz = part + ((long)blockIdx.y * 2 * NB + n) * d_out + j;
        pz[0] = mine;
        if (TANGENT) pz[(long)NB * d_out] = other;
      } else {
        float dphi;
        const float av = act_apply(act, mine + (b ? b[j] : 0.f), dphi);
        a_out[(long)n * d_out + j] = av;
        if (dphi_out) dphi_out[(long)n * d_out + j] = dphi;
        if (TANGENT) da_out[(long)n * d_out + j] = dphi * (other + ((HAS_V && Vb) ? Vb[j] : 0.f));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// MFMA forward + JVP (16-byte aligned rows).  One v_mfma_f32_16x16x4_f32 tile per row group:
//   A rows  0..7  = W[j .. j+7]        rows 8..15 = VW[j .. j+7]
//   B cols  0..7  = a[n = 0..7]        cols 8..15 = da[n = 0..7]
//   D[i][n] = z part, D[i][8+n] + D[8+i][n] = dz part (the fourth quadrant is unused).
// Lane (idx = l & 15, s = l >> 4) feeds ONE float4 of row idx at k + 4s for A (global) and for
// B (LDS); its four elements feed four MFMAs (the k-slot <-> k assignment is a free permutation
// as long as A and B agree).  A 16-k step costs RG global loads, one ds_read_b128 and 4 RG
// MFMAs per wave: ~30x fewer instructions than the VALU kernel above, no cross-lane reduction.
//
// grid = (row blocks of 4 waves x RG x 8 features, K ranges).  A block stages the activations of
// its K range in LDS ONCE (shared by its waves, which own different rows), then every wave
// streams its rows barrier-free.  Every byte a wave loads is weight data: measured on MI355X
// the kernel time tracks the bytes requested from L2/HBM (~4.2 TB/s) whatever their source, so
// operand re-reads and clamped prefetches must not exist.
// ------------------------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int MF_WAVES = 4;
constexpr int MF_KB_MAX = 736;  // 16 x (736+4) floats = 46 KiB of LDS

template <int RG, bool HAS_V, bool HAS_DA>
__global__ __launch_bounds__(MF_WAVES * 64) void fwd_mfma_kernel(
    const float *__restrict__ W, const float *__restrict__ b, const float *__restrict__ VW,
    const float *__restrict__ Vb, const float *__restrict__ a_in,
    const float *__restrict__ da_in, float *__restrict__ a_out, float *__restrict__ da_out,
    float *__restrict__ dphi_out, float *__restrict__ part, int N, int d_in, int d_out, int act,
    int k_per_block) {
  constexpr bool TANGENT = HAS_V || HAS_DA;
  extern __shared__ __attribute__((aligned(16))) float s_b[];  // [16][k_per_block + 4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, s4 = (lane >> 4) * 4;
  const int kb0 = blockIdx.y * k_per_block;
  const int kb1 = min(d_in, kb0 + k_per_block);
  const int klen = kb1 - kb0;                 // multiple of 4
  const int ldb = k_per_block + 4;

  // ---- stage B = [a ; da] of this K range (zero beyond N rows / beyond klen / absent da)
  {
    const int q4 = (k_per_block + 3) >> 2;    // float4 per row
    for (int e = tid; e < 16 * q4; e += MF_WAVES * 64) {
      const int c = e / q4, kq = (e - c * q4) * 4;
      float4 v = zero4();
      const int n = c & 7;
      const bool present = (c < 8 || HAS_DA) && n < N && kq < klen;
      if (present) v = ld4(((c < 8) ? a_in : da_in) + (long)n * d_in + kb0 + kq);
      *reinterpret_cast<float4 *>(&s_b[c * ldb + kq]) = v;
    }
  }
  __syncthreads();

  const int j0 = (blockIdx.x * MF_WAVES + wave) * RG * 8;
  if (j0 >= d_out) return;  // whole wave idle (after the only barrier)

  // Operand rows.  Without VW the upper half of A aliases W (its quadrants of D are unused);
  // rows beyond d_out alias the last row: no predicated loads anywhere.
  const float *pA[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    const int row = min(j0 + g * 8 + (idx & 7), d_out - 1);
    pA[g] = ((HAS_V && idx >= 8) ? VW : W) + (long)row * d_in + kb0 + s4;
  }
  const float *pBs = s_b + idx * ldb + s4;

  f32x4 acc[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int U = 2;  // steps per load group
  const int nfull = klen >> 4;
  int step = 0;
  // two register buffers: the weight loads of group g + 1 are in flight while group g feeds the MFMAs
  struct Group { float4 av[U][RG]; };
  auto load = [&](Group &gr, int st0) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int g = 0; g < RG; ++g) gr.av[u][g] = CLO_LDW(pA[g] + (st0 + u) * 16);
  };
  auto mma = [&](const Group &gr, int st0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float4 bv = ld4(pBs + (st0 + u) * 16);
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr.av[u][g].x, bv.x, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr.av[u][g].y, bv.y, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr.av[u][g].z, bv.z, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr.av[u][g].w, bv.w, acc[g], 0, 0, 0);
      }
    }
  };
  {
    Group ga, gb;
    const int ngroups = nfull / U;
    if (ngroups > 0) load(ga, 0);
    int gi = 0;
    for (; gi + 1 < ngroups; gi += 2) {
      load(gb, (gi + 1) * U);
      mma(ga, gi * U);
      if (gi + 2 < ngroups) load(ga, (gi + 2) * U);
      mma(gb, (gi + 1) * U);
    }
    if (gi < ngroups) mma(ga, gi * U);
    step = ngroups * U;
  }
  // remaining full steps and the partial one: B is zero beyond klen, A only needs a valid address
  for (; step * 16 < klen; ++step) {
    const bool ok = step * 16 + s4 < klen;
    const float4 bv = ld4(pBs + step * 16);
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const float4 av = ld4(pA[g] + (ok ? step * 16 : 0));
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[g], 0, 0, 0);
    }
  }

  // ---- epilogue straight from the accumulators.  D layout: row = (lane>>4)*4 + r, col = lane&15.
  // z[n][j0+i]  = D[i][n]            -> lanes with lane>>4 in {0,1}, col n < 8
  // dz[n][j0+i] = D[i][8+n] + D[8+i][n] -> lane (q, 8+n) adds the value of lane (q+2, n) = lane+24
  const int q = lane >> 4, col = lane & 15;
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc[g][r];
      const float up = __shfl(v, (lane + 24) & 63, 64);   // D[8+i][n] for lanes (q<2, col>=8)
      const float zsrc = __shfl(v, (lane + 56) & 63, 64);  // D[i][n] for the same lanes (lane-8)
      if (q < 2 && col >= 8) {
        const int n = col - 8, j = j0 + g * 8 + q * 4 + r;
        if (n < N && j < d_out) {
          const float zsum = zsrc;
          const float dzsum = (HAS_DA ? v : 0.f) + (HAS_V ? up : 0.f);
          if (gridDim.y > 1) {  // raw partial sums: part[split][2][NB][d_out]
            float *pz = part + ((long)blockIdx.y * 2 * NB + n) * d_out + j;
            pz[0] = zsum;
            if (TANGENT) pz[(long)NB * d_out] = dzsum;
          } else {
            float dphi;
            const float aval = act_apply(act, zsum + (b ? b[j] : 0.f), dphi);
            a_out[(long)n * d_out + j] = aval;
            if (dphi_out) dphi_out[(long)n * d_out + j] = dphi;
            if (TANGENT) da_out[(long)n * d_out + j] = dphi * (dzsum + ((HAS_V && Vb) ? Vb[j] : 0.f));
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------
// First-layer variant (no incoming tangent da): the WAVES waves of a block split K among
// themselves for the SAME RG x 8 features and merge through LDS, so the layer needs neither
// split-K slabs nor a finish launch.  B = [x ; 0] comes straight from global memory (each wave
// only touches its own K range of x: 8 rows against the 16 RG weight rows it streams).
// ------------------------------------------------------------------------------------------
template <int WAVES, int RG, int U>
__global__ __launch_bounds__(WAVES * 64) void fwd_mfma_first_kernel(
    const float *__restrict__ W, const float *__restrict__ b, const float *__restrict__ VW,
    const float *__restrict__ Vb, const float *__restrict__ a_in, float *__restrict__ a_out,
    float *__restrict__ da_out, float *__restrict__ dphi_out, int N, int d_in, int d_out, int act,
    int k_per_wave, int fpb) {
  // fpb <= RG * 8 features per block (chosen so that the grid fills the CUs evenly); tile rows
  // past the block's features alias its last row (same address within one load instruction)
  __shared__ float s_red[WAVES][RG][4][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, s4 = (lane >> 4) * 4;
  const int kb0 = min(wave * k_per_wave, d_in);
  const int klen = min(d_in, kb0 + k_per_wave) - kb0;  // multiple of 4, may be 0
  const int j0 = blockIdx.x * fpb;
  const int jlast = min(j0 + fpb, d_out) - 1;

  const float *pA[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    const int row = min(j0 + g * 8 + (idx & 7), jlast);
    // (no tangent weights: rows 8 .. 15 of the tile alias the W rows -- the same addresses inside one load instruction --
    // and da_out is not written; the forward pass of the K-column products runs layers of <= 8 rows without slabs this way)
    pA[g] = ((idx >= 8 && VW) ? VW : W) + (long)row * d_in + kb0 + s4;
  }
  // B lanes: columns 0..7 = x rows, columns 8..15 (the absent da) and rows >= N are zero: they
  // load a valid duplicate address and are masked
  const int nb = idx & 7;
  const unsigned bmask = (idx < 8 && nb < N) ? 0xffffffffu : 0u;
  const float *pB = a_in + (long)min(nb, N - 1) * d_in + kb0 + s4;

  f32x4 acc[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nfull = klen >> 4;
  int step = 0;
  for (; step + U <= nfull; step += U) {
    float4 av[U][RG], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int g = 0; g < RG; ++g) av[u][g] = CLO_LDW(pA[g] + (step + u) * 16);
      bv[u] = ld4(pB + (step + u) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);  // every load of the group is issued before the first MFMA waits
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float bx = __uint_as_float(__float_as_uint(bv[u].x) & bmask);
      const float by = __uint_as_float(__float_as_uint(bv[u].y) & bmask);
      const float bz = __uint_as_float(__float_as_uint(bv[u].z) & bmask);
      const float bw = __uint_as_float(__float_as_uint(bv[u].w) & bmask);
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g].x, bx, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g].y, by, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g].z, bz, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g].w, bw, acc[g], 0, 0, 0);
      }
    }
  }
  for (; step * 16 < klen; ++step) {  // leftover full steps and the partial one
    const bool ok = step * 16 + s4 < klen;
    const unsigned m = ok ? bmask : 0u;
    const float4 bv = ld4(pB + (ok ? step * 16 : 0));
    const float bx = __uint_as_float(__float_as_uint(bv.x) & m);
    const float by = __uint_as_float(__float_as_uint(bv.y) & m);
    const float bz = __uint_as_float(__float_as_uint(bv.z) & m);
    const float bw = __uint_as_float(__float_as_uint(bv.w) & m);
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const float4 av = ld4(pA[g] + (ok ? step * 16 : 0));
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bx, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, by, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bz, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bw, acc[g], 0, 0, 0);
    }
  }

  // ---- merge the waves' K ranges, then the epilogue of fwd_mfma_kernel by wave g for group g
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_red[wave][g][r][lane] = acc[g][r];
  __syncthreads();
  if (wave >= RG) return;
  const int g = wave;
  const int q = lane >> 4, col = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += s_red[w][g][r][lane];
    const float up = __shfl(v, (lane + 24) & 63, 64);    // D[8+i][n]: the VW . x part
    const float zsrc = __shfl(v, (lane + 56) & 63, 64);  // D[i][n]
    if (q < 2 && col >= 8) {
      const int n = col - 8, j = j0 + g * 8 + q * 4 + r;
      if (n < N && j <= jlast) {
        float dphi;
        const float aval = act_apply(act, zsrc + (b ? b[j] : 0.f), dphi);
        a_out[(long)n * d_out + j] = aval;
        if (dphi_out) dphi_out[(long)n * d_out + j] = dphi;
        if (da_out) da_out[(long)n * d_out + j] = dphi * (up + (Vb ? Vb[j] : 0.f));
      }
    }
  }
}

// Sum the split-K slabs of fwd_jvp_kernel and apply bias + activation.
__global__ void fwd_finish_kernel(const float *__restrict__ part, int ksplit,
                                  const float *__restrict__ b, const float *__restrict__ Vb,
                                  float *__restrict__ a_out, float *__restrict__ da_out,
                                  float *__restrict__ dphi_out, int N, int d_out, int act,
                                  int part_rows = NB) {
  const int total = N * d_out;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int n = e / d_out, j = e % d_out;
    float zz = b ? b[j] : 0.f, dzz = Vb ? Vb[j] : 0.f;
    const float *p0 = part + (long)n * d_out + j;
    const long sstride = 2L * part_rows * d_out, dzoff = (long)part_rows * d_out;
    int s = 0;
    for (; s + 3 < ksplit; s += 4) {  // four independent slabs per trip
      const float z0 = p0[(s + 0) * sstride], z1 = p0[(s + 1) * sstride];
      const float z2 = p0[(s + 2) * sstride], z3 = p0[(s + 3) * sstride];
      float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
      if (da_out) {
        d0 = p0[(s + 0) * sstride + dzoff]; d1 = p0[(s + 1) * sstride + dzoff];
        d2 = p0[(s + 2) * sstride + dzoff]; d3 = p0[(s + 3) * sstride + dzoff];
      }
      zz += (z0 + z1) + (z2 + z3);
      dzz += (d0 + d1) + (d2 + d3);
    }
    for (; s < ksplit; ++s) {
      zz += p0[s * sstride];
      if (da_out) dzz += p0[s * sstride + dzoff];
    }
    float dphi;
    a_out[e] = act_apply(act, zz, dphi);
    if (dphi_out) dphi_out[e] = dphi;
    if (da_out) da_out[e] = dphi * dzz;
  }
}

// ------------------------------------------------------------------------------------------
// Output-space curvature: one block per sample.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float *s_red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  return t;
}
__device__ __forceinline__ float block_max(float v, float *s_red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, s_red[w]);
  return t;
}

struct LossArgs {
  int kind;
  const float *f;       // [N][C] prediction
  const float *aux;     // RANK1: [N][rank][C]
  int aux_rank;
  const float *u;       // [N][C] J v
  const float *dphi_last;
  float *w;             // [N][C] result
  int C;
  float scale;
  // optional split-K slabs of the last (identity-activation) layer: part[s][2][part_rows][C]
  const float *part;
  int ksplit, part_rows;
  const float *b, *Vb;
  float *f_out, *u_out;
};

__global__ __launch_bounds__(256) void loss_hessian_kernel(const LossArgs p) {
  __shared__ float s_red[8];
  const int n = blockIdx.x, C = p.C;
  const float *fn = p.f + (long)n * C, *un = p.u + (long)n * C;
  if (p.part) {  // finish the last Linear layer first (identity activation)
    float *fo = p.f_out + (long)n * C, *uo = p.u_out + (long)n * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float zz = p.b ? p.b[c] : 0.f, dzz = p.Vb ? p.Vb[c] : 0.f;
      const float *q0 = p.part + (long)n * C + c;
      const long sstride = 2L * p.part_rows * C, dzoff = (long)p.part_rows * C;
      int s = 0;
      for (; s + 3 < p.ksplit; s += 4) {
        const float z0 = q0[(s + 0) * sstride], z1 = q0[(s + 1) * sstride];
        const float z2 = q0[(s + 2) * sstride], z3 = q0[(s + 3) * sstride];
        const float d0 = q0[(s + 0) * sstride + dzoff], d1 = q0[(s + 1) * sstride + dzoff];
        const float d2 = q0[(s + 2) * sstride + dzoff], d3 = q0[(s + 3) * sstride + dzoff];
        zz += (z0 + z1) + (z2 + z3);
        dzz += (d0 + d1) + (d2 + d3);
      }
      for (; s < p.ksplit; ++s) {
        zz += q0[s * sstride];
        dzz += q0[s * sstride + dzoff];
      }
      fo[c] = zz;
      uo[c] = dzz;
    }
    __syncthreads();
    fn = fo;
    un = uo;
  }
  float *wn = p.w + (long)n * C;
  const float *dp = p.dphi_last ? p.dphi_last + (long)n * C : nullptr;
  const float scale = p.scale;
  if (p.kind == CLO_LOSS_MSE) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) wn[c] = scale * un[c] * (dp ? dp[c] : 1.f);
  } else if (p.kind == CLO_LOSS_BCE) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      wn[c] = scale * sigmoid_prime(fn[c]) * un[c] * (dp ? dp[c] : 1.f);
    }
  } else if (p.kind == CLO_LOSS_CE) {
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += blockDim.x) mx = fmaxf(mx, fn[c]);
    mx = block_max(mx, s_red);
    float se = 0.f, spu = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float e = __expf(fn[c] - mx);
      se += e;
      spu += e * un[c];
    }
    se = block_sum(se, s_red);
    spu = block_sum(spu, s_red);
    const float inv = 1.f / se;
    const float pu = spu * inv;  // p . u
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float pc = __expf(fn[c] - mx) * inv;
      wn[c] = scale * pc * (un[c] - pu) * (dp ? dp[c] : 1.f);
    }
  } else if (loss_is_ef(p.kind)) {  // empirical Fisher from the targets: w = scale g <g, u>, g from (f, target)
    const float *t = ef_target_row(p.kind, p.aux, n, C);
    float mx = -INFINITY, inv = 0.f;
    if (p.kind == CLO_LOSS_EF_CE) {
      for (int c = threadIdx.x; c < C; c += blockDim.x) mx = fmaxf(mx, fn[c]);
      mx = block_max(mx, s_red);
      float se = 0.f;
      for (int c = threadIdx.x; c < C; c += blockDim.x) se += __expf(fn[c] - mx);
      inv = 1.f / block_sum(se, s_red);
    }
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) s += ef_grad_at(p.kind, fn[c], t, c, mx, inv) * un[c];
    s = block_sum(s, s_red);
    for (int c = threadIdx.x; c < C; c += blockDim.x)
      wn[c] = scale * ef_grad_at(p.kind, fn[c], t, c, mx, inv) * s * (dp ? dp[c] : 1.f);
  } else {  // CLO_LOSS_RANK1: H_n = sum_m g_nm g_nm^T
    for (int c = threadIdx.x; c < C; c += blockDim.x) wn[c] = 0.f;
    for (int m = 0; m < p.aux_rank; ++m) {
      const float *g = p.aux + ((long)n * p.aux_rank + m) * C;
      float s = 0.f;
      for (int c = threadIdx.x; c < C; c += blockDim.x) s += g[c] * un[c];
      s = block_sum(s, s_red);
      for (int c = threadIdx.x; c < C; c += blockDim.x) wn[c] += scale * g[c] * s;
    }
    if (dp) {
      __syncthreads();
      for (int c = threadIdx.x; c < C; c += blockDim.x) wn[c] *= dp[c];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fused backward through one Linear layer, N <= 8.
//   out_W[j][i] = beta out_W[j][i] + alpha sum_n delta[n][j] a_prev[n][i]       (if OUTER)
//   out_b[j]    = beta out_b[j]    + alpha sum_n delta[n][j]                    (if out_b)
//   P[jb][n][i] = sum_{j in rows(jb)} W[j][i] delta[n][j]                       (if DPREV)
// grid = (column chunks of 256, JB row ranges); block = 8 waves; a wave walks rows
// jbase + wave, +8, ... with the W loads of 4 rows in flight.  If JB == 1 the kernel
// applies dphi_prev and writes delta_prev directly, else bwd_finish_kernel sums the slabs.
// ------------------------------------------------------------------------------------------
template <bool VEC, bool OUTER, bool DPREV, bool ACCUM>
__device__ __forceinline__ void bwd_block(
    const float *__restrict__ W, const float *__restrict__ delta,
    const float *__restrict__ a_prev, const float *__restrict__ dphi_prev,
    float *__restrict__ out_W, float *__restrict__ out_b, float *__restrict__ dst, float alpha,
    float beta, int N, int d_in, int d_out, int rows_per_block, int final_write, int bx, int by,
    float *smem, const float *__restrict__ dslabs = nullptr, int dnjb = 0,
    const float *__restrict__ dphi_cur = nullptr) {
  float *s_d = smem;                          // [rows_per_block][NB]
  float *s_red = smem + rows_per_block * NB;  // [BWD_WAVES][NB][CW]   (DPREV), [NB][CW] a_prev (else)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = bx * CW;
  const int jbase = by * rows_per_block;
  const int jend = min(d_out, jbase + rows_per_block);

  // delta[n][jbase ..]: coalesced along j, stored transposed ([row][n]) for broadcast reads
  for (int e = tid; e < rows_per_block * NB; e += 512) {
    const int n = e / rows_per_block, jj = e - n * rows_per_block;
    const int j = jbase + jj;
    float v = 0.f;
    if (j < d_out && n < N) {
      if (dslabs) {  // delta = dphi * (sum of the producer's row-range slabs): no finish launch
        const float *ps = dslabs + (long)n * d_out + j;
        const long stride = (long)NB * d_out;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int jb = 0;
        for (; jb + 3 < dnjb; jb += 4) {
          s0 += ps[(jb + 0) * stride]; s1 += ps[(jb + 1) * stride];
          s2 += ps[(jb + 2) * stride]; s3 += ps[(jb + 3) * stride];
        }
        for (; jb < dnjb; ++jb) s0 += ps[jb * stride];
        v = ((s0 + s1) + (s2 + s3)) * dphi_cur[(long)n * d_out + j];
      } else {
        v = delta[(long)n * d_out + j];
      }
    }
    s_d[jj * NB + n] = v;
  }
  // a_prev[0..7][i0 .. i0+255]: loaded ONCE per block (wave n loads row n) and shared through
  // LDS -- every byte requested from L2/HBM costs the same, so no per-wave re-loads
  float4 a[NB];
  if (OUTER) {
    const int n = wave;  // BWD_WAVES == NB
    float4 v = load_row4<VEC>(a_prev + (long)(n < N ? n : 0) * d_in, i0, lane, 0, d_in);
    if (n >= N) v = zero4();
    if (VEC) {
      *reinterpret_cast<float4 *>(&s_red[n * CW + lane * 4]) = v;
    } else {
      s_red[n * CW + lane] = v.x; s_red[n * CW + 64 + lane] = v.y;
      s_red[n * CW + 128 + lane] = v.z; s_red[n * CW + 192 + lane] = v.w;
    }
  }
  __syncthreads();
  if (OUTER) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      if (VEC)
        a[n] = ld4(&s_red[n * CW + lane * 4]);
      else
        a[n] = make_float4(s_red[n * CW + lane], s_red[n * CW + 64 + lane],
                           s_red[n * CW + 128 + lane], s_red[n * CW + 192 + lane]);
    }
    if (DPREV) __syncthreads();  // s_red is reused for the cross-wave reduction below
  }

  if (OUTER && out_b && bx == 0) {
    for (int jj = tid; jj < jend - jbase; jj += 512) {
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < NB; ++n) s += s_d[jj * NB + n];
      const int j = jbase + jj;
      out_b[j] = (beta != 0.f ? beta * out_b[j] : 0.f) + alpha * s;
    }
  }

  float4 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) acc[n] = zero4();

  constexpr int P = (OUTER && DPREV && ACCUM) ? 4 : 8;  // rows in flight per wave (register budget: 128)
  const bool col_ok = VEC ? (i0 + lane * 4 < d_in) : true;
  for (int j = jbase + wave; j < jend; j += BWD_WAVES * P) {
    float4 w4[DPREV ? P : 1], old4[(OUTER && ACCUM) ? P : 1];
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int jt = min(j + t * BWD_WAVES, jend - 1);  // clamped: loads stay unconditional
      if (DPREV) w4[DPREV ? t : 0] = load_row4_raw<VEC>(W + (long)jt * d_in, i0, lane, 0, d_in);
      if (OUTER && ACCUM)
        old4[(OUTER && ACCUM) ? t : 0] = load_row4_raw<VEC>(out_W + (long)jt * d_in, i0, lane, 0, d_in);
    }
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int jt = j + t * BWD_WAVES;
      if (jt < jend) {
        const float *dj = &s_d[(jt - jbase) * NB];
        const float4 d0 = ld4(dj), d1 = ld4(dj + 4);
        const float dn[NB] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        if (OUTER) {
          float4 o = zero4();
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            o.x += dn[n] * a[n].x; o.y += dn[n] * a[n].y;
            o.z += dn[n] * a[n].z; o.w += dn[n] * a[n].w;
          }
          float4 r = make_float4(alpha * o.x, alpha * o.y, alpha * o.z, alpha * o.w);
          if (ACCUM) {
            const float4 od = old4[ACCUM ? t : 0];
            r.x += beta * od.x; r.y += beta * od.y; r.z += beta * od.z; r.w += beta * od.w;
          }
          float *po = out_W + (long)jt * d_in;
          if (VEC) {
            if (col_ok) CLO_STW(po + i0 + lane * 4, r);
          } else {
            const float *pr = reinterpret_cast<const float *>(&r);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = i0 + e * 64 + lane;
              if (i < d_in) po[i] = pr[e];
            }
          }
        }
        if (DPREV) {
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const float4 wv = w4[DPREV ? t : 0];
            acc[n].x += dn[n] * wv.x; acc[n].y += dn[n] * wv.y;
            acc[n].z += dn[n] * wv.z; acc[n].w += dn[n] * wv.w;
          }
        }
      }
    }
  }

  if (DPREV) {
    // cross-wave reduction through LDS; local column c = lane*4+e (VEC) or e*64+lane
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float *q = &s_red[(wave * NB + n) * CW];
      if (VEC) {
        *reinterpret_cast<float4 *>(q + lane * 4) = acc[n];
      } else {
        q[lane] = acc[n].x; q[64 + lane] = acc[n].y; q[128 + lane] = acc[n].z;
        q[192 + lane] = acc[n].w;
      }
    }
    __syncthreads();
    for (int e = tid; e < NB * CW; e += 512) {
      const int n = e >> 8, c = e & 255;
      const int i = i0 + c;
      if (n < N && i < d_in) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < BWD_WAVES; ++w) s += s_red[(w * NB + n) * CW + c];
        if (final_write)
          dst[(long)n * d_in + i] = s * dphi_prev[(long)n * d_in + i];
        else
          dst[((long)by * NB + n) * d_in + i] = s;
      }
    }
  }
}

template <bool VEC, bool OUTER, bool DPREV, bool ACCUM>
__global__ __launch_bounds__(512, 4) void bwd_fused_kernel(
    const float *__restrict__ W, const float *__restrict__ delta,
    const float *__restrict__ a_prev, const float *__restrict__ dphi_prev,
    float *__restrict__ out_W, float *__restrict__ out_b, float *__restrict__ dst, float alpha,
    float beta, int N, int d_in, int d_out, int rows_per_block, int final_write) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bwd_block<VEC, OUTER, DPREV, ACCUM>(W, delta, a_prev, dphi_prev, out_W, out_b, dst, alpha, beta, N,
                                      d_in, d_out, rows_per_block, final_write, blockIdx.x,
                                      blockIdx.y, smem);
}

// All parameter-gradient outer products of one matvec in ONE launch (write-only stream):
// out_W_l = beta out_W_l + delta_l^T a_{l-1} for up to 16 layers.  Measured on MI355X: a sweep
// that both reads W and writes out_W runs ~1.5x slower than the read-only and the write-only
// sweep back to back, so the data part (delta_prev) and this part are separate kernels.
constexpr int OUTER_MAXL = 16;
// rows per block: 32 (4 per wave) measured best on C2 (whole matvec: 128 rows 55.0 us, 64: 53.2, 32: 52.5,
// 256: 60.3) -- short blocks hide each other's staging prologue
#ifndef CLO_OUTER_ROWS
#define CLO_OUTER_ROWS 32
#endif
constexpr int OUTER_ROWS = CLO_OUTER_ROWS;
struct OuterAllArgs {
  int nlayers;
  int first_block[OUTER_MAXL + 1];
  const float *delta[OUTER_MAXL];
  const float *a_prev[OUTER_MAXL];
  float *out_W[OUTER_MAXL];
  float *out_b[OUTER_MAXL];
  int d_in[OUTER_MAXL], d_out[OUTER_MAXL], vec[OUTER_MAXL];
  const float *dslabs[OUTER_MAXL];  // non-null: delta_l = dphi_l * sum of these row-range slabs
  const float *dphi[OUTER_MAXL];
  int dnjb[OUTER_MAXL];
  float alpha, beta;
  int N;
};

template <bool ACCUM>
__global__ __launch_bounds__(512, 4) void outer_all_kernel(const OuterAllArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int l = 0;
  while (l + 1 < p.nlayers && (int)blockIdx.x >= p.first_block[l + 1]) ++l;
  const int local = blockIdx.x - p.first_block[l];
  const int cchunks = (p.d_in[l] + CW - 1) / CW;
  const int bx = local % cchunks, by = local / cchunks;
  if (p.vec[l])
    bwd_block<true, true, false, ACCUM>(nullptr, p.delta[l], p.a_prev[l], nullptr, p.out_W[l],
                                        p.out_b[l], nullptr, p.alpha, p.beta, p.N, p.d_in[l],
                                        p.d_out[l], OUTER_ROWS, 0, bx, by, smem, p.dslabs[l],
                                        p.dnjb[l], p.dphi[l]);
  else
    bwd_block<false, true, false, ACCUM>(nullptr, p.delta[l], p.a_prev[l], nullptr, p.out_W[l],
                                         p.out_b[l], nullptr, p.alpha, p.beta, p.N, p.d_in[l],
                                         p.d_out[l], OUTER_ROWS, 0, bx, by, smem, p.dslabs[l],
                                         p.dnjb[l], p.dphi[l]);
}

__global__ void bwd_finish_kernel(const float *__restrict__ P, const float *__restrict__ dphi_prev,
                                  float *__restrict__ delta_prev, int N, int d_in, int JB) {
  const int total = N * d_in;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int n = e / d_in, i = e % d_in;
    const float *p = P + (long)n * d_in + i;
    const long stride = (long)NB * d_in;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int jb = 0;
    for (; jb + 3 < JB; jb += 4) {
      s0 += p[(jb + 0) * stride]; s1 += p[(jb + 1) * stride];
      s2 += p[(jb + 2) * stride]; s3 += p[(jb + 3) * stride];
    }
    for (; jb < JB; ++jb) s0 += p[jb * stride];
    delta_prev[e] = ((s0 + s1) + (s2 + s3)) * dphi_prev[e];
  }
}

// ------------------------------------------------------------------------------------------
// "Head" fusion for a narrow last layer (d_L = C <= 16, identity output, N <= 8): the three tiny,
// latency-bound launches  fwd(L), loss Hessian, bwd(L)  are folded into the neighbours:
//   head_fwd_kernel : finish of layer L-1 (slab sum, bias, activation)  +  per-block partial
//                     products of layer L for the 256 features the block just produced
//   head_bwd_kernel : merge those partials -> f, J v ; loss Hessian -> delta_L ; backward through
//                     layer L (out_W_L, out_b_L, delta_{L-1}) for a 256-column chunk per block
// ------------------------------------------------------------------------------------------
constexpr int HEAD_CMAX = 16;

struct HeadFwdArgs {
  const float *part;  // slabs of layer L-1 (nullptr: a/da/dphi already final)
  int ksplit;
  const float *b, *Vb;          // bias of layer L-1 and its tangent
  float *a, *da, *dphi;         // [N][d] outputs of layer L-1 (inputs if part == nullptr)
  int N, d, act;
  const float *WL, *VL;         // [C][d] last-layer weight and its tangent
  int C;
  float *hp;                    // [N][nblk][2][HEAD_CMAX] partial z_L / dz_L
  int part_rows;                // rows per slab of `part` (0: NB)
};

__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadFwdArgs p) {
  const int tid = threadIdx.x;
  const int nblk = gridDim.x, n = blockIdx.y;
  const int j = blockIdx.x * 256 + tid;
  // last-layer operands of the partial product below: issued first, they do not depend on the
  // slab sum (latency-bound kernel: keep every independent load in flight from the start)
  const int hc = tid >> 4, hsub = tid & 15;  // 16 classes x 16 sub-ranges
  const int hjb = blockIdx.x * 256 + hsub * 16;
  float hw[16], hv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const bool ok = hc < p.C && hjb + q < p.d;
    const long off = ok ? (long)hc * p.d + hjb + q : 0;
    hw[q] = p.WL[off];
    hv[q] = p.VL[off];
    if (!ok) hw[q] = hv[q] = 0.f;
  }
  float av = 0.f, dav = 0.f;
  if (j < p.d) {
    const long e = (long)n * p.d + j;
    if (p.part) {
      float zz = p.b ? p.b[j] : 0.f, dzz = p.Vb ? p.Vb[j] : 0.f;
      const float *p0 = p.part + e;
      const int prow = p.part_rows ? p.part_rows : NB;
      const long sstride = 2L * prow * p.d, dzoff = (long)prow * p.d;
      // all slabs in flight at once (ksplit <= 16): clamped index + zero weight, no branches
      float zs[16], ds[16];
#pragma unroll
      for (int sp = 0; sp < 16; ++sp) {
        const int spc = sp < p.ksplit ? sp : 0;
        zs[sp] = p0[spc * sstride];
        ds[sp] = p0[spc * sstride + dzoff];
      }
#pragma unroll
      for (int sp = 0; sp < 16; ++sp) {
        zz += sp < p.ksplit ? zs[sp] : 0.f;
        dzz += sp < p.ksplit ? ds[sp] : 0.f;
      }
      for (int sp = 16; sp < p.ksplit; ++sp) {
        zz += p0[sp * sstride];
        dzz += p0[sp * sstride + dzoff];
      }
      float dphi;
      av = act_apply(p.act, zz, dphi);
      dav = dphi * dzz;
      p.a[e] = av;
      p.da[e] = dav;
      p.dphi[e] = dphi;
    } else {
      av = p.a[e];
      dav = p.da[e];
    }
  }
  // partial products of the last layer over this block's 256 features: thread (c, sub) sums 16
  // features from LDS, then a 16-lane shuffle reduction (short dependency chains)
  __shared__ float s_av[256], s_dav[256];
  s_av[tid] = av;
  s_dav[tid] = dav;
  __syncthreads();
  {
    const int c = hc, sub = hsub;
    float z = 0.f, dz = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float x = s_av[sub * 16 + q], dx = s_dav[sub * 16 + q];
      z = fmaf(hw[q], x, z);
      dz = fmaf(hw[q], dx, fmaf(hv[q], x, dz));
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      z += __shfl_xor(z, off, 64);
      dz += __shfl_xor(dz, off, 64);
    }
    if (sub == 0 && c < p.C) {
      float *o = p.hp + (((long)n * nblk + blockIdx.x) * 2) * HEAD_CMAX + c;
      o[0] = z;
      o[HEAD_CMAX] = dz;
    }
  }
}

struct HeadBwdArgs {
  const float *hp;  // [N][nblk][2][HEAD_CMAX]
  int nblk;
  const float *bL, *VbL;        // last-layer bias and its tangent (may be null)
  int kind;                     // CLO_LOSS_*
  const float *aux;             // RANK1: [N][rank][C]
  int aux_rank;
  float scale;                  // loss scale * alpha
  const float *WL;              // [C][d]
  const float *a_prev, *dphi_prev;  // [N][d]
  float *out_W, *out_b;         // [C][d], [C]
  float *delta_prev;            // [N][d]
  float beta;
  int N, d, C;
};

__global__ __launch_bounds__(256) void head_bwd_kernel(const HeadBwdArgs p) {
  __shared__ float s_f[NB][HEAD_CMAX], s_u[NB][HEAD_CMAX], s_dl[NB][HEAD_CMAX];
  const int tid = threadIdx.x;
  const int N = p.N, C = p.C;
  // operands of the column phase do not depend on the merge: get them in flight first
  const int i = blockIdx.x * 256 + tid;
  const int ic = i < p.d ? i : 0;
  float ap[NB], dpp[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const long off = (long)(n < N ? n : 0) * p.d + ic;
    ap[n] = p.a_prev[off];
    dpp[n] = p.delta_prev ? p.dphi_prev[off] : 0.f;
    if (n >= N) ap[n] = 0.f;
  }
  // ---- merge the head partials: f = b_L + sum_blk hp, u = Vb_L + sum_blk dhp  (every block)
  // thread = (n, which in {z, dz}, c): [2][HEAD_CMAX] is contiguous in hp, 8 slabs in flight
  {
    const int n = tid >> 5, r = tid & 31, c = r & (HEAD_CMAX - 1);
    float acc = 0.f;
    if (n < N && c < C) {
      const float *bias = (r < HEAD_CMAX) ? p.bL : p.VbL;
      acc = bias ? bias[c] : 0.f;
      const float *q = p.hp + ((long)n * p.nblk * 2) * HEAD_CMAX + r;
      for (int k0 = 0; k0 < p.nblk; k0 += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = q[(long)min(k0 + k, p.nblk - 1) * 2 * HEAD_CMAX];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += (k0 + k < p.nblk) ? t[k] : 0.f;
      }
    }
    if (r < HEAD_CMAX) s_f[n][c] = acc; else s_u[n][c] = acc;
  }
  __syncthreads();
  // ---- loss Hessian per sample (C <= 16: one thread per sample)
  if (tid < NB) {
    const int n = tid;
    if (n < N) {
      if (p.kind == CLO_LOSS_MSE) {
        for (int c = 0; c < C; ++c) s_dl[n][c] = p.scale * s_u[n][c];
      } else if (p.kind == CLO_LOSS_BCE) {
        for (int c = 0; c < C; ++c) {
          s_dl[n][c] = p.scale * sigmoid_prime(s_f[n][c]) * s_u[n][c];
        }
      } else if (p.kind == CLO_LOSS_CE) {
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, s_f[n][c]);
        float se = 0.f, spu = 0.f;
        for (int c = 0; c < C; ++c) {
          const float e = __expf(s_f[n][c] - mx);
          se += e;
          spu += e * s_u[n][c];
        }
        const float inv = 1.f / se, pu = spu * inv;
        for (int c = 0; c < C; ++c) {
          const float pc = __expf(s_f[n][c] - mx) * inv;
          s_dl[n][c] = p.scale * pc * (s_u[n][c] - pu);
        }
      } else if (loss_is_ef(p.kind)) {
        float g[HEAD_CMAX];
        ef_grad_row<HEAD_CMAX>(p.kind, &s_f[n][0], ef_target_row(p.kind, p.aux, n, C), C, g);
        float sdot = 0.f;
        for (int c = 0; c < C; ++c) sdot += g[c] * s_u[n][c];
        for (int c = 0; c < C; ++c) s_dl[n][c] = p.scale * g[c] * sdot;
      } else {
        for (int c = 0; c < C; ++c) s_dl[n][c] = 0.f;
        for (int m = 0; m < p.aux_rank; ++m) {
          const float *g = p.aux + ((long)n * p.aux_rank + m) * C;
          float sdot = 0.f;
          for (int c = 0; c < C; ++c) sdot += g[c] * s_u[n][c];
          for (int c = 0; c < C; ++c) s_dl[n][c] += p.scale * g[c] * sdot;
        }
      }
    } else {
      for (int c = 0; c < C; ++c) s_dl[n][c] = 0.f;
    }
  }
  __syncthreads();
  // ---- backward through the last layer for column i
  if (blockIdx.x == 0 && tid < C && p.out_b) {
    float sb = 0.f;
    for (int n = 0; n < N; ++n) sb += s_dl[n][tid];
    p.out_b[tid] = (p.beta != 0.f ? p.beta * p.out_b[tid] : 0.f) + sb;
  }
  if (i < p.d) {
    float acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = 0.f;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
      const float w = p.WL[(long)c * p.d + i];
      float o = 0.f;
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const float dl = s_dl[n][c];
        o = fmaf(dl, ap[n], o);
        acc[n] = fmaf(dl, w, acc[n]);
      }
      float *po = p.out_W + (long)c * p.d + i;
      *po = (p.beta != 0.f ? p.beta * *po : 0.f) + o;
    }
    if (p.delta_prev) {
#pragma unroll
      for (int n = 0; n < NB; ++n)
        if (n < N) p.delta_prev[(long)n * p.d + i] = acc[n] * dpp[n];
    }
  }
}

// ------------------------------------------------------------------------------------------
// 9 ... 64 batch rows: the weight-streaming chain with MFMA tiles in EVERY kernel (the VALU backward
// kernels above keep 8 rows of accumulators per lane and stop there; the GEMM engine needs a dozen
// launches on 128-wide tiles that are mostly padding at these sizes).  Npad = 16 NT rows, NT = 1 ... 4.
//   mid_fwd_kernel        forward + JVP: A = [8 rows of W ; 8 rows of V] straight from global memory,
//                         B = 16 batch rows of a (and of da) from LDS: acc1 = [W a ; V a], acc2 = [W da ; -]
//   head_fwd_kernel       slab sum + bias / activation (+ partial products of a narrow head)
//   head_bwd_rows_kernel  loss Hessian + backward through the head, NB rows at a time
//   mid_dprev_kernel      delta_{l-1} slabs: A = delta^T from LDS, B = 4 rows x 64 columns of W per load
//   mid_outer_kernel      out_W_l = beta out_W_l + delta_l^T a_{l-1} for all layers, 16 x 64 MFMA tiles
// ------------------------------------------------------------------------------------------
// row stride of [row][Npad] delta tiles in LDS: Npad + 17
// (odd since round 6: the tile is filled row index fastest -- coalesced reads of delta[n][j ..] -- and 49 / 81 are coprime with the
// 64 banks, so those writes are conflict-free too; the fragment reads pay one extra cycle on three banks)
constexpr int mid_ldd(int NT) { return NT <= 2 ? 49 : 81; }

#ifndef CLO_MIDF_SU
#define CLO_MIDF_SU 16
#endif
template <int NT, bool HAS_DA, int WV>
__global__ __launch_bounds__(WV * 64) void mid_fwd_kernel(
    const float *__restrict__ W, const float *__restrict__ VW, const float *__restrict__ a_in,
    const float *__restrict__ da_in, float *__restrict__ part, int N, int d_in, int d_out,
    int k_per_block
#ifdef CLO_MID_TIMING
    , unsigned long long *stamps
#endif
    ) {
#ifdef CLO_MID_TIMING
#define MIDF_STAMP(i) do { if (threadIdx.x == 0 && stamps) stamps[(4096 + blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define MIDF_STAMP(i) do { } while (0)
#endif
  MIDF_STAMP(0);
  // Round 6: a wave owns 16 features and runs THREE products per step -- z += W a, dz += W da, dz += V a -- with two
  // A fragments (16 rows of W, the same 16 rows of V).  The round-2 form stacked [8 rows of W ; 8 rows of V] into one
  // A tile and multiplied it with a AND da: four products, the fourth (V da) discarded -- a quarter of the MFMA time of
  // the kernel that is MFMA-bound from 32 batch rows on (C2 layer 2: 25.6 us at 32 rows, 45.6 at 64).
  constexpr int NP = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float s_b[];  // [(HAS_DA ? 2 : 1) * NP][k_per_block + 4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, s4 = (lane >> 4) * 4;
  const int kb0 = blockIdx.y * k_per_block;
  const int klen = min(d_in, kb0 + k_per_block) - kb0;  // multiple of 4
  const int ldb = k_per_block + 4;

  const int j0 = (blockIdx.x * WV + wave) * 16;
  const int row = min(j0 + idx, d_out - 1);
  const bool has_v = VW != nullptr;
  const float *pW = W + (long)row * d_in + kb0 + s4;
  const float *pV = (has_v ? VW : W) + (long)row * d_in + kb0 + s4;   // (no tangent weights: valid dummy addresses)
  constexpr int U = 2;
  struct Group { float4 w[U], v[U]; };
  auto load = [&](Group &gr, int st0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      gr.w[u] = CLO_LDW(pW + (st0 + u) * 16);
      gr.v[u] = CLO_LDW(pV + (st0 + u) * 16);
    }
  };
  const int nfull = klen >> 4, ngroups = nfull / U;
  Group ga, gb;
  // (the first weight group is requested BEHIND the first batch of activation loads, round 6: a CU's memory pipe delivers in issue
  // order, and the activations -- L2-resident, needed first -- would wait for weights coming from HBM)
  // (measured: -0.5 ... -1.2 us up to 48 rows, +0.7 ... +1.1 us at 49 ... 64 rows, where the kernel is MFMA-bound: old order there)
  bool w_first = ngroups > 0;
  if (NT >= 4 && w_first) {
    load(ga, 0);
    w_first = false;
  }

  {  // stage B: columns [0, NP) = rows of a, [NP, 2 NP) = rows of da; zero beyond N rows / klen.  CLO_MIDF_SU loads in flight per thread
    // (round 6: one load, one LDS write per trip was 16 dependent round trips at 64 rows -- 7.1 us before the first MFMA,
    // tools/r6/probe_mid_fwd_timeline.py); unconditional loads at clamped addresses, zeroed afterwards
    constexpr int NC = (HAS_DA ? 2 : 1) * NP, SU = NT >= 3 ? CLO_MIDF_SU : 8;   // (16 in flight lose 1.5 - 2 us up to 32 rows, win 1.2 us at 64)
    const int q4 = (k_per_block + 3) >> 2, total = NC * q4;
    for (int e0 = tid; e0 < total; e0 += SU * WV * 64) {
      float4 v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int e = min(e0 + u * WV * 64, total - 1);
        const int c = e / q4, kq = (e - c * q4) * 4;
        const int n = c < NP ? c : c - NP;
        const bool ok = n < N && kq < klen;
        v[u] = ld4((c < NP ? a_in : da_in) + (long)(ok ? n : 0) * d_in + kb0 + (ok ? kq : 0));
        if (!ok) v[u] = zero4();
      }
      if (w_first) {
        load(ga, 0);
        w_first = false;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int e = e0 + u * WV * 64;
        if (e < total) {
          const int c = e / q4, kq = (e - c * q4) * 4;
          *reinterpret_cast<float4 *>(&s_b[c * ldb + kq]) = v[u];
        }
      }
    }
    if (w_first) load(ga, 0);   // (threads beyond the tile)
  }
  __syncthreads();
  MIDF_STAMP(1);

  f32x4 accz[NT], accd[NT], accv[NT];   // W a | W da | V a  (three independent chains: no back-to-back dependency)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    accz[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    accd[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    accv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float *pBa = s_b + idx * ldb + s4;
  const float *pBd = s_b + (NP + idx) * ldb + s4;
  auto mma_step = [&](const float4 &wv, const float4 &vv, int st) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 ba = ld4(pBa + t * 16 * ldb + st * 16);
      float4 bd = zero4();
      if (HAS_DA) bd = ld4(pBd + t * 16 * ldb + st * 16);
#define CLO_MID_MM(E)                                                                              \
  accz[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.E, ba.E, accz[t], 0, 0, 0);                    \
  if (HAS_DA) accd[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.E, bd.E, accd[t], 0, 0, 0);        \
  if (has_v) accv[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.E, ba.E, accv[t], 0, 0, 0);
      CLO_MID_MM(x) CLO_MID_MM(y) CLO_MID_MM(z) CLO_MID_MM(w)
#undef CLO_MID_MM
    }
  };
  auto mma = [&](const Group &gr, int st0) {
#pragma unroll
    for (int u = 0; u < U; ++u) mma_step(gr.w[u], gr.v[u], st0 + u);
  };
  int step = 0;
  {
    int gi = 0;
    for (; gi + 1 < ngroups; gi += 2) {
      load(gb, (gi + 1) * U);
      mma(ga, gi * U);
      if (gi + 2 < ngroups) load(ga, (gi + 2) * U);
      mma(gb, (gi + 1) * U);
    }
    if (gi < ngroups) mma(ga, gi * U);
    step = ngroups * U;
  }
  for (; step * 16 < klen; ++step) {  // leftover full steps and the partial one (B is zero beyond klen)
    const bool ok = step * 16 + s4 < klen;
    const float4 wv = ld4(pW + (ok ? step * 16 : 0)), vv = ld4(pV + (ok ? step * 16 : 0));
    mma_step(wv, vv, step);
  }

  MIDF_STAMP(2);
  // D layout: row = (lane >> 4) * 4 + r = feature inside the wave's 16, col = lane & 15 = batch row in the tile: every
  // lane holds four consecutive features -> one float4 per (tile, quantity) into part[split][2][NP][d_out].
  const int q = lane >> 4, col = lane & 15;
  const int j = j0 + q * 4;
  if (j < d_out) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float *dst = part + (((long)blockIdx.y * 2) * NP + t * 16 + col) * d_out + j;
      st4(dst, make_float4(accz[t][0], accz[t][1], accz[t][2], accz[t][3]));
      st4(dst + (long)NP * d_out, make_float4(accd[t][0] + accv[t][0], accd[t][1] + accv[t][1],
                                                accd[t][2] + accv[t][2], accd[t][3] + accv[t][3]));
    }
  }
#ifdef CLO_MID_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MIDF_STAMP(3);
}

// Forward + JVP of a layer of the 9 ... 64-row chain WITHOUT split-K slabs: fwd_mfma_first_kernel's scheme
// with all 16 B columns of a tile carrying batch rows -- the 8 waves of a block split K among themselves for
// the same RG x 8 features, B (= a and, beyond the first layer, da) straight from global memory (the
// activations are L2-resident), merge through LDS, bias / activation in the same launch.
template <int NT, bool HAS_DA, int U>
__global__ __launch_bounds__(512) void mid_full_kernel(
    const float *__restrict__ W, const float *__restrict__ b, const float *__restrict__ VW,
    const float *__restrict__ Vb, const float *__restrict__ a_in, const float *__restrict__ da_in,
    float *__restrict__ a_out, float *__restrict__ da_out, float *__restrict__ dphi_out, int N, int d_in,
    int d_out, int act, int k_per_wave, int fpb) {
  constexpr int WAVES = 8, RG = 2;
  extern __shared__ __attribute__((aligned(16))) float s_mf[];   // [WAVES][RG][NT][2 (z, dz)][4][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, s4 = (lane >> 4) * 4;
  const int kb0 = min(wave * k_per_wave, d_in);
  const int klen = min(d_in, kb0 + k_per_wave) - kb0;  // multiple of 4, may be 0
  const int j0 = blockIdx.x * fpb;
  const int jlast = min(j0 + fpb, d_out) - 1;
  const float *pA[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    const int row = min(j0 + g * 8 + (idx & 7), jlast);
    // (no tangent weights: rows 8 .. 15 of the tile alias the W rows -- the same addresses inside one load instruction --
    // and da_out is not written; the forward pass of the K-column products runs layers of <= 8 rows without slabs this way)
    pA[g] = ((idx >= 8 && VW) ? VW : W) + (long)row * d_in + kb0 + s4;
  }
  long offB[NT];
  unsigned bmask[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = t * 16 + idx;
    bmask[t] = n < N ? 0xffffffffu : 0u;
    offB[t] = (long)min(n, N - 1) * d_in + kb0 + s4;
  }
  f32x4 acc1[RG][NT], acc2[RG][NT];
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc1[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc2[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  auto masked = [](float4 v, unsigned m) {
    return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                       __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
  };
  auto mm = [&](const float4 (&av)[RG], const float4 (&bv)[NT], const float4 (&bd)[NT], unsigned ok) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 x = masked(bv[t], bmask[t] & ok);
      float4 dx = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS_DA) dx = masked(bd[t], bmask[t] & ok);
#define CLO_MIDF_MM(E)                                                                                  \
  _Pragma("unroll") for (int g = 0; g < RG; ++g) {                                                     \
    acc1[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g].E, x.E, acc1[g][t], 0, 0, 0);              \
    if (HAS_DA) acc2[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g].E, dx.E, acc2[g][t], 0, 0, 0); \
  }
      CLO_MIDF_MM(x) CLO_MIDF_MM(y) CLO_MIDF_MM(z) CLO_MIDF_MM(w)
#undef CLO_MIDF_MM
    }
  };
  const int nfull = klen >> 4;
  int step = 0;
  for (; step + U <= nfull; step += U) {
    float4 av[U][RG], bv[U][NT], bd[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int g = 0; g < RG; ++g) av[u][g] = CLO_LDW(pA[g] + (step + u) * 16);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bv[u][t] = ld4(a_in + offB[t] + (step + u) * 16);
        if (HAS_DA) bd[u][t] = ld4(da_in + offB[t] + (step + u) * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // every load of the group is issued before the first MFMA waits
#pragma unroll
    for (int u = 0; u < U; ++u) mm(av[u], bv[u], bd[u], 0xffffffffu);
  }
  for (; step * 16 < klen; ++step) {  // leftover full steps and the partial one
    const bool ok = step * 16 + s4 < klen;
    float4 av[RG], bv[NT], bd[NT];
#pragma unroll
    for (int g = 0; g < RG; ++g) av[g] = ld4(pA[g] + (ok ? step * 16 : 0));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bv[t] = ld4(a_in + offB[t] + (ok ? step * 16 : 0));
      if (HAS_DA) bd[t] = ld4(da_in + offB[t] + (ok ? step * 16 : 0));
    }
    mm(av, bv, bd, ok ? 0xffffffffu : 0u);
  }
  // ---- per wave: z = D1 rows 0..7, dz = D1 rows 8..15 (V a) + D2 rows 0..7 (W da), held by lanes q < 2;
  // merge the waves' K ranges through LDS; (group, tile) pairs are finished by waves 0 .. RG NT - 1
  const int q = lane >> 4, col = lane & 15;
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc1[g][t][r];
        const float up = __shfl(v, (lane + 32) & 63, 64);
        if (q < 2) {
          float *dst = s_mf + ((((wave * RG + g) * NT + t) * 2) * 4 + r) * 32 + lane;
          dst[0] = v;
          dst[4 * 32] = up + (HAS_DA ? acc2[g][t][r] : 0.f);
        }
      }
  __syncthreads();
  if (q >= 2) return;
  for (int job = wave; job < RG * NT; job += WAVES) {
    const int g = job / NT, t = job % NT;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float z = 0.f, dz = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const float *src = s_mf + ((((w * RG + g) * NT + t) * 2) * 4 + r) * 32 + lane;
        z += src[0];
        dz += src[4 * 32];
      }
      const int n = t * 16 + col, j = j0 + g * 8 + q * 4 + r;
      if (n < N && j <= jlast) {
        float dphi;
        const float aval = act_apply(act, z + (b ? b[j] : 0.f), dphi);
        a_out[(long)n * d_out + j] = aval;
        dphi_out[(long)n * d_out + j] = dphi;
        da_out[(long)n * d_out + j] = dphi * (dz + (Vb ? Vb[j] : 0.f));
      }
    }
  }
}

// delta_{l-1} slabs for up to 64 rows: P[by][n][i] = sum_{j in rows(by)} delta[n][j] W[j][i].
// grid = (column chunks of 256, JB row ranges); 8 waves = 4 column quarters x 2 row halves (merged in LDS).
struct MidDelta {            // where delta_l comes from
  const float *delta;        // [N][ld] final (ld = d_out unless ld_delta is set), or null
  const float *dslabs;       // [njb][NP][d_out] row-range slabs of the previous launch (x dphi), or null
  const float *dphi;         // [N][d_out] (with dslabs)
  int njb;
  int ld_delta;              // row stride of `delta` (0: d_out)
};
template <int NT>
__device__ __forceinline__ float mid_delta_at(const MidDelta &md, int n, int j, int N, int d_out) {
  constexpr int NP = 16 * NT;
  if (n >= N || j >= d_out) return 0.f;
  if (md.dslabs) {
    const float *ps = md.dslabs + (long)n * d_out + j;
    const long stride = (long)NP * d_out;
    const float dp = md.dphi[(long)n * d_out + j];
    float acc = 0.f;
    for (int jb = 0; jb < md.njb; jb += 8) {  // 8 slabs in flight: clamped index + zero weight, no branches
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = ps[(long)min(jb + k, md.njb - 1) * stride];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += (jb + k < md.njb) ? t[k] : 0.f;
    }
    return acc * dp;
  }
  return md.delta[(long)n * (md.ld_delta ? md.ld_delta : d_out) + j];
}

struct MidDprevArgs {
  const float *W;
  MidDelta md;
  const float *dphi_prev;
  float *dst;
  int N, d_in, d_out, rows_per_block, final_write;
#ifdef CLO_MID_TIMING
  unsigned long long *stamps;
  int gx;
#endif
};
#ifdef CLO_MID_TIMING
#define MIDD_STAMP(i) do { if (threadIdx.x == 0 && dq.stamps) dq.stamps[(6144 + by * dq.gx + bx) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define MIDD_STAMP(i) do { } while (0)
#endif
// CQ = column groups of 64 per block (4: 256 columns x 2 row halves; 2, round 6: 128 columns x 4 row quarters -- half as many
// row-range slabs for the same number of blocks)
template <int NT, int CQ = 4>
__device__ __forceinline__ void mid_dprev_body(const MidDprevArgs &dq, int bx, int by, float *smem) {
  constexpr int NP = 16 * NT, MID_LDD = mid_ldd(NT), RH = 8 / CQ;
  MIDD_STAMP(0);
  const float *__restrict__ W = dq.W;
  const MidDelta &md = dq.md;
  const float *__restrict__ dphi_prev = dq.dphi_prev;
  float *__restrict__ dst = dq.dst;
  const int N = dq.N, d_in = dq.d_in, d_out = dq.d_out, rows_per_block = dq.rows_per_block, final_write = dq.final_write;
  float *s_d = smem;                                  // [rows_per_block (padded to 8)][MID_LDD]
  const int rpad = (rows_per_block + 7) & ~7;
  float *s_red = smem + rpad * MID_LDD;               // [4 waves][NT][4][4][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cq = wave % CQ, rh = wave / CQ;
  const int c16 = lane & 15, kg = lane >> 4;
  const int jbase = by * rows_per_block;
  const int i0 = bx * (64 * CQ) + cq * 64 + c16 * 4;
  const bool col_ok = i0 < d_in;   // d_in % 4 == 0: the lane's four columns are all in or all out
  // rows of this wave: its part of the block's range, in steps of 4
  const int half = ((rows_per_block + RH - 1) / RH + 3) & ~3;
  const int r0 = rh * half, r1 = min(rows_per_block, r0 + half);
  const float *pW = W + (long)(col_ok ? i0 : 0);
  auto wrow = [&](int jj) { return pW + (long)min(jbase + jj + kg, d_out - 1) * d_in; };
  constexpr int U = 8;
  float4 wv[U];
  // first weight loads in flight while delta is staged -- issued BEHIND the first batch of delta loads where delta is a plain array
  // (a CU's memory pipe delivers in issue order: delta, 64 KB from L2, would wait for 64 KB of weights from HBM)
  auto issue_w = [&]() {
#pragma unroll
    for (int u = 0; u < U; ++u) wv[u] = ld4(wrow(min(r0 + 4 * u, max(r1 - 1, r0))));
  };
  const bool delta_vec = !md.dslabs && (((md.ld_delta ? md.ld_delta : d_out) | d_out | rows_per_block) & 3) == 0 &&
                         (((unsigned long)md.delta) & 15ul) == 0;
  if (!delta_vec) issue_w();

  if (!md.dslabs) {
    // delta_l is a final array: consecutive threads take consecutive rows j of one batch row (coalesced), EIGHT unconditional loads at
    // clamped addresses in flight per thread, zeroed afterwards.  (Round 6: one element per trip, batch row fastest, was up to 31
    // dependent round trips of 64 four-byte requests each -- the staging, not the weight stream, set this kernel's time beyond 16 rows.)
    constexpr int SU = 8;
    const long ldd = md.ld_delta ? md.ld_delta : d_out;
    const int total = rpad * NP;
    if (delta_vec) {
      // 16 bytes = four rows j per load: a quarter of the load instructions, ONE batch in flight for up to 64 batch rows x 256 rows
      // (tools/r6/probe_mid_dprev_timeline.py: four batches of eight scalar loads, queued behind the weight loads issued above, took
      // 9.0 of the kernel's 20 us at 64 rows)
      const int rq = rpad >> 2, total4 = rq * NP;
      bool w_done = false;
      for (int e0 = tid; e0 < total4; e0 += SU * 512) {
        float4 v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int e = min(e0 + u * 512, total4 - 1);
          const int n = e / rq, jj = (e - n * rq) * 4;
          const bool ok = jj < rows_per_block && n < N && jbase + jj < d_out;
          v[u] = ld4(md.delta + (ok ? (long)n * ldd + jbase + jj : 0L));
          if (!ok) v[u] = zero4();
        }
        if (!w_done) {   // (first batch)
          issue_w();
          w_done = true;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int e = e0 + u * 512;
          if (e < total4) {
            const int n = e / rq, jj = (e - n * rq) * 4;
            s_d[jj * MID_LDD + n] = v[u].x;
            s_d[(jj + 1) * MID_LDD + n] = v[u].y;
            s_d[(jj + 2) * MID_LDD + n] = v[u].z;
            s_d[(jj + 3) * MID_LDD + n] = v[u].w;
          }
        }
      }
      if (!w_done) issue_w();   // (threads beyond the tile)
    } else
    for (int e0 = tid; e0 < total; e0 += SU * 512) {
      float v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int e = min(e0 + u * 512, total - 1);
        const int n = e / rpad, jj = e - n * rpad;
        const bool ok = jj < rows_per_block && n < N && jbase + jj < d_out;
        v[u] = md.delta[(ok ? (long)n * ldd + jbase + jj : 0L)];
        if (!ok) v[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int e = e0 + u * 512;
        if (e < total) {
          const int n = e / rpad, jj = e - n * rpad;
          s_d[jj * MID_LDD + n] = v[u];
        }
      }
    }
  } else {
    for (int e = tid; e < rpad * NP; e += 512) {
      const int n = e / rpad, jj = e - n * rpad;
      s_d[jj * MID_LDD + n] = jj < rows_per_block ? mid_delta_at<NT>(md, n, jbase + jj, N, d_out) : 0.f;
    }
  }
  __syncthreads();
  MIDD_STAMP(1);

  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int jr = r0; jr < r1; jr += 4 * U) {
    float4 cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = wv[u];
    if (jr + 4 * U < r1) {
#pragma unroll
      for (int u = 0; u < U; ++u) wv[u] = ld4(wrow(min(jr + 4 * U + 4 * u, r1 - 1)));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = jr + 4 * u;           // rows jj .. jj + 3 (kg selects the row of this lane)
      if (jj < r1) {                        // wave-uniform
        const bool rok = jj + kg < r1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float av = s_d[min(jj + kg, rpad - 1) * MID_LDD + t * 16 + c16];
          av = rok ? av : 0.f;
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cur[u].x, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cur[u].y, acc[t][1], 0, 0, 0);
          acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cur[u].z, acc[t][2], 0, 0, 0);
          acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cur[u].w, acc[t][3], 0, 0, 0);
        }
      }
    }
  }
  MIDD_STAMP(2);
  // merge the row parts pairwise through LDS (upper half of the parts -> lower half, RH / 2 = 4 / CQ waves x CQ column groups = 4 wave
  // slots per round), then D[n = 4 q + r][column c16 of component e] -> P[n][i0 + e]
#pragma unroll
  for (int stride = RH / 2; stride >= 1; stride /= 2) {
    if (rh >= stride && rh < 2 * stride) {
      const int slot = (rh - stride) * CQ + cq;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < 4; ++r) s_red[(((slot * NT + t) * 4 + e) * 4 + r) * 64 + lane] = acc[t][e][r];
    }
    __syncthreads();
    if (rh < stride && stride > 1) {
      const int slot = rh * CQ + cq;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][e][r] += s_red[(((slot * NT + t) * 4 + e) * 4 + r) * 64 + lane];
    }
    if (stride > 1) __syncthreads();
  }
  MIDD_STAMP(3);
  if (rh >= 1 || !col_ok) return;
  const int q = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = t * 16 + q * 4 + r;
      float4 v;
      v.x = acc[t][0][r] + s_red[(((cq * NT + t) * 4 + 0) * 4 + r) * 64 + lane];
      v.y = acc[t][1][r] + s_red[(((cq * NT + t) * 4 + 1) * 4 + r) * 64 + lane];
      v.z = acc[t][2][r] + s_red[(((cq * NT + t) * 4 + 2) * 4 + r) * 64 + lane];
      v.w = acc[t][3][r] + s_red[(((cq * NT + t) * 4 + 3) * 4 + r) * 64 + lane];
      if (final_write) {
        if (n < N) {
          const float4 dp = ld4(dphi_prev + (long)n * d_in + i0);
          st4(dst + (long)n * d_in + i0, make_float4(v.x * dp.x, v.y * dp.y, v.z * dp.z, v.w * dp.w));
        }
      } else {
        st4(dst + ((long)by * NP + n) * d_in + i0, v);
      }
    }
}

template <int NT, int CQ = 4>
__global__ __launch_bounds__(512) void mid_dprev_kernel(const MidDprevArgs q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mid_dprev_body<NT, CQ>(q, blockIdx.x, blockIdx.y, smem);
}

// All outer products of a matvec for up to 64 rows: out_W_l = beta out_W_l + alpha delta_l^T a_{l-1}, bias
// gradients = column sums of delta_l.  Block = 64 rows x 256 columns, 8 waves = 4 column quarters x 2
// row halves, each wave 2 x (16 x 64) MFMA tiles with K = the batch rows.
constexpr int MIDO_ROWS = 64;
struct MidOuterArgs {
  int nlayers;
  int first_block[OUTER_MAXL + 1];
  MidDelta md[OUTER_MAXL];
  const float *a_prev[OUTER_MAXL];
  float *out_W[OUTER_MAXL];
  float *out_b[OUTER_MAXL];
  int d_in[OUTER_MAXL], d_out[OUTER_MAXL];
  float alpha, beta;
  int N;
#ifdef CLO_MID_TIMING
  unsigned long long *stamps;   // [blocks][8]: entry, staged, MFMAs done, end, HW_ID
#endif
};
#ifdef CLO_MID_TIMING
static unsigned long long *g_mid_stamps_host = nullptr;
#define MIDO_STAMP(i) do { if (threadIdx.x == 0 && p.stamps && block < 4096) p.stamps[block * 8 + (i)] = wall_clock64(); } while (0)
#else
#define MIDO_STAMP(i) do { } while (0)
#endif
// CW = columns per block (256: 4 column quarters x 2 row halves of two 16-row tiles; 128 beyond 32 batch rows
// -- 2 column halves x 4 row quarters of one tile -- so that three blocks still fit a CU's LDS).
template <int NT, bool ACCUM, int CW>
__device__ __forceinline__ void mid_outer_body(const MidOuterArgs &p, int block, float *smem_o) {
  constexpr int NP = 16 * NT;
  constexpr int NCQ = CW / 64, NRH = 8 / NCQ, RT = MIDO_ROWS / (16 * NRH);
  constexpr int LDR = MIDO_ROWS + 16;  // [NP][rows + 16]: conflict-free A-operand reads
  float *s_dT = smem_o;              // [NP][LDR]
  float *s_a = smem_o + NP * LDR;    // [NP][CW]
  int l = 0;
  while (l + 1 < p.nlayers && block >= p.first_block[l + 1]) ++l;
  const int local = block - p.first_block[l];
  const int d_in = p.d_in[l], d_out = p.d_out[l], N = p.N;
  const int cchunks = (d_in + CW - 1) / CW;
  const int bx = local % cchunks, by = local / cchunks;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cq = wave % NCQ, rh = wave / NCQ;
  const int l16 = lane & 15, kg = lane >> 4;
  const int jbase = by * MIDO_ROWS;
  MIDO_STAMP(0);
#ifdef CLO_MID_TIMING
  if (threadIdx.x == 0 && p.stamps && block < 4096) { p.stamps[block * 8 + 4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); p.stamps[block * 8 + 5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }
#endif
  // stage delta^T [n][row] and a [n][256 columns]
  for (int e = tid; e < NP * MIDO_ROWS; e += 512) {
    const int n = e / MIDO_ROWS, jj = e - n * MIDO_ROWS;
    s_dT[n * LDR + jj] = mid_delta_at<NT>(p.md[l], n, jbase + jj, N, d_out);
  }
  for (int e = tid; e < NP * (CW / 4); e += 512) {
    const int n = e / (CW / 4), c4 = (e % (CW / 4)) * 4;
    const int i = bx * CW + c4;
    float4 v = zero4();
    if (n < N && i < d_in) v = ld4(p.a_prev[l] + (long)n * d_in + i);
    *reinterpret_cast<float4 *>(&s_a[n * CW + c4]) = v;
  }
  __syncthreads();
  MIDO_STAMP(1);
  if (p.out_b[l] && bx == 0 && tid < MIDO_ROWS && jbase + tid < d_out) {
    float sb = 0.f;
    for (int n = 0; n < NP; ++n) sb += s_dT[n * LDR + tid];
    float *pb = p.out_b[l] + jbase + tid;
    *pb = (ACCUM ? p.beta * *pb : 0.f) + p.alpha * sb;
  }
  f32x4 acc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[rt][e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NP / 4; ++s) {
    const float4 bv = *reinterpret_cast<const float4 *>(&s_a[(4 * s + kg) * CW + cq * 64 + l16 * 4]);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float av = s_dT[(4 * s + kg) * LDR + (rh * RT + rt) * 16 + l16];
      acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.x, acc[rt][0], 0, 0, 0);
      acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.y, acc[rt][1], 0, 0, 0);
      acc[rt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.z, acc[rt][2], 0, 0, 0);
      acc[rt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.w, acc[rt][3], 0, 0, 0);
    }
  }
  // D[row = 4 q + r][column l16 of component e] -> out_W[jbase + rh 32 + rt 16 + 4 q + r][i0 + e]
  MIDO_STAMP(2);
  const int i0 = bx * CW + cq * 64 + l16 * 4;
  if (i0 >= d_in) return;
  const int q = lane >> 4;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = jbase + (rh * RT + rt) * 16 + q * 4 + r;
      if (j < d_out) {
        float4 v = make_float4(p.alpha * acc[rt][0][r], p.alpha * acc[rt][1][r], p.alpha * acc[rt][2][r],
                               p.alpha * acc[rt][3][r]);
        float *po = p.out_W[l] + (long)j * d_in + i0;
        if (ACCUM) {
          const float4 od = ld4(po);
          v.x += p.beta * od.x; v.y += p.beta * od.y; v.z += p.beta * od.z; v.w += p.beta * od.w;
        }
        CLO_STW(po, v);
      }
    }
#ifdef CLO_MID_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MIDO_STAMP(3);
}

template <int NT, bool ACCUM, int CW>
__global__ __launch_bounds__(512) void mid_outer_kernel(const MidOuterArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem_o[];
  mid_outer_body<NT, ACCUM, CW>(p, blockIdx.x, smem_o);
}

// Round 6, second form of the outer products (tools/r6/probe_mid_outer_timeline.py showed why the tile-per-block kernel above takes
// 28 us at 64 rows for 8.6 us of MFMA work and 7.6 us of stores: every block stages 48 KB for 32 KB of output, all resident blocks
// stage, multiply and store in lock-step, and the grid takes two such rounds).  Here a block owns 128 output rows x a RANGE of columns:
// each wave keeps its 16 rows of delta^T in REGISTERS (the A operand, NP / 4 values per lane, loaded once), the rows of a_{l-1} stream
// through a double-buffered LDS tile 64 columns at a time (global loads of chunk c + 1 in flight behind the MFMAs and stores of chunk c,
// one barrier per chunk), and the result leaves as 4 rows x 256 bytes per store instruction.  Ranges are sized so that the whole grid
// is resident at once (about two blocks per CU).
#ifndef CLO_MO2_STAGE_DELTA
#define CLO_MO2_STAGE_DELTA 1
#endif
constexpr int MO2_ROWS = 128, MO2_CW = 64, MO2_DPITCH = 144;
struct MidOuter2Args {
  int nlayers;
  int first_block[OUTER_MAXL + 1];
  MidDelta md[OUTER_MAXL];
  const float *a_prev[OUTER_MAXL];
  float *out_W[OUTER_MAXL];
  float *out_b[OUTER_MAXL];
  int d_in[OUTER_MAXL], d_out[OUTER_MAXL];
  int nranges[OUTER_MAXL], cr[OUTER_MAXL];   // column ranges per 128-row strip; columns per range (a multiple of MO2_CW)
  float alpha, beta;
  int N;
#ifdef CLO_MID_TIMING
  unsigned long long *stamps;
#endif
};
template <int NT, bool ACCUM>
__global__ __launch_bounds__(512) void mid_outer2_kernel(const MidOuter2Args p) {
  constexpr int NP = 16 * NT, KS = NP / 4;
  constexpr int NF4 = NP * (MO2_CW / 4);           // float4 per staged chunk
  constexpr int SL = (NF4 + 511) / 512;            // ... per thread
  extern __shared__ __attribute__((aligned(16))) float s_o2[];   // [2][NP][MO2_CW]
  const int block = blockIdx.x;
  int l = 0;
  while (l + 1 < p.nlayers && block >= p.first_block[l + 1]) ++l;
  const int local = block - p.first_block[l];
  const int strip = local / p.nranges[l], rg = local - strip * p.nranges[l];
  const int d_in = p.d_in[l], d_out = p.d_out[l], N = p.N;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kg = lane >> 4;
  const int c_begin = rg * p.cr[l], c_end = min(d_in, c_begin + p.cr[l]);
  const int nchunks = (c_end - c_begin + MO2_CW - 1) / MO2_CW;
  const float *__restrict__ ap = p.a_prev[l];
  MIDO_STAMP(0);
#ifdef CLO_MID_TIMING
  if (threadIdx.x == 0 && p.stamps && block < 4096) { p.stamps[block * 8 + 4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); p.stamps[block * 8 + 5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); }
#endif

  // chunk loader: thread -> (row n, 4 columns); rows beyond N and columns beyond the range are zeros
  float4 pre[SL];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int u = 0; u < SL; ++u) {
      const int e = u * 512 + tid, n = e / (MO2_CW / 4), c4 = (e - n * (MO2_CW / 4)) * 4;
      const int i = c_begin + c * MO2_CW + c4;
      const bool ok = e < NF4 && n < N && i < c_end;
      const float4 v = ld4(ap + (long)(ok ? n : 0) * d_in + (ok ? i : 0));   // (unconditional load, clamped address)
      pre[u] = ok ? v : zero4();
    }
  };
  auto store_chunk = [&](float *buf) {
#pragma unroll
    for (int u = 0; u < SL; ++u) {
      const int e = u * 512 + tid;
      if (e < NF4) *reinterpret_cast<float4 *>(buf + 4 * e) = pre[u];
    }
  };
  if (nchunks > 0) load_chunk(0);

  // the wave's 16 rows of delta^T: lane (l16, kg) holds delta[4 s + kg][j] for s = 0 .. KS - 1
  const int j = strip * MO2_ROWS + wave * 16 + l16;
  float af[KS];
  const MidDelta &mdl = p.md[l];
  const long ldd = mdl.ld_delta ? mdl.ld_delta : d_out;
  if (CLO_MO2_STAGE_DELTA && !mdl.dslabs && ((ldd | (long)d_out) & 3) == 0 && (((unsigned long)mdl.delta) & 15ul) == 0) {
    // a plain, aligned delta array: the block's strip [NP][128 rows j] goes through LDS -- coalesced 16-byte loads, 512 bytes per batch
    // row -- instead of KS four-byte loads per lane that each touch four rows x 64 bytes (the buffer of the a_{l-1} chunks is free until
    // the first chunk is stored; pitch 144: the four k groups of a fragment read hit disjoint banks)
    float *S = s_o2;
    float4 dv[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int e = u * 512 + tid, n = e >> 5, j4 = (e & 31) * 4, jg = strip * MO2_ROWS + j4;
      const bool ok = n < N && jg < d_out;
      dv[u] = ld4(mdl.delta + (ok ? (long)n * ldd + jg : 0L));
      if (!ok) dv[u] = zero4();
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int e = u * 512 + tid, n = e >> 5, j4 = (e & 31) * 4;
      *reinterpret_cast<float4 *>(S + n * MO2_DPITCH + j4) = dv[u];
    }
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) af[s4] = S[(4 * s4 + kg) * MO2_DPITCH + wave * 16 + l16];
    __syncthreads();   // (every fragment is in registers before the first chunk overwrites the buffer)
  } else {
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) af[s4] = mid_delta_at<NT>(mdl, 4 * s4 + kg, j, N, d_out);
  }
  if (p.out_b[l] && rg == 0) {   // bias gradient: column sums of delta
    float sb = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) sb += af[s4];
    sb += __shfl_xor(sb, 16);
    sb += __shfl_xor(sb, 32);
    if (kg == 0 && j < d_out) {
      float *pb = p.out_b[l] + j;
      *pb = (ACCUM ? p.beta * *pb : 0.f) + p.alpha * sb;
    }
  }
  if (nchunks <= 0) return;
  store_chunk(s_o2);
  __syncthreads();
  MIDO_STAMP(1);

  const int q = lane >> 4;
  const int row0 = strip * MO2_ROWS + wave * 16 + q * 4;
  float *__restrict__ ow = p.out_W[l];
  for (int c = 0; c < nchunks; ++c) {
    const float *buf = s_o2 + (c & 1) * (NP * MO2_CW);
    if (c + 1 < nchunks) load_chunk(c + 1);
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < KS; ++s4) {
      const float4 bv = *reinterpret_cast<const float4 *>(buf + (4 * s4 + kg) * MO2_CW + l16 * 4);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s4], bv.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s4], bv.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s4], bv.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s4], bv.w, acc[3], 0, 0, 0);
    }
    // D[row = 4 q + r][column l16 of component e] -> out_W[row0 + r][i0 + e]
    const int i0 = c_begin + c * MO2_CW + l16 * 4;
    if (i0 < c_end) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jr = row0 + r;
        if (jr < d_out) {
          float4 v = make_float4(p.alpha * acc[0][r], p.alpha * acc[1][r], p.alpha * acc[2][r], p.alpha * acc[3][r]);
          float *po = ow + (long)jr * d_in + i0;
          if (ACCUM) {
            const float4 od = ld4(po);
            v.x += p.beta * od.x; v.y += p.beta * od.y; v.z += p.beta * od.z; v.w += p.beta * od.w;
          }
          CLO_STW(po, v);
        }
      }
    }
    if (c + 1 < nchunks) store_chunk(s_o2 + ((c + 1) & 1) * (NP * MO2_CW));
    __syncthreads();
  }
  MIDO_STAMP(2);
#ifdef CLO_MID_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MIDO_STAMP(3);
}

template <typename K>
static int set_smem(K kernel, size_t bytes);
// the outer products of `count` layers (fields of a MidOuterArgs filled as for mid_outer_kernel) in one mid_outer2_kernel launch
#ifndef CLO_MO2_PER_CU
#define CLO_MO2_PER_CU 2
#endif
template <int NT>
static int launch_mid_outer2(const MidOuterArgs &oa, int count, float beta, int N, hipStream_t st) {
  constexpr int NP = 16 * NT;
  MidOuter2Args o2{};
#ifdef CLO_MID_TIMING
  o2.stamps = g_mid_stamps_host;
#endif
  o2.nlayers = count; o2.alpha = oa.alpha; o2.beta = beta; o2.N = N;
  long strip_cols = 0;
  for (int k = 0; k < count; ++k) strip_cols += cdiv(oa.d_out[k], MO2_ROWS) * (long)oa.d_in[k];
  // about CLO_MO2_PER_CU resident blocks per CU (2, 3, 4 and 6 measured alike; 1 is 7 % slower), ranges of whole 64-column chunks
  // (ranges of a multiple of 16 columns that fill the two-per-CU grid more evenly were measured: no difference, 65 ... 128 rows)
  const size_t smem_blk = (size_t)std::max(2 * NP * MO2_CW, NP * MO2_DPITCH) * sizeof(float);
  const long per_cu = std::max<long>(1, std::min<long>(CLO_MO2_PER_CU, (160 * 1024) / (long)smem_blk));   // (beyond 128 rows: one block per CU)
  const int cr = (int)std::max<long>(MO2_CW, cdiv(cdiv(strip_cols, per_cu * kNumCU), MO2_CW) * MO2_CW);
  int nb2 = 0;
  for (int k = 0; k < count; ++k) {
    o2.first_block[k] = nb2;
    o2.md[k] = oa.md[k]; o2.a_prev[k] = oa.a_prev[k]; o2.out_W[k] = oa.out_W[k]; o2.out_b[k] = oa.out_b[k];
    o2.d_in[k] = oa.d_in[k]; o2.d_out[k] = oa.d_out[k];
    o2.cr[k] = cr; o2.nranges[k] = (int)cdiv(oa.d_in[k], cr);
    nb2 += (int)cdiv(oa.d_out[k], MO2_ROWS) * o2.nranges[k];
  }
  o2.first_block[count] = nb2;
  const size_t smem2 = (size_t)std::max(2 * NP * MO2_CW, NP * MO2_DPITCH) * sizeof(float);   // chunk double buffer | staged delta strip
  {
    const int rc = beta != 0.f ? set_smem(mid_outer2_kernel<NT, true>, smem2) : set_smem(mid_outer2_kernel<NT, false>, smem2);
    if (rc != CLO_OK) return rc;
  }
  if (beta != 0.f) hipLaunchKernelGGL((mid_outer2_kernel<NT, true>), dim3(nb2), dim3(512), smem2, st, o2);
  else hipLaunchKernelGGL((mid_outer2_kernel<NT, false>), dim3(nb2), dim3(512), smem2, st, o2);
  CLO_CHECK_LAUNCH("mid_outer2_kernel");
  return CLO_OK;
}

// Round 6: the data chain's step delta_{l-1} = delta_l W_l and the outer products that only need delta_l (layer l, and the
// head's layer at the first step) in ONE launch: the read-only sweep of W_l and the write-only stream of out_W_l are
// independent, and as two launches each paid its own boundary, staging prologue and drain with one workgroup per CU
// (C2, 16 rows: 11.6 + 12.3 us).  The data-chain blocks come first (the next step waits for them).
template <int NT, bool ACCUM, int CW>
__global__ __launch_bounds__(512) void mid_bwd_merged_kernel(const MidDprevArgs q, const MidOuterArgs p, int n_dprev,
                                                             int dprev_gx) {
  extern __shared__ __attribute__((aligned(16))) float smem_m[];
  const int b = blockIdx.x;
  if (b < n_dprev) mid_dprev_body<NT>(q, b % dprev_gx, b / dprev_gx, smem_m);
  else mid_outer_body<NT, ACCUM, CW>(p, b - n_dprev, smem_m);
}

// delta_l [N][d] = (sum of the row-range slabs of mid_dprev_kernel) x act'(z_l): beyond 32 batch rows the
// consumers (the next mid_dprev launch and mid_outer_kernel, one block per column chunk each) would
// otherwise re-add the slabs d_in / 256 times over.
__global__ __launch_bounds__(256) void mid_delta_finish_kernel(const float *__restrict__ slabs, int njb,
                                                               long slab_stride, const float *__restrict__ dphi,
                                                               float *__restrict__ out, long total4) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  float4 acc = zero4();
  const float4 dp = ld4(dphi + 4 * e);
  for (int jb = 0; jb < njb; jb += 8) {   // eight slabs in flight: clamped index + zero weight, no branches (the order of the sum is kept)
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ld4(slabs + (long)min(jb + k, njb - 1) * slab_stride + 4 * e);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (jb + k < njb) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
  }
  st4(out + 4 * e, make_float4(acc.x * dp.x, acc.y * dp.y, acc.z * dp.z, acc.w * dp.w));
}

// head_bwd_kernel for more than NB rows, one block per (256-column chunk, NB-row chunk): merge the head
// partials of the chunk, loss Hessian per sample -> delta_L rows (written to dL [N][HEAD_CMAX] for the
// outer-product launch, which also forms out_W_L / out_b_L) and the chunk's rows of delta_{L-1}.
__global__ __launch_bounds__(256) void head_bwd_rows_kernel(const HeadBwdArgs p, float *__restrict__ dL) {
  __shared__ float s_f[NB][HEAD_CMAX], s_u[NB][HEAD_CMAX], s_dl[NB][HEAD_CMAX];
  const int tid = threadIdx.x;
  const int N = p.N, C = p.C;
  const int i = blockIdx.x * 256 + tid;
  const int ic = i < p.d ? i : 0;
  const int n0 = blockIdx.y * NB, nn = min(NB, N - n0);
  float wcol[HEAD_CMAX], dpp[NB];
#pragma unroll
  for (int c = 0; c < HEAD_CMAX; ++c) wcol[c] = c < C ? p.WL[(long)c * p.d + ic] : 0.f;
#pragma unroll
  for (int n = 0; n < NB; ++n) dpp[n] = p.dphi_prev[(long)(n0 + (n < nn ? n : 0)) * p.d + ic];
  {  // merge the head partials of this chunk: thread = (n, which in {z, dz}, c)
    const int n = tid >> 5, r = tid & 31, c = r & (HEAD_CMAX - 1);
    float acc = 0.f;
    if (n < nn && c < C) {
      const float *bias = (r < HEAD_CMAX) ? p.bL : p.VbL;
      acc = bias ? bias[c] : 0.f;
      const float *q = p.hp + ((long)(n0 + n) * p.nblk * 2) * HEAD_CMAX + r;
      for (int k0 = 0; k0 < p.nblk; k0 += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = q[(long)min(k0 + k, p.nblk - 1) * 2 * HEAD_CMAX];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += (k0 + k < p.nblk) ? t[k] : 0.f;
      }
    }
    if (r < HEAD_CMAX) s_f[n][c] = acc; else s_u[n][c] = acc;
  }
  __syncthreads();
  if (tid < NB) {  // loss Hessian per sample
    const int n = tid;
    for (int c = 0; c < HEAD_CMAX; ++c) s_dl[n][c] = 0.f;
    if (n < nn) {
      if (p.kind == CLO_LOSS_MSE) {
        for (int c = 0; c < C; ++c) s_dl[n][c] = p.scale * s_u[n][c];
      } else if (p.kind == CLO_LOSS_BCE) {
        for (int c = 0; c < C; ++c) {
          s_dl[n][c] = p.scale * sigmoid_prime(s_f[n][c]) * s_u[n][c];
        }
      } else if (p.kind == CLO_LOSS_CE) {
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, s_f[n][c]);
        float se = 0.f, spu = 0.f;
        for (int c = 0; c < C; ++c) {
          const float e = __expf(s_f[n][c] - mx);
          se += e;
          spu += e * s_u[n][c];
        }
        const float inv = 1.f / se, pu = spu * inv;
        for (int c = 0; c < C; ++c) {
          const float pc = __expf(s_f[n][c] - mx) * inv;
          s_dl[n][c] = p.scale * pc * (s_u[n][c] - pu);
        }
      } else if (loss_is_ef(p.kind)) {
        float g[HEAD_CMAX];
        ef_grad_row<HEAD_CMAX>(p.kind, &s_f[n][0], ef_target_row(p.kind, p.aux, n0 + n, C), C, g);
        float sdot = 0.f;
        for (int c = 0; c < C; ++c) sdot += g[c] * s_u[n][c];
        for (int c = 0; c < C; ++c) s_dl[n][c] = p.scale * g[c] * sdot;
      } else {
        for (int m = 0; m < p.aux_rank; ++m) {
          const float *g = p.aux + ((long)(n0 + n) * p.aux_rank + m) * C;
          float sdot = 0.f;
          for (int c = 0; c < C; ++c) sdot += g[c] * s_u[n][c];
          for (int c = 0; c < C; ++c) s_dl[n][c] += p.scale * g[c] * sdot;
        }
      }
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid < NB * HEAD_CMAX && (tid >> 4) < nn)
    dL[(long)(n0 + (tid >> 4)) * HEAD_CMAX + (tid & 15)] = s_dl[tid >> 4][tid & 15];
  if (i < p.d && p.delta_prev) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < HEAD_CMAX; ++c) acc = fmaf(s_dl[n][c], wcol[c], acc);
      if (n < nn) p.delta_prev[(long)(n0 + n) * p.d + i] = acc * dpp[n];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Narrow last layer (C <= 16) for batches beyond one 8-row pass: three small kernels instead of
// 128-wide GEMM tiles on a 10-column problem.
// ------------------------------------------------------------------------------------------
constexpr int HR_ROWS = 4;
// f[n][c] = b[c] + a[n] . W[c] ;  u[n][c] = Vb[c] + da[n] . W[c] + a[n] . V[c]
// block = HR_ROWS batch rows, wave w = class c, lanes stride the features.
// grid.y > 1: few rows would leave the chip to a handful of blocks walking all of d, so the
// features are split over grid.y and the raw partial sums go to part[split][2][N][C]
// (loss_hessian_kernel adds them up together with the biases, as it does for the 8-row chain).
template <bool VEC>
__global__ __launch_bounds__(HEAD_CMAX * 64) void head_rows_fwd_kernel(
    const float *__restrict__ a, const float *__restrict__ da, const float *__restrict__ W,
    const float *__restrict__ V, const float *__restrict__ b, const float *__restrict__ Vb,
    float *__restrict__ f, float *__restrict__ u, int N, int d, int C, int d_per_split,
    float *__restrict__ part) {
  const int lane = threadIdx.x & 63, c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (c >= C) return;
  const int n0 = blockIdx.x * HR_ROWS;
  const int i0 = blockIdx.y * d_per_split, i1 = min(d, i0 + d_per_split);
  float z[HR_ROWS], dz[HR_ROWS];
#pragma unroll
  for (int r = 0; r < HR_ROWS; ++r) z[r] = dz[r] = 0.f;
  const float *w = W + (long)c * d, *v = V + (long)c * d;
  if (VEC) {  // d % 4 == 0, 16-byte aligned rows: 4 features per lane and trip
#pragma unroll 2
    for (int i = i0 + lane * 4; i < i1; i += 256) {
      const float4 wi = ld4(w + i), vi = ld4(v + i);
#pragma unroll
      for (int r = 0; r < HR_ROWS; ++r) {
        const long off = (long)min(n0 + r, N - 1) * d + i;
        const float4 x = ld4(a + off), dx = ld4(da + off);
        z[r] = fma4(wi, x, z[r]);
        dz[r] = fma4(wi, dx, fma4(vi, x, dz[r]));
      }
    }
  } else {
#pragma unroll 2
    for (int i = i0 + lane; i < i1; i += 64) {
      const float wi = w[i], vi = v[i];
#pragma unroll
      for (int r = 0; r < HR_ROWS; ++r) {
        const long off = (long)min(n0 + r, N - 1) * d + i;
        const float x = a[off], dx = da[off];
        z[r] = fmaf(wi, x, z[r]);
        dz[r] = fmaf(wi, dx, fmaf(vi, x, dz[r]));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < HR_ROWS; ++r) {
    const float zs = wave_sum(z[r]), dzs = wave_sum(dz[r]);
    if (lane == 0 && n0 + r < N) {
      if (gridDim.y > 1) {
        float *pz = part + ((long)blockIdx.y * 2 * N + (n0 + r)) * C + c;
        pz[0] = zs;
        pz[(long)N * C] = dzs;
      } else {
        f[(long)(n0 + r) * C + c] = zs + (b ? b[c] : 0.f);
        u[(long)(n0 + r) * C + c] = dzs + (Vb ? Vb[c] : 0.f);
      }
    }
  }
}

// delta_prev[n][i] = dphi_prev[n][i] * sum_c dL[n][c] W[c][i]      grid (ceil(d/256), N)
__global__ __launch_bounds__(256) void head_rows_bwd_kernel(const float *__restrict__ dL,
                                                            const float *__restrict__ W,
                                                            const float *__restrict__ dphi_prev,
                                                            float *__restrict__ delta_prev, int d,
                                                            int C) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (i >= d) return;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) acc = fmaf(dL[(long)n * C + c], W[(long)c * d + i], acc);
  delta_prev[(long)n * d + i] = acc * dphi_prev[(long)n * d + i];
}

// Everything behind delta_L of a narrow head, up to HB_MAX_N rows, in ONE launch (round 5: the three outer-product launches
// and head_rows_bwd_kernel took 22 of the 197 us of a 128-row product, each of them a few microseconds of latency):
//   out_W[c][j]      = beta out_W[c][j] + sum_n dL[n][c] a_prev[n][j]            blocks [0, nbj): 64 columns, all rows
//   out_b[c]         = beta out_b[c]    + sum_n dL[n][c]                         (block 0)
//   delta_prev[n][j] = dphi_prev[n][j] * sum_c dL[n][c] W[c][j]                  blocks [nbj, ...): 64 columns x HB_DROWS rows
// Every block has 8 waves = 8 row groups (rows n = grp, grp + 8, ...); dL sits in LDS (broadcast reads), the row groups'
// partial out_W sums are merged through LDS in a fixed order.
constexpr int HB_MAX_N = 256, HB_FUSE_N = 192, HB_GROUPS = 8, HB_DROWS = 32;
__global__ __launch_bounds__(HB_GROUPS * 64) void head_rows_back_kernel(
    const float *__restrict__ dL, const float *__restrict__ a_prev, const float *__restrict__ W,
    const float *__restrict__ dphi_prev, float *__restrict__ out_W, float *__restrict__ out_b,
    float *__restrict__ delta_prev, int N, int d, int C, float beta, int nbj) {
  __shared__ float s_dl[HB_MAX_N * HEAD_CMAX];
  __shared__ float s_acc[HEAD_CMAX][HB_GROUPS][64];
  const int lane = threadIdx.x & 63, grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool outer = (int)blockIdx.x < nbj;
  const int jb = outer ? (int)blockIdx.x : ((int)blockIdx.x - nbj) % nbj, rb = outer ? 0 : ((int)blockIdx.x - nbj) / nbj;
  const int j = jb * 64 + lane, jc = min(j, d - 1);
  const int n0 = outer ? 0 : rb * HB_DROWS, n1 = outer ? N : min(N, n0 + HB_DROWS);
  // (round 6: every global load of the block is in flight before the first wait -- delta_L for LDS in one batch, and in the outer-product
  // blocks all rows of a_prev this wave will use; they used to be 2 - 3 dependent trips for delta_L, then two batches of eight)
  constexpr int DLS = (HB_FUSE_N * HEAD_CMAX + HB_GROUPS * 64 - 1) / (HB_GROUPS * 64);   // staged elements per thread, at most
  constexpr int XR = HB_FUSE_N / HB_GROUPS;                                              // rows per wave, at most
  float dls[DLS];
  {
    const int e0 = n0 * C + (int)threadIdx.x, e1 = n1 * C;
#pragma unroll
    for (int u = 0; u < DLS; ++u) dls[u] = dL[min(e0 + u * HB_GROUPS * 64, max(e1 - 1, 0))];
  }
  float xr[XR];
  if (outer) {
#pragma unroll
    for (int r = 0; r < XR; ++r) xr[r] = a_prev[(long)min(grp + r * HB_GROUPS, N - 1) * d + jc];
  }
  {
    const int e0 = n0 * C + (int)threadIdx.x, e1 = n1 * C;
#pragma unroll
    for (int u = 0; u < DLS; ++u)
      if (e0 + u * HB_GROUPS * 64 < e1) s_dl[e0 + u * HB_GROUPS * 64] = dls[u];
    for (int e = e0 + DLS * HB_GROUPS * 64; e < e1; e += HB_GROUPS * 64) s_dl[e] = dL[e];   // (N beyond HB_FUSE_N: not launched that way)
  }
  if (outer) {
    float acc[HEAD_CMAX];
#pragma unroll
    for (int c = 0; c < HEAD_CMAX; ++c) acc[c] = 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < XR; ++r) {
      const int n = grp + r * HB_GROUPS;
      if (n < N) {   // (wave-uniform)
        const float x = xr[r];
        const float *dn = &s_dl[n * C];
#pragma unroll
        for (int c = 0; c < HEAD_CMAX; ++c)
          if (c < C) acc[c] = fmaf(dn[c], x, acc[c]);
      }
    }
    for (int n = grp + XR * HB_GROUPS; n < N; n += HB_GROUPS) {   // (N beyond HB_FUSE_N)
      const float x = a_prev[(long)n * d + jc];
      const float *dn = &s_dl[n * C];
#pragma unroll
      for (int c = 0; c < HEAD_CMAX; ++c)
        if (c < C) acc[c] = fmaf(dn[c], x, acc[c]);
    }
#pragma unroll
    for (int c = 0; c < HEAD_CMAX; ++c) s_acc[c][grp][lane] = acc[c];
    __syncthreads();
    for (int c = grp; c < C; c += HB_GROUPS) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < HB_GROUPS; ++q) t += s_acc[c][q][lane];
      if (j < d) {
        float *o = out_W + (long)c * d + j;
        *o = (beta != 0.f ? beta * *o : 0.f) + t;
      }
    }
    if (out_b && blockIdx.x == 0 && threadIdx.x < C) {
      float t = 0.f;
      for (int n = 0; n < N; ++n) t += s_dl[n * C + threadIdx.x];
      out_b[threadIdx.x] = (beta != 0.f ? beta * out_b[threadIdx.x] : 0.f) + t;
    }
  } else {
    float w[HEAD_CMAX];
#pragma unroll
    for (int c = 0; c < HEAD_CMAX; ++c) w[c] = c < C ? W[(long)c * d + jc] : 0.f;
    float ph[HB_DROWS / HB_GROUPS];
#pragma unroll
    for (int r = 0; r < HB_DROWS / HB_GROUPS; ++r) ph[r] = dphi_prev[(long)min(n0 + grp + r * HB_GROUPS, N - 1) * d + jc];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < HB_DROWS / HB_GROUPS; ++r) {
      const int n = n0 + grp + r * HB_GROUPS;
      if (n < n1) {
        const float *dn = &s_dl[n * C];
        float sdot = 0.f;
#pragma unroll
        for (int c = 0; c < HEAD_CMAX; ++c)
          if (c < C) sdot = fmaf(dn[c], w[c], sdot);
        if (j < d) delta_prev[(long)n * d + j] = sdot * ph[r];
      }
    }
  }
}

// out[c][j] = beta * out[c][j] + sum_n g[n][c] X[n][j]   (c < C <= 16; g == nullptr: ones, C = 1)
// block = 64 columns x 8 row groups (one wave each), merged through LDS.  Used for the bias
// gradients (column sums of delta) and the narrow last layer's weight block.
constexpr int SO_GROUPS = 8;
// grid = (ceil(d / 64), NS): with NS > 1 row chunks the block writes its partial sums to
// slab[chunk][c][j] and small_outer_reduce_kernel finishes.
template <bool HAS_G>
__global__ __launch_bounds__(SO_GROUPS * 64) void small_outer_kernel(
    float *__restrict__ out, const float *__restrict__ g, const float *__restrict__ X, int N, int d,
    int C, float beta, float *__restrict__ slab) {
  __shared__ float s_acc[HAS_G ? HEAD_CMAX : 1][SO_GROUPS][64];
  const int lane = threadIdx.x & 63, grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = blockIdx.x * 64 + lane, jc = min(j, d - 1);
  const int rows_per = (int)cdiv(N, (int)gridDim.y);
  const int nb = blockIdx.y * rows_per, ne = min(N, nb + rows_per);
  float acc[HAS_G ? HEAD_CMAX : 1];
#pragma unroll
  for (int c = 0; c < (HAS_G ? HEAD_CMAX : 1); ++c) acc[c] = 0.f;
#pragma unroll 4
  for (int n = nb + grp; n < ne; n += SO_GROUPS) {
    const float x = X[(long)n * d + jc];
    if (HAS_G) {
#pragma unroll
      for (int c = 0; c < HEAD_CMAX; ++c)
        if (c < C) acc[c] = fmaf(g[(long)n * C + c], x, acc[c]);
    } else {
      acc[0] += x;
    }
  }
#pragma unroll
  for (int c = 0; c < (HAS_G ? HEAD_CMAX : 1); ++c) s_acc[c][grp][lane] = acc[c];
  __syncthreads();
  for (int c = grp; c < C; c += SO_GROUPS) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < SO_GROUPS; ++q) t += s_acc[c][q][lane];
    if (j < d) {
      if (gridDim.y > 1) {
        slab[((long)blockIdx.y * C + c) * d + j] = t;
      } else {
        float *o = out + (long)c * d + j;
        *o = (beta != 0.f ? beta * *o : 0.f) + t;
      }
    }
  }
}
__global__ void small_outer_reduce_kernel(float *__restrict__ out, const float *__restrict__ slab,
                                          int ns, long cd, float beta) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= cd) return;
  float t = 0.f;
  for (int q = 0; q < ns; ++q) t += slab[q * cd + e];
  out[e] = (beta != 0.f ? beta * out[e] : 0.f) + t;
}

// ------------------------------------------------------------------------------------------
// K probe columns at once (reference: vmap over the trailing K axis, _torch_base.py:946-989).
// The tangent weights keep the reference's K-trailing layout: V_l[j][i][k] at
// V + ((j * d_in + i) * ldk + k), i.e. the rows of a [D, K] matrix, and so does the result.  Per
// column the traffic is then 4 D (V) + 4 D (result); W is shared by all columns:
//   z-path   : a_l, phi'_l once (fwd_mfma kernels without tangent)
//   kfwd     : dA_l[j][n][k] = phi'[n][j] (sum_i V_l[j][i][k] a[n][i] + (W_l dA_{l-1})[j][n][k] + Vb)
//              -- streams V_l once; the W_l dA_{l-1} term is a GEMM with N K columns
//   loss     : per (n, k)
//   kouter   : out_l[j][i][k] = beta out + sum_n a[n][i] delta_l[j][n][k]   -- streams the result
//   dprev    : GEMM delta_{l-1} = phi' * (W_l^T delta_l) with N K columns
// Tangents / deltas are stored feature-major: [d_l][N][K].  N <= 8 rows per pass, K % 4 == 0, K <= 64.
// ------------------------------------------------------------------------------------------
#ifndef CLO_KC_WAVES
#define CLO_KC_WAVES 8
#endif
constexpr int KC_WAVES = CLO_KC_WAVES;   // waves of a kfwd block: split the contraction among themselves
                              // (1 tile x 8 waves measured best among {1,2,3,4} tiles x {1,2,4,8} waves)
#ifndef CLO_KC_TPW
#define CLO_KC_TPW 1
#endif
constexpr int KC_TPW = CLO_KC_TPW;     // MFMA tiles per wave (1 measured best: more, smaller blocks)

// One MFMA tile = 16 "columns" c = (feature f = c / G, column quad kq = c % G), G = K / 4; lane
// (c, s = lane >> 4) loads ONE float4 V[j + f][i][4 kq ..] per i and feeds four MFMAs (one per
// column of the quad), all with the A operand a[n][i].  k-slot s takes i = ib + 4 s + t in step t,
// so the A operand is one float4 of row n per 16 i.
#ifndef CLO_KC_WPE
#define CLO_KC_WPE 1
#endif
template <bool ACC, int TPW, int KW>
__global__ __launch_bounds__(KW * 64, CLO_KC_WPE) void kfwd_stream_kernel(
    const float *__restrict__ V, long ldk, const float *__restrict__ Vb,
    const float *__restrict__ a, const float *__restrict__ dphi, float *__restrict__ dA, int N,
    int K, int d_in, int d_out) {
  // PERSISTENT: the grid is a few blocks per CU and every block walks feature tiles tb, tb + grid, ...  The block's waves
  // take the 16-row groups of the contraction range round robin (group g -> wave g % KW); (tile, group) pairs form ONE
  // flat sequence of steps per wave, software-pipelined over a ring of register groups: the loads of step s + 2 are
  // issued -- ALWAYS, from clamped addresses, no branch around them -- before the MFMAs of step s, also across a tile
  // boundary (the first groups of the next tile are in flight under the LDS merge of this one).  Round 4: with the loads under `if (more)`
  // hipcc joined the two paths at the MFMAs with `s_waitcnt vmcnt(3)`, i.e. it waited for the loads it had just issued;
  // a wave then had one group in flight instead of two.
  __shared__ float s_red[KW][TPW][4][4][64];  // [wave][tile][m][r][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, s = lane >> 4;
  const int G = K >> 2, FPT = 16 / G;        // features per tile
  const int f = c / G, kq = c - f * G;
  const bool cvalid = f < FPT;
  const int wb = 16 * wave, gstep = 16 * KW;
  const int NGW = (int)cdiv(cdiv(d_in, 16), KW);   // groups per wave and tile (the same for every wave: surplus ones masked)
  const unsigned bmask = cvalid ? 0xffffffffu : 0u;
  const int n = c;  // A operand row
  const unsigned amask = n < N ? 0xffffffffu : 0u;
  const float *arow = a + (long)min(n, N - 1) * d_in;
  const float *vcol = V + 4 * (cvalid ? kq : 0);
  const int ntb = (int)cdiv(d_out, FPT * TPW);
  const int ntl = (ntb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this block (grid <= ntb)
  const int nsteps = ntl * NGW;

  struct Group { float4 av; float4 bv[TPW][4]; };
  auto load = [&](Group &g, int tl, int gi) {   // group gi of the block's tile tl; rows beyond d_in read row 0 (masked)
    const int j0 = ((int)blockIdx.x + tl * (int)gridDim.x) * FPT * TPW;
    const int ib = wb + gi * gstep + 4 * s;
    g.av = ld4(arow + (ib + 3 < d_in ? ib : 0));
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int i = ib + st;
      const long off = (long)(i < d_in ? i : 0) * ldk;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int j = min(j0 + t * FPT + (cvalid ? f : 0), d_out - 1);
        g.bv[t][st] = ld4(vcol + ((long)j * d_in) * ldk + off);
      }
    }
  };
  f32x4 acc[TPW][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto mma = [&](const Group &g, int gi, bool live) {
    const unsigned am = (live && wb + gi * gstep + 4 * s + 3 < d_in) ? amask : 0u;
    const float avs[4] = {__uint_as_float(__float_as_uint(g.av.x) & am),
                          __uint_as_float(__float_as_uint(g.av.y) & am),
                          __uint_as_float(__float_as_uint(g.av.z) & am),
                          __uint_as_float(__float_as_uint(g.av.w) & am)};
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const float4 b = g.bv[t][st];
        const float bx = __uint_as_float(__float_as_uint(b.x) & bmask);
        const float by = __uint_as_float(__float_as_uint(b.y) & bmask);
        const float bz = __uint_as_float(__float_as_uint(b.z) & bmask);
        const float bw = __uint_as_float(__float_as_uint(b.w) & bmask);
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(avs[st], bx, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(avs[st], by, acc[t][1], 0, 0, 0);
        acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(avs[st], bz, acc[t][2], 0, 0, 0);
        acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(avs[st], bw, acc[t][3], 0, 0, 0);
      }
  };
  // merge of the waves' K ranges and the tile's epilogue (once per tile)
  auto finish_tile = [&](int tl) {
    const int j0 = ((int)blockIdx.x + tl * (int)gridDim.x) * FPT * TPW;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[wave][t][m][r][lane] = acc[t][m][r];
    __syncthreads();
    // D layout: row n = 4 (lane >> 4) + r, column c.  Wave w finishes r = w: sums the K ranges and
    // writes the quad's four columns as one float4.
    if (cvalid) {
      for (int r = wave; r < 4; r += KW) {
        const int nn = 4 * s + r;
        if (nn >= N) continue;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int j = j0 + t * FPT + f;
          if (j >= d_out) continue;
          float o[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < KW; ++w) v += s_red[w][t][m][r][lane];
            o[m] = v;
          }
          float *dst = dA + ((long)j * N + nn) * K + 4 * kq;
          if (ACC) {
            const float4 old = ld4(dst);
            o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
          }
          if (Vb) {
            const float4 vb = ld4(Vb + (long)j * ldk + 4 * kq);
            o[0] += vb.x; o[1] += vb.y; o[2] += vb.z; o[3] += vb.w;
          }
          const float dp = dphi ? dphi[(long)nn * d_out + j] : 1.f;
          *reinterpret_cast<float4 *>(dst) = make_float4(dp * o[0], dp * o[1], dp * o[2], dp * o[3]);
        }
      }
    }
    __syncthreads();  // s_red is reused by the next tile
  };

  // load cursor (one step ahead of the compute cursor; parked on the last step at the end) and compute cursor
  int lt = 0, lg = 0, ct = 0, cg = 0;
  auto next_load = [&]() {
    const bool wrap = lg + 1 == NGW, last = wrap && lt + 1 == ntl;
    lg = last ? lg : (wrap ? 0 : lg + 1);
    lt = last ? lt : (wrap ? lt + 1 : lt);
  };
  auto step = [&](const Group &cur, int sidx) {
    const bool live = sidx < nsteps;
    mma(cur, cg, live);
    if (live && cg + 1 == NGW) {
      finish_tile(ct);
      zero_acc();
    }
    const bool wrap = cg + 1 == NGW;
    cg = wrap ? 0 : cg + 1;
    ct = wrap ? ct + 1 : ct;
  };
#ifndef CLO_KF_DEPTH
#define CLO_KF_DEPTH 2
#endif
  // register ring of CLO_KF_DEPTH groups: DEPTH - 1 groups of loads are in flight while one feeds the MFMAs.  The
  // sched_barriers pin "issue the loads, THEN the MFMAs": the machine scheduler otherwise sinks the loads into the
  // MFMA sequence to save registers, which shortens the prefetch distance to half a step.
  Group ring[CLO_KF_DEPTH];
  zero_acc();
#pragma unroll
  for (int q = 0; q < CLO_KF_DEPTH - 1; ++q) {
    load(ring[q], lt, lg);
    next_load();
  }
  for (int s0 = 0; s0 < nsteps; s0 += CLO_KF_DEPTH) {
#pragma unroll
    for (int q = 0; q < CLO_KF_DEPTH; ++q) {
      load(ring[(q + CLO_KF_DEPTH - 1) % CLO_KF_DEPTH], lt, lg);
      next_load();
      __builtin_amdgcn_sched_barrier(0);
      step(ring[q], s0 + q);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// aT[i][0..7] = a[n][i] (zero beyond N): the outer-product stream reads 8 samples of one input
// feature as two float4.
// Up to 8 consecutive layers per launch (their aT blocks are consecutive in the workspace): the K-column product packs
// every layer input after its forward loop instead of one tiny launch per layer.
struct PackMulti {
  const float *a[8];
  int d[8];
  long start[9];   // first element of layer q in the concatenated [sum d][NB] block
  int nl, N;
};
__global__ void pack_at_multi_kernel(const PackMulti p, float *__restrict__ aT) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.start[p.nl]) return;
  int q = 0;
#pragma unroll
  for (int t = 1; t < 8; ++t) q += (t < p.nl && e >= p.start[t]) ? 1 : 0;
  const long r = e - p.start[q];
  const int i = (int)(r / NB), n = (int)(r % NB);
  aT[e] = n < p.N ? p.a[q][(long)n * p.d[q] + i] : 0.f;
}

// out[j][i][k] = beta out[j][i][k] + sum_n aT[i][n] delta[j][n][k] ; out_b[j][k] likewise with 1.
// One block per output feature j; lane = (input feature, column quad): a wave instruction stores
// (64 / G) consecutive rows of K floats = one contiguous run of the result.
template <bool ACCUM>
__global__ __launch_bounds__(256) void kouter_stream_kernel(
    float *__restrict__ out, long ldk, float *__restrict__ out_b, const float *__restrict__ aT,
    const float *__restrict__ delta, int N, int K, int d_in, float beta, float beta_b, int nchunk, int chunk_rows) {
  // a block writes ONE chunk of `chunk_rows` consecutive input features of output feature j (16 KB per trip): blocks
  // are dispatched in index order, so the chip writes one moving window instead of a slow stream per output feature
  const int j = blockIdx.x / nchunk, ch = blockIdx.x - j * nchunk;
  const int ibeg = ch * chunk_rows, iend = min(d_in, ibeg + chunk_rows);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = K >> 2, IPW = 64 / G;  // input features per wave instruction
  const int io = lane / G, kq = lane - io * G;
  if (io >= IPW) return;
  float4 d[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
    d[n] = n < N ? ld4(delta + ((long)j * N + n) * K + 4 * kq) : zero4();
  if (out_b && ch == 0 && wave == 0 && io == 0) {
    float4 sb = zero4();
#pragma unroll
    for (int n = 0; n < NB; ++n) { sb.x += d[n].x; sb.y += d[n].y; sb.z += d[n].z; sb.w += d[n].w; }
    float *ob = out_b + (long)j * ldk + 4 * kq;
    if (beta_b != 0.f) {   // (its own beta: the Hessian path accumulates the weights onto a product written before)
      const float4 o = ld4(ob);
      sb.x += beta_b * o.x; sb.y += beta_b * o.y; sb.z += beta_b * o.z; sb.w += beta_b * o.w;
    }
    *reinterpret_cast<float4 *>(ob) = sb;
  }
  float *oj = out + (long)j * d_in * ldk + 4 * kq;
#ifndef CLO_KO_U
#define CLO_KO_U 4
#endif
  constexpr int U = CLO_KO_U;
  const int step = 4 * IPW;  // input features per block trip
  for (int i0 = ibeg + wave * IPW + io; i0 < iend; i0 += U * step) {
    float4 x0[U], x1[U], old[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + u * step, iend - 1);
      x0[u] = ld4(aT + (long)i * NB);
      x1[u] = ld4(aT + (long)i * NB + 4);
      if (ACCUM) old[u] = ld4(oj + (long)i * ldk);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * step;
      if (i >= iend) break;
      const float xs[NB] = {x0[u].x, x0[u].y, x0[u].z, x0[u].w, x1[u].x, x1[u].y, x1[u].z, x1[u].w};
      float4 o = ACCUM ? make_float4(beta * old[u].x, beta * old[u].y, beta * old[u].z, beta * old[u].w)
                       : zero4();
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        o.x = fmaf(xs[n], d[n].x, o.x); o.y = fmaf(xs[n], d[n].y, o.y);
        o.z = fmaf(xs[n], d[n].z, o.z); o.w = fmaf(xs[n], d[n].w, o.w);
      }
      CLO_STW(oj + (long)i * ldk, o);
    }
  }
}

// Tangent of a NARROW last layer (d_out <= 16 outputs) in two launches instead of four (tangent GEMM, its split-K reduce and
// the weight stream, 47 us on 2688 -> 10 at K = 32: the stream kernel deals features to blocks and has ten of them):
//   dA_L[c][n][k] = phi'[n][c] (sum_i V[c][i][k] a[n][i] + W[c][i] dA_{L-1}[i][n][k] + Vb[c][k])
// klast_partial: block g takes KL_ROWS rows i of the contraction, thread = (n, column quad), all d_out outputs in
// registers; the partial [d_out][N][K] goes to slab g.  klast_finish: sums the slabs in a fixed order (16 lanes per output
// quad over the slab index, then an LDS tree), bias, phi'.
constexpr int KL_ROWS = 8, KL_CMAX = 16;
// SLOTS row slots of PP = 256 / SLOTS threads: slot s takes rows i0 + s, i0 + s + SLOTS, ... (KL_ROWS / SLOTS of them, all their
// loads in flight at once), thread = (slot, (n, column quad)); the slots are merged through LDS.
template <int SLOTS>
__global__ __launch_bounds__(256) void klast_partial_kernel(const float *__restrict__ V, long ldk, const float *__restrict__ W,
                                                            const float *__restrict__ a, const float *__restrict__ dAp,
                                                            float *__restrict__ slabs, int N, int K, int d_in, int C) {
  constexpr int PP = 256 / SLOTS, RPT = KL_ROWS / SLOTS;
  __shared__ float4 s_red[SLOTS - 1][KL_CMAX][PP];
  const int G = K >> 2, pairs = N * G;
  const int slot = threadIdx.x / PP, pp = threadIdx.x - slot * PP;
  const bool live = pp < pairs;
  const int pr = live ? pp : 0, n = pr / G, kq = pr - n * G;
  const int i0 = blockIdx.x * KL_ROWS + slot;
  float4 acc[KL_CMAX];
#pragma unroll
  for (int c = 0; c < KL_CMAX; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  float an[RPT];
  float4 t[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {   // rows beyond d_in: row 0 with a zero multiplier
    const int i = i0 + r * SLOTS, ic = i < d_in ? i : 0;
    an[r] = i < d_in ? a[(long)n * d_in + ic] : 0.f;
    t[r] = dAp && i < d_in ? ld4(dAp + ((long)ic * N + n) * K + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int c = 0; c < KL_CMAX; ++c) {
    if (c < C) {
      float4 v[RPT];
      float w[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int i = i0 + r * SLOTS, ic = i < d_in ? i : 0;
        v[r] = ld4(V + ((long)c * d_in + ic) * ldk + 4 * kq);
        w[r] = i < d_in ? W[(long)c * d_in + ic] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        acc[c].x = fmaf(v[r].x, an[r], fmaf(w[r], t[r].x, acc[c].x)); acc[c].y = fmaf(v[r].y, an[r], fmaf(w[r], t[r].y, acc[c].y));
        acc[c].z = fmaf(v[r].z, an[r], fmaf(w[r], t[r].z, acc[c].z)); acc[c].w = fmaf(v[r].w, an[r], fmaf(w[r], t[r].w, acc[c].w));
      }
    }
  }
  if (slot > 0) {
#pragma unroll
    for (int c = 0; c < KL_CMAX; ++c)
      if (c < C) s_red[slot - 1][c][pp] = acc[c];
  }
  __syncthreads();
  if (slot == 0 && live) {
    float *dst = slabs + (long)blockIdx.x * C * N * K + ((long)n * K + 4 * kq);
#pragma unroll
    for (int c = 0; c < KL_CMAX; ++c)
      if (c < C) {
        float4 o = acc[c];
#pragma unroll
        for (int q = 0; q < SLOTS - 1; ++q) { const float4 x = s_red[q][c][pp]; o.x += x.x; o.y += x.y; o.z += x.z; o.w += x.w; }
        *reinterpret_cast<float4 *>(dst + (long)c * N * K) = o;
      }
  }
}
// four output quads per block, 64 lanes over the slab index each
__global__ __launch_bounds__(256) void klast_finish_kernel(const float *__restrict__ slabs, int nslab, const float *__restrict__ Vb,
                                                           long ldk, const float *__restrict__ dphi, float *__restrict__ dA,
                                                           int N, int K, int C) {
  __shared__ float4 s_p[4][64];
  const int G = K >> 2;
  const long quads = (long)C * N * G;
  const int lane_g = threadIdx.x & 63, oq = threadIdx.x >> 6;
  const long q = (long)blockIdx.x * 4 + oq;
  float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < quads) {
    const float *src = slabs + 4 * q;
    const long stride = (long)C * N * K;
    int g = lane_g;
    for (; g + 192 < nslab; g += 256) {   // four loads in flight
      const float4 t0 = ld4(src + (long)g * stride), t1 = ld4(src + (long)(g + 64) * stride);
      const float4 t2 = ld4(src + (long)(g + 128) * stride), t3 = ld4(src + (long)(g + 192) * stride);
      sacc.x += (t0.x + t1.x) + (t2.x + t3.x); sacc.y += (t0.y + t1.y) + (t2.y + t3.y);
      sacc.z += (t0.z + t1.z) + (t2.z + t3.z); sacc.w += (t0.w + t1.w) + (t2.w + t3.w);
    }
    for (; g < nslab; g += 64) {
      const float4 t0 = ld4(src + (long)g * stride);
      sacc.x += t0.x; sacc.y += t0.y; sacc.z += t0.z; sacc.w += t0.w;
    }
  }
  s_p[oq][lane_g] = sacc;
  __syncthreads();
  if (lane_g < 16) {   // 64 -> 16 -> 1, fixed order
    float4 o = s_p[oq][lane_g];
    for (int g = 1; g < 4; ++g) { const float4 x = s_p[oq][lane_g + 16 * g]; o.x += x.x; o.y += x.y; o.z += x.z; o.w += x.w; }
    s_p[oq][lane_g] = o;
  }
  __syncthreads();
  if (lane_g == 0 && q < quads) {
    float4 o = s_p[oq][0];
    for (int g = 1; g < 16; ++g) { const float4 x = s_p[oq][g]; o.x += x.x; o.y += x.y; o.z += x.z; o.w += x.w; }
    const int kq = (int)(q % G), n = (int)((q / G) % N), c = (int)(q / ((long)G * N));
    if (Vb) {
      const float4 vb = ld4(Vb + (long)c * ldk + 4 * kq);
      o.x += vb.x; o.y += vb.y; o.z += vb.z; o.w += vb.w;
    }
    const float dp = dphi ? dphi[(long)n * C + c] : 1.f;
    *reinterpret_cast<float4 *>(dA + 4 * q) = make_float4(dp * o.x, dp * o.y, dp * o.z, dp * o.w);
  }
}

// delta_{L-1}[i][n][k] = phi'_{L-1}[n][i] sum_c W_L[c][i] delta_L[c][n][k] below a narrow last layer (d_out <= 16): a product
// with K = d_out on the GEMM engine pads the contraction to 64 and took 14 us on 2688 x 256 x 10; this is one pass over the
// output.  Thread = (feature i, (n, column quad)); delta_L (d_out x N K floats) is read through L1 by every block.
__global__ __launch_bounds__(256) void klast_delta_kernel(const float *__restrict__ W, const float *__restrict__ dL,
                                                          const float *__restrict__ dphi, float *__restrict__ dprev, int N,
                                                          int K, int d_in, int C) {
  const int Q = (N * K) >> 2;                      // float4 groups per feature
  const long total = (long)d_in * Q;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int i = (int)(e / Q), q = (int)(e - (long)i * Q), n = (4 * q) / K;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < KL_CMAX; ++c)
      if (c < C) {
        const float w = W[(long)c * d_in + i];
        const float4 t = ld4(dL + (long)c * N * K + 4 * q);
        acc.x = fmaf(w, t.x, acc.x); acc.y = fmaf(w, t.y, acc.y); acc.z = fmaf(w, t.z, acc.z); acc.w = fmaf(w, t.w, acc.w);
      }
    const float dp = dphi[(long)n * d_in + i];
    *reinterpret_cast<float4 *>(dprev + 4 * e) = make_float4(dp * acc.x, dp * acc.y, dp * acc.z, dp * acc.w);
  }
}

// delta_L[c][n][k] = phi'_L[n][c] * scale * (H(f_n) u[:, n, k])[c]  in place on u = dA_L.
// One thread per (n, k).
constexpr int LC_RMAX = 16;  // rank-M curvature: at most 16 backpropagated vectors per sample
__global__ void loss_cols_kernel(int kind, const float *__restrict__ f,
                                 const float *__restrict__ aux, int aux_rank,
                                 const float *__restrict__ dphi_last, float *__restrict__ u, int N,
                                 int K, int C, float scale) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * K) return;
  const int n = e / K;
  const long cs = (long)N * K;  // stride between classes
  float *un = u + e;
  const float *fn = f + (long)n * C;
  const float *dp = dphi_last ? dphi_last + (long)n * C : nullptr;
  if (loss_is_ef(kind)) {   // empirical Fisher from the targets: g from (f, target) once per thread, then w = scale g <g, u>
    const float *t = ef_target_row(kind, aux, n, C);
    float mx = -INFINITY, inv = 0.f;
    if (kind == CLO_LOSS_EF_CE) {
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, fn[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += __expf(fn[c] - mx);
      inv = 1.f / se;
    }
    float sdot = 0.f;
    for (int c = 0; c < C; ++c) sdot += ef_grad_at(kind, fn[c], t, c, mx, inv) * un[c * cs];
    sdot *= scale;
    for (int c = 0; c < C; ++c) un[c * cs] = ef_grad_at(kind, fn[c], t, c, mx, inv) * sdot * (dp ? dp[c] : 1.f);
    return;
  }
  if (C <= 16 && kind != CLO_LOSS_RANK1) {
    // narrow output: everything in registers, all loads issued before the first dependent use
    float uv[16], fv[16], dv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int cc = c < C ? c : 0;
      uv[c] = un[cc * cs];
      fv[c] = fn[cc];
      dv[c] = dp ? dp[cc] : 1.f;
    }
    float mx = -INFINITY, se = 0.f, spu = 0.f;
    if (kind == CLO_LOSS_CE) {
#pragma unroll
      for (int c = 0; c < 16; ++c) if (c < C) mx = fmaxf(mx, fv[c]);
#pragma unroll
      for (int c = 0; c < 16; ++c) if (c < C) {
        fv[c] = __expf(fv[c] - mx);
        se += fv[c];
        spu += fv[c] * uv[c];
      }
    }
    const float inv = kind == CLO_LOSS_CE ? 1.f / se : 0.f, pu = spu * inv;
#pragma unroll
    for (int c = 0; c < 16; ++c) if (c < C) {
      float w;
      if (kind == CLO_LOSS_MSE) w = uv[c];
      else if (kind == CLO_LOSS_BCE) { w = sigmoid_prime(fv[c]) * uv[c]; }
      else w = fv[c] * inv * (uv[c] - pu);
      un[c * cs] = scale * w * dv[c];
    }
    return;
  }
  if (kind == CLO_LOSS_MSE) {
    for (int c = 0; c < C; ++c) un[c * cs] = scale * un[c * cs] * (dp ? dp[c] : 1.f);
  } else if (kind == CLO_LOSS_BCE) {
    for (int c = 0; c < C; ++c) {
      un[c * cs] = scale * sigmoid_prime(fn[c]) * un[c * cs] * (dp ? dp[c] : 1.f);
    }
  } else if (kind == CLO_LOSS_CE) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, fn[c]);
    float se = 0.f, spu = 0.f;
    for (int c = 0; c < C; ++c) {
      const float ex = __expf(fn[c] - mx);
      se += ex;
      spu += ex * un[c * cs];
    }
    const float inv = 1.f / se, pu = spu * inv;
    for (int c = 0; c < C; ++c) {
      const float pc = __expf(fn[c] - mx) * inv;
      un[c * cs] = scale * pc * (un[c * cs] - pu) * (dp ? dp[c] : 1.f);
    }
  } else {  // rank-M: H_n = sum_m g_nm g_nm^T, M <= LC_RMAX: all <g_m, u> first, then overwrite u
    float sd[LC_RMAX];
#pragma unroll
    for (int m = 0; m < LC_RMAX; ++m) {
      sd[m] = 0.f;
      if (m < aux_rank) {
        const float *g = aux + ((long)n * aux_rank + m) * C;
        float sdot = 0.f;
        for (int c = 0; c < C; ++c) sdot += g[c] * un[c * cs];
        sd[m] = scale * sdot;
      }
    }
    for (int c = 0; c < C; ++c) {
      float w = 0.f;
#pragma unroll
      for (int m = 0; m < LC_RMAX; ++m)
        if (m < aux_rank) w += aux[((long)n * aux_rank + m) * C + c] * sd[m];
      un[c * cs] = w * (dp ? dp[c] : 1.f);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Exact Hessian-vector product of an MLP (hessian.py:13-69) by the R-operator: next to the tangent
// forward pass (a_l, da_l, phi'_l) it backpropagates the true gradient signal d_l = dL/dz_l AND its
// directional derivative Rd_l:
//   dA      = d_l W_l                       (signal at the previous activation)
//   T       = Rd_l W_l + d_l V_l
//   d_{l-1}  = phi'_{l-1} * dA
//   Rd_{l-1} = (phi'' dz)_{l-1} * dA + phi'_{l-1} * T,   phi'' dz from (a, da): 0 for ReLU / identity,
//             -2 a da for tanh, (1 - 2a) da for the sigmoid
//   out_W_l = beta out_W_l + Rd_l^T a_{l-1} + d_l^T da_{l-1},   out_b_l = beta out_b_l + sum_n Rd_l
// All products run on the GEMM engine; this kernel is the elementwise combine.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_second_times_dz(int act, float a, float da) {
  if (act == CLO_ACT_TANH) return -2.f * a * da;
  if (act == CLO_ACT_SIGMOID) return (1.f - 2.f * a) * da;
  return 0.f;
}
// d = dphi * dA ; Rd = act''dz * dA + dphi * T      (T == nullptr: treated as given in Rd itself)
__global__ void hess_combine_kernel(const float *__restrict__ dA, const float *__restrict__ T,
                                    const float *__restrict__ a, const float *__restrict__ da,
                                    const float *__restrict__ dphi, float *__restrict__ d,
                                    float *__restrict__ Rd, long n, int act,
                                    const float *__restrict__ T2 = nullptr) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long)gridDim.x * blockDim.x) {
    const float g = dA[e], ph = dphi[e];
    const float t = (T ? T[e] : Rd[e]) + (T2 ? T2[e] : 0.f);
    Rd[e] = act_second_times_dz(act, a[e], da[e]) * g + ph * t;
    d[e] = ph * g;
  }
}
// The same with the three products (d W, Rd W, d V) still spread over the row-range slabs of
// bwd_fused_kernel (P[jb][NB][d_in]; njb == 0: a final [N][d_in] array): the slab sums are folded
// into the combine instead of three bwd_finish launches.
struct HessSlabs {
  const float *dA, *T1, *T2;
  int n_dA, n_T1, n_T2;
};
__device__ __forceinline__ float slab_sum(const float *P, int njb, int n, int i, int d_in) {
  const float *p = P + (long)n * d_in + i;
  if (njb == 0) return p[0];
  const long stride = (long)NB * d_in;
  float s0 = 0.f, s1 = 0.f;
  int jb = 0;
  for (; jb + 1 < njb; jb += 2) { s0 += p[jb * stride]; s1 += p[(jb + 1) * stride]; }
  if (jb < njb) s0 += p[jb * stride];
  return s0 + s1;
}
__global__ void hess_combine_slabs_kernel(const HessSlabs hs, const float *__restrict__ a,
                                          const float *__restrict__ da, const float *__restrict__ dphi,
                                          float *__restrict__ d, float *__restrict__ Rd, int N, int d_in,
                                          int act) {
  const int total = N * d_in;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int n = e / d_in, i = e % d_in;
    const float g = slab_sum(hs.dA, hs.n_dA, n, i, d_in);
    const float t = slab_sum(hs.T1, hs.n_T1, n, i, d_in) + slab_sum(hs.T2, hs.n_T2, n, i, d_in);
    const float ph = dphi[e];
    Rd[e] = act_second_times_dz(act, a[e], da[e]) * g + ph * t;
    d[e] = ph * g;
  }
}
__global__ void fill_kernel(float *__restrict__ y, long n, float v) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) y[e] = v;
}
__global__ void mask_scale_kernel(float *__restrict__ y, const float *__restrict__ x,
                                  const float *__restrict__ m, long n, float s) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long)gridDim.x * blockDim.x)
    y[e] = s * x[e] * m[e];
}
__global__ void scale_copy_kernel(float *__restrict__ y, const float *__restrict__ x, long n, float s) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long)gridDim.x * blockDim.x)
    y[e] = s * x[e];
}

static inline unsigned ew_grid(long n) {
  return (unsigned)std::max<long>(1, std::min<long>(cdiv(n, 256), kNumCU * 8L));
}

// ---- launch geometry ----------------------------------------------------------------------
static int fwd_ksplit(int d_in, int d_out) {
  const long row_blocks = cdiv(d_out, FWD_ROWS);
  const long nchunks = cdiv(d_in, KC);
  if (row_blocks >= 96 || nchunks <= 1) return 1;
  return (int)std::min<long>(nchunks, cdiv(kNumCU, row_blocks));
}
// MFMA kernel: a block covers MF_WAVES * MF_RG * 8 features and one K range.  Aim at >= 2 blocks
// per CU; K ranges are multiples of 32 and at most MF_KB_MAX (LDS).
constexpr int MF_RG = 2;
constexpr int MF1_WAVES = 8, MF1_RG = 2, MF1_U = 4;  // fwd_mfma_first_kernel
static int mfma_kpb(int d_in, int d_out) {
  const long row_blocks = cdiv(d_out, MF_WAVES * MF_RG * 8);
#ifndef CLO_MF_BPC
#define CLO_MF_BPC 2
#endif
  long ksplit = std::max<long>(1, cdiv(CLO_MF_BPC * kNumCU, row_blocks));
  ksplit = std::min<long>(ksplit, std::max<long>(1, d_in / 64));
  ksplit = std::min<long>(ksplit, 16);  // slab merges stay short (narrow layers are tiny anyway)
  long kpb = cdiv(cdiv(d_in, ksplit), 32) * 32;
  kpb = std::min<long>(kpb, MF_KB_MAX);
  return (int)kpb;
}
static int mfma_ksplit(int d_in, int d_out) { return (int)cdiv(d_in, mfma_kpb(d_in, d_out)); }
static int bwd_jb(int d_in, int d_out, bool dprev) {
  const long cchunks = cdiv(d_in, CW);
  long jb = cdiv(kNumCU, cchunks);                    // ~1 block per CU
  jb = std::min<long>(jb, cdiv(d_out, 64));           // >= 64 rows (8 per wave) per block
#ifndef CLO_DPREV_JB
#define CLO_DPREV_JB 24
#endif
  if (dprev) jb = std::min<long>(jb, CLO_DPREV_JB);   // bound the slab traffic (32 -> 24: -1 us on C2)
  jb = std::max<long>(jb, cdiv(d_out, 1024));         // <= 1024 rows of delta in LDS
  return (int)std::max<long>(1, jb);
}

static bool vec_ok(int d, std::initializer_list<const void *> ptrs) {
  if (d % 4 != 0) return false;
  for (const void *p : ptrs)
    if (p && !aligned16(p)) return false;
  return true;
}

template <typename K>
static int set_smem(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    return check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                     "hipFuncSetAttribute(smem)");
  }
  return CLO_OK;
}

// One forward pass over <= 8 batch rows.
static int fwd_pass(const float *W, const float *b, const float *VW, const float *Vb,
                    const float *a_in, const float *da_in, float *a_out, float *da_out,
                    float *dphi_out, int N, int d_in, int d_out, int act, float *part,
                    bool leave_partials, int *ksplit_out, hipStream_t st) {
  const bool has_v = VW != nullptr, has_da = da_in != nullptr;
  const bool vec = vec_ok(d_in, {W, VW, a_in, da_in});
  // (without tangent weights only where a wave's K range is one round trip: d_in = 2688 measured 16.3 us against 9.0 + 4.9
  // for the slab kernel + finish, d_in = 1024 9.6 against 13.9)
  if (vec && N >= 1 && !has_da && (has_v || (!leave_partials && d_in <= 1024)) && d_in >= 256 &&
      cdiv(d_out, MF1_RG * 8) >= kNumCU / 2) {
    // ---- first layer: in-block split-K, no slabs, no finish launch
    if (ksplit_out) *ksplit_out = 1;
    const int kpw = (int)cdiv(cdiv(d_in, MF1_WAVES), 16) * 16;
    // features per block: fill the CUs evenly (11 for d_out = 2688), at most one RG x 8 tile group
    const int fpb = (int)std::min<long>(MF1_RG * 8, std::max<long>(4, cdiv(d_out, kNumCU)));
    ProfScope prof(0, 4.0 * d_in * d_out * (has_v ? 2 : 1), st);
    // all of a wave's weight loads in ONE round trip where its K range is 8 steps (d_in = 1024)
    if (kpw == 128 && d_in % 128 == 0)
      hipLaunchKernelGGL((fwd_mfma_first_kernel<MF1_WAVES, MF1_RG, 8>), dim3((unsigned)cdiv(d_out, fpb)),
                         dim3(MF1_WAVES * 64), 0, st, W, b, VW, Vb, a_in, a_out, has_v ? da_out : nullptr, dphi_out, N, d_in,
                         d_out, act, kpw, fpb);
    else
      hipLaunchKernelGGL((fwd_mfma_first_kernel<MF1_WAVES, MF1_RG, MF1_U>), dim3((unsigned)cdiv(d_out, fpb)),
                         dim3(MF1_WAVES * 64), 0, st, W, b, VW, Vb, a_in, a_out, has_v ? da_out : nullptr, dphi_out, N, d_in,
                         d_out, act, kpw, fpb);
    CLO_CHECK_LAUNCH("fwd_mfma_first_kernel");
    return CLO_OK;
  }
  if (vec && N >= 1 && d_in >= 16) {
    // ---- MFMA path
    int kpb = mfma_kpb(d_in, d_out);
    int ksplit = (int)cdiv(d_in, kpb);
    if (!part && ksplit > 1) {  // caller gave no slab workspace: single K range per block
      set_error("clo_mlp_fwd: split-K needs a workspace");
      return CLO_EINVAL;
    }
    if (ksplit_out) *ksplit_out = ksplit;
    dim3 grid((unsigned)cdiv(d_out, MF_WAVES * MF_RG * 8), (unsigned)ksplit), block(MF_WAVES * 64);
    const size_t smem = (size_t)16 * (kpb + 4) * sizeof(float);
    {
      ProfScope prof(0, 4.0 * d_in * d_out * (has_v ? 2 : 1), st);
#define CLO_MF(HV, HD)                                                                            \
  hipLaunchKernelGGL((fwd_mfma_kernel<MF_RG, HV, HD>), grid, block, smem, st, W, b, VW, Vb, a_in, \
                     da_in, a_out, da_out, dphi_out, part, N, d_in, d_out, act, kpb)
      if (has_v && has_da) CLO_MF(true, true);
      else if (has_v) CLO_MF(true, false);
      else if (has_da) CLO_MF(false, true);
      else CLO_MF(false, false);
#undef CLO_MF
      CLO_CHECK_LAUNCH("fwd_mfma_kernel");
    }
    if (ksplit > 1 && !leave_partials) {
      ProfScope pf(3, 0.0, st);
      hipLaunchKernelGGL(fwd_finish_kernel, dim3(ew_grid((long)N * d_out)), dim3(256), 0, st, part,
                         ksplit, b, Vb, a_out, (has_v || has_da) ? da_out : nullptr, dphi_out, N,
                         d_out, act);
      CLO_CHECK_LAUNCH("fwd_finish_kernel");
    }
    return CLO_OK;
  }
  int ksplit = part ? fwd_ksplit(d_in, d_out) : 1;
  const int nchunks = (int)cdiv(d_in, KC);
  const int cps = (int)cdiv(nchunks, ksplit);
  ksplit = (int)cdiv(nchunks, cps);
  if (ksplit_out) *ksplit_out = ksplit;
  dim3 grid((unsigned)cdiv(d_out, FWD_ROWS), (unsigned)ksplit), block(512);
  const size_t smem = (size_t)(has_da ? 4 : 2) * NB * KC * sizeof(float);
  // algorithmic bytes of this launch: W (+ VW) read once
  ProfScope prof(0, 4.0 * d_in * d_out * (has_v ? 2 : 1), st);
#define CLO_FWD(V, HV, HD)                                                                    \
  hipLaunchKernelGGL((fwd_jvp_kernel<V, HV, HD>), grid, block, smem, st, W, b, VW, Vb, a_in,  \
                     da_in, a_out, da_out, dphi_out, part, N, d_in, d_out, act, cps)
  if (vec) {
    if (has_v && has_da) CLO_FWD(true, true, true);
    else if (has_v) CLO_FWD(true, true, false);
    else if (has_da) CLO_FWD(true, false, true);
    else CLO_FWD(true, false, false);
  } else {
    if (has_v && has_da) CLO_FWD(false, true, true);
    else if (has_v) CLO_FWD(false, true, false);
    else if (has_da) CLO_FWD(false, false, true);
    else CLO_FWD(false, false, false);
  }
#undef CLO_FWD
  CLO_CHECK_LAUNCH("fwd_jvp_kernel");
  if (prof.on) { prof_end(st); prof.on = false; }
  if (ksplit > 1 && !leave_partials) {
    ProfScope pf(3, 0.0, st);
    hipLaunchKernelGGL(fwd_finish_kernel, dim3(ew_grid((long)N * d_out)), dim3(256), 0, st, part,
                       ksplit, b, Vb, a_out, (has_v || has_da) ? da_out : nullptr, dphi_out, N,
                       d_out, act);
    CLO_CHECK_LAUNCH("fwd_finish_kernel");
  }
  return CLO_OK;
}

static int bwd_pass(const float *W, const float *delta, const float *a_prev,
                    const float *dphi_prev, float *out_W, float *out_b, float *delta_prev,
                    float alpha, float beta, int N, int d_in, int d_out, float *ws,
                    hipStream_t st, int *leave_slabs_njb = nullptr) {
  const bool outer = out_W != nullptr, dprev = delta_prev != nullptr;
  if (!outer && !dprev) return CLO_OK;
  const int JB = bwd_jb(d_in, d_out, dprev);
  const int rpb = (int)cdiv(d_out, JB);
  const int JBe = (int)cdiv(d_out, rpb);
  const size_t smem = ((size_t)rpb * NB + (dprev ? BWD_WAVES * NB * CW : NB * CW)) * sizeof(float);
  dim3 grid((unsigned)cdiv(d_in, CW), (unsigned)JBe), block(512);
  float *dst = dprev ? (JBe == 1 ? delta_prev : ws) : nullptr;
  const int fin = JBe == 1 ? 1 : 0;
  const bool vec = vec_ok(d_in, {W, a_prev, out_W});
  int rc = CLO_OK;
  // algorithmic bytes: out_W written once (+ read when accumulating), W read once for delta_prev
  ProfScope prof(2, 4.0 * d_in * d_out * ((outer ? (beta != 0.f ? 2 : 1) : 0) + (dprev ? 1 : 0)), st);
#define CLO_BWD(V, O, D, A)                                                                     \
  do {                                                                                          \
    rc = set_smem(bwd_fused_kernel<V, O, D, A>, smem);                                          \
    if (rc != CLO_OK) return rc;                                                                \
    hipLaunchKernelGGL((bwd_fused_kernel<V, O, D, A>), grid, block, smem, st, W, delta, a_prev, \
                       dphi_prev, out_W, out_b, dst, alpha, beta, N, d_in, d_out, rpb, fin);    \
  } while (0)
#define CLO_BWD2(V, O, D)                  \
  do {                                     \
    if (outer && beta != 0.f) CLO_BWD(V, O, D, true); \
    else CLO_BWD(V, O, D, false);          \
  } while (0)
  if (vec) {
    if (outer && dprev) CLO_BWD2(true, true, true);
    else if (outer) CLO_BWD2(true, true, false);
    else CLO_BWD(true, false, true, false);
  } else {
    if (outer && dprev) CLO_BWD2(false, true, true);
    else if (outer) CLO_BWD2(false, true, false);
    else CLO_BWD(false, false, true, false);
  }
#undef CLO_BWD2
#undef CLO_BWD
  CLO_CHECK_LAUNCH("bwd_fused_kernel");
  if (prof.on) { prof_end(st); prof.on = false; }
  if (leave_slabs_njb) {  // the consumer sums the slabs itself (0: delta_prev is already final)
    *leave_slabs_njb = (dprev && JBe > 1) ? JBe : 0;
    return CLO_OK;
  }
  if (dprev && JBe > 1) {
    ProfScope pf(3, 0.0, st);
    hipLaunchKernelGGL(bwd_finish_kernel, dim3(ew_grid((long)N * d_in)), dim3(256), 0, st, ws,
                       dphi_prev, delta_prev, N, d_in, JBe);
    CLO_CHECK_LAUNCH("bwd_finish_kernel");
  }
  return CLO_OK;
}

constexpr int LAYER_MAX_N = 16;  // the per-layer entry points run up to two 8-row passes
constexpr int SKINNY_MAX_N = 8;  // whole-network matvec: one 8-row streaming pass; above that the
                                 // GEMM engine (32-row tiles, fused forward) is faster (measured)

static long gemm_ws_floats(int N, int dmax) {
  // split-K partial slabs for the widest product of the large-batch path; the 9 ... 64-row chain carves its forward slabs
  // (<= 16 x 2 x 64 rows) and two sets of <= 24 row-range slabs of delta out of the same area
  return 48L * (long)std::max(N, 128) * dmax;
}

// out[c][j] = beta out + sum_n g[n][c] X[n][j] (g == nullptr: column sums, C = 1); ws >= 16 C d
static int launch_small_outer(float *out, const float *g, const float *X, int N, int d, int C,
                              float beta, float *ws, long ws_floats, hipStream_t st) {
  int ns = (int)std::min<long>(16, cdiv(N, 64));
  if (!ws || ws_floats < (long)ns * C * d) ns = 1;
  dim3 grid((unsigned)cdiv(d, 64), (unsigned)ns), block(SO_GROUPS * 64);
  if (g) hipLaunchKernelGGL(small_outer_kernel<true>, grid, block, 0, st, out, g, X, N, d, C, beta, ws);
  else hipLaunchKernelGGL(small_outer_kernel<false>, grid, block, 0, st, out, g, X, N, d, C, beta, ws);
  CLO_CHECK_LAUNCH("small_outer_kernel");
  if (ns > 1) {
    const long cd = (long)C * d;
    hipLaunchKernelGGL(small_outer_reduce_kernel, dim3((unsigned)cdiv(cd, 256)), dim3(256), 0, st, out,
                       ws, ns, cd, beta);
    CLO_CHECK_LAUNCH("small_outer_reduce_kernel");
  }
  return CLO_OK;
}

static GemmArgs gemm_problem(int M, int N, int K, const float *A, long sa_m, long sa_k,
                             const float *B, long sb_k, long sb_n, float beta, float *C, long ldc) {
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.alpha = 1.f; g.beta = beta;
  g.A = A; g.sa_m = sa_m; g.sa_k = sa_k;
  g.B = B; g.sb_k = sb_k; g.sb_n = sb_n;
  g.C = C; g.ldc = ldc;
  return g;
}

static int launch_loss(int kind, const float *f, const float *aux, int aux_rank, const float *u,
                       const float *dphi_last, float *w, int N, int C, float scale,
                       const float *part, int ksplit, const float *b, const float *Vb, float *f_out,
                       float *u_out, hipStream_t st, int part_rows = NB) {
  LossArgs a{};
  a.kind = kind; a.f = f; a.aux = aux; a.aux_rank = aux_rank; a.u = u; a.dphi_last = dphi_last;
  a.w = w; a.C = C; a.scale = scale; a.part = part; a.ksplit = ksplit; a.part_rows = part_rows;
  a.b = b; a.Vb = Vb;
  a.f_out = f_out; a.u_out = u_out;
  ProfScope prof(1, 0.0, st);
  hipLaunchKernelGGL(loss_hessian_kernel, dim3(N), dim3(C <= 64 ? 64 : 256), 0, st, a);
  CLO_CHECK_LAUNCH("loss_hessian_kernel");
  return CLO_OK;
}

// mlp_mega.hip: the <= 8-row matvec of a three-layer net in one persistent launch
bool mega_shape_ok(int L, const int *dims, int N);
bool mega_ok(int L, const int *dims, const float *const *W, const float *const *VW, float *const *OW,
             const float *X, int N, int loss_kind, int aux_rank);
int mega_launch(const int *dims, const int *acts, const float *const *W, const float *const *b,
                const float *const *VW, const float *const *Vb, float *const *OW, float *const *Ob,
                const float *X, int N, int loss_kind, const float *aux, int aux_rank, float scale,
                float beta, float *xch, unsigned *sync, hipStream_t st);
long mega_xch_floats(int d1, int d2);
long mega_sync_words();
long mega_debug_floats();

}  // namespace clo

using namespace clo;

extern "C" long clo_mlp_fwd_ws_floats(int N, int d_in, int d_out) {
  (void)N;
  const long ks = std::max(fwd_ksplit(d_in, d_out), mfma_ksplit(d_in, d_out));
  return ks * 2 * NB * d_out + 64;
}
extern "C" long clo_mlp_bwd_ws_floats(int N, int d_in, int d_out) {
  (void)N;
  return (long)bwd_jb(d_in, d_out, true) * NB * d_in + 64;
}

extern "C" int clo_mlp_fwd_jvp_layer(const float *W, const float *b, const float *VW,
                                     const float *Vb, const float *a_in, const float *da_in,
                                     float *a_out, float *da_out, float *dphi_out, int N, int d_in,
                                     int d_out, int act, float *ws, void *stream) {
  CLO_REQUIRE(N >= 0 && d_in > 0 && d_out > 0, "clo_mlp_fwd_jvp_layer: bad sizes");
  CLO_REQUIRE(act >= 0 && act <= 3, "clo_mlp_fwd_jvp_layer: unknown activation %d", act);
  CLO_REQUIRE(W && a_in && a_out, "clo_mlp_fwd_jvp_layer: null operand");
  CLO_REQUIRE(!(VW || da_in) || da_out, "clo_mlp_fwd_jvp_layer: da_out required for a JVP");
  CLO_REQUIRE(N <= LAYER_MAX_N, "clo_mlp_fwd_jvp_layer: N=%d > %d, use clo_mlp_ggn_matvec", N,
              LAYER_MAX_N);
  hipStream_t st = (hipStream_t)stream;
  for (int n0 = 0; n0 < N; n0 += NB) {
    const int nn = std::min(NB, N - n0);
    int rc = fwd_pass(W, b, VW, Vb, a_in + (long)n0 * d_in,
                      da_in ? da_in + (long)n0 * d_in : nullptr, a_out + (long)n0 * d_out,
                      da_out ? da_out + (long)n0 * d_out : nullptr,
                      dphi_out ? dphi_out + (long)n0 * d_out : nullptr, nn, d_in, d_out, act, ws,
                      false, nullptr, st);
    if (rc != CLO_OK) return rc;
  }
  return CLO_OK;
}

extern "C" int clo_loss_hessian_apply(int kind, const float *f, const float *aux, int aux_rank,
                                      const float *u, const float *dphi_last, float *w, int N,
                                      int C, float scale, void *stream) {
  CLO_REQUIRE(kind >= 0 && kind <= CLO_LOSS_EF_BCE, "clo_loss_hessian_apply: unknown kind %d", kind);
  CLO_REQUIRE(!loss_is_ef(kind) || aux, "clo_loss_hessian_apply: EF_* needs the targets in aux");
  CLO_REQUIRE(N >= 0 && C > 0, "clo_loss_hessian_apply: bad sizes");
  if (N == 0) return CLO_OK;
  CLO_REQUIRE(f && u && w, "clo_loss_hessian_apply: null operand");
  CLO_REQUIRE(kind != CLO_LOSS_RANK1 || (aux && aux_rank >= 1),
              "clo_loss_hessian_apply: RANK1 needs aux and aux_rank >= 1");
  return launch_loss(kind, f, aux, aux_rank, u, dphi_last, w, N, C, scale, nullptr, 0, nullptr,
                     nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int clo_mlp_bwd_layer(const float *W, const float *delta, const float *a_prev,
                                 const float *dphi_prev, float *out_W, float *out_b,
                                 float *delta_prev, float alpha, float beta, int N, int d_in,
                                 int d_out, float *ws, void *stream) {
  CLO_REQUIRE(N >= 0 && d_in > 0 && d_out > 0, "clo_mlp_bwd_layer: bad sizes");
  CLO_REQUIRE(N <= LAYER_MAX_N, "clo_mlp_bwd_layer: N=%d > %d, use clo_mlp_ggn_matvec", N,
              LAYER_MAX_N);
  CLO_REQUIRE(delta && (!out_W || a_prev), "clo_mlp_bwd_layer: null operand");
  CLO_REQUIRE(!delta_prev || (W && dphi_prev && ws),
              "clo_mlp_bwd_layer: delta_prev needs W, dphi_prev, ws");
  hipStream_t st = (hipStream_t)stream;
  int n0 = 0;
  do {
    const int nn = std::max(0, std::min(NB, N - n0));
    int rc = bwd_pass(W, delta + (long)n0 * d_out, a_prev ? a_prev + (long)n0 * d_in : nullptr,
                      dphi_prev ? dphi_prev + (long)n0 * d_in : nullptr, out_W, out_b,
                      delta_prev ? delta_prev + (long)n0 * d_in : nullptr, alpha,
                      n0 == 0 ? beta : 1.f, nn, d_in, d_out, ws, st);
    if (rc != CLO_OK) return rc;
    n0 += NB;
  } while (n0 < N);
  return CLO_OK;
}

// ------------------------------------------------------------------------------------------
// 9 ... 64 rows, narrow linear head, float4-complete layers: the MFMA streaming chain (mid_* kernels).
// Scratch comes out of the GEMM slab area `gws` (unused on this path): forward slabs | head partials |
// two regions of delta slabs (ping-pong over the layers).
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Round 6: layers 1 and 2 of the forward + JVP pass of the 9 ... 64-row chain in ONE persistent launch
// (three-layer nets with a narrow head -- BASELINE config C2's class; reference ggn.py:61-66, the jvp half).
// As separate launches (mid_full_kernel, mid_fwd_kernel) the two weight streams of 22 + 58 MB each paid a
// launch boundary, a staging prologue and a drain with one workgroup per CU (C2, 16 / 32 / 64 rows:
// 10 + 15 / 15 + 23 / 21 + 40 us).  Here 256 workgroups form the 16 x 16 grid of csrc/mlp_mega.hip over layer 2:
// workgroup (fb, kb) first computes its <= 16 layer-1 features of K range kb for all rows (in-block split-K over the
// eight waves, as mid_full_kernel), publishes them write-through, and meets the 15 other workgroups of its COLUMN
// group on one counter -- the only seam; the W2 / V2 fragments of its tile are requested before it polls.  Then
// the [a1 ; da1] operand of the K range is staged in LDS and the tile's partial z2 / dz2 (three products per step)
// go to the slab of split kb: exactly what mid_fwd_kernel with 16 K ranges would write, so head_fwd_kernel finishes
// the layer as before.  Counters are monotonic (target = 16 x launch number, the launch number from a 64-bit count of
// finished workgroups), every spin is bounded; a timeout marks the slab with NaN and raises the device's fault word.
// ------------------------------------------------------------------------------------------
namespace clo {
constexpr int MFU_G = 256, MFU_T = 512, MFU_MAXS = 11, MFU_KR = 16 * MFU_MAXS, MFU_MAXCH = 11;
constexpr int MFU_SYNC_WORDS = 32 * 20 + 4096;   // (+ timing slots of -DCLO_MFU_TIMING builds)   // line 0: finished-workgroup count (64 bit); line 1: abort word; lines 2 .. 17: column groups

struct MidFusedArgs {
  const float *W1, *b1, *V1, *Vb1, *X, *W2, *V2;
  float *a1, *da1, *dphi1;   // [N][d1]
  float *part;               // [16][2][NP][d2]
  int N, d0, d1, d2, act1;
  unsigned *sync;
  unsigned *fault;
  unsigned spin_limit;
};

__device__ __forceinline__ f32x4 mfu_ld_sc1(__amdgpu_buffer_rsrc_t rs, long off_floats) {
  typedef unsigned int u32x4m __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(off_floats * 4), 0, 16));
}
__device__ __forceinline__ void mfu_st_sc1(__amdgpu_buffer_rsrc_t rs, long off_floats, f32x4 v) {
  typedef unsigned int u32x4m __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4m, v), rs, (unsigned)(off_floats * 4), 0, 16);
}

#ifdef CLO_MFU_TIMING
#define MFU_STAMP(i) do { if ((tid & 63) == 0 && (blockIdx.x & 63) == 0) reinterpret_cast<unsigned long long *>(p.sync + 32 * 20)[((blockIdx.x >> 6) * 8 + (tid >> 6)) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define MFU_STAMP(i) do { } while (0)
#endif
template <int NT>
__global__ __launch_bounds__(MFU_T) void mid_fused_fwd_kernel(const MidFusedArgs p) {
  constexpr int NP = 16 * NT, WAVES = 8, RG = 2, XW = WAVES - 1;   // wave XW: the exchange wave (owns no early tile loads)
  constexpr int LDB = MFU_KR + 4;
  constexpr int PRE = 4;                                            // tile steps requested before the seam
  constexpr int SPC = (NP * (MFU_KR / 4) + MFU_T - 1) / MFU_T;      // 16-byte pieces per thread and staged array
  extern __shared__ __attribute__((aligned(16))) float s_fu[];   // phase 1: [WAVES][RG][NT][2][4][32]; phase 2: [2 NP][LDB]
  __shared__ unsigned s_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, s4 = (lane >> 4) * 4;
  const int w = blockIdx.x;
  const int N = p.N, d0 = p.d0, d1 = p.d1, d2 = p.d2;
  // ---- geometry: workgroup w runs on XCD w % 8 (observed; used for speed only): a column group shares one L2
  const int fb = w >> 4, kb = 2 * (w & 7) + ((w >> 3) & 1);
  const int S1 = d1 >> 4;
  const int ks0 = kb * S1 / 16, ns = (kb + 1) * S1 / 16 - ks0;            // k16 steps of the K range (<= MFU_MAXS)
  const int k0 = ks0 * 16, kr = ns * 16;
  const int kq = kr >> 2;
  const int q0 = fb * kq / 16, nf1 = 4 * ((fb + 1) * kq / 16 - q0);       // layer-1 slice of this workgroup (<= 16 features)
  const int jA = k0 + 4 * q0;
  const int C2 = (d2 + 15) >> 4;                                           // 16-feature chunks of layer 2
  const int c0 = fb * C2 / 16, nch = (fb + 1) * C2 / 16 - c0;              // chunks of block fb (<= MFU_MAXCH)
  // ---- launch number from the count of finished workgroups (a late starter reads the same quotient)
  unsigned *sy = p.sync;
  const unsigned long long done = __hip_atomic_load(reinterpret_cast<unsigned long long *>(sy), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
  const unsigned target = (unsigned)(16ull * (done / MFU_G + 1ull));
  unsigned *c_err = sy + 32, *c_col = sy + 32 * (2 + kb);
  MFU_STAMP(0);

  float4 tw[MFU_MAXS], tv[MFU_MAXS];   // fragments of one 16-feature chunk of the layer-2 tile: [W row | V row] per k16 step
  auto rows_of = [&](int ch, const float *&pW, const float *&pV) {
    const int row = min((c0 + ch) * 16 + idx, d2 - 1);
    pW = p.W2 + (long)row * d1 + k0 + s4;
    pV = p.V2 + (long)row * d1 + k0 + s4;
  };
  const float *pW0, *pV0;
  rows_of(min(wave, max(nch - 1, 0)), pW0, pV0);   // the wave's first chunk (chunks wave, wave + 8)
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(p.a1, 0, (int)((long)N * d1 * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p.da1, 0, (int)((long)N * d1 * 4), 0x00020000);

  // =====================================================================================
  // phase 1: layer 1 for features [jA, jA + nf1), all rows: the eight waves split K (as mid_full_kernel)
  // =====================================================================================
  {
    const int kpw = (int)(((d0 + 7) / 8 + 15) / 16) * 16;
    const int kb0 = min(wave * kpw, d0);
    const int klen = min(d0, kb0 + kpw) - kb0;   // multiple of 4 (d0 % 16 == 0), may be 0
    const int jlast = max(jA + nf1 - 1, jA);
    const float *pA[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const int row = min(jA + g * 8 + (idx & 7), jlast);
      pA[g] = ((idx >= 8) ? p.V1 : p.W1) + (long)row * d0 + kb0 + s4;
    }
    long offB[NT];
    unsigned bmask[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = t * 16 + idx;
      bmask[t] = n < N ? 0xffffffffu : 0u;
      offB[t] = (long)min(n, N - 1) * d0 + kb0 + s4;
    }
    f32x4 acc1[RG][NT];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc1[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto masked = [](float4 v, unsigned m) {
      return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                         __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
    };
    constexpr int U = NT <= 2 ? 4 : 2;
    const int nfull = klen >> 4;
    int step = 0;
    for (; step + U <= nfull; step += U) {
      float4 av[U][RG], bv[U][NT];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int g = 0; g < RG; ++g) av[u][g] = CLO_LDW(pA[g] + (step + u) * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[u][t] = ld4(p.X + offB[t] + (step + u) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 x = masked(bv[u][t], bmask[t]);
#pragma unroll
          for (int g = 0; g < RG; ++g) {
            acc1[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g].x, x.x, acc1[g][t], 0, 0, 0);
            acc1[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g].y, x.y, acc1[g][t], 0, 0, 0);
            acc1[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g].z, x.z, acc1[g][t], 0, 0, 0);
            acc1[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][g].w, x.w, acc1[g][t], 0, 0, 0);
          }
        }
    }
    for (; step * 16 < klen; ++step) {   // leftover full steps and the partial one
      const bool ok = step * 16 + s4 < klen;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 x = masked(ld4(p.X + offB[t] + (ok ? step * 16 : 0)), ok ? bmask[t] : 0u);
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          const float4 a = ld4(pA[g] + (ok ? step * 16 : 0));
          acc1[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x.x, acc1[g][t], 0, 0, 0);
          acc1[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x.y, acc1[g][t], 0, 0, 0);
          acc1[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x.z, acc1[g][t], 0, 0, 0);
          acc1[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x.w, acc1[g][t], 0, 0, 0);
        }
      }
    }
    MFU_STAMP(1);
    // The first PRE steps of the wave's first tile chunk, behind the layer-1 loads.  A CU's vector-memory pipe delivers in
    // issue order: whatever is requested here is in the way of the seam's traffic below, so it is only as much as the seam
    // takes to run, and the exchange wave requests nothing.
    if (wave != XW) {
#pragma unroll
      for (int st = 0; st < PRE; ++st) {
        tw[st] = CLO_LDW(pW0 + min(st, ns - 1) * 16);
        tv[st] = CLO_LDW(pV0 + min(st, ns - 1) * 16);
      }
    }
    // per wave: z = D rows 0..7 (W1 x), dz = D rows 8..15 (V1 x), brought to the lanes q < 2; merge the waves through LDS
    const int q = lane >> 4, col = lane & 15;
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc1[g][t][r];
          const float up = __shfl(v, (lane + 32) & 63, 64);
          if (q < 2) {
            float *dst = s_fu + ((((wave * RG + g) * NT + t) * 2) * 4 + r) * 32 + lane;
            dst[0] = v;
            dst[4 * 32] = up;
          }
        }
    __syncthreads();
    MFU_STAMP(2);
    if (wave == XW) {
      // ---- the exchange wave: finish the slice (bias, activation), publish it write-through, arrive, poll
      __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(p.dphi1, 0, (int)((long)N * d1 * 4), 0x00020000);
      if (q < 2) {
        for (int job = 0; job < RG * NT; ++job) {
          const int g = job / NT, t = job % NT;
          const int n = t * 16 + col, f0 = g * 8 + q * 4;   // four consecutive features of the slice
          f32x4 av4, dv4, pv4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float z = 0.f, dz = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) {
              const float *src = s_fu + ((((ww * RG + g) * NT + t) * 2) * 4 + r) * 32 + lane;
              z += src[0];
              dz += src[4 * 32];
            }
            const int j = min(jA + f0 + r, d1 - 1);
            float dphi;
            const float aval = act_apply(p.act1, z + (p.b1 ? p.b1[j] : 0.f), dphi);
            av4[r] = aval;
            pv4[r] = dphi;
            dv4[r] = dphi * (dz + (p.Vb1 ? p.Vb1[j] : 0.f));
          }
          if (n < N && f0 < nf1) {   // (nf1 is a multiple of 4: the quad is all in or all out)
            const long off = (long)n * d1 + jA + f0;
            mfu_st_sc1(ra, off, av4);
            mfu_st_sc1(rd, off, dv4);
            mfu_st_sc1(rp, off, pv4);
          }
        }
      }
      MFU_STAMP(3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every write-through store of this wave acknowledged
      __builtin_amdgcn_wave_barrier();
      MFU_STAMP(4);
      if (lane == 0) {
        __hip_atomic_fetch_add(c_col, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0, bad = 0u;
        for (;;) {
          const unsigned seen = __hip_atomic_load(c_col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((int)(seen - target) >= 0) break;
          __builtin_amdgcn_s_sleep(1);
          ++spins;
          if ((spins & 255u) == 0u && __hip_atomic_load(c_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bad = 1u; break; }
          if (spins > p.spin_limit) {   // the grid is not co-resident: end the launch (NaN-marked below), raise the fault word
            __hip_atomic_store(c_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.fault) __hip_atomic_store(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            bad = 1u;
            break;
          }
        }
        s_flag = bad;
      }
      MFU_STAMP(5);
    }
    __syncthreads();
  }
  MFU_STAMP(6);
  const unsigned launch_bad = s_flag;
  // ---- the [a1 ; da1] operand of the K range: requested FIRST (write-through data: sc1 loads, no L1), then the rest of the
  // tile chunk; rows [0, NP) = a1, [NP, 2 NP) = da1 (one array at a time: the buffer resource is wave-uniform)
  {
    const int q4 = kr >> 2, tot = NP * q4;
    f32x4 sv[2][SPC];
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
      for (int u = 0; u < SPC; ++u) {
        const int e = min(tid + u * MFU_T, tot - 1);
        const int n = e / q4, kk = (e - n * q4) * 4;
        sv[which][u] = mfu_ld_sc1(which ? rd : ra, (long)min(n, N - 1) * d1 + k0 + kk);
      }
    if (wave != XW) {
#pragma unroll
      for (int st = PRE; st < MFU_MAXS; ++st) {
        tw[st] = CLO_LDW(pW0 + min(st, ns - 1) * 16);
        tv[st] = CLO_LDW(pV0 + min(st, ns - 1) * 16);
      }
    } else {
#pragma unroll
      for (int st = 0; st < MFU_MAXS; ++st) {
        tw[st] = CLO_LDW(pW0 + min(st, ns - 1) * 16);
        tv[st] = CLO_LDW(pV0 + min(st, ns - 1) * 16);
      }
    }
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
      for (int u = 0; u < SPC; ++u) {
        const int e = tid + u * MFU_T;
        if (e < tot) {
          const int n = e / q4, kk = (e - n * q4) * 4;
          *reinterpret_cast<f32x4 *>(&s_fu[(which * NP + n) * LDB + kk]) = n < N ? sv[which][u] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
  }
  MFU_STAMP(7);
  __syncthreads();
  MFU_STAMP(8);
  // =====================================================================================
  // phase 2: partial z2 / dz2 of tile (fb, kb): wave `wave` owns chunks wave, wave + 8 of the block; a chunk's fragments
  // (<= 11 steps x [W row ; V row]) live in registers, the next chunk's are requested step by step as this one's are used
  // =====================================================================================
  const float *pBa = s_fu + idx * LDB + s4;
  const float *pBd = s_fu + (NP + idx) * LDB + s4;
  const int q = lane >> 4, col = lane & 15;
  for (int ch = wave; ch < nch; ch += WAVES) {
    const bool has_next = ch + WAVES < nch;
    const float *pWn, *pVn;
    rows_of(has_next ? ch + WAVES : ch, pWn, pVn);
    f32x4 accz[NT], accd[NT], accv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      accz[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      accd[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      accv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int st = 0; st < MFU_MAXS; ++st) {
      if (st < ns) {
        const float4 wv = tw[st], vv = tv[st];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 ba = ld4(pBa + t * 16 * LDB + st * 16);
          const float4 bd = ld4(pBd + t * 16 * LDB + st * 16);
#define CLO_MFU_MM(E)                                                                 \
  accz[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.E, ba.E, accz[t], 0, 0, 0);      \
  accd[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.E, bd.E, accd[t], 0, 0, 0);      \
  accv[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.E, ba.E, accv[t], 0, 0, 0);
          CLO_MFU_MM(x) CLO_MFU_MM(y) CLO_MFU_MM(z) CLO_MFU_MM(w)
#undef CLO_MFU_MM
        }
      }
      if (has_next) {   // (wave-uniform) the slot just used takes the same step of the wave's next chunk
        tw[st] = CLO_LDW(pWn + min(st, ns - 1) * 16);
        tv[st] = CLO_LDW(pVn + min(st, ns - 1) * 16);
      }
    }
    MFU_STAMP(ch == wave ? 9 : 10);
    // D layout: row = (lane >> 4) * 4 + r = feature inside the chunk, col = batch row in the tile
    const int j = (c0 + ch) * 16 + q * 4;
    if (j < d2) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float *dst = p.part + (((long)kb * 2) * NP + t * 16 + col) * d2 + j;
        st4(dst, make_float4(accz[t][0], accz[t][1], accz[t][2], accz[t][3]));
        st4(dst + (long)NP * d2, make_float4(accd[t][0] + accv[t][0], accd[t][1] + accv[t][1],
                                              accd[t][2] + accv[t][2], accd[t][3] + accv[t][3]));
      }
    }
  }
  MFU_STAMP(11);
  __syncthreads();
  if (tid == 0) {
    // a launch whose wait was cut short says so in its own output (the slab entry this thread wrote itself)
    if (launch_bad != 0u && nch > 0) p.part[(((long)kb * 2) * NP) * d2 + (long)c0 * 16] = __builtin_nanf("");
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(sy), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace clo

static bool mid_fused_shape_ok(int L, const int *dims, int N) {
  if (L != 3 || N <= NB || N > 64) return false;
  const int d0 = dims[0], d1 = dims[1], d2 = dims[2];
  if (d0 % 16 || d1 % 16 || d2 % 4 || d0 < 64 || dims[3] > HEAD_CMAX) return false;
  if (d1 < 256 || cdiv(d1 / 16, 16) > MFU_MAXS) return false;        // K ranges of 1 .. 11 steps
  if (d2 < 256 || cdiv(cdiv(d2, 16), 16) > MFU_MAXCH) return false;  // feature blocks of <= 11 chunks
  return true;
}
template <int NT>
static size_t mid_fused_smem() {
  const size_t p1 = (size_t)8 * 2 * NT * 2 * 4 * 32, p2 = (size_t)2 * 16 * NT * (MFU_KR + 4);
  return std::max(p1, p2) * sizeof(float);
}
// admission: the column-group counters need all 256 workgroups resident (one per CU), a host-visible fault word, 16-byte
// aligned operands; otherwise the two-launch route serves
template <int NT>
static bool mid_fused_ok(int L, const int *dims, int N, const float *const *W, const float *const *VW, const float *X,
                         const unsigned *sync) {
  // BUILT, parity-green and MEASURED in round 6, OFF by default: 27.5 / 40.4 / 50.4 / 64.6 us at 16 / 32 / 48 / 64 rows against
  // 25.2 / 37 / 51 / 60.5 us for the two launches it replaces (profiles/r06_c2_mid_fused_forward_timeline.txt: the phases of one
  // workgroup per CU run back to back -- layer 1, slice epilogue, seam, operand staging, tile products -- where two launches
  // with two workgroups per CU hide each other's prologues).
#ifndef CLO_MLP_MID_FUSED
#define CLO_MLP_MID_FUSED 0
#endif
  static const bool on = CLO_MLP_MID_FUSED != 0;
  if (!on || !sync || !mid_fused_shape_ok(L, dims, N)) return false;
  if (!W[0] || !VW[0] || !W[1] || !VW[1]) return false;
  for (const void *q : {(const void *)W[0], (const void *)VW[0], (const void *)W[1], (const void *)VW[1], (const void *)X})
    if (!aligned16(q)) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (device_cu_count(dev) != MFU_G) return false;
  if (!fault_words_device(dev) || fault_disabled(dev, FAULT_MEGA)) return false;
  static int occ[64];
  static std::once_flag once;
  std::call_once(once, [] { for (int &o : occ) o = -1; });
  int o = __atomic_load_n(&occ[dev], __ATOMIC_RELAXED);
  if (o < 0) {
    int nb = 0;
    const void *fn = reinterpret_cast<const void *>(mid_fused_fwd_kernel<NT>);
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_fused_smem<NT>());
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, MFU_T, mid_fused_smem<NT>()) != hipSuccess) {
      (void)hipGetLastError();
      nb = 0;
    }
    o = nb >= 1 ? 1 : 0;
    __atomic_store_n(&occ[dev], o, __ATOMIC_RELAXED);
  }
  return o == 1;
}
template <int NT>
static int mid_fused_launch(const MidFusedArgs &a0, hipStream_t st) {
  MidFusedArgs a = a0;
  int dev = 0;
  int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
  if (rc != CLO_OK) return rc;
  a.fault = fault_words_device(dev);
  if (a.fault) a.fault += FAULT_MEGA;
  a.spin_limit = spin_limit();
  PersistGate &gate = PersistGate::of(dev);
  rc = gate.admit(st, MFU_G);
  if (rc != CLO_OK) return rc;
  {
    ProfScope prof(0, 8.0 * ((double)a.d0 * a.d1 + (double)a.d1 * a.d2), st);
    hipLaunchKernelGGL((mid_fused_fwd_kernel<NT>), dim3(MFU_G), dim3(MFU_T), mid_fused_smem<NT>(), st, a);
  }
  rc = check_hip(hipGetLastError(), "mid_fused_fwd_kernel");
  if (rc != CLO_OK) {
    gate.abort();
    return rc;
  }
  return gate.done(st);
}

constexpr int MID_MAX_N = 64;

static bool mid_chain_ok(int L, const int *dims, const float *const *W, const float *const *VW,
                         float *const *OW, int N) {
#ifndef CLO_MLP_NO_MID
#define CLO_MLP_NO_MID 0
#endif
  static const bool off = CLO_MLP_NO_MID != 0;
  if (off || N <= NB || N > MID_MAX_N || L < 2 || L > OUTER_MAXL) return false;
  for (int l = 1; l <= L - 1; ++l)
    if (dims[l - 1] % 4 != 0 || dims[l] % 4 != 0 || dims[l - 1] < 16 || !aligned16(W[l - 1]) ||
        !aligned16(VW[l - 1]) || !aligned16(OW[l - 1]))
      return false;
  return aligned16(OW[L - 1]);  // out_W_L is written by the float4 outer-product kernel
}

template <int NT>
static int mid_chain(int L, const int *dims, const int *acts, const float *const *W,
                     const float *const *b, const float *const *VW, const float *const *Vb,
                     float *const *OW, float *const *Ob, int N, int loss_kind, const float *aux,
                     int aux_rank, float scale, float beta, float *const *a, float *const *da,
                     float *const *dphi, float *const *dl, float *gws, long gws_sz, unsigned *fsync, hipStream_t st) {
  constexpr int NP = 16 * NT;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  const int C = dims[L], dh = dims[L - 1];
  const int head_nblk = (int)cdiv(dh, 256);
  // carve gws
  float *fslab = gws;
  // the GEMM workspace this chain borrows holds 4096 dmax floats: 32 K ranges / 24 row ranges of up to 32
  // padded rows, half as many of 48 / 64
  constexpr long KS_MAX = NT <= 2 ? 32 : 16, JB_MAX = 24;
  const long fslab_sz = KS_MAX * 2 * NP * dmax;
  float *hp = fslab + fslab_sz;
  const long hp_sz = (long)N * head_nblk * 2 * HEAD_CMAX + 64;
  float *dLbuf = hp + hp_sz;                          // [N][HEAD_CMAX] delta of the head
  const long dL_sz = (long)NP * HEAD_CMAX;
  float *dslab[2] = {dLbuf + dL_sz, dLbuf + dL_sz + JB_MAX * NP * dmax};
  if (fslab_sz + hp_sz + dL_sz + 2 * JB_MAX * NP * dmax > gws_sz) return CLO_EUNSUP;
  int rc;
  // ---- forward + JVP of layers 1 and 2 in ONE persistent launch where the network and the device allow it
  int l_first = 1;
  if (mid_fused_ok<NT>(L, dims, N, W, VW, a[0], fsync)) {
    MidFusedArgs fa{};
    fa.W1 = W[0]; fa.b1 = b ? b[0] : nullptr; fa.V1 = VW[0]; fa.Vb1 = Vb ? Vb[0] : nullptr; fa.X = a[0];
    fa.W2 = W[1]; fa.V2 = VW[1];
    fa.a1 = a[1]; fa.da1 = da[1]; fa.dphi1 = dphi[1]; fa.part = fslab;
    fa.N = N; fa.d0 = dims[0]; fa.d1 = dims[1]; fa.d2 = dims[2]; fa.act1 = acts[0];
    fa.sync = fsync;
    rc = mid_fused_launch<NT>(fa, st);
    if (rc != CLO_OK) return rc;
    {  // finish layer 2 (16 K ranges) + partial products of the head: as behind mid_fwd_kernel
      HeadFwdArgs ha{};
      ha.part = fslab; ha.ksplit = 16; ha.part_rows = NP;
      ha.b = b ? b[1] : nullptr; ha.Vb = Vb ? Vb[1] : nullptr;
      ha.a = a[2]; ha.da = da[2]; ha.dphi = dphi[2];
      ha.N = N; ha.d = dims[2]; ha.act = acts[1];
      ha.WL = W[L - 1]; ha.VL = VW[L - 1]; ha.C = C; ha.hp = hp;
      ProfScope pf(3, 0.0, st);
      hipLaunchKernelGGL(head_fwd_kernel, dim3(head_nblk, N), dim3(256), 0, st, ha);
      CLO_CHECK_LAUNCH("head_fwd_kernel");
    }
    l_first = L;   // (L == 3: both hidden layers are done)
  }
  // ---- forward + JVP: hidden layers 1 .. L-1
  for (int l = l_first; l <= L - 1; ++l) {
    const int di = dims[l - 1], dout = dims[l];
    const bool has_da = l > 1;
    // Layers without an incoming tangent (the first one) finish in their own launch (in-block split-K over the
    // 8 waves: no slabs, no finish launch: -2...4 us).  With a tangent the kernel issues twice the B loads per
    // weight load and measured slower than the slab route (C2 layer 2: 24.8 vs 17.9 us at 16 rows, 67 vs 46 us
    // at 64); CLO_MLP_MID_FULL_DA=1 forces it.
#ifndef CLO_MLP_NO_MID_FULL
#define CLO_MLP_NO_MID_FULL 0
#endif
    static const bool no_full = CLO_MLP_NO_MID_FULL != 0;
#ifndef CLO_MLP_MID_FULL_DA
#define CLO_MLP_MID_FULL_DA 0
#endif
    static const bool full_da = CLO_MLP_MID_FULL_DA != 0;
    if (!no_full && (!has_da || full_da) && di >= 256 && cdiv(dout, 16) >= kNumCU / 2) {
      const int kpw = (int)cdiv(cdiv(di, 8), 16) * 16;
      const int fpb = (int)std::min<long>(16, std::max<long>(4, cdiv(dout, kNumCU)));
      const size_t smem = (size_t)8 * 2 * NT * 2 * 4 * 32 * sizeof(float);
      {
        ProfScope prof(0, 8.0 * di * dout, st);
#define CLO_MIDFULL(DA, UU)                                                                                   \
  rc = set_smem(mid_full_kernel<NT, DA, UU>, smem);                                                           \
  if (rc != CLO_OK) return rc;                                                                                \
  hipLaunchKernelGGL((mid_full_kernel<NT, DA, UU>), dim3((unsigned)cdiv(dout, fpb)), dim3(512), smem, st,     \
                     W[l - 1], b ? b[l - 1] : nullptr, VW[l - 1], Vb ? Vb[l - 1] : nullptr, a[l - 1],         \
                     DA ? da[l - 1] : nullptr, a[l], da[l], dphi[l], N, di, dout, acts[l - 1], kpw, fpb)
        if (!has_da) {
#ifndef CLO_MID_FULL_U8
#define CLO_MID_FULL_U8 1
#endif
          // (all eight k steps of a wave in ONE batch of loads: a single memory round trip; round 6: also at 33 ... 48 rows, -1 ... -2 us;
          // at 49 ... 64 rows the batch needs 250 registers per lane and measures 1 - 1.5 us slower than two batches of four)
          if (kpw == 128 && di % 128 == 0 && (NT <= 2 || (CLO_MID_FULL_U8 && NT == 3))) { CLO_MIDFULL(false, 8); } else { CLO_MIDFULL(false, 4); }
        } else {
          if (NT <= 2) { CLO_MIDFULL(true, 4); } else { CLO_MIDFULL(true, 2); }
        }
#undef CLO_MIDFULL
        CLO_CHECK_LAUNCH("mid_full_kernel");
      }
      if (l == L - 1) {   // partial products of the head from the finished activations
        HeadFwdArgs fa{};
        fa.part = nullptr; fa.ksplit = 1; fa.part_rows = NP;
        fa.a = a[l]; fa.da = da[l]; fa.dphi = dphi[l];
        fa.N = N; fa.d = dout; fa.act = acts[l - 1];
        fa.WL = W[L - 1]; fa.VL = VW[L - 1]; fa.C = C; fa.hp = hp;
        ProfScope pf(3, 0.0, st);
        hipLaunchKernelGGL(head_fwd_kernel, dim3(head_nblk, N), dim3(256), 0, st, fa);
        CLO_CHECK_LAUNCH("head_fwd_kernel");
      }
      continue;
    }
    const int cols = (has_da ? 2 : 1) * NP;
    // Besides the weights a launch moves the activations every block stages (row blocks x cols x d_in) and
    // the split-K slabs (written here, read by the finish): both cost like weight bytes.  Pick waves per
    // block (4 / 8 = 64 / 128 features) and the K split that minimise them at >= ~0.8 blocks per CU.
    const long lds_b = NT <= 2 ? 65536 : 150000;   // the staged activations: one block per CU beyond 32 rows
    const long kpb_max = std::min<long>(MF_KB_MAX, ((lds_b / (4 * cols) - 4) / 32) * 32);
    int wv = 4;
    long ksplit = 1, kpb = 32;
    double best = 1e300;
#ifndef CLO_MLP_MID_WV
#define CLO_MLP_MID_WV 0
#endif
#ifndef CLO_MID_KGRAN16
#define CLO_MID_KGRAN16 0   // (measured neutral, same box: 64 rows 117.5 vs 117.2 us)
#endif
    for (int w : {4, 8}) {
      if (CLO_MLP_MID_WV && w != CLO_MLP_MID_WV) continue;   // (A/B builds: force the waves per block)
      const long rb = cdiv(dout, w * 16);
      for (long ks = 1; ks <= KS_MAX; ++ks) {
        const long kgran = (NT >= 3 && CLO_MID_KGRAN16) ? 16 : 32;   // (MFMA-bound beyond 32 rows: the finer K ranges fill the chip -- 21 x 12 blocks instead of 21 x 11)
        const long kp = cdiv(cdiv(di, ks), kgran) * kgran;
        if (kp > kpb_max) continue;
        const long kse = cdiv(di, kp);
        const long blocks = rb * kse;
        if (blocks * 5 < kNumCU * 4 && !(ks == KS_MAX && best == 1e300)) continue;
        // (round 6) ... weighted by how evenly the grid fills the chip: `slots` blocks run at a time (LDS and the 16 waves
        // of a CU), a grid of 273 one-per-CU blocks takes two rounds for the work of 1.07
        const long per_cu = std::max<long>(1, std::min<long>(160 * 1024 / ((long)cols * (kp + 4) * 4), 16 / w));
        const long slots = per_cu * kNumCU, rounds = cdiv(blocks, slots);
        // (up to 32 rows the kernel is bandwidth-bound and a single, partly filled round costs nothing extra -- its blocks share the
        // bandwidth the missing ones would have used; beyond that it is MFMA-bound and an idle CU is lost time: 64 rows 117 us with
        // 231 blocks, 124 us with 210)
        // (blocks spread over the CUs first: 273 blocks with two slots per CU still leave 17 CUs with twice the work of the others --
        // the last blocks of the 32-row forward ended at 24 us, the mean at 17)
        const long cu_rounds = cdiv(blocks, (long)kNumCU);
        const double fill = blocks > kNumCU ? (double)(cu_rounds * kNumCU) / (double)blocks
                                            : (NT >= 3 ? (double)kNumCU / (double)blocks : 1.0);
        (void)rounds;
        const double traffic = ((double)rb * cols * di + 2.0 * (double)kse * 2 * NP * dout) * fill;
        if (traffic < best) { best = traffic; wv = w; ksplit = kse; kpb = kp; }
      }
    }
    if (best == 1e300 || ksplit > KS_MAX) return CLO_EUNSUP;
    const long row_blocks = cdiv(dout, wv * 16);
    const size_t smem = (size_t)cols * (kpb + 4) * sizeof(float);
    dim3 grid((unsigned)row_blocks, (unsigned)ksplit), block(wv * 64);
    {
      ProfScope prof(0, 8.0 * di * dout, st);
#ifdef CLO_MID_TIMING
#define CLO_MIDF_STAMP_ARG , g_mid_stamps_host
#else
#define CLO_MIDF_STAMP_ARG
#endif
#define CLO_MIDF(DA, WVV)                                                                                  \
  rc = set_smem(mid_fwd_kernel<NT, DA, WVV>, smem);                                                        \
  if (rc != CLO_OK) return rc;                                                                             \
  hipLaunchKernelGGL((mid_fwd_kernel<NT, DA, WVV>), grid, block, smem, st, W[l - 1], VW[l - 1], a[l - 1],  \
                     DA ? da[l - 1] : nullptr, fslab, N, di, dout, (int)kpb CLO_MIDF_STAMP_ARG)
      if (has_da) { if (wv == 8) { CLO_MIDF(true, 8); } else { CLO_MIDF(true, 4); } }
      else { if (wv == 8) { CLO_MIDF(false, 8); } else { CLO_MIDF(false, 4); } }
#undef CLO_MIDF
      CLO_CHECK_LAUNCH("mid_fwd_kernel");
    }
    if (l < L - 1) {
      ProfScope pf(3, 0.0, st);
      hipLaunchKernelGGL(fwd_finish_kernel, dim3(ew_grid((long)N * dout)), dim3(256), 0, st, fslab, (int)ksplit,
                         b ? b[l - 1] : nullptr, Vb ? Vb[l - 1] : nullptr, a[l], da[l], dphi[l], N, dout,
                         acts[l - 1], NP);
      CLO_CHECK_LAUNCH("fwd_finish_kernel");
    } else {  // last hidden layer: finish + partial products of the head
      HeadFwdArgs fa{};
      fa.part = fslab; fa.ksplit = (int)ksplit; fa.part_rows = NP;
      fa.b = b ? b[l - 1] : nullptr; fa.Vb = Vb ? Vb[l - 1] : nullptr;
      fa.a = a[l]; fa.da = da[l]; fa.dphi = dphi[l];
      fa.N = N; fa.d = dout; fa.act = acts[l - 1];
      fa.WL = W[L - 1]; fa.VL = VW[L - 1]; fa.C = C; fa.hp = hp;
      ProfScope pf(3, 0.0, st);
      hipLaunchKernelGGL(head_fwd_kernel, dim3(head_nblk, N), dim3(256), 0, st, fa);
      CLO_CHECK_LAUNCH("head_fwd_kernel");
    }
  }
  // ---- head: loss Hessian, out_W_L / out_b_L, delta_{L-1}
  {
    HeadBwdArgs ba{};
    ba.hp = hp; ba.nblk = head_nblk;
    ba.bL = b ? b[L - 1] : nullptr; ba.VbL = Vb ? Vb[L - 1] : nullptr;
    ba.kind = loss_kind; ba.aux = aux; ba.aux_rank = aux_rank; ba.scale = scale;
    ba.WL = W[L - 1]; ba.a_prev = a[L - 1]; ba.dphi_prev = dphi[L - 1];
    ba.out_W = nullptr; ba.out_b = nullptr; ba.delta_prev = dl[L - 1];  // out_W_L / out_b_L: mid_outer_kernel
    ba.beta = beta; ba.N = N; ba.d = dh; ba.C = C;
    ProfScope pf(1, 4.0 * dh * C, st);
    hipLaunchKernelGGL(head_bwd_rows_kernel, dim3(head_nblk, (unsigned)cdiv(N, NB)), dim3(256), 0, st, ba, dLbuf);
    CLO_CHECK_LAUNCH("head_bwd_rows_kernel");
  }
  // ---- data chain delta_{l-1} = delta_l W_l for l = L-1 .. 2, each step merged with the outer products that need delta_l
  // only (layer l; at the first step also the head's layer L); the last launch forms the remaining outer products
#ifndef CLO_MID_OCW128
#define CLO_MID_OCW128 0
#endif
  constexpr int OCW = (NT <= 2 && !CLO_MID_OCW128) ? 256 : 128;
  const size_t osmem = (size_t)NP * (MIDO_ROWS + 16 + OCW) * sizeof(float);
  // (MEASURED in round 6 and left OFF: the merged launch is 0.5 - 3 us SLOWER at every row count, profiles/
  // r06_c2_mid_merged_backward_ab.txt -- a read-only sweep and a write-only stream sharing the chip slow each other down by
  // more than the saved boundary, as round 1 found for a kernel that did both per tile)
#ifndef CLO_MLP_MID_MERGE
#define CLO_MLP_MID_MERGE 0
#endif
  static const bool merge_on = CLO_MLP_MID_MERGE != 0;
  MidDelta md[OUTER_MAXL + 2];
  md[L - 1] = MidDelta{dl[L - 1], nullptr, nullptr, 0, 0};
  md[L] = MidDelta{dLbuf, nullptr, nullptr, 0, HEAD_CMAX};  // the head's delta, [N][HEAD_CMAX]
  bool outer_done[OUTER_MAXL + 2] = {};
  auto outer_args = [&](const int *layers, int count, MidOuterArgs &oa, double &bytes) {
    oa = MidOuterArgs{};
#ifdef CLO_MID_TIMING
    oa.stamps = g_mid_stamps_host;
#endif
    oa.nlayers = count; oa.alpha = 1.f; oa.beta = beta; oa.N = N;
    int nb = 0;
    bytes = 0;
    for (int k = 0; k < count; ++k) {
      const int l = layers[k];
      oa.first_block[k] = nb;
      oa.md[k] = md[l]; oa.a_prev[k] = a[l - 1];
      oa.out_W[k] = OW[l - 1]; oa.out_b[k] = Ob ? Ob[l - 1] : nullptr;
      oa.d_in[k] = dims[l - 1]; oa.d_out[k] = dims[l];
      nb += (int)(cdiv(dims[l - 1], OCW) * cdiv(dims[l], MIDO_ROWS));
      bytes += 4.0 * dims[l - 1] * dims[l] * (beta != 0.f ? 2 : 1);
    }
    oa.first_block[count] = nb;
    return nb;
  };
  for (int l = L - 1; l >= 2; --l) {
    const int di = dims[l - 1], dout = dims[l];
    // (beyond 32 rows a block's delta rows and merge buffer leave room for ONE block per CU: the grid must not exceed the CUs --
    // round 6: JB_MAX was 12 there, C2's layer 2 ran on 11 x 12 = 132 of the 256 CUs, 28.8 us at 64 rows)
#ifndef CLO_MID_DPREV_CQ2
#define CLO_MID_DPREV_CQ2 1
#endif
    // (beyond 32 rows, round 6: blocks of 128 columns x 4 row quarters -- the same number of blocks with half the row ranges, i. e.
    // half the slabs this launch writes and the finish launch reads)
#ifndef CLO_MID_DPREV_CQ2_ALL
#define CLO_MID_DPREV_CQ2_ALL 0
#endif
    constexpr int DCQ = ((NT >= 3 || CLO_MID_DPREV_CQ2_ALL) && CLO_MID_DPREV_CQ2 && !CLO_MLP_MID_MERGE) ? 2 : 4;
    constexpr int DCOLS = 64 * DCQ;
#ifndef CLO_MID_DPREV_FLOOR
#define CLO_MID_DPREV_FLOOR 1
#endif
    // (never more blocks than CUs: 11 x 24 = 264 blocks left eight CUs with two blocks each -- twice the time of the other 248)
    long JB = (NT <= 2 && !CLO_MID_DPREV_FLOOR) ? cdiv(kNumCU, cdiv(di, DCOLS)) : std::max<long>(1, kNumCU / cdiv(di, DCOLS));
    JB = std::min<long>({JB, cdiv(dout, 64), JB_MAX});
    JB = std::max<long>(JB, cdiv(dout, NT <= 2 ? 512 : 256));   // LDS: rows x (Npad + 16) delta + merge buffer
    const int rpb = (int)(cdiv(cdiv(dout, JB), 8) * 8);
    const int JBe = (int)cdiv(dout, rpb);
    if (JBe > JB_MAX) return CLO_EUNSUP;
    float *slab = dslab[l & 1];
    const size_t smem = ((size_t)((rpb + 7) & ~7) * mid_ldd(NT) + NT * 4096) * sizeof(float);
    const int fin = JBe == 1 ? 1 : 0;
    MidDprevArgs dq{W[l - 1], md[l], dphi[l - 1], fin ? dl[l - 1] : slab, N, di, dout, rpb, fin};
    const int gx = (int)cdiv(di, DCOLS);
#ifdef CLO_MID_TIMING
    dq.stamps = g_mid_stamps_host; dq.gx = gx;
#endif
    if (merge_on) {
      int layers[2] = {l, L};
      const int count = l == L - 1 ? 2 : 1;
      MidOuterArgs oa;
      double obytes;
      const int nbo = outer_args(layers, count, oa, obytes);
      const size_t msmem = std::max(smem, osmem);
      ProfScope prof(2, 4.0 * di * dout + obytes, st);
      if (beta != 0.f) {
        rc = set_smem(mid_bwd_merged_kernel<NT, true, OCW>, msmem);
        if (rc != CLO_OK) return rc;
        hipLaunchKernelGGL((mid_bwd_merged_kernel<NT, true, OCW>), dim3((unsigned)(gx * JBe + nbo)), dim3(512), msmem, st, dq,
                           oa, gx * JBe, gx);
      } else {
        rc = set_smem(mid_bwd_merged_kernel<NT, false, OCW>, msmem);
        if (rc != CLO_OK) return rc;
        hipLaunchKernelGGL((mid_bwd_merged_kernel<NT, false, OCW>), dim3((unsigned)(gx * JBe + nbo)), dim3(512), msmem, st, dq,
                           oa, gx * JBe, gx);
      }
      CLO_CHECK_LAUNCH("mid_bwd_merged_kernel");
      outer_done[l] = true;
      if (count == 2) outer_done[L] = true;
    } else {
      rc = set_smem(mid_dprev_kernel<NT, DCQ>, smem);
      if (rc != CLO_OK) return rc;
      ProfScope prof(2, 4.0 * di * dout, st);
      hipLaunchKernelGGL((mid_dprev_kernel<NT, DCQ>), dim3((unsigned)gx, (unsigned)JBe), dim3(512), smem, st, dq);
      CLO_CHECK_LAUNCH("mid_dprev_kernel");
    }
#ifndef CLO_MID_NT2_STREAM
#define CLO_MID_NT2_STREAM 0
#endif
#ifndef CLO_MID_FINISH_MIN_NT
#define CLO_MID_FINISH_MIN_NT 3
#endif
    if (!fin && (NT >= CLO_MID_FINISH_MIN_NT || (NT == 2 && CLO_MID_NT2_STREAM))) {   // one pass over the slabs instead of one per consumer block
      const long total4 = (long)N * di / 4;
      hipLaunchKernelGGL(mid_delta_finish_kernel, dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, st, slab, JBe,
                         (long)NP * di, dphi[l - 1], dl[l - 1], total4);
      CLO_CHECK_LAUNCH("mid_delta_finish_kernel");
      md[l - 1] = MidDelta{dl[l - 1], nullptr, nullptr, 0, 0};
    } else {
      md[l - 1] = fin ? MidDelta{dl[l - 1], nullptr, nullptr, 0, 0} : MidDelta{nullptr, slab, dphi[l - 1], JBe, 0};
    }
  }
  // ---- the outer products not formed yet (layer 1; every layer of a two-layer net), one launch
  {
    int layers[OUTER_MAXL + 1];
    int count = 0;
    for (int l = 1; l <= L; ++l)
      if (!outer_done[l]) layers[count++] = l;
    MidOuterArgs oa;
    double bytes;
    const int nb = outer_args(layers, count, oa, bytes);
    ProfScope prof(4, bytes, st);
#ifndef CLO_MLP_MID_OUTER2
#define CLO_MLP_MID_OUTER2 1
#endif
    bool aligned4 = true;
    for (int k = 0; k < count; ++k) aligned4 = aligned4 && dims[layers[k] - 1] % 4 == 0;
    // (measured: at <= 32 rows the tile-per-block kernel is 1.5 - 7 us faster -- its blocks are short there and delta_1 still
    // arrives as row-range slabs that every block would sum again; from 33 rows on the streaming form wins)
    if (CLO_MLP_MID_OUTER2 && aligned4 && (NT >= 3 || (NT == 2 && CLO_MID_NT2_STREAM))) return launch_mid_outer2<NT>(oa, count, beta, N, st);
    rc = beta != 0.f ? set_smem(mid_outer_kernel<NT, true, OCW>, osmem)
                     : set_smem(mid_outer_kernel<NT, false, OCW>, osmem);
    if (rc != CLO_OK) return rc;
    if (beta != 0.f) hipLaunchKernelGGL((mid_outer_kernel<NT, true, OCW>), dim3(nb), dim3(512), osmem, st, oa);
    else hipLaunchKernelGGL((mid_outer_kernel<NT, false, OCW>), dim3(nb), dim3(512), osmem, st, oa);
    CLO_CHECK_LAUNCH("mid_outer_kernel");
  }
  return CLO_OK;
}

// Forward + JVP of one layer on the GEMM engine: a_l = act(a W^T + b) with act' as second output,
// da_l = act' * (a VW^T + da W^T + Vb).  Preferred: one fused pass over W_l and V_l (three MFMA products per
// tile, gemm_fwd3_kernel); unaligned operands (layer inputs % 4 != 0): two or three plain products with bias /
// activation / derivative fused into whichever kernel writes C, the tangent's two products chained along K
// where the aligned engine allows it.
static int fwd_jvp_gemm(const float *a_in, const float *da_in, const float *Wl, const float *Vl, const float *bl,
                        const float *vbl, float *a_out, float *da_out, float *dphi_out, int N, int di, int dout,
                        int act, float *gws, long gws_sz, hipStream_t st) {
  int rc = launch_mlp_fwd3(a_in, da_in, Wl, Vl, bl, vbl, a_out, da_out, dphi_out, N, di, dout, act, gws, gws_sz, st);
  if (rc != CLO_EUNSUP) return rc;
  GemmArgs g1 = gemm_problem(N, dout, di, a_in, di, 1, Wl, 1, di, 0.f, a_out, dout);
  g1.epi = EPI_ACT; g1.e_act = act; g1.e_vec = bl; g1.e_out2 = dphi_out;
  rc = launch_gemm_auto(g1, gws, gws_sz, st);
  if (rc != CLO_OK) return rc;
  GemmArgs g2 = gemm_problem(N, dout, di, a_in, di, 1, Vl, 1, di, 0.f, da_out, dout);
  g2.epi = EPI_MUL; g2.e_vec = vbl; g2.e_mul = dphi_out; g2.ld_mul = dout;
  if (!da_in) return launch_gemm_auto(g2, gws, gws_sz, st);
  GemmArgs gc = g2;
  gc.K = 2 * di; gc.K1 = di; gc.A2 = da_in; gc.B2 = Wl;
  if (di % 32 == 0 && gemm_v2_eligible(gc, 1)) return launch_gemm_auto(gc, gws, gws_sz, st);
  g2.epi = EPI_NONE;
  rc = launch_gemm_auto(g2, gws, gws_sz, st);
  if (rc != CLO_OK) return rc;
  GemmArgs g3 = gemm_problem(N, dout, di, da_in, di, 1, Wl, 1, di, 1.f, da_out, dout);
  g3.epi = EPI_MUL; g3.e_vec = vbl; g3.e_mul = dphi_out; g3.ld_mul = dout;
  return launch_gemm_auto(g3, gws, gws_sz, st);
}

// Workspace layout of clo_mlp_ggn_matvec (floats):
//   per layer l = 1..L : a_l, da_l, dphi_l, each [N][d_l]
//   delta ping/pong    : 2 x [N][dmax]
//   slabs              : max over layers of the fwd split-K / bwd row-range partial slabs
//   GEMM split-K slabs : only when N > SKINNY_MAX_N
static long ggn_ws_core_floats(int L, const int *dims, int N);
// the persistent kernel's exchange area and counters sit behind everything else (64-float aligned)
static long ggn_ws_mega_offset(int L, const int *dims, int N) {
  return cdiv(ggn_ws_core_floats(L, dims, N), 64) * 64;
}
extern "C" long clo_mlp_ggn_ws_floats(int L, const int *dims, int N) {
  if (L <= 0 || !dims || N < 0) return 0;
  if (mid_fused_shape_ok(L, dims, N)) return ggn_ws_mega_offset(L, dims, N) + MFU_SYNC_WORDS;
  if (!mega_shape_ok(L, dims, N)) return ggn_ws_core_floats(L, dims, N);
  return ggn_ws_mega_offset(L, dims, N) + cdiv(mega_xch_floats(dims[1], dims[2]), 64) * 64 + mega_sync_words() +
         mega_debug_floats();
}
extern "C" int clo_mlp_ggn_ws_init(int L, const int *dims, int N, float *ws, void *stream) {
  CLO_REQUIRE(L >= 1 && dims && N >= 0 && ws, "clo_mlp_ggn_ws_init: bad arguments");
  if (mid_fused_shape_ok(L, dims, N))   // counters of the fused two-layer forward of the 9 ... 64-row chain
    return check_hip(hipMemsetAsync(ws + ggn_ws_mega_offset(L, dims, N), 0, (size_t)MFU_SYNC_WORDS * 4, (hipStream_t)stream),
                     "hipMemsetAsync(fused forward counters)");
  if (!mega_shape_ok(L, dims, N)) return CLO_OK;  // nothing to initialise
  // the counters AND the exchange area: its tagged slots (CLO_MG_TOPLL in mlp_mega.hip) must not show a tag of some earlier
  // owner of this memory
  float *xch = ws + ggn_ws_mega_offset(L, dims, N);
  const size_t words = (size_t)cdiv(mega_xch_floats(dims[1], dims[2]), 64) * 64 + (size_t)mega_sync_words();
  return check_hip(hipMemsetAsync(xch, 0, words * 4, (hipStream_t)stream), "hipMemsetAsync(matvec exchange area)");
}
static long ggn_ws_core_floats(int L, const int *dims, int N) {
  if (L <= 0 || !dims || N < 0) return 0;
  long total = 0;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  for (int l = 1; l <= L; ++l) total += 4L * N * dims[l];  // a, da, dphi, delta per layer
  total += 2L * N * dmax;
  long part = 0;
  for (int l = 1; l <= L; ++l) {
    part = std::max(part, clo_mlp_bwd_ws_floats(N, dims[l - 1], dims[l]));
    part = std::max(part, clo_mlp_fwd_ws_floats(N, dims[l - 1], dims[l]));
  }
  total += part;
  if (N > SKINNY_MAX_N) total += gemm_ws_floats(N, dmax);
  else total += (long)NB * (cdiv(dmax, 256) + 1) * 2 * HEAD_CMAX;  // head partials
  return total + 256;
}

extern "C" int clo_mlp_ggn_matvec(int L, const int *dims, const int *acts, const float *const *W,
                                  const float *const *b, const float *const *VW,
                                  const float *const *Vb, float *const *OW, float *const *Ob,
                                  const float *X, int N, int loss_kind, const float *aux,
                                  int aux_rank, float loss_scale, float alpha, float beta, int flags,
                                  float *ws, void *stream) {
  CLO_REQUIRE(L >= 1 && L <= 64 && dims && acts && W && VW && OW,
              "clo_mlp_ggn_matvec: bad layer table");
  CLO_REQUIRE((flags & ~1) == 0, "clo_mlp_ggn_matvec: unknown flags 0x%x", flags);
  CLO_REQUIRE(N >= 0 && X && ws, "clo_mlp_ggn_matvec: bad batch / workspace");
  CLO_REQUIRE(loss_kind >= 0 && loss_kind <= CLO_LOSS_EF_BCE, "clo_mlp_ggn_matvec: unknown loss kind %d",
              loss_kind);
  CLO_REQUIRE((loss_kind != CLO_LOSS_RANK1 && !loss_is_ef(loss_kind)) || (aux && aux_rank >= 1),
              "clo_mlp_ggn_matvec: RANK1 / EF_* need aux (backpropagated vectors / targets) and aux_rank >= 1");
  for (int l = 0; l <= L; ++l) CLO_REQUIRE(dims[l] > 0, "clo_mlp_ggn_matvec: dims[%d] <= 0", l);
  for (int l = 0; l < L; ++l) {
    CLO_REQUIRE(acts[l] >= 0 && acts[l] <= 3, "clo_mlp_ggn_matvec: unknown activation");
    CLO_REQUIRE(W[l] && VW[l] && OW[l], "clo_mlp_ggn_matvec: null weight pointer in layer %d", l);
  }
  hipStream_t st = (hipStream_t)stream;
  {
    // an earlier persistent launch on this device ran out of its spin budget (its grid was not co-resident): reported
    // once, here; the launch chain serves this device from now on (clo_common.h, "asynchronous faults")
    int fdev = 0;
    if (hipGetDevice(&fdev) == hipSuccess && fault_take(fdev, FAULT_MEGA)) {
      set_error("clo_mlp_ggn_matvec: an EARLIER product on device %d timed out inside the persistent kernel (its "
                "workgroups were not co-resident: GPU shared with another process or CU-masked); that product's result is "
                "invalid.  The launch chain is used on this device from now on -- repeat the call.", fdev);
      return CLO_EASYNC;
    }
  }
  if (N == 0) {  // empty batch contributes nothing; still honour beta
    if (beta != 1.f)
      for (int l = 0; l < L; ++l) {
        int rc = clo_axpby_f32(OW[l], OW[l], (long)dims[l] * dims[l + 1], 0.f, beta, stream);
        if (rc != CLO_OK) return rc;
        if (Ob && Ob[l]) {
          rc = clo_axpby_f32(Ob[l], Ob[l], dims[l + 1], 0.f, beta, stream);
          if (rc != CLO_OK) return rc;
        }
      }
    return CLO_OK;
  }

  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  // carve the workspace
  float *p = ws;
  float *a[65], *da[65], *dphi[65], *dl[65];
  a[0] = const_cast<float *>(X); da[0] = nullptr; dphi[0] = nullptr; dl[0] = nullptr;
  for (int l = 1; l <= L; ++l) {
    const long sz = (long)N * dims[l];
    a[l] = p; p += sz; da[l] = p; p += sz; dphi[l] = p; p += sz; dl[l] = p; p += sz;
  }
  float *dl0 = p; p += (long)N * dmax;
  float *dl1 = p; p += (long)N * dmax;
  float *part = p;
  long part_sz = 0;
  for (int l = 1; l <= L; ++l) {
    part_sz = std::max(part_sz, clo_mlp_bwd_ws_floats(N, dims[l - 1], dims[l]));
    part_sz = std::max(part_sz, clo_mlp_fwd_ws_floats(N, dims[l - 1], dims[l]));
  }
  p += part_sz;
  float *gws = p;
  const long gws_sz = N > SKINNY_MAX_N ? gemm_ws_floats(N, dmax) : 0;

  const bool skinny = N <= SKINNY_MAX_N;
  const bool last_linear = acts[L - 1] == CLO_ACT_IDENTITY;
  int rc;
  int last_ksplit = 1;
  // narrow linear head: fold fwd(L) + loss + bwd(L) into the neighbouring launches
  const bool narrow = L >= 2 && last_linear && dims[L] <= HEAD_CMAX;
  const bool head = narrow && N <= NB;       // fused into the neighbouring launches
  const bool head_rows = narrow && N > NB;   // row kernels
  const int Lf = narrow ? L - 1 : L;  // layers run by the generic forward loop
  float *hp = nullptr;
  int head_nblk = 0;
  if (head && !(flags & CLO_MLP_NO_PERSISTENT) && mega_ok(L, dims, W, VW, OW, X, N, loss_kind, aux_rank)) {
    float *xch = ws + ggn_ws_mega_offset(L, dims, N);
    unsigned *sync = reinterpret_cast<unsigned *>(xch + cdiv(mega_xch_floats(dims[1], dims[2]), 64) * 64);
    return mega_launch(dims, acts, W, b, VW, Vb, OW, Ob, X, N, loss_kind, aux, aux_rank, loss_scale * alpha, beta,
                       xch, sync, st);
  }
  if (narrow && mid_chain_ok(L, dims, W, VW, OW, N)) {
    // counters of the fused two-layer forward (behind the core workspace, zeroed by clo_mlp_ggn_ws_init)
    unsigned *fsync = mid_fused_shape_ok(L, dims, N) ? reinterpret_cast<unsigned *>(ws + ggn_ws_mega_offset(L, dims, N)) : nullptr;
#define CLO_MID_CHAIN(T)                                                                              \
  mid_chain<T>(L, dims, acts, W, b, VW, Vb, OW, Ob, N, loss_kind, aux, aux_rank, loss_scale * alpha, beta, \
               a, da, dphi, dl, gws, gws_sz, fsync, st)
#ifndef CLO_MLP_MID_MAX
#define CLO_MLP_MID_MAX MID_MAX_N
#endif
    static const int mid_max = CLO_MLP_MID_MAX;
    rc = N > mid_max ? CLO_EUNSUP
         : N <= 16 ? CLO_MID_CHAIN(1) : N <= 32 ? CLO_MID_CHAIN(2) : N <= 48 ? CLO_MID_CHAIN(3) : CLO_MID_CHAIN(4);
#undef CLO_MID_CHAIN
    if (rc != CLO_EUNSUP) return rc;
  }
  // ---- forward + JVP
  for (int l = 1; l <= Lf; ++l) {
    const int di = dims[l - 1], dout = dims[l];
    const float *bl = b ? b[l - 1] : nullptr, *vbl = Vb ? Vb[l - 1] : nullptr;
    if (skinny) {
      // the last layer's split-K slabs are merged by the loss kernel (single 8-row pass only);
      // with a fused head, layer L-1 leaves its slabs for head_fwd_kernel
      const bool defer = (l == L && last_linear && N <= NB) || (head && l == Lf);
      for (int n0 = 0; n0 < N; n0 += NB) {
        const int nn = std::min(NB, N - n0);
        rc = fwd_pass(W[l - 1], bl, VW[l - 1], vbl, a[l - 1] + (long)n0 * di,
                      da[l - 1] ? da[l - 1] + (long)n0 * di : nullptr, a[l] + (long)n0 * dout,
                      da[l] + (long)n0 * dout, dphi[l] + (long)n0 * dout, nn, di, dout,
                      acts[l - 1], part, defer, defer ? &last_ksplit : nullptr, st);
        if (rc != CLO_OK) return rc;
      }
    } else {
      // unaligned operands: a_l = act(a W^T + b) with act' as second output; da_l = act' * (a VW^T +
      // da W^T + Vb): two launches, bias / activation / derivative fused into whichever kernel
      // writes C, the tangent's two products chained along K in one pass
      // preferred: one fused pass over W_l and V_l (three MFMA products per tile, gemm_fwd3_kernel)
      rc = fwd_jvp_gemm(a[l - 1], da[l - 1], W[l - 1], VW[l - 1], bl, vbl, a[l], da[l], dphi[l], N, di, dout,
                        acts[l - 1], gws, gws_sz, st);
      if (rc != CLO_OK) return rc;
    }
  }
  float *dcur, *dnext;
  int lstart = L;
  if (head) {
    const int d = dims[L - 1], C = dims[L];
    head_nblk = (int)cdiv(d, 256);
    hp = gws;  // [N][nblk][2][HEAD_CMAX]; the GEMM slab area is unused on the skinny path
    HeadFwdArgs fa{};
    fa.part = last_ksplit > 1 ? part : nullptr; fa.ksplit = last_ksplit;
    fa.b = b ? b[L - 2] : nullptr; fa.Vb = Vb ? Vb[L - 2] : nullptr;
    fa.a = a[L - 1]; fa.da = da[L - 1]; fa.dphi = dphi[L - 1];
    fa.N = N; fa.d = d; fa.act = acts[L - 2];
    fa.WL = W[L - 1]; fa.VL = VW[L - 1]; fa.C = C; fa.hp = hp;
    {
      ProfScope pf(3, 0.0, st);
      hipLaunchKernelGGL(head_fwd_kernel, dim3(head_nblk, N), dim3(256), 0, st, fa);
      CLO_CHECK_LAUNCH("head_fwd_kernel");
    }
    HeadBwdArgs ba{};
    ba.hp = hp; ba.nblk = head_nblk;
    ba.bL = b ? b[L - 1] : nullptr; ba.VbL = Vb ? Vb[L - 1] : nullptr;
    ba.kind = loss_kind; ba.aux = aux; ba.aux_rank = aux_rank; ba.scale = loss_scale * alpha;
    ba.WL = W[L - 1]; ba.a_prev = a[L - 1]; ba.dphi_prev = dphi[L - 1];
    ba.out_W = OW[L - 1]; ba.out_b = Ob ? Ob[L - 1] : nullptr; ba.delta_prev = dl[L - 1];
    ba.beta = beta; ba.N = N; ba.d = d; ba.C = C;
    {
      ProfScope pf(1, 4.0 * d * C, st);
      hipLaunchKernelGGL(head_bwd_kernel, dim3(head_nblk), dim3(256), 0, st, ba);
      CLO_CHECK_LAUNCH("head_bwd_kernel");
    }
    dcur = dl[L - 1]; dnext = dl0;
    lstart = L - 1;
  } else if (head_rows) {
    const int d = dims[L - 1], C = dims[L];
    // few row blocks: split the features so that ~128 blocks share the two weight rows' stream
    const long row_blocks = cdiv(N, HR_ROWS);
    int hsplit = (int)std::max<long>(1, std::min<long>({128 / row_blocks, cdiv(d, 256), 16L}));
    if ((long)hsplit * 2 * N * C > gws_sz) hsplit = 1;
    const int d_per = (int)cdiv(cdiv(d, hsplit), 256) * 256;
    hsplit = (int)cdiv(d, d_per);
    const dim3 hgrid((unsigned)row_blocks, (unsigned)hsplit);
    if (vec_ok(d, {a[L - 1], da[L - 1], W[L - 1], VW[L - 1]}))
      hipLaunchKernelGGL(head_rows_fwd_kernel<true>, hgrid, dim3(HEAD_CMAX * 64), 0, st, a[L - 1],
                         da[L - 1], W[L - 1], VW[L - 1], b ? b[L - 1] : nullptr,
                         Vb ? Vb[L - 1] : nullptr, a[L], da[L], N, d, C, d_per, gws);
    else
      hipLaunchKernelGGL(head_rows_fwd_kernel<false>, hgrid, dim3(HEAD_CMAX * 64), 0, st, a[L - 1],
                         da[L - 1], W[L - 1], VW[L - 1], b ? b[L - 1] : nullptr,
                         Vb ? Vb[L - 1] : nullptr, a[L], da[L], N, d, C, d_per, gws);
    CLO_CHECK_LAUNCH("head_rows_fwd_kernel");
    if (hsplit > 1)
      rc = launch_loss(loss_kind, a[L], aux, aux_rank, da[L], nullptr, dl0, N, C, loss_scale * alpha,
                       gws, hsplit, b ? b[L - 1] : nullptr, Vb ? Vb[L - 1] : nullptr, a[L], da[L], st, N);
    else
      rc = launch_loss(loss_kind, a[L], aux, aux_rank, da[L], nullptr, dl0, N, C, loss_scale * alpha,
                       nullptr, 1, nullptr, nullptr, a[L], da[L], st);
    if (rc != CLO_OK) return rc;
#ifndef CLO_HEAD_BACK_FUSED
#define CLO_HEAD_BACK_FUSED 1
#endif
    static const int head_back_fused = CLO_HEAD_BACK_FUSED;
    if (head_back_fused && N <= HB_FUSE_N && C <= HEAD_CMAX) {   // (256 rows: 317 us with the one launch, 308 with the four)
      const int nbj = (int)cdiv(d, 64);
      hipLaunchKernelGGL(head_rows_back_kernel, dim3((unsigned)(nbj * (1 + cdiv(N, HB_DROWS)))), dim3(HB_GROUPS * 64), 0, st, dl0,
                         a[L - 1], W[L - 1], dphi[L - 1], OW[L - 1], Ob ? Ob[L - 1] : nullptr, dl1, N, d, C, beta, nbj);
      CLO_CHECK_LAUNCH("head_rows_back_kernel");
    } else {
      rc = launch_small_outer(OW[L - 1], dl0, a[L - 1], N, d, C, beta, gws, gws_sz, st);
      if (rc != CLO_OK) return rc;
      if (Ob && Ob[L - 1]) {
        rc = launch_small_outer(Ob[L - 1], nullptr, dl0, N, C, 1, beta, nullptr, 0, st);
        if (rc != CLO_OK) return rc;
      }
      hipLaunchKernelGGL(head_rows_bwd_kernel, dim3((unsigned)cdiv(d, 256), N), dim3(256), 0, st, dl0,
                         W[L - 1], dphi[L - 1], dl1, d, C);
      CLO_CHECK_LAUNCH("head_rows_bwd_kernel");
    }
    dcur = dl1; dnext = dl0;
    lstart = L - 1;
  } else {
  // ---- output-space curvature: delta_L = dphi_L * (alpha * s * H u)
  {
    const int C = dims[L];
    const bool merge = last_ksplit > 1;
    float *dL = (skinny && N <= NB) ? dl[L] : dl0;
    rc = launch_loss(loss_kind, a[L], aux, aux_rank, da[L], last_linear ? nullptr : dphi[L], dL, N,
                     C, loss_scale * alpha, merge ? part : nullptr, last_ksplit,
                     b ? b[L - 1] : nullptr, Vb ? Vb[L - 1] : nullptr, a[L], da[L], st);
    if (rc != CLO_OK) return rc;
  }
  dcur = (skinny && N <= NB) ? dl[L] : dl0; dnext = dl1;
  }
  // ---- backward, single 8-row pass: data chain (read-only sweeps of W_l) first, then ALL
  // parameter outer products in one write-only launch
  if (skinny && N <= NB) {
    if (lstart > OUTER_MAXL) {
      set_error("clo_mlp_ggn_matvec: more than %d layers", OUTER_MAXL);
      return CLO_EUNSUP;
    }
    int njb1 = 0;  // > 0: delta_1 is still spread over row-range slabs in `part`
    for (int l = lstart; l >= 2; --l) {
      rc = bwd_pass(W[l - 1], dl[l], nullptr, dphi[l - 1], nullptr, nullptr, dl[l - 1], 1.f, 0.f, N,
                    dims[l - 1], dims[l], part, st, l == 2 ? &njb1 : nullptr);
      if (rc != CLO_OK) return rc;
    }
    if (lstart >= 1) {
      OuterAllArgs oa{};
      oa.nlayers = lstart; oa.alpha = 1.f; oa.beta = beta; oa.N = N;
      int nb = 0;
      for (int l = 1; l <= lstart; ++l) {
        const int k = l - 1;
        oa.first_block[k] = nb;
        oa.delta[k] = dl[l]; oa.a_prev[k] = a[l - 1];
        oa.out_W[k] = OW[l - 1]; oa.out_b[k] = Ob ? Ob[l - 1] : nullptr;
        oa.d_in[k] = dims[l - 1]; oa.d_out[k] = dims[l];
        oa.vec[k] = vec_ok(dims[l - 1], {a[l - 1], OW[l - 1]}) ? 1 : 0;
        if (l == 1 && njb1 > 0) { oa.dslabs[k] = part; oa.dnjb[k] = njb1; oa.dphi[k] = dphi[1]; }
        nb += (int)(cdiv(dims[l - 1], CW) * cdiv(dims[l], OUTER_ROWS));
      }
      oa.first_block[lstart] = nb;
      const size_t smem = (size_t)(OUTER_ROWS * NB + NB * CW) * sizeof(float);
      double bytes = 0;
      for (int l = 1; l <= lstart; ++l) bytes += 4.0 * dims[l - 1] * dims[l] * (beta != 0.f ? 2 : 1);
      ProfScope prof(4, bytes, st);
      if (beta != 0.f)
        hipLaunchKernelGGL(outer_all_kernel<true>, dim3(nb), dim3(512), smem, st, oa);
      else
        hipLaunchKernelGGL(outer_all_kernel<false>, dim3(nb), dim3(512), smem, st, oa);
      CLO_CHECK_LAUNCH("outer_all_kernel");
    }
    return CLO_OK;
  }
  // ---- backward (two 8-row passes / GEMM path): layer by layer
#ifndef CLO_MLP_ROWS_OUTER2
#define CLO_MLP_ROWS_OUTER2 1
#endif
  // (measured up to 192 rows, nine to twelve row tiles, one block per CU: 3 ... 15 us SLOWER than the GEMM engine there)
  const bool rows_outer2 = CLO_MLP_ROWS_OUTER2 && !skinny && N <= 128;
  MidOuterArgs pend{};
  pend.alpha = 1.f; pend.beta = beta; pend.N = N;
  auto flush_outer2 = [&]() -> int {
    if (pend.nlayers == 0) return CLO_OK;
    int r;
    switch ((N + 15) / 16) {
      case 5: r = launch_mid_outer2<5>(pend, pend.nlayers, beta, N, st); break;
      case 6: r = launch_mid_outer2<6>(pend, pend.nlayers, beta, N, st); break;
      case 7: r = launch_mid_outer2<7>(pend, pend.nlayers, beta, N, st); break;
      case 8: r = launch_mid_outer2<8>(pend, pend.nlayers, beta, N, st); break;
      default: r = launch_mid_outer2<4>(pend, pend.nlayers, beta, N, st); break;   // (fewer rows on this path: padded to 64)
    }
    pend.nlayers = 0;
    return r;
  };
  for (int l = lstart; l >= 1; --l) {
    const int di = dims[l - 1], dout = dims[l];
    float *obl = Ob ? Ob[l - 1] : nullptr;
    float *dprev = l > 1 ? dnext : nullptr;
    if (skinny) {
      rc = clo_mlp_bwd_layer(W[l - 1], dcur, a[l - 1], dphi[l - 1], OW[l - 1], obl, dprev, 1.f, beta,
                             N, di, dout, part, stream);
      if (rc != CLO_OK) return rc;
    } else {
      // out_W = beta out_W + delta^T a_prev; with an implicit ones column appended to a_prev the
      // extra output column is the bias gradient (column sums of delta): one launch for both
      // up to 128 rows (round 6): the streaming outer-product kernel of the 9 ... 64-row chain, delta^T in registers -- the K = N <= 128
      // product on the GEMM engine is four k tiles per 128 x 128 output tile, all prologue and epilogue (2 x 23 us at 128 rows).
      // delta_l and delta_{l-1} live in the two ping-pong buffers at the same time, so the products of two neighbouring layers
      // share a launch, placed before the step that overwrites delta_l.
      if (rows_outer2 && vec_ok(di, {a[l - 1], OW[l - 1]})) {
        const int k = pend.nlayers++;
        pend.md[k] = MidDelta{dcur, nullptr, nullptr, 0, 0};
        pend.a_prev[k] = a[l - 1]; pend.out_W[k] = OW[l - 1]; pend.out_b[k] = obl; pend.d_in[k] = di; pend.d_out[k] = dout;
        if (pend.nlayers == 2 || l == 1) {
          rc = flush_outer2();
          if (rc != CLO_OK) return rc;
        }
      } else {
      rc = flush_outer2();   // (a pending neighbour's delta is overwritten by this layer's step below)
      if (rc != CLO_OK) return rc;
      GemmArgs go = gemm_problem(dout, di, N, dcur, 1, dout, a[l - 1], di, 1, beta, OW[l - 1], di);
      bool bias_done = false;
      if (obl) {
        GemmArgs gb = go;
        gb.N = di + 1; gb.ones_b = 1; gb.col_out = obl;
        if (gemm_v2_eligible(gb, 1)) {
          go = gb;
          bias_done = true;
        }
      }
      rc = launch_gemm_auto(go, gws, gws_sz, st);
      if (rc != CLO_OK) return rc;
      if (obl && !bias_done) {
        rc = launch_small_outer(obl, nullptr, dcur, N, dout, 1, beta, nullptr, 0, st);
        if (rc != CLO_OK) return rc;
      }
      }
      if (dprev) {  // delta_prev = act'_{l-1} * (delta W)
        GemmArgs gd = gemm_problem(N, di, dout, dcur, dout, 1, W[l - 1], di, 1, 0.f, dprev, di);
        gd.epi = EPI_MUL; gd.e_mul = dphi[l - 1]; gd.ld_mul = di;
        rc = launch_gemm_auto(gd, gws, gws_sz, st);
        if (rc != CLO_OK) return rc;
      }
    }
    std::swap(dcur, dnext);
  }
  return flush_outer2();
}

// K-column exact Hessian (clo_mlp_hessian_matmat): the two elementwise pieces next to the GGN pipeline.
// daP[n][i][k] = dA[i][n][k]: the tangent activations sample-major, the B operand of d_l^T da_{l-1}.
__global__ void kcols_perm_kernel(const float *__restrict__ dA, float *__restrict__ daP, int N, int K, int d) {
  const long total = (long)d * N * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % K), n = (int)((e / K) % N);
    const long i = e / ((long)K * N);
    daP[((long)n * d + i) * K + k] = dA[e];
  }
}
// Rd_{l-1}[i][n][k] = (phi'' dz)[i][n][k] dsig[n][i] + phi'[n][i] (T[i][n][k] + Xv[n][i][k]) in place on dA = da_{l-1};
// d_{l-1}[n][i] = phi'[n][i] dsig[n][i].   (T = W_l^T Rd_l, Xv = d_l V_l, dsig = d_l W_l)
__global__ void kcols_hess_combine_kernel(float *__restrict__ dA, const float *__restrict__ T,
                                          const float *__restrict__ Xv, const float *__restrict__ dsig,
                                          const float *__restrict__ a, const float *__restrict__ dphi,
                                          float *__restrict__ d_out, int N, int K, int d, int act) {
  const long total = (long)d * N * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % K), n = (int)((e / K) % N);
    const long i = e / ((long)K * N), ni = (long)n * d + i;
    const float g = dsig[ni], ph = dphi[ni];
    dA[e] = act_second_times_dz(act, a[ni], dA[e]) * g + ph * (T[e] + Xv[ni * K + k]);
    if (k == 0) d_out[ni] = ph * g;
  }
}

// ------------------------------------------------------------------------------------------
// K columns at once (see the kfwd / kouter kernels above).
// ------------------------------------------------------------------------------------------
static long matmat_gemm_ws(int dmax, int K) { return 16L * dmax * NB * K; }

static long matmat_hessian_extra_ws(int L, const int *dims, int K) {
  int dmax = 0;
  long total = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  for (int l = 1; l <= L; ++l) total += (long)NB * dims[l];      // d_l
  return total + (long)NB * dmax + 3L * dmax * NB * K + 64;        // dsig, T, Xv, daP
}
extern "C" long clo_mlp_ggn_matmat_ws_floats(int L, const int *dims, int N, int K);
extern "C" long clo_mlp_hessian_matmat_ws_floats(int L, const int *dims, int N, int K) {
  if (L <= 0 || !dims || K <= 0) return 0;
  return clo_mlp_ggn_matmat_ws_floats(L, dims, N, K) + matmat_hessian_extra_ws(L, dims, K);
}

extern "C" long clo_mlp_ggn_matmat_ws_floats(int L, const int *dims, int N, int K) {
  (void)N;
  if (L <= 0 || !dims || K <= 0) return 0;
  long total = 0;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  for (int l = 1; l <= L; ++l) total += 2L * NB * dims[l] + (long)dims[l] * NB * K;  // a, phi', dA
  for (int l = 0; l < L; ++l) total += (long)dims[l] * NB;                            // aT
  long part = 0;
  for (int l = 1; l <= L; ++l) part = std::max(part, clo_mlp_fwd_ws_floats(NB, dims[l - 1], dims[l]));
  return total + part + matmat_gemm_ws(dmax, K) + 256;
}

// out[.., k] = beta out[.., k] + alpha (J^T H J) V[.., k] for K columns.  VW[l] / OW[l] point at
// element (0, 0, 0) of the [d_out][d_in][K] blocks (column stride 1, row stride ldk floats), Vb[l]
// / Ob[l] at [d_out][K] blocks with the same ldk.  Requirements (else CLO_EUNSUP): K % 4 == 0,
// 4 <= K <= 64, ldk % 4 == 0, dims[0..L-1] % 4 == 0, 16-byte aligned operands.
// Gh == nullptr: GGN / EF / MC-GGN columns.  Gh [N][C] (gradient of the reduced mini-batch loss w.r.t. the model output):
// exact Hessian columns by the R-operator (hessian.py:66 under the vmap of _torch_base.py:946-989) -- the same
// pipeline plus, per layer, the gradient signal d_l (column independent), the product d_l V_l (the tangent weights
// stream a second time through the GEMM engine, M = 8 rows), the combine of Rd_{l-1} and d_l^T da_{l-1} added to the result
// (a rank-8 product over the [d_out][d_in K] block, written before the outer-product stream accumulates onto it).
static int mlp_matmat_impl(const char *what, int L, const int *dims, const int *acts, const float *const *W,
                           const float *const *b, const float *const *VW,
                           const float *const *Vb, float *const *OW, float *const *Ob,
                           long ldk, const float *X, int N, int K, int loss_kind,
                           const float *aux, int aux_rank, float loss_scale, float alpha,
                           float beta, float *ws, void *stream, const float *Gh) {
  CLO_REQUIRE(L >= 1 && L <= 64 && dims && acts && W && VW && OW, "clo_mlp_ggn_matmat: bad layer table");
  CLO_REQUIRE(N >= 0 && X && ws && K >= 1, "clo_mlp_ggn_matmat: bad batch / workspace / K");
  CLO_REQUIRE(loss_kind >= 0 && loss_kind <= CLO_LOSS_EF_BCE, "clo_mlp_ggn_matmat: unknown loss kind %d", loss_kind);
  CLO_REQUIRE((loss_kind != CLO_LOSS_RANK1 && !loss_is_ef(loss_kind)) || (aux && aux_rank >= 1),
              "clo_mlp_ggn_matmat: RANK1 / EF_* need aux (backpropagated vectors / targets) and aux_rank >= 1");
  bool ok = K % 4 == 0 && K >= 4 && K <= 64 && ldk % 4 == 0 && ldk >= K && aligned16(X) && aligned16(ws);
  for (int l = 0; l <= L; ++l) CLO_REQUIRE(dims[l] > 0, "clo_mlp_ggn_matmat: dims[%d] <= 0", l);
  for (int l = 0; l < L; ++l) {
    CLO_REQUIRE(acts[l] >= 0 && acts[l] <= 3, "clo_mlp_ggn_matmat: unknown activation");
    CLO_REQUIRE(W[l] && VW[l] && OW[l], "clo_mlp_ggn_matmat: null weight pointer in layer %d", l);
    ok = ok && dims[l] % 4 == 0 && aligned16(W[l]) && aligned16(VW[l]) && aligned16(OW[l]);
    if (Vb && Vb[l]) ok = ok && aligned16(Vb[l]);
    if (Ob && Ob[l]) ok = ok && aligned16(Ob[l]);
  }
  if (loss_kind == CLO_LOSS_RANK1 && aux_rank > LC_RMAX) ok = false;
  if (!ok) {
    set_error("%s: needs K %% 4 == 0, 4 <= K <= 64, ldk %% 4 == 0, layer inputs %% 4 == 0 "
              "and 16-byte aligned operands", what);
    return CLO_EUNSUP;
  }
  if (Gh && (ldk != K || acts[L - 1] != CLO_ACT_IDENTITY || loss_kind == CLO_LOSS_RANK1)) {
    set_error("%s: needs ldk == K, a linear last layer and an MSE / CE / BCE loss", what);
    return CLO_EUNSUP;
  }
  hipStream_t st = (hipStream_t)stream;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  if (N == 0) {  // empty batch: out = beta * out, row by row of the strided blocks
    if (beta != 1.f)
      for (int l = 0; l < L; ++l) {
        const long rows = (long)dims[l] * dims[l + 1];
        for (int pass = 0; pass < 2; ++pass) {
          float *o = pass == 0 ? OW[l] : (Ob ? Ob[l] : nullptr);
          const long r = pass == 0 ? rows : dims[l + 1];
          if (!o) continue;
          if (ldk == K) {
            int rc = clo_axpby_f32(o, o, r * K, 0.f, beta, stream);
            if (rc != CLO_OK) return rc;
          } else {
            for (long q = 0; q < r; ++q) {
              int rc = clo_axpby_f32(o + q * ldk, o + q * ldk, K, 0.f, beta, stream);
              if (rc != CLO_OK) return rc;
            }
          }
        }
      }
    return CLO_OK;
  }
  // carve the workspace
  float *p = ws;
  float *a[65], *dphi[65], *dA[65], *aT[65];
  for (int l = 1; l <= L; ++l) {
    a[l] = p; p += (long)NB * dims[l];
    dphi[l] = p; p += (long)NB * dims[l];
    dA[l] = p; p += (long)dims[l] * NB * K;
  }
  for (int l = 0; l < L; ++l) { aT[l] = p; p += (long)dims[l] * NB; }
  float *part = p;
  long part_sz = 0;
  for (int l = 1; l <= L; ++l) part_sz = std::max(part_sz, clo_mlp_fwd_ws_floats(NB, dims[l - 1], dims[l]));
  p += part_sz;
  float *gws = p;
  const long gws_sz = matmat_gemm_ws(dmax, K);
  p += gws_sz + 64;
  float *dH[65] = {nullptr}, *dsig = nullptr, *Tb = nullptr, *Xv = nullptr, *daP = nullptr;
  if (Gh) {
    for (int l = 1; l <= L; ++l) { dH[l] = p; p += (long)NB * dims[l]; }
    dsig = p; p += ((long)NB * dmax + 3) & ~3L;
    Tb = p; p += (long)dmax * NB * K;
    Xv = p; p += (long)dmax * NB * K;
    daP = p; p += (long)dmax * NB * K;
  }
  const int G = K / 4, FPT = 16 / G;
  const bool last_linear = acts[L - 1] == CLO_ACT_IDENTITY;
#ifndef CLO_KC_LAST
#define CLO_KC_LAST 1
#endif
  static const int kc_last = CLO_KC_LAST;   // narrow last layer: klast_* kernels instead of GEMM engine + weight stream

  for (int n0 = 0; n0 < N; n0 += NB) {
    const int nn = std::min(NB, N - n0);
    const int NK = nn * K;
    const float bt = n0 == 0 ? beta : 1.f;
    a[0] = const_cast<float *>(X) + (long)n0 * dims[0];
    // ---- forward: z path, tangent GEMM, tangent-weight stream
    for (int l = 1; l <= L; ++l) {
      const int di = dims[l - 1], dout = dims[l];
      int rc = fwd_pass(W[l - 1], b ? b[l - 1] : nullptr, nullptr, nullptr, a[l - 1], nullptr, a[l],
                        nullptr, dphi[l], nn, di, dout, acts[l - 1], part, false, nullptr, st);
      if (rc != CLO_OK) return rc;
      const int klast_slabs = (int)cdiv(di, KL_ROWS);
      if (kc_last && l == L && l >= 2 && dout <= KL_CMAX && nn * G <= 128 &&
          (long)klast_slabs * dout * NK <= gws_sz) {   // narrow last layer: tangent GEMM + weight stream in one pass
        const float *dpl = last_linear ? nullptr : dphi[l];
        if (nn * G <= 64)
          hipLaunchKernelGGL(klast_partial_kernel<4>, dim3(klast_slabs), dim3(256), 0, st, VW[l - 1], ldk, W[l - 1], a[l - 1],
                             dA[l - 1], gws, nn, K, di, dout);
        else
          hipLaunchKernelGGL(klast_partial_kernel<2>, dim3(klast_slabs), dim3(256), 0, st, VW[l - 1], ldk, W[l - 1], a[l - 1],
                             dA[l - 1], gws, nn, K, di, dout);
        CLO_CHECK_LAUNCH("klast_partial_kernel");
        hipLaunchKernelGGL(klast_finish_kernel, dim3((unsigned)cdiv((long)dout * nn * G, 4)), dim3(256), 0, st, gws,
                           klast_slabs, Vb ? Vb[l - 1] : nullptr, ldk, dpl, dA[l], nn, K, dout);
        CLO_CHECK_LAUNCH("klast_finish_kernel");
        continue;
      }
      if (l >= 2) {  // dA_l = W_l dA_{l-1}   ([dout x di] [di x NK])
        GemmArgs g = gemm_problem(dout, NK, di, W[l - 1], di, 1, dA[l - 1], NK, 1, 0.f, dA[l], NK);
        rc = launch_gemm_auto(g, gws, gws_sz, st);
        if (rc != CLO_OK) return rc;
      }
#ifndef CLO_KC_BPC
#define CLO_KC_BPC 4
#endif
      static const int kc_bpc = CLO_KC_BPC;
      // equal tile counts per block: with `slots` blocks in flight ceil(tiles / slots) rounds of tiles are needed anyway, so
      // the tiles are dealt to tiles / rounds blocks (d_out = 2688, K = 32: 1344 tiles on 672 blocks x 2 instead of 1024
      // blocks with 1 or 2; 96.8 -> 92.6 us per launch)
      const long kc_tiles = cdiv(dout, FPT * KC_TPW);
      dim3 grid((unsigned)cdiv(kc_tiles, cdiv(kc_tiles, (long)kc_bpc * kNumCU))), block(KC_WAVES * 64);
      const float *vb = Vb ? Vb[l - 1] : nullptr;
      const float *dp = (l == L && last_linear) ? nullptr : dphi[l];
      ProfScope prof(0, 4.0 * di * dout * K, st);
      if (l >= 2)
        hipLaunchKernelGGL((kfwd_stream_kernel<true, KC_TPW, KC_WAVES>), grid, block, 0, st, VW[l - 1], ldk, vb,
                           a[l - 1], dp, dA[l], nn, K, di, dout);
      else
        hipLaunchKernelGGL((kfwd_stream_kernel<false, KC_TPW, KC_WAVES>), grid, block, 0, st, VW[l - 1], ldk, vb,
                           a[l - 1], dp, dA[l], nn, K, di, dout);
      CLO_CHECK_LAUNCH("kfwd_stream_kernel");
    }
    // ---- layer inputs sample-minor for the result streams (needed from here on): 8 layers per launch
    for (int l0 = 0; l0 < L; l0 += 8) {
      PackMulti pm{};
      pm.nl = std::min(8, L - l0);
      pm.N = nn;
      for (int q = 0; q < pm.nl; ++q) {
        pm.a[q] = a[l0 + q];
        pm.d[q] = dims[l0 + q];
        pm.start[q] = aT[l0 + q] - aT[l0];
      }
      pm.start[pm.nl] = pm.start[pm.nl - 1] + (long)dims[l0 + pm.nl - 1] * NB;
      hipLaunchKernelGGL(pack_at_multi_kernel, dim3((unsigned)cdiv(pm.start[pm.nl], 256)), dim3(256), 0, st, pm, aT[l0]);
      CLO_CHECK_LAUNCH("pack_at_multi_kernel");
    }
    // ---- output-space curvature per (n, k), in place: dA_L becomes delta_L
    {
      const float *auxn = !aux ? nullptr : loss_is_ef(loss_kind) ? aux + ef_target_floats(loss_kind, n0, dims[L]) : aux + (long)n0 * aux_rank * dims[L];
      hipLaunchKernelGGL(loss_cols_kernel, dim3((unsigned)cdiv(NK, 64)), dim3(64), 0, st, loss_kind, a[L],
                         auxn, aux_rank, last_linear ? nullptr : dphi[L], dA[L], nn, K, dims[L],
                         loss_scale * alpha);
      CLO_CHECK_LAUNCH("loss_cols_kernel");
    }
    if (Gh) {  // d_L = alpha G (linear last layer)
      const long nc = (long)nn * dims[L];
      hipLaunchKernelGGL(scale_copy_kernel, dim3(ew_grid(nc)), dim3(256), 0, st, dH[L], Gh + (long)n0 * dims[L], nc, alpha);
      CLO_CHECK_LAUNCH("scale_copy_kernel");
    }
    // ---- backward: result stream, then the delta GEMM of the next layer down
    for (int l = L; l >= 1; --l) {
      const int di = dims[l - 1], dout = dims[l];
      float bw = bt;   // beta of the weight block in the outer-product stream
      if (Gh && l >= 2) {
        // out_W_l = bt out_W_l + d_l^T da_{l-1} first (da_{l-1} sample-major), the stream then adds Rd_l^T a_{l-1}
        const long nel = (long)di * nn * K;
        hipLaunchKernelGGL(kcols_perm_kernel, dim3(ew_grid(nel)), dim3(256), 0, st, dA[l - 1], daP, nn, K, di);
        CLO_CHECK_LAUNCH("kcols_perm_kernel");
        GemmArgs g2 = gemm_problem(dout, di * K, nn, dH[l], 1, dout, daP, (long)di * K, 1, bt, OW[l - 1], (long)di * K);
        int rc = launch_gemm_auto(g2, gws, gws_sz, st);
        if (rc != CLO_OK) return rc;
        bw = 1.f;
      }
      {
        ProfScope prof(4, 4.0 * di * dout * K * (bw != 0.f ? 2 : 1), st);
        float *ob = Ob ? Ob[l - 1] : nullptr;
#ifndef CLO_KO_TRIPS
#define CLO_KO_TRIPS 1
#endif
        // rows of a chunk: CLO_KO_TRIPS trips of 16 wave instructions = 16 KB each (0: the whole input range per block)
        const int trip_rows = 4 * CLO_KO_U * (64 / (K >> 2));
        const int chunk_rows = CLO_KO_TRIPS > 0 ? CLO_KO_TRIPS * trip_rows : di;
        const int nchunk = (int)cdiv(di, chunk_rows);
        const dim3 kgrid((unsigned)((long)dout * nchunk));
        if (bw != 0.f)
          hipLaunchKernelGGL(kouter_stream_kernel<true>, kgrid, dim3(256), 0, st, OW[l - 1], ldk, ob,
                             aT[l - 1], dA[l], nn, K, di, bw, bt, nchunk, chunk_rows);
        else
          hipLaunchKernelGGL(kouter_stream_kernel<false>, kgrid, dim3(256), 0, st, OW[l - 1], ldk,
                             ob, aT[l - 1], dA[l], nn, K, di, bw, bt, nchunk, chunk_rows);
        CLO_CHECK_LAUNCH("kouter_stream_kernel");
      }
      if (l >= 2 && !Gh && kc_last && l == L && dout <= KL_CMAX) {
        const long quads = (long)di * (NK >> 2);
        hipLaunchKernelGGL(klast_delta_kernel, dim3((unsigned)std::min<long>(cdiv(quads, 256), 8L * kNumCU)), dim3(256), 0, st,
                           W[l - 1], dA[l], dphi[l - 1], dA[l - 1], nn, K, di, dout);
        CLO_CHECK_LAUNCH("klast_delta_kernel");
      } else if (l >= 2 && !Gh) {  // delta_{l-1} = phi'_{l-1} * (W_l^T delta_l)   ([di x dout] [dout x NK])
        GemmArgs g = gemm_problem(di, NK, dout, W[l - 1], 1, di, dA[l], NK, 1, 0.f, dA[l - 1], NK);
        g.epi = EPI_MUL_T; g.e_mul = dphi[l - 1]; g.ld_mul = di; g.e_div = K;
        int rc = launch_gemm_auto(g, gws, gws_sz, st);
        if (rc != CLO_OK) return rc;
      } else if (l >= 2) {
        // T = W_l^T Rd_l ; dsig = d_l W_l ; Xv = d_l V_l ; Rd_{l-1}, d_{l-1} by the combine (in place on da_{l-1})
        GemmArgs g = gemm_problem(di, NK, dout, W[l - 1], 1, di, dA[l], NK, 1, 0.f, Tb, NK);
        int rc = launch_gemm_auto(g, gws, gws_sz, st);
        if (rc != CLO_OK) return rc;
        GemmArgs gs = gemm_problem(nn, di, dout, dH[l], dout, 1, W[l - 1], di, 1, 0.f, dsig, di);
        rc = launch_gemm_auto(gs, gws, gws_sz, st);
        if (rc != CLO_OK) return rc;
        GemmArgs gv = gemm_problem(nn, di * K, dout, dH[l], dout, 1, VW[l - 1], (long)di * K, 1, 0.f, Xv, (long)di * K);
        rc = launch_gemm_auto(gv, gws, gws_sz, st);
        if (rc != CLO_OK) return rc;
        const long nel = (long)di * nn * K;
        hipLaunchKernelGGL(kcols_hess_combine_kernel, dim3(ew_grid(nel)), dim3(256), 0, st, dA[l - 1], Tb, Xv, dsig,
                           a[l - 1], dphi[l - 1], dH[l - 1], nn, K, di, acts[l - 2]);
        CLO_CHECK_LAUNCH("kcols_hess_combine_kernel");
      }
    }
  }
  return CLO_OK;
}

extern "C" int clo_mlp_ggn_matmat(int L, const int *dims, const int *acts, const float *const *W,
                                  const float *const *b, const float *const *VW,
                                  const float *const *Vb, float *const *OW, float *const *Ob,
                                  long ldk, const float *X, int N, int K, int loss_kind,
                                  const float *aux, int aux_rank, float loss_scale, float alpha,
                                  float beta, float *ws, void *stream) {
  return mlp_matmat_impl("clo_mlp_ggn_matmat", L, dims, acts, W, b, VW, Vb, OW, Ob, ldk, X, N, K, loss_kind, aux, aux_rank,
                         loss_scale, alpha, beta, ws, stream, nullptr);
}

// out[.., k] = beta out[.., k] + alpha H V[.., k] (exact Hessian) for K columns in the K-trailing layout; G [N][C] =
// gradient of the reduced mini-batch loss w.r.t. the model output.  Requirements as clo_mlp_ggn_matmat plus ldk == K, a
// linear last layer and loss_kind in {MSE, CE, BCE} (else CLO_EUNSUP).  ws: clo_mlp_hessian_matmat_ws_floats floats.
extern "C" int clo_mlp_hessian_matmat(int L, const int *dims, const int *acts, const float *const *W,
                                      const float *const *b, const float *const *VW,
                                      const float *const *Vb, float *const *OW, float *const *Ob,
                                      long ldk, const float *X, int N, int K, const float *G, int loss_kind,
                                      float loss_scale, float alpha, float beta, float *ws, void *stream) {
  CLO_REQUIRE(G, "clo_mlp_hessian_matmat: null output gradient");
  return mlp_matmat_impl("clo_mlp_hessian_matmat", L, dims, acts, W, b, VW, Vb, OW, Ob, ldk, X, N, K, loss_kind, nullptr, 1,
                         loss_scale, alpha, beta, ws, stream, G);
}

// ------------------------------------------------------------------------------------------
// Hessian-vector product (see hess_combine_kernel).
// ------------------------------------------------------------------------------------------
extern "C" long clo_mlp_hessian_ws_floats(int L, const int *dims, int N) {
  if (L <= 0 || !dims || N < 0) return 0;
  long total = 0;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  for (int l = 1; l <= L; ++l) total += 3L * N * dims[l];  // a, da, phi'
  total += 6L * N * dmax;                                  // d, Rd (ping/pong), dA, T
  return total + gemm_ws_floats(N, dmax) + 256;
}

// out = beta out + alpha H v for one mini-batch.  G [N][C]: gradient of the (reduced) mini-batch
// loss w.r.t. the model output; loss_kind / aux / loss_scale describe its Hessian as in
// clo_mlp_ggn_matvec.  Any widths (float4-complete layer inputs and 16-byte aligned operands take the fast
// kernel variants).
extern "C" int clo_mlp_hessian_matvec(int L, const int *dims, const int *acts, const float *const *W,
                                      const float *const *b, const float *const *VW,
                                      const float *const *Vb, float *const *OW, float *const *Ob,
                                      const float *X, int N, const float *G, int loss_kind,
                                      const float *aux, int aux_rank, float loss_scale, float alpha,
                                      float beta, float *ws, void *stream) {
  CLO_REQUIRE(L >= 1 && L <= 64 && dims && acts && W && VW && OW, "clo_mlp_hessian_matvec: bad layer table");
  CLO_REQUIRE(N >= 1 && X && G && ws, "clo_mlp_hessian_matvec: bad batch / gradient / workspace");
  CLO_REQUIRE(loss_kind >= 0 && loss_kind <= 3, "clo_mlp_hessian_matvec: unknown loss kind %d", loss_kind);
  for (int l = 0; l <= L; ++l) CLO_REQUIRE(dims[l] > 0, "clo_mlp_hessian_matvec: dims[%d] <= 0", l);
  for (int l = 0; l < L; ++l) {
    CLO_REQUIRE(acts[l] >= 0 && acts[l] <= 3, "clo_mlp_hessian_matvec: unknown activation");
    CLO_REQUIRE(W[l] && VW[l] && OW[l], "clo_mlp_hessian_matvec: null weight pointer in layer %d", l);
  }
  // (layer inputs that are not multiples of 4 / operands that are not 16-byte aligned: every kernel below
  // picks its scalar-load variant, the GEMM engine its unaligned tile loader)
  hipStream_t st = (hipStream_t)stream;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  float *p = ws;
  float *a[65], *da[65], *dphi[65];
  a[0] = const_cast<float *>(X); da[0] = nullptr; dphi[0] = nullptr;
  for (int l = 1; l <= L; ++l) {
    const long sz = (long)N * dims[l];
    a[l] = p; p += sz; da[l] = p; p += sz; dphi[l] = p; p += sz;
  }
  const long nd = (long)N * dmax;
  float *d0 = p, *d1 = p + nd, *R0 = p + 2 * nd, *R1 = p + 3 * nd, *dAb = p + 4 * nd, *Tb = p + 5 * nd;
  p += 6 * nd;
  float *gws = p;
  const long gws_sz = gemm_ws_floats(N, dmax);
  int rc;
  // ---- up to 8 rows: the weight-streaming kernels of the GGN chain instead of GEMM tiles (each
  // weight matrix is the traffic; the GEMM engine spends ~10 us per launch on 8-row products).
  // Slabs, an all-ones mask and the d V product live in the (otherwise unused) GEMM slab region.
  long part_sz = 0;
  for (int l = 1; l <= L; ++l)
    part_sz = std::max({part_sz, clo_mlp_bwd_ws_floats(N, dims[l - 1], dims[l]),
                        clo_mlp_fwd_ws_floats(N, dims[l - 1], dims[l])});
  part_sz = (part_sz + 3) & ~3L;
#ifndef CLO_HESSIAN_GEMM
#define CLO_HESSIAN_GEMM 0
#endif
  static const int no_skinny = CLO_HESSIAN_GEMM;
  const bool skinny = N <= SKINNY_MAX_N && !no_skinny && 3 * part_sz + 2 * nd <= gws_sz;
  float *part = gws, *part2 = gws + part_sz, *part3 = gws + 2 * part_sz, *ones = gws + 3 * part_sz, *T2b = ones + nd;
  if (skinny) {
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(nd)), dim3(256), 0, st, ones, nd, 1.f);
    CLO_CHECK_LAUNCH("fill_kernel");
  }
  // ---- tangent forward pass, one fused launch per layer
  for (int l = 1; l <= L; ++l) {
    if (skinny)
      rc = fwd_pass(W[l - 1], b ? b[l - 1] : nullptr, VW[l - 1], Vb ? Vb[l - 1] : nullptr, a[l - 1], da[l - 1],
                    a[l], da[l], dphi[l], N, dims[l - 1], dims[l], acts[l - 1], part, false, nullptr, st);
    else
      rc = fwd_jvp_gemm(a[l - 1], da[l - 1], W[l - 1], VW[l - 1], b ? b[l - 1] : nullptr,
                        Vb ? Vb[l - 1] : nullptr, a[l], da[l], dphi[l], N, dims[l - 1], dims[l], acts[l - 1], gws,
                        gws_sz, st);
    if (rc != CLO_OK) return rc;
  }
  // ---- output layer: dA = alpha G, T = alpha s H(f) Jv  ->  d_L, Rd_L
  const int C = dims[L];
  const long nc = (long)N * C;
  hipLaunchKernelGGL(scale_copy_kernel, dim3(ew_grid(nc)), dim3(256), 0, st, dAb, G, nc, alpha);
  CLO_CHECK_LAUNCH("scale_copy_kernel");
  rc = launch_loss(loss_kind, a[L], aux, aux_rank, da[L], nullptr, Tb, N, C, loss_scale * alpha, nullptr, 1,
                   nullptr, nullptr, a[L], da[L], st);
  if (rc != CLO_OK) return rc;
  float *dcur = d0, *dnext = d1, *Rcur = R0, *Rnext = R1;
  hipLaunchKernelGGL(hess_combine_kernel, dim3(ew_grid(nc)), dim3(256), 0, st, dAb, Tb, a[L], da[L], dphi[L],
                     dcur, Rcur, nc, acts[L - 1], nullptr);
  CLO_CHECK_LAUNCH("hess_combine_kernel");
  // ---- backward
  for (int l = L; l >= 1; --l) {
    const int di = dims[l - 1], dout = dims[l];
    if (skinny) {
      // three weight-streaming passes: (Rd^T a_prev -> out_W, col sums -> out_b, Rd W -> T),
      // (d^T da_prev -> out_W +=, d W -> dA), (d V -> T2); the mask of the fused kernel is all ones
      const bool more = l > 1;
      int n1 = 0, n2 = 0, n3 = 0;  // > 0: the product is still spread over that many slabs
      rc = bwd_pass(W[l - 1], Rcur, a[l - 1], ones, OW[l - 1], Ob ? Ob[l - 1] : nullptr, more ? Tb : nullptr,
                    1.f, beta, N, di, dout, part, st, &n1);
      if (rc != CLO_OK) return rc;
      if (!more) break;
      rc = bwd_pass(W[l - 1], dcur, da[l - 1], ones, OW[l - 1], nullptr, dAb, 1.f, 1.f, N, di, dout, part2, st, &n2);
      if (rc != CLO_OK) return rc;
      rc = bwd_pass(VW[l - 1], dcur, nullptr, ones, nullptr, nullptr, T2b, 1.f, 0.f, N, di, dout, part3, st, &n3);
      if (rc != CLO_OK) return rc;
      const long ne = (long)N * di;
      HessSlabs hs{n2 ? part2 : dAb, n1 ? part : Tb, n3 ? part3 : T2b, n2, n1, n3};
      hipLaunchKernelGGL(hess_combine_slabs_kernel, dim3(ew_grid(ne)), dim3(256), 0, st, hs, a[l - 1], da[l - 1],
                         dphi[l - 1], dnext, Rnext, N, di, acts[l - 2]);
      CLO_CHECK_LAUNCH("hess_combine_slabs_kernel");
      std::swap(dcur, dnext);
      std::swap(Rcur, Rnext);
      continue;
    }
    // out_W = beta out_W + Rd^T a_prev (+ d^T da_prev)
    GemmArgs go = gemm_problem(dout, di, N, Rcur, 1, dout, a[l - 1], di, 1, beta, OW[l - 1], di);
    rc = launch_gemm_auto(go, gws, gws_sz, st);
    if (rc != CLO_OK) return rc;
    if (da[l - 1]) {
      GemmArgs g2 = gemm_problem(dout, di, N, dcur, 1, dout, da[l - 1], di, 1, 1.f, OW[l - 1], di);
      rc = launch_gemm_auto(g2, gws, gws_sz, st);
      if (rc != CLO_OK) return rc;
    }
    if (Ob && Ob[l - 1]) {
      rc = launch_small_outer(Ob[l - 1], nullptr, Rcur, N, dout, 1, beta, nullptr, 0, st);
      if (rc != CLO_OK) return rc;
    }
    if (l == 1) break;
    // dA = d W_l ;  T = Rd W_l + d V_l
    GemmArgs ga = gemm_problem(N, di, dout, dcur, dout, 1, W[l - 1], di, 1, 0.f, dAb, di);
    rc = launch_gemm_auto(ga, gws, gws_sz, st);
    if (rc != CLO_OK) return rc;
    GemmArgs gt = gemm_problem(N, di, dout, Rcur, dout, 1, W[l - 1], di, 1, 0.f, Tb, di);
    GemmArgs gc = gt;
    gc.K = 2 * dout; gc.K1 = dout; gc.A2 = dcur; gc.B2 = VW[l - 1];
    if (dout % 32 == 0 && gemm_v2_eligible(gc, 1)) {
      rc = launch_gemm_auto(gc, gws, gws_sz, st);
    } else {
      rc = launch_gemm_auto(gt, gws, gws_sz, st);
      if (rc != CLO_OK) return rc;
      GemmArgs gv = gemm_problem(N, di, dout, dcur, dout, 1, VW[l - 1], di, 1, 1.f, Tb, di);
      rc = launch_gemm_auto(gv, gws, gws_sz, st);
    }
    if (rc != CLO_OK) return rc;
    const long ne = (long)N * di;
    hipLaunchKernelGGL(hess_combine_kernel, dim3(ew_grid(ne)), dim3(256), 0, st, dAb, Tb, a[l - 1],
                       da[l - 1], dphi[l - 1], dnext, Rnext, ne, acts[l - 2], nullptr);
    CLO_CHECK_LAUNCH("hess_combine_kernel");
    std::swap(dcur, dnext);
    std::swap(Rcur, Rnext);
  }
  return CLO_OK;
}

// ------------------------------------------------------------------------------------------
// Jacobian / transposed-Jacobian products of an MLP (reference jacobian.py:14-358): the forward +
// JVP half and the VJP half of the GGN product, exposed on their own.
// ------------------------------------------------------------------------------------------
extern "C" long clo_mlp_jac_ws_floats(int L, const int *dims, int N) {
  if (L <= 0 || !dims || N < 0) return 0;
  long total = 0;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  for (int l = 1; l <= L; ++l) total += 3L * N * dims[l];
  return total + 2L * N * dmax + gemm_ws_floats(N, dmax) + 256;
}

// JV[n][c] = (J_theta f(x_n) v)[c]: tangent forward pass (one fused launch per layer; layers whose inputs are
// not float4-complete: two or three plain products).  Any widths / alignment.
extern "C" int clo_mlp_jvp(int L, const int *dims, const int *acts, const float *const *W,
                           const float *const *b, const float *const *VW, const float *const *Vb,
                           const float *X, int N, float *JV, float *ws, void *stream) {
  CLO_REQUIRE(L >= 1 && L <= 64 && dims && acts && W && VW, "clo_mlp_jvp: bad layer table");
  CLO_REQUIRE(N >= 1 && X && JV && ws, "clo_mlp_jvp: bad batch / output / workspace");
  for (int l = 0; l < L; ++l) {
    CLO_REQUIRE(dims[l] > 0 && dims[l + 1] > 0 && acts[l] >= 0 && acts[l] <= 3, "clo_mlp_jvp: bad layer %d", l);
    CLO_REQUIRE(W[l] && VW[l], "clo_mlp_jvp: null weight pointer in layer %d", l);
  }
  hipStream_t st = (hipStream_t)stream;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  float *p = ws;
  float *a[65], *da[65], *dphi[65];
  a[0] = const_cast<float *>(X); da[0] = nullptr;
  for (int l = 1; l <= L; ++l) {
    const long sz = (long)N * dims[l];
    a[l] = p; p += sz; da[l] = p; p += sz; dphi[l] = p; p += sz;
  }
  da[L] = JV;
  p += 2L * N * dmax;
  float *gws = p;
  const long gws_sz = gemm_ws_floats(N, dmax);
  for (int l = 1; l <= L; ++l) {
    int rc = fwd_jvp_gemm(a[l - 1], da[l - 1], W[l - 1], VW[l - 1], b ? b[l - 1] : nullptr,
                          Vb ? Vb[l - 1] : nullptr, a[l], da[l], dphi[l], N, dims[l - 1], dims[l], acts[l - 1], gws,
                          gws_sz, st);
    if (rc != CLO_OK) return rc;
  }
  return CLO_OK;
}

// G[n][c] = scale * d l_n / d f_n[c] for the MSE / CE / BCE loss of the prediction f = net(X): plain forward pass on the GEMM
// engine, then the per-sample loss gradient from (f, targets) as in csrc/mlp_loss.h (targets [N][C] floats, CE: [N] labels
// stored as floats).  What the exact-Hessian products take as `G` (hessian.py:13-69 differentiates the loss itself; the
// reference re-evaluates model and loss on every product) -- computed on the device from the LIVE parameters instead of a
// forward + autograd pass on the host.  ws: clo_mlp_jac_ws_floats(L, dims, N) floats.
__global__ __launch_bounds__(256) void loss_grad_kernel(int ef_kind, const float *__restrict__ f, const float *__restrict__ tgt,
                                                        float *__restrict__ out, int C, float scale) {
  __shared__ float s_red[8];
  const int n = blockIdx.x;
  const float *fn = f + (long)n * C, *t = ef_target_row(ef_kind, tgt, n, C);
  float mx = -INFINITY, inv = 0.f;
  if (ef_kind == CLO_LOSS_EF_CE) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) mx = fmaxf(mx, fn[c]);
    mx = block_max(mx, s_red);
    float se = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) se += __expf(fn[c] - mx);
    inv = 1.f / block_sum(se, s_red);
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) out[(long)n * C + c] = scale * ef_grad_at(ef_kind, fn[c], t, c, mx, inv);
}

extern "C" int clo_mlp_loss_grad(int L, const int *dims, const int *acts, const float *const *W, const float *const *b,
                                 const float *X, int N, int loss_kind, const float *targets, float scale, float *G,
                                 float *ws, void *stream) {
  CLO_REQUIRE(L >= 1 && L <= 64 && dims && acts && W, "clo_mlp_loss_grad: bad layer table");
  CLO_REQUIRE(N >= 1 && X && targets && G && ws, "clo_mlp_loss_grad: bad batch / targets / output / workspace");
  CLO_REQUIRE(loss_kind >= CLO_LOSS_MSE && loss_kind <= CLO_LOSS_BCE, "clo_mlp_loss_grad: loss kind %d is not MSE / CE / BCE", loss_kind);
  for (int l = 0; l < L; ++l) {
    CLO_REQUIRE(dims[l] > 0 && dims[l + 1] > 0 && acts[l] >= 0 && acts[l] <= 3, "clo_mlp_loss_grad: bad layer %d", l);
    CLO_REQUIRE(W[l], "clo_mlp_loss_grad: null weight pointer in layer %d", l);
  }
  hipStream_t st = (hipStream_t)stream;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  float *p = ws;
  float *a[65], *dphi[65];
  a[0] = const_cast<float *>(X);
  for (int l = 1; l <= L; ++l) {
    const long sz = (long)N * dims[l];
    a[l] = p; p += sz; p += sz; dphi[l] = p; p += sz;
  }
  p += 2L * N * dmax;
  float *gws = p;
  const long gws_sz = gemm_ws_floats(N, dmax);
  for (int l = 1; l <= L; ++l) {
    const int di = dims[l - 1], dout = dims[l];
    GemmArgs g = gemm_problem(N, dout, di, a[l - 1], di, 1, W[l - 1], 1, di, 0.f, a[l], dout);
    g.epi = EPI_ACT; g.e_act = acts[l - 1]; g.e_vec = b ? b[l - 1] : nullptr; g.e_out2 = dphi[l];
    int rc = launch_gemm_auto(g, gws, gws_sz, st);
    if (rc != CLO_OK) return rc;
  }
  hipLaunchKernelGGL(loss_grad_kernel, dim3(N), dim3(256), 0, st, loss_kind + CLO_LOSS_EF_MSE, a[L], targets, G, dims[L], scale);
  CLO_CHECK_LAUNCH("loss_grad_kernel");
  return CLO_OK;
}

// out = beta out + alpha J^T U for U [N][d_L]: plain forward pass (activations and their
// derivatives), then the backward chain on the GEMM engine.  Any widths / alignment.
extern "C" int clo_mlp_vjp(int L, const int *dims, const int *acts, const float *const *W,
                           const float *const *b, float *const *OW, float *const *Ob, const float *X,
                           int N, const float *U, float alpha, float beta, float *ws, void *stream) {
  CLO_REQUIRE(L >= 1 && L <= 64 && dims && acts && W && OW, "clo_mlp_vjp: bad layer table");
  CLO_REQUIRE(N >= 1 && X && U && ws, "clo_mlp_vjp: bad batch / cotangent / workspace");
  for (int l = 0; l < L; ++l) {
    CLO_REQUIRE(dims[l] > 0 && dims[l + 1] > 0 && acts[l] >= 0 && acts[l] <= 3, "clo_mlp_vjp: bad layer %d", l);
    CLO_REQUIRE(W[l] && OW[l], "clo_mlp_vjp: null weight pointer in layer %d", l);
  }
  hipStream_t st = (hipStream_t)stream;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  float *p = ws;
  float *a[65], *dphi[65];
  a[0] = const_cast<float *>(X);
  for (int l = 1; l <= L; ++l) {
    const long sz = (long)N * dims[l];
    a[l] = p; p += sz; p += sz; dphi[l] = p; p += sz;  // (the da slot of the shared layout stays unused)
  }
  float *dcur = p, *dnext = p + (long)N * dmax;
  p += 2L * N * dmax;
  float *gws = p;
  const long gws_sz = gemm_ws_floats(N, dmax);
  int rc;
  for (int l = 1; l <= L; ++l) {
    const int di = dims[l - 1], dout = dims[l];
    GemmArgs g = gemm_problem(N, dout, di, a[l - 1], di, 1, W[l - 1], 1, di, 0.f, a[l], dout);
    g.epi = EPI_ACT; g.e_act = acts[l - 1]; g.e_vec = b ? b[l - 1] : nullptr; g.e_out2 = dphi[l];
    rc = launch_gemm_auto(g, gws, gws_sz, st);
    if (rc != CLO_OK) return rc;
  }
  // delta_L = alpha * phi'_L * U
  {
    const long nc = (long)N * dims[L];
    hipLaunchKernelGGL(mask_scale_kernel, dim3(ew_grid(nc)), dim3(256), 0, st, dcur, U, dphi[L], nc, alpha);
    CLO_CHECK_LAUNCH("mask_scale_kernel");
  }
  for (int l = L; l >= 1; --l) {
    const int di = dims[l - 1], dout = dims[l];
    GemmArgs go = gemm_problem(dout, di, N, dcur, 1, dout, a[l - 1], di, 1, beta, OW[l - 1], di);
    rc = launch_gemm_auto(go, gws, gws_sz, st);
    if (rc != CLO_OK) return rc;
    if (Ob && Ob[l - 1]) {
      rc = launch_small_outer(Ob[l - 1], nullptr, dcur, N, dout, 1, beta, nullptr, 0, st);
      if (rc != CLO_OK) return rc;
    }
    if (l == 1) break;
    GemmArgs gd = gemm_problem(N, di, dout, dcur, dout, 1, W[l - 1], di, 1, 0.f, dnext, di);
    gd.epi = EPI_MUL; gd.e_mul = dphi[l - 1]; gd.ld_mul = di;
    rc = launch_gemm_auto(gd, gws, gws_sz, st);
    if (rc != CLO_OK) return rc;
    std::swap(dcur, dnext);
  }
  return CLO_OK;
}

#ifdef CLO_MID_TIMING
extern "C" void clo_mid_timing_set(unsigned long long *device_buffer) { clo::g_mid_stamps_host = device_buffer; }
#endif
