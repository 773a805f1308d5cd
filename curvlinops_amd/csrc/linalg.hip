// Dense factor post-processing: diagonal-block kernel of the blocked Cholesky inverse.
//
// (A + damping I)^-1 for KFAC's Kronecker factors (reference kronecker.py:328-373:
// cholesky + cholesky_inverse) is computed by a blocked right-looking algorithm whose O(n^3)
// work -- panel solves, trailing updates, the triangular inverse and L^-T L^-1 -- runs on the
// f32-MFMA GEMM of gemm.hip (driven from curvlinops_amd/_hip.py:cholesky_inverse).  The only
// non-GEMM piece is the factorisation of one nb x nb (nb <= 64) diagonal block and the
// inversion of its triangular factor, done here by a single workgroup in LDS.
#include "clo_common.h"

namespace clo {

constexpr int PNB = 64;

// Lower Cholesky factor of the nb x nb (nb <= 64) block at Ain (leading dimension ldin), written
// to A (may alias Ain); also writes Linv = L^-1 (lower triangular, zeros above the diagonal).
// *status is set to the 1-based pivot index if a non-positive pivot is met.
//
// A leaf sits on the critical path of the whole inverse (one per 64 rows of a factor), so its wall
// time is what matters.  ONE wavefront (barriers are free), the block in LDS, blocked by 16:
//   * 16 x 16 diagonal blocks: factor + triangular inverse in REGISTERS, lane i holds row i, fully
//     unrolled so every index is static; pivots and the other rows' column entries arrive by
//     v_readlane (an SGPR operand of the FMA): right-looking s_i[j] -= l_ik l_jk, then column-oriented
//     forward substitution for X = L^-1 (lane c owns column c).  ~1000 instructions: unrolling the
//     full 64 x 64 block the same way is 10x that, and a single wave running 80 KB of cold
//     straight-line code is instruction-fetch bound (measured 65 us); LDS loops with dynamic
//     indices instead wait ~650 ns per step on ds_read latency (83 us).
//   * everything else on v_mfma_f32_16x16x4_f32: the panel L_ik = S_ik X_kk^T, the trailing update
//     S_ij -= L_ik L_jk^T, and the off-diagonal blocks of the inverse
//     X_ij = -X_ii sum_{j <= k < i} L_ik X_kj.  All products are of the form P Q^T with both
//     operands row-major in LDS (one ds_read_b128 per lane feeds four MFMAs), so X is kept in
//     both orientations.
constexpr int PLD = PNB + 4;  // LDS row stride in floats: rows stay 16-byte aligned
constexpr int PSB = 16;       // sub-block
constexpr int TLD = PSB + 4;

#ifdef CLO_POTRF_DBG
__device__ long long potrf_dbg[16];
#define CLO_TICK(i) if (lane == 0) potrf_dbg[i] = wall_clock64();
#else
#define CLO_TICK(i)
#endif

__device__ __forceinline__ float read_lane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

using lf32x4 = __attribute__((ext_vector_type(4))) float;

// acc += P Q^T for 16 x 16 blocks given as per-lane fragments (row lane & 15, k = 4 (lane >> 4) ..+3)
__device__ __forceinline__ lf32x4 mm16(const float4 a, const float4 b, lf32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}

// One chain of 16 steps for the Cholesky factor L of a 16 x 16 block AND X = L^-1 (round 6).  Lane (idx = lane & 15) holds row idx of
// the block in s[] and column idx of X in x[]; every 16-lane row of the wave runs the same block.  Step k of the forward substitution
// needs exactly column k of L and 1 / L_kk, which step k of the factorisation has just formed, and the SAME broadcast L[j][k] feeds both
// updates; the broadcast is a DPP operand (row_newbcast: lane j of each row to all lanes of the row) instead of v_readlane through a
// scalar register -- 2.5 -> 1.x us per block (tools/r6/probe_potrf_timeline.py).  Same operations per element, in the same order.
template <int J>
__device__ __forceinline__ float row_lane(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + J, 0xF, 0xF, true));
}
template <int K, int J>
struct Chol16Inner {
  static __device__ __forceinline__ void run(float (&s)[16], float (&x)[16], float l, float xi) {
    const float lj = row_lane<J>(l);
    s[J] = fmaf(-l, lj, s[J]);
    x[J] = fmaf(-lj, xi, x[J]);
    Chol16Inner<K, J + 1>::run(s, x, l, xi);
  }
};
template <int K>
struct Chol16Inner<K, 16> {
  static __device__ __forceinline__ void run(float (&)[16], float (&)[16], float, float) {}
};
template <int K>
struct Chol16Step {
  // bad: first non-positive pivot (1-based), kept as data: no exits inside the chain
  static __device__ __forceinline__ void run(float (&s)[16], float (&x)[16], int idx, int &bad) {
    const float d = row_lane<K>(s[K]);
    // false for NaN too (padding rows have d == 1).  A bad pivot poisons the rest of the block with NaNs, which nobody reads
    bad = (bad == 0 && !(d > 0.f)) ? K + 1 : bad;
    const float inv = __builtin_amdgcn_rsqf(d);
    const float l = (idx == K) ? d * inv : s[K] * inv;
    s[K] = l;
    const float xi = x[K] * inv;
    x[K] = xi;  // X[K][idx]
    Chol16Inner<K, K + 1>::run(s, x, l, xi);
    Chol16Step<K + 1>::run(s, x, idx, bad);
  }
};
template <>
struct Chol16Step<16> {
  static __device__ __forceinline__ void run(float (&)[16], float (&)[16], int, int &) {}
};
static_assert(PSB == 16, "the chain is written for 16 x 16 blocks");

__global__ __launch_bounds__(64) void potrf_diag_kernel(const float *Ain, long ldin, float *A,
                                                        long lda, int nb, float *__restrict__ Linv,
                                                        long ldinv, int *__restrict__ status,
                                                        int pivot_base, long batch_stride) {
  // grid.x = batch: matrix b of a batch of equally sized factors lives batch_stride floats further
  Ain += blockIdx.x * batch_stride;
  A += blockIdx.x * batch_stride;
  Linv += blockIdx.x * batch_stride;
  status += blockIdx.x;
  __shared__ __attribute__((aligned(16))) float S[PNB * PLD];   // becomes L (lower blocks)
  __shared__ __attribute__((aligned(16))) float X[PNB * PLD];   // L^-1
  __shared__ __attribute__((aligned(16))) float XT[PNB * PLD];  // (L^-1)^T
  __shared__ __attribute__((aligned(16))) float TT[PSB * TLD];  // transposed 16 x 16 scratch
  const int lane = threadIdx.x;
  const int idx = lane & 15, s4 = (lane >> 4) * 4;
  // coalesced load (lane = column); rows / columns beyond nb: identity
  {
    float v[PNB];  // all 64 loads in flight: one memory round trip
    const int cl = min(lane, nb - 1);
#pragma unroll
    for (int r = 0; r < PNB; ++r) v[r] = Ain[(long)min(r, nb - 1) * ldin + cl];
#pragma unroll
    for (int r = 0; r < PNB; ++r) {
      S[r * PLD + lane] = (r < nb && lane < nb) ? v[r] : ((r == lane) ? 1.f : 0.f);
      X[r * PLD + lane] = 0.f;
      XT[r * PLD + lane] = 0.f;
    }
  }
  __syncthreads();
  CLO_TICK(1)

  auto frag = [&](const float *M, int r0, int c0) {
    return *reinterpret_cast<const float4 *>(M + (r0 + idx) * PLD + c0 + s4);
  };
  // D layout of the MFMA: acc[r] is element (s4 + r, idx) of the 16 x 16 result
  auto store = [&](float *M, int r0, int c0, const lf32x4 d, float scale) {
#pragma unroll
    for (int r = 0; r < 4; ++r) M[(r0 + s4 + r) * PLD + c0 + idx] = scale * d[r];
  };
  auto store_t = [&](float *M, int ld, int r0, int c0, const lf32x4 d, float scale) {
#pragma unroll
    for (int r = 0; r < 4; ++r) M[(r0 + idx) * ld + c0 + s4 + r] = scale * d[r];
  };

#pragma unroll 1
  for (int kb = 0; kb < PNB / PSB; ++kb) {
    const int o = kb * PSB;
    {  // ---- diagonal block in registers: every group of 16 lanes runs the same rows
      float s[PSB], x[PSB];
      const float *src = S + (o + idx) * PLD + o;
#pragma unroll
      for (int q = 0; q < PSB / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(src + 4 * q);
        s[4 * q] = v.x; s[4 * q + 1] = v.y; s[4 * q + 2] = v.z; s[4 * q + 3] = v.w;
      }
      int bad = 0;  // first non-positive pivot (1-based), kept as data: no exits inside the chain
#pragma unroll
      for (int r = 0; r < PSB; ++r) x[r] = (r == idx) ? 1.f : 0.f;
      Chol16Step<0>::run(s, x, idx, bad);   // (one chain of 16 steps for L and X = L^-1, above)
      bad = __builtin_amdgcn_readfirstlane(bad);
      if (bad) {  // uniform
        if (lane == 0) *status = pivot_base + o + bad;
        return;
      }
      if (lane < PSB) {
        float *dst = S + (o + idx) * PLD + o;
        float *xt = XT + (o + idx) * PLD + o;
#pragma unroll
        for (int j = 0; j < PSB; ++j) {
          dst[j] = (j <= idx) ? s[j] : 0.f;
          X[(o + j) * PLD + o + idx] = x[j];
          xt[j] = x[j];
        }
      }
    }
    __syncthreads();
    if (kb == 0) { CLO_TICK(2) }
    // ---- panel: L_ik = S_ik X_kk^T
    {
      const float4 b = frag(X, o, o);
      for (int ib = kb + 1; ib < PNB / PSB; ++ib) {
        const float4 a = frag(S, ib * PSB, o);
        const lf32x4 d = mm16(a, b, lf32x4{0.f, 0.f, 0.f, 0.f});
        store(S, ib * PSB, o, d, 1.f);
      }
    }
    __syncthreads();
    if (kb == 0) { CLO_TICK(3) }
    // ---- trailing update: S_ij -= L_ik L_jk^T (lower blocks)
    for (int ib = kb + 1; ib < PNB / PSB; ++ib) {
      const float4 a = frag(S, ib * PSB, o);
      for (int jb = kb + 1; jb <= ib; ++jb) {
        const float4 b = frag(S, jb * PSB, o);
        const lf32x4 d = mm16(a, b, lf32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(ib * PSB + s4 + r) * PLD + jb * PSB + idx] -= d[r];
      }
    }
    __syncthreads();
    if (kb == 0) { CLO_TICK(4) }
  }
  CLO_TICK(5)

  // ---- off-diagonal blocks of X = L^-1, column block by column block
#pragma unroll 1
  for (int jb = 0; jb < PNB / PSB - 1; ++jb)
#pragma unroll 1
    for (int ib = jb + 1; ib < PNB / PSB; ++ib) {
      lf32x4 t{0.f, 0.f, 0.f, 0.f};
      for (int k = jb; k < ib; ++k)  // (L_ik X_kj)[i][n] = sum_kk L_ik[i][kk] XT_jk[n][kk]
        t = mm16(frag(S, ib * PSB, k * PSB), frag(XT, jb * PSB, k * PSB), t);
      store_t(TT, TLD, 0, 0, t, 1.f);
      __syncthreads();
      const float4 tb = *reinterpret_cast<const float4 *>(TT + idx * TLD + s4);
      const lf32x4 d = mm16(frag(X, ib * PSB, ib * PSB), tb, lf32x4{0.f, 0.f, 0.f, 0.f});
      store(X, ib * PSB, jb * PSB, d, -1.f);
      store_t(XT, PLD, jb * PSB, ib * PSB, d, -1.f);
      __syncthreads();
    }

  CLO_TICK(6)
  // coalesced stores (lane = column): L lower triangle, L^-1 with its zeros
  if (lane < nb) {
    const float *sp = S + lane, *xp = X + lane;
    float *ap = A + lane, *lp = Linv + lane;
#pragma unroll 8
    for (int r = 0; r < nb; ++r) {
      const float lv = sp[r * PLD], xv = xp[r * PLD];
      lp[(long)r * ldinv] = xv;
      if (lane <= r) ap[(long)r * lda] = lv;
    }
  }
  CLO_TICK(7)
}

// ---- 65 ... 128 rows: the same factorisation + triangular inverse by FOUR waves of one workgroup -------------------
// Replaces, for every 128-row node of the recursion, two leaf launches and the four tiny products between them (each a
// latency chain of its own: 2 x 23 + 4 x 8 us measured at n = 4608) by one launch.  The chain of 16 x 16 diagonal blocks
// stays sequential on wave 0 (registers, as above); the panel, the trailing update and the columns of the inverse are
// spread over the waves.  LDS: S (becomes L) and X = L^-1, 128 x 132 floats each; X^T is not kept -- the one product
// that wants it reads X by columns (four ds_read_b32 instead of one b128).
#ifndef CLO_POTRF_QW
#define CLO_POTRF_QW 8
#endif
// (QW = 8 since round 6: the panel, trailing-update and L^-1 phases are chains of LDS round trips per wave -- twice the waves, half the
// chain per wave; the diagonal block stays with wave 0)
constexpr int QNB = 128, QLD = QNB + 4, QW = CLO_POTRF_QW, QBLK = QNB / PSB;
constexpr int QSMEM = (2 * QNB * QLD + QW * PSB * TLD) * (int)sizeof(float);

__global__ __launch_bounds__(QW * 64) void potrf_node128_kernel(const float *Ain, long ldin, float *A, long lda, int nb,
                                                                float *__restrict__ Linv, long ldinv,
                                                                int *__restrict__ status, int pivot_base,
                                                                long batch_stride
#ifdef CLO_POTRF_TIMING
                                                                , unsigned long long *stamps
#endif
                                                                ) {
#ifdef CLO_POTRF_TIMING
#define PQ_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && stamps) stamps[(i)] = wall_clock64(); } while (0)
#else
#define PQ_STAMP(i) do { } while (0)
#endif
  PQ_STAMP(0);
  Ain += blockIdx.x * batch_stride;
  A += blockIdx.x * batch_stride;
  Linv += blockIdx.x * batch_stride;
  status += blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) float qsm[];
  float *S = qsm, *X = qsm + QNB * QLD, *TTall = X + QNB * QLD;
  __shared__ int bad_sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, s4 = (lane >> 4) * 4;
  float *TT = TTall + wave * (PSB * TLD);
  if (tid == 0) bad_sh = 0;
  // (16-byte path: every operand a multiple of four floats wide and 16-byte aligned -- the working matrices of the blocked recursion
  // always are; nb is a multiple of 4 then)
  const bool vec = ((ldin | lda | ldinv | (long)nb) & 3) == 0 &&
                   (((unsigned long)Ain | (unsigned long)A | (unsigned long)Linv) & 15ul) == 0;
  if (vec) {   // load: thread -> (row, four columns); rows / columns beyond nb: identity
    float4 v[QNB * QNB / 4 / (QW * 64)];
#pragma unroll
    for (int it = 0; it < QNB * QNB / 4 / (QW * 64); ++it) {
      const int e = it * (QW * 64) + tid, row = e / (QNB / 4), c4 = (e % (QNB / 4)) * 4;
      v[it] = *reinterpret_cast<const float4 *>(Ain + (long)min(row, nb - 1) * ldin + min(c4, nb - 4));
    }
#pragma unroll
    for (int it = 0; it < QNB * QNB / 4 / (QW * 64); ++it) {
      const int e = it * (QW * 64) + tid, row = e / (QNB / 4), c4 = (e % (QNB / 4)) * 4;
      float4 w = v[it];
      if (row >= nb || c4 >= nb)
        w = make_float4(row == c4 ? 1.f : 0.f, row == c4 + 1 ? 1.f : 0.f, row == c4 + 2 ? 1.f : 0.f, row == c4 + 3 ? 1.f : 0.f);
      *reinterpret_cast<float4 *>(S + row * QLD + c4) = w;
      *reinterpret_cast<float4 *>(X + row * QLD + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else
  // load: thread t owns column t & 127 of half the rows; rows / columns beyond nb: identity
  {
    constexpr int RPT = QNB / (QW * 64 / QNB);   // rows per thread: the block's threads form QW * 64 / QNB groups of QNB columns
    const int col = tid & (QNB - 1), r0 = (tid >> 7) * RPT;
    const int cl = min(col, nb - 1);
    float v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) v[r] = Ain[(long)min(r0 + r, nb - 1) * ldin + cl];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int row = r0 + r;
      S[row * QLD + col] = (row < nb && col < nb) ? v[r] : ((row == col) ? 1.f : 0.f);
      X[row * QLD + col] = 0.f;
    }
  }
  __syncthreads();
  PQ_STAMP(1);

  auto frag = [&](const float *M, int r0, int c0) {
    return *reinterpret_cast<const float4 *>(M + (r0 + idx) * QLD + c0 + s4);
  };
  // the fragment of M^T: (M^T)[c0 + idx][r0 + s4 ..+3] = M[r0 + s4 + q][c0 + idx]
  auto frag_t = [&](const float *M, int ld, int r0, int c0) {
    const float *q = M + (r0 + s4) * ld + c0 + idx;
    return make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]);
  };
  auto store = [&](float *M, int ld, int r0, int c0, const lf32x4 d, float scale) {
#pragma unroll
    for (int r = 0; r < 4; ++r) M[(r0 + s4 + r) * ld + c0 + idx] = scale * d[r];
  };

  // block (ib, jb), jb < ib, of X = L^-1:  X_ij = -X_ii sum_{jb <= k < ib} L_ik X_kj  (needs row ib of L, X_ii and the rows < ib of X)
  auto x_block = [&](int ib, int jb) {
    lf32x4 t{0.f, 0.f, 0.f, 0.f};
    for (int k = jb; k < ib; ++k)  // (L_ik X_kj)[i][n] = sum_kk L_ik[i][kk] X_kj[kk][n]
      t = mm16(frag(S, ib * PSB, k * PSB), frag_t(X, QLD, k * PSB, jb * PSB), t);
    store(TT, TLD, 0, 0, t, 1.f);   // t as it is: TT[i][n]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes before its reads (other lanes' data)
    // d = X_ii t: P = X_ii rows, Q rows = t^T rows = columns of TT
    const lf32x4 d = mm16(frag(X, ib * PSB, ib * PSB), frag_t(TT, TLD, 0, 0), lf32x4{0.f, 0.f, 0.f, 0.f});
    store(X, QLD, ib * PSB, jb * PSB, d, -1.f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

#pragma unroll 1
  for (int kb = 0; kb < QBLK; ++kb) {
    const int o = kb * PSB;
    // Round 6: while wave 0 factors diagonal block kb, the other waves -- idle in this phase before -- form row kb - 1 of the
    // off-diagonal blocks of X = L^-1 (its L row, X_ii and the earlier rows of X are final since the previous step): the chains that
    // ran after the loop (5.1 us of 27, tools/r6/probe_potrf_timeline.py) are hidden behind the diagonal blocks, only the last row is left.
    if (wave != 0 && kb >= 2) {
      for (int jb = wave - 1; jb < kb - 1; jb += QW - 1) x_block(kb - 1, jb);
    }
    if (wave == 0) {  // ---- diagonal block in registers (every group of 16 lanes runs the same rows)
      float s[PSB], x[PSB];
      const float *src = S + (o + idx) * QLD + o;
#pragma unroll
      for (int q = 0; q < PSB / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(src + 4 * q);
        s[4 * q] = v.x; s[4 * q + 1] = v.y; s[4 * q + 2] = v.z; s[4 * q + 3] = v.w;
      }
      int bad = 0;
#pragma unroll
      for (int r = 0; r < PSB; ++r) x[r] = (r == idx) ? 1.f : 0.f;
      Chol16Step<0>::run(s, x, idx, bad);   // (one chain of 16 steps for L and X = L^-1, above)
      if (bad && lane == 0) {
        *status = pivot_base + o + bad;
        bad_sh = 1;
      }
      if (lane < PSB) {
        float *dst = S + (o + idx) * QLD + o;
#pragma unroll
        for (int j = 0; j < PSB; ++j) {
          dst[j] = (j <= idx) ? s[j] : 0.f;
          X[(o + j) * QLD + o + idx] = x[j];
        }
      }
    }
    __syncthreads();
    PQ_STAMP(2 + 3 * kb);
    if (bad_sh) return;  // uniform: a non-positive pivot ends the factorisation (the caller reads *status)
    // ---- panel: L_ik = S_ik X_kk^T, block rows spread over the waves
    {
      const float4 b = frag(X, o, o);
      for (int ib = kb + 1 + wave; ib < QBLK; ib += QW) {
        const float4 a = frag(S, ib * PSB, o);
        const lf32x4 d = mm16(a, b, lf32x4{0.f, 0.f, 0.f, 0.f});
        store(S, QLD, ib * PSB, o, d, 1.f);
      }
    }
    __syncthreads();
    PQ_STAMP(3 + 3 * kb);
    // ---- trailing update: S_ij -= L_ik L_jk^T (lower blocks), pairs spread over the waves.  A wave takes its pairs FOUR at a time: all
    // operand reads first, then four independent MFMA chains, then the writes (one pair at a time was one LDS round trip + one dependent
    // chain of four MFMAs each, 0.34 us per pair: tools/r6/probe_potrf_timeline.py).  The block itself is the accumulator operand.
    {
      const int m = QBLK - kb - 1, npairs = m * (m + 1) / 2;
      for (int p0 = wave; p0 < npairs; p0 += 4 * QW) {
        float4 a[4], b[4];
        lf32x4 c[4];
        int ro[4], co[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pr = p0 + u * QW;
          ok[u] = pr < npairs;   // (wave-uniform)
          if (!ok[u]) continue;
          const int pp = pr;
          int i = 0;
          while ((i + 1) * (i + 2) / 2 <= pp) ++i;
          const int j = pp - i * (i + 1) / 2;
          ro[u] = (kb + 1 + i) * PSB; co[u] = (kb + 1 + j) * PSB;
          a[u] = frag(S, ro[u], o);
          a[u].x = -a[u].x; a[u].y = -a[u].y; a[u].z = -a[u].z; a[u].w = -a[u].w;
          b[u] = frag(S, co[u], o);
#pragma unroll
          for (int r = 0; r < 4; ++r) c[u][r] = S[(ro[u] + s4 + r) * QLD + co[u] + idx];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ok[u]) c[u] = mm16(a[u], b[u], c[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ok[u]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(ro[u] + s4 + r) * QLD + co[u] + idx] = c[u][r];
          }
      }
    }
    __syncthreads();
    PQ_STAMP(4 + 3 * kb);
  }

  // ---- the last row of the off-diagonal blocks of X = L^-1 (the other rows were formed under the diagonal blocks above)
  for (int jb = wave; jb < QBLK - 1; jb += QW) x_block(QBLK - 1, jb);
  __syncthreads();
  PQ_STAMP(26);

  if (vec) {   // 16-byte stores, a full 512-byte row per 32 lanes: L^-1 with its zeros, L up to the diagonal
#pragma unroll 4
    for (int it = 0; it < QNB * QNB / 4 / (QW * 64); ++it) {
      const int e = it * (QW * 64) + tid, row = e / (QNB / 4), c4 = (e % (QNB / 4)) * 4;
      if (row < nb && c4 < nb) {
        *reinterpret_cast<float4 *>(Linv + (long)row * ldinv + c4) = *reinterpret_cast<const float4 *>(X + row * QLD + c4);
        if (c4 <= row) {
          const float4 lv = *reinterpret_cast<const float4 *>(S + row * QLD + c4);
          float *dst = A + (long)row * lda + c4;
          if (c4 + 3 <= row) *reinterpret_cast<float4 *>(dst) = lv;
          else {
            dst[0] = lv.x;
            if (c4 + 1 <= row) dst[1] = lv.y;
            if (c4 + 2 <= row) dst[2] = lv.z;
          }
        }
      }
    }
  } else
  // coalesced stores (thread = column of half the rows): L lower triangle, L^-1 with its zeros
  {
    constexpr int RPT = QNB / (QW * 64 / QNB);
    const int col = tid & (QNB - 1), r0 = (tid >> 7) * RPT;
    if (col < nb) {
      for (int r = r0; r < min(r0 + RPT, nb); ++r) {
        const float lv = S[r * QLD + col], xv = X[r * QLD + col];
        Linv[(long)r * ldinv + col] = xv;
        if (col <= r) A[(long)r * lda + col] = lv;
      }
    }
  }
#ifdef CLO_POTRF_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  PQ_STAMP(27);
}

#ifdef CLO_POTRF_TIMING
static unsigned long long *g_potrf_stamps = nullptr;
#endif
static int potrf_node128_launch(const float *Ain, long ldin, float *A, long lda, int nb, float *Linv, long ldinv,
                                int *status, int pivot_base, long batch_stride, int batch, hipStream_t st) {
  static bool attr_done[64] = {};  // per device
  int dev = 0;
  int rcd = check_hip(hipGetDevice(&dev), "hipGetDevice");
  if (rcd != CLO_OK) return rcd;
  bool &attr_set = attr_done[dev & 63];
  if (!attr_set) {
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_node128_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, QSMEM),
                       "hipFuncSetAttribute(potrf_node128_kernel)");
    if (rc != CLO_OK) return rc;
    attr_set = true;
  }
#ifdef CLO_POTRF_TIMING
  hipLaunchKernelGGL(potrf_node128_kernel, dim3(batch), dim3(QW * 64), QSMEM, st, Ain, ldin, A, lda, nb, Linv, ldinv,
                     status, pivot_base, batch_stride, g_potrf_stamps);
#else
  hipLaunchKernelGGL(potrf_node128_kernel, dim3(batch), dim3(QW * 64), QSMEM, st, Ain, ldin, A, lda, nb, Linv, ldinv,
                     status, pivot_base, batch_stride);
#endif
  CLO_CHECK_LAUNCH("potrf_node128_kernel");
  return CLO_OK;
}

// S = (A + damping * I) (+) I ; L = 0 ; Li = 0.  The working matrices are np x np with
// np = n rounded up to a multiple of 4 (identity on the padding): joint weight + bias factors have
// odd sizes (577, 1153, 2305, 4609), and only float4-complete rows run on the aligned GEMM engine
// (n = 4609 took 14.1 ms against 7.5 ms for n = 4608 before the padding).
struct InitBatch {
  static constexpr int MAX = 32;
  const float *A[MAX];
  long lda[MAX];
  float damping[MAX];
};
__global__ void chol_init_kernel(const InitBatch ib, float *__restrict__ S, float *__restrict__ L,
                                 float *__restrict__ Li, int n, int np, long stride) {
  const int b = blockIdx.y;
  const float *__restrict__ A = ib.A[b];
  const long lda = ib.lda[b];
  const float damping = ib.damping[b];
  S += b * stride; L += b * stride; Li += b * stride;
  const long total = (long)np * np;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int i = e / np, j = e % np;
    float v = (i == j) ? 1.f : 0.f;
    if (i < n && j < n) v = A[(long)i * lda + j] + (i == j ? damping : 0.f);
    S[e] = v;
    L[e] = 0.f;
    Li[e] = 0.f;
  }
}

// out_b[i][j] = src_b[i][j], i, j < n (padded product -> the caller's tensors)
struct OutBatch {
  static constexpr int MAX = 32;
  float *out[MAX];
  long ldo[MAX];
};
__global__ void chol_copy_out_kernel(const OutBatch ob, const float *__restrict__ src, int n, int np,
                                     long stride) {
  const int b = blockIdx.y;
  float *__restrict__ out = ob.out[b];
  const long ldo = ob.ldo[b];
  src += b * stride;
  const long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int i = e / n, j = e % n;
    out[(long)i * ldo + j] = src[(long)i * np + j];
  }
}

}  // namespace clo
#include <cstdlib>
#include <mutex>
#include <vector>

#include "gemm.h"
namespace clo {

struct CholCtx {
  float *S, *L, *Li, *T, *G;
  long gws;
  int n;        // padded size = leading dimension of S, L, Li
  int batch;    // equally sized factors processed by every launch
  long stride;  // floats between consecutive matrices in S, L, Li (and T)
  int *status;
  hipStream_t st;
};

// Recursive blocked Cholesky carrying the inverse of the triangular factor:
//   L11, L11^-1 = rec(A11);  L21 = A21 L11^-T;  S22 -= L21 L21^T;  L22, L22^-1 = rec(S22);
//   (L^-1)21 = -L22^-1 (L21 L11^-1)
// Every launch covers the whole batch (leaf: one workgroup per matrix, products: batched GEMMs).
static int chol_rec(const CholCtx &c, int o, int m) {
  const long n = c.n;
  if (m <= PNB) {
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(c.batch), dim3(64), 0, c.st, c.S + o * n + o, n,
                       c.L + o * n + o, n, m, c.Li + o * n + o, n, c.status, o, c.stride);
    CLO_CHECK_LAUNCH("potrf_diag_kernel");
    return CLO_OK;
  }
#ifndef CLO_CHOL_NODE128
#define CLO_CHOL_NODE128 1
#endif
  static const int node128 = CLO_CHOL_NODE128;
  if (m <= QNB && node128) {
    int rcq = potrf_node128_launch(c.S + o * n + o, n, c.L + o * n + o, n, m, c.Li + o * n + o, n, c.status, o,
                                   c.stride, c.batch, c.st);
    return rcq;
  }
  const int m1 = ((m / 2 + PNB - 1) / PNB) * PNB, m2 = m - m1;
  const int a = o, b = o + m1;
  int rc = chol_rec(c, a, m1);
  if (rc != CLO_OK) return rc;
  const float *L11i = c.Li + a * n + a;
  float *L21 = c.L + b * n + a;
  // one (batched) product on the GEMM engine; `tri` names the triangular operand so that only the
  // k range of each tile that can be nonzero is visited, `sym` computes the block-upper triangle
  // and mirrors
  auto gemm = [&](int M, int N, int K, float alpha, const float *A, long sa_m, long sa_k, long sa_b,
                  const float *B, long sb_k, long sb_n, long sb_b, float beta, float *C, long ldc,
                  long sc_b, int tri, int sym) {
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
    g.A = A; g.sa_m = sa_m; g.sa_k = sa_k; g.sa_b = sa_b;
    g.B = B; g.sb_k = sb_k; g.sb_n = sb_n; g.sb_b = sb_b;
    g.C = C; g.ldc = ldc; g.sc_b = sc_b; g.tri = tri; g.sym = sym;
    return launch_gemm_auto(g, c.G, c.gws, c.st, c.batch);
  };
  const long sb = c.stride;
  // L21 = S21 * L11i^T      (B(k,n) = L11i[n][k], zero for k > n)
  rc = gemm(m2, m1, m1, 1.f, c.S + b * n + a, n, 1, sb, L11i, 1, n, sb, 0.f, L21, n, sb, TRI_KLT_N, 0);
  if (rc != CLO_OK) return rc;
  // S22 -= L21 * L21^T      (symmetric: half the tiles, mirrored)
  rc = gemm(m2, m2, m1, -1.f, L21, n, 1, sb, L21, 1, n, sb, 1.f, c.S + b * n + b, n, sb, 0, 1);
  if (rc != CLO_OK) return rc;
  rc = chol_rec(c, b, m2);
  if (rc != CLO_OK) return rc;
  // T = L21 * L11i (B(k,n) = L11i[k][n], zero for k < n) ; Li21 = -L22i * T (A(m,k) zero for k > m)
  rc = gemm(m2, m1, m1, 1.f, L21, n, 1, sb, L11i, n, 1, sb, 0.f, c.T, m1, sb, TRI_KGE_N, 0);
  if (rc != CLO_OK) return rc;
  return gemm(m2, m1, m2, -1.f, c.Li + b * n + b, n, 1, sb, c.T, m1, 1, sb, 0.f, c.Li + b * n + a, n, sb,
              TRI_KLT_M, 0);
}

// ---- pipelined variant (round 3): the same arithmetic as a DAG over three streams -----------------------------------
// chol_rec above is one in-order chain: every product -- also the big trailing updates and the whole triangular inverse --
// sits between two diagonal nodes although the next node only needs its own 128 x 128 block.  Here the factorisation is
// right-looking over 128-row block columns with a look-ahead of one:
//   main:  node k (L_kk, L_kk^-1) -> panel L[k+1:, k] = S[k+1:, k] L_kk^-T -> update of block column k+1 only
//   side:  the rest of the trailing update S[k+2:, k+2:] -= L[k+2:, k] L[k+2:, k]^T (symmetric product), one step behind
//   inv:   the triangular inverse by the recursion of chol_rec ((L^-1)21 = -L22^-1 (L21 L11^-1), post-order), every leaf
//          waiting for "node k done": it runs under the factorisation of the blocks to the right
// so the critical path is nodes + two skinny products per block column (n = 4608: 36 x ~70 us) instead of everything.
// Events are recorded before they are waited for in host order (no wait can precede its record in a hardware queue).
struct CholAsync {
  hipStream_t main = nullptr, side = nullptr, inv = nullptr, inv2 = nullptr;
  std::vector<hipEvent_t> ev;
  hipEvent_t fork = nullptr;   // caller's stream -> main
  hipEvent_t done = nullptr;   // recorded on `main` at the end of the call that used the set last (the caller's stream waits for it)
  int err = CLO_OK;
  hipEvent_t event(size_t i) {
    while (ev.size() <= i) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { err = CLO_EHIP; return nullptr; }
      ev.push_back(e);
    }
    return ev[i];
  }
};
// Process-wide pool per device: a call takes a set for its duration and hands it back (the worker threads of
// linalg_native.concurrent_inverses come and go -- thread-local sets would leak streams).  A set is reused while work of
// its previous call may still be queued: the helper streams are in order, and every wait below follows its record.
static std::mutex g_chol_pool_mu;
static std::vector<CholAsync *> g_chol_pool[64];
// (Round 5 found that the helper's hardware queue shares one of the command processor's four dispatch pipes with the
// caller's queue whenever their creation ranks differ by a multiple of four -- tools/ubench_queue_pipes.py -- and picked the
// helper among candidates with a HOST-TIMED probe behind a hipDeviceSynchronize.  Round 6 removed that: a library entry
// point must not synchronise the device (it stalls the caller's other streams and is illegal while any thread captures a
// graph), and a wall-clock lottery must not decide a stream layout.  The rule is fixed now: a set is TWO streams of its own,
// created back to back (consecutive hardware-queue ranks = different dispatch pipes, whatever the caller's stream is), the
// critical chain on the first and the bulk products on the second; the caller's stream only forks into the set and joins it.)
static CholAsync *chol_async_acquire(int *dev_out, hipStream_t caller) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  *dev_out = dev & 63;
  RelaxedCaptureScope relaxed;   // event queries / stream creation below: never invalidate another thread's capture
  {
    // only a set whose previous call has COMPLETED on the device is taken again: re-recording an event that a queued wait
    // still refers to, or queueing behind the previous call's helper work, couples successive calls (ResNet-18's 42 factors
    // back to back: 18.4 instead of 11.7 ms)
    std::lock_guard<std::mutex> lk(g_chol_pool_mu);
    auto &pool = g_chol_pool[dev & 63];
    for (size_t i = 0; i < pool.size(); ++i) {
      if (!pool[i]->done || hipEventQuery(pool[i]->done) == hipSuccess) {
        CholAsync *a = pool[i];
        pool.erase(pool.begin() + i);
        return a;
      }
    }
    (void)hipGetLastError();   // hipEventQuery's "not ready" is not an error
  }
  CholAsync *a = new CholAsync();
  // lowest priority: whatever the caller's stream has ready (the nodes and skinny products of the critical path) is
  // dispatched ahead of the bulk products queued here
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
#ifdef CLO_CHOL_HELPER_PRIO   // (experiments: a fixed priority for the helper streams)
  least = CLO_CHOL_HELPER_PRIO;
#endif
  // ONE helper stream for the trailing updates and the inverse (three were measured first: HIP streams share a handful of
  // hardware queues, and when two helpers land on the same queue the trailing update the main stream is about to need waits
  // behind an inverse product that waits for a later node -- 4 x 2305: 2.5 ms alone, 5.2 ms after another call had shifted
  // the stream-to-queue assignment).  In one in-order stream every operation is queued when the node / panel it needs has
  // been queued, so nothing ever waits behind a later dependency.
#ifndef CLO_CHOL_HELPERS
#define CLO_CHOL_HELPERS 1
#endif
  static const int nhelp = CLO_CHOL_HELPERS;
  // The critical chain runs on a stream of THIS set as well (`main`, created right before its helper), not on the caller's
  // stream: two hardware queues that sit on the same dispatch pipe of the command processor halve each other's workgroup
  // dispatch rate -- every kernel of the pipeline then runs ~1.4 x longer (rocprofv3: 19.3 instead of 13.8 ms of kernel time
  // for ResNet-18's 42 factors, 16.8 instead of 11.5 ms wall) -- and which pipe the CALLER's queue sits on relative to a helper
  // created much later is a lottery (it flipped when a captured hipGraph with a second branch was alive in the process).
  // Two queues created back to back get consecutive ids, i.e. different pipes.
#ifndef CLO_CHOL_OWN_MAIN
#define CLO_CHOL_OWN_MAIN 1
#endif
  static const int own_main = [] {
    const char *e = getenv("CLO_CHOL_OWN_MAIN");   // (A/B runs of tools/probe_kfac_inverse.py; not a product knob)
    return e ? atoi(e) : CLO_CHOL_OWN_MAIN;
  }();
  (void)caller;
  bool ok = !own_main || (hipStreamCreateWithPriority(&a->main, hipStreamNonBlocking, least) == hipSuccess &&
                          hipEventCreateWithFlags(&a->fork, hipEventDisableTiming) == hipSuccess);
  ok = ok && hipStreamCreateWithPriority(&a->side, hipStreamNonBlocking, least) == hipSuccess;
  if (ok) {
    // first use creates the hardware queues: do it now, in this order (no synchronisation -- the kernels are empty)
    if (a->main) (void)launch_occupy(1, 0, 1, a->main);
    (void)launch_occupy(1, 0, 1, a->side);
  }
  if (ok && nhelp >= 3) {
    ok = hipStreamCreateWithPriority(&a->inv, hipStreamNonBlocking, least) == hipSuccess &&
         hipStreamCreateWithPriority(&a->inv2, hipStreamNonBlocking, least) == hipSuccess;
  } else {
    a->inv = a->inv2 = a->side;
  }
  if (!ok) {
    delete a;
    return nullptr;
  }
  return a;
}
static void chol_async_release(CholAsync *a, int dev) {
  std::lock_guard<std::mutex> lk(g_chol_pool_mu);
  g_chol_pool[dev].push_back(a);
}

struct CholPipe {
  const CholCtx *c;
  CholAsync *as;
  std::vector<int> off, bs;   // block columns
  float *G_side, *G_inv, *G_inv2;
  int nblk;
};

static int chol_gemm(const CholCtx &c, hipStream_t st, float *G, int M, int N, int K, float alpha, const float *A,
                     long sa_m, long sa_k, const float *B, long sb_k, long sb_n, float beta, float *C, long ldc,
                     int tri, int sym) {
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
  g.A = A; g.sa_m = sa_m; g.sa_k = sa_k; g.sa_b = c.stride;
  g.B = B; g.sb_k = sb_k; g.sb_n = sb_n; g.sb_b = c.stride;
  g.C = C; g.ldc = ldc; g.sc_b = c.stride; g.tri = tri; g.sym = sym;
  // (no stream-K here: its workgroups wait for each other inside the kernel, and the products of the pipeline run side by
  // side on several streams -- 4 x 2305: 5.2 ms with it, 2.5 without)
  return launch_gemm_auto(g, G, std::min(c.gws, gemm_streamk_ws_floats_square() - 1), st, c.batch);
}

// triangular inverse of the block range [b0, b1) on the `inv` stream (diagonal blocks come from the nodes).  The first
// product of a node, T = L21 L11^-1, only needs the LEFT half: it is queued before the right half's recursion, so that
// after the last diagonal node only one product per level of the right spine is left.  T buffers form a stack in c.T
// (a node's T stays live across its right child: sum over the spine < n^2 / 3).
// The recursion is first written down as a list of operations in post-order, each with the last diagonal node it needs;
// the factorisation loop then queues them AS SOON AS that node has been queued (the host submits the helper streams' work
// interleaved with the main chain -- queued after the loop it would only reach the GPU when the factorisation is over).
// `second`: the subtree runs on the second inverse stream (the right half of the whole matrix follows the nodes of that half
// while the first stream is busy with the big T of the top level).
struct CholInvOp {
  int kind;   // 0: leaf (wait for node b0); 1: T = L21 L11i of (b0, bm, b1); 2: Li21 = -L22i T (after the right subtree)
  int b0, bm, b1, need;
  long toff;
  bool second, fork;
};
static void chol_inv_plan(const CholPipe &P, std::vector<CholInvOp> &ops, int b0, int b1, long toff, bool second, bool top) {
  if (b1 - b0 == 1) { ops.push_back({0, b0, b0, b1, b0, toff, second, false}); return; }
  const int bm = (b0 + b1 + 1) / 2;
  chol_inv_plan(P, ops, b0, bm, toff, second, false);
  const int a = P.off[b0], b = P.off[bm], m1 = b - a, m2 = P.off[b1 - 1] + P.bs[b1 - 1] - b;
  const bool fork = top && b1 - bm >= 4;
  ops.push_back({1, b0, bm, b1, bm, toff, second, fork});
  chol_inv_plan(P, ops, bm, b1, toff + (long)m2 * m1, second || fork, false);
  ops.push_back({2, b0, bm, b1, b1 - 1, toff, second, fork});
}
static int chol_inv_emit(const CholPipe &P, const CholInvOp &op) {
  const CholCtx &c = *P.c;
  const long n = c.n;
  hipStream_t sv = op.second ? P.as->inv2 : P.as->inv;
  float *G = op.second ? P.G_inv2 : P.G_inv;
  if (op.kind == 0) return check_hip(hipStreamWaitEvent(sv, P.as->event(3 * op.b0), 0), "hipStreamWaitEvent");
  const int a = P.off[op.b0], b = P.off[op.bm], m1 = b - a, m2 = P.off[op.b1 - 1] + P.bs[op.b1 - 1] - b;
  float *T = c.T + op.toff;
  if (op.kind == 1) {
    // T = L21 * L11i: L21 = the rows below of the left half's panels, complete once node bm has been queued
    int rc = check_hip(hipStreamWaitEvent(sv, P.as->event(3 * op.bm), 0), "hipStreamWaitEvent");
    if (rc != CLO_OK) return rc;
    return chol_gemm(c, sv, G, m2, m1, m1, 1.f, c.L + b * n + a, n, 1, c.Li + a * n + a, n, 1, 0.f, T, m1, TRI_KGE_N, 0);
  }
  if (op.fork) {
    hipEvent_t e = P.as->event(3 * (size_t)P.nblk + 3 + P.nblk);   // (behind the group events)
    if (P.as->err != CLO_OK) { set_error("cholesky inverse: hipEventCreate failed"); return P.as->err; }
    int rc = check_hip(hipEventRecord(e, P.as->inv2), "hipEventRecord");
    if (rc != CLO_OK) return rc;
    rc = check_hip(hipStreamWaitEvent(sv, e, 0), "hipStreamWaitEvent");
    if (rc != CLO_OK) return rc;
  }
  // Li21 = -L22i * T
  return chol_gemm(c, sv, G, m2, m1, m2, -1.f, c.Li + b * n + b, n, 1, T, m1, 1, 0.f, c.Li + b * n + a, n, TRI_KLT_M, 0);
}

static int chol_pipe_run(const CholCtx &c, float *G_side, float *G_inv, float *G_inv2, CholAsync *as);
static int chol_pipe(const CholCtx &c, float *G_side, float *G_inv, float *G_inv2) {
  int dev = 0;
  CholAsync *as = chol_async_acquire(&dev, c.st);
  if (!as) { set_error("cholesky inverse: cannot create the side streams"); return CLO_EHIP; }
  CholCtx cm = c;
  int rc = CLO_OK;
  if (as->main) {   // fork: the set's main stream takes over behind everything queued on the caller's stream
    rc = check_hip(hipEventRecord(as->fork, c.st), "hipEventRecord");
    if (rc == CLO_OK) rc = check_hip(hipStreamWaitEvent(as->main, as->fork, 0), "hipStreamWaitEvent");
    cm.st = as->main;
  }
  if (rc == CLO_OK) rc = chol_pipe_run(cm, G_side, G_inv, G_inv2, as);
  if (!as->done && hipEventCreateWithFlags(&as->done, hipEventDisableTiming) != hipSuccess) as->done = nullptr;
  if (as->done && rc == CLO_OK) rc = check_hip(hipEventRecord(as->done, cm.st), "hipEventRecord");
  if (as->main && rc == CLO_OK) {   // join: the caller's stream continues when the pipeline is through
    if (!as->done) rc = check_hip(hipStreamSynchronize(as->main), "hipStreamSynchronize");
    else rc = check_hip(hipStreamWaitEvent(c.st, as->done, 0), "hipStreamWaitEvent");
  }
  chol_async_release(as, dev);
  return rc;
}
static int chol_pipe_run(const CholCtx &c, float *G_side, float *G_inv, float *G_inv2, CholAsync *as) {
  CholPipe P;
  P.c = &c; P.as = as; P.G_side = G_side; P.G_inv = G_inv; P.G_inv2 = G_inv2;
  const int np = c.n;
  for (int o = 0; o < np;) {   // 128-row block columns; a short tail is shared with the block before it
    int m = std::min(QNB, np - o);
    const int left = np - o - m;
    if (left > 0 && left < PNB) m = (((np - o) / 2 + 3) & ~3);
    P.off.push_back(o); P.bs.push_back(m);
    o += m;
  }
  const int nb = P.nblk = (int)P.off.size();
  const long n = c.n;
  hipStream_t st = c.st;
#define CHOL_HIP(call) { int rc_ = check_hip(call, #call); if (rc_ != CLO_OK) return rc_; }
  // both helper streams start after everything queued on the caller's stream (the initialised S, L, Li)
  hipEvent_t e0 = as->event(3 * nb);
  if (as->err != CLO_OK) { set_error("cholesky inverse: hipEventCreate failed"); return as->err; }
  CHOL_HIP(hipEventRecord(e0, st));
  CHOL_HIP(hipStreamWaitEvent(as->side, e0, 0));
  CHOL_HIP(hipStreamWaitEvent(as->inv, e0, 0));
  CHOL_HIP(hipStreamWaitEvent(as->inv2, e0, 0));
  // Trailing updates in groups of W block columns: the panels of group g are applied by ONE symmetric product of depth
  // ~128 W on the side stream to everything from group g + 2 on; the columns of group g + 1 get them just in time from the
  // main stream's column update (left-looking over the panels of groups g and g + 1 so far: one skinny product of depth
  // <= 256 W), so the main stream meets the side stream once per group and the bulk runs at a useful depth.
#ifndef CLO_CHOL_GROUP
#define CLO_CHOL_GROUP 4
#endif
  static const int W = std::max(1, CLO_CHOL_GROUP);
  const int ngrp = (nb + W - 1) / W;
  auto grp_begin = [&](int g) { return g >= ngrp ? np : P.off[g * W]; };
  const size_t EV_GRP = 3 * (size_t)nb + 3;   // events of the group products
  std::vector<CholInvOp> inv_ops;
  chol_inv_plan(P, inv_ops, 0, nb, 0, false, true);
  size_t inv_pos = 0;
  for (int k = 0; k < nb; ++k) {
    const int o = P.off[k], m = P.bs[k], r0 = o + m, rest = np - r0;
    hipEvent_t ev_node = as->event(3 * k);
    if (as->err != CLO_OK) { set_error("cholesky inverse: hipEventCreate failed"); return as->err; }
    int rc;
    if (m <= PNB) {
      hipLaunchKernelGGL(potrf_diag_kernel, dim3(c.batch), dim3(64), 0, st, c.S + o * n + o, n, c.L + o * n + o, n, m,
                         c.Li + o * n + o, n, c.status, o, c.stride);
      CLO_CHECK_LAUNCH("potrf_diag_kernel");
    } else {
      rc = potrf_node128_launch(c.S + o * n + o, n, c.L + o * n + o, n, m, c.Li + o * n + o, n, c.status, o, c.stride,
                                c.batch, st);
      if (rc != CLO_OK) return rc;
    }
    CHOL_HIP(hipEventRecord(ev_node, st));
    for (; inv_pos < inv_ops.size() && inv_ops[inv_pos].need <= k; ++inv_pos) {
      rc = chol_inv_emit(P, inv_ops[inv_pos]);
      if (rc != CLO_OK) return rc;
    }
    if (rest == 0) break;
    // panel: L[r0:, o:o+m] = S[r0:, o:o+m] * Li_kk^T     (B(k,n) = Li_kk[n][k], zero for k > n)
    rc = chol_gemm(c, st, c.G, rest, m, m, 1.f, c.S + r0 * n + o, n, 1, c.Li + o * n + o, 1, n, 0.f, c.L + r0 * n + o, n,
                   TRI_KLT_N, 0);
    if (rc != CLO_OK) return rc;
    const int g = k / W;
    if ((k + 1) % W == 0) {
      // group g complete: its panels go to S[R:, R:], R = first row of group g + 2 (symmetric: half the tiles, mirrored)
      const int R = grp_begin(g + 2), og = P.off[g * W];
      if (R < np) {
        hipEvent_t ev_panel = as->event(3 * k + 1);
        if (as->err != CLO_OK) { set_error("cholesky inverse: hipEventCreate failed"); return as->err; }
        CHOL_HIP(hipEventRecord(ev_panel, st));
        CHOL_HIP(hipStreamWaitEvent(as->side, ev_panel, 0));
        rc = chol_gemm(c, as->side, P.G_side, np - R, np - R, r0 - og, -1.f, c.L + R * n + og, n, 1, c.L + R * n + og, 1, n,
                       1.f, c.S + R * n + R, n, 0, 1);
        if (rc != CLO_OK) return rc;
        hipEvent_t ev_grp = as->event(EV_GRP + g);
        if (as->err != CLO_OK) { set_error("cholesky inverse: hipEventCreate failed"); return as->err; }
        CHOL_HIP(hipEventRecord(ev_grp, as->side));
      }
    }
    // block column k + 1: everything the side stream has not applied to it = the panels of the group before its own and of
    // its own group so far, in one product.  (A split of this step into "what node k + 1 needs" on this stream and "the
    // rows below" on a fourth stream was built and measured 3 x SLOWER, 12.9 ms at n = 4608: two streams that wait for each
    // other once per block column pay the cross-stream hand-over twice per step.)
    const int cn = k + 1, gc = cn / W, m_next = P.bs[cn];
    if (cn % W == 0 && gc >= 2) CHOL_HIP(hipStreamWaitEvent(st, as->event(EV_GRP + gc - 2), 0));   // first column of a group
    const int j0 = gc >= 1 ? (gc - 1) * W : 0, oj = P.off[j0];
#ifndef CLO_CHOL_COL_SPLIT
#define CLO_CHOL_COL_SPLIT 1
#endif
    static const int col_split = CLO_CHOL_COL_SPLIT;
    rc = chol_gemm(c, st, col_split ? c.G : nullptr, rest, m_next, r0 - oj, -1.f, c.L + r0 * n + oj, n, 1, c.L + r0 * n + oj,
                   1, n, 1.f, c.S + r0 * n + r0, n, 0, 0);
    if (rc != CLO_OK) return rc;
  }
  for (; inv_pos < inv_ops.size(); ++inv_pos) {
    int rc = chol_inv_emit(P, inv_ops[inv_pos]);
    if (rc != CLO_OK) return rc;
  }
  // join: the caller's stream continues after the inverse (which waited for every node) and the last updates
  hipEvent_t e1 = as->event(3 * nb + 1), e2 = as->event(3 * nb + 2);
  if (as->err != CLO_OK) { set_error("cholesky inverse: hipEventCreate failed"); return as->err; }
  CHOL_HIP(hipEventRecord(e1, as->inv));
  CHOL_HIP(hipEventRecord(e2, as->side));
  CHOL_HIP(hipStreamWaitEvent(st, e1, 0));
  CHOL_HIP(hipStreamWaitEvent(st, e2, 0));
#undef CHOL_HIP
  return CLO_OK;
}

static inline long pad4(long n) { return (n + 3) & ~3L; }
static const long kCholSlabFloats = 16L * 256 * 256;

}  // namespace clo

using namespace clo;

// workspace of a batch of `batch` equally sized n x n factors
extern "C" long clo_cholesky_inverse_batched_ws_floats(int n, int batch) {
  const long np = pad4(n), nn = np * np;
  // per matrix: S, L, Li, T (shares the stride), padded product; plus the split-K slabs
  return (long)batch * 5 * nn + 4L * batch * kCholSlabFloats + 1024;   // (a slab region per stream of the pipeline)
}
extern "C" long clo_cholesky_inverse_ws_floats(int n) { return clo_cholesky_inverse_batched_ws_floats(n, 1); }

// out_b = (A_b + damping_b I)^-1 for b < batch: symmetric positive definite n x n factors of EQUAL
// size (row-major, lda_b), out_b row-major ldo_b.  One chain of launches serves the whole batch: the
// leaves run one workgroup per matrix, every product is a batched GEMM -- the many small launches of
// the recursion, which leave most of the chip idle for a single factor, are shared by all factors
// (transformer blocks and ResNet stages repeat the same layer shapes).
// The pointer / stride / damping arrays are HOST arrays.  ws: clo_cholesky_inverse_batched_ws_floats
// floats; status: `batch` device ints (zeroed here) = offending pivot of each factor.
extern "C" int clo_cholesky_inverse_batched_f32(const float *const *A, const long *lda, float *const *out,
                                                const long *ldo, int n, int batch, const float *damping,
                                                float *ws, int *status, void *stream) {
  CLO_REQUIRE(n >= 0 && batch >= 0, "clo_cholesky_inverse_batched_f32: bad sizes");
  if (n == 0 || batch == 0) return CLO_OK;
  CLO_REQUIRE(A && lda && out && ldo && damping && ws && status, "clo_cholesky_inverse_batched_f32: null pointer");
  for (int b = 0; b < batch; ++b)
    CLO_REQUIRE(A[b] && out[b] && lda[b] >= n && ldo[b] >= n, "clo_cholesky_inverse_batched_f32: bad operand %d", b);
  hipStream_t st = (hipStream_t)stream;
  int rc = check_hip(hipMemsetAsync(status, 0, sizeof(int) * batch, st), "hipMemsetAsync");
  if (rc != CLO_OK) return rc;
  const int np = (int)pad4(n);
  const long nn = (long)np * np;
  CholCtx c;
  c.stride = nn;
  c.S = ws; c.L = ws + (long)batch * nn; c.Li = ws + 2L * batch * nn; c.T = ws + 3L * batch * nn;
  float *prod = ws + 4L * batch * nn;
  c.G = ws + 5L * batch * nn;
  c.gws = (long)batch * kCholSlabFloats;
  c.n = np; c.batch = batch; c.status = status; c.st = st;
  const unsigned iblocks = (unsigned)std::min<long>(cdiv(nn, 256), kNumCU * 8L);
  for (int b0 = 0; b0 < batch; b0 += InitBatch::MAX) {
    InitBatch ib{};
    const int cnt = std::min(InitBatch::MAX, batch - b0);
    for (int i = 0; i < cnt; ++i) { ib.A[i] = A[b0 + i]; ib.lda[i] = lda[b0 + i]; ib.damping[i] = damping[b0 + i]; }
    hipLaunchKernelGGL(chol_init_kernel, dim3(iblocks, cnt), dim3(256), 0, st, ib, c.S + (long)b0 * nn,
                       c.L + (long)b0 * nn, c.Li + (long)b0 * nn, n, np, nn);
    CLO_CHECK_LAUNCH("chol_init_kernel");
  }
#ifndef CLO_CHOL_PIPE
#define CLO_CHOL_PIPE 1
#endif
  static const int pipe = [] {
    const char *e = getenv("CLO_CHOL_PIPE");   // (A/B runs; not a product knob)
    return e ? atoi(e) : CLO_CHOL_PIPE;
  }();
  // (below ~1500 rows the single chain of chol_rec is as fast or faster: n = 512: 0.34 vs 0.43 ms, 1152: 0.90 vs 0.88 ms,
  // 2304: 2.11 vs 1.78 ms, 4608: 5.13 vs 3.95 ms)
#ifndef CLO_CHOL_PIPE_MIN
#define CLO_CHOL_PIPE_MIN 1536
#endif
  static const int pipe_min = CLO_CHOL_PIPE_MIN;
  // The pipeline serves the SINGLE large factor (n = 4608: 5.1 -> 4.0 ms).  Equal-size batches -- the groups of an
  // operator-level inverse -- run the plain chain: their launches are already batch-wide, and the operator drives several
  // groups side by side from worker threads (linalg_native.INVERSE_WORKERS); chain + 3 workers measured 10.3 - 11.5 ms for
  // ResNet-18's 42 factors at 4 AND 16 hardware queues, first call included, against 13.4 - 15 ms with a pipeline per group
  // (profiles/r06_cholesky_streams.txt).
  if (pipe && batch == 1 && np >= pipe_min && np > QNB)
    rc = chol_pipe(c, c.G + c.gws, c.G + 2 * c.gws, c.G + 3 * c.gws);
  else
    rc = chol_rec(c, 0, np);
  if (rc != CLO_OK) return rc;
  // A^-1 = Li^T Li: A(m,k) = Li[k][m] and B(k,n) = Li[k][n] are zero for k < m / k < n.  A single
  // factor with float4-complete rows goes straight to the caller's tensor.
  const bool direct = batch == 1 && np == n;
  GemmArgs g{};
  g.M = np; g.N = np; g.K = np; g.alpha = 1.f; g.beta = 0.f;
  g.A = c.Li; g.sa_m = 1; g.sa_k = np; g.sa_b = nn; g.B = c.Li; g.sb_k = np; g.sb_n = 1; g.sb_b = nn;
  g.C = direct ? out[0] : prod; g.ldc = direct ? ldo[0] : np; g.sc_b = nn;
  g.sym = 1; g.tri = TRI_KGE_M | TRI_KGE_N;
  rc = launch_gemm_auto(g, c.G, c.gws, st, batch);
  if (rc != CLO_OK || direct) return rc;
  const unsigned oblocks = (unsigned)std::min<long>(cdiv((long)n * n, 256), kNumCU * 8L);
  for (int b0 = 0; b0 < batch; b0 += OutBatch::MAX) {
    OutBatch ob{};
    const int cnt = std::min(OutBatch::MAX, batch - b0);
    for (int i = 0; i < cnt; ++i) { ob.out[i] = out[b0 + i]; ob.ldo[i] = ldo[b0 + i]; }
    hipLaunchKernelGGL(chol_copy_out_kernel, dim3(oblocks, cnt), dim3(256), 0, st, ob, prod + (long)b0 * nn, n,
                       np, nn);
    CLO_CHECK_LAUNCH("chol_copy_out_kernel");
  }
  return CLO_OK;
}

// out = (A + damping I)^-1, A symmetric positive definite n x n (row-major, lda), out row-major ldo.
// ws: clo_cholesky_inverse_ws_floats(n) floats; *status (device int, zeroed here) = offending pivot.
extern "C" int clo_cholesky_inverse_f32(const float *A, long lda, float *out, long ldo, int n,
                                        float damping, float *ws, int *status, void *stream) {
  CLO_REQUIRE(n >= 0 && lda >= n && ldo >= n, "clo_cholesky_inverse_f32: bad sizes");
  if (n == 0) return CLO_OK;
  CLO_REQUIRE(A && out && ws && status, "clo_cholesky_inverse_f32: null pointer");
  return clo_cholesky_inverse_batched_f32(&A, &lda, &out, &ldo, n, 1, &damping, ws, status, stream);
}

extern "C" int clo_potrf_diag_f32(float *A, long lda, int nb, float *Linv, long ldinv, int *status,
                                  int pivot_base, void *stream) {
  CLO_REQUIRE(nb >= 1 && nb <= QNB, "clo_potrf_diag_f32: nb must be in [1, %d], got %d", QNB, nb);
  CLO_REQUIRE(A && Linv && status && lda >= nb && ldinv >= nb, "clo_potrf_diag_f32: bad operand");
  if (nb > PNB)  // 65 ... 128 rows: the four-wave kernel
    return potrf_node128_launch(A, lda, A, lda, nb, Linv, ldinv, status, pivot_base, 0L, 1, (hipStream_t)stream);
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, lda, A, lda,
                     nb, Linv, ldinv, status, pivot_base, 0L);
  CLO_CHECK_LAUNCH("potrf_diag_kernel");
  return CLO_OK;
}

#ifdef CLO_POTRF_TIMING
extern "C" void clo_potrf_timing_set(unsigned long long *device_buffer) { clo::g_potrf_stamps = device_buffer; }
#endif
