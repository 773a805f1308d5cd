// Tall-skinny Gram matrices  C = beta C + alpha [X | 1]^T [X | 1]  with rows >> d, d + 1 <= 128.
//
// KFAC's factors of convolution layers (G_l: d_out = 6 ... 128 columns against B * H * W rows;
// A_1: C_in k^2 + 1 = 26 ... 28 columns), the Gram passes of the Hutch++ range basis ([85M, 32]) --
// reference computers/kfac_hooks.py:350,390 (einsum "b s i, b s j -> i j"), trace/meyer2020hutch.py:93.
// The 128 x 128 tile engine of gemm.hip spends its MFMAs on padding there and can only split K into 64
// slabs; this kernel instead streams X exactly once, linearly, at the HBM rate:
//
//   * persistent blocks (a few per CU) walk row chunks b, b + grid, ...; a chunk is staged in LDS as
//     [rows][P] (P = d + 1 padded to 16 / 32 / 64 / 96 / 128) with 16-byte global loads;
//   * the chunk's rows are the K dimension of v_mfma_f32_16x16x4 (P = 16) or 32x32x2 tiles; both
//     operands of a tile are columns of the SAME LDS rows, so a diagonal tile needs one ds_read_b32
//     per MFMA;  P <= 64: the four waves split the rows and hold all upper tiles; P > 64: wave w owns
//     tile row w;
//   * per-block partial Grams go to slabs, a small second kernel sums them (fixed order:
//     deterministic), applies alpha / beta and mirrors the lower triangle.
#include "clo_common.h"

namespace clo {

using f32x4g = __attribute__((ext_vector_type(4))) float;
using f32x16g = __attribute__((ext_vector_type(16))) float;

constexpr int GT_THREADS = 256;
constexpr int GT_LDS_FLOATS = 8192;  // 32 KiB staging area per block
// rows per chunk for a padded width P: a multiple of 8 (four waves x an even count)
constexpr int gram_rc(int P) { return (GT_LDS_FLOATS / P) / 8 * 8; }

struct GramArgs {
  const float *X;
  long rows, ldx;
  int d, ones, dd;
  float *slab;  // [grid][dd][dd] (upper tiles only are written)
  int vec;      // 16-byte loads allowed
};

// Stage rows [r0, r0 + RC) of [X | 1] into S[RC][P]; rows beyond `rows` and columns >= dd are zero.
template <int P, int RC>
__device__ __forceinline__ void gram_stage(const GramArgs &p, long r0, float *S, int tid) {
  const int d = p.d;
  if (p.vec) {
    const int q4 = d >> 2;
    for (int e = tid; e < RC * q4; e += GT_THREADS) {
      const int r = e / q4, q = e - r * q4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + r < p.rows) v = *reinterpret_cast<const float4 *>(p.X + (r0 + r) * p.ldx + 4 * q);
      *reinterpret_cast<float4 *>(S + r * P + 4 * q) = v;
    }
  } else {
    for (int e = tid; e < RC * d; e += GT_THREADS) {
      const int r = e / d, c = e - r * d;
      S[r * P + c] = (r0 + r < p.rows) ? p.X[(r0 + r) * p.ldx + c] : 0.f;
    }
  }
  const int npad = P - d;
  for (int e = tid; e < RC * npad; e += GT_THREADS) {
    const int r = e / npad, c = d + (e - r * npad);
    S[r * P + c] = (p.ones && c == d && r0 + r < p.rows) ? 1.f : 0.f;
  }
}

// P = 16: one 16x16x4 tile, four waves split the rows of the chunk.
__global__ __launch_bounds__(GT_THREADS) void gram16_kernel(const GramArgs p) {
  constexpr int P = 16, RC = gram_rc(P);  // 512 rows per chunk
  __shared__ __attribute__((aligned(16))) float S[GT_LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, ks = lane >> 4;
  f32x4g acc = {0.f, 0.f, 0.f, 0.f};
  const long nchunks = cdiv(p.rows, (long)RC);
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    gram_stage<P, RC>(p, c * RC, S, tid);
    __syncthreads();
    const float *s = S + (wave * (RC / 4) + ks) * P + i;
#pragma unroll 8
    for (int r = 0; r < RC / 4; r += 4) {
      const float a = s[r * P];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // cross-wave sum through LDS; D layout: row = 4 (lane >> 4) + r, col = lane & 15
#pragma unroll
  for (int r = 0; r < 4; ++r) S[(wave * 16 + 4 * ks + r) * 16 + i] = acc[r];
  __syncthreads();
  const int e = tid;  // 256 threads = 16 x 16 entries
  const int row = e >> 4, col = e & 15;
  const float v = S[row * 16 + col] + S[(16 + row) * 16 + col] + S[(32 + row) * 16 + col] + S[(48 + row) * 16 + col];
  if (row < p.dd && col < p.dd) p.slab[(long)blockIdx.x * p.dd * p.dd + row * p.dd + col] = v;
}

// P = 32 T: T x T tiles of 32 x 32 (upper triangle computed).  T <= 2: waves split the rows and hold
// every upper tile; T >= 3: wave w owns tile row w over all rows of the chunk.
template <int T>
__global__ __launch_bounds__(GT_THREADS) void gram32_kernel(const GramArgs p) {
  constexpr int P = 32 * T, RC = gram_rc(P);
  constexpr bool KSPLIT = T <= 2;
  constexpr int NT = KSPLIT ? T * (T + 1) / 2 : T;  // accumulator tiles per wave
  __shared__ __attribute__((aligned(16))) float S[GT_LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, ks = lane >> 5;
  f32x16g acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const long nchunks = cdiv(p.rows, (long)RC);
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    gram_stage<P, RC>(p, c * RC, S, tid);
    __syncthreads();
    if (KSPLIT) {
      const float *s = S + (wave * (RC / 4) + ks) * P + i;
#pragma unroll 4
      for (int r = 0; r < RC / 4; r += 2) {
        float x[T];
#pragma unroll
        for (int t = 0; t < T; ++t) x[t] = s[r * P + 32 * t];
        int k = 0;
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
          for (int b = a; b < T; ++b, ++k)
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[a], x[b], acc[k], 0, 0, 0);
      }
    } else if (wave < T) {
      const float *s = S + ks * P + i;
#pragma unroll 2
      for (int r = 0; r < RC; r += 2) {
        const float xa = s[r * P + 32 * wave];
#pragma unroll
        for (int b = 0; b < T; ++b) {
          if (b < wave) continue;  // (uniform per wave)
          const float xb = s[r * P + 32 * b];
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, xb, acc[b], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // ---- write the block's partial Gram.  D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float *out = p.slab + (long)blockIdx.x * p.dd * p.dd;
  if (KSPLIT) {
    // sum the four waves' tiles through LDS, one tile at a time (4 x 4 KiB)
    int k = 0;
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = a; b < T; ++b, ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * ks;
          S[wave * 1024 + row * 32 + i] = acc[k][r];
        }
        __syncthreads();
        for (int e = tid; e < 1024; e += GT_THREADS) {
          const int row = 32 * a + (e >> 5), col = 32 * b + (e & 31);
          if (row < p.dd && col < p.dd) out[row * p.dd + col] = (S[e] + S[1024 + e]) + (S[2048 + e] + S[3072 + e]);
        }
        __syncthreads();
      }
  } else if (wave < T) {
#pragma unroll
    for (int b = 0; b < T; ++b) {
      if (b < wave) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * ks, col = 32 * b + i;
        if (row < p.dd && col < p.dd) out[row * p.dd + col] = acc[b][r];
      }
    }
  }
}

// C = beta C + alpha sum_b slab[b]; the slabs hold the upper 32-wide tiles, the rest is mirrored.
__global__ void gram_reduce_kernel(float *__restrict__ C, long ldc, const float *__restrict__ slab,
                                   int nslab, int dd, int tile, float alpha, float beta) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= dd * dd) return;
  const int row = e / dd, col = e - row * dd;
  const bool upper = (row / tile) <= (col / tile);
  const long src = upper ? (long)row * dd + col : (long)col * dd + row;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long st = (long)dd * dd;
  int b = 0;
  for (; b + 3 < nslab; b += 4) {
    s0 += slab[b * st + src]; s1 += slab[(b + 1) * st + src];
    s2 += slab[(b + 2) * st + src]; s3 += slab[(b + 3) * st + src];
  }
  for (; b < nslab; ++b) s0 += slab[b * st + src];
  float *c = C + (long)row * ldc + col;
  *c = (beta != 0.f ? beta * *c : 0.f) + alpha * ((s0 + s1) + (s2 + s3));
}

static int gram_grid(long rows, int P) {
  const long nchunks = cdiv(rows, (long)gram_rc(P));
  return (int)std::max<long>(1, std::min<long>(nchunks, 4L * kNumCU));
}
static int gram_pad(int dd) { return dd <= 16 ? 16 : 32 * (int)cdiv(dd, 32); }

}  // namespace clo

using namespace clo;

// Worth it when the matrix is tall (rows >= 32 (d + 1), d + 1 <= 128) AND the aligned symmetric GEMM with
// its (now symmetric-aware) split-K would not be faster.  Measured (tools/probe_gram_vs_gemm.py): up to 32
// columns the streaming kernel wins at every height (131072 x 16: 25 vs 67 us); at 48 / 64 columns only
// beyond ~300 k rows (131072 x 64: 122 vs 71 us; 524288 x 64: 149 vs 248 us); from 96 columns the GEMM wins
// (131072 x 128: 278 vs 81 us).  Widths the aligned engine cannot take (d % 4 != 0, or a bias column on top
// of an aligned d) stay here: their alternative is the scalar-load engine.
extern "C" int clo_gram_tall_supported(long rows, int d, int ones_col) {
  const int dd = d + (ones_col ? 1 : 0);
  if (!(dd >= 1 && dd <= 128 && rows >= 32L * dd)) return 0;
  const bool gemm_aligned = d % 4 == 0 && d >= 4;   // [X | 1] with d % 4 == 0 is also fine for the v2 loader
  if (!gemm_aligned || dd <= 32) return 1;
  if (dd <= 64 + 1) return rows >= 300000 ? 1 : 0;
  return 0;
}
extern "C" long clo_gram_tall_ws_floats(long rows, int d, int ones_col) {
  const int dd = d + (ones_col ? 1 : 0);
  if (dd < 1 || dd > 128) return 0;
  return (long)gram_grid(rows, gram_pad(dd)) * dd * dd;
}

// C = beta C + alpha [X | 1]^T [X | 1]   (X row-major [rows][ldx], first d columns; C [dd][ldc]).
extern "C" int clo_gram_tall_f32(float *C, long ldc, const float *X, long rows, int d, long ldx,
                                 int ones_col, float alpha, float beta, float *ws, void *stream) {
  const int dd = d + (ones_col ? 1 : 0);
  CLO_REQUIRE(d >= 0 && rows >= 0 && dd >= 1 && dd <= 128, "clo_gram_tall_f32: needs 1 <= d + ones <= 128");
  CLO_REQUIRE(ldc >= dd && ldx >= d, "clo_gram_tall_f32: leading dimensions too small");
  CLO_REQUIRE(C && ws && (X || rows == 0 || d == 0), "clo_gram_tall_f32: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int P = gram_pad(dd);
  GramArgs a{};
  a.X = X; a.rows = rows; a.ldx = ldx; a.d = d; a.ones = ones_col ? 1 : 0; a.dd = dd; a.slab = ws;
  a.vec = (d % 4 == 0 && ldx % 4 == 0 && aligned16(X)) ? 1 : 0;
  const int grid = gram_grid(rows, P);
  switch (P) {
    case 16: hipLaunchKernelGGL(gram16_kernel, dim3(grid), dim3(GT_THREADS), 0, st, a); break;
    case 32: hipLaunchKernelGGL(gram32_kernel<1>, dim3(grid), dim3(GT_THREADS), 0, st, a); break;
    case 64: hipLaunchKernelGGL(gram32_kernel<2>, dim3(grid), dim3(GT_THREADS), 0, st, a); break;
    case 96: hipLaunchKernelGGL(gram32_kernel<3>, dim3(grid), dim3(GT_THREADS), 0, st, a); break;
    default: hipLaunchKernelGGL(gram32_kernel<4>, dim3(grid), dim3(GT_THREADS), 0, st, a); break;
  }
  CLO_CHECK_LAUNCH("gram_kernel");
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)cdiv(dd * dd, 256)), dim3(256), 0, st, C, ldc, ws, grid,
                     dd, P == 16 ? 16 : 32, alpha, beta);
  CLO_CHECK_LAUNCH("gram_reduce_kernel");
  return CLO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Round 6: the tall-skinny algebra of the randomised trace estimators (reference trace/meyer2020hutch.py:86-102,
// trace/epperly2024xtrace.py:52-101) as HBM-streaming kernels.
//
//   clo_tall_gram_f64    out[n1][n2] = X^T Y  for X [m][n1], Y [m][n2] float32, n1, n2 <= 64, EXACT products and float64
//                        accumulation (v_mfma_f64_16x16x4_f64: 24 x 24-bit products fit the 53-bit significand).  The
//                        range basis of Hutch++ is built from Gram matrices of an [85 M, 32] block whose singular values
//                        span many decades; a float32 Gram matrix resolves directions down to sigma / sigma_max ~ 3e-3
//                        only (sqrt(eps)), this one down to the rounding noise of the float32 DATA (~1e-7), which is what
//                        the reference's Householder QR of the same float32 block resolves.  Also the projections Q^T G.
//   clo_tall_apply_f32   out = beta G + Q C  for Q [m][k], C [k][n] (k, n <= 64), G / out [m][n]: the rows are the M
//                        dimension of v_mfma_f32_16x16x4 tiles, C lives in registers; one pass over Q, G and out
//                        (Q = X T of the basis, G - Q (Q^T G) of the deflation).
// Both read every operand exactly once with every lane of every CU streaming; bytes = 4 m (n1 + n2) resp. 4 m (k + 2 n).
// ------------------------------------------------------------------------------------------------------------------
namespace clo {

using f64x4g = __attribute__((ext_vector_type(4))) double;

constexpr int TG_THREADS = 256, TG_UNROLL = 8;

struct TallGramArgs {
  const float *X, *Y;
  long ldx, ldy, m;
  int n1, n2, sym, vec;
  double *slab;   // [grid][16 NI][16 NJ]
};

// NI x NJ tiles of 16 x 16.  A chunk of TG_RC rows of [X | Y] is staged in LDS with 16-byte global loads (unconditional,
// clamped addresses; the loads of chunk c + 1 are in flight while chunk c is multiplied), then every wave walks its quarter of
// the chunk in steps of 4 rows (the K dimension of one f64 MFMA): lane (c = lane % 16, r = lane / 16) reads row r, column
// 16 t + c of the staged slice.  The first version let each lane load its element straight from global memory -- 64-byte
// requests, 1.0 - 1.2 TB/s on the [85 M, 32] blocks of C5 (profiles/r06_c5_hutchpp_kernel_split.txt, first run: 13 ms per
// projection); staged, the pass is bound by HBM and the f64 matrix pipe together (~2 ms).
constexpr int TG_RC = 64;
template <int NI, int NJ, bool SYM>
__global__ __launch_bounds__(TG_THREADS) void tall_gram_kernel(const TallGramArgs p) {
  constexpr int P1 = 16 * NI, P2 = SYM ? 0 : 16 * NJ;
  constexpr int PT = P1 + P2;                              // (the symmetric form stages X only)
  constexpr int PITCH = ((PT + 15) / 32) * 32 + 16;        // = 16 mod 32: the four rows of a ds_read_b32 hit disjoint banks
  constexpr int QPR = PT / 4;                              // 16-byte pieces per staged row
  constexpr int NLD = (TG_RC * QPR + TG_THREADS - 1) / TG_THREADS;
  __shared__ __attribute__((aligned(16))) float St[TG_RC * PITCH];
  __shared__ double S[4 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, r = lane >> 4;
  constexpr bool sym = SYM;
  const int n1 = p.n1, n2 = sym ? 0 : p.n2;
  const bool vec = p.vec != 0;
  f64x4g acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f64x4g{0., 0., 0., 0.};
  // this thread's pieces of a chunk: (row, staged column); staged columns [0, P1) come from X, [P1, PT) from Y
  int prow[NLD], pcol[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int e = min(tid + u * TG_THREADS, TG_RC * QPR - 1);
    prow[u] = e / QPR;
    pcol[u] = (e - prow[u] * QPR) * 4;
  }
  const long mlast = max(p.m - 1, 0L);
  auto fetch = [&](long r0, float4 (&v)[NLD]) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const long row = r0 + prow[u];
      const bool rok = row < p.m;
      const int sc = pcol[u];
      const bool fromx = sc < P1;
      const int col = fromx ? sc : sc - P1;
      const int nn = fromx ? n1 : n2;
      const float *base = (fromx ? p.X + min(row, mlast) * p.ldx : p.Y + min(row, mlast) * p.ldy);
      if (vec) {
        const float4 t = *reinterpret_cast<const float4 *>(base + max(min(col, nn - 4), 0));
        const bool in = rok && col < nn;
        v[u] = make_float4(in ? t.x : 0.f, in ? t.y : 0.f, in ? t.z : 0.f, in ? t.w : 0.f);
      } else {
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = nn > 0 ? base[min(col + e, nn - 1)] : 0.f;
          t[e] = (rok && col + e < nn) ? x : 0.f;
        }
        v[u] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  };
  const long nchunks = cdiv(p.m, (long)TG_RC);
  float4 cur[NLD], nxt[NLD];
  if ((long)blockIdx.x < nchunks) fetch((long)blockIdx.x * TG_RC, cur);
  for (long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    __syncthreads();   // everybody is done with the previous chunk
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      if (tid + u * TG_THREADS < TG_RC * QPR) *reinterpret_cast<float4 *>(&St[prow[u] * PITCH + pcol[u]]) = cur[u];
    fetch((ch + gridDim.x) * TG_RC, nxt);   // (past the end: masked, clamped)
    __syncthreads();
    const float *sp = St + (wave * (TG_RC / 4) + r) * PITCH + c;
#pragma unroll
    for (int st = 0; st < TG_RC / 16; ++st) {
      double a[NI], b[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = (double)sp[st * 4 * PITCH + 16 * i];
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = sym ? (j < NI ? a[j < NI ? j : 0] : 0.) : (double)sp[st * 4 * PITCH + P1 + 16 * j];
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (sym && j < i) continue;   // (uniform: the reduce kernel mirrors)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) cur[u] = nxt[u];
  }
  // the four waves' tiles are summed through LDS in a fixed order; D layout of the f64 16x16x4 MFMA: row = (lane / 16) + 4 q,
  // col = lane % 16 (one row per register and lane group -- NOT the 4 (lane / 16) + q of the f32 16x16x4 tile)
  __syncthreads();
  double *out = p.slab + (long)blockIdx.x * (16 * NI) * (16 * NJ);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (sym && j < i) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) S[wave * 256 + (r + 4 * q) * 16 + c] = acc[i][j][q];
      __syncthreads();
      {
        const int e = tid, row = e >> 4, col = e & 15;
        out[(long)(16 * i + row) * (16 * NJ) + 16 * j + col] = (S[e] + S[256 + e]) + (S[512 + e] + S[768 + e]);
      }
      __syncthreads();
    }
}

// out[a][b] = sum over blocks (fixed order), mirrored for the symmetric form
__global__ void tall_gram_reduce_kernel(double *__restrict__ out, long ldo, const double *__restrict__ slab, int nslab,
                                        int n1, int n2, int P1, int P2, int sym) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n1 * n2) return;
  const int a = e / n2, b = e - a * n2;
  const bool upper = !sym || (a / 16) <= (b / 16);
  const long src = upper ? (long)a * P2 + b : (long)b * P2 + a;
  const long st = (long)P1 * P2;
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
  int k = 0;
  for (; k + 3 < nslab; k += 4) {
    s0 += slab[k * st + src]; s1 += slab[(k + 1) * st + src];
    s2 += slab[(k + 2) * st + src]; s3 += slab[(k + 3) * st + src];
  }
  for (; k < nslab; ++k) s0 += slab[k * st + src];
  out[(long)a * ldo + b] = (s0 + s1) + (s2 + s3);
}

static int tall_gram_grid(long m) {
  return (int)std::max<long>(1, std::min<long>(cdiv(m, (long)TG_RC), 4L * kNumCU));
}

struct TallApplyArgs {
  float *out;
  const float *G, *Q, *C;
  long ldo, ldg, ldq, ldc, m;
  int k, n;
  float beta;
};

// KH = ceil(k / 16), NT = ceil(n / 16).  A tile = 16 rows: lane (i = lane % 16, g = lane / 16) loads float4
// Q[row i][16 h + 4 g ..]; MFMA step (h, e) contracts over k = 16 h + 4 g + e on both operands.
template <int KH, int NT>
__global__ __launch_bounds__(256) void tall_apply_kernel(const TallApplyArgs p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  float cb[KH][4][NT];
#pragma unroll
  for (int h = 0; h < KH; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int kk = 16 * h + 4 * g + e, col = 16 * t + i;
        cb[h][e][t] = (kk < p.k && col < p.n) ? p.C[(long)kk * p.ldc + col] : 0.f;
      }
  const long ntiles = cdiv(p.m, 16L);
  const long nw = (long)gridDim.x * 4;
  const bool has_g = p.G != nullptr && p.beta != 0.f;
  for (long t0 = (long)blockIdx.x * 4 + wave; t0 < ntiles; t0 += nw) {
    const long r0 = t0 * 16;
    const long rowq = min(r0 + i, p.m - 1);
    float4 q4[KH];
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      const int kk = 16 * h + 4 * g;
      q4[h] = kk < p.k ? *reinterpret_cast<const float4 *>(p.Q + rowq * p.ldq + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4g acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long row = r0 + 4 * g + q;
        const int col = 16 * t + i;
        acc[t][q] = (has_g && row < p.m && col < p.n) ? p.beta * p.G[row * p.ldg + col] : 0.f;
      }
    }
#pragma unroll
    for (int h = 0; h < KH; ++h) {
#define CLO_TA_MM(E, EI)                                                                        \
  _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                \
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[h].E, cb[h][EI][t], acc[t], 0, 0, 0);
      CLO_TA_MM(x, 0) CLO_TA_MM(y, 1) CLO_TA_MM(z, 2) CLO_TA_MM(w, 3)
#undef CLO_TA_MM
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long row = r0 + 4 * g + q;
        const int col = 16 * t + i;
        if (row < p.m && col < p.n) p.out[row * p.ldo + col] = acc[t][q];
      }
  }
}

}  // namespace clo

extern "C" long clo_tall_gram_ws_bytes(long m, int n1, int n2) {
  if (n1 < 1 || n2 < 1 || n1 > 64 || n2 > 64) return 0;
  const long P1 = 16 * cdiv(n1, 16), P2 = 16 * cdiv(n2, 16);
  return (long)tall_gram_grid(m) * P1 * P2 * (long)sizeof(double);
}

// out[n1][ldo] (float64) = X^T Y; Y == nullptr (or Y == X with n1 == n2, ldx == ldy): the symmetric Gram X^T X.
extern "C" int clo_tall_gram_f64(double *out, long ldo, const float *X, long ldx, int n1, const float *Y, long ldy,
                                 int n2, long m, void *ws, void *stream) {
  CLO_REQUIRE(n1 >= 1 && n1 <= 64 && n2 >= 1 && n2 <= 64 && m >= 0, "clo_tall_gram_f64: needs 1 <= n1, n2 <= 64");
  CLO_REQUIRE(out && X && ws && ldo >= n2 && ldx >= n1, "clo_tall_gram_f64: bad operand");
  const bool sym = Y == nullptr || (Y == X && n1 == n2 && ldx == ldy);
  if (!sym) CLO_REQUIRE(ldy >= n2, "clo_tall_gram_f64: ldy too small");
  if (sym) CLO_REQUIRE(n1 == n2, "clo_tall_gram_f64: the symmetric form needs n1 == n2");
  hipStream_t st = (hipStream_t)stream;
  TallGramArgs a{};
  a.X = X; a.Y = sym ? X : Y; a.ldx = ldx; a.ldy = sym ? ldx : ldy; a.m = m; a.n1 = n1; a.n2 = n2; a.sym = sym ? 1 : 0;
  a.slab = static_cast<double *>(ws);
  a.vec = ((n1 & 3) == 0 && n1 >= 4 && (ldx & 3) == 0 && aligned16(X) &&
           (sym || ((n2 & 3) == 0 && n2 >= 4 && (ldy & 3) == 0 && aligned16(Y)))) ? 1 : 0;
  const int NI = (int)cdiv(n1, 16), NJ = (int)cdiv(n2, 16);
  const int grid = tall_gram_grid(m);
#define CLO_TG_CASE(I, J)                                                                                      \
  if (NI == I && NJ == J) {                                                                                    \
    if (sym) { if (I == J) hipLaunchKernelGGL((tall_gram_kernel<I, (I == J ? J : I), true>), dim3(grid), dim3(TG_THREADS), 0, st, a); } \
    else hipLaunchKernelGGL((tall_gram_kernel<I, J, false>), dim3(grid), dim3(TG_THREADS), 0, st, a);          \
  }
  CLO_TG_CASE(1, 1) CLO_TG_CASE(1, 2) CLO_TG_CASE(1, 3) CLO_TG_CASE(1, 4)
  CLO_TG_CASE(2, 1) CLO_TG_CASE(2, 2) CLO_TG_CASE(2, 3) CLO_TG_CASE(2, 4)
  CLO_TG_CASE(3, 1) CLO_TG_CASE(3, 2) CLO_TG_CASE(3, 3) CLO_TG_CASE(3, 4)
  CLO_TG_CASE(4, 1) CLO_TG_CASE(4, 2) CLO_TG_CASE(4, 3) CLO_TG_CASE(4, 4)
#undef CLO_TG_CASE
  CLO_CHECK_LAUNCH("tall_gram_kernel");
  hipLaunchKernelGGL(tall_gram_reduce_kernel, dim3((unsigned)cdiv(n1 * n2, 256)), dim3(256), 0, st, out, ldo,
                     static_cast<const double *>(ws), grid, n1, n2, 16 * NI, 16 * NJ, a.sym);
  CLO_CHECK_LAUNCH("tall_gram_reduce_kernel");
  return CLO_OK;
}

// out[m][ldo] = beta G + Q C   (Q [m][ldq] with k columns, C [k][ldc] with n columns; G may be null or alias out)
extern "C" int clo_tall_apply_f32(float *out, long ldo, const float *G, long ldg, float beta, const float *Q, long ldq,
                                  const float *C, long ldc, long m, int k, int n, void *stream) {
  CLO_REQUIRE(k >= 1 && k <= 64 && n >= 1 && n <= 64 && m >= 0, "clo_tall_apply_f32: needs 1 <= k, n <= 64");
  CLO_REQUIRE(out && Q && C && ldo >= n && ldq >= k && ldc >= n && (!G || ldg >= n), "clo_tall_apply_f32: bad operand");
  CLO_REQUIRE(k % 4 == 0 && ldq % 4 == 0 && aligned16(Q), "clo_tall_apply_f32: Q rows must be 16-byte aligned, k % 4 == 0");
  if (m == 0) return CLO_OK;
  hipStream_t st = (hipStream_t)stream;
  TallApplyArgs a{};
  a.out = out; a.G = G; a.Q = Q; a.C = C; a.ldo = ldo; a.ldg = ldg; a.ldq = ldq; a.ldc = ldc; a.m = m; a.k = k; a.n = n;
  a.beta = beta;
  const int KH = (int)cdiv(k, 16), NT = (int)cdiv(n, 16);
  const int grid = (int)std::max<long>(1, std::min<long>(cdiv(cdiv(m, 16L), 4L), 8L * kNumCU));
#define CLO_TA_CASE(H, T) \
  if (KH == H && NT == T) hipLaunchKernelGGL((tall_apply_kernel<H, T>), dim3(grid), dim3(256), 0, st, a);
  CLO_TA_CASE(1, 1) CLO_TA_CASE(1, 2) CLO_TA_CASE(1, 3) CLO_TA_CASE(1, 4)
  CLO_TA_CASE(2, 1) CLO_TA_CASE(2, 2) CLO_TA_CASE(2, 3) CLO_TA_CASE(2, 4)
  CLO_TA_CASE(3, 1) CLO_TA_CASE(3, 2) CLO_TA_CASE(3, 3) CLO_TA_CASE(3, 4)
  CLO_TA_CASE(4, 1) CLO_TA_CASE(4, 2) CLO_TA_CASE(4, 3) CLO_TA_CASE(4, 4)
#undef CLO_TA_CASE
  CLO_CHECK_LAUNCH("tall_apply_kernel");
  return CLO_OK;
}
