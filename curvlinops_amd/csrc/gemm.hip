// fp32 GEMM / SYRK on the gfx950 f32 MFMA pipe (v_mfma_f32_32x32x2_f32).
//
// Three kernels share GemmArgs (gemm.h):
//   gemm_v2_kernel    16-byte aligned operands (the hot path): operand layouts are template
//                     parameters; k-contiguous operands keep their memory order in LDS and one
//                     ds_read_b128 feeds four MFMAs; tiles 128x128 (4 or 8 waves), 64x256, 32x256;
//                     fused epilogues, second K segment, implicit ones column, symmetric output
//   gemm_fwd3_kernel  fused forward + JVP of the large-batch MLP path (three products per tile)
//   gemm_f32_kernel   v1: runtime layout modes and scalar loads for unaligned operands; also the
//                     squared-accumulate variant of the EKFAC eigenvalue correction
// All stage both operand tiles in double-buffered LDS through registers one tile ahead of the
// MFMAs (one barrier per k-tile), use an XCD-aware grouped raster, and split K into deterministic
// slabs chosen by a small cost model (suggest_splitk_tiles).
//
// Reference call sites this replaces: kronecker.py:141-171 (einsum 'abZ,Aa,Bb->ABZ'),
// eigh.py:84-105, computers/kfac_hooks.py:350,390 (einsum "b s i, b s j -> i j"),
// computers/ekfac_hooks.py:206-236.
#include <type_traits>

#include "clo_common.h"
#include "gemm.h"

namespace clo {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_STRIDE = BM + 4;  // floats; 16-byte multiple, breaks power-of-two rows

enum LoadMode { MODE_OC_VEC = 0, MODE_KC_VEC = 1, MODE_OC_SCALAR = 2, MODE_KC_SCALAR = 3 };



// Load one [BK x 128] operand tile into 8 registers per thread.
// Element (o, k) lives at P[o*so + k*sk]; o in [o0, o0+128), k in [k0, k0+16).
// `ones` (outer-contiguous modes only): outer index O-1 is an implicit column of ones.
__device__ __forceinline__ void tile_load(float (&r)[8], int mode, const float *__restrict__ P,
                                          long so, long sk, int o0, int k0, int O, int Kend,
                                          int tid, int ones) {
  const int Oreal = O - ones;  // entries that exist in memory
  if (mode == MODE_OC_VEC) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = tid + 256 * q;
      const int k = k0 + (f >> 5);
      const int o = o0 + ((f & 31) << 2);
      const float *p = P + (long)k * sk + o;
      if (k < Kend && o + 3 < Oreal) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        r[4 * q + 0] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          r[4 * q + e] = (k < Kend && o + e < Oreal) ? p[e]
                         : ((ones && k < Kend && o + e == Oreal) ? 1.f : 0.f);
      }
    }
  } else if (mode == MODE_KC_VEC) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = tid + 256 * q;
      const int o = o0 + (f >> 2);
      const int k = k0 + ((f & 3) << 2);
      const float *p = P + (long)o * so + k;
      if (o < O && k + 3 < Kend) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        r[4 * q + 0] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[4 * q + e] = (o < O && k + e < Kend) ? p[e] : 0.f;
      }
    }
  } else if (mode == MODE_OC_SCALAR) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q;
      const int o = o0 + (e & 127), k = k0 + (e >> 7);
      r[q] = (o < Oreal && k < Kend) ? P[(long)o * so + (long)k * sk]
             : ((ones && k < Kend && o == Oreal) ? 1.f : 0.f);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q;
      const int k = k0 + (e & 15), o = o0 + (e >> 4);
      r[q] = (o < O && k < Kend) ? P[(long)o * so + (long)k * sk] : 0.f;
    }
  }
}

// Write the staged registers into the k-major LDS tile S[BK][LDS_STRIDE].
__device__ __forceinline__ void tile_store(const float (&r)[8], int mode, float *S, int tid) {
  if (mode == MODE_OC_VEC) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = tid + 256 * q;
      float4 v = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
      *reinterpret_cast<float4 *>(&S[(f >> 5) * LDS_STRIDE + ((f & 31) << 2)]) = v;
    }
  } else if (mode == MODE_KC_VEC) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = tid + 256 * q;
      const int o = f >> 2, k = (f & 3) << 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) S[(k + e) * LDS_STRIDE + o] = r[4 * q + e];
    }
  } else if (mode == MODE_OC_SCALAR) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q;
      S[(e >> 7) * LDS_STRIDE + (e & 127)] = r[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q;
      S[(e & 15) * LDS_STRIDE + (e >> 4)] = r[q];
    }
  }
}

template <bool SQSUM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BK * LDS_STRIDE];
  float *As = lds;                        // [2][BK][LDS_STRIDE]
  float *Bs = lds + 2 * BK * LDS_STRIDE;  // [2][BK][LDS_STRIDE]

  // ---- block -> tile mapping: XCD-aware (block b runs on XCD b % 8; give every XCD a
  // contiguous run of logical tiles so neighbours share operand panels in one L2), then a
  // grouped raster (8 tile-rows per group) over the tile grid.
  const int ntiles = p.tiles_m * p.tiles_n;
  int bm, bn;
  if (p.sym) {
    // symmetric output: the grid holds ONLY the upper-triangular tiles, column by column
    // (t = bn (bn + 1) / 2 + bm).  Consecutive blocks land on different XCDs, so every XCD gets the
    // same share of every column -- with the rectangular raster below (and an early exit for the
    // lower tiles) the XCDs that own the last tile rows had almost nothing to do and the symmetry
    // bought no time at all (24 x 3072^2, L^-T L^-1: 11.0 ms with and without it).
    const int t = blockIdx.x;
    bn = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while ((bn + 1) * (bn + 2) / 2 <= t) ++bn;
    while (bn * (bn + 1) / 2 > t) --bn;
    bm = t - bn * (bn + 1) / 2;
  } else {
    int lid;
    if (p.tri) {
      // triangular operand: the work of a tile depends on its row / column; contiguous tile ranges
      // per XCD would give some XCDs only the short tiles, so neighbours go to different XCDs
      lid = blockIdx.x;
    } else {
      const int b = blockIdx.x;
      const int q = ntiles / kNumXCD, rem = ntiles % kNumXCD;
      const int xcd = b % kNumXCD, idx = b / kNumXCD;
      lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    constexpr int GROUP = 8;
    const int per_group = GROUP * p.tiles_n;
    const int g = lid / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(p.tiles_m - first_m, GROUP);
    const int in_g = lid % per_group;
    bm = first_m + in_g % gsz;
    bn = in_g / gsz;
  }

  // plain GEMM: grid.y = batch * splitk, each block one (batch, k-range).
  // SQSUM:     grid.y = splitk, each block sums (A_b B_b)^2 over its range of batches.
  const int z = blockIdx.y;
  const int batch = SQSUM ? 0 : z / p.splitk, split = SQSUM ? z : z % p.splitk;
  const int kb = SQSUM ? 0 : split * p.k_per_split;
  const int ke = SQSUM ? p.K : min(p.K, kb + p.k_per_split);
  const int b_begin = SQSUM ? split * p.batch_per_split : batch;
  const int b_end = SQSUM ? min(p.nbatch, b_begin + p.batch_per_split) : batch + 1;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = bm * BM, n0 = bn * BN;

  f32x16 acc[2][2], sq[SQSUM ? 2 : 1][SQSUM ? 2 : 1];
  if (SQSUM) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sq[SQSUM ? i : 0][SQSUM ? j : 0][r] = 0.f;
  }
  const int nk = (ke - kb + BK - 1) / BK;
  float ra[8], rb[8];

  for (int bcur = b_begin; bcur < b_end; ++bcur) {
  const float *A = p.A + gemm_off_a(p, bcur);
  const float *B = p.B + gemm_off_b(p, bcur);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (nk > 0) {
    tile_load(ra, p.mode_a, A, p.sa_m, p.sa_k, m0, kb, p.M, ke, tid, p.ones);
    tile_load(rb, p.mode_b, B, p.sb_n, p.sb_k, n0, kb, p.N, ke, tid, p.ones);
    tile_store(ra, p.mode_a, As, tid);
    tile_store(rb, p.mode_b, Bs, tid);
  }
  __syncthreads();

  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    const bool more = it + 1 < nk;
    if (more) {
      const int k0 = kb + (it + 1) * BK;
      tile_load(ra, p.mode_a, A, p.sa_m, p.sa_k, m0, k0, p.M, ke, tid, p.ones);
      tile_load(rb, p.mode_b, B, p.sb_n, p.sb_k, n0, k0, p.N, ke, tid, p.ones);
    }
    const float *as = As + cur * BK * LDS_STRIDE + wm * 64 + li;
    const float *bs = Bs + cur * BK * LDS_STRIDE + wn * 64 + li;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = as[(kk + lh) * LDS_STRIDE];
      const float a1 = as[(kk + lh) * LDS_STRIDE + 32];
      const float b0 = bs[(kk + lh) * LDS_STRIDE];
      const float b1 = bs[(kk + lh) * LDS_STRIDE + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      tile_store(ra, p.mode_a, As + (cur ^ 1) * BK * LDS_STRIDE, tid);
      tile_store(rb, p.mode_b, Bs + (cur ^ 1) * BK * LDS_STRIDE, tid);
    }
    __syncthreads();
  }
  if (SQSUM) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sq[SQSUM ? i : 0][SQSUM ? j : 0][r] += acc[i][j][r] * acc[i][j][r];
  }
  }  // batch loop
  if (SQSUM) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = sq[SQSUM ? i : 0][SQSUM ? j : 0];
  }

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool to_ws = p.splitk > 1;
  float *C = to_ws ? p.ws + (long)z * p.M * p.N : p.C + (long)batch * p.sc_b;  // SQSUM: batch == 0
  const long ldc = to_ws ? p.N : p.ldc;
  const float alpha = to_ws ? 1.f : p.alpha;
  const float beta = to_ws ? 0.f : p.beta;
  const bool mirror = p.sym && !to_ws && bm != bn;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + wn * 64 + nt * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M && col < p.N) {
          float v = alpha * acc[mt][nt][r];
          float *c = C + (long)row * ldc + col;
          if (!to_ws && p.epi != EPI_NONE) {
            store_final(p, c, row, col, acc[mt][nt][r], alpha, beta);
            continue;
          }
          if (beta != 0.f) v += beta * *c;
          *c = v;
          if (mirror) {
            float *ct = C + (long)col * ldc + row;
            float vt = alpha * acc[mt][nt][r];
            if (beta != 0.f) vt += beta * *ct;
            *ct = vt;
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------
// v2 engine for 16-byte-aligned operands (the common case): operand layouts are template
// parameters, so the loop body has no mode branches, and a k-contiguous operand keeps its
// memory order in LDS ([outer][BK + 4]) where ONE ds_read_b128 feeds four MFMAs: the k index a
// lane supplies is a free permutation as long as A and B agree, so within an 8-k group MFMA
// m (0..3) takes k = 4 * (lane >> 5) + m on both sides.  Outer-contiguous operands stay k-major
// ([BK][128 + 4]) and fetch that same k with ds_read_b32.  Loads are unconditional (clamped
// addresses, results zeroed by select) so hipcc keeps them in flight across the MFMA block.
// ------------------------------------------------------------------------------------------
template <bool KC, int BKT, int NTHR, int T>
struct TileIO {
  static constexpr int LD = KC ? BKT + 4 : T + 4;
  static constexpr int FLOATS = KC ? T * LD : BKT * LD;
  static constexpr int F4 = T * BKT / 4;                    // float4 per tile
  static constexpr int NF4 = (F4 + NTHR - 1) / NTHR;        // float4 per thread
  // a tile with fewer float4 than threads: the surplus threads repeat the work of thread f % F4
  // (same load, same LDS store) instead of branching around the loads
  const float *base[NF4];
  int koff[NF4];  // k offset of the float4 inside the tile
  int soff[NF4];  // LDS offset (floats)
  unsigned one_bits;  // OC + implicit ones column: float4 q starts at that column

  // element (o, k) at P[o * so + k * sk]; KC: sk == 1, OC: so == 1.  For OC the extent in memory
  // (O - ones) is a multiple of 4; outer index O - 1 is the implicit ones column when ones == 1.
  __device__ __forceinline__ void init(const float *P, long so, long sk, int o0, int O, int tid,
                                       int ones) {
    one_bits = 0u;
#pragma unroll
    for (int q = 0; q < NF4; ++q) {
      const int f = (tid + NTHR * q) % F4;
      if (KC) {
        const int o = f / (BKT / 4), kq = (f % (BKT / 4)) * 4;
        base[q] = P + (long)min(o0 + o, O - 1) * so;
        koff[q] = kq;
        soff[q] = o * LD + kq;
      } else {
        const int k = f / (T / 4), oq = (f % (T / 4)) * 4;
        const int Oreal = O - ones;
        base[q] = P + min(o0 + oq, Oreal - 4);
        if (ones && o0 + oq == Oreal) one_bits |= 1u << q;
        koff[q] = k;
        soff[q] = k * LD + oq;
      }
    }
  }
  // `delta` (floats, wave-uniform) moves the whole tile to the second K segment's operand
  __device__ __forceinline__ void load(float4 (&r)[NF4], int k0, int Kend, long sk,
                                       long delta) const {
#pragma unroll
    for (int q = 0; q < NF4; ++q) {
      const int k = k0 + koff[q];
      if (KC) {
        const bool ok = k + 3 < Kend;  // K % 4 == 0: all or nothing
        const float4 v = *reinterpret_cast<const float4 *>(base[q] + delta + (ok ? k : 0));
        r[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const bool ok = k < Kend;
        float4 v = *reinterpret_cast<const float4 *>(base[q] + delta + (long)(ok ? k : 0) * sk);
        if ((one_bits >> q) & 1u) v = make_float4(1.f, 0.f, 0.f, 0.f);
        r[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __device__ __forceinline__ void store(float *S, const float4 (&r)[NF4]) const {
#pragma unroll
    for (int q = 0; q < NF4; ++q) *reinterpret_cast<float4 *>(S + soff[q]) = r[q];
  }
  // operand values of this lane for the 8-k group g: v[m] feeds MFMA m
  static __device__ __forceinline__ float4 frag(const float *S, int outer, int g, int lh) {
    if (KC) return *reinterpret_cast<const float4 *>(S + outer * LD + g * 8 + 4 * lh);
    const float *p = S + (g * 8 + 4 * lh) * LD + outer;
    return make_float4(p[0], p[LD], p[2 * LD], p[3 * LD]);
  }
};

// Outer-contiguous operand whose elements are im2col patches generated on the fly (KFAC input
// covariance of a Conv2d layer, reference kfac_utils.py:78-121: unfold(x).transpose(1, 2)):
//   X[r][m],  r = (b, oh, ow),  m = (c, kh, kw)  ->  x[b][c][oh SH - PH + kh DH][ow SW - PW + kw DW]
// (zero outside the image; m == C KH KW is the implicit ones column of a joint weight + bias factor).
// Same LDS layout and fragment reads as TileIO<false, ...>; the four features of a float4 are four
// scalar loads (the input tensor is 1/(KH KW) of the patch matrix and lives in L2 / MALL).
template <int BKT, int NTHR, int T>
struct PatchIO {
  static constexpr int LD = T + 4;
  static constexpr int FLOATS = BKT * LD;
  static constexpr int F4 = T * BKT / 4;
  static constexpr int NF4 = (F4 + NTHR - 1) / NTHR;
  static constexpr int RSTEP = NTHR / (T / 4);  // rows between consecutive float4 of one thread
  static_assert(NTHR % (T / 4) == 0 && F4 % NTHR == 0, "a thread keeps its four features for the whole tile");
  const float *x;
  // the thread's four features (the same for all of its float4): offset inside the receptive field
  // ((c H + kh DH) W + kw DW, or -1: no such feature), (kh DH) << 16 | (kw DW), ones-column bits
  int off[4], dhw[4];
  unsigned one_bits;
  int krow, soff0;
  int H, W, OH, OW, SH, SW, PH, PW, CHW;
  // (b, oh, ow) of the thread's first row of the CURRENT tile, advanced tile by tile (no divisions in
  // the k loop); next_k0 = the tile start those coordinates belong to
  int cb, coh, cow, cur_k0;
  int d_ow, d_oh, d_b;  // RSTEP rows further in (b, oh, ow) coordinates (mixed radix, with carries)

  __device__ __forceinline__ void init_patch(const GemmArgs &p, int o0, int O, int tid) {
    x = p.A;
    H = p.cvH; W = p.cvW; OH = p.cvOH; OW = p.cvOW; SH = p.cvSH; SW = p.cvSW; PH = p.cvPH; PW = p.cvPW;
    CHW = p.cvC * p.cvH * p.cvW;
    const int Q = O - p.ones, KK = p.cvKH * p.cvKW;
    const int oq = (tid % (T / 4)) * 4;
    krow = tid / (T / 4);
    soff0 = krow * LD + oq;
    one_bits = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = o0 + oq + e;
      off[e] = -1;
      dhw[e] = 0;
      if (m < Q) {
        const int c = m / KK, r = m - c * KK, kh = r / p.cvKW, kw = r - kh * p.cvKW;
        off[e] = (c * H + kh * p.cvDH) * W + kw * p.cvDW;
        dhw[e] = ((kh * p.cvDH) << 16) | (kw * p.cvDW);
      } else if (p.ones && m == Q) {
        one_bits |= 1u << e;
      }
    }
    cur_k0 = -1;
    d_ow = RSTEP % OW;
    d_oh = (RSTEP / OW) % OH;
    d_b = (RSTEP / OW) / OH;
  }
  __device__ __forceinline__ void seek(int k0) {
    const int row = k0 + krow;
    cow = row % OW;
    const int t = row / OW;
    coh = t % OH;
    cb = t / OH;
    cur_k0 = k0;
  }
  __device__ __forceinline__ void load(float4 (&r)[NF4], int k0, int Kend, long, long) {
    if (k0 != cur_k0) seek(k0);  // first tile of this block (or a jump): the only divisions
    int b = cb, oh = coh, ow = cow;
#pragma unroll
    for (int q = 0; q < NF4; ++q) {
      const bool ok = k0 + krow + q * RSTEP < Kend;
      const int ih0 = oh * SH - PH, iw0 = ow * SW - PW;
      const long base = (long)b * CHW + (long)ih0 * W + iw0;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ih = ih0 + (dhw[e] >> 16), iw = iw0 + (dhw[e] & 0xffff);
        const bool in = ok && off[e] >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        const float val = x[in ? base + off[e] : 0];
        v[e] = in ? val : (((one_bits >> e) & 1u) && ok ? 1.f : 0.f);
      }
      r[q] = make_float4(v[0], v[1], v[2], v[3]);
      // next float4 of this thread: RSTEP rows further
      ow += d_ow;
      const int c1 = ow >= OW ? 1 : 0;
      ow -= c1 ? OW : 0;
      oh += d_oh + c1;
      const int c2 = oh >= OH ? 1 : 0;
      oh -= c2 ? OH : 0;
      b += d_b + c2;
    }
    // after NF4 steps the coordinates are those of row k0 + BKT + krow = this thread's first row of the
    // next tile (NF4 * RSTEP == BKT)
    cb = b; coh = oh; cow = ow;
    cur_k0 = k0 + BKT;
  }
  __device__ __forceinline__ void store(float *S, const float4 (&r)[NF4]) const {
#pragma unroll
    for (int q = 0; q < NF4; ++q) *reinterpret_cast<float4 *>(S + soff0 + q * RSTEP * LD) = r[q];
  }
  static __device__ __forceinline__ float4 frag(const float *S, int outer, int g, int lh) {
    const float *p = S + (g * 8 + 4 * lh) * LD + outer;
    return make_float4(p[0], p[LD], p[2 * LD], p[3 * LD]);
  }
};

// Block tile BMt x BNt, WVM x WVN waves, each wave (BMt / WVM) x (BNt / WVN) in 32x32 MFMA tiles:
//   128 x 128, 2 x 2 waves of 64x64 : large grids
//   128 x 128, 2 x 4 waves of 64x32 : two waves per SIMD inside ONE block, for grids that cannot put
//                                     two blocks on every CU
//    64 x 256, 1 x 8 waves of 64x32 ; 32 x 256, 1 x 8 waves of 32x32 : few rows (M <= 64 / 32): no
//                                     MFMA work on padding rows, the B operand streams once
// SQ (clo_gemm_sqsum_f32): grid.y = batch splits; the block walks its members b, forms the tile of A_b B_b over the whole K
// and adds its elementwise SQUARE to a second accumulator set, which is what the epilogue stores.
template <bool AKC, bool BKC, int BKT, int BMt, int BNt, int WVM, int WVN, bool PATCH = false, bool SQ = false>
__global__ __launch_bounds__(WVM * WVN * 64, 2) void gemm_v2_kernel(const GemmArgs p) {
  constexpr int NW = WVM * WVN;
  constexpr int WM = BMt / WVM, WNC = BNt / WVN;  // rows / columns per wave
  constexpr int MT = WM / 32, NT = WNC / 32;      // MFMA tiles per wave
  using TA = std::conditional_t<PATCH, PatchIO<BKT, NW * 64, BMt>, TileIO<AKC, BKT, NW * 64, BMt>>;
  using TB = std::conditional_t<PATCH, PatchIO<BKT, NW * 64, BNt>, TileIO<BKC, BKT, NW * 64, BNt>>;
  extern __shared__ __attribute__((aligned(16))) float lds2[];
  float *As = lds2;                    // [2][TA::FLOATS]
  float *Bs = lds2 + 2 * TA::FLOATS;   // [2][TB::FLOATS]

  const int ntiles = p.tiles_m * p.tiles_n;
  int bm, bn;
  if (p.sym) {
    // symmetric output: the grid holds ONLY the upper-triangular tiles, column by column
    // (t = bn (bn + 1) / 2 + bm).  Consecutive blocks land on different XCDs, so every XCD gets the
    // same share of every column -- with the rectangular raster below (and an early exit for the
    // lower tiles) the XCDs that own the last tile rows had almost nothing to do and the symmetry
    // bought no time at all (24 x 3072^2, L^-T L^-1: 11.0 ms with and without it).
    const int t = blockIdx.x;
    bn = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while ((bn + 1) * (bn + 2) / 2 <= t) ++bn;
    while (bn * (bn + 1) / 2 > t) --bn;
    bm = t - bn * (bn + 1) / 2;
  } else {
    int lid;
    if (p.tri) {
      // triangular operand: the work of a tile depends on its row / column; contiguous tile ranges
      // per XCD would give some XCDs only the short tiles, so neighbours go to different XCDs
      lid = blockIdx.x;
    } else {
      const int b = blockIdx.x;
      const int q = ntiles / kNumXCD, rem = ntiles % kNumXCD;
      const int xcd = b % kNumXCD, idx = b / kNumXCD;
      lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    constexpr int GROUP = 8;
    const int per_group = GROUP * p.tiles_n;
    const int g = lid / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(p.tiles_m - first_m, GROUP);
    const int in_g = lid % per_group;
    bm = first_m + in_g % gsz;
    bn = in_g / gsz;
  }

  const int z = blockIdx.y;
  const int batch = SQ ? 0 : z / p.splitk, split = SQ ? 0 : z % p.splitk;
  const int m0 = bm * BMt, n0 = bn * BNt;
  int kb = SQ ? 0 : split * p.k_per_split;
  int ke = SQ ? p.K : min(p.K, kb + p.k_per_split);
  const int b_begin = SQ ? z * p.batch_per_split : batch, b_end = SQ ? min(p.nbatch, b_begin + p.batch_per_split) : batch + 1;
  if (p.tri) {  // triangular operands: visit only the k range of this tile that can be nonzero
    int lo = 0, hi = p.K;
    if (p.tri & TRI_KGE_M) lo = max(lo, m0);
    if (p.tri & TRI_KGE_N) lo = max(lo, n0);
    if (p.tri & TRI_KLT_M) hi = min(hi, m0 + BMt);
    if (p.tri & TRI_KLT_N) hi = min(hi, n0 + BNt);
    kb = max(kb, lo & ~(BKT - 1));
    ke = min(ke, hi);
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WVN, wn = wave % WVN;
  const int li = lane & 31, lh = lane >> 5;

  TA la;
  TB lb;
  f32x16 acc[MT][NT], sq[SQ ? MT : 1][SQ ? NT : 1];
  if (SQ) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sq[SQ ? i : 0][SQ ? j : 0][r] = 0.f;
  }
  for (int bcur = b_begin; bcur < b_end; ++bcur) {   // (one trip unless SQ)
  if constexpr (PATCH) {
    la.init_patch(p, m0, p.M, tid);
    lb.init_patch(p, n0, p.N, tid);
  } else {
    // (SQ: the B operand's rows may be padded to n_mem >= N floats, a multiple of 4, which the float4 loader may read)
    la.init(p.A + gemm_off_a(p, bcur), p.sa_m, p.sa_k, m0, p.M, tid, p.ones);
    lb.init(p.B + gemm_off_b(p, bcur), p.sb_n, p.sb_k, n0, SQ && p.n_mem ? p.n_mem : p.N, tid, p.ones | p.ones_b);
  }

#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = ke > kb ? (ke - kb + BKT - 1) / BKT : 0;
  float4 ra[TA::NF4], rb[TB::NF4];
  // K segments: tile k0 lies in segment 2 iff k0 >= K1 (K1 is a multiple of the tile depth)
  const int K1 = p.A2 ? p.K1 : p.K;
  const long dA2 = p.A2 ? (p.A2 - p.A) : 0, dB2 = p.B2 ? (p.B2 - p.B) : 0;
  if (nk > 0) {
    const bool s2 = kb >= K1;
    const int kr = s2 ? kb - K1 : kb, kend = s2 ? ke - K1 : min(ke, K1);
    la.load(ra, kr, kend, p.sa_k, s2 ? dA2 : 0);
    lb.load(rb, kr, kend, p.sb_k, s2 ? dB2 : 0);
    la.store(As, ra);
    lb.store(Bs, rb);
  }
  __syncthreads();

  // Where the next tile's global loads are issued.  One 8-wave block per CU: behind the first
  // fragment reads (right after the barrier ALL waves of the CU wait for LDS data; the global loads
  // have the whole MFMA block to land): 2048^3 104 -> 111 TFLOP/s.  Two 4-wave blocks per CU cover
  // each other's barriers, there the earlier issue is the better one (8192^3: 132 vs 128).
  constexpr bool LATE_PREFETCH = NW == 8;
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    const int k0 = kb + (it + 1) * BKT;
    if (!LATE_PREFETCH) {
      const bool s2 = k0 >= K1;
      const int kr = s2 ? k0 - K1 : k0, kend = s2 ? ke - K1 : min(ke, K1);
      la.load(ra, kr, kend, p.sa_k, s2 ? dA2 : 0);
      lb.load(rb, kr, kend, p.sb_k, s2 ? dB2 : 0);
    }
    const float *as = As + cur * TA::FLOATS;
    const float *bs = Bs + cur * TB::FLOATS;
#pragma unroll
    for (int g = 0; g < BKT / 8; ++g) {
      float4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = TA::frag(as, wm * WM + i * 32 + li, g, lh);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = TB::frag(bs, wn * WNC + j * 32 + li, g, lh);
      if (LATE_PREFETCH && g == 0) {
        // prefetch the next tile (past the end: clamped addresses, zeroed values, never stored)
        __builtin_amdgcn_sched_barrier(0);
        const bool s2 = k0 >= K1;
        const int kr = s2 ? k0 - K1 : k0, kend = s2 ? ke - K1 : min(ke, K1);
        la.load(ra, kr, kend, p.sa_k, s2 ? dA2 : 0);
        lb.load(rb, kr, kend, p.sb_k, s2 ? dB2 : 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#define CLO_MM(E)                                                                               \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].E, bf[j].E, acc[i][j], 0, 0, 0);
      CLO_MM(x) CLO_MM(y) CLO_MM(z) CLO_MM(w)
#undef CLO_MM
    }
    if (it + 1 < nk) {
      la.store(As + (cur ^ 1) * TA::FLOATS, ra);
      lb.store(Bs + (cur ^ 1) * TB::FLOATS, rb);
    }
    __syncthreads();
  }
  if (SQ) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sq[SQ ? i : 0][SQ ? j : 0][r] += acc[i][j][r] * acc[i][j][r];
  }
  }  // members
  if (SQ) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = sq[SQ ? i : 0][SQ ? j : 0];
  }

  const bool to_ws = p.splitk > 1;
  float *C = to_ws ? p.ws + (long)z * p.M * p.N : p.C + (long)batch * p.sc_b;
  const long ldc = to_ws ? p.N : p.ldc;
  const float alpha = to_ws ? 1.f : p.alpha;
  const float beta = to_ws ? 0.f : p.beta;
  const bool mirror = p.sym && !to_ws && bm != bn;
#ifndef CLO_V2_WIDE_EPI
#define CLO_V2_WIDE_EPI 1
#endif
  // Wide form (round 6, as in gemm_v3.hip): the tile is staged in LDS (the loop's buffers are dead behind its last barrier; the
  // launch sizes the allocation for the larger of the two uses) and leaves as full-row 16-byte stores, its mirror image
  // likewise; the separate destination of the last column (col_out) is served from the staged tile.
  V3Epi E;
  E.kind = to_ws ? EPI_NONE : p.epi;
  E.act = p.e_act; E.div = p.e_div; E.vec = p.e_vec; E.mul = p.e_mul; E.ld_mul = p.ld_mul; E.out2 = p.e_out2; E.Cbase = p.C;
  const bool has_col_out = !to_ws && p.col_out;
  const bool wide = CLO_V2_WIDE_EPI && (ldc & 3) == 0 && ((unsigned long)C & 15ul) == 0 &&
                    (E.kind == EPI_NONE || (!mirror && !has_col_out && (E.kind != EPI_ACT || ((unsigned long)E.out2 & 15ul) == 0)));
  if (wide) {
    constexpr int NTHR = NW * 64, PD = BNt + 4, PM = BMt + 4;
    float *T = lds2;
    stage_tile_direct<MT, NT, WM, WNC, PD>(T, acc, wm, wn, li, lh);
    __syncthreads();
    v3_store_rows<BMt, BNt, NTHR>(T, PD, C, ldc, m0, n0, p.M, has_col_out ? p.N - 1 : p.N, alpha, beta, tid, E);
    if (has_col_out && n0 <= p.N - 1 && p.N - 1 < n0 + BNt) {
      for (int i = tid; i < BMt; i += NTHR)
        if (m0 + i < p.M) {
          float *c = p.col_out + m0 + i;
          float v = alpha * T[i * PD + (p.N - 1 - n0)];
          if (beta != 0.f) v += beta * *c;
          *c = v;
        }
    }
    if (mirror) {
      __syncthreads();
      stage_tile_mirror<MT, NT, WM, WNC, PM>(T, acc, wm, wn, li, lh);
      __syncthreads();
      v3_store_rows<BNt, BMt, NTHR>(T, PM, C, ldc, n0, m0, p.N, p.M, alpha, beta, tid, E);
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + wn * WNC + nt * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M && col < p.N) {
          float v = alpha * acc[mt][nt][r];
          float *c = (!to_ws && p.col_out && col == p.N - 1) ? p.col_out + row : C + (long)row * ldc + col;
          if (!to_ws && p.epi != EPI_NONE) {
            store_final(p, c, row, col, acc[mt][nt][r], alpha, beta);
            continue;
          }
          if (beta != 0.f) v += beta * *c;
          *c = v;
          if (mirror) {
            float *ct = C + (long)col * ldc + row;
            float vt = alpha * acc[mt][nt][r];
            if (beta != 0.f) vt += beta * *ct;
            *ct = vt;
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------
// Fused forward + JVP tile of the large-batch MLP path: ONE pass over W and V per layer,
//   Z  = A W^T            -> a  = act(Z + b),  phi' = act'(Z + b)
//   dZ = A V^T + dA W^T   -> da = phi' * (dZ + Vb)
// All four operands are k-contiguous (activations [N][d_in], weights [d_out][d_in]); three MFMA
// products per fragment pair, two accumulator sets.  Split-K writes both sets to slabs
// ws[split][2][N][d_out]; fwd3_reduce_kernel then applies the epilogue.
// ------------------------------------------------------------------------------------------
struct Fwd3Args {
  const float *A, *dA;   // [N][d_in], dA may be null (first layer)
  const float *W, *V;    // [d_out][d_in]
  const float *b, *Vb;   // [d_out] or null
  float *a, *da, *dphi;  // [N][d_out]
  float *ws;             // split-K slabs
  int N, d_in, d_out, act;
  int splitk, k_per_split, tiles_m, tiles_n;
};

__device__ __forceinline__ void fwd3_finish(const Fwd3Args &p, int row, int col, float z, float dz) {
  float dphi;
  const long e = (long)row * p.d_out + col;
  p.a[e] = act_apply(p.act, z + (p.b ? p.b[col] : 0.f), dphi);
  p.dphi[e] = dphi;
  p.da[e] = dphi * (dz + (p.Vb ? p.Vb[col] : 0.f));
}

template <int BKT, int BMt, int BNt, int WVM, int WVN, bool HAS_DA>
__global__ __launch_bounds__(WVM * WVN * 64, 2) void gemm_fwd3_kernel(const Fwd3Args p) {
  constexpr int NW = WVM * WVN;
  constexpr int WM = BMt / WVM, WNC = BNt / WVN;
  constexpr int MT = WM / 32, NT = WNC / 32;
  using TA = TileIO<true, BKT, NW * 64, BMt>;
  using TB = TileIO<true, BKT, NW * 64, BNt>;
  extern __shared__ __attribute__((aligned(16))) float lds3[];
  constexpr int STAGE = (HAS_DA ? 2 : 1) * TA::FLOATS + 2 * TB::FLOATS;
  // stage layout: [A][dA][W][V]
  const int tile = blockIdx.x;
  const int bm = tile % p.tiles_m, bn = tile / p.tiles_m;
  const int split = blockIdx.y;
  const int kb = split * p.k_per_split, ke = min(p.d_in, kb + p.k_per_split);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WVN, wn = wave % WVN;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = bm * BMt, n0 = bn * BNt;

  TA la;
  TB lb;
  la.init(p.A, p.d_in, 1, m0, p.N, tid, 0);
  lb.init(p.W, p.d_in, 1, n0, p.d_out, tid, 0);
  const long dDA = HAS_DA ? (p.dA - p.A) : 0, dV = p.V - p.W;

  f32x16 accz[MT][NT], accd[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accz[i][j][r] = accd[i][j][r] = 0.f;

  const int nk = (ke - kb + BKT - 1) / BKT;
  float4 ra[TA::NF4], rda[HAS_DA ? TA::NF4 : 1], rw[TB::NF4], rv[TB::NF4];
  auto load_all = [&](int k0) {
    la.load(ra, k0, ke, 1, 0);
    if constexpr (HAS_DA) la.load(rda, k0, ke, 1, dDA);
    lb.load(rw, k0, ke, 1, 0);
    lb.load(rv, k0, ke, 1, dV);
  };
  auto store_all = [&](float *S) {
    la.store(S, ra);
    if constexpr (HAS_DA) la.store(S + TA::FLOATS, rda);
    lb.store(S + (HAS_DA ? 2 : 1) * TA::FLOATS, rw);
    lb.store(S + (HAS_DA ? 2 : 1) * TA::FLOATS + TB::FLOATS, rv);
  };
  if (nk > 0) {
    load_all(kb);
    store_all(lds3);
  }
  __syncthreads();
  // as in gemm_v2_kernel: the next tile's global loads go behind the first fragment reads
  constexpr bool kLate = BMt < 128;  // (128-row tiles, many blocks: measured worse, 516 -> 526 us at N = 512)
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    if (!kLate) load_all(kb + (it + 1) * BKT);
    const float *S = lds3 + cur * STAGE;
    const float *as = S, *das = S + TA::FLOATS;
    const float *wsm = S + (HAS_DA ? 2 : 1) * TA::FLOATS, *vsm = wsm + TB::FLOATS;
#pragma unroll
    for (int g = 0; g < BKT / 8; ++g) {
      float4 af[MT], daf[MT], wf[NT], vf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        af[i] = TA::frag(as, wm * WM + i * 32 + li, g, lh);
        if (HAS_DA) daf[i] = TA::frag(das, wm * WM + i * 32 + li, g, lh);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        wf[j] = TB::frag(wsm, wn * WNC + j * 32 + li, g, lh);
        vf[j] = TB::frag(vsm, wn * WNC + j * 32 + li, g, lh);
      }
      if (kLate && g == 0) {
        __builtin_amdgcn_sched_barrier(0);
        load_all(kb + (it + 1) * BKT);
        __builtin_amdgcn_sched_barrier(0);
      }
#define CLO_M3(E)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j) {  \
    accz[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].E, wf[j].E, accz[i][j], 0, 0, 0);      \
    accd[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].E, vf[j].E, accd[i][j], 0, 0, 0);      \
    if (HAS_DA)                                                                                    \
      accd[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(daf[i].E, wf[j].E, accd[i][j], 0, 0, 0);   \
  }
      CLO_M3(x) CLO_M3(y) CLO_M3(z) CLO_M3(w)
#undef CLO_M3
    }
    if (it + 1 < nk) store_all(lds3 + (cur ^ 1) * STAGE);
    __syncthreads();
  }
  const bool to_ws = p.splitk > 1;
  const long MN = (long)p.N * p.d_out;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + wn * WNC + nt * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.N && col < p.d_out) {
          if (to_ws) {
            float *w = p.ws + (long)split * 2 * MN + (long)row * p.d_out + col;
            w[0] = accz[mt][nt][r];
            w[MN] = accd[mt][nt][r];
          } else {
            fwd3_finish(p, row, col, accz[mt][nt][r], accd[mt][nt][r]);
          }
        }
      }
    }
}

__global__ void fwd3_reduce_kernel(const Fwd3Args p) {
  const long MN = (long)p.N * p.d_out;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < MN;
       e += (long)gridDim.x * blockDim.x) {
    const float *w = p.ws + e;
    float z0 = 0.f, z1 = 0.f, d0 = 0.f, d1 = 0.f;
    int s = 0;
    for (; s + 1 < p.splitk; s += 2) {
      z0 += w[(2L * s) * MN]; d0 += w[(2L * s + 1) * MN];
      z1 += w[(2L * s + 2) * MN]; d1 += w[(2L * s + 3) * MN];
    }
    for (; s < p.splitk; ++s) { z0 += w[(2L * s) * MN]; d0 += w[(2L * s + 1) * MN]; }
    fwd3_finish(p, (int)(e / p.d_out), (int)(e % p.d_out), z0 + z1, d0 + d1);
  }
}

// Tiny products with operands the aligned engine cannot take (e.g. the 10-class head of a small
// network: K = 10 or M = 10): one thread per output element, the k loop straight from global memory
// (the operands are a few KB and stay in L2).  One short launch instead of the v1 tile kernel's
// latency chain (~15 us for a single 128x128 tile).
__global__ __launch_bounds__(256) void gemm_tiny_kernel(const GemmArgs p) {
  const long total = (long)p.M * p.N;
  const int b = blockIdx.y;
  const float *A = p.A + gemm_off_a(p, b), *B = p.B + gemm_off_b(p, b);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int m = e / p.N, n = e % p.N;
    if (p.sym && n < m) continue;
    const float *a = A + (long)m * p.sa_m, *bb = B + (long)n * p.sb_n;
    const bool one_a = p.ones && m == p.M - 1, one_b = p.ones && n == p.N - 1;
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 1 < p.K; k += 2) {
      s0 = fmaf(one_a ? 1.f : a[(long)k * p.sa_k], one_b ? 1.f : bb[(long)k * p.sb_k], s0);
      s1 = fmaf(one_a ? 1.f : a[(long)(k + 1) * p.sa_k], one_b ? 1.f : bb[(long)(k + 1) * p.sb_k], s1);
    }
    if (k < p.K) s0 = fmaf(one_a ? 1.f : a[(long)k * p.sa_k], one_b ? 1.f : bb[(long)k * p.sb_k], s0);
    const float acc = s0 + s1;
    float *c = p.C + (long)b * p.sc_b + (long)m * p.ldc + n;
    store_final(p, c, m, n, acc, p.alpha, p.beta);
    if (p.sym && n != m) store_final(p, p.C + (long)b * p.sc_b + (long)n * p.ldc + m, n, m, acc, p.alpha, p.beta);
  }
}

// C = alpha * sum_s ws[b][s] + beta * C ; for sym the lower block-triangle of ws was never
// written, take the transposed element instead.
__global__ void splitk_reduce_kernel(const GemmArgs p, int splitk) {
  const long total = (long)p.M * p.N;
  const int b = blockIdx.y;
  const int N = p.N;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int m = e / N, n = e % N;
    long src = e;
    if (p.sym && (n / p.tbn) < (m / p.tbm)) src = (long)n * N + m;
    const float *w = p.ws + (long)b * splitk * total + src;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < splitk; k += 4) {
      s0 += w[(k + 0) * total]; s1 += w[(k + 1) * total];
      s2 += w[(k + 2) * total]; s3 += w[(k + 3) * total];
    }
    for (; k < splitk; ++k) s0 += w[k * total];
    float *c = (p.col_out && n == N - 1) ? p.col_out + m : p.C + (long)b * p.sc_b + (long)m * p.ldc + n;
    store_final(p, c, m, n, (s0 + s1) + (s2 + s3), p.alpha, p.beta);
  }
}

static int pick_mode(const float *P, long so, long sk, long sbatch, int batch) {
  const bool batch_ok = batch <= 1 || (sbatch % 4 == 0);
  if (so == 1) return (sk % 4 == 0 && aligned16(P) && batch_ok) ? MODE_OC_VEC : MODE_OC_SCALAR;
  if (sk == 1) return (so % 4 == 0 && aligned16(P) && batch_ok) ? MODE_KC_VEC : MODE_KC_SCALAR;
  return so <= sk ? MODE_OC_SCALAR : MODE_KC_SCALAR;
}

// Tile configuration of the v2 engine: few-row problems (the batch dimension of the mid-size MLP
// path) get 32- / 64-row tiles so that no MFMA work is spent on padding rows.
//  64 x  64, 2 x 2 waves, k tiles of 64: tiny problems (a handful of tiles, short K), where one
//                                     block's k loop is a chain of memory round trips (~1.3 us
//                                     each whatever the tile depth): smaller tiles on more CUs,
//                                     4x deeper k tiles (Cholesky recursion, small-network layers)
struct V2Config { int bm, bn, bk; };
static V2Config v2_config(int M, int N, int K, long batch, int sym, bool allow_small = true) {
  // (few-row problems keep their 32- / 64-row tiles unless they are tiny)
#ifndef CLO_GEMM_SMALL_MAX
#define CLO_GEMM_SMALL_MAX (1024L * 1024L)
#endif
  static const long small_max = CLO_GEMM_SMALL_MAX;
  // (a batch of skinny products -- the panels and column updates of the batched Cholesky chain, N <= 128 -- is judged by
  // ONE matrix: 3 x (4480 x 128) on 128 x 128 x 32 tiles is 105 workgroups with k loops of up to 32 steps, 50-110 us per
  // product on the critical path; on 64 x 64 x 64 tiles 8-15 us)
  const long area = (long)M * N * (N <= 128 ? 1 : batch);
  if (allow_small && ((area <= 256L * 256L && K <= 1024) || (area <= small_max && (M > 64 || sym))))
    return {64, 64, 64};
  if (!sym && M <= 32) return {32, 256, 16};
  if (!sym && M <= 64) return {64, 256, 16};
  return {128, 128, 32};
}

// v2 needs float4-complete operands: 16-byte aligned, K % 4 == 0 for k-contiguous operands, the
// outer extent in memory % 4 == 0 for outer-contiguous ones.
bool gemm_v2_eligible(const GemmArgs &a, int batch) {
#ifndef CLO_GEMM_V1
#define CLO_GEMM_V1 0
#endif
  static const int v2_off = CLO_GEMM_V1;
  const int ma = pick_mode(a.A, a.sa_m, a.sa_k, a.sa_b, batch);
  const int mb = pick_mode(a.B, a.sb_n, a.sb_k, a.sb_b, batch);
  const bool a_vec = ma == MODE_OC_VEC || ma == MODE_KC_VEC;
  const bool b_vec = mb == MODE_OC_VEC || mb == MODE_KC_VEC;
  const bool a_kc = ma == MODE_KC_VEC, b_kc = mb == MODE_KC_VEC;
  const int Kc = a.A2 ? a.K1 : a.K;  // every segment must be float4-complete
  if (a.A2 && (!aligned16(a.A2) || !aligned16(a.B2))) return false;
  if (a.ones && (a_kc || b_kc)) return false;  // the ones column lives in the outer-contiguous loader
  if (a.ones_b && b_kc) return false;
  const int Mr = a.M - a.ones, Nr = a.N - (a.ones | a.ones_b);   // extents that exist in memory
  return !v2_off && a_vec && b_vec && a.K > 0 &&
         (a_kc ? (a.K % 4 == 0 && Kc % 4 == 0) : (Mr % 4 == 0 && Mr >= 4)) &&
         (b_kc ? (a.K % 4 == 0 && Kc % 4 == 0) : (Nr % 4 == 0 && Nr >= 4));
}

// pointer tables: the vector loaders need every member 16-byte aligned relative to the first
static bool tab_aligned(const long *off, int batch) {
  for (int b = 0; b < batch; ++b)
    if (off[b] % 4 != 0) return false;
  return true;
}

int launch_gemm(GemmArgs a, int batch, hipStream_t stream) {
  if (a.tab_a || a.tab_b) {
    if (batch > GEMM_TAB_MAX || (a.tab_a && !tab_aligned(a.off_a, batch)) || (a.tab_b && !tab_aligned(a.off_b, batch))) {
      set_error("clo_gemm: a pointer-table batch needs <= %d members, each 16-byte aligned relative to the first", GEMM_TAB_MAX);
      return CLO_EUNSUP;
    }
    if (a.tab_a) a.sa_b = 4;   // (what the layout / alignment predicates see: "some aligned batch stride")
    if (a.tab_b) a.sb_b = 4;
  }
  a.tiles_m = (int)cdiv(a.M, BM);
  a.tiles_n = (int)cdiv(a.N, BN);
  a.tbm = BM; a.tbn = BN;
  if (a.splitk < 1) a.splitk = 1;
  const int ktiles = (int)cdiv(a.K, BK);
  if (a.splitk > ktiles) a.splitk = ktiles > 0 ? ktiles : 1;
  a.k_per_split = (int)cdiv(ktiles, a.splitk) * BK;
  a.splitk = (int)cdiv(a.K, a.k_per_split) > 0 ? (int)cdiv(a.K, a.k_per_split) : 1;
  if (a.splitk > 1 && a.ws == nullptr) {
    set_error("clo_gemm: splitk=%d needs a workspace", a.splitk);
    return CLO_EINVAL;
  }
  a.mode_a = pick_mode(a.A, a.sa_m, a.sa_k, a.sa_b, batch);
  a.mode_b = pick_mode(a.B, a.sb_n, a.sb_k, a.sb_b, batch);
  auto tile_blocks = [&]() {  // symmetric output: upper-triangular tiles only
    return a.sym ? (unsigned)((long)a.tiles_m * (a.tiles_m + 1) / 2) : (unsigned)(a.tiles_m * a.tiles_n);
  };
  dim3 grid(tile_blocks(), batch * a.splitk);
  const bool a_kc = !a.patch && a.mode_a == MODE_KC_VEC, b_kc = !a.patch && a.mode_b == MODE_KC_VEC;
  const bool v2 = a.patch ? true : gemm_v2_eligible(a, batch);  // patch operands are generated, not loaded
  constexpr int bk2 = 32;
  if (a.A2 && !(v2 && a.K1 % bk2 == 0 && a.K1 > 0 && a.K1 < a.K)) {
    set_error("clo_gemm: a second K segment needs the aligned engine and K1 %% %d == 0", bk2);
    return CLO_EUNSUP;
  }
  if ((a.ones_b || a.col_out) && (!v2 || a.sym || a.epi != EPI_NONE || batch != 1)) {
    set_error("clo_gemm: the B-side ones column / col_out need the aligned engine (plain single product)");
    return CLO_EUNSUP;
  }
  if (v2) {
    const V2Config cfg = v2_config(a.M, a.N, a.K, batch, a.sym, !a.A2 || a.K1 % 64 == 0);
    a.tiles_m = (int)cdiv(a.M, cfg.bm);
    a.tiles_n = (int)cdiv(a.N, cfg.bn);
    a.tbm = cfg.bm; a.tbn = cfg.bn;
    a.k_per_split = (int)cdiv(cdiv(a.K, a.splitk), cfg.bk) * cfg.bk;
    a.splitk = (int)cdiv(a.K, a.k_per_split);
    grid = dim3(tile_blocks(), batch * a.splitk);
#define CLO_V2(AK, BKC_, BKV, BMV, BNV, WM_, WN_) CLO_V2X(AK, BKC_, BKV, BMV, BNV, WM_, WN_, false)
#define CLO_V2X(AK, BKC_, BKV, BMV, BNV, WM_, WN_, PT)                                            \
  {                                                                                               \
    constexpr int nthr = WM_ * WN_ * 64;                                                          \
    const size_t smem_loop =                                                                      \
        2 * (TileIO<AK, BKV, nthr, BMV>::FLOATS + TileIO<BKC_, BKV, nthr, BNV>::FLOATS) * sizeof(float); \
    /* (the wide epilogue stages the finished tile, or its mirror image, in the same allocation) */ \
    const size_t smem_epi = (size_t)std::max(BMV * (BNV + 4), BNV * (BMV + 4)) * sizeof(float);   \
    const size_t smem = std::max(smem_loop, smem_epi);                                            \
    auto kern = gemm_v2_kernel<AK, BKC_, BKV, BMV, BNV, WM_, WN_, PT>;                            \
    if (smem > 64 * 1024) {                                                                       \
      static bool attr_set = false;                                                               \
      if (!attr_set) {                                                                            \
        int rc_ = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),             \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                                (int)smem), "hipFuncSetAttribute");               \
        if (rc_ != CLO_OK) return rc_;                                                            \
        attr_set = true;                                                                          \
      }                                                                                           \
    }                                                                                             \
    hipLaunchKernelGGL(kern, grid, dim3(nthr), smem, stream, a);                                  \
  }
#define CLO_V2L(BKV, BMV, BNV, WM_, WN_)                                \
  if (a_kc && b_kc) CLO_V2(true, true, BKV, BMV, BNV, WM_, WN_)         \
  else if (a_kc) CLO_V2(true, false, BKV, BMV, BNV, WM_, WN_)           \
  else if (b_kc) CLO_V2(false, true, BKV, BMV, BNV, WM_, WN_)           \
  else CLO_V2(false, false, BKV, BMV, BNV, WM_, WN_)
    const long nblocks = (long)grid.x * grid.y;
#ifndef CLO_GEMM_V3_MIN_K
#define CLO_GEMM_V3_MIN_K 129
#endif
    // (K <= 128 is four k tiles: the LDS-DMA ring of the v3 engine is barely full before it drains, the register-staged loop
    // below wins -- 2688 x 2688 x 128: 34.8 -> 28.4 us, 2304 x 2304 x 128: 33.7 -> 30.3, 2688 x 1024 x 128: 21.8 -> 18.3; at
    // K = 256 the engine is ahead again, 41.2 vs 44.8 us)
    const bool v3_deep = a.K >= CLO_GEMM_V3_MIN_K || cfg.bk == 64;
    if (v3_deep && ((cfg.bm == 128 && cfg.bk == 32) || (cfg.bk == 64 && gemm_v3_small(a, batch))) && gemm_v3_eligible(a, batch)) {
      if (cfg.bk == 64) {   // (the LDS-DMA engine's k tiles are 32 deep)
        a.k_per_split = (int)cdiv(cdiv(a.K, a.splitk), 32) * 32;
        a.splitk = (int)cdiv(a.K, a.k_per_split);
      }
      // LDS-DMA engine (gemm_v3.hip); with a stream-K workspace it may finish the split tiles itself
      bool used_streamk = false;
      int rc3 = launch_gemm_v3(a, batch, a_kc, b_kc, stream, &used_streamk, &a.tbm, &a.tbn);
      if (rc3 != CLO_OK) return rc3;
      if (used_streamk) a.splitk = 1;
    }
    else if (a.patch) {
      if (cfg.bk == 64) CLO_V2X(false, false, 64, 64, 64, 2, 2, true)
      else if (nblocks < 2L * kNumCU) CLO_V2X(false, false, 32, 128, 128, 2, 4, true)
      else CLO_V2X(false, false, 32, 128, 128, 2, 2, true)
    }
    else if (cfg.bk == 64) { CLO_V2L(64, 64, 64, 2, 2) }
    else if (cfg.bm == 32) { CLO_V2L(16, 32, 256, 1, 8) }
    else if (cfg.bm == 64) { CLO_V2L(16, 64, 256, 1, 8) }
    // 8 waves (two per SIMD inside one block) when the grid cannot put two blocks on every CU
    else if (nblocks < 2L * kNumCU) { CLO_V2L(32, 128, 128, 2, 4) }
    else { CLO_V2L(32, 128, 128, 2, 2) }
#undef CLO_V2L
#undef CLO_V2
#undef CLO_V2X
    CLO_CHECK_LAUNCH("gemm_v2_kernel");
  } else if (!a.A2 && (long)a.M * a.N * batch <= 128L * 128L && (long)a.M * a.N * a.K * batch <= (1L << 19)) {
    a.splitk = 1;
    const long total = (long)a.M * a.N;
    hipLaunchKernelGGL(gemm_tiny_kernel, dim3((unsigned)cdiv(total, 256), batch), dim3(256), 0, stream, a);
    CLO_CHECK_LAUNCH("gemm_tiny_kernel");
  } else {
    hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, stream, a);
    CLO_CHECK_LAUNCH("gemm_f32_kernel");
  }
  if (a.splitk > 1) {
    const long total = (long)a.M * a.N;
    dim3 rgrid((unsigned)std::min<long>(cdiv(total, 256), 4096), batch);
    hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, stream, a, a.splitk);
    CLO_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return CLO_OK;
}

// C = beta*C + alpha * sum_b (A_b B_b)^2 (elementwise square); the batch range is split over
// grid.y into `splits` partial slabs (ws) that the reduce kernel sums.
int launch_gemm_sqsum(GemmArgs a, int batch, int splits, hipStream_t stream) {
  a.tiles_m = (int)cdiv(a.M, BM);
  a.tiles_n = (int)cdiv(a.N, BN);
  a.tbm = BM; a.tbn = BN;
  a.nbatch = batch;
  splits = std::max(1, std::min(splits, batch));
  a.batch_per_split = (int)cdiv(batch, splits);
  splits = (int)cdiv(batch, a.batch_per_split);
  a.splitk = splits;
  a.k_per_split = 0;
  if (splits > 1 && a.ws == nullptr) {
    set_error("clo_gemm_sqsum: %d batch splits need a workspace", splits);
    return CLO_EINVAL;
  }
  a.mode_a = pick_mode(a.A, a.sa_m, a.sa_k, a.sa_b, batch);
  a.mode_b = pick_mode(a.B, a.sb_n, a.sb_k, a.sb_b, batch);
  a.sym = 0;
  dim3 grid(a.tiles_m * a.tiles_n, splits);
  // aligned operands: the SQ variant of the register-staged engine (float4 global loads, ds_read_b128 fragments); B may be
  // padded to n_mem floats per k row (the EKFAC correction pads its rotated inputs: joint weight + bias blocks are d_in + 1
  // = odd wide)
  GemmArgs e = a;
  if (a.n_mem) e.N = a.n_mem;
#ifndef CLO_GEMM_SQSUM_V2
#define CLO_GEMM_SQSUM_V2 1
#endif
  static const int sq_v2 = CLO_GEMM_SQSUM_V2;
  if (sq_v2 && (a.n_mem == 0 || (a.n_mem % 4 == 0 && a.n_mem >= a.N && a.sb_n == 1)) && gemm_v2_eligible(e, batch)) {
    const bool a_kc = a.mode_a == MODE_KC_VEC, b_kc = a.mode_b == MODE_KC_VEC;
#define CLO_SQ(AK, BKC_)                                                                                          \
  {                                                                                                               \
    constexpr int nthr = 512;                                                                                     \
    const size_t smem = std::max<size_t>(2 * (TileIO<AK, 32, nthr, 128>::FLOATS + TileIO<BKC_, 32, nthr, 128>::FLOATS) * sizeof(float), \
                                         (size_t)128 * 132 * sizeof(float));   /* (loop buffers / staged tile of the wide epilogue) */ \
    auto kern = gemm_v2_kernel<AK, BKC_, 32, 128, 128, 2, 4, false, true>;                                        \
    static bool attr_set = false;                                                                                 \
    if (smem > 64 * 1024 && !attr_set) {                                                                          \
      int rc_ = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                               \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),             \
                          "hipFuncSetAttribute");                                                                 \
      if (rc_ != CLO_OK) return rc_;                                                                              \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, grid, dim3(nthr), smem, stream, a);                                                  \
  }
    if (a_kc && b_kc) CLO_SQ(true, true)
    else if (a_kc) CLO_SQ(true, false)
    else if (b_kc) CLO_SQ(false, true)
    else CLO_SQ(false, false)
#undef CLO_SQ
    CLO_CHECK_LAUNCH("gemm_v2_kernel<sqsum>");
  } else {
    a.n_mem = 0;
    hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, stream, a);
    CLO_CHECK_LAUNCH("gemm_f32_kernel<sqsum>");
  }
  if (splits > 1) {
    const long total = (long)a.M * a.N;
    dim3 rgrid((unsigned)std::min<long>(cdiv(total, 256), 4096), 1);
    hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, stream, a, splits);
    CLO_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return CLO_OK;
}

}  // namespace clo

using namespace clo;

extern "C" int clo_gemm_sqsum_suggest_splits(int M, int N, int batch) {
  const long tiles = cdiv(M, BM) * cdiv(N, BN);
  if (tiles >= kNumCU || batch <= 1) return 1;
  return (int)std::max<long>(1, std::min<long>({(long)batch, (2L * kNumCU) / tiles, 64L}));
}

extern "C" int clo_gemm_sqsum_f32(int M, int N, int K, float alpha, const float *A, long sa_m,
                                  long sa_k, long sa_b, const float *B, long sb_k, long sb_n,
                                  long sb_b, float beta, float *C, long ldc, int batch, int splits,
                                  float *ws, void *stream) {
  CLO_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "clo_gemm_sqsum_f32: negative size");
  CLO_REQUIRE(ldc >= N, "clo_gemm_sqsum_f32: ldc (%ld) < N (%d)", ldc, N);
  if (M == 0 || N == 0) return CLO_OK;
  CLO_REQUIRE(C && (batch == 0 || (A && B)), "clo_gemm_sqsum_f32: null operand");
  GemmArgs a{};
  a.M = M; a.N = N; a.K = K; a.alpha = alpha; a.beta = beta;
  a.A = A; a.sa_m = sa_m; a.sa_k = sa_k; a.sa_b = sa_b;
  a.B = B; a.sb_k = sb_k; a.sb_n = sb_n; a.sb_b = sb_b;
  a.C = C; a.ldc = ldc; a.sc_b = 0; a.ws = ws;
  if (batch == 0) {  // nothing to add: C = beta * C
    a.K = 0;
    a.splitk = 1;
    return launch_gemm(a, 1, (hipStream_t)stream);
  }
  return launch_gemm_sqsum(a, batch, splits, (hipStream_t)stream);
}

// Split-K factor from a small cost model (ns): the MFMA phase costs one 128x128 k-slice
// (32768 flop at 256 flop/clk/CU = 53 ns) per k per block, blocks spread evenly over the CUs, and
// a split pays the slab round trip ((2s + 1) M N floats at ~3.5 TB/s) plus one more launch.
namespace clo {
// tile_scale = block tile area / (128 x 128): MFMA time per k of one block
// waves = waves per block: with fewer than two waves per SIMD resident on a CU nothing hides a
// wave's barrier / LDS stalls (measured ~1.5x the MFMA time)
// cap: largest split the caller's slab workspace can hold -- the search runs INSIDE the cap (clamping
// the unconstrained optimum afterwards lands on block counts like 1.3 x the CUs: two rounds of work
// for one round's worth of blocks)
int suggest_splitk_tiles(long tiles, long K, long MN, double tile_scale, int waves, long cap = 64) {
  if (tiles <= 0 || K <= 0) return 1;
  double best = 1e30;
  int best_s = 1;
  const long smax = std::min<long>({64L, std::max<long>(1, K / 64), std::max<long>(1, cap)});
  for (long s = 1; s <= smax; ++s) {
    const long kps = cdiv(cdiv(K, s), 32) * 32;
    const long se = cdiv(K, kps);
    if (se != s) continue;
    const double rounds = (double)cdiv(tiles * s, kNumCU);
    const double resident = (double)waves * std::min<double>(2.0, rounds);
    double t = rounds * (kps * tile_scale + 48.0) * 53.0 * (resident >= 8.0 ? 1.0 : 1.5);
    if (s > 1) t += (2.0 * s + 1.0) * MN * 4.0 / 3500.0 + 4000.0;
    if (t < best) { best = t; best_s = (int)s; }
  }
  return best_s;
}
}  // namespace clo

// `aligned`: the v2 engine will take the problem (its tile configuration applies); otherwise the
// v1 kernel with its fixed 128 x 128 tiles runs
static int suggest_splitk_for(int M, int N, int K, long b, bool aligned, long cap = 64) {
  const V2Config cfg = aligned ? v2_config(M, N, K, b, 0) : V2Config{128, 128, 32};
  return clo::suggest_splitk_tiles(cdiv(M, cfg.bm) * cdiv(N, cfg.bn) * b, K, (long)M * N * b,
                                   (double)cfg.bm * cfg.bn / (128.0 * 128.0), cfg.bk == 64 ? 4 : 8, cap);
}

// Symmetric product C = X^T X (d x d from `rows` rows): only the upper-triangular tiles are computed
// (and only they write slabs), so the split that fills the chip is about twice the one of the full
// d x d product.  Measured (tools/probe_syrk_mid.py): 32768 x 576: 281 us at the full-product split 6,
// 177 us at 16; 8192 x 1152: 221 -> 183 us; 2048 x 2304: 197 -> 167 us.
extern "C" int clo_syrk_suggest_splitk(int d, long rows) {
  if (d <= 0 || rows <= 0) return 1;
  const int K = (int)std::min<long>(rows, 1L << 30);
  const bool aligned = d % 4 == 0;
  const V2Config cfg = aligned ? v2_config(d, d, K, 1, 1) : V2Config{128, 128, 32};
  const long t = cdiv(d, cfg.bm);
  return clo::suggest_splitk_tiles(t * (t + 1) / 2, K, (long)d * d * 6 / 10,
                                   (double)cfg.bm * cfg.bn / (128.0 * 128.0), cfg.bk == 64 ? 4 : 8, 64);
}

// -1: ask for the stream-K schedule of the LDS-DMA engine (workspace of clo_gemm_streamk_ws_floats() floats)
extern "C" int clo_gemm_suggest_splitk(int M, int N, int K, int batch) {
  // without the operands: float4-complete extents are taken as "aligned"
  const bool aligned = M % 4 == 0 && N % 4 == 0 && K % 4 == 0;
  const long b = batch > 0 ? batch : 1;
  if (aligned && M > 0 && N > 0) {
    const V2Config cfg = v2_config(M, N, K, b, 0);
    if (cfg.bm == 128 && cfg.bk == 32 && clo::gemm_v3_would_streamk(M, N, K, b)) return -1;
  }
  return suggest_splitk_for(M, N, K, b, aligned);
}

extern "C" long clo_gemm_streamk_ws_floats(void) { return clo::gemm_streamk_ws_floats(); }

extern "C" int clo_gemm_f32(int M, int N, int K, float alpha, const float *A, long sa_m, long sa_k,
                            long sa_b, const float *B, long sb_k, long sb_n, long sb_b, float beta,
                            float *C, long ldc, long sc_b, int batch, int splitk, float *ws,
                            void *stream) {
  CLO_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "clo_gemm_f32: negative size");
  CLO_REQUIRE(ldc >= N, "clo_gemm_f32: ldc (%ld) < N (%d)", ldc, N);
  if (M == 0 || N == 0 || batch == 0) return CLO_OK;
  CLO_REQUIRE(A && B && C, "clo_gemm_f32: null operand");
  GemmArgs a{};
  a.M = M; a.N = N; a.K = K; a.alpha = alpha; a.beta = beta;
  a.A = A; a.sa_m = sa_m; a.sa_k = sa_k; a.sa_b = sa_b;
  a.B = B; a.sb_k = sb_k; a.sb_n = sb_n; a.sb_b = sb_b;
  a.C = C; a.ldc = ldc; a.sc_b = sc_b;
  a.splitk = splitk; a.ws = ws; a.sym = 0;
  if (splitk < 0) {  // stream-K request; an engine that cannot honour it runs unsplit
    CLO_REQUIRE(ws, "clo_gemm_f32: splitk = -1 (stream-K) needs a workspace of clo_gemm_streamk_ws_floats() floats");
    a.splitk = 1;
    a.streamk = 2;   // the workspace holds the partial tiles of any configuration
  }
  return launch_gemm(a, batch, (hipStream_t)stream);
}

// Batched product whose members of A and / or B lie at arbitrary addresses: A_ptrs / B_ptrs are HOST arrays of `batch` device
// pointers (NULL: that operand is strided as in clo_gemm_f32, base A / B).  C is strided.  batch <= 8.
extern "C" int clo_gemm_ptrs_f32(int M, int N, int K, float alpha, const float *A, const float *const *A_ptrs, long sa_m,
                                 long sa_k, long sa_b, const float *B, const float *const *B_ptrs, long sb_k, long sb_n,
                                 long sb_b, float beta, float *C, long ldc, long sc_b, int batch, int splitk, float *ws,
                                 void *stream) {
  CLO_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0 && batch <= GEMM_TAB_MAX, "clo_gemm_ptrs_f32: bad extents (batch <= %d)", GEMM_TAB_MAX);
  CLO_REQUIRE(ldc >= N, "clo_gemm_ptrs_f32: ldc (%ld) < N (%d)", ldc, N);
  if (M == 0 || N == 0 || batch == 0) return CLO_OK;
  CLO_REQUIRE((A || A_ptrs) && (B || B_ptrs) && C, "clo_gemm_ptrs_f32: null operand");
  GemmArgs a{};
  a.M = M; a.N = N; a.K = K; a.alpha = alpha; a.beta = beta;
  a.A = A_ptrs ? A_ptrs[0] : A; a.sa_m = sa_m; a.sa_k = sa_k; a.sa_b = sa_b;
  a.B = B_ptrs ? B_ptrs[0] : B; a.sb_k = sb_k; a.sb_n = sb_n; a.sb_b = sb_b;
  a.C = C; a.ldc = ldc; a.sc_b = sc_b;
  for (int b = 0; b < batch; ++b) {
    if (A_ptrs) { CLO_REQUIRE(A_ptrs[b], "clo_gemm_ptrs_f32: null A member %d", b); a.off_a[b] = A_ptrs[b] - A_ptrs[0]; }
    if (B_ptrs) { CLO_REQUIRE(B_ptrs[b], "clo_gemm_ptrs_f32: null B member %d", b); a.off_b[b] = B_ptrs[b] - B_ptrs[0]; }
  }
  a.tab_a = A_ptrs ? 1 : 0; a.tab_b = B_ptrs ? 1 : 0;
  a.splitk = splitk; a.ws = ws; a.sym = 0;
  if (splitk < 0) {
    CLO_REQUIRE(ws, "clo_gemm_ptrs_f32: splitk = -1 (stream-K) needs a workspace of clo_gemm_streamk_ws_floats() floats");
    a.splitk = 1;
    a.streamk = 2;
  }
  return launch_gemm(a, batch, (hipStream_t)stream);
}

extern "C" int clo_syrk_accum_f32(float *C, long ldc, const float *X, long rows, int d, long ldx,
                                  int ones_col, float alpha, float beta, int splitk, float *ws,
                                  void *stream) {
  CLO_REQUIRE(d >= 0 && rows >= 0, "clo_syrk_accum_f32: negative size");
  const int dd = d + (ones_col ? 1 : 0);
  CLO_REQUIRE(ldc >= dd, "clo_syrk_accum_f32: ldc (%ld) < d (%d)", ldc, dd);
  CLO_REQUIRE(ldx >= d, "clo_syrk_accum_f32: ldx (%ld) < d (%d)", ldx, d);
  CLO_REQUIRE(rows < (1L << 31), "clo_syrk_accum_f32: rows must fit int32");
  if (dd == 0) return CLO_OK;
  CLO_REQUIRE(C && (X || rows == 0 || d == 0), "clo_syrk_accum_f32: null operand");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{};
  a.M = dd; a.N = dd; a.K = (int)rows; a.alpha = alpha; a.beta = beta;
  a.A = X; a.sa_m = 1; a.sa_k = ldx; a.sa_b = 0;   // A = [X | 1]^T : A(m,k) = X[k][m]
  a.B = X; a.sb_k = ldx; a.sb_n = 1; a.sb_b = 0;
  a.C = C; a.ldc = ldc; a.sc_b = 0;
  a.splitk = splitk; a.ws = ws; a.sym = 1;
  a.ones = ones_col ? 1 : 0;   // the ones column is synthesised by the tile loader
  int rc = launch_gemm(a, 1, st);
  if (rc != CLO_OK) return rc;
  return CLO_OK;
}

// KFAC input covariance of a Conv2d layer without the patch matrix (reference kfac_utils.py:78-121 +
// kfac_hooks.py:355-393): C = beta C + alpha [P | 1]^T [P | 1] with P = unfold(x)^T generated inside the
// tile loader of the symmetric MFMA GEMM.
extern "C" int clo_im2col_syrk_accum_f32(float *C, long ldc, const float *x, int B, int Cc, int H, int W,
                                         int KH, int KW, int SH, int SW, int PH, int PW, int DH, int DW,
                                         int OH, int OW, int ones_col, float alpha, float beta, int splitk,
                                         float *ws, void *stream) {
  CLO_REQUIRE(B >= 0 && Cc > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && SH > 0 && SW > 0 && DH > 0 &&
                  DW > 0 && PH >= 0 && PW >= 0,
              "clo_im2col_syrk_accum_f32: bad geometry");
  CLO_REQUIRE(OH == (H + 2 * PH - DH * (KH - 1) - 1) / SH + 1 &&
                  OW == (W + 2 * PW - DW * (KW - 1) - 1) / SW + 1 && OH > 0 && OW > 0,
              "clo_im2col_syrk_accum_f32: output size (%d, %d) inconsistent with the geometry", OH, OW);
  CLO_REQUIRE(KH * DH < 32768 && KW * DW < 32768, "clo_im2col_syrk_accum_f32: kernel extent too large");
  const long rows = (long)B * OH * OW;
  CLO_REQUIRE(rows < (1L << 31) && (long)B * Cc * H * W < (1L << 31),
              "clo_im2col_syrk_accum_f32: sizes must fit int32");
  const int dd = Cc * KH * KW + (ones_col ? 1 : 0);
  CLO_REQUIRE(ldc >= dd, "clo_im2col_syrk_accum_f32: ldc (%ld) < d (%d)", ldc, dd);
  CLO_REQUIRE(C && (x || rows == 0), "clo_im2col_syrk_accum_f32: null operand");
  GemmArgs a{};
  a.M = dd; a.N = dd; a.K = (int)rows; a.alpha = alpha; a.beta = beta;
  a.A = x; a.B = x; a.sa_m = 1; a.sa_k = dd; a.sb_n = 1; a.sb_k = dd;
  a.C = C; a.ldc = ldc;
  a.splitk = splitk; a.ws = ws; a.sym = 1;
  a.ones = ones_col ? 1 : 0;
  a.patch = 1;
  a.cvC = Cc; a.cvH = H; a.cvW = W; a.cvKH = KH; a.cvKW = KW; a.cvSH = SH; a.cvSW = SW;
  a.cvPH = PH; a.cvPW = PW; a.cvDH = DH; a.cvDW = DW; a.cvOH = OH; a.cvOW = OW;
  return launch_gemm(a, 1, (hipStream_t)stream);
}

namespace clo {
// Convenience wrapper for the MLP large-batch path: single problem, split-K chosen from the
// problem shape and capped by the caller's workspace.
int launch_gemm_simple(int M, int N, int K, float alpha, const float *A, long sa_m, long sa_k,
                       const float *B, long sb_k, long sb_n, float beta, float *C, long ldc,
                       float *ws, long ws_floats, hipStream_t st) {
  GemmArgs a{};
  a.M = M; a.N = N; a.K = K; a.alpha = alpha; a.beta = beta;
  a.A = A; a.sa_m = sa_m; a.sa_k = sa_k; a.sa_b = 0;
  a.B = B; a.sb_k = sb_k; a.sb_n = sb_n; a.sb_b = 0;
  a.C = C; a.ldc = ldc; a.sc_b = 0;
  const long per = (long)M * N;
  long s = suggest_splitk_for(M, N, K, 1, gemm_v2_eligible(a, 1), per > 0 && ws ? ws_floats / per : 1);
  if (per > 0) s = std::min<long>(s, ws ? ws_floats / per : 1);
  a.splitk = (int)std::max<long>(1, s);
  a.ws = ws; a.sym = 0;
  return launch_gemm(a, 1, st);
}

// Single problem described by `a` (operands, epilogue, optional second K segment); split-K from the
// cost model, capped by the caller's workspace.
int launch_gemm_auto(GemmArgs a, float *ws, long ws_floats, hipStream_t st, int batch) {
  const long per = (long)a.M * a.N * batch;
  long s = suggest_splitk_for(a.M, a.N, a.K, batch, gemm_v2_eligible(a, batch),
                              per > 0 && ws ? ws_floats / per : 1);
  if (per > 0) s = std::min<long>(s, ws ? ws_floats / per : 1);
  a.splitk = (int)std::max<long>(1, s);
  a.ws = ws;
  a.streamk = !ws ? 0 : ws_floats >= gemm_streamk_ws_floats() ? 2 : ws_floats >= gemm_streamk_ws_floats_square() ? 1 : 0;
  return launch_gemm(a, batch, st);
}

// Fused forward + JVP of one Linear layer for a batch of N rows (see gemm_fwd3_kernel).  Needs
// d_in % 4 == 0 and 16-byte aligned operands (returns CLO_EUNSUP otherwise: the caller then runs the
// separate GEMMs); ws_floats bounds the split-K slabs.
int launch_mlp_fwd3(const float *A, const float *dA, const float *W, const float *V, const float *b,
                    const float *Vb, float *a, float *da, float *dphi, int N, int d_in, int d_out,
                    int act, float *ws, long ws_floats, hipStream_t st) {
  const bool ok = d_in % 4 == 0 && aligned16(A) && aligned16(W) && aligned16(V) && (!dA || aligned16(dA));
  if (!ok) {
    set_error("launch_mlp_fwd3: unaligned operands");
    return CLO_EUNSUP;
  }
  Fwd3Args p{};
  p.A = A; p.dA = dA; p.W = W; p.V = V; p.b = b; p.Vb = Vb; p.a = a; p.da = da; p.dphi = dphi;
  p.ws = ws; p.N = N; p.d_in = d_in; p.d_out = d_out; p.act = act;
  // tiny layers (small networks): one block's k loop is a chain of memory round trips, so small
  // tiles on more CUs with 64-deep k steps (as v2_config does for plain products)
  const bool tiny = (long)N * d_out <= 256L * 256L && d_in <= 1024;
  // (round 6: 64-row tiles also where they pad less than 128-row tiles -- 129 ... 192 rows are three tiles of 64 instead of two of
  // 128, the second mostly padding: the 128 -> 129-row step of the matvec was 180 -> 257 us)
#ifndef CLO_FWD3_PAD64
#define CLO_FWD3_PAD64 1
#endif
  // (measured: 129 / 160 / 192 rows 267 -> 245 / 261 -> 243 / 276 -> 258 us; 257 ... 320 rows LOSE 23 us with five 64-row tiles)
  const bool pad64 = CLO_FWD3_PAD64 && N > 128 && N <= 192;
#ifndef CLO_FWD3_PAD32
#define CLO_FWD3_PAD32 1
#endif
  const bool pad32 = CLO_FWD3_PAD32 && N > 64 && N <= 96;   // (three 32-row tiles instead of one 128-row tile that is a third padding)
  const int bm = tiny ? 32 : (N <= 32 || pad32 ? 32 : (N <= 64 || pad64 ? 64 : 128));
  const int bn = tiny ? 64 : (bm == 128 ? 64 : 128);
  const int bk = tiny ? 64 : 16;
  p.tiles_m = (int)cdiv(N, bm);
  p.tiles_n = (int)cdiv(d_out, bn);
  const long tiles = (long)p.tiles_m * p.tiles_n, MN = (long)N * d_out;
  // two or three products per tile: the MFMA time of a plain tile of the same area times that
  long s = suggest_splitk_tiles(tiles, d_in, 2 * MN, (dA ? 3.0 : 2.0) * bm * bn / (128.0 * 128.0),
                                tiny ? 2 : (bm == 64 ? 8 : 4), MN > 0 && ws ? ws_floats / (2 * MN) : 1);
  if (MN > 0) s = std::min<long>(s, ws ? ws_floats / (2 * MN) : 1);
  s = std::max<long>(1, s);
  p.k_per_split = (int)cdiv(cdiv(d_in, s), bk) * bk;
  p.splitk = (int)cdiv(d_in, p.k_per_split);
  dim3 grid((unsigned)tiles, (unsigned)p.splitk);
#define CLO_F3(BKV, BMV, BNV, WM_, WN_, DA_)                                                       \
  {                                                                                                \
    constexpr int nthr = WM_ * WN_ * 64;                                                           \
    const size_t smem = 2 * ((DA_ ? 2 : 1) * TileIO<true, BKV, nthr, BMV>::FLOATS +                \
                             2 * TileIO<true, BKV, nthr, BNV>::FLOATS) * sizeof(float);            \
    auto kern = gemm_fwd3_kernel<BKV, BMV, BNV, WM_, WN_, DA_>;                                    \
    if (smem > 64 * 1024) {                                                                        \
      static bool attr_set = false;                                                                \
      if (!attr_set) {                                                                             \
        int rc_ = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),              \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                                (int)smem), "hipFuncSetAttribute");                \
        if (rc_ != CLO_OK) return rc_;                                                             \
        attr_set = true;                                                                           \
      }                                                                                            \
    }                                                                                              \
    hipLaunchKernelGGL(kern, grid, dim3(nthr), smem, st, p);                                       \
  }
#define CLO_F3L(BKV, BMV, BNV, WM_, WN_) \
  if (dA) CLO_F3(BKV, BMV, BNV, WM_, WN_, true) else CLO_F3(BKV, BMV, BNV, WM_, WN_, false)
  if (tiny) { CLO_F3L(64, 32, 64, 1, 2) }
  else if (bm == 32) { CLO_F3L(16, 32, 128, 1, 4) }
  else if (bm == 64) { CLO_F3L(16, 64, 128, 2, 4) }
  else { CLO_F3L(16, 128, 64, 2, 2) }
#undef CLO_F3L
#undef CLO_F3
  CLO_CHECK_LAUNCH("gemm_fwd3_kernel");
  if (p.splitk > 1) {
    hipLaunchKernelGGL(fwd3_reduce_kernel, dim3((unsigned)std::min<long>(cdiv(MN, 256), 4096)), dim3(256),
                       0, st, p);
    CLO_CHECK_LAUNCH("fwd3_reduce_kernel");
  }
  return CLO_OK;
}

// C = beta*C + alpha * X^T X (X row-major [rows][ldx], first d columns), symmetric block raster.
int launch_syrk_simple(float *C, long ldc, const float *X, long rows, int d, long ldx, float alpha,
                       float beta, float *ws, long ws_floats, hipStream_t st) {
  GemmArgs a{};
  a.M = d; a.N = d; a.K = (int)rows; a.alpha = alpha; a.beta = beta;
  a.A = X; a.sa_m = 1; a.sa_k = ldx; a.sa_b = 0;
  a.B = X; a.sb_k = ldx; a.sb_n = 1; a.sb_b = 0;
  a.C = C; a.ldc = ldc; a.sc_b = 0;
  const long td = cdiv(d, BM);
  const long per = (long)d * d;
  long s = suggest_splitk_tiles(td * (td + 1) / 2, rows, (long)d * d, 1.0, 8, per > 0 && ws ? ws_floats / per : 1);
  if (per > 0) s = std::min<long>(s, ws ? ws_floats / per : 1);
  a.splitk = (int)std::max<long>(1, s);
  a.ws = ws; a.sym = 1;
  return launch_gemm(a, 1, st);
}
}  // namespace clo
