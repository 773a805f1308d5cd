// fp32 GEMM / SYRK on the gfx950 f32 MFMA pipe (v_mfma_f32_32x32x2_f32).
//
// Block tile 128x128x16, 256 threads = 4 waves in a 2x2 arrangement, each wave owns a
// 64x64 sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator registers).  Both operand
// tiles are staged in LDS k-major ([k][outer], row stride 132 floats) so that the MFMA
// operand fetch (lane l needs element (outer = l&31, k = l>>5)) is a conflict-free
// ds_read_b32 whatever the global layout was; the global->LDS copy goes through
// registers, is coalesced for either "outer-contiguous" or "k-contiguous" operands and
// is software-pipelined one tile ahead of the MFMAs (double-buffered LDS, one barrier per
// k-tile).  f32 MFMA issues at 64 cycles per instruction, so LDS bandwidth is irrelevant;
// the design goal is simply to keep the matrix pipe fed while HBM/L2 latency is hidden.
//
// Reference call sites this replaces: kronecker.py:141-171 (einsum 'abZ,Aa,Bb->ABZ'),
// eigh.py:84-105, computers/kfac_hooks.py:350,390 (einsum "b s i, b s j -> i j").
#include "clo_common.h"

namespace clo {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_STRIDE = BM + 4;  // floats; 16-byte multiple, breaks power-of-two rows

enum LoadMode { MODE_OC_VEC = 0, MODE_KC_VEC = 1, MODE_OC_SCALAR = 2, MODE_KC_SCALAR = 3 };

struct GemmArgs {
  int M, N, K;
  float alpha, beta;
  const float *A;
  long sa_m, sa_k, sa_b;
  const float *B;
  long sb_k, sb_n, sb_b;
  float *C;
  long ldc, sc_b;
  int splitk;
  int k_per_split;  // multiple of BK
  float *ws;
  int sym;  // 1: compute only block-upper triangle, mirror on write (SYRK)
  int mode_a, mode_b;
  int tiles_m, tiles_n;
  int nbatch, batch_per_split;  // SQSUM mode only
  int ones;  // 1: outer index M-1 of A / N-1 of B is an implicit column of ones ([X | 1])
};

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
  int lid;
  {
    const int b = blockIdx.x;
    const int q = ntiles / kNumXCD, rem = ntiles % kNumXCD;
    const int xcd = b % kNumXCD, idx = b / kNumXCD;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  int bm, bn;
  {
    constexpr int GROUP = 8;
    const int per_group = GROUP * p.tiles_n;
    const int g = lid / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(p.tiles_m - first_m, GROUP);
    const int in_g = lid % per_group;
    bm = first_m + in_g % gsz;
    bn = in_g / gsz;
  }
  if (p.sym && bn < bm) return;

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
  const float *A = p.A + (long)bcur * p.sa_b;
  const float *B = p.B + (long)bcur * p.sb_b;
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

// C = alpha * sum_s ws[b][s] + beta * C ; for sym the lower block-triangle of ws was never
// written, take the transposed element instead.
__global__ void splitk_reduce_kernel(float *C, long ldc, long sc_b, const float *ws, int M, int N,
                                     int splitk, float alpha, float beta, int sym) {
  const long total = (long)M * N;
  const int b = blockIdx.y;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int m = e / N, n = e % N;
    long src = e;
    if (sym && (n / BN) < (m / BM)) src = (long)n * N + m;
    float s = 0.f;
    for (int k = 0; k < splitk; ++k) s += ws[((long)b * splitk + k) * total + src];
    float *c = C + (long)b * sc_b + (long)m * ldc + n;
    float v = alpha * s;
    if (beta != 0.f) v += beta * *c;
    *c = v;
  }
}

static int pick_mode(const float *P, long so, long sk, long sbatch, int batch) {
  const bool batch_ok = batch <= 1 || (sbatch % 4 == 0);
  if (so == 1) return (sk % 4 == 0 && aligned16(P) && batch_ok) ? MODE_OC_VEC : MODE_OC_SCALAR;
  if (sk == 1) return (so % 4 == 0 && aligned16(P) && batch_ok) ? MODE_KC_VEC : MODE_KC_SCALAR;
  return so <= sk ? MODE_OC_SCALAR : MODE_KC_SCALAR;
}

int launch_gemm(GemmArgs a, int batch, hipStream_t stream) {
  a.tiles_m = (int)cdiv(a.M, BM);
  a.tiles_n = (int)cdiv(a.N, BN);
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
  dim3 grid(a.tiles_m * a.tiles_n, batch * a.splitk);
  hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, stream, a);
  CLO_CHECK_LAUNCH("gemm_f32_kernel");
  if (a.splitk > 1) {
    const long total = (long)a.M * a.N;
    dim3 rgrid((unsigned)std::min<long>(cdiv(total, 256), 4096), batch);
    hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, stream, a.C, a.ldc, a.sc_b,
                       a.ws, a.M, a.N, a.splitk, a.alpha, a.beta, a.sym);
    CLO_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return CLO_OK;
}

// C = beta*C + alpha * sum_b (A_b B_b)^2 (elementwise square); the batch range is split over
// grid.y into `splits` partial slabs (ws) that the reduce kernel sums.
int launch_gemm_sqsum(GemmArgs a, int batch, int splits, hipStream_t stream) {
  a.tiles_m = (int)cdiv(a.M, BM);
  a.tiles_n = (int)cdiv(a.N, BN);
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
  hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, stream, a);
  CLO_CHECK_LAUNCH("gemm_f32_kernel<sqsum>");
  if (splits > 1) {
    const long total = (long)a.M * a.N;
    dim3 rgrid((unsigned)std::min<long>(cdiv(total, 256), 4096), 1);
    hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, stream, a.C, a.ldc, 0L, a.ws, a.M,
                       a.N, splits, a.alpha, a.beta, 0);
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

extern "C" int clo_gemm_suggest_splitk(int M, int N, int K, int batch) {
  const long tiles = cdiv(M, BM) * cdiv(N, BN) * (long)(batch > 0 ? batch : 1);
  const long ktiles = cdiv(K, BK);
  if (tiles >= 2 * kNumCU || ktiles < 16) return 1;
  long s = (2 * kNumCU) / tiles;          // aim at ~2 blocks per CU
  s = std::min<long>(s, ktiles / 8);      // keep >= 8 k-tiles (128 k) per split
  return (int)std::max<long>(1, std::min<long>(s, 64));
}

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
  long s = clo_gemm_suggest_splitk(M, N, K, 1);
  const long per = (long)M * N;
  if (per > 0) s = std::min<long>(s, ws ? ws_floats / per : 1);
  a.splitk = (int)std::max<long>(1, s);
  a.ws = ws; a.sym = 0;
  return launch_gemm(a, 1, st);
}

// C = beta*C + alpha * X^T X (X row-major [rows][ldx], first d columns), symmetric block raster.
int launch_syrk_simple(float *C, long ldc, const float *X, long rows, int d, long ldx, float alpha,
                       float beta, float *ws, long ws_floats, hipStream_t st) {
  GemmArgs a{};
  a.M = d; a.N = d; a.K = (int)rows; a.alpha = alpha; a.beta = beta;
  a.A = X; a.sa_m = 1; a.sa_k = ldx; a.sa_b = 0;
  a.B = X; a.sb_k = ldx; a.sb_n = 1; a.sb_b = 0;
  a.C = C; a.ldc = ldc; a.sc_b = 0;
  long s = clo_gemm_suggest_splitk(d, d, (int)rows, 1);
  const long per = (long)d * d;
  if (per > 0) s = std::min<long>(s, ws ? ws_floats / per : 1);
  a.splitk = (int)std::max<long>(1, s);
  a.ws = ws; a.sym = 1;
  return launch_gemm(a, 1, st);
}
}  // namespace clo
