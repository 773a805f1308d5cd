// All the small covariance products of one KFAC factor build in ONE launch (round 6).
//
//   C_p = beta_p C_p + alpha_p [X_p | 1]^T [X_p | 1],   p = 0 .. P-1,   X_p row-major [rows_p][ldx_p], d_p columns
//
// reference computers/kfac_hooks.py:335-393 (one einsum "b s i, b s j -> i j" per layer and backpropagated vector).  The
// output-gradient covariances of a ResNet-18 batch are 21 products of 64 ... 512 columns against 512 ... 131072 rows -- 6 GFLOP,
// 40 us of matrix pipe -- that ran as 32 separate split-K GEMMs + 29 reductions: 0.9 ms of launch-sized latencies
// (profiles/r05_kfac_resnet18_build_kernels.txt).  Here every (problem, 64 x 64 upper tile, row chunk) is one work item of a
// single grid; a chunk's partial tile goes to a slab, the LAST workgroup to finish a tile (returned atomic counter, agent-scope
// release / acquire around it) sums the slabs in chunk order -- the result does not depend on timing -- applies alpha / beta and
// mirrors the tile.  Tiles with one chunk skip the slab.  Rows are the K dimension of v_mfma_f32_32x32x2 tiles; operands are
// staged through LDS in 32-row slices (register double buffer), 16-byte loads where the rows allow it.
#include "clo_common.h"

#include <type_traits>

namespace clo {

using f32x16s = __attribute__((ext_vector_type(16))) float;

constexpr int SG_MAXP = 40;       // problems per launch (the descriptor table travels as kernel arguments)
constexpr int SG_T = 64;          // tile edge
constexpr int SG_ROWS = 32;       // rows per LDS slice
constexpr int SG_LD = 2 * SG_T + 4;
#ifndef CLO_SG_MAX_CHUNKS
#define CLO_SG_MAX_CHUNKS 64
#endif
#ifndef CLO_SG_PIPE
#define CLO_SG_PIPE 0
#endif
constexpr int SG_MAX_CHUNKS = CLO_SG_MAX_CHUNKS;   // chunks per tile, at most (a 131072-row stem factor: 64 chunks of 2048 rows)

struct SgProb {
  const float *X;
  float *C;
  long rows, ldx, ldc, chunk;
  int d, ones, T, nchunk, item0, tile0, vec;
  float alpha, beta;
};
struct SgArgs {
  int P, nitems;
  float *slab;        // [items with nchunk > 1][64][64]
  unsigned *cnt;      // [sum of tiles]  (zero on entry; the finisher of a tile puts its counter back to zero)
  SgProb pr[SG_MAXP];
};

__global__ __launch_bounds__(256) void syrk_grouped_kernel(const SgArgs a) {
  // CLO_SG_PIPE = 1: two LDS slices (one barrier per slice) and three slices in flight in registers -- measured SLOWER than
  // the plain form (ResNet-18's 20 gradient covariances: 284 vs 188 us, gpurun_out/r6_run7): 34 KB of LDS per workgroup
  // halve the workgroups per CU, and it is the number of resident workgroups that hides this kernel's load latency
  __shared__ __attribute__((aligned(16))) float S2[(CLO_SG_PIPE ? 2 : 1) * SG_ROWS * SG_LD];
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int item = blockIdx.x;
  int p = 0;
  while (p + 1 < a.P && item >= a.pr[p + 1].item0) ++p;
  const SgProb &pr = a.pr[p];
  const int local = item - pr.item0;
  const int tile = local / pr.nchunk, chunk = local - tile * pr.nchunk;
  // upper-triangular tiles, column by column: tile = bj (bj + 1) / 2 + bi
  int bj = (int)((sqrtf(8.f * (float)tile + 1.f) - 1.f) * 0.5f);
  while ((bj + 1) * (bj + 2) / 2 <= tile) ++bj;
  while (bj * (bj + 1) / 2 > tile) --bj;
  const int bi = tile - bj * (bj + 1) / 2;
  const bool diag = bi == bj;
  const int d = pr.d, dd = d + pr.ones;
  const long r_begin = (long)chunk * pr.chunk, r_end = min(pr.rows, r_begin + pr.chunk);

  // ---- loader: slice = 32 rows x (64 columns of block bi | 64 columns of block bj); 4 quads (16 B) per thread
  int lrow[4], lcol[4], gcol[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int q = tid + 256 * u, row = q >> 5, quad = q & 31;
    // diagonal tiles load only the first half (quads 0..15 of each row): q runs over 32 rows x 16 quads = 512
    const int rowd = q >> 4, quadd = q & 15;
    lrow[u] = diag ? rowd : row;
    const int qd = diag ? quadd : quad;
    lcol[u] = qd * 4;                                            // column inside the slice (0 .. 127)
    gcol[u] = (qd < 16 ? bi * SG_T : bj * SG_T - SG_T) + qd * 4;   // column of [X | 1]
  }
  // Loads are UNCONDITIONAL (clamped address + select): a predicated load makes hipcc branch and wait for vmcnt(0) right
  // behind it, which serialises the four round trips of a slice (the first version of this kernel spent 6 us per slice that
  // way).  `fast` (uniform per problem): 16-byte loads, every quad either fully inside the d columns or fully outside.
  const bool fast = pr.vec && (d & 3) == 0;
  const long rlast = max(r_end - 1, r_begin);
  auto fetch = [&](auto FAST, auto DIAG, long r0, float4 (&v)[4]) {
    constexpr int NQ = decltype(DIAG)::value ? 2 : 4;   // a diagonal tile loads one 64-column block only
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const long r = r0 + lrow[u];
      const int c = gcol[u];
      const bool rok = r < r_end;
      const float *row = pr.X + min(r, rlast) * pr.ldx;
      if constexpr (decltype(FAST)::value) {
        const float4 t = *reinterpret_cast<const float4 *>(row + min(c, d - 4));
        const bool in = rok && c < d;
        const float one = (rok && c == d && pr.ones) ? 1.f : 0.f;
        v[u] = make_float4(in ? t.x : one, in ? t.y : 0.f, in ? t.z : 0.f, in ? t.w : 0.f);
      } else {
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = d > 0 ? row[min(c + e, d - 1)] : 0.f;
          t[e] = !rok ? 0.f : (c + e < d ? x : (c + e == d && pr.ones ? 1.f : 0.f));
        }
        v[u] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  };
  auto stash = [&](auto DIAG, const float4 (&v)[4], float *S) {
    constexpr int NQ = decltype(DIAG)::value ? 2 : 4;
#pragma unroll
    for (int u = 0; u < NQ; ++u) *reinterpret_cast<float4 *>(&S[lrow[u] * SG_LD + lcol[u]]) = v[u];
  };

  const int wi = wave >> 1, wj = wave & 1;
  const int li = lane & 31, lk = lane >> 5;
  const int oa = lk * SG_LD + wi * 32 + li, ob = lk * SG_LD + (diag ? 0 : SG_T) + wj * 32 + li;
  f32x16s acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  auto k_loop = [&](auto FAST, auto DIAG) {
#if !CLO_SG_PIPE
  float4 cur[4], nxt[4];
  if (r_begin < r_end) fetch(FAST, DIAG, r_begin, cur);
  for (long r0 = r_begin; r0 < r_end; r0 += SG_ROWS) {
    float *S = S2;
    __syncthreads();            // everybody is done reading the previous slice
    stash(DIAG, cur, S);
    fetch(FAST, DIAG, r0 + SG_ROWS, nxt);   // (unconditional: rows past r_end are masked, addresses clamped)
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SG_ROWS / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(S[oa + 2 * s * SG_LD], S[ob + 2 * s * SG_LD], acc, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < (decltype(DIAG)::value ? 2 : 4); ++u) cur[u] = nxt[u];
  }
#else
  // three slices in flight in registers (the loads of slice s + 3 are issued before the MFMAs of slice s), the LDS image
  // double-buffered: slice s + 1 is written while other waves may still read slice s - 1's neighbour -- never the same buffer
  float4 r0v[4], r1v[4], r2v[4];
  if (r_begin < r_end) {
    fetch(FAST, DIAG, r_begin, r0v);
    fetch(FAST, DIAG, r_begin + SG_ROWS, r1v);        // (rows past r_end come back as zeros)
    fetch(FAST, DIAG, r_begin + 2 * SG_ROWS, r2v);
  }
  int buf = 0;
  for (long r0 = r_begin; r0 < r_end; r0 += SG_ROWS, buf ^= 1) {
    float *S = S2 + buf * (SG_ROWS * SG_LD);
    stash(DIAG, r0v, S);
#pragma unroll
    for (int u = 0; u < (decltype(DIAG)::value ? 2 : 4); ++u) { r0v[u] = r1v[u]; r1v[u] = r2v[u]; }
    fetch(FAST, DIAG, r0 + 3 * SG_ROWS, r2v);
    __syncthreads();   // slice visible; everybody has finished the MFMAs of the slice before (which read the OTHER buffer)
#pragma unroll
    for (int s = 0; s < SG_ROWS / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(S[oa + 2 * s * SG_LD], S[ob + 2 * s * SG_LD], acc, 0, 0, 0);
  }
#endif
  };
  if (fast) {
    if (diag) k_loop(std::true_type{}, std::true_type{}); else k_loop(std::true_type{}, std::false_type{});
  } else {
    if (diag) k_loop(std::false_type{}, std::true_type{}); else k_loop(std::false_type{}, std::false_type{});
  }

  // D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int jl = wj * 32 + li;
  if (pr.nchunk == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int il = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      const int gi = bi * SG_T + il, gj = bj * SG_T + jl;
      if (gi < dd && gj < dd) {
        // ONE value for the entry and its mirror image (C is symmetric on entry by contract: the mirrored entry's own old
        // value would be the same number, but a second expression could be contracted into a different fma)
        float *c = pr.C + (long)gi * pr.ldc + gj;
        const float v = pr.alpha * acc[r] + (pr.beta != 0.f ? pr.beta * *c : 0.f);
        *c = v;
        if (!diag) pr.C[(long)gj * pr.ldc + gi] = v;
      }
    }
    return;
  }
  float *mine = a.slab + (long)item * (SG_T * SG_T);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int il = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
    mine[il * SG_T + jl] = acc[r];
  }
  // publish: every store of the workgroup acknowledged, agent-scope release, then the arrival (returned)
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(a.cnt + pr.tile0 + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == (unsigned)(pr.nchunk - 1) ? 1u : 0u;
    if (s_last) {
      __hip_atomic_store(a.cnt + pr.tile0 + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  if (!s_last) return;
  // ---- finisher: the tile's slabs in chunk order (fixed order: deterministic), alpha / beta, mirror
  const float *first = a.slab + (long)(pr.item0 + tile * pr.nchunk) * (SG_T * SG_T);
  for (int e = tid; e < SG_T * SG_T / 4; e += 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < pr.nchunk; ++c) {
      typedef float __attribute__((ext_vector_type(4))) v4;
      const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(first + (long)c * (SG_T * SG_T)) + e);
      s.x += v[0]; s.y += v[1]; s.z += v[2]; s.w += v[3];
    }
    const int il = (4 * e) / SG_T, j0 = (4 * e) % SG_T;
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int gi = bi * SG_T + il, gj = bj * SG_T + j0 + k;
      if (gi < dd && gj < dd) {
        float *c = pr.C + (long)gi * pr.ldc + gj;
        const float v = pr.alpha * sv[k] + (pr.beta != 0.f ? pr.beta * *c : 0.f);
        *c = v;
        if (!diag) pr.C[(long)gj * pr.ldc + gi] = v;
      }
    }
  }
}

struct SgPlan {
  long chunk[SG_MAXP];
  int T[SG_MAXP], nchunk[SG_MAXP], item0[SG_MAXP], tile0[SG_MAXP];
  int nitems, ntiles;
};
static void sg_plan(int P, const long *rows, const int *d, const int *ones, SgPlan &pl) {
  double tile_rows = 0;
  for (int p = 0; p < P; ++p) {
    pl.T[p] = (int)cdiv(d[p] + (ones[p] ? 1 : 0), SG_T);
    tile_rows += (double)pl.T[p] * (pl.T[p] + 1) / 2 * (double)rows[p];
  }
  // ~4 work items per compute unit; at least 256 rows each; at most SG_MAX_CHUNKS chunks per tile
#ifndef CLO_SG_ITEMS_PER_CU
#define CLO_SG_ITEMS_PER_CU 4
#endif
  long common = (long)(tile_rows / ((double)CLO_SG_ITEMS_PER_CU * kNumCU));
  common = std::max<long>(256, cdiv(common, SG_ROWS) * SG_ROWS);
  pl.nitems = 0;
  pl.ntiles = 0;
  for (int p = 0; p < P; ++p) {
    long ch = std::max<long>(common, cdiv(cdiv(rows[p], SG_MAX_CHUNKS), SG_ROWS) * SG_ROWS);
    pl.chunk[p] = ch;
    pl.nchunk[p] = (int)std::max<long>(1, cdiv(rows[p], ch));
    pl.item0[p] = pl.nitems;
    pl.tile0[p] = pl.ntiles;
    const int nt = pl.T[p] * (pl.T[p] + 1) / 2;
    pl.nitems += nt * pl.nchunk[p];
    pl.ntiles += nt;
  }
}

}  // namespace clo

using namespace clo;

// Workspace: `*slab_floats` floats of partial tiles (uninitialised) and `*counters` unsigned counters that must be ZERO on
// entry (the kernel leaves them zero).  P <= clo_syrk_grouped_max_problems().
extern "C" int clo_syrk_grouped_max_problems(void) { return SG_MAXP; }
extern "C" int clo_syrk_grouped_ws(int P, const long *rows, const int *d, const int *ones_col, long *slab_floats,
                                   long *counters) {
  CLO_REQUIRE(P >= 1 && P <= SG_MAXP && rows && d && ones_col && slab_floats && counters,
              "clo_syrk_grouped_ws: needs 1 <= P <= %d problems", SG_MAXP);
  SgPlan pl;
  sg_plan(P, rows, d, ones_col, pl);
  *slab_floats = (long)pl.nitems * SG_T * SG_T;
  *counters = pl.ntiles;
  return CLO_OK;
}

extern "C" int clo_syrk_grouped_f32(int P, float *const *C, const long *ldc, const float *const *X, const long *rows,
                                    const int *d, const long *ldx, const int *ones_col, const float *alpha,
                                    const float *beta, float *slab, unsigned *counters, void *stream) {
  CLO_REQUIRE(P >= 1 && P <= SG_MAXP, "clo_syrk_grouped_f32: needs 1 <= P <= %d problems", SG_MAXP);
  CLO_REQUIRE(C && ldc && X && rows && d && ldx && ones_col && alpha && beta && slab && counters,
              "clo_syrk_grouped_f32: null argument");
  SgPlan pl;
  sg_plan(P, rows, d, ones_col, pl);
  SgArgs a{};
  a.P = P; a.nitems = pl.nitems; a.slab = slab; a.cnt = counters;
  for (int p = 0; p < P; ++p) {
    const int dd = d[p] + (ones_col[p] ? 1 : 0);
    CLO_REQUIRE(d[p] >= 0 && dd >= 1 && rows[p] >= 0 && C[p] && (X[p] || rows[p] == 0 || d[p] == 0) && ldc[p] >= dd &&
                    ldx[p] >= d[p],
                "clo_syrk_grouped_f32: bad operand %d", p);
    SgProb &q = a.pr[p];
    q.X = X[p]; q.C = C[p]; q.rows = rows[p]; q.ldx = ldx[p]; q.ldc = ldc[p]; q.chunk = pl.chunk[p];
    q.d = d[p]; q.ones = ones_col[p] ? 1 : 0; q.T = pl.T[p]; q.nchunk = pl.nchunk[p]; q.item0 = pl.item0[p];
    q.tile0 = pl.tile0[p];
    q.vec = (d[p] >= 4 && ldx[p] % 4 == 0 && aligned16(X[p])) ? 1 : 0;
    q.alpha = alpha[p]; q.beta = beta[p];
  }
  hipLaunchKernelGGL(syrk_grouped_kernel, dim3((unsigned)pl.nitems), dim3(256), 0, (hipStream_t)stream, a);
  CLO_CHECK_LAUNCH("syrk_grouped_kernel");
  return CLO_OK;
}
