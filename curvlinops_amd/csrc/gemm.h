// Argument block of the fp32 MFMA GEMM engines in gemm.hip (shared with the MLP large-batch path).
#pragma once
#include "clo_common.h"

namespace clo {

constexpr int GEMM_TAB_MAX = 8;   // batch members of a pointer-table launch

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
  int tbm, tbn;  // block tile extents of the engine that runs (set by launch_gemm)
  int nbatch, batch_per_split;  // SQSUM mode only
  int n_mem;  // SQSUM mode only: floats per k row of B that exist in memory (0: N; else a multiple of 4 >= N)
  int ones;  // 1: outer index M-1 of A / N-1 of B is an implicit column of ones ([X | 1])
  // fused epilogue (single problem only), applied by whichever kernel writes the final C:
  //   EPI_ACT: v = act(v + e_vec[col]) ; e_out2[row][col] = act'      (e_out2 shares ldc)
  //   EPI_MUL: v = (v + e_vec[col]) * e_mul[row * ld_mul + col]
  //   EPI_MUL_T: v = v * e_mul[(col / e_div) * ld_mul + row]   (mask stored sample-major, C is
  //              feature-major with e_div columns per sample)
  int epi, e_act, e_div;
  const float *e_vec, *e_mul;
  long ld_mul;
  float *e_out2;
  // second K segment (v2 engine only): for k >= K1 the operands are A2 / B2 (same strides),
  // i.e. C = A[:, :K1] B[:K1] + A2 B2 with K = K1 + K2 in one pass.  K1 % 32 == 0.
  const float *A2, *B2;
  int K1;
  // triangular-operand hint (v2 engine; other engines ignore it, the skipped products are zeros):
  // per output tile only the k range that can be nonzero is visited
  //   TRI_KGE_M: A(m, k) == 0 for k < m     TRI_KLT_M: A(m, k) == 0 for k > m
  //   TRI_KGE_N: B(k, n) == 0 for k < n     TRI_KLT_N: B(k, n) == 0 for k > n
  int tri;
  // B-side implicit ones column only (outer index N-1 of B; A has no extra row), and a separate
  // destination for that output column: C(:, N-1) goes to col_out[row] (same alpha / beta) while
  // the first N-1 columns keep ldc.  With B = layer inputs [rows][d_in] and A = delta^T this is the
  // weight gradient with the BIAS gradient (column sums of delta) as its extra column, one launch.
  // v2 engine only (launch_gemm returns CLO_EUNSUP otherwise).
  int ones_b;
  float *col_out;
  // patch mode (clo_im2col_syrk_accum_f32; v2 engine, both operands outer-contiguous): A and B are the
  // im2col matrix of a [B][C][H][W] tensor, X[(b, oh, ow)][(c, kh, kw)], generated in the tile loader
  // -- it is never written to memory.  A == B == the input tensor; M == N == C*KH*KW (+ ones).
  int patch;
  // stream-K request (LDS-DMA engine): ws holds gemm_streamk_ws_floats() floats (besides whatever splitk needs); the
  // engine decides whether the schedule pays (then no split-K reduction runs)
  int streamk;   // 0: no; 1: ws >= gemm_streamk_ws_floats_square(); 2: ws >= gemm_streamk_ws_floats()
  int cvC, cvH, cvW, cvKH, cvKW, cvSH, cvSW, cvPH, cvPW, cvDH, cvDW, cvOH, cvOW;
  // batch members at arbitrary addresses (clo_gemm_ptrs_f32, round 5): with tab_a / tab_b set, matrix b of that operand
  // starts at A + off_a[b] / B + off_b[b] floats (differences to member 0, any sign) instead of b * sa_b / b * sb_b --
  // equal-shape Kronecker factors of different layers enter ONE batched launch where they lie, no stacked copies.
  int tab_a, tab_b;
  long off_a[GEMM_TAB_MAX], off_b[GEMM_TAB_MAX];
};
__host__ __device__ inline long gemm_off_a(const GemmArgs &p, int b) { return p.tab_a ? p.off_a[b] : (long)b * p.sa_b; }
__host__ __device__ inline long gemm_off_b(const GemmArgs &p, int b) { return p.tab_b ? p.off_b[b] : (long)b * p.sb_b; }
enum { EPI_NONE = 0, EPI_ACT = 1, EPI_MUL = 2, EPI_MUL_T = 3 };
enum { TRI_KGE_M = 1, TRI_KLT_M = 2, TRI_KGE_N = 4, TRI_KLT_N = 8 };

// final value of C[row][col] from the accumulated product `acc`
__device__ __forceinline__ void store_final(const GemmArgs &p, float *c, int row, int col, float acc,
                                            float alpha, float beta) {
  float v = alpha * acc;
  if (beta != 0.f) v += beta * *c;
  if (p.epi == EPI_ACT) {
    float dphi;
    v = act_apply(p.e_act, v + (p.e_vec ? p.e_vec[col] : 0.f), dphi);
    if (p.e_out2) p.e_out2[(c - p.C)] = dphi;
  } else if (p.epi == EPI_MUL) {
    v = (v + (p.e_vec ? p.e_vec[col] : 0.f)) * p.e_mul[(long)row * p.ld_mul + col];
  } else if (p.epi == EPI_MUL_T) {
    v *= p.e_mul[(long)(col / p.e_div) * p.ld_mul + row];
  }
  *c = v;
}

using f32x4w = __attribute__((ext_vector_type(4))) float;

// Wide epilogue (round 6, both MFMA engines): a finished tile staged in LDS as R rows x CC columns (row pitch `pitch` floats, a multiple of 4) goes
// out as 16-byte stores, one full 512-byte (CC = 128) or 256-byte (CC = 64) row segment per 32 / 16 lanes; C(row, col) =
// epilogue(alpha t + beta C) with the fused epilogues of GemmArgs (store_final in gemm.h, same arithmetic).  (r0, c0) = the
// tile's origin in C, rows >= rmax / columns >= cmax are outside the matrix.
struct V3Epi {
  int kind, act, div;
  const float *vec, *mul;
  long ld_mul;
  float *out2;
  const float *Cbase;
};
__device__ __forceinline__ float v3_epi_one(const V3Epi &E, float v, int row, int col, const float *c) {
  if (E.kind == EPI_ACT) {
    float dphi;
    v = act_apply(E.act, v + (E.vec ? E.vec[col] : 0.f), dphi);
    if (E.out2) E.out2[c - E.Cbase] = dphi;
  } else if (E.kind == EPI_MUL) {
    v = (v + (E.vec ? E.vec[col] : 0.f)) * E.mul[(long)row * E.ld_mul + col];
  } else if (E.kind == EPI_MUL_T) {
    v *= E.mul[(long)(col / E.div) * E.ld_mul + row];
  }
  return v;
}
template <int R, int CC, int NTHR>
__device__ __forceinline__ void v3_store_rows(const float *T, int pitch, float *C, long ldc, int r0, int c0, int rmax, int cmax,
                                              float alpha, float beta, int tid, const V3Epi &E) {
  constexpr int V = CC / 4;
  static_assert((R * V) % NTHR == 0, "whole passes");
#pragma unroll
  for (int it = 0; it < R * V / NTHR; ++it) {
    const int idx = it * NTHR + tid, i = idx / V, c4 = idx % V;
    const int row = r0 + i, col = c0 + 4 * c4;
    if (row >= rmax || col >= cmax) continue;
    const f32x4w t = *reinterpret_cast<const f32x4w *>(T + i * pitch + 4 * c4);
    float *c = C + (long)row * ldc + col;
    if (col + 3 < cmax) {
      f32x4w v;
      v[0] = alpha * t[0]; v[1] = alpha * t[1]; v[2] = alpha * t[2]; v[3] = alpha * t[3];
      if (beta != 0.f) {
        const f32x4w o = *reinterpret_cast<const f32x4w *>(c);
        v[0] += beta * o[0]; v[1] += beta * o[1]; v[2] += beta * o[2]; v[3] += beta * o[3];
      }
      if (E.kind == EPI_ACT) {
        f32x4w d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dphi;
          v[e] = act_apply(E.act, v[e] + (E.vec ? E.vec[col + e] : 0.f), dphi);
          d[e] = dphi;
        }
        if (E.out2) *reinterpret_cast<f32x4w *>(E.out2 + (c - E.Cbase)) = d;   // (16-byte aligned: checked by the caller)
      } else if (E.kind == EPI_MUL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] + (E.vec ? E.vec[col + e] : 0.f)) * E.mul[(long)row * E.ld_mul + col + e];
      } else if (E.kind == EPI_MUL_T) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= E.mul[(long)((col + e) / E.div) * E.ld_mul + row];
      }
      *reinterpret_cast<f32x4w *>(c) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (col + e < cmax) {
          float v = alpha * t[e];
          if (beta != 0.f) v += beta * c[e];
          c[e] = v3_epi_one(E, v, row, col + e, c + e);
        }
    }
  }
}

// The 32 x 32 MFMA accumulators of a BMt x BNt tile (wave (wm, wn) holds WM x WNC of it as MT x NT blocks; lane (li, lh), register r
// = row (r & 3) + 8 (r >> 2) + 4 lh, column li of a block) into LDS, row-major [BMt][BNt + 4] ...
template <int MT, int NT, int WM, int WNC, int PD, typename Acc>
__device__ __forceinline__ void stage_tile_direct(float *T, const Acc &acc, int wm, int wn, int li, int lh) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        T[(wm * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PD + wn * WNC + nt * 32 + li] = acc[mt][nt][r];
}
// ... and transposed, [BNt][BMt + 4] (the mirror image of a symmetric product): four consecutive rows of a lane are one 16-byte
// LDS write
template <int MT, int NT, int WM, int WNC, int PM, typename Acc>
__device__ __forceinline__ void stage_tile_mirror(float *T, const Acc &acc, int wm, int wn, int li, int lh) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4w v;
        v[0] = acc[mt][nt][4 * q]; v[1] = acc[mt][nt][4 * q + 1]; v[2] = acc[mt][nt][4 * q + 2]; v[3] = acc[mt][nt][4 * q + 3];
        *reinterpret_cast<f32x4w *>(T + (wn * WNC + nt * 32 + li) * PM + wm * WM + mt * 32 + 8 * q + 4 * lh) = v;
      }
}

bool gemm_v2_eligible(const GemmArgs &a, int batch);
// LDS-DMA engine (gemm_v3.hip): 128 x 128 x 32 tiles, optional stream-K schedule
bool gemm_v3_eligible(const GemmArgs &a, int batch);
bool gemm_v3_would_streamk(int M, int N, int K, long batch);
bool gemm_v3_small(const GemmArgs &a, int batch);   // the 64 x 64 tile configuration of the LDS-DMA engine serves this problem
long gemm_streamk_ws_floats();          // any tile configuration
long gemm_streamk_ws_floats_square();   // 128 x 128 tiles only
int launch_gemm_v3(const GemmArgs &a, int batch, bool a_kc, bool b_kc, hipStream_t stream, bool *used_streamk,
                   int *tile_m = nullptr, int *tile_n = nullptr);   // tile_m / tile_n: block tile extents of the configuration that ran
int launch_gemm(GemmArgs a, int batch, hipStream_t stream);
// C = beta C + alpha sum_b (A_b B_b)^2 (elementwise square), the members split over `splits` slabs in a.ws (a.n_mem: see GemmArgs)
int launch_gemm_sqsum(GemmArgs a, int batch, int splits, hipStream_t stream);
int launch_gemm_auto(GemmArgs a, float *ws, long ws_floats, hipStream_t st, int batch = 1);
int launch_mlp_fwd3(const float *A, const float *dA, const float *W, const float *V, const float *b,
                    const float *Vb, float *a, float *da, float *dphi, int N, int d_in, int d_out,
                    int act, float *ws, long ws_floats, hipStream_t st);

}  // namespace clo
