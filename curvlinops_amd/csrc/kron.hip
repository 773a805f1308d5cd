// Kronecker-factored blocks behind ONE foreign call (SURVEY 8b export list; round 4):
//   clo_kron_matmat        Y_k = S1 X_k S2^T  (per factor: or its transpose)       reference kronecker.py:141-171 (einsum over the factors)
//   clo_eigh_apply         Y_k = Q1 (lam .* (Q1^T X_k Q2)) Q2^T      reference eigh.py:84-105 with a Kronecker eigenbasis
//   clo_kron_matmat_blocks every block of a block-diagonal KFAC / EKFAC operator in one call (reference
//                          block_diagonal.py: loop over blocks; kfac.py / ekfac.py build one such block per layer)
//   clo_ekfac_correction_f32  a layer's eigenvalue correction lam += alpha sum_{v,n} (Qg^T (sum_s g_vns a_ns^T) Qa)^2 from
//                          its layer inputs and output gradients: both rotations + the fused squared-product kernel
//                          (reference computers/ekfac_hooks.py:25-238)
// The operand is K-major: X [K][a*b], column k a row-major [a, b] matrix, and so is the result -- the layout the canonical
// converters (clo_canonical_pack_f32) produce and consume, for K == 1 simply the vector.  Per block two products on the MFMA
// GEMM engine: T = [X_0; ...; X_{K-1}] S2^T as ONE product with K a rows, then Y_k = S1 T_k batched over k.  Everything is
// queued on the caller's stream; the rounds 1-3 composition of these products in Python (kronecker.py, one ctypes call and
// one torch allocation per product) stays as the path for operands in other layouts.
#include <algorithm>

#include "../../include/curvlinops_amd.h"
#include "clo_common.h"
#include "gemm.h"

namespace clo {
namespace {

struct Fac {   // a factor as the effective matrix E [rows][cols] with element strides
  const float *p;
  int rows, cols;
  long sr, sc;
};
inline Fac fac(const float *S, int R, int C, long ld, bool trans) {
  return trans ? Fac{S, C, R, 1, ld} : Fac{S, R, C, ld, 1};   // S is row-major [R][C] with leading dimension ld >= C
}

__global__ void kron_scale_kernel(float *__restrict__ Z, const float *__restrict__ lam, long n, int K) {
  const long total = n * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) Z[e] *= lam[e % n];
}

inline long pad4k(long x) { return (x + 3) & ~3L; }
// dst [rows][ldd] = src (element (r, c) at src[r sr + c sc]) for c < cols, zero for cols <= c < ldd; rows >= rows_src are zero
__global__ void kron_pad_kernel(float *__restrict__ dst, long ldd, long rows, const float *__restrict__ src, long sr, long sc,
                                long rows_src, long cols) {
  const long q4 = ldd >> 2, total = rows * q4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / q4, c = (e - r * q4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows_src) {
      const float *p = src + r * sr + c * sc;
      if (c < cols) v.x = p[0];
      if (c + 1 < cols) v.y = p[sc];
      if (c + 2 < cols) v.z = p[2 * sc];
      if (c + 3 < cols) v.w = p[3 * sc];
    }
    *reinterpret_cast<float4 *>(dst + r * ldd + c) = v;
  }
}
// dst [rows][cols] (contiguous) = src [rows][lds][:cols]
__global__ void kron_unpad_kernel(float *__restrict__ dst, long cols, long rows, const float *__restrict__ src, long lds) {
  const long total = rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / cols, c = e - r * cols;
    dst[e] = src[r * lds + c];
  }
}
// Second factors of an order that is not a multiple of 4 (joint weight + bias blocks: d_in + 1) put both products on the
// scalar-loader kernel (512 x 4609 x 4609: 382 us, 178 us padded to 4612).  From KR_PAD_MIN_FLOP on, the operands are
// copied into zero-padded buffers with rows of pad4 floats, both products run on the aligned engines and the result is
// copied back: three extra passes over the (live) factor per product instead of a value cache.
constexpr double KR_PAD_MIN_FLOP = 2.0e8;
inline bool kron_wants_pad(int K, int C1, int C2, int R2) {
  return ((C2 & 3) || (R2 & 3)) && 2.0 * K * C1 * (double)C2 * R2 >= KR_PAD_MIN_FLOP;
}
inline long kron_pad_floats(int K, int C1, int R1, int C2, int R2) {   // S2p, Xp, Tp, Yp
  return pad4k(R2) * pad4k(C2) + (long)K * C1 * pad4k(C2) + (long)K * C1 * pad4k(R2) + (long)K * R1 * pad4k(R2);
}

// Y_k [R1, R2] = E1 X_k E2^T for k < K;  X [K][C1*C2], Y [K][R1*R2], T: K*C1*R2 floats
int kron_apply(float *Y, const Fac &E1, const Fac &E2, const float *X, int K, float *T, float *gws, long gws_floats,
               hipStream_t st) {
  const int C1 = E1.cols, C2 = E2.cols, R1 = E1.rows, R2 = E2.rows;
#ifndef CLO_KRON_PAD
#define CLO_KRON_PAD 1
#endif
  static const int pad_on = CLO_KRON_PAD;
  if (pad_on && kron_wants_pad(K, C1, C2, R2) && gws_floats >= kron_pad_floats(K, C1, R1, C2, R2) + (1L << 20)) {
    const long C2p = pad4k(C2), R2p = pad4k(R2);
    float *S2p = gws, *Xp = S2p + R2p * C2p, *Tp = Xp + (long)K * C1 * C2p, *Yp = Tp + (long)K * C1 * R2p;
    float *g2 = Yp + (long)K * R1 * R2p;
    const long g2_floats = gws_floats - (g2 - gws);
    auto grid = [](long n) { return dim3((unsigned)std::min<long>(cdiv(n, 256), 8192)); };
    hipLaunchKernelGGL(kron_pad_kernel, grid(R2p * (C2p >> 2)), dim3(256), 0, st, S2p, C2p, R2p, E2.p, E2.sr, E2.sc, (long)R2, (long)C2);
    CLO_CHECK_LAUNCH("kron_pad_kernel");
    hipLaunchKernelGGL(kron_pad_kernel, grid((long)K * C1 * (C2p >> 2)), dim3(256), 0, st, Xp, C2p, (long)K * C1, X, (long)C2, 1L,
                       (long)K * C1, (long)C2);
    CLO_CHECK_LAUNCH("kron_pad_kernel");
    {
      GemmArgs g{};
      g.M = K * C1; g.N = (int)R2p; g.K = (int)C2p; g.alpha = 1.f; g.beta = 0.f;
      g.A = Xp; g.sa_m = C2p; g.sa_k = 1;
      g.B = S2p; g.sb_k = 1; g.sb_n = C2p;            // B(k = c2, n = r2) = S2p[r2][c2]
      g.C = Tp; g.ldc = R2p;
      int rc = launch_gemm_auto(g, g2, g2_floats, st, 1);
      if (rc != CLO_OK) return rc;
    }
    {
      GemmArgs g{};
      g.M = R1; g.N = (int)R2p; g.K = C1; g.alpha = 1.f; g.beta = 0.f;
      g.A = E1.p; g.sa_m = E1.sr; g.sa_k = E1.sc; g.sa_b = 0;
      g.B = Tp; g.sb_k = R2p; g.sb_n = 1; g.sb_b = (long)C1 * R2p;
      g.C = Yp; g.ldc = R2p; g.sc_b = (long)R1 * R2p;
      int rc = launch_gemm_auto(g, g2, g2_floats, st, K);
      if (rc != CLO_OK) return rc;
    }
    hipLaunchKernelGGL(kron_unpad_kernel, grid((long)K * R1 * R2), dim3(256), 0, st, Y, (long)R2, (long)K * R1, Yp, R2p);
    CLO_CHECK_LAUNCH("kron_unpad_kernel");
    return CLO_OK;
  }
  {
    GemmArgs g{};
    g.M = K * C1; g.N = R2; g.K = C2; g.alpha = 1.f; g.beta = 0.f;
    g.A = X; g.sa_m = C2; g.sa_k = 1;
    g.B = E2.p; g.sb_k = E2.sc; g.sb_n = E2.sr;       // B(k = c2, n = r2) = E2[r2][c2]
    g.C = T; g.ldc = R2;
    int rc = launch_gemm_auto(g, gws, gws_floats, st, 1);
    if (rc != CLO_OK) return rc;
  }
  GemmArgs g{};
  g.M = R1; g.N = R2; g.K = C1; g.alpha = 1.f; g.beta = 0.f;
  g.A = E1.p; g.sa_m = E1.sr; g.sa_k = E1.sc; g.sa_b = 0;
  g.B = T; g.sb_k = R2; g.sb_n = 1; g.sb_b = (long)C1 * R2;
  g.C = Y; g.ldc = R2; g.sc_b = (long)R1 * R2;
  return launch_gemm_auto(g, gws, gws_floats, st, K);
}

constexpr long KR_GWS = 4L << 20;   // floats of split-K workspace behind the temporaries

long block_ws(int A, int a, int B, int b, int K, bool eig) {
  // T of the (larger) first product, and for eigen-decomposed blocks the coefficient block Z
  const long t = (long)K * std::max<long>((long)std::max(A, a) * std::max(B, b), 1);
  return (eig ? 2 : 1) * ((t + 3) & ~3L);
}
// floats behind the temporaries: split-K workspace, and the zero-padded operands of blocks whose second factor has an order
// that is not a multiple of 4 (kron_apply; either orientation of the factors)
long block_gws(int A, int a, int B, int b, int K) {
  const int m1 = std::max(A, a), m2 = std::max(B, b);
  const bool pad = ((B & 3) || (b & 3)) && 2.0 * K * m1 * (double)m2 * m2 >= KR_PAD_MIN_FLOP;
  return KR_GWS + (pad ? kron_pad_floats(K, m1, m1, m2, m2) + (1L << 20) : 0);
}

int one_block(float *Y, const float *S1, long ld1, const float *S2, long ld2, const float *lam, const float *X, int A, int a,
              int B, int b, int K, int trans, float *ws, long ws_floats, hipStream_t st) {
  CLO_REQUIRE(ld1 >= a && ld2 >= b, "clo_kron: leading dimensions %ld / %ld below the factor widths %d / %d", ld1, ld2, a, b);
  const long need = block_ws(A, a, B, b, K, lam != nullptr);
  CLO_REQUIRE(ws_floats >= need + 1024, "clo_kron: workspace too small (%ld < %ld floats)", ws_floats, need + 1024);
  float *T = ws, *gws = ws + need;
  const long gws_floats = ws_floats - need;
  const bool t1 = (trans & 1) != 0, t2 = (trans & 2) != 0;   // per factor: the array holds the transposed factor
  if (!lam) return kron_apply(Y, fac(S1, A, a, ld1, t1), fac(S2, B, b, ld2, t2), X, K, T, gws, gws_floats, st);
  // eigen-decomposed block: S1 = Q1 [A, A], S2 = Q2 [B, B] (a == A, b == B), eigenvectors in the COLUMNS; a factor's trans bit:
  // in the ROWS (what clo_eigh_f32 returns), i.e. the array holds Q^T
  float *Z = ws + need / 2;
  int rc = kron_apply(Z, fac(S1, A, A, ld1, !t1), fac(S2, B, B, ld2, !t2), X, K, T, gws, gws_floats, st);   // Z = Q1^T X Q2
  if (rc != CLO_OK) return rc;
  const long n = (long)A * B;
  hipLaunchKernelGGL(kron_scale_kernel, dim3((unsigned)std::min<long>(cdiv(n * K, 256), 2048)), dim3(256), 0, st, Z, lam, n, K);
  CLO_CHECK_LAUNCH("kron_scale_kernel");
  return kron_apply(Y, fac(S1, A, A, ld1, t1), fac(S2, B, B, ld2, t2), Z, K, T, gws, gws_floats, st);      // Y = Q1 Z Q2^T
}

}  // namespace
}  // namespace clo

using namespace clo;

extern "C" long clo_kron_ws_floats(int A, int a, int B, int b, int K, int eig) {
  if (A < 1 || a < 1 || B < 1 || b < 1 || K < 1) return 0;
  return block_ws(A, a, B, b, K, eig != 0) + block_gws(A, a, B, b, K);
}

extern "C" int clo_kron_matmat(float *Y, const float *S1, long ld1, const float *S2, long ld2, const float *X, int A, int a,
                               int B, int b, int K, int trans, float *ws, long ws_floats, void *stream) {
  CLO_REQUIRE(Y && S1 && S2 && X && ws, "clo_kron_matmat: null operand");
  CLO_REQUIRE(A >= 1 && a >= 1 && B >= 1 && b >= 1 && K >= 1 && (trans & ~3) == 0, "clo_kron_matmat: bad extents / flags");
  return one_block(Y, S1, ld1, S2, ld2, nullptr, X, A, a, B, b, K, trans, ws, ws_floats, (hipStream_t)stream);
}

extern "C" int clo_eigh_apply(float *Y, const float *Q1, long ld1, const float *Q2, long ld2, const float *lam, const float *X,
                              int n1, int n2, int K, int rows, float *ws, long ws_floats, void *stream) {
  CLO_REQUIRE(Y && Q1 && Q2 && lam && X && ws, "clo_eigh_apply: null operand");
  CLO_REQUIRE(n1 >= 1 && n2 >= 1 && K >= 1, "clo_eigh_apply: bad extents");
  return one_block(Y, Q1, ld1, Q2, ld2, lam, X, n1, n1, n2, n2, K, rows, ws, ws_floats, (hipStream_t)stream);
}

extern "C" int clo_kron_matmat_blocks(int nblocks, float *const *Y, const float *const *S1, const long *ld1,
                                      const float *const *S2, const long *ld2, const float *const *lam,
                                      const float *const *X, const int *A, const int *a, const int *B, const int *b,
                                      const int *trans, int K, float *ws, long ws_floats, void *stream) {
  CLO_REQUIRE(nblocks >= 0 && K >= 1, "clo_kron_matmat_blocks: bad extents");
  CLO_REQUIRE(nblocks == 0 || (Y && S1 && ld1 && S2 && ld2 && X && A && a && B && b && ws), "clo_kron_matmat_blocks: null operand");
  for (int i = 0; i < nblocks; ++i) {
    CLO_REQUIRE(Y[i] && S1[i] && S2[i] && X[i] && A[i] >= 1 && a[i] >= 1 && B[i] >= 1 && b[i] >= 1,
                "clo_kron_matmat_blocks: block %d has a null operand or an empty extent", i);
    const float *l = lam ? lam[i] : nullptr;
    CLO_REQUIRE(!l || (A[i] == a[i] && B[i] == b[i]), "clo_kron_matmat_blocks: eigen-decomposed block %d is not square", i);
    int rc = one_block(Y[i], S1[i], ld1[i], S2[i], ld2[i], l, X[i], A[i], a[i], B[i], b[i], K, trans ? trans[i] : 0, ws, ws_floats,
                       (hipStream_t)stream);
    if (rc != CLO_OK) return rc;
  }
  return CLO_OK;
}

// ---- EKFAC eigenvalue correction of one layer in one call ------------------------------------------------------------
namespace clo {
namespace {
// g2[n][i] = sum_v g[v][n][i]^2 ; a2 = a^2 (S == 1: the squared per-example gradient is the outer product of the squares)
__global__ void ekfac_square_kernel(const float *__restrict__ g, float *__restrict__ g2, long nb_d1, int V,
                                    const float *__restrict__ a, float *__restrict__ a2, long nb_d2) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nb_d1) {
    float sacc = 0.f;
    for (int v = 0; v < V; ++v) {
      const float x = g[(long)v * nb_d1 + e];
      sacc += x * x;
    }
    g2[e] = sacc;
  }
  if (e < nb_d2) a2[e] = a[e] * a[e];
}
inline long pad4l(long x) { return (x + 3) & ~3L; }
}  // namespace
}  // namespace clo

extern "C" long clo_ekfac_correction_ws_floats(int V, int B, int S, int d_out, int d_in) {
  if (V < 1 || B < 1 || S < 1 || d_out < 1 || d_in < 1) return 0;
  const long rot = pad4l((long)V * B * S * d_out) + (long)B * S * pad4l(d_in);   // (rotated inputs: rows padded to 4 floats)
  const long sq = S == 1 ? pad4l((long)B * d_out) + pad4l((long)B * d_in) : 0;
  const long splits = S == 1 ? 0 : (long)clo_gemm_sqsum_suggest_splits(d_out, d_in, B) * d_out * d_in;
  return rot + sq + pad4l(splits) + KR_GWS;
}

extern "C" int clo_ekfac_correction_f32(float *lam, long ld_lam, const float *Qg, long ldg, const float *Qa, long lda,
                                        int rows, const float *g, const float *a, int V, int B, int S, int d_out,
                                        int d_in, float alpha, float beta, float *ws, long ws_floats, void *stream) {
  CLO_REQUIRE(lam && Qg && Qa && g && a && ws, "clo_ekfac_correction_f32: null operand");
  CLO_REQUIRE(V >= 1 && B >= 1 && S >= 1 && d_out >= 1 && d_in >= 1 && ld_lam >= d_in && ldg >= d_out && lda >= d_in &&
                  (rows & ~3) == 0,
              "clo_ekfac_correction_f32: bad extents / flags");
  CLO_REQUIRE(ws_floats >= clo_ekfac_correction_ws_floats(V, B, S, d_out, d_in), "clo_ekfac_correction_f32: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // rotated inputs with rows of lda_r = d_in rounded up to 4 floats (S > 1): the squared-product kernel's float4 loader
  // then serves the joint weight + bias blocks too, whose width d_in + 1 is odd (it may read the pad, never stores it)
  const int lda_r = S == 1 ? d_in : (int)pad4l(d_in);
  const long ng = (long)V * B * S * d_out, na = (long)B * S * pad4l(d_in);
  float *g_rot = ws, *a_rot = g_rot + pad4l(ng), *p = a_rot + na;
  float *g2 = nullptr, *a2 = nullptr, *sws = nullptr;
  int splits = 1;
  if (S == 1) {
    g2 = p; p += pad4l((long)B * d_out);
    a2 = p; p += pad4l((long)B * d_in);
  } else {
    splits = clo_gemm_sqsum_suggest_splits(d_out, d_in, B);
    sws = p; p += pad4l((long)splits * d_out * d_in);
  }
  float *gws = p;
  const long gws_floats = ws_floats - (gws - ws);
  // rotations into the eigenbases: g_rot = g Qg, a_rot = a Qa (arrays that hold the eigenvectors in their rows: Q = array^T)
  auto rotate = [&](const float *X, long nrows, int d, const float *Q, long ldq, bool qrows, float *out, long ld_out) {
    GemmArgs m{};
    m.M = (int)nrows; m.N = d; m.K = d; m.alpha = 1.f; m.beta = 0.f;
    m.A = X; m.sa_m = d; m.sa_k = 1;
    m.B = Q; m.sb_k = qrows ? 1 : ldq; m.sb_n = qrows ? ldq : 1;
    m.C = out; m.ldc = ld_out;
    return launch_gemm_auto(m, gws, gws_floats, st, 1);
  };
  CLO_REQUIRE((long)V * B * S <= 2147483647L, "clo_ekfac_correction_f32: more than 2^31 - 1 gradient rows");
  int rc = rotate(g, (long)V * B * S, d_out, Qg, ldg, (rows & 1) != 0, g_rot, d_out);
  if (rc != CLO_OK) return rc;
  rc = rotate(a, (long)B * S, d_in, Qa, lda, (rows & 2) != 0, a_rot, lda_r);
  if (rc != CLO_OK) return rc;
  if (S == 1) {   // sum_{v,n} (g_vn a_n^T)^2 = (sum_v g_vn^2)^T (a_n^2): ONE product with K = B
    const long n1 = (long)B * d_out, n2 = (long)B * d_in;
    hipLaunchKernelGGL(ekfac_square_kernel, dim3((unsigned)cdiv(std::max(n1, n2), 256)), dim3(256), 0, st, g_rot, g2, n1, V,
                       a_rot, a2, n2);
    CLO_CHECK_LAUNCH("ekfac_square_kernel");
    GemmArgs m{};
    m.M = d_out; m.N = d_in; m.K = B; m.alpha = alpha; m.beta = beta;
    m.A = g2; m.sa_m = 1; m.sa_k = d_out;
    m.B = a2; m.sb_k = d_in; m.sb_n = 1;
    m.C = lam; m.ldc = ld_lam;
    return launch_gemm_auto(m, gws, gws_floats, st, 1);
  }
  for (int v = 0; v < V; ++v) {   // per-example products P_n = g_rot_n^T a_rot_n [d_out, d_in], squared and summed over n, fused
    GemmArgs q{};
    q.M = d_out; q.N = d_in; q.K = S; q.alpha = alpha; q.beta = v == 0 ? beta : 1.f;
    q.A = g_rot + (long)v * B * S * d_out; q.sa_m = 1; q.sa_k = d_out; q.sa_b = (long)S * d_out;
    q.B = a_rot; q.sb_k = lda_r; q.sb_n = 1; q.sb_b = (long)S * lda_r;
    q.C = lam; q.ldc = ld_lam; q.sc_b = 0; q.ws = splits > 1 ? sws : nullptr;
    q.n_mem = lda_r;
    rc = launch_gemm_sqsum(q, B, splits, st);
    if (rc != CLO_OK) return rc;
  }
  return CLO_OK;
}
