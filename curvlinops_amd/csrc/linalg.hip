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
// ONE wavefront (barriers are free), lane i owns row i.  Left-looking: column k of L is
//   L[i][k] = (S[i][k] - sum_{j<k} L[i][j] L[k][j]) / L[k][k]
// with L[i][j] read from the lane's own LDS row (stride 65: conflict-free) and L[k][j] a broadcast
// read; the triangular inverse is forward substitution with lane c owning column c of L^-1.
// ~2 x 2016 LDS-fed FMAs per lane instead of 64 x 3 workgroup barriers.
__global__ __launch_bounds__(64) void potrf_diag_kernel(const float *Ain, long ldin, float *A,
                                                        long lda, int nb, float *__restrict__ Linv,
                                                        long ldinv, int *__restrict__ status,
                                                        int pivot_base) {
  __shared__ float S[PNB][PNB + 1];   // becomes L (lower triangle)
  __shared__ float XT[PNB][PNB + 1];  // XT[c][i] = (L^-1)[i][c]
  const int lane = threadIdx.x;
  for (int j = 0; j < nb; ++j)
    S[lane][j] = lane < nb ? Ain[(long)lane * ldin + j] : 0.f;
  __syncthreads();
  for (int k = 0; k < nb; ++k) {
    float acc = S[lane][k];
#pragma unroll 4
    for (int j = 0; j < k; ++j) acc = fmaf(-S[lane][j], S[k][j], acc);
    const float d = __shfl(acc, k, 64);
    if (!(d > 0.f)) {  // uniform; also catches NaN
      if (lane == 0) *status = pivot_base + k + 1;
      return;
    }
    const float inv = rsqrtf(d);
    __syncthreads();
    if (lane >= k && lane < nb) S[lane][k] = (lane == k) ? d * inv : acc * inv;
    __syncthreads();
  }
  // forward substitution for column c = lane of X = L^-1
  for (int i = 0; i < nb; ++i) {
    float acc = (lane == i) ? 1.f : 0.f;
#pragma unroll 4
    for (int k = 0; k < i; ++k) acc = fmaf(-S[i][k], XT[lane][k], acc);
    XT[lane][i] = (i < lane || lane >= nb) ? 0.f : acc / S[i][i];
  }
  __syncthreads();
  if (lane < nb) {
    for (int j = 0; j <= lane; ++j) A[(long)lane * lda + j] = S[lane][j];
    for (int c = 0; c < nb; ++c) Linv[(long)lane * ldinv + c] = XT[c][lane];
  }
}

// S = A + damping * I ; L = 0 ; Li = 0
__global__ void chol_init_kernel(const float *__restrict__ A, long lda, float *__restrict__ S,
                                 float *__restrict__ L, float *__restrict__ Li, int n, float damping) {
  const long total = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int i = e / n, j = e % n;
    S[e] = A[(long)i * lda + j] + (i == j ? damping : 0.f);
    L[e] = 0.f;
    Li[e] = 0.f;
  }
}

int launch_gemm_simple(int M, int N, int K, float alpha, const float *A, long sa_m, long sa_k,
                       const float *B, long sb_k, long sb_n, float beta, float *C, long ldc,
                       float *ws, long ws_floats, hipStream_t st);
int launch_syrk_simple(float *C, long ldc, const float *X, long rows, int d, long ldx, float alpha,
                       float beta, float *ws, long ws_floats, hipStream_t st);

struct CholCtx {
  float *S, *L, *Li, *T, *G;
  long gws;
  int n;
  int *status;
  hipStream_t st;
};

// Recursive blocked Cholesky carrying the inverse of the triangular factor:
//   L11, L11^-1 = rec(A11);  L21 = A21 L11^-T;  S22 -= L21 L21^T;  L22, L22^-1 = rec(S22);
//   (L^-1)21 = -L22^-1 (L21 L11^-1)
static int chol_rec(const CholCtx &c, int o, int m) {
  const long n = c.n;
  if (m <= PNB) {
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(64), 0, c.st, c.S + o * n + o, n,
                       c.L + o * n + o, n, m, c.Li + o * n + o, n, c.status, o);
    CLO_CHECK_LAUNCH("potrf_diag_kernel");
    return CLO_OK;
  }
  const int m1 = ((m / 2 + PNB - 1) / PNB) * PNB, m2 = m - m1;
  const int a = o, b = o + m1;
  int rc = chol_rec(c, a, m1);
  if (rc != CLO_OK) return rc;
  const float *L11i = c.Li + a * n + a;
  float *L21 = c.L + b * n + a;
  // L21 = S21 * L11i^T      (B(k,n) = L11i[n][k])
  rc = launch_gemm_simple(m2, m1, m1, 1.f, c.S + b * n + a, n, 1, L11i, 1, n, 0.f, L21, n, c.G, c.gws, c.st);
  if (rc != CLO_OK) return rc;
  // S22 -= L21 * L21^T
  rc = launch_gemm_simple(m2, m2, m1, -1.f, L21, n, 1, L21, 1, n, 1.f, c.S + b * n + b, n, c.G, c.gws, c.st);
  if (rc != CLO_OK) return rc;
  rc = chol_rec(c, b, m2);
  if (rc != CLO_OK) return rc;
  // T = L21 * L11i ; Li21 = -L22i * T
  rc = launch_gemm_simple(m2, m1, m1, 1.f, L21, n, 1, L11i, n, 1, 0.f, c.T, m1, c.G, c.gws, c.st);
  if (rc != CLO_OK) return rc;
  return launch_gemm_simple(m2, m1, m2, -1.f, c.Li + b * n + b, n, 1, c.T, m1, 1, 0.f, c.Li + b * n + a, n,
                            c.G, c.gws, c.st);
}

}  // namespace clo

using namespace clo;

extern "C" long clo_cholesky_inverse_ws_floats(int n) {
  const long nn = (long)n * n;
  return 3 * nn + nn / 2 + n + 16L * 256 * 256 + 1024;  // S, L, Li, T, split-K slabs
}

// out = (A + damping I)^-1, A symmetric positive definite n x n (row-major, lda), out row-major ldo.
// ws: clo_cholesky_inverse_ws_floats(n) floats; *status (device int, zeroed here) = offending pivot.
extern "C" int clo_cholesky_inverse_f32(const float *A, long lda, float *out, long ldo, int n,
                                        float damping, float *ws, int *status, void *stream) {
  CLO_REQUIRE(n >= 0 && lda >= n && ldo >= n, "clo_cholesky_inverse_f32: bad sizes");
  if (n == 0) return CLO_OK;
  CLO_REQUIRE(A && out && ws && status, "clo_cholesky_inverse_f32: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int rc = check_hip(hipMemsetAsync(status, 0, sizeof(int), st), "hipMemsetAsync");
  if (rc != CLO_OK) return rc;
  const long nn = (long)n * n;
  CholCtx c;
  c.S = ws; c.L = ws + nn; c.Li = ws + 2 * nn; c.T = ws + 3 * nn;
  c.G = c.T + nn / 2 + n;
  c.gws = 16L * 256 * 256;
  c.n = n; c.status = status; c.st = st;
  hipLaunchKernelGGL(chol_init_kernel, dim3((unsigned)std::min<long>(cdiv(nn, 256), kNumCU * 8L)),
                     dim3(256), 0, st, A, lda, c.S, c.L, c.Li, n, damping);
  CLO_CHECK_LAUNCH("chol_init_kernel");
  rc = chol_rec(c, 0, n);
  if (rc != CLO_OK) return rc;
  // A^-1 = Li^T Li
  return launch_syrk_simple(out, ldo, c.Li, n, n, n, 1.f, 0.f, c.G, c.gws, st);
}

extern "C" int clo_potrf_diag_f32(float *A, long lda, int nb, float *Linv, long ldinv, int *status,
                                  int pivot_base, void *stream) {
  CLO_REQUIRE(nb >= 1 && nb <= PNB, "clo_potrf_diag_f32: nb must be in [1, %d], got %d", PNB, nb);
  CLO_REQUIRE(A && Linv && status && lda >= nb && ldinv >= nb, "clo_potrf_diag_f32: bad operand");
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, lda, A, lda,
                     nb, Linv, ldinv, status, pivot_base);
  CLO_CHECK_LAUNCH("potrf_diag_kernel");
  return CLO_OK;
}
