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

// In place: lower Cholesky factor of the nb x nb block at A (leading dimension lda); also writes
// Linv = L^-1 (lower triangular, zeros above the diagonal).  *status is set to the 1-based pivot
// index if a non-positive pivot is met (the block is then left unfinished).
__global__ __launch_bounds__(256) void potrf_diag_kernel(float *__restrict__ A, long lda, int nb,
                                                         float *__restrict__ Linv, long ldinv,
                                                         int *__restrict__ status, int pivot_base) {
  __shared__ float S[PNB][PNB + 1];
  __shared__ float X[PNB][PNB + 1];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  for (int e = tid; e < nb * nb; e += 256) {
    const int i = e / nb, j = e % nb;
    S[i][j] = A[(long)i * lda + j];
  }
  __syncthreads();
  for (int k = 0; k < nb; ++k) {
    const float d = S[k][k];
    if (!(d > 0.f)) {  // also catches NaN
      if (tid == 0) { bad = 1; *status = pivot_base + k + 1; }
    }
    __syncthreads();
    if (bad) return;
    const float r = rsqrtf(d);
    // column k below the diagonal (everyone recomputes r from the untouched S[k][k])
    for (int i = k + 1 + tid; i < nb; i += 256) S[i][k] *= r;
    __syncthreads();
    if (tid == 0) S[k][k] = d * r;  // sqrt(d)
    // trailing update of the lower triangle
    const int m = nb - k - 1;
    for (int e = tid; e < m * m; e += 256) {
      const int i = k + 1 + e / m, j = k + 1 + e % m;
      if (j <= i) S[i][j] -= S[i][k] * S[j][k];
    }
    __syncthreads();
  }
  // L^-1 by forward substitution, one column per thread
  if (tid < nb) {
    const int c = tid;
    for (int i = 0; i < nb; ++i) {
      float s = (i == c) ? 1.f : 0.f;
      for (int k = c; k < i; ++k) s -= S[i][k] * X[k][c];
      X[i][c] = (i < c) ? 0.f : s / S[i][i];
    }
  }
  __syncthreads();
  for (int e = tid; e < nb * nb; e += 256) {
    const int i = e / nb, j = e % nb;
    if (j <= i) A[(long)i * lda + j] = S[i][j];
    Linv[(long)i * ldinv + j] = X[i][j];
  }
}

}  // namespace clo

using namespace clo;

extern "C" int clo_potrf_diag_f32(float *A, long lda, int nb, float *Linv, long ldinv, int *status,
                                  int pivot_base, void *stream) {
  CLO_REQUIRE(nb >= 1 && nb <= PNB, "clo_potrf_diag_f32: nb must be in [1, %d], got %d", PNB, nb);
  CLO_REQUIRE(A && Linv && status && lda >= nb && ldinv >= nb, "clo_potrf_diag_f32: bad operand");
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, A, lda, nb, Linv,
                     ldinv, status, pivot_base);
  CLO_CHECK_LAUNCH("potrf_diag_kernel");
  return CLO_OK;
}
