// Output-space curvature of the empirical Fisher computed IN the kernels that already hold the prediction
// (round 5).  The reference re-derives the per-sample loss gradients g_n = d l_n / d f_n on every product
// (gradient_moments.py:48-87); kinds CLO_LOSS_EF_* let the native kernels do the same from the TARGETS instead of taking
// g_n as an operand the host has to recompute (one forward pass of the mini-batch per product) or cache (stale under
// parameter updates):   MSE  g = 2 (f - y)      CE  g = softmax(f) - onehot(label)      BCE  g = sigmoid(f) - y
// H_n u = scale * g_n <g_n, u>.   aux = targets: [N][C] floats (MSE / BCE) or [N] class labels stored as floats (CE).
#pragma once
#include "clo_common.h"

namespace clo {

__host__ __device__ inline bool loss_is_ef(int kind) { return kind >= CLO_LOSS_EF_MSE && kind <= CLO_LOSS_EF_BCE; }
// floats of targets the kernels read for N samples
__host__ __device__ inline long ef_target_floats(int kind, long N, int C) { return kind == CLO_LOSS_EF_CE ? N : N * C; }
__device__ __forceinline__ const float *ef_target_row(int kind, const float *aux, long n, int C) {
  return kind == CLO_LOSS_EF_CE ? aux + n : aux + n * C;
}

// g[0 .. C) of one sample from its prediction f (unit stride) and target row t; C <= CMAX, everything in registers
template <int CMAX>
__device__ __forceinline__ void ef_grad_row(int kind, const float *f, const float *t, int C, float (&g)[CMAX]) {
  if (kind == CLO_LOSS_EF_MSE) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c) g[c] = c < C ? 2.f * (f[c] - t[c]) : 0.f;
  } else if (kind == CLO_LOSS_EF_BCE) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c) g[c] = c < C ? 1.f / (1.f + __expf(-f[c])) - t[c] : 0.f;
  } else {
    float mx = -INFINITY, se = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) mx = fmaxf(mx, f[c]);
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      g[c] = c < C ? __expf(f[c] - mx) : 0.f;
      se += g[c];
    }
    const float inv = 1.f / se;
    const int label = (int)t[0];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) g[c] = c < C ? g[c] * inv - (c == label ? 1.f : 0.f) : 0.f;
  }
}

// one entry g_c for wide outputs (CE: `mx`, `inv` = max and 1 / sum exp of the sample, computed by the caller)
__device__ __forceinline__ float ef_grad_at(int kind, float fc, const float *t, int c, float mx, float inv) {
  if (kind == CLO_LOSS_EF_MSE) return 2.f * (fc - t[c]);
  if (kind == CLO_LOSS_EF_BCE) return 1.f / (1.f + __expf(-fc)) - t[c];
  return __expf(fc - mx) * inv - (c == (int)t[0] ? 1.f : 0.f);
}

}  // namespace clo
