// MLP fast path for GGN-type curvature-vector products (ggn.py:41-72 of the reference).
//
// For mini-batches of up to 8 rows per pass the per-layer products are GEMV-shaped and
// HBM-bound: the kernels below stream each weight matrix exactly once per pass with
// 16-byte-per-lane coalesced loads, keep the (tiny) activations in LDS / registers and
// reduce with wavefront shuffles.  Larger batches go through the MFMA GEMM (gemm.hip).
//
//   fwd_jvp_skinny : z = a W^T + b, dz = da W^T + a VW^T + Vb, activation + its derivative
//   loss_hessian   : w = s * H(f) u                    (per sample, tiny)
//   bwd_outer      : out_W = beta out_W + alpha delta^T a_prev   (pure write stream)
//   bwd_dprev      : delta_prev = dphi_prev * (delta W)           (reads W once)
#include "clo_common.h"

namespace clo {

int launch_gemm_simple(int M, int N, int K, float alpha, const float *A, long sa_m, long sa_k,
                       const float *B, long sb_k, long sb_n, float beta, float *C, long ldc,
                       float *ws, long ws_floats, hipStream_t st);

constexpr int KS = 256;  // k-slice per wave step: 64 lanes x 4 floats
constexpr int NB = 8;    // batch rows per skinny pass

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// ------------------------------------------------------------------------------------------
// Fused forward + JVP through one Linear layer, N <= 8 batch rows.
// Block = 4 waves, each wave owns R output features (rows of W); lanes split k.
// ------------------------------------------------------------------------------------------
template <int R, bool VEC, bool HAS_V, bool HAS_DA>
__global__ __launch_bounds__(256) void fwd_jvp_skinny_kernel(
    const float *__restrict__ W, const float *__restrict__ b, const float *__restrict__ VW,
    const float *__restrict__ Vb, const float *__restrict__ a_in,
    const float *__restrict__ da_in, float *__restrict__ a_out, float *__restrict__ da_out,
    float *__restrict__ dphi_out, int N, int d_in, int d_out, int act) {
  __shared__ __attribute__((aligned(16))) float s_a[NB * KS];
  __shared__ __attribute__((aligned(16))) float s_da[HAS_DA ? NB * KS : 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = (blockIdx.x * 4 + wave) * R;

  float z[R][NB], dz[R][NB];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < NB; ++n) { z[r][n] = 0.f; dz[r][n] = 0.f; }

  // staging registers: 2 x 4 floats per thread per activation array (8 rows x 256 floats)
  float4 st_a[2], st_da[2];
  float4 wn[R], vn[R];

  auto load_slice = [&](int ks) {
    // activations: thread t covers rows n = (t>>6) + 4q, elements (t&63)*4.. (VEC) or lane+64e
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = (tid >> 6) + 4 * q;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vd = va;
      if (n < N) {
        if (VEC) {
          const int k = ks + lane * 4;
          if (k < d_in) {
            va = *reinterpret_cast<const float4 *>(a_in + (long)n * d_in + k);
            if (HAS_DA) vd = *reinterpret_cast<const float4 *>(da_in + (long)n * d_in + k);
          }
        } else {
          float *pa = reinterpret_cast<float *>(&va), *pd = reinterpret_cast<float *>(&vd);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = ks + lane + 64 * e;
            if (k < d_in) {
              pa[e] = a_in[(long)n * d_in + k];
              if (HAS_DA) pd[e] = da_in[(long)n * d_in + k];
            }
          }
        }
      }
      st_a[q] = va;
      st_da[q] = vd;
    }
    // weights
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = w4;
      const int j = j0 + r;
      if (j < d_out) {
        if (VEC) {
          const int k = ks + lane * 4;
          if (k < d_in) {
            w4 = *reinterpret_cast<const float4 *>(W + (long)j * d_in + k);
            if (HAS_V) v4 = *reinterpret_cast<const float4 *>(VW + (long)j * d_in + k);
          }
        } else {
          float *pw = reinterpret_cast<float *>(&w4), *pv = reinterpret_cast<float *>(&v4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = ks + lane + 64 * e;
            if (k < d_in) {
              pw[e] = W[(long)j * d_in + k];
              if (HAS_V) pv[e] = VW[(long)j * d_in + k];
            }
          }
        }
      }
      wn[r] = w4;
      vn[r] = v4;
    }
  };

  load_slice(0);
  for (int ks = 0; ks < d_in; ks += KS) {
    __syncthreads();  // everyone finished reading the previous slice from LDS
    // stage -> LDS.  VEC layout: [n][4*lane .. 4*lane+3]; scalar layout: [n][lane + 64e]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = (tid >> 6) + 4 * q;
      if (VEC) {
        *reinterpret_cast<float4 *>(&s_a[n * KS + lane * 4]) = st_a[q];
        if (HAS_DA) *reinterpret_cast<float4 *>(&s_da[n * KS + lane * 4]) = st_da[q];
      } else {
        const float *pa = reinterpret_cast<const float *>(&st_a[q]);
        const float *pd = reinterpret_cast<const float *>(&st_da[q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s_a[n * KS + lane + 64 * e] = pa[e];
          if (HAS_DA) s_da[n * KS + lane + 64 * e] = pd[e];
        }
      }
    }
    float4 wc[R], vc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { wc[r] = wn[r]; vc[r] = vn[r]; }
    __syncthreads();
    if (ks + KS < d_in) load_slice(ks + KS);  // in flight while we compute

#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float4 a4, d4;
      if (VEC) {
        a4 = *reinterpret_cast<const float4 *>(&s_a[n * KS + lane * 4]);
        if (HAS_DA) d4 = *reinterpret_cast<const float4 *>(&s_da[n * KS + lane * 4]);
      } else {
        a4 = make_float4(s_a[n * KS + lane], s_a[n * KS + lane + 64], s_a[n * KS + lane + 128],
                         s_a[n * KS + lane + 192]);
        if (HAS_DA)
          d4 = make_float4(s_da[n * KS + lane], s_da[n * KS + lane + 64],
                           s_da[n * KS + lane + 128], s_da[n * KS + lane + 192]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        z[r][n] += dot4(wc[r], a4);
        if (HAS_V) dz[r][n] += dot4(vc[r], a4);
        if (HAS_DA) dz[r][n] += dot4(wc[r], d4);
      }
    }
  }

  // wavefront reduction; afterwards every lane holds every sum
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      z[r][n] = wave_sum(z[r][n]);
      if (HAS_V || HAS_DA) dz[r][n] = wave_sum(dz[r][n]);
    }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      if (lane == r * NB + n) {
        const int j = j0 + r;
        if (j < d_out && n < N) {
          float zz = z[r][n] + (b ? b[j] : 0.f);
          float dphi;
          const float av = act_apply(act, zz, dphi);
          a_out[(long)n * d_out + j] = av;
          if (dphi_out) dphi_out[(long)n * d_out + j] = dphi;
          if (HAS_V || HAS_DA) {
            float dzz = dz[r][n] + ((HAS_V && Vb) ? Vb[j] : 0.f);
            da_out[(long)n * d_out + j] = dphi * dzz;
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------
// Output-space curvature: one block per sample.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float *s_red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  return t;
}
__device__ __forceinline__ float block_max(float v, float *s_red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, s_red[w]);
  return t;
}

__global__ __launch_bounds__(256) void loss_hessian_kernel(
    int kind, const float *__restrict__ f, const float *__restrict__ aux, int aux_rank,
    const float *__restrict__ u, const float *__restrict__ dphi_last, float *__restrict__ w, int C,
    float scale) {
  __shared__ float s_red[8];
  const int n = blockIdx.x;
  const float *fn = f + (long)n * C, *un = u + (long)n * C;
  float *wn = w + (long)n * C;
  const float *dp = dphi_last ? dphi_last + (long)n * C : nullptr;
  if (kind == CLO_LOSS_MSE) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) wn[c] = scale * un[c] * (dp ? dp[c] : 1.f);
  } else if (kind == CLO_LOSS_BCE) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float s = 1.f / (1.f + __expf(-fn[c]));
      wn[c] = scale * s * (1.f - s) * un[c] * (dp ? dp[c] : 1.f);
    }
  } else if (kind == CLO_LOSS_CE) {
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += blockDim.x) mx = fmaxf(mx, fn[c]);
    mx = block_max(mx, s_red);
    float se = 0.f, spu = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float e = __expf(fn[c] - mx);
      se += e;
      spu += e * un[c];
    }
    se = block_sum(se, s_red);
    spu = block_sum(spu, s_red);
    const float inv = 1.f / se;
    const float pu = spu * inv;  // p . u
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float p = __expf(fn[c] - mx) * inv;
      wn[c] = scale * p * (un[c] - pu) * (dp ? dp[c] : 1.f);
    }
  } else {  // CLO_LOSS_RANK1 (rank-`aux_rank` sum of outer products g g^T)
    for (int c = threadIdx.x; c < C; c += blockDim.x) wn[c] = 0.f;
    for (int m = 0; m < aux_rank; ++m) {
      const float *g = aux + ((long)n * aux_rank + m) * C;
      float s = 0.f;
      for (int c = threadIdx.x; c < C; c += blockDim.x) s += g[c] * un[c];
      s = block_sum(s, s_red);
      for (int c = threadIdx.x; c < C; c += blockDim.x) wn[c] += scale * g[c] * s;
    }
    if (dp) {
      __syncthreads();
      for (int c = threadIdx.x; c < C; c += blockDim.x) wn[c] *= dp[c];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward, parameter part: out_W[j][i] = beta out_W[j][i] + alpha sum_n delta[n][j] a_prev[n][i]
// grid = (i-chunks of 256 columns, row tiles of JT rows); pure write stream.
// ------------------------------------------------------------------------------------------
constexpr int JT = 32;

template <bool VEC>
__global__ __launch_bounds__(256) void bwd_outer_kernel(
    const float *__restrict__ delta, const float *__restrict__ a_prev, float *__restrict__ out_W,
    float *__restrict__ out_b, float alpha, float beta, int N, int d_in, int d_out) {
  __shared__ __attribute__((aligned(16))) float s_d[JT * NB];  // [row][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * KS;
  const int jbase = blockIdx.y * JT;

  {  // delta tile -> LDS, transposed to [j][n]
    const int jj = tid >> 3, n = tid & 7;  // 32 rows x 8
    const int j = jbase + jj;
    s_d[jj * NB + n] = (j < d_out && n < N) ? delta[(long)n * d_out + j] : 0.f;
  }
  float4 a[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
      if (VEC) {
        const int i = i0 + lane * 4;
        if (i < d_in) v = *reinterpret_cast<const float4 *>(a_prev + (long)n * d_in + i);
      } else {
        float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = i0 + lane + 64 * e;
          if (i < d_in) pv[e] = a_prev[(long)n * d_in + i];
        }
      }
    }
    a[n] = v;
  }
  __syncthreads();

  if (out_b && blockIdx.x == 0 && tid < JT) {
    const int j = jbase + tid;
    if (j < d_out) {
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < NB; ++n) s += s_d[tid * NB + n];
      out_b[j] = (beta != 0.f ? beta * out_b[j] : 0.f) + alpha * s;
    }
  }

  for (int jj = wave; jj < JT; jj += 4) {
    const int j = jbase + jj;
    if (j >= d_out) break;
    const float4 d0 = *reinterpret_cast<const float4 *>(&s_d[jj * NB]);
    const float4 d1 = *reinterpret_cast<const float4 *>(&s_d[jj * NB + 4]);
    const float dn[NB] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      o.x += dn[n] * a[n].x; o.y += dn[n] * a[n].y; o.z += dn[n] * a[n].z; o.w += dn[n] * a[n].w;
    }
    if (VEC) {
      const int i = i0 + lane * 4;
      if (i < d_in) {
        float4 *po = reinterpret_cast<float4 *>(out_W + (long)j * d_in + i);
        float4 r = make_float4(alpha * o.x, alpha * o.y, alpha * o.z, alpha * o.w);
        if (beta != 0.f) {
          const float4 old = *po;
          r.x += beta * old.x; r.y += beta * old.y; r.z += beta * old.z; r.w += beta * old.w;
        }
        *po = r;
      }
    } else {
      const float *po4 = reinterpret_cast<const float *>(&o);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + lane + 64 * e;
        if (i < d_in) {
          float *po = out_W + (long)j * d_in + i;
          *po = (beta != 0.f ? beta * *po : 0.f) + alpha * po4[e];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward, data part: P[jb][n][i] = sum_{j in range(jb)} W[j][i] delta[n][j]
// grid = (i-chunks of 256 columns, JB row ranges); if JB == 1 the kernel applies dphi_prev and
// writes delta_prev directly, otherwise bwd_dprev_finish sums the JB partial slabs.
// ------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void bwd_dprev_kernel(
    const float *__restrict__ W, const float *__restrict__ delta,
    const float *__restrict__ dphi_prev, float *__restrict__ dst, int N, int d_in, int d_out,
    int rows_per_block, int final_write) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_d = smem;                           // [rows_per_block][NB]
  float *s_red = smem + rows_per_block * NB;   // [4][NB][KS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * KS;
  const int jbase = blockIdx.y * rows_per_block;
  const int jend = min(d_out, jbase + rows_per_block);

  for (int e = tid; e < rows_per_block * NB; e += 256) {
    const int jj = e >> 3, n = e & 7;
    const int j = jbase + jj;
    s_d[e] = (j < d_out && n < N) ? delta[(long)n * d_out + j] : 0.f;
  }
  __syncthreads();

  float4 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);

  // two rows in flight per wave iteration
  for (int j = jbase + wave; j < jend; j += 8) {
    float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
    const int j1 = j + 4;
    if (VEC) {
      const int i = i0 + lane * 4;
      if (i < d_in) {
        w0 = *reinterpret_cast<const float4 *>(W + (long)j * d_in + i);
        if (j1 < jend) w1 = *reinterpret_cast<const float4 *>(W + (long)j1 * d_in + i);
      }
    } else {
      float *p0 = reinterpret_cast<float *>(&w0), *p1 = reinterpret_cast<float *>(&w1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + lane + 64 * e;
        if (i < d_in) {
          p0[e] = W[(long)j * d_in + i];
          if (j1 < jend) p1[e] = W[(long)j1 * d_in + i];
        }
      }
    }
    {
      const float *dj = &s_d[(j - jbase) * NB];
      const float4 d0 = *reinterpret_cast<const float4 *>(dj);
      const float4 d1 = *reinterpret_cast<const float4 *>(dj + 4);
      const float dn[NB] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        acc[n].x += dn[n] * w0.x; acc[n].y += dn[n] * w0.y;
        acc[n].z += dn[n] * w0.z; acc[n].w += dn[n] * w0.w;
      }
    }
    if (j1 < jend) {
      const float *dj = &s_d[(j1 - jbase) * NB];
      const float4 d0 = *reinterpret_cast<const float4 *>(dj);
      const float4 d1 = *reinterpret_cast<const float4 *>(dj + 4);
      const float dn[NB] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        acc[n].x += dn[n] * w1.x; acc[n].y += dn[n] * w1.y;
        acc[n].z += dn[n] * w1.z; acc[n].w += dn[n] * w1.w;
      }
    }
  }

  // cross-wave reduction through LDS; column c of the chunk is lane*4+e (VEC) or lane+64e.
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    float *dstp = &s_red[(wave * NB + n) * KS];
    if (VEC) {
      *reinterpret_cast<float4 *>(dstp + lane * 4) = acc[n];
    } else {
      dstp[lane] = acc[n].x; dstp[lane + 64] = acc[n].y;
      dstp[lane + 128] = acc[n].z; dstp[lane + 192] = acc[n].w;
    }
  }
  __syncthreads();
  for (int e = tid; e < NB * KS; e += 256) {
    const int n = e >> 8, c = e & 255;
    const int i = i0 + c;
    if (n < N && i < d_in) {
      float s = s_red[(0 * NB + n) * KS + c] + s_red[(1 * NB + n) * KS + c] +
                s_red[(2 * NB + n) * KS + c] + s_red[(3 * NB + n) * KS + c];
      if (final_write) {
        dst[(long)n * d_in + i] = s * dphi_prev[(long)n * d_in + i];
      } else {
        dst[((long)blockIdx.y * NB + n) * d_in + i] = s;
      }
    }
  }
}

__global__ void bwd_dprev_finish_kernel(const float *__restrict__ P,
                                        const float *__restrict__ dphi_prev,
                                        float *__restrict__ delta_prev, int N, int d_in, int JB) {
  const long total = (long)N * d_in;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int n = e / d_in, i = e % d_in;
    float s = 0.f;
    for (int jb = 0; jb < JB; ++jb) s += P[((long)jb * NB + n) * d_in + i];
    delta_prev[e] = s * dphi_prev[e];
  }
}

// Elementwise epilogues of the GEMM (large-batch) path.
__global__ void fwd_epilogue_kernel(float *__restrict__ z_a, float *__restrict__ dz_da,
                                    float *__restrict__ dphi_out, const float *__restrict__ b,
                                    const float *__restrict__ Vb, long N, int d_out, int act) {
  const long total = N * d_out;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int j = e % d_out;
    float dphi;
    const float av = act_apply(act, z_a[e] + (b ? b[j] : 0.f), dphi);
    z_a[e] = av;
    if (dphi_out) dphi_out[e] = dphi;
    if (dz_da) dz_da[e] = dphi * (dz_da[e] + (Vb ? Vb[j] : 0.f));
  }
}
__global__ void mul_inplace_kernel(float *__restrict__ x, const float *__restrict__ m, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long)gridDim.x * blockDim.x)
    x[e] *= m[e];
}
__global__ void colsum_small_kernel(float *__restrict__ out, const float *__restrict__ X, long rows,
                                    int d, float alpha, float beta) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (j < d)
    for (long r = wave; r < rows; r += 4) s += X[r * d + j];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && j < d)
    out[j] = (beta != 0.f ? beta * out[j] : 0.f) +
             alpha * (part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}

static inline unsigned ew_grid(long n) {
  return (unsigned)std::max<long>(1, std::min<long>(cdiv(n, 256), kNumCU * 8L));
}

// number of row ranges for bwd_dprev: aim at ~2 blocks per CU, >= 32 rows per block
static int pick_jb(int d_in, int d_out) {
  const long ichunks = cdiv(d_in, KS);
  long jb = (2L * kNumCU) / ichunks;
  jb = std::min<long>(jb, cdiv(d_out, 32));
  jb = std::max<long>(jb, cdiv(d_out, 512));  // <= 512 rows of delta per block in LDS
  return (int)std::max<long>(1, jb);
}

template <bool VEC>
static int launch_fwd(const float *W, const float *b, const float *VW, const float *Vb,
                      const float *a_in, const float *da_in, float *a_out, float *da_out,
                      float *dphi_out, int N, int d_in, int d_out, int act, hipStream_t st) {
  // R = 2 rows per wave -> 8 rows per block; R = 1 for small layers (more blocks in flight)
  const bool has_v = VW != nullptr, has_da = da_in != nullptr;
  const bool small = d_out < 8 * 2 * kNumCU / 4;  // fewer than ~128 blocks at R=2
  const int R = small ? 1 : 2;
  dim3 grid((unsigned)cdiv(d_out, 4 * R)), block(256);
#define CLO_FWD(RR, HV, HD)                                                                    \
  hipLaunchKernelGGL((fwd_jvp_skinny_kernel<RR, VEC, HV, HD>), grid, block, 0, st, W, b, VW, Vb, \
                     a_in, da_in, a_out, da_out, dphi_out, N, d_in, d_out, act)
  if (R == 1) {
    if (has_v && has_da) CLO_FWD(1, true, true);
    else if (has_v) CLO_FWD(1, true, false);
    else if (has_da) CLO_FWD(1, false, true);
    else CLO_FWD(1, false, false);
  } else {
    if (has_v && has_da) CLO_FWD(2, true, true);
    else if (has_v) CLO_FWD(2, true, false);
    else if (has_da) CLO_FWD(2, false, true);
    else CLO_FWD(2, false, false);
  }
#undef CLO_FWD
  CLO_CHECK_LAUNCH("fwd_jvp_skinny_kernel");
  return CLO_OK;
}

static bool vec_ok(int d, std::initializer_list<const void *> ptrs) {
  if (d % 4 != 0) return false;
  for (const void *p : ptrs)
    if (p && !aligned16(p)) return false;
  return true;
}

// One skinny pass (N <= 8).
static int fwd_pass(const float *W, const float *b, const float *VW, const float *Vb,
                    const float *a_in, const float *da_in, float *a_out, float *da_out,
                    float *dphi_out, int N, int d_in, int d_out, int act, hipStream_t st) {
  if (vec_ok(d_in, {W, VW, a_in, da_in}))
    return launch_fwd<true>(W, b, VW, Vb, a_in, da_in, a_out, da_out, dphi_out, N, d_in, d_out, act,
                            st);
  return launch_fwd<false>(W, b, VW, Vb, a_in, da_in, a_out, da_out, dphi_out, N, d_in, d_out, act,
                           st);
}

static int bwd_pass(const float *W, const float *delta, const float *a_prev,
                    const float *dphi_prev, float *out_W, float *out_b, float *delta_prev,
                    float alpha, float beta, int N, int d_in, int d_out, float *ws,
                    hipStream_t st) {
  if (out_W) {
    dim3 grid((unsigned)cdiv(d_in, KS), (unsigned)cdiv(d_out, JT));
    if (vec_ok(d_in, {a_prev, out_W}))
      hipLaunchKernelGGL((bwd_outer_kernel<true>), grid, dim3(256), 0, st, delta, a_prev, out_W,
                         out_b, alpha, beta, N, d_in, d_out);
    else
      hipLaunchKernelGGL((bwd_outer_kernel<false>), grid, dim3(256), 0, st, delta, a_prev, out_W,
                         out_b, alpha, beta, N, d_in, d_out);
    CLO_CHECK_LAUNCH("bwd_outer_kernel");
  }
  if (delta_prev) {
    const int JB = pick_jb(d_in, d_out);
    const int rpb = (int)cdiv(d_out, JB);
    const int JBe = (int)cdiv(d_out, rpb);
    const size_t smem = ((size_t)rpb * NB + 4 * NB * KS) * sizeof(float);
    dim3 grid((unsigned)cdiv(d_in, KS), (unsigned)JBe);
    float *dst = JBe == 1 ? delta_prev : ws;
    if (vec_ok(d_in, {W}))
      hipLaunchKernelGGL((bwd_dprev_kernel<true>), grid, dim3(256), smem, st, W, delta, dphi_prev,
                         dst, N, d_in, d_out, rpb, JBe == 1 ? 1 : 0);
    else
      hipLaunchKernelGGL((bwd_dprev_kernel<false>), grid, dim3(256), smem, st, W, delta, dphi_prev,
                         dst, N, d_in, d_out, rpb, JBe == 1 ? 1 : 0);
    CLO_CHECK_LAUNCH("bwd_dprev_kernel");
    if (JBe > 1) {
      hipLaunchKernelGGL(bwd_dprev_finish_kernel, dim3(ew_grid((long)N * d_in)), dim3(256), 0, st,
                         ws, dphi_prev, delta_prev, N, d_in, JBe);
      CLO_CHECK_LAUNCH("bwd_dprev_finish_kernel");
    }
  }
  return CLO_OK;
}

constexpr int SKINNY_MAX_N = 16;  // up to two 8-row passes; above that the MFMA GEMM path wins

static long gemm_ws_floats(int N, int dmax) {
  // split-K partial slabs for the widest product of the large-batch path
  return 16L * (long)std::max(N, 128) * dmax;
}

}  // namespace clo

using namespace clo;

extern "C" long clo_mlp_bwd_ws_floats(int N, int d_in, int d_out) {
  (void)N;
  return (long)pick_jb(d_in, d_out) * NB * d_in + 64;
}

extern "C" int clo_mlp_fwd_jvp_layer(const float *W, const float *b, const float *VW,
                                     const float *Vb, const float *a_in, const float *da_in,
                                     float *a_out, float *da_out, float *dphi_out, int N, int d_in,
                                     int d_out, int act, void *stream) {
  CLO_REQUIRE(N >= 0 && d_in > 0 && d_out > 0, "clo_mlp_fwd_jvp_layer: bad sizes");
  CLO_REQUIRE(act >= 0 && act <= 3, "clo_mlp_fwd_jvp_layer: unknown activation %d", act);
  CLO_REQUIRE(W && a_in && a_out, "clo_mlp_fwd_jvp_layer: null operand");
  CLO_REQUIRE(!(VW || da_in) || da_out, "clo_mlp_fwd_jvp_layer: da_out required for a JVP");
  CLO_REQUIRE(N <= SKINNY_MAX_N, "clo_mlp_fwd_jvp_layer: N=%d > %d, use clo_mlp_ggn_matvec", N,
              SKINNY_MAX_N);
  hipStream_t st = (hipStream_t)stream;
  for (int n0 = 0; n0 < N; n0 += NB) {
    const int nn = std::min(NB, N - n0);
    int rc = fwd_pass(W, b, VW, Vb, a_in + (long)n0 * d_in,
                      da_in ? da_in + (long)n0 * d_in : nullptr, a_out + (long)n0 * d_out,
                      da_out ? da_out + (long)n0 * d_out : nullptr,
                      dphi_out ? dphi_out + (long)n0 * d_out : nullptr, nn, d_in, d_out, act, st);
    if (rc != CLO_OK) return rc;
  }
  return CLO_OK;
}

extern "C" int clo_loss_hessian_apply(int kind, const float *f, const float *aux, int aux_rank,
                                      const float *u, const float *dphi_last, float *w, int N,
                                      int C, float scale, void *stream) {
  CLO_REQUIRE(kind >= 0 && kind <= 3, "clo_loss_hessian_apply: unknown kind %d", kind);
  CLO_REQUIRE(N >= 0 && C > 0, "clo_loss_hessian_apply: bad sizes");
  if (N == 0) return CLO_OK;
  CLO_REQUIRE(f && u && w, "clo_loss_hessian_apply: null operand");
  CLO_REQUIRE(kind != CLO_LOSS_RANK1 || (aux && aux_rank >= 1),
              "clo_loss_hessian_apply: RANK1 needs aux and aux_rank >= 1");
  hipLaunchKernelGGL(loss_hessian_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, kind, f, aux,
                     aux_rank, u, dphi_last, w, C, scale);
  CLO_CHECK_LAUNCH("loss_hessian_kernel");
  return CLO_OK;
}

extern "C" int clo_mlp_bwd_layer(const float *W, const float *delta, const float *a_prev,
                                 const float *dphi_prev, float *out_W, float *out_b,
                                 float *delta_prev, float alpha, float beta, int N, int d_in,
                                 int d_out, float *ws, void *stream) {
  CLO_REQUIRE(N >= 0 && d_in > 0 && d_out > 0, "clo_mlp_bwd_layer: bad sizes");
  CLO_REQUIRE(N <= SKINNY_MAX_N, "clo_mlp_bwd_layer: N=%d > %d, use clo_mlp_ggn_matvec", N,
              SKINNY_MAX_N);
  CLO_REQUIRE(delta && (!out_W || a_prev), "clo_mlp_bwd_layer: null operand");
  CLO_REQUIRE(!delta_prev || (W && dphi_prev && ws), "clo_mlp_bwd_layer: delta_prev needs W, dphi_prev, ws");
  hipStream_t st = (hipStream_t)stream;
  for (int n0 = 0; n0 < N || n0 == 0; n0 += NB) {
    const int nn = std::max(0, std::min(NB, N - n0));
    int rc = bwd_pass(W, delta + (long)n0 * d_out, a_prev ? a_prev + (long)n0 * d_in : nullptr,
                      dphi_prev ? dphi_prev + (long)n0 * d_in : nullptr, out_W, out_b,
                      delta_prev ? delta_prev + (long)n0 * d_in : nullptr, alpha,
                      n0 == 0 ? beta : 1.f, nn, d_in, d_out, ws, st);
    if (rc != CLO_OK) return rc;
    if (N == 0) break;
  }
  return CLO_OK;
}

// Workspace layout of clo_mlp_ggn_matvec (floats):
//   per layer l = 1..L : a_l, da_l, dphi_l, each [N][d_l]
//   delta ping/pong    : 2 x [N][dmax]
//   bwd partial slabs  : max_l clo_mlp_bwd_ws_floats
//   GEMM split-K slabs : only when N > SKINNY_MAX_N
extern "C" long clo_mlp_ggn_ws_floats(int L, const int *dims, int N) {
  if (L <= 0 || !dims || N < 0) return 0;
  long total = 0;
  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  for (int l = 1; l <= L; ++l) total += 3L * N * dims[l];
  total += 2L * N * dmax;
  long part = 0;
  for (int l = 1; l <= L; ++l)
    part = std::max(part, clo_mlp_bwd_ws_floats(N, dims[l - 1], dims[l]));
  total += part;
  if (N > SKINNY_MAX_N) total += gemm_ws_floats(N, dmax);
  return total + 256;
}

extern "C" int clo_mlp_ggn_matvec(int L, const int *dims, const int *acts, const float *const *W,
                                  const float *const *b, const float *const *VW,
                                  const float *const *Vb, float *const *OW, float *const *Ob,
                                  const float *X, int N, int loss_kind, const float *aux,
                                  int aux_rank, float loss_scale, float alpha, float beta,
                                  float *ws, void *stream) {
  CLO_REQUIRE(L >= 1 && L <= 64 && dims && acts && W && VW && OW, "clo_mlp_ggn_matvec: bad layer table");
  CLO_REQUIRE(N >= 0 && X && ws, "clo_mlp_ggn_matvec: bad batch / workspace");
  CLO_REQUIRE(loss_kind >= 0 && loss_kind <= 3, "clo_mlp_ggn_matvec: unknown loss kind %d", loss_kind);
  CLO_REQUIRE(loss_kind != CLO_LOSS_RANK1 || (aux && aux_rank >= 1),
              "clo_mlp_ggn_matvec: RANK1 needs aux and aux_rank >= 1");
  for (int l = 0; l <= L; ++l) CLO_REQUIRE(dims[l] > 0, "clo_mlp_ggn_matvec: dims[%d] <= 0", l);
  for (int l = 0; l < L; ++l) {
    CLO_REQUIRE(acts[l] >= 0 && acts[l] <= 3, "clo_mlp_ggn_matvec: unknown activation");
    CLO_REQUIRE(W[l] && VW[l] && OW[l], "clo_mlp_ggn_matvec: null weight pointer in layer %d", l);
  }
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {  // empty batch contributes nothing; still honour beta
    if (beta != 1.f)
      for (int l = 0; l < L; ++l) {
        int rc = clo_axpby_f32(OW[l], OW[l], (long)dims[l] * dims[l + 1], 0.f, beta, stream);
        if (rc != CLO_OK) return rc;
        if (Ob && Ob[l]) {
          rc = clo_axpby_f32(Ob[l], Ob[l], dims[l + 1], 0.f, beta, stream);
          if (rc != CLO_OK) return rc;
        }
      }
    return CLO_OK;
  }

  int dmax = 0;
  for (int l = 0; l <= L; ++l) dmax = std::max(dmax, dims[l]);
  // carve the workspace
  float *p = ws;
  float *a[65], *da[65], *dphi[65];
  a[0] = const_cast<float *>(X); da[0] = nullptr; dphi[0] = nullptr;
  for (int l = 1; l <= L; ++l) {
    const long sz = (long)N * dims[l];
    a[l] = p; p += sz; da[l] = p; p += sz; dphi[l] = p; p += sz;
  }
  float *dl0 = p; p += (long)N * dmax;
  float *dl1 = p; p += (long)N * dmax;
  float *part = p;
  long part_sz = 0;
  for (int l = 1; l <= L; ++l)
    part_sz = std::max(part_sz, clo_mlp_bwd_ws_floats(N, dims[l - 1], dims[l]));
  p += part_sz;
  float *gws = p;
  const long gws_sz = N > SKINNY_MAX_N ? gemm_ws_floats(N, dmax) : 0;

  const bool skinny = N <= SKINNY_MAX_N;
  int rc;
  // ---- forward + JVP
  for (int l = 1; l <= L; ++l) {
    const int di = dims[l - 1], dout = dims[l];
    const float *bl = b ? b[l - 1] : nullptr, *vbl = Vb ? Vb[l - 1] : nullptr;
    if (skinny) {
      rc = clo_mlp_fwd_jvp_layer(W[l - 1], bl, VW[l - 1], vbl, a[l - 1], da[l - 1], a[l], da[l],
                                 dphi[l], N, di, dout, acts[l - 1], stream);
      if (rc != CLO_OK) return rc;
    } else {
      // Z = A W^T ; dZ = A VW^T (+ dA W^T)
      rc = launch_gemm_simple(N, dout, di, 1.f, a[l - 1], di, 1, W[l - 1], 1, di, 0.f, a[l], dout,
                              gws, gws_sz, st);
      if (rc != CLO_OK) return rc;
      rc = launch_gemm_simple(N, dout, di, 1.f, a[l - 1], di, 1, VW[l - 1], 1, di, 0.f, da[l], dout,
                              gws, gws_sz, st);
      if (rc != CLO_OK) return rc;
      if (da[l - 1]) {
        rc = launch_gemm_simple(N, dout, di, 1.f, da[l - 1], di, 1, W[l - 1], 1, di, 1.f, da[l],
                                dout, gws, gws_sz, st);
        if (rc != CLO_OK) return rc;
      }
      hipLaunchKernelGGL(fwd_epilogue_kernel, dim3(ew_grid((long)N * dout)), dim3(256), 0, st, a[l],
                         da[l], dphi[l], bl, vbl, (long)N, dout, acts[l - 1]);
      CLO_CHECK_LAUNCH("fwd_epilogue_kernel");
    }
  }
  // ---- output-space curvature: delta_L = dphi_L * (alpha * s * H u)
  {
    const int C = dims[L];
    const bool last_linear = acts[L - 1] == CLO_ACT_IDENTITY;
    hipLaunchKernelGGL(loss_hessian_kernel, dim3(N), dim3(256), 0, st, loss_kind, a[L], aux,
                       aux_rank, da[L], last_linear ? nullptr : dphi[L], dl0, C, loss_scale * alpha);
    CLO_CHECK_LAUNCH("loss_hessian_kernel");
  }
  // ---- backward
  float *dcur = dl0, *dnext = dl1;
  for (int l = L; l >= 1; --l) {
    const int di = dims[l - 1], dout = dims[l];
    float *obl = Ob ? Ob[l - 1] : nullptr;
    float *dprev = l > 1 ? dnext : nullptr;
    if (skinny) {
      rc = clo_mlp_bwd_layer(W[l - 1], dcur, a[l - 1], dphi[l - 1], OW[l - 1], obl, dprev, 1.f, beta,
                             N, di, dout, part, stream);
      if (rc != CLO_OK) return rc;
    } else {
      // out_W = beta out_W + delta^T a_prev
      rc = launch_gemm_simple(dout, di, N, 1.f, dcur, 1, dout, a[l - 1], di, 1, beta, OW[l - 1], di,
                              gws, gws_sz, st);
      if (rc != CLO_OK) return rc;
      if (obl) {
        hipLaunchKernelGGL(colsum_small_kernel, dim3((unsigned)cdiv(dout, 64)), dim3(256), 0, st,
                           obl, dcur, (long)N, dout, 1.f, beta);
        CLO_CHECK_LAUNCH("colsum_small_kernel");
      }
      if (dprev) {
        rc = launch_gemm_simple(N, di, dout, 1.f, dcur, dout, 1, W[l - 1], di, 1, 0.f, dprev, di, gws,
                                gws_sz, st);
        if (rc != CLO_OK) return rc;
        hipLaunchKernelGGL(mul_inplace_kernel, dim3(ew_grid((long)N * di)), dim3(256), 0, st, dprev,
                           dphi[l - 1], (long)N * di);
        CLO_CHECK_LAUNCH("mul_inplace_kernel");
      }
    }
    std::swap(dcur, dnext);
  }
  return CLO_OK;
}
