// Patch extraction for KFAC's input covariance of Conv2d layers (reference
// kfac_utils.py:78-121: group-mean + torch unfold + transpose).  One launch for the whole
// mini-batch (torch's unfold launches one im2col kernel per sample), written in the layout the
// SYRK wants: out[(b, oh, ow)][(c, kh, kw)] row-major, coalesced along the patch axis.
#include <algorithm>
#include <vector>

#include "clo_common.h"

namespace clo {

struct Im2colArgs {
  const float *x;
  float *out;
  int B, C, H, W, KH, KW, SH, SW, PH, PW, DH, DW, OH, OW;
};

__global__ __launch_bounds__(256) void im2col_kernel(const Im2colArgs p) {
  const int Q = p.C * p.KH * p.KW;          // patch length
  const long rows = (long)p.B * p.OH * p.OW;
  const long total = rows * Q;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long r = e / Q;
    const int q = (int)(e - r * Q);
    const int kw = q % p.KW, kh = (q / p.KW) % p.KH, c = q / (p.KW * p.KH);
    const int ow = (int)(r % p.OW), oh = (int)((r / p.OW) % p.OH);
    const long b = r / ((long)p.OW * p.OH);
    const int ih = oh * p.SH - p.PH + kh * p.DH, iw = ow * p.SW - p.PW + kw * p.DW;
    float v = 0.f;
    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
      v = p.x[((b * p.C + c) * p.H + ih) * p.W + iw];
    p.out[e] = v;
  }
}

// ---- input covariance of a convolution from the PIXEL Gram matrix ---------------------------------------------------
// KFAC-expand's A = P^T P over the patch matrix P [(b, oh, ow)][(c, kh, kw)] (kfac_utils.py:78-121 + kfac_hooks.py:350)
// only ever multiplies input pixels of the same sample:
//   A[(c1,t1)][(c2,t2)] = sum_b sum_pos x[b][c1][pos + t1] x[b][c2][pos + t2]
//                       = sum_{pos : both taps inside the image} Gam[(c1, pos + t1)][(c2, pos + t2)],
//   Gam = X^T X  with  X = x viewed as [B][C*H*W]  (the "pixel Gram", one dense SYRK with K = B).
// For images smaller than the kernel's reach squared -- (H W)^2 < OH OW (KH KW)^2, e.g. every 3x3 layer of a
// CIFAR-sized ResNet (8x8 ... 1x1 feature maps) -- Gam costs FEWER flops than P^T P (ResNet-18 layer4: 81 x fewer, most
// patch entries are padding zeros), needs no patch matrix at all and runs on aligned dense operands.  This kernel is
// the second half: the fold of Gam into A, one workgroup per (channel group, channel group) tile of Gam staged in LDS.
struct FoldArgs {
  float *C;
  long ldc;
  const float *G;   // [Cc*H*W][ldg] pixel Gram
  long ldg;
  const float *colsum;   // [Cc*H*W] sum_b x[b][c][q] (bias column) or nullptr
  int Cc, H, W, KH, KW, SH, SW, PH, PW, DH, DW, OH, OW;
  int g1, g2;       // channels per tile row / column group
  int ones;         // 1: A has the extra bias row / column
  int sparse;       // 1: C was zero-filled by the host (beta == 0): tap pairs that share no output position are not written
  float nrows;      // B * OH * OW  (corner entry, before alpha)
  float alpha, beta;
};

constexpr int FOLD_T = 256;

__global__ __launch_bounds__(FOLD_T) void patch_fold_kernel(const FoldArgs p) {
  extern __shared__ float fold_lds[];
  const int HW = p.H * p.W, P = p.OH * p.OW, T = p.KH * p.KW;
  const int nb2 = (int)cdiv(p.Cc, p.g2);
  const int tid = threadIdx.x;
  int *qtab = reinterpret_cast<int *>(fold_lds);          // [T][P] input pixel of (tap, output position) or -1
  float *tile = fold_lds + (((long)T * P + 3) & ~3L);
  for (int i = tid; i < T * P; i += FOLD_T) {
    const int t = i / P, pos = i - t * P;
    const int oh = pos / p.OW, ow = pos - oh * p.OW, kh = t / p.KW, kw = t - kh * p.KW;
    const int ih = oh * p.SH - p.PH + kh * p.DH, iw = ow * p.SW - p.PW + kw * p.DW;
    qtab[i] = (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) ? ih * p.W + iw : -1;
  }
  const int by = blockIdx.y, bx = blockIdx.x;
  const int c1_0 = by * p.g1, n1 = min(p.g1, p.Cc - c1_0);
  if (bx == nb2) {
    // bias column / row of this channel group, and (first group) the corner
    __syncthreads();
    for (int o = tid; o < n1 * T; o += FOLD_T) {
      const int c1 = o / T, t1 = o - c1 * T;
      float sacc = 0.f;
      for (int pos = 0; pos < P; ++pos) {
        const int q = qtab[t1 * P + pos];
        if (q >= 0) sacc += p.colsum[(long)(c1_0 + c1) * HW + q];
      }
      const long f = (long)(c1_0 + c1) * T + t1, last = (long)p.Cc * T;
      float *a = p.C + f * p.ldc + last, *b = p.C + last * p.ldc + f;
      const float v = p.alpha * sacc;
      *a = p.beta != 0.f ? p.beta * *a + v : v;
      *b = p.beta != 0.f ? p.beta * *b + v : v;
    }
    if (by == 0 && tid == 0) {
      float *c = p.C + (long)p.Cc * T * p.ldc + (long)p.Cc * T;
      const float v = p.alpha * p.nrows;
      *c = p.beta != 0.f ? p.beta * *c + v : v;
    }
    return;
  }
  const int c2_0 = bx * p.g2, n2 = min(p.g2, p.Cc - c2_0);
  const int rows = n1 * HW, cols = n2 * HW;
  const int ldt = cols | 1;   // odd row pitch: the gather's row index varies fastest across a tap's positions
  const float *src = p.G + (long)c1_0 * HW * p.ldg + (long)c2_0 * HW;
  if ((cols & 3) == 0 && (p.ldg & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
    const int c4n = cols >> 2;
    for (int e = tid; e < rows * c4n; e += FOLD_T) {
      const int r = e / c4n, c = (e - r * c4n) * 4;
      const float4 v = *reinterpret_cast<const float4 *>(src + (long)r * p.ldg + c);
      float *d = tile + r * ldt + c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int e = tid; e < rows * cols; e += FOLD_T) {
      const int r = e / cols, c = e - r * cols;
      tile[r * ldt + c] = src[(long)r * p.ldg + c];
    }
  }
  // tap pairs that share at least one output position (all of them for feature maps larger than the kernel's reach; ONE of
  // 81 for a 1x1 map under a 3x3 kernel): their list lives behind the tile
  int *npairs = reinterpret_cast<int *>(tile + (long)p.g1 * HW * (((long)p.g2 * HW) | 1));
  int *pairs = npairs + 1;
  if (tid == 0) *npairs = 0;
  __syncthreads();
  for (int e = tid; e < T * T; e += FOLD_T) {
    const int t1 = e / T, t2 = e - t1 * T;
    bool any = false;
    for (int pos = 0; pos < P && !any; ++pos) any = (qtab[t1 * P + pos] | qtab[t2 * P + pos]) >= 0;
    if (any) pairs[atomicAdd(npairs, 1)] = (t1 << 16) | t2;
  }
  __syncthreads();
  const int np = *npairs;
  if (p.sparse) {
    // C was zero-filled: only the pairs of the list are computed and written (scattered 4-byte stores, but few)
    const int nout = n1 * n2 * np;
    for (int o = tid; o < nout; o += FOLD_T) {
      const int c12 = o / np, pi = o - c12 * np;
      const int c1 = c12 / n2, c2 = c12 - c1 * n2;
      const int t1 = pairs[pi] >> 16, t2 = pairs[pi] & 0xffff;
      const int *q1p = qtab + t1 * P, *q2p = qtab + t2 * P;
      const float *t0 = tile + (long)c1 * HW * ldt + c2 * HW;
      float sacc = 0.f;
      for (int pos = 0; pos < P; ++pos) {
        const int q1 = q1p[pos], q2 = q2p[pos];
        const float v = t0[max(q1, 0) * ldt + max(q2, 0)];
        sacc += (q1 | q2) >= 0 ? v : 0.f;
      }
      p.C[((long)(c1_0 + c1) * T + t1) * p.ldc + (long)(c2_0 + c2) * T + t2] = p.alpha * sacc;
    }
    return;
  }
  // dense: thread = output column (c2, t2) of the tile's row (c1, t1), rows dealt to the thread rows of the block: the
  // index arithmetic of a column is done ONCE per thread, consecutive lanes write consecutive floats
  const int no2 = n2 * T, no1 = n1 * T;
  const int tcols = min(no2, FOLD_T), trows = FOLD_T / tcols;
  const int ty = tid / tcols, tx = tid - ty * tcols;
  if (ty < trows) {
    for (int o2 = tx; o2 < no2; o2 += tcols) {
      const int c2 = o2 / T, t2 = o2 - c2 * T;
      const int *q2p = qtab + t2 * P;
      for (int o1 = ty; o1 < no1; o1 += trows) {
        const int c1 = o1 / T, t1 = o1 - c1 * T;
        const int *q1p = qtab + t1 * P;
        const float *t0 = tile + (long)c1 * HW * ldt + c2 * HW;
        // branch-free gather (clamped index + select): the P reads of an output are independent and pipeline
        float sacc = 0.f;
#pragma unroll 8
        for (int pos = 0; pos < P; ++pos) {
          const int q1 = q1p[pos], q2 = q2p[pos];
          const float v = t0[max(q1, 0) * ldt + max(q2, 0)];
          sacc += (q1 | q2) >= 0 ? v : 0.f;
        }
        float *c = p.C + ((long)(c1_0 + c1) * T + t1) * p.ldc + (long)(c2_0 + c2) * T + t2;
        const float v = p.alpha * sacc;
        *c = p.beta != 0.f ? p.beta * *c + v : v;
      }
    }
  }
}

__global__ void fold_zero_kernel(float *__restrict__ C, long ldc, long d) {
  const long total = d * d;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    C[(e / d) * ldc + e % d] = 0.f;
}

}  // namespace clo

using namespace clo;

extern "C" int clo_patch_fold_supported(int Cc, int H, int W, int KH, int KW, int OH, int OW) {
  if (Cc < 1 || H < 1 || W < 1 || KH < 1 || KW < 1 || OH < 1 || OW < 1) return 0;
  const long HW = (long)H * W, P = (long)OH * OW, T = (long)KH * KW;
  // the tile of ONE channel pair and the tap table must fit the workgroup's LDS
  return ((T * P + 4) + HW * (HW + 1) + 1 + T * T) * 4 <= 64 * 1024 ? 1 : 0;
}

extern "C" int clo_patch_fold_f32(float *C, long ldc, const float *Gam, long ldg, const float *colsum, int B, int Cc,
                                  int H, int W, int KH, int KW, int SH, int SW, int PH, int PW, int DH, int DW,
                                  int OH, int OW, int ones_col, float alpha, float beta, void *stream) {
  CLO_REQUIRE(B >= 0 && Cc > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && SH > 0 && SW > 0 && DH > 0 && DW > 0 && PH >= 0 &&
                  PW >= 0,
              "clo_patch_fold_f32: bad geometry");
  CLO_REQUIRE(OH == (H + 2 * PH - DH * (KH - 1) - 1) / SH + 1 && OW == (W + 2 * PW - DW * (KW - 1) - 1) / SW + 1 &&
                  OH > 0 && OW > 0,
              "clo_patch_fold_f32: output size (%d, %d) inconsistent with the geometry", OH, OW);
  CLO_REQUIRE(clo_patch_fold_supported(Cc, H, W, KH, KW, OH, OW), "clo_patch_fold_f32: image too large for the LDS tile");
  const long d = (long)Cc * KH * KW + (ones_col ? 1 : 0);
  CLO_REQUIRE(C && Gam && ldc >= d && ldg >= (long)Cc * H * W && (!ones_col || colsum),
              "clo_patch_fold_f32: null operand / leading dimension too small");
  FoldArgs a{};
  a.C = C; a.ldc = ldc; a.G = Gam; a.ldg = ldg; a.colsum = colsum;
  a.Cc = Cc; a.H = H; a.W = W; a.KH = KH; a.KW = KW; a.SH = SH; a.SW = SW; a.PH = PH; a.PW = PW; a.DH = DH; a.DW = DW;
  a.OH = OH; a.OW = OW; a.ones = ones_col ? 1 : 0;
  a.nrows = (float)((double)B * OH * OW);
  a.alpha = alpha; a.beta = beta;
  // channel groups: the largest square-ish tile of Gam that fits 48 KB, shrunk until the grid fills the chip
  const long HW = (long)H * W, T = (long)KH * KW, P = (long)OH * OW;
  // (the tap table and the pair list share the workgroup's 64 KB with the tile: a 9 x 9 kernel on an 8 x 8 map needs 26 KB
  // for them, which a fixed 44 KB tile budget would push over the limit -- the tile grows only as far as the sum allows;
  // g1 = g2 = 1 always fits, that is what clo_patch_fold_supported checks)
  const long fixed_floats = ((T * P + 3) & ~3L) + 1 + T * T;
  const long budget = std::min<long>((44 * 1024) / 4, (64 * 1024) / 4 - fixed_floats);
  int g1 = 1, g2 = 1;
  auto fits = [&](int r, int c) { return (long)r * HW * (((long)c * HW) | 1) <= budget; };
  for (bool grew = true; grew;) {   // columns first: a tile row is one contiguous run of Gam
    grew = false;
    if (g2 * 2 <= Cc && fits(g1, g2 * 2)) { g2 *= 2; grew = true; }
    if (g1 * 2 <= Cc && fits(g1 * 2, g2)) { g1 *= 2; grew = true; }
  }
  while (cdiv(Cc, g1) * cdiv(Cc, g2) < 2L * kNumCU && (g1 > 1 || g2 > 1)) {
    if (g1 >= g2 && g1 > 1) g1 /= 2; else g2 /= 2;
  }
  a.g1 = g1; a.g2 = g2;
  // Tiny feature maps make A structurally sparse (a 1x1 map under a 3x3 kernel with padding 1: only the centre taps ever
  // see a pixel, 1 / 81 of the entries): with beta == 0 the matrix is then zero-filled by a memset and only tap pairs that
  // share an output position are written (ResNet-18 layer4: 85 MB of 4-byte scattered zero stores -> 20 us instead of 90).
  if (beta == 0.f) {
    long valid_pairs = 0;
    std::vector<unsigned char> ok((size_t)T * P);
    for (long t = 0; t < T; ++t)
      for (long pos = 0; pos < P; ++pos) {
        const long oh = pos / OW, ow = pos % OW, kh = t / KW, kw = t % KW;
        const long ih = oh * SH - PH + kh * DH, iw = ow * SW - PW + kw * DW;
        ok[t * P + pos] = ih >= 0 && ih < H && iw >= 0 && iw < W;
      }
    for (long t1 = 0; t1 < T; ++t1)
      for (long t2 = 0; t2 < T; ++t2) {
        bool any = false;
        for (long pos = 0; pos < P && !any; ++pos) any = ok[t1 * P + pos] && ok[t2 * P + pos];
        valid_pairs += any;
      }
    if (4 * valid_pairs < T * T) {
      a.sparse = 1;
      hipLaunchKernelGGL(fold_zero_kernel, dim3((unsigned)std::min<long>(cdiv(d * d, 1024), 8L * kNumCU)), dim3(256), 0,
                         (hipStream_t)stream, C, ldc, d);
      CLO_CHECK_LAUNCH("fold_zero_kernel");
    }
  }
  const size_t lds = (size_t)(((T * P + 3) & ~3L) + (long)g1 * HW * (((long)g2 * HW) | 1) + 1 + T * T) * sizeof(float);
  CLO_REQUIRE(lds <= 64 * 1024, "clo_patch_fold_f32: internal tile choice exceeds the LDS budget");
  dim3 grid((unsigned)cdiv(Cc, g2) + (ones_col ? 1 : 0), (unsigned)cdiv(Cc, g1));
  hipLaunchKernelGGL(patch_fold_kernel, grid, dim3(FOLD_T), lds, (hipStream_t)stream, a);
  CLO_CHECK_LAUNCH("patch_fold_kernel");
  return CLO_OK;
}

extern "C" int clo_im2col_f32(const float *x, float *out, int B, int C, int H, int W, int KH, int KW,
                              int SH, int SW, int PH, int PW, int DH, int DW, int OH, int OW,
                              void *stream) {
  CLO_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && SH > 0 && SW > 0 && DH > 0 &&
                  DW > 0 && PH >= 0 && PW >= 0,
              "clo_im2col_f32: bad geometry");
  CLO_REQUIRE(OH == (H + 2 * PH - DH * (KH - 1) - 1) / SH + 1 &&
                  OW == (W + 2 * PW - DW * (KW - 1) - 1) / SW + 1 && OH > 0 && OW > 0,
              "clo_im2col_f32: output size (%d, %d) inconsistent with the geometry", OH, OW);
  if (B == 0) return CLO_OK;
  CLO_REQUIRE(x && out, "clo_im2col_f32: null pointer");
  Im2colArgs a{x, out, B, C, H, W, KH, KW, SH, SW, PH, PW, DH, DW, OH, OW};
  const long total = (long)B * OH * OW * C * KH * KW;
  const unsigned grid = (unsigned)std::max<long>(1, std::min<long>(cdiv(total, 256), kNumCU * 16L));
  hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  CLO_CHECK_LAUNCH("im2col_kernel");
  return CLO_OK;
}
