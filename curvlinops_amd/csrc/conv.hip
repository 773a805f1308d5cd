// Patch extraction for KFAC's input covariance of Conv2d layers (reference
// kfac_utils.py:78-121: group-mean + torch unfold + transpose).  One launch for the whole
// mini-batch (torch's unfold launches one im2col kernel per sample), written in the layout the
// SYRK wants: out[(b, oh, ow)][(c, kh, kw)] row-major, coalesced along the patch axis.
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

}  // namespace clo

using namespace clo;

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
