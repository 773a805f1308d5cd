// Scratch: phase timestamps (100 MHz wall clock) inside the Cholesky leaf kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I curvlinops_amd/csrc tools/ubench/potrf_leaf.hip \
//         -L curvlinops_amd/lib -lclo_hip -Wl,-rpath,$PWD/curvlinops_amd/lib -o /tmp/potrf_leaf
#define CLO_POTRF_DBG 1
#define potrf_diag_kernel potrf_diag_kernel_dbg  // keep the library's own kernel out of the way
#include "../../curvlinops_amd/csrc/linalg.hip"
#include <cstdio>
#include <cmath>
#include <vector>
int main() {
  const int n = 64;
  std::vector<float> h(n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) h[i * n + j] = (i == j ? 2.f : 0.f) + 1.f / (1 + i + j);
  float *A, *L, *Li; int *st;
  hipMalloc(&A, n * n * 4); hipMalloc(&L, n * n * 4); hipMalloc(&Li, n * n * 4); hipMalloc(&st, 4);
  hipMemcpy(A, h.data(), n * n * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(clo::potrf_diag_kernel, dim3(1), dim3(64), 0, 0, A, (long)n, L, (long)n, n, Li, (long)n, st, 0, 0L);
    hipError_t e0 = hipGetLastError();
    hipError_t e1 = hipDeviceSynchronize();
    float l00 = 0; int hst = -1;
    (void)hipMemcpy(&l00, L, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&hst, st, 4, hipMemcpyDeviceToHost);
    printf("launch: %s, L[0][0] = %f (expect %f), status %d\n", hipGetErrorString(e0), l00, sqrtf(3.f), hst);
    long long t[16];
    hipError_t e2 = hipMemcpyFromSymbol(t, HIP_SYMBOL(clo::potrf_dbg), sizeof(t));
    if (e1 != hipSuccess || e2 != hipSuccess) printf("errors: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    printf("rep %d (10 ns ticks): load %lld diag0 %lld panel0 %lld trail0 %lld rest-of-factor %lld inverse %lld store %lld total %lld\n",
           rep, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[7] - t[0]);
  }
  return 0;
}
