// Scratch micro-benchmark (round 4, sytrd hand-off): what does it cost 256 workgroups to ADD their 264 partial sums into one
// record per 16-workgroup group with float atomics (agent scope, no return) and wait for the acknowledgements -- against
// storing private records (sc1) and waiting?  iters back-to-back rounds per launch, time per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
constexpr int NP = 264, GRP = 16;
template <int MODE>   // 0: private sc1 stores, 1: float atomics into the group record, 2: atomics into ONE record
__global__ __launch_bounds__(512) void k(float *rec, int iters, unsigned long long *cyc) {
  const int tid = threadIdx.x, g = blockIdx.x / GRP;
  unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (tid < NP) {
      const float v = 1e-3f * (tid + it);
      if (MODE == 0) __builtin_nontemporal_store(v, rec + (long)blockIdx.x * 320 + tid);
      else if (MODE == 1) __hip_atomic_fetch_add(rec + (long)(g * 3 + it % 3) * 320 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(rec + (long)(it % 3) * 320 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (tid == 0) cyc[blockIdx.x] = wall_clock64() - t0;
}
int main() {
  float *rec; unsigned long long *cyc; CK(hipMalloc(&rec, 1 << 22)); CK(hipMemset(rec, 0, 1 << 22)); CK(hipMalloc(&cyc, 256 * 8));
  unsigned long long h[256];
  const int iters = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, rec, iters, cyc);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, rec, iters, cyc);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, rec, iters, cyc);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mx = 0, mean = 0; for (int i = 0; i < 256; ++i) { mx = h[i] > mx ? h[i] : mx; mean += h[i] / 256.0; }
    const char *names[3] = {"private records, stores + ack", "float atomics into 16 group records + ack", "float atomics into ONE record + ack"};
    printf("%-48s: %.2f us per round (mean over workgroups), %.2f max   [100 MHz wall clock]\n", names[mode], mean / iters / 100.0, mx / iters / 100.0);
  }
  return 0;
}
