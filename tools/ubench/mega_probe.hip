// Round 3 micro-benchmarks behind the persistent C2 kernel (MI355X, 256 workgroups of 512 threads, one per CU):
//   1 barriers : 16-workgroup group barrier / flat 256 barrier / group + leaders hierarchical
//   2 seam     : publish 10 KB per workgroup (16-byte sc1 stores), group barrier, read 16 x 640 B (sc1 loads), verified
//   3 stream   : per-CU weight stream with every load of a workgroup in flight at once (312 KB per CU)
//   4 stores   : 156 KB of non-temporal stores per CU
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mega_probe.hip -o mega_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int G = 256, T = 512;
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ bool wait_ge(unsigned *cnt, unsigned target, unsigned *err) {
  unsigned spins = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > SPIN_LIMIT) { *err = 1; return false; }
  }
  return true;
}

// mode 0: 16 groups of 16; 1: flat 256; 2: group -> leader arrives on top counter -> everybody polls top
__global__ __launch_bounds__(T) void barrier_kernel(unsigned *cnt, int iters, int mode, unsigned *err) {
  extern __shared__ float smem[];
  smem[threadIdx.x] = 0.f;
  const int w = blockIdx.x, grp = w >> 4;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (mode == 0) {
        __hip_atomic_fetch_add(cnt + 32 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!wait_ge(cnt + 32 * grp, 16u * (it + 1), err)) iters = 0;
      } else if (mode == 1) {
        __hip_atomic_fetch_add(cnt + 32 * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!wait_ge(cnt + 32 * 16, 256u * (it + 1), err)) iters = 0;
      } else {
        __hip_atomic_fetch_add(cnt + 32 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((w & 15) == 0) {
          if (!wait_ge(cnt + 32 * grp, 16u * (it + 1), err)) iters = 0;
          __hip_atomic_fetch_add(cnt + 32 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!wait_ge(cnt + 32 * 17, 16u * (it + 1), err)) iters = 0;
      }
    }
    iters = __shfl(iters, 0);  // (only wave 0 sees a timeout; good enough for a probe)
  }
}

// every workgroup publishes 10 KB (sc1, 16 B per lane), arrives on its group's counter, waits, reads 640 B from each
// of the 16 peers (sc1) and checks the epoch stamp; DEP = 1 makes iteration i+1's payload depend on what was read
__global__ __launch_bounds__(T) void seam_kernel(float *buf, unsigned *cnt, int iters, unsigned *err, unsigned *bad) {
  extern __shared__ float smem[];
  const int w = blockIdx.x, grp = w >> 4, me = w & 15, tid = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 256 * 10240 * 2, 0x00020000);
  float carry = 0.f;
  for (int it = 0; it < iters; ++it) {
    float *mine = buf + (size_t)((it & 1) * 256 + w) * 2560;  // 10 KB = 2560 floats; two alternating sets
    // publish: 640 lanes x 16 B -> 512 threads take 1..2 float4
    for (int e = tid; e < 640; e += T) {
      f32x4v v = {(float)(it + 1), (float)w, (float)e, carry};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                             (unsigned)((mine - buf) * 4 + e * 16), 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt + 32 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_ge(cnt + 32 * grp, 16u * (it + 1), err);
    }
    __syncthreads();
    // read slice `me` (40 float4 = 640 B) of each of the 16 peers: 640 float4 in total
    float s = 0.f;
    for (int e = tid; e < 640; e += T) {
      const int peer = e / 40, q = e % 40;
      const float *src = buf + (size_t)((it & 1) * 256 + grp * 16 + peer) * 2560 + (me * 40 + q) * 4;
      u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((src - buf) * 4), 0, 16);
      f32x4v v = __builtin_bit_cast(f32x4v, r);
      if (v.x != (float)(it + 1) || v.y != (float)(grp * 16 + peer) || v.z != (float)(me * 40 + q)) atomicAdd(bad, 1u);
      s += v.w;
    }
    carry = s * 1e-9f;
  }
  if (carry == 123.f) buf[0] = carry;
}

// per-CU stream: each wave issues NL float4 loads (1 KB each) back to back, then sums
template <int NL>
__global__ __launch_bounds__(T) void stream_kernel(const float *__restrict__ src, float *out, size_t per_wg_floats) {
  extern __shared__ float smem[];
  const int w = blockIdx.x, tid = threadIdx.x;
  const float *p = src + (size_t)w * per_wg_floats + tid * 4;
  float4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) v[i] = *reinterpret_cast<const float4 *>(p + (size_t)i * T * 4);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
  if (s == 1234.5f) out[w] = s;
}

template <int NS>
__global__ __launch_bounds__(T) void store_kernel(float *dst, size_t per_wg_floats, float val) {
  extern __shared__ float smem[];
  const int w = blockIdx.x, tid = threadIdx.x;
  float *p = dst + (size_t)w * per_wg_floats + tid * 4;
  typedef float __attribute__((ext_vector_type(4))) v4;
#pragma unroll
  for (int i = 0; i < NS; ++i) __builtin_nontemporal_store(v4{val, val, val, val}, reinterpret_cast<v4 *>(p + (size_t)i * T * 4));
}

template <typename F>
static float time_us(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return 1e3f * ms / reps;
}

int main() {
  const size_t LDS = 100 * 1024;  // forces one workgroup per CU
  CK(hipFuncSetAttribute((const void *)barrier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void *)seam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  unsigned *cnt, *err, *bad;
  CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4));
  CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 4));
  float *buf;
  CK(hipMalloc(&buf, 256 * 10240 * 2));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  for (int mode = 0; mode < 3; ++mode) {
    for (int iters : {1, 101}) {
      float us = time_us([&] {
        CK(hipMemsetAsync(cnt, 0, 4096 * 4));
        hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(T), LDS, 0, cnt, iters, mode, err);
      }, 20);
      printf("barrier mode %d iters %3d: %8.2f us per launch\n", mode, iters, us);
    }
  }
  for (int iters : {1, 51}) {
    float us = time_us([&] {
      CK(hipMemsetAsync(cnt, 0, 4096 * 4));
      hipLaunchKernelGGL(seam_kernel, dim3(G), dim3(T), LDS, 0, buf, cnt, iters, err, bad);
    }, 20);
    printf("seam (10 KB publish + group barrier + 16 x 640 B read) iters %2d: %8.2f us per launch\n", iters, us);
  }
  unsigned herr = 0, hbad = 0;
  CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
  printf("timeouts %u, stale words %u\n", herr, hbad);
  // ---- streams: rotate through 2 GiB so that nothing is cache resident
  const size_t pool = (size_t)2 << 30;
  float *big;
  CK(hipMalloc(&big, pool));
  CK(hipMemset(big, 0, pool));
  float *out;
  CK(hipMalloc(&out, 4096));
  auto run_stream = [&](auto kern, int NL, const char *name) {
    const size_t per_wg = (size_t)NL * T * 4;             // floats
    const size_t launch_bytes = per_wg * 4 * G;
    const int slots = (int)(pool / launch_bytes);
    int slot = 0;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    float us = time_us([&] {
      hipLaunchKernelGGL(kern, dim3(G), dim3(T), LDS, 0, big + (size_t)slot * launch_bytes / 4, out, per_wg);
      slot = (slot + 1) % slots;
    }, 40);
    printf("%s: %d loads per lane in flight, %.1f KB per CU, %.1f MB per launch: %7.2f us = %.2f TB/s = %.1f GB/s per CU\n",
           name, NL, per_wg * 4 / 1024.0, launch_bytes / 1e6, us, launch_bytes / us / 1e6, launch_bytes / us / 1e3 / G);
  };
  run_stream(stream_kernel<11>, 11, "stream");
  run_stream(stream_kernel<22>, 22, "stream");
  run_stream(stream_kernel<39>, 39, "stream");
  auto run_store = [&](auto kern, int NS, const char *name) {
    const size_t per_wg = (size_t)NS * T * 4;
    const size_t launch_bytes = per_wg * 4 * G;
    const int slots = (int)(pool / launch_bytes);
    int slot = 0;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    float us = time_us([&] {
      hipLaunchKernelGGL(kern, dim3(G), dim3(T), LDS, 0, big + (size_t)slot * launch_bytes / 4, per_wg, 1.f);
      slot = (slot + 1) % slots;
    }, 40);
    printf("%s: %d nt stores per lane, %.1f KB per CU, %.1f MB per launch: %7.2f us = %.2f TB/s = %.1f GB/s per CU\n",
           name, NS, per_wg * 4 / 1024.0, launch_bytes / 1e6, us, launch_bytes / us / 1e6, launch_bytes / us / 1e3 / G);
  };
  run_store(store_kernel<6>, 6, "store");
  run_store(store_kernel<14>, 14, "store");
  run_store(store_kernel<20>, 20, "store");
  return 0;
}
