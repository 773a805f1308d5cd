// Round 4 micro-benchmark: two ways to build a 16-workgroup exchange seam inside a persistent launch (MI355X,
// 256 workgroups of 512 threads, one per CU).
//   counter  : 16-byte sc1 stores -> every storing wave drains vmcnt -> barrier -> one relaxed agent-scope atomicAdd
//              on the group's counter -> one lane polls it -> sc1 loads            (the seam of csrc/mlp_mega.hip, round 3)
//   sentinel : 16-byte sc1 stores, NO drain, NO counter; the consumers load the payload itself (sc1) and retry until no
//              word carries the sentinel bit pattern; each producer re-arms its slot of the set two iterations ahead
//              (three rotating sets, so that a slow reader of the previous iteration is never overwritten)
// payload as in the kernel: 10 KB published per workgroup, 16 x 640 B read back; every word is verified.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 seam_probe.hip -o seam_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int G = 256, T = 512;
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr unsigned SENT = 0xffffffffu;
constexpr int SLOT = 2560;  // floats per workgroup slot (10 KB)

__device__ __forceinline__ bool wait_ge(unsigned *cnt, unsigned target, unsigned *err) {
  unsigned spins = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > SPIN_LIMIT) { *err = 1; return false; }
  }
  return true;
}

__global__ __launch_bounds__(T) void seam_counter(float *buf, unsigned *cnt, int iters, unsigned *err, unsigned *bad) {
  extern __shared__ float smem[];
  const int w = blockIdx.x, grp = w >> 4, me = w & 15, tid = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, G * SLOT * 4 * 3, 0x00020000);
  float carry = 0.f;
  for (int it = 0; it < iters; ++it) {
    const size_t mine = (size_t)((it & 1) * G + w) * SLOT;
    for (int e = tid; e < 640; e += T) {
      f32x4v v = {(float)(it + 1), (float)w, (float)e, carry};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)(mine * 4 + e * 16), 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt + 32 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_ge(cnt + 32 * grp, 16u * (it + 1), err);
    }
    __syncthreads();
    float s = 0.f;
    for (int e = tid; e < 640; e += T) {
      const int peer = e / 40, q = e % 40;
      const size_t src = (size_t)((it & 1) * G + grp * 16 + peer) * SLOT + (me * 40 + q) * 4;
      f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(src * 4), 0, 16));
      if (v.x != (float)(it + 1) || v.y != (float)(grp * 16 + peer) || v.z != (float)(me * 40 + q)) atomicAdd(bad, 1u);
      s += v.w;
    }
    carry = s * 1e-9f;
  }
  if (carry == 123.f) buf[0] = carry;
}

// SLEEP: s_sleep argument between retries; HINT: poll one word per peer first (one lane per peer), then the payload
template <int SLEEP, bool HINT>
__global__ __launch_bounds__(T) void seam_sentinel(float *buf, int iters, unsigned *err, unsigned *bad, unsigned *retries) {
  extern __shared__ float smem[];
  const int w = blockIdx.x, grp = w >> 4, me = w & 15, tid = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, G * SLOT * 4 * 3, 0x00020000);
  float carry = 0.f;
  unsigned nretry = 0;
  for (int it = 0; it < iters; ++it) {
    const int set = it % 3, nxt = (it + 1) % 3;
    const size_t mine = (size_t)(set * G + w) * SLOT, rearm = (size_t)(nxt * G + w) * SLOT;
    for (int e = tid; e < 640; e += T) {
      f32x4v v = {(float)(it + 1), (float)w, (float)e, carry};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)(mine * 4 + e * 16), 0, 16);
    }
    for (int e = tid; e < 640; e += T)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{SENT, SENT, SENT, SENT}, rs, (unsigned)(rearm * 4 + e * 16), 0, 16);
    if (HINT) {
      if (tid < 16) {
        const size_t src = (size_t)(set * G + grp * 16 + tid) * SLOT + (me * 40 + 39) * 4;  // last word of the slice
        unsigned spins = 0;
        while (__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(src * 4 + 12), 0, 16) == SENT) {
          __builtin_amdgcn_s_sleep(SLEEP);
          if (++spins > SPIN_LIMIT) { *err = 1; break; }
        }
      }
      __syncthreads();
    }
    float s = 0.f;
    for (int e = tid; e < 640; e += T) {
      const int peer = e / 40, q = e % 40;
      const size_t src = (size_t)(set * G + grp * 16 + peer) * SLOT + (me * 40 + q) * 4;
      u32x4 r;
      unsigned spins = 0;
      while (true) {
        r = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(src * 4), 0, 16);
        if (r.x != SENT && r.y != SENT && r.z != SENT && r.w != SENT) break;
        ++nretry;
        __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > SPIN_LIMIT) { *err = 1; break; }
      }
      f32x4v v = __builtin_bit_cast(f32x4v, r);
      if (v.x != (float)(it + 1) || v.y != (float)(grp * 16 + peer) || v.z != (float)(me * 40 + q)) atomicAdd(bad, 1u);
      s += v.w;
    }
    carry = s * 1e-9f;
    __syncthreads();  // (the kernel has a workgroup barrier after every gather as well)
  }
  if (nretry) atomicAdd(retries, nretry);
  if (carry == 123.f) buf[0] = carry;
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
  unsigned *cnt, *err, *bad, *retries;
  CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&retries, 4));
  CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(retries, 0, 4));
  float *buf;
  const size_t bytes = (size_t)G * SLOT * 4 * 3;
  CK(hipMalloc(&buf, bytes));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  CK(hipFuncSetAttribute((const void *)seam_counter, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  float base[2];
  int k = 0;
  for (int iters : {1, 51}) {
    float us = time_us([&] {
      CK(hipMemsetAsync(cnt, 0, 4096 * 4));
      hipLaunchKernelGGL(seam_counter, dim3(G), dim3(T), LDS, 0, buf, cnt, iters, err, bad);
    }, 20);
    base[k++] = us;
    printf("counter  seam iters %2d: %8.2f us per launch\n", iters, us);
  }
  printf("counter  seam: %.2f us per seam\n", (base[1] - base[0]) / 50);
  auto run = [&](auto kern, const char *name) {
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    float t[2];
    int j = 0;
    for (int iters : {1, 51}) {
      CK(hipMemset(retries, 0, 4));
      float us = time_us([&] {
        CK(hipMemsetAsync(buf, 0xff, bytes));
        hipLaunchKernelGGL(kern, dim3(G), dim3(T), LDS, 0, buf, iters, err, bad, retries);
      }, 20);
      t[j++] = us;
      unsigned hr = 0;
      CK(hipMemcpy(&hr, retries, 4, hipMemcpyDeviceToHost));
      printf("%s iters %2d: %8.2f us per launch (incl. a %.1f MB memset), %.0f retried loads per launch\n", name, iters, us,
             bytes / 1e6, hr / 21.0);
    }
    printf("%s: %.2f us per seam\n", name, (t[1] - t[0]) / 50);
  };
  run(seam_sentinel<1, false>, "sentinel sleep1       ");
  run(seam_sentinel<4, false>, "sentinel sleep4       ");
  run(seam_sentinel<1, true>, "sentinel sleep1 + hint");
  run(seam_sentinel<4, true>, "sentinel sleep4 + hint");
  unsigned herr = 0, hbad = 0;
  CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
  printf("timeouts %u, stale / wrong words %u\n", herr, hbad);
  return 0;
}
