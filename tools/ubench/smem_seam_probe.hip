// Round 5 micro-benchmark: can the exchange seam of a persistent kernel travel through the SCALAR memory path?
// A CU's vector-memory pipe returns loads in issue order, so the poll and the gather of a seam queue behind every weight
// byte the compute waves requested before them (csrc/mlp_mega.hip paces its tile requests for that reason).  Scalar loads
// (s_load ... glc) use the scalar data cache's own path to L2.  256 workgroups x 512 threads (one per CU); per iteration
// waves 0..6 request NLOAD x 16 B per lane of "weights", wave 7 publishes 640 B (sc1 stores, drain, relaxed agent-scope
// arrive), waits for its 16-member group and gathers 16 x 640 B -- MODE 0: atomic-load poll + sc1 vector loads (as the
// kernel does), MODE 1: s_load glc poll + s_load_dwordx16 glc gather.  SAMEXCD 1: the group's members sit on one XCD (the
// kernel's column groups), 0: on all eight.  Every gathered 64-byte chunk is verified (stale data shows up in `bad`).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 smem_seam_probe.hip -o smem_seam_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int G = 256, T = 512, SLOT = 160;   // floats per slot (640 B)
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ unsigned long uniform64(const void *p) {
  const unsigned long v = (unsigned long)p;
  return ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
}

template <int MODE, int NLOAD, int SAMEXCD>
__global__ __launch_bounds__(T) void probe(const f32x4v *__restrict__ big, long big_elems, float *xch, unsigned *cnt, int iters,
                                           unsigned long long *stamps, unsigned *bad, float *sink) {
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = SAMEXCD ? (w & 7) * 2 + ((w >> 3) & 1) : w >> 4;
  const int me = SAMEXCD ? w >> 4 : w & 15;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xch, 0, 2 * G * SLOT * 4, 0x00020000);
  float carry = 0.f;
  unsigned long long t_seam = 0, t_iter = 0;
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    if (wave < 7) {
      f32x4v v[NLOAD > 0 ? NLOAD : 1];
      const long base = ((long)(it & 7) * G + w) * (7L * 64 * NLOAD) % (big_elems - 7L * 64 * NLOAD);
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) v[i] = big[base + ((long)i * 7 + wave) * 64 + lane];
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) acc += v[i];
    } else {
      const int set = it & 1;
      const long mine = ((long)set * G + grp * 16 + me) * SLOT;
      if (lane < SLOT / 4) {
        f32x4v v = {__int_as_float(it + 1), (float)w, (float)lane, carry};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)((mine + lane * 4) * 4), 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t1 = wall_clock64();
      unsigned *c = cnt + 32 * grp;
      const unsigned target = 16u * (it + 1);
      if (MODE == 0) {
        if (lane == 0) {
          __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned spins = 0;
          while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_wave_barrier();
        f32x4v gsum = {0.f, 0.f, 0.f, 0.f};
        f32x4v gv[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {   // 16 peers x 40 float4 = 640 float4 = 10 per lane
          const int e = i * 64 + lane, peer = e / 40, q = e % 40;
          const long src = ((long)set * G + grp * 16 + peer) * SLOT + q * 4;
          gv[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(src * 4), 0, 16));
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          if (__float_as_int(gv[i].x) != it + 1) ++nbad;
          gsum += gv[i];
        }
        carry = gsum.w * 1e-9f;
      } else {
        if (lane == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned seen = 0, spins = 0;
        do {   // scalar poll: s_load glc bypasses the scalar data cache
          asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seen) : "s"(uniform64(c)) : "memory");
          if (seen < target) __builtin_amdgcn_s_sleep(1);
        } while (seen < target && ++spins < SPIN_LIMIT);
        int chk = 0;
        const unsigned long gbase = uniform64(xch + ((long)set * G + grp * 16) * SLOT);   // the group's 16 slots are contiguous: 10 KB
#pragma unroll 4
        for (int i = 0; i < 160; i += 4) {   // 160 chunks of 64 B, four loads in flight
          i32x16 a, b, cc, d;
          asm volatile("s_load_dwordx16 %0, %4, %5 glc\n\ts_load_dwordx16 %1, %4, %6 glc\n\ts_load_dwordx16 %2, %4, %7 glc\n\t"
                       "s_load_dwordx16 %3, %4, %8 glc\n\ts_waitcnt lgkmcnt(0)"
                       : "=&s"(a), "=&s"(b), "=&s"(cc), "=&s"(d)
                       : "s"(gbase), "s"(__builtin_amdgcn_readfirstlane(i * 64)), "s"(__builtin_amdgcn_readfirstlane(i * 64 + 64)),
                         "s"(__builtin_amdgcn_readfirstlane(i * 64 + 128)), "s"(__builtin_amdgcn_readfirstlane(i * 64 + 192))
                       : "memory");
          // every 64-byte chunk holds four float4 {it + 1, w, lane, carry}: word 0 of each quarter is the iteration tag
          chk += (a[0] != it + 1) + (a[4] != it + 1) + (a[8] != it + 1) + (a[12] != it + 1);
          chk += (b[0] != it + 1) + (b[4] != it + 1) + (b[8] != it + 1) + (b[12] != it + 1);
          chk += (cc[0] != it + 1) + (cc[4] != it + 1) + (cc[8] != it + 1) + (cc[12] != it + 1);
          chk += (d[0] != it + 1) + (d[4] != it + 1) + (d[8] != it + 1) + (d[12] != it + 1);
        }
        if (lane == 0) nbad += chk;
      }
      t_seam += wall_clock64() - t1;
    }
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) sink[tid] = acc.x;
    __syncthreads();
    t_iter += wall_clock64() - t0;
  }
  if (wave == 7 && lane == 0) {
    stamps[2 * w] = t_seam;
    if (nbad) atomicAdd(bad, nbad);
  }
  if (tid == 0) stamps[2 * w + 1] = t_iter;
  if (carry == 123.f) sink[0] = carry;
}

template <int MODE, int NLOAD, int SAMEXCD>
static void run(const char *name, f32x4v *big, long big_elems, float *xch, unsigned *cnt, unsigned long long *stamps, unsigned *bad,
                float *sink) {
  const int iters = 200;
  CK(hipMemset(cnt, 0, 64 * 32 * 4));
  CK(hipMemset(bad, 0, 4));
  CK(hipMemset(xch, 0, 2 * G * SLOT * 4));
  hipLaunchKernelGGL((probe<MODE, NLOAD, SAMEXCD>), dim3(G), dim3(T), 0, 0, big, big_elems, xch, cnt, iters, stamps, bad, sink);
  CK(hipDeviceSynchronize());
  unsigned long long h[2 * G];
  unsigned hb = 0;
  CK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
  CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
  double seam = 0, iter = 0;
  for (int w = 0; w < G; ++w) { seam += h[2 * w]; iter += h[2 * w + 1]; }
  printf("%-46s seam (publish acknowledged -> gathered) %6.2f us | iteration %6.2f us | stale / wrong chunks %u\n", name,
         seam / G / iters * 0.01, iter / G / iters * 0.01, hb);
}

int main() {
  const long big_elems = 1L << 27;   // 2 GiB of float4: nothing stays cache resident
  f32x4v *big; float *xch, *sink; unsigned *cnt, *bad; unsigned long long *stamps;
  CK(hipMalloc(&big, big_elems * 16)); CK(hipMemset(big, 0, big_elems * 16));
  CK(hipMalloc(&xch, 2 * G * SLOT * 4)); CK(hipMalloc(&sink, T * 4)); CK(hipMalloc(&cnt, 64 * 32 * 4)); CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&stamps, 2 * G * 8));
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0, 1>("vector seam, no weight traffic, same XCD", big, big_elems, xch, cnt, stamps, bad, sink);
    run<1, 0, 1>("scalar seam, no weight traffic, same XCD", big, big_elems, xch, cnt, stamps, bad, sink);
    run<0, 33, 1>("vector seam behind 236 KB of loads, same XCD", big, big_elems, xch, cnt, stamps, bad, sink);
    run<1, 33, 1>("scalar seam beside 236 KB of loads, same XCD", big, big_elems, xch, cnt, stamps, bad, sink);
    run<0, 33, 0>("vector seam behind 236 KB of loads, 8 XCDs", big, big_elems, xch, cnt, stamps, bad, sink);
    run<1, 33, 0>("scalar seam beside 236 KB of loads, 8 XCDs", big, big_elems, xch, cnt, stamps, bad, sink);
  }
  return 0;
}
