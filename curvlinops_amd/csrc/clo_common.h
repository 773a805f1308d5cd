// Shared helpers for libclo_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/curvlinops_amd.h"

namespace clo {

void set_error(const char *fmt, ...);

inline int check_hip(hipError_t e, const char *what) {
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return CLO_EHIP;
  }
  return CLO_OK;
}

#define CLO_CHECK_LAUNCH(what)                                   \
  do {                                                           \
    int _rc = ::clo::check_hip(hipGetLastError(), what);         \
    if (_rc != CLO_OK) return _rc;                               \
  } while (0)

#define CLO_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::clo::set_error(__VA_ARGS__);      \
      return CLO_EINVAL;                  \
    }                                     \
  } while (0)

// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Tags: 0 fwd_jvp, 1 loss_hessian, 2 bwd_fused, 3 finish/reduce, 4 gemm, 5 other.
constexpr int kProfTags = 8;
bool prof_enabled();
void prof_begin(int tag, double alg_bytes, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
  hipStream_t st;
  bool on;
  ProfScope(int tag, double alg_bytes, hipStream_t s) : st(s), on(prof_enabled()) {
    if (on) prof_begin(tag, alg_bytes, st);
  }
  ~ProfScope() {
    if (on) prof_end(st);
  }
};

// ---- asynchronous faults of the kernels that wait on other workgroups (persistent grids, stream-K finishers) -----------
// A wait that runs out of its spin budget -- the grid is not co-resident: another process or a CU mask holds compute units
// -- does NOT trap (a trap is a sticky error of the whole HIP context).  The waiter raises the launch's device-side abort
// word (every other wait of that launch then returns at once: the kernel runs to its end on garbage, all addresses are
// shape-derived) and the device's fault word in host-pinned memory.  The NEXT call of the affected entry point sees the
// word, disables the mode on that device (persistent MLP kernel -> launch chain, stream-K -> split-K, tridiagonalisation
// -> half the workgroups) and returns CLO_EASYNC once, like HIP reports asynchronous errors: the results of the launch
// that timed out are invalid, the context stays healthy.
enum { FAULT_MEGA = 0, FAULT_SYTRD = 1, FAULT_STREAMK = 2, FAULT_KINDS = 3 };
unsigned *fault_words_device(int dev);        // [FAULT_KINDS] device-visible address of the host-pinned words (nullptr: unavailable)
bool fault_take(int dev, int kind);           // pending fault of `kind`: clears it, marks the mode disabled, returns true
bool fault_disabled(int dev, int kind);       // the mode was disabled after a fault on this device
unsigned spin_limit();                        // spin budget of every bounded wait (clo_test_set_spin_limit)
// `blocks` one-wave workgroups that each idle for `ticks` ticks of the 100 MHz wall clock (dispatch-bound for small `ticks`):
// the probe kernel of clo_test_occupy and of the helper-stream calibration in linalg.hip
int launch_occupy(int blocks, int lds_bytes, long ticks, hipStream_t st);

// Bookkeeping calls of the library (event queries of its stream / admission pools, creation of its helper streams and
// events) are harmless to a graph that ANOTHER thread is capturing in global mode, but the runtime would count them as
// "potentially unsafe" and invalidate that capture.  The calling thread switches to relaxed mode for their duration (the
// documented way, hipThreadExchangeStreamCaptureMode); its own captures are unaffected.
struct RelaxedCaptureScope {
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  RelaxedCaptureScope() { if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError(); }
  ~RelaxedCaptureScope() { if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError(); }
  RelaxedCaptureScope(const RelaxedCaptureScope &) = delete;
  RelaxedCaptureScope &operator=(const RelaxedCaptureScope &) = delete;
};

__host__ __device__ inline long cdiv(long a, long b) { return (a + b - 1) / b; }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;
constexpr int kNumCU = 256;   // MI355X
constexpr int kNumXCD = 8;

// Compute units of device `dev` as the runtime reports them (a partitioned or CU-masked part has fewer than kNumCU):
// what a persistent grid -- one workgroup per CU, all of them resident -- may count on.  Queried once per device.
inline int device_cu_count(int dev) {
  static int cached[64] = {0};
  const int slot = dev & 63;
  if (cached[slot] <= 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = kNumCU;
    cached[slot] = n;
  }
  return cached[slot];
}

// Sum over the 64 lanes of a wave; every lane gets the result.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- grid-level waits through the SCALAR memory path (round 5; profiles/r05_c2_scalar_seam.txt) ----
// A CU's vector-memory pipe returns in issue order, so a poll written as a vector load waits behind every byte the CU
// requested before it; s_load ... glc reaches L2 on the scalar cache's own path.  The whole WAVE runs the poll (scalar
// code); the counter must only grow while it is polled and whatever it guards must be read with device-scope (sc1)
// vector loads issued afterwards, so a scalar read can only be late, never wrong.  Every 64th spin reads the counter the
// architected way as well.  Same bounded-spin / abort protocol as the vector waits ("asynchronous faults" above).
__device__ __forceinline__ unsigned long uniform_ptr(const void *q) {
  const unsigned long v = (unsigned long)q;
  return ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
}
__device__ __forceinline__ void scalar_wait(unsigned *cnt, unsigned target, unsigned *err, unsigned *fault, unsigned limit,
                                            int lane) {
  const unsigned long c = uniform_ptr(cnt), e = uniform_ptr(err);
  unsigned spins = 0;
  for (;;) {
    unsigned seen;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seen) : "s"(c) : "memory");
    if (seen >= target) return;
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 63u) == 0u &&
        (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= target)
      return;
    if ((spins & 255u) == 0u) {
      unsigned bad;
      asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(bad) : "s"(e) : "memory");
      if (bad != 0u) return;
    }
    if (spins > limit) {   // ~seconds: the grid is not co-resident (or the counters were not initialised)
      if (lane == 0) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fault) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  }
}
// arrival of a whole wave through the scalar path (s_atomic_add, no return value)
__device__ __forceinline__ void scalar_arrive(unsigned *cnt) {
  const unsigned long c = uniform_ptr(cnt);
  const unsigned one = 1u;
  asm volatile("s_atomic_add %0, %1, 0x0" ::"s"(one), "s"(c) : "memory");
}

// sigma(x) (1 - sigma(x)) = e^-|x| / (1 + e^-|x|)^2: the product form loses all digits of 1 - sigma once sigma rounds to 1
// (|x| > 17 in float32: relative errors of 1e-2 in the BCE Hessian of saturated logits, tools/fuzz_native.py seed 500)
__device__ __forceinline__ float sigmoid_prime(float x) {
  const float e = __expf(-fabsf(x)), q = 1.f + e;
  return e / (q * q);
}

__device__ __forceinline__ float act_apply(int act, float z, float &dphi) {
  switch (act) {
    case CLO_ACT_RELU:
      dphi = z > 0.f ? 1.f : 0.f;
      return z > 0.f ? z : 0.f;
    case CLO_ACT_TANH: {
      float t = tanhf(z);
      dphi = 1.f - t * t;
      return t;
    }
    case CLO_ACT_SIGMOID: {
      float s = 1.f / (1.f + __expf(-z));
      dphi = sigmoid_prime(z);
      return s;
    }
    default:
      dphi = 1.f;
      return z;
  }
}

}  // namespace clo
