// Error state, version, and the HBM-bound streaming helpers of libclo_hip:
// axpby (batch accumulate), transpose ([D,K] <-> [K,D]), row scaling (eigenvalue
// scaling of EighDecomposed operators) and counter-based probe packing.
#include "clo_common.h"

#include <mutex>
#include <vector>

namespace clo {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- asynchronous fault words (clo_common.h) ------------------------------------------------------------------------
namespace {
struct FaultState {
  unsigned *host = nullptr, *dev = nullptr;
  bool tried = false;
  bool disabled[FAULT_KINDS] = {false, false, false};
};
FaultState g_fault[64];
std::mutex g_fault_mu;
unsigned g_spin_limit = 1u << 22;
FaultState &fault_state(int dev) {
  FaultState &f = g_fault[dev & 63];
  if (!f.tried) {
    std::lock_guard<std::mutex> lock(g_fault_mu);
    if (!f.tried) {
      void *h = nullptr, *d = nullptr;
      if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
        f.host = static_cast<unsigned *>(h);
        f.dev = static_cast<unsigned *>(d);
        for (int i = 0; i < 16; ++i) f.host[i] = 0u;
      } else {
        (void)hipGetLastError();
      }
      f.tried = true;
    }
  }
  return f;
}
}  // namespace
unsigned *fault_words_device(int dev) { return fault_state(dev).dev; }
bool fault_take(int dev, int kind) {
  FaultState &f = fault_state(dev);
  if (!f.host || __atomic_load_n(&f.host[kind], __ATOMIC_RELAXED) == 0u) return false;
  __atomic_store_n(&f.host[kind], 0u, __ATOMIC_RELAXED);
  f.disabled[kind] = true;
  return true;
}
bool fault_disabled(int dev, int kind) { return g_fault[dev & 63].disabled[kind]; }
unsigned spin_limit() { return g_spin_limit; }

__global__ void occupy_kernel(long ticks) {
  extern __shared__ float occ_lds[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) occ_lds[0] = 0.f;
  while ((long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
}

int launch_occupy(int blocks, int lds_bytes, long ticks, hipStream_t st) {
  if (lds_bytes > 64 * 1024) {
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(occupy_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes),
                       "occupy_kernel: LDS attribute");
    if (rc != CLO_OK) return rc;
  }
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(64), (size_t)lds_bytes, st, ticks);
  return check_hip(hipGetLastError(), "occupy_kernel");
}

// ---- event-based kernel timing -------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; int tag; double bytes; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<ProfRec> g_prof_pool;
bool prof_enabled() { return g_prof_on; }
void prof_begin(int tag, double alg_bytes, hipStream_t st) {
  ProfRec r;
  if (!g_prof_pool.empty()) { r = g_prof_pool.back(); g_prof_pool.pop_back(); }
  else { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); }
  r.tag = tag; r.bytes = alg_bytes;
  (void)hipEventRecord(r.a, st);
  g_prof.push_back(r);
}
void prof_end(hipStream_t st) { (void)hipEventRecord(g_prof.back().b, st); }

// Grid for a streaming kernel over n work items of `per_thread` elements: cap at ~8 blocks
// per CU and grid-stride the rest.
static inline unsigned stream_grid(long n_items, int block) {
  long g = cdiv(n_items, block);
  const long cap = (long)kNumCU * 8;
  return (unsigned)std::max<long>(1, std::min<long>(g, cap));
}

// y = beta*y + alpha*x, 16 B per lane when aligned.
__global__ void axpby_vec_kernel(float4 *__restrict__ y, const float4 *__restrict__ x, long n4,
                                 float alpha, float beta) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x) {
    const float4 xv = x[i];
    float4 yv;
    if (beta != 0.f) {
      yv = y[i];
      yv.x = beta * yv.x + alpha * xv.x; yv.y = beta * yv.y + alpha * xv.y;
      yv.z = beta * yv.z + alpha * xv.z; yv.w = beta * yv.w + alpha * xv.w;
    } else {
      yv = make_float4(alpha * xv.x, alpha * xv.y, alpha * xv.z, alpha * xv.w);
    }
    y[i] = yv;
  }
}
__global__ void axpby_scalar_kernel(float *__restrict__ y, const float *__restrict__ x, long n,
                                    float alpha, float beta) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x)
    y[i] = (beta != 0.f ? beta * y[i] : 0.f) + alpha * x[i];
}

// 64x64 tile transpose through LDS (padded), coalesced on both sides.
__global__ void transpose_kernel(float *__restrict__ out, const float *__restrict__ in, long rows,
                                 long cols) {
  __shared__ float tile[64][65];
  const long tiles_c = cdiv(cols, 64);
  const long ntiles = cdiv(rows, 64) * tiles_c;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows per pass
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long r = r0 + ty + 4 * i, c = c0 + tx;
      tile[ty + 4 * i][tx] = (r < rows && c < cols) ? in[r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long c = c0 + ty + 4 * i, r = r0 + tx;
      if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + 4 * i];
    }
    __syncthreads();
  }
}

// Canonical pack / unpack of one (W, b) parameter group fused with the [D, K] <-> [K, D] layout change of the
// Kronecker matvec (reference kfac_utils.py:280-306 `cat` + 338-385 slicing, each followed by a transpose on this
// engine): the canonical matrix of the group is [rows][bw + 1] with the bias as its LAST column; its K tangents are
// stored K-trailing in parameter space (w [rows * bw][K], bias [rows][K]) and K-major in canonical space
// (kmaj [K][rows][bw + 1]), which is what the batched GEMMs of the block read.  One pass: a tile of TR canonical
// entries x <= 64 columns goes through LDS, both sides coalesced (parameter side along K then along the entries --
// the tile is one contiguous run of TR K floats except where a bias entry is spliced in --, canonical side along the
// entries).  bias == nullptr: a plain transpose of w.  PACK: parameter -> canonical; else the reverse.
constexpr int CP_TR = 128, CP_KC = 64;
template <bool PACK>
__global__ __launch_bounds__(256) void canonical_tr_kernel(float *__restrict__ kmaj, float *__restrict__ w,
                                                           float *__restrict__ bias, long rows, long bw, long K) {
  __shared__ float tile[CP_KC][CP_TR + 1];
  const long bc = bias ? bw + 1 : bw, nv = rows * bc;
  const long tiles_k = cdiv(K, (long)CP_KC), ntiles = cdiv(nv, (long)CP_TR) * tiles_k;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long e0 = (t / tiles_k) * CP_TR, k0 = (t % tiles_k) * CP_KC;
    const int kc = (int)min((long)CP_KC, K - k0), ne = (int)min((long)CP_TR, nv - e0);
    // parameter side: idx -> (entry, k), k fastest
    // (32-bit index arithmetic inside the tile: the 64-bit divisions happen once per tile; a power-of-two column count
    // splits by shift, and with >= 128 canonical columns a tile crosses at most one row boundary)
    const long r0 = e0 / bc;
    const unsigned c0 = (unsigned)(e0 - r0 * bc), ubc = (unsigned)min(bc, 0x7fffffffL), ukc = (unsigned)kc;
    const int kshift = (ukc & (ukc - 1)) == 0 ? __builtin_ctz(ukc) : -1;
    auto split = [&](int idx, unsigned &el, unsigned &k) {
      if (kshift >= 0) { el = (unsigned)idx >> kshift; k = (unsigned)idx & (ukc - 1); }
      else { el = (unsigned)idx / ukc; k = (unsigned)idx - el * ukc; }
    };
    auto param_ptr = [&](unsigned el, unsigned k) -> float * {
      if (!bias) return w + (e0 + el) * K + k0 + k;
      const unsigned cc = c0 + el, dr = ubc >= CP_TR ? (cc >= ubc ? 1u : 0u) : cc / ubc, c = cc - dr * ubc;
      const long r = r0 + dr;
      return (c < bw ? w + (r * bw + c) * K : bias + r * K) + k0 + k;
    };
    if (PACK) {
      _Pragma("unroll 8") for (int idx = threadIdx.x; idx < ne * kc; idx += 256) {
        unsigned el, k;
        split(idx, el, k);
        tile[k][el] = *param_ptr(el, k);
      }
    } else {
      _Pragma("unroll 8") for (int idx = threadIdx.x; idx < kc * CP_TR; idx += 256) {
        const int k = idx / CP_TR, el = idx - k * CP_TR;
        if (el < ne) tile[k][el] = kmaj[(k0 + k) * nv + e0 + el];
      }
    }
    __syncthreads();
    if (PACK) {
      _Pragma("unroll 8") for (int idx = threadIdx.x; idx < kc * CP_TR; idx += 256) {
        const int k = idx / CP_TR, el = idx - k * CP_TR;
        if (el < ne) kmaj[(k0 + k) * nv + e0 + el] = tile[k][el];
      }
    } else {
      _Pragma("unroll 8") for (int idx = threadIdx.x; idx < ne * kc; idx += 256) {
        unsigned el, k;
        split(idx, el, k);
        *param_ptr(el, k) = tile[k][el];
      }
    }
    __syncthreads();
  }
}

// y[i][k] = f(s[i]) * x[i][k]; f(s) = s, or 1/(s+shift) when reciprocal.
__global__ void rowscale_kernel(float *__restrict__ y, const float *__restrict__ x,
                                const float *__restrict__ s, long rows, long K, int reciprocal,
                                float shift) {
  const long total = rows * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long i = e / K;
    float sv = s[i];
    if (reciprocal) sv = 1.f / (sv + shift);
    y[e] = sv * x[e];
  }
}

// ---- Philox4x32-10 (Salmon et al. 2011), counter = element-group index, key = seed.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float u01(uint32_t x) {  // (0, 1]
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// Each thread produces 4 consecutive outputs from one Philox block.
__global__ void pack_probes_kernel(float *__restrict__ out, long n, uint64_t seed, int dist) {
  const long ngroups = cdiv(n, 4);
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
       g += (long)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)g, (uint32_t)((uint64_t)g >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float v[4];
    if (dist == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (c[e] & 1u) ? 1.f : -1.f;
    } else {  // Box-Muller, two pairs
      const float r0 = sqrtf(-2.f * __logf(u01(c[0]))), t0 = 6.28318530718f * u01(c[1]);
      const float r1 = sqrtf(-2.f * __logf(u01(c[2]))), t1 = 6.28318530718f * u01(c[3]);
      v[0] = r0 * __cosf(t0); v[1] = r0 * __sinf(t0);
      v[2] = r1 * __cosf(t1); v[3] = r1 * __sinf(t1);
    }
    const long base = g * 4;
    if (base + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
      *reinterpret_cast<float4 *>(out + base) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (base + e < n) out[base + e] = v[e];
    }
  }
}

// <x, y> over n elements: per-block partial sums (float4 loads, fixed grid-stride order), then one
// block adds the partials in double -- deterministic for a given n, any n (64-bit indices).
constexpr int DOT_BLOCKS = 2048;
__global__ __launch_bounds__(256) void dot_partial_kernel(const float *__restrict__ x,
                                                          const float *__restrict__ y, long n,
                                                          double *__restrict__ part) {
  __shared__ double s_w[4];
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
  const long n4 = vec ? n / 4 : 0;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 u = reinterpret_cast<const float4 *>(x)[i], v = reinterpret_cast<const float4 *>(y)[i];
    a0 = fmaf(u.x, v.x, a0); a1 = fmaf(u.y, v.y, a1); a2 = fmaf(u.z, v.z, a2); a3 = fmaf(u.w, v.w, a3);
  }
  double acc = ((double)a0 + a1) + ((double)a2 + a3);
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    acc += (double)x[i] * y[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
__global__ __launch_bounds__(256) void dot_final_kernel(const double *__restrict__ part, int nblocks,
                                                        float *__restrict__ out, float scale) {
  __shared__ double s_w[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += part[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(scale * ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])));
}

// ---- fused vector updates of conjugate gradients (single right-hand side) --------------------
// x += a p ; r -= a ap ; partial sums of r.r      with a = rz / pap read from device scalars
__global__ __launch_bounds__(256) void cg_update_kernel(float *__restrict__ x, float *__restrict__ r,
                                                        const float *__restrict__ p,
                                                        const float *__restrict__ ap, long n,
                                                        const float *__restrict__ rz,
                                                        const float *__restrict__ pap,
                                                        double *__restrict__ part) {
  __shared__ double s_w[4];
  const float den = pap[0];
  const float a = fabsf(den) > 1e-30f ? rz[0] / den : 0.f;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(r) |
                     reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(ap)) & 15u) == 0;
  const long n4 = vec ? n / 4 : 0;
  const long stride = (long)gridDim.x * blockDim.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 pv = reinterpret_cast<const float4 *>(p)[i], av = reinterpret_cast<const float4 *>(ap)[i];
    float4 xv = reinterpret_cast<float4 *>(x)[i], rv = reinterpret_cast<float4 *>(r)[i];
    xv.x = fmaf(a, pv.x, xv.x); xv.y = fmaf(a, pv.y, xv.y); xv.z = fmaf(a, pv.z, xv.z); xv.w = fmaf(a, pv.w, xv.w);
    rv.x = fmaf(-a, av.x, rv.x); rv.y = fmaf(-a, av.y, rv.y); rv.z = fmaf(-a, av.z, rv.z); rv.w = fmaf(-a, av.w, rv.w);
    reinterpret_cast<float4 *>(x)[i] = xv;
    reinterpret_cast<float4 *>(r)[i] = rv;
    s0 = fmaf(rv.x, rv.x, s0); s1 = fmaf(rv.y, rv.y, s1); s2 = fmaf(rv.z, rv.z, s2); s3 = fmaf(rv.w, rv.w, s3);
  }
  double acc = ((double)s0 + s1) + ((double)s2 + s3);
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    x[i] = fmaf(a, p[i], x[i]);
    const float rv = fmaf(-a, ap[i], r[i]);
    r[i] = rv;
    acc += (double)rv * rv;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
// p = z + (num / den) p
__global__ void cg_direction_kernel(float *__restrict__ p, const float *__restrict__ z, long n,
                                    const float *__restrict__ num, const float *__restrict__ den) {
  const float d = den[0];
  const float b = fabsf(d) > 1e-30f ? num[0] / d : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    p[i] = fmaf(b, p[i], z[i]);
}

}  // namespace clo

using namespace clo;

extern "C" int clo_version(void) { return 100; }

extern "C" int clo_prof_enable(int on) {
  g_prof_on = on != 0;
  return CLO_OK;
}
// Sum the recorded intervals per tag: ms[t], count[t], alg_bytes[t] (arrays of 8); clears them.
extern "C" int clo_prof_collect(double *ms, long *count, double *alg_bytes) {
  for (int t = 0; t < kProfTags; ++t) { ms[t] = 0; count[t] = 0; alg_bytes[t] = 0; }
  for (auto &r : g_prof) {
    int rc = check_hip(hipEventSynchronize(r.b), "hipEventSynchronize");
    if (rc != CLO_OK) return rc;
    float e = 0.f;
    rc = check_hip(hipEventElapsedTime(&e, r.a, r.b), "hipEventElapsedTime");
    if (rc != CLO_OK) return rc;
    if (r.tag >= 0 && r.tag < kProfTags) { ms[r.tag] += e; count[r.tag] += 1; alg_bytes[r.tag] += r.bytes; }
    g_prof_pool.push_back(r);
  }
  g_prof.clear();
  return CLO_OK;
}
extern "C" const char *clo_last_error(void) { return g_err; }

extern "C" int clo_persistent_status(int dev) {
  int bits = 0;
  for (int k = 0; k < FAULT_KINDS; ++k) {
    (void)fault_take(dev, k);
    if (fault_disabled(dev, k)) bits |= 1 << k;
  }
  return bits;
}
extern "C" int clo_fault_pending(int dev) {
  int bits = 0;
  const FaultState &f = g_fault[dev & 63];
  if (f.tried && f.host)
    for (int k = 0; k < FAULT_KINDS; ++k)
      if (__atomic_load_n(&f.host[k], __ATOMIC_RELAXED) != 0u) bits |= 1 << k;
  return bits;
}
extern "C" int clo_test_set_spin_limit(unsigned polls) {
  CLO_REQUIRE(polls >= 16, "clo_test_set_spin_limit: at least 16 polls");
  g_spin_limit = polls;
  return CLO_OK;
}
extern "C" int clo_test_occupy(int blocks, int lds_bytes, long ticks, void *stream) {
  CLO_REQUIRE(blocks >= 1 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && ticks >= 0, "clo_test_occupy: bad arguments");
  return launch_occupy(blocks, lds_bytes, ticks, (hipStream_t)stream);
}

extern "C" int clo_axpby_f32(float *y, const float *x, long n, float alpha, float beta,
                             void *stream) {
  CLO_REQUIRE(n >= 0, "clo_axpby_f32: negative n");
  if (n == 0) return CLO_OK;
  CLO_REQUIRE(y && x, "clo_axpby_f32: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (n % 4 == 0 && aligned16(y) && aligned16(x)) {
    hipLaunchKernelGGL(axpby_vec_kernel, dim3(stream_grid(n / 4, 256)), dim3(256), 0, st,
                       reinterpret_cast<float4 *>(y), reinterpret_cast<const float4 *>(x), n / 4,
                       alpha, beta);
  } else {
    hipLaunchKernelGGL(axpby_scalar_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, st, y, x, n,
                       alpha, beta);
  }
  CLO_CHECK_LAUNCH("axpby");
  return CLO_OK;
}

extern "C" int clo_transpose_f32(float *out, const float *in, long rows, long cols, void *stream) {
  CLO_REQUIRE(rows >= 0 && cols >= 0, "clo_transpose_f32: negative size");
  if (rows == 0 || cols == 0) return CLO_OK;
  CLO_REQUIRE(out && in && out != in, "clo_transpose_f32: null or aliased pointers");
  const long ntiles = cdiv(rows, 64) * cdiv(cols, 64);
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)std::min<long>(ntiles, kNumCU * 16L)),
                     dim3(256), 0, (hipStream_t)stream, out, in, rows, cols);
  CLO_CHECK_LAUNCH("transpose_kernel");
  return CLO_OK;
}

static int canonical_tr(bool pack, float *kmaj, float *w, float *bias, long rows, long bw, long K, void *stream,
                        const char *what) {
  if (rows < 0 || bw < 0 || K < 0) { set_error("%s: negative size", what); return CLO_EINVAL; }
  const long nv = rows * (bias ? bw + 1 : bw);
  if (nv == 0 || K == 0) return CLO_OK;
  if (!kmaj || (bw > 0 && !w)) { set_error("%s: null pointer", what); return CLO_EINVAL; }
  const long ntiles = cdiv(nv, (long)CP_TR) * cdiv(K, (long)CP_KC);
  const dim3 grid((unsigned)std::min<long>(ntiles, kNumCU * 8L));
  if (pack) hipLaunchKernelGGL(canonical_tr_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, kmaj, w, bias, rows, bw, K);
  else hipLaunchKernelGGL(canonical_tr_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, kmaj, w, bias, rows, bw, K);
  CLO_CHECK_LAUNCH("canonical_tr_kernel");
  return CLO_OK;
}
extern "C" int clo_canonical_pack_f32(float *out_kmajor, const float *w, const float *bias, long rows, long cols_w,
                                      long K, void *stream) {
  return canonical_tr(true, out_kmajor, const_cast<float *>(w), const_cast<float *>(bias), rows, cols_w, K, stream,
                      "clo_canonical_pack_f32");
}
extern "C" int clo_canonical_unpack_f32(float *w, float *bias, const float *in_kmajor, long rows, long cols_w, long K,
                                        void *stream) {
  return canonical_tr(false, const_cast<float *>(in_kmajor), w, bias, rows, cols_w, K, stream,
                      "clo_canonical_unpack_f32");
}

extern "C" int clo_rowscale_f32(float *y, const float *x, const float *s, long rows, long K,
                                int reciprocal, float shift, void *stream) {
  CLO_REQUIRE(rows >= 0 && K >= 0, "clo_rowscale_f32: negative size");
  if (rows == 0 || K == 0) return CLO_OK;
  CLO_REQUIRE(y && x && s, "clo_rowscale_f32: null pointer");
  hipLaunchKernelGGL(rowscale_kernel, dim3(stream_grid(rows * K, 256)), dim3(256), 0,
                     (hipStream_t)stream, y, x, s, rows, K, reciprocal, shift);
  CLO_CHECK_LAUNCH("rowscale_kernel");
  return CLO_OK;
}

extern "C" int clo_pack_probes_f32(float *out, long D, long K, uint64_t seed, int dist,
                                   void *stream) {
  CLO_REQUIRE(D >= 0 && K >= 0, "clo_pack_probes_f32: negative size");
  CLO_REQUIRE(dist == 0 || dist == 1, "clo_pack_probes_f32: dist must be 0 or 1");
  if (D == 0 || K == 0) return CLO_OK;
  CLO_REQUIRE(out, "clo_pack_probes_f32: null pointer");
  const long n = D * K;
  hipLaunchKernelGGL(pack_probes_kernel, dim3(stream_grid(cdiv(n, 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, out, n, seed, dist);
  CLO_CHECK_LAUNCH("pack_probes_kernel");
  return CLO_OK;
}

extern "C" long clo_dot_ws_bytes(void) { return (long)DOT_BLOCKS * sizeof(double); }

// out[0] = scale * <x, y> (n elements, any n >= 0); ws: clo_dot_ws_bytes() bytes.
extern "C" int clo_dot_f32(const float *x, const float *y, long n, float scale, float *out, void *ws,
                           void *stream) {
  CLO_REQUIRE(n >= 0, "clo_dot_f32: negative n");
  CLO_REQUIRE(out && ws && (n == 0 || (x && y)), "clo_dot_f32: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)std::max<long>(1, std::min<long>(DOT_BLOCKS, cdiv(n, 1024)));
  hipLaunchKernelGGL(dot_partial_kernel, dim3(blocks), dim3(256), 0, st, x, y, n, (double *)ws);
  CLO_CHECK_LAUNCH("dot_partial_kernel");
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, st, (const double *)ws, blocks, out, scale);
  CLO_CHECK_LAUNCH("dot_final_kernel");
  return CLO_OK;
}

// One conjugate-gradient update for a single right-hand side, step size from DEVICE scalars (no host
// round trip):  a = rz / pap;  x += a p;  r -= a ap;  rr_out[0] = <r, r>.   ws: clo_dot_ws_bytes().
extern "C" int clo_cg_update_f32(float *x, float *r, const float *p, const float *ap, long n,
                                 const float *rz, const float *pap, float *rr_out, void *ws,
                                 void *stream) {
  CLO_REQUIRE(n >= 0, "clo_cg_update_f32: negative n");
  CLO_REQUIRE(rz && pap && rr_out && ws && (n == 0 || (x && r && p && ap)), "clo_cg_update_f32: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)std::max<long>(1, std::min<long>(DOT_BLOCKS, cdiv(n, 1024)));
  hipLaunchKernelGGL(cg_update_kernel, dim3(blocks), dim3(256), 0, st, x, r, p, ap, n, rz, pap, (double *)ws);
  CLO_CHECK_LAUNCH("cg_update_kernel");
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, st, (const double *)ws, blocks, rr_out, 1.f);
  CLO_CHECK_LAUNCH("dot_final_kernel");
  return CLO_OK;
}

// New search direction p = z + (num / den) p with num, den device scalars (beta = rz_new / rz_old).
extern "C" int clo_cg_direction_f32(float *p, const float *z, long n, const float *num, const float *den,
                                    void *stream) {
  CLO_REQUIRE(n >= 0, "clo_cg_direction_f32: negative n");
  if (n == 0) return CLO_OK;
  CLO_REQUIRE(p && z && num && den, "clo_cg_direction_f32: null pointer");
  hipLaunchKernelGGL(cg_direction_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, z, n,
                     num, den);
  CLO_CHECK_LAUNCH("cg_direction_kernel");
  return CLO_OK;
}
