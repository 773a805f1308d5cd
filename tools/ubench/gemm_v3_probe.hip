// Round 3 probe for the LDS-DMA GEMM loop (gfx950): operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds`
// (no staging registers, no ds_write), a ring of NST stages keeps NST - 1 tiles in flight across the ONE barrier per
// k tile, counted `s_waitcnt vmcnt(N)` instead of a drain, and the fragment reads of the next 8-k group (and of the
// next tile's first group) are issued before the MFMAs of the current one.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_v3_probe.hip -o gemm_v3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const float *A, *B;
  float *C;
  long sa_m, sa_k, sb_n, sb_k, ldc;
  int M, N, K, tiles_m, tiles_n;
};

__device__ __forceinline__ i32x4 make_srd(const float *p) {
  const unsigned long u = (unsigned long)p;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
  r.y = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
  r.z = 0x7ffffff0;
  r.w = 0x00020000;
  return r;
}
// one 1 KB piece: lane l's 16 bytes from srd + voff land at LDS byte lds_dst + 16 l (voff >= 2^31: zeros)
__device__ __forceinline__ void dma16(i32x4 srd, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(srd), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int kBK = 32;
// Operand tile of T outer indices x 32 k as a LINEAR LDS image written by 1 KB pieces (T / 8 of them):
//   KC (k contiguous in memory): [T][32]; the 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7)
//   OC (outer contiguous):       [32][T]; element (k, o) sits at column o ^ (((k >> 2) & 1) << 5)
// (both swizzles are applied on the SOURCE address of the lane that owns the LDS slot, and again on the read)
template <bool KC, int T, int NW>
struct DmaOp {
  static constexpr int PIECES = T / 8, PPW = PIECES / NW;
  static_assert(PIECES % NW == 0, "pieces per wave");
  unsigned voff[PPW];
  int kq[PPW];
  // stride = so (KC) or sk (OC), floats; the SRD base points at (o0, k0) of the current tile
  __device__ __forceinline__ void init(long stride, int o0, int O, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int piece = wave + NW * q;
      if (KC) {
        const int row = 8 * piece + (lane >> 3), pc = lane & 7, lc = pc ^ ((row >> 1) & 7);
        const int rr = min(o0 + row, O - 1) - o0;
        voff[q] = (unsigned)(rr * stride * 4 + lc * 16);
        kq[q] = 4 * lc;
      } else {
        const int idx = piece * 64 + lane, k = idx / (T / 4), pc = idx % (T / 4), lc = pc ^ (((k >> 2) & 1) << 3);
        const int oc = min(o0 + 4 * lc, O - 4) - o0;
        voff[q] = (unsigned)(k * stride * 4 + oc * 4);
        kq[q] = k;
      }
    }
  }
  // lim = number of valid k in this tile (<= 0: the whole tile is zeros)
  __device__ __forceinline__ void issue_one(int q, i32x4 srd, int lim, unsigned lds_byte, int wave) const {
    const unsigned v = kq[q] < lim ? voff[q] : 0x80000000u;
    dma16(srd, v, lds_byte + (unsigned)(wave + NW * q) * 1024u);
  }
  __device__ __forceinline__ void issue(i32x4 srd, int lim, unsigned lds_byte, int wave) const {
#pragma unroll
    for (int q = 0; q < PPW; ++q) issue_one(q, srd, lim, lds_byte, wave);
  }
  // fragment values of one lane for 8-k group g: v[m] feeds MFMA m (k = 8 g + 4 lh + m)
  static __device__ __forceinline__ float4 frag(const float *S, int outer, int g, int lh) {
    if (KC) return *reinterpret_cast<const float4 *>(S + outer * 32 + 4 * ((2 * g + lh) ^ ((outer >> 1) & 7)));
    const float *p = S + (8 * g + 4 * lh) * T + (outer ^ (lh << 5));
    return make_float4(p[0], p[T], p[2 * T], p[3 * T]);
  }
};

template <bool AKC, bool BKC, int BM, int BN, int WVM, int WVN, int NST, int ABL = 0>
__global__ __launch_bounds__(WVM *WVN * 64) void gemm_v3_kernel(const Args p) {
  constexpr int NW = WVM * WVN;
  constexpr int WM = BM / WVM, WN = BN / WVN, MT = WM / 32, NT = WN / 32;
  constexpr int A_FL = BM * kBK, ST_FL = (BM + BN) * kBK;
  using DA = DmaOp<AKC, BM, NW>;
  using DB = DmaOp<BKC, BN, NW>;
  constexpr int PER = DA::PPW + DB::PPW;  // DMA instructions per wave per tile
  extern __shared__ __attribute__((aligned(1024))) float lds3[];

  constexpr int kNumXCD = 8;
  const int ntiles = p.tiles_m * p.tiles_n;
  int lid;
  {
    const int b = blockIdx.x;
    const int q = ntiles / kNumXCD, rem = ntiles % kNumXCD;
    const int xcd = b % kNumXCD, idx = b / kNumXCD;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  constexpr int GROUP = 8;
  const int per_group = GROUP * p.tiles_n;
  const int gi = lid / per_group, first_m = gi * GROUP, gsz = min(p.tiles_m - first_m, GROUP);
  const int in_g = lid % per_group;
  const int bm = first_m + in_g % gsz, bn = in_g / gsz;
  const int m0 = bm * BM, n0 = bn * BN;
  const int kb = 0, ke = p.K;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WVN, wn = wave % WVN;
  const int li = lane & 31, lh = lane >> 5;

  DA da;
  DB db;
  da.init(AKC ? p.sa_m : p.sa_k, m0, p.M, wave, lane);
  db.init(BKC ? p.sb_n : p.sb_k, n0, p.N, wave, lane);
  const float *Ab = AKC ? p.A + (long)m0 * p.sa_m + kb : p.A + (long)kb * p.sa_k + m0;
  const float *Bb = BKC ? p.B + (long)n0 * p.sb_n + kb : p.B + (long)kb * p.sb_k + n0;
  const long stepA = AKC ? kBK : kBK * p.sa_k, stepB = BKC ? kBK : kBK * p.sb_k;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds3;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (ke - kb + kBK - 1) / kBK;
  auto issue_tile = [&](int t, int stage) {
    const int lim = ke - (kb + t * kBK);
    const unsigned sb = lds0 + (unsigned)stage * (ST_FL * 4);
    da.issue(make_srd(Ab + (long)t * stepA), lim, sb, wave);
    db.issue(make_srd(Bb + (long)t * stepB), lim, sb + A_FL * 4, wave);
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t) issue_tile(t, t);
  wait_vm_barrier<PER *(NST - 2)>();

  f32x4 fa[2][MT], fb[2][NT];
  auto ld4 = [](float4 v) { f32x4 r; r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; return r; };
  auto read_frags = [&](int stage, int g, int buf) {
    const float *as = lds3 + stage * ST_FL, *bs = as + A_FL;
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[buf][i] = ld4(DA::frag(as, wm * WM + i * 32 + li, g, lh));
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[buf][j] = ld4(DB::frag(bs, wn * WN + j * 32 + li, g, lh));
  };
#define V3_SB __builtin_amdgcn_sched_barrier(0);
  // One 8-k group: 4 MT NT MFMAs on fragment buffer BUF; behind MFMA number m goes ONE other instruction -- a
  // fragment read of group GN of stage SN (into the other buffer) or, with DMA set, a piece of tile it + NST - 1 --
  // so that each of them issues while the matrix pipe works on the MFMA just issued.
#define V3_GROUP(BUF, SN, GN, DMA)                                                                       \
  {                                                                                                      \
    const float *as_ = lds3 + (SN) * ST_FL, *bs_ = as_ + A_FL;                                           \
    _Pragma("unroll") for (int m = 0; m < 4 * MT * NT; ++m) {                                            \
      const int e = m / (MT * NT), i = (m % (MT * NT)) / NT, j = m % NT;                                 \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][i][e], fb[BUF][j][e], acc[i][j], 0, 0, 0); \
      V3_SB                                                                                              \
      if (!(ABL & 4)) {                                                                                  \
        if (m < MT) fa[(BUF) ^ 1][m] = ld4(DA::frag(as_, wm * WM + m * 32 + li, GN, lh));                \
        else if (m < MT + NT) fb[(BUF) ^ 1][m - MT] = ld4(DB::frag(bs_, wn * WN + (m - MT) * 32 + li, GN, lh)); \
      }                                                                                                  \
      if (DMA && !(ABL & 1) && m >= MT + NT && m - (MT + NT) < PER) {                                    \
        const int q = m - (MT + NT);                                                                     \
        if (q < DA::PPW) da.issue_one(q, sa, lim, sb, wave);                                             \
        else db.issue_one(q - DA::PPW, sbd, lim, sb + A_FL * 4, wave);                                   \
      }                                                                                                  \
      V3_SB                                                                                              \
    }                                                                                                    \
  }
  static_assert(MT + NT + PER <= 4 * MT * NT, "one slot per MFMA");

  read_frags(0, 0, 0);
  if (ABL & 4) read_frags(0, 1, 1);
  int st = 0;  // stage of tile it
  for (int it = 0; it < nk; ++it) {
    const int st_next = st + 1 == NST ? 0 : st + 1;
    const int st_free = st == 0 ? NST - 1 : st - 1;  // stage of tile it - 1 = where tile it + NST - 1 goes
    const int t = it + NST - 1, lim = ke - (kb + t * kBK);
    const unsigned sb = lds0 + (unsigned)st_free * (ST_FL * 4);
    const i32x4 sa = make_srd(Ab + (long)t * stepA), sbd = make_srd(Bb + (long)t * stepB);
    V3_GROUP(0, st, 1, false)
    V3_GROUP(1, st, 2, false)
    V3_GROUP(0, st, 3, false)
    // tile it + 1: this wave's pieces have landed, the barrier makes everybody's visible (and tells that every wave is
    // done reading tile it - 1, whose stage the DMA below overwrites)
    if (!(ABL & 2)) wait_vm_barrier<PER *(NST - 3)>();
    V3_GROUP(1, st_next, 0, true)
    st = st_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + wn * WN + nt * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M && col < p.N) p.C[(long)row * p.ldc + col] = acc[mt][nt][r];
      }
    }
}

__global__ void ref_kernel(const Args p, float *R) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.M * p.N) return;
  const int m = idx / p.N, n = idx % p.N;
  float s = 0.f;
  for (int k = 0; k < p.K; ++k) s = fmaf(p.A[m * p.sa_m + k * p.sa_k], p.B[n * p.sb_n + k * p.sb_k], s);
  R[idx] = s;
}

template <bool AKC, bool BKC, int BM, int BN, int WVM, int WVN, int NST, int ABL = 0>
static float run(const Args &a0, int reps) {
  Args a = a0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  auto kern = gemm_v3_kernel<AKC, BKC, BM, BN, WVM, WVN, NST, ABL>;
  const size_t smem = (size_t)NST * (BM + BN) * kBK * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  dim3 grid(a.tiles_m * a.tiles_n), block(WVM * WVN * 64);
  hipLaunchKernelGGL(kern, grid, block, smem, 0, a);
  CK(hipDeviceSynchronize());
  if (reps <= 0) return 0.f;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, block, smem, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

static double check(const Args &a, float *R, std::vector<float> &hc, std::vector<float> &hr) {
  const long total = (long)a.M * a.N;
  hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, a, R);
  CK(hipDeviceSynchronize());
  hc.resize(total);
  hr.resize(total);
  CK(hipMemcpy(hr.data(), R, total * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int m = 0; m < a.M; ++m) {
    CK(hipMemcpy(hc.data() + (long)m * a.N, a.C + (long)m * a.ldc, a.N * 4L, hipMemcpyDeviceToHost));
  }
  for (long i = 0; i < total; ++i) {
    const double d = std::fabs((double)hc[i] - hr[i]);
    if (!(d <= worst)) worst = d;  // NaN-propagating
  }
  return worst;
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  struct Shape { int M, N, K; };
  const Shape shapes[] = {{128, 128, 32},  {128, 128, 40},   {100, 132, 68},   {256, 384, 1000}, {512, 2304, 2304},
                          {256, 2304, 2304}, {512, 4608, 4608}, {1024, 1024, 1024}, {2048, 2048, 2048}, {4096, 4096, 4096}};
  const size_t cap = 4608UL * 4608UL + 4096;
  float *A, *B, *C, *R;
  CK(hipMalloc(&A, cap * 4));
  CK(hipMalloc(&B, cap * 4));
  CK(hipMalloc(&C, cap * 4));
  CK(hipMalloc(&R, cap * 4));
  std::vector<float> h(cap);
  srand(1);
  for (size_t i = 0; i < cap; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(A, h.data(), cap * 4, hipMemcpyHostToDevice));
  for (size_t i = 0; i < cap; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(B, h.data(), cap * 4, hipMemcpyHostToDevice));
  std::vector<float> hc, hr;
  if (argc > 2) {  // ablations on one shape: what does each part of the loop cost?
    Args a{};
    a.A = A; a.B = B; a.C = C;
    a.M = argc > 4 ? atoi(argv[2]) : 512; a.N = argc > 4 ? atoi(argv[3]) : 2304; a.K = argc > 4 ? atoi(argv[4]) : 2304;
    a.sa_m = a.K; a.sa_k = 1; a.sb_n = 1; a.sb_k = a.N; a.ldc = a.N;
    printf("full            %.1f us\n", run<true, false, 128, 128, 2, 4, 4, 0>(a, reps));
    printf("no dma          %.1f us\n", run<true, false, 128, 128, 2, 4, 4, 1>(a, reps));
    printf("no barrier      %.1f us\n", run<true, false, 128, 128, 2, 4, 4, 2>(a, reps));
    printf("no dma, barrier %.1f us\n", run<true, false, 128, 128, 2, 4, 4, 3>(a, reps));
    printf("no ds_read      %.1f us\n", run<true, false, 128, 128, 2, 4, 4, 4>(a, reps));
    printf("mfma only       %.1f us\n", run<true, false, 128, 128, 2, 4, 4, 7>(a, reps));
    printf("kc/kc full      %.1f us\n", run<true, true, 128, 128, 2, 4, 4, 0>(a, reps));
    printf("kc/kc mfma only %.1f us\n", run<true, true, 128, 128, 2, 4, 4, 7>(a, reps));
    return 0;
  }
  for (const Shape &s : shapes) {
    for (int lay = 0; lay < 4; ++lay) {
      const bool akc = lay & 1, bkc = lay & 2;
      Args a{};
      a.A = A; a.B = B; a.C = C;
      a.M = s.M; a.N = s.N; a.K = s.K;
      a.sa_m = akc ? s.K : 1; a.sa_k = akc ? 1 : s.M;
      a.sb_n = bkc ? s.K : 1; a.sb_k = bkc ? 1 : s.N;
      a.ldc = s.N;
      CK(hipMemset(C, 0xff, (size_t)s.M * s.N * 4));
      const bool big = (long)s.M * s.N * s.K > (1L << 30);
      float us = 0, us3 = 0;
      const int r = (long)s.M * s.N * s.K < (1L << 22) ? 0 : reps;
#define RUN(AK, BK_) \
  us = run<AK, BK_, 128, 128, 2, 4, 4>(a, r); const double err = (big && lay) ? -1.0 : check(a, R, hc, hr); \
  us3 = run<AK, BK_, 128, 128, 2, 4, 3>(a, r);
      if (akc && bkc) { RUN(true, true) printf("M=%d N=%d K=%d A%s B%s: nst4 %.1f us %.1f TF | nst3 %.1f us | err %.3g\n", s.M, s.N, s.K, "kc", "kc", us, 2.0 * s.M * s.N * s.K / us * 1e-6, us3, err); }
      else if (akc) { RUN(true, false) printf("M=%d N=%d K=%d A%s B%s: nst4 %.1f us %.1f TF | nst3 %.1f us | err %.3g\n", s.M, s.N, s.K, "kc", "oc", us, 2.0 * s.M * s.N * s.K / us * 1e-6, us3, err); }
      else if (bkc) { RUN(false, true) printf("M=%d N=%d K=%d A%s B%s: nst4 %.1f us %.1f TF | nst3 %.1f us | err %.3g\n", s.M, s.N, s.K, "oc", "kc", us, 2.0 * s.M * s.N * s.K / us * 1e-6, us3, err); }
      else { RUN(false, false) printf("M=%d N=%d K=%d A%s B%s: nst4 %.1f us %.1f TF | nst3 %.1f us | err %.3g\n", s.M, s.N, s.K, "oc", "oc", us, 2.0 * s.M * s.N * s.K / us * 1e-6, us3, err); }
      fflush(stdout);
    }
  }
  return 0;
}
