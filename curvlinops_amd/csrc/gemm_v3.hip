// fp32 MFMA GEMM, LDS-DMA engine (gfx950) -- the 128 x 128 x 32 tile loop of the aligned engine of gemm.hip rebuilt
// around `buffer_load_dwordx4 ... lds`:
//   * operand tiles go global -> LDS directly (no staging registers, no ds_write pass) into a ring of NST stages, so
//     NST - 1 k tiles are in flight across the ONE barrier per k tile; completion is counted (`s_waitcnt vmcnt(N)`),
//     never drained, and out-of-range k is zero-filled by the buffer range check (no select after the load);
//   * the LDS images are linear (the DMA writes wave-base + 16 lane); bank conflicts are avoided by a swizzle that is
//     applied on the SOURCE address of the lane that owns a slot and again on the fragment read;
//   * every non-MFMA instruction of the loop (fragment reads of the next 8-k group, the DMA pieces of the tile three
//     ahead) is placed behind one MFMA, so it issues while the matrix pipe is busy with that MFMA.
// Measured against the register-staged loop it replaces (one MI355X, tools/ubench/gemm_v3_probe): 4096^3 115-129 ->
// 141 TFLOP/s, 2048^3 104-111 -> 113-131, and the loop alone runs at 93 % of a loop with the MFMAs only.
//
// Scheduling: either one output tile per workgroup (`blockIdx.y` = batch x split-K slab, as in gemm.hip), or
// STREAM-K: the (tile, k tile) units of the whole problem are cut into `workers` equal contiguous ranges, one per
// workgroup (<= one per CU).  A range that ends inside a tile leaves its partial accumulator in a slot of the
// workspace and raises a flag; the workgroup whose range contains the tile's LAST k tile adds the partials of the
// workgroups before it (fixed order: the result does not depend on timing) and runs the epilogue.  Every workgroup
// walks its range from the last tile to the first, so the partial somebody waits for is written FIRST and the tile that
// needs other workgroups' partials is finished LAST: nobody waits in practice.  Ranges are laid out XCD by XCD (the
// workgroups of one XCD hold neighbouring tiles, which share operand rows in that XCD's L2).
#include "clo_common.h"
#include "gemm.h"

#include <map>
#include <mutex>
#include <utility>

namespace clo {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4v = __attribute__((ext_vector_type(4))) float;
using i32x4v = __attribute__((ext_vector_type(4))) int;
using u32x4v = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int V3_BK = 32;
constexpr int V3_BM = 128, V3_BN = 128, V3_WVM = 2, V3_WVN = 4, V3_NST = 4;
constexpr int V3_NTHR = V3_WVM * V3_WVN * 64;
constexpr int V3_FLAG_STRIDE = 16;       // unsigned per flag (64 bytes apart)
constexpr unsigned V3_SPIN = 1u << 24;   // polls before a waiting workgroup gives up (trap)
constexpr long V3_SK_MAX_TILES = 1024;   // stream-K below this many output tiles
constexpr int V3_SK_MAX_PARTS = 8;       // partial accumulators one finisher adds, at most (about)

// Timing builds (tools/r6/probe_gemm_timeline.py): wall_clock64() stamps (100 MHz) of thread 0 of every workgroup.
#ifdef CLO_V3_TIMING
#define V3_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && s.stamps) s.stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define V3_STAMP(i) do { } while (0)
#endif

struct V3Sched {
  int streamk;        // 0: one tile per workgroup; 1: stream-K
  int nkt;            // stream-K: k tiles per output tile
  int workers;        // stream-K: grid.x
  int team;           // stream-K: members per team (1: every workgroup has a range of its own)
  long units;         // stream-K: tiles x nkt
  int tiles_per_mat;  // output tiles of one matrix of the batch
  float *slots;       // stream-K: workers x (BM BN) partial accumulators
  unsigned *flags;    // stream-K: one per worker; a worker publishes its partial by storing `epoch`
  unsigned epoch;     // stream-K: unique per launch on this (device, stream)
  unsigned *fault;    // stream-K: host-pinned fault word of the device (or nullptr)
  unsigned spin_limit;
#ifdef CLO_V3_TIMING
  unsigned long long *stamps;   // [1024][8]
#endif
};

__device__ __forceinline__ i32x4v v3_srd(const float *p) {
  const unsigned long u = (unsigned long)p;
  i32x4v r;
  r.x = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
  r.y = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
  r.z = 0x7ffffff0;   // the loader clamps its own addresses; an offset >= 2^31 is out of range: the lane gets zeros
  r.w = 0x00020000;
  return r;
}
// one 1 KB piece: the 16 bytes at srd + voff of lane l land at LDS byte lds_dst + 16 l.  hipcc neither counts this
// load nor knows that it writes LDS: the loop below waits for it by count.
__device__ __forceinline__ void v3_dma16(i32x4v srd, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(srd), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void v3_wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// Operand tile of T outer indices x 32 k as a LINEAR LDS image written in 1 KB pieces (T / 8 of them):
//   KC (k contiguous in memory): [T][32]; the 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7), so the 16
//      rows one ds_read_b128 pass touches cover all 64 banks;
//   OC (outer contiguous):       [32][T]; element (k, o) sits at column o ^ (((k >> 2) & 1) << 5): the two half-waves
//      of a fragment read (k and k + 4) hit different bank halves.
template <bool KC, int T, int NW>
struct V3Op {
  static constexpr int PIECES = T / 8, PPW = PIECES / NW;
  static_assert(PIECES % NW == 0, "pieces per wave");
  unsigned voff[PPW];  // byte offset of this lane's 16 bytes relative to element (o0, k0) of the current k tile
  int kq[PPW];         // k (inside the tile) the lane's chunk starts at
  // stride = so (KC) or sk (OC), floats
  __device__ __forceinline__ void init(long stride, int o0, int O, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int piece = wave + NW * q;
      if (KC) {
        const int row = 8 * piece + (lane >> 3), pc = lane & 7, lc = pc ^ ((row >> 1) & 7);
        const int rr = min(o0 + row, O - 1) - o0;   // rows past the end repeat the last one (never stored)
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
  __device__ __forceinline__ void issue_one(int q, i32x4v srd, int lim, unsigned lds_byte, int wave) const {
    const unsigned v = kq[q] < lim ? voff[q] : 0x80000000u;
    v3_dma16(srd, v, lds_byte + (unsigned)(wave + NW * q) * 1024u);
  }
  // fragment values of one lane for the 8-k group g: v[m] feeds MFMA m (k = 8 g + 4 lh + m on both operands)
  static __device__ __forceinline__ f32x4v frag(const float *S, int outer, int g, int lh) {
    if (KC) return *reinterpret_cast<const f32x4v *>(S + outer * 32 + 4 * ((2 * g + lh) ^ ((outer >> 1) & 7)));
    const float *p = S + (8 * g + 4 * lh) * T + (outer ^ (lh << 5));
    f32x4v r;
    r[0] = p[0]; r[1] = p[T]; r[2] = p[2 * T]; r[3] = p[3 * T];
    return r;
  }
};

static_assert(alignof(V3Sched) == 8 && alignof(GemmArgs) == 8, "kernel-argument layout: V3Sched follows GemmArgs at the next multiple of 8");

struct V3Tile {
  int bm, bn, m0, n0, batch, split, kb, ke, nk;
};

// first unit of range w of W over U units.  W <= 256: below 2^23 units the product fits 32 bits (a 64-bit division is a
// few hundred instructions on this machine, and the stream-K prologue takes several)
__device__ __forceinline__ long v3_bound(int w, long U, int W) {
  if (U < (1L << 23)) return (long)((unsigned)w * (unsigned)U / (unsigned)W);
  return (long)w * U / W;
}

template <int BMt, int BNt>
__device__ __forceinline__ void v3_tile(const GemmArgs &p, long lin, int y, bool sk, int tiles_per_mat, V3Tile &t) {
  int tl = (int)lin;
  if (sk) {
    t.batch = lin < (1L << 31) ? (int)((unsigned)lin / (unsigned)tiles_per_mat) : (int)(lin / tiles_per_mat);
    tl = (int)(lin - (long)t.batch * tiles_per_mat);
    t.split = 0;
  } else {
    t.batch = y / p.splitk;
    t.split = y % p.splitk;
  }
  if (p.sym) {
    // upper-triangular tiles only, column by column: tl = bn (bn + 1) / 2 + bm
    int bn = (int)((sqrtf(8.f * (float)tl + 1.f) - 1.f) * 0.5f);
    while ((bn + 1) * (bn + 2) / 2 <= tl) ++bn;
    while (bn * (bn + 1) / 2 > tl) --bn;
    t.bn = bn;
    t.bm = tl - bn * (bn + 1) / 2;
  } else {
    constexpr int GROUP = 8;
    const int per_group = GROUP * p.tiles_n;
    const int g = tl / per_group, first_m = g * GROUP, gsz = min(p.tiles_m - first_m, GROUP);
    const int in_g = tl % per_group;
    t.bm = first_m + in_g % gsz;
    t.bn = in_g / gsz;
  }
  t.m0 = t.bm * BMt;
  t.n0 = t.bn * BNt;
  if (sk) {
    t.kb = 0;
    t.ke = p.K;
  } else {
    t.kb = t.split * p.k_per_split;
    t.ke = min(p.K, t.kb + p.k_per_split);
    if (p.tri) {  // triangular operands: only the k range of this tile that can be nonzero
      int lo = 0, hi = p.K;
      if (p.tri & TRI_KGE_M) lo = max(lo, t.m0);
      if (p.tri & TRI_KGE_N) lo = max(lo, t.n0);
      if (p.tri & TRI_KLT_M) hi = min(hi, t.m0 + BMt);
      if (p.tri & TRI_KLT_N) hi = min(hi, t.n0 + BNt);
      t.kb = max(t.kb, lo & ~(V3_BK - 1));
      t.ke = min(t.ke, hi);
    }
  }
  // (an empty k range -- triangular hint -- still owes beta C / a slab of zeros: one all-masked k tile)
  t.nk = t.ke > t.kb ? (t.ke - t.kb + V3_BK - 1) / V3_BK : 1;
}

// KG = 2 (round 6, small tiles without stream-K): EIGHT waves per output tile -- two groups of WVM x WVN waves, each with an
// LDS ring of its own, run the k loop on one half of the tile's k range each and meet at the same barrier once per k tile;
// group 1 hands its accumulators to group 0 through LDS at the end.  A SIMD then holds two waves of DIFFERENT rings: while one
// is parked at its s_waitcnt / barrier the other issues MFMAs (the four-wave 64 x 64 loop sat parked 39 % of its wave cycles,
// SQ_WAIT_ANY, profiles/r05_gemm_midsize_sweep.txt), and a tile's k loop is half as long.
template <bool AKC, bool BKC, int BMt, int BNt, int WVM, int WVN, int NST, bool SK, int KG = 1>
__global__ __launch_bounds__(WVM *WVN * 64 * KG) void gemm_v3_kernel(const GemmArgs p, const V3Sched s) {
  static_assert(KG == 1 || !SK, "the k-group form has no stream-K schedule");
  constexpr int NW = WVM * WVN, NTHR = NW * 64;
  constexpr int WM = BMt / WVM, WNC = BNt / WVN, MT = WM / 32, NT = WNC / 32;
  constexpr int A_FL = BMt * V3_BK, ST_FL = (BMt + BNt) * V3_BK;
  using DA = V3Op<AKC, BMt, NW>;
  using DB = V3Op<BKC, BNt, NW>;
  constexpr int PER = DA::PPW + DB::PPW;  // DMA instructions per wave per k tile
  constexpr int SLOTS = 4 * MT * NT;            // one slot behind each MFMA of an 8-k group
  static_assert(MT + NT + 1 <= SLOTS, "fragment reads and the bookkeeping statement need a slot each");
  constexpr int PER_SLOTTED = PER < SLOTS - (MT + NT) ? PER : SLOTS - (MT + NT);   // DMA pieces that get a slot; the rest
                                                                                  // (small tiles: 2 of 4) follow the group
  static_assert(NST >= 3, "ring depth");
  extern __shared__ __attribute__((aligned(1024))) float lds_all[];
  V3_STAMP(0);

  const int tid = threadIdx.x & (NTHR - 1);   // (thread index inside the k group)
  const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTHR) : 0;
  float *lds3 = lds_all + (long)kg * NST * ST_FL;   // the group's own ring
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WVN, wn = wave % WVN;
  const int li = lane & 31, lh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds3;

  // ---- this workgroup's units.  Stream-K walks the tiles of its range from the LAST to the first: the partial
  // accumulator other workgroups wait for (the head of the last tile) is written first, and the tile this workgroup
  // has to finish with the partials of the workgroups before it (the tail of the first tile) comes last.
  long c_lin;        // linear index of the consumer's output tile
  int c_kt, c_kend;  // the consumer's segment: next k tile and end (inside the tile's k tile range)
  int n_units;       // units of this workgroup
  long u0 = 0;       // (stream-K) first unit of the range
  int w = blockIdx.x;  // (stream-K) position of the range; neighbours in range order share operand tiles, so they go to
                       // the same XCD (workgroup b runs on XCD b % 8)
  // (stream-K) TEAMS: with G = s.team > 1 the tiles are taken as panels of G row tiles that share their B columns; the
  // G members of a team own the same (panel, k tile) range, member g on row tile g, and run side by side on one XCD --
  // the B tiles they stream are the same at the same time (one trip to memory instead of G).  Each member sequence is a
  // stream-K problem of its own over the panels (its own slots, flags and finishers).
  int g = 0, G = 1, W = 1;
  long U = 0;
  if (SK) {
    G = s.team;
    W = s.workers / G;
    U = s.units / G;
    const int xcd = blockIdx.x % kNumXCD, idx = blockIdx.x / kNumXCD;
    if (G > 1) {   // s.workers is a multiple of 8 G
      const int per_xcd = s.workers / kNumXCD / G;   // teams per XCD
      w = xcd * per_xcd + idx / G;
      g = idx % G;
    } else {
      const int q = s.workers / kNumXCD, rem = s.workers % kNumXCD;
      w = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    u0 = v3_bound(w, U, W);
    const long u1 = v3_bound(w + 1, U, W);
    c_lin = U < (1L << 23) ? (long)((unsigned)(u1 - 1) / (unsigned)s.nkt) : (u1 - 1) / s.nkt;
    c_kt = (int)(max(u0, c_lin * s.nkt) - c_lin * s.nkt);
    c_kend = (int)(u1 - c_lin * s.nkt);
    n_units = (int)(u1 - u0);
  } else {
    const int ntiles = s.tiles_per_mat;
    if (p.sym || p.tri) {
      c_lin = blockIdx.x;  // tiles of very different length: neighbours go to different XCDs
    } else {
      const int b = blockIdx.x, q = ntiles / kNumXCD, rem = ntiles % kNumXCD;
      const int xcd = b % kNumXCD, idx = b / kNumXCD;
      c_lin = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    c_kt = 0;
    c_kend = 0;
    n_units = 0;  // set from the tile below
  }
  V3Tile ct;
  v3_tile<BMt, BNt>(p, SK ? c_lin * G + g : c_lin, blockIdx.y, SK, s.tiles_per_mat, ct);
  if (KG > 1) {
    // group 0: the first ceil(nk / 2) k tiles, group 1: the rest (an odd count leaves it one all-masked tile: `lim` <= 0
    // zero-fills); BOTH run ceil(nk / 2) units, so they execute the same number of barriers
    const int half = (ct.nk + 1) / 2;
    const int mid = ct.kb + half * V3_BK;
    if (kg == 0) ct.ke = min(ct.ke, mid);
    else { ct.kb = mid; }   // (kb > ke: an empty range)
    ct.nk = half;
  }
  if (!SK) n_units = c_kend = ct.nk;
  int seg_kt0 = c_kt;  // first k tile of the consumer's current segment

  // ---- producer: runs NST - 1 units ahead of the consumer.  Its position is kept as the ADDRESSES of the next k tile of
  // the two operands and the number of valid k left (`lim`), advanced by a constant per unit: a handful of scalar
  // instructions in the loop.
  const int K1 = p.A2 ? p.K1 : 0x7fffffff;
  DA da;
  DB db;
  long p_lin = c_lin;
  int p_kt = c_kt, p_kend = c_kend, p_kb = ct.kb, p_ke = ct.ke, issued = 0;
  const long stepA = (AKC ? 1 : p.sa_k) * (long)(V3_BK * 4), stepB = (BKC ? 1 : p.sb_k) * (long)(V3_BK * 4);  // bytes
  unsigned long Ab = 0, Bb = 0;   // element (m0 / n0, k = 0) of the producer's tile
  unsigned long pa = 0, pb = 0;   // the k tile the producer issues next
  int lim = 0;
  auto rfl64 = [](unsigned long v) {
    return ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
  };
  auto prod_seek = [&]() {   // position = k tile p_kt of the tile's range (second K segment: its own operands)
    const int k0 = p_kb + p_kt * V3_BK;
    const bool s2 = k0 >= K1;
    const int kr = s2 ? k0 - K1 : k0;
    lim = (s2 ? p_ke - K1 : min(p_ke, K1)) - kr;
    pa = rfl64(Ab + (s2 ? (unsigned long)(p.A2 - p.A) * 4 : 0ul) + (unsigned long)(kr / V3_BK) * stepA);
    pb = rfl64(Bb + (s2 ? (unsigned long)(p.B2 - p.B) * 4 : 0ul) + (unsigned long)(kr / V3_BK) * stepB);
  };
  auto prod_setup = [&](const V3Tile &t) {
    da.init(AKC ? p.sa_m : p.sa_k, t.m0, p.M, wave, lane);
    db.init(BKC ? p.sb_n : p.sb_k, t.n0, p.N, wave, lane);
    Ab = (unsigned long)(p.A + gemm_off_a(p, t.batch) + (AKC ? (long)t.m0 * p.sa_m : (long)t.m0));
    Bb = (unsigned long)(p.B + gemm_off_b(p, t.batch) + (BKC ? (long)t.n0 * p.sb_n : (long)t.n0));
    p_kb = t.kb; p_ke = t.ke;
    prod_seek();
  };
  prod_setup(ct);
  // SRDs of the unit the producer issues next and its number of valid k (<= 0: zeros; past the range: zeros)
  i32x4v sa, sbd;
  int lim_u;
  auto prod_unit = [&]() {
    sa = v3_srd(reinterpret_cast<const float *>(pa));
    sbd = v3_srd(reinterpret_cast<const float *>(pb));
    lim_u = __builtin_amdgcn_readfirstlane(issued < n_units ? lim : 0);
  };
  auto prod_advance = [&]() {
    ++issued;
    ++p_kt;
    if (__builtin_expect(p_kt == p_kend && issued < n_units, 0)) {  // (stream-K only) on to the tile before this one
      --p_lin;
      p_kt = (int)max(u0 - p_lin * s.nkt, 0L);
      p_kend = s.nkt;
      V3Tile t;
      v3_tile<BMt, BNt>(p, p_lin * G + g, 0, true, s.tiles_per_mat, t);
      prod_setup(t);
    } else if (__builtin_expect(p.A2 && p_kb + p_kt * V3_BK == K1, 0)) {
      prod_seek();
    } else {
      pa += stepA;
      pb += stepB;
      lim -= V3_BK;
    }
  };
  auto issue_all = [&](int stage) {
    const unsigned sb = lds0 + (unsigned)stage * (ST_FL * 4);
#pragma unroll
    for (int q = 0; q < DA::PPW; ++q) da.issue_one(q, sa, lim_u, sb, wave);
#pragma unroll
    for (int q = 0; q < DB::PPW; ++q) db.issue_one(q, sbd, lim_u, sb + A_FL * 4, wave);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int t = 0; t < NST - 1; ++t) {
    prod_unit();
    issue_all(t);
    prod_advance();
  }
  V3_STAMP(7);
  v3_wait_vm_barrier<PER *(NST - 2)>();
  V3_STAMP(1);

  f32x4v fa[2][MT], fb[2][NT];
#define V3_SB __builtin_amdgcn_sched_barrier(0);
  // One 8-k group: 4 MT NT MFMAs on fragment buffer BUF; behind MFMA m goes ONE other instruction -- a fragment read
  // of group GN of stage SN (into the other buffer), with DMA set a piece of the unit NST - 1 ahead, or the scalar
  // bookkeeping statement EXTRA (in the first free slot).
#define V3_GROUP(BUF, SN, GN, DMA, EXTRA)                                                                      \
  {                                                                                                          \
    const float *as_ = lds3 + (SN) * ST_FL, *bs_ = as_ + A_FL;                                               \
    _Pragma("unroll") for (int m = 0; m < 4 * MT * NT; ++m) {                                                \
      const int e = m / (MT * NT), i = (m % (MT * NT)) / NT, j = m % NT;                                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[BUF][i][e], fb[BUF][j][e], acc[i][j], 0, 0, 0);   \
      V3_SB                                                                                                  \
      if (m < MT) fa[(BUF) ^ 1][m] = DA::frag(as_, wm * WM + m * 32 + li, GN, lh);                           \
      else if (m < MT + NT) fb[(BUF) ^ 1][m - MT] = DB::frag(bs_, wn * WNC + (m - MT) * 32 + li, GN, lh);    \
      else if (DMA && m - (MT + NT) < PER_SLOTTED) {                                                         \
        const int q = m - (MT + NT);                                                                         \
        if (q < DA::PPW) da.issue_one(q, sa, lim_u, sbyte, wave);                                            \
        else db.issue_one(q - DA::PPW, sbd, lim_u, sbyte + A_FL * 4, wave);                                  \
      } else if (!DMA && m == MT + NT) { EXTRA; }                                                            \
      V3_SB                                                                                                  \
    }                                                                                                        \
    if (DMA) {                                                                                               \
      _Pragma("unroll") for (int q = PER_SLOTTED; q < PER; ++q) {                                            \
        if (q < DA::PPW) da.issue_one(q, sa, lim_u, sbyte, wave);                                            \
        else db.issue_one(q - DA::PPW, sbd, lim_u, sbyte + A_FL * 4, wave);                                  \
      }                                                                                                      \
      V3_SB                                                                                                  \
    }                                                                                                        \
  }

  {
    const float *as = lds3, *bs = as + A_FL;
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[0][i] = DA::frag(as, wm * WM + i * 32 + li, 0, lh);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[0][j] = DB::frag(bs, wn * WNC + j * 32 + li, 0, lh);
  }
  int st = 0;  // stage of the consumer's unit
  for (int u = 0; u < n_units; ++u) {
    const int st_next = st + 1 == NST ? 0 : st + 1;
    const int st_free = st == 0 ? NST - 1 : st - 1;  // stage of unit u - 1 = where unit u + NST - 1 goes
    const unsigned sbyte = lds0 + (unsigned)st_free * (ST_FL * 4);
    // (the producer's position moves past the unit issued in the previous iteration, then the SRDs of the next one
    // are formed: scalar work, placed between MFMAs)
    V3_GROUP(0, st, 1, false, if (u > 0) prod_advance())
    V3_GROUP(1, st, 2, false, prod_unit())
    V3_GROUP(0, st, 3, false, )
    // unit u + 1: this wave's pieces have landed, the barrier makes everybody's visible (and tells that every wave is
    // done reading unit u - 1, whose stage the DMA below overwrites)
    v3_wait_vm_barrier<PER *(NST - 3)>();
    V3_GROUP(1, st_next, 0, true, )
    st = st_next;

    // ---- end of a segment (last k tile of the output tile, or of this workgroup's range)?
    if (__builtin_expect(c_kt + 1 == c_kend, 0)) {   // cold: laid out behind the loop
      const bool tile_done = c_kend == ct.nk;   // the segment contains the last k tile of the output tile
      const bool whole_from_start = seg_kt0 == 0;
      // What follows runs once per output tile and needs a dozen parameters (epilogue, slots, flags): they are re-read
      // from the kernel-argument segment here (through a pointer the compiler cannot see through) instead of living in
      // scalar registers across the k loop (60 scalar spills to VGPR lanes in the stream-K variant before, 6-8 now).
      const __attribute__((address_space(4))) char *ka =
          (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      const auto *pf = reinterpret_cast<const __attribute__((address_space(4))) GemmArgs *>(ka);
      const auto *sf = reinterpret_cast<const __attribute__((address_space(4))) V3Sched *>(ka + ((sizeof(GemmArgs) + 7) & ~7ul));
      V3_STAMP(tile_done ? 4 : 2);
      if (SK && !tile_done) {
        // partial accumulator -> slot of this workgroup, then the flag (the data is complete in memory first)
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            sf->slots + (long)(w * G + g) * (BMt * BNt), 0, BMt * BNt * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4v v;
              v[0] = acc[i][j][4 * q]; v[1] = acc[i][j][4 * q + 1]; v[2] = acc[i][j][4 * q + 2]; v[3] = acc[i][j][4 * q + 3];
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), rs,
                                                     (unsigned)(((((i * NT + j) * 4 + q) * NTHR) + tid) * 16), 0, 16);
            }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid == 0)
          __hip_atomic_store(sf->flags + (long)(w * G + g) * V3_FLAG_STRIDE, sf->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        V3_STAMP(3);
      } else {
        bool sk_bad = false;
        if (SK && !whole_from_start) {
          // the workgroups before this one hold the first part of the tile's k range.  All their flags are awaited first,
          // then the partials are added in the fixed order w - 1, w - 2, ... with the loads of the next one in flight behind the
          // adds of the current one (round 6: "wait, load 64 KB, add" per partial was one memory round trip EACH -- 4.9 us for
          // the three or four partials of a 512-row product, tools/r6/probe_gemm_timeline.py).
          const long tstart = c_lin * sf->nkt;
          int np = 0;
          for (int w2 = w - 1; w2 >= 0 && v3_bound(w2 + 1, U, W) > tstart; --w2) ++np;
          if (tid == 0) {
            for (int j = 0; j < np && !sk_bad; ++j) {
              unsigned spins = 0;
              while (__hip_atomic_load(sf->flags + (long)((w - 1 - j) * G + g) * V3_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sf->epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > sf->spin_limit) {   // the partial never came: raise the device's fault word and go on --
                  if (sf->fault) __hip_atomic_store(sf->fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                  sk_bad = true;                   // the tile is garbage and will SAY so (NaN in its first entry, below)
                  break;
                }
              }
            }
          }
          asm volatile("s_barrier" ::: "memory");
          constexpr int CH = MT * NT * 4;
          float *const slots = sf->slots;
          auto ld = [&](f32x4v(&b)[CH], int w2) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slots + (long)(w2 * G + g) * (BMt * BNt), 0, BMt * BNt * 4, 0x00020000);
#pragma unroll
            for (int c = 0; c < CH; ++c)
              b[c] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((c * NTHR + tid) * 16), 0, 16));
          };
          auto add = [&](const f32x4v(&b)[CH]) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const int ij = c / 4, q = c % 4;
              acc[ij / NT][ij % NT][4 * q] += b[c][0]; acc[ij / NT][ij % NT][4 * q + 1] += b[c][1];
              acc[ij / NT][ij % NT][4 * q + 2] += b[c][2]; acc[ij / NT][ij % NT][4 * q + 3] += b[c][3];
            }
          };
          if constexpr (CH <= 8) {
            f32x4v b0[CH], b1[CH];
            if (np > 0) ld(b0, w - 1);
            for (int j = 0; j < np; j += 2) {
              if (j + 1 < np) ld(b1, w - 2 - j);
              add(b0);
              if (j + 1 < np) {
                if (j + 2 < np) ld(b0, w - 3 - j);
                add(b1);
              }
            }
          } else {
            f32x4v b0[CH];
            for (int j = 0; j < np; ++j) {
              ld(b0, w - 1 - j);
              add(b0);
            }
          }
        }
        V3_STAMP(5);
        if (sk_bad) acc[0][0][0] = __builtin_nanf("");   // (thread 0 holds entry (m0, n0) of the tile: always inside C)
        if (KG > 1) {
          // group 1's half of the k range joins group 0's through LDS (an area behind both rings: zero tiles issued past the
          // end may still be landing in the rings); group 0 then runs the epilogue alone
          float *red = lds_all + (long)KG * NST * ST_FL;
          if (kg == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((i * NT + j) * 16 + r) * NTHR + tid] = acc[i][j][r];
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (kg == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
          }
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * NT + j) * 16 + r) * NTHR + tid];
        }
        // ---- epilogue of the finished tile
        const bool to_ws = !SK && pf->splitk > 1;
        const int z = blockIdx.y;
        float *C = to_ws ? pf->ws + (long)z * pf->M * pf->N : pf->C + (long)ct.batch * pf->sc_b;
        const long ldc = to_ws ? pf->N : pf->ldc;
        const float alpha = to_ws ? 1.f : pf->alpha;
        const float beta = to_ws ? 0.f : pf->beta;
        const bool mirror = pf->sym && !to_ws && ct.bm != ct.bn;
#ifndef CLO_V3_WIDE_EPI
#define CLO_V3_WIDE_EPI 1
#endif
        // Wide form: when this was the workgroup's last unit the ring is dead, and the tile (then its mirror image) is staged in
        // it and leaves as full-row 16-byte stores.  The per-lane form below wrote 32 four-byte lanes per row segment -- 256
        // store instructions per 128 x 128 tile, 4.7 - 5.6 us between the last MFMA and the end of the kernel -- and the mirror image
        // of a symmetric product in 16-byte pieces of 32 different rows per instruction.
        V3Epi E;
        E.kind = to_ws ? EPI_NONE : pf->epi;
        E.act = pf->e_act; E.div = pf->e_div; E.vec = pf->e_vec; E.mul = pf->e_mul; E.ld_mul = pf->ld_mul; E.out2 = pf->e_out2;
        E.Cbase = pf->C;
        const bool wide = CLO_V3_WIDE_EPI && KG == 1 && u + 1 == n_units && (ldc & 3) == 0 && ((unsigned long)C & 15ul) == 0 &&
                          (E.kind == EPI_NONE || (!mirror && (E.kind != EPI_ACT || ((unsigned long)E.out2 & 15ul) == 0)));
        if (wide) {
          const int Mx = pf->M, Nx = pf->N;
          asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // every fragment read and every DMA into the ring is over
          float *T = lds3;
          {
            constexpr int PD = BNt + 4;
            stage_tile_direct<MT, NT, WM, WNC, PD>(T, acc, wm, wn, li, lh);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            v3_store_rows<BMt, BNt, NTHR>(T, PD, C, ldc, ct.m0, ct.n0, Mx, Nx, alpha, beta, tid, E);
          }
          if (mirror) {
            constexpr int PM = BMt + 4;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the direct image has been read
            stage_tile_mirror<MT, NT, WM, WNC, PM>(T, acc, wm, wn, li, lh);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            v3_store_rows<BNt, BMt, NTHR>(T, PM, C, ldc, ct.n0, ct.m0, Nx, Mx, alpha, beta, tid, E);
          }
        } else
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int col = ct.n0 + wn * WNC + nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = ct.m0 + wm * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              if (row < pf->M && col < pf->N) {
                float *c = C + (long)row * ldc + col;
                if (!to_ws && pf->epi != EPI_NONE) {
                  const GemmArgs pl = *(const GemmArgs *)ka;   // (fused epilogues: the whole block, generic loads; rare path)
                  store_final(pl, c, row, col, acc[mt][nt][r], alpha, beta);
                  continue;
                }
                float v = alpha * acc[mt][nt][r];
                if (beta != 0.f) v += beta * *c;
                *c = v;
                if (mirror) {
                  float *ctp = C + (long)col * ldc + row;
                  float vt = alpha * acc[mt][nt][r];
                  if (beta != 0.f) vt += beta * *ctp;
                  *ctp = vt;
                }
              }
            }
          }
      }
      if (u + 1 < n_units) {  // (stream-K only) on to the tile before this one
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        --c_lin;
        v3_tile<BMt, BNt>(p, c_lin * G + g, 0, true, s.tiles_per_mat, ct);
        c_kt = seg_kt0 = (int)max(u0 - c_lin * s.nkt, 0L);
        c_kend = s.nkt;
      }
    } else {
      ++c_kt;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zero tiles issued past the end
  V3_STAMP(6);
#undef V3_GROUP
#undef V3_SB
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// (sized for the largest tile configuration, the tall 256 x 128 one)
long gemm_streamk_ws_floats() { return (long)kNumCU * 256 * 128; }
long gemm_streamk_ws_floats_square() { return (long)kNumCU * V3_BM * V3_BN; }

// Flags of the stream-K schedule: one array per (device, stream), zeroed when it is created.  Only the workgroup that
// owns a flag ever writes it (the launch's epoch, a per-array counter that never repeats); the others poll for that
// value.  Launches on one stream do not overlap, launches on different streams have different arrays.  A stream that
// is being captured into a graph gets no stream-K launch (a replay would repeat the epoch).
struct V3Flags { unsigned *ptr; unsigned epoch; };
static int v3_flags(hipStream_t stream, unsigned **flags, unsigned *epoch) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, V3Flags> reg;
  int dev = 0;
  int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
  if (rc != CLO_OK) return rc;
  std::lock_guard<std::mutex> lock(mu);
  auto it = reg.find({dev, stream});
  if (it == reg.end()) {
    unsigned *f = nullptr;
    const size_t bytes = (size_t)kNumCU * V3_FLAG_STRIDE * sizeof(unsigned);
    if ((rc = check_hip(hipMalloc(&f, bytes), "hipMalloc(stream-K flags)")) != CLO_OK) return rc;
    // zeroed ON THE LAUNCH STREAM: a hipMemset on the null stream is not ordered with a non-blocking stream and could
    // land after the first kernel has raised its flags
    if ((rc = check_hip(hipMemsetAsync(f, 0, bytes, stream), "hipMemsetAsync(stream-K flags)")) != CLO_OK) return rc;
    it = reg.insert({{dev, stream}, V3Flags{f, 0u}}).first;
  }
  if (++it->second.epoch == 0u) ++it->second.epoch;
  *flags = it->second.ptr;
  *epoch = it->second.epoch;
  return CLO_OK;
}

#ifdef CLO_V3_TIMING
static unsigned long long *g_v3_stamps_host = nullptr;
#endif
// the schedule decision for `tiles` output tiles (all matrices of the batch) of 128 x 128 and k extent K
static long v3_streamk_workers(long tiles, int K) {
  const int nkt = (int)cdiv(K, V3_BK);
  if (nkt <= 0 || tiles <= 0 || tiles >= V3_SK_MAX_TILES || tiles % kNumCU == 0) return 0;
  const long units = tiles * nkt;
  long workers = std::min<long>({(long)kNumCU, units, tiles * V3_SK_MAX_PARTS});
  // at least two k tiles per worker: a shorter range is all pipeline fill
  workers = std::max<long>(1, std::min<long>(workers, units / 2));
  if (workers <= tiles && tiles <= kNumCU) return 0;   // nothing to split: one tile per workgroup
#ifdef CLO_V3_SK_WORKERS   // (experiments: a fixed number of stream-K workers)
  workers = std::max<long>(1, std::min<long>(CLO_V3_SK_WORKERS, std::min<long>(units, kNumCU)));
#endif
  return workers;
}

bool gemm_v3_eligible(const GemmArgs &a, int batch) {
#ifndef CLO_GEMM_V3
#define CLO_GEMM_V3 1
#endif
  static const int off = !CLO_GEMM_V3;
  if (off || a.patch || a.ones || a.ones_b || a.col_out) return false;
  if (a.A2 && (a.K1 % V3_BK != 0)) return false;
  // 32-bit byte offsets inside one k tile of one output tile
  const long lim = 1L << 29;
  const long ra = (a.sa_k == 1 ? 256L * a.sa_m : (long)V3_BK * a.sa_k + 256);
  const long rb = (a.sb_k == 1 ? (long)V3_BN * a.sb_n : (long)V3_BK * a.sb_k + V3_BN);
  (void)batch;
  return ra < lim && rb < lim;
}

// Tile configurations: 128 x 128 (8 waves of 64 x 32, four stages) and the TALL 256 x 128 (8 waves of 64 x 64, three
// stages of 48 KB): half the re-reads of the B operand and 25 % fewer bytes into LDS per flop, for problems with
// enough rows and enough tiles.
constexpr int V3T_BM = 256, V3T_BN = 128, V3T_WVM = 4, V3T_WVN = 2, V3T_NST = 3;
static bool v3_use_tall(const GemmArgs &a, int batch) {
#ifndef CLO_GEMM_V3_TALL
#define CLO_GEMM_V3_TALL 1
#endif
  static const int mode = CLO_GEMM_V3_TALL;  // 0 never, 2 whenever legal
  if (!mode || a.sym || a.M < V3T_BM || a.K < 2048) return false;   // (short k: the longer fill / drain of the tall tile loses, 4608 x 4608 x 512: 226 -> 263 us)
  if (mode == 2) return true;
  const long tall_tiles = cdiv(a.M, V3T_BM) * cdiv(a.N, V3T_BN) * batch;
  // Measured (MI355X): +2-4 % on large squares (4096^3 NT 130.7 -> 135.4, 8192^3 136.4 -> 139.1 TFLOP/s); with stream-K
  // the 128 KB partial tiles cost more than the operand traffic saves (512 x 4608 x 4608: 210 -> 220 us), so the tall
  // tiles are used only where every CU gets at least two of them.
  return tall_tiles >= 2L * kNumCU;
}

// SMALL tiles, 64 x 64 (4 waves of 32 x 32, four stages of 16 KB: two workgroups per CU), round 5: for problems whose
// 128 x 128 tiles do not fill the chip.  Splitting those tiles along K (stream-K or slabs) moves 64 KB partial accumulators
// through each CU's ~30 GB/s memory pipe -- 1024^3: 15 us of MFMA work + 10 us of partial traffic --, while four times as
// many quarter-size tiles need no (or 16 KB) partials; the register-staged 64 x 64 x 64 loop of gemm.hip that served these
// shapes prefetches ONE k tile ahead, less than the memory latency at 0.4 us per k tile, and ran at half its MFMA rate.
#ifndef CLO_V3S_NST
#define CLO_V3S_NST 4
#endif
constexpr int V3S_BM = 64, V3S_BN = 64, V3S_WVM = 2, V3S_WVN = 2, V3S_NST = CLO_V3S_NST;
static bool v3_use_small(const GemmArgs &a, int batch) {
#ifndef CLO_GEMM_V3_SMALL
#define CLO_GEMM_V3_SMALL 1
#endif
  static const int mode = CLO_GEMM_V3_SMALL;   // 0 never
  if (!mode) return false;
  const long tm = cdiv(a.M, V3_BM), tn = cdiv(a.N, V3_BN);
  const long tiles = (a.sym ? tm * (tm + 1) / 2 : tm * tn) * batch;
  // Measured (tools/probe_gemm_sweep_r5.py, profiles/r05_gemm_midsize_sweep.txt): a 64 x 64 tile needs 16 KB of operands per
  // 0.43 us of MFMA work -- 38 GB/s per CU, above what a CU's memory pipe delivers (20 - 36 GB/s) --, so the small tiles only
  // win where few of them are in flight per unit of K: skinny outputs (min(M, N) <= 384: 128 x 2304 x 2304 29.8 -> 27.9 us,
  // 2688 x 256 x 2688 52.1 -> 46.3) and symmetric products (half the tiles: 512-row pixel Grams 46.7 -> 43.1 / 18.9 -> 17.6 us,
  // the patch products of round 4 175 -> 140 us); 512 x 2304 x 2304 loses (64 -> 81 us) and keeps the 128 x 128 tiles.
#ifndef CLO_V3_SMALL_EPI
#define CLO_V3_SMALL_EPI 0
#endif
  if (a.epi != EPI_NONE && !CLO_V3_SMALL_EPI) return false;   // (fused epilogues run the engine's generic store path per tile: C2 at 65 / 128 rows 180 -> 192 / 194 -> 201 us)
  if (mode == 2) return tiles < 2L * kNumCU;   // (probe builds: wherever the square tiles leave the second round half empty)
  // round 6: also the products the register-staged 64 x 64 x 64 loop of gemm.hip used to keep (M N <= 2^20, e. g. 1024^3: at most
  // one small tile per CU): 32.7 -> 29.5 us
  return tiles < kNumCU && (a.sym || std::min(a.M, a.N) <= 384 || (long)a.M * a.N <= 1024L * 1024L);
}

// Tile configuration and stream-K worker count (0: one tile per workgroup) for a problem; streamk_level as GemmArgs::streamk
// *tall_out: 0 square 128 x 128, 1 tall 256 x 128, 2 small 64 x 64
static long v3_plan(const GemmArgs &a, int batch, int streamk_level, int *tall_out) {
  if (v3_use_small(a, batch)) {
    const long tm = cdiv(a.M, V3S_BM), tn = cdiv(a.N, V3S_BN);
    const long tiles = (a.sym ? tm * (tm + 1) / 2 : tm * tn) * batch;
    *tall_out = 2;
    // stream-K over the small tiles only where they, too, leave most of the chip idle or its second round half empty
    long workers = 0;
    if (streamk_level >= 1 && !a.tri && (tiles < kNumCU / 2 + kNumCU / 4 || (tiles > kNumCU && tiles < 2L * kNumCU - 32))) {
      const int nkt = (int)cdiv(a.K, V3_BK);
      const long units = tiles * nkt;
      workers = std::min<long>({(long)kNumCU, units / 4, tiles * V3_SK_MAX_PARTS});   // (the flag array holds kNumCU flags)
      if (workers <= tiles) workers = 0;
    }
    return workers;
  }
  int tall = v3_use_tall(a, batch) ? 1 : 0;
  long workers = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const int bm = tall ? V3T_BM : V3_BM, bn = tall ? V3T_BN : V3_BN;
    const long tm = cdiv(a.M, bm), tn = cdiv(a.N, bn);
    const long tiles = (a.sym ? tm * (tm + 1) / 2 : tm * tn) * batch;
    workers = (streamk_level >= 1 && !a.tri && !tall) ? v3_streamk_workers(tiles, a.K) : 0;
    // the tall tiles without stream-K need enough tiles for every CU; otherwise fall back to the square ones
    if (tall && workers == 0 && tiles < 2L * kNumCU && pass == 0) { tall = 0; continue; }
    break;
  }
  *tall_out = tall;
  return workers;
}
bool gemm_v3_would_streamk(int M, int N, int K, long batch) {
  GemmArgs a{};
  a.M = M; a.N = N; a.K = K;
  int tall = 0;
  return v3_plan(a, (int)batch, 2, &tall) > 0;
}
bool gemm_v3_small(const GemmArgs &a, int batch) { return v3_use_small(a, batch); }

// Launches the main kernel for `a` (k_per_split / splitk set by launch_gemm for 32-deep k tiles; the tile counts are
// set here for the configuration that runs).  a.streamk != 0 asks for the stream-K schedule with a.ws as its workspace
// (1: >= gemm_streamk_ws_floats(128 x 128 tiles) floats, 2: enough for the tall tiles too); *used_streamk tells the
// caller that no split-K reduction is due.
int launch_gemm_v3(const GemmArgs &a0, int batch, bool a_kc, bool b_kc, hipStream_t stream, bool *used_streamk,
                   int *tile_m, int *tile_n) {
  GemmArgs a = a0;
  V3Sched s{};
#ifdef CLO_V3_TIMING
  s.stamps = g_v3_stamps_host;
#endif
#ifndef CLO_GEMM_STREAMK
#define CLO_GEMM_STREAMK 1
#endif
  static const int sk_off = !CLO_GEMM_STREAMK;
  const bool sk_ok = a.streamk && a.ws && !a.tri && !sk_off;
  int tall = 0;
  long workers = v3_plan(a, batch, sk_ok ? a.streamk : 0, &tall);
  const int nkt = (int)cdiv(a.K, V3_BK);
  {
    const int bm = tall == 1 ? V3T_BM : tall == 2 ? V3S_BM : V3_BM, bn = tall == 1 ? V3T_BN : tall == 2 ? V3S_BN : V3_BN;
    a.tiles_m = (int)cdiv(a.M, bm);
    a.tiles_n = (int)cdiv(a.N, bn);
    a.tbm = bm; a.tbn = bn;
    if (tile_m) *tile_m = bm;   // (the split-K reduction of a symmetric product mirrors by THESE tile extents)
    if (tile_n) *tile_n = bn;
  }
  const long tiles_mat = a.sym ? (long)a.tiles_m * (a.tiles_m + 1) / 2 : (long)a.tiles_m * a.tiles_n;
  const long tiles = tiles_mat * batch;
  s.tiles_per_mat = (int)tiles_mat;
  if (workers > 0) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) workers = 0;
  }
  if (workers > 0) {
    // a finisher of an earlier stream-K launch on this device never saw a partial arrive (clo_common.h, "asynchronous
    // faults"): reported once, here; split-K serves this device from now on
    int fdev = 0;
    if (hipGetDevice(&fdev) == hipSuccess) {
      if (fault_take(fdev, FAULT_STREAMK)) {
        set_error("GEMM: an EARLIER stream-K launch on device %d timed out waiting for a partial tile; its result is invalid. "
                  "Split-K is used on this device from now on -- repeat the call.", fdev);
        return CLO_EASYNC;
      }
      if (fault_disabled(fdev, FAULT_STREAMK) || !fault_words_device(fdev)) workers = 0;   // (no fault word: unmonitored -> off)
    }
  }
  const bool sk = workers > 0;
  if (sk) {
    s.streamk = 1;
    s.nkt = nkt;
    s.workers = (int)workers;
    s.units = tiles * nkt;
    s.slots = a.ws;
    // teams of G = tiles_m members when the row tiles are few (each panel = all row tiles of one block column)
#ifndef CLO_GEMM_SK_TEAM
#define CLO_GEMM_SK_TEAM 1
#endif
    static const int team_off = !CLO_GEMM_SK_TEAM;
    s.team = 1;
    const int G = a.tiles_m;
    if (!team_off && !a.sym && batch == 1 && (G == 2 || G == 4 || G == 8) && s.workers % (kNumXCD * G) == 0 &&
        (long)(s.workers / G) * 2 <= (long)a.tiles_n * nkt)
      s.team = G;
    const int rcf = v3_flags(stream, &s.flags, &s.epoch);
    if (rcf != CLO_OK) return rcf;
    int fdev = 0;
    (void)hipGetDevice(&fdev);
    s.fault = fault_words_device(fdev);
    if (s.fault) s.fault += FAULT_STREAMK;
    s.spin_limit = spin_limit();
  }
  *used_streamk = sk;
  int dev_ = 0;
  {
    const int rcd = check_hip(hipGetDevice(&dev_), "hipGetDevice");
    if (rcd != CLO_OK) return rcd;
  }
  dim3 grid;
  if (sk) {
    a.splitk = 1;
    grid = dim3((unsigned)s.workers, 1);
  } else {
    grid = dim3((unsigned)tiles_mat, (unsigned)(batch * a.splitk));
  }
#define CLO_V3(AK, BK_, SKV, BMV, BNV, WM_, WN_, NSTV) CLO_V3K(AK, BK_, SKV, BMV, BNV, WM_, WN_, NSTV, 1)
#define CLO_V3K(AK, BK_, SKV, BMV, BNV, WM_, WN_, NSTV, KGV)                                                   \
  {                                                                                                            \
    auto kern = gemm_v3_kernel<AK, BK_, BMV, BNV, WM_, WN_, NSTV, SKV, KGV>;                                   \
    const size_t smem = (size_t)KGV * NSTV * (BMV + BNV) * V3_BK * sizeof(float) +                             \
                        (KGV > 1 ? (size_t)BMV * BNV * sizeof(float) : 0);   /* + the k groups' hand-over area */ \
    static bool attr_done[64] = {};  /* per device: the attribute belongs to the device's copy of the kernel */ \
    bool &attr_set = attr_done[dev_ & 63];                                                                     \
    if (!attr_set) {                                                                                           \
      int rc_ = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                            \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),          \
                          "hipFuncSetAttribute");                                                              \
      if (rc_ != CLO_OK) return rc_;                                                                           \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    hipLaunchKernelGGL(kern, grid, dim3(WM_ * WN_ * 64 * KGV), smem, stream, a, s);                            \
  }
#define CLO_V3L(SKV, BMV, BNV, WM_, WN_, NSTV)                            \
  if (a_kc && b_kc) CLO_V3(true, true, SKV, BMV, BNV, WM_, WN_, NSTV)     \
  else if (a_kc) CLO_V3(true, false, SKV, BMV, BNV, WM_, WN_, NSTV)       \
  else if (b_kc) CLO_V3(false, true, SKV, BMV, BNV, WM_, WN_, NSTV)       \
  else CLO_V3(false, false, SKV, BMV, BNV, WM_, WN_, NSTV)
  if (tall == 1) {
    if (sk) { CLO_V3L(true, V3T_BM, V3T_BN, V3T_WVM, V3T_WVN, V3T_NST) }
    else { CLO_V3L(false, V3T_BM, V3T_BN, V3T_WVM, V3T_WVN, V3T_NST) }
  } else if (tall == 2) {
    // Eight waves per small tile (two k groups, KG = 2 above) once the k loop is long enough to split: BUILT AND MEASURED in
    // round 6, NOT the default -- it loses 10 - 25 % on every shape it applies to (profiles/r06_gemm_kgroup_ab.txt:
    // 128 x 2304 x 2304 34.9 vs 28.6 us, 384 x 1152 x 1152 32.9 vs 26.2, 256 x 2304 x 2304 49.6 vs 44.1).  The four-wave
    // tile needs 64 KB of LDS, so TWO of them share a CU already: eight waves per CU from two independent workgroups, whose
    // barriers do not couple, beat eight waves of one workgroup (144 KB: one per CU) that all meet at every k tile.
#ifndef CLO_GEMM_V3_KG
#define CLO_GEMM_V3_KG 1
#endif
    const long kspan = a.splitk > 1 ? a.k_per_split : a.K;
    if (sk) { CLO_V3L(true, V3S_BM, V3S_BN, V3S_WVM, V3S_WVN, V3S_NST) }
    else if (CLO_GEMM_V3_KG == 2 && kspan >= 8 * V3_BK) {
      if (a_kc && b_kc) CLO_V3K(true, true, false, V3S_BM, V3S_BN, V3S_WVM, V3S_WVN, V3S_NST, 2)
      else if (a_kc) CLO_V3K(true, false, false, V3S_BM, V3S_BN, V3S_WVM, V3S_WVN, V3S_NST, 2)
      else if (b_kc) CLO_V3K(false, true, false, V3S_BM, V3S_BN, V3S_WVM, V3S_WVN, V3S_NST, 2)
      else CLO_V3K(false, false, false, V3S_BM, V3S_BN, V3S_WVM, V3S_WVN, V3S_NST, 2)
    }
    else { CLO_V3L(false, V3S_BM, V3S_BN, V3S_WVM, V3S_WVN, V3S_NST) }
  } else {
    if (sk) { CLO_V3L(true, V3_BM, V3_BN, V3_WVM, V3_WVN, V3_NST) }
    else { CLO_V3L(false, V3_BM, V3_BN, V3_WVM, V3_WVN, V3_NST) }
  }
#undef CLO_V3L
#undef CLO_V3K
#undef CLO_V3
  CLO_CHECK_LAUNCH("gemm_v3_kernel");
  return CLO_OK;
}

}  // namespace clo

#ifdef CLO_V3_TIMING
extern "C" void clo_v3_timing_set(unsigned long long *device_buffer) { clo::g_v3_stamps_host = device_buffer; }
#endif
