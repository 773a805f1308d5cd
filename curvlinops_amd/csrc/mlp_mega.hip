// The <= 8-row GGN / Fisher matvec of a three-layer MLP with a narrow head in ONE persistent launch
// (reference ggn.py:41-72: jvp -> loss Hessian -> vjp; round-3 replacement of the six-launch chain of
// mlp.hip for networks of BASELINE config C2's class).
//
// 256 workgroups (one per CU, 8 waves) form a 16 x 16 grid over layer 2: workgroup (fb, kb) owns the
// tile  W2 / V2 [feature block fb (<= 168 rows)] x [K range kb (<= 176 columns)].  Every weight is read
// from HBM exactly once (12 D bytes per matvec): the W2 tile stays in LDS for the backward pass, the V2
// tile lives in registers, and all of a workgroup's weight loads are issued before the first dependency
// (the loads of layer 2 hide the first seam).  The all-to-all seams of the algorithm are 16-workgroup
// group exchanges of <= 12 KB per workgroup (write-through stores, one counter per group and seam,
// sc1 loads), except the head (10 output classes), which takes a leader per row group and one
// chip-wide counter:
//
//   phase 1  layer 1 for the workgroup's 8 / 12 features of K range kb (in-block split-K over 8 waves)
//            -> a1, da1 published                                  [column-group seam A]
//   phase 2  partial z2, dz2 of tile (fb, kb) on 16x16x4 MFMAs, W2 fragments copied to LDS
//            -> slab[kb] published                                 [row-group seam A]
//   phase 3  finish 8 / 12 features of block fb: bias, activation, phi'; head partials
//            -> phi'2, partials published                          [row-group seam B]
//            leader of the row group sums its 16 partials          [chip-wide counter]
//            everybody: f, J v, loss Hessian -> delta_3
//   phase 4  delta_2 for the block, partial delta_1 = delta_2 W2 from the LDS tile (MFMA)
//            -> slab2[fb] published                                [column-group seam B]
//            out_W2 tile, out_b2, out_W3, out_b3 (write-only)
//   phase 5  delta_1 for the 8 / 12 layer-1 features, out_W1 rows, out_b1
//
// Loads whose data is needed late are issued early; because a wave's loads return in order, everything
// that must arrive quickly (x, the exchanged activations) is loaded by wave 7, which owns no weight tile.
// All sums are taken in fixed orders: results are bit-for-bit repeatable.  Every spin is bounded.
#include "clo_common.h"
#include "mlp_loss.h"
#include "persist_gate.h"

#include <mutex>

namespace clo {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int MG_G = 256, MG_T = 512;
constexpr int MG_MAXS = 11;              // k16 steps per K range
constexpr int MG_MAXG = 3;               // feature groups (of 8) per compute wave
constexpr int MG_CWAVES = 7;             // waves 0..6 own the layer-2 tile, wave 7 does the exchanges
constexpr int MG_NJ = MG_CWAVES * MG_MAXG * 8;   // 168 features per block
constexpr int MG_KR = MG_MAXS * 16;      // 176 columns per K range
constexpr int MG_LDW = 176;              // row pitch of the W2 tile in LDS (= 16 mod 32: conflict-free b32 column reads)
constexpr int MG_LDB = MG_KR + 4;        // row pitch of the staged [a1 ; da1] operand
constexpr int MG_P1S = 8;                // k16 steps per wave in layer 1 (d0 <= 1024)
#ifndef CLO_MG_PRE
#define CLO_MG_PRE 2
#endif
#ifndef CLO_MG_PACE
#define CLO_MG_PACE 3
#endif
#ifndef CLO_MG_XCD
#define CLO_MG_XCD 1
#endif
// Skeleton builds (tools/run_mega_skeleton.sh): bit 0 drops the layer-1 / layer-2 MFMAs, bit 1 the delta_1 sweep's FMAs, bit 2
// the FMAs of out_W2, bit 3 those of out_W1 -- every load, LDS copy, seam and store stays, each loaded value still feeds
// (one add) what is stored, so the timing is that of the data movement and the exchanges alone; the results are garbage.
#ifndef CLO_MG_ABLATE
#define CLO_MG_ABLATE 0
#endif
// Scalar memory path at the seams (round 5; tools/ubench/smem_seam_probe.hip, profiles/r05_c2_scalar_seam.txt).  A CU's
// vector-memory pipe returns in issue order, so a seam's poll and gather wait behind every weight byte requested before
// them; s_load ... glc goes to L2 on the scalar cache's own path.  Tried and measured on the headline shape:
//   * gather of [a1 ; da1] with s_load_dwordx16 by all eight waves, whole tile requested up front: 48.3 us (paced vector
//     gather 44.5): the SGPR file holds four 64-byte chunks per wave, so 176 chunks take six ~0.9 us round trips;
//   * publish with s_store_dwordx4 + s_dcache_wb: correct, 0.8 us slower than the sc1 vector stores;
//   * POLL and ARRIVE of a counter: faster (below).  That is what is kept.
// Scalar POLL of group counters, CLO_MG_SPOLL (the gathers stay on the vector path).  A wait is one wave spinning on
// s_load_dword ... glc: it neither queues behind the weight loads / write-only stores of its CU nor adds to them.
// Counters are monotonic within a call and the data is read with sc1 vector loads issued after the poll returned, so the
// scalar read can only be late, never wrong (every 64th spin also reads the counter the architected way).  Measured per
// seam (bits below): seam A -1.2 us, the fanned-out top flag -0.7 us, the other three nothing; polling ONE top line from
// all 256 workgroups through the scalar path costs +15 us, hence CLO_MG_TOPFAN.
#ifndef CLO_MG_SPOLL
#define CLO_MG_SPOLL 9
#endif
constexpr int MG_PRE = CLO_MG_PRE;       // tile steps issued right behind the layer-1 loads
// round 4: the compute waves keep requesting the layer-2 tile WHILE wave 7 runs the first seam, never
// more than MG_PACE steps (3 KB per wave and step) ahead of what has landed: the vector-memory pipe of the CU stays busy
// through the seam, and the seam's gather queues behind <= MG_PACE x 21 KB instead of behind the whole tile.
constexpr int MG_PACE = CLO_MG_PACE;
// round 5: only the first MG_HOLD steps of the tile are requested before / during seam A; the rest is requested inside
// phase 2, MG_HOLD steps ahead of its use.  Phase 2 used to start with a wait for the WHOLE tile (the conditional requests
// made hipcc count conservatively), and the seam's publish / gather queued behind whatever had been requested.  Measured:
// seam A ends 1.6 us earlier (a1 in LDS at 12.9 instead of 14.5 us), phase 2 ends when it did before (19.9 us) -- it now
// waits for the held-back steps: the 82 MB of W1, V1, W2, V2 arrive at ~4.2 TB/s whatever the request schedule (16 rows x
// 64 bytes per request, 704 contiguous bytes per tile row).  MG_HOLD 3 .. 11: 40.8 .. 41.4 us per product.
#ifndef CLO_MG_HOLD
#define CLO_MG_HOLD 7
#endif
constexpr int MG_HOLD = CLO_MG_HOLD < MG_PRE ? MG_PRE : (CLO_MG_HOLD > MG_MAXS ? MG_MAXS : CLO_MG_HOLD);
constexpr int MG_NB = 8, MG_CMAX = 16;
constexpr int MG_MAXDEV = 64;            // device ordinals with per-device launch state
constexpr unsigned MG_SPIN = 1u << 22;   // bound of every spin (each poll is a fabric round trip + s_sleep)

// ---- LDS carve (floats)
constexpr int MG_OFF_W = 0;                              // [168][176]   W2 tile   (phase 1: x and the wave merge)
constexpr int MG_OFF_B = MG_OFF_W + MG_NJ * MG_LDW;      // [16][180]    a1 ; da1 of the K range
constexpr int MG_OFF_SL = MG_OFF_B + 16 * MG_LDB;        // [2][8][168]  z2 / dz2 partials; later [8][176] delta_1 partials
constexpr int MG_OFF_D2 = MG_OFF_SL + 2 * MG_NB * MG_NJ; // [168][8]     delta_2
constexpr int MG_OFF_M = MG_OFF_D2 + MG_NJ * MG_NB;      // misc, see below
constexpr int MG_M_PUB = 0;      // [2][8][16] a1 / da1 of the layer-1 slice
constexpr int MG_M_PHI1 = 256;   // [8][16]
constexpr int MG_M_FIN = 384;    // [2][8][16] a2 / da2 of the finished slice
constexpr int MG_M_W3 = 640;     // [2][16][16] W3 / V3 columns of the finished slice
constexpr int MG_M_F = 1152;     // [8][16] f
constexpr int MG_M_U = 1280;     // [8][16] J v
constexpr int MG_M_DL = 1408;    // [8][16] delta_3
constexpr int MG_M_D1 = 1536;    // [16][8] delta_1 of the layer-1 slice
constexpr int MG_M_FLAG = 1664;  // [4]
constexpr int MG_M_B1 = 1696;    // [2][16] b1 / Vb1 of the layer-1 slice (requested before any weight)
constexpr int MG_M_B2 = 1728;    // [2][16] b2 / Vb2 of the finished slice
constexpr int MG_M_B3 = 1760;    // [2][16] b3 / Vb3
constexpr int MG_M_TRASH = 1792;  // [64][4]
constexpr int MG_M_AUX = 2048;   // [N][aux_rank][C] backpropagated vectors of the rank-M curvature (<= MG_AUX_MAX floats)
constexpr int MG_AUX_MAX = 1024;
constexpr int MG_LDS_FLOATS = MG_OFF_M + 2048 + MG_AUX_MAX;
static_assert(MG_LDS_FLOATS * 4 <= 160 * 1024, "LDS carve exceeds the CU");
// phase-1 aliases inside the W2 tile area
constexpr int MG_OFF_RED = MG_OFF_W;                     // [8 waves][2][4][64]
constexpr int MG_OFF_XW = MG_OFF_W + 4096;               // [8 waves][8][16 MG_P1S + 4]  x slices

struct MegaArgs {
  const float *W1, *b1, *V1, *Vb1, *W2, *b2, *V2, *Vb2, *W3, *b3, *V3, *Vb3;
  float *O1, *Ob1, *O2, *Ob2, *O3, *Ob3;
  const float *X;
  int N, d0, d1, d2, C, act1, act2;
  int kind;
  const float *aux;
  int aux_rank;
  float scale, beta;
  float *xch;       // exchange area, see mega_xch_floats
  unsigned *sync;   // counters, see mega_sync_words
  unsigned *fault;  // host-pinned fault word of the device (or nullptr)
  unsigned spin_limit;
};

// exchange area (floats): a1, da1 [NB][d1] | slab [16][2][NB][d2] | dphi2 [NB][d2] | hp [256][256] | gsum [16][256] | slab2 [16][NB][d1]
__host__ __device__ inline long mg_off_a1(int, int) { return 0; }
__host__ __device__ inline long mg_off_da1(int d1, int) { return (long)MG_NB * d1; }
__host__ __device__ inline long mg_off_slab(int d1, int) { return 2L * MG_NB * d1; }
__host__ __device__ inline long mg_off_dphi2(int d1, int d2) { return mg_off_slab(d1, d2) + 32L * MG_NB * d2; }
__host__ __device__ inline long mg_off_hp(int d1, int d2) { return mg_off_dphi2(d1, d2) + (long)MG_NB * d2; }
__host__ __device__ inline long mg_off_gsum(int d1, int d2) { return mg_off_hp(d1, d2) + 256L * 256; }
__host__ __device__ inline long mg_off_slab2(int d1, int d2) { return mg_off_gsum(d1, d2) + 16L * 256; }
long mega_xch_floats(int d1, int d2) { return mg_off_slab2(d1, d2) + 16L * MG_NB * d1 + 64; }
// sync words: line 0 = {call, err}; two sets of 65 counters (colA[16], rowA[16], rowB[16], colB[16], top), one per 128-B line
// The chip-wide "top" flag is fanned out (CLO_MG_TOPFAN): each of the 16 row-group leaders adds to 16 lines (one
// instruction, lane = line) and a workgroup polls the line of its own row group -- 16 pollers per line instead of 256.
#ifndef CLO_MG_TOPFAN
#define CLO_MG_TOPFAN 1
#endif
constexpr int MG_SET_LINES = CLO_MG_TOPFAN ? 80 : 65;
long mega_sync_words() { return 32L * (1 + 2 * MG_SET_LINES); }
// -DCLO_MEGA_TIMING builds stamp wall_clock64() at 16 points per workgroup into the 8192 floats behind the counters
long mega_debug_floats() { return 16384; }
#ifdef CLO_MEGA_TIMING
#define MG_STAMP(i) do { if (tid == 0) reinterpret_cast<unsigned long long *>(sy + 32 * (1 + 2 * MG_SET_LINES))[w * 32 + (i)] = wall_clock64(); } while (0)
// stamps 16..31: lane 0 of the exchange wave (wave 7)
#define MG_STAMP7(i) do { if (lane == 0) reinterpret_cast<unsigned long long *>(sy + 32 * (1 + 2 * MG_SET_LINES))[w * 32 + (i)] = wall_clock64(); } while (0)
#else
#define MG_STAMP(i) do { } while (0)
#define MG_STAMP7(i) do { } while (0)
#endif

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x4 ld_x(__amdgpu_buffer_rsrc_t rs, long off_floats) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(off_floats * 4), 0, 16));
}
__device__ __forceinline__ void st_x(__amdgpu_buffer_rsrc_t rs, long off_floats, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)(off_floats * 4), 0, 16);
}
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one lane: arrive on a group counter / wait until it reaches `target` (bounded)
__device__ __forceinline__ void mg_arrive(unsigned *cnt) {
  __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (bounded: on a timeout -- or when another workgroup of this launch has timed out -- the wait simply ends; see
// "asynchronous faults" in clo_common.h)
struct MgAbort { unsigned *err; unsigned *fault; unsigned limit; };
__device__ __forceinline__ void mg_wait(unsigned *cnt, unsigned target, const MgAbort &ab) {
  unsigned spins = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 255u) == 0u && __hip_atomic_load(ab.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    if (spins > ab.limit) {  // ~seconds: the grid is not co-resident (or the counters were not initialised)
      __hip_atomic_store(ab.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ab.fault) __hip_atomic_store(ab.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}

__device__ __forceinline__ float4 mg_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// the weight stream (W1, V1, W2, V2: every byte is read by exactly ONE workgroup, once): CLO_MG_NTW = 1 marks these loads
// nontemporal (the guide's "nt-weights" row: issued -> landed -18 % on an LDS-DMA weight stream).  A/B of round 6:
// profiles/r06_c2_mega_nt_weights.txt.
#ifndef CLO_MG_NTW
#define CLO_MG_NTW 0
#endif
__device__ __forceinline__ float4 mg_ldw(const float *p) {
#if CLO_MG_NTW
  typedef float __attribute__((ext_vector_type(4))) v4;
  const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const float4 *>(p);
#endif
}
__device__ __forceinline__ void mg_st4nt(float *p, const float4 &v) {
  typedef float __attribute__((ext_vector_type(4))) v4;
  __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, reinterpret_cast<v4 *>(p));
}
__device__ __forceinline__ float mg_and(float x, unsigned m) { return __uint_as_float(__float_as_uint(x) & m); }

// scalar poll of a group counter (every lane of the wave runs it; clo_common.h)
__device__ __forceinline__ void mg_wait_scalar(unsigned *cnt, unsigned target, const MgAbort &ab, int lane) {
  scalar_wait(cnt, target, ab.err, ab.fault, ab.limit, lane);
}
// arrival through the scalar path as well (s_atomic_add, no return)
#ifndef CLO_MG_SARRIVE
#define CLO_MG_SARRIVE 1
#endif
__device__ __forceinline__ void mg_arrive_scalar(unsigned *cnt) { scalar_arrive(cnt); }
// CLO_MG_SARRIVE bits as CLO_MG_SPOLL's
#define MG_ARRIVE_B(cnt, bit)                                                   \
  do {                                                                          \
    if ((CLO_MG_SARRIVE) & (bit)) {                                             \
      if (wave == 0) mg_arrive_scalar(cnt);                                     \
    } else {                                                                    \
      if (tid == 0) mg_arrive(cnt);                                             \
    }                                                                           \
  } while (0)
#if CLO_MG_SARRIVE & 1
#define MG_ARRIVE7(cnt) mg_arrive_scalar(cnt)
#else
#define MG_ARRIVE7(cnt) do { if (lane == 0) mg_arrive(cnt); } while (0)
#endif
// CLO_MG_SPOLL bits: 1 colA, 2 rowA, 4 rowB, 8 top, 16 colB
#define MG_WAIT(cnt, target, bit)                                              \
  do {                                                                          \
    if ((CLO_MG_SPOLL) & (bit)) {                                               \
      if (wave == 0) mg_wait_scalar((cnt), (target), c_err, lane);              \
    } else {                                                                    \
      if (tid == 0) mg_wait((cnt), (target), c_err);                            \
    }                                                                           \
  } while (0)

template <bool ACCUM>
__global__ __launch_bounds__(MG_T) void mlp_mega_kernel(const MegaArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_w = smem + MG_OFF_W, *s_b = smem + MG_OFF_B, *s_sl = smem + MG_OFF_SL, *s_d2 = smem + MG_OFF_D2;
  float *s_m = smem + MG_OFF_M;
  float *s_trash = s_m + MG_M_TRASH;
  float *s_red = smem + MG_OFF_RED;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = lane & 15, s4 = (lane >> 4) * 4;
  const int w = blockIdx.x;
  const int N = p.N, C = p.C, d0 = p.d0, d1 = p.d1, d2 = p.d2;

  // ---- geometry (all splits are balanced integer splits; see mega_ok for the bounds)
  const int S1 = d1 >> 4;
#if CLO_MG_XCD
  // Workgroup w runs on XCD w % 8 (observed, used for speed only): an XCD holds the two K ranges 2x, 2x + 1 for every
  // feature block, so a column group (the 16 workgroups that exchange a1 / delta_1) shares one L2, and the K ranges
  // are cut in PAIRS on 128-byte lines (a step = 64 B): only the cut inside a pair can fall in the middle of a
  // line, and both halves of that line are fetched by the same XCD.
  const int fb = w >> 4, kb = 2 * (w & 7) + ((w >> 3) & 1);
  int ks0, ns;
  if ((S1 & 1) == 0) {
    const int P2 = S1 >> 1, pr = kb >> 1;
    const int l0 = pr * P2 / 8, l1 = (pr + 1) * P2 / 8;               // 128-byte lines of the pair
    ns = l1 - l0;                                                     // each half: (2 (l1 - l0)) / 2 steps
    ks0 = 2 * l0 + (kb & 1) * ns;
  } else {
    ks0 = kb * S1 / 16;
    ns = (kb + 1) * S1 / 16 - ks0;
  }
#else
  const int fb = w >> 4, kb = w & 15;
  const int ks0 = kb * S1 / 16, ns = (kb + 1) * S1 / 16 - ks0;          // k16 steps of the K range
#endif
  const int k0 = ks0 * 16, kr = ns * 16;
  const int G2 = d2 >> 3;
  const int g0 = fb * G2 / 16, ng = (fb + 1) * G2 / 16 - g0;            // feature groups of the block
  const int j0 = g0 * 8, nj = ng * 8;
  const int kq = kr >> 2;
  const int q0 = fb * kq / 16, nf1 = 4 * ((fb + 1) * kq / 16 - q0);     // layer-1 slice (<= 16 features)
  const int jA = k0 + 4 * q0;
  const int jq = nj >> 2;
  const int p0 = kb * jq / 16, nf2 = 4 * ((kb + 1) * jq / 16 - p0);     // finished slice of block fb (<= 16 features)
  const int jF = j0 + 4 * p0;

  // ---- sync state of this call
  unsigned *sy = p.sync;
  const unsigned call = sy[0];
  unsigned *set = sy + 32 * (1 + (call & 1) * MG_SET_LINES);
  unsigned *c_colA = set + 32 * kb, *c_rowA = set + 32 * (16 + fb), *c_rowB = set + 32 * (32 + fb);
  unsigned *c_colB = set + 32 * (48 + kb), *c_top = set + 32 * (64 + (CLO_MG_TOPFAN ? fb : 0));
  const MgAbort c_err{sy + 1, p.fault, p.spin_limit};
  MG_STAMP(0);
  if (w == 0) {  // zero the other set for the next call
    unsigned *other = sy + 32 * (1 + ((call & 1) ^ 1) * MG_SET_LINES);
    if (tid < MG_SET_LINES) __hip_atomic_store(other + 32 * tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xch, 0, (int)(4 * (mg_off_slab2(d1, d2) + 16L * MG_NB * d1)), 0x00020000);
  const long o_a1 = mg_off_a1(d1, d2), o_da1 = mg_off_da1(d1, d2), o_slab = mg_off_slab(d1, d2);
  const long o_dphi2 = mg_off_dphi2(d1, d2), o_hp = mg_off_hp(d1, d2), o_gsum = mg_off_gsum(d1, d2);
  const long o_slab2 = mg_off_slab2(d1, d2);

  // =====================================================================================
  // issue phase.  A CU's vector-memory pipe delivers in ISSUE order (~22-28 GB/s), so whatever must arrive early is
  // issued early: the small operands of the later phases (wave 7), x fragments and layer-1 weights, then only MG_PRE
  // steps of the layer-2 tile; the rest of the tile is requested at a bounded depth while the first seam runs.
  // Every load that sits on a dependency chain is UNCONDITIONAL (clamped address + select): a predicated load makes
  // hipcc branch and wait for vmcnt(0) right behind it, which serialises the round trips.
  // =====================================================================================
  const int kpw = (int)(((d0 + 7) / 8 + 15) / 16) * 16;       // K range of a wave in layer 1
  const int kb0 = min(wave * kpw, d0);
  const int klen = min(d0, kb0 + kpw) - kb0;
  const int kbs = klen > 0 ? kb0 : 0;                          // a wave without a K range loads valid dummies
  const int jlast1 = jA + nf1 - 1;
  float *s_xw = smem + MG_OFF_XW + wave * (MG_NB * (16 * MG_P1S + 4));
  const int xq = kpw >> 2;  // float4 per row
  float4 xv[MG_P1S / 2], a1v[MG_P1S][2];
  // wave 7: b1 / Vb1 / b2 / Vb2 of the two slices, b3 / Vb3, W3 / V3 columns of the finished slice (2.3 KB, in flight
  // before any weight of this CU; parked in registers until the layer-1 MFMAs are done)
  float bpre[3] = {0.f, 0.f, 0.f}, w3pre[8];
  if (wave == MG_CWAVES) {
    const int f = lane & 15;
    const float *bsrc[3] = {lane < 16 ? p.b1 : p.Vb1, lane < 16 ? p.b2 : p.Vb2, lane < 16 ? p.b3 : p.Vb3};
    const int bidx[3] = {jA + f, jF + f, f};
    const bool bok[3] = {f < nf1, f < nf2, f < C};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool ok = bsrc[i] != nullptr && bok[i] && lane < 32;
      const float *src = ok ? bsrc[i] + bidx[i] : p.W3;
      const float v = *src;
      bpre[i] = ok ? v : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = i * 64 + lane, c = (e >> 4) & 15;
      const bool ok = c < C && f < nf2;
      const float *src = (e >> 8) ? p.V3 : p.W3;
      const float v = src[ok ? (long)c * d2 + jF + f : 0];
      w3pre[i] = ok ? v : 0.f;
    }
  }
  {
    const float *pA1[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int row = max(0, min(jA + g * 8 + (idx & 7), jlast1));   // an empty slice (nf1 == 0) loads valid dummies
      pA1[g] = ((idx >= 8) ? p.V1 : p.W1) + (long)row * d0 + kbs + s4;
    }
    // B of layer 1 = x[:, K range of the wave]: loaded once per wave with linear 16-byte loads, kept in a private
    // LDS slice [8][kpw + 4] (rows >= N and columns beyond the range zero)
#pragma unroll
    for (int i = 0; i < MG_P1S / 2; ++i) {
      const int e = i * 64 + lane, n = e / xq, c4 = (e - n * xq) * 4;
      const bool ok = n < N && c4 < klen;
      xv[i] = mg_ld4(p.X + (ok ? (long)n * d0 + kbs + c4 : 0));
    }
#pragma unroll
    for (int s = 0; s < MG_P1S; ++s) {
      const bool ok = s * 16 + s4 < klen;
#pragma unroll
      for (int g = 0; g < 2; ++g) a1v[s][g] = mg_ldw(pA1[g] + (ok ? s * 16 : 0));
    }
  }
  // layer-2 tile: fragments [W rows ; V rows] of this wave's feature groups, step-major issue order; only
  // MG_PRE steps now (they cover the latency gap behind layer 1)
  float4 tv[MG_MAXS][MG_MAXG];
  const float *pA2[MG_MAXG];
#pragma unroll
  for (int g = 0; g < MG_MAXG; ++g) {
    const int gl = min(min(wave, MG_CWAVES - 1) * MG_MAXG + g, ng - 1);
    const int row = j0 + gl * 8 + (idx & 7);
    pA2[g] = ((idx >= 8) ? p.V2 : p.W2) + (long)row * d1 + k0 + s4;
  }
  // (every tile load is UNCONDITIONAL -- steps beyond the K range re-read its last step -- so that the issue order is one
  // straight line and hipcc waits for a step by exact count instead of for everything issued so far)
  if (wave < MG_CWAVES) {
#pragma unroll
    for (int s = 0; s < MG_PRE; ++s) {
#pragma unroll
      for (int g = 0; g < MG_MAXG; ++g) tv[s][g] = mg_ldw(pA2[g] + min(s, ns - 1) * 16);
    }
  }
#pragma unroll
  for (int i = 0; i < MG_P1S / 2; ++i) {
    const int e = i * 64 + lane, n = e / xq, c4 = (e - n * xq) * 4;
    if (n < MG_NB) {
      const bool ok = n < N && c4 < klen;
      *reinterpret_cast<float4 *>(&s_xw[n * (16 * MG_P1S + 4) + c4]) = ok ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  MG_STAMP(1);

  // =====================================================================================
  // phase 1: layer 1 for features [jA, jA + nf1): in-block split-K over the 8 waves
  // =====================================================================================
  {
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const unsigned bmask1 = idx < 8 ? 0xffffffffu : 0u;
    const float *pB1 = s_xw + (idx & 7) * (16 * MG_P1S + 4) + s4;
#pragma unroll
    for (int s = 0; s < MG_P1S; ++s) {
      const bool ok = s * 16 + s4 < klen;
      const unsigned m = ok ? bmask1 : 0u;
      const float4 bv = mg_ld4(pB1 + (ok ? s * 16 : 0));
      const float bx = mg_and(bv.x, m), by = mg_and(bv.y, m), bz = mg_and(bv.z, m), bw = mg_and(bv.w, m);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#if CLO_MG_ABLATE & 1
        acc[g] += f32x4{a1v[s][g].x + bx, a1v[s][g].y + by, a1v[s][g].z + bz, a1v[s][g].w + bw};
#else
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[s][g].x, bx, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[s][g].y, by, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[s][g].z, bz, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[s][g].w, bw, acc[g], 0, 0, 0);
#endif
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4 *>(&s_red[((wave * 2 + g) * 64 + lane) * 4]) = acc[g];
    if (wave == MG_CWAVES) {  // park the small operands (their loads were the wave's first: they have landed)
      if (lane < 32) {
        s_m[MG_M_B1 + lane] = bpre[0];
        s_m[MG_M_B2 + lane] = bpre[1];
        s_m[MG_M_B3 + lane] = bpre[2];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s_m[MG_M_W3 + i * 64 + lane] = w3pre[i];
    }
  }
  wg_barrier();
  MG_STAMP(2);
  // ---- waves 0..6: paced requests of the rest of the tile (never more than MG_PACE steps beyond what has landed);
  //      wave 7: merge of the eight K parts, epilogue of layer 1 and column-group seam A
  if (wave < MG_CWAVES) {
#pragma unroll
    for (int s = MG_PRE; s < MG_HOLD; ++s) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (MG_PACE - 1)) : "memory");
#pragma unroll
      for (int g = 0; g < MG_MAXG; ++g) tv[s][g] = mg_ldw(pA2[g] + min(s, ns - 1) * 16);
    }
    MG_STAMP(3);
  } else {
    __builtin_amdgcn_s_setprio(3);
    MG_STAMP7(16);
    const int q = lane >> 4, col = lane & 15;
    f32x4 mv[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      mv[g] = *reinterpret_cast<const f32x4 *>(&s_red[(g * 64 + lane) * 4]);
#pragma unroll
      for (int wv = 1; wv < 8; ++wv) mv[g] += *reinterpret_cast<const f32x4 *>(&s_red[((wv * 2 + g) * 64 + lane) * 4]);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = mv[g][r];
        const float up = __shfl(v, (lane + 24) & 63, 64);    // D[8+i][n] = V1 x
        const float zsrc = __shfl(v, (lane + 56) & 63, 64);  // D[i][n]   = W1 x
        if (q < 2 && col >= 8) {
          const int n = col - 8, f = g * 8 + q * 4 + r;
          float aval = 0.f, daval = 0.f, dphi = 0.f;
          if (f < nf1) {
            aval = act_apply(p.act1, zsrc + s_m[MG_M_B1 + f], dphi);
            daval = dphi * (up + s_m[MG_M_B1 + 16 + f]);
          }
          s_m[MG_M_PUB + n * 16 + f] = aval;
          s_m[MG_M_PUB + 128 + n * 16 + f] = daval;
          s_m[MG_M_PHI1 + n * 16 + f] = dphi;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    MG_STAMP7(17);
    const int nq = nf1 >> 2;
    for (int e = lane; e < 2 * MG_NB * nq; e += 64) {
      const int which = e / (MG_NB * nq), n = (e / nq) % MG_NB, q2 = e % nq;
      const float *src = &s_m[MG_M_PUB + which * 128 + n * 16 + q2 * 4];
      st_x(rs, (which ? o_da1 : o_a1) + (long)n * d1 + jA + q2 * 4, f32x4{src[0], src[1], src[2], src[3]});
    }
    drain_vm();
    MG_STAMP7(18);
#if CLO_MG_SPOLL & 1
    MG_ARRIVE7(c_colA);
    mg_wait_scalar(c_colA, 16u, c_err, lane);
#else
    if (lane == 0) {
      mg_arrive(c_colA);
      mg_wait(c_colA, 16u, c_err);
    }
#endif
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    MG_STAMP7(19);
    // gather [a1 ; da1] of the K range: all (<= 11) loads of a lane in flight at once
    {
      const int q4 = kr >> 2, tot = 16 * q4;
      f32x4 gv[MG_MAXS];
      int gdst[MG_MAXS];
#pragma unroll
      for (int i = 0; i < MG_MAXS; ++i) {
        const int e = min(i * 64 + lane, tot - 1);
        const int c = e / q4, kk = (e - c * q4) * 4;
        gv[i] = ld_x(rs, (c < 8 ? o_a1 : o_da1) + (long)(c & 7) * d1 + k0 + kk);
        gdst[i] = c * MG_LDB + kk;
      }
#pragma unroll
      for (int i = 0; i < MG_MAXS; ++i) *reinterpret_cast<f32x4 *>(&s_b[gdst[i]]) = gv[i];  // clamped duplicates rewrite equal data
    }
    MG_STAMP7(20);
    __builtin_amdgcn_s_setprio(0);
  }
  wg_barrier();
  MG_STAMP(4);

  // =====================================================================================
  // phase 2: partial z2 / dz2 of the tile; W fragments go to LDS for the backward pass
  // =====================================================================================
  if (wave < MG_CWAVES) {
    f32x4 acc[MG_MAXG];
#pragma unroll
    for (int g = 0; g < MG_MAXG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *pBs = s_b + idx * MG_LDB + s4;
#pragma unroll
    for (int s = 0; s < MG_MAXS; ++s) {
      if (s + MG_HOLD < MG_MAXS) {   // the steps held back during the seam are requested here, MG_HOLD steps ahead of their use
#pragma unroll
        for (int g = 0; g < MG_MAXG; ++g) tv[s + MG_HOLD][g] = mg_ldw(pA2[g] + min(s + MG_HOLD, ns - 1) * 16);
      }
      if (s < ns) {
        const float4 bv = mg_ld4(pBs + s * 16);
#pragma unroll
        for (int g = 0; g < MG_MAXG; ++g) {
          const int gl = wave * MG_MAXG + g;
#if CLO_MG_ABLATE & 1
          acc[g] += f32x4{tv[s][g].x + bv.x, tv[s][g].y + bv.y, tv[s][g].z + bv.z, tv[s][g].w + bv.w};
#else
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[s][g].x, bv.x, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[s][g].y, bv.y, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[s][g].z, bv.z, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[s][g].w, bv.w, acc[g], 0, 0, 0);
#endif
          // branch-free copy of the W half (lanes idx < 8) into the LDS tile; other lanes hit a trash slot
          float *dst = (idx < 8 && gl < ng) ? &s_w[(gl * 8 + idx) * MG_LDW + s * 16 + s4] : &s_trash[lane * 4];
          *reinterpret_cast<float4 *>(dst) = tv[s][g];
        }
      }
    }
    // z[n][i] = D[i][n], dz[n][i] = D[i][8+n] + D[8+i][n]  ->  s_sl[which][n][local feature]
    const int q = lane >> 4, col = lane & 15;
#pragma unroll
    for (int g = 0; g < MG_MAXG; ++g) {
      const int gl = wave * MG_MAXG + g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[g][r];
        const float up = __shfl(v, (lane + 24) & 63, 64);
        const float zsrc = __shfl(v, (lane + 56) & 63, 64);
        if (q < 2 && col >= 8 && gl < ng) {
          const int n = col - 8, jl = gl * 8 + q * 4 + r;
          s_sl[n * MG_NJ + jl] = zsrc;
          s_sl[(MG_NB + n) * MG_NJ + jl] = v + up;
        }
      }
    }
  }
  wg_barrier();
  MG_STAMP(5);
  // ---- row-group seam A: publish slab[kb][which][n][j0 .. j0 + nj)
  {
    const int nq = nj >> 2;
    for (int e = tid; e < 2 * MG_NB * nq; e += MG_T) {
      const int wn = e / nq, q = e - wn * nq;  // wn = which * 8 + n
      const float *src = &s_sl[wn * MG_NJ + q * 4];
      st_x(rs, o_slab + ((long)kb * 16 + wn) * d2 + j0 + q * 4, f32x4{src[0], src[1], src[2], src[3]});
    }
    // rank-M curvature: the backpropagated vectors (<= 4 KB) travel to LDS while the slabs are acknowledged
    const int naux = p.kind == CLO_LOSS_RANK1 ? N * p.aux_rank * C                    // <= MG_AUX_MAX (mega_ok)
                     : loss_is_ef(p.kind) ? (int)ef_target_floats(p.kind, N, C) : 0;   // targets of the in-kernel EF
    const float *abase = naux ? p.aux : p.W3;
    float auxv[MG_AUX_MAX / MG_T];
#pragma unroll
    for (int i = 0; i < MG_AUX_MAX / MG_T; ++i) auxv[i] = abase[i * MG_T + tid < naux ? i * MG_T + tid : 0];
    drain_vm();
#pragma unroll
    for (int i = 0; i < MG_AUX_MAX / MG_T; ++i)
      if (i * MG_T + tid < naux) s_m[MG_M_AUX + i * MG_T + tid] = auxv[i];
    wg_barrier();
    MG_ARRIVE_B(c_rowA, 2);
    MG_WAIT(c_rowA, 16u, 2);
    wg_barrier();
    MG_STAMP(6);
  }
  // =====================================================================================
  // phase 3: finish features [jF, jF + nf2) of block fb; head partials
  // =====================================================================================
  f32x4 d2_ph[4];
  float4 d2_w[MG_CMAX];
  {
    const int nq = nf2 >> 2;
    if (tid < 2 * MG_NB * nq) {  // thread = (which, n, quad): 16 slabs in flight, summed in order
      const int wn = tid / nq, q = tid - wn * nq;
      f32x4 t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = ld_x(rs, o_slab + ((long)k * 16 + wn) * d2 + jF + q * 4);
      f32x4 sacc = t[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) sacc += t[k];
      float *dst = &s_sl[wn * 16 + q * 4];  // s_sl reused: [which*8+n][16]
      dst[0] = sacc.x; dst[1] = sacc.y; dst[2] = sacc.z; dst[3] = sacc.w;
    }
    wg_barrier();
    if (tid < MG_NB * 16) {  // (n, f): bias, activation
      const int n = tid >> 4, f = tid & 15;
      float aval = 0.f, daval = 0.f, dphi = 0.f;
      if (f < nf2) {
        aval = act_apply(p.act2, s_sl[n * 16 + f] + s_m[MG_M_B2 + f], dphi);
        daval = dphi * (s_sl[(MG_NB + n) * 16 + f] + s_m[MG_M_B2 + 16 + f]);
      }
      s_m[MG_M_FIN + n * 16 + f] = aval;
      s_m[MG_M_FIN + 128 + n * 16 + f] = daval;
      s_sl[256 + n * 16 + f] = dphi;
    }
    wg_barrier();
    // publish phi'2 of the slice and the head partials hp[w][which][n][c]
    if (tid < MG_NB * nq) {
      const int n = tid / nq, q = tid - n * nq;
      const float *src = &s_sl[256 + n * 16 + q * 4];
      st_x(rs, o_dphi2 + (long)n * d2 + jF + q * 4, f32x4{src[0], src[1], src[2], src[3]});
    }
    if (tid >= 256) {
      const int e = tid - 256, which = e >> 7, n = (e >> 4) & 7, c = e & 15;
      float o = 0.f;
      const float *wl = &s_m[MG_M_W3 + c * 16], *vl = &s_m[MG_M_W3 + 256 + c * 16];
      const float *av = &s_m[MG_M_FIN + n * 16], *dav = &s_m[MG_M_FIN + 128 + n * 16];
      if (which == 0) {
#pragma unroll
        for (int f = 0; f < 16; ++f) o = fmaf(wl[f], av[f], o);
      } else {
#pragma unroll
        for (int f = 0; f < 16; ++f) o = fmaf(wl[f], dav[f], fmaf(vl[f], av[f], o));
      }
      s_sl[512 + e] = o;
    }
    wg_barrier();
    if (tid < 64) {
      const float *src = &s_sl[512 + tid * 4];
      st_x(rs, o_hp + (long)w * 256 + tid * 4, f32x4{src[0], src[1], src[2], src[3]});
    }
    drain_vm();
    wg_barrier();
    MG_STAMP(7);
    MG_ARRIVE_B(c_rowB, 4);
    // ---- leader of the row group: sum the 16 partials, publish, arrive on the chip-wide counter
    if (kb == 0) {
      MG_WAIT(c_rowB, 16u, 4);
      wg_barrier();
      if (tid < 64) {
        f32x4 t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = ld_x(rs, o_hp + (long)(fb * 16 + k) * 256 + tid * 4);
        f32x4 sacc = t[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) sacc += t[k];
        st_x(rs, o_gsum + (long)fb * 256 + tid * 4, sacc);
      }
      drain_vm();
      wg_barrier();
#if CLO_MG_TOPFAN
      if (tid < 16) mg_arrive(set + 32 * (64 + tid));
#else
      MG_ARRIVE_B(c_top, 8);
#endif
    }
    MG_WAIT(c_top, 16u, 8);
    // every workgroup has read the call counter long before all leaders arrived: safe to bump it now
    if (tid == 0 && w == 0) sy[0] = call + 1;
    wg_barrier();
    MG_STAMP(8);
    // operands of delta_2 for thread = (half of the rows, feature quad): the W3 columns of a quad are loaded ONCE
    // for four rows (one thread per (row, quad) moved 54 KB of W3 per workgroup and delta_2 waited for it);
    // in flight together with the group sums below
    {
      const int nq2 = nj >> 2;
      const bool has = tid < 2 * nq2;
      const int e2 = has ? tid : 0, h2 = e2 / nq2, q2 = e2 - h2 * nq2;
#pragma unroll
      for (int i = 0; i < 4; ++i) d2_ph[i] = ld_x(rs, o_dphi2 + (long)(h2 * 4 + i) * d2 + j0 + q2 * 4);
#pragma unroll
      for (int c = 0; c < MG_CMAX; ++c)
        if (c < C) d2_w[c] = mg_ld4(p.W3 + (long)c * d2 + j0 + q2 * 4);
    }
    if (tid < 64) {  // f / J v = bias + sum of the 16 group sums
      f32x4 t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = ld_x(rs, o_gsum + (long)k * 256 + tid * 4);
      f32x4 sacc = t[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) sacc += t[k];
      const int which = tid >> 5, n = (tid >> 2) & 7, c4 = (tid & 3) * 4;
      const float *bias = &s_m[MG_M_B3 + which * 16 + c4];   // zero beyond C and for absent biases
      float *dst = &s_m[(which ? MG_M_U : MG_M_F) + n * 16 + c4];
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = sacc[i] + bias[i];
    }
    wg_barrier();
    if (tid < MG_NB) {  // loss Hessian per sample (as head_bwd_kernel of mlp.hip)
      const int n = tid;
      float *dl = &s_m[MG_M_DL + n * 16];
      const float *fn = &s_m[MG_M_F + n * 16], *un = &s_m[MG_M_U + n * 16];
      for (int c = 0; c < MG_CMAX; ++c) dl[c] = 0.f;
      if (n < N) {
        if (p.kind == CLO_LOSS_MSE) {
          for (int c = 0; c < C; ++c) dl[c] = p.scale * un[c];
        } else if (p.kind == CLO_LOSS_BCE) {
          for (int c = 0; c < C; ++c) {
            dl[c] = p.scale * sigmoid_prime(fn[c]) * un[c];
          }
        } else if (p.kind == CLO_LOSS_CE) {
          float mx = -INFINITY;
          for (int c = 0; c < C; ++c) mx = fmaxf(mx, fn[c]);
          float se = 0.f, spu = 0.f;
          for (int c = 0; c < C; ++c) {
            const float e = __expf(fn[c] - mx);
            se += e;
            spu += e * un[c];
          }
          const float inv = 1.f / se, pu = spu * inv;
          for (int c = 0; c < C; ++c) dl[c] = p.scale * (__expf(fn[c] - mx) * inv) * (un[c] - pu);
        } else if (loss_is_ef(p.kind)) {
          float g[MG_CMAX];
          ef_grad_row<MG_CMAX>(p.kind, fn, ef_target_row(p.kind, &s_m[MG_M_AUX], n, C), C, g);
          float sdot = 0.f;
          for (int c = 0; c < C; ++c) sdot += g[c] * un[c];
          for (int c = 0; c < C; ++c) dl[c] = p.scale * g[c] * sdot;
        } else {
          for (int m = 0; m < p.aux_rank; ++m) {
            const float *g = &s_m[MG_M_AUX + (n * p.aux_rank + m) * C];
            float sdot = 0.f;
            for (int c = 0; c < C; ++c) sdot += g[c] * un[c];
            for (int c = 0; c < C; ++c) dl[c] += p.scale * g[c] * sdot;
          }
        }
      }
    }
    wg_barrier();
    MG_STAMP(9);
  }
  // =====================================================================================
  // phase 4: delta_2 of block fb, partial delta_1 from the LDS tile, then the write-only work
  // =====================================================================================
  {
    const int nq = nj >> 2;
    if (tid < 2 * nq) {  // (row half, quad): phi'2 (exchanged) x (delta_3 W3), operands prefetched above
      const int h = tid / nq, q = tid - h * nq;
      f32x4 sacc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) sacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < MG_CMAX; ++c) {
        if (c < C) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float dl = s_m[MG_M_DL + (h * 4 + i) * 16 + c];
            sacc[i].x = fmaf(dl, d2_w[c].x, sacc[i].x); sacc[i].y = fmaf(dl, d2_w[c].y, sacc[i].y);
            sacc[i].z = fmaf(dl, d2_w[c].z, sacc[i].z); sacc[i].w = fmaf(dl, d2_w[c].w, sacc[i].w);
          }
        }
      }
      // delta_2 is kept [feature][row]: one float4 = the four rows of this half
#pragma unroll
      for (int f = 0; f < 4; ++f)
        *reinterpret_cast<f32x4 *>(&s_d2[(q * 4 + f) * MG_NB + h * 4]) =
            f32x4{sacc[0][f] * d2_ph[0][f], sacc[1][f] * d2_ph[1][f], sacc[2][f] * d2_ph[2][f], sacc[3][f] * d2_ph[3][f]};
    }
    wg_barrier();
    MG_STAMP(10);
    // partial delta_1[n][k] = sum_{j in block} delta_2[n][j] W2[j][k] on the VALU: thread = (row lane rl, column quad
    // cq) walks rows rl, rl + rpp, ... of the LDS tile with PACKED FMAs (the sweep is VALU-bound: 512 plain FMAs per
    // thread took 2 us, and everything that writes results waits for it), four rows per trip with their LDS reads in
    // flight together; the row lanes are merged through the tile area once every wave is done with it
    const int ncq = kr >> 2, rpp = min(MG_T / ncq, 16);   // <= 16 row lanes: their partials fit the tile area
    const int rl = tid / ncq, cq = tid - rl * ncq;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    {
      f32x2 pa[MG_NB][2];
#pragma unroll
      for (int n = 0; n < MG_NB; ++n) pa[n][0] = pa[n][1] = f32x2{0.f, 0.f};
      if (rl < rpp) {
        for (int jb = rl; jb < nj; jb += 4 * rpp) {
          float4 wv[4], da[4], db[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {   // rows beyond the block repeat the last one; they are skipped below
            const int j = min(jb + u * rpp, nj - 1);
            wv[u] = mg_ld4(&s_w[j * MG_LDW + cq * 4]);
            da[u] = mg_ld4(&s_d2[j * MG_NB]);
            db[u] = mg_ld4(&s_d2[j * MG_NB + 4]);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (jb + u * rpp < nj) {
              const float dn[MG_NB] = {da[u].x, da[u].y, da[u].z, da[u].w, db[u].x, db[u].y, db[u].z, db[u].w};
              const f32x2 w01 = {wv[u].x, wv[u].y}, w23 = {wv[u].z, wv[u].w};
#if CLO_MG_ABLATE & 2
              pa[0][0] += w01 + f32x2{dn[0], dn[1]};
              pa[0][1] += w23 + f32x2{dn[2], dn[3]};
              pa[1][0] += f32x2{dn[4], dn[5]};
              pa[1][1] += f32x2{dn[6], dn[7]};
#else
#pragma unroll
              for (int n = 0; n < MG_NB; ++n) {
                const f32x2 dd = {dn[n], dn[n]};
                pa[n][0] = __builtin_elementwise_fma(dd, w01, pa[n][0]);
                pa[n][1] = __builtin_elementwise_fma(dd, w23, pa[n][1]);
              }
#endif
            }
          }
        }
      }
      float4 pacc[MG_NB];
#pragma unroll
      for (int n = 0; n < MG_NB; ++n) pacc[n] = make_float4(pa[n][0].x, pa[n][0].y, pa[n][1].x, pa[n][1].y);
      MG_STAMP(21);
      wg_barrier();  // nobody reads the W2 tile any more: its area takes the partials [rl][n][kr]
      MG_STAMP(22);
      if (rl < rpp) {
#pragma unroll
        for (int n = 0; n < MG_NB; ++n)
          *reinterpret_cast<float4 *>(&s_w[(rl * MG_NB + n) * MG_KR + cq * 4]) = pacc[n];
      }
    }
    wg_barrier();
    MG_STAMP(11);
    {
      for (int e = tid; e < MG_NB * ncq; e += MG_T) {
        const int n = e / ncq, q = e - n * ncq;
        float4 sacc = mg_ld4(&s_w[n * MG_KR + q * 4]);
        for (int r = 1; r < rpp; ++r) {
          const float4 t = mg_ld4(&s_w[(r * MG_NB + n) * MG_KR + q * 4]);
          sacc.x += t.x; sacc.y += t.y; sacc.z += t.z; sacc.w += t.w;
        }
        st_x(rs, o_slab2 + ((long)fb * MG_NB + n) * d1 + k0 + q * 4, f32x4{sacc.x, sacc.y, sacc.z, sacc.w});
      }
      drain_vm();
      wg_barrier();
      MG_ARRIVE_B(c_colB, 16);
      MG_STAMP(12);
    }
    // ---- write-only work: out_W2 tile, out_b2 / out_W3 of the finished slice, out_b3
    {
      if (rl < rpp) {   // out_W2[j][k] = sum_n delta_2[n][j] a1[n][k]: contiguous 16-byte stores along a row, packed FMAs
        f32x2 av[MG_NB][2];
#pragma unroll
        for (int n = 0; n < MG_NB; ++n) {
          const float4 t = mg_ld4(&s_b[n * MG_LDB + cq * 4]);
          av[n][0] = f32x2{t.x, t.y};
          av[n][1] = f32x2{t.z, t.w};
        }
        for (int j = rl; j < nj; j += rpp) {
          const float4 da = mg_ld4(&s_d2[j * MG_NB]), db = mg_ld4(&s_d2[j * MG_NB + 4]);
          const float dn[MG_NB] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
          f32x2 o01 = {0.f, 0.f}, o23 = {0.f, 0.f};
#if CLO_MG_ABLATE & 4
          o01 = av[0][0] + f32x2{dn[0], dn[1]} + f32x2{dn[4], dn[5]};
          o23 = av[0][1] + f32x2{dn[2], dn[3]} + f32x2{dn[6], dn[7]};
#else
#pragma unroll
          for (int n = 0; n < MG_NB; ++n) {
            const f32x2 dd = {dn[n], dn[n]};
            o01 = __builtin_elementwise_fma(dd, av[n][0], o01);
            o23 = __builtin_elementwise_fma(dd, av[n][1], o23);
          }
#endif
          float4 o = make_float4(o01.x, o01.y, o23.x, o23.y);
          float *po = p.O2 + (long)(j0 + j) * d1 + k0 + cq * 4;
          if (ACCUM) {
            const float4 od = mg_ld4(po);
            o.x = fmaf(p.beta, od.x, o.x); o.y = fmaf(p.beta, od.y, o.y);
            o.z = fmaf(p.beta, od.z, o.z); o.w = fmaf(p.beta, od.w, o.w);
          }
          mg_st4nt(po, o);
        }
      }
      if (kb == 0 && p.Ob2) {  // out_b2 of the whole block by its first workgroup
        for (int j = tid; j < nj; j += MG_T) {
          float sb = 0.f;
#pragma unroll
          for (int n = 0; n < MG_NB; ++n) sb += s_d2[j * MG_NB + n];
          p.Ob2[j0 + j] = (ACCUM ? p.beta * p.Ob2[j0 + j] : 0.f) + sb;
        }
      }
      if (tid < MG_CMAX * 16) {  // out_W3[c][jF + f] = sum_n delta_3[n][c] a2[n][f]
        const int c = tid >> 4, f = tid & 15;
        if (c < C && f < nf2) {
          float o = 0.f;
#pragma unroll
          for (int n = 0; n < MG_NB; ++n) o = fmaf(s_m[MG_M_DL + n * 16 + c], s_m[MG_M_FIN + n * 16 + f], o);
          float *po = p.O3 + (long)c * d2 + jF + f;
          *po = (ACCUM ? p.beta * *po : 0.f) + o;
        }
      }
      if (w == 0 && tid < C && p.Ob3) {
        float sb = 0.f;
        for (int n = 0; n < N; ++n) sb += s_m[MG_M_DL + n * 16 + tid];
        p.Ob3[tid] = (ACCUM ? p.beta * p.Ob3[tid] : 0.f) + sb;
      }
    }
  }
  // =====================================================================================
  // phase 5: delta_1 of the layer-1 slice, out_W1 rows, out_b1
  // =====================================================================================
  {
    MG_STAMP(13);
    // x rows for the out_W1 products: requested before the wait (L2 hits that would otherwise follow the slab loads)
    const int ncq = d0 >> 2, lanes_r = MG_T / ncq;   // d0 <= 1024: at least two row lanes
    const int rl = tid / ncq, cq = tid - rl * ncq;
    float4 xv[MG_NB];
#pragma unroll
    for (int n = 0; n < MG_NB; ++n) xv[n] = mg_ld4(p.X + (long)min(n, N - 1) * d0 + (rl < lanes_r ? cq * 4 : 0));
    MG_WAIT(c_colB, 16u, 16);
    // Behind the LAST wait of the launch: was any wait of this launch cut short (here or in a workgroup this one waited
    // for)?  Then what this workgroup computes is garbage, and it says so in its own output (below): a product that
    // timed out contains NaN instead of plausible numbers ("asynchronous faults", clo_common.h).  The load returns
    // behind the x rows requested above; it is consumed after the last store.
    unsigned launch_bad = 0u;
    if (tid == 0) launch_bad = __hip_atomic_load(c_err.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wg_barrier();
    MG_STAMP(14);
    const int nq = nf1 >> 2;
    if (tid < MG_NB * nq) {
      const int n = tid / nq, q = tid - n * nq;
      f32x4 t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = ld_x(rs, o_slab2 + ((long)k * MG_NB + n) * d1 + jA + q * 4);
      f32x4 sacc = t[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) sacc += t[k];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        s_m[MG_M_D1 + (q * 4 + i) * MG_NB + n] = n < N ? sacc[i] * s_m[MG_M_PHI1 + n * 16 + q * 4 + i] : 0.f;
    }
    wg_barrier();
    if (tid < nf1 && p.Ob1) {
      float sb = 0.f;
#pragma unroll
      for (int n = 0; n < MG_NB; ++n) sb += s_m[MG_M_D1 + tid * MG_NB + n];
      p.Ob1[jA + tid] = (ACCUM ? p.beta * p.Ob1[jA + tid] : 0.f) + sb;
    }
    if (rl < lanes_r) {
      // (rows n >= N carry delta_1 = 0 below: their clamped x rows drop out)
      for (int f = rl; f < nf1; f += lanes_r) {
        const float4 da = mg_ld4(&s_m[MG_M_D1 + f * MG_NB]), db = mg_ld4(&s_m[MG_M_D1 + f * MG_NB + 4]);
        const float dn[MG_NB] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#if CLO_MG_ABLATE & 8
        o = make_float4(xv[0].x + dn[0] + dn[4], xv[0].y + dn[1] + dn[5], xv[0].z + dn[2] + dn[6], xv[0].w + dn[3] + dn[7]);
#else
#pragma unroll
        for (int n = 0; n < MG_NB; ++n) {
          o.x = fmaf(dn[n], xv[n].x, o.x); o.y = fmaf(dn[n], xv[n].y, o.y);
          o.z = fmaf(dn[n], xv[n].z, o.z); o.w = fmaf(dn[n], xv[n].w, o.w);
        }
#endif
        float *po = p.O1 + (long)(jA + f) * d0 + cq * 4;
        if (ACCUM) {
          const float4 od = mg_ld4(po);
          o.x = fmaf(p.beta, od.x, o.x); o.y = fmaf(p.beta, od.y, o.y);
          o.z = fmaf(p.beta, od.z, o.z); o.w = fmaf(p.beta, od.w, o.w);
        }
        mg_st4nt(po, o);
      }
    }
    // (thread 0 wrote the first four entries of row jA itself: same thread, same address, program order)
    if (tid == 0 && launch_bad != 0u) p.O1[(long)jA * d0] = __builtin_nanf("");
  }
  MG_STAMP(15);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
bool mega_shape_ok(int L, const int *dims, int N) {
  if (L != 3 || N < 1 || N > MG_NB || dims[3] > MG_CMAX) return false;
  const int d0 = dims[0], d1 = dims[1], d2 = dims[2];
  if (d0 % 16 || d1 % 16 || d2 % 8 || d0 < 16) return false;
  if (d0 > 16 * MG_P1S * 8) return false;                       // layer-1 K range of a wave: <= 8 steps
  if (d1 < 256 || cdiv(d1 / 16, 16) > MG_MAXS) return false;    // K ranges: 1 .. 11 steps, slices of >= 4 features
  if (d2 < 512 || cdiv(d2 / 8, 16) > MG_CWAVES * MG_MAXG) return false;
  return true;
}

bool mega_ok(int L, const int *dims, const float *const *W, const float *const *VW, float *const *OW,
             const float *X, int N, int loss_kind, int aux_rank) {
  if (!mega_shape_ok(L, dims, N)) return false;
  if (loss_kind == CLO_LOSS_RANK1 && (long)N * aux_rank * dims[3] > MG_AUX_MAX) return false;  // staged in LDS
  for (int l = 0; l < 3; ++l)
    if (!aligned16(W[l]) || !aligned16(VW[l]) || !aligned16(OW[l])) return false;
  if (!aligned16(X)) return false;
  static int ncu[MG_MAXDEV];  // per device (a process may drive several GPUs)
  static std::once_flag once;
  std::call_once(once, [] { for (int &c : ncu) c = -1; });
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MG_MAXDEV) return false;
  int c = __atomic_load_n(&ncu[dev], __ATOMIC_RELAXED);
  if (c < 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    c = prop.multiProcessorCount;
    __atomic_store_n(&ncu[dev], c, __ATOMIC_RELAXED);
  }
  if (c != MG_G) return false;  // one workgroup per CU, all of them resident: the group counters rely on it
  if (!fault_words_device(dev)) return false;           // no host-visible fault word: a timeout could not be reported
  if (fault_disabled(dev, FAULT_MEGA)) return false;   // a launch on this device timed out before: the launch chain serves
  // every workgroup needs a CU to itself AND one must fit at all (LDS carve, registers) on this device
  static int occ[MG_MAXDEV];
  static std::once_flag once_occ;
  std::call_once(once_occ, [] { for (int &o : occ) o = -1; });
  int o = __atomic_load_n(&occ[dev], __ATOMIC_RELAXED);
  if (o < 0) {
    int nb = 0;
    const void *fn = reinterpret_cast<const void *>(mlp_mega_kernel<false>);
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MG_LDS_FLOATS * sizeof(float)));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, MG_T, MG_LDS_FLOATS * sizeof(float)) != hipSuccess) {
      (void)hipGetLastError();
      nb = 0;
    }
    o = nb >= 1 ? 1 : 0;
    __atomic_store_n(&occ[dev], o, __ATOMIC_RELAXED);
  }
  return o == 1;
}

int mega_launch(const int *dims, const int *acts, const float *const *W, const float *const *b,
                const float *const *VW, const float *const *Vb, float *const *OW, float *const *Ob,
                const float *X, int N, int loss_kind, const float *aux, int aux_rank, float scale,
                float beta, float *xch, unsigned *sync, hipStream_t st) {
  MegaArgs a{};
  a.W1 = W[0]; a.V1 = VW[0]; a.W2 = W[1]; a.V2 = VW[1]; a.W3 = W[2]; a.V3 = VW[2];
  a.b1 = b ? b[0] : nullptr; a.b2 = b ? b[1] : nullptr; a.b3 = b ? b[2] : nullptr;
  a.Vb1 = Vb ? Vb[0] : nullptr; a.Vb2 = Vb ? Vb[1] : nullptr; a.Vb3 = Vb ? Vb[2] : nullptr;
  a.O1 = OW[0]; a.O2 = OW[1]; a.O3 = OW[2];
  a.Ob1 = Ob ? Ob[0] : nullptr; a.Ob2 = Ob ? Ob[1] : nullptr; a.Ob3 = Ob ? Ob[2] : nullptr;
  a.X = X; a.N = N; a.d0 = dims[0]; a.d1 = dims[1]; a.d2 = dims[2]; a.C = dims[3];
  a.act1 = acts[0]; a.act2 = acts[1];
  a.kind = loss_kind; a.aux = aux; a.aux_rank = aux_rank; a.scale = scale; a.beta = beta;
  a.xch = xch; a.sync = sync;
  a.spin_limit = spin_limit();
  const size_t smem = (size_t)MG_LDS_FLOATS * sizeof(float);
  static bool attr_done[MG_MAXDEV][2];  // hipFuncSetAttribute is per device; guarded by `mu` below
  const int v = beta != 0.f ? 1 : 0;
  const void *fns[2] = {reinterpret_cast<const void *>(mlp_mega_kernel<false>),
                        reinterpret_cast<const void *>(mlp_mega_kernel<true>)};
  // Persistent grids never share the chip partly resident: csrc/persist_gate.h makes this stream wait (device side) for
  // the persistent launches other streams still have in flight until all of them plus this one (every CU) fit.
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MG_MAXDEV) {
    set_error("mlp_mega_kernel: device ordinal %d out of range", dev);
    return CLO_EINVAL;
  }
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!attr_done[dev][v]) {
      int rc = check_hip(hipFuncSetAttribute(fns[v], hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem),
                         "hipFuncSetAttribute(mlp_mega_kernel)");
      if (rc != CLO_OK) return rc;
      attr_done[dev][v] = true;
    }
  }
  a.fault = fault_words_device(dev);
  if (a.fault) a.fault += FAULT_MEGA;
  PersistGate &gate = PersistGate::of(dev);
  {
    int rc = gate.admit(st, MG_G);
    if (rc != CLO_OK) return rc;
  }
  const double D = (double)dims[0] * dims[1] + (double)dims[1] * dims[2] + (double)dims[2] * dims[3];
  {
    ProfScope prof(6, 12.0 * D, st);
    if (v == 0) hipLaunchKernelGGL((mlp_mega_kernel<false>), dim3(MG_G), dim3(MG_T), smem, st, a);
    else hipLaunchKernelGGL((mlp_mega_kernel<true>), dim3(MG_G), dim3(MG_T), smem, st, a);
  }
  {
    int rc = check_hip(hipGetLastError(), "mlp_mega_kernel");
    if (rc != CLO_OK) {
      gate.abort();
      return rc;
    }
    rc = gate.done(st);
    if (rc != CLO_OK) return rc;
  }
  return CLO_OK;
}

}  // namespace clo
