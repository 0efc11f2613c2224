// arap_resident.h -- the whole PCG loop of one Gauss-Newton step in ONE launch, state on chip.
//
// Why: at 854x480 one PCG iteration moves at most 65 MB, ~10 us at HBM speed, and a DAVIS-shaped frame
// (25 % of the vertices active) 16 MB; the two dependent global reductions per iteration make a
// kernel-per-phase solve latency bound (profiles/r01_v1_*: 20-27 us per kernel for 8 frames, 75 % of the
// workgroups empty).  Here every frame of the batch is solved by a GROUP of workgroups (256 threads, TWO
// workgroups per CU so one computes while the other waits; the host sizes the group by the frame's
// active tiles, ResWg below) that stays resident for all lIterations iterations:
//   * r, delta, M^-1, flags and the transient Ap live in registers (9 tile slots of 32x8 vertices per
//     workgroup, one vertex per lane per slot; tiles start at each 8-row band's first active vertex),
//   * the search direction p and cos/sin(A) live in LDS as 34x10 halo'd tiles (the stencil reads
//     every neighbour from LDS, one tile slot ahead of the arithmetic; cells that are not active
//     vertices are zero and invalid edges get a zero weight, so the phases are branch free),
//   * per iteration the group exchanges only (a) two 16-byte partial sums per workgroup (all-gather
//     of data-tagged granules, summed by every workgroup in the same fixed order -> deterministic) and
//     (b) the preconditioned residual z of tile-border vertices, from which each workgroup rebuilds
//     its halo of p = z + beta p locally, so there are exactly TWO group-wide waits per iteration,
//     the two the algorithm cannot avoid.
// Inter-workgroup visibility: everything one workgroup hands to another travels as DATA-TAGGED granules,
// {tag << 32 | 32 payload bits} written with one 8-byte store (agent scope, write-through; plain when the
// readers provably share the writer's L2) and read with sc1 loads that bypass the reader's L1: a reader
// that sees the tag of the iteration it is in has that iteration's payload -- no flag after the data, no
// "my stores have landed" wait on the writer's side, no ordering assumed between two stores.  The partial
// sums are polled (group_sum*); the border z is written by waves 1-3 while wave 0 runs the sum, read after
// the sum and its tags checked (they all but always match: the sum took longer than a store travels).
// Correctness never depends on placement; dealing a group workgroups of equal blockIdx & 7 merely tends to
// keep it on one XCD.  Every spin is bounded; a timeout sets an error word and the host redoes the step on
// the two-kernel path.
//
// Arithmetic: the same float32 operation list as k_pcg_a / k_pcg_b (and the CPU oracle), for the
// pixel-grid UrShape the frame solver always uses (CombinedSolver.h:207-221): d_s = U(c)-U(n) = -s.
#pragma once
#include <type_traits>
#include "arap_device.h"

namespace arap {

constexpr int RES_WGS = 512;             // workgroups per launch: TWO per CU of an MI355X (256 CUs), so that one
                                         // workgroup computes while its CU-mate (another frame's group) waits
constexpr int RES_MAX_GROUPS = 16;       // most equal groups ARAPOPT_RES_GROUPS may force (experiments)
constexpr int RES_THREADS = 256;         // 4 wavefronts = 4 x 2 rows of a 32x8 tile, one per SIMD
constexpr int RES_SLOTS = 9;             // tile slots per workgroup (register arrays, fully unrolled)
constexpr int RES_TILES_PER_WG = RES_SLOTS;
// Tile of the resident kernel: 32 x 8 vertices, one vertex per lane: a wavefront holds two rows of 32 (lanes 0-31 and
// 32-63), the four wavefronts the eight rows.  Tiles live in BANDS of 8 rows; within a band they start at the band's
// first active vertex (any x, chosen by the host: ResDev::bandx0) and follow each other every 32 columns.  Against
// 64 x 4 tiles on a fixed grid this cuts the tiles of a DAVIS-shaped blob by 12 % (432 instead of 494 at 854x480:
// 7 tile slots per workgroup instead of 8-9, the phases are proportional to the slots) and the halo cells per tile
// from 136 to 80.
constexpr int RT_X = 32, RT_Y = 8;
constexpr int RES_ZX = 2 * RT_X + 2 * RT_Y;      // published z entries per tile (its border)
constexpr int RES_ZG = 2 * RES_ZX;               // ... as granules: two per entry
constexpr int RES_MAX_L = 32000;                 // PCG iterations per launch: the z granules carry 16-bit tags (2 l + 3)
static_assert(RT_X * RT_Y == RES_THREADS && RT_Y == 2 * (RES_THREADS / 64), "one vertex per lane, two rows per wavefront");
constexpr int RES_MAX_HALO = RES_TILES_PER_WG * (2 * RT_X + 2 * RT_Y);          // 80 halo cells per tile
constexpr int RES_HALO_PER_THREAD = (RES_MAX_HALO + RES_THREADS - 1) / RES_THREADS;   // 3
constexpr int RES_MAX_TILES = RES_WGS * RES_TILES_PER_WG;   // 4608 tiles (one group of 512 workgroups)
constexpr int LROW = RT_X + 2;           // 34
constexpr int LROWS = RT_Y + 2;          // 10
constexpr int LPLANE = LROW * LROWS;     // 340 floats
constexpr int LTILE = 5 * LPLANE;        // float2 (px,py) plane | float2 (cos,sin) plane | float pa plane
constexpr int RES_LDS_BYTES = RES_TILES_PER_WG * LTILE * 4      // halo'd p / cos / sin tiles
                              + ((RES_MAX_HALO * 8 + 15) / 16) * 16   // halo list (u16), then the decoded halo table (uint2)
                              + RES_TILES_PER_WG * 8 + 8 + 384  // tile origins, tables, scratch
                              + RES_TILES_PER_WG * RES_ZX * 16;      // border z on its way out  (78.9 KB: 2 per CU)
static_assert(2 * ((RES_LDS_BYTES + 1279) / 1280 * 1280) <= 160 * 1024, "two workgroups per CU");
#ifndef RES_HALO_REG_SLOTS
#define RES_HALO_REG_SLOTS 8  // keep the decoded halo entries in registers when at most this many tile slots are in use
#endif
#ifndef RES_WREG_SLOTS
#define RES_WREG_SLOTS 7      // keep the per-vertex edge weights in registers when at most this many slots are in use
#endif
#ifndef RES_WREG_8
#define RES_WREG_8 2          // edge weights (of four) kept in registers at 8 slots (3 spill; 2: +1.5 %)
#endif
#ifndef RES_WREG_9
#define RES_WREG_9 2          // ... at 9 slots
#endif
#ifndef RES_WREG_FIT
#define RES_WREG_FIT 6        // ... the fit weight too when at most this many slots are in use (else only the four edge weights)
#endif
// A workgroup's first look at the group's granules comes this long (x 64 clocks) after it has published its own: the
// others publish at about the same time and a store needs ~300 clocks to become visible, so a look taken at once finds
// nothing and costs a full L2 round trip before the next one (2.54 -> 2.14 sweeps per sum, 4.62 -> 4.54 us per iteration)
#ifndef RES_FIRST_LOOK
#define RES_FIRST_LOOK 5
#endif
#ifndef RES_FIRST_LOOK_2
#define RES_FIRST_LOOK_2 14     // second level of a two-level sum: the leaders' granules travel through the fabric
                                // (854x480 mask == 0: 0 / 8 / 12 / 16 / 20 / 24 / 32 -> 4.88 / 4.94 / 4.99 / 4.97 / 4.85 / 4.73 / 4.46 frames/s)
#endif
#ifndef RES_FIRST_LOOK_X
#define RES_FIRST_LOOK_X 20     // one-hop sum of a group of two XCD runs
                                // (1920x1080 --multseg: 0 / 5 / 12 / 16 / 20 / 24 / 28 -> 3.85 / 3.90 / 3.96 / 4.11 / 4.14 / 4.13 / 3.98 frames/s)
#endif
#ifndef RES_OWN_LOCAL
#define RES_OWN_LOCAL 1       // a group wait takes the workgroup's own partial from its register, not from its granules
#endif
#ifndef RES_POLL_SLEEP
#define RES_POLL_SLEEP 1      // s_sleep between two sweeps of a group wait (0: poll back to back)
#endif
constexpr unsigned RES_SPIN_LIMIT = 1u << 18;
// granules of one launch (u64 entries): [2 parity][wgs][2] per group, packed back to back (4 per workgroup), then the
// second-level granules of groups that span XCDs: [2 parity][8 XCD runs][16: one line each] for up to 4 such groups
#ifndef RES_GRAN_STRIDE
#define RES_GRAN_STRIDE 2     // u64 between two workgroups' granule pairs: 2 = packed (8 pairs per 128-byte line), 16 = a line each
#endif
constexpr int RES_GS = RES_GRAN_STRIDE;
// (a group of several XCD runs addresses its runs in blocks of 64 workgroups, whatever the size of its last run: the
//  host reserves whole blocks, arapopt.hip: resident_deal -- at most 7 x 128 such units in a launch, hence 2 x RES_WGS)
constexpr int RES_GRAN_UNITS = 2 * RES_WGS;
constexpr int RES_GRAN_L1 = 2 * RES_GRAN_UNITS * RES_GS;
constexpr int RES_GRAN2_STRIDE = 16;            // u64 per second-level granule pair: one 128-byte line per XCD run
constexpr int RES_GRAN2_GROUP = 2 * 8 * RES_GRAN2_STRIDE;
// third region: a write-through copy of every workgroup's granules, same offsets as the first (group_sum_x)
constexpr int RES_GRAN_X = RES_GRAN_L1 + 4 * RES_GRAN2_GROUP;
constexpr int RES_GRAN_PER_LAUNCH = RES_GRAN_X + RES_GRAN_L1;
#ifndef RES_FLAT_MAX_RUNS
#define RES_FLAT_MAX_RUNS 2   // groups of up to this many XCD runs sum with group_sum_x (one hop), wider ones in two levels
#endif

// One entry per workgroup of a launch, written by the host (arapopt.hip: plan_resident_pack): which solve the
// workgroup works on, its rank in that solve's group, the group's size and where the group's granules start.
struct ResWg {
    int slot;                   // batch slot of the solve, -1: this workgroup is idle in this launch
    int rank;                   // 0 .. wgs-1
    int wgs;                    // workgroups of the group
    int gran;                   // first granule (in u64 units) of the group: [2 parity][wgs][2]
};

struct ResDev {
    const int* tilelist;        // [batch][RES_MAX_TILES] origin vertex index (x0 + W y0) of the active tiles, band by band
    const int* ntiles;          // [batch]
    const int* tilepos;         // [batch][rtX * rtY] position in the list of the k-th tile column of a band, -1 = inactive
    const int* bandx0;          // [batch][rtY] x of the first tile of every 8-row band
    // z of the tile-BORDER vertices, what a neighbouring tile's halo needs: per tile of a solve's list RES_ZX entries
    // {top row 0..31 | bottom row 32..63 | left column 64..71 | right column 72..79} (the four corners twice), every entry
    // two granules like those of the group sums, but with 16-bit tags so that three floats fit:
    //   {z_x | low half of z_y << 32 | tag << 48}, {z_alpha | high half of z_y << 32 | tag << 48}
    unsigned long long* zx;     // [batch][RES_MAX_TILES][RES_ZX][2]
    int rtX, rtY;               // ceil(W / 32) tile columns at most per band, ceil(H / 8) bands
    unsigned long long* gran;   // [RES_GRAN_PER_LAUNCH]  {tag << 32 | 32 value bits}
    unsigned* err;              // [1] 0 = ok
    const ResWg* wgmap;         // [RES_WGS] of this launch
    int allow_fast;             // 0: always use the write-through (placement independent) store flavour
    int force_fail;             // test hook (ARAPOPT_FORCE_RES_FAIL=1): behave as if a group wait had timed out
    int flat_runs;              // groups of up to this many XCD runs use group_sum_x (ARAPOPT_FLAT_RUNS, default RES_FLAT_MAX_RUNS)
    int nowait;                 // diagnostic (ARAPOPT_RES_NOWAIT=1): one sweep per group wait, whatever the tags say (results are garbage:
                                // measures the iteration without the waits, tools/res_stamps.py)
    int fuse_update;            // frame solver: the epilogue applies the step (X += delta, cos/sin of the new Angle) itself: no k_gn_update
    unsigned long long* stamps; // diagnostic build only (STAMPS = true): [RES_WGS][16] summed s_memrealtime ticks / clocks
};

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

// entry of tile-local vertex (x, y) in its tile's border export, -1 for an interior vertex
__device__ __forceinline__ int border_entry(int x, int y)
{
    return y == 0 ? x : (y == RT_Y - 1 ? RT_X + x : (x == 0 ? 2 * RT_X + y : (x == RT_X - 1 ? 2 * RT_X + RT_Y + y : -1)));
}
// One published granule: {tag, payload} in a single 8-byte store, so a reader that sees the tag of the iteration it
// is in also sees that iteration's payload.
// Same-XCD fast path (`fast`): when every reader of a workgroup's granules reports the same XCC id (checked at run
// time, see the kernel), the XCD's L2 is the coherence point for all of them, so they may be written with workgroup-
// scope stores (they stay in that L2 instead of being written through to memory) and are still read with sc1 loads
// (which bypass the reader's L1 and are served by that same L2).
__device__ __forceinline__ void st_tagged(unsigned long long* p, unsigned hi /* tag << 16 | 16 payload bits */, float v, bool fast)
{
    const unsigned long long g = ((unsigned long long)hi << 32) | __float_as_uint(v);
    if (fast) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef float v2f_t __attribute__((ext_vector_type(2)));
// (a.x*b.x + c.x, a.y*b.y + c.y) with one rounding each: v_pk_fma_f32
__device__ __forceinline__ float2 fma2(float2 a, float2 b, float2 c)
{
    const v2f_t r = __builtin_elementwise_fma((v2f_t){a.x, a.y}, (v2f_t){b.x, b.y}, (v2f_t){c.x, c.y});
    return make_float2(r.x, r.y);
}
// v if bit `BITNO` of f is set, else +0: v_bfe_i32 (0 / all ones) + v_and_b32, no lane mask in SGPRs
template <int BITNO>
__device__ __forceinline__ float keep_if(unsigned f, float v)
{
    int m;      // (as asm: the compiler would turn sext(bit) & v back into v_and + v_cmp + v_cndmask)
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(f), "n"(BITNO));
    return __uint_as_float(__float_as_uint(v) & (unsigned)m);
}

// block_sum8 (below) leaves a workgroup's sum in lanes 2 and 3 of wave 0; the same value in every lane:
__device__ __forceinline__ double block_sum_uniform(double t)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), 3), __builtin_amdgcn_readlane(__double2loint(t), 3));
}

// Group-wide sum of one double per workgroup.  `part` is this workgroup's partial (valid in wave 0,
// lanes 2 and 3: block_sum8).  Wave 0 publishes it as two {tag, 32 bits} granules and sweeps the group's granules until
// every tag equals `epoch`; lane k owns workgroups k, k+64, ... and adds their partials in that order,
// then the fixed DPP tree above adds the lanes, so every workgroup of the group computes the same bits.
// Returns the sum rounded to float in every thread; false on timeout.
__device__ __forceinline__ bool group_sum(double part, unsigned epoch, unsigned long long* gran_group, int rank,
                                          int wgs, float* bcast /* LDS, 2 floats */, unsigned* err, float& out,
                                          bool fast = false, double* out_d = nullptr, bool nowait = false,
                                          unsigned long long* tm = nullptr /* diagnostic: publish / poll / tail cycles */)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        unsigned long long c0 = tm ? __builtin_amdgcn_s_memtime() : 0ull, c1;
        // the whole group waits for the slowest publisher: this wave's few instructions go first on its SIMD
        __builtin_amdgcn_s_setprio(3);
        unsigned long long* buf = gran_group + (size_t)(epoch & 1u) * wgs * RES_GS;
        // lanes 2 and 3 hold the partial (block_sum8) and publish one half each: nothing between the sum and the stores
        if ((lane >> 1) == 1) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(part);
            const unsigned hw = lane == 2 ? (unsigned)bits : (unsigned)(bits >> 32);
            const unsigned long long gv = ((unsigned long long)epoch << 32) | hw;
            if (fast)
                __hip_atomic_store(buf + rank * RES_GS + (lane - 2), gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                __hip_atomic_store(buf + rank * RES_GS + (lane - 2), gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        part = block_sum_uniform(part);                // (for the sweep: this workgroup's own partial, in whichever lane)
        double v = 0.0;
        bool ok = false;
        if (tm) { c1 = __builtin_amdgcn_s_memtime(); tm[0] += c1 - c0; c0 = c1; }
        __builtin_amdgcn_s_sleep(RES_FIRST_LOOK);
        for (unsigned spins = 0; spins < RES_SPIN_LIMIT; ++spins) {
            if (tm) tm[3] += 1;
            v = 0.0;
            bool mine_ok = true;
            for (int m = lane; m < wgs; m += 64) {
                const unsigned long long lo = __hip_atomic_load(buf + RES_GS * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi =
                    __hip_atomic_load(buf + RES_GS * m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // this workgroup's own granules need not have travelled to L2 and back: its value is at hand (the
                // last workgroup to arrive -- the critical path -- then finishes with its first sweep)
                const bool own = RES_OWN_LOCAL && m == rank;
                mine_ok = mine_ok && (own || ((unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch));
                v += own ? part : __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
            }
            ok = __all(mine_ok) || nowait;
            if (ok) break;
            if (RES_POLL_SLEEP) __builtin_amdgcn_s_sleep(1);
        }
        if (tm) { c1 = __builtin_amdgcn_s_memtime(); tm[1] += c1 - c0; c0 = c1; }
        v = wave_sum_l63(v);
        if (lane == 63) {
            // value with the sign bit of the second word as the "timed out" flag (one ds_write_b64); the slot
            // alternates with the epoch, so the next call's write cannot overtake a slow reader of this one
            // (before wave 0 writes slot p again, every wave has passed the barrier of the call in between)
            *(float2*)(bcast + 2 * (epoch & 1u)) = make_float2((float)v, ok ? 1.0f : 0.0f);
            if (out_d) *out_d = v;         // LDS double, read by the caller after the barrier below
            if (!ok) atomicExch(err, 0xDEAD0000u | (epoch & 0xffffu));
        }
        __builtin_amdgcn_s_setprio(0);
        if (tm) { c1 = __builtin_amdgcn_s_memtime(); tm[2] += c1 - c0; }
    }
    __syncthreads();
    const float2 bc = *(const float2*)(bcast + 2 * (epoch & 1u));
    out = bc.x;
    return bc.y != 0.0f;
}

// The same sum for a group that spans XCDs (wgs = 64 nsub; XCD run `sub` holds ranks 64 sub .. 64 sub + 63, see
// arapopt.hip: resident_deal).  A flat all-gather would have every workgroup poll every granule through the fabric
// (256 pollers x 256 granules: measured 3.4 us per wait against 0.85 us inside one XCD).  Two levels instead:
//   1. all-gather inside the XCD run exactly as above (plain granule stores when the run really sits on one XCD:
//      `subfast`, checked at run time like `fast`) -> every workgroup of the run knows its run's sum S_sub;
//   2. the first workgroup of every run publishes S_sub write-through; every workgroup polls those nsub (<= 8)
//      granules and adds them in run order -> the same bits everywhere, deterministic.
// (The border z does not depend on these sums for its visibility: its granules carry their own tags.)
__device__ __forceinline__ bool group_sum_h(double part, unsigned epoch, unsigned long long* gran_group,
                                            unsigned long long* gran2 /* [2][8][16] of this group */, int rank, int wgs,
                                            float* bcast, unsigned* err, float& out, bool subfast)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        const int sub = rank >> 6, srank = rank & 63, nsub = wgs >> 6;
        unsigned long long* buf = gran_group + (size_t)sub * (128 * RES_GS) + (size_t)(epoch & 1u) * (64 * RES_GS);
        part = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(part)),
                                __builtin_amdgcn_readfirstlane(__double2loint(part)));
        if (lane < 2) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(part);
            const unsigned hw = lane == 0 ? (unsigned)bits : (unsigned)(bits >> 32);
            const unsigned long long gv = ((unsigned long long)epoch << 32) | hw;
            if (subfast)
                __hip_atomic_store(buf + srank * RES_GS + lane, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                __hip_atomic_store(buf + srank * RES_GS + lane, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double v = 0.0;
        bool ok = false;
        __builtin_amdgcn_s_sleep(RES_FIRST_LOOK);
        for (unsigned spins = 0; spins < RES_SPIN_LIMIT; ++spins) {           // level 1: lane k <-> workgroup k of the run
            const unsigned long long lo = __hip_atomic_load(buf + RES_GS * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi = __hip_atomic_load(buf + RES_GS * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool own = RES_OWN_LOCAL && lane == srank;
            v = own ? part : __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
            ok = __all(own || ((unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch));
            if (ok) break;
            if (RES_POLL_SLEEP) __builtin_amdgcn_s_sleep(1);
        }
        const double ssub = wave_sum(v);                                      // uniform
        unsigned long long* buf2 = gran2 + (size_t)(epoch & 1u) * (8 * RES_GRAN2_STRIDE);
        if (ok && srank == 0 && lane < 2) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(ssub);
            const unsigned hw = lane == 0 ? (unsigned)bits : (unsigned)(bits >> 32);
            __hip_atomic_store(buf2 + sub * RES_GRAN2_STRIDE + lane, ((unsigned long long)epoch << 32) | hw, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        bool ok2 = false;
        v = 0.0;
        if (ok) {
            __builtin_amdgcn_s_sleep(RES_FIRST_LOOK_2);
            for (unsigned spins = 0; spins < RES_SPIN_LIMIT; ++spins) {       // level 2: lane k <-> run k
                bool mine_ok = true;
                v = 0.0;
                if (lane < nsub) {
                    const unsigned long long lo = __hip_atomic_load(buf2 + RES_GRAN2_STRIDE * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long hi = __hip_atomic_load(buf2 + RES_GRAN2_STRIDE * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool own = RES_OWN_LOCAL && srank == 0 && lane == sub;      // the run's leader knows its run's sum
                    v = own ? ssub : __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
                    mine_ok = own || ((unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch);
                }
                ok2 = __all(mine_ok);
                if (ok2) break;
                if (RES_POLL_SLEEP) __builtin_amdgcn_s_sleep(1);
            }
        }
        v = wave_sum_l63(v);
        if (lane == 63) {
            *(float2*)(bcast + 2 * (epoch & 1u)) = make_float2((float)v, ok2 ? 1.0f : 0.0f);
            if (!ok2) atomicExch(err, 0xDEAD8000u | (epoch & 0x7fffu));
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    const float2 bc = *(const float2*)(bcast + 2 * (epoch & 1u));
    out = bc.x;
    return bc.y != 0.0f;
}

// One-hop sum for a group that spans a FEW XCDs (nsub <= RES_FLAT_MAX_RUNS runs of 64 ranks).  The two levels above cost
// two latency chains in sequence (a run's all-gather through its L2, then the leader's write-through store and every
// workgroup's poll of it through the fabric: 2.1-2.3 us measured).  Here every workgroup publishes its partial TWICE --
// into its run's buffer (plain stores when the run shares an XCD: stays in that L2) and, write-through, into a second
// buffer that every XCD can read -- and polls, in the same sweep, its own run's granules from the first buffer and the
// other runs' from the second.  Lane k adds ranks k, k + 64, ... in rank order, then the fixed DPP tree: the same bits
// in every workgroup.  The fabric carries (nsub - 1) x 64 granule pairs per poller and sweep: fine for 2 runs (128
// pollers x 1 KB), too much for 4 or 8 (round 1 measured 3.4 us for a flat gather over 256 workgroups).
__device__ __forceinline__ bool group_sum_x(double part, unsigned epoch, unsigned long long* gran_group,
                                            unsigned long long* granx_group, int rank, int wgs, float* bcast, unsigned* err,
                                            float& out, bool subfast)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        // (the last run may be shorter than 64: a solve's home XCD plus a piece elsewhere, arapopt.hip: resident_deal)
        const int sub = rank >> 6, srank = rank & 63, nsub = (wgs + 63) >> 6;
        unsigned long long* bufl = gran_group + (size_t)sub * (128 * RES_GS) + (size_t)(epoch & 1u) * (64 * RES_GS);
        unsigned long long* bufx = granx_group + (size_t)(epoch & 1u) * wgs * RES_GS;      // [wgs] pairs, rank order
        part = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(part)),
                                __builtin_amdgcn_readfirstlane(__double2loint(part)));
        if (lane < 4) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(part);
            const unsigned hw = (lane & 1) == 0 ? (unsigned)bits : (unsigned)(bits >> 32);
            const unsigned long long gv = ((unsigned long long)epoch << 32) | hw;
            if (lane < 2) {
                if (subfast)
                    __hip_atomic_store(bufl + srank * RES_GS + lane, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    __hip_atomic_store(bufl + srank * RES_GS + lane, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(bufx + rank * RES_GS + (lane & 1), gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        double v = 0.0;
        bool ok = false;
        __builtin_amdgcn_s_sleep(RES_FIRST_LOOK_X);
        for (unsigned spins = 0; spins < RES_SPIN_LIMIT; ++spins) {
            v = 0.0;
            bool mine_ok = true;
            for (int r = 0; r < nsub; ++r) {                          // rank r * 64 + lane: own run locally, others remotely
                if (r * 64 + lane >= wgs) break;                      // (the last run may be short)
                const unsigned long long* src = r == sub ? bufl + RES_GS * lane : bufx + RES_GS * (r * 64 + lane);
                const unsigned long long lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mine_ok = mine_ok && (unsigned)(lo >> 32) == epoch && (unsigned)(hi >> 32) == epoch;
                v += __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
            }
            ok = __all(mine_ok);
            if (ok) break;
            if (RES_POLL_SLEEP) __builtin_amdgcn_s_sleep(1);
        }
        v = wave_sum_l63(v);
        if (lane == 63) {
            *(float2*)(bcast + 2 * (epoch & 1u)) = make_float2((float)v, ok ? 1.0f : 0.0f);
            if (!ok) atomicExch(err, 0xDEAD4000u | (epoch & 0x3fffu));
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    const float2 bc = *(const float2*)(bcast + 2 * (epoch & 1u));
    out = bc.x;
    return bc.y != 0.0f;
}

// Append `val` to the workgroup's halo list for every lane with `take`: the wavefront reserves its entries with ONE LDS
// atomic and every taker writes at its rank among the takers.
__device__ __forceinline__ void halo_append(bool take, unsigned short val, unsigned short* hlist, int* nhalo)
{
    const unsigned long long m = __ballot(take);
    if (m == 0ull) return;                             // (uniform)
    const int first = __builtin_ctzll(m);
    int base = 0;
    if ((int)(threadIdx.x & 63) == first) base = atomicAdd(nhalo, __popcll(m));
    base = __builtin_amdgcn_readlane(base, first);
    const int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (take) hlist[base + before] = val;
}

// block-wide sum of a double over the 4 wavefronts; result valid in LANES 2 AND 3 of wave 0 (the lanes that publish
// it, group_sum).  Lane k of wave 0 reads the sums of wavefronts k & 3 and (k & 3) ^ 1 -- two independent ds_read_b64, four
// VGPRs -- adds them (lanes 0, 1: w0 + w1; lanes 2, 3: w2 + w3; the same bits in both lanes of a pair) and one DPP step
// (row_shr:2) makes (w2 + w3) + (w0 + w1) in lanes 2 and 3.  (Lane 0 reading all four took two ds_read2_b64 into eight
// VGPRs and four dependent additions; with fewer registers to spare the compiler issued the second read after the first
// had returned: one more LDS round trip on the critical path of every group sum, 2 % of the run.)
__device__ __forceinline__ double block_sum8(double v, double* wsum /* LDS, 4 doubles */)
{
    static_assert(RES_THREADS / 64 == 4, "one pair sum and one DPP step add four wavefronts");
    v = wave_sum_l63(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (wave == 0) {
        const double a = wsum[lane & 3], b = wsum[(lane & 3) ^ 1];
        t = dpp_add_f64<0x112, 0xf>(a + b);
    }
    return t;
}

// grid = 512 workgroups (groups x wgs), block = 256, dynamic LDS = RES_LDS_BYTES (two workgroups per CU)
//
// LDS map: 9 halo'd tiles x {px,py,pa,cos,sin} (61 200 B); halo table (uint2 per halo cell, <= 720; it
// starts life as the u16 cell list); tile origins int2[9]; the 10-entry M^-1_O table; broadcast + reduction scratch;
// the staging area of the border z (9 x 80 float4, 11 520 B).
// Registers per lane: r(3) delta(3) Ap(3) M^-1_A M^-1_O flags per slot (12 x NS), plus the edge weights where they fit
// (4 per slot up to 7 slots, 2 at 8 and 9): 253 VGPRs at 7 slots, 249 at 9, no scratch.
// NS = tile slots the loops run over (1 .. RES_SLOTS): the most tiles any workgroup of the LAUNCH holds (the host picks
// the instantiation per launch, arapopt.hip: launch_resident).  The phases are fully unrolled and branch free over the
// slots, so their length is proportional to NS: a launch whose solves need 7 tiles per workgroup runs the 7-slot kernel.
template <bool STAMPS, int NS>
__global__ __launch_bounds__(RES_THREADS, 2) void k_pcg_resident(PlanDev pd, ResDev rd, int L)
{
    static_assert(NS >= 1 && NS <= RES_SLOTS, "slots");
    unsigned long long tA = 0, tS1 = 0, tB = 0, tS2 = 0, tU = 0, t0 = 0, t1 = 0;
    unsigned long long tm[4] = {0, 0, 0, 0}, tbs = 0, tzr = 0;     // STAMPS: inside the group sums (shader clocks; wave 0)
#define RES_STAMP(acc) do { if (STAMPS) { t1 = __builtin_amdgcn_s_memrealtime(); acc += t1 - t0; t0 = t1; } } while (0)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // Group of this workgroup: dealt by the host (ResWg).  Speed only (never correctness): workgroups are dealt
    // round-robin over the 8 XCDs, so the host gives a group of <= 64 workgroups a run of blockIdx values with
    // the same blockIdx & 7: it stays on that XCD (same-XCD fast path), and since the two workgroups that share a
    // CU have local indices j and j + 32, several small groups on one XCD overlap each other's waits.
    const ResWg me = rd.wgmap[blockIdx.x];
    if (me.slot < 0) return;                           // whole groups leave together
    const int rank = me.rank;
    // A previous launch (or the test hook) gave up: do nothing, the host redoes the step on the two-kernel path.
    if (rd.force_fail) {
        if (threadIdx.x == 0) atomicExch(rd.err, 0xDEADFFFFu);
        return;
    }
    if (__hip_atomic_load(rd.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const int b = me.slot;
    const int wgs = me.wgs;
    const int W = pd.W, H = pd.H;
    const size_t gb = (size_t)b * pd.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = lane & (RT_X - 1), ly = (wave << 1) | (lane >> 5);     // this lane's vertex inside a tile
    unsigned short* hlist = (unsigned short*)(lds + RES_TILES_PER_WG * LTILE);    // [RES_MAX_HALO]
    int2* tbase = (int2*)((char*)hlist + ((RES_MAX_HALO * 8 + 15) / 16) * 16);    // [9] tile origin (x0, y0)
    float* moLUT = (float*)(tbase + RES_TILES_PER_WG + 1);                // [10] (+2 pad)
    float* bcast = moLUT + 12;                                            // 2 x {sum, ok} + nhalo (int) + pad
    int* nhalo = (int*)(bcast + 4);
    double* wsum = (double*)(bcast + 6);                                  // 4 doubles (+ 1 at wsum[8])
    unsigned* nbits = (unsigned*)(wsum + 10);                             // [16] bitmap: ranks owning my halo vertices
    unsigned* nremote = nbits + 16;                                       // [1] some of them sit on another XCD
    float4* zst = (float4*)((char*)lds + RES_LDS_BYTES - RES_TILES_PER_WG * RES_ZX * 16);     // [9][RES_ZX] border z, staged: (z_x, z_y, z_alpha, z_y)
    unsigned long long* gran_group = rd.gran + me.gran;
    unsigned long long* granx_group = rd.gran + RES_GRAN_X + me.gran;

    const int nt = rd.ntiles[b];
    // this workgroup's run of the frame's active-tile list: nt tiles dealt evenly, the first nt % wgs ranks
    // take one more (the two workgroups of a CU run in lockstep, so an even deal shortens every phase)
    const int tbase_n = nt / wgs, textra = nt - tbase_n * wgs;
    const int tp = tbase_n + (rank < textra ? 1 : 0);                     // tiles of this workgroup (<= 9)
    const int tfirst = rank * tbase_n + (rank < textra ? rank : textra);
    const int* tl = rd.tilelist + (size_t)b * RES_MAX_TILES;
    float wr2, wf2;
    {
        const Slot sl = pd.slots[b];
        wr2 = sl.wr * sl.wr;
        wf2 = sl.wf * sl.wf;
        // M^-1 of the Offset components as k_gn_init computes it: D_O = sum over valid edges of
        // (wr*wr + wr*wr), plus wf*wf if the fit term is on; it depends on (degree, fit) only.
        if (tid < 10) {
            const int deg = tid % 5, fit = tid / 5;
            float dO = 0.f;
            for (int k = 0; k < deg; ++k) dO = dO + (sl.wr * sl.wr + sl.wr * sl.wr);
            if (fit) dO = fmaf(sl.wf, sl.wf, dO);
            moLUT[tid] = ginv(dO);
        }
        if (tid == 0) { *nhalo = 0; *nremote = 0u; }
        if (tid < 16) nbits[tid] = 0u;
    }

    float rx[NS], ry[NS], ra[NS];
    float dx_[NS], dy_[NS], da_[NS];
    float apx[NS], apy[NS], apa[NS];
    float ma_[NS], mo_[NS];
    unsigned fl[NS];
    int ibase[NS];                                     // SGPRs
    // phase A weights an edge by wr^2 or +0 (fit term: wf^2 or +0) according to the vertex's flag bits.  With registers
    // to spare (NS <= RES_WREG_SLOTS) the five weights per slot are formed once, here; otherwise from the flags in
    // every iteration (two bit operations each: 10 of the 51 VALU instructions of a vertex)
    constexpr int NWR = NS <= RES_WREG_SLOTS ? 4 : (NS == 8 ? RES_WREG_8 : RES_WREG_9);     // edge weights kept per slot
    constexpr bool WREG = NWR > 0;
    float we[WREG ? NS : 1][5];
    const int loff = lx + W * ly;                      // this lane's vertex inside a tile: index = ibase + loff

    // LDS tile t: float2 P2[340] (px,py) | float2 CS[340] (cos,sin) | float PA[340]; cell = row*34 + col
    const int cell = (ly + 1) * LROW + (lx + 1);
#define TP2(T) ((float2*)(T))
#define TCS(T) ((float2*)(T) + LPLANE)
#define TPA(T) ((T) + 4 * LPLANE)

    // ---- prologue: load state, p0 and cos/sin with halos ------------------------------------------
    // Two memory round trips for all slots together: the tile origins, then everything else.  A vertex's flags and data
    // are fetched side by side and the data dropped if the vertex turns out excluded (flags -> branch -> data, slot after
    // slot, was 2 x NS dependent round trips: 33 us of every launch at 7 slots).
    int org_[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) org_[j] = j < tp ? tl[tfirst + j] : -1;
    // No granule of this workgroup's tiles may carry a tag from an earlier launch: tag 0 everywhere (write-through,
    // whatever the placement); the stores travel while the loads below do, and are waited for before the first granule.
    unsigned long long* const zx_b = rd.zx + (size_t)b * RES_MAX_TILES * RES_ZG;    // this solve's published border z
    for (int c = tid; c < tp * RES_ZG; c += RES_THREADS)
        __hip_atomic_store(zx_b + (size_t)tfirst * RES_ZG + c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float2 qC[NS], qR[NS], vR[NS], vC[NS], hR[NS], hC[NS];
    float qRa[NS], qM[NS], vRa[NS], vM[NS], hRa[NS], hM[NS];
    unsigned qF[NS], vF[NS], hF[NS];                   // flags of the own vertex / of the halo vertices this lane fetches
    const int vrow = ly == 0 ? 0 : RT_Y + 1, hcol = lx == 0 ? 0 : RT_X + 1;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        qC[j] = qR[j] = vR[j] = vC[j] = hR[j] = hC[j] = make_float2(0.f, 0.f);
        qRa[j] = qM[j] = vRa[j] = vM[j] = hRa[j] = hM[j] = 0.f;
        qF[j] = vF[j] = hF[j] = 0u;
        const int org = org_[j];
        const int y0 = org / W, x0 = org - y0 * W;
        const int x = x0 + lx, y = y0 + ly;
        if (org >= 0 && x < W && y < H) {
            const size_t i = gb + (size_t)(x + W * y);
            qF[j] = pd.flags[i];
            // (p0 = M^-1 r, of the own vertex and of the halo vertices, is formed below exactly as k_gn_init forms it: the
            //  init kernel in front of a resident launch does not even store it)
            qC[j] = pd.cs[i];
            qR[j] = pd.rO[i]; qRa[j] = pd.rA[i]; qM[j] = pd.preA[i];
            // halo cells this thread is responsible for: above / below its column, left / right of its row
            if ((ly == 0 && y0 > 0) || (ly == RT_Y - 1 && y + 1 < H)) {
                const size_t hi = ly == 0 ? i - W : i + W;
                vF[j] = pd.flags[hi]; vR[j] = pd.rO[hi]; vRa[j] = pd.rA[hi]; vM[j] = pd.preA[hi]; vC[j] = pd.cs[hi];
            }
            if ((lx == 0 && x0 > 0) || (lx == RT_X - 1 && x + 1 < W)) {
                const size_t hi = lx == 0 ? i - 1 : i + 1;
                hF[j] = pd.flags[hi]; hR[j] = pd.rO[hi]; hRa[j] = pd.rA[hi]; hM[j] = pd.preA[hi]; hC[j] = pd.cs[hi];
            }
        }
    }
    // every cell of the halo'd tiles holds a finite value: cells outside the image or of unused slots are never
    // written below, and phase A multiplies (not selects) the contributions of invalid edges by zero
    for (int c = tid; c < NS * LTILE; c += RES_THREADS) lds[c] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        dx_[j] = dy_[j] = da_[j] = 0.f;
        apx[j] = apy[j] = apa[j] = 0.f;
        const int org = org_[j];
        int x0 = -1, y0 = -1;
        if (org >= 0) { y0 = org / W; x0 = org - y0 * W; }
        float* T = lds + j * LTILE;
        // Only ACTIVE vertices enter LDS; every other cell keeps the zero written above (k_gn_init leaves p0 of
        // excluded vertices untouched and their cos/sin come from whatever Angle the caller holds: phase A multiplies
        // such neighbours by a zero weight, so they must be finite).
        const bool act = (qF[j] & F_ACT) != 0u;
        rx[j] = act ? qR[j].x : 0.f; ry[j] = act ? qR[j].y : 0.f; ra[j] = act ? qRa[j] : 0.f;
        ma_[j] = act ? qM[j] : 0.f;
        mo_[j] = moLUT[__popc(qF[j] & 15u) + 5 * (int)((qF[j] >> 4) & 1u)];      // M^-1 of the Offset components
        if (act) {
            TP2(T)[cell] = make_float2(mo_[j] * rx[j], mo_[j] * ry[j]);           // p0 = M^-1 r (k_gn_init)
            TCS(T)[cell] = qC[j];
            TPA(T)[cell] = ma_[j] * ra[j];
        }
        if (vF[j] & F_ACT) {
            const int hc = vrow * LROW + (lx + 1);
            const float mo = moLUT[__popc(vF[j] & 15u) + 5 * (int)((vF[j] >> 4) & 1u)];
            TP2(T)[hc] = make_float2(mo * vR[j].x, mo * vR[j].y); TCS(T)[hc] = vC[j]; TPA(T)[hc] = vM[j] * vRa[j];
        }
        if (hF[j] & F_ACT) {
            const int hc = (ly + 1) * LROW + hcol;
            const float mo = moLUT[__popc(hF[j] & 15u) + 5 * (int)((hF[j] >> 4) & 1u)];
            TP2(T)[hc] = make_float2(mo * hR[j].x, mo * hR[j].y); TCS(T)[hc] = hC[j]; TPA(T)[hc] = hM[j] * hRa[j];
        }
        fl[j] = qF[j];
        ibase[j] = __builtin_amdgcn_readfirstlane(x0 + W * y0);        // uniform: vertex index of the tile origin
        if (tid == 0) tbase[j] = make_int2(x0, y0);
    }
    float rho = read_scalar(pd.red + ((size_t)b * pd.nslots + 0) * NSHARD);     // rho_0 from k_gn_init
    __syncthreads();
    // ---- halo list: every halo cell whose adjacent interior vertex (this lane's) has the matching edge bit
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const unsigned f = fl[j];
        if (WREG) {
            if (NWR > 0) we[j][0] = keep_if<0>(f, wr2);
            if (NWR > 1) we[j][1] = keep_if<1>(f, wr2);
            if (NWR > 2) we[j][2] = keep_if<2>(f, wr2);
            if (NWR > 3) we[j][3] = keep_if<3>(f, wr2);
            if (NS <= RES_WREG_FIT) we[j][4] = keep_if<4>(f, wf2);
        }
        // (one LDS atomic per wavefront and kind, not one per entry: 560 atomics on one address took 2 us of every launch)
        halo_append(ly == 0 && (f & F_E3), (unsigned short)(j * LPLANE + 0 * LROW + lx + 1), hlist, nhalo);
        halo_append(ly == RT_Y - 1 && (f & F_E2), (unsigned short)(j * LPLANE + (RT_Y + 1) * LROW + lx + 1), hlist, nhalo);
        halo_append(lx == 0 && (f & F_E1), (unsigned short)(j * LPLANE + (ly + 1) * LROW + 0), hlist, nhalo);
        halo_append(lx == RT_X - 1 && (f & F_E0), (unsigned short)(j * LPLANE + (ly + 1) * LROW + RT_X + 1), hlist, nhalo);
    }
    __syncthreads();
    const int nh = *nhalo;
    // this lane's share of the halo list, decoded once into an LDS table (the u16 list is dead after decoding and
    // registers are needed elsewhere): entry c = tid + u * 256 holds {global vertex index of the halo cell,
    // (byte offset in the float2 planes) / 8 | (byte offset in the float plane) / 4 << 16}
    uint2* htab = (uint2*)hlist;                                          // [RES_MAX_HALO]
    // (with fewer than nine slots in use there are registers to spare: the entries stay in VGPRs and the update phase
    //  starts its z loads without an LDS round trip)
    constexpr bool HREG = NS <= RES_HALO_REG_SLOTS;
    uint2 hreg[RES_HALO_PER_THREAD];
    {
        uint2 e[RES_HALO_PER_THREAD];
#pragma unroll
        for (int u = 0; u < RES_HALO_PER_THREAD; ++u) {
            const int c = tid + u * RES_THREADS;
            e[u] = make_uint2(0u, 0u);
            if (c < nh) {
                const int id = hlist[c];
                const int k = id / LPLANE, rem = id - k * LPLANE;
                const int row = rem / LROW, col = rem - row * LROW;
                const int2 tb = tbase[k];
                const int gx = tb.x + col - 1, gy = tb.y + row - 1;
                e[u].y = (unsigned)(k * (LTILE / 2) + rem) | ((unsigned)(k * LTILE + 4 * LPLANE + rem) << 16);
                // which workgroup of the group owns that vertex (the even deal above)?  -> neighbour bitmap
                // the tile that holds (gx, gy): band gy / 8, column (gx - first x of that band) / 32 (the halo vertex is
                // active, so it is not left of its band's first tile)
                const int band = gy / RT_Y;
                const int kx = (gx - rd.bandx0[(size_t)b * rd.rtY + band]) / RT_X;
                const int pos = rd.tilepos[((size_t)b * rd.rtY + band) * rd.rtX + (kx < 0 ? 0 : (kx < rd.rtX ? kx : rd.rtX - 1))];
                const int nfull = textra * (tbase_n + 1);
                const int owner = pos < nfull ? pos / (tbase_n + 1) : textra + (pos - nfull) / (tbase_n > 0 ? tbase_n : 1);
                if (pos >= 0 && owner != rank) atomicOr(&nbits[(owner >> 5) & 15], 1u << (owner & 31));
                // where the owner publishes that vertex's z: its tile's border entry (my top halo row is the neighbour's
                // bottom row, my left halo column its right column, ...)
                const int kxc = kx < 0 ? 0 : (kx < rd.rtX ? kx : rd.rtX - 1);
                const int lxo = gx - (rd.bandx0[(size_t)b * rd.rtY + band] + kxc * RT_X), lyo = gy - band * RT_Y;
                const int zpos = border_entry(lxo, lyo);
                (void)row; (void)col;
                e[u].x = (unsigned)((pos < 0 ? 0 : pos) * RES_ZG + 2 * zpos);
            }
        }
        __syncthreads();                             // every u16 entry has been read
#pragma unroll
        for (int u = 0; u < RES_HALO_PER_THREAD; ++u) {
            if (tid + u * RES_THREADS < RES_MAX_HALO) htab[tid + u * RES_THREADS] = e[u];     // 3 x 256 > 720 entries
            hreg[u] = e[u];
        }
        // (each thread reads back only what it wrote: no barrier needed)
    }
    static_assert(RES_TILES_PER_WG * LTILE < 65536 && LTILE % 2 == 0, "halo table packs 16-bit cell offsets");

    // byte offsets of this lane's cell and its four neighbours inside a tile's float2 planes / float plane
    unsigned offP[5], offA[5];
    {
        const int dn[5] = {0, 1, -1, LROW, -LROW};
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            offP[n] = (unsigned)(cell + dn[n]) * 8u;
            offA[n] = (unsigned)(cell + dn[n]) * 4u;
            asm volatile("" : "+v"(offP[n]));
            asm volatile("" : "+v"(offA[n]));
        }
    }
    const int zent = border_entry(lx, ly);             // this lane's entry in its tile's border export
    bool alive = true;
    float alpha_last = 0.f;                            // alpha of the last iteration (its delta update happens after the loop)
    // (the tag-0 stores of the prologue have landed before this workgroup's epoch-1 granule below, which every reader waits for)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- do all workgroups of this group sit on one XCD?  (speed only: selects the store flavour) --------
    // every workgroup publishes xcc + 65536 xcc^2 through the placement-independent protocol (epoch 1);
    // the ids are all equal iff  wgs * sum(xcc^2) == (sum xcc)^2.
    bool fast = false, zfast = false;                  // store flavour of the granules / of this workgroup's z
    bool hier = false, subfast = false;                // two-level sums for a group that spans XCDs (group_sum_h)
    bool hierx = false;                                // ... or the one-hop form for a group of few runs (group_sum_x)
    // second-level granules of a group that spans XCDs: it starts at an even bin (blockIdx & 7) - (rank >> 6)
    unsigned long long* const gran2 =
        rd.gran + RES_GRAN_L1 + ((((int)(blockIdx.x & 7u) - (me.rank >> 6)) >> 1) & 3) * RES_GRAN2_GROUP;
    {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;     // HW_REG_XCC_ID
        float dummy;
        const double mine = (double)xcc + 65536.0 * (double)(xcc * xcc);
        alive = group_sum(mine, 1u, gran_group, rank, wgs, bcast, rd.err, dummy, false, wsum + 8);
        const double tot = wsum[8];
        const double s2 = floor(tot / 65536.0), s1 = tot - 65536.0 * s2;
        fast = rd.allow_fast && ((double)wgs * s2 == s1 * s1);
        __syncthreads();
        // A group that spans XCDs.  (a) A workgroup's z is read only by the owners of its halo vertices (the relation is
        // symmetric), so if all of THOSE report this workgroup's XCD its z may stay in that L2 (plain stores).  (b) The
        // group's sums are gathered in two levels (group_sum_h); the first level uses plain granule stores if the 64
        // workgroups of this XCD run really share an XCD.  The epoch-1 granules still hold every workgroup's id (a
        // granule that a faster workgroup has already reused carries another tag and counts as "elsewhere").
        // (a group of two runs always sums in one hop: its second run may be a short piece, and the second-level granules
        //  of group_sum_h are addressed by whole aligned bins)
        hier = !fast && wgs > 64;
        hierx = hier && (((wgs + 63) >> 6) <= 2 || ((wgs + 63) >> 6) <= rd.flat_runs);
        if (!fast && rd.allow_fast && alive) {
            if (wave == 0) {
                const unsigned long long* buf = gran_group + (size_t)1 * wgs * RES_GS;     // parity of epoch 1
                bool remote = false, elsewhere = false;
                for (int m = lane; m < wgs; m += 64) {
                    const bool nb = ((nbits[(m >> 5) & 15] >> (m & 31)) & 1u) != 0u;
                    const bool same_run = hier && (m >> 6) == (rank >> 6);
                    if (nb || same_run) {
                        const unsigned long long lo = __hip_atomic_load(buf + RES_GS * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long hi = __hip_atomic_load(buf + RES_GS * m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const double val = __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
                        const double q2 = floor(val / 65536.0);
                        const bool other = (unsigned)(lo >> 32) != 1u || (unsigned)(hi >> 32) != 1u ||
                                           (unsigned)(val - 65536.0 * q2) != xcc;
                        remote = remote || (nb && other);
                        elsewhere = elsewhere || (same_run && other);
                    }
                }
                const unsigned any_remote = __any(remote) ? 1u : 0u, any_elsewhere = __any(elsewhere) ? 2u : 0u;
                if (lane == 0) *nremote = any_remote | any_elsewhere;      // (the votes need every lane: outside the if)
            }
            __syncthreads();
            zfast = (*nremote & 1u) == 0u;
            subfast = hier && (*nremote & 2u) == 0u;
        } else {
            zfast = fast;
        }
        __syncthreads();
    }
    // Wave priorities inside the phases.  The two workgroups of a CU belong to the same solve and run the same phase at
    // the same time; the SIMD's arbiter prefers the OLDER wave, so the CU's second workgroup got what the first left over
    // (phase A 1.08 vs 1.60 us) -- and the group waits for its slowest member.  Every wave runs the first slots of a phase
    // at high priority and the last ones at low priority: whoever is behind is in its high part while the other is in its
    // low part, and the pair finishes together (phase A 1.18 ... 1.47 us; 5.05 -> 4.84 us per iteration with the split
    // in the middle, round 2).  The split belongs near the END (round 3; high slots of NS, frames/s same box):
    //   8 frames, NS = 7:          4 / 5 / 6 / 7 of 7 -> 28.47 / 28.65 / 28.63 / 28.25
    //   24 segment solves, NS = 8: 4 / 6 / 7 / 8 of 8 -> 27.28 / 28.25 / 27.65 / 27.06   (CU mates of different solves)
    // i.e. the last TWO slots low.  (Wave 0 polls the group sums at priority 3, above both.)
#ifndef RES_PRIO_HI
#define RES_PRIO_HI 2
#endif
#ifndef RES_PRIO_TAIL
#define RES_PRIO_TAIL 2     // the last RES_PRIO_TAIL slots of a phase run at low priority, the others at high
#endif
#define RES_PRIO(J) { if ((J) + RES_PRIO_TAIL >= NS) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(RES_PRIO_HI); }
#define RES_PRIO_END() __builtin_amdgcn_s_setprio(0);
    if (STAMPS) t0 = __builtin_amdgcn_s_memrealtime();
    for (int l = 0; l < L && alive; ++l) {
        // ---------------- phase A: Ap = J^T J p, sigma = p.Ap --------------------------------------
        // Software pipelined over the tile slots: the 15 LDS reads of slot j+1 (own cell + 4 neighbours x
        // {(px,py), (cos,sin), pa}) are issued before the arithmetic of slot j, so that with only two wavefronts
        // per SIMD the LDS round trip hides behind ~70 VALU instructions instead of stalling every slot.
        static_assert(F_E0 == 1u && F_E1 == 2u && F_E2 == 4u && F_E3 == 8u && F_FIT == 16u && F_ACT == 32u, "bit numbers below");
        double acc = 0.0;
        float2 Lpv[2], Lcs[2], LqO[2][4], Lcn[2][4];
        float Lpa[2], LqA[2][4];
        // Every read has its own opaque base (cell, +1, -1, +row, -row; float2 and float planes) plus an immediate
        // offset: that keeps them single ds_read_b64 / ds_read_b32 (2 LDS cycles per wave instruction); merged into
        // ds_read2_b64 by the compiler they take 8 cycles per pair (MI355X_MICROARCH.md, LDS table).
        // The reads of slot j+1 go out in two batches (own cell + two neighbours before the arithmetic of slot j, the
        // other two neighbours in the middle of it): lgkmcnt is a 4-bit counter, and with all 15 reads outstanding the
        // wait for slot j's operands also waits for the first read of slot j+1.
#define RES_LOAD_A(J)                                                                                  \
        {                                                                                              \
            const int s_ = (J) & 1;                                                                    \
            const char* T_ = (const char*)lds + (J) * (LTILE * 4);                                     \
            Lpv[s_] = *(const float2*)(T_ + offP[0]);                                                  \
            Lcs[s_] = *(const float2*)(T_ + offP[0] + LPLANE * 8);                                     \
            Lpa[s_] = *(const float*)(T_ + offA[0] + LPLANE * 16);                                     \
            _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_) {                                         \
                LqO[s_][n_] = *(const float2*)(T_ + offP[n_ + 1]);                                     \
                Lcn[s_][n_] = *(const float2*)(T_ + offP[n_ + 1] + LPLANE * 8);                        \
                LqA[s_][n_] = *(const float*)(T_ + offA[n_ + 1] + LPLANE * 16);                        \
            }                                                                                          \
        }
#define RES_LOAD_B(J)                                                                                  \
        {                                                                                              \
            const int s_ = (J) & 1;                                                                    \
            const char* T_ = (const char*)lds + (J) * (LTILE * 4);                                     \
            _Pragma("unroll") for (int n_ = 2; n_ < 4; ++n_) {                                         \
                LqO[s_][n_] = *(const float2*)(T_ + offP[n_ + 1]);                                     \
                Lcn[s_][n_] = *(const float2*)(T_ + offP[n_ + 1] + LPLANE * 8);                        \
                LqA[s_][n_] = *(const float*)(T_ + offA[n_ + 1] + LPLANE * 16);                        \
            }                                                                                          \
        }
        RES_LOAD_A(0)
        RES_LOAD_B(0)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            RES_PRIO(j)
            unsigned f = fl[j];
            // keep the flag tests inside the loop: hoisted, their 54 lane masks spill out of the SGPR file and
            // come back as two v_readlane per test, more than the two bit operations that make a weight here
            asm volatile("" : "+v"(f));
            const int sj = j & 1;
            if (j + 1 < NS) RES_LOAD_A(j + 1)
            __builtin_amdgcn_sched_barrier(0);
            {
                // Branch free: every lane evaluates all four edges (LDS reads stay inside the halo'd tile, whose
                // never-written cells were zero-filled in the prologue, so every operand is finite) and an edge
                // whose flag bit is clear gets the weight +0: fma(0, u, a) = a exactly.
                const float2 pv = Lpv[sj];
                const float2 csv = Lcs[sj];                      // (ci, si)
                const float pa_ = Lpa[sj];
                const float ci = csv.x, si = csv.y;
                const float2 pa2 = make_float2(pa_, pa_);
                float2 axy = make_float2(0.f, 0.f);
                float aa = 0.f;
                // On the pixel grid d = U(c)-U(n) = -s, so q = R'(A(c))d and h = R'(A(n))d are signed copies of
                // (si,ci) / (sn,cn) (a product with -1/0/1 and the addition of a zero are exact), and each
                // block is the generic k_pcg_a expression
                //   t = fma(-q, pa, dP) ; a_xy = fma(wr2, fma(-h, qA, dP + t), a_xy) ; aa = fma(-wr2, fma(qx,tx,qy ty), aa)
                // on (x,y) pairs (v_pk_fma_f32), value for value (only the sign of an exact zero may differ).
#define RES_EDGE(BITNO, E, NQX, NQY, NHX, NHY, QX, QY)                                                 \
                {                                                                                      \
                    const float2 qO = LqO[sj][E], cn2 = Lcn[sj][E];                                    \
                    const float qA = LqA[sj][E];                                                       \
                    const float cn = cn2.x, sn = cn2.y;                                                \
                    const float w = (BITNO < NWR) ? we[WREG ? j : 0][BITNO] : keep_if<BITNO>(f, wr2);  \
                    const float2 e = pv - qO;                                                          \
                    const float2 t = fma2(make_float2(NQX, NQY), pa2, e);                              \
                    const float2 u = fma2(make_float2(NHX, NHY), make_float2(qA, qA), e + t);          \
                    axy = fma2(make_float2(w, w), u, axy);                                             \
                    aa = fmaf(-w, fmaf(QX, t.x, (QY) * t.y), aa);                                      \
                    (void)cn; (void)sn;                                                                \
                }
                //     bit no edge    -q          -h          q
                RES_EDGE(0, 0,     -si,  ci,   -sn,  cn,    si, -ci)      // s=( 1, 0): q=( si,-ci) h=( sn,-cn)
                RES_EDGE(1, 1,      si, -ci,    sn, -cn,   -si,  ci)      // s=(-1, 0): q=(-si, ci) h=(-sn, cn)
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < NS) RES_LOAD_B(j + 1)
                __builtin_amdgcn_sched_barrier(0);
                RES_EDGE(2, 2,     -ci, -si,   -cn, -sn,    ci,  si)      // s=( 0, 1): q=( ci, si) h=( cn, sn)
                RES_EDGE(3, 3,      ci,  si,    cn,  sn,   -ci, -si)      // s=( 0,-1): q=(-ci,-si) h=(-cn,-sn)
#undef RES_EDGE
                {
                    const float wf = (WREG && NS <= RES_WREG_FIT) ? we[WREG ? j : 0][4] : keep_if<4>(f, wf2);
                    axy = fma2(make_float2(wf, wf), pv, axy);
                }
                const float ax = axy.x, ay = axy.y;
                apx[j] = ax; apy[j] = ay; apa[j] = aa;
                acc += (double)dot3(pv.x, pv.y, pa_, ax, ay, aa);      // an excluded lane has p = 0 and Ap = 0: no mask needed
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef RES_LOAD_A
#undef RES_LOAD_B
        float sigma;
        RES_PRIO_END()
        RES_STAMP(tA);
        {
            const unsigned long long cb = STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
            const double bs = block_sum8(acc, wsum);
            if (STAMPS) tbs += __builtin_amdgcn_s_memtime() - cb;
            alive = hierx ? group_sum_x(block_sum_uniform(bs), 2u * l + 2u, gran_group, granx_group, rank, wgs, bcast, rd.err, sigma, subfast)
                  : hier  ? group_sum_h(block_sum_uniform(bs), 2u * l + 2u, gran_group, gran2, rank, wgs, bcast, rd.err, sigma, subfast)
                          : group_sum(bs, 2u * l + 2u, gran_group, rank, wgs, bcast, rd.err, sigma, fast, nullptr, rd.nowait != 0,
                                      STAMPS ? tm : nullptr);
        }
        if (!alive) break;
        RES_STAMP(tS1);
        // ---------------- phase B: alpha, r, z, rho', delta ---------------------------------------------
        // Branch free except for the stores: an excluded lane has r = Ap = 0 and M^-1_A = 0, so its r, z stay 0.
        float alpha = 0.f;
        if (sigma > 0.f) alpha = rho / sigma;
        acc = 0.0;
        const unsigned ztag = 2u * l + 3u;                 // (the epoch of the sum that follows: unique in the launch, never 0)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            RES_PRIO(j)
            unsigned f = fl[j];
            asm volatile("" : "+v"(f));
            const float mo = mo_[j], ma = ma_[j];
            rx[j] = fmaf(-alpha, apx[j], rx[j]);
            ry[j] = fmaf(-alpha, apy[j], ry[j]);
            ra[j] = fmaf(-alpha, apa[j], ra[j]);
            const float zx = mo * rx[j], zy = mo * ry[j], za = ma * ra[j];
            // Only a tile's border is ever read by another tile (rows 0 and 7, the two end columns): those lanes drop
            // their z into the LDS staging area; waves 1-3 send it out after the block sum below.  (Vector memory stores
            // cost their issue slot whatever the number of active lanes: from here they would be 21 nearly empty store
            // instructions per wave, from the staging area 9 full ones.)
            if (zent >= 0) {
                zst[j * RES_ZX + zent] = make_float4(zx, zy, za, zy);
            }
            acc += (double)keep_if<5>(f, dot3(zx, zy, za, rx[j], ry[j], ra[j]));
            __builtin_amdgcn_sched_barrier(0);
        }
        // (delta += alpha p is left to the update phase, which reads the workgroup's own p anyway)
        float rhoNew;
        RES_PRIO_END()
        RES_STAMP(tB);
        {
            const unsigned long long cb = STAMPS ? __builtin_amdgcn_s_memtime() : 0ull;
            const double bs = block_sum8(acc, wsum);
            if (STAMPS) tbs += __builtin_amdgcn_s_memtime() - cb;
            // (past the barrier of the block sum: the staged border z is complete.)  Waves 1-3 publish it, every
            // component a {tag, bits} granule, while wave 0 is busy with the group sum; nothing waits for these stores --
            // the reader checks the tags (update phase below), and they have a whole group sum to travel.
            if (wave != 0) {
                // granule g of the export = half g & 1 of staged entry g >> 1: (z_x, z_y) -> {z_x, low half of z_y},
                // (z_alpha, z_y) -> {z_alpha, high half of z_y}; a thread's granules all have its parity (the stride is even)
                unsigned long long* const out = zx_b + (size_t)tfirst * RES_ZG;
                const float2* const st2 = (const float2*)zst;
                const unsigned sh = (tid & 1) ? 16u : 0u;
                static_assert(((RES_THREADS - 64) & 1) == 0, "a thread keeps the parity of its granules");
                if (zfast) {
                    for (int g = tid - 64; g < tp * RES_ZG; g += RES_THREADS - 64) {
                        const float2 v = st2[g];
                        st_tagged(out + g, (ztag << 16) | ((__float_as_uint(v.y) >> sh) & 0xffffu), v.x, true);
                    }
                } else {
                    for (int g = tid - 64; g < tp * RES_ZG; g += RES_THREADS - 64) {
                        const float2 v = st2[g];
                        st_tagged(out + g, (ztag << 16) | ((__float_as_uint(v.y) >> sh) & 0xffffu), v.x, false);
                    }
                }
            }
            alive = hierx ? group_sum_x(block_sum_uniform(bs), 2u * l + 3u, gran_group, granx_group, rank, wgs, bcast, rd.err, rhoNew, subfast)
                  : hier  ? group_sum_h(block_sum_uniform(bs), 2u * l + 3u, gran_group, gran2, rank, wgs, bcast, rd.err, rhoNew, subfast)
                          : group_sum(bs, 2u * l + 3u, gran_group, rank, wgs, bcast, rd.err, rhoNew, fast, nullptr, rd.nowait != 0,
                                      STAMPS ? tm : nullptr);
        }
        if (!alive) break;
        RES_STAMP(tS2);
        float beta = 0.f;
        if (rho > 0.f) beta = rhoNew / rho;
        rho = rhoNew;
        if (l + 1 == L) { alpha_last = alpha; break; }
        // ---------------- p = z + beta p ---------------------------------------------------------------
        // (1) issue the loads of the neighbours' border z for this workgroup's halo cells (all in flight)
        unsigned long long hg[RES_HALO_PER_THREAD][2];
#pragma unroll
        for (int u = 0; u < RES_HALO_PER_THREAD; ++u) {
            hg[u][0] = hg[u][1] = (unsigned long long)ztag << 48;
            if (tid + u * RES_THREADS < nh) {
                // uniform base + 32-bit byte offset formed here (precomputed 64-bit addresses would spill)
                const unsigned gi = HREG ? hreg[u].x : htab[tid + u * RES_THREADS].x;
                const unsigned long long* q_ = (const unsigned long long*)((const char*)zx_b + (size_t)(gi * 8u));
#pragma unroll
                for (int c = 0; c < 2; ++c) hg[u][c] = __hip_atomic_load(q_ + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // (2) own cells while those loads fly: delta += alpha p, then p = z + beta p.  Branch free (an excluded lane
        //     computes 0 + beta * 0).  The registers of Ap are free here: every slot's own p is fetched from LDS up front
        //     (one slot ahead, each slot paid an LDS round trip).
        {
            float2 Up[NS];
            float Ua[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const char* T_ = (const char*)lds + j * (LTILE * 4);
                Up[j] = *(const float2*)(T_ + offP[0]);
                Ua[j] = *(const float*)(T_ + offA[0] + LPLANE * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                RES_PRIO(j)
                char* T_ = (char*)lds + j * (LTILE * 4);
                const float mo = mo_[j], ma = ma_[j];
                const float zx = mo * rx[j], zy = mo * ry[j], za = ma * ra[j];
                dx_[j] = fmaf(alpha, Up[j].x, dx_[j]);
                dy_[j] = fmaf(alpha, Up[j].y, dy_[j]);
                da_[j] = fmaf(alpha, Ua[j], da_[j]);
                *(float2*)(T_ + offP[0]) = make_float2(fmaf(beta, Up[j].x, zx), fmaf(beta, Up[j].y, zy));
                *(float*)(T_ + offA[0] + LPLANE * 16) = fmaf(beta, Ua[j], za);
            }
        }
        // (3) halo cells: p_halo = z_halo + beta p_halo (the owner computes the same expression).  A granule whose tag is
        //     this iteration's holds this iteration's value; the group sum in between took far longer than a store
        //     travels, so the tags all but always match, and a lane whose granule is still the previous iteration's
        //     reads it again.
        {
            bool fresh = true;
#pragma unroll
            for (int u = 0; u < RES_HALO_PER_THREAD; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) fresh = fresh && (unsigned)(hg[u][c] >> 48) == ztag;
            if (!__all(fresh) && !rd.nowait) {
                for (unsigned spins = 0; spins < RES_SPIN_LIMIT; ++spins) {
                    if (STAMPS) tzr += 1;
                    fresh = true;
#pragma unroll
                    for (int u = 0; u < RES_HALO_PER_THREAD; ++u) {
                        if (tid + u * RES_THREADS < nh) {
                            const unsigned gi = HREG ? hreg[u].x : htab[tid + u * RES_THREADS].x;
                            const unsigned long long* q_ = (const unsigned long long*)((const char*)zx_b + (size_t)(gi * 8u));
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                hg[u][c] = __hip_atomic_load(q_ + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                fresh = fresh && (unsigned)(hg[u][c] >> 48) == ztag;
                            }
                        }
                    }
                    if (__all(fresh)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                // gave up: the host sees the error word and redoes the step on the two-kernel path (the launch runs on
                // with whatever it read: its results are discarded)
                if (!fresh) atomicExch(rd.err, 0xDEAD0000u | (ztag & 0xffffu));
            }
        }
#pragma unroll
        for (int u = 0; u < RES_HALO_PER_THREAD; ++u) {
            const float2 hz2_ = make_float2(__uint_as_float((unsigned)hg[u][0]),
                                            __uint_as_float(((unsigned)(hg[u][0] >> 32) & 0xffffu) | ((unsigned)(hg[u][1] >> 32) << 16)));
            const float hz1_ = __uint_as_float((unsigned)hg[u][1]);
            if (tid + u * RES_THREADS < nh) {
                const unsigned pk = HREG ? hreg[u].y : htab[tid + u * RES_THREADS].y;
                float2* P = (float2*)((char*)lds + (pk & 0xffffu) * 8u);
                float* A = (float*)((char*)lds + (pk >> 16) * 4u);
                const float2 po = *P;
                *P = make_float2(fmaf(beta, po.x, hz2_.x), fmaf(beta, po.y, hz2_.y));
                *A = fmaf(beta, *A, hz1_);
            }
        }
        RES_PRIO_END()
        __syncthreads();
        RES_STAMP(tU);
    }
    if (STAMPS && tid == 0) {
        unsigned long long* o = rd.stamps + (size_t)blockIdx.x * 16;
        o[8] = tbs; o[9] = tm[0]; o[10] = tm[1]; o[11] = tm[2]; o[12] = tm[3]; o[13] = tzr;     // block sums, publish, poll, tail (clocks); sweeps; repeated looks at the z tags (wave 0)
        o[0] = tA; o[1] = tS1; o[2] = tB; o[3] = tS2; o[4] = tU; o[5] = (unsigned long long)tp; { unsigned pc = 0; for (int q = 0; q < 16; ++q) pc += __popc(nbits[q]); o[6] = (unsigned long long)nh | ((unsigned long long)pc << 32) | ((unsigned long long)*nremote << 48); } o[7] = (fast ? 1ull : 0ull) | (zfast ? 2ull : 0ull) | (hier ? 4ull : 0ull) | (subfast ? 8ull : 0ull);
    }
    if (!alive) return;
    // ---- epilogue: the last iteration's delta += alpha p; then the step itself -- PCGLinearUpdate (:552-557) and the cos/sin
    // of the new Angle, what k_gn_update and the next step's k_gn_prep would do -- for the frame solver (rd.fuse_update: a
    // failed launch there is followed by a redo of the whole schedule from the reset, never by a step on stale unknowns),
    // or delta back to the plan images for k_gn_update (drop-in plans: a launch that gave up must leave X untouched) ----
    if (L > 0) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const char* T_ = (const char*)lds + j * (LTILE * 4);
            const float2 p2 = *(const float2*)(T_ + offP[0]);
            const float pa1 = *(const float*)(T_ + offA[0] + LPLANE * 16);
            dx_[j] = fmaf(alpha_last, p2.x, dx_[j]);
            dy_[j] = fmaf(alpha_last, p2.y, dy_[j]);
            da_[j] = fmaf(alpha_last, pa1, da_[j]);
        }
    }
    int lo = loff;
    asm volatile("" : "+v"(lo));       // (else the nine store addresses are formed before the loop and spilled)
    if (rd.fuse_update) {
        const Slot sl = pd.slots[b];
        float2 o_[NS];
        float a_[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            o_[j] = make_float2(0.f, 0.f); a_[j] = 0.f;
            if (fl[j] & F_ACT) { const int i = ibase[j] + lo; o_[j] = sl.O[i]; a_[j] = sl.A[i]; }
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (fl[j] & F_ACT) {
                const int i = ibase[j] + lo;
                float2 o = o_[j];
                o.x = o.x + dx_[j];
                o.y = o.y + dy_[j];
                const float a = a_[j] + da_[j];
                sl.O[i] = o;
                sl.A[i] = a;
                pd.cs[gb + i] = sincos_spec(a);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const unsigned f = fl[j];
        if (f & F_ACT) {
            const int i = ibase[j] + lo;
            pd.deltaO[gb + i] = make_float2(dx_[j], dy_[j]);
            pd.deltaA[gb + i] = da_[j];
        }
    }
}

#undef RES_STAMP
#undef RES_PRIO
#undef RES_PRIO_END
#undef TP2
#undef TCS
#undef TPA

}  // namespace arap
