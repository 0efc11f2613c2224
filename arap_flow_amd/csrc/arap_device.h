// arap_device.h -- per-vertex device math of the ARAP Gauss-Newton / PCG solve for gfx950.
//
// The energy is the one of the reference's arap_plan.t:1-23; the three generated functions of the
// reference (evalJTF o.t:2129-2172, applyJTJ o.t:2029-2089, cost o.t:2375-2385) are written here in
// closed form (derivation: DESIGN.md "The math").  Everything is float32 (precision.t:1-6) except
// the reductions, which accumulate float32 per-vertex terms in float64 (DESIGN.md "Reductions").
//
// This translation unit is compiled with -ffp-contract=off: each operator below is one IEEE-754
// operation, in the order written, and fused multiply-adds appear only where fmaf()/fma() is written
// (v_fma_f32 / v_pk_fma_f32: one rounding), so a CPU implementation with the same operation list
// (oracle/, gcc -mfma -ffp-contract=off) reproduces the results bit for bit (parity tier T3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace arap {

// ---- per-vertex flag byte, rebuilt at every Gauss-Newton step from Mask and Constraints --------
// bit s (0..3): edge to neighbour s of the stencil {(1,0),(-1,0),(0,1),(0,-1)} (arap_plan.t:14) is
//               valid: neighbour in bounds, Mask == 0 at both ends (arap_plan.t:17)
// bit 4       : fitting residual valid: Constraints.x >= 0 and Constraints.y >= 0 (arap_plan.t:22)
// bit 5       : vertex active, i.e. not excluded: Mask == 0 (arap_plan.t:11)
enum : unsigned { F_E0 = 1u, F_E1 = 2u, F_E2 = 4u, F_E3 = 8u, F_FIT = 16u, F_ACT = 32u };

constexpr int TILE_X = 64;   // one wavefront = 64 consecutive x of one row: 256-B coalesced rows
constexpr int TILE_Y = 4;    // 4 wavefronts per workgroup
constexpr int RED_TICK_STRIDE = 64;   // u32 between two groups' arrival counters
constexpr int NSHARD = 32;   // reduction groups (shards) per scalar: each shard is the order-fixed sum of one group of workgroups

// One frame ("slot") of a batch: the five problem images of arap_plan.t:2-6 and the two weights.
// Lives in device memory; the kernels index it with blockIdx.z.
struct Slot {
    float2* O;        // Offset      in/out  [N]
    float* A;         // Angle       in/out  [N]
    const float2* U;  // UrShape             [N]
    const float2* C;  // Constraints         [N]
    const float* M;   // Mask                [N]
    float wf, wr;     // w_fitSqrt, w_regSqrt
    int pad_[2];
};

// Plan-owned solver state (the GN subset of makePlan, solverGPUGaussNewton.t:1254-1284), one
// contiguous block per image with stride N per slot.  3-vectors are split like the unknowns:
// an Offset-shaped float2 image and an Angle-shaped float image.
struct PlanDev {
    int W, H, N;
    int tilesX, tilesY;
    int nslots;              // reduction slots per frame in `red`
    Slot* slots;             // [batch]
    float2 *deltaO, *rO, *zO, *pO0, *pO1, *ApO, *preO, *cs;
    float *deltaA, *rA, *zA, *pA0, *pA1, *ApA, *preA;
    uint8_t* flags;          // [batch][N]
    uint8_t* tileact;        // [batch][tilesX*tilesY]  1 = tile holds an active vertex
    double* red;             // [batch][nslots][NSHARD]  PCG scalars of the current GN step
    double* costred;         // [batch][ncost][NSHARD]   cost after Init (index 0) and after each step
    int ncost;
    // "LMGPU" solver kind only (arap_lm.h); lm == 0 for Gauss-Newton plans
    int lm;
    float2 *bO, *CtCO, *SSqO, *AdO;
    float *bA, *CtCA, *SSqA, *AdA;
    double* lmred;           // [lIterations + 2][NSHARD]: q after iteration l at l+1 (Q0 at 0), model cost last
    // error word of the resident kernel (arap_resident.h), NULL when the plan has none: once it is set the step's
    // update is skipped so that the host can redo the step on the two-kernel path from unchanged unknowns
    const unsigned* res_err;
    // resident path: k_gn_prep zeroes what the step's later kernels accumulate into or poll, instead of memset
    // nodes: reduction slot 0 of every frame (rho_0, written by k_gn_init) and the granules of all launches
    unsigned long long* res_gran;
    int res_gran_n;          // u64 entries to zero; 0 = the two-kernel path (host memsets all slots of `red`)
    // list launches of the per-step kernels (arap_kernels.h: vidx): every frame's active 64x4 tiles; NULL = whole grid
    const int* t64list;      // [batch][tilesX * tilesY]
    const int* t64n;         // [batch]
    // order-fixed reductions of the kernel-per-phase paths (block_reduce_fixed below)
    unsigned long long* part; // [batch][maxblk][4] every workgroup's partial(s) of the running launch, as tagged granules
    unsigned* tick;           // [batch][NSHARD][RED_TICK_STRIDE] per reduction group: arrivals of the running launch, one
                              // counter per 256 bytes (memory-side atomics on one line serialise: ~12 ns each)
    unsigned* gen;            // [batch][NSHARD] launches the group has seen (read-mostly: stays in L2)
    int maxblk;               // workgroups per frame any launch of this plan may have
};

// ---- cos/sin: same operation list as oracle/arap_oracle.c:arap_sincos_spec ----------------------
// The reference calls libdevice __nv_cosf/__nv_sinf (util.t:160-174).  Here: Cody-Waite reduction
// with the fdlibm split of pi/2 and Taylor polynomials, all in IEEE double +,-,*,rint, rounded to
// float once.  Evaluated once per vertex per Gauss-Newton step, never inside the PCG loop.
__device__ __forceinline__ float2 sincos_spec(float af)
{
    const double a = (double)af;
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    const double k = rint(a * two_over_pi);
    const double r = fma(-k, pio2_lo, fma(-k, pio2_hi, a));
    const double r2 = r * r;
    double ps = -1.0 / 1307674368000.0;
    ps = fma(ps, r2, 1.0 / 6227020800.0);
    ps = fma(ps, r2, -1.0 / 39916800.0);
    ps = fma(ps, r2, 1.0 / 362880.0);
    ps = fma(ps, r2, -1.0 / 5040.0);
    ps = fma(ps, r2, 1.0 / 120.0);
    ps = fma(ps, r2, -1.0 / 6.0);
    const double sr = fma(r, r2 * ps, r);
    double pc = 1.0 / 20922789888000.0;
    pc = fma(pc, r2, -1.0 / 87178291200.0);
    pc = fma(pc, r2, 1.0 / 479001600.0);
    pc = fma(pc, r2, -1.0 / 3628800.0);
    pc = fma(pc, r2, 1.0 / 40320.0);
    pc = fma(pc, r2, -1.0 / 720.0);
    pc = fma(pc, r2, 1.0 / 24.0);
    pc = fma(pc, r2, -0.5);
    const double cr = fma(r2, pc, 1.0);
    const int q = (int)((long long)k & 3);
    double c, s;
    if (q == 0) { c = cr; s = sr; }
    else if (q == 1) { c = -sr; s = cr; }
    else if (q == 2) { c = -cr; s = -sr; }
    else { c = sr; s = -cr; }
    return make_float2((float)c, (float)s);
}

// guardedInvert, CERES variant (solverGPUGaussNewton.t:323-332)
__device__ __forceinline__ float ginv(float d)
{
    // sqrtf and '/' are correctly rounded under hipcc's default
    // -fhip-fp32-correctly-rounded-divide-sqrt (the __fsqrt_rn intrinsic is NOT: it is the native
    // approximate v_sqrt_f32)
    const float t = 1.0f + sqrtf(d);
    return 1.0f / (t * t);
}

__device__ __forceinline__ float dot3(float ax, float ay, float aa, float bx, float by, float ba)
{
    return fmaf(aa, ba, fmaf(ay, by, ax * bx));
}

// neighbour index offset of stencil entry s
__device__ __forceinline__ int noff(int s, int W) { return s == 0 ? 1 : (s == 1 ? -1 : (s == 2 ? W : -W)); }

// ---- reductions -------------------------------------------------------------------------------
// wave64 DPP tree on doubles, then one value per wavefront combined through LDS, then the order-fixed
// two-level sum over the launch's workgroups (block_reduce_fixed below).
// Sum of a double over the 64 lanes with DPP row shifts / broadcasts (the order LLVM's wave scan uses on gfx9:
// row_shr 1, 2, 4, 8 inside each row of 16, then row_bcast:15 into rows 1 and 3, then row_bcast:31 into rows 2-3).
// The total is valid in LANE 63.  A __shfl tree costs six dependent ds_bpermute round trips (~0.3 us) per sum; the
// resident kernel has four sums on the critical path of every iteration, the two-kernel path one at the head
// (read_scalar) and one at the tail of every kernel.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);             // lanes without a source add +0.0
}
__device__ __forceinline__ double wave_sum_l63(double v)
{
    v = dpp_add_f64<0x111, 0xf>(v);                  // row_shr:1
    v = dpp_add_f64<0x112, 0xf>(v);                  // row_shr:2
    v = dpp_add_f64<0x114, 0xf>(v);                  // row_shr:4
    v = dpp_add_f64<0x118, 0xf>(v);                  // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add_f64<0x142, 0xa>(v);                  // row_bcast:15 into rows 1 and 3
    v = dpp_add_f64<0x143, 0xc>(v);                  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}

// the same sum, returned in every lane (v_readlane of lane 63)
__device__ __forceinline__ double wave_sum(double v)
{
    v = wave_sum_l63(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// ---- order-fixed reduction of one (or two) doubles over all workgroups of a launch ------------------------------
// The reference adds per-warp partials with red.global.add.f32 in arrival order (solverGPUGaussNewton.t:312-317): its
// sums differ from run to run.  Here a launch's sum does not depend on the order in which workgroups finish:
//   1. inside a workgroup: the fixed DPP tree per wavefront, then the wavefronts' sums added in wave order;
//   2. the workgroups of a frame form NSHARD groups (linear index mod NSHARD).  Every workgroup drops its partial into
//      its own slot of `part` and takes a ticket from its group's arrival counter; the group's LAST arriver reads all
//      the group's partials, adds them in index order (lane k: members k, k + 64, ...; then the DPP tree) and STORES the
//      result into shard g of the target scalar -- no value is ever accumulated by an atomic;
//   3. the consumer (read_scalar, the next kernel) adds the NSHARD shards with the same fixed tree.
// Hand-off inside the launch: a partial travels as two data-tagged granules {tag << 32 | 32 value bits}, each one 8-byte
// write-through store, tag = number of launches this group has seen + 1; nothing orders the stores before the ticket --
// the last arriver checks the tags and looks again at a granule that has not landed yet (bounded; a timeout writes NaN,
// which no caller can overlook).  Every workgroup of the launch must call this exactly once per frame it belongs to --
// also those with nothing to add (v = 0) -- with lb = its linear index among the frame's nlb workgroups.
__device__ __forceinline__ unsigned red_tag(const PlanDev& pd, int b, unsigned lb)
{
    // (may be read any time before the workgroup's own arrival: only the group's last arriver changes it)
    return __hip_atomic_load(pd.gen + (size_t)b * NSHARD + (lb % NSHARD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
}

// The sum in two calls, so that a streaming kernel can put its own stores BETWEEN them: red_arrive (block-level sum,
// partial out, ticket taken -- nothing is waited for) ... the kernel's stores ... red_finish (the ticket's value is
// needed only here).  vmcnt completes in order: a ticket taken AFTER a wave's streaming stores comes back only when those
// have drained, and the wave -- with it the workgroup's slot on the CU -- lives that long (measured at 1920x1080, mask == 0,
// phase B: 31 us with the ticket in front of the stores, 61 us behind them).
struct RedTicket {
    unsigned ticket;       // the group's arrival count before this workgroup's (valid in wave 0, lane 0; not yet waited for)
    unsigned tag, g, ng;
};

template <int NV>
__device__ __forceinline__ RedTicket red_arrive(const PlanDev& pd, int b, unsigned lb, unsigned nlb, double v0, double v1,
                                                unsigned tag = 0u)
{
    __shared__ double wsumF[2][16];
    const unsigned lin = threadIdx.y * blockDim.x + threadIdx.x, wave = lin >> 6, lane = lin & 63;
    const unsigned nw = (blockDim.x * blockDim.y + 63u) >> 6;
    RedTicket rt;
    rt.g = lb % NSHARD;
    rt.ng = (nlb - rt.g + NSHARD - 1) / NSHARD;                // members of this group: g, g + NSHARD, ...
    rt.ticket = 0u;
    if (tag == 0u && wave == 0) tag = red_tag(pd, b, lb);
    rt.tag = tag;
    v0 = wave_sum(v0);
    if (NV > 1) v1 = wave_sum(v1);
    if (lane == 0) { wsumF[0][wave] = v0; if (NV > 1) wsumF[1][wave] = v1; }
    __syncthreads();
    if (wave != 0) return rt;
    double t0 = 0.0, t1 = 0.0;
    for (unsigned w = 0; w < nw; ++w) { t0 += wsumF[0][w]; if (NV > 1) t1 += wsumF[1][w]; }
    unsigned long long* const mine = pd.part + ((size_t)b * pd.maxblk + lb) * 4;
    if (lane < 2u * NV) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(lane < 2 ? t0 : t1);
        const unsigned hw = (lane & 1u) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(mine + lane, ((unsigned long long)tag << 32) | hw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0)
        rt.ticket = __hip_atomic_fetch_add(pd.tick + ((size_t)b * NSHARD + rt.g) * RED_TICK_STRIDE, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
    return rt;
}

template <int NV>
__device__ __forceinline__ void red_finish(const PlanDev& pd, int b, const RedTicket& rt, double* tgt0, double* tgt1)
{
    const unsigned lin = threadIdx.y * blockDim.x + threadIdx.x, wave = lin >> 6, lane = lin & 63;
    if (wave != 0) return;
    const unsigned ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)rt.ticket);
    const unsigned g = rt.g, ng = rt.ng, tag = rt.tag;
    if (ticket + 1u != ng) return;
    // ---- the group's last arriver ----
    unsigned* tk = pd.tick + ((size_t)b * NSHARD + g) * RED_TICK_STRIDE;
    const unsigned long long* const base = pd.part + (size_t)b * pd.maxblk * 4;
    double a0 = 0.0, a1 = 0.0;
    bool ok = false;
    for (unsigned spins = 0; spins < (1u << 16) && !ok; ++spins) {
        a0 = a1 = 0.0;
        bool mine = true;
        for (unsigned m = lane; m < ng; m += 64) {
            const unsigned long long* q = base + (size_t)(g + NSHARD * m) * 4;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const unsigned long long lo = __hip_atomic_load(q + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(q + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mine = mine && (unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag;
                const double val = __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
                if (k == 0) a0 += val; else a1 += val;
            }
        }
        ok = __all(mine);
        if (!ok) __builtin_amdgcn_s_sleep(2);
    }
    a0 = wave_sum_l63(a0);
    if (NV > 1) a1 = wave_sum_l63(a1);
    if (lane == 63) {
        const double bad = __longlong_as_double(0x7ff8000000000000ll);
        tgt0[g] = ok ? a0 : bad;
        if (NV > 1) tgt1[g] = ok ? a1 : bad;
    }
    if (lane == 0) {           // ready for the next launch (seen there after the kernel boundary)
        __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pd.gen + (size_t)b * NSHARD + g, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NV>
__device__ __forceinline__ void block_reduce_fixed(const PlanDev& pd, int b, unsigned lb, unsigned nlb, double v0,
                                                   double v1, double* tgt0, double* tgt1, unsigned tag = 0u)
{
    const RedTicket rt = red_arrive<NV>(pd, b, lb, nlb, v0, v1, tag);
    red_finish<NV>(pd, b, rt, tgt0, tgt1);
}

// sum of the NSHARD shards of one scalar, by every wavefront for itself (lanes >= NSHARD add 0)
__device__ __forceinline__ float read_scalar(const double* shards)
{
    const unsigned lane = __lane_id();          // true lane: block shapes narrower than 64 put several rows in a wave
    double v = lane < (unsigned)NSHARD
                   ? __hip_atomic_load(shards + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                   : 0.0;
    return (float)wave_sum(v);
}

}  // namespace arap
