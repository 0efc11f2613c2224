// arapopt.hip -- libarapopt.so: host side of the MI355X-native ARAP solver + the C ABI of
// include/arap_opt.h.  Built for gfx950 only (see arap_flow_amd/build.py).
//
// Host-side structure mirrors the reference's plan lifecycle:
//   Opt_NewState/ProblemDefine/ProblemPlan   createwrapper.t:124-220, o.t:2521-2558
//   init / step / cost / setSolverParameter  solverGPUGaussNewton.t:956-1007, 1016-1177, 1179-1221
// but one Gauss-Newton step is a single hipGraph launch (2 + 2*lIterations + 1 kernel nodes and one
// memset node) with every PCG scalar kept in device memory: no per-iteration memset/memcpy calls,
// no host round trip inside a solve (the reference issues ~4 tiny API calls per PCG iteration,
// solverGPUGaussNewton.t:1058-1091, and a blocking read-back per step, :790-797).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/arap_opt.h"
#include "arap_kernels.h"
#include "arap_resident.h"
#include "arap_lm.h"
#include "arap_tiled.h"
#include "arap_stream.h"
#include "arap_warp.h"

using namespace arap;

#define ARAPOPT_VERSION "arapopt 0.1.0 gfx950"

// Device API failure -> message + exit, as the reference does (solverGPUGaussNewton.t:59-73).
static void hip_fatal(hipError_t e, const char* what, const char* file, int line)
{
    fprintf(stderr, "arapopt: HIP error %d (%s) in %s at %s:%d\n", (int)e, hipGetErrorString(e), what, file, line);
    exit((int)e ? (int)e : 1);
}
#define HC(call)                                                     \
    do {                                                             \
        hipError_t e_ = (call);                                      \
        if (e_ != hipSuccess) hip_fatal(e_, #call, __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
struct KernelTimer {            // collectPerKernelTimingInfo (Opt.h:23-25, util.t:414-511)
    struct Rec { std::string name; hipEvent_t a, b; };
    std::vector<Rec> recs;
    void clear()
    {
        for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        recs.clear();
    }
    void report()
    {
        std::map<std::string, std::pair<int, double>> agg;
        for (auto& r : recs) {
            float ms = 0.f;
            (void)hipEventSynchronize(r.b);
            (void)hipEventElapsedTime(&ms, r.a, r.b);
            agg[r.name].first++;
            agg[r.name].second += ms;
        }
        printf("--------------------------------------------------------\n");
        printf("        Kernel        |   Count  |   Total   | Average \n");
        printf("----------------------+----------+-----------+----------\n");
        for (auto& kv : agg)
            printf(" %-20s |   %4d   | %8.3fms| %7.4fms\n", kv.first.c_str(), kv.second.first,
                   kv.second.second, kv.second.second / kv.second.first);
        printf("--------------------------------------------------------\n");
        clear();
    }
};

struct Opt_State {
    int verbosity = 0;
    int timing = 0;
    int device = 0;
    hipStream_t stream = nullptr;     // where all work is enqueued (NULL = null stream)
    hipStream_t cap = nullptr;        // private stream used only for graph capture
    hipStream_t own_stream = nullptr; // ArapFlow_UseOwnStream
    hipEvent_t t0 = nullptr, t1 = nullptr;
    KernelTimer ktimer;
    bool use_graph = true;
    bool use_resident = true;   // ArapFlow_SetResident
    bool resident_failed = false;   // a resident launch has timed out at least once (GPU shared with another process?)
    // After a timeout the resident path pauses for `res_cooldown` solve calls (ArapFlow_SolverSolve / Opt_ProblemInit),
    // then it is tried again; every further timeout doubles the pause (8, 16, ... 1024), a checked success resets it.
    int res_cooldown = 0, res_backoff = 8;
    int tile = -1;              // ArapFlow_SetTile: phase-A variant of the two-kernel path; -1 = choose per solve
    bool force_b8 = false;      // ARAPOPT_B8=1 (counter calibration): the 8-byte-per-lane form of phase B
    int stream_a = 0;           // ARAPOPT_STREAM_A=1 (experiments): the tiled k_pcg_a_grid instead of the marching kernel
};

struct Opt_Problem {
    int kind;    // 0 = gaussNewtonGPU, 1 = LMGPU
};

// solver parameter table and defaults: solverGPUGaussNewton.t:26-39, :148-163
struct SolverParameters {
    int residual_reset_period = 10;
    float min_relative_decrease = 1e-3f, min_trust_region_radius = 1e-32f, max_trust_region_radius = 1e16f,
          q_tolerance = 1e-4f, function_tolerance = 1e-6f, trust_region_radius = 1e4f,
          radius_decrease_factor = 2.0f, min_lm_diagonal = 1e-6f, max_lm_diagonal = 1e32f;
    int nIterations = 10, lIterations = 10;
    int nIter = 0;
};

struct Opt_Plan {
    Opt_State* st = nullptr;
    int W = 0, H = 0, N = 0, batch = 1;
    PlanDev pd{};
    SolverParameters sp;
    std::vector<Slot> hslots;       // host mirror of pd.slots
    std::vector<Slot> uploaded;
    void* block = nullptr;          // all plan-owned images
    int lcap = 0;                   // lIterations capacity of pd.red
    int ccap = 0;                   // cost entries capacity of pd.costred
    int nb = 1;                     // slots active in the current solve (grid.z)
    // graph of one GN step, keyed by (lIterations, nb)
    hipGraphExec_t gexec = nullptr;
    hipGraph_t graph = nullptr;
    int g_l = -1, g_nb = -1, g_res = -1;   // g_res > 0: resident launches per step, < 0: two-kernel phase-A variant
    // resident PCG (arap_resident.h): only for the frame solver (pixel-grid UrShape, host-known masks)
    bool res_capable = false;       // device has 256 CUs and the kernel fits one workgroup per CU
    bool res_frames = false;        // plan is driven by ArapFlow_Solver (and the resident resources exist)
    bool res_frames_any = false;    // plan is driven by ArapFlow_Solver, whatever the device
    bool grid_u = false;            // UrShape is the pixel grid on every active vertex (frame solver: always; drop-in:
                                    // what the last analysis found): the streaming phase A without UrShape loads applies
    // ArapFlow_Solver reports one cost, the one after the last step of the last ramp iteration: the costs the
    // reference evaluates at Init and after every step (for its log) are skipped unless cost_wanted
    bool lazy_cost = false, cost_wanted = true;
    ResDev rd{};
    void* res_block = nullptr;
    std::vector<int> h_ntiles;
    std::vector<std::vector<int>> h_tiles;   // what rd.tilelist holds per slot (skip the upload when nothing changed)
    std::vector<uint8_t> h_tiles_valid;
    std::vector<std::vector<int>> h_tilepos, h_bandx0;
    uint8_t* d_resact = nullptr;    // [rtX * rtY] drop-in analysis: 32x8 tiles (fixed grid) that hold an active vertex
    // frame solver: every slot's active 64x4 tiles, for the list launches of k_gn_prep / k_gn_init / k_gn_update
    int* d_t64list = nullptr;       // [batch][tilesX * tilesY]
    int* d_t64n = nullptr;          // [batch]
    std::vector<std::vector<int>> h_t64;
    std::vector<int> h_t64n;
    int g_maxn = -1;                // list length the captured graph was built for
    int g_steps = 0;                // Gauss-Newton steps in the captured graph
    unsigned long long g_ns = 0;    // ... and its resident launches' slot counts (hashed)
    int res_tiles_all = 0;          // 32x8 tiles of the whole grid (share of active tiles: plan_active_tiles_majority)
    bool hole_pending = false;      // test hook ARAPOPT_FORCE_RES_FAIL=2: the next table upload leaves one workgroup out
    ResWg* d_wgmap = nullptr;       // [batch][RES_WGS]: one table per resident launch of a GN step
    ResWg* pin_wgmap = nullptr;     // pinned staging of the same size
    std::vector<ResWg> h_wgmap;     // what d_wgmap holds
    int res_sets = 0;               // resident launches per GN step
    std::vector<int> res_ns;        // per launch: tile slots in use = most tiles any of its workgroups holds
    int res_inflight = 0;           // solves of the fullest launch (diagnostic)
    unsigned res_launches = 0;
    // drop-in (Opt_*) plans: result of the Init-time analysis (k_analyse) of the caller's Mask / UrShape
    bool opt_res_ok = false;
    bool prep_done = false;         // the last enqueued step left flags and cos/sin ready for the next (resident launch with fuse_update)
    Slot opt_res_slot{};
    int* d_notgrid = nullptr;
    // "LMGPU" plans
    int kind = 0;
    void* lm_block = nullptr;       // b, CtC, SSq, Adelta, prevX
    float2* prevO = nullptr;
    float* prevA = nullptr;
    int lm_lcap = 0;
    float lm_radius = 0.f, lm_decrease = 0.f;      // pd.parameters.trust_region_radius / radius_decrease_factor
    double lm_prev_cost = 0.0;
    bool lm_done = false;

    dim3 grid() const { return dim3(pd.tilesX, pd.tilesY, nb); }
    dim3 blk() const { return dim3(TILE_X, TILE_Y, 1); }
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static Opt_Plan* plan_create(Opt_State* st, int W, int H, int batch)
{
    HC(hipSetDevice(st->device));
    Opt_Plan* p = new Opt_Plan();
    p->st = st;
    p->W = W; p->H = H; p->N = W * H; p->batch = batch;
    PlanDev& pd = p->pd;
    pd.W = W; pd.H = H; pd.N = p->N;
    pd.tilesX = (W + TILE_X - 1) / TILE_X;
    pd.tilesY = (H + TILE_Y - 1) / TILE_Y;
    const size_t BN = (size_t)batch * p->N;
    // 8 float2 images + 7 float images + flags + tileact + slots, zero initialised (o.t:627-632)
    const size_t sz2 = align_up(BN * sizeof(float2), 256), sz1 = align_up(BN * sizeof(float), 256);
    const size_t szf = align_up(BN, 256), szt = align_up((size_t)batch * pd.tilesX * pd.tilesY, 256);
    const size_t szs = align_up(sizeof(Slot) * batch, 256);
    // order-fixed reductions (arap_device.h: block_reduce_fixed): a slot per workgroup of the largest launch of any
    // kernel of this plan -- every tile shape is at least 16 wide and 4 high -- and the groups' tickets
    pd.maxblk = ((W + 15) / 16) * ((H + 3) / 4) + 16;
    const size_t szp = align_up((size_t)batch * pd.maxblk * 4 * sizeof(unsigned long long), 256);
    const size_t szk = align_up((size_t)batch * NSHARD * RED_TICK_STRIDE * sizeof(unsigned), 256);
    const size_t szg = align_up((size_t)batch * NSHARD * sizeof(unsigned), 256);
    const size_t total = 8 * sz2 + 7 * sz1 + szf + szt + szs + szp + szk + szg;
    HC(hipMalloc(&p->block, total));
    HC(hipMemsetAsync(p->block, 0, total, st->stream));
    char* c = (char*)p->block;
    auto take = [&](size_t s) { char* r = c; c += s; return r; };
    pd.deltaO = (float2*)take(sz2); pd.rO = (float2*)take(sz2); pd.zO = (float2*)take(sz2);
    pd.pO0 = (float2*)take(sz2); pd.pO1 = (float2*)take(sz2); pd.ApO = (float2*)take(sz2);
    pd.preO = (float2*)take(sz2); pd.cs = (float2*)take(sz2);
    pd.deltaA = (float*)take(sz1); pd.rA = (float*)take(sz1); pd.zA = (float*)take(sz1);
    pd.pA0 = (float*)take(sz1); pd.pA1 = (float*)take(sz1); pd.ApA = (float*)take(sz1);
    pd.preA = (float*)take(sz1);
    pd.flags = (uint8_t*)take(szf);
    pd.tileact = (uint8_t*)take(szt);
    pd.slots = (Slot*)take(szs);
    pd.part = (unsigned long long*)take(szp);
    pd.tick = (unsigned*)take(szk);
    pd.gen = (unsigned*)take(szg);
    pd.red = nullptr; pd.costred = nullptr; pd.nslots = 0; pd.ncost = 0;
    p->hslots.assign(batch, Slot{});
    p->h_ntiles.assign(batch, 0);
    p->h_tiles.assign(batch, std::vector<int>());
    p->h_tiles_valid.assign(batch, 0);
    p->h_tilepos.assign(batch, std::vector<int>());
    p->h_bandx0.assign(batch, std::vector<int>());
    p->h_t64.assign(batch, std::vector<int>());
    p->h_t64n.assign(batch, 0);
    return p;
}

// The resident kernel is instantiated per number of tile slots its loops run over (arap_resident.h): a launch takes
// the instantiation for the most tiles any of its workgroups holds.
typedef void (*ResidentKernel)(PlanDev, ResDev, int);
template <bool STAMPS, int... NS>
static const void* resident_kernel_of(int ns, std::integer_sequence<int, NS...>)
{
    static const ResidentKernel table[] = {k_pcg_resident<STAMPS, NS + 1>...};
    return (const void*)table[ns - 1];
}
static const void* resident_kernel(bool stamps, int ns)
{
    if (ns < 1) ns = 1;
    if (ns > RES_SLOTS) ns = RES_SLOTS;
    return stamps ? resident_kernel_of<true>(ns, std::make_integer_sequence<int, RES_SLOTS>())
                  : resident_kernel_of<false>(ns, std::make_integer_sequence<int, RES_SLOTS>());
}

// resident-path resources: active-tile lists, granules, error word
static void plan_enable_resident(Opt_Plan* p)
{
    Opt_State* st = p->st;
    const char* nr = getenv("ARAPOPT_NO_RESIDENT");
    if (nr && nr[0] == '1') return;
    hipDeviceProp_t prop;
    HC(hipGetDeviceProperties(&prop, st->device));
    if (prop.multiProcessorCount * 2 < RES_WGS) return;        // two resident workgroups per CU
    for (int ns = 1; ns <= RES_SLOTS; ++ns) {
        if (hipFuncSetAttribute(resident_kernel(false, ns), hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_BYTES) !=
            hipSuccess) { (void)hipGetLastError(); return; }
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, resident_kernel(false, ns), RES_THREADS, RES_LDS_BYTES) !=
                hipSuccess || occ < 2) {
            (void)hipGetLastError();
            return;
        }
    }
    const size_t sz_tl = align_up((size_t)p->batch * RES_MAX_TILES * sizeof(int), 256);
    const size_t sz_nt = align_up((size_t)p->batch * sizeof(int), 256);
    const size_t sz_gr = align_up((size_t)p->batch * RES_GRAN_PER_LAUNCH * 8, 256);   // one block per launch of a step
    const size_t sz_map = align_up((size_t)p->batch * RES_WGS * sizeof(ResWg), 256);
    p->rd.rtX = (p->W + RT_X - 1) / RT_X;
    p->rd.rtY = (p->H + RT_Y - 1) / RT_Y;
    const size_t sz_tp = align_up((size_t)p->batch * p->rd.rtX * p->rd.rtY * sizeof(int), 256);
    const size_t sz_bx = align_up((size_t)p->batch * p->rd.rtY * sizeof(int), 256);
    const size_t sz_ra = align_up((size_t)p->rd.rtX * p->rd.rtY, 256);               // drop-in analysis: tile activity
    {
        // border z of every tile (arap_resident.h: ResDev::zx)
        const size_t nz = (size_t)p->batch * RES_MAX_TILES * RES_ZG * sizeof(unsigned long long);
        HC(hipMalloc((void**)&p->rd.zx, nz));
        HC(hipMemsetAsync(p->rd.zx, 0, nz, st->stream));
    }
    HC(hipHostMalloc((void**)&p->pin_wgmap, sz_map, hipHostMallocDefault));
    HC(hipMalloc(&p->res_block, sz_gr + sz_tl + sz_nt + 256 + sz_map + sz_tp + sz_bx + sz_ra));
    HC(hipMemsetAsync(p->res_block, 0, sz_gr + sz_tl + sz_nt + 256 + sz_map + sz_tp + sz_bx + sz_ra, st->stream));
    char* c = (char*)p->res_block;
    p->rd.gran = (unsigned long long*)c; c += sz_gr;
    p->rd.tilelist = (const int*)c; c += sz_tl;
    p->rd.ntiles = (const int*)c; c += sz_nt;
    p->rd.err = (unsigned*)c; c += 256;
    p->d_wgmap = (ResWg*)c; c += sz_map;
    p->rd.tilepos = (const int*)c; c += sz_tp;
    p->rd.bandx0 = (const int*)c; c += sz_bx;
    p->d_resact = (uint8_t*)c;
    p->pd.res_err = p->rd.err;
    p->rd.stamps = nullptr;
    {
        const char* ff = getenv("ARAPOPT_FORCE_RES_FAIL");      // test hook
        p->rd.force_fail = (ff && ff[0] == '1') ? 1 : 0;
        // '2': ONE real timeout -- the first table upload of this plan leaves a workgroup of the first group out, so the
        // rest of that group spins in its first group wait until the bounded spin gives up (arap_resident.h: group_sum)
        p->hole_pending = ff && ff[0] == '2';
    }
    {
        const char* nf = getenv("ARAPOPT_NO_XCD_FAST");
        p->rd.allow_fast = (nf && nf[0] == '1') ? 0 : 1;
    }
    {
        const char* fr = getenv("ARAPOPT_FLAT_RUNS");           // experiments: 0 = always two levels, 4 / 8 = wider one-hop sums
        p->rd.flat_runs = fr ? atoi(fr) : RES_FLAT_MAX_RUNS;
    }
    {
        const char* nw = getenv("ARAPOPT_RES_NOWAIT");          // diagnostic: iteration time without the group waits
        p->rd.nowait = (nw && nw[0] == '1') ? 1 : 0;
    }
    const char* sd = getenv("ARAPOPT_STAMPS");      // diagnostic build of the resident kernel (tools/res_stamps.py)
    if (sd && sd[0] == '1') {
        for (int ns = 1; ns <= RES_SLOTS; ++ns)
            HC(hipFuncSetAttribute(resident_kernel(true, ns), hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_BYTES));
        HC(hipMalloc(&p->rd.stamps, RES_WGS * 16 * sizeof(unsigned long long)));
        HC(hipMemset(p->rd.stamps, 0, RES_WGS * 16 * sizeof(unsigned long long)));
    }
    p->res_capable = true;
}

// The resident kernel's work list of one solve (arap_resident.h): 32 x 8 tiles in bands of 8 rows; within a band the
// tiles start at `bandx0` (the band's first active x when `aligned`, else 0) and follow each other every 32 columns;
// tiles without an active vertex are left out.  `tiles` receives the origins (x0 + W y0) band by band.
static void build_resident_tiles(const uint8_t* mask_red, int W, int H, bool aligned, std::vector<int>& tiles,
                                 std::vector<int>& bandx0, uint64_t* nactive)
{
    const int rtY = (H + RT_Y - 1) / RT_Y;
    tiles.clear();
    bandx0.assign(rtY, 0);
    std::vector<uint8_t> col(W);
    uint64_t na = 0;
    for (int band = 0; band < rtY; ++band) {
        std::fill(col.begin(), col.end(), 0);
        const int y0 = band * RT_Y, y1 = std::min(H, y0 + RT_Y);
        for (int y = y0; y < y1; ++y) {
            const uint8_t* row = mask_red + (size_t)W * y;
            for (int x = 0; x < W; ++x) {
                const uint8_t a = row[x] == 0;
                col[x] |= a;
                na += a;
            }
        }
        int xmin = 0, xmax = -1;
        for (int x = 0; x < W; ++x)
            if (col[x]) { if (xmax < 0) xmin = x; xmax = x; }
        if (xmax < 0) continue;
        const int xs = aligned ? xmin : 0;
        bandx0[band] = xs;
        for (int x0 = xs; x0 <= xmax; x0 += RT_X) {
            bool any = false;
            for (int x = x0; x < W && x < x0 + RT_X && !any; ++x) any = col[x] != 0;
            if (any) tiles.push_back(x0 + W * y0);
        }
    }
    if (nactive) *nactive = na;
}

// Upload one slot's work list: origins, their count, the bands' first x and the inverse map (band, column) -> position.
// Enqueued on `cs` (the caller orders it before the kernels that read the lists and after those that still use the
// old ones).  The sources are plan-owned host vectors that live until the next upload of the slot.
static void plan_upload_tiles(Opt_Plan* p, int slot, const std::vector<int>& tiles, const std::vector<int>& bandx0,
                              hipStream_t cs)
{
    const int nt = (int)tiles.size();
    p->h_ntiles[slot] = nt;
    if (!p->res_capable) return;
    if (p->h_tiles_valid[slot] && p->h_tiles[slot] == tiles && p->h_bandx0[slot] == bandx0) return;   // already there
    p->h_tiles[slot] = tiles;
    p->h_bandx0[slot] = bandx0;
    p->h_tiles_valid[slot] = 1;
    const int rtX = p->rd.rtX, rtY = p->rd.rtY;
    if (nt <= RES_MAX_TILES) {
        std::vector<int>& pos = p->h_tilepos[slot];
        pos.assign((size_t)rtX * rtY, -1);
        for (int i = 0; i < nt; ++i) {
            const int y0 = tiles[i] / p->W, x0 = tiles[i] - y0 * p->W;
            const int band = y0 / RT_Y, k = (x0 - bandx0[band]) / RT_X;
            pos[(size_t)band * rtX + k] = i;
        }
        if (nt > 0)
            HC(hipMemcpyAsync((void*)(p->rd.tilelist + (size_t)slot * RES_MAX_TILES), p->h_tiles[slot].data(),
                              sizeof(int) * nt, hipMemcpyHostToDevice, cs));
        HC(hipMemcpyAsync((void*)(p->rd.tilepos + (size_t)slot * rtX * rtY), pos.data(), sizeof(int) * pos.size(),
                          hipMemcpyHostToDevice, cs));
        HC(hipMemcpyAsync((void*)(p->rd.bandx0 + (size_t)slot * rtY), p->h_bandx0[slot].data(), sizeof(int) * rtY,
                          hipMemcpyHostToDevice, cs));
    }
    HC(hipMemcpyAsync((void*)(p->rd.ntiles + slot), &p->h_ntiles[slot], sizeof(int), hipMemcpyHostToDevice, cs));
}

// Opt_ProblemInit and every Opt_ProblemStep of a drop-in plan: look at the caller's Mask and UrShape (one small
// kernel + a read-back of one byte per tile; the reference's init and step block on a device read-back too,
// solverGPUGaussNewton.t:1006,1117,790-797) to decide whether the step can take the resident kernel and with which
// active-tile list.  The reference re-reads every parameter at every Step (:960,1026) and lets the caller change
// them in between (Opt.h:58-66): so does this -- new Mask / UrShape contents or swapped buffers are seen here.
static void plan_analyse_for_resident(Opt_Plan* p)
{
    p->opt_res_ok = false;
    if (p->res_frames) return;
    Opt_State* st = p->st;
    // without the resident resources there is still one use of the result: the share of active tiles steers the
    // automatic phase-A variant of the two-kernel path
    const int rtX = (p->W + RT_X - 1) / RT_X, rtY = (p->H + RT_Y - 1) / RT_Y;
    const int nt_all = rtX * rtY;
    if (!p->d_notgrid) HC(hipMalloc(&p->d_notgrid, sizeof(int) + (size_t)nt_all));
    uint8_t* d_act = p->d_resact ? p->d_resact : (uint8_t*)(p->d_notgrid + 1);
    HC(hipMemsetAsync(p->d_notgrid, 0, sizeof(int), st->stream));
    hipLaunchKernelGGL(k_analyse, dim3(rtX, rtY, 1), dim3(RT_X, RT_Y, 1), 0, st->stream, p->pd, d_act, p->d_notgrid);
    std::vector<uint8_t> act(nt_all);
    int notgrid = 1;
    HC(hipMemcpyAsync(act.data(), d_act, nt_all, hipMemcpyDeviceToHost, st->stream));
    HC(hipMemcpyAsync(&notgrid, p->d_notgrid, sizeof(int), hipMemcpyDeviceToHost, st->stream));
    HC(hipStreamSynchronize(st->stream));
    std::vector<int> tiles, bandx0(rtY, 0);                     // fixed grid: every band starts at x = 0
    for (int t = 0; t < nt_all; ++t)
        if (act[t]) tiles.push_back((t % rtX) * RT_X + p->W * ((t / rtX) * RT_Y));
    const int nt = (int)tiles.size();
    p->h_ntiles[0] = nt;
    p->res_tiles_all = nt_all;
    p->grid_u = notgrid == 0;
    if (notgrid || !p->res_capable || !st->use_resident || st->res_cooldown > 0 || nt > RES_MAX_TILES) return;
    plan_upload_tiles(p, 0, tiles, bandx0, st->stream);
    p->opt_res_ok = true;
    p->opt_res_slot = p->hslots[0];
}

static bool plan_resident_eligible(const Opt_Plan* p)
{
    if (!p->res_capable || !p->st->use_resident) return false;
    if (p->st->res_cooldown > 0) return false;          // pausing after a timed-out launch (plan_resident_failed)
    if (p->sp.lIterations > RES_MAX_L) return false;    // (the border-z granules carry 16-bit iteration tags)
    if (!p->res_frames) {
        // drop-in plan: only with the images analysed just before this step (plan_analyse_for_resident)
        const Slot& a = p->opt_res_slot;
        const Slot& c = p->hslots[0];
        if (!p->opt_res_ok || a.M != c.M || a.U != c.U) return false;
    }
    for (int b = 0; b < p->nb; ++b)
        if (p->h_ntiles[b] > RES_MAX_TILES) return false;
    return true;
}

// Deal the solves of the current batch to resident launches and their 512 workgroups (ResWg tables).
// A launch is 8 bins of 64 workgroups: the workgroups that land on one XCD (blockIdx & 7 equal, local index
// blockIdx >> 3).  A solve of nt active tiles needs ceil(nt / 9) workgroups.
//  * NARROW (needs <= 64): shares a bin with others: the narrow solves are bin-packed into as few launches as
//    first-fit-decreasing needs, spread evenly over the shared bins (least-loaded first), and every group is then
//    widened to use its bin's spare workgroups (fewer tiles per workgroup = shorter phases).  Groups of one bin take
//    consecutive local indices, so the two workgroups of a CU (j, j + 32) usually serve different solves.
//  * MEDIUM (65 .. 128: the 1920x1080 --multseg segments, ~716 tiles = 80 workgroups): a whole bin as HOME for ranks
//    0 .. 63 plus a PIECE of need - 64 workgroups (ranks 64 ..) in a bin it shares with other pieces and narrow solves,
//    widened like those.  Six such solves fit a launch (six homes, two shared bins) where whole pairs of bins held four.
//    Tiles are dealt in list (row-major) order, so nearly all of a workgroup's halo neighbours share its XCD; the
//    group's sums are gathered in one hop (arap_resident.h: group_sum_x, runs of 64 ranks).
//  * WIDE (> 128): 4 or 8 whole bins, aligned to the width; every bin holds a run of 64 consecutive ranks; sums in two
//    levels (group_sum_h).
// Placement is for speed only: the kernel checks at run time which runs really share an XCD.
// Returns the number of launches; fills `map` ([launches][RES_WGS]) and `inflight_out` when given.
static int resident_deal(const int* ntiles, int nb, std::vector<ResWg>* map_out, int* inflight_out)
{
    const int XW = RES_WGS / 8;                                  // workgroups per XCD
    enum { NARROW = 0, MEDIUM = 1, WIDE = 2 };
    std::vector<int> need(nb), width(nb), kind(nb);
    int mx = 1;
    for (int b = 0; b < nb; ++b) {
        need[b] = (ntiles[b] + RES_TILES_PER_WG - 1) / RES_TILES_PER_WG;
        if (need[b] < 1) need[b] = 1;
        mx = need[b] > mx ? need[b] : mx;
        kind[b] = need[b] <= XW ? NARROW : (need[b] <= 2 * XW ? MEDIUM : WIDE);
        width[b] = 1;                                            // whole bins a WIDE solve takes: 4 or 8
        if (kind[b] == WIDE) { width[b] = 4; while (width[b] * XW < need[b]) width[b] *= 2; }
    }
    std::vector<ResWg> map;
    int nsets = 0, inflight = 0;
    const ResWg idle = {-1, 0, 0, 0};
    int forced = 0;
    {
        const char* fg = getenv("ARAPOPT_RES_GROUPS");           // experiments only: equal groups
        if (fg && atoi(fg) > 0) forced = atoi(fg);
    }
    if (forced && forced <= RES_MAX_GROUPS && (RES_WGS / forced) >= mx && (RES_WGS % forced) == 0) {
        const int groups = forced, wgs = RES_WGS / groups;
        nsets = (nb + groups - 1) / groups;
        map.assign((size_t)nsets * RES_WGS, idle);
        for (int set = 0; set < nsets; ++set)
            for (int i = 0; i < RES_WGS; ++i) {
                int g, rank;
                const int x = i & 7, j = i >> 3;
                if (groups >= 8) { g = x + 8 * (j / wgs); rank = j % wgs; }
                else { const int xper = 8 / groups; g = x / xper; rank = (x % xper) * XW + j; }
                const int sb = set * groups + g;
                if (sb < nb) map[(size_t)set * RES_WGS + i] = ResWg{sb, rank, wgs, 2 * RES_GS * g * wgs};
            }
        inflight = nb < groups ? nb : groups;
    } else {
        std::vector<int> order(nb);
        for (int b = 0; b < nb; ++b) order[b] = b;
        std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
            return kind[a] != kind[c] ? kind[a] > kind[c] : (width[a] != width[c] ? width[a] > width[c] : need[a] > need[c]);
        });
        // bin state over all launches.  owner >= 0: a wide solve, or a medium solve's home, holds the whole bin;
        // owner == -1: shared / free (load = workgroups spoken for)
        std::vector<int> owner, load;
        auto add_launch = [&]() { owner.insert(owner.end(), 8, -1); load.insert(load.end(), 8, 0); };
        // (1) wide solves: first launch with `width` aligned bins that nothing has touched yet
        for (int b : order) {
            if (kind[b] != WIDE) continue;
            size_t at = owner.size();
            for (size_t k = 0; k + width[b] <= owner.size() && at == owner.size(); k += width[b]) {
                bool free_run = true;
                for (int q = 0; q < width[b]; ++q) free_run = free_run && owner[k + q] < 0 && load[k + q] == 0;
                if (free_run) at = k;
            }
            if (at == owner.size()) add_launch();                // 8 is a multiple of every width: `at` is aligned
            for (int q = 0; q < width[b]; ++q) { owner[at + q] = b; load[at + q] = XW; }
        }
        // (2) medium solves: a free bin as home, the piece first-fit into a bin of the same launch that pieces already
        //     share (so that free bins stay available as homes), else into a free one
        std::vector<int> home(nb, -1);                           // medium: its home bin (global index)
        std::vector<int> ffbin(nb, -1);                          // first-fit bin of every shared item (piece or narrow solve)
        auto item_size = [&](int b) { return kind[b] == MEDIUM ? need[b] - XW : need[b]; };
        for (int b : order) {
            if (kind[b] != MEDIUM) continue;
            const int piece = item_size(b);
            int hb = -1, pb = -1;
            for (size_t L = 0; L * 8 < owner.size() && hb < 0; ++L) {
                int h = -1, pshared = -1, pfree = -1;
                for (int x = 0; x < 8; ++x) {
                    const size_t k = L * 8 + x;
                    if (owner[k] >= 0) continue;
                    if (load[k] == 0) { if (h < 0) h = (int)k; else if (pfree < 0) pfree = (int)k; }
                    else if (pshared < 0 && load[k] + piece <= XW) pshared = (int)k;
                }
                const int pk = pshared >= 0 ? pshared : pfree;
                if (h >= 0 && pk >= 0) { hb = h; pb = pk; }
            }
            if (hb < 0) { hb = (int)owner.size(); pb = hb + 1; add_launch(); }
            owner[hb] = b; load[hb] = XW; home[b] = hb;
            load[pb] += piece; ffbin[b] = pb;
        }
        // (3) narrow solves: number of launches by first fit decreasing over the shared bins ...
        for (int b : order) {
            if (kind[b] != NARROW) continue;
            size_t k = 0;
            while (k < owner.size() && (owner[k] >= 0 || load[k] + need[b] > XW)) ++k;
            if (k == owner.size()) add_launch();
            load[k] += need[b]; ffbin[b] = (int)k;
        }
        nsets = (int)owner.size() / 8;
        // ... then spread: pieces over the shared bins of their launch, narrow solves over all shared bins, least-loaded
        // bin that still fits first; keep the first-fit deal if that ever fails
        std::vector<int> bin_of(nb, -1), l2(owner.size(), 0);
        bool ok = true;
        for (int pass = 0; pass < 2 && ok; ++pass)
            for (int b : order) {
                if (kind[b] == WIDE || (pass == 0) != (kind[b] == MEDIUM)) continue;
                const int sz = item_size(b);
                const size_t lo = kind[b] == MEDIUM ? (size_t)(home[b] / 8) * 8 : 0;
                const size_t hi = kind[b] == MEDIUM ? lo + 8 : owner.size();
                int best = -1;
                for (size_t k = lo; k < hi; ++k)
                    if (owner[k] < 0 && l2[k] + sz <= XW && (best < 0 || l2[k] < l2[best])) best = (int)k;
                if (best < 0) { ok = false; break; }
                l2[best] += sz;
                bin_of[b] = best;
            }
        if (!ok) {
            bin_of = ffbin;
            std::fill(l2.begin(), l2.end(), 0);
            for (int b = 0; b < nb; ++b)
                if (kind[b] != WIDE) l2[bin_of[b]] += item_size(b);
        }
        // workgroups of every group: a shared item gets its bin's spare workgroups in proportion (>= its need)
        std::vector<int> wgs_of(nb, 0), part_w(nb, 0);
        for (int b = 0; b < nb; ++b) {
            if (kind[b] == WIDE) { wgs_of[b] = width[b] * XW; continue; }
            part_w[b] = XW * item_size(b) / l2[bin_of[b]];       // >= the item's size; a bin's parts sum to <= 64
            wgs_of[b] = kind[b] == MEDIUM ? XW + part_w[b] : part_w[b];
        }
        map.assign((size_t)nsets * RES_WGS, idle);
        std::vector<int> gran_of(nb, -1);
        for (int set = 0; set < nsets; ++set) {
            // granule space per group, in units of workgroups: a group of several runs addresses its runs in blocks of 64
            int ordinal = 0, count = 0;
            auto take_gran = [&](int b) {
                if (gran_of[b] >= 0) return;
                gran_of[b] = 2 * RES_GS * ordinal;
                ordinal += wgs_of[b] > XW ? ((wgs_of[b] + XW - 1) / XW) * XW : wgs_of[b];
                ++count;
            };
            for (int x = 0; x < 8; ++x) {
                const size_t k = (size_t)set * 8 + x;
                if (owner[k] >= 0) {
                    const int b = owner[k];
                    take_gran(b);
                    if (kind[b] == MEDIUM) {                     // home: ranks 0 .. 63
                        for (int j = 0; j < XW; ++j)
                            map[(size_t)set * RES_WGS + (size_t)j * 8 + x] = ResWg{b, j, wgs_of[b], gran_of[b]};
                        continue;
                    }
                    const bool first = x == 0 || owner[k - 1] != b;
                    if (!first) continue;                        // dealt with its first bin
                    for (int q = 0; q < width[b]; ++q)
                        for (int j = 0; j < XW; ++j)
                            map[(size_t)set * RES_WGS + (size_t)j * 8 + x + q] = ResWg{b, q * XW + j, wgs_of[b], gran_of[b]};
                    continue;
                }
                int j = 0;
                for (int b : order) {                            // the bin's items, largest first
                    if (kind[b] == WIDE || bin_of[b] != (int)k) continue;
                    take_gran(b);
                    const int r0 = kind[b] == MEDIUM ? XW : 0;
                    for (int r = 0; r < part_w[b]; ++r, ++j)
                        map[(size_t)set * RES_WGS + (size_t)j * 8 + x] = ResWg{b, r0 + r, wgs_of[b], gran_of[b]};
                }
            }
            inflight = count > inflight ? count : inflight;
        }
    }
    if (map_out) map_out->swap(map);
    if (inflight_out) *inflight_out = inflight;
    return nsets;
}

// Deal the current batch; true if the tables changed (the caller re-uploads them and drops the captured graph).
static bool plan_resident_pack(Opt_Plan* p)
{
    std::vector<ResWg> map;
    const int nsets = resident_deal(p->h_ntiles.data(), p->nb, &map, &p->res_inflight);
    std::vector<int> ns(nsets, 1);
    for (size_t i = 0; i < map.size(); ++i)
        if (map[i].slot >= 0) {
            const int t = (p->h_ntiles[map[i].slot] + map[i].wgs - 1) / map[i].wgs;      // tiles of the group's fullest workgroup
            int& m = ns[i / RES_WGS];
            m = t > m ? t : m;
        }
    if (const char* fn = getenv("ARAPOPT_RES_NS"))               // experiments: run at least this many tile slots
        for (int& m : ns) m = std::max(m, std::min(atoi(fn), (int)RES_SLOTS));
    const bool same = nsets == p->res_sets && ns == p->res_ns && map.size() == p->h_wgmap.size() &&
                      memcmp(map.data(), p->h_wgmap.data(), map.size() * sizeof(ResWg)) == 0;
    if (same) return false;
    p->h_wgmap.swap(map);
    p->res_ns.swap(ns);
    p->res_sets = nsets;
    return true;
}

// Did a resident launch of this plan give up (a bounded group wait timed out: its 512 workgroups were not all
// resident, e.g. because another process uses the GPU)?  Then the step's update was skipped on the device
// (k_gn_update), the error word is cleared, the resident path is switched off for this state and the caller redoes
// the work on the two-kernel path.  Requires a synchronised stream.
static bool plan_resident_failed(Opt_Plan* p)
{
    if (!p->res_capable || p->res_launches == 0) return false;
    unsigned e = 0;
    HC(hipMemcpyAsync(&e, p->rd.err, sizeof(e), hipMemcpyDeviceToHost, p->st->stream));
    HC(hipStreamSynchronize(p->st->stream));
    if (e == 0) return false;
    Opt_State* st = p->st;
    fprintf(stderr, "arapopt: resident PCG kernel gave up at a group wait (code 0x%08x); is the GPU shared? "
                    "Falling back to the two-kernel path for the next %d solve calls.\n", e, st->res_backoff);
    st->resident_failed = true;
    st->res_cooldown = st->res_backoff;
    st->res_backoff = st->res_backoff >= 1024 ? 1024 : 2 * st->res_backoff;
    HC(hipMemsetAsync((void*)p->rd.err, 0, sizeof(unsigned), st->stream));
    HC(hipStreamSynchronize(st->stream));
    p->h_wgmap.clear();                     // (the test hook's table with a hole must not survive: re-deal next time)
    p->res_sets = 0;
    return true;
}

static void plan_check_resident_error(Opt_Plan* p)
{
    if (plan_resident_failed(p)) {        // reached only if a caller consumed results without the checks below
        fprintf(stderr, "arapopt: resident PCG failure detected after results were consumed\n");
        exit(3);
    }
}

static void plan_drop_graph(Opt_Plan* p)
{
    if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
    if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
    p->g_l = p->g_nb = p->g_res = -1;
}

static void plan_free(Opt_Plan* p)
{
    if (!p) return;
    HC(hipStreamSynchronize(p->st->stream));
    plan_drop_graph(p);
    if (p->pd.red) (void)hipFree(p->pd.red);
    if (p->pd.costred) (void)hipFree(p->pd.costred);
    if (p->res_block) (void)hipFree(p->res_block);
    if (p->rd.zx) (void)hipFree(p->rd.zx);
    if (p->pin_wgmap) (void)hipHostFree(p->pin_wgmap);
    if (p->rd.stamps) (void)hipFree(p->rd.stamps);
    if (p->d_notgrid) (void)hipFree(p->d_notgrid);
    if (p->d_t64list) (void)hipFree(p->d_t64list);
    if (p->lm_block) (void)hipFree(p->lm_block);
    if (p->pd.lmred) (void)hipFree(p->pd.lmred);
    if (p->block) (void)hipFree(p->block);
    delete p;
}

// make sure the scalar arrays can hold lIterations PCG iterations / ncost cost entries
static void plan_reserve(Opt_Plan* p, int lIterations, int ncost)
{
    Opt_State* st = p->st;
    // (the stream is drained only where a buffer that earlier launches may still use is replaced: a first allocation
    //  must not wait for another solver object's running solve)
    if (lIterations > p->lcap || !p->pd.red) {
        if (p->pd.red) HC(hipStreamSynchronize(st->stream));
        plan_drop_graph(p);
        if (p->pd.red) HC(hipFree(p->pd.red));
        p->lcap = lIterations < 16 ? 16 : lIterations;
        p->pd.nslots = 2 * p->lcap + 1;
        HC(hipMalloc(&p->pd.red, (size_t)p->batch * p->pd.nslots * NSHARD * sizeof(double)));
    }
    if (ncost > p->ccap || !p->pd.costred) {
        if (p->pd.costred) HC(hipStreamSynchronize(st->stream));
        plan_drop_graph(p);
        if (p->pd.costred) HC(hipFree(p->pd.costred));
        p->ccap = ncost < 16 ? 16 : ncost;
        p->pd.ncost = p->ccap;
        HC(hipMalloc(&p->pd.costred, (size_t)p->batch * p->pd.ncost * NSHARD * sizeof(double)));
    }
}

static void plan_upload_slots(Opt_Plan* p)
{
    if (p->uploaded.size() == p->hslots.size() &&
        memcmp(p->uploaded.data(), p->hslots.data(), sizeof(Slot) * p->hslots.size()) == 0)
        return;
    // pageable source: the copy is staged before the call returns
    HC(hipMemcpyAsync(p->pd.slots, p->hslots.data(), sizeof(Slot) * p->hslots.size(), hipMemcpyHostToDevice,
                      p->st->stream));
    p->uploaded = p->hslots;
}

#define LAUNCH(p, st_, kname_, kern, grid, blk, ...)                                          \
    do {                                                                                    \
        if ((p)->st->timing) {                                                              \
            KernelTimer::Rec r_;                                                            \
            r_.name = kname_;                                                               \
            HC(hipEventCreate(&r_.a)); HC(hipEventCreate(&r_.b));                           \
            HC(hipEventRecord(r_.a, st_));                                                  \
            hipLaunchKernelGGL(kern, grid, blk, 0, st_, __VA_ARGS__);                       \
            HC(hipEventRecord(r_.b, st_));                                                  \
            (p)->st->ktimer.recs.push_back(r_);                                             \
        } else {                                                                            \
            hipLaunchKernelGGL(kern, grid, blk, 0, st_, __VA_ARGS__);                       \
        }                                                                                   \
    } while (0)

#define LAUNCH_DYN(p, st_, kname_, kern, grid, blk, lds_, ...)                               \
    do {                                                                                    \
        if ((p)->st->timing) {                                                              \
            KernelTimer::Rec r_;                                                            \
            r_.name = kname_;                                                               \
            HC(hipEventCreate(&r_.a)); HC(hipEventCreate(&r_.b));                           \
            HC(hipEventRecord(r_.a, st_));                                                  \
            hipLaunchKernelGGL(kern, grid, blk, lds_, st_, __VA_ARGS__);                    \
            HC(hipEventRecord(r_.b, st_));                                                  \
            (p)->st->ktimer.recs.push_back(r_);                                             \
        } else {                                                                            \
            hipLaunchKernelGGL(kern, grid, blk, lds_, st_, __VA_ARGS__);                    \
        }                                                                                   \
    } while (0)

static bool plan_active_tiles_majority(const Opt_Plan* p)
{
    long act = 0;
    for (int b = 0; b < p->nb; ++b) act += p->h_ntiles[b];
    const long all = (long)((p->W + RT_X - 1) / RT_X) * ((p->H + RT_Y - 1) / RT_Y);
    return 2 * act >= (long)p->nb * all;
}

// The lean streaming schedule (arap_stream.h: k_pcg_a_march2 / k_pcg_b4_r, 126 instead of 146 B per vertex and
// iteration): frame-solver plans only (nothing else reads z or an up-to-date delta between the two phases), pixel-grid
// UrShape, Gauss-Newton, 16-byte alignment of every frame's images -- and most tiles active: its phase A carries more
// loads per stage, which pays where the rows are full (1920x1080 mask == 0: 94.2 -> 86.3 ms per 4 x 400 iterations; eight
// 854x480 mask == 0 frames: 167.5 -> 138.9) and loses on sparse masks (eight DAVIS-shaped frames: 75.9 -> 80.0).
// ARAPOPT_STREAM_A=2 keeps the round-2 pair.
static bool plan_lean_stream(const Opt_Plan* p)
{
    return p->res_frames_any && p->grid_u && !p->pd.lm && p->st->tile < 0 && (p->N & 3) == 0 && !p->st->force_b8 &&
           p->st->stream_a == 0 && plan_active_tiles_majority(p);
}

// phase A of the two-kernel path: direct-load kernel or an LDS-staged tile shape (ArapFlow_SetTile)
static const int kTileShapes[6][2] = {{0, 0}, {16, 16}, {32, 8}, {64, 4}, {32, 16}, {64, 8}};

static void launch_pcg_a(Opt_Plan* p, hipStream_t s, int l)
{
    int v = p->st->tile;
    if (v < 0 && p->grid_u && !p->pd.lm) {
        // default for the pixel-grid UrShape (every frame-solver plan): the marching kernel of arap_stream.h -- no
        // UrShape loads, every vertex fetched once, XCD-aware strip order; 1-D launch of frames x 8 x ceil(tiles / 8)
        // blocks of 4 rows a workgroup marches through: 5 for the round-2 pair, 7 for the lean schedule, whose phase A holds
        // more loads per stage (84 VGPRs: 5 workgroups per CU) -- 30 x 39 = 1170 workgroups at 1920x1080 are all resident at
        // once with 7 blocks, 1624 with 5 are not (sweep 3 / 5 / 6 / 7 / 8 / 10: 46.6 / 43.8 / 44.5 / 38.4 / 38.6 / 42.2 us)
        constexpr int RB = 5, RB2 = 7;
        const int sX = p->pd.tilesX;
        if (p->st->stream_a == 0 && plan_lean_stream(p)) {
            const int cY2 = (p->H + 4 * RB2 - 1) / (4 * RB2), chunk2 = (sX * cY2 + 7) / 8;
            LAUNCH(p, s, "PCGStepA", (k_pcg_a_march2<RB2>), dim3((unsigned)(p->nb * 8 * chunk2)), dim3(256), p->pd, l, sX, cY2, chunk2);
            return;
        }
        const int cY = (p->H + 4 * RB - 1) / (4 * RB), chunk = (sX * cY + 7) / 8;
        if (p->st->stream_a == 1) {
            constexpr int TX = 64, TY = 8;
            const int tX = (p->W + TX - 1) / TX, tY = (p->H + TY - 1) / TY, ch = (tX * tY + 7) / 8;
            LAUNCH(p, s, "PCGStepA", (k_pcg_a_grid<TX, TY>), dim3((unsigned)(p->nb * 8 * ch)), dim3(TX, TY, 1), p->pd, l, tX,
                   tY, ch);
        } else {
            LAUNCH(p, s, "PCGStepA", (k_pcg_a_march<RB>), dim3((unsigned)(p->nb * 8 * chunk)), dim3(256), p->pd, l, sX, cY, chunk);
        }
        return;
    }
    if (v < 0) {
        // generic UrShape: LDS-staged 64x8 tiles when most tiles are active (profiles/r01_tile_sweep_two_kernel_path.txt:
        // +14 % at full masks), direct loads for sparse masks (the staging of empty halo rows does not pay)
        v = plan_active_tiles_majority(p) ? 5 : 0;
    }
    if (v == 0) {
        LAUNCH(p, s, "PCGStepA", k_pcg_a, p->grid(), p->blk(), p->pd, l);
        return;
    }
    const int TX = kTileShapes[v][0], TY = kTileShapes[v][1];
    const dim3 g((p->W + TX - 1) / TX, (p->H + TY - 1) / TY, p->nb), b(TX, TY, 1);
    switch (v) {
    case 1: LAUNCH(p, s, "PCGStepA", (k_pcg_a_lds<16, 16>), g, b, p->pd, l); break;
    case 2: LAUNCH(p, s, "PCGStepA", (k_pcg_a_lds<32, 8>), g, b, p->pd, l); break;
    case 3: LAUNCH(p, s, "PCGStepA", (k_pcg_a_lds<64, 4>), g, b, p->pd, l); break;
    case 4: LAUNCH(p, s, "PCGStepA", (k_pcg_a_lds<32, 16>), g, b, p->pd, l); break;
    default: LAUNCH(p, s, "PCGStepA", (k_pcg_a_lds<64, 8>), g, b, p->pd, l); break;
    }
}

// Frame solver on the resident path: the resident launch applies the step itself (X += delta, cos/sin of the new Angle:
// ResDev::fuse_update) and the init kernel zeroes the granules, so a step is [init, resident launches] and a lone k_gn_prep
// (flags, tile activity) runs only where no step came before (the first step after the ramp moved the constraints, or
// after a step on another path): Opt_Plan::prep_done.
// (not in a verbose solve: that one checks every step by itself and redoes a failed step alone, which needs the step's
//  update left undone)
static bool plan_fused_prep(Opt_Plan* p) { return plan_resident_eligible(p) && p->res_frames && p->st->verbosity == 0; }

// Grid of the list launches (k_gn_prep / k_gn_init / k_gn_update over the frames' active 64x4 tiles): the longest list,
// rounded up to a multiple of 64 workgroups so that batches of similar frames replay the same captured graph (a
// workgroup beyond its frame's list only reports a zero to the order-fixed sums).
static int plan_list_blocks(Opt_Plan* p)
{
    int maxn = 1;
    for (int k = 0; k < p->nb; ++k) maxn = std::max(maxn, p->h_t64n[k]);
    return std::min((maxn + 63) / 64 * 64, p->pd.tilesX * p->pd.tilesY);
}

// enqueue the kernels of one Gauss-Newton step (without the cost) on stream s
// part: GN_STEP_ALL = prep, init, PCG, update;  GN_STEP_PREP = the lone prep;  GN_STEP_FUSED = lean init, resident launch(es)
// that apply the step themselves
enum { GN_STEP_ALL = 0, GN_STEP_PREP = 1, GN_STEP_FUSED = 2 };
static void enqueue_gn_step(Opt_Plan* p, hipStream_t s, int part = GN_STEP_ALL)
{
    const int L = p->sp.lIterations;
    const dim3 g = p->grid(), b = p->blk();
    const bool res = plan_resident_eligible(p);
    PlanDev pd = p->pd;
    const size_t gran_per_launch = RES_GRAN_PER_LAUNCH;                  // u64 entries
    if (res) {
        // k_gn_prep zeroes slot 0 of `red` (rho_0) and the granules of every launch of this step: no memset nodes.
        // Granule tags restart at 1 in every launch (cdna guide G16 "re-initialise every call").
        pd.res_gran = p->rd.gran;
        pd.res_gran_n = (int)(gran_per_launch * p->res_sets);
    } else {
        // reduction slots 0 .. 2L of every active frame (contiguous because slot stride is nslots)
        HC(hipMemsetAsync(p->pd.red, 0, (size_t)p->nb * p->pd.nslots * NSHARD * sizeof(double), s));
    }
    // frame solver on the resident path: the per-step kernels visit the frames' active 64x4 tiles only
    const bool lists = res && p->res_frames && p->d_t64list != nullptr;
    PlanDev pdl = p->pd;
    dim3 gl = g;
    if (lists) {
        const int maxn = plan_list_blocks(p);
        pd.t64list = pdl.t64list = p->d_t64list;
        pd.t64n = pdl.t64n = p->d_t64n;
        gl = dim3((unsigned)maxn, 1, (unsigned)p->nb);
    }
    if (part != GN_STEP_FUSED) LAUNCH(p, s, "GNPrep", k_gn_prep, gl, b, pd);
    if (part == GN_STEP_PREP) return;
    // (frame solver on the resident path: no UrShape loads, no stores of what the resident kernel does not read)
    // (... and it zeroes the granules of the launches that follow: pd carries them)
    if (part == GN_STEP_FUSED) LAUNCH(p, s, "PCGInit1", k_gn_init_resf, gl, b, pd);
    else LAUNCH(p, s, "PCGInit1", k_gn_init, gl, b, pdl);
    if (res) {
        // all L iterations in one launch, state on chip (arap_resident.h)
        ResDev rd = p->rd;
        rd.fuse_update = part == GN_STEP_FUSED ? 1 : 0;          // frame solver: the launch applies the step itself
        for (int set = 0; set < p->res_sets; ++set) {
            rd.wgmap = p->d_wgmap + (size_t)set * RES_WGS;
            rd.gran = p->rd.gran + gran_per_launch * set;
            const ResidentKernel kern = (ResidentKernel)resident_kernel(rd.stamps != nullptr, p->res_ns[set]);
            if (rd.stamps)
                hipLaunchKernelGGL(kern, dim3(RES_WGS), dim3(RES_THREADS), RES_LDS_BYTES, s, p->pd, rd, L);
            else
                LAUNCH_DYN(p, s, "PCGResident", kern, dim3(RES_WGS), dim3(RES_THREADS), RES_LDS_BYTES, p->pd, rd, L);
        }
    } else {
        for (int l = 0; l < L; ++l) {
            launch_pcg_a(p, s, l);
            if ((p->N & 3) == 0 && !p->pd.lm && !p->st->force_b8) {     // 16-byte accesses need every frame's images 16-byte aligned
                const dim3 gq((p->N / 4 + 255) / 256, p->nb, 1);
                if (plan_lean_stream(p)) LAUNCH(p, s, "PCGStepB", k_pcg_b4_r, gq, dim3(256), p->pd, l);
                else if (p->st->tile < 0) LAUNCH(p, s, "PCGStepB", k_pcg_b4_lean, gq, dim3(256), p->pd, l);
                else LAUNCH(p, s, "PCGStepB", k_pcg_b4, gq, dim3(256), p->pd, l);        // (explicit variants: the sweep's baseline)
            }
            else
                LAUNCH(p, s, "PCGStepB", k_pcg_b, g, b, p->pd, l);
        }
    }
    // (the lean streaming schedule leaves the last iteration's delta += alpha p to the update kernel)
    const int lag = (!res && L > 0 && plan_lean_stream(p)) ? L - 1 : -1;
    if (part != GN_STEP_FUSED) LAUNCH(p, s, "PCGLinearUpdate", k_gn_update, gl, b, pdl, lag);
}

// nsteps consecutive Gauss-Newton steps (one graph launch: between two graphs the GPU idles 8.6 us, inside one 0.2 us
// per kernel boundary)
static void plan_gn_step(Opt_Plan* p, int nsteps = 1)
{
    Opt_State* st = p->st;
    const bool graph_ok = st->use_graph && !st->timing;
    const bool res = plan_resident_eligible(p);
    if (res && plan_resident_pack(p)) {
        // new deal of solves to workgroups (the frames' active-tile counts changed): upload the tables (pageable
        // source: staged before the call returns; stream ordered behind earlier launches) and re-capture
        if (p->hole_pending) {
            std::vector<ResWg> holed = p->h_wgmap;
            for (ResWg& w : holed)
                if (w.slot >= 0 && w.wgs > 1 && w.rank == w.wgs - 1) { w = ResWg{-1, 0, 0, 0}; break; }
            HC(hipStreamSynchronize(st->stream));
            HC(hipMemcpy(p->d_wgmap, holed.data(), holed.size() * sizeof(ResWg), hipMemcpyHostToDevice));
            p->hole_pending = false;
        } else {
            // (from pinned staging: a pageable source makes the call wait until the stream has drained -- i.e. for the
            //  other solver object's whole solve.  The staging buffer is rewritten only by this plan's next deal, which
            //  comes after this solve has been waited for.)
            memcpy(p->pin_wgmap, p->h_wgmap.data(), p->h_wgmap.size() * sizeof(ResWg));
            HC(hipMemcpyAsync(p->d_wgmap, p->pin_wgmap, p->h_wgmap.size() * sizeof(ResWg), hipMemcpyHostToDevice,
                              st->stream));
        }
        // (the captured launches bake in only the number of launches, their slot counts and the list length: all part
        //  of the graph's key below, so a new deal of the same shape replays the old graph)
    }
    if (res) p->res_launches += (unsigned)(p->res_sets * nsteps);          // launches executed (graph replays included)
    const bool fused = plan_fused_prep(p);
    const int part = fused ? GN_STEP_FUSED : GN_STEP_ALL;
    if (fused && !p->prep_done) enqueue_gn_step(p, st->stream, GN_STEP_PREP);
    p->prep_done = fused;
    if (!graph_ok) {
        for (int k = 0; k < nsteps; ++k) enqueue_gn_step(p, st->stream, part);
        return;
    }
    // the captured launches bake in the path (resident: number of launches; two-kernel: phase-A variant)
    const int res_now = res ? p->res_sets : -(2 + p->st->tile) - 8 * (int)(plan_active_tiles_majority(p)) - 16 * (int)p->grid_u -
                                            32 * (int)plan_lean_stream(p);
    int maxn_now = 0;
    if (res && p->res_frames && p->d_t64list) maxn_now = plan_list_blocks(p);
    unsigned long long ns_now = 1469598103934665603ull;          // (FNV-1a over the launches' slot counts)
    if (res)
        for (int set = 0; set < p->res_sets; ++set) ns_now = (ns_now ^ (unsigned long long)p->res_ns[set]) * 1099511628211ull;
    if (!p->gexec || p->g_l != p->sp.lIterations || p->g_nb != p->nb || p->g_res != res_now || p->g_maxn != maxn_now ||
        p->g_steps != nsteps || p->g_ns != ns_now) {
        plan_drop_graph(p);
        HC(hipStreamBeginCapture(st->cap, hipStreamCaptureModeRelaxed));
        for (int k = 0; k < nsteps; ++k) enqueue_gn_step(p, st->cap, part);
        HC(hipStreamEndCapture(st->cap, &p->graph));
        HC(hipGraphInstantiate(&p->gexec, p->graph, nullptr, nullptr, 0));
        p->g_l = p->sp.lIterations;
        p->g_nb = p->nb;
        p->g_res = res_now;
        p->g_maxn = maxn_now;
        p->g_steps = nsteps;
        p->g_ns = ns_now;
    }
    HC(hipGraphLaunch(p->gexec, st->stream));
}

static void plan_cost(Opt_Plan* p, int index)
{
    LAUNCH(p, p->st->stream, "computeCost", k_cost, p->grid(), p->blk(), p->pd, index);
}

// blocking read of the cost entry `index` of slot b (sum of its shards, rounded to float as the
// reference's device float, solverGPUGaussNewton.t:790-797)
static double plan_read_cost(Opt_Plan* p, int b, int index)
{
    double sh[NSHARD];
    HC(hipMemcpyAsync(sh, p->pd.costred + ((size_t)b * p->pd.ncost + index) * NSHARD, sizeof(sh),
                      hipMemcpyDeviceToHost, p->st->stream));
    HC(hipStreamSynchronize(p->st->stream));
    plan_check_resident_error(p);
    double t = 0.0;
    for (int i = 0; i < NSHARD; ++i) t += sh[i];
    return (double)(float)t;
}

static void slot_from_params(Slot& s, void** pp)
{
    // plan-declared indices: arap_plan.t:2-8 ; scalars are HOST pointers (util.t:664-692)
    s.O = (float2*)pp[0];
    s.A = (float*)pp[1];
    s.U = (const float2*)pp[2];
    s.C = (const float2*)pp[3];
    s.M = (const float*)pp[4];
    s.wf = *(const float*)pp[5];
    s.wr = *(const float*)pp[6];
}

// init: solverGPUGaussNewton.t:956-1007
static void plan_init(Opt_Plan* p)
{
    HC(hipSetDevice(p->st->device));
    p->sp.nIter = 0;
    p->prep_done = false;                  // the caller may have changed Mask / Constraints (the ramp does)
    plan_reserve(p, p->sp.lIterations, p->sp.nIterations + 1);
    plan_upload_slots(p);
    if (p->lazy_cost && !p->cost_wanted && p->st->verbosity == 0) return;
    HC(hipMemsetAsync(p->pd.costred, 0, (size_t)p->nb * p->pd.ncost * NSHARD * sizeof(double), p->st->stream));
    if (!p->lazy_cost || p->st->verbosity > 0 || p->sp.nIterations == 0) plan_cost(p, 0);
}

// step: solverGPUGaussNewton.t:1016-1177 (GN branch)
static int plan_step(Opt_Plan* p)
{
    if (p->sp.nIter < p->sp.nIterations) {
        plan_upload_slots(p);
        // the caller may have changed Mask / UrShape since Init or the last Step (Opt.h:58-66)
        // -- looked at before EVERY step, whatever path the step will take: grid_u (no UrShape loads in phase A of the
        // two-kernel path) is a property of the images as they are NOW, also with the resident kernel switched off,
        // paused after a timeout or absent on this device.
        if (!p->res_frames) plan_analyse_for_resident(p);
        const bool used_res = plan_resident_eligible(p);
        plan_gn_step(p);
        if (used_res && (!p->res_frames || p->st->verbosity > 0)) {
            // drop-in plan: the caller may read the unknowns right after this Step, so make sure it happened (a verbose
            // frame solve reads the costs below: same check, instead of at the end of ArapFlow_SolverSolve)
            HC(hipStreamSynchronize(p->st->stream));
            if (plan_resident_failed(p)) plan_gn_step(p);          // X untouched: redo on the two-kernel path
        }
        if (!p->lazy_cost || p->st->verbosity > 0 || (p->cost_wanted && p->sp.nIter + 1 == p->sp.nIterations))
            plan_cost(p, p->sp.nIter + 1);
        if (p->st->verbosity > 0) {
            const double a = plan_read_cost(p, 0, p->sp.nIter), b = plan_read_cost(p, 0, p->sp.nIter + 1);
            printf("cost: %f -> %f\n", a, b);
        }
        p->sp.nIter += 1;
        return 1;
    }
    if (p->st->timing && p->st->verbosity > 0) p->st->ktimer.report();
    return 0;
}

// Frame solver, quiet: all remaining Gauss-Newton steps of the ramp step as ONE graph launch (nothing on the host looks
// at a step's result before the next: the cost is wanted after the last one at most, lazy_cost).  False: not applicable,
// the caller steps one by one.
static bool plan_steps_batched(Opt_Plan* p)
{
    const int n = p->sp.nIterations - p->sp.nIter;
    if (!p->res_frames || !p->lazy_cost || p->st->verbosity > 0 || !p->st->use_graph || p->st->timing || n < 2 ||
        !plan_resident_eligible(p))
        return false;
    plan_upload_slots(p);
    plan_gn_step(p, n);
    if (p->cost_wanted) plan_cost(p, p->sp.nIterations);
    p->sp.nIter = p->sp.nIterations;
    return true;
}

// ---------------------------------------------------------------------------------------------
// "LMGPU": host loop of the Levenberg-Marquardt branch (solverGPUGaussNewton.t:1016-1177 with UsesLambda).
// Like the reference it is host driven: Q is read back after every PCG iteration for the zeta test (:1093-1102)
// and the costs after every step (:1119-1157); no graph, one frame (Opt_* plans only).
// ---------------------------------------------------------------------------------------------
static void plan_lm_alloc(Opt_Plan* p)
{
    if (p->lm_block) return;
    const size_t N = (size_t)p->N;
    const size_t sz2 = align_up(N * sizeof(float2), 256), sz1 = align_up(N * sizeof(float), 256);
    HC(hipMalloc(&p->lm_block, 5 * sz2 + 5 * sz1));
    HC(hipMemsetAsync(p->lm_block, 0, 5 * sz2 + 5 * sz1, p->st->stream));
    char* c = (char*)p->lm_block;
    auto take = [&](size_t s) { char* r = c; c += s; return r; };
    p->pd.bO = (float2*)take(sz2); p->pd.CtCO = (float2*)take(sz2); p->pd.SSqO = (float2*)take(sz2);
    p->pd.AdO = (float2*)take(sz2); p->prevO = (float2*)take(sz2);
    p->pd.bA = (float*)take(sz1); p->pd.CtCA = (float*)take(sz1); p->pd.SSqA = (float*)take(sz1);
    p->pd.AdA = (float*)take(sz1); p->prevA = (float*)take(sz1);
}

static double plan_read_shards(Opt_Plan* p, const double* dev)
{
    double sh[NSHARD];
    HC(hipMemcpyAsync(sh, dev, sizeof(sh), hipMemcpyDeviceToHost, p->st->stream));
    HC(hipStreamSynchronize(p->st->stream));
    double t = 0.0;
    for (int i = 0; i < NSHARD; ++i) t += sh[i];
    return t;
}

static void plan_init_lm(Opt_Plan* p)
{
    HC(hipSetDevice(p->st->device));
    plan_lm_alloc(p);
    p->sp.nIter = 0;
    p->pd.lm = 1;
    plan_reserve(p, p->sp.lIterations, 2);
    if (p->sp.lIterations + 2 > p->lm_lcap || !p->pd.lmred) {
        HC(hipStreamSynchronize(p->st->stream));
        if (p->pd.lmred) HC(hipFree(p->pd.lmred));
        p->lm_lcap = p->sp.lIterations + 2;
        HC(hipMalloc(&p->pd.lmred, (size_t)p->lm_lcap * NSHARD * sizeof(double)));
    }
    plan_upload_slots(p);
    p->lm_radius = p->sp.trust_region_radius;                 // init copies the solver parameters (:996-1001)
    p->lm_decrease = p->sp.radius_decrease_factor;
    p->lm_done = false;
    HC(hipMemsetAsync(p->pd.costred, 0, (size_t)p->pd.ncost * NSHARD * sizeof(double), p->st->stream));
    hipLaunchKernelGGL(k_cost, p->grid(), p->blk(), 0, p->st->stream, p->pd, 0);
    p->lm_prev_cost = plan_read_cost(p, 0, 0);
}

static int plan_step_lm(Opt_Plan* p)
{
    Opt_State* st = p->st;
    hipStream_t s = st->stream;
    const SolverParameters& sp = p->sp;
    if (p->lm_done || sp.nIter >= sp.nIterations) return 0;
    plan_upload_slots(p);
    const dim3 g = p->grid(), b = p->blk();
    const int L = sp.lIterations;
    const size_t N = (size_t)p->N;
    HC(hipMemsetAsync(p->pd.red, 0, (size_t)p->pd.nslots * NSHARD * sizeof(double), s));
    HC(hipMemsetAsync(p->pd.lmred, 0, (size_t)p->lm_lcap * NSHARD * sizeof(double), s));
    hipLaunchKernelGGL(k_gn_prep, g, b, 0, s, p->pd);
    hipLaunchKernelGGL(k_gn_init, g, b, 0, s, p->pd);
    HC(hipMemsetAsync(p->pd.red, 0, NSHARD * sizeof(double), s));            // scanAlphaNumerator again (:1041)
    hipLaunchKernelGGL(k_lm_prepare, g, b, 0, s, p->pd, p->lm_radius, sp.min_lm_diagonal, sp.max_lm_diagonal,
                       sp.nIter == 0 ? 1 : 0);
    float Q0 = (float)plan_read_shards(p, p->pd.lmred);
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(k_pcg_a, g, b, 0, s, p->pd, l);
        if (((l + 1) % sp.residual_reset_period) == 0) {
            hipLaunchKernelGGL(k_lm_step2a, g, b, 0, s, p->pd, l);
            hipLaunchKernelGGL(k_lm_apply, g, b, 0, s, p->pd, (const float2*)p->pd.deltaO, (const float*)p->pd.deltaA,
                               p->pd.AdO, p->pd.AdA);
            hipLaunchKernelGGL(k_lm_step2b, g, b, 0, s, p->pd, l);
        } else {
            hipLaunchKernelGGL(k_pcg_b, g, b, 0, s, p->pd, l);
        }
        const float Q1 = (float)plan_read_shards(p, p->pd.lmred + (size_t)(l + 1) * NSHARD);
        const float zeta = (float)(l + 1) * (Q1 - Q0) / Q1;
        if (zeta < sp.q_tolerance) break;
        Q0 = Q1;
    }
    hipLaunchKernelGGL(k_lm_model_cost, g, b, 0, s, p->pd, p->lm_lcap - 1);
    const float model_cost = (float)plan_read_shards(p, p->pd.lmred + (size_t)(p->lm_lcap - 1) * NSHARD);
    const float model_cost_change = (float)p->lm_prev_cost - model_cost;
    const Slot& sl = p->hslots[0];
    HC(hipMemcpyAsync(p->prevO, sl.O, N * sizeof(float2), hipMemcpyDeviceToDevice, s));   // savePreviousUnknowns
    HC(hipMemcpyAsync(p->prevA, sl.A, N * sizeof(float), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_gn_update, g, b, 0, s, p->pd, -1);
    HC(hipMemsetAsync(p->pd.costred + NSHARD, 0, NSHARD * sizeof(double), s));
    hipLaunchKernelGGL(k_cost, g, b, 0, s, p->pd, 1);
    const double newCost = plan_read_cost(p, 0, 1);
    const float cost_change = (float)p->lm_prev_cost - (float)newCost;
    const float relative_decrease = cost_change / model_cost_change;
    if (cost_change >= 0 && relative_decrease > sp.min_relative_decrease) {
        if (cost_change <= (float)p->lm_prev_cost * sp.function_tolerance) {
            if (st->verbosity > 0) printf("\nFunction tolerance reached, exiting\n");
            p->lm_done = true;
            return 0;
        }
        const double step_quality = relative_decrease;
        const double tmp_factor = 1.0 - pow(2.0 * step_quality - 1.0, 3.0);
        p->lm_radius = (float)((double)p->lm_radius / fmax(1.0 / 3.0, tmp_factor));
        p->lm_radius = (float)fmin((double)p->lm_radius, (double)sp.max_trust_region_radius);
        p->lm_decrease = 2.0f;
        p->lm_prev_cost = newCost;
    } else {
        HC(hipMemcpyAsync(sl.O, p->prevO, N * sizeof(float2), hipMemcpyDeviceToDevice, s));   // revertUpdate
        HC(hipMemcpyAsync(sl.A, p->prevA, N * sizeof(float), hipMemcpyDeviceToDevice, s));
        p->lm_radius = p->lm_radius / p->lm_decrease;
        p->lm_decrease = 2.0f * p->lm_decrease;
        if (p->lm_radius <= sp.min_trust_region_radius) {
            if (st->verbosity > 0) printf("\nTrust_region_radius is less than the min, exiting\n");
            p->lm_done = true;
            return 0;
        }
    }
    if (st->verbosity > 0) printf("cost: %f (trust_region_radius %g)\n", p->lm_prev_cost, p->lm_radius);
    p->sp.nIter += 1;
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Problem specification check.  The library hard-codes the energy of arap_plan.t:1-23; the file
// named in Opt_ProblemDefine is checked declaration by declaration against it.
// ---------------------------------------------------------------------------------------------
static std::string strip_spec(const std::string& src)
{
    std::string out;
    size_t i = 0;
    while (i < src.size()) {
        if (src[i] == '-' && i + 1 < src.size() && src[i + 1] == '-') {   // Lua comment
            while (i < src.size() && src[i] != '\n') ++i;
            continue;
        }
        if (!isspace((unsigned char)src[i])) out.push_back(src[i]);
        ++i;
    }
    return out;
}

static bool spec_is_arap(const std::string& stripped, std::string& why)
{
    // every structural element of the energy must be present, in this order of appearance
    static const char* need[] = {
        "Dim(\"W\",0)", "Dim(\"H\",1)",
        "Unknown(\"Offset\",opt_float2,{W,H},0)",
        "Unknown(\"Angle\",opt_float,{W,H},1)",
        "Array(\"UrShape\",opt_float2,{W,H},2)",
        "Array(\"Constraints\",opt_float2,{W,H},3)",
        "Array(\"Mask\",opt_float,{W,H},4)",
        "Param(\"w_fitSqrt\",float,5)",
        "Param(\"w_regSqrt\",float,6)",
        "UsePreconditioner(true)",
        "Exclude(Not(eq(Mask(0,0),0)))",
        "Stencil{{1,0},{-1,0},{0,1},{0,-1}}",
        "w_regSqrt*((Offset(0,0)-Offset(x,y))-Rotate2D(Angle(0,0),(UrShape(0,0)-UrShape(x,y))))",
        "InBounds(x,y)*eq(Mask(x,y),0)*eq(Mask(0,0),0)",
        "Energy(Select(valid,e_reg,0))",
        "(Offset(0,0)-Constraints(0,0))",
        "All(greatereq(Constraints(0,0),0))",
        "Energy(w_fitSqrt*Select(valid,e_fit,0.0))",
    };
    size_t pos = 0;
    for (const char* n : need) {
        size_t f = stripped.find(n, pos);
        if (f == std::string::npos) { why = std::string("missing or out of order: ") + n; return false; }
        pos = f + strlen(n);
    }
    // and nothing else that adds energy terms or unknowns
    size_t cnt = 0, at = 0;
    while ((at = stripped.find("Energy(", at)) != std::string::npos) { ++cnt; at += 7; }
    if (cnt != 2) { why = "expected exactly two Energy() terms"; return false; }
    cnt = 0; at = 0;
    while ((at = stripped.find("Unknown(", at)) != std::string::npos) { ++cnt; at += 8; }
    if (cnt != 2) { why = "expected exactly two Unknown() declarations"; return false; }
    return true;
}

// ---------------------------------------------------------------------------------------------
// C ABI, part 1
// ---------------------------------------------------------------------------------------------
extern "C" {

Opt_State* Opt_NewState(Opt_InitializationParameters params)
{
    if (params.doublePrecision) {
        fprintf(stderr, "arapopt: doublePrecision is not supported (float32 only, as the application uses)\n");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "arapopt: no HIP device available; this library has no CPU fallback\n");
        return nullptr;
    }
    Opt_State* st = new Opt_State();
    st->verbosity = params.verbosityLevel;
    st->timing = params.collectPerKernelTimingInfo;
    HC(hipGetDevice(&st->device));
    HC(hipStreamCreateWithFlags(&st->cap, hipStreamNonBlocking));
    HC(hipEventCreate(&st->t0));
    HC(hipEventCreate(&st->t1));
    const char* ng = getenv("ARAPOPT_NO_GRAPH");
    st->use_graph = !(ng && ng[0] == '1');
    if (const char* tv = getenv("ARAPOPT_TILE")) {                     // experiments: phase-A variant, "TXxTY" or "0x0"
        int tx = -1, ty = -1;
        if (sscanf(tv, "%dx%d", &tx, &ty) == 2)
            for (int v = 0; v < 6; ++v)
                if (kTileShapes[v][0] == tx && kTileShapes[v][1] == ty) st->tile = v;
    }
    if (const char* b8 = getenv("ARAPOPT_B8")) st->force_b8 = b8[0] == '1';   // experiments: 8-byte form of phase B
    if (const char* sa = getenv("ARAPOPT_STREAM_A")) st->stream_a = atoi(sa);
    return st;
}

void ArapFlow_FreeState(Opt_State* st)
{
    if (!st) return;
    (void)hipStreamDestroy(st->cap);
    if (st->own_stream) (void)hipStreamDestroy(st->own_stream);
    (void)hipEventDestroy(st->t0);
    (void)hipEventDestroy(st->t1);
    st->ktimer.clear();
    delete st;
}

Opt_Problem* Opt_ProblemDefine(Opt_State* state, const char* filename, const char* solverkind)
{
    if (!state || !filename || !solverkind) return nullptr;
    const int kind = strcmp(solverkind, "gaussNewtonGPU") == 0 ? 0 : (strcmp(solverkind, "LMGPU") == 0 ? 1 : -1);
    if (kind < 0) {                                                       // asserted at o.t:122
        fprintf(stderr, "arapopt: unknown solver kind '%s' (expected gaussNewtonGPU or LMGPU)\n", solverkind);
        return nullptr;
    }
    if (strcmp(filename, "builtin:arap") != 0) {
        FILE* f = fopen(filename, "rb");
        if (!f) {
            fprintf(stderr, "arapopt: cannot open problem specification '%s'\n", filename);
            return nullptr;
        }
        std::string src;
        char buf[4096];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) src.append(buf, n);
        fclose(f);
        std::string why;
        if (!spec_is_arap(strip_spec(src), why)) {
            fprintf(stderr,
                    "arapopt: '%s' is not the ARAP image-warping energy this library implements (%s)\n",
                    filename, why.c_str());
            return nullptr;
        }
    }
    Opt_Problem* pr = new Opt_Problem();
    pr->kind = kind;
    if (state->verbosity > 1) printf("arapopt: problem '%s' (%s) accepted\n", filename, solverkind);
    return pr;
}

void Opt_ProblemDelete(Opt_State*, Opt_Problem* problem) { delete problem; }

Opt_Plan* Opt_ProblemPlan(Opt_State* state, Opt_Problem* problem, unsigned int* dimensions)
{
    if (!state || !problem || !dimensions) return nullptr;
    const unsigned W = dimensions[0], H = dimensions[1];
    if (W == 0 || H == 0 || (uint64_t)W * H > (1ull << 30)) {
        fprintf(stderr, "arapopt: bad dimensions %u x %u\n", W, H);
        return nullptr;
    }
    Opt_Plan* p = plan_create(state, (int)W, (int)H, 1);
    p->kind = problem->kind;
    if (p->kind == 0) plan_enable_resident(p);
    return p;
}

void Opt_PlanFree(Opt_State*, Opt_Plan* plan) { plan_free(plan); }

void Opt_SetSolverParameter(Opt_State*, Opt_Plan* plan, const char* name, void* value)
{
    if (!plan || !name || !value) return;
    SolverParameters& sp = plan->sp;
#define SETI(f) if (strcmp(name, #f) == 0) { sp.f = *(int*)value; return; }
#define SETF(f) if (strcmp(name, #f) == 0) { sp.f = *(float*)value; return; }
    SETI(nIterations) SETI(lIterations) SETI(residual_reset_period)
    SETF(min_relative_decrease) SETF(min_trust_region_radius) SETF(max_trust_region_radius)
    SETF(q_tolerance) SETF(function_tolerance) SETF(trust_region_radius) SETF(radius_decrease_factor)
    SETF(min_lm_diagonal) SETF(max_lm_diagonal)
#undef SETI
#undef SETF
    if (plan->st->verbosity > 0) printf("Warning: tried to set nonexistent solver parameter %s\n", name);
}

void Opt_ProblemInit(Opt_State* state, Opt_Plan* plan, void** problemparams)
{
    if (state && state->res_cooldown > 0) --state->res_cooldown;
    plan->nb = 1;
    slot_from_params(plan->hslots[0], problemparams);
    if (plan->kind == 1) plan_init_lm(plan); else plan_init(plan);
}

int Opt_ProblemStep(Opt_State*, Opt_Plan* plan, void** problemparams)
{
    slot_from_params(plan->hslots[0], problemparams);
    return plan->kind == 1 ? plan_step_lm(plan) : plan_step(plan);
}

void Opt_ProblemSolve(Opt_State* state, Opt_Plan* plan, void** problemparams)
{
    Opt_ProblemInit(state, plan, problemparams);
    while (Opt_ProblemStep(state, plan, problemparams) != 0) {}
}

double Opt_ProblemCurrentCost(Opt_State*, Opt_Plan* plan)
{
    if (plan->kind == 1) return (double)(float)plan->lm_prev_cost;
    return plan_read_cost(plan, 0, plan->sp.nIter);
}

// ---------------------------------------------------------------------------------------------
// C ABI, part 2
// ---------------------------------------------------------------------------------------------
const char* ArapFlow_Version(void) { return ARAPOPT_VERSION; }

void ArapFlow_SetResident(Opt_State* state, int on)
{
    state->use_resident = on != 0;
    if (on) { state->res_cooldown = 0; state->res_backoff = 8; }     // an explicit "on" also ends a pause after a timeout
}

int ArapFlow_SetTile(Opt_State* state, int tile_x, int tile_y)
{
    if (tile_x < 0 && tile_y < 0) { state->tile = -1; return 0; }         // automatic (default)
    for (int v = 0; v < 6; ++v)
        if (kTileShapes[v][0] == tile_x && kTileShapes[v][1] == tile_y) { state->tile = v; return 0; }
    return -1;
}

void ArapFlow_SetKernelTiming(Opt_State* state, int on)
{
    HC(hipStreamSynchronize(state->stream));
    state->ktimer.clear();
    state->timing = on ? 1 : 0;
}

int ArapFlow_KernelTime(Opt_State* state, const char* kernel_name, double* total_ms, uint64_t* launches)
{
    double tot = 0.0;
    uint64_t n = 0;
    for (auto& r : state->ktimer.recs) {
        if (r.name != kernel_name) continue;
        float ms = 0.f;
        HC(hipEventSynchronize(r.b));
        HC(hipEventElapsedTime(&ms, r.a, r.b));
        tot += ms;
        ++n;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return n ? 0 : -1;
}

void ArapFlow_SetStream(Opt_State* state, void* hip_stream) { state->stream = (hipStream_t)hip_stream; }

int ArapFlow_UseOwnStream(Opt_State* state)
{
    if (!state) return -1;
    if (!state->own_stream) HC(hipStreamCreateWithFlags(&state->own_stream, hipStreamNonBlocking));
    state->stream = state->own_stream;
    return 0;
}

void ArapFlow_TimerBegin(Opt_State* state) { HC(hipEventRecord(state->t0, state->stream)); }

float ArapFlow_TimerEnd(Opt_State* state)
{
    float ms = 0.f;
    HC(hipEventRecord(state->t1, state->stream));
    HC(hipEventSynchronize(state->t1));
    HC(hipEventElapsedTime(&ms, state->t0, state->t1));
    return ms;
}

static Opt_Plan* temp_plan(Opt_State* st, unsigned W, unsigned H, const void* O, const void* A, const void* U,
                           const void* C, const void* M, float wf, float wr)
{
    Opt_Plan* p = plan_create(st, (int)W, (int)H, 1);
    Slot& s = p->hslots[0];
    s.O = (float2*)O; s.A = (float*)A; s.U = (const float2*)U; s.C = (const float2*)C; s.M = (const float*)M;
    s.wf = wf; s.wr = wr;
    p->nb = 1;
    plan_reserve(p, 1, 1);
    plan_upload_slots(p);
    HC(hipMemsetAsync(p->pd.red, 0, (size_t)p->pd.nslots * NSHARD * sizeof(double), st->stream));
    HC(hipMemsetAsync(p->pd.costred, 0, (size_t)p->pd.ncost * NSHARD * sizeof(double), st->stream));
    return p;
}

int ArapFlow_EvalJTF(Opt_State* st, unsigned W, unsigned H, const void* O, const void* A, const void* U,
                     const void* C, const void* M, float wf, float wr, void* gO, void* gA, void* dO, void* dA)
{
    Opt_Plan* p = temp_plan(st, W, H, O, A, U, C, M, wf, wr);
    hipLaunchKernelGGL(k_gn_prep, p->grid(), p->blk(), 0, st->stream, p->pd);
    hipLaunchKernelGGL(k_gn_init, p->grid(), p->blk(), 0, st->stream, p->pd);
    hipLaunchKernelGGL(k_export_jtf, p->grid(), p->blk(), 0, st->stream, p->pd, (float2*)gO, (float*)gA,
                       (float2*)dO, (float*)dA);
    hipError_t e = hipStreamSynchronize(st->stream);
    if (e == hipSuccess) e = hipGetLastError();
    plan_free(p);
    return (int)e;
}

int ArapFlow_ApplyJTJ(Opt_State* st, unsigned W, unsigned H, const void* A, const void* U, const void* C,
                      const void* M, float wf, float wr, const void* pO, const void* pA, void* outO, void* outA)
{
    Opt_Plan* p = temp_plan(st, W, H, nullptr, A, U, C, M, wf, wr);
    const size_t N = (size_t)W * H;
    hipLaunchKernelGGL(k_gn_prep, p->grid(), p->blk(), 0, st->stream, p->pd);
    HC(hipMemcpyAsync(p->pd.pO0, pO, N * sizeof(float2), hipMemcpyDeviceToDevice, st->stream));
    HC(hipMemcpyAsync(p->pd.pA0, pA, N * sizeof(float), hipMemcpyDeviceToDevice, st->stream));
    hipLaunchKernelGGL(k_pcg_a, p->grid(), p->blk(), 0, st->stream, p->pd, 0);
    HC(hipMemcpyAsync(outO, p->pd.ApO, N * sizeof(float2), hipMemcpyDeviceToDevice, st->stream));
    HC(hipMemcpyAsync(outA, p->pd.ApA, N * sizeof(float), hipMemcpyDeviceToDevice, st->stream));
    hipError_t e = hipStreamSynchronize(st->stream);
    if (e == hipSuccess) e = hipGetLastError();
    plan_free(p);
    return (int)e;
}

int ArapFlow_Cost(Opt_State* st, unsigned W, unsigned H, const void* O, const void* A, const void* U,
                  const void* C, const void* M, float wf, float wr, double* cost_host)
{
    Opt_Plan* p = temp_plan(st, W, H, O, A, U, C, M, wf, wr);
    hipLaunchKernelGGL(k_cost, p->grid(), p->blk(), 0, st->stream, p->pd, 0);
    *cost_host = plan_read_cost(p, 0, 0);
    hipError_t e = hipGetLastError();
    plan_free(p);
    return (int)e;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Batched frame solver (ArapFlow_Solver): CombinedSolver on the device
// ---------------------------------------------------------------------------------------------
namespace arap {

struct FrameDev {              // per-slot images owned by the frame solver
    float2 *O, *U, *C, *T, *flow;
    float *A, *M;
    uint8_t *mask, *rgb, *out_rgb, *out_mask;
    unsigned long long* key;
};

// resetGPU (CombinedSolver.h:207-221): U = O = (x,y), A = 0, Mask = (float)red
__global__ __launch_bounds__(256) void k_frame_reset(const FrameDev* fr, int W, int N)
{
    const FrameDev f = fr[blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int y = i / W, x = i - y * W;
    const float2 g = make_float2((float)x, (float)y);
    f.U[i] = g;
    f.O[i] = g;
    f.A[i] = 0.f;
    f.M[i] = (float)f.mask[i];
}

// setConstraintImage(alpha) (CombinedSolver.h:223-242).  T holds, per source pixel, the target of the
// last constraint placed there (host pre-pass in SetFrame, same overwrite order as the reference's
// loop), or NaN where there is none / where the mask is non-zero.
__global__ __launch_bounds__(256) void k_frame_ramp(const FrameDev* fr, int W, int N, float alpha)
{
    const FrameDev f = fr[blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float2 t = f.T[i];
    float2 c = make_float2(-1.0f, -1.0f);
    if (t.x == t.x) {
        const int y = i / W, x = i - y * W;
        c.x = (1.0f - alpha) * (float)x + alpha * t.x;
        c.y = (1.0f - alpha) * (float)y + alpha * t.y;
    }
    f.C[i] = c;
}

}  // namespace arap

extern "C" int ArapFlow_SolverWait(ArapFlow_Solver* s);

struct ArapFlow_Solver {
    Opt_State* st = nullptr;
    int W = 0, H = 0, N = 0, batch = 0;
    Opt_Plan* plan = nullptr;
    void* block = nullptr;
    std::vector<FrameDev> hfr;
    FrameDev* dfr = nullptr;
    WarpJob* djobs = nullptr;
    WarpJob* pin_jobs = nullptr;     // pinned staging of the warp jobs of one solve call
    unsigned* pin_err = nullptr;     // the resident kernel's error word as of the end of the last solve call (pinned)
    std::vector<uint8_t> has_rgb;
    std::vector<uint64_t> nactive;
    uint64_t last_pcg = 0, last_active = 0, last_grid = 0;
    unsigned last_n = 0;
    int last_cost_index = 0;
    // Host <-> device traffic runs on the solver's own copy stream through pinned staging, ordered against the
    // state's compute stream by events, so that a host can upload the next batch into one solver object and download
    // the previous results from it while ANOTHER solver object's solve occupies the compute stream (arap_deform).
    hipStream_t copy = nullptr;
    hipEvent_t ev_up = nullptr, ev_done = nullptr, ev_dl = nullptr;
    char* pin_in = nullptr;          // [batch] x {T float2[N], mask u8[N], rgb u8[3N]}
    char* pin_out = nullptr;         // [batch] x {flow float2[N], rgb u8[3N], mask u8[N]}   (allocated on first download)
    size_t pin_in_slot = 0, pin_out_slot = 0;
    bool uploads_pending = false;    // SetFrame since the last solve: the solve waits for ev_up
    bool inflight = false;           // a solve has been enqueued and not waited for
    bool retried = false;            // the last wait redid the schedule on the two-kernel path
    unsigned launches_at_enqueue = 0; // plan->res_launches when the pending solve call was enqueued
    unsigned a_n = 0, a_numIter = 0, a_nIt = 0, a_lIt = 0;
    int a_warp = 0, a_download = 0;
};

static void solver_enqueue_warp(ArapFlow_Solver* s, unsigned nframes)
{
    Opt_State* st = s->st;
    WarpJob* jobs = s->pin_jobs;             // (pinned: see plan_gn_step on pageable sources)
    for (unsigned b = 0; b < nframes; ++b) {
        const FrameDev& f = s->hfr[b];
        WarpJob& j = jobs[b];
        j.field = f.O; j.flow_in = nullptr;
        j.rgb = s->has_rgb[b] ? f.rgb : nullptr;
        j.mask = f.mask; j.flow_out = f.flow; j.key = f.key;
        j.out_rgb = s->has_rgb[b] ? f.out_rgb : nullptr;
        j.out_mask = f.out_mask;
    }
    HC(hipMemcpyAsync(s->djobs, jobs, sizeof(WarpJob) * nframes, hipMemcpyHostToDevice, st->stream));
    const dim3 g((s->W + 63) / 64, (s->H + 3) / 4, nframes);
    hipLaunchKernelGGL(k_warp_raster, g, dim3(64, 4), 0, st->stream, s->djobs, s->W, s->H);
    hipLaunchKernelGGL(k_warp_resolve, dim3((s->N + 255) / 256, 1, nframes), dim3(256), 0, st->stream, s->djobs,
                       s->N);
}

// the whole schedule of slots [0, a_n) on the compute stream (+ warp, + download on the copy stream), no waiting
static void solver_enqueue(ArapFlow_Solver* s)
{
    Opt_State* st = s->st;
    Opt_Plan* p = s->plan;
    const unsigned nframes = s->a_n, numIter = s->a_numIter;
    p->nb = (int)nframes;
    p->sp.nIterations = (int)s->a_nIt;
    p->sp.lIterations = (int)s->a_lIt;
    if (s->uploads_pending) {
        HC(hipEventRecord(s->ev_up, s->copy));
        HC(hipStreamWaitEvent(st->stream, s->ev_up, 0));
        s->uploads_pending = false;
    }
    const dim3 g1((s->N + 255) / 256, 1, nframes);
    // preSingleSolve = resetGPU (CombinedSolver.h:191-193)
    hipLaunchKernelGGL(k_frame_reset, g1, dim3(256), 0, st->stream, s->dfr, s->W, s->N);
    for (unsigned i = 0; i < numIter; ++i) {
        const float alpha = (float)(i + 1) / (float)numIter;          // CombinedSolver.h:199-201
        hipLaunchKernelGGL(k_frame_ramp, g1, dim3(256), 0, st->stream, s->dfr, s->W, s->N, alpha);
        p->lazy_cost = true;
        p->cost_wanted = i + 1 == numIter;
        plan_init(p);
        if (!plan_steps_batched(p))
            while (plan_step(p) != 0) {}
    }
    if (s->a_warp) solver_enqueue_warp(s, nframes);
    if (p->res_capable) {
        *s->pin_err = 0u;
        HC(hipMemcpyAsync(s->pin_err, p->rd.err, sizeof(unsigned), hipMemcpyDeviceToHost, st->stream));
    }
    HC(hipEventRecord(s->ev_done, st->stream));
    if (s->a_download) {
        if (!s->pin_out) {
            const size_t N = s->N;
            s->pin_out_slot = align_up(12 * N, 256);
            HC(hipHostMalloc((void**)&s->pin_out, s->pin_out_slot * s->batch, hipHostMallocDefault));
        }
        HC(hipStreamWaitEvent(s->copy, s->ev_done, 0));
        const size_t N = s->N;
        for (unsigned b = 0; b < nframes; ++b) {
            const FrameDev& f = s->hfr[b];
            char* o = s->pin_out + s->pin_out_slot * b;
            HC(hipMemcpyAsync(o, f.flow, 8 * N, hipMemcpyDeviceToHost, s->copy));
            if (s->has_rgb[b]) HC(hipMemcpyAsync(o + 8 * N, f.out_rgb, 3 * N, hipMemcpyDeviceToHost, s->copy));
            HC(hipMemcpyAsync(o + 11 * N, f.out_mask, N, hipMemcpyDeviceToHost, s->copy));
        }
        HC(hipEventRecord(s->ev_dl, s->copy));
    }
    s->last_cost_index = p->sp.nIter;
    s->last_n = nframes;
    s->last_pcg = (uint64_t)numIter * s->a_nIt * s->a_lIt;
    s->last_active = 0;
    for (unsigned b = 0; b < nframes; ++b) s->last_active += s->nactive[b];
    s->last_grid = (uint64_t)nframes * s->N;
}

extern "C" {

ArapFlow_Solver* ArapFlow_SolverCreate(Opt_State* st, unsigned W, unsigned H, unsigned batch)
{
    if (!st || W == 0 || H == 0 || batch == 0) return nullptr;
    HC(hipSetDevice(st->device));
    ArapFlow_Solver* s = new ArapFlow_Solver();
    s->st = st; s->W = (int)W; s->H = (int)H; s->N = (int)(W * H); s->batch = (int)batch;
    s->plan = plan_create(st, (int)W, (int)H, (int)batch);
    plan_enable_resident(s->plan);
    s->plan->res_frames = s->plan->res_capable;
    s->plan->res_frames_any = true;
    s->plan->grid_u = true;                                   // k_frame_reset writes U = the pixel grid
    {
        const size_t T = (size_t)s->plan->pd.tilesX * s->plan->pd.tilesY;
        HC(hipMalloc(&s->plan->d_t64list, (batch * T + batch) * sizeof(int)));
        HC(hipMemsetAsync(s->plan->d_t64list, 0, (batch * T + batch) * sizeof(int), st->stream));
        s->plan->d_t64n = s->plan->d_t64list + batch * T;
    }
    const size_t N = s->N;
    const size_t sz2 = align_up(N * sizeof(float2), 256), sz1 = align_up(N * sizeof(float), 256);
    const size_t szb = align_up(N, 256), sz3 = align_up(3 * N, 256), szk = align_up(N * 8, 256);
    const size_t per = 5 * sz2 + 2 * sz1 + 2 * szb + 2 * sz3 + szk;
    const size_t tail = align_up(sizeof(FrameDev) * batch, 256) + align_up(sizeof(WarpJob) * batch, 256);
    HC(hipMalloc(&s->block, per * batch + tail));
    HC(hipMemsetAsync(s->block, 0, per * batch + tail, st->stream));
    char* c = (char*)s->block;
    auto take = [&](size_t b) { char* r = c; c += b; return r; };
    s->hfr.resize(batch);
    for (unsigned b = 0; b < batch; ++b) {
        FrameDev& f = s->hfr[b];
        f.O = (float2*)take(sz2); f.U = (float2*)take(sz2); f.C = (float2*)take(sz2);
        f.T = (float2*)take(sz2); f.flow = (float2*)take(sz2);
        f.A = (float*)take(sz1); f.M = (float*)take(sz1);
        f.mask = (uint8_t*)take(szb); f.out_mask = (uint8_t*)take(szb);
        f.rgb = (uint8_t*)take(sz3); f.out_rgb = (uint8_t*)take(sz3);
        f.key = (unsigned long long*)take(szk);
    }
    s->dfr = (FrameDev*)take(align_up(sizeof(FrameDev) * batch, 256));
    s->djobs = (WarpJob*)take(align_up(sizeof(WarpJob) * batch, 256));
    HC(hipMemcpyAsync(s->dfr, s->hfr.data(), sizeof(FrameDev) * batch, hipMemcpyHostToDevice, st->stream));
    HC(hipStreamSynchronize(st->stream));
    HC(hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking));
    HC(hipEventCreateWithFlags(&s->ev_up, hipEventDisableTiming));
    HC(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
    HC(hipEventCreateWithFlags(&s->ev_dl, hipEventDisableTiming));
    s->pin_in_slot = align_up(12 * N, 256);
    HC(hipHostMalloc((void**)&s->pin_in, s->pin_in_slot * batch, hipHostMallocDefault));
    // (allocated here, not at first use: hipHostMalloc waits for the device, i.e. for another solver object's running solve)
    HC(hipHostMalloc((void**)&s->pin_jobs, sizeof(WarpJob) * batch, hipHostMallocDefault));
    HC(hipHostMalloc((void**)&s->pin_err, 64, hipHostMallocDefault));
    if (st->own_stream) {           // the asynchronous use (ArapFlow_UseOwnStream first): downloads will be asked for
        s->pin_out_slot = align_up(12 * (size_t)s->N, 256);
        HC(hipHostMalloc((void**)&s->pin_out, s->pin_out_slot * batch, hipHostMallocDefault));
    }
    s->has_rgb.assign(batch, 0);
    s->nactive.assign(batch, 0);
    const float wfit = sqrtf(100.0f), wreg = sqrtf(0.01f);   // CombinedSolver.h:173-177
    for (unsigned b = 0; b < batch; ++b) {
        Slot& sl = s->plan->hslots[b];
        const FrameDev& f = s->hfr[b];
        sl.O = f.O; sl.A = f.A; sl.U = f.U; sl.C = f.C; sl.M = f.M;
        sl.wf = wfit; sl.wr = wreg;
    }
    return s;
}

void ArapFlow_SolverFree(ArapFlow_Solver* s)
{
    if (!s) return;
    if (s->inflight) (void)ArapFlow_SolverWait(s);
    (void)hipStreamSynchronize(s->copy);
    plan_free(s->plan);
    (void)hipStreamDestroy(s->copy);
    (void)hipEventDestroy(s->ev_up); (void)hipEventDestroy(s->ev_done); (void)hipEventDestroy(s->ev_dl);
    if (s->pin_in) (void)hipHostFree(s->pin_in);
    if (s->pin_out) (void)hipHostFree(s->pin_out);
    if (s->pin_jobs) (void)hipHostFree(s->pin_jobs);
    if (s->pin_err) (void)hipHostFree(s->pin_err);
    (void)hipFree(s->block);
    delete s;
}

int ArapFlow_SolverSetFrame(ArapFlow_Solver* s, unsigned slot, const uint8_t* rgb, const uint8_t* mask_red,
                            const int32_t* cons, unsigned ncons, int add_border_pins)
{
    if (!s || slot >= (unsigned)s->batch || !mask_red || (ncons && !cons)) return -1;
    // the previous solve of THIS solver may still read the slot's images and tile lists
    if (s->inflight) { const int rc = ArapFlow_SolverWait(s); if (rc != 0) return rc; }     // (-2: the retry failed too)
    const int W = s->W, H = s->H;
    const size_t N = s->N;
    // the staging of this slot may still be the source of an earlier upload
    HC(hipStreamSynchronize(s->copy));
    char* stage = s->pin_in + s->pin_in_slot * slot;
    float2* T = (float2*)stage;
    uint8_t* smask = (uint8_t*)(stage + 8 * N);
    uint8_t* srgb = (uint8_t*)(stage + 9 * N);
    // host pre-pass of setConstraintImage's placement loop (CombinedSolver.h:230-240): file
    // constraints first, then border pins (main.cpp:130-136); later entries overwrite earlier ones;
    // only where Mask == 0.
    const float2 none = make_float2(NAN, NAN);
    for (size_t i = 0; i < N; ++i) T[i] = none;
    auto place = [&](int x, int y, int tx, int ty) {
        if (x < 0 || x >= W || y < 0 || y >= H) return;
        if (mask_red[x + (size_t)W * y] == 0) T[x + (size_t)W * y] = make_float2((float)tx, (float)ty);
    };
    for (unsigned k = 0; k < ncons; ++k) place(cons[4 * k], cons[4 * k + 1], cons[4 * k + 2], cons[4 * k + 3]);
    if (add_border_pins) {
        for (int x = 0; x < W; ++x) place(x, 0, x, 0);
        for (int y = 1; y + 1 < H; ++y) { place(0, y, 0, y); if (W > 1) place(W - 1, y, W - 1, y); }
        if (H > 1) for (int x = 0; x < W; ++x) place(x, H - 1, x, H - 1);
    }
    memcpy(smask, mask_red, N);
    if (rgb) memcpy(srgb, rgb, 3 * N);
    // active vertices and the resident kernel's work list of this frame (aligned 32x8 tiles, band by band)
    std::vector<int> tiles, bandx0;
    uint64_t na = 0;
    build_resident_tiles(mask_red, W, H, true, tiles, bandx0, &na);
    s->nactive[slot] = na;
    plan_upload_tiles(s->plan, (int)slot, tiles, bandx0, s->copy);
    {
        // active 64x4 tiles (list launches of the per-step kernels).  Those kernels then rewrite flags / tile activity
        // inside the listed tiles only, so what an earlier frame left in this slot is cleared here.
        Opt_Plan* p = s->plan;
        const int tX = p->pd.tilesX, tY = p->pd.tilesY;
        std::vector<int>& l64 = p->h_t64[slot];
        l64.clear();
        for (int ty = 0; ty < tY; ++ty)
            for (int tx = 0; tx < tX; ++tx) {
                bool any = false;
                for (int y = ty * TILE_Y; y < H && y < (ty + 1) * TILE_Y && !any; ++y) {
                    const uint8_t* row = mask_red + (size_t)W * y;
                    for (int x = tx * TILE_X; x < W && x < (tx + 1) * TILE_X; ++x)
                        if (row[x] == 0) { any = true; break; }
                }
                if (any) l64.push_back(ty * tX + tx);
            }
        p->h_t64n[slot] = (int)l64.size();
        const size_t T = (size_t)tX * tY;
        if (!l64.empty())
            HC(hipMemcpyAsync(p->d_t64list + slot * T, l64.data(), l64.size() * sizeof(int), hipMemcpyHostToDevice, s->copy));
        HC(hipMemcpyAsync(p->d_t64n + slot, &p->h_t64n[slot], sizeof(int), hipMemcpyHostToDevice, s->copy));
        HC(hipMemsetAsync(p->pd.flags + (size_t)slot * N, 0, N, s->copy));
        HC(hipMemsetAsync(p->pd.tileact + (size_t)slot * T, 0, T, s->copy));
    }
    const FrameDev& f = s->hfr[slot];
    HC(hipMemcpyAsync(f.T, T, N * sizeof(float2), hipMemcpyHostToDevice, s->copy));
    HC(hipMemcpyAsync(f.mask, smask, N, hipMemcpyHostToDevice, s->copy));
    if (rgb) HC(hipMemcpyAsync(f.rgb, srgb, 3 * N, hipMemcpyHostToDevice, s->copy));
    s->has_rgb[slot] = rgb ? 1 : 0;
    s->uploads_pending = true;
    return 0;
}

int ArapFlow_SolverSolveAsync(ArapFlow_Solver* s, unsigned nframes, unsigned numIter, unsigned nIterations,
                              unsigned lIterations, int warp, int download)
{
    if (!s || nframes == 0 || nframes > (unsigned)s->batch || numIter == 0) return -1;
    if (s->inflight && ArapFlow_SolverWait(s) != 0) return -1;
    Opt_State* st = s->st;
    HC(hipSetDevice(st->device));
    const bool paused = st->res_cooldown > 0;                 // this call runs on the two-kernel path: counts as one
    s->a_n = nframes; s->a_numIter = numIter; s->a_nIt = nIterations; s->a_lIt = lIterations;
    s->a_warp = warp; s->a_download = download;
    s->retried = false;
    s->launches_at_enqueue = s->plan->res_launches;
    solver_enqueue(s);
    if (paused) --st->res_cooldown;
    s->inflight = true;
    return 0;
}

int ArapFlow_SolverWait(ArapFlow_Solver* s)
{
    if (!s) return -1;
    if (!s->inflight) return 0;
    Opt_State* st = s->st;
    Opt_Plan* p = s->plan;
    HC(hipEventSynchronize(s->ev_done));
    // The resident path needs all its workgroups co-resident; if a launch gave up (GPU shared with another process) the
    // device skipped every later update: redo the whole schedule once, now on the two-kernel path (plan_resident_failed
    // pauses the resident path), from the reset.
    // (The error word came back with the solve, in stream order, into pinned memory: reading it through the compute
    //  stream here would wait for whatever ANOTHER solver object has enqueued there since -- with two alternating solver
    //  objects, for the other one's whole solve.  Only a non-zero word takes the blocking path.)
    if (p->res_launches > 0 && s->pin_err && *s->pin_err != 0u && plan_resident_failed(p)) {
        HC(hipStreamSynchronize(s->copy));
        solver_enqueue(s);
        s->retried = true;
        HC(hipEventSynchronize(s->ev_done));
        if (plan_resident_failed(p)) {
            fprintf(stderr, "arapopt: the two-kernel retry reported a resident failure\n");
            s->inflight = false;
            return -2;
        }
    } else if (p->res_launches != s->launches_at_enqueue) {
        st->res_backoff = 8;                                  // a CHECKED resident success (this call launched the kernel)
    }
    if (s->a_download) HC(hipEventSynchronize(s->ev_dl));
    s->inflight = false;
    return 0;
}

int ArapFlow_SolverSolve(ArapFlow_Solver* s, unsigned nframes, unsigned numIter, unsigned nIterations,
                         unsigned lIterations)
{
    const int rc = ArapFlow_SolverSolveAsync(s, nframes, numIter, nIterations, lIterations, 0, 0);
    return rc != 0 ? rc : ArapFlow_SolverWait(s);
}

int ArapFlow_SolverWarp(ArapFlow_Solver* s, unsigned nframes)
{
    if (!s || nframes == 0 || nframes > (unsigned)s->batch) return -1;
    if (s->inflight) { const int rc = ArapFlow_SolverWait(s); if (rc != 0) return rc; }
    solver_enqueue_warp(s, nframes);
    // the rasteriser reads the slots' mask / rgb and rewrites their outputs: every later call on this solver that touches
    // them (SetFrame, GetResults, ...) waits for it like for a solve
    HC(hipEventRecord(s->ev_done, s->st->stream));
    s->launches_at_enqueue = s->plan->res_launches;
    s->inflight = true;
    return 0;
}

int ArapFlow_SolverHostResults(ArapFlow_Solver* s, unsigned slot, const float** flow, const uint8_t** warped_rgb,
                               const uint8_t** warped_mask)
{
    if (!s || slot >= (unsigned)s->batch || !s->pin_out || !s->a_download || slot >= s->a_n) return -1;
    if (s->inflight && ArapFlow_SolverWait(s) != 0) return -1;
    const size_t N = s->N;
    const char* o = s->pin_out + s->pin_out_slot * slot;
    if (flow) *flow = (const float*)o;
    if (warped_rgb) *warped_rgb = s->has_rgb[slot] ? (const uint8_t*)(o + 8 * N) : nullptr;
    if (warped_mask) *warped_mask = (const uint8_t*)(o + 11 * N);
    return 0;
}

int ArapFlow_SolverGetResults(ArapFlow_Solver* s, unsigned slot, float* flow, uint8_t* warped_rgb,
                              uint8_t* warped_mask, float* offset, float* angle, double* final_cost)
{
    if (!s || slot >= (unsigned)s->batch) return -1;
    if (s->inflight && ArapFlow_SolverWait(s) != 0) return -1;
    HC(hipStreamSynchronize(s->st->stream));
    plan_check_resident_error(s->plan);
    const FrameDev& f = s->hfr[slot];
    const size_t N = s->N;
    hipStream_t cs = s->copy;
    if (flow) HC(hipMemcpyAsync(flow, f.flow, N * sizeof(float2), hipMemcpyDeviceToHost, cs));
    if (warped_rgb) HC(hipMemcpyAsync(warped_rgb, f.out_rgb, 3 * N, hipMemcpyDeviceToHost, cs));
    if (warped_mask) HC(hipMemcpyAsync(warped_mask, f.out_mask, N, hipMemcpyDeviceToHost, cs));
    if (offset) HC(hipMemcpyAsync(offset, f.O, N * sizeof(float2), hipMemcpyDeviceToHost, cs));
    if (angle) HC(hipMemcpyAsync(angle, f.A, N * sizeof(float), hipMemcpyDeviceToHost, cs));
    HC(hipStreamSynchronize(cs));
    if (final_cost) *final_cost = plan_read_cost(s->plan, (int)slot, s->last_cost_index);
    return 0;
}

int ArapFlow_SolverStats(ArapFlow_Solver* s, uint64_t* pcg, uint64_t* active, uint64_t* grid)
{
    if (!s) return -1;
    if (pcg) *pcg = s->last_pcg;
    if (active) *active = s->last_active;
    if (grid) *grid = s->last_grid;
    return 0;
}

uint64_t ArapFlow_SolverResidentLaunches(ArapFlow_Solver* s) { return s ? s->plan->res_launches : 0; }
uint64_t ArapFlow_PlanResidentLaunches(Opt_Plan* plan) { return plan ? plan->res_launches : 0; }
int ArapFlow_SolverLaunchesFor(ArapFlow_Solver* s, unsigned nframes)
{
    if (!s || nframes == 0 || nframes > (unsigned)s->plan->batch) return -1;
    Opt_Plan* p = s->plan;
    const int keep = p->nb;
    p->nb = (int)nframes;                              // eligibility looks at the first nb slots
    const int sets = plan_resident_eligible(p) ? resident_deal(p->h_ntiles.data(), (int)nframes, nullptr, nullptr) : 0;
    p->nb = keep;
    return sets;
}
int ArapFlow_ResidentDeal(const int* active_tiles, unsigned nsolves, int* table, unsigned table_launches)
{
    if (!active_tiles || nsolves == 0) return -1;
    for (unsigned b = 0; b < nsolves; ++b)
        if (active_tiles[b] < 0 || active_tiles[b] > RES_MAX_TILES) return -1;
    std::vector<ResWg> map;
    const int sets = resident_deal(active_tiles, (int)nsolves, &map, nullptr);
    if (table)
        for (size_t i = 0; i < map.size() && i < (size_t)table_launches * RES_WGS; ++i) {
            table[4 * i + 0] = map[i].slot; table[4 * i + 1] = map[i].rank;
            table[4 * i + 2] = map[i].wgs; table[4 * i + 3] = map[i].gran;
        }
    return sets;
}
int ArapFlow_ResidentTiles(const uint8_t* mask_red, unsigned W, unsigned H, int aligned, int* origins, unsigned cap,
                           int* bandx0)
{
    if (!mask_red || W == 0 || H == 0) return -1;
    std::vector<int> tiles, bx;
    build_resident_tiles(mask_red, (int)W, (int)H, aligned != 0, tiles, bx, nullptr);
    if (origins)
        for (size_t i = 0; i < tiles.size() && i < cap; ++i) origins[i] = tiles[i];
    if (bandx0)
        for (size_t i = 0; i < bx.size(); ++i) bandx0[i] = bx[i];
    return (int)tiles.size();
}
int ArapFlow_SolverResidentLayout(ArapFlow_Solver* s, int* launches_per_step, int* solves_in_flight)
{
    if (!s) return -1;
    const bool res = plan_resident_eligible(s->plan) && s->plan->res_sets > 0;
    if (launches_per_step) *launches_per_step = res ? s->plan->res_sets : 0;
    if (solves_in_flight) *solves_in_flight = res ? s->plan->res_inflight : 0;
    return 0;
}
int ArapFlow_ResidentFailed(Opt_State* state) { return state && state->resident_failed ? 1 : 0; }
int ArapFlow_SolverLeanStream(ArapFlow_Solver* s) { return s && plan_lean_stream(s->plan) ? 1 : 0; }

// diagnostic (ARAPOPT_STAMPS=1): copy the [256][8] phase-time table of the LAST resident launch
int ArapFlow_SolverStamps(ArapFlow_Solver* s, uint64_t* out)
{
    if (!s || !s->plan->rd.stamps) return -1;
    HC(hipStreamSynchronize(s->st->stream));
    HC(hipMemcpy(out, s->plan->rd.stamps, RES_WGS * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

uint64_t ArapFlow_WarpScratchBytes(unsigned W, unsigned H)
{
    return align_up((uint64_t)W * H * 8, 256) + 256;
}

int ArapFlow_Warp(Opt_State* st, unsigned W, unsigned H, const void* rgb, const void* mask_red, const void* flow,
                  void* out_rgb, void* out_mask, void* scratch)
{
    if (!st || !mask_red || !flow || !out_mask || !scratch) return -1;
    const size_t N = (size_t)W * H;
    WarpJob j{};
    j.field = nullptr; j.flow_in = (const float2*)flow;
    j.rgb = (const uint8_t*)rgb; j.mask = (const uint8_t*)mask_red;
    j.flow_out = nullptr;
    j.key = (unsigned long long*)scratch;
    j.out_rgb = (uint8_t*)out_rgb; j.out_mask = (uint8_t*)out_mask;
    WarpJob* dj = (WarpJob*)((char*)scratch + align_up(N * 8, 256));
    HC(hipMemsetAsync(scratch, 0, N * 8, st->stream));
    HC(hipMemcpyAsync(dj, &j, sizeof(j), hipMemcpyHostToDevice, st->stream));
    hipLaunchKernelGGL(k_warp_raster, dim3((W + 63) / 64, (H + 3) / 4, 1), dim3(64, 4), 0, st->stream, dj, (int)W,
                       (int)H);
    hipLaunchKernelGGL(k_warp_resolve, dim3((unsigned)((N + 255) / 256), 1, 1), dim3(256), 0, st->stream, dj,
                       (int)N);
    return (int)hipGetLastError();
}

}  // extern "C"
