// arap_stream.h -- the streaming (state in HBM / Infinity Cache) PCG kernels of the FRAME SOLVER's two-kernel path:
// what runs when a solve has more active tiles than the resident kernel holds (1920x1080 with every vertex active:
// 8100 tiles) or when the resident path pauses after a timed-out launch.
//
// Same arithmetic as k_pcg_a / k_pcg_b (arap_kernels.h), operation for operation; what differs is the traffic:
//   * k_pcg_a_grid: phase A specialised to the pixel-grid UrShape the frame solver always has (CombinedSolver.h:207-221:
//     d = U(c) - U(n) = -s, no UrShape loads: 8 B + 4 cached neighbour loads per vertex less), the new direction and
//     cos/sin staged in LDS with a one-vertex halo, and an XCD-AWARE TILE ORDER: workgroups are dealt round-robin to
//     the 8 XCDs, whose L2s do not share lines, so with the plain blockIdx -> tile map the two tiles either side of a
//     tile boundary sit on different XCDs and every halo row is fetched through the fabric a second time (measured at
//     1920x1080, mask == 0: FETCH_SIZE x 2 = 155 MB against 85 MB algorithmic).  Here XCD j works through the j-th
//     eighth of the tile list, top to bottom: vertically adjacent tiles share an L2 and are in flight together.
//   * k_pcg_b4_lean: phase B with 16-byte accesses that reads neither z (it is only written; read back only for a quad
//     with an excluded vertex, whose z must survive) nor the Offset preconditioner (a function of the vertex's degree
//     and fit flag: a 10-entry table, as in the resident kernel): 53 B read + 36 B written per vertex instead of 73 + 36.
#pragma once
#include "arap_kernels.h"

namespace arap {

// tile (tx, ty) and frame of a workgroup of a 1-D launch of nb * 8 * chunk blocks, chunk = ceil(tiles / 8)
__device__ __forceinline__ bool xcd_tile(int tilesX, int tilesY, int chunk, int& tx, int& ty, int& b, unsigned& lb)
{
    const int bid = blockIdx.x;
    b = bid / (8 * chunk);
    const int r = bid - b * 8 * chunk;
    lb = (unsigned)r;                                 // linear index among the frame's 8 * chunk workgroups
    const int tile = (r & 7) * chunk + (r >> 3);
    if (tile >= tilesX * tilesY) return false;
    ty = tile / tilesX;
    tx = tile - ty * tilesX;
    return true;
}

template <int TX, int TY>
__global__ __launch_bounds__(TX* TY) void k_pcg_a_grid(PlanDev pd, int l, int tilesX, int tilesY, int chunk)
{
    constexpr int LW = TX + 2, LH = TY + 2;
    __shared__ float2 sP[LH * LW];
    __shared__ float2 sC[LH * LW];
    __shared__ float sA[LH * LW];
    int btx, bty, b;
    unsigned lb;
    const bool has_tile = xcd_tile(tilesX, tilesY, chunk, btx, bty, b, lb);
    const unsigned nlb = 8u * (unsigned)chunk;
    double* const sigma_l = pd.red + ((size_t)b * pd.nslots + (2 * l + 1)) * NSHARD;
    if (!has_tile) { block_reduce_fixed<1>(pd, b, lb, nlb, 0.0, 0.0, sigma_l, nullptr); return; }
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = btx * TX + tx, y = bty * TY + ty;
    const int W = pd.W, H = pd.H;
    const bool in = x < W && y < H;
    const int i = x + W * y;
    const size_t gb = (size_t)b * pd.N;
    const unsigned f = in ? pd.flags[gb + i] : 0u;
    if (!__syncthreads_or((int)(f & F_ACT))) {                   // nothing active in this tile
        block_reduce_fixed<1>(pd, b, lb, nlb, 0.0, 0.0, sigma_l, nullptr);
        return;
    }
    const Slot sl = pd.slots[b];
    const float2* __restrict__ pinO = (l & 1) ? pd.pO1 : pd.pO0;
    const float* __restrict__ pinA = (l & 1) ? pd.pA1 : pd.pA0;
    float2* __restrict__ poutO = (l & 1) ? pd.pO0 : pd.pO1;
    float* __restrict__ poutA = (l & 1) ? pd.pA0 : pd.pA1;
    float beta = 0.f;
    if (l > 0) {
        const double* rs = pd.red + (size_t)b * pd.nslots * NSHARD;
        const float rhoNew = read_scalar(rs + (size_t)(2 * l) * NSHARD);
        const float rhoOld = read_scalar(rs + (size_t)(2 * l - 2) * NSHARD);
        if (rhoOld > 0.f) beta = rhoNew / rhoOld;
    }
    // p_l of vertex j (any in-image vertex; values at excluded vertices are never used)
    auto stage = [&](int j, int cell) {
        float2 pO = pinO[gb + j];
        float pA = pinA[gb + j];
        if (l > 0) {
            const float2 zO = pd.zO[gb + j];
            const float zA = pd.zA[gb + j];
            pO.x = fmaf(beta, pO.x, zO.x);
            pO.y = fmaf(beta, pO.y, zO.y);
            pA = fmaf(beta, pA, zA);
        }
        sP[cell] = pO;
        sA[cell] = pA;
        sC[cell] = pd.cs[gb + j];
        return make_float4(pO.x, pO.y, pA, 0.f);
    };
    const int cell = (ty + 1) * LW + (tx + 1);
    float4 own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) {
        own = stage(i, cell);
        if (f & F_ACT) { poutO[gb + i] = make_float2(own.x, own.y); poutA[gb + i] = own.z; }
        if (ty == 0 && y > 0) stage(i - W, cell - LW);
        if ((ty == TY - 1 || y == H - 1) && y + 1 < H) stage(i + W, cell + LW);
        if (tx == 0 && x > 0) stage(i - 1, cell - 1);
        if ((tx == TX - 1 || x == W - 1) && x + 1 < W) stage(i + 1, cell + 1);
    }
    __syncthreads();
    double d = 0.0;
    if (f & F_ACT) {
        const float wr2 = sl.wr * sl.wr;
        const float2 pO = make_float2(own.x, own.y);
        const float pA = own.z;
        const float2 csi = sC[cell];
        const float ci = csi.x, si = csi.y;
        float ax = 0.f, ay = 0.f, aa = 0.f;
        // k_pcg_a's edge block with d = U(c) - U(n) = -s put in: q = R'(A(c)) d and h = R'(A(n)) d are signed copies of
        // (si, ci) / (sn, cn) -- a product with -1 / 0 / 1 and the addition of a zero are exact, so every value equals
        // the generic kernel's (only the sign of an exact zero may differ), as in the resident kernel
#define STREAM_EDGE(BIT, DC, NQX, NQY, NHX, NHY, QX, QY)                                            \
        if (f & (BIT)) {                                                                            \
            const int nc = cell + (DC);                                                             \
            const float2 qO = sP[nc];                                                               \
            const float qA = sA[nc];                                                                \
            const float2 csn = sC[nc];                                                              \
            const float cn = csn.x, sn = csn.y;                                                     \
            const float px = pO.x - qO.x, py = pO.y - qO.y;                                         \
            const float tx_ = fmaf(NQX, pA, px), ty_ = fmaf(NQY, pA, py);                           \
            ax = fmaf(wr2, fmaf(NHX, qA, px + tx_), ax);                                            \
            ay = fmaf(wr2, fmaf(NHY, qA, py + ty_), ay);                                            \
            aa = fmaf(-wr2, fmaf(QX, tx_, (QY) * ty_), aa);                                         \
            (void)cn; (void)sn;                                                                     \
        }
        //          bit   cell      -q          -h          q
        STREAM_EDGE(F_E0, 1,      -si,  ci,   -sn,  cn,    si, -ci)      // s=( 1, 0): q=( si,-ci) h=( sn,-cn)
        STREAM_EDGE(F_E1, -1,      si, -ci,    sn, -cn,   -si,  ci)      // s=(-1, 0): q=(-si, ci) h=(-sn, cn)
        STREAM_EDGE(F_E2, LW,     -ci, -si,   -cn, -sn,    ci,  si)      // s=( 0, 1): q=( ci, si) h=( cn, sn)
        STREAM_EDGE(F_E3, -LW,     ci,  si,    cn,  sn,   -ci, -si)      // s=( 0,-1): q=(-ci,-si) h=(-cn,-sn)
#undef STREAM_EDGE
        if (f & F_FIT) {
            const float wf2 = sl.wf * sl.wf;
            ax = fmaf(wf2, pO.x, ax);
            ay = fmaf(wf2, pO.y, ay);
        }
        pd.ApO[gb + i] = make_float2(ax, ay);
        pd.ApA[gb + i] = aa;
        d = (double)dot3(pO.x, pO.y, pA, ax, ay, aa);
    }
    block_reduce_fixed<1>(pd, b, lb, nlb, d, 0.0, sigma_l, nullptr);
}

// Phase A as a MARCH down a 64-column strip: a workgroup of 4 wavefronts (one row of 64 vertices each) owns RB
// consecutive 4-row blocks of one strip.  A ring of four blocks in LDS holds the new direction and cos/sin; in step k
// the loads of block k+2 are issued, block k is computed from LDS (its upper neighbours are the last row of block k-1,
// the lower ones the first row of block k+1: consecutive ring rows) and block k+2 is then written to the ring.  Every
// vertex is fetched once (plus one halo row above and below the RB blocks, and the two halo columns), the global loads
// of the next block fly while the current one is computed, and the phase's dot product costs one atomic per workgroup.
// Blocks whose 64x4 tile holds no active vertex (pd.tileact, rebuilt by k_gn_prep) are neither loaded nor computed.
template <int RB>
__global__ __launch_bounds__(256) void k_pcg_a_march(PlanDev pd, int l, int stripsX, int chunksY, int chunk8)
{
    constexpr int LW = TILE_X + 2, RROWS = 16;                    // ring: 4 blocks x 4 rows
    __shared__ float2 sP[RROWS][LW];
    __shared__ float2 sC[RROWS][LW];
    __shared__ float sA[RROWS][LW];
    __shared__ unsigned char sF[RROWS][TILE_X];
    int sx, cy, b;
    unsigned lb;
    const bool has_strip = xcd_tile(stripsX, chunksY, chunk8, sx, cy, b, lb);
    const unsigned nlb = 8u * (unsigned)chunk8;
    double* const sigma_l = pd.red + ((size_t)b * pd.nslots + (2 * l + 1)) * NSHARD;
    // the tag of this launch's granules (order-fixed sum at the end): fetched now, used after the march
    const unsigned rtag = red_tag(pd, b, lb);
    if (!has_strip) { block_reduce_fixed<1>(pd, b, lb, nlb, 0.0, 0.0, sigma_l, nullptr, rtag); return; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int W = pd.W, H = pd.H;
    const size_t gb = (size_t)b * pd.N;
    const int x = sx * TILE_X + lane;
    const int ybase = cy * (4 * RB);
    const int nblk = min(RB, (H - ybase + 3) >> 2);               // blocks this workgroup owns
    const uint8_t* tact = pd.tileact + (size_t)b * pd.tilesX * pd.tilesY + sx;
    const int tyb = ybase >> 2;                                   // tile row of block 0
    auto active = [&](int blk) {                                  // does block blk (-1 .. nblk) hold an active vertex?
        const int ty = tyb + blk;
        return ty >= 0 && ty < pd.tilesY && tact[(size_t)ty * pd.tilesX] != 0;
    };
    // anything to do at all?  (uniform: tileact is per tile)
    {
        bool any = false;
        for (int k = 0; k < nblk; ++k) any = any || active(k);
        if (!any) { block_reduce_fixed<1>(pd, b, lb, nlb, 0.0, 0.0, sigma_l, nullptr, rtag); return; }
    }
    const Slot sl = pd.slots[b];
    const float2* __restrict__ pinO = (l & 1) ? pd.pO1 : pd.pO0;
    const float* __restrict__ pinA = (l & 1) ? pd.pA1 : pd.pA0;
    float2* __restrict__ poutO = (l & 1) ? pd.pO0 : pd.pO1;
    float* __restrict__ poutA = (l & 1) ? pd.pA0 : pd.pA1;
    float beta = 0.f;
    if (l > 0) {
        const double* rs = pd.red + (size_t)b * pd.nslots * NSHARD;
        const float rhoNew = read_scalar(rs + (size_t)(2 * l) * NSHARD);
        const float rhoOld = read_scalar(rs + (size_t)(2 * l - 2) * NSHARD);
        if (rhoOld > 0.f) beta = rhoNew / rhoOld;
    }
    // ---- staging of one block row per wavefront: loads (registers) ... later: p_l = z + beta p_{l-1} -> ring --------
    struct Stage {
        float2 pO, zO, cs, hpO, hzO, hcs;
        float pA, zA, hpA, hzA;
        unsigned f;
        int i, hcol;            // own vertex index (-1: nothing to stage), halo column cell (-1: none)
        int hi;
        bool owned;
    };
    auto issue = [&](int blk) {
        Stage s;
        s.i = -1; s.hcol = -1; s.hi = -1; s.f = 0u; s.owned = blk >= 0 && blk < nblk;
        s.pO = s.zO = s.cs = s.hpO = s.hzO = s.hcs = make_float2(0.f, 0.f);
        s.pA = s.zA = s.hpA = s.hzA = 0.f;
        const int y = ybase + 4 * blk + w;
        // block -1 contributes its last row only (halo above), block nblk its first row only (halo below)
        const bool row_wanted = blk <= nblk && (blk >= 0 || w == 3) && (blk < nblk || w == 0);
        if (!row_wanted || y < 0 || y >= H || !active(blk)) return s;
        if (x < W) {
            s.i = x + W * y;
            s.f = pd.flags[gb + s.i];
            s.pO = pinO[gb + s.i]; s.pA = pinA[gb + s.i]; s.cs = pd.cs[gb + s.i];
            if (l > 0) { s.zO = pd.zO[gb + s.i]; s.zA = pd.zA[gb + s.i]; }
            // halo columns: the strip's left neighbour column by lane 0, the right one by the last in-image lane
            if (lane == 0 && x > 0) { s.hi = s.i - 1; s.hcol = 0; }
            if ((lane == TILE_X - 1 || x == W - 1) && x + 1 < W) { s.hi = s.i + 1; s.hcol = lane + 2; }
            if (s.hi >= 0) {
                s.hpO = pinO[gb + s.hi]; s.hpA = pinA[gb + s.hi]; s.hcs = pd.cs[gb + s.hi];
                if (l > 0) { s.hzO = pd.zO[gb + s.hi]; s.hzA = pd.zA[gb + s.hi]; }
            }
        }
        return s;
    };
    auto finish = [&](int blk, const Stage& s) {
        if (s.i < 0) return;
        const int r = ((blk & 3) << 2) | w;
        float2 pO = s.pO;
        float pA = s.pA;
        if (l > 0) {
            pO.x = fmaf(beta, pO.x, s.zO.x);
            pO.y = fmaf(beta, pO.y, s.zO.y);
            pA = fmaf(beta, pA, s.zA);
        }
        sP[r][lane + 1] = pO; sA[r][lane + 1] = pA; sC[r][lane + 1] = s.cs;
        sF[r][lane] = (unsigned char)s.f;
        if (s.owned && (s.f & F_ACT)) { poutO[gb + s.i] = pO; poutA[gb + s.i] = pA; }
        if (s.hcol >= 0) {
            float2 hO = s.hpO;
            float hA = s.hpA;
            if (l > 0) {
                hO.x = fmaf(beta, hO.x, s.hzO.x);
                hO.y = fmaf(beta, hO.y, s.hzO.y);
                hA = fmaf(beta, hA, s.hzA);
            }
            sP[r][s.hcol] = hO; sA[r][s.hcol] = hA; sC[r][s.hcol] = s.hcs;
        }
    };
    // ---- prologue: halo row above, blocks 0 and 1 -------------------------------------------------------------------
    {
        const Stage a = issue(-1), c0 = issue(0), c1 = issue(1);
        finish(-1, a); finish(0, c0); finish(1, c1);
    }
    __syncthreads();
    const float wr2 = sl.wr * sl.wr, wf2 = sl.wf * sl.wf;
    double d = 0.0;
    // A block's Ap is stored one block late, the last block's after the workgroup's ticket has been taken (red_arrive:
    // a ticket behind streaming stores comes back only when they have drained)
    float2 pendO = make_float2(0.f, 0.f);
    float pendA = 0.f;
    int pendI = -1;
    for (int k = 0; k < nblk; ++k) {
        const Stage nx = issue(k + 2);                            // (block nblk: the halo row below; beyond: nothing)
        if (pendI >= 0) { pd.ApO[gb + pendI] = pendO; pd.ApA[gb + pendI] = pendA; pendI = -1; }
        const int y = ybase + 4 * k + w;
        if (active(k) && x < W && y < H) {
            const int r = ((k & 3) << 2) | w, ru = (r + RROWS - 1) & (RROWS - 1), rd = (r + 1) & (RROWS - 1);
            const unsigned f = sF[r][lane];
            if (f & F_ACT) {
                const int c = lane + 1;
                const float2 pO = sP[r][c];
                const float pA = sA[r][c];
                const float2 csi = sC[r][c];
                const float ci = csi.x, si = csi.y;
                float ax = 0.f, ay = 0.f, aa = 0.f;
#define MARCH_EDGE(BIT, RR, CC, NQX, NQY, NHX, NHY, QX, QY)                                         \
                if (f & (BIT)) {                                                                    \
                    const float2 qO = sP[RR][CC];                                                   \
                    const float qA = sA[RR][CC];                                                    \
                    const float2 csn = sC[RR][CC];                                                  \
                    const float cn = csn.x, sn = csn.y;                                             \
                    const float px = pO.x - qO.x, py = pO.y - qO.y;                                 \
                    const float tx_ = fmaf(NQX, pA, px), ty_ = fmaf(NQY, pA, py);                   \
                    ax = fmaf(wr2, fmaf(NHX, qA, px + tx_), ax);                                    \
                    ay = fmaf(wr2, fmaf(NHY, qA, py + ty_), ay);                                    \
                    aa = fmaf(-wr2, fmaf(QX, tx_, (QY) * ty_), aa);                                 \
                    (void)cn; (void)sn;                                                             \
                }
                MARCH_EDGE(F_E0, r, c + 1,    -si,  ci,   -sn,  cn,    si, -ci)      // s=( 1, 0)
                MARCH_EDGE(F_E1, r, c - 1,     si, -ci,    sn, -cn,   -si,  ci)      // s=(-1, 0)
                MARCH_EDGE(F_E2, rd, c,       -ci, -si,   -cn, -sn,    ci,  si)      // s=( 0, 1)
                MARCH_EDGE(F_E3, ru, c,        ci,  si,    cn,  sn,   -ci, -si)      // s=( 0,-1)
#undef MARCH_EDGE
                if (f & F_FIT) {
                    ax = fmaf(wf2, pO.x, ax);
                    ay = fmaf(wf2, pO.y, ay);
                }
                pendI = x + W * y;
                pendO = make_float2(ax, ay);
                pendA = aa;
                d += (double)dot3(pO.x, pO.y, pA, ax, ay, aa);
            }
        }
        finish(k + 2, nx);
        __syncthreads();
    }
    const RedTicket rt = red_arrive<1>(pd, b, lb, nlb, d, 0.0, rtag);
    if (pendI >= 0) { pd.ApO[gb + pendI] = pendO; pd.ApA[gb + pendI] = pendA; }
    red_finish<1>(pd, b, rt, sigma_l, nullptr);
}

// ---- the lean schedule (frame-solver plans): 126 instead of 146 bytes per vertex and iteration ------------------------
// What the two phases of an iteration must do is fixed by the two sums (sigma = p.Ap needs every p, rho' = z.r needs the
// new r everywhere); WHERE the element-wise work is done is free.  k_pcg_a_march2 / k_pcg_b4_r move it so that fewer
// bytes travel (same operations on the same operands: the bits do not change):
//   * z = M^-1 r is never stored: phase B needs it only for its dot product, and phase A forms it again from r, M^-1_A
//     and the flag byte (M^-1_O is a function of the flags) for every vertex it stages      (-12 B written, +4 B read)
//   * delta += alpha p of iteration l-1 is done by phase A of iteration l, which has p_{l-1} in hand anyway; phase B no
//     longer reads p or touches delta (the last iteration's update is folded into k_gn_update)     (-12 B read)
// Phase A': reads p3 r3 M^-1_A cs2 flags delta3, writes p3 Ap3 delta3 (85 B); phase B': reads r3 Ap3 M^-1_A flags, writes
// r3 (41 B).
#ifndef ARAP_MARCH2_WAVES
#define ARAP_MARCH2_WAVES 1
#endif
template <int RB>
__global__ __launch_bounds__(256, ARAP_MARCH2_WAVES) void k_pcg_a_march2(PlanDev pd, int l, int stripsX, int chunksY, int chunk8)
{
    constexpr int LW = TILE_X + 2, RROWS = 16;                    // ring: 4 blocks x 4 rows
    __shared__ float2 sP[RROWS][LW];
    __shared__ float2 sC[RROWS][LW];
    __shared__ float sA[RROWS][LW];
    __shared__ unsigned char sF[RROWS][TILE_X];
    __shared__ float moLUT[12];
    int sx, cy, b;
    unsigned lb;
    const bool has_strip = xcd_tile(stripsX, chunksY, chunk8, sx, cy, b, lb);
    const unsigned nlb = 8u * (unsigned)chunk8;
    double* const sigma_l = pd.red + ((size_t)b * pd.nslots + (2 * l + 1)) * NSHARD;
    const unsigned rtag = red_tag(pd, b, lb);
    if (!has_strip) { block_reduce_fixed<1>(pd, b, lb, nlb, 0.0, 0.0, sigma_l, nullptr, rtag); return; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int W = pd.W, H = pd.H;
    const size_t gb = (size_t)b * pd.N;
    const int x = sx * TILE_X + lane;
    const int ybase = cy * (4 * RB);
    const int nblk = min(RB, (H - ybase + 3) >> 2);
    const uint8_t* tact = pd.tileact + (size_t)b * pd.tilesX * pd.tilesY + sx;
    const int tyb = ybase >> 2;
    auto active = [&](int blk) {
        const int ty = tyb + blk;
        return ty >= 0 && ty < pd.tilesY && tact[(size_t)ty * pd.tilesX] != 0;
    };
    {
        bool any = false;
        for (int k = 0; k < nblk; ++k) any = any || active(k);
        if (!any) { block_reduce_fixed<1>(pd, b, lb, nlb, 0.0, 0.0, sigma_l, nullptr, rtag); return; }
    }
    const Slot sl = pd.slots[b];
    if (threadIdx.x < 10) {                                       // M^-1_O by (degree, fit), as k_gn_init computes it
        const int deg = threadIdx.x % 5, fit = threadIdx.x / 5;
        float dO = 0.f;
        for (int k = 0; k < deg; ++k) dO = dO + (sl.wr * sl.wr + sl.wr * sl.wr);
        if (fit) dO = fmaf(sl.wf, sl.wf, dO);
        moLUT[threadIdx.x] = ginv(dO);
    }
    const float2* __restrict__ pinO = (l & 1) ? pd.pO1 : pd.pO0;
    const float* __restrict__ pinA = (l & 1) ? pd.pA1 : pd.pA0;
    float2* __restrict__ poutO = (l & 1) ? pd.pO0 : pd.pO1;
    float* __restrict__ poutA = (l & 1) ? pd.pA0 : pd.pA1;
    float beta = 0.f, alpha_prev = 0.f;                           // (read AFTER the first stages' loads have been issued: below)
    struct Stage {
        float2 pO, rO, cs, hpO, hrO, hcs, dO;
        float pA, rA, mA, hpA, hrA, hmA, dA;
        unsigned f, hf;
        int i, hcol, hi;
        bool owned;
    };
    auto issue = [&](int blk) {
        Stage s;
        s.i = -1; s.hcol = -1; s.hi = -1; s.f = 0u; s.hf = 0u; s.owned = blk >= 0 && blk < nblk;
        s.pO = s.rO = s.cs = s.hpO = s.hrO = s.hcs = s.dO = make_float2(0.f, 0.f);
        s.pA = s.rA = s.mA = s.hpA = s.hrA = s.hmA = s.dA = 0.f;
        const int y = ybase + 4 * blk + w;
        const bool row_wanted = blk <= nblk && (blk >= 0 || w == 3) && (blk < nblk || w == 0);
        if (!row_wanted || y < 0 || y >= H || !active(blk)) return s;
        if (x < W) {
            s.i = x + W * y;
            s.f = pd.flags[gb + s.i];
            s.pO = pinO[gb + s.i]; s.pA = pinA[gb + s.i]; s.cs = pd.cs[gb + s.i];
            if (l > 0) {
                s.rO = pd.rO[gb + s.i]; s.rA = pd.rA[gb + s.i]; s.mA = pd.preA[gb + s.i];
                if (s.owned) { s.dO = pd.deltaO[gb + s.i]; s.dA = pd.deltaA[gb + s.i]; }
            }
            if (lane == 0 && x > 0) { s.hi = s.i - 1; s.hcol = 0; }
            if ((lane == TILE_X - 1 || x == W - 1) && x + 1 < W) { s.hi = s.i + 1; s.hcol = lane + 2; }
            if (s.hi >= 0) {
                s.hpO = pinO[gb + s.hi]; s.hpA = pinA[gb + s.hi]; s.hcs = pd.cs[gb + s.hi];
                if (l > 0) { s.hf = pd.flags[gb + s.hi]; s.hrO = pd.rO[gb + s.hi]; s.hrA = pd.rA[gb + s.hi]; s.hmA = pd.preA[gb + s.hi]; }
            }
        }
        return s;
    };
    auto finish = [&](int blk, const Stage& s) {
        if (s.i < 0) return;
        const int r = ((blk & 3) << 2) | w;
        float2 pO = s.pO;
        float pA = s.pA;
        if (l > 0) {
            if (s.owned && (s.f & F_ACT)) {                          // delta += alpha_{l-1} p_{l-1}
                pd.deltaO[gb + s.i] = make_float2(fmaf(alpha_prev, pO.x, s.dO.x), fmaf(alpha_prev, pO.y, s.dO.y));
                pd.deltaA[gb + s.i] = fmaf(alpha_prev, pA, s.dA);
            }
            const float mo = moLUT[__popc(s.f & 15u) + 5 * (int)((s.f >> 4) & 1u)];
            const float zx = mo * s.rO.x, zy = mo * s.rO.y, za = s.mA * s.rA;      // z = M^-1 r, as phase B formed it
            pO.x = fmaf(beta, pO.x, zx);
            pO.y = fmaf(beta, pO.y, zy);
            pA = fmaf(beta, pA, za);
        }
        sP[r][lane + 1] = pO; sA[r][lane + 1] = pA; sC[r][lane + 1] = s.cs;
        sF[r][lane] = (unsigned char)s.f;
        if (s.owned && (s.f & F_ACT)) { poutO[gb + s.i] = pO; poutA[gb + s.i] = pA; }
        if (s.hcol >= 0) {
            float2 hO = s.hpO;
            float hA = s.hpA;
            if (l > 0) {
                const float mo = moLUT[__popc(s.hf & 15u) + 5 * (int)((s.hf >> 4) & 1u)];
                hO.x = fmaf(beta, hO.x, mo * s.hrO.x);
                hO.y = fmaf(beta, hO.y, mo * s.hrO.y);
                hA = fmaf(beta, hA, s.hmA * s.hrA);
            }
            sP[r][s.hcol] = hO; sA[r][s.hcol] = hA; sC[r][s.hcol] = s.hcs;
        }
    };
    {
        // the loads of the first three stages go out first; the three scalars (two dependent round trips each: shards, then
        // nothing else) are fetched while they fly -- beta and alpha are needed only when a stage is finished
        const Stage a = issue(-1), c0 = issue(0), c1 = issue(1);
        if (l > 0) {
            const double* rs = pd.red + (size_t)b * pd.nslots * NSHARD;
            const float rhoNew = read_scalar(rs + (size_t)(2 * l) * NSHARD);
            const float rhoOld = read_scalar(rs + (size_t)(2 * l - 2) * NSHARD);
            const float sigOld = read_scalar(rs + (size_t)(2 * l - 1) * NSHARD);
            if (rhoOld > 0.f) beta = rhoNew / rhoOld;
            if (sigOld > 0.f) alpha_prev = rhoOld / sigOld;       // alpha of iteration l - 1 (PCGStep2 :446-489)
        }
        __syncthreads();                                          // moLUT
        finish(-1, a); finish(0, c0); finish(1, c1);
    }
    __syncthreads();
    const float wr2 = sl.wr * sl.wr, wf2 = sl.wf * sl.wf;
    double d = 0.0;
    float2 pendO = make_float2(0.f, 0.f);
    float pendA = 0.f;
    int pendI = -1;
    // (two stages of loads in flight -- block k + 3 issued while block k is computed -- measured no faster: 38.7 vs 38.3 us at
    //  1920x1080 mask == 0: the phase is not bound by the latency of a stage's loads)
    for (int k = 0; k < nblk; ++k) {
        const Stage nx = issue(k + 2);
        if (pendI >= 0) { pd.ApO[gb + pendI] = pendO; pd.ApA[gb + pendI] = pendA; pendI = -1; }
        const int y = ybase + 4 * k + w;
        if (active(k) && x < W && y < H) {
            const int r = ((k & 3) << 2) | w, ru = (r + RROWS - 1) & (RROWS - 1), rd = (r + 1) & (RROWS - 1);
            const unsigned f = sF[r][lane];
            if (f & F_ACT) {
                const int c = lane + 1;
                const float2 pO = sP[r][c];
                const float pA = sA[r][c];
                const float2 csi = sC[r][c];
                const float ci = csi.x, si = csi.y;
                float ax = 0.f, ay = 0.f, aa = 0.f;
#define MARCH_EDGE(BIT, RR, CC, NQX, NQY, NHX, NHY, QX, QY)                                         \
                if (f & (BIT)) {                                                                    \
                    const float2 qO = sP[RR][CC];                                                   \
                    const float qA = sA[RR][CC];                                                    \
                    const float2 csn = sC[RR][CC];                                                  \
                    const float cn = csn.x, sn = csn.y;                                             \
                    const float px = pO.x - qO.x, py = pO.y - qO.y;                                 \
                    const float tx_ = fmaf(NQX, pA, px), ty_ = fmaf(NQY, pA, py);                   \
                    ax = fmaf(wr2, fmaf(NHX, qA, px + tx_), ax);                                    \
                    ay = fmaf(wr2, fmaf(NHY, qA, py + ty_), ay);                                    \
                    aa = fmaf(-wr2, fmaf(QX, tx_, (QY) * ty_), aa);                                 \
                    (void)cn; (void)sn;                                                             \
                }
                MARCH_EDGE(F_E0, r, c + 1,    -si,  ci,   -sn,  cn,    si, -ci)      // s=( 1, 0)
                MARCH_EDGE(F_E1, r, c - 1,     si, -ci,    sn, -cn,   -si,  ci)      // s=(-1, 0)
                MARCH_EDGE(F_E2, rd, c,       -ci, -si,   -cn, -sn,    ci,  si)      // s=( 0, 1)
                MARCH_EDGE(F_E3, ru, c,        ci,  si,    cn,  sn,   -ci, -si)      // s=( 0,-1)
#undef MARCH_EDGE
                if (f & F_FIT) {
                    ax = fmaf(wf2, pO.x, ax);
                    ay = fmaf(wf2, pO.y, ay);
                }
                pendI = x + W * y;
                pendO = make_float2(ax, ay);
                pendA = aa;
                d += (double)dot3(pO.x, pO.y, pA, ax, ay, aa);
            }
        }
        finish(k + 2, nx);
        __syncthreads();
    }
    const RedTicket rt = red_arrive<1>(pd, b, lb, nlb, d, 0.0, rtag);
    if (pendI >= 0) { pd.ApO[gb + pendI] = pendO; pd.ApA[gb + pendI] = pendA; }
    red_finish<1>(pd, b, rt, sigma_l, nullptr);
}

// Phase B of the lean schedule: r -= alpha Ap; rho' = (M^-1 r) . r.  Four consecutive vertices per lane, 16-byte accesses.
__global__ __launch_bounds__(256) void k_pcg_b4_r(PlanDev pd, int l)
{
    __shared__ float moLUT[12];
    const int b = blockIdx.y;
    const unsigned rtag = red_tag(pd, b, blockIdx.x);
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int nq = pd.N >> 2;
    const size_t gb = (size_t)b * pd.N;
    {
        const Slot sl = pd.slots[b];
        if (threadIdx.x < 10) {
            const int deg = threadIdx.x % 5, fit = threadIdx.x / 5;
            float dO = 0.f;
            for (int k = 0; k < deg; ++k) dO = dO + (sl.wr * sl.wr + sl.wr * sl.wr);
            if (fit) dO = fmaf(sl.wf, sl.wf, dO);
            moLUT[threadIdx.x] = ginv(dO);
        }
    }
    double d = 0.0;
    const unsigned fw = q < nq ? ((const unsigned*)(pd.flags + gb))[q] : 0u;
    const bool any_active = (fw & 0x20202020u) != 0u;
    float4* rO4 = (float4*)(pd.rO + gb);
    float4* rA4 = (float4*)(pd.rA + gb);
    float apo[8], r[8], apa[4], ma[4], ra[4];
    if (any_active) {                                       // the data loads go out first; alpha is fetched while they fly
        const float4* ApO4 = (const float4*)(pd.ApO + gb);
        const float4* ApA4 = (const float4*)(pd.ApA + gb);
        const float4* mA4 = (const float4*)(pd.preA + gb);
        *(float4*)&apo[0] = ApO4[2 * q]; *(float4*)&apo[4] = ApO4[2 * q + 1];
        *(float4*)&r[0] = rO4[2 * q]; *(float4*)&r[4] = rO4[2 * q + 1];
        *(float4*)apa = ApA4[q]; *(float4*)ma = mA4[q]; *(float4*)ra = rA4[q];
    }
    __builtin_amdgcn_sched_barrier(0);
    const double* rs = pd.red + (size_t)b * pd.nslots * NSHARD;
    const float rho = read_scalar(rs + (size_t)(2 * l) * NSHARD);
    const float sigma = read_scalar(rs + (size_t)(2 * l + 1) * NSHARD);
    float alpha = 0.f;
    if (sigma > 0.f) alpha = rho / sigma;
    __syncthreads();
    if (any_active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned fk = (fw >> (8 * k)) & 0xffu;
            if (!(fk & F_ACT)) continue;
            const float mo = moLUT[__popc(fk & 15u) + 5 * (int)((fk >> 4) & 1u)];
            r[2 * k] = fmaf(-alpha, apo[2 * k], r[2 * k]);
            r[2 * k + 1] = fmaf(-alpha, apo[2 * k + 1], r[2 * k + 1]);
            ra[k] = fmaf(-alpha, apa[k], ra[k]);
            const float zx = mo * r[2 * k], zy = mo * r[2 * k + 1], za = ma[k] * ra[k];
            d += (double)dot3(zx, zy, za, r[2 * k], r[2 * k + 1], ra[k]);
        }
    }
    const RedTicket rt = red_arrive<1>(pd, b, blockIdx.x, gridDim.x, d, 0.0, rtag);
    if (any_active) {
        rO4[2 * q] = *(float4*)&r[0]; rO4[2 * q + 1] = *(float4*)&r[4];
        rA4[q] = *(float4*)ra;
    }
    red_finish<1>(pd, b, rt, pd.red + ((size_t)b * pd.nslots + (2 * l + 2)) * NSHARD, nullptr);
}

// Phase B (k_pcg_b4's update, four consecutive vertices per lane, 16-byte accesses) without the z and preO reads.
// grid = (ceil(N/4/256), frames), block = 256.  Gauss-Newton plans with N % 4 == 0 only.
__global__ __launch_bounds__(256) void k_pcg_b4_lean(PlanDev pd, int l)
{
    __shared__ float moLUT[12];
    const int b = blockIdx.y;
    const unsigned rtag = red_tag(pd, b, blockIdx.x);
    const int q = blockIdx.x * 256 + threadIdx.x;          // quad index
    const int nq = pd.N >> 2;
    const size_t gb = (size_t)b * pd.N;
    {
        // M^-1 of the Offset components as k_gn_init computes it: D_O = sum over valid edges of (wr*wr + wr*wr), plus
        // wf*wf if the fit term is on: it depends on (degree, fit) only
        const Slot sl = pd.slots[b];
        if (threadIdx.x < 10) {
            const int deg = threadIdx.x % 5, fit = threadIdx.x / 5;
            float dO = 0.f;
            for (int k = 0; k < deg; ++k) dO = dO + (sl.wr * sl.wr + sl.wr * sl.wr);
            if (fit) dO = fmaf(sl.wf, sl.wf, dO);
            moLUT[threadIdx.x] = ginv(dO);
        }
    }
    const float4* __restrict__ pO4 = (const float4*)(((l & 1) ? pd.pO0 : pd.pO1) + gb);
    const float4* __restrict__ pA4 = (const float4*)(((l & 1) ? pd.pA0 : pd.pA1) + gb);
    const double* rs = pd.red + (size_t)b * pd.nslots * NSHARD;
    const float rho = read_scalar(rs + (size_t)(2 * l) * NSHARD);
    const float sigma = read_scalar(rs + (size_t)(2 * l + 1) * NSHARD);
    float alpha = 0.f;
    if (sigma > 0.f) alpha = rho / sigma;
    __syncthreads();
    double d = 0.0;
    const unsigned fw = q < nq ? ((const unsigned*)(pd.flags + gb))[q] : 0u;      // 4 flag bytes
    const bool any_active = (fw & 0x20202020u) != 0u;
    float4* dO4 = (float4*)(pd.deltaO + gb); float4* rO4 = (float4*)(pd.rO + gb); float4* zO4 = (float4*)(pd.zO + gb);
    float4* dA4 = (float4*)(pd.deltaA + gb); float4* rA4 = (float4*)(pd.rA + gb); float4* zA4 = (float4*)(pd.zA + gb);
    float po[8], apo[8], dl[8], r[8], z[8], pa[4], apa[4], ma[4], dla[4], ra[4], za[4];
    if (any_active) {
        const float4* ApO4 = (const float4*)(pd.ApO + gb);
        const float4* ApA4 = (const float4*)(pd.ApA + gb); const float4* mA4 = (const float4*)(pd.preA + gb);
        *(float4*)&po[0] = pO4[2 * q]; *(float4*)&po[4] = pO4[2 * q + 1];
        *(float4*)&apo[0] = ApO4[2 * q]; *(float4*)&apo[4] = ApO4[2 * q + 1];
        *(float4*)&dl[0] = dO4[2 * q]; *(float4*)&dl[4] = dO4[2 * q + 1];
        *(float4*)&r[0] = rO4[2 * q]; *(float4*)&r[4] = rO4[2 * q + 1];
        *(float4*)pa = pA4[q]; *(float4*)apa = ApA4[q]; *(float4*)ma = mA4[q];
        *(float4*)dla = dA4[q]; *(float4*)ra = rA4[q];
        const bool all_active = (fw & 0x20202020u) == 0x20202020u;
        if (!all_active) {                                  // an excluded vertex keeps whatever its z holds
            *(float4*)&z[0] = zO4[2 * q]; *(float4*)&z[4] = zO4[2 * q + 1];
            *(float4*)za = zA4[q];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned fk = (fw >> (8 * k)) & 0xffu;
            if (!(fk & F_ACT)) continue;
            const float mo = moLUT[__popc(fk & 15u) + 5 * (int)((fk >> 4) & 1u)];
            dl[2 * k] = fmaf(alpha, po[2 * k], dl[2 * k]);
            dl[2 * k + 1] = fmaf(alpha, po[2 * k + 1], dl[2 * k + 1]);
            dla[k] = fmaf(alpha, pa[k], dla[k]);
            r[2 * k] = fmaf(-alpha, apo[2 * k], r[2 * k]);
            r[2 * k + 1] = fmaf(-alpha, apo[2 * k + 1], r[2 * k + 1]);
            ra[k] = fmaf(-alpha, apa[k], ra[k]);
            z[2 * k] = mo * r[2 * k];
            z[2 * k + 1] = mo * r[2 * k + 1];
            za[k] = ma[k] * ra[k];
            d += (double)dot3(z[2 * k], z[2 * k + 1], za[k], r[2 * k], r[2 * k + 1], ra[k]);
        }
    }
    // the workgroup's share of rho_{l+1} goes out, and its ticket is taken, BEFORE the nine streaming stores below (the
    // ticket's return would otherwise wait for them to drain: arap_device.h, red_arrive)
    const RedTicket rt = red_arrive<1>(pd, b, blockIdx.x, gridDim.x, d, 0.0, rtag);
    if (any_active) {
        dO4[2 * q] = *(float4*)&dl[0]; dO4[2 * q + 1] = *(float4*)&dl[4];
        rO4[2 * q] = *(float4*)&r[0]; rO4[2 * q + 1] = *(float4*)&r[4];
        zO4[2 * q] = *(float4*)&z[0]; zO4[2 * q + 1] = *(float4*)&z[4];
        dA4[q] = *(float4*)dla; rA4[q] = *(float4*)ra; zA4[q] = *(float4*)za;
    }
    red_finish<1>(pd, b, rt, pd.red + ((size_t)b * pd.nslots + (2 * l + 2)) * NSHARD, nullptr);
}

}  // namespace arap
