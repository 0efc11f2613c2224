// arap_tiled.h -- LDS-staged variant of phase A (k_pcg_a) for the general two-kernel path, templated on the tile
// shape so that BASELINE config 5's tile sweep {16x16, 32x8, 64x4, 32x16, 64x8} can be measured.
//
// Each workgroup stages the NEW search direction p_l = z + beta p_{l-1} and cos/sin(A) of a TX x TY tile plus a
// one-vertex halo in LDS (interior cells by their own thread, halo cells by the border threads, which redo the
// neighbour's update expression), then every thread reads its four neighbours from LDS.  Compared with k_pcg_a,
// which re-reads z, p and cos/sin of the four neighbours through L1/L2, a vertex's data is fetched from global
// memory (TX+2)(TY+2)/(TX TY) times instead of 5 times.  Arithmetic is k_pcg_a's, operation for operation.
#pragma once
#include "arap_kernels.h"

namespace arap {

template <int TX, int TY>
__global__ __launch_bounds__(TX* TY) void k_pcg_a_lds(PlanDev pd, int l)
{
    constexpr int LW = TX + 2, LH = TY + 2;
    __shared__ float2 sP[LH * LW];
    __shared__ float2 sC[LH * LW];
    __shared__ float sA[LH * LW];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * TX + tx, y = blockIdx.y * TY + ty, b = blockIdx.z;
    const int W = pd.W, H = pd.H;
    const bool in = x < W && y < H;
    const int i = x + W * y;
    const size_t gb = (size_t)b * pd.N;
    const unsigned f = in ? pd.flags[gb + i] : 0u;
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
    double* const sigma_l = pd.red + ((size_t)b * pd.nslots + (2 * l + 1)) * NSHARD;
    if (!__syncthreads_or((int)(f & F_ACT))) {                   // nothing active in this tile (it still reports: block_reduce_fixed)
        block_reduce_fixed<1>(pd, b, wg, nwg, 0.0, 0.0, sigma_l, nullptr);
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
        const float2 Ui = sl.U[i];
        float ax = 0.f, ay = 0.f, aa = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!(f & (1u << s))) continue;
            const int nc = cell + (s == 0 ? 1 : (s == 1 ? -1 : (s == 2 ? LW : -LW)));
            const float2 qO = sP[nc];
            const float qA = sA[nc];
            const float2 csn = sC[nc];
            const float cn = csn.x, sn = csn.y;
            const float2 Un = sl.U[i + noff(s, W)];
            const float dx = Ui.x - Un.x, dy = Ui.y - Un.y;
            const float qx = fmaf(-si, dx, -(ci * dy)), qy = fmaf(ci, dx, -(si * dy));
            const float hx = fmaf(-sn, dx, -(cn * dy)), hy = fmaf(cn, dx, -(sn * dy));
            const float px = pO.x - qO.x, py = pO.y - qO.y;
            const float tx_ = fmaf(-qx, pA, px), ty_ = fmaf(-qy, pA, py);
            ax = fmaf(wr2, fmaf(-hx, qA, px + tx_), ax);
            ay = fmaf(wr2, fmaf(-hy, qA, py + ty_), ay);
            aa = fmaf(-wr2, fmaf(qx, tx_, qy * ty_), aa);
        }
        if (f & F_FIT) {
            const float wf2 = sl.wf * sl.wf;
            ax = fmaf(wf2, pO.x, ax);
            ay = fmaf(wf2, pO.y, ay);
        }
        if (pd.lm) {
            const float2 c = pd.CtCO[gb + i];
            ax = fmaf(c.x, pO.x, ax);
            ay = fmaf(c.y, pO.y, ay);
            aa = fmaf(pd.CtCA[gb + i], pA, aa);
        }
        pd.ApO[gb + i] = make_float2(ax, ay);
        pd.ApA[gb + i] = aa;
        d = (double)dot3(pO.x, pO.y, pA, ax, ay, aa);
    }
    block_reduce_fixed<1>(pd, b, wg, nwg, d, 0.0, sigma_l, nullptr);
}

}  // namespace arap
