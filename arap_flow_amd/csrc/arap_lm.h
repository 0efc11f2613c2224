// arap_lm.h -- kernels of the "LMGPU" solver kind (Levenberg-Marquardt branch of the reference's solver,
// solverGPUGaussNewton.t, `problemSpec:UsesLambda()`): PCGSaveSSq :622-627, PCGComputeCtC :616-621 (computeCtC =
// diag(J^T J) / trust_region_radius, o.t:2255-2287), PCGFinalizeDiagonal :629-662, PCGStep2_1stHalf / computeAdelta /
// PCGStep2_2ndHalf :491-535,570-575, computeModelCost :665-678 (o.t:2180-2201).  applyJTJ + CtC*P and the q term of
// PCGStep2 live in k_pcg_a / k_pcg_b behind PlanDev::lm.  The application never selects this kind
// (CombinedSolverBase.h:75-77) and the reference holds no LM output, so its parity is pinned to the CPU
// restatement only.  Host loop: arapopt.hip:plan_step_lm.
#pragma once
#include "arap_kernels.h"

namespace arap {

// raw diag(J^T J) of vertex v (the accumulation order of k_gn_init)
__device__ __forceinline__ void diag_raw(const PlanDev& pd, const Slot& sl, const VIdx& v, unsigned f, float& DO, float& DA)
{
    const float wr = sl.wr, wf = sl.wf;
    const float2 csi = pd.cs[v.g];
    const float2 Ui = sl.U[v.i];
    float dO = 0.f, dA = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (!(f & (1u << s))) continue;
        const float2 Un = sl.U[v.i + noff(s, pd.W)];
        const float dx = Ui.x - Un.x, dy = Ui.y - Un.y;
        const float qx = fmaf(-csi.y, dx, -(csi.x * dy)), qy = fmaf(csi.x, dx, -(csi.y * dy));
        dO = dO + (wr * wr + wr * wr);
        dA = fmaf(wr * wr, fmaf(qx, qx, qy * qy), dA);
    }
    if (f & F_FIT) dO = fmaf(wf, wf, dO);
    DO = dO; DA = dA;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi)
{
    const float m = x > lo ? x : lo;
    return m < hi ? m : hi;
}

// after k_gn_prep + k_gn_init (which left r = -g, pre = guardedInvert(D), delta = 0):
// SSq (first step only), CtC, the true preconditioner, b = r, p = pre r, rho0, Q0
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_lm_prepare(PlanDev pd, float radius, float min_diag, float max_diag,
                                                                int first)
{
    const VIdx v = vidx(pd);
    double* const rho0 = pd.red + ((size_t)v.b * pd.nslots + 0) * NSHARD;
    if (!pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg]) { block_reduce_fixed<2>(pd, v.b, v.lb, v.nlb, 0.0, 0.0, rho0, pd.lmred); return; }
    const Slot sl = pd.slots[v.b];
    const unsigned f = v.in ? pd.flags[v.g] : 0u;
    double d = 0.0, q = 0.0;
    if (f & F_ACT) {
        float DO, DA;
        diag_raw(pd, sl, v, f, DO, DA);
        const float inv_radius = 1.0f / radius;
        if (first) { pd.SSqO[v.g] = pd.preO[v.g]; pd.SSqA[v.g] = pd.preA[v.g]; }
        const float ssO = pd.SSqO[v.g].x, ssA = pd.SSqA[v.g];
        const float uO = DO * inv_radius, uA = DA * inv_radius;
        const float mO = (1.0f / ssO) / radius, mA = (1.0f / ssA) / radius;
        const float cO = clampf(uO, min_diag * mO, max_diag * mO), cA = clampf(uA, min_diag * mA, max_diag * mA);
        pd.CtCO[v.g] = make_float2(cO, cO);
        pd.CtCA[v.g] = cA;
        const float pO_ = 1.0f / fmaf(radius, uO, cO), pA_ = 1.0f / fmaf(radius, uA, cA);
        pd.preO[v.g] = make_float2(pO_, pO_);
        pd.preA[v.g] = pA_;
        const float2 r = pd.rO[v.g];
        const float ra = pd.rA[v.g];
        pd.bO[v.g] = r;
        pd.bA[v.g] = ra;
        const float px = pO_ * r.x, py = pO_ * r.y, pa = pA_ * ra;
        pd.pO0[v.g] = make_float2(px, py);
        pd.pA0[v.g] = pa;
        d = (double)dot3(r.x, r.y, ra, px, py, pa);
        const float2 dl = pd.deltaO[v.g];
        q = (double)(0.5f * dot3(dl.x, dl.y, pd.deltaA[v.g], r.x + r.x, r.y + r.y, ra + ra));
    }
    block_reduce_fixed<2>(pd, v.b, v.lb, v.nlb, d, q, rho0, pd.lmred);
}

// out = J^T J in + CtC in   (computeAdelta :570-575); `in` may be any plan vector
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_lm_apply(PlanDev pd, const float2* inO, const float* inA, float2* outO,
                                                              float* outA)
{
    const VIdx v = vidx(pd);
    if (!pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg]) return;
    const unsigned f = v.in ? pd.flags[v.g] : 0u;
    if (!(f & F_ACT)) return;
    const Slot sl = pd.slots[v.b];
    const size_t gb = (size_t)v.b * pd.N;
    const float wr2 = sl.wr * sl.wr;
    const float2 pO = inO[v.g];
    const float pA = inA[v.g];
    const float2 csi = pd.cs[v.g];
    const float ci = csi.x, si = csi.y;
    const float2 Ui = sl.U[v.i];
    float ax = 0.f, ay = 0.f, aa = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (!(f & (1u << s))) continue;
        const int n = v.i + noff(s, pd.W);
        const float2 qO = inO[gb + n];
        const float qA = inA[gb + n];
        const float2 csn = pd.cs[gb + n];
        const float cn = csn.x, sn = csn.y;
        const float2 Un = sl.U[n];
        const float dx = Ui.x - Un.x, dy = Ui.y - Un.y;
        const float qx = fmaf(-si, dx, -(ci * dy)), qy = fmaf(ci, dx, -(si * dy));
        const float hx = fmaf(-sn, dx, -(cn * dy)), hy = fmaf(cn, dx, -(sn * dy));
        const float px = pO.x - qO.x, py = pO.y - qO.y;
        const float tx = fmaf(-qx, pA, px), ty = fmaf(-qy, pA, py);
        ax = fmaf(wr2, fmaf(-hx, qA, px + tx), ax);
        ay = fmaf(wr2, fmaf(-hy, qA, py + ty), ay);
        aa = fmaf(-wr2, fmaf(qx, tx, qy * ty), aa);
    }
    if (f & F_FIT) {
        const float wf2 = sl.wf * sl.wf;
        ax = fmaf(wf2, pO.x, ax);
        ay = fmaf(wf2, pO.y, ay);
    }
    const float2 c = pd.CtCO[v.g];
    outO[v.g] = make_float2(fmaf(c.x, pO.x, ax), fmaf(c.y, pO.y, ay));
    outA[v.g] = fmaf(pd.CtCA[v.g], pA, aa);
}

// PCGStep2_1stHalf: delta += alpha p   (p = the buffer k_pcg_a(l) wrote)
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_lm_step2a(PlanDev pd, int l)
{
    const VIdx v = vidx(pd);
    if (!pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg]) return;
    const float2* __restrict__ pO_ = (l & 1) ? pd.pO0 : pd.pO1;
    const float* __restrict__ pA_ = (l & 1) ? pd.pA0 : pd.pA1;
    const double* rs = pd.red + (size_t)v.b * pd.nslots * NSHARD;
    const float rho = read_scalar(rs + (size_t)(2 * l) * NSHARD);
    const float sigma = read_scalar(rs + (size_t)(2 * l + 1) * NSHARD);
    float alpha = 0.f;
    if (sigma > 0.f) alpha = rho / sigma;
    if (!v.in || !(pd.flags[v.g] & F_ACT)) return;
    float2 d = pd.deltaO[v.g];
    const float2 p = pO_[v.g];
    d.x = fmaf(alpha, p.x, d.x);
    d.y = fmaf(alpha, p.y, d.y);
    pd.deltaO[v.g] = d;
    pd.deltaA[v.g] = fmaf(alpha, pA_[v.g], pd.deltaA[v.g]);
}

// PCGStep2_2ndHalf: r = b - A delta ; z = pre r ; rho' ; q = 0.5 delta.(r + b)
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_lm_step2b(PlanDev pd, int l)
{
    const VIdx v = vidx(pd);
    double* const rho_next = pd.red + ((size_t)v.b * pd.nslots + (2 * l + 2)) * NSHARD;
    double* const q_next = pd.lmred + (size_t)(l + 1) * NSHARD;
    if (!pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg]) { block_reduce_fixed<2>(pd, v.b, v.lb, v.nlb, 0.0, 0.0, rho_next, q_next); return; }
    const unsigned f = v.in ? pd.flags[v.g] : 0u;
    double d = 0.0, q = 0.0;
    if (f & F_ACT) {
        const float2 b = pd.bO[v.g], Ad = pd.AdO[v.g], m = pd.preO[v.g], dl = pd.deltaO[v.g];
        const float ba = pd.bA[v.g], Ada = pd.AdA[v.g], ma = pd.preA[v.g], dla = pd.deltaA[v.g];
        const float rx = b.x - Ad.x, ry = b.y - Ad.y, ra = ba - Ada;
        const float zx = m.x * rx, zy = m.y * ry, za = ma * ra;
        pd.rO[v.g] = make_float2(rx, ry);
        pd.rA[v.g] = ra;
        pd.zO[v.g] = make_float2(zx, zy);
        pd.zA[v.g] = za;
        d = (double)dot3(zx, zy, za, rx, ry, ra);
        q = (double)(0.5f * dot3(dl.x, dl.y, dla, rx + b.x, ry + b.y, ra + ba));
    }
    block_reduce_fixed<2>(pd, v.b, v.lb, v.nlb, d, q, rho_next, q_next);
}

// computeModelCost: 0.5 * sum (F + J delta)^2 over the residuals centred on active vertices
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_lm_model_cost(PlanDev pd, int slot)
{
    const VIdx v = vidx(pd);
    if (!pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg]) { block_reduce_fixed<1>(pd, v.b, v.lb, v.nlb, 0.0, 0.0, pd.lmred + (size_t)slot * NSHARD, nullptr); return; }
    const Slot sl = pd.slots[v.b];
    const unsigned f = v.in ? pd.flags[v.g] : 0u;
    double d = 0.0;
    if (f & F_ACT) {
        const size_t gb = (size_t)v.b * pd.N;
        const float wr = sl.wr, wf = sl.wf;
        const float2 cs = pd.cs[v.g];
        const float2 Oi = sl.O[v.i], Ui = sl.U[v.i], di = pd.deltaO[v.g];
        const float dai = pd.deltaA[v.g];
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!(f & (1u << s))) continue;
            const int n = v.i + noff(s, pd.W);
            const float2 On = sl.O[n], Un = sl.U[n], dn = pd.deltaO[gb + n];
            const float dx = Ui.x - Un.x, dy = Ui.y - Un.y;
            const float ex = wr * ((Oi.x - On.x) - fmaf(cs.x, dx, -(cs.y * dy)));
            const float ey = wr * ((Oi.y - On.y) - fmaf(cs.y, dx, cs.x * dy));
            const float qx = fmaf(-cs.y, dx, -(cs.x * dy)), qy = fmaf(cs.x, dx, -(cs.y * dy));
            const float mx = fmaf(wr, fmaf(-qx, dai, di.x - dn.x), ex);
            const float my = fmaf(wr, fmaf(-qy, dai, di.y - dn.y), ey);
            t = fmaf(mx, mx, t);
            t = fmaf(my, my, t);
        }
        if (f & F_FIT) {
            const float2 Ci = sl.C[v.i];
            const float mx = fmaf(wf, di.x, wf * (Oi.x - Ci.x)), my = fmaf(wf, di.y, wf * (Oi.y - Ci.y));
            t = fmaf(mx, mx, t);
            t = fmaf(my, my, t);
        }
        d = (double)(0.5f * t);
    }
    block_reduce_fixed<1>(pd, v.b, v.lb, v.nlb, d, 0.0, pd.lmred + (size_t)slot * NSHARD, nullptr);
}

}  // namespace arap
