// arap_kernels.h -- HIP kernels of the Gauss-Newton / PCG solve (gfx950, wave64).
//
// Kernel map (reference kernels: solverGPUGaussNewton.t:361-592, see DESIGN.md "Kernels"):
//   k_gn_prep    : per GN step: cos/sin of Angle, flag byte, tile activity        (new; replaces the
//                  per-iteration re-evaluation of cos/sin, Mask and Constraints of the generated code)
//   k_gn_init    : PCGInit1 (:361-397)
//   k_pcg_a      : PCGStep3 of the previous iteration (:537-550) fused with PCGStep1 (:421-434)
//   k_pcg_b      : PCGStep2 (:446-489)
//   k_gn_update  : PCGLinearUpdate (:552-557)    (frame solver: done by the resident launch itself, ResDev::fuse_update)
//   k_cost       : computeCost (:580-592)
// Launch shape: workgroup = 64 x 4 threads (4 wavefronts, each 64 consecutive x of one row),
// grid = (ceil(W/64), ceil(H/4), batch); one thread per mesh vertex.
#pragma once
#include "arap_device.h"

namespace arap {

struct VIdx {
    int x, y, i, b;       // vertex coords, linear index within the frame, slot
    size_t g;             // b*N + i : index into the plan-owned [batch][N] images
    unsigned wg;          // linear index of the workgroup's 64x4 tile within the frame
    unsigned lb, nlb;     // linear index of the workgroup within the frame's launch, workgroups per frame in the launch
    bool in;              // inside the image
    bool tile_ok;         // the workgroup has a tile (list launches: blockIdx.x < the frame's list length)
};

// Two launch shapes: grid (tilesX, tilesY, frames) over the whole tile grid, or -- pd.t64list set -- grid (n, 1, frames)
// over the frames' lists of ACTIVE 64x4 tiles (frame solver on the resident path: the per-step kernels around the
// resident launch then touch a quarter of a DAVIS-shaped frame instead of all of it).
__device__ __forceinline__ VIdx vidx(const PlanDev& pd)
{
    VIdx v;
    v.b = blockIdx.z;
    int tx = blockIdx.x, ty = blockIdx.y;
    v.tile_ok = true;
    if (pd.t64list) {
        const int n = pd.t64n[v.b];
        v.tile_ok = (int)blockIdx.x < n;
        const int t = v.tile_ok ? pd.t64list[(size_t)v.b * pd.tilesX * pd.tilesY + blockIdx.x] : 0;
        ty = t / pd.tilesX;
        tx = t - ty * pd.tilesX;
        v.lb = blockIdx.x;
        v.nlb = gridDim.x;
    } else {
        v.lb = blockIdx.y * gridDim.x + blockIdx.x;
        v.nlb = gridDim.x * gridDim.y;
    }
    v.x = tx * TILE_X + threadIdx.x;
    v.y = ty * TILE_Y + threadIdx.y;
    v.in = v.tile_ok && v.x < pd.W && v.y < pd.H;
    v.i = v.x + pd.W * v.y;
    v.g = (size_t)v.b * pd.N + (v.in ? v.i : 0);
    v.wg = (unsigned)(ty * pd.tilesX + tx);
    return v;
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gn_prep_body(const PlanDev& pd, const VIdx& v)
{
    const Slot sl = pd.slots[v.b];
    unsigned f = 0;
    if (v.in) {
        const int W = pd.W, H = pd.H;
        const bool act = sl.M[v.i] == 0.0f;
        if (act) {
            f = F_ACT;
            if (v.x + 1 < W && sl.M[v.i + 1] == 0.0f) f |= F_E0;
            if (v.x > 0 && sl.M[v.i - 1] == 0.0f) f |= F_E1;
            if (v.y + 1 < H && sl.M[v.i + W] == 0.0f) f |= F_E2;
            if (v.y > 0 && sl.M[v.i - W] == 0.0f) f |= F_E3;
            const float2 c = sl.C[v.i];
            if (c.x >= 0.0f && c.y >= 0.0f) f |= F_FIT;
        }
        pd.flags[v.g] = (uint8_t)f;
        pd.cs[v.g] = sincos_spec(sl.A[v.i]);
    }
    const int any = __syncthreads_or((int)(f & F_ACT));
    if (v.tile_ok && threadIdx.x == 0 && threadIdx.y == 0)
        pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg] = any ? 1 : 0;
    if (pd.res_gran_n) {
        const int t = threadIdx.y * TILE_X + threadIdx.x;
        if (v.lb == 0 && t < NSHARD) pd.red[(size_t)v.b * pd.nslots * NSHARD + t] = 0.0;
        if (v.b == 0)
            for (int i = (int)v.lb * (TILE_X * TILE_Y) + t; i < pd.res_gran_n; i += (int)v.nlb * (TILE_X * TILE_Y))
                pd.res_gran[i] = 0ull;
    }
}

__global__ __launch_bounds__(TILE_X* TILE_Y) void k_gn_prep(PlanDev pd)
{
    gn_prep_body(pd, vidx(pd));
}

// ------------------------------------------------------------------------------------------------
// Drop-in API, before every step: which 32x8 tiles (fixed grid; launch with 32x8 blocks, one per tile) hold an active
// vertex, and is UrShape the pixel grid on every active vertex (then d = U(c)-U(n) = -s on every valid edge, which is
// what arap_resident.h specialises on; the application always passes that grid, CombinedSolver.h:207-221).
__global__ __launch_bounds__(256) void k_analyse(PlanDev pd, uint8_t* tile_active, int* not_grid)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    const Slot sl = pd.slots[0];
    int act = 0, bad = 0;
    if (x < pd.W && y < pd.H) {
        const int i = x + pd.W * y;
        if (sl.M[i] == 0.0f) {
            act = 1;
            const float2 u = sl.U[i];
            bad = !(u.x == (float)x && u.y == (float)y);
        }
    }
    const int any = __syncthreads_or(act);
    const int anybad = __syncthreads_or(bad);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        tile_active[blockIdx.y * gridDim.x + blockIdx.x] = any ? 1 : 0;
        if (anybad) atomicOr(not_grid, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// PCGInit1: delta = 0; (g, D) = evalJTF; r = -g; pre = guardedInvert(D); p = pre*r; rho0 += r.p
// RESF (frame solver in front of a resident launch): UrShape is the pixel grid the library itself wrote (k_frame_reset),
// so d = U(c) - U(n) is -1 / +1 / +0 exactly and is not loaded; and what the resident kernel does not read -- delta (it
// starts from zero in registers and writes its own result), M^-1 of the Offset components (it derives them from the
// flags; p0 = M^-1 r, which it forms from r) -- is not stored: 40 of ~85 bytes per vertex less.
template <bool RESF>
__device__ __forceinline__ void gn_init_body(const PlanDev& pd)
{
    const VIdx v = vidx(pd);
    // (frame solver: the granules of the resident launches that follow (every workgroup of the launch takes its share, before any leaves) -- tags restart at 1 in every launch; k_gn_prep does
    //  this where it runs)
    if (RESF && pd.res_gran_n && v.b == 0) {
        const int t = threadIdx.y * TILE_X + threadIdx.x;
        for (int i = (int)v.lb * (TILE_X * TILE_Y) + t; i < pd.res_gran_n; i += (int)v.nlb * (TILE_X * TILE_Y))
            pd.res_gran[i] = 0ull;
    }
    double* const rho0 = pd.red + ((size_t)v.b * pd.nslots + 0) * NSHARD;
    // (a workgroup with nothing to do still reports to the order-fixed sum: arap_device.h, block_reduce_fixed)
    if (!pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg]) { block_reduce_fixed<1>(pd, v.b, v.lb, v.nlb, 0.0, 0.0, rho0, nullptr); return; }
    const Slot sl = pd.slots[v.b];
    const unsigned f = v.in ? pd.flags[v.g] : 0u;
    double d = 0.0;
    if (f & F_ACT) {
        const float wr = sl.wr, wf = sl.wf;
        const size_t gb = (size_t)v.b * pd.N;
        const float2 csi = pd.cs[v.g];
        const float ci = csi.x, si = csi.y;
        const float2 Oi = sl.O[v.i];
        float2 Ui = make_float2(0.f, 0.f);
        if (!RESF) Ui = sl.U[v.i];
        float gx = 0.f, gy = 0.f, ga = 0.f, dO = 0.f, dA = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!(f & (1u << s))) continue;
            const int n = v.i + noff(s, pd.W);
            const float2 csn = pd.cs[gb + n];
            const float cn = csn.x, sn = csn.y;
            const float2 On = sl.O[n];
            float dx, dy;
            if (RESF) {
                dx = s == 0 ? -1.0f : (s == 1 ? 1.0f : 0.0f);          // x - (x + 1), x - (x - 1), x - x
                dy = s == 2 ? -1.0f : (s == 3 ? 1.0f : 0.0f);
            } else {
                const float2 Un = sl.U[n];
                dx = Ui.x - Un.x; dy = Ui.y - Un.y;
            }
            const float ox = Oi.x - On.x, oy = Oi.y - On.y;
            const float ex = wr * (ox - fmaf(ci, dx, -(si * dy)));
            const float ey = wr * (oy - fmaf(si, dx, ci * dy));
            const float fx = wr * (fmaf(cn, dx, -(sn * dy)) - ox);
            const float fy = wr * (fmaf(sn, dx, cn * dy) - oy);
            const float qx = fmaf(-si, dx, -(ci * dy)), qy = fmaf(ci, dx, -(si * dy));
            gx = fmaf(wr, ex - fx, gx);
            gy = fmaf(wr, ey - fy, gy);
            ga = fmaf(-wr, fmaf(qx, ex, qy * ey), ga);
            dO = dO + (wr * wr + wr * wr);
            dA = fmaf(wr * wr, fmaf(qx, qx, qy * qy), dA);
        }
        float dOf = dO;
        if (f & F_FIT) {
            const float2 Ci = sl.C[v.i];
            gx = fmaf(wf, wf * (Oi.x - Ci.x), gx);
            gy = fmaf(wf, wf * (Oi.y - Ci.y), gy);
            dOf = fmaf(wf, wf, dO);
        }
        const float rx = -gx, ry = -gy, ra = -ga;
        const float mo = ginv(dOf), ma = ginv(dA);
        const float px = mo * rx, py = mo * ry, pa = ma * ra;
        if (!RESF) {
            pd.deltaO[v.g] = make_float2(0.f, 0.f);
            pd.deltaA[v.g] = 0.f;
            pd.preO[v.g] = make_float2(mo, mo);
        }
        pd.rO[v.g] = make_float2(rx, ry);
        pd.rA[v.g] = ra;
        pd.preA[v.g] = ma;
        if (!RESF) {                                      // (the resident kernel forms p0 = M^-1 r itself)
            pd.pO0[v.g] = make_float2(px, py);
            pd.pA0[v.g] = pa;
        }
        d = (double)dot3(rx, ry, ra, px, py, pa);
    } else if (v.in && !RESF) {
        pd.preO[v.g] = make_float2(0.f, 0.f);   // PCGInit1 stores pre = 0 on excluded vertices (:395)
        pd.preA[v.g] = 0.f;
    }
    block_reduce_fixed<1>(pd, v.b, v.lb, v.nlb, d, 0.0, rho0, nullptr);
}

__global__ __launch_bounds__(TILE_X* TILE_Y) void k_gn_init(PlanDev pd) { gn_init_body<false>(pd); }
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_gn_init_resf(PlanDev pd) { gn_init_body<true>(pd); }

// ------------------------------------------------------------------------------------------------
// Iteration l, phase A.  p_l = (l == 0) ? p_init : z + beta * p_{l-1}, with
// beta = rho_l / rho_{l-1} if rho_{l-1} > 0 else 0 (PCGStep3).  Then Ap = J^T J p_l and
// sigma_l += p_l . Ap (PCGStep1).  p is double-buffered (read `pin`, write `pout`) because the
// stencil needs the neighbours' p_l, which this kernel recomputes from their z and p_{l-1}.
// reduction slots of the GN step: 0 = rho_0 ; 2l+1 = sigma_l ; 2l+2 = rho_{l+1}
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_pcg_a(PlanDev pd, int l)
{
    const VIdx v = vidx(pd);
    double* const sigma_l = pd.red + ((size_t)v.b * pd.nslots + (2 * l + 1)) * NSHARD;
    if (!pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg]) { block_reduce_fixed<1>(pd, v.b, v.lb, v.nlb, 0.0, 0.0, sigma_l, nullptr); return; }
    const Slot sl = pd.slots[v.b];
    const size_t gb = (size_t)v.b * pd.N;
    const float2* __restrict__ pinO = (l & 1) ? pd.pO1 : pd.pO0;
    const float* __restrict__ pinA = (l & 1) ? pd.pA1 : pd.pA0;
    float2* __restrict__ poutO = (l & 1) ? pd.pO0 : pd.pO1;
    float* __restrict__ poutA = (l & 1) ? pd.pA0 : pd.pA1;
    float beta = 0.f;
    if (l > 0) {
        const double* rs = pd.red + (size_t)v.b * pd.nslots * NSHARD;
        const float rhoNew = read_scalar(rs + (size_t)(2 * l) * NSHARD);
        const float rhoOld = read_scalar(rs + (size_t)(2 * l - 2) * NSHARD);
        if (rhoOld > 0.f) beta = rhoNew / rhoOld;
    }
    const unsigned f = v.in ? pd.flags[v.g] : 0u;
    double d = 0.0;
    if (f & F_ACT) {
        const float wr2 = sl.wr * sl.wr;
        float2 pO = pinO[v.g];
        float pA = pinA[v.g];
        if (l > 0) {
            const float2 zO = pd.zO[v.g];
            const float zA = pd.zA[v.g];
            pO.x = fmaf(beta, pO.x, zO.x);
            pO.y = fmaf(beta, pO.y, zO.y);
            pA = fmaf(beta, pA, zA);
        }
        poutO[v.g] = pO;
        poutA[v.g] = pA;
        const float2 csi = pd.cs[v.g];
        const float ci = csi.x, si = csi.y;
        const float2 Ui = sl.U[v.i];
        float ax = 0.f, ay = 0.f, aa = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!(f & (1u << s))) continue;
            const int n = v.i + noff(s, pd.W);
            float2 qO = pinO[gb + n];
            float qA = pinA[gb + n];
            if (l > 0) {
                const float2 zO = pd.zO[gb + n];
                const float zA = pd.zA[gb + n];
                qO.x = fmaf(beta, qO.x, zO.x);
                qO.y = fmaf(beta, qO.y, zO.y);
                qA = fmaf(beta, qA, zA);
            }
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
        if (pd.lm) {                                   // applyJTJ + CtC*P (o.t:2076-2082)
            const float2 c = pd.CtCO[v.g];
            ax = fmaf(c.x, pO.x, ax);
            ay = fmaf(c.y, pO.y, ay);
            aa = fmaf(pd.CtCA[v.g], pA, aa);
        }
        pd.ApO[v.g] = make_float2(ax, ay);
        pd.ApA[v.g] = aa;
        d = (double)dot3(pO.x, pO.y, pA, ax, ay, aa);
    }
    block_reduce_fixed<1>(pd, v.b, v.lb, v.nlb, d, 0.0, sigma_l, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Iteration l, phase B = PCGStep2: alpha = rho_l / sigma_l if sigma_l > 0 else 0;
// delta += alpha p ; r -= alpha Ap ; z = pre * r ; rho_{l+1} += z.r
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_pcg_b(PlanDev pd, int l)
{
    const VIdx v = vidx(pd);
    double* const rho_next = pd.red + ((size_t)v.b * pd.nslots + (2 * l + 2)) * NSHARD;
    double* const q_next = pd.lm ? pd.lmred + (size_t)(l + 1) * NSHARD : nullptr;
    if (!pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg]) {
        if (pd.lm) block_reduce_fixed<2>(pd, v.b, v.lb, v.nlb, 0.0, 0.0, rho_next, q_next);
        else block_reduce_fixed<1>(pd, v.b, v.lb, v.nlb, 0.0, 0.0, rho_next, nullptr);
        return;
    }
    const float2* __restrict__ pO_ = (l & 1) ? pd.pO0 : pd.pO1;   // written by k_pcg_a(l)
    const float* __restrict__ pA_ = (l & 1) ? pd.pA0 : pd.pA1;
    const double* rs = pd.red + (size_t)v.b * pd.nslots * NSHARD;
    const float rho = read_scalar(rs + (size_t)(2 * l) * NSHARD);
    const float sigma = read_scalar(rs + (size_t)(2 * l + 1) * NSHARD);
    float alpha = 0.f;
    if (sigma > 0.f) alpha = rho / sigma;
    const unsigned f = v.in ? pd.flags[v.g] : 0u;
    double d = 0.0, q = 0.0;
    if (f & F_ACT) {
        const float2 pO = pO_[v.g], ApO = pd.ApO[v.g], mO = pd.preO[v.g];
        const float pA = pA_[v.g], ApA = pd.ApA[v.g], mA = pd.preA[v.g];
        float2 dO = pd.deltaO[v.g], rO = pd.rO[v.g];
        float dA = pd.deltaA[v.g], rA = pd.rA[v.g];
        dO.x = fmaf(alpha, pO.x, dO.x);
        dO.y = fmaf(alpha, pO.y, dO.y);
        dA = fmaf(alpha, pA, dA);
        rO.x = fmaf(-alpha, ApO.x, rO.x);
        rO.y = fmaf(-alpha, ApO.y, rO.y);
        rA = fmaf(-alpha, ApA, rA);
        const float zx = mO.x * rO.x, zy = mO.y * rO.y, za = mA * rA;
        pd.deltaO[v.g] = dO;
        pd.deltaA[v.g] = dA;
        pd.rO[v.g] = rO;
        pd.rA[v.g] = rA;
        pd.zO[v.g] = make_float2(zx, zy);
        pd.zA[v.g] = za;
        d = (double)dot3(zx, zy, za, rO.x, rO.y, rA);
        if (pd.lm) {                                   // computeQ (:477-482): q = 0.5 delta.(r + b)
            const float2 b = pd.bO[v.g];
            q = (double)(0.5f * dot3(dO.x, dO.y, dA, rO.x + b.x, rO.y + b.y, rA + pd.bA[v.g]));
        }
    }
    if (pd.lm) block_reduce_fixed<2>(pd, v.b, v.lb, v.nlb, d, q, rho_next, q_next);
    else block_reduce_fixed<1>(pd, v.b, v.lb, v.nlb, d, 0.0, rho_next, nullptr);
}

// ------------------------------------------------------------------------------------------------
// k_pcg_b for N % 4 == 0 (Gauss-Newton plans): the same element-wise update, four consecutive vertices per
// lane so that every global access is 16 bytes wide (float4 of the Angle-shaped images, 2 x float4 of the
// Offset-shaped ones) -- cdna guide G13: 16 B per lane is the coalescing sweet spot.  1-D grid over N/4 quads:
// grid = (ceil(N/4/256), 1, frames), block = 256.  Excluded vertices are written back unchanged.
__global__ __launch_bounds__(256) void k_pcg_b4(PlanDev pd, int l)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;          // quad index
    const int nq = pd.N >> 2;
    const size_t gb = (size_t)b * pd.N;
    const float4* __restrict__ pO4 = (const float4*)(((l & 1) ? pd.pO0 : pd.pO1) + gb);
    const float4* __restrict__ pA4 = (const float4*)(((l & 1) ? pd.pA0 : pd.pA1) + gb);
    const double* rs = pd.red + (size_t)b * pd.nslots * NSHARD;
    const float rho = read_scalar(rs + (size_t)(2 * l) * NSHARD);
    const float sigma = read_scalar(rs + (size_t)(2 * l + 1) * NSHARD);
    float alpha = 0.f;
    if (sigma > 0.f) alpha = rho / sigma;
    double d = 0.0;
    const unsigned fw = q < nq ? ((const unsigned*)(pd.flags + gb))[q] : 0u;      // 4 flag bytes
    if (fw & 0x20202020u) {
        float4* dO4 = (float4*)(pd.deltaO + gb); float4* rO4 = (float4*)(pd.rO + gb); float4* zO4 = (float4*)(pd.zO + gb);
        float4* dA4 = (float4*)(pd.deltaA + gb); float4* rA4 = (float4*)(pd.rA + gb); float4* zA4 = (float4*)(pd.zA + gb);
        const float4* ApO4 = (const float4*)(pd.ApO + gb); const float4* mO4 = (const float4*)(pd.preO + gb);
        const float4* ApA4 = (const float4*)(pd.ApA + gb); const float4* mA4 = (const float4*)(pd.preA + gb);
        float po[8], apo[8], mo[8], dl[8], r[8], z[8], pa[4], apa[4], ma[4], dla[4], ra[4], za[4];
        *(float4*)&po[0] = pO4[2 * q]; *(float4*)&po[4] = pO4[2 * q + 1];
        *(float4*)&apo[0] = ApO4[2 * q]; *(float4*)&apo[4] = ApO4[2 * q + 1];
        *(float4*)&mo[0] = mO4[2 * q]; *(float4*)&mo[4] = mO4[2 * q + 1];
        *(float4*)&dl[0] = dO4[2 * q]; *(float4*)&dl[4] = dO4[2 * q + 1];
        *(float4*)&r[0] = rO4[2 * q]; *(float4*)&r[4] = rO4[2 * q + 1];
        *(float4*)&z[0] = zO4[2 * q]; *(float4*)&z[4] = zO4[2 * q + 1];
        *(float4*)pa = pA4[q]; *(float4*)apa = ApA4[q]; *(float4*)ma = mA4[q];
        *(float4*)dla = dA4[q]; *(float4*)ra = rA4[q]; *(float4*)za = zA4[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!((fw >> (8 * k)) & F_ACT)) continue;
            dl[2 * k] = fmaf(alpha, po[2 * k], dl[2 * k]);
            dl[2 * k + 1] = fmaf(alpha, po[2 * k + 1], dl[2 * k + 1]);
            dla[k] = fmaf(alpha, pa[k], dla[k]);
            r[2 * k] = fmaf(-alpha, apo[2 * k], r[2 * k]);
            r[2 * k + 1] = fmaf(-alpha, apo[2 * k + 1], r[2 * k + 1]);
            ra[k] = fmaf(-alpha, apa[k], ra[k]);
            z[2 * k] = mo[2 * k] * r[2 * k];
            z[2 * k + 1] = mo[2 * k + 1] * r[2 * k + 1];
            za[k] = ma[k] * ra[k];
            d += (double)dot3(z[2 * k], z[2 * k + 1], za[k], r[2 * k], r[2 * k + 1], ra[k]);
        }
        dO4[2 * q] = *(float4*)&dl[0]; dO4[2 * q + 1] = *(float4*)&dl[4];
        rO4[2 * q] = *(float4*)&r[0]; rO4[2 * q + 1] = *(float4*)&r[4];
        zO4[2 * q] = *(float4*)&z[0]; zO4[2 * q + 1] = *(float4*)&z[4];
        dA4[q] = *(float4*)dla; rA4[q] = *(float4*)ra; zA4[q] = *(float4*)za;
    }
    block_reduce_fixed<1>(pd, b, blockIdx.x, gridDim.x, d, 0.0, pd.red + ((size_t)b * pd.nslots + (2 * l + 2)) * NSHARD, nullptr);
}

// ------------------------------------------------------------------------------------------------
// PCGLinearUpdate: X += delta on non-excluded vertices
// lag_l >= 0 (the lean streaming schedule, arap_stream.h): the delta images still lack the last PCG iteration's
// update delta += alpha_l p_l (every other iteration's was made by phase A of the iteration after it): made here.
__device__ __forceinline__ void gn_update_body(const PlanDev& pd, const VIdx& v, int lag_l)
{
    if (pd.res_err && *pd.res_err) return;          // the resident kernel gave up: leave X as it was (see PlanDev)
    if (!pd.tileact[(size_t)v.b * pd.tilesX * pd.tilesY + v.wg]) return;
    float alpha = 0.f;
    if (lag_l >= 0) {
        const double* rs = pd.red + (size_t)v.b * pd.nslots * NSHARD;
        const float rho = read_scalar(rs + (size_t)(2 * lag_l) * NSHARD);
        const float sigma = read_scalar(rs + (size_t)(2 * lag_l + 1) * NSHARD);
        if (sigma > 0.f) alpha = rho / sigma;
    }
    if (!v.in || !(pd.flags[v.g] & F_ACT)) return;
    const Slot sl = pd.slots[v.b];
    float2 d = pd.deltaO[v.g];
    float da = pd.deltaA[v.g];
    if (lag_l >= 0) {
        const float2 p = ((lag_l & 1) ? pd.pO0 : pd.pO1)[v.g];          // written by phase A of iteration lag_l
        const float pa = ((lag_l & 1) ? pd.pA0 : pd.pA1)[v.g];
        d.x = fmaf(alpha, p.x, d.x);
        d.y = fmaf(alpha, p.y, d.y);
        da = fmaf(alpha, pa, da);
    }
    float2 o = sl.O[v.i];
    o.x = o.x + d.x;
    o.y = o.y + d.y;
    sl.O[v.i] = o;
    sl.A[v.i] = sl.A[v.i] + da;
}

__global__ __launch_bounds__(TILE_X* TILE_Y) void k_gn_update(PlanDev pd, int lag_l)
{
    gn_update_body(pd, vidx(pd), lag_l);
}

// ------------------------------------------------------------------------------------------------
// computeCost: 0.5 * sum of squared residuals centred on each non-excluded vertex.  Self contained
// (reads Mask / Constraints / Angle directly) because it also runs at Init, before any k_gn_prep.
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_cost(PlanDev pd, int cost_index)
{
    const VIdx v = vidx(pd);
    const Slot sl = pd.slots[v.b];
    double d = 0.0;
    if (v.in && sl.M[v.i] == 0.0f) {
        const int W = pd.W, H = pd.H;
        const float wr = sl.wr, wf = sl.wf;
        const float2 cs = sincos_spec(sl.A[v.i]);
        const float2 Oi = sl.O[v.i], Ui = sl.U[v.i];
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int nx = v.x + (s == 0 ? 1 : (s == 1 ? -1 : 0));
            const int ny = v.y + (s == 2 ? 1 : (s == 3 ? -1 : 0));
            if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;
            const int n = nx + W * ny;
            if (sl.M[n] != 0.0f) continue;
            const float2 On = sl.O[n], Un = sl.U[n];
            const float dx = Ui.x - Un.x, dy = Ui.y - Un.y;
            const float ex = wr * ((Oi.x - On.x) - fmaf(cs.x, dx, -(cs.y * dy)));
            const float ey = wr * ((Oi.y - On.y) - fmaf(cs.y, dx, cs.x * dy));
            t = fmaf(ex, ex, t);
            t = fmaf(ey, ey, t);
        }
        const float2 Ci = sl.C[v.i];
        if (Ci.x >= 0.0f && Ci.y >= 0.0f) {
            const float fx = wf * (Oi.x - Ci.x), fy = wf * (Oi.y - Ci.y);
            t = fmaf(fx, fx, t);
            t = fmaf(fy, fy, t);
        }
        d = (double)(0.5f * t);
    }
    block_reduce_fixed<1>(pd, v.b, v.lb, v.nlb, d, 0.0, pd.costred + ((size_t)v.b * pd.ncost + cost_index) * NSHARD, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Kernel-level test entry points (ArapFlow_EvalJTF / ArapFlow_ApplyJTJ): raw gradient, diagonal and
// operator on caller images, using the same device code as the solver kernels above through a
// temporary plan (see arap_solver.hip).
__global__ __launch_bounds__(TILE_X* TILE_Y) void k_export_jtf(PlanDev pd, float2* gO, float* gA, float2* dO,
                                                                float* dA)
{
    // after k_gn_prep + k_gn_init on slot 0: g = -r ; D recovered from pre is lossy, so recompute D
    const VIdx v = vidx(pd);
    if (!v.in) return;
    const Slot sl = pd.slots[0];
    const unsigned f = pd.flags[v.g];
    float2 go = make_float2(0.f, 0.f), d_o = make_float2(0.f, 0.f);
    float ga = 0.f, d_a = 0.f;
    if (f & F_ACT) {
        const float2 r = pd.rO[v.g];
        go = make_float2(-r.x, -r.y);
        ga = -pd.rA[v.g];
        const float wr = sl.wr, wf = sl.wf;
        const float2 csi = pd.cs[v.g];
        const float2 Ui = sl.U[v.i];
        float DO = 0.f, DA = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!(f & (1u << s))) continue;
            const float2 Un = sl.U[v.i + noff(s, pd.W)];
            const float dx = Ui.x - Un.x, dy = Ui.y - Un.y;
            const float qx = fmaf(-csi.y, dx, -(csi.x * dy)), qy = fmaf(csi.x, dx, -(csi.y * dy));
            DO = DO + (wr * wr + wr * wr);
            DA = fmaf(wr * wr, fmaf(qx, qx, qy * qy), DA);
        }
        if (f & F_FIT) DO = fmaf(wf, wf, DO);
        d_o = make_float2(DO, DO);
        d_a = DA;
    }
    gO[v.i] = go; gA[v.i] = ga; dO[v.i] = d_o; dA[v.i] = d_a;
}

}  // namespace arap
