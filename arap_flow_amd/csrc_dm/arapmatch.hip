// arapmatch.hip -- libarapmatch.so: dense matching of a frame pair for gfx950 (include/arap_match.h).
//
// Replaces the external DeepMatching 1.2.2 binary the reference shells out to (/root/reference/para_gen.py:227-240).
// The binary's source is not in the reference tree: these kernels implement the published algorithm (Revaud et al.,
// IJCV 2016) exactly as oracle/dm_oracle.py restates it -- same coordinate conventions, same order of operations where
// float32 rounding could change an argmax; that file is the specification, cited per kernel below.
//
//   k_gray_half, k_blur_h/v, k_orient, k_normalize   pixel descriptors                       dm_oracle.descriptors  (Sec. 3.1)
//   k_corr0                                          bottom-level correlation maps, MFMA     dm_oracle.level0       (Sec. 3.2)
//   k_level_up                                       max-pool + average + x^1.4              dm_oracle.level_up     (Alg. 1)
//   k_argmax, k_bt_step                              entry points, backtracking              dm_oracle.backtrack    (Alg. 2)
//   k_bin, k_emit                                    one match per 4x4 cell of frame 2       dm_oracle.matches      (Sec. 3.3)
//
// k_corr0 is the one dense contraction of the product: for a row of 32 atomic patches (M = 32) and 32 consecutive
// placements in frame 2 (N = 32) the 4x4x9 = 144 products per (patch, placement) are summed by 72
// v_mfma_f32_32x32x2_f32 (float32 in, float32 accumulate: the published arithmetic is float32).  The patches'
// descriptors sit in registers (the A operand: 72 VGPRs per lane), frame 2's descriptor rows in an LDS ring of four rows
// that advances one row per vertical displacement; a patch's own window |dx| <= r is a band of the 32 x N products, the
// rest is computed and dropped (44 % useful at r = 50).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/arap_match.h"

#define HC(call)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "arapmatch: HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                                 \
        }                                                                                            \
    } while (0)

namespace dm {

constexpr int PATCH = 4, NCH = 9, KDIM = PATCH * PATCH * NCH;      // 144
constexpr float LAMBDA = 1.4f, NINTH = 0.3f, SIGMOID = 0.2f;

// float32(exp(-x^2/2) / sum), x = -3..3; float32(cos(k pi/4)), float32(sin(k pi/4)): the values numpy gives the oracle
__constant__ float c_g7[7] = {0x1.228634p-8f, 0x1.ba69eap-5f, 0x1.efb0b0p-3f, 0x1.98a0a2p-2f, 0x1.efb0b0p-3f, 0x1.ba69eap-5f, 0x1.228634p-8f};
__constant__ float c_cos[8] = {0x1.0p+0f, 0x1.6a09e6p-1f, 0x1.1a6264p-54f, -0x1.6a09e6p-1f, -0x1.0p+0f, -0x1.6a09e6p-1f, -0x1.a79394p-53f, 0x1.6a09e6p-1f};
__constant__ float c_sin[8] = {0.0f, 0x1.6a09e6p-1f, 0x1.0p+0f, 0x1.6a09e6p-1f, 0x1.1a6264p-53f, -0x1.6a09e6p-1f, -0x1.0p+0f, -0x1.6a09e6p-1f};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// gray = (R + G + B) / 3, then the mean of every 2 x 2 block: half resolution
__global__ __launch_bounds__(256) void k_gray_half(const uint8_t* __restrict__ rgb, float* __restrict__ out, int W, int h, int w)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint8_t* p = rgb + ((size_t)(2 * y + (q >> 1)) * W + 2 * x + (q & 1)) * 3;
        g[q] = (((float)p[0] + (float)p[1]) + (float)p[2]) * (1.0f / 3.0f);
    }
    out[i] = (((g[0] + g[1]) + g[2]) + g[3]) * 0.25f;
}

// 7-tap Gaussian along x / y, replicated borders, C interleaved channels; SIG: the sigmoid 2 / (1 + exp(-0.2 v)) - 1 on the way out
__global__ __launch_bounds__(256) void k_blur_h(const float* __restrict__ in, float* __restrict__ out, int h, int w, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w * C) return;
    const int c = i % C, px = i / C, y = px / w, x = px - y * w;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 7; ++t) acc = acc + c_g7[t] * in[((size_t)y * w + clampi(x + t - 3, 0, w - 1)) * C + c];
    out[i] = acc;
}
template <bool SIG>
__global__ __launch_bounds__(256) void k_blur_v(const float* __restrict__ in, float* __restrict__ out, int h, int w, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w * C) return;
    const int c = i % C, px = i / C, y = px / w, x = px - y * w;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 7; ++t) acc = acc + c_g7[t] * in[((size_t)clampi(y + t - 3, 0, h - 1) * w + x) * C + c];
    if (SIG) acc = 2.0f / (1.0f + expf(-SIGMOID * acc)) - 1.0f;
    out[i] = acc;
}

// centred differences (replicated borders), projected on 8 orientations, negative parts cut
__global__ __launch_bounds__(256) void k_orient(const float* __restrict__ sm, float* __restrict__ ori, int h, int w)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const float gx = (sm[y * w + clampi(x + 1, 0, w - 1)] - sm[y * w + clampi(x - 1, 0, w - 1)]) * 0.5f;
    const float gy = (sm[clampi(y + 1, 0, h - 1) * w + x] - sm[clampi(y - 1, 0, h - 1) * w + x]) * 0.5f;
#pragma unroll
    for (int k = 0; k < 8; ++k) ori[(size_t)i * 8 + k] = fmaxf(0.0f, gx * c_cos[k] + gy * c_sin[k]);
}

// ninth channel 0.3, unit L2 norm per pixel
__global__ __launch_bounds__(256) void k_normalize(const float* __restrict__ ori, float* __restrict__ desc, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v[NCH], s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = ori[(size_t)i * 8 + k]; s = s + v[k] * v[k]; }
    v[8] = NINTH;
    s = s + NINTH * NINTH;
    const float nrm = sqrtf(s);
#pragma unroll
    for (int k = 0; k < NCH; ++k) desc[(size_t)i * NCH + k] = v[k] / nrm;
}

// ---- bottom-level correlation maps (dm_oracle.level0) --------------------------------------------------------------
// grid (ceil(gw / 32), gh), 256 threads.  Workgroup = patch row j, patches i0 .. i0 + 31.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_corr0(const float* __restrict__ d1, const float* __restrict__ d2, float* __restrict__ maps,
                                               int h, int w, int gh, int gw, int r)
{
    extern __shared__ float smem[];
    const int j = blockIdx.y, i0 = blockIdx.x * 32;
    const int S = 2 * r + 1, NP = 128 + 2 * r, Wwin = NP + 3;       // placements of the tile, pixels per staged row
    float* const sA = smem;                                          // [4][128][9] the patches' pixels
    float* const ring = smem + 4 * 128 * NCH;                        // [4][Wwin][9] four rows of frame 2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, kk = lane >> 5;
    for (int idx = tid; idx < 4 * 128 * NCH; idx += 256) {
        const int b = idx / (128 * NCH), rem = idx - b * (128 * NCH), px = rem / NCH, c = rem - px * NCH;
        const int x = 4 * i0 + px, y = 4 * j + b;
        sA[idx] = (x < w && y < h) ? d1[((size_t)y * w + x) * NCH + c] : 0.f;
    }
    __syncthreads();
    // A operand of v_mfma_f32_32x32x2_f32: lane l holds A[row = l & 31][k = l >> 5]; step s covers k = 2 s, 2 s + 1
    float areg[KDIM / 2];
#pragma unroll
    for (int s = 0; s < KDIM / 2; ++s) {
        const int k = 2 * s + kk, pix = k / NCH, c = k - pix * NCH, b = pix >> 2, a = pix & 3;
        areg[s] = sA[(b * 128 + 4 * m + a) * NCH + c];
    }
    const int xs = 4 * i0 - r;
    auto load_row = [&](int y) {                                     // image row y of frame 2 into ring slot y & 3
        float* dst = ring + (size_t)(y & 3) * Wwin * NCH;
        const bool yin = y >= 0 && y < h;
        for (int idx = tid; idx < Wwin * NCH; idx += 256) {
            const int p = idx / NCH, c = idx - p * NCH, x = xs + p;
            dst[idx] = (yin && x >= 0 && x < w) ? d2[((size_t)y * w + x) * NCH + c] : 0.f;
        }
    };
    // the vertical displacements are split over gridDim.z workgroups (two per patch row: two wavefronts per SIMD, so that
    // one's LDS reads and epilogue stores hide behind the other's MFMAs)
    const int per = (S + (int)gridDim.z - 1) / (int)gridDim.z;
    const int dy0 = -r + (int)blockIdx.z * per, dy1 = min(r, dy0 + per - 1);
    for (int b = 0; b < 3; ++b) load_row(4 * j + dy0 + b);
    const int ntiles = (NP + 31) / 32;
    for (int dy = dy0; dy <= dy1; ++dy) {
        load_row(4 * j + dy + 3);
        __syncthreads();
        int rowoff[4];                                               // ring offset of the patch row b under this dy
#pragma unroll
        for (int b = 0; b < 4; ++b) rowoff[b] = ((4 * j + dy + b) & 3) * Wwin * NCH;
        for (int nt = wave; nt < ntiles; nt += 4) {
            const int p = nt * 32 + m;                               // this lane's placement (B column)
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // the 72 B values of this lane first (all LDS reads in flight together), then 72 MFMAs back to back
            float breg[KDIM / 2];
#pragma unroll
            for (int s = 0; s < KDIM / 2; ++s) {
                // B[k][n] = frame 2 at (row b, pixel p + a, channel c), k = (4 b + a) 9 + c: compile-time for either half of the lanes
                const int k0 = 2 * s, pix0 = k0 / NCH, c0 = k0 - pix0 * NCH, b0 = pix0 >> 2, a0 = pix0 & 3;
                const int k1 = 2 * s + 1, pix1 = k1 / NCH, c1 = k1 - pix1 * NCH, b1 = pix1 >> 2, a1 = pix1 & 3;
                const int o0 = rowoff[b0] + (p + a0) * NCH + c0, o1 = rowoff[b1] + (p + a1) * NCH + c1;
                breg[s] = (p < NP) ? ring[kk ? o1 : o0] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < KDIM / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], breg[s], acc, 0, 0, 0);
            // C: lane l holds column l & 31, rows (v & 3) + 8 (v >> 2) + 4 (l >> 5) in register v
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int mrow = (v & 3) + 8 * (v >> 2) + 4 * kk;
                const int dx = p - r - 4 * mrow, i = i0 + mrow;
                if (i < gw && dx >= -r && dx <= r)
                    maps[(((size_t)j * gw + i) * S + (dy + r)) * S + (dx + r)] = acc[v] * (1.0f / 16.0f);
            }
        }
        __syncthreads();
    }
}

// ---- one level up (dm_oracle.level_up): 3x3 max-pool + subsample of the four children, mean, x^1.4 ----------------
__global__ __launch_bounds__(256) void k_level_up(const float* __restrict__ below, float* __restrict__ out, int level, int gwb, int Sb,
                                                  int o, int nh, int nw, int S2)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)nh * nw * S2 * S2;
    if (idx >= total) return;
    const int kx = (int)(idx % S2), ky = (int)((idx / S2) % S2);
    const int pidx = (int)(idx / ((size_t)S2 * S2)), J = pidx / nw, I = pidx - J * nw;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = q >> 1, a = q & 1;
        const int cj = level == 0 ? J + b : 2 * J + 2 * b, ci = level == 0 ? I + a : 2 * I + 2 * a;
        const float* mp = below + ((size_t)cj * gwb + ci) * Sb * Sb;
        float mx = -INFINITY;
#pragma unroll
        for (int u = -1; u <= 1; ++u)
#pragma unroll
            for (int v = -1; v <= 1; ++v) {
                const int y = 2 * ky + o + u, x = 2 * kx + o + v;
                if (y >= 0 && y < Sb && x >= 0 && x < Sb) mx = fmaxf(mx, mp[y * Sb + x]);
            }
        acc = acc + mx;
    }
    acc = acc * 0.25f;
    out[idx] = powf(fmaxf(acc, 0.0f), LAMBDA);
}

// ---- entry points: the first maximum of every patch's map (one wavefront per patch) -------------------------------
struct Ent { int j, i, ky, kx; float s; };

__global__ __launch_bounds__(256) void k_argmax(const float* __restrict__ maps, int npatch, int nw, int S, Ent* __restrict__ out)
{
    const int pidx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pidx >= npatch) return;
    const float* mp = maps + (size_t)pidx * S * S;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < S * S; c += 64) {
        const float v = mp[c];
        if (v > bv) { bv = v; bi = c; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ov = __shfl_xor(bv, d);
        const int oi = __shfl_xor(bi, d);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) out[pidx] = Ent{pidx / nw, pidx % nw, bi / S, bi % S, bv};
}

// one step down: entry (patch of level lv, cell) -> its four children of level lv - 1, each with the best cell of its 3x3
// pooling window (first maximum in (u, v) order), scores added.  lv == 1: the children are atomic patches: the candidate
// goes to best[patch] = max over {score bits << 32 | ~cell code} (ties: the smaller cell index)
__global__ __launch_bounds__(256) void k_bt_step(const Ent* __restrict__ in, int n, int lv, const float* __restrict__ below, int gwb, int Sb,
                                                 int o, Ent* __restrict__ out, unsigned long long* __restrict__ best)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 4 * n) return;
    const Ent e = in[t >> 2];
    const int q = t & 3, b = q >> 1, a = q & 1;
    const int cj = lv == 1 ? e.j + b : 2 * e.j + 2 * b, ci = lv == 1 ? e.i + a : 2 * e.i + 2 * a;
    const float* mp = below + ((size_t)cj * gwb + ci) * Sb * Sb;
    float bv = -INFINITY;
    int by = 0, bx = 0;
    for (int u = -1; u <= 1; ++u)
        for (int v = -1; v <= 1; ++v) {
            const int y = 2 * e.ky + o + u, x = 2 * e.kx + o + v;
            if (y >= 0 && y < Sb && x >= 0 && x < Sb) {
                const float val = mp[y * Sb + x];
                if (val > bv) { bv = val; by = y; bx = x; }
            }
        }
    const float s = e.s + bv;
    if (lv == 1) {
        if (s > 0.0f) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(s) << 32) | (0xffffffffu - (unsigned)(by * Sb + bx));
            atomicMax(best + (size_t)cj * gwb + ci, key);
        }
    } else {
        out[t] = Ent{cj, ci, by, bx, s};
    }
}

// one match per 4x4 cell of frame 2: the better score wins (ties: the smaller patch index)
__global__ __launch_bounds__(256) void k_bin(const unsigned long long* __restrict__ best, int gh, int gw, int S0, int c0, int h, int w,
                                             unsigned long long* __restrict__ bins, int bw)
{
    const int pidx = blockIdx.x * 256 + threadIdx.x;
    if (pidx >= gh * gw) return;
    const unsigned long long key = best[pidx];
    if (key == 0ull) return;
    const unsigned code = 0xffffffffu - (unsigned)(key & 0xffffffffull);
    const int ky = (int)(code / (unsigned)S0), kx = (int)(code % (unsigned)S0);
    const int j = pidx / gw, i = pidx - j * gw;
    const int x2 = PATCH * i + 2 + kx - c0, y2 = PATCH * j + 2 + ky - c0;
    if (x2 < 0 || x2 >= w || y2 < 0 || y2 >= h) return;
    const unsigned long long bk = (key & 0xffffffff00000000ull) | (0xffffffffu - (unsigned)pidx);
    atomicMax(bins + (size_t)(y2 / PATCH) * bw + (x2 / PATCH), bk);
}

// rows x1 y1 x2 y2 score valid, one per atomic patch (valid = 1: the patch won its cell of frame 2)
__global__ __launch_bounds__(256) void k_emit(const unsigned long long* __restrict__ best, int gh, int gw, int S0, int c0, int h, int w,
                                              const unsigned long long* __restrict__ bins, int bw, float* __restrict__ rows)
{
    const int pidx = blockIdx.x * 256 + threadIdx.x;
    if (pidx >= gh * gw) return;
    float* o = rows + (size_t)pidx * 6;
    o[5] = 0.f;
    const unsigned long long key = best[pidx];
    if (key == 0ull) return;
    const unsigned code = 0xffffffffu - (unsigned)(key & 0xffffffffull);
    const int ky = (int)(code / (unsigned)S0), kx = (int)(code % (unsigned)S0);
    const int j = pidx / gw, i = pidx - j * gw;
    const int x2 = PATCH * i + 2 + kx - c0, y2 = PATCH * j + 2 + ky - c0;
    if (x2 < 0 || x2 >= w || y2 < 0 || y2 >= h) return;
    const unsigned long long bk = (key & 0xffffffff00000000ull) | (0xffffffffu - (unsigned)pidx);
    if (bins[(size_t)(y2 / PATCH) * bw + (x2 / PATCH)] != bk) return;
    o[0] = (float)(2 * (PATCH * i + 2)); o[1] = (float)(2 * (PATCH * j + 2));
    o[2] = (float)(2 * x2); o[3] = (float)(2 * y2);
    o[4] = __uint_as_float((unsigned)(key >> 32));
    o[5] = 1.f;
}

struct Level { int nh, nw, S, c, o; float* maps; };       // o: subsampling offset used to build THIS level from the one below

}  // namespace dm

using namespace dm;

struct ArapMatch {
    int W, H, w, h, r;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, c0 = nullptr, c1 = nullptr;     // whole Run; k_corr0 alone
    uint8_t* d_rgb = nullptr;             // one frame at a time
    float *gray = nullptr, *t0 = nullptr, *t1 = nullptr, *ori = nullptr, *ori2 = nullptr;
    float* desc[2] = {nullptr, nullptr};
    std::vector<Level> lv;
    Ent* ent[2] = {nullptr, nullptr};
    unsigned long long *best = nullptr, *bins = nullptr;
    float* rows = nullptr;
    std::vector<float> hrows;
    int bw = 0, bh = 0;
    float last_ms = 0.f, last_corr_ms = 0.f;
    size_t corr_lds = 0;
};

static inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

static void describe(ArapMatch* m, const uint8_t* rgb_host, int which)
{
    const int h = m->h, w = m->w;
    const size_t n = (size_t)h * w;
    HC(hipMemcpyAsync(m->d_rgb, rgb_host, (size_t)m->W * m->H * 3, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_gray_half, dim3(blocks(n)), dim3(256), 0, m->stream, m->d_rgb, m->gray, m->W, h, w);
    hipLaunchKernelGGL(k_blur_h, dim3(blocks(n)), dim3(256), 0, m->stream, m->gray, m->t0, h, w, 1);
    hipLaunchKernelGGL(k_blur_v<false>, dim3(blocks(n)), dim3(256), 0, m->stream, m->t0, m->t1, h, w, 1);
    hipLaunchKernelGGL(k_orient, dim3(blocks(n)), dim3(256), 0, m->stream, m->t1, m->ori, h, w);
    hipLaunchKernelGGL(k_blur_h, dim3(blocks(n * 8)), dim3(256), 0, m->stream, m->ori, m->ori2, h, w, 8);
    hipLaunchKernelGGL(k_blur_v<true>, dim3(blocks(n * 8)), dim3(256), 0, m->stream, m->ori2, m->ori, h, w, 8);
    hipLaunchKernelGGL(k_blur_h, dim3(blocks(n * 8)), dim3(256), 0, m->stream, m->ori, m->ori2, h, w, 8);
    hipLaunchKernelGGL(k_blur_v<false>, dim3(blocks(n * 8)), dim3(256), 0, m->stream, m->ori2, m->ori, h, w, 8);
    hipLaunchKernelGGL(k_normalize, dim3(blocks(n)), dim3(256), 0, m->stream, m->ori, m->desc[which], (int)n);
}

extern "C" {

ArapMatch* ArapMatch_Create(unsigned W, unsigned H, unsigned ngh_rad)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "arapmatch: no HIP device available; this library has no CPU fallback\n");
        return nullptr;
    }
    if (W < 16 || H < 16 || (uint64_t)W * H > (1ull << 28)) return nullptr;
    ArapMatch* m = new ArapMatch();
    m->W = (int)W; m->H = (int)H; m->w = (int)W / 2; m->h = (int)H / 2;
    m->r = (int)(ngh_rad >> 1);
    if (m->r < 1) m->r = 1;
    if (m->r > 96) m->r = 96;                                        // LDS ring of k_corr0: (131 + 2 r) pixels per row
    HC(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    HC(hipEventCreate(&m->e0));
    HC(hipEventCreate(&m->e1));
    HC(hipEventCreate(&m->c0));
    HC(hipEventCreate(&m->c1));
    const size_t n = (size_t)m->h * m->w;
    HC(hipMalloc(&m->d_rgb, (size_t)W * H * 3));
    HC(hipMalloc(&m->gray, n * 4)); HC(hipMalloc(&m->t0, n * 4)); HC(hipMalloc(&m->t1, n * 4));
    HC(hipMalloc(&m->ori, n * 8 * 4)); HC(hipMalloc(&m->ori2, n * 8 * 4));
    HC(hipMalloc(&m->desc[0], n * NCH * 4)); HC(hipMalloc(&m->desc[1], n * NCH * 4));
    // pyramid geometry (dm_oracle.pyramid / pool_geometry / children_of)
    Level L0{m->h / PATCH, m->w / PATCH, 2 * m->r + 1, m->r, 0, nullptr};
    m->lv.push_back(L0);
    for (;;) {
        const Level& b = m->lv.back();
        const int level = (int)m->lv.size() - 1;
        int nh, nw;
        if (level == 0) { nh = b.nh - 1; nw = b.nw - 1; }
        else { nh = b.nh >= 3 ? (b.nh - 3) / 2 + 1 : 0; nw = b.nw >= 3 ? (b.nw - 3) / 2 + 1 : 0; }
        if (nh <= 0 || nw <= 0) break;
        const int o = b.c & 1, S2 = (b.S - 1 - o) / 2 + 1, c2 = (b.c - o) / 2;
        if (S2 < 1) break;
        m->lv.push_back(Level{nh, nw, S2, c2, o, nullptr});
        if (S2 == 1) break;
    }
    for (Level& l : m->lv) HC(hipMalloc(&l.maps, (size_t)l.nh * l.nw * l.S * l.S * 4));
    const size_t n0 = (size_t)m->lv[0].nh * m->lv[0].nw;
    HC(hipMalloc(&m->ent[0], (n0 + 64) * sizeof(Ent)));
    HC(hipMalloc(&m->ent[1], (n0 + 64) * sizeof(Ent)));
    HC(hipMalloc(&m->best, n0 * 8));
    m->bw = m->w / PATCH + 1; m->bh = m->h / PATCH + 1;
    HC(hipMalloc(&m->bins, (size_t)m->bw * m->bh * 8));
    HC(hipMalloc(&m->rows, n0 * 6 * 4));
    m->hrows.resize(n0 * 6);
    m->corr_lds = (size_t)(4 * 128 * NCH + 4 * (131 + 2 * m->r) * NCH) * sizeof(float);
    HC(hipFuncSetAttribute((const void*)k_corr0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->corr_lds));
    return m;
}

void ArapMatch_Free(ArapMatch* m)
{
    if (!m) return;
    (void)hipStreamSynchronize(m->stream);
    for (Level& l : m->lv) (void)hipFree(l.maps);
    (void)hipFree(m->d_rgb); (void)hipFree(m->gray); (void)hipFree(m->t0); (void)hipFree(m->t1);
    (void)hipFree(m->ori); (void)hipFree(m->ori2); (void)hipFree(m->desc[0]); (void)hipFree(m->desc[1]);
    (void)hipFree(m->ent[0]); (void)hipFree(m->ent[1]); (void)hipFree(m->best); (void)hipFree(m->bins); (void)hipFree(m->rows);
    (void)hipEventDestroy(m->e0); (void)hipEventDestroy(m->e1); (void)hipEventDestroy(m->c0); (void)hipEventDestroy(m->c1);
    (void)hipStreamDestroy(m->stream);
    delete m;
}

int ArapMatch_Run(ArapMatch* m, const uint8_t* rgb1, const uint8_t* rgb2, float* out, unsigned cap)
{
    if (!m || !rgb1 || !rgb2 || (!out && cap)) return -1;
    hipStream_t s = m->stream;
    HC(hipEventRecord(m->e0, s));
    describe(m, rgb1, 0);
    describe(m, rgb2, 1);
    const Level& L0 = m->lv[0];
    const int gh = L0.nh, gw = L0.nw;
    HC(hipEventRecord(m->c0, s));
    hipLaunchKernelGGL(k_corr0, dim3((gw + 31) / 32, gh, 2), dim3(256), m->corr_lds, s, m->desc[0], m->desc[1], L0.maps, m->h, m->w,
                       gh, gw, m->r);
    HC(hipEventRecord(m->c1, s));
    for (size_t l = 1; l < m->lv.size(); ++l) {
        const Level& b = m->lv[l - 1];
        const Level& t = m->lv[l];
        const size_t total = (size_t)t.nh * t.nw * t.S * t.S;
        hipLaunchKernelGGL(k_level_up, dim3(blocks(total)), dim3(256), 0, s, b.maps, t.maps, (int)l - 1, b.nw, b.S, t.o, t.nh, t.nw, t.S);
    }
    const size_t n0 = (size_t)gh * gw;
    HC(hipMemsetAsync(m->best, 0, n0 * 8, s));
    HC(hipMemsetAsync(m->bins, 0, (size_t)m->bw * m->bh * 8, s));
    for (size_t top = 1; top < m->lv.size(); ++top) {
        const Level& t = m->lv[top];
        int n = t.nh * t.nw, cur = 0;
        hipLaunchKernelGGL(k_argmax, dim3((n + 3) / 4), dim3(256), 0, s, t.maps, n, t.nw, t.S, m->ent[0]);
        for (size_t l = top; l >= 1; --l) {
            const Level& b = m->lv[l - 1];
            hipLaunchKernelGGL(k_bt_step, dim3(blocks((size_t)4 * n)), dim3(256), 0, s, m->ent[cur], n, (int)l, b.maps, b.nw, b.S,
                               m->lv[l].o, m->ent[cur ^ 1], m->best);
            n *= 4;
            cur ^= 1;
        }
    }
    hipLaunchKernelGGL(k_bin, dim3(blocks(n0)), dim3(256), 0, s, m->best, gh, gw, L0.S, L0.c, m->h, m->w, m->bins, m->bw);
    hipLaunchKernelGGL(k_emit, dim3(blocks(n0)), dim3(256), 0, s, m->best, gh, gw, L0.S, L0.c, m->h, m->w, m->bins, m->bw, m->rows);
    HC(hipMemcpyAsync(m->hrows.data(), m->rows, n0 * 6 * 4, hipMemcpyDeviceToHost, s));
    HC(hipEventRecord(m->e1, s));
    HC(hipStreamSynchronize(s));
    HC(hipEventElapsedTime(&m->last_ms, m->e0, m->e1));
    HC(hipEventElapsedTime(&m->last_corr_ms, m->c0, m->c1));
    unsigned cnt = 0;
    for (size_t p = 0; p < n0; ++p) {
        const float* rr = m->hrows.data() + p * 6;
        if (rr[5] == 0.f) continue;
        if (cnt < cap) {
            float* o = out + (size_t)cnt * 6;
            o[0] = rr[0]; o[1] = rr[1]; o[2] = rr[2]; o[3] = rr[3]; o[4] = rr[4]; o[5] = (float)cnt;
        }
        ++cnt;
    }
    return (int)cnt;
}

int ArapMatch_Levels(ArapMatch* m) { return m ? (int)m->lv.size() : -1; }

int ArapMatch_LevelInfo(ArapMatch* m, int level, int* nh, int* nw, int* S, int* c)
{
    if (!m || level < 0 || level >= (int)m->lv.size()) return -1;
    const Level& l = m->lv[level];
    if (nh) *nh = l.nh;
    if (nw) *nw = l.nw;
    if (S) *S = l.S;
    if (c) *c = l.c;
    return 0;
}

int ArapMatch_GetLevel(ArapMatch* m, int level, float* host)
{
    if (!m || !host || level < 0 || level >= (int)m->lv.size()) return -1;
    const Level& l = m->lv[level];
    HC(hipStreamSynchronize(m->stream));
    HC(hipMemcpy(host, l.maps, (size_t)l.nh * l.nw * l.S * l.S * 4, hipMemcpyDeviceToHost));
    return 0;
}

int ArapMatch_GetDescriptors(ArapMatch* m, int which, float* host)
{
    if (!m || !host || which < 0 || which > 1) return -1;
    HC(hipStreamSynchronize(m->stream));
    HC(hipMemcpy(host, m->desc[which], (size_t)m->h * m->w * NCH * 4, hipMemcpyDeviceToHost));
    return 0;
}

float ArapMatch_LastRunMs(ArapMatch* m) { return m ? m->last_ms : -1.f; }
float ArapMatch_LastCorrMs(ArapMatch* m) { return m ? m->last_corr_ms : -1.f; }

}  // extern "C"
