// arap_warp.h -- forward triangle rasteriser + flow emission on the GPU (gfx950).
//
// Reference: ARAP/warping/src/main.cpp:69-104 (PointInTriangleLK), :110-142 (rasterizeTriangle),
// :145-225 (Warp); the same code lives in ARAP/deformation/src/CombinedSolver.h:61-97,248-342, and
// flow = Offset - grid is CombinedSolver.h:352-366.
//
// The reference rasterises quads sequentially (y outer, x inner; triangle (00,01,10) then
// (10,01,11)) and later writes overwrite earlier ones.  Here every mesh vertex owns the quad to its
// lower right and rasterises both triangles concurrently; the sequential order is restored with a
// 64-bit atomicMax per covered pixel on the key  (triangle_index + 1) << 32 | r << 16 | g << 8 | b :
// the largest triangle index wins, which is exactly the last writer of the sequential loop, and the
// colour rides along in the low bits.  A second kernel unpacks the keys (and clears them for the
// next frame).  Per-pixel arithmetic is the reference's float expression, operation for operation
// (-ffp-contract=off), so the output is bit exact against the CPU code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace arap {

struct WarpJob {                    // one frame
    const float2* field;            // warp field [N] (Offset), or NULL when `flow_in` is given
    const float2* flow_in;          // flow [N]: warp field = (x,y) + flow (main.cpp:159-166), or NULL
    const uint8_t* rgb;             // [N][3] or NULL
    const uint8_t* mask;            // [N]  0 = object
    float2* flow_out;               // [N] or NULL: Offset - (x,y)
    unsigned long long* key;        // [N] scratch, all zero on entry
    uint8_t* out_rgb;               // [N][3] or NULL
    uint8_t* out_mask;              // [N]
};

__device__ __forceinline__ float2 warp_pos(const WarpJob& j, int x, int y, int i)
{
    if (j.field) return j.field[i];
    const float2 f = j.flow_in[i];
    return make_float2((float)x + f.x, (float)y + f.y);
}

__device__ __forceinline__ void raster_tri(const WarpJob& j, int W, int H, unsigned tri, float2 p0, float2 p1,
                                           float2 p2, const float c0[3], const float c1[3], const float c2[3])
{
    const float minx = floorf(fminf(p0.x, fminf(p1.x, p2.x)));
    const float miny = floorf(fminf(p0.y, fminf(p1.y, p2.y)));
    const float maxx = ceilf(fmaxf(p0.x, fmaxf(p1.x, p2.x)));
    const float maxy = ceilf(fmaxf(p0.y, fmaxf(p1.y, p2.y)));
    if (!(minx == minx && miny == miny && maxx == maxx && maxy == maxy)) return;
    const int xa = minx < 0.f ? 0 : (minx > (float)W ? W : (int)minx);
    const int ya = miny < 0.f ? 0 : (miny > (float)H ? H : (int)miny);
    for (int x = xa; x < W && (float)x <= maxx; ++x)
        for (int y = ya; y < H && (float)y <= maxy; ++y) {
            const float sx = (float)x, sy = (float)y;
            const float X0 = p0.x - sx * 1.0f, X1 = p1.x - sx * 1.0f, X2 = p2.x - sx * 1.0f;
            const float Y0 = p0.y - sy * 1.0f, Y1 = p1.y - sy * 1.0f, Y2 = p2.y - sy * 1.0f;
            float d01 = X0 * Y1 - Y0 * X1;
            float d12 = X1 * Y2 - Y1 * X2;
            float d20 = X2 * Y0 - Y2 * X0;
            if ((d01 < 0) & (d12 < 0) & (d20 < 0)) continue;
            const float OneOverD = 1.f / ((d01 + d12) + d20);
            d01 *= OneOverD;
            d12 *= OneOverD;
            d20 *= OneOverD;
            if (!(d01 >= 0 && d12 >= 0 && d20 >= 0)) continue;
            const float b0 = d12, b1 = d20, b2 = d01;
            unsigned rgbv = 0;
            if (j.rgb) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float v = (c0[k] * b0 + c1[k] * b1) + c2[k] * b2;
                    rgbv = (rgbv << 8) | (unsigned)(unsigned char)v;
                }
            }
            const unsigned long long key = ((unsigned long long)(tri + 1u) << 32) | rgbv;
            atomicMax(j.key + (x + (size_t)W * y), key);
        }
}

// grid = (ceil(W/64), ceil(H/4), njobs), block = (64,4)
__global__ __launch_bounds__(256) void k_warp_raster(const WarpJob* jobs, int W, int H)
{
    const WarpJob j = jobs[blockIdx.z];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int i = x + W * y;
    if (j.flow_out) {
        const float2 o = j.field[i];
        j.flow_out[i] = make_float2(o.x - (float)x, o.y - (float)y);
    }
    if (!(x + 1 < W && y + 1 < H)) return;
    const int i01 = i + 1, i10 = i + W, i11 = i + W + 1;
    if (!(j.mask[i] == 0 && j.mask[i10] == 0 && j.mask[i01] == 0 && j.mask[i11] == 0)) return;
    const float2 p00 = warp_pos(j, x, y, i), p01 = warp_pos(j, x + 1, y, i01);
    const float2 p10 = warp_pos(j, x, y + 1, i10), p11 = warp_pos(j, x + 1, y + 1, i11);
    float v00[3] = {0, 0, 0}, v01[3] = {0, 0, 0}, v10[3] = {0, 0, 0}, v11[3] = {0, 0, 0};
    if (j.rgb) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v00[k] = (float)j.rgb[3 * (size_t)i + k];
            v01[k] = (float)j.rgb[3 * (size_t)i01 + k];
            v10[k] = (float)j.rgb[3 * (size_t)i10 + k];
            v11[k] = (float)j.rgb[3 * (size_t)i11 + k];
        }
    }
    raster_tri(j, W, H, 2u * (unsigned)i, p00, p01, p10, v00, v01, v10);
    raster_tri(j, W, H, 2u * (unsigned)i + 1u, p10, p01, p11, v10, v01, v11);
}

// grid = (ceil(N/256), 1, njobs), block = 256
__global__ __launch_bounds__(256) void k_warp_resolve(const WarpJob* jobs, int N)
{
    const WarpJob j = jobs[blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const unsigned long long k = j.key[i];
    j.key[i] = 0ull;
    if (j.out_rgb) {
        j.out_rgb[3 * (size_t)i + 0] = (uint8_t)((k >> 16) & 0xffu);
        j.out_rgb[3 * (size_t)i + 1] = (uint8_t)((k >> 8) & 0xffu);
        j.out_rgb[3 * (size_t)i + 2] = (uint8_t)(k & 0xffu);
    }
    j.out_mask[i] = k ? 255 : 0;
}

}  // namespace arap
