/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see arap_oracle_impl.h header).  Build: oracle/Makefile.
 *
 * Contents
 *   arap_oracle_impl.h x2 : the ARAP GN/PCG solve in float32 (_f32) and float64 (_f64)
 *   oracle_warp           : forward triangle rasteriser of ARAP/warping/src/main.cpp:69-225
 *                           (= ARAP/deformation/src/CombinedSolver.h:61-97,248-342)
 *   oracle_flow_from_offset: CombinedSolver.h:352-366
 *
 * Parity pinning (tests/test_oracle.py, tests/test_warp_oracle.py):
 *   - evalJTF / applyJTJ vs finite differences of oracle_residuals (float64)
 *   - full 19/8/400 schedule vs the reference golden ARAP/warping/cat512_iFlo.flo (tier T4)
 *   - oracle_warp vs the reference golden cat512_wRGB.png / cat512_wMsk.png and vs the reference's
 *     own warp_image compiled from its sources into oracle/_ref/ (bit exact)
 * Compiled with -ffp-contract=off: every operator below is a single IEEE-754 operation; fused multiply-adds are
 * written explicitly (fmaf/fma, compiled to hardware FMA with -mfma) at exactly the sites where the HIP kernels fuse.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * arap_sincos_spec: cos/sin of a float-valued angle from IEEE double +,-,*,fma,rint only, in a fixed
 * operation order, so that a second implementation with the same operation list (the HIP kernel,
 * arap_flow_amd/csrc/arap_device.h) returns the same bits.  The reference calls CUDA libdevice
 * __nv_cosf/__nv_sinf (ARAP/API/src/util.t:160-174), accurate to ~1-2 ulp; this routine is
 * accurate to < 1e-13 before the final rounding to float, i.e. within the reference's own error.
 * Cody-Waite reduction with the fdlibm split of pi/2, Taylor polynomials on [-pi/4, pi/4].
 * ------------------------------------------------------------------------------------------- */
void arap_sincos_spec(double a, double *c, double *s)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00; /* 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11; /* pi/2 - pio2_hi */
    double k = rint(a * two_over_pi);
    double r = fma(-k, pio2_lo, fma(-k, pio2_hi, a));
    double r2 = r * r;
    /* sin r = r (1 + r2 (S1 + r2 (S2 + ...))) ; cos r = 1 + r2 (C1 + r2 (C2 + ...)), Horner with fused steps */
    double ps = -1.0 / 1307674368000.0;            /* -1/15! */
    ps = fma(ps, r2, 1.0 / 6227020800.0);          /*  1/13! */
    ps = fma(ps, r2, -1.0 / 39916800.0);           /* -1/11! */
    ps = fma(ps, r2, 1.0 / 362880.0);              /*  1/9!  */
    ps = fma(ps, r2, -1.0 / 5040.0);               /* -1/7!  */
    ps = fma(ps, r2, 1.0 / 120.0);                 /*  1/5!  */
    ps = fma(ps, r2, -1.0 / 6.0);                  /* -1/3!  */
    double sr = fma(r, r2 * ps, r);
    double pc = 1.0 / 20922789888000.0;            /*  1/16! */
    pc = fma(pc, r2, -1.0 / 87178291200.0);        /* -1/14! */
    pc = fma(pc, r2, 1.0 / 479001600.0);           /*  1/12! */
    pc = fma(pc, r2, -1.0 / 3628800.0);            /* -1/10! */
    pc = fma(pc, r2, 1.0 / 40320.0);               /*  1/8!  */
    pc = fma(pc, r2, -1.0 / 720.0);                /* -1/6!  */
    pc = fma(pc, r2, 1.0 / 24.0);                  /*  1/4!  */
    pc = fma(pc, r2, -0.5);                        /* -1/2!  */
    double cr = fma(r2, pc, 1.0);
    long long q = (long long)k;
    switch ((int)(q & 3)) {
    case 0: *c = cr; *s = sr; break;
    case 1: *c = -sr; *s = cr; break;
    case 2: *c = -cr; *s = -sr; break;
    default: *c = sr; *s = -cr; break;
    }
}

/* cos/sin used by the standalone kernel-level entry points (oracle_evalJTF, oracle_applyJTJ,
 * oracle_cost, oracle_residuals): 0 = libm, 1 = arap_sincos_spec */
static int g_oracle_trig = 0;
void oracle_set_trig(int t) { g_oracle_trig = t; }

#define REAL float
#define SUF _f32
#include "arap_oracle_impl.h"
#undef REAL
#undef SUF

#define REAL double
#define SUF _f64
#include "arap_oracle_impl.h"
#undef REAL
#undef SUF

/* ---------------------------------------------------------------------------------------------
 * Warp: ARAP/warping/src/main.cpp
 * ------------------------------------------------------------------------------------------- */

/* PointInTriangleLK, main.cpp:69-104 with w0 = w1 = w2 = 1 (the only way it is called, :129-131) */
static int point_in_triangle_lk(float x0, float y0, float x1, float y1, float x2, float y2, float sx,
                                float sy, float *wt0, float *wt1, float *wt2)
{
    float X0 = x0 - sx * 1.0f, X1 = x1 - sx * 1.0f, X2 = x2 - sx * 1.0f;
    float Y0 = y0 - sy * 1.0f, Y1 = y1 - sy * 1.0f, Y2 = y2 - sy * 1.0f;
    float d01 = X0 * Y1 - Y0 * X1;
    float d12 = X1 * Y2 - Y1 * X2;
    float d20 = X2 * Y0 - Y2 * X0;
    if ((d01 < 0) & (d12 < 0) & (d20 < 0)) return 0; /* backfacing */
    float OneOverD = 1.f / (d01 + d12 + d20);
    d01 *= OneOverD;
    d12 *= OneOverD;
    d20 *= OneOverD;
    *wt0 = d12;
    *wt1 = d20;
    *wt2 = d01;
    return (d01 >= 0 && d12 >= 0 && d20 >= 0);
}

/* rasterizeTriangle, main.cpp:110-142.  Bounds: floor(min) .. ceil(max) inclusive, compared as
 * int <= float (:122-123); pixels outside the image skipped (:124).  RGB value: vec3f -> vec3uc is a
 * C cast per channel (ARAP/external/mLib/include/core-math/vec3.h:32-37).  Mask value: the
 * expression `int(val > (void*)0) * 255` (:134) compares the vector's data pointer with null, which
 * is always true, so every covered pixel becomes 255. */
static void rasterize_triangle(unsigned char *img, int W, int H, int channels, const float p0[2],
                               const float p1[2], const float p2[2], const float *c0, const float *c1,
                               const float *c2, int ismask)
{
    float minx = floorf(fminf(p0[0], fminf(p1[0], p2[0])));
    float miny = floorf(fminf(p0[1], fminf(p1[1], p2[1])));
    float maxx = ceilf(fmaxf(p0[0], fmaxf(p1[0], p2[0])));
    float maxy = ceilf(fmaxf(p0[1], fmaxf(p1[1], p2[1])));
    if (!(minx == minx && miny == miny && maxx == maxx && maxy == maxy)) return; /* NaN: UB in ref */
    /* clamp to the image: pixels outside are skipped by the reference's own test at :124 */
    int xa = minx < 0.f ? 0 : (minx > (float)W ? W : (int)minx);
    int ya = miny < 0.f ? 0 : (miny > (float)H ? H : (int)miny);
    for (int x = xa; x < W && (float)x <= maxx; ++x)
        for (int y = ya; y < H && (float)y <= maxy; ++y) {
            float b0, b1, b2;
            if (point_in_triangle_lk(p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], (float)x, (float)y, &b0,
                                     &b1, &b2)) {
                unsigned char *dst = img + (size_t)channels * (x + (size_t)W * y);
                if (ismask) {
                    dst[0] = 255;
                } else {
                    for (int k = 0; k < 3; ++k) {
                        float v = c0[k] * b0 + c1[k] * b1 + c2[k] * b2;
                        dst[k] = (unsigned char)v;
                    }
                }
            }
        }
}

/* Warp, main.cpp:145-225.  rgb u8[H][W][3], mask_red u8[H][W] (0 = object), flow f32[H][W][2];
 * out_rgb u8[H][W][3], out_mask u8[H][W] (255 = object).  Quads in row-major order, triangles
 * (00,01,10) then (10,01,11), later writes overwrite earlier ones. */
void oracle_warp(int W, int H, const unsigned char *rgb, const unsigned char *mask_red, const float *flow,
                 unsigned char *out_rgb, unsigned char *out_mask)
{
    size_t N = (size_t)W * H;
    float *wf = (float *)malloc(sizeof(float) * 2 * N);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)(x + W * y);
            wf[2 * i] = (float)x + flow[2 * i];           /* :159-166 */
            wf[2 * i + 1] = (float)y + flow[2 * i + 1];
        }
    memset(out_rgb, 0, 3 * N);
    memset(out_mask, 0, N);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (!(y + 1 < H && x + 1 < W)) continue;
            size_t i00 = (size_t)(x + W * y), i01 = i00 + 1, i10 = i00 + W, i11 = i10 + 1;
            if (mask_red[i00] != 0) continue;
            if (!(mask_red[i00] == 0 && mask_red[i10] == 0 && mask_red[i01] == 0 && mask_red[i11] == 0))
                continue;
            float v00[3], v01[3], v10[3], v11[3];
            for (int k = 0; k < 3; ++k) {
                v00[k] = (float)rgb[3 * i00 + k];
                v01[k] = (float)rgb[3 * i01 + k];
                v10[k] = (float)rgb[3 * i10 + k];
                v11[k] = (float)rgb[3 * i11 + k];
            }
            rasterize_triangle(out_rgb, W, H, 3, wf + 2 * i00, wf + 2 * i01, wf + 2 * i10, v00, v01, v10, 0);
            rasterize_triangle(out_rgb, W, H, 3, wf + 2 * i10, wf + 2 * i01, wf + 2 * i11, v10, v01, v11, 0);
            rasterize_triangle(out_mask, W, H, 1, wf + 2 * i00, wf + 2 * i01, wf + 2 * i10, 0, 0, 0, 1);
            rasterize_triangle(out_mask, W, H, 1, wf + 2 * i10, wf + 2 * i01, wf + 2 * i11, 0, 0, 0, 1);
        }
    free(wf);
}

/* warpField(): ARAP/deformation/src/CombinedSolver.h:352-366 : flow = Offset - (x, y) */
void oracle_flow_from_offset(int W, int H, const float *O, float *flow)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)(x + W * y);
            flow[2 * i] = O[2 * i] - (float)x;
            flow[2 * i + 1] = O[2 * i + 1] - (float)y;
        }
}

/* the arap_deform variant of the rasteriser works on the Offset image directly
 * (CombinedSolver.h:280-342): same loop, warp field = Offset, no flow round trip. */
void oracle_warp_offset(int W, int H, const unsigned char *rgb, const unsigned char *mask_red,
                        const float *O, unsigned char *out_rgb, unsigned char *out_mask)
{
    size_t N = (size_t)W * H;
    /* a flow f with (x + f) == O exactly does not exist in general, so rasterise O directly */
    memset(out_rgb, 0, 3 * N);
    memset(out_mask, 0, N);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (!(y + 1 < H && x + 1 < W)) continue;
            size_t i00 = (size_t)(x + W * y), i01 = i00 + 1, i10 = i00 + W, i11 = i10 + 1;
            if (!(mask_red[i00] == 0 && mask_red[i10] == 0 && mask_red[i01] == 0 && mask_red[i11] == 0))
                continue;
            float v00[3], v01[3], v10[3], v11[3];
            for (int k = 0; k < 3; ++k) {
                v00[k] = (float)rgb[3 * i00 + k];
                v01[k] = (float)rgb[3 * i01 + k];
                v10[k] = (float)rgb[3 * i10 + k];
                v11[k] = (float)rgb[3 * i11 + k];
            }
            rasterize_triangle(out_rgb, W, H, 3, O + 2 * i00, O + 2 * i01, O + 2 * i10, v00, v01, v10, 0);
            rasterize_triangle(out_rgb, W, H, 3, O + 2 * i10, O + 2 * i01, O + 2 * i11, v10, v01, v11, 0);
            rasterize_triangle(out_mask, W, H, 1, O + 2 * i00, O + 2 * i01, O + 2 * i10, 0, 0, 0, 1);
            rasterize_triangle(out_mask, W, H, 1, O + 2 * i10, O + 2 * i01, O + 2 * i11, 0, 0, 0, 1);
        }
}
