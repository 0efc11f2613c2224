// warp_image -- C++ host program with the argv contract of ARAP/warping/src/main.cpp:302-336:
//   ./warp_image image mask flow warped_image warped_mask
// The rasterisation runs on the GPU (ArapFlow_Warp); output is bit exact against the reference's executable.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <string>
#include <vector>

extern "C" {
#include "../../include/arap_opt.h"
}
#include "flo_io.h"
#include "png_io.h"

static void usage()
{
#define p(msg) printf(msg "\n");
    p("Usage:");
    p("./warp_image image mask flow warped_image warped_mask");
    p("Mask and warp image using the provided optical flow field.")
    p("\timage: path to image with png extension")
    p("\tmask: path to mask image with png extension, 0 for object, 1 for background")
    p("\tflo: path to optical flow image with flo extension")
    p("\twarped_image: path to output warped image (.png), all intermediate directories must exist")
    p("\twarped_mask: path to output warped mask (.png), all intermediate directories must exist")
#undef p
}

#define HCHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, const char* argv[])
{
    if (argc != 6) {
        printf("Invalid Input! ");
        usage();
        return 1;
    }
    std::string err;
    arapio::Image rgb, msk;
    if (!arapio::read_png_rgb(argv[1], rgb, err) || !arapio::read_png_rgb(argv[2], msk, err)) { printf("%s\n", err.c_str()); return 1; }
    std::vector<float> flow;
    int fw = 0, fh = 0;
    if (!arapio::read_flo(argv[3], flow, fw, fh)) return 1;
    if (fw != rgb.w || fh != rgb.h || msk.w != rgb.w || msk.h != rgb.h) { printf("image, mask and flow sizes differ\n"); return 1; }
    const int w = rgb.w, h = rgb.h;
    const size_t N = (size_t)w * h;
    std::vector<uint8_t> mred(N);
    for (size_t i = 0; i < N; ++i) mred[i] = msk.rgb[3 * i];
    Opt_InitializationParameters ip = {0, 0, 0, 0};
    Opt_State* state = Opt_NewState(ip);
    if (!state) return 1;
    void *d_rgb, *d_msk, *d_flow, *d_orgb, *d_omsk, *d_scr;
    HCHECK(hipMalloc(&d_rgb, 3 * N)); HCHECK(hipMalloc(&d_msk, N)); HCHECK(hipMalloc(&d_flow, 8 * N));
    HCHECK(hipMalloc(&d_orgb, 3 * N)); HCHECK(hipMalloc(&d_omsk, N));
    HCHECK(hipMalloc(&d_scr, ArapFlow_WarpScratchBytes((unsigned)w, (unsigned)h)));
    HCHECK(hipMemcpy(d_rgb, rgb.rgb.data(), 3 * N, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(d_msk, mred.data(), N, hipMemcpyHostToDevice));
    HCHECK(hipMemcpy(d_flow, flow.data(), 8 * N, hipMemcpyHostToDevice));
    if (ArapFlow_Warp(state, (unsigned)w, (unsigned)h, d_rgb, d_msk, d_flow, d_orgb, d_omsk, d_scr) != 0) { printf("ArapFlow_Warp failed\n"); return 1; }
    HCHECK(hipDeviceSynchronize());
    std::vector<uint8_t> orgb(3 * N), omsk(N);
    HCHECK(hipMemcpy(orgb.data(), d_orgb, 3 * N, hipMemcpyDeviceToHost));
    HCHECK(hipMemcpy(omsk.data(), d_omsk, N, hipMemcpyDeviceToHost));
    if (!arapio::write_png_rgb(argv[4], w, h, orgb.data(), err) || !arapio::write_png_mask1(argv[5], w, h, omsk.data(), err)) {
        printf("%s\n", err.c_str());
        return 1;
    }
    printf("Saved\n");
    return 0;
}
