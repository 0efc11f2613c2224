// arap_deform -- C++ host program over libarapopt.so with the argv contract, schedule and outputs of the
// reference's executable (ARAP/deformation/src/main.cpp:162-241 + CombinedSolver.h + CombinedSolverBase.h):
//   ./arap_deform RGB Mask Constraint Flow warped_RGB warped_Mask        (one frame)
//   ./arap_deform listfile                                               (six paths per line)
// The reference keeps one CombinedSolver (one Opt plan) and feeds it frame after frame (main.cpp:223-238);
// here consecutive frames of equal size are handed to the device-resident batched solver, as many as fit one launch
// (ArapFlow_Solver = CombinedSolver on the GPU: reset, 19-step constraint ramp, 8 GN x 400 PCG, flow, rasteriser).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <future>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

extern "C" {
#include "../../include/arap_opt.h"
}
#include "flo_io.h"
#include "png_io.h"

// one solve = one list-file line (ARAP/deformation/src/main.cpp:4-11,183-191)
struct SolvePaths {
    std::string rgb, mask, constraints, flow, warped_rgb, warped_mask;
};

// the usage text of the reference's executable (main.cpp:13-24), verbatim: it is part of the CLI contract
static const char kUsage[] =
    "Usage:\n\n"
    "./arap_deform RGB Mask Constraint Flow warped_RGB warped_Mask\n\n"
    "Mask and warp image using the provided optical flow field.\n\n"
    "RGB \t\t [input]  path to an input RGB image (.png only)\n"
    "Mask\t\t [input]  path to an input mask image (.png only) where 0 for object, 1 for background\n"
    "Constraint \t [input]  path to list of constraints, text file\n"
    "Flow \t\t [output] path to optical flow image with (.flo only)\n"
    "warped_RGB \t [output] path to output warped image (.png), all intermediate directories must exist\n"
    "warped_Mask \t [output] path to output warped mask (.png), all intermediate directories must exist\n";

// constraint file (main.cpp:26-50): a count n, then n rows of four integers x1 y1 x2 y2
static bool read_constraint_file(const std::string& path, std::vector<int32_t>& rows)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) {
        std::cout << "Could not open marker file " << path << std::endl;
        return false;
    }
    unsigned n = 0;
    rows.clear();
    if (fscanf(f, "%u", &n) == 1) {
        rows.reserve(4 * (size_t)n);
        for (size_t k = 0; k < 4 * (size_t)n; ++k) {
            int v = 0;
            if (fscanf(f, "%d", &v) != 1) v = 0;       // the reference's stream extraction leaves 0 on a short file
            rows.push_back(v);
        }
    }
    fclose(f);
    return true;
}

struct Frame {
    SolvePaths paths;
    arapio::Image rgb;
    std::vector<uint8_t> mask_red;
    std::vector<int32_t> constraints;      // x1 y1 x2 y2 rows, file order then border pins
};

// loadData of the reference (main.cpp:116-138): constraints file, PNGs, then a pin-to-self constraint for every border pixel
static bool load_frame(const SolvePaths& paths, Frame& f)
{
    f.paths = paths;
    if (!read_constraint_file(paths.constraints, f.constraints)) return false;
    std::string err;
    if (!arapio::read_png_rgb(paths.rgb, f.rgb, err)) { printf("%s\n", err.c_str()); return false; }
    arapio::Image msk;
    if (!arapio::read_png_rgb(paths.mask, msk, err)) { printf("%s\n", err.c_str()); return false; }
    if (msk.w != f.rgb.w || msk.h != f.rgb.h) {
        printf("Mask %s and image %s differ in size\n", paths.mask.c_str(), paths.rgb.c_str());
        return false;
    }
    const int width = f.rgb.w, height = f.rgb.h;
    f.mask_red.resize((size_t)width * height);
    for (size_t i = 0; i < f.mask_red.size(); ++i) f.mask_red[i] = msk.rgb[3 * i];      // red channel
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            if (y == 0 || x == 0 || y == (height - 1) || x == (width - 1)) {
                f.constraints.push_back(x); f.constraints.push_back(y);
                f.constraints.push_back(x); f.constraints.push_back(y);
            }
    return true;
}

int main(int argc, const char* argv[])
{
    std::vector<SolvePaths> lines;
    if (argc == 7) {                                                     // one frame on the command line
        lines.push_back(SolvePaths{argv[1], argv[2], argv[3], argv[4], argv[5], argv[6]});
    } else if (argc == 2) {                                              // list file
        std::ifstream list(argv[1]);
        for (std::string line; std::getline(list, line);) {
            std::istringstream tok(line);
            SolvePaths q;
            if (tok >> q.rgb >> q.mask >> q.constraints >> q.flow >> q.warped_rgb >> q.warped_mask) lines.push_back(q);
        }
    } else {
        printf("Invalid Input!\n");
        fputs(kUsage, stdout);
        return 1;
    }
    if (lines.empty()) {
        printf("No file to be processed");
        return 1;
    }
    Opt_InitializationParameters ip = {0, 0, 0, 0};
    Opt_State* state = Opt_NewState(ip);
    if (!state) return 1;
    const char* planPath = getenv("ARAP_PLAN");                          // main.cpp:206-213
    if (planPath) {
        printf("Optimization plan at %s\n", planPath);
        std::ifstream f(planPath);
        if (!f.good()) {
            printf(" Not found! Please run export ARAP_PLAN=/path/to/plan.t or copy the file to the running folder "
                   "with name arap_plan.t");
            return 1;
        }
        Opt_Problem* pr = Opt_ProblemDefine(state, planPath, "gaussNewtonGPU");
        if (!pr) return 1;
        Opt_ProblemDelete(state, pr);
    }
    const unsigned numIter = 19, nonLinearIter = 8, linearIter = 400;    // main.cpp:215-221
    // Frames per solve call: the library gives every solve a group of the resident launch's workgroups sized by its
    // active tiles and a launch costs the same however full it is, so frames join a batch while they still fit ONE
    // launch (ArapFlow_SolverLaunchesFor); minFill frames per call when the resident kernel does not apply.
    const unsigned maxBatch = 32, minFill = 8;

    // Host work overlaps the GPU: list lines are decoded ahead by worker threads (loadData: PNGs, constraint file,
    // border pins) and a finished batch is encoded and written by another thread while the next batch is being solved.
    struct Loaded { bool ok = false; Frame f; };
    const size_t kAhead = 2 * maxBatch;
    std::vector<std::future<Loaded>> loading(lines.size());
    std::vector<std::unique_ptr<Loaded>> loaded(lines.size());
    size_t started = 0;
    auto frame_at = [&](size_t k) -> Loaded& {
        for (; started < lines.size() && started <= k + kAhead; ++started) {
            const SolvePaths* q = &lines[started];
            loading[started] = std::async(std::launch::async, [q]() { Loaded l; l.ok = load_frame(*q, l.f); return l; });
        }
        if (!loaded[k]) loaded[k].reset(new Loaded(loading[k].get()));
        return *loaded[k];
    };
    struct Result { SolvePaths paths; std::vector<float> flow; std::vector<uint8_t> wrgb, wmsk; };
    std::future<void> writer;                                            // at most one batch being written

    ArapFlow_Solver* solver = nullptr;
    int sw = 0, sh = 0;
    size_t i = 0;
    while (i < lines.size()) {
        if (!frame_at(i).ok) return 1;
        const int w = frame_at(i).f.rgb.w, h = frame_at(i).f.rgb.h;
        if (w != sw || h != sh) {
            if (solver) {
                printf("Warning: Input image has different size to one in the prebuilt plan.\n"
                       "To avoid re-building the plan and to save time, put images of the same size in the same list.\n"
                       "Starting to re-build plan...\n");                // CombinedSolver.h:151-153
                ArapFlow_SolverFree(solver);
            }
            solver = ArapFlow_SolverCreate(state, (unsigned)w, (unsigned)h, maxBatch);
            if (!solver) return 1;
            sw = w; sh = h;
        }
        auto add_image = [&](unsigned b, const Frame& f) {               // addImage
            return ArapFlow_SolverSetFrame(solver, b, f.rgb.rgb.data(), f.mask_red.data(), f.constraints.data(),
                                           (unsigned)(f.constraints.size() / 4), 0) == 0;
        };
        std::vector<SolvePaths> batch;
        size_t j = i;
        while (j < lines.size() && batch.size() < maxBatch) {
            Loaded& l = frame_at(j);
            if (!l.ok) return 1;
            if (l.f.rgb.w != w || l.f.rgb.h != h) break;                 // next batch starts here
            const unsigned b = (unsigned)batch.size();
            if (!add_image(b, l.f)) return 1;
            if (b > 0) {
                const int launches = ArapFlow_SolverLaunchesFor(solver, b + 1);
                if (launches > 1 || (launches == 0 && b >= minFill)) break;   // this frame opens the next batch
            }
            batch.push_back(l.f.paths);
            loaded[j].reset();                                           // the device holds it now
            ++j;
        }
        ArapFlow_SolverSolve(solver, (unsigned)batch.size(), numIter, nonLinearIter, linearIter);   // solveAll
        ArapFlow_SolverWarp(solver, (unsigned)batch.size());
        auto results = std::make_shared<std::vector<Result>>(batch.size());
        for (size_t b = 0; b < batch.size(); ++b) {                      // copyResultToCPU (blocks until the GPU is done)
            Result& r = (*results)[b];
            r.paths = batch[b];
            r.flow.resize((size_t)w * h * 2); r.wrgb.resize((size_t)w * h * 3); r.wmsk.resize((size_t)w * h);
            ArapFlow_SolverGetResults(solver, (unsigned)b, r.flow.data(), r.wrgb.data(), r.wmsk.data(), nullptr, nullptr, nullptr);
        }
        if (writer.valid()) writer.get();
        writer = std::async(std::launch::async, [results, w, h]() {
            for (const Result& r : *results) {
                std::string err;
                if (!arapio::write_png_rgb(r.paths.warped_rgb, w, h, r.wrgb.data(), err)) printf("%s\n", err.c_str());
                if (!arapio::write_png_mask1(r.paths.warped_mask, w, h, r.wmsk.data(), err)) printf("%s\n", err.c_str());
                arapio::write_flo(r.paths.flow, r.flow.data(), w, h);
                printf("Saved\n");
            }
        });
        i = j;
    }
    if (writer.valid()) writer.get();
    if (solver) ArapFlow_SolverFree(solver);
    ArapFlow_FreeState(state);
    return 0;
}
