// arap_deform -- C++ host program over libarapopt.so with the argv contract, schedule and outputs of the
// reference's executable (ARAP/deformation/src/main.cpp:162-241 + CombinedSolver.h + CombinedSolverBase.h):
//   ./arap_deform RGB Mask Constraint Flow warped_RGB warped_Mask        (one frame)
//   ./arap_deform listfile                                               (six paths per line)
//   ./arap_deform --serve                                                (addition: the same lines on stdin, until EOF)
// The reference keeps one CombinedSolver (one Opt plan) and feeds it frame after frame (main.cpp:223-238);
// here consecutive frames of equal size are handed to the device-resident batched solver, as many as fit one launch
// (ArapFlow_Solver = CombinedSolver on the GPU: reset, 19-step constraint ramp, 8 GN x 400 PCG, flow, rasteriser).
//
// The GPU never waits for the host: two solver objects alternate.  While one batch is being solved, the next one is
// decoded (worker threads), uploaded into the other object (its own copy stream, pinned staging) and the previous
// batch's results are read back and encoded (worker threads).  --serve is what para_gen.py starts once per GPU: a
// persistent worker that is fed list-file lines over a pipe as the front end produces them and reports
// "Done <flow path>" per finished solve, instead of one child process (HIP start-up, plan, graph capture) per hand-out.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <future>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unistd.h>
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

static bool parse_line(const std::string& line, SolvePaths& q)
{
    std::istringstream tok(line);
    return (bool)(tok >> q.rgb >> q.mask >> q.constraints >> q.flow >> q.warped_rgb >> q.warped_mask);
}

// ---- where the lines come from: a finished list, or stdin as it arrives (--serve) ------------------------------------
struct Loaded { bool ok = false; Frame f; };

class FrameSource {
  public:
    explicit FrameSource(std::vector<SolvePaths> fixed) : eof_(true)
    {
        for (auto& q : fixed) lines_.push_back(std::move(q));
    }
    FrameSource() : eof_(false)                                         // --serve: a thread reads stdin
    {
        reader_ = std::thread([this]() {
            for (std::string line; std::getline(std::cin, line);) {
                SolvePaths q;
                if (!parse_line(line, q)) continue;
                { std::lock_guard<std::mutex> g(m_); lines_.push_back(std::move(q)); }
                cv_.notify_all();
            }
            { std::lock_guard<std::mutex> g(m_); eof_ = true; }
            cv_.notify_all();
        });
    }
    ~FrameSource() { if (reader_.joinable()) reader_.join(); }

    // Next decoded frame in line order.  wait_ms < 0: block until one is there or the source is exhausted;
    // otherwise give up after wait_ms.  Returns 1 (frame in *out), 0 (nothing within the time), -1 (exhausted).
    int next(std::unique_ptr<Loaded>* out, int wait_ms)
    {
        using clock = std::chrono::steady_clock;
        const auto deadline = clock::now() + std::chrono::milliseconds(wait_ms < 0 ? 0 : wait_ms);
        for (;;) {
            start_loads();
            if (!loading_.empty()) {
                std::future<Loaded>& f = loading_.front();
                if (wait_ms < 0) f.wait();
                else if (f.wait_until(deadline) != std::future_status::ready) return 0;
                out->reset(new Loaded(f.get()));
                loading_.pop_front();
                return 1;
            }
            std::unique_lock<std::mutex> g(m_);
            if (!lines_.empty()) continue;
            if (eof_) return -1;
            if (wait_ms < 0) cv_.wait(g, [this]() { return !lines_.empty() || eof_; });
            else if (!cv_.wait_until(g, deadline, [this]() { return !lines_.empty() || eof_; })) return 0;
        }
    }

  private:
    void start_loads()                                                  // decode ahead: loadData on worker threads
    {
        std::lock_guard<std::mutex> g(m_);
        while (!lines_.empty() && loading_.size() < kAhead) {
            auto q = std::make_shared<SolvePaths>(std::move(lines_.front()));
            lines_.pop_front();
            loading_.push_back(std::async(std::launch::async, [q]() { Loaded l; l.ok = load_frame(*q, l.f); return l; }));
        }
    }
    static constexpr size_t kAhead = 48;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<SolvePaths> lines_;
    std::deque<std::future<Loaded>> loading_;
    bool eof_;
    std::thread reader_;
};

// ---- results: read back from the solver's pinned buffers, encoded and written by worker threads ---------------------
struct Result { SolvePaths paths; std::vector<float> flow; std::vector<uint8_t> wrgb, wmsk; };

class Writer {
  public:
    explicit Writer(bool report) : report_(report) {}
    void submit(std::shared_ptr<Result> r, int w, int h)
    {
        while (tasks_.size() >= 24) { tasks_.front().get(); tasks_.pop_front(); }
        const bool report = report_;
        std::mutex* pm = &print_;
        tasks_.push_back(std::async(std::launch::async, [r, w, h, report, pm]() {
            std::string err;
            if (!arapio::write_png_rgb(r->paths.warped_rgb, w, h, r->wrgb.data(), err)) printf("%s\n", err.c_str());
            if (!arapio::write_png_mask1(r->paths.warped_mask, w, h, r->wmsk.data(), err)) printf("%s\n", err.c_str());
            arapio::write_flo(r->paths.flow, r->flow.data(), w, h);
            std::lock_guard<std::mutex> g(*pm);
            if (report) printf("Done %s\n", r->paths.flow.c_str());     // --serve: one line per finished solve
            else printf("Saved\n");
            fflush(stdout);
        }));
    }
    void finish() { for (auto& t : tasks_) t.get(); tasks_.clear(); }

  private:
    bool report_;
    std::mutex print_;
    std::deque<std::future<void>> tasks_;
};

struct Lane {                              // one of the two alternating solver objects
    ArapFlow_Solver* solver = nullptr;
    std::vector<SolvePaths> batch;         // frames set into the slots, in slot order
    bool inflight = false;
};

int main(int argc, const char* argv[])
{
    std::unique_ptr<FrameSource> source;
    bool serve = false;
    if (argc == 7) {                                                     // one frame on the command line
        source.reset(new FrameSource(std::vector<SolvePaths>{SolvePaths{argv[1], argv[2], argv[3], argv[4], argv[5], argv[6]}}));
    } else if (argc == 2 && strcmp(argv[1], "--serve") == 0) {
        serve = true;
    } else if (argc == 2) {                                              // list file
        std::ifstream list(argv[1]);
        std::vector<SolvePaths> lines;
        for (std::string line; std::getline(list, line);) {
            SolvePaths q;
            if (parse_line(line, q)) lines.push_back(q);
        }
        if (lines.empty()) {
            printf("No file to be processed");
            return 1;
        }
        source.reset(new FrameSource(std::move(lines)));
    } else {
        printf("Invalid Input!\n");
        fputs(kUsage, stdout);
        return 1;
    }
    Opt_InitializationParameters ip = {0, 0, 0, 0};
    Opt_State* state = Opt_NewState(ip);
    if (!state) return 1;
    ArapFlow_UseOwnStream(state);            // uploads / downloads overlap the solves (see the header comment)
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
    if (serve) {
        printf("Ready\n");                   // HIP is up: the parent may start its clock / feed lines
        fflush(stdout);
        source.reset(new FrameSource());
    }
    const unsigned numIter = 19, nonLinearIter = 8, linearIter = 400;    // main.cpp:215-221
    // Frames per solve call: the library gives every solve a group of the resident launch's workgroups sized by its
    // active tiles and a launch costs the same however full it is, so frames join a batch while they still fit ONE
    // launch (ArapFlow_SolverLaunchesFor); minFill frames per call when the resident kernel does not apply.
    // Both bounds follow the frame size: a solver object pins 24 bytes x vertices x maxBatch of host staging (two objects:
    // 0.6 GB at 854x480 x 32 -- but 3.2 GB at 1920x1080 x 32, where a launch holds four segment solves anyway).
    auto max_batch_for = [](int w, int h) -> unsigned {
        const double rel = (double)w * h / (854.0 * 480.0);
        const int m = (int)(32.0 / (rel < 1.0 ? 1.0 : rel));
        return (unsigned)(m < 8 ? 8 : m);
    };
    unsigned maxBatch = 32;
    const unsigned minFill = 8;
    // --serve: how long a partly filled batch waits for another line when the GPU is idle
    int linger_ms = 30;
    if (const char* e = getenv("ARAP_DEFORM_LINGER_MS")) linger_ms = atoi(e);

    Writer writer(serve);
    Lane lanes[2];
    int sw = 0, sh = 0, cur = 0;
    std::unique_ptr<Loaded> carry;           // decoded frame that did not fit the batch it was offered to
    bool exhausted = false;
    int rc = 0;

    auto drain = [&](Lane& L) -> bool {      // wait for the lane's solve, hand its results to the writer threads
        if (!L.inflight) return true;
        if (ArapFlow_SolverWait(L.solver) != 0) { printf("ARAP solve failed\n"); return false; }
        const size_t n = (size_t)sw * sh;
        for (size_t b = 0; b < L.batch.size(); ++b) {                    // copyResultToCPU
            const float* flow; const uint8_t *wrgb, *wmsk;
            if (ArapFlow_SolverHostResults(L.solver, (unsigned)b, &flow, &wrgb, &wmsk) != 0 || !wrgb) {
                printf("ARAP results unavailable\n");
                return false;
            }
            auto r = std::make_shared<Result>();
            r->paths = L.batch[b];
            r->flow.assign(flow, flow + 2 * n);
            r->wrgb.assign(wrgb, wrgb + 3 * n);
            r->wmsk.assign(wmsk, wmsk + n);
            writer.submit(r, sw, sh);
        }
        L.batch.clear();
        L.inflight = false;
        return true;
    };
    auto launch = [&](Lane& L) -> bool {
        if (L.batch.empty()) return true;
        if (ArapFlow_SolverSolveAsync(L.solver, (unsigned)L.batch.size(), numIter, nonLinearIter, linearIter, 1, 1) != 0) {
            printf("ARAP solve could not be started\n");
            return false;
        }
        if (serve) { printf("Batch %zu\n", L.batch.size()); fflush(stdout); }    // (para_gen.py keeps statistics)
        L.inflight = true;
        return true;
    };

    while (rc == 0) {
        Lane& L = lanes[cur];
        Lane& other = lanes[cur ^ 1];
        // ---- next frame: the carried one, or whatever the source has.  An empty batch with nothing in flight blocks;
        //      a partly filled one waits `linger_ms` (while the other lane is solving, waiting costs nothing: drain it
        //      first, new lines may arrive meanwhile)
        std::unique_ptr<Loaded> fr;
        bool full = false;
        if (carry) fr = std::move(carry);
        else if (!exhausted) {
            int got;
            if (L.batch.empty() && !other.inflight) got = source->next(&fr, -1);
            else {
                got = source->next(&fr, 0);
                if (got == 0 && other.inflight) {
                    if (!drain(other)) { rc = 1; break; }
                    got = source->next(&fr, 0);
                }
                if (got == 0) got = source->next(&fr, L.batch.empty() ? -1 : linger_ms);
            }
            if (got < 0) exhausted = true;
        }
        if (fr) {
            if (!fr->ok) { rc = 1; break; }
            const int w = fr->f.rgb.w, h = fr->f.rgb.h;
            if (w != sw || h != sh) {
                // another frame size: finish everything of the old size, then re-build (CombinedSolver.h:149-160)
                if (!launch(L) || !drain(other) || !drain(L)) { rc = 1; break; }
                if (lanes[0].solver) {
                    printf("Warning: Input image has different size to one in the prebuilt plan.\n"
                           "To avoid re-building the plan and to save time, put images of the same size in the same list.\n"
                           "Starting to re-build plan...\n");            // CombinedSolver.h:151-153
                    ArapFlow_SolverFree(lanes[0].solver);
                    ArapFlow_SolverFree(lanes[1].solver);
                }
                maxBatch = max_batch_for(w, h);
                lanes[0].solver = ArapFlow_SolverCreate(state, (unsigned)w, (unsigned)h, maxBatch);
                lanes[1].solver = ArapFlow_SolverCreate(state, (unsigned)w, (unsigned)h, maxBatch);
                if (!lanes[0].solver || !lanes[1].solver) { rc = 1; break; }
                sw = w; sh = h;
            }
            const unsigned b = (unsigned)L.batch.size();
            if (ArapFlow_SolverSetFrame(L.solver, b, fr->f.rgb.rgb.data(), fr->f.mask_red.data(), fr->f.constraints.data(),
                                        (unsigned)(fr->f.constraints.size() / 4), 0) != 0) { rc = 1; break; }   // addImage
            bool fits = true;
            if (b > 0) {
                const int launches = ArapFlow_SolverLaunchesFor(L.solver, b + 1);
                fits = !(launches > 1 || (launches == 0 && b >= minFill));
            }
            if (fits) {
                L.batch.push_back(fr->f.paths);
                full = L.batch.size() >= maxBatch;
            } else {
                carry = std::move(fr);                                   // opens the next batch (set again there)
                full = true;
            }
            if (!full) continue;
        }
        // ---- nothing more joins this batch: start it, then turn to the other lane (its results, then its next batch)
        if (!L.batch.empty()) {
            if (!launch(L)) { rc = 1; break; }
            if (!drain(other)) { rc = 1; break; }
            cur ^= 1;
        } else if (exhausted && !carry) {
            break;
        }
    }
    if (rc == 0 && (!drain(lanes[cur ^ 1]) || !drain(lanes[cur]))) rc = 1;
    writer.finish();
    for (Lane& L : lanes)
        if (L.solver) ArapFlow_SolverFree(L.solver);
    ArapFlow_FreeState(state);
    fflush(stdout);
    if (rc != 0) _exit(rc);                  // (--serve: the stdin reader may still be blocked in getline)
    return rc;
}
