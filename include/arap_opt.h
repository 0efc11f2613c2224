/*
 * arap_opt.h -- C ABI of libarapopt.so, the MI355X-native (HIP, gfx950) replacement for the ARAP
 * hot path of lhoangan/arap_flow.
 *
 * Part 1 re-declares, with identical names, argument order and meaning, the ten entry points of the
 * reference's Opt C API (reference: ARAP/API/release/include/Opt.h:35-71, implemented by the
 * thunks of ARAP/API/src/createwrapper.t:124-220 around ARAP/API/src/o.t:2521-2558).  A program
 * written against the reference's Opt.h (ARAP/shared/OptSolver.h:43-91) links against this library
 * unchanged -- see INTEGRATION.md.
 *
 * Part 2 (ArapFlow_*) are additions that have no counterpart symbol in the reference: they move the
 * work the reference does on the host around each Opt_ProblemSolve (constraint ramp, reset, flow
 * extraction, triangle rasteriser: ARAP/deformation/src/CombinedSolver.h:199-366,
 * ARAP/warping/src/main.cpp:110-225) onto the GPU and batch independent frames.
 *
 * Plain C: pointers and sizes only.  Device pointers are HIP device pointers.
 */
#ifndef ARAP_OPT_H
#define ARAP_OPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Part 1: the reference's Opt API
 * ---------------------------------------------------------------------------------------------- */

typedef struct Opt_State Opt_State;     /* Opt.h:3 */
typedef struct Opt_Plan Opt_Plan;       /* Opt.h:4 */
typedef struct Opt_Problem Opt_Problem; /* Opt.h:5 */

/* Opt.h:10-30.  Passed BY VALUE to Opt_NewState.
 *   doublePrecision            must be 0 (the application never sets it: OptSolver.h:49); a non-zero
 *                              value makes Opt_NewState print an error and return NULL.
 *   verbosityLevel             0 silent, >=1 per-GN-step cost lines ("cost: a -> b",
 *                              solverGPUGaussNewton.t:1160).
 *   collectPerKernelTimingInfo non-zero: hipEvent brackets around every launch, table printed when
 *                              the solve ends (util.t:451-511).
 *   threadsPerBlock            ignored (the tile shape is fixed by the kernels). */
struct Opt_InitializationParameters {
    int doublePrecision;
    int verbosityLevel;
    int collectPerKernelTimingInfo;
    int threadsPerBlock;
};
typedef struct Opt_InitializationParameters Opt_InitializationParameters;

/* Opt.h:35.  New independent context on the current HIP device.  Never freed by the reference;
 * ArapFlow_FreeState below is the addition that frees it. */
Opt_State* Opt_NewState(Opt_InitializationParameters params);

/* Opt.h:40-41.  `filename` is the problem specification.  This library implements exactly one
 * energy, the one of the reference's arap_plan.t:1-23; the file is read, stripped of comments and
 * white space, and its declarations are checked against that energy.  Anything else prints a
 * diagnostic and returns NULL (the reference returns NULL from Opt_ProblemPlan when compilation
 * fails, o.t:861-881).  The literal name "builtin:arap" selects the energy without a file.
 * `solverkind`: "gaussNewtonGPU" (what the application uses, CombinedSolverBase.h:75-77) or "LMGPU" (the
 * Levenberg-Marquardt branch of the same solver, solverGPUGaussNewton.t:615-680,1038-1157: trust region, CtC,
 * model cost, revert); anything else: diagnostic + NULL (asserted at o.t:122). */
Opt_Problem* Opt_ProblemDefine(Opt_State* state, const char* filename, const char* solverkind);
void Opt_ProblemDelete(Opt_State* state, Opt_Problem* problem);

/* Opt.h:46-47.  dimensions[0] = W, dimensions[1] = H (Dim("W",0), Dim("H",1), arap_plan.t:1).
 * Allocates the solver state (solverGPUGaussNewton.t:1254-1284). */
Opt_Plan* Opt_ProblemPlan(Opt_State* state, Opt_Problem* problem, unsigned int* dimensions);
void Opt_PlanFree(Opt_State* state, Opt_Plan* plan);

/* Opt.h:51.  `value` points to an int for "nIterations", "lIterations", "residual_reset_period"
 * and to a float for the nine LM parameters (solverGPUGaussNewton.t:148-163), which only "LMGPU" plans read.
 * Unknown names print a warning (:1220). */
void Opt_SetSolverParameter(Opt_State* state, Opt_Plan* plan, const char* name, void* value);

/* Opt.h:56-66.  problemparams is indexed by the plan's declared indices (arap_plan.t:2-8,
 * unpacked as in util.t:664-692):
 *   [0] float2* device  Offset      in/out   [1] float*  device  Angle  in/out
 *   [2] float2* device  UrShape     in       [3] float2* device  Constraints in
 *   [4] float*  device  Mask        in       [5] float*  HOST    w_fitSqrt
 *   [6] float*  HOST    w_regSqrt
 * Images are W*H, row major, x fastest, no padding (o.t:376-387).  The array is re-read on every
 * Init and every Step (solverGPUGaussNewton.t:960,1026).
 * Opt_ProblemSolve = Init, then Step until it returns 0 (o.t:2548-2551).
 * Opt_ProblemStep: one Gauss-Newton iteration = PCGInit1, lIterations PCG iterations, update,
 * cost (:1016-1177); returns 0 once nIterations steps have been taken. */
void Opt_ProblemSolve(Opt_State* state, Opt_Plan* plan, void** problemparams);
void Opt_ProblemInit(Opt_State* state, Opt_Plan* plan, void** problemparams);
int Opt_ProblemStep(Opt_State* state, Opt_Plan* plan, void** problemparams);

/* Opt.h:71.  Cost after the last completed Init/Step, float upconverted to double
 * (solverGPUGaussNewton.t:1179-1182).  Synchronises the plan's stream. */
double Opt_ProblemCurrentCost(Opt_State* state, Opt_Plan* plan);

/* ------------------------------------------------------------------------------------------------
 * Part 2: additions (no reference symbol)
 * ---------------------------------------------------------------------------------------------- */

/* Library identification: "arapopt <version> gfx950". */
const char* ArapFlow_Version(void);

void ArapFlow_FreeState(Opt_State* state);

/* All work of `state` is enqueued on this HIP stream (hipStream_t as void*; NULL = the null
 * stream, which is what the reference uses: util.t:828). */
void ArapFlow_SetStream(Opt_State* state, void* hip_stream);
/* The same with a non-blocking stream the library creates and owns: what a host program that overlaps uploads and
 * downloads with solves wants (the null stream synchronises with every blocking stream of the process). */
int ArapFlow_UseOwnStream(Opt_State* state);

/* hipEvent stopwatch on the state's stream, for callers that have no HIP binding of their own
 * (bench.py).  Begin records an event; End records a second one, synchronises on it and returns
 * the elapsed milliseconds. */
void ArapFlow_TimerBegin(Opt_State* state);
float ArapFlow_TimerEnd(Opt_State* state);

/* Per-kernel hipEvent timing (the reference's collectPerKernelTimingInfo, Opt.h:23-25, util.t:414-511)
 * switched at run time.  While on, every kernel launch is bracketed by two events on the stream and
 * hipGraph replay is disabled.  ArapFlow_KernelTime sums the records of the kernel named
 * `kernel_name` ("GNPrep", "PCGInit1", "PCGStepA", "PCGStepB", "PCGLinearUpdate", "computeCost")
 * since timing was switched on; returns -1 if there is none. */
void ArapFlow_SetKernelTiming(Opt_State* state, int on);
int ArapFlow_KernelTime(Opt_State* state, const char* kernel_name, double* total_ms, uint64_t* launches);

/* Kernel-level entry points on raw device images, used by the parity tests (tier T1).  All pointers
 * are device pointers; vectors are split like the unknowns: an Offset-shaped float2 image and an
 * Angle-shaped float image.  wf/wr = w_fitSqrt/w_regSqrt.  Synchronous.
 *   ArapFlow_EvalJTF : gradient J^T F and diag(J^T J) of o.t:2129-2172
 *   ArapFlow_ApplyJTJ: (J^T J P) of o.t:2029-2089
 *   ArapFlow_Cost    : o.t:2375-2385
 * Return 0 on success, a HIP error code otherwise. */
int ArapFlow_EvalJTF(Opt_State* state, unsigned W, unsigned H, const void* Offset, const void* Angle,
                     const void* UrShape, const void* Constraints, const void* Mask, float wf, float wr,
                     void* gO, void* gA, void* dO, void* dA);
int ArapFlow_ApplyJTJ(Opt_State* state, unsigned W, unsigned H, const void* Angle, const void* UrShape,
                      const void* Constraints, const void* Mask, float wf, float wr, const void* pO,
                      const void* pA, void* outO, void* outA);
int ArapFlow_Cost(Opt_State* state, unsigned W, unsigned H, const void* Offset, const void* Angle,
                  const void* UrShape, const void* Constraints, const void* Mask, float wf, float wr,
                  double* cost_host);

/* Batched frame solver: the device-resident counterpart of the reference's CombinedSolver
 * (ARAP/deformation/src/CombinedSolver.h:99-390) driven as in ARAP/deformation/src/main.cpp:140-160.
 * One object serves frames of one size; `batch` frames are solved concurrently, one set of PCG
 * scalars per frame.  Frames are independent: no exchange between slots. */
typedef struct ArapFlow_Solver ArapFlow_Solver;

ArapFlow_Solver* ArapFlow_SolverCreate(Opt_State* state, unsigned W, unsigned H, unsigned batch);
void ArapFlow_SolverFree(ArapFlow_Solver* s);

/* addImage (CombinedSolver.h:139-170) for slot `slot`.  HOST pointers:
 *   rgb       uint8[H][W][3] or NULL (no warp wanted)
 *   mask_red  uint8[H][W]    red channel of the mask PNG: 0 = deformable object (CombinedSolver.h:213)
 *   cons      int32[ncons][4] = x1 y1 x2 y2 rows of the constraint file (main.cpp:26-50), file order
 *   add_border_pins  non-zero: append (x,y,x,y) for every border pixel (main.cpp:130-136)
 * The host buffers are staged into pinned memory before the call returns (they may be reused at once); the copies
 * to the device run on the solver's own copy stream and the next solve waits for them.  If a solve of this solver is
 * still in flight the call waits for it first.  Returns 0, or -1 on bad arguments. */
int ArapFlow_SolverSetFrame(ArapFlow_Solver* s, unsigned slot, const uint8_t* rgb, const uint8_t* mask_red,
                            const int32_t* cons, unsigned ncons, int add_border_pins);

/* solveAll (CombinedSolverBase.h:23-31,99-120) for slots [0, nframes): reset (CombinedSolver.h:207-221),
 * then for i < numIter: constraints ramped to alpha = (i+1)/numIter (:199-201,223-242) and one
 * Opt_ProblemSolve with nIterations/lIterations, unknowns carried over.  The application's values
 * are 19, 8, 400 (main.cpp:215-221).  Returns when the solve has finished. */
int ArapFlow_SolverSolve(ArapFlow_Solver* s, unsigned nframes, unsigned numIter, unsigned nIterations,
                         unsigned lIterations);

/* Pipelined form of the three calls around a solve, for hosts that keep the GPU busy (arap_deform --serve): enqueue
 * the whole schedule of slots [0, nframes) on the state's stream -- behind this solver's pending SetFrame uploads,
 * which travel on the solver's own copy stream from pinned staging -- then, if `warp`, the flow emission + rasteriser,
 * then, if `download`, the copy of every slot's flow / warped RGB / warped mask into pinned host buffers owned by the
 * solver (again on the copy stream), and return WITHOUT waiting.  With two solver objects a host uploads batch k+1
 * and reads back batch k-1 while batch k is being solved.  ArapFlow_SolverWait blocks until everything enqueued for
 * this solver is done (and, if a resident launch gave up, redoes the schedule on the two-kernel path first); every
 * other call on a solver with work in flight waits by itself.  ArapFlow_SolverSolve = SolveAsync(.., 0, 0) + Wait.
 * ArapFlow_SolverHostResults returns pointers into the pinned buffers of a `download` solve (valid until the next
 * solve of this solver; warped_rgb NULL if the slot has no RGB).  Return 0, -1 on bad arguments. */
int ArapFlow_SolverSolveAsync(ArapFlow_Solver* s, unsigned nframes, unsigned numIter, unsigned nIterations,
                              unsigned lIterations, int warp, int download);
int ArapFlow_SolverWait(ArapFlow_Solver* s);
int ArapFlow_SolverHostResults(ArapFlow_Solver* s, unsigned slot, const float** flow, const uint8_t** warped_rgb,
                               const uint8_t** warped_mask);

/* copyResultToCPU + warpField (CombinedSolver.h:280-366) on the device for slots [0, nframes):
 * flow = Offset - grid, and the forward triangle rasterisation of rgb and mask with the solved
 * Offset as warp field.  Asynchronous. */
int ArapFlow_SolverWarp(ArapFlow_Solver* s, unsigned nframes);

/* Synchronise and copy one slot's results to HOST buffers (any may be NULL):
 *   flow float[H][W][2], warped_rgb uint8[H][W][3], warped_mask uint8[H][W] (255 = object),
 *   offset float[H][W][2], angle float[H][W], final_cost = Opt_ProblemCurrentCost of the last solve. */
int ArapFlow_SolverGetResults(ArapFlow_Solver* s, unsigned slot, float* flow, uint8_t* warped_rgb,
                              uint8_t* warped_mask, float* offset, float* angle, double* final_cost);

/* Counters of the last ArapFlow_SolverSolve (for the bench): number of PCG iterations executed per
 * frame and number of active (Mask == 0) vertices summed over the solved slots. */
int ArapFlow_SolverStats(ArapFlow_Solver* s, uint64_t* pcg_iterations_per_frame, uint64_t* active_vertices,
                         uint64_t* grid_vertices);

/* The frame solver runs the PCG loop of a Gauss-Newton step either as two kernels per iteration
 * replayed from a hipGraph, or -- when every frame's active 64x4 tiles fit on chip -- as ONE resident
 * launch that keeps the PCG state in registers/LDS.  Both perform the same float32
 * operations; results are identical.  ArapFlow_SetResident(state, 0) forces the two-kernel path;
 * ArapFlow_SolverResidentLaunches counts resident launches since the solver was created. */
void ArapFlow_SetResident(Opt_State* state, int on);
/* Phase-A kernel of the two-kernel path: (0,0) = neighbours read through L1/L2, or an LDS-staged tile of (16,16),
 * (32,8), (64,4), (32,16) or (64,8) vertices -- BASELINE config 5's tile sweep; (-1,-1) = automatic (default: 64x8
 * when most tiles are active, direct otherwise).  Same arithmetic, identical results.  -1 for any other shape. */
int ArapFlow_SetTile(Opt_State* state, int tile_x, int tile_y);
uint64_t ArapFlow_SolverResidentLaunches(ArapFlow_Solver* s);
/* How the last solve's frames were dealt to resident launches: launches per Gauss-Newton step and the largest
 * number of solves in flight in one launch (every solve gets a group of the launch's 512 workgroups sized by its
 * active-tile count; small solves share an XCD).  Both 0 when the two-kernel path ran.  Returns 0, -1 on NULL. */
int ArapFlow_SolverResidentLayout(ArapFlow_Solver* s, int* launches_per_step, int* solves_in_flight);
/* Resident launches per Gauss-Newton step that a solve of slots [0, nframes) would take with the frames set so far
 * (0: the two-kernel path would run, -1: bad arguments).  Every launch costs about the same time however full it
 * is, so a host that wants the best throughput adds frames to a batch while this stays 1 (arap_deform does). */
int ArapFlow_SolverLaunchesFor(ArapFlow_Solver* s, unsigned nframes);
/* The deal itself, as a pure host function (no device needed): given the active 64x4-tile counts of `nsolves` solves
 * (each <= 4608) it returns the number of resident launches and, when `table` is given, writes for each of the first
 * `table_launches` launches the 512 workgroup entries {solve or -1, rank in its group, group size, granule offset}
 * (int[table_launches][512][4]).  -1 on bad arguments. */
int ArapFlow_ResidentDeal(const int* active_tiles, unsigned nsolves, int* table, unsigned table_launches);
/* The resident kernel's work list of one solve, as a pure host function (no device needed): 32x8 tiles in bands of 8
 * rows; inside a band the tiles start at the band's first active vertex (`aligned` != 0; else at x = 0) and follow each
 * other every 32 columns; tiles without an active vertex (mask_red == 0) are left out.  Writes up to `cap` tile origins
 * (x0 + W * y0, band by band) and the ceil(H / 8) band start columns; returns the number of tiles, -1 on bad arguments. */
int ArapFlow_ResidentTiles(const uint8_t* mask_red, unsigned W, unsigned H, int aligned, int* origins, unsigned cap,
                           int* bandx0);
/* The drop-in path (Opt_ProblemInit/Step/Solve) takes the resident kernel too when the caller's UrShape is the
 * pixel grid on every active vertex (what the application passes, CombinedSolver.h:207-221) and the active tiles
 * fit.  Mask and UrShape are looked at before EVERY step (the reference re-reads its parameters at every step and
 * lets the caller change them in between, Opt.h:58-66): new contents or swapped buffers are honoured.  Counts the
 * resident launches executed for this plan. */
uint64_t ArapFlow_PlanResidentLaunches(Opt_Plan* plan);
/* The resident kernel needs its 512 workgroups co-resident.  If a launch gives up at a bounded group wait (e.g.
 * another process is using the GPU), the step's update is skipped on the device, the work is redone on the
 * two-kernel path (same results) and the resident path pauses for the next 8 solve calls (doubling with every
 * further timeout, up to 1024; a checked success resets it).  Returns 1 once any launch of this state has given up. */
int ArapFlow_ResidentFailed(Opt_State* state);
/* 1 if the batch of the last solve call, where it runs kernel per phase (solves the resident kernel cannot hold, or a
 * pause after a timeout), takes the lean streaming schedule (arap_stream.h: k_pcg_a_march2 / k_pcg_b4_r: 85 + 41 bytes
 * per vertex and iteration instead of 57 + 89), which it does when most of its tiles are active. */
int ArapFlow_SolverLeanStream(ArapFlow_Solver* s);
/* Diagnostic only (env ARAPOPT_STAMPS=1 selects an instrumented build of the resident kernel): copies
 * out[512][16] = per workgroup {phase A, wait 1, phase B, wait 2, update} summed 100 MHz ticks of the
 * last resident launch, tiles per workgroup, halo cells.  Returns -1 when stamps are off. */
int ArapFlow_SolverStamps(ArapFlow_Solver* s, uint64_t* out);

/* warp_image (ARAP/warping/src/main.cpp:145-225) on DEVICE buffers: rgb uint8[H][W][3], mask_red
 * uint8[H][W], flow float[H][W][2] -> out_rgb uint8[H][W][3], out_mask uint8[H][W].
 * `scratch` is a device buffer of ArapFlow_WarpScratchBytes(W,H) bytes.  Asynchronous on the state's
 * stream.  Returns 0 or a HIP error code. */
uint64_t ArapFlow_WarpScratchBytes(unsigned W, unsigned H);
int ArapFlow_Warp(Opt_State* state, unsigned W, unsigned H, const void* rgb, const void* mask_red,
                  const void* flow, void* out_rgb, void* out_mask, void* scratch);

#ifdef __cplusplus
}
#endif
#endif /* ARAP_OPT_H */
