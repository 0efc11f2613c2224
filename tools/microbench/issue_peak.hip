// issue_peak.hip -- what the vector-ALU and LDS pipes of one gfx950 SIMD / CU sustain, in the occupancy the resident
// PCG kernel runs at (two 256-thread workgroups per CU = two wavefronts per SIMD), so that the utilisation figures of
// bench.py's roofline are quoted against MEASURED ceilings ("calibrate the binding-ceiling fields").
//
//   hipcc --offload-arch=gfx950 -O3 -o issue_peak issue_peak.hip && ./issue_peak            (JSON lines on stdout)
//   rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- ./issue_peak      (counter ceilings)
//
// Kernels (512 workgroups x 256 threads, 78.9 KB of dynamic LDS each => 2 per CU; or 256 x 256 with 160 KB => 1 per CU):
//   fma     16 independent v_fma_f32 chains per lane
//   pkfma   16 independent v_pk_fma_f32 chains per lane
//   mix     the resident kernel's phase-A recipe per "vertex": 13 v_pk_fma_f32, 8 v_pk_add_f32, 15 v_fma_f32, 12 v_and_b32
//   lds64   ds_read_b64, 16 in flight per lane, conflict free (lane-consecutive 8-byte cells)
//   lds32   ds_read_b32, likewise
//   mixlds  mix + the 15 LDS reads of a phase-A vertex (10 x ds_read_b64, 5 x ds_read_b32) per 48 VALU: do the pipes overlap?
// Every kernel stamps s_memtime / s_memrealtime around its loop in wave 0 of every workgroup: cycles and the clock held.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

struct Stamp { unsigned long long cyc, rt; };

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_issue(float* out, Stamp* stamps, int iters, float a, float b)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    // touch the LDS so that the allocation is real; 16 KB of finite values for the read kernels
    for (int i = tid; i < 4096; i += 256) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    float acc[16];
    v2f acc2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = (float)(tid + i); acc2[i] = (v2f){(float)(tid + i), (float)(tid - i)}; }
    const v2f a2 = (v2f){a, a}, b2 = (v2f){b, b};
    const unsigned m = 0x7fffffffu;
    const unsigned base64 = (unsigned)tid * 8u, base32 = (unsigned)tid * 4u;
    unsigned long long c0 = 0, r0 = 0;
    if ((tid & 63) == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
        } else if (MODE == 2 || MODE == 5) {
            // 48 VALU instructions in the proportions of phase A (13 pk_fma, 8 pk_add, 15 fma, 12 and)
            v2f l64[10];
            float l32[5];
            if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 10; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(l64[i]) : "v"(base64), "n"(i * 2048));
#pragma unroll
                for (int i = 0; i < 5; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(l32[i]) : "v"(base32), "n"(i * 1024));
            }
#pragma unroll
            for (int i = 0; i < 13; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc2[(i + 5) & 15]) : "v"(b2));
#pragma unroll
            for (int i = 0; i < 15; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(acc[(i + 3) & 15]) : "v"(m));
            if (MODE == 5) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 10; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc2[i]) : "v"(l64[i]));
#pragma unroll
                for (int i = 0; i < 5; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(l32[i]));
            }
        } else if (MODE == 3) {
            v2f l[16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(l[i]) : "v"(base64), "n"((i & 7) * 2048));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(l[i]));
            }
            acc2[0] += l[0];
        } else if (MODE == 4) {
            float l[16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(l[i]) : "v"(base32), "n"((i & 15) * 1024));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(l[i]));
            }
            acc[0] += l[0];
        }
    }
    if ((tid & 63) == 0) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) { stamps[blockIdx.x].cyc = c1 - c0; stamps[blockIdx.x].rt = r1 - r0; }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i].x + acc2[i].y;
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;        // never true: keeps the chains alive
}

typedef void (*Kern)(float*, Stamp*, int, float, float);
struct Mode { const char* name; Kern k; int valu_per_iter, pk_per_iter, lds64_per_iter, lds32_per_iter; };

int main(int argc, char** argv)
{
    const Mode modes[] = {
        {"fma", k_issue<0>, 64, 0, 0, 0}, {"pkfma", k_issue<1>, 64, 64, 0, 0}, {"mix", k_issue<2>, 48, 21, 0, 0},
        {"lds64", k_issue<3>, 0, 0, 64, 0}, {"lds32", k_issue<4>, 0, 0, 0, 64}, {"mixlds", k_issue<5>, 63, 31, 10, 5},
    };
    int iters = 20000;
    const char* only = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else only = argv[i];
    }
    float* out; Stamp* st;
    HC(hipMalloc(&out, 512 * 256 * sizeof(float)));
    HC(hipMalloc(&st, 512 * sizeof(Stamp)));
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    for (const Mode& m : modes) {
        if (only && strcmp(only, m.name)) continue;
        for (int per_cu = 2; per_cu >= 1; --per_cu) {
            const int lds_bytes = per_cu == 2 ? 80768 : 163840 - 1024, wgs = 256 * per_cu;
            HC(hipFuncSetAttribute((const void*)m.k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            hipLaunchKernelGGL(m.k, dim3(wgs), dim3(256), lds_bytes, 0, out, st, 200, 1.0001f, 0.5f);     // warm-up
            HC(hipDeviceSynchronize());
            HC(hipEventRecord(e0));
            hipLaunchKernelGGL(m.k, dim3(wgs), dim3(256), lds_bytes, 0, out, st, iters, 1.0001f, 0.5f);
            HC(hipEventRecord(e1));
            HC(hipEventSynchronize(e1));
            float ms = 0.f;
            HC(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Stamp> h(wgs);
            HC(hipMemcpy(h.data(), st, wgs * sizeof(Stamp), hipMemcpyDeviceToHost));
            std::vector<double> cyc, clk;
            for (auto& s : h) { cyc.push_back((double)s.cyc); clk.push_back((double)s.cyc / (double)s.rt * 100.0); }   // s_memrealtime: 100 MHz
            std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
            const double c = cyc[cyc.size() / 2], mhz = clk[clk.size() / 2];
            // per SIMD: per_cu wavefronts, each `iters` x the per-iteration instruction counts
            const double valu = (double)m.valu_per_iter * iters * per_cu, l64 = (double)m.lds64_per_iter * iters * per_cu,
                         l32 = (double)m.lds32_per_iter * iters * per_cu;
            printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"loop_cycles_median\": %.0f, \"clock_MHz\": %.0f, "
                   "\"valu_insts_per_simd\": %.0f, \"cycles_per_valu_inst_per_simd\": %s, \"pk_share\": %.2f, "
                   "\"lds_b64_per_cu\": %.0f, \"lds_b32_per_cu\": %.0f, \"lds_bytes_per_clk_per_cu\": %.1f}\n",
                   m.name, per_cu, ms, c, mhz, valu, valu > 0 ? (std::to_string(c / valu)).c_str() : "null",
                   m.valu_per_iter ? (double)m.pk_per_iter / m.valu_per_iter : 0.0, 4 * l64, 4 * l32,
                   (4 * l64 * 512.0 + 4 * l32 * 256.0) / c);
            fflush(stdout);
        }
    }
    return 0;
}
