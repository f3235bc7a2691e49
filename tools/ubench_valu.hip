// ubench_valu.hip -- f32 VALU issue rates on gfx950 (round 4): cycles per wave64 instruction per SIMD for
//   dep:   v_fmac_f32 chains where every instruction depends on the one before (the LK rows' shape: one tmp register)
//   ind:   eight independent accumulators
//   pk:    v_pk_fma_f32, eight independent accumulator pairs (two FMAs per lane and instruction)
//   fmix:  v_fma_mix_f32 (f16 / f32 sources mixed, f32 result), alone and as 5 of the 7 instructions of the LK tap
//   mix:   the LK tap (sub, fmac, sub, fmac, sub, fmac, fmac through one tmp), two taps interleaved on two tmps
// at 1..8 waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 2000;
#define R8(X) X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b = seed * 0.5f, c = seed * 0.25f, t0 = 0, t1 = 0;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (MODE == 0) {           // 64 dependent
            asm volatile(R8(R8("v_fmac_f32 %0, %1, %0\n\t")) : "+v"(a0) : "v"(b));
        } else if constexpr (MODE == 1) {    // 64 = 8 x 8 independent
            asm volatile(R8("v_fmac_f32 %0, %8, %9\n\tv_fmac_f32 %1, %8, %9\n\tv_fmac_f32 %2, %8, %9\n\tv_fmac_f32 %3, %8, %9\n\t"
                            "v_fmac_f32 %4, %8, %9\n\tv_fmac_f32 %5, %8, %9\n\tv_fmac_f32 %6, %8, %9\n\tv_fmac_f32 %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (MODE == 2) {    // 64 packed, 4 independent pairs
            asm volatile(R8(R8("v_pk_fma_f32 v[100:101], v[102:103], v[104:105], v[100:101]\n\t")) ::: "v100", "v101", "v102", "v103", "v104", "v105");
        } else if constexpr (MODE == 3) {    // 64 packed, 4 independent pairs
            asm volatile(R8("v_pk_fma_f32 v[100:101], v[108:109], v[110:111], v[100:101]\n\tv_pk_fma_f32 v[102:103], v[108:109], v[110:111], v[102:103]\n\t"
                            "v_pk_fma_f32 v[104:105], v[108:109], v[110:111], v[104:105]\n\tv_pk_fma_f32 v[106:107], v[108:109], v[110:111], v[106:107]\n\t"
                            "v_pk_fma_f32 v[100:101], v[108:109], v[110:111], v[100:101]\n\tv_pk_fma_f32 v[102:103], v[108:109], v[110:111], v[102:103]\n\t"
                            "v_pk_fma_f32 v[104:105], v[108:109], v[110:111], v[104:105]\n\tv_pk_fma_f32 v[106:107], v[108:109], v[110:111], v[106:107]\n\t")
                         ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");
        } else if constexpr (MODE == 6) {    // 64 v_fma_mix_f32 (f16 source 0, f32 sources 1 and 2), 8 independent accumulators
            asm volatile(R8("v_fma_mix_f32 %0, %8, %9, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %8, %9, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                            "v_fma_mix_f32 %2, %8, %9, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %8, %9, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                            "v_fma_mix_f32 %4, %8, %9, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %8, %9, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                            "v_fma_mix_f32 %6, %8, %9, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %7, %8, %9, %7 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (MODE == 7) {    // the LK tap on f16-packed LDS data: 5 of the 7 instructions are v_fma_mix_f32 (9 taps: 63)
#define TAPM(P, X, T) "v_fma_mix_f32 %8, " P ", 1.0, -" P " op_sel:[1,0,0] op_sel_hi:[1,0,1]\n\tv_fma_mix_f32 " X ", %10, %8, " P " op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t" \
                      "v_sub_f32 %8, " X ", " T "\n\tv_fmac_f32 " T ", %11, %8\n\t" \
                      "v_fma_mix_f32 %8, %10, 1.0, -" T " op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %6, %10, %8, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
                      "v_fma_mix_f32 %7, %11, %8, %7 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            asm volatile(TAPM("%0", "%1", "%2") TAPM("%3", "%4", "%5") TAPM("%0", "%2", "%1") TAPM("%3", "%5", "%4") TAPM("%0", "%1", "%2")
                         TAPM("%3", "%4", "%5") TAPM("%0", "%2", "%1") TAPM("%3", "%5", "%4") TAPM("%0", "%1", "%2")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(t0), "+v"(t1) : "v"(b), "v"(c));
        } else if constexpr (MODE == 4) {    // the LK tap, 9 taps through one tmp: 63 instructions
#define TAP(L0, L1, T) "v_sub_f32 %8, " L1 ", " L0 "\n\tv_fmac_f32 " L0 ", %10, %8\n\tv_sub_f32 %8, " L0 ", " T "\n\tv_fmac_f32 " T ", %11, %8\n\t" \
                       "v_sub_f32 %8, %10, " T "\n\tv_fmac_f32 %6, %11, %8\n\tv_fmac_f32 %7, %10, %8\n\t"
            asm volatile(TAP("%0", "%1", "%2") TAP("%1", "%2", "%3") TAP("%2", "%3", "%4") TAP("%3", "%4", "%5") TAP("%4", "%5", "%0")
                         TAP("%5", "%0", "%1") TAP("%0", "%1", "%2") TAP("%1", "%2", "%3") TAP("%2", "%3", "%4")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(t0), "+v"(t1) : "v"(b), "v"(c));
        } else {                             // the same, two taps interleaved instruction by instruction on two tmps (two pixels per lane): 56 instructions
#define TAP2(L0, L1, T, M0, M1, U) "v_sub_f32 %8, " L1 ", " L0 "\n\tv_sub_f32 %9, " M1 ", " M0 "\n\tv_fmac_f32 " L0 ", %10, %8\n\tv_fmac_f32 " M0 ", %10, %9\n\t" \
                       "v_sub_f32 %8, " L0 ", " T "\n\tv_sub_f32 %9, " M0 ", " U "\n\tv_fmac_f32 " T ", %11, %8\n\tv_fmac_f32 " U ", %11, %9\n\t" \
                       "v_sub_f32 %8, %10, " T "\n\tv_sub_f32 %9, %10, " U "\n\tv_fmac_f32 %6, %11, %8\n\tv_fmac_f32 %7, %11, %9\n\tv_fmac_f32 %6, %10, %8\n\tv_fmac_f32 %7, %10, %9\n\t"
            asm volatile(TAP2("%0", "%1", "%2", "%3", "%4", "%5") TAP2("%1", "%2", "%0", "%4", "%5", "%3") TAP2("%2", "%0", "%1", "%5", "%3", "%4")
                         TAP2("%0", "%1", "%2", "%3", "%4", "%5")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(t0), "+v"(t1) : "v"(b), "v"(c));
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + t0 + t1 == 123456.0f) out[0] = a0;
}

template <int MODE>
void run(const char* name, int per_iter, float* d, int cus) {
    printf("%-44s", name);
    for (int wps : {1, 2, 4, 6, 8}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int blocks = cus * wps;                     // 256 threads = one wave per SIMD per block
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %d w/SIMD %5.2f", wps, ms * 1e-3 * 2.4e9 / ((double)wps * ITER * per_iter));
    }
    printf("   (clocks at 2.4 GHz per wave instruction per SIMD)\n");
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float* d; CHECK(hipMalloc(&d, 64));
    printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    run<0>("v_fmac_f32, dependent chain", 64, d, p.multiProcessorCount);
    run<1>("v_fmac_f32, 8 independent accumulators", 64, d, p.multiProcessorCount);
    run<2>("v_pk_fma_f32, dependent chain", 64, d, p.multiProcessorCount);
    run<3>("v_pk_fma_f32, 4 independent pairs", 64, d, p.multiProcessorCount);
    run<4>("LK tap x 9 through one tmp", 63, d, p.multiProcessorCount);
    run<5>("LK tap, two pixels interleaved, two tmps", 56, d, p.multiProcessorCount);
    run<6>("v_fma_mix_f32, 8 independent accumulators", 64, d, p.multiProcessorCount);
    run<7>("LK tap x 9 with 5 of 7 as v_fma_mix_f32", 63, d, p.multiProcessorCount);
    return 0;
}
