// ubench_f64.hip -- f64 VALU issue rates on gfx950 (round 5, for farneback.hip's f64 sums): clocks per wave64 instruction per SIMD for
// v_add_f64 / v_mul_f64 / v_fma_f64 / v_cvt_f64_f32 / v_cvt_f32_f64 / v_rcp_f64, eight independent destinations, 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o tools/ubench_f64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 2000;
#define R8(X) X X X X X X X X
#define OPS8(OP) OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" OP " %3, %3, %8\n\t" OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" OP " %6, %6, %8\n\t" OP " %7, %7, %8\n\t"

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 1e-9;
    float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (MODE == 0) asm volatile(R8(OPS8("v_add_f64")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else if constexpr (MODE == 1) asm volatile(R8(OPS8("v_mul_f64")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else if constexpr (MODE == 2)
            asm volatile(R8("v_fma_f64 %0, %8, %8, %0\n\tv_fma_f64 %1, %8, %8, %1\n\tv_fma_f64 %2, %8, %8, %2\n\tv_fma_f64 %3, %8, %8, %3\n\t"
                            "v_fma_f64 %4, %8, %8, %4\n\tv_fma_f64 %5, %8, %8, %5\n\tv_fma_f64 %6, %8, %8, %6\n\tv_fma_f64 %7, %8, %8, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        else if constexpr (MODE == 3)
            asm volatile(R8("v_cvt_f64_f32 %0, %8\n\tv_cvt_f64_f32 %1, %9\n\tv_cvt_f64_f32 %2, %10\n\tv_cvt_f64_f32 %3, %11\n\t"
                            "v_cvt_f64_f32 %4, %12\n\tv_cvt_f64_f32 %5, %13\n\tv_cvt_f64_f32 %6, %14\n\tv_cvt_f64_f32 %7, %15\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
        else if constexpr (MODE == 4)
            asm volatile(R8("v_cvt_f32_f64 %0, %8\n\tv_cvt_f32_f64 %1, %9\n\tv_cvt_f32_f64 %2, %10\n\tv_cvt_f32_f64 %3, %11\n\t"
                            "v_cvt_f32_f64 %4, %12\n\tv_cvt_f32_f64 %5, %13\n\tv_cvt_f32_f64 %6, %14\n\tv_cvt_f32_f64 %7, %15\n\t")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        else if constexpr (MODE == 5)
            asm volatile(R8("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\tv_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else    // the expansion's horizontal step for one k as the compiler sees it: 9 f32 + 5 cvt + 8 f64
            asm volatile(R8("v_add_f32 %8, %8, %9\n\tv_sub_f32 %9, %9, %10\n\tv_add_f32 %10, %10, %11\n\tv_sub_f32 %11, %11, %12\n\tv_add_f32 %12, %12, %13\n\t"
                            "v_mul_f32 %13, %9, %14\n\tv_mul_f32 %14, %10, %15\n\tv_mul_f32 %15, %11, %8\n\tv_mul_f32 %8, %12, %9\n\t"
                            "v_cvt_f64_f32 %0, %8\n\tv_cvt_f64_f32 %1, %13\n\tv_cvt_f64_f32 %2, %14\n\tv_cvt_f64_f32 %3, %15\n\tv_cvt_f64_f32 %4, %9\n\t"
                            "v_mul_f64 %5, %0, %16\n\tv_add_f64 %6, %6, %5\n\tv_mul_f64 %5, %0, %16\n\tv_add_f64 %7, %7, %5\n\t"
                            "v_add_f64 %6, %6, %1\n\tv_add_f64 %7, %7, %2\n\tv_add_f64 %6, %6, %3\n\tv_add_f64 %7, %7, %4\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                           "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b));
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 123456.0) out[0] = a0;
}

template <int MODE>
void run(const char* name, int per_iter, double* d, int cus) {
    printf("%-52s", name);
    for (int wps : {1, 2, 4, 8}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int blocks = cus * wps;                     // 256 threads = one wave per SIMD per block
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %d w/SIMD %5.2f", wps, ms * 1e-3 * 2.4e9 / ((double)wps * ITER * per_iter));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    double* d; CHECK(hipMalloc(&d, 64));
    printf("%s, %d CUs; clocks at 2.4 GHz per wave64 instruction per SIMD\n", p.gcnArchName, p.multiProcessorCount);
    run<0>("v_add_f64, 8 independent", 64, d, p.multiProcessorCount);
    run<1>("v_mul_f64, 8 independent", 64, d, p.multiProcessorCount);
    run<2>("v_fma_f64, 8 independent", 64, d, p.multiProcessorCount);
    run<3>("v_cvt_f64_f32, 8 independent", 64, d, p.multiProcessorCount);
    run<4>("v_cvt_f32_f64, 8 independent", 64, d, p.multiProcessorCount);
    run<5>("v_rcp_f64, 8 independent", 64, d, p.multiProcessorCount);
    run<6>("expansion step: 9 f32 + 5 cvt + 8 f64 (per instr)", 22 * 8, d, p.multiProcessorCount);
    return 0;
}
