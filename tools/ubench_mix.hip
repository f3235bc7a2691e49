// ubench_mix.hip -- how f32 VALU work and LDS reads overlap on gfx950 (round 4).  One "tap" = NV v_fmac_f32 (four independent
// accumulators) + NB ds_read_b32 + NQ ds_read_b128 (conflict-free addresses: consecutive lanes, consecutive elements), the reads
// waited for one tap later (s_waitcnt lgkmcnt(NB + NQ)): the shape of a tap of the LK rows.  Prints clocks per tap per SIMD at
// 4 and 6 waves per SIMD beside what the two pipes alone would need (VALU 2.25 clocks per instruction per SIMD; LDS array 2 / 4
// clocks per b32 / b128 per CU = 8 / 16 clocks of each SIMD's time with four SIMDs sharing the array).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mix.hip -o tools/ubench_mix ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 1500, TAPS = 8;

template <int NV, int NB, int NQ>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    __shared__ float4 buf[1024];                       // 16 KB
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned a32 = (unsigned)(uintptr_t)&buf[0] + wave * 1024 + lane * 4, a128 = (unsigned)(uintptr_t)&buf[0] + wave * 2048 + lane * 16;
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = seed * 0.5f, c = seed * 0.25f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            if constexpr (NB >= 1) asm volatile("ds_read_b32 v100, %0 offset:0" :: "v"(a32) : "v100");
            if constexpr (NB >= 2) asm volatile("ds_read_b32 v101, %0 offset:256" :: "v"(a32) : "v101");
            if constexpr (NB >= 3) asm volatile("ds_read_b32 v102, %0 offset:512" :: "v"(a32) : "v102");
            if constexpr (NQ >= 1) asm volatile("ds_read_b128 v[104:107], %0 offset:0" :: "v"(a128) : "v104", "v105", "v106", "v107");
            if constexpr (NQ >= 2) asm volatile("ds_read_b128 v[108:111], %0 offset:1024" :: "v"(a128) : "v108", "v109", "v110", "v111");
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if ((v & 3) == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
                if ((v & 3) == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
                if ((v & 3) == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
                if ((v & 3) == 3) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
            }
            if constexpr (NB + NQ > 0) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NB + NQ));      // the previous tap's reads
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (a0 + a1 + a2 + a3 == 123456.0f) out[0] = a0;
}

template <int NV, int NB, int NQ>
void run(float* d, int cus) {
    printf("  %2d VALU + %d b32 + %d b128 per tap:", NV, NB, NQ);
    for (int wps : {2, 4, 6, 8}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<NV, NB, NQ>), dim3(cus * wps), dim3(256), 0, 0, d, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NV, NB, NQ>), dim3(cus * wps), dim3(256), 0, 0, d, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %dw %6.2f", wps, ms * 1e-3 * 2.4e9 / ((double)wps * ITER * TAPS));
    }
    printf("   | alone: VALU %5.2f, LDS %5.2f\n", NV * 2.25, NB * 8.0 + NQ * 16.0);
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float* d; CHECK(hipMalloc(&d, 64));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs; clocks (2.4 GHz) per tap per SIMD at 2 / 4 / 6 / 8 waves per SIMD\n", p.gcnArchName, cus);
    run<7, 0, 0>(d, cus); run<0, 1, 0>(d, cus); run<0, 0, 1>(d, cus); run<0, 1, 1>(d, cus); run<0, 2, 1>(d, cus);
    run<7, 1, 0>(d, cus); run<7, 0, 1>(d, cus); run<7, 1, 1>(d, cus); run<7, 2, 1>(d, cus); run<7, 3, 2>(d, cus);
    run<14, 2, 1>(d, cus); run<14, 0, 1>(d, cus); run<14, 2, 0>(d, cus); run<10, 1, 1>(d, cus); run<4, 1, 1>(d, cus);
    return 0;
}
