// ubench_sad.hip -- issue-rate microbenchmark for the packed SAD instructions on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_sad.hip -o tools/ubench_sad ; run on the GPU box.
// Each wave runs ITER x 32 independent-chain instructions; 8 accumulator chains per lane hide the
// dependent-issue latency.  Reports wave-instructions per cycle per CU at the measured wall time and
// a nominal 2.4 GHz, next to v_add_u32 (known: 2 cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 2000;

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned* out, unsigned seed) {
    unsigned long long a[8];
    unsigned long long w = ((unsigned long long)(threadIdx.x * 2654435761u) << 32) | (seed + threadIdx.x);
    unsigned c = __builtin_amdgcn_readfirstlane(seed * 77u + blockIdx.x);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = k;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (OP == 0) a[k] = __builtin_amdgcn_qsad_pk_u16_u8(w, c, a[k]);
                else if (OP == 1) a[k] = (unsigned long long)__builtin_amdgcn_sad_u8((unsigned)w, c, (unsigned)a[k]);
                else if (OP == 2) a[k] = (unsigned long long)((unsigned)a[k] + (unsigned)w);
                else if (OP == 3) a[k] = __builtin_amdgcn_mqsad_pk_u16_u8(w, c, a[k]);
                else if (OP == 4) a[k] = (unsigned long long)__builtin_amdgcn_sad_u16((unsigned)w, c, (unsigned)a[k]);
                else if (OP == 5) a[k] = (unsigned long long)__builtin_amdgcn_msad_u8((unsigned)w, c, (unsigned)a[k]);
            }
        }
        asm volatile("" : "+v"(w));
    }
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s ^= a[k];
    if ((unsigned)(s ^ (s >> 32)) == 0x12345678u) out[0] = (unsigned)s;
}

template <int OP>
void run(const char* name, unsigned* d_out, int cus) {
    const int blocks = cus * 8;          // 8 WGs x 4 waves = 32 waves per CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 4 * ITER * 32;
    const double per_cu_per_s = wave_instr / cus / (ms * 1e-3);
    printf("%-22s %8.3f ms  %7.3f wave-instr/clk/CU @2.4GHz  => %5.2f cycles per wave-instr per SIMD\n", name, ms,
           per_cu_per_s / 2.4e9, 4.0 / (per_cu_per_s / 2.4e9));
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    unsigned* d; CHECK(hipMalloc(&d, 64));
    run<2>("v_add_u32", d, p.multiProcessorCount);
    run<1>("v_sad_u8", d, p.multiProcessorCount);
    run<0>("v_qsad_pk_u16_u8", d, p.multiProcessorCount);
    run<3>("v_mqsad_pk_u16_u8", d, p.multiProcessorCount);
    run<4>("v_sad_u16", d, p.multiProcessorCount);
    run<5>("v_msad_u8", d, p.multiProcessorCount);
    return 0;
}
