// ubench_valu.hip -- issue-rate microbenchmark of the VALU instructions the hot path is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu ; run on the GPU box.
// Every measured instruction is an `asm volatile` statement (the compiler can neither fold nor reorder it away:
// the r01 table's v_add_u32 row was folded by the optimiser and read 0.6 cycles).  Each wave runs ITER x 32
// instructions over 8 independent accumulator chains per lane; 32 waves per CU.  Cycles are per wave64 instruction
// per SIMD at the nominal 2.4 GHz; the effective clock (lower under load) is printed from s_memtime-free wall time
// of a pure v_mov chain for scale.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 2000;

#define OP32(name, text) \
    struct name { static constexpr const char* label = #name; \
        __device__ static __forceinline__ void op(unsigned long long& a, unsigned long long w, unsigned c) { \
            unsigned lo = (unsigned)a; asm volatile(text : "+v"(lo) : "v"((unsigned)w), "s"(c)); a = lo; } };
#define OP64(name, text) \
    struct name { static constexpr const char* label = #name; \
        __device__ static __forceinline__ void op(unsigned long long& a, unsigned long long w, unsigned c) { \
            asm volatile(text : "+v"(a) : "v"(w), "s"(c)); } };

OP32(v_add_u32, "v_add_u32 %0, %1, %0")
OP32(v_add_f32, "v_add_f32 %0, %1, %0")
OP32(v_mul_f32, "v_mul_f32 %0, %1, %0")
OP32(v_fma_f32, "v_fma_f32 %0, %1, %1, %0")
OP32(v_rcp_f32, "v_rcp_f32 %0, %0")
OP32(v_fmac_f32, "v_fmac_f32 %0, %1, %1")
OP32(v_fma_mix_f32, "v_fma_mix_f32 %0, %1, %1, %0 op_sel_hi:[1,0,0]")
OP32(v_cvt_f32_f16, "v_cvt_f32_f16 %0, %0")
OP32(v_min_u32, "v_min_u32 %0, %1, %0")
OP32(v_max_f32, "v_max_f32 %0, %1, %0")
OP32(v_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
OP32(v_sad_u8, "v_sad_u8 %0, %1, %2, %0")
OP32(v_pk_min_u16, "v_pk_min_u16 %0, %1, %0")
OP32(v_pk_add_u16, "v_pk_add_u16 %0, %1, %0")
OP32(v_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
OP32(v_perm_b32, "v_perm_b32 %0, %1, %0, %2")
OP32(v_alignbit_b32, "v_alignbit_b32 %0, %1, %0, 8")
OP64(v_pk_add_f32, "v_pk_add_f32 %0, %1, %0")
OP64(v_pk_mul_f32, "v_pk_mul_f32 %0, %1, %0")
OP64(v_pk_fma_f32, "v_pk_fma_f32 %0, %1, %1, %0")
OP64(v_qsad_pk_u16_u8, "v_qsad_pk_u16_u8 %0, %1, %2, %0")
OP64(v_lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")

template <class O>
__global__ __launch_bounds__(256) void rate_kernel(unsigned* out, unsigned seed) {
    unsigned long long a[8];
    unsigned long long w = ((unsigned long long)(threadIdx.x * 2654435761u) << 32) | (seed + threadIdx.x);
    unsigned c = __builtin_amdgcn_readfirstlane(seed * 77u + blockIdx.x);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = k + 0x3f8000003f800000ull;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) O::op(a[k], w, c);
        }
    }
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s ^= a[k];
    if ((unsigned)(s ^ (s >> 32)) == 0x12345678u) out[0] = (unsigned)s;
}

template <class O>
void run(unsigned* d_out, int cus) {
    const int blocks = cus * 8;          // 8 WGs x 4 waves = 32 waves per CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<O>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<O>, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 4 * ITER * 32;
    const double per_cu_per_s = wave_instr / cus / (ms * 1e-3);
    printf("%-22s %8.3f ms  %7.3f wave-instr/clk/CU @2.4GHz  => %5.2f cycles per wave-instr per SIMD\n", O::label, ms,
           per_cu_per_s / 2.4e9, 4.0 / (per_cu_per_s / 2.4e9));
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    unsigned* d; CHECK(hipMalloc(&d, 64));
    const int cus = p.multiProcessorCount;
    run<v_add_u32>(d, cus); run<v_add_f32>(d, cus); run<v_mul_f32>(d, cus); run<v_fma_f32>(d, cus);
    run<v_pk_add_f32>(d, cus); run<v_pk_mul_f32>(d, cus); run<v_pk_fma_f32>(d, cus);
    run<v_fmac_f32>(d, cus); run<v_fma_mix_f32>(d, cus); run<v_cvt_f32_f16>(d, cus);
    run<v_rcp_f32>(d, cus); run<v_max_f32>(d, cus); run<v_min_u32>(d, cus); run<v_cndmask_b32>(d, cus);
    run<v_pk_min_u16>(d, cus); run<v_pk_add_u16>(d, cus); run<v_cvt_f32_ubyte0>(d, cus); run<v_perm_b32>(d, cus);
    run<v_alignbit_b32>(d, cus); run<v_lshlrev_b64>(d, cus);
    run<v_sad_u8>(d, cus); run<v_qsad_pk_u16_u8>(d, cus);
    return 0;
}
