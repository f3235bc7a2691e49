// ubench_dpp.hip -- VERDICT r4 item 5 (i): what an LK tap costs when the I-side record (I, Ix, Iy: today one ds_read_b128 per tap) comes
// from a NEIGHBOURING LANE's registers through DPP operands instead of from LDS.  A DPP row is 16 lanes; a 9-wide window needs the
// records of lanes i .. i + 8, so for tap t > 0 the lanes i > 15 - t must take theirs from the next 16 columns (a second register
// set): every consumer of a record field is issued twice, once with row_shl:t on set A (lanes whose source is inside the row write),
// once with row_shr:(16 - t) on set B (the other lanes write; lanes without a valid source are disabled, bound_ctrl off).
//   base   7 VALU + 10/9 ds_read_b32 + 1 ds_read_b128 per tap            (the product's tap: lk.hip LK_ROW9_TAP + LK_ROW9_USE)
//   dpp    4 VALU + 3 x 2 DPP VALU + 10/9 ds_read_b32 + 2/9 ds_read_b128 per tap (tap 0 needs no shift: 3 instead of 6)
//   dpp1   the same with ONE DPP form per consumer (as if no seam existed: the lower bound of any DPP scheme): 7 VALU
// Prints clocks per tap per SIMD at 2 / 4 / 6 / 8 waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 tools/ubench_dpp.hip -o tools/ubench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 1500;

#define TAP4 "v_sub_f32 %[tmp], %[l1], %[l0]\n\tv_fmac_f32 %[l0], %[ax], %[tmp]\n\tv_sub_f32 %[tmp], %[l0], %[t]\n\tv_fmac_f32 %[t], %[ay], %[tmp]\n\t"
#define USE_LDS "v_sub_f32 %[tmp], v104, %[t]\n\tv_fmac_f32 %[bx], v105, %[tmp]\n\tv_fmac_f32 %[by], v106, %[tmp]\n\t"
#define USE_DPP2(K, K16) "v_sub_f32_dpp %[tmp], v104, %[t] row_shl:" #K " row_mask:0xf bank_mask:0xf\n\tv_sub_f32_dpp %[tmp], v108, %[t] row_shr:" #K16 " row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f32_dpp %[bx], v105, %[tmp] row_shl:" #K " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %[bx], v109, %[tmp] row_shr:" #K16 " row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f32_dpp %[by], v106, %[tmp] row_shl:" #K " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %[by], v110, %[tmp] row_shr:" #K16 " row_mask:0xf bank_mask:0xf\n\t"
#define USE_DPP1(K) "v_sub_f32_dpp %[tmp], v104, %[t] row_shl:" #K " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %[bx], v105, %[tmp] row_shl:" #K " row_mask:0xf bank_mask:0xf\n\t" \
                    "v_fmac_f32_dpp %[by], v106, %[tmp] row_shl:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define RD32(O) "ds_read_b32 v100, %[a32] offset:" #O "\n\t"
#define RDQ(R, O) "ds_read_b128 v[" R "], %[a128] offset:" #O "\n\t"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    __shared__ float4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned a32 = (unsigned)(uintptr_t)&buf[0] + wave * 1024 + lane * 4, a128 = (unsigned)(uintptr_t)&buf[0] + wave * 2048 + lane * 16;
    float l0 = seed, l1 = seed + 1, t = seed + 2, bx = 0, by = 0, tmp = 0, ax = seed * 0.5f, ay = seed * 0.25f;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (MODE == 0)            // the product's row: per tap 1 b32 + 1 b128 (+ one more b32 per row), waited for one tap later
            asm volatile(RD32(0) RD32(4) RDQ("104:107", 0)
                         TAP4 USE_LDS RD32(8) RDQ("104:107", 16) "s_waitcnt lgkmcnt(2)\n\t" TAP4 USE_LDS RD32(12) RDQ("104:107", 32) "s_waitcnt lgkmcnt(2)\n\t"
                         TAP4 USE_LDS RD32(16) RDQ("104:107", 48) "s_waitcnt lgkmcnt(2)\n\t" TAP4 USE_LDS RD32(20) RDQ("104:107", 64) "s_waitcnt lgkmcnt(2)\n\t"
                         TAP4 USE_LDS RD32(24) RDQ("104:107", 80) "s_waitcnt lgkmcnt(2)\n\t" TAP4 USE_LDS RD32(28) RDQ("104:107", 96) "s_waitcnt lgkmcnt(2)\n\t"
                         TAP4 USE_LDS RD32(32) RDQ("104:107", 112) "s_waitcnt lgkmcnt(2)\n\t" TAP4 USE_LDS RD32(36) RDQ("104:107", 128) "s_waitcnt lgkmcnt(2)\n\t"
                         TAP4 USE_LDS "s_waitcnt lgkmcnt(0)\n\t"
                         : [l0] "+v"(l0), [l1] "+v"(l1), [t] "+v"(t), [bx] "+v"(bx), [by] "+v"(by), [tmp] "+v"(tmp)
                         : [ax] "v"(ax), [ay] "v"(ay), [a32] "v"(a32), [a128] "v"(a128) : "v100", "v104", "v105", "v106", "v107", "memory");
        else if constexpr (MODE == 1)       // DPP with the 16-lane seam: two record sets per row (2 b128), every consumer twice
            asm volatile(RD32(0) RD32(4) RDQ("104:107", 0) RDQ("108:111", 256) "s_waitcnt lgkmcnt(0)\n\t"
                         TAP4 USE_LDS RD32(8) TAP4 USE_DPP2(1, 15) RD32(12) "s_waitcnt lgkmcnt(1)\n\t" TAP4 USE_DPP2(2, 14) RD32(16) "s_waitcnt lgkmcnt(1)\n\t"
                         TAP4 USE_DPP2(3, 13) RD32(20) "s_waitcnt lgkmcnt(1)\n\t" TAP4 USE_DPP2(4, 12) RD32(24) "s_waitcnt lgkmcnt(1)\n\t"
                         TAP4 USE_DPP2(5, 11) RD32(28) "s_waitcnt lgkmcnt(1)\n\t" TAP4 USE_DPP2(6, 10) RD32(32) "s_waitcnt lgkmcnt(1)\n\t"
                         TAP4 USE_DPP2(7, 9) RD32(36) "s_waitcnt lgkmcnt(1)\n\t" TAP4 USE_DPP2(8, 8) "s_waitcnt lgkmcnt(0)\n\t"
                         : [l0] "+v"(l0), [l1] "+v"(l1), [t] "+v"(t), [bx] "+v"(bx), [by] "+v"(by), [tmp] "+v"(tmp)
                         : [ax] "v"(ax), [ay] "v"(ay), [a32] "v"(a32), [a128] "v"(a128)
                         : "v100", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "memory");
        else                                // DPP as if there were no seam (lower bound): one DPP form per consumer, one record set per row
            asm volatile(RD32(0) RD32(4) RDQ("104:107", 0) "s_waitcnt lgkmcnt(0)\n\t"
                         TAP4 USE_LDS RD32(8) TAP4 USE_DPP1(1) RD32(12) "s_waitcnt lgkmcnt(1)\n\t" TAP4 USE_DPP1(2) RD32(16) "s_waitcnt lgkmcnt(1)\n\t"
                         TAP4 USE_DPP1(3) RD32(20) "s_waitcnt lgkmcnt(1)\n\t" TAP4 USE_DPP1(4) RD32(24) "s_waitcnt lgkmcnt(1)\n\t"
                         TAP4 USE_DPP1(5) RD32(28) "s_waitcnt lgkmcnt(1)\n\t" TAP4 USE_DPP1(6) RD32(32) "s_waitcnt lgkmcnt(1)\n\t"
                         TAP4 USE_DPP1(7) RD32(36) "s_waitcnt lgkmcnt(1)\n\t" TAP4 USE_DPP1(8) "s_waitcnt lgkmcnt(0)\n\t"
                         : [l0] "+v"(l0), [l1] "+v"(l1), [t] "+v"(t), [bx] "+v"(bx), [by] "+v"(by), [tmp] "+v"(tmp)
                         : [ax] "v"(ax), [ay] "v"(ay), [a32] "v"(a32), [a128] "v"(a128) : "v100", "v104", "v105", "v106", "v107", "memory");
    }
    if (l0 + l1 + t + bx + by + tmp == 123456.0f) out[0] = l0;
}

template <int MODE>
void run(const char* name, float* d, int cus) {
    printf("  %-66s", name);
    for (int wps : {2, 4, 6, 8}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<MODE>), dim3(cus * wps), dim3(256), 0, 0, d, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE>), dim3(cus * wps), dim3(256), 0, 0, d, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %dw %6.2f", wps, ms * 1e-3 * 2.4e9 / ((double)wps * ITER * 9));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float* d; CHECK(hipMalloc(&d, 64));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs; clocks (2.4 GHz) per tap per SIMD (a 9-tap window row) at 2 / 4 / 6 / 8 waves per SIMD\n", p.gcnArchName, cus);
    run<0>("records from LDS: 7 VALU + 10/9 b32 + 1 b128 per tap (product)", d, cus);
    run<1>("records through DPP, 16-lane seam: 9.3 VALU + 10/9 b32 + 2/9 b128", d, cus);
    run<2>("records through DPP, no seam (lower bound): 7 VALU + 10/9 b32 + 1/9 b128", d, cus);
    return 0;
}
