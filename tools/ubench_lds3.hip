// ubench_lds3.hip -- ds_read_b64 / ds_read2_b32 at 4-byte (not 8-byte) aligned addresses on gfx950 (round 4): lane i reads the two
// floats at element index i + ODD, i.e. overlapping pairs at a 4-byte stride -- the LK rows' texel pattern if two texels came per read.
// Prints correctness and LDS clocks per wave instruction per CU.  build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lds3.hip -o ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4000;

template <int MODE>   // 0: ds_read_b64, 1: ds_read2_b32 offset1 = offset0 + 1, 2: ds_read_b32 x 2
__global__ __launch_bounds__(256, 6) void k(float* out, unsigned odd, unsigned stride_b) {
    __shared__ float buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = (float)i;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(uintptr_t)&buf[0] + wave * 4096 + lane * stride_b + odd * 4;
    float acc = 0.0f;
    float2 first = make_float2(0, 0);
    for (int it = 0; it < ITER; ++it) {
        if constexpr (MODE == 0) {
            asm volatile("ds_read_b64 v[64:65], %1\n\tds_read_b64 v[66:67], %1 offset:8\n\tds_read_b64 v[68:69], %1 offset:16\n\tds_read_b64 v[70:71], %1 offset:24\n\t"
                         "ds_read_b64 v[72:73], %1 offset:256\n\tds_read_b64 v[74:75], %1 offset:264\n\tds_read_b64 v[76:77], %1 offset:272\n\tds_read_b64 v[78:79], %1 offset:280\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                         : "+v"(acc) : "v"(base) : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","memory");
            if (it == 0) asm volatile("v_mov_b32 %0, v64\n\tv_mov_b32 %1, v65" : "=v"(first.x), "=v"(first.y));
        } else if constexpr (MODE == 1) {
            asm volatile("ds_read2_b32 v[64:65], %1 offset0:0 offset1:1\n\tds_read2_b32 v[66:67], %1 offset0:2 offset1:3\n\tds_read2_b32 v[68:69], %1 offset0:4 offset1:5\n\tds_read2_b32 v[70:71], %1 offset0:6 offset1:7\n\t"
                         "ds_read2_b32 v[72:73], %1 offset0:64 offset1:65\n\tds_read2_b32 v[74:75], %1 offset0:66 offset1:67\n\tds_read2_b32 v[76:77], %1 offset0:68 offset1:69\n\tds_read2_b32 v[78:79], %1 offset0:70 offset1:71\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                         : "+v"(acc) : "v"(base) : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","memory");
            if (it == 0) asm volatile("v_mov_b32 %0, v64\n\tv_mov_b32 %1, v65" : "=v"(first.x), "=v"(first.y));
        } else {
            asm volatile("ds_read_b32 v64, %1\n\tds_read_b32 v65, %1 offset:4\n\tds_read_b32 v66, %1 offset:8\n\tds_read_b32 v67, %1 offset:12\n\t"
                         "ds_read_b32 v68, %1 offset:256\n\tds_read_b32 v69, %1 offset:260\n\tds_read_b32 v70, %1 offset:264\n\tds_read_b32 v71, %1 offset:268\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                         : "+v"(acc) : "v"(base) : "v64","v65","v66","v67","v68","v69","v70","v71","memory");
            if (it == 0) asm volatile("v_mov_b32 %0, v64\n\tv_mov_b32 %1, v65" : "=v"(first.x), "=v"(first.y));
        }
    }
    if (blockIdx.x == 0 && wave == 0) { out[2 * lane] = first.x; out[2 * lane + 1] = first.y; }
    if (acc == 123456.0f) out[200] = acc;
}

template <int MODE>
void run(const char* name, unsigned odd, unsigned stride_b, float* d_out, int cus) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(cus * 6), dim3(256), 0, 0, d_out, odd, stride_b);
    CHECK(hipDeviceSynchronize());
    float h[128];
    CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) { const float e = (float)(l * (stride_b / 4) + odd); if (h[2 * l] != e || h[2 * l + 1] != e + 1) ++bad; }
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(cus * 6), dim3(256), 0, 0, d_out, odd, stride_b);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr_per_cu = 6.0 * 4 * ITER * 8;
    printf("  %-14s first element %s, lane stride %2u B: %s  %6.2f LDS clocks per wave instruction per CU\n", name, odd ? "odd " : "even", stride_b,
           bad ? "WRONG DATA" : "data ok   ", ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float* d; CHECK(hipMalloc(&d, 4096));
    printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    for (unsigned stride : {4u, 8u}) for (unsigned odd : {0u, 1u}) {
        run<0>("ds_read_b64", odd, stride, d, p.multiProcessorCount);
        run<1>("ds_read2_b32", odd, stride, d, p.multiProcessorCount);
        run<2>("ds_read_b32", odd, stride, d, p.multiProcessorCount);
    }
    return 0;
}
