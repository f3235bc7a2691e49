// ubench_lds.hip -- LDS read throughput per CU on gfx950 by instruction width, for the access patterns of the LK level kernel
// (tools: hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o tools/ubench_lds ; run on the GPU box).
// Each wave issues ITER x 16 reads of one kind with 8 in flight (s_waitcnt per 8); 24 waves per CU (6 workgroups of 256).
// Patterns: "rec16" = lane l reads 16 B records at (l + k) * 16 (the tile taps: conflict-free for b128); "dword" = lane l
// reads dwords at (l + k) * 4 (the texel rows).  Reported: LDS clocks per wave instruction per CU at 2.4 GHz nominal and
// bytes per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4000;

template <int KIND>
__global__ __launch_bounds__(256, 6) void lds_kernel(float* out) {
    __shared__ float4 buf[1024];                       // 16 KB
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base16 = (unsigned)(uintptr_t)&buf[0] + (lane + wave * 64) * 16;
    const unsigned base4 = (unsigned)(uintptr_t)&buf[0] + (lane + wave * 64) * 4;
    float acc = 0.0f;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (KIND == 0) {         // ds_read_b128, 16 per iteration
            asm volatile(
                "ds_read_b128 v[64:67], %1\n\tds_read_b128 v[68:71], %1 offset:16\n\tds_read_b128 v[72:75], %1 offset:32\n\tds_read_b128 v[76:79], %1 offset:48\n\t"
                "ds_read_b128 v[64:67], %1 offset:64\n\tds_read_b128 v[68:71], %1 offset:80\n\tds_read_b128 v[72:75], %1 offset:96\n\tds_read_b128 v[76:79], %1 offset:112\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "ds_read_b128 v[64:67], %1\n\tds_read_b128 v[68:71], %1 offset:16\n\tds_read_b128 v[72:75], %1 offset:32\n\tds_read_b128 v[76:79], %1 offset:48\n\t"
                "ds_read_b128 v[64:67], %1 offset:64\n\tds_read_b128 v[68:71], %1 offset:80\n\tds_read_b128 v[72:75], %1 offset:96\n\tds_read_b128 v[76:79], %1 offset:112\n\t"
                "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                : "+v"(acc) : "v"(base16) : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","memory");
        } else if constexpr (KIND == 1) {  // ds_read_b96
            asm volatile(
                "ds_read_b96 v[64:66], %1\n\tds_read_b96 v[68:70], %1 offset:16\n\tds_read_b96 v[72:74], %1 offset:32\n\tds_read_b96 v[76:78], %1 offset:48\n\t"
                "ds_read_b96 v[64:66], %1 offset:64\n\tds_read_b96 v[68:70], %1 offset:80\n\tds_read_b96 v[72:74], %1 offset:96\n\tds_read_b96 v[76:78], %1 offset:112\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "ds_read_b96 v[64:66], %1\n\tds_read_b96 v[68:70], %1 offset:16\n\tds_read_b96 v[72:74], %1 offset:32\n\tds_read_b96 v[76:78], %1 offset:48\n\t"
                "ds_read_b96 v[64:66], %1 offset:64\n\tds_read_b96 v[68:70], %1 offset:80\n\tds_read_b96 v[72:74], %1 offset:96\n\tds_read_b96 v[76:78], %1 offset:112\n\t"
                "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                : "+v"(acc) : "v"(base16) : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","memory");
        } else if constexpr (KIND == 2) {  // ds_read_b64 on 16-byte records (8 of the 16 bytes)
            asm volatile(
                "ds_read_b64 v[64:65], %1\n\tds_read_b64 v[68:69], %1 offset:16\n\tds_read_b64 v[72:73], %1 offset:32\n\tds_read_b64 v[76:77], %1 offset:48\n\t"
                "ds_read_b64 v[64:65], %1 offset:64\n\tds_read_b64 v[68:69], %1 offset:80\n\tds_read_b64 v[72:73], %1 offset:96\n\tds_read_b64 v[76:77], %1 offset:112\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "ds_read_b64 v[64:65], %1\n\tds_read_b64 v[68:69], %1 offset:16\n\tds_read_b64 v[72:73], %1 offset:32\n\tds_read_b64 v[76:77], %1 offset:48\n\t"
                "ds_read_b64 v[64:65], %1 offset:64\n\tds_read_b64 v[68:69], %1 offset:80\n\tds_read_b64 v[72:73], %1 offset:96\n\tds_read_b64 v[76:77], %1 offset:112\n\t"
                "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                : "+v"(acc) : "v"(base16) : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","memory");
        } else if constexpr (KIND == 3) {  // ds_read_b32 dwords
            asm volatile(
                "ds_read_b32 v64, %1\n\tds_read_b32 v65, %1 offset:4\n\tds_read_b32 v66, %1 offset:8\n\tds_read_b32 v67, %1 offset:12\n\t"
                "ds_read_b32 v68, %1 offset:16\n\tds_read_b32 v69, %1 offset:20\n\tds_read_b32 v70, %1 offset:24\n\tds_read_b32 v71, %1 offset:28\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "ds_read_b32 v64, %1\n\tds_read_b32 v65, %1 offset:4\n\tds_read_b32 v66, %1 offset:8\n\tds_read_b32 v67, %1 offset:12\n\t"
                "ds_read_b32 v68, %1 offset:16\n\tds_read_b32 v69, %1 offset:20\n\tds_read_b32 v70, %1 offset:24\n\tds_read_b32 v71, %1 offset:28\n\t"
                "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                : "+v"(acc) : "v"(base4) : "v64","v65","v66","v67","v68","v69","v70","v71","memory");
        } else if constexpr (KIND == 4) {  // ds_read2_b32 dword pairs
            asm volatile(
                "ds_read2_b32 v[64:65], %1 offset1:1\n\tds_read2_b32 v[66:67], %1 offset0:2 offset1:3\n\tds_read2_b32 v[68:69], %1 offset0:4 offset1:5\n\tds_read2_b32 v[70:71], %1 offset0:6 offset1:7\n\t"
                "ds_read2_b32 v[72:73], %1 offset0:8 offset1:9\n\tds_read2_b32 v[74:75], %1 offset0:10 offset1:11\n\tds_read2_b32 v[76:77], %1 offset0:12 offset1:13\n\tds_read2_b32 v[78:79], %1 offset0:14 offset1:15\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "ds_read2_b32 v[64:65], %1 offset1:1\n\tds_read2_b32 v[66:67], %1 offset0:2 offset1:3\n\tds_read2_b32 v[68:69], %1 offset0:4 offset1:5\n\tds_read2_b32 v[70:71], %1 offset0:6 offset1:7\n\t"
                "ds_read2_b32 v[72:73], %1 offset0:8 offset1:9\n\tds_read2_b32 v[74:75], %1 offset0:10 offset1:11\n\tds_read2_b32 v[76:77], %1 offset0:12 offset1:13\n\tds_read2_b32 v[78:79], %1 offset0:14 offset1:15\n\t"
                "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                : "+v"(acc) : "v"(base4) : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","memory");
        } else {                           // ds_read_b128 on 4-byte strided dwords: lane l reads 16 B at l * 4 (unaligned-to-16, overlapping): the texel row as one wide read
            asm volatile(
                "ds_read_b64 v[64:65], %1\n\tds_read_b64 v[66:67], %1 offset:8\n\tds_read_b64 v[68:69], %1 offset:16\n\tds_read_b64 v[70:71], %1 offset:24\n\t"
                "ds_read_b64 v[72:73], %1 offset:32\n\tds_read_b64 v[74:75], %1 offset:40\n\tds_read_b64 v[76:77], %1 offset:48\n\tds_read_b64 v[78:79], %1 offset:56\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "ds_read_b64 v[64:65], %1\n\tds_read_b64 v[66:67], %1 offset:8\n\tds_read_b64 v[68:69], %1 offset:16\n\tds_read_b64 v[70:71], %1 offset:24\n\t"
                "ds_read_b64 v[72:73], %1 offset:32\n\tds_read_b64 v[74:75], %1 offset:40\n\tds_read_b64 v[76:77], %1 offset:48\n\tds_read_b64 v[78:79], %1 offset:56\n\t"
                "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                : "+v"(acc) : "v"(base4 & ~7u) : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","memory");
        }
    }
    if (acc == 123456.0f) out[0] = acc;
}

template <int KIND>
void run(const char* label, int bytes_per_lane, float* d_out, int cus) {
    const int blocks = cus * 6;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(lds_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(lds_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr_per_cu = 6.0 * 4 * ITER * 16;
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-34s %8.3f ms  %6.2f LDS clocks per wave instruction per CU  %7.1f useful B/clk/CU\n", label, ms, clk / wave_instr_per_cu,
           wave_instr_per_cu * 64 * bytes_per_lane / clk);
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float* d; CHECK(hipMalloc(&d, 64));
    printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    run<0>("ds_read_b128  (16 B records)", 16, d, p.multiProcessorCount);
    run<1>("ds_read_b96   (16 B records)", 12, d, p.multiProcessorCount);
    run<2>("ds_read_b64   (16 B records)", 8, d, p.multiProcessorCount);
    run<3>("ds_read_b32   (dwords)", 4, d, p.multiProcessorCount);
    run<4>("ds_read2_b32  (dword pairs)", 8, d, p.multiProcessorCount);
    run<5>("ds_read_b64   (8 B at lane*4&~7)", 8, d, p.multiProcessorCount);
    return 0;
}
