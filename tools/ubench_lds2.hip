// ubench_lds2.hip -- LDS read cost by the OFFSET between the two half-waves of the LK level kernel's reads (round 4): a wave covers two
// pixel rows of a 32-wide tile, so lanes 0..31 read consecutive elements and lanes 32..63 the same columns one row further -- OFF bytes
// apart, OFF = the row pitch of the array.  ds_read_b32 (texels, pitch of jl[][]) and ds_read_b128 (records, pitch of tile[][]).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lds2.hip -o tools/ubench_lds2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4000;

template <bool B128>
__global__ __launch_bounds__(256, 6) void k(float* out, unsigned off_bytes) {
    __shared__ float4 buf[1536];                       // 24 KB
    for (int i = threadIdx.x; i < 1536; i += 256) buf[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(uintptr_t)&buf[0] + wave * 2048 + (lane & 31) * (B128 ? 16 : 4) + (lane >> 5) * off_bytes;
    float acc = 0.0f;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (B128) {
            asm volatile(
                "ds_read_b128 v[64:67], %1\n\tds_read_b128 v[68:71], %1 offset:16\n\tds_read_b128 v[72:75], %1 offset:32\n\tds_read_b128 v[76:79], %1 offset:48\n\t"
                "ds_read_b128 v[64:67], %1 offset:64\n\tds_read_b128 v[68:71], %1 offset:80\n\tds_read_b128 v[72:75], %1 offset:96\n\tds_read_b128 v[76:79], %1 offset:112\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "ds_read_b128 v[64:67], %1\n\tds_read_b128 v[68:71], %1 offset:16\n\tds_read_b128 v[72:75], %1 offset:32\n\tds_read_b128 v[76:79], %1 offset:48\n\t"
                "ds_read_b128 v[64:67], %1 offset:64\n\tds_read_b128 v[68:71], %1 offset:80\n\tds_read_b128 v[72:75], %1 offset:96\n\tds_read_b128 v[76:79], %1 offset:112\n\t"
                "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                : "+v"(acc) : "v"(base) : "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","memory");
        } else {
            asm volatile(
                "ds_read_b32 v64, %1\n\tds_read_b32 v65, %1 offset:4\n\tds_read_b32 v66, %1 offset:8\n\tds_read_b32 v67, %1 offset:12\n\t"
                "ds_read_b32 v68, %1 offset:16\n\tds_read_b32 v69, %1 offset:20\n\tds_read_b32 v70, %1 offset:24\n\tds_read_b32 v71, %1 offset:28\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "ds_read_b32 v64, %1\n\tds_read_b32 v65, %1 offset:4\n\tds_read_b32 v66, %1 offset:8\n\tds_read_b32 v67, %1 offset:12\n\t"
                "ds_read_b32 v68, %1 offset:16\n\tds_read_b32 v69, %1 offset:20\n\tds_read_b32 v70, %1 offset:24\n\tds_read_b32 v71, %1 offset:28\n\t"
                "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, v64"
                : "+v"(acc) : "v"(base) : "v64","v65","v66","v67","v68","v69","v70","v71","memory");
        }
    }
    if (acc == 123456.0f) out[0] = acc;
}

template <bool B128>
void run(unsigned off, float* d_out, int cus) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<B128>, dim3(cus * 6), dim3(256), 0, 0, d_out, off);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<B128>, dim3(cus * 6), dim3(256), 0, 0, d_out, off);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr_per_cu = 6.0 * 4 * ITER * 16;
    printf("  %-13s half-wave offset %4u B: %6.2f LDS clocks per wave instruction per CU\n", B128 ? "ds_read_b128" : "ds_read_b32", off,
           ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    float* d; CHECK(hipMalloc(&d, 64));
    printf("%s, %d CUs; lanes 0..31 consecutive elements, lanes 32..63 the same + offset\n", p.gcnArchName, p.multiProcessorCount);
    for (unsigned off : {128u, 256u, 260u, 264u, 272u, 288u, 320u, 384u}) run<false>(off, d, p.multiProcessorCount);
    for (unsigned off : {512u, 640u, 656u, 672u, 704u, 768u, 528u, 576u}) run<true>(off, d, p.multiProcessorCount);
    return 0;
}
