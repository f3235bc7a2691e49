// Micro-benchmark: cost of a device-wide barrier between 254 resident workgroups of 1024 threads on gfx950
// (a) cooperative-groups grid.sync() under hipLaunchCooperativeKernel, (b) a hand-rolled monotonic-counter barrier.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(1024) void k_cg(int iters, float* out) {
    cg::grid_group grid = cg::this_grid();
    float acc = threadIdx.x;
    for (int i = 0; i < iters; ++i) { acc = acc * 1.0001f + 1.0f; grid.sync(); }
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = acc;
}

__global__ __launch_bounds__(1024) void k_ctr(int iters, unsigned* counter, float* out) {
    float acc = threadIdx.x;
    const unsigned nwg = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        acc = acc * 1.0001f + 1.0f;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)(i + 1) * nwg;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = acc;
}

int main() {
    float* out; unsigned* ctr;
    hipMalloc(&out, 4); hipMalloc(&ctr, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nwg : {8, 64, 254}) {
        for (int iters : {30, 300}) {
            void* args[] = {&iters, &out};
            float ms;
            hipLaunchCooperativeKernel((void*)k_cg, dim3(nwg), dim3(1024), args, 0, 0);   // warm
            hipEventRecord(a);
            hipError_t e = hipLaunchCooperativeKernel((void*)k_cg, dim3(nwg), dim3(1024), args, 0, 0);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            printf("cg    nwg=%3d iters=%3d  %.3f ms  -> %.2f us per sync (%s)\n", nwg, iters, ms, ms * 1e3 / iters, hipGetErrorString(e));
            hipMemset(ctr, 0, 4);
            hipLaunchKernelGGL(k_ctr, dim3(nwg), dim3(1024), 0, 0, iters, ctr, out);
            hipMemset(ctr, 0, 4);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_ctr, dim3(nwg), dim3(1024), 0, 0, iters, ctr, out);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            printf("ctr   nwg=%3d iters=%3d  %.3f ms  -> %.2f us per sync\n", nwg, iters, ms, ms * 1e3 / iters);
        }
    }
    return 0;
}
