// ubench_allgather.hip -- what ONE step's exchange of the Almeida cluster solver costs with the arithmetic taken out:
// nwg co-resident workgroups (one wave each, round-robin over the XCDs), every round each publishes a 16-byte tagged
// granule (write-through store) and polls until it has seen all nwg granules of the round -- the solver's "allgather"
// (every workgroup folds all partials itself; no broadcast, no separate barrier).  Variants: the polling loop (load, wait,
// check, sleep | the same without the sleep | K polls in flight), and a stretch of local work between rounds (s_sleep)
// as long as a step's records + reduction + update, to see how much of the solver's gather time is lateness of the last
// publisher rather than the hand-off itself.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_allgather.hip -o tools/ubench_allgather ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int ROUNDS = 600;
constexpr unsigned SPIN_LIMIT = 1u << 20;
constexpr size_t kReplica = 64 * 256;      // granules between the replicas of a row (256 KB)
constexpr size_t kParity = 4 * kReplica;    // and between the two parities

__device__ __forceinline__ void st4(u4* p, u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void issue(u4& x, const u4* p) { asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(x) : "v"(p) : "memory"); }
template <int LEFT> __device__ __forceinline__ void wait_left(u4& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(LEFT) : "memory"); }

// MODE 0: load/wait/check/s_sleep 1; MODE 1: the same without the sleep; MODE 2 / 4: that many polls in flight
template <int MODE>
__global__ __launch_bounds__(64) void allgather(u4* slots, int nwg, unsigned base, int work_sleeps, int stride, int replicas, unsigned long long* out) {
    const int blk = blockIdx.x, lane = threadIdx.x;
    const int src = lane < nwg ? lane : nwg - 1;
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned timeouts = 0;
    unsigned long long gather_cycles = 0;
    for (int r = 1; r <= ROUNDS && !timeouts; ++r) {
        for (int w = 0; w < work_sleeps; ++w) __builtin_amdgcn_s_sleep(8);     // 512 clocks each: the step's local work
        const unsigned tag = base + r;
        u4* row = slots + (size_t)(r & 1) * kParity;
        const unsigned long long g0 = __builtin_readcyclecounter();
        if (lane < replicas) { u4 v; v.x = blk; v.y = tag; v.z = r; v.w = tag; st4(row + (size_t)lane * kReplica + (size_t)blk * stride, v); }
        if constexpr (MODE <= 1) {
            u4 x;
            for (unsigned spins = 0;; ++spins) {
                issue(x, row + (size_t)src * stride); wait_left<0>(x);
                if (__all(x.y == tag && x.w == tag)) break;
                if (spins > SPIN_LIMIT) { ++timeouts; break; }
                if constexpr (MODE == 0) __builtin_amdgcn_s_sleep(1);
            }
        } else {
            constexpr int K = MODE;
            u4 x[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { issue(x[k], row + (size_t)src * stride + (replicas > 1 ? (size_t)k * kReplica : 0)); if (k + 1 < K) __builtin_amdgcn_s_sleep(K == 2 ? 6 : 3); }
            bool done = false;
            for (unsigned spins = 0; !done; ++spins) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (done) break;
                    wait_left<K - 1>(x[k]);
                    if (__all(x[k].y == tag && x[k].w == tag)) { done = true; break; }
                    issue(x[k], row + (size_t)src * stride + (replicas > 1 ? (size_t)k * kReplica : 0));
                }
                if (spins > SPIN_LIMIT) { ++timeouts; break; }
            }
            // (the solver drains after its update; here the local work stands in for it)
            for (int w = 0; w < 2; ++w) __builtin_amdgcn_s_sleep(8);
            if constexpr (K == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) :: "memory");
        }
        gather_cycles += __builtin_readcyclecounter() - g0;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[blk * 3] = t1 - t0; out[blk * 3 + 1] = timeouts; out[blk * 3 + 2] = gather_cycles; }
}

template <int MODE>
static void run(const char* name, u4* slots, unsigned long long* d_out, int nwg, int work, int stride, int replicas, unsigned& base) {
    std::vector<unsigned long long> h(3 * 256);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(allgather<MODE>, dim3(nwg), dim3(64), 0, 0, slots, nwg, base, work, stride, replicas, d_out);
        CHECK(hipDeviceSynchronize());
        base += ROUNDS + 8;
    }
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
    double tot = 0, gat = 0; unsigned long long to = 0;
    for (int b = 0; b < nwg; ++b) { tot += (double)h[b * 3]; to += h[b * 3 + 1]; gat += (double)h[b * 3 + 2]; }
    const int extra = MODE >= 2 ? 2 : 0;
    printf("%-28s stride %4d B  nwg=%3d local work %5d clk: %7.0f cycles per round, of which publish->all seen %6.0f%s\n", name, stride * 16, nwg, (work + extra) * 512,
           tot / nwg / ROUNDS, gat / nwg / ROUNDS - extra * 512.0, to ? "   ** TIMED OUT **" : "");
}


// the same exchange in the solver's shape: 256-thread workgroups, a barrier (the block reduction's) before thread 0
// publishes, wave 2 alone polls, a barrier ends the round
template <int MODE>
__global__ __launch_bounds__(256) void allgather_wg(u4* slots, int nwg, unsigned base, int work_sleeps, int stride, int pubwave, unsigned long long* out) {
    const int blk = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int src = lane < nwg ? lane : nwg - 1;
    unsigned long long t0 = __builtin_readcyclecounter();
    __shared__ int fail;
    if (threadIdx.x == 0) fail = 0;
    __syncthreads();
    unsigned long long gather_cycles = 0;
    for (int r = 1; r <= ROUNDS; ++r) {
        for (int w = 0; w < work_sleeps; ++w) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        const unsigned tag = base + r;
        u4* row = slots + (size_t)(r & 1) * kParity;
        const unsigned long long g0 = __builtin_readcyclecounter();
        if (threadIdx.x == pubwave * 64) { u4 v; v.x = blk; v.y = tag; v.z = r; v.w = tag; st4(row + (size_t)blk * stride, v); }
        if (wave == 2) {
            u4 x;
            for (unsigned spins = 0;; ++spins) {
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(row + (size_t)src * stride) : "memory");
                if (__all(x.y == tag && x.w == tag)) break;
                if (spins > SPIN_LIMIT) { fail = 1; break; }
                if constexpr (MODE == 0) __builtin_amdgcn_s_sleep(1);
            }
            gather_cycles += __builtin_readcyclecounter() - g0;
        }
        __syncthreads();
        if (fail) break;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 128) { out[blk * 3] = t1 - t0; out[blk * 3 + 1] = fail; out[blk * 3 + 2] = gather_cycles; }
}
template <int MODE>
static void run_wg(const char* name, u4* slots, unsigned long long* d_out, int nwg, int work, int stride, int pubwave, unsigned& base) {
    std::vector<unsigned long long> h(3 * 256);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(allgather_wg<MODE>, dim3(nwg), dim3(256), 0, 0, slots, nwg, base, work, stride, pubwave, d_out);
        CHECK(hipDeviceSynchronize());
        base += ROUNDS + 8;
    }
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
    double tot = 0, gat = 0; unsigned long long to = 0;
    for (int b = 0; b < nwg; ++b) { tot += (double)h[b * 3]; to += h[b * 3 + 1]; gat += (double)h[b * 3 + 2]; }
    printf("%-28s stride %4d B  nwg=%3d local work %5d clk: %7.0f cycles per round, of which publish->all seen %6.0f%s\n", name, stride * 16, nwg, work * 512,
           tot / nwg / ROUNDS, gat / nwg / ROUNDS, to ? "   ** TIMED OUT **" : "");
}

int main() {
    u4* slots; unsigned long long* d_out;
    const size_t bytes = 2 * kParity * sizeof(u4);
    CHECK(hipMalloc(&slots, bytes)); CHECK(hipMemset(slots, 0, bytes));
    CHECK(hipMalloc(&d_out, 3 * 256 * sizeof(unsigned long long)));
    unsigned base = 16;
    for (int nwg : {2, 8, 32, 64})
        for (int stride : {1, 8, 16, 256}) {
            const int work = 8;
            run<0>("load/wait/check/sleep", slots, d_out, nwg, work, stride, 1, base);
            run<1>("load/wait/check", slots, d_out, nwg, work, stride, 1, base);
            run<2>("2 polls in flight", slots, d_out, nwg, work, stride, 1, base);
            run<4>("4 polls in flight", slots, d_out, nwg, work, stride, 1, base);
            run<2>("2 polls, 2 replicas", slots, d_out, nwg, work, stride, 2, base);
            run<4>("4 polls, 4 replicas", slots, d_out, nwg, work, stride, 4, base);
        }
    for (int nwg : {8, 32})
        for (int stride : {1, 8}) {
            run_wg<0>("256 thr, thread 0 publishes", slots, d_out, nwg, 8, stride, 0, base);
            run_wg<1>("256 thr, t0 pub, no sleep", slots, d_out, nwg, 8, stride, 0, base);
            run_wg<0>("256 thr, wave 2 publishes", slots, d_out, nwg, 8, stride, 2, base);
        }
    return 0;
}
