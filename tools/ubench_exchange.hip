// ubench_exchange.hip -- round-trip latency of a flag exchange between two workgroups on gfx950, by placement (same
// XCD / different XCDs) and by the cache-coherence bits of the store and the polling load.  It answers: how much of the
// Almeida cluster solver's per-step gather is cross-XCD memory latency, and which bits are enough when every
// participant sits on ONE XCD (shared L2).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_exchange.hip -o tools/ubench_exchange ; run on the GPU box.
//
// 64 workgroups of 64 threads are launched; each records its XCC_ID (s_getreg HW_REG_XCC_ID).  Workgroup `a` and
// workgroup `b` then play ping-pong for ROUNDS rounds on two 16-byte granules 4 KB apart: a stores round r, b polls
// until it sees r and answers, a polls for the answer.  s_memtime in a, per round trip = two one-way exchanges.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ROUNDS = 2000;
constexpr unsigned SPIN_LIMIT = 1u << 20;

template <int BITS> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
    if constexpr (BITS == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if constexpr (BITS == 1) asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else if constexpr (BITS == 2) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int BITS> __device__ __forceinline__ unsigned ld(const unsigned* p) {
    unsigned v;
    if constexpr (BITS == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (BITS == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (BITS == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int SB, int LB>
__global__ __launch_bounds__(64) void pingpong(unsigned* flags, int a, int b, unsigned base, unsigned* xcc, unsigned long long* out) {
    const int blk = blockIdx.x;
    if (threadIdx.x == 0) xcc[blk] = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // HW_REG_XCC_ID, bits 3:0
    if (blk != a && blk != b) return;
    if (threadIdx.x != 0) return;
    unsigned* ping = flags;            // written by a
    unsigned* pong = flags + 1024;     // written by b
    unsigned long long t0 = 0, t1 = 0;
    unsigned timeouts = 0;
    if (blk == a) {
        t0 = __builtin_readcyclecounter();
        for (int r = 1; r <= ROUNDS; ++r) {
            st<SB>(ping, base + r);
            unsigned spins = 0;
            while (ld<LB>(pong) != base + r) { if (++spins > SPIN_LIMIT) { ++timeouts; break; } }
            if (timeouts) break;
        }
        t1 = __builtin_readcyclecounter();
        out[0] = t1 - t0; out[1] = timeouts;
    } else {
        for (int r = 1; r <= ROUNDS; ++r) {
            unsigned spins = 0;
            while (ld<LB>(ping) != base + r) { if (++spins > SPIN_LIMIT) { ++timeouts; break; } }
            if (timeouts) break;
            st<SB>(pong, base + r);
        }
        out[2] = timeouts;
    }
}

template <int SB, int LB>
static void run(const char* label, unsigned* flags, unsigned* xcc, unsigned long long* out, int a, int b, unsigned& base) {
    CHECK(hipMemset(out, 0, 3 * sizeof(unsigned long long)));
    hipLaunchKernelGGL((pingpong<SB, LB>), dim3(64), dim3(64), 0, 0, flags, a, b, base, xcc, out);
    CHECK(hipDeviceSynchronize());
    base += ROUNDS + 16;
    unsigned long long h[3];
    unsigned hx[64];
    CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
    printf("%-28s wg %2d (xcc %u) <-> wg %2d (xcc %u): %8.0f cycles per round trip%s\n", label, a, hx[a], b, hx[b],
           (double)h[0] / ROUNDS, (h[1] || h[2]) ? "   ** TIMED OUT: the bits do not make the store visible **" : "");
}

int main() {
    unsigned *flags, *xcc;
    unsigned long long* out;
    CHECK(hipMalloc(&flags, 8192));
    CHECK(hipMalloc(&xcc, 64 * 4));
    CHECK(hipMalloc(&out, 3 * 8));
    CHECK(hipMemset(flags, 0, 8192));
    unsigned base = 0;
    // placement under round-robin dispatch: workgroup i -> XCD i % 8
    for (int pass = 0; pass < 2; ++pass) {
        const int a = 0, b = pass == 0 ? 8 : 1;
        printf("--- %s ---\n", pass == 0 ? "same XCD expected (wg 0, wg 8)" : "different XCDs expected (wg 0, wg 1)");
        run<3, 3>("store sc0 sc1 / load sc0 sc1", flags, xcc, out, a, b, base);
        run<2, 2>("store sc1     / load sc1", flags, xcc, out, a, b, base);
        run<2, 3>("store sc1     / load sc0 sc1", flags, xcc, out, a, b, base);
        run<3, 2>("store sc0 sc1 / load sc1", flags, xcc, out, a, b, base);
        run<1, 1>("store sc0     / load sc0", flags, xcc, out, a, b, base);
        run<0, 1>("store plain   / load sc0", flags, xcc, out, a, b, base);
        run<0, 2>("store plain   / load sc1", flags, xcc, out, a, b, base);
    }
    unsigned hx[64];
    CHECK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
    printf("XCC_ID of workgroups 0..63:");
    for (int i = 0; i < 64; ++i) printf(" %u", hx[i]);
    printf("\n");
    return 0;
}
