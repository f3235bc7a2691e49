// sad.hip -- N1: full-search SAD block matcher for gfx950 (the "hip_sad" Decoder's compute).
//
// No reference counterpart exists; the spec is include/ofps_hip.h /
// DESIGN.md "N1" and the output record follows av-decoder/src/lib.rs:404-419.
//
// Kernel shape (sad_qsad_kernel):
//   * one workgroup = 4 waves = 4 horizontally adjacent blocks; the (4B+2R) x (B+2R) search
//     window of the previous frame is staged once in LDS with coalesced dword row loads;
//   * one wave = one block.  The block of the current frame is wave-uniform, so it is read
//     through the scalar cache into SGPRs and feeds v_qsad_pk_u16_u8 as its 32-bit operand;
//   * a lane owns a (4 consecutive dx) x (K consecutive dy) patch of the candidate grid.  It
//     walks B+K-1 window rows once, reading B/4+1 dwords per row from LDS, and issues one
//     v_qsad_pk_u16_u8 per (row, 4-pixel column, dy) -- 16 absolute differences per lane-op,
//     byte alignment of the 4 dx shifts handled by the instruction; costs accumulate as
//     packed u16 (a 16x16 block's SAD <= 65280);
//   * the LDS row stride S is chosen so that lane l = chunk*NG + group hits bank l mod 32:
//     K*S == NG (mod 32) makes every ds_read_b32 conflict-free;
//   * argmin: per-lane strict-< scan over its 4K candidates on a 32-bit key (SAD<<16 | d2),
//     then a 64-bit (key, dy+R, dx+R) butterfly min over the wave.  The key is a total order,
//     so the result is independent of the reduction order and bit-exact vs the scalar oracle.
#include "common.hpp"

#include <cstdlib>
#include <utility>

namespace {

constexpr int kWavesPerWG = 4;

constexpr int pick_stride(int min_s, int k, int ng) {
    // smallest S >= min_s with K*S == NG (mod 32); K odd guarantees a solution within 32 steps
    for (int s = min_s; s < min_s + 64; ++s)
        if ((k * s) % 32 == ng % 32) return s;
    return min_s | 1;
}

template <int B, int R, int K>
struct SadCfg {
    static_assert(B % 4 == 0 && R % 4 == 0, "tile origin must stay dword aligned");
    static_assert(B * B * 255 <= 65535, "packed u16 SAD accumulators would overflow");
    static constexpr int NCAND = 2 * R + 1;
    static constexpr int NG = (NCAND + 3) / 4;              // dx groups of 4
    static constexpr int NC = (NCAND + K - 1) / K;          // dy chunks of K
    static constexpr int T = NG * NC;                       // lane tasks per block
    static constexpr int PASSES = (T + 63) / 64;
    static constexpr int BW = B / 4;                        // dwords per block row
    static constexpr int TILE_H = B + 2 * R;                // rows that hold image data
    static constexpr int TILE_ROWS = NC * K + B - 1;        // rows the last chunk may touch
    static constexpr int TILE_WD = (kWavesPerWG * B + 2 * R) / 4;
    static constexpr int MAXCOL = (kWavesPerWG - 1) * BW + (NG - 1) + BW;
    static constexpr int MIN_S = (MAXCOL + 1 > TILE_WD) ? MAXCOL + 1 : TILE_WD;
    static constexpr int S = pick_stride(MIN_S, K, NG);
    static constexpr int LDS_DWORDS = TILE_ROWS * S;
};

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}

struct __attribute__((packed, aligned(4))) U64A4 { unsigned long long v; };   // 8-byte LDS window, dword aligned

struct SadParams {
    const uint8_t* prev_base;   // pair k: prev = prev_base + k*prev_pitch, cur = cur_base + k*cur_pitch
    const uint8_t* cur_base;
    size_t prev_pitch, cur_pitch;
    int W, H, stride;
    int nbx, nby;
    float nx, ny;           // 1/W, 1/H computed on the host in f32 (av-decoder/src/lib.rs:404-405)
    float4* out_entries;
    int* out_best;          // may be null
    // strip kernels: ceil(2^48 / d) for d = strips per pair and strips per row -- the wave's strip index is split into (pair, block
    // row, first block) on the scalar unit (multiply-high + shift) instead of two emulated VALU divisions per strip
    unsigned long long div_spp, div_spr;
};
inline unsigned long long div_magic48(unsigned d) { return ((1ull << 48) + d - 1) / d; }
// floor(n / d) for n * d < 2^48 and n / d < 2^16 (strip indices: asserted where the magic numbers are made)
__device__ __forceinline__ unsigned div48(unsigned n, unsigned long long m) { return (unsigned)(((unsigned long long)n * m) >> 48); }

__device__ __forceinline__ void write_block_result(const SadParams& p, int pair, int bx, int by, int B, int R,
                                                   unsigned long long key) {
    const int sad = (int)(key >> 48);
    const int dy = (int)((key >> 8) & 0xFF) - R;
    const int dx = (int)(key & 0xFF) - R;
    const size_t k = ((size_t)pair * p.nby + by) * p.nbx + bx;
    const float cx = (float)(bx * B + B / 2 + dx), cy = (float)(by * B + B / 2 + dy);
    float4 e;
    e.x = cx * p.nx;
    e.y = cy * p.ny;
    e.z = ((float)dx / 1.0f) * (-p.nx);
    e.w = ((float)dy / 1.0f) * (-p.ny);
    p.out_entries[k] = e;
    if (p.out_best) {
        p.out_best[3 * k + 0] = dx;
        p.out_best[3 * k + 1] = dy;
        p.out_best[3 * k + 2] = sad;
    }
}

template <int B, int R, int K>
__global__ __launch_bounds__(256) void sad_qsad_kernel(const SadParams p) {
    using C = SadCfg<B, R, K>;
    __shared__ uint32_t tile[C::LDS_DWORDS];

    const int pair = blockIdx.z;
    const uint8_t* __restrict__ prev = p.prev_base + (size_t)pair * p.prev_pitch;
    const uint8_t* __restrict__ cur = p.cur_base + (size_t)pair * p.cur_pitch;
    const int by = blockIdx.y;
    const int bx0 = blockIdx.x * kWavesPerWG;
    const int tid = threadIdx.x;

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int bx = bx0 + wave;
    const int x0 = bx * B, y0 = by * B;

    // ---- current block: wave-uniform address -> scalar loads, values live in SGPRs.  Issued
    // before the window staging so their latency hides behind it.  Waves past the last block
    // column (they exit after the barrier) read the last valid block instead of running off the row.
    uint32_t c[B][C::BW];
    {
        const int bxc = bx < p.nbx ? bx : p.nbx - 1;
        const uint32_t* __restrict__ cp = reinterpret_cast<const uint32_t*>(cur + (size_t)y0 * p.stride + bxc * B);
        const int sdw = p.stride >> 2;
#pragma unroll
        for (int y = 0; y < B; ++y)
#pragma unroll
            for (int q = 0; q < C::BW; ++q) c[y][q] = cp[y * sdw + q];
    }

    // ---- stage the search window: coalesced dword loads along rows, guarded at the frame edge.
    // Cells outside the frame stay unwritten: only candidates that are masked out below read them.
    {
        const int tx0 = bx0 * B - R, ty0 = by * B - R;
        for (int idx = tid; idx < C::TILE_H * C::TILE_WD; idx += 256) {
            const int row = idx / C::TILE_WD, col = idx - row * C::TILE_WD;
            const int gx = tx0 + 4 * col, gy = ty0 + row;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                tile[row * C::S + col] = *reinterpret_cast<const uint32_t*>(prev + (size_t)gy * p.stride + gx);
        }
    }
    __syncthreads();
    if (bx >= p.nbx) return;

    unsigned long long best = ~0ull;
#pragma unroll 1
    for (int pass = 0; pass < C::PASSES; ++pass) {
        const int t = pass * 64 + lane;
        const bool active = t < C::T;
        const int tc = active ? t : 0;
        const int chunk = tc / C::NG, g = tc - chunk * C::NG;
        const int dx0 = -R + 4 * g, dy0 = -R + K * chunk;

        unsigned long long acc[K];
#pragma unroll
        for (int i = 0; i < K; ++i) acc[i] = 0;

        // Window rows are walked once.  hipcc's scheduler, left alone, hoists all B+K-1 rows of
        // ds_reads to the top (170+ VGPRs, 2 waves/SIMD); the sched_group_barrier ladder below
        // asks for "one row of LDS reads, then one row's worth of packed SADs" instead.
        const uint32_t* trow = tile + (K * chunk) * C::S + wave * C::BW + g;
#pragma unroll
        for (int rr = 0; rr < B + K - 1; ++rr) {
            // each 8-byte window is its own ds_read2_b32 straight into an even-aligned VGPR pair: LDS has
            // the bandwidth to spare (v_qsad issues every 16 cycles), and building the odd windows from a
            // 5-dword row with v_mov/v_pk_mov cost ~25% extra VALU issue slots.
            unsigned long long win[C::BW];
#pragma unroll
            for (int q = 0; q < C::BW; ++q) win[q] = reinterpret_cast<const U64A4*>(trow + rr * C::S + q)->v;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int y = rr - i;
                if (y >= 0 && y < B) {
#pragma unroll
                    for (int q = 0; q < C::BW; ++q) acc[i] = __builtin_amdgcn_qsad_pk_u16_u8(win[q], c[y][q], acc[i]);
                }
            }
        }
        // two rows of reads in flight ahead of the VALU work that consumes them
        __builtin_amdgcn_sched_group_barrier(0x100, C::BW, 0);
#pragma unroll
        for (int rr = 0; rr < B + K - 1; ++rr) {
            __builtin_amdgcn_sched_group_barrier(0x100, C::BW, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, K * C::BW, 0);
        }

        // keep the accumulate section one straight-line region: the empty asm pins the finished
        // sums here, so the scan's control flow below cannot pull the packed SADs out of the ladder
#pragma unroll
        for (int i = 0; i < K; ++i) asm volatile("" : "+v"(acc[i]));

        // ---- per-lane scan in (dy, dx) order with strict <: ties on (SAD, d2) keep the smaller (dy, dx)
        uint32_t bkey = 0xFFFFFFFFu;
        int bci = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int dy = dy0 + i;
            const int vy = (int)active & (int)(dy <= R) & (int)(y0 + dy >= 0) & (int)(y0 + dy + B <= p.H);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dx = dx0 + j;
                const int v = vy & (int)(dx <= R) & (int)(x0 + dx >= 0) & (int)(x0 + dx + B <= p.W);
                // v is 0/1: (v - 1) is all-ones for a masked candidate -> key saturates to 0xFFFFFFFF
                const uint32_t d2inv = (uint32_t)(dx * dx + dy * dy) | (uint32_t)(v - 1);
                const uint32_t half = (j < 2) ? (uint32_t)acc[i] : (uint32_t)(acc[i] >> 32);
                const uint32_t key = ((j & 1) ? (half & 0xFFFF0000u) : (half << 16)) | d2inv;
                const bool lt = key < bkey;
                bkey = lt ? key : bkey;
                bci = lt ? (i * 4 + j) : bci;
            }
        }
        if (bkey != 0xFFFFFFFFu) {
            const int bi = bci >> 2, bj = bci & 3;
            const unsigned long long k64 =
                ((unsigned long long)bkey << 32) | (uint32_t)(((dy0 + bi + R) << 8) | (dx0 + bj + R));
            best = k64 < best ? k64 : best;
        }
    }

#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = shfl_xor_u64(best, m);
        best = o < best ? o : best;
    }
    if (lane == 0) write_block_result(p, pair, bx, by, B, R, best);
}

// ------------------------------------------------------------------------------------------------
// sad_strip_kernel: the throughput kernel (B16 R16 = BASELINE configs[1], and the other table entries).
//
// rocprofv3 on sad_qsad_kernel<16,16,5> (profiles/r01/sad_block): VALU 98 % busy, 724 VALU instructions
// per block of which 320 are packed SADs -- per-block setup, LDS addressing and the argmin scan cost as
// much issue time as a quarter of the SADs, and 15 % of the issued SADs are masked candidates.  Here:
//   * one wave owns a strip of NB = 64/NG horizontally adjacent blocks, NG = 2R/4 (8 blocks for +-16);
//     lane = (block b, dx group g) covers dx in [-R, R-1] exactly (no masked dx) and walks ALL 2R+1 dy:
//     2(2R+1) packed-u16 accumulator VGPRs, its block of the current frame in B*B/4 VGPRs;
//   * the strip's (NB*B+2R) x (B+2R) window is staged once per wave with 16-byte coalesced loads; every
//     window row is read once (B/4 ds_read2_b32 per lane) and feeds up to B*B/4 packed SADs;
//   * the odd column dx = +R (2R+1 is never a multiple of 4) is a short second pass: lane = (block,
//     chunk of KE dy) with v_sad_u8 on dword-aligned window data (R % 4 == 0) -- 4 % extra SAD-unit
//     time instead of a ninth dx group that wastes 3 of its 4 shifts and one lane in 64;
//   * lane-local argmin key = SAD<<16 | d2 (dy^2 added to the row minimum), walked in ascending dy with a strict <: inside one dx group
//     |dx| is distinct, so equal (SAD, d2) in one lane means different dy and the walk keeps the spec's
//     (SAD, d2, dy, dx) order; masked columns saturate through a clamped add; vertical clipping is a
//     scalar branch (uniform per strip); a 64-bit key finishes the order across the NG lanes of a block;
//   * workgroup -> strip mapping is XCD-aware: the 8 XCDs (workgroup w runs on XCD w % 8) each take a
//     contiguous range of strips, so vertically adjacent strips share their halo rows in one L2.
template <int B, int R>
struct StripCfg {
    static_assert(B % 4 == 0 && R % 4 == 0, "tile origin and the dx = +R column must stay dword aligned");
    static_assert(B * B * 255 <= 65535, "packed u16 SAD accumulators would overflow");
    static constexpr int NCAND = 2 * R + 1;
    static constexpr int NGA = (2 * R) / 4;                  // active dx groups of 4 covering [-R, R-1]
    static constexpr int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
    // lanes per block: the next power of two, so the cross-lane argmin stays an aligned xor butterfly; for ranges
    // whose group count is not a power of two (+-12, 20, 24, 28) the lanes g >= NGA idle in the main pass (12-37 %)
    // -- still 20x the generic kernel
    static constexpr int NG = pow2_ceil(NGA);
    static_assert(NGA >= 1 && NG <= 64, "search range too wide for one wave per strip");
    static constexpr int NB = 64 / NG;                       // blocks per wave
    static constexpr int BW = B / 4;
    static constexpr int TILE_H = B + 2 * R;
    static constexpr int KE = (NCAND + NG - 1) / NG;         // dy per lane in the dx = +R pass
    static constexpr int EDGE_COL = (2 * R) / 4;             // dword offset of the dx = +R block inside a block's window
    // staging granule: the window origin bx0*B - R must be a multiple of it (NB*B and R both are)
    static constexpr int GRAN = ((NB * B) % 16 == 0 && R % 16 == 0) ? 16 : (((NB * B) % 8 == 0 && R % 8 == 0) ? 8 : 4);
    static constexpr int GDW = GRAN / 4;                          // dwords per granule
    static constexpr int TILE_WG = (NB * B + 2 * R + GRAN - 1) / GRAN;   // granules per window row
    static constexpr int MAXCOL = (NB - 1) * BW + (NG - 1) + BW;  // highest dword a lane touches
    static constexpr int SW = ((MAXCOL + 1 > TILE_WG * GDW ? MAXCOL + 1 : TILE_WG * GDW) + 3) / 4 * 4;  // 16-B rows
    static_assert((NB - 1) * BW + EDGE_COL + BW <= SW, "edge column must lie inside a window row");
    // the dx = +R pass gives lane chunk g the window rows g*KE .. g*KE + B + KE - 2: the last chunks run past the window (those dy are
    // masked); the tile is allocated that much taller so the walk needs no row clamp (uninitialised LDS, never used in a result)
    static constexpr int TILE_ROWS = TILE_H > NG * KE + B - 1 ? TILE_H : NG * KE + B - 1;
    static constexpr int TILE_DWORDS = TILE_ROWS * SW;
    static_assert(2 * R * R < 65536, "lane key holds d2 in 16 bits");
    // accumulators + current block + ~40 working registers: ask for 3 waves/SIMD (<= 168 VGPRs) when that fits,
    // otherwise hipcc spreads into all 256 registers it is allowed and occupancy drops to 2 for nothing
    // dy is walked in SPLIT passes of NP candidates so the packed accumulators never need more than ~66 VGPRs:
    // +-32 would otherwise hold 130 of them and drop to 2 waves per SIMD.  The window rows are re-read from LDS
    // (which has bandwidth to spare); staging, the current block and the argmin state are shared by the passes.
    static constexpr int SPLIT = (NCAND + 32) / 33;
    static constexpr int NP = (NCAND + SPLIT - 1) / SPLIT;
    static constexpr int MIN_WAVES = (2 * NP + B * BW + 40 <= 168) ? 3 : 2;
    // small ranges: (d2 << 6 | dy index) fits the low 16 bits of the lane key, so the key alone identifies the
    // candidate and the scan needs no separate index tracking
    static constexpr bool COMPACT_KEY = ((2 * R * R) << 6) + NCAND < 65536 && NCAND <= 64;
    // wider ranges (+-24 .. +-32): d2 << 3 | code still fits the low 16 bits, where the 3-bit code orders the (at most 8)
    // candidates of one lane that share a d2 -- inside a group of 4 consecutive dx the |dx| are distinct, so given d2 a
    // candidate is (which of the 4 dx, sign of dy): code = rank of |dx| for dy <= 0 (more negative dy = smaller |dx|
    // first), 7 - rank for dy > 0 (smaller dy = larger |dx| first).  The key alone is then the spec's (SAD, d2, dy, dx)
    // order within the lane and identifies the candidate: a plain min per dy, no index tracked beside it.
    static constexpr bool RANK_KEY = !COMPACT_KEY && ((2 * R * R) << 3) + 7 < 65536;
    static constexpr int KSHIFT = COMPACT_KEY ? 6 : (RANK_KEY ? 3 : 0);
};

// ds_read2_b32 reaches 255 dwords from its base register; left to itself hipcc spends one v_add per read on
// addresses.  The walk keeps ONE running dword index into the workgroup's LDS array that advances every
// ROWS_PER_BASE rows and is made opaque to the optimiser, so every read inside a group is base + immediate.
// (The opaque value is an integer index, not a pointer: hiding a pointer behind inline asm loses its LDS address
// space and turns every read into a flat load -- measured 30 % slower.)
template <int B, int R>
struct StripWalk {
    static constexpr int ROWS_PER_BASE = (255 - StripCfg<B, R>::BW) / StripCfg<B, R>::SW + 1;
};

template <int B, int R, int I0, int RR>
__device__ __forceinline__ void strip_row(unsigned long long (&acc)[StripCfg<B, R>::NP], const uint32_t (&c)[B][B / 4],
                                          const uint32_t* lds, uint32_t& base) {
    using C = StripCfg<B, R>;
    constexpr int G = StripWalk<B, R>::ROWS_PER_BASE;
    if constexpr (RR > 0 && RR % G == 0) {
        base += G * C::SW;
        asm volatile("" : "+v"(base));
    }
    unsigned long long win[C::BW];
#pragma unroll
    for (int q = 0; q < C::BW; ++q) win[q] = reinterpret_cast<const U64A4*>(lds + base + (RR % G) * C::SW + q)->v;
#pragma unroll
    for (int ii = 0; ii < C::NP; ++ii) {
        const int y = RR - ii;                                  // window row I0+RR belongs to candidate I0+ii, block row y
        if (I0 + ii < C::NCAND && y >= 0 && y < B) {
#pragma unroll
            for (int q = 0; q < C::BW; ++q) acc[ii] = __builtin_amdgcn_qsad_pk_u16_u8(win[q], c[y][q], acc[ii]);
        }
    }
}

// lds: the workgroup's LDS array; tile_off: dword index of this lane's window origin inside it
template <int B, int R, int I0, int... RR>
__device__ __forceinline__ void strip_rows(unsigned long long (&acc)[StripCfg<B, R>::NP], const uint32_t (&c)[B][B / 4],
                                           const uint32_t* lds, uint32_t tile_off, std::integer_sequence<int, RR...>) {
    uint32_t base = tile_off + I0 * StripCfg<B, R>::SW;
    asm volatile("" : "+v"(base));
    (strip_row<B, R, I0, RR>(acc, c, lds, base), ...);
}

// One dy pass: accumulate candidates I0 .. I0+NP-1 and fold them into the lane's running (key, dy index).
// key = SAD<<16 | dx^2 is minimised over the 4 dx first; dy^2 (a compile-time constant per step) is added to
// the row minimum afterwards -- adding the same constant to four keys does not change their order, and
// dx^2 + dy^2 < 65536 cannot carry into the SAD field; a clipped column holds all-ones and the clamped add
// keeps it saturated.
// The argmin's running minimum (round 5; profiles/r05/sad_colkeys.txt).  The kernel is bound by the VALU's SAD unit, and everything
// else it issues on the VALU is overhead: per dy the argmin used to cost 8 operations (4 key builds, 3 to reduce the 4 dx, 1 running
// min).  Now there is one running key per COLUMN (dx fixed: within a column the spec's order is (SAD, dy^2, dy), so the key's low
// half is a per-dy constant in an SGPR) and the running minimum is kept by the LDS UNIT, which has cycles to spare: ds_min_u32
// (no return) into a per-lane slot -- 4 key builds on the VALU per dy, 4 LDS atomics beside them.  The four column minima are joined
// with dx^2 once per strip.  -DOFPS_SAD_COLKEYS=0 (per-lane keys, rounds 1-4) and =1 (per-column keys, v_min3_u32 on the VALU) stay
// selectable for A/B builds (tools/sad_colkeys_ab.sh): 8x8 +-32: 17.28 / 16.98 / 16.73 ms per 64-pair 4K batch, same bits.
#ifndef OFPS_SAD_COLKEYS
#define OFPS_SAD_COLKEYS 2
#endif
constexpr int kSadColKeys = OFPS_SAD_COLKEYS;
constexpr int kColSlotDwords = kSadColKeys == 2 ? 4 * 64 : 0;       // LDS per wave for the column minima

template <int B, int R, int I0>
__device__ __forceinline__ void strip_pass(const uint32_t (&c)[B][B / 4], const uint32_t* lds, uint32_t tile_off,
                                           const uint32_t (&colk)[4], const uint32_t (&colk_pos)[4], int y0, int H, uint32_t& bkey, int& bi,
                                           uint32_t (&colmin)[4], uint32_t* amin) {
    using C = StripCfg<B, R>;
    unsigned long long acc[C::NP];
#pragma unroll
    for (int ii = 0; ii < C::NP; ++ii) acc[ii] = 0;
    constexpr int ROWS = ((I0 + C::NP < C::NCAND ? C::NP : C::NCAND - I0) + B - 1);
    strip_rows<B, R, I0>(acc, c, lds, tile_off, std::make_integer_sequence<int, ROWS>{});
#pragma unroll
    for (int ii = 0; ii < C::NP; ++ii) asm volatile("" : "+v"(acc[ii]));
    if constexpr (kSadColKeys != 0 && (C::RANK_KEY || C::COMPACT_KEY)) {
        // per-column keys: SAD << 16 | low, low = dy^2 << 1 | (dy > 0) (RANK_KEY) or dy^2 << 6 | dy index (COMPACT_KEY): the order
        // (SAD, dy^2, dy) inside a column.  A dy clipped by the frame (uniform over the strip) takes the all-ones constant, which
        // turns both build forms into all-ones: no branch.
        auto low_of = [&](int ii) -> uint32_t {
            const int i = I0 + ii, dy = -R + i;
            const bool ok = i < C::NCAND && y0 + dy >= 0 && y0 + dy + B <= H;
            return ok ? (uint32_t)(C::RANK_KEY ? ((dy * dy) << 1 | (dy > 0 ? 1 : 0)) : ((dy * dy) << 6 | i)) : 0xFFFFFFFFu;
        };
        // `low` lives in an SGPR; (x & 0xFFFF0000) | low with the mask as a literal would put TWO scalar operands on one VOP3
        // (gfx9: one constant-bus read per instruction) and hipcc splits it into v_and_b32 + v_or_b32 -- 6 operations per dy
        // for the four builds instead of 4, which is what made round 4's per-column variant slower than the per-lane one
        // (profiles/r05/sad_colkeys.txt).  With the mask in a VGPR the build is one v_and_or_b32.
        uint32_t hmask = 0xFFFF0000u;
        asm volatile("" : "+v"(hmask));
        auto keys_of = [&](int ii, uint32_t low, uint32_t (&k)[4]) {
            const uint32_t lo = (uint32_t)acc[ii], hi = (uint32_t)(acc[ii] >> 32);
            k[0] = (lo << 16) | low; k[1] = (lo & hmask) | low; k[2] = (hi << 16) | low; k[3] = (hi & hmask) | low;
        };
        if constexpr (kSadColKeys == 1) {
#pragma unroll
            for (int ii = 0; ii < C::NP; ii += 2) {                          // two dy per v_min3_u32 and column
                if (I0 + ii >= C::NCAND) break;
                uint32_t ka[4], kb[4];
                keys_of(ii, low_of(ii), ka);
                if (ii + 1 < C::NP && I0 + ii + 1 < C::NCAND) {
                    keys_of(ii + 1, low_of(ii + 1), kb);
#pragma unroll
                    for (int j = 0; j < 4; ++j)      // (left to itself hipcc forms v_min3_u32 for half of these and two v_min_u32 for the rest)
                        asm("v_min3_u32 %0, %1, %2, %3" : "=v"(colmin[j]) : "v"(colmin[j]), "v"(ka[j]), "v"(kb[j]));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) colmin[j] = min(colmin[j], ka[j]);
                }
            }
        } else {
#pragma unroll
            for (int ii = 0; ii < C::NP; ++ii) {
                if (I0 + ii >= C::NCAND) break;
                uint32_t k[4];
                keys_of(ii, low_of(ii), k);
#pragma unroll
                for (int j = 0; j < 4; ++j) __hip_atomic_fetch_min(amin + 64 * j, k[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        return;
    }
#pragma unroll
    for (int ii = 0; ii < C::NP; ++ii) {
        const int i = I0 + ii;
        const int dy = -R + i;
        if (i < C::NCAND && y0 + dy >= 0 && y0 + dy + B <= H) {          // uniform over the strip: scalar branch
            const uint32_t lo = (uint32_t)acc[ii], hi = (uint32_t)(acc[ii] >> 32);
            const uint32_t (&ck)[4] = (C::RANK_KEY && dy > 0) ? colk_pos : colk;
            const uint32_t k0 = (lo << 16) | ck[0];
            const uint32_t k1 = (lo & 0xFFFF0000u) | ck[1];
            const uint32_t k2 = (hi << 16) | ck[2];
            const uint32_t k3 = (hi & 0xFFFF0000u) | ck[3];
            if constexpr (C::RANK_KEY) {
                bkey = min(bkey, __builtin_elementwise_add_sat(min(min(k0, k1), min(k2, k3)), (uint32_t)((dy * dy) << 3)));
            } else if constexpr (C::COMPACT_KEY) {
                const uint32_t m = __builtin_elementwise_add_sat(min(min(k0, k1), min(k2, k3)), (uint32_t)((dy * dy) << 6 | i));
                bkey = min(bkey, m);            // ascending dy + the dy index inside the key: ties keep the smaller dy
            } else {
                const uint32_t m = __builtin_elementwise_add_sat(min(min(k0, k1), min(k2, k3)), (uint32_t)(dy * dy));
                const bool lt = m < bkey;
                bkey = lt ? m : bkey;
                bi = lt ? i : bi;
            }
        }
    }
}

template <int B, int R, int... S>
__device__ __forceinline__ void strip_passes(const uint32_t (&c)[B][B / 4], const uint32_t* lds, uint32_t tile_off,
                                             const uint32_t (&colk)[4], const uint32_t (&colk_pos)[4], int y0, int H, uint32_t& bkey, int& bi,
                                             uint32_t (&colmin)[4], uint32_t* amin, std::integer_sequence<int, S...>) {
    (strip_pass<B, R, S * StripCfg<B, R>::NP>(c, lds, tile_off, colk, colk_pos, y0, H, bkey, bi, colmin, amin), ...);
}

constexpr int kStripWaves = 4;      // independent waves (strips) per workgroup; no workgroup barrier

// one strip: staged window -> main pass -> dx = +R pass -> argmin -> records
template <int B, int R>
__device__ __forceinline__ void strip_body(const SadParams& p, int strips_per_row, int strip, uint32_t* tiles, int wave, int lane) {
    using C = StripCfg<B, R>;
    const int strips_per_pair = strips_per_row * p.nby;
    const int pair = (int)div48((unsigned)strip, p.div_spp);                   // strip / strips_per_pair, scalar unit
    const int rem = strip - pair * strips_per_pair;
    const int by = (int)div48((unsigned)rem, p.div_spr);                       // rem / strips_per_row
    const int bx0 = (rem - by * strips_per_row) * C::NB;

    const uint8_t* __restrict__ prev = p.prev_base + (size_t)pair * p.prev_pitch;
    const uint8_t* __restrict__ cur = p.cur_base + (size_t)pair * p.cur_pitch;
    uint32_t* tile = tiles + wave * C::TILE_DWORDS;

    const int b = lane / C::NG;          // block of the strip
    const int g = lane % C::NG;          // dx group in the main pass, dy chunk in the dx = +R pass
    const int bx = bx0 + b;
    const bool blk_on = bx < p.nbx;
    const int y0 = by * B;

    // ---- this lane's block of the current frame -> VGPRs (lanes of one block read the same lines)
    uint32_t c[B][C::BW];
    {
        const int bxc = blk_on ? bx : p.nbx - 1;
        const uint8_t* cp = cur + (size_t)y0 * p.stride + bxc * B;
#pragma unroll
        for (int y = 0; y < B; ++y) {
            if constexpr (C::BW % 4 == 0) {
#pragma unroll
                for (int q4 = 0; q4 < C::BW; q4 += 4) {
                    const uint4 v = *reinterpret_cast<const uint4*>(cp + (size_t)y * p.stride + 4 * q4);
                    c[y][q4 + 0] = v.x; c[y][q4 + 1] = v.y; c[y][q4 + 2] = v.z; c[y][q4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int q2 = 0; q2 < C::BW; q2 += 2) {
                    const uint2 v = *reinterpret_cast<const uint2*>(cp + (size_t)y * p.stride + 4 * q2);
                    c[y][q2 + 0] = v.x; c[y][q2 + 1] = v.y;
                }
            }
        }
    }

    // ---- stage the strip's search window: GRAN-byte granules, coalesced along rows.  A lane keeps ONE granule column and walks
    // the rows LR at a time: its source pointer advances by a row pitch, its LDS destination is base + immediate -- no
    // per-granule division, no per-granule address arithmetic (the flat index / TILE_WG form cost ~20 VALU operations per
    // step, 140-160 per strip: 2 % of a strip that is otherwise packed SADs; profiles/r05/sad_colkeys.txt)
    {
        constexpr int LR = 64 / C::TILE_WG;                            // window rows per step (TILE_WG <= 64 lanes)
        constexpr int STEPS = (C::TILE_H + LR - 1) / LR;
        const int tx0 = bx0 * B - R, ty0 = y0 - R;
        const int sr = lane / C::TILE_WG, sc = lane - sr * C::TILE_WG;
        const int gx = tx0 + C::GRAN * sc;
        const bool lane_on = sr < LR && gx >= 0 && gx < p.W;
        const uint8_t* src = prev + (ptrdiff_t)(ty0 + sr) * p.stride + gx;       // dereferenced only for rows inside the frame
        uint32_t* dst = tile + sr * C::SW + C::GDW * sc;
#pragma unroll
        for (int it = 0; it < STEPS; ++it) {
            const int row = sr + it * LR;
            const bool row_in_tile = (it + 1) * LR <= C::TILE_H || row < C::TILE_H;
            if (lane_on && row_in_tile && (unsigned)(ty0 + row) < (unsigned)p.H) {
                const uint8_t* sp = src + (ptrdiff_t)(it * LR) * p.stride;
                uint32_t* dp = dst + it * LR * C::SW;
                if constexpr (C::GRAN == 16) *reinterpret_cast<uint4*>(dp) = *reinterpret_cast<const uint4*>(sp);
                else if constexpr (C::GRAN == 8) *reinterpret_cast<uint2*>(dp) = *reinterpret_cast<const uint2*>(sp);
                else *dp = *reinterpret_cast<const uint32_t*>(sp);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- main pass: every lane, all dy (in SPLIT passes), dx = dx0 .. dx0+3.  The row walk is expanded through
    // template packs: a plain `#pragma unroll` over TILE_H x NCAND steps exceeds hipcc's unroll budget, which
    // leaves the row index dynamic and sends the c[][] block to scratch memory.
    // colk[j] = dx_j^2, or all-ones for a column clipped by the frame.
    const int dx0 = -R + 4 * g;
    // (RANK_KEY: + the rank code of the column, colk for dy <= 0 and colk_pos for dy > 0; see StripCfg)
    uint32_t colk[4], colk_pos[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dx = dx0 + j;
        const int x = bx * B + dx;
        const bool v = blk_on && g < C::NGA && x >= 0 && x + B <= p.W;
        const uint32_t rank = dx0 < 0 ? 3u - j : (uint32_t)j;      // rank of |dx| inside the group (a group never straddles 0)
        colk[j] = v ? ((uint32_t)(dx * dx) << C::KSHIFT) | (C::RANK_KEY ? rank : 0u) : 0xFFFFFFFFu;
        colk_pos[j] = v ? ((uint32_t)(dx * dx) << C::KSHIFT) | (7u - rank) : 0xFFFFFFFFu;
    }
    uint32_t bkey = 0xFFFFFFFFu;
    int bi = 0;
    uint32_t colmin[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t* amin = nullptr;
    if constexpr (kSadColKeys == 2 && (C::RANK_KEY || C::COMPACT_KEY)) {
        amin = tiles + kStripWaves * C::TILE_DWORDS + wave * kColSlotDwords + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) amin[64 * j] = 0xFFFFFFFFu;
    }
    strip_passes<B, R>(c, tiles, (uint32_t)(wave * C::TILE_DWORDS + b * C::BW + g), colk, colk_pos, y0, p.H, bkey, bi, colmin, amin,
                       std::make_integer_sequence<int, C::SPLIT>{});
    if constexpr (kSadColKeys != 0 && (C::RANK_KEY || C::COMPACT_KEY)) {
        // column minima -> the lane key of the product layout (SAD << 16 | d2 << KSHIFT | code / dy index), once per strip
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t kc;
            if constexpr (kSadColKeys == 2) kc = __hip_atomic_load(amin + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else kc = colmin[j];
            uint32_t lk;
            if constexpr (C::RANK_KEY) {
                // low = dy^2 << 1 | (dy > 0)  ->  (dx^2 + dy^2) << 3 | code, code = rank for dy <= 0, 7 - rank = rank ^ 7 for dy > 0
                const uint32_t x = (((kc & 0xFFFEu) << 2) + colk[j]) ^ ((kc & 1u) * 7u);
                lk = (kc & 0xFFFF0000u) | x;
            } else {
                lk = (kc & 0xFFFF0000u) | ((kc & 0xFFFFu) + colk[j]);
            }
            lk = (colk[j] == 0xFFFFFFFFu || kc == 0xFFFFFFFFu) ? 0xFFFFFFFFu : lk;
            bkey = min(bkey, lk);
        }
    }
    // decode (d2, dy) -> dx; build the cross-lane key (SAD, d2, dy, dx)
    unsigned long long best = ~0ull;
    if (bkey != 0xFFFFFFFFu) {
        const uint32_t d2 = (bkey & 0xFFFFu) >> C::KSHIFT;
        int dx = dx0, dy;
        if constexpr (C::RANK_KEY) {
            const int code = (int)(bkey & 7u);
            const bool neg = code < 4;
            const int rank = neg ? code : 7 - code;
            dx = dx0 + (dx0 < 0 ? 3 - rank : rank);
            const int ady = (int)(__builtin_sqrtf((float)((int)d2 - dx * dx)) + 0.5f);     // exact: a perfect square below 2^12
            dy = neg ? -ady : ady;
        } else {
            if constexpr (C::COMPACT_KEY) bi = (int)(bkey & 63u);
            dy = -R + bi;
            const int dxsq = (int)d2 - dy * dy;
#pragma unroll
            for (int j = 1; j < 4; ++j) dx = ((dx0 + j) * (dx0 + j) == dxsq) ? dx0 + j : dx;
        }
        // cross-lane key in the spec's layout: (SAD << 16 | d2) << 32 | (dy+R) << 8 | (dx+R)
        best = ((unsigned long long)((bkey & 0xFFFF0000u) | d2) << 32) | (unsigned long long)(uint32_t)(((dy + R) << 8) | (dx + R));
    }
    // ---- dx = +R pass (after the main scan, so acc[] is dead and the wave stays within 168 VGPRs):
    // lane = (block b, dy chunk g): KE consecutive dy, one v_sad_u8 per dword
    uint32_t eacc[C::KE];
#pragma unroll
    for (int i = 0; i < C::KE; ++i) eacc[i] = 0;
    {
        const uint32_t* ecol = tile + b * C::BW + C::EDGE_COL;
        const int erow0 = g * C::KE;
        // one-row software prefetch; the empty asm with a memory clobber keeps hipcc from hoisting all
        // B+KE-1 rows of LDS reads above the loop (80 extra live VGPRs next to acc[] and c[][])
        uint32_t nxt[C::BW];
#pragma unroll
        for (int q = 0; q < C::BW; ++q) nxt[q] = ecol[erow0 * C::SW + q];
#pragma unroll
        for (int rr = 0; rr < B + C::KE - 1; ++rr) {
            uint32_t ref[C::BW];
#pragma unroll
            for (int q = 0; q < C::BW; ++q) ref[q] = nxt[q];
            if (rr + 1 < B + C::KE - 1) {
                // the last chunks overshoot the window by up to KE*NG - NCAND rows (those dy are masked): the tile has the rows
                const int row = erow0 + rr + 1;
#pragma unroll
                for (int q = 0; q < C::BW; ++q) nxt[q] = ecol[row * C::SW + q];
            }
#pragma unroll
            for (int i = 0; i < C::KE; ++i) {
                const int y = rr - i;
                if (y >= 0 && y < B) {
#pragma unroll
                    for (int q = 0; q < C::BW; ++q) eacc[i] = __builtin_amdgcn_sad_u8(ref[q], c[y][q], eacc[i]);
                }
            }
            asm volatile("" : "+v"(eacc[0]) : : "memory");
        }
    }

    // ---- candidates of the dx = +R pass (ascending dy, strict <)
    {
        const bool xok = blk_on && bx * B + R + B <= p.W;
#pragma unroll
        for (int i = 0; i < C::KE; ++i) {
            const int dy = -R + g * C::KE + i;
            const bool v = xok && dy <= R && y0 + dy >= 0 && y0 + dy + B <= p.H;
            const unsigned long long k64 =
                ((unsigned long long)((eacc[i] << 16) | (uint32_t)(R * R + dy * dy)) << 32) |
                (unsigned long long)(uint32_t)(((dy + R) << 8) | (2 * R));
            best = (v && k64 < best) ? k64 : best;
        }
    }
    // ---- min over the NG lanes of each block (aligned power-of-two segments: xor butterfly)
#pragma unroll
    for (int m = 1; m < C::NG; m <<= 1) {
        const unsigned long long o = shfl_xor_u64(best, m);
        best = o < best ? o : best;
    }
    if (blk_on && g == 0) {
        const int sad = (int)(best >> 48);
        const int dy = (int)((best >> 8) & 0xFF) - R, dx = (int)(best & 0xFF) - R;
        const size_t k = ((size_t)pair * p.nby + by) * p.nbx + bx;
        float4 e;
        e.x = (float)(bx * B + B / 2 + dx) * p.nx;
        e.y = (float)(y0 + B / 2 + dy) * p.ny;
        e.z = ((float)dx / 1.0f) * (-p.nx);
        e.w = ((float)dy / 1.0f) * (-p.ny);
        p.out_entries[k] = e;
        if (p.out_best) {
            p.out_best[3 * k + 0] = dx;
            p.out_best[3 * k + 1] = dy;
            p.out_best[3 * k + 2] = sad;
        }
    }
}

template <int B, int R>
__global__ __launch_bounds__(64 * kStripWaves, (StripCfg<B, R>::MIN_WAVES)) void sad_strip_kernel(const SadParams p, int strips_per_row,
                                                                      int total_strips) {
    using C = StripCfg<B, R>;
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kStripWaves * (C::TILE_DWORDS + kColSlotDwords)];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // XCD-aware remap: workgroup w lands on XCD w % 8; give each XCD a contiguous run of strips
    // (the grid is padded to a multiple of 8 workgroups so the remap is a bijection)
    const int per_xcd = gridDim.x / 8;
    const int lwg = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    const int strip = lwg * kStripWaves + wave;
    if (strip >= total_strips) return;
    strip_body<B, R>(p, strips_per_row, strip, tiles, wave, lane);
}

// The same search over an indirect list (overflow strips of the pruned mode): strip_list[0] = count, ids follow.  A short
// grid walks the list with a stride (the count is only known on the device, and a full grid of workgroups that find
// nothing to do costs ~3 us per thousand); hardware order, so consecutive entries spread over the XCDs.
template <int B, int R>
__global__ __launch_bounds__(64 * kStripWaves, (StripCfg<B, R>::MIN_WAVES)) void sad_strip_list_kernel(const SadParams p, int strips_per_row,
                                                                           const uint32_t* __restrict__ strip_list) {
    using C = StripCfg<B, R>;
    __shared__ __attribute__((aligned(16))) uint32_t tiles[kStripWaves * (C::TILE_DWORDS + kColSlotDwords)];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int count = (int)strip_list[0];
#pragma unroll 1
    for (int k = (int)blockIdx.x * kStripWaves + wave; k < count; k += (int)gridDim.x * kStripWaves)
        strip_body<B, R>(p, strips_per_row, (int)strip_list[1 + k], tiles, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// sad_pde_kernel: exact search with partial-distortion elimination (opt-in mode OFPS_HIP_SAD_PRUNED, B16 R16).
//
// The SAD over a subset of the block's rows is a lower bound of the SAD over all of them.  The kernel
//   1. walks the strip exactly like sad_strip_kernel but with block rows 0, 4, 8, 12 only: the partial SAD of EVERY
//      candidate for a quarter of the SAD-unit time (the 33 x 4 packed accumulators stay in registers);
//   2. takes each block's minimum-partial candidate c0 and evaluates its full SAD U (8 lanes x 2 rows, v_sad_u8 on
//      byte-aligned window data);
//   3. lists the candidates whose partial SAD does not exceed U -- every other candidate has SAD >= partial > U and
//      cannot win -- in a per-wave LDS list, and evaluates them exactly, 8 per wave step, folding the 64-bit key
//      (SAD, d2, dy, dx) into the block's minimum with an LDS atomic;
//   4. a strip whose list overflows (content with nothing to prune: noise, flat frames, blocks that straddle a motion
//      discontinuity) is handed to sad_strip_kernel through strip_list.
// The winner is the minimum of the same total-order key over a candidate set that provably contains the exhaustive
// winner: identical output, bit for bit, on any content; only the run time depends on the content (DESIGN.md
// section 3).  It replaces the successive-elimination kernel of the first version, whose bound (sub-block sums) was
// both weaker and costlier than a quarter of the SADs themselves.
struct PdeCfg {
    static constexpr int B = 16, R = 16, STEP = 4, PR = B / STEP;
    using C = StripCfg<B, R>;
    static constexpr int CUR_DW = B * C::NB * C::BW;          // the strip of the current frame: 16 rows x 32 dwords
    static constexpr int CAP = 512;                           // survivors per strip before falling back
    static constexpr int WAVE_DW = C::TILE_DWORDS + 4 + CUR_DW + CAP + 2 * C::NB;       // +4: the dx=+R read of the last row
    static constexpr int OFF_CUR = C::TILE_DWORDS + 4, OFF_LIST = OFF_CUR + CUR_DW, OFF_BEST = OFF_LIST + CAP;
    static_assert(OFF_BEST % 2 == 0, "64-bit keys must be 8-byte aligned");
};

template <int RR>
__device__ __forceinline__ void pde_row(unsigned long long (&acc)[PdeCfg::C::NP], const uint32_t (&c)[PdeCfg::PR][PdeCfg::C::BW],
                                        const uint32_t* lds, uint32_t& base) {
    using C = PdeCfg::C;
    constexpr int B = PdeCfg::B, R = PdeCfg::R, STEP = PdeCfg::STEP;
    constexpr int G = StripWalk<B, R>::ROWS_PER_BASE;
    if constexpr (RR > 0 && RR % G == 0) {
        base += G * C::SW;
        asm volatile("" : "+v"(base));
    }
    unsigned long long win[C::BW];
#pragma unroll
    for (int q = 0; q < C::BW; ++q) win[q] = reinterpret_cast<const U64A4*>(lds + base + (RR % G) * C::SW + q)->v;
#pragma unroll
    for (int ii = 0; ii < C::NP; ++ii) {
        const int y = RR - ii;                                  // window row RR belongs to candidate ii, block row y
        if (ii < C::NCAND && y >= 0 && y < B && y % STEP == 0) {
#pragma unroll
            for (int q = 0; q < C::BW; ++q) acc[ii] = __builtin_amdgcn_qsad_pk_u16_u8(win[q], c[y / STEP][q], acc[ii]);
        }
    }
}
template <int... RR>
__device__ __forceinline__ void pde_rows(unsigned long long (&acc)[PdeCfg::C::NP], const uint32_t (&c)[PdeCfg::PR][PdeCfg::C::BW],
                                         const uint32_t* lds, uint32_t tile_off, std::integer_sequence<int, RR...>) {
    uint32_t base = tile_off;
    asm volatile("" : "+v"(base));
    (pde_row<RR>(acc, c, lds, base), ...);
}

// rows 2*r8 and 2*r8+1 of candidate (dyi, dxi) of block bb: 8 lanes cover a block
__device__ __forceinline__ uint32_t pde_two_rows(const uint32_t* tile, const uint32_t* curl, int bb, int dyi, int dxi, int r8) {
    using C = PdeCfg::C;
    const int sh = dxi & 3;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int y = 2 * r8 + k;
        const uint32_t* rp = tile + (dyi + y) * C::SW + bb * C::BW + (dxi >> 2);
        const uint32_t* cp = curl + y * (C::NB * C::BW) + bb * C::BW;
        uint32_t d[C::BW + 1];
#pragma unroll
        for (int q = 0; q <= C::BW; ++q) d[q] = rp[q];
#pragma unroll
        for (int q = 0; q < C::BW; ++q) s = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(d[q + 1], d[q], sh), cp[q], s);
    }
    return s;
}

__global__ __launch_bounds__(64 * kStripWaves, 3) void sad_pde_kernel(const SadParams p, int strips_per_row, int total_strips,
                                                                       uint32_t* __restrict__ strip_list) {
    using P = PdeCfg;
    using C = P::C;
    constexpr int B = P::B, R = P::R;
    __shared__ __attribute__((aligned(16))) uint32_t lds[kStripWaves * P::WAVE_DW];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int per_xcd = gridDim.x / 8;                        // XCD-aware remap, see sad_strip_kernel
    const int lwg = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    const int strip = lwg * kStripWaves + wave;
    if (strip >= total_strips) return;
    const int strips_per_pair = strips_per_row * p.nby;
    const int pair = strip / strips_per_pair;
    const int rem = strip - pair * strips_per_pair;
    const int by = rem / strips_per_row;
    const int bx0 = (rem - by * strips_per_row) * C::NB;
    const uint8_t* __restrict__ prev = p.prev_base + (size_t)pair * p.prev_pitch;
    const uint8_t* __restrict__ cur = p.cur_base + (size_t)pair * p.cur_pitch;
    uint32_t* tile = lds + wave * P::WAVE_DW;
    uint32_t* curl = tile + P::OFF_CUR;
    uint32_t* list = tile + P::OFF_LIST;
    unsigned long long* bestk = reinterpret_cast<unsigned long long*>(tile + P::OFF_BEST);

    const int b = lane / C::NG, g = lane % C::NG;
    const int bx = bx0 + b;
    const bool blk_on = bx < p.nbx;
    const int y0 = by * B;

    // ---- stage the search window (16-byte granules) and the strip of the current frame
    {
        const int tx0 = bx0 * B - R, ty0 = y0 - R;
        for (int idx = lane; idx < C::TILE_H * C::TILE_WG; idx += 64) {
            const int row = idx / C::TILE_WG, col = idx - row * C::TILE_WG;
            const int gx = tx0 + 16 * col, gy = ty0 + row;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                *reinterpret_cast<uint4*>(tile + row * C::SW + 4 * col) = *reinterpret_cast<const uint4*>(prev + (size_t)gy * p.stride + gx);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = lane + 64 * k, row = idx / C::NB, blk = idx % C::NB;
            const int bxc = min(bx0 + blk, p.nbx - 1);
            *reinterpret_cast<uint4*>(curl + row * (C::NB * C::BW) + blk * C::BW) =
                *reinterpret_cast<const uint4*>(cur + (size_t)(y0 + row) * p.stride + bxc * B);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- 1. partial SADs (rows 0, 4, 8, 12) of every candidate of the main pass
    uint32_t c4[P::PR][C::BW];
#pragma unroll
    for (int r = 0; r < P::PR; ++r)
#pragma unroll
        for (int q = 0; q < C::BW; ++q) c4[r][q] = curl[(P::STEP * r) * (C::NB * C::BW) + b * C::BW + q];
    unsigned long long acc[C::NP];
#pragma unroll
    for (int ii = 0; ii < C::NP; ++ii) acc[ii] = 0;
    pde_rows(acc, c4, lds, (uint32_t)(wave * P::WAVE_DW + b * C::BW + g), std::make_integer_sequence<int, C::NCAND + B - 1>{});
#pragma unroll
    for (int ii = 0; ii < C::NP; ++ii) asm volatile("" : "+v"(acc[ii]));

    const int dx0 = -R + 4 * g;
    bool colv[4];
    uint32_t colk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = bx * B + dx0 + j;
        colv[j] = blk_on && x >= 0 && x + B <= p.W;
        colk[j] = colv[j] ? (uint32_t)((dx0 + j) * (dx0 + j)) << C::KSHIFT : 0xFFFFFFFFu;
    }
    static_assert(C::COMPACT_KEY && C::SPLIT == 1, "written for the +-16 geometry");
    uint32_t bkey = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < C::NCAND; ++i) {
        const int dy = -R + i;
        if (y0 + dy >= 0 && y0 + dy + B <= p.H) {                 // uniform over the strip
            const uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
            const uint32_t k0 = (lo << 16) | colk[0], k1 = (lo & 0xFFFF0000u) | colk[1];
            const uint32_t k2 = (hi << 16) | colk[2], k3 = (hi & 0xFFFF0000u) | colk[3];
            bkey = min(bkey, __builtin_elementwise_add_sat(min(min(k0, k1), min(k2, k3)), (uint32_t)((dy * dy) << 6 | i)));
        }
    }
    unsigned long long pbest = ~0ull;                             // (partial, d2, dy, dx): only used to pick c0
    if (bkey != 0xFFFFFFFFu) {
        const int bi = (int)(bkey & 63u), dy = -R + bi;
        const uint32_t d2 = (bkey & 0xFFFFu) >> C::KSHIFT;
        const int dxsq = (int)d2 - dy * dy;
        int dx = dx0;
#pragma unroll
        for (int j = 1; j < 4; ++j) dx = ((dx0 + j) * (dx0 + j) == dxsq) ? dx0 + j : dx;
        pbest = ((unsigned long long)((bkey & 0xFFFF0000u) | d2) << 32) | (unsigned long long)(uint32_t)(((dy + R) << 8) | (dx + R));
    }
    // ---- the dx = +R column: lane = (block, chunk of KE dy), partial rows only
    uint32_t eacc[C::KE];
    const bool xok = blk_on && bx * B + R + B <= p.W;
    {
        const uint32_t* ecol = tile + b * C::BW + C::EDGE_COL;
#pragma unroll
        for (int i = 0; i < C::KE; ++i) {
            const int dyi = min(g * C::KE + i, C::NCAND - 1);
            uint32_t sacc = 0;
#pragma unroll
            for (int r = 0; r < P::PR; ++r)
#pragma unroll
                for (int q = 0; q < C::BW; ++q) sacc = __builtin_amdgcn_sad_u8(ecol[(dyi + P::STEP * r) * C::SW + q], c4[r][q], sacc);
            eacc[i] = sacc;
            const int dy = -R + g * C::KE + i;
            const bool v = xok && dy <= R && y0 + dy >= 0 && y0 + dy + B <= p.H;
            const unsigned long long k64 = ((unsigned long long)((sacc << 16) | (uint32_t)(R * R + dy * dy)) << 32) |
                                           (unsigned long long)(uint32_t)(((dy + R) << 8) | (2 * R));
            pbest = (v && k64 < pbest) ? k64 : pbest;
        }
    }
#pragma unroll
    for (int m = 1; m < C::NG; m <<= 1) {
        const unsigned long long o = shfl_xor_u64(pbest, m);
        pbest = o < pbest ? o : pbest;
    }
    // ---- 2. c0 = the block's minimum-partial candidate ((0,0) is always valid for a full block) and its exact SAD
    const int dy0i = blk_on ? (int)((pbest >> 8) & 0xFF) : R, dx0i = blk_on ? (int)(pbest & 0xFF) : R;
    auto key64 = [&](uint32_t sad, int dyi, int dxi) {
        const int dy = dyi - R, dx = dxi - R;
        return ((unsigned long long)((sad << 16) | (uint32_t)(dx * dx + dy * dy)) << 32) | (unsigned long long)(uint32_t)((dyi << 8) | dxi);
    };
    uint32_t U = pde_two_rows(tile, curl, b, dy0i, dx0i, g);
#pragma unroll
    for (int m = 1; m < C::NG; m <<= 1) U += __shfl_xor(U, m, 64);
    if (g == 0) bestk[b] = blk_on ? key64(U, dy0i, dx0i) : ~0ull;
    // ---- 3. survivors: partial <= U.  One wave-wide ballot per (dy, column): no hit (the common case) costs a compare
    // and a uniform branch; hits are appended at ballot-prefix positions, so the list needs no atomics.  c0 itself
    // qualifies (its partial cannot exceed its SAD) and is simply evaluated a second time.
    int nlist = 0;                                                // wave-uniform
    {
        int uj[4];                                                // per-column threshold; -1 rejects a clipped column
#pragma unroll
        for (int j = 0; j < 4; ++j) uj[j] = colv[j] ? (int)U : -1;
        const unsigned long long below = (1ull << lane) - 1ull;
        // fast reject of a whole dy (4 columns, all 64 lanes) with packed-u16 minima: a half survives min(., U+1)
        // unchanged-from-U+1 exactly when it exceeds U
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        const uint32_t u1 = min(U + 1u, 0xFFFFu) * 0x10001u;
        auto any_le = [&](uint32_t v) {
            const us2 m = __builtin_elementwise_min(__builtin_bit_cast(us2, v), __builtin_bit_cast(us2, u1));
            return __builtin_bit_cast(uint32_t, m) ^ u1;
        };
#pragma unroll
        for (int i = 0; i < C::NCAND; ++i) {
            const int dy = -R + i;
            if (y0 + dy >= 0 && y0 + dy + B <= p.H) {
                const uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
                if (!__ballot((any_le(lo) | any_le(hi)) != 0u)) continue;
                const int s4[4] = {(int)(lo & 0xFFFFu), (int)(lo >> 16), (int)(hi & 0xFFFFu), (int)(hi >> 16)};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool hit = s4[j] <= uj[j];
                    const unsigned long long mask = __ballot(hit);
                    if (mask) {
                        const int pos = nlist + __popcll(mask & below);
                        if (hit && pos < P::CAP) list[pos] = ((uint32_t)b << 16) | ((uint32_t)i << 8) | (uint32_t)(4 * g + j);
                        nlist += __popcll(mask);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < C::KE; ++i) {
            const int dyi = g * C::KE + i, dy = dyi - R;
            const bool hit = xok && dy <= R && y0 + dy >= 0 && y0 + dy + B <= p.H && eacc[i] <= U;
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                const int pos = nlist + __popcll(mask & below);
                if (hit && pos < P::CAP) list[pos] = ((uint32_t)b << 16) | ((uint32_t)dyi << 8) | (uint32_t)(2 * R);
                nlist += __popcll(mask);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = nlist;
    if (n > P::CAP) {                                             // nothing to prune here: the exhaustive kernel takes the strip
        if (lane == 0) {
            const uint32_t pos = atomicAdd(&strip_list[0], 1u);
            strip_list[1 + pos] = (uint32_t)strip;
        }
        return;
    }
    // ---- 4. exact SAD of the survivors: 8 lanes x 2 rows per candidate, two independent candidates per lane group and
    // step (the chain list -> window rows -> SAD -> 3 cross-lane adds -> LDS atomic is latency-bound on its own)
    constexpr int PER_STEP = 2 * (64 / C::NG);
    for (int base = 0; base < n; base += PER_STEP) {
        const int idx0 = base + lane / C::NG, idx1 = idx0 + 64 / C::NG;
        const uint32_t e0 = list[min(idx0, n - 1)], e1 = list[min(idx1, n - 1)];
        const int bb0 = (int)(e0 >> 16), dyi0 = (int)((e0 >> 8) & 0xFF), dxi0 = (int)(e0 & 0xFF);
        const int bb1 = (int)(e1 >> 16), dyi1 = (int)((e1 >> 8) & 0xFF), dxi1 = (int)(e1 & 0xFF);
        uint32_t s0 = pde_two_rows(tile, curl, bb0, dyi0, dxi0, g);
        uint32_t s1 = pde_two_rows(tile, curl, bb1, dyi1, dxi1, g);
#pragma unroll
        for (int m = 1; m < C::NG; m <<= 1) { s0 += __shfl_xor(s0, m, 64); s1 += __shfl_xor(s1, m, 64); }
        if (g == 0 && idx0 < n) atomicMin(&bestk[bb0], key64(s0, dyi0, dxi0));
        if (g == 0 && idx1 < n) atomicMin(&bestk[bb1], key64(s1, dyi1, dxi1));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (blk_on && g == 0) write_block_result(p, pair, bx, by, B, R, bestk[b]);
}

// Generic kernel for block/range pairs the packed-SAD kernel does not cover: one wave per block,
// lanes over candidates, bytes straight from global memory (L1/L2 absorb the reuse).  Slow path.
__global__ __launch_bounds__(64) void sad_generic_kernel(const SadParams p, int B, int R) {
    const int pair = blockIdx.z, by = blockIdx.y, bx = blockIdx.x;
    const uint8_t* __restrict__ prev = p.prev_base + (size_t)pair * p.prev_pitch;
    const uint8_t* __restrict__ cur = p.cur_base + (size_t)pair * p.cur_pitch;
    const int x0 = bx * B, y0 = by * B, n = 2 * R + 1;
    unsigned long long best = ~0ull;
    for (int cand = threadIdx.x; cand < n * n; cand += 64) {
        const int dy = cand / n - R, dx = cand % n - R;
        if (x0 + dx < 0 || x0 + dx + B > p.W || y0 + dy < 0 || y0 + dy + B > p.H) continue;
        uint32_t sad = 0;
        for (int y = 0; y < B; ++y) {
            const uint8_t* cr = cur + (size_t)(y0 + y) * p.stride + x0;
            const uint8_t* pr = prev + (size_t)(y0 + dy + y) * p.stride + x0 + dx;
            for (int x = 0; x < B; ++x) {
                const int d = (int)cr[x] - (int)pr[x];
                sad += (uint32_t)(d < 0 ? -d : d);
            }
        }
        // same total order as the packed kernel; SAD can exceed 16 bits here, so use a wider layout
        const unsigned long long key = ((unsigned long long)sad << 32) | ((unsigned long long)(dx * dx + dy * dy) << 16) |
                                       (unsigned long long)(((dy + R) << 8) | (dx + R));
        best = key < best ? key : best;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = shfl_xor_u64(best, m);
        best = o < best ? o : best;
    }
    if (threadIdx.x == 0) {
        const int sad = (int)(best >> 32);
        const int dy = (int)((best >> 8) & 0xFF) - R, dx = (int)(best & 0xFF) - R;
        const size_t k = ((size_t)pair * p.nby + by) * p.nbx + bx;
        float4 e;
        e.x = (float)(x0 + B / 2 + dx) * p.nx;
        e.y = (float)(y0 + B / 2 + dy) * p.ny;
        e.z = ((float)dx / 1.0f) * (-p.nx);
        e.w = ((float)dy / 1.0f) * (-p.ny);
        p.out_entries[k] = e;
        if (p.out_best) {
            p.out_best[3 * k + 0] = dx;
            p.out_best[3 * k + 1] = dy;
            p.out_best[3 * k + 2] = sad;
        }
    }
}

template <int B, int R, int K>
void launch_qsad(const SadParams& p, int pairs, hipStream_t s) {
    dim3 grid((p.nbx + kWavesPerWG - 1) / kWavesPerWG, p.nby, pairs);
    hipLaunchKernelGGL((sad_qsad_kernel<B, R, K>), grid, dim3(256), 0, s, p);
}

// div48's preconditions (n * d < 2^48, n / d < 2^16) are sad_pairs_device's `strip_ok`: < 2^28 blocks per batch (a 4K 8x8 batch of
// 2,000 pairs), < 2^20 blocks per frame, <= 65,535 pairs; anything larger takes the per-block kernel
inline void strip_div_magic(SadParams& q, int strips_per_row) {
    q.div_spp = div_magic48((unsigned)strips_per_row * (unsigned)q.nby);
    q.div_spr = div_magic48((unsigned)strips_per_row);
}

template <int B, int R>
void launch_strip(const SadParams& p, int pairs, hipStream_t s) {
    using C = StripCfg<B, R>;
    const int strips_per_row = (p.nbx + C::NB - 1) / C::NB;
    const int total = strips_per_row * p.nby * pairs;
    const int nwg = ((total + kStripWaves - 1) / kStripWaves + 7) / 8 * 8;     // multiple of 8: see the XCD remap
    SadParams q = p;
    strip_div_magic(q, strips_per_row);
    hipLaunchKernelGGL((sad_strip_kernel<B, R>), dim3(nwg), dim3(64 * kStripWaves), 0, s, q, strips_per_row, total);
}

// pruned search: sad_pde_kernel over every strip, then the exhaustive kernel over the overflow strips
int launch_pde_16_16(ofps_hip_ctx* ctx, const SadParams& p, int pairs, hipStream_t s) {
    using C = StripCfg<16, 16>;
    const int strips_per_row = (p.nbx + C::NB - 1) / C::NB;
    const int total = strips_per_row * p.nby * pairs;
    auto* strip_list = static_cast<uint32_t*>(ofps::scratch(ctx, ofps::S_SAD_LIST, (size_t)(total + 1) * sizeof(uint32_t)));
    if (!strip_list) return OFPS_HIP_ENOMEM;
    OFPS_HIP_TRY(ctx, hipMemsetAsync(strip_list, 0, sizeof(uint32_t), s));
    const int nwg = ((total + kStripWaves - 1) / kStripWaves + 7) / 8 * 8;     // multiple of 8: see the XCD remap
    hipLaunchKernelGGL(sad_pde_kernel, dim3(nwg), dim3(64 * kStripWaves), 0, s, p, strips_per_row, total, strip_list);
    const int nwg_ind = nwg < 8 * ctx->num_cus ? nwg : 8 * ctx->num_cus;      // grid-stride walk of the overflow list
    SadParams q = p;
    strip_div_magic(q, strips_per_row);
    hipLaunchKernelGGL((sad_strip_list_kernel<16, 16>), dim3(nwg_ind), dim3(64 * kStripWaves), 0, s, q, strips_per_row,
                       (const uint32_t*)strip_list);
    return OFPS_HIP_OK;
}

}  // namespace

extern "C" {

size_t ofps_hip_sad_block_count(int W, int H, int block) {
    if (W <= 0 || H <= 0 || block <= 0) return 0;
    return (size_t)(W / block) * (size_t)(H / block);
}

}  // extern "C" (reopened below)

namespace ofps {
// Shared by the batched entry point and the per-frame pipeline (pipeline.hip): `pairs` searches,
// pair k between prev_base + k*prev_pitch and cur_base + k*cur_pitch.
int sad_pairs_device(ofps_hip_ctx* ctx, const uint8_t* prev_base, size_t prev_pitch, const uint8_t* cur_base,
                     size_t cur_pitch, int pairs, int W, int H, int stride, int block, int range, void* d_out_entries,
                     void* d_out_best) {
    OFPS_REQUIRE(ctx, W > 0 && H > 0 && stride >= W, "sad_flow: bad geometry W=%d H=%d stride=%d", W, H, stride);
    OFPS_REQUIRE(ctx, stride % 4 == 0 && ((uintptr_t)prev_base % 4) == 0 && ((uintptr_t)cur_base % 4) == 0 &&
                          prev_pitch % 4 == 0 && cur_pitch % 4 == 0,
                 "sad_flow: rows must be 4-byte aligned (stride=%d)", stride);
    OFPS_REQUIRE(ctx, block >= 1 && block <= 64 && range >= 0 && range <= 64,
                 "sad_flow: block=%d range=%d outside [1,64]/[0,64]", block, range);
    SadParams p{};
    p.prev_base = prev_base; p.cur_base = cur_base;
    p.prev_pitch = prev_pitch; p.cur_pitch = cur_pitch;
    p.W = W; p.H = H; p.stride = stride;
    p.nbx = W / block; p.nby = H / block;
    p.nx = 1.0f / (float)W; p.ny = 1.0f / (float)H;
    p.out_entries = static_cast<float4*>(d_out_entries);
    p.out_best = static_cast<int*>(d_out_best);
    if (p.nbx == 0 || p.nby == 0 || pairs <= 0) return OFPS_HIP_OK;
    OFPS_REQUIRE(ctx, pairs <= 65535 && p.nby <= 65535, "sad_flow: grid too large");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int key = block * 1000 + range;
    const bool force_block = ctx->opt.sad_force_block != 0;     // OFPS_HIP_SAD_KERNEL=block (A/B profiling)
    const bool strip_ok = !force_block && stride % 16 == 0 && ((uintptr_t)prev_base % 16) == 0 &&
                          ((uintptr_t)cur_base % 16) == 0 && prev_pitch % 16 == 0 && cur_pitch % 16 == 0 &&
                          (long long)p.nbx * p.nby * pairs < (1ll << 28) && (long long)p.nbx * p.nby < (1ll << 20);
    switch (key) {
        // strip kernels need 16-byte aligned rows; otherwise (or with OFPS_HIP_SAD_KERNEL=block, A/B
        // profiling only) the per-block kernel handles the pair.
        case 16016:
            if (strip_ok && ctx->sad_mode == OFPS_HIP_SAD_PRUNED) {
                const int rc = launch_pde_16_16(ctx, p, pairs, s);
                if (rc != OFPS_HIP_OK) return rc;
            } else if (strip_ok) launch_strip<16, 16>(p, pairs, s);
            else launch_qsad<16, 16, 5>(p, pairs, s);
            break;
        case 16008:
            if (strip_ok) launch_strip<16, 8>(p, pairs, s); else launch_qsad<16, 8, 3>(p, pairs, s);
            break;
        case 16032:
            if (strip_ok) launch_strip<16, 32>(p, pairs, s); else launch_qsad<16, 32, 5>(p, pairs, s);
            break;
        case 8032:
            if (strip_ok) launch_strip<8, 32>(p, pairs, s); else launch_qsad<8, 32, 5>(p, pairs, s);
            break;
        case 8016:
            if (strip_ok) launch_strip<8, 16>(p, pairs, s); else launch_qsad<8, 16, 5>(p, pairs, s);
            break;
        case 8008:
            if (strip_ok) launch_strip<8, 8>(p, pairs, s); else launch_qsad<8, 8, 3>(p, pairs, s);
            break;
        // the other multiples of 4 inside the plugins' "Search range" property (8..32): strip kernel with idle lanes;
        // unaligned rows fall through to the generic kernel
#define OFPS_STRIP_CASE(BB, RR)                                         \
        case BB * 1000 + RR:                                            \
            if (strip_ok) { launch_strip<BB, RR>(p, pairs, s); break; } \
            [[fallthrough]];
        OFPS_STRIP_CASE(16, 12) OFPS_STRIP_CASE(16, 20) OFPS_STRIP_CASE(16, 24) OFPS_STRIP_CASE(16, 28)
        OFPS_STRIP_CASE(8, 12) OFPS_STRIP_CASE(8, 20) OFPS_STRIP_CASE(8, 24) OFPS_STRIP_CASE(8, 28)
#undef OFPS_STRIP_CASE
        default: {
            dim3 grid(p.nbx, p.nby, pairs);
            hipLaunchKernelGGL(sad_generic_kernel, grid, dim3(64), 0, s, p, block, range);
        }
    }
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}
}  // namespace ofps

extern "C" {

int ofps_hip_sad_flow_dev(ofps_hip_ctx* ctx, const void* d_frames, int n_frames, int W, int H, int stride,
                          size_t frame_pitch, int ref_mode, int block, int range, void* d_out_entries,
                          void* d_out_best) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_frames && d_out_entries, "sad_flow: null device pointer");
    OFPS_REQUIRE(ctx, n_frames >= 2, "sad_flow: need at least 2 frames (got %d)", n_frames);
    OFPS_REQUIRE(ctx, frame_pitch >= (size_t)stride * (size_t)H, "sad_flow: frame_pitch smaller than a frame");
    OFPS_REQUIRE(ctx, ref_mode == 0 || ref_mode == 1, "sad_flow: ref_mode must be 0 or 1");
    const auto* f = static_cast<const uint8_t*>(d_frames);
    return ofps::sad_pairs_device(ctx, f, ref_mode ? 0 : frame_pitch, f + frame_pitch, frame_pitch, n_frames - 1, W, H, stride,
                                  block, range, d_out_entries, d_out_best);
}

int ofps_hip_sad_flow(ofps_hip_ctx* ctx, const uint8_t* prev, const uint8_t* cur, int W, int H, int stride,
                      int block, int range, float* out_entries, int32_t* out_best, size_t* n_out) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, prev && cur && out_entries, "sad_flow: null host pointer");
    OFPS_REQUIRE(ctx, W > 0 && H > 0 && stride >= W && block >= 1, "sad_flow: bad geometry");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // repack to a 64-byte-multiple device stride so any host stride is accepted
    const int dstride = (W + 63) & ~63;
    const size_t pitch = (size_t)dstride * H;
    const size_t nblk = ofps_hip_sad_block_count(W, H, block);
    auto* d_frames = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_FRAMES, 2 * pitch));
    auto* d_ent = static_cast<float*>(ofps::scratch(ctx, ofps::S_ENTRIES, nblk * 4 * sizeof(float)));
    auto* d_best = static_cast<int32_t*>(ofps::scratch(ctx, ofps::S_BEST, nblk * 3 * sizeof(int32_t)));
    if (!d_frames || !d_ent || !d_best) return OFPS_HIP_ENOMEM;
    OFPS_HIP_TRY(ctx, ofps::upload_rows(d_frames, dstride, prev, stride, W, H, ctx->stream));
    OFPS_HIP_TRY(ctx, ofps::upload_rows(d_frames + pitch, dstride, cur, stride, W, H, ctx->stream));
    int rc = ofps_hip_sad_flow_dev(ctx, d_frames, 2, W, H, dstride, pitch, 0, block, range, d_ent, d_best);
    if (rc != OFPS_HIP_OK) return rc;
    if (nblk) {
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_entries, d_ent, nblk * 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        if (out_best)
            OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_best, d_best, nblk * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n_out) *n_out = nblk;
    return OFPS_HIP_OK;
}

int ofps_hip_sad_pruned_overflow_strips(ofps_hip_ctx* ctx, uint32_t* count) {
    if (!ctx || !count) return OFPS_HIP_EINVAL;
    *count = 0;
    auto& sc = ctx->scratch[ofps::S_SAD_LIST];            // its own slot: no other stage writes here between the search and this read
    if (!sc.p) return OFPS_HIP_OK;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    OFPS_HIP_TRY(ctx, hipMemcpy(count, sc.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return OFPS_HIP_OK;
}

int ofps_hip_set_sad_mode(ofps_hip_ctx* ctx, int mode) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, mode == OFPS_HIP_SAD_EXHAUSTIVE || mode == OFPS_HIP_SAD_PRUNED, "set_sad_mode: unknown mode %d", mode);
    ctx->sad_mode = mode;
    return OFPS_HIP_OK;
}

}  // extern "C"
