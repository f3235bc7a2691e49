// densify.hip -- A1-A4: MotionFieldDensifier::add_vector + MotionField::from on gfx950
// (ofps/src/motion_field.rs:133-190, 297-308) and cv-decoder's downsample output stage
// (cv-decoder/src/lib.rs:244-291).
//
// The reference adds vectors to their cell one after another in input order, in f32.  To return
// the same bits (not just the same value to 1e-4) the GPU keeps that order:
//   1. cell_kernel        one thread per entry: nalgebra clamp quirk + round-half-away -> cell id
//   2. stable LSD radix sort of (cell id, entry index), 8-bit digits, 1 pass for <= 256 cells,
//      2 for <= 65536.  Ranks inside a 1024-entry tile come from an 8-ballot "same digit" match
//      per wave plus a 16-slot per-digit prefix in LDS; tile bases from a digit-major scan (one
//      workgroup per digit) plus the prefix over the 256 digit totals.
//   3. bounds_kernel      first/last sorted position of every cell
//   4. cell_sum_kernel    one wave per cell: lanes gather 64 entries at a time into LDS, lanes 0/1
//      then add x / y sequentially in input order; counts replay eps + 1 + 1 + ...;
//      field = sum / count (IEEE divide), exactly `component_div` (motion_field.rs:304).
// Traffic: 16 B/entry read twice (cell pass + gather) plus 8 B/entry/pass of sort keys; the
// kernels are launch/latency-bound at detector sizes (N ~ 1e4) and HBM-bound at per-pixel
// sizes (N ~ 2e6).
#include "common.hpp"

namespace ofps {

constexpr int kTile = 1024;       // entries per sort tile (256 threads x 4 rounds)
constexpr float kF32Eps = 1.1920929e-07f;

// motion_field.rs:164-178 -> (x, y).  `clamp` on a Point2 compares all components at once
// (SURVEY.md A.6); f32::round is half-away-from-zero; `as usize` saturates (NaN/negatives -> 0).
__device__ __forceinline__ void densifier_cell(float px, float py, int w, int h, uint32_t& x, uint32_t& y) {
    float cx, cy;
    if (px > 0.0f && py > 0.0f) {
        if (px < 1.0f && py < 1.0f) { cx = px; cy = py; }
        else { cx = 1.0f; cy = 1.0f; }
    } else { cx = 0.0f; cy = 0.0f; }
    const float fx = roundf(cx * (float)(w - 1)), fy = roundf(cy * (float)(h - 1));
    x = fx > 0.0f ? (uint32_t)fx : 0u;
    y = fy > 0.0f ? (uint32_t)fy : 0u;
}

__global__ __launch_bounds__(256) void cell_kernel(const float4* __restrict__ entries, size_t n, int w, int h,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                   uint32_t* __restrict__ out_cells, uint32_t* __restrict__ begin,
                                                   uint32_t* __restrict__ end, size_t cells) {
    const size_t item = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    // clear the per-cell [begin, end) tables bounds_kernel fills later on this stream (saves two memset launches)
    for (size_t c = i; c < cells; c += (size_t)gridDim.x * 256) { begin[item * cells + c] = 0; end[item * cells + c] = 0; }
    if (i >= n) return;
    const float4 e = entries[item * n + i];
    uint32_t x, y;
    densifier_cell(e.x, e.y, w, h, x, y);
    keys[item * n + i] = y * (uint32_t)w + x;
    vals[item * n + i] = (uint32_t)i;
    if (out_cells) {
        out_cells[2 * (item * n + i)] = x;
        out_cells[2 * (item * n + i) + 1] = y;
    }
}

// per-tile digit histogram -> hist[item][digit][tile]
__global__ __launch_bounds__(256) void sort_count_kernel(const uint32_t* __restrict__ keys, size_t n, int shift,
                                                         int ntiles, uint32_t* __restrict__ hist) {
    __shared__ uint32_t cnt[256];
    const size_t item = blockIdx.y;
    const int tile = blockIdx.x;
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)tile * kTile;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(keys[item * n + i] >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    hist[(item * 256 + threadIdx.x) * ntiles + tile] = cnt[threadIdx.x];
}

// Digit-major scan, one workgroup per (digit, item): exclusive prefix of that digit's counts over the tiles
// (coalesced 256-element chunks, Hillis-Steele in LDS, running carry) + the digit's total.  The scatter
// kernel adds the exclusive prefix over the 256 digit totals itself.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

__global__ __launch_bounds__(256) void sort_scan_kernel(uint32_t* __restrict__ hist, int ntiles,
                                                        uint32_t* __restrict__ digit_total) {
    // 256 tiles per round: inclusive scan inside each wave (shuffles), the four wave totals through LDS -- one barrier
    // per round (the slots alternate) instead of the 16 of a workgroup-wide Hillis-Steele scan
    __shared__ uint32_t wtot[2][4];
    const size_t item = blockIdx.y;
    const int digit = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* h = hist + (item * 256 + digit) * (size_t)ntiles;
    uint32_t carry = 0;                                    // the same value in every thread
    int par = 0;
    for (int c0 = 0; c0 < ntiles; c0 += 256, par ^= 1) {
        const int i = c0 + threadIdx.x;
        const uint32_t v = i < ntiles ? h[i] : 0u;
        const uint32_t incl = wave_incl_scan_u32(v, lane);
        if (lane == 63) wtot[par][wave] = incl;
        __syncthreads();
        const uint32_t w0 = wtot[par][0], w1 = wtot[par][1], w2 = wtot[par][2], w3 = wtot[par][3];
        const uint32_t before = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
        if (i < ntiles) h[i] = carry + before + incl - v;
        carry += w0 + w1 + w2 + w3;
    }
    if (threadIdx.x == 0) digit_total[item * 256 + digit] = carry;
}

// stable scatter of one tile
__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in, size_t n, int shift,
                                                           int ntiles, const uint32_t* __restrict__ hist,
                                                           const uint32_t* __restrict__ digit_total,
                                                           uint32_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t slot_cnt[16][256];      // [round*4 + wave][digit]
    __shared__ uint32_t dbase[256];             // exclusive prefix over the digit totals of this item
    const size_t item = blockIdx.y;
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16 * 256; i += 256) (&slot_cnt[0][0])[i] = 0;
    {   // exclusive prefix over the 256 digit totals: wave scans + the four wave totals (one barrier)
        __shared__ uint32_t dtot[4];
        const uint32_t mine = digit_total[item * 256 + tid];
        const uint32_t incl = wave_incl_scan_u32(mine, lane);
        if (lane == 63) dtot[wave] = incl;
        __syncthreads();
        const uint32_t before = (wave > 0 ? dtot[0] : 0u) + (wave > 1 ? dtot[1] : 0u) + (wave > 2 ? dtot[2] : 0u);
        dbase[tid] = before + incl - mine;
    }
    __syncthreads();
    const size_t base = (size_t)tile * kTile;
    uint32_t key[4], val[4], rank[4];
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t i = base + r * 256 + tid;
        ok[r] = i < n;
        key[r] = ok[r] ? keys_in[item * n + i] : 0xFFFFFFFFu;
        val[r] = ok[r] ? vals_in[item * n + i] : 0u;
        const uint32_t d = (key[r] >> shift) & 0xFF;
        // lanes holding the same digit (inactive tail lanes are excluded through `ok`)
        unsigned long long same = __ballot(ok[r]);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned long long below = same & ((1ull << lane) - 1ull);
        rank[r] = (uint32_t)__popcll(below);
        if (ok[r] && below == 0) slot_cnt[r * 4 + wave][d] = (uint32_t)__popcll(same);
    }
    __syncthreads();
    {   // thread d: exclusive prefix over the 16 (round, wave) slots of digit d
        uint32_t run = 0;
#pragma unroll
        for (int s = 0; s < 16; ++s) { const uint32_t v = slot_cnt[s][tid]; slot_cnt[s][tid] = run; run += v; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (!ok[r]) continue;
        const uint32_t d = (key[r] >> shift) & 0xFF;
        const size_t pos = (size_t)dbase[d] + hist[(item * 256 + d) * ntiles + tile] + slot_cnt[r * 4 + wave][d] + rank[r];
        keys_out[item * n + pos] = key[r];
        vals_out[item * n + pos] = val[r];
    }
}

// cell_begin / cell_end from the sorted keys (arrays pre-zeroed: empty cell <=> begin == end)
__global__ __launch_bounds__(256) void bounds_kernel(const uint32_t* __restrict__ keys, size_t n, size_t cells,
                                                     uint32_t* __restrict__ cell_begin, uint32_t* __restrict__ cell_end) {
    const size_t item = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t* k = keys + item * n;
    const uint32_t c = k[i];
    if (i == 0 || k[i - 1] != c) cell_begin[item * cells + c] = (uint32_t)i;
    if (i == n - 1 || k[i + 1] != c) cell_end[item * cells + c] = (uint32_t)(i + 1);
}

// one wave per cell; 4 cells per workgroup
__global__ __launch_bounds__(256) void cell_sum_kernel(const float4* __restrict__ entries, size_t n,
                                                       const uint32_t* __restrict__ vals,
                                                       const uint32_t* __restrict__ cell_begin,
                                                       const uint32_t* __restrict__ cell_end, size_t cells,
                                                       float2* __restrict__ out_field, float2* __restrict__ out_sum,
                                                       float* __restrict__ out_cnt, const float* __restrict__ weights) {
    // weights (optional, one per entry): add_vector_weighted (motion_field.rs:164-178); absent = add_vector's 1.0
    __shared__ float2 stage[4][64];
    __shared__ float wstage[4][64];
    const size_t item = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t cell = (size_t)blockIdx.x * 4 + wave;
    if (cell >= cells) return;                         // wave-uniform; no block barrier below
    const uint32_t b = cell_begin[item * cells + cell], e = cell_end[item * cells + cell];
    float sum = 0.0f, cnt = kF32Eps;                   // motion_field.rs:133-138
    for (uint32_t k0 = b; k0 < e; k0 += 64) {
        const uint32_t k = k0 + lane;
        if (k < e) {
            const uint32_t src = vals[item * n + k];
            const float4 en = entries[item * n + src];
            stage[wave][lane] = make_float2(en.z, en.w);
            wstage[wave][lane] = weights ? weights[item * n + src] : 1.0f;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int m = (int)min(64u, e - k0);
        if (lane < 2) {
            const float* col = reinterpret_cast<const float*>(&stage[wave][0]) + lane;
            for (int j = 0; j < m; ++j) {
                const float wgt = wstage[wave][j];
                cnt += wgt;                            // :142-143
                sum = col[2 * j] * wgt + sum;          // :144-146 (motion * weight + column)
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane < 2) {
        if (out_field) reinterpret_cast<float*>(out_field + item * cells + cell)[lane] = sum / cnt;   // :304
        if (out_sum) reinterpret_cast<float*>(out_sum + item * cells + cell)[lane] = sum;      // densifier state before
        if (out_cnt && lane == 0) out_cnt[item * cells + cell] = cnt;                          // the final divide
    }
}

// ---- block-vector sized problems (n <= 8,192 entries, <= 256 cells: the detector's 14 x 14 grid over one frame pair's
// vectors): the whole densifier in ONE workgroup per item -- the same stable counting sort and the same per-cell
// sequential sums as the six-kernel path, with workgroup barriers where that path has launch boundaries (each costs
// ~5 us against a microsecond of work at this size).  Entries keep their input order through (slot = 64 consecutive
// entries, lane); keys are cell ids (one 8-bit digit); the motions wait in LDS for the per-cell walk.
constexpr int kSmallMaxN = 8192, kSmallMaxCells = 256, kSmallSlots = kSmallMaxN / 64;
constexpr size_t kSmallLds = (size_t)kSmallMaxN * sizeof(float2) + (size_t)kSmallMaxN * sizeof(uint16_t) +
                             (size_t)kSmallSlots * 256 * sizeof(uint16_t) + (4 * 256 + 256 + 256) * sizeof(uint32_t);

__global__ __launch_bounds__(1024) void densify_small_kernel(const float4* __restrict__ entries, uint32_t n, int w, int h,
                                                             float2* __restrict__ out_field, uint32_t* __restrict__ cell_begin,
                                                             uint32_t* __restrict__ cell_end) {
    extern __shared__ __attribute__((aligned(16))) uint8_t small_lds[];
    float2* mot = reinterpret_cast<float2*>(small_lds);                          // [kSmallMaxN] motion of entry i
    uint16_t* sorted = reinterpret_cast<uint16_t*>(mot + kSmallMaxN);            // [kSmallMaxN] entry index by (cell, input order)
    uint16_t* slot = sorted + kSmallMaxN;                                        // [kSmallSlots][256] entries of cell c in slot s
    uint32_t* partsum = reinterpret_cast<uint32_t*>(slot + kSmallSlots * 256);   // [4][256] slot-quarter totals -> their prefix
    uint32_t* total = partsum + 4 * 256;                                         // [256] entries per cell
    uint32_t* dbase = total + 256;                                               // [256] first sorted position of a cell
    const size_t item = blockIdx.x;
    const int cells = w * h;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < kSmallSlots * 256 / 8; i += 1024) reinterpret_cast<uint4*>(slot)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // ---- cell ids, motions to LDS, rank inside the 64-entry slot (8-ballot same-cell match)
    constexpr int ROUNDS = kSmallSlots / 16;                                     // slots per wave
    uint32_t key[ROUNDS], rank[ROUNDS];
    bool ok[ROUNDS];
    float4 ent[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {                                           // all of a thread's entries requested at once: the
        const uint32_t i = (uint32_t)(wave * ROUNDS + r) * 64u + (uint32_t)lane;  // ballots below would expose one round trip per round
        ent[r] = entries[item * n + (i < n ? i : n - 1)];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int s = wave * ROUNDS + r;
        const uint32_t i = (uint32_t)s * 64u + (uint32_t)lane;
        ok[r] = i < n;
        uint32_t k = 0;
        if (ok[r]) {
            const float4 e = ent[r];
            uint32_t x, y;
            densifier_cell(e.x, e.y, w, h, x, y);
            k = y * (uint32_t)w + x;
            mot[i] = make_float2(e.z, e.w);
        }
        key[r] = k;
        unsigned long long same = __ballot(ok[r]);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((k >> b) & 1u);
            same &= ((k >> b) & 1u) ? bal : ~bal;
        }
        const unsigned long long below = same & ((1ull << lane) - 1ull);
        rank[r] = (uint32_t)__popcll(below);
        if (ok[r] && below == 0) slot[s * 256 + k] = (uint16_t)__popcll(same);
    }
    __syncthreads();
    // ---- per cell: exclusive prefix of its counts over the slots (four threads per cell, a quarter of the slots each)
    {
        const int d = tid & 255, part = tid >> 8;
        constexpr int PER = kSmallSlots / 4;
        uint32_t run = 0;
        for (int s = part * PER; s < (part + 1) * PER; ++s) { const uint32_t v = slot[s * 256 + d]; slot[s * 256 + d] = (uint16_t)run; run += v; }
        partsum[part * 256 + d] = run;
    }
    __syncthreads();
    if (tid < 256) {
        const uint32_t t0 = partsum[tid], t1 = partsum[256 + tid], t2 = partsum[512 + tid], t3 = partsum[768 + tid];
        partsum[tid] = 0; partsum[256 + tid] = t0; partsum[512 + tid] = t0 + t1; partsum[768 + tid] = t0 + t1 + t2;
        total[tid] = t0 + t1 + t2 + t3;
    }
    __syncthreads();
    if (tid < 64) {                                                              // exclusive prefix over the 256 cell totals
        const uint32_t a0 = total[4 * tid], a1 = total[4 * tid + 1], a2 = total[4 * tid + 2], a3 = total[4 * tid + 3];
        uint32_t incl = a0 + a1 + a2 + a3;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const uint32_t base = incl - (a0 + a1 + a2 + a3);
        dbase[4 * tid] = base; dbase[4 * tid + 1] = base + a0; dbase[4 * tid + 2] = base + a0 + a1; dbase[4 * tid + 3] = base + a0 + a1 + a2;
    }
    __syncthreads();
    if (tid < cells) { cell_begin[item * cells + tid] = dbase[tid]; cell_end[item * cells + tid] = dbase[tid] + total[tid]; }
    // ---- stable scatter of the entry indices
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (!ok[r]) continue;
        const int s = wave * ROUNDS + r;
        const uint32_t pos = dbase[key[r]] + partsum[(s / (kSmallSlots / 4)) * 256 + key[r]] + slot[s * 256 + key[r]] + rank[r];
        sorted[pos] = (uint16_t)(s * 64 + lane);
    }
    __syncthreads();
    // ---- per cell and component: the reference's sequential sum in input order, then sum / count
    if (tid < 2 * cells) {
        const int c = tid >> 1, comp = tid & 1;
        const uint32_t b = dbase[c], e = b + total[c];
        const float* m = reinterpret_cast<const float*>(mot) + comp;
        float sum = 0.0f, cnt = kF32Eps;                   // motion_field.rs:133-138
        const float wgt = 1.0f;
        uint32_t k = b;
        for (; k + 4 <= e; k += 4) {                       // four independent index -> motion fetches, then the dependent adds in order
            const float v0 = m[2 * sorted[k]], v1 = m[2 * sorted[k + 1]], v2 = m[2 * sorted[k + 2]], v3 = m[2 * sorted[k + 3]];
            cnt += wgt; sum = v0 * wgt + sum;              // :142-146
            cnt += wgt; sum = v1 * wgt + sum;
            cnt += wgt; sum = v2 * wgt + sum;
            cnt += wgt; sum = v3 * wgt + sum;
        }
        for (; k < e; ++k) {
            cnt += wgt;                                    // :142-143
            sum = m[2 * sorted[k]] * wgt + sum;            // :144-146
        }
        reinterpret_cast<float*>(out_field + item * cells + c)[comp] = sum / cnt;      // :304
    }
}

// ---- per-pixel producers (cv-decoder/src/lib.rs:239-291: one record per pixel in raster order, position
// ((x+.5)/W, (y+.5)/H), optionally only where a mask is set).  For such input the cell of a record depends on its column
// only (x index) and on its row only (y index), monotonically, so the records of one cell are a RECTANGLE of pixels and
// their input order is the raster order of that rectangle: no sort is needed to add them in the reference's order.  One
// wave per cell finds its rectangle by bisection on the densifier's own cell function, gathers 64 records at a time
// (mask-compacted in order through a ballot) and adds them exactly like cell_sum_kernel -- same operations in the same
// order, same bits, one launch instead of nine and the records read once instead of twice.
__device__ __forceinline__ int raster_cell_of(int i, float inv_n, int cells_1d) {   // cell index of pixel column/row i
    uint32_t cx, cy;
    const float p = ((float)i + 0.5f) * inv_n;              // the producer's position expression (lk.hip lk_store)
    densifier_cell(p, p, cells_1d, cells_1d, cx, cy);        // 0 < p < 1: the all-component clamp acts per component
    return (int)cx;
}
// min i in [0, n] with cell(i) >= c (n when there is none).  The cell function is monotone, so any starting guess walks to
// the boundary; the guess inverts round(p * (cells - 1)) and is off by at most a pixel or two -- ~3 evaluations of the
// exact function instead of the 11 of a bisection over 1920 columns (which, run redundantly by every lane of every
// cell's wave, was most of this kernel's time).
__device__ __forceinline__ int raster_first_at_least(int c, int n, float inv_n, int cells_1d) {
    if (c <= 0) return 0;
    if (c >= cells_1d) return n;
    const float est = ((float)c - 0.5f) * (float)n / (float)(cells_1d - 1) - 0.5f;
    int i = (int)ceilf(est);
    i = i < 0 ? 0 : (i > n ? n : i);
    while (i > 0 && raster_cell_of(i - 1, inv_n, cells_1d) >= c) --i;
    while (i < n && raster_cell_of(i, inv_n, cells_1d) < c) ++i;
    return i;
}

__global__ __launch_bounds__(256) void raster_cell_sum_kernel(const float4* __restrict__ entries, const uint8_t* __restrict__ mask,
                                                              int W, int H, int w, int h, float2* __restrict__ out_field,
                                                              uint32_t* __restrict__ cell_begin, uint32_t* __restrict__ cell_end,
                                                              float4* __restrict__ xmajor = nullptr) {
    __shared__ float2 stage[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + wave;
    if (cell >= w * h) return;                         // wave-uniform; no block barrier below
    const int cx = cell % w, cy = cell / w;
    const float nx = 1.0f / (float)W, ny = 1.0f / (float)H;
    // the four boundaries on four lanes at once
    const bool is_y = (lane & 2) != 0;
    const int bnd = raster_first_at_least((is_y ? cy : cx) + (lane & 1), is_y ? H : W, is_y ? ny : nx, is_y ? h : w);
    const int x0 = __builtin_amdgcn_readlane(bnd, 0), x1 = __builtin_amdgcn_readlane(bnd, 1);
    const int y0 = __builtin_amdgcn_readlane(bnd, 2), y1 = __builtin_amdgcn_readlane(bnd, 3);
    const int cw = x1 - x0, total = cw * (y1 - y0);
    float sum = 0.0f, cnt = kF32Eps;                   // motion_field.rs:133-138
    uint32_t kept = 0;
    // four 64-record rounds are requested together (a 1080p -> 150 x 84 cell holds ~170 records: one memory round trip
    // instead of three), then staged and added round by round
    constexpr int AHEAD = 4;
    for (int k00 = 0; k00 < total; k00 += 64 * AHEAD) {
        bool on[AHEAD];
        float2 mv[AHEAD];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            const int k = k00 + 64 * a + lane;
            on[a] = k < total;
            mv[a] = make_float2(0.0f, 0.0f);
            if (on[a]) {
                const int ry = k / cw;
                const size_t idx = (size_t)(y0 + ry) * W + (x0 + (k - ry * cw));
                on[a] = !mask || mask[idx];
                if (on[a]) { const float4 en = entries[idx]; mv[a] = make_float2(en.z, en.w); }
            }
        }
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            if (k00 + 64 * a >= total) break;          // uniform
            const unsigned long long bal = __ballot(on[a]);
            if (on[a]) stage[wave][__popcll(bal & ((1ull << lane) - 1ull))] = mv[a];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int m = __popcll(bal);
            kept += (uint32_t)m;
            if (lane < 2) {
                const float* col = reinterpret_cast<const float*>(&stage[wave][0]) + lane;
                const float wgt = 1.0f;
                int j = 0;
                for (; j + 8 <= m; j += 8) {               // eight staged values fetched together, then the dependent adds in order
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = col[2 * (j + u)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sum = v[u] * wgt + sum;                      // :144-146 (motion * weight + column)
                }
                for (; j < m; ++j) sum = col[2 * j] * wgt + sum;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    // :142-143 adds 1.0 to the count per vector, starting from f32::EPSILON (:133-138).  That chain has a closed form: eps + 1 = 1 + 2^-23
    // exactly; (1 + 2^-23) + 1 is the tie 2 + 2^-23 and rounds to the even 2.0; from there every + 1 is exact -- so the count after k
    // vectors is eps, 1 + eps, or min(k, 2^24) itself, and the walk carries ONE dependent chain per lane instead of two (it is bound by VALU issue:
    // two lanes of a wave do the adding)
    cnt = kept == 0 ? kF32Eps : (kept == 1 ? 1.0f + kF32Eps : (kept >= (1u << 24) ? 16777216.0f : (float)kept));      // (+ 1 stops changing 2^24)
    if (lane < 2) reinterpret_cast<float*>(out_field + cell)[lane] = sum / cnt;   // :304
    if (lane == 0) { cell_begin[cell] = 0; cell_end[cell] = kept; }                // visited <=> end > begin
    // the same cell in the order the records go out in ((x, y)-sorted: cv-decoder/src/lib.rs:279-291), with its visited flag, for
    // cells_to_entries_x_kernel: that kernel then reads 16 coalesced bytes per cell instead of three words a row pitch apart
    if (xmajor && lane < 3) reinterpret_cast<float*>(xmajor + (size_t)cx * h + cy)[lane] = lane < 2 ? sum / cnt : (kept ? 1.0f : 0.0f);
}

// cells_to_entries_kernel for one item whose cells arrive (x, y)-sorted with their visited flag (raster_cell_sum_kernel's xmajor).  One
// workgroup per 1,024 cells: it counts the visited cells before its chunk itself (flags of at most 64 K cells: a few loads per thread),
// so the chunks' records -- usually headed for a page-locked host block -- leave from several CUs at once instead of one.
__global__ __launch_bounds__(1024) void cells_to_entries_x_kernel(const float4* __restrict__ xmajor, int w, int h, float4* __restrict__ out_entries,
                                                                  uint32_t* __restrict__ out_count) {
    __shared__ uint32_t wave_cnt[2][16];
    const size_t cells = (size_t)w * h;
    const float nx = 1.0f / (float)w, ny = 1.0f / (float)h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t c0 = (size_t)blockIdx.x * 1024;
    uint32_t mine = 0;
    for (size_t i = threadIdx.x; i < c0; i += 1024) mine += xmajor[i].z != 0.0f ? 1u : 0u;
    const size_t o = c0 + threadIdx.x;
    const float4 v = o < cells ? xmajor[o] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool vis = v.z != 0.0f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    const unsigned long long bal = __ballot(vis);
    if (lane == 0) { wave_cnt[0][wave] = mine; wave_cnt[1][wave] = (uint32_t)__popcll(bal); }
    __syncthreads();
    uint32_t base = 0, before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { base += wave_cnt[0][k]; const uint32_t c = wave_cnt[1][k]; total += c; before += k < wave ? c : 0u; }
    if (vis) {
        const int x = (int)(o / h), y = (int)(o % h);
        const uint32_t pos = base + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        out_entries[pos] = make_float4(((float)x + 0.5f) * nx, ((float)y + 0.5f) * ny, v.x, v.y);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out_count[0] = base + total;
}

// test aid: does every record sit where the rectangle walk assumes?  flag[0] counts records whose stored position maps
// to another cell than (cell(x), cell(y)) or whose stored position is not the producer's expression
__global__ __launch_bounds__(256) void raster_check_kernel(const float4* __restrict__ entries, int W, int H, int w, int h,
                                                           uint32_t* __restrict__ flag) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float4 e = entries[(size_t)y * W + x];
    uint32_t cx, cy;
    densifier_cell(e.x, e.y, w, h, cx, cy);
    const float nx = 1.0f / (float)W, ny = 1.0f / (float)H;
    const bool ok = (int)cx == raster_cell_of(x, nx, w) && (int)cy == raster_cell_of(y, ny, h) &&
                    e.x == ((float)x + 0.5f) * nx && e.y == ((float)y + 0.5f) * ny;
    if (!ok) atomicAdd(flag, 1u);
}

// cv-decoder/src/lib.rs:279-291: visited cells in BTreeSet<(x,y)> order -> entries.  One
// workgroup per item walks the x-major cell order in 1024-cell chunks with a running offset.
__global__ __launch_bounds__(1024) void cells_to_entries_kernel(const float2* __restrict__ field,
                                                                const uint32_t* __restrict__ cell_begin,
                                                                const uint32_t* __restrict__ cell_end, int w, int h,
                                                                float4* __restrict__ out_entries,
                                                                uint32_t* __restrict__ out_count) {
    // ordered compaction, 1024 cells per round: rank inside the wave from a ballot, the 16 wave totals through LDS
    // (two barriers per round; a 10-step scan over the workgroup with its 20 barriers took 37 us at 150 x 84)
    __shared__ uint32_t wave_cnt[2][16];
    const size_t item = blockIdx.x;
    const size_t cells = (size_t)w * h;
    const float nx = 1.0f / (float)w, ny = 1.0f / (float)h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t base = 0;                                     // the same running offset in every thread
    int par = 0;
    for (size_t c0 = 0; c0 < cells; c0 += 1024, par ^= 1) {
        const size_t o = c0 + threadIdx.x;                 // position in (x, y)-sorted order
        const int x = (int)(o / h), y = (int)(o % h);
        const size_t idx = (size_t)y * w + x;
        const bool vis = o < cells && cell_end[item * cells + idx] > cell_begin[item * cells + idx];
        const unsigned long long bal = __ballot(vis);
        if (lane == 0) wave_cnt[par][wave] = (uint32_t)__popcll(bal);
        __syncthreads();                                   // (the slots alternate, so one barrier per round is enough)
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t v = wave_cnt[par][k]; total += v; before += k < wave ? v : 0u; }
        if (vis) {
            const uint32_t pos = base + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            const float2 m = field[item * cells + idx];
            out_entries[item * cells + pos] =
                make_float4(((float)x + 0.5f) * nx, ((float)y + 0.5f) * ny, m.x, m.y);
        }
        base += total;
    }
    if (threadIdx.x == 0) out_count[item] = base;
}

// MotionFieldDensifier::interpolate_empty_cells (motion_field.rs:193-294) + MotionField::from.
// The reference pops cells from a BTreeSet ordered by (-filled_neighbours, index); every fill changes the
// keys of its neighbours and the data later fills read, so the walk is sequential by definition.  One
// workgroup per item: all threads find the next cell (argmin of the key over the queue), thread 0 performs
// the reference's f32 arithmetic for that cell in the reference's neighbour order -- same operations, same
// order, same bits.  Cost grows with the number of empty cells (an offline path: flow-extract/src/main.rs:81).
__device__ __forceinline__ int interp_filled_neighbours(const float* cnt, int w, int h, int i) {
    const int x = i % w, y = i / w;
    const int ox[6] = {-1, 0, -1, 1, 0, 1}, oy[6] = {0, -1, -1, 0, 1, 1};          // motion_field.rs:207
    int c = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int nx = x + ox[k], ny = y + oy[k];
        if (nx >= 0 && nx < w && ny >= 0 && ny < h && cnt[nx + ny * w] > 0.1f) ++c;
    }
    return c;
}

__global__ __launch_bounds__(256) void interpolate_kernel(float2* __restrict__ sum, float* __restrict__ cnt,
                                                          int* __restrict__ key, int w, int h,
                                                          float2* __restrict__ out_field) {
    __shared__ unsigned long long red[4];
    __shared__ int chosen;
    const size_t item = blockIdx.x;
    const int cells = w * h, tid = threadIdx.x;
    sum += item * cells; cnt += item * cells; key += item * cells; out_field += item * cells;
    const int NOT_QUEUED = 0x7FFFFFFF;
    // queue = cells with count < 0.5, key = -filled neighbours (:230-241)
    int queued = 0;
    for (int i = tid; i < cells; i += 256) {
        const bool q = cnt[i] < 0.5f;
        key[i] = q ? -interp_filled_neighbours(cnt, w, h, i) : NOT_QUEUED;
        queued += q;
    }
    queued = __syncthreads_count(queued > 0) ? 1 : 0;     // any queued cell at all?
    {
        // "no motion vectors at all" (:243-246): every cell queued -> nothing to interpolate from
        int filled = 0;
        for (int i = tid; i < cells; i += 256) filled += key[i] == NOT_QUEUED;
        if (!__syncthreads_or(filled)) queued = 0;
    }
    while (queued) {
        // smallest (key, index) over the queue
        unsigned long long best = ~0ull;
        for (int i = tid; i < cells; i += 256) {
            const int k = key[i];
            if (k != NOT_QUEUED) {
                const unsigned long long v = ((unsigned long long)(uint32_t)(k + 16) << 32) | (uint32_t)i;
                best = v < best ? v : best;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)best, m, 64), hi = __shfl_xor((unsigned)(best >> 32), m, 64);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            best = o < best ? o : best;
        }
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = red[0];
            for (int k = 1; k < 4; ++k) b = red[k] < b ? red[k] : b;
            chosen = (b == ~0ull) ? -1 : (int)(uint32_t)b;
            if (chosen >= 0) {
                const int i = chosen, x = i % w, y = i / w;
                const int ox[6] = {-1, 0, -1, 1, 0, 1}, oy[6] = {0, -1, -1, 0, 1, 1};
                key[i] = NOT_QUEUED;                                             // queue.take
                float2 acc = sum[i];
                float c = cnt[i];
                bool added = false;
                for (int k = 0; k < 6; ++k) {                                    // :255-268
                    const int nx = x + ox[k], ny = y + oy[k];
                    if (nx < 0 || nx >= w || ny < 0 || ny >= h) continue;
                    const int idx = nx + ny * w;
                    const float nc = cnt[idx];
                    if (nc > 0.1f) {
                        const float scale = 1.0f - sqrtf((float)(ox[k] * ox[k] + oy[k] * oy[k])) * 0.5f;
                        const float inv_cnt = 1.0f / nc;
                        const float sc = scale * inv_cnt;
                        const float2 nv = sum[idx];
                        c += scale;                                              // add_vector_idx (:141-147)
                        acc.x = (sc * nv.x) * scale + acc.x;
                        acc.y = (sc * nv.y) * scale + acc.y;
                        added = true;
                    }
                }
                if (!added) chosen = -1;           // cannot happen with a filled cell present; the reference would spin
                sum[i] = acc;
                cnt[i] = c;
                __threadfence_block();
                if (added)
                    for (int k = 0; k < 6; ++k) {                                // :273-289
                        const int nx = x + ox[k], ny = y + oy[k];
                        if (nx < 0 || nx >= w || ny < 0 || ny >= h) continue;
                        const int idx = nx + ny * w;
                        if (key[idx] != NOT_QUEUED) key[idx] = -interp_filled_neighbours(cnt, w, h, idx);
                    }
            }
        }
        __syncthreads();
        if (chosen < 0) break;
        __syncthreads();
    }
    __syncthreads();
    for (int i = tid; i < cells; i += 256) {                                     // MotionField::from
        const float2 sv = sum[i];
        const float c = cnt[i];
        out_field[i] = make_float2(sv.x / c, sv.y / c);
    }
}

// Shared by detect.hip: densify `batch` items of n entries into (w x h) fields.  Leaves the
// per-cell [begin,end) tables in S_WORK3 (begin) / S_WORK4 (end).
int densify_device(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int batch, int w, int h, float2* d_field,
                   uint32_t* d_cells, uint32_t** out_begin, uint32_t** out_end) {
    return densify_device_raw(ctx, d_entries, n, batch, w, h, d_field, d_cells, out_begin, out_end, nullptr, nullptr, nullptr);
}

int densify_device_raw(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int batch, int w, int h, float2* d_field,
                       uint32_t* d_cells, uint32_t** out_begin, uint32_t** out_end, float2* d_sum, float* d_cnt,
                       const float* d_weights) {
    const size_t cells = (size_t)w * (size_t)h;
    OFPS_REQUIRE(ctx, w >= 1 && h >= 1 && cells <= 65536, "densify: grid %dx%d unsupported (1..65536 cells)", w, h);
    OFPS_REQUIRE(ctx, batch >= 1 && batch <= 65535, "densify: batch %d out of range", batch);
    OFPS_REQUIRE(ctx, n < (1ull << 31), "densify: too many entries");
    hipStream_t s = ctx->stream;
    const size_t tot = n * (size_t)batch;
    auto* begin = static_cast<uint32_t*>(scratch(ctx, S_WORK3, cells * batch * sizeof(uint32_t)));
    auto* end = static_cast<uint32_t*>(scratch(ctx, S_WORK4, cells * batch * sizeof(uint32_t)));
    if (!begin || !end) return OFPS_HIP_ENOMEM;
    if (n == 0) {                                   // otherwise cell_kernel clears them
        OFPS_HIP_TRY(ctx, hipMemsetAsync(begin, 0, cells * batch * sizeof(uint32_t), s));
        OFPS_HIP_TRY(ctx, hipMemsetAsync(end, 0, cells * batch * sizeof(uint32_t), s));
    }
    if (out_begin) *out_begin = begin;
    if (out_end) *out_end = end;
    if (n > 0 && n <= (size_t)kSmallMaxN && cells <= (size_t)kSmallMaxCells && !d_cells && !d_sum && !d_cnt && !d_weights && d_field &&
        !ctx->opt.densify_no_small) {                              // (the variable: A/B runs and tests of the general path)
        static bool attr_set[64] = {};
        if (!attr_set[ctx->device & 63]) {
            OFPS_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(densify_small_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmallLds));
            attr_set[ctx->device & 63] = true;
        }
        hipLaunchKernelGGL(densify_small_kernel, dim3(batch), dim3(1024), kSmallLds, s, d_entries, (uint32_t)n, w, h, d_field, begin, end);
        OFPS_HIP_TRY(ctx, hipGetLastError());
        return OFPS_HIP_OK;
    }
    uint32_t* sorted_vals = nullptr;
    if (n > 0) {
        const int ntiles = (int)((n + kTile - 1) / kTile);
        auto* k0 = static_cast<uint32_t*>(scratch(ctx, S_WORK0, 2 * tot * sizeof(uint32_t)));
        auto* k1 = static_cast<uint32_t*>(scratch(ctx, S_WORK1, 2 * tot * sizeof(uint32_t)));
        auto* hist = static_cast<uint32_t*>(scratch(ctx, S_WORK2, (size_t)batch * 256 * (ntiles + 1) * sizeof(uint32_t)));
        if (!k0 || !k1 || !hist) return OFPS_HIP_ENOMEM;
        uint32_t* digit_total = hist + (size_t)batch * 256 * ntiles;
        uint32_t *keys_a = k0, *vals_a = k0 + tot, *keys_b = k1, *vals_b = k1 + tot;
        const dim3 ge((unsigned)((n + 255) / 256), batch), gt(ntiles, batch);
        hipLaunchKernelGGL(cell_kernel, ge, dim3(256), 0, s, d_entries, n, w, h, keys_a, vals_a, d_cells, begin, end, cells);
        const int passes = cells <= 256 ? 1 : 2;
        for (int p = 0; p < passes; ++p) {
            const int shift = 8 * p;
            hipLaunchKernelGGL(sort_count_kernel, gt, dim3(256), 0, s, keys_a, n, shift, ntiles, hist);
            hipLaunchKernelGGL(sort_scan_kernel, dim3(256, batch), dim3(256), 0, s, hist, ntiles, digit_total);
            hipLaunchKernelGGL(sort_scatter_kernel, gt, dim3(256), 0, s, keys_a, vals_a, n, shift, ntiles, hist, digit_total,
                               keys_b, vals_b);
            uint32_t* t;
            t = keys_a; keys_a = keys_b; keys_b = t;
            t = vals_a; vals_a = vals_b; vals_b = t;
        }
        hipLaunchKernelGGL(bounds_kernel, ge, dim3(256), 0, s, keys_a, n, cells, begin, end);
        sorted_vals = vals_a;
    }
    hipLaunchKernelGGL(cell_sum_kernel, dim3((unsigned)((cells + 3) / 4), batch), dim3(256), 0, s, d_entries, n,
                       sorted_vals, begin, end, cells, d_field, d_sum, d_cnt, d_weights);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

// densify + cv-decoder's visited-cell records (cv-decoder/src/lib.rs:279-291), all on the device
int densify_entries_device(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int w, int h, float2* d_field,
                           float4* d_out_entries, uint32_t* d_count) {
    uint32_t *begin = nullptr, *end = nullptr;
    int rc = densify_device(ctx, d_entries, n, 1, w, h, d_field, nullptr, &begin, &end);
    if (rc != OFPS_HIP_OK) return rc;
    hipLaunchKernelGGL(cells_to_entries_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_field, begin, end, w, h, d_out_entries,
                       d_count);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

// densify for per-pixel raster producers (see raster_cell_sum_kernel).  d_mask: W*H bytes or nullptr.  Leaves visited
// tables in S_WORK3 / S_WORK4 like densify_device.
int densify_raster_device(ofps_hip_ctx* ctx, const float4* d_entries, const uint8_t* d_mask, int W, int H, int w, int h,
                          float2* d_field, uint32_t** out_begin, uint32_t** out_end, float4* d_xmajor) {
    const size_t cells = (size_t)w * (size_t)h;
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && (size_t)W * H < (1ull << 31), "densify_raster: bad frame %dx%d", W, H);
    OFPS_REQUIRE(ctx, w >= 1 && h >= 1 && cells <= 65536, "densify_raster: grid %dx%d unsupported (1..65536 cells)", w, h);
    auto* begin = static_cast<uint32_t*>(scratch(ctx, S_WORK3, cells * sizeof(uint32_t)));
    auto* end = static_cast<uint32_t*>(scratch(ctx, S_WORK4, cells * sizeof(uint32_t)));
    if (!begin || !end) return OFPS_HIP_ENOMEM;
    if (out_begin) *out_begin = begin;
    if (out_end) *out_end = end;
    hipLaunchKernelGGL(raster_cell_sum_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, ctx->stream, d_entries, d_mask, W, H,
                       w, h, d_field, begin, end, d_xmajor);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

int densify_raster_entries_device(ofps_hip_ctx* ctx, const float4* d_entries, const uint8_t* d_mask, int W, int H, int w, int h,
                                  float2* d_field, float4* d_out_entries, uint32_t* d_count) {
    uint32_t *begin = nullptr, *end = nullptr;
    auto* xmajor = static_cast<float4*>(scratch(ctx, S_XMAJOR, (size_t)w * h * sizeof(float4)));
    if (!xmajor) return OFPS_HIP_ENOMEM;
    int rc = densify_raster_device(ctx, d_entries, d_mask, W, H, w, h, d_field, &begin, &end, xmajor);
    if (rc != OFPS_HIP_OK) return rc;
    hipLaunchKernelGGL(cells_to_entries_x_kernel, dim3((unsigned)(((size_t)w * h + 1023) / 1024)), dim3(1024), 0, ctx->stream, (const float4*)xmajor, w, h,
                       d_out_entries, d_count);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

}  // namespace ofps

extern "C" {

int ofps_hip_densify_interpolated(ofps_hip_ctx* ctx, const float* entries, size_t n, int w, int h, float* out_field) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, out_field && (entries || n == 0), "densify_interpolated: null host pointer");
    OFPS_REQUIRE(ctx, w >= 1 && h >= 1 && (size_t)w * h <= 65536, "densify_interpolated: grid %dx%d unsupported", w, h);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)w * h;
    auto* d_ent = static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES, n * sizeof(float4)));
    auto* d_field = static_cast<float2*>(ofps::scratch(ctx, ofps::S_FIELD, cells * sizeof(float2)));
    auto* d_state = static_cast<char*>(ofps::scratch(ctx, ofps::S_BEST, cells * (sizeof(float2) + sizeof(float) + sizeof(int))));
    if (!d_ent || !d_field || !d_state) return OFPS_HIP_ENOMEM;
    auto* d_sum = reinterpret_cast<float2*>(d_state);
    auto* d_cnt = reinterpret_cast<float*>(d_state + cells * sizeof(float2));
    auto* d_key = reinterpret_cast<int*>(d_state + cells * (sizeof(float2) + sizeof(float)));
    if (n) OFPS_HIP_TRY(ctx, hipMemcpyAsync(d_ent, entries, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    int rc = ofps::densify_device_raw(ctx, d_ent, n, 1, w, h, nullptr, nullptr, nullptr, nullptr, d_sum, d_cnt, nullptr);
    if (rc != OFPS_HIP_OK) return rc;
    hipLaunchKernelGGL(ofps::interpolate_kernel, dim3(1), dim3(256), 0, ctx->stream, d_sum, d_cnt, d_key, w, h, d_field);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_field, d_field, cells * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

int ofps_hip_densify_dev(ofps_hip_ctx* ctx, const void* d_entries, size_t n_per_item, int batch, int w, int h,
                         void* d_out_field, void* d_out_cells) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_out_field && (d_entries || n_per_item == 0), "densify: null device pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ofps::densify_device(ctx, static_cast<const float4*>(d_entries), n_per_item, batch, w, h,
                                static_cast<float2*>(d_out_field), static_cast<uint32_t*>(d_out_cells), nullptr,
                                nullptr);
}

int ofps_hip_densify(ofps_hip_ctx* ctx, const float* entries, size_t n, int w, int h, float* out_field,
                     uint32_t* out_cells) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, out_field && (entries || n == 0), "densify: null host pointer");
    OFPS_REQUIRE(ctx, w >= 1 && h >= 1, "densify: bad grid %dx%d", w, h);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)w * h;
    auto* d_ent = static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES, n * sizeof(float4)));
    auto* d_field = static_cast<float2*>(ofps::scratch(ctx, ofps::S_FIELD, cells * sizeof(float2)));
    auto* d_cells = out_cells ? static_cast<uint32_t*>(ofps::scratch(ctx, ofps::S_CELLS, 2 * n * sizeof(uint32_t))) : nullptr;
    if (!d_ent || !d_field || (out_cells && !d_cells)) return OFPS_HIP_ENOMEM;
    if (n) OFPS_HIP_TRY(ctx, hipMemcpyAsync(d_ent, entries, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    int rc = ofps::densify_device(ctx, d_ent, n, 1, w, h, d_field, d_cells, nullptr, nullptr);
    if (rc != OFPS_HIP_OK) return rc;
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_field, d_field, cells * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    if (out_cells && n)
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_cells, d_cells, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

int ofps_hip_densify_weighted(ofps_hip_ctx* ctx, const float* entries, const float* weights, size_t n, int w, int h,
                              float* out_field, uint32_t* out_cells) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, out_field && ((entries && weights) || n == 0), "densify_weighted: null host pointer");
    OFPS_REQUIRE(ctx, w >= 1 && h >= 1, "densify_weighted: bad grid %dx%d", w, h);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)w * h;
    auto* d_ent = static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES, n * sizeof(float4)));
    auto* d_wgt = static_cast<float*>(ofps::scratch(ctx, ofps::S_ENTRIES2, n * sizeof(float)));
    auto* d_field = static_cast<float2*>(ofps::scratch(ctx, ofps::S_FIELD, cells * sizeof(float2)));
    auto* d_cells = out_cells ? static_cast<uint32_t*>(ofps::scratch(ctx, ofps::S_CELLS, 2 * n * sizeof(uint32_t))) : nullptr;
    if (!d_ent || !d_wgt || !d_field || (out_cells && !d_cells)) return OFPS_HIP_ENOMEM;
    if (n) {
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(d_ent, entries, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(d_wgt, weights, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    int rc = ofps::densify_device_raw(ctx, d_ent, n, 1, w, h, d_field, d_cells, nullptr, nullptr, nullptr, nullptr, d_wgt);
    if (rc != OFPS_HIP_OK) return rc;
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_field, d_field, cells * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    if (out_cells && n)
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_cells, d_cells, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

int ofps_hip_densify_to_entries(ofps_hip_ctx* ctx, const float* entries, size_t n, int w, int h, float* out_entries,
                                size_t* n_out) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, out_entries && n_out && (entries || n == 0), "densify_to_entries: null host pointer");
    OFPS_REQUIRE(ctx, w >= 1 && h >= 1, "densify_to_entries: bad grid %dx%d", w, h);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)w * h;
    auto* d_ent = static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES, n * sizeof(float4)));
    auto* d_field = static_cast<float2*>(ofps::scratch(ctx, ofps::S_FIELD, cells * sizeof(float2)));
    auto* d_out = static_cast<float4*>(ofps::scratch(ctx, ofps::S_BEST, cells * sizeof(float4)));
    auto* d_cnt = static_cast<uint32_t*>(ofps::scratch(ctx, ofps::S_RESULT, 16));
    if (!d_ent || !d_field || !d_out || !d_cnt) return OFPS_HIP_ENOMEM;
    if (n) OFPS_HIP_TRY(ctx, hipMemcpyAsync(d_ent, entries, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    int rc = ofps::densify_entries_device(ctx, d_ent, n, w, h, d_field, d_out, d_cnt);
    if (rc != OFPS_HIP_OK) return rc;
    uint32_t cnt = 0;
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(&cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (cnt) OFPS_HIP_TRY(ctx, hipMemcpy(out_entries, d_out, (size_t)cnt * sizeof(float4), hipMemcpyDeviceToHost));
    *n_out = cnt;
    return OFPS_HIP_OK;
}

int ofps_hip_densify_raster_dev(ofps_hip_ctx* ctx, const void* d_entries, const void* d_mask, int W, int H, int w, int h,
                                void* d_out_field, int verify) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_entries && d_out_field, "densify_raster: null device pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (verify) {                                       // blocking; for tests and for hosts that do not own the producer
        OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && w >= 1 && h >= 1, "densify_raster: bad geometry");
        auto* flag = static_cast<uint32_t*>(ofps::scratch(ctx, ofps::S_RESULT, 16));
        if (!flag) return OFPS_HIP_ENOMEM;
        OFPS_HIP_TRY(ctx, hipMemsetAsync(flag, 0, sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(ofps::raster_check_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, ctx->stream,
                           static_cast<const float4*>(d_entries), W, H, w, h, flag);
        uint32_t bad = 0;
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(&bad, flag, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
        OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        OFPS_REQUIRE(ctx, bad == 0, "densify_raster: %u records are not the per-pixel lattice of a %dx%d frame", bad, W, H);
    }
    return ofps::densify_raster_device(ctx, static_cast<const float4*>(d_entries), static_cast<const uint8_t*>(d_mask), W, H, w, h,
                                       static_cast<float2*>(d_out_field), nullptr, nullptr);
}

}  // extern "C"
