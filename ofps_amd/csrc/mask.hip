// mask.hip -- cv-decoder's contrast mask and the masked per-pixel record loop on gfx950
// (cv-decoder/src/lib.rs:203-237 mask, :251-276 loop).
//
//   Sobel(gray, CV_32F, dx=1, dy=1, ksize 5, BORDER_DEFAULT) -> threshold(> 20) -> dilate(MORPH_ELLIPSE 11x11)
//
// The reference gets these from OpenCV (absent here: "parity unpinned"); the definitions restated in
// oracle/ofps_oracle.c (orc_contrast_mask) are what this kernel reproduces, bit for bit -- every intermediate is
// a small integer (|Sobel| <= 36*255), so int32 arithmetic equals OpenCV's f32 accumulation exactly.
//
// One fused kernel: a workgroup owns a 64x16 tile of mask pixels; it stages the 78x30 luma window (halo 7 =
// 2 Sobel + 5 dilation) in LDS, runs the separable derivative ([-1,-2,0,2,1] along x, then along y), keeps the
// thresholded 74x26 window as ballot-packed row masks in LDS and resolves the ellipse with four 128-bit window
// extractions per pixel (rows of equal half-width OR-ed first).  HBM traffic: 1.27 B read per pixel (halo) + 1 B written.
//
// compact_*: order-preserving stream compaction of the per-pixel records by the mask (the reference's raster
// loop `continue`s on masked pixels, so the surviving records keep raster order -- which is also the order
// the densifier's f32 sums depend on).  Tile counts -> single-workgroup scan -> scatter with wave ballots.
#include "common.hpp"

namespace ofps {

constexpr int MT_W = 64, MT_H = 16;            // mask tile
constexpr int MG_W = MT_W + 14, MG_H = MT_H + 14;   // luma window (halo 7)
constexpr int MS_W = MT_W + 10, MS_H = MT_H + 10;   // thresholded window (halo 5)

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - p - 2;
    return p;
}

// extract bits [s, s + n) of the 128-bit row mask (lo = columns 0..63, hi = columns 64..)
__device__ __forceinline__ bool mask_any(unsigned long long lo, unsigned long long hi, int s, int n) {
    const unsigned long long v = s >= 64 ? hi >> (s - 64) : (s ? (lo >> s) | (hi << (64 - s)) : lo);
    return (v & ((1ull << n) - 1ull)) != 0ull;
}

__global__ __launch_bounds__(256) void contrast_mask_kernel(const uint8_t* __restrict__ gray, int W, int H, int stride,
                                                            uint8_t* __restrict__ mask) {
    __shared__ uint8_t g[MG_H][MG_W + 2];
    __shared__ short hx[MG_H][MS_W + 2];
    __shared__ unsigned long long rowbits[MS_H][2];          // thresholded window, one bit per column
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * MT_W, y0 = blockIdx.y * MT_H;
    // luma window; coordinates outside the image follow BORDER_REFLECT_101 (only consumed by Sobel taps of in-image
    // pixels: out-of-image thresholded pixels are forced to 0 below).  One column per thread, rows strided: no
    // per-element division.
    {
        constexpr int RPP = 256 / MG_W;                       // 3 rows per pass
        const int c = tid % MG_W, r0 = tid / MG_W;
        if (r0 < RPP) {
            const int xx = reflect101(x0 - 7 + c, W);
            for (int r = r0; r < MG_H; r += RPP) g[r][c] = gray[(size_t)reflect101(y0 - 7 + r, H) * stride + xx];
        }
    }
    __syncthreads();
    // d/dx: hx(r, c) for the thresholded window's columns (window col c <-> luma col c + 2)
    {
        constexpr int RPP = 256 / MS_W;                       // 3
        const int c = tid % MS_W, r0 = tid / MS_W;
        if (r0 < RPP) {
            for (int r = r0; r < MG_H; r += RPP)
                hx[r][c] = (short)(-(int)g[r][c] - 2 * (int)g[r][c + 1] + 2 * (int)g[r][c + 3] + (int)g[r][c + 4]);
        }
    }
    __syncthreads();
    // d/dy + threshold, packed by ballots: wave w owns rows w, w+4, ...; lanes = columns 0..63, then 64..73.
    // Pixels outside the image never win the dilation's max.
    for (int r = wave; r < MS_H; r += 4) {
        const int yy = y0 - 5 + r;
        const bool row_in = yy >= 0 && yy < H;
        auto thr_at = [&](int c) {
            const int s = -(int)hx[r][c] - 2 * (int)hx[r + 1][c] + 2 * (int)hx[r + 3][c] + (int)hx[r + 4][c];
            const int xx = x0 - 5 + c;
            return s > 20 && row_in && xx >= 0 && xx < W;
        };
        const unsigned long long lo = __ballot(thr_at(lane));
        const unsigned long long hi = __ballot(lane < MS_W - 64 && thr_at(64 + lane));
        if (lane == 0) { rowbits[r][0] = lo; rowbits[r][1] = hi; }
    }
    __syncthreads();
    // dilation by the 11x11 ellipse, row half-widths cvRound(5*sqrt(1 - dy^2/25)) = {0,3,4,5,5,5,5,5,4,3,0}: rows that
    // share a half-width are OR-ed first (wave-uniform), then one window extraction per half-width and lane
    const int lx = lane;
#pragma unroll
    for (int k = 0; k < MT_H / 4; ++k) {
        const int ly = wave + 4 * k;
        unsigned long long m0l = rowbits[ly][0] | rowbits[ly + 10][0], m0h = rowbits[ly][1] | rowbits[ly + 10][1];
        unsigned long long m3l = rowbits[ly + 1][0] | rowbits[ly + 9][0], m3h = rowbits[ly + 1][1] | rowbits[ly + 9][1];
        unsigned long long m4l = rowbits[ly + 2][0] | rowbits[ly + 8][0], m4h = rowbits[ly + 2][1] | rowbits[ly + 8][1];
        unsigned long long m5l = 0, m5h = 0;
#pragma unroll
        for (int i = 3; i <= 7; ++i) { m5l |= rowbits[ly + i][0]; m5h |= rowbits[ly + i][1]; }
        const bool m = mask_any(m0l, m0h, lx + 5, 1) || mask_any(m3l, m3h, lx + 2, 7) || mask_any(m4l, m4h, lx + 1, 9) ||
                       mask_any(m5l, m5h, lx, 11);
        const int x = x0 + lx, y = y0 + ly;
        if (x < W && y < H) mask[(size_t)y * W + x] = m ? 1 : 0;
    }
}

// ---- ordered compaction of 16-byte records by a byte mask ------------------------------------------------
constexpr int CT = 1024;          // records per tile (256 threads x 4 consecutive records)

__global__ __launch_bounds__(256) void compact_count_kernel(const uint8_t* __restrict__ mask, size_t n,
                                                            uint32_t* __restrict__ tile_cnt) {
    __shared__ uint32_t wsum[4];
    const size_t base = (size_t)blockIdx.x * CT + (size_t)threadIdx.x * 4;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c += (base + k < n && mask[base + k]) ? 1u : 0u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the tile counts in place; total -> *out_count
__global__ __launch_bounds__(1024) void compact_scan_kernel(uint32_t* __restrict__ tile_cnt, int ntiles,
                                                            uint32_t* __restrict__ out_count) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < ntiles; t0 += 1024) {
        const int t = t0 + tid;
        const uint32_t v = t < ntiles ? tile_cnt[t] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < wave; ++k) woff += wtot[k];
        const uint32_t c0 = carry;
        if (t < ntiles) tile_cnt[t] = c0 + woff + inc - v;
        __syncthreads();
        if (tid == 1023) carry = c0 + woff + inc;
        __syncthreads();
    }
    if (tid == 0) *out_count = carry;
}

__global__ __launch_bounds__(256) void compact_scatter_kernel(const float4* __restrict__ in, const uint8_t* __restrict__ mask,
                                                              size_t n, const uint32_t* __restrict__ tile_off,
                                                              float4* __restrict__ out) {
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)blockIdx.x * CT + (size_t)tid * 4;
    bool keep[4];
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { keep[k] = base + k < n && mask[base + k]; c += keep[k] ? 1u : 0u; }
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t pos = tile_off[blockIdx.x] + inc - c;
    for (int k = 0; k < wave; ++k) pos += wsum[k];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (keep[k]) out[pos++] = in[base + k];
}

// Order-preserving compaction of a SMALL record set (a reduced frame: 150 x 84 = 12,600 records) in ONE launch of one workgroup: 1,024
// records per round -- rank inside the wave from a ballot, the 16 wave totals through LDS, a running base --, the surviving records and
// their count written straight to the destination (the ticket's page-locked block).  The general path is three launches (count, scan,
// scatter) plus the copy to the block: four dependent launches of ~4 us each for work that fits one CU.
__global__ __launch_bounds__(1024) void compact_small_kernel(const float4* __restrict__ in, const uint8_t* __restrict__ mask, uint32_t n,
                                                             float4* __restrict__ out, uint32_t* __restrict__ out_count) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
        const uint32_t i = i0 + tid;
        const bool keep = i < n && mask[i];
        float4 v;
        if (keep) v = in[i];
        const unsigned long long bal = __ballot(keep);
        const uint32_t rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = base_sh;
        for (int k = 0; k < wave; ++k) off += wtot[k];
        if (keep) out[off + rank] = v;
        __syncthreads();
        if (tid == 0) { uint32_t t = 0; for (int k = 0; k < 16; ++k) t += wtot[k]; base_sh += t; }
        __syncthreads();
    }
    if (tid == 0) *out_count = base_sh;
}

// records of the unmasked pixels, in order, and their count straight to (d_out, d_count) -- device memory or the device address of a page-locked block
int compact_small_device(ofps_hip_ctx* ctx, const float4* d_in, const uint8_t* d_mask, size_t n, float4* d_out, uint32_t* d_count) {
    OFPS_REQUIRE(ctx, n >= 1 && n <= kCompactSmallMax, "compact_small: %zu records", n);
    hipLaunchKernelGGL(compact_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_in, d_mask, (uint32_t)n, d_out, d_count);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

int contrast_mask_device(ofps_hip_ctx* ctx, const uint8_t* d_gray, int W, int H, int stride, uint8_t* d_mask, hipStream_t st) {
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && stride >= W, "contrast_mask: bad geometry W=%d H=%d stride=%d", W, H, stride);
    hipLaunchKernelGGL(contrast_mask_kernel, dim3((W + MT_W - 1) / MT_W, (H + MT_H - 1) / MT_H), dim3(256), 0, st ? st : ctx->stream,
                       d_gray, W, H, stride, d_mask);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

// d_out may not alias d_in.  d_count: one uint32 on the device.  Uses S_WORK0 for the tile table.
int compact_entries_device(ofps_hip_ctx* ctx, const float4* d_in, const uint8_t* d_mask, size_t n, float4* d_out,
                           uint32_t* d_count) {
    OFPS_REQUIRE(ctx, n < (1ull << 31), "compact: too many records");
    hipStream_t s = ctx->stream;
    if (n == 0) {
        OFPS_HIP_TRY(ctx, hipMemsetAsync(d_count, 0, sizeof(uint32_t), s));
        return OFPS_HIP_OK;
    }
    const int ntiles = (int)((n + CT - 1) / CT);
    auto* tiles = static_cast<uint32_t*>(scratch(ctx, S_WORK0, (size_t)ntiles * sizeof(uint32_t)));
    if (!tiles) return OFPS_HIP_ENOMEM;
    hipLaunchKernelGGL(compact_count_kernel, dim3(ntiles), dim3(256), 0, s, d_mask, n, tiles);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, tiles, ntiles, d_count);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(ntiles), dim3(256), 0, s, d_in, d_mask, n, tiles, d_out);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

}  // namespace ofps

extern "C" {

int ofps_hip_contrast_mask_dev(ofps_hip_ctx* ctx, const void* d_gray, int W, int H, int stride, void* d_out_mask) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_gray && d_out_mask, "contrast_mask_dev: null device pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ofps::contrast_mask_device(ctx, static_cast<const uint8_t*>(d_gray), W, H, stride, static_cast<uint8_t*>(d_out_mask));
}

int ofps_hip_contrast_mask(ofps_hip_ctx* ctx, const uint8_t* gray, int W, int H, int stride, uint8_t* out_mask) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, gray && out_mask, "contrast_mask: null host pointer");
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && stride >= W, "contrast_mask: bad geometry W=%d H=%d stride=%d", W, H, stride);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)W * H;
    auto* d_gray = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_FRAMES, px));
    auto* d_mask = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_MASK, px));
    if (!d_gray || !d_mask) return OFPS_HIP_ENOMEM;
    OFPS_HIP_TRY(ctx, ofps::upload_rows(d_gray, W, gray, stride, W, H, ctx->stream));
    int rc = ofps::contrast_mask_device(ctx, d_gray, W, H, W, d_mask);
    if (rc != OFPS_HIP_OK) return rc;
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_mask, d_mask, px, hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

}  // extern "C"
