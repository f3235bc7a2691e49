// lk.hip -- N2: dense per-pixel flow, pyramidal Lucas-Kanade ("hip_lk" Decoder).
//
// The reference has no per-pixel flow of its own: cv-decoder calls OpenCV's calcOpticalFlowFarneback
// (cv-decoder/src/lib.rs:188-199) and only converts the result to MotionEntry records (:239-291).  This file
// therefore implements a build-defined algorithm (spec: oracle/ofps_oracle.c:orc_lk_flow, DESIGN.md "N2") whose
// OUTPUT CONVENTION is cv-decoder's: prev(x,y) ~ cur(x+u,y+v); pos = ((x+.5)/W,(y+.5)/H), motion = flow/(W,H).
// "Parity unpinned" w.r.t. the reference; bit-exact vs the build's own CPU restatement: every thread owns one
// pixel and runs the oracle's loops in the oracle's order (f32, no FMA contraction, IEEE divide).
//
// Launches per frame pair (radius 2 / 4 / 6, the tiled path): ONE for levels 1 and 2 of both pyramids ([1 4 6 4 1]/16 x [1 4 6 4 1]/16
// down-sampling straight from the u8 frames, lk_pyr12_kernel; further levels one launch each) and ONE for the whole pyramid of
// Gauss-Newton steps (lk_levels_kernel): a 32 x 8 tile of a level is a workgroup that keeps the previous frame's window (I and its
// central-difference gradients, made from the staged window), the flow and the 2x2 structure tensor G (summed by the level's first
// step: it does not depend on the flow) on chip for all `iters` steps (b = sum grad * (I - J(q + flow)), flow += G^-1 b), and starts
// when its parent tile of the next coarser level -- a workgroup of the same launch -- has published its flows.  Other radii take
// the plain per-step kernels further down.  All of it is window/stencil work: LDS- and VALU-bound on the bilinear sampling -- no
// contraction wide enough for MFMA (the normal equations are 2x2 per pixel).
#include "common.hpp"

#include <cstring>
#include <type_traits>
#include <vector>

namespace ofps {

__device__ __forceinline__ int lk_clampi(int v, int lo, int hi) { return max(lo, min(v, hi)); }   // lo <= hi everywhere: v_max_i32 + v_min_i32 (or one v_med3_i32)

// The arithmetic of one window tap, in ONE place (spec: oracle/ofps_oracle.c:orc_lk_flow, DESIGN.md "N2").
// Revision 2 of the build-defined spec fuses the multiply-adds of the bilinear sample and of the two residual sums --
// lerp(a, b, t) = fma(t, b - a, a); b += g * d as fma(g, d, b) -- 7 operations per tap instead of 11, each result rounded
// once instead of twice.  OFPS_LK_SPEC_FMA = 0 rebuilds revision 1 (separate multiply and add) for A/B runs; the oracle
// (oracle/ofps_oracle.c) carries the same switch and the two must be built alike.
#ifndef OFPS_LK_SPEC_FMA
#define OFPS_LK_SPEC_FMA 1
#endif
#ifndef OFPS_LK_JS
#define OFPS_LK_JS 64            // row pitch of the staged current-frame rectangle, floats
#endif
__device__ __forceinline__ float lk_lerp(float a, float b, float t) {
#if OFPS_LK_SPEC_FMA
    return __builtin_fmaf(t, b - a, a);
#else
    return a + t * (b - a);
#endif
}
__device__ __forceinline__ void lk_accum(float g, float d, float& b) {
#if OFPS_LK_SPEC_FMA
    b = __builtin_fmaf(g, d, b);
#else
    b += g * d;
#endif
}

// wave-wide integer min / max on the DPP data path (cross-lane operands of ordinary VALU instructions; __shfl_xor is
// six dependent ds_bpermute_b32 per value).  Result is wave-uniform (read from lane 63).
template <bool MAX, int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int lk_dpp_minmax(int x) {
    const int moved = __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xF, false);     // unwritten lanes keep their own value
    return MAX ? max(x, moved) : min(x, moved);
}
template <bool MAX>
__device__ __forceinline__ int lk_wave_minmax(int x) {
    x = lk_dpp_minmax<MAX, 0xB1>(x);                 // quad_perm [1,0,3,2]
    x = lk_dpp_minmax<MAX, 0x4E>(x);                 // quad_perm [2,3,0,1]
    x = lk_dpp_minmax<MAX, 0x141>(x);                // row_half_mirror
    x = lk_dpp_minmax<MAX, 0x140>(x);                // row_mirror
    x = lk_dpp_minmax<MAX, 0x142, 0xA>(x);           // row_bcast15 into rows 1, 3
    x = lk_dpp_minmax<MAX, 0x143, 0xC>(x);           // row_bcast31 into rows 2, 3
    return __builtin_amdgcn_readlane(x, 63);
}

// The box of a workgroup's sample origins: two minima and two maxima at once, one DPP instruction per value and step (the
// intrinsic form above is a v_mov_dpp + v_min per step, and hipcc runs the four trees one after the other with wait states
// in between); the four chains interleave, so every DPP operand was written three instructions earlier.  Results are
// wave-uniform (lane 63).
__device__ __forceinline__ void lk_wave_box(int& mn0, int& mx0, int& mn1, int& mx1) {
#define LK_BOX_STEP(CTRL)                                   \
    "v_min_i32_dpp %0, %0, %0 " CTRL "\n\t"                  \
    "v_max_i32_dpp %1, %1, %1 " CTRL "\n\t"                  \
    "v_min_i32_dpp %2, %2, %2 " CTRL "\n\t"                  \
    "v_max_i32_dpp %3, %3, %3 " CTRL "\n\t"
    asm volatile(
        "s_nop 1\n\t"
        LK_BOX_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        LK_BOX_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        LK_BOX_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
        LK_BOX_STEP("row_mirror row_mask:0xf bank_mask:0xf")
        LK_BOX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        LK_BOX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(mn0), "+v"(mx0), "+v"(mn1), "+v"(mx1));
#undef LK_BOX_STEP
    mn0 = __builtin_amdgcn_readlane(mn0, 63); mx0 = __builtin_amdgcn_readlane(mx0, 63);
    mn1 = __builtin_amdgcn_readlane(mn1, 63); mx1 = __builtin_amdgcn_readlane(mx1, 63);
}

// XCD-aware workgroup -> tile mapping for the tiled kernels: a launch is a 1-D grid padded to a multiple of 8 workgroups;
// workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD takes a contiguous run of tiles in raster order, so
// vertically adjacent tiles -- which share 2R of their 4 + 2R window rows -- meet in one L2 instead of eight.
__device__ __forceinline__ bool lk_tile_of_block(int tiles_x, int ntiles, int& tx, int& ty) {
    const int per_xcd = (int)gridDim.x / 8;
    const int t = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    if (t >= ntiles) return false;
    ty = t / tiles_x; tx = t - ty * tiles_x;
    return true;
}

// One launch over ALL pyramid levels (gradients, structure tensors: they depend on the previous frame only, not on the
// flow): level l owns blocks [start[l], start[l + 1]), each range a multiple of 8 blocks so that a block's XCD (block % 8)
// is the same seen from the launch and from its level, and lk_tile_of_block's mapping applies inside the range.
struct LkPyr {
    int levels;
    int w[8], h[8];
    unsigned off[8];          // level's offset inside a pyramid buffer, in pixels
    unsigned start[9];
};
__device__ __forceinline__ bool lk_level_tile_of_block(const LkPyr& P, int tile_w, int tile_h, int& l, int& tx, int& ty) {
    const unsigned b = blockIdx.x;
    l = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) l += (k < P.levels && b >= P.start[k]) ? 1 : 0;
    const unsigned local = b - P.start[l], nloc = P.start[l + 1] - P.start[l];
    const int t = (int)((local % 8u) * (nloc / 8u) + local / 8u);
    const int tiles_x = (P.w[l] + tile_w - 1) / tile_w, ntiles = tiles_x * ((P.h[l] + tile_h - 1) / tile_h);
    if (t >= ntiles) return false;
    ty = t / tiles_x; tx = t - ty * tiles_x;
    return true;
}

// One pyramid step, both frames (blockIdx.z), tiled: a workgroup produces 32 x 8 pixels of level l+1 from the 68 x 20
// level-l window it stages in LDS (coordinates clamped per element, which is what the oracle's per-tap clamping amounts
// to) with the oracle's two separable passes (lk_pyr_down: rows, then columns, each ((((a + 4b) + 6c) + 4d) + e) / 16,
// same operation order, same bits).  The first step reads the u8 frames directly and writes the 64 x 16 level-0 pixels at
// the window's centre to the f32 planes on the way: the u8 -> f32 launch and its 17 MB read-back go.
// TIn = uint8_t: the frames (f0 / f1 receive the f32 planes); TIn = float: a pyramid level (f0 = f1 = nullptr, stride = W).
constexpr int kP0X = 32, kP0Y = 8;
template <typename TIn>
__global__ __launch_bounds__(256) void lk_pyr0_kernel(const TIn* __restrict__ s0, const TIn* __restrict__ s1, int W, int H, int stride,
                                                      float* __restrict__ f0, float* __restrict__ f1, float* __restrict__ o0,
                                                      float* __restrict__ o1, int w1, int h1, float* __restrict__ gx0p,
                                                      float* __restrict__ gy0p) {
    constexpr int RW = 2 * kP0X + 4, RH = 2 * kP0Y + 4;           // level-0 window
    __shared__ float win[RH][RW + 1];
    __shared__ float hor[RH][kP0X + 1];
    int tx, ty;
    if (!lk_tile_of_block((w1 + kP0X - 1) / kP0X, ((w1 + kP0X - 1) / kP0X) * ((h1 + kP0Y - 1) / kP0Y), tx, ty)) return;
    const TIn* src = blockIdx.z ? s1 : s0;
    float* f = blockIdx.z ? f1 : f0;
    float* out = blockIdx.z ? o1 : o0;
    const int x0 = tx * kP0X, y0 = ty * kP0Y;                    // level-1 tile origin
    const int gx0 = 2 * x0 - 2, gy0 = 2 * y0 - 2;                // level-0 window origin
    for (int t = threadIdx.x; t < RW * RH; t += 256) {
        const int r = t / RW, c = t - r * RW;
        const int gx = gx0 + c, gy = gy0 + r;
        const float v = (float)src[(size_t)lk_clampi(gy, 0, H - 1) * stride + lk_clampi(gx, 0, W - 1)];
        win[r][c] = v;
        // the window's centre is this tile's share of the level-0 plane (windows overlap only in their 2-pixel rims)
        if (f && c >= 2 && c < RW - 2 && r >= 2 && r < RH - 2 && gx < W && gy < H) f[(size_t)gy * W + gx] = v;
    }
    __syncthreads();
    // the previous frame's level-0 gradients from the same window (an element holds I at its clamped coordinates, so its
    // neighbours are lk_grad's clamped taps): the gradient launch then only has the coarser levels left
    if (blockIdx.z == 0 && gx0p) {
        for (int t = threadIdx.x; t < 2 * kP0X * 2 * kP0Y; t += 256) {
            const int r = 2 + t / (2 * kP0X), c = 2 + t % (2 * kP0X);
            const int gx = gx0 + c, gy = gy0 + r;
            if (gx < W && gy < H) {
                gx0p[(size_t)gy * W + gx] = (win[r][c + 1] - win[r][c - 1]) * 0.5f;
                gy0p[(size_t)gy * W + gx] = (win[r + 1][c] - win[r - 1][c]) * 0.5f;
            }
        }
    }
    for (int t = threadIdx.x; t < kP0X * RH; t += 256) {
        const int r = t / kP0X, x = t - r * kP0X;
        const float* p = &win[r][2 * x];
        hor[r][x] = ((((p[0] + 4.0f * p[1]) + 6.0f * p[2]) + 4.0f * p[3]) + p[4]) * 0.0625f;
    }
    __syncthreads();
    const int lx = threadIdx.x % kP0X, ly = threadIdx.x / kP0X;
    const int x = x0 + lx, y = y0 + ly;
    if (x < w1 && y < h1)
        out[(size_t)y * w1 + x] = ((((hor[2 * ly][lx] + 4.0f * hor[2 * ly + 1][lx]) + 6.0f * hor[2 * ly + 2][lx]) + 4.0f * hor[2 * ly + 3][lx]) +
                                   hor[2 * ly + 4][lx]) * 0.0625f;
}

// Levels 1 AND 2 of both frames' pyramids from the u8 frames in one launch (round 4: the two pyramid steps used to be two
// launches, 10 + 6 us, the second one latency-bound).  A workgroup owns a 32 x 8 tile of level 2 = 64 x 16 of level 1; it
// stages the 140 x 44 level-0 window those need, makes the 68 x 20 level-1 window (its own 64 x 16 and the 2-pixel rim the
// level-2 filter reads: a quarter more level-1 values than the tile owns) and from it the level-2 tile.  Every value is the
// oracle's lk_pyr_down expression on the oracle's operands (coordinates clamped per element: a window element at level-1
// coordinate x holds level1[clamp(x)], made from level-0 taps clamp(2 clamp(x) + d)) -- same bits whoever computes it.
__global__ __launch_bounds__(256) void lk_pyr12_kernel(const uint8_t* __restrict__ s0, const uint8_t* __restrict__ s1, int W, int H, int stride,
                                                       float* __restrict__ a1, float* __restrict__ b1, int w1, int h1,
                                                       float* __restrict__ a2, float* __restrict__ b2, int w2, int h2) {
    constexpr int W1 = 2 * kP0X + 4, H1 = 2 * kP0Y + 4;           // level-1 window: 68 x 20
    constexpr int W0 = 2 * W1 + 4, H0 = 2 * H1 + 4;               // level-0 window: 140 x 44
    __shared__ alignas(16) uint8_t win0[H0][W0 + 4];
    __shared__ float hor1[H0][W1 + 1];
    __shared__ float lvl1[H1][W1 + 1];
    __shared__ float hor2[H1][kP0X + 1];
    int tx, ty;
    if (!lk_tile_of_block((w2 + kP0X - 1) / kP0X, ((w2 + kP0X - 1) / kP0X) * ((h2 + kP0Y - 1) / kP0Y), tx, ty)) return;
    const uint8_t* src = blockIdx.z ? s1 : s0;
    float* o1 = blockIdx.z ? b1 : a1;
    float* o2 = blockIdx.z ? b2 : a2;
    const int x2_0 = tx * kP0X, y2_0 = ty * kP0Y;                 // level-2 tile origin
    const int x1_0 = 2 * x2_0 - 2, y1_0 = 2 * y2_0 - 2;           // level-1 window origin
    const int gx0 = 2 * x1_0 - 2, gy0 = 2 * y1_0 - 2;             // level-0 window origin
    // interior tiles of aligned frames: the window's rows as dwords.  gx0 = 4 x2_0 - 6 sits two bytes behind a 4-byte boundary, so
    // the row is loaded from gx0 - 2: (W0 + 4) / 4 = 36 dwords, 6 loads per thread for the window instead of 24 clamped byte loads
    // (the window then starts at column `lead` = 2 of win0's rows).  Uniform branch; same bytes either way.
    static_assert((W0 + 4) % 4 == 0, "window row is a whole number of dwords");
    const bool vec = gx0 >= 2 && gy0 >= 0 && gx0 + W0 + 2 <= W && gy0 + H0 <= H && (stride & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0;
    const int lead = vec ? 2 : 0;
    if (vec) {
        constexpr int Q = (W0 + 4) / 4;
        for (int t = threadIdx.x; t < Q * H0; t += 256) {
            const int r = t / Q, q = t - r * Q;
            *reinterpret_cast<uint32_t*>(&win0[r][4 * q]) = *reinterpret_cast<const uint32_t*>(src + (size_t)(gy0 + r) * stride + (gx0 - 2) + 4 * q);
        }
    } else {
        for (int t = threadIdx.x; t < W0 * H0; t += 256) {
            const int r = t / W0, c = t - r * W0;
            win0[r][c] = src[(size_t)lk_clampi(gy0 + r, 0, H - 1) * stride + lk_clampi(gx0 + c, 0, W - 1)];
        }
    }
    __syncthreads();
    // rows pass of level 1: every level-0 row of the window, at the window's 68 (clamped) level-1 columns
    for (int t = threadIdx.x; t < W1 * H0; t += 256) {
        const int r = t / W1, i = t - r * W1;
        const int cx = lk_clampi(x1_0 + i, 0, w1 - 1);
        const uint8_t* p = &win0[r][2 * cx - gx0 - 2 + lead];
        hor1[r][i] = ((((float)p[0] + 4.0f * (float)p[1]) + 6.0f * (float)p[2]) + 4.0f * (float)p[3]) + (float)p[4];
        hor1[r][i] *= 0.0625f;
    }
    __syncthreads();
    // columns pass: the 68 x 20 level-1 window; its centre is this tile's share of the level-1 plane
    for (int t = threadIdx.x; t < W1 * H1; t += 256) {
        const int j = t / W1, i = t - j * W1;
        const int cy = lk_clampi(y1_0 + j, 0, h1 - 1);
        const int r = 2 * cy - gy0 - 2;
        const float v = ((((hor1[r][i] + 4.0f * hor1[r + 1][i]) + 6.0f * hor1[r + 2][i]) + 4.0f * hor1[r + 3][i]) + hor1[r + 4][i]) * 0.0625f;
        lvl1[j][i] = v;
        const int x1 = x1_0 + i, y1 = y1_0 + j;
        if (i >= 2 && i < W1 - 2 && j >= 2 && j < H1 - 2 && x1 < w1 && y1 < h1) o1[(size_t)y1 * w1 + x1] = v;
    }
    __syncthreads();
    // level 2 from the level-1 window (whose elements sit at their clamped coordinates already)
    for (int t = threadIdx.x; t < kP0X * H1; t += 256) {
        const int j = t / kP0X, x = t - j * kP0X;
        const float* p = &lvl1[j][2 * x];
        hor2[j][x] = ((((p[0] + 4.0f * p[1]) + 6.0f * p[2]) + 4.0f * p[3]) + p[4]) * 0.0625f;
    }
    __syncthreads();
    const int lx = threadIdx.x % kP0X, ly = threadIdx.x / kP0X;
    const int x2 = x2_0 + lx, y2 = y2_0 + ly;
    if (x2 < w2 && y2 < h2)
        o2[(size_t)y2 * w2 + x2] = ((((hor2[2 * ly][lx] + 4.0f * hor2[2 * ly + 1][lx]) + 6.0f * hor2[2 * ly + 2][lx]) + 4.0f * hor2[2 * ly + 3][lx]) +
                                    hor2[2 * ly + 4][lx]) * 0.0625f;
}

__global__ __launch_bounds__(256) void lk_u8_to_f32_pair_kernel(const uint8_t* __restrict__ s0, const uint8_t* __restrict__ s1, int W, int H,
                                                                int stride, float* __restrict__ d0, float* __restrict__ d1) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const uint8_t* src = blockIdx.z ? s1 : s0;
    float* dst = blockIdx.z ? d1 : d0;
    dst[(size_t)y * W + x] = (float)src[(size_t)y * stride + x];
}

// central-difference gradients of the previous frame, every level of the pyramid in one launch (64 x 4 pixels per block)
__global__ __launch_bounds__(256) void lk_grad_all_kernel(const float* __restrict__ Ip, const LkPyr P, float* __restrict__ gxp,
                                                          float* __restrict__ gyp) {
    int l, tx, ty;
    if (!lk_level_tile_of_block(P, 64, 4, l, tx, ty)) return;
    const int w = P.w[l], h = P.h[l];
    const int x = tx * 64 + (threadIdx.x & 63), y = ty * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float* I = Ip + P.off[l];
    float* gx = gxp + P.off[l];
    float* gy = gyp + P.off[l];
    gx[(size_t)y * w + x] = (I[(size_t)y * w + lk_clampi(x + 1, 0, w - 1)] - I[(size_t)y * w + lk_clampi(x - 1, 0, w - 1)]) * 0.5f;
    gy[(size_t)y * w + x] = (I[(size_t)lk_clampi(y + 1, 0, h - 1) * w + x] - I[(size_t)lk_clampi(y - 1, 0, h - 1) * w + x]) * 0.5f;
}

// flow_l(x,y) = 2 * flow_{l+1}(x/2, y/2)
__global__ __launch_bounds__(256) void lk_upsample_kernel(const float2* __restrict__ coarse, int w1, int h1, float2* __restrict__ fine,
                                                          int w, int h) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float2 c = coarse[(size_t)lk_clampi(y / 2, 0, h1 - 1) * w1 + lk_clampi(x / 2, 0, w1 - 1)];
    fine[(size_t)y * w + x] = make_float2(2.0f * c.x, 2.0f * c.y);
}

// structure tensor, window sums in the oracle's order (dy outer, dx inner, clamped coordinates)
__global__ __launch_bounds__(256) void lk_tensor_kernel(const float* __restrict__ gx, const float* __restrict__ gy, int w, int h,
                                                        int radius, float4* __restrict__ G) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    float gxx = 0.0f, gxy = 0.0f, gyy = 0.0f;
    for (int dy = -radius; dy <= radius; ++dy) {
        const size_t row = (size_t)lk_clampi(y + dy, 0, h - 1) * w;
        for (int dx = -radius; dx <= radius; ++dx) {
            const int qx = lk_clampi(x + dx, 0, w - 1);
            const float ix = gx[row + qx], iy = gy[row + qx];
            lk_accum(ix, ix, gxx); lk_accum(ix, iy, gxy); lk_accum(iy, iy, gyy);
        }
    }
    G[(size_t)y * w + x] = make_float4(gxx, gxy, gyy, 0.0f);
}

__device__ __forceinline__ float lk_bilinear(const float* __restrict__ J, int w, int h, float fx, float fy) {
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float ax = fx - x0f, ay = fy - y0f;
    const float cx = x0f < -1.0f ? -1.0f : (x0f > (float)w ? (float)w : x0f);
    const float cy = y0f < -1.0f ? -1.0f : (y0f > (float)h ? (float)h : y0f);
    const int x0 = (int)cx, y0 = (int)cy;
    const int xa = lk_clampi(x0, 0, w - 1), xb = lk_clampi(x0 + 1, 0, w - 1);
    const int ya = lk_clampi(y0, 0, h - 1), yb = lk_clampi(y0 + 1, 0, h - 1);
    const float j00 = J[(size_t)ya * w + xa], j10 = J[(size_t)ya * w + xb], j01 = J[(size_t)yb * w + xa], j11 = J[(size_t)yb * w + xb];
    const float top = lk_lerp(j00, j10, ax);
    const float bot = lk_lerp(j01, j11, ax);
    return lk_lerp(top, bot, ay);
}

// One Gauss-Newton step for every pixel.  RADIUS > 0: compile-time window; the per-sample floor/clamp of the
// oracle's bilinear fetch depends only on the window column (x side) or row (y side), so it is hoisted into
// 2r+1 column records and one row record -- the same operations on the same inputs, hence the same bits.
// RADIUS == 0: run-time radius, the plain per-sample form.
template <int RADIUS>
__global__ __launch_bounds__(256) void lk_step_kernel(const float* __restrict__ I, const float* __restrict__ J,
                                                      const float* __restrict__ gx, const float* __restrict__ gy,
                                                      const float4* __restrict__ G, int w, int h, int radius,
                                                      const float2* __restrict__ flow_in, float2* __restrict__ flow_out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float2 f = flow_in[(size_t)y * w + x];
    float bx = 0.0f, by = 0.0f;
    if constexpr (RADIUS > 0) {
        constexpr int N = 2 * RADIUS + 1;
        int qx[N], xa[N], xb[N];
        float ax[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            qx[k] = lk_clampi(x + k - RADIUS, 0, w - 1);
            const float fx = (float)qx[k] + f.x;
            const float x0f = floorf(fx);
            ax[k] = fx - x0f;
            const float cx = x0f < -1.0f ? -1.0f : (x0f > (float)w ? (float)w : x0f);
            const int x0 = (int)cx;
            xa[k] = lk_clampi(x0, 0, w - 1);
            xb[k] = lk_clampi(x0 + 1, 0, w - 1);
        }
#pragma unroll 1
        for (int dy = -RADIUS; dy <= RADIUS; ++dy) {
            const int qy = lk_clampi(y + dy, 0, h - 1);
            const float fy = (float)qy + f.y;
            const float y0f = floorf(fy);
            const float ay = fy - y0f;
            const float cy = y0f < -1.0f ? -1.0f : (y0f > (float)h ? (float)h : y0f);
            const int y0 = (int)cy;
            const float* ra = J + (size_t)lk_clampi(y0, 0, h - 1) * w;
            const float* rb = J + (size_t)lk_clampi(y0 + 1, 0, h - 1) * w;
            const size_t row = (size_t)qy * w;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float j00 = ra[xa[k]], j10 = ra[xb[k]], j01 = rb[xa[k]], j11 = rb[xb[k]];
                const float top = lk_lerp(j00, j10, ax[k]);
                const float bot = lk_lerp(j01, j11, ax[k]);
                const float d = I[row + qx[k]] - lk_lerp(top, bot, ay);
                lk_accum(gx[row + qx[k]], d, bx);
                lk_accum(gy[row + qx[k]], d, by);
            }
        }
    } else {
        for (int dy = -radius; dy <= radius; ++dy) {
            const int qy = lk_clampi(y + dy, 0, h - 1);
            const size_t row = (size_t)qy * w;
            for (int dx = -radius; dx <= radius; ++dx) {
                const int qx = lk_clampi(x + dx, 0, w - 1);
                const float d = I[row + qx] - lk_bilinear(J, w, h, (float)qx + f.x, (float)qy + f.y);
                lk_accum(gx[row + qx], d, bx);
                lk_accum(gy[row + qx], d, by);
            }
        }
    }
    const float4 g = G[(size_t)y * w + x];
    const float det = g.x * g.z - g.y * g.y;
    float du = 0.0f, dv = 0.0f;
    if (det > 0.01f) {
        du = (g.z * bx - g.y * by) / det;
        dv = (g.x * by - g.y * bx) / det;
    }
    flow_out[(size_t)y * w + x] = make_float2(f.x + du, f.y + dv);
}

// ---- tiled variants (compile-time radius).  A workgroup owns kTX x kTY = 32 x 8 pixels; the previous frame's window (I and
// its gradients, one 16-byte record per element) is staged once in LDS with the oracle's coordinate clamping applied at
// staging time, so tile[ly+dy+R][lx+dx+R] is exactly (I, gx, gy)[clamp(y+dy)][clamp(x+dx)]; the current frame's sample rectangle
// is staged the same way, and inside a window the horizontal interpolation of a sample row serves two window rows (the
// bottom row of one is the top row of the next whenever the rows are consecutive: almost always; the rare exceptions
// recompute).  Same values, same operation order as the untiled kernels, hence the same bits.

// pixels per workgroup of the tiled kernels (kTX * kTY = 256 threads; a wave covers 64 / kTX tile rows).  Measured at
// 1080p, radius 4: 64 x 4 0.385 ms (0.58 on +-16 px region jumps), 32 x 8 0.38 (0.535), 16 x 16 0.43 (0.55): the squarer
// tile stages fewer window elements per pixel (2.5 vs 3.4) and its rectangle tolerates wilder flows; 16 lanes per row
// lose on the 16-byte staging rows.
constexpr int kTX = 32, kTY = 8;
constexpr int kJMargin = 2;          // pixels of slack staged around the current frame's rectangle (lk_level_body)
// Pixels per thread (round 4, radius 4): a lane owns two vertically adjacent pixels, the tile is kTX x 2 kTY.  The second
// pixel's window row r holds the records of the first one's row r + 1, so the ten record rows under the pair are read from LDS
// once: 90 ds_read_b128 per pair and step instead of 162 -- the step's time follows the LDS cycles more than anything else
// (profiles/r04/lk_lds_experiments.txt).  OFPS_LK_PP = 1 rebuilds the one-pixel kernel for A/B runs.
#ifndef OFPS_LK_PP
#define OFPS_LK_PP 1
#endif
template <int RADIUS>
constexpr int kLkPP = (RADIUS == 4 && OFPS_LK_SPEC_FMA) ? OFPS_LK_PP : 1;
static_assert(OFPS_LK_PP == 1 || OFPS_LK_PP == 2, "one or two pixels per thread");

template <int RADIUS>
struct LkTile {
    static constexpr int R = RADIUS, N = 2 * RADIUS + 1, PP = kLkPP<RADIUS>, TY = kTY * PP, TW = kTX + 2 * RADIUS, TH = TY + 2 * RADIUS;
};
// rows of a tile of the tiled kernels at a run-time radius (host side: tile counts, flags)
static inline int lk_tile_rows(int radius) { return radius == 4 ? LkTile<4>::TY : kTY; }

// One whole 16-byte LDS read (ds_read_b128: 4 LDS cycles per wave).  Left to itself the compiler narrows a float4 read
// whose .w is unused to ds_read_b96, which takes 8; an empty asm statement that "uses" .w keeps the read whole without
// making it volatile (a volatile read is issued right before its use and waited for on the spot: nine exposed LDS
// latencies per window row).
typedef float lk_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lk_f4 lk_lds_read4(const float4* p) {
    const float4 v = *p;
    asm volatile("" :: "v"(v.w));
    lk_f4 r = {v.x, v.y, v.z, v.w};
    return r;
}

// Level 0 straight from the u8 frames (round 3).  The previous frame's window records (I, gx, gy) are made from ONE u8
// window with a 1-pixel rim -- 42 x 18 bytes instead of three f32 planes' 40 x 16 x 12 bytes -- in two phases through an
// LDS scratch: (A) the window as f32 at clamped image coordinates, (B) per tile element its value and central differences
// at ITS clamped coordinates (the expressions lk_pyr0_kernel used to write to the level-0 gradient planes: same operands,
// same bits).  (float)u8 is exact, so are all the values downstream.  `scratch` needs (TH + 2) * (TW + 2) floats (the
// level kernel lends its rectangle buffer, which is not in use yet).  Ends with the barrier phase B's readers need.
template <int RADIUS>
struct LkU8Window {                                   // the f32 copy of the u8 window in LDS: TH + 2 rows of WP floats
    using T = LkTile<RADIUS>;
    static constexpr int WH = T::TH + 2;
    static constexpr int WP = (T::TW + 2 + 3 + 3) / 4 * 4;          // room for a dword-aligned start (up to 3 pixels early)
    static constexpr int FLOATS = WH * WP;
};
// in three parts, so that a caller can put work between the request and the use of the window's bytes: `issue` requests them
// (registers), `spill` writes them to the scratch as f32 (a barrier must follow), `records` makes the tile records from the
// scratch (a barrier must follow before the tile is read).
template <int RADIUS, typename TIn = uint8_t>
struct LkU8Regs {
    using T = LkTile<RADIUS>;
    using U = LkU8Window<RADIUS>;
    // aligned form: one request = 4 pixels (a dword of u8 / a float4); element form: one pixel
    static constexpr int NQ = (U::WP / 4 * U::WH + 255) / 256;              // requests per thread, aligned form
    static constexpr int NB = ((T::TW + 2) * U::WH + 255) / 256;           // pixels per thread, element form
    using Quad = std::conditional_t<std::is_same_v<TIn, uint8_t>, uint32_t, float4>;
    bool vec;
    int ox;
    Quad q[NQ];
    TIn b[NB];
};
// The window's source is the u8 frame (level 0) or an f32 pyramid plane (levels >= 1, round 4: the f32 levels used to stage
// three planes -- I and its two gradient planes, written by the pyramid / gradient launches -- and now make the gradients from
// the I window like level 0 does: no gradient planes, no gradient launch, a third of the staging bytes).
template <int RADIUS, typename TIn>
__device__ __forceinline__ void lk_stage3_u8_issue(const TIn* __restrict__ src, int stride, int w, int h, int x0, int y0, LkU8Regs<RADIUS, TIn>& g) {
    using T = LkTile<RADIUS>;
    using U = LkU8Window<RADIUS>;
    constexpr int WW = T::TW + 2, WH = U::WH, WP = U::WP;
    constexpr bool kU8 = std::is_same_v<TIn, uint8_t>;
    int ox = x0 - T::R - 1;                                           // image x of scratch column 0
    const int oy = y0 - T::R - 1;
    // interior tiles of aligned sources: the window starts at the 4-pixel boundary at or before its first pixel and is
    // loaded four pixels per request (one request per thread for the whole window at radius 4); everything else pixel by
    // pixel with the coordinates clamped per element.  Uniform branch.
    g.vec = ox >= 0 && oy >= 0 && oy + WH <= h && (ox & ~3) + WP <= w && (stride & 3) == 0 &&
            (reinterpret_cast<uintptr_t>(src) & (kU8 ? 3 : 15)) == 0;
    if (g.vec) {
        ox &= ~3;
        constexpr int Q = WP / 4;
#pragma unroll
        for (int k = 0; k < LkU8Regs<RADIUS, TIn>::NQ; ++k) {
            const int t = threadIdx.x + 256 * k;
            if (t < Q * WH) {
                const int r = t / Q, c4 = t - r * Q;
                g.q[k] = *reinterpret_cast<const typename LkU8Regs<RADIUS, TIn>::Quad*>(src + (size_t)(oy + r) * stride + ox + 4 * c4);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < LkU8Regs<RADIUS, TIn>::NB; ++k) {
            const int t = threadIdx.x + 256 * k;
            if (t < WW * WH) {
                const int r = t / WW, c = t - r * WW;
                g.b[k] = src[(size_t)lk_clampi(oy + r, 0, h - 1) * stride + lk_clampi(ox + c, 0, w - 1)];
            }
        }
    }
    g.ox = ox;
}
template <int RADIUS, typename TIn>
__device__ __forceinline__ void lk_stage3_u8_spill(float* scratch, const LkU8Regs<RADIUS, TIn>& g) {
    using T = LkTile<RADIUS>;
    using U = LkU8Window<RADIUS>;
    constexpr int WW = T::TW + 2, WH = U::WH, WP = U::WP;
    if (g.vec) {
        constexpr int Q = WP / 4;
#pragma unroll
        for (int k = 0; k < LkU8Regs<RADIUS, TIn>::NQ; ++k) {
            const int t = threadIdx.x + 256 * k;
            if (t < Q * WH) {
                const int r = t / Q, c4 = t - r * Q;
                if constexpr (std::is_same_v<TIn, uint8_t>) {
                    const uint32_t q = g.q[k];
                    *reinterpret_cast<float4*>(scratch + r * WP + 4 * c4) =
                        make_float4((float)(q & 0xFFu), (float)((q >> 8) & 0xFFu), (float)((q >> 16) & 0xFFu), (float)(q >> 24));
                } else {
                    *reinterpret_cast<float4*>(scratch + r * WP + 4 * c4) = g.q[k];
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < LkU8Regs<RADIUS, TIn>::NB; ++k) {
            const int t = threadIdx.x + 256 * k;
            if (t < WW * WH) { const int r = t / WW, c = t - r * WW; scratch[r * WP + c] = (float)g.b[k]; }
        }
    }
}
template <int RADIUS>
__device__ __forceinline__ void lk_stage3_u8_records(const float* scratch, float4 (*tile)[LkTile<RADIUS>::TW], int ox, bool interior, int w, int h,
                                                     int x0, int y0) {
    using T = LkTile<RADIUS>;
    constexpr int WP = LkU8Window<RADIUS>::WP;
    const int oy = y0 - T::R - 1;
    if (interior) {
        // no coordinate leaves the image: a thread owns column c of rows r0, r0 + RPP, ...; its fifteen-odd scratch reads are
        // independent of each other and issued together (the clamped form below waits on one element after the other)
        constexpr int RPP = 256 / T::TW, PASSES = (T::TH + RPP - 1) / RPP;
        const int c = threadIdx.x % T::TW, r0 = threadIdx.x / T::TW;
        if (r0 >= RPP) return;
        const int ix = x0 - T::R + c - ox;                            // scratch column of tile column c
        float v[PASSES], xl[PASSES], xr[PASSES], yu[PASSES], yd[PASSES];
#pragma unroll
        for (int k = 0; k < PASSES; ++k) {
            const int r = r0 + k * RPP;
            if (r < T::TH) {
                const float* p = scratch + (r + 1) * WP + ix;         // scratch row of tile row r is r + 1 (the rim)
                v[k] = p[0]; xl[k] = p[-1]; xr[k] = p[1]; yu[k] = p[-WP]; yd[k] = p[WP];
            }
        }
#pragma unroll
        for (int k = 0; k < PASSES; ++k) {
            const int r = r0 + k * RPP;
            if (r < T::TH) tile[r][c] = make_float4(v[k], (xr[k] - xl[k]) * 0.5f, (yd[k] - yu[k]) * 0.5f, 0.0f);
        }
        return;
    }
    for (int t = threadIdx.x; t < T::TW * T::TH; t += 256) {
        const int r = t / T::TW, c = t - r * T::TW;
        const int xc = lk_clampi(x0 - T::R + c, 0, w - 1), yc = lk_clampi(y0 - T::R + r, 0, h - 1);
        const int ix = xc - ox, iy = yc - oy;                         // where (xc, yc) sits in the scratch window
        const int il = lk_clampi(xc - 1, 0, w - 1) - ox, ir = lk_clampi(xc + 1, 0, w - 1) - ox;
        const int iu = lk_clampi(yc - 1, 0, h - 1) - oy, id = lk_clampi(yc + 1, 0, h - 1) - oy;
        const float v = scratch[iy * WP + ix];
        const float gxv = (scratch[iy * WP + ir] - scratch[iy * WP + il]) * 0.5f;
        const float gyv = (scratch[id * WP + ix] - scratch[iu * WP + ix]) * 0.5f;
        tile[r][c] = make_float4(v, gxv, gyv, 0.0f);
    }
}
// One window row of the level kernel at radius 4 (9 taps), spec revision 2, hand-scheduled.  hipcc's own code for this loop
// copies the carried interpolation row (8 v_mov per row), re-reads every tile record into the same four registers with a
// full wait in front of each use, and -- asked to unroll by two so that the carried row could change name instead of
// registers -- hoists LDS reads until it spills 100 registers.  Here:
//   * l0..l9 receive the ten texels of sample row yi + 1 and are turned IN PLACE into that row's nine horizontal
//     interpolations (v_fmac: l[k] = ax[k] * (l[k+1] - l[k]) + l[k]); t0..t8 hold the previous sample row's and are
//     consumed in place (t[k] = ay * (l[k] - t[k]) + t[k], the bilinear sample).  The caller alternates two register sets
//     between the roles, so nothing is ever copied;
//   * the tile records (I, gx, gy, -) are double-buffered in two fixed register quads, the read of tap k + 2 issued as
//     soon as tap k's quad is free: one LDS latency is exposed per row instead of nine;
//   * 7 VALU operations per tap (the spec's count), 63 per row.
// Same operations on the same operands in the same order as the C++ form (lk_lerp / lk_accum): same bits.
// The quads are v[72:75] / v[76:79]: the top of the 80-register budget of a 256-thread workgroup at 6 waves per SIMD.
// the two record quads: the top eight registers of the kernel's budget (80 at 6 waves per SIMD; OFPS_LK_WAVES4 = 7: 72)
#ifndef OFPS_LK_WAVES4
#define OFPS_LK_WAVES4 6
#endif
#if OFPS_LK_PP == 2                      // two pixels per thread: 4 waves per SIMD, 128 registers
#define LK_QE "v[120:123]"
#define LK_QE0 "v120"
#define LK_QE1 "v121"
#define LK_QE2 "v122"
#define LK_QO "v[124:127]"
#define LK_QO0 "v124"
#define LK_QO1 "v125"
#define LK_QO2 "v126"
#define LK_Q_CLOBBERS LK_QE0, LK_QE1, LK_QE2, "v123", LK_QO0, LK_QO1, LK_QO2, "v127"
#elif OFPS_LK_WAVES4 >= 7
#define LK_QE "v[64:67]"
#define LK_QE0 "v64"
#define LK_QE1 "v65"
#define LK_QE2 "v66"
#define LK_QO "v[68:71]"
#define LK_QO0 "v68"
#define LK_QO1 "v69"
#define LK_QO2 "v70"
#define LK_Q_CLOBBERS "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"
#else
#define LK_QE "v[72:75]"
#define LK_QE0 "v72"
#define LK_QE1 "v73"
#define LK_QE2 "v74"
#define LK_QO "v[76:79]"
#define LK_QO0 "v76"
#define LK_QO1 "v77"
#define LK_QO2 "v78"
#define LK_Q_CLOBBERS LK_QE0, LK_QE1, LK_QE2, "v75", LK_QO0, LK_QO1, LK_QO2, "v79"
#endif
#define LK_QUADS LK_QE, LK_QE0, LK_QE1, LK_QE2, LK_QO, LK_QO0, LK_QO1, LK_QO2
#define LK_ROWS9_APPLY(M, ...) M(__VA_ARGS__)          // (LK_QUADS expands to the eight arguments before M is invoked)
#define LK_ROW9_TAP(K, KN, WAIT, NEXT)                                        \
    "v_sub_f32 %[tmp], %[l" #KN "], %[l" #K "]\n\t"                             \
    "v_fmac_f32 %[l" #K "], %[a" #K "], %[tmp]\n\t"                             \
    "v_sub_f32 %[tmp], %[l" #K "], %[t" #K "]\n\t"                              \
    "v_fmac_f32 %[t" #K "], %[ay], %[tmp]\n\t"                                  \
    "s_waitcnt lgkmcnt(" #WAIT ")\n\t"                                          \
    NEXT
#define LK_ROW9_USE(K, Q0, Q1, Q2, ISSUE)                                     \
    "v_sub_f32 %[tmp], " Q0 ", %[t" #K "]\n\t"                                 \
    "v_fmac_f32 %[bx], " Q1 ", %[tmp]\n\t"                                     \
    "v_fmac_f32 %[by], " Q2 ", %[tmp]\n\t"                                     \
    ISSUE
// the same with the tap's share of the structure tensor (first step of a level): gxx += gx gx, gxy += gx gy, gyy += gy gy
#define LK_ROW9_USE_G(K, Q0, Q1, Q2, ISSUE)                                   \
    "v_sub_f32 %[tmp], " Q0 ", %[t" #K "]\n\t"                                 \
    "v_fmac_f32 %[gxx], " Q1 ", " Q1 "\n\t"                                    \
    "v_fmac_f32 %[bx], " Q1 ", %[tmp]\n\t"                                     \
    "v_fmac_f32 %[gxy], " Q1 ", " Q2 "\n\t"                                    \
    "v_fmac_f32 %[by], " Q2 ", %[tmp]\n\t"                                     \
    "v_fmac_f32 %[gyy], " Q2 ", " Q2 "\n\t"                                    \
    ISSUE
// r[0..18] are the two register sets: PARITY 0: l = r[0..9], t = r[10..18]; PARITY 1: l = r[10..18] + r[9], t = r[0..8] --
// after a PARITY 0 row the nine interpolations of its lower sample row sit in r[0..8], exactly where a PARITY 1 row
// expects its upper row, and vice versa.
template <int PARITY>
__device__ __forceinline__ void lk_row9_asm(float (&r)[19], const float (&a)[9], float ay, float& bx, float& by, uint32_t jaddr,
                                            uint32_t taddr) {
    constexpr auto L = [](int k) constexpr { return PARITY ? (k < 9 ? 10 + k : 9) : k; };
    constexpr auto T = [](int k) constexpr { return PARITY ? k : 10 + k; };
    float tmp;
    asm volatile(
        "ds_read_b32 %[l0], %[ja]\n\t"
        "ds_read_b32 %[l1], %[ja] offset:4\n\t"
        "ds_read_b32 %[l2], %[ja] offset:8\n\t"
        "ds_read_b32 %[l3], %[ja] offset:12\n\t"
        "ds_read_b32 %[l4], %[ja] offset:16\n\t"
        "ds_read_b32 %[l5], %[ja] offset:20\n\t"
        "ds_read_b32 %[l6], %[ja] offset:24\n\t"
        "ds_read_b32 %[l7], %[ja] offset:28\n\t"
        "ds_read_b32 %[l8], %[ja] offset:32\n\t"
        "ds_read_b32 %[l9], %[ja] offset:36\n\t"
        "ds_read_b128 " LK_QE ", %[ta]\n\t"
        "ds_read_b128 " LK_QO ", %[ta] offset:16\n\t"
        "s_waitcnt lgkmcnt(10)\n\t"                                            // l0, l1 are there
        // tap k: horizontal + vertical interpolation while its tile record is in flight, then the residual sums; the quad
        // it used is refilled with tap k + 2's record.  Waits: 12 reads issued; before tap k's first use of l[k+1] at most
        // 10 - k of the texel reads ... may be outstanding behind the two tile reads (counted below per tap).
        LK_ROW9_TAP(0, 1, 1, "") LK_ROW9_USE(0, LK_QE0, LK_QE1, LK_QE2, "ds_read_b128 " LK_QE ", %[ta] offset:32\n\t")
        LK_ROW9_TAP(1, 2, 1, "") LK_ROW9_USE(1, LK_QO0, LK_QO1, LK_QO2, "ds_read_b128 " LK_QO ", %[ta] offset:48\n\t")
        LK_ROW9_TAP(2, 3, 1, "") LK_ROW9_USE(2, LK_QE0, LK_QE1, LK_QE2, "ds_read_b128 " LK_QE ", %[ta] offset:64\n\t")
        LK_ROW9_TAP(3, 4, 1, "") LK_ROW9_USE(3, LK_QO0, LK_QO1, LK_QO2, "ds_read_b128 " LK_QO ", %[ta] offset:80\n\t")
        LK_ROW9_TAP(4, 5, 1, "") LK_ROW9_USE(4, LK_QE0, LK_QE1, LK_QE2, "ds_read_b128 " LK_QE ", %[ta] offset:96\n\t")
        LK_ROW9_TAP(5, 6, 1, "") LK_ROW9_USE(5, LK_QO0, LK_QO1, LK_QO2, "ds_read_b128 " LK_QO ", %[ta] offset:112\n\t")
        LK_ROW9_TAP(6, 7, 1, "") LK_ROW9_USE(6, LK_QE0, LK_QE1, LK_QE2, "ds_read_b128 " LK_QE ", %[ta] offset:128\n\t")
        LK_ROW9_TAP(7, 8, 1, "") LK_ROW9_USE(7, LK_QO0, LK_QO1, LK_QO2, "")
        LK_ROW9_TAP(8, 9, 0, "") LK_ROW9_USE(8, LK_QE0, LK_QE1, LK_QE2, "")
        : [l0] "=&v"(r[L(0)]), [l1] "=&v"(r[L(1)]), [l2] "=&v"(r[L(2)]), [l3] "=&v"(r[L(3)]), [l4] "=&v"(r[L(4)]), [l5] "=&v"(r[L(5)]),
          [l6] "=&v"(r[L(6)]), [l7] "=&v"(r[L(7)]), [l8] "=&v"(r[L(8)]), [l9] "=&v"(r[L(9)]),
          [t0] "+v"(r[T(0)]), [t1] "+v"(r[T(1)]), [t2] "+v"(r[T(2)]), [t3] "+v"(r[T(3)]), [t4] "+v"(r[T(4)]), [t5] "+v"(r[T(5)]),
          [t6] "+v"(r[T(6)]), [t7] "+v"(r[T(7)]), [t8] "+v"(r[T(8)]), [bx] "+v"(bx), [by] "+v"(by), [tmp] "=&v"(tmp)
        : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),
          [a8] "v"(a[8]), [ay] "v"(ay), [ja] "v"(jaddr), [ta] "v"(taddr)
        : LK_Q_CLOBBERS, "memory");
}
// lk_row9_asm with the structure-tensor sums (first step of a level).
// r[0..18] are the two register sets: PARITY 0: l = r[0..9], t = r[10..18]; PARITY 1: l = r[10..18] + r[9], t = r[0..8] --
// after a PARITY 0 row the nine interpolations of its lower sample row sit in r[0..8], exactly where a PARITY 1 row
// expects its upper row, and vice versa.
template <int PARITY>
__device__ __forceinline__ void lk_row9_asm_g(float (&r)[19], const float (&a)[9], float ay, float& bx, float& by, float& gxx,
                                              float& gxy, float& gyy, uint32_t jaddr, uint32_t taddr) {
    constexpr auto L = [](int k) constexpr { return PARITY ? (k < 9 ? 10 + k : 9) : k; };
    constexpr auto T = [](int k) constexpr { return PARITY ? k : 10 + k; };
    float tmp;
    asm volatile(
        "ds_read_b32 %[l0], %[ja]\n\t"
        "ds_read_b32 %[l1], %[ja] offset:4\n\t"
        "ds_read_b32 %[l2], %[ja] offset:8\n\t"
        "ds_read_b32 %[l3], %[ja] offset:12\n\t"
        "ds_read_b32 %[l4], %[ja] offset:16\n\t"
        "ds_read_b32 %[l5], %[ja] offset:20\n\t"
        "ds_read_b32 %[l6], %[ja] offset:24\n\t"
        "ds_read_b32 %[l7], %[ja] offset:28\n\t"
        "ds_read_b32 %[l8], %[ja] offset:32\n\t"
        "ds_read_b32 %[l9], %[ja] offset:36\n\t"
        "ds_read_b128 " LK_QE ", %[ta]\n\t"
        "ds_read_b128 " LK_QO ", %[ta] offset:16\n\t"
        "s_waitcnt lgkmcnt(10)\n\t"                                            // l0, l1 are there
        // tap k: horizontal + vertical interpolation while its tile record is in flight, then the residual sums; the quad
        // it used is refilled with tap k + 2's record.  Waits: 12 reads issued; before tap k's first use of l[k+1] at most
        // 10 - k of the texel reads ... may be outstanding behind the two tile reads (counted below per tap).
        LK_ROW9_TAP(0, 1, 1, "") LK_ROW9_USE_G(0, LK_QE0, LK_QE1, LK_QE2, "ds_read_b128 " LK_QE ", %[ta] offset:32\n\t")
        LK_ROW9_TAP(1, 2, 1, "") LK_ROW9_USE_G(1, LK_QO0, LK_QO1, LK_QO2, "ds_read_b128 " LK_QO ", %[ta] offset:48\n\t")
        LK_ROW9_TAP(2, 3, 1, "") LK_ROW9_USE_G(2, LK_QE0, LK_QE1, LK_QE2, "ds_read_b128 " LK_QE ", %[ta] offset:64\n\t")
        LK_ROW9_TAP(3, 4, 1, "") LK_ROW9_USE_G(3, LK_QO0, LK_QO1, LK_QO2, "ds_read_b128 " LK_QO ", %[ta] offset:80\n\t")
        LK_ROW9_TAP(4, 5, 1, "") LK_ROW9_USE_G(4, LK_QE0, LK_QE1, LK_QE2, "ds_read_b128 " LK_QE ", %[ta] offset:96\n\t")
        LK_ROW9_TAP(5, 6, 1, "") LK_ROW9_USE_G(5, LK_QO0, LK_QO1, LK_QO2, "ds_read_b128 " LK_QO ", %[ta] offset:112\n\t")
        LK_ROW9_TAP(6, 7, 1, "") LK_ROW9_USE_G(6, LK_QE0, LK_QE1, LK_QE2, "ds_read_b128 " LK_QE ", %[ta] offset:128\n\t")
        LK_ROW9_TAP(7, 8, 1, "") LK_ROW9_USE_G(7, LK_QO0, LK_QO1, LK_QO2, "")
        LK_ROW9_TAP(8, 9, 0, "") LK_ROW9_USE_G(8, LK_QE0, LK_QE1, LK_QE2, "")
        : [l0] "=&v"(r[L(0)]), [l1] "=&v"(r[L(1)]), [l2] "=&v"(r[L(2)]), [l3] "=&v"(r[L(3)]), [l4] "=&v"(r[L(4)]), [l5] "=&v"(r[L(5)]),
          [l6] "=&v"(r[L(6)]), [l7] "=&v"(r[L(7)]), [l8] "=&v"(r[L(8)]), [l9] "=&v"(r[L(9)]),
          [t0] "+v"(r[T(0)]), [t1] "+v"(r[T(1)]), [t2] "+v"(r[T(2)]), [t3] "+v"(r[T(3)]), [t4] "+v"(r[T(4)]), [t5] "+v"(r[T(5)]),
          [t6] "+v"(r[T(6)]), [t7] "+v"(r[T(7)]), [t8] "+v"(r[T(8)]), [bx] "+v"(bx), [by] "+v"(by), [gxx] "+v"(gxx), [gxy] "+v"(gxy), [gyy] "+v"(gyy), [tmp] "=&v"(tmp)
        : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),
          [a8] "v"(a[8]), [ay] "v"(ay), [ja] "v"(jaddr), [ta] "v"(taddr)
        : LK_Q_CLOBBERS, "memory");
}
#undef LK_ROW9_TAP
#undef LK_ROW9_USE
#undef LK_ROW9_USE_G

// All nine rows of a step as ONE asm statement, LDS reads software-pipelined across the rows (tools/gen_lk_rows9.py has the
// schedule and derives every wait count; lk_rows9.inc is its output).  For waves all of whose member pixels have consecutive
// window columns AND rows: sample row of window row r = the first + r, so every address is the base + an immediate.
//   ja: LDS address of texel 0 of the UPPER sample row of window row 0;  ta: of the record of window row 0, tap 0
//   yf0 = (float)(y - R), fy = the flow's v: row r's fraction is v_fract_f32((yf0 + r) + fy), the oracle's sum
#include "lk_rows9.inc"
template <bool WITH_G>
__device__ __forceinline__ void lk_rows9_asm(const float (&a)[9], float fy, float yf0, float& bx, float& by, float& gxx, float& gxy,
                                             float& gyy, uint32_t ja, uint32_t ta) {
    float r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15, r16, r17, r18, tmp, ay;
#ifdef OFPS_LK_X_DUMMY                   // timing experiments (LK_GEN_DUMMY of the generator): a scratch register for the extra instructions
    float ay2;
#define LK_ROWS9_X , [ay2] "=&v"(ay2)
#else
#define LK_ROWS9_X
#endif
#define LK_ROWS9_REGS                                                                                                             \
    [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5), [r6] "=&v"(r6), [r7] "=&v"(r7),  \
    [r8] "=&v"(r8), [r9] "=&v"(r9), [r10] "=&v"(r10), [r11] "=&v"(r11), [r12] "=&v"(r12), [r13] "=&v"(r13), [r14] "=&v"(r14),       \
    [r15] "=&v"(r15), [r16] "=&v"(r16), [r17] "=&v"(r17), [r18] "=&v"(r18), [tmp] "=&v"(tmp), [ay] "=&v"(ay) LK_ROWS9_X
#define LK_ROWS9_INS                                                                                                              \
    [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),    \
    [a8] "v"(a[8]), [fy] "v"(fy), [yf0] "v"(yf0), [ja] "v"(ja), [ta] "v"(ta)
    if constexpr (WITH_G) {
        asm volatile(LK_ROWS9_APPLY(LK_ROWS9_BODY_G, LK_QUADS)
                     : LK_ROWS9_REGS, [bx] "+v"(bx), [by] "+v"(by), [gxx] "+v"(gxx), [gxy] "+v"(gxy), [gyy] "+v"(gyy)
                     : LK_ROWS9_INS
                     : LK_Q_CLOBBERS, "memory");
    } else {
        asm volatile(LK_ROWS9_APPLY(LK_ROWS9_BODY, LK_QUADS)
                     : LK_ROWS9_REGS, [bx] "+v"(bx), [by] "+v"(by)
                     : LK_ROWS9_INS
                     : LK_Q_CLOBBERS, "memory");
    }
#undef LK_ROWS9_REGS
#undef LK_ROWS9_INS
}

// The same for the two pixels of a lane (A above B; LK_ROWS9_PAIR_BODY): the ten record rows under the pair are read once.
//   a / b: the column fractions of A / B;  fy / fz: their flows' v;  yf0 = (float)(yA - R): B's row r has the fraction
//   v_fract_f32((yf0 + (r + 1)) + fz), the oracle's sum for y = yA + 1;  ja / jb: LDS address of texel 0 of the upper sample
//   row of each pixel's window row 0;  ta: of the record of A's window row 0, tap 0.  sums[0..4] = bx, by, gxx, gxy, gyy of A,
//   sums[5..9] of B (the tensors only read and written WITH_G).
template <bool WITH_G>
__device__ __forceinline__ void lk_rows9_pair_asm(const float (&a)[9], const float (&b)[9], float fy, float fz, float yf0, float (&sums)[10],
                                                  uint32_t ja, uint32_t jb, uint32_t ta) {
    float r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15, r16, r17, r18, tmp, ay;
    float s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15, s16, s17, s18, tmq, az;
#define LK_ROWS9_REGS                                                                                                             \
    [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5), [r6] "=&v"(r6), [r7] "=&v"(r7),  \
    [r8] "=&v"(r8), [r9] "=&v"(r9), [r10] "=&v"(r10), [r11] "=&v"(r11), [r12] "=&v"(r12), [r13] "=&v"(r13), [r14] "=&v"(r14),       \
    [r15] "=&v"(r15), [r16] "=&v"(r16), [r17] "=&v"(r17), [r18] "=&v"(r18), [tmp] "=&v"(tmp), [ay] "=&v"(ay),                       \
    [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [s3] "=&v"(s3), [s4] "=&v"(s4), [s5] "=&v"(s5), [s6] "=&v"(s6), [s7] "=&v"(s7),  \
    [s8] "=&v"(s8), [s9] "=&v"(s9), [s10] "=&v"(s10), [s11] "=&v"(s11), [s12] "=&v"(s12), [s13] "=&v"(s13), [s14] "=&v"(s14),       \
    [s15] "=&v"(s15), [s16] "=&v"(s16), [s17] "=&v"(s17), [s18] "=&v"(s18), [tmq] "=&v"(tmq), [az] "=&v"(az)
#define LK_ROWS9_INS                                                                                                              \
    [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]),    \
    [a8] "v"(a[8]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6]),    \
    [b7] "v"(b[7]), [b8] "v"(b[8]), [fy] "v"(fy), [fz] "v"(fz), [yf0] "v"(yf0), [ja] "v"(ja), [jb] "v"(jb), [ta] "v"(ta)
    if constexpr (WITH_G) {
        asm volatile(LK_ROWS9_APPLY(LK_ROWS9_PAIR_BODY_G, LK_QUADS)
                     : LK_ROWS9_REGS, [bx] "+v"(sums[0]), [by] "+v"(sums[1]), [gxx] "+v"(sums[2]), [gxy] "+v"(sums[3]), [gyy] "+v"(sums[4]),
                       [cx] "+v"(sums[5]), [cy] "+v"(sums[6]), [hxx] "+v"(sums[7]), [hxy] "+v"(sums[8]), [hyy] "+v"(sums[9])
                     : LK_ROWS9_INS
                     : LK_Q_CLOBBERS, "memory");
    } else {
        asm volatile(LK_ROWS9_APPLY(LK_ROWS9_PAIR_BODY, LK_QUADS)
                     : LK_ROWS9_REGS, [bx] "+v"(sums[0]), [by] "+v"(sums[1]), [cx] "+v"(sums[5]), [cy] "+v"(sums[6])
                     : LK_ROWS9_INS
                     : LK_Q_CLOBBERS, "memory");
    }
#undef LK_ROWS9_REGS
#undef LK_ROWS9_INS
}

// Current-frame window in LDS.  The bilinear fetches J(q + flow(p)) of a workgroup land in the rectangle
// [tile + window] shifted by the flows of its pixels; when those flows differ by at most SPREAD_X / SPREAD_Y pixels (almost every
// tile) the rectangle fits jl[][] and is staged once, coordinates clamped at staging time exactly like the oracle
// clamps xa/xb/ya/yb -- the inner loop then has no global loads at all.  Tiles with wilder flows are taken in groups of pixels
// whose boxes share a capacity-sized rectangle, and whatever no group takes reads the current frame from global memory
// (lk_level_body); tiles whose window columns do not sample consecutive texels (left/right image border, windows across a
// binade) read every column's own texel pair from the same rectangle.  Same values, same operation order either way,
// hence the same bits.
template <int RADIUS>
struct LkStepShared {
    using T = LkTile<RADIUS>;
    // capacity of the current-frame rectangle: flows inside a tile may differ by up to SPREAD_X / SPREAD_Y pixels; only the
    // rectangle a tile really needs is staged, so the capacity costs LDS space, not time.  Horizontally the capacity stops
    // where the row stride reaches 64 floats at radius 4 (a wider stride, e.g. 76 for +-32, costs 5 % on ordinary content:
    // measured 0.399 vs 0.380 ms); vertically it is only rows (+-16 px region jumps: 0.535 -> 0.475 ms)
#ifndef OFPS_LK_SPREAD_Y
#define OFPS_LK_SPREAD_Y 32
#endif
    static constexpr int SPREAD_X = 22, SPREAD_Y = OFPS_LK_SPREAD_Y, LW = T::TW + 1 + SPREAD_X, LH = T::TH + 1 + SPREAD_Y;
    // row stride in floats: a multiple of 4, so rows start 16-byte aligned.  OFPS_LK_JS (64 or 68; lk_rows9.inc carries a body
    // for each): at 64 -- the number of LDS banks -- two lanes that read the same column of different rows collide
    static constexpr int JS = RADIUS == 4 ? OFPS_LK_JS : (LW + 4) / 4 * 4;
    static_assert(JS >= (LW + 4) / 4 * 4, "row stride shorter than the rectangle");
    // jl first: its reads are ds_read2_b32, whose two offsets are 8 bits of dwords -- at LDS offset 0 the N+1 texels of a
    // row are reachable from one address register, behind the tile each pair costs a v_add_u32
    alignas(16) float jl[LH][JS];
    float4 tile[T::TH][T::TW];         // (I, gx, gy, -) of the previous frame's window
    int box[2][4][4];                  // [step parity][wave]: min x0, max x0+1, min y0, max y0+1 of the wave's sample origins
    int anchor[2][4][8];               // grouped tiles, [round parity][wave]: has a pending pixel, that pixel's box (x0, x0+1, y0, y0+1)
    // the f32 copy of the previous frame's window (u8 frame at level 0, f32 plane above) the tile records are made from.  A
    // buffer of its own (3.4 KB at radius 4; 6 workgroups per CU still fit): the records are made while the first rectangle's
    // loads are in flight
    alignas(16) float u8win[LkU8Window<RADIUS>::FLOATS];
    // what the LDS footprint allows (160 KB per CU, 4 waves per workgroup): the register budget hipcc is held to
    // (radius 4 measured at 5 / 6 / 7 waves per SIMD: 0.396 / 0.380 / 0.411 ms -- 94 registers without spills, 80 with 3
    // spilled outside the row loop, 72 with 11)
#ifndef OFPS_LK_WAVES4
#define OFPS_LK_WAVES4 6
#endif
    // (two pixels per thread: 35 KB of LDS per workgroup -> 4 workgroups per CU -> 128 registers)
    static constexpr int WAVES_PER_SIMD = RADIUS <= 2 ? 7 : RADIUS <= 4 ? (T::PP == 2 ? 4 : OFPS_LK_WAVES4) : 4;      // (radius 2: 69 registers -> 7, and 7 x 20.5 KB of LDS fit)
};

// integer sample origin of a window column / row: the oracle's floor + float clamp to [-1, lim]
__device__ __forceinline__ int lk_origin(int q, float fl, int lim, float& frac) {
    const float fq = (float)q + fl;
    const float q0f = floorf(fq);
    frac = fq - q0f;
    // the oracle's two-sided clamp of the floor; one v_med3_f32 (flows are finite: they come from u8 frames through
    // guarded 2x2 solves, so the NaN cases in which a median and a compare chain differ do not arise)
    const float c = __builtin_amdgcn_fmed3f(q0f, -1.0f, (float)lim);
    return (int)c;
}

// Where a level kernel takes a pixel's flow from and where it puts the result.  Two small kernels per level are folded in:
//   * start of a level: flow_in = 2 * coarse(x/2, y/2) (lk_upsample_kernel's expression, evaluated on the fly), or zero
//     at the coarsest level;
//   * end of level 0: the per-pixel records cv-decoder emits (lk_entries_kernel's expressions) are written directly; the
//     flow plane itself only if somebody asked for it.
struct LkFlowIO {
    const float2* coarse;       // flow of the next coarser level (w1 x h1), or nullptr: the level starts from `init` / zero flow
    const float2* init;         // coarsest level only: the caller's starting flow (w1 = its row pitch), or nullptr = zero
    int w1, h1;
    float2* flow_out;           // or nullptr
    float4* out_entries;        // or nullptr
    float nx, ny;               // 1/W, 1/H for the records
    // One launch runs the whole pyramid (lk_levels_kernel): a level's flow plane is written by workgroups of that launch and
    // read by others, possibly on another XCD (own L2): such planes are written through (sc0 sc1 stores) and read past the
    // caches (sc0 sc1 loads), the idiom of the cluster Almeida solver's granules (almeida.hip).
    int coarse_shared;          // `coarse` was written by this launch
    int out_shared;             // `flow_out` is read by this launch
};

typedef float lk_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 lk_load_shared(const float2* p) {
    lk_f2 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void lk_store_shared(float2* p, float2 v) {
    const lk_f2 q = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(q) : "memory");
}

__device__ __forceinline__ float2 lk_flow_read(const LkFlowIO& io, int x, int y) {
    if (io.coarse) {                                                   // uniform
        const float2* p = io.coarse + (size_t)lk_clampi(y / 2, 0, io.h1 - 1) * io.w1 + lk_clampi(x / 2, 0, io.w1 - 1);
        const float2 c = io.coarse_shared ? lk_load_shared(p) : *p;
        return make_float2(2.0f * c.x, 2.0f * c.y);
    }
    if (io.init) return io.init[(size_t)y * io.w1 + x];               // uniform
    return make_float2(0.0f, 0.0f);
}

// the Gauss-Newton update of one pixel: G^-1 b added to the flow (zero step when det <= 0.01)
__device__ __forceinline__ float2 lk_solve(const float4 g, const float2 f, float bx, float by) {
    const float det = g.x * g.z - g.y * g.y;
    float du = 0.0f, dv = 0.0f;
    if (det > 0.01f) {
        du = (g.z * bx - g.y * by) / det;
        dv = (g.x * by - g.y * bx) / det;
    }
    return make_float2(f.x + du, f.y + dv);
}

__device__ __forceinline__ void lk_store(const float2 out, int x, int y, int w, const LkFlowIO& io) {
    const size_t idx = (size_t)y * w + x;
    if (io.flow_out) { if (io.out_shared) lk_store_shared(io.flow_out + idx, out); else io.flow_out[idx] = out; }
    if (io.out_entries) io.out_entries[idx] = make_float4(((float)x + 0.5f) * io.nx, ((float)y + 0.5f) * io.ny, out.x * io.nx, out.y * io.ny);
}

// main kernel: one workgroup per kTX x kTY tile runs ALL `iters` Gauss-Newton steps of a pyramid level -- a step of a pixel
// depends on nothing but that pixel's own flow, so the previous frame's window (I, gx, gy), the tensor G and the flow
// stay on chip between steps; only the current frame's rectangle is restaged (it moves with the flow).
// A tile whose rectangle does not fit LDS at some step (flows that disagree by more than SPREAD_X / SPREAD_Y pixels: region
// borders of +-16 px content, 8 % of the level-0 tiles there) is finished INSIDE this launch (round 4; rounds 2-3 parked
// its flow for a second, per-lane-gather kernel: three launches per pair, +34 % on such content): its pixels are taken in
// GROUPS -- the first pixel not yet done anchors a capacity-sized rectangle, every pending pixel whose own sample box lies
// inside it joins, the rectangle is staged and the members run the ordinary rows; 2-3 rounds for a region border
// (tools/lk_tile_stats.py), at most kMaxRounds -- and whatever is still pending after that (flows without any
// coherence) takes the oracle's per-sample form from global memory, tile records from LDS.  A pixel's arithmetic does not
// depend on which path it takes: same operands, same operations, same order.
// U8 = true (level 0): I_ / J_ are the u8 frames themselves, rows src_stride bytes apart; no f32 level-0 planes and no
// level-0 gradient planes exist (lk_stage3_u8; the rectangle is converted while it is staged).
constexpr int kLkMaxRounds = 8;
constexpr long long kLkHelpAfterTicks = 15000;      // wall_clock64 ticks (100 MHz) per ancestor level a tile waits for its parent before it computes it itself
template <int RADIUS, bool U8>
__device__ __forceinline__ bool lk_level_body(LkStepShared<RADIUS>& sh, const void* __restrict__ I_, const void* __restrict__ J_, int src_stride,
                                              int w, int h, int iters, const LkFlowIO io, unsigned long long* __restrict__ prof,
                                              int force_fall_arg, int tile_x, int tile_y, const uint32_t* parent_flag, uint32_t* done_flag,
                                              uint32_t epoch, int wait_budget_arg, int depth) {
    // force_fall (libofps_hip_testhooks.so only; compiled out of the product library): low 4 bits = a step at which every
    // other tile is treated as not fitting, so that the grouped path in the middle of a level is exercised on inputs that
    // would never trigger it; bits 4.. = how many grouping rounds those tiles get (0 = the default kLkMaxRounds; 1 + n = n
    // rounds, so 1 sends every pixel through the per-sample leftover path)
#ifdef OFPS_HIP_TEST_HOOKS
    const int force_fall = force_fall_arg < 0 ? -1 : (force_fall_arg & 15);
    const int max_rounds = force_fall_arg >= 16 ? (force_fall_arg >> 4) - 1 : kLkMaxRounds;
#else
    constexpr int force_fall = -1;
    constexpr int max_rounds = kLkMaxRounds;
#endif
    // prof (diagnostics, normally null): per-workgroup s_memtime stamps at the phase boundaries of the first step
#define OFPS_LK_STAMP(slot) do { if (prof && threadIdx.x == 0) prof[((size_t)tile_y * tiles_x + tile_x) * 6 + (slot)] = __builtin_readcyclecounter(); } while (0)
    using T = LkTile<RADIUS>;
    using S = LkStepShared<RADIUS>;
    const int tiles_x = (w + kTX - 1) / kTX;
    OFPS_LK_STAMP(0);
    constexpr int N = T::N;
    constexpr int PP = T::PP;                                // pixels per lane: lane (lx, lq) owns tile rows PP lq ... PP lq + PP - 1
    const int x0 = tile_x * kTX, y0 = tile_y * T::TY;
    using TIn = std::conditional_t<U8, uint8_t, float>;
    const float* J = static_cast<const float*>(J_);
    const uint8_t* J8 = static_cast<const uint8_t*>(J_);
    const int lx = threadIdx.x % kTX, ly = (threadIdx.x / kTX) * PP, px = x0 + lx, py = y0 + ly;     // the lane's first pixel
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bool active[PP];                                         // (a lane's second pixel is only active if its first one is)
#pragma unroll
    for (int p = 0; p < PP; ++p) active[p] = px < w && py + p < h;
    // the previous frame's window is requested first (it depends on nothing) and used after the first box exchange: its
    // latency runs beside the wait for the parent tile, the flow read, the column origins and the exchange
    LkU8Regs<RADIUS, TIn> u8g;
    lk_stage3_u8_issue<RADIUS, TIn>(static_cast<const TIn*>(I_), U8 ? src_stride : w, w, h, x0, y0, u8g);
    // One launch runs the whole pyramid (lk_levels_kernel): this tile's starting flows are the results of ONE tile of the next
    // coarser level (its parent), normally a workgroup of the same launch with a lower block index that is running or done.  Thread 0
    // polls the parent's flag past the caches until it carries this launch's epoch.  FORWARD PROGRESS (round 6) does not depend on that
    // "normally": a tile whose parent has not published within kLkHelpAfterTicks per ancestor level (0.15 ms each; an in-order launch makes
    // a tile wait at most about half of that per level) stops waiting and COMPUTES the missing ancestors itself, coarsest first, then itself (lk_help_tile) -- a tile's flows are a pure
    // function of the frames, so a tile computed twice is written twice with the same bits and its flag set twice with the same epoch.
    // Every workgroup therefore finishes in bounded time whatever the dispatcher does (any order, any number of resident workgroups, a
    // CU-masked stream), and no flow is ever made from an unfinished parent.  Rounds 4-5 went on "with whatever the plane holds" and
    // left the repeat to the host.
    if (parent_flag) {                                                   // uniform
        __shared__ int s_help;
        if (threadIdx.x == 0) {
            uint32_t v;
#ifdef OFPS_HIP_TEST_HOOKS                       // OFPS_HIP_LK_TEST_WAIT_BUDGET: a budget of one tick makes most tiles compute their ancestors themselves
            const long long budget = wait_budget_arg > 0 ? wait_budget_arg : kLkHelpAfterTicks * depth;
#else
            const long long budget = kLkHelpAfterTicks * depth;
#endif
            const long long t_start = wall_clock64();                    // 100 MHz
            int help = 0;
            for (;;) {
                asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(parent_flag) : "memory");
                if (v == epoch) break;
                if (wall_clock64() - t_start >= budget) { help = 1; break; }
                __builtin_amdgcn_s_sleep(16);
            }
            s_help = help;
        }
        __syncthreads();
        if (s_help) return false;                                        // uniform; rare: the caller hands the tile to lk_help_tile (nothing has been written yet)
    }
    float2 f[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) f[p] = active[p] ? lk_flow_read(io, px, py + p) : make_float2(0.0f, 0.0f);
    // The 2x2 structure tensor of the pixel's window does not depend on the flow: it is summed by the level's FIRST step, from
    // the very tile records that step reads for the residual (three fused multiply-adds per tap more, no LDS traffic of
    // its own), and stays in three registers for the later steps.
    float gxx[PP], gxy[PP], gyy[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) gxx[p] = gxy[p] = gyy[p] = 0.0f;
    bool st_valid = false;                                   // the rectangle of the current frame held in jl[][] (uniform)
    int st_x0 = 0, st_x1 = -1, st_y0 = 0, st_y1 = -1, st_xs = 0;
    int x = 0, y = 0;                                        // the lane's first pixel (remake_xy)
    bool cons_x[PP], cons_y[PP];                             // per pixel: the window's columns / rows sample consecutive texels
#pragma unroll
    for (int p = 0; p < PP; ++p) cons_x[p] = cons_y[p] = false;
#ifdef OFPS_LK_NO_FAST_ORIGINS                              // A/B: every tile takes the clamped, per-column origin chains
    constexpr bool kFastOrigins = false;
#else
    constexpr bool kFastOrigins = true;
#endif
    // uniform: no window column / row of this tile is clamped to the image
    const bool tile_in_x = kFastOrigins && x0 >= RADIUS && x0 + kTX - 1 + RADIUS <= w - 1;
    const bool tile_in_y = kFastOrigins && y0 >= RADIUS && y0 + T::TY - 1 + RADIUS <= h - 1;
    // A pixel's sample box: origins of its first / last window column and row (the last + 1: the texel the interpolation also
    // reads), and whether its columns / rows sample consecutive texels.  Unclamped windows (interior tiles): the sums
    // fq_k = (x + k - R) + u grow with k, and while they are >= 0 so does their rounding step -- a sum that rounds up to an
    // integer is followed by sums that do too, so the floors advance by 1 or (across a binade) 2, never 0: consecutive exactly
    // when the last floor is 2R above the first.  Two floor chains instead of N per axis (tests/test_lk_origin_claims.py).
    // The lemma is about UNCLAMPED floors and lk_origin clamps to [-1, lim]: a last floor that reached lim may have been
    // clamped down to it (floors 1022, 1023, 1025, ..., 1031 across the 1024 binade at w = 1030 read as 1022 .. 1030), so
    // such windows count as not consecutive; a first floor below 0 likewise.
    auto lane_box = [&](int xq, int yq, const float2 fl, int& a0, int& a1, int& b0, int& b1, bool& cx, bool& cy) {
        float dm;
        if (tile_in_x) {
            a0 = lk_origin(xq - RADIUS, fl.x, w, dm);
            const int al = lk_origin(xq + RADIUS, fl.x, w, dm);
            cx = a0 >= 0 && al < w && al - a0 == 2 * RADIUS;
            a1 = al + 1;
        } else {
            int prev = 0, al = 0;
            cx = true;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int o = lk_origin(lk_clampi(xq + k - RADIUS, 0, w - 1), fl.x, w, dm);
                if (k > 0) cx = cx && (o == prev + 1);
                if (k == 0) a0 = o;
                al = o; prev = o;
            }
            a1 = al + 1;
        }
        if (tile_in_y) {
            b0 = lk_origin(yq - RADIUS, fl.y, h, dm);
            const int bl = lk_origin(yq + RADIUS, fl.y, h, dm);
            cy = b0 >= 0 && bl < h && bl - b0 == 2 * RADIUS;
            b1 = bl + 1;
        } else {
            // window rows are monotone in r, so the extremes are the first and the last
            b0 = lk_origin(lk_clampi(yq - RADIUS, 0, h - 1), fl.y, h, dm);
            b1 = lk_origin(lk_clampi(yq + RADIUS, 0, h - 1), fl.y, h, dm) + 1;
            cy = false;
        }
    };
    // the pixel coordinates pass through an empty asm so that the compiler does not hoist the clamped window coordinates (2N
    // integers + their float conversions) out of the step loop: that costs 40 VGPRs and two waves per SIMD for a handful of
    // integer operations per step (made from the thread index every time instead of copied from px / py: two registers less
    // across the row loop)
    auto remake_xy = [&]() {
        int tidx = (int)threadIdx.x;
        asm volatile("" : "+v"(tidx));
        x = x0 + tidx % kTX; y = y0 + tidx / kTX * PP;
    };
    // The box exchange that opens a step: every wave's box of sample origins goes to sh.box (slots alternate with the step: a
    // step that reuses the staged rectangle has no second barrier, so a fast wave may write the next step's box while a slow
    // one still reads this step's).  The caller's barrier follows.
    auto box_exchange = [&](int it) {
        remake_xy();
        int bx0 = 0x7FFFFFFF, bx1 = -0x7FFFFFFF, by0 = 0x7FFFFFFF, by1 = -0x7FFFFFFF;
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            int a0, a1, b0, b1;
            lane_box(x, y + p, f[p], a0, a1, b0, b1, cons_x[p], cons_y[p]);
            bx0 = active[p] ? min(bx0, a0) : bx0; bx1 = active[p] ? max(bx1, a1) : bx1;
            by0 = active[p] ? min(by0, b0) : by0; by1 = active[p] ? max(by1, b1) : by1;
        }
        lk_wave_box(bx0, bx1, by0, by1);
        if (lane == 0) { int* b = sh.box[it & 1][wave]; b[0] = bx0; b[1] = bx1; b[2] = by0; b[3] = by1; }
    };
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    // (Re)stages columns [rx0, rx1] x rows [ry0, ry1] of the current frame into jl[][], coordinates clamped per element exactly
    // like the oracle clamps xa / xb / ya / yb; 16-byte loads when the padded rectangle lies inside the frame.  Uniform.
    auto stage_rect = [&](int rx0, int rx1, int ry0, int ry1) {
        // (thread index through an empty asm: the staging addresses are made here, per staging, not hoisted out of the step
        // loop into registers that are then spilled for the whole level)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        int xs_new = rx0;
        const int chh = ry1 - ry0 + 1;
        const int xa4 = rx0 & ~3, cw4 = (rx1 - xa4 + 4) >> 2;                        // float4 per row after padding
        const bool vec = rx0 >= 0 && xa4 + 4 * cw4 <= w && 4 * cw4 <= S::JS &&
                         (U8 ? (src_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(J8) & 3) == 0
                             : (w & 3) == 0 && (reinterpret_cast<uintptr_t>(J) & 15) == 0);
        if (vec) {                                                   // uniform
            xs_new = xa4;
            // 32 lanes per rectangle row (cw4 <= JS / 4 <= 32), 8 rows per pass: no division by the run-time width
            static_assert(S::JS / 4 <= 32, "J staging assumes at most 32 float4 per rectangle row");
            const int c4 = tid & 31;
            if (c4 < cw4) {
                for (int cy = tid >> 5; cy < chh; cy += 8) {
                    if constexpr (U8) {                              // four pixels per dword, converted on the way in
                        const uint32_t q = *reinterpret_cast<const uint32_t*>(J8 + (size_t)lk_clampi(ry0 + cy, 0, h - 1) * src_stride + xa4 + 4 * c4);
                        *reinterpret_cast<float4*>(&sh.jl[cy][4 * c4]) =
                            make_float4((float)(q & 0xFFu), (float)((q >> 8) & 0xFFu), (float)((q >> 16) & 0xFFu), (float)(q >> 24));
                    } else {
                        *reinterpret_cast<float4*>(&sh.jl[cy][4 * c4]) =
                            *reinterpret_cast<const float4*>(J + (size_t)lk_clampi(ry0 + cy, 0, h - 1) * w + xa4 + 4 * c4);
                    }
                }
            }
        } else {
            const int cw = rx1 - rx0 + 1;
            const int cx = tid & 127, cy0 = tid >> 7;            // 128 threads per row, two rows per pass
            if (cx < cw) {
                const int gxc = lk_clampi(rx0 + cx, 0, w - 1);
                for (int cy = cy0; cy < chh; cy += 2) {
                    if constexpr (U8) sh.jl[cy][cx] = (float)J8[(size_t)lk_clampi(ry0 + cy, 0, h - 1) * src_stride + gxc];
                    else sh.jl[cy][cx] = J[(size_t)lk_clampi(ry0 + cy, 0, h - 1) * w + gxc];
                }
            }
        }
        st_x0 = rx0; st_x1 = rx1; st_y0 = ry0; st_y1 = ry1; st_xs = xs_new;
    };

    // ---- step 0's box exchange in front of the loop: the only place that touches the u8 window's staging registers, whose
    // lives therefore end before the loop
    if (iters > 0) {
        box_exchange(0);
        lk_stage3_u8_spill<RADIUS, TIn>(sh.u8win, u8g);                  // the window's pixels, requested in the prologue; same barrier as the box
        OFPS_LK_STAMP(1);
        __syncthreads();
        OFPS_LK_STAMP(2);
        // the tile records from the f32 copy of the u8 window: visible after the barrier that follows the first staging
        lk_stage3_u8_records<RADIUS>(sh.u8win, sh.tile, u8g.ox, u8g.vec, w, h, x0, y0);
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (it > 0) {
            box_exchange(it);
            __syncthreads();                                             // also: everybody is done reading jl[] of the previous step
        }
        // (box[] holds the same numbers for every thread: they go to SCALAR registers -- the rectangle's bounds live across the
        // whole level, and as vector registers they were part of what got spilled around the row loop.  One dword per lane
        // and sixteen v_readlane: four 16-byte reads per lane took sixteen vector registers for a moment, and the allocator
        // spilled three live values around them in every step)
        const int bv = reinterpret_cast<const int*>(sh.box[it & 1])[lane & 15];
        auto bl = [&](int k) { return __builtin_amdgcn_readlane(bv, k); };
        const int xmin = min(min(bl(0), bl(4)), min(bl(8), bl(12)));
        const int xmax = max(max(bl(1), bl(5)), max(bl(9), bl(13)));
        const int bymin = min(min(bl(2), bl(6)), min(bl(10), bl(14)));
        const int ymax = max(max(bl(3), bl(7)), max(bl(11), bl(15)));
        const bool fits = xmax >= xmin && xmax - xmin < S::LW && ymax - bymin < S::LH &&
                          !(it == force_fall && ((tile_x + tile_y) & 1));
        // The rows of one Gauss-Newton step for the lanes that call it (the members of the staged rectangle), ending with the flow
        // update: instantiated for the ordinary path (the whole tile) and once more inside the grouping rounds of an unfit tile,
        // so that nothing of the rare path is live across the ordinary one.
        auto rows_one = [&](const int p) {
            const int xs = st_xs, ymin = st_y0;                          // origin of jl[][] in frame coordinates
            // this pixel's state (p is uniform: selects, not indexed registers) -- written back at the end
            const bool second = PP == 2 && p != 0;
            float2 fl = second ? f[PP - 1] : f[0];
            float g0 = second ? gxx[PP - 1] : gxx[0], g1 = second ? gxy[PP - 1] : gxy[0], g2 = second ? gyy[PP - 1] : gyy[0];
            const int yp = y + p, lyp = ly + p;
            // wave-uniform: every member lane's window columns / rows sample consecutive texels
            const bool all_cons = __all(second ? cons_x[PP - 1] : cons_x[0]);
            const bool fast_x = tile_in_x && all_cons;
            const bool fast_y = tile_in_y && __all(second ? cons_y[PP - 1] : cons_y[0]);
            float ax[N];
            int xi0;                                                     // origin of window column 0 (all the consecutive-column rows need)
            {
                int xr = x;
                asm volatile("" : "+v"(xr));                              // opaque: a second evaluation, not the first one kept alive
                if (fast_x) {
                    // unclamped columns with floors 0 <= xi0, xi0 + 1, ... < w: no clamp is active, and for a sum fq >= 0 the
                    // fraction fq - floor(fq) is exact, which is what v_fract_f32 returns -- the oracle's values in 3
                    // operations per column instead of 7
                    const float xf0 = (float)(xr - RADIUS);
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const float fq = (xf0 + (float)k) + fl.x;            // (float)(x + k - R), exact, + u: the oracle's sum
                        ax[k] = __builtin_amdgcn_fractf(fq);
                        if (k == 0) xi0 = (int)__builtin_floorf(fq);
                    }
                } else {
                    float frac;
#pragma unroll
                    for (int k = 0; k < N; ++k) { const int o = lk_origin(lk_clampi(xr + k - RADIUS, 0, w - 1), fl.x, w, frac); ax[k] = frac; if (k == 0) xi0 = o; }
                }
            }
            // hup[k] = horizontal interpolation of the UPPER sample row at column k.  The lower row of one window row is the
            // upper row of the next whenever the sample row advanced by exactly one (always, away from the top/bottom
            // border), and its interpolation j01 + ax[k] * (j11 - j01) is then the very expression the next row evaluates
            // as j00 + ax[k] * (j10 - j00) on the same texels: carried over instead of recomputed -- same operations on the
            // same inputs, 11 instead of 14 VALU operations per tap.  Decided per wave so the branch is uniform; recomputing
            // is always correct.
            float bx = 0.0f, by = 0.0f;
            bool done = false;
            if constexpr (RADIUS == 4 && OFPS_LK_SPEC_FMA) {
                if (fast_x && fast_y) {
                    // interior tile, every member's columns and rows consecutive (the ordinary case): all nine rows in one
                    // hand-scheduled block whose LDS reads are pipelined across the rows (lk_rows9_asm)
                    static_assert(S::JS * sizeof(float) == LK_ROWS9_JSB && T::TW * sizeof(float4) == LK_ROWS9_TRB, "lk_rows9.inc was generated for other pitches");
                    int yr = yp;
                    asm volatile("" : "+v"(yr));
                    const float yf0 = (float)(yr - RADIUS);
                    const int yi0 = (int)__builtin_floorf(yf0 + fl.y) - ymin;
                    const uint32_t ja = (uint32_t)reinterpret_cast<uintptr_t>(&sh.jl[yi0][xi0 - xs]);
                    const uint32_t ta = (uint32_t)reinterpret_cast<uintptr_t>(&sh.tile[lyp][lx]);
                    if (it == 0) lk_rows9_asm<true>(ax, fl.y, yf0, bx, by, g0, g1, g2, ja, ta);      // the level's first step also sums the structure tensor
                    else lk_rows9_asm<false>(ax, fl.y, yf0, bx, by, g0, g1, g2, ja, ta);
                    done = true;
                } else if (all_cons) {
                    // consecutive columns but clamped or irregular rows (top / bottom image border, flows that jump in y):
                    // hand-scheduled rows one at a time (lk_row9_asm): two register sets alternate between "this row's texels,
                    // turned into its interpolations" and "the previous row's interpolations", so the row loop runs in pairs
                    float rr[19];
                    int prev_yi = -0x7FFFFFFF;
                    const uint32_t jl0 = (uint32_t)reinterpret_cast<uintptr_t>(&sh.jl[0][xi0 - xs]);
                    const uint32_t tl0 = (uint32_t)reinterpret_cast<uintptr_t>(&sh.tile[lyp][lx]);
                    auto row = [&](int r, auto parity, auto with_g) {
                        constexpr int P = decltype(parity)::value;
                        float ay;
                        const int yi = lk_origin(lk_clampi(yp + r - RADIUS, 0, h - 1), fl.y, h, ay) - ymin;
                        // (the first row always makes its upper sample row: stated at compile time, so that the carried
                        // registers are not live into the loop -- hipcc spilled their undefined contents around every step)
                        const bool reuse = r > 0 && __all(yi == prev_yi + 1);
                        prev_yi = yi;
                        if (!reuse) {                              // the upper sample row is not the one carried over: make it
                            const float* ra = &sh.jl[yi][xi0 - xs];
                            float jb[N + 1];
#pragma unroll
                            for (int k = 0; k <= N; ++k) jb[k] = ra[k];
#pragma unroll
                            for (int k = 0; k < N; ++k) rr[P ? k : 10 + k] = lk_lerp(jb[k], jb[k + 1], ax[k]);
                        }
                        const uint32_t ja = jl0 + (uint32_t)(yi + 1) * (uint32_t)(S::JS * sizeof(float));
                        const uint32_t ta = tl0 + (uint32_t)r * (uint32_t)(T::TW * sizeof(float4));
                        if constexpr (decltype(with_g)::value) lk_row9_asm_g<P>(rr, ax, ay, bx, by, g0, g1, g2, ja, ta);
                        else lk_row9_asm<P>(rr, ax, ay, bx, by, ja, ta);
                    };
                    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
                    if (it == 0) {                                 // the level's first step also sums the structure tensor
                        row(0, P0{}, std::true_type{});
#pragma unroll 1
                        for (int r = 1; r < N; r += 2) { row(r, P1{}, std::true_type{}); row(r + 1, P0{}, std::true_type{}); }
                    } else {
                        row(0, P0{}, std::false_type{});
#pragma unroll 1
                        for (int r = 1; r < N; r += 2) { row(r, P1{}, std::false_type{}); row(r + 1, P0{}, std::false_type{}); }
                    }
                    done = true;
                }
            }
            if (!done) {
                float hup[N];
                int prev_yi = -0x7FFFFFFF;
                if (all_cons) {
                    const int xo = xi0 - xs;
                    float jb[N + 1];
                    // (measured and rejected: unrolling the row loop, fully or by two/three with ping-pong hup arrays -- hipcc
                    // then hoists the next row's LDS reads, 104+ VGPRs, 4 waves per SIMD, 0.53 vs 0.50 ms)
#pragma unroll 1
                    for (int r = 0; r < N; ++r) {
                        float ay;
                        const int yi = lk_origin(lk_clampi(yp + r - RADIUS, 0, h - 1), fl.y, h, ay) - ymin;
                        const bool reuse = __all(yi == prev_yi + 1);
                        prev_yi = yi;
                        if (!reuse) {
                            const float* ra = &sh.jl[yi][xo];
#pragma unroll
                            for (int k = 0; k <= N; ++k) jb[k] = ra[k];
#pragma unroll
                            for (int k = 0; k < N; ++k) hup[k] = lk_lerp(jb[k], jb[k + 1], ax[k]);
                        }
                        const float* rb = &sh.jl[yi + 1][xo];
#pragma unroll
                        for (int k = 0; k <= N; ++k) jb[k] = rb[k];
#pragma unroll
                        for (int k = 0; k < N; ++k) {
                            const float top = hup[k];
                            const float bot = lk_lerp(jb[k], jb[k + 1], ax[k]);
                            const lk_f4 t = lk_lds_read4(&sh.tile[lyp + r][lx + k]);
                            const float d = t.x - lk_lerp(top, bot, ay);
                            lk_accum(t.y, d, bx);
                            lk_accum(t.z, d, by);
                            if (it == 0) { lk_accum(t.y, t.y, g0); lk_accum(t.y, t.z, g1); lk_accum(t.z, t.z, g2); }
                            hup[k] = bot;
                        }
                    }
                } else {
                    // (image-border tiles and windows across a binade only: every column's own origin, made here -- kept alive
                    // from the pass above they were spilled around the hand-scheduled rows of every other tile)
                    int xi[N];
                    {
                        int xr = x;
                        asm volatile("" : "+v"(xr));
                        float frac;
#pragma unroll
                        for (int k = 0; k < N; ++k) xi[k] = lk_origin(lk_clampi(xr + k - RADIUS, 0, w - 1), fl.x, w, frac) - xs;
                    }
#pragma unroll 1
                    for (int r = 0; r < N; ++r) {
                        float ay;
                        const int yi = lk_origin(lk_clampi(yp + r - RADIUS, 0, h - 1), fl.y, h, ay) - ymin;
                        const bool reuse = __all(yi == prev_yi + 1);
                        prev_yi = yi;
                        if (!reuse) {
                            const float* ra = &sh.jl[yi][0];
#pragma unroll
                            for (int k = 0; k < N; ++k) { const float j0 = ra[xi[k]], j1 = ra[xi[k] + 1]; hup[k] = lk_lerp(j0, j1, ax[k]); }
                        }
                        const float* rb = &sh.jl[yi + 1][0];
#pragma unroll
                        for (int k = 0; k < N; ++k) {
                            const float j0 = rb[xi[k]], j1 = rb[xi[k] + 1];
                            const float top = hup[k];
                            const float bot = lk_lerp(j0, j1, ax[k]);
                            const lk_f4 t = lk_lds_read4(&sh.tile[lyp + r][lx + k]);
                            const float d = t.x - lk_lerp(top, bot, ay);
                            lk_accum(t.y, d, bx);
                            lk_accum(t.z, d, by);
                            if (it == 0) { lk_accum(t.y, t.y, g0); lk_accum(t.y, t.z, g1); lk_accum(t.z, t.z, g2); }
                            hup[k] = bot;
                        }
                    }
                }
            }
            fl = lk_solve(make_float4(g0, g1, g2, 0.0f), fl, bx, by);
            if (second) { f[PP - 1] = fl; gxx[PP - 1] = g0; gxy[PP - 1] = g1; gyy[PP - 1] = g2; }
            else { f[0] = fl; gxx[0] = g0; gxy[0] = g1; gyy[0] = g2; }
        };
        // The rows of one Gauss-Newton step for the member pixels m[] of the lanes that call it, ending with the flow update.
        auto rows_and_solve = [&](const bool (&m)[PP]) {
            if constexpr (PP == 2) {
                // interior tile, both pixels of every lane here are members with consecutive columns and rows (the ordinary case):
                // the two-pixel block -- the ten record rows under the pair are read once (lk_rows9_pair_asm)
                const bool pair_lane = m[0] && m[1] && cons_x[0] && cons_x[1] && cons_y[0] && cons_y[1];
                if (tile_in_x && tile_in_y && __all(pair_lane)) {
                    static_assert(S::JS * sizeof(float) == LK_ROWS9_JSB && T::TW * sizeof(float4) == LK_ROWS9_TRB, "lk_rows9.inc was generated for other pitches");
                    const int xs = st_xs, ymin = st_y0;
                    int xr = x, yr = y;
                    asm volatile("" : "+v"(xr), "+v"(yr));                // opaque: evaluated here, not kept alive from the box exchange
                    // (the single-pixel path's expressions: rows_one)
                    const float xf0 = (float)(xr - RADIUS), yf0 = (float)(yr - RADIUS);
                    float axa[N], axb[N];
                    int xia = 0, xib = 0;
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const float fa = (xf0 + (float)k) + f[0].x, fb = (xf0 + (float)k) + f[PP - 1].x;
                        axa[k] = __builtin_amdgcn_fractf(fa); axb[k] = __builtin_amdgcn_fractf(fb);
                        if (k == 0) { xia = (int)__builtin_floorf(fa); xib = (int)__builtin_floorf(fb); }
                    }
                    const int yia = (int)__builtin_floorf(yf0 + f[0].y) - ymin;
                    const int yib = (int)__builtin_floorf((yf0 + 1.0f) + f[PP - 1].y) - ymin;        // (float)(y + 1 - R) = yf0 + 1, exact
                    const uint32_t ja = (uint32_t)reinterpret_cast<uintptr_t>(&sh.jl[yia][xia - xs]);
                    const uint32_t jb = (uint32_t)reinterpret_cast<uintptr_t>(&sh.jl[yib][xib - xs]);
                    const uint32_t ta = (uint32_t)reinterpret_cast<uintptr_t>(&sh.tile[ly][lx]);
                    float sums[10] = {0.0f, 0.0f, gxx[0], gxy[0], gyy[0], 0.0f, 0.0f, gxx[PP - 1], gxy[PP - 1], gyy[PP - 1]};
                    if (it == 0) {                                        // the level's first step also sums the structure tensors
                        lk_rows9_pair_asm<true>(axa, axb, f[0].y, f[PP - 1].y, yf0, sums, ja, jb, ta);
                        gxx[0] = sums[2]; gxy[0] = sums[3]; gyy[0] = sums[4]; gxx[PP - 1] = sums[7]; gxy[PP - 1] = sums[8]; gyy[PP - 1] = sums[9];
                    } else {
                        lk_rows9_pair_asm<false>(axa, axb, f[0].y, f[PP - 1].y, yf0, sums, ja, jb, ta);
                    }
                    f[0] = lk_solve(make_float4(gxx[0], gxy[0], gyy[0], 0.0f), f[0], sums[0], sums[1]);
                    f[PP - 1] = lk_solve(make_float4(gxx[PP - 1], gxy[PP - 1], gyy[PP - 1], 0.0f), f[PP - 1], sums[5], sums[6]);
                    return;
                }
            }
#pragma unroll 1
            for (int p = 0; p < PP; ++p) {
                if (p ? m[PP - 1] : m[0]) rows_one(p);
            }
        };
        if (__builtin_expect(fits, 1)) {                                     // uniform: box[] is the same for every thread.  (expect: the register
            // allocator must keep the rare path's spills inside the rare path, not in front of the branch)
            // The rectangle is staged with a margin and KEPT: a later step whose box still lies inside it (flows move by a
            // fraction of a pixel per step once the coarser levels have done their work) reuses it -- no global loads, no
            // second barrier in that step.
            const bool inside = st_valid && xmin >= st_x0 && xmax <= st_x1 && bymin >= st_y0 && ymax <= st_y1;
            if (!inside) {
                // margins: up to kJMargin pixels on every side, as far as the capacity allows (measured in round 4: a wider margin
                // for tiles that have to stage again at a later step -- 4, 6, 10 pixels -- changes nothing on either content)
                const int mx = min(kJMargin, (S::LW - (xmax - xmin + 1)) / 2), my = min(kJMargin, (S::LH - (ymax - bymin + 1)) / 2);
                stage_rect(xmin - mx, xmax + mx, bymin - my, ymax + my);
                st_valid = true;
                __syncthreads();
            }
            if (it == 0) OFPS_LK_STAMP(3);
            if (active[0]) rows_and_solve(active);
        } else {
            // ---- a tile whose sample rectangle does not fit: its pixels in groups, at most max_rounds of them
            bool pending[PP];
#pragma unroll
            for (int p = 0; p < PP; ++p) pending[p] = active[p];
            st_valid = false;                                                // whatever jl[][] holds after this step is not the tile's rectangle
#pragma unroll 1
            for (int round = 0; round < max_rounds; ++round) {
                // anchor of this round: the first pending pixel of the first wave that has one.  The boxes are made again here (kept
                // alive across the rows they would be four more registers per pixel around the hot loop of every tile).
                remake_xy();
                int a0[PP], a1[PP], b0[PP], b1[PP];
                bool c1, c2;
#pragma unroll
                for (int p = 0; p < PP; ++p) lane_box(x, y + p, f[p], a0[p], a1[p], b0[p], b1[p], c1, c2);
                const bool lane_pending = pending[0] || pending[PP - 1];
                const bool first = pending[0];                               // the lane's first pending pixel is its first one
                int* an = sh.anchor[round & 1][wave];
                const bool any_pending = __any(lane_pending);
                if (lane == 0) an[0] = any_pending ? 1 : 0;
                if (lane_pending) {                                          // the first active lane is the first pending one; every pending lane stores its numbers
                    an[1] = uni(first ? a0[0] : a0[PP - 1]); an[2] = uni(first ? a1[0] : a1[PP - 1]);
                    an[3] = uni(first ? b0[0] : b0[PP - 1]); an[4] = uni(first ? b1[0] : b1[PP - 1]);
                }
                __syncthreads();                                             // also: the previous round's members are done reading jl[]
                const int (*aq)[8] = sh.anchor[round & 1];
                const int h0 = uni(aq[0][0]), h1 = uni(aq[1][0]), h2 = uni(aq[2][0]), h3 = uni(aq[3][0]);
                if (!(h0 | h1 | h2 | h3)) break;                             // uniform: nothing pending anywhere
                const int wv = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
                const int A0 = uni(aq[wv][1]), A1 = uni(aq[wv][2]), B0 = uni(aq[wv][3]), B1 = uni(aq[wv][4]);
                // the capacity-sized rectangle centred on the anchor's own box; its left edge on a multiple of 4 so that the
                // 16-byte staging applies (a pixel's box is at most 2R + 2 wide and high: the anchor itself always fits)
                const int rx0 = (A0 - (S::LW - 4 - (A1 - A0 + 1)) / 2) & ~3, rx1 = rx0 + S::LW - 1;
                const int ry0 = B0 - (S::LH - (B1 - B0 + 1)) / 2, ry1 = ry0 + S::LH - 1;
                bool member[PP];
#pragma unroll
                for (int p = 0; p < PP; ++p) member[p] = pending[p] && a0[p] >= rx0 && a1[p] <= rx1 && b0[p] >= ry0 && b1[p] <= ry1;
                stage_rect(rx0, rx1, ry0, ry1);
                __syncthreads();
                if (member[0] || member[PP - 1]) {
                    rows_and_solve(member);
#pragma unroll
                    for (int p = 0; p < PP; ++p) pending[p] = pending[p] && !member[p];
                }
            }
#pragma unroll 1
            for (int p = 0; p < PP; ++p) {
                if (!(p ? pending[PP - 1] : pending[0])) continue;
                // Leftover of a tile without coherent flows: the oracle's per-sample form for this pixel, the current frame read from
                // global memory (two clamped texel pairs per tap), the previous frame's records from the tile.  Small and slow on
                // purpose (it bounds the work of an incoherent tile at what the per-lane-gather kernel of rounds 2-3 cost); same
                // operands, same operations, same order as every other path.
                const bool second = PP == 2 && p != 0;
                float2 fl = second ? f[PP - 1] : f[0];
                float g0 = second ? gxx[PP - 1] : gxx[0], g1 = second ? gxy[PP - 1] : gxy[0], g2 = second ? gyy[PP - 1] : gyy[0];
                float bx = 0.0f, by = 0.0f;
                const int jpitch = U8 ? src_stride : w;
                auto jat = [&](size_t idx) -> float { if constexpr (U8) return (float)J8[idx]; else return J[idx]; };
                int xr = x, yr = y + p;
                const int lyp = ly + p;
                asm volatile("" : "+v"(xr), "+v"(yr));
    #pragma unroll 1
                for (int r = 0; r < N; ++r) {
                    float ay;
                    const int yi = lk_origin(lk_clampi(yr + r - RADIUS, 0, h - 1), fl.y, h, ay);
                    const size_t ra = (size_t)lk_clampi(yi, 0, h - 1) * jpitch, rb = (size_t)lk_clampi(yi + 1, 0, h - 1) * jpitch;
    #pragma unroll 1
                    for (int k = 0; k < N; ++k) {
                        float axk;
                        const int xi = lk_origin(lk_clampi(xr + k - RADIUS, 0, w - 1), fl.x, w, axk);
                        const int xa = lk_clampi(xi, 0, w - 1), xb = lk_clampi(xi + 1, 0, w - 1);
                        const float top = lk_lerp(jat(ra + xa), jat(ra + xb), axk);
                        const float bot = lk_lerp(jat(rb + xa), jat(rb + xb), axk);
                        const lk_f4 t = lk_lds_read4(&sh.tile[lyp + r][lx + k]);
                        const float d = t.x - lk_lerp(top, bot, ay);
                        lk_accum(t.y, d, bx);
                        lk_accum(t.z, d, by);
                        if (it == 0) { lk_accum(t.y, t.y, g0); lk_accum(t.y, t.z, g1); lk_accum(t.z, t.z, g2); }
                    }
                }
                fl = lk_solve(make_float4(g0, g1, g2, 0.0f), fl, bx, by);
                if (second) { f[PP - 1] = fl; gxx[PP - 1] = g0; gxy[PP - 1] = g1; gyy[PP - 1] = g2; }
                else { f[0] = fl; gxx[0] = g0; gxy[0] = g1; gyy[0] = g2; }
            }
        }
        if (it == 0) OFPS_LK_STAMP(4);
    }
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        if (active[p]) {
            int tidx = (int)threadIdx.x;
            asm volatile("" : "+v"(tidx));
            lk_store(f[p], x0 + tidx % kTX, y0 + tidx / kTX * PP + p, w, io);
        }
    }
    if (done_flag) {                                                     // uniform: children of this tile are waiting for it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this thread's write-through flow stores have been acknowledged
        __syncthreads();                                                 // ... everybody's
        if (threadIdx.x == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(done_flag), "v"(epoch) : "memory");
    }
    OFPS_LK_STAMP(5);
#undef OFPS_LK_STAMP
    return true;
}

// The whole pyramid in ONE launch.  Blocks are ordered coarsest level first (each level's range padded to a multiple of 8 so
// that a block's XCD -- block % 8 -- is the same seen from the launch and from its level, and lk_tile_of_block's XCD-contiguous
// raster runs apply inside the range); a tile starts once its parent tile has published its flows (lk_level_body).  Launched
// level by level the coarse levels leave most of the chip idle -- 510 and 2,040 tiles of a 1080p pyramid on 1,536 workgroup
// slots: 25 + 56 us for a quarter of level 0's work -- and each launch ends with a partly filled last round; in one launch
// the levels' tiles share the slots.
struct LkLevelArgs {
    const void* I; const void* J;
    int stride, w, h;
    LkFlowIO io;
    unsigned start, count;        // first block of the level, blocks of the level (a multiple of 8)
    int tiles_x, ntiles;
    unsigned flag_off;            // the level's tile flags inside LkLevelsArgs::flags (levels with children)
    int u8;                       // level 0: I / J are the u8 frames
};
struct LkLevelsArgs {
    int levels, iters, force_fall, wait_budget;
    uint32_t epoch;
    uint32_t* flags;
    int test_order;               // libofps_hip_testhooks.so only: 1 = blocks take the positions in reverse, 2 = permuted (test_mul, test_add)
    unsigned test_mul, test_add;
    unsigned long long* prof;
    LkLevelArgs lv[8];            // lv[0] = the coarsest level ... lv[levels - 1] = level 0
};
// A tile whose parent did not publish in time: the missing ancestors of tile (tx, ty) of level index k (lv[0] = the coarsest), then the tile
// itself, computed by this workgroup.  Thread 0 climbs from the grandparent up while the flags are not this launch's (one look each), then
// the tiles are made coarsest first -- each one's parent is done by then -- and published like any other.  Not inlined, and called in
// TAIL position (lk_levels_kernel has nothing to do afterwards): the cold path costs the level kernel neither registers nor spills around a
// call -- a first form that called a helper from inside lk_level_body and went on cost 2.8 % (SGPR spills 8 -> 72: profiles/r06/
// lk_forward_progress.txt).  A: the kernel's own argument segment (the level table is read from there).
template <int RADIUS>
__device__ __noinline__ void lk_help_tile(LkStepShared<RADIUS>* sh, const LkLevelsArgs* A, int k, int tx, int ty) {
    uint32_t* const flags = A->flags;
    const uint32_t epoch = A->epoch;
    __shared__ int s_from;
    if (threadIdx.x == 0) {
        int from = k - 1;                                                // (the parent: the caller's wait has just expired)
        for (int j = k - 2; j >= 0; --j) {
            const uint32_t* fl = flags + A->lv[j].flag_off + (size_t)(ty >> (k - j)) * A->lv[j].tiles_x + (tx >> (k - j));
            uint32_t v;
            asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(fl) : "memory");
            if (v == epoch) break;
            from = j;
        }
        s_from = from;
        atomicAdd(flags, (uint32_t)(k - from));                          // diagnostics (ofps_hip_lk_helped_tiles): tiles computed by a waiting child
    }
    __syncthreads();
    const int from = s_from;
    for (int j = from; j <= k; ++j) {
        const int ax = tx >> (k - j), ay = ty >> (k - j);
        uint32_t* done = j < A->levels - 1 ? flags + A->lv[j].flag_off + (size_t)ay * A->lv[j].tiles_x + ax : nullptr;
        LkFlowIO io = A->lv[j].io;
        if (A->lv[j].u8)
            lk_level_body<RADIUS, true>(*sh, A->lv[j].I, A->lv[j].J, A->lv[j].stride, A->lv[j].w, A->lv[j].h, A->iters, io, nullptr, A->force_fall, ax, ay,
                                        nullptr, done, epoch, 0, 0);
        else
            lk_level_body<RADIUS, false>(*sh, A->lv[j].I, A->lv[j].J, A->lv[j].stride, A->lv[j].w, A->lv[j].h, A->iters, io, nullptr, A->force_fall, ax, ay,
                                         nullptr, done, epoch, 0, 0);
        __syncthreads();                                                 // (everybody is done with `sh`)
    }
}

// block -> (level index, tile); false: a padding block of a level's range.  delay: the test hook's start delay (first call only)
__device__ __forceinline__ bool lk_position(const LkLevelsArgs& A, bool delay, int* k_out, int* tx_out, int* ty_out) {
    unsigned b = blockIdx.x;
#ifdef OFPS_HIP_TEST_HOOKS
    // OFPS_HIP_LK_TEST_ORDER: the launch as another dispatcher would run it -- 1: the blocks take the positions in REVERSE (level 0's tiles
    // first, the coarsest level last), 2: a pseudo-random permutation of the positions, and every workgroup starts after a pseudo-random delay
    if (A.test_order == 1) b = gridDim.x - 1 - b;
    else if (A.test_order == 2) b = (unsigned)(((unsigned long long)b * A.test_mul + A.test_add) % gridDim.x);
    if (A.test_order && delay) {
        unsigned hsh = blockIdx.x * 2654435761u;
        hsh ^= hsh >> 15;
        for (unsigned i = 0; i < (hsh & 63u); ++i) __builtin_amdgcn_s_sleep(64);
    }
#endif
    int k = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) k += (i < A.levels && b >= A.lv[i].start) ? 1 : 0;
    const LkLevelArgs& L = A.lv[k];
    const unsigned local = b - L.start;
    const int t = (int)((local % 8u) * (L.count / 8u) + local / 8u);
    if (t >= L.ntiles) return false;
    const int ty = t / L.tiles_x;
    *k_out = k; *ty_out = ty; *tx_out = t - ty * L.tiles_x;
    return true;
}

template <int RADIUS>
__global__ __launch_bounds__(256, LkStepShared<RADIUS>::WAVES_PER_SIMD) void lk_levels_kernel(const LkLevelsArgs A) {
    __shared__ LkStepShared<RADIUS> sh;
    int k, tx, ty;
    if (!lk_position(A, true, &k, &tx, &ty)) return;
    const LkLevelArgs& L = A.lv[k];
    // the parent: the tile of the next coarser level (lv[k - 1]) that holds this tile's half-resolution pixels
    const uint32_t* parent = k > 0 ? A.flags + A.lv[k - 1].flag_off + (size_t)(ty / 2) * A.lv[k - 1].tiles_x + tx / 2 : nullptr;
    uint32_t* done = k < A.levels - 1 ? A.flags + L.flag_off + (size_t)ty * L.tiles_x + tx : nullptr;
    bool ok;
    if (L.u8)
        ok = lk_level_body<RADIUS, true>(sh, L.I, L.J, L.stride, L.w, L.h, A.iters, L.io, A.prof, A.force_fall, tx, ty, parent, done, A.epoch, A.wait_budget, k);
    else
        ok = lk_level_body<RADIUS, false>(sh, L.I, L.J, L.stride, L.w, L.h, A.iters, L.io, nullptr, A.force_fall, tx, ty, parent, done, A.epoch, A.wait_budget, k);
    if (!ok) {                                                           // uniform; rare
        // (the position is made AGAIN here instead of being kept alive across the tile: the cold path must not cost the hot one registers)
        int k2, tx2, ty2;
        lk_position(A, false, &k2, &tx2, &ty2);
        lk_help_tile<RADIUS>(&sh, (const LkLevelsArgs*)__builtin_amdgcn_kernarg_segment_ptr(), k2, tx2, ty2);     // the kernel's only argument is at offset 0
    }
}

// cv-decoder/src/lib.rs:239-243,262-269: per-pixel records, raster order
__global__ __launch_bounds__(256) void lk_entries_kernel(const float2* __restrict__ flow, int W, int H, float nx, float ny,
                                                         float4* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float2 f = flow[(size_t)y * W + x];
    out[(size_t)y * W + x] = make_float4(((float)x + 0.5f) * nx, ((float)y + 0.5f) * ny, f.x * nx, f.y * ny);
}

static dim3 lk_grid(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }
// 1-D grid of the XCD-mapped kernels (lk_tile_of_block): the tile count rounded up to a multiple of 8
static dim3 lk_grid_xcd(int w, int h, int tx = 64, int ty = 4) {
    const unsigned n = (unsigned)((w + tx - 1) / tx) * (unsigned)((h + ty - 1) / ty);
    return dim3((n + 7) / 8 * 8);
}

// d_prev/d_cur: u8 luma on the device.  d_flow: W*H float2.  Workspace comes from the context.
// d_flow (W*H float2) and/or d_entries (W*H float4 records) receive the result; at least one of them.
int lk_flow_device(ofps_hip_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_cur, int W, int H, int stride, int levels,
                   int radius, int iters, float2* d_flow, float4* d_entries, const float2* d_init = nullptr) {
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && stride >= W, "lk_flow: bad geometry W=%d H=%d stride=%d", W, H, stride);
    OFPS_REQUIRE(ctx, levels >= 1 && levels <= 8 && radius >= 1 && radius <= 15 && iters >= 1 && iters <= 64,
                 "lk_flow: levels=%d radius=%d iters=%d out of range", levels, radius, iters);
    hipStream_t s = ctx->stream;
    int ws[8], hs[8];
    size_t off[9];
    ws[0] = W; hs[0] = H; off[0] = 0;
    for (int l = 1; l < levels; ++l) { ws[l] = (ws[l - 1] + 1) / 2; hs[l] = (hs[l - 1] + 1) / 2; }
    for (int l = 0; l < levels; ++l) off[l + 1] = off[l] + (size_t)ws[l] * hs[l];
    // every segment is rounded up to a multiple of 4 floats so that G (float4) and the flow planes (float2) stay
    // naturally aligned inside the workspace whatever the frame size
    const size_t plane0 = ((size_t)W * H + 3) & ~size_t(3), pyr = (off[levels] + 3) & ~size_t(3);
    // layout: I pyramid | J pyramid | gx pyramid | gy pyramid | G pyramid (float4) | flowA (float2) | flowB (float2): the
    // gradients and tensors of all levels are formed up front (they do not depend on the flow), one launch each
    const size_t floats = 2 * pyr + 2 * pyr + 4 * pyr + 2 * plane0 + 2 * plane0;
    auto* base = static_cast<float*>(scratch(ctx, S_WORK0, floats * sizeof(float)));
    if (!base) return OFPS_HIP_ENOMEM;
    float* Ip = base; float* Jp = Ip + pyr; float* gxp = Jp + pyr; float* gyp = gxp + pyr;
    float4* Gp = reinterpret_cast<float4*>(gyp + pyr);
    float2* fa = reinterpret_cast<float2*>(reinterpret_cast<float*>(Gp) + 4 * pyr);
    float2* fb = fa + plane0;

    const int tile_rows = lk_tile_rows(radius);
    const size_t tiles0 = (size_t)((W + kTX - 1) / kTX) * (size_t)((H + tile_rows - 1) / tile_rows);     // tiles of the tiled kernels at level 0
    const bool tiled = radius == 2 || radius == 4 || radius == 6;      // kernels that run a whole level and fold the upsample / record passes in
    unsigned long long* prof = nullptr;                           // OFPS_HIP_LK_PROF=1: phase table of level 0
    if (tiled && ctx->opt.lk_prof) {
        prof = static_cast<unsigned long long*>(scratch(ctx, S_WORK2, tiles0 * 6 * sizeof(unsigned long long)));
        if (!prof) return OFPS_HIP_ENOMEM;
    }
    int pyr_from = 2;                                             // first level the per-level pyramid launches below still have to make
    if (tiled && levels >= 3) {                                   // levels 1 and 2 from the u8 frames in one launch
        dim3 g2 = lk_grid_xcd(ws[2], hs[2], kP0X, kP0Y); g2.z = 2;
        hipLaunchKernelGGL(lk_pyr12_kernel, g2, dim3(256), 0, s, d_prev, d_cur, W, H, stride, Ip + off[1], Jp + off[1], ws[1], hs[1],
                           Ip + off[2], Jp + off[2], ws[2], hs[2]);
        pyr_from = 3;
    } else if (levels >= 2) {                                     // level 1 from the u8 frames
        // (the tiled path's level 0 works on the u8 frames themselves: no f32 level-0 planes, no level-0 gradient planes)
        dim3 g2 = lk_grid_xcd(ws[1], hs[1], kP0X, kP0Y); g2.z = 2;
        hipLaunchKernelGGL(lk_pyr0_kernel<uint8_t>, g2, dim3(256), 0, s, d_prev, d_cur, W, H, stride, tiled ? (float*)nullptr : Ip,
                           tiled ? (float*)nullptr : Jp, Ip + off[1], Jp + off[1], ws[1], hs[1], tiled ? (float*)nullptr : gxp,
                           tiled ? (float*)nullptr : gyp);
    } else if (!tiled) {
        dim3 g2 = lk_grid(W, H); g2.z = 2;
        hipLaunchKernelGGL(lk_u8_to_f32_pair_kernel, g2, dim3(256), 0, s, d_prev, d_cur, W, H, stride, Ip, Jp);
    }
    for (int l = pyr_from; l < levels; ++l) {                     // level l from level l-1 (the untiled path's gradients of level l-1 come out of the same window)
        dim3 g2 = lk_grid_xcd(ws[l], hs[l], kP0X, kP0Y); g2.z = 2;
        hipLaunchKernelGGL(lk_pyr0_kernel<float>, g2, dim3(256), 0, s, (const float*)(Ip + off[l - 1]), (const float*)(Jp + off[l - 1]),
                           ws[l - 1], hs[l - 1], ws[l - 1], (float*)nullptr, (float*)nullptr, Ip + off[l], Jp + off[l], ws[l], hs[l],
                           tiled ? (float*)nullptr : gxp + off[l - 1], tiled ? (float*)nullptr : gyp + off[l - 1]);   // (the tiled level kernels make their gradients from the I window)
    }
    float2* cur_flow = fa;
    float2* other = fb;
    float2* plain_flow = d_flow;                                       // the untiled path always produces a flow plane
    if (!tiled && !plain_flow) {
        plain_flow = static_cast<float2*>(scratch(ctx, S_WORK1, plane0 * sizeof(float2)));
        if (!plain_flow) return OFPS_HIP_ENOMEM;
    }
    const int force_fall = ctx->opt.test_lk_fall;       // -1 in the product library (a test hook of libofps_hip_testhooks.so)
    if (!tiled) {   // run-time-radius path: gradients of the coarsest level (the pyramid kernels wrote the finer ones), 64 x 4 pixels per block
        LkPyr P{};
        P.levels = levels;
        unsigned nb = 0;
        for (int l = 0; l < levels; ++l) {
            P.w[l] = ws[l]; P.h[l] = hs[l]; P.off[l] = (unsigned)off[l]; P.start[l] = nb;
            if (l == levels - 1) nb += lk_grid_xcd(ws[l], hs[l], 64, 4).x;
        }
        P.start[levels] = nb;
        if (nb) hipLaunchKernelGGL(lk_grad_all_kernel, dim3(nb), dim3(256), 0, s, Ip, P, gxp, gyp);
        // (the tiled path has neither gradient planes nor tensor planes: its level kernels make the gradients from the window
        // they stage and sum the structure tensor inside the level's first step)
    }
    if (tiled) {
        // ---- the whole pyramid in one launch (lk_levels_kernel).  Flow planes of the levels above 0: a packed pyramid inside `fa`
        // (every level its own plane: all levels are in flight together)
        LkLevelsArgs A{};
        A.levels = levels; A.iters = iters; A.force_fall = force_fall; A.prof = prof;
        A.wait_budget = ctx->opt.test_lk_wait_budget;          // 0 in the product library (a test hook of libofps_hip_testhooks.so)
        unsigned nb = 0, nflags = 2;                                  // word 0 of the flag buffer counts the tiles a waiting child computed itself (ofps_hip_lk_helped_tiles); word 1 spare
        for (int k = 0; k < levels; ++k) {                            // k = 0: the coarsest level
            const int l = levels - 1 - k;
            LkLevelArgs& L = A.lv[k];
            const bool last = l == 0;
            L.I = last ? (const void*)d_prev : (const void*)(Ip + off[l]);
            L.J = last ? (const void*)d_cur : (const void*)(Jp + off[l]);
            L.stride = last ? stride : ws[l]; L.w = ws[l]; L.h = hs[l];
            L.tiles_x = (ws[l] + kTX - 1) / kTX; L.ntiles = L.tiles_x * ((hs[l] + tile_rows - 1) / tile_rows);
            L.start = nb; L.count = ((unsigned)L.ntiles + 7u) / 8u * 8u; nb += L.count;
            L.flag_off = nflags; if (!last) nflags += (unsigned)L.ntiles;
            L.u8 = last ? 1 : 0;
            LkFlowIO& io = L.io;
            io.coarse = l == levels - 1 ? nullptr : fa + (off[l + 1] - off[1]);
            io.coarse_shared = 1;
            io.init = l == levels - 1 ? d_init : nullptr;
            io.w1 = l + 1 < levels ? ws[l + 1] : ws[l]; io.h1 = l + 1 < levels ? hs[l + 1] : hs[l];
            io.flow_out = last ? d_flow : fa + (off[l] - off[1]);      // the last level skips the flow plane nobody asked for
            io.out_shared = last ? 0 : 1;
            io.out_entries = last ? d_entries : nullptr;
            io.nx = 1.0f / (float)W; io.ny = 1.0f / (float)H;
        }
        // tile flags carry the launch's epoch: no clearing between calls (zeroed when (re)allocated or when the counter wraps)
        auto* flags = static_cast<uint32_t*>(scratch(ctx, S_LK_FLAGS, (size_t)nflags * sizeof(uint32_t)));
        if (!flags) return OFPS_HIP_ENOMEM;
        if (ctx->lk_flags_gen != ctx->scratch[S_LK_FLAGS].gen || ctx->lk_epoch == 0xFFFFFFFFu) {
            OFPS_HIP_TRY(ctx, hipMemsetAsync(flags, 0, ctx->scratch[S_LK_FLAGS].cap, s));
            ctx->lk_flags_gen = ctx->scratch[S_LK_FLAGS].gen;
            ctx->lk_epoch = 0;
        }
        A.epoch = ++ctx->lk_epoch;
        A.flags = flags;
        A.test_order = ctx->opt.test_lk_order;                   // 0 in the product library
        if (A.test_order == 2) {                                 // b -> (b * mul + add) mod nb with gcd(mul, nb) = 1: a permutation of the positions
            unsigned mul = 40503u % nb;
            auto gcd = [](unsigned a, unsigned b) { while (b) { const unsigned t = a % b; a = b; b = t; } return a; };
            while (mul < 2 || gcd(mul, nb) != 1) ++mul;
            A.test_mul = mul; A.test_add = A.epoch * 7919u % nb;
        }
        auto launch = [&](const LkLevelsArgs& B, unsigned blocks) {
            switch (radius) {
                case 2: hipLaunchKernelGGL(lk_levels_kernel<2>, dim3(blocks), dim3(256), 0, s, B); break;
                case 4: hipLaunchKernelGGL(lk_levels_kernel<4>, dim3(blocks), dim3(256), 0, s, B); break;
                default: hipLaunchKernelGGL(lk_levels_kernel<6>, dim3(blocks), dim3(256), 0, s, B); break;
            }
        };
        if (ctx->opt.lk_serial) {
            // one launch per level, coarsest first: a level's parents are complete before its launch starts, nothing waits on a
            // flag.  OFPS_HIP_LK_SERIAL, for A/B runs.
            for (int k = 0; k < levels; ++k) {
                LkLevelsArgs B = A;
                B.levels = 1;
                B.lv[0] = A.lv[k];
                B.lv[0].start = 0;
                B.test_order = 0;
                launch(B, A.lv[k].count);
            }
        } else {
            launch(A, nb);
        }
    }
    for (int l = levels - 1; l >= 0 && !tiled; --l) {
        const int w = ws[l], h = hs[l];
        float* gx = gxp + off[l]; float* gy = gyp + off[l];
        float4* G = Gp + off[l];
        hipLaunchKernelGGL(lk_tensor_kernel, lk_grid(w, h), dim3(256), 0, s, gx, gy, w, h, radius, G);
        if (l == levels - 1) {
            if (d_init) OFPS_HIP_TRY(ctx, hipMemcpyAsync(cur_flow, d_init, (size_t)w * h * sizeof(float2), hipMemcpyDeviceToDevice, s));
            else OFPS_HIP_TRY(ctx, hipMemsetAsync(cur_flow, 0, (size_t)w * h * sizeof(float2), s));
        } else {
            hipLaunchKernelGGL(lk_upsample_kernel, lk_grid(w, h), dim3(256), 0, s, cur_flow, ws[l + 1], hs[l + 1], other, w, h);
            float2* t = cur_flow; cur_flow = other; other = t;
        }
        for (int it = 0; it < iters; ++it) {
            const bool last = l == 0 && it == iters - 1;
            hipLaunchKernelGGL(lk_step_kernel<0>, lk_grid(w, h), dim3(256), 0, s, Ip + off[l], Jp + off[l], gx, gy, G, w, h, radius,
                               cur_flow, last ? plain_flow : other);
            if (!last) { float2* t = cur_flow; cur_flow = other; other = t; }
        }
    }
    if (!tiled && d_entries)
        hipLaunchKernelGGL(lk_entries_kernel, lk_grid(W, H), dim3(256), 0, s, plain_flow, W, H, 1.0f / (float)W, 1.0f / (float)H, d_entries);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    if (prof) {
        std::vector<unsigned long long> hst(tiles0 * 6);
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(hst.data(), prof, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        OFPS_HIP_TRY(ctx, hipStreamSynchronize(s));
        double ph[5] = {0, 0, 0, 0, 0};
        unsigned long long t_min = ~0ull, t_max = 0;
        for (size_t b = 0; b < tiles0; ++b) {
            const unsigned long long* r = hst.data() + b * 6;
            for (int k = 0; k < 5; ++k) ph[k] += (double)(r[k + 1] - r[k]);
            t_min = r[0] < t_min ? r[0] : t_min; t_max = r[5] > t_max ? r[5] : t_max;
        }
        fprintf(stderr, "[lk prof] level 0, %zu tiles: cycles per workgroup  stage-tile+origins %.0f  barrier %.0f  stage-J %.0f  "
                        "rows+solve %.0f  (first step)   later steps+store %.0f   kernel span %.0f cycles\n", tiles0, ph[0] / tiles0, ph[1] / tiles0, ph[2] / tiles0,
                ph[3] / tiles0, ph[4] / tiles0, (double)(t_max - t_min));
    }
    return OFPS_HIP_OK;
}

}  // namespace ofps

extern "C" {

int ofps_hip_lk_spec_revision(void) { return OFPS_LK_SPEC_FMA ? 2 : 1; }

int ofps_hip_lk_helped_tiles(ofps_hip_ctx* ctx, uint64_t* count) {
    if (!ctx || !count) return OFPS_HIP_EINVAL;
    *count = 0;
    const void* d = ctx->scratch[ofps::S_LK_FLAGS].p;
    if (!d || !ctx->lk_flags_gen) return OFPS_HIP_OK;            // no pyramid launch on this context yet
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t v = 0;
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(&v, d, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *count = v;
    return OFPS_HIP_OK;
}

static int lk_flow_dev_call(ofps_hip_ctx* ctx, const void* d_prev, const void* d_cur, int W, int H, int stride, int levels, int radius,
                            int iters, const void* d_init_flow, void* d_out_flow, void* d_out_entries) {
    return ofps::lk_flow_device(ctx, static_cast<const uint8_t*>(d_prev), static_cast<const uint8_t*>(d_cur), W, H, stride, levels, radius, iters,
                                static_cast<float2*>(d_out_flow), static_cast<float4*>(d_out_entries), static_cast<const float2*>(d_init_flow));
}

int ofps_hip_lk_flow_dev(ofps_hip_ctx* ctx, const void* d_prev, const void* d_cur, int W, int H, int stride, int levels,
                         int radius, int iters, void* d_out_flow, void* d_out_entries) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_prev && d_cur && (d_out_flow || d_out_entries), "lk_flow: null device pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lk_flow_dev_call(ctx, d_prev, d_cur, W, H, stride, levels, radius, iters, nullptr, d_out_flow, d_out_entries);
}

int ofps_hip_lk_flow_init_dev(ofps_hip_ctx* ctx, const void* d_prev, const void* d_cur, int W, int H, int stride, int levels,
                              int radius, int iters, const void* d_init_flow, void* d_out_flow, void* d_out_entries) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_prev && d_cur && (d_out_flow || d_out_entries), "lk_flow_init: null device pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lk_flow_dev_call(ctx, d_prev, d_cur, W, H, stride, levels, radius, iters, d_init_flow, d_out_flow, d_out_entries);
}

// The body of a "hip_lk" Decoder::process_frame (cv-decoder/src/lib.rs:82-294): dense flow, per-pixel records,
// optionally filtered by the contrast mask of :203-237 (OFPS_HIP_LK_CONTRAST_MASK, computed on `cur` like the
// reference's `self.gray`), then either down-sampled through the densifier to the (max_w, max_h)-capped grid of
// :98-121 with one record per visited cell in BTreeSet<(x,y)> order ("Process Fullres" = true, the default), or -- with
// OFPS_HIP_LK_REDUCED, "Process Fullres" = false -- computed on frames resized to that grid first (frontend.hip) and returned per pixel
// of the reduced frame in raster order (the `mf.push` branch, :274-276).  Only the final records leave the device.
}  // extern "C"

namespace {

struct LkGrid {                 // what a process_frame returns for this geometry / these flags
    int gw = 0, gh = 0;         // the record grid
    int fw = 0, fh = 0;         // the frames the mask and the flow run on: W x H, or the record grid itself with OFPS_HIP_LK_REDUCED
    bool per_pixel = false;     // one record per (unmasked) pixel of the fw x fh frames (OFPS_HIP_LK_FULLRES_RECORDS, OFPS_HIP_LK_REDUCED)
    bool use_mask = false;
    bool farneback = false;     // OFPS_HIP_FLOW_FARNEBACK: the flow is farneback.hip's (the "hip_flow" decoder), not the iterative LK
    bool use_previous = false;  // OFPS_HIP_FLOW_USE_PREVIOUS: the stream's previous flow is the initial flow (cv-decoder/src/lib.rs:161-165)
    bool reduced = false;       // OFPS_HIP_LK_REDUCED: cv-decoder's "Process Fullres" = false (:124-133,274-276)
    int fmt = OFPS_HIP_FMT_LUMA, cn = 1;      // the arriving frames' pixel format (bits 8-9 of the flags)
    bool frontend = false;      // the arriving frames are resized and / or converted behind their upload (frontend.hip)
    size_t raw_row = 0;         // bytes per row of the staged arriving frame (W * cn rounded up to 4)
    size_t max_records = 0;     // capacity the records need
};

constexpr unsigned kLkFlagBits = OFPS_HIP_LK_CONTRAST_MASK | OFPS_HIP_LK_FULLRES_RECORDS | OFPS_HIP_FLOW_FARNEBACK | OFPS_HIP_FLOW_USE_PREVIOUS |
                                 OFPS_HIP_LK_REDUCED | OFPS_HIP_FRAME_FORMAT_MASK;

int lk_grid_of(ofps_hip_ctx* ctx, int W, int H, int stride, int max_w, int max_h, unsigned flags, LkGrid* g) {
    g->use_mask = flags & OFPS_HIP_LK_CONTRAST_MASK;
    g->farneback = flags & OFPS_HIP_FLOW_FARNEBACK;
    g->use_previous = flags & OFPS_HIP_FLOW_USE_PREVIOUS;
    g->reduced = flags & OFPS_HIP_LK_REDUCED;
    const bool fullres_records = flags & OFPS_HIP_LK_FULLRES_RECORDS;
    g->fmt = (int)((flags & OFPS_HIP_FRAME_FORMAT_MASK) >> 8);
    g->cn = ofps::frame_format_channels(g->fmt);
    OFPS_REQUIRE(ctx, !g->use_previous || g->farneback, "OFPS_HIP_FLOW_USE_PREVIOUS without OFPS_HIP_FLOW_FARNEBACK (the iterative LK has no initial flow across pairs)");
    OFPS_REQUIRE(ctx, !(g->reduced && fullres_records), "OFPS_HIP_LK_REDUCED with OFPS_HIP_LK_FULLRES_RECORDS: the reduced mode has no full-resolution flow");
    OFPS_REQUIRE(ctx, stride >= W * g->cn, "dense decoder: stride %d < %d bytes per row (%d x %d channels)", stride, W * g->cn, W, g->cn);
    int gw = 0, gh = 0;
    ofps::cv_grid(W, H, max_w, max_h, &gw, &gh);            // cv-decoder/src/lib.rs:98-121 with aspect_ratio_scale = (1, 1)
    g->per_pixel = fullres_records || g->reduced;
    if (g->reduced) OFPS_REQUIRE(ctx, gw >= 1 && gh >= 1, "dense decoder: the capped grid of %dx%d under (%d, %d) is empty", W, H, max_w, max_h);
    else if (!fullres_records) OFPS_REQUIRE(ctx, gw >= 1 && gh >= 1 && (size_t)gw * gh <= 65536, "lk_decode: field %dx%d unsupported", gw, gh);
    g->fw = g->reduced ? gw : W; g->fh = g->reduced ? gh : H;
    g->gw = fullres_records ? W : gw; g->gh = fullres_records ? H : gh;
    g->max_records = (size_t)g->gw * g->gh;
    g->frontend = g->reduced || g->fmt != OFPS_HIP_FMT_LUMA;
    g->raw_row = ((size_t)W * g->cn + 3) & ~(size_t)3;
    return OFPS_HIP_OK;
}

// The flow's own parameter limits, checked where a stream's FIRST frame is pushed (it runs no flow) and before any upload: a stream must not
// accept a frame and then fail every later one.  Farneback: winsize = 2 radius + 1 <= 15, and the layers of the fw x fh frames.
int lk_check_flow_params(ofps_hip_ctx* ctx, const LkGrid& g, int levels, int radius, int iters, const char* who) {
    OFPS_REQUIRE(ctx, levels >= 1 && levels <= 8 && radius >= 1 && radius <= 15 && iters >= 1 && iters <= 64,
                 "%s: levels=%d radius=%d iters=%d out of range", who, levels, radius, iters);
    if (g.farneback) {
        const int rc = ofps::farneback_check_params(ctx, g.fw, g.fh, levels, 2 * radius + 1, 7);
        if (rc != OFPS_HIP_OK) return rc;
    }
    return OFPS_HIP_OK;
}

// the arriving frame (host memory) -> the fw x fh luma frame the flow reads, on stream `up`: a plain upload, or upload into `d_raw` + front-end
int lk_upload_frame(ofps_hip_ctx* ctx, const LkGrid& g, const uint8_t* frame, int W, int H, int stride, uint8_t* d_raw, uint8_t* d_dst, hipStream_t up) {
    if (!g.frontend) {
        OFPS_HIP_TRY(ctx, ofps::upload_rows(d_dst, W, frame, stride, W, H, up));
        return OFPS_HIP_OK;
    }
    // A strongly reduced frame needs two of every H / fh source rows (168 of 1080 at the default cap): when the caller's frame is page-locked
    // -- device-addressable -- the front-end gathers them straight from host memory (~1 MB over PCIe instead of the 6.2 MB of a 1080p BGR
    // frame; the pixels of a row are 38 bytes apart, so the touched rows cross whole) and nothing is staged.  The frame must stay valid until
    // the ticket is collected, which the read-ahead form asks for anyway (include/ofps_hip.h).
    void* mapped = nullptr;
    if (g.reduced && g.fh * 3 <= H && ofps::device_address_of(frame, &mapped))
        return ofps::frontend_device(ctx, static_cast<const uint8_t*>(mapped), W, H, stride, g.fmt, true, d_dst, g.fw, g.fh, up);
    OFPS_HIP_TRY(ctx, ofps::upload_rows(d_raw, g.raw_row, frame, stride, (size_t)W * g.cn, H, up));
    return ofps::frontend_device(ctx, d_raw, W, H, (int)g.raw_row, g.fmt, true, d_dst, g.fw, g.fh, up);
}

// records 0 .. *d_count - 1 (or n_max when d_count is null) to a device-addressable destination, the count to cnt_dst
__global__ __launch_bounds__(256) void lk_copy_records_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                              const uint32_t* __restrict__ d_count, size_t n_max, uint32_t* __restrict__ cnt_dst) {
    size_t n = d_count ? (size_t)*d_count : n_max;
    if (n > n_max) n = n_max;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (cnt_dst && blockIdx.x == 0 && threadIdx.x == 0) *cnt_dst = (uint32_t)n;
}

// Enqueues everything of a process_frame that follows the uploads on ctx->stream: flow [-> contrast mask] -> output stage.
// The record count lands at cnt_dst and the records at rec_dst -- device scratch, or the device address of a page-locked
// block (the kernels store there directly: no read-back launch of their own).
int lk_enqueue_frame(ofps_hip_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_cur, int W, int H, int levels, int radius, int iters,
                     const LkGrid& g, float4* rec_dst, uint32_t* cnt_dst, uint64_t prev_id = 0,
                     uint64_t cur_id = 0, const uint8_t* d_mask_ready = nullptr) {
    const size_t px = (size_t)W * H, cells = g.per_pixel ? 1 : (size_t)g.gw * g.gh;
    auto* d_ent = static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES, px * sizeof(float4)));
    auto* d_field = static_cast<float2*>(ofps::scratch(ctx, ofps::S_FIELD, cells * sizeof(float2)));
    auto* d_cnt = static_cast<uint32_t*>(ofps::scratch(ctx, ofps::S_RESULT, 16));
    if (!d_ent || !d_field || !d_cnt) return OFPS_HIP_ENOMEM;
    int rc;
    if (g.farneback) {
        // cv-decoder's call (cv-decoder/src/lib.rs:188-199): levels = pyramid levels, winsize = 2 * radius + 1, iters = iterations,
        // poly_n 7, poly_sigma 1.5.  prev_id / cur_id: the
        // stream's frame ids -- the first frame's pyramid + expansion are the previous call's (farneback.hip)
        // OFPS_HIP_FLOW_USE_PREVIOUS: the flow of the pair that ended with this pair's first frame is the initial flow, and this pair's flow is
        // kept for the next one (read by the coarsest layer's first kernel, written by the last kernel of the call: one buffer)
        float2* d_keep = nullptr;
        const float2* d_init = nullptr;
        ofps_hip_ctx::FbPrevFlow& pf = ctx->fb_prev_flow;
        if (g.use_previous && cur_id != 0) {
            d_keep = static_cast<float2*>(ofps::scratch(ctx, ofps::S_FB_FLOW, px * sizeof(float2)));
            if (!d_keep) return OFPS_HIP_ENOMEM;
            // (the stream's LAST flow, whichever frames it related: cv-decoder's self.flow persists across skipped reads, cv-decoder/src/lib.rs:161-165;
            // ofps_hip_lk_reset and a geometry change forget it, ofps_hip_lk_rewind does not)
            if (pf.valid && pf.W == W && pf.H == H && pf.gen == ctx->scratch[ofps::S_FB_FLOW].gen) d_init = d_keep;
        }
        if (d_keep) pf.valid = false;               // (until this call has enqueued everything; a call that keeps nothing leaves the buffer alone)
        rc = ofps::farneback_flow_device(ctx, d_prev, d_cur, W, H, W, levels, 2 * radius + 1, iters, 7, 1.5, d_init, d_keep, d_ent, prev_id, cur_id);
        if (rc != OFPS_HIP_OK) return rc;
        if (d_keep) { pf.valid = true; pf.id = cur_id; pf.W = W; pf.H = H; pf.gen = ctx->scratch[ofps::S_FB_FLOW].gen; }
    } else {
        rc = ofps::lk_flow_device(ctx, d_prev, d_cur, W, H, W, levels, radius, iters, nullptr, d_ent, nullptr);
        if (rc != OFPS_HIP_OK) return rc;
    }
    const uint8_t* d_mask = d_mask_ready;          // (stream forms: made on the upload's stream already, beside the previous pair's flow)
    if (g.use_mask && !d_mask) {
        auto* m = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_MASK, px));
        if (!m) return OFPS_HIP_ENOMEM;
        rc = ofps::contrast_mask_device(ctx, d_cur, W, H, W, m);
        if (rc != OFPS_HIP_OK) return rc;
        d_mask = m;
    }
    if (g.per_pixel) {
        const float4* d_rec = d_ent;
        const uint32_t* d_n = nullptr;
        if (g.use_mask && px <= ofps::kCompactSmallMax)      // a reduced frame's records: one single-workgroup launch, straight into the block
            return ofps::compact_small_device(ctx, d_ent, d_mask, px, rec_dst, cnt_dst);
        if (g.use_mask) {                                  // the masked records themselves: order-preserving compaction
            auto* d_ent2 = static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES2, px * sizeof(float4)));
            if (!d_ent2) return OFPS_HIP_ENOMEM;
            rc = ofps::compact_entries_device(ctx, d_ent, d_mask, px, d_ent2, d_cnt + 1);
            if (rc != OFPS_HIP_OK) return rc;
            d_rec = d_ent2; d_n = d_cnt + 1;
        }
        hipLaunchKernelGGL(lk_copy_records_kernel, dim3(1024), dim3(256), 0, ctx->stream, d_rec, rec_dst, d_n, px, cnt_dst);
        OFPS_HIP_TRY(ctx, hipGetLastError());
        return OFPS_HIP_OK;
    }
    // down-sampled output (cv-decoder/src/lib.rs:244-291): the records are this call's own per-pixel lattice, so the
    // densifier walks each cell's rectangle of pixels (masked ones skipped in place) instead of sorting 2 M records
    return ofps::densify_raster_entries_device(ctx, d_ent, d_mask, W, H, g.gw, g.gh, d_field, rec_dst, cnt_dst);
}

// a page-locked block [count, pad x 3][records]; grows, never shrinks
int lk_pinned_block(ofps_hip_ctx* ctx, void** p, size_t* cap, size_t max_records) {
    const size_t need = 16 + max_records * sizeof(float4);
    if (*cap >= need) return OFPS_HIP_OK;
    if (*p) { OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); OFPS_HIP_TRY(ctx, hipHostFree(*p)); *p = nullptr; *cap = 0; }
    OFPS_HIP_TRY(ctx, hipHostMalloc(p, need, OFPS_HIP_HOST_BLOCK_FLAGS));      // fine-grained: kernels write it, the host reads it after an event
    *cap = need;
    return OFPS_HIP_OK;
}

// count + records of a finished block -> the caller's buffer
void lk_collect(const void* pinned, size_t max_records, float* out_entries, size_t* n_out) {
    uint32_t cnt = 0;
    memcpy(&cnt, pinned, sizeof(cnt));
    if (cnt > max_records) cnt = (uint32_t)max_records;
    if (cnt) memcpy(out_entries, static_cast<const char*>(pinned) + 16, (size_t)cnt * sizeof(float4));
    *n_out = cnt;
}

int lk_stream_setup(ofps_hip_ctx* ctx) {
    if (ctx->lk_copy_stream) return OFPS_HIP_OK;
    OFPS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->lk_copy_stream, hipStreamNonBlocking));
    for (auto& t : ctx->lk_ticket) {
        OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
        OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&t.uploaded, hipEventDisableTiming));
    }
    return OFPS_HIP_OK;
}

// waits for every ticket in flight and forgets the stream position (reset / geometry change / reallocation of the ring)
int lk_stream_drain(ofps_hip_ctx* ctx) {
    for (auto& t : ctx->lk_ticket) {
        if (t.pending && t.done) OFPS_HIP_TRY(ctx, hipEventSynchronize(t.done));
        t.pending = false;
    }
    if (ctx->lk_copy_stream) OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->lk_copy_stream));
    ctx->lk_frames = 0;
    return OFPS_HIP_OK;
}

}  // namespace

extern "C" {

int ofps_hip_lk_decode(ofps_hip_ctx* ctx, const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels,
                       int radius, int iters, int max_w, int max_h, unsigned flags, float* out_entries, size_t* n_out,
                       int* out_w, int* out_h) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, prev && cur && out_entries && n_out, "lk_decode: null host pointer");
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && max_w >= 1 && max_h >= 1, "lk_decode: bad geometry");
    OFPS_REQUIRE(ctx, (flags & ~kLkFlagBits) == 0, "lk_decode: unknown flags 0x%x", flags);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    LkGrid g;
    int rc = lk_grid_of(ctx, W, H, stride, max_w, max_h, flags, &g);
    if (rc != OFPS_HIP_OK) return rc;
    rc = lk_check_flow_params(ctx, g, levels, radius, iters, "lk_decode");
    if (rc != OFPS_HIP_OK) return rc;
    const size_t px = (size_t)g.fw * g.fh, raw_bytes = g.raw_row * (size_t)H;
    auto* d_frames = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_FRAMES, 2 * px));
    auto* d_raw = g.frontend ? static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_FE_RAW_PAIR, 2 * raw_bytes)) : nullptr;
    if (!d_frames || (g.frontend && !d_raw)) return OFPS_HIP_ENOMEM;
    rc = lk_upload_frame(ctx, g, prev, W, H, stride, d_raw, d_frames, ctx->stream);
    if (rc != OFPS_HIP_OK) return rc;
    rc = lk_upload_frame(ctx, g, cur, W, H, stride, d_raw ? d_raw + raw_bytes : nullptr, d_frames + px, ctx->stream);
    if (rc != OFPS_HIP_OK) return rc;
    // the count and the records come back in ONE page-locked block and one wait; only the visited cells' records reach the
    // caller's buffer
    rc = lk_pinned_block(ctx, &ctx->lk_pinned, &ctx->lk_pinned_cap, g.max_records);
    if (rc != OFPS_HIP_OK) return rc;
    void* mapped = nullptr;
    OFPS_REQUIRE(ctx, ofps::device_address_of(ctx->lk_pinned, &mapped), "lk_decode: page-locked block is not device-addressable");
    rc = lk_enqueue_frame(ctx, d_frames, d_frames + px, g.fw, g.fh, levels, radius, iters, g, reinterpret_cast<float4*>(static_cast<char*>(mapped) + 16),
                          static_cast<uint32_t*>(mapped));
    if (rc != OFPS_HIP_OK) return rc;
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    lk_collect(ctx->lk_pinned, g.max_records, out_entries, n_out);
    if (out_w) *out_w = g.gw;
    if (out_h) *out_h = g.gh;
    return OFPS_HIP_OK;
}

int ofps_hip_lk_push_frame_async(ofps_hip_ctx* ctx, const uint8_t* frame, int W, int H, int stride, int levels, int radius, int iters,
                                 int max_w, int max_h, unsigned flags, int* ticket) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, frame && ticket, "lk_push_frame_async: null pointer");
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && max_w >= 1 && max_h >= 1, "lk_push_frame_async: bad geometry");
    OFPS_REQUIRE(ctx, (flags & ~kLkFlagBits) == 0, "lk_push_frame_async: unknown flags 0x%x", flags);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = lk_stream_setup(ctx);
    if (rc != OFPS_HIP_OK) return rc;
    LkGrid g;
    rc = lk_grid_of(ctx, W, H, stride, max_w, max_h, flags, &g);
    if (rc != OFPS_HIP_OK) return rc;
    // (a stream's first frame runs no flow: the flow's parameters -- Farneback's own limits included -- are refused here, not one frame later)
    rc = lk_check_flow_params(ctx, g, levels, radius, iters, "lk_push_frame_async");
    if (rc != OFPS_HIP_OK) return rc;
    const long tno = ctx->lk_next_ticket;
    auto& t = ctx->lk_ticket[tno % ofps_hip_ctx::kLkTickets];
    OFPS_REQUIRE(ctx, !t.pending, "lk_push_frame_async: ticket %ld has not been collected (at most %d frames in flight)",
                 tno - ofps_hip_ctx::kLkTickets, ofps_hip_ctx::kLkTickets);
    const size_t px = (size_t)g.fw * g.fh, raw_bytes = g.raw_row * (size_t)H;      // the ring holds the frames the flow reads (reduced / converted)
    // A new geometry restarts the stream and may reallocate the ring.  With a ticket in flight that would throw its records away
    // (draining marks it collected: the caller's later lk_frame_wait would fail with "already collected"): refused, like the
    // multi-device form does -- collect first, or ofps_hip_lk_reset.
    const bool other_pending = ctx->lk_ticket[(tno + 1) % ofps_hip_ctx::kLkTickets].pending;
    // (so does a change of the front-end -- "Process Fullres" flipped, another pixel format: the ring's frames are the other mode's; cv-decoder
    // returns Ok(false) for that frame because gray and old_gray differ in size, cv-decoder/src/lib.rs:156-158)
    const bool restart = ctx->lk_w != W || ctx->lk_h != H || ctx->lk_fw != g.fw || ctx->lk_fh != g.fh || ctx->lk_fmt != g.fmt ||
                         ctx->scratch[ofps::S_LK_FRAMES].cap < ofps_hip_ctx::kLkSlots * px ||
                         (g.frontend && ctx->scratch[ofps::S_FE_RAW].cap < ofps_hip_ctx::kLkTickets * raw_bytes);
    OFPS_REQUIRE(ctx, !(restart && other_pending), "lk_push_frame_async: geometry change %dx%d -> %dx%d with a ticket in flight "
                 "(collect it with ofps_hip_lk_frame_wait first)", ctx->lk_w, ctx->lk_h, W, H);
    // hip_flow expands a new frame ahead of its pair, into the plane slot of the oldest frame of THIS parameter set: other Farneback parameters
    // re-plan the workspace (other layers), which a flow still in flight would be reading -- refused like a geometry change
    const bool fb_params_changed = g.farneback && ctx->lk_fb_params_valid && (ctx->lk_fb_levels != levels || ctx->lk_fb_radius != radius);
    OFPS_REQUIRE(ctx, !(fb_params_changed && other_pending), "lk_push_frame_async: Farneback parameters changed (levels %d -> %d, radius %d -> %d) with a "
                 "ticket in flight (collect it with ofps_hip_lk_frame_wait first)", ctx->lk_fb_levels, levels, ctx->lk_fb_radius, radius);
    if (g.farneback) { ctx->lk_fb_params_valid = true; ctx->lk_fb_levels = levels; ctx->lk_fb_radius = radius; }
    if (restart) {
        rc = lk_stream_drain(ctx);
        if (rc != OFPS_HIP_OK) return rc;
        ctx->lk_w = W; ctx->lk_h = H; ctx->lk_fw = g.fw; ctx->lk_fh = g.fh; ctx->lk_fmt = g.fmt;
        ctx->fb_prev_flow.valid = false;
    }
    // the stream's frames have slots of their own: no other entry point (lk_decode, lk_flow, sad_flow, contrast_mask stage
    // their frames in S_FRAMES) can overwrite or reallocate a previous frame behind the stream's back.  Three slots: frame
    // k + 1 is uploaded (copy stream) into the slot of frame k - 2, whose last reader -- ticket k - 1 -- has been collected
    // by the time a third push is accepted.
    auto* d_frames = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_LK_FRAMES, ofps_hip_ctx::kLkSlots * px));
    // the arriving frame of a stream with a front-end is staged per ticket in flight (the ticket's upload + front-end run on one stream; the
    // buffer's previous user, ticket tno - 2, has been collected)
    auto* d_raw_all = g.frontend ? static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_FE_RAW, ofps_hip_ctx::kLkTickets * raw_bytes)) : nullptr;
    if (!d_frames || (g.frontend && !d_raw_all)) return OFPS_HIP_ENOMEM;
    if (ctx->lk_frames_gen != ctx->scratch[ofps::S_LK_FRAMES].gen) {         // (re)allocated: whatever was there is gone
        ctx->lk_frames_gen = ctx->scratch[ofps::S_LK_FRAMES].gen;
        ctx->lk_frames = 0;
    }
    hipStream_t s = ctx->stream;
    // with another ticket in flight the upload goes to the copy stream and overlaps that ticket's flow; a lone frame is
    // copied on the compute stream itself (no cross-stream event on the latency path of the synchronous call)
    const bool overlap = other_pending;
    const int slot = (int)(ctx->lk_frames % ofps_hip_ctx::kLkSlots);
    hipStream_t up = overlap ? ctx->lk_copy_stream : s;
    ctx->lk_slot_id[slot] = ++ctx->lk_frame_serial;               // (a new id for whatever is in the slot now, also if the push fails below)
    rc = lk_upload_frame(ctx, g, frame, W, H, stride, d_raw_all ? d_raw_all + (size_t)(tno % ofps_hip_ctx::kLkTickets) * raw_bytes : nullptr,
                         d_frames + (size_t)slot * px, up);
    if (rc != OFPS_HIP_OK) return rc;
    // cv-decoder's contrast mask depends on the new frame only: it is made right behind the upload, on the upload's stream -- with
    // another ticket in flight that is beside that ticket's flow instead of after this one's (one mask buffer per ticket in flight)
    const uint8_t* d_mask_ready = nullptr;
    if (g.use_mask && ctx->lk_frames + 1 >= 2) {
        auto* masks = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_LK_MASKS, (size_t)ofps_hip_ctx::kLkTickets * px));
        if (!masks) return OFPS_HIP_ENOMEM;
        uint8_t* m = masks + (size_t)(tno % ofps_hip_ctx::kLkTickets) * px;
        rc = ofps::contrast_mask_device(ctx, d_frames + (size_t)slot * px, g.fw, g.fh, g.fw, m, up);
        if (rc != OFPS_HIP_OK) return rc;
        d_mask_ready = m;
    }
    // hip_flow: the new frame's pyramid + polynomial expansion depend on that frame only, like the mask: made here, behind the upload on the
    // upload's stream -- with another ticket in flight beside that ticket's flow, whose coarse layers (a chain of dependent round trips) leave
    // most of the device idle (farneback.hip: farneback_prepare_device; the pair's flow below finds the planes by the frame's id)
    if (g.farneback && ctx->opt.fb_prepare_ahead) {
        rc = ofps::farneback_prepare_device(ctx, d_frames + (size_t)slot * px, g.fw, g.fh, g.fw, levels, 2 * radius + 1, 7, 1.5, ctx->lk_slot_id[slot], up);
        if (rc != OFPS_HIP_OK) return rc;
    }
    if (overlap) {
        OFPS_HIP_TRY(ctx, hipEventRecord(t.uploaded, up));
        OFPS_HIP_TRY(ctx, hipStreamWaitEvent(s, t.uploaded, 0));       // everything of this ticket on the compute stream comes after the upload
        if (g.farneback) ofps::farneback_mark_ordered(ctx, s);         // ... the new frame's pyramid + expansion included
    }
    // the stream position and the ticket change only once everything is enqueued: a failure below leaves both as they were (the
    // frame's slot is simply uploaded again by the next push) -- ADVICE r4
    const long frames_after = ctx->lk_frames + 1;
    int have_vectors = 0;
    if (frames_after >= 2) {                                        // cv-decoder/src/lib.rs:156-158: flow needs two frames
        rc = lk_pinned_block(ctx, &t.pinned, &t.pinned_cap, g.max_records);
        if (rc != OFPS_HIP_OK) return rc;
        void* mapped = nullptr;
        OFPS_REQUIRE(ctx, ofps::device_address_of(t.pinned, &mapped), "lk_push_frame_async: page-locked block is not device-addressable");
        const int prev_slot = (int)((frames_after - 2) % ofps_hip_ctx::kLkSlots);
        rc = lk_enqueue_frame(ctx, d_frames + (size_t)prev_slot * px, d_frames + (size_t)slot * px, g.fw, g.fh, levels, radius, iters, g,
                              reinterpret_cast<float4*>(static_cast<char*>(mapped) + 16), static_cast<uint32_t*>(mapped),
                              ctx->lk_slot_id[prev_slot], ctx->lk_slot_id[slot], d_mask_ready);
        if (rc != OFPS_HIP_OK) return rc;
        have_vectors = 1;
    }
    OFPS_HIP_TRY(ctx, hipEventRecord(t.done, s));
    ctx->lk_frames = frames_after;
    t.have_vectors = have_vectors; t.gw = g.gw; t.gh = g.gh; t.max_records = g.max_records; t.fixed_count = -1;
    t.pending = true;
    *ticket = (int)(tno & 0x7FFFFFFF);
    ctx->lk_next_ticket = tno + 1;
    return OFPS_HIP_OK;
}

int ofps_hip_lk_frame_wait(ofps_hip_ctx* ctx, int ticket, float* out_entries, size_t* n_out, int* out_w, int* out_h, int* have_vectors) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, out_entries && n_out && have_vectors, "lk_frame_wait: null pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long newest = ctx->lk_next_ticket - 1;
    long tno = -1;
    for (long k = newest; k >= 0 && k > newest - ofps_hip_ctx::kLkTickets; --k)
        if ((int)(k & 0x7FFFFFFF) == ticket) { tno = k; break; }
    OFPS_REQUIRE(ctx, tno >= 0, "lk_frame_wait: ticket %d is not in flight", ticket);
    auto& t = ctx->lk_ticket[tno % ofps_hip_ctx::kLkTickets];
    OFPS_REQUIRE(ctx, t.pending, "lk_frame_wait: ticket %d has already been collected", ticket);
    OFPS_HIP_TRY(ctx, hipEventSynchronize(t.done));
    t.pending = false;
    *n_out = 0;
    *have_vectors = t.have_vectors;
    if (out_w) *out_w = t.gw;
    if (out_h) *out_h = t.gh;
    if (t.have_vectors) lk_collect(t.pinned, t.max_records, out_entries, n_out);
    return OFPS_HIP_OK;
}

int ofps_hip_lk_push_frame(ofps_hip_ctx* ctx, const uint8_t* frame, int W, int H, int stride, int levels, int radius, int iters,
                           int max_w, int max_h, unsigned flags, float* out_entries, size_t* n_out, int* out_w, int* out_h,
                           int* have_vectors) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, frame && out_entries && n_out && have_vectors, "lk_push_frame: null host pointer");
    for (const auto& t : ctx->lk_ticket)
        OFPS_REQUIRE(ctx, !t.pending, "lk_push_frame: a read-ahead ticket is in flight (collect it with ofps_hip_lk_frame_wait first)");
    int ticket = 0;
    const int rc = ofps_hip_lk_push_frame_async(ctx, frame, W, H, stride, levels, radius, iters, max_w, max_h, flags, &ticket);
    if (rc != OFPS_HIP_OK) return rc;
    return ofps_hip_lk_frame_wait(ctx, ticket, out_entries, n_out, out_w, out_h, have_vectors);     // the caller may reuse `frame` right away
}

int ofps_hip_lk_reset(ofps_hip_ctx* ctx) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->fb_prev_flow.valid = false;                       // a new stream starts from zero flow, like a new CvDecoder
    return lk_stream_drain(ctx);
}

int ofps_hip_lk_rewind(ofps_hip_ctx* ctx) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return lk_stream_drain(ctx);                           // the frames are forgotten, the kept flow (OFPS_HIP_FLOW_USE_PREVIOUS) is not
}

int ofps_hip_lk_flow(ofps_hip_ctx* ctx, const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels,
                     int radius, int iters, float* out_flow, float* out_entries) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, prev && cur && (out_flow || out_entries), "lk_flow: null host pointer");
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && stride >= W, "lk_flow: bad geometry");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)W * H;
    auto* d_frames = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_FRAMES, 2 * px));
    auto* d_flow = static_cast<float2*>(ofps::scratch(ctx, ofps::S_WORK1, px * sizeof(float2)));
    auto* d_ent = out_entries ? static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES, px * sizeof(float4))) : nullptr;
    if (!d_frames || !d_flow || (out_entries && !d_ent)) return OFPS_HIP_ENOMEM;
    OFPS_HIP_TRY(ctx, ofps::upload_rows(d_frames, W, prev, stride, W, H, ctx->stream));
    OFPS_HIP_TRY(ctx, ofps::upload_rows(d_frames + px, W, cur, stride, W, H, ctx->stream));
    int rc = ofps::lk_flow_device(ctx, d_frames, d_frames + px, W, H, W, levels, radius, iters, d_flow, d_ent);
    if (rc != OFPS_HIP_OK) return rc;
    if (out_flow) OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_flow, d_flow, px * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    if (out_entries) OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_entries, d_ent, px * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

}  // extern "C"
