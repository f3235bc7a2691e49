// frontend.hip -- cv-decoder's frame front-end on the device (cv-decoder/src/lib.rs:98-135): the capped record grid, imgproc::resize(..,
// INTER_LINEAR) to that grid when "Process Fullres" is false (:124-133) and cvt_color(.., COLOR_BGR2GRAY) (:135).
//
// The arithmetic is OpenCV's 8-bit code path, restated integer for integer (spec + sources named in oracle/frontend_oracle.c; PARITY
// UNPINNED: OpenCV is not part of the reference tree):
//   resize:   per axis  f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s  (f64 product, f32 fraction), the horizontal edge rule,
//             11-bit coefficients rint((1 - f) * 2048), rint(f * 2048); horizontal pass in int, vertical pass
//             (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2; exact 2 x 2 reductions are the area mean (a + b + c + d + 2) >> 2;
//   gray:     (B * 1868 + G * 9617 + R * 4899 + 8192) >> 14.
// A reduced frame is tiny (150 x 84 from 1080p): one thread per destination pixel computes its own two coefficient pairs -- no tables, no
// state -- and gathers its 2 x 2 source pixels; only 2 of every ~13 source rows are touched at all.  Resize and colour conversion are ONE
// kernel (the reference's order: resize the colour frame, then convert).  The full-resolution conversion is a streaming kernel, four
// pixels per thread (three or four dword loads, one dword store).
#include "common.hpp"

namespace ofps {
namespace {

struct FeAxis { int s; int c0, c1; };

// one destination coordinate of one axis (oracle/frontend_oracle.c:linear_axis)
__device__ __forceinline__ FeAxis fe_axis(int d, int src, double scale, bool edge_rule) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (edge_rule) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    FeAxis a;
    a.s = s;
    a.c0 = (int)rintf((1.f - f) * 2048.f);
    a.c1 = (int)rintf(f * 2048.f);
    return a;
}

__device__ __forceinline__ int fe_gray(int b, int g, int r) { return (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14; }

// MODE 0: general bilinear; 1: exact 2 x 2 area mean; 2: same size (copy / conversion only)
// FMT: 0 = keep the CN channels; 1 BGR / 2 RGBA / 3 BGRA -> one gray byte
template <int CN, int FMT, int MODE>
__global__ __launch_bounds__(256) void fe_resize_kernel(const uint8_t* __restrict__ src, int W, int H, int stride, uint8_t* __restrict__ dst,
                                                        int dw, int dh, double scale_x, double scale_y) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    int v[CN];
    if (MODE == 0) {
        const FeAxis ax = fe_axis(dx, W, scale_x, true), ay = fe_axis(dy, H, scale_y, false);
        const int y0 = min(max(ay.s, 0), H - 1), y1 = min(max(ay.s + 1, 0), H - 1);
        const bool last = ax.s + 1 >= W;
        const uint8_t* r0 = src + (size_t)y0 * stride + (size_t)ax.s * CN;
        const uint8_t* r1 = src + (size_t)y1 * stride + (size_t)ax.s * CN;
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            const int d0 = last ? r0[c] * 2048 : r0[c] * ax.c0 + r0[c + CN] * ax.c1;
            const int d1 = last ? r1[c] * 2048 : r1[c] * ax.c0 + r1[c + CN] * ax.c1;
            v[c] = (((ay.c0 * (d0 >> 4)) >> 16) + ((ay.c1 * (d1 >> 4)) >> 16) + 2) >> 2;
        }
    } else if (MODE == 1) {
        const uint8_t* r0 = src + (size_t)(2 * dy) * stride + (size_t)(2 * dx) * CN;
        const uint8_t* r1 = r0 + stride;
#pragma unroll
        for (int c = 0; c < CN; ++c) v[c] = (r0[c] + r0[c + CN] + r1[c] + r1[c + CN] + 2) >> 2;
    } else {
        const uint8_t* r0 = src + (size_t)dy * stride + (size_t)dx * CN;
#pragma unroll
        for (int c = 0; c < CN; ++c) v[c] = r0[c];
    }
    if (FMT == 0) {
        uint8_t* o = dst + ((size_t)dy * dw + dx) * CN;
#pragma unroll
        for (int c = 0; c < CN; ++c) o[c] = (uint8_t)v[c];
    } else {
        const int b = FMT == 2 ? v[2 % CN] : v[0], r = FMT == 2 ? v[0] : v[2 % CN];
        dst[(size_t)dy * dw + dx] = (uint8_t)fe_gray(b, v[1 % CN], r);
    }
}

// full-resolution colour -> gray: four pixels per thread from aligned dwords (rows start on 4-byte boundaries), dense gray rows
template <int CN, int FMT>
__global__ __launch_bounds__(256) void fe_gray4_kernel(const uint8_t* __restrict__ src, int W, int H, int stride, uint8_t* __restrict__ dst) {
    const int q = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int x = q * 4;
    if (x >= W) return;
    const uint8_t* s = src + (size_t)y * stride + (size_t)x * CN;
    uint8_t* o = dst + (size_t)y * W + x;
    if (x + 4 <= W) {
        uint32_t w[CN];
        const uint32_t* p = reinterpret_cast<const uint32_t*>(s);
#pragma unroll
        for (int k = 0; k < CN; ++k) w[k] = p[k];
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int ch[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { const int byte = k * CN + c; ch[c] = (w[byte >> 2] >> (8 * (byte & 3))) & 0xFF; }
            const int g = FMT == 2 ? fe_gray(ch[2], ch[1], ch[0]) : fe_gray(ch[0], ch[1], ch[2]);
            out |= (uint32_t)g << (8 * k);
        }
        if ((W & 3) == 0) *reinterpret_cast<uint32_t*>(o) = out;
        else { o[0] = out & 0xFF; o[1] = (out >> 8) & 0xFF; o[2] = (out >> 16) & 0xFF; o[3] = out >> 24; }
    } else {
        for (int k = 0; x + k < W; ++k) {
            const uint8_t* t = s + k * CN;
            o[k] = (uint8_t)(FMT == 2 ? fe_gray(t[2], t[1], t[0]) : fe_gray(t[0], t[1], t[2]));
        }
    }
}

template <int CN, int FMT>
void fe_launch_resize(const uint8_t* src, int W, int H, int stride, uint8_t* dst, int dw, int dh, hipStream_t s) {
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4), block(256);
    const double scale_x = 1. / ((double)dw / W), scale_y = 1. / ((double)dh / H);      // cv::resize: inv_scale = dsize / ssize; scale = 1. / inv_scale
    if (dw == W && dh == H) hipLaunchKernelGGL((fe_resize_kernel<CN, FMT, 2>), grid, block, 0, s, src, W, H, stride, dst, dw, dh, scale_x, scale_y);
    else if (W == 2 * dw && H == 2 * dh) hipLaunchKernelGGL((fe_resize_kernel<CN, FMT, 1>), grid, block, 0, s, src, W, H, stride, dst, dw, dh, scale_x, scale_y);
    else hipLaunchKernelGGL((fe_resize_kernel<CN, FMT, 0>), grid, block, 0, s, src, W, H, stride, dst, dw, dh, scale_x, scale_y);
}

}  // namespace

int frame_format_channels(int fmt) { return fmt == OFPS_HIP_FMT_LUMA ? 1 : fmt == OFPS_HIP_FMT_BGR ? 3 : (fmt == OFPS_HIP_FMT_RGBA || fmt == OFPS_HIP_FMT_BGRA) ? 4 : 0; }

void cv_grid(int W, int H, int max_w, int max_h, int* gw, int* gh) {
    // cv-decoder/src/lib.rs:98-121 with aspect_ratio_scale = (1, 1): usize arithmetic
    const size_t cw = (size_t)(max_w < W ? max_w : W), ch = (size_t)(max_h < H ? max_h : H);
    const size_t wb0 = cw, wb1 = cw * (size_t)H / (size_t)W, hb0 = ch * (size_t)W / (size_t)H, hb1 = ch;
    *gw = (int)(wb0 < hb0 ? wb0 : hb0); *gh = (int)(wb0 < hb0 ? wb1 : hb1);
}

// [resize to dw x dh] -> [gray]: what cv-decoder leaves in self.gray for one frame.  d_src: H rows of W pixels of `fmt`, `stride` BYTES apart;
// d_dst: dh x dw dense bytes (to_gray) or dh x dw x cn (keep the channels).  st: the stream (nullptr = ctx->stream).
int frontend_device(ofps_hip_ctx* ctx, const uint8_t* d_src, int W, int H, int stride, int fmt, bool to_gray, uint8_t* d_dst, int dw, int dh,
                    hipStream_t st) {
    hipStream_t s = st ? st : ctx->stream;
    const int cn = frame_format_channels(fmt);
    OFPS_REQUIRE(ctx, cn != 0, "frontend: unknown frame format %d", fmt);
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && dw >= 1 && dh >= 1 && stride >= W * cn, "frontend: bad geometry %dx%d (stride %d, %d channels) -> %dx%d", W, H, stride, cn, dw, dh);
    OFPS_REQUIRE(ctx, (size_t)H * (size_t)stride < (1ull << 31), "frontend: frame too large");
    if (to_gray && cn > 1 && dw == W && dh == H && (stride & 3) == 0 && (reinterpret_cast<uintptr_t>(d_src) & 3) == 0) {
        const dim3 grid(((W + 3) / 4 + 255) / 256, H);
        if (fmt == OFPS_HIP_FMT_BGR) hipLaunchKernelGGL((fe_gray4_kernel<3, 1>), grid, dim3(256), 0, s, d_src, W, H, stride, d_dst);
        else if (fmt == OFPS_HIP_FMT_RGBA) hipLaunchKernelGGL((fe_gray4_kernel<4, 2>), grid, dim3(256), 0, s, d_src, W, H, stride, d_dst);
        else hipLaunchKernelGGL((fe_gray4_kernel<4, 3>), grid, dim3(256), 0, s, d_src, W, H, stride, d_dst);
    } else if (cn == 1) {
        fe_launch_resize<1, 0>(d_src, W, H, stride, d_dst, dw, dh, s);
    } else if (!to_gray) {
        if (cn == 3) fe_launch_resize<3, 0>(d_src, W, H, stride, d_dst, dw, dh, s);
        else fe_launch_resize<4, 0>(d_src, W, H, stride, d_dst, dw, dh, s);
    } else if (fmt == OFPS_HIP_FMT_BGR) {
        fe_launch_resize<3, 1>(d_src, W, H, stride, d_dst, dw, dh, s);
    } else if (fmt == OFPS_HIP_FMT_RGBA) {
        fe_launch_resize<4, 2>(d_src, W, H, stride, d_dst, dw, dh, s);
    } else {
        fe_launch_resize<4, 3>(d_src, W, H, stride, d_dst, dw, dh, s);
    }
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

}  // namespace ofps

extern "C" {

int ofps_hip_cv_grid(int W, int H, int max_w, int max_h, int* gw, int* gh) {
    if (W < 1 || H < 1 || max_w < 1 || max_h < 1 || !gw || !gh) return OFPS_HIP_EINVAL;
    ofps::cv_grid(W, H, max_w, max_h, gw, gh);
    return OFPS_HIP_OK;
}

int ofps_hip_frame_channels(int fmt) { return ofps::frame_format_channels(fmt); }

int ofps_hip_resize_linear_dev(ofps_hip_ctx* ctx, const void* d_src, int W, int H, int stride, int fmt, void* d_dst, int dw, int dh) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_src && d_dst, "resize_linear_dev: null device pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ofps::frontend_device(ctx, static_cast<const uint8_t*>(d_src), W, H, stride, fmt, false, static_cast<uint8_t*>(d_dst), dw, dh, nullptr);
}

int ofps_hip_cv_frontend_dev(ofps_hip_ctx* ctx, const void* d_frame, int W, int H, int stride, int fmt, int reduced, int max_w, int max_h,
                             void* d_out_gray, int* out_w, int* out_h) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_frame && d_out_gray, "cv_frontend_dev: null device pointer");
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && max_w >= 1 && max_h >= 1, "cv_frontend_dev: bad geometry");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int dw = W, dh = H;
    if (reduced) ofps::cv_grid(W, H, max_w, max_h, &dw, &dh);
    OFPS_REQUIRE(ctx, dw >= 1 && dh >= 1, "cv_frontend_dev: the capped grid of %dx%d under (%d, %d) is empty", W, H, max_w, max_h);
    if (out_w) *out_w = dw;
    if (out_h) *out_h = dh;
    return ofps::frontend_device(ctx, static_cast<const uint8_t*>(d_frame), W, H, stride, fmt, true, static_cast<uint8_t*>(d_out_gray), dw, dh, nullptr);
}

// host-pointer forms: stage, run, read back (tests and one-off callers; the decoders run the front-end behind their uploads)
static int fe_host_call(ofps_hip_ctx* ctx, const uint8_t* src, int W, int H, int stride, int fmt, bool to_gray, uint8_t* dst, int dw, int dh) {
    const int cn = ofps::frame_format_channels(fmt);
    OFPS_REQUIRE(ctx, cn != 0, "frontend: unknown frame format %d", fmt);
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && dw >= 1 && dh >= 1 && stride >= W * cn, "frontend: bad geometry");
    const size_t row = (size_t)W * cn, row_al = (row + 3) & ~(size_t)3, out_bytes = (size_t)dw * dh * (to_gray ? 1 : cn);
    auto* d_src = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_FE_RAW_PAIR, row_al * H));
    auto* d_dst = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_FRAMES, out_bytes));
    if (!d_src || !d_dst) return OFPS_HIP_ENOMEM;
    OFPS_HIP_TRY(ctx, ofps::upload_rows(d_src, row_al, src, stride, row, H, ctx->stream));
    const int rc = ofps::frontend_device(ctx, d_src, W, H, (int)row_al, fmt, to_gray, d_dst, dw, dh, nullptr);
    if (rc != OFPS_HIP_OK) return rc;
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(dst, d_dst, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

int ofps_hip_resize_linear(ofps_hip_ctx* ctx, const uint8_t* src, int W, int H, int stride, int fmt, uint8_t* dst, int dw, int dh) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, src && dst, "resize_linear: null host pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return fe_host_call(ctx, src, W, H, stride, fmt, false, dst, dw, dh);
}

int ofps_hip_cv_frontend(ofps_hip_ctx* ctx, const uint8_t* frame, int W, int H, int stride, int fmt, int reduced, int max_w, int max_h,
                         uint8_t* out_gray, int* out_w, int* out_h) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, frame && out_gray, "cv_frontend: null host pointer");
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && max_w >= 1 && max_h >= 1, "cv_frontend: bad geometry");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int dw = W, dh = H;
    if (reduced) ofps::cv_grid(W, H, max_w, max_h, &dw, &dh);
    OFPS_REQUIRE(ctx, dw >= 1 && dh >= 1, "cv_frontend: the capped grid of %dx%d under (%d, %d) is empty", W, H, max_w, max_h);
    if (out_w) *out_w = dw;
    if (out_h) *out_h = dh;
    return fe_host_call(ctx, frame, W, H, stride, fmt, true, out_gray, dw, dh);
}

}  // extern "C"
