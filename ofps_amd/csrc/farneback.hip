// farneback.hip -- N2 in the reference's own algorithm family: Farneback's polynomial-expansion dense flow as cv-decoder calls it
//   cv-decoder/src/lib.rs:188-199:  calc_optical_flow_farneback(old_gray, gray, flow, 0.5, 5, 13, 3, 7, 1.5, flags)
// (the "hip_flow" decoder's compute; hip_lk is the build-defined iterative Lucas-Kanade).  The arithmetic the reference runs lives in
// OpenCV (not under /root/reference, not installed: PARITY UNPINNED); what is implemented is the published algorithm in the form
// OpenCV's calcOpticalFlowFarneback gives it, stage by stage with the same precision per stage (f32 blur / resize / vertical half of
// the expansion / matrices, f64 horizontal half / window sums / 2x2 solve) -- DESIGN.md "N2b", oracle/farneback_oracle.c.
//
// Kernels (layers k = K .. 0, scale 0.5^k; OpenCV makes every layer from the ORIGINAL frame: Gaussian blur, then bilinear resize):
//   fb_pyr_h_kernel      ONE launch for all layers >= 1: a workgroup takes eight frame rows (bytes) into LDS and evaluates one layer's row
//                        filter at the columns that layer's resize will sample                               -> T_k [2][H][2 w_k]
//   fb_pyr_v_kernel      ONE launch for all layers >= 1: column filter at the sampled rows + the bilinear combine -> I_k [2][h_k][w_k]
//   fb_polyexp_kernel    ONE launch for all layers and both frames: 15-tap separable polynomial expansion, tile + halo in LDS; layer 0's
//                        image is never stored (its [1 2 1]/4 blur of the u8 frame is evaluated while the tile is filled) -> R [5][h][w] planar
//   fb_start_kernel      per layer: FarnebackUpdateMatrices from the flow the layer starts with (zero / the coarser layer's flow resized
//                        on the fly x 2 / the caller's initial flow)                                         -> M [5][h][w]
//   fb_iter_kernel x iters   per update: a workgroup owns a 32 x 16 tile: a thread per (channel, column) of tile + window halo takes its
//                        column of M from global memory and makes the 16 row-window sums in registers (f64, ascending: the oracle's
//                        order) -> LDS; then per pixel the column-window sums, the 2 x 2 solve, and -- except in a layer's last update --
//                        the NEXT update's matrices for its own pixels, written where the flow they depend on is produced; the flow
//                        itself is stored only when somebody reads it (next layer, caller).  Tiles are dealt to the XCDs in contiguous runs.
// Stream forms keep the second frame's pyramid + expansion for the next pair (ofps_hip_ctx::fb_cache).
// All streaming: the honest roofline of this path is HBM (R0 + R1 + M in / out per update); measured numbers: DESIGN.md "N2b".
#include "common.hpp"

#include <cmath>
#include <vector>

namespace {

constexpr int kMaxBlurTaps = 160;       // 2 r + 1 <= 159 for the six layers above the frame (k = 6: sigma 31.5); deeper pyramids are refused
constexpr int kMaxPolyN = 15;

struct FbBlur {                         // Gaussian taps of one layer (getGaussianKernel(ksize, sigma, CV_32F))
    int r;
    float taps[kMaxBlurTaps];
};
struct FbPoly {                         // FarnebackPrepareGaussian
    int n;
    float g[kMaxPolyN + 1], xg[kMaxPolyN + 1], xxg[kMaxPolyN + 1];
    double ig11, ig03, ig33, ig55;
};

int round_half_even(double v) { return (int)std::nearbyint(v); }       // cvRound

bool make_blur(int k, FbBlur* b) {
    double scale = 1.0;
    for (int i = 0; i < k; ++i) scale *= 0.5;
    const double sigma = (1.0 / scale - 1.0) * 0.5;
    int ksize = round_half_even(sigma * 5) | 1;
    if (ksize < 3) ksize = 3;
    if (ksize > kMaxBlurTaps) return false;
    b->r = ksize / 2;
    if (sigma <= 0) { b->taps[0] = 0.25f; b->taps[1] = 0.5f; b->taps[2] = 0.25f; return true; }
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < ksize; ++i) {
        const double x = i - (ksize - 1) * 0.5;
        b->taps[i] = (float)std::exp(s2 * x * x);
        sum += b->taps[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < ksize; ++i) b->taps[i] = (float)(b->taps[i] * sum);
    return true;
}

void make_poly(int n, double sigma, FbPoly* p) {
    if (sigma < 1.1920929e-07) sigma = n * 0.3;
    std::vector<float> full((size_t)(2 * n + 1));
    double s = 0;
    for (int x = -n; x <= n; ++x) { full[(size_t)(x + n)] = (float)std::exp(-x * x / (2 * sigma * sigma)); s += full[(size_t)(x + n)]; }
    s = 1.0 / s;
    double b = 0, c = 0;
    for (int x = -n; x <= n; ++x) {
        const float gv = (float)(full[(size_t)(x + n)] * s);
        if (x >= 0) { p->g[x] = gv; p->xg[x] = (float)(x * gv); p->xxg[x] = (float)(x * x * gv); }
        b += (double)gv * x * x; c += (double)gv * x * x * x * x;
    }
    double a = 0;
    for (int x = -n; x <= n; ++x) a += (double)p->g[x < 0 ? -x : x];
    // moment matrix in the basis (1, x, y, x^2, y^2, xy): G00 = a^2, G11 = G03 = a b, G33 = a c, G34 = G55 = b^2; closed-form inverse
    const double A = a * a, B = a * b, C = a * c, D = b * b;
    const double det = (C - D) * (A * (C + D) - 2 * B * B);
    p->n = n;
    p->ig11 = 1.0 / B; p->ig03 = -B * (C - D) / det; p->ig33 = (A * C - B * B) / det; p->ig55 = 1.0 / D;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// resize INTER_LINEAR, one axis: source index and fraction of destination index d (half-pixel centres, clamped); inv = 1 / (dn / sn) in f64
__device__ __forceinline__ void resize_axis(int d, int sn, double inv, int* s0, int* s1, float* f) {
    float fr = (float)((d + 0.5) * inv - 0.5);
    int s = (int)floorf(fr);
    fr -= (float)s;
    if (s < 0) { fr = 0; s = 0; }
    if (s >= sn - 1) { fr = 0; s = sn - 1; }
    *s0 = s; *s1 = s + 1 < sn ? s + 1 : sn - 1; *f = fr;
}

// ---- the pyramid above layer 0, both passes for ALL layers in one launch each -----------------------------------------------------
// Every layer is OpenCV's "blur the original, then resize": the resize samples two columns and two rows per output, so the row filter
// is evaluated only at those columns (T_k [2][H][2 w_k]) and the column filter only at those rows.
constexpr int kMaxLayers = 7;           // k = 0 .. 6
constexpr int kMaxTaps = 320;           // 3 + 9 + 19 + 39 + 79 + 159 for k = 1 .. 6
struct FbPyr {
    int K;                              // layers 1 .. K are made here (layer 0 inside the expansion kernel)
    int w[kMaxLayers], h[kMaxLayers], r[kMaxLayers], toff[kMaxLayers];
    double inv_x[kMaxLayers], inv_y[kMaxLayers];
    size_t t_off[kMaxLayers];           // T_k inside T (floats, per image block of the layer: [2][H][2 w_k])
    size_t i_off[kMaxLayers];           // I_k inside I (floats: [2][h_k][w_k])
    int blk0[kMaxLayers + 1];           // fb_pyr_v_kernel / multi-layer launches: first block of layer k
    float taps[kMaxTaps];
};

// One workgroup per (group of 8 frame rows, image, layer): the rows' bytes go to LDS once (dword copies, reflect-101 margins), the layer's
// row filter is evaluated at the columns its resize samples.  Lanes are dealt so that a wave's LDS reads fall into different banks
// without any index arithmetic: the sampled columns of layer k are 2^k bytes apart, i.e. 2^(k-2) dwords; a wave takes 64 / nr columns of
// nr = 1, 1, 2, 4, 8 (k = 1 .. 5) different rows whose starts are 1 dword (mod 64) apart, and every lane walks the 8 / nr rows it owns
// with one tap in a register -- round 5: the first form (one row per workgroup, f32 row with a pad word per 32, border branch per
// column, the resize's f64 index arithmetic per column AND row) spent 2,340 VALU instructions per wave on 75 tap pairs.
constexpr int kPR = 8;
constexpr int kPyrRS0 = 4 * (32 * 18 + 2), kPyrRS1 = 4 * (32 * 34 + 2), kPyrRS2 = 4 * (32 * 132 + 2);     // 2,312 / 4,360 / 16,904 bytes: W + margins up to 16,384 + 2 x 160
template <int NRL, int RS>                       // NRL rows per lane; nr = kPR / NRL rows per wave instruction; RS bytes from row to row (immediate offsets)
__device__ __forceinline__ void fb_pyr_h_cols(const uint8_t* __restrict__ rp, const float* __restrict__ taps, int r, float (&s)[NRL]) {
    constexpr int nr = kPR / NRL;
    constexpr int JB = NRL >= 8 ? 1 : (NRL == 4 ? 2 : (NRL == 2 ? 4 : 8));      // taps per batch: 16 byte reads requested before the first is used
    const float t0 = taps[r];                                                    // (a lane's sum is a chain over j: without batches every tap pair
#pragma unroll                                                                   //  waited out its own LDS round trip -- 39 of them in layer 5)
    for (int i = 0; i < NRL; ++i) s[i] = t0 * (float)rp[i * nr * RS];
    int j = 1;
    for (; j + JB - 1 <= r; j += JB) {
        float t[JB];
        uint8_t a[JB][NRL], b[JB][NRL];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            t[jj] = taps[r + j + jj];
#pragma unroll
            for (int i = 0; i < NRL; ++i) { a[jj][i] = rp[i * nr * RS - (j + jj)]; b[jj][i] = rp[i * nr * RS + (j + jj)]; }
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
            for (int i = 0; i < NRL; ++i) s[i] += t[jj] * (float)((int)a[jj][i] + (int)b[jj][i]);      // (= (float)a + (float)b exactly: one conversion)
    }
    for (; j <= r; ++j) {
        const float t = taps[r + j];
#pragma unroll
        for (int i = 0; i < NRL; ++i) s[i] += t * (float)((int)rp[i * nr * RS - j] + (int)rp[i * nr * RS + j]);
    }
}

template <int NRL, int RS>
__device__ __forceinline__ void fb_pyr_h_layer(const uint8_t* __restrict__ rows, int PAD, const float* __restrict__ taps, int r, int W, int H,
                                               double inv_x, int ncol, int y0, float* __restrict__ Tz) {
    constexpr int nr = kPR / NRL, CW = 64 / nr;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q0 = lane / CW, cl = lane - q0 * CW;
    for (int c0 = wave * CW; c0 < ncol; c0 += 4 * CW) {
        const int c = c0 + cl, cc = c < ncol ? c : ncol - 1;
        int s0, s1; float f;
        resize_axis(cc >> 1, W, inv_x, &s0, &s1, &f);
        const int xs = (cc & 1) ? s1 : s0;
        float s[NRL];
        fb_pyr_h_cols<NRL, RS>(rows + q0 * RS + PAD + xs, taps, r, s);
        if (c < ncol) {
#pragma unroll
            for (int i = 0; i < NRL; ++i) {
                const int y = y0 + i * nr + q0;
                if (y < H) Tz[(size_t)y * ncol + c] = s[i];
            }
        }
    }
}

template <int RS>
__global__ __launch_bounds__(256) void fb_pyr_h_kernel(const uint8_t* __restrict__ img0, const uint8_t* __restrict__ img1, int W, int H, int stride,
                                                       const FbPyr P, int PAD, float* __restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t srows[];      // [kPR][RS] bytes, then the layer's taps
    const int y0 = blockIdx.x * kPR, z = blockIdx.y, k = P.K - (int)blockIdx.z, r = P.r[k];     // (the long filters' workgroups are dispatched first)
    float* staps = reinterpret_cast<float*>(srows + kPR * RS);
    const uint8_t* img = z ? img1 : img0;
    for (int i = threadIdx.x; i < 2 * r + 1; i += 256) staps[i] = P.taps[P.toff[k] + i];
    const bool dwords = (stride & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0;
    const int W4 = dwords ? W / 4 : 0;                                   // columns [0, 4 W4) as dwords, the rest and the margins byte by byte
    {   // 32 threads per row
        const int q = threadIdx.x >> 5, l = threadIdx.x & 31;
        const int y = y0 + q < H ? y0 + q : H - 1;
        const uint8_t* src = img + (size_t)y * stride;
        uint8_t* dst = srows + q * RS + PAD;
        for (int c4 = l; c4 < W4; c4 += 32) *reinterpret_cast<uint32_t*>(dst + 4 * c4) = *reinterpret_cast<const uint32_t*>(src + 4 * c4);
        const int rest = W - 4 * W4 + 2 * r;                             // columns [4 W4, W) and the margins [-r, 0), [W, W + r)
        for (int e = l; e < rest; e += 32) {
            const int x = e < r ? e - r : (e < 2 * r ? W + (e - r) : 4 * W4 + (e - 2 * r));
            dst[x] = src[reflect101(x, W)];
        }
    }
    __syncthreads();
    const int ncol = 2 * P.w[k];
    float* Tz = T + P.t_off[k] + (size_t)z * H * ncol;
    if (k <= 2) fb_pyr_h_layer<8, RS>(srows, PAD, staps, r, W, H, P.inv_x[k], ncol, y0, Tz);
    else if (k == 3) fb_pyr_h_layer<4, RS>(srows, PAD, staps, r, W, H, P.inv_x[k], ncol, y0, Tz);
    else if (k == 4) fb_pyr_h_layer<2, RS>(srows, PAD, staps, r, W, H, P.inv_x[k], ncol, y0, Tz);
    else fb_pyr_h_layer<1, RS>(srows, PAD, staps, r, W, H, P.inv_x[k], ncol, y0, Tz);
}

// column filter at the sampled rows + HResizeLinear / VResizeLinear; 1-D grid over (layer, tile, image).  The launch is
// bound by round trips, not by arithmetic (138 VALU instructions per wave): short filters (r <= 4: layers 1, 2 -- 97 % of the outputs) take
// one thread per output with all 4 (2 r + 1) loads issued before the first sum; long filters one thread per column sum (the serial
// part: up to 159 taps), eight tap pairs' loads in flight at a time, the quad's first lane combining the four sums.
constexpr int kShortR = 4;
template <int R_>
__device__ __forceinline__ void fb_pyr_v_short(const float* __restrict__ Tz, int ncol, int W, int H, double inv_x, double inv_y,
                                               const float* __restrict__ taps, int x, int y, float* __restrict__ out) {
    int y0, y1, xs0, xs1; float a1, b1;
    resize_axis(x, W, inv_x, &xs0, &xs1, &a1);
    resize_axis(y, H, inv_y, &y0, &y1, &b1);
    float v[2][2][2 * R_ + 1];                                   // [row y0 / y1][column 2 x / 2 x + 1][tap]
#pragma unroll
    for (int t = 0; t < 2 * R_ + 1; ++t)
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
            const int yy = reflect101((ry ? y1 : y0) + t - R_, H);
            const float2 p = *reinterpret_cast<const float2*>(Tz + (size_t)yy * ncol + 2 * x);
            v[ry][0][t] = p.x; v[ry][1][t] = p.y;
        }
    float cs[2][2];
#pragma unroll
    for (int ry = 0; ry < 2; ++ry)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
            float s = taps[R_] * v[ry][cx][R_];
#pragma unroll
            for (int j = 1; j <= R_; ++j) s += taps[R_ + j] * (v[ry][cx][R_ - j] + v[ry][cx][R_ + j]);
            cs[ry][cx] = s;
        }
    const float a0 = 1.0f - a1, b0 = 1.0f - b1;
    const float h0 = cs[0][0] * a0 + cs[0][1] * a1;
    const float h1 = cs[1][0] * a0 + cs[1][1] * a1;
    *out = h0 * b0 + h1 * b1;
}

__global__ __launch_bounds__(256) void fb_pyr_v_kernel(const float* __restrict__ T, int W, int H, const FbPyr P, int n_img, float* __restrict__ I) {
    // (dispatch order reversed: the long filters' few workgroups -- each a serial chain of up to 159 taps -- start first and run beside
    // the bulk of the short ones instead of after them)
    // (1-D grid, the image in the lowest bit: BOTH images' long chains are at the front of the dispatch order)
    const int bz = (int)gridDim.x - 1 - (int)blockIdx.x, bx = bz / n_img, z = bz - bx * n_img;
    int k = 1;
    while (k < P.K && bx >= P.blk0[k + 1]) ++k;
    const int w = P.w[k], h = P.h[k], r = P.r[k];
    const int ncol = 2 * w, b = bx - P.blk0[k];
    const float* Tz = T + P.t_off[k] + (size_t)z * H * ncol;
    const float* taps = P.taps + P.toff[k];                      // wave-uniform indices: scalar loads
    if (r <= kShortR) {
        const int tiles_x = (w + 63) / 64;
        const int x = (b % tiles_x) * 64 + (threadIdx.x & 63), y = (b / tiles_x) * 4 + (threadIdx.x >> 6);
        if (x >= w || y >= h) return;
        float* out = I + P.i_off[k] + ((size_t)z * h + y) * w + x;
        switch (r) {                                              // compile-time tap counts: the tap arrays stay in registers
            case 1: fb_pyr_v_short<1>(Tz, ncol, W, H, P.inv_x[k], P.inv_y[k], taps, x, y, out); break;
            case 2: fb_pyr_v_short<2>(Tz, ncol, W, H, P.inv_x[k], P.inv_y[k], taps, x, y, out); break;
            case 3: fb_pyr_v_short<3>(Tz, ncol, W, H, P.inv_x[k], P.inv_y[k], taps, x, y, out); break;
            default: fb_pyr_v_short<4>(Tz, ncol, W, H, P.inv_x[k], P.inv_y[k], taps, x, y, out); break;
        }
        return;
    }
    const int tiles_x = (w + 15) / 16;
    const int o = threadIdx.x >> 2, corner = threadIdx.x & 3;
    int x = (b % tiles_x) * 16 + (o & 15), y = (b / tiles_x) * 4 + (o >> 4);
    const bool live = x < w && y < h;
    x = x < w ? x : w - 1; y = y < h ? y : h - 1;                 // (every lane takes part in the shuffles)
    int y0, y1, xs0, xs1; float a1, b1;
    resize_axis(x, W, P.inv_x[k], &xs0, &xs1, &a1);               // (the two source columns are T's columns 2 x and 2 x + 1)
    resize_axis(y, H, P.inv_y[k], &y0, &y1, &b1);
    const int ys = (corner & 2) ? y1 : y0, c = 2 * x + (corner & 1);
    const float* p = Tz + (size_t)ys * ncol + c;
    float s = taps[r] * p[0];
    if (ys - r >= 0 && ys + r < H) {
        // up to eight tap pairs' loads in flight (r is uniform: the guards are scalar branches): 39 pairs = five round trips, not four + seven single ones; more in flight costs the bulk its occupancy
        constexpr int kB = 8;
        for (int j = 1; j <= r; j += kB) {
            float lo[kB], hi[kB];
#pragma unroll
            for (int q = 0; q < kB; ++q)
                if (j + q <= r) { lo[q] = p[-(ptrdiff_t)(j + q) * ncol]; hi[q] = p[(ptrdiff_t)(j + q) * ncol]; }
#pragma unroll
            for (int q = 0; q < kB; ++q)
                if (j + q <= r) s += taps[r + j + q] * (lo[q] + hi[q]);
        }
    } else {
        for (int j = 1; j <= r; ++j) s += taps[r + j] * (Tz[(size_t)reflect101(ys - j, H) * ncol + c] + Tz[(size_t)reflect101(ys + j, H) * ncol + c]);
    }
    const float c01 = __shfl_down(s, 1), c10 = __shfl_down(s, 2), c11 = __shfl_down(s, 3);
    if (corner == 0 && live) {
        const float a0 = 1.0f - a1, b0 = 1.0f - b1;
        const float h0 = s * a0 + c01 * a1;
        const float h1 = c10 * a0 + c11 * a1;
        I[P.i_off[k] + ((size_t)z * h + y) * w + x] = h0 * b0 + h1 * b1;
    }
}

// ---- FarnebackPolyExp: I -> R [5][h][w] planar; tile 64 x 16, halo n, replicate border.  One launch covers a list of (layer, image)
// jobs.  Layer 0's image is never stored: its [1 2 1]/4 x [1 2 1]/4 blur of the u8 frame is evaluated while the tile is filled.
constexpr int kPX = 64, kPY = 16;
struct FbExpJob { const float* I; const uint8_t* u8; int stride; int w, h; float* R; int blk0, tiles_x; };
struct FbExp { int njobs; FbExpJob job[2 * kMaxLayers]; };

template <int N_>                               // N_ > 0: poly_n known at compile time (loops unrolled, taps in registers); 0: any
__global__ __launch_bounds__(256) void fb_polyexp_kernel(const FbExp E, const FbPoly P, float t_c, float t_s) {
    extern __shared__ float sh[];
    int jb = 0;
    while (jb + 1 < E.njobs && (int)blockIdx.x >= E.job[jb + 1].blk0) ++jb;
    const FbExpJob J = E.job[jb];
    const int w = J.w, h = J.h;
    const int b = (int)blockIdx.x - J.blk0;
    const int n = N_ > 0 ? N_ : P.n, HW = kPX + 2 * n, HH = kPY + 2 * n;
    float* sI = sh;                               // [HH][HW]
    float* sV = sh + HH * HW;                     // [3][kPY][HW]
    const int x0 = (b % J.tiles_x) * kPX, y0 = (b / J.tiles_x) * kPY;
    if (J.u8) {
        // layer 0: [t_s t_c t_s] row filter then column filter of the u8 frame (reflect-101), evaluated at the CLAMPED pixel of every
        // tile + halo element (the expansion's border is replicate).  The frame's bytes go to LDS once, as dwords (round 5: nine byte
        // loads per element from global memory made this launch bound by its memory instructions), then rows, then columns.
        const int PADX = (n + 4) & ~3, UW = kPX + 2 * PADX, UH = HH + 2;          // U: frame bytes of [UX0, UX0 + UW) x [UY0, UY0 + UH), reflected
        const int UX0 = x0 - PADX, UY0 = y0 - n - 1;
        uint8_t* U = reinterpret_cast<uint8_t*>(sI);                                 // (sI itself is written after U's last read)
        float* Bh = sV;                                                              // [UH][HW] row-filtered values (sV is not in use yet)
        const bool whole = UX0 >= 0 && UX0 + UW <= w && UY0 >= 0 && UY0 + UH <= h && (J.stride & 3) == 0 &&
                           (reinterpret_cast<uintptr_t>(J.u8) & 3) == 0;
        if (whole) {
            for (int i = threadIdx.x; i < UH * (UW / 4); i += 256) {
                const int r = i / (UW / 4), c4 = i - r * (UW / 4);
                reinterpret_cast<uint32_t*>(U)[i] = *reinterpret_cast<const uint32_t*>(J.u8 + (size_t)(UY0 + r) * J.stride + UX0 + 4 * c4);
            }
        } else {
            for (int i = threadIdx.x; i < UH * UW; i += 256) {
                const int r = i / UW, c = i - r * UW;
                U[i] = J.u8[(size_t)reflect101(UY0 + r, h) * J.stride + reflect101(UX0 + c, w)];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < UH * HW; i += 256) {
            const int r = i / HW, hx = i - r * HW;
            int gx = x0 - n + hx;
            gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
            const uint8_t* u = U + r * UW + (gx - UX0);
            Bh[i] = t_c * (float)u[0] + t_s * ((float)u[-1] + (float)u[1]);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < HH * HW; i += 256) {
            const int hy = i / HW, hx = i - hy * HW;
            int gy = y0 - n + hy;
            gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
            const float* b = Bh + (gy - UY0) * HW + hx;
            sI[i] = t_c * b[0] + t_s * (b[-HW] + b[HW]);
        }
    } else {
        for (int i = threadIdx.x; i < HH * HW; i += 256) {
            const int hy = i / HW, hx = i - hy * HW;
            int gx = x0 - n + hx, gy = y0 - n + hy;
            gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
            gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
            sI[i] = J.I[(size_t)gy * w + gx];
        }
    }
    __syncthreads();
    // vertical half (f32).  The oracle clamps ROW INDICES (y - k, y + k) to the image, which the clamped fill above reproduces.
    for (int i = threadIdx.x; i < kPY * HW; i += 256) {
        const int ty = i / HW, hx = i - ty * HW;
        const float* c = sI + (ty + n) * HW + hx;
        float t0 = c[0] * P.g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 1; k <= n; ++k) {
            const float a = c[-k * HW], bb = c[k * HW];
            const float p = a + bb;
            t0 = t0 + P.g[k] * p;
            t1 = t1 + P.xg[k] * (bb - a);
            t2 = t2 + P.xxg[k] * p;
        }
        sV[i] = t0; sV[kPY * HW + i] = t1; sV[2 * kPY * HW + i] = t2;
    }
    __syncthreads();
    const size_t plane = (size_t)w * h;
    for (int i = threadIdx.x; i < kPY * kPX; i += 256) {
        const int ty = i / kPX, tx = i - ty * kPX;
        const int gx = x0 + tx, gy = y0 + ty;
        if (gx >= w || gy >= h) continue;
        const float* r0 = sV + ty * HW + tx + n;
        const float* r1 = r0 + kPY * HW;
        const float* r2 = r1 + kPY * HW;
        const float g0 = P.g[0];
        double b1 = r0[0] * g0, b2 = 0, b3 = r1[0] * g0, b4 = 0, b5 = r2[0] * g0, b6 = 0;
#pragma unroll
        for (int k = 1; k <= n; ++k) {
            const double tg = r0[k] + r0[-k];
            const float gk = P.g[k];
            // (tg and the taps are f32 values: their f64 product is exact, so the fused form rounds exactly where the oracle's separate
            // multiply and add round -- one instruction instead of two on the quarter-rate f64 pipe)
            b1 = __builtin_fma(tg, (double)gk, b1);
            b4 = __builtin_fma(tg, (double)P.xxg[k], b4);
            b2 += (r0[k] - r0[-k]) * P.xg[k];
            b3 += (r1[k] + r1[-k]) * gk;
            b6 += (r1[k] - r1[-k]) * P.xg[k];
            b5 += (r2[k] + r2[-k]) * gk;
        }
        const size_t o = (size_t)gy * w + gx;
        J.R[o] = (float)(b3 * P.ig11);
        J.R[plane + o] = (float)(b2 * P.ig11);
        J.R[2 * plane + o] = (float)(b1 * P.ig03 + b5 * P.ig33);
        J.R[3 * plane + o] = (float)(b1 * P.ig03 + b4 * P.ig33);
        J.R[4 * plane + o] = (float)(b6 * P.ig55);
    }
}

// ---- FarnebackUpdateMatrices for one pixel: R0, R1 planar [5][h][w]
typedef float fb_f2u __attribute__((ext_vector_type(2), aligned(4)));       // two neighbouring floats at any 4-byte alignment: one global_load_dwordx2
struct FbLayer { const float* R0; const float* R1; int w, h; };
// q: the pixel's own coefficients R0[0..4][y][x] (loaded by the caller: they do not depend on the flow, so they can be on their way early)
__device__ __forceinline__ void fb_matrices(const FbLayer& L, int x, int y, float dx, float dy, const float (&q)[5], float (&m)[5]) {
    const int w = L.w, h = L.h;
    const size_t plane = (size_t)w * h;
    const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    float r2, r3, r4, r5, r6;
    fx -= x1; fy -= y1;
    if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
        // the two corners of a row are neighbours in memory: one 8-byte load (any 4-byte alignment) instead of two 4-byte loads --
        // the update kernels are bound by the number of memory instructions their gathers issue, not by the bytes
        const float* p = L.R1 + (size_t)y1 * w + x1;
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        fb_f2u t[5], b[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            t[c] = *reinterpret_cast<const fb_f2u*>(p + c * plane);
            b[c] = *reinterpret_cast<const fb_f2u*>(p + c * plane + w);
        }
        r2 = a00 * t[0].x + a01 * t[0].y + a10 * b[0].x + a11 * b[0].y;
        r3 = a00 * t[1].x + a01 * t[1].y + a10 * b[1].x + a11 * b[1].y;
        r4 = a00 * t[2].x + a01 * t[2].y + a10 * b[2].x + a11 * b[2].y;
        r5 = a00 * t[3].x + a01 * t[3].y + a10 * b[3].x + a11 * b[3].y;
        r6 = a00 * t[4].x + a01 * t[4].y + a10 * b[4].x + a11 * b[4].y;
        r4 = (q2 + r4) * 0.5f; r5 = (q3 + r5) * 0.5f; r6 = (q4 + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = q2; r5 = q3; r6 = q4 * 0.5f;
    }
    r2 = (q0 - r2) * 0.5f; r3 = (q1 - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    constexpr int BORDER = 5;
    if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
        const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        const float scale = (x < BORDER ? border[x] : 1.f) * (x >= w - BORDER ? border[w - x - 1] : 1.f) *
                            (y < BORDER ? border[y] : 1.f) * (y >= h - BORDER ? border[h - y - 1] : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

// ---- a layer's first matrices, from the flow the layer starts with: 0 = zero, 2 = the coarser layer's flow resized x 2,
// 3 = a flow already at this layer's size (the caller's initial flow after fb_area_kernel; coarsest layer only)
struct FbStart { FbLayer L; int mode; const float2* flow_in; int pw, ph; double inv_x, inv_y; float* M; };

__device__ __forceinline__ float2 fb_start_flow(const FbStart& a, int x, int y) {
    if (a.mode == 2) {
        int x0, x1, y0, y1; float a1, b1;
        resize_axis(x, a.pw, a.inv_x, &x0, &x1, &a1);
        resize_axis(y, a.ph, a.inv_y, &y0, &y1, &b1);
        const float a0 = 1.0f - a1, b0 = 1.0f - b1;
        const float2 p00 = a.flow_in[(size_t)y0 * a.pw + x0], p01 = a.flow_in[(size_t)y0 * a.pw + x1];
        const float2 p10 = a.flow_in[(size_t)y1 * a.pw + x0], p11 = a.flow_in[(size_t)y1 * a.pw + x1];
        float2 f;
        { const float h0 = p00.x * a0 + p01.x * a1, h1 = p10.x * a0 + p11.x * a1; f.x = (h0 * b0 + h1 * b1) * 2.0f; }
        { const float h0 = p00.y * a0 + p01.y * a1, h1 = p10.y * a0 + p11.y * a1; f.y = (h0 * b0 + h1 * b1) * 2.0f; }
        return f;
    }
    if (a.mode == 3) return a.flow_in[(size_t)y * a.L.w + x];      // the caller's flow, already brought to this layer by fb_area_kernel
    return make_float2(0.f, 0.f);
}

// OPTFLOW_USE_INITIAL_FLOW: INTER_AREA of the caller's flow (pw x ph) onto the coarsest layer (w x h), times the layer's scale, in the
// order OpenCV's resizeArea_ works in (the oracle's): every source row of a layer pixel's footprint (~2^K x 2^K source pixels) gives its
// horizontal weighted sums (u, v, weight: f64 chains over the columns), then the rows' sums are accumulated with the row weights.
// One 64-lane workgroup per layer pixel: the footprint comes into LDS with coalesced loads (rows one float2 apart in pitch, so that
// lanes walking different rows hit different banks), lane r walks row r, lane 0 the rows' sums.  (Round 5, first form: a thread per
// layer pixel walking its footprint in global memory, 1,024 loads 256 bytes apart from lane to lane, then one 1,024-term chain: 170 us.)
struct FbArea { const float2* src; int pw, ph, w, h, max_n; float scale; float2* dst; };
__global__ __launch_bounds__(64) void fb_area_kernel(const FbArea a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_area_lds[];
    double* wx = reinterpret_cast<double*>(fb_area_lds);
    double* wy = wx + a.max_n;
    double* rsum = wy + a.max_n;                                   // [max_n][3]: a row's sums
    float2* blk = reinterpret_cast<float2*>(rsum + 3 * a.max_n);   // [ny][nx + 1]
    const int x = (int)blockIdx.x % a.w, y = (int)blockIdx.x / a.w, lane = threadIdx.x;
    const double fx = (double)a.pw / a.w, fy = (double)a.ph / a.h;
    const double xa = x * fx, xb = (x + 1) * fx, ya = y * fy, yb = (y + 1) * fy;
    const int x_lo = (int)floor(xa), y_lo = (int)floor(ya);
    int x_hi = (int)ceil(xb), y_hi = (int)ceil(yb);
    x_hi = x_hi < a.pw ? x_hi : a.pw; y_hi = y_hi < a.ph ? y_hi : a.ph;
    const int nx = x_hi - x_lo, ny = y_hi - y_lo, pitch = nx + 1;
    for (int i = lane; i < nx; i += 64) wx[i] = fmin((double)(x_lo + i + 1), xb) - fmax((double)(x_lo + i), xa);
    for (int i = lane; i < ny; i += 64) wy[i] = fmin((double)(y_lo + i + 1), yb) - fmax((double)(y_lo + i), ya);
    for (int i0 = 0; i0 < nx * ny; i0 += 64 * 8) {              // eight loads per lane in flight
        float2 t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 + 64 * q + lane;
            if (i < nx * ny) { const int r = i / nx; t[q] = a.src[(size_t)(y_lo + r) * a.pw + x_lo + (i - r * nx)]; }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 + 64 * q + lane;
            if (i < nx * ny) { const int r = i / nx; blk[r * pitch + (i - r * nx)] = t[q]; }
        }
    }
    __syncthreads();
    for (int r = lane; r < ny; r += 64) {
        const float2* br = blk + r * pitch;
        double bx = 0, by = 0, bw = 0;
        for (int c = 0; c < nx; ++c) {
            const double wgt = wx[c];
            const float2 v = br[c];
            bx += wgt * v.x; by += wgt * v.y; bw += wgt;
        }
        rsum[3 * r] = bx; rsum[3 * r + 1] = by; rsum[3 * r + 2] = bw;
    }
    __syncthreads();
    if (lane == 0) {
        double sx = 0, sy = 0, sw = 0;
        for (int r = 0; r < ny; ++r) {
            const double wgt = wy[r];
            sx += wgt * rsum[3 * r]; sy += wgt * rsum[3 * r + 1]; sw += wgt * rsum[3 * r + 2];
        }
        a.dst[(size_t)y * a.w + x] = make_float2((float)(sx / sw * (double)a.scale), (float)(sy / sw * (double)a.scale));
    }
}

__global__ __launch_bounds__(256) void fb_start_kernel(const FbStart a) {
    // 64 x 4 pixels per workgroup, dealt to the XCDs in contiguous row-major runs like fb_iter_kernel's tiles: the rows of R1 a
    // workgroup gathers from are the rows its neighbours above / below gather from (the 2-D grid fetched 1.85 x the planes' bytes)
    const int tiles_x = (a.L.w + 63) / 64;
    const int per_xcd = gridDim.x / 8;
    const int tile = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    const int x = (tile % tiles_x) * 64 + (threadIdx.x & 63), y = (tile / tiles_x) * 4 + (threadIdx.x >> 6);
    if (x >= a.L.w || y >= a.L.h) return;
    const size_t plane = (size_t)a.L.w * a.L.h, o = (size_t)y * a.L.w + x;
    float q[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) q[c] = a.L.R0[c * plane + o];
    const float2 f = fb_start_flow(a, x, y);
    float m[5];
    fb_matrices(a.L, x, y, f.x, f.y, q, m);
#pragma unroll
    for (int c = 0; c < 5; ++c) a.M[c * plane + o] = m[c];
}

// ---- one update: FarnebackUpdateFlow_Blur on a 32 x 16 tile (window radius m <= 7; window sums in f64, rows then columns, ascending)
// and -- except in a layer's last update -- FarnebackUpdateMatrices for the tile's own pixels from the flow just solved: the next
// update's matrices are written where the flow they depend on is produced, and the flow itself is stored only when somebody reads it
// (the next layer's start, the caller).
constexpr int kTX = 32, kTY = 16, kMaxM = 7;
struct FbIter {
    FbLayer L; int m;
    const float* M_in;                         // [5][h][w]
    float* M_out;                              // or nullptr (last update of the layer)
    float2* flow_out;                          // or nullptr
    float4* out_entries; float nx, ny;         // layer 0, last update: the per-pixel records (cv-decoder/src/lib.rs:262-269), or nullptr
};

template <int M_>
__global__ __launch_bounds__(256) void fb_iter_kernel(const FbIter a) {
    constexpr int HW = kTX + 2 * M_, HH = kTY + 2 * M_, WIN = 2 * M_ + 1;
    static_assert(5 * HW <= 256, "one thread per (channel, column) of the tile + halo");
    // the row sums (f64) of all five channels: the only LDS array.  Round 5: the f32 tile + halo used to be staged in LDS as well
    // (52.8 KB per workgroup at m = 6 -> 3 workgroups per CU, and a tile is a chain of two global round trips and several LDS passes:
    // the launch was bound by how many such chains a CU had in flight); now a thread takes its column of the tile + halo straight
    // from global memory into registers and adds up the 16 row windows there.
    __shared__ __attribute__((aligned(16))) double sV[5][kTY][HW];
    const int w = a.L.w, h = a.L.h;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): every XCD takes a contiguous run of tiles in row-major order, so the
    // halo a tile shares with its neighbours is in ONE L2 (the plain 2-D grid fetched the M planes 2.4 x from HBM: rocprofv3 FETCH_SIZE)
    const int tiles_x = (w + kTX - 1) / kTX;
    const int per_xcd = gridDim.x / 8;
    const int tile = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    if (tile >= tiles_x * ((h + kTY - 1) / kTY)) return;
    const int x0 = (tile % tiles_x) * kTX, y0 = (tile / tiles_x) * kTY;
    const size_t plane = (size_t)w * h;
    // one thread = two horizontally adjacent pixels of the tile; their own coefficients (for the next update's matrices) are
    // requested now, long before the flow that the rest of those matrices depends on exists
    const int ty = threadIdx.x / (kTX / 2), tx = 2 * (threadIdx.x - ty * (kTX / 2));
    const int y = y0 + ty;
    const bool in0 = x0 + tx < w && y < h, in1 = x0 + tx + 1 < w && y < h;
    // window sums over rows: thread (channel, column) loads the column's kTY + 2m values (replicate border = the clamped pixel's
    // matrices) and makes the kTY sums, each ascending over its 2m + 1 rows in f64: the oracle's order
    if (threadIdx.x < 5 * HW) {
        const int c = threadIdx.x / HW, hx = threadIdx.x - c * HW;
        int x = x0 - M_ + hx;
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        const float* p = a.M_in + c * plane + x;
        float v[HH];
        if (y0 - M_ >= 0 && y0 + kTY + M_ <= h) {
#pragma unroll
            for (int j = 0; j < HH; ++j) v[j] = p[(size_t)(y0 - M_ + j) * w];
        } else {
#pragma unroll
            for (int j = 0; j < HH; ++j) {
                int yy = y0 - M_ + j;
                yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
                v[j] = p[(size_t)yy * w];
            }
        }
#pragma unroll
        for (int q = 0; q < kTY; ++q) {
            double t = 0;
#pragma unroll
            for (int j = 0; j < WIN; ++j) t += v[q + j];
            sV[c][q][hx] = t;
        }
    }
    fb_f2u q01[5];
    if (a.M_out && in0) {
        const float* p = a.L.R0 + (size_t)y * w + x0 + tx;
        if (in1) {
#pragma unroll
            for (int c = 0; c < 5; ++c) q01[c] = *reinterpret_cast<const fb_f2u*>(p + c * plane);
        } else {
#pragma unroll
            for (int c = 0; c < 5; ++c) { q01[c].x = p[c * plane]; q01[c].y = 0.f; }
        }
    }
    __syncthreads();
    double s[2][5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        double v[WIN + 1];                        // WIN + 1 is even and tx is even: 16-byte reads, conflict-free across the wave's four rows
#pragma unroll
        for (int j = 0; j < WIN + 1; j += 2) {
            const double2 p = *reinterpret_cast<const double2*>(&sV[c][ty][tx + j]);
            v[j] = p.x; v[j + 1] = p.y;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            double t = 0;
#pragma unroll
            for (int j = 0; j < WIN; ++j) t += v[q + j];
            s[q][c] = t;
        }
        // (the sums are finished HERE: left to itself the scheduler keeps every channel's 2m + 2 row sums in registers)
        asm volatile("" : "+v"(s[0][c]), "+v"(s[1][c]));
    }
    const double scale = 1.0 / (WIN * WIN);
    float u[2], v[2], mm[2][5];
    const bool in[2] = {in0, in1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const double g11 = s[q][0] * scale, g12 = s[q][1] * scale, g22 = s[q][2] * scale, h1 = s[q][3] * scale, h2 = s[q][4] * scale;
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        u[q] = (float)((g11 * h2 - g12 * h1) * idet); v[q] = (float)((g22 * h1 - g12 * h2) * idet);
    }
    // the next update's matrices of both pixels before any store: their gathers overlap (a store in between would order them)
    if (a.M_out) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (in[q]) {
                const float qq[5] = {q ? q01[0].y : q01[0].x, q ? q01[1].y : q01[1].x, q ? q01[2].y : q01[2].x, q ? q01[3].y : q01[3].x,
                                     q ? q01[4].y : q01[4].x};
                fb_matrices(a.L, x0 + tx + q, y, u[q], v[q], qq, mm[q]);
            }
    }
    if (!in0) return;
    const int x = x0 + tx;
    const size_t o = (size_t)y * w + x;
    if (in1) {                                   // both pixels: 8- and 16-byte stores (any 4-byte alignment) where the two are neighbours in memory
        if (a.flow_out) {
            typedef float fb_f4u __attribute__((ext_vector_type(4), aligned(4)));
            fb_f4u f4; f4.x = u[0]; f4.y = v[0]; f4.z = u[1]; f4.w = v[1];
            *reinterpret_cast<fb_f4u*>(a.flow_out + o) = f4;
        }
        if (a.out_entries) {
            a.out_entries[o] = make_float4(((float)x + 0.5f) * a.nx, ((float)y + 0.5f) * a.ny, u[0] * a.nx, v[0] * a.ny);
            a.out_entries[o + 1] = make_float4(((float)(x + 1) + 0.5f) * a.nx, ((float)y + 0.5f) * a.ny, u[1] * a.nx, v[1] * a.ny);
        }
        if (a.M_out) {
#pragma unroll
            for (int c = 0; c < 5; ++c) { fb_f2u m2; m2.x = mm[0][c]; m2.y = mm[1][c]; *reinterpret_cast<fb_f2u*>(a.M_out + c * plane + o) = m2; }
        }
    } else {
        if (a.flow_out) a.flow_out[o] = make_float2(u[0], v[0]);
        if (a.out_entries) a.out_entries[o] = make_float4(((float)x + 0.5f) * a.nx, ((float)y + 0.5f) * a.ny, u[0] * a.nx, v[0] * a.ny);
        if (a.M_out) {
#pragma unroll
            for (int c = 0; c < 5; ++c) a.M_out[c * plane + o] = mm[0][c];
        }
    }
}

template <int M_>
void launch_iter(const FbIter& a, hipStream_t s) {
    const int tiles = ((a.L.w + kTX - 1) / kTX) * ((a.L.h + kTY - 1) / kTY);
    hipLaunchKernelGGL((fb_iter_kernel<M_>), dim3((tiles + 7) / 8 * 8), dim3(256), 0, s, a);
}

}  // namespace

namespace ofps {

// layers k = 0 .. result that calcOpticalFlowFarneback runs (a layer under 32 px ends the pyramid)
int farneback_layers(int W, int H, int levels) {
    double scale = 1.0;
    int k;
    for (k = 0; k < levels; ++k) {
        scale *= 0.5;
        if (W * scale < 32 || H * scale < 32) break;
    }
    return k;
}

// d_prev / d_cur: u8 luma on the device (row pitch `stride`).  d_init: nullptr or W x H float2 (OPTFLOW_USE_INITIAL_FLOW).  d_flow (W x H
// float2) and / or d_entries (W x H float4 records) receive the result.  Everything is enqueued on ctx->stream.
// Everything farneback_flow_device refuses for a geometry / parameter set, without touching the device: the stream forms call it when a
// stream's FIRST frame arrives (no flow runs yet), so that a stream never accepts a frame and fails the next.
int farneback_check_params(ofps_hip_ctx* ctx, int W, int H, int levels, int winsize, int poly_n) {
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1, "farneback: bad geometry W=%d H=%d", W, H);
    OFPS_REQUIRE(ctx, levels >= 0 && levels <= 16 && winsize >= 1 && (winsize & 1) && poly_n >= 1,
                 "farneback: levels=%d winsize=%d poly_n=%d out of range", levels, winsize, poly_n);
    if (winsize / 2 > kMaxM || poly_n > kMaxPolyN)
        return set_error(ctx, OFPS_HIP_EUNSUPPORTED, "farneback: winsize %d > %d or poly_n %d > %d has no kernel", winsize, 2 * kMaxM + 1, poly_n, kMaxPolyN);
    const int K = farneback_layers(W, H, levels);
    if (K >= kMaxLayers || W > 16384)
        return set_error(ctx, OFPS_HIP_EUNSUPPORTED, "farneback: %d pyramid layers above the frame (max %d) or width %d > 16384 has no kernel", K, kMaxLayers - 1, W);
    int ntaps = 0;
    for (int k = 0; k <= K; ++k) {
        double scale = 1.0;
        for (int i = 0; i < k; ++i) scale *= 0.5;
        FbBlur blur;
        if (!make_blur(k, &blur) || ntaps + 2 * blur.r + 1 > kMaxTaps) return set_error(ctx, OFPS_HIP_EUNSUPPORTED, "farneback: layer %d needs too many blur taps", k);
        ntaps += 2 * blur.r + 1;
        if (k >= 1 && round_half_even(W * scale) == W && round_half_even(H * scale) == H)
            return set_error(ctx, OFPS_HIP_EUNSUPPORTED, "farneback: degenerate layer %d", k);
    }
    return OFPS_HIP_OK;
}

// ---- the call's geometry, taps and workspace
struct FbPlan {
    int W, H, K, poly_n;
    double poly_sigma;
    FbPyr Y;
    FbPoly P;
    size_t r_off[kMaxLayers], r_px, px;
    int vblocks;
    float *T, *I, *R, *Mb[2];
    float2* Fp[2];
    uint64_t gen;
    // expansion planes of layer k in R slot `slot` (three slots per layer: the frames of a stream's two pairs in flight)
    float* Rk(int k, int slot) const { return R + (size_t)ofps_hip_ctx::kFbSlots * 5 * r_off[k] + (size_t)slot * 5 * Y.w[k] * Y.h[k]; }
};

int fb_plan(ofps_hip_ctx* ctx, int W, int H, int levels, int winsize, int poly_n, double poly_sigma, FbPlan* pl) {
    const int rc_params = farneback_check_params(ctx, W, H, levels, winsize, poly_n);
    if (rc_params != OFPS_HIP_OK) return rc_params;
    FbPlan& q = *pl;
    q.W = W; q.H = H; q.poly_n = poly_n; q.poly_sigma = poly_sigma;
    const int K = q.K = farneback_layers(W, H, levels);
    make_poly(poly_n, poly_sigma, &q.P);
    FbPyr& Y = q.Y;
    Y = FbPyr{};
    Y.K = K;
    size_t t_floats = 0, i_floats = 0;
    q.r_px = 0;
    int ntaps = 0;
    q.vblocks = 0;
    for (int k = 0; k <= K; ++k) {
        double scale = 1.0;
        for (int i = 0; i < k; ++i) scale *= 0.5;
        Y.w[k] = round_half_even(W * scale); Y.h[k] = round_half_even(H * scale);
        FbBlur blur;
        if (!make_blur(k, &blur) || ntaps + 2 * blur.r + 1 > kMaxTaps) return set_error(ctx, OFPS_HIP_EUNSUPPORTED, "farneback: layer %d needs too many blur taps", k);
        Y.r[k] = blur.r; Y.toff[k] = ntaps;
        memcpy(Y.taps + ntaps, blur.taps, sizeof(float) * (size_t)(2 * blur.r + 1));
        ntaps += 2 * blur.r + 1;
        Y.inv_x[k] = 1.0 / ((double)Y.w[k] / W); Y.inv_y[k] = 1.0 / ((double)Y.h[k] / H);
        q.r_off[k] = q.r_px; q.r_px += (size_t)Y.w[k] * Y.h[k];
        Y.blk0[k] = q.vblocks;
        if (k >= 1) {
            if (Y.w[k] == W && Y.h[k] == H) return set_error(ctx, OFPS_HIP_EUNSUPPORTED, "farneback: degenerate layer %d", k);
            Y.t_off[k] = t_floats; t_floats += (size_t)2 * H * 2 * Y.w[k];
            Y.i_off[k] = i_floats; i_floats += (size_t)2 * Y.w[k] * Y.h[k];
            q.vblocks += (blur.r <= kShortR ? (Y.w[k] + 63) / 64 : (Y.w[k] + 15) / 16) * ((Y.h[k] + 3) / 4);
        }
    }
    Y.blk0[K + 1] = q.vblocks;
    q.px = (size_t)W * H;
    // T | I (layers >= 1) | R [layer][3 slots][5][h][w] | M x 2 [5][H][W] | two flow planes [H][W] float2
    const size_t floats = t_floats + i_floats + (size_t)ofps_hip_ctx::kFbSlots * 5 * q.r_px + 2 * 5 * q.px + 2 * 2 * q.px;
    auto* base = static_cast<float*>(scratch(ctx, S_FB_WORK, floats * sizeof(float)));
    if (!base) return OFPS_HIP_ENOMEM;
    q.T = base; q.I = q.T + t_floats; q.R = q.I + i_floats;
    q.Mb[0] = q.R + (size_t)ofps_hip_ctx::kFbSlots * 5 * q.r_px; q.Mb[1] = q.Mb[0] + 5 * q.px;
    q.Fp[0] = reinterpret_cast<float2*>(q.Mb[1] + 5 * q.px); q.Fp[1] = q.Fp[0] + q.px;
    q.gen = ctx->scratch[S_FB_WORK].gen;
    // the cache of expanded frames belongs to one workspace, geometry and parameter set
    ofps_hip_ctx::FbCache& fc = ctx->fb_cache;
    if (fc.gen != q.gen || fc.W != W || fc.H != H || fc.K != K || fc.poly_n != poly_n || fc.poly_sigma != poly_sigma) {
        for (auto& id : fc.id) id = 0;
        fc.gen = q.gen; fc.W = W; fc.H = H; fc.K = K; fc.poly_n = poly_n; fc.poly_sigma = poly_sigma;
    }
    return OFPS_HIP_OK;
}

// pyramid + polynomial expansion of one or two frames into R slots, on stream st.  The T / I planes are the pyramid's temporaries and
// shared by every prepare of the context: a prepare on ANOTHER stream than the previous one waits for it -- through an event recorded then,
// on the previous prepare's stream (a record covers everything enqueued before it); prepares that follow each other on one stream, the
// steady state of both the read-ahead and the synchronous forms, cost no event at all.
static int fb_order_behind_last_prepare(ofps_hip_ctx* ctx, hipStream_t st) {
    if (!ctx->fb_prep_recorded || ctx->fb_prep_stream == st || ctx->fb_prep_synced == st) return OFPS_HIP_OK;
    if (!ctx->fb_prep_done) OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->fb_prep_done, hipEventDisableTiming));
    OFPS_HIP_TRY(ctx, hipEventRecord(ctx->fb_prep_done, ctx->fb_prep_stream));
    OFPS_HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->fb_prep_done, 0));
    ctx->fb_prep_synced = st;
    return OFPS_HIP_OK;
}
int fb_prepare(ofps_hip_ctx* ctx, const FbPlan& pl, const uint8_t* const* imgs, const int* slots, int n_img, int stride, hipStream_t st) {
    const FbPyr& Y = pl.Y;
    const int K = pl.K, W = pl.W, H = pl.H;
    {
        const int rc_order = fb_order_behind_last_prepare(ctx, st);
        if (rc_order != OFPS_HIP_OK) return rc_order;
    }
    const uint8_t* i0 = imgs[0];
    const uint8_t* i1 = imgs[n_img - 1];
    // ---- pyramid above layer 0: two launches
    if (K >= 1) {
        // LDS rows: margins of the longest filter (a multiple of 4 bytes, so that the dword copies stay aligned), starts 1 dword (mod 64) apart
        // (ds_read_u8: 32 banks of 4 bytes, lanes 0-31 and 32-63 apart; row starts 2 dwords (mod 32) apart keep a lane pair's two
        // bytes clear of the next row's bank when they straddle a dword).  Three row pitches are compiled in (immediate offsets).
        const int PAD = (Y.r[K] + 3) & ~3;
        const int need = W + 2 * PAD;
        const dim3 grid((H + kPR - 1) / kPR, n_img, K);
#define OFPS_FB_PYR_H(RS_) hipLaunchKernelGGL((fb_pyr_h_kernel<RS_>), grid, dim3(256), (size_t)kPR * RS_ + (size_t)(2 * Y.r[K] + 1) * sizeof(float), st, \
                                              i0, i1, W, H, stride, Y, PAD, pl.T)
        if (need <= kPyrRS0) OFPS_FB_PYR_H(kPyrRS0);
        else if (need <= kPyrRS1) OFPS_FB_PYR_H(kPyrRS1);
        else {                                                                        // 135 KB of dynamic LDS: above the 64 KB a launch gets unasked
            OFPS_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&fb_pyr_h_kernel<kPyrRS2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)(kPR * kPyrRS2 + (2 * Y.r[K] + 1) * sizeof(float))));
            OFPS_FB_PYR_H(kPyrRS2);
        }
#undef OFPS_FB_PYR_H
        hipLaunchKernelGGL(fb_pyr_v_kernel, dim3(pl.vblocks * n_img), dim3(256), 0, st, (const float*)pl.T, W, H, Y, n_img, pl.I);
    }
    // ---- polynomial expansion of every layer and the frames: one launch (layer 0 blurs the u8 frame while it fills its tiles)
    {
        FbExp E{};
        int blk = 0;
        for (int k = 0; k <= K; ++k)
            for (int z = 0; z < n_img; ++z) {
                FbExpJob& J = E.job[E.njobs++];
                J.w = Y.w[k]; J.h = Y.h[k]; J.R = pl.Rk(k, slots[z]); J.blk0 = blk; J.tiles_x = (J.w + kPX - 1) / kPX;
                if (k == 0) { J.u8 = imgs[z]; J.stride = stride; J.I = nullptr; }
                else { J.u8 = nullptr; J.stride = 0; J.I = pl.I + Y.i_off[k] + (size_t)z * J.w * J.h; }
                blk += J.tiles_x * ((J.h + kPY - 1) / kPY);
            }
        const int poly_n = pl.poly_n;
        const size_t lds = (size_t)((kPY + 2 * poly_n) * (kPX + 2 * poly_n) + 3 * kPY * (kPX + 2 * poly_n)) * sizeof(float);
        const float t_c = Y.taps[Y.toff[0] + 1], t_s = Y.taps[Y.toff[0] + 2];
        if (poly_n == 7) hipLaunchKernelGGL((fb_polyexp_kernel<7>), dim3(blk), dim3(256), lds, st, E, pl.P, t_c, t_s);
        else if (poly_n == 5) hipLaunchKernelGGL((fb_polyexp_kernel<5>), dim3(blk), dim3(256), lds, st, E, pl.P, t_c, t_s);
        else hipLaunchKernelGGL((fb_polyexp_kernel<0>), dim3(blk), dim3(256), lds, st, E, pl.P, t_c, t_s);
    }
    OFPS_HIP_TRY(ctx, hipGetLastError());
    ctx->fb_prep_recorded = true; ctx->fb_prep_stream = st; ctx->fb_prep_synced = nullptr;
    return OFPS_HIP_OK;
}

// which R slot holds the expansion of the stream frame `id` (0 = none does)
int fb_find_slot(const ofps_hip_ctx::FbCache& fc, uint64_t id) {
    if (id == 0) return -1;
    for (int i = 0; i < ofps_hip_ctx::kFbSlots; ++i)
        if (fc.id[i] == id) return i;
    return -1;
}
// a slot to overwrite: an empty one, else the one with the oldest frame -- never `keep` (the other frame of the pair being computed)
int fb_victim_slot(const ofps_hip_ctx::FbCache& fc, int keep) {
    int best = -1;
    for (int i = 0; i < ofps_hip_ctx::kFbSlots; ++i) {
        if (i == keep) continue;
        if (best < 0 || fc.id[i] < fc.id[best]) best = i;
    }
    return best;
}

// Stream forms (ofps_hip_lk_push_frame[_async]): the NEW frame's pyramid + expansion, enqueued on the upload's stream right behind the
// upload -- with another ticket in flight that is beside the previous pair's flow, whose coarse layers are a chain of dependent round trips
// that leaves most of the device idle (round 6).  The flow call for the pair then finds both frames' planes by their ids.
// A frame's planes are written into the slot of the oldest frame: its last reader is the flow of a ticket that has been collected.
int farneback_prepare_device(ofps_hip_ctx* ctx, const uint8_t* d_img, int W, int H, int stride, int levels, int winsize, int poly_n, double poly_sigma,
                             uint64_t id, hipStream_t st) {
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && stride >= W && id != 0, "farneback_prepare: bad arguments");
    FbPlan pl;
    int rc = fb_plan(ctx, W, H, levels, winsize, poly_n, poly_sigma, &pl);
    if (rc != OFPS_HIP_OK) return rc;
    ofps_hip_ctx::FbCache& fc = ctx->fb_cache;
    if (fb_find_slot(fc, id) >= 0) return OFPS_HIP_OK;
    const int slot = fb_victim_slot(fc, -1);
    fc.id[slot] = 0;
    rc = fb_prepare(ctx, pl, &d_img, &slot, 1, stride, st);
    if (rc != OFPS_HIP_OK) return rc;
    fc.id[slot] = id;
    return OFPS_HIP_OK;
}

// the caller has made `s` wait for everything enqueued so far on the stream of the latest prepare (the stream forms' `uploaded` event):
// the flow on `s` needs no event of its own
void farneback_mark_ordered(ofps_hip_ctx* ctx, hipStream_t s) { ctx->fb_prep_synced = s; }

int farneback_flow_device(ofps_hip_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_cur, int W, int H, int stride, int levels, int winsize,
                          int iters, int poly_n, double poly_sigma, const float2* d_init, float2* d_flow, float4* d_entries,
                          uint64_t prev_id, uint64_t cur_id) {
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && stride >= W, "farneback: bad geometry W=%d H=%d stride=%d", W, H, stride);
    OFPS_REQUIRE(ctx, iters >= 1 && iters <= 64, "farneback: iters=%d out of range", iters);
    OFPS_REQUIRE(ctx, d_flow || d_entries, "farneback: no output");
    hipStream_t s = ctx->stream;
    FbPlan pl;
    int rc = fb_plan(ctx, W, H, levels, winsize, poly_n, poly_sigma, &pl);
    if (rc != OFPS_HIP_OK) return rc;
    const FbPyr& Y = pl.Y;
    const int K = pl.K;
    float* const* Mb = pl.Mb;
    float2* const* Fp = pl.Fp;
    // Stream forms (prev_id / cur_id != 0: frames of ofps_hip_lk_push_frame[_async]): a frame's expansion planes may be there already --
    // the first frame's from the previous pair (its second frame), the second frame's from farneback_prepare_device on the upload's
    // stream -- if they were made for the same frame id, geometry and parameters and the workspace has not moved since (fb_plan).
    // Whatever is missing goes through the pyramid and the expansion here, on ctx->stream.
    ofps_hip_ctx::FbCache& fc = ctx->fb_cache;
    int slot_prev = fb_find_slot(fc, prev_id), slot_cur = fb_find_slot(fc, cur_id);
    if (slot_prev >= 0) ctx->fb_cache_hits += 1;
    {
        const uint8_t* imgs[2]; int slots[2]; int n = 0;
        if (slot_prev < 0) { slot_prev = fb_victim_slot(fc, slot_cur); fc.id[slot_prev] = 0; imgs[n] = d_prev; slots[n++] = slot_prev; }
        if (slot_cur < 0) {
            int v = -1;                                            // not the first frame's slot
            for (int i = 0; i < ofps_hip_ctx::kFbSlots; ++i)
                if (i != slot_prev && (v < 0 || fc.id[i] < fc.id[v])) v = i;
            slot_cur = v; fc.id[slot_cur] = 0; imgs[n] = d_cur; slots[n++] = slot_cur;
        }
        if (n) {
            rc = fb_prepare(ctx, pl, imgs, slots, n, stride, s);
            if (rc != OFPS_HIP_OK) return rc;
        } else {
            // both frames' planes are there: the flow must come after the prepare that made the newer ones.  The stream forms have ordered
            // ctx->stream behind the upload's stream already (ofps::farneback_mark_ordered); anybody else gets the event
            rc = fb_order_behind_last_prepare(ctx, s);
            if (rc != OFPS_HIP_OK) return rc;
        }
    }
    // ---- layers, coarsest first
    int pw = 0, ph = 0;
    const float2* coarse = nullptr;
    for (int k = K; k >= 0; --k) {
        const int w = Y.w[k], h = Y.h[k];
        FbLayer L{pl.Rk(k, slot_prev), pl.Rk(k, slot_cur), w, h};
        FbStart st{};
        st.L = L; st.M = Mb[0];
        if (k == K) {
            if (d_init) {
                double sc = 1.0;
                for (int i = 0; i < k; ++i) sc *= 0.5;
                FbArea ar{d_init, W, H, w, h, 0, (float)sc, Fp[1]};
                ar.max_n = (int)std::ceil(std::max((double)W / w, (double)H / h)) + 2;
                hipLaunchKernelGGL(fb_area_kernel, dim3(w * h), dim3(64), (size_t)ar.max_n * 5 * sizeof(double) + (size_t)ar.max_n * (ar.max_n + 1) * sizeof(float2), s, ar);
                st.mode = 3; st.flow_in = Fp[1];
            } else st.mode = 0;
        } else {
            st.mode = 2; st.flow_in = coarse; st.pw = pw; st.ph = ph;
            st.inv_x = 1.0 / ((double)w / pw); st.inv_y = 1.0 / ((double)h / ph);
        }
        float2* layer_flow = coarse == Fp[0] ? Fp[1] : Fp[0];
        hipLaunchKernelGGL(fb_start_kernel, dim3((((w + 63) / 64) * ((h + 3) / 4) + 7) / 8 * 8), dim3(256), 0, s, st);
        for (int it = 0; it < iters; ++it) {
            const bool last_it = it == iters - 1, last = k == 0 && last_it;
            FbIter a{};
            a.L = L; a.m = winsize / 2;
            a.M_in = Mb[it & 1]; a.M_out = last_it ? nullptr : Mb[(it + 1) & 1];
            a.flow_out = last ? d_flow : (last_it ? layer_flow : nullptr);
            a.out_entries = last ? d_entries : nullptr;
            a.nx = 1.0f / (float)W; a.ny = 1.0f / (float)H;
            switch (a.m) {
                case 0: launch_iter<0>(a, s); break;
                case 1: launch_iter<1>(a, s); break;
                case 2: launch_iter<2>(a, s); break;
                case 3: launch_iter<3>(a, s); break;
                case 4: launch_iter<4>(a, s); break;
                case 5: launch_iter<5>(a, s); break;
                case 6: launch_iter<6>(a, s); break;
                default: launch_iter<7>(a, s); break;
            }
        }
        coarse = layer_flow; pw = w; ph = h;
    }
    OFPS_HIP_TRY(ctx, hipGetLastError());
    fc.id[slot_prev] = prev_id; fc.id[slot_cur] = cur_id;          // (0: a pair on its own -- its planes are nobody's)
    return OFPS_HIP_OK;
}

}  // namespace ofps

extern "C" {

int ofps_hip_flow_cache_hits(ofps_hip_ctx* ctx, uint64_t* count) {
    if (!ctx || !count) return OFPS_HIP_EINVAL;
    *count = ctx->fb_cache_hits;
    return OFPS_HIP_OK;
}

int ofps_hip_farneback_flow_dev(ofps_hip_ctx* ctx, const void* d_prev, const void* d_cur, int W, int H, int stride, int levels, int winsize,
                                int iters, int poly_n, float poly_sigma, const void* d_init_flow, void* d_out_flow, void* d_out_entries) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_prev && d_cur && (d_out_flow || d_out_entries), "farneback_flow: null device pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ofps::farneback_flow_device(ctx, static_cast<const uint8_t*>(d_prev), static_cast<const uint8_t*>(d_cur), W, H, stride, levels, winsize,
                                       iters, poly_n, (double)poly_sigma, static_cast<const float2*>(d_init_flow), static_cast<float2*>(d_out_flow),
                                       static_cast<float4*>(d_out_entries));
}

int ofps_hip_farneback_flow(ofps_hip_ctx* ctx, const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int winsize, int iters,
                            int poly_n, float poly_sigma, const float* init_flow, float* out_flow, float* out_entries) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, prev && cur && (out_flow || out_entries), "farneback_flow: null host pointer");
    OFPS_REQUIRE(ctx, W >= 1 && H >= 1 && stride >= W, "farneback_flow: bad geometry");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)W * H;
    auto* d_frames = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_FRAMES, 2 * px));
    auto* d_flow = static_cast<float2*>(ofps::scratch(ctx, ofps::S_WORK1, px * sizeof(float2)));
    auto* d_ent = out_entries ? static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES, px * sizeof(float4))) : nullptr;
    auto* d_init = init_flow ? static_cast<float2*>(ofps::scratch(ctx, ofps::S_WORK2, px * sizeof(float2))) : nullptr;
    if (!d_frames || !d_flow || (out_entries && !d_ent) || (init_flow && !d_init)) return OFPS_HIP_ENOMEM;
    OFPS_HIP_TRY(ctx, ofps::upload_rows(d_frames, W, prev, stride, W, H, ctx->stream));
    OFPS_HIP_TRY(ctx, ofps::upload_rows(d_frames + px, W, cur, stride, W, H, ctx->stream));
    if (d_init) OFPS_HIP_TRY(ctx, hipMemcpyAsync(d_init, init_flow, px * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    const int rc = ofps::farneback_flow_device(ctx, d_frames, d_frames + px, W, H, W, levels, winsize, iters, poly_n, (double)poly_sigma, d_init,
                                               d_flow, d_ent);
    if (rc != OFPS_HIP_OK) return rc;
    if (out_flow) OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_flow, d_flow, px * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    if (out_entries) OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_entries, d_ent, px * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

}  // extern "C"
