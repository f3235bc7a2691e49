/*
 * farneback_oracle.c -- CPU restatement of Farneback's dense optical flow as the reference's cv-decoder calls it (TEST INFRASTRUCTURE
 * ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline; the product never links it).
 *
 *   cv-decoder/src/lib.rs:188-199:  calc_optical_flow_farneback(old_gray, gray, flow, pyr_scale 0.5, levels 5, winsize 13,
 *                                   iterations 3, poly_n 7, poly_sigma 1.5, flags)
 *
 * The arithmetic lives in OpenCV (`opencv = "0.62"` binds whatever libopencv the host has, cv-decoder/Cargo.toml:22; not under
 * /root/reference, not installed here): PARITY UNPINNED.  What is restated is the published algorithm (G. Farneback, "Two-Frame
 * Motion Estimation Based on Polynomial Expansion", SCIA 2003) in the form OpenCV's calcOpticalFlowFarneback gives it (flags = 0:
 * box window, zero initial flow; OPTFLOW_USE_INITIAL_FLOW through `init`):
 *   scales    k = levels .. 0 (levels + 1 layers unless a layer would be smaller than 32 px), scale = 0.5^k, size = round-half-even(W * scale);
 *             every layer is made from the ORIGINAL frame: Gaussian blur (sigma = (1/scale - 1)/2, ksize = max(round(5 sigma) | 1, 3),
 *             reflect-101 border; [1 2 1]/4 at scale 1) then bilinear resize (half-pixel centres); the coarser layer's flow is resized
 *             bilinearly and doubled;
 *   poly exp  separable, radius poly_n, Gaussian applicability sigma poly_sigma, replicate border: five coefficients per pixel
 *             [y, x, y^2, x^2, xy] (the constant term is not kept);
 *   matrices  per pixel: second image's coefficients sampled bilinearly at (x + dx, y + dy), averaged with the first image's, the
 *             2x2 system G = A^T A, h = A^T db with A = [[r4, r6], [r6, r5]]; the outermost 5 px scaled by (0.14, 0.14, 0.4472 x 3);
 *   update    box sums of the five matrix channels over winsize x winsize (replicate border), flow = G^-1 h with det + 1e-3;
 *             `iters` times per scale, the matrices recomputed from the new flow between them.
 * Arithmetic: f32 where OpenCV's CPU path uses float (blur, resize, vertical half of the expansion, matrices), f64 where it uses
 * double (horizontal half of the expansion, window sums, the 2x2 solve), no fused multiply-adds (-ffp-contract=off).  Window sums are
 * direct sums in ascending row / column order (OpenCV slides a running double sum: the same value up to 1 ulp of a double).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ofps_oracle.h"

static int round_half_even(double v) { return (int)nearbyint(v); }      /* cvRound (default rounding mode: to nearest even) */

/* layers calcOpticalFlowFarneback keeps: k = 0 .. *levels (optflowgf.cpp: min_size 32) */
int orc_farneback_layers(int W, int H, int levels) {
    double scale = 1.0;
    int k;
    for (k = 0; k < levels; ++k) {
        scale *= 0.5;
        if (W * scale < 32 || H * scale < 32) break;
    }
    return k;
}

void orc_farneback_layer_size(int W, int H, int k, int* w, int* h) {
    double scale = 1.0;
    for (int i = 0; i < k; ++i) scale *= 0.5;
    *w = round_half_even(W * scale);
    *h = round_half_even(H * scale);
}

/* getGaussianKernel(ksize, sigma, CV_32F): float taps normalised by their double sum; sigma <= 0 with ksize 3: the fixed [1 2 1] / 4 */
int orc_farneback_blur_kernel(int k, float* taps /* >= 2 r + 1 */) {
    double scale = 1.0;
    for (int i = 0; i < k; ++i) scale *= 0.5;
    const double sigma = (1.0 / scale - 1.0) * 0.5;
    int ksize = round_half_even(sigma * 5) | 1;
    if (ksize < 3) ksize = 3;
    const int r = ksize / 2;
    if (sigma <= 0) {
        taps[0] = 0.25f; taps[1] = 0.5f; taps[2] = 0.25f;
        return 1;
    }
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < ksize; ++i) {
        const double x = i - (ksize - 1) * 0.5;
        taps[i] = (float)exp(s2 * x * x);
        sum += taps[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < ksize; ++i) taps[i] = (float)(taps[i] * sum);
    return r;
}

static int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

/* Which published form of OpenCV's separable Gaussian the layers are blurred with (the part of the restatement where OpenCV builds differ
 * among themselves; everything else is scalar code with one form).  Test infrastructure for tools/external_parity/opencv_compare.py:
 *   0  (the SPEC the HIP kernels are built to: every committed vector, every parity test): symmetric pairing in both passes,
 *      s = c[r] * S[0] + sum_j c[r + j] * (S[-j] + S[+j]), separate multiply and add;
 *   1  the row pass as filter.simd.hpp's RowFilter has it for kernels of more than 5 taps (SymmRowSmallFilter serves 3 and 5 only: layers
 *      k >= 2 have 9 .. 159 taps): taps in ascending order, s = sum_k c[k] * S[k - r]; the column pass is SymmColumnFilter's pairing;
 *   2  variant 1 with the multiply-adds fused (v_muladd on builds whose dispatch has FMA3 / NEON: RowVec_32f, SymmColumnVec_32f).
 * Set by orc_farneback_set_blur_variant; read by every orc_farneback_* call of the process. */
static int g_blur_variant = 0;
int orc_farneback_set_blur_variant(int v) {
    const int prev = g_blur_variant;
    if (v >= 0 && v <= 2) g_blur_variant = v;
    return prev;
}

static void gaussian_blur(const float* src, int W, int H, const float* taps, int r, float* tmp, float* dst) {
    const int variant = g_blur_variant;
    const int ascending_rows = variant >= 1 && 2 * r + 1 > 5;
    const int fused = variant == 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float s;
            if (ascending_rows) {
                s = 0.0f;
                for (int k = 0; k <= 2 * r; ++k) {
                    const float v = src[(size_t)y * W + reflect101(x + k - r, W)];
                    s = fused ? fmaf(v, taps[k], s) : s + taps[k] * v;
                }
            } else {
                s = taps[r] * src[(size_t)y * W + x];
                for (int j = 1; j <= r; ++j) {
                    const float p = src[(size_t)y * W + reflect101(x - j, W)] + src[(size_t)y * W + reflect101(x + j, W)];
                    s = fused ? fmaf(p, taps[r + j], s) : s + taps[r + j] * p;
                }
            }
            tmp[(size_t)y * W + x] = s;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float s = taps[r] * tmp[(size_t)y * W + x];
            for (int j = 1; j <= r; ++j) {
                const float p = tmp[(size_t)reflect101(y - j, H) * W + x] + tmp[(size_t)reflect101(y + j, H) * W + x];
                s = fused ? fmaf(p, taps[r + j], s) : s + taps[r + j] * p;
            }
            dst[(size_t)y * W + x] = s;
        }
}

/* resize INTER_LINEAR on `ch` interleaved float channels: source index and weight per destination index (half-pixel centres, clamped) */
static void resize_axis(int dn, int sn, int* idx, float* frac) {
    const double scale = 1.0 / ((double)dn / sn);
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= sn - 1) { f = 0; s = sn - 1; }
        idx[d] = s; frac[d] = f;
    }
}

static void resize_linear(const float* src, int sw, int sh, int ch, float* dst, int dw, int dh, float mul) {
    int* xi = malloc(sizeof(int) * (size_t)dw); float* xf = malloc(sizeof(float) * (size_t)dw);
    int* yi = malloc(sizeof(int) * (size_t)dh); float* yf = malloc(sizeof(float) * (size_t)dh);
    resize_axis(dw, sw, xi, xf); resize_axis(dh, sh, yi, yf);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        const int y0 = yi[y], y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
        const float b1 = yf[y], b0 = 1.0f - b1;
        for (int x = 0; x < dw; ++x) {
            const int x0 = xi[x], x1 = x0 + 1 < sw ? x0 + 1 : sw - 1;
            const float a1 = xf[x], a0 = 1.0f - a1;
            for (int c = 0; c < ch; ++c) {
                const float h0 = src[((size_t)y0 * sw + x0) * ch + c] * a0 + src[((size_t)y0 * sw + x1) * ch + c] * a1;   /* HResizeLinear */
                const float h1 = src[((size_t)y1 * sw + x0) * ch + c] * a0 + src[((size_t)y1 * sw + x1) * ch + c] * a1;
                float v = h0 * b0 + h1 * b1;                                                                                 /* VResizeLinear */
                if (mul != 1.0f) v *= mul;
                dst[((size_t)y * dw + x) * ch + c] = v;
            }
        }
    }
    free(xi); free(xf); free(yi); free(yf);
}

/* FarnebackPrepareGaussian: taps g, x g, x^2 g (float) and the four entries of G^-1 the expansion uses.  With sum g = 1 the moment
 * matrix is [[1,0,0,b,b,0],[0,b,..],[0,0,b,..],[b,0,0,c,b^2,0],[b,0,0,b^2,c,0],[0,..,b^2]] (b = sum x^2 g, c = sum x^4 g):
 * closed-form inverse (OpenCV inverts the same 6x6 numerically). */
void orc_farneback_poly_kernel(int n, double sigma, float* g, float* xg, float* xxg /* each n + 1: index 0 .. n */, double ig[4]) {
    if (sigma < 1.1920929e-07) sigma = n * 0.3;
    double s = 0;
    float* full = malloc(sizeof(float) * (size_t)(2 * n + 1));
    for (int x = -n; x <= n; ++x) { full[x + n] = (float)exp(-x * x / (2 * sigma * sigma)); s += full[x + n]; }
    s = 1.0 / s;
    double b = 0, c = 0;
    for (int x = -n; x <= n; ++x) {
        const float gv = (float)(full[x + n] * s);
        if (x >= 0) { g[x] = gv; xg[x] = (float)(x * gv); xxg[x] = (float)(x * x * gv); }
        b += (double)gv * x * x; c += (double)gv * x * x * x * x;
    }
    free(full);
    /* the taps are floats whose sum is 1 only to rounding: carry the actual zeroth moment like the numerical inverse would */
    double a = 0;
    for (int x = -n; x <= n; ++x) a += (double)g[x < 0 ? -x : x];
    const double B = a * b, C = a * c, D = b * b, A = a * a;      /* G(0,0) = A, G(1,1) = G(0,3) = B, G(3,3) = C, G(3,4) = G(5,5) = D */
    /* 3x3 block [[A,B,B],[B,C,D],[B,D,C]]: inv(0,1) = -B (C - D) / det, inv(1,1) = (A C - B^2) / det, det = (C - D) (A (C + D) - 2 B^2) */
    const double det = (C - D) * (A * (C + D) - 2 * B * B);
    ig[0] = 1.0 / B;                        /* ig11 */
    ig[1] = -B * (C - D) / det;             /* ig03 */
    ig[2] = (A * C - B * B) / det;          /* ig33 */
    ig[3] = 1.0 / D;                        /* ig55 */
}

/* FarnebackPolyExp: src w x h float -> dst w x h x 5 float */
static void poly_exp(const float* src, int w, int h, int n, const float* g, const float* xg, const float* xxg, const double ig[4], float* dst) {
    const double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
#pragma omp parallel
    {
        float* rowbuf = malloc(sizeof(float) * (size_t)(w + 2 * n) * 3);
        float* row = rowbuf + n * 3;
#pragma omp for schedule(static)
        for (int y = 0; y < h; ++y) {
            const float g0 = g[0];
            const float* s0 = src + (size_t)y * w;
            for (int x = 0; x < w; ++x) { row[x * 3] = s0[x] * g0; row[x * 3 + 1] = row[x * 3 + 2] = 0.f; }
            for (int k = 1; k <= n; ++k) {                                   /* vertical half, f32, replicate border */
                const float gk = g[k], g1 = xg[k], g2 = xxg[k];
                const float* a = src + (size_t)(y - k > 0 ? y - k : 0) * w;
                const float* b = src + (size_t)(y + k < h - 1 ? y + k : h - 1) * w;
                for (int x = 0; x < w; ++x) {
                    const float p = a[x] + b[x];
                    const float t0 = row[x * 3] + gk * p;
                    const float t1 = row[x * 3 + 1] + g1 * (b[x] - a[x]);
                    const float t2 = row[x * 3 + 2] + g2 * p;
                    row[x * 3] = t0; row[x * 3 + 1] = t1; row[x * 3 + 2] = t2;
                }
            }
            for (int x = 0; x < n; ++x)                                      /* replicate the first / last pixel's three sums */
                for (int c = 0; c < 3; ++c) { row[(-1 - x) * 3 + c] = row[c]; row[(w + x) * 3 + c] = row[(w - 1) * 3 + c]; }
            float* d = dst + (size_t)y * w * 5;
            for (int x = 0; x < w; ++x) {                                    /* horizontal half, f64 */
                double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
                for (int k = 1; k <= n; ++k) {
                    const double tg = row[(x + k) * 3] + row[(x - k) * 3];
                    const float gk = g[k];
                    b1 += tg * gk;
                    b4 += tg * xxg[k];
                    b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                    b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * gk;
                    b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                    b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * gk;
                }
                d[x * 5 + 1] = (float)(b2 * ig11);
                d[x * 5] = (float)(b3 * ig11);
                d[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
                d[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
                d[x * 5 + 4] = (float)(b6 * ig55);
            }
        }
        free(rowbuf);
    }
}

/* FarnebackUpdateMatrices */
static void update_matrices(const float* R0, const float* R1, const float* flow, int w, int h, float* M) {
    static const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const int BORDER = 5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float* r0 = R0 + ((size_t)y * w + x) * 5;
            const float dx = flow[((size_t)y * w + x) * 2], dy = flow[((size_t)y * w + x) * 2 + 1];
            float fx = x + dx, fy = y + dy;
            const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= y1;
            if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
                const float* p = R1 + ((size_t)y1 * w + x1) * 5;
                const size_t st = (size_t)w * 5;
                const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * p[0] + a01 * p[5] + a10 * p[st] + a11 * p[st + 5];
                r3 = a00 * p[1] + a01 * p[6] + a10 * p[st + 1] + a11 * p[st + 6];
                r4 = a00 * p[2] + a01 * p[7] + a10 * p[st + 2] + a11 * p[st + 7];
                r5 = a00 * p[3] + a01 * p[8] + a10 * p[st + 3] + a11 * p[st + 8];
                r6 = a00 * p[4] + a01 * p[9] + a10 * p[st + 4] + a11 * p[st + 9];
                r4 = (r0[2] + r4) * 0.5f; r5 = (r0[3] + r5) * 0.5f; r6 = (r0[4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = r0[2]; r5 = r0[3]; r6 = r0[4] * 0.5f;
            }
            r2 = (r0[0] - r2) * 0.5f; r3 = (r0[1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
                const float scale = (x < BORDER ? border[x] : 1.f) * (x >= w - BORDER ? border[w - x - 1] : 1.f) *
                                    (y < BORDER ? border[y] : 1.f) * (y >= h - BORDER ? border[h - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            float* m = M + ((size_t)y * w + x) * 5;
            m[0] = r4 * r4 + r6 * r6;
            m[1] = (r4 + r5) * r6;
            m[2] = r5 * r5 + r6 * r6;
            m[3] = r4 * r2 + r6 * r3;
            m[4] = r6 * r2 + r5 * r3;
        }
}

/* FarnebackUpdateFlow_Blur without the matrix update: window sums (replicate border) in f64, ascending rows then ascending columns */
static void update_flow_blur(const float* M, int w, int h, int winsize, float* flow) {
    const int m = winsize / 2;
    const double scale = 1.0 / (winsize * winsize);
#pragma omp parallel
    {
        double* vsum = malloc(sizeof(double) * (size_t)w * 5);
#pragma omp for schedule(static)
        for (int y = 0; y < h; ++y) {
            for (int i = 0; i < w * 5; ++i) vsum[i] = 0;
            for (int j = -m; j <= m; ++j) {
                const int yy = y + j < 0 ? 0 : (y + j > h - 1 ? h - 1 : y + j);
                const float* r = M + (size_t)yy * w * 5;
                for (int i = 0; i < w * 5; ++i) vsum[i] += r[i];
            }
            for (int x = 0; x < w; ++x) {
                double s[5] = {0, 0, 0, 0, 0};
                for (int j = -m; j <= m; ++j) {
                    const int xx = x + j < 0 ? 0 : (x + j > w - 1 ? w - 1 : x + j);
                    for (int c = 0; c < 5; ++c) s[c] += vsum[xx * 5 + c];
                }
                const double g11 = s[0] * scale, g12 = s[1] * scale, g22 = s[2] * scale, h1 = s[3] * scale, h2 = s[4] * scale;
                const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                flow[((size_t)y * w + x) * 2] = (float)((g11 * h2 - g12 * h1) * idet);
                flow[((size_t)y * w + x) * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
            }
        }
        free(vsum);
    }
}

/* One layer's image from the u8 frame: float, blur, resize */
static void make_layer(const uint8_t* img, int W, int H, int stride, int k, int w, int h, float* scratch_a, float* scratch_b, float* scratch_c,
                       float* I) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) scratch_a[(size_t)y * W + x] = (float)img[(size_t)y * stride + x];
    float taps[256];                                   /* k <= 6: 159 taps (orc_farneback_blur_kernel writes 2 r + 1) */
    const int r = orc_farneback_blur_kernel(k, taps);
    gaussian_blur(scratch_a, W, H, taps, r, scratch_b, scratch_c);
    if (w == W && h == H) memcpy(I, scratch_c, sizeof(float) * (size_t)W * H);
    else resize_linear(scratch_c, W, H, 1, I, w, h, 1.0f);
}

/* The intermediate planes of one layer, for stage-wise parity tests: any pointer may be NULL */
int orc_farneback_layer_debug(const uint8_t* img, int W, int H, int stride, int k, int poly_n, double poly_sigma, float* out_I, float* out_R) {
    int w, h;
    orc_farneback_layer_size(W, H, k, &w, &h);
    float* a = malloc(sizeof(float) * (size_t)W * H); float* b = malloc(sizeof(float) * (size_t)W * H); float* c = malloc(sizeof(float) * (size_t)W * H);
    float* I = malloc(sizeof(float) * (size_t)w * h);
    if (!a || !b || !c || !I) { free(a); free(b); free(c); free(I); return -1; }
    make_layer(img, W, H, stride, k, w, h, a, b, c, I);
    if (out_I) memcpy(out_I, I, sizeof(float) * (size_t)w * h);
    if (out_R) {
        float* g = malloc(sizeof(float) * (size_t)(poly_n + 1) * 3);
        double ig[4];
        orc_farneback_poly_kernel(poly_n, poly_sigma, g, g + poly_n + 1, g + 2 * (poly_n + 1), ig);
        poly_exp(I, w, h, poly_n, g, g + poly_n + 1, g + 2 * (poly_n + 1), ig, out_R);
        free(g);
    }
    free(a); free(b); free(c); free(I);
    return 0;
}

/* init: NULL (zero flow at the coarsest layer) or a W x H x 2 flow (OPTFLOW_USE_INITIAL_FLOW: resized to the coarsest layer and scaled).
 * out_flow: W x H x 2 (dx, dy) per pixel of `prev`: prev(x, y) ~ cur(x + dx, y + dy).  -> 0, or -1 on a bad argument / no memory */
int orc_farneback_flow(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int winsize, int iters, int poly_n,
                       double poly_sigma, const float* init, float* out_flow) {
    if (!prev || !cur || !out_flow || W < 1 || H < 1 || stride < W || levels < 0 || levels > 16 || winsize < 1 || (winsize & 1) == 0 || iters < 1 ||
        poly_n < 1 || poly_n > 15)
        return -1;
    const int L = orc_farneback_layers(W, H, levels);
    if (L > 6) return -1;                              /* layer 7 would need 317 blur taps (make_layer holds 256): frames beyond 4096 px with levels >= 7 */
    const size_t px = (size_t)W * H;
    float* a = malloc(sizeof(float) * px); float* b = malloc(sizeof(float) * px); float* c = malloc(sizeof(float) * px);
    float* I = malloc(sizeof(float) * px);
    float* R0 = malloc(sizeof(float) * px * 5); float* R1 = malloc(sizeof(float) * px * 5); float* M = malloc(sizeof(float) * px * 5);
    float* flow = malloc(sizeof(float) * px * 2); float* prev_flow = malloc(sizeof(float) * px * 2);
    float* g = malloc(sizeof(float) * (size_t)(poly_n + 1) * 3);
    int rc = -1;
    if (!a || !b || !c || !I || !R0 || !R1 || !M || !flow || !prev_flow || !g) goto done;
    double ig[4];
    orc_farneback_poly_kernel(poly_n, poly_sigma, g, g + poly_n + 1, g + 2 * (poly_n + 1), ig);
    int pw = 0, ph = 0;
    for (int k = L; k >= 0; --k) {
        int w, h;
        orc_farneback_layer_size(W, H, k, &w, &h);
        float* f = k > 0 ? flow : out_flow;
        if (k == L) {
            if (init) {
                /* resize(flow0, flow, INTER_AREA) * scale: for the exact power-of-two ratios this build supports the area resize is the
                 * mean over the source footprint; restated as the bilinear resize of the same grid only when L == 0 (identity) */
                if (L == 0) memcpy(f, init, sizeof(float) * px * 2);
                else {
                    double scale = 1.0;
                    for (int i = 0; i < k; ++i) scale *= 0.5;
                    /* INTER_AREA, general ratio: box average over [x/sx, (x+1)/sx) with fractional edge weights, in the order OpenCV's
                     * resizeArea_ works in -- every source row's horizontal weighted sum first, the rows' sums accumulated with the row
                     * weights after -- in f64 with unnormalised weights, divided by the footprint's weight at the end */
                    const double fx = (double)W / w, fy = (double)H / h;
                    for (int y = 0; y < h; ++y)
                        for (int x = 0; x < w; ++x) {
                            const double x0 = x * fx, x1 = (x + 1) * fx, y0 = y * fy, y1 = (y + 1) * fy;
                            double sx = 0, sy = 0, sw = 0;
                            for (int yy = (int)floor(y0); yy < (int)ceil(y1) && yy < H; ++yy) {
                                const double wy = fmin(yy + 1, y1) - fmax(yy, y0);
                                double bx = 0, by = 0, bw = 0;
                                for (int xx = (int)floor(x0); xx < (int)ceil(x1) && xx < W; ++xx) {
                                    const double wx = fmin(xx + 1, x1) - fmax(xx, x0);
                                    bx += wx * init[((size_t)yy * W + xx) * 2]; by += wx * init[((size_t)yy * W + xx) * 2 + 1]; bw += wx;
                                }
                                sx += wy * bx; sy += wy * by; sw += wy * bw;
                            }
                            f[((size_t)y * w + x) * 2] = (float)(sx / sw * scale);
                            f[((size_t)y * w + x) * 2 + 1] = (float)(sy / sw * scale);
                        }
                }
            } else {
                memset(f, 0, sizeof(float) * (size_t)w * h * 2);
            }
        } else {
            resize_linear(prev_flow, pw, ph, 2, f, w, h, 2.0f);          /* flow *= 1 / pyr_scale */
        }
        make_layer(prev, W, H, stride, k, w, h, a, b, c, I);
        poly_exp(I, w, h, poly_n, g, g + poly_n + 1, g + 2 * (poly_n + 1), ig, R0);
        make_layer(cur, W, H, stride, k, w, h, a, b, c, I);
        poly_exp(I, w, h, poly_n, g, g + poly_n + 1, g + 2 * (poly_n + 1), ig, R1);
        update_matrices(R0, R1, f, w, h, M);
        for (int i = 0; i < iters; ++i) {
            update_flow_blur(M, w, h, winsize, f);
            if (i < iters - 1) update_matrices(R0, R1, f, w, h, M);
        }
        if (k > 0) memcpy(prev_flow, f, sizeof(float) * (size_t)w * h * 2);
        pw = w; ph = h;
    }
    rc = 0;
done:
    free(a); free(b); free(c); free(I); free(R0); free(R1); free(M); free(flow); free(prev_flow); free(g);
    return rc;
}
