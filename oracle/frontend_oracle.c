/*
 * frontend_oracle.c -- CPU restatement of cv-decoder's frame front-end (TEST INFRASTRUCTURE ONLY; see ofps_oracle.h).
 *
 * cv-decoder/src/lib.rs:98-135: every frame read is (a) sized against the capped motion-field grid (:98-121), (b) with
 * "Process Fullres" = false resized to that grid with imgproc::resize(.., INTER_LINEAR) (:124-133), (c) converted with
 * cvt_color(.., COLOR_BGR2GRAY) (:135).  The arithmetic of (b) and (c) is OpenCV's, which is neither under /root/reference nor
 * installed here: PARITY UNPINNED.  What is restated is OpenCV's own (non-IPP, non-OpenCL) 8-bit code path as published in
 * modules/imgproc/src/resize.cpp and color_rgb.simd.hpp:
 *
 *   resize, INTER_LINEAR, CV_8U (cv::hal::resize -> resizeGeneric_<HResizeLinear<uchar,int,short,2048>, VResizeLinear<uchar,int,short,
 *   FixedPtCast<int,uchar,22>>>):
 *     inv_scale = (double)dst / src;  scale = 1. / inv_scale;                    (dsize given, fx = fy = 0)
 *     per destination column:  fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = floor(fx);  fx -= sx;
 *                              sx < 0 -> (fx, sx) = (0, 0);   sx >= src_w - 1 -> (fx, sx) = (0, src_w - 1)
 *                              alpha = { saturate_cast<short>((1.f - fx) * 2048), saturate_cast<short>(fx * 2048) }      (round half to even)
 *     per destination row the same with fy, sy but WITHOUT the edge rule: the two source rows are clip(sy + k, 0, src_h - 1)
 *     horizontal pass (int):  D = S[sx] * alpha0 + S[sx + 1] * alpha1        (S[sx] * 2048 from the first column with sx + 1 >= src_w on)
 *     vertical pass, the uchar specialisation:  dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2
 *       (its SIMD twin VResizeLinearVec_32s8u computes the same integers; the generic FixedPtCast form (x + (1 << 21)) >> 22 is what the
 *        other element types use -- orc_resize_linear_u8_ex(.., variant 1) evaluates that one, for the external kit to tell builds apart)
 *     exact 2 x 2 reduction (src = 2 * dst in both directions): cv::hal::resize turns INTER_LINEAR into INTER_AREA ("INTER_AREA (fast)
 *       also is equal to INTER_LINEAR"): dst = (a + b + c + d + 2) >> 2 (resizeAreaFast_, ResizeAreaFastVec).
 *     dst size == src size: a copy.
 *   cvtColor BGR2GRAY, CV_8U (RGB2Gray<uchar>): gray = (B * 1868 + G * 9617 + R * 4899 + (1 << 13)) >> 14.
 *
 * The product's kernels (ofps_amd/csrc/frontend.hip) are checked against these functions bit for bit.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ofps_oracle.h"

/* cv-decoder/src/lib.rs:98-121 with aspect_ratio_scale = (1, 1) (the descriptor's value, :10-15): usize arithmetic */
void orc_cv_grid(int W, int H, int max_w, int max_h, int* gw, int* gh) {
    const size_t ax = (size_t)W, ay = (size_t)H;
    const size_t ratio0 = ax * 1, ratio1 = ay * 1;                                   /* :103-106 */
    size_t w = (size_t)max_w, h = (size_t)max_h;                                     /* :107 */
    w = w < ax ? w : ax; h = h < ay ? h : ay;                                        /* :109-112 */
    const size_t wb0 = w, wb1 = w * ratio1 / ratio0;                                 /* :114 */
    const size_t hb0 = h * ratio0 / ratio1, hb1 = h;                                 /* :115 */
    if (wb0 < hb0) { *gw = (int)wb0; *gh = (int)wb1; } else { *gw = (int)hb0; *gh = (int)hb1; }      /* :116-120 */
}

static short sat_short_rne(float v) {                   /* saturate_cast<short>(float): cvRound (round half to even), then saturate */
    const long r = lrintf(v);                           /* the default rounding mode is to-nearest-even */
    return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

/* one axis of the coefficient table; edge_rule: the horizontal axis' (fx, sx) adjustment */
static void linear_axis(int src, int dst, int edge_rule, int* ofs, short* coef) {
    const double inv_scale = (double)dst / src;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (edge_rule) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= src - 1) { f = 0.f; s = src - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = sat_short_rne((1.f - f) * 2048.f);
        coef[2 * d + 1] = sat_short_rne(f * 2048.f);
    }
}

static int clip_row(int y, int h) { return y >= 0 ? (y < h ? y : h - 1) : 0; }

/* src: H rows of W pixels of cn interleaved bytes, `stride` bytes apart; dst: dh x dw x cn dense.  variant 0 = the uchar specialisation
 * (what an 8-bit resize runs), 1 = the generic FixedPtCast form.  Returns 0, or -1 for bad arguments. */
int orc_resize_linear_u8_ex(const uint8_t* src, int W, int H, int stride, int cn, uint8_t* dst, int dw, int dh, int variant) {
    if (!src || !dst || W < 1 || H < 1 || dw < 1 || dh < 1 || cn < 1 || cn > 4 || stride < W * cn) return -1;
    if (dw == W && dh == H) {
        for (int y = 0; y < H; ++y) memcpy(dst + (size_t)y * W * cn, src + (size_t)y * stride, (size_t)W * cn);
        return 0;
    }
    if (W == 2 * dw && H == 2 * dh) {                     /* INTER_LINEAR -> INTER_AREA (fast), scale 2 x 2 */
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < cn; ++c) {
                    const uint8_t* s0 = src + (size_t)(2 * y) * stride + (size_t)(2 * x) * cn + c;
                    const uint8_t* s1 = s0 + stride;
                    dst[((size_t)y * dw + x) * cn + c] = (uint8_t)((s0[0] + s0[cn] + s1[0] + s1[cn] + 2) >> 2);
                }
        return 0;
    }
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
    int* yofs = (int*)malloc(sizeof(int) * (size_t)dh);
    short* alpha = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
    short* beta = (short*)malloc(sizeof(short) * 2 * (size_t)dh);
    int* rows = (int*)malloc(sizeof(int) * 2 * (size_t)dw * cn);
    if (!xofs || !yofs || !alpha || !beta || !rows) { free(xofs); free(yofs); free(alpha); free(beta); free(rows); return -1; }
    linear_axis(W, dw, 1, xofs, alpha);
    linear_axis(H, dh, 0, yofs, beta);
    for (int dy = 0; dy < dh; ++dy) {
        for (int k = 0; k < 2; ++k) {                     /* horizontal pass of the two source rows */
            const uint8_t* S = src + (size_t)clip_row(yofs[dy] + k, H) * stride;
            int* D = rows + (size_t)k * dw * cn;
            for (int dx = 0; dx < dw; ++dx) {
                const int sx = xofs[dx];
                for (int c = 0; c < cn; ++c)
                    D[dx * cn + c] = sx + 1 >= W ? S[sx * cn + c] * 2048
                                                 : S[sx * cn + c] * alpha[2 * dx] + S[(sx + 1) * cn + c] * alpha[2 * dx + 1];
            }
        }
        const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        const int* S0 = rows;
        const int* S1 = rows + (size_t)dw * cn;
        uint8_t* out = dst + (size_t)dy * dw * cn;
        for (int x = 0; x < dw * cn; ++x) {
            if (variant == 0) {
                out[x] = (uint8_t)((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2);
            } else {
                int v = (S0[x] * b0 + S1[x] * b1 + (1 << 21)) >> 22;
                out[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
    }
    free(xofs); free(yofs); free(alpha); free(beta); free(rows);
    return 0;
}

int orc_resize_linear_u8(const uint8_t* src, int W, int H, int stride, int cn, uint8_t* dst, int dw, int dh) {
    return orc_resize_linear_u8_ex(src, W, H, stride, cn, dst, dw, dh, 0);
}

/* the coefficient tables themselves (tests look at them: edge rule, rounding) */
int orc_resize_linear_axis(int src, int dst, int edge_rule, int* ofs, short* coef) {
    if (src < 1 || dst < 1 || !ofs || !coef) return -1;
    linear_axis(src, dst, edge_rule, ofs, coef);
    return 0;
}

/* fmt: 1 = BGR (3 bytes per pixel: what VideoCapture::read hands cv-decoder), 2 = RGBA (4 bytes, ofps::RGBA's order, alpha ignored),
 * 3 = BGRA (4 bytes).  cvtColor's RGB2Gray<uchar> with the coefficients of the channel each byte carries. */
int orc_to_gray_u8(const uint8_t* src, int W, int H, int stride, int fmt, uint8_t* dst) {
    const int cn = fmt == 1 ? 3 : (fmt == 2 || fmt == 3) ? 4 : 0;
    if (!src || !dst || !cn || W < 1 || H < 1 || stride < W * cn) return -1;
    const int ib = fmt == 2 ? 2 : 0, ir = fmt == 2 ? 0 : 2;
    for (int y = 0; y < H; ++y) {
        const uint8_t* s = src + (size_t)y * stride;
        for (int x = 0; x < W; ++x, s += cn)
            dst[(size_t)y * W + x] = (uint8_t)((s[ib] * 1868 + s[1] * 9617 + s[ir] * 4899 + (1 << 13)) >> 14);
    }
    return 0;
}
