/*
 * ofps_hip.h -- C ABI of libofps_hip.so, the MI355X (gfx950) backend for the OFPS flow hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  A Rust cdylib
 * shim implements the reference's plugin traits on top of these calls:
 *
 *   ofps_hip_sad_flow*      -> Decoder::process_frame          (ofps/src/decoder.rs:45-73); output
 *                              record convention of av-decoder   (av-decoder/src/lib.rs:404-419)
 *   ofps_hip_densify*       -> MotionFieldDensifier::add_vector + MotionField::from
 *                                                               (ofps/src/motion_field.rs:133-190,297-308)
 *   ofps_hip_densify_to_entries -> cv-decoder's downsample stage (cv-decoder/src/lib.rs:244-291)
 *   ofps_hip_detect*        -> Detector::detect_motion          (ofps/src/detection.rs:11,
 *                                                               block-motion-detector/src/lib.rs:49-118)
 *   ofps_hip_almeida*       -> Estimator::estimate              (ofps/src/estimator.rs:19-24,
 *                                                               almeida-estimator/src/lib.rs:100-251)
 *
 * A MotionEntry is 4 consecutive f32 [pos.x, pos.y, motion.x, motion.y] (decoder.rs:40-42), the
 * same record the reference's .mvec files hold (motion-extract/src/main.rs:23-35).
 *
 * Conventions
 *   - every call returns 0 on success or a negative OFPS_HIP_E* code; ofps_hip_last_error(ctx)
 *     returns a human-readable message for the last failure on that context.
 *   - one context per plugin instance; calls on one context must be serialised by the caller
 *     (plugins are Send, not Sync: ofps/src/plugins/mod.rs:244,261,278); different contexts may
 *     be used concurrently from different host threads.
 *   - the library never frees or retains caller memory; all outputs are caller-allocated.
 *   - "*_dev" entry points take device pointers (hipMalloc'd on the context's device), enqueue
 *     on the context's stream and return without synchronising; host-pointer entry points
 *     copy in/out and synchronise before returning.
 *   - there is no CPU fallback: without a usable gfx950 device ofps_hip_init fails.
 */
#ifndef OFPS_HIP_H
#define OFPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFPS_HIP_API_VERSION 2

enum {
    OFPS_HIP_OK = 0,
    OFPS_HIP_EINVAL = -1,      /* bad argument (message says which) */
    OFPS_HIP_EDEVICE = -2,     /* HIP runtime error / no device */
    OFPS_HIP_EUNSUPPORTED = -3, /* parameter combination has no kernel */
    OFPS_HIP_ENOMEM = -4
};

typedef struct ofps_hip_ctx ofps_hip_ctx;

/* ---- context ---- */
int  ofps_hip_api_version(void);
int  ofps_hip_device_count(void);
int  ofps_hip_init(int device, ofps_hip_ctx** out);
void ofps_hip_destroy(ofps_hip_ctx* ctx);
const char* ofps_hip_last_error(const ofps_hip_ctx* ctx);   /* ctx may be NULL: last init error */
/* Run on a caller-owned hipStream_t (e.g. torch's current stream).  NULL is a valid handle: HIP's
 * default stream.  ofps_hip_use_own_stream() goes back to the stream the context created. */
int   ofps_hip_set_stream(ofps_hip_ctx* ctx, void* hip_stream);
int   ofps_hip_use_own_stream(ofps_hip_ctx* ctx);
void* ofps_hip_get_stream(ofps_hip_ctx* ctx);
int   ofps_hip_sync(ofps_hip_ctx* ctx);
/* Diagnostic / A-B switches (table in INTEGRATION.md).  `name` is the switch's environment-variable name, e.g.
 * "OFPS_HIP_ALMEIDA_HIER"; value NULL or "" restores the default.  ofps_hip_init reads the same variables from the
 * environment ONCE; no other entry point looks at the environment.  The fault injectors
 * (OFPS_HIP_ALMEIDA_TEST_FAULT, OFPS_HIP_LK_TEST_FALL, OFPS_HIP_LK_TEST_WAIT_BUDGET, OFPS_HIP_LK_TEST_ORDER) exist only in libofps_hip_testhooks.so (built with
 * -DOFPS_HIP_TEST_HOOKS, used by the parity tests), can only be armed through this call, and are refused with
 * OFPS_HIP_EUNSUPPORTED by the product library. */
int   ofps_hip_set_option(ofps_hip_ctx* ctx, const char* name, const char* value);
int   ofps_hip_has_test_hooks(void);                         /* 1 in libofps_hip_testhooks.so, 0 in libofps_hip.so */

/* ---- device memory plumbing for hosts without their own HIP binding ---- */
int ofps_hip_malloc(ofps_hip_ctx* ctx, size_t bytes, void** dptr);
int ofps_hip_free(ofps_hip_ctx* ctx, void* dptr);
/* page-locked host memory for frame buffers (H2D by DMA, no staging copy); plain malloc'ed buffers work everywhere too */
int ofps_hip_host_alloc(ofps_hip_ctx* ctx, size_t bytes, void** hptr);
int ofps_hip_host_free(ofps_hip_ctx* ctx, void* hptr);
int ofps_hip_memcpy_h2d(ofps_hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int ofps_hip_memcpy_d2h(ofps_hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- timing helper: HIP events on the context's stream (bench.py roofline leg) ---- */
int ofps_hip_timer_start(ofps_hip_ctx* ctx);
int ofps_hip_timer_stop(ofps_hip_ctx* ctx, float* elapsed_ms);   /* synchronises on the stop event */

/* ---- N1: full-search SAD block matcher ("hip_sad" Decoder) ----
 * Blocks on a block x block lattice from (0,0), full blocks only; candidates (dx,dy) in
 * [-range,range]^2 whose block lies inside the frame; winner = min of
 * (SAD, dx*dx+dy*dy, dy+range, dx+range); entry = (pos = (centre+d)/(W,H), motion = -d/(W,H)).
 * Supported: block in {8,16} with range in {8,12,16,20,24,28,32} and 16-byte aligned rows run on the packed-SAD strip
 * kernel (the same geometries with rows only 4-byte aligned: the per-block packed kernel); any other block <= 64,
 * range <= 64 runs on the generic kernel (correctness path, ~30x slower).  stride % 4 == 0. */
size_t ofps_hip_sad_block_count(int W, int H, int block);
/* Search strategy of ofps_hip_sad_flow*: both return the spec's winner bit for bit.
 * EXHAUSTIVE evaluates every candidate (content-independent run time, the default).
 * PRUNED (partial-distortion elimination) computes the SAD over 4 of the 16 block rows for every candidate -- a lower
 * bound of the full SAD -- and evaluates in full only the candidates whose bound does not exceed the exact SAD of the
 * minimum-bound candidate; run time depends on the content (faster on smooth camera motion with little noise, slower
 * where there is nothing to prune); 16x16 blocks, range 16 only -- other geometries ignore the mode. */
enum { OFPS_HIP_SAD_EXHAUSTIVE = 0, OFPS_HIP_SAD_PRUNED = 1 };
int ofps_hip_set_sad_mode(ofps_hip_ctx* ctx, int mode);
/* Diagnostics: how many strips of the last PRUNED call overflowed their survivor lists and were redone by the
 * exhaustive kernel (synchronises the stream). */
int ofps_hip_sad_pruned_overflow_strips(ofps_hip_ctx* ctx, uint32_t* count);
int ofps_hip_sad_flow(ofps_hip_ctx* ctx, const uint8_t* prev, const uint8_t* cur,
                      int W, int H, int stride, int block, int range,
                      float* out_entries /* 4*nblk */, int32_t* out_best /* 3*nblk (dx,dy,sad) or NULL */,
                      size_t* n_out);
/* Batched, device-resident: n_frames luma frames at d_frames + k*frame_pitch (bytes).
 * ref_mode 0: pairs (k, k+1); ref_mode 1: pairs (0, k+1) (shared key frame).  n_frames-1 pairs.
 * d_out_entries: [(n_frames-1) * nblk * 4] f32; d_out_best: [(n_frames-1) * nblk * 3] i32 or NULL. */
int ofps_hip_sad_flow_dev(ofps_hip_ctx* ctx, const void* d_frames, int n_frames,
                          int W, int H, int stride, size_t frame_pitch, int ref_mode,
                          int block, int range, void* d_out_entries, void* d_out_best);

/* ---- N2: dense per-pixel flow, pyramidal Lucas-Kanade ("hip_lk" Decoder) ----
 * The reference's only per-pixel flow is OpenCV's Farneback inside cv-decoder (cv-decoder/src/lib.rs:188-199); this
 * is a build-defined algorithm (oracle/ofps_oracle.c:orc_lk_flow) with cv-decoder's conventions: prev(x,y) ~
 * cur(x+u,y+v); records pos = ((x+.5)/W,(y+.5)/H), motion = flow/(W,H) in raster order (:239-243,262-269).
 * levels in [1,8] (cfg3: 3), radius in [1,15] (window (2r+1)^2), iters >= 1 Gauss-Newton steps per level.
 * out_flow: 2*W*H f32 (u,v) or NULL; out_entries: 4*W*H f32 or NULL (at least one of them). */
int ofps_hip_lk_flow(ofps_hip_ctx* ctx, const uint8_t* prev, const uint8_t* cur, int W, int H, int stride,
                     int levels, int radius, int iters, float* out_flow, float* out_entries);
/* Revision of the build-defined N2 arithmetic this library was compiled with: 2 = fused multiply-adds in the bilinear
 * sample, the residual sums and the structure tensor (the default), 1 = separate multiply and add (-DOFPS_LK_SPEC_FMA=0,
 * A/B builds).  The oracle exports the same number (orc_lk_spec_revision); the parity tests assert they agree. */
int ofps_hip_lk_spec_revision(void);
/* The flow runs its whole pyramid as ONE launch in which a tile waits for its parent tile of the coarser level to publish its flows.
 * Forward progress does not depend on how the device schedules the launch: a tile whose parent has not published within 0.3 ms computes the
 * missing ancestors itself (a tile's flows are a pure function of the frames: computed twice they are written twice with the same bits), so
 * every workgroup finishes in bounded time in any dispatch order, with any number of resident workgroups, on a CU-masked stream -- and no
 * flow is ever made from an unfinished parent.  Diagnostics (synchronises): how many tiles a waiting child computed since the flag buffer
 * was last allocated -- 0 on a whole device with an in-order dispatcher; a non-zero count costs time, never bits.  OFPS_HIP_LK_SERIAL=1 runs
 * one launch per pyramid level instead (A/B runs). */
int ofps_hip_lk_helped_tiles(ofps_hip_ctx* ctx, uint64_t* count);
/* ---- cv-decoder's frame front-end (cv-decoder/src/lib.rs:98-135): capped grid, resize(INTER_LINEAR), cvt_color(BGR2GRAY) ----
 * OpenCV's 8-bit arithmetic restated integer for integer (oracle/frontend_oracle.c names the sources; "parity unpinned": OpenCV is not vendored
 * by the reference): 11-bit bilinear coefficients with half-pixel centres, the horizontal edge rule, the uchar vertical pass
 * (((b0*(D0>>4))>>16) + ((b1*(D1>>4))>>16) + 2) >> 2, exact 2 x 2 reductions as the area mean; gray = (B*1868 + G*9617 + R*4899 + 8192) >> 14. */
enum { OFPS_HIP_FMT_LUMA = 0, OFPS_HIP_FMT_BGR = 1 /* VideoCapture's frames */, OFPS_HIP_FMT_RGBA = 2 /* ofps::RGBA's byte order */, OFPS_HIP_FMT_BGRA = 3 };
int ofps_hip_frame_channels(int fmt);                                            /* bytes per pixel: 1, 3, 4, 4; 0 = unknown format */
int ofps_hip_cv_grid(int W, int H, int max_w, int max_h, int* gw, int* gh);      /* :98-121 with aspect_ratio_scale (1, 1) */
/* resize keeping the channels: src H rows of W pixels of `fmt`, `stride` bytes apart -> dst dh x dw x channels, dense */
int ofps_hip_resize_linear(ofps_hip_ctx* ctx, const uint8_t* src, int W, int H, int stride, int fmt, uint8_t* dst, int dw, int dh);
int ofps_hip_resize_linear_dev(ofps_hip_ctx* ctx, const void* d_src, int W, int H, int stride, int fmt, void* d_dst, int dw, int dh);
/* what cv-decoder's read loop leaves in `self.gray` for one frame: [reduced != 0: resize to the capped grid ->] gray.  out_gray: *out_w x *out_h
 * dense bytes (capacity W * H is always enough). */
int ofps_hip_cv_frontend(ofps_hip_ctx* ctx, const uint8_t* frame, int W, int H, int stride, int fmt, int reduced, int max_w, int max_h,
                         uint8_t* out_gray, int* out_w, int* out_h);
int ofps_hip_cv_frontend_dev(ofps_hip_ctx* ctx, const void* d_frame, int W, int H, int stride, int fmt, int reduced, int max_w, int max_h,
                             void* d_out_gray, int* out_w, int* out_h);
/* cv-decoder's contrast mask (cv-decoder/src/lib.rs:203-237): Sobel(gray, CV_32F, 1, 1, ksize 5) -> threshold(> 20)
 * -> dilate(MORPH_ELLIPSE 11x11); a pixel contributes a record only where the mask is set (:253-257).  Restated
 * from OpenCV's published definitions (oracle/ofps_oracle.c:orc_contrast_mask; "parity unpinned": OpenCV is not
 * vendored by the reference).  out_mask: W*H bytes, 1 = keep. */
int ofps_hip_contrast_mask(ofps_hip_ctx* ctx, const uint8_t* gray, int W, int H, int stride, uint8_t* out_mask);
int ofps_hip_contrast_mask_dev(ofps_hip_ctx* ctx, const void* d_gray, int W, int H, int stride, void* d_out_mask);

/* ofps_hip_lk_decode / _lk_push_frame / _lk_push_frame_async flags */
#define OFPS_HIP_LK_CONTRAST_MASK 1u /* drop records of pixels outside the contrast mask of `cur` (the reference's
                                        Farneback path always masks, :203-237,253-257) */
#define OFPS_HIP_LK_FULLRES_RECORDS 2u /* an output form of this build, NOT a cv-decoder mode: one record per (unmasked) pixel of the
                                        full-resolution flow in raster order, no down-sampling -- the 2.07 M records of BASELINE
                                        configs[2] that feed ofps_hip_densify_raster_dev / ofps_hip_almeida_dev directly
                                        (API version 1 called this bit OFPS_HIP_LK_PER_PIXEL and mislabelled it "Process Fullres = false") */
#define OFPS_HIP_FLOW_FARNEBACK   4u /* the flow is Farneback's (ofps_hip_farneback_flow: levels = pyramid levels, winsize = 2 * radius + 1,
                                        iters = iterations, poly_n 7, poly_sigma 1.5) instead of the iterative Lucas-Kanade: "hip_flow" */
#define OFPS_HIP_FLOW_USE_PREVIOUS 8u /* with OFPS_HIP_FLOW_FARNEBACK, stream forms: the flow of the stream's previous pair is this pair's initial
                                        flow -- OPTFLOW_USE_INITIAL_FLOW exactly as cv-decoder sets it from its second pair on
                                        (cv-decoder/src/lib.rs:161-165: `self.flow` persists between process_frame calls).  A stream's first
                                        pair (and ofps_hip_lk_decode, a pair on its own) starts from zero flow, like cv-decoder's first. */
#define OFPS_HIP_LK_REDUCED      16u /* cv-decoder's "Process Fullres" = false (cv-decoder/src/lib.rs:124-133,274-276): every frame is resized
                                        (imgproc::resize, INTER_LINEAR) to the (max_w, max_h)-capped grid of :98-121 BEFORE the colour
                                        conversion, mask and flow -- which therefore run on the ~150 x 84 frame -- and every unmasked
                                        pixel of that REDUCED frame is one record at ((x+.5)/gw, (y+.5)/gh) in raster order; no densifier.
                                        Excludes OFPS_HIP_LK_FULLRES_RECORDS. */
/* The frames' pixel format, bits 8-9 of `flags` (one of the OFPS_HIP_FMT_* values below shifted left by 8; 0 = 8-bit luma, this build's
 * raw-stream input).  With a colour format `stride` is the row pitch in BYTES (>= W * channels) and every frame goes through
 * cvt_color(.., COLOR_BGR2GRAY)'s integer formula (cv-decoder/src/lib.rs:135) behind its upload -- after the resize when
 * OFPS_HIP_LK_REDUCED is set, as in the reference. */
#define OFPS_HIP_FRAME_FORMAT(fmt) ((unsigned)(fmt) << 8)
#define OFPS_HIP_FRAME_FORMAT_MASK 0x300u
/* One Decoder::process_frame of a "hip_lk" plugin (cv-decoder/src/lib.rs:82-294): flow -> per-pixel records
 * [-> contrast mask] -> down-sampled through the densifier to the (max_w, max_h)-capped grid of :98-121 (defaults
 * 150 x 150 -> 150 x 84 at 16:9) -> one record per visited cell in (x, y)-sorted order.  out_entries capacity:
 * 4 * min(max_w,W) * min(max_h,H) floats (4 * W * H with OFPS_HIP_LK_FULLRES_RECORDS).  out_w/out_h: the record grid (with
 * OFPS_HIP_LK_REDUCED also the size of the frames the flow ran on). */
int ofps_hip_lk_decode(ofps_hip_ctx* ctx, const uint8_t* prev, const uint8_t* cur, int W, int H, int stride,
                       int levels, int radius, int iters, int max_w, int max_h, unsigned flags,
                       float* out_entries, size_t* n_out, int* out_w, int* out_h);
/* The same for a STREAM of frames (cv-decoder keeps its previous gray frame and starts emitting with the second one,
 * cv-decoder/src/lib.rs:142-158): `frame` is uploaded once and is the next call's previous frame.  *have_vectors = 0 for
 * the first frame of a stream -- after ofps_hip_init, ofps_hip_lk_reset or a change of W/H.  The stream's two frames
 * live in device slots of their own: no other entry point of the context disturbs them. */
int ofps_hip_lk_push_frame(ofps_hip_ctx* ctx, const uint8_t* frame, int W, int H, int stride,
                           int levels, int radius, int iters, int max_w, int max_h, unsigned flags,
                           float* out_entries, size_t* n_out, int* out_w, int* out_h, int* have_vectors);
/* The same iteration split in two for read-ahead callers (cv-decoder's loop, cv-decoder/src/lib.rs:82-158, run one frame
 * ahead on a decoder thread as ofps-suite/src/app/tracking/worker.rs:165-226 does): push_frame_async returns once the frame's
 * H2D copy (on a copy stream when another ticket is in flight), the flow of (previous frame, this frame) and the output
 * stage are enqueued; the stage's last kernel writes the records and their count straight into the ticket's page-locked
 * block.  ofps_hip_lk_frame_wait(ticket) blocks until they are there and copies them to out_entries.  Up to 2 tickets in
 * flight: the upload of frame k+1 overlaps the flow of pair (k-1, k), and the host collects pair k-1's records meanwhile.
 * `frame` must stay valid until its ticket is collected (page-locked memory: ofps_hip_host_alloc; pageable memory makes the
 * copy synchronous).  ofps_hip_lk_push_frame == push_frame_async + frame_wait, bit for bit; both forms share one stream
 * of frames.  out_entries capacity as for ofps_hip_lk_decode. */
int ofps_hip_lk_push_frame_async(ofps_hip_ctx* ctx, const uint8_t* frame, int W, int H, int stride,
                                 int levels, int radius, int iters, int max_w, int max_h, unsigned flags, int* ticket);
int ofps_hip_lk_frame_wait(ofps_hip_ctx* ctx, int ticket, float* out_entries, size_t* n_out, int* out_w, int* out_h,
                           int* have_vectors);
int ofps_hip_lk_reset(ofps_hip_ctx* ctx);                  /* waits for tickets in flight; the next frame starts a new stream (zero initial flow) */
/* The same, but the stream's last flow stays the next pair's initial flow (OFPS_HIP_FLOW_USE_PREVIOUS): what a decoder calls when it skipped
 * frames and pushes the pair's first frame again -- cv-decoder's `self.flow` persists across skipped reads (cv-decoder/src/lib.rs:92-142,161-165). */
int ofps_hip_lk_rewind(ofps_hip_ctx* ctx);
int ofps_hip_lk_flow_dev(ofps_hip_ctx* ctx, const void* d_prev, const void* d_cur, int W, int H, int stride,
                         int levels, int radius, int iters, void* d_out_flow, void* d_out_entries);
/* The same with a starting flow for the COARSEST pyramid level (d_init_flow: 2 f32 per pixel of that level, whose size is
 * W, H halved -- rounding up -- levels - 1 times; NULL = zero = ofps_hip_lk_flow_dev): a temporal prior, e.g. the previous
 * pair's flow reduced to that level -- the role of OPTFLOW_USE_INITIAL_FLOW in the call cv-decoder makes with flags 0
 * (cv-decoder/src/lib.rs:188-199).  Spec: oracle/ofps_oracle.c:orc_lk_flow_init. */
int ofps_hip_lk_flow_init_dev(ofps_hip_ctx* ctx, const void* d_prev, const void* d_cur, int W, int H, int stride,
                              int levels, int radius, int iters, const void* d_init_flow, void* d_out_flow,
                              void* d_out_entries);

/* ---- N2 in the reference's own algorithm family: Farneback's polynomial-expansion flow as cv-decoder calls it
 * (cv-decoder/src/lib.rs:188-199: pyr_scale 0.5, levels 5, winsize 13, iterations 3, poly_n 7, poly_sigma 1.5; pyr_scale is fixed at 0.5).
 * The arithmetic the reference runs is OpenCV's (not part of the reference tree: PARITY UNPINNED); this is the published algorithm in the
 * form calcOpticalFlowFarneback gives it (flags = 0: box window), precision per stage as in OpenCV's CPU path -- DESIGN.md "N2b".
 * out_flow: 2*W*H f32 (dx, dy) per pixel of `prev` (prev(x,y) ~ cur(x+dx, y+dy)) or NULL; out_entries: 4*W*H f32 records or NULL (at least
 * one).  init_flow: NULL, or a 2*W*H flow to start from (OPTFLOW_USE_INITIAL_FLOW: cv-decoder passes its previous result).
 * winsize odd <= 15, poly_n <= 15, at most 6 + 1 pyramid layers of blur taps (levels <= 6 at any size): else OFPS_HIP_EUNSUPPORTED.
 * OFPS_HIP_FLOW_FARNEBACK in the `flags` of ofps_hip_lk_decode / _lk_push_frame / _lk_push_frame_async selects this flow for the
 * decoder entry points (levels = pyramid levels, winsize = 2 * radius + 1, iters = iterations; poly_n 7, poly_sigma 1.5). */
int ofps_hip_farneback_flow(ofps_hip_ctx* ctx, const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int winsize,
                            int iters, int poly_n, float poly_sigma, const float* init_flow, float* out_flow, float* out_entries);
int ofps_hip_farneback_flow_dev(ofps_hip_ctx* ctx, const void* d_prev, const void* d_cur, int W, int H, int stride, int levels, int winsize,
                                int iters, int poly_n, float poly_sigma, const void* d_init_flow, void* d_out_flow, void* d_out_entries);
/* In the stream forms (ofps_hip_lk_push_frame[_async] with OFPS_HIP_FLOW_FARNEBACK) a pair's second frame is the next pair's first: its
 * pyramid and polynomial expansion are kept on the device and only the new frame goes through them (same results, bit for bit).
 * Diagnostics: how many calls of this context found the first frame's expansion already there. */
int ofps_hip_flow_cache_hits(ofps_hip_ctx* ctx, uint64_t* count);
/* ---- A1-A4: MotionFieldDensifier ---- */
int ofps_hip_densify(ofps_hip_ctx* ctx, const float* entries, size_t n, int w, int h,
                     float* out_field /* 2*w*h, cell (x,y) at 2*(y*w+x) */,
                     uint32_t* out_cells /* 2*n (x,y) per entry, or NULL */);
/* add_vector_weighted (ofps/src/motion_field.rs:164-178): entry i is inserted with weights[i]: counts += w,
 * sum = motion * w + sum, in input order; ofps_hip_densify is the weights == 1 case (add_vector, :188-190). */
int ofps_hip_densify_weighted(ofps_hip_ctx* ctx, const float* entries, const float* weights, size_t n, int w, int h,
                              float* out_field /* 2*w*h */, uint32_t* out_cells /* 2*n or NULL */);
/* batch items of n_per_item entries each (contiguous); outputs per item. */
int ofps_hip_densify_dev(ofps_hip_ctx* ctx, const void* d_entries, size_t n_per_item, int batch,
                         int w, int h, void* d_out_field, void* d_out_cells /* or NULL */);
/* The same fields for a PER-PIXEL producer (cv-decoder/src/lib.rs:239-291): d_entries holds W*H records in raster order
 * whose positions are ((x+.5)/W, (y+.5)/H) -- what ofps_hip_lk_flow_dev writes -- and d_mask (W*H bytes, or NULL) selects
 * the records that are inserted (cv-decoder's contrast mask, :251-276).  A cell's records are then a rectangle of pixels
 * in raster order, which is their input order: one launch, no sort, the same bits as ofps_hip_densify_dev on the
 * (compacted) records.  verify != 0 checks the precondition on the device first (blocking) and fails with
 * OFPS_HIP_EINVAL when a record is not where the lattice puts it; without it a violated precondition gives an
 * unspecified field. */
int ofps_hip_densify_raster_dev(ofps_hip_ctx* ctx, const void* d_entries, const void* d_mask /* or NULL */, int W, int H,
                                int w, int h, void* d_out_field /* 2*w*h floats */, int verify);
int ofps_hip_densify_to_entries(ofps_hip_ctx* ctx, const float* entries, size_t n, int w, int h,
                                float* out_entries /* capacity 4*w*h */, size_t* n_out);

/* new_densifier + add_vector (all entries) + interpolate_empty_cells + from_densifier: the sequence of
 * flow-extract/src/main.rs:74-83 (ofps/src/motion_field.rs:193-294).  Sequential by definition. */
int ofps_hip_densify_interpolated(ofps_hip_ctx* ctx, const float* entries, size_t n, int w, int h, float* out_field);

/* ---- A5: BlockMotionDetection::detect_motion ("hip_block_motion" Detector) ---- */
int ofps_hip_block_dim(float min_size, size_t subdivide);
int ofps_hip_detect(ofps_hip_ctx* ctx, const float* entries, size_t n,
                    float min_size, size_t subdivide, float target_motion,
                    int* has_motion, size_t* area, int* dim, float* out_field /* 2*dim*dim */);
/* per item result record: int32[4] = {has_motion, area, dim, 0}; field 2*dim*dim f32 per item. */
int ofps_hip_detect_dev(ofps_hip_ctx* ctx, const void* d_entries, size_t n_per_item, int batch,
                        float min_size, size_t subdivide, float target_motion,
                        void* d_out_result, void* d_out_field);

/* ---- A6-A12: Almeida estimator ("hip_almeida" Estimator) ----
 * out_quat = (w,i,j,k) of the returned UnitQuaternion; out_tr = translation (always 0,
 * almeida-estimator/src/lib.rs:120).  seed drives the counter-based RANSAC sampler (the reference draws from
 * thread_rng): callers should advance it per call -- a constant seed samples the same positions every frame -- and
 * in a batched call item b uses seed + b.  Fields of more than 65,536 vectors (per-pixel records) are solved with
 * reciprocal-multiply quotients instead of IEEE division and fused multiply-adds (<= 1 ulp per operation; quaternion
 * within 2e-6 of the exact path).  Like the reference's estimator (singular system -> zero step,
 * almeida-estimator/src/lib.rs:181-185; < 3 inliers -> identity, :246-250) no entry point ever returns NaN: a cluster
 * launch whose workgroups were not all resident (another process holding CUs) finishes the solve inside the same
 * launch after its bounded wait (~0.3 s), on every entry point incl. the device-pointer and per-frame ones;
 * ofps_hip_almeida_recoveries counts how often that happened on this context (diagnostics; synchronises). */
int ofps_hip_almeida(ofps_hip_ctx* ctx, const float* entries, size_t n,
                     float aspect, float fov_y_deg, int use_ransac, size_t num_iters,
                     float inlier_deg, size_t num_samples, uint64_t seed,
                     float out_quat[4], float out_tr[3]);
int ofps_hip_almeida_dev(ofps_hip_ctx* ctx, const void* d_entries, size_t n_per_item, int batch,
                         float aspect, float fov_y_deg, int use_ransac, size_t num_iters,
                         float inlier_deg, size_t num_samples, uint64_t seed,
                         void* d_out_quat /* 4 f32 per item */);
int ofps_hip_almeida_recoveries(ofps_hip_ctx* ctx, uint64_t* count);

/* ---- fused per-frame path (live streams): decoder -> detector + estimator, vectors stay on the device ----
 * One call per arriving luma frame = one iteration of the reference's worker loops
 * (ofps-suite/src/app/detection.rs:111-148, tracking/worker.rs:328-361).  The context keeps the previous
 * frames on the device (a ring of three slots); the first frame after ofps_hip_init / ofps_hip_reset_frames / a geometry change
 * yields have_vectors = 0 (Decoder::process_frame -> Ok(false)). */
typedef struct {
    int block, range;                                   /* hip_sad */
    int run_detector;                                   /* hip_block_motion */
    float min_size; size_t subdivide; float target_motion;
    int run_estimator;                                  /* hip_almeida */
    float aspect, fov_y_deg; int use_ransac; size_t num_iters; float inlier_deg; size_t num_samples; uint64_t seed;
} ofps_hip_frame_params;
typedef struct {
    int have_vectors; size_t n_vectors;
    int has_motion; size_t area; int dim;               /* detector: Some((area, field dim x dim)) / None */
    float quat[4];                                      /* estimator: (w,i,j,k); identity when not run */
} ofps_hip_frame_result;
int ofps_hip_reset_frames(ofps_hip_ctx* ctx);
/* upload a frame as the stream's newest frame without computing anything: the frames `Decoder::process_frame` reads
 * past when called with `skip_frames > 0` (ofps/src/decoder.rs:47-60; cv-decoder/src/lib.rs:92-142 loops `cnt <= skip`) */
int ofps_hip_stage_frame(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride);
int ofps_hip_push_frame(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride,
                        const ofps_hip_frame_params* params, ofps_hip_frame_result* out,
                        float* out_entries /* 4*nblk or NULL */, float* out_field /* 2*dim*dim or NULL */);
/* The same iteration split in two for read-ahead callers (the reference decodes on its own thread into a double
 * buffer, ofps-suite/src/app/tracking/worker.rs:165-226): push_frame_async returns once the frame's H2D copy (on a
 * copy stream), its search and its tail are enqueued; ofps_hip_frame_wait(ticket) blocks until that frame's results
 * are on the host and fills `out`.  Up to 2 tickets may be in flight, so the upload of frame k+1 overlaps the search of
 * pair (k-1, k); tickets must be collected in order before a third push.  `luma`, `out_entries` and `out_field` must
 * stay valid until the wait returns; use ofps_hip_host_alloc'ed (page-locked) buffers -- pageable memory turns the
 * copies synchronous.  ofps_hip_push_frame == push_frame_async + frame_wait, bit for bit. */
int ofps_hip_push_frame_async(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride,
                              const ofps_hip_frame_params* params, float* out_entries /* 4*nblk or NULL */,
                              float* out_field /* 2*dim*dim or NULL */, int* ticket);
int ofps_hip_frame_wait(ofps_hip_ctx* ctx, int ticket, ofps_hip_frame_result* out);
/* Batched read-ahead form: n consecutive frames of a stream per ticket (a decoder running n frames ahead).  `frames` holds
 * them frame_pitch bytes apart; ONE upload, one search launch over the batch's pairs, one detector chain and one
 * estimator launch over the batch, one read-back: a handful of HIP calls per batch instead of ~9 per frame.  Frame j is
 * paired with the stream's previous frame (the last frame of the previous batch for j = 0; the very first frame of a
 * stream yields have_vectors = 0).  Vectors and detector results equal n single pushes bit for bit; the quaternions agree
 * to the solver's parity bound (2e-6: a batch is solved by one launch over its items, a lone frame by the cluster solver --
 * two fixed summation orders), frame j's RANSAC seed = params->seed + j.  out_entries (n * nblk * 4 floats, or NULL) receives every frame's vectors at j * nblk * 4.  Up to 2 batches in
 * flight; `frames` / `out_entries` must stay valid until ofps_hip_frames_wait(ticket) returns, which fills out[0..n-1].
 * The batched stream is separate from the single-frame calls' (its own previous frame); ofps_hip_reset_frames resets both. */
int ofps_hip_push_frames_async(ofps_hip_ctx* ctx, const uint8_t* frames, int n, int W, int H, int stride, size_t frame_pitch,
                               const ofps_hip_frame_params* params, float* out_entries /* n*4*nblk or NULL */, int* ticket);
int ofps_hip_frames_wait(ofps_hip_ctx* ctx, int ticket, ofps_hip_frame_result* out /* n entries */);

/* ---- one host process, several GPUs: frame pairs are independent units, so a batch is split into
 * contiguous pair ranges, one worker thread + one context + one stream per entry of `devices` (an entry may repeat: the
 * workers are then independent contexts on one GPU).  No collective on the data path; in key mode (ref_mode 1: pair k =
 * frames 0, k+1) the key frame is uploaded once and fanned out device to device (hipMemcpyPeerAsync, xGMI point to
 * point).  Results come back in pair order.  The reference's counterpart is its worker-thread model
 * (ofps-suite/src/app/tracking/worker.rs:251-260,347-352); it has no multi-GPU code of its own. ---- */
/* Per-item checksum of device-resident data (records, fields): d_out_u64[item] = wrapping sum of the item's bytes read
 * as u64 words -- what a host gathers across GPUs instead of the records when it only needs to confirm them.  Enqueues on
 * the context's stream; bytes_per_item % 8 == 0. */
int ofps_hip_checksum_dev(ofps_hip_ctx* ctx, const void* d_data, size_t bytes_per_item, int batch, void* d_out_u64);
typedef struct ofps_hip_multi ofps_hip_multi;
int  ofps_hip_multi_init(const int* devices, int n, ofps_hip_multi** out);
void ofps_hip_multi_destroy(ofps_hip_multi* m);
const char* ofps_hip_multi_last_error(const ofps_hip_multi* m);       /* m may be NULL: last init error */
int  ofps_hip_multi_worker_count(const ofps_hip_multi* m);
/* How the shared key frame of ref_mode 1 reaches the workers' devices: 0 = hipMemcpyPeerAsync from the first worker's copy (the default),
 * 1 = ONE ncclBroadcast over an RCCL communicator of the workers' devices (xGMI) -- chosen at ofps_hip_multi_init when the environment
 * has OFPS_HIP_MULTI_RCCL=1, the devices are distinct and librccl.so can be dlopen'ed (the library does not link it).  *broadcasts (may
 * be NULL): key frames that went through ncclBroadcast so far. */
int  ofps_hip_multi_fanout(const ofps_hip_multi* m, uint64_t* broadcasts);
/* The partition, as pure functions (no device needed): worker k of n gets pairs [first, first + count) -- the first
 * n_pairs % n workers one more -- and keeps frames [first_frame, first_frame + n_frames) resident: count + 1 frames in
 * pair mode (one halo frame shared with the next worker), count frames in key mode (frame 0 arrives by the fan-out). */
void ofps_hip_multi_pair_range(size_t n_pairs, int n_workers, int k, size_t* first, size_t* count);
void ofps_hip_multi_frame_range(size_t n_pairs, int n_workers, int k, int ref_mode, size_t* first_frame, size_t* n_frames);
/* Host frames in (n_frames luma frames at frames + k * frame_pitch), vectors out in pair order:
 * out_entries [(n_frames - 1) * nblk * 4] f32.  = stage_frames + run_resident(1 step) + fetch. */
int ofps_hip_multi_sad_flow(ofps_hip_multi* m, const uint8_t* frames, int n_frames, int W, int H, int stride,
                            size_t frame_pitch, int ref_mode, int block, int range, float* out_entries);
/* The same in three parts, for callers that search a resident batch repeatedly (bench.py --launcher threads): upload and
 * partition; `steps` searches of every worker's resident pairs back to back (returns when all workers are through); copy
 * the last results back in pair order. */
int ofps_hip_multi_stage_frames(ofps_hip_multi* m, const uint8_t* frames, int n_frames, int W, int H, int stride,
                                size_t frame_pitch, int ref_mode);
int ofps_hip_multi_run_resident(ofps_hip_multi* m, int block, int range, int steps,
                                float* worker_ms /* NULL, or one entry per worker: HIP-event time of its `steps` launches */);
int ofps_hip_multi_fetch(ofps_hip_multi* m, int block, float* out_entries);     /* results of the LAST run of the staged batch, same block size */
/* The dispatcher's entry points are serialised among themselves (one lock per handle): several host threads may share a handle,
 * they just do not overlap inside it (ofps_hip_multi_frames_wait blocks outside the lock). */

/* A STREAM of frames over several devices: the multi-device form of ofps_hip_push_frames_async / ofps_hip_frames_wait (same
 * parameters, same results per frame).  Batch g of n consecutive frames goes to worker g % n_workers together with the one frame
 * in front of it (kept by the dispatcher in page-locked memory), through that worker's two-ticket batched read-ahead: upload on
 * a copy stream, one search launch, detector and estimator over the batch, results read back by kernel.  Up to 2 * n_workers
 * batches in flight; ofps_hip_multi_frames_wait(ticket) fills out[0..n-1] -- frame order is batch order.  `frames` and
 * `out_entries` must stay valid until the wait returns; batches in pageable memory are copied into a worker's page-locked
 * staging on that worker's thread.  Vectors and detector results equal the single-context stream's bit for bit, quaternions
 * too for equal batch sizes (one launch per batch either way).  The reference's counterpart: the decoder thread + double
 * buffer and the estimator fan-out of ofps-suite/src/app/tracking/worker.rs:165-226,347-361. */
int ofps_hip_multi_push_frames_async(ofps_hip_multi* m, const uint8_t* frames, int n, int W, int H, int stride, size_t frame_pitch,
                                     const ofps_hip_frame_params* params, float* out_entries /* n*4*nblk or NULL */, int* ticket);
int ofps_hip_multi_frames_wait(ofps_hip_multi* m, int ticket, ofps_hip_frame_result* out /* n entries */);
int ofps_hip_multi_reset_frames(ofps_hip_multi* m);
/* The dealing, as a pure function (no device needed): batch g of a stream -> the worker that takes it, the halo buffer that
 * holds the frame in front of it, and the ticket slot it occupies (at most two batches in flight per worker). */
void ofps_hip_multi_stream_plan(long batch, int n_workers, int* worker, int* halo_slot, int* ticket_slot);

#ifdef __cplusplus
}
#endif
#endif
