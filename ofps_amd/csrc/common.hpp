// common.hpp -- context object and error plumbing shared by the libofps_hip.so translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/ofps_hip.h"

struct ofps_hip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;        // the stream work is enqueued on (own or caller's)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    int num_cus = 0;
    int stream_cus = 0;                  // compute units `stream` may use (hipExtStreamGetCUMask; a caller's stream may carry a CU mask)
    int sad_mode = OFPS_HIP_SAD_EXHAUSTIVE;
    char err[512] = {0};

    // Diagnostic / A-B switches (INTEGRATION.md lists them).  Read from the environment ONCE, in ofps_hip_init; a live
    // context changes them through ofps_hip_set_option.  No entry point reads the environment after that.
    struct Options {
        int sad_force_block = 0;         // OFPS_HIP_SAD_KERNEL=block
        int densify_no_small = 0;        // OFPS_HIP_DENSIFY_NO_SMALL
        int almeida_path = 0;            // OFPS_HIP_ALMEIDA_PATH: 0 default, 1 step, 2 wg, 3 cluster
        int almeida_ept = 0;             // OFPS_HIP_ALMEIDA_EPT: 0 = cost model, else 1/2/4/8
        int almeida_block = 0;           // OFPS_HIP_ALMEIDA_BLOCK: 0 = 1024, else 256/1024
        int almeida_hier = 1;            // OFPS_HIP_ALMEIDA_HIER: 0 never, 1 when it pays, 2 always
        int almeida_fast = -1;           // OFPS_HIP_ALMEIDA_FAST: -1 by size, 0 exact, 1 folded
        int almeida_one_xcd = 1;         // OFPS_HIP_ALMEIDA_ONE_XCD: 0 never, 1 small clusters run on one XCD and exchange through its L2 (almeida.hip)
        int almeida_prof = 0;            // OFPS_HIP_ALMEIDA_PROF
        int lk_prof = 0;                 // OFPS_HIP_LK_PROF
        int fb_prepare_ahead = 1;        // OFPS_HIP_FB_PREPARE_AHEAD: a hip_flow stream's new frame is expanded on the upload's stream when it is pushed (0: inside the pair's flow, round 5's order)
        int lk_serial = 0;               // OFPS_HIP_LK_SERIAL: one launch per pyramid level instead of one for the pyramid
        int multi_rccl = 0;              // OFPS_HIP_MULTI_RCCL: ofps_hip_multi_init fans the shared key frame out by ncclBroadcast (multi.hip)
        // fault injectors: only builds with -DOFPS_HIP_TEST_HOOKS (libofps_hip_testhooks.so) can set them, and only
        // through ofps_hip_set_option -- never from the environment
        int test_almeida_fault = 0;      // OFPS_HIP_ALMEIDA_TEST_FAULT: workgroup (value - 1) withholds its step-3 granule
        int test_lk_fall = -1;           // OFPS_HIP_LK_TEST_FALL: every other tile hands over at this step
        int test_lk_wait_budget = 0;     // OFPS_HIP_LK_TEST_WAIT_BUDGET: 100 MHz ticks a tile waits for its parent's flag before it computes its ancestors itself (0 = the product's budget)
        int test_lk_order = 0;           // OFPS_HIP_LK_TEST_ORDER: the one-launch pyramid's blocks take their positions 1: in reverse, 2: permuted, after random delays
    } opt;

    // hip_lk stream state (lk.hip: ofps_hip_lk_push_frame): frame k of the stream lives in slot k % 2 of S_FRAMES
    int lk_w = 0, lk_h = 0;              // the arriving frames' size
    int lk_fw = 0, lk_fh = 0, lk_fmt = 0; // the size of the frames in the ring (reduced with OFPS_HIP_LK_REDUCED) and the arriving frames' format
    long lk_frames = 0;
    bool lk_fb_params_valid = false; int lk_fb_levels = 0, lk_fb_radius = 0;     // the hip_flow stream's last Farneback parameters (a change with a ticket in flight is refused)
    uint64_t lk_frames_gen = 0;          // generation of the S_LK_FRAMES allocation the count refers to
    uint32_t lk_epoch = 0;               // lk_levels_kernel: tag of the last launch in the tile flags (S_LK_FLAGS)
    uint64_t lk_flags_gen = 0;           // generation of the flag buffer the tags refer to
    void* lk_pinned = nullptr;           // page-locked staging for a frame's records + their count (one D2H, one wait)
    size_t lk_pinned_cap = 0;
    // read-ahead form (ofps_hip_lk_push_frame_async / ofps_hip_lk_frame_wait): a ring of three device frame slots, the new frame
    // uploaded on a copy stream while the previous pair's flow runs, two tickets in flight, each with a page-locked block
    // [count, pad x 3][records] that the last kernels of the ticket write directly
    static constexpr int kLkSlots = 3, kLkTickets = 2;
    hipStream_t lk_copy_stream = nullptr;
    struct LkTicket {
        bool pending = false;
        hipEvent_t done = nullptr, uploaded = nullptr;
        void* pinned = nullptr; size_t pinned_cap = 0;
        int have_vectors = 0, gw = 0, gh = 0;
        size_t max_records = 0;
        long fixed_count = -1;           // >= 0: the record count is known on the host (per-pixel output without a mask)
    } lk_ticket[kLkTickets];
    long lk_next_ticket = 0;
    // hip_flow in the stream forms (farneback.hip): the pyramid + polynomial expansion of a pair's second frame is the next pair's
    // first, and the NEW frame's is made on the upload's stream beside the previous pair's flow.  Frames of the stream carry ids (never
    // reused); `fb_cache.id[i]` is the frame whose expansion planes are in R slot i of the S_FB_WORK allocation of generation `gen`, made
    // with these parameters.
    uint64_t lk_frame_serial = 0;
    uint64_t lk_slot_id[kLkSlots] = {0, 0, 0};
    static constexpr int kFbSlots = 3;   // expansion-plane slots per layer: the frames of two pairs in flight (k - 1, k, k + 1)
    struct FbCache {
        uint64_t id[kFbSlots] = {0, 0, 0};   // the stream frame whose planes a slot holds (0 = none)
        int W = 0, H = 0, K = 0, poly_n = 0;
        double poly_sigma = 0;
        uint64_t gen = 0;
    } fb_cache;
    hipEvent_t fb_prep_done = nullptr;   // the last pyramid + expansion of this context (its T / I temporaries are shared; prepares may run on two streams)
    bool fb_prep_recorded = false;       // a prepare has been enqueued (on fb_prep_stream) since the context was made
    hipStream_t fb_prep_stream = nullptr;
    hipStream_t fb_prep_synced = nullptr; // a stream that has been ordered behind the latest prepare by its caller (the stream forms' `uploaded` event)
    uint64_t fb_cache_hits = 0;          // (tests: how many calls skipped the first frame's pyramid + expansion)
    struct FbPrevFlow { bool valid = false; int W = 0, H = 0; uint64_t id = 0, gen = 0; } fb_prev_flow;     // S_FB_FLOW holds the flow of the pair whose second frame has this id

    // per-frame pipeline state (pipeline.hip): a ring of three device frame slots (the new frame is uploaded on the copy
    // stream while the previous pair is still being searched), two tickets in flight
    static constexpr int kPipeSlots = 3, kPipeTickets = 2;
    int pipe_w = 0, pipe_h = 0, pipe_stride = 0;
    long pipe_frames = 0;                // frames pushed since the last reset; frame k lives in slot k % 3
    hipStream_t pipe_copy_stream = nullptr;
    hipStream_t pipe_aux_stream = nullptr;       // the detector runs here beside the estimator (both read the same vectors)
    hipEvent_t pipe_fork = nullptr, pipe_join = nullptr;
    hipEvent_t pipe_uploaded[kPipeSlots] = {};   // H2D of the frame in this slot finished (recorded on the copy stream)
    hipEvent_t pipe_slot_read[kPipeSlots] = {};  // last search that reads this slot finished (recorded on the compute stream)
    bool pipe_slot_read_valid[kPipeSlots] = {};
    bool pipe_uploaded_on_compute[kPipeSlots] = {};   // the upload was enqueued on the compute stream (ordered by it)
    struct PipeTicket {
        bool pending = false;            // pushed, not yet collected by ofps_hip_frame_wait
        hipEvent_t done = nullptr;       // everything of this ticket, read-backs included
        void* pinned = nullptr;          // page-locked PipeOut block for the result read-back
        int have_vectors = 0, run_detector = 0, run_estimator = 0;
        size_t n_vectors = 0;
    } pipe_ticket[kPipeTickets];
    long pipe_next_ticket = 0;

    // batched form of the same pipeline (ofps_hip_push_frames_async): n frames per ticket, ONE upload, ONE search launch
    // over the batch's pairs, ONE read-back.  Two batch buffers of (capacity + 1) frames alternate: slot 0 holds the last
    // frame of the previous batch (the first pair's previous frame).  Its stream of frames is separate from the
    // single-frame calls'.
    static constexpr int kBatchTickets = 2;
    struct BatchTicket {
        bool pending = false;
        hipEvent_t done = nullptr, uploaded = nullptr, prev_copied = nullptr;
        bool prev_copied_valid = false;
        void* pinned = nullptr; size_t pinned_cap = 0;       // n x {result[4], quat[4]}
        int n = 0, first_has_prev = 0, run_detector = 0, run_estimator = 0;
        size_t n_vectors = 0;
    } batch_ticket[kBatchTickets];
    long batch_next_ticket = 0;
    long batch_frames = 0;               // frames pushed through the batched form since the last reset
    int batch_w = 0, batch_h = 0;
    void* batch_last_frame = nullptr;    // device address of the newest frame of the batched stream

    // cluster Almeida solver (almeida.hip): granule exchange buffer state.  Tags are unique per call (tag base advances
    // by 32 per launch), so the buffer is zeroed only when (re)allocated or when the 32-bit tag space wraps.
    uint32_t gran_tag_base = 0;
    uint64_t gran_zeroed_gen = 0;        // generation (Scratch::gen) of the allocation the zeroing was done for

    // grow-only device scratch owned by the context (staging for host-pointer entry points and
    // kernel workspaces); never shrinks, freed in ofps_hip_destroy.
    struct Scratch { void* p = nullptr; size_t cap = 0; uint64_t gen = 0; };   // gen: bumped by every (re)allocation of the slot
    static constexpr int kNumScratch = 40;
    Scratch scratch[kNumScratch];
};

namespace ofps {

enum ScratchSlot {
    S_FRAMES = 0, S_ENTRIES, S_BEST, S_FIELD, S_CELLS, S_WORK0, S_WORK1, S_WORK2, S_WORK3, S_RESULT,
    S_QUAT, S_WORK4, S_PIPE_FRAMES, S_PIPE_ENTRIES, S_PIPE_OUT, S_MASK, S_ENTRIES2, S_GRAN, S_SAD_LIST,
    // the estimator's own workspaces: it may run beside the detector (pipeline.hip), so the two share no slot
    S_ALM_PART, S_ALM_STATE, S_ALM_HYP, S_ALM_COUNTS, S_ALM_SEL, S_ALM_SELN, S_ALM_PROF, S_ALM_RECOVER,
    S_LK_FRAMES, S_BATCH_FRAMES, S_BATCH_ENTRIES, S_BATCH_OUT, S_BATCH_FIELD,
    // the one-launch LK pyramid's tile flags + helped-tile counter: they carry state ACROSS calls (epoch tags, never cleared), so the slot
    // is nobody else's (round 4: they sat in S_WORK3, which the densifier's per-cell tables also use -- a decode call wiped the counter,
    // and a begin[] value equal to a later launch's epoch would have read as "parent tile done")
    S_LK_FLAGS,
    S_FB_WORK,              // farneback.hip: blur / image / expansion / flow planes of one pair
    S_FB_FLOW,              // hip_flow streams with OFPS_HIP_FLOW_USE_PREVIOUS: the last pair's flow (the next pair's initial flow)
    S_XMAJOR,               // densify.hip, raster producers: the field + visited flag in (x, y)-sorted cell order (the record order)
    S_LK_MASKS,             // dense decoders, stream forms: one contrast mask per ticket in flight (made on the upload's stream, beside the previous pair's flow)
    S_FE_RAW,               // frontend.hip: the frames as they arrive (colour and / or full size) when the decoder resizes / converts them: one per ticket in flight
    S_FE_RAW_PAIR           // ... of the stateless calls (ofps_hip_lk_decode, ofps_hip_cv_frontend, ofps_hip_resize_linear): never the stream's staging, whose
                            // upload + front-end may still be running on the upload stream when such a call comes in
};
static_assert(S_FE_RAW_PAIR < ofps_hip_ctx::kNumScratch, "scratch table too small");

// Page-locked blocks that kernels write directly and the host reads after an event (ticket result blocks, ofps_hip_host_alloc):
// fine-grained host memory, asked for explicitly.  A/B builds (tools/read_ahead_bisect.sh) override the two constants with -D.
#ifndef OFPS_HIP_HOST_BLOCK_FLAGS
#define OFPS_HIP_HOST_BLOCK_FLAGS hipHostMallocCoherent
#endif
#ifndef OFPS_HIP_HOST_USER_FLAGS
#define OFPS_HIP_HOST_USER_FLAGS hipHostMallocCoherent
#endif
#ifndef OFPS_HIP_FRAME_WAIT_SPIN_US
#define OFPS_HIP_FRAME_WAIT_SPIN_US 500
#endif

int set_error(ofps_hip_ctx* ctx, int code, const char* fmt, ...);
int check_hip(ofps_hip_ctx* ctx, hipError_t e, const char* what);
// returns nullptr (and sets the error) on failure
void* scratch(ofps_hip_ctx* ctx, int slot, size_t bytes);

// ---- device-side stage entry points shared between translation units (all enqueue on ctx->stream)
int sad_pairs_device(ofps_hip_ctx* ctx, const uint8_t* prev_base, size_t prev_pitch, const uint8_t* cur_base,
                     size_t cur_pitch, int pairs, int W, int H, int stride, int block, int range, void* d_out_entries,
                     void* d_out_best);
int densify_device(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int batch, int w, int h, float2* d_field,
                   uint32_t* d_cells, uint32_t** out_begin, uint32_t** out_end);
int densify_device_raw(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int batch, int w, int h, float2* d_field,
                       uint32_t* d_cells, uint32_t** out_begin, uint32_t** out_end, float2* d_sum, float* d_cnt,
                       const float* d_weights);
int densify_entries_device(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int w, int h, float2* d_field,
                           float4* d_out_entries, uint32_t* d_count);
int densify_raster_device(ofps_hip_ctx* ctx, const float4* d_entries, const uint8_t* d_mask, int W, int H, int w, int h,
                          float2* d_field, uint32_t** out_begin, uint32_t** out_end, float4* d_xmajor = nullptr);
int densify_raster_entries_device(ofps_hip_ctx* ctx, const float4* d_entries, const uint8_t* d_mask, int W, int H, int w, int h,
                                  float2* d_field, float4* d_out_entries, uint32_t* d_count);
// frontend.hip: cv-decoder's capped grid (cv-decoder/src/lib.rs:98-121) and its per-frame [resize ->] gray step (:124-135)
int frame_format_channels(int fmt);
void cv_grid(int W, int H, int max_w, int max_h, int* gw, int* gh);
int frontend_device(ofps_hip_ctx* ctx, const uint8_t* d_src, int W, int H, int stride, int fmt, bool to_gray, uint8_t* d_dst, int dw, int dh,
                    hipStream_t st = nullptr);
int contrast_mask_device(ofps_hip_ctx* ctx, const uint8_t* d_gray, int W, int H, int stride, uint8_t* d_mask, hipStream_t st = nullptr);   // st: nullptr = ctx->stream
int compact_entries_device(ofps_hip_ctx* ctx, const float4* d_in, const uint8_t* d_mask, size_t n, float4* d_out,
                           uint32_t* d_count);
constexpr size_t kCompactSmallMax = 32768;          // record sets up to this size are compacted by one workgroup, straight to their destination (mask.hip)
int compact_small_device(ofps_hip_ctx* ctx, const float4* d_in, const uint8_t* d_mask, size_t n, float4* d_out, uint32_t* d_count);
int detect_device(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int batch, float min_size, size_t subdivide,
                  float target_motion, int* d_result, float2* d_out_field, int* out_dim);
int farneback_flow_device(ofps_hip_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_cur, int W, int H, int stride, int levels, int winsize,
                          int iters, int poly_n, double poly_sigma, const float2* d_init, float2* d_flow, float4* d_entries,
                          uint64_t prev_id = 0, uint64_t cur_id = 0);     // ids != 0: frames of a stream (ofps_hip_ctx::fb_cache)
void farneback_mark_ordered(ofps_hip_ctx* ctx, hipStream_t s);
int farneback_prepare_device(ofps_hip_ctx* ctx, const uint8_t* d_img, int W, int H, int stride, int levels, int winsize, int poly_n, double poly_sigma,
                             uint64_t id, hipStream_t st);          // a stream frame's pyramid + expansion ahead of its pair's flow, on stream st
int farneback_check_params(ofps_hip_ctx* ctx, int W, int H, int levels, int winsize, int poly_n);       // what farneback_flow_device would refuse, without running it
void cluster_gate_context_created(int device);      // almeida.hip: the cluster launches' per-device gate counts the contexts alive on a device
void cluster_gate_context_destroyed(int device);
int almeida_device(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int batch, float aspect, float fov_y_deg,
                   int use_ransac, size_t num_iters, float inlier_deg, size_t num_samples, uint64_t seed, float4* d_quat);

// device -> host: by a copy KERNEL when the destination is page-locked (device-addressable) -- a D2H DMA would queue behind the
// next frame's H2D --, by hipMemcpyAsync otherwise (pipeline.hip).  bytes % 4 == 0.
int read_back_device(ofps_hip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes, hipStream_t s);
// the device address of page-locked host memory, or false for pageable memory
bool device_address_of(const void* host_ptr, void** dev_ptr);

// the batched read-ahead push with the previous frame supplied by the caller (pipeline.hip; multi.hip deals batches to workers)
int push_frames_impl(ofps_hip_ctx* ctx, const uint8_t* frames, int n, int W, int H, int stride, size_t frame_pitch,
                     const ofps_hip_frame_params* prm, float* out_entries, int* ticket, int halo_mode, const uint8_t* halo, int halo_stride);

// rows of `width` bytes, host -> device; one linear copy when both sides are dense (the 2-D path is slower)
inline hipError_t upload_rows(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                              hipStream_t stream) {
    if (dpitch == width && spitch == width) return hipMemcpyAsync(dst, src, width * height, hipMemcpyHostToDevice, stream);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, stream);
}

#define OFPS_HIP_TRY(ctx, expr)                                          \
    do {                                                                 \
        hipError_t _e = (expr);                                          \
        if (_e != hipSuccess) return ofps::check_hip((ctx), _e, #expr);  \
    } while (0)

#define OFPS_REQUIRE(ctx, cond, ...)                                              \
    do {                                                                          \
        if (!(cond)) return ofps::set_error((ctx), OFPS_HIP_EINVAL, __VA_ARGS__);  \
    } while (0)

}  // namespace ofps
