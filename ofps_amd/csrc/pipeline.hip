// pipeline.hip -- the fused per-frame path (BASELINE configs[4]: live stream, SAD decoder + block-motion
// detector + Almeida estimator per frame).  One ticket per arriving frame reproduces one iteration of the
// reference's worker loops -- decoder.process_frame -> detector.detect_motion
// (ofps-suite/src/app/detection.rs:111-148) and -> estimator.estimate
// (ofps-suite/src/app/tracking/worker.rs:328-361) -- without the motion vectors leaving the device:
//   H2D of the new luma frame into the free slot of a three-slot device ring, on a COPY stream
//   -> (compute stream, after the copy's event) SAD search between the two newest slots
//   -> detect + estimate on the device-resident vectors
//   -> one small D2H (result record, quaternion; vectors / field only when the caller asks for them).
// ofps_hip_push_frame_async returns once that is enqueued; ofps_hip_frame_wait collects a ticket.  With two tickets in
// flight the upload of frame k+1 overlaps the search of pair (k-1, k) -- what the reference's read-ahead decoder thread
// does on the host (ofps-suite/src/app/tracking/worker.rs:165-226).  ofps_hip_push_frame = push_async + wait.
#include "common.hpp"

#include <chrono>

namespace {
struct PipeOut {                 // layout of the pinned read-back block
    int result[4];               // has_motion, area, dim, 0
    float quat[4];
};

// Device -> page-locked host memory by a kernel (the destination is device-addressable) instead of hipMemcpyAsync: a
// D2H copy sits in the same in-order DMA queue as the NEXT frame's H2D and -- because it has to wait for this frame's
// search -- held that upload back until the search was over (rocprofv3 memory-copy trace, ROCm 7.2 runtime): no overlap
// at all.  The copy kernel runs on the compute stream, where it belongs.
__global__ __launch_bounds__(256) void pipe_copy_out_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n_words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// Host -> device by a copy KERNEL reading the page-locked source over PCIe: the batched read-ahead form's upload (round 5).
// hipMemcpyAsync leaves the choice of SDMA engine to the runtime, which binds a stream to the lowest-numbered engine that happens to be
// free at the stream's first copy; which engine the copy stream ends up with depends on what else was in flight at that moment, and
// the engines are not equally fast for 33 MB copies: the batched form ran at 0.0385 ms per 1080p frame in most processes and 0.046 in
// the rest (round 4's "0.95 vs 0.79 of the PCIe ceiling").  A 16-workgroup copy kernel on the copy stream sustains the same link rate
// (0.0389, every process), leaves the CUs to the search that runs beside it (64 workgroups: 0.043, 128: 0.046) and makes small
// batches faster (4 frames per batch: 0.042 against 0.050).  The single-frame form keeps the DMA engine: a lone 2 MB frame is on the
// latency path and crosses faster that way (0.0545 against 0.0628 / 0.0558 with 16 / 32 workgroups).  profiles/r05/batched_bimodal.txt;
// -DOFPS_HIP_UPLOAD_WGS=n for A/B builds (tools/upload_ab.sh).
#ifndef OFPS_HIP_UPLOAD_WGS
#define OFPS_HIP_UPLOAD_WGS 16
#endif
#ifndef OFPS_HIP_UPLOAD_KERNEL_SINGLE
#define OFPS_HIP_UPLOAD_KERNEL_SINGLE 0          // A/B: the single-frame form through the copy kernel as well
#endif
typedef unsigned int pipe_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void pipe_upload_kernel(const pipe_u32x4* __restrict__ src, pipe_u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {                // four 16-byte reads in flight per lane
        const pipe_u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const pipe_u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

bool device_can_write(const void* host_ptr, void** dev_ptr);
// dense bytes, host -> device on stream s; by_kernel: through pipe_upload_kernel when the source is page-locked (else the DMA engine)
int pipe_h2d(ofps_hip_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s, bool by_kernel) {
    void* mapped = nullptr;
    if (by_kernel && OFPS_HIP_UPLOAD_WGS > 0 && bytes % 16 == 0 && ((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0 && device_can_write(src, &mapped)) {
        hipLaunchKernelGGL(pipe_upload_kernel, dim3(OFPS_HIP_UPLOAD_WGS), dim3(256), 0, s, static_cast<const pipe_u32x4*>(mapped), static_cast<pipe_u32x4*>(dst),
                           bytes / 16);
        OFPS_HIP_TRY(ctx, hipGetLastError());
        return OFPS_HIP_OK;
    }
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    return OFPS_HIP_OK;
}

bool device_can_write(const void* host_ptr, void** dev_ptr) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, host_ptr) != hipSuccess) { (void)hipGetLastError(); return false; }   // pageable memory
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
    *dev_ptr = a.devicePointer;
    return true;
}

// bytes % 4 == 0.  Page-locked destination: copy kernel; anything else: the DMA engine.
int pipe_read_back(ofps_hip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes, hipStream_t s) {
    void* mapped = nullptr;
    if (device_can_write(host_dst, &mapped)) {
        const size_t words = bytes / 4;
        const unsigned blocks = (unsigned)((words + 255) / 256 < 64 ? (words + 255) / 256 : 64);
        hipLaunchKernelGGL(pipe_copy_out_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, static_cast<const uint32_t*>(dev_src),
                           static_cast<uint32_t*>(mapped), words);
        OFPS_HIP_TRY(ctx, hipGetLastError());
        return OFPS_HIP_OK;
    }
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s));
    return OFPS_HIP_OK;
}

int pipe_setup(ofps_hip_ctx* ctx) {
    if (ctx->pipe_copy_stream) return OFPS_HIP_OK;
    OFPS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->pipe_copy_stream, hipStreamNonBlocking));
    OFPS_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->pipe_aux_stream, hipStreamNonBlocking));
    OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_fork, hipEventDisableTiming));
    OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_join, hipEventDisableTiming));
    for (int k = 0; k < ofps_hip_ctx::kPipeSlots; ++k) {
        OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_uploaded[k], hipEventDisableTiming));
        OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_slot_read[k], hipEventDisableTiming));
    }
    for (auto& t : ctx->pipe_ticket) {
        OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
        // kernels store the result record straight into this block and the host reads it after a hipEventDisableTiming event
        // (no release-to-system fence of its own): the block must be FINE-GRAINED host memory whatever HIP_HOST_COHERENT or
        // a future runtime default says -- asked for explicitly
        OFPS_HIP_TRY(ctx, hipHostMalloc(&t.pinned, sizeof(PipeOut), OFPS_HIP_HOST_BLOCK_FLAGS));
    }
    return OFPS_HIP_OK;
}

// Waits for every ticket still in flight and forgets the stream position (geometry change / reset).
int pipe_drain(ofps_hip_ctx* ctx) {
    for (auto& t : ctx->pipe_ticket) {
        if (t.pending && t.done) OFPS_HIP_TRY(ctx, hipEventSynchronize(t.done));
        t.pending = false;
    }
    if (ctx->pipe_copy_stream) OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->pipe_copy_stream));
    ctx->pipe_frames = 0;
    for (bool& v : ctx->pipe_slot_read_valid) v = false;
    return OFPS_HIP_OK;
}

// Enqueues the H2D of one luma frame as frame number ctx->pipe_frames (slot = number % 3).  With another ticket in
// flight the copy goes to the copy stream, so that it overlaps that ticket's search; a lone frame is copied on the
// compute stream itself (no cross-stream events on the latency path of the synchronous call: 0.10 vs 0.17 ms per
// 1080p frame).
int pipe_upload(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride, bool overlap, uint8_t** slots_out,
                size_t* pitch_out, int* dstride_out) {
    int rc = pipe_setup(ctx);
    if (rc != OFPS_HIP_OK) return rc;
    const int dstride = (W + 63) & ~63;
    const size_t pitch = (size_t)dstride * H;
    if (W != ctx->pipe_w || H != ctx->pipe_h) {            // geometry change restarts the stream (decoder.rs:66-72)
        rc = pipe_drain(ctx);
        if (rc != OFPS_HIP_OK) return rc;
        ctx->pipe_w = W; ctx->pipe_h = H; ctx->pipe_stride = dstride;
    }
    auto* slots = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_PIPE_FRAMES, ofps_hip_ctx::kPipeSlots * pitch));
    if (!slots) return OFPS_HIP_ENOMEM;
    const int slot = (int)(ctx->pipe_frames % ofps_hip_ctx::kPipeSlots);
    hipStream_t up = overlap ? ctx->pipe_copy_stream : ctx->stream;
    // the slot's previous tenant (frame number - 3) may still be read by the search of ticket number - 2
    if (overlap && ctx->pipe_slot_read_valid[slot]) OFPS_HIP_TRY(ctx, hipStreamWaitEvent(up, ctx->pipe_slot_read[slot], 0));
    if (dstride == W && stride == W) {
        rc = pipe_h2d(ctx, slots + (size_t)slot * pitch, luma, pitch, up, OFPS_HIP_UPLOAD_KERNEL_SINGLE != 0);
        if (rc != OFPS_HIP_OK) return rc;
    } else {
        OFPS_HIP_TRY(ctx, ofps::upload_rows(slots + (size_t)slot * pitch, dstride, luma, stride, W, H, up));
    }
    if (overlap) OFPS_HIP_TRY(ctx, hipEventRecord(ctx->pipe_uploaded[slot], up));
    ctx->pipe_uploaded_on_compute[slot] = !overlap;       // ... in which case the compute stream never has to wait for it
    ctx->pipe_frames += 1;
    *slots_out = slots; *pitch_out = pitch; *dstride_out = dstride;
    return OFPS_HIP_OK;
}
}  // namespace

namespace ofps {
int read_back_device(ofps_hip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes, hipStream_t s) {
    return pipe_read_back(ctx, host_dst, dev_src, bytes, s);
}
bool device_address_of(const void* host_ptr, void** dev_ptr) { return device_can_write(host_ptr, dev_ptr); }
}  // namespace ofps

extern "C" {

int ofps_hip_reset_frames(ofps_hip_ctx* ctx) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (auto& bt : ctx->batch_ticket) {
        if (bt.pending && bt.done) OFPS_HIP_TRY(ctx, hipEventSynchronize(bt.done));
        bt.pending = false;
    }
    ctx->batch_frames = 0; ctx->batch_last_frame = nullptr;
    return pipe_drain(ctx);
}

int ofps_hip_stage_frame(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, luma, "stage_frame: null pointer");
    OFPS_REQUIRE(ctx, W > 0 && H > 0 && stride >= W, "stage_frame: bad geometry W=%d H=%d stride=%d", W, H, stride);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint8_t* slots; size_t pitch; int dstride;
    int rc = pipe_upload(ctx, luma, W, H, stride, /*overlap=*/false, &slots, &pitch, &dstride);
    if (rc != OFPS_HIP_OK) return rc;
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));             // the caller may reuse `luma` right away
    return OFPS_HIP_OK;
}

int ofps_hip_push_frame_async(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride,
                              const ofps_hip_frame_params* prm, float* out_entries, float* out_field, int* ticket) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, luma && prm && ticket, "push_frame_async: null pointer");
    OFPS_REQUIRE(ctx, W > 0 && H > 0 && stride >= W, "push_frame_async: bad geometry W=%d H=%d stride=%d", W, H, stride);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = pipe_setup(ctx);
    if (rc != OFPS_HIP_OK) return rc;
    const long tno = ctx->pipe_next_ticket;
    auto& t = ctx->pipe_ticket[tno % ofps_hip_ctx::kPipeTickets];
    OFPS_REQUIRE(ctx, !t.pending, "push_frame_async: ticket %ld has not been collected (at most %d frames in flight)",
                 tno - ofps_hip_ctx::kPipeTickets, ofps_hip_ctx::kPipeTickets);
    hipStream_t s = ctx->stream;
    uint8_t* slots; size_t pitch; int dstride;
    const bool overlap = ctx->pipe_ticket[(tno + 1) % ofps_hip_ctx::kPipeTickets].pending;      // the other ticket is in flight
    rc = pipe_upload(ctx, luma, W, H, stride, overlap, &slots, &pitch, &dstride);
    if (rc != OFPS_HIP_OK) return rc;
    const long frame_no = ctx->pipe_frames - 1;                       // the frame just enqueued
    const int cur_slot = (int)(frame_no % ofps_hip_ctx::kPipeSlots);
    t.have_vectors = 0; t.n_vectors = 0; t.run_detector = prm->run_detector; t.run_estimator = prm->run_estimator;
    const size_t nblk = ofps_hip_sad_block_count(W, H, prm->block);
    if (frame_no == 0) {                                             // first frame of a stream: Ok(false), no vectors yet
        if (!ctx->pipe_uploaded_on_compute[cur_slot]) {
            OFPS_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->pipe_uploaded[cur_slot], 0));
            ctx->pipe_uploaded_on_compute[cur_slot] = true;
        }
        OFPS_HIP_TRY(ctx, hipEventRecord(t.done, s));
        t.pending = true;
        *ticket = (int)(tno & 0x7FFFFFFF);
        ctx->pipe_next_ticket = tno + 1;
        return OFPS_HIP_OK;
    }
    const int prev_slot = (int)((frame_no - 1) % ofps_hip_ctx::kPipeSlots);
    const int tix = (int)(tno % ofps_hip_ctx::kPipeTickets);
    auto* d_ent_all = static_cast<float4*>(ofps::scratch(ctx, ofps::S_PIPE_ENTRIES, ofps_hip_ctx::kPipeTickets * nblk * sizeof(float4)));
    constexpr size_t kOutBytes = 4096 + (size_t)160 * 160 * sizeof(float2);
    auto* d_out_all = static_cast<char*>(ofps::scratch(ctx, ofps::S_PIPE_OUT, ofps_hip_ctx::kPipeTickets * kOutBytes));
    if (!d_ent_all || !d_out_all) return OFPS_HIP_ENOMEM;
    float4* d_ent = d_ent_all + (size_t)tix * nblk;
    char* d_out = d_out_all + (size_t)tix * kOutBytes;
    // the search needs both frames on the device: the previous frame's upload was waited for by the previous ticket
    // (or by the stage_frame that made it), this frame's by the event
    // (uploads made on the compute stream itself are ordered by the stream; one wait per upload is enough)
    for (int slot : {prev_slot, cur_slot}) {
        if (!ctx->pipe_uploaded_on_compute[slot]) {
            OFPS_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->pipe_uploaded[slot], 0));
            ctx->pipe_uploaded_on_compute[slot] = true;
        }
    }
    rc = ofps::sad_pairs_device(ctx, slots + (size_t)prev_slot * pitch, 0, slots + (size_t)cur_slot * pitch, 0, 1, W, H, dstride,
                                prm->block, prm->range, d_ent, nullptr);
    if (rc != OFPS_HIP_OK) return rc;
    // the older slot may be overwritten once this search is through; the same event forks the detector's stream below
    // (one barrier packet between the search and the estimator instead of two)
    OFPS_HIP_TRY(ctx, hipEventRecord(ctx->pipe_slot_read[prev_slot], s));
    ctx->pipe_slot_read_valid[prev_slot] = true;
    t.have_vectors = 1;
    t.n_vectors = nblk;

    int dim = 0;
    int* d_res = reinterpret_cast<int*>(d_out);
    float4* d_quat = reinterpret_cast<float4*>(d_out + 16);
    float2* d_field = reinterpret_cast<float2*>(d_out + 4096);
    // The 32 bytes the caller waits for (island result, quaternion) are written by the detector's and the estimator's last
    // kernels STRAIGHT into the ticket's page-locked block -- it is device-addressable, each is one thread's store at the
    // end of a kernel -- instead of into device scratch and from there by a copy launch behind the join: that launch and
    // the gap in front of it were 14 us of a 170 us frame (rocprofv3 kernel trace, tools/trace_stream.sh).
    void* mapped_out = nullptr;
    const bool direct = device_can_write(t.pinned, &mapped_out);
    if (direct) {
        d_res = reinterpret_cast<int*>(static_cast<char*>(mapped_out) + offsetof(PipeOut, result));
        d_quat = reinterpret_cast<float4*>(static_cast<char*>(mapped_out) + offsetof(PipeOut, quat));
    }
    // detector and estimator read the same device-resident vectors and share no workspace: with both enabled the
    // detector's chain of small launches runs on an auxiliary stream beside the estimator (fork after the search, join
    // before the read-back) instead of in front of it
    const bool fork = prm->run_detector && prm->run_estimator;
    if (fork) OFPS_HIP_TRY(ctx, hipStreamWaitEvent(ctx->pipe_aux_stream, ctx->pipe_slot_read[prev_slot], 0));
    // the estimator is enqueued first: it is the long pole (0.1 ms of dependent steps against the detector's seven small
    // launches), and whatever is enqueued second starts a host-enqueue time later
    if (prm->run_estimator) {
        rc = ofps::almeida_device(ctx, d_ent, nblk, 1, prm->aspect, prm->fov_y_deg, prm->use_ransac, prm->num_iters,
                                  prm->inlier_deg, prm->num_samples, prm->seed, d_quat);
        if (rc != OFPS_HIP_OK) return rc;
    }
    if (prm->run_detector) {
        if (fork) ctx->stream = ctx->pipe_aux_stream;           // the stage entry points enqueue on ctx->stream
        rc = ofps::detect_device(ctx, d_ent, nblk, 1, prm->min_size, prm->subdivide, prm->target_motion, d_res, d_field, &dim);
        if (fork) {
            ctx->stream = s;
            if (rc == OFPS_HIP_OK) OFPS_HIP_TRY(ctx, hipEventRecord(ctx->pipe_join, ctx->pipe_aux_stream));
        }
        if (rc != OFPS_HIP_OK) return rc;
    }
    if (fork) OFPS_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->pipe_join, 0));
    if ((prm->run_detector || prm->run_estimator) && !direct) {
        rc = pipe_read_back(ctx, t.pinned, d_out, sizeof(PipeOut), s);
        if (rc != OFPS_HIP_OK) return rc;
    }
    if (out_entries && nblk) {
        rc = pipe_read_back(ctx, out_entries, d_ent, nblk * sizeof(float4), s);
        if (rc != OFPS_HIP_OK) return rc;
    }
    if (out_field && prm->run_detector) {
        rc = pipe_read_back(ctx, out_field, d_field, (size_t)dim * dim * sizeof(float2), s);
        if (rc != OFPS_HIP_OK) return rc;
    }
    OFPS_HIP_TRY(ctx, hipEventRecord(t.done, s));
    t.pending = true;
    *ticket = (int)(tno & 0x7FFFFFFF);
    ctx->pipe_next_ticket = tno + 1;
    return OFPS_HIP_OK;
}

int ofps_hip_frame_wait(ofps_hip_ctx* ctx, int ticket, ofps_hip_frame_result* out) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, out, "frame_wait: null pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long newest = ctx->pipe_next_ticket - 1;
    long tno = -1;
    for (long k = newest; k >= 0 && k > newest - ofps_hip_ctx::kPipeTickets; --k)
        if ((int)(k & 0x7FFFFFFF) == ticket) { tno = k; break; }
    OFPS_REQUIRE(ctx, tno >= 0, "frame_wait: ticket %d is not in flight", ticket);
    auto& t = ctx->pipe_ticket[tno % ofps_hip_ctx::kPipeTickets];
    OFPS_REQUIRE(ctx, t.pending, "frame_wait: ticket %d was already collected", ticket);
    // a per-frame result is tens of microseconds away: poll first (hipEventSynchronize may put the thread to sleep, and a
    // wake-up costs more than the whole frame -- 0.23 vs 0.06 ms per frame measured inside a process that initialised
    // torch's runtime), then block
    {
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t q;
        while ((q = hipEventQuery(t.done)) == hipErrorNotReady) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(OFPS_HIP_FRAME_WAIT_SPIN_US)) break;
        }
        if (q != hipSuccess) {
            if (q != hipErrorNotReady) OFPS_HIP_TRY(ctx, q);
            (void)hipGetLastError();
            OFPS_HIP_TRY(ctx, hipEventSynchronize(t.done));
        }
    }
    t.pending = false;
    memset(out, 0, sizeof(*out));
    out->quat[0] = 1.0f;
    out->have_vectors = t.have_vectors;
    out->n_vectors = t.n_vectors;
    if (t.have_vectors) {
        const auto* host = static_cast<const PipeOut*>(t.pinned);
        if (t.run_detector) {
            out->has_motion = host->result[0];
            out->area = (size_t)host->result[1];
            out->dim = host->result[2];
        }
        if (t.run_estimator) memcpy(out->quat, host->quat, sizeof(out->quat));
    }
    return OFPS_HIP_OK;
}

// ---- batched read-ahead form: n consecutive frames of the stream per ticket.  What a decoder that runs n frames ahead
// (ofps-suite/src/app/tracking/worker.rs:165-226 decodes into a buffer on its own thread) hands over in one go: ONE H2D of the
// n frames (contiguous at frame_pitch), one search launch over the batch's pairs, one detector chain and one estimator
// launch over the batch, one read-back -- a handful of HIP calls per BATCH instead of ~9 per frame, which is what kept
// the single-frame loop 20 % under the PCIe ceiling.  Frame j of the batch is pair (previous frame of the stream, frame
// j); the previous frame of frame 0 is the last frame of the previous batch, kept in slot 0 of the other batch buffer.
}  // extern "C"

namespace ofps {
// halo_mode 0: frame 0 of the batch is paired with the context's own previous frame (ofps_hip_push_frames_async).
// halo_mode 1: the caller supplies the previous frame (`halo`, host memory, rows `halo_stride` bytes apart; 0 = `stride`) or says there is none
// (halo == nullptr: the very first frame of a stream) -- the multi-device dispatcher's form (multi.hip): a worker sees every
// n_workers-th batch of a stream, so the frame in front of its batch is not the last one IT saw.
int push_frames_impl(ofps_hip_ctx* ctx, const uint8_t* frames, int n, int W, int H, int stride, size_t frame_pitch,
                     const ofps_hip_frame_params* prm, float* out_entries, int* ticket, int halo_mode, const uint8_t* halo, int halo_stride) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, frames && prm && ticket && n >= 1 && n <= 4096, "push_frames_async: bad arguments (n=%d)", n);
    OFPS_REQUIRE(ctx, W > 0 && H > 0 && stride >= W && frame_pitch >= (size_t)stride * H, "push_frames_async: bad geometry W=%d H=%d stride=%d", W, H, stride);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = pipe_setup(ctx);
    if (rc != OFPS_HIP_OK) return rc;
    const long tno = ctx->batch_next_ticket;
    auto& t = ctx->batch_ticket[tno % ofps_hip_ctx::kBatchTickets];
    OFPS_REQUIRE(ctx, !t.pending, "push_frames_async: ticket %ld has not been collected (at most %d batches in flight)",
                 tno - ofps_hip_ctx::kBatchTickets, ofps_hip_ctx::kBatchTickets);
    if (!t.done) {
        OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
        OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&t.uploaded, hipEventDisableTiming));
        OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&t.prev_copied, hipEventDisableTiming));
    }
    if (W != ctx->batch_w || H != ctx->batch_h) {              // geometry change restarts the stream
        for (auto& bt : ctx->batch_ticket)
            if (bt.pending && bt.done) { OFPS_HIP_TRY(ctx, hipEventSynchronize(bt.done)); bt.pending = false; }
        ctx->batch_w = W; ctx->batch_h = H; ctx->batch_frames = 0; ctx->batch_last_frame = nullptr;
    }
    const int dstride = (W + 63) & ~63;
    const size_t pitch = (size_t)dstride * H;
    const size_t nblk = ofps_hip_sad_block_count(W, H, prm->block);
    const int tix = (int)(tno % ofps_hip_ctx::kBatchTickets);
    // capacity: both buffers and the per-ticket outputs are sized for the largest batch seen (grow-only; growing waits
    // for work in flight)
    constexpr size_t kOutBytes = 32;                             // {result[4], quat[4]} per frame
    const size_t cap_frames = (size_t)n + 1;
    auto& fs = ctx->scratch[ofps::S_BATCH_FRAMES];
    size_t per_buf = fs.cap / ofps_hip_ctx::kBatchTickets / (pitch ? pitch : 1);
    if (per_buf < cap_frames) {
        for (auto& bt : ctx->batch_ticket)
            if (bt.pending && bt.done) OFPS_HIP_TRY(ctx, hipEventSynchronize(bt.done));
        // the newest frame of the stream lives in the old allocation: keep a copy
        void* keep = nullptr;
        if (ctx->batch_last_frame) {
            OFPS_HIP_TRY(ctx, hipMalloc(&keep, pitch));
            OFPS_HIP_TRY(ctx, hipMemcpyAsync(keep, ctx->batch_last_frame, pitch, hipMemcpyDeviceToDevice, ctx->stream));
            OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        auto* nb = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_BATCH_FRAMES, ofps_hip_ctx::kBatchTickets * cap_frames * pitch));
        if (!nb) { if (keep) (void)hipFree(keep); return OFPS_HIP_ENOMEM; }
        per_buf = cap_frames;
        if (keep) {
            // parked in the LAST slot of the buffer this ticket does not use: nothing writes there before it is consumed
            uint8_t* park = nb + ((size_t)(tix ^ 1) * per_buf + (per_buf - 1)) * pitch;
            OFPS_HIP_TRY(ctx, hipMemcpyAsync(park, keep, pitch, hipMemcpyDeviceToDevice, ctx->stream));
            OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(keep);
            ctx->batch_last_frame = park;
        }
    }
    auto* bufs = static_cast<uint8_t*>(fs.p);
    uint8_t* buf = bufs + (size_t)tix * per_buf * pitch;        // [slot 0 = previous frame][n frames]
    auto* d_ent_all = static_cast<float4*>(ofps::scratch(ctx, ofps::S_BATCH_ENTRIES, ofps_hip_ctx::kBatchTickets * (size_t)n * nblk * sizeof(float4)));
    auto* d_out_all = static_cast<char*>(ofps::scratch(ctx, ofps::S_BATCH_OUT, ofps_hip_ctx::kBatchTickets * (size_t)n * kOutBytes));
    if (!d_ent_all || !d_out_all) return OFPS_HIP_ENOMEM;
    const size_t ent_per_ticket = ctx->scratch[ofps::S_BATCH_ENTRIES].cap / ofps_hip_ctx::kBatchTickets / sizeof(float4);
    const size_t out_per_ticket = ctx->scratch[ofps::S_BATCH_OUT].cap / ofps_hip_ctx::kBatchTickets;
    float4* d_ent = d_ent_all + (size_t)tix * ent_per_ticket;
    char* d_out = d_out_all + (size_t)tix * out_per_ticket;
    if (t.pinned_cap < (size_t)n * kOutBytes) {
        if (t.pinned) OFPS_HIP_TRY(ctx, hipHostFree(t.pinned));
        t.pinned = nullptr; t.pinned_cap = 0;
        OFPS_HIP_TRY(ctx, hipHostMalloc(&t.pinned, (size_t)n * kOutBytes, OFPS_HIP_HOST_BLOCK_FLAGS));   // fine-grained: written by kernels, read after an event
        t.pinned_cap = (size_t)n * kOutBytes;
    }
    hipStream_t s = ctx->stream, up = ctx->pipe_copy_stream;
    // ---- copy stream: the n frames in one transfer (the buffer's previous tenant, ticket tno - 2, has been collected:
    // its work is done); compute stream: the previous frame into slot 0
    // the other ticket's copy of ITS previous frame reads slot n of this buffer's previous tenant: wait for it
    auto& other = ctx->batch_ticket[(tno + 1) % ofps_hip_ctx::kBatchTickets];
    if (other.pending && other.prev_copied_valid) OFPS_HIP_TRY(ctx, hipStreamWaitEvent(up, other.prev_copied, 0));
    if (frame_pitch == (size_t)W * H && stride == W && dstride == W) {
        rc = pipe_h2d(ctx, buf + pitch, frames, (size_t)n * pitch, up, true);                                       // the whole batch
        if (rc != OFPS_HIP_OK) return rc;
    } else {
        for (int j = 0; j < n; ++j)
            OFPS_HIP_TRY(ctx, ofps::upload_rows(buf + (size_t)(j + 1) * pitch, dstride, frames + (size_t)j * frame_pitch, stride, W, H, up));
    }
    if (halo_mode && halo) OFPS_HIP_TRY(ctx, ofps::upload_rows(buf, dstride, halo, halo_stride ? halo_stride : stride, W, H, up));       // the caller's previous frame into slot 0
    OFPS_HIP_TRY(ctx, hipEventRecord(t.uploaded, up));
    const bool has_prev = halo_mode ? halo != nullptr : ctx->batch_last_frame != nullptr;
    t.prev_copied_valid = false;
    if (has_prev && !halo_mode) {
        // by a copy KERNEL, not hipMemcpyAsync: the runtime may hand a device-to-device copy to the SDMA engine that is busy with the
        // NEXT batch's 33 MB upload, and then this 2 MB copy -- and the search behind it -- waits 0.1-0.6 ms for that upload.  Which
        // engine a stream's copies get is decided per process: the batched form ran at 209 Mvectors/s in some processes and 173 in
        // others (round 4's 0.95 vs 0.79 of the PCIe ceiling; profiles/r05/batched_bimodal.txt)
        hipLaunchKernelGGL(pipe_copy_out_kernel, dim3(256), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(ctx->batch_last_frame),
                           reinterpret_cast<uint32_t*>(buf), pitch / 4);
        OFPS_HIP_TRY(ctx, hipGetLastError());
        OFPS_HIP_TRY(ctx, hipEventRecord(t.prev_copied, s));
        t.prev_copied_valid = true;
    }
    OFPS_HIP_TRY(ctx, hipStreamWaitEvent(s, t.uploaded, 0));
    // ---- compute stream: pairs (slot j, slot j + 1), j = first .. n - 1
    const int first = has_prev ? 0 : 1;                           // the stream's very first frame has no pair
    const int pairs = n - first;
    t.n = n; t.first_has_prev = has_prev ? 1 : 0; t.run_detector = prm->run_detector; t.run_estimator = prm->run_estimator; t.n_vectors = nblk;
    if (pairs > 0) {
        float4* ent0 = d_ent + (size_t)first * nblk;
        rc = ofps::sad_pairs_device(ctx, buf + (size_t)first * pitch, pitch, buf + (size_t)(first + 1) * pitch, pitch, pairs, W, H, dstride,
                                    prm->block, prm->range, ent0, nullptr);
        if (rc != OFPS_HIP_OK) return rc;
        int* d_res = reinterpret_cast<int*>(d_out);                                 // [n][4]
        float4* d_quat = reinterpret_cast<float4*>(d_out + (size_t)n * 16);          // [n]
        if (prm->run_estimator) {
            rc = ofps::almeida_device(ctx, ent0, nblk, pairs, prm->aspect, prm->fov_y_deg, prm->use_ransac, prm->num_iters, prm->inlier_deg,
                                      prm->num_samples, prm->seed + (uint64_t)first, d_quat + first);
            if (rc != OFPS_HIP_OK) return rc;
        }
        if (prm->run_detector) {
            int dim = 0;
            auto* d_field = static_cast<float2*>(ofps::scratch(ctx, ofps::S_BATCH_FIELD, (size_t)pairs * 160 * 160 * sizeof(float2)));   // its own slot: the densifier works in S_WORK*
            if (!d_field) return OFPS_HIP_ENOMEM;
            rc = ofps::detect_device(ctx, ent0, nblk, pairs, prm->min_size, prm->subdivide, prm->target_motion, d_res + 4 * first, d_field, &dim);
            if (rc != OFPS_HIP_OK) return rc;
        }
        if (prm->run_detector || prm->run_estimator) {
            rc = pipe_read_back(ctx, t.pinned, d_out, (size_t)n * kOutBytes, s);
            if (rc != OFPS_HIP_OK) return rc;
        }
        if (out_entries) {
            rc = pipe_read_back(ctx, out_entries + (size_t)first * nblk * 4, ent0, (size_t)pairs * nblk * sizeof(float4), s);
            if (rc != OFPS_HIP_OK) return rc;
        }
    }
    OFPS_HIP_TRY(ctx, hipEventRecord(t.done, s));
    ctx->batch_last_frame = buf + (size_t)n * pitch;
    ctx->batch_frames += n;
    t.pending = true;
    *ticket = (int)(tno & 0x7FFFFFFF);
    ctx->batch_next_ticket = tno + 1;
    return OFPS_HIP_OK;
}
}  // namespace ofps

extern "C" {

int ofps_hip_push_frames_async(ofps_hip_ctx* ctx, const uint8_t* frames, int n, int W, int H, int stride, size_t frame_pitch,
                               const ofps_hip_frame_params* prm, float* out_entries, int* ticket) {
    return ofps::push_frames_impl(ctx, frames, n, W, H, stride, frame_pitch, prm, out_entries, ticket, 0, nullptr, 0);
}

int ofps_hip_frames_wait(ofps_hip_ctx* ctx, int ticket, ofps_hip_frame_result* out /* n of them */) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, out, "frames_wait: null pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long newest = ctx->batch_next_ticket - 1;
    long tno = -1;
    for (long k = newest; k >= 0 && k > newest - ofps_hip_ctx::kBatchTickets; --k)
        if ((int)(k & 0x7FFFFFFF) == ticket) { tno = k; break; }
    OFPS_REQUIRE(ctx, tno >= 0, "frames_wait: ticket %d is not in flight", ticket);
    auto& t = ctx->batch_ticket[tno % ofps_hip_ctx::kBatchTickets];
    OFPS_REQUIRE(ctx, t.pending, "frames_wait: ticket %d was already collected", ticket);
    OFPS_HIP_TRY(ctx, hipEventSynchronize(t.done));
    t.pending = false;
    const auto* res = static_cast<const int*>(t.pinned);
    const auto* quat = reinterpret_cast<const float*>(static_cast<const char*>(t.pinned) + (size_t)t.n * 16);
    for (int j = 0; j < t.n; ++j) {
        ofps_hip_frame_result& o = out[j];
        memset(&o, 0, sizeof(o));
        o.quat[0] = 1.0f;
        const bool has = j > 0 || t.first_has_prev;
        o.have_vectors = has ? 1 : 0;
        o.n_vectors = has ? t.n_vectors : 0;
        if (!has) continue;
        if (t.run_detector) { o.has_motion = res[4 * j]; o.area = (size_t)res[4 * j + 1]; o.dim = res[4 * j + 2]; }
        if (t.run_estimator) memcpy(o.quat, quat + 4 * j, sizeof(o.quat));
    }
    return OFPS_HIP_OK;
}

int ofps_hip_push_frame(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride,
                        const ofps_hip_frame_params* prm, ofps_hip_frame_result* out, float* out_entries,
                        float* out_field) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, luma && prm && out, "push_frame: null pointer");
    int ticket = 0;
    int rc = ofps_hip_push_frame_async(ctx, luma, W, H, stride, prm, out_entries, out_field, &ticket);
    if (rc != OFPS_HIP_OK) return rc;
    return ofps_hip_frame_wait(ctx, ticket, out);
}

}  // extern "C"
