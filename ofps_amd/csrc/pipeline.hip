// pipeline.hip -- the fused per-frame path (BASELINE configs[4]: live stream, SAD decoder + block-motion
// detector + Almeida estimator per frame).  One call per arriving frame reproduces one iteration of the
// reference's worker loops -- decoder.process_frame -> detector.detect_motion
// (ofps-suite/src/app/detection.rs:111-148) and -> estimator.estimate
// (ofps-suite/src/app/tracking/worker.rs:328-361) -- without the motion vectors leaving the device:
//   H2D of the new luma frame into one of two device slots (the other holds the previous frame)
//   -> SAD search between the slots -> detect + estimate on the device-resident vectors
//   -> one small D2H (result record, quaternion; vectors / field only when the caller asks for them).
#include "common.hpp"

namespace {
struct PipeOut {                 // layout of the pinned read-back block
    int result[4];               // has_motion, area, dim, 0
    float quat[4];
};
}  // namespace

extern "C" {

int ofps_hip_reset_frames(ofps_hip_ctx* ctx) {
    if (!ctx) return OFPS_HIP_EINVAL;
    ctx->pipe_newest = -1;
    return OFPS_HIP_OK;
}

// H2D of one luma frame into the device slot that does not hold the newest frame; slots/pitch/prev/cur are outputs
static int upload_frame(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride, uint8_t** slots_out, size_t* pitch_out,
                        int* dstride_out, int* prev_slot_out, int* cur_slot_out) {
    const int dstride = (W + 63) & ~63;
    const size_t pitch = (size_t)dstride * H;
    if (W != ctx->pipe_w || H != ctx->pipe_h) {            // geometry change restarts the stream (decoder.rs:66-72)
        ctx->pipe_w = W; ctx->pipe_h = H; ctx->pipe_stride = dstride; ctx->pipe_newest = -1;
    }
    auto* slots = static_cast<uint8_t*>(ofps::scratch(ctx, ofps::S_PIPE_FRAMES, 2 * pitch));
    if (!slots) return OFPS_HIP_ENOMEM;
    const int prev_slot = ctx->pipe_newest, cur_slot = ctx->pipe_newest < 0 ? 0 : 1 - ctx->pipe_newest;
    OFPS_HIP_TRY(ctx, ofps::upload_rows(slots + (size_t)cur_slot * pitch, dstride, luma, stride, W, H, ctx->stream));
    ctx->pipe_newest = cur_slot;
    *slots_out = slots; *pitch_out = pitch; *dstride_out = dstride; *prev_slot_out = prev_slot; *cur_slot_out = cur_slot;
    return OFPS_HIP_OK;
}

int ofps_hip_stage_frame(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, luma, "stage_frame: null pointer");
    OFPS_REQUIRE(ctx, W > 0 && H > 0 && stride >= W, "stage_frame: bad geometry W=%d H=%d stride=%d", W, H, stride);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint8_t* slots; size_t pitch; int dstride, prev_slot, cur_slot;
    int rc = upload_frame(ctx, luma, W, H, stride, &slots, &pitch, &dstride, &prev_slot, &cur_slot);
    if (rc != OFPS_HIP_OK) return rc;
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the caller may reuse `luma` right away
    return OFPS_HIP_OK;
}

int ofps_hip_push_frame(ofps_hip_ctx* ctx, const uint8_t* luma, int W, int H, int stride,
                        const ofps_hip_frame_params* prm, ofps_hip_frame_result* out, float* out_entries,
                        float* out_field) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, luma && prm && out, "push_frame: null pointer");
    OFPS_REQUIRE(ctx, W > 0 && H > 0 && stride >= W, "push_frame: bad geometry W=%d H=%d stride=%d", W, H, stride);
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (!ctx->pipe_pinned) OFPS_HIP_TRY(ctx, hipHostMalloc(&ctx->pipe_pinned, sizeof(PipeOut), hipHostMallocDefault));
    uint8_t* slots; size_t pitch; int dstride, prev_slot, cur_slot;
    {
        int rc = upload_frame(ctx, luma, W, H, stride, &slots, &pitch, &dstride, &prev_slot, &cur_slot);
        if (rc != OFPS_HIP_OK) return rc;
    }

    memset(out, 0, sizeof(*out));
    out->quat[0] = 1.0f;
    const size_t nblk = ofps_hip_sad_block_count(W, H, prm->block);
    if (prev_slot < 0) {                                    // first frame of a stream: Ok(false), no vectors yet
        OFPS_HIP_TRY(ctx, hipStreamSynchronize(s));
        return OFPS_HIP_OK;
    }
    auto* d_ent = static_cast<float4*>(ofps::scratch(ctx, ofps::S_PIPE_ENTRIES, nblk * sizeof(float4)));
    if (!d_ent) return OFPS_HIP_ENOMEM;
    int rc = ofps::sad_pairs_device(ctx, slots + (size_t)prev_slot * pitch, 0, slots + (size_t)cur_slot * pitch, 0, 1, W, H, dstride,
                                    prm->block, prm->range, d_ent, nullptr);
    if (rc != OFPS_HIP_OK) return rc;
    out->have_vectors = 1;
    out->n_vectors = nblk;

    int dim = 0;
    float2* d_field = nullptr;
    auto* d_out = static_cast<char*>(ofps::scratch(ctx, ofps::S_PIPE_OUT, 4096 + (size_t)160 * 160 * sizeof(float2)));
    if (!d_out) return OFPS_HIP_ENOMEM;
    int* d_res = reinterpret_cast<int*>(d_out);
    float4* d_quat = reinterpret_cast<float4*>(d_out + 16);
    d_field = reinterpret_cast<float2*>(d_out + 4096);
    if (prm->run_detector) {
        rc = ofps::detect_device(ctx, d_ent, nblk, 1, prm->min_size, prm->subdivide, prm->target_motion, d_res, d_field, &dim);
        if (rc != OFPS_HIP_OK) return rc;
    }
    if (prm->run_estimator) {
        rc = ofps::almeida_device(ctx, d_ent, nblk, 1, prm->aspect, prm->fov_y_deg, prm->use_ransac, prm->num_iters,
                                  prm->inlier_deg, prm->num_samples, prm->seed, d_quat);
        if (rc != OFPS_HIP_OK) return rc;
    }
    auto* host = static_cast<PipeOut*>(ctx->pipe_pinned);
    if (prm->run_detector || prm->run_estimator)
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(host, d_out, sizeof(PipeOut), hipMemcpyDeviceToHost, s));
    if (out_entries && nblk)
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_entries, d_ent, nblk * sizeof(float4), hipMemcpyDeviceToHost, s));
    if (out_field && prm->run_detector)
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_field, d_field, (size_t)dim * dim * sizeof(float2), hipMemcpyDeviceToHost, s));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(s));
    if (prm->run_detector) {
        out->has_motion = host->result[0];
        out->area = (size_t)host->result[1];
        out->dim = host->result[2];
    }
    if (prm->run_estimator) memcpy(out->quat, host->quat, sizeof(out->quat));
    return OFPS_HIP_OK;
}

}  // extern "C"
