// detect.hip -- A5: BlockMotionDetection::detect_motion on gfx950
// (block-motion-detector/src/lib.rs:49-118; trait ofps/src/detection.rs:11).
//
// densify (densify.hip, exact input-order sums) -> one workgroup per item:
//   * map[cell] = |m| >= target_motion                                  (lib.rs:63-68)
//   * 8-connected components by min-label propagation with pointer jumping in LDS; the label of
//     a component is its smallest cell index = the raster-scan seed of the reference's flood
//     fill (lib.rs:74-76)
//   * area per label with LDS integer atomics; winner = max (area, earliest seed): the
//     reference keeps the first island under a strict `>` (lib.rs:106-109)
//   * output = winner's cells except its seed cell, which the reference never copies
//     (lib.rs:79-102); Some iff area / dim^2 >= min_size                (lib.rs:114-118)
// Everything integer is order-independent, so area and membership are bit-exact by construction.
#include "common.hpp"

namespace ofps {

// dynamic LDS: area[cells] (u32) then label[cells] (u16; every cell is owned by one thread, so
// labels need no atomics).  160x160 cells -> 150 KiB, inside the 160 KiB of one CU.
__global__ __launch_bounds__(1024) void detect_kernel(const float2* __restrict__ field, int dim, float target_motion,
                                                      float min_size, int* __restrict__ out_result,
                                                      float2* __restrict__ out_field) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ unsigned long long best_key;
    const int cells = dim * dim;
    uint32_t* area = lds;
    uint16_t* label = reinterpret_cast<uint16_t*>(lds + cells);
    const size_t item = blockIdx.x;
    const float2* f = field + item * cells;
    const uint32_t NONE = 0xFFFFu;

    for (int i = threadIdx.x; i < cells; i += 1024) {
        const float2 m = f[i];
        const float mag = sqrtf(m.x * m.x + m.y * m.y);                 // Vector2::magnitude
        label[i] = (uint16_t)((mag >= target_motion) ? (uint32_t)i : NONE);
        area[i] = 0;
    }
    if (threadIdx.x == 0) best_key = 0;
    __syncthreads();

    for (;;) {
        int changed = 0;
        for (int i = threadIdx.x; i < cells; i += 1024) {
            uint32_t l = label[i];
            if (l == NONE) continue;
            const int x = i % dim, y = i / dim;
            uint32_t m = l;
#pragma unroll
            for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
                for (int ox = -1; ox <= 1; ++ox) {
                    const int nx = x + ox, ny = y + oy;
                    if (nx < 0 || nx >= dim || ny < 0 || ny >= dim) continue;
                    const uint32_t nl = label[ny * dim + nx];
                    if (nl != NONE) m = min(m, nl);
                }
            const uint32_t root = label[m];                              // pointer jump (root <= m)
            if (root != NONE) m = min(m, root);
            if (m < l) { label[i] = (uint16_t)m; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }

    for (int i = threadIdx.x; i < cells; i += 1024) {
        const uint32_t l = label[i];
        if (l != NONE) atomicAdd(&area[l], 1u);
    }
    __syncthreads();
    // winner: max area, ties -> smallest seed index
    unsigned long long k = 0;
    for (int i = threadIdx.x; i < cells; i += 1024) {
        const uint32_t a = area[i];
        if (a) {
            const unsigned long long key = ((unsigned long long)a << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
            k = key > k ? key : k;
        }
    }
    if (k) atomicMax(&best_key, k);
    __syncthreads();
    const unsigned long long bk = best_key;
    const uint32_t barea = (uint32_t)(bk >> 32);
    const uint32_t seed = 0xFFFFFFFFu - (uint32_t)(bk & 0xFFFFFFFFu);
    const bool some = bk != 0 && ((float)barea / (float)cells >= min_size);
    float2* o = out_field + item * cells;
    for (int i = threadIdx.x; i < cells; i += 1024) {
        const bool in = some && label[i] == seed && (uint32_t)i != seed;
        o[i] = in ? f[i] : make_float2(0.0f, 0.0f);
    }
    if (threadIdx.x == 0) {
        out_result[4 * item + 0] = some ? 1 : 0;
        out_result[4 * item + 1] = some ? (int)barea : 0;
        out_result[4 * item + 2] = dim;
        out_result[4 * item + 3] = 0;
    }
}

static int block_dim_host(float min_size, size_t subdivide) {          // lib.rs:53-54, f32 throughout
    const float block_width = sqrtf(min_size) / (float)subdivide;
    const float d = ceilf(1.0f / block_width);
    if (!(d > 0.0f)) return 0;
    if (d > 1.0e9f) return 1000000000;
    return (int)d;
}

int detect_device(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int batch, float min_size, size_t subdivide,
                  float target_motion, int* d_result, float2* d_out_field, int* out_dim) {
    const int dim = block_dim_host(min_size, subdivide);
    if (out_dim) *out_dim = dim;
    OFPS_REQUIRE(ctx, dim >= 1 && dim <= 160, "detect: block_dim %d outside [1,160] (min_size=%g subdivide=%zu)", dim,
                 (double)min_size, subdivide);
    const size_t cells = (size_t)dim * dim;
    auto* d_field = static_cast<float2*>(scratch(ctx, S_FIELD, cells * batch * sizeof(float2)));
    if (!d_field) return OFPS_HIP_ENOMEM;
    int rc = densify_device(ctx, d_entries, n, batch, dim, dim, d_field, nullptr, nullptr, nullptr);
    if (rc != OFPS_HIP_OK) return rc;
    const size_t lds = cells * sizeof(uint32_t) + ((cells * sizeof(uint16_t) + 15) & ~size_t(15));
    if (lds > 48 * 1024)
        OFPS_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(detect_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(detect_kernel, dim3(batch), dim3(1024), lds, ctx->stream, d_field, dim, target_motion, min_size,
                       d_result, d_out_field);
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

}  // namespace ofps

extern "C" {

int ofps_hip_block_dim(float min_size, size_t subdivide) { return ofps::block_dim_host(min_size, subdivide); }

int ofps_hip_detect_dev(ofps_hip_ctx* ctx, const void* d_entries, size_t n_per_item, int batch, float min_size,
                        size_t subdivide, float target_motion, void* d_out_result, void* d_out_field) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_out_result && d_out_field && (d_entries || n_per_item == 0), "detect: null device pointer");
    OFPS_REQUIRE(ctx, batch >= 1, "detect: batch must be >= 1");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ofps::detect_device(ctx, static_cast<const float4*>(d_entries), n_per_item, batch, min_size, subdivide,
                               target_motion, static_cast<int*>(d_out_result), static_cast<float2*>(d_out_field), nullptr);
}

int ofps_hip_detect(ofps_hip_ctx* ctx, const float* entries, size_t n, float min_size, size_t subdivide,
                    float target_motion, int* has_motion, size_t* area, int* dim, float* out_field) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, has_motion && area && dim && out_field && (entries || n == 0), "detect: null host pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int d = ofps::block_dim_host(min_size, subdivide);
    OFPS_REQUIRE(ctx, d >= 1 && d <= 160, "detect: block_dim %d outside [1,160]", d);
    const size_t cells = (size_t)d * d;
    auto* d_ent = static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES, n * sizeof(float4)));
    auto* d_res = static_cast<int*>(ofps::scratch(ctx, ofps::S_RESULT, 4 * sizeof(int)));
    auto* d_out = static_cast<float2*>(ofps::scratch(ctx, ofps::S_BEST, cells * sizeof(float2)));
    if (!d_ent || !d_res || !d_out) return OFPS_HIP_ENOMEM;
    if (n) OFPS_HIP_TRY(ctx, hipMemcpyAsync(d_ent, entries, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    int rc = ofps::detect_device(ctx, d_ent, n, 1, min_size, subdivide, target_motion, d_res, d_out, nullptr);
    if (rc != OFPS_HIP_OK) return rc;
    int res[4] = {0, 0, 0, 0};
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(res, d_res, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_field, d_out, cells * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *has_motion = res[0];
    *area = (size_t)res[1];
    *dim = res[2];
    return OFPS_HIP_OK;
}

}  // extern "C"
