// ctx.hip -- context lifetime, stream selection, device-memory plumbing and the event timer
// behind include/ofps_hip.h.  No compute lives here.
#include "common.hpp"

#include <cstdlib>

static thread_local char g_init_err[512] = {0};

namespace ofps {

int set_error(ofps_hip_ctx* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_init_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_hip(ofps_hip_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return OFPS_HIP_OK;
    int code = (e == hipErrorOutOfMemory) ? OFPS_HIP_ENOMEM : OFPS_HIP_EDEVICE;
    return set_error(ctx, code, "%s failed: %s", what, hipGetErrorString(e));
}

void* scratch(ofps_hip_ctx* ctx, int slot, size_t bytes) {
    auto& s = ctx->scratch[slot];
    if (bytes == 0) bytes = 16;
    if (s.cap >= bytes) return s.p;
    if (s.p) {
        // the old buffer may still be referenced by enqueued work
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { /* fall through to the free */ }
        (void)hipFree(s.p);
        s.p = nullptr; s.cap = 0;
    }
    size_t cap = (bytes + 4095) & ~size_t(4095);
    hipError_t e = hipMalloc(&s.p, cap);
    if (e != hipSuccess) {
        check_hip(ctx, e, "hipMalloc(scratch)");
        s.p = nullptr;
        return nullptr;
    }
    s.cap = cap;
    s.gen += 1;
    return s.p;
}

// One table for the diagnostic switches: the environment variable read once by ofps_hip_init and the name
// ofps_hip_set_option takes are the same string.  value == nullptr or "" restores the default.
int apply_option(ofps_hip_ctx* ctx, const char* name, const char* value, bool from_env) {
    auto& o = ctx->opt;
    const ofps_hip_ctx::Options dflt{};
    const bool unset = !value || !*value;
    const int iv = unset ? 0 : atoi(value);
    if (!strcmp(name, "OFPS_HIP_SAD_KERNEL")) {
        if (!unset && strcmp(value, "block") && strcmp(value, "strip"))
            return set_error(ctx, OFPS_HIP_EINVAL, "%s: '%s' is not block|strip", name, value);
        o.sad_force_block = !unset && !strcmp(value, "block");
    } else if (!strcmp(name, "OFPS_HIP_DENSIFY_NO_SMALL")) {
        o.densify_no_small = unset ? dflt.densify_no_small : (iv != 0);
    } else if (!strcmp(name, "OFPS_HIP_ALMEIDA_PATH")) {
        if (unset) o.almeida_path = 0;
        else if (!strcmp(value, "step")) o.almeida_path = 1;
        else if (!strcmp(value, "wg")) o.almeida_path = 2;
        else if (!strcmp(value, "cluster")) o.almeida_path = 3;
        else return set_error(ctx, OFPS_HIP_EINVAL, "%s: '%s' is not step|wg|cluster", name, value);
    } else if (!strcmp(name, "OFPS_HIP_ALMEIDA_EPT")) {
        if (!unset && iv != 1 && iv != 2 && iv != 4 && iv != 8) return set_error(ctx, OFPS_HIP_EINVAL, "%s: %d is not 1|2|4|8", name, iv);
        o.almeida_ept = unset ? 0 : iv;
    } else if (!strcmp(name, "OFPS_HIP_ALMEIDA_BLOCK")) {
        if (!unset && iv != 256 && iv != 1024) return set_error(ctx, OFPS_HIP_EINVAL, "%s: %d is not 256|1024", name, iv);
        o.almeida_block = unset ? 0 : iv;
    } else if (!strcmp(name, "OFPS_HIP_ALMEIDA_HIER")) {
        if (!unset && (iv < 0 || iv > 2)) return set_error(ctx, OFPS_HIP_EINVAL, "%s: %d is not 0|1|2", name, iv);
        o.almeida_hier = unset ? dflt.almeida_hier : iv;
    } else if (!strcmp(name, "OFPS_HIP_ALMEIDA_FAST")) {
        o.almeida_fast = unset ? -1 : (iv != 0);
    } else if (!strcmp(name, "OFPS_HIP_ALMEIDA_ONE_XCD")) {
        o.almeida_one_xcd = unset ? dflt.almeida_one_xcd : (iv != 0);
    } else if (!strcmp(name, "OFPS_HIP_ALMEIDA_PROF")) {
        o.almeida_prof = unset ? 0 : (iv != 0);
    } else if (!strcmp(name, "OFPS_HIP_LK_PROF")) {
        o.lk_prof = unset ? 0 : (iv != 0);
    } else if (!strcmp(name, "OFPS_HIP_FB_PREPARE_AHEAD")) {
        o.fb_prepare_ahead = unset ? dflt.fb_prepare_ahead : (iv != 0);
    } else if (!strcmp(name, "OFPS_HIP_LK_SERIAL")) {
        o.lk_serial = unset ? 0 : (iv != 0);
    } else if (!strcmp(name, "OFPS_HIP_MULTI_RCCL")) {
        o.multi_rccl = unset ? 0 : (iv != 0);
    } else if (!strcmp(name, "OFPS_HIP_ALMEIDA_TEST_FAULT") || !strcmp(name, "OFPS_HIP_LK_TEST_FALL") ||
               !strcmp(name, "OFPS_HIP_LK_TEST_WAIT_BUDGET") || !strcmp(name, "OFPS_HIP_LK_TEST_ORDER")) {
#ifdef OFPS_HIP_TEST_HOOKS
        if (from_env) return OFPS_HIP_OK;                    // fault injectors are never armed from the environment
        if (!strcmp(name, "OFPS_HIP_ALMEIDA_TEST_FAULT")) o.test_almeida_fault = unset ? 0 : iv;
        else if (!strcmp(name, "OFPS_HIP_LK_TEST_WAIT_BUDGET")) o.test_lk_wait_budget = unset || iv < 0 ? 0 : iv;
        else if (!strcmp(name, "OFPS_HIP_LK_TEST_ORDER")) o.test_lk_order = unset || iv < 0 || iv > 2 ? 0 : iv;
        else o.test_lk_fall = unset ? -1 : iv;
#else
        if (from_env) return OFPS_HIP_OK;
        return set_error(ctx, OFPS_HIP_EUNSUPPORTED, "%s is a test hook: this library was built without OFPS_HIP_TEST_HOOKS", name);
#endif
    } else {
        return set_error(ctx, OFPS_HIP_EINVAL, "unknown option '%s'", name);
    }
    return OFPS_HIP_OK;
}

static const char* const kOptionNames[] = {
    "OFPS_HIP_SAD_KERNEL", "OFPS_HIP_DENSIFY_NO_SMALL", "OFPS_HIP_ALMEIDA_PATH", "OFPS_HIP_ALMEIDA_EPT", "OFPS_HIP_ALMEIDA_BLOCK",
    "OFPS_HIP_ALMEIDA_HIER", "OFPS_HIP_ALMEIDA_FAST", "OFPS_HIP_ALMEIDA_ONE_XCD", "OFPS_HIP_ALMEIDA_PROF", "OFPS_HIP_LK_PROF", "OFPS_HIP_LK_SERIAL", "OFPS_HIP_FB_PREPARE_AHEAD", "OFPS_HIP_MULTI_RCCL"};

}  // namespace ofps

extern "C" {

int ofps_hip_api_version(void) { return OFPS_HIP_API_VERSION; }

int ofps_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int stream_cu_count(ofps_hip_ctx* ctx, hipStream_t s);

int ofps_hip_init(int device, ofps_hip_ctx** out) {
    if (!out) return ofps::set_error(nullptr, OFPS_HIP_EINVAL, "ofps_hip_init: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return ofps::set_error(nullptr, OFPS_HIP_EDEVICE,
                               "ofps_hip_init: no HIP device (%s); this backend has no CPU fallback",
                               e == hipSuccess ? "count == 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return ofps::set_error(nullptr, OFPS_HIP_EINVAL, "ofps_hip_init: device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return ofps::check_hip(nullptr, e, "hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ofps::set_error(nullptr, OFPS_HIP_EUNSUPPORTED,
                               "ofps_hip_init: device %d is %s; this library ships gfx950 code only", device,
                               prop.gcnArchName);
    auto* ctx = new ofps_hip_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    if ((e = hipSetDevice(device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_start)) != hipSuccess || (e = hipEventCreate(&ctx->ev_stop)) != hipSuccess) {
        int rc = ofps::check_hip(nullptr, e, "context setup");
        delete ctx;
        return rc;
    }
    ctx->stream = ctx->own_stream;
    ctx->stream_cus = stream_cu_count(ctx, ctx->stream);     // (HSA_CU_MASK / ROC_GLOBAL_CU_MASK narrow every stream)
    // the only place the library looks at the environment (OFPS_HIP_SAD_PRUNED is the host layers' business)
    for (const char* name : ofps::kOptionNames)
        if (const char* v = getenv(name)) {
            // a malformed A/B variable left in a production environment must not take the backend (with
            // ofps_hip_multi_init: every worker) down: it is reported once and ignored; only ofps_hip_set_option is strict
            if (ofps::apply_option(ctx, name, v, /*from_env=*/true) != OFPS_HIP_OK) {
                fprintf(stderr, "[ofps_hip] warning: ignoring environment variable %s=%s (%s)\n", name, v, ctx->err);
                ctx->err[0] = '\0';
            }
        }
    ofps::cluster_gate_context_created(device);
    *out = ctx;
    return OFPS_HIP_OK;
}

int ofps_hip_set_option(ofps_hip_ctx* ctx, const char* name, const char* value) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, name, "set_option: null name");
    return ofps::apply_option(ctx, name, value, /*from_env=*/false);
}

int ofps_hip_has_test_hooks(void) {
#ifdef OFPS_HIP_TEST_HOOKS
    return 1;
#else
    return 0;
#endif
}

int ofps_hip_almeida_recoveries(ofps_hip_ctx* ctx, uint64_t* count) {
    if (!ctx || !count) return OFPS_HIP_EINVAL;
    *count = 0;
    const void* d = ctx->scratch[ofps::S_ALM_RECOVER].p;
    if (!d) return OFPS_HIP_OK;                              // no cluster launch yet
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long v = 0;
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(&v, d, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *count = v;
    return OFPS_HIP_OK;
}

void ofps_hip_destroy(ofps_hip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& s : ctx->scratch)
        if (s.p) (void)hipFree(s.p);
    for (auto& t : ctx->pipe_ticket) {
        if (t.pinned) (void)hipHostFree(t.pinned);
        if (t.done) (void)hipEventDestroy(t.done);
    }
    for (auto& t : ctx->batch_ticket) {
        if (t.pinned) (void)hipHostFree(t.pinned);
        if (t.done) (void)hipEventDestroy(t.done);
        if (t.uploaded) (void)hipEventDestroy(t.uploaded);
        if (t.prev_copied) (void)hipEventDestroy(t.prev_copied);
    }
    for (int k = 0; k < ofps_hip_ctx::kPipeSlots; ++k) {
        if (ctx->pipe_uploaded[k]) (void)hipEventDestroy(ctx->pipe_uploaded[k]);
        if (ctx->pipe_slot_read[k]) (void)hipEventDestroy(ctx->pipe_slot_read[k]);
    }
    if (ctx->lk_pinned) (void)hipHostFree(ctx->lk_pinned);
    for (auto& t : ctx->lk_ticket) {
        if (t.pinned) (void)hipHostFree(t.pinned);
        if (t.done) (void)hipEventDestroy(t.done);
        if (t.uploaded) (void)hipEventDestroy(t.uploaded);
    }
    if (ctx->lk_copy_stream) (void)hipStreamDestroy(ctx->lk_copy_stream);
    if (ctx->pipe_copy_stream) (void)hipStreamDestroy(ctx->pipe_copy_stream);
    if (ctx->pipe_aux_stream) (void)hipStreamDestroy(ctx->pipe_aux_stream);
    if (ctx->fb_prep_done) (void)hipEventDestroy(ctx->fb_prep_done);
    if (ctx->pipe_fork) (void)hipEventDestroy(ctx->pipe_fork);
    if (ctx->pipe_join) (void)hipEventDestroy(ctx->pipe_join);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    ofps::cluster_gate_context_destroyed(ctx->device);
    delete ctx;
}

const char* ofps_hip_last_error(const ofps_hip_ctx* ctx) { return ctx ? ctx->err : g_init_err; }

// Work enqueued on the old stream (and the frame uploads of the per-frame pipeline that are ordered by it) is finished
// before the context moves to another stream: nothing in the library then depends on cross-stream ordering it did not
// set up itself.
// compute units the stream's kernels may be dealt to; the whole device when the runtime cannot say
static int stream_cu_count(ofps_hip_ctx* ctx, hipStream_t s) {
    uint32_t words[16] = {0};
    if (hipExtStreamGetCUMask(s, 16, words) != hipSuccess) { (void)hipGetLastError(); return ctx->num_cus; }
    int n = 0;
    for (uint32_t w : words) n += __builtin_popcount(w);
    return n > 0 && n < ctx->num_cus ? n : ctx->num_cus;
}

static int switch_stream(ofps_hip_ctx* ctx, hipStream_t next) {
    if (next == ctx->stream) return OFPS_HIP_OK;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pipe_copy_stream) OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->pipe_copy_stream));
    if (ctx->pipe_aux_stream) OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->pipe_aux_stream));
    if (ctx->lk_copy_stream) OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->lk_copy_stream));
    ctx->stream = next;
    ctx->stream_cus = stream_cu_count(ctx, next);
    return OFPS_HIP_OK;
}

int ofps_hip_set_stream(ofps_hip_ctx* ctx, void* hip_stream) {
    if (!ctx) return OFPS_HIP_EINVAL;
    return switch_stream(ctx, reinterpret_cast<hipStream_t>(hip_stream));
}

int ofps_hip_use_own_stream(ofps_hip_ctx* ctx) {
    if (!ctx) return OFPS_HIP_EINVAL;
    return switch_stream(ctx, ctx->own_stream);
}

void* ofps_hip_get_stream(ofps_hip_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }

int ofps_hip_sync(ofps_hip_ctx* ctx) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

int ofps_hip_malloc(ofps_hip_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 16));
    return OFPS_HIP_OK;
}

int ofps_hip_free(ofps_hip_ctx* ctx, void* dptr) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipFree(dptr));
    return OFPS_HIP_OK;
}

// pinned (page-locked) host memory: frames a decoder reads straight into it cross PCIe by DMA without a staging copy
int ofps_hip_host_alloc(ofps_hip_ctx* ctx, size_t bytes, void** hptr) {
    if (!ctx || !hptr) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // fine-grained (coherent) explicitly: callers hand these buffers to push_frame_async as out_entries / out_field, which
    // kernels write directly and the host reads after an event without a system-scope release of its own
    OFPS_HIP_TRY(ctx, hipHostMalloc(hptr, bytes ? bytes : 16, OFPS_HIP_HOST_USER_FLAGS));
    return OFPS_HIP_OK;
}

int ofps_hip_host_free(ofps_hip_ctx* ctx, void* hptr) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipHostFree(hptr));
    return OFPS_HIP_OK;
}

int ofps_hip_memcpy_h2d(ofps_hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

int ofps_hip_memcpy_d2h(ofps_hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OFPS_HIP_OK;
}

int ofps_hip_timer_start(ofps_hip_ctx* ctx) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    return OFPS_HIP_OK;
}

int ofps_hip_timer_stop(ofps_hip_ctx* ctx, float* elapsed_ms) {
    if (!ctx || !elapsed_ms) return OFPS_HIP_EINVAL;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    OFPS_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    OFPS_HIP_TRY(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
    return OFPS_HIP_OK;
}

}  // extern "C"
