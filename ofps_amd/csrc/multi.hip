// multi.hip -- in-process multi-device dispatcher behind the C ABI ("one host thread + one HIP stream per
// device").  A host that binds libofps_hip.so directly -- the Rust shim of INTEGRATION.md, ofps_amd/host/ -- spreads a batch
// of independent frame pairs (BASELINE configs[3]: 64 pairs of 4K frames) over the GPUs of one node without a launcher
// of its own; the reference's threading model is one worker per plugin object, decoders on their own threads
// (ofps-suite/src/app/tracking/worker.rs:251-260,347-352).
//
//   * one ofps_hip_ctx + one worker thread per entry of the device list (a device may be listed more than once: the
//     workers are then independent contexts on one GPU -- how the one-GPU tests exercise the partitioning);
//   * pairs are split into contiguous ranges (first n_pairs % n workers get one more: distributed.pair_range's rule),
//     ref_mode 0 (pair k = frames k, k+1): a worker's range + ONE halo frame; ref_mode 1 (pair k = frames 0, k+1): the key
//     frame is uploaded to the first worker's device once and fanned out device-to-device (hipMemcpyPeerAsync: xGMI
//     point to point, no ring -- RCCL is not linked), every worker holds its own `cur` frames;
//   * no collective on the data path; results come back in pair order into caller memory.
// Frame pairs are independent units, so the workers never talk to each other except for that one fan-out.
#include "common.hpp"

#include <dlfcn.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

struct ofps_hip_multi {
    struct Worker {
        int device = 0;
        ofps_hip_ctx* ctx = nullptr;
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::deque<std::function<void(Worker&)>> jobs;   // FIFO: a worker's context is only ever touched by its own thread
        bool quit = false;
        // resident batch (ofps_hip_multi_stage_frames)
        uint8_t* d_frames = nullptr; size_t frames_cap = 0;
        float* d_out = nullptr; size_t out_cap = 0;
        size_t first_pair = 0, n_pairs = 0, n_res_frames = 0;
        hipEvent_t key_ready = nullptr;        // worker 0: the key frame is on its device
        uint64_t run_gen = 0; int run_block = 0, run_range = 0; size_t run_pairs = 0;   // what d_out holds (ofps_hip_multi_fetch checks it)
        // stream pipeline (ofps_hip_multi_push_frames_async): page-locked staging for batches that arrive in pageable memory
        uint8_t* stage[2] = {nullptr, nullptr}; size_t stage_cap[2] = {0, 0};
        long stream_batches = 0;               // batches this worker has taken
    };
    std::vector<Worker*> w;
    std::recursive_mutex api;                  // the dispatcher's entry points are serialised (they were not re-entrant)
    char err[512] = {0};
    // geometry of the resident batch
    int W = 0, H = 0, dstride = 0, ref_mode = 0;
    size_t pitch = 0, total_pairs = 0;
    bool staged = false;
    uint64_t stage_gen = 0;                    // bumped by every ofps_hip_multi_stage_frames
    // ---- stream pipeline: batches of consecutive frames dealt to the workers round-robin, results back in frame order
    struct StreamTicket {
        bool pending = false;
        int worker = 0, worker_ticket = 0, n = 0, rc = OFPS_HIP_OK;
        bool enqueued = false;                 // the worker has run the push job (rc is valid)
        std::mutex m; std::condition_variable cv;
    };
    std::vector<std::unique_ptr<StreamTicket>> tickets;     // ring of 2 * n_workers
    long next_ticket = 0;
    std::vector<uint8_t*> halo;                // page-locked copies of every batch's last frame (ring of 2 * n_workers + 1)
    size_t halo_bytes = 0;
    int sW = 0, sH = 0;
    long stream_frames = 0;
    // ---- optional RCCL fan-out of the key frame (OFPS_HIP_MULTI_RCCL=1 at ofps_hip_multi_init; north_star: "RCCL broadcast of shared
    // reference frames over xGMI").  librccl.so is dlopen'ed: the library itself keeps linking libamdhip64 only.
    struct Rccl {
        void* lib = nullptr;
        std::vector<void*> comm;               // ncclComm_t per worker (rank = worker index)
        int (*CommInitAll)(void**, int, const int*) = nullptr;
        int (*CommDestroy)(void*) = nullptr;
        int (*GroupStart)() = nullptr;
        int (*GroupEnd)() = nullptr;
        int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
        const char* (*GetErrorString)(int) = nullptr;
        bool active = false;
        unsigned long broadcasts = 0;          // key frames fanned out through ncclBroadcast (diagnostics: ofps_hip_multi_fanout)
    } rccl;
};

static thread_local char g_multi_init_err[512] = {0};

namespace {

// Per-item checksum of device-resident records: the wrapping sum of an item's bytes read as u64 words.  What a host that
// gathers per-pair results across GPUs exchanges instead of the records themselves (bench.py's strong-scaling step: 8 B
// per pair on the wire), and what it compares against the same sum over a CPU result.
__global__ __launch_bounds__(256) void checksum_kernel(const unsigned long long* __restrict__ words, size_t words_per_item,
                                                       unsigned long long* __restrict__ out) {
    __shared__ unsigned long long part[4];
    const size_t item = blockIdx.y;
    const unsigned long long* w = words + item * words_per_item;
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words_per_item; i += (size_t)gridDim.x * 256) acc += w[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + item, part[0] + part[1] + part[2] + part[3]);
}

using Worker = ofps_hip_multi::Worker;

int multi_error(ofps_hip_multi* m, int code, const char* fmt, ...) {
    char* dst = m ? m->err : g_multi_init_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

void worker_main(Worker* w) {
    for (;;) {
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return !w->jobs.empty() || w->quit; });
        if (w->jobs.empty()) return;             // quit, and nothing left to do
        auto job = std::move(w->jobs.front());
        w->jobs.pop_front();
        lk.unlock();
        job(*w);
    }
}

void enqueue(Worker* w, std::function<void(Worker&)> job) {
    std::lock_guard<std::mutex> lk(w->m);
    w->jobs.push_back(std::move(job));
    w->cv.notify_all();
}

// runs job on every worker concurrently (behind whatever each has queued), joins, returns the first failure (message copied
// from that worker's context)
int run_all(ofps_hip_multi* m, const std::function<int(Worker&, int)>& job) {
    struct Join { std::mutex m; std::condition_variable cv; size_t left; std::vector<int> rc; } j;
    j.left = m->w.size(); j.rc.assign(m->w.size(), OFPS_HIP_OK);
    for (size_t k = 0; k < m->w.size(); ++k) {
        const int idx = (int)k;
        enqueue(m->w[k], [&j, &job, idx](Worker& ww) {
            const int rc = job(ww, idx);
            std::lock_guard<std::mutex> lk(j.m);
            j.rc[idx] = rc;
            if (--j.left == 0) j.cv.notify_all();
        });
    }
    { std::unique_lock<std::mutex> lk(j.m); j.cv.wait(lk, [&] { return j.left == 0; }); }
    for (size_t k = 0; k < m->w.size(); ++k)
        if (j.rc[k] != OFPS_HIP_OK) {
            snprintf(m->err, sizeof(m->err), "worker %zu (device %d): %s", k, m->w[k]->device, ofps_hip_last_error(m->w[k]->ctx));
            return j.rc[k];
        }
    return OFPS_HIP_OK;
}

int grow(Worker& w, uint8_t** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return OFPS_HIP_OK;
    if (*p) { OFPS_HIP_TRY(w.ctx, hipStreamSynchronize(w.ctx->stream)); OFPS_HIP_TRY(w.ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
    OFPS_HIP_TRY(w.ctx, hipMalloc(reinterpret_cast<void**>(p), bytes ? bytes : 16));
    *cap = bytes;
    return OFPS_HIP_OK;
}

}  // namespace

extern "C" {

int ofps_hip_checksum_dev(ofps_hip_ctx* ctx, const void* d_data, size_t bytes_per_item, int batch, void* d_out_u64) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_out_u64 && (d_data || bytes_per_item == 0) && batch >= 0 && batch <= 65535, "checksum: bad arguments");
    OFPS_REQUIRE(ctx, bytes_per_item % 8 == 0 && (reinterpret_cast<uintptr_t>(d_data) & 7) == 0, "checksum: items must be whole, 8-byte aligned u64 words");
    if (batch == 0) return OFPS_HIP_OK;
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    OFPS_HIP_TRY(ctx, hipMemsetAsync(d_out_u64, 0, (size_t)batch * sizeof(unsigned long long), ctx->stream));
    const size_t words = bytes_per_item / 8;
    if (words == 0) return OFPS_HIP_OK;
    unsigned gx = (unsigned)((words + 4 * 256 - 1) / (4 * 256));
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(checksum_kernel, dim3(gx ? gx : 1, batch), dim3(256), 0, ctx->stream, static_cast<const unsigned long long*>(d_data), words,
                       static_cast<unsigned long long*>(d_out_u64));
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

void ofps_hip_multi_pair_range(size_t n_pairs, int n_workers, int k, size_t* first, size_t* count) {
    size_t f = 0, c = 0;
    if (n_workers >= 1 && k >= 0 && k < n_workers) {
        const size_t base = n_pairs / (size_t)n_workers, extra = n_pairs % (size_t)n_workers;
        c = base + ((size_t)k < extra ? 1 : 0);
        f = (size_t)k * base + ((size_t)k < extra ? (size_t)k : extra);
    }
    if (first) *first = f;
    if (count) *count = c;
}

void ofps_hip_multi_frame_range(size_t n_pairs, int n_workers, int k, int ref_mode, size_t* first_frame, size_t* n_frames) {
    size_t f = 0, c = 0;
    ofps_hip_multi_pair_range(n_pairs, n_workers, k, &f, &c);
    size_t ff = f, fc = 0;
    if (c) {
        if (ref_mode == 0) { ff = f; fc = c + 1; }        // count pairs need count + 1 frames: one halo frame shared with the next worker
        else { ff = f + 1; fc = c; }                      // the key frame (frame 0) arrives by the fan-out
    }
    if (first_frame) *first_frame = ff;
    if (n_frames) *n_frames = fc;
}

int ofps_hip_multi_init(const int* devices, int n, ofps_hip_multi** out) {
    if (!out) return multi_error(nullptr, OFPS_HIP_EINVAL, "multi_init: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return multi_error(nullptr, OFPS_HIP_EINVAL, "multi_init: need 1..64 devices");
    auto* m = new ofps_hip_multi();
    for (int k = 0; k < n; ++k) {
        auto* w = new Worker();
        w->device = devices[k];
        const int rc = ofps_hip_init(devices[k], &w->ctx);
        if (rc != OFPS_HIP_OK) {
            snprintf(g_multi_init_err, sizeof(g_multi_init_err), "multi_init: device %d: %s", devices[k], ofps_hip_last_error(nullptr));
            delete w;
            ofps_hip_multi_destroy(m);
            return rc;
        }
        m->w.push_back(w);
    }
    // peer access towards the first worker's device (the key-frame fan-out); failure is not fatal: hipMemcpyPeerAsync
    // stages through the host when there is no direct path.  (The calling thread's current device is put back afterwards:
    // a library call must not leave it changed.)
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    for (size_t k = 1; k < m->w.size(); ++k) {
        if (m->w[k]->device == m->w[0]->device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->w[k]->device, m->w[0]->device) == hipSuccess && can) {
            (void)hipSetDevice(m->w[k]->device);
            const hipError_t e = hipDeviceEnablePeerAccess(m->w[0]->device, 0);
            if (e != hipSuccess) (void)hipGetLastError();          // already enabled / unsupported: ignore
        }
    }
    // OFPS_HIP_MULTI_RCCL=1: one RCCL communicator over the workers' devices (ncclCommInitAll: one process, one rank per device) for the
    // key-frame fan-out.  Needs distinct devices (RCCL refuses a device twice in a communicator: workers sharing a GPU keep the copy path)
    // and librccl.so at run time; anything missing is not an error -- the peer-copy path is the default and always there.
    if (m->w[0]->ctx->opt.multi_rccl) {                       // (read from the environment by ofps_hip_init, like every switch)
        bool distinct = true;
        for (size_t a = 0; a < m->w.size(); ++a)
            for (size_t b = a + 1; b < m->w.size(); ++b) distinct = distinct && m->w[a]->device != m->w[b]->device;
        auto& R = m->rccl;
        if (distinct) R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!R.lib && distinct) R.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (R.lib) {
            R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(dlsym(R.lib, "ncclCommInitAll"));
            R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.lib, "ncclCommDestroy"));
            R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(dlsym(R.lib, "ncclGroupStart"));
            R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(dlsym(R.lib, "ncclGroupEnd"));
            R.Broadcast = reinterpret_cast<decltype(R.Broadcast)>(dlsym(R.lib, "ncclBroadcast"));
            R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.lib, "ncclGetErrorString"));
            if (R.CommInitAll && R.CommDestroy && R.GroupStart && R.GroupEnd && R.Broadcast) {
                std::vector<int> devs;
                for (auto* w : m->w) devs.push_back(w->device);
                R.comm.assign(m->w.size(), nullptr);
                const int e = R.CommInitAll(R.comm.data(), (int)devs.size(), devs.data());
                if (e == 0) R.active = true;
                else {
                    snprintf(m->err, sizeof(m->err), "multi_init: ncclCommInitAll failed (%s): key frames go by peer copies",
                             R.GetErrorString ? R.GetErrorString(e) : "?");
                    R.comm.clear();
                }
            }
        }
    }
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    for (auto* w : m->w) w->th = std::thread(worker_main, w);
    *out = m;
    return OFPS_HIP_OK;
}

// 0 = key frames travel by hipMemcpyPeerAsync (the default), 1 = by ncclBroadcast over the communicator made at init; *broadcasts (may be
// NULL): how many key frames went that way
int ofps_hip_multi_fanout(const ofps_hip_multi* m, uint64_t* broadcasts) {
    if (!m) return OFPS_HIP_EINVAL;
    if (broadcasts) *broadcasts = m->rccl.broadcasts;
    return m->rccl.active ? 1 : 0;
}

void ofps_hip_multi_destroy(ofps_hip_multi* m) {
    if (!m) return;
    // the communicator first: its ranks were given the workers' streams for the broadcasts (workers are idle or about to be joined; every
    // broadcast was synchronised where it was issued)
    if (m->rccl.active) {
        for (size_t k = 0; k < m->rccl.comm.size(); ++k)
            if (m->rccl.comm[k]) { (void)hipSetDevice(m->w[k]->device); (void)m->rccl.CommDestroy(m->rccl.comm[k]); }
        m->rccl.active = false;
    }
    for (auto* w : m->w) {
        if (w->th.joinable()) {
            { std::lock_guard<std::mutex> lk(w->m); w->quit = true; w->cv.notify_all(); }
            w->th.join();
        }
        if (w->ctx) {
            (void)hipSetDevice(w->device);
            (void)hipDeviceSynchronize();
            if (w->d_frames) (void)hipFree(w->d_frames);
            if (w->d_out) (void)hipFree(w->d_out);
            for (auto* st : w->stage) if (st) (void)hipHostFree(st);
            if (w->key_ready) (void)hipEventDestroy(w->key_ready);
            ofps_hip_destroy(w->ctx);
        }
        delete w;
    }
    for (auto* h : m->halo) if (h) (void)hipHostFree(h);
    if (m->rccl.lib) (void)dlclose(m->rccl.lib);
    delete m;
}

const char* ofps_hip_multi_last_error(const ofps_hip_multi* m) { return m ? m->err : g_multi_init_err; }
int ofps_hip_multi_worker_count(const ofps_hip_multi* m) { return m ? (int)m->w.size() : 0; }

// Uploads a batch and leaves it resident: worker k gets the frames of its pair range (ofps_hip_multi_frame_range).
int ofps_hip_multi_stage_frames(ofps_hip_multi* m, const uint8_t* frames, int n_frames, int W, int H, int stride, size_t frame_pitch,
                                int ref_mode) {
    if (!m) return OFPS_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> api(m->api);
    if (!frames || n_frames < 1 || W < 1 || H < 1 || stride < W || frame_pitch < (size_t)stride * H || (ref_mode != 0 && ref_mode != 1))
        return multi_error(m, OFPS_HIP_EINVAL, "multi_stage_frames: bad arguments (n_frames=%d W=%d H=%d stride=%d ref_mode=%d)", n_frames, W, H,
                           stride, ref_mode);
    const int n = (int)m->w.size();
    m->W = W; m->H = H; m->dstride = (W + 63) & ~63; m->pitch = (size_t)m->dstride * H; m->ref_mode = ref_mode;
    m->total_pairs = (size_t)n_frames - 1;
    m->staged = false;
    m->stage_gen += 1;                         // whatever the workers' result buffers hold belongs to an older batch now
    // phase 1: every worker uploads its own frames; worker 0 also the key frame (slot 0 of its buffer) in key mode
    int rc = run_all(m, [&](Worker& w, int k) -> int {
        OFPS_HIP_TRY(w.ctx, hipSetDevice(w.device));
        size_t ff = 0, fc = 0;
        ofps_hip_multi_pair_range(m->total_pairs, n, k, &w.first_pair, &w.n_pairs);
        ofps_hip_multi_frame_range(m->total_pairs, n, k, ref_mode, &ff, &fc);
        const size_t slots = fc + ((ref_mode == 1 && fc) ? 1 : 0);          // key mode: slot 0 holds the key frame
        w.n_res_frames = slots;
        int r = grow(w, &w.d_frames, &w.frames_cap, (slots ? slots : 1) * m->pitch);
        if (r != OFPS_HIP_OK) return r;
        if (!w.key_ready) OFPS_HIP_TRY(w.ctx, hipEventCreateWithFlags(&w.key_ready, hipEventDisableTiming));
        hipStream_t s = w.ctx->stream;
        const size_t base_slot = (ref_mode == 1 && fc) ? 1 : 0;
        for (size_t j = 0; j < fc; ++j)
            OFPS_HIP_TRY(w.ctx, ofps::upload_rows(w.d_frames + (base_slot + j) * m->pitch, m->dstride, frames + (ff + j) * frame_pitch, stride, W,
                                                  H, s));
        if (ref_mode == 1 && k == 0) {
            if (!fc) { r = grow(w, &w.d_frames, &w.frames_cap, m->pitch); if (r != OFPS_HIP_OK) return r; }
            OFPS_HIP_TRY(w.ctx, ofps::upload_rows(w.d_frames, m->dstride, frames, stride, W, H, s));
            OFPS_HIP_TRY(w.ctx, hipEventRecord(w.key_ready, s));
        }
        OFPS_HIP_TRY(w.ctx, hipStreamSynchronize(s));
        return OFPS_HIP_OK;
    });
    if (rc != OFPS_HIP_OK) return rc;
    // phase 2 (key mode), RCCL form: ONE ncclBroadcast of the key frame from worker 0's device inside a group call, each rank on its worker's
    // stream (the single-process multi-device idiom; the workers are idle: phase 1 has been synchronised).  Also with one worker (a
    // communicator of one rank: the broadcast is a self-copy in place) so that the path can be exercised on a one-GPU box.
    if (ref_mode == 1 && m->rccl.active) {
        auto& R = m->rccl;
        int e = R.GroupStart();
        for (int k = 0; k < n && e == 0; ++k) {
            Worker& w = *m->w[(size_t)k];
            if (k != 0 && !w.n_pairs) continue;
            e = R.Broadcast(w.d_frames, w.d_frames, m->pitch, /*ncclUint8*/ 1, /*root*/ 0, R.comm[(size_t)k], w.ctx->stream);
        }
        const int e2 = R.GroupEnd();
        if (e != 0 || e2 != 0)
            return multi_error(m, OFPS_HIP_EDEVICE, "multi_stage_frames: ncclBroadcast of the key frame failed (%s)",
                               R.GetErrorString ? R.GetErrorString(e ? e : e2) : "?");
        for (int k = 0; k < n; ++k) {
            Worker& w = *m->w[(size_t)k];
            if (hipSetDevice(w.device) != hipSuccess || hipStreamSynchronize(w.ctx->stream) != hipSuccess)
                return multi_error(m, OFPS_HIP_EDEVICE, "multi_stage_frames: synchronising worker %d after the broadcast failed", k);
        }
        R.broadcasts += 1;
    } else
    // phase 2 (key mode): the key frame travels device to device from worker 0's copy -- one xGMI hop per receiver
    if (ref_mode == 1 && n > 1) {
        Worker* w0 = m->w[0];
        rc = run_all(m, [&](Worker& w, int k) -> int {
            if (k == 0 || !w.n_pairs) return OFPS_HIP_OK;
            OFPS_HIP_TRY(w.ctx, hipSetDevice(w.device));
            hipStream_t s = w.ctx->stream;
            OFPS_HIP_TRY(w.ctx, hipStreamWaitEvent(s, w0->key_ready, 0));
            if (w.device == w0->device)
                OFPS_HIP_TRY(w.ctx, hipMemcpyAsync(w.d_frames, w0->d_frames, m->pitch, hipMemcpyDeviceToDevice, s));
            else
                OFPS_HIP_TRY(w.ctx, hipMemcpyPeerAsync(w.d_frames, w.device, w0->d_frames, w0->device, m->pitch, s));
            OFPS_HIP_TRY(w.ctx, hipStreamSynchronize(s));
            return OFPS_HIP_OK;
        });
        if (rc != OFPS_HIP_OK) return rc;
    }
    m->staged = true;
    return OFPS_HIP_OK;
}

// `steps` searches of every worker's resident pairs, back to back on its stream; returns when all workers are through.
int ofps_hip_multi_run_resident(ofps_hip_multi* m, int block, int range, int steps, float* worker_ms) {
    if (!m) return OFPS_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> api(m->api);
    if (!m->staged) return multi_error(m, OFPS_HIP_EINVAL, "multi_run_resident: no staged batch");
    if (steps < 1) return multi_error(m, OFPS_HIP_EINVAL, "multi_run_resident: steps must be >= 1");
    const size_t nblk = ofps_hip_sad_block_count(m->W, m->H, block);
    return run_all(m, [&](Worker& w, int k) -> int {
        if (worker_ms) worker_ms[k] = 0.0f;
        if (!w.n_pairs) return OFPS_HIP_OK;
        OFPS_HIP_TRY(w.ctx, hipSetDevice(w.device));
        uint8_t* out8 = reinterpret_cast<uint8_t*>(w.d_out);
        size_t cap = w.out_cap;
        int r = grow(w, &out8, &cap, w.n_pairs * nblk * 4 * sizeof(float));
        w.d_out = reinterpret_cast<float*>(out8); w.out_cap = cap;
        if (r != OFPS_HIP_OK) return r;
        w.run_gen = m->stage_gen; w.run_block = block; w.run_range = range; w.run_pairs = w.n_pairs;       // what d_out is about to hold
        if (worker_ms) { r = ofps_hip_timer_start(w.ctx); if (r != OFPS_HIP_OK) return r; }
        for (int sidx = 0; sidx < steps; ++sidx) {
            r = ofps_hip_sad_flow_dev(w.ctx, w.d_frames, (int)w.n_res_frames, m->W, m->H, m->dstride, m->pitch, m->ref_mode, block, range,
                                      w.d_out, nullptr);
            if (r != OFPS_HIP_OK) return r;
        }
        if (worker_ms) return ofps_hip_timer_stop(w.ctx, &worker_ms[k]);      // HIP events on the worker's stream; synchronises
        OFPS_HIP_TRY(w.ctx, hipStreamSynchronize(w.ctx->stream));
        return OFPS_HIP_OK;
    });
}

// results of the last run, in pair order: out_entries [(n_frames - 1) * nblk * 4] floats
int ofps_hip_multi_fetch(ofps_hip_multi* m, int block, float* out_entries) {
    if (!m || !out_entries) return OFPS_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> api(m->api);
    if (!m->staged) return multi_error(m, OFPS_HIP_EINVAL, "multi_fetch: no staged batch");
    const size_t nblk = ofps_hip_sad_block_count(m->W, m->H, block);
    return run_all(m, [&](Worker& w, int) -> int {
        if (!w.n_pairs) return OFPS_HIP_OK;
        // the results must be those of THIS staged batch, searched with this block size (a re-stage without a
        // new run, or a run with another block size that happened to fit the buffer, used to return stale vectors)
        if (!w.d_out || w.run_gen != m->stage_gen || w.run_block != block || w.run_pairs != w.n_pairs)
            return ofps::set_error(w.ctx, OFPS_HIP_EINVAL, "multi_fetch: no results for the staged batch (run_resident after stage_frames, same block size)");
        OFPS_HIP_TRY(w.ctx, hipSetDevice(w.device));
        OFPS_HIP_TRY(w.ctx, hipMemcpyAsync(out_entries + w.first_pair * nblk * 4, w.d_out, w.n_pairs * nblk * 4 * sizeof(float),
                                           hipMemcpyDeviceToHost, w.ctx->stream));
        OFPS_HIP_TRY(w.ctx, hipStreamSynchronize(w.ctx->stream));
        return OFPS_HIP_OK;
    });
}

// ---- stream pipeline across devices.  Batches of n consecutive frames of ONE stream are dealt to the workers round-robin (batch g
// -> worker g % n_workers); every batch travels with the one frame in front of it (the last frame of batch g - 1, kept in a
// page-locked halo buffer), so the workers never talk to each other; per worker the batch goes through the context's
// two-ticket batched read-ahead (pipeline.hip: one H2D on the copy stream, one search launch, one detector chain, one estimator
// launch, read-back by kernel into the ticket's page-locked block).  Results come back per batch, i.e. in frame order.  The
// reference decodes on its own thread into a double buffer and runs its estimators beside it
// (ofps-suite/src/app/tracking/worker.rs:165-226,347-361; detection.rs:111-148); it has no multi-device code.
void ofps_hip_multi_stream_plan(long batch, int n_workers, int* worker, int* halo_slot, int* ticket_slot) {
    const int nw = n_workers < 1 ? 1 : n_workers;
    if (worker) *worker = (int)(batch % nw);
    if (halo_slot) *halo_slot = (int)(batch % (2 * nw + 1));       // the frame in front of batch g lives in halo[g % (2 nw + 1)]
    if (ticket_slot) *ticket_slot = (int)(batch % (2 * nw));       // at most two batches in flight per worker
}

int ofps_hip_multi_reset_frames(ofps_hip_multi* m) {
    if (!m) return OFPS_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> api(m->api);
    for (auto& tp : m->tickets)
        if (tp && tp->pending) return multi_error(m, OFPS_HIP_EINVAL, "multi_reset_frames: ticket(s) in flight: collect them first");
    const int rc = run_all(m, [&](Worker& w, int) -> int { w.stream_batches = 0; return ofps_hip_reset_frames(w.ctx); });
    m->stream_frames = 0; m->next_ticket = 0; m->sW = m->sH = 0;
    return rc;
}

int ofps_hip_multi_push_frames_async(ofps_hip_multi* m, const uint8_t* frames, int n, int W, int H, int stride, size_t frame_pitch,
                                     const ofps_hip_frame_params* params, float* out_entries, int* ticket) {
    if (!m) return OFPS_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> api(m->api);
    if (!frames || !params || !ticket || n < 1 || n > 4096 || W < 1 || H < 1 || stride < W || frame_pitch < (size_t)stride * H)
        return multi_error(m, OFPS_HIP_EINVAL, "multi_push_frames_async: bad arguments (n=%d W=%d H=%d stride=%d)", n, W, H, stride);
    const int nw = (int)m->w.size();
    if (m->tickets.empty())
        for (int k = 0; k < 2 * nw; ++k) m->tickets.emplace_back(new ofps_hip_multi::StreamTicket());
    if (W != m->sW || H != m->sH) {                               // a new geometry restarts the stream (decoder.rs:66-72)
        for (auto& tp : m->tickets)
            if (tp->pending) return multi_error(m, OFPS_HIP_EINVAL, "multi_push_frames_async: geometry change with tickets in flight");
        const int rc = ofps_hip_multi_reset_frames(m);
        if (rc != OFPS_HIP_OK) return rc;
        m->sW = W; m->sH = H;
    }
    const long g = m->next_ticket;
    int wk, hs, ts;
    ofps_hip_multi_stream_plan(g, nw, &wk, &hs, &ts);
    auto* t = m->tickets[ts].get();
    if (t->pending)
        return multi_error(m, OFPS_HIP_EINVAL, "multi_push_frames_async: ticket %ld has not been collected (at most %d batches in flight)",
                           g - 2 * nw, 2 * nw);
    // halo buffers: page-locked, one frame each, stored DENSELY (W bytes per row): the batch that consumes a halo may arrive with
    // another row stride than the batch that produced it (it used to be saved with the producer's stride and uploaded
    // with the consumer's), and the buffers no longer grow -- and restart the stream -- when only the stride does
    const size_t hbytes = (size_t)W * H;
    if (m->halo.empty() || m->halo_bytes < hbytes) {
        for (auto& tp : m->tickets)
            if (tp->pending) return multi_error(m, OFPS_HIP_EINVAL, "multi_push_frames_async: frame size grew with tickets in flight");
        for (auto* h : m->halo) if (h) (void)hipHostFree(h);
        m->halo.assign((size_t)(2 * nw + 1), nullptr);
        for (auto& h : m->halo)
            if (hipHostMalloc(reinterpret_cast<void**>(&h), hbytes, hipHostMallocDefault) != hipSuccess)
                return multi_error(m, OFPS_HIP_ENOMEM, "multi_push_frames_async: hipHostMalloc(%zu) failed", hbytes);
        m->halo_bytes = hbytes;
        m->stream_frames = 0;                                      // whatever halo there was is gone: the stream starts over
    }
    const uint8_t* halo = m->stream_frames ? m->halo[hs] : nullptr;    // the stream's first batch has no frame in front of it
    // the last frame of THIS batch is the halo of the next one (copied now: the caller's buffer is only promised until its wait)
    int nhs;
    ofps_hip_multi_stream_plan(g + 1, nw, nullptr, &nhs, nullptr);
    {
        const uint8_t* last = frames + (size_t)(n - 1) * frame_pitch;
        if (stride == W) memcpy(m->halo[nhs], last, hbytes);
        else for (int y = 0; y < H; ++y) memcpy(m->halo[nhs] + (size_t)y * W, last + (size_t)y * stride, (size_t)W);
    }
    t->pending = true; t->enqueued = false; t->worker = wk; t->n = n; t->rc = OFPS_HIP_OK;
    const ofps_hip_frame_params prm = *params;
    enqueue(m->w[wk], [=](Worker& w) {
        int rc = OFPS_HIP_OK, wt = 0;
        const uint8_t* src = frames;
        size_t pitch = frame_pitch;
        void* dev = nullptr;
        if (!ofps::device_address_of(frames, &dev)) {
            // pageable source: into this worker's page-locked staging (two buffers in turn, like its two tickets) -- the copy
            // runs on the worker's thread, beside the other workers' copies
            const int sb = (int)(w.stream_batches & 1);
            const size_t need = (size_t)n * frame_pitch;
            if (w.stage_cap[sb] < need) {
                if (w.stage[sb]) (void)hipHostFree(w.stage[sb]);
                w.stage[sb] = nullptr; w.stage_cap[sb] = 0;
                if (hipHostMalloc(reinterpret_cast<void**>(&w.stage[sb]), need, hipHostMallocDefault) != hipSuccess)
                    rc = ofps::set_error(w.ctx, OFPS_HIP_ENOMEM, "multi_push_frames_async: hipHostMalloc(%zu) for staging failed", need);
                else w.stage_cap[sb] = need;
            }
            if (rc == OFPS_HIP_OK) { memcpy(w.stage[sb], frames, need); src = w.stage[sb]; }
        }
        if (rc == OFPS_HIP_OK)
            rc = ofps::push_frames_impl(w.ctx, src, n, W, H, stride, pitch, &prm, out_entries, &wt, /*halo_mode=*/1, halo, /*halo_stride=*/W);
        w.stream_batches += 1;
        std::lock_guard<std::mutex> lk(t->m);
        t->rc = rc; t->worker_ticket = wt; t->enqueued = true;
        t->cv.notify_all();
    });
    m->stream_frames += n;
    m->next_ticket = g + 1;
    *ticket = (int)(g & 0x7FFFFFFF);
    return OFPS_HIP_OK;
}

int ofps_hip_multi_frames_wait(ofps_hip_multi* m, int ticket, ofps_hip_frame_result* out) {
    if (!m || !out) return OFPS_HIP_EINVAL;
    ofps_hip_multi::StreamTicket* t = nullptr;
    {
        std::lock_guard<std::recursive_mutex> api(m->api);
        const int nw = (int)m->w.size();
        const long newest = m->next_ticket - 1;
        long g = -1;
        for (long k = newest; k >= 0 && k > newest - 2 * nw; --k)
            if ((int)(k & 0x7FFFFFFF) == ticket) { g = k; break; }
        if (g < 0 || m->tickets.empty()) return multi_error(m, OFPS_HIP_EINVAL, "multi_frames_wait: ticket %d is not in flight", ticket);
        int ts;
        ofps_hip_multi_stream_plan(g, nw, nullptr, nullptr, &ts);
        t = m->tickets[ts].get();
        if (!t->pending) return multi_error(m, OFPS_HIP_EINVAL, "multi_frames_wait: ticket %d was already collected", ticket);
    }
    // (outside the dispatcher's lock: another host thread may push the next batch while this one waits)
    { std::unique_lock<std::mutex> lk(t->m); t->cv.wait(lk, [&] { return t->enqueued; }); }
    Worker* w = m->w[t->worker];
    int rc = t->rc;
    if (rc == OFPS_HIP_OK) {
        struct Done { std::mutex m; std::condition_variable cv; bool done = false; int rc = 0; } d;
        const int wt = t->worker_ticket;
        enqueue(w, [&d, wt, out](Worker& ww) {
            const int r = ofps_hip_frames_wait(ww.ctx, wt, out);
            std::lock_guard<std::mutex> lk(d.m);
            d.rc = r; d.done = true; d.cv.notify_all();
        });
        std::unique_lock<std::mutex> lk(d.m);
        d.cv.wait(lk, [&] { return d.done; });
        rc = d.rc;
    }
    std::lock_guard<std::recursive_mutex> api(m->api);
    if (rc != OFPS_HIP_OK) snprintf(m->err, sizeof(m->err), "worker %d (device %d): %s", t->worker, w->device, ofps_hip_last_error(w->ctx));
    t->pending = false;
    return rc;
}

int ofps_hip_multi_sad_flow(ofps_hip_multi* m, const uint8_t* frames, int n_frames, int W, int H, int stride, size_t frame_pitch,
                            int ref_mode, int block, int range, float* out_entries) {
    if (!m) return OFPS_HIP_EINVAL;
    std::lock_guard<std::recursive_mutex> api(m->api);
    if (!out_entries) return multi_error(m, OFPS_HIP_EINVAL, "multi_sad_flow: out_entries is NULL");
    int rc = ofps_hip_multi_stage_frames(m, frames, n_frames, W, H, stride, frame_pitch, ref_mode);
    if (rc != OFPS_HIP_OK) return rc;
    if (n_frames < 2) return OFPS_HIP_OK;
    rc = ofps_hip_multi_run_resident(m, block, range, 1, nullptr);
    if (rc != OFPS_HIP_OK) return rc;
    return ofps_hip_multi_fetch(m, block, out_entries);
}

}  // extern "C"
