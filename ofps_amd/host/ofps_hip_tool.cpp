// ofps_hip_tool -- small CLI over the C++ host layer; the counterparts of the reference's non-GUI callers:
//   extract <decoder> <arg> <out.mvec> [max_frames [Name=value[@frame] ...]]     motion-extract/src/main.rs:7-38 (decode -> .mvec); properties set by name
//   detect  <decoder> <arg> [max_frames]                detection loop, ofps-suite/src/app/detection.rs:92-168
//   detect  --config <saved.json> [--perf-csv <dir>] [max_frames]
//                                                       the same loop from a saved MotionDetectionConfig (detection.rs:45-50):
//                                                       plugins, their saved properties, max_frame_gap / min_frames
//   parse-config <saved.json>                           prints what a saved configuration says (no GPU)
//   stream-bench <w> <h> <frames> [sync|ahead|batch <n>|multi <n> [dev ...]] [--block B --range R  (multi; default 16 16)]
//                                                       PCIe-inclusive per-frame time of the hip_sad process_frame shape; `multi`:
//                                                       the stream over several workers (ofps_hip_multi_push_frames_async), with
//                                                       detector + estimator per frame
//   track   <decoder> <arg> [aspect fov_y] [lsq|ransac] tracking loop, ofps-suite/src/app/tracking/worker.rs:305-412
//   mvec-copy <in.mvec> <out.mvec>                      CPU-only .mvec round trip (reader + writer)
//   multi-extract <raw.y> <w> <h> <out.mvec> [dev ...]  the whole clip over several GPUs (ofps_hip_multi_*): same bytes as `extract hip_sad`
// decoder = hip_sad / hip_lk / hip_flow ("<input>?w=..&h=..&fps=..") or mvec ("<input>"); <input> is a file path, "tcp://host:port"
// (connect) or "tcp://@:port" (listen, accept one connection) as in ofps/src/utils.rs:92-118.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <iostream>
#include <sstream>

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "ofps_host.hpp"

using namespace ofps;

static double mean(const std::vector<double>& v) {
    double s = 0;
    for (double x : v) s += x;
    return v.empty() ? 0.0 : s / (double)v.size();
}

int main(int argc, char** argv) {
    try {
        const std::string cmd = argc > 1 ? argv[1] : "";
        if (cmd == "mvec-copy" && argc >= 4) {
            MvecFileDecoder in(std::make_unique<std::ifstream>(argv[2], std::ios::binary));
            std::ofstream out(argv[3], std::ios::binary);
            MotionVectors mv;
            size_t frames = 0, total = 0;
            for (;;) {
                mv.clear();
                try { in.process_frame(mv, nullptr, nullptr, 0); } catch (const Error&) { break; }
                write_mvec_frame(out, mv);
                ++frames; total += mv.size();
            }
            std::printf("{\"frames\": %zu, \"vectors\": %zu}\n", frames, total);
            return 0;
        }
        if (cmd == "extract" && argc >= 5) {
            auto dec = create_decoder(argv[2], argv[3]);
            std::ofstream out(argv[4], std::ios::binary);
            const size_t max_frames = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : SIZE_MAX;
            // "Name=value[@k]": the property is set (Properties::set_prop, plugins/properties.rs:6-18) before frame k is read (default 0) --
            // what the GUI's property panel does to a running decoder (ofps-suite/src/app/widgets.rs)
            struct PropArg { std::string name, value; size_t at; };
            std::vector<PropArg> prop_args;
            for (int i = 6; i < argc; ++i) {
                std::string a = argv[i];
                const auto eq = a.find('=');
                if (eq == std::string::npos) throw Error("extract: expected Name=value[@frame], got " + a);
                PropArg pa{a.substr(0, eq), a.substr(eq + 1), 0};
                if (const auto at = pa.value.rfind('@'); at != std::string::npos) { pa.at = std::stoul(pa.value.substr(at + 1)); pa.value = pa.value.substr(0, at); }
                prop_args.push_back(pa);
            }
            auto apply_props = [&](size_t frame) {
                for (const auto& pa : prop_args) {
                    if (pa.at != frame) continue;
                    bool found = false;
                    for (const auto& [n, cur] : dec->props()) {
                        if (n != pa.name) continue;
                        found = true;
                        Property v = cur;
                        if (std::holds_alternative<bool>(cur)) v = (pa.value == "true" || pa.value == "1");
                        else if (auto* f = std::get_if<BoundedProp<float>>(&v)) f->val = std::stof(pa.value);
                        else if (auto* u = std::get_if<BoundedProp<size_t>>(&v)) u->val = std::stoul(pa.value);
                        else v = pa.value;
                        dec->set_prop(n, v);
                    }
                    if (!found) throw Error("extract: the decoder has no property \"" + pa.name + "\"");
                }
            };
            MotionVectors mv;
            size_t frames = 0, total = 0;
            while (frames < max_frames) {                              // while c.process_frame(..).is_ok()
                mv.clear();
                apply_props(frames);
                try { dec->process_frame(mv, nullptr, nullptr, 0); } catch (const Error&) { break; }
                write_mvec_frame(out, mv);                             // a frame without vectors is written with count 0
                ++frames; total += mv.size();
            }
            std::printf("{\"frames\": %zu, \"vectors\": %zu}\n", frames, total);
            return 0;
        }
        if (cmd == "multi-extract" && argc >= 6) {
            // the whole clip at once over several GPUs (or several workers on one): same .mvec as `extract hip_sad`
            const size_t W = std::strtoull(argv[3], nullptr, 10), H = std::strtoull(argv[4], nullptr, 10);
            std::ifstream in(argv[2], std::ios::binary);
            if (!in) throw Error(std::string("cannot open ") + argv[2]);
            std::vector<uint8_t> clip((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
            const size_t n_frames = clip.size() / (W * H);
            std::vector<int> devices;
            for (int a = 6; a < argc; ++a) devices.push_back(std::atoi(argv[a]));
            if (devices.empty()) devices.push_back(0);
            MultiDeviceSad md(devices);
            const auto pairs = md.search(clip.data(), n_frames, W, H, 16, 16);
            std::ofstream out(argv[5], std::ios::binary);
            if (n_frames) write_mvec_frame(out, MotionVectors{});             // first frame of a stream: no vectors
            size_t total = 0;
            for (const auto& mv : pairs) { write_mvec_frame(out, mv); total += mv.size(); }
            std::printf("{\"frames\": %zu, \"vectors\": %zu, \"workers\": %d}\n", n_frames, total, md.workers());
            return 0;
        }
        if (cmd == "detect" && argc >= 4 && std::string(argv[2]) == "--config") {
            std::ifstream f(argv[3]);
            if (!f) throw Error(std::string("cannot open ") + argv[3]);
            std::stringstream ss; ss << f.rdbuf();
            const MotionDetectionConfig cfg = MotionDetectionConfig::from_json(ss.str());
            std::string perf_dir;
            size_t max_frames = SIZE_MAX;
            for (int a = 4; a < argc; ++a) {
                if (std::string(argv[a]) == "--perf-csv" && a + 1 < argc) perf_dir = argv[++a];
                else max_frames = std::strtoull(argv[a], nullptr, 10);
            }
            auto dec = create_decoder(cfg.decoder.selected_plugin, cfg.decoder.arg);
            auto det = create_detector(cfg.detector.selected_plugin, cfg.detector.arg);
            transfer_props(cfg.decoder_properties, *dec);             // detection.rs:100-109
            transfer_props(cfg.detector_properties, *det);
            const DetectionRun run = run_detection(*dec, *det, max_frames);
            if (!perf_dir.empty())
                export_perf_csv(perf_dir, cfg.decoder.selected_plugin, {{"decoder", &run.decoder_ms}, {cfg.detector.selected_plugin, &run.detector_ms}});
            std::printf("{\"frames\": %zu, \"decoder_ms_mean\": %.4f, \"detector_ms_mean\": %.4f, \"max_frame_gap\": %zu, \"min_frames\": %zu, "
                        "\"detector_properties\": {", run.frames, mean(run.decoder_ms), mean(run.detector_ms), cfg.max_frame_gap, cfg.min_frames);
            bool first = true;
            for (auto& [n, p] : det->props()) {
                std::printf("%s\"%s\": ", first ? "" : ", ", n.c_str());
                if (auto fl = std::get_if<BoundedProp<float>>(&p)) std::printf("%.9g", (double)fl->val);
                else if (auto u = std::get_if<BoundedProp<size_t>>(&p)) std::printf("%zu", u->val);
                else if (auto b = std::get_if<bool>(&p)) std::printf("%s", *b ? "true" : "false");
                else std::printf("\"%s\"", std::get<std::string>(p).c_str());
                first = false;
            }
            std::printf("}, \"motion_ranges\": [");
            first = true;
            for (auto [s, e] : run.filtered(cfg.max_frame_gap, cfg.min_frames)) {
                std::printf("%s[%zu, %zu]", first ? "" : ", ", s, e);
                first = false;
            }
            std::printf("]}\n");
            return 0;
        }
        if (cmd == "parse-config" && argc >= 3) {                     // no GPU: what a saved configuration says
            std::ifstream f(argv[2]);
            if (!f) throw Error(std::string("cannot open ") + argv[2]);
            std::stringstream ss; ss << f.rdbuf();
            const MotionDetectionConfig cfg = MotionDetectionConfig::from_json(ss.str());
            auto dump = [](const std::vector<std::pair<std::string, Property>>& props) {
                bool first = true;
                for (const auto& [n, p] : props) {
                    std::printf("%s\"%s\": ", first ? "" : ", ", n.c_str());
                    if (auto fl = std::get_if<BoundedProp<float>>(&p)) std::printf("[\"Float\", %.9g, %.9g, %.9g]", (double)fl->val, (double)fl->min, (double)fl->max);
                    else if (auto u = std::get_if<BoundedProp<size_t>>(&p)) std::printf("[\"Usize\", %zu, %zu, %zu]", u->val, u->min, u->max);
                    else if (auto b = std::get_if<bool>(&p)) std::printf("[\"Bool\", %s]", *b ? "true" : "false");
                    else std::printf("[\"String\", \"%s\"]", std::get<std::string>(p).c_str());
                    first = false;
                }
            };
            std::printf("{\"decoder\": [\"%s\", \"%s\", %s], \"detector\": [\"%s\", \"%s\", %s], \"max_frame_gap\": %zu, \"min_frames\": %zu, "
                        "\"overlay_mf\": %s, \"realtime_processing\": %s, \"decoder_properties\": {",
                        cfg.decoder.selected_plugin.c_str(), cfg.decoder.arg.c_str(), cfg.decoder_open ? "true" : "false",
                        cfg.detector.selected_plugin.c_str(), cfg.detector.arg.c_str(), cfg.detector_open ? "true" : "false",
                        cfg.max_frame_gap, cfg.min_frames, cfg.overlay_mf ? "true" : "false", cfg.realtime_processing ? "true" : "false");
            dump(cfg.decoder_properties);
            std::printf("}, \"detector_properties\": {");
            dump(cfg.detector_properties);
            std::printf("}}\n");
            return 0;
        }
        if (cmd == "stream-bench" && argc >= 5) {
            // optional `--block B --range R` anywhere behind the mode (default 16 / 16): BASELINE configs[3] is 8 / 32
            // `--probe` (batch mode): also print the bare H2D rate of one batch buffer, the CPU this thread runs on and the NUMA
            // node the page-locked buffers landed on (the batched form is bimodal from process to process: profiles/r05/batched_bimodal.txt)
            int sb_block = 16, sb_range = 16;
            bool sb_probe = false;
            {
                int keep = 0;
                for (int a = 0; a < argc; ++a) {
                    if (std::string(argv[a]) == "--probe") { sb_probe = true; continue; }
                    if (a + 1 < argc && std::string(argv[a]) == "--block") { sb_block = std::atoi(argv[++a]); continue; }
                    if (a + 1 < argc && std::string(argv[a]) == "--range") { sb_range = std::atoi(argv[++a]); continue; }
                    argv[keep++] = argv[a];
                }
                argc = keep;
            }
            // the Decoder::process_frame shape without Python in the loop: frames are already in page-locked buffers (a
            // decoder that writes there), one H2D + search + 16 B/vector D2H per frame
            const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
            const int frames = std::atoi(argv[4]);
            const bool ahead = !(argc > 5 && std::string(argv[5]) == "sync");
            const int batch = (argc > 6 && std::string(argv[5]) == "batch") ? std::atoi(argv[6]) : 0;
            if (argc > 6 && std::string(argv[5]) == "multi") {
                // the same stream dealt to several workers (one context + thread per device entry; entries may repeat): batches of
                // `mb` frames from page-locked blocks, 2 per worker in flight, vectors + island + quaternion per frame come back
                const int mb = std::atoi(argv[6]);
                std::vector<int> devices;
                for (int a = 7; a < argc; ++a) devices.push_back(std::atoi(argv[a]));
                if (devices.empty()) devices.push_back(0);
                ofps_hip_multi* m = nullptr;
                if (ofps_hip_multi_init(devices.data(), (int)devices.size(), &m) != OFPS_HIP_OK) throw Error(ofps_hip_multi_last_error(nullptr));
                const int nw = (int)devices.size(), slots = 2 * nw;
                const size_t nblk = ofps_hip_sad_block_count(W, H, sb_block);
                HipContext alloc;                                           // page-locked memory comes from any context
                std::vector<uint8_t*> pin((size_t)slots); std::vector<float*> ent((size_t)slots);
                for (auto& p : pin) { void* q; alloc.check(ofps_hip_host_alloc(alloc.get(), (size_t)mb * W * H, &q)); p = static_cast<uint8_t*>(q); }
                for (auto& p : ent) { void* q; alloc.check(ofps_hip_host_alloc(alloc.get(), (size_t)mb * nblk * 16, &q)); p = static_cast<float*>(q); }
                uint32_t st = 12345;
                for (auto& p : pin) for (size_t i = 0; i < (size_t)mb * W * H; ++i) { st = st * 1664525u + 1013904223u; p[i] = (uint8_t)(st >> 24); }
                ofps_hip_frame_params prm{};
                prm.block = sb_block; prm.range = sb_range; prm.run_detector = 1; prm.min_size = 0.05f; prm.subdivide = 3; prm.target_motion = 0.003f;
                prm.run_estimator = 1; prm.aspect = (float)W / (float)H; prm.fov_y_deg = 39.6f * (float)H / (float)W;
                std::vector<ofps_hip_frame_result> res((size_t)mb);
                auto mcheck = [&](int rc) { if (rc != OFPS_HIP_OK) throw Error(ofps_hip_multi_last_error(m)); };
                auto run = [&](int nb) {
                    mcheck(ofps_hip_multi_reset_frames(m));
                    std::vector<int> q;
                    for (int k = 0; k < nb; ++k) {
                        if ((int)q.size() == slots) { mcheck(ofps_hip_multi_frames_wait(m, q.front(), res.data())); q.erase(q.begin()); }
                        int t = 0;
                        prm.seed = (uint64_t)k * (uint64_t)mb;
                        mcheck(ofps_hip_multi_push_frames_async(m, pin[(size_t)(k % slots)], mb, W, H, W, (size_t)W * H, &prm, ent[(size_t)(k % slots)], &t));
                        q.push_back(t);
                    }
                    for (int t : q) mcheck(ofps_hip_multi_frames_wait(m, t, res.data()));
                };
                const int nb = (frames + mb - 1) / mb;
                run(2 * slots);
                const auto t0 = std::chrono::steady_clock::now();
                run(nb);
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / ((double)nb * mb);
                std::printf("{\"mode\": \"multi_read_ahead_batched\", \"workers\": %d, \"batch\": %d, \"frames\": %d, \"block\": %d, \"range\": %d, "
                            "\"ms_per_frame\": %.4f, \"Mvectors_per_s\": %.2f, \"per_frame\": \"vectors + block-motion island + almeida LSQ quaternion\"}\n",
                            nw, mb, nb * mb, sb_block, sb_range, ms, (double)nblk / ms / 1e3);
                for (auto p : pin) ofps_hip_host_free(alloc.get(), p);
                for (auto p : ent) ofps_hip_host_free(alloc.get(), p);
                ofps_hip_multi_destroy(m);
                return 0;
            }
            if (batch > 0) {
                // batched read-ahead form: `batch` frames per ticket from ONE page-locked block, two tickets in flight
                HipContext ctx;
                const size_t nblk = ofps_hip_sad_block_count(W, H, 16);
                uint8_t* pin[2]; float* ent[2];
                for (auto& p : pin) { void* q; ctx.check(ofps_hip_host_alloc(ctx.get(), (size_t)batch * W * H, &q)); p = static_cast<uint8_t*>(q); }
                for (auto& p : ent) { void* q; ctx.check(ofps_hip_host_alloc(ctx.get(), (size_t)batch * nblk * 16, &q)); p = static_cast<float*>(q); }
                uint32_t st = 12345;
                for (auto& p : pin) for (size_t i = 0; i < (size_t)batch * W * H; ++i) { st = st * 1664525u + 1013904223u; p[i] = (uint8_t)(st >> 24); }
                double probe_gbs[2] = {0, 0};
                int probe_node[2] = {-1, -1}, probe_cpu = -1;
                if (sb_probe) {
                    void* d = nullptr;
                    const size_t bytes = (size_t)batch * W * H;
                    ctx.check(ofps_hip_malloc(ctx.get(), bytes, &d));
                    for (int k = 0; k < 2; ++k) {
                        ctx.check(ofps_hip_memcpy_h2d(ctx.get(), d, pin[k], bytes));
                        const auto t0 = std::chrono::steady_clock::now();
                        for (int r = 0; r < 8; ++r) ctx.check(ofps_hip_memcpy_h2d(ctx.get(), d, pin[k], bytes));
                        probe_gbs[k] = 8.0 * (double)bytes / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e9;
                        void* page = pin[k]; int status = -1;
                        if (syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0) == 0) probe_node[k] = status;
                    }
                    probe_cpu = sched_getcpu();
                    ctx.check(ofps_hip_free(ctx.get(), d));
                }
                ofps_hip_frame_params prm{}; prm.block = 16; prm.range = 16;
                std::vector<ofps_hip_frame_result> res((size_t)batch);
                auto run = [&](int nb) {
                    ctx.check(ofps_hip_reset_frames(ctx.get()));
                    int prev = -1, t = 0;
                    for (int k = 0; k < nb; ++k) {
                        ctx.check(ofps_hip_push_frames_async(ctx.get(), pin[k % 2], batch, W, H, W, (size_t)W * H, &prm, ent[k % 2], &t));
                        if (prev >= 0) ctx.check(ofps_hip_frames_wait(ctx.get(), prev, res.data()));
                        prev = t;
                    }
                    if (prev >= 0) ctx.check(ofps_hip_frames_wait(ctx.get(), prev, res.data()));
                };
                const int nb = (frames + batch - 1) / batch;
                run(4);
                const auto t0 = std::chrono::steady_clock::now();
                run(nb);
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / ((double)nb * batch);
                if (sb_probe)
                    std::printf("{\"mode\": \"read_ahead_batched\", \"batch\": %d, \"frames\": %d, \"ms_per_frame\": %.4f, \"Mvectors_per_s\": %.2f, "
                                "\"bare_h2d_GBs\": [%.1f, %.1f], \"buffer_numa_node\": [%d, %d], \"cpu\": %d}\n", batch, nb * batch, ms,
                                (double)nblk / ms / 1e3, probe_gbs[0], probe_gbs[1], probe_node[0], probe_node[1], probe_cpu);
                else
                std::printf("{\"mode\": \"read_ahead_batched\", \"batch\": %d, \"frames\": %d, \"ms_per_frame\": %.4f, \"Mvectors_per_s\": %.2f}\n", batch,
                            nb * batch, ms, (double)nblk / ms / 1e3);
                for (auto p : pin) ofps_hip_host_free(ctx.get(), p);
                for (auto p : ent) ofps_hip_host_free(ctx.get(), p);
                return 0;
            }
            HipContext ctx;
            uint8_t* pin[3]; float* ent[2];
            const size_t nblk = ofps_hip_sad_block_count(W, H, 16);
            for (auto& p : pin) { void* q; ctx.check(ofps_hip_host_alloc(ctx.get(), (size_t)W * H, &q)); p = static_cast<uint8_t*>(q); }
            for (auto& p : ent) { void* q; ctx.check(ofps_hip_host_alloc(ctx.get(), nblk * 16, &q)); p = static_cast<float*>(q); }
            uint32_t st = 12345;
            for (auto& p : pin) for (size_t i = 0; i < (size_t)W * H; ++i) { st = st * 1664525u + 1013904223u; p[i] = (uint8_t)(st >> 24); }
            ofps_hip_frame_params prm{}; prm.block = 16; prm.range = 16;
            ofps_hip_frame_result res{};
            auto run = [&](int n) {
                ctx.check(ofps_hip_reset_frames(ctx.get()));
                int prev = -1, t = 0;
                for (int k = 0; k < n; ++k) {
                    ctx.check(ofps_hip_push_frame_async(ctx.get(), pin[k % 3], W, H, W, &prm, ent[k % 2], nullptr, &t));
                    if (!ahead) { ctx.check(ofps_hip_frame_wait(ctx.get(), t, &res)); continue; }
                    if (prev >= 0) ctx.check(ofps_hip_frame_wait(ctx.get(), prev, &res));
                    prev = t;
                }
                if (ahead && prev >= 0) ctx.check(ofps_hip_frame_wait(ctx.get(), prev, &res));
            };
            run(20);
            const auto t0 = std::chrono::steady_clock::now();
            run(frames);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
            std::printf("{\"mode\": \"%s\", \"frames\": %d, \"ms_per_frame\": %.4f, \"Mvectors_per_s\": %.2f}\n", ahead ? "read_ahead" : "sync",
                        frames, ms, (double)nblk / ms / 1e3);
            for (auto p : pin) ofps_hip_host_free(ctx.get(), p);
            for (auto p : ent) ofps_hip_host_free(ctx.get(), p);
            return 0;
        }
        if (cmd == "detect" && argc >= 4) {
            auto dec = create_decoder(argv[2], argv[3]);
            auto det = create_detector("hip_block_motion", "");
            const size_t max_frames = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : SIZE_MAX;
            const DetectionRun run = run_detection(*dec, *det, max_frames);
            std::printf("{\"frames\": %zu, \"decoder_ms_mean\": %.4f, \"detector_ms_mean\": %.4f, \"motion_ranges\": [", run.frames,
                        mean(run.decoder_ms), mean(run.detector_ms));
            bool first = true;
            for (auto [s, e] : run.filtered(2, 2)) {                   // defaults of the sample config
                std::printf("%s[%zu, %zu]", first ? "" : ", ", s, e);
                first = false;
            }
            std::printf("]}\n");
            return 0;
        }
        if (cmd == "track" && argc >= 4) {
            auto dec = create_decoder(argv[2], argv[3]);
            auto est = std::make_unique<HipAlmeidaEstimator>();
            const float aspect = argc > 5 ? std::strtof(argv[4], nullptr) : 16.0f / 9.0f;
            const float fov_y = argc > 5 ? std::strtof(argv[5], nullptr) : 39.6f * 9.0f / 16.0f;   // worker.rs:445
            est->use_ransac = !(argc > 6 && std::string(argv[6]) == "lsq");
            const StandardCamera cam(aspect, fov_y);
            const TrackingRun run = run_tracking(*dec, *est, cam);
            std::printf("frame,w,i,j,k,decoder_ms,estimator_ms\n");
            for (size_t f = 0; f < run.frames; ++f)
                std::printf("%zu,%.9g,%.9g,%.9g,%.9g,%.4f,%.4f\n", f, run.rotations[f].w, run.rotations[f].i, run.rotations[f].j,
                            run.rotations[f].k, run.decoder_ms[f], run.estimator_ms[f]);
            return 0;
        }
        std::fprintf(stderr, "usage: ofps_hip_tool extract|detect|track|mvec-copy|stream-bench ... (see source header)\n");
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
