// ofps_hip_tool -- small CLI over the C++ host layer; the counterparts of the reference's non-GUI callers:
//   extract <decoder> <arg> <out.mvec> [max_frames]     motion-extract/src/main.rs:7-38 (decode -> .mvec)
//   detect  <decoder> <arg> [max_frames]                detection loop, ofps-suite/src/app/detection.rs:92-168
//   track   <decoder> <arg> [aspect fov_y] [lsq|ransac] tracking loop, ofps-suite/src/app/tracking/worker.rs:305-412
//   mvec-copy <in.mvec> <out.mvec>                      CPU-only .mvec round trip (reader + writer)
// decoder = hip_sad ("<raw luma file>?w=..&h=..&fps=..") or mvec ("<file.mvec>").
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

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
            MotionVectors mv;
            size_t frames = 0, total = 0;
            while (frames < max_frames) {                              // while c.process_frame(..).is_ok()
                mv.clear();
                try { dec->process_frame(mv, nullptr, nullptr, 0); } catch (const Error&) { break; }
                write_mvec_frame(out, mv);                             // a frame without vectors is written with count 0
                ++frames; total += mv.size();
            }
            std::printf("{\"frames\": %zu, \"vectors\": %zu}\n", frames, total);
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
        std::fprintf(stderr, "usage: ofps_hip_tool extract|detect|track|mvec-copy ... (see source header)\n");
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
