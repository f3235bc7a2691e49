// ofps_host.hpp -- C++ host layer above the C ABI (include/ofps_hip.h).
//
// The reference's host code is Rust and its toolchain is not available here, so this layer mirrors the
// reference's operator interface for the hot path in C++: same names, argument meaning and error
// behaviour as the traits in ofps/src/{decoder,estimator,detection}.rs and the Properties trait of
// ofps/src/plugins/properties.rs.  It contains no arithmetic of its own: every result comes from
// libofps_hip.so.  (The Rust shim a maintainer would add is in INTEGRATION.md.)
#pragma once

#include <cstddef>
#include <cstdint>
#include <istream>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/ofps_hip.h"

namespace ofps {

// ---- ofps/src/decoder.rs:8-15, 40-42
struct RGBA { uint8_t r, g, b, a; };
struct MotionEntry { float pos_x, pos_y, motion_x, motion_y; };     // (Point2<f32>, Vector2<f32>)
static_assert(sizeof(MotionEntry) == 16, "MotionEntry must be the 16-byte record of the C ABI / .mvec");
using MotionVectors = std::vector<MotionEntry>;

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };   // stands in for anyhow::Error

// ---- ofps/src/plugins/properties.rs:6-18, 63-68, 120-188
template <class T> struct BoundedProp { T val, min, max; };
using Property = std::variant<std::string, bool, BoundedProp<float>, BoundedProp<size_t>>;
struct PropertyMut {                       // a typed reference into the plugin object
    std::variant<std::string*, bool*, float*, size_t*> ref;
    float fmin = 0, fmax = 0; size_t umin = 0, umax = 0;
    static PropertyMut string(std::string* s) { PropertyMut p; p.ref = s; return p; }
    static PropertyMut boolean(bool* b) { PropertyMut p; p.ref = b; return p; }
    static PropertyMut float_(float* v, float lo, float hi) { PropertyMut p; p.ref = v; p.fmin = lo; p.fmax = hi; return p; }
    static PropertyMut usize(size_t* v, size_t lo, size_t hi) { PropertyMut p; p.ref = v; p.umin = lo; p.umax = hi; return p; }
    Property get() const;
    void set(const Property& other);       // PropertyMut::set (properties.rs:180-188): mismatched kinds are ignored
};
class Properties {
public:
    virtual ~Properties() = default;
    virtual std::vector<std::pair<std::string, PropertyMut>> props_mut() { return {}; }
    std::vector<std::pair<std::string, Property>> props();
    bool set_prop(const std::string& name, const Property& value);   // transfer_props of one entry
};

// ---- ofps/src/camera.rs:12-35, 166-177 (parameters only; the projection math runs on the device)
class StandardCamera {
public:
    StandardCamera(float aspect, float fov_y_deg) : aspect_(aspect), fov_y_(fov_y_deg) {}
    float aspect_ratio() const { return aspect_; }
    std::pair<float, float> fov() const;          // (horizontal, vertical) degrees
private:
    float aspect_, fov_y_;
};

// ---- ofps/src/motion_field.rs:7-115 (storage + accessors)
class MotionField {
public:
    MotionField() = default;
    MotionField(size_t width, size_t height) : vf_(2 * width * height, 0.0f), width_(width) {}
    std::pair<size_t, size_t> dim() const { return width_ == 0 ? std::make_pair<size_t, size_t>(0, 0) : std::make_pair(width_, vf_.size() / 2 / width_); }
    size_t size() const { return vf_.size() / 2; }
    const std::vector<float>& as_slice() const { return vf_; }      // [u00,v00,u01,v01,...] row-major cells
    std::vector<float>& raw() { return vf_; }
    void set_motion(size_t x, size_t y, float mx, float my) { vf_[2 * (width_ * y + x)] = mx; vf_[2 * (width_ * y + x) + 1] = my; }
    std::pair<float, float> get_motion(size_t x, size_t y) const { return {vf_[2 * (width_ * y + x)], vf_[2 * (width_ * y + x) + 1]}; }
    MotionVectors motion_iter() const;                               // pos = (x/w, y/h), motion_field.rs:106-114
private:
    std::vector<float> vf_;
    size_t width_ = 0;
};

struct UnitQuaternion {
    float w = 1, i = 0, j = 0, k = 0;
    UnitQuaternion operator*(const UnitQuaternion& b) const;         // Hamilton product
    void rotate(const float v[3], float out[3]) const;
};
struct Vector3 { float x = 0, y = 0, z = 0; };

// ---- traits
class Decoder : public Properties {                                  // ofps/src/decoder.rs:45-73
public:
    // true: vectors appended to `field`; false: no vectors this frame; throws Error: end of stream / failure.
    virtual bool process_frame(MotionVectors& field, std::vector<RGBA>* out_frame, size_t* out_height, size_t skip_frames) = 0;
    virtual std::optional<double> get_framerate() const = 0;
    virtual std::optional<std::pair<size_t, size_t>> get_aspect() const = 0;
};
class Estimator : public Properties {                                // ofps/src/estimator.rs:8-53
public:
    virtual std::pair<UnitQuaternion, Vector3> estimate(const MotionEntry* mv, size_t n, const StandardCamera& camera,
                                                        std::optional<float> move_magnitude) = 0;
    void motion_step(const MotionEntry* mv, size_t n, const StandardCamera& camera, std::optional<float> move_magnitude,
                     UnitQuaternion& rot, float pos[3]);               // pos += rot*tr; rot = r*rot
};
class Detector : public Properties {                                 // ofps/src/detection.rs:6-12
public:
    virtual std::optional<std::pair<size_t, MotionField>> detect_motion(const MotionEntry* mv, size_t n) = 0;
};

// ---- owning wrapper of one ofps_hip_ctx
class HipContext {
public:
    explicit HipContext(int device = 0);
    ~HipContext();
    HipContext(const HipContext&) = delete;
    HipContext& operator=(const HipContext&) = delete;
    ofps_hip_ctx* get() const { return ctx_; }
    void check(int rc) const;            // throws Error with ofps_hip_last_error
private:
    ofps_hip_ctx* ctx_ = nullptr;
};

// One process, several GPUs: ofps_hip_multi_* (include/ofps_hip.h).  A batch of frame pairs is split into contiguous ranges
// over the devices and comes back in pair order -- the host-side counterpart of the reference's one-worker-per-plugin
// threading (ofps-suite/src/app/tracking/worker.rs:251-260,347-352) for a machine with more than one GPU.
class MultiDeviceSad {
public:
    explicit MultiDeviceSad(const std::vector<int>& devices);
    ~MultiDeviceSad();
    MultiDeviceSad(const MultiDeviceSad&) = delete;
    MultiDeviceSad& operator=(const MultiDeviceSad&) = delete;
    // frames: n_frames luma frames of w x h bytes, back to back; ref_mode 0: pairs (k, k+1), 1: pairs (0, k+1).
    // -> one MotionVectors per pair, in pair order
    std::vector<MotionVectors> search(const uint8_t* frames, size_t n_frames, size_t w, size_t h, int block, int range, int ref_mode = 0);
    int workers() const;
private:
    ofps_hip_multi* m_ = nullptr;
};

// ---- plugins on the HIP path
// "hip_sad": raw 8-bit luma frames (W*H bytes, back to back) from a stream -> MotionEntry per block.
class HipSadDecoder : public Decoder {
public:
    HipSadDecoder(std::unique_ptr<std::istream> input, size_t width, size_t height, std::optional<double> fps, int device = 0);
    ~HipSadDecoder() override;
    bool process_frame(MotionVectors& field, std::vector<RGBA>* out_frame, size_t* out_height, size_t skip_frames) override;
    std::optional<double> get_framerate() const override { return fps_; }
    std::optional<std::pair<size_t, size_t>> get_aspect() const override { return std::make_pair(w_, h_); }
    std::vector<std::pair<std::string, PropertyMut>> props_mut() override;
private:
    HipContext ctx_;
    std::unique_ptr<std::istream> in_;
    size_t w_, h_, block_ = 16, range_ = 16;
    bool pruned_ = false;                  // OFPS_HIP_SAD_PRUNED: identical vectors, content-dependent run time
    std::optional<double> fps_;
    uint8_t* frame_ = nullptr;             // page-locked staging buffer for the frame being read
    std::vector<float> out_;
};

// "hip_lk": dense per-pixel flow in cv-decoder's full-resolution mode (cv-decoder/src/lib.rs:82-294): one record per
// visited cell of the (Width, Height)-capped grid; property names as in cv-decoder (:35-52).
class HipLkDecoder : public Decoder {
public:
    // farneback = true is the "hip_flow" plugin: Farneback's polynomial-expansion flow with cv-decoder's arguments
    // (cv-decoder/src/lib.rs:188-199: levels 5, winsize 13 = 2 * 6 + 1, 3 iterations, poly_n 7, poly_sigma 1.5) through the same
    // entry points (OFPS_HIP_FLOW_FARNEBACK); the properties keep their names, "Window radius" r means winsize 2 r + 1
    // frame_format: OFPS_HIP_FMT_LUMA (raw luma streams, the default) or a colour format (OFPS_HIP_FMT_BGR = what cv-decoder's VideoCapture
    // hands it): colour frames are converted on the device with cvt_color(BGR2GRAY)'s formula (cv-decoder/src/lib.rs:135)
    HipLkDecoder(std::unique_ptr<std::istream> input, size_t width, size_t height, std::optional<double> fps, int device = 0,
                 bool farneback = false, int frame_format = OFPS_HIP_FMT_LUMA);
    bool process_frame(MotionVectors& field, std::vector<RGBA>* out_frame, size_t* out_height, size_t skip_frames) override;
    std::optional<double> get_framerate() const override { return fps_; }
    // cv-decoder/src/lib.rs:296-298: the size of `self.gray` -- the reduced frame's with "Process Fullres" = false
    std::optional<std::pair<size_t, size_t>> get_aspect() const override {
        if (process_fullres_) return std::make_pair(w_, h_);
        int gw = 0, gh = 0;
        ofps_hip_cv_grid((int)w_, (int)h_, (int)max_w_, (int)max_h_, &gw, &gh);
        return std::make_pair((size_t)gw, (size_t)gh);
    }
    std::vector<std::pair<std::string, PropertyMut>> props_mut() override;
private:
    HipContext ctx_;
    std::unique_ptr<std::istream> in_;
    size_t w_, h_, max_w_ = 150, max_h_ = 150, levels_ = 3, radius_ = 4, iters_ = 3;
    // "Process Fullres" has cv-decoder's meaning (cv-decoder/src/lib.rs:124-133,274-276): false = frames resized to the capped grid before
    // the flow, one record per unmasked pixel of the reduced frame.  "Fullres records" is this build's own output form.
    bool contrast_mask_ = true, process_fullres_ = true, fullres_records_ = false;      // cv-decoder: the Farneback path always masks (:203-237)
    std::optional<double> fps_;
    int fmt_ = OFPS_HIP_FMT_LUMA;
    size_t cn_ = 1;
    std::vector<uint8_t> prev_, cur_, shown_;
    std::vector<float> out_;
    size_t frames_read_ = 0;
    bool on_device_ = false;           // the last frame of the previous call is on the device (ofps_hip_lk_push_frame's state)
    unsigned on_device_flags_ = 0;     // ... pushed with these flags / parameters (a property change restarts the library's stream)
    size_t on_device_params_[5] = {0, 0, 0, 0, 0};
    bool farneback_ = false;
};

// MvecFile of motion-loader/src/lib.rs:31-83 (pure host I/O, no GPU)
class MvecFileDecoder : public Decoder {
public:
    explicit MvecFileDecoder(std::unique_ptr<std::istream> input) : in_(std::move(input)) {}
    bool process_frame(MotionVectors& field, std::vector<RGBA>*, size_t*, size_t) override;
    std::optional<double> get_framerate() const override { return std::nullopt; }
    std::optional<std::pair<size_t, size_t>> get_aspect() const override { return std::nullopt; }
private:
    std::unique_ptr<std::istream> in_;
};
// writer side: motion-extract/src/main.rs:23-35
void write_mvec_frame(std::ostream& out, const MotionVectors& mv);

class HipBlockMotionDetection : public Detector {                    // block-motion-detector/src/lib.rs:13-46
public:
    explicit HipBlockMotionDetection(int device = 0) : ctx_(device) {}
    float min_size = 0.05f; size_t subdivide = 3; float target_motion = 0.003f;
    std::optional<std::pair<size_t, MotionField>> detect_motion(const MotionEntry* mv, size_t n) override;
    std::vector<std::pair<std::string, PropertyMut>> props_mut() override;
private:
    HipContext ctx_;
};

class HipAlmeidaEstimator : public Estimator {                       // almeida-estimator/src/lib.rs:57-121
public:
    explicit HipAlmeidaEstimator(int device = 0) : ctx_(device) {}
    bool use_ransac = true; size_t num_iters = 200; float inlier_angle = 0.05f; size_t ransac_samples = 1000;
    uint64_t seed = 0;                                                // advanced once per estimate()
    std::pair<UnitQuaternion, Vector3> estimate(const MotionEntry* mv, size_t n, const StandardCamera& camera,
                                                std::optional<float> move_magnitude) override;
    std::vector<std::pair<std::string, PropertyMut>> props_mut() override;
private:
    HipContext ctx_;
};

// ---- input streams: ofps::utils::open_file (ofps/src/utils.rs:92-118).  "tcp://host:port" connects, "tcp://@:port"
// listens on 0.0.0.0:port and accepts one connection, anything else is a file path.
std::unique_ptr<std::istream> open_file(const std::string& input);

// ---- minimal JSON value (objects, arrays, strings, numbers, booleans, null) for the saved configurations
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json* get(const std::string& key) const;                   // object member or nullptr
    static Json parse(const std::string& text);                      // throws Error on malformed input
};

// ---- the saved detection configuration (ofps-suite/src/app/detection.rs:23-50,171-179; widgets.rs:273-278;
// Property is the externally tagged enum of ofps/src/plugins/properties.rs:63-68)
struct CreatePluginConfig { std::string selected_plugin, arg; };
struct MotionDetectionConfig {
    CreatePluginConfig decoder, detector;
    bool decoder_open = false, detector_open = false;               // second element of the (config, bool) tuples
    std::vector<std::pair<std::string, Property>> decoder_properties, detector_properties;
    bool realtime_processing = false, overlay_mf = false;
    size_t max_frame_gap = 2, min_frames = 2;                        // Default (:33-43)
    static MotionDetectionConfig from_json(const std::string& text);
};
// transfer_props (ofps/src/plugins/properties.rs:120-135): every saved property whose name the plugin has is applied
void transfer_props(const std::vector<std::pair<std::string, Property>>& saved, Properties& plugin);

// ---- creation by name (the part of PluginStore the hot path needs: ofps/src/plugins/mod.rs:396-453)
std::unique_ptr<Decoder> create_decoder(const std::string& name, const std::string& arg);    // "hip_sad", "hip_lk", "hip_flow", "mvec"
std::unique_ptr<Detector> create_detector(const std::string& name, const std::string& arg);  // "hip_block_motion"
std::unique_ptr<Estimator> create_estimator(const std::string& name, const std::string& arg);// "hip_almeida"

// ---- caller harnesses
// Detection loop of ofps-suite/src/app/detection.rs:92-168 + filtered_motion_ranges (:195-212).
struct DetectionRun {
    size_t frames = 0;
    std::vector<std::pair<size_t, size_t>> motion_ranges;
    std::vector<double> decoder_ms, detector_ms;
    std::vector<std::pair<size_t, size_t>> filtered(size_t max_frame_gap, size_t min_frames) const;
};
// "Export all stats" (ofps-suite/src/app/utils/perf_stats.rs:86-121): one perf_<name>_<decoder>.csv per timed stage in
// `dir`, one millisecond value per line (csv::Writer::serialize of an f32: no header).
void export_perf_csv(const std::string& dir, const std::string& decoder_name,
                     const std::vector<std::pair<std::string, const std::vector<double>*>>& stages);
DetectionRun run_detection(Decoder& decoder, Detector& detector, size_t max_frames = SIZE_MAX);
// Tracking loop of ofps-suite/src/app/tracking/worker.rs:62-69,305-412 (one estimator).
struct TrackingRun {
    size_t frames = 0;
    std::vector<UnitQuaternion> rotations;          // accumulated camera rotation after each frame
    std::vector<double> decoder_ms, estimator_ms;
};
TrackingRun run_tracking(Decoder& decoder, Estimator& estimator, const StandardCamera& camera, size_t max_frames = SIZE_MAX);

}  // namespace ofps
