// ofps_host.cpp -- implementation of the C++ host layer (see ofps_host.hpp).  No arithmetic on the data
// path lives here: decode, detect and estimate are calls into libofps_hip.so.
#include "ofps_host.hpp"

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <streambuf>

namespace ofps {

// ------------------------------------------------------------------ Properties
Property PropertyMut::get() const {
    if (auto s = std::get_if<std::string*>(&ref)) return Property{**s};
    if (auto b = std::get_if<bool*>(&ref)) return Property{**b};
    if (auto f = std::get_if<float*>(&ref)) return Property{BoundedProp<float>{**f, fmin, fmax}};
    return Property{BoundedProp<size_t>{*std::get<size_t*>(ref), umin, umax}};
}

void PropertyMut::set(const Property& other) {
    if (auto s = std::get_if<std::string*>(&ref)) { if (auto o = std::get_if<std::string>(&other)) **s = *o; }
    else if (auto b = std::get_if<bool*>(&ref)) { if (auto o = std::get_if<bool>(&other)) **b = *o; }
    else if (auto f = std::get_if<float*>(&ref)) { if (auto o = std::get_if<BoundedProp<float>>(&other)) **f = o->val; }
    else if (auto u = std::get_if<size_t*>(&ref)) { if (auto o = std::get_if<BoundedProp<size_t>>(&other)) **u = o->val; }
}

std::vector<std::pair<std::string, Property>> Properties::props() {
    std::vector<std::pair<std::string, Property>> out;
    for (auto& [n, p] : props_mut()) out.emplace_back(n, p.get());
    return out;
}

bool Properties::set_prop(const std::string& name, const Property& value) {
    for (auto& [n, p] : props_mut())
        if (n == name) { p.set(value); return true; }
    return false;
}

// ------------------------------------------------------------------ small types
std::pair<float, float> StandardCamera::fov() const {            // camera.rs:166-170
    const float to_rad = 3.14159265358979323846264338327950288f / 180.0f;
    const float to_deg = 57.2957795130823208767981548141051703f;
    const float ty = std::tan(fov_y_ * to_rad / 2.0f);
    const float tx = aspect_ * ty;
    return {std::atan(tx) * to_deg * 2.0f, fov_y_};
}

MotionVectors MotionField::motion_iter() const {
    MotionVectors out;
    const auto [w, h] = dim();
    for (size_t y = 0; y < h; ++y)
        for (size_t x = 0; x < w; ++x) {
            const auto [mx, my] = get_motion(x, y);
            out.push_back({(float)x / (float)w, (float)y / (float)h, mx, my});
        }
    return out;
}

UnitQuaternion UnitQuaternion::operator*(const UnitQuaternion& b) const {
    UnitQuaternion r;
    r.w = w * b.w - i * b.i - j * b.j - k * b.k;
    r.i = w * b.i + i * b.w + j * b.k - k * b.j;
    r.j = w * b.j - i * b.k + j * b.w + k * b.i;
    r.k = w * b.k + i * b.j - j * b.i + k * b.w;
    return r;
}

void UnitQuaternion::rotate(const float v[3], float out[3]) const {   // v + 2 q.v x (q.v x v + w v)
    const float u[3] = {i, j, k};
    float t[3] = {2 * (u[1] * v[2] - u[2] * v[1]), 2 * (u[2] * v[0] - u[0] * v[2]), 2 * (u[0] * v[1] - u[1] * v[0])};
    const float c[3] = {u[1] * t[2] - u[2] * t[1], u[2] * t[0] - u[0] * t[2], u[0] * t[1] - u[1] * t[0]};
    for (int a = 0; a < 3; ++a) out[a] = (t[a] * w + c[a]) + v[a];
}

void Estimator::motion_step(const MotionEntry* mv, size_t n, const StandardCamera& camera, std::optional<float> mag,
                            UnitQuaternion& rot, float pos[3]) {       // estimator.rs:38-53
    auto [r, tr] = estimate(mv, n, camera, mag);
    const float t[3] = {tr.x, tr.y, tr.z};
    float moved[3];
    rot.rotate(t, moved);
    for (int a = 0; a < 3; ++a) pos[a] += moved[a];
    rot = r * rot;
}

// ------------------------------------------------------------------ context
HipContext::HipContext(int device) {
    const int rc = ofps_hip_init(device, &ctx_);
    if (rc != OFPS_HIP_OK) throw Error(std::string("ofps_hip_init: ") + ofps_hip_last_error(nullptr));
}
HipContext::~HipContext() { ofps_hip_destroy(ctx_); }
void HipContext::check(int rc) const {
    if (rc != OFPS_HIP_OK) throw Error(std::string("libofps_hip (") + std::to_string(rc) + "): " + ofps_hip_last_error(ctx_));
}

// ------------------------------------------------------------------ hip_sad
HipSadDecoder::HipSadDecoder(std::unique_ptr<std::istream> input, size_t width, size_t height, std::optional<double> fps,
                             int device)
    : ctx_(device), in_(std::move(input)), w_(width), h_(height), fps_(fps) {
    if (!in_ || !*in_) throw Error("hip_sad: cannot open input");
    if (w_ == 0 || h_ == 0) throw Error("hip_sad: frame size required (arg \"path?w=..&h=..\")");
    void* p = nullptr;
    ctx_.check(ofps_hip_host_alloc(ctx_.get(), w_ * h_, &p));         // page-locked: the frame is DMA'd, not staged
    frame_ = static_cast<uint8_t*>(p);
}

HipSadDecoder::~HipSadDecoder() {
    if (frame_) ofps_hip_host_free(ctx_.get(), frame_);
}

// The vectors relate the last two frames read from the stream.  Only newly read frames cross PCIe: the previous
// frame stays on the device (ofps_hip_stage_frame / ofps_hip_push_frame keep a two-slot ring in the context).
bool HipSadDecoder::process_frame(MotionVectors& field, std::vector<RGBA>* out_frame, size_t* out_height, size_t skip) {
    const size_t bytes = w_ * h_;
    for (size_t s = 0; s <= skip; ++s) {
        in_->read(reinterpret_cast<char*>(frame_), (std::streamsize)bytes);
        if ((size_t)in_->gcount() != bytes) throw Error("hip_sad: end of stream");   // Err ends the caller's loop
        if (s + 1 == skip) ctx_.check(ofps_hip_stage_frame(ctx_.get(), frame_, (int)w_, (int)h_, (int)w_));
    }
    if (out_frame && out_height) {
        *out_height = h_;
        out_frame->clear();
        out_frame->reserve(bytes);
        for (size_t i = 0; i < bytes; ++i) out_frame->push_back(RGBA{frame_[i], frame_[i], frame_[i], 255});
    }
    const size_t nblk = ofps_hip_sad_block_count((int)w_, (int)h_, (int)block_);
    out_.resize(4 * nblk);
    ofps_hip_frame_params prm{};
    prm.block = (int)block_;
    prm.range = (int)range_;
    ofps_hip_frame_result res{};
    ctx_.check(ofps_hip_set_sad_mode(ctx_.get(), pruned_ ? OFPS_HIP_SAD_PRUNED : OFPS_HIP_SAD_EXHAUSTIVE));
    ctx_.check(ofps_hip_push_frame(ctx_.get(), frame_, (int)w_, (int)h_, (int)w_, &prm, &res, out_.data(), nullptr));
    if (!res.have_vectors) return false;                              // first frame: no pair yet
    const size_t base = field.size();
    field.resize(base + res.n_vectors);                               // vectors are APPENDED (callers clear)
    std::memcpy(field.data() + base, out_.data(), res.n_vectors * sizeof(MotionEntry));
    return true;
}

std::vector<std::pair<std::string, PropertyMut>> HipSadDecoder::props_mut() {
    return {{"Block size", PropertyMut::usize(&block_, 8, 16)}, {"Search range", PropertyMut::usize(&range_, 8, 32)},
            {"Exact pruning", PropertyMut::boolean(&pruned_)}};       // same vectors; faster on smooth camera motion
}

// ------------------------------------------------------------------ hip_lk
HipLkDecoder::HipLkDecoder(std::unique_ptr<std::istream> input, size_t width, size_t height, std::optional<double> fps, int device,
                           bool farneback, int frame_format)
    : ctx_(device), in_(std::move(input)), w_(width), h_(height), fps_(fps), fmt_(frame_format), farneback_(farneback) {
    if (farneback_) { levels_ = 5; radius_ = 6; iters_ = 3; }
    if (!in_ || !*in_) throw Error("hip_lk: cannot open input");
    if (w_ == 0 || h_ == 0) throw Error("hip_lk: frame size required (arg \"path?w=..&h=..\")");
    cn_ = (size_t)ofps_hip_frame_channels(fmt_);
    if (cn_ == 0) throw Error("hip_lk: unknown frame format");
    prev_.resize(w_ * h_ * cn_);
    cur_.resize(w_ * h_ * cn_);
}

bool HipLkDecoder::process_frame(MotionVectors& field, std::vector<RGBA>* out_frame, size_t* out_height, size_t skip) {
    for (size_t s = 0; s <= skip; ++s) {                              // cv-decoder/src/lib.rs:92-142
        std::swap(prev_, cur_);
        in_->read(reinterpret_cast<char*>(cur_.data()), (std::streamsize)cur_.size());
        if ((size_t)in_->gcount() != cur_.size()) throw Error("hip_lk: failed to grab frame");
    }
    if (out_frame && out_height) {                                    // :144-154: `self.frame` -- the arriving frame, or the REDUCED one ("Process Fullres" = false)
        size_t ow = w_, oh = h_;
        const uint8_t* p = cur_.data();
        if (!process_fullres_) {
            int gw = 0, gh = 0;
            ctx_.check(ofps_hip_cv_grid((int)w_, (int)h_, (int)max_w_, (int)max_h_, &gw, &gh));
            shown_.resize((size_t)gw * gh * cn_);
            ctx_.check(ofps_hip_resize_linear(ctx_.get(), cur_.data(), (int)w_, (int)h_, (int)(w_ * cn_), fmt_, shown_.data(), gw, gh));
            ow = (size_t)gw; oh = (size_t)gh; p = shown_.data();
        }
        *out_height = oh;
        out_frame->clear();
        out_frame->reserve(ow * oh);
        for (size_t i = 0; i < ow * oh; ++i, p += cn_) {
            if (fmt_ == OFPS_HIP_FMT_LUMA) out_frame->push_back(RGBA{p[0], p[0], p[0], 255});
            else if (fmt_ == OFPS_HIP_FMT_RGBA) out_frame->push_back(RGBA{p[0], p[1], p[2], 255});
            else out_frame->push_back(RGBA{p[2], p[1], p[0], 255});   // BGR / BGRA
        }
    }
    // :156-158 compares the sizes of gray and old_gray: flow is computed as soon as TWO frames have been read, also when
    // both were read by this very call (skip >= 1 on the first call)
    frames_read_ += skip + 1;
    if (frames_read_ < 2) return false;
    const bool records_fullres = process_fullres_ && fullres_records_;
    out_.resize(records_fullres ? 4 * w_ * h_ : 4 * std::min(max_w_, w_) * std::min(max_h_, h_));
    size_t n_out = 0;
    const unsigned flags = (contrast_mask_ ? OFPS_HIP_LK_CONTRAST_MASK : 0u) | (process_fullres_ ? 0u : OFPS_HIP_LK_REDUCED) |
                           (records_fullres ? OFPS_HIP_LK_FULLRES_RECORDS : 0u) | OFPS_HIP_FRAME_FORMAT(fmt_) |
                           (farneback_ ? OFPS_HIP_FLOW_FARNEBACK | OFPS_HIP_FLOW_USE_PREVIOUS : 0u);   // cv-decoder/src/lib.rs:161-165: its previous flow is the initial flow
    const size_t params[5] = {levels_, radius_, iters_, max_w_, max_h_};
    const bool same_mode = on_device_ && flags == on_device_flags_ && std::equal(params, params + 5, on_device_params_);
    const int pitch = (int)(w_ * cn_);
    // the frame uploaded by the previous call is this call's previous frame unless frames were skipped in between or a property
    // changed: then (and for the first pair) the previous frame goes up first.  After a skip the stream's last flow stays the initial
    // flow (cv-decoder's self.flow persists across skipped reads): rewind, not reset.
    int have = 0;
    if (!(skip == 0 && same_mode)) {
        ctx_.check(same_mode ? ofps_hip_lk_rewind(ctx_.get()) : ofps_hip_lk_reset(ctx_.get()));
        ctx_.check(ofps_hip_lk_push_frame(ctx_.get(), prev_.data(), (int)w_, (int)h_, pitch, (int)levels_, (int)radius_, (int)iters_,
                                          (int)max_w_, (int)max_h_, flags, out_.data(), &n_out, nullptr, nullptr, &have));
    }
    ctx_.check(ofps_hip_lk_push_frame(ctx_.get(), cur_.data(), (int)w_, (int)h_, pitch, (int)levels_, (int)radius_, (int)iters_,
                                      (int)max_w_, (int)max_h_, flags, out_.data(), &n_out, nullptr, nullptr, &have));
    on_device_ = true;
    on_device_flags_ = flags;
    std::copy(params, params + 5, on_device_params_);
    if (!have) throw Error("hip_lk: no vectors for the second frame of a pair");
    const size_t base = field.size();
    field.resize(base + n_out);
    std::memcpy(field.data() + base, out_.data(), n_out * sizeof(MotionEntry));
    return true;
}

std::vector<std::pair<std::string, PropertyMut>> HipLkDecoder::props_mut() {   // cv-decoder/src/lib.rs:35-52 + the flow's knobs
    return {{"Width", PropertyMut::usize(&max_w_, 1, 2000)}, {"Height", PropertyMut::usize(&max_h_, 1, 2000)},
            {"Pyramid levels", PropertyMut::usize(&levels_, 1, 8)}, {"Window radius", PropertyMut::usize(&radius_, 1, farneback_ ? 7 : 15)},
            {"Iterations", PropertyMut::usize(&iters_, 1, 64)}, {"Contrast mask", PropertyMut::boolean(&contrast_mask_)},
            {"Process Fullres", PropertyMut::boolean(&process_fullres_)}, {"Fullres records", PropertyMut::boolean(&fullres_records_)}};
}

// ------------------------------------------------------------------ .mvec
bool MvecFileDecoder::process_frame(MotionVectors& field, std::vector<RGBA>*, size_t*, size_t) {
    uint8_t hdr[4];
    in_->read(reinterpret_cast<char*>(hdr), 4);                       // motion-loader/src/lib.rs:52-53
    if (in_->gcount() != 4) throw Error("mvec: end of stream");
    const uint32_t count = (uint32_t)hdr[0] | ((uint32_t)hdr[1] << 8) | ((uint32_t)hdr[2] << 16) | ((uint32_t)hdr[3] << 24);
    const size_t base = field.size();
    field.resize(base + count);
    in_->read(reinterpret_cast<char*>(field.data() + base), (std::streamsize)count * 16);    // 4 x f32 LE per vector
    if ((size_t)in_->gcount() != (size_t)count * 16) { field.resize(base); throw Error("mvec: truncated frame"); }
    return true;
}

void write_mvec_frame(std::ostream& out, const MotionVectors& mv) {   // motion-extract/src/main.rs:25-32
    const uint32_t n = (uint32_t)mv.size();
    const uint8_t hdr[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    out.write(reinterpret_cast<const char*>(hdr), 4);
    out.write(reinterpret_cast<const char*>(mv.data()), (std::streamsize)mv.size() * 16);   // host is little-endian
}

// ------------------------------------------------------------------ hip_block_motion
std::optional<std::pair<size_t, MotionField>> HipBlockMotionDetection::detect_motion(const MotionEntry* mv, size_t n) {
    const int dim = ofps_hip_block_dim(min_size, subdivide);
    if (dim < 1) return std::nullopt;
    MotionField mf((size_t)dim, (size_t)dim);
    int has = 0, odim = 0;
    size_t area = 0;
    const int rc = ofps_hip_detect(ctx_.get(), reinterpret_cast<const float*>(mv), n, min_size, subdivide, target_motion, &has,
                                   &area, &odim, mf.raw().data());
    if (rc != OFPS_HIP_OK || !has) return std::nullopt;               // the trait has no error channel (detection.rs:11)
    return std::make_pair(area, std::move(mf));
}

std::vector<std::pair<std::string, PropertyMut>> HipBlockMotionDetection::props_mut() {   // lib.rs:29-46
    return {{"Min size", PropertyMut::float_(&min_size, 0.01f, 1.0f)},
            {"Subdivisions", PropertyMut::usize(&subdivide, 1, 16)},
            {"Target motion", PropertyMut::float_(&target_motion, 0.0001f, 0.1f)}};
}

// ------------------------------------------------------------------ hip_almeida
std::pair<UnitQuaternion, Vector3> HipAlmeidaEstimator::estimate(const MotionEntry* mv, size_t n, const StandardCamera& camera,
                                                                 std::optional<float>) {
    float q[4], tr[3];
    ++seed;
    ctx_.check(ofps_hip_almeida(ctx_.get(), reinterpret_cast<const float*>(mv), n, camera.aspect_ratio(), camera.fov().second,
                                use_ransac ? 1 : 0, num_iters, inlier_angle, ransac_samples, seed, q, tr));
    return {UnitQuaternion{q[0], q[1], q[2], q[3]}, Vector3{tr[0], tr[1], tr[2]}};
}

std::vector<std::pair<std::string, PropertyMut>> HipAlmeidaEstimator::props_mut() {       // lib.rs:80-98
    return {{"Use ransac", PropertyMut::boolean(&use_ransac)},
            {"Ransac iters", PropertyMut::usize(&num_iters, 1, 500)},
            {"Inlier threshold", PropertyMut::float_(&inlier_angle, 0.01f, 1.0f)},
            {"Ransac samples", PropertyMut::usize(&ransac_samples, 100, 16000)}};
}

// ------------------------------------------------------------------ input streams (ofps/src/utils.rs:92-118)
namespace {
class FdStreamBuf : public std::streambuf {                           // read side of a connected socket
public:
    explicit FdStreamBuf(int fd) : fd_(fd), buf_(1 << 16) {}
    ~FdStreamBuf() override { if (fd_ >= 0) ::close(fd_); }
protected:
    int_type underflow() override {
        if (gptr() < egptr()) return traits_type::to_int_type(*gptr());
        ssize_t n;
        do { n = ::recv(fd_, buf_.data(), buf_.size(), 0); } while (n < 0 && errno == EINTR);
        if (n <= 0) return traits_type::eof();
        setg(buf_.data(), buf_.data(), buf_.data() + n);
        return traits_type::to_int_type(*gptr());
    }
    std::streamsize xsgetn(char* s, std::streamsize count) override {  // large frame reads go straight to the caller's buffer
        std::streamsize got = 0;
        while (got < count) {
            if (gptr() < egptr()) {
                const std::streamsize k = std::min<std::streamsize>(count - got, egptr() - gptr());
                std::memcpy(s + got, gptr(), (size_t)k);
                gbump((int)k); got += k;
                continue;
            }
            ssize_t n;
            do { n = ::recv(fd_, s + got, (size_t)(count - got), 0); } while (n < 0 && errno == EINTR);
            if (n <= 0) break;
            got += n;
        }
        return got;
    }
private:
    int fd_;
    std::vector<char> buf_;
};
class FdIStream : public std::istream {
public:
    explicit FdIStream(int fd) : std::istream(nullptr), sb_(fd) { rdbuf(&sb_); }
private:
    FdStreamBuf sb_;
};
}  // namespace

std::unique_ptr<std::istream> open_file(const std::string& input) {
    const std::string prefix = "tcp://";
    if (input.rfind(prefix, 0) != 0) {                                // std::fs::File::open
        auto f = std::make_unique<std::ifstream>(input, std::ios::binary);
        if (!*f) throw Error("cannot open " + input);
        return f;
    }
    const std::string rest = input.substr(prefix.size());
    const auto colon = rest.find(':');                                // split_once(':')
    if (colon == std::string::npos) throw Error("Invalid format");
    const std::string addr = rest.substr(0, colon), port_s = rest.substr(colon + 1);
    char* end = nullptr;
    const unsigned long port = std::strtoul(port_s.c_str(), &end, 10);
    if (port_s.empty() || *end != 0 || port > 65535) throw Error("invalid port: " + port_s);
    int fd = -1;
    if (addr == "@") {                                                // TcpListener::bind("0.0.0.0:port") + accept
        const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) throw Error("socket() failed");
        int one = 1;
        ::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in sa{};
        sa.sin_family = AF_INET; sa.sin_addr.s_addr = htonl(INADDR_ANY); sa.sin_port = htons((uint16_t)port);
        if (::bind(ls, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) != 0 || ::listen(ls, 1) != 0) {
            ::close(ls);
            throw Error("cannot listen on port " + port_s);
        }
        sockaddr_in peer{};
        socklen_t pl = sizeof(peer);
        fd = ::accept(ls, reinterpret_cast<sockaddr*>(&peer), &pl);
        ::close(ls);
        if (fd < 0) throw Error("accept() failed");
        char ip[64] = {0};
        ::inet_ntop(AF_INET, &peer.sin_addr, ip, sizeof(ip));
        std::cout << "Accept " << ip << ":" << ntohs(peer.sin_port) << std::endl;
    } else {                                                          // TcpStream::connect(host:port)
        std::cout << "Connecting to " << rest << std::endl;
        addrinfo hints{}, *res = nullptr;
        hints.ai_family = AF_UNSPEC; hints.ai_socktype = SOCK_STREAM;
        if (::getaddrinfo(addr.c_str(), port_s.c_str(), &hints, &res) != 0 || !res) throw Error("cannot resolve " + addr);
        for (addrinfo* a = res; a && fd < 0; a = a->ai_next) {
            fd = ::socket(a->ai_family, a->ai_socktype, a->ai_protocol);
            if (fd >= 0 && ::connect(fd, a->ai_addr, a->ai_addrlen) != 0) { ::close(fd); fd = -1; }
        }
        ::freeaddrinfo(res);
        if (fd < 0) throw Error("cannot connect to " + rest);
    }
    std::cout << "Got stream!" << std::endl;
    return std::make_unique<FdIStream>(fd);
}

// ------------------------------------------------------------------ JSON + saved configuration
namespace {
struct JsonParser {
    const std::string& s;
    size_t p = 0;
    [[noreturn]] void fail(const char* what) const { throw Error(std::string("json: ") + what + " at offset " + std::to_string(p)); }
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < s.size() && s[p] == c) { ++p; return true; } return false; }
    void lit(const char* w) { for (const char* q = w; *q; ++q) { if (p >= s.size() || s[p] != *q) fail("bad literal"); ++p; } }
    std::string string_() {
        std::string out;
        if (!eat('"')) fail("expected string");
        while (true) {
            if (p >= s.size()) fail("unterminated string");
            const char c = s[p++];
            if (c == '"') return out;
            if (c != '\\') { out.push_back(c); continue; }
            if (p >= s.size()) fail("bad escape");
            const char e = s[p++];
            switch (e) {
                case 'n': out.push_back('\n'); break; case 't': out.push_back('\t'); break; case 'r': out.push_back('\r'); break;
                case 'b': out.push_back('\b'); break; case 'f': out.push_back('\f'); break;
                case 'u': {
                    if (p + 4 > s.size()) fail("bad \\u escape");
                    const unsigned cp = (unsigned)std::stoul(s.substr(p, 4), nullptr, 16); p += 4;
                    if (cp < 0x80) out.push_back((char)cp);
                    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    break;
                }
                default: out.push_back(e);                            // \" \\ \/
            }
        }
    }
    Json value() {
        ws();
        if (p >= s.size()) fail("unexpected end");
        Json j;
        const char c = s[p];
        if (c == '{') {
            ++p; j.kind = Json::Object;
            if (eat('}')) return j;
            do { std::string k = string_(); if (!eat(':')) fail("expected ':'"); j.obj.emplace_back(std::move(k), value()); } while (eat(','));
            if (!eat('}')) fail("expected '}'");
        } else if (c == '[') {
            ++p; j.kind = Json::Array;
            if (eat(']')) return j;
            do { j.arr.push_back(value()); } while (eat(','));
            if (!eat(']')) fail("expected ']'");
        } else if (c == '"') { j.kind = Json::String; j.str = string_(); }
        else if (c == 't') { lit("true"); j.kind = Json::Bool; j.b = true; }
        else if (c == 'f') { lit("false"); j.kind = Json::Bool; j.b = false; }
        else if (c == 'n') { lit("null"); }
        else {
            size_t used = 0;
            try { j.num = std::stod(s.substr(p), &used); } catch (...) { fail("bad number"); }
            j.kind = Json::Number; p += used;
        }
        return j;
    }
};

Property property_from_json(const Json& j) {                         // externally tagged: {"Float": {"val":..,"min":..,"max":..}}
    if (j.kind != Json::Object || j.obj.size() != 1) throw Error("config: a Property must be an object with one variant key");
    const auto& [tag, v] = j.obj[0];
    auto num = [&](const char* k) { const Json* m = v.get(k); if (!m || m->kind != Json::Number) throw Error(std::string("config: Property field ") + k); return m->num; };
    if (tag == "String" && v.kind == Json::String) return Property{v.str};
    if (tag == "Bool" && v.kind == Json::Bool) return Property{v.b};
    if (tag == "Float") return Property{BoundedProp<float>{(float)num("val"), (float)num("min"), (float)num("max")}};
    if (tag == "Usize") return Property{BoundedProp<size_t>{(size_t)num("val"), (size_t)num("min"), (size_t)num("max")}};
    throw Error("config: unknown Property variant " + tag);
}

std::vector<std::pair<std::string, Property>> property_map(const Json* j) {
    std::vector<std::pair<std::string, Property>> out;
    if (!j || j->kind == Json::Null) return out;                       // #[serde(default)]
    if (j->kind != Json::Object) throw Error("config: properties must be an object");
    for (const auto& [k, v] : j->obj) out.emplace_back(k, property_from_json(v));
    return out;
}

void plugin_tuple(const Json* j, const char* what, CreatePluginConfig& cfg, bool& flag) {
    if (!j || j->kind != Json::Array || j->arr.size() != 2 || j->arr[0].kind != Json::Object || j->arr[1].kind != Json::Bool)
        throw Error(std::string("config: ") + what + " must be [ {selected_plugin, arg, extra}, bool ]");
    const Json* sp = j->arr[0].get("selected_plugin");
    const Json* arg = j->arr[0].get("arg");
    if (!sp || sp->kind != Json::String || !arg || arg->kind != Json::String) throw Error(std::string("config: ") + what + ".selected_plugin / arg");
    cfg.selected_plugin = sp->str; cfg.arg = arg->str; flag = j->arr[1].b;
}
}  // namespace

const Json* Json::get(const std::string& key) const {
    for (const auto& kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
}

Json Json::parse(const std::string& text) {
    JsonParser ps{text};
    Json j = ps.value();
    ps.ws();
    if (ps.p != text.size()) ps.fail("trailing characters");
    return j;
}

MotionDetectionConfig MotionDetectionConfig::from_json(const std::string& text) {
    const Json root = Json::parse(text);
    if (root.kind != Json::Object) throw Error("config: expected an object");
    MotionDetectionConfig c;
    plugin_tuple(root.get("decoder"), "decoder", c.decoder, c.decoder_open);
    plugin_tuple(root.get("detector"), "detector", c.detector, c.detector_open);
    const Json* st = root.get("settings");
    if (!st || st->kind != Json::Object) throw Error("config: settings missing");
    const Json* worker = st->get("worker");
    if (!worker || worker->kind != Json::Object) throw Error("config: settings.worker missing");
    c.decoder_properties = property_map(worker->get("decoder_properties"));
    c.detector_properties = property_map(worker->get("detector_properties"));
    if (const Json* r = worker->get("realtime_processing"); r && r->kind == Json::Bool) c.realtime_processing = r->b;
    auto need_usize = [&](const char* k) { const Json* m = st->get(k); if (!m || m->kind != Json::Number) throw Error(std::string("config: settings.") + k); return (size_t)m->num; };
    if (const Json* o = st->get("overlay_mf"); o && o->kind == Json::Bool) c.overlay_mf = o->b; else throw Error("config: settings.overlay_mf");
    c.max_frame_gap = need_usize("max_frame_gap");
    c.min_frames = need_usize("min_frames");
    return c;
}

void transfer_props(const std::vector<std::pair<std::string, Property>>& saved, Properties& plugin) {
    for (const auto& [name, value] : saved) plugin.set_prop(name, value);   // unknown names are skipped, mismatched kinds ignored
}

// ------------------------------------------------------------------ creation by name
static std::unique_ptr<std::istream> open_input(const std::string& path) { return open_file(path); }

// ------------------------------------------------------------------ several GPUs behind one object
MultiDeviceSad::MultiDeviceSad(const std::vector<int>& devices) {
    const int rc = ofps_hip_multi_init(devices.data(), (int)devices.size(), &m_);
    if (rc != OFPS_HIP_OK) throw Error(std::string("ofps_hip_multi_init: ") + ofps_hip_multi_last_error(nullptr));
}
MultiDeviceSad::~MultiDeviceSad() { ofps_hip_multi_destroy(m_); }
int MultiDeviceSad::workers() const { return ofps_hip_multi_worker_count(m_); }
std::vector<MotionVectors> MultiDeviceSad::search(const uint8_t* frames, size_t n_frames, size_t w, size_t h, int block, int range, int ref_mode) {
    std::vector<MotionVectors> out;
    if (n_frames < 2) return out;
    const size_t nblk = ofps_hip_sad_block_count((int)w, (int)h, block);
    std::vector<float> ent((n_frames - 1) * nblk * 4);
    const int rc = ofps_hip_multi_sad_flow(m_, frames, (int)n_frames, (int)w, (int)h, (int)w, w * h, ref_mode, block, range, ent.data());
    if (rc != OFPS_HIP_OK) throw Error(std::string("ofps_hip_multi_sad_flow: ") + ofps_hip_multi_last_error(m_));
    out.resize(n_frames - 1);
    for (size_t k = 0; k + 1 < n_frames; ++k) {
        out[k].reserve(nblk);
        const float* e = ent.data() + k * nblk * 4;
        for (size_t i = 0; i < nblk; ++i) out[k].push_back(MotionEntry{e[4 * i], e[4 * i + 1], e[4 * i + 2], e[4 * i + 3]});
    }
    return out;
}

std::unique_ptr<Decoder> create_decoder(const std::string& name, const std::string& arg) {
    if (name == "mvec") return std::make_unique<MvecFileDecoder>(open_input(arg));
    if (name == "hip_sad" || name == "hip_lk" || name == "hip_flow") {                      // "<path>?w=1920&h=1080&fps=60"
        std::string path = arg;
        size_t w = 0, h = 0;
        int fmt = OFPS_HIP_FMT_LUMA;
        std::optional<double> fps;
        if (auto q = arg.find('?'); q != std::string::npos) {
            path = arg.substr(0, q);
            std::stringstream ss(arg.substr(q + 1));
            std::string kv;
            while (std::getline(ss, kv, '&')) {
                const auto eq = kv.find('=');
                if (eq == std::string::npos) continue;
                const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
                if (k == "w") w = std::stoul(v); else if (k == "h") h = std::stoul(v); else if (k == "fps") fps = std::stod(v);
                else if (k == "fmt") {                                  // hip_lk / hip_flow: the stream's pixel format (default luma)
                    if (v == "luma") fmt = OFPS_HIP_FMT_LUMA; else if (v == "bgr") fmt = OFPS_HIP_FMT_BGR;
                    else if (v == "rgba") fmt = OFPS_HIP_FMT_RGBA; else if (v == "bgra") fmt = OFPS_HIP_FMT_BGRA;
                    else throw Error("unknown frame format: " + v);
                }
            }
        }
        if (name == "hip_lk") return std::make_unique<HipLkDecoder>(open_input(path), w, h, fps, 0, false, fmt);
        if (name == "hip_flow") return std::make_unique<HipLkDecoder>(open_input(path), w, h, fps, 0, /*farneback=*/true, fmt);
        if (fmt != OFPS_HIP_FMT_LUMA) throw Error("hip_sad reads luma frames only");
        return std::make_unique<HipSadDecoder>(open_input(path), w, h, fps);
    }
    throw Error("unknown decoder plugin: " + name);
}
std::unique_ptr<Detector> create_detector(const std::string& name, const std::string&) {
    if (name == "hip_block_motion") return std::make_unique<HipBlockMotionDetection>();
    throw Error("unknown detector plugin: " + name);
}
std::unique_ptr<Estimator> create_estimator(const std::string& name, const std::string&) {
    if (name == "hip_almeida") return std::make_unique<HipAlmeidaEstimator>();
    throw Error("unknown estimator plugin: " + name);
}

// ------------------------------------------------------------------ harnesses
static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

DetectionRun run_detection(Decoder& decoder, Detector& detector, size_t max_frames) {
    DetectionRun run;
    MotionVectors mv;
    while (run.frames < max_frames) {
        mv.clear();                                                   // detection.rs:97
        auto t0 = std::chrono::steady_clock::now();
        try { decoder.process_frame(mv, nullptr, nullptr, 0); } catch (const Error&) { break; }   // :111-123
        run.frames += 1;
        run.decoder_ms.push_back(ms_since(t0));
        t0 = std::chrono::steady_clock::now();
        const auto motion = detector.detect_motion(mv.data(), mv.size());                        // :146-148
        run.detector_ms.push_back(ms_since(t0));
        if (motion) {                                                 // :150-155
            if (!run.motion_ranges.empty() && run.motion_ranges.back().second == run.frames) run.motion_ranges.back().second += 1;
            else run.motion_ranges.emplace_back(run.frames, run.frames + 1);
        }
    }
    return run;
}

std::vector<std::pair<size_t, size_t>> DetectionRun::filtered(size_t max_frame_gap, size_t min_frames) const {   // :195-212
    std::vector<std::pair<size_t, size_t>> merged;
    for (const auto& r : motion_ranges) {
        if (!merged.empty() && r.first - merged.back().second <= max_frame_gap) merged.back().second = r.second;
        else merged.push_back(r);
    }
    std::vector<std::pair<size_t, size_t>> out;
    for (const auto& r : merged)
        if (r.second - r.first >= min_frames) out.push_back(r);
    return out;
}

void export_perf_csv(const std::string& dir, const std::string& decoder_name,
                     const std::vector<std::pair<std::string, const std::vector<double>*>>& stages) {   // perf_stats.rs:86-121
    for (const auto& [name, times] : stages) {
        std::ofstream f(dir + "/perf_" + name + "_" + decoder_name + ".csv");
        if (!f) throw Error("cannot write perf csv in " + dir);
        for (double ms : *times) f << (float)ms << "\n";
    }
}

TrackingRun run_tracking(Decoder& decoder, Estimator& estimator, const StandardCamera& camera, size_t max_frames) {
    TrackingRun run;
    MotionVectors mv;
    UnitQuaternion rot;                                               // poses start at identity
    while (run.frames < max_frames) {
        mv.clear();
        auto t0 = std::chrono::steady_clock::now();
        bool have;
        try { have = decoder.process_frame(mv, nullptr, nullptr, 0); } catch (const Error&) { break; }
        run.decoder_ms.push_back(ms_since(t0));
        run.frames += 1;
        if (!have) { run.rotations.push_back(rot); run.estimator_ms.push_back(0.0); continue; }
        t0 = std::chrono::steady_clock::now();
        const auto [frot, tr] = estimator.estimate(mv.data(), mv.size(), camera, std::nullopt);   // worker.rs:361
        run.estimator_ms.push_back(ms_since(t0));
        (void)tr;                                                     // Almeida: translation is always 0
        rot = frot * rot;                                             // apply_pose, worker.rs:62-69
        run.rotations.push_back(rot);
    }
    return run;
}

}  // namespace ofps
