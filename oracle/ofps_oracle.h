/*
 * ofps_oracle.h -- CPU restatement of the OFPS flow hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X backend.  It restates, in plain C with f32
 * arithmetic in the same operation order, the reference functions listed below.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it;
 * the product path (libofps_hip.so) never does.
 *
 * PARITY STATUS (tests/test_reference_vectors.py, tests/test_oracle.py)
 *   - StandardCamera::delta (A6-A8): pinned by REFERENCE-HELD vectors -- the seven 144-row tables of
 *     docs/report/mfield (base, este, 0..4 .csv) are reproduced to 1.5e-6 (camera 16/9, 99 deg; the divide by NDC z of
 *     camera.rs:77 included) -- plus the point_angle doctest (ofps/src/camera.rs:139-149).
 *   - Almeida estimator (A9-A12): the reference's own known-answer test (almeida-estimator/src/lib.rs:253-373,
 *     32 rotations, error < 10 % of the rotation) on fields built by the oracle AND by an independent float64 model;
 *     the planted rotation of the reference-held base.csv recovered to 0.01 deg; the report-time iteration
 *     (este.csv, 0..4.csv: alpha 0.3, 5 steps, order yaw*pitch*roll) reproduced to the f32 noise of the
 *     eps-prototypes (4e-4) through orc_almeida_model.  Not bit-pinned: no Rust toolchain, and the reference's
 *     RANSAC is unseeded.
 *   - MotionFieldDensifier, BlockMotionDetection (A1-A5): the reference holds no test or fixture for them ->
 *     "parity unpinned" by the reference; pinned by hand-derived literals (cells + f32 bit patterns written from the
 *     Rust text), a second, independent NumPy restatement (oracle/np_oracle.py) and committed golden vectors.
 *   - SAD full-search block matcher, pyramidal LK: NO reference counterpart;
 *     the specs are defined in DESIGN.md ("N1", "N2") -> "parity unpinned".
 *   The Rust reference cannot be built here (no rustc/cargo), so no oracle/_ref exists.
 *
 * Matrices are row-major float[16] / float[9].  Quaternions are (w, i, j, k).
 * A MotionEntry is 4 consecutive floats [pos.x, pos.y, motion.x, motion.y]
 * (ofps/src/decoder.rs:40-42).
 */
#ifndef OFPS_ORACLE_H
#define OFPS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- StandardCamera (ofps/src/camera.rs:12-35) ---- */
typedef struct {
    float aspect;   /* camera.rs:13 */
    float fov_y;    /* degrees, camera.rs:14 */
    float m00, m11, m22, m23;      /* Perspective3 non-trivial entries (m32 = -1) */
    float r00, r11, r32, r33;      /* Perspective3::inverse non-trivial entries (r23 = -1) */
} orc_camera;

void orc_camera_new(orc_camera* c, float aspect, float fov_y_deg);          /* camera.rs:26-35 */
void orc_camera_unproject(const orc_camera* c, const float p[2], const float inv_view[16],
                          float out[3]);                                   /* camera.rs:45-55 */
void orc_camera_project(const orc_camera* c, const float world[3], const float view[16],
                        float out[2]);                                     /* camera.rs:72-81 */
void orc_camera_rotate(const orc_camera* c, const float p[2], const float rot[16],
                       float out[2]);                                      /* camera.rs:89-112 */
void orc_camera_delta(const orc_camera* c, const float p[2], const float rot[16],
                      float out[2]);                                       /* camera.rs:115-117 */
void orc_camera_point_angle(const orc_camera* c, const float p[2], float out[2]); /* :150-161 */

/* ---- nalgebra helpers used on the path ---- */
void orc_mat4_from_euler(float roll, float pitch, float yaw, float out[16]);
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);
void orc_mat4_transform_point(const float m[16], const float p[3], float out[3]);
void orc_mat4_look_at_rh(const float eye[3], const float target[3], const float up[3], float out[16]);
void orc_quat_from_euler(float roll, float pitch, float yaw, float q[4]);
void orc_quat_mul(const float a[4], const float b[4], float out[4]);
void orc_quat_to_homogeneous(const float q[4], float out[16]);
void orc_quat_inverse(const float q[4], float out[4]);
void orc_quat_transform_vector(const float q[4], const float v[3], float out[3]);
float orc_quat_angle_to(const float a[4], const float b[4]);
int  orc_lu3_solve(const float a[9], const float b[3], float x[3]);  /* 1 = solved, 0 = singular */

/* ---- MotionFieldDensifier / MotionField (ofps/src/motion_field.rs) ---- */
/* add_vector for every entry then MotionField::from (motion_field.rs:133-190, 297-308).
 * out_field: 2*w*h floats, cell (x,y) at [2*(y*w+x)], as MotionField::as_slice (:42-49).
 * out_cells (optional): 2*n uint32 (x,y) = the return value of add_vector per entry.
 * out_counts (optional): 2*w*h floats, the densifier's counts matrix before the divide. */
void orc_densify(const float* entries, size_t n, int w, int h,
                 float* out_field, uint32_t* out_cells, float* out_counts);

/* cv-decoder's downsample-through-densifier output stage (cv-decoder/src/lib.rs:244-291):
 * densify to (w,h), then emit one entry per visited cell in BTreeSet<(x,y)> order
 * (x-major), pos = ((x+.5)/w, (y+.5)/h), motion = cell average.  Returns entry count. */
void orc_densify_weighted(const float* entries, const float* weights, size_t n, int w, int h, float* out_field,
                          uint32_t* out_cells);                                     /* motion_field.rs:164-178 */
size_t orc_densify_to_entries(const float* entries, size_t n, int w, int h, float* out_entries);

/* MotionFieldDensifier::interpolate_empty_cells (motion_field.rs:193-294) followed by
 * MotionField::from; same output layout as orc_densify. */
void orc_densify_interpolated(const float* entries, size_t n, int w, int h, float* out_field);

/* ---- BlockMotionDetection::detect_motion (block-motion-detector/src/lib.rs:49-118) ---- */
int orc_block_dim(float min_size, size_t subdivide);                        /* lib.rs:53-54 */
/* out_field must hold 2*dim*dim floats (dim <= 160*... use orc_block_dim).  Returns 1 for
 * Some((area, field)), 0 for None.  out_field is zero-filled when None. */
int orc_detect_motion(const float* entries, size_t n, float min_size, size_t subdivide,
                      float target_motion, size_t* out_area, int* out_dim, float* out_field);

/* ---- Almeida estimator (almeida-estimator/src/lib.rs) ---- */
void orc_solve_ypr_given(const float* entries, size_t n, const orc_camera* cam, float q[4]); /* :123-200 */
/* the same loop with ALPHA, the step count and the composition order (0 = pitch*roll*yaw, today's :193; 1 =
 * yaw*pitch*roll, the build behind docs/report/mfield) as arguments; q_steps (optional) = rotation after each step */
/* one pass of the loop body (:140-183): raw LU solution in units of EPS, zero when the LU fails */
void orc_almeida_model(const float* entries, size_t n, const orc_camera* cam, const float rotation[4], float model[3]);
void orc_solve_ypr_given_ex(const float* entries, size_t n, const orc_camera* cam, float alpha, size_t limit,
                            int order, float* q_steps, float q[4]);
/* :202-251 with the build's counter-based sampler in place of rand::thread_rng (the
 * reference is unseeded).  out_inliers (optional) receives the indices of the
 * best inlier set (capacity num_samples), *out_n_inliers its size. */
void orc_solve_ypr_ransac(const float* entries, size_t n, const orc_camera* cam,
                          size_t num_iters, float inlier_deg, size_t num_samples, uint64_t seed,
                          float q[4], uint32_t* out_inliers, size_t* out_n_inliers);
/* The two solvers on `threads` host threads, same bits as the single-thread forms (the per-vector loops and the
 * hypotheses are independent; every sum keeps its sequential order): the all-core CPU baseline of bench.py. */
void orc_solve_ypr_given_mt(const float* entries, size_t n, const orc_camera* cam, int threads, float q[4]);
void orc_solve_ypr_ransac_mt(const float* entries, size_t n, const orc_camera* cam, size_t num_iters, float inlier_deg,
                             size_t num_samples, uint64_t seed, int threads, float q[4]);
/* The sampler itself (shared definition with the HIP kernels, DESIGN.md "RANSAC sampler"):
 * i-th of the distinct indices drawn for (seed, iter, stream) out of [0,n). */
uint32_t orc_sample_index(uint64_t seed, uint32_t iter, uint32_t stream, uint32_t i, uint32_t n);

/* ---- N1: full-search SAD block matcher (build-defined spec) ---- */
/* prev/cur: H rows of `stride` bytes, W valid.  Blocks: B x B lattice from (0,0), full blocks
 * only.  Candidates (dx,dy) in [-R,R]^2 whose block lies inside the frame.  Winner = min of
 * (SAD, dx*dx+dy*dy, dy+R, dx+R) lexicographic.  out_entries: 4 floats per block, raster
 * order; out_best (optional): 3 int32 per block (dx, dy, sad).  Returns number of blocks.
 * threads <= 1 -> scalar single thread; otherwise OpenMP over block rows. */
int orc_sad_simd_level(void);   /* inner loop of the timed baseline on this host: 2 = AVX2 vmpsadbw, 1 = SSE2 psadbw, 0 = scalar */
size_t orc_sad_flow(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride,
                    int B, int R, float* out_entries, int32_t* out_best, int threads);

/* simd = 0 forces the scalar definition loop; simd = 1 allows the exact SSE2 psadbw shortcut. */
size_t orc_sad_flow_ex(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride,
                       int B, int R, float* out_entries, int32_t* out_best, int threads, int simd);

/* ---- N2: dense pyramidal Lucas-Kanade flow (build-defined spec; no reference arithmetic exists:
 * the reference calls OpenCV's calcOpticalFlowFarneback, cv-decoder/src/lib.rs:188-199) -> "parity unpinned".
 * out_flow: 2*W*H floats (u,v) per pixel, prev(x,y) ~ cur(x+u,y+v).  Returns 1, or 0 on bad parameters. */
int orc_lk_spec_revision(void);   /* 2: fused multiply-adds in the bilinear sample and the residual sums (current); 1: unfused */
int orc_lk_flow(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int radius, int iters,
                float* out_flow);
/* the same with a caller-supplied starting flow for the COARSEST level (2 floats per pixel of that level, whose size is
 * W, H halved (rounding up) levels - 1 times; NULL = zero = orc_lk_flow): a temporal prior, the role of OpenCV's
 * OPTFLOW_USE_INITIAL_FLOW -- and the way the parity tests put chosen flows into a level's first step */
int orc_lk_flow_init(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int radius, int iters,
                     const float* init, float* out_flow);
/* ... and with the flow ENTERING every Gauss-Newton step written to `trace` (coarsest level first; per level `iters` planes
 * of 2 * w * h floats): what tools/lk_tile_stats.py and the tile-grouping tests look at (which tiles' sample rectangles fit) */
int orc_lk_flow_trace(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int radius, int iters,
                      const float* init, float* out_flow, float* trace);
/* per-pixel MotionEntry records in cv-decoder's convention (cv-decoder/src/lib.rs:239-243,262-269) */
void orc_flow_to_entries(const float* flow, int W, int H, float* out_entries);

/* cv-decoder's Sobel(1,1,k5) > 20 -> dilate(ellipse 11x11) contrast mask (cv-decoder/src/lib.rs:203-237), restated
 * from OpenCV's published definitions ("parity unpinned").  out_mask: W*H bytes, 1 = pixel contributes a record. */
void orc_contrast_mask(const uint8_t* gray, int W, int H, int stride, uint8_t* out_mask);
/* records of the unmasked pixels in raster order (mask may be NULL = all); returns the count */
size_t orc_masked_flow_to_entries(const float* flow, const uint8_t* mask, int W, int H, float* out_entries);

int orc_num_threads(void);
int orc_set_num_threads(int n);   /* for the OpenMP loops without a thread argument (orc_lk_flow*); returns the previous setting */

/* ---- Farneback dense flow as cv-decoder calls it (cv-decoder/src/lib.rs:188-199; OpenCV's calcOpticalFlowFarneback restated from the
 * published algorithm: farneback_oracle.c).  PARITY UNPINNED: OpenCV is neither under /root/reference nor installed. ---- */
/* 0 (default, the SPEC: symmetric pairing) / 1 (ascending row taps beyond 5 taps) / 2 (1 + fused multiply-adds): which published form of
 * OpenCV's separable Gaussian blurs the layers -- for the external kit only; returns the previous setting */
int orc_farneback_set_blur_variant(int v);
int orc_farneback_layers(int W, int H, int levels);                       /* highest layer index k kept (layers k = 0 .. result) */
void orc_farneback_layer_size(int W, int H, int k, int* w, int* h);
int orc_farneback_blur_kernel(int k, float* taps);                        /* -> radius; taps[2 r + 1] */
void orc_farneback_poly_kernel(int n, double sigma, float* g, float* xg, float* xxg, double ig[4]);
int orc_farneback_layer_debug(const uint8_t* img, int W, int H, int stride, int k, int poly_n, double poly_sigma, float* out_I, float* out_R);
int orc_farneback_flow(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int winsize, int iters, int poly_n,
                       double poly_sigma, const float* init, float* out_flow);

/* ---- cv-decoder's frame front-end (cv-decoder/src/lib.rs:98-135; OpenCV's 8-bit resize(INTER_LINEAR) and cvtColor(BGR2GRAY) restated from
 * the published code: frontend_oracle.c).  PARITY UNPINNED: OpenCV is neither under /root/reference nor installed. ---- */
void orc_cv_grid(int W, int H, int max_w, int max_h, int* gw, int* gh);     /* the capped record grid of :98-121 */
int orc_resize_linear_u8(const uint8_t* src, int W, int H, int stride, int cn, uint8_t* dst, int dw, int dh);
int orc_resize_linear_u8_ex(const uint8_t* src, int W, int H, int stride, int cn, uint8_t* dst, int dw, int dh, int variant);
int orc_resize_linear_axis(int src, int dst, int edge_rule, int* ofs, short* coef);
int orc_to_gray_u8(const uint8_t* src, int W, int H, int stride, int fmt /* 1 BGR, 2 RGBA, 3 BGRA */, uint8_t* dst);

#ifdef __cplusplus
}
#endif
#endif
