// almeida.hip -- A6-A12: StandardCamera::delta + the Almeida rotation estimator on gfx950
// (ofps/src/camera.rs:26-161; almeida-estimator/src/lib.rs:100-251; trait ofps/src/estimator.rs:19-24).
//
// delta(): closed form of unproject -> rotate -> project.  With the fixed view matrix of
// camera.rs:91-96 every dropped term of the generic 4x4 products is an exact 0 or +-1 factor,
// so the closed form returns the same bits as the generic path (checked against the oracle).
// No FMA contraction (-ffp-contract=off), IEEE divides.
//
// Least squares (lib.rs:123-200), 30 damped Gauss-Newton steps on N entries:
//   * the three EPS prototypes (roll/pitch/yaw deltas, lib.rs:30-42) do not depend on the
//     iteration; they are computed once per entry and kept on chip (registers; LDS for 8 entries
//     per thread), and so is A = J^T J, which is built from the prototypes alone (small problems);
//   * per step each thread forms its products (6 unique A entries -- A is symmetric bit for bit
//     -- once, 3 b entries per step), a fixed-shape wave butterfly + LDS tree reduces them, thread 0 runs the
//     partial-pivot LU (nalgebra order) and the quaternion update, and the rotation is broadcast
//     through LDS.  The sum ORDER differs from the reference's sequential f32 sum, so results
//     agree to rounding (<= 1e-6 on the quaternion), not bit for bit;
//   * N <= 8192: one workgroup per problem does all 30 steps in one launch (almeida_lsq_wg_kernel);
//     larger N: one launch per step over a grid of workgroups whose partial sums are combined
//     in a fixed order by every workgroup at the start of the next step (almeida_lsq_step_kernel).
// MFMA is deliberately not used: the only contraction is J^T J with J in R^{2N x 3}.
//
// RANSAC (lib.rs:202-251): one thread per hypothesis solves its 3-sample problem sequentially (same
// op order as the oracle), one workgroup per hypothesis counts inliers over the drawn samples,
// the best hypothesis' inliers are compacted in sample order and re-solved.  The reference's
// rand::thread_rng is replaced by the counter-based sampler defined in oracle/ofps_oracle.c
// (orc_sample_index): keyed 4-round Feistel permutation with cycle walking.
#include "common.hpp"

#include <cmath>
#include <mutex>
#include <vector>

namespace ofps {

struct Camera {            // same fields as the oracle's orc_camera (camera.rs:26-35)
    float aspect, fov_y;
    float m00, m11, m22, m23;
    float r00, r11, r32, r33;
};

static float to_radians_host(float deg) {
    const float k = 3.14159265358979323846264338327950288f / 180.0f;
    return deg * k;
}

static Camera camera_new(float aspect, float fov_y_deg) {   // Perspective3::new + inverse
    Camera c;
    const float zn = 0.1f, zf = 10.0f;
    const float fovy = to_radians_host(fov_y_deg);
    c.aspect = aspect; c.fov_y = fov_y_deg;
    c.m11 = 1.0f / tanf(fovy / 2.0f);
    c.m00 = c.m11 / aspect;
    c.m22 = (zf + zn) / (zn - zf);
    c.m23 = zf * zn * 2.0f / (zn - zf);
    c.r00 = 1.0f / c.m00;
    c.r11 = 1.0f / c.m11;
    c.r32 = 1.0f / c.m23;
    c.r33 = c.m22 * c.r32;
    return c;
}

struct Mat3 { float m[9]; };   // row-major 3x3 rotation

// camera.rs:115-117 (rotate - coords), closed form; see file header.
__device__ __forceinline__ float2 cam_delta(const Camera& c, float px, float py, const Mat3& R) {
    const float cx = px * 2.0f - 1.0f, cy = py * 2.0f - 1.0f;
    const float n0 = c.r32 + c.r33;
    const float wx = ((-c.r00) * cx) / n0;
    const float wy = -1.0f / n0;
    const float wz = (c.r11 * cy) / n0;
    const float rx = (R.m[0] * wx + R.m[1] * wy) + R.m[2] * wz;
    const float ry = (R.m[3] * wx + R.m[4] * wy) + R.m[5] * wz;
    const float rz = (R.m[6] * wx + R.m[7] * wy) + R.m[8] * wz;
    const float qx = -rx, qy = rz, qz = ry;                   // view: (-x, z, y)
    const float inv = -1.0f / qz;
    const float sx = c.m00 * qx * inv;
    const float sy = c.m11 * qy * inv;
    const float sz = (c.m22 * qz + c.m23) * inv;
    const float ox = (sx / sz + 1.0f) * 0.5f;                 // camera.rs:77: divide by NDC z
    const float oy = (sy / sz + 1.0f) * 0.5f;
    return make_float2(ox - px, oy - py);
}

// a / b.  EXACT = IEEE division (what the reference computes; ~12 VALU instructions on gfx950).  The dense-field
// solver (n > 65536, the per-pixel regime of cfg3) uses v_rcp_f32 * a instead (<= 1 ulp per quotient, 2 instructions):
// its 15 divisions per entry per step are 2/3 of that kernel's instruction stream, and with >= 65k entries per sum
// the extra half-ulp of per-quotient rounding noise averages out far below the 2e-6 the parity tests allow
// (north_star tolerance: 1e-4); the same regime fuses its multiply-adds (cam_delta_w<true>, the right-hand-side sums).
// Everything pinned to the oracle's exact operation order -- the one-workgroup
// solver, RANSAC hypotheses and inlier tests, block-vector sized problems -- keeps IEEE division.
template <bool FAST>
__device__ __forceinline__ float fdiv(float a, float b) {
    if constexpr (FAST) return a * __builtin_amdgcn_rcpf(b);
    else return a / b;
}

// The unprojected point (camera.rs:45-55) does not depend on the rotation: hoisted out of the 30-step loop.
struct Unproj { float wx, wy, wz; };
template <bool FAST = false>
__device__ __forceinline__ Unproj cam_unproject(const Camera& c, float px, float py) {
    const float cx = FAST ? __builtin_fmaf(px, 2.0f, -1.0f) : px * 2.0f - 1.0f, cy = FAST ? __builtin_fmaf(py, 2.0f, -1.0f) : py * 2.0f - 1.0f;
    const float n0 = c.r32 + c.r33;
    Unproj u;
    u.wx = fdiv<FAST>((-c.r00) * cx, n0);
    u.wy = fdiv<FAST>(-1.0f, n0);
    u.wz = fdiv<FAST>(c.r11 * cy, n0);
    return u;
}
// rotate + project + subtract: same operations, same order as cam_delta after its first four lines
template <bool FAST = false>
__device__ __forceinline__ float2 cam_delta_w(const Camera& c, float px, float py, const Unproj& u, const Mat3& R) {
    if constexpr (FAST) {
        // dense regime: fused multiply-adds on top of the reciprocal quotients (fdiv) -- 15 instead of 24 instructions for
        // the rotation + projection, each result rounded once instead of twice; inside the same 2e-6 parity bound
        const float rx = __builtin_fmaf(R.m[2], u.wz, __builtin_fmaf(R.m[1], u.wy, R.m[0] * u.wx));
        const float ry = __builtin_fmaf(R.m[5], u.wz, __builtin_fmaf(R.m[4], u.wy, R.m[3] * u.wx));
        const float rz = __builtin_fmaf(R.m[8], u.wz, __builtin_fmaf(R.m[7], u.wy, R.m[6] * u.wx));
        // project_point scales x, y and z by the same -1/qz and camera.rs:77 divides x and y by that z: the factor cancels,
        // (m00 * qx) / (m22 * qz + m23) -- one reciprocal instead of two and three multiplies less, fewer roundings
        const float isz = __builtin_amdgcn_rcpf(__builtin_fmaf(c.m22, ry, c.m23));
        const float ox = __builtin_fmaf((c.m00 * -rx) * isz, 0.5f, 0.5f);
        const float oy = __builtin_fmaf((c.m11 * rz) * isz, 0.5f, 0.5f);
        return make_float2(ox - px, oy - py);
    }
    const float rx = (R.m[0] * u.wx + R.m[1] * u.wy) + R.m[2] * u.wz;
    const float ry = (R.m[3] * u.wx + R.m[4] * u.wy) + R.m[5] * u.wz;
    const float rz = (R.m[6] * u.wx + R.m[7] * u.wy) + R.m[8] * u.wz;
    const float qx = -rx, qy = rz, qz = ry;
    const float inv = fdiv<FAST>(-1.0f, qz);
    const float sx = c.m00 * qx * inv;
    const float sy = c.m11 * qy * inv;
    const float sz = (c.m22 * qz + c.m23) * inv;
    const float ox = (fdiv<FAST>(sx, sz) + 1.0f) * 0.5f;
    const float oy = (fdiv<FAST>(sy, sz) + 1.0f) * 0.5f;
    return make_float2(ox - px, oy - py);
}

// Dense regime only.  For a fixed rotation, `delta` is a projective map of the record's position: with c = 2p - 1 the
// unprojected ray is (kx * c.x, wy, kz * c.y) (kx = -r00 / n0, kz = r11 / n0, wy = -1 / n0), each rotated component is
// affine in p, and the projection divides two of them by a third.  The nine coefficients below fold camera and rotation
// (the camera's factors in f64 on the host, one f32 product with the step's rotation matrix per coefficient), so
// a record costs three 2-term fused affine forms, one reciprocal and two fused scale-and-offsets:
//     X = -1/2 m00 (R row 0 . ray),  Y = 1/2 m11 (R row 2 . ray),  D = m22 (R row 1 . ray) + m23,
//     delta = (X / D + 1/2 - px,  Y / D + 1/2 - py)
// -- 11 instructions instead of ~24 for unproject + rotate + project, every operand rounded fewer times.  Worst
// deviation of the solve from the oracle's: tools/almeida_dense_margin.py.
struct DeltaAffine { float xa, xb, xc, ya, yb, yc, da, db, dc; };
// the camera's share of the nine coefficients: products of camera constants, formed in f64 on the host and rounded once
struct DeltaConsts { float x0, x1, x2, y0, y1, y2, d0, d1, d2, doff; };
static DeltaConsts delta_consts(const Camera& c) {
    const double n0 = (double)(c.r32 + c.r33);                     // the oracle's f32 sum
    const double kx = -(double)c.r00 / n0, kz = (double)c.r11 / n0, wy = -1.0 / n0;
    const double sx = -0.5 * (double)c.m00, sy = 0.5 * (double)c.m11, sd = (double)c.m22;
    DeltaConsts k;
    k.x0 = (float)(2.0 * sx * kx); k.x1 = (float)(sx * wy); k.x2 = (float)(2.0 * sx * kz);
    k.y0 = (float)(2.0 * sy * kx); k.y1 = (float)(sy * wy); k.y2 = (float)(2.0 * sy * kz);
    k.d0 = (float)(2.0 * sd * kx); k.d1 = (float)(sd * wy); k.d2 = (float)(2.0 * sd * kz);
    k.doff = c.m23;
    return k;
}
// rotated ray component j, scaled: (k0 R[j][0]) px + (k2 R[j][2]) py + (k1 R[j][1] - (k0 R[j][0] + k2 R[j][2]) / 2) [+ offset]
__device__ __forceinline__ DeltaAffine delta_affine(const DeltaConsts& k, const Mat3& R) {
    DeltaAffine A;
    A.xa = k.x0 * R.m[0]; A.xb = k.x2 * R.m[2]; A.xc = __builtin_fmaf(k.x1, R.m[1], -0.5f * (A.xa + A.xb));               // view x = -(row 0)
    A.ya = k.y0 * R.m[6]; A.yb = k.y2 * R.m[8]; A.yc = __builtin_fmaf(k.y1, R.m[7], -0.5f * (A.ya + A.yb));               // view y = row 2
    A.da = k.d0 * R.m[3]; A.db = k.d2 * R.m[5]; A.dc = __builtin_fmaf(k.d1, R.m[4], -0.5f * (A.da + A.db)) + k.doff;      // view z = row 1
    return A;
}
__device__ __forceinline__ float2 cam_delta_affine(const DeltaAffine& A, float px, float py) {
    const float X = __builtin_fmaf(A.xa, px, __builtin_fmaf(A.xb, py, A.xc));
    const float Y = __builtin_fmaf(A.ya, px, __builtin_fmaf(A.yb, py, A.yc));
    const float D = __builtin_fmaf(A.da, px, __builtin_fmaf(A.db, py, A.dc));
    const float inv = __builtin_amdgcn_rcpf(D);
    return make_float2(__builtin_fmaf(X, inv, 0.5f) - px, __builtin_fmaf(Y, inv, 0.5f) - py);
}
__device__ __forceinline__ DeltaAffine delta_affine_uniform(const DeltaAffine& a) {   // wave-uniform -> scalar registers
    DeltaAffine r;
    const float* s = &a.xa; float* d = &r.xa;
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s[k])));
    return r;
}

// the nine coefficients in LDS as three float4: written by one lane, read by every wave with all three reads in flight
__device__ __forceinline__ void aff_store(float4 (&d)[3], const DeltaAffine& a) {
    d[0] = make_float4(a.xa, a.xb, a.xc, a.ya); d[1] = make_float4(a.yb, a.yc, a.da, a.db); d[2] = make_float4(a.dc, 0.0f, 0.0f, 0.0f);
}
__device__ __forceinline__ DeltaAffine aff_load_uniform(const float4 (&d)[3]) {
    float4 v0 = d[0], v1 = d[1];
    float v2 = d[2].x;
    asm volatile("" : "+v"(v0.x), "+v"(v0.y), "+v"(v0.z), "+v"(v0.w), "+v"(v1.x), "+v"(v1.y), "+v"(v1.z), "+v"(v1.w), "+v"(v2));   // all nine loaded before the first is used
    const DeltaAffine a = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2};
    return delta_affine_uniform(a);
}

// Two records per instruction: gfx950's packed-f32 VALU forms (v_pk_fma_f32 / v_pk_add_f32) retire two IEEE fused
// multiply-adds per lane per issue slot -- each half rounds exactly like the scalar v_fma_f32 it replaces.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_splat(float x) { return v2f{x, x}; }

// camera.rs:150-161
__device__ __forceinline__ float2 cam_point_angle(float fx, float fy, float px, float py) {
    return make_float2(atanf((px - 0.5f) / fx), atanf((py - 0.5f) / fy));
}

struct alignas(16) Quat { float w, i, j, k; };      // (16-byte aligned: one ds_read_b128 / global dwordx4 per quaternion)

__device__ __forceinline__ Quat quat_mul(const Quat& a, const Quat& b) {         // nalgebra Hamilton product
    Quat r;
    r.w = a.w * b.w - a.i * b.i - a.j * b.j - a.k * b.k;
    r.i = a.w * b.i + a.i * b.w + a.j * b.k - a.k * b.j;
    r.j = a.w * b.j - a.i * b.k + a.j * b.w + a.k * b.i;
    r.k = a.w * b.k + a.i * b.j - a.j * b.i + a.k * b.w;
    return r;
}

__device__ __forceinline__ Quat quat_from_euler(float roll, float pitch, float yaw) {
    const float sr = sinf(roll * 0.5f), cr = cosf(roll * 0.5f);
    const float sp = sinf(pitch * 0.5f), cp = cosf(pitch * 0.5f);
    const float sy = sinf(yaw * 0.5f), cy = cosf(yaw * 0.5f);
    Quat q;
    q.w = cr * cp * cy + sr * sp * sy;
    q.i = sr * cp * cy - cr * sp * sy;
    q.j = cr * sp * cy + sr * cp * sy;
    q.k = cr * cp * sy - sr * sp * cy;
    return q;
}

__device__ __forceinline__ Mat3 quat_to_mat3(const Quat& q) {                    // to_homogeneous, 3x3 part
    const float w = q.w, i = q.i, j = q.j, k = q.k;
    const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    const float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
    const float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    Mat3 r;
    r.m[0] = ww + ii - jj - kk; r.m[1] = ij - wk;           r.m[2] = wj + ik;
    r.m[3] = wk + ij;           r.m[4] = ww - ii + jj - kk; r.m[5] = jk - wi;
    r.m[6] = ik - wj;           r.m[7] = wi + jk;           r.m[8] = ww - ii - jj + kk;
    return r;
}

// A wave-uniform matrix moved to scalar registers (v_readfirstlane): nine VGPRs less per matrix in kernels whose
// per-record state fills the register file; VALU instructions read the element as their scalar operand.
__device__ __forceinline__ Mat3 mat3_uniform(const Mat3& a) {
    Mat3 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.m[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.m[k])));
    return r;
}

// Matrix3::lu().solve.  a is row-major; returns false when a U diagonal is exactly 0.
// Rows live in named registers and the pivot exchange is a chain of compile-time-indexed conditional swaps: a
// run-time row index (m[3 * piv + c]) sends the whole matrix to scratch memory.
__device__ __forceinline__ void lu3_swap_rows(bool doit, float (&ra)[3], float (&rb)[3], float& ba, float& bb) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float t = ra[c]; ra[c] = doit ? rb[c] : ra[c]; rb[c] = doit ? t : rb[c]; }
    const float t = ba; ba = doit ? bb : ba; bb = doit ? t : bb;
}

__device__ __forceinline__ bool lu3_solve(const float a_in[9], const float b_in[3], float x[3]) {
    float r0[3] = {a_in[0], a_in[1], a_in[2]}, r1[3] = {a_in[3], a_in[4], a_in[5]}, r2[3] = {a_in[6], a_in[7], a_in[8]};
    float b0 = b_in[0], b1 = b_in[1], b2 = b_in[2];
    // ---- column 0: pivot = first row of maximal |.| (strict >), rows 0..2
    {
        int piv = 0;
        float best = fabsf(r0[0]);
        if (fabsf(r1[0]) > best) { best = fabsf(r1[0]); piv = 1; }
        if (fabsf(r2[0]) > best) { piv = 2; }
        const float diag = piv == 0 ? r0[0] : (piv == 1 ? r1[0] : r2[0]);
        if (diag != 0.0f) {
            lu3_swap_rows(piv == 1, r0, r1, b0, b1);
            lu3_swap_rows(piv == 2, r0, r2, b0, b2);
            const float inv_diag = 1.0f / diag;
            r1[0] *= inv_diag; r2[0] *= inv_diag;
#pragma unroll
            for (int c = 1; c < 3; ++c) {
                const float neg = -r0[c];
                r1[c] = neg * r1[0] + r1[c];
                r2[c] = neg * r2[0] + r2[c];
            }
        }
    }
    // ---- column 1: rows 1..2
    {
        const bool p2 = fabsf(r2[1]) > fabsf(r1[1]);
        const float diag = p2 ? r2[1] : r1[1];
        if (diag != 0.0f) {
            lu3_swap_rows(p2, r1, r2, b1, b2);
            const float inv_diag = 1.0f / diag;
            r2[1] *= inv_diag;
            const float neg = -r1[2];
            r2[2] = neg * r2[1] + r2[2];
        }
    }
    // ---- column 2: nothing below the diagonal; a zero diagonal is caught by the back substitution
    // forward substitution with the unit-lower factor (nalgebra divides by the unit diagonal: b / 1)
    {
        const float c0 = b0 / 1.0f;
        b1 = (-c0) * r1[0] + b1;
        b2 = (-c0) * r2[0] + b2;
        const float c1 = b1 / 1.0f;
        b2 = (-c1) * r2[1] + b2;
    }
    // back substitution with U
    if (r2[2] == 0.0f) return false;
    const float x2 = b2 / r2[2];
    b0 = (-x2) * r0[2] + b0;
    b1 = (-x2) * r1[2] + b1;
    if (r1[1] == 0.0f) return false;
    const float x1 = b1 / r1[1];
    b0 = (-x1) * r0[1] + b0;
    if (r0[0] == 0.0f) return false;
    const float x0 = b0 / r0[0];
    x[0] = x0; x[1] = x1; x[2] = x2;
    return true;
}

// NOTE on the permutation: nalgebra permutes b by the recorded swaps before the triangular
// solves; the forward substitution above has not started when a swap is recorded, and the swaps
// are applied in recording order, so permuting b on the fly is the same sequence of exchanges.

// The normal matrix A does not change from step to step, so neither does its factorisation: lu3_factor is lu3_solve's
// elimination (the same operations on the same numbers), lu3_apply its row exchanges of b, forward and back substitution.
struct Lu3 { float r0[3], r1[3], r2[3]; int piv0, p2; float inv[3]; };   // L multipliers below the diagonal, U on and above it; inv[k] = 1 / U[k][k]
// (IEEE quotients, formed once per solve: the back substitution of each of the 30 steps multiplies by them -- one rounding more than the
// division it replaces, 1e-7 of a step that is itself 1e-5 rad: six orders inside the parity bound, and ~27 dependent instructions less
// on every step's serial chain)
__device__ __forceinline__ Lu3 lu3_factor(const float a_in[9]) {
    Lu3 f;
    float r0[3] = {a_in[0], a_in[1], a_in[2]}, r1[3] = {a_in[3], a_in[4], a_in[5]}, r2[3] = {a_in[6], a_in[7], a_in[8]};
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;                          // stand-ins for b in the shared row exchange helper
    f.piv0 = 0; f.p2 = 0;
    {
        int piv = 0;
        float best = fabsf(r0[0]);
        if (fabsf(r1[0]) > best) { best = fabsf(r1[0]); piv = 1; }
        if (fabsf(r2[0]) > best) { piv = 2; }
        const float diag = piv == 0 ? r0[0] : (piv == 1 ? r1[0] : r2[0]);
        if (diag != 0.0f) {
            f.piv0 = piv;
            lu3_swap_rows(piv == 1, r0, r1, d0, d1);
            lu3_swap_rows(piv == 2, r0, r2, d0, d2);
            const float inv_diag = 1.0f / diag;
            r1[0] *= inv_diag; r2[0] *= inv_diag;
#pragma unroll
            for (int c = 1; c < 3; ++c) {
                const float neg = -r0[c];
                r1[c] = neg * r1[0] + r1[c];
                r2[c] = neg * r2[0] + r2[c];
            }
        }
    }
    {
        const bool p2 = fabsf(r2[1]) > fabsf(r1[1]);
        const float diag = p2 ? r2[1] : r1[1];
        if (diag != 0.0f) {
            f.p2 = p2 ? 1 : 0;
            lu3_swap_rows(p2, r1, r2, d1, d2);
            const float inv_diag = 1.0f / diag;
            r2[1] *= inv_diag;
            const float neg = -r1[2];
            r2[2] = neg * r2[1] + r2[2];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { f.r0[c] = r0[c]; f.r1[c] = r1[c]; f.r2[c] = r2[c]; }
    f.inv[0] = r0[0] != 0.0f ? 1.0f / r0[0] : 0.0f; f.inv[1] = r1[1] != 0.0f ? 1.0f / r1[1] : 0.0f; f.inv[2] = r2[2] != 0.0f ? 1.0f / r2[2] : 0.0f;
    return f;
}
template <bool FAST = false>
__device__ __forceinline__ bool lu3_apply(const Lu3& f, const float b_in[3], float x[3]) {
    float b0 = b_in[0], b1 = b_in[1], b2 = b_in[2];
    { float t; if (f.piv0 == 1) { t = b0; b0 = b1; b1 = t; } if (f.piv0 == 2) { t = b0; b0 = b2; b2 = t; } if (f.p2) { t = b1; b1 = b2; b2 = t; } }
    {
        const float c0 = b0 / 1.0f;
        b1 = (-c0) * f.r1[0] + b1;
        b2 = (-c0) * f.r2[0] + b2;
        const float c1 = b1 / 1.0f;
        b2 = (-c1) * f.r2[1] + b2;
    }
    if (f.r2[2] == 0.0f || f.r1[1] == 0.0f || f.r0[0] == 0.0f) return false;      // a zero on U's diagonal: lib.rs:181-183
    const float x2 = b2 * f.inv[2];
    b0 = (-x2) * f.r0[2] + b0;
    b1 = (-x2) * f.r1[2] + b1;
    const float x1 = b1 * f.inv[1];
    b0 = (-x1) * f.r0[1] + b0;
    const float x0 = b0 * f.inv[0];
    x[0] = x0; x[1] = x1; x[2] = x2;
    return true;
}

struct Protos { Mat3 roll, pitch, yaw; };

__device__ __forceinline__ Mat3 mat3_from_euler(float roll, float pitch, float yaw) {
    const float sr = sinf(roll), cr = cosf(roll), sp = sinf(pitch), cp = cosf(pitch), sy = sinf(yaw), cy = cosf(yaw);
    Mat3 r;
    r.m[0] = cy * cp; r.m[1] = cy * sp * sr - sy * cr; r.m[2] = cy * sp * cr + sy * sr;
    r.m[3] = sy * cp; r.m[4] = sy * sp * sr + cy * cr; r.m[5] = sy * sp * cr - cy * sr;
    r.m[6] = -sp;     r.m[7] = cp * sr;                r.m[8] = cp * cr;
    return r;
}

// One Gauss-Newton update from the 9 sums (lib.rs:159-195).  s = {a11,a12,a13,a22,a23,a33,b1,b2,b3}
__device__ __forceinline__ Quat almeida_update(const Quat& rotation, const float s[9], float eps, float alpha) {
    const float a[9] = {s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]};
    const float b[3] = {s[6], s[7], s[8]};
    float model[3];
    if (!lu3_solve(a, b, model)) { model[0] = model[1] = model[2] = 0.0f; }       // :181-183
    model[0] = model[0] * eps * alpha;                                            // :185
    model[1] = model[1] * eps * alpha;
    model[2] = model[2] * eps * alpha;
    // :189-191.  Each factor has a single non-zero Euler angle; with the other two half-angle sines
    // exactly 0 and cosines exactly 1, from_euler_angles reduces bit for bit to (cos, sin on one axis),
    // so two trig evaluations per factor instead of six (this section runs on one lane per step).
    float s0, c0, s1, c1, s2, c2;
    sincosf(model[0] * 0.5f, &s0, &c0);
    sincosf(model[1] * 0.5f, &s1, &c1);
    sincosf(-model[2] * 0.5f, &s2, &c2);
    const Quat roll = {c0, 0.0f, s0, 0.0f};     // from_euler_angles(0, m0, 0): pitch slot -> j
    const Quat pitch = {c1, s1, 0.0f, 0.0f};    // from_euler_angles(m1, 0, 0): roll slot  -> i
    const Quat yaw = {c2, 0.0f, 0.0f, s2};      // from_euler_angles(0, 0, -m2): yaw slot  -> k
    const Quat rot = quat_mul(quat_mul(pitch, roll), yaw);                        // :193
    return quat_mul(rotation, rot);                                               // :195
}

// sin and cos of a half-angle of one Gauss-Newton step.  These angles are tiny (a whole 10 degree rotation is 0.087 rad as
// a half-angle): below 0.25 rad the Taylor polynomials to x^7 / x^8 are exact to less than half an ulp of truncation error
// (next terms 1e-11 and 3e-13 relative) and cost ten fused operations instead of the library routine's argument
// reduction; anything larger takes the library routine.  The branch is wave-uniform.
__device__ __forceinline__ void sincos_small(float x, float& sn, float& cs) {
    if (__all(fabsf(x) < 0.25f)) {
        const float x2 = x * x;
        const float ps = __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, -1.0f / 5040.0f, 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
        sn = x * ps;
        cs = __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, 1.0f / 40320.0f, -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
    } else {
        sincosf(x, &sn, &cs);
    }
}
// The Gauss-Newton update run by a whole wave on a factorisation made once per solve (A is the same in every step): the
// three half-angle sincos evaluations go to lanes 0..2 in parallel and come back through v_readlane; everything else is
// wave-uniform.
template <bool FAST = false>
__device__ __forceinline__ Quat almeida_update_wave_lu(const Quat& rotation, const Lu3& f, float b0, float b1, float b2, float eps,
                                                       float alpha) {
    const float b[3] = {b0, b1, b2};
    float model[3];
    if (!lu3_apply<FAST>(f, b, model)) { model[0] = model[1] = model[2] = 0.0f; }       // :181-183
    // :185 and the half angles of :189-191 in one product per lane: alpha (0.5 or 1) and the 0.5 are powers of two, so
    // x * (eps * alpha * 0.5) carries the bits of ((x * eps) * alpha) * 0.5
    const int lane = threadIdx.x & 63;
    const float xsel = lane == 0 ? model[0] : (lane == 1 ? model[1] : -model[2]);
    const float half = xsel * (eps * alpha * 0.5f);
    float sn, cs;
    sincos_small(half, sn, cs);
    const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), 0)), c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), 0));
    const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), 1)), c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), 1));
    const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), 2)), c2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), 2));
    // :193, (pitch * roll) * yaw with pitch = (c1, s1, 0, 0), roll = (c0, 0, s0, 0), yaw = (c2, 0, 0, s2): the Hamilton products
    // with the terms that are exact zeros left out -- x * 0 is an exact zero and adding it changes nothing, so these are
    // quat_mul's values (16 operations instead of 56 on the step's serial chain; a lone wave issues one every ~5 cycles)
    // Sums of products are fused here (28 dependent instructions instead of 44): each sum is rounded once instead of after every
    // term -- closer to the exact product, 1e-8 from the unfused one.
    const Quat pr = {c1 * c0, s1 * c0, c1 * s0, s1 * s0};
    const Quat rot = {__builtin_fmaf(pr.w, c2, -(pr.k * s2)), __builtin_fmaf(pr.i, c2, pr.j * s2), __builtin_fmaf(pr.j, c2, -(pr.i * s2)),
                      __builtin_fmaf(pr.w, s2, pr.k * c2)};
    const Quat& a = rotation;
    Quat r;                                                                       // :195, quat_mul(rotation, rot)
    r.w = __builtin_fmaf(-a.k, rot.k, __builtin_fmaf(-a.j, rot.j, __builtin_fmaf(-a.i, rot.i, a.w * rot.w)));
    r.i = __builtin_fmaf(-a.k, rot.j, __builtin_fmaf(a.j, rot.k, __builtin_fmaf(a.i, rot.w, a.w * rot.i)));
    r.j = __builtin_fmaf(a.k, rot.i, __builtin_fmaf(a.j, rot.w, __builtin_fmaf(-a.i, rot.k, a.w * rot.j)));
    r.k = __builtin_fmaf(a.k, rot.w, __builtin_fmaf(-a.j, rot.i, __builtin_fmaf(a.i, rot.j, a.w * rot.k)));
    return r;
}

constexpr int kIters = 30;                      // ceil(15 / ALPHA), lib.rs:132
constexpr int kProfSlots = 7;                   // OFPS_HIP_ALMEIDA_PROF: stamps per step, all by the wave that carries the serial chain
__device__ __forceinline__ float almeida_eps() { return 0.001f * 3.14159265358979323846264338327950288f / 180.0f; }

// Wave-wide f32 sum on the DPP data path (cross-lane operands of ordinary VALU adds: no LDS round trip -- the
// __shfl_xor butterfly compiles to six dependent ds_bpermute_b32, ~700 cycles per value).  Fixed order: quad, quad pair,
// half row, row (every lane of a 16-lane row then holds its row's sum), row 0 -> row 1 and row 2 -> row 3 (row_bcast15),
// rows 0+1 -> rows 2,3 (row_bcast31); lane 63 holds the total.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float x) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false);
    return x + __int_as_float(moved);
}
__device__ __forceinline__ float row_sum16(float x) {        // every lane of a row <- sum over its 16-lane row
    x = dpp_add<0xB1>(x);                                     // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);                                     // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);                                    // row_half_mirror
    x = dpp_add<0x140>(x);                                    // row_mirror
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {         // -> total, uniform (read from lane 63)
    x = row_sum16(x);
    x = dpp_add<0x142, 0xA>(x);                               // row_bcast15 into rows 1 and 3
    x = dpp_add<0x143, 0xC>(x);                               // row_bcast31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// three sums at once: the same operations per value as wave_sum (same bits), issued in lockstep so that the DPP wait
// states of one chain are filled by the other two (hipcc emits three separate wave_sum calls back to back, 6 dependent
// DPP adds each)
__device__ __forceinline__ void row_sum16x3(float& a, float& b, float& c) {
    a = dpp_add<0xB1>(a); b = dpp_add<0xB1>(b); c = dpp_add<0xB1>(c);
    a = dpp_add<0x4E>(a); b = dpp_add<0x4E>(b); c = dpp_add<0x4E>(c);
    a = dpp_add<0x141>(a); b = dpp_add<0x141>(b); c = dpp_add<0x141>(c);
    a = dpp_add<0x140>(a); b = dpp_add<0x140>(b); c = dpp_add<0x140>(c);
}
__device__ __forceinline__ void wave_sum3(float& a, float& b, float& c) {
    row_sum16x3(a, b, c);
    a = dpp_add<0x142, 0xA>(a); b = dpp_add<0x142, 0xA>(b); c = dpp_add<0x142, 0xA>(c);
    a = dpp_add<0x143, 0xC>(a); b = dpp_add<0x143, 0xC>(b); c = dpp_add<0x143, 0xC>(c);
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
    c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));
}

// fixed-shape sum over a workgroup (up to 1024 threads) of the values v[K0..K1) of every thread; result valid in the first
// thread of wave W (in its lanes 0..15).  Each component is reduced independently, so reducing a sub-range gives the same bits as
// reducing all nine.
template <int K0, int K1, int W = 0>
__device__ __forceinline__ void block_sum(float v[9], float (*red)[9]) {
    static_assert((K1 - K0) % 3 == 0, "values are reduced three at a time");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = K0; k < K1; k += 3) {
        float a = v[k], b = v[k + 1], c = v[k + 2];
        wave_sum3(a, b, c);
        if (lane == 0) { red[wave][k] = a; red[wave][k + 1] = b; red[wave][k + 2] = c; }
    }
    __syncthreads();
    // second level: the first 16 lanes of wave W hold one wave-partial each and finish with a row sum
    if (wave == W) {
        const int nw = blockDim.x >> 6;
#pragma unroll
        for (int k = K0; k < K1; k += 3) {
            float a = (lane < nw) ? red[lane][k] : 0.0f, b = (lane < nw) ? red[lane][k + 1] : 0.0f, c = (lane < nw) ? red[lane][k + 2] : 0.0f;
            row_sum16x3(a, b, c);
            v[k] = a; v[k + 1] = b; v[k + 2] = c;
        }
    }
}
__device__ __forceinline__ void block_sum9(float v[9], float (*red)[9]) { block_sum<0, 9>(v, red); }

// The step loop's form for three values: every wave stops at its four 16-lane row sums (12 DPP adds instead of 18 + the
// two broadcasts and three v_readlane), the 4 * waves row sums go through LDS, and wave W alone runs the full wave tree
// over them.  -> a, b, c uniform in wave W (undefined elsewhere).  One barrier, like block_sum.
template <int W>
__device__ __forceinline__ void block_sum3_rows(float& a, float& b, float& c, float (*red4)[64]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    row_sum16x3(a, b, c);
    if ((lane & 15) == 0) { const int slot = wave * 4 + (lane >> 4); red4[0][slot] = a; red4[1][slot] = b; red4[2][slot] = c; }
    __syncthreads();
    if (wave == W) {
        const int nrow = (int)(blockDim.x >> 4);
        a = lane < nrow ? red4[0][lane] : 0.0f; b = lane < nrow ? red4[1][lane] : 0.0f; c = lane < nrow ? red4[2][lane] : 0.0f;
        wave_sum3(a, b, c);
    }
}

// ---- small problems: one workgroup per item, EPT entries per thread, all 30 steps in-kernel.
// n_dev (optional): per-item entry count on the device (RANSAC refit); stride = entries per item.
// The normal matrix A = J^T J is built from the three epsilon-prototypes only (lib.rs:159-173), which do not depend
// on the rotation being refined: the reference recomputes it in every step and gets the same numbers every time, so
// it is summed once here.  EPT = 8 keeps two of the three prototype pairs in LDS (128 KB) instead of registers:
// 13 floats x 8 entries per thread do not fit the 128 VGPRs a 1024-thread workgroup leaves (117 spilled before).
template <int EPT>
__global__ __launch_bounds__(1024) void almeida_lsq_wg_kernel(const float4* __restrict__ entries, size_t stride,
                                                              size_t n_fixed, const uint32_t* __restrict__ n_dev,
                                                              uint32_t min_n, Camera cam, float4* __restrict__ out_quat) {
    constexpr bool P_LDS = EPT >= 8;
    __shared__ float red[16][9];
    __shared__ float red4[3][64];                       // the steps' row sums (block_sum3_rows)
    __shared__ Quat rot_sh[2];
    __shared__ float4 plds[P_LDS ? EPT * 1024 : 1];     // (roll.x, roll.y, pitch.x, pitch.y) per entry
    const size_t item = blockIdx.x;
    const size_t n = n_dev ? (size_t)n_dev[item] : n_fixed;
    const float eps = almeida_eps();
    if (n < min_n) {                                   // lib.rs:247-251: fewer than 3 inliers -> identity
        if (threadIdx.x == 0) out_quat[item] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const Mat3 mroll = mat3_uniform(mat3_from_euler(0.0f, eps, 0.0f));     // lib.rs:30-34 (wave-uniform: scalar registers)
    const Mat3 mpitch = mat3_uniform(mat3_from_euler(eps, 0.0f, 0.0f));    // lib.rs:36-38
    const Mat3 myaw = mat3_uniform(mat3_from_euler(0.0f, 0.0f, -eps));     // lib.rs:40-42
    float4 e[EPT];
    float2 pr[P_LDS ? 1 : EPT], pp[P_LDS ? 1 : EPT], py[EPT];
    float uwx[EPT], uwz[EPT];                                       // Unproj::wy = -1/n0 is the same for every entry
    float uwy = 0.0f;
    bool ok[EPT];
    float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const size_t i = (size_t)t * 1024 + threadIdx.x;
        ok[t] = i < n;
        e[t] = ok[t] ? entries[item * stride + i] : make_float4(0.5f, 0.5f, 0.0f, 0.0f);
        const Unproj un = cam_unproject(cam, e[t].x, e[t].y);
        uwx[t] = un.wx; uwz[t] = un.wz; uwy = un.wy;
        const float2 r = cam_delta_w(cam, e[t].x, e[t].y, un, mroll);
        const float2 p = cam_delta_w(cam, e[t].x, e[t].y, un, mpitch);
        py[t] = cam_delta_w(cam, e[t].x, e[t].y, un, myaw);
        if constexpr (P_LDS) plds[t * 1024 + threadIdx.x] = make_float4(r.x, r.y, p.x, p.y);
        else { pr[t] = r; pp[t] = p; }
        if (ok[t]) {
            s[0] += r.x * r.x + r.y * r.y;
            s[1] += r.x * p.x + r.y * p.y;
            s[2] += r.x * py[t].x + r.y * py[t].y;
            s[3] += p.x * p.x + p.y * p.y;
            s[4] += p.x * py[t].x + p.y * py[t].y;
            s[5] += py[t].x * py[t].x + py[t].y * py[t].y;
        }
    }
    block_sum<0, 6>(s, red);
    __shared__ float a_sh[6];                                      // A leaves the registers: it would stay live through all 30 steps
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a_sh[k] = s[k];
    }
    __syncthreads();
    __shared__ Lu3 lu_sh;                                          // A is the same in every step: factorised once (wave 0)
    if (threadIdx.x < 64) {
        const float am[9] = {a_sh[0], a_sh[1], a_sh[2], a_sh[1], a_sh[3], a_sh[4], a_sh[2], a_sh[4], a_sh[5]};
        const Lu3 lu = lu3_factor(am);
        if (threadIdx.x == 0) lu_sh = lu;
    }
    Quat rotation = {1.0f, 0.0f, 0.0f, 0.0f};
    for (int it = 0; it < kIters; ++it) {
        const float alpha = (it == kIters - 1) ? 1.0f : 0.5f;      // lib.rs:138
        const Mat3 rotm = mat3_uniform(quat_to_mat3(rotation));    // lib.rs:140
        s[6] = 0.0f; s[7] = 0.0f; s[8] = 0.0f;
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            if (!ok[t]) continue;
            const Unproj un = {uwx[t], uwy, uwz[t]};
            const float2 d = cam_delta_w(cam, e[t].x, e[t].y, un, rotm);
            const float rx = e[t].z - d.x, ry = e[t].w - d.y;      // motion - delta
            float2 r, p;
            if constexpr (P_LDS) { const float4 v = plds[t * 1024 + threadIdx.x]; r = make_float2(v.x, v.y); p = make_float2(v.z, v.w); }
            else { r = pr[t]; p = pp[t]; }
            s[6] += r.x * rx + r.y * ry;
            s[7] += p.x * rx + p.y * ry;
            s[8] += py[t].x * rx + py[t].y * ry;
        }
        block_sum3_rows<0>(s[6], s[7], s[8], red4);               // (uniform in wave 0 afterwards)
        if (threadIdx.x < 64) {                                    // wave 0, all lanes: see almeida_update_wave_lu
            const Lu3 lu = lu_sh;                                  // (written by this very wave before the loop)
            const Quat q = almeida_update_wave_lu(rotation, lu, s[6], s[7], s[8], eps, alpha);
            if (threadIdx.x == 0) rot_sh[it & 1] = q;
        }
        __syncthreads();
        // one barrier per step: the slot alternates, so thread 0 cannot overwrite a rotation that a slow wave has yet
        // to read (it gets back to this slot only after everybody passed the next step's barrier), and `red` is not
        // written again before this barrier, which wave 0 reaches only after its second-level read
        rotation = rot_sh[it & 1];
    }
    if (threadIdx.x == 0) out_quat[item] = make_float4(rotation.w, -rotation.i, -rotation.j, -rotation.k);  // :199
}

// ---- large problems: one launch per step.  state[it & 1][item] holds the rotation entering step
// `it` (double-buffered: late workgroups of a launch still read the previous slot);
// partials[item][blk][9] are this step's per-workgroup sums.  Every workgroup first folds the
// previous step's partials (fixed order) into its private copy of the rotation.
template <bool FAST>
__global__ __launch_bounds__(1024) void almeida_lsq_step_kernel(const float4* __restrict__ entries, size_t n, int it,
                                                                Camera cam, const float* __restrict__ part_prev,
                                                                float* __restrict__ part_out, Quat* __restrict__ state,
                                                                int batch, float4* __restrict__ out_quat) {
    // dense variant: the first PRE entries of every thread are loaded, and everything about them that does not need the
    // new rotation (unprojection, the three prototypes, the A sums) is computed, BEFORE the workgroup waits for the
    // fold + update of the previous step -- that prologue is a chain of dependent global loads and ~400 serial
    // instructions on one lane and used to cost 4 of the 16 us of a step with every other wave idle.  Two of the three
    // prototype pairs wait in LDS (128 KB), the third and the entry in registers.
    constexpr int PRE = FAST ? 8 : 0;
    __shared__ float red[16][9];
    __shared__ float fold_sh[9];
    __shared__ Quat rot_sh;
    __shared__ float4 plds[FAST ? PRE * 1024 : 1];
    const size_t item = blockIdx.y;
    const int nblk = gridDim.x;
    const float eps = almeida_eps();
    const Mat3 mroll = mat3_from_euler(0.0f, eps, 0.0f);
    const Mat3 mpitch = mat3_from_euler(eps, 0.0f, 0.0f);
    const Mat3 myaw = mat3_from_euler(0.0f, 0.0f, -eps);
    const size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x, istride = (size_t)nblk * 1024;
    // ---- fold the previous step's partials, part 1: wave k sums entry k over the workgroups (lane-strided, then a
    // butterfly) -- a fixed order, identical in every workgroup, without a 9*nblk serial chain on one lane
    float facc = 0.0f;
    const bool folder = it > 0 && threadIdx.x < 9 * 64;
    if (folder) {
        const int k = threadIdx.x >> 6, l = threadIdx.x & 63;
        for (int b = l; b < nblk; b += 64) facc += part_prev[(item * nblk + b) * 9 + k];
    }
    Quat prev_rot = {1.0f, 0.0f, 0.0f, 0.0f};
    if (threadIdx.x == 0 && it > 0) prev_rot = state[(size_t)((it - 1) & 1) * batch + item];
    float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float4 pre[PRE > 0 ? PRE : 1];
    float2 ppy[PRE > 0 ? PRE : 1];
    float uwx[PRE > 0 ? PRE : 1], uwz[PRE > 0 ? PRE : 1];
    float uwy = 0.0f;
    if constexpr (FAST) {
        if (it < kIters) {
#pragma unroll
            for (int k = 0; k < PRE; ++k) {
                const size_t i = i0 + (size_t)k * istride;
                pre[k] = i < n ? entries[item * n + i] : make_float4(0.5f, 0.5f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int k = 0; k < PRE; ++k) {
                const float4 e = pre[k];
                const Unproj un = cam_unproject<FAST>(cam, e.x, e.y);
                uwx[k] = un.wx; uwz[k] = un.wz; uwy = un.wy;
                const float2 pr = cam_delta_w<FAST>(cam, e.x, e.y, un, mroll);
                const float2 pp = cam_delta_w<FAST>(cam, e.x, e.y, un, mpitch);
                const float2 py = cam_delta_w<FAST>(cam, e.x, e.y, un, myaw);
                ppy[k] = py;
                plds[k * 1024 + threadIdx.x] = make_float4(pr.x, pr.y, pp.x, pp.y);
                if (i0 + (size_t)k * istride < n) {
                    s[0] += pr.x * pr.x + pr.y * pr.y;
                    s[1] += pr.x * pp.x + pr.y * pp.y;
                    s[2] += pr.x * py.x + pr.y * py.y;
                    s[3] += pp.x * pp.x + pp.y * pp.y;
                    s[4] += pp.x * py.x + pp.y * py.y;
                    s[5] += py.x * py.x + py.y * py.y;
                }
            }
        }
    }
    // ---- fold, part 2 + the update on one lane
    if (folder) {
        facc = wave_sum(facc);
        if ((threadIdx.x & 63) == 0) fold_sh[threadIdx.x >> 6] = facc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Quat rotation = prev_rot;
        if (it > 0) {
            float f[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) f[k] = fold_sh[k];
            const float alpha = (it - 1 == kIters - 1) ? 1.0f : 0.5f;
            rotation = almeida_update(rotation, f, eps, alpha);
        }
        rot_sh = rotation;
    }
    __syncthreads();
    const Quat rotation = rot_sh;
    if (it == kIters) {                                  // epilogue launch: publish the result
        if (blockIdx.x == 0 && threadIdx.x == 0)
            out_quat[item] = make_float4(rotation.w, -rotation.i, -rotation.j, -rotation.k);
        return;
    }
    const Mat3 rotm = quat_to_mat3(rotation);
    if constexpr (FAST) {
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            if (i0 + (size_t)k * istride >= n) continue;
            const float4 e = pre[k];
            const Unproj un = {uwx[k], uwy, uwz[k]};
            const float2 d = cam_delta_w<FAST>(cam, e.x, e.y, un, rotm);
            const float rx = e.z - d.x, ry = e.w - d.y;
            const float4 v = plds[k * 1024 + threadIdx.x];
            s[6] += v.x * rx + v.y * ry;
            s[7] += v.z * rx + v.w * ry;
            s[8] += ppy[k].x * rx + ppy[k].y * ry;
        }
    }
    for (size_t i = i0 + (size_t)PRE * istride; i < n; i += istride) {
        const float4 e = entries[item * n + i];
        const Unproj un = cam_unproject<FAST>(cam, e.x, e.y);
        const float2 d = cam_delta_w<FAST>(cam, e.x, e.y, un, rotm);
        const float2 pr = cam_delta_w<FAST>(cam, e.x, e.y, un, mroll);
        const float2 pp = cam_delta_w<FAST>(cam, e.x, e.y, un, mpitch);
        const float2 py = cam_delta_w<FAST>(cam, e.x, e.y, un, myaw);
        const float rx = e.z - d.x, ry = e.w - d.y;
        s[0] += pr.x * pr.x + pr.y * pr.y;
        s[1] += pr.x * pp.x + pr.y * pp.y;
        s[2] += pr.x * py.x + pr.y * py.y;
        s[3] += pp.x * pp.x + pp.y * pp.y;
        s[4] += pp.x * py.x + pp.y * py.y;
        s[5] += py.x * py.x + py.y * py.y;
        s[6] += pr.x * rx + pr.y * ry;
        s[7] += pp.x * rx + pp.y * ry;
        s[8] += py.x * rx + py.y * ry;
    }
    block_sum9(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) part_out[(item * nblk + blockIdx.x) * 9 + k] = s[k];
        if (blockIdx.x == 0) state[(size_t)(it & 1) * batch + item] = rotation;   // same value in every workgroup
    }
}

// ---- cluster solver: ONE launch, all 30 steps, a problem spread over nblk co-resident workgroups (one per CU).
// Every thread keeps its EPT records, their hoisted unprojection and the three prototypes on chip for the whole solve
// (registers; two prototype pairs in LDS for EPT = 8), so the records cross HBM exactly once (33 MB at 1080p per-pixel
// instead of 30 x 33 MB re-streamed from the Infinity Cache by the launch-per-step kernel).  Per step the workgroups
// exchange their three right-hand-side partial sums (nine in step 0: A = J^T J is rotation-independent and summed
// once) as 8-byte {tag, f32} granules -- one write-through agent-scope store per value, the data is its own flag, no
// fence, no separate barrier (MI355X_MICROARCH.md "allgather", cdna_hip_programming.md G16 form R2) -- and EVERY
// workgroup folds all partials in the same fixed order and runs the same LU + quaternion update, so all of them hold
// bit-identical rotations without a broadcast.  Slots alternate with the step parity: a fast workgroup can be at most
// one step ahead of the slowest, so it never overwrites a granule somebody still has to read.  Tags = tag_base + step
// + 1 with a per-call tag_base (the buffer is zeroed when allocated, never between calls).  Every spin is bounded: on
// a timeout (the workgroups were not co-resident, e.g. another process holds CUs with a persistent kernel of its
// own) the workgroups leave the step loop and the last one to leave solves the item alone (almeida_solo_solve):
// no caller ever sees a NaN.
// A granule = ONE naturally aligned 16-byte write-through (sc0 sc1) store of three f32 sums.  Each 8-byte half carries
// the step's 16-bit tag, so a reader accepts a granule only when both halves belong to the step it waits for -- correct
// even if the two halves of the store became visible separately (not observed on gfx950, not an architectural promise
// either):   x = a,  y = tag << 16 | b[31:16],  z = c,  w = tag << 16 | b[15:0].
typedef unsigned int gran_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gran_store3(gran_u4* p, uint32_t tag16, float a, float b, float c) {
    const uint32_t bb = __float_as_uint(b);
    gran_u4 v;
    v.x = __float_as_uint(a); v.y = (tag16 << 16) | (bb >> 16); v.z = __float_as_uint(c); v.w = (tag16 << 16) | (bb & 0xFFFFu);
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
// J granule loads in flight, ONE wait (hipcc does not count the memory operations inside an asm statement, so the wait
// is part of it; outputs are early-clobber because the first load lands before the last address is consumed)
__device__ __forceinline__ void gran_load3x1(const gran_u4* p0, gran_u4 (&x)[4]) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x[0]) : "v"(p0) : "memory");
}
__device__ __forceinline__ void gran_load3x2(const gran_u4* p0, const gran_u4* p1, gran_u4 (&x)[4]) {
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]) : "v"(p0), "v"(p1) : "memory");
}
__device__ __forceinline__ void gran_load3x4(const gran_u4* p0, const gran_u4* p1, const gran_u4* p2, const gran_u4* p3,
                                             gran_u4 (&x)[4]) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}

// The same granule for readers on the WRITER'S XCD only: a plain store stays in that XCD's L2, where an `sc1` load (L1
// bypass) of a workgroup on the same XCD finds it -- 540 instead of 950 cycles per hand-off (tools/ubench_exchange); a
// reader on another XCD would never see it.
__device__ __forceinline__ void gran_store3_local(gran_u4* p, uint32_t tag16, float a, float b, float c) {
    const uint32_t bb = __float_as_uint(b);
    gran_u4 v;
    v.x = __float_as_uint(a); v.y = (tag16 << 16) | (bb >> 16); v.z = __float_as_uint(c); v.w = (tag16 << 16) | (bb & 0xFFFFu);
    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void gran_load3_local(const gran_u4* p0, gran_u4& x) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p0) : "memory");
}

constexpr unsigned kSpinLimit = 1u << 18;            // ~0.3 s of polling before giving up

// The item's fail word (holds the launch's tag once a workgroup's spin expired; see almeida_solo_solve).  Pollers look at
// it every 256th poll, so that workgroups which only became resident after the first ones left do not sit out a
// timeout of their own.
struct FailFlag { uint32_t* p; uint32_t tag; };
__device__ __forceinline__ bool fail_flag_up(const FailFlag& f) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(f.p) : "memory");
    return v == f.tag;
}
__device__ __forceinline__ bool spin_expired(unsigned spins, const FailFlag& f) {
    if (spins >= kSpinLimit) return true;
    return (spins & 255u) == 255u && fail_flag_up(f);
}

// same-XCD form of gran_sweep_sum3 for at most 64 granules
__device__ __forceinline__ bool gran_sweep_sum3_local(const gran_u4* g, int count, uint32_t tag16, const FailFlag& ff, float& ta, float& tb, float& tc, int gs = 1) {
    const int lane = threadIdx.x & 63;
    const gran_u4* p = g + (size_t)(lane < count ? lane : count - 1) * gs;
    gran_u4 x;
    for (unsigned spins = 0;; ++spins) {
        gran_load3_local(p, x);
        const bool ok = (x.y >> 16) == tag16 && (x.w >> 16) == tag16;
        if (__all(ok)) break;
        if (spin_expired(spins, ff)) return false;
    }
    const bool in = lane < count;
    ta = in ? __uint_as_float(x.x) : 0.0f;
    tb = in ? __uint_as_float((x.y << 16) | (x.w & 0xFFFFu)) : 0.0f;
    tc = in ? __uint_as_float(x.z) : 0.0f;
    wave_sum3(ta, tb, tc);
    return true;
}
// every workgroup's XCC_ID, published once per launch: all workgroups read all of them and reach the same verdict on
// whether workgroup b runs on XCD b % 8 (round-robin dispatch), the premise of the two-level gather.  false = timed out.
__device__ __forceinline__ bool xcc_sweep_check(const uint32_t* xccs, int nblk, uint32_t tag16, const FailFlag& ff, bool& round_robin, bool& one_xcd) {
    const int lane = threadIdx.x & 63;
    bool rr = true, same = true;
    uint32_t first = 0;
    for (int j0 = 0; j0 < nblk; j0 += 64) {
        const int idx = j0 + lane < nblk ? j0 + lane : nblk - 1;
        uint32_t v;
        for (unsigned spins = 0;; ++spins) {
            asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(xccs + idx) : "memory");
            if (__all((v >> 16) == tag16)) break;
            if (spin_expired(spins, ff)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        rr = rr && ((int)(v & 0xFFu) == (idx & 7));
        if (j0 == 0) first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v & 0xFFu));
        same = same && (v & 0xFFu) == first;
    }
    round_robin = __all(rr);
    one_xcd = __all(same);
    return true;
}

// wave-wide: component sums over the nblk (<= 256) granules of one triple once all carry `tag16`; fixed order
// (lane-strided, then the DPP tree), identical in every workgroup.  false = timed out.
template <int J>
__device__ __forceinline__ bool gran_sweep_sum3_j(const gran_u4* g, int gs, int nblk, uint32_t tag16, const FailFlag& ff, float& ta, float& tb, float& tc) {
    const int lane = threadIdx.x & 63;
    const int last = nblk - 1;
    const gran_u4* p[J];
#pragma unroll
    for (int j = 0; j < J; ++j) p[j] = g + (size_t)(lane + 64 * j <= last ? lane + 64 * j : last) * gs;   // out-of-range lanes re-read a valid one
    gran_u4 x[4];
    for (unsigned spins = 0;; ++spins) {
        if constexpr (J == 1) gran_load3x1(p[0], x);
        else if constexpr (J == 2) gran_load3x2(p[0], p[1], x);
        else gran_load3x4(p[0], p[1], p[2], p[3], x);
        bool ok = true;
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (64 * j < nblk) ok = ok && (x[j].y >> 16) == tag16 && (x[j].w >> 16) == tag16;
        if (__all(ok)) break;
        if (spin_expired(spins, ff)) return false;
        if (nblk > 64) __builtin_amdgcn_s_sleep(1);             // hundreds of pollers on the same lines: leave the channel some air (measured)
    }
    // lane-strided partial sums in the order ((x0 + x1) + x2) + x3; absent terms are exact zeros
    ta = 0.0f; tb = 0.0f; tc = 0.0f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const bool in = lane + 64 * j < nblk;
        const float va = in ? __uint_as_float(x[j].x) : 0.0f;
        const float vb = in ? __uint_as_float((x[j].y << 16) | (x[j].w & 0xFFFFu)) : 0.0f;
        const float vc = in ? __uint_as_float(x[j].z) : 0.0f;
        if (j == 0) { ta = va; tb = vb; tc = vc; }
        else { ta += va; tb += vb; tc += vc; }
    }
    wave_sum3(ta, tb, tc);
    return true;
}
__device__ __forceinline__ bool gran_sweep_sum3(const gran_u4* g, int gs, int nblk, uint32_t tag16, const FailFlag& ff, float& ta, float& tb, float& tc) {
    if (nblk <= 64) return gran_sweep_sum3_j<1>(g, gs, nblk, tag16, ff, ta, tb, tc);
    if (nblk <= 128) return gran_sweep_sum3_j<2>(g, gs, nblk, tag16, ff, ta, tb, tc);
    return gran_sweep_sum3_j<4>(g, gs, nblk, tag16, ff, ta, tb, tc);
}

// ---- what a cluster launch does when its workgroups were NOT all there (bounded spin expired): the reference's
// estimator cannot fail (singular LU -> zero step, almeida-estimator/src/lib.rs:181-185; < 3 inliers -> identity,
// :246-250), so neither may this one, on ANY entry point -- device-pointer callers never synchronise, so the recovery
// cannot live on the host.  A workgroup whose spin expired swaps the launch's tag into the item's fail word; the FIRST
// one to do so (the swap returns something else) solves the item alone, right away, from the records in memory -- it
// needs nothing from the others: 30 passes over all N records by one workgroup (milliseconds; after a 0.3 s stall nobody
// counts them).  Everybody else who gives up, or sees the word while polling, just leaves.  The path that completes
// pays nothing for this: no atomics, no extra barrier.  Same operations per record as the launch-per-step kernel (no
// folded delta), fixed summation order: inside the 2e-6 parity bound like every other path.  (A spin that expires in
// the LAST step while other workgroups still complete it leaves two writers of the same estimate -- the solo result
// and the cluster's, equal to rounding; whichever lands last stays.)
template <bool FAST, int BLOCK>
__device__ __forceinline__ void almeida_solo_solve(const float4* __restrict__ ent, size_t n, const Camera& cam, float (*red)[9],
                                                   Quat* rot_sh, float4* __restrict__ out) {
    float eps = almeida_eps();
    asm volatile("" : "+v"(eps));        // opaque: keeps the compiler from sharing the prototype matrices with the caller's step
                                         // loop, which would keep 27 registers alive through it for this cold path
    const Mat3 mroll = mat3_from_euler(0.0f, eps, 0.0f);
    const Mat3 mpitch = mat3_from_euler(eps, 0.0f, 0.0f);
    const Mat3 myaw = mat3_from_euler(0.0f, 0.0f, -eps);
    Quat rotation = {1.0f, 0.0f, 0.0f, 0.0f};
    for (int it = 0; it < kIters; ++it) {
        const float alpha = (it == kIters - 1) ? 1.0f : 0.5f;
        const Mat3 rotm = quat_to_mat3(rotation);
        float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
        for (size_t i = threadIdx.x; i < n; i += BLOCK) {
            const float4 e = ent[i];
            const Unproj un = cam_unproject<FAST>(cam, e.x, e.y);
            const float2 d = cam_delta_w<FAST>(cam, e.x, e.y, un, rotm);
            const float2 pr = cam_delta_w<FAST>(cam, e.x, e.y, un, mroll);
            const float2 pp = cam_delta_w<FAST>(cam, e.x, e.y, un, mpitch);
            const float2 py = cam_delta_w<FAST>(cam, e.x, e.y, un, myaw);
            const float rx = e.z - d.x, ry = e.w - d.y;
            s[0] += pr.x * pr.x + pr.y * pr.y;
            s[1] += pr.x * pp.x + pr.y * pp.y;
            s[2] += pr.x * py.x + pr.y * py.y;
            s[3] += pp.x * pp.x + pp.y * pp.y;
            s[4] += pp.x * py.x + pp.y * py.y;
            s[5] += py.x * py.x + py.y * py.y;
            s[6] += pr.x * rx + pr.y * ry;
            s[7] += pp.x * rx + pp.y * ry;
            s[8] += py.x * rx + py.y * ry;
        }
        block_sum9(s, red);
        if (threadIdx.x == 0) rot_sh[it & 1] = almeida_update(rotation, s, eps, alpha);
        __syncthreads();
        rotation = rot_sh[it & 1];
    }
    if (threadIdx.x == 0) *out = make_float4(rotation.w, -rotation.i, -rotation.j, -rotation.k);   // :199
}

constexpr int kXcdSlots = 32;                 // workgroups per XCD at most (256 / 8)
constexpr int kHierMinBlocks = 65;            // up to 64 workgroups (one granule per polling lane, a cache line each) the flat gather is faster: 0.083 vs 0.091 ms at 64
// granules per item: the flat exchange, the per-XCD partials and sums, the XCC_ID table (256 dwords = 64 granules), the
// status word (one granule of its own)
// Granules that cross XCDs (the workgroup partials of the flat gather, the XCD sums of the two-level one) sit `gs`
// granules apart: 1 (16 bytes: one cache line holds eight) for at most 8 publishers, kGranLine (128 bytes: a line each)
// above.  tools/ubench_allgather (profiles/ubench_allgather_r03.txt), publish -> all seen, cycles: 8 workgroups 1.9k
// packed / 2.45k a line each; 32 workgroups 3.3k packed -- 32 publishers and 32 x 64 polling lanes meet in four lines of
// one memory channel -- / 2.55k a line each.
constexpr int kGranLine = 8;
// One-XCD launches (block-vector sized fields): the launch is `spread` = 8 times as wide as the cluster, only every
// eighth workgroup works (round-robin dispatch puts those on ONE XCD; the others return at once), and from step 1 on
// the partial sums are exchanged through that XCD's L2 alone -- plain stores, `sc1` loads, no write-through to memory:
// 540 instead of 950 cycles per hand-off and a single level (tools/ubench_exchange).  Step 0 verifies the premise from the
// workgroups' XCC_IDs (hier_sh = 2); a launch that is not on one XCD keeps the flat write-through exchange.  Both forms
// add the same numbers in the same order (one granule per lane, then the wave tree): same bits.
constexpr int kOneXcdMax = 64;                // workgroups of such a cluster at most (one granule per polling lane)
#ifndef OFPS_ALMEIDA_REC_GROUP
#define OFPS_ALMEIDA_REC_GROUP 1
#endif
#ifndef OFPS_ALMEIDA_XG_STRIDE
#define OFPS_ALMEIDA_XG_STRIDE 8
#endif
constexpr int kXgStride = OFPS_ALMEIDA_XG_STRIDE;   // the 8 XCD sums, polled by every workgroup of a two-level launch
__host__ __device__ inline int cluster_gran_stride(int nblk) { return nblk <= 8 ? 1 : kGranLine; }
__host__ __device__ inline size_t cluster_gran_per_item(int nblk) {     // a multiple of 8 granules: every item starts on a cache line
    return 6 * (size_t)nblk * cluster_gran_stride(nblk) + 2 * 8 * kXgStride + 2 * 8 * kXcdSlots + 64 + 8 + 2 * kOneXcdMax * kGranLine;
}

#ifdef OFPS_HIP_TEST_HOOKS
#define OFPS_TEST_FAULT(x) (x)
#else
#define OFPS_TEST_FAULT(x) 0u          /* the fault injector does not exist in the product library */
#endif

template <bool FAST, int EPT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void almeida_lsq_cluster_kernel(const float4* __restrict__ entries, size_t n, Camera cam,
                                                                   const DeltaConsts dk, gran_u4* gran, uint32_t tag_base,
                                                                   float4* __restrict__ out_quat,
                                                                   unsigned long long* __restrict__ prof, uint32_t fault_arg, int hier_mode,
                                                                   unsigned long long* __restrict__ recoveries, int spread) {
    if (spread > 1 && (blockIdx.x % (unsigned)spread) != 0) return;     // one-XCD launch: the workgroups dealt to the other XCDs
    // fault (libofps_hip_testhooks.so only; compiled out of the product library): workgroup fault-1 withholds its step-3
    // granule, which is what a workgroup that never became resident looks like to the others -- exercises the timeout
    // and the in-kernel recovery (almeida_solo_solve)
    const uint32_t fault = OFPS_TEST_FAULT(fault_arg);
    // prof (diagnostics, normally null): the serial wave of every workgroup stamps s_memtime at the phase boundaries of each step
#define OFPS_STAMP(slot) do { if (prof && threadIdx.x == kSerialWave * 64) prof[(((size_t)blockIdx.y * (gridDim.x / spread) + blockIdx.x / spread) * kIters + it) * kProfSlots + (slot)] = __builtin_readcyclecounter(); } while (0)
#define OFPS_STAMP_W2(slot) do { if (prof && threadIdx.x == kSerialWave * 64) prof[(((size_t)blockIdx.y * (gridDim.x / spread) + blockIdx.x / spread) * kIters + it) * kProfSlots + (slot)] = __builtin_readcyclecounter(); } while (0)
    constexpr bool P_LDS = EPT >= 8;
    // dense regime, an even number of records per thread: the step loop works on PAIRS of records with packed-f32 fused
    // multiply-adds (14 v_pk_fma_f32 + 2 v_rcp_f32 per pair instead of 28 v_fma_f32 + 2 v_rcp_f32); the two halves keep
    // partial sums of their own (even / odd records), joined once per step
    constexpr bool PK = FAST && (EPT % 2 == 0);
    // the wave that carries a step's serial chain: finishes the block sum, publishes the granule, gathers, updates
    constexpr int kSerialWave = 2;                       // (waves 0 and 1 gather the A triples in step 0)
    static_assert(kSerialWave == 2 && BLOCK >= 256, "the step-0 gather of A uses waves 0 and 1, the two-level gather wave 3");
    __shared__ float red[BLOCK / 64][9];
    __shared__ float red4[3][64];                       // the step's row sums (block_sum3_rows)
    __shared__ float apart_sh[6];                       // this workgroup's partial of A = J^T J, published in step 0
    __shared__ float a_sh[6];                           // A folded over all workgroups (rotation-independent)
    __shared__ Quat rot_sh[2];
    __shared__ float4 aff_sh[2][3];                     // dense regime: the folded camera + rotation of rot_sh[] (DeltaAffine's nine coefficients), same slots;
                                                        // three 16-byte reads in flight per wave and step (as a plain struct: five dependent LDS round trips)
    __shared__ int fail_sh;
    __shared__ struct { const float4* entries; size_t n; float4* out; unsigned long long* recoveries; uint32_t* flag; uint32_t tag; Camera cam; } cold_sh;   // what the recovery path needs, parked
    __shared__ int hier_sh;                             // steps >= 1 gather in two levels (per XCD through its L2, then across)
    __shared__ Lu3 lu_sh;                               // factorisation of the folded A, made in step 0 by the updating wave
    __shared__ float4 plds[P_LDS ? EPT * BLOCK : 1];     // (roll.x, roll.y, pitch.x, pitch.y) per record
    const int nblk = (int)(gridDim.x / (unsigned)spread), blk = (int)(blockIdx.x / (unsigned)spread);
    const size_t item = blockIdx.y;
    gran_u4* g = gran + item * cluster_gran_per_item(nblk);            // [parity][triple: A0-2, A3-5, b][workgroup], then:
    const int gs = cluster_gran_stride(nblk);
    gran_u4* xl = g + 6 * (size_t)nblk * gs;                                // [parity][XCD][rank in XCD]: per-XCD partials (same-XCD readers)
    gran_u4* xg = xl + 2 * 8 * kXcdSlots;                              // [parity][XCD]: per-XCD sums (every reader)
    uint32_t* xccs = reinterpret_cast<uint32_t*>(xg + 2 * 8 * kXgStride);          // [workgroup]: tag << 16 | XCC_ID
    // the item's fail word (a granule of its own) and this launch's value for it: unique per launch until the tags wrap
    // (the buffer is re-zeroed then)
    const FailFlag ff = {reinterpret_cast<uint32_t*>(xg + 2 * 8 * kXgStride + 64), (tag_base + 1u) | 0x80000000u};
    gran_u4* x1 = xg + 2 * 8 * kXgStride + 64 + 8;                     // [parity][workgroup], a line each: the one-XCD exchange
    const int xcd = blk & 7, xrank = blk >> 3;                         // where round-robin dispatch puts this workgroup
    const int xmembers = (nblk - xcd + 7) / 8, nxcd = nblk < 8 ? nblk : 8;
    const float eps = almeida_eps();
    const Mat3 mroll = mat3_uniform(mat3_from_euler(0.0f, eps, 0.0f));     // lib.rs:30-34
    const Mat3 mpitch = mat3_uniform(mat3_from_euler(eps, 0.0f, 0.0f));    // lib.rs:36-38
    const Mat3 myaw = mat3_uniform(mat3_from_euler(0.0f, 0.0f, -eps));     // lib.rs:40-42
    float4 e[EPT];
    float2 pr[P_LDS ? 1 : EPT], pp[P_LDS ? 1 : EPT], py[EPT];
    // the hoisted unprojection costs two registers per record; with reciprocal-multiply quotients it is four VALU
    // instructions per component to recompute (same function, same bits), which is cheaper than spilling at EPT = 8
    constexpr bool RECOMPUTE_UNPROJ = FAST;             // the dense regime's step loop does not use the unprojection (delta_affine)
    float uwx[RECOMPUTE_UNPROJ ? 1 : EPT], uwz[RECOMPUTE_UNPROJ ? 1 : EPT];
    float uwy = 0.0f;
    bool ok[EPT];
    float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (threadIdx.x == 0) {                              // slot 1 = rotation entering step 0
        fail_sh = 0; hier_sh = 0; rot_sh[1] = Quat{1.0f, 0.0f, 0.0f, 0.0f};
        cold_sh.entries = entries + item * n; cold_sh.n = n; cold_sh.cam = cam; cold_sh.out = out_quat + item;
        cold_sh.recoveries = recoveries; cold_sh.flag = ff.p; cold_sh.tag = ff.tag;
        if (hier_mode) {
            const uint32_t me = (((tag_base + 1u) & 0xFFFFu) << 16) | ((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFFu);   // HW_REG_XCC_ID
            asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(xccs + blk), "v"(me) : "memory");
        }
        if constexpr (FAST) aff_store(aff_sh[1], delta_affine(dk, quat_to_mat3(rot_sh[1])));
    }
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const size_t i = ((size_t)blk * EPT + t) * BLOCK + threadIdx.x;
        ok[t] = i < n;
        e[t] = ok[t] ? entries[item * n + i] : make_float4(0.5f, 0.5f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
        const Unproj un = cam_unproject<FAST>(cam, e[t].x, e[t].y);
        if constexpr (!RECOMPUTE_UNPROJ) { uwx[t] = un.wx; uwz[t] = un.wz; }
        uwy = un.wy;
        // a slot past the end of the field holds a finite dummy record with ALL-ZERO prototypes: every product it
        // contributes below is an exact +0, so the 30-step loop needs no per-record validity branch
        const float2 zero2 = make_float2(0.0f, 0.0f);
        const float2 r = ok[t] ? cam_delta_w<FAST>(cam, e[t].x, e[t].y, un, mroll) : zero2;
        const float2 p = ok[t] ? cam_delta_w<FAST>(cam, e[t].x, e[t].y, un, mpitch) : zero2;
        py[t] = ok[t] ? cam_delta_w<FAST>(cam, e[t].x, e[t].y, un, myaw) : zero2;
        if constexpr (P_LDS && PK) {
            // pair u = t / 2 keeps (roll.x of both records, roll.y of both) in slot 2u and the pitch prototypes in slot 2u + 1:
            // one ds_read_b128 per slot hands the step loop its packed operands
            float* lf = reinterpret_cast<float*>(plds);
            const int h = t & 1;
            const size_t a0 = ((size_t)(t & ~1) * BLOCK + threadIdx.x) * 4, a1 = ((size_t)(t | 1) * BLOCK + threadIdx.x) * 4;
            lf[a0 + h] = r.x; lf[a0 + 2 + h] = r.y; lf[a1 + h] = p.x; lf[a1 + 2 + h] = p.y;
        } else if constexpr (P_LDS) plds[t * BLOCK + threadIdx.x] = make_float4(r.x, r.y, p.x, p.y);
        else { pr[t] = r; pp[t] = p; }
        s[0] += r.x * r.x + r.y * r.y;
        s[1] += r.x * p.x + r.y * p.y;
        s[2] += r.x * py[t].x + r.y * py[t].y;
        s[3] += p.x * p.x + p.y * p.y;
        s[4] += p.x * py[t].x + p.y * py[t].y;
        s[5] += py[t].x * py[t].x + py[t].y * py[t].y;
        if constexpr (EPT >= 8) __builtin_amdgcn_sched_barrier(0);  // one record at a time: see the step loop
    }
    if constexpr (FAST) {
#pragma unroll
        for (int t = 0; t < EPT; ++t) { e[t].z += e[t].x - 0.5f; e[t].w += e[t].y - 0.5f; }    // see the step loop
    }
    // PK: the per-record state of the step loop, two records side by side
    constexpr int NP = PK ? EPT / 2 : 1;
    v2f ex2[NP], ey2[NP], ez2[NP], ew2[NP], yx2[NP], yy2[NP];
    if constexpr (PK) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            ex2[u] = v2f{e[2 * u].x, e[2 * u + 1].x}; ey2[u] = v2f{e[2 * u].y, e[2 * u + 1].y};
            ez2[u] = v2f{e[2 * u].z, e[2 * u + 1].z}; ew2[u] = v2f{e[2 * u].w, e[2 * u + 1].w};
            yx2[u] = v2f{py[2 * u].x, py[2 * u + 1].x}; yy2[u] = v2f{py[2 * u].y, py[2 * u + 1].y};
        }
    }
    // the six A partials leave the registers before the step loop (they would stay live through all 30 steps otherwise)
    block_sum<0, 6>(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) apart_sh[k] = s[k];
    }
    __syncthreads();
    // the rotation lives in LDS between steps (rot_sh[(it + 1) & 1] enters step it): nothing but the per-record state
    // stays in vector registers across the step loop
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < kIters; ++it) {
        const float alpha = (it == kIters - 1) ? 1.0f : 0.5f;      // lib.rs:138
        OFPS_STAMP(0);
        Mat3 rotm;
        DeltaAffine aff;
        if constexpr (FAST) aff = aff_load_uniform(aff_sh[(it + 1) & 1]);                                   // lib.rs:140, folded by the updating wave
        else rotm = mat3_uniform(quat_to_mat3(rot_sh[(it + 1) & 1]));                                          // lib.rs:140
        s[6] = 0.0f; s[7] = 0.0f; s[8] = 0.0f;
        if constexpr (PK) {
            const v2f xa = pk_splat(aff.xa), xb = pk_splat(aff.xb), xc = pk_splat(aff.xc);
            const v2f ya = pk_splat(aff.ya), yb = pk_splat(aff.yb), yc = pk_splat(aff.yc);
            const v2f da = pk_splat(aff.da), db = pk_splat(aff.db), dc = pk_splat(aff.dc);
            v2f s6 = pk_splat(0.0f), s7 = pk_splat(0.0f), s8 = pk_splat(0.0f);
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const v2f X = pk_fma(xa, ex2[u], pk_fma(xb, ey2[u], xc));
                const v2f Y = pk_fma(ya, ex2[u], pk_fma(yb, ey2[u], yc));
                const v2f D = pk_fma(da, ex2[u], pk_fma(db, ey2[u], dc));
                const v2f ninv = v2f{-__builtin_amdgcn_rcpf(D.x), -__builtin_amdgcn_rcpf(D.y)};
                const v2f rx = pk_fma(X, ninv, ez2[u]), ry = pk_fma(Y, ninv, ew2[u]);      // motion - delta, as below
                v2f rrx, rry, ppx, ppy;
                if constexpr (P_LDS) {
                    const float4 v0 = plds[(2 * u) * BLOCK + threadIdx.x], v1 = plds[(2 * u + 1) * BLOCK + threadIdx.x];
                    rrx = v2f{v0.x, v0.y}; rry = v2f{v0.z, v0.w}; ppx = v2f{v1.x, v1.y}; ppy = v2f{v1.z, v1.w};
                } else {
                    rrx = v2f{pr[2 * u].x, pr[2 * u + 1].x}; rry = v2f{pr[2 * u].y, pr[2 * u + 1].y};
                    ppx = v2f{pp[2 * u].x, pp[2 * u + 1].x}; ppy = v2f{pp[2 * u].y, pp[2 * u + 1].y};
                }
                s6 = pk_fma(rry, ry, pk_fma(rrx, rx, s6));
                s7 = pk_fma(ppy, ry, pk_fma(ppx, rx, s7));
                s8 = pk_fma(yy2[u], ry, pk_fma(yx2[u], rx, s8));
                if constexpr (EPT >= 8) __builtin_amdgcn_sched_barrier(0);  // one pair at a time (registers: see below)
            }
            s[6] = s6.x + s6.y; s[7] = s7.x + s7.y; s[8] = s8.x + s8.y;
        } else {
#pragma unroll
        for (int t = 0; t < EPT; ++t) {
            float rx, ry;                                          // motion - delta
            if constexpr (FAST) {
                // delta = (X / D + 0.5) - p: the record keeps motion + p - 0.5 (folded once, below the prologue), so the
                // residual is one fused operation per component after the quotient
                const float X = __builtin_fmaf(aff.xa, e[t].x, __builtin_fmaf(aff.xb, e[t].y, aff.xc));
                const float Y = __builtin_fmaf(aff.ya, e[t].x, __builtin_fmaf(aff.yb, e[t].y, aff.yc));
                const float D = __builtin_fmaf(aff.da, e[t].x, __builtin_fmaf(aff.db, e[t].y, aff.dc));
                const float ninv = -__builtin_amdgcn_rcpf(D);
                rx = __builtin_fmaf(X, ninv, e[t].z); ry = __builtin_fmaf(Y, ninv, e[t].w);
            } else {
                const Unproj un = Unproj{uwx[t], uwy, uwz[t]};
                const float2 d = cam_delta_w<false>(cam, e[t].x, e[t].y, un, rotm);
                rx = e[t].z - d.x; ry = e[t].w - d.y;
            }
            float2 r, p;
            if constexpr (P_LDS) { const float4 v = plds[t * BLOCK + threadIdx.x]; r = make_float2(v.x, v.y); p = make_float2(v.z, v.w); }
            else { r = pr[t]; p = pp[t]; }
            if constexpr (FAST) {
                s[6] = __builtin_fmaf(r.y, ry, __builtin_fmaf(r.x, rx, s[6]));
                s[7] = __builtin_fmaf(p.y, ry, __builtin_fmaf(p.x, rx, s[7]));
                s[8] = __builtin_fmaf(py[t].y, ry, __builtin_fmaf(py[t].x, rx, s[8]));
            } else {
                s[6] += r.x * rx + r.y * ry;
                s[7] += p.x * rx + p.y * ry;
                s[8] += py[t].x * rx + py[t].y * ry;
            }
            // 8 records x ~15 temporaries interleaved do not fit beside the 64 resident registers: keep the scheduler
            // from overlapping more than two records (4 waves per SIMD hide the latency instead)
            if constexpr (EPT >= 8) { if ((t & (OFPS_ALMEIDA_REC_GROUP - 1)) == OFPS_ALMEIDA_REC_GROUP - 1) __builtin_amdgcn_sched_barrier(0); }
        }
        }
        OFPS_STAMP(1);
        block_sum3_rows<kSerialWave>(s[6], s[7], s[8], red4);
        OFPS_STAMP(2);
        const uint32_t tag = (tag_base + (uint32_t)it + 1u) & 0xFFFFu;
        gran_u4* gp = g + (size_t)(it & 1) * 3 * nblk * gs;
        // (uniform: hier_sh was written before the barrier that ended step 0; launches without the two-level gather do not
        // even read it -- an LDS round trip less on the serial wave's chain)
        const int hsel = (hier_mode != 0 && it > 0) ? hier_sh : 0;       // 0 flat, 1 two levels, 2 everybody on one XCD
        const bool hier = hsel == 1, onex = hsel == 2;
        if (threadIdx.x == kSerialWave * 64) {
            if (it == 0) {
                gran_store3(gp + (size_t)blk * gs, tag, apart_sh[0], apart_sh[1], apart_sh[2]);
                gran_store3(gp + (size_t)(nblk + blk) * gs, tag, apart_sh[3], apart_sh[4], apart_sh[5]);
            }
            if (!(fault && it == 3 && (uint32_t)blk == fault - 1u)) {
                if (hier) gran_store3_local(xl + ((size_t)(it & 1) * 8 + xcd) * kXcdSlots + xrank, tag, s[6], s[7], s[8]);
                else if (onex) gran_store3_local(x1 + ((size_t)(it & 1) * kOneXcdMax + blk) * kGranLine, tag, s[6], s[7], s[8]);
                else gran_store3(gp + (2 * (size_t)nblk + blk) * gs, tag, s[6], s[7], s[8]);
            }
        }
        // wave k gathers triple k (the A triples in step 0 only); wave 2 -- the serial wave: it finished the block sum and
        // published the granule above -- gathers the right-hand side and goes straight on to the LU + quaternion update: no
        // barrier and no LDS round trip anywhere between a step's block sum and its new rotation.
        // Two-level form (steps >= 1 of launches with many workgroups, round-robin dispatch verified in step 0): wave 3 of
        // each XCD's first workgroup sums that XCD's partials -- plain stores found in the shared L2 by `sc1` loads -- and
        // publishes the XCD's sum write-through; wave 2 of EVERY workgroup then gathers the <= 8 XCD sums instead of up to
        // 256 partials.  All workgroups add the same numbers in the same order, so the rotations stay bit-identical.
        float ta = 0.0f, tb = 0.0f, tc = 0.0f;
        bool got = true;
        if (hier) {
            if (xrank == 0 && wave == 3) {
                float la = 0.0f, lb = 0.0f, lc = 0.0f;
                const bool lgot = gran_sweep_sum3_local(xl + ((size_t)(it & 1) * 8 + xcd) * kXcdSlots, xmembers, tag, ff, la, lb, lc);
                if (lane == 0) {
                    if (lgot) gran_store3(xg + ((size_t)(it & 1) * 8 + xcd) * kXgStride, tag, la, lb, lc);
                    else fail_sh = 1;
                }
            }
            if (wave == kSerialWave) got = gran_sweep_sum3(xg + (size_t)(it & 1) * 8 * kXgStride, kXgStride, nxcd, tag, ff, ta, tb, tc);
        } else if (onex) {
            if (wave == kSerialWave) got = gran_sweep_sum3_local(x1 + (size_t)(it & 1) * kOneXcdMax * kGranLine, nblk, tag, ff, ta, tb, tc, kGranLine);
        } else if (wave < 3 && (wave == kSerialWave || it == 0)) {
            got = gran_sweep_sum3(gp + (size_t)wave * nblk * gs, gs, nblk, tag, ff, ta, tb, tc);
        }
        if (it == 0) {
            if (wave < 2 && lane == 0) {
                if (got) { a_sh[3 * wave] = ta; a_sh[3 * wave + 1] = tb; a_sh[3 * wave + 2] = tc; }
                else fail_sh = 1;
            }
            if (wave == 3 && hier_mode) {               // does workgroup b sit on XCD b % 8?  every workgroup reaches the same verdict
                bool rr = false, same = false;
                const bool xgot = xcc_sweep_check(xccs, nblk, tag, ff, rr, same);
                if (lane == 0) {
                    if (!xgot) fail_sh = 1;
                    else if (hier_mode == 3) hier_sh = (same && nblk <= kOneXcdMax) ? 2 : 0;
                    else hier_sh = rr ? 1 : 0;
                }
            }
            __syncthreads();
        }
        OFPS_STAMP(3);
        if (wave == kSerialWave) {
            OFPS_STAMP_W2(5);
            if (got) {
                Lu3 lu;
                if (it == 0) {
                    const float am[9] = {a_sh[0], a_sh[1], a_sh[2], a_sh[1], a_sh[3], a_sh[4], a_sh[2], a_sh[4], a_sh[5]};
                    lu = lu3_factor(am);
                    if (lane == 0) lu_sh = lu;
                } else {
                    lu = lu_sh;
                }
                const Quat q = almeida_update_wave_lu<FAST>(rot_sh[(it + 1) & 1], lu, ta, tb, tc, eps, alpha);
                if constexpr (FAST) {                   // fold camera and new rotation once, here, for every wave's next step
                    const DeltaAffine A = delta_affine(dk, quat_to_mat3(q));
                    if (lane == 0) aff_store(aff_sh[it & 1], A);
                }
                if (lane == 0) rot_sh[it & 1] = q;
                OFPS_STAMP_W2(6);
            } else if (lane == 0) {
                fail_sh = 1;
            }
        }
        __syncthreads();
        if (fail_sh) break;                              // a bounded spin expired somewhere in this workgroup
        OFPS_STAMP(4);
    }
#undef OFPS_STAMP
#undef OFPS_STAMP_W2
    if (!fail_sh) {
        const Quat rotation = rot_sh[(kIters - 1) & 1];
        if (blk == 0 && threadIdx.x == 0) out_quat[item] = make_float4(rotation.w, -rotation.i, -rotation.j, -rotation.k);  // :199
        return;
    }
    // ---- a spin expired: see almeida_solo_solve.  Everything this path needs was parked in LDS by the prologue, so
    // the step loop above carries no state for it.
    __shared__ int solo_sh;
    if (threadIdx.x == 0) {
        const uint32_t old = __hip_atomic_exchange(cold_sh.flag, cold_sh.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        solo_sh = old != cold_sh.tag;
        if (solo_sh && cold_sh.recoveries) atomicAdd(cold_sh.recoveries, 1ull);
    }
    __syncthreads();
    if (!solo_sh) return;
    almeida_solo_solve<FAST, BLOCK>(cold_sh.entries, cold_sh.n, cold_sh.cam, red, rot_sh, cold_sh.out);
}

// ---- RANSAC sampler: must stay bit-identical to orc_sample_index (oracle/ofps_oracle.c)
__host__ __device__ inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct SampleKey { uint32_t key[4]; uint32_t half, mask; };
__device__ __forceinline__ SampleKey sample_key(uint64_t seed, uint32_t iter, uint32_t stream, uint32_t n) {
    SampleKey k;
    uint32_t bits = 2;
    while (bits < 32 && (1ull << bits) < (uint64_t)n) bits += 2;
    k.half = bits / 2; k.mask = (1u << k.half) - 1u;
    const uint64_t kk = mix64(seed ^ mix64(((uint64_t)iter << 1) | (uint64_t)(stream & 1u)));
#pragma unroll
    for (int r = 0; r < 4; ++r) k.key[r] = (uint32_t)mix64(kk + (uint64_t)r);
    return k;
}
__device__ __forceinline__ uint32_t sample_index(const SampleKey& k, uint32_t i, uint32_t n) {
    if (n <= 1) return 0;
    uint32_t x = i;
    do {
        uint32_t l = x >> k.half, r = x & k.mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            uint32_t f = r * 0x9E3779B1u + k.key[round];
            f ^= f >> 15; f *= 0x85EBCA77u; f ^= f >> 13;
            const uint32_t nl = r, nr = l ^ (f & k.mask);
            l = nl; r = nr;
        }
        x = (l << k.half) | r;
    } while (x >= n);
    return x;
}

// one thread per (item, hypothesis): 3-sample solve, sequential sums in sample order (lib.rs:215-217);
// writes the homogeneous matrix of fit.inverse() (lib.rs:224)
// One hypothesis per QUAD of lanes (16 per wave): lane q < 3 of a quad owns sample q -- its record, prototypes and, per
// step, its `delta` -- and the three per-sample terms of every sum are added in sample order through quad_perm
// broadcasts (DPP operands, no LDS), so every lane of the quad holds the sums the one-lane walk produced, bit for bit.
// The update runs in all four lanes with the three half-angle sincos spread over lanes 0..2 (almeida_update_wave_lu's
// trick at quad width).  A hypothesis is a chain of 30 dependent steps on a handful of lanes either way -- latency,
// not throughput -- and the quad form shortens the chain: three `delta` and three sincos side by side instead of in a row.
template <int J>
__device__ __forceinline__ float quad_bcast(float x) {            // value of lane J of the caller's quad
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), J * 0x55, 0xF, 0xF, false));
}
__device__ __forceinline__ float quad_sum3(float init, float t, uint32_t n3) {   // ((init + t[0]) + t[1]) + t[2], first n3 terms
    float s = init;
    const float t0 = quad_bcast<0>(t), t1 = quad_bcast<1>(t), t2 = quad_bcast<2>(t);
    if (n3 > 0) s += t0;
    if (n3 > 1) s += t1;
    if (n3 > 2) s += t2;
    return s;
}
// One Gauss-Newton update of a quad's hypothesis (lib.rs:181-195) on a factorisation made once per hypothesis (A = J^T J of the three
// samples does not change over the 30 steps), with almeida_update_wave_lu's shortened chain: reciprocal-multiply back substitution, one product for scale and half angle, the
// polynomial sincos of a step-sized angle, fused quaternion products (~250 instead of ~400 dependent instructions per step).
__device__ __forceinline__ Quat almeida_update_quad_lu(const Quat& rotation, const Lu3& f, float b0, float b1, float b2, float eps, float alpha, int q) {
    const float b[3] = {b0, b1, b2};
    float model[3];
    if (!lu3_apply(f, b, model)) { model[0] = model[1] = model[2] = 0.0f; }       // :181-183
    const float xsel = q == 0 ? model[0] : (q == 1 ? model[1] : -model[2]);
    const float half = xsel * (eps * alpha * 0.5f);                               // :185, :189-191 (see almeida_update_wave_lu)
    float sn, cs;
    sincos_small(half, sn, cs);
    const float c0 = quad_bcast<0>(cs), s0 = quad_bcast<0>(sn), c1 = quad_bcast<1>(cs), s1 = quad_bcast<1>(sn), c2 = quad_bcast<2>(cs), s2 = quad_bcast<2>(sn);
    const Quat pr = {c1 * c0, s1 * c0, c1 * s0, s1 * s0};                         // :193, (pitch * roll) * yaw without the exact-zero terms
    const Quat rot = {__builtin_fmaf(pr.w, c2, -(pr.k * s2)), __builtin_fmaf(pr.i, c2, pr.j * s2), __builtin_fmaf(pr.j, c2, -(pr.i * s2)),
                      __builtin_fmaf(pr.w, s2, pr.k * c2)};
    const Quat& a = rotation;
    Quat r;                                                                       // :195
    r.w = __builtin_fmaf(-a.k, rot.k, __builtin_fmaf(-a.j, rot.j, __builtin_fmaf(-a.i, rot.i, a.w * rot.w)));
    r.i = __builtin_fmaf(-a.k, rot.j, __builtin_fmaf(a.j, rot.k, __builtin_fmaf(a.i, rot.w, a.w * rot.i)));
    r.j = __builtin_fmaf(a.k, rot.i, __builtin_fmaf(a.j, rot.w, __builtin_fmaf(-a.i, rot.k, a.w * rot.j)));
    r.k = __builtin_fmaf(a.k, rot.w, __builtin_fmaf(-a.j, rot.i, __builtin_fmaf(a.i, rot.j, a.w * rot.k)));
    return r;
}

constexpr int kHypPerWave = 16;

__global__ __launch_bounds__(64) void ransac_hyp_kernel(const float4* __restrict__ entries, uint32_t n, uint32_t iters,
                                                        uint64_t seed, Camera cam, Mat3* __restrict__ hyp) {
    const size_t item = blockIdx.y;
    const int q = threadIdx.x & 3;
    const uint32_t it_raw = blockIdx.x * kHypPerWave + (threadIdx.x >> 2);
    const bool live = it_raw < iters;
    const uint32_t it = live ? it_raw : iters - 1;                // tail quads repeat the last hypothesis (all lanes stay active for the DPP reads)
    const float eps = almeida_eps();
    const uint32_t n3 = n < 3 ? n : 3;
    const SampleKey sk = sample_key(seed + item, it, 0, n);        // every item of a batch draws its own samples
    const Mat3 mroll = mat3_from_euler(0.0f, eps, 0.0f), mpitch = mat3_from_euler(eps, 0.0f, 0.0f),
               myaw = mat3_from_euler(0.0f, 0.0f, -eps);
    // this lane's sample (lane 3 of a quad, and lanes past n3, carry a harmless copy of sample 0: their terms are never added)
    const uint32_t j = (uint32_t)q < n3 ? (uint32_t)q : 0u;
    const float4 e = entries[item * n + sample_index(sk, j, n)];
    const float2 pr = cam_delta(cam, e.x, e.y, mroll);
    const float2 pp = cam_delta(cam, e.x, e.y, mpitch);
    const float2 py = cam_delta(cam, e.x, e.y, myaw);
    // A = J^T J depends on the prototypes only: summed once, in sample order (the reference re-adds the same
    // numbers in every step, lib.rs:159-173)
    float a[6];
    a[0] = quad_sum3(0.0f, pr.x * pr.x + pr.y * pr.y, n3);
    a[1] = quad_sum3(0.0f, pr.x * pp.x + pr.y * pp.y, n3);
    a[2] = quad_sum3(0.0f, pr.x * py.x + pr.y * py.y, n3);
    a[3] = quad_sum3(0.0f, pp.x * pp.x + pp.y * pp.y, n3);
    a[4] = quad_sum3(0.0f, pp.x * py.x + pp.y * py.y, n3);
    a[5] = quad_sum3(0.0f, py.x * py.x + py.y * py.y, n3);
    const float am[9] = {a[0], a[1], a[2], a[1], a[3], a[4], a[2], a[4], a[5]};
    const Lu3 lu = lu3_factor(am);
    Quat rotation = {1.0f, 0.0f, 0.0f, 0.0f};
    for (int s_it = 0; s_it < kIters; ++s_it) {
        const float alpha = (s_it == kIters - 1) ? 1.0f : 0.5f;
        const Mat3 rotm = quat_to_mat3(rotation);
        const float2 d = cam_delta(cam, e.x, e.y, rotm);
        const float rx = e.z - d.x, ry = e.w - d.y;
        const float b0 = quad_sum3(0.0f, pr.x * rx + pr.y * ry, n3);
        const float b1 = quad_sum3(0.0f, pp.x * rx + pp.y * ry, n3);
        const float b2 = quad_sum3(0.0f, py.x * rx + py.y * ry, n3);
        rotation = almeida_update_quad_lu(rotation, lu, b0, b1, b2, eps, alpha, q);
    }
    // fit = rotation.inverse(); mat = fit.inverse().to_homogeneous() = to_homogeneous(rotation)
    if (live && q == 0) hyp[item * iters + it] = quat_to_mat3(rotation);
}

__device__ __forceinline__ bool ransac_is_inlier(const Camera& cam, float fx, float fy, const Mat3& mat, const float4& e,
                                                 float thr2) {
    const float2 d = cam_delta(cam, e.x, e.y, mat);                           // lib.rs:229-231
    const float2 ang = cam_point_angle(fx, fy, e.x + d.x, e.y + d.y);    // :234
    const float vx = (e.z - d.x) * cosf(ang.x), vy = (e.w - d.y) * cosf(ang.y);
    return vx * vx + vy * vy <= thr2;                                         // :236
}

// one workgroup per (hypothesis, item): inlier count over the drawn samples.  kCountThreads threads: with the reference's 1,000 samples one
// inlier test per thread (a chain of ~300 dependent instructions: sampler, gather, delta, two atan, two cos) instead of four in a row
constexpr int kCountThreads = 1024;
__global__ __launch_bounds__(kCountThreads) void ransac_count_kernel(const float4* __restrict__ entries, uint32_t n, uint32_t ns,
                                                           uint32_t iters, uint64_t seed, Camera cam, float fx, float fy,
                                                           float thr2, const Mat3* __restrict__ hyp,
                                                           uint32_t* __restrict__ counts) {
    __shared__ uint32_t total;
    const size_t item = blockIdx.y;
    const uint32_t it = blockIdx.x;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const Mat3 mat = hyp[item * iters + it];
    const SampleKey sk = sample_key(seed + item, it, 1, n);
    uint32_t c = 0;
    for (uint32_t j = threadIdx.x; j < ns; j += kCountThreads) {
        const float4 e = entries[item * n + sample_index(sk, j, n)];
        c += ransac_is_inlier(cam, fx, fy, mat, e, thr2) ? 1u : 0u;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) counts[item * iters + it] = total;
}

// one workgroup per item: first hypothesis with the largest count (strict > in lib.rs:243), then
// its inliers compacted in sample order into sel[item][0..count)
__global__ __launch_bounds__(1024) void ransac_select_kernel(const float4* __restrict__ entries, uint32_t n, uint32_t ns,
                                                             uint32_t iters, uint64_t seed, Camera cam, float fx, float fy,
                                                             float thr2, const Mat3* __restrict__ hyp,
                                                             const uint32_t* __restrict__ counts,
                                                             float4* __restrict__ sel, uint32_t* __restrict__ sel_idx,
                                                             uint32_t* __restrict__ sel_n) {
    __shared__ unsigned long long best;
    __shared__ uint32_t base;
    const size_t item = blockIdx.x;
    if (threadIdx.x == 0) { best = 0; base = 0; }
    __syncthreads();
    unsigned long long k = 0;
    for (uint32_t it = threadIdx.x; it < iters; it += 1024) {
        const unsigned long long key = ((unsigned long long)counts[item * iters + it] << 32) | (0xFFFFFFFFu - it);
        k = key > k ? key : k;
    }
    atomicMax(&best, k);
    __syncthreads();
    const uint32_t bcount = (uint32_t)(best >> 32);
    const uint32_t bit = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFu);
    if (bcount == 0) {                          // best_inliers stays empty (len > 0 never true)
        if (threadIdx.x == 0) sel_n[item] = 0;
        return;
    }
    const Mat3 mat = hyp[item * iters + bit];
    const SampleKey sk = sample_key(seed + item, bit, 1, n);
    // ordered compaction, 1024 samples per round: rank inside the wave from a ballot, the 16 wave totals through LDS (one
    // barrier per round: the slots alternate), the running offset kept in every thread
    __shared__ uint32_t wave_cnt[2][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t run = 0;
    int par = 0;
    for (uint32_t j0 = 0; j0 < ns; j0 += 1024, par ^= 1) {
        const uint32_t j = j0 + threadIdx.x;
        uint32_t idx = 0;
        bool in = false;
        float4 e = make_float4(0, 0, 0, 0);
        if (j < ns) {
            idx = sample_index(sk, j, n);
            e = entries[item * n + idx];
            in = ransac_is_inlier(cam, fx, fy, mat, e, thr2);
        }
        const unsigned long long bal = __ballot(in);
        if (lane == 0) wave_cnt[par][wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t v = wave_cnt[par][k]; total += v; before += k < wave ? v : 0u; }
        if (in) {
            const uint32_t pos = run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            sel[item * ns + pos] = e;
            if (sel_idx) sel_idx[item * ns + pos] = idx;
        }
        run += total;
    }
    if (threadIdx.x == 0) base = run;
    __syncthreads();
    if (threadIdx.x == 0) sel_n[item] = base;
}

// Persistent (spin-waiting) launches of different contexts must not interleave on one device: two clusters that each
// hold part of the CUs would wait for each other's missing workgroups.  Within a process they are chained on the GPU
// through one event per device (no host blocking); across processes the in-kernel timeout is the safety net.
// A device with ONE live context needs none of this -- its launches follow each other on its stream, and a stream change drains the old
// stream first (switch_stream) -- and pays nothing: no event is recorded behind its launches (the record is a barrier packet between the
// estimator's kernel and whatever follows it on the stream: ~4 us of every estimate).  A second context on the device switches the gate on:
// its ofps_hip_init drains the device under the gate's lock, so a launch made while the count read one is over before the count reads two.
struct ClusterGate {
    std::mutex m;
    hipEvent_t ev[64] = {};
    hipStream_t last[64] = {};      // the stream of the device's latest cluster launch: stream order already chains launches on it
    int live[64] = {};              // contexts alive on the device
};
static ClusterGate g_cluster_gate;

void cluster_gate_context_created(int device) {
    std::lock_guard<std::mutex> lk(g_cluster_gate.m);
    if (++g_cluster_gate.live[device & 63] == 2) {
        (void)hipDeviceSynchronize();                       // (the caller has made `device` current) whatever the lone context launched ungated is over
        g_cluster_gate.last[device & 63] = nullptr;
    }
}
void cluster_gate_context_destroyed(int device) {
    std::lock_guard<std::mutex> lk(g_cluster_gate.m);
    if (g_cluster_gate.live[device & 63] > 0) --g_cluster_gate.live[device & 63];
}

template <bool FAST, int EPT, int BLOCK = 1024>
static void launch_cluster(ofps_hip_ctx* ctx, hipStream_t s, int nblk, int items, const float4* d_entries, size_t n, const Camera& cam,
                           gran_u4* gran, uint32_t tag_base, float4* d_quat, unsigned long long* prof, unsigned long long* recoveries) {
    const uint32_t fault = (uint32_t)ctx->opt.test_almeida_fault;   // always 0 in the product library (ofps_hip_set_option refuses it)
    int hier_mode = ctx->opt.almeida_hier;                    // 0 never, 1 when it pays (>= kHierMinBlocks workgroups), 2 always (A/B, tests)
    if (hier_mode == 1 && nblk < kHierMinBlocks) hier_mode = 0;     // few workgroups: the flat gather is as fast and needs no XCC_ID round
    // small clusters of 256-thread workgroups: a launch eight times as wide whose every eighth workgroup works -- all on one
    // XCD under round-robin dispatch, exchanging through its L2 (kOneXcdMax; verified in step 0, flat otherwise)
    int spread = 1;
    // (a CU-masked stream may leave that XCD a handful of CUs -- too few for the working workgroups to be co-resident: measured
    // 57 ms of timeouts + solo solves per call on a 32-CU mask dealt four CUs to each XCD -- so masked streams keep the flat form)
    if (BLOCK == 256 && ctx->opt.almeida_one_xcd && ctx->opt.almeida_hier != 2 && nblk * items <= kOneXcdMax && nblk >= 2 &&
        ctx->stream_cus >= ctx->num_cus) { spread = 8; hier_mode = 3; }
    hipLaunchKernelGGL((almeida_lsq_cluster_kernel<FAST, EPT, BLOCK>), dim3(nblk * spread, items), dim3(BLOCK), 0, s, d_entries, n, cam,
                       delta_consts(cam), gran,
                       tag_base, d_quat, prof, fault, hier_mode, recoveries, spread);
}

// -> 1 launched, 0 not applicable (caller falls back to the launch-per-step kernel), < 0 error
static int lsq_cluster(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int batch, const Camera& cam, float4* d_quat) {
    // records per thread: fewer = less arithmetic per step on the critical path, more = fewer workgroups to gather from
    // and -- for batches -- more items whose workgroups are co-resident in one launch.  Cost model fitted to
    // tools/almeida_prof.py (ms per launch ~ 0.11 + 0.004 * ept + 0.00001 * workgroups per item, + 0.006 when the gather has
    // two levels (more than 64 workgroups): 129,600 vectors 64 x 2 flat 0.090 ms, 127 x 1 two-level 0.096 -- with the two-level gather
    // a workgroup more costs next to nothing, a record more per thread is arithmetic on the critical path); the count that
    // minimises launches x cost wins (lone problems: the smallest that fits; 64 x 129,600 vectors: 8 -> 4 launches).
    // The exact-arithmetic variant (fields of <= 65,536 vectors) stops at 4 records per thread: with 8, the hoisted
    // unprojection beside 8 records and their prototypes does not fit 128 VGPRs (13 spilled, 56 B of scratch per lane);
    // the dense variant does not keep the unprojection and fits.
    const bool dense_by_size = ctx->opt.almeida_fast >= 0 ? ctx->opt.almeida_fast != 0 : n > 65536;
    const int ept_max = dense_by_size ? 8 : 4;
    int ept = ept_max;
    {
        double best = 1e30;
        for (int e : {1, 2, 4, 8}) {
            if (e > ept_max) continue;
            const size_t nb = (n + (size_t)e * 1024 - 1) / ((size_t)e * 1024);
            if (nb < 1 || nb > 256 || nb > (size_t)ctx->num_cus) continue;
            const int per = ctx->num_cus / (int)nb;
            const double cost = (double)((batch + per - 1) / per) * (0.11 + 0.004 * e + 0.00001 * (double)nb + (nb >= (size_t)kHierMinBlocks ? 0.006 : 0.0));
            if (cost < best) { best = cost; ept = e; }
        }
    }
    // Block-vector sized fields (exact arithmetic, <= 65,536 vectors) whose items all fit one launch: 256-thread workgroups,
    // at most 64 of them per item (one granule per polling lane).  A step's serial chain runs on one wave per workgroup;
    // with one wave per SIMD the records + block sum in front of it take 1.3k cycles instead of 2.5k, and since the
    // granules sit a cache line apart (cluster_gran_stride) 16-64 publishers cost the exchange 0.6k more than 8, not
    // 1.4k (round 2 had measured this variant at 0.104 vs 0.105 ms and rejected it: its granules were packed).  Records per
    // thread: the fewest that keep an item within 64 workgroups (8,040 vectors: 32 x 1: 0.076 ms, 16 x 2: 0.080, 8 x 4: 0.083).
    int block = 1024;
    const int e256 = n <= 16384 ? 1 : (n <= 32768 ? 2 : 4);
    if (!dense_by_size) {
        const size_t nb256 = (n + (size_t)e256 * 256 - 1) / ((size_t)e256 * 256);
        if (nb256 <= 64 && nb256 * (size_t)batch <= (size_t)ctx->num_cus) block = 256;
    }
    if (ctx->opt.almeida_block) block = ctx->opt.almeida_block;      // A/B (OFPS_HIP_ALMEIDA_BLOCK)
    if (block == 256) ept = e256;
    if (ctx->opt.almeida_ept) ept = ctx->opt.almeida_ept < ept_max ? ctx->opt.almeida_ept : ept_max;   // A/B (OFPS_HIP_ALMEIDA_EPT)
    const size_t per_wg = (size_t)ept * block;
    const size_t nblk_sz = (n + per_wg - 1) / per_wg;
    if (nblk_sz < 1 || nblk_sz > 256 || nblk_sz > (size_t)ctx->num_cus) return 0;
    const int nblk = (int)nblk_sz;
    const int per_launch = ctx->num_cus / nblk;              // items whose workgroups are all co-resident (1 per CU)
    const bool dense = dense_by_size;                        // per-pixel regime: reciprocal-multiply quotients, see fdiv (A/B: OFPS_HIP_ALMEIDA_FAST)
    const size_t gran_bytes = (size_t)per_launch * cluster_gran_per_item(nblk) * sizeof(gran_u4);
    auto* gran = static_cast<gran_u4*>(scratch(ctx, S_GRAN, gran_bytes));
    if (!gran) return OFPS_HIP_ENOMEM;
    hipStream_t s = ctx->stream;
    unsigned long long* prof = nullptr;
    if (ctx->opt.almeida_prof) {
        prof = static_cast<unsigned long long*>(scratch(ctx, S_ALM_PROF, (size_t)per_launch * nblk * kIters * kProfSlots * sizeof(unsigned long long)));
        if (!prof) return OFPS_HIP_ENOMEM;
    }
    // how many launches of this context had to be finished by almeida_solo_solve (ofps_hip_almeida_recoveries)
    const bool fresh_counter = ctx->scratch[S_ALM_RECOVER].p == nullptr;
    auto* recoveries = static_cast<unsigned long long*>(scratch(ctx, S_ALM_RECOVER, sizeof(unsigned long long)));
    if (!recoveries) return OFPS_HIP_ENOMEM;
    if (fresh_counter) OFPS_HIP_TRY(ctx, hipMemsetAsync(recoveries, 0, sizeof(unsigned long long), s));
    const size_t cap = ctx->scratch[S_GRAN].cap;
    for (int b0 = 0; b0 < batch; b0 += per_launch) {
        const int items = batch - b0 < per_launch ? batch - b0 : per_launch;
        // zeroed once per ALLOCATION (the generation counts them: a regrown buffer may come back at the old address with
        // a tail nobody zeroed) and before the 16-bit tags wrap
        if (ctx->gran_zeroed_gen != ctx->scratch[S_GRAN].gen || ctx->gran_tag_base > 0xFF00u) {
            OFPS_HIP_TRY(ctx, hipMemsetAsync(gran, 0, cap, s));
            ctx->gran_zeroed_gen = ctx->scratch[S_GRAN].gen; ctx->gran_tag_base = 0;
        }
        const uint32_t tag_base = ctx->gran_tag_base;
        ctx->gran_tag_base += 32;
        const float4* ent = d_entries + (size_t)b0 * n;
        float4* q = d_quat + b0;
        std::lock_guard<std::mutex> lk(g_cluster_gate.m);
        const bool gated = g_cluster_gate.live[ctx->device & 63] > 1;            // several contexts on this device: chain their launches
        hipEvent_t& ev = g_cluster_gate.ev[ctx->device & 63];
        hipStream_t& last = g_cluster_gate.last[ctx->device & 63];
        if (gated) {
            if (!ev) OFPS_HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            else if (last != s) OFPS_HIP_TRY(ctx, hipStreamWaitEvent(s, ev, 0));  // (a barrier packet costs ~4 us in front of the kernel)
            last = s;
        }
        if (block == 256 && ept >= 4) launch_cluster<false, 4, 256>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        else if (block == 256 && ept == 2) launch_cluster<false, 2, 256>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        else if (block == 256) launch_cluster<false, 1, 256>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        else if (dense && ept == 8) launch_cluster<true, 8>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        else if (dense && ept == 4) launch_cluster<true, 4>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        else if (dense && ept == 2) launch_cluster<true, 2>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        else if (dense) launch_cluster<true, 1>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        else if (ept == 4) launch_cluster<false, 4>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        else if (ept == 2) launch_cluster<false, 2>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        else launch_cluster<false, 1>(ctx, s, nblk, items, ent, n, cam, gran, tag_base, q, prof, recoveries);
        OFPS_HIP_TRY(ctx, hipGetLastError());
        if (gated) OFPS_HIP_TRY(ctx, hipEventRecord(ev, s));
    }
    if (prof) {                                              // diagnostics: phase table of the last launch to stderr
        const size_t cnt = (size_t)nblk * kIters * kProfSlots;
        std::vector<unsigned long long> h(cnt);
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(h.data(), prof, cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        OFPS_HIP_TRY(ctx, hipStreamSynchronize(s));
        // 0 step start, 1 records done, 2 block sum done, 3/5 granule published and every workgroup's gathered, 6 rotation
        // updated, 4 step end (after the barrier)
        double ph[6] = {0, 0, 0, 0, 0, 0};
        for (int b = 0; b < nblk; ++b)
            for (int it = 0; it < kIters; ++it) {
                const unsigned long long* r = h.data() + ((size_t)b * kIters + it) * kProfSlots;
                ph[0] += (double)(r[1] - r[0]); ph[1] += (double)(r[2] - r[1]); ph[2] += (double)(r[3] - r[2]);
                ph[3] += (double)(r[5] - r[3]); ph[4] += (double)(r[6] - r[5]); ph[5] += (double)(r[4] - r[6]);
            }
        if (nblk <= 64) {                                      // per workgroup: step start -> granule published, published -> all seen
            fprintf(stderr, "[almeida cluster prof] per workgroup (to publish / gather):");
            for (int b = 0; b < nblk; ++b) {
                double tp = 0, tg = 0;
                for (int it = 1; it < kIters; ++it) {
                    const unsigned long long* r = h.data() + ((size_t)b * kIters + it) * kProfSlots;
                    tp += (double)(r[3] - r[0]); tg += (double)(r[5] - r[3]);
                }
                fprintf(stderr, " %d:%.0f/%.0f", b, tp / (kIters - 1), tg / (kIters - 1));
            }
            fprintf(stderr, "\n");
        }
        const double den = (double)nblk * kIters;
        fprintf(stderr, "[almeida cluster prof] n=%zu nblk=%d ept=%d block=%d  cycles/step on the serial wave: records %.0f  block sum %.0f  "
                        "publish + gather %.0f  update %.0f  barrier %.0f   wg0 total %.0f\n",
                n, nblk, ept, block, ph[0] / den, ph[1] / den, (ph[2] + ph[3]) / den, ph[4] / den, ph[5] / den,
                (double)(h[(size_t)(kIters - 1) * kProfSlots + 4] - h[0]));
    }
    return 1;
}

static int lsq_stepped(ofps_hip_ctx* ctx, const float4* d_entries, size_t n_max, int batch, const Camera& cam, float4* d_quat) {
    hipStream_t s = ctx->stream;
    // enough workgroups to fill the chip, few enough that the fixed-order fold stays short
    const size_t per_wg = n_max > 65536 ? 8 * 1024 : 1024;
    int nblk = (int)((n_max + per_wg - 1) / per_wg);
    const int cap = (2 * ctx->num_cus + batch - 1) / batch;
    if (nblk > cap) nblk = cap;
    if (nblk < 1) nblk = 1;                                  // empty fields (forced A/B path): one workgroup that sums nothing
    auto* part = static_cast<float*>(scratch(ctx, S_ALM_PART, 2 * (size_t)batch * nblk * 9 * sizeof(float)));
    auto* state = static_cast<Quat*>(scratch(ctx, S_ALM_STATE, 2 * (size_t)batch * sizeof(Quat)));
    if (!part || !state) return OFPS_HIP_ENOMEM;
    float* pa = part;
    float* pb = part + (size_t)batch * nblk * 9;
    const bool dense = n_max > 65536;          // per-pixel regime: reciprocal-multiply quotients, see fdiv
    for (int it = 0; it <= kIters; ++it) {
        if (dense)
            hipLaunchKernelGGL(almeida_lsq_step_kernel<true>, dim3(nblk, batch), dim3(1024), 0, s, d_entries, n_max, it, cam,
                               pa, pb, state, batch, d_quat);
        else
            hipLaunchKernelGGL(almeida_lsq_step_kernel<false>, dim3(nblk, batch), dim3(1024), 0, s, d_entries, n_max, it, cam,
                               pa, pb, state, batch, d_quat);
        float* t = pa; pa = pb; pb = t;
    }
    OFPS_HIP_TRY(ctx, hipGetLastError());
    return OFPS_HIP_OK;
}

static int lsq_device(ofps_hip_ctx* ctx, const float4* d_entries, size_t stride, size_t n_max, const uint32_t* d_n,
                      uint32_t min_n, int batch, const Camera& cam, float4* d_quat) {
    hipStream_t s = ctx->stream;
    // One-workgroup solver: one launch, 30 steps at 2.5-6 us each on a single CU (0.076 ms at N = 576, 0.185 ms at
    // N = 8,040, flat up to 256 items) -- the path for batches of block-vector sized problems and mandatory when the
    // entry count lives on the device (RANSAC refit).  Larger problems, and lone problems above `cluster_min`, go to
    // the cluster solver (one launch, granule exchange between co-resident workgroups); what it cannot hold
    // (N > 256 x 8192) or a forced A/B run takes one launch per step.
    bool wg_path = n_max <= 8192;
    bool cluster = d_n == nullptr && stride == n_max;
    size_t cluster_min = batch == 1 ? 1536 : 8192;                       // lone problems above this size use the cluster (one XCD: 0.061 ms at any size up to 8,040; the
                                                                         // one-workgroup solver: 1,280 vectors 0.055, 1,600 0.062, 2,048 0.066 -- tools/almeida_threshold_ab.py)
    if (ctx->opt.almeida_path == 1 && d_n == nullptr) { wg_path = false; cluster = false; }   // A/B experiments only (OFPS_HIP_ALMEIDA_PATH)
    if (ctx->opt.almeida_path == 2) cluster = false;
    if (ctx->opt.almeida_path == 3) cluster_min = 0;
    if (cluster && n_max > cluster_min && n_max > 0) {
        const int rc = lsq_cluster(ctx, d_entries, n_max, batch, cam, d_quat);
        if (rc != 0) return rc < 0 ? rc : OFPS_HIP_OK;
    }
    if (wg_path) {
        const dim3 g(batch), b(1024);
        if (n_max <= 1024) hipLaunchKernelGGL((almeida_lsq_wg_kernel<1>), g, b, 0, s, d_entries, stride, n_max, d_n, min_n, cam, d_quat);
        else if (n_max <= 2048) hipLaunchKernelGGL((almeida_lsq_wg_kernel<2>), g, b, 0, s, d_entries, stride, n_max, d_n, min_n, cam, d_quat);
        else if (n_max <= 4096) hipLaunchKernelGGL((almeida_lsq_wg_kernel<4>), g, b, 0, s, d_entries, stride, n_max, d_n, min_n, cam, d_quat);
        else hipLaunchKernelGGL((almeida_lsq_wg_kernel<8>), g, b, 0, s, d_entries, stride, n_max, d_n, min_n, cam, d_quat);
        OFPS_HIP_TRY(ctx, hipGetLastError());
        return OFPS_HIP_OK;
    }
    OFPS_REQUIRE(ctx, d_n == nullptr && stride == n_max, "almeida: device-side counts need n <= 8192");
    return lsq_stepped(ctx, d_entries, n_max, batch, cam, d_quat);
}

int almeida_device(ofps_hip_ctx* ctx, const float4* d_entries, size_t n, int batch, float aspect, float fov_y_deg,
                          int use_ransac, size_t num_iters, float inlier_deg, size_t num_samples, uint64_t seed,
                          float4* d_quat) {
    OFPS_REQUIRE(ctx, batch >= 1 && batch <= 65535, "almeida: batch %d out of range", batch);
    OFPS_REQUIRE(ctx, n < (1ull << 31), "almeida: too many entries");
    OFPS_REQUIRE(ctx, aspect > 0.0f && fov_y_deg > 0.0f && fov_y_deg < 180.0f, "almeida: bad camera (aspect=%g fov_y=%g)",
                 (double)aspect, (double)fov_y_deg);
    const Camera cam = camera_new(aspect, fov_y_deg);
    if (!use_ransac) return lsq_device(ctx, d_entries, n, n, nullptr, 0, batch, cam, d_quat);

    OFPS_REQUIRE(ctx, num_iters >= 1 && num_iters <= 65535, "almeida: ransac iters %zu out of range", num_iters);
    OFPS_REQUIRE(ctx, num_samples >= 1, "almeida: ransac samples must be >= 1");
    const uint32_t iters = (uint32_t)num_iters;
    const uint32_t ns = (uint32_t)(num_samples < n ? num_samples : n);
    hipStream_t s = ctx->stream;
    auto* hyp = static_cast<Mat3*>(scratch(ctx, S_ALM_HYP, (size_t)batch * iters * sizeof(Mat3)));
    auto* counts = static_cast<uint32_t*>(scratch(ctx, S_ALM_COUNTS, (size_t)batch * iters * sizeof(uint32_t)));
    auto* sel = static_cast<float4*>(scratch(ctx, S_ALM_SEL, (size_t)batch * (ns ? ns : 1) * sizeof(float4)));
    auto* sel_n = static_cast<uint32_t*>(scratch(ctx, S_ALM_SELN, (size_t)batch * sizeof(uint32_t)));
    if (!hyp || !counts || !sel || !sel_n) return OFPS_HIP_ENOMEM;
    const float target = to_radians_host(inlier_deg);                              // lib.rs:210
    const float thr2 = target * target;
    const float fy = 0.5f / tanf(to_radians_host(cam.fov_y) / 2.0f);                // camera.rs:121-122
    const float fx = fy / cam.aspect;
    if (n == 0 || ns == 0) {
        OFPS_HIP_TRY(ctx, hipMemsetAsync(sel_n, 0, (size_t)batch * sizeof(uint32_t), s));
    } else {
        hipLaunchKernelGGL(ransac_hyp_kernel, dim3((iters + kHypPerWave - 1) / kHypPerWave, batch), dim3(64), 0, s, d_entries, (uint32_t)n, iters,
                           seed, cam, hyp);
        hipLaunchKernelGGL(ransac_count_kernel, dim3(iters, batch), dim3(kCountThreads), 0, s, d_entries, (uint32_t)n, ns, iters, seed,
                           cam, fx, fy, thr2, hyp, counts);
        hipLaunchKernelGGL(ransac_select_kernel, dim3(batch), dim3(1024), 0, s, d_entries, (uint32_t)n, ns, iters, seed, cam,
                           fx, fy, thr2, hyp, counts, sel, (uint32_t*)nullptr, sel_n);
    }
    OFPS_HIP_TRY(ctx, hipGetLastError());
    const size_t cap = ns ? ns : 1;
    if (cap <= 8192) return lsq_device(ctx, sel, cap, cap, sel_n, 3, batch, cam, d_quat);
    // "Ransac samples" can reach 16000 (lib.rs:92-95): refit sets above 8192 need their size on the
    // host to pick the multi-launch path -- one small read-back per item.
    for (int b = 0; b < batch; ++b) {
        uint32_t cnt = 0;
        OFPS_HIP_TRY(ctx, hipMemcpyAsync(&cnt, sel_n + b, sizeof(cnt), hipMemcpyDeviceToHost, s));
        OFPS_HIP_TRY(ctx, hipStreamSynchronize(s));
        int rc;
        if (cnt <= 8192) rc = lsq_device(ctx, sel + (size_t)b * cap, cap, cnt < 1 ? 1 : cnt, sel_n + b, 3, 1, cam, d_quat + b);
        else rc = lsq_device(ctx, sel + (size_t)b * cap, cnt, cnt, nullptr, 0, 1, cam, d_quat + b);
        if (rc != OFPS_HIP_OK) return rc;
    }
    return OFPS_HIP_OK;
}

}  // namespace ofps

extern "C" {

int ofps_hip_almeida_dev(ofps_hip_ctx* ctx, const void* d_entries, size_t n_per_item, int batch, float aspect,
                         float fov_y_deg, int use_ransac, size_t num_iters, float inlier_deg, size_t num_samples,
                         uint64_t seed, void* d_out_quat) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, d_out_quat && (d_entries || n_per_item == 0), "almeida: null device pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ofps::almeida_device(ctx, static_cast<const float4*>(d_entries), n_per_item, batch, aspect, fov_y_deg, use_ransac,
                                num_iters, inlier_deg, num_samples, seed, static_cast<float4*>(d_out_quat));
}

int ofps_hip_almeida(ofps_hip_ctx* ctx, const float* entries, size_t n, float aspect, float fov_y_deg, int use_ransac,
                     size_t num_iters, float inlier_deg, size_t num_samples, uint64_t seed, float out_quat[4],
                     float out_tr[3]) {
    if (!ctx) return OFPS_HIP_EINVAL;
    OFPS_REQUIRE(ctx, out_quat && (entries || n == 0), "almeida: null host pointer");
    OFPS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto* d_ent = static_cast<float4*>(ofps::scratch(ctx, ofps::S_ENTRIES, n * sizeof(float4)));
    auto* d_q = static_cast<float4*>(ofps::scratch(ctx, ofps::S_QUAT, sizeof(float4)));
    if (!d_ent || !d_q) return OFPS_HIP_ENOMEM;
    if (n) OFPS_HIP_TRY(ctx, hipMemcpyAsync(d_ent, entries, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    int rc = ofps::almeida_device(ctx, d_ent, n, 1, aspect, fov_y_deg, use_ransac, num_iters, inlier_deg, num_samples, seed,
                                  d_q);
    if (rc != OFPS_HIP_OK) return rc;
    OFPS_HIP_TRY(ctx, hipMemcpyAsync(out_quat, d_q, 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    OFPS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (out_tr) { out_tr[0] = out_tr[1] = out_tr[2] = 0.0f; }                       // lib.rs:120
    return OFPS_HIP_OK;
}

}  // extern "C"
