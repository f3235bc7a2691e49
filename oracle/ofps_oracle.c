/*
 * ofps_oracle.c -- CPU restatement of the OFPS flow hot path.  TEST INFRASTRUCTURE ONLY:
 * see ofps_oracle.h for who may call this and for the parity status of every function.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  Contraction must
 * stay off: the Rust reference never fuses a*b+c, and neither may this file.
 *
 * All arithmetic is f32 and follows the operation order of the Rust source / nalgebra 0.30
 * (nalgebra is not vendored under /root/reference; its semantics are restated from
 * SURVEY.md Appendix A and validated through the reference's own Almeida test).
 */
#include "ofps_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__SSE2__)
#include <emmintrin.h>
#include <immintrin.h>
#endif

/* f32::to_radians: self * (PI / 180) with the ratio folded in f32 (Rust core). */
static float to_radians(float deg) {
    const float k = 3.14159265358979323846264338327950288f / 180.0f;
    return deg * k;
}

/* ------------------------------------------------------------------------------------ */
/* nalgebra helpers                                                                     */
/* ------------------------------------------------------------------------------------ */

/* Matrix4::transform_point: v = M[0:3,0:3]*p + t, n = M[3,0:3].p + M[3,3];
 * v/n if n != 0.  The 3x3 product is nalgebra's gemv in axpy form: column 0 first, then
 * res += col_j * p_j. */
void orc_mat4_transform_point(const float m[16], const float p[3], float out[3]) {
    float n = (m[12] * p[0] + m[13] * p[1]) + m[14] * p[2];
    n = n + m[15];
    float v[3];
    for (int i = 0; i < 3; ++i) {
        float acc = m[4 * i + 0] * p[0];
        acc = m[4 * i + 1] * p[1] + acc;
        acc = m[4 * i + 2] * p[2] + acc;
        v[i] = acc + m[4 * i + 3];
    }
    if (n != 0.0f) {
        out[0] = v[0] / n; out[1] = v[1] / n; out[2] = v[2] / n;
    } else {
        out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
    }
}

/* 4x4 product, axpy order over k (column of a times b[k][j]). */
void orc_mat4_mul(const float a[16], const float b[16], float out[16]) {
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float acc = a[4 * i + 0] * b[0 * 4 + j];
            for (int k = 1; k < 4; ++k) acc = a[4 * i + k] * b[4 * k + j] + acc;
            r[4 * i + j] = acc;
        }
    memcpy(out, r, sizeof(r));
}

/* Rotation3::from_euler_angles(roll,pitch,yaw).to_homogeneous(). */
void orc_mat4_from_euler(float roll, float pitch, float yaw, float out[16]) {
    float sr = sinf(roll), cr = cosf(roll);
    float sp = sinf(pitch), cp = cosf(pitch);
    float sy = sinf(yaw), cy = cosf(yaw);
    float m[16] = {
        cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, 0.0f,
        sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, 0.0f,
        -sp,     cp * sr,                cp * cr,                0.0f,
        0.0f,    0.0f,                   0.0f,                   1.0f};
    memcpy(out, m, sizeof(m));
}

static void vec3_normalize(float v[3]) {
    float n = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] /= n; v[1] /= n; v[2] /= n;
}
static void vec3_cross(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* Matrix4::look_at_rh (tests only). */
void orc_mat4_look_at_rh(const float eye[3], const float target[3], const float up[3], float out[16]) {
    float f[3] = {target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]};
    vec3_normalize(f);
    float s[3]; vec3_cross(f, up, s); vec3_normalize(s);
    float u[3]; vec3_cross(s, f, u);
    float r[9] = {s[0], s[1], s[2], u[0], u[1], u[2], -f[0], -f[1], -f[2]};
    float m[16] = {0};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) m[4 * i + j] = r[3 * i + j];
        m[4 * i + 3] = -((r[3 * i] * eye[0] + r[3 * i + 1] * eye[1]) + r[3 * i + 2] * eye[2]);
    }
    m[15] = 1.0f;
    memcpy(out, m, sizeof(m));
}

/* UnitQuaternion::from_euler_angles, half-angle form.  q = (w,i,j,k). */
void orc_quat_from_euler(float roll, float pitch, float yaw, float q[4]) {
    float sr = sinf(roll * 0.5f), cr = cosf(roll * 0.5f);
    float sp = sinf(pitch * 0.5f), cp = cosf(pitch * 0.5f);
    float sy = sinf(yaw * 0.5f), cy = cosf(yaw * 0.5f);
    q[0] = cr * cp * cy + sr * sp * sy;
    q[1] = sr * cp * cy - cr * sp * sy;
    q[2] = cr * sp * cy + sr * cp * sy;
    q[3] = cr * cp * sy - sr * sp * cy;
}

/* Hamilton product a*b, nalgebra's expression order. */
void orc_quat_mul(const float a[4], const float b[4], float out[4]) {
    float aw = a[0], ai = a[1], aj = a[2], ak = a[3];
    float bw = b[0], bi = b[1], bj = b[2], bk = b[3];
    float w = aw * bw - ai * bi - aj * bj - ak * bk;
    float i = aw * bi + ai * bw + aj * bk - ak * bj;
    float j = aw * bj - ai * bk + aj * bw + ak * bi;
    float k = aw * bk + ai * bj - aj * bi + ak * bw;
    out[0] = w; out[1] = i; out[2] = j; out[3] = k;
}

void orc_quat_to_homogeneous(const float q[4], float out[16]) {
    float w = q[0], i = q[1], j = q[2], k = q[3];
    float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
    float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    float m[16] = {
        ww + ii - jj - kk, ij - wk,           wj + ik,           0.0f,
        wk + ij,           ww - ii + jj - kk, jk - wi,           0.0f,
        ik - wj,           wi + jk,           ww - ii - jj + kk, 0.0f,
        0.0f,              0.0f,              0.0f,              1.0f};
    memcpy(out, m, sizeof(m));
}

void orc_quat_inverse(const float q[4], float out[4]) {
    out[0] = q[0]; out[1] = -q[1]; out[2] = -q[2]; out[3] = -q[3];
}

/* UnitQuaternion * Vector3 (tests only): v + 2*cross(q.v, cross(q.v, v) + w*v). */
void orc_quat_transform_vector(const float q[4], const float v[3], float out[3]) {
    float u[3] = {q[1], q[2], q[3]};
    float t[3]; vec3_cross(u, v, t);
    t[0] *= 2.0f; t[1] *= 2.0f; t[2] *= 2.0f;
    float c[3]; vec3_cross(u, t, c);
    out[0] = (t[0] * q[0] + c[0]) + v[0];
    out[1] = (t[1] * q[0] + c[1]) + v[1];
    out[2] = (t[2] * q[0] + c[2]) + v[2];
}

/* angle_to(a,b) = angle of b * a^-1 (tests only). */
float orc_quat_angle_to(const float a[4], const float b[4]) {
    float ai[4], d[4];
    orc_quat_inverse(a, ai);
    orc_quat_mul(b, ai, d);
    float w = fabsf(d[0]);
    float n = sqrtf((d[1] * d[1] + d[2] * d[2]) + d[3] * d[3]);
    return atan2f(n, w) * 2.0f;
}

/* Matrix3::lu().solve(b): partial pivoting, multipliers scaled by the
 * reciprocal of the pivot, axpy updates; None when a U diagonal is exactly zero. */
int orc_lu3_solve(const float a_in[9], const float b_in[3], float x[3]) {
    float m[9]; memcpy(m, a_in, sizeof(m));
    int perm_a[3], perm_b[3], nperm = 0;
    for (int i = 0; i < 3; ++i) {
        int piv = i; float best = fabsf(m[3 * i + i]);
        for (int r = i + 1; r < 3; ++r) {
            float v = fabsf(m[3 * r + i]);
            if (v > best) { best = v; piv = r; }
        }
        float diag = m[3 * piv + i];
        if (diag == 0.0f) continue;
        if (piv != i) {
            perm_a[nperm] = i; perm_b[nperm] = piv; ++nperm;
            for (int c = 0; c < 3; ++c) { float t = m[3 * i + c]; m[3 * i + c] = m[3 * piv + c]; m[3 * piv + c] = t; }
        }
        float inv_diag = 1.0f / diag;
        for (int r = i + 1; r < 3; ++r) m[3 * r + i] *= inv_diag;
        for (int c = i + 1; c < 3; ++c) {
            float neg = -m[3 * i + c];
            for (int r = i + 1; r < 3; ++r) m[3 * r + c] = neg * m[3 * r + i] + m[3 * r + c];
        }
    }
    float b[3] = {b_in[0], b_in[1], b_in[2]};
    for (int p = 0; p < nperm; ++p) { float t = b[perm_a[p]]; b[perm_a[p]] = b[perm_b[p]]; b[perm_b[p]] = t; }
    /* unit lower-triangular forward substitution */
    for (int i = 0; i < 2; ++i) {
        float coeff = b[i] / 1.0f;
        for (int r = i + 1; r < 3; ++r) b[r] = (-coeff) * m[3 * r + i] + b[r];
    }
    /* upper-triangular back substitution */
    for (int i = 2; i >= 0; --i) {
        float diag = m[3 * i + i];
        if (diag == 0.0f) return 0;
        float coeff = b[i] / diag;
        b[i] = coeff;
        for (int r = 0; r < i; ++r) b[r] = (-coeff) * m[3 * r + i] + b[r];
    }
    x[0] = b[0]; x[1] = b[1]; x[2] = b[2];
    return 1;
}

/* ------------------------------------------------------------------------------------ */
/* StandardCamera                                                                       */
/* ------------------------------------------------------------------------------------ */

/* camera.rs:26-35: Perspective3::new(aspect, fov_y.to_radians(), 0.1, 10.0) + inverse()
 * (set_fovy then set_aspect: m00 = m11 / aspect). */
void orc_camera_new(orc_camera* c, float aspect, float fov_y_deg) {
    const float zn = 0.1f, zf = 10.0f;
    float fovy = to_radians(fov_y_deg);
    c->aspect = aspect;
    c->fov_y = fov_y_deg;
    c->m11 = 1.0f / tanf(fovy / 2.0f);
    c->m00 = c->m11 / aspect;
    c->m22 = (zf + zn) / (zn - zf);
    c->m23 = zf * zn * 2.0f / (zn - zf);
    c->r00 = 1.0f / c->m00;
    c->r11 = 1.0f / c->m11;
    c->r32 = 1.0f / c->m23;
    c->r33 = c->m22 * c->r32;
}

static void camera_inv_proj(const orc_camera* c, float m[16]) {
    memset(m, 0, 16 * sizeof(float));
    m[0] = c->r00; m[5] = c->r11; m[11] = -1.0f; m[14] = c->r32; m[15] = c->r33;
}

/* camera.rs:45-55 */
void orc_camera_unproject(const orc_camera* c, const float p[2], const float inv_view[16], float out[3]) {
    float cx = p[0] * 2.0f - 1.0f, cy = p[1] * 2.0f - 1.0f;
    float ip[16], m[16];
    camera_inv_proj(c, ip);
    orc_mat4_mul(inv_view, ip, m);
    float pt[3] = {cx, cy, 1.0f};
    orc_mat4_transform_point(m, pt, out);
}

/* camera.rs:72-81: project_point then divide x,y by the NDC z (sic, :77). */
void orc_camera_project(const orc_camera* c, const float world[3], const float view[16], float out[2]) {
    float p[3];
    orc_mat4_transform_point(view, world, p);
    float inverse_denom = -1.0f / p[2];
    float sx = c->m00 * p[0] * inverse_denom;
    float sy = c->m11 * p[1] * inverse_denom;
    float sz = (c->m22 * p[2] + c->m23) * inverse_denom;
    float x = sx / sz, y = sy / sz;
    out[0] = (x + 1.0f) * 0.5f;
    out[1] = (y + 1.0f) * 0.5f;
}

static const float ORC_VIEW[16] = {   /* camera.rs:91-96, Z up / Y forward; view == view^T */
    -1.0f, 0.0f, 0.0f, 0.0f,
     0.0f, 0.0f, 1.0f, 0.0f,
     0.0f, 1.0f, 0.0f, 0.0f,
     0.0f, 0.0f, 0.0f, 1.0f};

/* camera.rs:89-112 */
void orc_camera_rotate(const orc_camera* c, const float p[2], const float rot[16], float out[2]) {
    float world[3], rw[3];
    orc_camera_unproject(c, p, ORC_VIEW /* transpose of a symmetric matrix */, world);
    orc_mat4_transform_point(rot, world, rw);
    orc_camera_project(c, rw, ORC_VIEW, out);
}

/* camera.rs:115-117 */
void orc_camera_delta(const orc_camera* c, const float p[2], const float rot[16], float out[2]) {
    float r[2];
    orc_camera_rotate(c, p, rot, r);
    out[0] = r[0] - p[0];
    out[1] = r[1] - p[1];
}

/* camera.rs:120-129, 150-161 */
void orc_camera_point_angle(const orc_camera* c, const float p[2], float out[2]) {
    float fy = 0.5f / tanf(to_radians(c->fov_y) / 2.0f);
    float fx = fy / c->aspect;
    float px = p[0] - 0.5f, py = p[1] - 0.5f;
    out[0] = atanf(px / fx);
    out[1] = atanf(py / fy);
}

/* ------------------------------------------------------------------------------------ */
/* MotionFieldDensifier                                                                 */
/* ------------------------------------------------------------------------------------ */

#define ORC_F32_EPSILON 1.1920929e-07f

/* `as usize` on an f32: saturating, NaN -> 0. */
static size_t f32_as_usize(float v) {
    if (!(v > 0.0f)) return 0;            /* negatives, -0, NaN */
    if (v >= 18446744073709551616.0f) return (size_t)-1;
    return (size_t)v;
}

/* motion_field.rs:164-178: nalgebra::clamp on a Point2 (all-components ordering),
 * then round-half-away-from-zero of pos*(dim-1). */
static void densifier_cell(float px, float py, int w, int h, size_t* ox, size_t* oy) {
    float cx, cy;
    if (px > 0.0f && py > 0.0f) {
        if (px < 1.0f && py < 1.0f) { cx = px; cy = py; }
        else { cx = 1.0f; cy = 1.0f; }
    } else { cx = 0.0f; cy = 0.0f; }
    *ox = f32_as_usize(roundf(cx * (float)(w - 1)));
    *oy = f32_as_usize(roundf(cy * (float)(h - 1)));
}

typedef struct { float* sum; float* cnt; int w, h; } densifier;

static void densifier_init(densifier* d, int w, int h) {            /* motion_field.rs:133-138 */
    size_t cells = (size_t)w * (size_t)h;
    d->w = w; d->h = h;
    d->sum = (float*)calloc(2 * cells + 2, sizeof(float));
    d->cnt = (float*)malloc((2 * cells + 2) * sizeof(float));
    for (size_t i = 0; i < 2 * cells; ++i) d->cnt[i] = ORC_F32_EPSILON;
}
static void densifier_free(densifier* d) { free(d->sum); free(d->cnt); }

static void densifier_add_idx(densifier* d, size_t idx, float mx, float my, float weight) { /* :141-147 */
    d->cnt[2 * idx + 0] += weight;
    d->cnt[2 * idx + 1] += weight;
    d->sum[2 * idx + 0] = mx * weight + d->sum[2 * idx + 0];
    d->sum[2 * idx + 1] = my * weight + d->sum[2 * idx + 1];
}

static void densifier_add_all(densifier* d, const float* entries, size_t n, uint32_t* out_cells) {
    for (size_t i = 0; i < n; ++i) {
        const float* e = entries + 4 * i;
        size_t x, y;
        densifier_cell(e[0], e[1], d->w, d->h, &x, &y);
        densifier_add_idx(d, y * (size_t)d->w + x, e[2], e[3], 1.0f);       /* :150-153, :188-190 */
        if (out_cells) { out_cells[2 * i] = (uint32_t)x; out_cells[2 * i + 1] = (uint32_t)y; }
    }
}

void orc_densify(const float* entries, size_t n, int w, int h,
                 float* out_field, uint32_t* out_cells, float* out_counts) {
    densifier d; densifier_init(&d, w, h);
    densifier_add_all(&d, entries, n, out_cells);
    size_t cells = (size_t)w * (size_t)h;
    if (out_counts) memcpy(out_counts, d.cnt, 2 * cells * sizeof(float));
    for (size_t i = 0; i < 2 * cells; ++i) out_field[i] = d.sum[i] / d.cnt[i];   /* :297-308 */
    densifier_free(&d);
}

/* add_vector_weighted for every entry (:164-178) */
void orc_densify_weighted(const float* entries, const float* weights, size_t n, int w, int h, float* out_field,
                          uint32_t* out_cells) {
    densifier d; densifier_init(&d, w, h);
    for (size_t i = 0; i < n; ++i) {
        const float* e = entries + 4 * i;
        size_t x, y;
        densifier_cell(e[0], e[1], d.w, d.h, &x, &y);
        densifier_add_idx(&d, y * (size_t)d.w + x, e[2], e[3], weights[i]);
        if (out_cells) { out_cells[2 * i] = (uint32_t)x; out_cells[2 * i + 1] = (uint32_t)y; }
    }
    size_t cells = (size_t)w * (size_t)h;
    for (size_t i = 0; i < 2 * cells; ++i) out_field[i] = d.sum[i] / d.cnt[i];
    densifier_free(&d);
}

size_t orc_densify_to_entries(const float* entries, size_t n, int w, int h, float* out_entries) {
    size_t cells = (size_t)w * (size_t)h;
    float* field = (float*)malloc((2 * cells + 2) * sizeof(float));
    uint32_t* xy = (uint32_t*)malloc((2 * n + 2) * sizeof(uint32_t));
    unsigned char* visited = (unsigned char*)calloc(cells + 1, 1);
    orc_densify(entries, n, w, h, field, xy, NULL);
    for (size_t i = 0; i < n; ++i) visited[(size_t)xy[2 * i + 1] * w + xy[2 * i]] = 1;
    float nx = 1.0f / (float)w, ny = 1.0f / (float)h;                /* cv-decoder/src/lib.rs:281 */
    size_t k = 0;
    for (int x = 0; x < w; ++x)                                       /* BTreeSet<(x,y)> order */
        for (int y = 0; y < h; ++y) {
            size_t idx = (size_t)y * w + x;
            if (!visited[idx]) continue;
            out_entries[4 * k + 0] = ((float)x + 0.5f) * nx;          /* :286-288 */
            out_entries[4 * k + 1] = ((float)y + 0.5f) * ny;
            out_entries[4 * k + 2] = field[2 * idx];
            out_entries[4 * k + 3] = field[2 * idx + 1];
            ++k;
        }
    free(field); free(xy); free(visited);
    return k;
}

/* motion_field.rs:193-294.  The BTreeSet<InterpCell{neighbors, idx}> is restated as two
 * arrays (membership + key) with a linear scan for the minimum: same total order. */
static const int ORC_INTERP_NB[6][2] = {{-1, 0}, {0, -1}, {-1, -1}, {1, 0}, {0, 1}, {1, 1}};

static long interp_calc_counts(const densifier* d, size_t i) {        /* :209-228 */
    long cnt = 0;
    long x = (long)(i % (size_t)d->w), y = (long)(i / (size_t)d->w);
    for (int k = 0; k < 6; ++k) {
        long nx = x + ORC_INTERP_NB[k][0], ny = y + ORC_INTERP_NB[k][1];
        if (nx >= 0 && nx < d->w && ny >= 0 && ny < d->h &&
            d->cnt[2 * ((size_t)nx + (size_t)ny * d->w)] > 0.1f) ++cnt;
    }
    return cnt;
}

static int densifier_interpolate(densifier* d) {
    size_t cells = (size_t)d->w * (size_t)d->h;
    unsigned char* inq = (unsigned char*)calloc(cells + 1, 1);
    long* key = (long*)calloc(cells + 1, sizeof(long));
    size_t qlen = 0;
    for (size_t i = 0; i < cells; ++i)
        if (d->cnt[2 * i] < 0.5f) { inq[i] = 1; key[i] = -interp_calc_counts(d, i); ++qlen; }   /* :230-241 */
    int ok = 1;
    if (qlen == cells) goto done;                                     /* :243-246 */
    for (;;) {
        /* queue.iter().next(): smallest (neighbors, idx) */
        size_t best = cells; long bk = 0;
        for (size_t i = 0; i < cells; ++i)
            if (inq[i] && (best == cells || key[i] < bk)) { best = i; bk = key[i]; }
        if (best == cells) break;
        size_t i = best;
        inq[i] = 0;
        long x = (long)(i % (size_t)d->w), y = (long)(i / (size_t)d->w);
        int added = 0;
        for (int k = 0; k < 6; ++k) {                                 /* :255-268 */
            long ox = ORC_INTERP_NB[k][0], oy = ORC_INTERP_NB[k][1];
            long nx = x + ox, ny = y + oy;
            if (nx >= 0 && nx < d->w && ny >= 0 && ny < d->h) {
                size_t idx = (size_t)nx + (size_t)ny * d->w;
                float cnt = d->cnt[2 * idx];
                if (cnt > 0.1f) {
                    float scale = 1.0f - sqrtf((float)(ox * ox + oy * oy)) * 0.5f;
                    float inv_cnt = 1.0f / cnt;
                    float s = scale * inv_cnt;
                    densifier_add_idx(d, i, s * d->sum[2 * idx], s * d->sum[2 * idx + 1], scale);
                    added = 1;
                }
            }
        }
        if (!added) {
            inq[i] = 1;                                               /* :270-271 (would spin) */
            ok = 0; break;
        }
        for (int k = 0; k < 6; ++k) {                                 /* :273-289 */
            long nx = x + ORC_INTERP_NB[k][0], ny = y + ORC_INTERP_NB[k][1];
            if (nx >= 0 && nx < d->w && ny >= 0 && ny < d->h) {
                size_t idx = (size_t)nx + (size_t)ny * d->w;
                long cnt = -interp_calc_counts(d, idx) + 1;
                if (inq[idx] && key[idx] == cnt) key[idx] = cnt - 1;
                else if (cnt != 0 && d->cnt[2 * idx] < 0.1f) { ok = 0; goto done; }  /* unreachable!() */
            }
        }
    }
done:
    free(inq); free(key);
    return ok;
}

void orc_densify_interpolated(const float* entries, size_t n, int w, int h, float* out_field) {
    densifier d; densifier_init(&d, w, h);
    densifier_add_all(&d, entries, n, NULL);
    densifier_interpolate(&d);
    size_t cells = (size_t)w * (size_t)h;
    for (size_t i = 0; i < 2 * cells; ++i) out_field[i] = d.sum[i] / d.cnt[i];
    densifier_free(&d);
}

/* ------------------------------------------------------------------------------------ */
/* BlockMotionDetection                                                                 */
/* ------------------------------------------------------------------------------------ */

int orc_block_dim(float min_size, size_t subdivide) {                 /* lib.rs:53-54 */
    float block_width = sqrtf(min_size) / (float)subdivide;
    return (int)f32_as_usize(ceilf(1.0f / block_width));
}

int orc_detect_motion(const float* entries, size_t n, float min_size, size_t subdivide,
                      float target_motion, size_t* out_area, int* out_dim, float* out_field) {
    int dim = orc_block_dim(min_size, subdivide);
    size_t cells = (size_t)dim * (size_t)dim;
    float* mf = (float*)malloc((2 * cells + 2) * sizeof(float));
    orc_densify(entries, n, dim, dim, mf, NULL, NULL);                /* lib.rs:57-61 */

    unsigned char* map = (unsigned char*)calloc(cells + 1, 1);
    for (size_t i = 0; i < cells; ++i) {                              /* lib.rs:63-68 */
        float mx = mf[2 * i], my = mf[2 * i + 1];
        float mag = sqrtf(mx * mx + my * my);
        if (mag >= target_motion) map[i] = 1;
    }

    size_t biggest_area = 0;
    float* biggest_mf = NULL;
    float* mf2 = (float*)malloc((2 * cells + 2) * sizeof(float));
    size_t* stack = (size_t*)malloc((cells + 1) * sizeof(size_t));
    for (int y = 0; y < dim; ++y)
        for (int x = 0; x < dim; ++x) {
            if (!map[(size_t)y * dim + x]) continue;                  /* lib.rs:74-76 */
            size_t area = 0;
            memset(mf2, 0, 2 * cells * sizeof(float));
            map[(size_t)y * dim + x] = 0;
            size_t sp = 0;
            stack[sp++] = (size_t)y * dim + x;
            while (sp) {                                              /* lib.rs:83-104 */
                size_t cur = stack[--sp];
                long cx = (long)(cur % (size_t)dim), cy = (long)(cur / (size_t)dim);
                ++area;
                for (long ox = -1; ox <= 1; ++ox)                     /* x-offset outer, y inner */
                    for (long oy = -1; oy <= 1; ++oy) {
                        long nx = cx + ox, ny = cy + oy;
                        if (nx < 0 || nx >= dim || ny < 0 || ny >= dim) continue;
                        size_t ni = (size_t)ny * dim + (size_t)nx;
                        if (map[ni]) {
                            mf2[2 * ni] = mf[2 * ni]; mf2[2 * ni + 1] = mf[2 * ni + 1];
                            stack[sp++] = ni;
                            map[ni] = 0;
                        }
                    }
            }
            if (area > biggest_area) {                                /* lib.rs:106-109 */
                biggest_area = area;
                if (!biggest_mf) biggest_mf = (float*)malloc((2 * cells + 2) * sizeof(float));
                memcpy(biggest_mf, mf2, 2 * cells * sizeof(float));
            }
        }

    int some = 0;
    if ((float)biggest_area / (float)(cells) >= min_size && biggest_mf) {   /* lib.rs:114-118 */
        some = 1;
        memcpy(out_field, biggest_mf, 2 * cells * sizeof(float));
    } else {
        memset(out_field, 0, 2 * cells * sizeof(float));
    }
    if (out_area) *out_area = some ? biggest_area : 0;
    if (out_dim) *out_dim = dim;
    free(mf); free(map); free(mf2); free(stack); free(biggest_mf);
    return some;
}

/* ------------------------------------------------------------------------------------ */
/* Almeida estimator                                                                    */
/* ------------------------------------------------------------------------------------ */

/* almeida-estimator/src/lib.rs:17-18 */
static float almeida_eps(void) { return 0.001f * 3.14159265358979323846264338327950288f / 180.0f; }
#define ORC_ALPHA 0.5f

/* One pass of the loop body, :140-183: residual and the three prototypes per vector, A and b as sequential f32 sums in
 * input order, partial-pivot LU; `model` is the raw solution (in units of EPS; zero when LU fails, :183). */
/* threads > 1 (bench.py's all-core CPU leg): the per-vector loop is split over OpenMP threads and the
 * twelve dot products -- each still ONE sequential f32 sum in input order -- run side by side, so the result is the
 * single-thread result bit for bit (tests/test_oracle.py). */
static void almeida_model_mt(const float* entries, size_t n, const orc_camera* cam, const float rotation[4], float model[3],
                             int threads);
void orc_almeida_model(const float* entries, size_t n, const orc_camera* cam, const float rotation[4], float model[3]) {
    almeida_model_mt(entries, n, cam, rotation, model, 1);
}
static void almeida_model_mt(const float* entries, size_t n, const orc_camera* cam, const float rotation[4], float model[3],
                             int threads) {
    const float EPS = almeida_eps();
    float m_roll[16], m_pitch[16], m_yaw[16], rotm[16];
    orc_mat4_from_euler(0.0f, EPS, 0.0f, m_roll);                     /* :30-34 */
    orc_mat4_from_euler(EPS, 0.0f, 0.0f, m_pitch);                    /* :36-38 */
    orc_mat4_from_euler(0.0f, 0.0f, -EPS, m_yaw);                     /* :40-42 */
    orc_quat_to_homogeneous(rotation, rotm);                          /* :140 */
    float* v = (float*)malloc((8 * n + 8) * sizeof(float));           /* [motion-delta, roll, pitch, yaw] */
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (size_t i = 0; i < n; ++i) {                                  /* :142-157 */
        const float* e = entries + 4 * i;
        float d[2];
        orc_camera_delta(cam, e, rotm, d);
        float* vi = v + 8 * i;
        vi[0] = e[2] - d[0]; vi[1] = e[3] - d[1];
        orc_camera_delta(cam, e, m_roll, vi + 2);
        orc_camera_delta(cam, e, m_pitch, vi + 4);
        orc_camera_delta(cam, e, m_yaw, vi + 6);
    }
    /* :159-179: each dot summed over the Vec in input order, starting from 0 */
    float a[9], b[3];
#pragma omp parallel for schedule(static, 1) num_threads(threads < 12 ? threads : 12) if (threads > 1)
    for (int k = 0; k < 12; ++k) {
        if (k < 9) {
            const int c = k / 3, r = k % 3;
            float acc = 0.0f;
            for (size_t i = 0; i < n; ++i) {
                const float* p = v + 8 * i + 2 * (c + 1);
                const float* s = v + 8 * i + 2 * (r + 1);
                acc += p[0] * s[0] + p[1] * s[1];
            }
            a[3 * r + c] = acc;       /* from_iterator is column-major: element k -> (k%3, k/3) */
        } else {
            const int r = k - 9;
            float acc = 0.0f;
            for (size_t i = 0; i < n; ++i) {
                const float* p = v + 8 * i + 2 * (r + 1);
                const float* s = v + 8 * i;
                acc += p[0] * s[0] + p[1] * s[1];
            }
            b[r] = acc;
        }
    }
    free(v);
    if (!orc_lu3_solve(a, b, model)) { model[0] = model[1] = model[2] = 0.0f; }   /* :181-183 */
}

/* The solver with its two constants and the composition order of the per-step rotation exposed.  The reference
 * today is (ALPHA 0.5, limit ceil(15/ALPHA) = 30, order 0 = pitch*roll*yaw, :18,:132,:193); order 1 = yaw*pitch*roll
 * is what the build that drew docs/report/mfield used (tests/test_reference_vectors.py).  q_steps (optional): the
 * cumulative `rotation` after every step, 4 floats each -- NOT inverted. */
static void solve_ypr_given_mt(const float* entries, size_t n, const orc_camera* cam, float alpha_c, size_t limit,
                               int order, float* q_steps, float q_out[4], int threads);
void orc_solve_ypr_given_ex(const float* entries, size_t n, const orc_camera* cam, float alpha_c, size_t limit,
                            int order, float* q_steps, float q_out[4]) {
    solve_ypr_given_mt(entries, n, cam, alpha_c, limit, order, q_steps, q_out, 1);
}
static void solve_ypr_given_mt(const float* entries, size_t n, const orc_camera* cam, float alpha_c, size_t limit,
                               int order, float* q_steps, float q_out[4], int threads) {
    const float EPS = almeida_eps();
    float rotation[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    for (size_t it = 0; it < limit; ++it) {
        float alpha = (it == limit - 1) ? 1.0f : alpha_c;             /* :138 */
        float model[3];
        almeida_model_mt(entries, n, cam, rotation, model, threads);
        model[0] = model[0] * EPS * alpha;                            /* :185 */
        model[1] = model[1] * EPS * alpha;
        model[2] = model[2] * EPS * alpha;
        float roll[4], pitch[4], yaw[4], pr[4], rot[4], nr[4];
        orc_quat_from_euler(0.0f, model[0], 0.0f, roll);              /* :189-191 */
        orc_quat_from_euler(model[1], 0.0f, 0.0f, pitch);
        orc_quat_from_euler(0.0f, 0.0f, -model[2], yaw);
        if (order == 0) {
            orc_quat_mul(pitch, roll, pr);                            /* :193 */
            orc_quat_mul(pr, yaw, rot);
        } else {
            orc_quat_mul(yaw, pitch, pr);
            orc_quat_mul(pr, roll, rot);
        }
        orc_quat_mul(rotation, rot, nr);                              /* :195 */
        memcpy(rotation, nr, sizeof(nr));
        if (q_steps) memcpy(q_steps + 4 * it, rotation, sizeof(nr));
    }
    orc_quat_inverse(rotation, q_out);                                /* :199 */
}

void orc_solve_ypr_given(const float* entries, size_t n, const orc_camera* cam, float q_out[4]) {
    orc_solve_ypr_given_ex(entries, n, cam, ORC_ALPHA, (size_t)ceilf(15.0f / ORC_ALPHA) /* :132 */, 0, NULL, q_out);
}
/* the same solve on `threads` host threads: identical bits (see almeida_model_mt) */
void orc_solve_ypr_given_mt(const float* entries, size_t n, const orc_camera* cam, int threads, float q_out[4]) {
    solve_ypr_given_mt(entries, n, cam, ORC_ALPHA, (size_t)ceilf(15.0f / ORC_ALPHA), 0, NULL, q_out, threads);
}

/* --- counter-based sampler standing in for rand::thread_rng + choose_multiple.
 * A keyed 4-round Feistel permutation of [0, 2^b) with cycle walking down to [0, n):
 * distinct indices, any index computable independently (needed by the GPU). --- */
static uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
uint32_t orc_sample_index(uint64_t seed, uint32_t iter, uint32_t stream, uint32_t i, uint32_t n) {
    if (n <= 1) return 0;
    uint32_t bits = 2;
    while (bits < 32 && (1ull << bits) < (uint64_t)n) bits += 2;      /* even number of bits */
    uint32_t half = bits / 2, mask = (1u << half) - 1u;
    uint64_t k = mix64(seed ^ mix64(((uint64_t)iter << 1) | (uint64_t)(stream & 1u)));
    uint32_t key[4];
    for (int r = 0; r < 4; ++r) key[r] = (uint32_t)mix64(k + (uint64_t)r);
    uint32_t x = i;
    do {
        uint32_t l = x >> half, r = x & mask;
        for (int round = 0; round < 4; ++round) {
            uint32_t f = r * 0x9E3779B1u + key[round];
            f ^= f >> 15; f *= 0x85EBCA77u; f ^= f >> 13;
            uint32_t nl = r, nr = l ^ (f & mask);
            l = nl; r = nr;
        }
        x = (l << half) | r;
    } while (x >= n);
    return x;
}

/* The hypotheses of :214-245 are independent of each other: on `threads` host threads each thread evaluates a strided
 * share of them and the winner is the FIRST hypothesis with the largest inlier set, as in the sequential loop -> the same
 * bits as orc_solve_ypr_ransac (tests/test_oracle.py).  bench.py's all-core CPU leg. */
void orc_solve_ypr_ransac_mt(const float* entries, size_t n, const orc_camera* cam, size_t num_iters, float inlier_deg,
                             size_t num_samples, uint64_t seed, int threads, float q_out[4]) {
    float target_delta = to_radians(inlier_deg);
    size_t ns = num_samples < n ? num_samples : n;
    size_t n3 = n < 3 ? n : 3;
    float thr2 = target_delta * target_delta;
    if (threads < 1) threads = 1;
    size_t* lens = (size_t*)calloc(num_iters + 1, sizeof(size_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (threads > 1)
    for (size_t it = 0; it < num_iters; ++it) {
        float samples[12];
        for (size_t j = 0; j < n3; ++j) {
            uint32_t idx = orc_sample_index(seed, (uint32_t)it, 0, (uint32_t)j, (uint32_t)n);
            memcpy(samples + 4 * j, entries + 4 * (size_t)idx, 4 * sizeof(float));
        }
        float fit[4], inv[4], mat[16];
        orc_solve_ypr_given(samples, n3, cam, fit);
        orc_quat_inverse(fit, inv);
        orc_quat_to_homogeneous(inv, mat);
        size_t len = 0;
        for (size_t j = 0; j < ns; ++j) {
            uint32_t idx = orc_sample_index(seed, (uint32_t)it, 1, (uint32_t)j, (uint32_t)n);
            const float* e = entries + 4 * (size_t)idx;
            float d[2], sample[2], vec[2], ang[2];
            orc_camera_delta(cam, e, mat, d);
            sample[0] = e[0] + d[0]; sample[1] = e[1] + d[1];
            vec[0] = e[2] - d[0]; vec[1] = e[3] - d[1];
            orc_camera_point_angle(cam, sample, ang);
            float vx = vec[0] * cosf(ang[0]), vy = vec[1] * cosf(ang[1]);
            if (vx * vx + vy * vy <= thr2) ++len;
        }
        lens[it] = len;
    }
    size_t best_it = 0, best_len = 0;
    for (size_t it = 0; it < num_iters; ++it)
        if (lens[it] > best_len) { best_len = lens[it]; best_it = it; }      /* strict >: the first maximum, :243 */
    free(lens);
    if (best_len >= 3) {
        /* the winner's inlier set again, in sample order, then the refit (:247-251) */
        float samples[12];
        for (size_t j = 0; j < n3; ++j) {
            uint32_t idx = orc_sample_index(seed, (uint32_t)best_it, 0, (uint32_t)j, (uint32_t)n);
            memcpy(samples + 4 * j, entries + 4 * (size_t)idx, 4 * sizeof(float));
        }
        float fit[4], inv[4], mat[16];
        orc_solve_ypr_given(samples, n3, cam, fit);
        orc_quat_inverse(fit, inv);
        orc_quat_to_homogeneous(inv, mat);
        float* sel = (float*)malloc((4 * best_len + 4) * sizeof(float));
        size_t len = 0;
        for (size_t j = 0; j < ns; ++j) {
            uint32_t idx = orc_sample_index(seed, (uint32_t)best_it, 1, (uint32_t)j, (uint32_t)n);
            const float* e = entries + 4 * (size_t)idx;
            float d[2], sample[2], vec[2], ang[2];
            orc_camera_delta(cam, e, mat, d);
            sample[0] = e[0] + d[0]; sample[1] = e[1] + d[1];
            vec[0] = e[2] - d[0]; vec[1] = e[3] - d[1];
            orc_camera_point_angle(cam, sample, ang);
            float vx = vec[0] * cosf(ang[0]), vy = vec[1] * cosf(ang[1]);
            if (vx * vx + vy * vy <= thr2 && len < best_len) { memcpy(sel + 4 * len, e, 4 * sizeof(float)); ++len; }
        }
        solve_ypr_given_mt(sel, len, cam, ORC_ALPHA, (size_t)ceilf(15.0f / ORC_ALPHA), 0, NULL, q_out, threads);
        free(sel);
    } else {
        q_out[0] = 1.0f; q_out[1] = q_out[2] = q_out[3] = 0.0f;
    }
}

void orc_solve_ypr_ransac(const float* entries, size_t n, const orc_camera* cam,
                          size_t num_iters, float inlier_deg, size_t num_samples, uint64_t seed,
                          float q_out[4], uint32_t* out_inliers, size_t* out_n_inliers) {
    float target_delta = to_radians(inlier_deg);                      /* :210 */
    size_t ns = num_samples < n ? num_samples : n;
    size_t n3 = n < 3 ? n : 3;
    uint32_t* best = (uint32_t*)malloc((ns + 1) * sizeof(uint32_t));
    uint32_t* cur = (uint32_t*)malloc((ns + 1) * sizeof(uint32_t));
    size_t best_len = 0;
    float thr2 = target_delta * target_delta;
    for (size_t it = 0; it < num_iters; ++it) {                       /* :214 */
        float samples[12];
        for (size_t j = 0; j < n3; ++j) {                             /* :215 */
            uint32_t idx = orc_sample_index(seed, (uint32_t)it, 0, (uint32_t)j, (uint32_t)n);
            memcpy(samples + 4 * j, entries + 4 * (size_t)idx, 4 * sizeof(float));
        }
        float fit[4], inv[4], mat[16];
        orc_solve_ypr_given(samples, n3, cam, fit);                   /* :217 */
        orc_quat_inverse(fit, inv);
        orc_quat_to_homogeneous(inv, mat);                            /* :224 */
        size_t len = 0;
        for (size_t j = 0; j < ns; ++j) {                             /* :219-241 */
            uint32_t idx = orc_sample_index(seed, (uint32_t)it, 1, (uint32_t)j, (uint32_t)n);
            const float* e = entries + 4 * (size_t)idx;
            float d[2], sample[2], vec[2], ang[2];
            orc_camera_delta(cam, e, mat, d);
            sample[0] = e[0] + d[0]; sample[1] = e[1] + d[1];
            vec[0] = e[2] - d[0]; vec[1] = e[3] - d[1];
            orc_camera_point_angle(cam, sample, ang);
            float vx = vec[0] * cosf(ang[0]), vy = vec[1] * cosf(ang[1]);
            if (vx * vx + vy * vy <= thr2) cur[len++] = idx;
        }
        if (len > best_len) {                                         /* :243-245 */
            best_len = len;
            memcpy(best, cur, len * sizeof(uint32_t));
        }
    }
    if (best_len >= 3) {                                              /* :247-251 */
        float* sel = (float*)malloc((4 * best_len + 4) * sizeof(float));
        for (size_t j = 0; j < best_len; ++j) memcpy(sel + 4 * j, entries + 4 * (size_t)best[j], 4 * sizeof(float));
        orc_solve_ypr_given(sel, best_len, cam, q_out);
        free(sel);
    } else {
        q_out[0] = 1.0f; q_out[1] = q_out[2] = q_out[3] = 0.0f;
    }
    if (out_n_inliers) *out_n_inliers = best_len;
    if (out_inliers) memcpy(out_inliers, best, best_len * sizeof(uint32_t));
    free(best); free(cur);
}

/* ------------------------------------------------------------------------------------ */
/* N1: full-search SAD block matcher (build-defined; no reference counterpart)           */
/* ------------------------------------------------------------------------------------ */

/* SAD of one candidate.  The scalar loop is the definition; the SSE2 psadbw path (B = 8 or 16) is an
 * exact integer shortcut used to give the timed CPU baseline a fair inner loop. */
static uint32_t sad_candidate(const uint8_t* c, const uint8_t* p, int stride, int B, int simd) {
#if defined(__SSE2__)
    if (simd && B == 16) {
        __m128i acc = _mm_setzero_si128();
        for (int y = 0; y < 16; ++y)
            acc = _mm_add_epi64(acc, _mm_sad_epu8(_mm_loadu_si128((const __m128i*)(c + (size_t)y * stride)),
                                                  _mm_loadu_si128((const __m128i*)(p + (size_t)y * stride))));
        return (uint32_t)(_mm_cvtsi128_si32(acc) + _mm_cvtsi128_si32(_mm_srli_si128(acc, 8)));
    }
    if (simd && B == 8) {
        __m128i acc = _mm_setzero_si128();
        for (int y = 0; y < 8; ++y)
            acc = _mm_add_epi64(acc, _mm_sad_epu8(_mm_loadl_epi64((const __m128i*)(c + (size_t)y * stride)),
                                                  _mm_loadl_epi64((const __m128i*)(p + (size_t)y * stride))));
        return (uint32_t)_mm_cvtsi128_si32(acc);
    }
#endif
    (void)simd;
    uint32_t sad = 0;
    for (int y = 0; y < B; ++y) {
        const uint8_t* cr = c + (size_t)y * stride;
        const uint8_t* pr = p + (size_t)y * stride;
        for (int x = 0; x < B; ++x) sad += (uint32_t)abs((int)cr[x] - (int)pr[x]);
    }
    return sad;
}

/* ---- AVX2 inner loop for the timed CPU baseline (16x16 blocks): vmpsadbw evaluates EIGHT consecutive dx candidates
 * of a 4-byte column group per instruction -- the instruction x264-class encoders build their full search on -- with
 * two block rows per 256-bit register.  Exact integer arithmetic (16x16 SAD <= 65,280 fits the u16 accumulators), so
 * it returns the spec's numbers; selected at run time (__builtin_cpu_supports), SSE2 psadbw otherwise.
 * out[i] = SAD of the block at `cur` against the block at prev + i, i = 0..7; reads prev[0 .. 23] of every row. */
#if defined(__x86_64__)
__attribute__((target("avx2")))
static void sad16_dx8_avx2(const __m256i crow[8], const uint8_t* prev, int stride, uint16_t out[8]) {
    __m256i acc = _mm256_setzero_si256();
    for (int y = 0; y < 8; ++y) {                                       /* row pair 2y, 2y+1 */
        const uint8_t* p0 = prev + (size_t)(2 * y) * stride;
        const uint8_t* p1 = p0 + stride;
        const __m256i b0 = _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128((const __m128i*)p0)),
                                                   _mm_loadu_si128((const __m128i*)p1), 1);           /* bytes 0..15 */
        const __m256i b1 = _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128((const __m128i*)(p0 + 8))),
                                                   _mm_loadu_si128((const __m128i*)(p1 + 8)), 1);     /* bytes 8..23 */
        /* imm per lane: bit 2 = window offset 4 in the first operand, bits 1:0 = 4-byte group of the second */
        acc = _mm256_add_epi16(acc, _mm256_mpsadbw_epu8(b0, crow[y], 0x00));      /* columns  0.. 3 */
        acc = _mm256_add_epi16(acc, _mm256_mpsadbw_epu8(b0, crow[y], 0x2D));      /* columns  4.. 7 */
        acc = _mm256_add_epi16(acc, _mm256_mpsadbw_epu8(b1, crow[y], 0x12));      /* columns  8..11 */
        acc = _mm256_add_epi16(acc, _mm256_mpsadbw_epu8(b1, crow[y], 0x3F));      /* columns 12..15 */
    }
    const __m128i s = _mm_add_epi16(_mm256_castsi256_si128(acc), _mm256_extracti128_si256(acc, 1));
    _mm_storeu_si128((__m128i*)out, s);
}
__attribute__((target("avx2")))
static void load_cur16_avx2(const uint8_t* c, int stride, __m256i crow[8]) {
    for (int y = 0; y < 8; ++y)
        crow[y] = _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128((const __m128i*)(c + (size_t)(2 * y) * stride))),
                                          _mm_loadu_si128((const __m128i*)(c + (size_t)(2 * y + 1) * stride)), 1);
}
static int sad_have_avx2(void) {
    static int have = -1;
    if (have < 0) have = __builtin_cpu_supports("avx2") ? 1 : 0;
    return have;
}
#else
static int sad_have_avx2(void) { return 0; }
#endif

/* what the timed baseline runs on this host: 2 = AVX2 vmpsadbw, 1 = SSE2 psadbw, 0 = scalar */
int orc_sad_simd_level(void) {
#if defined(__SSE2__)
    return sad_have_avx2() ? 2 : 1;
#else
    return 0;
#endif
}

static void sad_block_run(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride,
                          int B, int R, int by, int bx_begin, int bx_end, int nbx, float* out_entries,
                          int32_t* out_best, int simd) {
    const float nx = 1.0f / (float)W, ny = 1.0f / (float)H;          /* av-decoder/src/lib.rs:404-405 */
    const int wide = simd && B == 16 && sad_have_avx2();
    for (int bx = bx_begin; bx < bx_end; ++bx) {
        int x0 = bx * B, y0 = by * B;
        uint64_t best_key = ~0ull; int best_dx = 0, best_dy = 0; uint32_t best_sad = 0;
#if defined(__x86_64__)
        __m256i crow[8];
        if (wide) load_cur16_avx2(cur + (size_t)y0 * stride + x0, stride, crow);
#endif
        for (int dy = -R; dy <= R; ++dy) {
            if (y0 + dy < 0 || y0 + dy + B > H) continue;
            for (int dx = -R; dx <= R; ++dx) {
                if (x0 + dx < 0 || x0 + dx + B > W) continue;
#if defined(__x86_64__)
                /* eight candidates dx .. dx+7 at once when all of them are valid and the 24-byte row reads stay inside
                 * the frame row; the argmin key is a total order, so the evaluation order does not matter */
                if (wide && dx + 7 <= R && x0 + dx + 7 + B <= W && x0 + dx + 24 <= W) {
                    uint16_t s8[8];
                    sad16_dx8_avx2(crow, prev + (size_t)(y0 + dy) * stride + x0 + dx, stride, s8);
                    for (int i = 0; i < 8; ++i) {
                        const int d = dx + i;
                        const uint64_t key = ((uint64_t)s8[i] << 32) | ((uint64_t)(uint32_t)(d * d + dy * dy) << 16) |
                                             ((uint64_t)(uint32_t)(dy + R) << 8) | (uint64_t)(uint32_t)(d + R);
                        if (key < best_key) { best_key = key; best_dx = d; best_dy = dy; best_sad = s8[i]; }
                    }
                    dx += 7;
                    continue;
                }
#endif
                uint32_t sad = sad_candidate(cur + (size_t)y0 * stride + x0,
                                             prev + (size_t)(y0 + dy) * stride + x0 + dx, stride, B, simd);
                uint64_t key = ((uint64_t)sad << 32) | ((uint64_t)(uint32_t)(dx * dx + dy * dy) << 16) |
                               ((uint64_t)(uint32_t)(dy + R) << 8) | (uint64_t)(uint32_t)(dx + R);
                if (key < best_key) { best_key = key; best_dx = dx; best_dy = dy; best_sad = sad; }
            }
        }
        size_t k = (size_t)by * nbx + bx;
        float cx = (float)(x0 + B / 2 + best_dx), cy = (float)(y0 + B / 2 + best_dy);
        out_entries[4 * k + 0] = cx * nx;                             /* :409-411: src * frame_norm */
        out_entries[4 * k + 1] = cy * ny;
        out_entries[4 * k + 2] = ((float)best_dx / 1.0f) * (-nx);     /* :412-417 */
        out_entries[4 * k + 3] = ((float)best_dy / 1.0f) * (-ny);
        if (out_best) { out_best[3 * k] = best_dx; out_best[3 * k + 1] = best_dy; out_best[3 * k + 2] = (int32_t)best_sad; }
    }
}

size_t orc_sad_flow(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride,
                    int B, int R, float* out_entries, int32_t* out_best, int threads) {
    return orc_sad_flow_ex(prev, cur, W, H, stride, B, R, out_entries, out_best, threads, 1);
}

size_t orc_sad_flow_ex(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride,
                       int B, int R, float* out_entries, int32_t* out_best, int threads, int simd) {
    int nbx = W / B, nby = H / B;
    if (R > 127) return 0;          /* key packing above holds dy+R, dx+R in 8 bits */
    (void)threads;
#ifdef _OPENMP
    if (threads > 1) {
        /* one task = a run of up to 8 blocks of one block row: enough tasks for 100+ threads */
        const int runs = (nbx + 7) / 8;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (int t = 0; t < nby * runs; ++t) {
            const int by = t / runs, b0 = (t % runs) * 8;
            const int b1 = b0 + 8 < nbx ? b0 + 8 : nbx;
            sad_block_run(prev, cur, W, H, stride, B, R, by, b0, b1, nbx, out_entries, out_best, simd);
        }
        return (size_t)nbx * (size_t)nby;
    }
#endif
    for (int by = 0; by < nby; ++by) sad_block_run(prev, cur, W, H, stride, B, R, by, 0, nbx, nbx, out_entries, out_best, simd);
    return (size_t)nbx * (size_t)nby;
}

/* ---- cv-decoder's contrast mask (cv-decoder/src/lib.rs:203-237) --------------------------------------------
 * Sobel(gray, CV_32F, dx=1, dy=1, ksize=5, scale 1, delta 0, BORDER_DEFAULT) -> threshold(> 20 -> 255, else 0)
 * -> dilate(getStructuringElement(MORPH_ELLIPSE, 11x11, anchor (5,5)), 1 iteration, default border value).
 * OpenCV itself is not under /root/reference and not installed ("parity unpinned"); this restates its
 * published definitions: getDerivKernels(ksize 5): order-1 taps [-1,-2,0,2,1] on both axes for dx=dy=1,
 * applied as a correlation; BORDER_DEFAULT = BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba); the ellipse rows
 * are j in [c-dx, c+dx] with dx = cvRound(c*sqrt((r*r-dy*dy)/(r*r))), r = c = 5; dilation's default border value
 * makes out-of-image taps never win the max.  All sums are integers < 2^24, exact in f32, so the mask is
 * integer-exact by definition.  out_mask: W bytes per row (1 = keep the pixel: `mask >= 0.1`, lib.rs:253-257). */
static int orc_reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - p - 2;
    return p;
}

void orc_contrast_mask(const uint8_t* gray, int W, int H, int stride, uint8_t* out_mask) {
    static const int K[5] = {-1, -2, 0, 2, 1};
    int hw[11];
    for (int i = 0; i < 11; ++i) {
        int dy = i - 5;
        hw[i] = (int)lrint(5.0 * sqrt((25.0 - (double)(dy * dy)) * (1.0 / 25.0)));
    }
    uint8_t* thr = (uint8_t*)malloc((size_t)W * H + 1);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float acc = 0.0f;                                        /* CV_32F accumulation; exact integers */
            for (int i = 0; i < 5; ++i) {
                const uint8_t* row = gray + (size_t)orc_reflect101(y + i - 2, H) * stride;
                for (int j = 0; j < 5; ++j)
                    acc += (float)(K[i] * K[j]) * (float)row[orc_reflect101(x + j - 2, W)];
            }
            thr[(size_t)y * W + x] = acc > 20.0f ? 255 : 0;          /* THRESH_BINARY */
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int m = 0;
            for (int i = 0; i < 11 && !m; ++i) {
                int yy = y + i - 5;
                if (yy < 0 || yy >= H) continue;
                for (int xx = x - hw[i]; xx <= x + hw[i]; ++xx)
                    if (xx >= 0 && xx < W && thr[(size_t)yy * W + xx]) { m = 1; break; }
            }
            out_mask[(size_t)y * W + x] = (uint8_t)m;
        }
    free(thr);
}

/* the masked per-pixel loop of cv-decoder/src/lib.rs:251-276 in the non-"Process Fullres" form: records of
 * the unmasked pixels in raster order.  Returns the record count. */
size_t orc_masked_flow_to_entries(const float* flow, const uint8_t* mask, int W, int H, float* out_entries) {
    const float nx = 1.0f / (float)W, ny = 1.0f / (float)H;
    size_t k = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (mask && !mask[(size_t)y * W + x]) continue;
            float* e = out_entries + 4 * k++;
            e[0] = ((float)x + 0.5f) * nx;
            e[1] = ((float)y + 0.5f) * ny;
            e[2] = flow[2 * ((size_t)y * W + x)] * nx;
            e[3] = flow[2 * ((size_t)y * W + x) + 1] * ny;
        }
    return k;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads the OpenMP loops without a thread argument of their own (orc_lk_flow*) use from now on; returns the previous
 * setting.  bench_legs.py times the N2 restatement on one thread and on all the host cores with it. */
int orc_set_num_threads(int n) {
#ifdef _OPENMP
    const int prev = omp_get_max_threads();
    if (n >= 1) omp_set_num_threads(n);
    return prev;
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* N2: dense pyramidal Lucas-Kanade flow (build-defined; the reference only calls OpenCV) */
/* ------------------------------------------------------------------------------------ */
/* Spec (DESIGN.md "N2"): f32 throughout, no FMA, every loop in the order written here.
 *   pyramid   level 0 = luma as f32; level l+1 = [1 4 6 4 1]/16 separable blur (replicated border),
 *             every second sample; size (w+1)/2 x (h+1)/2
 *   gradient  central differences * 0.5 on the PREVIOUS frame, replicated border
 *   G         per pixel sums over the (2r+1)^2 window (dy outer, dx inner, clamped coordinates) of
 *             Ix*Ix, Ix*Iy, Iy*Iy
 *   step      b = sum_w grad(q) * (I(q) - J(q + flow(p))), J sampled bilinearly with edge clamp;
 *             flow(p) += G^-1 b when det(G) > 0.01; `iters` steps per level
 *   levels    coarse to fine, flow_l(x,y) = 2 * flow_{l+1}(x/2, y/2); the coarsest level starts at 0
 * Convention: prev(x,y) ~ cur(x+u, y+v), as calcOpticalFlowFarneback's output that cv-decoder consumes
 * (cv-decoder/src/lib.rs:188-199,262-269). */
static int lk_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void lk_pyr_down(const float* in, int w, int h, float* out, int w1, int h1, float* tmp /* w1*h */) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w1; ++x) {
            const float* r = in + (size_t)y * w;
            float a = r[lk_clampi(2 * x - 2, 0, w - 1)], b = r[lk_clampi(2 * x - 1, 0, w - 1)], c = r[lk_clampi(2 * x, 0, w - 1)],
                  d = r[lk_clampi(2 * x + 1, 0, w - 1)], e = r[lk_clampi(2 * x + 2, 0, w - 1)];
            tmp[(size_t)y * w1 + x] = ((((a + 4.0f * b) + 6.0f * c) + 4.0f * d) + e) * 0.0625f;
        }
    for (int y = 0; y < h1; ++y)
        for (int x = 0; x < w1; ++x) {
            float a = tmp[(size_t)lk_clampi(2 * y - 2, 0, h - 1) * w1 + x], b = tmp[(size_t)lk_clampi(2 * y - 1, 0, h - 1) * w1 + x],
                  c = tmp[(size_t)lk_clampi(2 * y, 0, h - 1) * w1 + x], d = tmp[(size_t)lk_clampi(2 * y + 1, 0, h - 1) * w1 + x],
                  e = tmp[(size_t)lk_clampi(2 * y + 2, 0, h - 1) * w1 + x];
            out[(size_t)y * w1 + x] = ((((a + 4.0f * b) + 6.0f * c) + 4.0f * d) + e) * 0.0625f;
        }
}

/* N2 spec, revision 2 (DESIGN.md "N2"): the bilinear sample, the two residual sums and the three structure-tensor sums
 * fuse their multiply-adds -- lerp(a, b, t) = fma(t, b - a, a), b += g * d as fma(g, d, b), gxx += ix * ix as
 * fma(ix, ix, gxx): 7 operations per window tap and step instead of 11 (+ 3 instead of 6 per tap for the tensor), each
 * result rounded once instead of twice.  ORC_LK_SPEC_FMA = 0 rebuilds revision 1 (separate multiply and add) for A/B runs;
 * ofps_amd/csrc/lk.hip carries the same switch (OFPS_LK_SPEC_FMA) and the two must be built alike.  fmaf() is the
 * correctly rounded fused operation whatever the host: orc_lk_flow is cloned for FMA3 hosts (one vfmadd instruction) with
 * the libm call as the portable clone. */
#ifndef ORC_LK_SPEC_FMA
#define ORC_LK_SPEC_FMA 1
#endif
static inline float lk_lerp(float a, float b, float t) {
#if ORC_LK_SPEC_FMA
    return fmaf(t, b - a, a);
#else
    return a + t * (b - a);
#endif
}
static inline float lk_accum(float g, float d, float b) {
#if ORC_LK_SPEC_FMA
    return fmaf(g, d, b);
#else
    return b + g * d;
#endif
}
int orc_lk_spec_revision(void) { return ORC_LK_SPEC_FMA ? 2 : 1; }

static inline float lk_bilinear(const float* J, int w, int h, float fx, float fy) {
    float x0f = floorf(fx), y0f = floorf(fy);
    float ax = fx - x0f, ay = fy - y0f;
    /* clamp in float first so wild flows cannot overflow the int conversion */
    float cx = x0f < -1.0f ? -1.0f : (x0f > (float)w ? (float)w : x0f);
    float cy = y0f < -1.0f ? -1.0f : (y0f > (float)h ? (float)h : y0f);
    int x0 = (int)cx, y0 = (int)cy;
    int xa = lk_clampi(x0, 0, w - 1), xb = lk_clampi(x0 + 1, 0, w - 1);
    int ya = lk_clampi(y0, 0, h - 1), yb = lk_clampi(y0 + 1, 0, h - 1);
    float j00 = J[(size_t)ya * w + xa], j10 = J[(size_t)ya * w + xb], j01 = J[(size_t)yb * w + xa], j11 = J[(size_t)yb * w + xb];
    float top = lk_lerp(j00, j10, ax);
    float bot = lk_lerp(j01, j11, ax);
    return lk_lerp(top, bot, ay);
}

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("fma", "default")))
#endif
int orc_lk_flow_trace(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int radius, int iters,
                      const float* init /* coarsest level's starting flow, 2 * ws * hs floats, or NULL = zero */, float* out_flow,
                      float* trace /* or NULL: the flow ENTERING every step, coarsest level first, iters planes of 2 * w * h per level */) {
    if (levels < 1 || levels > 8 || radius < 1 || radius > 15 || iters < 1 || W < 1 || H < 1) return 0;
    int ws[8], hs[8];
    float *I[8], *J[8];
    ws[0] = W; hs[0] = H;
    for (int l = 1; l < levels; ++l) { ws[l] = (ws[l - 1] + 1) / 2; hs[l] = (hs[l - 1] + 1) / 2; }
    for (int l = 0; l < levels; ++l) {
        I[l] = (float*)malloc((size_t)ws[l] * hs[l] * sizeof(float));
        J[l] = (float*)malloc((size_t)ws[l] * hs[l] * sizeof(float));
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            I[0][(size_t)y * W + x] = (float)prev[(size_t)y * stride + x];
            J[0][(size_t)y * W + x] = (float)cur[(size_t)y * stride + x];
        }
    float* tmp = (float*)malloc((size_t)W * H * sizeof(float));
    for (int l = 1; l < levels; ++l) {
        lk_pyr_down(I[l - 1], ws[l - 1], hs[l - 1], I[l], ws[l], hs[l], tmp);
        lk_pyr_down(J[l - 1], ws[l - 1], hs[l - 1], J[l], ws[l], hs[l], tmp);
    }
    float* flow = (float*)calloc((size_t)2 * W * H, sizeof(float));
    float* next = (float*)malloc((size_t)2 * W * H * sizeof(float));
    float* gx = (float*)malloc((size_t)W * H * sizeof(float));
    float* gy = (float*)malloc((size_t)W * H * sizeof(float));
    for (int l = levels - 1; l >= 0; --l) {
        const int w = ws[l], h = hs[l];
        const float* Il = I[l]; const float* Jl = J[l];
        if (l < levels - 1) {                                   /* flow of the coarser level, doubled */
            const int w1 = ws[l + 1], h1 = hs[l + 1];
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const int sx = lk_clampi(x / 2, 0, w1 - 1), sy = lk_clampi(y / 2, 0, h1 - 1);
                    next[2 * ((size_t)y * w + x)] = 2.0f * flow[2 * ((size_t)sy * w1 + sx)];
                    next[2 * ((size_t)y * w + x) + 1] = 2.0f * flow[2 * ((size_t)sy * w1 + sx) + 1];
                }
            memcpy(flow, next, (size_t)2 * w * h * sizeof(float));
        } else if (init) {                                      /* a caller-supplied prior (the role of OpenCV's OPTFLOW_USE_INITIAL_FLOW) */
            memcpy(flow, init, (size_t)2 * w * h * sizeof(float));
        } else {
            memset(flow, 0, (size_t)2 * w * h * sizeof(float));
        }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                gx[(size_t)y * w + x] = (Il[(size_t)y * w + lk_clampi(x + 1, 0, w - 1)] - Il[(size_t)y * w + lk_clampi(x - 1, 0, w - 1)]) * 0.5f;
                gy[(size_t)y * w + x] = (Il[(size_t)lk_clampi(y + 1, 0, h - 1) * w + x] - Il[(size_t)lk_clampi(y - 1, 0, h - 1) * w + x]) * 0.5f;
            }
        for (int it = 0; it < iters; ++it) {
            if (trace) { memcpy(trace, flow, (size_t)2 * w * h * sizeof(float)); trace += (size_t)2 * w * h; }
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const float u = flow[2 * ((size_t)y * w + x)], v = flow[2 * ((size_t)y * w + x) + 1];
                    float gxx = 0.0f, gxy = 0.0f, gyy = 0.0f, bx = 0.0f, by = 0.0f;
                    for (int dy = -radius; dy <= radius; ++dy)
                        for (int dx = -radius; dx <= radius; ++dx) {
                            const int qx = lk_clampi(x + dx, 0, w - 1), qy = lk_clampi(y + dy, 0, h - 1);
                            const float ix = gx[(size_t)qy * w + qx], iy = gy[(size_t)qy * w + qx];
                            const float d = Il[(size_t)qy * w + qx] - lk_bilinear(Jl, w, h, (float)qx + u, (float)qy + v);
                            gxx = lk_accum(ix, ix, gxx); gxy = lk_accum(ix, iy, gxy); gyy = lk_accum(iy, iy, gyy);
                            bx = lk_accum(ix, d, bx); by = lk_accum(iy, d, by);
                        }
                    const float det = gxx * gyy - gxy * gxy;
                    float du = 0.0f, dv = 0.0f;
                    if (det > 0.01f) {
                        du = (gyy * bx - gxy * by) / det;
                        dv = (gxx * by - gxy * bx) / det;
                    }
                    next[2 * ((size_t)y * w + x)] = u + du;
                    next[2 * ((size_t)y * w + x) + 1] = v + dv;
                }
            memcpy(flow, next, (size_t)2 * w * h * sizeof(float));
        }
    }
    memcpy(out_flow, flow, (size_t)2 * W * H * sizeof(float));
    for (int l = 0; l < levels; ++l) { free(I[l]); free(J[l]); }
    free(tmp); free(flow); free(next); free(gx); free(gy);
    return 1;
}

int orc_lk_flow_init(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int radius, int iters,
                     const float* init, float* out_flow) {
    return orc_lk_flow_trace(prev, cur, W, H, stride, levels, radius, iters, init, out_flow, NULL);
}
int orc_lk_flow(const uint8_t* prev, const uint8_t* cur, int W, int H, int stride, int levels, int radius, int iters,
                float* out_flow) {
    return orc_lk_flow_trace(prev, cur, W, H, stride, levels, radius, iters, NULL, out_flow, NULL);
}

/* cv-decoder's record convention for per-pixel flow (cv-decoder/src/lib.rs:239-243, 262-269):
 * pos = ((x+.5), (y+.5)) * (1/W, 1/H), motion = flow * (1/W, 1/H); raster order. */
void orc_flow_to_entries(const float* flow, int W, int H, float* out_entries) {
    const float nx = 1.0f / (float)W, ny = 1.0f / (float)H;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float* e = out_entries + 4 * ((size_t)y * W + x);
            e[0] = ((float)x + 0.5f) * nx;
            e[1] = ((float)y + 0.5f) * ny;
            e[2] = flow[2 * ((size_t)y * W + x)] * nx;
            e[3] = flow[2 * ((size_t)y * W + x) + 1] * ny;
        }
}
