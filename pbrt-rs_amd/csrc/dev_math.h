// dev_math.h — device-side scalar/vector arithmetic for the gfx950 kernels.
//
// Every function is a fixed sequence of IEEE-754 binary32/binary64 operations (compiled with
// -ffp-contract=off, correctly rounded divide/sqrt, denormals on), written in the operation
// order of the reference's Rust so hit decisions agree bit for bit with a CPU evaluation:
//   src/core/pbrt.rs:43-91          next_float_up/down, gamma
//   src/core/geometry.rs:62-393     Vector3 ops (dot = x*x' + y*y' + z*z' left to right; f32 cross)
//   src/core/geometry.rs:1139-1154  offset_ray_origin
//   src/core/rng.rs:21-48           PCG32
//   src/core/sampling.rs:258-313    concentric disk / cosine hemisphere / uniform triangle / power heuristic
//   src/core/reflection.rs:19-40, 142-156  fr_dielectric, refract
// sin/cos/acos/atan2 (Rust calls the platform libm, which defines no bit pattern) are
// evaluated by fixed fma polynomials (Cephes single-precision kernels) so they are reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PB_DEV __device__ __forceinline__

namespace pb {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 1.0f / kPi;
constexpr float kInv2Pi = kInvPi / 2.0f;
constexpr float kPiOver2 = kPi / 2.0f;
constexpr float kPiOver4 = kPi / 4.0f;
constexpr float kShadowEpsilon = 0.0001f;
constexpr float kEpsilon = 1.1920928955078125e-07f;          // f32::EPSILON
constexpr float kMachineEpsilon = 0.5f * kEpsilon;           // src/core/mod.rs:206
constexpr float kOneMinusEpsilon = 1.0f - kEpsilon;
constexpr float kInf = __builtin_huge_valf();
constexpr float kFloatMax = 3.402823466e+38f;

constexpr float gamma_c(float n) { return n * kMachineEpsilon / (1.0f - n * kMachineEpsilon); }
constexpr float kGamma2 = gamma_c(2.0f), kGamma3 = gamma_c(3.0f), kGamma5 = gamma_c(5.0f),
                kGamma6 = gamma_c(6.0f), kGamma7 = gamma_c(7.0f);
constexpr float kSlabScale = 1.0f + 2.0f * kGamma3;  // src/core/geometry.rs:722 (D2: intended for z too)

struct V3 {
    float x, y, z;
};
PB_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
PB_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PB_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PB_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
PB_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
PB_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
PB_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PB_DEV float absdot(V3 a, V3 b) { return __builtin_fabsf(dot(a, b)); }
PB_DEV float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
PB_DEV float length(V3 a) { return __builtin_sqrtf(len2(a)); }
PB_DEV V3 normalize(V3 a) { return a / length(a); }
PB_DEV V3 vabs(V3 a) { return V3{__builtin_fabsf(a.x), __builtin_fabsf(a.y), __builtin_fabsf(a.z)}; }
PB_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PB_DEV float fmaxr(float a, float b) { return a > b ? a : b; }
PB_DEV float fminr(float a, float b) { return a < b ? a : b; }
PB_DEV float max3(float a, float b, float c) { return fmaxr(a, fmaxr(b, c)); }
PB_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
PB_DEV float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// pbrt Faceforward(n, v): n flipped into v's hemisphere
PB_DEV V3 faceforward(V3 n, V3 v) { return dot(n, v) < 0.0f ? -n : n; }
PB_DEV void coordinate_system(V3 v1, V3* v2, V3* v3) {  // src/core/geometry.rs:374-382
    if (__builtin_fabsf(v1.x) > __builtin_fabsf(v1.y))
        *v2 = normalize(V3{-v1.z, 0.0f, v1.x});
    else
        *v2 = normalize(V3{0.0f, v1.z, -v1.y});
    *v3 = cross(v1, *v2);
}

PB_DEV float next_float_up(float n) {
    if (__builtin_isinf(n) && n > 0.0f) return n;
    if (n == -0.0f) n = 0.0f;
    uint32_t u = __float_as_uint(n);
    u = (n >= 0.0f) ? u + 1 : u - 1;
    return __uint_as_float(u);
}
PB_DEV float next_float_down(float n) {
    if (__builtin_isinf(n) && n < 0.0f) return n;
    if (n == 0.0f) n = -0.0f;
    uint32_t u = __float_as_uint(n);
    u = (n > 0.0f) ? u - 1 : u + 1;
    return __uint_as_float(u);
}
PB_DEV V3 offset_ray_origin(V3 p, V3 p_error, V3 n, V3 w) {
    float d = dot(vabs(n), p_error);
    V3 offset = n * d;
    if (dot(w, n) < 0.0f) offset = -offset;
    V3 po = p + offset;
    po.x = offset.x > 0.0f ? next_float_up(po.x) : (offset.x < 0.0f ? next_float_down(po.x) : po.x);
    po.y = offset.y > 0.0f ? next_float_up(po.y) : (offset.y < 0.0f ? next_float_down(po.y) : po.y);
    po.z = offset.z > 0.0f ? next_float_up(po.z) : (offset.z < 0.0f ? next_float_down(po.z) : po.z);
    return po;
}

// ---- reproducible elementary functions ----
PB_DEV void det_sincos(float x, float* s_out, float* c_out) {
    float q = x * 0.63661977236758134308f;
    float k = __builtin_rintf(q);
    float r = __builtin_fmaf(-k, 1.5707397460937500f, x);
    r = __builtin_fmaf(-k, 5.6579709053039550781e-05f, r);
    r = __builtin_fmaf(-k, 9.9209362947050294680e-10f, r);
    float z = r * r;
    float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    float sr = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    float cr = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    int ki = (int)k & 3;
    float s = (ki == 0) ? sr : (ki == 1) ? cr : (ki == 2) ? -sr : -cr;
    float c = (ki == 0) ? cr : (ki == 1) ? -sr : (ki == 2) ? -cr : sr;
    *s_out = s;
    *c_out = c;
}
PB_DEV float det_sin(float x) {
    float s, c;
    det_sincos(x, &s, &c);
    return s;
}
PB_DEV float det_asin_core(float x) {
    float z = x * x;
    float p = __builtin_fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z, 1.6666752422e-1f);
    return __builtin_fmaf(p * z, x, x);
}
PB_DEV float det_acos(float x) {
    if (x < -0.5f) {
        float t = __builtin_sqrtf(0.5f * (1.0f + x));
        return kPi - 2.0f * det_asin_core(t);
    }
    if (x > 0.5f) {
        float t = __builtin_sqrtf(0.5f * (1.0f - x));
        return 2.0f * det_asin_core(t);
    }
    return kPiOver2 - det_asin_core(x);
}
PB_DEV float det_atan_pos(float x) {
    float y;
    if (x > 2.414213562373095f) {
        y = kPiOver2;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = kPiOver4;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    float p = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    return y + __builtin_fmaf(p * z, x, x);
}
PB_DEV float det_atan2(float y, float x) {
    if (x == 0.0f) return y > 0.0f ? kPiOver2 : (y < 0.0f ? -kPiOver2 : 0.0f);
    float a = det_atan_pos(__builtin_fabsf(y / x));
    if (x < 0.0f) a = kPi - a;
    return y < 0.0f ? -a : a;
}

// ---- PCG32 (src/core/rng.rs) ----
constexpr uint64_t kPcgMult = 0x5851f42d4c957f2dULL;
constexpr uint64_t kPcgDefaultState = 0x853c49e6748fea9bULL;
struct Rng {
    uint64_t state, inc;
};
PB_DEV uint32_t rng_u32(Rng& r) {
    uint64_t old = r.state;
    r.state = old * kPcgMult + r.inc;
    uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
}
PB_DEV void rng_set_sequence(Rng& r, uint64_t seq) {
    r.state = 0;
    r.inc = (seq << 1) | 1;
    rng_u32(r);
    r.state += kPcgDefaultState;
    rng_u32(r);
}
PB_DEV float rng_float(Rng& r) { return fminr(kOneMinusEpsilon, (float)rng_u32(r) * 2.3283064365386963e-10f); }

// ---- sampling ----
PB_DEV void concentric_sample_disk(float ux, float uy, float* dx, float* dy) {
    float ox = ux * 2.0f - 1.0f, oy = uy * 2.0f - 1.0f;
    if (ox == 0.0f && oy == 0.0f) {
        *dx = 0.0f;
        *dy = 0.0f;
        return;
    }
    float r, theta;
    if (__builtin_fabsf(ox) > __builtin_fabsf(oy)) {
        r = ox;
        theta = kPiOver4 * (oy / ox);
    } else {
        r = oy;
        theta = kPiOver2 - kPiOver4 * (ox / oy);
    }
    float s, c;
    det_sincos(theta, &s, &c);
    *dx = c * r;
    *dy = s * r;
}
PB_DEV V3 cosine_sample_hemisphere(float ux, float uy) {
    float dx, dy;
    concentric_sample_disk(ux, uy, &dx, &dy);
    float z = __builtin_sqrtf(fmaxr(1.0f - dx * dx - dy * dy, 0.0f));
    return V3{dx, dy, z};
}
PB_DEV float power_heuristic1(float f_pdf, float g_pdf) {
    float f = 1.0f * f_pdf, g = 1.0f * g_pdf;
    return (f * f) / (f * f + g * g);
}

// ---- Fresnel / refraction ----
PB_DEV float fr_dielectric(float cos_theta_i, float eta_i, float eta_t) {
    cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
    bool entering = cos_theta_i > 0.0f;
    if (!entering) {
        float tmp = eta_i;
        eta_i = eta_t;
        eta_t = tmp;
        cos_theta_i = __builtin_fabsf(cos_theta_i);
    }
    float sin_theta_i = __builtin_sqrtf(fmaxr(1.0f - cos_theta_i * cos_theta_i, 0.0f));
    float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    float cos_theta_t = __builtin_sqrtf(fmaxr(1.0f - sin_theta_t * sin_theta_t, 0.0f));
    float r_parl = (eta_t * cos_theta_i - eta_i * cos_theta_t) / (eta_t * cos_theta_i + eta_i * cos_theta_t);
    float r_perp = (eta_i * cos_theta_i - eta_t * cos_theta_t) / (eta_i * cos_theta_i + eta_t * cos_theta_t);
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
PB_DEV bool refract(V3 wi, V3 n, float eta, V3* wt) {
    float cos_theta_i = dot(n, wi);
    float sin2_theta_i = fmaxr(1.0f - cos_theta_i * cos_theta_i, 0.0f);
    float sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t >= 1.0f) return false;
    float cos_theta_t = __builtin_sqrtf(1.0f - sin2_theta_t);
    *wt = (-wi) * eta + n * (eta * cos_theta_i - cos_theta_t);
    return true;
}

}  // namespace pb
