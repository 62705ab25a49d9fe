// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// CPU restatement of the pbrt-rs hot path (parity unpinned: the reference holds no
// golden vectors for this path and cannot be built or run here; see oracle/README.md).
//
// o_math.h — Float, constants, gamma, next_float_*, Vector3, Bounds3, Ray,
// offset_ray_origin, and the deterministic trig used by both sides of the parity check.
//
// Follows (reference file:line, relative to /root/reference):
//   src/core/pbrt.rs:16-28      Float = f32 and the constants
//   src/core/pbrt.rs:43-77      next_float_up / next_float_down
//   src/core/pbrt.rs:89-91      gamma(n)
//   src/core/mod.rs:206         machine_epsilon = f32::EPSILON * 0.5
//   src/core/geometry.rs:62-314 Vector ops (operator* of two vectors is the dot product)
//   src/core/geometry.rs:358-393 cross / coordinate_system / face_forward
//   src/core/geometry.rs:430-588 Bounds3 (Default = {+MAX, -MAX}, union, maximum_extent, offset)
//   src/core/geometry.rs:757-763 Ray
//   src/core/geometry.rs:1139-1154 offset_ray_origin
// Defect dispositions: D1 (RealNum::max calls min) — intended max.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace oracle {

typedef float Float;

// src/core/pbrt.rs:16-28
static const Float PI = 3.14159265358979323846f;
static const Float INV_PI = 1.0f / PI;
static const Float INV_2_PI = INV_PI / 2.0f;
static const Float INV_4_PI = INV_PI / 4.0f;
static const Float PI_OVER_2 = PI / 2.0f;
static const Float PI_OVER_4 = PI / 4.0f;
static const Float SHADOW_EPSILON = 0.0001f;
static const Float EPSILON = std::numeric_limits<float>::epsilon();
static const Float MACHINE_EPSILON = 0.5f * EPSILON;
static const Float ONE_MINUS_EPSILON = 1.0f - EPSILON;
static const Float FLOAT_MAX = std::numeric_limits<float>::max();
static const Float FLOAT_INF = std::numeric_limits<float>::infinity();

// Quirk bits: reproduce a [Q] defect *as written* in the reference.
// Default (0) is the intended pbrt-v3 behaviour.
enum Quirk : uint32_t {
    Q_D2_ZSLAB_SCALE = 1u << 0,    // geometry.rs:738  tz_max *= 1 + 2 + gamma(3)
    Q_D9_SHEAR_SX = 1u << 1,       // triangle.rs:107  p2t.y += sx * p2t.z
    Q_D10_RANGE_PRECEDENCE = 1u << 2,  // triangle.rs:127  (A && B) || C
    Q_D11_DELTA_E = 1u << 3,       // triangle.rs:144-147 delta_y used twice
    Q_D13_DEGENERATE_UV = 1u << 4,  // triangle.rs:203 determinant < 1e-8 (no abs)
    Q_D36_LOCAL_TO_WORLD = 1u << 5,  // reflection.rs:260 ts.y*v.y * ns.y*v.z
    Q_D37_REFRACT = 1u << 6,       // reflection.rs:150 tests sin2_theta_i >= 1
    Q_D39_COSINE_Z = 1u << 7,      // sampling.rs:292 z lacks sqrt
};

inline uint32_t float_to_bits(Float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline Float bits_to_float(uint32_t u) {
    Float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// src/core/pbrt.rs:43-59
inline Float next_float_up(Float n) {
    if (std::isinf(n) && n > 0.0f) return n;
    if (n == -0.0f) n = 0.0f;
    uint32_t u = float_to_bits(n);
    if (n >= 0.0f)
        u = u + 1;
    else
        u = u - 1;
    return bits_to_float(u);
}
// src/core/pbrt.rs:61-77
inline Float next_float_down(Float n) {
    if (std::isinf(n) && n < 0.0f) return n;
    if (n == 0.0f) n = -0.0f;
    uint32_t u = float_to_bits(n);
    if (n > 0.0f)
        u = u - 1;
    else
        u = u + 1;
    return bits_to_float(u);
}

// src/core/pbrt.rs:89-91
inline Float gamma(Float n) { return n * MACHINE_EPSILON / (1.0f - n * MACHINE_EPSILON); }

inline Float clampf(Float v, Float lo, Float hi) { return v < lo ? lo : (v > hi ? hi : v); }
// Rust f32::min/max semantics for non-NaN inputs.
inline Float fminr(Float a, Float b) { return a < b ? a : b; }
inline Float fmaxr(Float a, Float b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------
// Deterministic elementary functions.
// The reference calls Rust's f32::sin/cos/acos/atan2 (platform libm; not correctly
// rounded, so no particular bit pattern is defined by the reference). Both the oracle
// and the HIP kernels evaluate the SAME fixed sequence of IEEE-754 single operations
// (+, *, fma, /, sqrt, round-to-nearest-even) so results agree bit for bit on CPU and
// GPU. Polynomials: Cephes single-precision sinf/cosf/asinf/atanf kernels.
// ---------------------------------------------------------------------------------
inline Float fmaf_(Float a, Float b, Float c) { return __builtin_fmaf(a, b, c); }

inline void det_sincos(Float x, Float* s_out, Float* c_out) {
    // k = nearest integer to x * 2/pi; r = x - k*pi/2 by three-term Cody-Waite.
    Float q = x * 0.63661977236758134308f;
    Float k = __builtin_rintf(q);
    Float r = fmaf_(-k, 1.5707397460937500f, x);
    r = fmaf_(-k, 5.6579709053039550781e-05f, r);
    r = fmaf_(-k, 9.9209362947050294680e-10f, r);
    Float z = r * r;
    // sin(r), |r| <= pi/4
    Float ps = fmaf_(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf_(ps, z, -1.6666654611e-1f);
    Float sr = fmaf_(ps * z, r, r);
    // cos(r)
    Float pc = fmaf_(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf_(pc, z, 4.166664568298827e-2f);
    Float cr = fmaf_(pc * z, z, fmaf_(-0.5f, z, 1.0f));
    int ki = (int)k;
    Float s, c;
    switch (ki & 3) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
    *s_out = s;
    *c_out = c;
}
inline Float det_sin(Float x) {
    Float s, c;
    det_sincos(x, &s, &c);
    return s;
}
inline Float det_cos(Float x) {
    Float s, c;
    det_sincos(x, &s, &c);
    return c;
}
// asin on |x| <= 0.5 (Cephes asinf polynomial)
inline Float det_asin_core(Float x) {
    Float z = x * x;
    Float p = fmaf_(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = fmaf_(p, z, 4.5470025998e-2f);
    p = fmaf_(p, z, 7.4953002686e-2f);
    p = fmaf_(p, z, 1.6666752422e-1f);
    return fmaf_(p * z, x, x);
}
inline Float det_acos(Float x) {
    if (x < -0.5f) {
        Float t = std::sqrt(0.5f * (1.0f + x));
        return PI - 2.0f * det_asin_core(t);
    }
    if (x > 0.5f) {
        Float t = std::sqrt(0.5f * (1.0f - x));
        return 2.0f * det_asin_core(t);
    }
    return PI_OVER_2 - det_asin_core(x);
}
// atan for x >= 0 (Cephes atanf)
inline Float det_atan_pos(Float x) {
    Float y;
    if (x > 2.414213562373095f) {
        y = PI_OVER_2;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = PI_OVER_4;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    Float z = x * x;
    Float p = fmaf_(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf_(p, z, 1.99777106478e-1f);
    p = fmaf_(p, z, -3.33329491539e-1f);
    return y + fmaf_(p * z, x, x);
}
inline Float det_atan2(Float y, Float x) {
    if (x == 0.0f) {
        if (y > 0.0f) return PI_OVER_2;
        if (y < 0.0f) return -PI_OVER_2;
        return 0.0f;
    }
    Float a = det_atan_pos(std::fabs(y / x));
    if (x < 0.0f) a = PI - a;
    return y < 0.0f ? -a : a;
}

// ---------------------------------------------------------------------------------
// Vector3 — src/core/geometry.rs:62-314, 341-393
// ---------------------------------------------------------------------------------
struct Vector3f {
    Float x, y, z;
    Vector3f() : x(0), y(0), z(0) {}
    Vector3f(Float x_, Float y_, Float z_) : x(x_), y(y_), z(z_) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    Vector3f operator+(const Vector3f& r) const { return Vector3f(x + r.x, y + r.y, z + r.z); }
    Vector3f operator-(const Vector3f& r) const { return Vector3f(x - r.x, y - r.y, z - r.z); }
    Vector3f operator*(Float s) const { return Vector3f(x * s, y * s, z * s); }
    Vector3f operator/(Float s) const { return Vector3f(x / s, y / s, z / s); }
    Vector3f operator-() const { return Vector3f(-x, -y, -z); }
    Vector3f& operator+=(const Vector3f& r) {
        x += r.x; y += r.y; z += r.z;
        return *this;
    }
    Vector3f& operator*=(Float s) {
        x *= s; y *= s; z *= s;
        return *this;
    }
    // geometry.rs:228-234 (x*x + y*y + z*z, left to right)
    Float dot(const Vector3f& v) const { return x * v.x + y * v.y + z * v.z; }
    Float abs_dot(const Vector3f& v) const { return std::fabs(dot(v)); }
    Float length_squared() const { return x * x + y * y + z * z; }
    Float length() const { return std::sqrt(length_squared()); }
    Vector3f normalize() const { return *this / length(); }  // geometry.rs:117-119
    Vector3f abs() const { return Vector3f(std::fabs(x), std::fabs(y), std::fabs(z)); }
    // geometry.rs:37-51 make_extent!: x > y && x > z ? 0 : (y > z ? 1 : 2)
    int max_dimension() const { return (x > y && x > z) ? 0 : (y > z ? 1 : 2); }
    // geometry.rs:95-97, D1 disposition: true max. x.max(y.max(z))
    Float max_component() const { return fmaxr(x, fmaxr(y, z)); }
    Vector3f permute(int a, int b, int c) const { return Vector3f((*this)[a], (*this)[b], (*this)[c]); }
    // geometry.rs:359-372 (f32 products, no f64 promotion)
    Vector3f cross(const Vector3f& v) const {
        return Vector3f(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x);
    }
    // geometry.rs:374-382
    void coordinate_system(Vector3f* v2, Vector3f* v3) const {
        if (std::fabs(x) > std::fabs(y))
            *v2 = Vector3f(-z, 0.0f, x).normalize();
        else
            *v2 = Vector3f(0.0f, z, -y).normalize();
        *v3 = cross(*v2);
    }
    // geometry.rs:384-390 `self.face_forward(v)` returns +-v as written; every call site
    // (triangle.rs:340, interaction.rs:311-313, reflection.rs:703,797) needs pbrt-v3's
    // Faceforward(n, v) = +-n. Disposition (D46, found while restating): intended.
    Vector3f face_forward(const Vector3f& v) const { return dot(v) < 0.0f ? -(*this) : *this; }
    Float distance_square(const Vector3f& p) const { return (*this - p).length_squared(); }
    bool has_nans() const { return std::isnan(x) || std::isnan(y) || std::isnan(z); }
};
typedef Vector3f Point3f;
typedef Vector3f Normal3f;

struct Point2f {
    Float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(Float x_, Float y_) : x(x_), y(y_) {}
    Float operator[](int i) const { return i == 0 ? x : y; }
};

// ---------------------------------------------------------------------------------
// Bounds3 — src/core/geometry.rs:430-588, 657-674
// ---------------------------------------------------------------------------------
struct Bounds3f {
    Point3f min, max;
    // geometry.rs:439-448: Default = {min: +MAX, max: -MAX}
    Bounds3f() : min(FLOAT_MAX, FLOAT_MAX, FLOAT_MAX), max(-FLOAT_MAX, -FLOAT_MAX, -FLOAT_MAX) {}
    Bounds3f(const Point3f& a, const Point3f& b)
        : min(fminr(a.x, b.x), fminr(a.y, b.y), fminr(a.z, b.z)),
          max(fmaxr(a.x, b.x), fmaxr(a.y, b.y), fmaxr(a.z, b.z)) {}
    const Point3f& operator[](int i) const { return i == 0 ? min : max; }
    // geometry.rs:523-539 (D1: true max)
    Bounds3f union_(const Bounds3f& b) const {
        Bounds3f r;
        r.min = Point3f(fminr(min.x, b.min.x), fminr(min.y, b.min.y), fminr(min.z, b.min.z));
        r.max = Point3f(fmaxr(max.x, b.max.x), fmaxr(max.y, b.max.y), fmaxr(max.z, b.max.z));
        return r;
    }
    Bounds3f union_(const Point3f& p) const {
        Bounds3f r;
        r.min = Point3f(fminr(min.x, p.x), fminr(min.y, p.y), fminr(min.z, p.z));
        r.max = Point3f(fmaxr(max.x, p.x), fmaxr(max.y, p.y), fmaxr(max.z, p.z));
        return r;
    }
    Vector3f diagonal() const { return max - min; }
    // geometry.rs:667-670
    Float surface_area() const {
        Vector3f d = diagonal();
        return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    int maximum_extent() const {
        Vector3f d = diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        if (d.y > d.z) return 1;
        return 2;
    }
    // pbrt-v3 Bounds3::Offset
    Vector3f offset(const Point3f& p) const {
        Vector3f o = p - min;
        if (max.x > min.x) o.x /= max.x - min.x;
        if (max.y > min.y) o.y /= max.y - min.y;
        if (max.z > min.z) o.z /= max.z - min.z;
        return o;
    }
    void bounding_sphere(Point3f* c, Float* rad) const {
        *c = (min + max) / 2.0f;
        bool inside = c->x >= min.x && c->x <= max.x && c->y >= min.y && c->y <= max.y &&
                      c->z >= min.z && c->z <= max.z;
        *rad = inside ? (*c - max).length() : 0.0f;
    }
};

// src/core/geometry.rs:757-763 (medium dropped: handle_media = false on every call of this path)
struct Ray {
    Point3f o;
    Vector3f d;
    mutable Float t_max;
    Float time;
    Ray() : t_max(FLOAT_INF), time(0) {}
    Ray(const Point3f& o_, const Vector3f& d_, Float t_max_ = FLOAT_INF, Float time_ = 0.0f)
        : o(o_), d(d_), t_max(t_max_), time(time_) {}
    Point3f at(Float t) const { return o + d * t; }
};

// src/core/geometry.rs:1139-1154
inline Point3f offset_ray_origin(const Point3f& p, const Vector3f& p_error, const Normal3f& n,
                                 const Vector3f& w) {
    Float d = n.abs().dot(p_error);
    Vector3f offset = n * d;
    if (w.dot(n) < 0.0f) offset = -offset;
    Point3f po = p + offset;
    for (int i = 0; i < 3; ++i) {
        if (offset[i] > 0.0f)
            po[i] = next_float_up(po[i]);
        else if (offset[i] < 0.0f)
            po[i] = next_float_down(po[i]);
    }
    return po;
}

// src/core/geometry.rs:709-751 — Bounds3f::intersect_p(ray, inv_dir, dir_is_neg).
// D2 [Q]: z slab scale as written is 1 + 2 + gamma(3); intended 1 + 2*gamma(3).
inline bool bounds_intersect_p(const Bounds3f& b, const Ray& ray, const Vector3f& inv_dir,
                               const int dir_is_neg[3], uint32_t quirks = 0) {
    Float t_min = (b[dir_is_neg[0]].x - ray.o.x) * inv_dir.x;
    Float t_max = (b[1 - dir_is_neg[0]].x - ray.o.x) * inv_dir.x;
    Float ty_min = (b[dir_is_neg[1]].y - ray.o.y) * inv_dir.y;
    Float ty_max = (b[1 - dir_is_neg[1]].y - ray.o.y) * inv_dir.y;

    t_max *= 1.0f + 2.0f * gamma(3.0f);
    ty_max *= 1.0f + 2.0f * gamma(3.0f);
    if (t_min > ty_max || ty_min > t_max) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;

    Float tz_min = (b[dir_is_neg[2]].z - ray.o.z) * inv_dir.z;
    Float tz_max = (b[1 - dir_is_neg[2]].z - ray.o.z) * inv_dir.z;
    if (quirks & Q_D2_ZSLAB_SCALE)
        tz_max *= 1.0f + 2.0f + gamma(3.0f);
    else
        tz_max *= 1.0f + 2.0f * gamma(3.0f);
    if (t_min > tz_max || tz_min > t_max) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    return t_min < ray.t_max && t_max > 0.0f;
}

// RGB spectrum — src/core/spectrum.rs:112-385, 653-716 (Spectrum = RGBSpectrum)
struct Spectrum {
    Float c[3];
    Spectrum() { c[0] = c[1] = c[2] = 0.0f; }
    explicit Spectrum(Float v) { c[0] = c[1] = c[2] = v; }
    Spectrum(Float r, Float g, Float b) { c[0] = r; c[1] = g; c[2] = b; }
    Spectrum operator+(const Spectrum& o) const { return Spectrum(c[0] + o.c[0], c[1] + o.c[1], c[2] + o.c[2]); }
    Spectrum operator-(const Spectrum& o) const { return Spectrum(c[0] - o.c[0], c[1] - o.c[1], c[2] - o.c[2]); }
    Spectrum operator*(const Spectrum& o) const { return Spectrum(c[0] * o.c[0], c[1] * o.c[1], c[2] * o.c[2]); }
    Spectrum operator*(Float s) const { return Spectrum(c[0] * s, c[1] * s, c[2] * s); }
    Spectrum operator/(Float s) const { return Spectrum(c[0] / s, c[1] / s, c[2] / s); }
    Spectrum& operator+=(const Spectrum& o) {
        c[0] += o.c[0]; c[1] += o.c[1]; c[2] += o.c[2];
        return *this;
    }
    Spectrum& operator*=(const Spectrum& o) {
        c[0] *= o.c[0]; c[1] *= o.c[1]; c[2] *= o.c[2];
        return *this;
    }
    Spectrum& operator*=(Float s) {
        c[0] *= s; c[1] *= s; c[2] *= s;
        return *this;
    }
    Spectrum& operator/=(Float s) {
        c[0] /= s; c[1] /= s; c[2] /= s;
        return *this;
    }
    // spectrum.rs:176-183, D34 disposition: intended (true iff every channel is zero)
    bool is_black() const { return c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f; }
    bool has_nans() const { return std::isnan(c[0]) || std::isnan(c[1]) || std::isnan(c[2]); }
    // spectrum.rs:679-682
    Float y_value() const { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }
    // spectrum.rs:161-165
    Float max_component_value() const {
        Float m = -FLOAT_MAX;
        for (int i = 0; i < 3; ++i) m = (m > c[i]) ? m : c[i];
        return m;
    }
    // spectrum.rs:103-108
    void to_xyz(Float xyz[3]) const {
        xyz[0] = 0.412453f * c[0] + 0.357580f * c[1] + 0.180423f * c[2];
        xyz[1] = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2];
        xyz[2] = 0.019334f * c[0] + 0.119193f * c[1] + 0.950227f * c[2];
    }
};
// spectrum.rs:96-100
inline void xyz_to_rgb(const Float xyz[3], Float rgb[3]) {
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

}  // namespace oracle
