// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header / oracle/README.md).
//
// o_sampling.h — PCG32 RNG, RandomSampler (per-(pixel,sample) streams), sampling routines,
// Distribution1D / Distribution2D.
//
// Follows:
//   src/core/rng.rs:5-98              RNG (PCG32): set_sequence, uniform_u32, uniform_float
//   src/core/sampler.rs:15-33         Sampler::get_1d/get_2d/get_camera_sample (2D, 1D, 2D)
//   src/samplers/random.rs:12-56      RandomSampler
//   src/core/sampling.rs:62-154       Distribution1D (find_interval predicate `cdf[i] < u` as written)
//   src/core/sampling.rs:168-213      Distribution2D
//   src/core/sampling.rs:258-313      concentric_sample_disk, uniform_sample_triangle,
//                                     cosine_sample_hemisphere, power_heuristic
//   src/core/pbrt.rs:229-243          find_interval
// Defect dispositions: D39 [Q] (cosine z lacks sqrt), D40 (Distribution1D::new indexes an empty
// Vec; Distribution2D rows sliced with v*nv) — intended; D8 (find_interval usize underflow) —
// intended (signed clamp). Distribution2D::pdf casts before multiplying (sampling.rs:201-210) —
// intended `(p*count) as int`.
#pragma once
#include <vector>

#include "o_math.h"

namespace oracle {

// src/core/rng.rs
struct RNG {
    uint64_t state, inc;
    static constexpr uint64_t PCG32_DEFAULT_STATE = 0x853c49e6748fea9bULL;
    static constexpr uint64_t PCG32_DEFAULT_STREAM = 0xda3e39cb94b95bdbULL;
    static constexpr uint64_t PCG32_MULT = 0x5851f42d4c957f2dULL;
    RNG() : state(PCG32_DEFAULT_STATE), inc(PCG32_DEFAULT_STREAM) {}
    explicit RNG(uint64_t seq) { set_sequence(seq); }
    // rng.rs:21-27
    void set_sequence(uint64_t sequence_index) {
        state = 0;
        inc = (sequence_index << 1) | 1;
        uniform_u32();
        state += PCG32_DEFAULT_STATE;
        uniform_u32();
    }
    // rng.rs:29-35
    uint32_t uniform_u32() {
        uint64_t old_state = state;
        state = old_state * PCG32_MULT + inc;
        uint32_t xor_shifted = (uint32_t)(((old_state >> 18) ^ old_state) >> 27);
        uint32_t rot = (uint32_t)(old_state >> 59);
        return (xor_shifted >> rot) | (xor_shifted << ((~rot + 1) & 31));
    }
    // rng.rs:46-48
    Float uniform_float() { return fminr(ONE_MINUS_EPSILON, (Float)uniform_u32() * 2.3283064365386963e-10f); }
};

// The reference seeds one stream per 16x16 tile and consumes it serially across pixels
// (integrator.rs:414-415). That is inherently sequential; oracle and kernels key one PCG32
// stream per (pixel, sample) instead (SURVEY.md §7 "Random-number parity"):
//   sequence = seed ^ ((y*W + x)*spp + s)
// The Sampler trait surface (get_1d / get_2d / get_camera_sample) is unchanged.
struct CameraSample {
    Point2f p_film, p_lens;
    Float time;
};
struct Sampler {
    RNG rng;
    int64_t samples_per_pixel;
    uint64_t n_draws = 0;  // instrumentation
    Sampler() : samples_per_pixel(1) {}
    void start_sample(uint64_t seed, int64_t pixel_index, int64_t spp, int64_t s) {
        samples_per_pixel = spp;
        rng.set_sequence(seed ^ (uint64_t)(pixel_index * spp + s));
    }
    Float get_1d() {
        ++n_draws;
        return rng.uniform_float();
    }
    Point2f get_2d() {
        Float a = get_1d();
        Float b = get_1d();
        return Point2f(a, b);
    }
    // sampler.rs:27-33
    CameraSample get_camera_sample(int px, int py) {
        CameraSample cs;
        Point2f u = get_2d();
        cs.p_film = Point2f((Float)px + u.x, (Float)py + u.y);
        cs.time = get_1d();
        cs.p_lens = get_2d();
        return cs;
    }
};

// src/core/pbrt.rs:229-243
template <class Pred>
inline int find_interval(int size, Pred pred) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (pred(middle)) {
            first = middle + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    int v = first - 1;
    return v < 0 ? 0 : (v > size - 2 ? size - 2 : v);
}

// src/core/sampling.rs:62-154
struct Distribution1D {
    std::vector<Float> func, cdf;
    Float func_int = 0;
    Distribution1D() {}
    Distribution1D(const Float* f, int n) : func(f, f + n), cdf(n + 1) {
        cdf[0] = 0.0f;
        for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (Float)n;
        func_int = cdf[n];
        if (func_int == 0.0f) {
            for (int i = 1; i < n + 1; ++i) cdf[i] = (Float)i / (Float)n;
        } else {
            for (int i = 1; i < n + 1; ++i) cdf[i] /= func_int;
        }
    }
    int count() const { return (int)func.size(); }
    Float sample_continuous(Float u, Float* pdf, int* off) const {
        int offset = find_interval((int)cdf.size(), [&](int i) { return cdf[i] < u; });
        if (off) *off = offset;
        Float du = u - cdf[offset];
        if (cdf[offset + 1] - cdf[offset] > 0.0f) du /= cdf[offset + 1] - cdf[offset];
        if (pdf) *pdf = func_int > 0.0f ? func[offset] / func_int : 0.0f;
        return ((Float)offset + du) / (Float)count();
    }
    int sample_discrete(Float u, Float* pdf) const {
        int offset = find_interval((int)cdf.size(), [&](int i) { return cdf[i] < u; });
        if (pdf) *pdf = func_int > 0.0f ? func[offset] / (func_int * (Float)count()) : 0.0f;
        return offset;
    }
    Float discrete_pdf(int index) const { return func[index] / (func_int * (Float)count()); }
};

// src/core/sampling.rs:168-213
struct Distribution2D {
    std::vector<Distribution1D> conditional;
    Distribution1D marginal;
    Distribution2D() {}
    Distribution2D(const Float* data, int nu, int nv) {
        for (int v = 0; v < nv; ++v) conditional.emplace_back(data + v * nu, nu);
        std::vector<Float> mf;
        for (int v = 0; v < nv; ++v) mf.push_back(conditional[v].func_int);
        marginal = Distribution1D(mf.data(), nv);
    }
    Point2f sample_continuous(const Point2f& u, Float* pdf) const {
        Float pdfs[2];
        int v;
        Float d1 = marginal.sample_continuous(u.y, &pdfs[1], &v);
        Float d0 = conditional[v].sample_continuous(u.x, &pdfs[0], nullptr);
        if (pdf) *pdf = pdfs[0] * pdfs[1];
        return Point2f(d0, d1);
    }
    Float pdf(const Point2f& p) const {
        int nu = conditional[0].count(), nv = marginal.count();
        int iu = (int)(p.x * (Float)nu);
        iu = iu < 0 ? 0 : (iu > nu - 1 ? nu - 1 : iu);
        int iv = (int)(p.y * (Float)nv);
        iv = iv < 0 ? 0 : (iv > nv - 1 ? nv - 1 : iv);
        return conditional[iv].func[iu] / marginal.func_int;
    }
};

// src/core/sampling.rs:258-273 (theta.cos()/theta.sin() -> deterministic sincos)
inline Point2f concentric_sample_disk(const Point2f& u) {
    Float ox = u.x * 2.0f - 1.0f, oy = u.y * 2.0f - 1.0f;
    if (ox == 0.0f && oy == 0.0f) return Point2f(0.0f, 0.0f);
    Float r, theta;
    if (std::fabs(ox) > std::fabs(oy)) {
        r = ox;
        theta = PI_OVER_4 * (oy / ox);
    } else {
        r = oy;
        theta = PI_OVER_2 - PI_OVER_4 * (ox / oy);
    }
    Float s, c;
    det_sincos(theta, &s, &c);
    return Point2f(c * r, s * r);
}
// src/core/sampling.rs:275-278
inline Point2f uniform_sample_triangle(const Point2f& u) {
    Float su0 = std::sqrt(u.x);
    return Point2f(1.0f - su0, u.y * su0);
}
// src/core/sampling.rs:290-294 — D39 [Q]: as written z = max(0, 1 - x^2 - y^2) without sqrt.
inline Vector3f cosine_sample_hemisphere(const Point2f& u, uint32_t quirks = 0) {
    Point2f d = concentric_sample_disk(u);
    Float z = fmaxr(1.0f - d.x * d.x - d.y * d.y, 0.0f);
    if (!(quirks & Q_D39_COSINE_Z)) z = std::sqrt(z);
    return Vector3f(d.x, d.y, z);
}
inline Float cosine_hemisphere_pdf(Float cos_theta) { return cos_theta * INV_PI; }
// src/core/sampling.rs:307-313
inline Float power_heuristic(int nf, Float f_pdf, int ng, Float g_pdf) {
    Float f = (Float)nf * f_pdf, g = (Float)ng * g_pdf;
    return (f * f) / (f * f + g * g);
}

}  // namespace oracle
