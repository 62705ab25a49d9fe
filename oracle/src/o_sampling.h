// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header / oracle/README.md).
//
// o_sampling.h — PCG32 RNG, RandomSampler (per-(pixel,sample) streams), sampling routines,
// Distribution1D / Distribution2D.
//
// Follows:
//   src/core/rng.rs:5-98              RNG (PCG32): set_sequence, uniform_u32, uniform_float
//   src/core/sampler.rs:15-33         Sampler::get_1d/get_2d/get_camera_sample (2D, 1D, 2D)
//   src/samplers/random.rs:12-56      RandomSampler
//   src/core/sampler.rs:252-318       PixelSampler (per-pixel tables for the first n dimensions, RNG beyond)
//   src/samplers/stratified.rs:13-112 StratifiedSampler::start_pixel; src/core/sampling.rs:11-66, 280-287
//   src/samplers/halton.rs:24-155     HaltonSampler; src/core/sampler.rs:320-400 GlobalSampler;
//                                     src/core/lowdiscrepancy.rs:293-390 radical inverses and their permutations
//   src/samplers/zerotwosequence.rs   ZeroTwoSequenceSampler; src/core/lowdiscrepancy.rs:416-505 (gray-code
//                                     van der Corput / Sobol' (0,2) points, scrambles, shuffles)
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
#include <algorithm>
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
    // rng.rs:37-45
    uint32_t uniform_u32_bounded(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        for (;;) {
            uint32_t r = uniform_u32();
            if (r >= threshold) return r % b;
        }
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
// sampling.rs:11-17
inline void stratified_sample_1d(Float* samples, int n_samples, RNG& rng, bool jitter) {
    Float inv_n_samples = 1.0f / (Float)n_samples;
    for (int i = 0; i < n_samples; ++i) {
        Float delta = jitter ? rng.uniform_float() : 0.5f;
        samples[i] = fminr(ONE_MINUS_EPSILON, ((Float)i + delta) * inv_n_samples);
    }
}
// sampling.rs:19-41
inline void stratified_sample_2d(Point2f* samples, int nx, int ny, RNG& rng, bool jitter) {
    Float dx = 1.0f / (Float)nx, dy = 1.0f / (Float)ny;
    int i = 0;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            Float jx = 0.5f, jy = 0.5f;
            if (jitter) {
                jx = rng.uniform_float();
                jy = rng.uniform_float();
            }
            samples[i].x = fminr(ONE_MINUS_EPSILON, ((Float)x + jx) * dx);
            samples[i].y = fminr(ONE_MINUS_EPSILON, ((Float)y + jy) * dy);
            ++i;
        }
}
// sampling.rs:280-287: blocks of n_dimensions values are swapped
template <class T>
inline void shuffle(T* samp, int count, int n_dimensions, RNG& rng) {
    for (int i = 0; i < count; ++i) {
        int other = i + (int)rng.uniform_u32_bounded((uint32_t)(count - i));
        for (int j = 0; j < n_dimensions; ++j) std::swap(samp[n_dimensions * i + j], samp[n_dimensions * other + j]);
    }
}
// sampling.rs:44-66 for Point2f samples (n_dim = 2)
inline void latin_hyper_cube_2d(Point2f* samples, int n_samples, RNG& rng) {
    Float inv_n_samples = 1.0f / (Float)n_samples;
    for (int i = 0; i < n_samples; ++i)
        for (int j = 0; j < 2; ++j) {
            Float sj = ((Float)i + rng.uniform_float()) * inv_n_samples;
            (j == 0 ? samples[i].x : samples[i].y) = fminr(ONE_MINUS_EPSILON, sj);
        }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < n_samples; ++j) {
            int other = j + (int)rng.uniform_u32_bounded((uint32_t)(n_samples - j));
            if (i == 0)
                std::swap(samples[j].x, samples[other].x);
            else
                std::swap(samples[j].y, samples[other].y);
        }
}
// lowdiscrepancy.rs:416-434 gray_code_sample / gray_code_sample_2d, :436-505 van_der_corput / sobol_2d.
// D54 (intended): van_der_corput shuffles the block at `i * n_pixel_samples` (lowdiscrepancy.rs:452), which is
// out of range for i >= 1 when one sample per pixel sample is asked for; pbrt-v3 (and sobol_2d, :496) use
// `i * n_samples_per_pixel_sample`.
inline int count_trailing_zeros(uint32_t v) { return __builtin_ctz(v); }
static const uint32_t C_SOBOL_1[32] = {
    0x80000000u, 0xc0000000u, 0xa0000000u, 0xf0000000u, 0x88000000u, 0xcc000000u, 0xaa000000u, 0xff000000u,
    0x80800000u, 0xc0c00000u, 0xa0a00000u, 0xf0f00000u, 0x88880000u, 0xcccc0000u, 0xaaaa0000u, 0xffff0000u,
    0x80008000u, 0xc000c000u, 0xa000a000u, 0xf000f000u, 0x88008800u, 0xcc00cc00u, 0xaa00aa00u, 0xff00ff00u,
    0x80808080u, 0xc0c0c0c0u, 0xa0a0a0a0u, 0xf0f0f0f0u, 0x88888888u, 0xccccccccu, 0xaaaaaaaau, 0xffffffffu};
inline uint32_t c_van_der_corput(int i) { return 0x80000000u >> i; }  // the identity generator matrix
inline void van_der_corput(int n_per_pixel_sample, int n_pixel_samples, Float* samples, RNG& rng) {
    uint32_t scramble = rng.uniform_u32();
    int total = n_per_pixel_sample * n_pixel_samples;
    uint32_t v = scramble;
    for (int i = 0; i < total; ++i) {
        samples[i] = fminr(ONE_MINUS_EPSILON, (Float)v * 2.3283064365386963e-10f);
        v ^= c_van_der_corput(count_trailing_zeros((uint32_t)i + 1u));
    }
    for (int i = 0; i < n_pixel_samples; ++i) shuffle(samples + i * n_per_pixel_sample, n_per_pixel_sample, 1, rng);
    shuffle(samples, n_pixel_samples, n_per_pixel_sample, rng);
}
inline void sobol_2d(int n_per_pixel_sample, int n_pixel_samples, Point2f* samples, RNG& rng) {
    uint32_t s0 = rng.uniform_u32(), s1 = rng.uniform_u32();
    int total = n_per_pixel_sample * n_pixel_samples;
    uint32_t v0 = s0, v1 = s1;
    for (int i = 0; i < total; ++i) {
        samples[i].x = fminr(ONE_MINUS_EPSILON, (Float)v0 * 2.3283064365386963e-10f);
        samples[i].y = fminr(ONE_MINUS_EPSILON, (Float)v1 * 2.3283064365386963e-10f);
        int tz = count_trailing_zeros((uint32_t)i + 1u);
        v0 ^= c_van_der_corput(tz);
        v1 ^= C_SOBOL_1[tz];
    }
    for (int i = 0; i < n_pixel_samples; ++i) shuffle(samples + i * n_per_pixel_sample, n_per_pixel_sample, 1, rng);
    shuffle(samples, n_pixel_samples, n_per_pixel_sample, rng);
}
// pbrt.rs:185-194
inline int64_t round_up_pow2(int64_t v) {
    v -= 1;
    v |= v >> 1;
    v |= v >> 2;
    v |= v >> 4;
    v |= v >> 8;
    v |= v >> 16;
    v |= v >> 32;
    return v + 1;
}

// ---- Halton points: lowdiscrepancy.rs:11-170 (prime tables), :293-390 ----
// The first PRIME_TABLE_SIZE + 23 primes and their running sums are generated instead of tabulated.
static const int PRIME_TABLE_SIZE = 1000;
struct PrimeTables {
    std::vector<uint32_t> primes, prime_sums;  // primes[1023], prime_sums[i] = sum of primes[0..i)
    std::vector<uint16_t> perms;               // compute_radical_inverse_permutations (lowdiscrepancy.rs:333-349)
    PrimeTables() {
        for (uint32_t v = 2; primes.size() < (size_t)PRIME_TABLE_SIZE + 23; ++v) {
            bool is_prime = true;
            for (uint32_t q : primes) {
                if (q * q > v) break;
                if (v % q == 0) {
                    is_prime = false;
                    break;
                }
            }
            if (is_prime) primes.push_back(v);
        }
        prime_sums.resize(PRIME_TABLE_SIZE);
        uint32_t acc = 0;
        for (int i = 0; i < PRIME_TABLE_SIZE; ++i) {
            prime_sums[i] = acc;
            acc += primes[i];
        }
        perms.resize(acc);
        RNG rng;  // RNG::default (halton.rs:17-22)
        uint16_t* p = perms.data();
        for (int i = 0; i < PRIME_TABLE_SIZE; ++i) {
            for (uint32_t j = 0; j < primes[i]; ++j) p[j] = (uint16_t)j;
            shuffle(p, (int)primes[i], 1, rng);
            p += primes[i];
        }
    }
};
inline const PrimeTables& prime_tables() {
    static const PrimeTables t;
    return t;
}
inline uint32_t reverse_bits32(uint32_t n) {  // lowdiscrepancy.rs:371-378
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
    return n;
}
inline uint64_t reverse_bits64(uint64_t n) {  // :364-369
    uint64_t n0 = reverse_bits32((uint32_t)n), n1 = reverse_bits32((uint32_t)(n >> 32));
    return (n0 << 32) | n1;
}
// D53 (intended): radical_inverse_specialized never accumulates `reversed_digits` and squares inv_base
// (lowdiscrepancy.rs:293-305: `let _reversed_digits = ...; inv_base *= inv_base`), so it returns 0 for every
// index -> pbrt-v3 lowdiscrepancy.cpp RadicalInverseSpecialized.
inline Float radical_inverse_specialized(uint64_t base, uint64_t a) {
    Float inv_base = 1.0f / (Float)base;
    uint64_t reversed_digits = 0;
    Float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversed_digits = reversed_digits * base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    return fminr(ONE_MINUS_EPSILON, (Float)reversed_digits * inv_base_n);
}
// D55 (intended): the closing expression multiplies by inv_base instead of inv_base_n
// (lowdiscrepancy.rs:318-319) -> pbrt-v3 `invBaseN * (reversedDigits + invBase * perm[0] / (1 - invBase))`.
inline Float scrambled_radical_inverse_specialized(uint64_t base, const uint16_t* perm, uint64_t a) {
    Float inv_base = 1.0f / (Float)base;
    uint64_t reversed_digits = 0;
    Float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversed_digits = reversed_digits * base + perm[digit];
        inv_base_n *= inv_base;
        a = next;
    }
    return fminr(ONE_MINUS_EPSILON, inv_base_n * ((Float)reversed_digits + inv_base * (Float)perm[0] / (1.0f - inv_base)));
}
inline Float radical_inverse(int base_index, uint64_t a) {  // :322-331
    if (base_index == 0) return fminr(ONE_MINUS_EPSILON, (Float)reverse_bits64(a) * 5.4210108624275222e-20f);
    return radical_inverse_specialized(prime_tables().primes[base_index], a);
}
inline uint64_t inverse_radical_inverse(uint64_t base, uint64_t inverse, int n_digits) {  // :381-390
    uint64_t index = 0;
    for (int i = 0; i < n_digits; ++i) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}
// halton.rs:40-61
inline void extended_gcd(uint64_t a, uint64_t b, int64_t* x, int64_t* y) {
    if (b == 0) {
        *x = 1;
        *y = 0;
        return;
    }
    int64_t d = (int64_t)(a / b), xp = 0, yp = 0;
    extended_gcd(b, a % b, &xp, &yp);
    *x = yp;
    *y = xp - d * yp;
}
inline uint64_t multiplicative_inverse(int64_t a, int64_t n) {
    int64_t x = 0, y = 0;
    extended_gcd((uint64_t)a, (uint64_t)n, &x, &y);
    int64_t r = x - (x / n) * n;
    return (uint64_t)(r < 0 ? r + n : r);
}
// HaltonSampler::new (halton.rs:63-98): the constants derived from the film's sample bounds
struct HaltonSetup {
    int base_scales[2] = {1, 1}, base_exponents[2] = {0, 0};
    int64_t sample_stride = 1;
    uint64_t mult_inverse[2] = {0, 0};
    static constexpr int MAX_RESOLUTION = 128;
    void init(int res_x, int res_y) {
        const int res[2] = {res_x, res_y};
        for (int i = 0; i < 2; ++i) {
            int base = i == 0 ? 2 : 3, scale = 1, exp = 0;
            while (scale < std::min(MAX_RESOLUTION, res[i])) {
                scale *= base;
                ++exp;
            }
            base_scales[i] = scale;
            base_exponents[i] = exp;
        }
        sample_stride = (int64_t)base_scales[0] * base_scales[1];
        mult_inverse[0] = multiplicative_inverse(base_scales[1], base_scales[0]);
        mult_inverse[1] = multiplicative_inverse(base_scales[0], base_scales[1]);
    }
    // get_index_for_sample (halton.rs:118-142). D56 (intended): `current_pixel % MAX_RESOLUTION` is negative for
    // the pixels a wide filter adds left of / above the film -> pbrt-v3's Mod (non-negative remainder).
    int64_t offset_for_pixel(int px, int py) const {
        int64_t offset = 0;
        if (sample_stride > 1) {
            const int pm[2] = {((px % MAX_RESOLUTION) + MAX_RESOLUTION) % MAX_RESOLUTION,
                               ((py % MAX_RESOLUTION) + MAX_RESOLUTION) % MAX_RESOLUTION};
            for (int i = 0; i < 2; ++i) {
                uint64_t dim_offset = inverse_radical_inverse(i == 0 ? 2 : 3, (uint64_t)pm[i], base_exponents[i]);
                offset += (int64_t)(dim_offset * (uint64_t)(sample_stride / base_scales[i]) * mult_inverse[i]);
            }
            offset %= sample_stride;
        }
        return offset;
    }
    // sample_dimension (halton.rs:144-155), sample_at_pixel_center = false
    Float sample_dimension(int64_t index, int dim) const {
        const PrimeTables& t = prime_tables();
        // PRIME_TABLE_SIZE dimensions exist; halton.rs:100-108 only logs past that (and then indexes out of range):
        // later draws reuse the last dimension
        dim = std::min(dim, PRIME_TABLE_SIZE - 1);
        if (dim == 0) return radical_inverse(0, (uint64_t)index >> base_exponents[0]);
        if (dim == 1) return radical_inverse(1, (uint64_t)index / (uint64_t)base_scales[1]);
        return scrambled_radical_inverse_specialized(t.primes[dim], t.perms.data() + t.prime_sums[dim], (uint64_t)index);
    }
};

enum SamplerKind { SAMPLER_RANDOM = 0, SAMPLER_STRATIFIED = 1, SAMPLER_ZEROTWO = 2, SAMPLER_HALTON = 3 };
// What the sampler constructors and the integrators' request_2d_array calls fix before rendering starts.
struct SamplerSpec {
    int kind = SAMPLER_RANDOM;
    int nx = 1, ny = 1;       // StratifiedSampler::new x_pixel_samples, y_pixel_samples
    bool jitter = true;
    int n_dims = 4;           // n_sampled_dimensions
    std::vector<int> arrays_2d;  // Sampler::request_2d_array sizes in request order (sampler.rs:41-46)
    HaltonSetup halton;          // HaltonSampler::new over the film's sample bounds
    // samples per pixel the sampler actually takes (stratified.rs:30-33, zerotwosequence.rs:20)
    int64_t samples_per_pixel(int64_t requested) const {
        if (kind == SAMPLER_STRATIFIED) return (int64_t)nx * ny;
        if (kind == SAMPLER_ZEROTWO) return round_up_pow2(requested);
        return requested;
    }
    // Sampler::round_count (sampler.rs:48-50, zerotwosequence.rs:62-64)
    int round_count(int n) const { return kind == SAMPLER_ZEROTWO ? (int)round_up_pow2(n) : n; }
};

// RandomSampler, and PixelSampler (sampler.rs:252-318) with the StratifiedSampler / ZeroTwoSequenceSampler
// start_pixel tables. Streams (the reference consumes one tile stream serially, which no parallel renderer can
// reproduce): start_pixel draws from the pixel's own stream, seed ^ (2^62 | pixel_index); the draws past the
// tabulated dimensions come from the (pixel, sample) stream of the random sampler.
struct Sampler {
    RNG rng;
    int64_t samples_per_pixel;
    uint64_t n_draws = 0;  // instrumentation
    const SamplerSpec* spec = nullptr;
    std::vector<std::vector<Float>> samples_1d;      // [dimension][pixel sample]
    std::vector<std::vector<Point2f>> samples_2d;
    std::vector<std::vector<Point2f>> sample_array_2d;  // [array][pixel sample * n + k]
    int current_1d_dimension = 0, current_2d_dimension = 0, array_2d_offset = 0;
    int64_t current_pixel_sample_index = 0;
    Sampler() : samples_per_pixel(1) {}
    bool tabulated() const { return spec && spec->kind != SAMPLER_RANDOM; }  // hands out requested sample arrays
    bool halton() const { return spec && spec->kind == SAMPLER_HALTON; }
    // GlobalSampler state (sampler.rs:320-400)
    static const int ARRAY_START_DIM = 5;
    int dimension = 0, array_end_dim = 0;
    int64_t interval_sample_index = 0, offset_for_current_pixel = 0;
    std::vector<std::vector<Point2f>> halton_arrays;  // arrays handed out for the current pixel sample, generated on demand
    int64_t index_for_sample(int64_t sample_num) const { return offset_for_current_pixel + sample_num * spec->halton.sample_stride; }
    // StratifiedSampler::start_pixel (stratified.rs:44-104), ZeroTwoSequenceSampler::start_pixel
    // (zerotwosequence.rs:28-60); 1D arrays are never requested on this path.
    void start_pixel(uint64_t seed, int64_t pixel_index, int64_t spp, int px = 0, int py = 0) {
        samples_per_pixel = spp;
        if (!tabulated()) return;
        if (halton()) {  // GlobalSampler::start_pixel (sampler.rs:341-365); arrays are evaluated when asked for
            offset_for_current_pixel = spec->halton.offset_for_pixel(px, py);
            array_end_dim = ARRAY_START_DIM + 2 * (int)spec->arrays_2d.size();
            return;
        }
        RNG prng;
        prng.set_sequence(seed ^ (0x4000000000000000ULL | (uint64_t)pixel_index));
        const int n = (int)spp;
        samples_1d.assign(spec->n_dims, std::vector<Float>(n));
        samples_2d.assign(spec->n_dims, std::vector<Point2f>(n));
        sample_array_2d.resize(spec->arrays_2d.size());
        for (size_t i = 0; i < sample_array_2d.size(); ++i) sample_array_2d[i].assign((size_t)spec->arrays_2d[i] * n, Point2f());
        if (spec->kind == SAMPLER_STRATIFIED) {
            for (auto& d : samples_1d) {
                stratified_sample_1d(d.data(), n, prng, spec->jitter);
                shuffle(d.data(), n, 1, prng);
            }
            for (auto& d : samples_2d) {
                stratified_sample_2d(d.data(), spec->nx, spec->ny, prng, spec->jitter);
                shuffle(d.data(), n, 1, prng);
            }
            for (size_t i = 0; i < sample_array_2d.size(); ++i)
                for (int j = 0; j < n; ++j) {
                    int count = spec->arrays_2d[i];
                    latin_hyper_cube_2d(sample_array_2d[i].data() + (size_t)j * count, count, prng);
                }
        } else {
            for (auto& d : samples_1d) van_der_corput(1, n, d.data(), prng);
            for (auto& d : samples_2d) sobol_2d(1, n, d.data(), prng);
            for (size_t i = 0; i < sample_array_2d.size(); ++i) sobol_2d(spec->arrays_2d[i], n, sample_array_2d[i].data(), prng);
        }
    }
    void start_sample(uint64_t seed, int64_t pixel_index, int64_t spp, int64_t s) {
        samples_per_pixel = spp;
        rng.set_sequence(seed ^ (uint64_t)(pixel_index * spp + s));
        current_1d_dimension = current_2d_dimension = array_2d_offset = 0;
        current_pixel_sample_index = s;
        dimension = 0;  // GlobalSampler::set_sample_number (sampler.rs:394-398)
        if (halton()) interval_sample_index = index_for_sample(s);
    }
    // sampler.rs:284-292
    Float get_1d() {
        ++n_draws;
        if (halton()) {  // sampler.rs:367-374
            if (dimension >= ARRAY_START_DIM && dimension < array_end_dim) dimension = array_end_dim;
            return spec->halton.sample_dimension(interval_sample_index, dimension++);
        }
        if (tabulated() && current_1d_dimension < spec->n_dims)
            return samples_1d[current_1d_dimension++][current_pixel_sample_index];
        return rng.uniform_float();
    }
    // sampler.rs:294-302
    Point2f get_2d() {
        if (halton()) {  // sampler.rs:376-386
            n_draws += 2;
            if (dimension + 1 >= ARRAY_START_DIM && dimension < array_end_dim) dimension = array_end_dim;
            Float a = spec->halton.sample_dimension(interval_sample_index, dimension);
            Float b = spec->halton.sample_dimension(interval_sample_index, dimension + 1);
            dimension += 2;
            return Point2f(a, b);
        }
        if (tabulated() && current_2d_dimension < spec->n_dims) {
            n_draws += 2;
            return samples_2d[current_2d_dimension++][current_pixel_sample_index];
        }
        n_draws += 2;
        Float a = rng.uniform_float();
        Float b = rng.uniform_float();
        return Point2f(a, b);
    }
    // sampler.rs:64-75: the next requested array, or null when all have been handed out. The random sampler
    // keeps its on-demand draws (its arrays are plain uniform numbers, random.rs:29-42), so it returns null.
    const Point2f* get_2d_array(int n) {
        if (halton()) {
            // GlobalSampler::start_pixel fills array i, element j with dimensions (5 + 2i, 6 + 2i) of the pixel's
            // j-th sample index (sampler.rs:354-364); get_2d_array returns the slice of the current pixel sample
            if (array_2d_offset == (int)spec->arrays_2d.size()) return nullptr;
            if (halton_arrays.size() < spec->arrays_2d.size()) halton_arrays.resize(spec->arrays_2d.size());
            std::vector<Point2f>& out = halton_arrays[array_2d_offset];
            int dim = ARRAY_START_DIM + 2 * array_2d_offset++;
            out.resize(n);
            for (int k = 0; k < n; ++k) {
                int64_t idx = index_for_sample(current_pixel_sample_index * n + k);
                out[k] = Point2f(spec->halton.sample_dimension(idx, dim), spec->halton.sample_dimension(idx, dim + 1));
            }
            return out.data();
        }
        if (!tabulated() || array_2d_offset == (int)sample_array_2d.size()) return nullptr;
        const std::vector<Point2f>& a = sample_array_2d[array_2d_offset++];
        return a.data() + (size_t)current_pixel_sample_index * n;
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
