// wf_sampler.h — Sampler on the device: tabulated PixelSampler dimensions (stratified, (0,2)), Halton, the path's PCG32 stream (part of wavefront.h)
#pragma once
#include "wf_state.h"

namespace pb {

// ---- Sampler (sampler.rs:15-33, PixelSampler :284-302): tabulated dimensions first, then the path's RNG ----
struct Samp {
    Rng rng;
    int pix, s;  // column of the pixel in the tables, pixel sample index
    int dim1, dim2, arr;  // Halton: dim1 = GlobalSampler::dimension
    long long h_offset;   // HaltonSampler::offset_for_current_pixel
};
// ---- Halton points (lowdiscrepancy.rs:293-390; D53 / D55 intended as in the oracle) ----
PB_DEV float halton_radical_inverse(uint32_t base, const uint16_t* perm, unsigned long long a) {
    float inv_base = 1.0f / (float)base;
    unsigned long long reversed = 0;
    float inv_base_n = 1.0f;
    if (a >> 32) {  // rare: 64-bit digits until the rest fits 32 bits
        while (a >> 32) {
            unsigned long long next = a / base;
            uint32_t digit = (uint32_t)(a - next * base);
            reversed = reversed * base + (perm ? perm[digit] : digit);
            inv_base_n *= inv_base;
            a = next;
        }
    }
    uint32_t a32 = (uint32_t)a;
    while (a32 != 0) {
        uint32_t next = a32 / base;
        uint32_t digit = a32 - next * base;
        reversed = reversed * base + (perm ? perm[digit] : digit);
        inv_base_n *= inv_base;
        a32 = next;
    }
    if (!perm) return fminr(kOneMinusEpsilon, (float)reversed * inv_base_n);
    return fminr(kOneMinusEpsilon, inv_base_n * ((float)reversed + inv_base * (float)perm[0] / (1.0f - inv_base)));
}
PB_DEV float halton_dimension(const SamplerParams& sp, long long index, int dim) {  // halton.rs:144-155
    // the tables hold PRIME_TABLE_SIZE = 1000 dimensions (halton.rs:100-108 only logs past that): later draws reuse the last
    dim = dim > 999 ? 999 : dim;
    if (dim == 0) {
        unsigned long long a = (unsigned long long)index >> sp.h_exp[0];
        unsigned long long r = ((unsigned long long)__brev((uint32_t)a) << 32) | (unsigned long long)__brev((uint32_t)(a >> 32));
        return fminr(kOneMinusEpsilon, (float)r * 5.4210108624275222e-20f);
    }
    if (dim == 1) return halton_radical_inverse(3u, nullptr, (unsigned long long)index / (unsigned long long)sp.h_scale[1]);
    return halton_radical_inverse(sp.primes[dim], sp.perms + sp.primes[1000 + dim], (unsigned long long)index);
}
PB_DEV long long halton_pixel_offset(const SamplerParams& sp, int px, int py) {  // halton.rs:118-142 (D56 intended)
    long long offset = 0;
    if (sp.h_stride > 1) {
        int pm[2] = {((px % 128) + 128) % 128, ((py % 128) + 128) % 128};
        for (int i = 0; i < 2; ++i) {
            unsigned int base = i == 0 ? 2u : 3u, inverse = (unsigned int)pm[i], index = 0;
            for (int d = 0; d < sp.h_exp[i]; ++d) {
                unsigned int digit = inverse % base;
                inverse /= base;
                index = index * base + digit;
            }
            offset += (long long)((unsigned long long)index * (unsigned long long)(sp.h_stride / sp.h_scale[i]) * sp.h_minv[i]);
        }
        offset %= sp.h_stride;
    }
    return offset;
}
PB_DEV float samp_1d(const PassParams& pp, Samp& sm) {
    if (pp.smp.kind == PBRT_SAMPLER_HALTON) {  // GlobalSampler::get_1d (sampler.rs:367-374)
        if (sm.dim1 >= 5 && sm.dim1 < pp.smp.array_end_dim) sm.dim1 = pp.smp.array_end_dim;
        float v = halton_dimension(pp.smp, sm.h_offset + (long long)sm.s * pp.smp.h_stride, sm.dim1);
        sm.dim1 = sm.dim1 + 1 > 1000 ? 1000 : sm.dim1 + 1;
        return v;
    }
    if (pp.smp.kind != PBRT_SAMPLER_RANDOM && sm.dim1 < pp.smp.n_dims) {
        int e = sm.dim1 * pp.spp + sm.s;
        sm.dim1 += 1;
        return pp.smp.tables[(size_t)e * pp.n_pix + sm.pix];
    }
    return rng_float(sm.rng);
}
PB_DEV void samp_2d(const PassParams& pp, Samp& sm, float* u0, float* u1) {
    if (pp.smp.kind == PBRT_SAMPLER_HALTON) {  // GlobalSampler::get_2d (sampler.rs:376-386)
        if (sm.dim1 + 1 >= 5 && sm.dim1 < pp.smp.array_end_dim) sm.dim1 = pp.smp.array_end_dim;
        long long index = sm.h_offset + (long long)sm.s * pp.smp.h_stride;
        *u0 = halton_dimension(pp.smp, index, sm.dim1);
        *u1 = halton_dimension(pp.smp, index, sm.dim1 + 1);
        sm.dim1 = sm.dim1 + 2 > 1000 ? 1000 : sm.dim1 + 2;
        return;
    }
    if (pp.smp.kind != PBRT_SAMPLER_RANDOM && sm.dim2 < pp.smp.n_dims) {
        int e = pp.smp.off2 + (sm.dim2 * pp.spp + sm.s) * 2;
        sm.dim2 += 1;
        *u0 = pp.smp.tables[(size_t)e * pp.n_pix + sm.pix];
        *u1 = pp.smp.tables[(size_t)(e + 1) * pp.n_pix + sm.pix];
        return;
    }
    *u0 = rng_float(sm.rng);
    *u1 = rng_float(sm.rng);
}
// element k of requested array a for this pixel sample (Sampler::get_2d_array, sampler.rs:64-75)
PB_DEV void samp_array_2d(const PassParams& pp, const Samp& sm, int a, int k, float* u0, float* u1) {
    int2 ar = pp.smp.arrays[a];
    if (pp.smp.kind == PBRT_SAMPLER_HALTON) {  // GlobalSampler::start_pixel (sampler.rs:354-364), evaluated on demand
        long long index = sm.h_offset + ((long long)sm.s * ar.x + k) * pp.smp.h_stride;
        *u0 = halton_dimension(pp.smp, index, 5 + 2 * a);
        *u1 = halton_dimension(pp.smp, index, 6 + 2 * a);
        return;
    }
    int e = ar.y + (sm.s * ar.x + k) * 2;
    *u0 = pp.smp.tables[(size_t)e * pp.n_pix + sm.pix];
    *u1 = pp.smp.tables[(size_t)(e + 1) * pp.n_pix + sm.pix];
}
PB_DEV void samp_store(const PathState& ps, uint32_t p, const Samp& sm) {
    ps.rng[p] = sm.rng.state;
    ps.samp[p] = sm.dim1 | (sm.dim2 << 10) | (sm.arr << 16);
}

// ---- PixelSampler::start_pixel for one pixel per thread: StratifiedSampler (stratified.rs:44-104) and
// ZeroTwoSequenceSampler (zerotwosequence.rs:28-60), drawing from the pixel's own stream ----
struct PixelColumn {
    float* base;
    size_t n_pix;
    PB_DEV float& at(int e) const { return base[(size_t)e * n_pix]; }
};
PB_DEV uint32_t rng_bounded(Rng& r, uint32_t b) {  // rng.rs:37-45
    uint32_t threshold = (~b + 1u) % b;
    for (;;) {
        uint32_t v = rng_u32(r);
        if (v >= threshold) return v % b;
    }
}
// sampling.rs:280-287 over elements of `width` floats, in blocks of n_dimensions elements
PB_DEV void table_shuffle(const PixelColumn& c, int off, int count, int n_dimensions, int width, Rng& rng) {
    for (int i = 0; i < count; ++i) {
        int other = i + (int)rng_bounded(rng, (uint32_t)(count - i));
        for (int j = 0; j < n_dimensions * width; ++j) {
            float& a = c.at(off + n_dimensions * width * i + j);
            float& b = c.at(off + n_dimensions * width * other + j);
            float t = a;
            a = b;
            b = t;
        }
    }
}
PB_DEV uint32_t sobol_c1(int i) {  // second generator matrix of the (0,2) sequence (lowdiscrepancy.rs:481-488)
    // column i of Pascal's triangle mod 2: c1[0] = 1 << 31, c1[i] = c1[i-1] ^ (c1[i-1] >> 1)
    uint32_t v = 0x80000000u;
    for (int k = 0; k < i; ++k) v ^= v >> 1;
    return v;
}
// van_der_corput / sobol_2d (lowdiscrepancy.rs:436-505; D54 intended), width = 1 or 2
PB_DEV void table_gray_code(const PixelColumn& c, int off, int n_per, int n_pixel_samples, int width, Rng& rng) {
    uint32_t v0 = rng_u32(rng), v1 = width == 2 ? rng_u32(rng) : 0u;
    int total = n_per * n_pixel_samples;
    for (int i = 0; i < total; ++i) {
        c.at(off + i * width) = fminr(kOneMinusEpsilon, (float)v0 * 2.3283064365386963e-10f);
        if (width == 2) c.at(off + i * 2 + 1) = fminr(kOneMinusEpsilon, (float)v1 * 2.3283064365386963e-10f);
        int tz = __builtin_ctz((uint32_t)i + 1u);
        v0 ^= 0x80000000u >> tz;
        if (width == 2) v1 ^= sobol_c1(tz);
    }
    for (int i = 0; i < n_pixel_samples; ++i) table_shuffle(c, off + i * n_per * width, n_per, 1, width, rng);
    table_shuffle(c, off, n_pixel_samples, n_per, width, rng);
}
__global__ void k_sampler_tables(PassParams pp, TileList tiles) {
    int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= pp.n_pix) return;
    int2 org = tiles.origin[pix >> 8];
    int x = org.x + (pix & 15), y = org.y + ((pix & 255) >> 4);
    if (!(x >= pp.x0 && x < pp.x1 && y >= pp.y0 && y < pp.y1)) return;
    const SamplerParams& sp = pp.smp;
    PixelColumn c{sp.tables + pix, (size_t)pp.n_pix};
    Rng rng;
    rng_set_sequence(rng, pp.seed ^ (0x4000000000000000ULL | (uint64_t)pixel_number(pp, x, y)));
    const int n = pp.spp;
    if (sp.kind == PBRT_SAMPLER_STRATIFIED) {
        for (int d = 0; d < sp.n_dims; ++d) {  // stratified_sample_1d (sampling.rs:11-17) + shuffle
            float inv_n = 1.0f / (float)n;
            for (int i = 0; i < n; ++i) {
                float delta = sp.jitter ? rng_float(rng) : 0.5f;
                c.at(d * n + i) = fminr(kOneMinusEpsilon, ((float)i + delta) * inv_n);
            }
            table_shuffle(c, d * n, n, 1, 1, rng);
        }
        for (int d = 0; d < sp.n_dims; ++d) {  // stratified_sample_2d (sampling.rs:19-41) + shuffle
            float dx = 1.0f / (float)sp.nx, dy = 1.0f / (float)sp.ny;
            int off = sp.off2 + d * n * 2, i = 0;
            for (int yy = 0; yy < sp.ny; ++yy)
                for (int xx = 0; xx < sp.nx; ++xx) {
                    float jx = 0.5f, jy = 0.5f;
                    if (sp.jitter) {
                        jx = rng_float(rng);
                        jy = rng_float(rng);
                    }
                    c.at(off + 2 * i) = fminr(kOneMinusEpsilon, ((float)xx + jx) * dx);
                    c.at(off + 2 * i + 1) = fminr(kOneMinusEpsilon, ((float)yy + jy) * dy);
                    ++i;
                }
            table_shuffle(c, off, n, 1, 2, rng);
        }
        for (int a = 0; a < sp.n_arrays; ++a) {  // latin_hyper_cube per pixel sample (sampling.rs:44-66)
            int2 ar = sp.arrays[a];
            int count = ar.x;
            float inv_n = 1.0f / (float)count;
            for (int j = 0; j < n; ++j) {
                int off = ar.y + j * count * 2;
                for (int i = 0; i < count; ++i)
                    for (int k = 0; k < 2; ++k) c.at(off + 2 * i + k) = fminr(kOneMinusEpsilon, ((float)i + rng_float(rng)) * inv_n);
                for (int k = 0; k < 2; ++k)
                    for (int i = 0; i < count; ++i) {
                        int other = i + (int)rng_bounded(rng, (uint32_t)(count - i));
                        float& p0 = c.at(off + 2 * i + k);
                        float& p1 = c.at(off + 2 * other + k);
                        float t = p0;
                        p0 = p1;
                        p1 = t;
                    }
            }
        }
    } else {
        for (int d = 0; d < sp.n_dims; ++d) table_gray_code(c, d * n, 1, n, 1, rng);
        for (int d = 0; d < sp.n_dims; ++d) table_gray_code(c, sp.off2 + d * n * 2, 1, n, 2, rng);
        for (int a = 0; a < sp.n_arrays; ++a) table_gray_code(c, sp.arrays[a].y, sp.arrays[a].x, n, 2, rng);
    }
}

}  // namespace pb
