// wavefront.h — wavefront path tracer: SamplerIntegrator::render (src/core/integrator.rs:399-480)
// + PathIntegrator::li (src/integrators/path.rs:65-213) + uniform_sample_one_light / estimate_direct
// (src/core/integrator.rs:92-266) as a sequence of kernels over SoA path state in HBM.
//
// One pass handles `spp_per_pass` samples of every pixel of this GPU's tiles concurrently
// (N = pixels x samples paths). Per bounce: k_trace traces every pending ray of every path
// (continuation and MIS rays = Scene::intersect, shadow rays = Scene::intersect_p) in ONE launch;
// k_shade then, per path, (1) resolves the previous bounce's direct-lighting estimate from the
// shadow / MIS results, (2) processes the continuation hit: emission, SurfaceInteraction
// (src/shapes/triangle.rs:193-250), BSDF, light sampling, BSDF sampling, Russian roulette, and
// appends the next rays with wave-aggregated queue appends. Random numbers are drawn in the
// reference's order from the path's own PCG32 stream, so every decision matches the CPU path.
#pragma once
#include <hip/hip_runtime.h>

#include "scene.h"
#include "trace.h"
#include "trace_persistent.h"

namespace pb {

constexpr int kTile = 16;  // integrator.rs:404 TILE_SIZE
enum PathFlags : int {
    PF_SPECULAR_BOUNCE = 1,
    PF_ALIVE = 2,       // a continuation ray is pending in ray slot 0
    PF_NEE_SHADOW = 4,  // a shadow ray is pending in slot 2
    PF_NEE_MIS = 8,     // a BSDF-sampled MIS ray is pending in slot 1
    PF_VALID = 16,      // the path belongs to a pixel inside pixel_bounds
};
enum RaySlot : int { RS_CONT = 0, RS_MIS = 1, RS_SHADOW = 2 };

struct PathState {
    float4* ray;     // [ray_index(p, slot) + k]: (o.xyz, d.x) (d.yz, t_max, -)
    float4* hit;     // [ray_index(p, slot) + k]: (t, b0, b1, b2) (slot, -, -, -); shadow slot: .x = occluded
    size_t n_paths;  // paths of a pass (the stride between the three slots' arrays)
    uint64_t* rng;   // PCG32 state (inc is recomputed from the sample index)
    float4* L;       // L.rgb, eta_scale
    float4* beta;    // beta.rgb, (bounces << 8 | flags) as int bits
    float4* nee_a;   // light-sampling contribution (if unoccluded) rgb, light pick pdf
    float4* nee_f;   // BSDF-sampled f * |cos| rgb, MIS weight
    float4* nee_b;   // beta at the NEE vertex rgb, scattering pdf
    int* nee_light;  // light index of the pending estimate
    float2* pfilm;   // CameraSample::p_film
    int* samp;       // sampler counters: current_1d_dimension (Halton: dimension) | current_2d_dimension << 10 | array_2d_offset << 16
};

// Rays and hit records are kept slot-major, [slot][path][2 x float4]: the lanes of a wave that write (k_generate,
// k_shade) or read (the unsorted wavefronts of k_trace) the same slot of consecutive paths touch consecutive 32-B
// pieces, i.e. whole cache lines; path-major [path][slot] left two thirds of every written line untouched.
#ifndef PB_RAY_PATH_MAJOR
#define PB_RAY_PATH_MAJOR 0
#endif
#ifndef PB_HIT_PATH_MAJOR
#define PB_HIT_PATH_MAJOR 0
#endif
PB_DEV size_t ray_index(const PathState& ps, uint32_t p, int slot) {
    return PB_RAY_PATH_MAJOR ? ((size_t)p * 3 + slot) * 2 : ((size_t)slot * ps.n_paths + p) * 2;
}
PB_DEV size_t hit_index(const PathState& ps, uint32_t p, int slot) {
    return PB_HIT_PATH_MAJOR ? ((size_t)p * 3 + slot) * 2 : ((size_t)slot * ps.n_paths + p) * 2;
}

// PixelSampler tables (sampler.rs:252-318) of this GPU's pixels, one column per pixel:
// tables[elem * n_pix + pix]; elem: 1D dimension d, sample s -> d*spp + s; 2D -> off2 + (d*spp + s)*2 + c;
// requested 2D array a (n values per pixel sample) -> arrays[a].y + (s*n + k)*2 + c.
struct SamplerParams {
    int kind;      // PbrtSamplerKind
    int n_dims;    // n_sampled_dimensions
    int nx, ny, jitter;
    int n_arrays;
    int off2;      // first element of the 2D tables
    int n_elems;   // elements per pixel
    float* tables;
    const int2* arrays;  // per requested array: (n, first element)
    // HaltonSampler (halton.rs:24-37, 63-98) + GlobalSampler::array_end_dim (sampler.rs:344-345)
    int h_scale[2], h_exp[2];
    int h_stride, array_end_dim;
    unsigned int h_minv[2];
    const uint16_t* perms;         // compute_radical_inverse_permutations (lowdiscrepancy.rs:333-349)
    const uint32_t* primes;        // [0, 1000): primes, [1000, 2000): prime sums
};

struct PassParams {
    SamplerParams smp;
    int n_pix;         // pixels in this GPU's tile set (n_tiles * 256)
    int n_samples;     // samples of this pass
    int sample0;       // first sample index of this pass
    int spp;           // total samples per pixel (RNG keying)
    int width, height;
    int x0, y0, x1, y1;
    uint64_t seed;
    int max_depth;
    float rr_threshold;
    int light_strategy;
    float filter_rx, filter_ry;       // reconstruction filter radius
    const float* filter_table;        // 16 x 16 table (device), nullptr = 0.5 box (exact in-order path)
    float max_sample_luminance;       // Film::max_sample_luminance (film.rs:24), +inf = no clamp
};

struct Queues {
    uint32_t* trace;   // entries: path*4 + slot
    uint32_t* shade;   // entries: path
    // counts64[0]: low 32 bits = trace-queue length, high 32 bits = shadow rays among them;
    // counts64[1]: shade-queue length
    unsigned long long* counts64;
    // sort key of every trace-queue entry (ray_sort_key), written with the entry when the next wavefront will be
    // traced in Morton order; null otherwise
    uint32_t* keys;
    float key_lo[3], key_inv[3];  // scene bounds: lower corner, 1 / extent
};

#ifndef PB_SORT_AXIS_BITS
#define PB_SORT_AXIS_BITS 5
#endif
#ifndef PB_SORT_OCTANT
#define PB_SORT_OCTANT 0
#endif
constexpr int kSortKeyBits = 3 * PB_SORT_AXIS_BITS + 3 * PB_SORT_OCTANT + 1;
// Morton code of the cell of `o` in the scene bounds, PB_SORT_AXIS_BITS bits per axis
PB_DEV uint32_t ray_sort_cell(float ox, float oy, float oz, const float* lo, const float* inv) {
    constexpr float kCells = (float)(1 << PB_SORT_AXIS_BITS);
    float fx = (ox - lo[0]) * inv[0], fy = (oy - lo[1]) * inv[1], fz = (oz - lo[2]) * inv[2];
    uint32_t q[3] = {(uint32_t)fminf(fmaxf(fx * kCells, 0.0f), kCells - 1.0f), (uint32_t)fminf(fmaxf(fy * kCells, 0.0f), kCells - 1.0f),
                     (uint32_t)fminf(fmaxf(fz * kCells, 0.0f), kCells - 1.0f)};
    uint32_t code = 0;
    for (int b = 0; b < PB_SORT_AXIS_BITS; ++b)
        for (int k = 0; k < 3; ++k) code |= ((q[k] >> b) & 1u) << (3 * b + k);
    return code;
}

PB_DEV uint64_t sample_sequence(const PassParams& pp, int x, int y, int s) {
    return pp.seed ^ (uint64_t)(((int64_t)y * pp.width + x) * (int64_t)pp.spp + s);
}

// Block-aggregated queue append. Same-address atomics saturate near 10^2 per microsecond on the
// whole chip, so one atomic per wave (260 k waves per launch) would cost milliseconds: lanes are
// ranked inside the wave with ballot + mbcnt, waves inside the block through LDS, and ONE lane
// per block reserves the block's range with a single 64-bit atomicAdd per queue.
struct BlockAppend {
    uint32_t wave_rays[16];   // per-wave totals (blocks of up to 1024 threads)
    uint32_t wave_shadow[16];
    uint32_t wave_paths[16];
    uint32_t base_rays, base_paths;
};
PB_DEV uint32_t lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}
// Every thread of the block must call this. n_cont/n_mis/n_shadow in {0,1}; again = path stays in the shade queue.
// cell = ray_sort_cell of the point the path's rays leave from (used only when q.keys is set).
PB_DEV void block_append(BlockAppend& sh, const Queues& q, uint32_t p, bool cont, bool mis, bool shadow, bool again,
                         uint32_t cell = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63) >> 6;
    unsigned long long mc = __ballot(cont), mm = __ballot(mis), ms = __ballot(shadow), ma = __ballot(again);
    uint32_t wc = (uint32_t)__popcll(mc), wm = (uint32_t)__popcll(mm), ws = (uint32_t)__popcll(ms);
    if (lane == 0) {
        sh.wave_rays[wave] = wc + wm + ws;
        sh.wave_shadow[wave] = ws;
        sh.wave_paths[wave] = (uint32_t)__popcll(ma);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tr = 0, tsd = 0, tp = 0;
        for (int w = 0; w < n_waves; ++w) {
            uint32_t r = sh.wave_rays[w], pth = sh.wave_paths[w];
            sh.wave_rays[w] = tr;   // exclusive prefix
            sh.wave_paths[w] = tp;
            tr += r;
            tsd += sh.wave_shadow[w];
            tp += pth;
        }
        unsigned long long old = 0;
        if (tr) old = atomicAdd(&q.counts64[0], (unsigned long long)tr | ((unsigned long long)tsd << 32));
        sh.base_rays = (uint32_t)old;
        sh.base_paths = tp ? (uint32_t)atomicAdd(&q.counts64[1], (unsigned long long)tp) : 0u;
    }
    __syncthreads();
    uint32_t rbase = sh.base_rays + sh.wave_rays[wave];
    // within the wave: all continuation rays, then MIS rays, then shadow rays
    uint32_t ic = rbase + lane_prefix(mc), im = rbase + wc + lane_prefix(mm), is = rbase + wc + wm + lane_prefix(ms);
    if (cont) q.trace[ic] = p * 4u + RS_CONT;
    if (mis) q.trace[im] = p * 4u + RS_MIS;
    if (shadow) q.trace[is] = p * 4u + RS_SHADOW;
    if (q.keys) {  // the three rays leave from the same surface point: one cell, the any-hit flag on top
        if (PB_SORT_OCTANT) cell <<= 3;
        if (cont) q.keys[ic] = cell;
        if (mis) q.keys[im] = cell;
        if (shadow) q.keys[is] = cell | (1u << (kSortKeyBits - 1));
    }
    if (again) q.shade[sh.base_paths + sh.wave_paths[wave] + lane_prefix(ma)] = p;
}

PB_DEV void store_ray(const PathState& ps, uint32_t p, int slot, V3 o, V3 d, float tmax) {
    size_t i = ray_index(ps, p, slot);
    ps.ray[i] = make_float4(o.x, o.y, o.z, d.x);
    ps.ray[i + 1] = make_float4(d.y, d.z, tmax, 0.0f);
}

// ---- camera: PerspectiveCamera::generate_ray (cameras/perspective.rs:90-112) ----
struct DevCamera {
    float c2w[16], r2c[16];
    float lens_radius, focal_distance, shutter_open, shutter_close;
    int kind;  // PbrtCameraKind
};
PB_DEV V3 xform_point(const float* m, V3 p) {  // transform.rs:351-370
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1.0f) return V3{xp, yp, zp};
    return V3{xp, yp, zp} / wp;
}
PB_DEV V3 xform_vector(const float* m, V3 v) {  // transform.rs:372-385
    return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z};
}

struct TileList {
    const int2* origin;  // tile origins of this GPU
    int n_tiles;
};

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
    rng_set_sequence(rng, pp.seed ^ (0x4000000000000000ULL | (uint64_t)((int64_t)y * pp.width + x)));
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

__global__ void k_generate(PathState ps, Queues q, PassParams pp, DevCamera cam, TileList tiles) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = (uint32_t)pp.n_pix * pp.n_samples;
    if (p >= n) return;
    int s_local = p / pp.n_pix, pix = p % pp.n_pix;
    int tile = pix >> 8, within = pix & 255;
    int2 org = tiles.origin[tile];
    int x = org.x + (within & 15), y = org.y + (within >> 4);
    bool valid = x >= pp.x0 && x < pp.x1 && y >= pp.y0 && y < pp.y1;
    int flags = 0;
    if (valid) {
        int s = pp.sample0 + s_local;
        Samp sm;
        rng_set_sequence(sm.rng, sample_sequence(pp, x, y, s));
        sm.pix = pix;
        sm.s = s;
        sm.dim1 = sm.dim2 = sm.arr = 0;
        sm.h_offset = pp.smp.kind == PBRT_SAMPLER_HALTON ? halton_pixel_offset(pp.smp, x, y) : 0;
        // Sampler::get_camera_sample (sampler.rs:27-33): 2D film, 1D time, 2D lens
        float u0, u1, l0, l1;
        samp_2d(pp, sm, &u0, &u1);
        float pfx = (float)x + u0, pfy = (float)y + u1;
        float time_u = samp_1d(pp, sm);
        samp_2d(pp, sm, &l0, &l1);
        V3 o = V3{0.0f, 0.0f, 0.0f}, d;
        if (cam.kind == PBRT_CAMERA_ENVIRONMENT) {
            // EnvironmentCamera::generate_ray (cameras/environment.rs:37-56)
            float theta = kPi * pfy / (float)pp.height;
            float phi = 2.0f * kPi * pfx / (float)pp.width;
            float st, ct, sp, cp;
            det_sincos(theta, &st, &ct);
            det_sincos(phi, &sp, &cp);
            d = V3{st * cp, ct, st * sp};
        } else if (cam.kind == PBRT_CAMERA_ORTHOGRAPHIC) {
            // OrthographicCamera::generate_ray (cameras/orthographic.rs:82-104; D58: the lens point is added to the origin)
            o = xform_point(cam.r2c, V3{pfx, pfy, 0.0f});
            d = V3{0.0f, 0.0f, 1.0f};
            if (cam.lens_radius > 0.0f) {
                float lx, ly;
                concentric_sample_disk(l0, l1, &lx, &ly);
                lx *= cam.lens_radius;
                ly *= cam.lens_radius;
                float ft = cam.focal_distance / d.z;
                V3 p_focus = o + d * ft;
                o.x += lx;
                o.y += ly;
                d = normalize(p_focus - o);
            }
        } else {
            V3 p_camera = xform_point(cam.r2c, V3{pfx, pfy, 0.0f});
            d = normalize(p_camera);
            if (cam.lens_radius > 0.0f) {
                float lx, ly;
                concentric_sample_disk(l0, l1, &lx, &ly);
                lx *= cam.lens_radius;
                ly *= cam.lens_radius;
                float ft = cam.focal_distance / d.z;
                V3 p_focus = o + d * ft;
                o = V3{lx, ly, 0.0f};
                d = normalize(p_focus - o);
            }
        }
        (void)time_u;  // ray.time only feeds animated transforms / media (out of scope)
        // Ray through camera_to_world with origin error (geometry.rs:865-881, 898-935)
        const float* m = cam.c2w;
        V3 ow = xform_point(m, o);
        float xa = __builtin_fabsf(m[0] * o.x) + __builtin_fabsf(m[1] * o.y) + __builtin_fabsf(m[2] * o.z) + __builtin_fabsf(m[3]);
        float ya = __builtin_fabsf(m[4] * o.x) + __builtin_fabsf(m[5] * o.y) + __builtin_fabsf(m[6] * o.z) + __builtin_fabsf(m[7]);
        float za = __builtin_fabsf(m[8] * o.x) + __builtin_fabsf(m[9] * o.y) + __builtin_fabsf(m[10] * o.z) + __builtin_fabsf(m[11]);
        V3 o_err = V3{xa, ya, za} * kGamma3;
        V3 dw = xform_vector(m, d);
        float l2 = len2(dw);
        float tmax = kInf;
        if (l2 > 0.0f) {
            float dt = dot(vabs(dw), o_err) / l2;
            ow = ow + dw * dt;
            tmax -= dt;
        }
        store_ray(ps, p, RS_CONT, ow, dw, tmax);
        samp_store(ps, p, sm);
        ps.pfilm[p] = make_float2(pfx, pfy);
        flags = PF_VALID | PF_ALIVE;
    } else {
        ps.pfilm[p] = make_float2(0.0f, 0.0f);
    }
    ps.L[p] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    ps.beta[p] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(flags));
    // The first wavefront is every path in order: identity queues, no atomics. Paths of pixels
    // outside pixel_bounds (partial border tiles) carry a ray that misses at once (t_max < 0).
    if (!valid) store_ray(ps, p, RS_CONT, V3{0.0f, 0.0f, 0.0f}, V3{0.0f, 0.0f, 1.0f}, -1.0f);
    q.trace[p] = p * 4u + RS_CONT;
    q.shade[p] = p;
}

// ---- trace: every pending ray of the wavefront (trace_persistent.h) ----
struct WavefrontRayIO {
    PathState ps;
    const uint32_t* __restrict__ queue;
    uint32_t count;
    int n_segments;
    PB_DEV uint32_t n() const { return count; }
    PB_DEV int segments() const { return n_segments; }
    PB_DEV bool load(uint32_t i, TravRay* r, bool* any) const {
        uint32_t e = queue[i];
        uint32_t p = e >> 2, slot = e & 3u;
        size_t ri = ray_index(ps, p, slot);
        float4 a = ps.ray[ri], b = ps.ray[ri + 1];
        *r = TravRay{a.x, a.y, a.z, a.w, b.x, b.y, b.z};
        *any = slot == RS_SHADOW;
        // t_max < 0 marks the placeholder ray of a path outside pixel_bounds (k_generate): not a ray of the frame
        return !(b.z < 0.0f);
    }
    PB_DEV void store(uint32_t i, bool any, bool found, float t, float b0, float b1, float b2, int slot, int inst) const {
        uint32_t e = queue[i];
        uint32_t p = e >> 2, rs = e & 3u;
        size_t ri = hit_index(ps, p, rs);
        if (any) {
            ps.hit[ri] = make_float4(found ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            ps.hit[ri] = make_float4(t, b0, b1, b2);
            ps.hit[ri + 1] = make_float4(__int_as_float(found ? slot : -1), __int_as_float(inst), 0.0f, 0.0f);
        }
    }
};
// Sort key of a queued ray: any-hit flag, then a 15-bit Morton code of the origin inside the scene bounds. Rays that
// start close together walk the same part of the tree: in cache order the traversal kernel runs 20 % faster on
// incoherent bounce rays (tools/probe_sorting.py), which pays for the two 8-bit radix passes.
// The stand-alone form of the key (the fused one is written by k_shade's block_append): PBRT_HIP_SORT_FUSED=0, and
// the builds with direction-octant bits.
__global__ void k_ray_sort_keys(PathState ps, const uint32_t* __restrict__ queue, uint32_t n, float3 lo, float3 inv_extent,
                                uint32_t* __restrict__ keys) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t e = queue[i];
    uint32_t p = e >> 2, slot = e & 3u;
    float4 a = ps.ray[ray_index(ps, p, slot)];
    const float l[3] = {lo.x, lo.y, lo.z}, iv[3] = {inv_extent.x, inv_extent.y, inv_extent.z};
    uint32_t code = ray_sort_cell(a.x, a.y, a.z, l, iv);
#if PB_SORT_OCTANT
    float4 d = ps.ray[ray_index(ps, p, slot) + 1];  // (d.y, d.z, t_max, -); d.x rides in a.w
    code = (code << 3) | (a.w < 0.0f ? 1u : 0u) | (d.x < 0.0f ? 2u : 0u) | (d.y < 0.0f ? 4u : 0u);
#endif
    keys[i] = code | ((slot == RS_SHADOW ? 1u : 0u) << (kSortKeyBits - 1));
}

#ifndef PB_TRACE_WAVES
#define PB_TRACE_WAVES 6
#endif
#ifndef PB_INST_WAVES
#define PB_INST_WAVES 5
#endif
template <bool COUNT, bool INST, bool SPH = false>
__global__ void __launch_bounds__(kTraceBlock, (COUNT || SPH) ? 4 : (INST ? PB_INST_WAVES : PB_TRACE_WAVES))
    k_trace(DevBVH bvh, PathState ps, const uint32_t* __restrict__ queue, uint32_t n, unsigned int* work_counter,
            unsigned long long* counters, int segments) {
    __shared__ uint2 lds_stack[kStackLds * kTraceBlock];
    WavefrontRayIO io{ps, queue, n, segments};
    trace_persistent<WavefrontRayIO, COUNT, INST, SPH>(bvh, io, work_counter, lds_stack + threadIdx.x,
                                                  blockIdx.x * kTraceBlock + threadIdx.x, counters);
}

// ---- shading helpers ----
struct Surf {  // the parts of SurfaceInteraction the path needs
    V3 p, p_error, n, dpdu, wo;
    V3 ns, sdpdu;         // shading.n, shading.dpdu (= n, dpdu without per-vertex normals)
    int material, light;  // light = index or -1
};

PB_DEV void tri_vertices(const DevBVH& bvh, int slot, V3* p0, V3* p1, V3* p2, int* prim, int* mat, int* light) {
    float4 a = bvh.tris[3 * (size_t)slot], b = bvh.tris[3 * (size_t)slot + 1], c = bvh.tris[3 * (size_t)slot + 2];
    *p0 = V3{a.x, a.y, a.z};
    *p1 = V3{a.w, b.x, b.y};
    *p2 = V3{b.z, b.w, c.x};
    *prim = __float_as_int(c.y);
    *mat = __float_as_int(c.z);
    *light = (__float_as_int(c.w) & kPrimLightMask) - 1;
}

// Triangle::intersect past the hit test (triangle.rs:193-316): dpdu from the (default or per-vertex) uvs, the
// geometric normal, and with per-vertex normals the shading frame of :252-312 with
// set_shading_geometry(.., orientation_is_authoritative = true) (interaction.rs:302-316), which also flips
// the geometric normal to the shading normal's side. Outputs: n, shading.n, dpdu, shading.dpdu.
PB_DEV void tri_shading_geometry(const DevBVH& bvh, int slot, V3 p0, V3 p1, V3 p2, float b0, float b1, float b2, V3* n_out,
                                 V3* ns_out, V3* dpdu_out, V3* sdpdu_out) {
    float uv0x = 0.0f, uv0y = 0.0f, uv1x = 1.0f, uv1y = 0.0f, uv2x = 1.0f, uv2y = 1.0f;  // triangle.rs:66-70
    float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, s2 = s0, s3 = s0, s4 = s0;
    if (bvh.tri_shading) {
        const float4* sh = bvh.tri_shading + 6 * (size_t)slot;
        s0 = sh[0];
        s1 = sh[1];
        s2 = sh[2];
        s3 = sh[3];
        s4 = sh[4];
        if (bvh.has_uvs) {
            float4 s5 = sh[5];
            uv0x = s4.z;
            uv0y = s4.w;
            uv1x = s5.x;
            uv1y = s5.y;
            uv2x = s5.z;
            uv2y = s5.w;
        }
    }
    float duv02x = uv0x - uv2x, duv02y = uv0y - uv2y, duv12x = uv1x - uv2x, duv12y = uv1y - uv2y;
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerate_uv = __builtin_fabsf(determinant) < 1e-8f;
    V3 dpdu = V3{0.0f, 0.0f, 0.0f}, dpdv = V3{0.0f, 0.0f, 0.0f};
    if (!degenerate_uv) {
        float inv_det = 1.0f / determinant;
        dpdu = (dp02 * duv12y - dp12 * duv02y) * inv_det;
        dpdv = (dp02 * -duv12x + dp12 * duv02x) * inv_det;
    }
    if (degenerate_uv || len2(cross(dpdu, dpdv)) == 0.0f) {
        V3 ng = cross(p2 - p0, p1 - p0);  // zero only for triangles flagged kTriDegenerate, which never get here
        coordinate_system(normalize(ng), &dpdu, &dpdv);
    }
    V3 n = normalize(cross(dp02, dp12));  // triangle.rs:244-245 (no orientation flip: D14)
    V3 ns = n, sdpdu = dpdu;
    if (bvh.tri_shading && (bvh.has_normals || bvh.has_tangents)) {
        V3 nsi = n;
        if (bvh.has_normals) {
            V3 n0 = V3{s0.x, s0.y, s0.z}, n1 = V3{s0.w, s1.x, s1.y}, n2 = V3{s1.z, s1.w, s2.x};
            nsi = n0 * b0 + n1 * b1 + n2 * b2;
            nsi = len2(nsi) > 0.0f ? normalize(nsi) : n;
        }
        V3 ss = normalize(dpdu);
        if (bvh.has_tangents) {  // triangle.rs:265-275
            V3 t0 = V3{s2.y, s2.z, s2.w}, t1 = V3{s3.x, s3.y, s3.z}, t2 = V3{s3.w, s4.x, s4.y};
            V3 si = t0 * b0 + t1 * b1 + t2 * b2;
            if (len2(si) > 0.0f) ss = normalize(si);
        }
        V3 ts = cross(ss, nsi);
        if (len2(ts) > 0.0f) {
            ts = normalize(ts);
            ss = cross(ts, nsi);
        } else {
            coordinate_system(nsi, &ss, &ts);
        }
        ns = normalize(cross(ss, ts));          // set_shading_geometry: shading.n = normalize(dpdu x dpdv)
        if (dot(n, ns) < 0.0f) n = -n;          // n = n.face_forward(shading.n)
        sdpdu = ss;
    }
    *n_out = n;
    *ns_out = ns;
    *dpdu_out = dpdu;
    *sdpdu_out = sdpdu;
}
PB_DEV Surf make_surface(const DevBVH& bvh, int slot, float b0, float b1, float b2, V3 ray_d) {
    V3 p0, p1, p2;
    int prim;
    Surf s;
    tri_vertices(bvh, slot, &p0, &p1, &p2, &prim, &s.material, &s.light);
    tri_shading_geometry(bvh, slot, p0, p1, p2, b0, b1, b2, &s.n, &s.ns, &s.dpdu, &s.sdpdu);
    float xs = __builtin_fabsf(b0 * p0.x) + __builtin_fabsf(b1 * p1.x) + __builtin_fabsf(b2 * p2.x);
    float ys = __builtin_fabsf(b0 * p0.y) + __builtin_fabsf(b1 * p1.y) + __builtin_fabsf(b2 * p2.y);
    float zs = __builtin_fabsf(b0 * p0.z) + __builtin_fabsf(b1 * p1.z) + __builtin_fabsf(b2 * p2.z);
    s.p_error = V3{xs, ys, zs} * kGamma7;
    s.p = p0 * b0 + p1 * b1 + p2 * b2;
    s.wo = -ray_d;
    return s;
}
// Transform::operator()(SurfaceInteraction) of pbrt-v3 (transform.rs:620-627 is a TODO in the reference, D6):
// object-space hit -> world space through the instance's matrices (primitive.rs:145-147).
PB_DEV void instance_to_world(const DevBVH& bvh, int inst_slot, Surf* s) {
    const float4* m = bvh.instances + 7 * (size_t)inst_slot;
    float4 o0 = m[0], o1 = m[1], o2 = m[2];  // to_object rows (= inverse of to_world)
    float4 w0 = m[3], w1 = m[4], w2 = m[5];  // to_world rows
    V3 p = s->p, pe = s->p_error;
    // point with incoming absolute error (geometry.rs:936-1000)
    float xp = w0.x * p.x + w0.y * p.y + w0.z * p.z + w0.w;
    float yp = w1.x * p.x + w1.y * p.y + w1.z * p.z + w1.w;
    float zp = w2.x * p.x + w2.y * p.y + w2.z * p.z + w2.w;
    const float g3 = kGamma3;
    V3 err;
    err.x = (g3 + 1.0f) * (__builtin_fabsf(w0.x * pe.x) + __builtin_fabsf(w0.y * pe.y) + __builtin_fabsf(w0.z * pe.z)) +
            g3 * (__builtin_fabsf(w0.x * p.x) + __builtin_fabsf(w0.y * p.y) + __builtin_fabsf(w0.z * p.z) + __builtin_fabsf(w0.w));
    err.y = (g3 + 1.0f) * (__builtin_fabsf(w1.x * pe.x) + __builtin_fabsf(w1.y * pe.y) + __builtin_fabsf(w1.z * pe.z)) +
            g3 * (__builtin_fabsf(w1.x * p.x) + __builtin_fabsf(w1.y * p.y) + __builtin_fabsf(w1.z * p.z) + __builtin_fabsf(w1.w));
    err.z = (g3 + 1.0f) * (__builtin_fabsf(w2.x * pe.x) + __builtin_fabsf(w2.y * pe.y) + __builtin_fabsf(w2.z * pe.z)) +
            g3 * (__builtin_fabsf(w2.x * p.x) + __builtin_fabsf(w2.y * p.y) + __builtin_fabsf(w2.z * p.z) + __builtin_fabsf(w2.w));
    s->p = V3{xp, yp, zp};
    s->p_error = err;
    // normal: (M^-1)^T n (transform.rs:387-403, intended form), then normalised
    V3 n = s->n;
    s->n = normalize(V3{o0.x * n.x + o1.x * n.y + o2.x * n.z, o0.y * n.x + o1.y * n.y + o2.y * n.z,
                        o0.z * n.x + o1.z * n.y + o2.z * n.z});
    {   // shading.n, shading.dpdu, then shading.n = face_forward(shading.n, n) (pbrt-v3 Transform(SurfaceInteraction))
        V3 sn = s->ns, sd = s->sdpdu;
        sn = normalize(V3{o0.x * sn.x + o1.x * sn.y + o2.x * sn.z, o0.y * sn.x + o1.y * sn.y + o2.y * sn.z,
                          o0.z * sn.x + o1.z * sn.y + o2.z * sn.z});
        s->ns = dot(sn, s->n) < 0.0f ? -sn : sn;
        s->sdpdu = V3{w0.x * sd.x + w0.y * sd.y + w0.z * sd.z, w1.x * sd.x + w1.y * sd.y + w1.z * sd.z,
                      w2.x * sd.x + w2.y * sd.y + w2.z * sd.z};
    }
    V3 wo = s->wo, du = s->dpdu;
    s->wo = normalize(V3{w0.x * wo.x + w0.y * wo.y + w0.z * wo.z, w1.x * wo.x + w1.y * wo.y + w1.z * wo.z,
                         w2.x * wo.x + w2.y * wo.y + w2.z * wo.z});
    s->dpdu = V3{w0.x * du.x + w0.y * du.y + w0.z * du.z, w1.x * du.x + w1.y * du.y + w1.z * du.z,
                 w2.x * du.x + w2.y * du.y + w2.z * du.z};
    int mat = __float_as_int(m[6].x);
    if (mat >= 0) s->material = mat;
    s->light = -1;  // instanced primitives carry no area lights
}
// Sphere::intersect past the hit test (sphere.rs:38-92) for a full sphere placed by translate(centre): partial
// derivatives, SurfaceInteraction::new, then pbrt-v3's Transform(SurfaceInteraction) through that translation (every
// product of the general matrix formulas is kept, as in sphere_object_ray). ph = the refined object-space hit point.
PB_DEV Surf make_surface_sphere(const DevBVH& bvh, int slot, V3 ph, V3 rd) {
    float4 a = bvh.tris[3 * (size_t)slot], c4 = bvh.tris[3 * (size_t)slot + 2];
    float cx = a.x, cy = a.y, cz = a.z, radius = a.w;
    Surf s;
    s.material = __float_as_int(c4.z);
    s.light = (__float_as_int(c4.w) & kPrimLightMask) - 1;
    const float phi_max = 360.0f * (kPi / 180.0f);
    const float theta_min = det_acos(clampf(fminr(-radius, radius) / radius, -1.0f, 1.0f));
    const float theta_max = det_acos(clampf(fmaxr(-radius, radius) / radius, -1.0f, 1.0f));
    float theta = det_acos(clampf(ph.z / radius, -1.0f, 1.0f));
    float z_radius = __builtin_sqrtf(ph.x * ph.x + ph.y * ph.y);
    float inv_z_radius = 1.0f / z_radius;
    float cos_phi = ph.x * inv_z_radius, sin_phi = ph.y * inv_z_radius;
    V3 dpdu = V3{-phi_max * ph.y, phi_max * ph.x, 0.0f};
    V3 dpdv = V3{ph.z * cos_phi, ph.z * sin_phi, -radius * det_sin(theta)} * (theta_max - theta_min);
    V3 n = normalize(cross(dpdu, dpdv));  // SurfaceInteraction::new (interaction.rs:248-300)
    V3 pe = vabs(ph) * kGamma5;
    // Transform(SurfaceInteraction) with m = translate(c), m_inv = translate(-c)
    const float g3 = kGamma3;
    float x = ph.x, y = ph.y, z = ph.z;
    s.p = V3{1.0f * x + 0.0f * y + 0.0f * z + cx, 0.0f * x + 1.0f * y + 0.0f * z + cy, 0.0f * x + 0.0f * y + 1.0f * z + cz};
    s.p_error.x = (g3 + 1.0f) * (__builtin_fabsf(1.0f * pe.x) + __builtin_fabsf(0.0f * pe.y) + __builtin_fabsf(0.0f * pe.z)) +
                  g3 * (__builtin_fabsf(1.0f * x) + __builtin_fabsf(0.0f * y) + __builtin_fabsf(0.0f * z) + __builtin_fabsf(cx));
    s.p_error.y = (g3 + 1.0f) * (__builtin_fabsf(0.0f * pe.x) + __builtin_fabsf(1.0f * pe.y) + __builtin_fabsf(0.0f * pe.z)) +
                  g3 * (__builtin_fabsf(0.0f * x) + __builtin_fabsf(1.0f * y) + __builtin_fabsf(0.0f * z) + __builtin_fabsf(cy));
    s.p_error.z = (g3 + 1.0f) * (__builtin_fabsf(0.0f * pe.x) + __builtin_fabsf(0.0f * pe.y) + __builtin_fabsf(1.0f * pe.z)) +
                  g3 * (__builtin_fabsf(0.0f * x) + __builtin_fabsf(0.0f * y) + __builtin_fabsf(1.0f * z) + __builtin_fabsf(cz));
    auto through = [](V3 v) {  // upper 3x3 of either matrix (the identity), as xform_vector / xform_normal evaluate it
        return V3{1.0f * v.x + 0.0f * v.y + 0.0f * v.z, 0.0f * v.x + 1.0f * v.y + 0.0f * v.z, 0.0f * v.x + 0.0f * v.y + 1.0f * v.z};
    };
    s.n = normalize(through(n));
    V3 d_obj = through(rd);  // the object-space ray direction of sphere_object_ray
    s.wo = normalize(through(-d_obj));
    s.dpdu = through(dpdu);
    V3 sn = normalize(through(n));  // shading.n = n before the transform (D47)
    s.ns = dot(sn, s.n) < 0.0f ? -sn : sn;
    s.sdpdu = through(dpdu);
    return s;
}
// Hit record -> world-space surface. `rd` is the world-space ray direction.
PB_DEV Surf surface_from_hit(const DevBVH& bvh, int slot, int inst_slot, float b0, float b1, float b2, V3 rd) {
    if (bvh.has_spheres && (__float_as_int(bvh.tris[3 * (size_t)slot + 2].w) & kPrimSphere))
        return make_surface_sphere(bvh, slot, V3{b0, b1, b2}, rd);
    if (bvh.instanced && inst_slot >= 0) {
        const float4* m = bvh.instances + 7 * (size_t)inst_slot;
        float4 r0 = m[0], r1 = m[1], r2 = m[2];
        // the object-space ray direction TransformedPrimitive::intersect traced (geometry.rs:872)
        V3 d_obj = V3{r0.x * rd.x + r0.y * rd.y + r0.z * rd.z, r1.x * rd.x + r1.y * rd.y + r1.z * rd.z,
                      r2.x * rd.x + r2.y * rd.y + r2.z * rd.z};
        Surf s = make_surface(bvh, slot, b0, b1, b2, d_obj);
        instance_to_world(bvh, inst_slot, &s);
        return s;
    }
    return make_surface(bvh, slot, b0, b1, b2, rd);
}
// SurfaceInteraction::n of a hit at barycentrics (b0, b1, b2): the geometric normal, on the shading normal's side
PB_DEV V3 tri_interaction_normal(const DevBVH& bvh, int slot, float b0, float b1, float b2) {
    if (bvh.has_spheres && (__float_as_int(bvh.tris[3 * (size_t)slot + 2].w) & kPrimSphere))
        return make_surface_sphere(bvh, slot, V3{b0, b1, b2}, V3{0.0f, 0.0f, 1.0f}).n;  // (b0, b1, b2) = the hit point
    V3 p0, p1, p2;
    int a, b, c;
    tri_vertices(bvh, slot, &p0, &p1, &p2, &a, &b, &c);
    if (!(bvh.tri_shading && (bvh.has_normals || bvh.has_tangents))) return normalize(cross(p0 - p2, p1 - p2));
    V3 n, ns, dpdu, sdpdu;
    tri_shading_geometry(bvh, slot, p0, p1, p2, b0, b1, b2, &n, &ns, &dpdu, &sdpdu);
    return n;
}

struct Frame {  // BSDF::new (reflection.rs:220-234)
    V3 ss, ts, ns, ng;
};
PB_DEV V3 to_local(const Frame& f, V3 v) { return V3{dot(v, f.ss), dot(v, f.ts), dot(v, f.ns)}; }
PB_DEV V3 to_world(const Frame& f, V3 v) {
    return V3{f.ss.x * v.x + f.ts.x * v.y + f.ns.x * v.z, f.ss.y * v.x + f.ts.y * v.y + f.ns.y * v.z,
              f.ss.z * v.x + f.ts.z * v.y + f.ns.z * v.z};
}

// BSDF::f and BSDF::pdf for the non-specular query of estimate_direct: only the Lambertian lobe of
// a matte material matches (reflection.rs:264-283, 414-446, 475-481, 840-842).
PB_DEV void matte_f_pdf(const Frame& fr, V3 kd, V3 wo_w, V3 wi_w, V3* f, float* pdf) {
    V3 wi = to_local(fr, wi_w), wo = to_local(fr, wo_w);
    *f = V3{0.0f, 0.0f, 0.0f};
    *pdf = 0.0f;
    if (wo.z == 0.0f) return;
    bool reflect = dot(wi_w, fr.ng) * dot(wo_w, fr.ng) > 0.0f;
    if (reflect) *f = V3{0.0f + kd.x * kInvPi, 0.0f + kd.y * kInvPi, 0.0f + kd.z * kInvPi};
    float p = (wo.z * wi.z > 0.0f) ? __builtin_fabsf(wi.z) * kInvPi : 0.0f;
    *pdf = (0.0f + p) / 1.0f;
}
// BSDF::sample_f with one Lambertian lobe (reflection.rs:285-377, 459-472)
PB_DEV V3 matte_sample_f(const Frame& fr, V3 kd, V3 wo_w, float u0, float u1, V3* wi_w, float* pdf, bool* ok) {
    *ok = false;
    V3 zero = V3{0.0f, 0.0f, 0.0f};
    float ur = fminr(u0 * 1.0f - 0.0f, kOneMinusEpsilon);
    V3 wo = to_local(fr, wo_w);
    if (wo.z == 0.0f) return zero;  // pdf keeps the caller's value (reflection.rs:323-326)
    V3 wi = cosine_sample_hemisphere(ur, u1);
    if (wo.z < 0.0f) wi.z *= -1.0f;
    *pdf = (wo.z * wi.z > 0.0f) ? __builtin_fabsf(wi.z) * kInvPi : 0.0f;
    if (*pdf == 0.0f) return zero;
    *wi_w = to_world(fr, wi);
    *ok = true;
    bool reflect = dot(*wi_w, fr.ng) * dot(wo_w, fr.ng) > 0.0f;
    if (!reflect) return zero;
    return V3{0.0f + kd.x * kInvPi, 0.0f + kd.y * kInvPi, 0.0f + kd.z * kInvPi};
}

PB_DEV bool is_black(V3 c) { return c.x == 0.0f && c.y == 0.0f && c.z == 0.0f; }
PB_DEV V3 mulv(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PB_DEV float max_comp(V3 c) {
    float m = -kFloatMax;
    m = (m > c.x) ? m : c.x;
    m = (m > c.y) ? m : c.y;
    m = (m > c.z) ? m : c.z;
    return m;
}

// find_interval over a cdf with predicate cdf[i] < u (pbrt.rs:229-243, sampling.rs:107)
PB_DEV int find_interval_cdf(const float* cdf, int size, float u) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] < u) {
            first = middle + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    int v = first - 1;
    return v < 0 ? 0 : (v > size - 2 ? size - 2 : v);
}
// Distribution1D::sample_continuous (sampling.rs:99-123) on a 2-bin table
PB_DEV float sample_continuous2(const float* func, const float* cdf, float func_int, float u, float* pdf, int* off) {
    int offset = find_interval_cdf(cdf, 3, u);
    *off = offset;
    float du = u - cdf[offset];
    if (cdf[offset + 1] - cdf[offset] > 0.0f) du /= cdf[offset + 1] - cdf[offset];
    *pdf = func_int > 0.0f ? func[offset] / func_int : 0.0f;
    return ((float)offset + du) / 2.0f;
}

// Full Triangle::intersect of ONE triangle for Shape::pdf2 (shape.rs:54-69): returns hit point and normal
PB_DEV bool light_triangle_intersect(const DevBVH& bvh, int slot, V3 o, V3 d, V3* p_hit, V3* n_hit) {
    V3 p0, p1, p2;
    int flags;
    load_tri(bvh.tris, slot, &p0, &p1, &p2, &flags);
    TravRay r{o.x, o.y, o.z, d.x, d.y, d.z, kInf};
    TriRayConst c = tri_ray_setup(r);
    float b0, b1, b2, t;
    if (!triangle_test(p0, p1, p2, r, c, kInf, &b0, &b1, &b2, &t)) return false;
    if (flags & kTriDegenerate) return false;
    *p_hit = p0 * b0 + p1 * b1 + p2 * b2;
    *n_hit = tri_interaction_normal(bvh, slot, b0, b1, b2);
    return true;
}

struct ShadeConsts {
    DevBVH bvh;
    const DevMaterial* materials;
    const DevLight* lights;
    int n_lights, n_infinite;
    const int* infinite_ids;
    DevDistribution1D distrib;  // light_distribution.lookup (lightdistrib.rs:43/66)
    float env_cond_func[2][2], env_cond_cdf[2][3], env_cond_int[2];
    float env_marg_func[2], env_marg_cdf[3], env_marg_int;
    float world_radius;
    // DirectLightingIntegrator (directlighting.rs:58-78): per-light sample counts, prefix sums
    const int* light_sample_prefix;  // [n_lights + 1]
    int total_light_samples;
    // SpatialLightDistribution (lightdistrib.rs:76-220): one Distribution1D per voxel of the scene bounds,
    // spatial[v * (2 n + 2)] = func[n], cdf[n + 1], func_int; null = the fixed `distrib`
    const float* spatial;
    int n_voxel[3];
};

// SpatialLightDistribution::lookup (lightdistrib.rs:171-182): the voxel of p, then its distribution
PB_DEV DevDistribution1D light_distribution_lookup(const ShadeConsts& sc, V3 p) {
    if (!sc.spatial) return sc.distrib;
    const float* mn = sc.bvh.root_min;
    const float* mx = sc.bvh.root_max;
    float o[3] = {p.x - mn[0], p.y - mn[1], p.z - mn[2]};  // Bounds3::offset
    int pi[3];
    for (int i = 0; i < 3; ++i) {
        if (mx[i] > mn[i]) o[i] /= mx[i] - mn[i];
        int v = (int)(o[i] * (float)sc.n_voxel[i]);
        pi[i] = v < 0 ? 0 : (v > sc.n_voxel[i] - 1 ? sc.n_voxel[i] - 1 : v);
    }
    size_t voxel = ((size_t)pi[2] * sc.n_voxel[1] + pi[1]) * sc.n_voxel[0] + pi[0];
    const float* t = sc.spatial + voxel * (size_t)(2 * sc.n_lights + 2);
    DevDistribution1D d;
    d.func = t;
    d.cdf = t + sc.n_lights;
    d.func_int = t[2 * sc.n_lights + 1];
    d.n = sc.n_lights;
    return d;
}

PB_DEV Samp path_sampler(const PathState& ps, const PassParams& pp, const TileList& tiles, uint32_t p) {
    // this path's stream: inc from the (pixel, sample) index, state and the dimension counters from memory
    int s_local = p / pp.n_pix, pix = p % pp.n_pix;
    int2 org = tiles.origin[pix >> 8];
    int x = org.x + (pix & 15), y = org.y + ((pix & 255) >> 4);
    Samp sm;
    sm.rng.inc = (sample_sequence(pp, x, y, pp.sample0 + s_local) << 1) | 1;
    sm.rng.state = ps.rng[p];
    sm.pix = pix;
    sm.s = pp.sample0 + s_local;
    int c = ps.samp[p];
    sm.dim1 = c & 0x3ff;
    sm.dim2 = (c >> 10) & 0x3f;
    sm.arr = (c >> 16) & 0xffff;
    sm.h_offset = pp.smp.kind == PBRT_SAMPLER_HALTON ? halton_pixel_offset(pp.smp, x, y) : 0;
    return sm;
}

PB_DEV Frame make_frame(const Surf& sf) {  // BSDF::new (reflection.rs:220-234)
    Frame fr;
    fr.ns = sf.ns;
    fr.ng = sf.n;
    fr.ss = normalize(sf.sdpdu);
    fr.ts = cross(fr.ns, fr.ss);
    return fr;
}

// SurfaceInteraction::le (interaction.rs:387-395) -> DiffuseAreaLight::l (diffuse.rs:150-156)
PB_DEV V3 surface_le(const ShadeConsts& sc, const Surf& sf, V3 w) {
    if (sf.light >= 0) {
        DevLight lt = sc.lights[sf.light];
        if (lt.two_sided || dot(sf.n, w) > 0.0f) return V3{lt.L[0], lt.L[1], lt.L[2]};
    }
    return V3{0.0f, 0.0f, 0.0f};
}

// ---- a Sphere as the shape of a DiffuseAreaLight: Sphere::sample / sample2 / pdf2 (sphere.rs:103-192), full sphere
// placed by translate(c) ----
PB_DEV V3 sphere_through(V3 v) {  // upper 3x3 of translate(+-c) as the matrix formulas evaluate it
    return V3{1.0f * v.x + 0.0f * v.y + 0.0f * v.z, 0.0f * v.x + 1.0f * v.y + 0.0f * v.z, 0.0f * v.x + 0.0f * v.y + 1.0f * v.z};
}
PB_DEV float sphere_area(float radius) { return (360.0f * (kPi / 180.0f)) * radius * (radius - (-radius)); }  // sphere.rs:99-101
// Shape::sample2 for the sphere: point, error, normal on the sphere and the solid-angle pdf from `sf`
PB_DEV void sphere_light_sample(float cx, float cy, float cz, float radius, const Surf& sf, float u0, float u1, V3* p_o,
                                V3* err_o, V3* n_o, float* pdf_o) {
    V3 pc = V3{1.0f * 0.0f + 0.0f * 0.0f + 0.0f * 0.0f + cx, 0.0f * 0.0f + 1.0f * 0.0f + 0.0f * 0.0f + cy,
               0.0f * 0.0f + 0.0f * 0.0f + 1.0f * 0.0f + cz};
    V3 p_origin = offset_ray_origin(sf.p, sf.p_error, sf.n, pc - sf.p);
    if (len2(p_origin - pc) <= radius * radius) {
        // inside: Sphere::sample (sphere.rs:103-121), area pdf converted to solid angle
        float z = 1.0f - 2.0f * u0;
        float rr = __builtin_sqrtf(fmaxr(1.0f - z * z, 0.0f));
        float sp, cp;
        det_sincos(2.0f * kPi * u1, &sp, &cp);
        V3 obj = V3{rr * cp, rr * sp, z} * radius;
        V3 n = normalize(sphere_through(obj));
        obj = obj * (radius / length(obj));
        V3 oe = vabs(obj) * kGamma5;
        const float g3 = kGamma3;
        V3 p = V3{1.0f * obj.x + 0.0f * obj.y + 0.0f * obj.z + cx, 0.0f * obj.x + 1.0f * obj.y + 0.0f * obj.z + cy,
                  0.0f * obj.x + 0.0f * obj.y + 1.0f * obj.z + cz};
        V3 err;
        err.x = (g3 + 1.0f) * (__builtin_fabsf(1.0f * oe.x) + __builtin_fabsf(0.0f * oe.y) + __builtin_fabsf(0.0f * oe.z)) +
                g3 * (__builtin_fabsf(1.0f * obj.x) + __builtin_fabsf(0.0f * obj.y) + __builtin_fabsf(0.0f * obj.z) + __builtin_fabsf(cx));
        err.y = (g3 + 1.0f) * (__builtin_fabsf(0.0f * oe.x) + __builtin_fabsf(1.0f * oe.y) + __builtin_fabsf(0.0f * oe.z)) +
                g3 * (__builtin_fabsf(0.0f * obj.x) + __builtin_fabsf(1.0f * obj.y) + __builtin_fabsf(0.0f * obj.z) + __builtin_fabsf(cy));
        err.z = (g3 + 1.0f) * (__builtin_fabsf(0.0f * oe.x) + __builtin_fabsf(0.0f * oe.y) + __builtin_fabsf(1.0f * oe.z)) +
                g3 * (__builtin_fabsf(0.0f * obj.x) + __builtin_fabsf(0.0f * obj.y) + __builtin_fabsf(1.0f * obj.z) + __builtin_fabsf(cz));
        float pdf = 1.0f / sphere_area(radius);
        V3 wi = p - sf.p;
        if (len2(wi) == 0.0f) {
            pdf = 0.0f;
        } else {
            wi = normalize(wi);
            pdf *= len2(sf.p - p) / absdot(n, -wi);
        }
        if (__builtin_isinf(pdf)) pdf = 0.0f;
        *p_o = p;
        *err_o = err;
        *n_o = n;
        *pdf_o = pdf;
        return;
    }
    // outside: uniform sampling of the cone the sphere subtends (sphere.rs:140-178)
    float dc = length(sf.p - pc);
    float inv_dc = 1.0f / dc;
    V3 wc = (pc - sf.p) * inv_dc, wc_x, wc_y;
    coordinate_system(wc, &wc_x, &wc_y);
    float sin_theta_max = radius * inv_dc;
    float sin_theta_max2 = sin_theta_max * sin_theta_max;
    float inv_sin_theta_max = 1.0f / sin_theta_max;
    float cos_theta_max = __builtin_sqrtf(fmaxr(1.0f - sin_theta_max2, 0.0f));
    float cos_theta = (cos_theta_max - 1.0f) * u0 + 1.0f;
    float sin_theta2 = 1.0f - cos_theta * cos_theta;
    if (sin_theta_max2 < 0.00068523f) {
        sin_theta2 = sin_theta_max2 * u0;
        cos_theta = __builtin_sqrtf(1.0f - sin_theta2);
    }
    float cos_alpha = sin_theta2 * inv_sin_theta_max +
                      cos_theta * __builtin_sqrtf(fmaxr(1.0f - sin_theta2 * inv_sin_theta_max * inv_sin_theta_max, 0.0f));
    float sin_alpha = __builtin_sqrtf(fmaxr(1.0f - cos_alpha * cos_alpha, 0.0f));
    float sp, cp;
    det_sincos(u1 * 2.0f * kPi, &sp, &cp);
    V3 n_world = (-wc_x) * sin_alpha * cp + (-wc_y) * sin_alpha * sp + (-wc) * cos_alpha;
    V3 p_world = pc + n_world * radius;
    *p_o = p_world;
    *err_o = vabs(p_world) * kGamma5;
    *n_o = n_world;
    *pdf_o = 1.0f / (2.0f * kPi * (1.0f - cos_theta_max));
}

// Light::sample_li (light.rs:35-42): DiffuseAreaLight (diffuse.rs:60-81 with Triangle::sample / Shape::sample2)
// or InfiniteAreaLight (infinite.rs:96-129). Outputs the visibility tester's far point (p1, error, normal).
PB_DEV void light_sample_li(const ShadeConsts& sc, const Surf& sf, const DevLight& lt, float ul0, float ul1, V3* wi_o,
                            float* pdf_o, V3* li_o, V3* p1_o, V3* p1_err_o, V3* p1_n_o) {
    V3 wi = V3{0.0f, 0.0f, 0.0f};
    float light_pdf = 0.0f;
    V3 li = V3{0.0f, 0.0f, 0.0f};
    V3 p1 = V3{0.0f, 0.0f, 0.0f}, p1_err = V3{0.0f, 0.0f, 0.0f}, p1_n = V3{0.0f, 0.0f, 0.0f};
    V3 Lc = V3{lt.L[0], lt.L[1], lt.L[2]};
    if (lt.type == PBRT_LIGHT_DIFFUSE_AREA && sc.bvh.has_spheres &&
        (__float_as_int(sc.bvh.tris[3 * (size_t)lt.slot + 2].w) & kPrimSphere)) {
        float4 rec = sc.bvh.tris[3 * (size_t)lt.slot];
        float pdf;
        sphere_light_sample(rec.x, rec.y, rec.z, rec.w, sf, ul0, ul1, &p1, &p1_err, &p1_n, &pdf);
        // DiffuseAreaLight::sample_li (diffuse.rs:60-81)
        if (pdf == 0.0f || len2(p1 - sf.p) == 0.0f) {
            light_pdf = 0.0f;
        } else {
            light_pdf = pdf;
            wi = normalize(p1 - sf.p);
            if (lt.two_sided || dot(p1_n, -wi) > 0.0f) li = Lc;
        }
    } else if (lt.type == PBRT_LIGHT_DIFFUSE_AREA) {
        // Triangle::sample (triangle.rs:330-348) + Shape::sample2 (shape.rs:38-53)
        V3 q0, q1, q2;
        int fl;
        load_tri(sc.bvh.tris, lt.slot, &q0, &q1, &q2, &fl);
        float su0 = __builtin_sqrtf(ul0);
        float bx = 1.0f - su0, by = ul1 * su0;
        float bz = 1.0f - bx - by;
        p1 = q0 * bx + q1 * by + q2 * bz;
        p1_n = normalize(cross(q1 - q0, q2 - q0));
        if (sc.bvh.tri_shading && sc.bvh.has_normals) {  // Triangle::sample with mesh.n (triangle.rs:337-341)
            const float4* sh = sc.bvh.tri_shading + 6 * (size_t)lt.slot;
            float4 s0 = sh[0], s1 = sh[1], s2 = sh[2];
            V3 nsi = V3{s0.x, s0.y, s0.z} * bx + V3{s0.w, s1.x, s1.y} * by + V3{s1.z, s1.w, s2.x} * bz;
            if (dot(p1_n, nsi) < 0.0f) p1_n = -p1_n;
        }
        p1_err = (vabs(q0 * bx) + vabs(q1 * by) + vabs(q2 * bz)) * kGamma6;
        float pdf = 1.0f / lt.area;
        V3 w = p1 - sf.p;
        if (len2(w) == 0.0f) {
            pdf = 0.0f;
        } else {
            w = normalize(w);
            V3 dd = sf.p - p1;
            pdf *= len2(dd) / absdot(p1_n, -w);
            if (__builtin_isinf(pdf)) pdf = 0.0f;
        }
        // DiffuseAreaLight::sample_li (diffuse.rs:60-81)
        if (pdf == 0.0f || len2(p1 - sf.p) == 0.0f) {
            light_pdf = 0.0f;
        } else {
            light_pdf = pdf;
            wi = normalize(p1 - sf.p);
            if (lt.two_sided || dot(p1_n, -wi) > 0.0f) li = Lc;
        }
    } else if (lt.type == PBRT_LIGHT_POINT || lt.type == PBRT_LIGHT_SPOT) {
        // PointLight::sample_li (point.rs:47-63), SpotLight::sample_li (spot.rs:70-88)
        p1 = V3{lt.pos[0], lt.pos[1], lt.pos[2]};
        V3 dv = p1 - sf.p;
        wi = normalize(dv);
        light_pdf = 1.0f;
        float d2 = len2(dv);
        if (lt.type == PBRT_LIGHT_POINT) {
            li = Lc / d2;
        } else {
            // SpotLight::falloff (spot.rs:49-62; D52: wl normalised)
            V3 w = -wi;
            V3 wl = V3{lt.w2l[0] * w.x + lt.w2l[1] * w.y + lt.w2l[2] * w.z, lt.w2l[3] * w.x + lt.w2l[4] * w.y + lt.w2l[5] * w.z,
                       lt.w2l[6] * w.x + lt.w2l[7] * w.y + lt.w2l[8] * w.z};
            wl = normalize(wl);
            float cos_theta = wl.z, fall;
            if (cos_theta < lt.cos_total_width) {
                fall = 0.0f;
            } else if (cos_theta >= lt.cos_falloff_start) {
                fall = 1.0f;
            } else {
                float delta = (cos_theta - lt.cos_total_width) / (lt.cos_falloff_start - lt.cos_total_width);
                fall = (delta * delta) * (delta * delta);
            }
            li = Lc * fall / d2;
        }
    } else if (lt.type == PBRT_LIGHT_DISTANT) {
        // DistantLight::sample_li (distant.rs:54-74)
        wi = V3{lt.pos[0], lt.pos[1], lt.pos[2]};
        light_pdf = 1.0f;
        p1 = sf.p + wi * (2.0f * sc.world_radius);
        li = Lc;
    } else {
        // InfiniteAreaLight::sample_li (infinite.rs:96-129)
        float pdf1, pdf0;
        int v;
        float d1 = sample_continuous2(sc.env_marg_func, sc.env_marg_cdf, sc.env_marg_int, ul1, &pdf1, &v);
        int dummy;
        float d0 = sample_continuous2(sc.env_cond_func[v], sc.env_cond_cdf[v], sc.env_cond_int[v], ul0, &pdf0, &dummy);
        float map_pdf = pdf0 * pdf1;
        if (map_pdf != 0.0f) {
            float theta = d1 * kPi, phi = d0 * 2.0f * kPi;
            float st, ct, sp, cp;
            det_sincos(theta, &st, &ct);
            det_sincos(phi, &sp, &cp);
            wi = V3{st * cp, st * sp, ct};
            light_pdf = map_pdf / (2.0f * kPi * kPi * st);
            if (st == 0.0f) light_pdf = 0.0f;
            p1 = sf.p + wi * (2.0f * sc.world_radius);
            li = Lc;
        }
    }
    *wi_o = wi;
    *pdf_o = light_pdf;
    *li_o = li;
    *p1_o = p1;
    *p1_err_o = p1_err;
    *p1_n_o = p1_n;
}

// SpatialLightDistribution::compute_distribution (lightdistrib.rs:109-163; D57 / D53 intended) for every voxel:
// 128 radical-inverse points of the voxel, Light::sample_li from each, Distribution1D over sum(Li.y / pdf).
PB_DEV float radical_inverse_small(int base_index, uint32_t a) {  // lowdiscrepancy.rs:322-331 for the first five primes
    if (base_index == 0) {
        unsigned long long r = (unsigned long long)__brev(a) << 32;
        return fminr(kOneMinusEpsilon, (float)r * 5.4210108624275222e-20f);
    }
    const uint32_t primes[5] = {2u, 3u, 5u, 7u, 11u};
    return halton_radical_inverse(primes[base_index], nullptr, a);
}
__global__ void k_spatial_light_tables(ShadeConsts sc, float* __restrict__ table) {
    size_t voxel = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n_voxels = (size_t)sc.n_voxel[0] * sc.n_voxel[1] * sc.n_voxel[2];
    if (voxel >= n_voxels) return;
    int pi[3] = {(int)(voxel % sc.n_voxel[0]), (int)((voxel / sc.n_voxel[0]) % sc.n_voxel[1]),
                 (int)(voxel / ((size_t)sc.n_voxel[0] * sc.n_voxel[1]))};
    const float* mn = sc.bvh.root_min;
    const float* mx = sc.bvh.root_max;
    float vlo[3], vhi[3];
    for (int i = 0; i < 3; ++i) {
        float t0 = (float)pi[i] / (float)sc.n_voxel[i], t1 = (float)(pi[i] + 1) / (float)sc.n_voxel[i];
        float a = (1.0f - t0) * mn[i] + t0 * mx[i], b = (1.0f - t1) * mn[i] + t1 * mx[i];  // Bounds3::lerp
        vlo[i] = fminr(a, b);                                                                // Bounds3::from((p0, p1))
        vhi[i] = fmaxr(a, b);
    }
    const int n = sc.n_lights;
    float* func = table + voxel * (size_t)(2 * n + 2);
    float* cdf = func + n;
    for (int j = 0; j < n; ++j) func[j] = 0.0f;
    for (uint32_t i = 0; i < 128u; ++i) {
        float t[3] = {radical_inverse_small(0, i), radical_inverse_small(1, i), radical_inverse_small(2, i)};
        Surf sf;
        sf.n = V3{0.0f, 0.0f, 0.0f};        // BaseInteraction::new(po, Normal3f::default(), ..) (lightdistrib.rs:133-140)
        sf.p_error = V3{0.0f, 0.0f, 0.0f};
        sf.p = V3{(1.0f - t[0]) * vlo[0] + t[0] * vhi[0], (1.0f - t[1]) * vlo[1] + t[1] * vhi[1], (1.0f - t[2]) * vlo[2] + t[2] * vhi[2]};
        float u0 = radical_inverse_small(3, i), u1 = radical_inverse_small(4, i);
        for (int j = 0; j < n; ++j) {
            DevLight lt = sc.lights[j];
            V3 wi, li, p1, p1_err, p1_n;
            float pdf;
            light_sample_li(sc, sf, lt, u0, u1, &wi, &pdf, &li, &p1, &p1_err, &p1_n);
            if (pdf > 0.0f) func[j] += (0.212671f * li.x + 0.715160f * li.y + 0.072169f * li.z) / pdf;
        }
    }
    float sum = 0.0f;
    for (int j = 0; j < n; ++j) sum += func[j];
    float avg = sum / (float)(128 * n);
    float min_contrib = avg > 0.0f ? 0.001f * avg : 1.0f;
    for (int j = 0; j < n; ++j) func[j] = fmaxr(func[j], min_contrib);
    // Distribution1D::new (sampling.rs:69-95)
    cdf[0] = 0.0f;
    for (int j = 1; j < n + 1; ++j) cdf[j] = cdf[j - 1] + func[j - 1] / (float)n;
    float func_int = cdf[n];
    if (func_int == 0.0f) {
        for (int j = 1; j < n + 1; ++j) cdf[j] = (float)j / (float)n;
    } else {
        for (int j = 1; j < n + 1; ++j) cdf[j] /= func_int;
    }
    func[2 * n + 1] = func_int;
}

// estimate_direct (integrator.rs:136-266), first part: sample the light, evaluate the BSDF, sample the
// BSDF, evaluate the light pdf. Writes the shadow ray (slot 2), the MIS ray (slot 1) and the pending
// terms into the path state; returns PF_NEE_* flags for the rays that must be traced. `matte` = the
// BSDF has a non-specular (Lambertian) lobe with reflectance kd; otherwise f == 0 and nothing is emitted.
PB_DEV int estimate_direct_emit(const ShadeConsts& sc, const PathState& ps, uint32_t p, const Surf& sf, const Frame& fr,
                                bool matte, V3 kd, int light_num, float ul0, float ul1, float us0, float us1,
                                float pick_pdf, V3 beta) {
    if (!matte) return 0;
    V3 wo = sf.wo;
    DevLight lt = sc.lights[light_num];
    V3 wi, li, p1, p1_err, p1_n;
    float light_pdf;
    light_sample_li(sc, sf, lt, ul0, ul1, &wi, &light_pdf, &li, &p1, &p1_err, &p1_n);
    int nee_flags = 0;
    V3 A = V3{0.0f, 0.0f, 0.0f};
    if (light_pdf > 0.0f && !is_black(li)) {
        V3 f;
        float scattering_pdf;
        matte_f_pdf(fr, kd, wo, wi, &f, &scattering_pdf);
        f = f * absdot(wi, fr.ns);
        if (!is_black(f)) {
            // VisibilityTester::un_occluded -> spawn_ray_to (interaction.rs:147-153)
            V3 origin = offset_ray_origin(sf.p, sf.p_error, sf.n, p1 - sf.p);
            V3 target = offset_ray_origin(p1, p1_err, p1_n, origin - p1);
            V3 d = target - origin;
            store_ray(ps, p, RS_SHADOW, origin, d, 1.0f - kShadowEpsilon);
            if (lt.delta) {
                A = mulv(li, f) / light_pdf;  // integrator.rs:196-198: no MIS weight for a delta light
            } else {
                float weight = power_heuristic1(light_pdf, scattering_pdf);
                A = mulv(li, f) * weight / light_pdf;
            }
            nee_flags |= PF_NEE_SHADOW;
        }
    }
    // -- BSDF sampling half, only for non-delta lights (integrator.rs:207) --
    V3 wi2;
    float spdf = 0.0f;
    bool ok = false;
    V3 f2 = V3{0.0f, 0.0f, 0.0f};
    // scattering_pdf keeps the light-half value if wo.z == 0 (then it is 0 as well)
    if (!lt.delta) f2 = matte_sample_f(fr, kd, wo, us0, us1, &wi2, &spdf, &ok);
    if (ok) f2 = f2 * absdot(wi2, fr.ns);
    if (ok && !is_black(f2) && spdf > 0.0f) {
        float lpdf;
        if (lt.type == PBRT_LIGHT_DIFFUSE_AREA && sc.bvh.has_spheres &&
            (__float_as_int(sc.bvh.tris[3 * (size_t)lt.slot + 2].w) & kPrimSphere)) {
            // Sphere::pdf2 (sphere.rs:181-192)
            float4 rec = sc.bvh.tris[3 * (size_t)lt.slot];
            float radius = rec.w;
            V3 pc = V3{1.0f * 0.0f + 0.0f * 0.0f + 0.0f * 0.0f + rec.x, 0.0f * 0.0f + 1.0f * 0.0f + 0.0f * 0.0f + rec.y,
                       0.0f * 0.0f + 0.0f * 0.0f + 1.0f * 0.0f + rec.z};
            V3 p_origin = offset_ray_origin(sf.p, sf.p_error, sf.n, pc - sf.p);
            if (len2(p_origin - pc) < radius * radius) {
                // Shape::pdf2 (shape.rs:54-69): intersect the sphere along wi
                V3 o2 = offset_ray_origin(sf.p, sf.p_error, sf.n, wi2);
                TravRay rr{o2.x, o2.y, o2.z, wi2.x, wi2.y, wi2.z, kInf};
                float th, phi_h;
                V3 ph;
                if (!sphere_test(rec.x, rec.y, rec.z, radius, rr, kInf, &th, &ph, &phi_h)) {
                    lpdf = 0.0f;
                } else {
                    Surf hs = make_surface_sphere(sc.bvh, lt.slot, ph, wi2);
                    lpdf = len2(sf.p - hs.p) / (absdot(hs.n, -wi2) * sphere_area(radius));
                    if (__builtin_isinf(lpdf)) lpdf = 0.0f;
                }
            } else {
                float sin_theta_max2 = radius * radius / len2(sf.p - pc);
                float cos_theta_max = __builtin_sqrtf(fmaxr(1.0f - sin_theta_max2, 0.0f));
                lpdf = 1.0f / (2.0f * kPi * (1.0f - cos_theta_max));
            }
        } else if (lt.type == PBRT_LIGHT_DIFFUSE_AREA) {
            // Shape::pdf2 (shape.rs:54-69)
            V3 o2 = offset_ray_origin(sf.p, sf.p_error, sf.n, wi2);
            V3 ph, nh;
            if (!light_triangle_intersect(sc.bvh, lt.slot, o2, wi2, &ph, &nh)) {
                lpdf = 0.0f;
            } else {
                lpdf = len2(sf.p - ph) / (absdot(nh, -wi2) * lt.area);
                if (__builtin_isinf(lpdf)) lpdf = 0.0f;
            }
        } else {
            // InfiniteAreaLight::pdf_li (infinite.rs:140-151)
            float theta = det_acos(clampf(wi2.z, -1.0f, 1.0f));
            float ph = det_atan2(wi2.y, wi2.x);
            if (ph < 0.0f) ph = ph + 2.0f * kPi;
            float st = det_sin(theta);
            if (st == 0.0f) {
                lpdf = 0.0f;
            } else {
                int iu = (int)(ph * kInv2Pi * 2.0f);
                iu = iu < 0 ? 0 : (iu > 1 ? 1 : iu);
                int iv = (int)(theta * kInvPi * 2.0f);
                iv = iv < 0 ? 0 : (iv > 1 ? 1 : iv);
                lpdf = sc.env_cond_func[iv][iu] / sc.env_marg_int / (2.0f * kPi * kPi * st);
            }
        }
        if (lpdf != 0.0f) {
            float weight = power_heuristic1(spdf, lpdf);
            V3 o2 = offset_ray_origin(sf.p, sf.p_error, sf.n, wi2);
            store_ray(ps, p, RS_MIS, o2, wi2, kInf);
            ps.nee_f[p] = make_float4(f2.x, f2.y, f2.z, weight);
            nee_flags |= PF_NEE_MIS;
        }
    }
    if (nee_flags) {
        ps.nee_a[p] = make_float4(A.x, A.y, A.z, pick_pdf);
        ps.nee_b[p] = make_float4(beta.x, beta.y, beta.z, spdf);
        ps.nee_light[p] = light_num;
    }
    return nee_flags;
}

// estimate_direct, second part: combine the traced shadow / MIS results into Ld (before the division
// by the light-pick pdf). Also returns the pick pdf and the throughput stored with the estimate.
PB_DEV V3 estimate_direct_resolve(const ShadeConsts& sc, const PathState& ps, uint32_t p, int flags, float* pick_pdf,
                                  V3* beta_at_vertex) {
    float4 na = ps.nee_a[p], nf = ps.nee_f[p], nb = ps.nee_b[p];
    int light_id = ps.nee_light[p];
    V3 ld = V3{0.0f, 0.0f, 0.0f};
    if (flags & PF_NEE_SHADOW) {
        bool occluded = ps.hit[hit_index(ps, p, RS_SHADOW)].x != 0.0f;
        if (!occluded) ld = ld + V3{na.x, na.y, na.z};
    }
    if (flags & PF_NEE_MIS) {
        int hslot = __float_as_int(ps.hit[hit_index(ps, p, RS_MIS) + 1].x);
        float4 r0 = ps.ray[ray_index(ps, p, RS_MIS)], r1 = ps.ray[ray_index(ps, p, RS_MIS) + 1];
        V3 wi = V3{r0.w, r1.x, r1.y};
        DevLight lt = sc.lights[light_id];
        V3 li = V3{0.0f, 0.0f, 0.0f};
        if (hslot >= 0) {
            // D26 (intended): Le only when the hit primitive's area light is this light
            int hl = (__float_as_int(sc.bvh.tris[3 * (size_t)hslot + 2].w) & kPrimLightMask) - 1;
            if (hl == light_id) {
                float4 hb = ps.hit[hit_index(ps, p, RS_MIS)];
                V3 n = tri_interaction_normal(sc.bvh, hslot, hb.y, hb.z, hb.w);
                if (lt.two_sided || dot(n, -wi) > 0.0f) li = V3{lt.L[0], lt.L[1], lt.L[2]};
            }
        } else if (lt.type == PBRT_LIGHT_INFINITE) {
            li = V3{lt.L[0], lt.L[1], lt.L[2]};
        }
        if (!is_black(li)) {
            V3 f = V3{nf.x, nf.y, nf.z};
            ld = ld + mulv(li, f) * nf.w / nb.w;
        }
    }
    *pick_pdf = na.w;
    *beta_at_vertex = V3{nb.x, nb.y, nb.z};
    return ld;
}

// Perfect-specular lobes (reflection.rs:614-819). `which`: 0 = FresnelSpecular (glass with
// allow_multiple_lobes) or the mirror's SpecularReflection(FresnelNoOp); 1 = SpecularReflection lobe only
// (mirror: FresnelNoOp; glass: FresnelDielectric(1, eta)); 2 = SpecularTransmission lobe only (glass).
// Returns f (local), sets wi (local), pdf, transmission flag. pdf = 0 when nothing was sampled.
PB_DEV V3 sample_specular_local(const DevMaterial& mat, V3 kd, V3 kt, V3 wol, float ur, int which, V3* wil, float* pdf,
                                bool* transmission) {
    *pdf = 0.0f;
    *transmission = false;
    V3 zero = V3{0.0f, 0.0f, 0.0f};
    if (mat.type == PBRT_MAT_MIRROR) {
        if (which == 2 || is_black(kd)) return zero;
        *wil = V3{-wol.x, -wol.y, wol.z};
        *pdf = 1.0f;
        return mulv(kd, V3{1.0f, 1.0f, 1.0f}) / __builtin_fabsf(wil->z);
    }
    if (mat.type != PBRT_MAT_GLASS) return zero;
    if (which == 0) {
        // FresnelSpecular (reflection.rs:733-819), TransportMode::Radiance
        float F = fr_dielectric(wol.z, 1.0f, mat.eta);
        if (ur < F) {
            *wil = V3{-wol.x, -wol.y, wol.z};
            *pdf = F;
            return kd * F / __builtin_fabsf(wil->z);
        }
        bool entering = wol.z > 0.0f;
        float eta_i = entering ? 1.0f : mat.eta, eta_t = entering ? mat.eta : 1.0f;
        if (!refract(wol, faceforward(V3{0.0f, 0.0f, 1.0f}, wol), eta_i / eta_t, wil)) return zero;
        V3 ft = kt * (1.0f - F);
        ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
        *pdf = 1.0f - F;
        *transmission = true;
        return ft / __builtin_fabsf(wil->z);
    }
    if (which == 1) {
        // SpecularReflection with FresnelDielectric(1, eta) (reflection.rs:590-659)
        if (is_black(kd)) return zero;
        *wil = V3{-wol.x, -wol.y, wol.z};
        *pdf = 1.0f;
        float F = fr_dielectric(wil->z, 1.0f, mat.eta);
        return mulv(kd, V3{F, F, F}) / __builtin_fabsf(wil->z);
    }
    // SpecularTransmission (reflection.rs:661-731)
    if (is_black(kt)) return zero;
    bool entering = wol.z > 0.0f;
    float eta_i = entering ? 1.0f : mat.eta, eta_t = entering ? mat.eta : 1.0f;
    if (!refract(wol, faceforward(V3{0.0f, 0.0f, 1.0f}, wol), eta_i / eta_t, wil)) return zero;
    *pdf = 1.0f;
    *transmission = true;
    float F = fr_dielectric(wil->z, 1.0f, mat.eta);
    V3 ft = mulv(kt, V3{1.0f - F, 1.0f - F, 1.0f - F});
    ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
    return ft / __builtin_fabsf(wil->z);
}

__global__ void __launch_bounds__(256) k_shade(ShadeConsts sc, PathState ps, Queues qin, Queues qout, PassParams pp,
                                                 TileList tiles, uint32_t n_in) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n_in;
    uint32_t p = active ? qin.shade[i] : 0u;
    bool emit_cont = false, emit_mis = false, emit_shadow = false;

    uint32_t cell = 0;  // sort cell of the rays this path emits
    if (active) {
        float4 Lq = ps.L[p], bq = ps.beta[p];
        V3 L = V3{Lq.x, Lq.y, Lq.z};
        float eta_scale = Lq.w;
        V3 beta = V3{bq.x, bq.y, bq.z};
        int fb = __float_as_int(bq.w);
        int flags = fb & 0xff, bounces = fb >> 8;
        size_t rbase = ray_index(ps, p, RS_CONT), hbase = hit_index(ps, p, RS_CONT);

        // ---- (1) resolve the pending direct-lighting estimate (integrator.rs:136-266) ----
        if (flags & (PF_NEE_SHADOW | PF_NEE_MIS)) {
            float pick_pdf;
            V3 beta_v;
            V3 ld = estimate_direct_resolve(sc, ps, p, flags, &pick_pdf, &beta_v);
            ld = ld / pick_pdf;                 // integrator.rs:133
            L = L + mulv(beta_v, ld);           // path.rs:113-120
            flags &= ~(PF_NEE_SHADOW | PF_NEE_MIS);
        }

        // ---- (2) the continuation hit ----
        if (flags & PF_ALIVE) {
            flags &= ~PF_ALIVE;
            float4 r0 = ps.ray[rbase], r1 = ps.ray[rbase + 1];
            V3 rd = V3{r0.w, r1.x, r1.y};
            float4 h0 = ps.hit[hbase];
            float4 h1 = ps.hit[hbase + 1];
            int hslot = __float_as_int(h1.x);
            bool found = hslot >= 0;
            Surf sf;
            if (found) {
                sf = surface_from_hit(sc.bvh, hslot, __float_as_int(h1.y), h0.y, h0.z, h0.w, rd);
                if (qout.keys) cell = ray_sort_cell(sf.p.x, sf.p.y, sf.p.z, qout.key_lo, qout.key_inv);
            }
            // path.rs:80-88
            if (bounces == 0 || (flags & PF_SPECULAR_BOUNCE)) {
                if (found) {
                    L = L + mulv(beta, surface_le(sc, sf, -rd));
                } else {
                    for (int k = 0; k < sc.n_infinite; ++k) {
                        DevLight lt = sc.lights[sc.infinite_ids[k]];
                        L = L + mulv(beta, V3{lt.L[0], lt.L[1], lt.L[2]});
                    }
                }
            }
            if (found && bounces < pp.max_depth) {  // path.rs:90
                DevMaterial mat = sc.materials[sf.material];
                Samp sm = path_sampler(ps, pp, tiles, p);
                if (mat.type == PBRT_MAT_NONE) {
                    // path.rs:95-98: no BSDF -> continue through the surface, bounces unchanged
                    V3 o = offset_ray_origin(sf.p, sf.p_error, sf.n, rd);
                    store_ray(ps, p, RS_CONT, o, rd, kInf);
                    flags |= PF_ALIVE;
                    emit_cont = true;
                } else {
                    Frame fr = make_frame(sf);
                    V3 kd = V3{mat.kd[0], mat.kd[1], mat.kd[2]};
                    V3 kt = V3{mat.kt[0], mat.kt[1], mat.kt[2]};
                    V3 wo = -rd;  // path.rs:122 `let wo = -ray.d` (estimate_direct uses isect.wo = sf.wo)
                    bool has_lobe;  // which BxDFs the material adds: pbrt-v3 rules (matte / mirror / glass)
                    if (mat.type == PBRT_MAT_GLASS) has_lobe = !(is_black(kd) && is_black(kt));
                    else has_lobe = !is_black(kd);
                    bool nonspecular = (mat.type == PBRT_MAT_MATTE) && has_lobe;

                    // ---- uniform_sample_one_light (integrator.rs:92-134) ----
                    if (nonspecular && sc.n_lights > 0) {
                        float u_pick = samp_1d(pp, sm);
                        DevDistribution1D distrib = light_distribution_lookup(sc, sf.p);  // path.rs:115
                        int light_num = find_interval_cdf(distrib.cdf, distrib.n + 1, u_pick);
                        float pick_pdf = distrib.func_int > 0.0f ? distrib.func[light_num] / (distrib.func_int * (float)distrib.n)
                                                                 : 0.0f;
                        if (pick_pdf != 0.0f) {
                            float ul0, ul1, us0, us1;
                            samp_2d(pp, sm, &ul0, &ul1);
                            samp_2d(pp, sm, &us0, &us1);
                            int nee_flags = estimate_direct_emit(sc, ps, p, sf, fr, true, kd, light_num, ul0, ul1, us0, us1,
                                                                 pick_pdf, beta);
                            flags |= nee_flags;
                            emit_shadow = (nee_flags & PF_NEE_SHADOW) != 0;
                            emit_mis = (nee_flags & PF_NEE_MIS) != 0;
                        }
                    }

                    // ---- BSDF sampling for the next vertex (path.rs:123-152) ----
                    float u0, u1;
                    samp_2d(pp, sm, &u0, &u1);
                    V3 wi = V3{0.0f, 0.0f, 0.0f}, f = V3{0.0f, 0.0f, 0.0f};
                    float pdf = 0.0f;
                    bool sampled_specular = false, sampled_transmission = false;
                    if (has_lobe) {
                        if (mat.type == PBRT_MAT_MATTE) {
                            bool ok;
                            f = matte_sample_f(fr, kd, wo, u0, u1, &wi, &pdf, &ok);
                            if (!ok) pdf = 0.0f;
                        } else {
                            V3 wol = to_local(fr, wo);
                            float ur = fminr(u0 * 1.0f - 0.0f, kOneMinusEpsilon);
                            if (wol.z != 0.0f) {
                                V3 wil = V3{0.0f, 0.0f, 0.0f};
                                f = sample_specular_local(mat, kd, kt, wol, ur, 0, &wil, &pdf, &sampled_transmission);
                                sampled_specular = pdf != 0.0f;
                                if (pdf != 0.0f) wi = to_world(fr, wil);
                                else f = V3{0.0f, 0.0f, 0.0f};
                            }
                        }
                    }
                    if (!(is_black(f) || pdf == 0.0f)) {  // path.rs:136
                        beta = mulv(beta, f * (absdot(wi, fr.ns) / pdf));
                        flags = (flags & ~PF_SPECULAR_BOUNCE) | (sampled_specular ? PF_SPECULAR_BOUNCE : 0);
                        if (sampled_specular && sampled_transmission) {
                            float eta = mat.eta;
                            eta_scale *= (dot(wo, sf.n) > 0.0f) ? (eta * eta) : 1.0f / (eta * eta);
                        }
                        V3 o = offset_ray_origin(sf.p, sf.p_error, sf.n, wi);
                        bool alive = true;
                        // path.rs:200-207 Russian roulette (D27 intended)
                        V3 rr_beta = beta * eta_scale;
                        if (max_comp(rr_beta) < pp.rr_threshold && bounces > 3) {
                            float qq = fmaxr(0.05f, 1.0f - max_comp(rr_beta));
                            if (samp_1d(pp, sm) < qq) alive = false;
                            else beta = beta / (1.0f - qq);
                        }
                        if (alive) {
                            store_ray(ps, p, RS_CONT, o, wi, kInf);
                            flags |= PF_ALIVE;
                            emit_cont = true;
                            bounces += 1;
                        }
                    }
                }
                samp_store(ps, p, sm);
            }
        }
        ps.L[p] = make_float4(L.x, L.y, L.z, eta_scale);
        ps.beta[p] = make_float4(beta.x, beta.y, beta.z, __int_as_float((bounces << 8) | flags));
    }

    // ---- queue appends (block-aggregated) ----
    __shared__ BlockAppend sh;
    block_append(sh, qout, p, emit_cont, emit_mis, emit_shadow, emit_cont || emit_mis || emit_shadow, cell);
}

// -----------------------------------------------------------------------------------------------
// DirectLightingIntegrator::li (directlighting.rs:79-127) with specular_reflect / specular_transmit
// (integrator.rs:294-392). The reference recurses; here every path carries an explicit stack of
// vertices whose transmit branch is still to be followed (depth-first, the recursion's order, so the
// path's random stream is consumed in the reference's order). Per vertex the kernel walks a stage
// counter: light samples 0..total-1 (one estimate_direct per call when rays must be traced), then the
// reflect branch, then the transmit branch. L accumulates throughput * (Le + Ld).
// -----------------------------------------------------------------------------------------------
struct DirectState {
    int* stage;        // low 16 bits: stage at the current vertex; high 16 bits: frame stack height
    float4* ld_acc;    // estimate_direct sum over the samples of the current light
    float4* frames;    // [p * max_depth * 3 + k*3 + {0,1,2}]: (ray.d xyz, b0) (b1, b2, slot, depth) (T rgb, -)
    int light_strategy;  // 0 UniformSampleAll, 1 UniformSampleOne
    int mode;            // PBRT_INTEGRATOR_DIRECT, _WHITTED or _AO (the three share the vertex state machine)
    int ao_samples;      // AOIntegrator::n_samples
    int ao_cos_sample;   // AOIntegrator::cos_sample
};

#ifndef PB_DIRECT_WAVES
#define PB_DIRECT_WAVES 2  // 128 VGPRs: +5..9 % on direct lighting / Whitted / AO over the unconstrained 256-VGPR build
#endif
template <int MODE>  // PBRT_INTEGRATOR_DIRECT / _WHITTED / _AO: one instantiation each, the other integrators' stages compile away
__global__ void __launch_bounds__(256, PB_DIRECT_WAVES) k_shade_direct(ShadeConsts sc, PathState ps, DirectState ds, Queues qin,
                                                        Queues qout, PassParams pp, TileList tiles, uint32_t n_in) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n_in;
    uint32_t p = active ? qin.shade[i] : 0u;
    bool emit_cont = false, emit_mis = false, emit_shadow = false;
    if (active && !(__float_as_int(ps.beta[p].w) & PF_VALID)) active = false;  // placeholder path outside pixel_bounds

    if (active) {
        float4 Lq = ps.L[p], bq = ps.beta[p];
        V3 L = V3{Lq.x, Lq.y, Lq.z};
        V3 T = V3{bq.x, bq.y, bq.z};  // throughput of the current vertex
        int fb = __float_as_int(bq.w);
        int flags = fb & 0xff, depth = fb >> 8;
        int sg = ds.stage[p];
        int stage = sg & 0xffff, sp = sg >> 16;
        float4 accq = ds.ld_acc[p];
        V3 ld_acc = V3{accq.x, accq.y, accq.z};
        size_t rbase = ray_index(ps, p, RS_CONT), hbase = hit_index(ps, p, RS_CONT);
        constexpr int mode = MODE;
        const bool sample_all = mode == PBRT_INTEGRATOR_DIRECT && ds.light_strategy == 0;
        // stages at a vertex: direct = light samples, Whitted = one per light (whitted.rs:75), AO = hemisphere samples
        Samp sm = path_sampler(ps, pp, tiles, p);
        // uniform_sample_all_lights with a tabulating sampler (integrator.rs:55-89): the vertex takes its lights'
        // sample arrays while the max_depth requested sets last (sm.arr counts the arrays handed out); after that
        // every light gets ONE get_2d pair (bit 15 of sm.arr marks such a vertex).
        const bool tabulated = pp.smp.kind != PBRT_SAMPLER_RANDOM;
        bool fallback = (sm.arr & 0x8000) != 0;
        auto stages_at_vertex = [&]() {
            return (mode == PBRT_INTEGRATOR_AO)        ? ds.ao_samples
                   : (sc.n_lights == 0)                ? 0
                   : (mode == PBRT_INTEGRATOR_WHITTED) ? sc.n_lights
                   : !sample_all                       ? 1
                   : fallback                          ? sc.n_lights
                                                       : sc.total_light_samples;
        };
        int total = stages_at_vertex();
        bool have_vertex = !(flags & PF_ALIVE);  // ALIVE: a continuation ray was traced, its hit is a new vertex

        // finish the light sample whose rays were traced
        if (flags & (PF_NEE_SHADOW | PF_NEE_MIS)) {
            float pick_pdf;
            V3 beta_v;
            V3 ld = estimate_direct_resolve(sc, ps, p, flags, &pick_pdf, &beta_v);
            flags &= ~(PF_NEE_SHADOW | PF_NEE_MIS);
            if (sample_all) {
                ld_acc = ld_acc + ld;
            } else {
                L = L + mulv(T, ld / pick_pdf);
            }
        }
        // uniform_sample_all_lights (integrator.rs:44-90): close a light when its last sample is in
        auto close_light_if_done = [&](int st) {
            if (!sample_all || st == 0) return;
            if (fallback) {  // one sample per light, no division (integrator.rs:57-69)
                L = L + mulv(T, ld_acc);
                ld_acc = V3{0.0f, 0.0f, 0.0f};
                return;
            }
            // st = number of light samples finished so far at this vertex
            int lo = 0;
            while (sc.light_sample_prefix[lo + 1] < st) ++lo;  // light of sample st-1
            if (sc.light_sample_prefix[lo + 1] == st) {
                int ns = sc.light_sample_prefix[lo + 1] - sc.light_sample_prefix[lo];
                L = L + mulv(T, ld_acc / (float)ns);
                ld_acc = V3{0.0f, 0.0f, 0.0f};
            }
        };

        bool done = false;
        Surf sf;
        Frame fr;
        DevMaterial mat;
        V3 kd = V3{0, 0, 0}, kt = V3{0, 0, 0}, rd = V3{0, 0, 0};
        bool surface_ready = false;
        auto load_surface = [&]() {
            float4 r0 = ps.ray[rbase], r1 = ps.ray[rbase + 1];
            rd = V3{r0.w, r1.x, r1.y};
            float4 h0 = ps.hit[hbase], h1 = ps.hit[hbase + 1];
            int hslot = __float_as_int(h1.x);
            sf = surface_from_hit(sc.bvh, hslot, __float_as_int(h1.y), h0.y, h0.z, h0.w, rd);
            mat = sc.materials[sf.material];
            fr = make_frame(sf);
            kd = V3{mat.kd[0], mat.kd[1], mat.kd[2]};
            kt = V3{mat.kt[0], mat.kt[1], mat.kt[2]};
            surface_ready = true;
        };
        if (have_vertex && stage > 0 && stage <= total) close_light_if_done(stage);

        while (!done) {
            if (!have_vertex) {
                // ---- arrive at the hit of the continuation ray: directlighting.rs:86-106 ----
                flags &= ~PF_ALIVE;
                int hslot = __float_as_int(ps.hit[hbase + 1].x);
                if (hslot < 0) {
                    float4 r0 = ps.ray[rbase];
                    (void)r0;
                    // Σ light.le(ray): only infinite lights emit on a miss (AO: nothing, ao.rs:66)
                    for (int k = 0; k < sc.n_infinite && mode != PBRT_INTEGRATOR_AO; ++k) {
                        DevLight lt = sc.lights[sc.infinite_ids[k]];
                        L = L + mulv(T, V3{lt.L[0], lt.L[1], lt.L[2]});
                    }
                    stage = 0xfffe;  // nothing more at this vertex: unwind
                } else {
                    load_surface();
                    if (mat.type == PBRT_MAT_NONE) {
                        // directlighting.rs:97-104 (D28): continue through the surface at the same depth
                        V3 o = offset_ray_origin(sf.p, sf.p_error, sf.n, rd);
                        store_ray(ps, p, RS_CONT, o, rd, kInf);
                        flags |= PF_ALIVE;
                        emit_cont = true;
                        break;
                    }
                    if (mode != PBRT_INTEGRATOR_AO) L = L + mulv(T, surface_le(sc, sf, sf.wo));  // D28: + isect.Le(wo)
                    stage = 0;
                    if (sample_all && tabulated && sc.n_lights > 0) {
                        int handed = sm.arr & 0x7fff;
                        fallback = handed + 2 * sc.n_lights > pp.smp.n_arrays;
                        if (!fallback) handed += 2 * sc.n_lights;
                        sm.arr = handed | (fallback ? 0x8000 : 0);
                        total = stages_at_vertex();
                    }
                }
                have_vertex = true;
            }
            if (stage < total && mode == PBRT_INTEGRATOR_AO) {
                // ---- one hemisphere sample of AOIntegrator::li (ao.rs:73-99; D51: unoccluded directions count) ----
                if (!surface_ready) load_surface();
                V3 n = dot(sf.n, -rd) < 0.0f ? -sf.n : sf.n;  // face_forward(isect.n, -ray.d), D46 intended
                V3 s = normalize(sf.dpdu);
                V3 t = cross(sf.n, s);
                float u0, u1;
                if (tabulated)
                    samp_array_2d(pp, sm, 0, stage, &u0, &u1);  // ao.rs:77-81: the one requested array
                else
                    samp_2d(pp, sm, &u0, &u1);
                V3 wl;
                float pdf;
                if (ds.ao_cos_sample) {
                    wl = cosine_sample_hemisphere(u0, u1);
                    pdf = __builtin_fabsf(wl.z) * kInvPi;
                } else {
                    float r = __builtin_sqrtf(fmaxr(1.0f - u0 * u0, 0.0f));
                    float sp_, cp_;
                    det_sincos(2.0f * kPi * u1, &sp_, &cp_);
                    wl = V3{r * cp_, r * sp_, u0};
                    pdf = kInv2Pi;
                }
                V3 wi = V3{s.x * wl.x + t.x * wl.y + n.x * wl.z, s.y * wl.x + t.y * wl.y + n.y * wl.z,
                           s.z * wl.x + t.z * wl.y + n.z * wl.z};
                float a = dot(wi, n) / (pdf * (float)ds.ao_samples);
                V3 o = offset_ray_origin(sf.p, sf.p_error, sf.n, wi);
                store_ray(ps, p, RS_SHADOW, o, wi, kInf);
                ps.nee_a[p] = make_float4(a, a, a, 1.0f);
                flags |= PF_NEE_SHADOW;
                emit_shadow = true;
                stage += 1;
                break;
            }
            if (stage < total && mode == PBRT_INTEGRATOR_WHITTED) {
                // ---- one light of WhittedIntegrator::li's loop (whitted.rs:75-91) ----
                if (!surface_ready) load_surface();
                DevLight lt = sc.lights[stage];
                float ul0, ul1;
                samp_2d(pp, sm, &ul0, &ul1);
                stage += 1;
                V3 wi, li, p1, p1_err, p1_n;
                float pdf;
                light_sample_li(sc, sf, lt, ul0, ul1, &wi, &pdf, &li, &p1, &p1_err, &p1_n);
                if (is_black(li) || pdf == 0.0f) continue;
                V3 f = V3{0.0f, 0.0f, 0.0f};
                float spdf;
                if (mat.type == PBRT_MAT_MATTE && !is_black(kd)) matte_f_pdf(fr, kd, sf.wo, wi, &f, &spdf);  // BSDF::f, all lobes
                if (is_black(f)) continue;
                V3 origin = offset_ray_origin(sf.p, sf.p_error, sf.n, p1 - sf.p);
                V3 target = offset_ray_origin(p1, p1_err, p1_n, origin - p1);
                store_ray(ps, p, RS_SHADOW, origin, target - origin, 1.0f - kShadowEpsilon);
                V3 A = mulv(f, li) * absdot(wi, fr.ns) / pdf;
                ps.nee_a[p] = make_float4(A.x, A.y, A.z, 1.0f);
                flags |= PF_NEE_SHADOW;
                emit_shadow = true;
                break;
            }
            if (stage < total) {
                // ---- one estimate_direct ----
                if (!surface_ready) load_surface();
                bool matte = (mat.type == PBRT_MAT_MATTE) && !is_black(kd);
                int light_num;
                float pick_pdf = 1.0f;
                if (sample_all && fallback) {
                    light_num = stage;
                } else if (sample_all) {
                    light_num = 0;
                    while (sc.light_sample_prefix[light_num + 1] <= stage) ++light_num;
                } else {
                    // uniform_sample_one_light without a distribution (integrator.rs:113-117)
                    float nl = (float)sc.n_lights;
                    light_num = (int)fminr(samp_1d(pp, sm) * nl, nl - 1.0f);
                    pick_pdf = 1.0f / nl;
                }
                float ul0, ul1, us0, us1;
                if (sample_all && tabulated && !fallback) {
                    int k = stage - sc.light_sample_prefix[light_num];
                    int first = (sm.arr & 0x7fff) - 2 * sc.n_lights + 2 * light_num;  // this vertex's arrays of this light
                    samp_array_2d(pp, sm, first, k, &ul0, &ul1);
                    samp_array_2d(pp, sm, first + 1, k, &us0, &us1);
                } else {
                    samp_2d(pp, sm, &ul0, &ul1);
                    samp_2d(pp, sm, &us0, &us1);
                }
                int nee_flags = estimate_direct_emit(sc, ps, p, sf, fr, matte, kd, light_num, ul0, ul1, us0, us1, pick_pdf, T);
                stage += 1;
                if (nee_flags) {
                    flags |= nee_flags;
                    emit_shadow = (nee_flags & PF_NEE_SHADOW) != 0;
                    emit_mis = (nee_flags & PF_NEE_MIS) != 0;
                    break;  // trace, then come back to this vertex
                }
                close_light_if_done(stage);  // the sample contributed nothing
                continue;
            }
            // ---- specular branches (directlighting.rs:121-125), only while depth + 1 < max_depth ----
            bool branched = false;
            if ((stage == total || stage == total + 1) && depth + 1 < pp.max_depth && mode != PBRT_INTEGRATOR_AO) {
                if (!surface_ready) load_surface();
                for (; stage <= total + 1 && !branched; ++stage) {
                    int which = (stage == total) ? 1 : 2;  // reflect first, then transmit
                    float u0, u1;
                    samp_2d(pp, sm, &u0, &u1);
                    (void)u1;
                    V3 wol = to_local(fr, sf.wo);
                    // BSDF::sample_f with type = REFLECTION|SPECULAR or TRANSMISSION|SPECULAR (one matching lobe)
                    bool lobe = (mat.type == PBRT_MAT_MIRROR && which == 1 && !is_black(kd)) ||
                                (mat.type == PBRT_MAT_GLASS && ((which == 1 && !is_black(kd)) || (which == 2 && !is_black(kt))));
                    if (!lobe || wol.z == 0.0f) continue;
                    float ur = fminr(u0 * 1.0f - 0.0f, kOneMinusEpsilon);
                    V3 wil = V3{0, 0, 0};
                    float pdf;
                    bool tr;
                    V3 f = sample_specular_local(mat, kd, kt, wol, ur, which, &wil, &pdf, &tr);
                    if (pdf == 0.0f) continue;
                    V3 wi = to_world(fr, wil);
                    float ad = absdot(wi, fr.ns);
                    if (!(pdf > 0.0f && !is_black(f) && ad != 0.0f)) continue;  // integrator.rs:316
                    if (which == 1) {
                        // remember this vertex: its transmit branch runs after the reflected subtree
                        size_t fi = ((size_t)p * pp.max_depth + sp) * 3;
                        float4 h0 = ps.hit[hbase];
                        ds.frames[fi] = make_float4(rd.x, rd.y, rd.z, h0.y);
                        float4 h1 = ps.hit[hbase + 1];
                        ds.frames[fi + 1] = make_float4(h0.z, h0.w, h1.x, __int_as_float(depth));
                        ds.frames[fi + 2] = make_float4(T.x, T.y, T.z, h1.y);
                        sp += 1;
                    }
                    T = mulv(T, f * (ad / pdf));
                    depth += 1;
                    V3 o = offset_ray_origin(sf.p, sf.p_error, sf.n, wi);
                    store_ray(ps, p, RS_CONT, o, wi, kInf);
                    flags |= PF_ALIVE;
                    emit_cont = true;
                    branched = true;
                }
            }
            if (branched) break;
            // ---- vertex finished: unwind to the innermost vertex that still owes its transmit branch ----
            if (sp == 0) {
                done = true;
                break;
            }
            sp -= 1;
            size_t fi = ((size_t)p * pp.max_depth + sp) * 3;
            float4 f0 = ds.frames[fi], f1 = ds.frames[fi + 1], f2 = ds.frames[fi + 2];
            // restore the vertex into the continuation slot so the surface can be rebuilt
            float4 r0 = ps.ray[rbase];
            ps.ray[rbase] = make_float4(r0.x, r0.y, r0.z, f0.x);
            ps.ray[rbase + 1] = make_float4(f0.y, f0.z, kInf, 0.0f);
            ps.hit[hbase] = make_float4(0.0f, f0.w, f1.x, f1.y);
            ps.hit[hbase + 1] = make_float4(f1.z, f2.w, 0.0f, 0.0f);
            depth = __float_as_int(f1.w);
            T = V3{f2.x, f2.y, f2.z};
            stage = total + 1;  // transmit branch
            surface_ready = false;
            have_vertex = true;
        }
        samp_store(ps, p, sm);
        ps.L[p] = make_float4(L.x, L.y, L.z, 1.0f);
        ps.beta[p] = make_float4(T.x, T.y, T.z, __int_as_float((depth << 8) | flags));
        ds.stage[p] = (stage & 0xffff) | (sp << 16);
        ds.ld_acc[p] = make_float4(ld_acc.x, ld_acc.y, ld_acc.z, 0.0f);
    }

    __shared__ BlockAppend sh;
    block_append(sh, qout, p, emit_cont, emit_mis, emit_shadow, emit_cont || emit_mis || emit_shadow);
}

// FilmTile::add_sample's first statement (film.rs:253-255): scale the sample down to Film::max_sample_luminance
PB_DEV V3 clamp_sample_luminance(V3 L, float max_lum) {
    float y = 0.212671f * L.x + 0.715160f * L.y + 0.072169f * L.z;
    if (y > max_lum) L = L * (max_lum / y);
    return L;
}

// ---- film: FilmTile::add_sample (film.rs:252-295) with the 0.5 box filter, samples summed in order ----
__global__ void k_film_accumulate(PathState ps, PassParams pp, TileList tiles, float4* accum, float* d_film) {
    uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (uint32_t)pp.n_pix) return;
    int tile = pix >> 8, within = pix & 255;
    int2 org = tiles.origin[tile];
    int x = org.x + (within & 15), y = org.y + (within >> 4);
    if (!(x >= pp.x0 && x < pp.x1 && y >= pp.y0 && y < pp.y1)) return;
    float4 acc = accum[pix];
    for (int s = 0; s < pp.n_samples; ++s) {
        uint32_t p = (uint32_t)s * pp.n_pix + pix;
        float4 Lq = ps.L[p];
        V3 L = V3{Lq.x, Lq.y, Lq.z};
        float yv = 0.212671f * L.x + 0.715160f * L.y + 0.072169f * L.z;
        // integrator.rs:455 (D23 intended: is_infinite)
        if (__builtin_isnan(L.x) || __builtin_isnan(L.y) || __builtin_isnan(L.z) || yv < -1e-5f || __builtin_isinf(yv))
            L = V3{0.0f, 0.0f, 0.0f};
        L = clamp_sample_luminance(L, pp.max_sample_luminance);
        float2 pf = ps.pfilm[p];
        float dx = pf.x - 0.5f, dy = pf.y - 0.5f;
        int px0 = max((int)__builtin_ceilf(dx - 0.5f), 0), py0 = max((int)__builtin_ceilf(dy - 0.5f), 0);
        int px1 = min((int)__builtin_floorf(dx + 0.5f) + 1, pp.width), py1 = min((int)__builtin_floorf(dy + 0.5f) + 1, pp.height);
        for (int yy = py0; yy < py1; ++yy)
            for (int xx = px0; xx < px1; ++xx) {
                V3 c = L * 1.0f * 1.0f;  // l * sample_weight * filter_weight (box filter table is all ones)
                if (xx == x && yy == y) {
                    acc.x += c.x;
                    acc.y += c.y;
                    acc.z += c.z;
                    acc.w += 1.0f;
                } else {
                    // a film offset of exactly 0.0 also lands on the previous pixel (ceil in add_sample);
                    // that pixel may belong to another tile / GPU: add its XYZ directly to the film
                    float* fp = d_film + ((size_t)yy * pp.width + xx) * 4;
                    atomicAdd(fp + 0, 0.412453f * c.x + 0.357580f * c.y + 0.180423f * c.z);
                    atomicAdd(fp + 1, 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z);
                    atomicAdd(fp + 2, 0.019334f * c.x + 0.119193f * c.y + 0.950227f * c.z);
                    atomicAdd(fp + 3, 1.0f);
                }
            }
    }
    accum[pix] = acc;
}

// General reconstruction filter: FilmTile::add_sample (film.rs:252-295) as a scatter. Each sample adds
// L * weight * filter and the filter weight to every pixel of its footprint with float atomics (the
// footprints of neighbouring samples, tiles and GPUs overlap); XYZ conversion is linear, so it is applied
// per contribution instead of per tile (film.rs:111-123).
__global__ void k_film_splat(PathState ps, PassParams pp, TileList tiles, float* d_film) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (uint32_t)pp.n_pix * pp.n_samples) return;
    if (!(__float_as_int(ps.beta[p].w) & PF_VALID)) return;
    float4 Lq = ps.L[p];
    V3 L = V3{Lq.x, Lq.y, Lq.z};
    float yv = 0.212671f * L.x + 0.715160f * L.y + 0.072169f * L.z;
    if (__builtin_isnan(L.x) || __builtin_isnan(L.y) || __builtin_isnan(L.z) || yv < -1e-5f || __builtin_isinf(yv))
        L = V3{0.0f, 0.0f, 0.0f};
    L = clamp_sample_luminance(L, pp.max_sample_luminance);
    float2 pf = ps.pfilm[p];
    float dx = pf.x - 0.5f, dy = pf.y - 0.5f;
    int px0 = max((int)__builtin_ceilf(dx - pp.filter_rx), 0), py0 = max((int)__builtin_ceilf(dy - pp.filter_ry), 0);
    int px1 = min((int)__builtin_floorf(dx + pp.filter_rx) + 1, pp.width);
    int py1 = min((int)__builtin_floorf(dy + pp.filter_ry) + 1, pp.height);
    const float inv_rx = 1.0f / pp.filter_rx, inv_ry = 1.0f / pp.filter_ry;
    for (int yy = py0; yy < py1; ++yy) {
        float fy = __builtin_fabsf(((float)yy - dy) * inv_ry * 16.0f);
        int ify = min(15, (int)__builtin_floorf(fy));
        for (int xx = px0; xx < px1; ++xx) {
            float fx = __builtin_fabsf(((float)xx - dx) * inv_rx * 16.0f);
            int ifx = min(15, (int)__builtin_floorf(fx));
            float fw = pp.filter_table[ify * 16 + ifx];
            V3 c = L * 1.0f * fw;
            float* fp = d_film + ((size_t)yy * pp.width + xx) * 4;
            atomicAdd(fp + 0, 0.412453f * c.x + 0.357580f * c.y + 0.180423f * c.z);
            atomicAdd(fp + 1, 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z);
            atomicAdd(fp + 2, 0.019334f * c.x + 0.119193f * c.y + 0.950227f * c.z);
            atomicAdd(fp + 3, fw);
        }
    }
}

// Film::merge_film_tile (film.rs:111-123): contrib_sum -> XYZ, accumulated into the film
__global__ void k_film_merge(PassParams pp, TileList tiles, const float4* accum, float* d_film) {
    uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (uint32_t)pp.n_pix) return;
    int tile = pix >> 8, within = pix & 255;
    int2 org = tiles.origin[tile];
    int x = org.x + (within & 15), y = org.y + (within >> 4);
    if (!(x >= pp.x0 && x < pp.x1 && y >= pp.y0 && y < pp.y1)) return;
    float4 a = accum[pix];
    float* fp = d_film + ((size_t)y * pp.width + x) * 4;
    atomicAdd(fp + 0, 0.412453f * a.x + 0.357580f * a.y + 0.180423f * a.z);
    atomicAdd(fp + 1, 0.212671f * a.x + 0.715160f * a.y + 0.072169f * a.z);
    atomicAdd(fp + 2, 0.019334f * a.x + 0.119193f * a.y + 0.950227f * a.z);
    atomicAdd(fp + 3, a.w);
}

}  // namespace pb

int wavefront_render(PbrtHipScene* s, const PbrtCamera& camera, const PbrtRenderParams& rp, float* d_film,
                     PbrtRenderStats* stats);
