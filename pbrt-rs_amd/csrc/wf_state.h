// wf_state.h — path state in HBM, ray / shade queues and their block-aggregated append, camera and tile records (part of wavefront.h)
#pragma once
#include <hip/hip_runtime.h>

#include "scene.h"
#include "trace.h"
#include "trace_persistent.h"
#include "trace_wide.h"
#include "trace_stackless.h"

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
// A trace-queue entry is path << 2 | what: the three ray slots, and RS_MIS_BOOL = the path's MIS ray (slot RS_MIS) of which
// only `found` is wanted. estimate_direct (integrator.rs:232-262) calls scene.intersect for the BSDF-sampled direction and
// then looks at the hit only if the light is an area light (`light_isect.primitive.get_area_light() == light`); for an
// infinite light everything that follows depends on `found_surface_interaction` alone. BVHAccel::intersect finds its first
// hit under the ray's original t_max, walking exactly as intersect_p walks, so `found` = "some leaf whose box passes holds a
// triangle that Triangle::intersect accepts" — a boolean that does not depend on the visiting order and needs no closest
// hit: such a ray is traced as an any-hit ray that skips the triangles Triangle::intersect rejects (IO::strict), ends at its
// first hit, and sorts with the shadow rays (+2.3 % of a config-3 frame, +4.3 % of config 5's: profiles/r04_wide_kernel_ladder.txt).
constexpr uint32_t RS_MIS_BOOL = 3;
// estimate_direct_emit's return value carries this beside the PF_NEE_* bits (it is not a path flag)
constexpr int NEE_MIS_BOOL = 0x100;

struct PathState {
    float4* ray;     // [ray_index(p, slot) + k]: (o.xyz, d.x) (d.yz, t_max, -)
    // [hit_index(p, slot)]: (leaf slot of the hit or -1 as int bits, b0, b1, b2) — t is not kept: nothing downstream reads it
    // (the hit point comes from the barycentrics, triangle.rs:160-178) and ONE 16-B store per closest-hit ray is one memory
    // instruction less on the traversal kernel's refill path; [.. + 1]: (instance slot, -, -, -), written and read in two-level
    // scenes only; shadow slot: .x = occluded
    float4* hit;
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
    int two_level;   // the scene has instances: hit records carry the instance slot in their second float4
};
// the instance slot of a hit record (-1: none)
PB_DEV int hit_instance(const PathState& ps, size_t hit_base) { return ps.two_level ? __float_as_int(ps.hit[hit_base + 1].x) : -1; }

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
    int group_shift;   // path layout: a wave of 64 paths = (64 >> group_shift) consecutive pixels x (1 << group_shift) consecutive
                       // samples of each (0: 64 pixels of one sample index; 6: the 64 samples of one pixel); n_samples is a
                       // multiple of 1 << group_shift (path_to_sample_pixel / sample_pixel_to_path)
    int sample0;       // first sample index of this pass
    int spp;           // total samples per pixel (RNG keying)
    int width, height;
    int x0, y0, x1, y1;
    int sb_x0, sb_y0, sb_w;  // Film::get_sample_bounds of the whole film: origin and width (pixel_number)
    uint64_t seed;
    int max_depth;
    float rr_threshold;
    int light_strategy;
    float filter_rx, filter_ry;       // reconstruction filter radius
    const float* filter_table;        // 16 x 16 table (device), nullptr = 0.5 box (exact in-order path)
    float max_sample_luminance;       // Film::max_sample_luminance (film.rs:24), +inf = no clamp
    // pbrt_hip_li: the caller's stream key of every path (RNG::set_sequence argument); nullptr = the (pixel, sample) keys
    const uint64_t* stream_keys;
};

// path number <-> (sample of the pass, pixel of the tile set). Paths are laid out wave by wave: wave w holds the pixel block
// b = w % (n_pix / Pw) (Pw = 64 >> group_shift consecutive pixels) and the sample group g = w / (n_pix / Pw) (G = 1 << group_shift
// consecutive samples); lane l of it is pixel b Pw + l % Pw, sample g G + l / Pw. group_shift 0 is path = sample * n_pix + pixel.
PB_DEV void path_to_sample_pixel(const PassParams& pp, uint32_t p, int* s_local, int* pix) {
    const uint32_t sh = (uint32_t)pp.group_shift, pw_bits = 6u - sh;
    const uint32_t w = p >> 6, l = p & 63u, blocks = (uint32_t)pp.n_pix >> pw_bits;
    const uint32_t g = w / blocks, b = w - g * blocks;
    *pix = (int)((b << pw_bits) | (l & ((1u << pw_bits) - 1u)));
    *s_local = (int)((g << sh) | (l >> pw_bits));
}
PB_DEV uint32_t sample_pixel_to_path(const PassParams& pp, int s_local, uint32_t pix) {
    const uint32_t sh = (uint32_t)pp.group_shift, pw_bits = 6u - sh;
    const uint32_t g = (uint32_t)s_local >> sh, b = pix >> pw_bits, blocks = (uint32_t)pp.n_pix >> pw_bits;
    const uint32_t l = (((uint32_t)s_local & ((1u << sh) - 1u)) << pw_bits) | (pix & ((1u << pw_bits) - 1u));
    return ((g * blocks + b) << 6) | l;
}

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
constexpr int kSortKeyBits = 3 * PB_SORT_AXIS_BITS + 1;
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

// The pixel's number over the film's sample bounds (film.rs:76-81): with the 0.5 box filter that is y * width + x;
// with a wider filter the pixels sampled left of / above the film get numbers of their own ((width, y) and (0, y + 1)
// would share y * width + x, and with it their random streams).
PB_DEV int64_t pixel_number(const PassParams& pp, int x, int y) {
    return (int64_t)(y - pp.sb_y0) * pp.sb_w + (x - pp.sb_x0);
}
PB_DEV uint64_t sample_sequence(const PassParams& pp, int x, int y, int s) {
    return pp.seed ^ (uint64_t)(pixel_number(pp, x, y) * (int64_t)pp.spp + s);
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
// cell = ray_sort_cell of the point the path's rays leave from (used only when q.keys is set). mis_bool: the MIS ray is
// queued as RS_MIS_BOOL.
PB_DEV void block_append(BlockAppend& sh, const Queues& q, uint32_t p, bool cont, bool mis, bool shadow, bool again,
                         uint32_t cell = 0, bool mis_bool = false) {
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
    if (mis) q.trace[im] = p * 4u + (mis_bool ? RS_MIS_BOOL : (uint32_t)RS_MIS);
    if (shadow) q.trace[is] = p * 4u + RS_SHADOW;
    if (q.keys) {  // the three rays leave from the same surface point: one cell, the any-hit flag on top
        if (cont) q.keys[ic] = cell;
        if (mis) q.keys[im] = mis_bool ? (cell | (1u << (kSortKeyBits - 1))) : cell;
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

}  // namespace pb
