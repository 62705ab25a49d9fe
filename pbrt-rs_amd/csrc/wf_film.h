// wf_film.h — film kernels: in-order box accumulation, filter splat, merge (part of wavefront.h)
#pragma once
#include "wf_direct.h"

namespace pb {

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
    // one sample into the pixel (or, on a pixel border, also into the neighbour): in sample order, as the reference's loop adds them
    auto add_sample = [&](const float4 Lq, const float2 pf) {
        V3 L = V3{Lq.x, Lq.y, Lq.z};
        float yv = 0.212671f * L.x + 0.715160f * L.y + 0.072169f * L.z;
        // integrator.rs:455 (D23 intended: is_infinite)
        if (__builtin_isnan(L.x) || __builtin_isnan(L.y) || __builtin_isnan(L.z) || yv < -1e-5f || __builtin_isinf(yv))
            L = V3{0.0f, 0.0f, 0.0f};
        L = clamp_sample_luminance(L, pp.max_sample_luminance);
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
    };
    // Four samples are fetched before the first is used: under PbrtRenderParams.samples_per_wave a pixel's consecutive samples
    // are neighbours in memory (at 64 per wave: consecutive paths, four to a 64-byte line), and a line asked for four times in a
    // row is fetched once; one sample at a time the line is gone from L1 before its turn comes again (6.4 ms instead of 0.6).
    int s = 0;
    for (; s + 4 <= pp.n_samples; s += 4) {
        float4 Lq[4];
        float2 pf[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t p = sample_pixel_to_path(pp, s + k, pix);
            Lq[k] = ps.L[p];
            pf[k] = ps.pfilm[p];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) add_sample(Lq[k], pf[k]);
    }
    for (; s < pp.n_samples; ++s) {
        const uint32_t p = sample_pixel_to_path(pp, s, pix);
        add_sample(ps.L[p], ps.pfilm[p]);
    }
    accum[pix] = acc;
}

// The same accumulation where a wave of paths is the 64 samples of ONE pixel (PassParams::group_shift == 6, the default layout):
// sixteen lanes serve a pixel and fetch sixteen consecutive samples in one coalesced piece (a wave = four pixels x 256 bytes per
// load instruction, as line-efficient as the sample-major layout was for k_film_accumulate); every lane forms its own sample's
// contribution, then the sixteen are added to the pixel IN SAMPLE ORDER — lane after lane through the row — because float addition
// is not associative and the reference's loop (and the oracle's) adds them one after the other.
__global__ void k_film_accumulate_rows(PathState ps, PassParams pp, TileList tiles, float4* accum, float* d_film) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pix = t >> 4, k = t & 15u;
    if (pix >= (uint32_t)pp.n_pix) return;  // (whole rows of sixteen lanes leave together)
    int tile = pix >> 8, within = pix & 255;
    int2 org = tiles.origin[tile];
    int x = org.x + (within & 15), y = org.y + (within >> 4);
    if (!(x >= pp.x0 && x < pp.x1 && y >= pp.y0 && y < pp.y1)) return;
    const int row0 = (int)(threadIdx.x & 63u & ~15u);
    float4 acc = accum[pix];
    for (int s0 = 0; s0 < pp.n_samples; s0 += 16) {
        const uint32_t p = sample_pixel_to_path(pp, s0 + (int)k, pix);
        const float4 Lq = ps.L[p];
        const float2 pf = ps.pfilm[p];
        V3 L = V3{Lq.x, Lq.y, Lq.z};
        float yv = 0.212671f * L.x + 0.715160f * L.y + 0.072169f * L.z;
        // integrator.rs:455 (D23 intended: is_infinite)
        if (__builtin_isnan(L.x) || __builtin_isnan(L.y) || __builtin_isnan(L.z) || yv < -1e-5f || __builtin_isinf(yv))
            L = V3{0.0f, 0.0f, 0.0f};
        L = clamp_sample_luminance(L, pp.max_sample_luminance);
        float dx = pf.x - 0.5f, dy = pf.y - 0.5f;
        int px0 = max((int)__builtin_ceilf(dx - 0.5f), 0), py0 = max((int)__builtin_ceilf(dy - 0.5f), 0);
        int px1 = min((int)__builtin_floorf(dx + 0.5f) + 1, pp.width), py1 = min((int)__builtin_floorf(dy + 0.5f) + 1, pp.height);
        float own = 0.0f;  // the sample's weight in its own pixel (1 unless its footprint misses it: a film position outside the pixel)
        for (int yy = py0; yy < py1; ++yy)
            for (int xx = px0; xx < px1; ++xx) {
                if (xx == x && yy == y) {
                    own = 1.0f;
                } else {  // a sample on a pixel border also lands on the neighbour, which may belong to another tile / GPU
                    float* fp = d_film + ((size_t)yy * pp.width + xx) * 4;
                    atomicAdd(fp + 0, 0.412453f * L.x + 0.357580f * L.y + 0.180423f * L.z);
                    atomicAdd(fp + 1, 0.212671f * L.x + 0.715160f * L.y + 0.072169f * L.z);
                    atomicAdd(fp + 2, 0.019334f * L.x + 0.119193f * L.y + 0.950227f * L.z);
                    atomicAdd(fp + 3, 1.0f);
                }
            }
        for (int j = 0; j < 16; ++j) {  // sample s0 + j: every lane of the row adds it to its copy of the pixel, in order
            const float cx = __shfl(L.x, row0 + j, 64), cy = __shfl(L.y, row0 + j, 64), cz = __shfl(L.z, row0 + j, 64);
            const float w = __shfl(own, row0 + j, 64);
            if (w != 0.0f) {
                acc.x += cx;
                acc.y += cy;
                acc.z += cz;
                acc.w += 1.0f;
            }
        }
    }
    if (k == 0) accum[pix] = acc;
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
