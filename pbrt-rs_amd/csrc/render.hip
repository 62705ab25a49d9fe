// render.hip — Integrator::render behind the C ABI: pbrt_hip_render / pbrt_hip_render_device
// (SamplerIntegrator::render, src/core/integrator.rs:399-480) = wavefront_render, the host driver of the kernels in
// wavefront.h; the per-vertex shading data of the TriangleMesh (pbrt_hip_scene_set_shading_data); the host side of the
// HaltonSampler tables.
#include <hip/hip_runtime.h>
#include "abi_guard.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "scene.h"
#include "trace.h"
#include "wavefront.h"

using namespace pb;

// ------------------------------------------------------------------------------------
// render: see wavefront.h
// ------------------------------------------------------------------------------------
extern "C" int pbrt_hip_render_device(PbrtHipScene* s, const PbrtCamera* camera, const PbrtRenderParams* params,
                                      float* d_film, PbrtRenderStats* stats) try {
    if (!s || !camera || !params || !d_film) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(s->ctx);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    return wavefront_render(s, *camera, *params, d_film, stats);
}
PB_ABI_CATCH

extern "C" int pbrt_hip_render(PbrtHipScene* s, const PbrtCamera* camera, const PbrtRenderParams* params,
                               float* film_xyzw, PbrtRenderStats* stats) try {
    if (!s || !camera || !params || !film_xyzw) return PBRT_HIP_ERR_INVALID;
    PbrtHipContext* ctx = s->ctx;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (params->width <= 0 || params->height <= 0) return PBRT_HIP_ERR_INVALID;
    size_t bytes = (size_t)params->width * params->height * 4 * sizeof(float);
    float* d_film = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_film, bytes));
    int rc = wavefront_render(s, *camera, *params, d_film, stats);
    if (rc == PBRT_HIP_OK && !hip_ok(ctx, hipMemcpy(film_xyzw, d_film, bytes, hipMemcpyDeviceToHost), "film D2H"))
        rc = PBRT_HIP_ERR_DEVICE;
    // after a deadline (ctx->lost) an abandoned kernel may still write the film and hipFree waits for the device: it stays
    if (!ctx->lost) (void)hipFree(d_film);
    return rc;
}
PB_ABI_CATCH

// A film on the context's device for hosts that have no HIP binding of their own (a Rust or C caller): what
// pbrt_hip_render_device renders into, pbrt_hip_film_reduce merges and pbrt_hip_film_download brings home.
extern "C" int pbrt_hip_film_create(PbrtHipContext* ctx, int64_t n_pixels, float** d_film_out) try {
    if (!ctx || !d_film_out || n_pixels <= 0) return PBRT_HIP_ERR_INVALID;
    *d_film_out = nullptr;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n_pixels * 4 * sizeof(float);
    float* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        ctx->last_error = "pbrt_hip_film_create: out of device memory";
        return PBRT_HIP_ERR_OOM;
    }
    HIP_TRY(ctx, e);
    if (!hip_ok(ctx, hipMemsetAsync(d, 0, bytes, ctx->stream), "hipMemsetAsync (film)") ||
        !hip_ok(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize (film)")) {
        (void)hipFree(d);
        return PBRT_HIP_ERR_DEVICE;
    }
    *d_film_out = d;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_film_download(PbrtHipContext* ctx, const float* d_film_xyzw, int64_t n_pixels, float* film_xyzw) try {
    if (!ctx || !d_film_xyzw || !film_xyzw || n_pixels < 0) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // renders and reduces of this context run on its stream
    HIP_TRY(ctx, hipMemcpy(film_xyzw, d_film_xyzw, (size_t)n_pixels * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" void pbrt_hip_film_destroy(PbrtHipContext* ctx, float* d_film_xyzw) {
    if (!ctx || !d_film_xyzw) return;
    PB_LOCK(ctx);
    if (ctx->lost) return;  // an abandoned kernel may still write it and hipFree would wait for the device: it goes with the process
    if (hipSetDevice(ctx->device) == hipSuccess) (void)hipFree(d_film_xyzw);
}

// ------------------------------------------------------------------------------------
// Integrator::li in batch form (integrator.rs:29-42) and the camera-ray stage on its own
// ------------------------------------------------------------------------------------
static int li_params_to_render(const PbrtLiParams* lp, PbrtRenderParams* rp) {
    std::memset(rp, 0, sizeof(*rp));
    rp->integrator = lp->integrator;
    rp->max_depth = lp->max_depth;
    rp->rr_threshold = lp->rr_threshold;
    rp->light_strategy = lp->light_strategy;
    rp->ao_samples = lp->ao_samples;
    rp->tile_world = 1;
    return (lp->draws_before_li < 0 || lp->draws_before_li > 4096) ? PBRT_HIP_ERR_INVALID : PBRT_HIP_OK;
}

extern "C" int pbrt_hip_li_device(PbrtHipScene* s, const PbrtLiParams* lp, const PbrtRay* d_rays, const uint64_t* d_stream_keys,
                                  int64_t n, float* d_rgb, PbrtRenderStats* stats) try {
    if (!s || !lp || n < 0 || (n > 0 && (!d_rays || !d_stream_keys || !d_rgb))) return PBRT_HIP_ERR_INVALID;
    PbrtRenderStats zero{};
    if (stats) *stats = zero;
    if (n == 0) return PBRT_HIP_OK;
    PB_ENTER(s->ctx);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    PbrtRenderParams rp;
    if (li_params_to_render(lp, &rp) != PBRT_HIP_OK) {
        s->ctx->last_error = "li: draws_before_li must be in [0, 4096]";
        return PBRT_HIP_ERR_INVALID;
    }
    LiBatch b;
    b.d_rays = d_rays;
    b.d_keys = d_stream_keys;
    b.n = n;
    b.skip = lp->draws_before_li;
    b.d_rgb = d_rgb;
    PbrtCamera cam{};
    return wavefront_render(s, cam, rp, nullptr, stats, &b);
}
PB_ABI_CATCH

extern "C" int pbrt_hip_li(PbrtHipScene* s, const PbrtLiParams* lp, const PbrtRay* rays, const uint64_t* stream_keys, int64_t n,
                           float* rgb, PbrtRenderStats* stats) try {
    if (!s || !lp || n < 0 || (n > 0 && (!rays || !stream_keys || !rgb))) return PBRT_HIP_ERR_INVALID;
    PbrtRenderStats zero{};
    if (stats) *stats = zero;
    if (n == 0) return PBRT_HIP_OK;
    PbrtHipContext* ctx = s->ctx;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    PbrtRay* d_rays = nullptr;
    uint64_t* d_keys = nullptr;
    float* d_rgb = nullptr;
    int rc = PBRT_HIP_OK;
    if (!hip_ok(ctx, hipMalloc((void**)&d_rays, (size_t)n * sizeof(PbrtRay)), "hipMalloc") ||
        !hip_ok(ctx, hipMalloc((void**)&d_keys, (size_t)n * sizeof(uint64_t)), "hipMalloc") ||
        !hip_ok(ctx, hipMalloc((void**)&d_rgb, (size_t)n * 3 * sizeof(float)), "hipMalloc"))
        rc = PBRT_HIP_ERR_OOM;
    if (rc == PBRT_HIP_OK && (!hip_ok(ctx, hipMemcpy(d_rays, rays, (size_t)n * sizeof(PbrtRay), hipMemcpyHostToDevice), "H2D") ||
                              !hip_ok(ctx, hipMemcpy(d_keys, stream_keys, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice), "H2D")))
        rc = PBRT_HIP_ERR_DEVICE;
    if (rc == PBRT_HIP_OK) rc = pbrt_hip_li_device(s, lp, d_rays, d_keys, n, d_rgb, stats);
    if (rc == PBRT_HIP_OK && !hip_ok(ctx, hipMemcpy(rgb, d_rgb, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost), "D2H"))
        rc = PBRT_HIP_ERR_DEVICE;
    if (!ctx->lost) {  // as pbrt_hip_render: what an abandoned kernel may still touch is never freed
        (void)hipFree(d_rays);
        (void)hipFree(d_keys);
        (void)hipFree(d_rgb);
    }
    return rc;
}
PB_ABI_CATCH

static int round_up_pow2(int v);
extern "C" int pbrt_hip_camera_rays(PbrtHipScene* s, const PbrtCamera* camera, const PbrtRenderParams* params, int64_t capacity,
                                    PbrtRay* rays, uint64_t* stream_keys, float* p_film, int32_t* pixel_sample, int64_t* n_out) try {
    if (!s || !camera || !params || !n_out) return PBRT_HIP_ERR_INVALID;
    PbrtHipContext* ctx = s->ctx;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (params->width <= 0 || params->height <= 0 || params->spp <= 0 || params->x0 > params->x1 || params->y0 > params->y1)
        return PBRT_HIP_ERR_INVALID;
    int32_t n_tiles = 0;
    const int world = params->tile_world <= 0 ? 1 : params->tile_world;
    if (pbrt_hip_tile_partition_order(params->x0, params->y0, params->x1, params->y1, params->tile_rank, world, params->tile_order, nullptr, 0,
                                      &n_tiles) != PBRT_HIP_OK) {
        ctx->last_error = "camera rays: bad tile_rank / tile_world / tile_order";
        return PBRT_HIP_ERR_INVALID;
    }
    // the samples per pixel the sampler takes, as wavefront_render will fix them (stratified.rs:30-33: nx * ny; zerotwosequence.rs:20:
    // the next power of two): the outputs are sized by THAT count, not by the caller's spp
    int64_t spp = params->spp;
    if (params->sampler == PBRT_SAMPLER_STRATIFIED) {
        if (params->sampler_x < 1 || params->sampler_y < 1 || (int64_t)params->sampler_x * params->sampler_y > 65536) {
            ctx->last_error = "stratified sampler: sampler_x * sampler_y must be in [1, 65536]";
            return PBRT_HIP_ERR_INVALID;
        }
        spp = (int64_t)params->sampler_x * params->sampler_y;
    } else if (params->sampler == PBRT_SAMPLER_ZEROTWO) {
        if (params->spp > 65536) {
            ctx->last_error = "(0,2)-sequence sampler: spp too large";
            return PBRT_HIP_ERR_INVALID;
        }
        spp = round_up_pow2(params->spp);
    }
    const int64_t n = (int64_t)n_tiles * kTile * kTile * spp;  // whole tiles: pixels outside the bounds carry pixel = (-1, -1)
    *n_out = n;
    if (n == 0) return PBRT_HIP_OK;
    if (capacity < n || !rays || !stream_keys || !p_film || !pixel_sample) {
        ctx->last_error = "camera rays: output capacity too small (n_out holds the need)";
        return PBRT_HIP_ERR_INVALID;
    }
    LiBatch b;
    int rc = PBRT_HIP_OK;
    if (!hip_ok(ctx, hipMalloc((void**)&b.out_rays, (size_t)n * sizeof(PbrtRay)), "hipMalloc") ||
        !hip_ok(ctx, hipMalloc((void**)&b.out_keys, (size_t)n * sizeof(uint64_t)), "hipMalloc") ||
        !hip_ok(ctx, hipMalloc((void**)&b.out_pfilm, (size_t)n * 2 * sizeof(float)), "hipMalloc") ||
        !hip_ok(ctx, hipMalloc((void**)&b.out_pixel_sample, (size_t)n * 3 * sizeof(int32_t)), "hipMalloc"))
        rc = PBRT_HIP_ERR_OOM;
    if (rc == PBRT_HIP_OK) rc = wavefront_render(s, *camera, *params, nullptr, nullptr, &b);
    if (rc == PBRT_HIP_OK && (!hip_ok(ctx, hipMemcpy(rays, b.out_rays, (size_t)n * sizeof(PbrtRay), hipMemcpyDeviceToHost), "D2H") ||
                              !hip_ok(ctx, hipMemcpy(stream_keys, b.out_keys, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost), "D2H") ||
                              !hip_ok(ctx, hipMemcpy(p_film, b.out_pfilm, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost), "D2H") ||
                              !hip_ok(ctx, hipMemcpy(pixel_sample, b.out_pixel_sample, (size_t)n * 3 * sizeof(int32_t), hipMemcpyDeviceToHost), "D2H")))
        rc = PBRT_HIP_ERR_DEVICE;
    if (!ctx->lost) {
        (void)hipFree(b.out_rays);
        (void)hipFree(b.out_keys);
        (void)hipFree(b.out_pfilm);
        (void)hipFree(b.out_pixel_sample);
    }
    return rc;
}
PB_ABI_CATCH

// ------------------------------------------------------------------------------------
// wavefront_render — host driver of the kernels in wavefront.h
// ------------------------------------------------------------------------------------
namespace {
// Device buffers of one call. Blocks come from (and return to) the context's cache, so that repeated renders of the
// same size do not pay hipMalloc / hipFree again.
struct DevBuf {
    std::vector<size_t> taken;  // indices into ctx->block_cache
    PbrtHipContext* ctx;
    explicit DevBuf(PbrtHipContext* c) : ctx(c) {}
    ~DevBuf() {
        if (ctx->lost) return;  // a kernel of the abandoned call may still be writing these blocks: they stay taken for good
        for (size_t i : taken) ctx->block_cache[i].in_use = false;
        size_t idle = 0;
        for (const auto& b : ctx->block_cache)
            if (!b.in_use) idle += b.bytes;
        if (idle > ((size_t)160 << 30)) trim(ctx);  // renders of many different sizes: do not sit on most of the HBM
    }
    static void trim(PbrtHipContext* ctx) {  // release every block no call is using (slots stay, indices are stable)
        for (auto& b : ctx->block_cache)
            if (!b.in_use && b.ptr) {
                (void)hipFree(b.ptr);
                b.ptr = nullptr;
                b.bytes = 0;
            }
    }
    template <class T>
    T* alloc(size_t n, bool* ok) {
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        // best fit among the free cached blocks that are not wastefully large
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < ctx->block_cache.size(); ++i) {
            const auto& b = ctx->block_cache[i];
            if (!b.in_use && b.ptr && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + 4096 &&
                (best == SIZE_MAX || b.bytes < ctx->block_cache[best].bytes))
                best = i;
        }
        if (best != SIZE_MAX) {
            ctx->block_cache[best].in_use = true;
            taken.push_back(best);
            return (T*)ctx->block_cache[best].ptr;
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {  // out of memory with idle cached blocks around: drop them and retry once
            (void)hipGetLastError();
            trim(ctx);
            e = hipMalloc(&p, bytes);
        }
        if (!hip_ok(ctx, e, "hipMalloc (path state)")) {
            *ok = false;
            return nullptr;
        }
        size_t slot = ctx->block_cache.size();
        for (size_t i = 0; i < ctx->block_cache.size(); ++i)
            if (!ctx->block_cache[i].ptr && !ctx->block_cache[i].in_use) {
                slot = i;
                break;
            }
        if (slot == ctx->block_cache.size()) ctx->block_cache.push_back({nullptr, 0, false});
        ctx->block_cache[slot] = {p, bytes, true};
        taken.push_back(slot);
        return (T*)p;
    }
};
}  // namespace

// ---- optional per-vertex normals / uvs of the TriangleMesh (triangle.rs:17-26, used at :60-72, 252-312, 337-341) ----
__global__ void k_set_shading(const int* __restrict__ slot_prim, const int* __restrict__ idx, const float* __restrict__ pos,
                              const float* __restrict__ normals, const float* __restrict__ tangents,
                              const float* __restrict__ uvs, int n, float4* __restrict__ out, float4* __restrict__ tris) {
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    int prim = slot_prim[slot];
    int v[3] = {idx[3 * (size_t)prim], idx[3 * (size_t)prim + 1], idx[3 * (size_t)prim + 2]};
    float nn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, uv[6] = {0.0f, 0.0f, 1.0f, 0.0f, 1.0f, 1.0f};
    for (int k = 0; k < 3; ++k) {
        if (normals)
            for (int c = 0; c < 3; ++c) nn[3 * k + c] = normals[3 * (size_t)v[k] + c];
        if (tangents)
            for (int c = 0; c < 3; ++c) tt[3 * k + c] = tangents[3 * (size_t)v[k] + c];
        if (uvs)
            for (int c = 0; c < 2; ++c) uv[2 * k + c] = uvs[2 * (size_t)v[k] + c];
    }
    float4* o = out + 6 * (size_t)slot;
    o[0] = make_float4(nn[0], nn[1], nn[2], nn[3]);
    o[1] = make_float4(nn[4], nn[5], nn[6], nn[7]);
    o[2] = make_float4(nn[8], tt[0], tt[1], tt[2]);
    o[3] = make_float4(tt[3], tt[4], tt[5], tt[6]);
    o[4] = make_float4(tt[7], tt[8], uv[0], uv[1]);
    o[5] = make_float4(uv[2], uv[3], uv[4], uv[5]);
    // "Triangle::intersect returns false" depends on the uvs (triangle.rs:197-216): refresh the flag
    float a[3], b[3], c[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = pos[3 * (size_t)v[0] + k];
        b[k] = pos[3 * (size_t)v[1] + k];
        c[k] = pos[3 * (size_t)v[2] + k];
    }
    float4 t2 = tris[3 * (size_t)slot + 2];
    int w = __float_as_int(t2.w) & ~kTriDegenerate;
    if (triangle_rejected_by_intersect(a, b, c, uv)) w |= kTriDegenerate;
    t2.w = __int_as_float(w);
    tris[3 * (size_t)slot + 2] = t2;
}

// the wide-order triangle copies (wide_bvh.h) carry the kTriDegenerate flag too
__global__ void k_wide_refresh_flags(float4* __restrict__ wtris, int wide_stride, const float4* __restrict__ tris, int n) {
    int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    float4 c = wtris[(size_t)wide_stride * (size_t)pos + 2];
    c.z = tris[3 * (size_t)__float_as_int(c.y) + 2].w;
    wtris[(size_t)wide_stride * (size_t)pos + 2] = c;
}

extern "C" int pbrt_hip_scene_set_shading_data(PbrtHipScene* s, const float* positions, int32_t n_verts,
                                               const int32_t* indices, int32_t n_tris, const float* normals,
                                               const float* tangents, const float* uvs) try {
    if (!s) return PBRT_HIP_ERR_INVALID;
    PbrtHipContext* ctx = s->ctx;
    PB_ENTER(ctx);
    auto fail = [&](const char* msg) {
        ctx->last_error = msg;
        return PBRT_HIP_ERR_INVALID;
    };
    if (!positions || !indices || n_tris != s->n_tris || n_verts <= 0) return fail("mesh does not match the scene");
    if (!normals && !tangents && !uvs) return fail("no normals, tangents or uvs given");
    if (s->d.bvh.tri_shading) return fail("shading data already set");
    if (s->d.bvh.has_spheres) return fail("per-vertex shading data is not supported for scenes with spheres");
    for (int64_t i = 0; i < 3 * (int64_t)n_tris; ++i)
        if (indices[i] < 0 || indices[i] >= n_verts) return fail("vertex index out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf tmp(ctx);
    bool ok = true;
    int* d_idx = tmp.alloc<int>(3 * (size_t)n_tris, &ok);
    float* d_pos = tmp.alloc<float>(3 * (size_t)n_verts, &ok);
    float* d_n = normals ? tmp.alloc<float>(3 * (size_t)n_verts, &ok) : nullptr;
    float* d_t = tangents ? tmp.alloc<float>(3 * (size_t)n_verts, &ok) : nullptr;
    float* d_uv = uvs ? tmp.alloc<float>(2 * (size_t)n_verts, &ok) : nullptr;
    void* out = nullptr;
    if (!ok || !hip_ok(ctx, hipMalloc(&out, (size_t)n_tris * 96), "hipMalloc shading data")) return PBRT_HIP_ERR_OOM;
    s->allocs.push_back(out);
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(d_idx, indices, 3 * (size_t)n_tris * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_pos, positions, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    if (d_n) HIP_TRY(ctx, hipMemcpyAsync(d_n, normals, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    if (d_t) HIP_TRY(ctx, hipMemcpyAsync(d_t, tangents, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    if (d_uv) HIP_TRY(ctx, hipMemcpyAsync(d_uv, uvs, 2 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_set_shading, dim3((n_tris + 255) / 256), dim3(256), 0, st, s->d.slot_prim, d_idx, d_pos, d_n, d_t, d_uv,
                       n_tris, (float4*)out, const_cast<float4*>(s->d.bvh.tris));
    if (s->has_wide)
        hipLaunchKernelGGL(k_wide_refresh_flags, dim3((n_tris + 255) / 256), dim3(256), 0, st, const_cast<float4*>(s->wide.tris),
                           s->wide.vec_stride, s->d.bvh.tris, n_tris);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    s->d.bvh.tri_shading = (const float4*)out;
    s->d.bvh.has_normals = normals ? 1 : 0;
    s->d.bvh.has_tangents = tangents ? 1 : 0;
    s->d.bvh.has_uvs = uvs ? 1 : 0;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

namespace pb {
int sort_pairs_u32(hipStream_t st, void* temp, size_t* temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                   const uint32_t* vals_in, uint32_t* vals_out, size_t n, int bits);
}

// ---- HaltonSampler host side: prime tables and compute_radical_inverse_permutations (lowdiscrepancy.rs:11-170,
// 333-349: RNG::default + shuffle per prime), HaltonSampler::new constants (halton.rs:40-98) ----
namespace {
struct HostPcg {  // rng.rs
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t u32() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
    }
    uint32_t bounded(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        for (;;) {
            uint32_t r = u32();
            if (r >= threshold) return r % b;
        }
    }
};
void halton_host_tables(std::vector<uint32_t>* primes_and_sums, std::vector<uint16_t>* perms) {
    std::vector<uint32_t> primes;
    for (uint32_t v = 2; primes.size() < 1000; ++v) {
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
    primes_and_sums->assign(2000, 0);
    uint32_t acc = 0;
    for (int i = 0; i < 1000; ++i) {
        (*primes_and_sums)[i] = primes[i];
        (*primes_and_sums)[1000 + i] = acc;
        acc += primes[i];
    }
    perms->resize(acc);
    HostPcg rng;
    uint16_t* p = perms->data();
    for (int i = 0; i < 1000; ++i) {
        int count = (int)primes[i];
        for (int j = 0; j < count; ++j) p[j] = (uint16_t)j;
        for (int j = 0; j < count; ++j) std::swap(p[j], p[j + (int)rng.bounded((uint32_t)(count - j))]);  // sampling.rs:280-287
        p += count;
    }
}
void extended_gcd(uint64_t a, uint64_t b, int64_t* x, int64_t* y) {  // halton.rs:51-61
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
uint64_t multiplicative_inverse(int64_t a, int64_t n) {  // halton.rs:40-49
    int64_t x = 0, y = 0;
    extended_gcd((uint64_t)a, (uint64_t)n, &x, &y);
    int64_t r = x - (x / n) * n;
    return (uint64_t)(r < 0 ? r + n : r);
}
}  // namespace

static int round_up_pow2(int v) {  // pbrt.rs:174-182
    v -= 1;
    v |= v >> 1;
    v |= v >> 2;
    v |= v >> 4;
    v |= v >> 8;
    v |= v >> 16;
    return v + 1;
}

int wavefront_render(PbrtHipScene* s, const PbrtCamera& camera, const PbrtRenderParams& rp_in, float* d_film,
                     PbrtRenderStats* stats, const LiBatch* li) {
    const bool li_mode = li && li->d_rays, export_mode = li && li->out_rays;
    PbrtHipContext* ctx = s->ctx;
    auto invalid = [&](const char* m) {
        ctx->last_error = m;
        return PBRT_HIP_ERR_INVALID;
    };
    PbrtRenderParams rp = rp_in;
    if (li_mode) {  // the batch stands in for a 1-spp "frame" of n pixels in a row; nothing of the film plumbing is used
        if (li->n <= 0 || li->n > (1ll << 28)) return invalid("li: batch size must be in [1, 2^28]");
        rp.width = (int32_t)li->n;
        rp.height = 1;
        rp.spp = 1;
        rp.x0 = rp.y0 = 0;
        rp.x1 = rp.width;
        rp.y1 = 1;
        rp.tile_rank = 0;
        rp.tile_world = 1;
        rp.spp_per_pass = 1;
        rp.sampler = PBRT_SAMPLER_RANDOM;  // the caller's Sampler is a RandomSampler stream per ray (sampler.rs:452, random.rs)
        rp.filter_table = nullptr;
    }
    if (rp.width <= 0 || rp.height <= 0 || rp.spp <= 0) return invalid("width, height and spp must be positive");
    // ---- sampler: samples per pixel and Sampler::round_count (stratified.rs:30-33, zerotwosequence.rs:20, 62-64) ----
    if (rp.sampler < PBRT_SAMPLER_RANDOM || rp.sampler > PBRT_SAMPLER_HALTON) return invalid("unknown sampler");
    const bool tabulated = rp.sampler != PBRT_SAMPLER_RANDOM;
    if (tabulated && (rp.sampler_dims < 0 || rp.sampler_dims > 63)) return invalid("sampler_dims must be in [0, 63]");
    if (rp.sampler == PBRT_SAMPLER_STRATIFIED) {
        if (rp.sampler_x < 1 || rp.sampler_y < 1 || (int64_t)rp.sampler_x * rp.sampler_y > 65536)
            return invalid("stratified sampler: sampler_x * sampler_y must be in [1, 65536]");
        rp.spp = rp.sampler_x * rp.sampler_y;
    } else if (rp.sampler == PBRT_SAMPLER_ZEROTWO) {
        if (rp.spp > 65536) return invalid("(0,2)-sequence sampler: spp too large");
        rp.spp = round_up_pow2(rp.spp);
    }
    auto round_count = [&](int n) { return rp.sampler == PBRT_SAMPLER_ZEROTWO ? round_up_pow2(n) : n; };
    if (rp.integrator == PBRT_INTEGRATOR_AO && rp.ao_samples >= 1 && rp.ao_samples <= 65535 && tabulated)
        rp.ao_samples = round_count(rp.ao_samples);  // ao.rs:36
    float frx = rp.filter_radius[0] > 0.0f ? rp.filter_radius[0] : 0.5f;
    float fry = rp.filter_radius[1] > 0.0f ? rp.filter_radius[1] : 0.5f;
    bool box = true;  // the 0.5 box filter: exact in-order accumulation (k_film_accumulate)
    if (rp.filter_table) {
        if (frx != 0.5f || fry != 0.5f) box = false;
        for (int i = 0; i < 256; ++i)
            if (rp.filter_table[i] != 1.0f) box = false;
    } else {
        frx = fry = 0.5f;
    }
    if (frx > 64.0f || fry > 64.0f) return invalid("filter radius too large");
    {
        int32_t sb[4];
        pbrt_hip_sample_bounds(rp.width, rp.height, frx, fry, sb);
        if (rp.x0 < sb[0] || rp.y0 < sb[1] || rp.x1 > sb[2] || rp.y1 > sb[3] || rp.x0 > rp.x1 || rp.y0 > rp.y1)
            return invalid("pixel bounds outside the film's sample bounds");
    }
    if (rp.integrator < PBRT_INTEGRATOR_PATH || rp.integrator > PBRT_INTEGRATOR_AO)
        return invalid("integrator must be a PbrtIntegratorKind");
    // DirectLighting, Whitted and AO share the per-vertex stage machine of k_shade_direct
    const bool direct = rp.integrator != PBRT_INTEGRATOR_PATH;
    if (direct && rp.max_depth > 64) return invalid("direct lighting / Whitted: max_depth > 64 (frame stack)");
    if (rp.integrator == PBRT_INTEGRATOR_AO && (rp.ao_samples < 1 || rp.ao_samples > 65535))
        return invalid("ambient occlusion: ao_samples must be in [1, 65535]");
    if (rp.max_depth < 0 || rp.max_depth > 1 << 20) return invalid("bad max_depth");
    int world = rp.tile_world <= 0 ? 1 : rp.tile_world;
    int rank = rp.tile_rank;
    if (rank < 0 || rank >= world) return invalid("tile_rank outside [0, tile_world)");
    if (rp.tile_order != PBRT_TILE_ORDER_MORTON && rp.tile_order != PBRT_TILE_ORDER_ROW_MAJOR)
        return invalid("tile_order must be a PbrtTileOrder");
    if (rp.samples_per_wave < 0 || rp.samples_per_wave > 64 || (rp.samples_per_wave & (rp.samples_per_wave - 1)) != 0)
        return invalid("samples_per_wave must be 0 (library default) or a power of two up to 64");
    hipStream_t st = ctx->stream;

    size_t film_bytes = (size_t)rp.width * rp.height * 4 * sizeof(float);
    if (d_film) HIP_TRY(ctx, hipMemsetAsync(d_film, 0, film_bytes, st));

    // tiles of the sample bounds (integrator.rs:402-409), dealt round-robin in rp.tile_order to the GPUs
    std::vector<int2> origins;
    if (li_mode) {
        origins.assign((size_t)((li->n + kTile * kTile - 1) / (kTile * kTile)), make_int2(0, 0));  // 256 rays per "tile"
    } else {
        int32_t n_mine = 0;
        if (pbrt_hip_tile_partition_order(rp.x0, rp.y0, rp.x1, rp.y1, rank, world, rp.tile_order, nullptr, 0, &n_mine) != PBRT_HIP_OK)
            return invalid("tile partition: too many tiles");
        origins.resize(n_mine);
        pbrt_hip_tile_partition_order(rp.x0, rp.y0, rp.x1, rp.y1, rank, world, rp.tile_order, (int32_t*)origins.data(), n_mine, &n_mine);
    }
    PbrtRenderStats local{};
    if (origins.empty()) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (stats) *stats = local;
        return PBRT_HIP_OK;
    }
    const int n_pix = (int)origins.size() * kTile * kTile;
    int64_t valid_pixels = 0;
    for (const int2& o : origins)
        valid_pixels += (int64_t)(std::min(o.x + kTile, rp.x1) - o.x) * (std::min(o.y + kTile, rp.y1) - o.y);
    if (li_mode) valid_pixels = li->n;
    if (export_mode) rp.spp_per_pass = rp.spp;  // one pass: the export is indexed by the paths of that pass
    int spp_pass = rp.spp_per_pass > 0 ? rp.spp_per_pass : 0;
    if (spp_pass == 0) {
        // As many samples of a pixel in flight as half of the free HBM holds (about 400 B of path state, queue and
        // sort slots per path; the stage machine of the other integrators adds its frame stack): a whole 64-spp
        // 1080p frame is 133 M paths = 50 GB of the 288 GB and runs 6 large wavefronts instead of 48 small ones
        // (+10 % on config 3: fewer launches and host round trips, shorter tails).
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        for (const auto& b : ctx->block_cache)
            if (!b.in_use) free_b += b.bytes;  // blocks kept from the previous render are available to this one
        const int64_t per_path = 400 + (rp.integrator != PBRT_INTEGRATOR_PATH ? 24 + 48ll * std::max(1, rp.max_depth) : 0);
        const int64_t target_paths = std::max<int64_t>(1ll << 20, std::min<int64_t>(1ll << 28, (int64_t)(free_b / 2) / per_path));
        spp_pass = (int)std::max<int64_t>(1, std::min<int64_t>(rp.spp, target_paths / n_pix));
    }
    spp_pass = std::min(spp_pass, rp.spp);
    if (export_mode && spp_pass != rp.spp) return invalid("camera rays: too many samples for one pass");
    const size_t N = (size_t)n_pix * spp_pass;
    if (N * 4 >= (1ull << 32)) return invalid("too many concurrent paths for 32-bit queue entries; lower spp_per_pass");

    DevBuf buf(ctx);
    bool ok = true;
    PathState ps;
    ps.n_paths = N;
    ps.two_level = s->d.bvh.instanced ? 1 : 0;
    ps.ray = buf.alloc<float4>(N * 6, &ok);
    ps.hit = buf.alloc<float4>(N * 6, &ok);
    ps.rng = buf.alloc<uint64_t>(N, &ok);
    ps.L = buf.alloc<float4>(N, &ok);
    ps.beta = buf.alloc<float4>(N, &ok);
    ps.nee_a = buf.alloc<float4>(N, &ok);
    ps.nee_f = buf.alloc<float4>(N, &ok);
    ps.nee_b = buf.alloc<float4>(N, &ok);
    ps.nee_light = buf.alloc<int>(N, &ok);
    ps.pfilm = buf.alloc<float2>(N, &ok);
    Queues q[2];
    for (int k = 0; k < 2; ++k) {
        q[k].trace = buf.alloc<uint32_t>(N * 3, &ok);
        q[k].shade = buf.alloc<uint32_t>(N, &ok);
        q[k].counts64 = buf.alloc<unsigned long long>(2, &ok);
        q[k].keys = nullptr;
        for (int a = 0; a < 3; ++a) {
            const float lo = s->d.bvh.root_min[a], hi = s->d.bvh.root_max[a];
            q[k].key_lo[a] = lo;
            q[k].key_inv[a] = hi > lo ? 1.0f / (hi - lo) : 0.0f;
        }
    }
    // ray-queue sort (spatial order for the bounce rays): keys in / out, sorted entries, rocPRIM scratch
    // only the path integrator's later bounces are incoherent; the stage machine of the other integrators keeps
    // shooting from the camera rays' hit points, which are already in pixel order (AO: -6 % with the sort).
    // PbrtRenderParams.ray_order = 1 leaves the queues as the shading kernel filled them (results do not depend on it).
    if (rp.ray_order < 0 || rp.ray_order > 1) return invalid("ray_order must be 0 (Morton order from the second bounce on) or 1 (queue order)");
    const bool sort_rays = rp.ray_order == 0 && rp.integrator == PBRT_INTEGRATOR_PATH;
    constexpr int sort_from = 2;  // first sorted wavefront: the first bounce still follows the pixel order of its camera rays
    uint32_t *sort_keys[2] = {nullptr, nullptr}, *sort_vals = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    if (sort_rays) {
        sort_keys[0] = buf.alloc<uint32_t>(N * 3, &ok);
        sort_keys[1] = buf.alloc<uint32_t>(N * 3, &ok);
        sort_vals = buf.alloc<uint32_t>(N * 3, &ok);
        if (pb::sort_pairs_u32(st, nullptr, &sort_tmp_bytes, sort_keys[0], sort_keys[1], sort_vals, sort_vals, N * 3, kSortKeyBits) != 0)
            return invalid("rocPRIM radix sort: size query failed");
        sort_tmp = buf.alloc<char>(sort_tmp_bytes, &ok);
    }
    // rays the wide kernel leaves to the binary one (wide_bvh.h): room for every entry of a trace queue
    const bool use_wide = s->has_wide && ctx->count_traversal != 1 && ctx->traversal == PBRT_TRAVERSAL_AUTO;
    const bool stackless = ctx->traversal == PBRT_TRAVERSAL_STACKLESS;
    if (stackless && !stackless_applies(s)) return PBRT_HIP_ERR_INVALID;
    uint32_t* special_list = use_wide ? buf.alloc<uint32_t>(N * 3, &ok) : nullptr;
    // sort-by-material shading (wf_path.h): PbrtRenderParams.shade_order; 2 = the whole shade queue in material order
    if (rp.shade_order < 0 || rp.shade_order > 2) return invalid("shade_order must be 0 (queue order), 1 (by material inside blocks) or 2 (sorted queue)");
    const bool sort_shade = rp.shade_order == 2 && rp.integrator == PBRT_INTEGRATOR_PATH;
    const bool bin_shade = rp.shade_order == 1 && rp.integrator == PBRT_INTEGRATOR_PATH;
    uint32_t *shade_keys[2] = {nullptr, nullptr}, *shade_sorted = nullptr;
    void* shade_tmp = nullptr;
    size_t shade_tmp_bytes = 0;
    if (sort_shade) {
        shade_keys[0] = buf.alloc<uint32_t>(N, &ok);
        shade_keys[1] = buf.alloc<uint32_t>(N, &ok);
        shade_sorted = buf.alloc<uint32_t>(N, &ok);
        if (pb::sort_pairs_u32(st, nullptr, &shade_tmp_bytes, shade_keys[0], shade_keys[1], shade_sorted, shade_sorted, N, 3) != 0)
            return invalid("rocPRIM radix sort: size query failed");
        shade_tmp = buf.alloc<char>(shade_tmp_bytes, &ok);
    }
    DirectState ds{};
    std::vector<int> prefix(s->d.n_lights + 1, 0);
    for (int i = 0; i < s->d.n_lights; ++i)  // directlighting.rs:58-62: round_count with a tabulating sampler
        prefix[i + 1] = prefix[i] + (tabulated ? round_count(s->light_samples[i]) : s->light_samples[i]);
    // ---- PixelSampler tables: n_dims 1D + n_dims 2D dimensions and the requested 2D arrays, per pixel ----
    SamplerParams smp{};
    smp.kind = rp.sampler;
    smp.n_dims = rp.sampler_dims;
    smp.nx = rp.sampler_x;
    smp.ny = rp.sampler_y;
    smp.jitter = rp.sampler_jitter;
    ps.samp = buf.alloc<int>(N, &ok);
    if (rp.sampler == PBRT_SAMPLER_HALTON) {
        smp.n_dims = 0;  // nothing is tabulated per pixel: every value is a function of (sample index, dimension)
        if (!ctx->d_halton_primes) {
            std::vector<uint32_t> primes;
            std::vector<uint16_t> perms;
            halton_host_tables(&primes, &perms);
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_halton_primes, primes.size() * sizeof(uint32_t)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_halton_perms, perms.size() * sizeof(uint16_t)));
            HIP_TRY(ctx, hipMemcpy(ctx->d_halton_primes, primes.data(), primes.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(ctx->d_halton_perms, perms.data(), perms.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        smp.primes = ctx->d_halton_primes;
        smp.perms = ctx->d_halton_perms;
        // HaltonSampler::new(spp, film.get_sample_bounds(), false) (halton.rs:63-98)
        int32_t sb[4];
        pbrt_hip_sample_bounds(rp.width, rp.height, frx, fry, sb);
        const int res[2] = {sb[2] - sb[0], sb[3] - sb[1]};
        for (int i = 0; i < 2; ++i) {
            int base = i == 0 ? 2 : 3, scale = 1, exp = 0;
            while (scale < std::min(128, res[i])) {
                scale *= base;
                ++exp;
            }
            smp.h_scale[i] = scale;
            smp.h_exp[i] = exp;
        }
        smp.h_stride = smp.h_scale[0] * smp.h_scale[1];
        smp.h_minv[0] = (unsigned int)multiplicative_inverse(smp.h_scale[1], smp.h_scale[0]);
        smp.h_minv[1] = (unsigned int)multiplicative_inverse(smp.h_scale[0], smp.h_scale[1]);
    }
    if (tabulated) {
        std::vector<int2> arrays;
        int64_t elems = (int64_t)smp.n_dims * rp.spp * 3;
        smp.off2 = smp.n_dims * rp.spp;
        auto request_2d_array = [&](int n) {  // sampler.rs:41-46
            arrays.push_back(make_int2(n, (int)elems));
            elems += (int64_t)n * rp.spp * 2;
        };
        if (rp.integrator == PBRT_INTEGRATOR_DIRECT && rp.light_strategy == 0) {
            for (int i = 0; i < rp.max_depth; ++i)  // directlighting.rs:64-75
                for (int j = 0; j < s->d.n_lights; ++j) {
                    request_2d_array(prefix[j + 1] - prefix[j]);
                    request_2d_array(prefix[j + 1] - prefix[j]);
                }
        } else if (rp.integrator == PBRT_INTEGRATOR_AO) {
            request_2d_array(rp.ao_samples);  // ao.rs:37
        }
        if (arrays.size() > 0x7fff) return invalid("too many sample arrays (max_depth x lights)");
        if (elems * n_pix * 4 > (64ll << 30) || elems >= (1ll << 31))
            return invalid("sampler tables exceed 64 GB: lower spp, sampler_dims or the light sample counts");
        smp.n_arrays = (int)arrays.size();
        smp.array_end_dim = 5 + 2 * smp.n_arrays;  // sampler.rs:344-345
        if (rp.sampler == PBRT_SAMPLER_HALTON) {
            // start_pixel fills the arrays' dimensions 5..array_end_dim (sampler.rs:355-368); the first one past the
            // prime table indexes PRIME_SUMS out of range (halton.rs:100-108) and the reference panics there
            if (smp.array_end_dim > 1000) return invalid("Halton sampler: too many sample arrays for 1000 dimensions");
            elems = 0;
        }
        smp.n_elems = (int)elems;
        smp.tables = buf.alloc<float>((size_t)elems * n_pix, &ok);
        int2* d_arrays = buf.alloc<int2>(arrays.size(), &ok);
        smp.arrays = d_arrays;
        if (ok && !arrays.empty())
            HIP_TRY(ctx, hipMemcpyAsync(d_arrays, arrays.data(), arrays.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        if (ok) HIP_TRY(ctx, hipStreamSynchronize(st));  // `arrays` leaves scope
    }
    int* d_prefix = buf.alloc<int>(prefix.size(), &ok);
    if (direct) {
        ds.stage = buf.alloc<int>(N, &ok);
        ds.ld_acc = buf.alloc<float4>(N, &ok);
        ds.frames = buf.alloc<float4>(N * (size_t)std::max(1, rp.max_depth) * 3, &ok);
        ds.light_strategy = rp.light_strategy;
        ds.mode = rp.integrator;
        ds.ao_samples = rp.ao_samples;
        ds.ao_cos_sample = rp.light_strategy != 0;
        if (prefix.back() >= 0xfff0) return invalid("too many light samples per vertex");
    }
    float* d_filter = nullptr;
    if (!box) {
        d_filter = buf.alloc<float>(256, &ok);
        if (ok) HIP_TRY(ctx, hipMemcpyAsync(d_filter, rp.filter_table, 256 * sizeof(float), hipMemcpyHostToDevice, st));
    }
    float4* accum = buf.alloc<float4>(n_pix, &ok);
    int2* d_origins = buf.alloc<int2>(origins.size(), &ok);
    if (!ok) return PBRT_HIP_ERR_OOM;
    HIP_TRY(ctx, hipMemcpyAsync(d_origins, origins.data(), origins.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(accum, 0, (size_t)n_pix * sizeof(float4), st));
    HIP_TRY(ctx, hipMemcpyAsync(d_prefix, prefix.data(), prefix.size() * sizeof(int), hipMemcpyHostToDevice, st));
    TileList tiles{d_origins, (int)origins.size()};

    DevCamera cam;
    std::memcpy(cam.c2w, camera.camera_to_world, 64);
    std::memcpy(cam.r2c, camera.raster_to_camera, 64);
    cam.lens_radius = camera.lens_radius;
    cam.focal_distance = camera.focal_distance;
    cam.shutter_open = camera.shutter_open;
    cam.shutter_close = camera.shutter_close;
    cam.kind = camera.kind;
    if (camera.kind < PBRT_CAMERA_PERSPECTIVE || camera.kind > PBRT_CAMERA_ENVIRONMENT) return invalid("unknown camera kind");

    ShadeConsts sc;
    sc.bvh = s->d.bvh;
    sc.materials = s->d.materials;
    sc.lights = s->d.lights;
    sc.n_lights = s->d.n_lights;
    sc.n_infinite = s->d.n_infinite;
    sc.infinite_ids = s->d.infinite_ids;
    // create_light_sample_distribution (lightdistrib.rs:222-232): "uniform", or one light -> uniform
    sc.distrib = (rp.light_strategy == 0 || s->d.n_lights == 1) ? s->d.light_distrib_uniform : s->d.light_distrib_power;
    if (rp.integrator == PBRT_INTEGRATOR_PATH && (rp.light_strategy < 0 || rp.light_strategy > 2))
        return invalid("path: light_strategy must be 0 (uniform), 1 (power) or 2 (spatial)");
    std::memcpy(sc.env_cond_func, s->d.env_cond_func, sizeof(sc.env_cond_func));
    std::memcpy(sc.env_cond_cdf, s->d.env_cond_cdf, sizeof(sc.env_cond_cdf));
    std::memcpy(sc.env_cond_int, s->d.env_cond_int, sizeof(sc.env_cond_int));
    std::memcpy(sc.env_marg_func, s->d.env_marg_func, sizeof(sc.env_marg_func));
    std::memcpy(sc.env_marg_cdf, s->d.env_marg_cdf, sizeof(sc.env_marg_cdf));
    sc.env_marg_int = s->d.env_marg_int;
    sc.world_radius = s->d.world_radius;
    sc.light_sample_prefix = d_prefix;
    sc.total_light_samples = prefix.back();
    sc.spatial = nullptr;
    sc.n_voxel[0] = sc.n_voxel[1] = sc.n_voxel[2] = 1;
    sc.mis_bool = ctx->count_traversal != 1 ? 1 : 0;  // wf_state.h: RS_MIS_BOOL
    if (rp.integrator == PBRT_INTEGRATOR_PATH && rp.light_strategy == 2 && s->d.n_lights > 1) {
        // create_light_sample_distribution("spatial") -> SpatialLightDistribution::new(scene, 64) (lightdistrib.rs:85-107, 228)
        if (!s->d_spatial) {
            const float* mn = s->d.bvh.root_min;
            const float* mx = s->d.bvh.root_max;
            float diag[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
            int ext = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);
            for (int i = 0; i < 3; ++i) s->spatial_voxels[i] = std::max(1, (int)std::round(diag[i] / diag[ext] * 64.0f));
            size_t n_voxels = (size_t)s->spatial_voxels[0] * s->spatial_voxels[1] * s->spatial_voxels[2];
            size_t floats = n_voxels * (size_t)(2 * s->d.n_lights + 2);
            if (floats > (1ull << 30)) return invalid("spatial light distribution: voxels x lights exceed 4 GB; use \"power\"");
            void* p = nullptr;
            HIP_TRY(ctx, hipMalloc(&p, floats * sizeof(float)));
            s->allocs.push_back(p);
            for (int i = 0; i < 3; ++i) sc.n_voxel[i] = s->spatial_voxels[i];
            hipLaunchKernelGGL(k_spatial_light_tables, dim3((unsigned)((n_voxels + 63) / 64)), dim3(64), 0, st, sc, (float*)p);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipStreamSynchronize(st));
            s->d_spatial = (float*)p;
        }
        sc.spatial = s->d_spatial;
        for (int i = 0; i < 3; ++i) sc.n_voxel[i] = s->spatial_voxels[i];
    }

    hipEvent_t e_begin = nullptr, e_end = nullptr, e_t0 = nullptr, e_t1 = nullptr;
    auto cleanup_events = [&]() {
        for (hipEvent_t e : {e_begin, e_end, e_t0, e_t1})
            if (e) (void)hipEventDestroy(e);
    };
    if (!hip_ok(ctx, hipEventCreate(&e_begin), "hipEventCreate") || !hip_ok(ctx, hipEventCreate(&e_end), "hipEventCreate") ||
        !hip_ok(ctx, hipEventCreate(&e_t0), "hipEventCreate") || !hip_ok(ctx, hipEventCreate(&e_t1), "hipEventCreate")) {
        cleanup_events();
        return PBRT_HIP_ERR_DEVICE;
    }
    int rc = PBRT_HIP_OK;
#define RENDER_TRY(call)                                  \
    if (rc == PBRT_HIP_OK && !hip_ok(ctx, (call), #call)) rc = PBRT_HIP_ERR_DEVICE;

    RENDER_TRY(hipEventRecord(e_begin, st));
    bool tables_ready = !tabulated || rp.sampler == PBRT_SAMPLER_HALTON;
    for (int s0 = 0; s0 < rp.spp && rc == PBRT_HIP_OK; s0 += spp_pass) {
        PassParams pp;
        pp.smp = smp;
        pp.n_pix = n_pix;
        pp.n_samples = std::min(spp_pass, rp.spp - s0);
        {   // PbrtRenderParams.samples_per_wave: the largest power of two <= the request that divides the pass's samples per pixel
            // default: a wave = the samples of ONE pixel (as many as the pass has, up to 64). The tabulating samplers keep the
            // old layout: their tables are [element][pixel], read coalesced when a wave's lanes are 64 pixels of one sample
            const int want = rp.samples_per_wave > 0 ? rp.samples_per_wave : ((tabulated && rp.sampler != PBRT_SAMPLER_HALTON) ? 1 : 64);
            pp.group_shift = 0;
            while ((2 << pp.group_shift) <= want && pp.group_shift < 6 && pp.n_samples % (2 << pp.group_shift) == 0) ++pp.group_shift;
            if (li_mode || export_mode) pp.group_shift = 0;  // batches of rays / the exported camera rays keep the caller's order
        }
        pp.sample0 = s0;
        pp.spp = rp.spp;
        pp.width = rp.width;
        pp.height = rp.height;
        pp.x0 = rp.x0;
        pp.y0 = rp.y0;
        pp.x1 = rp.x1;
        pp.y1 = rp.y1;
        {
            int32_t sb[4];
            pbrt_hip_sample_bounds(rp.width, rp.height, frx, fry, sb);
            pp.sb_x0 = sb[0];
            pp.sb_y0 = sb[1];
            pp.sb_w = sb[2] - sb[0];
        }
        pp.seed = rp.seed;
        pp.max_depth = rp.max_depth;
        pp.rr_threshold = rp.rr_threshold;
        pp.light_strategy = rp.light_strategy;
        pp.filter_rx = frx;
        pp.filter_ry = fry;
        pp.filter_table = d_filter;
        pp.max_sample_luminance = rp.max_sample_luminance > 0.0f ? rp.max_sample_luminance : INFINITY;
        pp.stream_keys = li_mode ? li->d_keys : nullptr;
        uint32_t n_paths = (uint32_t)n_pix * pp.n_samples;
        int cur = 0;
        if (!tables_ready) {  // Sampler::start_pixel for every pixel of this GPU, once per render
            hipLaunchKernelGGL(k_sampler_tables, dim3((n_pix + 63) / 64), dim3(64), 0, st, pp, tiles);
            RENDER_TRY(hipGetLastError());
            tables_ready = true;
        }
        if (li_mode)
            hipLaunchKernelGGL(k_li_generate, dim3((n_paths + 255) / 256), dim3(256), 0, st, ps, q[cur], li->d_rays, li->d_keys,
                               (uint32_t)li->n, n_paths, li->skip);
        else
            hipLaunchKernelGGL(k_generate, dim3((n_paths + 255) / 256), dim3(256), 0, st, ps, q[cur], pp, cam, tiles);
        RENDER_TRY(hipGetLastError());
        if (export_mode) {
            hipLaunchKernelGGL(k_camera_rays_out, dim3((n_paths + 255) / 256), dim3(256), 0, st, ps, pp, tiles, li->out_rays, li->out_keys,
                               li->out_pfilm, li->out_pixel_sample);
            RENDER_TRY(hipGetLastError());
            break;
        }
        if (direct) {
            RENDER_TRY(hipMemsetAsync(ds.stage, 0, (size_t)n_paths * sizeof(int), st));
            RENDER_TRY(hipMemsetAsync(ds.ld_acc, 0, (size_t)n_paths * sizeof(float4), st));
        }
        // the first wavefront is the identity: every path traces its camera ray and is shaded
        unsigned long long counts[2] = {n_paths, n_paths};
        local.camera_samples += (uint64_t)valid_pixels * pp.n_samples;
        local.rays_closest += (uint64_t)valid_pixels * pp.n_samples;
        bool first = true;
        int wavefront = 0;  // 0 = camera rays, 1 = first bounce + its shadow rays, ...
        while (rc == PBRT_HIP_OK && counts[1] > 0) {
            uint32_t n_trace = (uint32_t)counts[0], n_shade = (uint32_t)counts[1];
            if (n_trace > 0) {
                RENDER_TRY(hipMemsetAsync(ctx->d_work_counter, 0, kWorkCounters * sizeof(unsigned int), st));
                const uint32_t* trace_queue = q[cur].trace;
                if (sort_rays && wavefront >= sort_from && n_trace >= (1u << 20)) {
                    // from the second bounce on the rays of a wavefront start all over the scene (the first bounce still
                    // follows the pixel order of its camera rays): trace them in Morton order of their origins (the keys were
                    // written by k_shade together with the queue entries)
                    size_t tb = sort_tmp_bytes;
                    if (pb::sort_pairs_u32(st, sort_tmp, &tb, sort_keys[0], sort_keys[1], q[cur].trace, sort_vals, n_trace, kSortKeyBits) != 0 &&
                        rc == PBRT_HIP_OK) {
                        ctx->last_error = "rocPRIM radix sort failed";
                        rc = PBRT_HIP_ERR_DEVICE;
                    }
                    trace_queue = sort_vals;
                }
                RENDER_TRY(hipEventRecord(e_t0, st));
                {
                    dim3 grid(persistent_grid(s)), block(kTraceBlock);
                    const int inst = s->d.bvh.instanced ? (s->d.bvh.general_top ? 2 : 1) : 0;  // trace_persistent.h: INST
                    int segments = (wavefront == 0 && !inst) ? kQueueSegments : 1;  // see trace.h
                    const bool count_ref = ctx->count_traversal == 1, count_wide = ctx->count_traversal == 2;
#define PB_LAUNCH_BINARY(COUNT, INST, SPH) \
    hipLaunchKernelGGL((k_trace<COUNT, INST, SPH>), grid, block, 0, st, s->d.bvh, ps, trace_queue, n_trace, ctx->d_work_counter, ctx->d_counters, segments)
#define PB_LAUNCH_WIDE(COUNT, INST)                                                                                                        \
    hipLaunchKernelGGL((k_trace_wide<COUNT, INST>), dim3(persistent_grid(s, (COUNT || INST) ? PB_WIDE_INST_WAVES : PB_WIDE_WAVES, wide_stack_lds(INST), wide_world_lds_bytes(INST))), \
                       block, 0, st, wt, ps, trace_queue, n_trace, ctx->d_work_counter, segments, ctx->d_counters)
#define PB_LAUNCH_SPECIAL(INST)                                                                                   \
    hipLaunchKernelGGL(k_trace_special<INST>, grid, block, 0, st, s->d.bvh, ps, trace_queue, n_trace, special_list, \
                       ctx->d_work_counter + kSpecialCount, ctx->d_work_counter + kFollowUpCounter)
                    if (stackless) {
                        hipLaunchKernelGGL(k_trace_stackless, dim3(stackless_grid(s)), block, 0, st, s->d.bvh, ps, trace_queue, n_trace,
                                           ctx->d_work_counter, segments);
                    } else if (use_wide) {
                        WideTrees wt = s->wide;
                        wt.special_list = special_list;
                        wt.special_count = ctx->d_work_counter + kSpecialCount;
                        if (inst == 2) {
                            if (count_wide) PB_LAUNCH_WIDE(true, 2); else PB_LAUNCH_WIDE(false, 2);
                            PB_LAUNCH_SPECIAL(2);
                        } else if (inst == 1) {
                            if (count_wide) PB_LAUNCH_WIDE(true, 1); else PB_LAUNCH_WIDE(false, 1);
                            PB_LAUNCH_SPECIAL(1);
                        } else {
                            if (count_wide) PB_LAUNCH_WIDE(true, 0); else PB_LAUNCH_WIDE(false, 0);
                            PB_LAUNCH_SPECIAL(0);
                        }
                    } else if (s->d.bvh.has_spheres) {
                        if (count_ref) PB_LAUNCH_BINARY(true, 0, true); else PB_LAUNCH_BINARY(false, 0, true);
                    } else if (inst == 2) {
                        if (count_ref) PB_LAUNCH_BINARY(true, 2, false); else PB_LAUNCH_BINARY(false, 2, false);
                    } else if (inst == 1) {
                        if (count_ref) PB_LAUNCH_BINARY(true, 1, false); else PB_LAUNCH_BINARY(false, 1, false);
                    } else {
                        if (count_ref) PB_LAUNCH_BINARY(true, 0, false); else PB_LAUNCH_BINARY(false, 0, false);
                    }
#undef PB_LAUNCH_BINARY
#undef PB_LAUNCH_WIDE
#undef PB_LAUNCH_SPECIAL
                }
                RENDER_TRY(hipGetLastError());
                RENDER_TRY(hipEventRecord(e_t1, st));
            }
            int nxt = cur ^ 1;
            RENDER_TRY(hipMemsetAsync(q[nxt].counts64, 0, 2 * sizeof(unsigned long long), st));
            // the wavefront this launch of k_shade fills is traced in Morton order: have it write the keys as well
            q[nxt].keys = (sort_rays && wavefront + 1 >= sort_from) ? sort_keys[0] : nullptr;
            if (direct)
{
                dim3 sg((n_shade + 255) / 256), sb(256);
                if (rp.integrator == PBRT_INTEGRATOR_DIRECT)
                    hipLaunchKernelGGL((k_shade_direct<PBRT_INTEGRATOR_DIRECT>), sg, sb, 0, st, sc, ps, ds, q[cur], q[nxt], pp, tiles, n_shade);
                else if (rp.integrator == PBRT_INTEGRATOR_WHITTED)
                    hipLaunchKernelGGL((k_shade_direct<PBRT_INTEGRATOR_WHITTED>), sg, sb, 0, st, sc, ps, ds, q[cur], q[nxt], pp, tiles, n_shade);
                else
                    hipLaunchKernelGGL((k_shade_direct<PBRT_INTEGRATOR_AO>), sg, sb, 0, st, sc, ps, ds, q[cur], q[nxt], pp, tiles, n_shade);
            }
            else {
                Queues qin = q[cur];
                if (sort_shade && wavefront >= 1 && n_shade >= (1u << 10)) {  // the first wavefront is all camera hits in pixel order
                    hipLaunchKernelGGL(k_shade_sort_keys, dim3((n_shade + 255) / 256), dim3(256), 0, st, sc, ps, q[cur].shade, n_shade,
                                       pp.max_depth, shade_keys[0]);
                    size_t tb = shade_tmp_bytes;
                    if (pb::sort_pairs_u32(st, shade_tmp, &tb, shade_keys[0], shade_keys[1], q[cur].shade, shade_sorted, n_shade, 3) != 0 &&
                        rc == PBRT_HIP_OK) {
                        ctx->last_error = "rocPRIM radix sort failed";
                        rc = PBRT_HIP_ERR_DEVICE;
                    }
                    qin.shade = shade_sorted;
                }
                if (bin_shade && wavefront >= 1)
                    hipLaunchKernelGGL(k_shade<true>, dim3((n_shade + 255) / 256), dim3(256), 0, st, sc, ps, qin, q[nxt], pp, tiles, n_shade);
                else
                    hipLaunchKernelGGL(k_shade<false>, dim3((n_shade + 255) / 256), dim3(256), 0, st, sc, ps, qin, q[nxt], pp, tiles, n_shade);
            }
            RENDER_TRY(hipGetLastError());
            RENDER_TRY(hipMemcpyAsync(ctx->h_counts, q[nxt].counts64, sizeof(counts), hipMemcpyDeviceToHost, st));
            RENDER_TRY(hipEventRecord(ctx->ev_sync, st));
            if (rc == PBRT_HIP_OK) {
                // spin on the event (a blocking sync costs a scheduler wake-up per wavefront), but not for ever: a kernel
                // that never finishes must surface as an error of this call, with the context's lock released
                hipError_t qe;
                const auto spin_start = std::chrono::steady_clock::now();
                uint32_t polls = 0;
                while ((qe = hipEventQuery(ctx->ev_sync)) == hipErrorNotReady) {
                    if ((++polls & 0xfffu) == 0) {
                        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - spin_start).count();
                        if (waited > ctx->wavefront_deadline_s) {
                            ctx->last_error = "a wavefront did not finish within the deadline (hung kernel?)";
                            ctx->lost = true;  // the stream is not drained below: nothing may follow on this context
                            rc = PBRT_HIP_ERR_DEVICE;
                            break;
                        }
                        if (waited > 0.05) std::this_thread::yield();  // long wavefronts: stop burning a core
                    }
                }
                if (rc == PBRT_HIP_OK) RENDER_TRY(qe);
            }
            counts[0] = ctx->h_counts[0];
            counts[1] = ctx->h_counts[1];
            if (n_trace > 0 && rc == PBRT_HIP_OK) {
                float ms = 0.0f;
                RENDER_TRY(hipEventElapsedTime(&ms, e_t0, e_t1));
                local.trace_ms += ms;
                local.trace_launches += 1;
                if (ctx->trace_log)
                    std::fprintf(stderr, "[pbrt_hip] k_trace: %u rays %.3f ms (%.0f Mrays/s)\n", n_trace, ms, n_trace / ms * 1e-3);
                ctx->trace_ms += ms;
                ctx->trace_launches += 1;
            }
            {
                uint64_t n_rays = counts[0] & 0xffffffffull, n_shadow = counts[0] >> 32;
                local.rays_closest += n_rays - n_shadow;
                local.rays_shadow += n_shadow;
                counts[0] = n_rays;
            }
            (void)first;
            first = false;
            wavefront += 1;
            cur = nxt;
        }
        if (li_mode) {
            hipLaunchKernelGGL(k_li_output, dim3((unsigned)((li->n + 255) / 256)), dim3(256), 0, st, ps, (uint32_t)li->n, li->d_rgb);
            RENDER_TRY(hipGetLastError());
            continue;
        }
        if (!box) {
            hipLaunchKernelGGL(k_film_splat, dim3((n_paths + 255) / 256), dim3(256), 0, st, ps, pp, tiles, d_film);
            RENDER_TRY(hipGetLastError());
            continue;
        }
        if (pp.group_shift == 6)  // a wave of paths = one pixel's samples: sixteen lanes per pixel, coalesced reads, ordered row sums
            hipLaunchKernelGGL(k_film_accumulate_rows, dim3((unsigned)(((size_t)n_pix * 16 + 255) / 256)), dim3(256), 0, st, ps, pp, tiles, accum, d_film);
        else
            hipLaunchKernelGGL(k_film_accumulate, dim3((n_pix + 255) / 256), dim3(256), 0, st, ps, pp, tiles, accum, d_film);
        RENDER_TRY(hipGetLastError());
        if (s0 + spp_pass >= rp.spp) {
            hipLaunchKernelGGL(k_film_merge, dim3((n_pix + 255) / 256), dim3(256), 0, st, pp, tiles, accum, d_film);
            RENDER_TRY(hipGetLastError());
        }
    }
    RENDER_TRY(hipEventRecord(e_end, st));
    // Nothing of this call may still run when its buffers go back to the cache: drain the stream, also after an error
    // (an asynchronous fault of the last kernels surfaces here). After the deadline the stream is presumed hung: it is
    // not waited on, the context is lost and the buffers are never reused (DevBuf).
    if (!ctx->lost) {
        const hipError_t drained = hipStreamSynchronize(st);
        if (rc == PBRT_HIP_OK && !hip_ok(ctx, drained, "hipStreamSynchronize (end of render)")) rc = PBRT_HIP_ERR_DEVICE;
    }
    if (rc == PBRT_HIP_OK) {
        float ms = 0.0f;
        RENDER_TRY(hipEventElapsedTime(&ms, e_begin, e_end));
        local.total_ms = ms;
    }
#undef RENDER_TRY
    cleanup_events();
    if (stats) *stats = local;
    return rc;
}

#ifdef PB_LANE_STATS
extern "C" int pbrt_hip_debug_wide_stats(unsigned long long* out24, int reset) try {
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(pb::g_wide_stats), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pb::g_wide_stats), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
PB_ABI_CATCH
extern "C" int pbrt_hip_debug_lane_stats(unsigned long long* out8, int reset) try {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(pb::g_lane_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pb::g_lane_stats), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
PB_ABI_CATCH
#endif
