// wf_generate_trace.h — k_generate (camera rays), the wavefront's ray IO for trace_persistent, sort keys, k_trace (part of wavefront.h)
#pragma once
#include "wf_sampler.h"

namespace pb {

__global__ void k_generate(PathState ps, Queues q, PassParams pp, DevCamera cam, TileList tiles) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = (uint32_t)pp.n_pix * pp.n_samples;
    if (p >= n) return;
    int s_local, pix;
    path_to_sample_pixel(pp, p, &s_local, &pix);
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

// ---- pbrt_hip_li: a batch of Integrator::li calls (integrator.rs:29-42). The caller made the rays (its own Camera) and
// names the random stream of each (its own Sampler's RNG::set_sequence argument, rng.rs:21-35); `skip` values were
// already drawn from that stream before li (SamplerIntegrator::render draws the 5 of the CameraSample, integrator.rs:430). ----
__global__ void k_li_generate(PathState ps, Queues q, const PbrtRay* __restrict__ rays, const uint64_t* __restrict__ keys, uint32_t n,
                              uint32_t n_padded, int skip) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_padded) return;
    int flags = 0;
    if (p < n) {
        Rng rng;
        rng_set_sequence(rng, keys[p]);
        for (int k = 0; k < skip; ++k) (void)rng_u32(rng);
        ps.rng[p] = rng.state;
        ps.samp[p] = 0;
        const float4* rp = reinterpret_cast<const float4*>(rays + p);
        float4 a = rp[0], b = rp[1];
        store_ray(ps, p, RS_CONT, V3{a.x, a.y, a.z}, V3{a.w, b.x, b.y}, b.z);
        flags = PF_VALID | PF_ALIVE;
    } else {
        store_ray(ps, p, RS_CONT, V3{0.0f, 0.0f, 0.0f}, V3{0.0f, 0.0f, 1.0f}, -1.0f);  // padding: misses at once, not a ray
    }
    ps.pfilm[p] = make_float2(0.0f, 0.0f);
    ps.L[p] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    ps.beta[p] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(flags));
    q.trace[p] = p * 4u + RS_CONT;
    q.shade[p] = p;
}
__global__ void k_li_output(PathState ps, uint32_t n, float* __restrict__ rgb) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float4 L = ps.L[p];  // what li returns; the NaN / negative-luminance guard is render's (integrator.rs:455), not li's
    rgb[3 * (size_t)p] = L.x;
    rgb[3 * (size_t)p + 1] = L.y;
    rgb[3 * (size_t)p + 2] = L.z;
}
// pbrt_hip_camera_rays: what k_generate made, in the caller's record types (path order)
__global__ void k_camera_rays_out(PathState ps, PassParams pp, TileList tiles, PbrtRay* __restrict__ rays, uint64_t* __restrict__ keys,
                                  float* __restrict__ pfilm, int32_t* __restrict__ pixel_sample) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = (uint32_t)pp.n_pix * pp.n_samples;
    if (p >= n) return;
    int s_local, pix;
    path_to_sample_pixel(pp, p, &s_local, &pix);
    int2 org = tiles.origin[pix >> 8];
    int x = org.x + (pix & 15), y = org.y + ((pix & 255) >> 4);
    bool valid = (__float_as_int(ps.beta[p].w) & PF_VALID) != 0;
    size_t ri = ray_index(ps, p, RS_CONT);
    float4 a = ps.ray[ri], b = ps.ray[ri + 1];
    float4* out = reinterpret_cast<float4*>(rays + p);
    out[0] = a;
    out[1] = make_float4(b.x, b.y, b.z, 0.0f);
    keys[p] = sample_sequence(pp, x, y, pp.sample0 + s_local);
    float2 pf = ps.pfilm[p];
    pfilm[2 * (size_t)p] = pf.x;
    pfilm[2 * (size_t)p + 1] = pf.y;
    pixel_sample[3 * (size_t)p] = valid ? x : -1;
    pixel_sample[3 * (size_t)p + 1] = valid ? y : -1;
    pixel_sample[3 * (size_t)p + 2] = pp.sample0 + s_local;
}

// ---- trace: every pending ray of the wavefront (trace_persistent.h) ----
// TWO_LEVEL: the scene has instances — hit records carry the instance slot in a second float4 (wf_state.h)
template <bool TWO_LEVEL>
struct WavefrontRayIO {
    PathState ps;
    const uint32_t* __restrict__ queue;
    uint32_t count;
    int n_segments;
    PB_DEV uint32_t n() const { return count; }
    PB_DEV int segments() const { return n_segments; }
    // The traversal kernels keep a ray's TOKEN from load to store — here the queue entry itself (path << 2 | ray slot), not
    // its position: the store then needs no second look at the queue (one dependent load less in the refill path every
    // idle lane of the wave waits in).
    PB_DEV uint32_t token(uint32_t i) const { return queue[i]; }
    PB_DEV bool strict(uint32_t e) const { return (e & 3u) == RS_MIS_BOOL; }
    PB_DEV bool load(uint32_t e, TravRay* r, bool* any) const {
        uint32_t p = e >> 2, slot = e & 3u;
        size_t ri = ray_index(ps, p, slot == RS_MIS_BOOL ? (uint32_t)RS_MIS : slot);
        float4 a = ps.ray[ri], b = ps.ray[ri + 1];
        *r = TravRay{a.x, a.y, a.z, a.w, b.x, b.y, b.z};
        *any = slot >= RS_SHADOW;  // RS_SHADOW, RS_MIS_BOOL: only the boolean is wanted
        // t_max < 0 marks the placeholder ray of a path outside pixel_bounds (k_generate): not a ray of the frame
        return !(b.z < 0.0f);
    }
    PB_DEV void store(uint32_t e, bool any, bool found, float t, float b0, float b1, float b2, int slot, int inst) const {
        uint32_t p = e >> 2, rs = e & 3u;
        if (rs == RS_MIS_BOOL) {  // `found` where the MIS ray's hit record would say it: a leaf slot >= 0 or -1
            ps.hit[hit_index(ps, p, RS_MIS)] = make_float4(__int_as_float(found ? 0 : -1), 0.0f, 0.0f, 0.0f);
            return;
        }
        size_t ri = hit_index(ps, p, rs);
        if (any) {
            ps.hit[ri] = make_float4(found ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            (void)t;
            ps.hit[ri] = make_float4(__int_as_float(found ? slot : -1), b0, b1, b2);
            if (TWO_LEVEL) ps.hit[ri + 1] = make_float4(__int_as_float(inst), 0.0f, 0.0f, 0.0f);
        }
    }
};
// Sort key of a queued ray: any-hit flag, then a 15-bit Morton code of the origin inside the scene bounds. Rays that
// start close together walk the same part of the tree: in cache order the traversal kernel runs 20 % faster on
// incoherent bounce rays (tools/probe_sorting.py), which pays for the two 8-bit radix passes.
// (the key is written by k_shade's block_append together with the queue entry, wf_state.h)

template <bool COUNT, int INST, bool SPH = false>
__global__ void __launch_bounds__(kTraceBlock, (COUNT || SPH) ? 4 : (INST ? PB_INST_WAVES : PB_TRACE_WAVES))
    k_trace(DevBVH bvh, PathState ps, const uint32_t* __restrict__ queue, uint32_t n, unsigned int* work_counter,
            unsigned long long* counters, int segments) {
    __shared__ uint2 lds_stack[kStackLds * kTraceBlock];
    WavefrontRayIO<INST != 0> io{ps, queue, n, segments};
    trace_persistent<WavefrontRayIO<INST != 0>, COUNT, INST, SPH>(bvh, io, work_counter, lds_stack + threadIdx.x,
                                                  blockIdx.x * kTraceBlock + threadIdx.x, counters);
}

// the binary records without a stack (trace_stackless.h; PBRT_TRAVERSAL_STACKLESS)
__global__ void __launch_bounds__(kTraceBlock, PB_STACKLESS_WAVES)
    k_trace_stackless(DevBVH bvh, PathState ps, const uint32_t* __restrict__ queue, uint32_t n, unsigned int* work_counter, int segments) {
    WavefrontRayIO<false> io{ps, queue, n, segments};
    trace_stackless<WavefrontRayIO<false>>(bvh, io, work_counter);
}

// the same wavefront over the 4-wide records (trace_wide.h) ...
template <bool COUNT, int INST = 0>
__global__ void __launch_bounds__(kTraceBlock, (COUNT || INST) ? PB_WIDE_INST_WAVES : PB_WIDE_WAVES)
    k_trace_wide(WideTrees wt, PathState ps, const uint32_t* __restrict__ queue, uint32_t n, unsigned int* work_counter,
                 int segments, unsigned long long* counters) {
    __shared__ uint2 lds_stack[wide_stack_lds(INST) * kTraceBlock];
    __shared__ float lds_world[INST ? kWideWorldFloats * kTraceBlock : 1];
    WavefrontRayIO<INST != 0> io{ps, queue, n, segments};
    trace_wide<WavefrontRayIO<INST != 0>, COUNT, INST>(wt, io, work_counter, lds_stack + threadIdx.x, blockIdx.x * kTraceBlock + threadIdx.x,
                                            counters, lds_world + (INST ? threadIdx.x : 0));
}
// ... and the rays it left to the binary records (axis-parallel directions and the like; usually none)
template <int INST = 0>
__global__ void __launch_bounds__(kTraceBlock, INST ? PB_INST_WAVES : PB_TRACE_WAVES)
    k_trace_special(DevBVH bvh, PathState ps, const uint32_t* __restrict__ queue, uint32_t n, const uint32_t* __restrict__ list,
                    const unsigned int* __restrict__ count, unsigned int* work_counter) {
    // Usually there is nothing to do (44 of 644 M rays in config 3), and the launch must not cost what a traversal launch
    // costs: a block whose first chunk would start past the end of the list leaves before it touches the shared queue
    // head (5000 waves x one same-address atomic each were most of the 0.22 ms an empty launch took). The blocks that stay
    // are enough: a persistent wave keeps grabbing chunks until the list is exhausted.
    if ((unsigned long long)blockIdx.x * (kTraceBlock / 64) * kChunk >= (unsigned long long)*count) return;
    __shared__ uint2 lds_stack[kStackLds * kTraceBlock];
    SpecialListIO<WavefrontRayIO<INST != 0>> io{WavefrontRayIO<INST != 0>{ps, queue, n, 1}, list, count};
    trace_persistent<SpecialListIO<WavefrontRayIO<INST != 0>>, false, INST, false>(bvh, io, work_counter, lds_stack + threadIdx.x,
                                                                        blockIdx.x * kTraceBlock + threadIdx.x, nullptr);
}

}  // namespace pb
