// wf_direct.h — k_shade_direct: DirectLighting / Whitted / AO integrators as a per-path stage machine (part of wavefront.h)
#pragma once
#include "wf_path.h"

namespace pb {

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
    bool emit_cont = false, emit_mis = false, emit_shadow = false, mis_bool = false;
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
            float4 h0 = ps.hit[hbase];
            int hslot = __float_as_int(h0.x);
            sf = surface_from_hit(sc.bvh, hslot, hit_instance(ps, hbase), h0.y, h0.z, h0.w, rd);
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
                int hslot = __float_as_int(ps.hit[hbase].x);
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
                    flags |= nee_flags & 0xff;
                    emit_shadow = (nee_flags & PF_NEE_SHADOW) != 0;
                    emit_mis = (nee_flags & PF_NEE_MIS) != 0;
                    mis_bool = (nee_flags & NEE_MIS_BOOL) != 0;
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
                        ds.frames[fi + 1] = make_float4(h0.z, h0.w, h0.x, __int_as_float(depth));
                        ds.frames[fi + 2] = make_float4(T.x, T.y, T.z, __int_as_float(hit_instance(ps, hbase)));
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
            ps.hit[hbase] = make_float4(f1.z, f0.w, f1.x, f1.y);
            if (ps.two_level) ps.hit[hbase + 1] = make_float4(f2.w, 0.0f, 0.0f, 0.0f);
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
    block_append(sh, qout, p, emit_cont, emit_mis, emit_shadow, emit_cont || emit_mis || emit_shadow, 0, mis_bool);
}

}  // namespace pb
