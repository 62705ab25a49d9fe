// wf_path.h — k_shade: PathIntegrator::li, one bounce per launch (part of wavefront.h)
#pragma once
#include "wf_lights.h"

namespace pb {

// Sort-by-material shading (north_star; the reference dispatches per hit to the material, interaction.rs:318-329 ->
// material.rs:16-55). Key of a shade-queue entry = what k_shade will branch on: 0 no continuation hit to shade (escaped /
// dead path, only the pending estimate is resolved), 1 + material type otherwise. PbrtRenderParams.shade_order selects:
//   0  queue order (one path per lane as the queue has them),
//   1  material order INSIDE each block of 256 queue entries: the block's entries are counting-sorted by key through LDS
//      before shading, so a wave shades (mostly) one material while the block still touches the same 256 paths' state —
//      the reads stay as coalesced as the queue is (round 3),
//   2  the whole queue radix-sorted by key (round 2's experiment: lane utilisation doubles, the 290 B of path state per
//      path become scattered gathers and the kernel gets slower).
// Films are bit-identical in all three: a path's arithmetic does not depend on which lane runs it.
PB_DEV uint32_t shade_key(const ShadeConsts& sc, const PathState& ps, uint32_t p, int max_depth) {
    int fb = __float_as_int(ps.beta[p].w);
    uint32_t key = 0;
    if ((fb & PF_ALIVE) && (fb >> 8) < max_depth) {
        const size_t hb = hit_index(ps, p, RS_CONT);
        int slot = __float_as_int(ps.hit[hb].x), inst = hit_instance(ps, hb);
        if (slot >= 0) {
            int mat = __float_as_int(sc.bvh.tris[3 * (size_t)slot + 2].z);
            if (sc.bvh.instanced && inst >= 0) {
                int over = __float_as_int(sc.bvh.instances[7 * (size_t)inst + 6].x);
                if (over >= 0) mat = over;
            }
            key = 1u + (uint32_t)sc.materials[mat].type;
        }
    }
    return key;
}
__global__ void k_shade_sort_keys(ShadeConsts sc, PathState ps, const uint32_t* __restrict__ shade_queue, uint32_t n, int max_depth,
                                  uint32_t* __restrict__ keys) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = shade_key(sc, ps, shade_queue[i], max_depth);
}

constexpr int kShadeKeys = 6;  // 0 nothing to shade, 1 + PBRT_MAT_* (none, matte, mirror, glass), 5 = no queue entry (block tail)
struct ShadeBins {
    uint32_t count[4][kShadeKeys];  // per wave of the block, per key
    uint32_t path[256];
};

#ifndef PB_SHADE_WAVES
#define PB_SHADE_WAVES 3  // 162 VGPRs; 4 waves (128 VGPRs, spills) measured in profiles/r04_wide_kernel_ladder.txt
#endif
template <bool BIN>
__global__ void __launch_bounds__(256, PB_SHADE_WAVES) k_shade(ShadeConsts sc, PathState ps, Queues qin, Queues qout, PassParams pp,
                                                 TileList tiles, uint32_t n_in) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n_in;
    uint32_t p = active ? qin.shade[i] : 0u;
    if (BIN) {
        // counting sort of the block's 256 entries by key: rank inside the wave by ballot + mbcnt, waves and keys through LDS
        __shared__ ShadeBins bins;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint32_t key = active ? shade_key(sc, ps, p, pp.max_depth) : (uint32_t)(kShadeKeys - 1);
        uint32_t rank = 0;
#pragma unroll
        for (int k = 0; k < kShadeKeys; ++k) {
            const unsigned long long m = __ballot(key == (uint32_t)k);
            if (key == (uint32_t)k) rank = lane_prefix(m);
            if (lane == 0) bins.count[wave][k] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        uint32_t before = 0;  // entries of smaller keys in the block, and of this key in the waves before this one
        for (int k = 0; k < kShadeKeys; ++k)
            for (int w = 0; w < 4; ++w)
                before += ((uint32_t)k < key || ((uint32_t)k == key && w < wave)) ? bins.count[w][k] : 0u;
        bins.path[before + rank] = p;
        uint32_t n_active = 0;
        for (int k = 0; k < kShadeKeys - 1; ++k)
            for (int w = 0; w < 4; ++w) n_active += bins.count[w][k];
        __syncthreads();
        p = bins.path[threadIdx.x];
        active = threadIdx.x < n_active;
    }
    bool emit_cont = false, emit_mis = false, emit_shadow = false, mis_bool = false;

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
            int hslot = __float_as_int(h0.x);
            bool found = hslot >= 0;
            Surf sf;
            if (found) {
                sf = surface_from_hit(sc.bvh, hslot, hit_instance(ps, hbase), h0.y, h0.z, h0.w, rd);
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
                            flags |= nee_flags & 0xff;
                            emit_shadow = (nee_flags & PF_NEE_SHADOW) != 0;
                            emit_mis = (nee_flags & PF_NEE_MIS) != 0;
                            mis_bool = (nee_flags & NEE_MIS_BOOL) != 0;
                        }
                    }

                    // ---- BSDF sampling for the next vertex (path.rs:123-152) ----
                    float u0 = 0.0f, u1 = 0.0f;
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
    const bool again = emit_cont || emit_mis || emit_shadow;
    block_append(sh, qout, p, emit_cont, emit_mis, emit_shadow, again, cell, mis_bool);
}

}  // namespace pb
