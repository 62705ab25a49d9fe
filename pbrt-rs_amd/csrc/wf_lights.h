// wf_lights.h — Light::sample_li for every light type, spatial light tables, estimate_direct (emit / resolve), specular lobes (part of wavefront.h)
#pragma once
#include "wf_surface.h"

namespace pb {

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
            // only an area light looks at WHAT the ray hit (integrator.rs:247-256); for the others `found` is enough
            if (sc.mis_bool && lt.type != PBRT_LIGHT_DIFFUSE_AREA) nee_flags |= NEE_MIS_BOOL;
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
        int hslot = __float_as_int(ps.hit[hit_index(ps, p, RS_MIS)].x);
        DevLight lt = sc.lights[light_id];
        V3 li = V3{0.0f, 0.0f, 0.0f};
        if (hslot >= 0) {
            // D26 (intended): Le only when the hit primitive's area light is this light (an RS_MIS_BOOL ray left 0 for
            // "found" here, not a leaf slot: it is never queued for an area light)
            int hl = lt.type == PBRT_LIGHT_DIFFUSE_AREA ? (__float_as_int(sc.bvh.tris[3 * (size_t)hslot + 2].w) & kPrimLightMask) - 1 : -2;
            if (hl == light_id) {
                // (the MIS ray's direction and the hit's barycentrics are read only here: an area light's emitter was hit)
                float4 r0 = ps.ray[ray_index(ps, p, RS_MIS)], r1 = ps.ray[ray_index(ps, p, RS_MIS) + 1];
                V3 wi = V3{r0.w, r1.x, r1.y};
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

}  // namespace pb
