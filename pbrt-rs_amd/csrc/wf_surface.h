// wf_surface.h — SurfaceInteraction from a hit record, shading frames, matte BSDF, distributions, ShadeConsts (part of wavefront.h)
#pragma once
#include "wf_generate_trace.h"

namespace pb {

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
    // MIS rays towards lights that are not area lights are queued as RS_MIS_BOOL (wf_state.h); off while the reference's
    // loops are being counted (pbrt_hip_set_counting(1): the reference walks those rays to their closest hit)
    int mis_bool;
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
    Samp sm;
    if (pp.stream_keys) {  // a batch of Integrator::li calls: the caller's own stream per ray, RandomSampler only
        sm.rng.inc = (pp.stream_keys[p] << 1) | 1;
        sm.rng.state = ps.rng[p];
        sm.pix = 0;
        sm.s = 0;
        sm.dim1 = sm.dim2 = sm.arr = 0;
        sm.h_offset = 0;
        return sm;
    }
    int s_local, pix;
    path_to_sample_pixel(pp, p, &s_local, &pix);
    int2 org = tiles.origin[pix >> 8];
    int x = org.x + (pix & 15), y = org.y + ((pix & 255) >> 4);
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

}  // namespace pb
