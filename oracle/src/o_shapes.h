// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header / oracle/README.md).
//
// o_shapes.h — Interaction records, Shape interface, TriangleMesh / Triangle.
//
// Follows:
//   src/core/interaction.rs:100-153   BaseInteraction, spawn_ray, spawn_ray_to
//   src/core/interaction.rs:224-316   SurfaceInteraction::{new, set_shading_geometry}
//   src/core/shape.rs:15-99           trait Shape (sample2 / pdf2 defaults :38-69)
//   src/shapes/triangle.rs:17-58      TriangleMesh, Triangle::new
//   src/shapes/triangle.rs:74-158     Triangle::intersect_test (watertight test)
//   src/shapes/triangle.rs:182-316    Triangle::intersect
//   src/shapes/triangle.rs:318-348    intersect_p, area, sample
// Defect dispositions: D9, D10, D11, D13 are [Q] (as-written behind quirk
// bits); D12 (edge functions always in f64) is kept as written — it fixes the arithmetic;
// D14 (no world transform in Triangle::new, swap-handedness hard-wired false) as written:
// mesh positions are world space; D15 (mesh unconstructible) — constructor provided.
// Ray differentials (interaction.rs:338-385) only feed texture filtering and are omitted.
#pragma once
#include <memory>
#include <vector>

#include "o_sampling.h"

namespace oracle {

// src/core/interaction.rs:100-153
struct BaseInteraction {
    Point3f p;
    Float time = 0;
    Vector3f error;
    Vector3f wo;
    Normal3f n;
    // interaction.rs:132-135
    Ray spawn_ray(const Vector3f& d) const {
        Point3f origin = offset_ray_origin(p, error, n, d);
        return Ray(origin, d, FLOAT_INF, time);
    }
    // interaction.rs:147-153
    Ray spawn_ray_to(const BaseInteraction& it) const {
        Point3f origin = offset_ray_origin(p, error, n, it.p - p);
        Point3f target = offset_ray_origin(it.p, it.error, it.n, origin - it.p);
        Vector3f d = target - origin;
        return Ray(origin, d, 1.0f - SHADOW_EPSILON, time);
    }
};

struct ShadingGeom {
    Normal3f n;
    Vector3f dpdu, dpdv;
    Normal3f dndu, dndv;
};

// src/core/interaction.rs:224-245
struct SurfaceInteraction : BaseInteraction {
    Point2f uv;
    Vector3f dpdu, dpdv;
    Normal3f dndu, dndv;
    ShadingGeom shading;
    int instance_id = -1;  // TransformedPrimitive that was entered (primitive.rs:136-159), -1 = none
    int prim_id = -1;  // stands for `primitive: Option<PrimitiveDt>` (set by the aggregate; see D-note at primitive.rs:71)
    int face_index = 0;
    SurfaceInteraction() {}
    // interaction.rs:248-300
    SurfaceInteraction(const Point3f& p_, const Vector3f& err, const Point2f& uv_, const Vector3f& wo_,
                       const Vector3f& dpdu_, const Vector3f& dpdv_, const Normal3f& dndu_,
                       const Normal3f& dndv_, Float time_, int face_index_) {
        p = p_;
        n = dpdu_.cross(dpdv_).normalize();
        error = err;
        wo = wo_;
        time = time_;
        uv = uv_;
        dpdu = dpdu_;
        dpdv = dpdv_;
        dndu = dndu_;
        dndv = dndv_;
        face_index = face_index_;
        // `shading: Default::default()` in the reference; Triangle::intersect overwrites n.
        shading.dpdu = dpdu_;
        shading.dpdv = dpdv_;
        shading.dndu = dndu_;
        shading.dndv = dndv_;
    }
    // interaction.rs:302-316
    void set_shading_geometry(const Vector3f& dpdu_, const Vector3f& dpdv_, const Normal3f& dndu_,
                              const Normal3f& dndv_, bool orientation_is_authoritative) {
        shading.n = dpdu_.cross(dpdv_).normalize();
        if (orientation_is_authoritative)
            n = n.face_forward(shading.n);
        else
            shading.n = shading.n.face_forward(n);
        shading.dpdu = dpdu_;
        shading.dpdv = dpdv_;
        shading.dndu = dndu_;
        shading.dndv = dndv_;
    }
};

// src/core/shape.rs:15-99
struct Shape {
    bool reverse_orientation = false;
    bool transform_swap_handedness = false;  // shapes/mod.rs:37 hard-wires false (D14)
    virtual ~Shape() {}
    virtual Bounds3f world_bound() const = 0;
    virtual bool intersect(const Ray& ray, Float* t_hit, SurfaceInteraction* si) const = 0;
    virtual bool intersect_p(const Ray& ray) const = 0;
    virtual Float area() const = 0;
    virtual BaseInteraction sample(const Point2f& u, Float* pdf) const = 0;
    // shape.rs:38-53
    virtual BaseInteraction sample2(const BaseInteraction& ref, const Point2f& u, Float* pdf) const {
        BaseInteraction intr = sample(u, pdf);
        Vector3f wi = intr.p - ref.p;
        if (wi.length_squared() == 0.0f) {
            *pdf = 0.0f;
        } else {
            wi = wi.normalize();
            *pdf *= ref.p.distance_square(intr.p) / intr.n.abs_dot(-wi);
            if (std::isinf(*pdf)) *pdf = 0.0f;
        }
        return intr;
    }
    // shape.rs:54-69
    virtual Float pdf2(const BaseInteraction& ref, const Vector3f& wi) const {
        Ray ray = ref.spawn_ray(wi);
        Float t_hit = 0.0f;
        SurfaceInteraction isect_light;
        if (!intersect(ray, &t_hit, &isect_light)) return 0.0f;
        Float pdf = ref.p.distance_square(isect_light.p) / (isect_light.n.abs_dot(-wi) * area());
        if (std::isinf(pdf)) pdf = 0.0f;
        return pdf;
    }
};

// src/shapes/triangle.rs:17-26
struct TriangleMesh {
    std::vector<int32_t> vertex_indices;
    std::vector<Point3f> p;
    std::vector<Normal3f> n;  // optional (empty = None)
    std::vector<Vector3f> s;  // optional per-vertex tangents
    std::vector<Point2f> uv;  // optional
    int n_triangles = 0, n_vertices = 0;
};

struct TriHit {
    bool hit;
    Float b0, b1, b2, t;
};

// src/shapes/triangle.rs:74-158 — free function so KATs can call it with explicit vertices.
inline TriHit triangle_intersect_test(const Point3f& p0, const Point3f& p1, const Point3f& p2,
                                      const Ray& ray, uint32_t quirks = 0) {
    const TriHit err = {false, 0.0f, 0.0f, 0.0f, 0.0f};
    Vector3f p0t = p0 - ray.o;
    Vector3f p1t = p1 - ray.o;
    Vector3f p2t = p2 - ray.o;

    int kz = ray.d.abs().max_dimension();
    int kx = kz + 1;
    if (kx == 3) kx = 0;
    int ky = kx + 1;
    if (ky == 3) ky = 0;
    Vector3f d = ray.d.permute(kx, ky, kz);
    p0t = p0t.permute(kx, ky, kz);
    p1t = p1t.permute(kx, ky, kz);
    p2t = p2t.permute(kx, ky, kz);

    Float sx = -d.x / d.z;
    Float sy = -d.y / d.z;
    Float sz = 1.0f / d.z;
    p0t.x += sx * p0t.z;
    p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z;
    p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z;
    if (quirks & Q_D9_SHEAR_SX)
        p2t.y += sx * p2t.z;  // triangle.rs:107 as written
    else
        p2t.y += sy * p2t.z;

    // triangle.rs:109-111 (D12: always f64)
    Float e0 = (Float)((double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x);
    Float e1 = (Float)((double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x);
    Float e2 = (Float)((double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x);

    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return err;
    Float det = e0 + e1 + e2;
    if (det == 0.0f) return err;

    p0t.z *= sz;
    p1t.z *= sz;
    p2t.z *= sz;
    Float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (quirks & Q_D10_RANGE_PRECEDENCE) {
        // triangle.rs:127 as written: (det < 0 && t_scaled >= 0) || t_scaled < t_max*det
        if ((det < 0.0f && t_scaled >= 0.0f) || t_scaled < ray.t_max * det) return err;
        else if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > ray.t_max * det)) return err;
    } else {
        if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < ray.t_max * det)) return err;
        else if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > ray.t_max * det)) return err;
    }

    Float inv_det = 1.0f / det;
    Float b0 = e0 * inv_det;
    Float b1 = e1 * inv_det;
    Float b2 = e2 * inv_det;
    Float t = t_scaled * inv_det;

    // triangle.rs:139-151 (D1 disposition: max_component is a true max)
    Float max_zt = Vector3f(p0t.z, p1t.z, p2t.z).abs().max_component();
    Float delta_z = gamma(3.0f) * max_zt;
    Float max_xt = Vector3f(p0t.x, p1t.x, p2t.x).abs().max_component();
    Float max_yt = Vector3f(p0t.y, p1t.y, p2t.y).abs().max_component();
    Float delta_x = gamma(5.0f) * (max_xt + max_zt);
    Float delta_y = gamma(5.0f) * (max_yt + max_zt);
    Float delta_e;
    if (quirks & Q_D11_DELTA_E)
        delta_e = 2.0f * (gamma(2.0f) * max_xt * max_yt + delta_y * max_xt + delta_y * max_yt);
    else
        delta_e = 2.0f * (gamma(2.0f) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    Float max_e = Vector3f(e0, e1, e2).abs().max_component();
    Float delta_t =
        3.0f * (gamma(3.0f) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * std::fabs(inv_det);
    if (t <= delta_t) return err;
    TriHit h = {true, b0, b1, b2, t};
    return h;
}

// src/shapes/triangle.rs:28-58
struct Triangle : Shape {
    std::shared_ptr<TriangleMesh> mesh;
    int face_index = 0;
    int v[3];
    uint32_t quirks = 0;
    Triangle(const std::shared_ptr<TriangleMesh>& m, int tri_number, bool ro = false, uint32_t q = 0)
        : mesh(m), quirks(q) {
        reverse_orientation = ro;
        for (int i = 0; i < 3; ++i) v[i] = m->vertex_indices[3 * tri_number + i];
    }
    // triangle.rs:60-72
    void get_uvs(Point2f uv[3]) const {
        if (!mesh->uv.empty()) {
            uv[0] = mesh->uv[v[0]];
            uv[1] = mesh->uv[v[1]];
            uv[2] = mesh->uv[v[2]];
        } else {
            uv[0] = Point2f(0.0f, 0.0f);
            uv[1] = Point2f(1.0f, 0.0f);
            uv[2] = Point2f(1.0f, 1.0f);
        }
    }
    TriHit intersect_test(const Ray& ray) const {
        return triangle_intersect_test(mesh->p[v[0]], mesh->p[v[1]], mesh->p[v[2]], ray, quirks);
    }
    // triangle.rs:175-180
    Bounds3f world_bound() const override {
        return Bounds3f(mesh->p[v[0]], mesh->p[v[1]]).union_(mesh->p[v[2]]);
    }
    // triangle.rs:182-316
    bool intersect(const Ray& ray, Float* t_hit, SurfaceInteraction* si) const override {
        TriHit h = intersect_test(ray);
        if (!h.hit) return false;
        Float b0 = h.b0, b1 = h.b1, b2 = h.b2;
        const Point3f &p0 = mesh->p[v[0]], &p1 = mesh->p[v[1]], &p2 = mesh->p[v[2]];
        Point2f uv[3];
        get_uvs(uv);
        Float duv02x = uv[0].x - uv[2].x, duv02y = uv[0].y - uv[2].y;
        Float duv12x = uv[1].x - uv[2].x, duv12y = uv[1].y - uv[2].y;
        Vector3f dp02 = p0 - p2, dp12 = p1 - p2;
        Float determinant = duv02x * duv12y - duv02y * duv12x;
        bool degenerate_uv = (quirks & Q_D13_DEGENERATE_UV) ? (determinant < 1e-8f)
                                                             : (std::fabs(determinant) < 1e-8f);
        Vector3f dpdu, dpdv;
        if (!degenerate_uv) {
            Float inv_det = 1.0f / determinant;
            dpdu = (dp02 * duv12y - dp12 * duv02y) * inv_det;
            dpdv = (dp02 * -duv12x + dp12 * duv02x) * inv_det;
        }
        if (degenerate_uv || dpdu.cross(dpdv).length_squared() == 0.0f) {
            Vector3f ng = (p2 - p0).cross(p1 - p0);
            if (ng.length_squared() == 0.0f) return false;
            ng.normalize().coordinate_system(&dpdu, &dpdv);
        }
        Float x_abs_sum = std::fabs(b0 * p0.x) + std::fabs(b1 * p1.x) + std::fabs(b2 * p2.x);
        Float y_abs_sum = std::fabs(b0 * p0.y) + std::fabs(b1 * p1.y) + std::fabs(b2 * p2.y);
        Float z_abs_sum = std::fabs(b0 * p0.z) + std::fabs(b1 * p1.z) + std::fabs(b2 * p2.z);
        Vector3f p_error = Vector3f(x_abs_sum, y_abs_sum, z_abs_sum) * gamma(7.0f);
        Point3f p_hit = p0 * b0 + p1 * b1 + p2 * b2;
        Point2f uv_hit(uv[0].x * b0 + uv[1].x * b1 + uv[2].x * b2, uv[0].y * b0 + uv[1].y * b1 + uv[2].y * b2);

        *si = SurfaceInteraction(p_hit, p_error, uv_hit, -ray.d, dpdu, dpdv, Normal3f(), Normal3f(), ray.time,
                                 face_index);
        si->shading.n = dp02.cross(dp12).normalize();
        si->n = si->shading.n;
        if (reverse_orientation ^ transform_swap_handedness) {
            si->n = -si->n;
            si->shading.n = si->n;
        }
        if (!mesh->n.empty() || !mesh->s.empty()) {  // triangle.rs:252-312
            const std::vector<Normal3f>& n = mesh->n;
            Normal3f ns = si->n;
            if (!n.empty()) {
                ns = n[v[0]] * b0 + n[v[1]] * b1 + n[v[2]] * b2;
                if (ns.length_squared() > 0.0f)
                    ns = ns.normalize();
                else
                    ns = si->n;
            }
            Vector3f ss = si->dpdu.normalize();
            if (!mesh->s.empty()) {
                Vector3f si_ = mesh->s[v[0]] * b0 + mesh->s[v[1]] * b1 + mesh->s[v[2]] * b2;
                if (si_.length_squared() > 0.0f) ss = si_.normalize();
            }
            Vector3f ts = ss.cross(ns);
            if (ts.length_squared() > 0.0f) {
                ts = ts.normalize();
                ss = ts.cross(ns);
            } else {
                ns.coordinate_system(&ss, &ts);
            }
            Normal3f dndu, dndv;
            if (!n.empty()) {
                Vector3f dn1 = n[v[0]] - n[v[2]];
                Vector3f dn2 = n[v[1]] - n[v[2]];
                if (degenerate_uv) {
                    Vector3f dn = (n[v[2]] - n[v[0]]).cross(n[v[1]] - n[v[0]]);
                    if (dn.length_squared() != 0.0f) {
                        Vector3f dnu, dnv;
                        dn.coordinate_system(&dnu, &dnv);
                        dndu = dnu.normalize();
                        dndv = dnv.normalize();
                    }
                } else {
                    Float inv_det = 1.0f / determinant;
                    dndu = (dn1 * duv12y - dn2 * duv02y) * inv_det;
                    dndv = (dn1 * -duv12x + dn2 * duv02x) * inv_det;
                }
            }
            if (reverse_orientation) ts = -ts;
            si->set_shading_geometry(ss, ts, dndu, dndv, true);
        }
        *t_hit = h.t;
        return true;
    }
    bool intersect_p(const Ray& ray) const override { return intersect_test(ray).hit; }
    // triangle.rs:323-328
    Float area() const override {
        const Point3f &p0 = mesh->p[v[0]], &p1 = mesh->p[v[1]], &p2 = mesh->p[v[2]];
        return (p1 - p0).cross(p2 - p0).length() * 0.5f;
    }
    // triangle.rs:330-348
    BaseInteraction sample(const Point2f& u, Float* pdf) const override {
        Point2f b = uniform_sample_triangle(u);
        const Point3f &p0 = mesh->p[v[0]], &p1 = mesh->p[v[1]], &p2 = mesh->p[v[2]];
        BaseInteraction it;
        it.p = p0 * b.x + p1 * b.y + p2 * (1.0f - b.x - b.y);
        it.n = (p1 - p0).cross(p2 - p0).normalize();
        if (!mesh->n.empty()) {
            Normal3f ns = mesh->n[v[0]] * b.x + mesh->n[v[1]] * b.y + mesh->n[v[2]] * (1.0f - b.x - b.y);
            it.n = it.n.face_forward(ns);
        } else if (reverse_orientation ^ transform_swap_handedness) {
            it.n *= -1.0f;
        }
        Vector3f p_abs_sum = (p0 * b.x).abs() + (p1 * b.y).abs() + (p2 * (1.0f - b.x - b.y)).abs();
        it.error = p_abs_sum * gamma(6.0f);
        *pdf = 1.0f / area();
        return it;
    }
};

}  // namespace oracle
