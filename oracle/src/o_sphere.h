// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header / oracle/README.md).
//
// o_sphere.h — EFloat and the Sphere shape (BASELINE config 1: CPU-only plumbing; the device path
// handles triangles only).
//
// Follows:
//   src/core/efloat.rs:8-197            EFloat interval arithmetic, EFloat::quadratic
//   src/shapes/sphere.rs:38-92          Sphere::intersect
//   src/shapes/sphere.rs:99-201         area, sample, sample2 (cone sampling), pdf2
//   src/shapes/sphere.rs:228-284        intersect_test
//   src/shapes/mod.rs:72-91             compute_normal_differential
//   src/core/geometry.rs:1077-1137      Ray through a Transform with origin / direction error bounds
// Defect dispositions: D7 (EFloat::new never stores v) — intended; D6 (Transform * SurfaceInteraction
// is a TODO) — xform_si of o_transform.h. The hit-selection loop (sphere.rs:259-281) is kept as written.
#pragma once
#include "o_transform.h"

namespace oracle {

struct EFloat {
    Float v = 0, low = 0, high = 0;
    EFloat() {}
    EFloat(Float v_, Float err) : v(v_) {  // efloat.rs:15-25 (D7: v stored)
        if (err == 0.0f) {
            low = high = v_;
        } else {
            low = next_float_down(v_ - err);
            high = next_float_up(v_ + err);
        }
    }
    Float upper_bound() const { return high; }
    Float lower_bound() const { return low; }
    EFloat operator+(const EFloat& r) const {
        EFloat o;
        o.v = v + r.v;
        o.low = next_float_down(low + r.low);
        o.high = next_float_up(high + r.high);
        return o;
    }
    EFloat operator-(const EFloat& r) const {
        EFloat o;
        o.v = v - r.v;
        o.low = next_float_down(low - r.high);
        o.high = next_float_up(high - r.low);
        return o;
    }
    EFloat operator*(const EFloat& r) const {
        EFloat o;
        o.v = v * r.v;
        Float p[4] = {low * r.low, high * r.low, low * r.high, high * r.high};
        o.low = next_float_down(fminr(fminr(p[0], p[1]), fminr(p[2], p[3])));
        o.high = next_float_up(fmaxr(fmaxr(p[0], p[1]), fmaxr(p[2], p[3])));
        return o;
    }
    EFloat operator/(const EFloat& r) const {
        EFloat o;
        o.v = v / r.v;
        if (r.low < 0.0f && r.high > 0.0f) {
            o.low = -FLOAT_INF;
            o.high = FLOAT_INF;
        } else {
            Float d[4] = {low / r.low, high / r.low, low / r.high, high / r.high};
            o.low = next_float_down(fminr(fminr(d[0], d[1]), fminr(d[2], d[3])));
            o.high = next_float_up(fmaxr(fmaxr(d[0], d[1]), fmaxr(d[2], d[3])));
        }
        return o;
    }
    EFloat operator*(Float f) const { return *this * EFloat(f, 0.0f); }
    // efloat.rs:64-87
    static bool quadratic(const EFloat& a, const EFloat& b, const EFloat& c, EFloat* t0, EFloat* t1) {
        double discrim = (double)b.v * (double)b.v - 4.0 * (double)a.v * (double)c.v;
        if (discrim < 0.0) return false;
        double root_discrim = std::sqrt(discrim);
        EFloat float_root_discrim((Float)root_discrim, MACHINE_EPSILON * (Float)root_discrim);
        EFloat q = (b.v < 0.0f) ? (b - float_root_discrim) * -0.5f : (b + float_root_discrim) * -0.5f;
        *t0 = q / a;
        *t1 = c / q;
        if (t0->v > t1->v) std::swap(*t0, *t1);
        return true;
    }
};

// geometry.rs:1077-1137
inline Ray xform_ray_err(const Matrix4& t, const Ray& r, Vector3f* o_err, Vector3f* d_err) {
    Float x = r.o.x, y = r.o.y, z = r.o.z;
    Point3f o = xform_point(t, r.o);
    *o_err = Vector3f(std::fabs(t.m[0][0] * x) + std::fabs(t.m[0][1] * y) + std::fabs(t.m[0][2] * z) + std::fabs(t.m[0][3]),
                      std::fabs(t.m[1][0] * x) + std::fabs(t.m[1][1] * y) + std::fabs(t.m[1][2] * z) + std::fabs(t.m[1][3]),
                      std::fabs(t.m[2][0] * x) + std::fabs(t.m[2][1] * y) + std::fabs(t.m[2][2] * z) + std::fabs(t.m[2][3])) *
             gamma(3.0f);
    Float dx = r.d.x, dy = r.d.y, dz = r.d.z;
    *d_err = Vector3f(std::fabs(t.m[0][0] * dx) + std::fabs(t.m[0][1] * dy) + std::fabs(t.m[0][2] * dz),
                      std::fabs(t.m[1][0] * dx) + std::fabs(t.m[1][1] * dy) + std::fabs(t.m[1][2] * dz),
                      std::fabs(t.m[2][0] * dx) + std::fabs(t.m[2][1] * dy) + std::fabs(t.m[2][2] * dz)) *
             gamma(3.0f);
    Vector3f d = xform_vector(t, r.d);
    Float length_squared = d.length_squared();
    if (length_squared > 0.0f) {
        Float dt = d.abs().dot(*o_err) / length_squared;
        o += d * dt;
    }
    return Ray(o, d, r.t_max, r.time);
}

// shapes/mod.rs:72-91
inline void compute_normal_differential(const Vector3f& dpdu, const Vector3f& dpdv, const Vector3f& d2pduu,
                                        const Vector3f& d2pduv, const Vector3f& d2pdvv, Normal3f* dndu, Normal3f* dndv) {
    Float e = dpdu.dot(dpdu), f = dpdu.dot(dpdv), g = dpdv.dot(dpdv);
    Vector3f n = dpdu.cross(dpdv).normalize();
    Float ee = n.dot(d2pduu), ff = n.dot(d2pduv), gg = n.dot(d2pdvv);
    Float inv_egf2 = 1.0f / (e * g - f * f);
    *dndu = dpdu * inv_egf2 * (ff * f - ee * g) + dpdv * inv_egf2 * (ee * f - ff * e);
    *dndv = dpdu * inv_egf2 * (gg * f - ff * g) + dpdv * inv_egf2 * (ff * f - gg * e);
}

struct Sphere : Shape {
    Matrix4 object_to_world, world_to_object;
    Float radius, z_min, z_max, theta_min, theta_max, phi_max;
    // sphere.rs:204-225 (full sphere unless clipped); translation-only placement helper below
    Sphere(const Matrix4& o2w, const Matrix4& w2o, bool ro, Float r, Float zmin, Float zmax, Float phimax_deg)
        : object_to_world(o2w), world_to_object(w2o), radius(r) {
        reverse_orientation = ro;
        z_min = clampf(fminr(zmin, zmax), -r, r);
        z_max = clampf(fmaxr(zmin, zmax), -r, r);
        theta_min = det_acos(clampf(fminr(zmin, zmax) / r, -1.0f, 1.0f));
        theta_max = det_acos(clampf(fmaxr(zmin, zmax) / r, -1.0f, 1.0f));
        phi_max = clampf(phimax_deg, 0.0f, 360.0f) * (PI / 180.0f);
    }
    static std::shared_ptr<Sphere> at(const Point3f& c, Float r) {
        Matrix4 a, b;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) a.m[i][j] = b.m[i][j] = (i == j) ? 1.0f : 0.0f;
        a.m[0][3] = c.x; a.m[1][3] = c.y; a.m[2][3] = c.z;
        b.m[0][3] = -c.x; b.m[1][3] = -c.y; b.m[2][3] = -c.z;
        return std::make_shared<Sphere>(a, b, false, r, -r, r, 360.0f);
    }
    Bounds3f world_bound() const override {
        return xform_bounds(object_to_world, Bounds3f(Point3f(-radius, -radius, z_min), Point3f(radius, radius, z_max)));
    }
    // sphere.rs:228-284
    bool intersect_test(const Ray& r, Point3f* p_hit, Float* phi, Ray* ray_out, EFloat* t_shape_hit) const {
        Vector3f o_err, d_err;
        Ray ray = xform_ray_err(world_to_object, r, &o_err, &d_err);
        EFloat ox(ray.o.x, o_err.x), oy(ray.o.y, o_err.y), oz(ray.o.z, o_err.z);
        EFloat dx(ray.d.x, d_err.x), dy(ray.d.y, d_err.y), dz(ray.d.z, d_err.z);
        EFloat a = dx * dx + dy * dy + dz * dz;
        EFloat b = (dx * ox + dy * oy + dz * oz) * 2.0f;
        EFloat c = ox * ox + oy * oy + oz * oz - EFloat(radius, 0.0f) * EFloat(radius, 0.0f);
        EFloat t0, t1;
        if (!EFloat::quadratic(a, b, c, &t0, &t1)) return false;
        const EFloat ts[2] = {t0, t1};
        for (const EFloat& t : ts) {
            if (t.lower_bound() < 0.0f || t.upper_bound() > ray.t_max) continue;
            Point3f ph = ray.at(t.v);
            ph *= radius / ph.length();
            if (ph.x == 0.0f && ph.y == 0.0f) ph.x = 1e-5f * radius;
            Float ph_phi = det_atan2(ph.y, ph.x);
            if (ph_phi < 0.0f) ph_phi += 2.0f * PI;
            if ((z_min > -radius && ph.z < z_min) || (z_max < radius && ph.z > z_max) || ph_phi > phi_max) continue;
            *p_hit = ph;
            *phi = ph_phi;
            *ray_out = ray;
            *t_shape_hit = t;
            return true;
        }
        return false;
    }
    // sphere.rs:38-92
    bool intersect(const Ray& r, Float* t_hit, SurfaceInteraction* si) const override {
        Point3f p_hit;
        Float phi;
        Ray ray;
        EFloat t;
        if (!intersect_test(r, &p_hit, &phi, &ray, &t)) return false;
        Float u = phi / phi_max;
        Float theta = det_acos(clampf(p_hit.z / radius, -1.0f, 1.0f));
        Float v = (theta - theta_min) / (theta_max - theta_min);
        Float z_radius = std::sqrt(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
        Float inv_z_radius = 1.0f / z_radius;
        Float cos_phi = p_hit.x * inv_z_radius, sin_phi = p_hit.y * inv_z_radius;
        Vector3f dpdu(-phi_max * p_hit.y, phi_max * p_hit.x, 0.0f);
        Vector3f dpdv = Vector3f(p_hit.z * cos_phi, p_hit.z * sin_phi, -radius * det_sin(theta)) * (theta_max - theta_min);
        Vector3f d2pduu = Vector3f(p_hit.x, p_hit.y, 0.0f) * (-phi_max * phi_max);
        Vector3f d2pduv = Vector3f(-sin_phi, cos_phi, 0.0f) * (theta_max - theta_min) * p_hit.z * phi_max;
        Vector3f d2pdvv = Vector3f(p_hit.x, p_hit.y, p_hit.z) * (theta_min - theta_max) * (theta_max - theta_min);
        Normal3f dndu, dndv;
        compute_normal_differential(dpdu, dpdv, d2pduu, d2pduv, d2pdvv, &dndu, &dndv);
        Vector3f p_error = p_hit.abs() * gamma(5.0f);
        SurfaceInteraction obj(p_hit, p_error, Point2f(u, v), -ray.d, dpdu, dpdv, dndu, dndv, ray.time, 0);
        obj.shading.n = obj.n;  // SurfaceInteraction::new: shading geometry = true geometry (D47)
        if (reverse_orientation ^ transform_swap_handedness) {
            obj.n = -obj.n;
            obj.shading.n = -obj.shading.n;
        }
        *si = xform_si(object_to_world, world_to_object, obj);
        *t_hit = t.v;
        return true;
    }
    bool intersect_p(const Ray& r) const override {
        Point3f p;
        Float phi;
        Ray ray;
        EFloat t;
        return intersect_test(r, &p, &phi, &ray, &t);
    }
    Float area() const override { return phi_max * radius * (z_max - z_min); }
    // sphere.rs:103-121
    BaseInteraction sample(const Point2f& u, Float* pdf) const override {
        Float z = 1.0f - 2.0f * u.x;  // uniform_sample_sphere (sampling.rs:229-234)
        Float rr = std::sqrt(fmaxr(1.0f - z * z, 0.0f));
        Float ph = 2.0f * PI * u.y, s, c;
        det_sincos(ph, &s, &c);
        Point3f obj = Vector3f(rr * c, rr * s, z) * radius;
        BaseInteraction it;
        it.n = xform_normal(world_to_object, obj).normalize();
        if (reverse_orientation) it.n *= -1.0f;
        obj *= radius / obj.length();
        Vector3f obj_error = obj.abs() * gamma(5.0f);
        it.p = xform_point_err(object_to_world, obj, obj_error, &it.error);
        *pdf = 1.0f / area();
        return it;
    }
    // sphere.rs:123-179
    BaseInteraction sample2(const BaseInteraction& ref, const Point2f& u, Float* pdf) const override {
        Point3f p_center = xform_point(object_to_world, Point3f(0, 0, 0));
        Point3f p_origin = offset_ray_origin(ref.p, ref.error, ref.n, p_center - ref.p);
        if (p_origin.distance_square(p_center) <= radius * radius) {
            BaseInteraction intr = sample(u, pdf);
            Vector3f wi = intr.p - ref.p;
            if (wi.length_squared() == 0.0f) {
                *pdf = 0.0f;
            } else {
                wi = wi.normalize();
                *pdf *= ref.p.distance_square(intr.p) / intr.n.abs_dot(-wi);
            }
            if (std::isinf(*pdf)) *pdf = 0.0f;
            return intr;
        }
        Float dc = (ref.p - p_center).length();
        Float inv_dc = 1.0f / dc;
        Vector3f wc = (p_center - ref.p) * inv_dc, wc_x, wc_y;
        wc.coordinate_system(&wc_x, &wc_y);
        Float sin_theta_max = radius * inv_dc;
        Float sin_theta_max2 = sin_theta_max * sin_theta_max;
        Float inv_sin_theta_max = 1.0f / sin_theta_max;
        Float cos_theta_max = std::sqrt(fmaxr(1.0f - sin_theta_max2, 0.0f));
        Float cos_theta = (cos_theta_max - 1.0f) * u.x + 1.0f;
        Float sin_theta2 = 1.0f - cos_theta * cos_theta;
        if (sin_theta_max2 < 0.00068523f) {
            sin_theta2 = sin_theta_max2 * u.x;
            cos_theta = std::sqrt(1.0f - sin_theta2);
        }
        Float cos_alpha = sin_theta2 * inv_sin_theta_max +
                          cos_theta * std::sqrt(fmaxr(1.0f - sin_theta2 * inv_sin_theta_max * inv_sin_theta_max, 0.0f));
        Float sin_alpha = std::sqrt(fmaxr(1.0f - cos_alpha * cos_alpha, 0.0f));
        Float phi = u.y * 2.0f * PI, sp, cp;
        det_sincos(phi, &sp, &cp);
        // spherical_direction(sin_alpha, cos_alpha, phi, -wc_x, -wc_y, -wc) (geometry.rs:1156-1165)
        Vector3f n_world = (-wc_x) * sin_alpha * cp + (-wc_y) * sin_alpha * sp + (-wc) * cos_alpha;
        Point3f p_world = p_center + n_world * radius;
        BaseInteraction it;
        it.p = p_world;
        it.error = p_world.abs() * gamma(5.0f);
        it.n = n_world;
        if (reverse_orientation) it.n *= -1.0f;
        *pdf = 1.0f / (2.0f * PI * (1.0f - cos_theta_max));
        return it;
    }
    // sphere.rs:181-192
    Float pdf2(const BaseInteraction& ref, const Vector3f& wi) const override {
        Point3f p_center = xform_point(object_to_world, Point3f(0, 0, 0));
        Point3f p_origin = offset_ray_origin(ref.p, ref.error, ref.n, p_center - ref.p);
        if (p_origin.distance_square(p_center) < radius * radius) return Shape::pdf2(ref, wi);
        Float sin_theta_max2 = radius * radius / ref.p.distance_square(p_center);
        Float cos_theta_max = std::sqrt(fmaxr(1.0f - sin_theta_max2, 0.0f));
        return 1.0f / (2.0f * PI * (1.0f - cos_theta_max));  // uniform_cone_pdf (sampling.rs:244-246)
    }
};

}  // namespace oracle
