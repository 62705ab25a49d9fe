// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header / oracle/README.md).
//
// o_transform.h — applying a 4x4 Transform to points / vectors / normals / rays / bounds /
// SurfaceInteractions.
//
// Follows:
//   src/core/transform.rs:351-403     Transform * Point3 / Vector3 / Normal3
//   src/core/transform.rs:568-607     Transform * Bounds3f (8 corners)
//   src/core/geometry.rs:865-881      Ray through a Transform (origin error -> shift along d, t_max -= dt)
//   src/core/geometry.rs:898-1000     Point3 through a Transform with absolute error (with / without incoming error)
// Defect dispositions: the normal transform at transform.rs:387-403 indexes m_inv without the
// transpose (pbrt-v3: (M^-1)^T n) — intended (D49, found while restating); Transform * SurfaceInteraction
// (transform.rs:620-627) is a TODO that returns Default (D6) — restated from pbrt-v3.
#pragma once
#include "o_shapes.h"

namespace oracle {

struct Matrix4 {
    Float m[4][4];
};
// transform.rs:351-370
inline Point3f xform_point(const Matrix4& t, const Point3f& p) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = t.m[0][0] * x + t.m[0][1] * y + t.m[0][2] * z + t.m[0][3];
    Float yp = t.m[1][0] * x + t.m[1][1] * y + t.m[1][2] * z + t.m[1][3];
    Float zp = t.m[2][0] * x + t.m[2][1] * y + t.m[2][2] * z + t.m[2][3];
    Float wp = t.m[3][0] * x + t.m[3][1] * y + t.m[3][2] * z + t.m[3][3];
    if (wp == 1.0f) return Point3f(xp, yp, zp);
    return Point3f(xp, yp, zp) / wp;
}
// transform.rs:372-385
inline Vector3f xform_vector(const Matrix4& t, const Vector3f& v) {
    Float x = v.x, y = v.y, z = v.z;
    return Vector3f(t.m[0][0] * x + t.m[0][1] * y + t.m[0][2] * z, t.m[1][0] * x + t.m[1][1] * y + t.m[1][2] * z,
                    t.m[2][0] * x + t.m[2][1] * y + t.m[2][2] * z);
}
// geometry.rs:898-935 (point with absolute error) + :865-881 (ray through a transform)
inline Ray xform_ray(const Matrix4& t, const Ray& r) {
    Float x = r.o.x, y = r.o.y, z = r.o.z;
    Point3f o = xform_point(t, r.o);
    Float x_abs = std::fabs(t.m[0][0] * x) + std::fabs(t.m[0][1] * y) + std::fabs(t.m[0][2] * z) + std::fabs(t.m[0][3]);
    Float y_abs = std::fabs(t.m[1][0] * x) + std::fabs(t.m[1][1] * y) + std::fabs(t.m[1][2] * z) + std::fabs(t.m[1][3]);
    Float z_abs = std::fabs(t.m[2][0] * x) + std::fabs(t.m[2][1] * y) + std::fabs(t.m[2][2] * z) + std::fabs(t.m[2][3]);
    Vector3f o_error = Vector3f(x_abs, y_abs, z_abs) * gamma(3.0f);
    Vector3f d = xform_vector(t, r.d);
    Float length_squared = d.length_squared();
    Float t_max = r.t_max;
    if (length_squared > 0.0f) {
        Float dt = d.abs().dot(o_error) / length_squared;
        o += d * dt;
        t_max -= dt;
    }
    return Ray(o, d, t_max, r.time);
}

// transform.rs:387-403, intended: n' = (M^-1)^T n
inline Normal3f xform_normal(const Matrix4& m_inv, const Normal3f& n) {
    Float x = n.x, y = n.y, z = n.z;
    return Normal3f(m_inv.m[0][0] * x + m_inv.m[1][0] * y + m_inv.m[2][0] * z,
                    m_inv.m[0][1] * x + m_inv.m[1][1] * y + m_inv.m[2][1] * z,
                    m_inv.m[0][2] * x + m_inv.m[1][2] * y + m_inv.m[2][2] * z);
}

// geometry.rs:936-1000: point with incoming absolute error
inline Point3f xform_point_err(const Matrix4& t, const Point3f& pt, const Vector3f& pe, Vector3f* abs_error) {
    Float x = pt.x, y = pt.y, z = pt.z;
    Float xp = t.m[0][0] * x + t.m[0][1] * y + t.m[0][2] * z + t.m[0][3];
    Float yp = t.m[1][0] * x + t.m[1][1] * y + t.m[1][2] * z + t.m[1][3];
    Float zp = t.m[2][0] * x + t.m[2][1] * y + t.m[2][2] * z + t.m[2][3];
    Float wp = t.m[3][0] * x + t.m[3][1] * y + t.m[3][2] * z + t.m[3][3];
    const Float g3 = gamma(3.0f);
    abs_error->x = (g3 + 1.0f) * (std::fabs(t.m[0][0] * pe.x) + std::fabs(t.m[0][1] * pe.y) + std::fabs(t.m[0][2] * pe.z)) +
                   g3 * (std::fabs(t.m[0][0] * x) + std::fabs(t.m[0][1] * y) + std::fabs(t.m[0][2] * z) + std::fabs(t.m[0][3]));
    abs_error->y = (g3 + 1.0f) * (std::fabs(t.m[1][0] * pe.x) + std::fabs(t.m[1][1] * pe.y) + std::fabs(t.m[1][2] * pe.z)) +
                   g3 * (std::fabs(t.m[1][0] * x) + std::fabs(t.m[1][1] * y) + std::fabs(t.m[1][2] * z) + std::fabs(t.m[1][3]));
    abs_error->z = (g3 + 1.0f) * (std::fabs(t.m[2][0] * pe.x) + std::fabs(t.m[2][1] * pe.y) + std::fabs(t.m[2][2] * pe.z)) +
                   g3 * (std::fabs(t.m[2][0] * x) + std::fabs(t.m[2][1] * y) + std::fabs(t.m[2][2] * z) + std::fabs(t.m[2][3]));
    if (wp == 1.0f) return Point3f(xp, yp, zp);
    return Point3f(xp, yp, zp) / wp;
}

// transform.rs:568-607
inline Bounds3f xform_bounds(const Matrix4& m, const Bounds3f& b) {
    Point3f p = xform_point(m, Point3f(b.min.x, b.min.y, b.min.z));
    Bounds3f ret(p, p);
    ret = ret.union_(xform_point(m, Point3f(b.max.x, b.min.y, b.min.z)));
    ret = ret.union_(xform_point(m, Point3f(b.min.x, b.max.y, b.min.z)));
    ret = ret.union_(xform_point(m, Point3f(b.min.x, b.min.y, b.max.z)));
    ret = ret.union_(xform_point(m, Point3f(b.min.x, b.max.y, b.max.z)));
    ret = ret.union_(xform_point(m, Point3f(b.max.x, b.max.y, b.min.z)));
    ret = ret.union_(xform_point(m, Point3f(b.max.x, b.min.y, b.max.z)));
    ret = ret.union_(xform_point(m, Point3f(b.max.x, b.max.y, b.max.z)));
    return ret;
}

// pbrt-v3 Transform::operator()(const SurfaceInteraction&) (transform.rs:620-627 is a TODO, D6)
inline SurfaceInteraction xform_si(const Matrix4& m, const Matrix4& m_inv, const SurfaceInteraction& si) {
    SurfaceInteraction ret = si;
    ret.p = xform_point_err(m, si.p, si.error, &ret.error);
    ret.n = xform_normal(m_inv, si.n).normalize();
    ret.wo = xform_vector(m, si.wo).normalize();
    ret.dpdu = xform_vector(m, si.dpdu);
    ret.dpdv = xform_vector(m, si.dpdv);
    ret.dndu = xform_normal(m_inv, si.dndu);
    ret.dndv = xform_normal(m_inv, si.dndv);
    ret.shading.n = xform_normal(m_inv, si.shading.n).normalize();
    ret.shading.dpdu = xform_vector(m, si.shading.dpdu);
    ret.shading.dpdv = xform_vector(m, si.shading.dpdv);
    ret.shading.dndu = xform_normal(m_inv, si.shading.dndu);
    ret.shading.dndv = xform_normal(m_inv, si.shading.dndv);
    ret.shading.n = ret.shading.n.face_forward(ret.n);
    return ret;
}

}  // namespace oracle

