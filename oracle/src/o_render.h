// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header / oracle/README.md).
//
// o_render.h — lights, Scene, light distributions, integrators, camera, film, render loop.
//
// Follows:
//   src/core/light.rs:18-72, 114-135     LightFlags, is_delta_light, trait Light, VisibilityTester::un_occluded
//   src/lights/diffuse.rs:19-90, 150-156 DiffuseAreaLight::{sample_li, pdf_li, l, power}
//   src/lights/infinite.rs:23-155        InfiniteAreaLight (constant 1x1 map only; le :84-88,
//                                        sample_li :96-129, pdf_li :140-151, power :131-133, pre_process :135-139)
//   src/core/scene.rs:11-46              Scene
//   src/core/lightdistrib.rs:21-69, 222-232  Uniform / Power light distributions
//   src/core/integrator.rs:44-90         uniform_sample_all_lights
//   src/core/integrator.rs:92-134        uniform_sample_one_light
//   src/core/integrator.rs:136-266       estimate_direct
//   src/core/integrator.rs:268-277       compute_light_power_distribution
//   src/core/integrator.rs:294-392       specular_reflect / specular_transmit (without ray differentials)
//   src/core/integrator.rs:399-480       SamplerIntegrator::render
//   src/integrators/path.rs:31-213       PathIntegrator
//   src/integrators/directlighting.rs:17-127  DirectLightingIntegrator
//   src/core/interaction.rs:387-395      SurfaceInteraction::le
//   src/cameras/perspective.rs:34-161    PerspectiveCamera
//   src/core/transform.rs:351-385        Transform * Point / Vector
//   src/core/geometry.rs:865-881, 898-935  Ray through a Transform with origin error
//   src/core/film.rs:9-123, 252-295      Film / FilmTile (box filter, src/filters/boxf.rs:14-27)
// Defect dispositions, all intended: D22 (render -> derived li), D23 (is_infinite),
// D24 (non-specular mask), D25 (hot-loop log dropped), D26 (MIS ray: Le only if the hit primitive's
// area light is this light; light.le on miss), D27 (RR q = max(0.05, 1 - max(beta))), D28
// (DirectLighting adds isect.Le), D29 (infinite_lights holds lights WITH the INFINITE flag), D30
// (delta if either bit), D31 (un_occluded = !intersect_p), D32/D33/D34, D42 (film bounds / clamp),
// D43 (box radius 0.5 x 0.5).
#pragma once
#include <unordered_map>
#include <atomic>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

#include "o_bvh.h"
#include "o_reflection.h"

namespace oracle {

enum LightFlags : uint8_t { LIGHT_DELTA_POSITION = 1, LIGHT_DELTA_DIRECTION = 2, LIGHT_AREA = 4, LIGHT_INFINITE = 8 };
// light.rs:28-31 (D30: intended)
inline bool is_delta_light(uint8_t flags) { return (flags & LIGHT_DELTA_POSITION) || (flags & LIGHT_DELTA_DIRECTION); }

struct Scene;

// light.rs:114-135
struct VisibilityTester {
    BaseInteraction p0, p1;
    bool un_occluded(const Scene& scene, TraversalCounters* ctr) const;
};

// light.rs:33-72
struct Light {
    uint8_t flags;
    int n_samples;
    explicit Light(uint8_t f, int ns = 1) : flags(f), n_samples(std::max(1, ns)) {}
    virtual ~Light() {}
    virtual Spectrum sample_li(const BaseInteraction& ref, const Point2f& u, Vector3f* wi, Float* pdf,
                               VisibilityTester* vis) const = 0;
    virtual Spectrum power() const = 0;
    virtual void pre_process(const Scene&) {}
    virtual Spectrum le(const Ray&) const { return Spectrum(0.0f); }
    virtual Float pdf_li(const BaseInteraction& ref, const Vector3f& wi) const = 0;
    virtual Spectrum l(const BaseInteraction&, const Vector3f&) const { return Spectrum(0.0f); }
};

// lights/diffuse.rs
struct DiffuseAreaLight : Light {
    Spectrum l_emit;
    std::shared_ptr<Shape> shape;
    bool two_sided;
    Float area;
    DiffuseAreaLight(const Spectrum& le, int ns, const std::shared_ptr<Shape>& s, bool two)
        : Light(LIGHT_AREA, ns), l_emit(le), shape(s), two_sided(two), area(s->area()) {}
    // diffuse.rs:60-81
    Spectrum sample_li(const BaseInteraction& ref, const Point2f& u, Vector3f* wi, Float* pdf,
                       VisibilityTester* vis) const override {
        BaseInteraction p_shape = shape->sample2(ref, u, pdf);
        if (*pdf == 0.0f || (p_shape.p - ref.p).length_squared() == 0.0f) {
            *pdf = 0.0f;
            return Spectrum(0.0f);
        }
        *wi = (p_shape.p - ref.p).normalize();
        vis->p0 = ref;
        vis->p1 = p_shape;
        return l(p_shape, -*wi);
    }
    // diffuse.rs:83-85
    Spectrum power() const override { return l_emit * ((two_sided ? 2.0f : 1.0f) * area * PI); }
    // diffuse.rs:87-90
    Float pdf_li(const BaseInteraction& ref, const Vector3f& wi) const override { return shape->pdf2(ref, wi); }
    // diffuse.rs:150-156
    Spectrum l(const BaseInteraction& si, const Vector3f& w) const override {
        return (two_sided || si.n.dot(w) > 0.0f) ? l_emit : Spectrum(0.0f);
    }
};

// geometry.rs:1196-1209
inline Float spherical_theta(const Vector3f& v) { return det_acos(clampf(v.z, -1.0f, 1.0f)); }
inline Float spherical_phi(const Vector3f& v) {
    Float p = det_atan2(v.y, v.x);
    return p < 0.0f ? p + 2.0f * PI : p;
}

// lights/infinite.rs with a 1x1 (constant) map and identity light_to_world. The reference's
// constructor builds a 2x2 sin-weighted Distribution2D from the 1x1 MIPMap (:59-73); the
// MIPMap's bilinear lookup of a single texel (mipmap.rs:283-295) is taken as that texel.
struct InfiniteAreaLight : Light {
    Spectrum l_const;
    Point3f world_center;
    Float world_radius = 0.0f;
    Distribution2D distribution;
    InfiniteAreaLight(const Spectrum& power, int ns) : Light(LIGHT_INFINITE, ns), l_const(power) {
        const int width = 2, height = 2;
        Float img[width * height];
        for (int v = 0; v < height; ++v) {
            Float vp = ((Float)v + 0.5f) / (Float)height;
            Float sin_theta = det_sin(PI * vp);
            for (int u = 0; u < width; ++u) {
                img[u + v * width] = l_const.y_value();
                img[u + v * width] *= sin_theta;
            }
        }
        distribution = Distribution2D(img, width, height);
    }
    // infinite.rs:84-88
    Spectrum le(const Ray&) const override { return l_const; }
    // infinite.rs:96-129
    Spectrum sample_li(const BaseInteraction& ref, const Point2f& u, Vector3f* wi, Float* pdf,
                       VisibilityTester* vis) const override {
        Float map_pdf = 0.0f;
        Point2f uv = distribution.sample_continuous(u, &map_pdf);
        if (map_pdf == 0.0f) return Spectrum(0.0f);
        Float theta = uv.y * PI, phi = uv.x * 2.0f * PI;
        Float cos_theta, sin_theta, sin_phi, cos_phi;
        det_sincos(theta, &sin_theta, &cos_theta);
        det_sincos(phi, &sin_phi, &cos_phi);
        *wi = Vector3f(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
        *pdf = map_pdf / (2.0f * PI * PI * sin_theta);
        if (sin_theta == 0.0f) *pdf = 0.0f;
        vis->p0 = ref;
        vis->p1 = BaseInteraction();
        vis->p1.p = ref.p + *wi * (2.0f * world_radius);
        vis->p1.time = ref.time;
        return l_const;
    }
    // infinite.rs:131-133
    Spectrum power() const override { return l_const * (PI * world_radius * world_radius); }
    // infinite.rs:135-139
    void pre_process(const Scene& scene) override;
    // infinite.rs:140-151
    Float pdf_li(const BaseInteraction&, const Vector3f& w) const override {
        Float theta = spherical_theta(w), phi = spherical_phi(w);
        Float sin_theta = det_sin(theta);
        if (sin_theta == 0.0f) return 0.0f;
        return distribution.pdf(Point2f(phi * INV_2_PI, theta * INV_PI)) / (2.0f * PI * PI * sin_theta);
    }
};

// lights/point.rs:17-63. p_light = light_to_world * (0,0,0) is computed by the caller (point.rs:27).
struct PointLight : Light {
    Point3f p_light;
    Spectrum i;
    PointLight(const Point3f& p, const Spectrum& i_) : Light(LIGHT_DELTA_POSITION, 1), p_light(p), i(i_) {}
    // point.rs:47-63
    Spectrum sample_li(const BaseInteraction& ref, const Point2f&, Vector3f* wi, Float* pdf,
                       VisibilityTester* vis) const override {
        *wi = (p_light - ref.p).normalize();
        *pdf = 1.0f;
        vis->p0 = ref;
        vis->p1 = BaseInteraction();
        vis->p1.p = p_light;
        vis->p1.time = ref.time;
        return i / (p_light - ref.p).length_squared();
    }
    Spectrum power() const override { return i * (4.0f * PI); }  // point.rs:65-67
    Float pdf_li(const BaseInteraction&, const Vector3f&) const override { return 0.0f; }
};

// lights/spot.rs:18-92. cos_total_width / cos_falloff_start = cos(radians(..)) (spot.rs:38-39) and the
// upper 3x3 of world_to_light come from the caller.
// D52 (intended): falloff() reads wl.z of the un-normalised world_to_light * w (spot.rs:50-51); pbrt-v3
// normalises wl first (spot.cpp:64), which matters when light_to_world carries a scale.
struct SpotLight : Light {
    Point3f p_light;
    Spectrum i;
    Float cos_total_width, cos_falloff_start;
    Float w2l[9];
    SpotLight(const Point3f& p, const Spectrum& i_, Float ctw, Float cfs, const Float* m)
        : Light(LIGHT_DELTA_POSITION, 1), p_light(p), i(i_), cos_total_width(ctw), cos_falloff_start(cfs) {
        for (int k = 0; k < 9; ++k) w2l[k] = m[k];
    }
    // spot.rs:49-62
    Float falloff(const Vector3f& w) const {
        Vector3f wl(w2l[0] * w.x + w2l[1] * w.y + w2l[2] * w.z, w2l[3] * w.x + w2l[4] * w.y + w2l[5] * w.z,
                    w2l[6] * w.x + w2l[7] * w.y + w2l[8] * w.z);
        wl = wl.normalize();
        Float cos_theta = wl.z;
        if (cos_theta < cos_total_width) return 0.0f;
        if (cos_theta >= cos_falloff_start) return 1.0f;
        Float delta = (cos_theta - cos_total_width) / (cos_falloff_start - cos_total_width);
        return (delta * delta) * (delta * delta);
    }
    // spot.rs:70-88
    Spectrum sample_li(const BaseInteraction& ref, const Point2f&, Vector3f* wi, Float* pdf,
                       VisibilityTester* vis) const override {
        *wi = (p_light - ref.p).normalize();
        *pdf = 1.0f;
        vis->p0 = ref;
        vis->p1 = BaseInteraction();
        vis->p1.p = p_light;
        vis->p1.time = ref.time;
        return i * falloff(-*wi) / (p_light - ref.p).length_squared();
    }
    // spot.rs:90-92
    Spectrum power() const override { return i * (2.0f * PI * (1.0f - 0.5f * (cos_falloff_start + cos_total_width))); }
    Float pdf_li(const BaseInteraction&, const Vector3f&) const override { return 0.0f; }
};

// lights/distant.rs:18-87. w_light = normalize(light_to_world * w) comes from the caller (distant.rs:30).
struct DistantLight : Light {
    Spectrum l_emit;
    Vector3f w_light;
    Point3f world_center;
    Float world_radius = 0.0f;
    DistantLight(const Spectrum& l, const Vector3f& w) : Light(LIGHT_DELTA_DIRECTION, 1), l_emit(l), w_light(w) {}
    // distant.rs:54-74
    Spectrum sample_li(const BaseInteraction& ref, const Point2f&, Vector3f* wi, Float* pdf,
                       VisibilityTester* vis) const override {
        *wi = w_light;
        *pdf = 1.0f;
        vis->p0 = ref;
        vis->p1 = BaseInteraction();
        vis->p1.p = ref.p + w_light * (2.0f * world_radius);
        vis->p1.time = ref.time;
        return l_emit;
    }
    Spectrum power() const override { return l_emit * (PI * world_radius * world_radius); }  // distant.rs:76-78
    void pre_process(const Scene& scene) override;                                            // distant.rs:80-84
    Float pdf_li(const BaseInteraction&, const Vector3f&) const override { return 0.0f; }
};

// scene.rs:11-46
struct Scene {
    std::vector<std::shared_ptr<Light>> lights;
    std::vector<std::shared_ptr<Light>> infinite_lights;
    std::shared_ptr<BVHAccel> aggregate;
    Bounds3f world_bound;
    std::vector<MaterialDesc> materials;
    // per caller-order primitive: material / area light ids (GeometricPrimitive fields)
    std::vector<int> prim_material, prim_light;
    std::vector<int> instance_material;  // per instance: material of every primitive inside it, or -1 = the primitive's own
    uint32_t quirks = 0;
    int material_of(const SurfaceInteraction& si) const {
        if (si.instance_id >= 0 && instance_material[si.instance_id] >= 0) return instance_material[si.instance_id];
        return prim_material[si.prim_id];
    }
    void finish() {
        world_bound = aggregate->world_bound();
        for (auto& l : lights) {
            l->pre_process(*this);
            if (l->flags & LIGHT_INFINITE) infinite_lights.push_back(l);  // D29: intended
        }
    }
    bool intersect(const Ray& ray, SurfaceInteraction* isect, TraversalCounters* ctr) const {
        tl_counters() = ctr;
        return aggregate->intersect(ray, isect);
    }
    bool intersect_p(const Ray& ray, TraversalCounters* ctr) const {
        tl_counters() = ctr;
        return aggregate->intersect_p(ray);
    }
};
inline void InfiniteAreaLight::pre_process(const Scene& scene) {
    scene.world_bound.bounding_sphere(&world_center, &world_radius);
}
inline void DistantLight::pre_process(const Scene& scene) {
    scene.world_bound.bounding_sphere(&world_center, &world_radius);
}
// light.rs:126-135 (D31: intended negation)
inline bool VisibilityTester::un_occluded(const Scene& scene, TraversalCounters* ctr) const {
    return !scene.intersect_p(p0.spawn_ray_to(p1), ctr);
}

// interaction.rs:387-395
inline Spectrum surface_le(const Scene& scene, const SurfaceInteraction& si, const Vector3f& w) {
    // instanced primitives carry no area lights (pbrt-v3: "area lights not supported with object instancing")
    int lid = (si.prim_id >= 0 && si.instance_id < 0) ? scene.prim_light[si.prim_id] : -1;
    if (lid >= 0) return scene.lights[lid]->l(si, w);
    return Spectrum(0.0f);
}

// integrator.rs:268-277
inline std::shared_ptr<Distribution1D> compute_light_power_distribution(const Scene& scene) {
    if (scene.lights.empty()) return nullptr;
    std::vector<Float> light_power;
    for (auto& l : scene.lights) light_power.push_back(l->power().y_value());
    return std::make_shared<Distribution1D>(light_power.data(), (int)light_power.size());
}
// lightdistrib.rs:21-24, 222-232: LightDistribution::lookup(p). "uniform" / "power" hold one Distribution1D;
// "spatial" (lightdistrib.rs:76-220) one per voxel of a grid over the scene bounds, filled in on first use.
// D57 (intended): compute_distribution's upper corner is written `pi as Float + 1.0 / n_voxel`
// (lightdistrib.rs:117-121), i.e. pi + 1/n instead of (pi + 1)/n, which puts the voxel's samples outside the scene
// for every pi > 0 -> pbrt-v3 `(pi + 1) / nVoxels`. Its sample points use radical_inverse (D53 intended).
struct LightDistribution {
    std::shared_ptr<Distribution1D> fixed;  // uniform / power
    // spatial
    const Scene* scene = nullptr;
    int n_voxel[3] = {1, 1, 1};
    mutable std::mutex mutex;
    mutable std::unordered_map<uint64_t, std::shared_ptr<Distribution1D>> voxels;
    static Float lerp1(Float t, Float a, Float b) { return (1.0f - t) * a + t * b; }  // pbrt.rs:224-226
    // SpatialLightDistribution::new (lightdistrib.rs:85-107), max_voxels = 64 (:228)
    void init_spatial(const Scene& sc, int max_voxels) {
        scene = &sc;
        Vector3f diag = sc.world_bound.diagonal();
        Float b_max = diag[sc.world_bound.maximum_extent()];
        for (int i = 0; i < 3; ++i) n_voxel[i] = std::max(1, (int)std::round(diag[i] / b_max * (Float)max_voxels));
    }
    // lightdistrib.rs:109-163
    std::shared_ptr<Distribution1D> compute_distribution(const int pi[3]) const {
        const Bounds3f& wb = scene->world_bound;
        Point3f p0((Float)pi[0] / (Float)n_voxel[0], (Float)pi[1] / (Float)n_voxel[1], (Float)pi[2] / (Float)n_voxel[2]);
        Point3f p1((Float)(pi[0] + 1) / (Float)n_voxel[0], (Float)(pi[1] + 1) / (Float)n_voxel[1],
                   (Float)(pi[2] + 1) / (Float)n_voxel[2]);
        auto lerp_b = [&](const Bounds3f& b, const Point3f& t) {
            return Point3f(lerp1(t.x, b.min.x, b.max.x), lerp1(t.y, b.min.y, b.max.y), lerp1(t.z, b.min.z, b.max.z));
        };
        Bounds3f voxel_bounds(lerp_b(wb, p0), lerp_b(wb, p1));
        const int n_samples = 128;
        std::vector<Float> light_contrib(scene->lights.size(), 0.0f);
        for (int i = 0; i < n_samples; ++i) {
            Point3f po = lerp_b(voxel_bounds, Point3f(radical_inverse(0, (uint64_t)i), radical_inverse(1, (uint64_t)i),
                                                      radical_inverse(2, (uint64_t)i)));
            BaseInteraction intr;
            intr.p = po;
            intr.wo = Vector3f(1.0f, 0.0f, 0.0f);
            Point2f u(radical_inverse(3, (uint64_t)i), radical_inverse(4, (uint64_t)i));
            for (size_t j = 0; j < scene->lights.size(); ++j) {
                Float pdf = 0.0f;
                Vector3f wi;
                VisibilityTester vis;
                Spectrum li = scene->lights[j]->sample_li(intr, u, &wi, &pdf, &vis);
                if (pdf > 0.0f) light_contrib[j] += li.y_value() / pdf;
            }
        }
        Float sum_contrib = 0.0f;
        for (Float c : light_contrib) sum_contrib += c;
        Float avg_contrib = sum_contrib / (Float)(n_samples * (int)light_contrib.size());
        Float min_contrib = avg_contrib > 0.0f ? 0.001f * avg_contrib : 1.0f;
        for (Float& c : light_contrib) c = fmaxr(c, min_contrib);
        return std::make_shared<Distribution1D>(light_contrib.data(), (int)light_contrib.size());
    }
    // lightdistrib.rs:43-46, 66-69, 171-219 (the hash table is a cache: any map from the voxel to its distribution does)
    const Distribution1D* lookup(const Point3f& p) const {
        if (fixed || !scene) return fixed.get();
        Vector3f offset = scene->world_bound.offset(p);
        int pi[3];
        for (int i = 0; i < 3; ++i) pi[i] = std::min(std::max((int)(offset[i] * (Float)n_voxel[i]), 0), n_voxel[i] - 1);
        uint64_t packed = ((uint64_t)pi[0] << 40) | ((uint64_t)pi[1] << 20) | (uint64_t)pi[2];
        std::lock_guard<std::mutex> guard(mutex);
        auto it = voxels.find(packed);
        if (it == voxels.end()) it = voxels.emplace(packed, compute_distribution(pi)).first;
        return it->second.get();
    }
};
inline std::shared_ptr<LightDistribution> create_light_sample_distribution(const std::string& name, const Scene& scene) {
    auto ld = std::make_shared<LightDistribution>();
    if (scene.lights.empty()) return ld;
    if (name == "uniform" || scene.lights.size() == 1) {
        std::vector<Float> prob(scene.lights.size(), 1.0f);
        ld->fixed = std::make_shared<Distribution1D>(prob.data(), (int)prob.size());
    } else if (name == "power") {
        ld->fixed = compute_light_power_distribution(scene);
    } else {
        ld->init_spatial(scene, 64);
    }
    return ld;
}

struct RenderCtx {  // per-thread state threaded through li()
    Sampler sampler;
    TraversalCounters ctr;
};

// integrator.rs:136-266
inline Spectrum estimate_direct(const SurfaceInteraction& it, const BSDF& bsdf, const Point2f& u_scattering,
                                const Light& light, int light_id, const Point2f& u_light, const Scene& scene,
                                RenderCtx& rc, bool specular = false) {
    uint8_t bsdf_flags = specular ? (uint8_t)BSDF_ALL : (uint8_t)(BSDF_ALL & ~BSDF_SPECULAR);  // D24
    Spectrum ld(0.0f);
    Vector3f wi;
    Float light_pdf = 0.0f, scattering_pdf = 0.0f;
    VisibilityTester visibility;
    Spectrum li = light.sample_li(it, u_light, &wi, &light_pdf, &visibility);
    if (light_pdf > 0.0f && !li.is_black()) {
        scattering_pdf = bsdf.pdf(it.wo, wi, bsdf_flags);
        Spectrum f = bsdf.f(it.wo, wi, bsdf_flags) * wi.abs_dot(it.shading.n);
        if (!f.is_black()) {
            if (!visibility.un_occluded(scene, &rc.ctr)) li = Spectrum(0.0f);
            if (!li.is_black()) {
                if (is_delta_light(light.flags)) {
                    ld += li * f / light_pdf;
                } else {
                    Float weight = power_heuristic(1, light_pdf, 1, scattering_pdf);
                    ld += li * f * weight / light_pdf;
                }
            }
        }
    }
    if (!is_delta_light(light.flags)) {
        uint8_t sampled_type = BSDF_NONE;
        Spectrum f = bsdf.sample_f(it.wo, &wi, u_scattering, &scattering_pdf, bsdf_flags, &sampled_type);
        f *= wi.abs_dot(it.shading.n);
        bool sampled_specular = (sampled_type & BSDF_SPECULAR) != 0;
        if (!f.is_black() && scattering_pdf > 0.0f) {
            Float weight = 1.0f;
            if (!sampled_specular) {
                light_pdf = light.pdf_li(it, wi);
                if (light_pdf == 0.0f) return ld;
                weight = power_heuristic(1, scattering_pdf, 1, light_pdf);
            }
            SurfaceInteraction light_isect;
            Ray ray = it.spawn_ray(wi);
            bool found = scene.intersect(ray, &light_isect, &rc.ctr);
            Spectrum li2(0.0f);
            if (found) {
                // D26: Le only if the hit primitive's area light is this light
                if (light_isect.prim_id >= 0 && light_isect.instance_id < 0 &&
                    scene.prim_light[light_isect.prim_id] == light_id)
                    li2 = surface_le(scene, light_isect, -wi);
            } else {
                li2 = light.le(ray);
            }
            if (!li2.is_black()) ld += li2 * f * weight / scattering_pdf;
        }
    }
    return ld;
}

// integrator.rs:92-134
inline Spectrum uniform_sample_one_light(const SurfaceInteraction& it, const BSDF& bsdf, const Scene& scene,
                                         RenderCtx& rc, const Distribution1D* light_distrib) {
    int n_lights = (int)scene.lights.size();
    if (n_lights == 0) return Spectrum(0.0f);
    int light_num;
    Float light_pdf;
    if (light_distrib) {
        light_num = light_distrib->sample_discrete(rc.sampler.get_1d(), &light_pdf);
        if (light_pdf == 0.0f) return Spectrum(0.0f);
    } else {
        light_num = (int)fminr(rc.sampler.get_1d() * (Float)n_lights, (Float)n_lights - 1.0f);
        light_pdf = 1.0f / (Float)n_lights;
    }
    const Light& light = *scene.lights[light_num];
    Point2f u_light = rc.sampler.get_2d();
    Point2f u_scattering = rc.sampler.get_2d();
    return estimate_direct(it, bsdf, u_scattering, light, light_num, u_light, scene, rc) / light_pdf;
}

// integrator.rs:44-90. The reference pre-requests per-light sample arrays in pre_process
// (directlighting.rs:58-78) and RandomSampler::start_pixel fills them for all spp at once
// (random.rs:29-42); with per-(pixel,sample) streams the entries are drawn on demand, in the
// same per-light (u_light[k], u_scatter[k]) order.
inline Spectrum uniform_sample_all_lights(const SurfaceInteraction& it, const BSDF& bsdf, const Scene& scene,
                                          RenderCtx& rc, const std::vector<int>& n_light_samples) {
    Spectrum l(0.0f);
    for (size_t j = 0; j < scene.lights.size(); ++j) {
        const Light& light = *scene.lights[j];
        int n_samples = n_light_samples[j];
        if (rc.sampler.tabulated()) {
            // integrator.rs:55-89: the two requested arrays, or ONE get_2d pair once they have run out
            const Point2f* u_light_array = rc.sampler.get_2d_array(n_samples);
            const Point2f* u_scattering_array = rc.sampler.get_2d_array(n_samples);
            if (!u_light_array || !u_scattering_array) {
                Point2f u_light = rc.sampler.get_2d();
                Point2f u_scattering = rc.sampler.get_2d();
                l += estimate_direct(it, bsdf, u_scattering, light, (int)j, u_light, scene, rc);
            } else {
                Spectrum ld(0.0f);
                for (int k = 0; k < n_samples; ++k)
                    ld += estimate_direct(it, bsdf, u_scattering_array[k], light, (int)j, u_light_array[k], scene, rc);
                l += ld / (Float)n_samples;
            }
            continue;
        }
        Spectrum ld(0.0f);
        for (int k = 0; k < n_samples; ++k) {
            Point2f u_light = rc.sampler.get_2d();
            Point2f u_scattering = rc.sampler.get_2d();
            ld += estimate_direct(it, bsdf, u_scattering, light, (int)j, u_light, scene, rc);
        }
        l += ld / (Float)n_samples;
    }
    return l;
}

// integrator.rs:29-42
struct Integrator {
    virtual ~Integrator() {}
    virtual void pre_process(const Scene&) {}
    // the sample-array requests an integrator makes of the sampler in its constructor / pre_process
    virtual void request_samples(const Scene&, SamplerSpec&) {}
    virtual Spectrum li(Ray ray, const Scene& scene, RenderCtx& rc, int depth) const = 0;
};

// integrators/path.rs
struct PathIntegrator : Integrator {
    int max_depth;
    Float rr_threshold;
    std::string light_sample_strategy;
    std::shared_ptr<LightDistribution> light_distribution;
    PathIntegrator(int md, Float rr, const std::string& strat)
        : max_depth(md), rr_threshold(rr), light_sample_strategy(strat) {}
    void pre_process(const Scene& scene) override {
        light_distribution = create_light_sample_distribution(light_sample_strategy, scene);
    }
    // path.rs:65-213
    Spectrum li(Ray ray, const Scene& scene, RenderCtx& rc, int) const override {
        Spectrum l(0.0f), beta(1.0f);
        bool specular_bounce = false;
        int bounces = 0;
        Float eta_scale = 1.0f;
        for (;;) {
            SurfaceInteraction isect;
            bool found = scene.intersect(ray, &isect, &rc.ctr);
            if (bounces == 0 || specular_bounce) {
                if (found) {
                    l += beta * surface_le(scene, isect, -ray.d);
                } else {
                    for (auto& light : scene.infinite_lights) l += beta * light->le(ray);
                }
            }
            if (!found || bounces >= max_depth) break;
            const MaterialDesc& mat = scene.materials[scene.material_of(isect)];
            std::shared_ptr<BSDF> bsdf = compute_scattering_functions(mat, isect, MODE_RADIANCE, true, scene.quirks);
            if (!bsdf) {
                ray = isect.spawn_ray(ray.d);
                continue;
            }
            if (bsdf->num_components((uint8_t)(BSDF_ALL & ~BSDF_SPECULAR)) > 0) {
                const Distribution1D* distrib = light_distribution->lookup(isect.p);  // path.rs:115
                Spectrum ld = beta * uniform_sample_one_light(isect, *bsdf, scene, rc, distrib);
                l += ld;
            }
            Vector3f wo = -ray.d, wi;
            Float pdf = 0.0f;
            uint8_t flags = BSDF_NONE;
            Spectrum f = bsdf->sample_f(wo, &wi, rc.sampler.get_2d(), &pdf, BSDF_ALL, &flags);
            if (f.is_black() || pdf == 0.0f) break;
            beta *= f * (wi.abs_dot(isect.shading.n) / pdf);
            specular_bounce = (flags & BSDF_SPECULAR) != 0;
            if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                Float eta = bsdf->eta;
                eta_scale *= (wo.dot(isect.n) > 0.0f) ? (eta * eta) : 1.0f / (eta * eta);
            }
            ray = isect.spawn_ray(wi);
            // path.rs:154-198 BSSRDF branch: no material sets a bssrdf.
            Spectrum rr_beta = beta * eta_scale;
            if (rr_beta.max_component_value() < rr_threshold && bounces > 3) {
                Float q = fmaxr(0.05f, 1.0f - rr_beta.max_component_value());  // D27
                if (rc.sampler.get_1d() < q) break;
                beta /= 1.0f - q;
            }
            bounces += 1;
        }
        return l;
    }
};

// integrators/directlighting.rs
enum LightStrategy { UNIFORM_SAMPLE_ALL = 0, UNIFORM_SAMPLE_ONE = 1 };
struct DirectLightingIntegrator : Integrator {
    LightStrategy strategy;
    int max_depth;
    std::vector<int> n_light_samples;
    DirectLightingIntegrator(LightStrategy s, int md) : strategy(s), max_depth(md) {}
    // directlighting.rs:58-78
    void pre_process(const Scene& scene) override {
        n_light_samples.clear();
        if (strategy == UNIFORM_SAMPLE_ALL)
            for (auto& l : scene.lights) n_light_samples.push_back(l->n_samples);
    }
    // directlighting.rs:58-76: round_count, then two arrays per light for each of max_depth vertices
    void request_samples(const Scene& scene, SamplerSpec& spec) override {
        if (strategy != UNIFORM_SAMPLE_ALL) return;
        n_light_samples.clear();
        for (auto& l : scene.lights) n_light_samples.push_back(spec.round_count(l->n_samples));
        for (int i = 0; i < max_depth; ++i)
            for (size_t j = 0; j < scene.lights.size(); ++j) {
                spec.arrays_2d.push_back(n_light_samples[j]);
                spec.arrays_2d.push_back(n_light_samples[j]);
            }
    }
    // directlighting.rs:79-127
    Spectrum li(Ray ray, const Scene& scene, RenderCtx& rc, int depth) const override {
        Spectrum l(0.0f);
        SurfaceInteraction isect;
        if (!scene.intersect(ray, &isect, &rc.ctr)) {
            for (auto& light : scene.lights) l += light->le(ray);
            return l;
        }
        const MaterialDesc& mat = scene.materials[scene.material_of(isect)];
        std::shared_ptr<BSDF> bsdf = compute_scattering_functions(mat, isect, MODE_RADIANCE, false, scene.quirks);
        if (!bsdf) return li(isect.spawn_ray(ray.d), scene, rc, depth);
        Vector3f wo = isect.wo;
        l += surface_le(scene, isect, wo);  // D28
        if (!scene.lights.empty()) {
            if (strategy == UNIFORM_SAMPLE_ALL)
                l += uniform_sample_all_lights(isect, *bsdf, scene, rc, n_light_samples);
            else
                l += uniform_sample_one_light(isect, *bsdf, scene, rc, nullptr);
        }
        if (depth + 1 < max_depth) {
            l += specular_bounce(ray, isect, *bsdf, scene, rc, depth, (uint8_t)(BSDF_REFLECTION | BSDF_SPECULAR));
            l += specular_bounce(ray, isect, *bsdf, scene, rc, depth, (uint8_t)(BSDF_TRANSMISSION | BSDF_SPECULAR));
        }
        return l;
    }
    // integrator.rs:294-334 specular_reflect, :336-392 specular_transmit (differentials dropped)
    Spectrum specular_bounce(const Ray&, const SurfaceInteraction& isect, const BSDF& bsdf, const Scene& scene,
                             RenderCtx& rc, int depth, uint8_t type) const {
        Vector3f wo = isect.wo, wi;
        Float pdf = 0.0f;
        uint8_t sampled = BSDF_NONE;
        Spectrum f = bsdf.sample_f(wo, &wi, rc.sampler.get_2d(), &pdf, type, &sampled);
        const Normal3f& ns = isect.shading.n;
        if (pdf > 0.0f && !f.is_black() && wi.abs_dot(ns) != 0.0f) {
            Ray rd = isect.spawn_ray(wi);
            return f * li(rd, scene, rc, depth + 1) * wi.abs_dot(ns) / pdf;
        }
        return Spectrum(0.0f);
    }
};

// integrators/whitted.rs:47-98. One Light::sample_li per light with BSDF::f over all lobes (no MIS, no
// BSDF sampling), emission at every vertex, then the specular_reflect / specular_transmit recursion of
// integrator.rs:294-392 shared with DirectLightingIntegrator.
struct WhittedIntegrator : DirectLightingIntegrator {
    WhittedIntegrator(int md) : DirectLightingIntegrator(UNIFORM_SAMPLE_ONE, md) {}
    void pre_process(const Scene&) override {}
    void request_samples(const Scene&, SamplerSpec&) override {}
    Spectrum li(Ray ray, const Scene& scene, RenderCtx& rc, int depth) const override {
        Spectrum l(0.0f);
        SurfaceInteraction isect;
        if (!scene.intersect(ray, &isect, &rc.ctr)) {
            for (auto& light : scene.lights) l += light->le(ray);  // whitted.rs:57-62
            return l;
        }
        const Normal3f n = isect.shading.n;
        Vector3f wo = isect.wo;
        const MaterialDesc& mat = scene.materials[scene.material_of(isect)];
        std::shared_ptr<BSDF> bsdf = compute_scattering_functions(mat, isect, MODE_RADIANCE, false, scene.quirks);
        if (!bsdf) return li(isect.spawn_ray(ray.d), scene, rc, depth);  // whitted.rs:68-71
        l += surface_le(scene, isect, wo);
        for (auto& light : scene.lights) {  // whitted.rs:75-91
            Vector3f wi;
            Float pdf = 0.0f;
            VisibilityTester visibility;
            Spectrum li_ = light->sample_li(isect, rc.sampler.get_2d(), &wi, &pdf, &visibility);
            if (li_.is_black() || pdf == 0.0f) continue;
            Spectrum f = bsdf->f(wo, wi, BSDF_ALL);
            if (!f.is_black() && visibility.un_occluded(scene, &rc.ctr)) l += f * li_ * wi.abs_dot(n) / pdf;
        }
        if (depth + 1 < max_depth) {
            l += specular_bounce(ray, isect, *bsdf, scene, rc, depth, (uint8_t)(BSDF_REFLECTION | BSDF_SPECULAR));
            l += specular_bounce(ray, isect, *bsdf, scene, rc, depth, (uint8_t)(BSDF_TRANSMISSION | BSDF_SPECULAR));
        }
        return l;
    }
};

// src/core/sampling.rs:219-228
inline Vector3f uniform_sample_hemisphere(const Point2f& u) {
    Float z = u.x;
    Float r = std::sqrt(fmaxr(1.0f - z * z, 0.0f));
    Float phi = 2.0f * PI * u.y;
    Float sp, cp;
    det_sincos(phi, &sp, &cp);
    return Vector3f(r * cp, r * sp, z);
}
inline Float uniform_hemisphere_pdf() { return INV_2_PI; }

// integrators/ao.rs:55-104. n_samples hemisphere directions about the geometric normal at the first
// surface that has a BSDF; each contributes cos / (pdf * n_samples) when its ray escapes.
// D51 (intended): ao.rs:96 adds the term when scene.intersect_p(...) is TRUE, i.e. for the occluded
// directions — the complement of ambient occlusion (pbrt-v3 ao.cpp:79 tests !IntersectP). The
// unoccluded directions contribute here.
// The 2D array of ao.rs:77-81 (Sampler::get_2d_array) is drawn from the sample's stream in index order.
struct AOIntegrator : Integrator {
    bool cos_sample;
    int n_samples;
    AOIntegrator(bool cs, int ns) : cos_sample(cs), n_samples(ns) {}
    // ao.rs:36-37
    void request_samples(const Scene&, SamplerSpec& spec) override {
        n_samples = spec.round_count(n_samples);
        spec.arrays_2d.push_back(n_samples);
    }
    Spectrum li(Ray ray, const Scene& scene, RenderCtx& rc, int) const override {
        Spectrum l(0.0f);
        SurfaceInteraction isect;
        for (;;) {
            if (scene.intersect(ray, &isect, &rc.ctr)) {
                const MaterialDesc& mat = scene.materials[scene.material_of(isect)];
                std::shared_ptr<BSDF> bsdf = compute_scattering_functions(mat, isect, MODE_RADIANCE, true, scene.quirks);
                if (!bsdf) {
                    ray = isect.spawn_ray(ray.d);
                    continue;
                }
                Normal3f n = isect.n.face_forward(-ray.d);
                Vector3f s = isect.dpdu.normalize();
                Vector3f t = isect.n.cross(s);
                const Point2f* u_array = rc.sampler.get_2d_array(n_samples);  // ao.rs:77-81
                for (int i = 0; i < n_samples; ++i) {
                    Point2f u = u_array ? u_array[i] : rc.sampler.get_2d();
                    Vector3f wi;
                    Float pdf;
                    if (cos_sample) {
                        wi = cosine_sample_hemisphere(u, scene.quirks);
                        pdf = cosine_hemisphere_pdf(std::fabs(wi.z));
                    } else {
                        wi = uniform_sample_hemisphere(u);
                        pdf = uniform_hemisphere_pdf();
                    }
                    wi = Vector3f(s.x * wi.x + t.x * wi.y + n.x * wi.z, s.y * wi.x + t.y * wi.y + n.y * wi.z,
                                  s.z * wi.x + t.z * wi.y + n.z * wi.z);
                    Ray shadow = isect.spawn_ray(wi);
                    if (!scene.intersect_p(shadow, &rc.ctr)) l += Spectrum(wi.dot(n) / (pdf * (Float)n_samples));
                }
            }
            break;
        }
        return l;
    }
};

// ---------------------------------------------------------------------------------
// Camera — cameras/perspective.rs. The 4x4 matrices are supplied by the caller (the host side
// computes Transform::perspective / look_at, transform.rs:510-566); the oracle applies them.
// ---------------------------------------------------------------------------------
enum CameraKind { CAMERA_PERSPECTIVE = 0, CAMERA_ORTHOGRAPHIC = 1, CAMERA_ENVIRONMENT = 2 };
struct PerspectiveCamera {
    Matrix4 camera_to_world, raster_to_camera;
    Float lens_radius = 0.0f, focal_distance = 1e6f;
    Float shutter_open = 0.0f, shutter_close = 1.0f;
    int kind = CAMERA_PERSPECTIVE;
    int film_width = 1, film_height = 1;  // full_resolution (environment camera)
    // cameras/orthographic.rs:82-104. D58 (intended): with a lens the reference REPLACES the ray origin by the lens
    // point (orthographic.rs:97), which collapses the parallel projection onto the lens -> pbrt-v3 adds the lens
    // offset to the origin (orthographic.cpp:70-72).
    Float generate_ray_orthographic(const CameraSample& sample, Ray* ray) const {
        Point3f p_film(sample.p_film.x, sample.p_film.y, 0.0f);
        Point3f p_camera = xform_point(raster_to_camera, p_film);
        *ray = Ray(p_camera, Vector3f(0.0f, 0.0f, 1.0f), FLOAT_INF, 0.0f);
        if (lens_radius > 0.0f) {
            Point2f pl = concentric_sample_disk(sample.p_lens);
            Point2f p_lens(pl.x * lens_radius, pl.y * lens_radius);
            Float ft = focal_distance / ray->d.z;
            Point3f p_focus = ray->at(ft);
            ray->o.x += p_lens.x;
            ray->o.y += p_lens.y;
            ray->d = (p_focus - ray->o).normalize();
        }
        ray->time = (1.0f - sample.time) * shutter_open + sample.time * shutter_close;
        *ray = xform_ray(camera_to_world, *ray);
        return 1.0f;
    }
    // cameras/environment.rs:37-56
    Float generate_ray_environment(const CameraSample& sample, Ray* ray) const {
        Float theta = PI * sample.p_film.y / (Float)film_height;
        Float phi = 2.0f * PI * sample.p_film.x / (Float)film_width;
        Float st, ct, sp, cp;
        det_sincos(theta, &st, &ct);
        det_sincos(phi, &sp, &cp);
        *ray = Ray(Point3f(0.0f, 0.0f, 0.0f), Vector3f(st * cp, ct, st * sp), FLOAT_INF,
                   (1.0f - sample.time) * shutter_open + sample.time * shutter_close);
        *ray = xform_ray(camera_to_world, *ray);
        return 1.0f;
    }
    // perspective.rs:90-112 (generate_ray_differential :114-161 differs only in the differentials)
    Float generate_ray(const CameraSample& sample, Ray* ray) const {
        if (kind == CAMERA_ORTHOGRAPHIC) return generate_ray_orthographic(sample, ray);
        if (kind == CAMERA_ENVIRONMENT) return generate_ray_environment(sample, ray);
        Point3f p_film(sample.p_film.x, sample.p_film.y, 0.0f);
        Point3f p_camera = xform_point(raster_to_camera, p_film);
        *ray = Ray(Point3f(0.0f, 0.0f, 0.0f), p_camera.normalize(), FLOAT_INF, 0.0f);
        if (lens_radius > 0.0f) {
            Point2f pl = concentric_sample_disk(sample.p_lens);
            Point2f p_lens(pl.x * lens_radius, pl.y * lens_radius);
            Float ft = focal_distance / ray->d.z;
            Point3f p_focus = ray->at(ft);
            ray->o = Point3f(p_lens.x, p_lens.y, 0.0f);
            ray->d = (p_focus - ray->o).normalize();
        }
        // lerp(t, a, b) = (1 - t) * a + t * b  (pbrt.rs:222-226)
        ray->time = (1.0f - sample.time) * shutter_open + sample.time * shutter_close;
        *ray = xform_ray(camera_to_world, *ray);
        return 1.0f;
    }
};

// ---------------------------------------------------------------------------------
// Film — core/film.rs with the box filter (filters/boxf.rs; radius 0.5 x 0.5, D43).
// Output pixel = {xyz[3], filter_weight_sum} (film.rs:9-15, the 16 bytes SamplerIntegrator writes).
// ---------------------------------------------------------------------------------
struct Film {
    int width, height;
    Float filter_radius_x = 0.5f, filter_radius_y = 0.5f;
    Float max_sample_luminance = FLOAT_INF;
    static const int FILTER_TABLE_WIDTH = 16;
    Float filter_table[FILTER_TABLE_WIDTH * FILTER_TABLE_WIDTH];
    std::vector<Float> pixels;  // width*height*4: xyz + filter_weight_sum
    Film(int w, int h) : width(w), height(h), pixels((size_t)w * h * 4, 0.0f) {
        for (int i = 0; i < FILTER_TABLE_WIDTH * FILTER_TABLE_WIDTH; ++i) filter_table[i] = 1.0f;  // film.rs:52-63, boxf.rs:25-27
    }
    // film.rs:76-81 (D42: intended): floor(min + 0.5 - r), ceil(max - 0.5 + r) over the whole film
    void sample_bounds(int* x0, int* y0, int* x1, int* y1) const {
        *x0 = (int)std::floor(0.0f + 0.5f - filter_radius_x);
        *y0 = (int)std::floor(0.0f + 0.5f - filter_radius_y);
        *x1 = (int)std::ceil((Float)width - 0.5f + filter_radius_x);
        *y1 = (int)std::ceil((Float)height - 0.5f + filter_radius_y);
    }
};

// Reconstruction filters: src/filters/{boxf,gaussian,mitchell,sinc,triangle}.rs; Film::new tabulates
// filter.evaluate over the positive quadrant (film.rs:52-63).
enum FilterType { FILTER_BOX = 0, FILTER_GAUSSIAN = 1, FILTER_MITCHELL = 2, FILTER_LANCZOS = 3, FILTER_TRIANGLE = 4 };
inline Float filter_evaluate(int type, Float rx, Float ry, Float a, Float b, Float px, Float py) {
    switch (type) {
        case FILTER_GAUSSIAN: {  // gaussian.rs:17-39
            Float exp_x = std::exp(-a * rx * rx), exp_y = std::exp(-a * ry * ry);
            Float gx = fmaxr(std::exp(-a * px * px) - exp_x, 0.0f), gy = fmaxr(std::exp(-a * py * py) - exp_y, 0.0f);
            return gx * gy;
        }
        case FILTER_MITCHELL: {  // mitchell.rs:24-47
            auto m1d = [&](Float x) {
                x = std::fabs(2.0f * x);
                if (x > 1.0f)
                    return ((-a - 6.0f * b) * x * x * x + (6.0f * a + 30.0f * b) * x * x + (-12.0f * a - 48.0f * b) * x +
                            (8.0f * a + 24.0f * b)) * (1.0f / 6.0f);
                return ((12.0f - 9.0f * a - 6.0f * b) * x * x * x + (-18.0f + 12.0f * a + 6.0f * b) * x * x + (6.0f - 2.0f * a)) *
                       (1.0f / 6.0f);
            };
            return m1d(px * (1.0f / rx)) * m1d(py * (1.0f / ry));
        }
        case FILTER_LANCZOS: {  // sinc.rs:22-48
            auto sinc = [](Float x) {
                x = std::fabs(x);
                if (x < 1e-5f) return 1.0f;
                return std::sin(PI * x) / (PI * x);
            };
            auto wsinc = [&](Float x, Float radius) {
                x = std::fabs(x);
                if (x > radius) return 0.0f;
                return sinc(x) * sinc(x / a);
            };
            return wsinc(px, rx) * wsinc(py, ry);
        }
        case FILTER_TRIANGLE:  // triangle.rs:21-27
            return fmaxr(rx - std::fabs(px), 0.0f) * fmaxr(ry - std::fabs(py), 0.0f);
        default:
            return 1.0f;  // boxf.rs:25-27
    }
}
inline void filter_table(int type, Float rx, Float ry, Float a, Float b, Float* table) {
    const int n = Film::FILTER_TABLE_WIDTH;
    int offset = 0;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            Float px = ((Float)x + 0.5f) * rx / (Float)n, py = ((Float)y + 0.5f) * ry / (Float)n;
            table[offset++] = filter_evaluate(type, rx, ry, a, b, px, py);
        }
}
struct FilmTilePixel {
    Spectrum contrib_sum;
    Float filter_weight_sum = 0.0f;
};
// film.rs:231-295; pixel bounds [x0,x1) x [y0,y1)
struct FilmTile {
    int x0, y0, x1, y1;
    const Film& film;
    std::vector<FilmTilePixel> pixels;
    FilmTile(const Film& f, int x0_, int y0_, int x1_, int y1_)
        : x0(x0_), y0(y0_), x1(x1_), y1(y1_), film(f), pixels((size_t)std::max(0, (x1_ - x0_) * (y1_ - y0_))) {}
    // film.rs:252-295 (D42: p1 clamped with min; coordinates stay signed)
    void add_sample(const Point2f& p_film, Spectrum l, Float sample_weight) {
        if (l.y_value() > film.max_sample_luminance) l *= film.max_sample_luminance / l.y_value();
        Float dx = p_film.x - 0.5f, dy = p_film.y - 0.5f;
        int px0 = std::max((int)std::ceil(dx - film.filter_radius_x), x0);
        int py0 = std::max((int)std::ceil(dy - film.filter_radius_y), y0);
        int px1 = std::min((int)std::floor(dx + film.filter_radius_x) + 1, x1);
        int py1 = std::min((int)std::floor(dy + film.filter_radius_y) + 1, y1);
        const int ts = Film::FILTER_TABLE_WIDTH;
        for (int y = py0; y < py1; ++y) {
            Float fy = std::fabs(((Float)y - dy) * (1.0f / film.filter_radius_y) * (Float)ts);
            int ify = std::min(ts - 1, (int)std::floor(fy));
            for (int x = px0; x < px1; ++x) {
                Float fx = std::fabs(((Float)x - dx) * (1.0f / film.filter_radius_x) * (Float)ts);
                int ifx = std::min(ts - 1, (int)std::floor(fx));
                Float filter_weight = film.filter_table[ify * ts + ifx];
                FilmTilePixel& pixel = pixels[(size_t)(y - y0) * (x1 - x0) + (x - x0)];
                pixel.contrib_sum += l * sample_weight * filter_weight;
                pixel.filter_weight_sum += filter_weight;
            }
        }
    }
};

struct RenderParams {
    int spp = 1;
    uint64_t seed = 0;
    // rectangle of pixels to render [x0,x1) x [y0,y1) (pixel_bounds of SamplerIntegrator)
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    int n_threads = 1;
    SamplerSpec sampler;
};
struct RenderStats {
    TraversalCounters ctr;
    uint64_t camera_samples = 0;
    double seconds = 0.0;
};

// integrator.rs:399-480 — 16x16 tiles, one task per tile (parallel.rs:4-21 -> std::thread pool).
inline void render(const Scene& scene, const PerspectiveCamera& camera, Integrator& integrator, Film& film,
                   const RenderParams& rp_in, RenderStats* stats) {
    RenderParams rp = rp_in;
    integrator.pre_process(scene);
    rp.sampler.arrays_2d.clear();
    integrator.request_samples(scene, rp.sampler);
    rp.spp = (int)rp.sampler.samples_per_pixel(rp.spp);
    if (rp.sampler.kind == SAMPLER_HALTON) {  // HaltonSampler::new(spp, film.get_sample_bounds(), false)
        int sb[4];
        film.sample_bounds(&sb[0], &sb[1], &sb[2], &sb[3]);
        rp.sampler.halton.init(sb[2] - sb[0], sb[3] - sb[1]);
    }
    const int TILE_SIZE = 16;
    // get_sample_bounds with a 0.5 box filter = the pixel rectangle itself (film.rs:76-81, D42)
    int sx0 = rp.x0, sy0 = rp.y0, sx1 = rp.x1, sy1 = rp.y1;
    int ntx = (sx1 - sx0 + TILE_SIZE - 1) / TILE_SIZE, nty = (sy1 - sy0 + TILE_SIZE - 1) / TILE_SIZE;
    int fsb[4];
    film.sample_bounds(&fsb[0], &fsb[1], &fsb[2], &fsb[3]);
    std::atomic<int> next_tile(0);
    std::mutex film_mutex;
    int n_threads = std::max(1, rp.n_threads);
    std::vector<RenderStats> tstats(n_threads);
    auto worker = [&](int tid) {
        RenderCtx rc;
        rc.sampler.spec = &rp.sampler;
        for (;;) {
            int tile = next_tile.fetch_add(1);
            if (tile >= ntx * nty) break;
            int tx = tile % ntx, ty = tile / ntx;
            int x0 = sx0 + tx * TILE_SIZE, x1 = std::min(x0 + TILE_SIZE, sx1);
            int y0 = sy0 + ty * TILE_SIZE, y1 = std::min(y0 + TILE_SIZE, sy1);
            // get_film_tile (film.rs:93-109): tile pixel bounds = [ceil(min - 0.5 - r), floor(max - 0.5 + r) + 1)
            // clipped to the film. The reference omits pbrt-v3's "+ (1,1)" on the upper corner
            // (film.rs:100-102), which would drop the samples whose film position rounds up to the next
            // pixel at a tile's last column / row — intended (D50).
            int tx0 = (int)std::ceil((Float)x0 - 0.5f - film.filter_radius_x), ty0 = (int)std::ceil((Float)y0 - 0.5f - film.filter_radius_y);
            int tx1 = (int)std::floor((Float)x1 - 0.5f + film.filter_radius_x) + 1, ty1 = (int)std::floor((Float)y1 - 0.5f + film.filter_radius_y) + 1;
            FilmTile film_tile(film, std::max(tx0, 0), std::max(ty0, 0), std::min(tx1, film.width), std::min(ty1, film.height));
            for (int py = y0; py < y1; ++py)
                for (int px = x0; px < x1; ++px) {
                    // the pixel's number over the film's SAMPLE bounds (film.rs:76-81): with a filter wider than the 0.5
                    // box, pixels left of / above the film are sampled too and must not share a stream with a film pixel
                    // (y * width + x would give (width, y) and (0, y + 1) the same one)
                    const int64_t pixel_index = (int64_t)(py - fsb[1]) * (fsb[2] - fsb[0]) + (px - fsb[0]);
                    rc.sampler.start_pixel(rp.seed, pixel_index, rp.spp, px, py);
                    for (int s = 0; s < rp.spp; ++s) {
                        rc.sampler.start_sample(rp.seed, pixel_index, rp.spp, s);
                        CameraSample cs = rc.sampler.get_camera_sample(px, py);
                        Ray ray;
                        Float ray_weight = camera.generate_ray(cs, &ray);
                        Spectrum l(0.0f);
                        if (ray_weight > 0.0f) l = integrator.li(ray, scene, rc, 0);
                        if (l.has_nans() || l.y_value() < -1e-5f || std::isinf(l.y_value())) l = Spectrum(0.0f);  // D23
                        film_tile.add_sample(cs.p_film, l, ray_weight);
                        tstats[tid].camera_samples++;
                    }
                }
            // merge_film_tile (film.rs:111-123) under the film's write lock (camera.rs:95 RwLock)
            std::lock_guard<std::mutex> guard(film_mutex);
            for (int y = film_tile.y0; y < film_tile.y1; ++y)
                for (int x = film_tile.x0; x < film_tile.x1; ++x) {
                    const FilmTilePixel& tp = film_tile.pixels[(size_t)(y - film_tile.y0) * (film_tile.x1 - film_tile.x0) + (x - film_tile.x0)];
                    Float xyz[3];
                    tp.contrib_sum.to_xyz(xyz);
                    Float* mp = &film.pixels[((size_t)y * film.width + x) * 4];
                    for (int i = 0; i < 3; ++i) mp[i] += xyz[i];
                    mp[3] += tp.filter_weight_sum;
                }
        }
        tstats[tid].ctr = rc.ctr;
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < n_threads; ++t) threads.emplace_back(worker, t);
    worker(0);
    for (auto& t : threads) t.join();
    if (stats)
        for (auto& s : tstats) {
            stats->ctr.add(s.ctr);
            stats->camera_samples += s.camera_samples;
        }
}

}  // namespace oracle
