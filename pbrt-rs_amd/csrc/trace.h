// trace.h — BVH traversal + watertight triangle test for gfx950, one ray per lane.
//
// Computes exactly what BVHAccel::intersect / intersect_p (src/accelerators/bvh.rs:828-932) compute
// with Bounds3f::intersect_p (src/core/geometry.rs:709-751) and Triangle::intersect_test
// (src/shapes/triangle.rs:74-158): same visiting order (near child first by dir_is_neg[axis]), same
// box test, same strictness in every comparison, same shrinking of ray.t_max, so the closest
// hit, its barycentrics and tie-breaks are bit-identical.
//
// Device layout (built by scene.h from the reference-order LinearBVHNode array):
//   * one 64-byte record per INTERIOR node holding BOTH children's boxes, child references and
//     the split axis. One coalesced 64-B fetch replaces the reference's two dependent 32-B fetches
//     (node, then child); the far child's box test result is kept on the stack as its entry
//     distance and re-checked against the current t_max when popped, which is the same decision
//     the reference makes when it tests the box at pop time.
//   * child reference: >= 0 interior record index; < 0 leaf: ~ref = (first_slot << count_bits) | (n - 1).
//   * triangles in leaf order, 48 B each: {v0.xyz v1.x | v1.yz v2.xy | v2.z prim_id material light}.
//   * traversal stack: STACK_LDS entries per lane in LDS ([entry][lane], conflict-free),
//     deeper entries spill to a per-lane global slab.
// The traversal loop itself is in trace_persistent.h.
#pragma once
#include "dev_math.h"

namespace pb {

struct DevBVH {
    const float4* __restrict__ inodes;  // 4 x float4 per interior node
    const float4* __restrict__ tris;    // 3 x float4 per leaf slot
    float root_min[3], root_max[3];
    int root_ref;     // child-style reference of the root (leaf if the tree is a single leaf)
    int count_bits;   // bits of (n_primitives - 1) in a leaf reference
    int n_slots;
    uint2* __restrict__ spill;  // [entry][global lane] overflow stack
    int spill_stride;           // number of lanes the slab was sized for
    // two-level scenes (src/core/primitive.rs:105-159): the fields above describe the top level, whose leaf slots hold
    // TransformedPrimitives of object-space aggregates and, beside them, plain world-space triangles.
    int instanced;
    // 7 x float4 per top-level leaf slot: to_object rows 0-2, to_world rows 0-2, (material, instance id, object, kind);
    // kind 1 = a world-space triangle, its leaf slot in `tris` in the third field
    const float4* __restrict__ instances;
    // 2 x float4 per object aggregate: (root box min, root reference) (root box max, -)
    const float4* __restrict__ objects;
    int blas_count_bits;  // of the object-level leaf references (one width for all objects)
    // config 5's shape — one object aggregate, no world-space triangles — keeps the root in the kernel arguments and
    // skips the per-entry kind / object lookups (-11 % on config 5 when every entry paid for them)
    int general_top;
    int blas_root_ref;
    float blas_root_min[3], blas_root_max[3];
    // optional per-vertex shading data of TriangleMesh (triangle.rs:17-26: n, s, uv), 6 x float4 per leaf slot:
    // (n0.xyz, n1.x) (n1.yz, n2.xy) (n2.z, s0.xyz) (s1.xyz, s2.x) (s2.yz, uv0.xy) (uv1.xy, uv2.xy); null = none
    const float4* __restrict__ tri_shading;
    int has_normals, has_tangents, has_uvs;
    int has_spheres;  // some leaf slots hold spheres (kPrimSphere): the SPH kernels are launched
};

// waves per SIMD requested for the traversal kernels (occupancy sweep: DESIGN.md section 4)
#ifndef PB_TRACE_WAVES
#define PB_TRACE_WAVES 6
#endif
#ifndef PB_INST_WAVES
#define PB_INST_WAVES 5
#endif
#ifndef PB_STACK_LDS
#define PB_STACK_LDS 12
#endif
constexpr int kStackLds = PB_STACK_LDS;
constexpr int kStackSpill = 64 - PB_STACK_LDS;  // together: the reference's 64-entry stack (bvh.rs:839)
constexpr int kTraceBlock = 256;
// The ray queue can be cut into one contiguous segment per XCD (workgroups are dealt round-robin over the 8 XCDs,
// each with its own L2; a workgroup whose segment is empty helps the others): IO::segments(). Measured with 8
// segments: camera rays +9 % (neighbouring pixels walk the same part of the tree), bounce rays +-1 %, the two-level
// kernel -3 %: the render loop uses 8 segments for the camera-ray wavefront of single-level scenes and 1 otherwise.
constexpr int kQueueSegments = 8;  // counters allocated per context

struct TravRay {
    float ox, oy, oz, dx, dy, dz, tmax;
};
struct TravHit {
    float t, b0, b1, b2;
    int slot;  // leaf slot of the hit triangle, -1 = miss
};

// tri.z.w of the third float4 carries flags in the top bits of `light`: see scene.h
constexpr int kTriDegenerate = 1 << 30;  // Triangle::intersect returns false (triangle.rs:212-216)
constexpr int kPrimSphere = 1 << 29;     // the slot holds a Sphere: (centre.xyz, radius) in the first float4 (shapes/sphere.rs)
constexpr int kPrimLightMask = 0x1fffffff;  // light index + 1

PB_DEV bool slab_test(float bx0, float bx1, float by0, float by1, float bz0, float bz1, const TravRay& r, float idx,
                      float idy, float idz, float tmax_ray, float* entry) {
    // bx0 = bounds[dir_is_neg[0]].x, bx1 = bounds[1 - dir_is_neg[0]].x, ...
    // Straight-line form of geometry.rs:709-751: the early returns of the reference become flags that are and-ed at
    // the end (same comparisons, same NaN behaviour); a wave almost always holds a lane that passes the x / y test,
    // so skipping the z slab per lane only costs exec-mask bookkeeping.
    float t_min = (bx0 - r.ox) * idx;
    float t_max = (bx1 - r.ox) * idx;
    float ty_min = (by0 - r.oy) * idy;
    float ty_max = (by1 - r.oy) * idy;
    float tz_min = (bz0 - r.oz) * idz;
    float tz_max = (bz1 - r.oz) * idz;
    t_max *= kSlabScale;
    ty_max *= kSlabScale;
    tz_max *= kSlabScale;
    int fail_xy = (int)(t_min > ty_max) | (int)(ty_min > t_max);
    t_min = (ty_min > t_min) ? ty_min : t_min;
    t_max = (ty_max < t_max) ? ty_max : t_max;
    int fail_z = (int)(t_min > tz_max) | (int)(tz_min > t_max);
    t_min = (tz_min > t_min) ? tz_min : t_min;
    t_max = (tz_max < t_max) ? tz_max : t_max;
    *entry = t_min;
    return ((fail_xy | fail_z) == 0) & (t_min < tmax_ray) & (t_max > 0.0f);
}
// The same test, and beside it what it would say with ray.t_max = +inf (`keep`: the ray meets the box's slabs at all) — the
// part of the decision that does not depend on t_max, which can move (trace_persistent.h): one evaluation, two answers.
PB_DEV bool slab_test_keep(float bx0, float bx1, float by0, float by1, float bz0, float bz1, const TravRay& r, float idx,
                           float idy, float idz, float tmax_ray, float* entry, bool* keep) {
    float e;
    const bool any_t = slab_test(bx0, bx1, by0, by1, bz0, bz1, r, idx, idy, idz, kInf, &e);
    *entry = e;
    *keep = any_t;
    return any_t & (e < tmax_ray);
}

// Per-ray constants of the watertight test (triangle.rs:84-101)
struct TriRayConst {
    int kz;
    float sx, sy, sz;
};
PB_DEV TriRayConst tri_ray_setup(const TravRay& r) {
    float ax = __builtin_fabsf(r.dx), ay = __builtin_fabsf(r.dy), az = __builtin_fabsf(r.dz);
    int kz = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);
    // permute(kx, ky, kz): kz=0 -> (y,z,x); kz=1 -> (z,x,y); kz=2 -> (x,y,z)
    float px = kz == 0 ? r.dy : (kz == 1 ? r.dz : r.dx);
    float py = kz == 0 ? r.dz : (kz == 1 ? r.dx : r.dy);
    float pz = kz == 0 ? r.dx : (kz == 1 ? r.dy : r.dz);
    TriRayConst c;
    c.kz = kz;
    c.sx = -px / pz;
    c.sy = -py / pz;
    c.sz = 1.0f / pz;
    return c;
}
// the same with the ray's reciprocal direction at hand: 1 / d[kz] IS the traversal's inv_dir component (one correctly
// rounded division of the same operands), two divisions instead of three
PB_DEV TriRayConst tri_ray_setup(const TravRay& r, float idx, float idy, float idz) {
    float ax = __builtin_fabsf(r.dx), ay = __builtin_fabsf(r.dy), az = __builtin_fabsf(r.dz);
    int kz = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);
    float px = kz == 0 ? r.dy : (kz == 1 ? r.dz : r.dx);
    float py = kz == 0 ? r.dz : (kz == 1 ? r.dx : r.dy);
    float pz = kz == 0 ? r.dx : (kz == 1 ? r.dy : r.dz);
    TriRayConst c;
    c.kz = kz;
    c.sx = -px / pz;
    c.sy = -py / pz;
    c.sz = kz == 0 ? idx : (kz == 1 ? idy : idz);
    return c;
}
PB_DEV V3 permute_kz(V3 v, int kz) {
    return V3{kz == 0 ? v.y : (kz == 1 ? v.z : v.x), kz == 0 ? v.z : (kz == 1 ? v.x : v.y),
              kz == 0 ? v.x : (kz == 1 ? v.y : v.z)};
}

// triangle.rs:74-158. Returns true and (b0,b1,b2,t) when the ray hits within (0, tmax].
PB_DEV bool triangle_test(V3 p0, V3 p1, V3 p2, const TravRay& r, const TriRayConst& c, float tmax, float* b0o,
                          float* b1o, float* b2o, float* to) {
    V3 o = V3{r.ox, r.oy, r.oz};
    V3 p0t = permute_kz(p0 - o, c.kz);
    V3 p1t = permute_kz(p1 - o, c.kz);
    V3 p2t = permute_kz(p2 - o, c.kz);
    p0t.x += c.sx * p0t.z;
    p0t.y += c.sy * p0t.z;
    p1t.x += c.sx * p1t.z;
    p1t.y += c.sy * p1t.z;
    p2t.x += c.sx * p2t.z;
    p2t.y += c.sy * p2t.z;
    // edge functions in f64 (triangle.rs:109-111, D12)
    float e0 = (float)((double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x);
    float e1 = (float)((double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x);
    float e2 = (float)((double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x);
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    p0t.z *= c.sz;
    p1t.z *= c.sz;
    p2t.z *= c.sz;
    float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < tmax * det)) return false;
    if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > tmax * det)) return false;
    float inv_det = 1.0f / det;
    float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
    float t = t_scaled * inv_det;
    float max_zt = max3(__builtin_fabsf(p0t.z), __builtin_fabsf(p1t.z), __builtin_fabsf(p2t.z));
    float delta_z = kGamma3 * max_zt;
    float max_xt = max3(__builtin_fabsf(p0t.x), __builtin_fabsf(p1t.x), __builtin_fabsf(p2t.x));
    float max_yt = max3(__builtin_fabsf(p0t.y), __builtin_fabsf(p1t.y), __builtin_fabsf(p2t.y));
    float delta_x = kGamma5 * (max_xt + max_zt);
    float delta_y = kGamma5 * (max_yt + max_zt);
    float delta_e = 2.0f * (kGamma2 * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = max3(__builtin_fabsf(e0), __builtin_fabsf(e1), __builtin_fabsf(e2));
    float delta_t = 3.0f * (kGamma3 * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * __builtin_fabsf(inv_det);
    if (t <= delta_t) return false;
    *b0o = b0;
    *b1o = b1;
    *b2o = b2;
    *to = t;
    return true;
}

PB_DEV void load_tri(const float4* __restrict__ tris, int slot, V3* p0, V3* p1, V3* p2, int* flags) {
    float4 a = tris[3 * slot], b = tris[3 * slot + 1], c = tris[3 * slot + 2];
    *p0 = V3{a.x, a.y, a.z};
    *p1 = V3{a.w, b.x, b.y};
    *p2 = V3{b.z, b.w, c.x};
    *flags = __float_as_int(c.w);
}

// ---- Sphere (src/shapes/sphere.rs) with EFloat (src/core/efloat.rs; D7 intended), full spheres placed by a
// translation: object_to_world = translate(centre). Same operation order as the oracle (o_sphere.h). ----
struct EFloat {
    float v, low, high;
};
PB_DEV EFloat ef_make(float v, float err) {  // efloat.rs:15-25
    EFloat e;
    e.v = v;
    if (err == 0.0f) {
        e.low = e.high = v;
    } else {
        e.low = next_float_down(v - err);
        e.high = next_float_up(v + err);
    }
    return e;
}
PB_DEV EFloat ef_add(EFloat a, EFloat b) {
    return EFloat{a.v + b.v, next_float_down(a.low + b.low), next_float_up(a.high + b.high)};
}
PB_DEV EFloat ef_sub(EFloat a, EFloat b) {
    return EFloat{a.v - b.v, next_float_down(a.low - b.high), next_float_up(a.high - b.low)};
}
PB_DEV float ef_min4(float a, float b, float c, float d) {
    float ab = a < b ? a : b, cd = c < d ? c : d;  // fminr(fminr(a, b), fminr(c, d))
    return ab < cd ? ab : cd;
}
PB_DEV float ef_max4(float a, float b, float c, float d) {
    float ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
PB_DEV EFloat ef_mul(EFloat a, EFloat b) {
    float p0 = a.low * b.low, p1 = a.high * b.low, p2 = a.low * b.high, p3 = a.high * b.high;
    return EFloat{a.v * b.v, next_float_down(ef_min4(p0, p1, p2, p3)), next_float_up(ef_max4(p0, p1, p2, p3))};
}
PB_DEV EFloat ef_div(EFloat a, EFloat b) {
    EFloat o;
    o.v = a.v / b.v;
    if (b.low < 0.0f && b.high > 0.0f) {
        o.low = -kInf;
        o.high = kInf;
    } else {
        float d0 = a.low / b.low, d1 = a.high / b.low, d2 = a.low / b.high, d3 = a.high / b.high;
        o.low = next_float_down(ef_min4(d0, d1, d2, d3));
        o.high = next_float_up(ef_max4(d0, d1, d2, d3));
    }
    return o;
}
PB_DEV EFloat ef_mulf(EFloat a, float f) { return ef_mul(a, ef_make(f, 0.0f)); }

// The object-space ray of Sphere::intersect_test (geometry.rs:1077-1137 through translate(-centre)) with its
// origin / direction error bounds.
struct SphereRay {
    float ox, oy, oz, dx, dy, dz;
    float oex, oey, oez, dex, dey, dez;
};
PB_DEV SphereRay sphere_object_ray(float cx, float cy, float cz, const TravRay& r) {
    SphereRay q;
    float x = r.ox, y = r.oy, z = r.oz;
    // rows of translate(-c): (1 0 0 -cx) (0 1 0 -cy) (0 0 1 -cz); every product of the general formula is kept
    q.ox = 1.0f * x + 0.0f * y + 0.0f * z + (-cx);
    q.oy = 0.0f * x + 1.0f * y + 0.0f * z + (-cy);
    q.oz = 0.0f * x + 0.0f * y + 1.0f * z + (-cz);
    q.oex = (__builtin_fabsf(1.0f * x) + __builtin_fabsf(0.0f * y) + __builtin_fabsf(0.0f * z) + __builtin_fabsf(-cx)) * kGamma3;
    q.oey = (__builtin_fabsf(0.0f * x) + __builtin_fabsf(1.0f * y) + __builtin_fabsf(0.0f * z) + __builtin_fabsf(-cy)) * kGamma3;
    q.oez = (__builtin_fabsf(0.0f * x) + __builtin_fabsf(0.0f * y) + __builtin_fabsf(1.0f * z) + __builtin_fabsf(-cz)) * kGamma3;
    float dx = r.dx, dy = r.dy, dz = r.dz;
    q.dex = (__builtin_fabsf(1.0f * dx) + __builtin_fabsf(0.0f * dy) + __builtin_fabsf(0.0f * dz)) * kGamma3;
    q.dey = (__builtin_fabsf(0.0f * dx) + __builtin_fabsf(1.0f * dy) + __builtin_fabsf(0.0f * dz)) * kGamma3;
    q.dez = (__builtin_fabsf(0.0f * dx) + __builtin_fabsf(0.0f * dy) + __builtin_fabsf(1.0f * dz)) * kGamma3;
    q.dx = 1.0f * dx + 0.0f * dy + 0.0f * dz;
    q.dy = 0.0f * dx + 1.0f * dy + 0.0f * dz;
    q.dz = 0.0f * dx + 0.0f * dy + 1.0f * dz;
    float l2 = q.dx * q.dx + q.dy * q.dy + q.dz * q.dz;
    if (l2 > 0.0f) {
        float dt = (__builtin_fabsf(q.dx) * q.oex + __builtin_fabsf(q.dy) * q.oey + __builtin_fabsf(q.dz) * q.oez) / l2;
        q.ox += q.dx * dt;
        q.oy += q.dy * dt;
        q.oz += q.dz * dt;
    }
    return q;
}
// Sphere::intersect_test (sphere.rs:228-284) for a full sphere; t_max is the world ray's. Outputs t (EFloat value)
// and the refined object-space hit point + its phi.
PB_DEV bool sphere_test(float cx, float cy, float cz, float radius, const TravRay& r, float tmax, float* t_out, V3* p_hit,
                        float* phi_out) {
    SphereRay q = sphere_object_ray(cx, cy, cz, r);
    EFloat ox = ef_make(q.ox, q.oex), oy = ef_make(q.oy, q.oey), oz = ef_make(q.oz, q.oez);
    EFloat dx = ef_make(q.dx, q.dex), dy = ef_make(q.dy, q.dey), dz = ef_make(q.dz, q.dez);
    EFloat a = ef_add(ef_add(ef_mul(dx, dx), ef_mul(dy, dy)), ef_mul(dz, dz));
    EFloat b = ef_mulf(ef_add(ef_add(ef_mul(dx, ox), ef_mul(dy, oy)), ef_mul(dz, oz)), 2.0f);
    EFloat rr = ef_make(radius, 0.0f);
    EFloat c = ef_sub(ef_add(ef_add(ef_mul(ox, ox), ef_mul(oy, oy)), ef_mul(oz, oz)), ef_mul(rr, rr));
    // EFloat::quadratic (efloat.rs:64-87)
    double discrim = (double)b.v * (double)b.v - 4.0 * (double)a.v * (double)c.v;
    if (discrim < 0.0) return false;
    double root = __builtin_sqrt(discrim);
    EFloat froot = ef_make((float)root, kMachineEpsilon * (float)root);
    EFloat qq = (b.v < 0.0f) ? ef_mulf(ef_sub(b, froot), -0.5f) : ef_mulf(ef_add(b, froot), -0.5f);
    EFloat t0 = ef_div(qq, a), t1 = ef_div(c, qq);
    if (t0.v > t1.v) {
        EFloat tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    const float phi_max = 360.0f * (kPi / 180.0f);
    for (int k = 0; k < 2; ++k) {  // the hit-selection loop of sphere.rs:259-281
        EFloat t = k == 0 ? t0 : t1;
        if (t.low < 0.0f || t.high > tmax) continue;
        V3 ph = V3{q.ox + q.dx * t.v, q.oy + q.dy * t.v, q.oz + q.dz * t.v};
        float scale = radius / __builtin_sqrtf(ph.x * ph.x + ph.y * ph.y + ph.z * ph.z);
        ph = V3{ph.x * scale, ph.y * scale, ph.z * scale};
        if (ph.x == 0.0f && ph.y == 0.0f) ph.x = 1e-5f * radius;
        float phi = det_atan2(ph.y, ph.x);
        if (phi < 0.0f) phi += 2.0f * kPi;
        if (phi > phi_max) continue;
        *t_out = t.v;
        *p_hit = ph;
        *phi_out = phi;
        return true;
    }
    return false;
}

// wave-reduce the instrumented counts (all 64 lanes active), one atomic set per wave
PB_DEV void count_flush(unsigned long long* counters, uint32_t n_node, uint32_t n_prim, uint32_t n_rays, uint32_t n_inst) {
    for (int o = 32; o > 0; o >>= 1) {
        n_node += __shfl_xor(n_node, o, 64);
        n_prim += __shfl_xor(n_prim, o, 64);
        n_rays += __shfl_xor(n_rays, o, 64);
        n_inst += __shfl_xor(n_inst, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counters[0], (unsigned long long)n_node);
        atomicAdd(&counters[1], (unsigned long long)n_prim);
        atomicAdd(&counters[2], (unsigned long long)n_rays);
        atomicAdd(&counters[3], (unsigned long long)n_inst);
    }
}

}  // namespace pb
