// wide_bvh.h — 4-wide, 48-byte quantised node records laid over the reference's binary BVH (layout + the box filter).
//
// Why: k_trace is bound by the number of 16-B lane requests the texture-address / L1 path has to process
// (profiles/r01_pmc_summary.txt: TA busy 91 %), four per traversal step of the 64-B child-pair records. A record that
// holds FOUR children (two levels of the reference's tree) in 48 bytes costs three requests per step and halves the
// steps. The boxes in it are 8-bit, conservative: they only FILTER. What the reference decides is decided exactly:
//
//   BVHAccel::intersect (src/accelerators/bvh.rs:828-879) visits a leaf iff the leaf's own box passes
//   Bounds3f::intersect_p (src/core/geometry.rs:709-751) with the ray.t_max current at that moment: every ancestor
//   box contains the leaf box and was tested earlier (t_max only shrinks), and the slab test is monotone in the box
//   and in t_max (no 0 * inf: rays with a non-finite reciprocal direction never take this path), so the ancestors'
//   tests add nothing. The reference is therefore "for every leaf in near-first depth-first order: if the leaf box
//   passes with the current t_max, test its triangles". The wide traversal enumerates a SUPERSET of those leaves in
//   the same order (children of a record are ranked by the two levels of dir_is_neg[axis] decisions, bvh.rs:857-865),
//   applies the exact slab test to the exact leaf box, then the same Triangle::intersect_test: same sequence of
//   triangle tests with the same t_max values, hence the same (prim, t, b0, b1, b2), ties included.
//
// Record (12 dwords), slots 0,1 = children of the binary node's first child, 2,3 = of its second child (a binary
// child that is a leaf sits in the even slot, the odd one is empty):
//   dw0..2  base.xyz as floats: a point below the node box's lower corner whose low mantissa byte IS m[0..2] (the
//           builder picks the upper 24 bits so that the float, byte included, stays below the corner: no masking)
//   dw3     ex | ey << 6 | ez << 12 (6-bit signed, cell = 2^e) | axis_root << 18 | axis_c0 << 20 | axis_c1 << 22 | m[3] << 24
//   dw4..9  lo.x, hi.x, lo.y, hi.y, lo.z, hi.z: byte s = slot s, plane = base + q * cell, at least kWideSlack cells
//           outside the child's float box (lo rounded down, hi up)
//   dw10    index of the first interior child (the interior children of a record are contiguous)
//   dw11    ~(first triangle of the first leaf child << 2) (the leaf children's triangles are contiguous in `wtris`)
//   m[s]    0xFF empty | 0x80 + k: interior child k | (triangle offset << 2) | (n - 1): leaf of n <= 4 triangles
// Child reference (stack entry / `cur`): >= 0 record index; < 0 leaf, ~ref = first_wide_triangle << 2 | (n - 1).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PB_HD __host__ __device__ __forceinline__
#else
#define PB_HD inline
#endif

namespace pb {

constexpr int kWideNodeDwords = 12;
// How the records and the wide-order triangles lie in HBM (WideTrees::vec_stride, chosen per scene). Builders, host arrays and
// pbrt_hip_debug_wide_export speak the packed 48-byte form. A scene whose records + triangles exceed kWideLineAlignBytes spreads
// both arrays to one 64-byte line per record / triangle when it takes them over (pbrt_hip.hip): the same three 16-B loads per
// lane, but no record straddles two lines any more (at a 48-B stride every second one does) — dependent gathers from a table
// beyond L2 run 11-14 % faster that way (tools/micro/probe_stride.hip), the config-3 frame +1.0 %; a tree that fits L2 keeps the
// packed form (padding only costs it cache: config 5's 258 KB of records -0.6 %). profiles/r05_line_aligned.txt.
constexpr size_t kWideLineAlignBytes = 8u << 20;
constexpr int kExpMin = -32, kExpMax = 13;  // cell = 2^e, stored as a 6-bit signed field
constexpr double kWideSlack = 1.0 / 256.0;   // every quantised plane lies at least this many cells outside the float box
constexpr float kWideCoordLimit = 1048576.0f;  // |scene coordinate| <= 2^20, else the scene keeps the binary records
// rays outside these ranges (a zero or denormal direction component, an origin far outside any scene this path
// accepts) are traced over the binary records by the exact kernel: the bound below assumes no overflow / 0 * inf
constexpr float kWideInvDirMin = 9.094947017729282e-13f;  // 2^-40
constexpr float kWideInvDirMax = 1099511627776.0f;        // 2^40
constexpr float kWideOriginLimit = 16777216.0f;           // 2^24

// The filter's arithmetic, shared by the kernel and the host-side conservativeness test (tests/native/).
// For a covered ray and one axis:   A = (base - o) * inv_d  (two roundings),  S = inv_d * 2^e  (exact),
// plane value  fma(q, S, fma(-/+k, |A|, A)),  k = 2^-19 = 32 u.  For every float box inside the quantised one
// near_q <= near_exact  and  far_q >= far_exact * (1 + 2 gamma_3): the roundings of A, of the two fmas and of the
// exact formula's own two (three) add up to < 14 u (|A| + 256 |S|) per plane; k |A| covers the |A| part and the
// kWideSlack cells the builder leaves between the quantised and the float plane (2^-8 |S|) cover the |S| part
// (DESIGN.md section 4.1).
struct WideSetup {
    float Sx, Sy, Sz;
    float Anx, Afx, Any, Afy, Anz, Afz;
};
PB_HD float wide_abs(float x) { return __builtin_fabsf(x); }
PB_HD float wide_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
PB_HD float wide_fmin(float a, float b) { return __builtin_fminf(a, b); }
PB_HD float wide_as_float(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
// per record and ray: dw0..dw3 of the record, ray origin and reciprocal direction
PB_HD WideSetup wide_setup(uint32_t dw0, uint32_t dw1, uint32_t dw2, uint32_t dw3, float ox, float oy, float oz, float idx,
                           float idy, float idz) {
    WideSetup w;
    w.Sx = __builtin_ldexpf(idx, ((int)(dw3 << 26)) >> 26);
    w.Sy = __builtin_ldexpf(idy, ((int)(dw3 << 20)) >> 26);
    w.Sz = __builtin_ldexpf(idz, ((int)(dw3 << 14)) >> 26);
    float Ax = (wide_as_float(dw0) - ox) * idx;
    float Ay = (wide_as_float(dw1) - oy) * idy;
    float Az = (wide_as_float(dw2) - oz) * idz;
    // per axis: the error of an axis' plane values scales with that axis' magnitudes only (a ray nearly parallel to
    // a slab has huge values there and must not blur the other two)
    const float k = 1.9073486328125e-06f;  // 2^-19 = 32 u
    w.Anx = __builtin_fmaf(-k, wide_abs(Ax), Ax);
    w.Afx = __builtin_fmaf(k, wide_abs(Ax), Ax);
    w.Any = __builtin_fmaf(-k, wide_abs(Ay), Ay);
    w.Afy = __builtin_fmaf(k, wide_abs(Ay), Ay);
    w.Anz = __builtin_fmaf(-k, wide_abs(Az), Az);
    w.Afz = __builtin_fmaf(k, wide_abs(Az), Az);
    return w;
}
// one child: nq* / fq* = the near / far plane dwords of the three axes (chosen by the sign of the direction), slot s.
// Returns "the ray may hit a float box inside this quantised box before t_max"; *entry <= the exact test's entry distance.
PB_HD bool wide_child_test(const WideSetup& w, uint32_t nqx, uint32_t nqy, uint32_t nqz, uint32_t fqx, uint32_t fqy, uint32_t fqz,
                           int s, float tmax, float* entry) {
    float tnx = __builtin_fmaf((float)((nqx >> (8 * s)) & 0xffu), w.Sx, w.Anx);
    float tny = __builtin_fmaf((float)((nqy >> (8 * s)) & 0xffu), w.Sy, w.Any);
    float tnz = __builtin_fmaf((float)((nqz >> (8 * s)) & 0xffu), w.Sz, w.Anz);
    float tfx = __builtin_fmaf((float)((fqx >> (8 * s)) & 0xffu), w.Sx, w.Afx);
    float tfy = __builtin_fmaf((float)((fqy >> (8 * s)) & 0xffu), w.Sy, w.Afy);
    float tfz = __builtin_fmaf((float)((fqz >> (8 * s)) & 0xffu), w.Sz, w.Afz);
    float t0 = wide_fmax(wide_fmax(tnx, tny), tnz);
    float t1 = wide_fmin(wide_fmin(tfx, tfy), wide_fmin(tfz, tmax));
    *entry = t0;
    return wide_fmax(t0, 0.0f) <= t1;  // (t0 <= far) & (t0 <= t_max) & (far >= 0) & (t_max >= 0)
}
// rays the bound does not cover (traced over the binary records instead)
PB_HD bool wide_ray_covered(float ox, float oy, float oz, float idx, float idy, float idz) {
    float ax = wide_abs(idx), ay = wide_abs(idy), az = wide_abs(idz);
    return ax >= kWideInvDirMin && ax <= kWideInvDirMax && ay >= kWideInvDirMin && ay <= kWideInvDirMax && az >= kWideInvDirMin &&
           az <= kWideInvDirMax && wide_abs(ox) <= kWideOriginLimit && wide_abs(oy) <= kWideOriginLimit &&
           wide_abs(oz) <= kWideOriginLimit;
}

#if defined(__HIPCC__)
// device arrays of one scene's wide records (built by host_wide.cpp, uploaded by pbrt_hip.hip)
struct WideTrees {
    const uint4* __restrict__ nodes;        // 3 x uint4 per record, at a stride of vec_stride
    const float4* __restrict__ tris;        // 3 x float4 per wide-order triangle, at a stride of vec_stride: (v0.xyz v1.x) (v1.yz v2.xy) (v2.z slot flags -)
    const float4* __restrict__ leaf_boxes;  // 2 x float4 per wide-order triangle position (leaves of n >= 2)
    int root_ref;
    int vec_stride;  // uint4 / float4 per record and per wide-order triangle: 3 packed, 4 one 64-byte line each
    uint2* __restrict__ spill;  // [entry][global lane]
    int spill_stride;
    uint32_t* __restrict__ special_list;  // tokens (IO::token) of the rays left to the binary kernel
    unsigned int* __restrict__ special_count;
    // two-level scenes: `nodes` holds the top-level tree's records, then every object aggregate's; the leaves of the
    // top-level tree hold top-level primitives in wide order:
    // 5 x float4 per position: (exact box of the entry's leaf: min, binary-layout top slot) (max, object index or
    // 0x40000000 | leaf slot of a world-space triangle) and the three world-to-object rows
    const float4* __restrict__ top_slots;
    const float4* __restrict__ top_boxes;  // 2 x float4 per position: the exact box of the top-level leaf the position belongs to (debug export)
    const float4* __restrict__ objects;    // 2 x float4 per object aggregate: (root box min, wide root reference) (root box max, -)
    const float4* __restrict__ slot_tris;  // DevBVH::tris (the world-space triangles of top-level leaves are read by leaf slot)
    // instances of ONE object aggregate with nothing beside them (config 5's shape, the kernels' INST == 1): that object's
    // root box and wide root reference ride in the kernel arguments
    int general_top;
    int obj0_root;
    float obj0_min[3], obj0_max[3];
};
#endif

}  // namespace pb
