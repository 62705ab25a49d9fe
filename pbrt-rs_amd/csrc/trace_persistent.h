// trace_persistent.h — persistent-threads traversal with per-lane ray replacement.
//
// Same arithmetic and visiting order as BVHAccel::intersect / intersect_p
// (src/accelerators/bvh.rs:828-932) with Bounds3f::intersect_p (src/core/geometry.rs:709-751) and
// Triangle::intersect_test (src/shapes/triangle.rs:74-158); with INST also TransformedPrimitive
// (src/core/primitive.rs:136-159): a top-level BVHAccel whose leaves hold instances of one
// object-space BVHAccel. What is MI355X-specific is how rays are scheduled onto the 64 lanes of a
// wavefront. Ray lengths in an incoherent batch are heavy-tailed, so a wave that traces 64 rays
// to completion idles most lanes (measured: ~9 % VALU lane utilisation). Here
//   * a wave owns a contiguous chunk of the ray queue (one atomicAdd per kChunk rays),
//   * a lane that finishes its ray takes the next ray of the chunk (refill is batched: it runs
//     when >= kRefillThresh lanes are idle, one ballot/mbcnt prefix, no atomics),
//   * interior-node steps run in a tight loop while enough lanes are at interior nodes; lanes
//     that reached a leaf wait and the leaf (<= max_prims triangles) is processed for all of them
//     together (while-while with postponed leaves).
// Traversal state lives in registers, the stack in LDS ([entry][lane]) with a global spill slab.
//
// SPH = true: leaf slots may hold spheres (kPrimSphere; BASELINE config 1).
// COUNT = true is the instrumented variant: it also counts the box tests (bvh.rs:841-842),
// triangle tests (triangle.rs:74) and instance entries (primitive.rs:136) that the REFERENCE's
// loops perform for the same rays. The reference pushes the far child untested and tests it when
// popped, so a far child whose box already failed is still pushed (entry distance +inf) and
// counted when popped; entries never popped (any-hit early exit) are not counted.
#pragma once
#include "trace.h"

namespace pb {

#ifndef PB_CHUNK
#define PB_CHUNK 128
#endif
#ifndef PB_REFILL_THRESH
#define PB_REFILL_THRESH 8
#endif
#ifndef PB_INTERIOR_THRESH
#define PB_INTERIOR_THRESH 32
#endif
constexpr int kChunk = PB_CHUNK;                    // rays per wave-level queue grab
constexpr int kRefillThresh = PB_REFILL_THRESH;     // refill when at least this many lanes are idle
constexpr int kInteriorThresh = PB_INTERIOR_THRESH; // keep stepping interior nodes while at least this many lanes do

#ifdef PB_LANE_STATS
// development instrumentation (tools/lane_stats.py): wave-level iteration counts and active-lane sums
__device__ unsigned long long g_lane_stats[8];
#define PB_STAT(i, v) stat[i] += (v)
#else
#define PB_STAT(i, v)
#endif

struct LaneState {
    TravRay r;  // the ray being traversed (object-space inside an instance)
    float idx, idy, idz;
    TriRayConst trc;
    float tmax;
    float b0, b1, b2;
    int hit_slot;
    int cur, sp;
    uint32_t index;  // token of this lane's ray (IO::token: the queue entry / batch position load and store name it by)
    bool nx, ny, nz, any, has_work;
};
// two-level state (INST)
struct InstState {
    float wox, woy, woz, wdx, wdy, wdz;  // world-space ray
    float tmax_world;
    int leaf_first, leaf_cnt, leaf_next;  // instance leaf being processed
    int cur_inst, hit_inst;               // instance slots
    int base_sp;                          // stack height when the instance was entered
    bool in_instance, hit_here;
};

PB_DEV void stack_push(const DevBVH& bvh, uint2* lds_stack, int spill_lane, int& sp, int node, float entry) {
    uint2 ent = make_uint2((uint32_t)node, __float_as_uint(entry));
    if (sp < kStackLds)
        lds_stack[sp * kTraceBlock] = ent;
    else
        bvh.spill[(size_t)(sp - kStackLds) * bvh.spill_stride + spill_lane] = ent;
    ++sp;
}
PB_DEV uint2 stack_read(const DevBVH& bvh, const uint2* lds_stack, int spill_lane, int sp) {
    return (sp < kStackLds) ? lds_stack[sp * kTraceBlock]
                            : bvh.spill[(size_t)(sp - kStackLds) * bvh.spill_stride + spill_lane];
}

PB_DEV void ray_constants(LaneState& s) {
    s.idx = 1.0f / s.r.dx;  // bvh.rs:831
    s.idy = 1.0f / s.r.dy;
    s.idz = 1.0f / s.r.dz;
    s.nx = s.idx < 0.0f;    // bvh.rs:832-836
    s.ny = s.idy < 0.0f;
    s.nz = s.idz < 0.0f;
    s.trc = tri_ray_setup(s.r);
}
PB_DEV bool root_box_test(const LaneState& s, const float* mn, const float* mx) {
    float e;
    return slab_test(s.nx ? mx[0] : mn[0], s.nx ? mn[0] : mx[0], s.ny ? mx[1] : mn[1], s.ny ? mn[1] : mx[1],
                     s.nz ? mx[2] : mn[2], s.nz ? mn[2] : mx[2], s.r, s.idx, s.idy, s.idz, s.tmax, &e);
}

// IO policy: n(), segments(), token(i) of queue position i, load(token, &ray, &any) -> bool real ray,
// strict(token): an any-hit ray that asks for the boolean of Primitive::intersect instead of intersect_p's — a hit on a
// triangle that Triangle::intersect rejects (kTriDegenerate, triangle.rs:197-216) does not count (wf_state.h: RS_MIS_BOOL),
// store(token, any, found, t, b0, b1, b2, slot, instance)
// INST: 0 = one level; 1 = instances of ONE object aggregate and nothing beside them (config 5's shape: the object's root
// rides in the kernel arguments); 2 = the general top level (several objects, world-space triangles beside the instances).
// A template value rather than a run-time flag: with the general code compiled in, the 96-register build of the
// single-object kernel spilled 100 B instead of 56 B and config 5 lost 15 %.
template <class IO, bool COUNT, int INST, bool SPH = false>
PB_DEV void trace_persistent(const DevBVH& bvh, const IO& io, unsigned int* __restrict__ work_counter,
                             uint2* lds_stack, int spill_lane, unsigned long long* counters) {
    const uint32_t n = io.n();
    const int lane = threadIdx.x & 63;
    const int count_mask = (1 << bvh.count_bits) - 1;
    const int tri_count_mask = INST ? ((1 << bvh.blas_count_bits) - 1) : count_mask;
    const int tri_count_bits = INST ? bvh.blas_count_bits : bvh.count_bits;
    LaneState s;
    InstState w;
    s.has_work = false;
    s.cur = 0;
    s.sp = 0;
    s.any = false;
    w.in_instance = false;
    w.hit_here = false;
    w.base_sp = 0;
    w.hit_inst = -1;
    w.cur_inst = -1;
    w.leaf_first = w.leaf_cnt = w.leaf_next = 0;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    const int n_seg = io.segments();
    int seg = (int)(blockIdx.x % (unsigned)n_seg), seg_tries = 0;
    uint32_t n_node = 0, n_prim = 0, n_inst = 0, n_rays = 0;
#ifdef PB_LANE_STATS
    unsigned long long stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    auto finish = [&](bool found) {
        io.store(s.index, s.any, found, s.tmax, s.b0, s.b1, s.b2, s.hit_slot, INST ? w.hit_inst : -1);
        s.has_work = false;
    };
    // TransformedPrimitive::intersect, first half (primitive.rs:136-139): world ray -> object space
    // (geometry.rs:865-881, 898-935), then the object aggregate's root box. Returns false if it misses.
    auto enter_instance = [&](int slot) -> bool {
        const float4* m = bvh.instances + 7 * (size_t)slot;
        float4 r0 = m[0], r1 = m[1], r2 = m[2];
        float x = w.wox, y = w.woy, z = w.woz;
        float ox = r0.x * x + r0.y * y + r0.z * z + r0.w;
        float oy = r1.x * x + r1.y * y + r1.z * z + r1.w;
        float oz = r2.x * x + r2.y * y + r2.z * z + r2.w;
        float xa = __builtin_fabsf(r0.x * x) + __builtin_fabsf(r0.y * y) + __builtin_fabsf(r0.z * z) + __builtin_fabsf(r0.w);
        float ya = __builtin_fabsf(r1.x * x) + __builtin_fabsf(r1.y * y) + __builtin_fabsf(r1.z * z) + __builtin_fabsf(r1.w);
        float za = __builtin_fabsf(r2.x * x) + __builtin_fabsf(r2.y * y) + __builtin_fabsf(r2.z * z) + __builtin_fabsf(r2.w);
        float ex = xa * kGamma3, ey = ya * kGamma3, ez = za * kGamma3;
        float dx = r0.x * w.wdx + r0.y * w.wdy + r0.z * w.wdz;
        float dy = r1.x * w.wdx + r1.y * w.wdy + r1.z * w.wdz;
        float dz = r2.x * w.wdx + r2.y * w.wdy + r2.z * w.wdz;
        float l2 = dx * dx + dy * dy + dz * dz;
        float tmax = w.tmax_world;
        if (l2 > 0.0f) {
            float dt = (__builtin_fabsf(dx) * ex + __builtin_fabsf(dy) * ey + __builtin_fabsf(dz) * ez) / l2;
            ox = ox + dx * dt;
            oy = oy + dy * dt;
            oz = oz + dz * dt;
            tmax -= dt;
        }
        s.r = TravRay{ox, oy, oz, dx, dy, dz, tmax};
        s.tmax = tmax;
        ray_constants(s);
        w.in_instance = true;
        w.hit_here = false;
        w.cur_inst = slot;
        w.base_sp = s.sp;
        if (COUNT) {
            n_inst += 1;
            n_node += 1;  // the object aggregate tests its root box (bvh.rs:841-842)
        }
        if (INST != 2) {
            if (!root_box_test(s, bvh.blas_root_min, bvh.blas_root_max)) return false;
            s.cur = bvh.blas_root_ref;
            return true;
        }
        const float4* ob = bvh.objects + 2 * (size_t)__float_as_int(m[6].z);
        const float4 o0 = ob[0], o1 = ob[1];
        const float mn[3] = {o0.x, o0.y, o0.z}, mx[3] = {o1.x, o1.y, o1.z};
        if (!root_box_test(s, mn, mx)) return false;
        s.cur = __float_as_int(o0.w);
        return true;
    };
    // second half (primitive.rs:140-143): r.t_max = ray.t_max on a hit; back to the world ray
    auto exit_instance = [&]() {
        if (w.hit_here) w.tmax_world = s.tmax;
        s.r = TravRay{w.wox, w.woy, w.woz, w.wdx, w.wdy, w.wdz, w.tmax_world};
        s.tmax = w.tmax_world;
        s.idx = 1.0f / s.r.dx;  // no triangles at the top level: the triangle constants are set at the next entry
        s.idy = 1.0f / s.r.dy;
        s.idz = 1.0f / s.r.dz;
        s.nx = s.idx < 0.0f;
        s.ny = s.idy < 0.0f;
        s.nz = s.idz < 0.0f;
        w.in_instance = false;
    };
    // After a node / leaf is done: find the next node to visit. Returns false when the ray is finished.
    // Leaving an instance (and entering the next one of the same top-level leaf) is NOT done here: advance() runs
    // inside the interior loop for single lanes, and the ray transform + root test are ~300 instructions. The lane
    // is parked on kLeaveInstance instead and the transition happens in the leaf phase, for all such lanes together.
    constexpr int kLeaveInstance = (int)0x80000000;
    auto advance = [&]() -> bool {
        for (;;) {
            if (INST && w.in_instance) {
                if (s.sp > w.base_sp) {
                    --s.sp;
                    uint2 ent = stack_read(bvh, lds_stack, spill_lane, s.sp);
                    if (COUNT) n_node += 1;
                    if (__uint_as_float(ent.y) < s.tmax) {
                        s.cur = (int)ent.x;
                        return true;
                    }
                    continue;
                }
                s.cur = kLeaveInstance;
                return true;
            }
            if (s.sp == 0) return false;
            --s.sp;
            uint2 ent = stack_read(bvh, lds_stack, spill_lane, s.sp);
            if (COUNT) n_node += 1;
            if (__uint_as_float(ent.y) < s.tmax) {
                s.cur = (int)ent.x;
                return true;
            }
        }
    };

    for (;;) {
        // ---------------- refill idle lanes ----------------
        unsigned long long idle_mask = __ballot(!s.has_work);
        int n_idle = __popcll(idle_mask);
        if (!exhausted && (n_idle >= kRefillThresh)) {
            if (chunk_next >= chunk_end) {
                // next chunk of this workgroup's queue segment; when the segment has run dry, of the following ones
                while (seg_tries < n_seg) {
                    uint32_t seg_begin = (uint32_t)(((unsigned long long)n * (unsigned)seg) / (unsigned)n_seg);
                    uint32_t seg_end = (uint32_t)(((unsigned long long)n * (unsigned)(seg + 1)) / (unsigned)n_seg);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(work_counter + seg, (unsigned int)kChunk);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane(base) + seg_begin;
                    if (base < seg_end && base >= seg_begin) {
                        chunk_next = base;
                        chunk_end = (base + kChunk) < seg_end ? (base + kChunk) : seg_end;
                        break;
                    }
                    seg = (seg + 1 == n_seg) ? 0 : seg + 1;
                    seg_tries += 1;
                }
                if (chunk_next >= chunk_end) exhausted = true;
            }
            uint32_t avail = chunk_end - chunk_next;
            uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0));
            bool take = !s.has_work && prefix < avail;
            uint32_t my = chunk_next + prefix;
            chunk_next += ((uint32_t)n_idle < avail) ? (uint32_t)n_idle : avail;
            if (take) {
                s.index = io.token(my);
                bool real = io.load(s.index, &s.r, &s.any);
                s.tmax = s.r.tmax;
                s.hit_slot = -1;
                s.b0 = s.b1 = s.b2 = 0.0f;
                s.sp = 0;
                s.has_work = true;
                if (INST) {
                    w.wox = s.r.ox;
                    w.woy = s.r.oy;
                    w.woz = s.r.oz;
                    w.wdx = s.r.dx;
                    w.wdy = s.r.dy;
                    w.wdz = s.r.dz;
                    w.tmax_world = s.r.tmax;
                    w.in_instance = false;
                    w.hit_inst = -1;
                    w.leaf_cnt = w.leaf_next = 0;
                }
                if (!real) {
                    finish(false);  // placeholder of a path outside pixel_bounds: not a ray of the frame
                } else {
                    ray_constants(s);
                    s.cur = bvh.root_ref;
                    if (COUNT) {
                        n_rays += 1;
                        n_node += 1;
                    }
                    if (!root_box_test(s, bvh.root_min, bvh.root_max)) finish(false);
                }
            }
        }
        if (!__any(s.has_work)) {
            if (exhausted) break;
            continue;
        }

        // ---------------- interior nodes (both levels) ----------------
        for (;;) {
            bool interior = s.has_work && s.cur >= 0;
            unsigned long long im = __ballot(interior);
            int n_int = __popcll(im);
            if (n_int == 0) break;
            if (n_int < kInteriorThresh) {
                // too few lanes at interior nodes: stop if someone has a leaf to process or idle
                // lanes could be refilled; otherwise keep going (nothing better to do)
                bool leaf_pending = __any(s.has_work && s.cur < 0);
                bool can_refill = !exhausted && (__popcll(__ballot(!s.has_work)) >= kRefillThresh);
                if (leaf_pending || can_refill) break;
            }
            PB_STAT(0, 1);      // interior iterations of this wave
            PB_STAT(1, n_int);  // lanes doing a node step
            if (interior) {
                const float4* nd = bvh.inodes + 4 * (size_t)s.cur;
                float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
                float e0, e1;
                bool k0, k1;  // the t_max-independent part of the two tests (see far_keep)
                bool h0 = slab_test_keep(s.nx ? q0.w : q0.x, s.nx ? q0.x : q0.w, s.ny ? q1.x : q0.y, s.ny ? q0.y : q1.x,
                                         s.nz ? q1.y : q0.z, s.nz ? q0.z : q1.y, s.r, s.idx, s.idy, s.idz, s.tmax, &e0, &k0);
                bool h1 = slab_test_keep(s.nx ? q2.y : q1.z, s.nx ? q1.z : q2.y, s.ny ? q2.z : q1.w, s.ny ? q1.w : q2.z,
                                         s.nz ? q2.w : q2.x, s.nz ? q2.x : q2.w, s.r, s.idx, s.idy, s.idz, s.tmax, &e1, &k1);
                int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y), axis = __float_as_int(q3.z);
                bool neg = axis == 0 ? s.nx : (axis == 1 ? s.ny : s.nz);  // bvh.rs:857-865
                int near_c = neg ? c1 : c0, far_c = neg ? c0 : c1;
                bool near_h = neg ? h1 : h0, far_h = neg ? h0 : h1;
                float far_e = neg ? e0 : e1;
                // The reference tests the far child when it POPS it, with the t_max of that moment — and t_max can move UP by
                // an ulp: the range test of triangle.rs:127-130 compares t_scaled with t_max * det, the quotient t = t_scaled /
                // det is rounded again and may come out one ulp above the t_max it was accepted under (two hits within an ulp
                // of each other: rays through a shared vertex). A far child behind the hit NOW may be in front of it THEN, so
                // whether it is kept cannot depend on t_max here: it is kept whenever the ray meets its slabs at all, and its
                // entry distance is compared with the t_max current at the pop, which is exactly the reference's test
                // (found by tests/test_gpu_wide.py's adversarial meshes at 3000 examples, round 3).
                const bool far_keep = neg ? k0 : k1;
                if (COUNT) {
                    n_node += 1;               // the near child is tested as soon as it is visited
                    if (!near_h) n_node += 1;  // near missed: the far child is popped and tested next
                    if (!far_keep) far_e = kInf;
                }
                if (near_h) {
                    s.cur = near_c;
                    if (far_keep || COUNT) stack_push(bvh, lds_stack, spill_lane, s.sp, far_c, far_e);
                } else if (far_h) {
                    s.cur = far_c;
                } else if (!advance()) {
                    finish(s.hit_slot >= 0);
                }
            }
        }

        // ---------------- leaves ----------------
#ifdef PB_LANE_STATS
        {
            unsigned long long lm = __ballot(s.has_work && s.cur < 0);
            if (lm) {
                int my_cnt = (s.has_work && s.cur < 0) ? (((~s.cur) & tri_count_mask) + 1) : 0, mx = my_cnt, sum = my_cnt;
                for (int o = 32; o > 0; o >>= 1) {
                    mx = max(mx, __shfl_xor(mx, o, 64));
                    sum += __shfl_xor(sum, o, 64);
                }
                PB_STAT(2, 1);                // leaf sections
                PB_STAT(3, __popcll(lm));     // lanes with a leaf
                PB_STAT(4, mx);               // triangle-loop trips of the wave
                PB_STAT(5, sum);              // triangle tests
            }
            PB_STAT(6, 1);                    // outer iterations
            PB_STAT(7, __popcll(__ballot(s.has_work)));
        }
#endif
        if (s.has_work && s.cur < 0) {
            if (INST && (!w.in_instance || s.cur == kLeaveInstance)) {
                if (w.in_instance) {
                    exit_instance();  // second half of TransformedPrimitive::intersect; the leaf's other instances follow
                } else {
                    // top-level leaf: TransformedPrimitives in leaf order (bvh.rs:844-850)
                    int ref = ~s.cur;
                    w.leaf_cnt = (ref & count_mask) + 1;
                    w.leaf_first = ref >> bvh.count_bits;
                    w.leaf_next = 0;
                }
                bool entered = false, done = false;
                while (w.leaf_next < w.leaf_cnt && !entered && !done) {
                    int slot = w.leaf_first + w.leaf_next;
                    w.leaf_next += 1;
                    float4 kind = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (INST == 2) kind = bvh.instances[7 * (size_t)slot + 6];
                    if (INST == 2 && __float_as_int(kind.w) == 1) {
                        // a GeometricPrimitive beside the instances (primitive.rs:65-78): the world ray against its triangle
                        const int tslot = __float_as_int(kind.z);
                        V3 p0, p1, p2;
                        int flags;
                        load_tri(bvh.tris, tslot, &p0, &p1, &p2, &flags);
                        if (COUNT) n_prim += 1;
                        float b0, b1, b2, t;
                        const TriRayConst c = tri_ray_setup(s.r);
                        if (triangle_test(p0, p1, p2, s.r, c, s.tmax, &b0, &b1, &b2, &t)) {
                            if (s.any) {
                                done = !(io.strict(s.index) && (flags & kTriDegenerate));
                            } else if (!(flags & kTriDegenerate)) {
                                s.tmax = t;
                                w.tmax_world = t;
                                s.b0 = b0;
                                s.b1 = b1;
                                s.b2 = b2;
                                s.hit_slot = tslot;
                                w.hit_inst = -1;
                            }
                        }
                        continue;
                    }
                    entered = enter_instance(slot);
                    if (!entered) exit_instance();
                }
                if (done)
                    finish(true);
                else if (!entered && !advance())
                    finish(s.hit_slot >= 0);
            } else {
                int ref = ~s.cur;
                int cnt = (ref & tri_count_mask) + 1;
                int first = ref >> tri_count_bits;
                bool done = false;
                for (int i = 0; i < cnt; ++i) {
                    V3 p0, p1, p2;
                    int flags;
                    load_tri(bvh.tris, first + i, &p0, &p1, &p2, &flags);
                    float b0, b1, b2, t;
                    if (COUNT) n_prim += 1;
                    bool hit;
                    if (SPH && (flags & kPrimSphere)) {
                        // Sphere::intersect_test (sphere.rs:228-284); the hit record carries the refined hit point
                        V3 ph;
                        float phi;
                        hit = sphere_test(p0.x, p0.y, p0.z, p1.x, s.r, s.tmax, &t, &ph, &phi);
                        b0 = ph.x;  // a sphere's hit record carries the refined object-space hit point
                        b1 = ph.y;
                        b2 = ph.z;
                    } else {
                        hit = triangle_test(p0, p1, p2, s.r, s.trc, s.tmax, &b0, &b1, &b2, &t);
                    }
                    if (hit) {
                        if (s.any) {
                            if (io.strict(s.index) && (flags & kTriDegenerate)) continue;  // (see IO::strict)
                            done = true;
                            break;
                        }
                        if (!(flags & kTriDegenerate)) {
                            s.tmax = t;  // primitive.rs:70
                            s.b0 = b0;
                            s.b1 = b1;
                            s.b2 = b2;
                            s.hit_slot = first + i;
                            if (INST) {
                                w.hit_here = true;
                                w.hit_inst = w.cur_inst;
                            }
                        }
                    }
                }
                if (done) {
                    s.hit_slot = first;
                    finish(true);
                } else if (!advance()) {
                    finish(s.hit_slot >= 0);
                }
            }
        }
    }
    if (COUNT) count_flush(counters, n_node, n_prim, n_rays, n_inst);
#ifdef PB_LANE_STATS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_lane_stats[i], stat[i]);
#endif
}

}  // namespace pb
