// trace_persistent.h — persistent-threads traversal with per-lane ray replacement.
//
// Same arithmetic and visiting order as traverse() in trace.h (BVHAccel::intersect / intersect_p,
// src/accelerators/bvh.rs:828-932); what changes is how rays are scheduled onto the 64 lanes of a
// wavefront. Ray lengths in an incoherent batch are heavy-tailed, so a wave that traces 64 rays
// to completion idles most lanes (measured: ~9 % VALU lane utilisation). Here
//   * a wave owns a contiguous chunk of the ray queue (one atomicAdd per kChunk rays),
//   * a lane that finishes its ray takes the next ray of the chunk (refill is batched: it runs
//     when >= kRefillThresh lanes are idle, one ballot/mbcnt prefix, no atomics),
//   * interior-node steps run in a tight loop while enough lanes are at interior nodes; lanes
//     that reached a leaf wait and the leaf (<= max_prims triangles) is processed for all of them
//     together (while-while with postponed leaves).
// Traversal state lives in registers, the stack in LDS ([entry][lane]) with a global spill slab.
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

struct LaneState {
    TravRay r;
    float idx, idy, idz;
    TriRayConst trc;
    float tmax;
    float b0, b1, b2;
    int hit_slot;
    int cur, sp;
    uint32_t index;  // queue position of this lane's ray
    bool nx, ny, nz, any, has_work;
};

PB_DEV void stack_push(const DevBVH& bvh, uint2* lds_stack, int spill_lane, int& sp, int node, float entry) {
    uint2 ent = make_uint2((uint32_t)node, __float_as_uint(entry));
    if (sp < kStackLds)
        lds_stack[sp * kTraceBlock] = ent;
    else
        bvh.spill[(size_t)(sp - kStackLds) * bvh.spill_stride + spill_lane] = ent;
    ++sp;
}
// pop entries until one still beats t_max; returns false when the stack is empty (ray finished)
PB_DEV bool stack_pop(const DevBVH& bvh, uint2* lds_stack, int spill_lane, int& sp, float tmax, int& cur) {
    for (;;) {
        if (sp == 0) return false;
        --sp;
        uint2 ent = (sp < kStackLds) ? lds_stack[sp * kTraceBlock]
                                     : bvh.spill[(size_t)(sp - kStackLds) * bvh.spill_stride + spill_lane];
        if (__uint_as_float(ent.y) < tmax) {
            cur = (int)ent.x;
            return true;
        }
    }
}

// IO policy: n(), load(i, &ray, &any), store(i, any, found, t, b0, b1, b2, slot)
template <class IO>
PB_DEV void trace_persistent(const DevBVH& bvh, const IO& io, unsigned int* __restrict__ work_counter,
                             uint2* lds_stack, int spill_lane) {
    const uint32_t n = io.n();
    const int lane = threadIdx.x & 63;
    const int count_mask = (1 << bvh.count_bits) - 1;
    LaneState s;
    s.has_work = false;
    s.cur = 0;
    s.sp = 0;
    s.any = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;

    for (;;) {
        // ---------------- refill idle lanes ----------------
        unsigned long long idle_mask = __ballot(!s.has_work);
        int n_idle = __popcll(idle_mask);
        if (!exhausted && (n_idle >= kRefillThresh)) {
            if (chunk_next >= chunk_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (unsigned int)kChunk);
                base = __builtin_amdgcn_readfirstlane(base);
                chunk_next = base < n ? base : n;
                chunk_end = (base + kChunk) < n ? (base + kChunk) : n;
                if (chunk_next >= chunk_end) exhausted = true;
            }
            uint32_t avail = chunk_end - chunk_next;
            uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0));
            bool take = !s.has_work && prefix < avail;
            uint32_t my = chunk_next + prefix;
            chunk_next += ((uint32_t)n_idle < avail) ? (uint32_t)n_idle : avail;
            if (take) {
                s.index = my;
                io.load(my, &s.r, &s.any);
                s.tmax = s.r.tmax;
                s.idx = 1.0f / s.r.dx;  // bvh.rs:831
                s.idy = 1.0f / s.r.dy;
                s.idz = 1.0f / s.r.dz;
                s.nx = s.idx < 0.0f;    // bvh.rs:832-836
                s.ny = s.idy < 0.0f;
                s.nz = s.idz < 0.0f;
                s.trc = tri_ray_setup(s.r);
                s.hit_slot = -1;
                s.b0 = s.b1 = s.b2 = 0.0f;
                s.sp = 0;
                s.cur = bvh.root_ref;
                s.has_work = true;
                float e;
                bool root_hit = slab_test(s.nx ? bvh.root_max[0] : bvh.root_min[0], s.nx ? bvh.root_min[0] : bvh.root_max[0],
                                          s.ny ? bvh.root_max[1] : bvh.root_min[1], s.ny ? bvh.root_min[1] : bvh.root_max[1],
                                          s.nz ? bvh.root_max[2] : bvh.root_min[2], s.nz ? bvh.root_min[2] : bvh.root_max[2],
                                          s.r, s.idx, s.idy, s.idz, s.tmax, &e);
                if (!root_hit) {
                    io.store(s.index, s.any, false, s.tmax, 0.0f, 0.0f, 0.0f, -1);
                    s.has_work = false;
                }
            }
        }
        if (!__any(s.has_work)) {
            if (exhausted) break;
            continue;
        }

        // ---------------- interior nodes ----------------
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
            if (interior) {
                const float4* nd = bvh.inodes + 4 * (size_t)s.cur;
                float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
                float e0, e1;
                bool h0 = slab_test(s.nx ? q0.w : q0.x, s.nx ? q0.x : q0.w, s.ny ? q1.x : q0.y, s.ny ? q0.y : q1.x,
                                    s.nz ? q1.y : q0.z, s.nz ? q0.z : q1.y, s.r, s.idx, s.idy, s.idz, s.tmax, &e0);
                bool h1 = slab_test(s.nx ? q2.y : q1.z, s.nx ? q1.z : q2.y, s.ny ? q2.z : q1.w, s.ny ? q1.w : q2.z,
                                    s.nz ? q2.w : q2.x, s.nz ? q2.x : q2.w, s.r, s.idx, s.idy, s.idz, s.tmax, &e1);
                int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y), axis = __float_as_int(q3.z);
                bool neg = axis == 0 ? s.nx : (axis == 1 ? s.ny : s.nz);  // bvh.rs:857-865
                int near_c = neg ? c1 : c0, far_c = neg ? c0 : c1;
                bool near_h = neg ? h1 : h0, far_h = neg ? h0 : h1;
                float far_e = neg ? e0 : e1;
                if (near_h) {
                    s.cur = near_c;
                    if (far_h) stack_push(bvh, lds_stack, spill_lane, s.sp, far_c, far_e);
                } else if (far_h) {
                    s.cur = far_c;
                } else if (!stack_pop(bvh, lds_stack, spill_lane, s.sp, s.tmax, s.cur)) {
                    io.store(s.index, s.any, s.hit_slot >= 0, s.tmax, s.b0, s.b1, s.b2, s.hit_slot);
                    s.has_work = false;
                }
            }
        }

        // ---------------- leaves ----------------
        if (s.has_work && s.cur < 0) {
            int ref = ~s.cur;
            int cnt = (ref & count_mask) + 1;
            int first = ref >> bvh.count_bits;
            bool done = false;
            for (int i = 0; i < cnt; ++i) {
                V3 p0, p1, p2;
                int flags;
                load_tri(bvh.tris, first + i, &p0, &p1, &p2, &flags);
                float b0, b1, b2, t;
                if (triangle_test(p0, p1, p2, s.r, s.trc, s.tmax, &b0, &b1, &b2, &t)) {
                    if (s.any) {
                        done = true;
                        break;
                    }
                    if (!(flags & kTriDegenerate)) {
                        s.tmax = t;  // primitive.rs:70
                        s.b0 = b0;
                        s.b1 = b1;
                        s.b2 = b2;
                        s.hit_slot = first + i;
                    }
                }
            }
            if (done) {
                io.store(s.index, true, true, s.tmax, 0.0f, 0.0f, 0.0f, first);
                s.has_work = false;
            } else if (!stack_pop(bvh, lds_stack, spill_lane, s.sp, s.tmax, s.cur)) {
                io.store(s.index, s.any, s.hit_slot >= 0, s.tmax, s.b0, s.b1, s.b2, s.hit_slot);
                s.has_work = false;
            }
        }
    }
}

}  // namespace pb
