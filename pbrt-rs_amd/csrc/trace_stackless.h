// trace_stackless.h — the binary records walked WITHOUT a traversal stack: a 64-bit trail per lane and parent links.
//
// BASELINE.json's north_star names "a stackless BVH traversal kernel"; this is that kernel, selected with
// pbrt_hip_context_set_traversal(ctx, PBRT_TRAVERSAL_STACKLESS), parity-tested like the others and measured beside them
// (DESIGN.md section 4.6). It is not the default: see the numbers there.
//
// What has to be preserved is the reference's ORDER (src/accelerators/bvh.rs:828-932): depth first, the child on the
// ray's side of the split axis first (bvh.rs:857-865), the other child pushed untested and tested with the ray.t_max of
// the moment it is POPPED (bvh.rs:841-842 at the top of the next iteration). A stack of (node, entry distance) pairs
// does that in trace_persistent.h. Here the stack is replaced by
//   * a parent link in every 64-B child-pair record (the fourth field of its last float4, trace.h), and
//   * one bit per level of the current root-to-node path, newest level in bit 0: 1 = "the far child of the node on that
//     level is still to be visited" (a bit trail, after the skip-pointer-free family of stackless walks: Hapala et al.
//     2011 walk parent links with a three-state automaton; Afra & Szirmay-Kalos 2013 add the bit per level so that a far
//     child that was never reached is not tested again on the way up).
// Going DOWN at an interior node: both children's slabs are tested as in trace_persistent.h; descending into the near
// child shifts the trail left and records whether the far child's slabs are met at all (t_max does not enter that bit:
// t_max can move up by an ulp before the pop, trace_persistent.h). Going UP (a leaf is done, or both children missed):
// the lane re-reads the parent's record; if the level's bit is set it clears it and tests the far child's box with the
// CURRENT t_max — the reference's pop-time test, same operands — and descends there when it passes; otherwise it shifts
// the trail right and continues one level higher, until the root's parent link (-1) ends the ray. The visiting order and
// every box / triangle decision are those of the stack walk; the price is one dependent record fetch per level climbed
// where the stack walk pays one LDS read per pop.
//
// A trail of 64 bits covers 64 levels; both tree builders refuse trees deeper than the reference's 64-entry stack
// (pbrt_hip.hip convert_tree, hlbvh_gpu.hip), so no path outgrows it. Single-level triangle scenes only (INST = 0, no
// spheres): the selector refuses the others.
#pragma once
#include "trace_persistent.h"

namespace pb {

template <class IO>
PB_DEV void trace_stackless(const DevBVH& bvh, const IO& io, unsigned int* __restrict__ work_counter) {
    const uint32_t n = io.n();
    const int lane = threadIdx.x & 63;
    const int count_mask = (1 << bvh.count_bits) - 1;
    LaneState s;
    s.has_work = false;
    s.cur = 0;
    s.any = false;
    // one bit per interior node from the root to the PARENT of the node the lane goes down into next — or, going up, to the
    // node it re-enters — newest in bit 0
    unsigned long long trail = 0;
    int par = -1;                  // parent record of a pending leaf
    bool up = false;               // s.cur >= 0: the record to read; up = it is being re-entered from below
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    const int n_seg = io.segments();
    int seg = (int)(blockIdx.x % (unsigned)n_seg), seg_tries = 0;

    auto finish = [&](bool found) {
        io.store(s.index, s.any, found, s.tmax, s.b0, s.b1, s.b2, s.hit_slot, -1);
        s.has_work = false;
    };

    for (;;) {
        // ---------------- refill idle lanes (as trace_persistent.h) ----------------
        unsigned long long idle_mask = __ballot(!s.has_work);
        int n_idle = __popcll(idle_mask);
        if (!exhausted && (n_idle >= kRefillThresh)) {
            if (chunk_next >= chunk_end) {
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
                s.has_work = true;
                trail = 0;
                par = -1;
                up = false;
                if (!real) {
                    finish(false);
                } else {
                    ray_constants(s);
                    s.cur = bvh.root_ref;
                    if (!root_box_test(s, bvh.root_min, bvh.root_max)) finish(false);
                }
            }
        }
        if (!__any(s.has_work)) {
            if (exhausted) break;
            continue;
        }

        // ---------------- record steps: down into a node, or up through it ----------------
        for (;;) {
            bool interior = s.has_work && s.cur >= 0;
            int n_int = __popcll(__ballot(interior));
            if (n_int == 0) break;
            if (n_int < kInteriorThresh) {
                bool leaf_pending = __any(s.has_work && s.cur < 0);
                bool can_refill = !exhausted && (__popcll(__ballot(!s.has_work)) >= kRefillThresh);
                if (leaf_pending || can_refill) break;
            }
            if (interior) {
                const float4* nd = bvh.inodes + 4 * (size_t)s.cur;
                // a lane climbing through a node whose far child is not pending needs the parent link only: one 16-B
                // request instead of four (the kernel sits on the texture addressers). The condition is in registers, so
                // the loads that are issued still go out together.
                const bool pending = (trail & 1ull) != 0;
                const float4 q3 = nd[3];
                float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q1 = q0, q2 = q0;
                if (!up || pending) {
                    q0 = nd[0];
                    q1 = nd[1];
                    q2 = nd[2];
                }
                // (a lane that only climbs holds zeros in q0..q2: its box results are forced to false below rather than left
                // to whatever 0-sized boxes give for that ray — nothing reads them on that path, go_near / go_far are masked)
                const bool boxes = !up || pending;
                float e0, e1;
                bool k0, k1;
                bool h0 = slab_test_keep(s.nx ? q0.w : q0.x, s.nx ? q0.x : q0.w, s.ny ? q1.x : q0.y, s.ny ? q0.y : q1.x,
                                         s.nz ? q1.y : q0.z, s.nz ? q0.z : q1.y, s.r, s.idx, s.idy, s.idz, s.tmax, &e0, &k0);
                bool h1 = slab_test_keep(s.nx ? q2.y : q1.z, s.nx ? q1.z : q2.y, s.ny ? q2.z : q1.w, s.ny ? q1.w : q2.z,
                                         s.nz ? q2.w : q2.x, s.nz ? q2.x : q2.w, s.r, s.idx, s.idy, s.idz, s.tmax, &e1, &k1);
                h0 = h0 && boxes;
                h1 = h1 && boxes;
                k0 = k0 && boxes;
                k1 = k1 && boxes;
                const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y), axis = __float_as_int(q3.z);
                const int parent = __float_as_int(q3.w);
                const bool neg = axis == 0 ? s.nx : (axis == 1 ? s.ny : s.nz);  // bvh.rs:857-865
                const int near_c = neg ? c1 : c0, far_c = neg ? c0 : c1;
                const bool near_h = neg ? h1 : h0, far_h = neg ? h0 : h1;
                const bool far_keep = neg ? k0 : k1;
                // down: near child first, the far one left as a bit; up: the far child if its bit is set and its box
                // passes NOW (bvh.rs:841-842 at the pop), else one level higher
                const bool go_near = !up && near_h;
                const bool go_far = up ? (pending && far_h) : (!near_h && far_h);
                if (go_near) {
                    trail = (trail << 1) | (far_keep ? 1ull : 0ull);
                    par = s.cur;
                    s.cur = near_c;
                } else if (go_far) {
                    trail = up ? (trail & ~1ull) : (trail << 1);  // up: this node's bit is there already, now cleared
                    par = s.cur;
                    s.cur = far_c;
                    up = false;
                } else {
                    // nothing (left) below this node. Coming down, the node's own bit is not on the trail yet;
                    // coming up, it is dropped.
                    if (up) trail >>= 1;
                    up = true;
                    s.cur = parent;
                    if (parent < 0) finish(s.hit_slot >= 0);
                }
            }
        }

        // ---------------- leaves (as trace_persistent.h) ----------------
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
                        if (io.strict(s.index) && (flags & kTriDegenerate)) continue;
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
                s.hit_slot = first;
                finish(true);
            } else if (par < 0) {
                finish(s.hit_slot >= 0);  // the tree is one leaf
            } else {
                up = true;  // re-enter the parent from below: its bit is bit 0 of the trail
                s.cur = par;
            }
        }
    }
}

}  // namespace pb
