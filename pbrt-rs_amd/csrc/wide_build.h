// wide_build.h — the per-node / per-record arithmetic of laying 4-wide quantised records (wide_bvh.h) over a
// reference-order LinearBVHNode array, shared by the host builder (host_wide.cpp) and the device builder
// (wide_gpu.hip): one source, so the two produce the same bytes. Every rounding here is outward: a quantised box
// contains the float box it stands for in exact arithmetic, so the traversal kernel's filter can only pass more, never
// less (wide_bvh.h). Double arithmetic throughout (exactly rounded operations only: +, -, ldexp, floor, ceil,
// nextafter).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/pbrt_hip.h"
#include "wide_bvh.h"

namespace pb {

// why a tree keeps the binary records only
enum WideBuildError {
    kWideOk = 0,
    kWideErrCoords,
    kWideErrInverted,
    kWideErrLeafSize,
    kWideErrLooseLeaf,
    kWideErrChildOutside,
    kWideErrExponent,
    kWideErrQuantisation,
    kWideErrCoarse,
};
// A record is COARSE when keeping m[0..2] in the low mantissa bytes of base.xyz (a displacement of up to 256 ulp of the
// coordinate) forces cells more than 8 times larger than the node's largest extent needs: far from the origin the 8-bit planes
// then stop filtering — hits stay exact, steps per ray do not. A tree with more than one coarse record in eight keeps the
// binary records.
constexpr int kWideCoarseShift = 3;
inline const char* wide_error_text(int code) {
    switch (code) {
        case kWideOk: return nullptr;
        case kWideErrCoords: return "coordinates beyond 2^20";
        case kWideErrInverted: return "inverted node box";
        case kWideErrLeafSize: return "leaf with more than 4 primitives";
        case kWideErrLooseLeaf: return "single-triangle leaf box is not the triangle's bounds";
        case kWideErrChildOutside: return "child box not inside its parent's";
        case kWideErrExponent: return "node extent beyond the exponent range";
        case kWideErrCoarse: return "scene too far from the origin for the 8-bit planes (one record in eight would not filter)";
        default: return "quantisation out of range";
    }
}

PB_HD uint32_t wb_f2u(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
PB_HD float wb_u2f(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
PB_HD double wb_next_up(double x) { return nextafter(x, __builtin_huge_val()); }
PB_HD double wb_next_down(double x) { return nextafter(x, -__builtin_huge_val()); }
// the largest float <= target whose low mantissa byte is `m` (the record keeps m[0..2] in the low bytes of base.xyz)
PB_HD float wb_base_with_byte(double target, uint32_t m) {
    float t = (float)target;
    if ((double)t > target) t = nextafterf(t, -__builtin_huge_valf());
    uint32_t u = wb_f2u(t);
    if (!(u & 0x80000000u)) {  // t >= +0: float order = integer order
        uint32_t cand = (u & ~0xffu) | m;
        if (cand <= u) return wb_u2f(cand);
        if (u >= 0x100u) return wb_u2f(cand - 0x100u);
        return wb_u2f(0x80000000u | m);  // below the smallest positive step: a tiny negative value (or -0)
    }
    uint32_t mag = u & 0x7fffffffu, cand = (mag & ~0xffu) | m;  // negative: the magnitude has to be >= |t|
    if (cand < mag) cand += 0x100u;
    return wb_u2f(0x80000000u | cand);
}
// smallest exponent e >= kExpMin with (255 - 2 kWideSlack) * 2^e >= extent
PB_HD int wb_cell_exponent(double extent) {
    const double cells = 255.0 - 2.0 * kWideSlack;
    if (!(extent > 0.0)) return kExpMin;
    int e = ilogb(extent / cells);
    while (ldexp(cells, e) < extent) ++e;
    while (e > kExpMin && ldexp(cells, e - 1) >= extent) --e;
    return e > kExpMin ? e : kExpMin;
}

// the properties the exactness argument needs, per node: checked, not assumed. tris: the 48-B leaf-order records
// (12 floats per slot), nullptr for opaque primitives.
PB_HD int wide_check_node(const PbrtLinearBVHNode* nodes, int32_t i, const float* tris) {
    const PbrtLinearBVHNode& nd = nodes[i];
    for (int k = 0; k < 3; ++k) {
        if (!(fabsf(nd.bounds_min[k]) <= kWideCoordLimit) || !(fabsf(nd.bounds_max[k]) <= kWideCoordLimit)) return kWideErrCoords;
        if (!(nd.bounds_min[k] <= nd.bounds_max[k])) return kWideErrInverted;
    }
    if (nd.n_primitives > 0) {
        if (nd.n_primitives > 4) return kWideErrLeafSize;
        if (nd.n_primitives == 1 && tris) {
            // the single-triangle leaf box is recomputed from the vertices by the kernel: it has to BE the tight box
            const float* t = tris + 12 * (size_t)nd.offset;
            for (int k = 0; k < 3; ++k) {
                float lo = fminf(t[k], fminf(t[3 + k], t[6 + k]));
                float hi = fmaxf(t[k], fmaxf(t[3 + k], t[6 + k]));
                if (lo != nd.bounds_min[k] || hi != nd.bounds_max[k]) return kWideErrLooseLeaf;
            }
        }
    } else {
        const PbrtLinearBVHNode* ch[2] = {&nodes[i + 1], &nodes[nd.offset]};
        for (int c = 0; c < 2; ++c)
            for (int k = 0; k < 3; ++k)
                if (ch[c]->bounds_min[k] < nd.bounds_min[k] || ch[c]->bounds_max[k] > nd.bounds_max[k]) return kWideErrChildOutside;
    }
    return kWideOk;
}

// the four slots of the record of interior node i: binary node per slot (-1 empty), split axes of the two binary children
PB_HD void wide_slots_of(const PbrtLinearBVHNode* nodes, int32_t i, int32_t slot_node[4], int axis_c[2]) {
    const PbrtLinearBVHNode& nd = nodes[i];
    const int32_t c[2] = {i + 1, nd.offset};
    slot_node[0] = slot_node[1] = slot_node[2] = slot_node[3] = -1;
    axis_c[0] = axis_c[1] = 0;
    for (int j = 0; j < 2; ++j) {
        if (nodes[c[j]].n_primitives > 0) {
            slot_node[2 * j] = c[j];
        } else {
            slot_node[2 * j] = c[j] + 1;
            slot_node[2 * j + 1] = nodes[c[j]].offset;
            axis_c[j] = nodes[c[j]].axis;
        }
    }
}

// The record of interior node i. first_child / first_tri: absolute record index of its first interior child, absolute
// wide-order position of its first leaf child's first triangle. tri_off_of_slot[s]: offset of slot s's triangles inside
// the record's run (leaf slots only). Returns a WideBuildError.
PB_HD int wide_make_record(const PbrtLinearBVHNode* nodes, int32_t i, const int32_t slot_node[4], const int axis_c[2],
                           uint32_t first_child, uint32_t first_tri, uint32_t rec[kWideNodeDwords], int tri_off_of_slot[4],
                           int* coarse = nullptr) {
    const PbrtLinearBVHNode& nd = nodes[i];
    // children first: the bytes m[] are part of base.xyz
    uint32_t m[4] = {0xff, 0xff, 0xff, 0xff};
    int n_interior = 0, tri_off = 0;
    for (int s = 0; s < 4; ++s) {
        tri_off_of_slot[s] = 0;
        if (slot_node[s] < 0) continue;
        const PbrtLinearBVHNode& ch = nodes[slot_node[s]];
        if (ch.n_primitives > 0) {
            tri_off_of_slot[s] = tri_off;
            m[s] = (uint32_t)(tri_off << 2) | (uint32_t)(ch.n_primitives - 1);
            tri_off += ch.n_primitives;
        } else {
            m[s] = 0x80u | (uint32_t)n_interior;
            ++n_interior;
        }
    }
    float base[3];
    int e[3];
    for (int k = 0; k < 3; ++k) {
        // base: kWideSlack cells (and a little more) below the lower corner; cell 2^e: the smallest with 255 cells
        // reaching kWideSlack cells beyond the upper corner. The two depend on each other: settle in a few rounds.
        int ek = kExpMin;
        for (int round = 0; round < 8; ++round) {
            base[k] = wb_base_with_byte((double)nd.bounds_min[k] - 2.0 * kWideSlack * ldexp(1.0, ek), m[k]);
            int need = wb_cell_exponent(wb_next_up((double)nd.bounds_max[k] - (double)base[k]));
            if (need <= ek) break;
            ek = need;
        }
        if (ek > kExpMax) return kWideErrExponent;
        e[k] = ek;
    }
    if (coarse) {  // against the node's LARGEST extent: a thick cell on a flat node's thin axis costs nothing
        double ext = 0.0;
        for (int k = 0; k < 3; ++k) ext = fmax(ext, (double)nd.bounds_max[k] - (double)nd.bounds_min[k]);
        const int need = wb_cell_exponent(ext);
        for (int k = 0; k < 3; ++k)
            if (e[k] > need + kWideCoarseShift) *coarse = 1;
    }
    uint32_t q[6] = {0, 0, 0, 0, 0, 0};
    for (int s = 0; s < 4; ++s) {
        uint32_t qlo[3] = {255, 255, 255}, qhi[3] = {0, 0, 0};  // empty slot: inverted (and masked out by m = 0xFF)
        if (slot_node[s] >= 0) {
            const PbrtLinearBVHNode& ch = nodes[slot_node[s]];
            for (int k = 0; k < 3; ++k) {
                double dlo = wb_next_down((double)ch.bounds_min[k] - (double)base[k]);
                double dhi = wb_next_up((double)ch.bounds_max[k] - (double)base[k]);
                double flo = floor(ldexp(dlo, -e[k]) - kWideSlack), fhi = ceil(ldexp(dhi, -e[k]) + kWideSlack);
                if (flo < 0.0 || fhi > 255.0 || flo > fhi) return kWideErrQuantisation;  // excluded by the choice of base and e
                qlo[k] = (uint32_t)flo;
                qhi[k] = (uint32_t)fhi;
            }
        }
        for (int k = 0; k < 3; ++k) {
            q[2 * k] |= qlo[k] << (8 * s);
            q[2 * k + 1] |= qhi[k] << (8 * s);
        }
    }
    for (int k = 0; k < 3; ++k) rec[k] = wb_f2u(base[k]);
    rec[3] = ((uint32_t)e[0] & 63u) | ((uint32_t)e[1] & 63u) << 6 | ((uint32_t)e[2] & 63u) << 12 | (uint32_t)nd.axis << 18 |
             (uint32_t)axis_c[0] << 20 | (uint32_t)axis_c[1] << 22 | m[3] << 24;
    for (int k = 0; k < 6; ++k) rec[4 + k] = q[k];
    rec[10] = first_child;
    rec[11] = ~(first_tri << 2);
    return kWideOk;
}

// a leaf's triangles to wide-order position `first` (12 floats per position: 9 vertex floats, leaf slot, flags, 0) and,
// where the kernel reads one (n >= 2), its exact box
PB_HD void wide_emit_leaf(const PbrtLinearBVHNode& lf, const float* tris, int32_t slot_base, size_t first, float* out_tris,
                          float* out_boxes) {
    for (int j = 0; j < lf.n_primitives; ++j) {
        const float* src = tris + 12 * (size_t)(lf.offset + j);
        float* dst = out_tris + 12 * (first + (size_t)j);
        for (int k = 0; k < 9; ++k) dst[k] = src[k];
        int32_t slot = lf.offset + j + slot_base;
        __builtin_memcpy(dst + 9, &slot, 4);
        __builtin_memcpy(dst + 10, src + 11, 4);  // flags
        dst[11] = 0.0f;
    }
    if (lf.n_primitives >= 2) {
        float* b = out_boxes + 8 * first;
        for (int k = 0; k < 3; ++k) {
            b[k] = lf.bounds_min[k];
            b[4 + k] = lf.bounds_max[k];
        }
    }
}

}  // namespace pb
