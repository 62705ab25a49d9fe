// host_wide.h — result of host_wide.cpp's build_wide_tree (the 4-wide quantised records of wide_bvh.h, host arrays).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/pbrt_hip.h"

namespace pb {

struct WideTree {
    std::vector<uint32_t> nodes;    // 12 dwords per record
    std::vector<float> tris;        // 12 floats per wide-order triangle: 9 vertex floats, leaf slot, flags, 0
    std::vector<float> leaf_boxes;  // 8 floats per wide-order triangle position; filled at the first triangle of a leaf with n >= 2
    int32_t root_ref = 0;
    int n_records = 0;
    int stack_need = 0;  // most entries any ray can have on the traversal stack
    std::vector<int32_t> order;  // opaque primitives only: wide-order position -> leaf slot
};

// Where a tree's records, triangles and leaf slots sit when several trees share the device arrays (two-level scenes:
// the top-level tree, then one tree per object aggregate). All zero for a single-level scene.
struct WideBase {
    int32_t record = 0;  // index of this tree's first record in the shared record array
    int32_t tri = 0;     // wide-order position of this tree's first primitive
    int32_t slot = 0;    // leaf slot of this tree's first primitive (written into the wide triangles)
};

// Returns nullptr on success, else the reason the scene keeps the binary records only.
// tris: the 48-B leaf-order triangle records of the tree's primitives (12 floats per slot: 9 vertex floats, prim,
// material, flags). tris == nullptr: the primitives are opaque (the TransformedPrimitives / triangles of a top-level
// aggregate): every leaf then keeps its exact box (leaf_boxes, at every position of the leaf) and out->order[position] = the
// leaf slot of the primitive at that wide-order position; no triangles are copied.
const char* build_wide_tree(const PbrtLinearBVHNode* nodes, int32_t n_nodes, const float* tris, int32_t n_slots, WideTree* out,
                            const WideBase& base = WideBase());

}  // namespace pb
