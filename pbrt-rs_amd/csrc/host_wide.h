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
};

// Returns nullptr on success, else the reason the scene keeps the binary records only.
// tris: the 48-B leaf-order triangle records of the scene (12 floats per slot: 9 vertex floats, prim, material, flags).
const char* build_wide_tree(const PbrtLinearBVHNode* nodes, int32_t n_nodes, const float* tris, int32_t n_slots, WideTree* out);

}  // namespace pb
