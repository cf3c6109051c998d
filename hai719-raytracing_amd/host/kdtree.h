// Flattened KD-tree with ropes -- the host-side builder that replaces the
// reference's pointer tree (KDTree.cpp:87-151, "median of triangle mins,
// axis = depth%3, leaf <= 40").  The reference's trees are near-degenerate
// (3 nodes for flamingo_lowpoly); any tree over the same
// triangles returns the same closest hit, so this builder uses
// a surface-area heuristic and emits the 16-byte "nodelet" array that the
// HIP kernel walks without a stack (include/hrt.h, hrt_kdunit).
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/hrt.h"

namespace hrt_host {

struct KDBuildParams {
    uint32_t leaf_max = 4;      // stop splitting at this many triangles
    uint32_t max_depth = 0;     // 0 = 8 + 1.3*log2(n)
    float cost_traverse = 1.0f;
    float cost_intersect = 1.5f;
    float empty_bonus = 0.8f;   // SAH multiplier when one side is empty
    uint32_t threads = 0;       // worker threads for the subtrees below depth 5; 0 = hardware_concurrency (HRT_KD_THREADS overrides)
    hrt_kd_builder_fn builder = nullptr;  // replaces the split search + partition (include/hrt.h; e.g. hrt_kd_build_gpu); ropes and
    void *builder_user = nullptr;         // flattening stay here
};

struct FlatKDTree {
    std::vector<hrt_kdunit> units;
    std::vector<uint32_t> leaf_tris;
    uint32_t root = HRT_KD_NIL;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};  // root cell
    // statistics (reported by tests / DESIGN.md)
    uint32_t n_inner = 0, n_leaves = 0, n_empty_leaves = 0, depth = 0;
};

// positions: 3*nv floats ALREADY multiplied by HRT_TRIANGLE_SCALING (the
// triangles the kernel intersects); indices: 3*nt.  skip (nt bytes or NULL): triangles with skip[t] != 0 stay out of the
// tree (the irregular triangles of ref_tree.h, which the kernel tests through their reference leaf boxes instead).
FlatKDTree build_flat_kdtree(const float *positions, uint32_t nv, const uint32_t *indices,
                             uint32_t nt, const KDBuildParams &params = KDBuildParams(), const uint8_t *skip = nullptr);

}  // namespace hrt_host
