// Which triangles of a mesh the flattened SAH tree may NOT own, because the reference's own KD-tree treats them in a way
// no other tree reproduces (DESIGN.md "Irregular triangles").
//
// The reference finds a triangle only through the leaves of ITS tree (KDTree.cpp:31-69): a ray tests a triangle when it
// passes the box of a leaf that holds it.  For almost every triangle this is unobservable -- a hit point lies inside the
// triangle's bounds, hence inside some leaf box that holds the triangle, so any tree over the same triangles selects the
// same closest hit.  Two kinds of triangle break that argument:
//   * DROPPED triangles.  buildTree returns nullptr below depth 100 (KDTree.cpp:101), silently losing the triangles of
//     that subtree there; they survive only in the other leaves they were copied to (or nowhere).
//   * SLIVERS.  For a (nearly) collinear triangle computeBarycentricCoordinates (Triangle.h:62-75) cancels to noise and
//     getIntersection accepts "phantom" points far outside the triangle -- wherever the ray happens to be while it is
//     inside one of the leaf boxes that hold the sliver.
// (DEAD triangles -- zero area, so that Triangle's constructor divides the normal by 0 and every comparison of
// getIntersection fails on NaN -- can never be hit; they are simply left out.)
// Both are reproduced exactly by giving the kernel what the reference has: the irregular triangle together with the
// boxes of the reference leaves that hold it ("exceptions", include/hrt.h hrt_tri_exception).  The kernel tests an
// exception triangle when AABB::intersects (exact form) passes for one of its boxes; all other triangles go into the
// SAH rope tree.  This needs the leaf boxes of the reference's builder, so its partition rule is restated here
// (KDTree.cpp:87-151: median of the triangle minima on axis depth % 3, straddlers copied to both sides, leaves of at
// most 40, stop when both sides get equally many, cut off below depth 100) -- only to learn those boxes; no reference
// tree is kept or walked.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/hrt.h"

namespace hrt_host {

struct RefTreeAnalysis {
    std::vector<uint8_t> irregular;             // per triangle: 1 = keep out of the SAH tree
    std::vector<hrt_tri_exception> exceptions;  // (triangle, reference leaf, box) entries of the live irregular triangles (include/hrt.h): one box per
                                                // leaf, more where an ancestor's box does not contain the leaf's (`group` = the leaf)
    uint32_t n_dropped = 0, n_slivers = 0, n_dead = 0, n_pairs = 0;  // statistics (n_pairs: (triangle, leaf) pairs)
    uint32_t ref_leaves = 0, ref_depth = 0;
};

// positions: 3*nv floats, UNscaled (the reference builds its tree on the unscaled vertices, KDTree.cpp:91,131);
// aabb_min/max: Mesh::computeAABB's box (the root box of the reference tree, Mesh.cpp:107-110).
RefTreeAnalysis analyse_reference_tree(const float *positions, uint32_t nv, const uint32_t *indices, uint32_t nt,
                                       const float aabb_min[3], const float aabb_max[3]);

}  // namespace hrt_host
