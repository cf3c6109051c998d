#include "ref_tree.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace hrt_host {
namespace {

struct Bounds { float lo[3], hi[3]; };

struct Analysis {
    const float *pos;
    const uint32_t *idx;
    std::vector<Bounds> tb;                  // Triangle::getAABB of every triangle (Triangle.h:128-141), unscaled
    std::vector<uint8_t> dropped;            // lost by some subtree below depth 100
    std::vector<std::vector<uint32_t>> leaf_tris;
    std::vector<Bounds> leaf_box;
    std::vector<std::vector<Bounds>> leaf_gate;  // per leaf: the boxes a ray must ALL pass to reach it (see gate_of)
    std::vector<Bounds> path;                    // the boxes from the root to the node in hand

    // KDTree::Node::intersect descends through `aabb.intersects(ray)` of EVERY node on the way (KDTree.cpp:32).  A child's box is
    // its parent's with one face moved to the cutting plane -- inside the parent when the plane is, and then whoever passes the
    // child passes the parent (AABB::intersects is monotone in the faces).  But cut() takes the median of UNCLIPPED triangle
    // bounds (:87-98): the plane can lie outside the node, one child then CONTAINS its parent and sticks out of it, and a ray
    // through the part that sticks out reaches the leaf's box without reaching the leaf.  So a leaf is reached through its own
    // box and every ancestor box that does not contain a box already kept (as AABB::intersects reads a box: faces in either order).
    static bool contains(const Bounds &outer, const Bounds &inner) {
        for (int a = 0; a < 3; ++a)
            if (!(std::min(outer.lo[a], outer.hi[a]) <= std::min(inner.lo[a], inner.hi[a]) && std::max(outer.lo[a], outer.hi[a]) >= std::max(inner.lo[a], inner.hi[a])))
                return false;
        return true;
    }
    std::vector<Bounds> gate_of() const {
        std::vector<Bounds> keep{path.back()};
        for (size_t k = path.size() - 1; k-- > 0;) {
            bool implied = false;
            for (const Bounds &b : keep) implied = implied || contains(path[k], b);
            if (!implied) keep.push_back(path[k]);
        }
        return keep;
    }
    uint32_t depth_reached = 0;
    // The reference's rule copies a straddling triangle to BOTH sides and only stops at 40 triangles, at equal halves or below
    // depth 100: on a soup of large overlapping triangles nothing ever separates and the recursion doubles per level -- the
    // reference itself never finishes on such a mesh (2^100 nodes).  The analysis follows it only as far as a finished
    // reference tree could plausibly go (a tree of depth ~25 that keeps ~4 copies of every triangle handles ~100 nt
    // references in all) and then says so instead of hanging.
    uint64_t work = 0, budget = 0;
    bool exhausted = false;
    uint32_t outside_splits = 0;

    // KDTree::buildTree, KDTree.cpp:100-151.  `tris` by value semantics of the reference's vectors; box = the node's AABB.
    void partition(const std::vector<uint32_t> &tris, const Bounds &box, unsigned depth) {
        if (tris.empty() || exhausted) return;
        work += tris.size();
        if (work > budget) { exhausted = true; return; }
        if (depth > 100u) {                  // KDTREE_MAX_DEPTH: the subtree is not built, its triangles are lost HERE
            for (uint32_t t : tris) dropped[t] = 1;
            return;
        }
        depth_reached = std::max(depth_reached, depth);
        struct PathGuard { std::vector<Bounds> &p; PathGuard(std::vector<Bounds> &p_, const Bounds &b) : p(p_) { p.push_back(b); } ~PathGuard() { p.pop_back(); } } on_path(path, box);
        auto leaf = [&]() { leaf_tris.push_back(tris); leaf_box.push_back(box); leaf_gate.push_back(gate_of()); };
        if (tris.size() <= 40u) { leaf(); return; }   // KDTREE_TRIANGLES_PER_LEAF
        // cut(), KDTree.cpp:87-98: median of the triangles' lower bounds on axis depth % 3, plus EPSILON (a double), as float
        const int axis = (int)(depth % 3u);
        std::vector<float> mins(tris.size());
        for (size_t i = 0; i < tris.size(); ++i) mins[i] = tb[tris[i]].lo[axis];
        std::sort(mins.begin(), mins.end());
        const float position = (float)((double)mins[mins.size() / 2] + HRT_EPSILON);
        if (!(position > box.lo[axis] && position < box.hi[axis])) ++outside_splits;  // (the cut uses UNCLIPPED triangle bounds)
        std::vector<uint32_t> left, right;
        for (uint32_t t : tris) {            // :129-140, comparisons in double as written
            if ((double)tb[t].hi[axis] <= (double)position - HRT_EPSILON) left.push_back(t);
            else if ((double)tb[t].lo[axis] >= (double)position + HRT_EPSILON) right.push_back(t);
            else { left.push_back(t); right.push_back(t); }
        }
        if (left.size() == right.size()) { leaf(); return; }   // :143-146
        Bounds lb = box, rb = box;           // AABB::split, AABB.h:67-75
        lb.hi[axis] = position;
        rb.lo[axis] = position;
        partition(left, lb, depth + 1u);
        partition(right, rb, depth + 1u);
    }
};

}  // namespace

RefTreeAnalysis analyse_reference_tree(const float *positions, uint32_t nv, const uint32_t *indices, uint32_t nt,
                                       const float aabb_min[3], const float aabb_max[3]) {
    (void)nv;
    RefTreeAnalysis out;
    out.irregular.assign(nt, 0);
    if (nt == 0) return out;
    Analysis A;
    A.pos = positions;
    A.idx = indices;
    A.tb.resize(nt);
    A.dropped.assign(nt, 0);
    for (uint32_t t = 0; t < nt; ++t) {
        Bounds &b = A.tb[t];
        for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
        for (int k = 0; k < 3; ++k) {
            const float *p = positions + 3 * (size_t)indices[3 * (size_t)t + k];
            for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], p[a]); b.hi[a] = std::max(b.hi[a], p[a]); }
        }
    }
    std::vector<uint32_t> all(nt);
    for (uint32_t t = 0; t < nt; ++t) all[t] = t;
    Bounds root;
    for (int a = 0; a < 3; ++a) { root.lo[a] = aabb_min[a]; root.hi[a] = aabb_max[a]; }
    A.budget = 400ull * nt + 10000000ull;
    A.partition(all, root, 0u);
    if (A.exhausted)
        throw std::runtime_error("mesh of " + std::to_string(nt) + " triangles: the reference's KD builder (KDTree.cpp:100-151: median cut, straddlers copied to both "
                                 "sides, leaves of 40, depth 100) does not terminate on it -- " + std::to_string(A.work) + " triangle references by depth " +
                                 std::to_string(A.depth_reached) + " and growing; its triangles overlap too much to be separated, and the reference cannot "
                                 "render this mesh either");
    if (std::getenv("HRT_REF_VERBOSE")) std::fprintf(stderr, "ref tree: %u triangles, %zu leaves, depth %u, %u splits outside their node\n", nt, A.leaf_box.size(), A.depth_reached, A.outside_splits);
    out.ref_leaves = (uint32_t)A.leaf_box.size();
    out.ref_depth = A.depth_reached;

    // Dead triangles: Triangle's constructor (Triangle.h:32-37) normalises cross(c1 - c0, c2 - c0) by its length; a zero
    // length makes the normal NaN and every test of getIntersection false -- such a triangle is never hit and is left out.
    // Slivers: the barycentric solve of Triangle.h:62-75 has condition ~ 1 / sin^2(angle at c0); in fp32 it returns noise
    // once sin^2 <~ 1e-6.  Everything below 1e-4 (an angle under 0.6 degrees) is treated as irregular -- two orders of
    // margin; a well-shaped triangle accepts points at most ~1e-6 edge lengths outside itself.
    // Both evaluated on the scaled vertices in fp32, like the constants the kernel folds (hrt_api.hip fold_triangle).
    std::vector<uint8_t> dead(nt, 0);
    for (uint32_t t = 0; t < nt; ++t) {
        float c[3][3];
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a) c[k][a] = positions[3 * (size_t)indices[3 * (size_t)t + k] + a] * HRT_TRIANGLE_SCALING;
        float e1[3], e2[3];
        for (int a = 0; a < 3; ++a) { e1[a] = c[1][a] - c[0][a]; e2[a] = c[2][a] - c[0][a]; }
        const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
        const float norm = (float)std::sqrt((double)(nx * nx + ny * ny + nz * nz));
        if (!(norm > 0.f)) {  // 0 or NaN: n / norm is NaN
            dead[t] = 1; out.irregular[t] = 1; ++out.n_dead;
            continue;
        }
        const float d00 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
        const float d01 = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
        const float d11 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
        const float denom = d00 * d11 - d01 * d01;
        const bool sliver = !(denom >= 1e-4f * (d00 * d11));
        if (sliver) { out.irregular[t] = 1; ++out.n_slivers; }
        if (A.dropped[t]) { out.irregular[t] = 1; ++out.n_dropped; }
    }
    // The reference leaves that hold each live irregular triangle.  One that no leaf holds is never hit: no entry.
    for (size_t l = 0; l < A.leaf_box.size(); ++l)
        for (uint32_t t : A.leaf_tris[l])
            if (out.irregular[t] && !dead[t]) {
                ++out.n_pairs;
                for (const Bounds &b : A.leaf_gate[l]) {  // the leaf's own box first, then the ancestors that stick in
                    hrt_tri_exception e;
                    e.triangle = t;
                    e.group = (uint32_t)l;
                    for (int a = 0; a < 3; ++a) { e.box_min[a] = b.lo[a]; e.box_max[a] = b.hi[a]; }
                    out.exceptions.push_back(e);
                }
            }
    return out;
}

}  // namespace hrt_host
