#include "kdtree.h"

#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <deque>

namespace hrt_host {
namespace {

struct Box {
    float lo[3], hi[3];
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

struct TriRef {
    uint32_t id;
    Box b;  // triangle bounds clipped to the cell that currently owns the ref
};

struct BuildNode {
    int axis = -1;  // -1: leaf
    float split = 0.f;
    int left = -1, right = -1;
    Box cell;
    std::vector<uint32_t> tris;  // leaf only
    int rope[6] = {-1, -1, -1, -1, -1, -1};
    uint32_t unit = 0;  // position in the flattened array
};

struct Builder {
    const KDBuildParams &prm;
    uint32_t max_depth;
    std::vector<BuildNode> nodes;
    uint32_t depth_reached = 0;

    explicit Builder(const KDBuildParams &p, uint32_t nt) : prm(p) {
        max_depth = p.max_depth ? p.max_depth : (uint32_t)(8.0 + 1.3 * std::log2((double)std::max(nt, 1u)));
    }

    int make_leaf(const Box &cell, const std::vector<TriRef> &refs) {
        BuildNode n;
        n.cell = cell;
        n.tris.reserve(refs.size());
        for (const TriRef &r : refs) n.tris.push_back(r.id);
        std::sort(n.tris.begin(), n.tris.end());
        nodes.push_back(std::move(n));
        return (int)nodes.size() - 1;
    }

    // Surface-area heuristic over all triangle-bound planes strictly inside the cell.  The three axes are
    // independent; for big nodes (the sequential top of a threaded build) they run on three threads and are
    // reduced in axis order with the same strict `<`, so the chosen plane is the sequential one.
    struct AxisBest { float cost; float pos; bool found; };
    AxisBest best_on_axis(const Box &cell, const std::vector<TriRef> &refs, int a, float leaf_cost) const {
        AxisBest r{leaf_cost, 0.f, false};
        if (!(cell.hi[a] > cell.lo[a])) return r;
        const size_t n = refs.size();
        const float inv_area = 1.f / std::max(cell.area(), 1e-30f);
        std::vector<float> mins(n), maxs(n);
        for (size_t i = 0; i < n; ++i) { mins[i] = refs[i].b.lo[a]; maxs[i] = refs[i].b.hi[a]; }
        std::sort(mins.begin(), mins.end());
        std::sort(maxs.begin(), maxs.end());
        auto consider = [&](float p) {
            if (!(p > cell.lo[a]) || !(p < cell.hi[a])) return;
            size_t nl = std::lower_bound(mins.begin(), mins.end(), p) - mins.begin();  // min < p
            size_t nr = n - (std::upper_bound(maxs.begin(), maxs.end(), p) - maxs.begin());  // max > p
            Box L = cell, R = cell;
            L.hi[a] = p;
            R.lo[a] = p;
            float c = prm.cost_traverse + prm.cost_intersect * inv_area * (L.area() * (float)nl + R.area() * (float)nr);
            if (nl == 0 || nr == 0) c *= prm.empty_bonus;
            if (c < r.cost) { r.cost = c; r.pos = p; r.found = true; }
        };
        float last = NAN;
        for (size_t i = 0; i < n; ++i) if (mins[i] != last) { last = mins[i]; consider(last); }
        last = NAN;
        for (size_t i = 0; i < n; ++i) if (maxs[i] != last) { last = maxs[i]; consider(last); }
        return r;
    }

    bool best_split(const Box &cell, const std::vector<TriRef> &refs, int &axis_out, float &pos_out) {
        const float leaf_cost = prm.cost_intersect * (float)refs.size();  // cost of not splitting
        AxisBest ab[3];
        if (tasks && refs.size() > 4096) {
            std::thread t1([&] { ab[1] = best_on_axis(cell, refs, 1, leaf_cost); });
            std::thread t2([&] { ab[2] = best_on_axis(cell, refs, 2, leaf_cost); });
            ab[0] = best_on_axis(cell, refs, 0, leaf_cost);
            t1.join();
            t2.join();
        } else {
            for (int a = 0; a < 3; ++a) ab[a] = best_on_axis(cell, refs, a, leaf_cost);
        }
        float best = leaf_cost;
        bool found = false;
        for (int a = 0; a < 3; ++a)
            if (ab[a].found && ab[a].cost < best) { best = ab[a].cost; axis_out = a; pos_out = ab[a].pos; found = true; }
        return found;
    }

    // Subtrees handed to worker threads: the top of the tree is built here, every subtree that has shrunk to
    // `spawn_below` references becomes a task with its own Builder, and the task's nodes are spliced in
    // afterwards.  Node numbers differ from a sequential build, the TREE does not (each subtree is a pure
    // function of its cell and references), and the flattening numbers units breadth-first from the root,
    // so the emitted arrays are identical for any thread count.
    struct Task {
        Box cell;
        std::vector<TriRef> refs;
        uint32_t depth = 0;
        std::vector<BuildNode> nodes;  // result
        int root = -1;
        uint32_t depth_reached = 0;
    };
    std::vector<Task> *tasks = nullptr;  // non-null while building the top with spawning enabled
    size_t spawn_below = 0;  // a subtree with at most this many references (and more than 256) becomes a task

    int build(const Box &cell, std::vector<TriRef> &refs, uint32_t depth) {
        depth_reached = std::max(depth_reached, depth);
        if (tasks && depth >= 1 && refs.size() > 256 && refs.size() <= spawn_below) {
            Task t;
            t.cell = cell;
            t.refs.swap(refs);
            t.depth = depth;
            tasks->push_back(std::move(t));
            return -2 - (int)(tasks->size() - 1);  // placeholder, patched by splice()
        }
        int axis = -1;
        float pos = 0.f;
        if (refs.size() <= prm.leaf_max || depth >= max_depth || !best_split(cell, refs, axis, pos))
            return make_leaf(cell, refs);
        std::vector<TriRef> lrefs, rrefs;
        for (const TriRef &r : refs) {
            const bool to_left = (r.b.lo[axis] < pos) || (r.b.hi[axis] <= pos);
            const bool to_right = (r.b.hi[axis] > pos) || (r.b.lo[axis] >= pos);
            if (to_left) { TriRef c = r; c.b.hi[axis] = std::min(c.b.hi[axis], pos); lrefs.push_back(c); }
            if (to_right) { TriRef c = r; c.b.lo[axis] = std::max(c.b.lo[axis], pos); rrefs.push_back(c); }
        }
        std::vector<TriRef>().swap(refs);
        Box lc = cell, rc = cell;
        lc.hi[axis] = pos;
        rc.lo[axis] = pos;
        int self = (int)nodes.size();
        nodes.emplace_back();
        nodes[self].axis = axis;
        nodes[self].split = pos;
        nodes[self].cell = cell;
        int l = build(lc, lrefs, depth + 1);
        int r = build(rc, rrefs, depth + 1);
        nodes[self].left = l;
        nodes[self].right = r;
        return self;
    }

    void run_tasks(std::vector<Task> &ts, unsigned threads) {
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i = next.fetch_add(1); i < ts.size(); i = next.fetch_add(1)) {
                Builder sub(prm, 1);
                sub.max_depth = max_depth;
                ts[i].root = sub.build(ts[i].cell, ts[i].refs, ts[i].depth);
                ts[i].nodes = std::move(sub.nodes);
                ts[i].depth_reached = sub.depth_reached;
            }
        };
        std::vector<std::thread> pool;
        const unsigned n = (unsigned)std::min<size_t>(threads, ts.size());
        for (unsigned k = 1; k < n; ++k) pool.emplace_back(worker);
        worker();
        for (std::thread &t : pool) t.join();
    }

    void splice(std::vector<Task> &ts) {
        std::vector<int> root_of(ts.size());
        for (size_t i = 0; i < ts.size(); ++i) {
            const int off = (int)nodes.size();
            for (BuildNode &n : ts[i].nodes) {
                if (n.axis >= 0) { n.left += off; n.right += off; }
                nodes.push_back(std::move(n));
            }
            root_of[i] = ts[i].root + off;
            depth_reached = std::max(depth_reached, ts[i].depth_reached);
        }
        for (BuildNode &n : nodes) {
            if (n.axis < 0) continue;
            if (n.left <= -2) n.left = root_of[(size_t)(-2 - n.left)];
            if (n.right <= -2) n.right = root_of[(size_t)(-2 - n.right)];
        }
    }

    // Push a rope down to the deepest node whose cell still covers the whole face `f` of `box`.
    int tighten(int r, int f, const Box &box) const {
        while (r >= 0 && nodes[r].axis >= 0) {
            const BuildNode &n = nodes[r];
            const int fa = f >> 1;
            if (n.axis == fa) r = (f & 1) ? n.left : n.right;
            else if (n.split <= box.lo[n.axis]) r = n.right;
            else if (n.split >= box.hi[n.axis]) r = n.left;
            else break;
        }
        return r;
    }

    void assign_ropes(int ni, const int in[6]) {
        BuildNode &n = nodes[ni];
        int rp[6];
        for (int f = 0; f < 6; ++f) rp[f] = tighten(in[f], f, n.cell);
        if (n.axis < 0) {
            std::memcpy(n.rope, rp, sizeof(rp));
            return;
        }
        int lr[6], rr[6];
        std::memcpy(lr, rp, sizeof(rp));
        std::memcpy(rr, rp, sizeof(rp));
        lr[2 * n.axis + 1] = n.right;
        rr[2 * n.axis] = n.left;
        const int l = n.left, r = n.right;  // n may dangle after recursion only if nodes grew; it does not here
        assign_ropes(l, lr);
        assign_ropes(r, rr);
    }
};

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// Ropes and the flattened unit array of a built tree (nodes in b.nodes, whoever built them).
FlatKDTree finish_flat_kdtree(Builder &b, int root_node, const Box &root, uint32_t nt, std::chrono::steady_clock::time_point t_start, unsigned threads = 0) {
    FlatKDTree out;
    const auto t_built = std::chrono::steady_clock::now();
    const int nil[6] = {-1, -1, -1, -1, -1, -1};
    b.assign_ropes(root_node, nil);
    const auto t_roped = std::chrono::steady_clock::now();

    // Numbering in 16-byte units (inner = 1 unit, leaf = 4 units), made for 64-byte cache lines: the walk's loads are
    // dependent (node -> child -> ... -> leaf header -> rope), and a hop that stays inside the line the previous hop just
    // brought in costs an L1 hit instead of an L2 / Infinity-Cache round trip.
    //   * clusters {node, its inner children} are kept inside one 64-byte line (padding where needed): going from a cluster
    //     root to either child is a same-line hop, so only every second level of a descent pays a far access;
    //   * leaves start on 64-byte boundaries: bounds, triangle range and all six ropes come with one line;
    //   * clusters are emitted breadth-first, so a prefix of the array is still the top of the tree (what the kernels stage
    //     into LDS).
    std::vector<int> order;
    order.reserve(b.nodes.size());
    std::deque<int> queue{root_node};  // cluster roots
    uint32_t next_unit = 0;
    auto place = [&](int ni, uint32_t units) { b.nodes[ni].unit = next_unit; next_unit += units; order.push_back(ni); };
    auto align4 = [&]() { next_unit = (next_unit + 3u) & ~3u; };
    while (!queue.empty()) {
        const int ri = queue.front();
        queue.pop_front();
        if (b.nodes[ri].axis < 0) { align4(); place(ri, 4u); continue; }  // a leaf root (a tree of one cell)
        const int kids[2] = {b.nodes[ri].left, b.nodes[ri].right};
        uint32_t inner_kids = 0;
        for (int k : kids) if (b.nodes[k].axis >= 0) ++inner_kids;
        if ((next_unit & 3u) + 1u + inner_kids > 4u) align4();        // the root and its inner children share a line
        place(ri, 1u);
        for (int k : kids) if (b.nodes[k].axis >= 0) place(k, 1u);
        for (int k : kids) if (b.nodes[k].axis < 0) { align4(); place(k, 4u); }
        for (int k : kids)
            if (b.nodes[k].axis >= 0) { queue.push_back(b.nodes[k].left); queue.push_back(b.nodes[k].right); }  // grandchildren: new clusters
    }
    auto ref_of = [&](int ni) -> uint32_t {
        if (ni < 0) return HRT_KD_NIL;
        const BuildNode &n = b.nodes[ni];
        return n.unit | (n.axis < 0 ? HRT_KD_LEAF : 0u);
    };
    out.units.assign(next_unit, hrt_kdunit{{0, 0, 0, 0}});
    for (int ni : order) {
        const BuildNode &n = b.nodes[ni];
        hrt_kdunit *u = &out.units[n.unit];
        if (n.axis >= 0) {
            u[0].w[0] = f2u(n.split);
            u[0].w[1] = (uint32_t)n.axis;
            u[0].w[2] = ref_of(n.left);
            u[0].w[3] = ref_of(n.right);
            ++out.n_inner;
        } else {
            u[0].w[0] = f2u(n.cell.lo[0]); u[0].w[1] = f2u(n.cell.lo[1]); u[0].w[2] = f2u(n.cell.lo[2]);
            u[0].w[3] = (uint32_t)out.leaf_tris.size();
            u[1].w[0] = f2u(n.cell.hi[0]); u[1].w[1] = f2u(n.cell.hi[1]); u[1].w[2] = f2u(n.cell.hi[2]);
            u[1].w[3] = (uint32_t)n.tris.size();
            for (int f = 0; f < 4; ++f) u[2].w[f] = ref_of(n.rope[f]);
            u[3].w[0] = ref_of(n.rope[4]);
            u[3].w[1] = ref_of(n.rope[5]);
            out.leaf_tris.insert(out.leaf_tris.end(), n.tris.begin(), n.tris.end());
            ++out.n_leaves;
            if (n.tris.empty()) ++out.n_empty_leaves;
        }
    }
    out.root = ref_of(root_node);
    for (int a = 0; a < 3; ++a) { out.root_lo[a] = root.lo[a]; out.root_hi[a] = root.hi[a]; }
    out.depth = b.depth_reached;
    if (std::getenv("HRT_KD_VERBOSE")) {
        auto ms = [](auto a, auto b2) { return std::chrono::duration<double, std::milli>(b2 - a).count(); };
        std::fprintf(stderr, "kd build: %u triangles, %u threads: tree %.1f ms, ropes %.1f ms, flatten %.1f ms\n", nt, threads,
                     ms(t_start, t_built), ms(t_built, t_roped), ms(t_roped, std::chrono::steady_clock::now()));
    }
    return out;
}

}  // namespace

FlatKDTree build_flat_kdtree(const float *positions, uint32_t nv, const uint32_t *indices,
                             uint32_t nt, const KDBuildParams &params, const uint8_t *skip) {
    FlatKDTree out;
    (void)nv;
    std::vector<TriRef> refs;
    refs.reserve(nt);
    Box root;
    for (int a = 0; a < 3; ++a) { root.lo[a] = INFINITY; root.hi[a] = -INFINITY; }
    for (uint32_t t = 0; t < nt; ++t) {
        if (skip && skip[t]) continue;
        refs.emplace_back();
        TriRef &r = refs.back();
        r.id = t;
        for (int a = 0; a < 3; ++a) { r.b.lo[a] = INFINITY; r.b.hi[a] = -INFINITY; }
        for (int k = 0; k < 3; ++k) {
            const float *p = positions + 3 * (size_t)indices[3 * (size_t)t + k];
            for (int a = 0; a < 3; ++a) {
                r.b.lo[a] = std::min(r.b.lo[a], p[a]);
                r.b.hi[a] = std::max(r.b.hi[a], p[a]);
            }
        }
        for (int a = 0; a < 3; ++a) {
            root.lo[a] = std::min(root.lo[a], r.b.lo[a]);
            root.hi[a] = std::max(root.hi[a], r.b.hi[a]);
        }
    }
    if (refs.empty()) return out;  // no (regular) triangle: no tree
    nt = (uint32_t)refs.size();
    // Pad the root cell so that hits on the hull are strictly inside it.
    for (int a = 0; a < 3; ++a) {
        float pad = 1e-4f * std::max(1.f, std::max(std::fabs(root.lo[a]), std::fabs(root.hi[a])));
        root.lo[a] -= pad;
        root.hi[a] += pad;
    }

    const auto t_start = std::chrono::steady_clock::now();
    KDBuildParams tuned = params;  // (A/B sweeps of the heuristic's constants on the GPU box: tools/time_only.py)
    if (const char *e = std::getenv("HRT_KD_CT")) tuned.cost_traverse = (float)atof(e);
    if (const char *e = std::getenv("HRT_KD_CI")) tuned.cost_intersect = (float)atof(e);
    if (const char *e = std::getenv("HRT_KD_EB")) tuned.empty_bonus = (float)atof(e);
    Builder b(tuned, nt);
    if (params.builder) {
        // The replaceable step (include/hrt.h hrt_kd_builder_fn): the builder returns the nodes, ropes and flattening follow below.
        std::vector<uint32_t> ids(nt);
        std::vector<float> lo(3 * (size_t)nt), hi(3 * (size_t)nt);
        for (uint32_t i = 0; i < nt; ++i) {
            ids[i] = refs[i].id;
            for (int a = 0; a < 3; ++a) { lo[3 * (size_t)i + a] = refs[i].b.lo[a]; hi[3 * (size_t)i + a] = refs[i].b.hi[a]; }
        }
        hrt_kd_build_input in;
        in.n_refs = nt; in.ids = ids.data(); in.lo = lo.data(); in.hi = hi.data();
        for (int a = 0; a < 3; ++a) { in.cell_lo[a] = root.lo[a]; in.cell_hi[a] = root.hi[a]; }
        in.leaf_max = tuned.leaf_max; in.max_depth = b.max_depth;
        in.cost_traverse = tuned.cost_traverse; in.cost_intersect = tuned.cost_intersect; in.empty_bonus = tuned.empty_bonus;
        hrt_kd_build_output res;
        std::memset(&res, 0, sizeof(res));
        const int rc = params.builder(&in, &res, params.builder_user);
        bool ok = rc == 0 && res.nodes && res.n_nodes > 0 && res.root >= 0 && (uint32_t)res.root < res.n_nodes;
        for (uint32_t i = 0; ok && i < res.n_nodes; ++i) {
            const hrt_kd_build_node &g = res.nodes[i];
            if (g.axis >= 0) ok = g.axis <= 2 && g.left >= 0 && g.right >= 0 && (uint32_t)g.left < res.n_nodes && (uint32_t)g.right < res.n_nodes;
            else ok = (uint64_t)g.first_tri + g.n_tris <= res.n_tris && (g.n_tris == 0 || res.tris);
        }
        if (ok) {
            b.nodes.resize(res.n_nodes);
            for (uint32_t i = 0; i < res.n_nodes; ++i) {
                const hrt_kd_build_node &g = res.nodes[i];
                BuildNode &n = b.nodes[i];
                n.axis = g.axis; n.split = g.split; n.left = g.left; n.right = g.right;
                for (int a = 0; a < 3; ++a) { n.cell.lo[a] = g.lo[a]; n.cell.hi[a] = g.hi[a]; }
                if (g.axis < 0) { n.tris.assign(res.tris + g.first_tri, res.tris + g.first_tri + g.n_tris); std::sort(n.tris.begin(), n.tris.end()); }
            }
            b.depth_reached = res.depth;
        }
        const int root_from_builder = res.root;
        std::free(res.nodes);
        std::free(res.tris);
        if (!ok) throw std::runtime_error("the KD builder set with hrt_host_scene_set_kd_builder failed or returned a malformed tree");
        return finish_flat_kdtree(b, root_from_builder, root, nt, t_start);
    }
    unsigned threads = params.threads ? params.threads : std::thread::hardware_concurrency();
    if (const char *e = std::getenv("HRT_KD_THREADS")) threads = (unsigned)std::max(1, atoi(e));
    threads = std::min(threads, 16u);  // measured on the MI355X host (pool flamingo, 31 575 triangles): 1 -> 273 ms, 4 -> 177, 16 -> 112, 64 -> 173
    int root_node;
    if (threads > 1 && nt > 4096) {
        std::vector<Builder::Task> tasks;
        b.tasks = &tasks;
        b.spawn_below = std::max<size_t>(512, nt / (4u * threads));  // SAH first cuts empty space: split by size, not by depth
        root_node = b.build(root, refs, 0);  // the root itself (depth 0) is never a task
        b.tasks = nullptr;
        const auto t_top = std::chrono::steady_clock::now();
        b.run_tasks(tasks, threads);
        b.splice(tasks);
        if (std::getenv("HRT_KD_VERBOSE"))
            std::fprintf(stderr, "kd build: top %.1f ms, %zu subtree tasks\n",
                         std::chrono::duration<double, std::milli>(t_top - t_start).count(), tasks.size());
    } else {
        root_node = b.build(root, refs, 0);
    }
    return finish_flat_kdtree(b, root_node, root, nt, t_start, threads);
}

}  // namespace hrt_host
