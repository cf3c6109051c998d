// hrt_kd_build_gpu -- the split search and partition of the KD build on the device (include/hrt.h hrt_kd_builder_fn; SURVEY 8
// f-2, replacing KDTree::buildTree, KDTree.cpp:87-151).  Included by hrt_api.hip.
//
// The host builder (host/kdtree.cpp) is a recursion: per node, the surface-area heuristic over EVERY triangle-bound plane
// strictly inside the cell on three axes (two sorted lists and binary searches per axis), then the references are split, clipped
// to the children's cells, and the children built.  Here the tree grows LEVEL BY LEVEL and every open node of a level is
// searched at once:
//   kd_split_search   one thread per candidate plane (a reference's lower or upper bound on one axis of one node); it counts the
//                     references of its node that start below / end above the plane -- the whole node streams through LDS in
//                     tiles of 256 -- and prices the plane with the host's arithmetic (fp32, same order, no contraction); a
//                     workgroup reduces its 256 candidates to the best by (cost, lower-bound list before upper-bound list,
//                     position), which is the order in which the host's loop meets them with its strict `<`
//   kd_count          how many references each child of a split node receives (both, when the triangle straddles the plane)
//   kd_partition      the references move to their children, bounds clipped to the plane (kdtree.cpp build())
// The host side of a level is bookkeeping: which nodes are leaves (few references, the depth limit, no plane cheaper than not
// splitting), the reduction of the workgroups' bests per node and axis in axis order, the children's ranges.  Work is
// exhaustive, references x candidates per node -- 6e9 pairs for the root of the 31 575-triangle flamingo, well under a
// millisecond of the chip -- so no sort is needed and nothing depends on an order of the references; it would not scale to
// millions of triangles, where the host builder remains the one to use.
// Same nodes as the host builder: tests/test_gpu_kdbuild.py compares the flattened arrays of every mesh of the configurations.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace hrtkd {

struct DNode {            // an open node of the current level
    uint32_t begin, count;
    float lo[3], hi[3];
    int32_t axis;         // set by the host after the search: the split (or -1)
    float pos;
    uint32_t child_begin[2];  // where the children's references start in the next level's arrays
};
struct Work { uint32_t node, axis, chunk; };
struct Best { float cost; uint32_t list; float pos; uint32_t valid; };

__device__ __forceinline__ float box_area(const float lo[3], const float hi[3]) {  // host/kdtree.cpp Box::area
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}
__device__ __forceinline__ bool better(const Best &a, const Best &b) {  // a comes before b in the host's loop and is at least as cheap
    if (!a.valid) return false;
    if (!b.valid) return true;
    if (a.cost < b.cost) return true;
    if (b.cost < a.cost) return false;
    if (a.list != b.list) return a.list < b.list;
    return a.pos < b.pos;
}

// lo / hi: [3][cap] (one array per axis).  One workgroup = 256 candidates of one (node, axis).
extern "C" __global__ void __launch_bounds__(256) kd_split_search(const float *__restrict__ lo, const float *__restrict__ hi, uint32_t cap,
                                                                  const DNode *__restrict__ nodes, const Work *__restrict__ work, Best *__restrict__ best,
                                                                  float ct, float ci, float eb) {
    __shared__ float s_lo[256], s_hi[256];
    __shared__ Best s_best[256];
    const Work w = work[blockIdx.x];
    const DNode nd = nodes[w.node];
    const uint32_t n = nd.count, a = w.axis, tid = threadIdx.x;
    const float *alo = lo + (size_t)a * cap + nd.begin, *ahi = hi + (size_t)a * cap + nd.begin;
    const uint32_t c = w.chunk * 256u + tid;
    const bool exists = c < 2u * n;
    const uint32_t list = c >= n ? 1u : 0u;
    const float p = exists ? (list ? ahi[c - n] : alo[c]) : 0.f;
    const bool inside = exists && p > nd.lo[a] && p < nd.hi[a];  // strictly inside the cell (best_on_axis `consider`)
    uint32_t nl = 0, nr = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += 256u) {
        const uint32_t m = min(256u, n - t0);
        __syncthreads();
        if (tid < m) { s_lo[tid] = alo[t0 + tid]; s_hi[tid] = ahi[t0 + tid]; }
        __syncthreads();
        for (uint32_t j = 0; j < m; ++j) {   // wave-uniform LDS reads (broadcast)
            nl += s_lo[j] < p ? 1u : 0u;     // min < p   (lower_bound)
            nr += s_hi[j] > p ? 1u : 0u;     // max > p   (upper_bound)
        }
    }
    Best b;
    b.valid = 0u; b.cost = 0.f; b.list = list; b.pos = p;
    if (inside) {
        const float inv_area = 1.f / fmaxf(box_area(nd.lo, nd.hi), 1e-30f);
        float llo[3] = {nd.lo[0], nd.lo[1], nd.lo[2]}, lhi[3] = {nd.hi[0], nd.hi[1], nd.hi[2]};
        float rlo[3] = {nd.lo[0], nd.lo[1], nd.lo[2]}, rhi[3] = {nd.hi[0], nd.hi[1], nd.hi[2]};
        lhi[a] = p;
        rlo[a] = p;
        float cst = ct + ci * inv_area * (box_area(llo, lhi) * (float)nl + box_area(rlo, rhi) * (float)nr);
        if (nl == 0u || nr == 0u) cst *= eb;
        b.cost = cst;
        b.valid = cst == cst ? 1u : 0u;  // (a NaN cost is never `<` anything in the host's loop)
    }
    s_best[tid] = b;
    __syncthreads();
    for (uint32_t s = 128u; s > 0u; s >>= 1) {
        if (tid < s && better(s_best[tid + s], s_best[tid])) s_best[tid] = s_best[tid + s];
        __syncthreads();
    }
    if (tid == 0) best[blockIdx.x] = s_best[0];
}

// One workgroup = 256 references of one SPLIT node (Work::axis unused).  counts[2 * node + side]
extern "C" __global__ void __launch_bounds__(256) kd_count(const float *__restrict__ lo, const float *__restrict__ hi, uint32_t cap,
                                                           const DNode *__restrict__ nodes, const Work *__restrict__ work, uint32_t *__restrict__ counts) {
    const Work w = work[blockIdx.x];
    const DNode nd = nodes[w.node];
    const uint32_t i = w.chunk * 256u + threadIdx.x;
    if (i >= nd.count) return;
    const uint32_t a = (uint32_t)nd.axis;
    const float l = lo[(size_t)a * cap + nd.begin + i], h = hi[(size_t)a * cap + nd.begin + i], pos = nd.pos;
    const bool to_left = (l < pos) || (h <= pos), to_right = (h > pos) || (l >= pos);  // kdtree.cpp build()
    if (to_left) atomicAdd(&counts[2u * w.node], 1u);
    if (to_right) atomicAdd(&counts[2u * w.node + 1u], 1u);
}

// The references of the split nodes move to the next level's arrays (cursors: one per child, starting at child_begin).
extern "C" __global__ void __launch_bounds__(256) kd_partition(const uint32_t *__restrict__ ids, const float *__restrict__ lo, const float *__restrict__ hi, uint32_t cap,
                                                               const DNode *__restrict__ nodes, const Work *__restrict__ work, uint32_t *__restrict__ cursors,
                                                               uint32_t *__restrict__ ids2, float *__restrict__ lo2, float *__restrict__ hi2, uint32_t cap2) {
    const Work w = work[blockIdx.x];
    const DNode nd = nodes[w.node];
    const uint32_t i = w.chunk * 256u + threadIdx.x;
    if (i >= nd.count) return;
    const uint32_t a = (uint32_t)nd.axis, src = nd.begin + i;
    float l[3], h[3];
    for (uint32_t k = 0; k < 3u; ++k) { l[k] = lo[(size_t)k * cap + src]; h[k] = hi[(size_t)k * cap + src]; }
    const float pos = nd.pos;
    const bool to_left = (l[a] < pos) || (h[a] <= pos), to_right = (h[a] > pos) || (l[a] >= pos);
    const uint32_t id = ids[src];
    if (to_left) {
        const uint32_t d = nd.child_begin[0] + atomicAdd(&cursors[2u * w.node], 1u);
        ids2[d] = id;
        for (uint32_t k = 0; k < 3u; ++k) { lo2[(size_t)k * cap2 + d] = l[k]; hi2[(size_t)k * cap2 + d] = k == a ? fminf(h[k], pos) : h[k]; }
    }
    if (to_right) {
        const uint32_t d = nd.child_begin[1] + atomicAdd(&cursors[2u * w.node + 1u], 1u);
        ids2[d] = id;
        for (uint32_t k = 0; k < 3u; ++k) { lo2[(size_t)k * cap2 + d] = k == a ? fmaxf(l[k], pos) : l[k]; hi2[(size_t)k * cap2 + d] = h[k]; }
    }
}

template <class T>
struct DevBuf {   // a device array that only grows
    T *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace hrtkd

// The level loop.  (g_rt / fail / HIP_TRY come from hrt_api.hip, which includes this file inside its anonymous-namespace scope's
// translation unit; the function itself is exported through include/hrt.h.)
static int kd_build_gpu_impl(const hrt_kd_build_input *in, hrt_kd_build_output *out) {
    using namespace hrtkd;
    struct HNode { uint32_t begin, count, depth; float lo[3], hi[3]; int32_t index; };  // an open node and its place in `nodes`
    std::vector<hrt_kd_build_node> nodes;
    std::vector<uint32_t> tris;
    uint32_t depth_reached = 0;
    const uint32_t n0 = in->n_refs;

    // level arrays on the device, ping-pong
    DevBuf<uint32_t> ids[2], counts, cursors;
    DevBuf<float> lo[2], hi[2];
    DevBuf<DNode> d_nodes;
    DevBuf<Work> d_work;
    DevBuf<Best> d_best;
    size_t cap[2] = {std::max<size_t>(n0, 1), 0};
    HIP_TRY(ids[0].need(cap[0]));
    HIP_TRY(lo[0].need(3 * cap[0]));
    HIP_TRY(hi[0].need(3 * cap[0]));
    {
        std::vector<float> soa(3 * cap[0]);
        HIP_TRY(hipMemcpy(ids[0].p, in->ids, n0 * sizeof(uint32_t), hipMemcpyHostToDevice));
        for (int pass = 0; pass < 2; ++pass) {
            const float *src = pass ? in->hi : in->lo;
            for (uint32_t i = 0; i < n0; ++i)
                for (int a = 0; a < 3; ++a) soa[(size_t)a * cap[0] + i] = src[3 * (size_t)i + a];
            HIP_TRY(hipMemcpy(pass ? hi[0].p : lo[0].p, soa.data(), 3 * cap[0] * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    std::vector<HNode> level(1);
    level[0].begin = 0; level[0].count = n0; level[0].depth = 0; level[0].index = 0;
    for (int a = 0; a < 3; ++a) { level[0].lo[a] = in->cell_lo[a]; level[0].hi[a] = in->cell_hi[a]; }
    nodes.emplace_back();
    int cur = 0;
    std::vector<uint32_t> h_ids;
    while (!level.empty()) {
        const size_t total = level.back().begin + level.back().count;  // references of this level (ranges are consecutive)
        h_ids.resize(total);
        if (total) HIP_TRY(hipMemcpy(h_ids.data(), ids[cur].p, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
        // ---- which nodes are searched
        std::vector<DNode> dn(level.size());
        std::vector<Work> work;
        std::vector<uint32_t> first_work(level.size() + 1, 0);
        for (size_t k = 0; k < level.size(); ++k) {
            const HNode &h = level[k];
            depth_reached = std::max(depth_reached, h.depth);
            DNode &d = dn[k];
            d.begin = h.begin; d.count = h.count; d.axis = -1; d.pos = 0.f; d.child_begin[0] = d.child_begin[1] = 0;
            for (int a = 0; a < 3; ++a) { d.lo[a] = h.lo[a]; d.hi[a] = h.hi[a]; }
            first_work[k] = (uint32_t)work.size();
            if (h.count > in->leaf_max && h.depth < in->max_depth)
                for (uint32_t a = 0; a < 3; ++a)
                    for (uint32_t c = 0; c < (2u * h.count + 255u) / 256u; ++c) work.push_back(Work{(uint32_t)k, a, c});
        }
        first_work[level.size()] = (uint32_t)work.size();
        std::vector<Best> best(work.size());
        HIP_TRY(d_nodes.need(dn.size()));
        HIP_TRY(hipMemcpy(d_nodes.p, dn.data(), dn.size() * sizeof(DNode), hipMemcpyHostToDevice));
        if (!work.empty()) {
            HIP_TRY(d_work.need(work.size()));
            HIP_TRY(d_best.need(work.size()));
            HIP_TRY(hipMemcpy(d_work.p, work.data(), work.size() * sizeof(Work), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(kd_split_search, dim3((uint32_t)work.size()), dim3(256), 0, 0, lo[cur].p, hi[cur].p, (uint32_t)cap[cur], d_nodes.p, d_work.p,
                               d_best.p, in->cost_traverse, in->cost_intersect, in->empty_bonus);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(best.data(), d_best.p, work.size() * sizeof(Best), hipMemcpyDeviceToHost));
        }
        // ---- the best plane of every searched node: axes in order, strict `<` (kdtree.cpp best_split), against the cost of a leaf
        std::vector<Work> split_work;
        for (size_t k = 0; k < level.size(); ++k) {
            const HNode &h = level[k];
            float bc = in->cost_intersect * (float)h.count;
            int axis = -1;
            float pos = 0.f;
            uint32_t w = first_work[k];
            for (uint32_t a = 0; a < 3 && w < first_work[k + 1]; ++a) {
                Best ab;
                ab.valid = 0; ab.cost = 0.f; ab.list = 0; ab.pos = 0.f;
                for (; w < first_work[k + 1] && work[w].axis == a; ++w) {
                    const Best &b = best[w];
                    const bool wins = b.valid && (!ab.valid || b.cost < ab.cost || (b.cost == ab.cost && (b.list < ab.list || (b.list == ab.list && b.pos < ab.pos))));
                    if (wins) ab = b;
                }
                if (ab.valid && ab.cost < bc) { bc = ab.cost; axis = (int)a; pos = ab.pos; }
            }
            hrt_kd_build_node &n = nodes[(size_t)h.index];
            n.axis = axis; n.split = pos; n.left = n.right = -1; n.first_tri = 0; n.n_tris = 0;
            for (int a = 0; a < 3; ++a) { n.lo[a] = h.lo[a]; n.hi[a] = h.hi[a]; }
            dn[k].axis = axis; dn[k].pos = pos;
            if (axis < 0) {  // a leaf: its triangle ids, ascending
                n.first_tri = (uint32_t)tris.size(); n.n_tris = h.count;
                tris.insert(tris.end(), h_ids.begin() + h.begin, h_ids.begin() + h.begin + h.count);
                std::sort(tris.end() - h.count, tris.end());
            } else {
                for (uint32_t c = 0; c < (h.count + 255u) / 256u; ++c) split_work.push_back(Work{(uint32_t)k, 0u, c});
            }
        }
        if (split_work.empty()) break;
        // ---- children: sizes, ranges, cells
        HIP_TRY(hipMemcpy(d_nodes.p, dn.data(), dn.size() * sizeof(DNode), hipMemcpyHostToDevice));
        HIP_TRY(d_work.need(split_work.size()));
        HIP_TRY(hipMemcpy(d_work.p, split_work.data(), split_work.size() * sizeof(Work), hipMemcpyHostToDevice));
        HIP_TRY(counts.need(2 * level.size()));
        HIP_TRY(cursors.need(2 * level.size()));
        HIP_TRY(hipMemset(counts.p, 0, 2 * level.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemset(cursors.p, 0, 2 * level.size() * sizeof(uint32_t)));
        hipLaunchKernelGGL(kd_count, dim3((uint32_t)split_work.size()), dim3(256), 0, 0, lo[cur].p, hi[cur].p, (uint32_t)cap[cur], d_nodes.p, d_work.p, counts.p);
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> h_counts(2 * level.size());
        HIP_TRY(hipMemcpy(h_counts.data(), counts.p, h_counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        std::vector<HNode> next;
        uint64_t run = 0;
        for (size_t k = 0; k < level.size(); ++k) {
            if (dn[k].axis < 0) continue;
            const HNode &h = level[k];
            for (int side = 0; side < 2; ++side) {
                HNode c;
                c.begin = (uint32_t)run; c.count = h_counts[2 * k + side]; c.depth = h.depth + 1;
                for (int a = 0; a < 3; ++a) { c.lo[a] = h.lo[a]; c.hi[a] = h.hi[a]; }
                if (side == 0) c.hi[dn[k].axis] = dn[k].pos; else c.lo[dn[k].axis] = dn[k].pos;
                c.index = (int32_t)nodes.size();
                nodes.emplace_back();
                if (side == 0) nodes[(size_t)h.index].left = c.index; else nodes[(size_t)h.index].right = c.index;
                dn[k].child_begin[side] = c.begin;
                run += c.count;
                next.push_back(c);
            }
        }
        if (run > 0x7fffffffull) return fail(HRT_ERR_INVALID, "hrt_kd_build_gpu: more than 2^31 triangle references in one level");
        const int nxt = cur ^ 1;
        cap[nxt] = std::max<size_t>((size_t)run, 1);
        HIP_TRY(ids[nxt].need(cap[nxt]));
        HIP_TRY(lo[nxt].need(3 * cap[nxt]));
        HIP_TRY(hi[nxt].need(3 * cap[nxt]));
        cap[nxt] = std::min(ids[nxt].cap, std::min(lo[nxt].cap, hi[nxt].cap) / 3);  // (the arrays only grow: the axis stride is the capacity)
        HIP_TRY(hipMemcpy(d_nodes.p, dn.data(), dn.size() * sizeof(DNode), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(kd_partition, dim3((uint32_t)split_work.size()), dim3(256), 0, 0, ids[cur].p, lo[cur].p, hi[cur].p, (uint32_t)cap[cur], d_nodes.p, d_work.p,
                           cursors.p, ids[nxt].p, lo[nxt].p, hi[nxt].p, (uint32_t)cap[nxt]);
        HIP_TRY(hipGetLastError());
        level.swap(next);
        cur = nxt;
    }
    HIP_TRY(hipDeviceSynchronize());
    out->n_nodes = (uint32_t)nodes.size();
    out->n_tris = (uint32_t)tris.size();
    out->nodes = (hrt_kd_build_node *)std::malloc(std::max<size_t>(nodes.size(), 1) * sizeof(hrt_kd_build_node));
    out->tris = (uint32_t *)std::malloc(std::max<size_t>(tris.size(), 1) * sizeof(uint32_t));
    if (!out->nodes || !out->tris) { std::free(out->nodes); std::free(out->tris); out->nodes = nullptr; out->tris = nullptr; return fail(HRT_ERR_STATE, "hrt_kd_build_gpu: out of memory"); }
    std::memcpy(out->nodes, nodes.data(), nodes.size() * sizeof(hrt_kd_build_node));
    if (!tris.empty()) std::memcpy(out->tris, tris.data(), tris.size() * sizeof(uint32_t));
    out->root = 0;
    out->depth = depth_reached;
    return HRT_OK;
}
