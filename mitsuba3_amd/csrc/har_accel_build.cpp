/*
 * har_accel_build.cpp -- host-side builder of the compressed 8-wide BVH.
 *
 * Replaces the accel builders the reference delegates to third-party code
 * (Embree rtcCommitScene, src/render/scene_embree.inl:198-292; OptiX GAS/IAS,
 * src/render/scene_optix.inl:484-629; build_metal_accel, src/render/metal/accel.h:27-33)
 * and consumes the same lowering as SceneIRBuilder::build
 * (src/render/scene_ir.cpp:12-92): one BLAS for the top-level triangle geometry,
 * one BLAS per ShapeGroup, and a flattened TLAS with one identity entry for the
 * top-level BLAS plus one transformed entry per Instance.
 *
 * Pipeline: binned-SAH binary BVH (<= 3 primitives per leaf) -> greedy
 * surface-area collapse to 8 children -> octant-ordered slot assignment ->
 * 8-bit quantisation of the child boxes against the (padded) parent box.
 */
#include "har_accel_build.h"
#include "har_cpu.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <limits>
#include <string>
#include <thread>

namespace har {

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; } }
    void grow(const Box &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float area() const { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return 2.f * (dx * dy + dy * dz + dz * dx); }
};

struct BNode { Box box; int left = -1, right = -1; uint32_t first = 0, count = 0; };

struct Builder {
    const std::vector<PrimBox> &prims;
    std::vector<uint32_t> order;
    std::vector<BNode> bn;

    uint32_t max_leaf = 3;
    explicit Builder(const std::vector<PrimBox> &p) : prims(p), order(p.size()) { for (uint32_t i = 0; i < p.size(); ++i) order[i] = i; }

    Box prim_box(uint32_t i) const { Box b; std::memcpy(b.lo, prims[i].lo, 12); std::memcpy(b.hi, prims[i].hi, 12); return b; }
    float centroid(uint32_t i, int a) const { return 0.5f * (prims[i].lo[a] + prims[i].hi[a]); }

    /* binned-SAH binary build of order[begin, end) appended to `bn` in pre-order (a node precedes its subtrees, the left subtree the right one).
     * Large ranges fork: the two children are built concurrently into private vectors and spliced in afterwards -- same splits, same node
     * order as the sequential recursion, so the tree does not depend on the thread count (1M triangles on the 256-thread host of the GPU box: 0.81 -> 0.18 s). */
    /* one primitive per leaf (the only leaf size in use): a subtree over c primitives has exactly 2c - 1 nodes, so in pre-order its root sits at a known index, its left
     * subtree right behind it and its right subtree 2 * c_left entries further -- the array is sized once and concurrent subtree builds write disjoint ranges of it
     * (no private vectors, no splicing: that copied every node once per fork level) */
    int build(uint32_t begin, uint32_t end) {
        if (max_leaf == 1 && end > begin) { bn.assign(2 * (size_t) (end - begin) - 1, BNode()); return build_into(bn, begin, end, 0, 0); }
        return build_into(bn, begin, end, 0);
    }
    int build_into(std::vector<BNode> &bn, uint32_t begin, uint32_t end, int depth, int at = -1) {
        int idx = at;
        if (at < 0) { idx = (int) bn.size(); bn.emplace_back(); }
        /* bounds, then one binning sweep for the three axes; nodes with many primitives split both sweeps over threads (per-thread partial
         * boxes / bins, merged in order: min / max and integer counts are exact, so the result is the sequential one) */
        constexpr int NB = 16;
        const uint32_t count = end - begin;
        static const uint32_t sweep_par_min = getenv("HAR_BUILD_SWEEP_MIN") ? (uint32_t) atol(getenv("HAR_BUILD_SWEEP_MIN")) : 131072u;
        static const uint32_t hw = har_usable_cores();      /* affinity mask and container quota, looked up once */
        const uint32_t chunks = count >= sweep_par_min ? std::min<uint32_t>(std::min(hw, 32u), count / 32768u) : 1u;
        auto for_chunks = [&](auto &&body) {
            if (chunks <= 1) { body(0u, begin, end); return; }
            std::vector<std::thread> pool;
            for (uint32_t c = 1; c < chunks; ++c) {
                const uint32_t b0 = begin + (uint32_t) ((uint64_t) count * c / chunks), e0 = begin + (uint32_t) ((uint64_t) count * (c + 1) / chunks);
                try { pool.emplace_back([&body, c, b0, e0] { body(c, b0, e0); }); } catch (...) { body(c, b0, e0); }
            }
            body(0u, begin, begin + (uint32_t) ((uint64_t) count / chunks));
            for (auto &t : pool) t.join();
        };
        Box box; box.reset(); Box cb; cb.reset();
        {
            Box pb[32], pc[32];
            for_chunks([&](uint32_t c, uint32_t b0, uint32_t e0) {
                Box bb; bb.reset(); Box cc; cc.reset();
                for (uint32_t i = b0; i < e0; ++i) {
                    bb.grow(prim_box(order[i]));
                    for (int a = 0; a < 3; ++a) { float ce = centroid(order[i], a); cc.lo[a] = std::min(cc.lo[a], ce); cc.hi[a] = std::max(cc.hi[a], ce); }
                }
                pb[c] = bb; pc[c] = cc;
            });
            for (uint32_t c = 0; c < chunks; ++c) { box.grow(pb[c]); cb.grow(pc[c]); }
        }
        bn[idx].box = box;
        if (count <= max_leaf) { bn[idx].first = begin; bn[idx].count = count; return idx; }
        float best = std::numeric_limits<float>::infinity(); int best_axis = -1, best_split = -1;
        struct Bins { Box box[3][NB]; uint32_t cnt[3][NB]; };
        float ext3[3], scale3[3];
        for (int axis = 0; axis < 3; ++axis) { ext3[axis] = cb.hi[axis] - cb.lo[axis]; scale3[axis] = ext3[axis] > 0.f ? NB / ext3[axis] : 0.f; }
        Bins single; std::vector<Bins> multi; Bins *part = &single;            /* no heap traffic for the millions of small nodes */
        if (chunks > 1) { multi.resize(chunks); part = multi.data(); }
        for_chunks([&](uint32_t c, uint32_t b0, uint32_t e0) {
            Bins &B = part[c];
            for (int axis = 0; axis < 3; ++axis) for (int b = 0; b < NB; ++b) { B.box[axis][b].reset(); B.cnt[axis][b] = 0; }
            for (uint32_t i = b0; i < e0; ++i) {
                const Box pbx = prim_box(order[i]);
                for (int axis = 0; axis < 3; ++axis) {
                    if (!(ext3[axis] > 0.f)) continue;
                    int b = std::min(NB - 1, std::max(0, (int) ((centroid(order[i], axis) - cb.lo[axis]) * scale3[axis])));
                    B.box[axis][b].grow(pbx); B.cnt[axis][b]++;
                }
            }
        });
        for (int axis = 0; axis < 3; ++axis) {
            if (!(ext3[axis] > 0.f)) continue;
            Box bins[NB]; uint32_t cnt[NB];
            for (int b = 0; b < NB; ++b) { bins[b].reset(); cnt[b] = 0; for (uint32_t c = 0; c < chunks; ++c) { bins[b].grow(part[c].box[axis][b]); cnt[b] += part[c].cnt[axis][b]; } }
            Box rb[NB]; uint32_t rc[NB]; Box acc; acc.reset(); uint32_t c = 0;
            for (int b = NB - 1; b >= 0; --b) { acc.grow(bins[b]); c += cnt[b]; rb[b] = acc; rc[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < NB - 1; ++b) {
                acc.grow(bins[b]); c += cnt[b];
                if (c == 0 || rc[b + 1] == 0) continue;
                float cost = c * acc.area() + rc[b + 1] * rb[b + 1].area();
                if (cost < best) { best = cost; best_axis = axis; best_split = b; }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            mid = begin + count / 2;
        } else {
            float ext = cb.hi[best_axis] - cb.lo[best_axis], scale = NB / ext, lo = cb.lo[best_axis];
            int axis = best_axis, split = best_split;
            auto it = std::partition(order.begin() + begin, order.begin() + end, [&](uint32_t p) {
                int b = std::min(NB - 1, std::max(0, (int) ((centroid(p, axis) - lo) * scale)));
                return b <= split; });
            mid = (uint32_t) (it - order.begin());
            if (mid == begin || mid == end) mid = begin + count / 2;
        }
        static const uint32_t par_min = getenv("HAR_BUILD_PAR_MIN") ? (uint32_t) atol(getenv("HAR_BUILD_PAR_MIN")) : 16384u;
        if (at >= 0) {
            const int l = at + 1, r = at + 2 * (int) (mid - begin);
            if (count >= par_min && depth < 7) {
                bool forked = false; std::thread t;
                try { t = std::thread([&] { build_into(bn, begin, mid, depth + 1, l); }); forked = true; } catch (...) { }      /* no thread to be had: this one does both */
                if (!forked) build_into(bn, begin, mid, depth + 1, l);
                build_into(bn, mid, end, depth + 1, r);
                if (forked) t.join();
            } else { build_into(bn, begin, mid, depth + 1, l); build_into(bn, mid, end, depth + 1, r); }
            bn[idx].left = l; bn[idx].right = r;
            return idx;
        }
        if (count >= par_min && depth < 7) {
            std::vector<BNode> L, R;
            bool forked = false; std::thread t;
            try { t = std::thread([&] { build_into(L, begin, mid, depth + 1); }); forked = true; } catch (...) { }      /* no thread to be had: this one does both */
            if (!forked) build_into(L, begin, mid, depth + 1);
            build_into(R, mid, end, depth + 1);
            if (forked) t.join();
            auto splice = [&](const std::vector<BNode> &sub) {
                const int base = (int) bn.size();
                for (BNode n : sub) { if (n.left >= 0) { n.left += base; n.right += base; } bn.push_back(n); }
                return base;
            };
            const int l = splice(L), r = splice(R);
            bn[idx].left = l; bn[idx].right = r;
            return idx;
        }
        int l = build_into(bn, begin, mid, depth + 1), r = build_into(bn, mid, end, depth + 1);
        bn[idx].left = l; bn[idx].right = r;
        return idx;
    }
};

/* threads for the collapse and the node emission of a tree of `n` binary nodes (HAR_BUILD_EMIT_THREADS overrides; 1 = the sequential code) */
static unsigned emit_threads(size_t n) {
    static const int forced = getenv("HAR_BUILD_EMIT_THREADS") ? atoi(getenv("HAR_BUILD_EMIT_THREADS")) : 0;
    if (forced > 0) return (unsigned) forced;
    return n >= 65536 ? std::min(har_usable_cores(), 32u) : 1u;
}
/* fn(0) ... fn(n - 1) on up to `threads` threads (work items handed out by an atomic counter; a thread that cannot be created is simply missing) */
template <typename F> static void run_parallel(unsigned threads, size_t n, F &&fn) {
    std::atomic<size_t> next{ 0 };
    auto worker = [&] { for (size_t i; (i = next.fetch_add(1)) < n; ) fn(i); };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads && t < n; ++t) { try { pool.emplace_back(worker); } catch (...) { break; } }
    worker();
    for (auto &t : pool) t.join();
}

} // namespace

void pad_prim_box(PrimBox &b) { pad_box(b.lo, b.hi); }

uint32_t build_bvh8(const std::vector<PrimBox> &prims, std::vector<Node8> &nodes, uint32_t leaf_base,
                    std::vector<uint32_t> &leaf_order, Bvh8Stats *stats, uint32_t max_leaf, float prim_cost, uint32_t dp_min_prims) {
    const uint32_t root_out = (uint32_t) nodes.size();
    nodes.emplace_back();
    std::memset(&nodes[root_out], 0, sizeof(Node8));
    if (prims.empty()) {            // empty BLAS: a node without children
        nodes[root_out].ex = nodes[root_out].ey = nodes[root_out].ez = 1;
        return root_out;
    }
    static const bool timing = getenv("HAR_BUILD_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    Builder B(prims);
    (void) max_leaf;
    B.max_leaf = 1;                /* one primitive per leaf: the node format holds one bit per child slot (Node8::lmask); 2 and 3 were measured slower in round 1 */
    B.bn.reserve(prims.size());
    int broot = B.build(0, (uint32_t) prims.size());
    auto t1 = now();

    /* ---- optimal wide-BVH collapse by dynamic programming over the binary tree.
     * C(n, i) = least SAH cost of representing the subtree of binary node n by at most i roots (child slots of the parent wide node):
     *   C(n, 1) = min(leaf: A(n) * c_prim * count,  wide node: A(n) * c_node + min_k C(left, k) + C(right, 8 - k))
     *   C(n, i) = min(C(n, i - 1), min_k C(left, k) + C(right, i - k))
     * dp_split[n][i] = k of the best distribution, 0 = "use i - 1 roots"; dp_split[n][1] = k of the wide node made of n. */
    /* measured on MI355X (tools/ab_collapse.sh): +5 % paths/s on the 1M-triangle instanced scene, +1 % on the flattened one, but -5 % on the 36-triangle
     * Cornell box, where the greedy collapse's extra, tighter nodes pay off -> small primitive sets keep the greedy collapse */
    static const bool dp_enabled = !(getenv("HAR_BVH_COLLAPSE") && std::string(getenv("HAR_BVH_COLLAPSE")) == "greedy");
    const bool use_dp = dp_enabled && prims.size() >= (size_t) dp_min_prims;
    static const float c_node = getenv("HAR_BVH_CNODE") ? (float) atof(getenv("HAR_BVH_CNODE")) : 1.f;
    /* what-if: narrower nodes in the same 80-byte records (children packed into the first W storage positions, octant slots unchanged) */
    static const int W = getenv("HAR_BVH_WIDTH") ? std::max(2, std::min(8, atoi(getenv("HAR_BVH_WIDTH")))) : 8;
    const float c_prim = prim_cost;
    std::vector<float> dp_cost; std::vector<uint8_t> dp_split;
    if (use_dp) {
        const size_t nb = B.bn.size();
        dp_cost.assign(nb * 9, 0.f); dp_split.assign(nb * 9, 0);
        /* children have larger indices than their parent (pre-order construction): a reverse sweep is a post-order traversal, and a subtree is a contiguous
         * index range [root, end) -- the subtrees below a fork level are swept concurrently, the few nodes above them afterwards (same arithmetic per node,
         * so the tables do not depend on the thread count) */
        auto dp_node = [&](size_t n) {
            const BNode &b = B.bn[n];
            float *Cn = dp_cost.data() + n * 9; uint8_t *Sn = dp_split.data() + n * 9;
            const float A = b.box.area();
            if (b.count) { for (int i = 1; i <= W; ++i) Cn[i] = A * c_prim * (float) b.count; return; }
            const float *Cl = dp_cost.data() + (size_t) b.left * 9, *Cr = dp_cost.data() + (size_t) b.right * 9;
            float best = std::numeric_limits<float>::infinity(); int bk = 1;
            for (int k = 1; k < W; ++k) { float c = Cl[k] + Cr[W - k]; if (c < best) { best = c; bk = k; } }
            Cn[1] = A * c_node + best; Sn[1] = (uint8_t) bk;
            for (int i = 2; i <= W; ++i) {
                float bi = Cn[i - 1]; int ki = 0;
                for (int k = 1; k < i; ++k) { float c = Cl[k] + Cr[i - k]; if (c < bi) { bi = c; ki = k; } }
                Cn[i] = bi; Sn[i] = (uint8_t) ki;
            }
        };
        const unsigned threads = emit_threads(nb);
        if (threads <= 1) { for (size_t n = nb; n-- > 0; ) dp_node(n); }
        else {
            /* subtree ranges below the fork level: end(n) = end of its right subtree; a leaf ends at n + 1 */
            struct Range { size_t begin, end; };
            std::vector<Range> ranges; std::vector<size_t> top;
            struct Item { int n; int depth; };
            std::vector<Item> st; st.push_back({ broot, 0 });
            auto subtree_end = [&](int n) { while (!B.bn[n].count) n = B.bn[n].right; return (size_t) n + 1; };
            const size_t grain = std::max<size_t>(nb / (16 * (size_t) threads), 4096);
            while (!st.empty()) {
                Item it = st.back(); st.pop_back();
                const size_t e = subtree_end(it.n);
                if (B.bn[it.n].count || e - (size_t) it.n <= grain) { ranges.push_back({ (size_t) it.n, e }); continue; }
                top.push_back((size_t) it.n);
                st.push_back({ B.bn[it.n].left, it.depth + 1 }); st.push_back({ B.bn[it.n].right, it.depth + 1 });
            }
            run_parallel(threads, ranges.size(), [&](size_t r) { for (size_t n = ranges[r].end; n-- > ranges[r].begin; ) dp_node(n); });
            std::sort(top.begin(), top.end());
            for (size_t k = top.size(); k-- > 0; ) dp_node(top[k]);
        }
    }
    /* roots chosen for subtree m when it may use up to i slots */
    struct Collector {
        const Builder &B; const std::vector<uint8_t> &split;
        void operator()(int m, int i, int *child, int &nc) const {
            const BNode &b = B.bn[m];
            while (i > 1 && !b.count && split[(size_t) m * 9 + i] == 0) --i;
            if (i == 1 || b.count) { child[nc++] = m; return; }
            const int k = split[(size_t) m * 9 + i];
            (*this)(b.left, k, child, nc); (*this)(b.right, i - k, child, nc);
        }
    } collect{ B, dp_split };

    /* ---- node emission.  The sequential algorithm is a depth-first walk with an explicit stack: a wide node takes its (already reserved) output slot, reserves one
     * contiguous block for its inner children at the end of `nodes`, appends its leaf records to `leaf_order`, pushes the inner children in slot order and continues with
     * the one pushed LAST -- so a subtree owns one contiguous range of `nodes` and one of `leaf_order`, both known once the subtree's node and leaf COUNTS are.  A first
     * pass counts (cheap: it only follows the collapse's choices), then subtrees are emitted concurrently into their ranges by the same walk with private cursors: the arrays
     * are the sequential walk's byte for byte, whatever the thread count (tests/test_cpu_host.py::test_builder_threads_do_not_change_the_tree). */
    auto children_of = [&](int b, int *child) {
        int nc = 0;
        const BNode &bn = B.bn[b];
        if (bn.count) { child[nc++] = b; return nc; }
        if (use_dp) {
            /* SAH-optimal collapse for the given binary topology (Ylitie, Karras, Laine 2017, sec. 3.2): the children of this wide
             * node are the <= 8 roots the dynamic program chose for the left and right binary subtrees */
            const int k = dp_split[(size_t) b * 9 + 1];
            collect(bn.left, k, child, nc); collect(bn.right, W - k, child, nc);
        } else {
            child[nc++] = bn.left; child[nc++] = bn.right;
            // greedy collapse: open the internal child with the largest surface area
            for (;;) {
                if (nc >= W) break;
                int best = -1; float ba = -1.f;
                for (int i = 0; i < nc; ++i) if (!B.bn[child[i]].count) { float a = B.bn[child[i]].box.area(); if (a > ba) { ba = a; best = i; } }
                if (best < 0) break;
                int c = child[best];
                child[best] = B.bn[c].left; child[nc++] = B.bn[c].right;
            }
        }
        return nc;
    };
    struct Work { int b; uint32_t out; uint32_t depth; };
    struct Task { Work w; uint32_t node_cursor, leaf_cursor; };
    const unsigned threads = emit_threads(B.bn.size());
    /* pass 1: wide nodes and leaf records of the wide subtree rooted at every binary node that becomes a wide node */
    std::vector<uint32_t> sub_nodes, sub_leaves;
    {
        sub_nodes.assign(B.bn.size(), 0u); sub_leaves.assign(B.bn.size(), 0u);
        struct Frame { int b; int child[8]; int nc; int next; };
        std::vector<Frame> st; st.reserve(64);
        auto open = [&](int b) { Frame f; f.b = b; f.nc = children_of(b, f.child); f.next = 0; sub_nodes[b] = 1u; sub_leaves[b] = 0u; st.push_back(f); };
        open(broot);
        while (!st.empty()) {
            Frame &f = st.back();
            if (f.next == f.nc) {
                const int b = f.b; st.pop_back();
                if (!st.empty()) { sub_nodes[st.back().b] += sub_nodes[b]; sub_leaves[st.back().b] += sub_leaves[b]; }
                continue;
            }
            const int c = f.child[f.next++];
            if (B.bn[c].count) sub_leaves[f.b] += 1u; else open(c);           /* `f` is not used after open(): the vector may move */
        }
    }
    /* the walk of one subtree: `node_cursor` / `leaf_cursor` = the next free entries of its ranges; with `defer` the subtrees of at most `grain` wide nodes are not
     * entered but filed as tasks (their ranges are skipped) */
    const uint32_t first_leaf = (uint32_t) leaf_order.size();
    nodes.resize((size_t) root_out + sub_nodes[broot]);                  /* the root's own record is nodes[root_out] (reserved above) */
    leaf_order.resize((size_t) first_leaf + sub_leaves[broot]);
    auto emit = [&](Task t, std::vector<Task> *defer, uint32_t grain) -> uint32_t {
        uint32_t depth_seen = 1, node_cursor = t.node_cursor, leaf_cursor = t.leaf_cursor;
        std::vector<Work> queue; queue.push_back(t.w);
        while (!queue.empty()) {
            Work w = queue.back(); queue.pop_back();
            if (defer && w.b != t.w.b && sub_nodes[w.b] <= grain) {
                defer->push_back({ w, node_cursor, leaf_cursor });
                node_cursor += sub_nodes[w.b] - 1u; leaf_cursor += sub_leaves[w.b];
                continue;
            }
            depth_seen = std::max(depth_seen, w.depth);
            int child[8]; const int nc = children_of(w.b, child);
            // node box
            Box nb; nb.reset();
            for (int i = 0; i < nc; ++i) nb.grow(B.bn[child[i]].box);
            // octant-ordered slot assignment (greedy on cost = <centroid offset, slot direction>)
            int slot_of[8]; bool slot_used[8] = { false }; bool done[8] = { false };
            float cx = 0.5f * (nb.lo[0] + nb.hi[0]), cy = 0.5f * (nb.lo[1] + nb.hi[1]), cz = 0.5f * (nb.lo[2] + nb.hi[2]);
            for (int k = 0; k < nc; ++k) {
                float bestc = std::numeric_limits<float>::infinity(); int bi = -1, bs = -1;
                for (int i = 0; i < nc; ++i) {
                    if (done[i]) continue;
                    const Box &cb = B.bn[child[i]].box;
                    float ox = 0.5f * (cb.lo[0] + cb.hi[0]) - cx, oy = 0.5f * (cb.lo[1] + cb.hi[1]) - cy, oz = 0.5f * (cb.lo[2] + cb.hi[2]) - cz;
                    for (int s = 0; s < 8; ++s) {
                        if (slot_used[s]) continue;
                        float cost = ox * ((s & 4) ? -1.f : 1.f) + oy * ((s & 2) ? -1.f : 1.f) + oz * ((s & 1) ? -1.f : 1.f);
                        if (cost < bestc) { bestc = cost; bi = i; bs = s; }
                    }
                }
                done[bi] = true; slot_used[bs] = true; slot_of[bi] = bs;
            }
            int child_in_slot[8]; for (int s = 0; s < 8; ++s) child_in_slot[s] = -1;
            for (int i = 0; i < nc; ++i) child_in_slot[slot_of[i]] = child[i];

            Node8 n; std::memset(&n, 0, sizeof(n));
            node_set_frame(n, nb.lo, nb.hi);
            n.child_base = node_cursor;
            n.tri_base = leaf_base + leaf_cursor;
            uint32_t n_internal = 0;
            for (int s = 0; s < 8; ++s) {
                int c = child_in_slot[s];
                if (c < 0) continue;
                const BNode &cn = B.bn[c];
                node_quantise_child(n, s, cn.box.lo, cn.box.hi);
                if (cn.count) {
                    /* leaf slot: its record is tri_base + (number of leaf slots below s) -- the records are appended in slot order */
                    n.lmask |= (uint8_t) (1u << s);
                    leaf_order[leaf_cursor++] = B.order[cn.first];
                } else {
                    n.imask |= (uint8_t) (1u << s);
                    ++n_internal;
                }
            }
            nodes[w.out] = n;
            const uint32_t base = node_cursor;
            node_cursor += n_internal;
            uint32_t k = 0;
            for (int s = 0; s < 8; ++s) {
                int c = child_in_slot[s];
                if (c < 0 || B.bn[c].count) continue;
                queue.push_back({ c, base + k, w.depth + 1 }); ++k;
            }
        }
        return depth_seen;
    };
    uint32_t max_depth = 1;
    const Task whole{ { broot, root_out, 1 }, root_out + 1u, first_leaf };
    if (threads <= 1) max_depth = emit(whole, nullptr, 0u);
    else {
        std::vector<Task> tasks;
        max_depth = emit(whole, &tasks, std::max(256u, sub_nodes[broot] / (8u * threads)));
        std::vector<uint32_t> depth_of(tasks.size(), 1u);
        run_parallel(threads, tasks.size(), [&](size_t i) { depth_of[i] = emit(tasks[i], nullptr, 0u); });
        for (uint32_t d : depth_of) max_depth = std::max(max_depth, d);
    }
    if (stats) { stats->max_depth = std::max(stats->max_depth, max_depth); }
    if (timing && prims.size() > 1000) {
        auto t2 = now();
        fprintf(stderr, "[hip_ad_rgb] build_bvh8: %zu prims, binary build %.3f s, collapse + node emission %.3f s\n", prims.size(),
                std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
    }
    return root_out;
}

} // namespace har
