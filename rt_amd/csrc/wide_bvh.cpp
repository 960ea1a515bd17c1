// wide_bvh.cpp -- see wide_bvh.h.
#include "wide_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "bvh_build.h"

namespace rtamd {

namespace {

struct Info {
    float mn[3], mx[3]; // tight bounds of the subtree
    int32_t first;      // DFS rank of the first primitive below
    int32_t count;      // primitives below
};

inline float half_area(const Info &b) {
    float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}

} // namespace

bool dfs_leaf_order(const rt_bvh_node32 *nodes, int m, int n, std::vector<int32_t> &order, std::string &err) {
    order.clear();
    if (n <= 0 || m != 2 * n - 1) {
        err = "node count must be 2*n-1";
        return false;
    }
    order.reserve(n);
    std::vector<uint8_t> seen(m, 0), prim_seen(n, 0);
    std::vector<int32_t> stk;
    stk.push_back(m - 1);
    while (!stk.empty()) {
        int32_t c = stk.back();
        stk.pop_back();
        if (c < 0 || c >= m || seen[c]) {
            err = "node index out of range or reached twice: not a tree";
            return false;
        }
        seen[c] = 1;
        const rt_bvh_node32 &nd = nodes[c];
        if (nd.left == nd.right) {
            if (nd.left < 0 || nd.left >= n || prim_seen[nd.left]) {
                err = "leaf primitive index out of range or repeated";
                return false;
            }
            prim_seen[nd.left] = 1;
            order.push_back(nd.left);
        } else {
            stk.push_back(nd.right);
            stk.push_back(nd.left);
        }
    }
    if ((int)order.size() != n) {
        err = "tree does not cover every primitive";
        return false;
    }
    return true;
}

bool build_wide_bvh(const rt_triangle148 *tris, int n, const rt_bvh_node32 *nodes, int m, int leaf_size, int box_mode,
                    const int32_t *ref_rank, WideBvh &out, std::string &err, int collapse) {
    out = WideBvh();
    if (n <= 0 || m <= 0) {
        err = "empty scene";
        return false;
    }
    if (m != 2 * n - 1) {
        err = "node count must be 2*n-1";
        return false;
    }
    if ((unsigned)n > kLeafStartMask) {
        err = "too many primitives for the 28-bit leaf reference";
        return false;
    }
    leaf_size = std::max(1, std::min(leaf_size, kMaxLeafPrims));

    // ---- pass 1: DFS from the root; rank leaves, collect per-subtree info, validate the tree ----
    std::vector<Info> info(m);
    std::vector<uint8_t> seen(m, 0);
    std::vector<int32_t> order; // order[rank] = primitive index
    order.reserve(n);
    std::vector<uint8_t> prim_seen(n, 0);
    struct Frame {
        int32_t node;
        int32_t phase;
    };
    std::vector<Frame> stk;
    std::vector<int32_t> post; // every node after its children
    post.reserve(m);
    stk.push_back({m - 1, 0});
    while (!stk.empty()) {
        Frame f = stk.back();
        stk.pop_back();
        const rt_bvh_node32 &nd = nodes[f.node];
        Info &I = info[f.node];
        if (f.phase == 0) {
            if (seen[f.node]) {
                err = "node reached twice: not a tree";
                return false;
            }
            seen[f.node] = 1;
            I.first = (int32_t)order.size();
            if (nd.left == nd.right) {
                if (nd.left < 0 || nd.left >= n || prim_seen[nd.left]) {
                    err = "leaf primitive index out of range or repeated";
                    return false;
                }
                prim_seen[nd.left] = 1;
                order.push_back(nd.left);
                I.count = 1;
                prim_box_tight(tris[nd.left], I.mn, I.mx);
                post.push_back(f.node);
            } else {
                if (nd.left < 0 || nd.left >= m || nd.right < 0 || nd.right >= m) {
                    err = "child index out of range";
                    return false;
                }
                stk.push_back({f.node, 1});
                stk.push_back({nd.right, 0});
                stk.push_back({nd.left, 0});
            }
        } else {
            const Info &L = info[nd.left], &R = info[nd.right];
            I.count = L.count + R.count;
            for (int a = 0; a < 3; a++) {
                I.mn[a] = std::min(L.mn[a], R.mn[a]);
                I.mx[a] = std::max(L.mx[a], R.mx[a]);
            }
            post.push_back(f.node);
        }
    }
    if ((int)order.size() != n) {
        err = "tree does not cover every primitive";
        return false;
    }

    // ---- primitives in DFS leaf order ----
    out.prims.resize(n);
    out.shade.resize(n);
    for (int r = 0; r < n; r++) {
        const rt_triangle148 &t = tris[order[r]];
        DevPrim &p = out.prims[r];
        DevShade &s = out.shade[r];
        if (t.shapeType == 0) {
            p.a[0] = t.v0[0]; p.a[1] = t.v0[1]; p.a[2] = t.v0[2]; p.a[3] = t.edge1[0];
            p.b[0] = t.edge1[1]; p.b[1] = t.edge1[2]; p.b[2] = t.edge2[0]; p.b[3] = t.edge2[1];
            p.e2z = t.edge2[2];
            p.shape = 0;
            s.normal[0] = t.normal[0]; s.normal[1] = t.normal[1]; s.normal[2] = t.normal[2];
        } else {
            p.a[0] = t.v0[0]; p.a[1] = t.v0[1]; p.a[2] = t.v0[2]; p.a[3] = t.v1[0];
            p.b[0] = p.b[1] = p.b[2] = p.b[3] = 0.0f;
            p.e2z = 0.0f;
            p.shape = 1;
            s.normal[0] = s.normal[1] = s.normal[2] = 0.0f;
        }
        p.orig = order[r];
        p.ref_rank = ref_rank ? ref_rank[order[r]] : r;
        s.mat = t.matType;
    }

    // ---- texture coordinates, only for scenes that use them ----
    bool textured = false;
    for (int i = 0; i < n && !textured; i++) textured = tris[i].shapeType == 0 && tris[i].texIdx != -1;
    if (textured) {
        out.uv.resize(n);
        for (int r = 0; r < n; r++) {
            const rt_triangle148 &t = tris[order[r]];
            DevUV &q = out.uv[r];
            for (int k = 0; k < 3; k++) q.u[k] = t.u[k];
            q.v0 = t.v[0];
            q.v12[0] = t.v[1];
            q.v12[1] = t.v[2];
            q.tex = t.shapeType == 0 ? t.texIdx : -1; // a sphere's hit record always says -1 (triangle.h:207)
            q.pad = 0;
        }
    }

    // ---- pad for re-fitted boxes: 2^-17 of the largest |coordinate| of the scene ----
    float extent = 0.0f;
    {
        const Info &R = info[m - 1];
        for (int a = 0; a < 3; a++) extent = std::max(extent, std::max(std::fabs(R.mn[a]), std::fabs(R.mx[a])));
    }
    out.pad = std::ldexp(extent > 1.0f ? extent : 1.0f, -17);

    auto write_box = [&](WideNode &w, int slot, int32_t bnode) {
        if (box_mode == 1) {
            const rt_bvh_node32 &nd = nodes[bnode];
            for (int a = 0; a < 3; a++) {
                w.plane[a][slot] = nd.box_min[a];
                w.plane[3 + a][slot] = nd.box_max[a];
            }
        } else {
            const Info &I = info[bnode];
            for (int a = 0; a < 3; a++) {
                w.plane[a][slot] = I.mn[a] - out.pad;
                w.plane[3 + a][slot] = I.mx[a] + out.pad;
            }
        }
    };

    // ---- collapse 1: cost[i-1] = least surface-area cost of the subtree of binary node b as at most i children of a
    // wide node (i = 1..3).  One child is a leaf (subtrees of at most leaf_size primitives, always) or a wide node,
    // whose own four slots are shared between b's two children in the cheapest way (left4 = the left child's share);
    // with i > 1 slots b may also dissolve into use[i-1] of them, left[i-1] going to its left child.
    const float kNodeCost = 1.0f, kPrimCost = 1.0f; // a node step and a primitive test cost about the same per lane
    struct Dp {
        float cost[3];
        uint8_t use[3], left[3], left4;
    };
    std::vector<Dp> dp;
    std::vector<uint8_t> leafy;
    if (collapse == 1) {
        dp.resize(m);
        leafy.resize(m);
        for (int32_t b : post) {
            Dp &D = dp[b];
            const float area = half_area(info[b]);
            leafy[b] = nodes[b].left == nodes[b].right || info[b].count <= leaf_size;
            if (leafy[b]) {
                D.cost[0] = D.cost[1] = D.cost[2] = area * kPrimCost * (float)info[b].count;
                D.use[0] = D.use[1] = D.use[2] = 1;
                D.left[0] = D.left[1] = D.left[2] = D.left4 = 0;
                continue;
            }
            const Dp &L = dp[nodes[b].left], &R = dp[nodes[b].right];
            float dist[5];
            uint8_t arg[5];
            for (int j = 2; j <= 4; j++) { // b's subtree over exactly-at-most j slots, k of them to the left child
                dist[j] = std::numeric_limits<float>::infinity();
                arg[j] = 1;
                for (int k = std::max(1, j - 3); k <= std::min(3, j - 1); k++) {
                    const float c = L.cost[k - 1] + R.cost[j - k - 1];
                    if (c < dist[j]) {
                        dist[j] = c;
                        arg[j] = (uint8_t)k;
                    }
                }
            }
            D.left4 = arg[4];
            D.cost[0] = area * kNodeCost + dist[4];
            D.use[0] = 1;
            D.left[0] = 0;
            for (int i = 2; i <= 3; i++) {
                if (dist[i] < D.cost[i - 2]) {
                    D.cost[i - 1] = dist[i];
                    D.use[i - 1] = (uint8_t)i;
                    D.left[i - 1] = arg[i];
                } else { // one slot less was as cheap: the same choice
                    D.cost[i - 1] = D.cost[i - 2];
                    D.use[i - 1] = D.use[i - 2];
                    D.left[i - 1] = D.left[i - 2];
                }
            }
        }
    }
    // the children binary node b contributes when it may take up to i slots, in DFS order
    struct Expand {
        const std::vector<Dp> &dp;
        const std::vector<uint8_t> &leafy;
        const rt_bvh_node32 *nodes;
        void run(int32_t b, int i, int32_t *slots, int &ns) const {
            const Dp &D = dp[b];
            if (leafy[b] || D.use[i - 1] <= 1) {
                slots[ns++] = b;
                return;
            }
            run(nodes[b].left, D.left[i - 1], slots, ns);
            run(nodes[b].right, D.use[i - 1] - D.left[i - 1], slots, ns);
        }
    };

    // ---- pass 2: breadth-first collapse, largest half-area first ----
    struct Work {
        int32_t bnode; // binary subtree to expand into this wide node
        int32_t wnode;
        int32_t level;
    };
    std::vector<Work> queue;
    out.nodes.reserve((size_t)n / 2 + 16);
    out.nodes.emplace_back();
    queue.push_back({m - 1, 0, 1});
    int max_level = 1;
    for (size_t qi = 0; qi < queue.size(); qi++) {
        Work wk = queue[qi];
        max_level = std::max(max_level, wk.level);
        int32_t slots[4];
        int ns = 0;
        const rt_bvh_node32 &b = nodes[wk.bnode];
        if (info[wk.bnode].count <= leaf_size || b.left == b.right) {
            slots[ns++] = wk.bnode; // whole (sub)scene fits one leaf: a node with a single leaf child
        } else if (collapse == 1) {
            const Expand ex{dp, leafy, nodes};
            const int k = dp[wk.bnode].left4;
            ex.run(b.left, k, slots, ns);
            ex.run(b.right, 4 - k, slots, ns);
        } else {
            slots[ns++] = b.left;
            slots[ns++] = b.right;
            while (ns < 4) {
                int pick = -1;
                float best = -1.0f;
                for (int s = 0; s < ns; s++) {
                    const rt_bvh_node32 &c = nodes[slots[s]];
                    if (c.left == c.right || info[slots[s]].count <= leaf_size) continue;
                    float a = half_area(info[slots[s]]);
                    if (a > best) {
                        best = a;
                        pick = s;
                    }
                }
                if (pick < 0) break;
                const rt_bvh_node32 &c = nodes[slots[pick]];
                // keep DFS order among the slots: left child takes the slot, right child goes right after it
                for (int s = ns; s > pick + 1; s--) slots[s] = slots[s - 1];
                slots[pick] = c.left;
                slots[pick + 1] = c.right;
                ns++;
            }
        }
        WideNode w;
        std::memset(&w, 0, sizeof(w));
        for (int s = 0; s < 4; s++) {
            if (s < ns) {
                const Info &I = info[slots[s]];
                const rt_bvh_node32 &c = nodes[slots[s]];
                write_box(w, s, slots[s]);
                if (c.left == c.right || I.count <= leaf_size) {
                    w.child[s] = kLeafBit | ((uint32_t)(I.count - 1) << 28) | (uint32_t)I.first;
                } else {
                    int32_t idx = (int32_t)out.nodes.size();
                    out.nodes.emplace_back();
                    w.child[s] = (uint32_t)idx;
                    queue.push_back({slots[s], idx, wk.level + 1});
                }
            } else {
                for (int a = 0; a < 3; a++) {
                    w.plane[a][s] = std::numeric_limits<float>::infinity();
                    w.plane[3 + a][s] = -std::numeric_limits<float>::infinity();
                }
                w.child[s] = kEmptyRef;
            }
        }
        out.nodes[wk.wnode] = w;
    }
    out.depth = max_level;
    out.stack_entries = 3 * max_level + 2;

    // ---- quantized copy: 8-bit child boxes on a power-of-two grid anchored at the node's min corner ----
    out.qnodes.resize(out.nodes.size());
    for (size_t i = 0; i < out.nodes.size(); i++) {
        const WideNode &w = out.nodes[i];
        QuantNode q;
        std::memset(&q, 0, sizeof(q));
        uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
        for (int a = 0; a < 3; a++) {
            float org = std::numeric_limits<float>::infinity(), top = -std::numeric_limits<float>::infinity();
            for (int s = 0; s < 4; s++)
                if (w.child[s] != kEmptyRef) {
                    org = std::min(org, w.plane[a][s]);
                    top = std::max(top, w.plane[3 + a][s]);
                }
            double extent = (double)top - (double)org;
            int e = -100;
            if (extent > 0.0) {
                e = (int)std::ceil(std::log2(extent / 255.0));
                if (e < -100) e = -100;
            }
            for (;; e++) { // grow the step until every decoded plane is conservative
                bool ok = true;
                double step = std::ldexp(1.0, e);
                for (int s = 0; s < 4 && ok; s++) {
                    if (w.child[s] == kEmptyRef) continue;
                    double lo = std::floor(((double)w.plane[a][s] - (double)org) / step);
                    double hi = std::ceil(((double)w.plane[3 + a][s] - (double)org) / step);
                    if (lo < 0.0) lo = 0.0;
                    if (hi > 255.0 || lo > 255.0) ok = false;
                    else if ((double)org + lo * step > (double)w.plane[a][s] || (double)org + hi * step < (double)w.plane[3 + a][s])
                        ok = false;
                }
                if (ok) break;
            }
            double step = std::ldexp(1.0, e);
            uint32_t plo = 0, phi = 0;
            for (int s = 0; s < 4; s++) {
                uint32_t lo = 255u, hi = 0u; // empty slot: inverted, can never be hit
                if (w.child[s] != kEmptyRef) {
                    double l = std::floor(((double)w.plane[a][s] - (double)org) / step);
                    double h = std::ceil(((double)w.plane[3 + a][s] - (double)org) / step);
                    lo = (uint32_t)(l < 0.0 ? 0.0 : l);
                    hi = (uint32_t)h;
                }
                plo |= lo << (8 * s);
                phi |= hi << (8 * s);
            }
            qlo[a] = plo;
            qhi[a] = phi;
            q.origin[a] = org;
            (a == 0 ? q.step_x : a == 1 ? q.step_y : q.step_z) = (float)step; // 2^e, e in [-100, 127]: exact
        }
        q.qx_lo = qlo[0]; q.qx_hi = qhi[0];
        q.qy_lo = qlo[1]; q.qy_hi = qhi[1];
        q.qz_lo = qlo[2]; q.qz_hi = qhi[2];
        for (int s = 0; s < 4; s++) q.child[s] = w.child[s];
        out.qnodes[i] = q;
    }
    return true;
}

bool build_pack_scene(const WideBvh &w, PackScene &out, std::string &why) {
    out.recs.clear();
    out.rec_of_prim.clear();
    const size_t n_wide = w.qnodes.size(), n_prims = w.prims.size();
    if (n_wide == 0 || n_prims == 0) return why = "no scene", false;
    if (3 * (uint64_t)(n_wide + n_prims) >= kPackMaxUnits) return why = "2^28 units or more", false;
    out.recs.resize(n_wide + n_prims);
    out.rec_of_prim.assign(n_prims, 0xFFFFFFFFu);
    // wide nodes are in breadth-first order and every child has a larger index than its parent: walking them by index,
    // a node's own record is known (its parent placed it) by the time its child block is laid down
    std::vector<uint32_t> rec_of_node(n_wide, 0xFFFFFFFFu);
    rec_of_node[0] = 0;
    uint32_t next = 1; // next free record
    for (size_t i = 0; i < n_wide; i++) {
        const QuantNode &q = w.qnodes[i];
        if (rec_of_node[i] == 0xFFFFFFFFu) return out.recs.clear(), out.rec_of_prim.clear(), why = "internal: unreferenced wide node", false;
        PackNode pk;
        std::memcpy(pk.origin, q.origin, sizeof(pk.origin));
        pk.qx_lo = q.qx_lo, pk.qx_hi = q.qx_hi, pk.qy_lo = q.qy_lo, pk.qy_hi = q.qy_hi, pk.qz_lo = q.qz_lo, pk.qz_hi = q.qz_hi;
        const float steps[3] = {q.step_x, q.step_y, q.step_z};
        uint32_t exps = 0;
        for (int a = 0; a < 3; a++) {
            int e = 0;
            const float m = std::frexp(steps[a], &e); // step = m * 2^e, m = 0.5 for a power of two
            e -= 1;
            if (m != 0.5f || e < -126 || e > 126) return out.recs.clear(), out.rec_of_prim.clear(), why = "grid step is not a normal power of two", false;
            exps |= ((uint32_t)e & 0xFFu) << (8 * a);
        }
        pk.exps = exps;
        const uint32_t block = next;
        pk.base4 = 4u * 3u * block;
        uint32_t bytes = 0;
        for (int s = 0; s < 4; s++) {
            const uint32_t r = q.child[s];
            uint32_t b = kPackEmpty;
            if (r != kEmptyRef && (r & kLeafBit)) {
                const uint32_t first = r & kLeafStartMask, cnt = ((r >> 28) & 7u) + 1u;
                if (cnt > 2u) return out.recs.clear(), out.rec_of_prim.clear(), why = "a leaf holds more than two primitives", false;
                b = (3u * (next - block)) << 2 | kPackLeafBit | (cnt - 1u);
                for (uint32_t k = 0; k < cnt; k++) {
                    if (first + k >= n_prims || out.rec_of_prim[first + k] != 0xFFFFFFFFu)
                        return out.recs.clear(), out.rec_of_prim.clear(), why = "internal: leaves do not partition the primitives", false;
                    out.rec_of_prim[first + k] = next;
                    std::memcpy(&out.recs[next], &w.prims[first + k], sizeof(PackRecord));
                    next++;
                }
            } else if (r != kEmptyRef) {
                if (r >= n_wide || r <= i || rec_of_node[r] != 0xFFFFFFFFu)
                    return out.recs.clear(), out.rec_of_prim.clear(), why = "internal: child reference out of order", false;
                b = (3u * (next - block)) << 2;
                rec_of_node[r] = next++;
            }
            bytes |= b << (8 * s);
        }
        pk.child_bytes = bytes;
        std::memcpy(&out.recs[rec_of_node[i]], &pk, sizeof(PackRecord));
    }
    if (next != n_wide + n_prims) return out.recs.clear(), out.rec_of_prim.clear(), why = "internal: records do not add up", false;
    return true;
}

} // namespace rtamd
