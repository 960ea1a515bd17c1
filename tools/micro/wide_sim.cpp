// wide_sim.cpp -- host-side work model of W-wide trees (W = 4, 8) over the SAME binned-SAH binary tree the device
// layout is made from: how many node visits, child-box tests and primitive tests an ordered, t-shrinking
// closest-hit traversal needs per ray, for camera rays and two generations of cosine-scattered bounce rays.
// Answers VERDICT r01 item 4c ("measure 8-wide nodes") as far as a model can: the kernel is bound by L1 accesses
// (bytes per node visit / 16) and vector instructions (~ child-box tests), both of which follow from these counts.
//   g++ -O2 -std=c++17 tools/micro/wide_sim.cpp rt_amd/csrc/sah_build.o rt_amd/csrc/bvh_build.o -pthread -o tools/micro/wide_sim
//   tools/micro/wide_sim scene.bin   (scene.bin: rt_triangle148 records, e.g. scenes.heightfield(708).tofile)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../rt_amd/csrc/bvh_build.h"
#include "../../rt_amd/csrc/sah_build.h"

using namespace rtamd;

struct V3 { float x, y, z; };
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 unit(V3 a) { float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }

struct Box { float mn[3], mx[3]; };
struct WNode { // W <= 8 children
    int n = 0;
    Box box[8];
    int32_t child[8]; // >= 0: wide node; < 0: leaf ~(first) with count in cnt
    int32_t first[8], cnt[8];
};

struct Sim {
    const rt_triangle148 *tris;
    int n;
    std::vector<rt_bvh_node32> bn;
    std::vector<int32_t> count, first, order;
    std::vector<Box> tight;
    std::vector<WNode> wn;
    int W, leaf;

    void prep() {
        const int m = (int)bn.size();
        count.assign(m, 0), first.assign(m, 0), tight.resize(m);
        // post-order array, root last: children precede parents -> one forward sweep gives counts and tight boxes
        for (int i = 0; i < m; i++) {
            const rt_bvh_node32 &b = bn[i];
            if (b.left == b.right) {
                count[i] = 1;
                prim_box_tight(tris[b.left], tight[i].mn, tight[i].mx);
            } else {
                count[i] = count[b.left] + count[b.right];
                for (int a = 0; a < 3; a++) {
                    tight[i].mn[a] = std::min(tight[b.left].mn[a], tight[b.right].mn[a]);
                    tight[i].mx[a] = std::max(tight[b.left].mx[a], tight[b.right].mx[a]);
                }
            }
        }
        order.clear();
        std::vector<int32_t> st{m - 1};
        while (!st.empty()) { // DFS leaf order
            int32_t i = st.back();
            st.pop_back();
            const rt_bvh_node32 &b = bn[i];
            first[i] = (int32_t)order.size(); // valid for leaves; for inner nodes set below
            if (b.left == b.right) order.push_back(b.left);
            else st.push_back(b.right), st.push_back(b.left);
        }
        // first[] of inner nodes = first of their leftmost leaf
        for (int i = 0; i < m; i++)
            if (bn[i].left != bn[i].right) {
                int32_t j = i;
                while (bn[j].left != bn[j].right) j = bn[j].left;
                first[i] = first[j];
            }
    }
    static float harea(const Box &b) {
        float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
    bool leafy(int32_t b) const { return bn[b].left == bn[b].right || count[b] <= leaf; }
    void collapse(int W_, int leaf_) {
        W = W_, leaf = leaf_;
        wn.clear();
        wn.emplace_back();
        struct Work { int32_t b, w; };
        std::vector<Work> q{{(int32_t)bn.size() - 1, 0}};
        for (size_t qi = 0; qi < q.size(); qi++) {
            Work wk = q[qi];
            int32_t slots[8];
            int ns = 0;
            if (leafy(wk.b)) slots[ns++] = wk.b;
            else {
                slots[ns++] = bn[wk.b].left, slots[ns++] = bn[wk.b].right;
                while (ns < W) { // expand the child with the largest half-area, keeping DFS order
                    int pick = -1;
                    float best = -1;
                    for (int s = 0; s < ns; s++)
                        if (!leafy(slots[s]) && harea(tight[slots[s]]) > best) best = harea(tight[slots[s]]), pick = s;
                    if (pick < 0) break;
                    for (int s = ns; s > pick + 1; s--) slots[s] = slots[s - 1];
                    int32_t c = slots[pick];
                    slots[pick] = bn[c].left, slots[pick + 1] = bn[c].right;
                    ns++;
                }
            }
            WNode w;
            w.n = ns;
            for (int s = 0; s < ns; s++) {
                w.box[s] = tight[slots[s]];
                w.first[s] = first[slots[s]], w.cnt[s] = count[slots[s]];
                if (leafy(slots[s])) w.child[s] = -1;
                else {
                    w.child[s] = (int32_t)wn.size();
                    wn.emplace_back();
                    q.push_back({slots[s], w.child[s]});
                }
            }
            wn[wk.w] = w;
        }
    }
    struct Counts { double rays = 0, visits = 0, box = 0, prim = 0, hits = 0, cullable = 0, nohit = 0; };
    bool tri_hit(int32_t prim, V3 o, V3 d, float tmax, float &t) const {
        const rt_triangle148 &T = tris[prim];
        if (T.shapeType != 0) return false;
        V3 v0{T.v0[0], T.v0[1], T.v0[2]}, e1{T.edge1[0], T.edge1[1], T.edge1[2]}, e2{T.edge2[0], T.edge2[1], T.edge2[2]};
        V3 p = cross(d, e2);
        float det = dot(e1, p);
        if (det == 0.0f) return false;
        float inv = 1.0f / det;
        V3 tv = o - v0;
        float u = dot(tv, p) * inv;
        if (u < 0 || u > 1) return false;
        V3 qv = cross(tv, e1);
        float v = dot(d, qv) * inv;
        if (v < 0 || u + v > 1) return false;
        t = dot(e2, qv) * inv;
        return t > 0.001f && t < tmax;
    }
    int trace(V3 o, V3 d, float &tbest, Counts &c) const {
        const float inv[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z}, og[3] = {o.x, o.y, o.z};
        int best = -1;
        tbest = 3.0e38f;
        int32_t stack[256];
        float stack_tn[256]; // entry distance of the pushed child: a visit whose entry lies beyond the best hit is wasted
        int sp = 0;
        stack_tn[sp] = 0.0f, stack[sp++] = 0;
        c.rays++;
        while (sp) {
            --sp;
            int32_t cur = stack[sp];
            if (cur < 0) continue;
            const WNode &w = wn[cur];
            c.visits++;
            if (stack_tn[sp] > tbest) c.cullable++;
            c.box += W; // the kernel tests every slot of a node, empty or not
            struct Hitc { float tn; int s; } hc[8];
            int nh = 0;
            for (int s = 0; s < w.n; s++) {
                float tn = 0.001f, tf = tbest;
                for (int a = 0; a < 3; a++) {
                    float t0 = (w.box[s].mn[a] - og[a]) * inv[a], t1 = (w.box[s].mx[a] - og[a]) * inv[a];
                    if (t0 > t1) std::swap(t0, t1);
                    tn = std::max(tn, t0), tf = std::min(tf, t1);
                }
                if (tn <= tf) hc[nh++] = {tn, s};
            }
            if (nh == 0) c.nohit++;
            std::sort(hc, hc + nh, [](const Hitc &a, const Hitc &b) { return a.tn > b.tn; }); // far first onto the stack
            for (int k = 0; k < nh; k++) {
                int s = hc[k].s;
                if (w.child[s] >= 0) stack_tn[sp] = hc[k].tn, stack[sp++] = w.child[s];
                else { // leaves are tested where they are met (the kernel postpones them; same counts)
                    for (int i = 0; i < w.cnt[s]; i++) {
                        float t;
                        c.prim++;
                        if (tri_hit(order[w.first[s] + i], o, d, tbest, t)) tbest = t, best = order[w.first[s] + i];
                    }
                }
            }
        }
        if (best >= 0) c.hits++;
        return best;
    }
};

int main(int argc, char **argv) {
    if (argc < 2) return std::fprintf(stderr, "usage: wide_sim scene.bin [nx ny]\n"), 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return std::perror(argv[1]), 2;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<rt_triangle148> tris(bytes / sizeof(rt_triangle148));
    if (std::fread(tris.data(), sizeof(rt_triangle148), tris.size(), f) != tris.size()) return 2;
    std::fclose(f);
    const int nx = argc > 3 ? std::atoi(argv[2]) : 240, ny = argc > 3 ? std::atoi(argv[3]) : 135;
    Sim S;
    S.tris = tris.data(), S.n = (int)tris.size();
    build_sah_bvh(tris.data(), S.n, S.bn);
    S.prep();
    // bench camera: (0,14,18) -> origin, 40 degrees, 16:9
    const V3 from{0, 14, 18}, w = unit(from - V3{0, 0, 0}), u = unit(cross(V3{0, 1, 0}, w)), v = cross(w, u);
    const float hh = std::tan(20.0f * 3.14159265f / 180.0f), hw = hh * 16.0f / 9.0f;
    for (int W : {4, 8})
        for (int leaf : {2, 4}) {
            S.collapse(W, leaf);
            Sim::Counts c[3];
            std::mt19937 rng(7);
            std::uniform_real_distribution<float> U(0.0f, 1.0f);
            for (int j = 0; j < ny; j++)
                for (int i = 0; i < nx; i++) {
                    V3 o = from;
                    V3 d = unit((2 * (i + 0.5f) / nx - 1) * hw * u + (2 * (j + 0.5f) / ny - 1) * hh * v - w);
                    for (int b = 0; b < 3; b++) {
                        float t;
                        int hit = S.trace(o, d, t, c[b]);
                        if (hit < 0) break;
                        const rt_triangle148 &T = tris[hit];
                        V3 nrm{T.normal[0], T.normal[1], T.normal[2]};
                        if (dot(nrm, d) > 0) nrm = -1.0f * nrm;
                        V3 a = std::fabs(nrm.x) > 0.9f ? V3{0, 1, 0} : V3{1, 0, 0};
                        V3 tv = unit(cross(nrm, a)), tu = cross(nrm, tv);
                        float r1 = U(rng), r2 = U(rng), phi = 6.2831853f * r1, sr = std::sqrt(r2);
                        o = o + t * d;
                        d = unit(std::cos(phi) * sr * tu + std::sin(phi) * sr * tv + std::sqrt(1 - r2) * nrm);
                    }
                }
            const char *names[3] = {"camera", "bounce 1", "bounce 2"};
            for (int b = 0; b < 3; b++)
                std::printf("{\"width\": %d, \"leaf_size\": %d, \"wide_nodes\": %zu, \"rays\": \"%s\", \"n\": %.0f, \"hit_share\": %.3f, "
                            "\"node_visits_per_ray\": %.2f, \"box_tests_per_ray\": %.1f, \"prim_tests_per_ray\": %.2f, \"visits_entered_beyond_best_hit\": %.3f, \"visits_hitting_no_child\": %.3f}\n",
                            W, leaf, S.wn.size(), names[b], c[b].rays, c[b].hits / c[b].rays, c[b].visits / c[b].rays, c[b].box / c[b].rays,
                            c[b].prim / c[b].rays, c[b].cullable / c[b].visits, c[b].nohit / c[b].visits);
        }
    return 0;
}
