// sah_build.cpp -- see sah_build.h.  Top-down binned SAH (16 bins, all three axes), median fallback.
#include "sah_build.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <thread>

#include "bvh_build.h"

namespace rtamd {

namespace {

constexpr int kBins = 16;

struct Box3 {
    float mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            mn[a] = std::numeric_limits<float>::infinity();
            mx[a] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const Box3 &b) {
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], b.mn[a]);
            mx[a] = std::max(mx[a], b.mx[a]);
        }
    }
    void grow_point(const float *p) {
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], p[a]);
            mx[a] = std::max(mx[a], p[a]);
        }
    }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Sah {
    const std::vector<Box3> &box;
    const std::vector<float> &cen; // 3 per primitive
    std::vector<int32_t> idx;
    rt_bvh_node32 *out = nullptr;
    int par_depth = 0;

    Sah(const std::vector<Box3> &b, const std::vector<float> &c) : box(b), cen(c) {}

    int32_t emit_leaf(int32_t prim, int64_t base) {
        rt_bvh_node32 nd;
        for (int a = 0; a < 3; a++) {
            nd.box_min[a] = box[prim].mn[a];
            nd.box_max[a] = box[prim].mx[a];
        }
        nd.left = nd.right = prim;
        out[base] = nd;
        return (int32_t)base;
    }

    // A subtree over cnt primitives always emits 2*cnt-1 nodes (post-order: left subtree, right subtree, parent),
    // so its slots [base, base + 2*cnt - 1) are known up front and sibling subtrees can be built concurrently.
    int32_t build(int lo, int hi, int depth, int64_t base) {
        if (hi - lo == 1) return emit_leaf(idx[lo], base);
        Box3 cb, bb;
        cb.reset();
        bb.reset();
        for (int i = lo; i < hi; i++) {
            cb.grow_point(&cen[3 * (size_t)idx[i]]);
            bb.grow(box[idx[i]]);
        }
        int best_axis = -1, best_split = -1;
        float best_cost = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3 && depth < 48; a++) { // very deep chains: fall back to median splits (bounded depth)
            float ext = cb.mx[a] - cb.mn[a];
            if (!(ext > 0.0f)) continue;
            float scale = (float)kBins / ext;
            int cnt[kBins] = {0};
            Box3 bins[kBins];
            for (int b = 0; b < kBins; b++) bins[b].reset();
            for (int i = lo; i < hi; i++) {
                int p = idx[i];
                int b = (int)((cen[3 * (size_t)p + a] - cb.mn[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                cnt[b]++;
                bins[b].grow(box[p]);
            }
            float right_area[kBins];
            int right_cnt[kBins];
            Box3 acc;
            acc.reset();
            int c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc.grow(bins[b]);
                c += cnt[b];
                right_area[b] = acc.half_area();
                right_cnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 0; b < kBins - 1; b++) {
                acc.grow(bins[b]);
                c += cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_split = b;
                }
            }
        }
        int mid;
        if (best_axis >= 0) {
            float ext = cb.mx[best_axis] - cb.mn[best_axis];
            float scale = (float)kBins / ext;
            float mn = cb.mn[best_axis];
            int a = best_axis, s = best_split;
            auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](int32_t p) {
                int b = (int)((cen[3 * (size_t)p + a] - mn) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= s;
            });
            mid = (int)(it - idx.begin());
        } else {
            mid = lo; // all centroids coincide
        }
        if (mid == lo || mid == hi) { // fallback: median on the widest box axis (by primitive order if degenerate)
            int a = 0;
            float e = -1.0f;
            for (int k = 0; k < 3; k++) {
                float x = bb.mx[k] - bb.mn[k];
                if (x > e) {
                    e = x;
                    a = k;
                }
            }
            mid = lo + (hi - lo) / 2;
            std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int32_t p, int32_t q) {
                float cp = cen[3 * (size_t)p + a], cq = cen[3 * (size_t)q + a];
                return cp < cq || (cp == cq && p < q);
            });
        }
        const int64_t rbase = base + 2 * (int64_t)(mid - lo) - 1;
        int32_t l = 0, r = 0;
        if (depth < par_depth) {
            fork_join([&]() { l = build(lo, mid, depth + 1, base); }, [&]() { r = build(mid, hi, depth + 1, rbase); });
        } else {
            l = build(lo, mid, depth + 1, base);
            r = build(mid, hi, depth + 1, rbase);
        }
        rt_bvh_node32 nd;
        for (int a = 0; a < 3; a++) {
            nd.box_min[a] = bb.mn[a];
            nd.box_max[a] = bb.mx[a];
        }
        nd.left = l;
        nd.right = r;
        const int64_t self = base + 2 * (int64_t)(hi - lo) - 2;
        out[self] = nd;
        return (int32_t)self;
    }
};

} // namespace

void build_sah_bvh(const rt_triangle148 *tris, int n, std::vector<rt_bvh_node32> &out) {
    out.clear();
    if (n <= 0) return;
    std::vector<Box3> box(n);
    std::vector<float> cen(3 * (size_t)n);
    for (int i = 0; i < n; i++) {
        prim_box_tight(tris[i], box[i].mn, box[i].mx);
        for (int a = 0; a < 3; a++) cen[3 * (size_t)i + a] = 0.5f * (box[i].mn[a] + box[i].mx[a]);
    }
    Sah s(box, cen);
    s.idx.resize(n);
    for (int i = 0; i < n; i++) s.idx[i] = i;
    out.resize(2 * (size_t)n - 1);
    s.out = out.data();
    s.par_depth = n >= (1 << 15) && std::thread::hardware_concurrency() > 1 ? 4 : 0;
    s.build(0, n, 0, 0);
}

} // namespace rtamd
