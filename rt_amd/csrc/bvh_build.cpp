// bvh_build.cpp -- see bvh_build.h.  Host code; compiled with -ffp-contract=off.
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>

namespace rtamd {

// ---------------------------------------------------------------- glibc rand()
// random_r.c, TYPE_3: degree 31, separation 3, 310 outputs discarded after seeding.
GlibcRand::GlibcRand(uint32_t seed) {
    if (seed == 0) seed = 1;
    r_[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        // 16807 * r mod (2^31 - 1) by Schrage's method, as glibc does to avoid overflow
        long hi = r_[i - 1] / 127773;
        long lo = r_[i - 1] % 127773;
        long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r_[i] = (int32_t)word;
    }
    f_ = 3;
    b_ = 0;
    for (int k = 0; k < 310; k++) (void)next();
}

int32_t GlibcRand::next() {
    uint32_t sum = (uint32_t)r_[f_] + (uint32_t)r_[b_];
    r_[f_] = (int32_t)sum;
    int32_t result = (int32_t)(sum >> 1);
    if (++f_ >= 31) {
        f_ = 0;
        ++b_;
    } else if (++b_ >= 31) {
        b_ = 0;
    }
    return result;
}

// ---------------------------------------------------------------- std::sort
namespace {

struct KeyLess {
    const float *key;
    bool operator()(int32_t a, int32_t b) const { return key[a] < key[b]; }
};

constexpr int kThreshold = 16;

inline void swap_i(int32_t &a, int32_t &b) {
    int32_t t = a;
    a = b;
    b = t;
}

// sift `value` up from hole towards top
void heap_push(int32_t *first, long hole, long top, int32_t value, KeyLess lt) {
    long parent = (hole - 1) / 2;
    while (hole > top && lt(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

void heap_adjust(int32_t *first, long hole, long len, int32_t value, KeyLess lt) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    heap_push(first, hole, top, value, lt);
}

void heap_sort_all(int32_t *first, int32_t *last, KeyLess lt) {
    // partial_sort(first, last, last): make_heap over the whole range, then sort_heap
    long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            int32_t value = first[parent];
            heap_adjust(first, parent, len, value, lt);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        int32_t value = *last;
        *last = *first;
        heap_adjust(first, 0, last - first, value, lt);
    }
}

inline void median_to_first(int32_t *result, int32_t *a, int32_t *b, int32_t *c, KeyLess lt) {
    if (lt(*a, *b)) {
        if (lt(*b, *c))
            swap_i(*result, *b);
        else if (lt(*a, *c))
            swap_i(*result, *c);
        else
            swap_i(*result, *a);
    } else if (lt(*a, *c))
        swap_i(*result, *a);
    else if (lt(*b, *c))
        swap_i(*result, *c);
    else
        swap_i(*result, *b);
}

inline int32_t *partition_unguarded(int32_t *first, int32_t *last, int32_t *pivot, KeyLess lt) {
    for (;;) {
        while (lt(*first, *pivot)) ++first;
        --last;
        while (lt(*pivot, *last)) --last;
        if (!(first < last)) return first;
        swap_i(*first, *last);
        ++first;
    }
}

void intro_loop(int32_t *first, int32_t *last, long depth_limit, KeyLess lt) {
    while (last - first > kThreshold) {
        if (depth_limit == 0) {
            heap_sort_all(first, last, lt);
            return;
        }
        --depth_limit;
        int32_t *mid = first + (last - first) / 2;
        median_to_first(first, first + 1, mid, last - 1, lt);
        int32_t *cut = partition_unguarded(first + 1, last, first, lt);
        intro_loop(cut, last, depth_limit, lt);
        last = cut;
    }
}

inline void linear_insert_unguarded(int32_t *last, KeyLess lt) {
    int32_t value = *last;
    int32_t *next = last - 1;
    while (lt(value, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = value;
}

void insertion_sort(int32_t *first, int32_t *last, KeyLess lt) {
    if (first == last) return;
    for (int32_t *i = first + 1; i != last; ++i) {
        if (lt(*i, *first)) {
            int32_t value = *i;
            std::memmove(first + 1, first, (size_t)(i - first) * sizeof(int32_t));
            *first = value;
        } else {
            linear_insert_unguarded(i, lt);
        }
    }
}

inline long floor_log2(long n) {
    long k = 0;
    while (n > 1) {
        n >>= 1;
        k++;
    }
    return k;
}

} // namespace

void introsort_by_key(const float *key, int32_t *idx, int n) {
    if (n <= 0) return;
    KeyLess lt{key};
    int32_t *first = idx, *last = idx + n;
    intro_loop(first, last, floor_log2(n) * 2, lt);
    if (last - first > kThreshold) {
        insertion_sort(first, first + kThreshold, lt);
        for (int32_t *i = first + kThreshold; i != last; ++i) linear_insert_unguarded(i, lt);
    } else {
        insertion_sort(first, last, lt);
    }
}

// ---------------------------------------------------------------- boxes
namespace {
const float kShift = (float)1e-2; // SHIFT (macros.h:5-7) after the vec3 ctor's conversion to float
inline float lo2(float a, float b) { return a < b ? a : b; } // ffmin, triangle.h:17
inline float hi2(float a, float b) { return a > b ? a : b; } // ffmax, triangle.h:21
} // namespace

void prim_box_tight(const rt_triangle148 &t, float mn[3], float mx[3]) {
    if (t.shapeType == 0) {
        for (int a = 0; a < 3; a++) {
            mn[a] = lo2(lo2(t.v0[a], t.v1[a]), t.v2[a]);
            mx[a] = hi2(hi2(t.v0[a], t.v1[a]), t.v2[a]);
        }
    } else {
        float r = t.v1[0];
        for (int a = 0; a < 3; a++) {
            mn[a] = t.v0[a] - r;
            mx[a] = t.v0[a] + r;
        }
    }
}

void prim_box_padded(const rt_triangle148 &t, float mn[3], float mx[3]) {
    prim_box_tight(t, mn, mx);
    for (int a = 0; a < 3; a++) { // aabb ctor, triangle.h:28-31
        mn[a] = mn[a] + (-kShift);
        mx[a] = mx[a] + kShift;
    }
}

// ---------------------------------------------------------------- make_bvh_nodes
namespace {

// Number of make_bvh_nodes invocations (= rand() calls) a subtree of cnt primitives makes.  It depends on cnt
// alone, so every subtree's position in the rand() sequence -- and, because a subtree of cnt primitives always emits
// 2*cnt-1 nodes, its position in the post-order node array -- is known before it is built.  That is what lets
// sibling subtrees be built concurrently with a bit-identical result ("Scale").
struct CallCount {
    std::unordered_map<int, int64_t> memo;
    int64_t operator()(int cnt) {
        if (cnt <= 2) return 1;
        auto it = memo.find(cnt);
        if (it != memo.end()) return it->second;
        int64_t v = 1 + (*this)(cnt / 2) + (*this)(cnt - cnt / 2);
        memo[cnt] = v;
        return v;
    }
    int64_t get(int cnt) const { // after the root's count has been evaluated every reachable size is in the memo
        if (cnt <= 2) return 1;
        return memo.find(cnt)->second;
    }
};

struct Builder {
    const float *key[3]; // padded box min per axis, per primitive (what X/Y/Z_CMP compare, triangle.h:394-434)
    const float *pmn, *pmx; // padded boxes, 3 floats per primitive
    std::vector<int32_t> order;
    rt_bvh_node32 *out = nullptr;
    std::vector<uint8_t> axis_of_call; // axis = int(3 * drand48()) for every rand() call of the build, in order
    int par_depth = 0;                 // subtrees above this depth are built on their own threads

    rt_bvh_node32 leaf(int32_t prim) const {
        rt_bvh_node32 nd;
        for (int a = 0; a < 3; a++) {
            nd.box_min[a] = pmn[3 * (size_t)prim + a];
            nd.box_max[a] = pmx[3 * (size_t)prim + a];
        }
        nd.left = prim;
        nd.right = prim;
        return nd;
    }
    rt_bvh_node32 parent(int32_t l, int32_t r) const {
        const rt_bvh_node32 &a = out[l], &b = out[r];
        rt_bvh_node32 nd;
        for (int k = 0; k < 3; k++) { // surrounding_box, triangle.h:76-85: union, then padded AGAIN by the ctor
            nd.box_min[k] = std::fmin(a.box_min[k], b.box_min[k]) + (-kShift);
            nd.box_max[k] = std::fmax(a.box_max[k], b.box_max[k]) + kShift;
        }
        nd.left = l;
        nd.right = r;
        return nd;
    }
    // make_bvh_nodes(_p, m, n, ...), pbrt_engine.cu:335-368.  call = index of this invocation's rand() call;
    // base = index of the first node this subtree emits.
    void go(int m, int n, int64_t call, int64_t base, int depth, const CallCount *calls) {
        int axis = axis_of_call[(size_t)call];
        const float *k = axis == 0 ? key[0] : (axis == 1 ? key[1] : key[2]); // axis 3 -> z, as the else branch
        introsort_by_key(k, order.data() + m, n - m);
        int cnt = n - m;
        if (cnt == 1) {
            out[base] = leaf(order[m]);
        } else if (cnt == 2) {
            out[base] = leaf(order[m]);
            out[base + 1] = leaf(order[m + 1]);
            out[base + 2] = parent((int32_t)base, (int32_t)base + 1);
        } else if (cnt != 0) {
            const int l = cnt / 2;
            const int64_t lcalls = calls->get(l);
            const int64_t rbase = base + 2 * (int64_t)l - 1;
            if (depth < par_depth) {
                fork_join([=]() { go(m, m + l, call + 1, base, depth + 1, calls); },
                          [=]() { go(m + l, n, call + 1 + lcalls, rbase, depth + 1, calls); });
            } else {
                go(m, m + l, call + 1, base, depth + 1, calls);
                go(m + l, n, call + 1 + lcalls, rbase, depth + 1, calls);
            }
            out[base + 2 * (int64_t)cnt - 2] = parent((int32_t)(rbase - 1), (int32_t)(base + 2 * (int64_t)cnt - 3));
        }
    }
};

} // namespace

void build_binary_bvh(const rt_triangle148 *tris, int n, int64_t rand_calls_before, std::vector<rt_bvh_node32> &out) {
    out.clear();
    if (n <= 0) return;
    std::vector<float> pmn(3 * (size_t)n), pmx(3 * (size_t)n);
    std::vector<float> kx(n), ky(n), kz(n);
    for (int i = 0; i < n; i++) {
        prim_box_padded(tris[i], &pmn[3 * (size_t)i], &pmx[3 * (size_t)i]);
        kx[i] = pmn[3 * (size_t)i + 0];
        ky[i] = pmn[3 * (size_t)i + 1];
        kz[i] = pmn[3 * (size_t)i + 2];
    }
    Builder b;
    CallCount calls;
    const int64_t total_calls = calls(n);
    {
        GlibcRand rng(1); // rand() is never seeded by the reference: default state == srand(1)
        for (int64_t i = 0; i < rand_calls_before; i++) (void)rng.next();
        b.axis_of_call.resize((size_t)total_calls);
        for (int64_t i = 0; i < total_calls; i++) {
            float dr = (float)rng.next() / (float)2147483647; // rand() / (float)RAND_MAX, triangle.h:12-15
            b.axis_of_call[(size_t)i] = (uint8_t)(int)(3 * dr);
        }
    }
    b.key[0] = kx.data();
    b.key[1] = ky.data();
    b.key[2] = kz.data();
    b.pmn = pmn.data();
    b.pmx = pmx.data();
    b.order.resize(n);
    for (int i = 0; i < n; i++) b.order[i] = i;
    out.resize(2 * (size_t)n - 1);
    b.out = out.data();
    b.par_depth = n >= (1 << 15) && std::thread::hardware_concurrency() > 1 ? 4 : 0; // 16 concurrent subtrees
    if (const char *e = std::getenv("RT_BUILD_PAR_DEPTH")) // 2^depth threads: held to 0..6
        b.par_depth = std::max(0, std::min(6, std::atoi(e)));
    b.go(0, n, 0, 0, 0, &calls);
}

} // namespace rtamd
