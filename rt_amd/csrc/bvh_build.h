// bvh_build.h -- bit-exact host builder for the reference's binary BVH.
//
// Restates make_bvh_nodes (reference src/pbrt/pbrt_engine.cu:335-368) together with the two
// toolchain pieces its output depends on, so that the tree does not change with the libc or the
// C++ library of the box it runs on:
//   * glibc rand() (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31]), which the
//     reference's drand48 wrapper calls (include/RT/triangle.h:12-15);
//   * libstdc++ std::sort (introsort: median-of-3 to first, unguarded partition, threshold 16,
//     heapsort fallback at depth 2*floor(log2 n), final insertion sort), whose order among equal
//     keys shapes the tree (ties are common: whole columns of a grid mesh share one min().x).
#pragma once
#include <cstdint>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/rt_abi.h"

namespace rtamd {

// glibc random_r TYPE_3 state, seeded like srand(seed).
class GlibcRand {
  public:
    explicit GlibcRand(uint32_t seed = 1);
    int32_t next(); // == rand()
  private:
    int32_t r_[34];
    int f_, b_; // front / rear indices
};

// std::sort(idx, idx+n, [&](a,b){ return key[a] < key[b]; }) restated.
void introsort_by_key(const float *key, int32_t *idx, int n);

// Padded bounding box of one primitive (Triangle::bounding_box, triangle.h:250-273).
void prim_box_padded(const rt_triangle148 &t, float mn[3], float mx[3]);
// Tight bounds (no pad).
void prim_box_tight(const rt_triangle148 &t, float mn[3], float mx[3]);

// Runs `left` on a thread of its own while the caller runs `right`; when the thread cannot be created (std::thread
// throws std::system_error when the process is out of threads) `left` runs on the caller instead.  The builders'
// recursions allocate nothing, so nothing else can throw between the spawn and the join.
template <class L, class R> inline void fork_join(L left, R right) {
    std::thread th;
    bool spawned = false;
    try {
        th = std::thread(left);
        spawned = true;
    } catch (const std::system_error &) {
    }
    if (!spawned) left();
    right();
    if (spawned) th.join();
}

// Builds the post-order node array (root last).  rand_calls_before advances the generator first.
void build_binary_bvh(const rt_triangle148 *tris, int n, int64_t rand_calls_before, std::vector<rt_bvh_node32> &out);

} // namespace rtamd
