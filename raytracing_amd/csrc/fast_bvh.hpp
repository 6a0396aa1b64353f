// fast_bvh.hpp -- host-side builder of the device's own bounding hierarchy ("fast tree").
//
// The reference's trees (sd::BVH::BuildBVHFromNodes, CPURayTrace/SimplifiedData.cpp:322-399, and
// BVH::BuildBVH, CPURayTrace/BVH.hpp:20-71) split at the median of the longest axis and start the box of
// every node with three or more leaves from FLT_MIN / zero, so their interior boxes can be far larger than
// what they hold.  The interior boxes only ever PRUNE: which primitive a ray ends on is decided by the leaf
// boxes, the primitive tests and the order in which leaves are met.  The fast tree therefore keeps the
// reference's leaf boxes bit for bit and puts a surface-area-heuristic hierarchy of exact unions above
// them; the kernel that walks it reproduces the reference's result from the leaves alone.
//
// Output: nodes in depth-first pre-order, each with the size of its subtree, so that "descend" is
// index + 1 and "prune" is index + size -- no traversal stack.
#pragma once

#include <algorithm>
#include <cstdint>
#include <future>
#include <vector>

namespace fastbvh {

struct Item { // one leaf of the reference tree
    float bmin[3], bmax[3];
    uint32_t id; // caller's handle
};

struct Node {
    float bmin[3], bmax[3];
    uint32_t size; // nodes in this subtree, this one included
    uint32_t item; // index into the item array for a leaf, 0xFFFFFFFF for an interior node
};

namespace detail {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; k++) {
            lo[k] = 3.402823466e+38f;
            hi[k] = -3.402823466e+38f;
        }
    }
    void grow(const float *mn, const float *mx) {
        for (int k = 0; k < 3; k++) {
            if (mn[k] < lo[k]) lo[k] = mn[k];
            if (mx[k] > hi[k]) hi[k] = mx[k];
        }
    }
    float half_area() const {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        if (!(x >= 0.f) || !(y >= 0.f) || !(z >= 0.f))
            return 0.f;
        return x * y + y * z + z * x;
    }
};

constexpr int kBins = 32;

constexpr uint32_t kParallelItems = 1u << 15; // ranges below this size are built by one thread
constexpr int kParallelDepth = 5;              // levels at which a range is shared between two threads

struct Builder {
    const std::vector<Item> &items;
    std::vector<uint32_t> idx; // disjoint ranges are permuted in place, so threads never share an element

    explicit Builder(const std::vector<Item> &it) : items(it) {}

    float centroid2(uint32_t i, int axis) const { return items[i].bmin[axis] + items[i].bmax[axis]; }

    // Chooses a split of idx[s, e) and partitions it; returns the first index of the second part (s < mid < e).
    uint32_t split(uint32_t s, uint32_t e) {
        const uint32_t n = e - s;
        float clo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
        float chi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        for (uint32_t i = s; i < e; i++)
            for (int k = 0; k < 3; k++) {
                const float c = centroid2(idx[i], k);
                if (c < clo[k]) clo[k] = c;
                if (c > chi[k]) chi[k] = c;
            }
        int best_axis = -1, best_bin = 0;
        float best_cost = 3.402823466e+38f;
        for (int axis = 0; axis < 3; axis++) {
            const float ext = chi[axis] - clo[axis];
            if (!(ext > 0.f))
                continue;
            const float scale = (float)kBins / ext;
            Box bb[kBins];
            uint32_t cnt[kBins];
            for (int b = 0; b < kBins; b++) {
                bb[b].reset();
                cnt[b] = 0;
            }
            for (uint32_t i = s; i < e; i++) {
                int b = (int)((centroid2(idx[i], axis) - clo[axis]) * scale);
                if (b >= kBins) b = kBins - 1;
                if (b < 0) b = 0;
                bb[b].grow(items[idx[i]].bmin, items[idx[i]].bmax);
                cnt[b]++;
            }
            float right_area[kBins];
            uint32_t right_cnt[kBins];
            Box acc;
            acc.reset();
            uint32_t c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                if (cnt[b]) acc.grow(bb[b].lo, bb[b].hi);
                c += cnt[b];
                right_area[b] = acc.half_area();
                right_cnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 0; b < kBins - 1; b++) { // split between bin b and b + 1
                if (cnt[b]) acc.grow(bb[b].lo, bb[b].hi);
                c += cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0)
                    continue;
                const float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = b;
                }
            }
        }
        if (best_axis >= 0) {
            const float ext = chi[best_axis] - clo[best_axis];
            const float scale = (float)kBins / ext;
            const float lo = clo[best_axis];
            const int ax = best_axis, bin = best_bin;
            auto mid_it = std::partition(idx.begin() + s, idx.begin() + e, [&](uint32_t i) {
                int b = (int)((centroid2(i, ax) - lo) * scale);
                if (b >= kBins) b = kBins - 1;
                if (b < 0) b = 0;
                return b <= bin;
            });
            const uint32_t mid = (uint32_t)(mid_it - idx.begin());
            if (mid > s && mid < e)
                return mid;
        }
        return s + n / 2; // all centroids coincide: any split is as good as another
    }

    void node_box(Node &nd, uint32_t s, uint32_t e) const {
        Box b;
        b.reset();
        for (uint32_t i = s; i < e; i++)
            b.grow(items[idx[i]].bmin, items[idx[i]].bmax);
        for (int k = 0; k < 3; k++) {
            nd.bmin[k] = b.lo[k];
            nd.bmax[k] = b.hi[k];
        }
    }

    // Orders idx[s, e) so that the smaller part of the split comes first; returns where the larger part starts.
    uint32_t split_small_first(uint32_t s, uint32_t e) {
        const uint32_t mid = split(s, e);
        if (mid - s <= e - mid)
            return mid;
        std::rotate(idx.begin() + s, idx.begin() + mid, idx.begin() + e);
        return s + (e - mid);
    }

    // Same nodes in the same order as build(), with the two parts of the top `depth` splits built by
    // different threads into vectors of their own and joined afterwards.
    void build_shared(uint32_t s, uint32_t e, std::vector<Node> &out, int depth) {
        if (depth <= 0 || e - s < kParallelItems) {
            build(s, e, out);
            return;
        }
        const size_t p = out.size();
        out.emplace_back();
        node_box(out[p], s, e);
        out[p].item = 0xFFFFFFFFu;
        const uint32_t mid = split_small_first(s, e);
        std::vector<Node> first;
        auto other = std::async(std::launch::async, [&] { build_shared(s, mid, first, depth - 1); });
        std::vector<Node> second;
        build_shared(mid, e, second, depth - 1);
        other.get();
        out.insert(out.end(), first.begin(), first.end());
        out.insert(out.end(), second.begin(), second.end());
        out[p].size = (uint32_t)(out.size() - p);
    }

    void build(uint32_t s, uint32_t e, std::vector<Node> &out) {
        std::vector<uint32_t> spine; // nodes whose subtree ends where this call's range ends
        for (;;) {
            const uint32_t p = (uint32_t)out.size();
            out.emplace_back();
            node_box(out[p], s, e);
            out[p].item = 0xFFFFFFFFu;
            out[p].size = 1;
            if (e - s == 1) {
                out[p].item = idx[s];
                // the leaf keeps the item's own box (bit for bit), not a recomputed one
                for (int k = 0; k < 3; k++) {
                    out[p].bmin[k] = items[idx[s]].bmin[k];
                    out[p].bmax[k] = items[idx[s]].bmax[k];
                }
                break;
            }
            spine.push_back(p);
            // The smaller part is emitted first through recursion (its size at least halves every time, so the
            // recursion depth stays logarithmic however lopsided the splits are); the larger part continues here.
            const uint32_t mid = split_small_first(s, e);
            build(s, mid, out);
            s = mid;
        }
        const uint32_t end = (uint32_t)out.size();
        for (uint32_t p : spine)
            out[p].size = end - p;
    }
};

} // namespace detail

// Appends the hierarchy over `items` to `out` (pre-order).  Every interior box is the exact union (plain
// float min/max) of the leaf boxes below it.  An empty item list appends nothing.
inline void build(const std::vector<Item> &items, std::vector<Node> &out) {
    if (items.empty())
        return;
    detail::Builder b(items);
    b.idx.resize(items.size());
    for (uint32_t i = 0; i < (uint32_t)items.size(); i++)
        b.idx[i] = i;
    if (items.size() < detail::kParallelItems)
        b.build(0, (uint32_t)items.size(), out);
    else
        b.build_shared(0, (uint32_t)items.size(), out, detail::kParallelDepth);
}

} // namespace fastbvh
