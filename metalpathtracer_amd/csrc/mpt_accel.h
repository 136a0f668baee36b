// mpt_accel.h — host-side builder of the product's OWN acceleration structure (no HIP in here).
//
// The reference's closest hit (R/Renderer/Shaders/PathTracing.h:75-204) walks its BVH in a fixed order (right child
// first) with an int stack.  Because a child box always lies inside its parent's, that walk is equivalent to a linear
// scan over the LEAVES in visit order: "if the leaf's slab test passes with the best t so far, test its primitives in
// index order" — the inner nodes only skip work.  Hence the answer is the primitive with the smallest t (first in visit
// order on ties) whenever that primitive is CONSISTENT with its own leaf box (t >= the leaf's slab entry), which
// closest_hit_ordered() checks at the end; the rare other rays are re-traced in reference order (mpt_ordered.h).
//
// So the product is free to reach the leaves any way it likes.  This builder makes a 4-wide BVH over the reference
// LEAVES (every own leaf box contains the reference leaf box, so a ray that the reference would let into a leaf is
// let into it here too), binned SAH, laid out breadth-first so that the top of the tree can be staged in LDS:
//   node = 7 float4 (112 B: a stride that spreads 16 different nodes over all 64 LDS banks):
//     lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] child[4]
//     child: node index | 0x80000000 | (count-1) << 27 | first  (leaf: primitives [first, first+count) of the device
//            primitive array) | 0xFFFFFFFF (empty slot, box = +inf/-inf)
// Spheres (the reference puts them in leaves next to the root; their t suffers catastrophic cancellation for the
// r = 10^4 ground sphere, so no distance bound holds for them) are kept out of the tree on an "always" list that every
// ray tests first; the triangles that share a reference leaf with a sphere get a tight box of their own.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

namespace mpt_accel {

struct Box {
    float lo[3], hi[3];
};
static inline Box empty_box() { return Box{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}}; }
static inline void grow(Box& a, const Box& b) {
    for (int i = 0; i < 3; ++i) {
        a.lo[i] = std::min(a.lo[i], b.lo[i]);
        a.hi[i] = std::max(a.hi[i], b.hi[i]);
    }
}
static inline float half_area(const Box& b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// One item = one device leaf: <= 16 primitives that are contiguous in the device primitive array.
struct Item {
    Box box;          // own box (already padded)
    uint32_t count;   // primitives in the leaf (1..16)
    uint32_t key;     // caller's handle (position of the leaf in the caller's list)
};

// node stride in global memory, in float4: 7 = packed (112 B).  8 (one node = one 128-byte cache line; the LDS image keeps
// the 7-float4 stride either way) was measured on bunny x20: 66.3 ms against 65.1 — the lines it saves are already L1 /
// L2 hits (4.4 L2 requests per ray), the 14 % larger tree costs more.
#ifndef MPT_OT_NODE_STRIDE
#define MPT_OT_NODE_STRIDE 7u
#endif
#define MPT_ACCEL_NODE_FLOATS (4u * MPT_OT_NODE_STRIDE)
#define MPT_ACCEL_LEAF 0x80000000u
#define MPT_ACCEL_EMPTY 0xFFFFFFFFu
#define MPT_ACCEL_MAX_ALWAYS 16u

struct Tree {
    std::vector<float> nodes;          // MPT_ACCEL_NODE_FLOATS per node, breadth-first; node 0 = root (always exists)
    std::vector<uint32_t> item_order;  // items in the order their primitives must be laid out (breadth-first leaves)
    uint32_t depth = 0;
};

namespace detail {
struct BNode {
    Box b;
    int l, r, item;
};
struct Builder {
    const std::vector<Item>& items;
    std::vector<BNode> bn;
    std::vector<uint32_t> idx;
    explicit Builder(const std::vector<Item>& it) : items(it) {}
    int build(uint32_t lo, uint32_t hi) {
        BNode n;
        n.b = empty_box();
        n.l = n.r = n.item = -1;
        for (uint32_t i = lo; i < hi; ++i) grow(n.b, items[idx[i]].box);
        const int id = (int)bn.size();
        bn.push_back(n);
        if (hi - lo == 1) {
            bn[id].item = (int)idx[lo];
            return id;
        }
        Box cb = empty_box();
        for (uint32_t i = lo; i < hi; ++i) {
            const Box& b = items[idx[i]].box;
            Box c;
            for (int a = 0; a < 3; ++a) c.lo[a] = c.hi[a] = 0.5f * (b.lo[a] + b.hi[a]);
            grow(cb, c);
        }
        const int NB = 16;
        float best = INFINITY;
        int baxis = -1, bsplit = -1;
        for (int a = 0; a < 3; ++a) {
            const float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
            Box bb[NB];
            uint32_t bc[NB];
            for (int k = 0; k < NB; ++k) bb[k] = empty_box(), bc[k] = 0;
            for (uint32_t i = lo; i < hi; ++i) {
                const Item& it = items[idx[i]];
                int k = std::min(NB - 1, std::max(0, (int)(NB * ((0.5f * (it.box.lo[a] + it.box.hi[a]) - cb.lo[a]) / ext))));
                grow(bb[k], it.box);
                bc[k] += it.count;  // cost = primitives, not leaves
            }
            float ra[NB];
            Box acc = empty_box();
            uint32_t cnt = 0;
            for (int k = NB - 1; k > 0; --k) {
                grow(acc, bb[k]);
                cnt += bc[k];
                ra[k] = cnt ? half_area(acc) * (float)cnt : 0.0f;
            }
            acc = empty_box();
            cnt = 0;
            uint32_t total = 0;
            for (int k = 0; k < NB; ++k) total += bc[k];
            for (int k = 0; k < NB - 1; ++k) {
                grow(acc, bb[k]);
                cnt += bc[k];
                if (cnt == 0 || cnt == total) continue;
                const float c = half_area(acc) * (float)cnt + ra[k + 1];
                if (c < best) best = c, baxis = a, bsplit = k;
            }
        }
        uint32_t mid = (lo + hi) / 2;
        if (baxis >= 0) {
            const float ext = cb.hi[baxis] - cb.lo[baxis], base = cb.lo[baxis];
            auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t ii) {
                const Box& b = items[ii].box;
                int k = std::min(NB - 1, std::max(0, (int)(NB * ((0.5f * (b.lo[baxis] + b.hi[baxis]) - base) / ext))));
                return k <= bsplit;
            });
            const uint32_t m = (uint32_t)(it - idx.begin());
            if (m != lo && m != hi) mid = m;
        }
        const int l = build(lo, mid), r = build(mid, hi);
        bn[id].l = l;
        bn[id].r = r;
        return id;
    }
};
}  // namespace detail

// 4-wide tree over the items (binary binned SAH, then the child with the largest area is opened until four).
// first_of[item] must give the position of the item's first primitive AFTER the caller has laid the primitives out in
// tree.item_order — so the layout is produced in two steps: build_topology() then emit() with the final positions.
struct Topology {
    struct WNode {
        int child[4];  // >= 0: wide node (index into wn), < 0: ~item, INT32_MIN: empty
        Box cb[4];
    };
    std::vector<WNode> wn;  // breadth-first
    std::vector<uint32_t> item_order;
    uint32_t depth = 0;
};

static inline Topology build_topology(const std::vector<Item>& items) {
    Topology tp;
    if (items.empty()) {
        Topology::WNode n;
        for (int c = 0; c < 4; ++c) n.child[c] = INT32_MIN, n.cb[c] = empty_box();
        tp.wn.push_back(n);
        return tp;
    }
    detail::Builder B(items);
    B.idx.resize(items.size());
    for (uint32_t i = 0; i < items.size(); ++i) B.idx[i] = i;
    B.bn.reserve(items.size() * 2);
    const int root = B.build(0, (uint32_t)items.size());
    // breadth-first collapse
    struct Q {
        int bnode;
        uint32_t depth;
    };
    std::vector<Q> queue;
    queue.push_back({root, 0});
    if (B.bn[root].item >= 0) {  // a single item: a root node with one leaf child
        Topology::WNode n;
        for (int c = 0; c < 4; ++c) n.child[c] = INT32_MIN, n.cb[c] = empty_box();
        n.child[0] = ~B.bn[root].item;
        n.cb[0] = B.bn[root].b;
        tp.wn.push_back(n);
        tp.item_order.push_back((uint32_t)B.bn[root].item);
        tp.depth = 1;
        return tp;
    }
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const int b = queue[qi].bnode;
        const uint32_t d = queue[qi].depth;
        tp.depth = std::max(tp.depth, d + 1);
        int ch[4] = {B.bn[b].l, B.bn[b].r, -1, -1};
        int n = 2;
        while (n < 4) {
            int bi = -1;
            float ba = -1.0f;
            for (int i = 0; i < n; ++i)
                if (B.bn[ch[i]].item < 0 && half_area(B.bn[ch[i]].b) > ba) ba = half_area(B.bn[ch[i]].b), bi = i;
            if (bi < 0) break;
            const int c = ch[bi];
            ch[bi] = B.bn[c].l;
            ch[n++] = B.bn[c].r;
        }
        Topology::WNode w;
        for (int c = 0; c < 4; ++c) w.child[c] = INT32_MIN, w.cb[c] = empty_box();
        for (int c = 0; c < n; ++c) {
            w.cb[c] = B.bn[ch[c]].b;
            if (B.bn[ch[c]].item >= 0) {
                w.child[c] = ~B.bn[ch[c]].item;
                tp.item_order.push_back((uint32_t)B.bn[ch[c]].item);
            } else {
                w.child[c] = (int)queue.size();  // its breadth-first index = its queue position
                queue.push_back({ch[c], d + 1});
            }
        }
        tp.wn.push_back(w);
    }
    return tp;
}

// nodes in the device format; first_of[item] = position of the item's first primitive in the device array
static inline std::vector<float> emit(const Topology& tp, const std::vector<Item>& items, const std::vector<uint32_t>& first_of) {
    std::vector<float> out((size_t)tp.wn.size() * MPT_ACCEL_NODE_FLOATS, 0.0f);
    for (size_t i = 0; i < tp.wn.size(); ++i) {
        const Topology::WNode& w = tp.wn[i];
        float* o = out.data() + i * MPT_ACCEL_NODE_FLOATS;
        for (int c = 0; c < 4; ++c) {
            const bool empty = w.child[c] == INT32_MIN;
            for (int a = 0; a < 3; ++a) {
                // an empty slot is a box no walked ray can enter: both x planes at +inf put its x slab at [+inf, +inf] or
                // [-inf, -inf] (1/d is finite and non-zero for every ray the walk takes, mpt_ordered.h ot_degenerate), so the
                // device needs no "is there a child" test next to the slab test
                o[4 * a + c] = empty ? (a == 0 ? INFINITY : 0.0f) : w.cb[c].lo[a];
                o[12 + 4 * a + c] = empty ? (a == 0 ? INFINITY : 0.0f) : w.cb[c].hi[a];
            }
            uint32_t ref = MPT_ACCEL_EMPTY;
            if (w.child[c] >= 0) {
                ref = (uint32_t)w.child[c];
            } else if (w.child[c] != INT32_MIN) {
                const uint32_t it = (uint32_t)~w.child[c];
                ref = MPT_ACCEL_LEAF | ((items[it].count - 1u) << 27) | first_of[it];
            }
            memcpy(&o[24 + c], &ref, 4);
        }
    }
    return out;
}

}  // namespace mpt_accel
