// mpt_devbuild.h — build -> render without the host: everything mpt_upload_scene derives on the host from
// the reference's flat arrays (R/Scene/Scene.h:71-93,195-317 build, :99-167 packers) is derived here ON THE DEVICE from the
// radix tree of mpt_lbvh.h, and stays there:
//   * the threaded reference-order tree (mpt_device.h: hit link = the child the reference pops first, the RIGHT one,
//     PathTracing.h:188-193; miss link = the node after the subtree), breadth-first so that its top can be staged in LDS
//   * the primitive records in leaf order (48 B: v0, e1, e2 + reference leaf, material, original id), the de-duplicated
//     material table, the reference leaf boxes
//   * the product's own 4-wide tree (mpt_accel.h format) as a collapse of the SAME binary tree — the child with the largest
//     box is opened until four — with the spheres on the always list and kept out of the boxes
// The same tree in the reference's buffer format is kept on the device as well (mpt_download_bvh: what the oracle, or the
// reference's own shader, would walk), so parity is checked as for every other builder: oracle(downloaded arrays) == HIP.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <thread>

#include "mpt_accel.h"
#include "mpt_device.h"
#include "mpt_lbvh.h"
#include "mpt_radix.h"

namespace mpt_devbuild {
using mpt_lbvh::Radix;
using mpt_lbvh::Scratch;
using mpt_sah::SahState;
using mpt_sah::empty4;
using mpt_sah::half_area4;

struct Scalars {             // device-side results the host reads back once, at the end
    uint32_t n_spheres;      // spheres found (the first 32 positions are recorded)
    uint32_t sphere_pos[32]; // their positions in the device primitive array
    uint32_t tri_extent;     // bits of the largest finite |coordinate| of a triangle vertex
    uint32_t n_mats;
    uint32_t n_leaves;
    uint32_t n_acc_nodes, acc_depth;
    uint32_t odd_leaf;       // a leaf without a sphere has an empty (NaN) own box: the own tree is then refitted bottom-up (k_own_tree), not copied
    uint32_t mat_collision;  // two different materials with the same 32-bit sort key were seen (build() then sorts on all 64 bits)
};

__device__ __forceinline__ int span_of(const int2* range, int n, int node) { return node >= n - 1 ? 1 : range[node].y - range[node].x + 1; }
__device__ __forceinline__ int first_of(const int2* range, int n, int node) { return node >= n - 1 ? node - (n - 1) : range[node].x; }

// ---- materials: sort by a 64-bit hash, mark the runs, number them ---------------------------------------------------------
// keys: all 64 bits of the hash; keys32 (if not null): its upper half (>> key_shift: the tests force collisions by keeping a few bits only —
// the TOP ones, so that without a collision the order is still that of the whole hash)
__global__ void k_mat_hash(const float4* mats, uint32_t n, unsigned long long* keys, uint32_t* keys32, uint32_t key_shift, uint32_t* ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 a = ((const uint4*)mats)[2 * (size_t)i], b = ((const uint4*)mats)[2 * (size_t)i + 1];
    unsigned long long h = 0xcbf29ce484222325ull;
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    for (int k = 0; k < 8; ++k) {
        h ^= w[k];
        h *= 0x100000001b3ull;
        h ^= h >> 29;
    }
    if (keys32) keys32[i] = (uint32_t)(h >> 32) >> key_shift;
    else keys[i] = h;
    ids[i] = i;
}
// keys32_sorted (if not null): a run of equal 32-bit keys that holds two different materials may hold them interleaved — flagged
__global__ void k_mat_heads(const float4* mats, const uint32_t* ids, const uint32_t* keys32_sorted, uint32_t n, uint32_t* head, Scalars* sc) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    bool h = j == 0;
    if (!h) {
        const uint4* m = (const uint4*)mats;
        const uint4 a0 = m[2 * (size_t)ids[j]], a1 = m[2 * (size_t)ids[j] + 1], b0 = m[2 * (size_t)ids[j - 1]], b1 = m[2 * (size_t)ids[j - 1] + 1];
        h = a0.x != b0.x || a0.y != b0.y || a0.z != b0.z || a0.w != b0.w || a1.x != b1.x || a1.y != b1.y || a1.z != b1.z || a1.w != b1.w;
        if (h && keys32_sorted && keys32_sorted[j] == keys32_sorted[j - 1]) sc->mat_collision = 1u;
    }
    head[j] = h ? 1u : 0u;
}
__global__ void k_mat_scatter(const float4* mats, const uint32_t* ids, const uint32_t* head, const uint32_t* rank /* inclusive scan of head */,
                              uint32_t n, uint32_t* mat_of_prim, float4* table, Scalars* sc) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t id = rank[j] - 1u;
    mat_of_prim[ids[j]] = id;
    if (head[j]) {
        table[2 * (size_t)id] = mats[2 * (size_t)ids[j]];
        table[2 * (size_t)id + 1] = mats[2 * (size_t)ids[j] + 1];
    }
    if (j == n - 1) sc->n_mats = rank[j];
}

// ---- triangle extent (for the box padding) --------------------------------------------------------------------------------
// (grid-stride, one atomic per wave of a bounded grid, and only from a wave that can still raise the value: with a wave per 64 primitives
//  the 15,600 same-address atomics of a 1 M-primitive scene took 184 us; folded into mpt_lbvh.h k_boxes — one wave per 64 primitives
//  again — the guard's load of the hot word before the atomic made THAT kernel 14 -> 85 us.  30 us here, on the side stream, beside the tree build)
__global__ void k_tri_extent(const float4* prims, uint32_t n, Scalars* sc) {
    float m = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p0 = prims[3 * (size_t)i], p1 = prims[3 * (size_t)i + 1], p2 = prims[3 * (size_t)i + 2];
        if ((int)p0.w == 1) {
            const float v[9] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
            for (int k = 0; k < 9; ++k)
                if (isfinite(v[k])) m = fmaxf(m, fabsf(v[k]));
        }
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63u) == 0 && __float_as_uint(m) > __hip_atomic_load(&sc->tri_extent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&sc->tri_extent, __float_as_uint(m));
}

// ---- leaves: number them, lay their primitives out, their reference boxes, their own (sphere-free) boxes ----------------------
__global__ void k_leaf_flags(int n, int leaf_max, const int2* range, const uint32_t* keep, uint32_t* is_leaf) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1) return;
    is_leaf[node] = keep[node] && span_of(range, n, node) <= leaf_max ? 1u : 0u;
}
// Where a leaf's primitive records go in the device array.  Large scenes: the builder's order (neighbouring leaves next to each
// other: what the caches like).  Scenes the reference-order kernel renders (fewer than MPT_AUTO_ORDERED_PRIMS primitives) stage
// a PREFIX of that array in LDS — scene.xml: the tree and about a third of the primitives — so there the leaves a ray most
// probably enters come first: leaves of spheres only (tested by nearly every ray), then by falling box area (the SAH's own
// probability), the builder's order among equals.  scene.xml: 19.9 -> 19.1 ms per 256 spp.
__global__ void k_pfirst_builder_order(int n, const int2* range, const uint32_t* is_leaf, uint32_t* pfirst) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1 || !is_leaf[node]) return;
    pfirst[node] = (uint32_t)first_of(range, n, node);
}
__global__ void k_leaf_keys(int n, const int2* range, const uint32_t* is_leaf, const uint32_t* leaf_id, const uint32_t* vals, const float4* prims, const float4* nlo,
                            const float4* nhi, uint32_t* key, uint32_t* node_of) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1 || !is_leaf[node]) return;
    const int first = first_of(range, n, node), count = span_of(range, n, node);
    bool tri = false;
    for (int k = 0; k < count; ++k) tri = tri || (int)prims[3 * (size_t)vals[first + k]].w == 1;
    const float4 lo = nlo[node], hi = nhi[node];
    const float a = half_area4(lo, hi);
    // 12 bits of the area (exponent + 4 mantissa bits), large first; anything not a positive finite number last
    const uint32_t q = a > 0.0f && a < INFINITY ? __float_as_uint(a) >> 19 : 0u;
    const uint32_t leaf = leaf_id[node];
    key[leaf] = tri ? 1u + (0xFFFu - q) : 0u;
    node_of[leaf] = (uint32_t)node;
}
__global__ void k_leaf_counts(int n, uint32_t n_leaves, const int2* range, const uint32_t* node_sorted, uint32_t* cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_leaves) cnt[i] = (uint32_t)span_of(range, n, (int)node_sorted[i]);
}
__global__ void k_pfirst_scatter(uint32_t n_leaves, const uint32_t* node_sorted, const uint32_t* pos, uint32_t* pfirst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_leaves) pfirst[node_sorted[i]] = pos[i];
}
// one thread per output leaf.  own box of the leaf BEFORE the final padding: the reference leaf box; for a leaf that holds a
// sphere, a box around its triangles only (+5 % of their extent + pad, clipped to the leaf box) — empty if it has none
__global__ void k_leaves(int n, int leaf_max, const int2* range, const uint32_t* is_leaf, const uint32_t* leaf_id, const uint32_t* vals,
                         const float4* prims, const uint32_t* mat_of_prim, const float4* nlo, const float4* nhi, const uint32_t* pfirst, float4* dprims,
                         float4* refleaf, float4* olo, float4* ohi, Scalars* sc, int use_always_hint) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1 || !is_leaf[node]) return;
    const uint32_t leaf = leaf_id[node];
    const int first = first_of(range, n, node), count = span_of(range, n, node);
    const uint32_t at = pfirst[node];   // where the leaf's records go in the device primitive array
    const float4 lo = nlo[node], hi = nhi[node];
    refleaf[2 * (size_t)leaf] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    refleaf[2 * (size_t)leaf + 1] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    float tl[3] = {INFINITY, INFINITY, INFINITY}, th[3] = {-INFINITY, -INFINITY, -INFINITY};
    int ntri = 0, nsph = 0;
    for (int k = 0; k < count; ++k) {
        const uint32_t pid = vals[first + k];
        const float4 p0 = prims[3 * (size_t)pid], p1 = prims[3 * (size_t)pid + 1], p2 = prims[3 * (size_t)pid + 2];
        const int type = (int)p0.w;
        float4 r0, r1, r2;
        const uint32_t tag = (leaf << 1) | (type == 1 ? 1u : 0u);
        if (type == 1) {  // triangle: v0, e1 = v1 - v0, e2 = v2 - v0 (PathTracing.h:149-150), the subtraction the shader performs
            r0 = make_float4(p0.x, p0.y, p0.z, __uint_as_float(tag));
            r1 = make_float4(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, __uint_as_float(mat_of_prim[pid]));
            r2 = make_float4(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z, __uint_as_float(pid));
            const float v[3][3] = {{p0.x, p0.y, p0.z}, {p0.x + r1.x, p0.y + r1.y, p0.z + r1.z}, {p0.x + r2.x, p0.y + r2.y, p0.z + r2.z}};
            for (int q = 0; q < 3; ++q)
                for (int a = 0; a < 3; ++a) {
                    tl[a] = fminf(tl[a], v[q][a]);
                    th[a] = fmaxf(th[a], v[q][a]);
                }
            ntri++;
        } else {  // sphere; anything that is neither is never hit (PathTracing.h:120,143): kept as a sphere of radius NaN
            r0 = make_float4(p0.x, p0.y, p0.z, __uint_as_float(tag));
            r1 = make_float4(type == 0 ? p1.x : NAN, 0.0f, 0.0f, __uint_as_float(mat_of_prim[pid]));
            r2 = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(pid));
            nsph++;
            const uint32_t s = atomicAdd(&sc->n_spheres, 1u);
            if (s < 32u) sc->sphere_pos[s] = at + (uint32_t)k;
        }
        dprims[3 * (size_t)(at + k)] = r0;
        dprims[3 * (size_t)(at + k) + 1] = r1;
        dprims[3 * (size_t)(at + k) + 2] = r2;
    }
    float4 bl = make_float4(lo.x, lo.y, lo.z, 0.0f), bh = make_float4(hi.x, hi.y, hi.z, 0.0f);
    if (nsph != 0 && use_always_hint) {
        if (ntri == 0) {
            bl = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
            bh = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
        } else {
            const float pad = fmaxf(__uint_as_float(sc->tri_extent), 1e-6f) * 6.103515625e-05f;
            const float ext = fmaxf(th[0] - tl[0], fmaxf(th[1] - tl[1], th[2] - tl[2]));
            const float g = 0.05f * ext + pad;
            bl = make_float4(fmaxf(lo.x, tl[0] - g), fmaxf(lo.y, tl[1] - g), fmaxf(lo.z, tl[2] - g), 0.0f);
            bh = make_float4(fminf(hi.x, th[0] + g), fminf(hi.y, th[1] + g), fminf(hi.z, th[2] + g), 0.0f);
        }
    }
    olo[node] = bl;
    ohi[node] = bh;
    if (nsph == 0 && empty4(bl, bh)) sc->odd_leaf = 1u;
}
// ---- the threaded reference-order tree --------------------------------------------------------------------------------------
// depth of every output node and its skip link: the node the reference visits after this node's subtree.  The reference
// pops the right child first (PathTracing.h:188-193): after a RIGHT child's subtree comes its left sibling; a left child
// inherits its parent's skip link.
__global__ void k_depth_skip(int n, const uint32_t* keep, const uint32_t* index, const int2* child, const int* parent, uint32_t* depth_c /* by compact index */,
                             uint32_t* id_c, int* skip /* by id: node id or -1 = the end */) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1 || !keep[node]) return;
    uint32_t d = 0;
    int sk = -2;
    for (int y = node, p = parent[node]; p >= 0; y = p, p = parent[p]) {
        if (sk == -2 && y == child[p].y) sk = child[p].x;
        ++d;
    }
    skip[node] = sk == -2 ? -1 : sk;
    depth_c[index[node]] = d < 255u ? d : 255u;
    id_c[index[node]] = (uint32_t)node;
}
__global__ void k_positions(uint32_t n_out, const uint32_t* order /* breadth-first position -> id */, uint32_t* tpos /* by id */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) tpos[order[i]] = i;
}
__global__ void k_emit_threaded(uint32_t n_out, int n, const uint32_t* order, const uint32_t* tpos, const uint32_t* is_leaf, const int2* child,
                                const int2* range, const uint32_t* pfirst, const int* skip, const float4* nlo, const float4* nhi, float4* nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const int id = (int)order[i];
    const float4 lo = nlo[id], hi = nhi[id];
    const uint32_t next = skip[id] < 0 ? n_out : tpos[skip[id]];
    uint32_t A, B = next;
    if (is_leaf[id]) A = MPT_NODE_HOLD | (pfirst[id] * 16u + (uint32_t)(span_of(range, n, id) - 1));
    else A = tpos[child[id].y];   // box hit: the child the reference pops first
    nodes[2 * (size_t)i] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(A));
    nodes[2 * (size_t)i + 1] = make_float4(hi.x, hi.y, hi.z, __uint_as_float(B));
}

// ---- the binary tree UNDER the own 4-wide tree: top-down 16-bin SAH over the leaves, on the device ------------------------------
// Collapsing the Morton-order binary tree itself gives a poor 4-wide tree: on bunny x20 8.0 node visits per ray and 35 % of the
// rays let in by the root's four boxes, against 6.9 and 26 % for the host's top-down builder over the SAME leaves (12-15 %
// in render time; a bottom-up clustering sees 16 neighbours, the first Morton splits fall in empty space).  So the own tree
// gets the host's algorithm (mpt_accel.h: binned SAH on box centres, 16 bins, cost = primitives) run on the device: one
// WAVE per node, level by level — a node's items are binned with LDS atomics, the 45 candidate planes are priced by 45 lanes,
// the items are partitioned into the other of two index arrays.  The first levels are few long tasks (the root: one wave over
// all leaves, ~1.5 ms for 500 k), the later ones many short ones; ~25 launches in all.  Leaves = the reference leaves
// (k_leaves: own boxes, sphere-free); node ids 2n - 1 + k are this tree's inner nodes.
// Items of the own tree = the leaves whose own box is not empty, numbered by the position of their first primitive.
__global__ void k_item_flags(int n, const int2* range, const uint32_t* is_leaf, const float4* olo, const float4* ohi, uint32_t* flag_pos /* [n + 1] */) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1 || !is_leaf[node]) return;
    if (!empty4(olo[node], ohi[node])) flag_pos[first_of(range, n, node)] = 1u;
}
// item record, 32 bytes, moved along by every partition (the passes stream it, nothing is looked up through an index):
//   lo = (own box min, bits(leaf node id))   hi = (own box max, bits(primitives in the leaf))
__global__ void k_items(int n, const int2* range, const uint32_t* is_leaf, const float4* olo, const float4* ohi, const uint32_t* rank, float4* it_lo, float4* it_hi) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1 || !is_leaf[node]) return;
    const float4 l = olo[node], h = ohi[node];
    if (empty4(l, h)) return;
    const uint32_t i = rank[first_of(range, n, node)];
    it_lo[i] = make_float4(l.x, l.y, l.z, __int_as_float(node));
    it_hi[i] = make_float4(h.x, h.y, h.z, __int_as_float(span_of(range, n, node)));
}
// With the "sah" builder the binary tree under the reference-format arrays IS a binned-SAH tree over the primitives, and its
// nodes above the leaves are the own tree's binary tree already: no second SAH.  What differs is the boxes (the own box of a
// leaf leaves its spheres out, and may be empty) — so one bottom-up pass from the leaves refits them, and splices out what is
// empty: eff[node] = the node that stands for the sub-tree (the leaf; TOP + node for an inner node with two non-empty
// children; the other child's eff when one is empty; -1 when both are).
using mpt_lbvh::ld_agent;
using mpt_lbvh::st4;
using mpt_lbvh::st_agent;
// (hand-over between threads as in mpt_lbvh.h k_refit: agent-scope stores and loads, handoff_release() = an explicit
//  s_waitcnt vmcnt(0) in front of the arrival counter)
using mpt_lbvh::handoff_release;
__global__ void k_own_tree(int n, const int* parent, const int2* child, const uint32_t* is_leaf, const float4* olo, const float4* ohi, int* eff, float4* s_lo,
                           float4* s_hi, int2* s_child, int* arrived, SahState* st, const Scalars* skip_unless_odd) {
    const int leaf = blockIdx.x * blockDim.x + threadIdx.x, TOP = 2 * n - 1;
    if (leaf >= TOP || !is_leaf[leaf]) return;
    if (skip_unless_odd && !skip_unless_odd->odd_leaf) return;   // (k_own_copy + k_own_chain have made the tree)
    int e = empty4(olo[leaf], ohi[leaf]) ? -1 : leaf;
    st_agent(&eff[leaf], e);
    int cur = leaf;
    for (;;) {
        const int p = parent[cur];
        if (p < 0) {
            st->root = e;
            return;
        }
        handoff_release();
        if (atomicAdd(&arrived[p], 1) == 0) return;   // the sibling sub-tree is not finished yet
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int2 c = child[p];
        const int ex = ld_agent(&eff[c.x]), ey = ld_agent(&eff[c.y]);
        if (ex >= 0 && ey >= 0) {
            const float4 a0 = ex < TOP ? olo[ex] : mpt_lbvh::ld4(s_lo + (ex - TOP)), a1 = ex < TOP ? ohi[ex] : mpt_lbvh::ld4(s_hi + (ex - TOP));
            const float4 b0 = ey < TOP ? olo[ey] : mpt_lbvh::ld4(s_lo + (ey - TOP)), b1 = ey < TOP ? ohi[ey] : mpt_lbvh::ld4(s_hi + (ey - TOP));
            st4(s_lo + p, fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z));
            st4(s_hi + p, fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z));
            s_child[p] = make_int2(ex, ey);
            e = TOP + p;
        } else {
            e = ex >= 0 ? ex : ey;
        }
        st_agent(&eff[p], e);
        cur = p;
    }
}

// The common case of the above needs no walk (round 5; the walk is a chain of ~40 dependent atomics per leaf, 180 us for 1 M primitives):
// with the spheres hoisted, no leaf below the SAH's root holds one, so every such leaf's own box IS its reference box, and unless a leaf is
// empty (triangles with NaN corners: Scalars::odd_leaf) every inner node has two non-empty children — its box is the node box the builder
// already has (minima and maxima are exact, whatever the order), its children are its children.  k_own_copy writes that for every inner node
// at once; k_own_chain then redoes the root and the <= 15 chain nodes that hold the hoisted items — whose leaves ARE empty where they hold a
// sphere — with the rule of k_own_tree, one thread, top of the tree only.  Same arrays as the walk, entry for entry, where the collapse reads.
__global__ void k_own_copy(int n, const int2* child, const uint32_t* keep, const uint32_t* is_leaf, const float4* nlo, const float4* nhi, const Scalars* sc, float4* s_lo,
                           float4* s_hi, int2* s_child) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, TOP = 2 * n - 1;
    if (p >= n - 1 || sc->odd_leaf || !keep[p] || is_leaf[p]) return;
    const int2 c = child[p];
    const float4 l = nlo[p], h = nhi[p];
    s_lo[p] = make_float4(l.x, l.y, l.z, 0.0f);
    s_hi[p] = make_float4(h.x, h.y, h.z, 0.0f);
    s_child[p] = make_int2(is_leaf[c.x] ? c.x : TOP + c.x, is_leaf[c.y] ? c.y : TOP + c.y);
}
__global__ void k_own_chain(int n, int n_hoisted, const int2* child, const uint32_t* is_leaf, const float4* olo, const float4* ohi, const Scalars* sc, float4* s_lo,
                            float4* s_hi, int2* s_child, SahState* st) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || sc->odd_leaf) return;
    const int TOP = 2 * n - 1;
    // the root and the chain below it, top-down: node[0] = the root, node[j] = child[node[j - 1]].x while that is a chain node
    int node[17];
    int m = 0;
    node[m++] = 0;
    if (n_hoisted >= 1 && !is_leaf[0]) {
        int c = child[0].x;
        for (int j = 1; j < n_hoisted && !is_leaf[c]; ++j) {
            node[m++] = c;
            c = child[c].y;
        }
    }
    // the sub-tree that stands for a node (k_own_tree's eff): bottom-up over the chain; everything else is as k_own_copy left it
    auto eff_of = [&](int id, const int* eff_chain) -> int {
        if (is_leaf[id]) return empty4(olo[id], ohi[id]) ? -1 : id;
        for (int j = 0; j < m; ++j)
            if (node[j] == id) return eff_chain[j];
        return TOP + id;
    };
    int eff_chain[17];
    for (int j = m - 1; j >= 0; --j) {
        const int p = node[j];
        if (is_leaf[p]) {   // (a tree whose root is a leaf)
            eff_chain[j] = empty4(olo[p], ohi[p]) ? -1 : p;
            continue;
        }
        const int2 c = child[p];
        const int ex = eff_of(c.x, eff_chain), ey = eff_of(c.y, eff_chain);
        if (ex >= 0 && ey >= 0) {
            const float4 a0 = ex < TOP ? olo[ex] : s_lo[ex - TOP], a1 = ex < TOP ? ohi[ex] : s_hi[ex - TOP];
            const float4 b0 = ey < TOP ? olo[ey] : s_lo[ey - TOP], b1 = ey < TOP ? ohi[ey] : s_hi[ey - TOP];
            s_lo[p] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), 0.0f);
            s_hi[p] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), 0.0f);
            s_child[p] = make_int2(ex, ey);
            eff_chain[j] = TOP + p;
        } else {
            eff_chain[j] = ex >= 0 ? ex : ey;
        }
    }
    st->root = eff_chain[0];
}

// ---- the own 4-wide tree: breadth-first collapse of the SAH tree, level by level -----------------------------------------------
// Level by level (the wide tree of 1 M primitives has ~12): every node of the level picks its <= 4 children — the two
// children of its binary node, the one with the largest box opened until four — and counts the inner ones (k_collapse_pick);
// an exclusive scan of those counts numbers the next level in order (deterministic layout); k_collapse_emit writes the nodes.
// Node ids >= 2n - 1 are the inner nodes of the SAH tree, ids below are its leaves = the reference leaves.
struct CollapseAcc {
    const int2* s_child;
    const float4 *s_lo, *s_hi, *olo_b, *ohi_b;
    int TOP;
    __device__ __forceinline__ bool is_leaf(int id) const { return id < TOP; }
    __device__ __forceinline__ int2 child(int id) const { return s_child[id - TOP]; }
    __device__ __forceinline__ float4 lo(int id) const { return id < TOP ? olo_b[id] : s_lo[id - TOP]; }
    __device__ __forceinline__ float4 hi(int id) const { return id < TOP ? ohi_b[id] : s_hi[id - TOP]; }
};
// the <= 4 children of the wide node that stands on binary node b (-1: none), and how many of them are inner nodes
__device__ __forceinline__ int4 collapse_pick(const CollapseAcc& A, int b, uint32_t* nint_out) {
    int ch[4] = {-1, -1, -1, -1};
    int nc = 0;
    if (b >= 0) {
        if (A.is_leaf(b)) {  // (only the root can be a leaf: a tree of one leaf)
            ch[nc++] = b;
        } else {
            const int2 c = A.child(b);
            ch[nc++] = c.x;
            ch[nc++] = c.y;
            for (int round = 0; round < 2 && nc < 4; ++round) {   // open the inner child with the largest box
                int bi = -1;
                float ba = -1.0f;
                for (int i = 0; i < nc; ++i)
                    if (!A.is_leaf(ch[i])) {
                        const float a = half_area4(A.lo(ch[i]), A.hi(ch[i]));
                        if (a > ba) ba = a, bi = i;
                    }
                if (bi < 0) break;
                const int2 g = A.child(ch[bi]);
                if (g.x < 0 || g.y < 0) break;   // (a node nobody made: cannot be reached from the root either — k_collapse_picks prices every node)
                ch[bi] = g.x;
                ch[nc++] = g.y;
            }
        }
    }
    uint32_t nint = 0;
    for (int i = 0; i < nc; ++i) nint += A.is_leaf(ch[i]) ? 0u : 1u;
    *nint_out = nint;
    return make_int4(ch[0], ch[1], ch[2], ch[3]);
}
// Round 5: the picks of ALL inner nodes of the binary tree at once, before the level loop.  A pick is a chain of five dependent loads (the
// children, their boxes, the opened child's children, their boxes, ...); made by the thread that emits the parent, as in the first half of
// the round, that chain was the duration of every level (17-24 us each, thirteen levels); here half a million of them overlap.  Only about
// half of the nodes become wide nodes, the others' picks are never read.
// dense: the tree is run_sah's (inner nodes 0 .. n_nodes - 1, all made); else the refitted one, whose unmade nodes hold child -1.
__global__ void k_collapse_picks(CollapseAcc A, uint32_t n_bound, const SahState* st, int dense, int4* picks, uint32_t* pn) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_bound) return;
    if (dense ? k >= st->n_nodes : A.s_child[k].x < 0 || A.s_child[k].y < 0) return;
    picks[k] = collapse_pick(A, A.TOP + (int)k, pn + k);
}
// No host wait inside the level loop (round 5; until round 4 the host read the size of the next level back after every level: 12 idle gaps
// of ~20 us for 1 M primitives, and three more launches a level).  The levels' bounds live on the device (lev[i] = first node, end), the
// grids are sized from an upper bound (a level has at most four times the nodes of the one before), a level is one scan and one kernel.  The
// size of the next level goes to a slot of pinned host memory with a stamp; the host reads it a level late, to end the loop (mpt_sah.h
// run_sah does the same).  The first levels — while a level has at most 1024 nodes — are made by ONE workgroup in one launch
// (k_collapse_top: six levels of a 1 M-primitive tree), which leaves the first larger level's nodes for the loop.
struct CollapseLevel {
    uint32_t begin, end;
};
struct CollapseHead {
    uint32_t depth_base;   // levels made by k_collapse_top
    uint32_t pad[3];
    CollapseLevel lev[132];   // lev[i]: the nodes of the loop's i-th level
};
// writes wide node k, whose children are p; its inner children are nodes at, at + 1, ...: their binary nodes go to w_next[0 ..], the number
// of inner children of each to n_next[0 ..] (both already offset to this node's first child)
__device__ __forceinline__ void collapse_emit(const CollapseAcc& A, int n, const int2* range, const uint32_t* pfirst, const uint32_t* pn, int4 p, uint32_t k, uint32_t at,
                                              uint32_t cap, float pad, float4* acc_nodes, int* w_next, uint32_t* n_next) {
    const int ch[4] = {p.x, p.y, p.z, p.w};
    float o[4 * MPT_OT_NODE_STRIDE];
    for (uint32_t q = 0; q < 4u * MPT_OT_NODE_STRIDE; ++q) o[q] = 0.0f;
    uint32_t j = 0;
    for (int c = 0; c < 4; ++c) {
        uint32_t ref = MPT_ACCEL_EMPTY;
        float lo[3] = {INFINITY, 0.0f, 0.0f}, hi[3] = {INFINITY, 0.0f, 0.0f};   // empty slot: a box no walked ray enters (mpt_accel.h emit)
        if (ch[c] >= 0) {
            const float4 l = A.lo(ch[c]), h = A.hi(ch[c]);
            lo[0] = l.x - pad; lo[1] = l.y - pad; lo[2] = l.z - pad;
            hi[0] = h.x + pad; hi[1] = h.y + pad; hi[2] = h.z + pad;
            if (A.is_leaf(ch[c])) {
                ref = MPT_ACCEL_LEAF | ((uint32_t)(span_of(range, n, ch[c]) - 1) << 27) | pfirst[ch[c]];
            } else {
                ref = at + j;
                if (at + j < cap) {
                    w_next[j] = ch[c];
                    if (n_next) n_next[j] = pn[ch[c] - A.TOP];
                }
                ++j;
            }
        }
        for (int a = 0; a < 3; ++a) {
            o[4 * a + c] = lo[a];
            o[12 + 4 * a + c] = hi[a];
        }
        o[24 + c] = __uint_as_float(ref);
    }
    float4* dst = acc_nodes + (size_t)k * MPT_OT_NODE_STRIDE;
    for (uint32_t q = 0; q < MPT_OT_NODE_STRIDE; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}
#define MPT_COLLAPSE_TOP 1024u   // k_collapse_top makes the levels of at most this many nodes
__global__ __launch_bounds__(MPT_COLLAPSE_TOP) void k_collapse_top(CollapseAcc A, int n, const int2* range, const uint32_t* pfirst, const SahState* st, const int4* picks,
                                                                   const uint32_t* pn, CollapseHead* head, int* w_out, uint32_t* n_out, float4* acc_nodes, uint32_t cap,
                                                                   Scalars* sc) {
    __shared__ int s_w[2][4u * MPT_COLLAPSE_TOP];   // the binary nodes of the level, and of the next one (at most four times as many)
    __shared__ uint32_t s_wave[MPT_COLLAPSE_TOP / 64u];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const float pad = fmaxf(__uint_as_float(sc->tri_extent), 1e-6f) * 6.103515625e-05f;  // 2^-14: covers the rcp / fma slab arithmetic
    if (tid == 0) s_w[0][0] = st->root;
    uint32_t begin = 0, end = 1, cur = 0, depth = 0;
    __syncthreads();
    while (end - begin != 0u && end - begin <= MPT_COLLAPSE_TOP) {
        const uint32_t cnt = end - begin;
        int4 p = make_int4(-1, -1, -1, -1);
        uint32_t nint = 0;
        if (tid < cnt) {
            const int wb = s_w[cur][tid];
            if (wb >= A.TOP) {
                p = picks[wb - A.TOP];
                nint = pn[wb - A.TOP];
            } else {
                p.x = wb;   // (the root is a leaf, or there is no tree: -1)
            }
        }
        // exclusive scan of nint over the workgroup
        uint32_t x = nint;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off);
            if ((int)lane >= off) x += y;
        }
        if (lane == 63u) s_wave[wv] = x;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < MPT_COLLAPSE_TOP / 64u; ++w) {
            if (w < wv) before += s_wave[w];
            total += s_wave[w];
        }
        const uint32_t first = before + x - nint;
        if (tid < cnt && begin + tid < cap) collapse_emit(A, n, range, pfirst, pn, p, begin + tid, end + first, cap, pad, acc_nodes, &s_w[cur ^ 1u][first], nullptr);
        __syncthreads();
        ++depth;
        begin = end;
        end += total;
        cur ^= 1u;
    }
    // what is left is the loop's: its first level's nodes, their counts of inner children
    const uint32_t cnt = end - begin;
    for (uint32_t i = tid; i < cnt && begin + i < cap; i += MPT_COLLAPSE_TOP) {
        const int wb = s_w[cur][i];
        w_out[i] = wb;
        n_out[i] = pn[wb - A.TOP];
    }
    if (tid == 0) {
        head->depth_base = depth;
        head->lev[0] = CollapseLevel{begin, end};
        sc->n_acc_nodes = begin;
        sc->acc_depth = depth;
    }
}
__global__ void k_collapse_level(CollapseAcc A, int n, const int2* range, const uint32_t* pfirst, CollapseHead* head, uint32_t L, const int4* picks, const uint32_t* pn,
                                 const int* wbin, const uint32_t* offs, int* w_next, uint32_t* n_next, float4* acc_nodes, uint32_t cap, Scalars* sc,
                                 unsigned long long* host_slot, uint32_t stamp) {
    const uint32_t begin = head->lev[L].begin, end = head->lev[L].end, cnt = end - begin;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        const uint32_t total = cnt ? offs[cnt] : 0u;   // inner children of the level = the nodes of the next one
        head->lev[L + 1u] = CollapseLevel{end, end + total};
        if (cnt) {
            sc->n_acc_nodes = end;
            sc->acc_depth = head->depth_base + L + 1u;
        }
        // (ONE 8-byte store: no system-scope fence — a write-back of the L2 — beside the level's other workgroups; mpt_sah.h sah_level_prologue)
        __hip_atomic_store(host_slot, (unsigned long long)stamp << 32 | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t k = begin + t;
    if (t >= cnt || k >= cap) return;
    const float pad = fmaxf(__uint_as_float(sc->tri_extent), 1e-6f) * 6.103515625e-05f;  // 2^-14: covers the rcp / fma slab arithmetic
    const uint32_t first = offs[t];
    collapse_emit(A, n, range, pfirst, pn, picks[wbin[t] - A.TOP], k, end + first, cap, pad, acc_nodes, w_next + first, n_next + first);
}

// refbox[2 i], [2 i + 1] = the box of primitive i's reference leaf, what ot_final_check (mpt_ordered.h) tests the winner against:
// indexed by the primitive, so that the check is ONE memory round trip (not the primitive record first and the leaf's box after it)
__global__ void k_prim_refbox(const float4* dprims, const float4* refleaf, uint32_t n_prims, float4* refbox) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_prims) return;
    const uint32_t leaf = prim_ref_leaf(dprims[3 * (size_t)i]);
    refbox[2 * (size_t)i] = refleaf[2 * (size_t)leaf];
    refbox[2 * (size_t)i + 1] = refleaf[2 * (size_t)leaf + 1];
}

// ---- the always list (<= 16 spheres): 5 float4 each, in the order of the device primitive array ---------------------------------
__global__ void k_always(Scalars* sc, float4* dprims, const float4* refleaf, float4* always) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t ns = sc->n_spheres < 32u ? sc->n_spheres : 32u;
    for (uint32_t i = 1; i < ns; ++i) {   // (found in atomic order: sort by position)
        const uint32_t v = sc->sphere_pos[i];
        uint32_t j = i;
        for (; j > 0 && sc->sphere_pos[j - 1] > v; --j) sc->sphere_pos[j] = sc->sphere_pos[j - 1];
        sc->sphere_pos[j] = v;
    }
    if (sc->n_spheres > MPT_ACCEL_MAX_ALWAYS) return;   // too many: the closest-first pipeline is not used for this scene
    for (uint32_t k = 0; k < ns; ++k) {
        const uint32_t i = sc->sphere_pos[k];
        float4 r0 = dprims[3 * (size_t)i], r1 = dprims[3 * (size_t)i + 1];
        const float4 r2 = dprims[3 * (size_t)i + 2];
        const uint32_t leaf = __float_as_uint(r0.w) >> 1;
        r1.z = __uint_as_float(k);             // the primitive record points at its always-list entry ...
        dprims[3 * (size_t)i + 1] = r1;
        r1.y = __uint_as_float(i);             // ... and the entry at the primitive (what the walk reports as the winner)
        always[5 * (size_t)k] = r0;
        always[5 * (size_t)k + 1] = r1;
        always[5 * (size_t)k + 2] = r2;
        always[5 * (size_t)k + 3] = refleaf[2 * (size_t)leaf];
        always[5 * (size_t)k + 4] = refleaf[2 * (size_t)leaf + 1];
    }
}

// Everything above, in order, on one stream.  Inputs: the caller's primitive and material arrays, ALREADY on the device
// (d_prims_in: 3 float4 per primitive, d_mats_in: 2 float4 per primitive).  All outputs are hipMalloc'ed here and handed to
// the caller (who frees them); scratch is freed on return.
struct Built {
    // views into `block` (ONE allocation, sized from upper bounds before the first kernel: n_out <= 2n - 1 nodes, <= n leaves, <= n + 2 wide
    // nodes, <= n materials — ~390 bytes per primitive; until round 4 nine hipMallocs of the exact sizes sat on the build's critical path)
    float4 *nodes = nullptr, *prims = nullptr, *mats = nullptr, *acc_nodes = nullptr, *refleaf = nullptr, *refbox = nullptr, *always = nullptr;
    float4* ref_bvh = nullptr;   // the same tree in the reference's buffer format (2 float4 per node) ...
    int* ref_idx = nullptr;      // ... and its primitiveIndices
    void* block = nullptr;
    size_t block_bytes = 0;
    uint32_t n_nodes = 0, n_prims = 0, n_mats = 0, n_acc_nodes = 0, n_always = 0, n_ref_leaves = 0, acc_depth = 0, n_spheres = 0;
    float tri_extent = 0.0f;
    void release() {
        hipFree(block);
        *this = Built{};
    }
};
static size_t out_block_bytes(uint32_t n) {
    const size_t nn = 2 * (size_t)n - 1, al = 256;
    const size_t parts[9] = {nn * 32, (size_t)n * 4, (size_t)n * 48, (size_t)n * 32, nn * 32, ((size_t)n + 2) * MPT_OT_NODE_STRIDE * 16, (size_t)MPT_ACCEL_MAX_ALWAYS * 80,
                             (size_t)n * 32, (size_t)n * 32};
    size_t total = 0;
    for (size_t b : parts) total += (b + al - 1) & ~(al - 1);
    return total;
}

// spare / spare_bytes (in, out): a device allocation the caller has no more use for — taken for the outputs if it is large enough.
// mats_host (may be null): the caller's material array in host memory, NOT yet copied to d_mats_in — the copy is then made here, on the side
// stream by a helper thread (a copy from pageable memory keeps its host thread for its whole length: 1.1 ms for the 32 MB of 1 M
// primitives), while this thread drives the tree build on the main stream.
static hipError_t build_pass(hipStream_t stream, float4* d_prims_in, float4* d_mats_in, const float* mats_host, uint32_t n, int leaf_max, int builder,
                             uint32_t n_spheres_hint, Built& out, mpt_lbvh::ScratchPool* pool, void** spare, size_t* spare_bytes, bool wide_mat_sort, bool* mat_collision) {
    Scratch sc(pool);
    // Two streams (round 5): the material chain (hash, sort, table: ~0.19 ms of small kernels for 1 M primitives) needs nothing of the tree until
    // the leaves are written, and the threaded tree (depths, their sort, the emission: ~0.13 ms) nothing of the own tree's collapse — each runs
    // on the pool's side stream beside the main one, forked and joined by events.  Without a pool: everything on `stream`, as before.
    hipStream_t side = stream;
    if (pool && getenv("MPT_BUILD_ONE_STREAM") == nullptr) MPT_LB(pool->side_stream(&side));
    const bool two = side != stream;
    struct SideGuard {   // (no scratch chunk goes back to the pool while the side stream may still use it: declared after `sc`, run before it)
        hipStream_t s;
        ~SideGuard() { if (s) hipStreamSynchronize(s); }
    } side_guard{two ? side : nullptr};
    struct Helper {   // (joined before the side stream is drained and the scratch goes back: declared after both)
        std::thread t;
        hipError_t err = hipSuccess;
        ~Helper() { if (t.joinable()) t.join(); }
    } helper;
    auto hand_over = [&](hipStream_t from, hipStream_t to, int e) -> hipError_t {   // what `to` does next comes after what `from` has been given so far
        if (!two) return hipSuccess;
        MPT_LB(hipEventRecord(pool->ev[e], from));
        return hipStreamWaitEvent(to, pool->ev[e], 0);
    };
    {
        const size_t need = out_block_bytes(n);
        if (spare && *spare && *spare_bytes >= need) {
            out.block = *spare;
            out.block_bytes = *spare_bytes;
            *spare = nullptr;
            *spare_bytes = 0;
        } else {
            MPT_LB(hipMalloc(&out.block, need));
            out.block_bytes = need;
        }
        char* q = (char*)out.block;
        auto carve = [&](size_t bytes) {
            char* r = q;
            q += (bytes + 255) & ~(size_t)255;
            return r;
        };
        const size_t nn_ = 2 * (size_t)n - 1;
        out.ref_bvh = (float4*)carve(nn_ * 32);
        out.ref_idx = (int*)carve((size_t)n * 4);
        out.prims = (float4*)carve((size_t)n * 48);
        out.refleaf = (float4*)carve((size_t)n * 32);
        out.nodes = (float4*)carve(nn_ * 32);
        out.acc_nodes = (float4*)carve(((size_t)n + 2) * MPT_OT_NODE_STRIDE * 16);
        out.always = (float4*)carve((size_t)MPT_ACCEL_MAX_ALWAYS * 80);
        out.refbox = (float4*)carve((size_t)n * 32);
        out.mats = (float4*)carve((size_t)n * 32);
    }
    MPT_LB(sc.reserve((size_t)n * 720 + ((size_t)8 << 20)));   // (measured: ~620 bytes per primitive)
    Radix R;
    Scalars* d_sc;
    mpt_lbvh::PinnedWords pinned;   // read-backs of a few words go through pinned memory (a pageable target costs ~0.3 ms per copy)
    MPT_LB(pinned.get(pool, 0));    // (words 64 .. 111: the level slots of mpt_sah::run_sah, 160 .. 175: the collapse's, 192 ..: the scalars read at the end)
    uint32_t* pin = pinned.p;
    MPT_LB(sc.alloc(&d_sc, 1));
    MPT_LB(hipMemsetAsync(d_sc, 0, sizeof(Scalars), stream));
    MPT_LB(hand_over(stream, side, 0));   // (the caller's uploads and the zeroed scalars)
    const uint32_t B = 256, gn = (n + B - 1) / B;
    // materials
    unsigned long long *mk, *mk2;
    uint32_t *mi, *mi2, *mhead, *mrank, *mat_of_prim;
    float4* const mtable = out.mats;   // (the de-duplicated table is written where it stays)
    mk = mk2 = nullptr;
    if (wide_mat_sort) {
        MPT_LB(sc.alloc(&mk, n));
        MPT_LB(sc.alloc(&mk2, n));
    }
    MPT_LB(sc.alloc(&mi, n));
    MPT_LB(sc.alloc(&mi2, n));
    MPT_LB(sc.alloc(&mhead, n));
    MPT_LB(sc.alloc(&mrank, n));
    MPT_LB(sc.alloc(&mat_of_prim, n));
    // Sorted on the upper 32 bits of the hash, as 32-bit keys, by four passes of the builders' own radix sort (mpt_radix.h: hipcub sorts
    // 1 M pairs by merging, 25 launches and ~0.2 ms whatever the key width).  The order is that of the full 64-bit sort unless two DIFFERENT materials share those 32 bits —
    // k_mat_heads sees that (equal keys, other contents) and the caller builds again with wide_mat_sort (build() below).
    // [hipcub::DeviceRadixSort::SortPairs over bits [32, 64) of the 64-bit keys themselves — the obvious way to do this — is WRONG on this
    // toolchain (ROCm 7.2.0, gfx950) from a few thousand items on: the keys it returns are not even a permutation of its input
    // (tests/experiments/hipcub_partial_bits.hip, run on MI355X: 4,971 / 99,362 / 1,000,003 items, its own queried temporary size, canaries
    // behind every buffer intact, input untouched; 5 items: right).  That is what round 4's experiment hit: material ids from garbage keys
    // — images off in 1.5 % of the pixels, and once ids far outside the table, a memory fault inside this build (gpurun_out/r04/
    // s46_tests.log).  The same program finds every range the product uses right: [0, 64), [0, 63) (mpt_lbvh.h), [0, 13), [0, 8) on
    // 32-bit keys (docs/HISTORY.md); full-width 32-bit keys are checked by the recorded digests, tests/golden/devbuild_digests.json.]
    uint32_t *mk32 = nullptr, *mk32s = nullptr;
    if (!wide_mat_sort) {
        MPT_LB(sc.alloc(&mk32, n));
        MPT_LB(sc.alloc(&mk32s, n));
    }
    uint32_t key_shift = 0u;
    if (const char* e = getenv("MPT_DEBUG_MAT_KEY_BITS")) key_shift = 32u - (uint32_t)std::min(std::max(atoi(e), 1), 32);
    // (everything the chain needs is allocated here, by this thread: the scratch allocator is not for two)
    char *sort_tmp = nullptr, *scan_tmp = nullptr;
    size_t sort_bytes = 0, scan_bytes = 0;
    mpt_radix::RadixTemp RT;
    if (wide_mat_sort) {
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, mk, mk2, mi, mi2, (int)n, 0, 64, side));
        MPT_LB(sc.alloc(&sort_tmp, sort_bytes));
    } else {
        MPT_LB(mpt_radix::radix_reserve(sc, n, side, RT));
    }
    MPT_LB(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, mhead, mrank, (int)n, side));
    MPT_LB(sc.alloc(&scan_tmp, scan_bytes));
    auto mats_chain = [&]() -> hipError_t {
        hipLaunchKernelGGL(k_tri_extent, dim3(std::min(gn, 1024u)), dim3(B), 0, side, (const float4*)d_prims_in, n, d_sc);   // (needed by k_leaves: joins with the materials)
        hipLaunchKernelGGL(k_mat_hash, dim3(gn), dim3(B), 0, side, (const float4*)d_mats_in, n, mk, mk32, key_shift, mi);
        const uint32_t *ids_sorted = mi2, *keys32_sorted = nullptr;
        if (wide_mat_sort) {
            size_t bytes = sort_bytes;
            MPT_LB(hipcub::DeviceRadixSort::SortPairs(sort_tmp, bytes, mk, mk2, mi, mi2, (int)n, 0, 64, side));
        } else {
            bool second = false;
            MPT_LB(mpt_radix::radix_sort_pairs(side, RT, mk32, mi, mk32s, mi2, n, 4, &second));
            ids_sorted = second ? mi2 : mi;
            keys32_sorted = second ? mk32s : mk32;
        }
        hipLaunchKernelGGL(k_mat_heads, dim3(gn), dim3(B), 0, side, (const float4*)d_mats_in, ids_sorted, keys32_sorted, n, mhead, d_sc);
        size_t sb = scan_bytes;
        MPT_LB(hipcub::DeviceScan::InclusiveSum(scan_tmp, sb, mhead, mrank, (int)n, side));
        hipLaunchKernelGGL(k_mat_scatter, dim3(gn), dim3(B), 0, side, (const float4*)d_mats_in, ids_sorted, (const uint32_t*)mhead, (const uint32_t*)mrank, n,
                           mat_of_prim, mtable, d_sc);
        return hipGetLastError();
    };
    if (mats_host && two && (size_t)n * 32 >= ((size_t)1 << 20) && getenv("MPT_BUILD_NO_HELPER") == nullptr) {   // (below 1 MB the thread costs what the copy does)
        int dev = 0;
        MPT_LB(hipGetDevice(&dev));
        helper.t = std::thread([&, dev]() {
            helper.err = hipSetDevice(dev);
            if (helper.err == hipSuccess) helper.err = hipMemcpyAsync(d_mats_in, mats_host, (size_t)n * 32, hipMemcpyHostToDevice, side);
            if (helper.err == hipSuccess) helper.err = mats_chain();
        });
    } else {
        if (mats_host) MPT_LB(hipMemcpyAsync(d_mats_in, mats_host, (size_t)n * 32, hipMemcpyHostToDevice, side));
        MPT_LB(mats_chain());
    }
    // the binary tree
    MPT_LB(mpt_lbvh::build_radix(stream, sc, d_prims_in, n, leaf_max, builder, R));
    const uint32_t n_out = R.n_out;
    const size_t nn = 2 * (size_t)n - 1;
    const uint32_t gnn = (uint32_t)((nn + B - 1) / B), go = (n_out + B - 1) / B;
    MPT_LB(mpt_lbvh::emit_reference_format(stream, R, out.ref_bvh, out.ref_idx));
    // leaves
    uint32_t *is_leaf, *leaf_id;
    float4 *olo, *ohi;
    MPT_LB(sc.alloc(&is_leaf, nn + 1));
    MPT_LB(sc.alloc(&leaf_id, nn + 1));
    MPT_LB(sc.alloc(&olo, nn));
    MPT_LB(sc.alloc(&ohi, nn));
    hipLaunchKernelGGL(k_leaf_flags, dim3(gnn), dim3(B), 0, stream, (int)n, leaf_max, (const int2*)R.range, (const uint32_t*)R.keep, is_leaf);
    MPT_LB(hipMemsetAsync(is_leaf + nn, 0, 4, stream));
    {
        size_t sb = 0;
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, is_leaf, leaf_id, (int)nn + 1, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, sb));
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(tmp, sb, is_leaf, leaf_id, (int)nn + 1, stream));
    }
    // (n_spheres_hint = 0xFFFFFFFF: not counted by the caller — the top-down builder has counted them on the device)
    const uint32_t n_sph = n_spheres_hint != 0xFFFFFFFFu ? n_spheres_hint : R.n_spheres;
    if (n_sph == 0xFFFFFFFFu) return hipErrorInvalidValue;
    const int use_always = n_sph <= MPT_ACCEL_MAX_ALWAYS ? 1 : 0;
    // where each leaf's records go (see k_leaf_keys)
    uint32_t* pfirst;
    MPT_LB(sc.alloc(&pfirst, nn));
    if (n >= MPT_AUTO_ORDERED_PRIMS) {
        hipLaunchKernelGGL(k_pfirst_builder_order, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)R.range, (const uint32_t*)is_leaf, pfirst);
    } else {
        uint32_t *key, *key_s, *node_of, *node_s, *cnt, *pos;
        MPT_LB(sc.alloc(&key, n_out));
        MPT_LB(sc.alloc(&key_s, n_out));
        MPT_LB(sc.alloc(&node_of, n_out));
        MPT_LB(sc.alloc(&node_s, n_out));
        MPT_LB(sc.alloc(&cnt, n_out + 1));
        MPT_LB(sc.alloc(&pos, n_out + 1));
        MPT_LB(hipMemcpyAsync(pin, leaf_id + nn, 4, hipMemcpyDeviceToHost, stream));
        MPT_LB(hipStreamSynchronize(stream));
        const uint32_t nl = pin[0], gl = (nl + B - 1) / B;
        hipLaunchKernelGGL(k_leaf_keys, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)R.range, (const uint32_t*)is_leaf, (const uint32_t*)leaf_id,
                           (const uint32_t*)R.vals, (const float4*)d_prims_in, (const float4*)R.nlo, (const float4*)R.nhi, key, node_of);
        size_t bytes = 0, sb = 0;
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, key, key_s, node_of, node_s, (int)nl, 0, 13, stream));
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, cnt, pos, (int)nl, stream));
        char *tmp, *tmp2;
        MPT_LB(sc.alloc(&tmp, bytes));
        MPT_LB(sc.alloc(&tmp2, sb));
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, key, key_s, node_of, node_s, (int)nl, 0, 13, stream));   // (stable: the builder's order among equals)
        hipLaunchKernelGGL(k_leaf_counts, dim3(gl), dim3(B), 0, stream, (int)n, nl, (const int2*)R.range, (const uint32_t*)node_s, cnt);
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(tmp2, sb, cnt, pos, (int)nl, stream));
        hipLaunchKernelGGL(k_pfirst_scatter, dim3(gl), dim3(B), 0, stream, nl, (const uint32_t*)node_s, (const uint32_t*)pos, pfirst);
    }
    if (helper.t.joinable()) helper.t.join();   // (the side stream has been GIVEN everything up to the material table)
    MPT_LB(helper.err);
    MPT_LB(hand_over(side, stream, 1));   // (mat_of_prim)
    hipLaunchKernelGGL(k_leaves, dim3(gnn), dim3(B), 0, stream, (int)n, leaf_max, (const int2*)R.range, (const uint32_t*)is_leaf, (const uint32_t*)leaf_id,
                       (const uint32_t*)R.vals, (const float4*)d_prims_in, (const uint32_t*)mat_of_prim, (const float4*)R.nlo, (const float4*)R.nhi,
                       (const uint32_t*)pfirst, out.prims, out.refleaf, olo, ohi, d_sc, use_always);
    // threaded tree, breadth-first — and the per-primitive leaf boxes: on the side stream, beside the own tree
    MPT_LB(hand_over(stream, side, 2));   // (the leaves)
    uint32_t *depth_c, *depth_s, *id_c, *order, *tpos;
    int* skip;
    MPT_LB(sc.alloc(&depth_c, n_out));
    MPT_LB(sc.alloc(&depth_s, n_out));
    MPT_LB(sc.alloc(&id_c, n_out));
    MPT_LB(sc.alloc(&order, n_out));
    MPT_LB(sc.alloc(&tpos, nn));
    MPT_LB(sc.alloc(&skip, nn));
    hipLaunchKernelGGL(k_depth_skip, dim3(gnn), dim3(B), 0, side, (int)n, (const uint32_t*)R.keep, (const uint32_t*)R.index, (const int2*)R.child,
                       (const int*)R.parent, depth_c, id_c, skip);
    {   // (one counting pass over the 8-bit depths, stable: breadth-first, the builder's order within a level)
        mpt_radix::RadixTemp RT;
        MPT_LB(mpt_radix::radix_reserve(sc, n_out, side, RT));
        bool second = false;
        MPT_LB(mpt_radix::radix_sort_pairs(side, RT, depth_c, id_c, depth_s, order, n_out, 1, &second));
    }
    hipLaunchKernelGGL(k_positions, dim3(go), dim3(B), 0, side, n_out, (const uint32_t*)order, tpos);
    hipLaunchKernelGGL(k_emit_threaded, dim3(go), dim3(B), 0, side, n_out, (int)n, (const uint32_t*)order, (const uint32_t*)tpos, (const uint32_t*)is_leaf,
                       (const int2*)R.child, (const int2*)R.range, (const uint32_t*)pfirst, (const int*)skip, (const float4*)R.nlo, (const float4*)R.nhi, out.nodes);
    hipLaunchKernelGGL(k_prim_refbox, dim3((n + 255u) / 256u), dim3(256), 0, side, (const float4*)out.prims, (const float4*)out.refleaf, n, out.refbox);
    MPT_LB(hipGetLastError());
    // own tree: its binary tree (the builder's own SAH tree refitted, or a binned SAH over the leaves: mpt_sah.h), then the 4-wide collapse
    const uint32_t max_items = n_out;   // (leaves <= output nodes)
    SahState* d_st;
    int2* s_child;
    float4 *s_lo, *s_hi;
    // (the builder's own tree serves when its SAH nodes hold no sphere — the builder hangs up to 16 spheres under the root,
    //  mpt_lbvh.h — or when no always list is used; with spheres inside the SAH their boxes shape the top splits, which the own
    //  tree leaves out: a second SAH over the leaves then.  MPT_OWN_TREE = refit | sah forces either way)
    bool refit = builder == mpt_lbvh::BUILDER_SAH && n > 2 && (R.spheres_hoisted || n_sph == 0 || !use_always);
    if (const char* e = getenv("MPT_OWN_TREE")) refit = builder == mpt_lbvh::BUILDER_SAH && n > 2 && strcmp(e, "refit") == 0;
    if (refit) {
        int *eff, *arrived;
        MPT_LB(sc.alloc(&d_st, 1));
        MPT_LB(sc.alloc(&eff, nn));
        MPT_LB(sc.alloc(&arrived, nn));
        MPT_LB(sc.alloc(&s_child, n));
        MPT_LB(sc.alloc(&s_lo, n));
        MPT_LB(sc.alloc(&s_hi, n));
        MPT_LB(hipMemsetAsync(arrived, 0, nn * 4, stream));
        MPT_LB(hipMemsetAsync(d_st, 0xFF, sizeof(SahState), stream));   // root = -1
        MPT_LB(hipMemsetAsync(s_child, 0xFF, (size_t)n * sizeof(int2), stream));   // (children -1: a node nobody made — k_collapse_picks)
        // (the copy is valid when every leaf's own box is its reference box or, in the chain of hoisted items, empty: spheres hoisted or none
        //  — with spheres INSIDE the SAH's leaves and no always list the boxes agree too, but then nothing marks the chain: the walk)
        const bool fast = (R.spheres_hoisted || n_sph == 0) && getenv("MPT_OWN_TREE_WALK") == nullptr;
        if (fast) {
            hipLaunchKernelGGL(k_own_copy, dim3((n + B - 1) / B), dim3(B), 0, stream, (int)n, (const int2*)R.child, (const uint32_t*)R.keep, (const uint32_t*)is_leaf,
                               (const float4*)R.nlo, (const float4*)R.nhi, (const Scalars*)d_sc, s_lo, s_hi, s_child);
            hipLaunchKernelGGL(k_own_chain, dim3(1), dim3(64), 0, stream, (int)n, (int)R.n_hoisted, (const int2*)R.child, (const uint32_t*)is_leaf, (const float4*)olo,
                               (const float4*)ohi, (const Scalars*)d_sc, s_lo, s_hi, s_child, d_st);
        }
        hipLaunchKernelGGL(k_own_tree, dim3(gnn), dim3(B), 0, stream, (int)n, (const int*)R.parent, (const int2*)R.child, (const uint32_t*)is_leaf, (const float4*)olo,
                           (const float4*)ohi, eff, s_lo, s_hi, s_child, arrived, d_st, fast ? (const Scalars*)d_sc : (const Scalars*)nullptr);
    } else {
        uint32_t *flag_pos, *rank;
        float4 *it_lo_a, *it_hi_a;
        MPT_LB(sc.alloc(&flag_pos, (size_t)n + 1));
        MPT_LB(sc.alloc(&rank, (size_t)n + 1));
        MPT_LB(sc.alloc(&it_lo_a, max_items));
        MPT_LB(sc.alloc(&it_hi_a, max_items));
        MPT_LB(hipMemsetAsync(flag_pos, 0, ((size_t)n + 1) * 4, stream));
        hipLaunchKernelGGL(k_item_flags, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)R.range, (const uint32_t*)is_leaf, (const float4*)olo, (const float4*)ohi,
                           flag_pos);
        size_t sb = 0;
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, flag_pos, rank, (int)n + 1, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, sb));
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(tmp, sb, flag_pos, rank, (int)n + 1, stream));
        hipLaunchKernelGGL(k_items, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)R.range, (const uint32_t*)is_leaf, (const float4*)olo, (const float4*)ohi,
                           (const uint32_t*)rank, it_lo_a, it_hi_a);
        mpt_sah::SahTree T;
        MPT_LB(mpt_sah::run_sah(stream, sc, pin, (int)(2 * n - 1), (const uint32_t*)(rank + n), 0u, max_items, it_lo_a, it_hi_a, T));
        d_st = T.st;
        s_child = T.child;
        s_lo = T.lo;
        s_hi = T.hi;
    }
    const uint32_t cap = std::min(max_items, n) + 2u;   // wide nodes <= inner nodes of the SAH tree over the leaves (+ the root of a one-leaf tree)
    {
        uint32_t *nint_a, *nint_b, *c_offs, *pn;
        int *wbin_a, *wbin_b;
        int4* picks;
        CollapseHead* head;
        MPT_LB(sc.alloc(&nint_a, cap + 1));
        MPT_LB(sc.alloc(&nint_b, cap + 1));
        MPT_LB(sc.alloc(&c_offs, cap + 1));
        MPT_LB(sc.alloc(&wbin_a, cap));
        MPT_LB(sc.alloc(&wbin_b, cap));
        MPT_LB(sc.alloc(&picks, n));
        MPT_LB(sc.alloc(&pn, n));
        MPT_LB(sc.alloc(&head, 1));
        const CollapseAcc A = {s_child, s_lo, s_hi, olo, ohi, (int)(2 * n - 1)};
        size_t sb = 0;
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, nint_a, c_offs, (int)cap + 1, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, sb));
        static uint32_t s_epoch = 0;
        const uint32_t epoch = (++s_epoch & 0xFFFFu) << 16;
        volatile unsigned long long* slots = (volatile unsigned long long*)(pin + 160);   // eight slots of (stamp << 32 | size of the next level), behind run_sah's (words 64 .. 111)
        unsigned long long* d_slots = nullptr;
        MPT_LB(hipHostGetDevicePointer((void**)&d_slots, pin + 160, 0));
        for (int q = 0; q < 8; ++q) slots[q] = 0ull;
        hipLaunchKernelGGL(k_collapse_picks, dim3((n + B - 1) / B), dim3(B), 0, stream, A, n, (const SahState*)d_st, refit ? 0 : 1, picks, pn);
        hipLaunchKernelGGL(k_collapse_top, dim3(1), dim3(MPT_COLLAPSE_TOP), 0, stream, A, (int)n, (const int2*)R.range, (const uint32_t*)pfirst, (const SahState*)d_st,
                           (const int4*)picks, (const uint32_t*)pn, head, wbin_a, nint_a, out.acc_nodes, cap, d_sc);
        uint32_t bound = std::min(4u * MPT_COLLAPSE_TOP, cap);   // upper bound of the level about to be enqueued (the first: what k_collapse_top leaves)
        bool done = false;
        for (uint32_t L = 0; L < 128u && !done; ++L) {
            MPT_LB(hipcub::DeviceScan::ExclusiveSum(tmp, sb, nint_a, c_offs, (int)bound + 1, stream));
            hipLaunchKernelGGL(k_collapse_level, dim3((bound + B - 1) / B), dim3(B), 0, stream, A, (int)n, (const int2*)R.range, (const uint32_t*)pfirst, head, L,
                               (const int4*)picks, (const uint32_t*)pn, (const int*)wbin_a, (const uint32_t*)c_offs, wbin_b, nint_b, out.acc_nodes, cap, d_sc,
                               d_slots + (L & 7u), epoch | (L + 1u));
            MPT_LB(hipGetLastError());
            uint32_t next_bound = (uint32_t)std::min<uint64_t>(4ull * bound, cap);
            if (L >= 1u) {   // the size of level L (slot L - 1), which the device is at or past: nothing there = the tree is complete
                volatile unsigned long long* sl = slots + ((L - 1u) & 7u);
                const uint32_t want = epoch | L;
                const auto t0 = std::chrono::steady_clock::now();
                for (unsigned long long spin = 0; (uint32_t)(*sl >> 32) != want; ++spin)
                    if ((spin & 0xFFFu) == 0xFFFu) {
                        const hipError_t q = hipStreamQuery(stream);
                        if (q != hipSuccess && q != hipErrorNotReady) return q;
                        if (q == hipSuccess && (uint32_t)(*sl >> 32) != want) return hipErrorUnknown;   // the stream is drained and the stamp never came
                        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) return hipErrorLaunchTimeOut;
                    }
                const uint32_t size_L = (uint32_t)*sl;
                if (size_L == 0u) done = true;
                next_bound = (uint32_t)std::min<uint64_t>(4ull * size_L, cap);
            }
            bound = std::max(next_bound, 1u);
            std::swap(nint_a, nint_b);
            std::swap(wbin_a, wbin_b);
        }
        if (!done) return hipErrorUnknown;
    }
    hipLaunchKernelGGL(k_always, dim3(1), dim3(64), 0, stream, d_sc, out.prims, (const float4*)out.refleaf, out.always);
    MPT_LB(hipGetLastError());
    MPT_LB(hand_over(side, stream, 3));
    Scalars& h = *(Scalars*)(pin + 192);   // (pinned: a pageable target costs ~0.3 ms per copy)
    static_assert(sizeof(Scalars) + 4 <= 256, "Scalars must fit the last quarter of the pinned block");
    uint32_t& n_leaves = pin[192 + sizeof(Scalars) / 4];
    MPT_LB(hipMemcpyAsync(&h, d_sc, sizeof h, hipMemcpyDeviceToHost, stream));
    MPT_LB(hipMemcpyAsync(&n_leaves, leaf_id + nn, 4, hipMemcpyDeviceToHost, stream));
    MPT_LB(hipStreamSynchronize(stream));
    const uint32_t acc_nodes_n = h.n_acc_nodes, acc_depth = h.acc_depth;
    out.n_nodes = n_out;
    out.n_prims = n;
    out.n_mats = h.n_mats;
    out.n_acc_nodes = acc_nodes_n;
    out.acc_depth = acc_depth;
    out.n_spheres = h.n_spheres;
    out.n_always = h.n_spheres <= MPT_ACCEL_MAX_ALWAYS ? h.n_spheres : 0u;
    out.n_ref_leaves = n_leaves;
    memcpy(&out.tri_extent, &h.tri_extent, 4);
    *mat_collision = h.mat_collision != 0u;
    return hipSuccess;
}
static hipError_t build(hipStream_t stream, float4* d_prims_in, float4* d_mats_in, const float* mats_host, uint32_t n, int leaf_max, int builder, uint32_t n_spheres_hint,
                        Built& out, mpt_lbvh::ScratchPool* pool = nullptr, void** spare = nullptr, size_t* spare_bytes = nullptr) {
    bool collision = false;
    MPT_LB(build_pass(stream, d_prims_in, d_mats_in, mats_host, n, leaf_max, builder, n_spheres_hint, out, pool, spare, spare_bytes, false, &collision));
    if (!collision) return hipSuccess;
    // two different materials share the upper half of their hashes (one scene in ~2^33 / materials^2): once more, sorted on all 64 bits,
    // into the same block
    void* block = out.block;
    size_t bytes = out.block_bytes;
    out = Built{};
    const hipError_t e = build_pass(stream, d_prims_in, d_mats_in, nullptr, n, leaf_max, builder, n_spheres_hint, out, pool, &block, &bytes, true, &collision);   // (the materials are on the device by now)
    if (block) hipFree(block);   // (not taken: the pass failed before it got there)
    return e;
}

}  // namespace mpt_devbuild
