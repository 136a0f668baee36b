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

#include "mpt_accel.h"
#include "mpt_device.h"
#include "mpt_lbvh.h"

namespace mpt_devbuild {
using mpt_lbvh::Radix;
using mpt_lbvh::Scratch;

struct Scalars {             // device-side results the host reads back once, at the end
    uint32_t n_spheres;      // spheres found (the first 32 positions are recorded)
    uint32_t sphere_pos[32]; // their positions in the device primitive array
    uint32_t tri_extent;     // bits of the largest finite |coordinate| of a triangle vertex
    uint32_t n_mats;
    uint32_t n_leaves;
    uint32_t n_acc_nodes, acc_depth;
};

__device__ __forceinline__ int span_of(const int2* range, int n, int node) { return node >= n - 1 ? 1 : range[node].y - range[node].x + 1; }
__device__ __forceinline__ int first_of(const int2* range, int n, int node) { return node >= n - 1 ? node - (n - 1) : range[node].x; }

// ---- materials: sort by a 64-bit hash, mark the runs, number them ---------------------------------------------------------
__global__ void k_mat_hash(const float4* mats, uint32_t n, unsigned long long* keys, uint32_t* ids) {
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
    keys[i] = h;
    ids[i] = i;
}
__global__ void k_mat_heads(const float4* mats, const uint32_t* ids, uint32_t n, uint32_t* head) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    bool h = j == 0;
    if (!h) {
        const uint4* m = (const uint4*)mats;
        const uint4 a0 = m[2 * (size_t)ids[j]], a1 = m[2 * (size_t)ids[j] + 1], b0 = m[2 * (size_t)ids[j - 1]], b1 = m[2 * (size_t)ids[j - 1] + 1];
        h = a0.x != b0.x || a0.y != b0.y || a0.z != b0.z || a0.w != b0.w || a1.x != b1.x || a1.y != b1.y || a1.z != b1.z || a1.w != b1.w;
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
__global__ void k_tri_extent(const float4* prims, uint32_t n, Scalars* sc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    if (i < n) {
        const float4 p0 = prims[3 * (size_t)i], p1 = prims[3 * (size_t)i + 1], p2 = prims[3 * (size_t)i + 2];
        if ((int)p0.w == 1) {
            const float v[9] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
            for (int k = 0; k < 9; ++k)
                if (isfinite(v[k])) m = fmaxf(m, fabsf(v[k]));
        }
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63u) == 0 && m > 0.0f) atomicMax(&sc->tri_extent, __float_as_uint(m));
}

// ---- leaves: number them, lay their primitives out, their reference boxes, their own (sphere-free) boxes ----------------------
__global__ void k_leaf_flags(int n, int leaf_max, const int2* range, const uint32_t* keep, uint32_t* is_leaf) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1) return;
    is_leaf[node] = keep[node] && span_of(range, n, node) <= leaf_max ? 1u : 0u;
}
// one thread per output leaf.  own box of the leaf BEFORE the final padding: the reference leaf box; for a leaf that holds a
// sphere, a box around its triangles only (+5 % of their extent + pad, clipped to the leaf box) — empty if it has none
__global__ void k_leaves(int n, int leaf_max, const int2* range, const uint32_t* is_leaf, const uint32_t* leaf_id, const uint32_t* vals,
                         const float4* prims, const uint32_t* mat_of_prim, const float4* nlo, const float4* nhi, float4* dprims,
                         float4* refleaf, float4* olo, float4* ohi, Scalars* sc, int use_always_hint) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1 || !is_leaf[node]) return;
    const uint32_t leaf = leaf_id[node];
    const int first = first_of(range, n, node), count = span_of(range, n, node);
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
            if (s < 32u) sc->sphere_pos[s] = (uint32_t)(first + k);
        }
        dprims[3 * (size_t)(first + k)] = r0;
        dprims[3 * (size_t)(first + k) + 1] = r1;
        dprims[3 * (size_t)(first + k) + 2] = r2;
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
                                const int2* range, const int* skip, const float4* nlo, const float4* nhi, float4* nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const int id = (int)order[i];
    const float4 lo = nlo[id], hi = nhi[id];
    const uint32_t next = skip[id] < 0 ? n_out : tpos[skip[id]];
    uint32_t A, B = next;
    if (is_leaf[id]) A = MPT_NODE_HOLD | ((uint32_t)first_of(range, n, id) * 16u + (uint32_t)(span_of(range, n, id) - 1));
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
struct SahTask {
    uint32_t b, e;     // items [b, e) of the current index array
    int parent;        // inner node that waits for this sub-tree (-1: the root)
    uint32_t side;
};
#ifndef MPT_SAH_BIG
#define MPT_SAH_BIG 2048u   // tasks of at least this many items get a whole workgroup (k_sah_level_big), smaller ones a wave
#endif
struct SahState {
    uint32_t n_next;      // tasks pushed for the next level (small ones: a wave each)
    uint32_t n_next_big;  // ... and the big ones
    uint32_t n_nodes;  // inner nodes created
    int root;          // -1: no items
    uint32_t n_items;
};
__device__ __forceinline__ float half_area4(float4 lo, float4 hi) {
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ bool empty4(float4 lo, float4 hi) { return !(hi.x >= lo.x && hi.y >= lo.y && hi.z >= lo.z); }
__device__ __forceinline__ float axis_of(float4 v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }
__device__ __forceinline__ int f2o(float f) {  // order-preserving float -> int
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float o2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

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
__global__ void k_sah_init(const uint32_t* rank, int n, SahState* st, SahTask* tasks, SahTask* big_tasks) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t m = rank[n];
    st->n_items = m;
    st->n_nodes = 0u;
    st->root = -1;
    st->n_next = 0u;
    st->n_next_big = 0u;
    if (m >= MPT_SAH_BIG) {
        big_tasks[0] = SahTask{0u, m, -1, 0u};
        st->n_next_big = 1u;
    } else if (m != 0u) {
        tasks[0] = SahTask{0u, m, -1, 0u};
        st->n_next = 1u;
    }
}
#define MPT_SAH_WAVES 4   // per workgroup
__device__ __forceinline__ void sah_attach(SahState* st, int2* s_child, int parent, uint32_t side, int id) {
    if (parent < 0) st->root = id;
    else if (side == 0u) s_child[parent].x = id;
    else s_child[parent].y = id;
}
// a finished split: a single item is attached at once, anything larger becomes a task of the next level
__device__ __forceinline__ void sah_push_children(SahState* st, int2* s_child, SahTask* next, SahTask* next_big, uint32_t b, uint32_t e, uint32_t nlft,
                                                  uint32_t k, int one_left, int one_right) {
    const uint32_t mid = b + nlft, nrgt = e - mid;
    if (nlft == 1u) s_child[k].x = one_left;
    else if (nlft >= MPT_SAH_BIG) next_big[atomicAdd(&st->n_next_big, 1u)] = SahTask{b, mid, (int)k, 0u};
    else next[atomicAdd(&st->n_next, 1u)] = SahTask{b, mid, (int)k, 0u};
    if (nrgt == 1u) s_child[k].y = one_right;
    else if (nrgt >= MPT_SAH_BIG) next_big[atomicAdd(&st->n_next_big, 1u)] = SahTask{mid, e, (int)k, 1u};
    else next[atomicAdd(&st->n_next, 1u)] = SahTask{mid, e, (int)k, 1u};
}
__global__ __launch_bounds__(64 * MPT_SAH_WAVES) void k_sah_level(int n, const float4* in_lo, const float4* in_hi, float4* out_lo, float4* out_hi,
                                                                  const SahTask* tasks, uint32_t n_tasks, SahTask* next, SahTask* next_big, SahState* st,
                                                                  int2* s_child, float4* s_lo, float4* s_hi) {
    __shared__ int bins[MPT_SAH_WAVES][3][16][7];   // per wave: (lo xyz, hi xyz as ordered ints, primitive count) per axis and bin
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * MPT_SAH_WAVES + wv;
    if (t >= n_tasks) return;
    const int TOP = 2 * n - 1;
    const SahTask task = tasks[t];
    const uint32_t b = task.b, e = task.e, m = e - b;
    if (m == 1u) {   // (only the root task of a one-leaf tree comes here: children of one item are attached when they are split off)
        if (lane == 0) sah_attach(st, s_child, task.parent, task.side, __float_as_int(in_lo[b].w));
        return;
    }
    if (m == 2u) {   // two items: the node, nothing to choose (a third of all tasks, at the bottom of the tree)
        if (lane == 0) {
            const float4 l0 = in_lo[b], h0 = in_hi[b], l1 = in_lo[b + 1u], h1 = in_hi[b + 1u];
            const uint32_t k2 = atomicAdd(&st->n_nodes, 1u);
            s_lo[k2] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.0f);
            s_hi[k2] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.0f);
            s_child[k2] = make_int2(__float_as_int(l0.w), __float_as_int(l1.w));
            sah_attach(st, s_child, task.parent, task.side, TOP + (int)k2);
        }
        return;
    }
    // pass 1: the node's box and the bounds of the box centres
    float nl[3] = {INFINITY, INFINITY, INFINITY}, nh[3] = {-INFINITY, -INFINITY, -INFINITY}, cl[3] = {INFINITY, INFINITY, INFINITY},
          ch[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = b + lane; i < e; i += 64u) {
        const float4 l = in_lo[i], h = in_hi[i];
        const float lo3[3] = {l.x, l.y, l.z}, hi3[3] = {h.x, h.y, h.z};
        for (int a = 0; a < 3; ++a) {
            nl[a] = fminf(nl[a], lo3[a]);
            nh[a] = fmaxf(nh[a], hi3[a]);
            const float c = 0.5f * (lo3[a] + hi3[a]);
            cl[a] = fminf(cl[a], c);
            ch[a] = fmaxf(ch[a], c);
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; ++a) {
            nl[a] = fminf(nl[a], __shfl_xor(nl[a], off));
            nh[a] = fmaxf(nh[a], __shfl_xor(nh[a], off));
            cl[a] = fminf(cl[a], __shfl_xor(cl[a], off));
            ch[a] = fmaxf(ch[a], __shfl_xor(ch[a], off));
        }
    uint32_t k = 0;
    if (lane == 0) {
        k = atomicAdd(&st->n_nodes, 1u);
        s_lo[k] = make_float4(nl[0], nl[1], nl[2], 0.0f);
        s_hi[k] = make_float4(nh[0], nh[1], nh[2], 0.0f);
        sah_attach(st, s_child, task.parent, task.side, TOP + (int)k);
    }
    k = (uint32_t)__shfl((int)k, 0);
    // pass 2: bins
    for (uint32_t q = lane; q < 3u * 16u * 7u; q += 64u) {
        const uint32_t f = q % 7u;
        (&bins[wv][0][0][0])[q] = f < 3u ? 0x7FFFFFFF : f < 6u ? (int)0x80000000 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    float inv[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = ch[a] - cl[a];
        inv[a] = ext > 0.0f && isfinite(ext) ? 16.0f / ext : 0.0f;
    }
    // (a node of more than 256 items is binned from an evenly spaced sample of ~256 of them: the 21 LDS atomics per item are
    //  what this pass costs, and 256 boxes choose among 45 planes as well as 100,000 do; box and partition stay exact)
    const uint32_t step = m > 256u ? m / 256u : 1u;
    for (uint32_t i = b + lane * step; i < e; i += 64u * step) {
        const float4 l = in_lo[i], h = in_hi[i];
        const int cnt = __float_as_int(h.w);
        const float lo3[3] = {l.x, l.y, l.z}, hi3[3] = {h.x, h.y, h.z};
        for (int a = 0; a < 3; ++a) {
            if (inv[a] == 0.0f) continue;
            int q = (int)((0.5f * (lo3[a] + hi3[a]) - cl[a]) * inv[a]);
            q = q < 0 ? 0 : (q > 15 ? 15 : q);
            int* B = bins[wv][a][q];
            atomicMin(&B[0], f2o(l.x)); atomicMin(&B[1], f2o(l.y)); atomicMin(&B[2], f2o(l.z));
            atomicMax(&B[3], f2o(h.x)); atomicMax(&B[4], f2o(h.y)); atomicMax(&B[5], f2o(h.z));
            atomicAdd(&B[6], cnt);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // pass 3: lane = (axis, split after bin s): cost = area(L) * count(L) + area(R) * count(R)
    float cost = INFINITY;
    if (lane < 45u) {
        const int a = (int)(lane / 15u), sp = (int)(lane % 15u);
        if (inv[a] != 0.0f) {
            float L[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, R[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int cL = 0, cR = 0;
            for (int q = 0; q < 16; ++q) {
                const int* B = bins[wv][a][q];
                if (B[6] == 0) continue;
                float* D = q <= sp ? L : R;
                for (int c = 0; c < 3; ++c) {
                    D[c] = fminf(D[c], o2f(B[c]));
                    D[3 + c] = fmaxf(D[3 + c], o2f(B[3 + c]));
                }
                if (q <= sp) cL += B[6];
                else cR += B[6];
            }
            if (cL != 0 && cR != 0)
                cost = half_area4(make_float4(L[0], L[1], L[2], 0), make_float4(L[3], L[4], L[5], 0)) * (float)cL +
                       half_area4(make_float4(R[0], R[1], R[2], 0), make_float4(R[3], R[4], R[5], 0)) * (float)cR;
        }
    }
    float best = cost;
    for (int off = 32; off > 0; off >>= 1) best = fminf(best, __shfl_xor(best, off));
    const unsigned long long who = __ballot(cost == best && best < INFINITY);
    const int pick = who != 0ull ? (int)__ffsll((long long)who) - 1 : -1;   // ties: the lowest (axis, split)
    const int paxis = pick >= 0 ? pick / 15 : 0, psplit = pick >= 0 ? pick % 15 : 0;
    // partition into the other array: left from b upwards, right from e - 1 downwards
    uint32_t nlft = 0, nrgt = 0;
    int one_left = 0, one_right = 0;   // (the first item that went to either side: THE item if it stays alone)
    for (uint32_t base = b; base < e; base += 64u) {
        const uint32_t i = base + lane;
        const bool valid = i < e;
        float4 l = make_float4(0, 0, 0, 0), h = l;
        bool left = false;
        if (valid) {
            l = in_lo[i];
            h = in_hi[i];
            if (pick >= 0) {
                int q = (int)((0.5f * (axis_of(l, paxis) + axis_of(h, paxis)) - cl[paxis]) * inv[paxis]);
                q = q < 0 ? 0 : (q > 15 ? 15 : q);
                left = q <= psplit;
            } else {
                left = i - b < m / 2u;   // no plane separates the box centres: halves
            }
        }
        const unsigned long long lm = __ballot(valid && left), rm = __ballot(valid && !left);
        if (valid) {
            const uint32_t dst = left ? b + nlft + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))
                                      : e - 1u - nrgt - (uint32_t)__popcll(rm & ((1ull << lane) - 1ull));
            out_lo[dst] = l;
            out_hi[dst] = h;
        }
        if (nlft == 0u && lm != 0ull) one_left = __shfl(__float_as_int(l.w), __ffsll((long long)lm) - 1);
        if (nrgt == 0u && rm != 0ull) one_right = __shfl(__float_as_int(l.w), __ffsll((long long)rm) - 1);
        nlft += (uint32_t)__popcll(lm);
        nrgt += (uint32_t)__popcll(rm);
    }
    if (lane == 0) sah_push_children(st, s_child, next, next_big, b, e, nlft, k, one_left, one_right);
}

// The same for a BIG task, by a whole workgroup of 1024 threads (the root of a 500 k-leaf tree: 1.5 ms instead of 70): bins
// shared in LDS, the partition chunk by chunk with a prefix over the waves' ballots (deterministic item order).
#define MPT_SAH_BIG_THREADS 1024
__global__ __launch_bounds__(MPT_SAH_BIG_THREADS) void k_sah_level_big(int n, const float4* in_lo, const float4* in_hi, float4* out_lo, float4* out_hi,
                                                                      const SahTask* tasks, SahTask* next, SahTask* next_big, SahState* st, int2* s_child,
                                                                      float4* s_lo, float4* s_hi) {
    __shared__ int bounds[12];          // node box lo/hi, centre bounds lo/hi (ordered ints)
    __shared__ int bins[3][16][7];
    __shared__ float s_cost[48];
    __shared__ uint32_t s_wl[16], s_wr[16], s_k;
    __shared__ int s_one[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const int TOP = 2 * n - 1;
    const SahTask task = tasks[blockIdx.x];
    const uint32_t b = task.b, e = task.e, m = e - b;
    if (tid < 12u) bounds[tid] = tid % 6u < 3u ? 0x7FFFFFFF : (int)0x80000000;
    for (uint32_t q = tid; q < 3u * 16u * 7u; q += MPT_SAH_BIG_THREADS) {
        const uint32_t f = q % 7u;
        (&bins[0][0][0])[q] = f < 3u ? 0x7FFFFFFF : f < 6u ? (int)0x80000000 : 0;
    }
    __syncthreads();
    // pass 1
    {
        float nl[3] = {INFINITY, INFINITY, INFINITY}, nh[3] = {-INFINITY, -INFINITY, -INFINITY}, cl[3] = {INFINITY, INFINITY, INFINITY},
              ch[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = b + tid; i < e; i += MPT_SAH_BIG_THREADS) {
            const float4 l = in_lo[i], h = in_hi[i];
            const float lo3[3] = {l.x, l.y, l.z}, hi3[3] = {h.x, h.y, h.z};
            for (int a = 0; a < 3; ++a) {
                nl[a] = fminf(nl[a], lo3[a]);
                nh[a] = fmaxf(nh[a], hi3[a]);
                const float c = 0.5f * (lo3[a] + hi3[a]);
                cl[a] = fminf(cl[a], c);
                ch[a] = fmaxf(ch[a], c);
            }
        }
        for (int off = 32; off > 0; off >>= 1)
            for (int a = 0; a < 3; ++a) {
                nl[a] = fminf(nl[a], __shfl_xor(nl[a], off));
                nh[a] = fmaxf(nh[a], __shfl_xor(nh[a], off));
                cl[a] = fminf(cl[a], __shfl_xor(cl[a], off));
                ch[a] = fmaxf(ch[a], __shfl_xor(ch[a], off));
            }
        if (lane == 0)
            for (int a = 0; a < 3; ++a) {
                atomicMin(&bounds[a], f2o(nl[a]));
                atomicMax(&bounds[3 + a], f2o(nh[a]));
                atomicMin(&bounds[6 + a], f2o(cl[a]));
                atomicMax(&bounds[9 + a], f2o(ch[a]));
            }
    }
    __syncthreads();
    float cl[3], inv[3];
    for (int a = 0; a < 3; ++a) {
        cl[a] = o2f(bounds[6 + a]);
        const float ext = o2f(bounds[9 + a]) - cl[a];
        inv[a] = ext > 0.0f && isfinite(ext) ? 16.0f / ext : 0.0f;
    }
    if (tid == 0) {
        const uint32_t k = atomicAdd(&st->n_nodes, 1u);
        s_k = k;
        s_lo[k] = make_float4(o2f(bounds[0]), o2f(bounds[1]), o2f(bounds[2]), 0.0f);
        s_hi[k] = make_float4(o2f(bounds[3]), o2f(bounds[4]), o2f(bounds[5]), 0.0f);
        sah_attach(st, s_child, task.parent, task.side, TOP + (int)k);
    }
    // pass 2
    const uint32_t step = m > 4096u ? m / 4096u : 1u;   // (binned from a sample of ~4096 items: see k_sah_level)
    for (uint32_t i = b + tid * step; i < e; i += MPT_SAH_BIG_THREADS * step) {
        const float4 l = in_lo[i], h = in_hi[i];
        const int cnt = __float_as_int(h.w);
        const float lo3[3] = {l.x, l.y, l.z}, hi3[3] = {h.x, h.y, h.z};
        for (int a = 0; a < 3; ++a) {
            if (inv[a] == 0.0f) continue;
            int q = (int)((0.5f * (lo3[a] + hi3[a]) - cl[a]) * inv[a]);
            q = q < 0 ? 0 : (q > 15 ? 15 : q);
            int* B = bins[a][q];
            atomicMin(&B[0], f2o(l.x)); atomicMin(&B[1], f2o(l.y)); atomicMin(&B[2], f2o(l.z));
            atomicMax(&B[3], f2o(h.x)); atomicMax(&B[4], f2o(h.y)); atomicMax(&B[5], f2o(h.z));
            atomicAdd(&B[6], cnt);
        }
    }
    __syncthreads();
    // pass 3
    if (tid < 48u) {
        float cost = INFINITY;
        if (tid < 45u) {
            const int a = (int)(tid / 15u), sp = (int)(tid % 15u);
            if (inv[a] != 0.0f) {
                float L[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, R[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int cL = 0, cR = 0;
                for (int q = 0; q < 16; ++q) {
                    const int* B = bins[a][q];
                    if (B[6] == 0) continue;
                    float* D = q <= sp ? L : R;
                    for (int c = 0; c < 3; ++c) {
                        D[c] = fminf(D[c], o2f(B[c]));
                        D[3 + c] = fmaxf(D[3 + c], o2f(B[3 + c]));
                    }
                    if (q <= sp) cL += B[6];
                    else cR += B[6];
                }
                if (cL != 0 && cR != 0)
                    cost = half_area4(make_float4(L[0], L[1], L[2], 0), make_float4(L[3], L[4], L[5], 0)) * (float)cL +
                           half_area4(make_float4(R[0], R[1], R[2], 0), make_float4(R[3], R[4], R[5], 0)) * (float)cR;
            }
        }
        s_cost[tid] = cost;
    }
    if (tid < 2u) s_one[tid] = 0;
    __syncthreads();
    int pick = -1;
    {
        float best = INFINITY;
        for (int q = 0; q < 45; ++q)
            if (s_cost[q] < best) best = s_cost[q], pick = q;   // ties: the lowest (axis, split), as in the wave kernel
    }
    const int paxis = pick >= 0 ? pick / 15 : 0, psplit = pick >= 0 ? pick % 15 : 0;
    // partition, chunk by chunk
    uint32_t nlft = 0, nrgt = 0;   // (uniform: running totals)
    for (uint32_t base = b; base < e; base += MPT_SAH_BIG_THREADS) {
        const uint32_t i = base + tid;
        const bool valid = i < e;
        float4 l = make_float4(0, 0, 0, 0), h = l;
        bool left = false;
        if (valid) {
            l = in_lo[i];
            h = in_hi[i];
            if (pick >= 0) {
                int q = (int)((0.5f * (axis_of(l, paxis) + axis_of(h, paxis)) - cl[paxis]) * inv[paxis]);
                q = q < 0 ? 0 : (q > 15 ? 15 : q);
                left = q <= psplit;
            } else {
                left = i - b < m / 2u;
            }
        }
        const unsigned long long lm = __ballot(valid && left), rm = __ballot(valid && !left);
        if (lane == 0) {
            s_wl[wv] = (uint32_t)__popcll(lm);
            s_wr[wv] = (uint32_t)__popcll(rm);
        }
        __syncthreads();
        uint32_t pl = 0, pr = 0, tl = 0, tr = 0;
        for (uint32_t w = 0; w < 16u; ++w) {
            if (w < wv) pl += s_wl[w], pr += s_wr[w];
            tl += s_wl[w];
            tr += s_wr[w];
        }
        if (valid) {
            const uint32_t dst = left ? b + nlft + pl + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))
                                      : e - 1u - nrgt - pr - (uint32_t)__popcll(rm & ((1ull << lane) - 1ull));
            out_lo[dst] = l;
            out_hi[dst] = h;
        }
        if (nlft == 0u && tl != 0u && lm != 0ull && pl == 0u && lane == (uint32_t)(__ffsll((long long)lm) - 1)) s_one[0] = __float_as_int(l.w);
        if (nrgt == 0u && tr != 0u && rm != 0ull && pr == 0u && lane == (uint32_t)(__ffsll((long long)rm) - 1)) s_one[1] = __float_as_int(l.w);
        nlft += tl;
        nrgt += tr;
        __syncthreads();
    }
    if (tid == 0) sah_push_children(st, s_child, next, next_big, b, e, nlft, s_k, s_one[0], s_one[1]);
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
__global__ void k_collapse_root(const SahState* st, uint32_t* wbin) {
    if (threadIdx.x == 0 && blockIdx.x == 0) wbin[0] = (uint32_t)st->root;
}
__global__ void k_collapse_pick(CollapseAcc A, const uint32_t* wbin, uint32_t begin, uint32_t end, int4* picked, uint32_t* nint_out) {
    const uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= end) return;
    int ch[4] = {-1, -1, -1, -1};
    int nc = 0;
    const int b = (int)wbin[k];
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
                ch[bi] = g.x;
                ch[nc++] = g.y;
            }
        }
    }
    uint32_t nint = 0;
    for (int i = 0; i < nc; ++i) nint += A.is_leaf(ch[i]) ? 0u : 1u;
    picked[k - begin] = make_int4(ch[0], ch[1], ch[2], ch[3]);
    nint_out[k - begin] = nint;
}
__global__ void k_collapse_emit(CollapseAcc A, int n, const int2* range, uint32_t* wbin, uint32_t begin, uint32_t end, const int4* picked, const uint32_t* offs,
                                float4* acc_nodes, uint32_t cap, const Scalars* sc) {
    const uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= end || k >= cap) return;
    const float pad = fmaxf(__uint_as_float(sc->tri_extent), 1e-6f) * 6.103515625e-05f;  // 2^-14: covers the rcp / fma slab arithmetic
    const int4 p = picked[k - begin];
    const int ch[4] = {p.x, p.y, p.z, p.w};
    const uint32_t at = end + offs[k - begin];
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
                ref = MPT_ACCEL_LEAF | ((uint32_t)(span_of(range, n, ch[c]) - 1) << 27) | (uint32_t)first_of(range, n, ch[c]);
            } else {
                ref = at + j;
                if (at + j < cap) wbin[at + j] = (uint32_t)ch[c];
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
    float4 *nodes = nullptr, *prims = nullptr, *mats = nullptr, *acc_nodes = nullptr, *refleaf = nullptr, *always = nullptr;
    float4* ref_bvh = nullptr;   // the same tree in the reference's buffer format (2 float4 per node) ...
    int* ref_idx = nullptr;      // ... and its primitiveIndices
    uint32_t n_nodes = 0, n_prims = 0, n_mats = 0, n_acc_nodes = 0, n_always = 0, n_ref_leaves = 0, acc_depth = 0, n_spheres = 0;
    float tri_extent = 0.0f;
    void release() {
        hipFree(nodes); hipFree(prims); hipFree(mats); hipFree(acc_nodes); hipFree(refleaf); hipFree(always); hipFree(ref_bvh); hipFree(ref_idx);
        *this = Built{};
    }
};

static hipError_t build(hipStream_t stream, float4* d_prims_in, const float4* d_mats_in, uint32_t n, int leaf_max, bool use_ploc, uint32_t n_spheres_hint, Built& out) {
    Scratch sc;
    Radix R;
    Scalars* d_sc;
    struct Pinned {   // read-backs of a few words per level go through pinned memory (a pageable target costs ~0.3 ms per copy)
        uint32_t* p = nullptr;
        ~Pinned() { if (p) hipHostFree(p); }
    } pinned;
    MPT_LB(hipHostMalloc((void**)&pinned.p, 256, hipHostMallocDefault));
    uint32_t* pin = pinned.p;
    MPT_LB(sc.alloc(&d_sc, 1));
    MPT_LB(hipMemsetAsync(d_sc, 0, sizeof(Scalars), stream));
    const uint32_t B = 256, gn = (n + B - 1) / B;
    hipLaunchKernelGGL(k_tri_extent, dim3(gn), dim3(B), 0, stream, (const float4*)d_prims_in, n, d_sc);
    // materials
    unsigned long long *mk, *mk2;
    uint32_t *mi, *mi2, *mhead, *mrank, *mat_of_prim;
    float4* mtable;
    MPT_LB(sc.alloc(&mk, n));
    MPT_LB(sc.alloc(&mk2, n));
    MPT_LB(sc.alloc(&mi, n));
    MPT_LB(sc.alloc(&mi2, n));
    MPT_LB(sc.alloc(&mhead, n));
    MPT_LB(sc.alloc(&mrank, n));
    MPT_LB(sc.alloc(&mat_of_prim, n));
    MPT_LB(sc.alloc(&mtable, 2 * (size_t)n));
    hipLaunchKernelGGL(k_mat_hash, dim3(gn), dim3(B), 0, stream, d_mats_in, n, mk, mi);
    {
        size_t bytes = 0;
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, mk, mk2, mi, mi2, (int)n, 0, 64, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, bytes));
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, mk, mk2, mi, mi2, (int)n, 0, 64, stream));
        hipLaunchKernelGGL(k_mat_heads, dim3(gn), dim3(B), 0, stream, d_mats_in, (const uint32_t*)mi2, n, mhead);
        size_t sb = 0;
        MPT_LB(hipcub::DeviceScan::InclusiveSum(nullptr, sb, mhead, mrank, (int)n, stream));
        char* tmp2;
        MPT_LB(sc.alloc(&tmp2, sb));
        MPT_LB(hipcub::DeviceScan::InclusiveSum(tmp2, sb, mhead, mrank, (int)n, stream));
        hipLaunchKernelGGL(k_mat_scatter, dim3(gn), dim3(B), 0, stream, d_mats_in, (const uint32_t*)mi2, (const uint32_t*)mhead, (const uint32_t*)mrank, n,
                           mat_of_prim, mtable, d_sc);
    }
    // the binary tree
    MPT_LB(mpt_lbvh::build_radix(stream, sc, d_prims_in, n, leaf_max, use_ploc, R));
    const uint32_t n_out = R.n_out;
    const size_t nn = 2 * (size_t)n - 1;
    const uint32_t gnn = (uint32_t)((nn + B - 1) / B), go = (n_out + B - 1) / B;
    MPT_LB(hipMalloc(&out.ref_bvh, (size_t)n_out * 32));
    MPT_LB(hipMalloc(&out.ref_idx, (size_t)n * 4));
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
    MPT_LB(hipMalloc(&out.prims, (size_t)n * 48));
    MPT_LB(hipMalloc(&out.refleaf, (size_t)n_out * 32));   // (leaves <= output nodes)
    const int use_always = n_spheres_hint <= MPT_ACCEL_MAX_ALWAYS ? 1 : 0;
    hipLaunchKernelGGL(k_leaves, dim3(gnn), dim3(B), 0, stream, (int)n, leaf_max, (const int2*)R.range, (const uint32_t*)is_leaf, (const uint32_t*)leaf_id,
                       (const uint32_t*)R.vals, (const float4*)d_prims_in, (const uint32_t*)mat_of_prim, (const float4*)R.nlo, (const float4*)R.nhi, out.prims,
                       out.refleaf, olo, ohi, d_sc, use_always);
    // threaded tree, breadth-first
    uint32_t *depth_c, *depth_s, *id_c, *order, *tpos;
    int* skip;
    MPT_LB(sc.alloc(&depth_c, n_out));
    MPT_LB(sc.alloc(&depth_s, n_out));
    MPT_LB(sc.alloc(&id_c, n_out));
    MPT_LB(sc.alloc(&order, n_out));
    MPT_LB(sc.alloc(&tpos, nn));
    MPT_LB(sc.alloc(&skip, nn));
    hipLaunchKernelGGL(k_depth_skip, dim3(gnn), dim3(B), 0, stream, (int)n, (const uint32_t*)R.keep, (const uint32_t*)R.index, (const int2*)R.child,
                       (const int*)R.parent, depth_c, id_c, skip);
    {
        size_t bytes = 0;
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, depth_c, depth_s, id_c, order, (int)n_out, 0, 8, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, bytes));
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, depth_c, depth_s, id_c, order, (int)n_out, 0, 8, stream));
    }
    hipLaunchKernelGGL(k_positions, dim3(go), dim3(B), 0, stream, n_out, (const uint32_t*)order, tpos);
    MPT_LB(hipMalloc(&out.nodes, (size_t)n_out * 32));
    hipLaunchKernelGGL(k_emit_threaded, dim3(go), dim3(B), 0, stream, n_out, (int)n, (const uint32_t*)order, (const uint32_t*)tpos, (const uint32_t*)is_leaf,
                       (const int2*)R.child, (const int2*)R.range, (const int*)skip, (const float4*)R.nlo, (const float4*)R.nhi, out.nodes);
    // own tree: binned SAH over the leaves (a wave or a workgroup per node, one launch pair per level), then the 4-wide collapse
    uint32_t *flag_pos, *rank;
    float4 *it_lo_a, *it_hi_a, *it_lo_b, *it_hi_b;
    SahState* d_st;
    const uint32_t max_items = n_out;   // (leaves <= output nodes)
    MPT_LB(sc.alloc(&flag_pos, (size_t)n + 1));
    MPT_LB(sc.alloc(&rank, (size_t)n + 1));
    MPT_LB(sc.alloc(&d_st, 1));
    MPT_LB(sc.alloc(&it_lo_a, max_items));
    MPT_LB(sc.alloc(&it_hi_a, max_items));
    MPT_LB(sc.alloc(&it_lo_b, max_items));
    MPT_LB(sc.alloc(&it_hi_b, max_items));
    SahTask *tasks_a, *tasks_b, *big_a, *big_b;
    int2* s_child;
    float4 *s_lo, *s_hi;
    MPT_LB(sc.alloc(&tasks_a, max_items + 2));
    MPT_LB(sc.alloc(&tasks_b, max_items + 2));
    MPT_LB(sc.alloc(&big_a, max_items / MPT_SAH_BIG + 2));
    MPT_LB(sc.alloc(&big_b, max_items / MPT_SAH_BIG + 2));
    MPT_LB(sc.alloc(&s_child, max_items));
    MPT_LB(sc.alloc(&s_lo, max_items));
    MPT_LB(sc.alloc(&s_hi, max_items));
    MPT_LB(hipMemsetAsync(flag_pos, 0, ((size_t)n + 1) * 4, stream));
    hipLaunchKernelGGL(k_item_flags, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)R.range, (const uint32_t*)is_leaf, (const float4*)olo, (const float4*)ohi, flag_pos);
    {
        size_t sb = 0;
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, flag_pos, rank, (int)n + 1, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, sb));
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(tmp, sb, flag_pos, rank, (int)n + 1, stream));
    }
    hipLaunchKernelGGL(k_items, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)R.range, (const uint32_t*)is_leaf, (const float4*)olo, (const float4*)ohi,
                       (const uint32_t*)rank, it_lo_a, it_hi_a);
    hipLaunchKernelGGL(k_sah_init, dim3(1), dim3(64), 0, stream, (const uint32_t*)rank, (int)n, d_st, tasks_a, big_a);
    {
        SahState& h = *(SahState*)pin;
        MPT_LB(hipMemcpyAsync(&h, d_st, sizeof h, hipMemcpyDeviceToHost, stream));
        MPT_LB(hipStreamSynchronize(stream));
        uint32_t n_tasks = h.n_next, n_big = h.n_next_big;
        for (int level = 0; level < 4096 && (n_tasks | n_big) != 0u; ++level) {
            MPT_LB(hipMemsetAsync(&d_st->n_next, 0, 8, stream));   // n_next, n_next_big
            if (n_big)
                hipLaunchKernelGGL(k_sah_level_big, dim3(n_big), dim3(MPT_SAH_BIG_THREADS), 0, stream, (int)n, (const float4*)it_lo_a, (const float4*)it_hi_a, it_lo_b,
                                   it_hi_b, (const SahTask*)big_a, tasks_b, big_b, d_st, s_child, s_lo, s_hi);
            if (n_tasks)
                hipLaunchKernelGGL(k_sah_level, dim3((n_tasks + MPT_SAH_WAVES - 1) / MPT_SAH_WAVES), dim3(64 * MPT_SAH_WAVES), 0, stream, (int)n,
                                   (const float4*)it_lo_a, (const float4*)it_hi_a, it_lo_b, it_hi_b, (const SahTask*)tasks_a, n_tasks, tasks_b, big_b, d_st,
                                   s_child, s_lo, s_hi);
            MPT_LB(hipMemcpyAsync(&h, d_st, sizeof h, hipMemcpyDeviceToHost, stream));
            MPT_LB(hipStreamSynchronize(stream));
            n_tasks = h.n_next;
            n_big = h.n_next_big;
            std::swap(tasks_a, tasks_b);
            std::swap(big_a, big_b);
            std::swap(it_lo_a, it_lo_b);
            std::swap(it_hi_a, it_hi_b);
        }
        if ((n_tasks | n_big) != 0u) return hipErrorUnknown;
    }
    const uint32_t cap = max_items + 2u;   // wide nodes <= inner nodes of the SAH tree (+ the root of a one-leaf tree)
    uint32_t *wbin, *c_nint, *c_offs;
    int4* c_picked;
    MPT_LB(sc.alloc(&wbin, cap));
    MPT_LB(sc.alloc(&c_nint, cap + 1));
    MPT_LB(sc.alloc(&c_offs, cap + 1));
    MPT_LB(sc.alloc(&c_picked, cap));
    MPT_LB(hipMalloc(&out.acc_nodes, (size_t)cap * MPT_OT_NODE_STRIDE * 16));
    uint32_t acc_nodes_n = 0, acc_depth = 0;
    {
        const CollapseAcc A = {s_child, s_lo, s_hi, olo, ohi, (int)(2 * n - 1)};
        size_t sb = 0;
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, c_nint, c_offs, (int)cap + 1, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, sb));
        hipLaunchKernelGGL(k_collapse_root, dim3(1), dim3(64), 0, stream, (const SahState*)d_st, wbin);
        uint32_t begin = 0, end = 1;
        while (begin < end && end <= cap) {
            const uint32_t cnt = end - begin, g = (cnt + B - 1) / B;
            ++acc_depth;
            hipLaunchKernelGGL(k_collapse_pick, dim3(g), dim3(B), 0, stream, A, (const uint32_t*)wbin, begin, end, c_picked, c_nint);
            MPT_LB(hipMemsetAsync(c_nint + cnt, 0, 4, stream));
            MPT_LB(hipcub::DeviceScan::ExclusiveSum(tmp, sb, c_nint, c_offs, (int)cnt + 1, stream));
            hipLaunchKernelGGL(k_collapse_emit, dim3(g), dim3(B), 0, stream, A, (int)n, (const int2*)R.range, wbin, begin, end, (const int4*)c_picked,
                               (const uint32_t*)c_offs, out.acc_nodes, cap, (const Scalars*)d_sc);
            MPT_LB(hipMemcpyAsync(pin, c_offs + cnt, 4, hipMemcpyDeviceToHost, stream));
            MPT_LB(hipStreamSynchronize(stream));
            begin = end;
            end = end + pin[0];
        }
        acc_nodes_n = begin;
    }
    MPT_LB(hipMalloc(&out.always, (size_t)MPT_ACCEL_MAX_ALWAYS * 80));
    hipLaunchKernelGGL(k_always, dim3(1), dim3(64), 0, stream, d_sc, out.prims, (const float4*)out.refleaf, out.always);
    MPT_LB(hipGetLastError());
    Scalars h;
    MPT_LB(hipMemcpyAsync(&h, d_sc, sizeof h, hipMemcpyDeviceToHost, stream));
    uint32_t n_leaves = 0;
    MPT_LB(hipMemcpyAsync(&n_leaves, leaf_id + nn, 4, hipMemcpyDeviceToHost, stream));
    MPT_LB(hipStreamSynchronize(stream));
    MPT_LB(hipMalloc(&out.mats, (size_t)std::max(h.n_mats, 1u) * 32));
    MPT_LB(hipMemcpyAsync(out.mats, mtable, (size_t)h.n_mats * 32, hipMemcpyDeviceToDevice, stream));
    MPT_LB(hipStreamSynchronize(stream));
    out.n_nodes = n_out;
    out.n_prims = n;
    out.n_mats = h.n_mats;
    out.n_acc_nodes = acc_nodes_n;
    out.acc_depth = acc_depth;
    out.n_spheres = h.n_spheres;
    out.n_always = h.n_spheres <= MPT_ACCEL_MAX_ALWAYS ? h.n_spheres : 0u;
    out.n_ref_leaves = n_leaves;
    memcpy(&out.tri_extent, &h.tri_extent, 4);
    return hipSuccess;
}

}  // namespace mpt_devbuild
