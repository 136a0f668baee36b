// EXPERIMENT RECORD (rounds 3-5) — not part of the product, not compiled by the Makefile.
//
// The branches of compile-time experiment flags that were REJECTED on measurement and pruned from metalpathtracer_amd/csrc/ in round 5
// (VERDICT r4 housekeeping), as they stood when they were removed: every line below was behind one of
//   mpt_ordered.h   MPT_OT_OCT (eight fresh-ray rings keyed by the direction octant: bunny x20 71.4 ms against 64.3), MPT_OT_QNODES (64-byte
//                   quantised global nodes: 68.8 against 64.4), MPT_OT_NT_POP (non-temporal ring pops: 67.9 against 64.3), MPT_OT_NT_PUSH
//                   (non-temporal ring pushes, round 5: 63.7 against 57.3), MPT_OT_TOUCH (touch of the leaf's first primitive: 60.2 against
//                   59.85), MPT_OT_PREFETCH (primitive k + 1 loaded while k is tested: 28.7 against 28.1), MPT_OT_DIAG_DUP / _VALU (the dummy-load
//                   and dummy-ALU sensitivity builds of DESIGN.md 5), MPT_OT_NO_INPLACE (no in-place walks, round 5: 58.7 against 57.2),
//                   MPT_OT_DIAG_NOFINAL (the what-if build without the final check, round 5: 55.8 against 57.2 = what the check costs),
//                   and the "off" sides of the adopted MPT_OT_HITRING / MPT_OT_LEAF2 / MPT_OT_REFBOX / MPT_WL_DIET;
//   mpt_device.h    MPT_WL_PREFETCH (18.40 against 17.62 ms), MPT_WL_TRI_FLAT (19.54 against 19.03);
//   mpt_kernels.h   MPT_DIET_NOSLOT / MPT_DIET_RINGPLUS (round 4's traffic what-ifs: -3.2 % / +3.0 %), the "off" sides of MPT_WL_DIET /
//                   MPT_WL_HITRING.
// The lines are what the mini-unifdef of round 5 took out, in file order, without their surrounding code: git history (round 4's
// final commit 2992c37) has the files whole.  DESIGN.md 5 and docs/HISTORY.md carry the measurements.
#if 0

// ===== mpt_ordered.h =====
#define MPT_OT_OCT 0
#define MPT_OT_REFBOX 1   // 1: the final check reads the reference leaf's box by primitive index (one round trip); 0: through the primitive record (two)
#define MPT_OT_LEAF2 1   // 1: the first two primitives of a leaf are loaded together
#define MPT_OT_HITRING 1
#define MPT_OT_QNODES 0
        const float4* qq = ac.qnodes + 4u * (size_t)n;
        const float4 w0 = qq[0], w1 = qq[1], w2 = qq[2], w3 = qq[3];
        {   // (no branch on the loaded data: a test of w0 before the other three loads are issued costs a second round trip per visit —
            //  measured: 72.6 ms instead of 64.7.  A node that cannot be quantised gets boxes that every ray enters, see k_quantize_nodes)
            const uint32_t qlx = __float_as_uint(w1.z), qly = __float_as_uint(w1.w), qlz = __float_as_uint(w2.x), qhx = __float_as_uint(w2.y),
                           qhy = __float_as_uint(w2.z), qhz = __float_as_uint(w2.w);
            nd.ref = make_uint4(__float_as_uint(w3.x), __float_as_uint(w3.y), __float_as_uint(w3.z), __float_as_uint(w3.w));
#define OT_DEQ(word, k, s_, o_) fmaf((float)(((word) >> (8 * (k))) & 255u), (s_), (o_))
            nd.lx = make_float4(OT_DEQ(qlx, 0, w0.w, w0.x), OT_DEQ(qlx, 1, w0.w, w0.x), OT_DEQ(qlx, 2, w0.w, w0.x), OT_DEQ(qlx, 3, w0.w, w0.x));
            nd.hx = make_float4(OT_DEQ(qhx, 0, w0.w, w0.x), OT_DEQ(qhx, 1, w0.w, w0.x), OT_DEQ(qhx, 2, w0.w, w0.x), OT_DEQ(qhx, 3, w0.w, w0.x));
            nd.ly = make_float4(OT_DEQ(qly, 0, w1.x, w0.y), OT_DEQ(qly, 1, w1.x, w0.y), OT_DEQ(qly, 2, w1.x, w0.y), OT_DEQ(qly, 3, w1.x, w0.y));
            nd.hy = make_float4(OT_DEQ(qhy, 0, w1.x, w0.y), OT_DEQ(qhy, 1, w1.x, w0.y), OT_DEQ(qhy, 2, w1.x, w0.y), OT_DEQ(qhy, 3, w1.x, w0.y));
            nd.lz = make_float4(OT_DEQ(qlz, 0, w1.y, w0.z), OT_DEQ(qlz, 1, w1.y, w0.z), OT_DEQ(qlz, 2, w1.y, w0.z), OT_DEQ(qlz, 3, w1.y, w0.z));
            nd.hz = make_float4(OT_DEQ(qhz, 0, w1.y, w0.z), OT_DEQ(qhz, 1, w1.y, w0.z), OT_DEQ(qhz, 2, w1.y, w0.z), OT_DEQ(qhz, 3, w1.y, w0.z));
#undef OT_DEQ
            // an empty child slot: both x planes at +inf, as in the float form (no walked ray enters it)
            const float inf = __uint_as_float(0x7F800000u);
            if (nd.ref.x == 0xFFFFFFFFu) nd.lx.x = nd.hx.x = inf;
            if (nd.ref.y == 0xFFFFFFFFu) nd.lx.y = nd.hx.y = inf;
            if (nd.ref.z == 0xFFFFFFFFu) nd.lx.z = nd.hx.z = inf;
            if (nd.ref.w == 0xFFFFFFFFu) nd.lx.w = nd.hx.w = inf;
            return nd;
        }
                         "global_load_dwordx4 %0, %3, off offset:48\n\tglobal_load_dwordx4 %1, %3, off offset:64\n\tglobal_load_dwordx4 %2, %3, off offset:80\n\t"
                         "global_load_dwordx4 %0, %3, off offset:96\n\t"
    {
        float x = nd.lx.x;
#pragma unroll
        for (int k = 0; k < MPT_OT_DIAG_VALU; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
        asm volatile("" ::"v"(x));
    }
    float touch = 0.0f;
            if (cur != MPT_OT_DONE && cur >= MPT_OT_LEAF && (cur & 0x07FFFFFFu) >= sc.n_lds_prims) touch = sc.prims[3u * (size_t)(cur & 0x07FFFFFFu)].x;
            Prim3 nxt = load_prim(sc, lds, first);
            for (uint32_t k = 0; k < count; ++k) {
                const Prim3 pr = nxt;
                if (k + 1u < count) nxt = load_prim(sc, lds, first + k + 1u);
        asm volatile("" ::"v"(touch));
    return true;
    const Prim3 pr = load_prim(sc, lds, (uint32_t)W);
    float4 n0, n1;
    if (ac.n_always != 0u && prim_type(pr.p0) == 0) {  // a sphere of the always list: its leaf box is in LDS
        const LdsNodes q = lds + ac.lds_always_off + 5u * (uint32_t)__float_as_int(pr.p1.z) + 3u;
        const v4f a = q[0], b = q[1];
        n0 = make_float4(a.x, a.y, a.z, 0.0f);
        n1 = make_float4(b.x, b.y, b.z, 0.0f);
    } else {
        const uint32_t leaf = prim_ref_leaf(pr.p0);
        n0 = ac.refleaf[2u * (size_t)leaf];
        n1 = ac.refleaf[2u * (size_t)leaf + 1u];
    }
    const v4f_nt v = __builtin_nontemporal_load((const v4f_nt*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#define MPT_OT_NT_PUSH 0
    const v4f_nt w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (v4f_nt*)p);
            const uint4 ia = ot_pop4u(ring.ia() + at);
            ps.thr.z = __uint_as_float(ia.x);
            ps.path = ia.y;
            g.pixel = ia.z;
            ps.bounce = ia.w & 0xFFu;
            g.sample = sample_of_path(pp, ps.path);
            ps.L = f3(0.0f, 0.0f, 0.0f);
            ps.La = 0.0f;
            if ((ia.w & MPT_RING_HAS_LIGHT) != 0u) {
                const float4 cc = ot_pop4(ring.tl() + at);
                ps.L = f3(cc.x, cc.y, cc.z);
                ps.La = cc.w;
            }
            const float4 cc = ot_pop4(ring.tl() + at);
            const uint4 ia = ot_pop4u(ring.ia() + at);
            ps.thr.z = cc.x;
            ps.L = f3(cc.y, cc.z, cc.w);
            ps.La = __uint_as_float(ia.y);
            ps.path = ia.x;
            ps.bounce = ia.w >> 27;
            g.pixel = ia.z;
            g.sample = ia.w & 0x07FFFFFFu;
            if (false) {
            if (shade_bounce(pp.scene, lds, pp.sp, g, ps, T, W))
                dest = MPT_OT_RING_R + (MPT_OT_NR > 1u ? (ps.d.x < 0.0f ? 1u : 0u) | (ps.d.y < 0.0f ? 2u : 0u) | (ps.d.z < 0.0f ? 4u : 0u) : 0u);
            else store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
                {
                    const bool lit = ring_has_light(ps);
                    ring.ia()[to] = make_uint4(__float_as_uint(ps.thr.z), ps.path, g.pixel, ps.bounce | (lit ? MPT_RING_HAS_LIGHT : 0u));
                    if (lit) ring.tl()[to] = make_float4(ps.L.x, ps.L.y, ps.L.z, ps.La);
                }
                ring.tl()[to] = make_float4(ps.thr.z, ps.L.x, ps.L.y, ps.L.z);
                ring.ia()[to] = make_uint4(ps.path, __float_as_uint(ps.La), g.pixel, g.sample | (ps.bounce << 27));
// ===== mpt_device.h =====
    Prim3 nxt = load_prim(sc, lds, first);
        const Prim3 pr = nxt;
        if (k + 1u < count) nxt = load_prim(sc, lds, first + k + 1u);
            // (experiment) without early-outs, as ot_test_prim in mpt_ordered.h: the same operations give the same values; where the
            // reference leaves early the rest is computed from garbage and discarded by `hit`
            {
                const float f = 1.0f / a;
                const F3 s = o - v0;
                const float u = f * dot3(s, h);
                const F3 q = cross3(s, e1);
                const float v = f * dot3(d, q);
                const float tt = f * dot3(e2, q);
                const bool hit = fabsf(a) > 1e-5f && u >= 0.0f && u <= 1.0f && v >= 0.0f && u + v <= 1.0f && tt > 0.0001f && tt < best_t;
                best_t = hit ? tt : best_t;
                best_prim = hit ? (int)(first + k) : best_prim;
            }
            if (false) {
// ===== mpt_kernels.h =====
    if (r == 12345.678f) slots[path] = make_float4(r, g, b, a);   // (never true for a clamped colour: the store is compiled, not executed)
    return;
#define MPT_WL_DIET 1
    ring.tl()[at] = make_float4(ps.thr.z, ps.L.x, ps.L.y, ps.L.z);
    ring.ia()[at] = make_uint4(ps.path, __float_as_uint(ps.La), g.pixel, g.sample | (ps.bounce << 27));
    const float4 cc = ring.tl()[at];
    ps.thr = f3(b.z, b.w, cc.x);
    ps.L = f3(cc.y, cc.z, cc.w);
    ps.La = __uint_as_float(ia.y);
    ps.path = ia.x;
    ps.bounce = ia.w >> 27;
    g.pixel = ia.z;
    g.sample = ia.w & 0x07FFFFFFu;
#define MPT_WL_HITRING 1
                ring_pop(pp, ring, at, ps, g);
                alive = shade_bounce(pp.scene, lds_nodes, pp.sp, g, ps, best_t, best_prim);
                if (!alive)
                    store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
                ring_push(ring, at, ps, g);
                ring.tv()[at] = make_uint4(ps.path, 0u, 0u, 0u);
                ring_push(ring, at, ps, g);
// ===== mpt_ordered.h: k_quantize_nodes (MPT_OT_QNODES) =====
// 112-byte float nodes -> 64-byte nodes (see AccelDev).  One thread per node; runs once per scene (mpt_upload_scene, mpt_build_and_upload).
__global__ void k_quantize_nodes(const float4* nodes, uint32_t n_nodes, float4* qnodes, uint32_t* n_float /* nodes whose boxes could not be quantised */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const float4* q = nodes + MPT_OT_NODE_STRIDE * (size_t)i;
    float lo[3][4], hi[3][4];
    uint32_t ref[4];
    {
        const float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6];
        const float L[3][4] = {{a.x, a.y, a.z, a.w}, {b.x, b.y, b.z, b.w}, {c.x, c.y, c.z, c.w}};
        const float H[3][4] = {{d.x, d.y, d.z, d.w}, {e.x, e.y, e.z, e.w}, {f.x, f.y, f.z, f.w}};
        for (int ax = 0; ax < 3; ++ax)
            for (int j = 0; j < 4; ++j) lo[ax][j] = L[ax][j], hi[ax][j] = H[ax][j];
        ref[0] = __float_as_uint(g.x), ref[1] = __float_as_uint(g.y), ref[2] = __float_as_uint(g.z), ref[3] = __float_as_uint(g.w);
    }
    bool ok = true;
    float o[3] = {0.0f, 0.0f, 0.0f}, s[3] = {1.0f, 1.0f, 1.0f};
    uint32_t ql[3] = {0u, 0u, 0u}, qh[3] = {0u, 0u, 0u};
    for (int ax = 0; ax < 3; ++ax) {
        float mn = INFINITY, mx = -INFINITY;
        for (int j = 0; j < 4; ++j) {
            if (ref[j] == 0xFFFFFFFFu) continue;
            if (!(isfinite(lo[ax][j]) && isfinite(hi[ax][j]) && lo[ax][j] <= hi[ax][j])) ok = false;
            mn = fminf(mn, lo[ax][j]);
            mx = fmaxf(mx, hi[ax][j]);
        }
        if (!(mn <= mx)) {   // no child at all (cannot happen) or nothing finite
            ok = false;
            mn = mx = 0.0f;
        }
        float sc = (mx - mn) / 255.0f;
        if (!(sc > 1e-30f)) sc = 1e-30f;
        for (int k = 0; k < 64 && fmaf(255.0f, sc, mn) < mx; ++k) sc = nextafterf(sc, INFINITY);
        if (!isfinite(sc) || !isfinite(fmaf(255.0f, sc, mn)) || fmaf(255.0f, sc, mn) < mx) ok = false;
        o[ax] = mn;
        s[ax] = sc;
        for (int j = 0; j < 4; ++j) {
            uint32_t a = 255u, b = 0u;   // an empty slot: decoded planes are overwritten with +inf by the walk
            if (ref[j] != 0xFFFFFFFFu && ok) {
                float fa = floorf((lo[ax][j] - mn) / sc), fb = ceilf((hi[ax][j] - mn) / sc);
                fa = fminf(fmaxf(fa, 0.0f), 255.0f);
                fb = fminf(fmaxf(fb, 0.0f), 255.0f);
                a = (uint32_t)fa;
                b = (uint32_t)fb;
                while (a > 0u && fmaf((float)a, sc, mn) > lo[ax][j]) --a;      // the decoded plane, in the walk's own arithmetic,
                while (b < 255u && fmaf((float)b, sc, mn) < hi[ax][j]) ++b;    // must not cut into the float box
                if (fmaf((float)a, sc, mn) > lo[ax][j] || fmaf((float)b, sc, mn) < hi[ax][j]) ok = false;
            }
            ql[ax] |= a << (8 * j);
            qh[ax] |= b << (8 * j);
        }
    }
    float4* out = qnodes + 4u * (size_t)i;
    if (!ok) {   // a plane that is not finite, or an extent that overflows (degenerate input): boxes from -2.5e38 to +2.5e38 on every axis —
                 // every ray enters every child, whose own node or primitives are then tested as always (conservative; counted)
        atomicAdd(n_float, 1u);
        for (int ax = 0; ax < 3; ++ax) {
            o[ax] = -2.5e38f;
            s[ax] = 1.9607843e36f;   // (5e38 / 255)
            ql[ax] = 0u;
            qh[ax] = 0xFFFFFFFFu;
        }
    }
    out[0] = make_float4(o[0], o[1], o[2], s[0]);
    out[1] = make_float4(s[1], s[2], __uint_as_float(ql[0]), __uint_as_float(ql[1]));
    out[2] = make_float4(__uint_as_float(ql[2]), __uint_as_float(qh[0]), __uint_as_float(qh[1]), __uint_as_float(qh[2]));
    out[3] = make_float4(__uint_as_float(ref[0]), __uint_as_float(ref[1]), __uint_as_float(ref[2]), __uint_as_float(ref[3]));
}


#endif
