// EXPERIMENT RECORD (round 3) — not part of the product, not compiled by the Makefile.
//
// Two restructurings of the closest-first walk that were built, verified bit-identical on the GPU (tests/test_gpu_ordered.py)
// and then REJECTED on measurement (bunny x20, binned tree, 1920x1080 x 256 spp; k_ordered as shipped: 65.1 ms):
//   * k_ordered_rf — tree-walk steps with lane refill and a one-deep hit buffer (below): 85.3 ms (knob sweep 84.4-101).
//     The walkers' occupancy is fine (a walk call runs with 44-64 lanes), but refilled rays are out of phase with the lanes
//     they join: 50.8 % of the lanes take part in a node trip against 62.8 % in k_ordered, whose M1 steps run rays sorted by
//     the walk they have behind them — 23 % more node trips, plus 60-76 B of scratch for the extra per-lane state.
//   * ot_walk_u — unified trips (a node OR one primitive per trip, one record buffer, at the end of this file): 78.8 ms
//     in k_ordered, 91.3 ms with refill.  Every trip then pays the issue time of both code paths; the round trip it saves
//     per phase is not what bounds the loop.
// Also measured there: 128-byte node stride 66.3 ms (mpt_accel.h), two primitives per leaf trip 65.95 ms against 66.3.
// To rebuild: copy this file next to mpt_ordered.h, include it from mpt_hip.hip and select the kernel at launch
// (git history of round 3 has the wiring: ordered_rf_kernel / RfKnobs / MPT_OT_REFILL).
//
// mpt_refill.h — k_ordered_rf: the closest-first pipeline (mpt_ordered.h) with LANE REFILL in its tree-walk steps.  gfx950 only.
//
// Why: on scenes whose tree comes from L2 (bunny x20, 1 M triangles) the walk is 62 % of k_ordered's time and runs at
// ~50 % lane utilisation: the walks of 64 rays differ in length by 10x, and a step waits for its slowest lanes (budgets
// and parking — rings M0 / M1 of k_ordered — recover half of that, at 144 bytes of ring traffic per parked ray).  A node
// trip costs the wave ~2,500 cycles (seven divergent 16-byte loads per lane, ~470 cycles each in the vector L1, then 130
// vector instructions) whether 20 or 64 lanes take part.  So the lanes are kept busy instead:
//
//   tree-walk step (ring M, started once M holds MPT_RF_TRIGGER rays):
//     repeat
//       REFILL   lanes without a ray pop one from ring M (origin, direction, best t / primitive so far, walk state)
//       WALK     ot_walk (mpt_ordered.h) until fewer than `refill_min` lanes still walk — or `park_min` once M is empty
//       COLLECT  a lane whose walk is complete keeps (t, primitive, record position) in a one-deep HIT BUFFER in
//                registers and is free for the next ray; a lane that finishes a second ray before the buffer is emptied waits
//       SHADE    when `shade_min` lanes hold a buffered hit (or lanes wait): final check, one bounce of shading for all of
//                them at once — the rest of the record is read back from ring M, where it has stayed — survivors -> ring R
//     until M is empty and fewer than `park_min` lanes walk; those few are parked in M again with their stacks.
//
// Unfinished walks stay in their lanes: no budget ladder, no ring hop per pause.  Shading still runs at (nearly) full
// width, because hits are buffered until enough of them have come together.  Every ray sees exactly the sequence of
// tests it would see in k_ordered (its walk state is private), so images are bit-identical.
// Ring capacity: primary and ring-R steps run only while M < trigger and add <= 64 rays each, so M <= trigger + 63; a
// tree-walk step moves rays from M to R / E one for one: R <= 127 + M.  With trigger <= 256 all stay below MPT_WL_RING.
#pragma once
#include "mpt_ordered.h"

struct RfKnobs {
    uint32_t trigger;      // a tree-walk step starts when ring M holds this many rays (<= 256)
    uint32_t refill_min;   // while M has rays: the walk pauses for a refill once fewer lanes than this still walk
    uint32_t shade_min;    // buffered hits that start a shading phase
    uint32_t park_min;     // M empty and fewer walking lanes than this: park them, end the step
    uint32_t block_max;    // lanes waiting for their hit buffer that force a shading phase
};

template <bool COUNT, bool ALL_LDS>
__global__ __launch_bounds__(MPT_OT_THREADS, MPT_OT_WAVES) void k_ordered_rf(PassParams pp, AccelDev ac, OtRings ring, RfKnobs kn,
                                                                             uint32_t wl_block, uint32_t wl_min, uint32_t wl_div) {
    extern __shared__ float4 lds_raw[];
    ot_stage(pp.scene, ac, lds_raw);
    const LdsNodes lds = (LdsNodes)lds_raw;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_paths = pp.desc->total_paths;
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const OtStack st = ot_stack(ac, lds_raw, wave_id);
    const uint32_t wbase = wave_id * (MPT_OT_RINGS * MPT_WL_RING);
    const uint32_t rbase = wbase + MPT_OT_RING_R * MPT_WL_RING, ebase = wbase + MPT_OT_RING_E * MPT_WL_RING,
                   mbase = wbase + MPT_OT_RING_M * MPT_WL_RING;
    uint32_t cnt_r = 0, cnt_e = 0, cnt_m = 0;   // wave-uniform ring fills (the rings are stacks: newest first)
    uint32_t cur = 0, end = 0;
    const uint32_t n_tiles = total_paths / (pp.S * 64u);
    const uint32_t waves_per_group = (n_waves + MPT_NGROUP - 1u) / MPT_NGROUP;
    uint32_t grp = blockIdx.x & (MPT_NGROUP - 1u);
    uint32_t seen = 0;
    bool exhausted = false;
    uint32_t tile_cached = 0xFFFFFFFFu, tile_xy_cached = 0u;
    uint32_t n_rays = 0, n_paths = 0, n_flagged = 0, n_parked = 0;
    WorkCount wc = {};
#ifdef MPT_OT_TIMES
    unsigned long long ot_acc[OT_NREG] = {}, rf_n[8] = {};   // regions: select, refill / fetch, top test, walk, collect, exact walk, shade, push / park
#endif

    // a full record: what a ray needs to be shaded and to go on (64 bytes)
    auto write_record = [&](uint32_t to, const PathState& ps, const PathRngDev& g) {
        ring.od[to] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
        ring.dt[to] = make_float4(ps.d.y, ps.d.z, ps.thr.x, ps.thr.y);
        ring.tl[to] = make_float4(ps.thr.z, ps.L.x, ps.L.y, ps.L.z);
        ring.ia[to] = make_uint4(ps.path, __float_as_uint(ps.La), g.pixel, g.sample | (ps.bounce << 27));
    };
    auto read_record = [&](uint32_t at, PathState& ps, PathRngDev& g) {
        const float4 a = ring.od[at], b = ring.dt[at], cc = ring.tl[at];
        const uint4 ia = ring.ia[at];
        ps.o = f3(a.x, a.y, a.z);
        ps.d = f3(a.w, b.x, b.y);
        ps.thr = f3(b.z, b.w, cc.x);
        ps.L = f3(cc.y, cc.z, cc.w);
        ps.La = __uint_as_float(ia.y);
        ps.path = ia.x;
        ps.bounce = ia.w >> 27;
        g.pixel = ia.z;
        g.sample = ia.w & 0x07FFFFFFu;
        g.lit_seed = 0;
        if (pp.sp.rng_mode == 0) g.lit_seed = pcg_hash(pcg_hash(pp.pixel_seed[g.pixel]));
    };
    // wave64 compaction of the lanes with `go` onto the top of a ring (ballot + mbcnt prefix); returns the lane's slot
    auto ring_slot = [&](bool go, uint32_t base, uint32_t& cnt) -> uint32_t {
        const unsigned long long m = __ballot(go);
        const uint32_t to = base + cnt + wave_rank(m);
        cnt += (uint32_t)__popcll(m);
        return to;
    };

    for (;;) {
        OT_TIC();
        // ---- step choice ---------------------------------------------------------------------------------------------
        uint32_t kind = MPT_OT_NONE;  // ring to pop from; NONE = primary step
        if (cnt_e >= 64u) kind = MPT_OT_RING_E;
        else if (cnt_m >= kn.trigger) kind = MPT_OT_RING_M;
        else if (cnt_r >= 64u) kind = MPT_OT_RING_R;
        if (kind == MPT_OT_NONE) {
            if (!exhausted && cur == end) {  // guided self-scheduling of path ids, as k_wavelocal (mpt_kernels.h)
                uint32_t k = 0, blk = 0, rend = 0;
                bool got = false;
                if (lane == 0) {
                    for (uint32_t t = 0; t < MPT_NGROUP && !got; ++t) {
                        const uint32_t re = range_paths(n_tiles, pp.S, grp);
                        const uint32_t left = seen < re ? re - seen : 0u;
                        blk = (left / (wl_div * waves_per_group)) & ~63u;
                        blk = blk < wl_min ? wl_min : (blk > wl_block ? wl_block : blk);
                        if (t > 0u) blk = wl_min;
                        k = atomicAdd(&pp.ctr[MPT_CTR_CURSOR(grp)], blk);
                        if (k < re) {
                            got = true;
                            rend = re;
                        } else {
                            grp = (grp + 1u) & (MPT_NGROUP - 1u);
                            seen = 0;
                        }
                    }
                }
                got = __builtin_amdgcn_readfirstlane((int)got) != 0;
                k = __builtin_amdgcn_readfirstlane(k);
                blk = __builtin_amdgcn_readfirstlane(blk);
                rend = __builtin_amdgcn_readfirstlane(rend);
                grp = __builtin_amdgcn_readfirstlane(grp);
                seen = k;
                if (!got) {
                    exhausted = true;
                } else {
                    cur = k;
                    end = (k + blk < rend) ? k + blk : rend;
                }
            }
            if (exhausted) {  // drain: tree walks first (they feed ring R), then fresh rays, then the flagged ones
                if (cnt_m != 0u) kind = MPT_OT_RING_M;
                else if (cnt_r != 0u) kind = MPT_OT_RING_R;
                else if (cnt_e != 0u) kind = MPT_OT_RING_E;
                else break;
            }
        }

        OT_TOC(0);
        if (kind == MPT_OT_RING_M) {
#ifdef MPT_OT_TIMES
            rf_n[0]++;
#endif
            // =========================================================================================================
            // tree-walk step with lane refill
            // =========================================================================================================
            F3 wo = f3(1, 1, 1), wd = f3(1, 1, 1);
            float wT = INFINITY;
            int wW = -1;
            uint32_t wcur = MPT_OT_DONE, wsp = 0u, wat = 0u, wstat = 0u;
            bool wlost = false, wagain = false;
            bool walking = false, blocked = false;
            float hT = INFINITY;   // the hit buffer
            int hW = -1;
            uint32_t hat = 0u, hstat = 0u;   // 0 empty, 1 a finished walk (final check pending), 2 goes to the reference-order walk
            const uint32_t park_min = exhausted ? 1u : kn.park_min;

            // one bounce of shading for every buffered hit
            auto shade_buffered = [&]() {
                const bool has = hstat != 0u;
                if (__ballot(has) == 0ull) return;
                PathState ps;
                PathRngDev g;
                bool to_r = false, to_e = false;
                if (has) {
                    read_record(hat, ps, g);
                    bool exact = hstat == 2u;
                    if (!exact && hW >= 0) {
                        const OtRay r = ot_ray(ps.o, ps.d);
                        exact = !ot_final_check(ac, pp.scene, lds, ps.o, ps.d, r, hT, hW);
                    }
                    if (exact) {
                        to_e = true;
                    } else {
                        n_rays++;
                        if (shade_bounce(pp.scene, lds, pp.sp, g, ps, hT, hW)) to_r = true;
                        else store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
                    }
                }
                n_flagged += to_e ? 1u : 0u;
                const uint32_t slot_r = ring_slot(to_r, rbase, cnt_r), slot_e = ring_slot(to_e, ebase, cnt_e);
                if (to_r || to_e) write_record(to_r ? slot_r : slot_e, ps, g);
                hstat = 0u;
            };

            for (;;) {
                // ---- REFILL ---------------------------------------------------------------------------------------------
                {
                    const bool idle = !walking && !blocked;
                    const unsigned long long im = __ballot(idle);
                    const uint32_t n_idle = (uint32_t)__popcll(im);
                    const uint32_t take = n_idle < cnt_m ? n_idle : cnt_m;
                    if (take != 0u) {
                        const uint32_t rk = wave_rank(im);
                        const bool get = idle && rk < take;
                        uint32_t deepest = 0u;
                        if (get) {
                            wat = mbase + (cnt_m - 1u - rk);   // newest first: still in L2
                            const float4 a = ring.od[wat], b = ring.dt[wat];
                            const uint4 tv = ring.tv[wat];
                            wo = f3(a.x, a.y, a.z);
                            wd = f3(a.w, b.x, b.y);
                            wT = __uint_as_float(tv.x);
                            wW = (int)tv.y;
                            wcur = tv.z;
                            wsp = tv.w & 0xFFFFu;
                            wlost = (tv.w & 0x40000000u) != 0u;
                            wagain = (tv.w & 0x80000000u) != 0u;
                            walking = true;
                            deepest = wsp;
                        }
                        deepest = wave_max_u32(deepest);   // (only rays parked at the end of an earlier step bring a stack)
#pragma unroll
                        for (uint32_t k = 0; k < MPT_OT_PARK / 2u; ++k) {
                            if (2u * k >= deepest) break;
                            if (get && wsp > 2u * k) {
                                const uint4 e = ring.sk[k][wat];
                                st.lds[(2u * k) * 64u] = v2u{e.x, e.y};
                                st.lds[(2u * k + 1u) * 64u] = v2u{e.z, e.w};
                            }
                        }
                        cnt_m -= take;
#ifdef MPT_OT_TIMES
                        rf_n[2]++;
                        rf_n[3] += take;
#endif
                    }
                }
                OT_TOC(1);
                // ---- WALK -----------------------------------------------------------------------------------------------
                bool tie = false;
                if (__ballot(walking) != 0ull) {
                    const uint32_t min_act = cnt_m != 0u ? kn.refill_min : park_min;
                    const OtRay r = ot_ray(wo, wd);
                    if (!walking) wcur = MPT_OT_DONE;
                    ot_walk_u<COUNT, true, ALL_LDS>(ac, pp.scene, lds, st, wo, wd, r, wcur, wsp, wT, wW, tie, wlost, 0x7FFFFFFFu, min_act, wc);
#ifdef MPT_OT_TIMES
                    rf_n[1]++;
#endif
                }
                OT_TOC(3);
                // ---- COLLECT: finished walks -> hit buffer ----------------------------------------------------------------------
                if (walking && (tie || wcur == MPT_OT_DONE)) {
                    uint32_t status = 1u;
                    if (tie) {
                        status = 2u;                 // (whatever is left of the walk does not matter then)
                    } else if (wlost) {              // the stack dropped entries: once more from the root, now with the t found
                        if (wagain) {
                            status = 2u;
                        } else {
                            wcur = 0u;
                            wsp = 0u;
                            wlost = false;
                            wagain = true;
                            status = 0u;
                        }
                    }
                    if (status != 0u) {
                        walking = false;
                        blocked = true;
                        wstat = status;
                    }
                }
                if (blocked && hstat == 0u) {
                    hT = wT;
                    hW = wW;
                    hat = wat;
                    hstat = wstat;
                    blocked = false;
                }
                // ---- SHADE ----------------------------------------------------------------------------------------------
                const uint32_t n_hits = (uint32_t)__popcll(__ballot(hstat != 0u));
                const uint32_t n_blocked = (uint32_t)__popcll(__ballot(blocked));
                const uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
                const bool ending = cnt_m == 0u && n_walk < park_min;
                OT_TOC(4);
                if (n_hits >= kn.shade_min || n_blocked >= kn.block_max || (ending && n_hits != 0u)) {
#ifdef MPT_OT_TIMES
                    rf_n[4]++;
                    rf_n[5] += n_hits;
#endif
                    shade_buffered();   // (the only call site: one copy of the shading code in this loop)
                    if (blocked) {      // hstat is 0 everywhere now: the waiting lanes move up
                        hT = wT;
                        hW = wW;
                        hat = wat;
                        hstat = wstat;
                        blocked = false;
                    }
                    OT_TOC(6);
                    continue;           // (an ending step comes back here until nothing is buffered any more)
                }
                if (!ending) continue;
                // ---- end of the step: the few lanes still walking are parked in ring M again, with their stacks -----------------
#ifdef MPT_OT_TIMES
                rf_n[6] += n_walk;
#endif
                if (n_walk != 0u) {
                    float4 b = make_float4(0, 0, 0, 0), cc = b;
                    uint4 ia = make_uint4(0, 0, 0, 0);
                    if (walking) {   // (every read of the old records before any write: the new slots may be the old ones)
                        b = ring.dt[wat];
                        cc = ring.tl[wat];
                        ia = ring.ia[wat];
                    }
                    const uint32_t to = ring_slot(walking, mbase, cnt_m);
                    if (walking) {
                        ring.od[to] = make_float4(wo.x, wo.y, wo.z, wd.x);
                        ring.dt[to] = make_float4(wd.y, wd.z, b.z, b.w);
                        ring.tl[to] = cc;
                        ring.ia[to] = ia;
                        ring.tv[to] = make_uint4(__float_as_uint(wT), (uint32_t)wW, wcur, wsp | (wlost ? 0x40000000u : 0u) | (wagain ? 0x80000000u : 0u));
#pragma unroll
                        for (uint32_t k = 0; k < MPT_OT_PARK / 2u; ++k) {
                            if (wsp > 2u * k) {
                                const v2u e0 = st.lds[(2u * k) * 64u], e1 = st.lds[(2u * k + 1u) * 64u];
                                ring.sk[k][to] = make_uint4(e0.x, e0.y, e1.x, e1.y);
                            }
                        }
                    }
                }
                OT_TOC(7);
                break;
            }
            if ((cnt_r > cnt_e ? cnt_r : cnt_e) > MPT_WL_RING || cnt_m > MPT_WL_RING) pp.desc->overflow = 1u;  // cannot happen
            continue;
        }

        // =============================================================================================================
        // primary step / ring R step (top test) / ring E step (reference-order walk): 64 rays, shaded where they finish
        // =============================================================================================================
        PathState ps;
        PathRngDev g;
        bool valid = false;
        float T = INFINITY;
        int W = -1;
        if (kind == MPT_OT_NONE) {
            const uint32_t pchunk = range_chunk_to_path_chunk(pp, cur >> 6, grp);  // = tile * S + sample
            cur += 64u;
            uint32_t tl, sidx;
            if (pp.s_shift != 0xFFu) {
                tl = pchunk >> pp.s_shift;
                sidx = pchunk & (pp.S - 1u);
            } else {
                tl = pchunk / pp.S;
                sidx = pchunk - tl * pp.S;
            }
            if (tl != tile_cached) {
                tile_cached = tl;
                tile_xy_cached = (uint32_t)__builtin_amdgcn_readfirstlane((int)pp.tile_xy[tl]);
            }
            ps.path = pchunk * 64u + lane;
            const uint32_t px = (tile_xy_cached & 0xFFFFu) * 8u + (lane & 7u), py = (tile_xy_cached >> 16) * 8u + (lane >> 3);
            if (px < pp.width && py < pp.height) {
                gen_primary(pp, px, py, pp.sample_begin + sidx, ps, g);
                valid = true;
                n_paths++;
            }
        } else {
            const uint32_t c = kind == MPT_OT_RING_R ? cnt_r : cnt_e;
            const uint32_t take = c < 64u ? c : 64u;
            valid = lane < take;
            const uint32_t at = (kind == MPT_OT_RING_R ? rbase : ebase) + (c - take + lane);
            if (kind == MPT_OT_RING_R) cnt_r = c - take;
            else cnt_e = c - take;
            if (valid) read_record(at, ps, g);
        }
        OT_TOC(1);
        bool shade = false, to_m = false, to_e = false;
        if (kind == MPT_OT_RING_E) {
            if (valid) {  // reference-order walk (PathTracing.h:75-204 as closest_hit_resume restates it)
                uint32_t node = 0;
                closest_hit_resume<COUNT, false, false>(pp.scene, lds, ps.o, ps.d, node, T, W, 0xFFFFFFFFu, wc);
                shade = true;
            }
            OT_TOC(5);
        } else if (valid) {
            // TOP TEST: always-list spheres + the root's boxes
            bool tie = false, need = false;
            if (ot_degenerate(ps.o, ps.d, ac.o_limit)) {
                to_e = true;
            } else {
                const OtRay r = ot_ray(ps.o, ps.d);
                ot_top_test<COUNT>(ac, lds, ps.o, ps.d, r, T, W, tie, need, wc);
                if (tie) to_e = true;
                else if (need) to_m = true;
                else if (W >= 0 && !ot_final_check(ac, pp.scene, lds, ps.o, ps.d, r, T, W)) to_e = true;
                else shade = true;
            }
        }
        OT_TOC(2);
        bool to_r = false;
        if (shade) {
            n_rays++;
            if (shade_bounce(pp.scene, lds, pp.sp, g, ps, T, W)) to_r = true;
            else store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
        }
        OT_TOC(6);
        n_flagged += to_e ? 1u : 0u;
        n_parked += to_m ? 1u : 0u;
        if (__ballot(to_r || to_e || to_m) != 0ull) {
            const uint32_t slot_r = ring_slot(to_r, rbase, cnt_r), slot_e = ring_slot(to_e, ebase, cnt_e),
                           slot_m = ring_slot(to_m, mbase, cnt_m);
            if (to_r || to_e || to_m) {
                const uint32_t to = to_r ? slot_r : to_e ? slot_e : slot_m;
                write_record(to, ps, g);
                if (to_m) ring.tv[to] = make_uint4(__float_as_uint(T), (uint32_t)W, 0u, 0u);   // the walk starts at the root
            }
        }
        const uint32_t worst = cnt_r > cnt_e ? (cnt_r > cnt_m ? cnt_r : cnt_m) : (cnt_e > cnt_m ? cnt_e : cnt_m);
        if (worst > MPT_WL_RING) pp.desc->overflow = 1u;  // cannot happen (capacity note at the top)
        OT_TOC(7);
    }
#ifdef MPT_OT_TIMES
    ot_flush_walk_times(wc, lane);
    if (lane == 0) {
        for (int k = 0; k < OT_NREG; ++k) atomicAdd(&g_ot_times[k], ot_acc[k]);
        for (int k = 0; k < 8; ++k) atomicAdd(&g_rf[k], rf_n[k]);
    }
#endif
    flush_stats<COUNT>(pp.desc, n_rays, n_paths, wc);
    {
        unsigned long long a = n_flagged, b = n_parked;
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
        }
        if (lane == 0) {
            if (a) atomicAdd(&pp.desc->flagged, a);
            if (b) atomicAdd(&pp.desc->parked, b);
        }
    }
}

// ---- ot_walk_u (was in mpt_ordered.h, in front of ot_final_check; referenced above) -----------------------------------
#if 0
// The same walk with UNIFIED trips (-DMPT_OT_UNIFIED): every trip a lane takes ONE unit of work — a node (four box tests) or
// one primitive of the leaf it holds — so lanes in different phases advance side by side, and the wave pays one memory
// round trip per trip instead of one per phase.  (The while-while form above keeps box tests and primitive tests in
// loops of their own: on bunny x20 62 % of the lanes take part in a node trip and 35 % in a leaf trip, each trip a round
// trip to L1 / L2 of ~2,500 cycles.)  One record buffer serves both kinds: the 7 float4 of a node or the 3 of a
// primitive, so the loads of a trip are the same seven instructions whatever the lanes hold.  A ray sees the same
// sequence of tests as in ot_walk; `budget` counts trips of this loop.
template <bool COUNT, bool BUDGETED, bool ALL_LDS>
__device__ __forceinline__ bool ot_walk_u(const AccelDev& ac, const SceneDev& sc, LdsNodes lds, const OtStack& st, F3 o, F3 d,
                                          const OtRay& r, uint32_t& cur, uint32_t& sp, float& T, int& W, bool& tie,
                                          bool& overflow, uint32_t budget, uint32_t min_active, WorkCount& wc) {
    uint32_t trips = 0;   // wave-uniform
    for (;;) {
        const bool is_node = cur < MPT_OT_LEAF, is_leaf = !is_node && cur != MPT_OT_DONE;
        const unsigned long long act = __ballot(is_node || is_leaf);
        if (act == 0ull) break;
        if (BUDGETED && (trips >= budget || (uint32_t)__popcll(act) < min_active)) break;
        if (BUDGETED) trips++;
        const uint32_t first = cur & 0x07FFFFFFu;
        float4 b0, b1, b2, b3 = make_float4(0, 0, 0, 0), b4 = b3, b5 = b3, b6 = b3;
        b0 = b1 = b2 = b3;
        if (is_node || is_leaf) {
            const bool in_lds = is_node ? (ALL_LDS || cur < ac.n_lds_nodes) : first < sc.n_lds_prims;
            if (in_lds) {
                const LdsNodes q = lds + (is_node ? 7u * cur : sc.lds_prim_off + 3u * first);
                const v4f a = q[0], b = q[1], c = q[2];
                b0 = make_float4(a.x, a.y, a.z, a.w);
                b1 = make_float4(b.x, b.y, b.z, b.w);
                b2 = make_float4(c.x, c.y, c.z, c.w);
                if (is_node) {
                    const v4f e = q[3], f = q[4], g = q[5], h = q[6];
                    b3 = make_float4(e.x, e.y, e.z, e.w);
                    b4 = make_float4(f.x, f.y, f.z, f.w);
                    b5 = make_float4(g.x, g.y, g.z, g.w);
                    b6 = make_float4(h.x, h.y, h.z, h.w);
                }
            } else {
                const float4* q = is_node ? ac.nodes + MPT_OT_NODE_STRIDE * (size_t)cur : sc.prims + 3u * (size_t)first;
                b0 = q[0];
                b1 = q[1];
                b2 = q[2];
                if (is_node) {
                    b3 = q[3];
                    b4 = q[4];
                    b5 = q[5];
                    b6 = q[6];
                }
            }
        }
        if (is_node) {
            const uint4 ref = make_uint4(__float_as_uint(b6.x), __float_as_uint(b6.y), __float_as_uint(b6.z), __float_as_uint(b6.w));
            const float lim = ot_cull_limit(T, ac);
            uint32_t k0 = ot_box_key(r, b0.x, b1.x, b2.x, b3.x, b4.x, b5.x, ref.x, lim, 0u);
            uint32_t k1 = ot_box_key(r, b0.y, b1.y, b2.y, b3.y, b4.y, b5.y, ref.y, lim, 1u);
            uint32_t k2 = ot_box_key(r, b0.z, b1.z, b2.z, b3.z, b4.z, b5.z, ref.z, lim, 2u);
            uint32_t k3 = ot_box_key(r, b0.w, b1.w, b2.w, b3.w, b4.w, b5.w, ref.w, lim, 3u);
            if (COUNT) {
                wc.node_visits++;
                wc.aabb_hits += (k0 < MPT_OT_KEY_MISS) + (k1 < MPT_OT_KEY_MISS) + (k2 < MPT_OT_KEY_MISS) + (k3 < MPT_OT_KEY_MISS);
                if (first_active_lane()) wc.node_iters++;
            }
#ifdef MPT_OT_TIMES
            if (first_active_lane()) wc.ot_node_trips++;
            wc.ot_node_lanes++;
#endif
            ot_sort2(k0, k1);
            ot_sort2(k2, k3);
            ot_sort2(k0, k2);
            ot_sort2(k1, k3);
            ot_sort2(k1, k2);
            if (k0 < MPT_OT_KEY_MISS) {
                cur = ot_pick(ref, k0);
                if (k1 < MPT_OT_KEY_MISS) ot_push_sorted(st, sp, ref, k1, k2, k3, overflow);
            } else {
                cur = ot_pop_next(st, sp, lim);
            }
        } else if (is_leaf) {
            Prim3 pr;
            pr.p0 = b0;
            pr.p1 = b1;
            pr.p2 = b2;
            if (COUNT && first_active_lane()) wc.prim_iters++;
#ifdef MPT_OT_TIMES
            if (first_active_lane()) wc.ot_leaf_trips++;
            wc.ot_leaf_lanes++;
#endif
            if (!(ac.n_always != 0u && prim_type(pr.p0) == 0)) {   // spheres are on the always list
                if (COUNT) wc.prim_tests++;
                ot_test_prim(pr, first, o, d, T, W, tie);
            }
            const uint32_t left = (cur >> 27) & 15u;   // primitives of the leaf behind this one
            cur = left != 0u ? (MPT_OT_LEAF | ((left - 1u) << 27) | (first + 1u)) : ot_pop_next(st, sp, ot_cull_limit(T, ac));
        }
    }
    return cur == MPT_OT_DONE;
}

#endif

// ---- two primitives per leaf trip (was in ot_walk's leaf phase under MPT_OT_LEAF2) ----------------------------------------
#if 0
   // two primitives per trip: both records are requested before either is tested (leaves hold <= 2
                      // primitives with the product's builders: one memory round trip per leaf instead of two)
            for (uint32_t k = 0; k < count; k += 2u) {
                const bool two = k + 1u < count;
                const Prim3 pa = load_prim(sc, lds, first + k);
                Prim3 pb = pa;
                if (two) pb = load_prim(sc, lds, first + k + 1u);
                if (COUNT && first_active_lane()) wc.prim_iters++;
                if (!(ac.n_always != 0u && prim_type(pa.p0) == 0)) {
                    if (COUNT) wc.prim_tests++;
                    ot_test_prim(pa, first + k, o, d, T, W, tie);
                }
                if (two && !(ac.n_always != 0u && prim_type(pb.p0) == 0)) {
                    if (COUNT) wc.prim_tests++;
                    ot_test_prim(pb, first + k + 1u, o, d, T, W, tie);
                }
            }
#endif

// ---- top-up of the last tree-walk ring's step (MPT_OT_CARRY, round 3): unfinished walks stay in their lanes and the next
// step fills the free lanes from the ring instead of parking 24 rays.  Measured slower: scene.xml 21.1 -> 24.4 ms, bunny x20
// 64.4 -> 72.0 ms (binned tree, 256 spp; bit-identical output): the 13 carried registers spill across shade_bounce
// (25 VGPRs to scratch) and the hits are shaded at <= 40 of 64 lanes however the step ends.  The patch against mpt_ordered.h:
#if 0
diff --git a/metalpathtracer_amd/csrc/mpt_ordered.h b/metalpathtracer_amd/csrc/mpt_ordered.h
index 13455a6..b123c09 100644
--- a/metalpathtracer_amd/csrc/mpt_ordered.h
+++ b/metalpathtracer_amd/csrc/mpt_ordered.h
@@ -548,11 +548,25 @@ __global__ __launch_bounds__(MPT_OT_THREADS, MPT_OT_WAVES) void k_ordered(PassPa
     WorkCount wc = {};
 #ifdef MPT_OT_TIMES
     unsigned long long ot_acc[OT_NREG] = {}, ot_steps[MPT_OT_RINGS + 1] = {}, ot_lanes[MPT_OT_RINGS + 1] = {};
+#endif
+#ifdef MPT_OT_CARRY
+    // unfinished walks of a step of the last tree-walk ring stay in their lanes (registers + LDS stack column) and the next
+    // step tops the wave up from the ring, instead of parking them (144 bytes written and read back per ray)
+    bool carry = false;
+    uint32_t n_carry = 0u;  // wave-uniform
+    F3 c_o = f3(0, 0, 0), c_d = f3(0, 0, 0);
+    float c_tx = 0.f, c_ty = 0.f, c_T = 0.f;
+    int c_W = -1;
+    uint32_t c_cur = 0u, c_sp = 0u, c_at = 0u;
 #endif
     for (;;) {
         OT_TIC();
         // ---- step choice: a full wave of the most advanced kind of work; else new paths; else what is left ----------------
         uint32_t kind = MPT_OT_NONE;  // ring to pop from; NONE = primary step
+#ifdef MPT_OT_CARRY
+        if (n_carry != 0u) kind = MPT_OT_RINGS - 1u;
+        else
+#endif
         if (cnt[MPT_OT_RING_E] >= 64u) kind = MPT_OT_RING_E;
 #pragma unroll
         for (int k = (int)MPT_OT_RINGS - 1; k >= (int)MPT_OT_RING_M; --k)  // the longest walks first
@@ -641,9 +655,17 @@ __global__ __launch_bounds__(MPT_OT_THREADS, MPT_OT_WAVES) void k_ordered(PassPa
 #pragma unroll
             for (uint32_t k = 0; k < MPT_OT_RINGS; ++k)
                 if (k == kind) c = cnt[k];
+#ifdef MPT_OT_CARRY
+            const uint32_t room = 64u - n_carry;
+            const uint32_t take = c < room ? c : room;
+            const uint32_t rk = wave_rank(~__ballot(carry));
+            valid = !carry && rk < take;
+            at = wbase + kind * MPT_WL_RING + (c - take + rk);
+#else
             const uint32_t take = c < 64u ? c : 64u;
             valid = lane < take;
             at = wbase + kind * MPT_WL_RING + (c - take + lane);
+#endif
 #pragma unroll
             for (uint32_t k = 0; k < MPT_OT_RINGS; ++k)
                 if (k == kind) cnt[k] = c - take;
@@ -664,7 +686,11 @@ __global__ __launch_bounds__(MPT_OT_THREADS, MPT_OT_WAVES) void k_ordered(PassPa
                 }
             }
             if (kind >= MPT_OT_RING_M) {  // a parked walk brings its stack along: back into this lane's LDS column
+#ifdef MPT_OT_CARRY
+                if (!valid && !carry) walk_cur = MPT_OT_DONE;
+#else
                 if (!valid) walk_cur = MPT_OT_DONE;
+#endif
                 const uint32_t deepest = wave_max_u32(valid ? walk_sp : 0u);
 #pragma unroll
                 for (uint32_t k = 0; k < MPT_OT_PARK / 2u; ++k) {
@@ -675,6 +701,24 @@ __global__ __launch_bounds__(MPT_OT_THREADS, MPT_OT_WAVES) void k_ordered(PassPa
                         st.lds[(2u * k + 1u) * 64u] = v2u{e.z, e.w};
                     }
                 }
+#ifdef MPT_OT_CARRY
+                if (carry) {  // (its stack column was not touched)
+                    ps.o = c_o;
+                    ps.d = c_d;
+                    ps.thr.x = c_tx;
+                    ps.thr.y = c_ty;
+                    T = c_T;
+                    W = c_W;
+                    walk_cur = c_cur;
+                    walk_sp = c_sp & 0xFFFFu;
+                    walk_lost = (c_sp & 0x40000000u) != 0u;
+                    walk_again = (c_sp & 0x80000000u) != 0u;
+                    at = c_at;
+                    valid = true;
+                }
+                carry = false;
+                n_carry = 0u;
+#endif
             }
         }
         // the rest of a record is not needed by the tree walk: ring M steps load it afterwards
@@ -749,6 +793,28 @@ __global__ __launch_bounds__(MPT_OT_THREADS, MPT_OT_WAVES) void k_ordered(PassPa
             else
                 done = ot_walk<COUNT, false, ALL_LDS>(ac, pp.scene, lds, st, ps.o, ps.d, r, walk_cur, walk_sp, T, W, tie, walk_lost,
                                                       0xFFFFFFFFu, 0u, wc);
+#ifdef MPT_OT_CARRY
+            if (kind == MPT_OT_RINGS - 1u && !exhausted) {
+                const bool unf = walking && !tie && !done;
+                const uint32_t nu = (uint32_t)__popcll(__ballot(unf));
+                if (nu != 0u && cnt[MPT_OT_RINGS - 1u] + nu >= 64u) {
+                    carry = unf;
+                    n_carry = nu;
+                    if (unf) {
+                        c_o = ps.o;
+                        c_d = ps.d;
+                        c_tx = ps.thr.x;
+                        c_ty = ps.thr.y;
+                        c_T = T;
+                        c_W = W;
+                        c_cur = walk_cur;
+                        c_sp = walk_sp | (walk_lost ? 0x40000000u : 0u) | (walk_again ? 0x80000000u : 0u);
+                        c_at = at;
+                        walking = false;
+                    }
+                }
+            }
+#endif
             if (walking) {
                 if (kind == walk_kind) load_rest();   // (a step of ring M; an in-place walk has everything in registers)
                 if (tie) dest = MPT_OT_RING_E;   // (whatever is left of the walk does not matter then)
#endif
