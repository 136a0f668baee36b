// EXPERIMENT RECORD (round 5) — not part of the product, not compiled by the Makefile.
//
// Restructurings of the trace kernels that were built, verified bit-identical on the GPU (the full -m gpu suite passed with each) and then
// REJECTED on measurement this round.  Same-box A/B, min of 3-4 serial 256-spp renders (tools/gpu_ab.sh), logs under gpurun_out/r05/.
//
// 1. k_wavelocal: "the hits stay in their lanes at the end of a pass" (MPT_WL_CARRY) — VERDICT r4 item 4 asked for a shorter tail.
//    Idea: once a wave is out of path ids, its step parked nothing and its rings are empty, shade the hits where they are and trace
//    their bounce rays at once, in a loop, instead of pushing them to ring 0 and popping them in the next step (two memory round trips
//    and a step's bookkeeping per bounce generation of the deepest paths).  The loop is the block below, placed in wavelocal_body right
//    behind `const unsigned long long am = __ballot(alive), pm = __ballot(parked);`.
//    Measured (gpurun_out/r05/s4_ab.log, s4_tail.log): scene.xml 17.36-17.45 ms against 16.90 without it (+3 %: the second inlined
//    copy of the closest-hit loop and of the shading moved the register allocation of the MAIN loop: 77 -> 80 VGPRs, 0 -> 8-12 bytes of
//    scratch); an emulated 1/8 shard's serial step 2.60 against 2.55 ms; the frame protocol at 1280x720 2.475 against 2.483 ms per
//    literal frame, 1.555 against 1.566 philox — i.e. NO gain where the tail is nearly everything.  The ring hops are not what the tail
//    waits for.  (A first version that kept the carried hits in loop-carried variables across the step loop cost 80 VGPRs + 16 bytes
//    of scratch before it was measured.)
//
// 2. k_ordered: six per-ray plane offsets instead of three (MPT_OT_SIGNSEL=2 in mpt_ordered.h, still selectable): 57.91 against 57.47 ms
//    on bunny x20 — six additions per node instead of seven, but three more live registers (SGPR spills 71 -> 98).
//
// 3. k_ordered occupancy-for-LDS trades (-DMPT_OT_THREADS / -DMPT_OT_WAVES, gpurun_out/r05/s2_ab_*.log), bunny x20 /
//    the 1 M-triangle shard of configs[4]: 2 x 512 threads, 4 waves/SIMD, 410 nodes in LDS: 65.0 / 130.6 ms; 1 x 1024, 4 waves/SIMD, 850
//    nodes: 64.9 / 129.9; against 57.9 / 118.0 for 5 x 256 (5 waves/SIMD, 100 nodes = 39 % / 32 % of the node visits served from LDS):
//    a fifth wave per SIMD is worth 12 %, eight times the LDS nodes 0-1 %.  2 x 768, 6 waves/SIMD at 80 VGPRs (24 spilled, 76 B scratch):
//    56.8 / 120.5 (-2 % / +2 %: not a default).  Stacks of 4 / 6 entries (MPT_OT_STACK; more nodes in LDS): 107 / 66 ms — the
//    re-walks of overflowed stacks cost far more than the nodes buy.  No primitives in LDS (MPT_OT_LDS_PRIMS=0, 36 more nodes): +-0.
//
// 4. IEEE division: the reciprocal chain without v_div_scale / v_div_fixup (mpt_device.h rcp_chain, proven bit-identical to 1.0f / x on
//    every operand with 2^-126 <= |x| <= 2^126 by k_kat_rcp) — unguarded (wrong outside that range; MPT_FAST_RCP=2): scene.xml 16.38
//    against 16.63-16.69 ms (-1.7 %: the price of the 22 divisions), bunny x20 60.05 against 58.9 (+1.8 %: register allocation);
//    guarded per lane (MPT_FAST_RCP=1): 16.56 (-0.5 %) / 59.06 (+0.2 %).  See DESIGN.md 5 for the wave-uniform guard's numbers.

// 5. k_wavelocal, utilisation rules where it had none (the rule that pays in k_ordered: walk steps end when fewer than N lanes still walk):
//    ring-0 steps (the bounce rays of 64 shaded hits; budget 8 trips, no rule): (8; 24, 24) scene.xml on the device-built tree 16.19 against
//    16.41-16.48 ms (-1.6 %), but on the reference's tree 19.19 against 18.89-18.96 (+1.4 %) and Cornell 34.31 against 32.91-32.97 (+4 %);
//    (1000; 32, 24): 16.19 / 19.47 / 34.35 — not a default (gpurun_out/r05/s13_ab_wl*.log).  PRIMARY steps (no budget, traced to the end;
//    MPT_PRIMARY_MIN_ACTIVE = 12 / 16 / 24 / 32: stragglers parked in ring 1): 16.59-16.73 against 16.34-16.47, 19.4-19.5 against 18.8-19.0,
//    33.1 against 32.9 — worse everywhere (s14_ab_pma*.log); the code is gone again.
//
// 6. k_wavelocal: another early-leave threshold of the box loop for the steps of the walk ring (the loop leaves for the leaf phase when fewer
//    than 1/N of the lanes that entered still search; N = 8 everywhere): N = 2 / 3 / 4 / 16 for ring 1 only: scene.xml 16.71 / 16.51 / 16.30 /
//    16.47 against 16.35-16.39 ms, reference tree 19.83 / 19.24 / 18.99 / 19.07 against 18.93-19.00, Cornell 32.86 / 33.15 / 33.07 / 32.98 against
//    32.72-32.86 (s16_ab_le*.log): nothing.  k_ordered: its threshold 1/3 instead of 1/2: 54.95 against 54.0 ms [115.5 against 114.0-114.5];
//    three walk rings 55.3 [118.7]; ONE walk ring 54.75 [113.05 against 114.0-114.5: mixed] (s15_ab*.log).
//
#if 0   // ---- 1. the carry loop (k_wavelocal) ---------------------------------------------------------------------------------------
#if MPT_WL_CARRY
        // The end of a pass (round 5): the wave is out of path ids, this step parked nothing and its rings are empty — what is left is
        // the hits in its lanes, a handful of long paths whose dependency chain (bounce after bounce, each a step of one nearly empty
        // wave) IS the time of the tail.  Through the ring every link of that chain carried two memory round trips (records written,
        // then read back by the same wave) plus a step's bookkeeping; here the hits are shaded and their bounce rays traced where they
        // are, until every path has ended.  Each ray sees the arithmetic it always saw.  (A 1/8 tile shard's pass and the frame
        // protocol's one-sample frames are mostly tail: DESIGN.md 6.)
        if (exhausted && pm == 0ull && am != 0ull) {
            uint32_t left = 0;
#pragma unroll
            for (int k = 0; k < (int)MPT_WL_LEVELS; ++k) left += cnt[k];
            if (left == 0u) {
                for (;;) {
                    if (alive) {   // as a ring-0 step shades the hits it pops
                        uint32_t px, py, sidx;
                        path_to_pixel(pp, ps.path, px, py, sidx);
                        g.pixel = py * pp.width + px;
                        g.sample = pp.sample_begin + sidx;
                        g.lit_seed = 0;
                        if (pp.sp.rng_mode == 0) g.lit_seed = pcg_hash(pcg_hash(pp.pixel_seed[g.pixel]));
                        if (!shade_bounce(pp.scene, lds_nodes, pp.sp, g, ps, best_t, best_prim)) {
                            store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
                            alive = false;
                        }
                    }
                    if (__ballot(alive) == 0ull) break;
                    if (alive) {   // ... and traces their bounce rays, here to the end
                        node = 0u;
                        best_t = INFINITY;
                        best_prim = -1;
                        closest_hit_resume<COUNT, ALL_LDS, false>(pp.scene, lds_nodes, ps.o, ps.d, node, best_t, best_prim, 0xFFFFFFFFu, wc);
                        n_rays++;
                        if (best_prim < 0) {
                            shade_bounce(pp.scene, lds_nodes, pp.sp, g, ps, best_t, -1);
                            store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
                            alive = false;
                        }
                    }
                }
                continue;   // (nothing to push; the next turn of the step loop finds the rings empty and leaves)
            }
        }
#endif
#endif

// ---- device build (mpt_sah.h), round 5 ------------------------------------------------------------------------------------------------
// REJECTED: two phases in run_sah — while a level has big tasks (>= 2048 items), the mid and small tasks it splits off WAIT in queues of
// their own (a task carried the item buffer its items are in, SahTask::side bit 1), and one level takes all of them up together once the
// big tasks are through.  Idea: sub-trees need nothing of one another, and the ~2000-item tasks that the last big levels shed keep five
// levels in a row at 55-60 us each with a tenth of the chip at work; taken up together they should share those levels.  Result (same
// box, gpurun_out/r05/s35-s36): the arrays stay the same (all 15 digests), the big levels get shorter (85 -> 71 us each, no mid / small
// grids beside them) — and the build gets 5 % SLOWER, 2.91 -> 3.08 ms, 3.00 -> 3.17 ms: the levels after the take-up are as long as before
// (62 / 55 / 63 / 70 / 74 us — a level lasts as long as its LONGEST task, and unbalanced splits keep tasks of ~2000 items alive for five
// levels whenever they start), and there are two more of them, because the tasks that used to be worked off beside the big levels now
// start later.  What those levels need is more lanes per long task (half a workgroup instead of a wave), not fewer levels.
//
// REJECTED: the mid and small tasks of a level in ONE kernel for every level (k_sah_tasks with both roles, instead of k_sah_level +
// k_sah_small): the last ten levels of 1 M items 885 -> 1060 us (the two kinds of blocks share a kernel's register and LDS budget, and the
// late levels are throughput-bound, not launch-bound); kept only for the levels that have big tasks, where the other grids are nearly empty.
//
// REJECTED: MPT_SAH_WAVES = 4 / 8 tasks per workgroup instead of 16: +13 % / +1 % on the build (more pushes per task).
//
// REJECTED: four trips of loads in flight in the partition loop of the mid tasks too (as in their bounds pass): 2.87 -> 3.00 ms for 1 M
// primitives (three of four same-box runs, gpurun_out/r05/s40): the registers it takes cost the throughput-bound late levels more than the
// five latency-bound ones gain.  One trip ahead (kept) was worth 62 -> 55 us on those levels.
