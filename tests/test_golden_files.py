"""The committed fixtures that GPU tests and tools compare against are well formed (checked here, without a GPU)."""
import json
import os
import re

from conftest import ROOT


def test_devbuild_digests_hold_sixteen_words_for_five_scenes_and_three_builders():
    """tests/golden/devbuild_digests.json — written by tools/gpu_scene_digest.py --write from the build BEFORE round 5's rewrite of the device
    builder: per scene and builder the digests of the nine device arrays (mpt_scene_digest, include/mpt.h) and seven counts."""
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "devbuild_digests.json")))
    scenes = ("scene.xml", "cornell.xml", "glass.xml", "bunny20.xml", "config4")
    assert sorted(d) == sorted("%s/%s" % (s, b) for s in scenes for b in ("sah", "ploc", "lbvh"))
    for key, words in d.items():
        assert len(words) == 16 and all(re.fullmatch(r"[0-9a-f]{16}", w) for w in words), key
        counts = [int(w, 16) for w in words[9:]]
        n_nodes, n_prims, n_mats, n_own, n_leaves, n_always, depth = counts
        assert 1 <= n_prims <= 1000003 and 1 <= n_nodes <= 2 * n_prims - 1 and 1 <= n_mats <= n_prims, key
        assert 1 <= n_leaves <= n_prims and 1 <= n_own <= n_prims + 2 and n_always <= 16 and 1 <= depth <= 64, key
        # the same scene has the same primitives and materials whatever builds its tree
        sah = d[key.split("/")[0] + "/sah"]
        assert words[10] == sah[10] and words[11] == sah[11] and words[2] == sah[2], key
    assert int(d["config4/sah"][10], 16) == 1000003 and int(d["bunny20.xml/sah"][10], 16) == 99362
