"""CPU-side checks of the BUILT device code object (libmpt_hip.so is cross-compiled for gfx950 here; no GPU needed).

The bottom-up refit kernels of the device builder (k_refit, mpt_lbvh.h; k_own_tree, mpt_devbuild.h) hand boxes from one
thread to a thread of ANOTHER workgroup through sc1 stores followed by a relaxed agent-scope atomicAdd on an arrival
counter.  What orders the stores in front of the atomic is an explicit `s_waitcnt vmcnt(0)` (handoff_release(): on gfx9
vmcnt covers stores) — not the workgroup-scope fence next to it, which compiles to `s_waitcnt lgkmcnt(0)` only.  This
test reads the disassembly of the built library and fails if a compiler or source change ever loses that wait."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIB = os.path.join(ROOT, "metalpathtracer_amd", "lib", "libmpt_hip.so")


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not present")
    d = tmp_path_factory.mktemp("codeobj")
    shutil.copy(LIB, d / "lib.so")                      # (--offloading writes the bundles next to its input)
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
    co = [f for f in os.listdir(d) if "gfx950" in f]
    assert len(co) == 1, os.listdir(d)
    out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", co[0]], cwd=d, check=True, capture_output=True, text=True).stdout
    funcs = {}
    cur = None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur and line.startswith("\t"):
            funcs[cur].append(line.split("//")[0].strip())
    return funcs


@pytest.mark.parametrize("kernel", ["k_refit", "k_own_tree"])
def test_handover_stores_are_drained_before_the_arrival_atomic(disassembly, kernel):
    names = [n for n in disassembly if kernel in n and n.startswith("_Z")]
    assert len(names) == 1, names
    ins = disassembly[names[0]]
    atomics = [i for i, s in enumerate(ins) if s.startswith("global_atomic_add")]
    assert atomics, "no arrival counter in %s" % kernel
    for a in atomics:
        # walk back to the previous store of the hand-over (or the loop head): a vmcnt(0) wait must come in between
        waited = False
        for j in range(a - 1, -1, -1):
            s = ins[j]
            if s.startswith("s_waitcnt") and "vmcnt(0)" in s:
                waited = True
                break
            if s.startswith(("global_store", "global_atomic", "flat_store", "buffer_store")):
                break
        assert waited, "%s: no s_waitcnt vmcnt(0) between the hand-over stores and the atomic at instruction %d:\n%s" % (
            kernel, a, "\n".join(ins[max(0, a - 12):a + 1]))
        # and the stores of the hand-over are agent-scope (sc1) ones
    stores = [s for s in ins if s.startswith("global_store") or s.startswith("global_atomic_swap")]
    assert any("sc1" in s for s in stores), stores[:8]


def test_trace_kernels_hold_their_register_budget(disassembly):
    """The operating points DESIGN.md states: k_wavelocal without scratch at 6 waves/SIMD (<= 80 VGPRs), k_ordered<.., 5> at 5 (<= 96).
    Read from the kernel descriptors' symbol table is not possible here; the instruction stream must not touch scratch."""
    for key in ("k_wavelocalILb0ELb1E", "k_orderedILb0ELb0ELi5E", "k_orderedILb0ELb1ELi5E", "k_wavelocalILb0ELb0E"):
        names = [n for n in disassembly if key in n and n.startswith("_Z")]
        assert len(names) == 1, (key, names)
        scratch = [s for s in disassembly[names[0]] if s.startswith(("scratch_", "buffer_store_dword v", "buffer_load_dword v")) and "off" in s and "s[0:3]" in s]
        assert not [s for s in disassembly[names[0]] if s.startswith("scratch_")], (key, scratch[:4])


def test_six_wave_closest_first_kernel_spills_around_its_walk_loops_not_inside(disassembly):
    """k_ordered<.., 6> (2 x 768 threads per CU, 80 VGPRs: mpt_ordered.h) may spill a step's bookkeeping to scratch — a few dwords per
    STEP — but the node loop of the walk, the hot code of the kernel, must stay free of scratch traffic: no scratch instruction within
    the 30 instructions in front of a node fetch (seven 16-byte loads off the tree's base pointer in scalar registers) or the 260 behind it."""
    for key in ("k_orderedILb0ELb0ELi6E",):
        names = [n for n in disassembly if key in n and n.startswith("_Z")]
        assert len(names) == 1, (key, names)
        ins = disassembly[names[0]]
        loads = [i for i, s in enumerate(ins) if s.startswith("global_load_dwordx4") and re.search(r"s\[\d+:\d+\]", s)]
        fetches = [i for k, i in enumerate(loads) if k + 6 < len(loads) and loads[k + 6] - i <= 24]     # first load of a 7-load cluster
        fetches = [i for k, i in enumerate(fetches) if k == 0 or i - fetches[k - 1] > 24]
        assert len(fetches) >= 2, loads                      # (the budgeted and the unbudgeted instantiation of the walk)
        scratch = [i for i, s in enumerate(ins) if s.startswith("scratch_")]
        assert len(scratch) <= 80, len(scratch)
        for f in fetches:
            near = [i for i in scratch if f - 30 <= i <= f + 260]
            assert not near, (key, f, [ins[i] for i in near][:4])
