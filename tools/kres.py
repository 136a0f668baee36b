"""Print VGPR/SGPR/scratch/occupancy per kernel for a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-I/root/repo/include",
       "-I/root/repo/metalpathtracer_amd/csrc", "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    if not any(x in k for x in ("k_step", "k_mega", "k_regen", "k_wavelocal", "k_stream", "k_trace", "k_ordered")): continue
    print("%-48s VGPR %3s SGPR %3s scratch %4s occ %s spillS %s spillV %s" % (k[:48], v.get("VGPRs"), v.get("TotalSGPRs"), v.get("ScratchSize [bytes/lane]"), v.get("Occupancy [waves/SIMD]"), v.get("SGPRs Spill"), v.get("VGPRs Spill")))
