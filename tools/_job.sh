set -e
mkdir -p gpurun_out/r05
D=$(mktemp -d /tmp/mpt_occ_XXXX); cp assets/bunny.obj $D/
{ echo "<Scene>"; sed -n 3,4p assets/bunny20.xml; grep "<Mesh" assets/bunny20.xml | head -8; echo "</Scene>"; } > $D/b8.xml
python3 - $D <<'PY'
import sys
d = sys.argv[1]
with open(d + "/b80.xml", "w") as f:
    f.write('<Scene>\n<Sphere position="0,-10000,0" radius="10000" albedo="0.8,0.8,0.8" emission="0,0,0" materialType="0" emissionPower="0" />\n')
    f.write('<Sphere position="0,60,-20" radius="10" albedo="0.0,0.0,0.0" emission="1.0,0.9,0.7" materialType="0" emissionPower="5" />\n')
    for z in range(8):
        for x in range(10):
            f.write('<Mesh file="bunny.obj" position="%d,0,%d" scale="5.0" albedo="0.9,0.5,0.3" emission="0,0,0" materialType="0" emissionPower="0" />\n' % (-36 + 8 * x, 8 - 8 * z))
    f.write("</Scene>\n")
PY
V='base t768w6 base:MPT_TILE_ORDER=3 t768w6:MPT_TILE_ORDER=3 base'
SCENES="$D/b8.xml $D/b80.xml" REPS=4 tools/gpu_ab.sh $V > gpurun_out/r05/s7_ab_occ.log 2>&1
cat gpurun_out/r05/s7_ab_occ.log
