set -e
mkdir -p gpurun_out/r05
tests/experiments/_build/hipcub_partial_bits > gpurun_out/r05/s3_hipcub.log 2>&1 || true
tail -3 gpurun_out/r05/s3_hipcub.log
SCENES=bunny20.xml REPS=4 tools/gpu_ab.sh base sel2 t768w6 base > gpurun_out/r05/s3_ab_bunny.log 2>&1
cat gpurun_out/r05/s3_ab_bunny.log
MPT_LIB=$PWD/metalpathtracer_amd/lib/libmpt_hip_times.so BVH=3 python3 tools/gpu_ot_times.py bunny20.xml 64 > gpurun_out/r05/s3_ot_bunny20.txt 2>&1
cat gpurun_out/r05/s3_ot_bunny20.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05/s3_tests.log 2>&1
tail -5 gpurun_out/r05/s3_tests.log
