set -e
mkdir -p gpurun_out/r05
V='base le2 le3 le4 le16 base'
SCENES="scene.xml" REPS=7 tools/gpu_ab.sh $V > gpurun_out/r05/s16_ab_le.log 2>&1
cat gpurun_out/r05/s16_ab_le.log
SCENES="scene.xml" BVH=0 REPS=5 tools/gpu_ab.sh $V > gpurun_out/r05/s16_ab_le_ref.log 2>&1
cat gpurun_out/r05/s16_ab_le_ref.log
SCENES="cornell.xml" CAM=cornell REPS=5 tools/gpu_ab.sh $V > gpurun_out/r05/s16_ab_le_cornell.log 2>&1
cat gpurun_out/r05/s16_ab_le_cornell.log
