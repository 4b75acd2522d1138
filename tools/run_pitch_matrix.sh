#!/bin/bash
# Developer tool: wall time and event-kernel times of one evaluation per staged shape with the LDS row pitch of the policy
# (set_windows_impl) and with the other one forced (EINCM_PITCH_ALIGNED).  Output: gpurun_out/pitch/out.txt (profiles/r03/pitch_by_shape.txt).
set -e
mkdir -p gpurun_out/pitch
: > gpurun_out/pitch/out.txt
for shape in "260 346 1000000 8" "260 346 1000000 1" "260 346 3000000 2" "480 640 2500000 1" "480 640 5000000 1" "480 640 10000000 1"; do
  for h in 1 16; do
    EINCM_PITCH_ALIGNED=1 python tools/dev_c5_times.py $h 0 $shape >> gpurun_out/pitch/out.txt
    EINCM_PITCH_ALIGNED=0 python tools/dev_c5_times.py $h 0 $shape >> gpurun_out/pitch/out.txt
    python tools/dev_c5_times.py $h 0 $shape >> gpurun_out/pitch/out.txt
  done
done
cat gpurun_out/pitch/out.txt
