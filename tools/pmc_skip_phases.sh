#!/bin/bash
# VALU instruction counts of the phase kernels under removal builds
R=$(pwd); export TMPDIR=/tmp
cd /tmp
for k in 0 1 2 3 32 35; do
  lib=librt_hip.so; [ $k != 0 ] && lib=librt_hip_skip$k.so
  export RT_HIP_LIB=$R/hslu_i/ba_raytracing/f2501_raytracer_amd/$lib
  O=$R/gpurun_out/r04j_skip$k
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 $R/bench.py --steps 3 --warmup 1 --in-flight 1 --sub-frames 1 --no-cpu-baseline --no-boundary-costs --no-other-workloads --workload c3 --phases 2 > $O.log 2>&1
  echo "== SKIP=$k"; python3 $R/tools/pmc_by_kernel.py $O 24 | grep -A1 -E "^rt_sets0_list|^rt_classify0" | grep -v "^--" | cut -c1-330
done
