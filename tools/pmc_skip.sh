#!/bin/bash
REPO=$(pwd); export TMPDIR=/tmp; D=$REPO/hslu_i/ba_raytracing/f2501_raytracer_amd
cd /tmp
for v in ${PMC_SKIP_LIBS:-librt_hip librt_hip_skip2 librt_hip_skip4}; do
  rm -rf $REPO/gpurun_out/pmcs_$v
  RT_HIP_LIB=$D/$v.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/pmcs_$v -- python3 $REPO/tools/perf_ab.py c3 > $REPO/gpurun_out/pmcs_$v.log 2>&1
  echo "== $v"; grep kernel $REPO/gpurun_out/pmcs_$v.log | cut -c1-70; python3 $REPO/tools/pmc_by_kernel.py $REPO/gpurun_out/pmcs_$v
done
