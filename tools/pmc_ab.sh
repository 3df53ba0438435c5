#!/bin/bash
# PMC A/B of one workload under two rt_tuning settings: tools/pmc_ab.sh <workload> "<tuning A>" "<tuning B>"
# (instruction counts and cycles of the render kernels, 4 launches each; separate rocprofv3 --pmc pass per setting)
WL=${1:-c3}; A=$2; B=$3
REPO=$(pwd); export TMPDIR=/tmp
cd /tmp
for T in "$A" "$B"; do
  tag=$(echo "x$T" | tr -c 'a-zA-Z0-9' '_')
  rm -rf $REPO/gpurun_out/pmcab_$tag
  RT_AB_TUNING="$T" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $REPO/gpurun_out/pmcab_$tag -- python3 $REPO/tools/perf_ab.py $WL > $REPO/gpurun_out/pmcab_$tag.log 2>&1
  echo "== tuning '$T'"; grep kernel $REPO/gpurun_out/pmcab_$tag.log; python3 $REPO/tools/pmc_by_kernel.py $REPO/gpurun_out/pmcab_$tag
done
