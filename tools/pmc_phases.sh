#!/bin/bash
# Per-kernel PMC figures of one workload under one pipeline form: tools/pmc_phases.sh <tag> <workload> <phases 1|2> [lib.so]
# (two rocprofv3 --pmc passes of a short bench.py run, counters only; 12 frames per pass)
TAG=$1; WL=$2; PH=$3; LIB=$4
R=$(pwd); export TMPDIR=/tmp
[ -n "$LIB" ] && export RT_HIP_LIB=$R/hslu_i/ba_raytracing/f2501_raytracer_amd/$LIB
ARGS="--steps 5 --warmup 2 --in-flight 1 --sub-frames 1 --no-cpu-baseline --no-boundary-costs --no-other-workloads --workload $WL --phases $PH"
cd /tmp
O=$R/gpurun_out/${TAG}_${WL}_p${PH}_pmc
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d ${O}1 -- python3 $R/bench.py $ARGS > ${O}1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${O}2 -- python3 $R/bench.py $ARGS > ${O}2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${O}3 -- python3 $R/bench.py $ARGS > ${O}3.log 2>&1
echo "== $TAG $WL phases=$PH"
for i in 1 2 3; do python3 $R/tools/pmc_by_kernel.py ${O}$i 12; done
