#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3: one kernel-trace/stats pass and separate PMC passes
# (never combined with tracing domains other than --kernel-trace).  Output: gpurun_out/prof_<tag>_<workload>/,
# condensed into profiles/<tag>_<workload>_* by tools/summarize_profile.py.
# Usage: tools/profile.sh <tag> [workload (c3|c4|c5)] [extra bench args...]
set -e
TAG=${1:-r02}; shift || true
WL=${1:-c3}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
export TMPDIR=/tmp
# 20 timed frames + 2 warm-up + bench.py's 20 isolated frames for the roofline = 42 frames per pass, all on one stream
# (--in-flight 1 --sub-frames 1: overlapping launches would stretch each other's durations)
STEPS=${RT_PROFILE_STEPS:-20}
ARGS="--steps $STEPS --warmup 2 --in-flight 1 --sub-frames 1 --no-cpu-baseline --no-boundary-costs --no-other-workloads --workload $WL $@"
echo $((STEPS + 2 + (STEPS < 20 ? STEPS : 20))) > $OUT/n_frames.txt
python3 -c "import sys; sys.path.insert(0, '$REPO'); from hslu_i.ba_raytracing.f2501_raytracer_amd import _lib; print(_lib.load().rt_build_id().decode())" > $OUT/build_id.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 || true
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq3 -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq3.log 2>&1 || true
# transcendental share of the VALU instructions (for the issue-cost weighting of the VALU bound), if the counter exists
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq4 -- python3 $REPO/bench.py $ARGS > $OUT/pmc_sq4.log 2>&1 || true
cd $REPO
python3 tools/summarize_profile.py $TAG $WL
