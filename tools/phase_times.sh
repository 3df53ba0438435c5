#!/bin/bash
# Per-kernel average durations of one workload under one build / pipeline form (rocprofv3 --kernel-trace --stats of a short
# bench.py run, frames one after the other on one stream).  Usage: tools/phase_times.sh <tag> <workload> <phases 1|2> [lib.so]
TAG=$1; WL=$2; PH=$3; LIB=$4
R=$(pwd)
export TMPDIR=/tmp
[ -n "$LIB" ] && export RT_HIP_LIB=$R/hslu_i/ba_raytracing/f2501_raytracer_amd/$LIB
OUT=$R/gpurun_out/${TAG}_${WL}_p${PH}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 10 --warmup 2 --in-flight ${INFLIGHT:-1} --sub-frames ${SUBF:-1} --no-cpu-baseline --no-boundary-costs --no-other-workloads --workload $WL --phases $PH --levels ${LEVELS:-0} > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "== $TAG $WL phases=$PH lib=${LIB:-librt_hip.so}: $(python3 -c "import json,sys; d=json.loads(open('$OUT.log').read().strip().splitlines()[-1]); print('%.3f ms/frame' % d['ms_per_step'])" 2>/dev/null)"
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rt_" in r["Name"]]
for r in rows:
    name = r["Name"].split("::")[-1].split("(")[0]
    print(f"   {name:28s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e6:8.3f} ms  total/frame {float(r['TotalDurationNs'])/1e6/22:8.3f} ms")
PY
