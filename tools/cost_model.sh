set -e
mkdir -p gpurun_out
{
for k in 0 1 2 3 5; do echo "lights=$k"; RT_AB_LIGHTS=$k python tools/perf_ab.py c3 --reps 3; done
echo N1; python tools/perf_ab.py sem=high_resolution+anti_aliasing//text --reps 3
echo N19; python tools/perf_ab.py sem=high_resolution+high_quality//text --reps 3
echo N28; python tools/perf_ab.py sem=high_resolution+extreme_quality//text --reps 3
} > gpurun_out/model.log 2>&1
