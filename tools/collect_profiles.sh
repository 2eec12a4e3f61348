#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + the two HBM PMC passes of the bench command, summarised under
# gpurun_out/profiles_<tag>/ (copy what should be judged into profiles/).  PMC passes are separate runs and never
# combined with sys/hip/hsa tracing (MI355X_MICROARCH.md, gpurun policy).
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o ppo -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o ppo -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o ppo -- $CMD > $OUT/pmc_write.log 2>&1
python $ROOT/tools/summarize_pmc.py $OUT $TAG
cp $OUT/stats/ppo_kernel_stats.csv $OUT/${TAG}_ppo_c2_kernel_stats.csv
grep "^{" $OUT/bench_under_rocprof.log | tail -1 > $OUT/${TAG}_ppo_c2_bench_under_rocprof.json
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
