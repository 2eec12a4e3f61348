#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + three PMC passes (matrix-pipe busy cycles, HBM fetch, HBM write) of the
# C4-shape PPO loop (tools/bench_c4.py), summarised under gpurun_out/c4p/ (copy what should be judged into profiles/).
# PMC passes are separate runs and never combined with sys/hip/hsa tracing (MI355X_MICROARCH.md, gpurun policy).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c4p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c4 -- python $R/tools/bench_c4.py > $O/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p_mfma -o c4 -- python $R/tools/bench_c4.py > $O/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -o c4 -- python $R/tools/bench_c4.py > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_write -o c4 -- python $R/tools/bench_c4.py > $O/pmc_write.log 2>&1
python $R/tools/summarize_pmc_generic.py $O/p "python tools/bench_c4.py" > $O/${TAG}_c4_pmc.json
cp $O/stats/c4_kernel_stats.csv $O/${TAG}_c4_kernel_stats.csv
grep "^{" $O/bench_under_rocprof.log | tail -1 > $O/${TAG}_c4_bench_under_rocprof.json
rm -rf $O/stats $O/p_mfma $O/p_fetch $O/p_write
ls -la $O; head -40 $O/${TAG}_c4_pmc.json
