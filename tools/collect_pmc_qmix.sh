#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel stats + PMC passes (matrix-pipe busy cycles, FETCH_SIZE, WRITE_SIZE; separate passes,
# kernel-trace only) of the two QMIX-3m loops of bench.py's `secondary`.  Summaries land in gpurun_out/pmc_qmix/.
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_qmix
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for NAME in ff gru; do
  CMD="tools/profile_qmix_loops.py $NAME"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$NAME -o x -- python $ROOT/$CMD > /tmp/st_$NAME.log 2>&1
  cp /tmp/st_$NAME/x_kernel_stats.csv $OUT/${TAG}_qmix_${NAME}_kernel_stats.csv
  for PASS in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    P=${PASS%%:*}; CTR=${PASS#*:}
    rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d /tmp/pmc_${NAME}_$P -o x -- python $ROOT/$CMD > /tmp/pmc_${NAME}_$P.log 2>&1
  done
  python $ROOT/tools/summarize_pmc_generic.py /tmp/pmc_${NAME} "$CMD" > $OUT/${TAG}_qmix_${NAME}_pmc.json
done
ls -la $OUT
