#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + PMC passes (matrix-pipe busy cycles, HBM fetch / write; separate passes,
# kernel-trace only) of the DQN-C3 update graph (tools/profile_dqn_c3.py) and of the recurrent QMIX update (tools/profile_qmix_rnn.py)
# -> gpurun_out/pmc_c3/<tag>_* (copy what should be judged into profiles/).
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_c3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for WHAT in "dqn_c3:tools/profile_dqn_c3.py" "qmix_gru:tools/profile_qmix_rnn.py"; do
  NAME=${WHAT%%:*}; CMD=${WHAT#*:}
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_${NAME} -o x -- python $ROOT/$CMD > $OUT/${TAG}_${NAME}_run.log 2>&1
  cp /tmp/st_${NAME}/x_kernel_stats.csv $OUT/${TAG}_${NAME}_kernel_stats.csv
  for PASS in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    P=${PASS%%:*}; CTR=${PASS#*:}
    timeout 200 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d /tmp/pmc_${NAME}_$P -o x -- python $ROOT/$CMD > /tmp/pmc_${NAME}_$P.log 2>&1
  done
  python $ROOT/tools/summarize_pmc_generic.py /tmp/pmc_${NAME} "$CMD" > $OUT/${TAG}_${NAME}_pmc.json
done
grep update_us $OUT/*_run.log
ls -la $OUT
