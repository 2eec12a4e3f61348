#!/bin/bash
# Run ON THE GPU BOX (via gpurun): PMC passes for the off-policy update graphs (BASELINE.json north_star: "achieved HBM GB/s
# on the GAE/TD kernels and MFMA utilisation on the mixer GEMM").  Separate passes, kernel-trace only (no sys/hip/hsa
# tracing together with --pmc).  Summaries land in gpurun_out/pmc_off/ (copy what should be judged into profiles/).
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_off
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for WHAT in "qmix_rnn:tools/profile_qmix_rnn.py bptt" "dqn_c3:tools/profile_dqn_c3.py" "scale:tools/scale_sweep.py"; do
  NAME=${WHAT%%:*}; CMD=${WHAT#*:}
  for PASS in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    P=${PASS%%:*}; CTR=${PASS#*:}
    rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d /tmp/pmc_${NAME}_$P -o x -- python $ROOT/$CMD > /tmp/pmc_${NAME}_$P.log 2>&1
  done
  python $ROOT/tools/summarize_pmc_generic.py /tmp/pmc_${NAME} "$CMD" > $OUT/${TAG}_${NAME}_pmc.json
done
ls -la $OUT
