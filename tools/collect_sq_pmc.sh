#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ issue / wait counters of the headline's kernels (one PMC pass per group of <= 8 SQ counters; PMC runs
# carry --kernel-trace only).  Writes gpurun_out/<tag>_sq_pmc.json (per kernel: mean counter values per launch).
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -o ppo -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p2 -o ppo -- $CMD > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, json, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
doc = {k: {c: round(v[0] / max(v[1], 1), 1) for c, v in cs.items()} | {"launches": max(v[1] for v in cs.values())} for k, cs in acc.items() if "xrl::" in k}
json.dump(doc, open("$ROOT/gpurun_out/${TAG}_sq_pmc.json", "w"), indent=1)
for k in doc:
    if "trunk" in k or "reduce_adam" in k: print(k, json.dumps(doc[k]))
PY
tail -3 $OUT/p1.log; tail -3 $OUT/p2.log
rm -rf $OUT/p1 $OUT/p2
