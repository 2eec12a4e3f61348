#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of `python <script> [args]`, the last N kernels as a timeline (name, duration, gap)
# under gpurun_out/timeline_<tag>/.  usage: profile_timeline.sh <tag> <n_last> <script> [args...]
TAG=$1; NLAST=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/timeline_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o t -- python $R/"$@" > $OUT/run.log 2>&1
cp $OUT/stats/t_kernel_stats.csv $OUT/kernel_stats.csv
python - "$OUT" "$NLAST" <<'P'
import csv, sys
out, nlast = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(out + "/stats/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev, lines = None, []
for r in rows[-nlast:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    lines.append("%-64s %7.2f us gap %6.2f" % (r["Kernel_Name"][:64], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
open(out + "/timeline.txt", "w").write("\n".join(lines))
P
rm -rf $OUT/stats
tail -2 $OUT/run.log
