cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/c3prof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3prof/stats -o c3 -- python $R/tools/profile_dqn_c3.py > $R/gpurun_out/c3prof/run.log 2>&1
cp $R/gpurun_out/c3prof/stats/c3_kernel_stats.csv $R/gpurun_out/c3prof/kernel_stats.csv
python - <<'P'
import csv,os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
rows=list(csv.DictReader(open(R+"/gpurun_out/c3prof/stats/c3_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 2 graph launches: print the sequence of the last ~80 kernels with durations and gaps
tail=rows[-70:]
out=[]
prev=None
for r in tail:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    out.append("%-60s %7.2f us gap %6.2f grid %s"%(r["Kernel_Name"][:60],(e-s)/1e3,(s-prev)/1e3 if prev else 0,r.get("Grid_Size","")))
    prev=e
open(R+"/gpurun_out/c3prof/timeline.txt","w").write("\n".join(out))
P
rm -rf $R/gpurun_out/c3prof/stats
grep update_us $R/gpurun_out/c3prof/run.log
