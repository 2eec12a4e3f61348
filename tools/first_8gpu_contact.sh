#!/usr/bin/env bash
# First contact with a multi-GPU node: one short bench.py run per gradient-averaging path, each under its own time limit, so that a
# single session yields a usable record even if one path misbehaves on real xGMI (fine-grained peer memory of the in-launch exchange,
# RCCL under graph capture).  Per path and GPU count: pass / fail, the exit code, ms per step, env-steps/s -> JSON lines.
#
#   tools/first_8gpu_contact.sh [out.jsonl] [gpu counts, default "2 4 8"]
#
# paths: exchange = gradients averaged INSIDE the optimiser launch through IPC-mapped peer buffers (xrl_reduce_adam_exchange);
#        captured = one flat RCCL all-reduce per optimiser step, captured into the update-phase graph;
#        cut      = the update phase cut into graphs AT the collectives (RCCL calls between graph launches);
#        measure  = bench.py's default: time every usable path on the real workload and adopt the fastest.
set -u
cd "$(dirname "$0")/.."
OUT="${1:-gpurun_out/first_multi_gpu_contact.jsonl}"
COUNTS="${2:-2 4 8}"
mkdir -p "$(dirname "$OUT")"
export HSA_ENABLE_IPC_MODE_LEGACY="${HSA_ENABLE_IPC_MODE_LEGACY:-0}"
for n in $COUNTS; do
  for path in exchange captured cut measure; do
    log="$(mktemp)"
    t0=$(date +%s.%N)
    timeout 600 python bench.py --gpus "$n" --steps 5 --warmup 2 --grad-path "$path" --no-secondary --no-cpu-baseline --no-roofline >"$log" 2>"$log.err"
    rc=$?
    t1=$(date +%s.%N)
    python - "$n" "$path" "$rc" "$log" "$t0" "$t1" >>"$OUT" <<'PY'
import json, sys
n, path, rc, log, t0, t1 = sys.argv[1:7]
line = None
for l in open(log):
    l = l.strip()
    if l.startswith("{"):
        try:
            line = json.loads(l)
        except ValueError:
            pass
rec = {"n_gpus": int(n), "gradient_path": path, "rc": int(rc), "ok": int(rc) == 0 and line is not None, "wall_s": round(float(t1) - float(t0), 1)}
if line:
    rec.update(value=line["value"], ms_per_step=line["ms_per_step"], rccl_world=line["config"].get("rccl_world"),
               gradient_average=line["config"].get("gradient_average"), gradient_paths_ms=line["config"].get("gradient_paths_ms"))
else:
    rec["stderr_tail"] = open(log + ".err").read()[-600:]
print(json.dumps(rec))
PY
    rm -f "$log" "$log.err"
  done
done
cat "$OUT"
