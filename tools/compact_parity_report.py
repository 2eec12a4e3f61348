"""profiles/r03_parity_errors_gpu.json from the XRL_PARITY_REPORT jsonl of a strict `-m gpu` run (tests/conftest.py writes one line
per comparison): per test the number of comparisons and the worst err / tol, plus every record above a quarter of its tolerance and
all float64-anchored / chain / noise records.  usage: compact_parity_report.py <in.jsonl> <out.json> <n_tests>"""
import json, sys
src, dst, n_tests = sys.argv[1], sys.argv[2], int(sys.argv[3])
recs = [json.loads(l) for l in open(src) if l.strip()]
per, keep = {}, []
for r in recs:
    t = per.setdefault(r["test"], {"n": 0, "worst": 0.0, "worst_what": ""})
    t["n"] += 1
    ratio = r["err"] / r["tol"] if r.get("tol") else 0.0
    if ratio > t["worst"]:
        t["worst"], t["worst_what"] = round(ratio, 4), r["what"]
    tagged = any(k in r for k in ("err64", "ref64", "bound")) or any(w in r["what"] for w in ("chain after", "noise", "float64", "EXCEPTION")) \
        or r.get("decided_by") == "f64"
    if ratio > 0.25 or tagged:
        keep.append({k: v for k, v in r.items() if k != "err_legacy"})
# the explicit exception list: every gradient that is NOT within its tolerance of the reference's float32 value and passed on the float64 rule
exc = [{"test": r["test"], "what": r["what"].split(" [")[0], "vs_f32_ref": r["d32"], "vs_f64_twin": r["d64"], "f32_ref_vs_f64_twin": r["ref32_vs_ref64"],
        "margin": round(r["d64"] / max(r["ref32_vs_ref64"], 1e-30), 3)} for r in recs if r.get("decided_by") == "f64"]
exc.sort(key=lambda e: -e["vs_f32_ref"])
n_f32 = sum(1 for r in recs if r.get("decided_by") == "f32")
out = {"gradient_rule": {"within_tol_of_the_float32_reference": n_f32, "exceptions_passed_on_the_float64_rule": len(exc),
                         "note": "exceptions: further than tol from the reference's float32 gradient, but no further from the reference's float64 twin than "
                                 "the float32 reference itself is (margin = that ratio, <= 1)", "exceptions": exc},
       "what": "Strict -m gpu run (%d tests; profiles/rNN_* = round NN): every numeric comparison goes through tests/conftest.py (assert_close: error "
               "relative to the tensor's OWN scale; assert_grad_close: float64-anchored; LearnerFixtureCheck: parameter steps through "
               "Adam's conditioning; 'chain after..': the engine against the reference's own 64-update float32 / float64 chains).  "
               "per_test: number of comparisons and the worst err / tol; records: every comparison above a quarter of its tolerance "
               "plus all float64-anchored / chain / noise records" % n_tests,
       "n_comparisons": len(recs), "per_test": per, "records": keep}
json.dump(out, open(dst, "w"))
print(len(recs), "comparisons,", len(keep), "records kept, worst ratio", max(v["worst"] for v in per.values()))
print("gradients within tol of the float32 reference:", n_f32, "; exceptions on the float64 rule:", len(exc))
for e in exc[:40]:
    print("  %-70s vs f32 %.2e  vs f64 %.2e  (reference itself %.2e)  %s" % (e["what"][:70], e["vs_f32_ref"], e["vs_f64_twin"], e["f32_ref_vs_f64_twin"], e["test"].split("::")[-1]))
