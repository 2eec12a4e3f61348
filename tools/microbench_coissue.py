"""Scratch (round 6): can vector work hide under the fp32 matrix instructions of the OTHER wave of its SIMD?  (The premise of a
ping-pong between two row blocks inside the minibatch kernel.)  tools/csrc/probe.hip: coissue_probe_kernel -- waves 0..3 run role A,
waves 4..7 (the SIMD partners) role B, each timed with the shader clock."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from tools import probe_lib
ROLES = {0: "idle", 1: "f32 MFMA 32x32x2, one accumulator", 2: "f32 MFMA 32x32x2, four accumulators", 3: "VALU fma chains",
         4: "LDS ds_read_b128 + adds", 5: "bf16 MFMA 32x32x16, one accumulator", 6: "VALU v_exp chains"}
out = torch.zeros(16, dtype=torch.int64, device="cuda")
sink = torch.zeros(1024, device="cuda")


def run(a, b, ia, ib, prio=0):
    best = None
    for _ in range(5):
        out.zero_()
        probe_lib.call("xrl_probe_coissue", a, b, ia, ib, prio, 256, C.c_void_p(out.data_ptr()), C.c_void_p(sink.data_ptr()),
                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        d = out.cpu().numpy()
        t = (int(d[:4].max()), int(d[4:8].max()))
        best = t if best is None or sum(t) < sum(best) else best
    return best


res = []
IT = {1: 64, 2: 64, 3: 128, 4: 256, 5: 64, 6: 128}      # iterations: each role alone runs for roughly 65 k cycles
print("alone (cycles):")
alone = {}
for r in (1, 2, 3, 4, 5, 6):
    alone[r] = run(r, 0, IT[r], 0)[0]
    print("  %-40s %7d cycles" % (ROLES[r], alone[r]))
for a in (1, 2, 5):
    for b in (3, 6, 4, 1, 2):
        for prio in (0, 1):
            ta, tb = run(a, b, IT[a], IT[b], prio)
            line = {"a": ROLES[a], "b": ROLES[b], "prio_b": 3 * prio, "a_alone": alone[a], "b_alone": alone[b], "a_with_b": ta, "b_with_a": tb,
                    "sum_alone": alone[a] + alone[b], "max_together": max(ta, tb)}
            res.append(line)
            print("A = %-38s B = %-28s prio_b %d:  A %7d (alone %7d)  B %7d (alone %7d)   both done after %7d  (serial: %7d)"
                  % (ROLES[a], ROLES[b], 3 * prio, ta, alone[a], tb, alone[b], max(ta, tb), alone[a] + alone[b]))
# the SAME wave: fmas between its own MFMAs (64 iterations x 16 MFMAs, NV fmas behind each)
same = {}
for role, nv in ((7, 4), (8, 8), (9, 12), (10, 16)):
    t1 = run(role, 0, 64, 0)[0]
    t2 = run(role, role, 64, 64)
    same[str(nv)] = {"one_wave_per_simd": t1, "two_waves_per_simd": max(t2)}
    print("same wave: 1 f32 MFMA + %2d fma, x 1024: %7d cycles with one wave per SIMD (MFMAs alone: %d; the fmas alone: ~%d), %7d with two such waves per SIMD"
          % (nv, t1, alone[2], 1024 * nv * 4, max(t2)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"alone": {ROLES[k]: v for k, v in alone.items()}, "pairs": res, "same_wave_fma_per_mfma": same}, open(os.path.join(ROOT, "gpurun_out", "r06_c_coissue.json"), "w"), indent=1)
