"""Compile every csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and -S, and list what costs memory round trips where
none is expected: kernels / noinline device functions with scratch (spilled VGPRs, stack objects, callee-saved registers saved around nested
calls) and FLAT accesses (generic pointers: e.g. `cond ? *lds_ptr : local` compiles to a select between an LDS address and a stack copy).
Round 5 found two such cases on hot paths this way (actor_rollout_kernel, qf_lin_fwd_fn).  No GPU needed.

    python tools/scan_kernel_resources.py [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xuance_amd.build import FLAGS, hipcc  # noqa: E402

INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "xuance_amd", "csrc")]


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except FileNotFoundError:
        return name


def scan(src, tmp):
    base = os.path.basename(src)[:-4]
    r = subprocess.run([hipcc(), *FLAGS, *INC, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(tmp, base + ".o")],
                       capture_output=True, text=True)
    res = {}
    for blk in r.stderr.split("remark: Function Name: ")[1:]:
        name = blk.split(" ")[0]
        g = lambda k: int(m.group(1)) if (m := re.search(k + r": (\d+)", blk)) else -1       # noqa: E731
        res[name] = dict(vgpr=g("VGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"), vspill=g("VGPRs Spill"), sspill=g("SGPRs Spill"),
                         occ=g(r"Occupancy \[waves/SIMD\]"))
    asm = os.path.join(tmp, base + ".s")
    subprocess.run([hipcc(), *FLAGS, *INC, "-S", "--cuda-device-only", "-o", asm, src], capture_output=True, text=True)
    cur, ops = None, {}
    if os.path.exists(asm):
        for line in open(asm):
            m = re.match(r"^(_Z\S+):", line)
            if m:
                cur = m.group(1)
            if cur:
                o = ops.setdefault(cur, dict(flat=0, scratch_ops=0))
                o["flat"] += bool(re.search(r"\bflat_(load|store|atomic)", line))
                o["scratch_ops"] += "scratch_" in line
    return base, res, ops


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "xuance_amd", "csrc", "*.hip")))
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            base, res, ops = scan(f, tmp)
            for name in sorted(set(res) | set(ops)):
                r, o = res.get(name, {}), ops.get(name, {})
                if r.get("scratch", 0) > 0 or r.get("vspill", 0) > 0 or o.get("flat", 0) or o.get("scratch_ops", 0):
                    print(f"{base:14s} {demangle(name)[:88]:88s} scratch={r.get('scratch', '-')} vspill={r.get('vspill', '-')} "
                          f"sspill={r.get('sspill', '-')} vgpr={r.get('vgpr', '-')} flat={o.get('flat', 0)} scratch_ops={o.get('scratch_ops', 0)}")
