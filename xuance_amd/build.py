"""Build recipe for libxrl_hip.so (gfx950 only): explicit hipcc, in-tree output so the .so travels with the repo.
Every csrc/*.hip is compiled to its own object (in parallel, re-compiled only when it or a header changed), then linked."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libxrl_hip.so")
ARCH = "gfx950"
# -ffp-contract=off: the GAE scan and the running statistics must round every multiply / add like NumPy (bit-exact
# parity), and the shape-specialised kernels are tested bit-identical to their any-shape twins -- a contraction the
# compiler applies in one twin and not in the other would break that; the MFMA and packed-FMA paths are explicit.
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=off"]


def extra_flags():
    """XRL_BUILD_DEFINES="-DXRL_TILE_PROBE": diagnostic builds (phase stamps in the fused kernels; use with --force)."""
    return os.environ.get("XRL_BUILD_DEFINES", "").split()


def hipcc():
    h = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return h if os.path.exists(h) else "hipcc"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(os.path.dirname(HERE), "include", "xrl_hip.h")]


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB_PATH, sources() + headers())


def build(force=False, verbose=True):
    """Compile every HIP source with hipcc --offload-arch=gfx950 and link xuance_amd/lib/libxrl_hip.so."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = headers()
    todo = [s for s in sources() if force or _stale(_obj(s), [s] + hdrs)]

    def compile_one(src):
        cmd = [hipcc(), *FLAGS, *extra_flags(), "-c", src, "-o", _obj(src)]
        if verbose:
            print("[xuance_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    keep = {_obj(s) for s in sources()}
    for f in os.listdir(OBJ_DIR):                               # objects of sources that no longer exist
        if os.path.join(OBJ_DIR, f) not in keep:
            os.remove(os.path.join(OBJ_DIR, f))
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *sorted(keep), "-o", LIB_PATH]
    if verbose:
        print("[xuance_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
