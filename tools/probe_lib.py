"""Builds and loads tools/lib/libxrl_probe.so: the diagnostic kernels of tools/csrc/probe.hip (shader-clock / MFMA-issue /
I-cache / XCD-barrier probes).  Test and measurement infrastructure only -- not linked into libxrl_hip.so."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "probe.hip")
LIB = os.path.join(HERE, "lib", "libxrl_probe.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "xuance_amd", "csrc"), "-I", os.path.join(ROOT, "include"), SRC, "-o", LIB],
                   check=True)
    return LIB


_lib = None


def call(name, *args):
    """xrl_probe_<name>(*args); the probes return 0 on success."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (PyTorch-ROCm's HIP runtime first)
        _lib = C.CDLL(build())
    fn = getattr(_lib, name)
    fn.restype = C.c_int
    rc = fn(*[C.c_void_p(a) if isinstance(a, int) and a > 2 ** 31 else a for a in args])
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc})")


if __name__ == "__main__":
    print(build(force=True))
