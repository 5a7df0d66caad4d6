"""Build libviterbi.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libviterbi.so")
SOURCES = ["vit_api.hip", "vit_wave.hip", "vit_pk.hip", "vit_sort.hip", "rs_kernels.hip", "vit_multi.hip", "vit_lat.hip"]
PK8_SOURCE = "vit_pk8.hip"  # round-3 experiment (8 frames per wavefront, slower): only with extra=["-DVIT_WITH_PK8"]
DEPS = SOURCES + ["vit_internal.h", "vit_pk_dev.h", "exports.map"]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.join(ROOT, "include", "viterbi_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=(), out=None):
    """extra/out: kernel experiments (tools/exp/ab.sh) build flag variants next to the product library"""
    if out is None and not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
           "-o", out or SO] + list(extra) + [os.path.join(CSRC, s) for s in SOURCES + ([PK8_SOURCE] if "-DVIT_WITH_PK8" in extra else [])] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or SO


if __name__ == "__main__":
    build(force="-f" in sys.argv, verbose=True)
    print(SO)
