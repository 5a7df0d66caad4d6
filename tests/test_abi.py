"""CPU-only: libviterbi.so loads, exports every symbol include/viterbi_amd.h declares, and fails
loudly (documented error values, no CPU fallback) when no gfx950 device exists."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "viterbi_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src)))


def test_header_declares_the_reference_exports():
    names = _header_functions()
    for ref in ("deconvolve", "initialize", "RScheckSuperframe", "GetCPUCaps", "WakeUpYMM"):  # viterbi.def:4-8
        assert ref in names


def test_library_exports_every_declared_symbol(V):
    lib = V.lib()
    for name in _header_functions():
        assert hasattr(lib, name), "missing export " + name
    assert set(V.EXPORTS) == set(_header_functions())
    out = subprocess.check_output(["nm", "-D", "--defined-only", V.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert exported == set(_header_functions()), "export table differs from the header (exports.map)"


def test_frame_desc_layout(V):
    assert C.sizeof(V.FrameDesc) == 24 and V.DESC_DTYPE.itemsize == 24
    d, sb, ob = V.make_descs([768, 288, 8])
    assert d["sym_offset"].tolist() == [0, 3096, 3096 + 1176] and sb == 3096 + 1176 + 56
    assert d["out_offset"].tolist() == [0, 96, 132] and ob == 133


def test_sort_descs(V):
    d, _, _ = V.make_descs([768, 288, 3072, 768, 8, 3072])
    before = {(int(x["framebits"]), int(x["sym_offset"]), int(x["out_offset"])) for x in d}
    V.sort_descs(d)
    assert d["framebits"].tolist() == [3072, 3072, 768, 768, 288, 8]
    assert {(int(x["framebits"]), int(x["sym_offset"]), int(x["out_offset"])) for x in d} == before
    assert d["sym_offset"][0] < d["sym_offset"][1] and d["sym_offset"][2] < d["sym_offset"][3]  # stable


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_device_fails_loudly_without_cpu_fallback(V):
    assert V.device_count() == 0
    assert V.GetCPUCaps() == 0
    assert V.initialize() is True  # cheap and idempotent even without a device
    sym = np.full(4 * (768 + 6), 128, np.uint32)
    rc, out = V.deconvolve(768, sym)
    assert rc == 1 and not out.any()  # reference "save mode" value, output untouched
    assert "gfx950" in V.last_error()
    rc, _ = V.RScheckSuperframe(np.zeros(120 * 4, np.uint8), 0, 4)
    assert rc == -1
    with pytest.raises(V.ViterbiError):
        V.decode_batch_host(np.zeros((1, 4 * 774), np.uint8), 768)
    # framebits == 0 is a successful no-op like the reference's C path
    assert V.lib().deconvolve(0, None, 0, None) == 0
    # the multi-GPU entry: same loud failure, before RCCL is even looked for
    devs = (C.c_int * 2)(0, 1)
    assert V.lib().vit_decode_stream_multi(None, None, 768, 16, devs, 2, 4, -1, 0, None) == 2  # VIT_ERR_NO_DEVICE
    assert "gfx950" in V.last_error()


def test_call_log_env(tmp_path):
    """VITERBI_AMD_LOG=<file>: one line per exported call (analogue of VIT_WRITE_LOGFILE,
    deconvolve.cpp:568-649).  Runs in a child process because the variable is read at load time."""
    import sys
    log = tmp_path / "vit.log"
    code = ("import sys; sys.path.insert(0, %r); import _vitpkg, numpy as np; V = _vitpkg.load_package();"
            "V.deconvolve(768, np.zeros(4*774, np.uint32)); V.lib().deconvolve(0, None, 0, None);"
            "V.RScheckSuperframe(np.zeros(480, np.uint8), 0, 4)" % ROOT)
    env = dict(os.environ, VITERBI_AMD_LOG=str(log))
    subprocess.check_call([sys.executable, "-c", code], env=env)
    lines = log.read_text().strip().splitlines()
    assert len(lines) == 3
    assert "deconvolve" in lines[0] and "size: 768" in lines[0] and "ret:" in lines[0]
    assert "size: 0" in lines[1] and lines[1].rstrip().endswith("ret: 0")
    assert "RScheckSuperframe" in lines[2] and "size: 4" in lines[2]


def test_ingest_stage_environment_knobs():
    """VITERBI_AMD_BATCH_WINDOW_US / _MIN_CALLERS / _DEPTH and VITERBI_AMD_SPIN_CPUS configure the ingest stage for hosts
    that bind only the five reference exports (read once, when the library is loaded); the setters return the previous
    value.  Defaults: stage on (window 50 us), engaged from the first call in flight, three batches in flight, waiting
    callers spin while the calls in flight fit the process's CPU budget."""
    import sys
    code = ("import sys; sys.path.insert(0, %r); import _vitpkg; V = _vitpkg.load_package();"
            "print('RESULT', V.set_batch_window_us(0), V.set_batch_min_callers(8), V.set_batch_depth(5), V.set_batch_depth(99),"
            " V.set_batch_depth(0), V.set_batch_depth(2), V.set_batch_spin_cpus(7), V.set_batch_spin_cpus(-3), V.set_batch_spin_cpus(1))" % ROOT)
    env = dict(os.environ, VITERBI_AMD_BATCH_WINDOW_US="75", VITERBI_AMD_BATCH_MIN_CALLERS="12", VITERBI_AMD_BATCH_DEPTH="4",
               VITERBI_AMD_SPIN_CPUS="9")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True)
    assert "RESULT 75 12 4 5 16 1 9 7 0" in out  # depth is clamped to 1..16, spin_cpus to >= 0
    clean = {k: v for k, v in os.environ.items() if not k.startswith("VITERBI_AMD_BATCH") and k != "VITERBI_AMD_SPIN_CPUS"}
    out = subprocess.check_output([sys.executable, "-c", code], env=clean, text=True)
    ncpu = len(os.sched_getaffinity(0))
    m = re.search(r"RESULT 50 1 4 5 16 1 (\d+) 7 0", out)
    assert m and 0 <= int(m.group(1)) <= ncpu // 4  # a quarter of: affinity mask, capped by a cgroup CPU quota


def test_default_renormalise_comparator_is_the_shipped_dll_s():
    """a fresh process: the comparator defaults to `>= 150` (the reference's MASM decoders, configuration Rel_asm, which
    its README tells users to build - what an installed viterbi.dll runs); VITERBI_AMD_RENORM_GE=0 selects the C decoders"""
    import sys
    code = ("import sys; sys.path.insert(0, %r); import _vitpkg; V = _vitpkg.load_package();"
            "print('RESULT', V.set_renorm_ge(0), V.set_renorm_ge(1), V.set_renorm_ge(1))" % ROOT)
    clean = {k: v for k, v in os.environ.items() if k != "VITERBI_AMD_RENORM_GE"}
    assert "RESULT 1 0 1" in subprocess.check_output([sys.executable, "-c", code], env=clean, text=True)
    assert "RESULT 0 0 1" in subprocess.check_output([sys.executable, "-c", code], env=dict(clean, VITERBI_AMD_RENORM_GE="0"), text=True)
