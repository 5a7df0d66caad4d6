"""viterbi.dll_amd -- Python host mirror of the drop-in C ABI (libviterbi.so).

The product is the shared library built from csrc/ (hand-written HIP kernels for
gfx950 behind the reference's exported C functions, include/viterbi_amd.h).  This
module is the thin ctypes binding used by tests and bench.py; names and argument
meaning follow the reference exports (viterbi.def:4-8): ``deconvolve``,
``RScheckSuperframe``, ``initialize``, ``GetCPUCaps``, ``WakeUpYMM`` -- plus the
batched device-resident extension.  There is no CPU fallback here: if the
library is missing, or no gfx950 device is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITERBI_AMD_LIB") or os.path.join(_HERE, "libviterbi.so")  # env: kernel experiments
MAX_FRAMEBITS = 9216
TAIL = 6

EXPORTS = [
    "deconvolve", "initialize", "RScheckSuperframe", "RSCheckSuperframe", "GetCPUCaps", "WakeUpYMM",
    "vit_last_error", "vit_device_count", "vit_set_kernel", "vit_set_renorm_ge", "vit_set_batch_window_us", "vit_set_batch_min_callers", "vit_set_batch_depth", "vit_set_batch_spin_cpus",
    "vit_decode_batch_dev",
    "vit_decode_batch_dev_u32", "vit_decode_varlen_dev", "vit_decode_varlen_dev_checked", "vit_pack_symbols_dev", "vit_sort_descs",
    "vit_decode_batch_host", "vit_rs_batch_dev", "vit_rs_batch_host", "vit_dabplus_superframes_dev",
    "vit_decode_stream_multi",
]
MULTI_LOOPBACK = 0x1


class ViterbiError(RuntimeError):
    pass


class FrameDesc(C.Structure):
    """vit_frame_desc of include/viterbi_amd.h"""
    _fields_ = [("sym_offset", C.c_uint64), ("out_offset", C.c_uint64),
                ("framebits", C.c_uint32), ("reserved", C.c_uint32)]


DESC_DTYPE = np.dtype([("sym_offset", "<u8"), ("out_offset", "<u8"), ("framebits", "<u4"), ("reserved", "<u4")])

_lib = None


def build(force=False, extra=(), out=None):
    from importlib import util as _u
    spec = _u.spec_from_file_location("_vit_build", os.path.join(_HERE, "build.py"))
    mod = _u.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=force, extra=extra, out=out)


def lib():
    """Load libviterbi.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ViterbiError("libviterbi.so not built: run `python __graft_entry__.py build` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.deconvolve.argtypes = [C.c_uint, vp, C.c_int, vp]
        L.deconvolve.restype = C.c_int
        L.initialize.restype = C.c_ubyte
        L.RScheckSuperframe.argtypes = [vp, C.c_int, C.c_uint, vp]
        L.RSCheckSuperframe.argtypes = [vp, C.c_int, C.c_uint, vp]
        L.GetCPUCaps.restype = C.c_int
        L.WakeUpYMM.restype = None
        L.vit_last_error.restype = C.c_char_p
        L.vit_set_kernel.argtypes = [C.c_int]
        L.vit_set_renorm_ge.argtypes = [C.c_int]
        L.vit_set_batch_window_us.argtypes = [C.c_int]
        L.vit_set_batch_min_callers.argtypes = [C.c_int]
        L.vit_set_batch_depth.argtypes = [C.c_int]
        L.vit_set_batch_spin_cpus.argtypes = [C.c_int]
        L.vit_decode_batch_dev.argtypes = [vp, vp, C.c_uint32, C.c_int64, vp]
        L.vit_decode_batch_dev_u32.argtypes = [vp, vp, C.c_uint32, C.c_int64, vp]
        L.vit_decode_varlen_dev.argtypes = [vp, vp, vp, C.c_int64, C.c_uint32, vp]
        L.vit_decode_varlen_dev_checked.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, C.c_int64, C.c_uint32, vp]
        L.vit_pack_symbols_dev.argtypes = [vp, vp, C.c_int64, vp]
        L.vit_sort_descs.argtypes = [vp, C.c_int64]
        L.vit_sort_descs.restype = None
        L.vit_decode_batch_host.argtypes = [vp, vp, C.c_uint32, C.c_int64]
        L.vit_rs_batch_dev.argtypes = [vp, vp, vp, C.c_uint32, C.c_int64, vp]
        L.vit_rs_batch_host.argtypes = [vp, vp, vp, C.c_uint32, C.c_int64]
        L.vit_dabplus_superframes_dev.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_int64, vp]
        L.vit_decode_stream_multi.argtypes = [vp, vp, C.c_uint32, C.c_int64, vp, C.c_int, C.c_int64, C.c_int64, C.c_uint, vp]
        _lib = L
    return _lib


def last_error():
    return lib().vit_last_error().decode()


def _check(rc, what):
    if rc != 0:
        raise ViterbiError("%s failed (rc=%d): %s" % (what, rc, last_error()))


def _np(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- the reference's exports ------------------------------------------------

def initialize():
    return bool(lib().initialize())


def GetCPUCaps():
    return int(lib().GetCPUCaps())


def WakeUpYMM():
    lib().WakeUpYMM()


def device_count():
    return int(lib().vit_device_count())


def set_kernel(which):
    return int(lib().vit_set_kernel(int(which)))


def set_renorm_ge(on):
    """0: renormalise on `> 150` (reference C decoders, Rel_cpp); 1: on `>= 150` (MASM decoders, Rel_asm)"""
    return int(lib().vit_set_renorm_ge(1 if on else 0))


def set_batch_window_us(us):
    return int(lib().vit_set_batch_window_us(int(us)))


def set_batch_min_callers(n):
    return int(lib().vit_set_batch_min_callers(int(n)))


def set_batch_depth(n):
    return int(lib().vit_set_batch_depth(int(n)))


def set_batch_spin_cpus(n):
    return int(lib().vit_set_batch_spin_cpus(int(n)))


def deconvolve(framebits, symbols, unused=0, decoded=None):
    """int deconvolve(framebits, u32 symbols[4*(framebits+6)], unused, u8 out[framebits/8]).
    Returns (rc, decoded) with rc as the reference returns it (0 ok, 1 failure)."""
    symbols = np.ascontiguousarray(symbols, np.uint32)
    if symbols.size < 4 * (framebits + TAIL):
        raise ValueError("need 4*(framebits+6) symbols")
    if decoded is None:
        decoded = np.zeros((framebits + 7) // 8, np.uint8)
    rc = lib().deconvolve(framebits, _np(symbols), unused, _np(decoded))
    return int(rc), decoded


def RScheckSuperframe(p, startIx, RSDims, outVector=None):
    """int RScheckSuperframe(u8 p[120*RSDims], startIx, RSDims, u8 out[110*RSDims])."""
    p = np.ascontiguousarray(p, np.uint8)
    if p.size < 120 * RSDims:
        raise ValueError("need 120*RSDims bytes")
    if outVector is None:
        outVector = np.zeros(110 * RSDims, np.uint8)
    rc = lib().RScheckSuperframe(_np(p), startIx, RSDims, _np(outVector))
    return int(rc), outVector


# ---- batched extension, host buffers ----------------------------------------

def decode_batch_host(symbols_u8, framebits):
    symbols_u8 = np.ascontiguousarray(symbols_u8, np.uint8).reshape(-1, 4 * (framebits + TAIL))
    n = symbols_u8.shape[0]
    out = np.zeros((n, (framebits + 7) // 8), np.uint8)
    _check(lib().vit_decode_batch_host(_np(symbols_u8), _np(out), framebits, n), "vit_decode_batch_host")
    return out


def rs_batch_host(p, RSDims, out_init=None):
    p = np.ascontiguousarray(p, np.uint8).reshape(-1, 120 * RSDims)
    n = p.shape[0]
    out = (np.zeros((n, 110 * RSDims), np.uint8) if out_init is None
           else np.array(out_init, np.uint8).reshape(n, 110 * RSDims).copy())
    ret = np.zeros(n, np.int32)
    _check(lib().vit_rs_batch_host(_np(p), _np(out), _np(ret), RSDims, n), "vit_rs_batch_host")
    return ret, out


# ---- batched extension, device-resident (torch tensors are only the memory) ----

def _stream_ptr(stream):
    if stream is None:
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(int(stream))


def decode_batch_dev(d_symbols_u8, d_out, framebits, nframes, stream=None):
    """d_symbols_u8 / d_out: torch uint8 CUDA tensors (device format, see header)."""
    _check(lib().vit_decode_batch_dev(C.c_void_p(d_symbols_u8.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                      framebits, nframes, _stream_ptr(stream)), "vit_decode_batch_dev")


def decode_batch_dev_u32(d_symbols_u32, d_out, framebits, nframes, stream=None):
    _check(lib().vit_decode_batch_dev_u32(C.c_void_p(d_symbols_u32.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                          framebits, nframes, _stream_ptr(stream)), "vit_decode_batch_dev_u32")


def decode_varlen_dev(d_symbols_u8, d_out, d_desc, nframes, max_framebits, stream=None):
    _check(lib().vit_decode_varlen_dev(C.c_void_p(d_symbols_u8.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                       C.c_void_p(d_desc.data_ptr()), nframes, max_framebits,
                                       _stream_ptr(stream)), "vit_decode_varlen_dev")


def decode_varlen_dev_checked(d_symbols_u8, d_out, d_desc, nframes, max_framebits, stream=None, sym_bytes=None,
                              out_bytes=None):
    """descriptors that reach outside the two buffers (sizes default to the tensors' sizes) are skipped on the device"""
    _check(lib().vit_decode_varlen_dev_checked(
        C.c_void_p(d_symbols_u8.data_ptr()), d_symbols_u8.numel() if sym_bytes is None else sym_bytes,
        C.c_void_p(d_out.data_ptr()), d_out.numel() if out_bytes is None else out_bytes,
        C.c_void_p(d_desc.data_ptr()), nframes, max_framebits, _stream_ptr(stream)), "vit_decode_varlen_dev_checked")


def sort_descs(desc):
    """in-place, longest first (host numpy array of DESC_DTYPE)"""
    assert desc.dtype == DESC_DTYPE and desc.flags["C_CONTIGUOUS"]
    lib().vit_sort_descs(_np(desc), desc.size)
    return desc


def pack_symbols_dev(d_symbols_u32, d_symbols_u8, nsym, stream=None):
    _check(lib().vit_pack_symbols_dev(C.c_void_p(d_symbols_u32.data_ptr()), C.c_void_p(d_symbols_u8.data_ptr()),
                                      nsym, _stream_ptr(stream)), "vit_pack_symbols_dev")


def rs_batch_dev(d_p, d_out, d_ret, RSDims, nsf, stream=None):
    _check(lib().vit_rs_batch_dev(C.c_void_p(d_p.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                  C.c_void_p(d_ret.data_ptr()), RSDims, nsf, _stream_ptr(stream)),
           "vit_rs_batch_dev")


def dabplus_superframes_dev(d_symbols_u8, d_work, d_rs_out, d_ret, RSDims, nsf, stream=None):
    """decode 5*nsf frames of 192*RSDims bits, then RScheckSuperframe on every group of five"""
    _check(lib().vit_dabplus_superframes_dev(C.c_void_p(d_symbols_u8.data_ptr()), C.c_void_p(d_work.data_ptr()),
                                             C.c_void_p(d_rs_out.data_ptr()), C.c_void_p(d_ret.data_ptr()),
                                             RSDims, nsf, _stream_ptr(stream)), "vit_dabplus_superframes_dev")


def decode_stream_multi(d_symbols_u8, d_out, framebits, nframes, devices, chunk_frames, root_frames=-1, flags=0,
                        stream=None):
    """ONE process, several GPUs (include/viterbi_amd.h Part 3): the stream lives on devices[0]; synchronous."""
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    _check(lib().vit_decode_stream_multi(C.c_void_p(d_symbols_u8.data_ptr()), C.c_void_p(d_out.data_ptr()), framebits,
                                         nframes, devs, len(devices), chunk_frames, root_frames, flags,
                                         _stream_ptr(stream)), "vit_decode_stream_multi")


def make_descs(framebits_list, sym_align=4):
    """Contiguous layout for a variable-length batch -> (desc array, sym bytes, out bytes)."""
    fb = np.asarray(framebits_list, np.int64)
    sym_sz = 4 * (fb + TAIL)
    out_sz = (fb + 7) // 8
    d = np.zeros(fb.size, DESC_DTYPE)
    d["framebits"] = fb
    d["sym_offset"] = np.concatenate(([0], np.cumsum(sym_sz)[:-1]))
    d["out_offset"] = np.concatenate(([0], np.cumsum(out_sz)[:-1]))
    return d, int(sym_sz.sum()), int(out_sz.sum())
