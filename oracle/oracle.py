"""ctypes/numpy front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/vit_oracle.h).  The product path never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvitoracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("vit_oracle.c", "vit_avx2.c", "vit_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, u32p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.c_void_p
        L.vo_deconvolve.argtypes = [C.c_uint, vp, C.c_int, vp]
        L.vo_deconvolve_opt.argtypes = [C.c_uint, vp, vp, C.c_int]
        L.vo_deconvolve_u8.argtypes = [C.c_uint, vp, vp]
        L.vo_trace_state0_u8.argtypes = [C.c_uint, vp, C.c_int, vp]
        L.vo_decode_batch_u8.argtypes = [C.c_uint, vp, vp, C.c_long, C.c_int]
        L.vo_decode_batch_u8_opt.argtypes = [C.c_uint, vp, vp, C.c_long, C.c_int, C.c_int]
        L.vo_deconvolve_avx2_u8.argtypes = [C.c_uint, vp, vp]
        L.vo_decode_batch_avx2_u8.argtypes = [C.c_uint, vp, vp, C.c_long, C.c_int]
        L.vo_encode.argtypes = [C.c_uint, vp, vp]
        L.vo_encode.restype = None
        L.vo_fill_uniform.argtypes = [C.POINTER(C.c_uint64), vp, C.c_long]
        L.vo_fill_uniform.restype = None
        L.vo_make_noisy_frame.argtypes = [C.POINTER(C.c_uint64), C.c_uint, C.c_double, vp, vp]
        L.vo_make_noisy_frame.restype = None
        L.vo_fnv1a64.argtypes = [vp, C.c_long]
        L.vo_fnv1a64.restype = C.c_uint64
        L.vo_rs_tables.argtypes = [vp, vp]
        L.vo_rs_tables.restype = None
        L.vo_decode_rs.argtypes = [vp]
        L.vo_rs_check_superframe.argtypes = [vp, C.c_int, C.c_uint, vp]
        L.vo_rs_encode.argtypes = [vp, vp]
        L.vo_rs_encode.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


KAT_SEED = 88172645463325252


def sym_len(framebits):
    return 4 * (framebits + 6)


def uniform_symbols(n, seed=KAT_SEED):
    """n bytes of (xorshift64(13,7,17) >> 11) & 255, the SURVEY KAT stream."""
    st = C.c_uint64(seed)
    a = np.empty(n, np.uint8)
    lib().vo_fill_uniform(C.byref(st), _p(a), n)
    return a


def noisy_frames(nframes, framebits, seed=1, ebn0_db=3.0, return_bits=False):
    st = C.c_uint64(seed if seed else 1)
    sl = sym_len(framebits)
    sym = np.empty((nframes, sl), np.uint8)
    bits = np.empty((nframes, framebits), np.uint8)
    for f in range(nframes):
        lib().vo_make_noisy_frame(C.byref(st), framebits, ebn0_db, _p(sym[f]), _p(bits[f]))
    return (sym, bits) if return_bits else sym


def hard_random_symbols(nframes, framebits, seed=1):
    """i.i.d. hard-decision symbols 0/255 (no code structure): a family on which the `>150` and `>=150`
    renormalise comparators give different outputs (metrics reach the 0 and 255 clamps)."""
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 2, (nframes, sym_len(framebits)), dtype=np.uint8) * 255).astype(np.uint8)


def hard_flipped_frames(nframes, framebits, flip=0.2, seed=1):
    """random bits -> mother code -> hard symbols 0/255 with a fraction `flip` of the symbols inverted"""
    rng = np.random.default_rng(seed)
    out = np.empty((nframes, sym_len(framebits)), np.uint8)
    for f in range(nframes):
        hard = encode(rng.integers(0, 2, framebits, dtype=np.uint8))
        inv = rng.random(hard.size) < flip
        out[f] = np.where(inv, 1 - hard, hard) * 255
    return out


def encode(bits):
    bits = np.ascontiguousarray(bits, np.uint8)
    hard = np.empty(4 * (bits.size + 6), np.uint8)
    lib().vo_encode(bits.size, _p(bits), _p(hard))
    return hard


def deconvolve_u32(framebits, sym_u32, ge=False):
    sym_u32 = np.ascontiguousarray(sym_u32, np.uint32)
    assert sym_u32.size >= sym_len(framebits)
    out = np.zeros((framebits + 7) // 8, np.uint8)
    rc = lib().vo_deconvolve_opt(framebits, _p(sym_u32), _p(out), 1 if ge else 0)
    assert rc == 0
    return out


def trace_state0(framebits, sym_u8, ge=False):
    """metric of state 0 after every trellis step (after the renormalisation where there is one): framebits+6 values"""
    sym_u8 = np.ascontiguousarray(sym_u8, np.uint8)
    assert sym_u8.size == sym_len(framebits)
    tr = np.zeros(framebits + 6, np.uint8)
    assert lib().vo_trace_state0_u8(framebits, _p(sym_u8), 1 if ge else 0, _p(tr)) == 0
    return tr


def decode_batch(framebits, sym_u8, nthreads=1, avx2=False, ge=False):
    """sym_u8: (nframes, 4*(framebits+6)) uint8 -> (nframes, (framebits+7)//8) uint8
    ge: the MASM decoders' `>=150` renormalise comparator (decon_avx2.asm:97,114) instead of the C path's `>150`"""
    sym_u8 = np.ascontiguousarray(sym_u8, np.uint8).reshape(-1, sym_len(framebits))
    n = sym_u8.shape[0]
    out = np.zeros((n, (framebits + 7) // 8), np.uint8)
    if ge:
        assert not avx2, "the AVX2 port implements the C path's comparator only"
        rc = lib().vo_decode_batch_u8_opt(framebits, _p(sym_u8), _p(out), n, nthreads, 1)
    else:
        fn = lib().vo_decode_batch_avx2_u8 if avx2 else lib().vo_decode_batch_u8
        rc = fn(framebits, _p(sym_u8), _p(out), n, nthreads)
    if rc != 0:
        raise RuntimeError("oracle decode failed rc=%d" % rc)
    return out


def has_avx2():
    return bool(lib().vo_has_avx2())


def fnv1a64(a):
    a = np.ascontiguousarray(a, np.uint8)
    return int(lib().vo_fnv1a64(_p(a), a.size))


def rs_tables():
    ato = np.empty(768, np.uint8)
    iof = np.empty(256, np.uint8)
    lib().vo_rs_tables(_p(ato), _p(iof))
    return ato, iof


def rs_encode(msg):
    msg = np.ascontiguousarray(msg, np.uint8)
    assert msg.size == 110
    cw = np.empty(120, np.uint8)
    lib().vo_rs_encode(_p(msg), _p(cw))
    return cw


def rs_decode_word(word):
    d = np.ascontiguousarray(word, np.uint8).astype(np.uint32)
    assert d.size == 120
    rc = lib().vo_decode_rs(_p(d))
    return rc, d.astype(np.uint8)


def rs_check_superframe(p, rsdims, out=None):
    p = np.ascontiguousarray(p, np.uint8)
    assert p.size == 120 * rsdims
    if out is None:
        out = np.zeros(110 * rsdims, np.uint8)
    rc = lib().vo_rs_check_superframe(_p(p), 0, rsdims, _p(out))
    return rc, out


def rs_check_batch(p, rsdims, out_init=None):
    """p: (nsf, 120*rsdims) -> (ret[nsf] int32, out (nsf,110*rsdims))"""
    p = np.ascontiguousarray(p, np.uint8).reshape(-1, 120 * rsdims)
    n = p.shape[0]
    out = (np.zeros((n, 110 * rsdims), np.uint8) if out_init is None
           else np.array(out_init, np.uint8).reshape(n, 110 * rsdims).copy())
    ret = np.zeros(n, np.int32)
    for i in range(n):
        ret[i] = lib().vo_rs_check_superframe(_p(p[i]), 0, rsdims, _p(out[i]))
    return ret, out
