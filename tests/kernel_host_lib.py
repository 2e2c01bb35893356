"""TEST INFRASTRUCTURE: builds tests/kernel_host/libkernel_host.so - the DEVICE arithmetic of the Smith-Waterman bodies
(frizbee_amd/csrc/dp_body.h, dp_cf.h) compiled for the host with ROCm's clang++ through a stand-in <hip/hip_runtime.h> - and loads
it with ctypes.  It lets the CPU test suite fuzz the code the GPU runs against the oracle.  Never imported by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "kernel_host")
CSRC = os.path.join(ROOT, "frizbee_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def available():
    return os.path.exists(CLANG)


def build():
    so = os.path.join(HERE, "libkernel_host.so")
    srcs = [os.path.join(HERE, "dp_host.cpp"), os.path.join(HERE, "shim", "hip", "hip_runtime.h")] + [
        os.path.join(CSRC, f) for f in ("dp_body.h", "dp_cf.h", "dp_cfm.h", "dp_quad.h", "dp_unicode.h", "kernels_common.h", "fzb_internal.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + os.path.join(HERE, "shim"), "-I" + CSRC,
                               "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function", "-pthread", "-o", so, os.path.join(HERE, "dp_host.cpp")])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.kh_dp_single.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.kh_window.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_uint32)]
        _lib.kh_dp_unicode.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint16), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.kh_unicode_window.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_uint32)]
        _lib.kh_unicode_regs.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint16), C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        _lib.kh_dp_multi.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.kh_dp_multi_long.argtypes = _lib.kh_dp_multi.argtypes
        _lib.kh_dp_quad.argtypes = _lib.kh_dp_multi.argtypes + [C.c_int]
        _lib.kh_window_typos.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_uint32)]
        _lib.kh_dp_batch.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return _lib


def dp_single(needle, hay, scoring, case_sensitive=False, include_prefix=True, swl=64, form=3, real=16):
    sc = (C.c_uint16 * 9)(*scoring)
    return lib().kh_dp_single(needle, len(needle), int(case_sensitive), sc, hay, len(hay), int(include_prefix), swl, form, real)


def window(needle, hay, case_sensitive=False):
    """((ws, we) of dp_body.h's search, (ws, we) of dp_cf.h's) for a haystack of at most 32 bytes"""
    out = (C.c_uint32 * 4)()
    assert lib().kh_window(needle, len(needle), int(case_sensitive), hay, len(hay), out) == 0
    return (out[0], out[1]), (out[2], out[3])


def dp_unicode(rows, hay, scoring, include_prefix=True, swl=64, real=None, form=0):
    """score of a single-chunk window by the unicode scorer; rows = [(scalar bytes[4], flipped bytes[4], utf8 length)] as the oracle's
    case_needle_unicode returns them (what fzb_matcher_create stores in NeedleDev::uc / uf / ulen)"""
    sc = (C.c_uint16 * 9)(*scoring)
    uc = b"".join(r[0] for r in rows)
    uf = b"".join(r[1] for r in rows)
    ul = bytes(r[2] for r in rows)
    return lib().kh_dp_unicode(uc, uf, ul, len(rows), sc, hay, len(hay), int(include_prefix), swl, swl // 2 if real is None else real, form)


def dp_unicode_multi(rows, hay, scoring, include_prefix=True, swl=64, is_u8=True, form=0):
    """score of a unicode window wider than one chunk (swl < len(hay) <= 1024): form 0 = dp_unicode_multi_chunk (first form), 1 =
    dp_unicode_multi_chunk_t with the UTF-8 shortcut where the window allows it, 2 = its general steps"""
    sc = (C.c_uint16 * 9)(*scoring)
    uc = b"".join(r[0] for r in rows)
    uf = b"".join(r[1] for r in rows)
    ul = bytes(r[2] for r in rows)
    return lib().kh_dp_unicode_multi(uc, uf, ul, len(rows), int(is_u8), sc, hay, len(hay), int(include_prefix), swl, form)


def unicode_window(rows, hay):
    """(start, end) of the 0-typo unicode window as the unicode scorer finds it for an accepted haystack"""
    out = (C.c_uint32 * 2)()
    assert lib().kh_unicode_window(b"".join(r[0] for r in rows), b"".join(r[1] for r in rows), bytes(r[2] for r in rows), len(rows), hay, len(hay), out) == 0
    return int(out[0]), int(out[1])


def unicode_regs(rows, hay, scoring, swl=64):
    """a haystack of at most 32 bytes through the short-corpus kernel's register path: ((ws, we) of unicode_window_regs, (ws, we) of
    unicode_window_first_last, score of the register path, score of the memory path over the same window)"""
    out = (C.c_uint32 * 6)()
    sc = (C.c_uint16 * 9)(*scoring)
    rc = lib().kh_unicode_regs(b"".join(r[0] for r in rows), b"".join(r[1] for r in rows), bytes(r[2] for r in rows), len(rows), sc, hay, len(hay), swl, out)
    assert rc == 0, rc
    return (int(out[0]), int(out[1])), (int(out[2]), int(out[3])), int(out[4]), int(out[5])


def dp_multi(needle, hay, scoring, case_sensitive=False, include_prefix=True, swl=64, form=6, is_u8=True):
    """score of a window wider than one chunk (swl < len(hay) <= 1024): form 5 = first form (dp_body.h), 6 = dp_cfm.h"""
    sc = (C.c_uint16 * 9)(*scoring)
    return lib().kh_dp_multi(needle, len(needle), int(case_sensitive), int(is_u8), sc, hay, len(hay), int(include_prefix), swl, form)


def dp_multi_long(needle, hay, scoring, case_sensitive=False, include_prefix=True, swl=32, form=6, is_u8=False):
    """score of a window of 1..1024 bytes under a LONG needle (any number of rows; NeedleLongDev): form 5 = first form, 6 = dp_cfm.h"""
    sc = (C.c_uint16 * 9)(*scoring)
    return lib().kh_dp_multi_long(needle, len(needle), int(case_sensitive), int(is_u8), sc, hay, len(hay), int(include_prefix), swl, form)


def dp_quad(needle, hay, scoring, case_sensitive=False, include_prefix=True, swl=64, form=0, is_u8=True, long_needle=False):
    """score of a window of 1..1024 bytes by dp_quad.h's four lanes (four host threads in lockstep): form 0 = row by row in registers, 1 = chunk
    by chunk with the rows parked (LDS layout), 2 = the slab layout requested a row ahead; long_needle: through NeedleLongRows"""
    sc = (C.c_uint16 * 9)(*scoring)
    return lib().kh_dp_quad(needle, len(needle), int(case_sensitive), int(is_u8), sc, hay, len(hay), int(include_prefix), swl, form, int(long_needle))


def window_typos(needle, hay, max_typos, case_sensitive=False):
    """(ws, we) of the short kernel's lane-free typo window for a haystack of at most 32 bytes"""
    out = (C.c_uint32 * 2)()
    assert lib().kh_window_typos(needle, len(needle), int(case_sensitive), max_typos, hay, len(hay), out) == 0
    return out[0], out[1]


def dp_batch(needle, hays, scoring, case_sensitive, include_prefix, swl, form, real):
    """hays: list of bytes; include_prefix: list of bools -> np.int32 scores"""
    sc = (C.c_uint16 * 9)(*scoring)
    blob = np.frombuffer(b"".join(hays), dtype=np.uint8).copy() if hays else np.zeros(1, np.uint8)
    lens = np.array([len(h) for h in hays], dtype=np.int32)
    ip = np.array(include_prefix, dtype=np.uint8)
    out = np.zeros(len(hays), dtype=np.int32)
    lib().kh_dp_batch(needle, len(needle), int(case_sensitive), sc, blob.ctypes.data, lens.ctypes.data, len(hays), ip.ctypes.data, swl, form, real, out.ctypes.data)
    return out
