"""ctypes binding of libs2st_hip.so (C ABI declared in include/s2st_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no
CPU fallback: if the shared object is missing or no HIP device is visible, the first
compute call raises.  ``load_library(path)`` with an explicit path exists so the test-suite
can load the host build made against the wave64 emulator (tests/hipemu) -- the product
never does that.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(_HERE), "csrc", "libs2st_hip.so")

_lib: Optional[C.CDLL] = None
_lib_is_emulator = False


class Split(C.Structure):
    _fields_ = [("ld", C.c_int64), ("bs", C.c_int64), ("per", C.c_int32), ("_pad", C.c_int32)]


class GemmOperand(C.Structure):
    _fields_ = [("p", C.c_void_p), ("kmajor", C.c_int32), ("dtype", C.c_int32), ("sp", Split),
                ("zo", C.c_int64), ("zi", C.c_int64)]


class GemmOut(C.Structure):
    _fields_ = [("p", C.c_void_p), ("sp", Split), ("zo", C.c_int64), ("zi", C.c_int64), ("h", C.c_void_p)]


class GemmEpilogue(C.Structure):
    _fields_ = [("alpha", C.c_float), ("act", C.c_int32), ("bias", C.c_void_p),
                ("drop_p", C.c_float), ("accumulate", C.c_int32), ("seed", C.c_uint64),
                ("resid", C.c_void_p), ("mask_y", C.c_void_p), ("mask_scale", C.c_float), ("colsum", C.c_void_p),
                ("colsum_part", C.c_void_p), ("bias_zo", C.c_int64)]


class GemmArgs(C.Structure):
    _fields_ = [("A", GemmOperand), ("B", GemmOperand), ("C", GemmOut), ("ep", GemmEpilogue),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32),
                ("zdiv", C.c_int32), ("precise", C.c_int32), ("splitk", C.c_int32),
                ("kchunk", C.c_int32), ("avec", C.c_int32), ("bvec", C.c_int32), ("cvec", C.c_int32),
                ("tiles_n", C.c_int32), ("ws", C.c_void_p), ("ws_floats", C.c_int64), ("slab", C.c_void_p)]


class S2STHipError(RuntimeError):
    pass


def load_library(path: Optional[str] = None, emulator: bool = False) -> C.CDLL:
    """Load the kernel library.  ``emulator=True`` is for tests only."""
    global _lib, _lib_is_emulator
    path = path or os.environ.get("S2ST_HIP_LIB") or DEFAULT_LIB
    if not os.path.exists(path):
        raise S2STHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _lib = C.CDLL(path)
    _lib_is_emulator = emulator
    _lib.s2st_version.restype = C.c_int
    _lib.s2st_device_count.restype = C.c_int
    return _lib


def lib() -> C.CDLL:
    if _lib is None:
        load_library()
    return _lib


def is_emulator() -> bool:
    return _lib_is_emulator


def check(rc: int, what: str):
    if rc != 0:
        raise S2STHipError(f"{what} failed with code {rc}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr() -> Optional[int]:
    """Current torch HIP stream as a raw hipStream_t (None = default stream / emulator)."""
    if _lib_is_emulator or not torch.cuda.is_available():
        return None
    return torch.cuda.current_stream().cuda_stream


def require_device(t: torch.Tensor):
    if not _lib_is_emulator and not t.is_cuda:
        raise S2STHipError("s2st HIP ops need device (cuda/HIP) tensors; there is no CPU path")


def make_split(ld: int, per: int = 0, bs: int = 0) -> Split:
    return Split(ld, bs, per, 0)


def gemm(A: torch.Tensor, B: torch.Tensor, Cout: torch.Tensor, M: int, N: int, K: int, *,
         a_kmajor=True, a_ld=None, a_per=0, a_bs=0, a_zo=0, a_zi=0,
         b_kmajor=True, b_ld=None, b_per=0, b_bs=0, b_zo=0, b_zi=0,
         c_ld=None, c_per=0, c_bs=0, c_zo=0, c_zi=0,
         batch=1, zdiv=1, alpha=1.0, bias=None, act=0, drop_p=0.0, seed=0, resid=None,
         accumulate=False, precise=False, a_off=0, b_off=0, c_off=0, c_bf16=None, ws=None,
         mask_y=None, mask_scale=1.0, colsum=None, return_tile=False):
    """Raw GEMM entry (s2st_gemm_f32). Offsets are in elements.  A and B are both fp32 or both
    torch.bfloat16 tensors; ``Cout`` (fp32) may be None when only ``c_bf16`` is wanted."""
    require_device(A)
    if A.dtype != B.dtype:
        raise S2STHipError("gemm operands must have the same dtype")
    if A.dtype == torch.bfloat16:
        return _gemm_bf16(A, B, Cout, M, N, K, locals())
    g = GemmArgs()
    g.A = GemmOperand(A.data_ptr() + 4 * a_off, 1 if a_kmajor else 0, 0,
                      make_split(a_ld if a_ld is not None else (K if a_kmajor else M), a_per, a_bs),
                      a_zo, a_zi)
    g.B = GemmOperand(B.data_ptr() + 4 * b_off, 1 if b_kmajor else 0, 0,
                      make_split(b_ld if b_ld is not None else (K if b_kmajor else N), b_per, b_bs),
                      b_zo, b_zi)
    g.C = GemmOut(Cout.data_ptr() + 4 * c_off, make_split(c_ld if c_ld is not None else N, c_per, c_bs),
                  c_zo, c_zi, None)
    g.ep = GemmEpilogue(alpha, act, ptr(bias), drop_p, 1 if accumulate else 0, seed, ptr(resid), None, 1.0, None)
    g.M, g.N, g.K, g.batch, g.zdiv, g.precise = M, N, K, batch, zdiv, 1 if precise else 0
    check(lib().s2st_gemm_f32(C.byref(g), C.c_void_p(stream_ptr())), "s2st_gemm_f32")


def _gemm_bf16(A, B, Cout, M, N, K, kw):
    g = GemmArgs()
    a_ld = kw["a_ld"] if kw["a_ld"] is not None else (K if kw["a_kmajor"] else M)
    b_ld = kw["b_ld"] if kw["b_ld"] is not None else (K if kw["b_kmajor"] else N)
    g.A = GemmOperand(A.data_ptr() + 2 * kw["a_off"], 1 if kw["a_kmajor"] else 0, 1,
                      make_split(a_ld, kw["a_per"], kw["a_bs"]), kw["a_zo"], kw["a_zi"])
    g.B = GemmOperand(B.data_ptr() + 2 * kw["b_off"], 1 if kw["b_kmajor"] else 0, 1,
                      make_split(b_ld, kw["b_per"], kw["b_bs"]), kw["b_zo"], kw["b_zi"])
    h = kw["c_bf16"]
    g.C = GemmOut(Cout.data_ptr() + 4 * kw["c_off"] if Cout is not None else None,
                  make_split(kw["c_ld"] if kw["c_ld"] is not None else N, kw["c_per"], kw["c_bs"]),
                  kw["c_zo"], kw["c_zi"], h.data_ptr() + 2 * kw["c_off"] if h is not None else None)
    g.ep = GemmEpilogue(kw["alpha"], kw["act"], ptr(kw["bias"]), kw["drop_p"], 1 if kw["accumulate"] else 0,
                        kw["seed"], ptr(kw["resid"]), ptr(kw.get("mask_y")), kw.get("mask_scale", 1.0),
                        ptr(kw.get("colsum")))
    g.M, g.N, g.K, g.batch, g.zdiv, g.precise = M, N, K, kw["batch"], kw["zdiv"], 0
    if kw["ws"] is not None:
        g.ws, g.ws_floats = kw["ws"].data_ptr(), kw["ws"].numel()
    if kw.get("return_tile"):
        t = C.c_int32(0)
        check(lib().s2st_gemm_tile_f32(C.byref(g), C.byref(t), C.c_void_p(stream_ptr())), "s2st_gemm_tile_f32")
        return (t.value // 1000, t.value % 1000)
    check(lib().s2st_gemm_f32(C.byref(g), C.c_void_p(stream_ptr())), "s2st_gemm_f32")


def gemm_args_bf16(A, B, Cout, M, N, K, *, a_kmajor=True, b_kmajor=True, a_ld=None, b_ld=None, c_ld=None, bias=None,
                   act=0, resid=None, accumulate=False, c_bf16=None, alpha=1.0) -> GemmArgs:
    """One batch-1 bf16 problem for ``gemm_group`` (plain strides)."""
    g = GemmArgs()
    g.A = GemmOperand(A.data_ptr(), 1 if a_kmajor else 0, 1, make_split(a_ld or (K if a_kmajor else M)), 0, 0)
    g.B = GemmOperand(B.data_ptr(), 1 if b_kmajor else 0, 1, make_split(b_ld or (K if b_kmajor else N)), 0, 0)
    g.C = GemmOut(ptr(Cout), make_split(c_ld or N), 0, 0, ptr(c_bf16))
    g.ep = GemmEpilogue(alpha, act, ptr(bias), 0.0, 1 if accumulate else 0, 0, ptr(resid), None, 1.0, None)
    g.M, g.N, g.K, g.batch, g.zdiv, g.precise = M, N, K, 1, 1, 0
    return g


def gemm_group(problems):
    """s2st_gemm_group_f32: up to 8 bf16 problems of the same operand layouts in one persistent launch."""
    arr = (GemmArgs * len(problems))(*problems)
    fn = lib().s2st_gemm_group_f32
    fn.argtypes = [C.POINTER(GemmArgs), C.c_int32, C.c_void_p]
    check(fn(arr, len(problems), C.c_void_p(stream_ptr())), "s2st_gemm_group_f32")


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("ldq", C.c_int64), ("ldk", C.c_int64),
                ("ldv", C.c_int64), ("o", C.c_void_p), ("oh", C.c_void_p), ("lse", C.c_void_p), ("klen", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("T", C.c_int32), ("S", C.c_int32), ("dh", C.c_int32),
                ("causal", C.c_int32), ("scale", C.c_float), ("drop_p", C.c_float), ("seed", C.c_uint64),
                ("ld_drop", C.c_int32), ("doh", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
                ("dqh", C.c_void_p), ("dkh", C.c_void_p), ("dvh", C.c_void_p), ("dbq", C.c_void_p), ("dbk", C.c_void_p),
                ("dbv", C.c_void_p)]


def flash_attention(q, k, v, H, *, klen=None, causal=False, scale=None, drop_p=0.0, seed=0, dO=None, bf16_grads=False,
                    bf16_o=False, scratch=None):
    """Fused attention on bf16 [B, T, H*dh] / [B, S, H*dh] projections (s2st_flash_attn_*_bf16).
    Returns (o fp32, lse) and, when dO (fp32 [B, T, H*dh]) is given, also (dq, dk, dv) fp32.  ``bf16_o``: the forward also
    leaves the bf16 copy of o (what the engine does); the backward then forms D = rowsum(dO * o) inside its kernels from
    the bf16 copies instead of running the fp32 row kernel first."""
    require_device(q)
    B, T, Cm = q.shape
    S = k.shape[1]
    dh = Cm // H
    a = AttnArgs()
    a.q, a.k, a.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    a.ldq = a.ldk = a.ldv = Cm
    o = torch.zeros(B, T, Cm, dtype=torch.float32, device=q.device)
    lse = torch.zeros(B * H * T, dtype=torch.float32, device=q.device)
    oh = torch.zeros(B, T, Cm, dtype=torch.bfloat16, device=q.device) if bf16_o else None
    a.o, a.oh, a.lse, a.klen = o.data_ptr(), ptr(oh), lse.data_ptr(), ptr(klen)
    a.B, a.H, a.T, a.S, a.dh, a.causal = B, H, T, S, dh, 1 if causal else 0
    a.scale = scale if scale is not None else dh ** -0.5
    a.drop_p, a.seed, a.ld_drop = drop_p, seed, (S + 7) // 8 * 8
    lib().s2st_flash_attn_fwd_bf16.argtypes = [C.POINTER(AttnArgs), C.c_void_p]
    check(lib().s2st_flash_attn_fwd_bf16(C.byref(a), C.c_void_p(stream_ptr())), "s2st_flash_attn_fwd_bf16")
    if dO is None:
        return o, lse.view(B, H, T)
    doh = dO.to(torch.bfloat16).contiguous()
    dq, dk, dv = (torch.zeros_like(x, dtype=torch.float32) for x in (q, k, v))
    if scratch is None:  # (tools/attn_stamp.py passes a larger one: the private stamp build writes clock stamps there)
        scratch = torch.zeros(B * H * T, dtype=torch.float32, device=q.device)
    a.doh, a.dq, a.dk, a.dv = doh.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    extra = None
    if bf16_grads:  # also the bf16 copies + bias-gradient column sums
        extra = [torch.zeros_like(x, dtype=torch.bfloat16) for x in (q, k, v)] + \
                [torch.zeros(Cm, dtype=torch.float32, device=q.device) for _ in range(3)]
        a.dqh, a.dkh, a.dvh = (t.data_ptr() for t in extra[:3])
        a.dbq, a.dbk, a.dbv = (t.data_ptr() for t in extra[3:])
    lib().s2st_flash_attn_bwd_bf16.argtypes = [C.POINTER(AttnArgs), C.c_void_p, C.c_void_p, C.c_void_p]
    check(lib().s2st_flash_attn_bwd_bf16(C.byref(a), dO.data_ptr(), scratch.data_ptr(), C.c_void_p(stream_ptr())),
          "s2st_flash_attn_bwd_bf16")
    if extra is not None:
        return o, lse.view(B, H, T), dq, dk, dv, extra
    return o, lse.view(B, H, T), dq, dk, dv


# ---------------------------------------------------------------------------------------------
# generic call path: argtypes are derived from include/s2st_hip.h so the binding cannot drift
# from the declared C ABI
# ---------------------------------------------------------------------------------------------
import re

HEADER = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "s2st_hip.h")
_CT = {"int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "float": C.c_float,
       "int": C.c_int, "s2st_split": Split}
_protos = None


def header_prototypes():
    """{name: (restype, [(ctype, argname), ...])} parsed from the C header."""
    global _protos
    if _protos is not None:
        return _protos
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|int32_t|int64_t|void)\s+(s2st_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    alist.append((C.c_void_p, a.split("*")[-1].strip()))
                elif a == "void":
                    continue
                else:
                    ty, nm = a.rsplit(" ", 1)
                    alist.append((_CT[ty.replace("const ", "").strip()], nm))
        out[name] = ({"int64_t": C.c_int64, "void": None}.get(ret, C.c_int), alist)
    _protos = out
    return out


def _bind(name):
    fn = getattr(lib(), name)
    if getattr(fn, "_s2st_bound", False):
        return fn
    ret, alist = header_prototypes()[name]
    fn.restype = ret
    fn.argtypes = [t for t, _ in alist]
    fn._s2st_bound = True
    return fn


def call(name: str, *args):
    """Call a C-ABI entry point; torch tensors become device pointers, the trailing `stream`
    argument is filled with the current torch stream."""
    fn = _bind(name)
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            require_device(a)
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    conv.append(stream_ptr())
    rc = fn(*conv)
    check(rc, name)
