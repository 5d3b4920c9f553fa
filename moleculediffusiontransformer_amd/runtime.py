"""ctypes binding of libmdt_hip.so (C ABI: include/mdt_hip.h).

There is no CPU fallback: if the library cannot be loaded (or built with hipcc), every
entry point raises.  PyTorch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

N_EXT = 8
ABI_VERSION = 5        # MDT_ABI_VERSION of include/mdt_hip.h this binding was written against
ABI_TUNING_BIT = 0x40000000   # set in mdt_abi_version() by a -DMDT_TUNING build (csrc/mdt_kernels.h)
SP_NONE, SP_WEIGHT, SP_ACT, SP_SHR, SP_EXT0 = 0, 1, 2, 3, 4
OP_GEMM, OP_GN_STATS, OP_ATTN, OP_CONCAT, OP_PATCH, OP_TIME_EMBED, OP_TBLOCK, OP_GN_ACT, OP_RCONV = 1, 2, 3, 4, 5, 6, 7, 8, 9
OP_RESBLOCK = 10
OP_TF128 = 11
OP_TF256 = 12
OP_RES256 = 15
OP_ATTN_CTX = 13
OP_PREP16 = 14
W_KB = 23          # ring-kernel ops: KB of the weight stream (prefetched into the L2s by the previous launch)
OP_NAMES = {OP_GEMM: "k_gemm", OP_GN_STATS: "k_gn_stats", OP_ATTN: "k_attn", OP_CONCAT: "k_concat", OP_PATCH: "k_patch",
            OP_TIME_EMBED: "k_time_embed", OP_TBLOCK: "k_tblock", OP_GN_ACT: "k_gn_act", OP_RCONV: "k_rconv",
            OP_RESBLOCK: "k_resblock", OP_TF128: "k_tf128", OP_TF256: "k_tf256", OP_RES256: "k_res256", OP_ATTN_CTX: "k_attn_ctx", OP_PREP16: "k_prep16"}
TB_SELF, TB_CROSS, TB_FF = 0, 1, 2
PRO_NONE, PRO_LAYERNORM, PRO_GROUPNORM, PRO_SILU = 0, 1, 2, 3

# integer slots (enum mdt_gemm_i etc. in mdt_hip.h)
G_R_OUT, G_R_IN, G_LDA, G_CIN, G_TAPS, G_T_STRIDE, G_T_DJ, G_T_OFF, G_N, G_LDC, G_O_ROWS, G_O_STRIDE, \
    G_O_OFF, G_LDR, G_PRO, G_GROUPS, G_GSIZE, G_PRO_SILU, G_ACT, G_M_MODE, G_A_COL, G_O_COL, G_PHASES, G_WFMT = range(24)
N_ROWS, N_LD, N_GROUPS, N_GSIZE, N_SILU, N_OUT16, N_CA = range(7)
A_T, A_TK, A_HEADS, A_LDQ, A_LDKV, A_LDO, A_KV_BSTRIDE, A_OUT16, A_QCOL, A_KCOL, A_SPLIT, A_IN16 = range(12)
C_ROWS, C_CA, C_CB = range(3)
P_ROWS_IN, P_C_IN, P_LD_IN, P_LD_OUT, P_PATCH, P_INVERSE = range(6)
T_HALF, T_LD = range(2)
R_T, R_C, R_LDA, R_LDC, R_LDR, R_TAPS, R_GSIZE, R_SILU, R_FILM_LD, R_LDA2, R_WF32, R_KSRC, R_HALF_OUT, R_NB = range(14)
K_T, K_CIN, K_COUT, K_FILM_LD, K_WF32, K_CIN_REAL, K_COUT_REAL, K_PATCH_IN, K_PATCH_OUT = range(9)
B_MODE, B_C, B_T, B_NCHUNK, B_NBIAS, B_TK, B_KV_BSTRIDE, B_LDKV, B_HEADS, B_VARIANT, B_POST = range(11)
B_KV2, B_WF32 = 11, 12
# MDT_OP_TF128 (enum mdt_tf128_i)
(F_C, F_T, F_NT, F_NVEC, F_TK, F_KV_BSTRIDE, F_LDKV, F_HEADS, F_HAS_IN, F_NBLOCKS, F_NFF, F_NPOST, F_KV2, F_CROSS,
 F_KV_LSTRIDE, F_RES_KIND, F_N_RES, F_RES_PAIR1, F_RES_PAIR2, F_NFILM, F_NSPLIT, F_PAIR_STRIDE, F_WF32) = range(23)
FF_EPS_LN, FF_SCALE, FF_EPS_GN, FF_EPS_RES, FF_SKIP_SCALE = range(5)


class MdtRef(C.Structure):
    _fields_ = [("space", C.c_int32), ("reserved", C.c_int32), ("off", C.c_int64)]


class MdtOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32),
                ("a", MdtRef), ("a2", MdtRef), ("w", MdtRef), ("bias", MdtRef), ("out", MdtRef),
                ("res", MdtRef), ("p0", MdtRef), ("p1", MdtRef), ("p2", MdtRef), ("p3", MdtRef),
                ("i", C.c_int32 * 24), ("f", C.c_float * 8)]


class MdtBindings(C.Structure):
    _fields_ = [("weights", C.c_void_p), ("act", C.c_void_p), ("shr", C.c_void_p),
                ("ext", C.c_void_p * N_EXT)]


# every symbol include/mdt_hip.h declares: name -> (restype, argtypes)
_F, _P, _I, _L, _U64, _U32 = C.c_float, C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32
SYMBOLS = {
    "mdt_abi_version": (_I, []),
    "mdt_last_error": (C.c_char_p, []),
    "mdt_set_tuning": (_I, [C.c_char_p, _I]),
    "mdt_pair_capacity": (_I, []),
    "mdt_test_occupy": (_I, [_I, _I, _U64, _P]),
    "mdt_program_create": (_P, [C.POINTER(MdtOp), _I]),
    "mdt_program_destroy": (None, [_P]),
    "mdt_program_num_ops": (_I, [_P]),
    "mdt_program_run": (_I, [_P, C.POINTER(MdtBindings), _I, _I, _I, _I, _P]),
    "mdt_cond_embed": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mdt_cond_embed_add": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mdt_precond_in": (_I, [_P, _P, _F, _I, _I, _I, _I, _P]),
    "mdt_precond_out": (_I, [_P, _P, _P, _F, _F, _I, _I, _I, _I, _P, _P]),
    "mdt_dyn_scale": (_I, [_P, _P, _P, _F, _F, _F, _I, _I, _I, _I, _P]),
    "mdt_cfg_mix": (_I, [_P, _P, _P, _F, _L, _P]),
    "mdt_adpm2_mid": (_I, [_P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _I, _I, _I, _P, _P]),
    "mdt_adpm2_next": (_I, [_P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _U64, _U32, _L, _I, _I, _I, _I, _P, _P, _P]),
    "mdt_adpm2_euler": (_I, [_P, _P, _P, _P, _P, _F, _F, _F, _I, _U64, _U32, _L, _I, _I, _I, _P]),
    "mdt_init_noise": (_I, [_P, _P, _F, _U64, _U32, _L, _I, _I, _I, _P]),
    "mdt_clamp": (_I, [_P, _F, _F, _L, _P]),
    "mdt_copy_f32": (_I, [_P, _P, _L, _P]),
    "mdt_inpaint_merge": (_I, [_P, _P, _P, _P, _F, _U64, _U32, _L, _I, _I, _I, _P]),
    "mdt_add_noise": (_I, [_P, _P, _F, _U64, _U32, _L, _I, _I, _I, _P]),
    "mdt_argmax_tokens": (_I, [_P, _P, _I, _I, _I, _P]),
    "mdt_timer_create": (_P, [_I]),
    "mdt_timer_destroy": (None, [_P]),
    "mdt_timer_start": (_I, [_P, _P]),
    "mdt_timer_stop": (_I, [_P, _P]),
    "mdt_timer_collect": (_I, [_P, C.POINTER(C.c_float), _I]),
}

_lib: Optional[C.CDLL] = None


def load_library(allow_build: bool = True) -> C.CDLL:
    """Loads (building first if the in-tree .so is missing or stale) libmdt_hip.so."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.library_path()
    # MDT_NO_BUILD=1 (tuning A/B only, with MDT_LIB_TAG): load the tagged library as it is, e.g. one built from an older commit
    allow_build = allow_build and os.environ.get("MDT_NO_BUILD", "0") != "1"
    if allow_build and not _build.is_fresh():
        try:
            _build.build_library()
        except Exception as e:
            # A present-but-stale library would run OLD kernels against the current op encoding (confusing launch errors,
            # or tests passing against stale code), so it is refused unless explicitly allowed.
            if not os.path.exists(path) or os.environ.get("MDT_ALLOW_STALE", "0") != "1":
                raise RuntimeError(f"libmdt_hip.so is missing or older than its sources and could not be rebuilt: {e} "
                                   "(set MDT_ALLOW_STALE=1 to load the stale library anyway)") from e
            import warnings
            warnings.warn(f"libmdt_hip.so is STALE (sources changed, rebuild failed: {e}); loading it because MDT_ALLOW_STALE=1")
    if not os.path.exists(path):
        raise RuntimeError("libmdt_hip.so not found; run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    abi = lib.mdt_abi_version()
    if abi & ABI_TUNING_BIT:
        # built with -DMDT_TUNING: may contain ablation switches that give WRONG results (csrc/mdt_kernels.h)
        if os.environ.get("MDT_ALLOW_TUNING", "0") != "1":
            raise RuntimeError(f"{path} is a TUNING build (-DMDT_TUNING: timing-only switches that give wrong results may be "
                               "compiled in); it is refused for sampling.  Tools that time such builds set MDT_ALLOW_TUNING=1")
        abi &= ~ABI_TUNING_BIT
    if abi != ABI_VERSION:
        raise RuntimeError(f"libmdt_hip.so ABI version {lib.mdt_abi_version()} != {ABI_VERSION} expected by this package: rebuild it")
    _lib = lib
    return lib


def pair_capacity() -> int:
    """Workgroups of a pair-split 256-channel transformer launch that the current device keeps resident at once
    (mdt_pair_capacity: compute units x occupancy; 256 on an MI355X)."""
    n = int(load_library().mdt_pair_capacity())
    if n <= 0:
        raise RuntimeError("libmdt_hip: the device's co-residency capacity for pair-split launches could not be determined")
    return n


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("libmdt_hip: " + load_library().mdt_last_error().decode())


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


class Program:
    """Owned handle of an mdt_program."""

    def __init__(self, ops):
        lib = load_library()
        arr = (MdtOp * len(ops))(*ops)
        self._h = lib.mdt_program_create(arr, len(ops))
        if not self._h:
            raise RuntimeError("libmdt_hip: " + lib.mdt_last_error().decode())
        self.n_ops = len(ops)

    def run(self, bindings: MdtBindings, B: int, n_shared_rows: int = 0, first: int = 0, count: int = -1,
            stream: Optional[int] = None) -> None:
        lib = load_library()
        check(lib.mdt_program_run(self._h, C.byref(bindings), B, n_shared_rows, first, count,
                                  current_stream() if stream is None else stream))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and _lib is not None:
                _lib.mdt_program_destroy(self._h)
        except Exception:
            pass


class EventTimer:
    """HIP-event interval timer on the caller's stream (mdt_timer_*)."""

    def __init__(self, max_intervals: int):
        self._lib = load_library()
        self._h = self._lib.mdt_timer_create(max_intervals)
        if not self._h:
            raise RuntimeError("libmdt_hip: " + self._lib.mdt_last_error().decode())
        self.cap = max_intervals

    def start(self, stream=None):
        check(self._lib.mdt_timer_start(self._h, current_stream() if stream is None else stream))

    def stop(self, stream=None):
        check(self._lib.mdt_timer_stop(self._h, current_stream() if stream is None else stream))

    def collect(self):
        buf = (C.c_float * self.cap)()
        n = self._lib.mdt_timer_collect(self._h, buf, self.cap)
        if n < 0:
            raise RuntimeError("libmdt_hip: " + self._lib.mdt_last_error().decode())
        return [buf[i] for i in range(n)]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.mdt_timer_destroy(self._h)
        except Exception:
            pass
