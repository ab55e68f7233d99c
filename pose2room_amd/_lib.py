"""ctypes loader for libp2r_hip.so (C ABI declared in include/p2r_hip.h).

The library is built in-tree by `pose2room_amd/csrc/Makefile`
(`__graft_entry__.build()`).  Loading fails loudly: there is no Python or CPU
fallback for any entry point.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# P2R_LIB_PATH: another build of the SAME library (A/B timing of kernel variants on one box, tools/ab_bench.sh); the
# loader's checks (ABI version, every declared symbol) apply to it unchanged
LIB_PATH = os.environ.get("P2R_LIB_PATH") or os.path.join(_HERE, "libp2r_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "p2r_hip.h")

ABI_VERSION = 3     # p2r_abi_version() of the library this loader was written against (include/p2r_hip.h)

_lib = None


class P2RLibraryError(RuntimeError):
    pass


def declared_symbols(header_path=HEADER_PATH):
    """Names of every function the C header declares (used by the export test)."""
    with open(header_path) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(p2r_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P2RLibraryError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C pose2room_amd/csrc`). pose2room_amd has no CPU fallback.")
        try:
            l = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise P2RLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        l.p2r_abi_version.restype = ctypes.c_int
        l.p2r_build_arch.restype = ctypes.c_char_p
        if l.p2r_abi_version() != ABI_VERSION:
            raise P2RLibraryError(f"libp2r_hip.so ABI version {l.p2r_abi_version()} != {ABI_VERSION}: stale library, rebuild it "
                                  "(make -C pose2room_amd/csrc)")
        for name in declared_symbols():
            fn = getattr(l, name)
            if name in ("p2r_stgcn_gcn3_signature", "p2r_stgcn_gcn3h_signature", "p2r_stgcn_gcn3h_weight_grad_signature"):
                fn.restype = ctypes.c_uint64
            elif name not in ("p2r_build_arch",):
                fn.restype = ctypes.c_int
        _lib = l
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"libp2r_hip: {what} failed with status {status}")


def ptr(t):
    """Device pointer of a torch tensor (or NULL for None)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def current_stream(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def sum_leading(part, tr64=False):
    """part [P, ...] f32 (kernel partials, one row per workgroup) -> part.sum(0), optionally with every trailing
    64 x 64 block transposed; one streaming launch (csrc/bn_act.hip: p2r_sum_leading)."""
    import torch
    P = part.shape[0]
    M = part.numel() // P
    if not (part.is_cuda and part.dtype == torch.float32 and part.is_contiguous() and M % 4 == 0
            and part.data_ptr() % 16 == 0):
        out = part.sum(0)
        return out.transpose(-1, -2).contiguous() if tr64 else out
    out = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
    with torch.cuda.device(part.device):
        check(lib().p2r_sum_leading(P, ctypes.c_longlong(M), ptr(part), ptr(out), int(bool(tr64)),
                                    current_stream(part.device)), "sum_leading")
    return out
