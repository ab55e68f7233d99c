"""Arithmetic mode of the ST-GCN block kernels: `exact` (default) or `split16` (opt-in).

exact    every product of the graph conv / temporal conv and their gradients is an fp32 MFMA product
         (v_mfma_f32_16x16x4_f32): the reference's arithmetic (stgcn_layers.py:50-67,399-439 run in fp32), the mode
         every headline number is quoted in.
split16  the same operators with every fp32 product formed on the 16-bit matrix pipe as three
         v_mfma_f32_16x16x32_f16 products of two-part fp16 operands, fp32 accumulation (csrc/split16.h).  Operands
         carry 22 of fp32's 24 significand bits; range is handled by per-tensor powers of two that never visit the
         host.  Selected by the environment variable P2R_MATH=split16, `set_mode('split16')`, or `with use('split16')`.

The mode is read when an op runs (forward), and the backward of an op follows the mode its forward ran in.

Range words.  A split16 kernel that takes a runtime tensor as an MFMA operand (the incoming gradients; the block
inputs of the graph conv) needs max |x| of that tensor to place it in fp16's range.  The kernel that WROTE the tensor
leaves that maximum as one uint32 in device memory (`announce`); the consumer picks it up by the tensor's identity
(`range_word`), or computes it with one extra read of the tensor when nobody announced it.
"""
import contextlib
import ctypes
import os
from collections import OrderedDict

import torch

from .. import _lib

MODES = ('exact', 'split16')
_mode = os.environ.get('P2R_MATH', 'exact') or 'exact'
if _mode not in MODES:
    raise ValueError(f"P2R_MATH={_mode!r}: expected one of {MODES}")


def mode():
    return _mode


def split16():
    return _mode == 'split16'


def set_mode(m):
    global _mode
    if m not in MODES:
        raise ValueError(f"math mode {m!r}: expected one of {MODES}")
    _mode = m


@contextlib.contextmanager
def use(m):
    prev = _mode
    set_mode(m)
    try:
        yield
    finally:
        set_mode(prev)


# ---- range words ------------------------------------------------------------------------------------------------------
_WORDS = OrderedDict()      # data_ptr -> (version counter, numel, word tensor); bounded, cleared per forward pass
_MAX_WORDS = 64
FALLBACK_PASSES = 0         # how many range words had to be computed by a separate pass (tests, profiling)


def new_word(device):
    return torch.empty(1, dtype=torch.int32, device=device)


def announce(t, word):
    """`word` (int32[1], device) holds the float bits of max |t| -- written by the kernel that produced `t`."""
    _WORDS[t.data_ptr()] = (t._version, t.numel(), word)
    while len(_WORDS) > _MAX_WORDS:
        _WORDS.popitem(last=False)


def range_word(t, keep=False):
    """The range word of `t`: the announced one when `t` is the very tensor (address, version, size) its producer
    announced, otherwise one pass over `t` (p2r_absmax_bits)."""
    global FALLBACK_PASSES
    e = _WORDS.get(t.data_ptr()) if keep else _WORDS.pop(t.data_ptr(), None)
    if e is not None and e[0] == t._version and e[1] == t.numel() and e[2].device == t.device:
        return e[2]
    word = new_word(t.device)
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().p2r_absmax_bits(ctypes.c_longlong(t.numel()), _lib.ptr(t), _lib.ptr(word),
                                              _lib.current_stream(t.device)), "absmax_bits")
    FALLBACK_PASSES += 1
    return word


def reset():
    _WORDS.clear()


# ---- weights: power-of-two scale and the two fp16 parts (device arithmetic only, no host round trip) --------------------
def weight_scale(w, dims):
    """w fp32; amax over `dims` -> (s, inv) = (2^S, 2^-S) with amax * 2^S in [2^12, 2^13); S = 0 where amax is 0."""
    amax = w.detach().abs().amax(dim=dims, keepdim=True)
    e = torch.frexp(amax)[1]                                   # amax = m * 2^e, m in [0.5, 1)
    S = torch.where(amax > 0, 13 - e, torch.zeros_like(e)).clamp_(-100, 100)
    one = torch.ones_like(amax)
    return torch.ldexp(one, S), torch.ldexp(one, -S)


def split_parts(ws):
    """ws fp32 weights (already scaled into fp16's range) -> (p, q, ps) fp16: p = fp16(ws), q = fp16(ws - p), and
    ps = 2^-11 p -- the partner of a runtime operand's residual, which the kernels keep scaled by 2^11 (csrc/split16.h)."""
    p = ws.to(torch.float16)
    pf = p.to(torch.float32)
    q = (ws - pf).to(torch.float16)
    ps = (pf * 2.0 ** -11).to(torch.float16)
    return p, q, ps


def pack_parts(W, n_lead):
    """W fp32 [lead..., ...] -> (flat [lead..., 3 M + 1] fp16, inv [lead..., 1] fp32), M = elements per leading index:
    the three fp16 planes (w1, w2, 2^-11 w1) of 2^S W one behind the other (in W's own element order) and one zero.  The
    kernels' operand layouts are then ONE index_select each along the last axis (`gather_layout`): the whole per-step
    weight preparation of a stack of blocks is a dozen launches, whatever the number of layouts."""
    lead = W.shape[:n_lead]
    s, inv = weight_scale(W, dims=tuple(range(n_lead, W.dim())))
    p, q, ps = split_parts(W.detach() * s)
    zero = torch.zeros(*lead, 1, dtype=torch.float16, device=W.device)
    flat = torch.cat([p.reshape(*lead, -1), q.reshape(*lead, -1), ps.reshape(*lead, -1), zero], dim=-1)
    return flat, inv.reshape(*lead, 1).contiguous()


_INDEX = {}     # (layout key, device) -> int64 index tensor


def gather_layout(flat, key, build):
    """flat from `pack_parts`; `build()` -> numpy int64 array of source positions in [0, 3 M] (3 M = the zero) for the
    operand layout `key`; cached per device."""
    k = (key, str(flat.device))
    idx = _INDEX.get(k)
    if idx is None:
        idx = _INDEX[k] = torch.from_numpy(build()).to(flat.device)
    return flat.index_select(-1, idx)

