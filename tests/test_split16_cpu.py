"""CPU: host side of the opt-in split16 arithmetic (pose2room_amd.p2rnet.math_mode and the weight operand layouts of
gcn_op / tconv_op): power-of-two scales, the three fp16 planes, and every operand layout against its stated formula."""
import numpy as np
import pytest
import torch


def test_mode_switch_and_env():
    from pose2room_amd.p2rnet import math_mode
    assert math_mode.mode() == 'exact' and not math_mode.split16()        # the default (P2R_MATH unset in the test run)
    with math_mode.use('split16'):
        assert math_mode.split16()
        with math_mode.use('exact'):
            assert not math_mode.split16()
        assert math_mode.split16()
    assert math_mode.mode() == 'exact'
    with pytest.raises(ValueError):
        math_mode.set_mode('bf16')


def test_weight_scale_and_parts():
    from pose2room_amd.p2rnet import math_mode
    g = torch.Generator().manual_seed(0)
    W = torch.randn(5, 3, 64, 64, generator=g) * torch.tensor([1e-6, 1e-2, 1.0, 37.0, 1e4]).view(5, 1, 1, 1)
    W[1] = 0.0                                                                            # a zero tensor: scale 1
    s, inv = math_mode.weight_scale(W, dims=(1, 2, 3))
    assert torch.equal(s * inv, torch.ones_like(s))
    assert torch.equal(torch.frexp(s)[0], torch.full_like(s, 0.5))                        # exact powers of two
    amax = W.abs().amax(dim=(1, 2, 3)) * s.view(-1)
    assert float(s[1]) == 1.0
    for b in (0, 2, 3, 4):
        assert 2.0 ** 12 <= float(amax[b]) < 2.0 ** 13
    p, q, ps = math_mode.split_parts(W * s)
    assert p.dtype == q.dtype == ps.dtype == torch.float16 and torch.isfinite(p.float()).all()
    ws = (W * s).double()
    assert ((p.double() + q.double()) - ws).abs().max().item() <= 2.0 ** -22 * 2.0 ** 13   # 22 significand bits of the largest
    # 2^-11 w1: exact wherever the result is a normal fp16 number (|w1| >= 2^-3, i.e. weights above 2^-16 of the largest);
    # below that it is rounded to fp16's subnormal grid (absolute 2^-25 -> 2^-14 of w1's scale after the 2^11)
    big = p.double().abs() >= 2.0 ** -3
    assert torch.equal((ps.double() * 2048.0)[big], p.double()[big])
    assert (ps.double() * 2048.0 - p.double()).abs().max().item() <= 2.0 ** -14
    flat, inv2 = math_mode.pack_parts(W, 1)
    M = 3 * 64 * 64
    assert flat.shape == (5, 3 * M + 1) and torch.equal(inv2.view(-1), inv.view(-1))
    assert torch.equal(flat[:, :M], p.reshape(5, -1)) and torch.equal(flat[:, 2 * M:3 * M], ps.reshape(5, -1))
    assert float(flat[:, -1].abs().max()) == 0.0


def _rand_sel(shape, n, seed):
    rng = np.random.default_rng(seed)
    return [tuple(int(rng.integers(0, s)) for s in shape) for _ in range(n)]


def test_graph_conv_operand_layouts():
    """wh[pair][ph][part][m][16 kg + r][i] = part of 2^S W_{plane}[16 m + r][16 ph + kg + 4 (i & 3)] (and of W^T for the data
    gradient); wd[k][ph][part][ks][16 kg + r][i] = part of 2^S W_k[16 ph + r][32 ks + 16 (i >> 2) + 4 (i & 3) + kg]"""
    from pose2room_amd.p2rnet import gcn_op, math_mode
    g = torch.Generator().manual_seed(1)
    K = 11
    W = torch.randn(2, K, 64, 64, generator=g) / 8
    pairs = [(0, 1), (2, -1), (3, 5), (4, 6), (7, 9), (8, 10)]
    s, _ = math_mode.weight_scale(W, dims=(1, 2, 3))
    parts = math_mode.split_parts(W * s)
    for transposed in (False, True):
        wh, inv = gcn_op.split_planes(W, pairs, transposed=transposed)
        assert wh.shape == (2, 6, 4, 3, 4, 64, 8) and wh.is_contiguous() and inv.shape == (2, 1)
        for (b, pi, ph, part, m, lane, i) in _rand_sel(wh.shape, 400, 2):
            kg, r = lane >> 4, lane & 15
            plane = pairs[pi][i >> 2]
            row, col = 16 * m + r, 16 * ph + kg + 4 * (i & 3)
            want = 0.0 if plane < 0 else float(parts[part][b, plane, col, row] if transposed else parts[part][b, plane, row, col])
            assert float(wh[b, pi, ph, part, m, lane, i]) == want
    wd, inv = gcn_op.split_planes_coef_grad(W)
    assert wd.shape == (2, K, 4, 2, 2, 64, 8)
    for (b, k, ph, part, ks, lane, i) in _rand_sel(wd.shape, 400, 3):
        kg, r = lane >> 4, lane & 15
        assert float(wd[b, k, ph, part, ks, lane, i]) == float(parts[part][b, k, 16 * ph + r, 32 * ks + 16 * (i >> 2) + 4 * (i & 3) + kg])
    # every channel of a k-step is covered exactly once by its (kg, i) pairs
    cols = sorted(32 * 0 + 16 * (i >> 2) + 4 * (i & 3) + kg for kg in range(4) for i in range(8))
    assert cols == list(range(32))


def test_temporal_conv_operand_layouts():
    """wh[part][tap][ks][w][16 kg + r][i] = part of 2^S A[tap][16 w + r][32 ks + 8 kg + i]; A = W3 (forward) or
    A[p'] = W3[2 - p']^T (data gradient); both source layouts"""
    from pose2room_amd.p2rnet import math_mode, tconv_op
    g = torch.Generator().manual_seed(4)
    conv_w = torch.randn(3, 64, 64, 3, generator=g) / 8                       # (B, co, ci, tap): Conv2d weights of 3 blocks
    W3 = conv_w.permute(0, 3, 1, 2).contiguous()                              # (B, tap, co, ci)
    s, _ = math_mode.weight_scale(W3, dims=(1, 2, 3))
    parts = math_mode.split_parts(W3 * s)
    for gradient in (False, True):
        a, ainv = tconv_op.split_taps(conv_w, layout='cit', gradient=gradient)
        b_, binv = tconv_op.split_taps(W3, layout='tci', gradient=gradient)
        assert torch.equal(a, b_) and torch.equal(ainv, binv) and a.shape == (3, 3, 3, 2, 4, 64, 8)
        for (b, part, tap, ks, w, lane, i) in _rand_sel(a.shape, 400, 5):
            kg, r = lane >> 4, lane & 15
            arow, acol = 16 * w + r, 32 * ks + 8 * kg + i
            want = parts[part][b, 2 - tap, acol, arow] if gradient else parts[part][b, tap, arow, acol]
            assert float(a[b, part, tap, ks, w, lane, i]) == float(want)


def test_split_schedules_match_their_generator():
    """the committed schedules of the split16 graph-conv kernels equal what their generators produce from the skeleton
    (tools/check_build.py runs the same comparison at build time)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, forms in (('gen_gcn_split_sched', ('c', 'r')), ('gen_gcn_split_dw_sched', (None,))):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, 'tools', name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for f in forms:
            path, text = (mod.path(f), mod.generate(f)) if f else (mod.path(), mod.generate())
            assert open(path).read() == text, path


def test_split_schedule_unit_counts():
    """the FLOP model of bench.py's split16 roofline counts the units the generated schedules contain"""
    import re, os
    from pose2room_amd.p2rnet import gcn_op
    from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
    tables = gcn_op.GraphTables(Graph().A)
    if not tables.gen3h:
        pytest.skip('library not built for this skeleton')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def units(name, macro):
        return int(re.search(r'#define %s (\d+)' % macro, open(os.path.join(root, 'pose2room_amd', 'csrc', name)).read()).group(1))
    assert gcn_op.split_unit_counts(tables) == (units('gcn3h_sched_c.inc', 'H3_UNITS'), units('gcn3h_sched_r.inc', 'H3_UNITS'))
    assert gcn_op.split_weight_grad_units(tables) == units('gcn3dwh_sched.inc', 'DW_UNITS')
