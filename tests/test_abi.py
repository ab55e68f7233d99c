"""The C-ABI library loads on a CPU-only box and exports every symbol the
header declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from pose2room_amd import _lib


def test_library_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    # in-tree, next to the package, so the GPU-side loader sees it
    assert os.path.dirname(_lib.LIB_PATH).endswith("pose2room_amd")


def test_header_symbols_all_exported():
    names = _lib.declared_symbols()
    assert "p2r_ball_query" in names and "p2r_furthest_point_sampling" in names
    l = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(l, n)]
    assert not missing, f"declared in include/p2r_hip.h but not exported: {missing}"
    # ... and nothing is exported behind the header's back
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("p2r_")}
    undeclared = sorted(exported - set(names))
    assert not undeclared, f"exported but not declared in include/p2r_hip.h: {undeclared}"


def test_abi_identity():
    l = _lib.lib()
    assert l.p2r_abi_version() == _lib.ABI_VERSION == 3
    assert l.p2r_build_arch() == b"gfx950"


def test_reference_launcher_names_are_cited():
    """every _ext entry point cites the reference launcher it replaces"""
    text = open(_lib.HEADER_PATH).read()
    for wrapper in ["furthest_point_sampling_kernel_wrapper", "gather_points_kernel_wrapper",
                    "gather_points_grad_kernel_wrapper", "query_ball_point_kernel_wrapper",
                    "group_points_kernel_wrapper", "group_points_grad_kernel_wrapper",
                    "three_nn_kernel_wrapper", "three_interpolate_kernel_wrapper",
                    "three_interpolate_grad_kernel_wrapper"]:
        assert wrapper in text
    assert len(re.findall(r"src/[a-z_]+\.(?:cpp|cu):\d+", text)) >= 18


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(_lib.LIB_PATH))
    bad = []
    for d, _, files in os.walk(os.path.join(root, "pose2room_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, f"product code must not import the oracle: {bad}"


def test_ops_refuse_cpu_tensors():
    import torch
    from pose2room_amd.pointnet2_ops import _ext
    from pose2room_amd.net_utils.nn_distance import nn_distance
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.ball_query(torch.zeros(1, 2, 3), torch.zeros(1, 4, 3), 0.3, 4)
    with pytest.raises(RuntimeError):
        nn_distance(torch.zeros(1, 2, 3), torch.zeros(1, 4, 3))


def _tool(name):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, 'tools', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_static_schedule_is_in_sync_with_its_generator():
    """csrc/gcn3_sched.inc (committed, compiled into the library) must be what tools/gen_gcn_sched.py generates from
    the skeleton in stgcn_layers.Graph, and the library's pattern signatures must equal those of the run-time tables
    -- otherwise the statically scheduled kernels would silently never be taken (GraphTables.gen3 False).  The same
    check runs in `__graft_entry__.build()`."""
    _tool('check_build').check_schedule_sync()


def test_reserved_register_check_is_part_of_the_build():
    """csrc/stgcn_tconv3.hip keeps its A-operand sets in v224..v255 by name and caps the compiler at v223 with
    amdgpu_num_vgpr -- a target the allocator exceeds under pressure instead of spilling (round 3: a variant with 64
    more live registers in the epilogue silently overwrote the operands of the next tile).  The Makefile compiles
    every source that names registers to assembly with ITS flags and refuses to link on an offender; here: the stamp
    exists for the library under test, the checker finds the named registers, and it does flag a register the
    compiler uses."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, 'pose2room_amd', 'csrc')
    assert '$(CHECK)' in open(os.path.join(csrc, 'Makefile')).read()
    stamp = os.path.join(csrc, '.reserved_vgprs.ok')
    assert os.path.exists(stamp), 'libp2r_hip.so was linked without the reserved-register check'
    chk = _tool('check_reserved_vgprs')
    assert chk.named_registers(os.path.join(csrc, 'stgcn_tconv3.hip')) == set(range(224, 256))
    assert chk.named_registers(os.path.join(csrc, 'ball_query.hip')) == set()
    assert chk.offenders(os.path.join(csrc, 'ball_query.hip'), reserved={0, 1}), "the checker must see compiler-used registers"


def test_shape_restricted_entry_points_refuse_other_shapes():
    """Entry points that exist for one shape only return P2R_EINVAL for anything else -- before they touch the device,
    so this runs on a CPU-only box (the callers fall back to the generic launches on that status)."""
    l = _lib.lib()
    EINVAL = -22
    n = None
    # weight gradient + BatchNorm-backward apply: 53 joints, 3 taps only; NULL operands
    assert l.p2r_stgcn_tconv_weight_grad_dz(2, 16, 20, 3, n, n, n, n, n, n, 256, n, n, n) == EINVAL
    assert l.p2r_stgcn_tconv_weight_grad_dz(2, 16, 53, 1, n, n, n, n, n, n, 256, n, n, n) == EINVAL
    assert l.p2r_stgcn_tconv_weight_grad_dz(2, 16, 53, 3, n, n, n, n, n, n, 256, n, n, n) == EINVAL
    assert l.p2r_stgcn_tconv_weight_grad(2, 16, 65, 3, n, n, n, n, 256, n, n, n) == EINVAL
    assert l.p2r_stgcn_tconv_weight_grad(2, 16, 53, 2, n, n, n, n, 256, n, n, n) == EINVAL
