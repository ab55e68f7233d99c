"""Build-time consistency checks that must hold before libp2r_hip.so is used (run by `__graft_entry__.build()` and by
tests/test_abi.py):

* `check_schedule_sync()` -- csrc/gcn3_sched.inc and csrc/gcn3h_sched_{c,r}.inc (committed, compiled into the library)
  are exactly what tools/gen_gcn_sched.py / tools/gen_gcn_split_sched.py generate from the skeleton in stgcn_layers.Graph, and the pattern signatures the library
  carries equal those of the run-time tables; otherwise the statically scheduled kernels would silently never be taken.
* the reserved-register check of tools/check_reserved_vgprs.py runs from the Makefile itself (target
  `.reserved_vgprs.ok`, with the Makefile's own compiler and flags); `check_reserved_registers()` re-runs it on demand.
"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'tools', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def check_schedule_sync():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    gen = _load('gen_gcn_sched')
    from pose2room_amd.p2rnet import gcn_tables, gcn_op
    from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
    A = Graph().A
    out = []
    for form, tr in ((0, False), (1, True)):
        nbr, gidx, Lk = gcn_tables.build(A, transpose=tr)
        gen.emit_form(form, nbr, gidx, Lk, out)
        if form == 1:
            gen.emit_coef_grad(nbr, gidx, Lk, out)
            gen.emit_weight_grad(nbr, gidx, Lk, out)
    committed = open(os.path.join(ROOT, 'pose2room_amd', 'csrc', 'gcn3_sched.inc')).read()
    if '\n'.join(out) not in committed:
        raise RuntimeError('csrc/gcn3_sched.inc is stale: run python tools/gen_gcn_sched.py and rebuild')
    if not gcn_op.GraphTables(A).gen3:
        raise RuntimeError('the pattern signatures of libp2r_hip.so differ from the run-time graph tables: '
                           'the statically scheduled kernels would never be taken')
    # the split16 kernels' schedules (csrc/gcn3h_sched_{c,r}.inc, tools/gen_gcn_split_sched.py)
    gen_h = _load('gen_gcn_split_sched')
    for form in ('c', 'r'):
        if open(gen_h.path(form)).read() != gen_h.generate(form):
            raise RuntimeError(f'csrc/gcn3h_sched_{form}.inc is stale: run python tools/gen_gcn_split_sched.py and rebuild')
    gen_dw = _load('gen_gcn_split_dw_sched')
    if open(gen_dw.path()).read() != gen_dw.generate():
        raise RuntimeError('csrc/gcn3dwh_sched.inc is stale: run python tools/gen_gcn_split_dw_sched.py and rebuild')
    if not gcn_op.GraphTables(A).gen3h:
        raise RuntimeError('the split16 schedules of libp2r_hip.so were generated for another adjacency pattern')


def check_reserved_registers():
    csrc = os.path.join(ROOT, 'pose2room_amd', 'csrc')
    subprocess.check_call(['make', '-C', csrc, '-B', '.reserved_vgprs.ok'])


if __name__ == '__main__':
    check_schedule_sync()
    print('schedule in sync')
