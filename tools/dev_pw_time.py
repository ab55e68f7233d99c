"""Event-timed durations of the point-wise layer kernels (csrc/pw_layers.hip) and the fused vote aggregation at the
bench shapes (vote head: 32 x 512 columns, 256 -> 256; proposal head: 32 x 128 columns, four 128 -> 128 jobs).
    python tools/dev_pw_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pose2room_amd import _lib
from pose2room_amd.p2rnet import pw_op

dev = torch.device('cuda:0')
st = _lib.current_stream(dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def rnd(*s):
    return torch.randn(*s, device=dev)


B, S, K = 32, 512, 128
x_nlc, x_ncl, z = rnd(B, S, 256), rnd(B, 256, S), rnd(B, 256, S)
W = rnd(256, 256) * 0.05
W3, b3 = rnd(259, 256) * 0.05, rnd(259)
out, out3 = torch.empty(B, 256, S, device=dev), torch.empty(B, S, 259, device=dev)
fin = torch.stack([rnd(256), rnd(256).abs() + 0.5, rnd(256), rnd(256)]).contiguous()
coef = torch.stack([rnd(256), rnd(256), rnd(256)]).contiguous()
stats = torch.empty(B * S // 64, 256, 3, device=dev)
bst = torch.empty(B * S // 64, 256, 2, device=dev)
A = pw_op._at
rows = []
rows.append(('vote L1 fwd (NLC in, stats)', lambda: pw_op._gemm([dict(x=A(x_nlc), x_nlc=1, x_ctot=256, w=A(W), out=A(out), out_ctot=256, stats=A(stats), k=256, rows=256)], B, S, st), 2 * 256 * 256 * B * S))
rows.append(('vote L2 fwd (BN in, stats)', lambda: pw_op._gemm([dict(x=A(x_ncl), x_ctot=256, tr=A(fin, 512), tr_mode=1, tr_ld=256, w=A(W), out=A(out), out_ctot=256, stats=A(stats), k=256, rows=256)], B, S, st), 2 * 256 * 256 * B * S))
rows.append(('vote L3 fwd (259 rows, NLC out)', lambda: pw_op._gemm([dict(x=A(x_ncl), x_ctot=256, tr=A(fin, 512), tr_mode=1, tr_ld=256, w=A(W3), bias=A(b3), out=A(out3), out_ctot=259, out_nlc=1, k=256, rows=259)], B, S, st), 2 * 259 * 256 * B * S))
rows.append(('vote L3 dgrad (k=259 NLC, mask+sums)', lambda: pw_op._gemm([dict(x=A(out3), x_nlc=1, x_ctot=259, w=A(W3), w_t=1, out=A(out), out_ctot=256, stats=A(bst), k=259, rows=256, epilogue=1, mz=A(z), mz_ctot=256, mfin=A(fin), mfin_ld=256)], B, S, st), 2 * 259 * 256 * B * S))
rows.append(('vote L2 dgrad (lazy, mask+sums)', lambda: pw_op._gemm([dict(x=A(x_ncl), x2=A(z), tr=A(coef), tr_mode=2, tr_ld=256, x_ctot=256, w=A(W), w_t=1, out=A(out), out_ctot=256, stats=A(bst), k=256, rows=256, epilogue=1, mz=A(z), mz_ctot=256, mfin=A(fin), mfin_ld=256)], B, S, st), 2 * 256 * 256 * B * S))
for sp in (8, 16, 32, 64):
    pw = torch.empty(sp, 256, 256, device=dev)
    rows.append((f'vote wgrad 256x256 split {sp}', lambda pw=pw, sp=sp: pw_op._wgrad([dict(x=A(x_ncl), x2=A(z), tr=A(coef), tr_mode=2, tr_ld=256, x_ctot=256, rows=256, y=A(z), y_ctot=256, ytr=A(fin, 512), ytr_ld=256, k=256, dw_part=A(pw), split=sp)], B, S, st), 2 * 256 * 256 * B * S))
# proposal level: four 128 -> 128 jobs on 32 x 128 columns
xp, op_ = rnd(B, 512, K), torch.empty(B, 512, K, device=dev)
Wp = [rnd(128, 128) * 0.1 for _ in range(4)]
finp = torch.stack([rnd(512), rnd(512).abs() + 0.5, rnd(512), rnd(512)]).contiguous()
sp_ = [torch.empty(B * K // 64, 128, 3, device=dev) for _ in range(4)]
rows.append(('proposal level fwd (4 x 128->128)', lambda: pw_op._gemm([dict(x=A(xp, j * 128 * K), tr=A(finp, 2 * 512 + 128 * j), tr_mode=1, tr_ld=512, w=A(Wp[j]), out=A(op_, j * 128 * K), stats=A(sp_[j]), k=128, rows=128, x_ctot=512, out_ctot=512) for j in range(4)], B, K, st), 4 * 2 * 128 * 128 * B * K))
for name, fn, flops in rows:
    us = timed(fn)
    print(f'{name:44s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s')

# fused vote aggregation
from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetSAModuleVotes
sa = PointnetSAModuleVotes(npoint=128, radius=0.3, nsample=16, mlp=[256, 256, 256], use_xyz=False, normalize_xyz=True, bn=False).to(dev)
xyz = (torch.cumsum(torch.randn(B, S, 3, device=dev) * 0.05, 1)).contiguous()
feat = rnd(B, 256, S).requires_grad_(True)
def fwd():
    return sa(xyz, feat)
o = fwd()
go = torch.randn_like(o[1])
print(f'sa module forward (fps + sa_votes)       {timed(lambda: fwd()):8.1f} us')
def fb():
    feat.grad = None
    sa(xyz, feat)[1].backward(go)
print(f'sa module forward + backward             {timed(lambda: fb()):8.1f} us')

with torch.no_grad():
    print(f'sa module forward, inference (no G / H / arg-max stores) {timed(lambda: sa(xyz, feat)):8.1f} us')
from pose2room_amd.pointnet2_ops import _ext
print(f'fps alone                                 {timed(lambda: _ext.furthest_point_sampling(xyz, 128)):8.1f} us')
