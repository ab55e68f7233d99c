import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
from pose2room_amd import _lib
from pose2room_amd.pointnet2_ops import fused, _ext
from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetSAModuleVotes
from tests import cases
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, N, M = 2, 512, 128
mod = PointnetSAModuleVotes(npoint=M, radius=0.3, nsample=16, mlp=[256, 256, 256], use_xyz=False, normalize_xyz=True, bn=False).to(dev)
xyz = cases.cloud(B, N, 21, 'walk').to(dev)
feats = torch.randn(B, 256, N, device=dev)
new_xyz = cases.centres_from(xyz.cpu(), M, 11).to(dev)
c1, _, c2, _ = mod.mlp_module
w1, w2 = c1.weight.detach().reshape(256, 256).contiguous(), c2.weight.detach().reshape(256, 256).contiguous()
out, idx, G, H, amax = fused._forward(xyz, new_xyz, feats, 0.3, 16, w1, c1.bias.detach().contiguous(), w2, c2.bias.detach().contiguous(), True)
grouped = _ext.group_points(feats, idx)
print('G', (G - grouped).abs().max().item())
Hr = torch.relu(torch.einsum('oc,bcms->boms', w1, grouped) + c1.bias.view(1, -1, 1, 1))
print('H', (H - Hr).abs().max().item())
Y = torch.relu(torch.einsum('oc,bcms->boms', w2, Hr) + c2.bias.view(1, -1, 1, 1))
mx, am = Y.max(dim=3)
print('out', (out - mx).abs().max().item(), 'amax mismatches', ((amax.long() != am) & (mx > 0)).sum().item())
dout = torch.randn(B, 256, M, device=dev)
dZ2, dZ1, dG = (torch.empty(B, 256, M, 16, device=dev) for _ in range(3))
w2t, w1t = w2.t().contiguous(), w1.t().contiguous()
_lib.check(_lib.lib().p2r_sa_votes_backward(B, M, 16, 256, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(amax), _lib.ptr(H), _lib.ptr(w2t), _lib.ptr(w1t), _lib.ptr(dZ2), _lib.ptr(dZ1), _lib.ptr(dG), _lib.current_stream(dev)), 'bwd')
dZ2r = torch.zeros_like(dZ2); dZ2r.scatter_(3, am.unsqueeze(-1), (dout * (mx > 0)).unsqueeze(-1))
print('dZ2', (dZ2 - dZ2r).abs().max().item())
dHr = torch.einsum('oc,boms->bcms', w2, dZ2r)
dZ1r = dHr * (Hr > 0)
print('dZ1', (dZ1 - dZ1r).abs().max().item(), dZ1r.abs().max().item())
dGr = torch.einsum('oc,boms->bcms', w1, dZ1r)
print('dG', (dG - dGr).abs().max().item(), dGr.abs().max().item())
dw2 = fused._weight_grad(dZ2, H); dw2r = torch.einsum('boms,bcms->oc', dZ2r, Hr)
print('dW2', (dw2 - dw2r).abs().max().item(), dw2r.abs().max().item())
dw1 = fused._weight_grad(dZ1, G); dw1r = torch.einsum('boms,bcms->oc', dZ1r, grouped)
print('dW1', (dw1 - dw1r).abs().max().item(), dw1r.abs().max().item())
