"""The embedding MLPs of the ST-GCN backbone (reference models/p2rnet/modules/stgcn.py:46-63: `pos_embed`, `sk_feat` =
SingleConv 'cbr' (3 -> 64), 'cbr' (64 -> 64), 'c' (64 -> 64) on (B, 3, L) point lists) as ONE autograd function.

Forward: the kernels of tconv_op (streaming 3 -> 64 layer whose BatchNorm statistics come from the input moments,
BatchNorm + ReLU folded into the following pointwise convolution, optional broadcast addend in the last layer).

Backward: one pass per layer (csrc/embed_bwd.hip).  The gradient of a layer's conv output never exists in HBM: a layer
hands its predecessor the masked gradient g of the activation plus the two BatchNorm-backward sums, and the predecessor
forms a*g + b*z + c from g, its saved output z and three constants per channel while it stages its tiles.  Per 64 -> 64
layer that is 1.8 GB of traffic at bs=32, T=1024 instead of the 3.1 GB of the three-pass form (data gradient,
BatchNorm-backward apply, weight gradient), and 8 launches per MLP instead of ~25.
"""
import torch
from torch.autograd import Function

from .. import _lib
from . import bn_op, pw_op, tconv_op

_N_BLOCKS = 512
USE_FUSED = True      # tests switch it off to reach the layer-by-layer functions of tconv_op


def _pointwise64(conv):
    return (isinstance(conv, torch.nn.Conv1d) and conv.in_channels == 64 and conv.out_channels == 64
            and tuple(conv.kernel_size) == (1,) and tuple(conv.stride) == (1,) and tuple(conv.padding) == (0,)
            and conv.groups == 1)


def supported(seq, x, inner, add_ct):
    if not (USE_FUSED and len(seq) == 3 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3):
        return False
    s0, s1, s2 = seq
    L = x.shape[2]
    bns = [getattr(s0, 'batchnorm', None), getattr(s1, 'batchnorm', None)]
    return (all(b is not None and b.affine and b.track_running_stats for b in bns) and not hasattr(s2, 'batchnorm')
            and tconv_op.supported_embed3(x, s0.conv) and _pointwise64(s1.conv) and _pointwise64(s2.conv)
            and L % 64 == 0 and L % inner == 0 and inner <= 64 and x.shape[0] > 0
            and bns[0].training == bns[1].training
            and (add_ct is None or (add_ct.is_cuda and add_ct.dtype == torch.float32)))


class _EmbedMLP(Function):
    @staticmethod
    def forward(ctx, x, add_ct, w0, b0, g0, be0, w1, b1, g1, be1, w2, b2, seq, inner):
        s0, s1, s2 = seq
        x = x.contiguous()
        B, _, L = x.shape
        rows = L // inner
        train = s0.batchnorm.training
        dev = x.device

        def fin_of(bn, part):
            if train:
                return bn_op.finalize(part, B * L, bn)          # also updates the running statistics
            invstd = torch.rsqrt(bn.running_var + bn.eps)
            scale = bn.weight * invstd
            return torch.stack([bn.running_mean, invstd, scale, bn.bias - bn.running_mean * scale]).detach().contiguous()

        if train:
            z1, st0 = tconv_op.embed3(x, s0.conv, want_stats=True)
        else:
            z1, st0 = tconv_op.embed3(x, s0.conv), None
        z1 = z1.detach()
        fin1 = fin_of(s0.batchnorm, st0)
        W1 = w1.reshape(64, 64, 1).permute(2, 0, 1).contiguous()
        W2 = w2.reshape(64, 64, 1).permute(2, 0, 1).contiguous()
        z1v = z1.view(B, 64, rows, inner)
        if train:
            z2, st1 = tconv_op._tconv(z1v, fin1[2], fin1[3], W1, b1.contiguous() if b1 is not None else None, True)
        else:
            z2, st1 = tconv_op._tconv(z1v, fin1[2], fin1[3], W1, b1.contiguous() if b1 is not None else None), None
        fin2 = fin_of(s1.batchnorm, st1)
        bias2 = b2.contiguous() if b2 is not None else None
        out = None
        if add_ct is not None:
            add_c = add_ct.contiguous()
            if (tconv_op.USE_GEN3 and inner == 53 and rows % 16 == 0 and add_c.shape == (B, 64, rows)
                    and z2.data_ptr() % 16 == 0):
                out = torch.empty_like(z2)
                with torch.cuda.device(dev):
                    _lib.check(_lib.lib().p2r_stgcn_tconv3_forward_add(
                        B, rows, inner, _lib.ptr(z2), _lib.ptr(fin2[2]), _lib.ptr(fin2[3]),
                        _lib.ptr(tconv_op._permute_taps(W2)), _lib.ptr(bias2), _lib.ptr(add_c), _lib.ptr(out),
                        _lib.current_stream(dev)), "stgcn_tconv3_forward_add")
        if out is None:
            out = tconv_op._tconv(z2, fin2[2], fin2[3], W2, bias2)
            if add_ct is not None:
                out = out + add_ct.unsqueeze(-1)
        ctx.save_for_backward(x, z1, z2.view(B, 64, L), fin1, fin2, w0, w1, w2)
        ctx.seq, ctx.inner, ctx.train = seq, inner, train
        ctx.has = (b0 is not None, b1 is not None, b2 is not None, add_ct is not None)
        return out.view(B, 64, L)

    @staticmethod
    def backward(ctx, dout):
        x, z1, z2, fin1, fin2, w0, w1, w2 = ctx.saved_tensors
        has_b0, has_b1, has_b2, has_add = ctx.has
        assert not ctx.needs_input_grad[0], "embedding MLP: the point coordinates are inputs, no data gradient"
        B, _, L = x.shape
        inner, train = ctx.inner, ctx.train
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        du = dout.contiguous()
        P = _N_BLOCKS
        lib = _lib.lib()
        g2, g1 = torch.empty((B, 64, L), **f32), torch.empty((B, 64, L), **f32)
        sum2, sum1 = torch.empty((P, 64, 2), **f32), torch.empty((P, 64, 2), **f32)
        dw2p, dw1p = torch.empty((P, 64, 64), **f32), torch.empty((P, 64, 64), **f32)
        db2p = torch.empty((P, 64), **f32) if has_b2 else None
        db1p = torch.empty((P, 64), **f32) if has_b1 else None
        coef2, coef1 = torch.empty((3, 64), **f32), torch.empty((3, 64), **f32)
        w0p = torch.empty((B, 64 * 4), **f32)
        W1m, W2m = w1.reshape(64, 64).contiguous(), w2.reshape(64, 64).contiguous()
        d_add = None
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            if has_add and ctx.needs_input_grad[1]:
                from . import seed_op
                d_add = seed_op._rowsum(du.view(B, 64, L // inner, inner), inner, 1.0)
            _lib.check(lib.p2r_embed_layer_backward(B, L, _lib.ptr(du), None, None, _lib.ptr(z2), _lib.ptr(fin2),
                                                    _lib.ptr(W2m), _lib.ptr(g2), _lib.ptr(sum2), P, _lib.ptr(dw2p),
                                                    _lib.ptr(db2p), st), "embed_layer_backward")
            (dg1, dbe1), = pw_op._bn_bwd_finalize([sum2], fin2, coef2, [0], [64], B * L, train, st)
            _lib.check(lib.p2r_embed_layer_backward(B, L, _lib.ptr(g2), _lib.ptr(z2), _lib.ptr(coef2), _lib.ptr(z1),
                                                    _lib.ptr(fin1), _lib.ptr(W1m), _lib.ptr(g1), _lib.ptr(sum1), P,
                                                    _lib.ptr(dw1p), _lib.ptr(db1p), st), "embed_layer_backward")
            (dg0, dbe0), = pw_op._bn_bwd_finalize([sum1], fin1, coef1, [0], [64], B * L, train, st)
            _lib.check(lib.p2r_embed3_weight_grad_lazy(B, L, _lib.ptr(x), _lib.ptr(g1), _lib.ptr(z1), _lib.ptr(coef1),
                                                       _lib.ptr(w0p), st), "embed3_weight_grad_lazy")
            dW2, dW1 = torch.empty(w2.shape, **f32), torch.empty(w1.shape, **f32)
            w0g = torch.empty((64, 4), **f32)
            red = [(dw2p, dW2), (dw1p, dW1), (w0p, w0g)]
            db2 = db1 = None
            if has_b2:
                db2 = torch.empty((64,), **f32)
                red.append((db2p, db2))
            if has_b1:
                db1 = torch.empty((64,), **f32)
                red.append((db1p, db1))
            pw_op._reduce(red, st)
        dW0 = w0g[:, :3].reshape(w0.shape).contiguous()
        db0 = w0g[:, 3].contiguous() if has_b0 else None
        return None, d_add, dW0, db0, dg0, dbe0, dW1, db1, dg1, dbe1, dW2, db2, None, None


def embed_mlp(seq, x, inner, add_ct=None):
    """seq = `_point_mlp` stack; x (B,3,L) -> (B,64,L) (+ add_ct (B,64,L/inner) broadcast over `inner`)."""
    s0, s1, s2 = seq
    return _EmbedMLP.apply(x, add_ct, s0.conv.weight, s0.conv.bias, s0.batchnorm.weight, s0.batchnorm.bias,
                           s1.conv.weight, s1.conv.bias, s1.batchnorm.weight, s1.batchnorm.bias,
                           s2.conv.weight, s2.conv.bias, seq, inner)
