"""Point-wise convolution stacks of the vote / proposal heads on the job-list kernels of csrc/pw_layers.hip.

`vote_head(module, seed_features)` equals `module.conv_input(seed_features.transpose(1, 2))` of CenterVoteModule
(reference models/p2rnet/modules/vote_center.py:28-48); `proposal_heads(net, features, eps)` equals the four conv
stems, the three mixture heads' backbone / pi convolutions with their read-out and `conv_sem_obj` of ProposalNet.forward
(proposal_net.py:183-191, mdn.py:34-83,141-161) -- same parameters (the modules own them, `state_dict` unchanged),
same BatchNorm1d semantics (batch statistics + running-statistics update in training mode, running statistics in
evaluation mode), differentiable w.r.t. every parameter and the input.

A layer is one launch each way and layers of equal depth share a launch (include/p2r_hip.h: p2r_pw_gemm and friends):
the proposal head is 10 launches forward and 15 backward instead of ~115 / ~230 library and ATen launches, the vote head
5 and 9 instead of ~20 / ~60.  The BatchNorm + ReLU between two layers is applied by the consumer while it stages its
input tile; pre-BatchNorm gradients exist only as (masked gradient, saved conv output, three constants per channel).
"""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib
from . import bn_op

_P = ctypes.c_void_p
_I = ctypes.c_int


class _Job(ctypes.Structure):
    _fields_ = [(n, _P) for n in ('x', 'x2', 'tr', 'w', 'bias', 'mz', 'mfin', 'out', 'stats')] + \
               [(n, _I) for n in ('k', 'rows', 'x_ctot', 'x_nlc', 'tr_mode', 'tr_ld', 'w_t', 'out_ctot', 'out_nlc',
                                  'epilogue', 'mz_ctot', 'mfin_ld')]


class _WJob(ctypes.Structure):
    _fields_ = [(n, _P) for n in ('x', 'x2', 'tr', 'y', 'ytr', 'dw_part', 'db_part')] + \
               [(n, _I) for n in ('rows', 'k', 'x_ctot', 'x_nlc', 'tr_mode', 'tr_ld', 'y_ctot', 'y_nlc', 'ytr_ld',
                                  'split')]


class _BnJob(ctypes.Structure):
    _fields_ = [(n, _P) for n in ('part', 'gamma', 'beta', 'running_mean', 'running_var', 'fin',
                                  'num_batches_tracked')] + \
               [('eps', ctypes.c_double), ('momentum', ctypes.c_double)] + [(n, _I) for n in ('P', 'C', 'fin_ld')]


class _BnbJob(ctypes.Structure):
    _fields_ = [(n, _P) for n in ('part', 'fin', 'coef', 'dgamma', 'dbeta')] + [('M', ctypes.c_double)] + \
               [(n, _I) for n in ('P', 'C', 'fin_ld', 'coef_ld', 'train')]


class _RJob(ctypes.Structure):
    _fields_ = [('in_', _P), ('out', _P), ('P', _I), ('M', _I)]


class _MixHead(ctypes.Structure):
    _fields_ = [(n, _P) for n in ('logit', 'log_sigma', 'mu', 'eps', 'dpred', 'pred', 'dmu', 'pi', 'dlogit',
                                  'dlog_sigma')] + [('D', _I), ('f64', _I)]


def _at(t, off=0):
    """device address of element `off` (in elements) of tensor t; None -> NULL"""
    if t is None:
        return None
    return t.data_ptr() + off * t.element_size()


def _launch(fn_name, struct, jobs, *tail):
    arr = (struct * len(jobs))(*[struct(**j) for j in jobs])
    _lib.check(getattr(_lib.lib(), fn_name)(len(jobs), arr, *tail), fn_name)


def _gemm(jobs, B, L, st):
    for j in jobs:
        for k, v in (('x2', None), ('tr', None), ('bias', None), ('mz', None), ('mfin', None), ('stats', None),
                     ('x_nlc', 0), ('tr_mode', 0), ('tr_ld', 0), ('w_t', 0), ('out_nlc', 0), ('epilogue', 0),
                     ('mz_ctot', 0), ('mfin_ld', 0)):
            j.setdefault(k, v)
    _launch('p2r_pw_gemm', _Job, jobs, B, L, st)


def _wgrad(jobs, B, L, st):
    for j in jobs:
        for k, v in (('x2', None), ('tr', None), ('ytr', None), ('db_part', None), ('x_nlc', 0), ('tr_mode', 0),
                     ('tr_ld', 0), ('y_nlc', 0), ('ytr_ld', 0)):
            j.setdefault(k, v)
    _launch('p2r_pw_wgrad', _WJob, jobs, B, L, st)


def _reduce(jobs, st):
    """jobs: list of (partials [P, ...], out) -> out = partials.sum(0), one launch for up to 48 of them."""
    for i in range(0, len(jobs), 48):
        chunk = [dict(in_=_at(p), out=_at(o), P=p.shape[0], M=o.numel()) for p, o in jobs[i:i + 48]]
        _launch('p2r_pw_reduce', _RJob, chunk, st)


def _momentum(bn):
    if bn.momentum is None:      # cumulative moving average: the factor depends on the step counter
        return 1.0 / float(bn.num_batches_tracked + 1)
    return float(bn.momentum)


def _bn_finalize(bns, parts, fin, offs, st):
    """bns: BatchNorm1d modules; parts: statistics partials per module (None: evaluation mode); fin [4, Ctot];
    offs: channel offset of each module inside fin."""
    jobs = []
    for bn, part, off in zip(bns, parts, offs):
        train = part is not None
        jobs.append(dict(part=_at(part), gamma=_at(bn.weight), beta=_at(bn.bias),
                         running_mean=_at(bn.running_mean), running_var=_at(bn.running_var), fin=_at(fin, off),
                         num_batches_tracked=_at(bn.num_batches_tracked) if train else None,
                         eps=float(bn.eps), momentum=_momentum(bn) if train else -1.0,
                         P=part.shape[0] if train else 0, C=bn.num_features, fin_ld=fin.shape[1]))
    _launch('p2r_pw_bn_finalize', _BnJob, jobs, st)


def _bn_bwd_finalize(parts, fin, coef, offs, widths, M, train, st):
    """-> list of (dgamma, dbeta) per job; coef [3, Ctot] filled."""
    jobs, grads = [], []
    for part, off, C in zip(parts, offs, widths):
        dg = torch.empty(C, dtype=torch.float32, device=fin.device)
        db = torch.empty(C, dtype=torch.float32, device=fin.device)
        grads.append((dg, db))
        jobs.append(dict(part=_at(part), fin=_at(fin, off), coef=_at(coef, off), dgamma=_at(dg), dbeta=_at(db),
                         M=float(M), P=part.shape[0], C=C, fin_ld=fin.shape[1], coef_ld=coef.shape[1],
                         train=int(train)))
    _launch('p2r_pw_bn_bwd_finalize', _BnbJob, jobs, st)
    return grads


def _split_for(rows, k, chunks, njobs):
    """split-K factor of a weight-gradient job: ~512 workgroups per launch, at least 2 column chunks each"""
    tiles = ((rows + 63) // 64) * ((k + 63) // 64)
    return max(1, min(chunks // 2 if chunks >= 2 else 1, -(-512 // (tiles * njobs))))


def _w2(conv):
    """Conv1d(k=1) weight as a contiguous [out][in] matrix (a view for contiguous parameters)."""
    return conv.weight.reshape(conv.out_channels, conv.in_channels).contiguous()


def _mix_forward(logits, offs, G, L, mdns, eps):
    """logits (B, Ctot, L) f32 holding head i's G rows from channel offs[i]; eps[i] None: the mixture mean.
    -> [pred_i (B, L, D_i) in mu_i's dtype]; one launch for all heads."""
    B = logits.shape[0]
    heads, preds, keep = [], [], []
    for off, mdn, e in zip(offs, mdns, eps):
        mu, ls = mdn.mu.contiguous(), mdn.log_sigma.contiguous()
        D = mu.shape[1]
        pred = torch.empty((B, L, D), dtype=mu.dtype, device=logits.device)
        keep += [mu, ls]
        preds.append(pred)
        heads.append(dict(logit=_at(logits, off * L), log_sigma=_at(ls), mu=_at(mu), eps=_at(e), pred=_at(pred), D=D,
                          f64=int(mu.dtype == torch.float64)))
    arr = (_MixHead * len(heads))(*[_MixHead(**h) for h in heads])
    _lib.check(_lib.lib().p2r_mdn_mix_forward(len(heads), arr, B, G, L, logits.shape[1],
                                              _lib.current_stream(logits.device)), "mdn_mix_forward")
    return preds


def _mix_backward(logits, dlogits, offs, G, L, mdns, eps, dpreds):
    """-> ([dmu_i], [dlog_sigma_i]); dlogits' rows offs[i] .. offs[i] + G written."""
    B = logits.shape[0]
    heads, dmus, dlss, keep = [], [], [], []
    for off, mdn, e, dp in zip(offs, mdns, eps, dpreds):
        mu, ls = mdn.mu.contiguous(), mdn.log_sigma.contiguous()
        D = mu.shape[1]
        dmu, dls = torch.empty_like(mu), torch.empty((G, D), dtype=torch.float32, device=logits.device)
        keep += [mu, ls]
        dmus.append(dmu)
        dlss.append(dls)
        heads.append(dict(logit=_at(logits, off * L), log_sigma=_at(ls), mu=_at(mu), eps=_at(e), dpred=_at(dp),
                          dmu=_at(dmu), dlogit=_at(dlogits, off * L), dlog_sigma=_at(dls), D=D,
                          f64=int(mu.dtype == torch.float64)))
    arr = (_MixHead * len(heads))(*[_MixHead(**h) for h in heads])
    _lib.check(_lib.lib().p2r_mdn_mix_backward(len(heads), arr, B, G, L, logits.shape[1], dlogits.shape[1],
                                               _lib.current_stream(logits.device)), "mdn_mix_backward")
    return dmus, dlss


# ------------------------------------------------------------------------------------------------------------------
# proposal head
# ------------------------------------------------------------------------------------------------------------------
_HEADS = ('center', 'size', 'heading')


def _proposal_modules(net):
    stems = [net.conv_center, net.conv_size, net.conv_heading, net.conv_sem_obj]
    gmms = [net.gmm_center, net.gmm_size, net.gmm_heading]
    return stems, gmms


def _proposal_params(net):
    stems, gmms = _proposal_modules(net)
    ps = []
    for s in stems:
        ps += [s[0].conv.weight, s[0].batchnorm.weight, s[0].batchnorm.bias,
               s[1].conv.weight, s[1].batchnorm.weight, s[1].batchnorm.bias]
    ps += [net.conv_sem_obj[2].conv.weight, net.conv_sem_obj[2].conv.bias]
    for gm in gmms:
        ps += [gm.backbone.conv.weight, gm.backbone.batchnorm.weight, gm.backbone.batchnorm.bias,
               gm.mdn.pi.conv.weight, gm.mdn.pi.conv.bias, gm.mdn.mu, gm.mdn.log_sigma]
    return ps


def proposal_heads_supported(net, features):
    if not (features.is_cuda and features.dtype == torch.float32 and features.dim() == 3 and features.shape[1] == 256
            and features.shape[2] % 64 == 0 and features.shape[0] > 0):
        return False
    stems, gmms = _proposal_modules(net)
    try:
        for s in stems:
            if not (hasattr(s[0], 'batchnorm') and hasattr(s[1], 'batchnorm') and s[0].conv.out_channels == 128
                    and s[1].conv.out_channels == 128 and s[0].conv.bias is None and s[1].conv.bias is None):
                return False
        if not (len(net.conv_sem_obj) == 3 and net.conv_sem_obj[2].conv.bias is not None
                and net.conv_sem_obj[2].conv.out_channels % 4 == 0):
            return False
        for gm in gmms:
            if gm.hparams.batch_norm_continuous_input or gm.backbone.conv.out_channels != 128 or \
                    gm.mdn.pi.conv.bias is None or gm.mdn.hparams.n_samples != 1 or \
                    gm.mdn.hparams.central_tendency != 'mean' or gm.mdn.mu.shape[1] > 4:
                return False
        G = gmms[0].mdn.mu.shape[0]
        if any(gm.mdn.mu.shape[0] != G for gm in gmms) or (G * features.shape[2]) % 4 != 0:
            return False
        modes = {bn.training for s in stems for bn in (s[0].batchnorm, s[1].batchnorm)} | \
                {gm.backbone.batchnorm.training for gm in gmms}
        return len(modes) == 1
    except (AttributeError, IndexError, TypeError):
        return False


class _ProposalHeads(Function):
    """inputs: features (B,256,K), eps_center, eps_size, eps_heading ((B*K, G, 1, D), or None: mixture means),
    then the 47 parameters in `_proposal_params` order.  outputs: pred_center (B,K,3), pred_size (B,K,3),
    pred_heading (B,K,2) f64, sem_obj (B,K,24) -- all in (B, K, D) memory order (callers transpose the views) -- and
    the mixture logits (B, 3 G, K) (not differentiable: `generate` reads the mixture weights off them)."""

    @staticmethod
    def forward(ctx, net, features, eps_c, eps_s, eps_h, *params):
        stems, gmms = _proposal_modules(net)
        feats = features.contiguous()
        B, _, K = feats.shape
        dev = feats.device
        cols, cb = B * K, B * K // 64
        train = stems[0][0].batchnorm.training
        f32 = dict(dtype=torch.float32, device=dev)
        G = gmms[0].mdn.mu.shape[0]
        nsem = net.conv_sem_obj[2].conv.out_channels
        Z1, Z2 = torch.empty((B, 512, K), **f32), torch.empty((B, 512, K), **f32)
        Z3 = torch.empty((B, 384, K), **f32)
        fin1, fin2, fin3 = torch.empty((4, 512), **f32), torch.empty((4, 512), **f32), torch.empty((4, 384), **f32)
        sem = torch.empty((B, K, nsem), **f32)
        logits = torch.empty((B, 3 * G, K), **f32)
        w = {}
        for j, s in enumerate(stems):
            w['s0', j], w['s1', j] = _w2(s[0].conv), _w2(s[1].conv)
        for j, gm in enumerate(gmms):
            w['bb', j], w['pi', j] = _w2(gm.backbone.conv), _w2(gm.mdn.pi.conv)
        w['sem'] = _w2(net.conv_sem_obj[2].conv)

        def stats(n, C):
            return [torch.empty((cb, C, 3), **f32) if train else None for _ in range(n)]

        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            s1, s2, s3 = stats(4, 128), stats(4, 128), stats(3, 128)
            _gemm([dict(x=_at(feats), w=_at(w['s0', j]), out=_at(Z1, j * 128 * K), stats=_at(s1[j]), k=256, rows=128,
                        x_ctot=256, out_ctot=512) for j in range(4)], B, K, st)
            _bn_finalize([s[0].batchnorm for s in stems], s1, fin1, [128 * j for j in range(4)], st)
            _gemm([dict(x=_at(Z1, j * 128 * K), tr=_at(fin1, 2 * 512 + 128 * j), tr_mode=1, tr_ld=512, w=_at(w['s1', j]),
                        out=_at(Z2, j * 128 * K), stats=_at(s2[j]), k=128, rows=128, x_ctot=512, out_ctot=512)
                   for j in range(4)], B, K, st)
            _bn_finalize([s[1].batchnorm for s in stems], s2, fin2, [128 * j for j in range(4)], st)
            jobs = [dict(x=_at(Z2, j * 128 * K), tr=_at(fin2, 2 * 512 + 128 * j), tr_mode=1, tr_ld=512, w=_at(w['bb', j]),
                         out=_at(Z3, j * 128 * K), stats=_at(s3[j]), k=128, rows=128, x_ctot=512, out_ctot=384)
                    for j in range(3)]
            jobs.append(dict(x=_at(Z2, 3 * 128 * K), tr=_at(fin2, 2 * 512 + 384), tr_mode=1, tr_ld=512, w=_at(w['sem']),
                             bias=_at(net.conv_sem_obj[2].conv.bias), out=_at(sem), k=128, rows=nsem, x_ctot=512,
                             out_ctot=nsem, out_nlc=1))
            _gemm(jobs, B, K, st)
            _bn_finalize([gm.backbone.batchnorm for gm in gmms], s3, fin3, [128 * j for j in range(3)], st)
            _gemm([dict(x=_at(Z3, j * 128 * K), tr=_at(fin3, 2 * 384 + 128 * j), tr_mode=1, tr_ld=384, w=_at(w['pi', j]),
                        bias=_at(gmms[j].mdn.pi.conv.bias), out=_at(logits, j * G * K), k=128, rows=G, x_ctot=384,
                        out_ctot=3 * G) for j in range(3)], B, K, st)
            eps = [e.contiguous() if e is not None else None for e in (eps_c, eps_s, eps_h)]
            preds = _mix_forward(logits, [j * G for j in range(3)], G, K, [gm.mdn for gm in gmms], eps)
        ctx.net, ctx.train, ctx.dims = net, train, (B, K, G, nsem)
        ctx.eps = eps
        ctx.save_for_backward(feats, Z1, Z2, Z3, fin1, fin2, fin3, logits, *params)
        ctx.mark_non_differentiable(logits)
        return preds[0], preds[1], preds[2], sem, logits

    @staticmethod
    def backward(ctx, d_c, d_s, d_h, d_sem, _d_logits=None):
        net, train = ctx.net, ctx.train
        B, K, G, nsem = ctx.dims
        feats, Z1, Z2, Z3, fin1, fin2, fin3, logits = ctx.saved_tensors[:8]
        stems, gmms = _proposal_modules(net)
        dev = feats.device
        cols, cb = B * K, B * K // 64
        f32 = dict(dtype=torch.float32, device=dev)
        w = {}
        for j, s in enumerate(stems):
            w['s0', j], w['s1', j] = _w2(s[0].conv), _w2(s[1].conv)
        for j, gm in enumerate(gmms):
            w['bb', j], w['pi', j] = _w2(gm.backbone.conv), _w2(gm.mdn.pi.conv)
        w['sem'] = _w2(net.conv_sem_obj[2].conv)
        dpred = []
        for j, (d, gm) in enumerate(zip((d_c, d_s, d_h), gmms)):
            D, dt = gm.mdn.mu.shape[1], gm.mdn.mu.dtype
            dpred.append(torch.zeros((B, K, D), dtype=dt, device=dev) if d is None else d.to(dt).contiguous())
        d_sem = torch.zeros((B, K, nsem), **f32) if d_sem is None else d_sem.contiguous()
        dlogits = torch.empty((B, 3 * G, K), **f32)
        g3, g2, g1 = torch.empty((B, 384, K), **f32), torch.empty((B, 512, K), **f32), torch.empty((B, 512, K), **f32)
        coef3, coef2, coef1 = torch.empty((3, 384), **f32), torch.empty((3, 512), **f32), torch.empty((3, 512), **f32)
        need_dx = ctx.needs_input_grad[1]
        red = []            # (partials, out) pairs of the final reduction launch

        def wpart(rows, k, njobs, bias=False):
            sp = _split_for(rows, k, cb, njobs)
            return (torch.empty((sp, rows, k), **f32), torch.empty((sp, rows), **f32) if bias else None, sp)

        def bparts(n, C):
            return [torch.empty((cb, C, 2), **f32) for _ in range(n)]

        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            dmu, dls = _mix_backward(logits, dlogits, [j * G for j in range(3)], G, K, [gm.mdn for gm in gmms],
                                     ctx.eps, dpred)
            # ---- level 4: pi convolutions ----
            b3 = bparts(3, 128)
            _gemm([dict(x=_at(dlogits, j * G * K), w=_at(w['pi', j]), w_t=1, out=_at(g3, j * 128 * K), stats=_at(b3[j]),
                        k=G, rows=128, x_ctot=3 * G, out_ctot=384, epilogue=1, mz=_at(Z3, j * 128 * K), mz_ctot=384,
                        mfin=_at(fin3, 128 * j), mfin_ld=384) for j in range(3)], B, K, st)
            p4 = [wpart(G, 128, 3, bias=True) for _ in range(3)]
            _wgrad([dict(x=_at(dlogits, j * G * K), x_ctot=3 * G, rows=G, y=_at(Z3, j * 128 * K), y_ctot=384,
                         ytr=_at(fin3, 2 * 384 + 128 * j), ytr_ld=384, k=128, dw_part=_at(p4[j][0]),
                         db_part=_at(p4[j][1]), split=p4[j][2]) for j in range(3)], B, K, st)
            gb3 = _bn_bwd_finalize(b3, fin3, coef3, [128 * j for j in range(3)], [128] * 3, cols, train, st)
            # ---- level 3: mixture backbones + the last sem_obj layer ----
            b2 = bparts(4, 128)
            jobs = [dict(x=_at(g3, j * 128 * K), x2=_at(Z3, j * 128 * K), tr=_at(coef3, 128 * j), tr_mode=2, tr_ld=384,
                         w=_at(w['bb', j]), w_t=1, out=_at(g2, j * 128 * K), stats=_at(b2[j]), k=128, rows=128,
                         x_ctot=384, out_ctot=512, epilogue=1, mz=_at(Z2, j * 128 * K), mz_ctot=512,
                         mfin=_at(fin2, 128 * j), mfin_ld=512) for j in range(3)]
            jobs.append(dict(x=_at(d_sem), x_nlc=1, w=_at(w['sem']), w_t=1, out=_at(g2, 3 * 128 * K), stats=_at(b2[3]),
                             k=nsem, rows=128, x_ctot=nsem, out_ctot=512, epilogue=1, mz=_at(Z2, 3 * 128 * K),
                             mz_ctot=512, mfin=_at(fin2, 384), mfin_ld=512))
            _gemm(jobs, B, K, st)
            p3 = [wpart(128, 128, 4) for _ in range(3)] + [wpart(nsem, 128, 4, bias=True)]
            jobs = [dict(x=_at(g3, j * 128 * K), x2=_at(Z3, j * 128 * K), tr=_at(coef3, 128 * j), tr_mode=2, tr_ld=384,
                         x_ctot=384, rows=128, y=_at(Z2, j * 128 * K), y_ctot=512, ytr=_at(fin2, 2 * 512 + 128 * j),
                         ytr_ld=512, k=128, dw_part=_at(p3[j][0]), split=p3[j][2]) for j in range(3)]
            jobs.append(dict(x=_at(d_sem), x_nlc=1, x_ctot=nsem, rows=nsem, y=_at(Z2, 3 * 128 * K), y_ctot=512,
                             ytr=_at(fin2, 2 * 512 + 384), ytr_ld=512, k=128, dw_part=_at(p3[3][0]),
                             db_part=_at(p3[3][1]), split=p3[3][2]))
            _wgrad(jobs, B, K, st)
            gb2 = _bn_bwd_finalize(b2, fin2, coef2, [128 * j for j in range(4)], [128] * 4, cols, train, st)
            # ---- level 2: second stem layers ----
            b1 = bparts(4, 128)
            _gemm([dict(x=_at(g2, j * 128 * K), x2=_at(Z2, j * 128 * K), tr=_at(coef2, 128 * j), tr_mode=2, tr_ld=512,
                        w=_at(w['s1', j]), w_t=1, out=_at(g1, j * 128 * K), stats=_at(b1[j]), k=128, rows=128,
                        x_ctot=512, out_ctot=512, epilogue=1, mz=_at(Z1, j * 128 * K), mz_ctot=512,
                        mfin=_at(fin1, 128 * j), mfin_ld=512) for j in range(4)], B, K, st)
            p2 = [wpart(128, 128, 4) for _ in range(4)]
            _wgrad([dict(x=_at(g2, j * 128 * K), x2=_at(Z2, j * 128 * K), tr=_at(coef2, 128 * j), tr_mode=2, tr_ld=512,
                         x_ctot=512, rows=128, y=_at(Z1, j * 128 * K), y_ctot=512, ytr=_at(fin1, 2 * 512 + 128 * j),
                         ytr_ld=512, k=128, dw_part=_at(p2[j][0]), split=p2[j][2]) for j in range(4)], B, K, st)
            gb1 = _bn_bwd_finalize(b1, fin1, coef1, [128 * j for j in range(4)], [128] * 4, cols, train, st)
            # ---- level 1: first stem layers ----
            dfeat = None
            if need_dx:
                dx4 = torch.empty((4, B, 256, K), **f32)
                _gemm([dict(x=_at(g1, j * 128 * K), x2=_at(Z1, j * 128 * K), tr=_at(coef1, 128 * j), tr_mode=2,
                            tr_ld=512, w=_at(w['s0', j]), w_t=1, out=_at(dx4, j * B * 256 * K), k=128, rows=256,
                            x_ctot=512, out_ctot=256) for j in range(4)], B, K, st)
                dfeat = torch.empty((B, 256, K), **f32)
                red.append((dx4, dfeat))
            p1 = [wpart(128, 256, 4) for _ in range(4)]
            _wgrad([dict(x=_at(g1, j * 128 * K), x2=_at(Z1, j * 128 * K), tr=_at(coef1, 128 * j), tr_mode=2, tr_ld=512,
                         x_ctot=512, rows=128, y=_at(feats), y_ctot=256, k=256, dw_part=_at(p1[j][0]), split=p1[j][2])
                    for j in range(4)], B, K, st)
            # ---- one reduction launch for every partial ----
            def out_w(part, conv):
                o = torch.empty(conv.weight.shape, **f32)
                red.append((part, o))
                return o
            dW = {}
            for j, s in enumerate(stems):
                dW['s0', j], dW['s1', j] = out_w(p1[j][0], s[0].conv), out_w(p2[j][0], s[1].conv)
            for j, gm in enumerate(gmms):
                dW['bb', j], dW['pi', j] = out_w(p3[j][0], gm.backbone.conv), out_w(p4[j][0], gm.mdn.pi.conv)
            dW['sem'] = out_w(p3[3][0], net.conv_sem_obj[2].conv)
            dbias_pi = []
            for j in range(3):
                o = torch.empty((G,), **f32)
                red.append((p4[j][1], o))
                dbias_pi.append(o)
            dbias_sem = torch.empty((nsem,), **f32)
            red.append((p3[3][1], dbias_sem))
            _reduce(red, st)
        grads = []
        for j in range(4):
            grads += [dW['s0', j], gb1[j][0], gb1[j][1], dW['s1', j], gb2[j][0], gb2[j][1]]
        grads += [dW['sem'], dbias_sem]
        for j in range(3):
            grads += [dW['bb', j], gb3[j][0], gb3[j][1], dW['pi', j], dbias_pi[j], dmu[j], dls[j]]
        return (None, dfeat, None, None, None) + tuple(grads)


def _mixture_noise(e, mu, n_rows, name):
    """Caller-supplied noise (the eps dict or `mdn.noise_hook`) as the read-out kernel reads it: (n_rows, G, 1, D)
    values of mu's dtype, contiguous, on mu's device.  The module chain (`eps * sigma + mu`, mdn.py:40-47) broadcasts
    and type-promotes; the kernel takes a raw pointer, so the same is done here (a float32 draw for the float64
    heading head is widened, a broadcastable shape is expanded) and anything else is an error."""
    want = (n_rows, mu.shape[0], 1, mu.shape[1])
    if not torch.is_tensor(e):
        raise TypeError(f"mixture noise for '{name}' must be a tensor, got {type(e).__name__}")
    if tuple(e.shape) != want:
        try:
            e = e.expand(want)
        except RuntimeError:
            raise ValueError(f"mixture noise for '{name}' has shape {tuple(e.shape)}, expected {want}") from None
    return e.to(device=mu.device, dtype=mu.dtype).contiguous()


def proposal_heads(net, features, eps, return_pi=False):
    """-> (pred_center (B,3,K), pred_size (B,3,K), pred_heading (B,2,K) f64, sem_obj_feature (B,2+C,K)) as the module
    chain of ProposalNet.forward returns them (transposed views of (B,K,D) memory, which is the order `decode_scores`
    and the loss want).  eps: dict of explicit mixture noise or None entries (drawn like the reference then);
    eps = False: the mixture means (`generate` with multi_mode off).  return_pi: also the three (B,G,K) mixture
    weights."""
    stems, gmms = _proposal_modules(net)
    n_rows = features.shape[0] * features.shape[2]
    if eps is False:
        draws = [None, None, None]
    else:
        # same draws, same order as the module path: center, size, heading (mdn.py:34-46)
        draws = []
        for name, gm in zip(_HEADS, gmms):
            e = (eps or {}).get(name)
            draws.append(_mixture_noise(e if e is not None else gm.mdn._eps(n_rows, 1), gm.mdn.mu, n_rows, name))
    pc, ps, ph, sem, logits = _ProposalHeads.apply(net, features, draws[0], draws[1], draws[2],
                                                   *_proposal_params(net))
    out = (pc.transpose(1, 2), ps.transpose(1, 2), ph.transpose(1, 2), sem.transpose(1, 2))
    if return_pi:
        G = logits.shape[1] // 3
        out += (tuple(torch.sigmoid(logits[:, j * G:(j + 1) * G]) for j in range(3)),)
    return out


# ------------------------------------------------------------------------------------------------------------------
# vote head
# ------------------------------------------------------------------------------------------------------------------
def vote_head_supported(module, seed_features):
    seq = module.conv_input
    try:
        ok = (seed_features.is_cuda and seed_features.dtype == torch.float32 and seed_features.dim() == 3
              and seed_features.shape[2] == 256 and seed_features.shape[1] % 64 == 0 and seed_features.shape[0] > 0
              and len(seq) == 3 and hasattr(seq[0], 'batchnorm') and hasattr(seq[1], 'batchnorm')
              and not hasattr(seq[2], 'batchnorm') and seq[0].conv.out_channels == 256
              and seq[1].conv.out_channels == 256 and seq[0].conv.bias is None and seq[1].conv.bias is None
              and seq[2].conv.bias is not None and seq[0].batchnorm.training == seq[1].batchnorm.training
              # the backward runs p2r_pw_gemm with k = R (rounded up to 16) against its PW_KMAX = 560 rows of LDS
              and ((seq[2].conv.out_channels + 15) & ~15) <= 560 and seq[2].conv.in_channels == 256)
        return bool(ok)
    except (AttributeError, IndexError, TypeError):
        return False


class _VoteHead(Function):
    """seed_features (B,S,256) -> net (B,S,R) (memory order; the caller transposes the view), R = (3 + 256) * vote_factor."""

    @staticmethod
    def forward(ctx, module, x, w0, g0, b0, w1, g1, b1, w2, bias2):
        seq = module.conv_input
        x = x.contiguous()
        B, S, _ = x.shape
        dev = x.device
        cb = B * S // 64
        R = seq[2].conv.out_channels
        train = seq[0].batchnorm.training
        f32 = dict(dtype=torch.float32, device=dev)
        Z1, Z2 = torch.empty((B, 256, S), **f32), torch.empty((B, 256, S), **f32)
        fin1, fin2 = torch.empty((4, 256), **f32), torch.empty((4, 256), **f32)
        net = torch.empty((B, S, R), **f32)
        W0, W1, W2 = _w2(seq[0].conv), _w2(seq[1].conv), _w2(seq[2].conv)
        s1 = torch.empty((cb, 256, 3), **f32) if train else None
        s2 = torch.empty((cb, 256, 3), **f32) if train else None
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            _gemm([dict(x=_at(x), x_nlc=1, x_ctot=256, w=_at(W0), out=_at(Z1), out_ctot=256, stats=_at(s1), k=256,
                        rows=256)], B, S, st)
            _bn_finalize([seq[0].batchnorm], [s1], fin1, [0], st)
            if bn_op.GATE_HOOK is not None:      # (the next layer evaluates relu(Z1 * scale + shift) on load)
                bn_op.GATE_HOOK(seq[0].batchnorm, Z1, fin1[2], fin1[3], None)
            _gemm([dict(x=_at(Z1), x_ctot=256, tr=_at(fin1, 512), tr_mode=1, tr_ld=256, w=_at(W1), out=_at(Z2),
                        out_ctot=256, stats=_at(s2), k=256, rows=256)], B, S, st)
            _bn_finalize([seq[1].batchnorm], [s2], fin2, [0], st)
            if bn_op.GATE_HOOK is not None:
                bn_op.GATE_HOOK(seq[1].batchnorm, Z2, fin2[2], fin2[3], None)
            _gemm([dict(x=_at(Z2), x_ctot=256, tr=_at(fin2, 512), tr_mode=1, tr_ld=256, w=_at(W2), bias=_at(seq[2].conv.bias),
                        out=_at(net), out_ctot=R, out_nlc=1, k=256, rows=R)], B, S, st)
        ctx.module, ctx.train, ctx.dims = module, train, (B, S, R)
        ctx.save_for_backward(x, Z1, Z2, fin1, fin2, w0, g0, b0, w1, g1, b1, w2, bias2)
        return net

    @staticmethod
    def backward(ctx, dnet):
        module, train = ctx.module, ctx.train
        seq = module.conv_input
        B, S, R = ctx.dims
        x, Z1, Z2, fin1, fin2 = ctx.saved_tensors[:5]
        dev = x.device
        cols, cb = B * S, B * S // 64
        f32 = dict(dtype=torch.float32, device=dev)
        dnet = dnet.contiguous()
        W0, W1, W2 = _w2(seq[0].conv), _w2(seq[1].conv), _w2(seq[2].conv)
        g2, g1 = torch.empty((B, 256, S), **f32), torch.empty((B, 256, S), **f32)
        coef2, coef1 = torch.empty((3, 256), **f32), torch.empty((3, 256), **f32)
        b2, b1 = torch.empty((cb, 256, 2), **f32), torch.empty((cb, 256, 2), **f32)
        sp2, sp1, sp0 = _split_for(R, 256, cb, 1), _split_for(256, 256, cb, 1), _split_for(256, 256, cb, 1)
        pw2, pb2 = torch.empty((sp2, R, 256), **f32), torch.empty((sp2, R), **f32)
        pw1, pw0 = torch.empty((sp1, 256, 256), **f32), torch.empty((sp0, 256, 256), **f32)
        need_dx = ctx.needs_input_grad[1]
        dx = None
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            _gemm([dict(x=_at(dnet), x_nlc=1, x_ctot=R, w=_at(W2), w_t=1, out=_at(g2), out_ctot=256, stats=_at(b2), k=R,
                        rows=256, epilogue=1, mz=_at(Z2), mz_ctot=256, mfin=_at(fin2), mfin_ld=256)], B, S, st)
            _wgrad([dict(x=_at(dnet), x_nlc=1, x_ctot=R, rows=R, y=_at(Z2), y_ctot=256, ytr=_at(fin2, 512), ytr_ld=256,
                         k=256, dw_part=_at(pw2), db_part=_at(pb2), split=sp2)], B, S, st)
            (dg1, db1), = _bn_bwd_finalize([b2], fin2, coef2, [0], [256], cols, train, st)
            _gemm([dict(x=_at(g2), x2=_at(Z2), tr=_at(coef2), tr_mode=2, tr_ld=256, x_ctot=256, w=_at(W1), w_t=1,
                        out=_at(g1), out_ctot=256, stats=_at(b1), k=256, rows=256, epilogue=1, mz=_at(Z1), mz_ctot=256,
                        mfin=_at(fin1), mfin_ld=256)], B, S, st)
            _wgrad([dict(x=_at(g2), x2=_at(Z2), tr=_at(coef2), tr_mode=2, tr_ld=256, x_ctot=256, rows=256, y=_at(Z1),
                         y_ctot=256, ytr=_at(fin1, 512), ytr_ld=256, k=256, dw_part=_at(pw1), split=sp1)], B, S, st)
            (dg0, db0), = _bn_bwd_finalize([b1], fin1, coef1, [0], [256], cols, train, st)
            if need_dx:
                dx = torch.empty((B, S, 256), **f32)
                _gemm([dict(x=_at(g1), x2=_at(Z1), tr=_at(coef1), tr_mode=2, tr_ld=256, x_ctot=256, w=_at(W0), w_t=1,
                            out=_at(dx), out_ctot=256, out_nlc=1, k=256, rows=256)], B, S, st)
            _wgrad([dict(x=_at(g1), x2=_at(Z1), tr=_at(coef1), tr_mode=2, tr_ld=256, x_ctot=256, rows=256, y=_at(x),
                         y_nlc=1, y_ctot=256, k=256, dw_part=_at(pw0), split=sp0)], B, S, st)
            dW0, dW1 = torch.empty(seq[0].conv.weight.shape, **f32), torch.empty(seq[1].conv.weight.shape, **f32)
            dW2, dbias2 = torch.empty(seq[2].conv.weight.shape, **f32), torch.empty((R,), **f32)
            _reduce([(pw0, dW0), (pw1, dW1), (pw2, dW2), (pb2, dbias2)], st)
        return None, dx, dW0, dg0, db0, dW1, dg1, db1, dW2, dbias2


def vote_head(module, seed_features):
    """-> net (B, R, S) = module.conv_input(seed_features.transpose(1, 2)), as a transposed view of (B, S, R) memory."""
    seq = module.conv_input
    net = _VoteHead.apply(module, seed_features, seq[0].conv.weight, seq[0].batchnorm.weight, seq[0].batchnorm.bias,
                          seq[1].conv.weight, seq[1].batchnorm.weight, seq[1].batchnorm.bias, seq[2].conv.weight,
                          seq[2].conv.bias)
    return net.transpose(1, 2)


class _VoteFinish(Function):
    """net (B,S,3+256) memory order, seed_features (B,S,256), hip (B,S,3) -> vote_xyz (B,S,3), vote features
    normalised to unit length, channel-major (B,256,S) (csrc/seed_ops.hip: p2r_vote_finish)."""

    @staticmethod
    def forward(ctx, net, seed_features, hip):
        net, sf, hip = net.contiguous(), seed_features.contiguous(), hip.contiguous()
        B, S, _ = sf.shape
        dev = sf.device
        f32 = dict(dtype=torch.float32, device=dev)
        xyz, feat, inv = torch.empty((B, S, 3), **f32), torch.empty((B, 256, S), **f32), torch.empty((B, S), **f32)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().p2r_vote_finish(B, S, 256, _lib.ptr(net), _lib.ptr(sf), _lib.ptr(hip), _lib.ptr(xyz),
                                                  _lib.ptr(feat), _lib.ptr(inv), _lib.current_stream(dev)), "vote_finish")
        ctx.save_for_backward(feat, inv)
        ctx.set_materialize_grads(False)
        return xyz, feat

    @staticmethod
    def backward(ctx, d_xyz, d_feat):
        feat, inv = ctx.saved_tensors
        B, _, S = feat.shape
        dev = feat.device
        d_net = torch.empty((B, S, 259), dtype=torch.float32, device=dev)
        d_sf = torch.empty((B, S, 256), dtype=torch.float32, device=dev)
        d_xyz = d_xyz.contiguous() if d_xyz is not None else None
        d_feat = d_feat.contiguous() if d_feat is not None else None
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().p2r_vote_finish_grad(B, S, 256, _lib.ptr(d_xyz), _lib.ptr(d_feat), _lib.ptr(feat),
                                                       _lib.ptr(inv), _lib.ptr(d_net), _lib.ptr(d_sf),
                                                       _lib.current_stream(dev)), "vote_finish_grad")
        return d_net, d_sf, d_xyz          # hip: vote_xyz = hip + offset


def votes_normalized(module, seed_xyz, seed_features):
    """CenterVoteModule.forward followed by the feature normalisation of P2RNet (network.py:68-71) on the fused path:
    -> vote_xyz (B,S,3), vote_features (B,S,256) of unit length -- the latter a transposed view of channel-major
    memory, so `features.transpose(1, 2).contiguous()` in front of the vote aggregation is free."""
    net = vote_head(module, seed_features)                     # (B,259,S) view of (B,S,259) memory
    hip = seed_xyz[:, :, module.origin_joint_id]
    xyz, feat = _VoteFinish.apply(net.transpose(1, 2), seed_features, hip)
    return xyz, feat.transpose(1, 2)


def votes_normalized_supported(module, seed_xyz, seed_features):
    return (vote_head_supported(module, seed_features) and module.vote_factor == 1
            and module.conv_input[2].conv.out_channels == 259 and seed_xyz.dim() == 4 and seed_xyz.is_cuda
            and seed_xyz.dtype == torch.float32)
