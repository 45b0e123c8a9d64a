"""Drop-in for BayTorch's MeanFieldVI wrapper (reference: BayTorch/freq_to_bayes.py:7-89,
BayTorch/modules/{module,reparam_layers,conv}.py) on the MI355X HIP library.

    net = MeanFieldVI(get_net(...), prior={'mu': 0.0, 'sigma': np.sqrt(temp) * sigma}, replace_layers='all',
                      device=device, reparam='')
    out = net(net_input); kl = net.kl(); (nll + temp * kl).backward(); optimizer.step()

Same constructor, same `.forward/.kl()/.parameters()/.modules()/.state_dict()` surface and key names.  Every Conv2d of
the wrapped skip() net becomes a Conv2dRT parameter holder (W_mu, W_rho, bias_mu, bias_rho); all arithmetic — the
fused reparameterised convolutions, train-mode BatchNorm, LeakyReLU, bilinear up-sampling, concat, their backward, and
the KL reduction — runs in libmfvi_hip through one compiled layer program.  torch supplies storage and autograd glue.
Anything outside the MFVI runners' configuration raises NotImplementedError instead of silently running elsewhere.
"""
import math

import torch
import torch.nn as nn

from . import _lib as L
from .nets import Concat
from .program import Program


class Conv2dRT(nn.Module):
    """Parameter holder with the attribute surface of BayTorch Conv2dRT/VIModule (modules/conv.py:6-38,
    modules/module.py:9-85): W_mu, W_rho, bias_mu, bias_rho, prior, kwargs, `_kl`."""

    def __init__(self, in_channels, out_channels, kernel_size, bias, stride, padding, dilation, groups, prior, posteriors, kl_type, owner):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = tuple(kernel_size)
        self.kwargs = dict(stride=stride, padding=padding, dilation=dilation, groups=groups)
        self.kl_type = kl_type
        self.prior_mu, self.prior_sigma = float(prior['mu']), float(prior['sigma']) + 1e-6          # module.py:38
        self.posterior_mu_initial, self.posterior_rho_initial = posteriors['mu'], posteriors['rho']
        for name in ('W_mu', 'W_rho', 'bias_mu', 'bias_rho'):          # filled with views of the flat buffer by the owner
            self.register_parameter(name, None)
        self._has_bias = bias
        self.__dict__['_owner'] = owner                                   # not a submodule

    @property
    def _kl(self):
        """KL(prior || posterior) of this layer (modules/module.py:64-74), computed by the HIP reduction."""
        return self.__dict__['_owner']._layer_kl(self)

    def forward(self, x):
        raise RuntimeError("Conv2dRT layers are executed by the owning MeanFieldVI through the fused HIP layer program")


class Conv2dLRT(Conv2dRT):
    """The same parameter holder for the local-reparameterisation layer (BayTorch/modules/conv.py:74-106): sampling happens in
    activation space, parameters and KL are those of Conv2dRT."""


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x, *params):
        out, token = owner._hip_forward(x)
        ctx.owner, ctx.token, ctx.x_requires = owner, token, x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        grads, dz = ctx.owner._hip_backward(ctx.token, dout.contiguous(), ctx.x_requires)
        return (None, dz) + tuple(grads)


class _NetFunctionFlat(torch.autograd.Function):
    """flat_parameters=True: ONE differentiable input, the flat [MU | RHO | BN] buffer, instead of 254 Parameter views — one AccumulateGrad
    node and one gradient tensor per backward instead of 254."""

    @staticmethod
    def forward(ctx, owner, x, flat):
        out, token = owner._hip_forward(x)
        ctx.owner, ctx.token, ctx.x_requires = owner, token, x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        g, dz = ctx.owner._hip_backward(ctx.token, dout.contiguous(), ctx.x_requires, flat=True)
        return None, dz, g


class _KLFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, lo, hi, prior_mu, prior_sigma, *params):
        ctx.args = (owner, lo, hi, prior_mu, prior_sigma)
        return owner._hip_kl(lo, hi, prior_mu, prior_sigma)

    @staticmethod
    def backward(ctx, g):
        owner, lo, hi, pm, ps = ctx.args
        return (None, None, None, None, None) + tuple(owner._hip_kl_backward(lo, hi, pm, ps, g))


class _KLFunctionFlat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, lo, hi, prior_mu, prior_sigma, flat):
        ctx.args = (owner, lo, hi, prior_mu, prior_sigma)
        return owner._hip_kl(lo, hi, prior_mu, prior_sigma)

    @staticmethod
    def backward(ctx, g):
        owner, lo, hi, pm, ps = ctx.args
        return None, None, None, None, None, owner._hip_kl_backward(lo, hi, pm, ps, g, flat=True)


class MeanFieldVI(nn.Module):
    def __init__(self, net, prior=None, posteriors=None, kl_type='reverse', reparam='local', replace_layers='all',
                 device=torch.device('cpu'), seed=None, n_samples=1, autotune=True, flat_parameters=False):
        """Arguments as BayTorch/freq_to_bayes.py:9-16, plus seed / n_samples / autotune and
        flat_parameters (default False: `.parameters()` yields the reference's 254 leaf Parameters W_mu, W_rho, bias_mu, bias_rho, BN
        weight / bias, and autograd delivers a .grad to each — exact torch semantics, ~3 ms of per-Parameter Python / autograd /
        optimizer work per iteration).  True: `.parameters()` yields ONE Parameter, the flat [MU | RHO | BN] buffer all of those are
        views of, so `torch.optim.AdamW(net.parameters(), ...)` (bayesian_optimization.py:1356-1357 unchanged) updates it with one
        fused launch and a backward produces one gradient tensor; state_dict / named_parameters / modules() keep the reference's names,
        but the per-layer Parameters then carry no .grad of their own."""
        super().__init__()
        self._flat_mode = bool(flat_parameters)
        self._autotune = bool(autotune)      # False: keep the built-in tiling heuristics (no ~1 s search on the first forward of a shape)
        # reparam == 'local' (the reference's default): every conv becomes a Conv2dLRT (sampling in activation space); anything else
        # (the runners pass ''): Conv2dRT (sampling in weight space) — freq_to_bayes.py:22-29
        self._lrt = reparam == 'local'
        if replace_layers != 'all':
            raise NotImplementedError("replace_layers=%r: only 'all' (the runners' setting) is built" % (replace_layers,))
        if kl_type != 'reverse':
            raise NotImplementedError("kl_type=%r: only 'reverse' (the default, used by all runners) is built" % (kl_type,))
        prior = {'mu': 0, 'sigma': 0.1} if prior is None else prior                         # modules/module.py:23-30
        posteriors = {'mu': (0, 0.1), 'rho': (-3., 0.1)} if posteriors is None else posteriors
        if 'pi' in prior:
            raise NotImplementedError("scale-mixture priors (MixtureNormal) are not used by the runners and not built")
        device = torch.device(device)
        if device.type != 'cuda':
            raise NotImplementedError("this implementation runs on the GPU only (device=%s); there is no CPU path" % device)
        L.lib()                                              # fail loudly if the HIP library is missing
        self.device = device
        self.net = net
        self.n_samples = int(n_samples)
        self.seed = int(torch.initial_seed() if seed is None else seed) & ((1 << 63) - 1)
        self._step = 0
        self._token = 0
        self._plans = {}
        self._sampling = True
        self._drops = [m for m in net.modules() if isinstance(m, nn.Dropout2d)]
        self._replace(net, prior, posteriors, kl_type)
        self._flatten(posteriors)
        self.net = net.to(device) if False else net          # parameters already live on `device` (flat buffer)
        if self._flat_mode:      # not registered as a module parameter: state_dict / named_parameters keep the reference's key list
            object.__setattr__(self, "_flat_p", nn.Parameter(self._flat))

    def parameters(self, recurse=True):
        if getattr(self, "_flat_mode", False):
            if not self._views_intact():
                self._reflatten()
            if self._flat_p.data_ptr() != self._flat.data_ptr():
                self._flat_p.data = self._flat
            return iter([self._flat_p])
        return super().parameters(recurse)

    # ------------------------------------------------------------------ construction
    def _replace(self, module, prior, posteriors, kl_type):
        """Leaf swap of nn.Conv2d by Conv2dRT, recursing like freq_to_bayes.py:50-89."""
        for key, m in list(module._modules.items()):
            if len(m._modules):
                self._replace(m, prior, posteriors, kl_type)
            elif isinstance(m, nn.Conv2d):
                if m.groups != 1 or m.dilation != (1, 1) or m.padding != (0, 0):
                    raise NotImplementedError("Conv2d with groups/dilation/own padding is outside the skip() family")
                module._modules[key] = (Conv2dLRT if self._lrt else Conv2dRT)(m.in_channels, m.out_channels, m.kernel_size, torch.is_tensor(m.bias), m.stride,
                                                m.padding, m.dilation, m.groups, prior, posteriors, kl_type, self)
            elif isinstance(m, (nn.Linear, nn.Conv3d)):
                raise NotImplementedError("Linear / Conv3d Bayesian layers are not part of the inverse-problem nets")

    def _flatten(self, posteriors):
        """All parameters become views of ONE flat fp32 buffer [MU | RHO | BN] (layout of include/mfvi_hip.h)."""
        self._vi = [m for m in self.net.modules() if isinstance(m, Conv2dRT)]
        self._bn = [m for m in self.net.modules() if isinstance(m, nn.BatchNorm2d)]
        n_vi = 0
        for m in self._vi:
            kh, kw = m.kernel_size
            m._w_off = n_vi; n_vi += m.out_channels * m.in_channels * kh * kw
            m._b_off = -1
            if m._has_bias:
                m._b_off = n_vi; n_vi += m.out_channels
        n_bn = sum(2 * b.num_features for b in self._bn)
        self.n_vi, self.n_bn = n_vi, n_bn
        flat = torch.empty(2 * n_vi + n_bn, dtype=torch.float32, device=self.device)
        self._flat = flat
        mu, rho, bn = flat[:n_vi], flat[n_vi:2 * n_vi], flat[2 * n_vi:]
        mu.normal_(*posteriors['mu']); rho.normal_(*posteriors['rho'])              # modules/module.py:56-62
        for m in self._vi:
            kh, kw = m.kernel_size
            nw = m.out_channels * m.in_channels * kh * kw
            shape = (m.out_channels, m.in_channels, kh, kw)
            m.W_mu = nn.Parameter(mu[m._w_off:m._w_off + nw].view(shape)); m.W_rho = nn.Parameter(rho[m._w_off:m._w_off + nw].view(shape))
            if m._has_bias:
                m.bias_mu = nn.Parameter(mu[m._b_off:m._b_off + m.out_channels]); m.bias_rho = nn.Parameter(rho[m._b_off:m._b_off + m.out_channels])
        off = 0
        for b in self._bn:
            c = b.num_features
            if not b.affine:
                raise NotImplementedError("non-affine BatchNorm")
            bn[off:off + c] = 1.0; bn[off + c:off + 2 * c] = 0.0
            b._bn_off = off
            b.weight = nn.Parameter(bn[off:off + c]); b.bias = nn.Parameter(bn[off + c:off + 2 * c])
            b.to(self.device)                                  # running stats buffers
            b.weight.data = bn[off:off + c]; b.bias.data = bn[off + c:off + 2 * c]
            off += 2 * c
        self._bind_running()
        self._param_list = []
        for m in self._vi:
            self._param_list += [m.W_mu, m.W_rho] + ([m.bias_mu, m.bias_rho] if m._has_bias else [])
        for b in self._bn:
            self._param_list += [b.weight, b.bias]

    def _bind_running(self):
        """running_mean / running_var of every BatchNorm2d become views of ONE flat buffer laid out like the BN block (mean at the gamma
        slots, variance at the beta slots), which mfvi_plan_bn_update_running / mfvi_plan_set_bn_eval take (models/common.py:96-97)."""
        run = torch.zeros(max(self.n_bn, 1), dtype=torch.float32, device=self.device)
        for b in self._bn:
            c, off = b.num_features, b._bn_off
            if b.track_running_stats:
                run[off:off + c] = b.running_mean.to(run); run[off + c:off + 2 * c] = b.running_var.to(run)
                b.running_mean = run[off:off + c]; b.running_var = run[off + c:off + 2 * c]
                b.num_batches_tracked = b.num_batches_tracked.to(self.device)
        self._running = run

    def _running_intact(self):
        base = self._running.data_ptr()
        return all((not b.track_running_stats) or (b.running_mean.data_ptr() == base + 4 * b._bn_off and b.running_var.data_ptr() == base + 4 * (b._bn_off + b.num_features))
                   for b in self._bn)

    def _views_intact(self):
        base = self._flat.data_ptr()
        m = self._vi[-1]
        ok = m.W_mu.data_ptr() == base + 4 * m._w_off and m.W_rho.data_ptr() == base + 4 * (self.n_vi + m._w_off)
        if self._bn:
            b = self._bn[-1]
            ok = ok and b.weight.data_ptr() == base + 4 * (2 * self.n_vi + b._bn_off)
        return ok

    def _reflatten(self):
        """Re-establish the flat layout if someone replaced parameter storage (.to(), .float(), load with assign=True)."""
        new = torch.empty_like(self._flat)
        n_vi = self.n_vi
        for m in self._vi:
            nw = m.W_mu.numel()
            new[m._w_off:m._w_off + nw] = m.W_mu.data.reshape(-1).to(new); new[n_vi + m._w_off:n_vi + m._w_off + nw] = m.W_rho.data.reshape(-1).to(new)
            if m._has_bias:
                c = m.out_channels
                new[m._b_off:m._b_off + c] = m.bias_mu.data.to(new); new[n_vi + m._b_off:n_vi + m._b_off + c] = m.bias_rho.data.to(new)
        for b in self._bn:
            c = b.num_features; o = 2 * n_vi + b._bn_off
            new[o:o + c] = b.weight.data.to(new); new[o + c:o + 2 * c] = b.bias.data.to(new)
        self._flat = new
        for m in self._vi:
            nw = m.W_mu.numel()
            m.W_mu.data = new[m._w_off:m._w_off + nw].view_as(m.W_mu); m.W_rho.data = new[n_vi + m._w_off:n_vi + m._w_off + nw].view_as(m.W_rho)
            if m._has_bias:
                c = m.out_channels
                m.bias_mu.data = new[m._b_off:m._b_off + c]; m.bias_rho.data = new[n_vi + m._b_off:n_vi + m._b_off + c]
        for b in self._bn:
            c = b.num_features; o = 2 * n_vi + b._bn_off
            b.weight.data = new[o:o + c]; b.bias.data = new[o + c:o + 2 * c]

    # ------------------------------------------------------------------ layer program
    def _compile(self, cin, H, W):
        """Symbolic execution of the module tree -> fused-op program (Sequential / Concat / ReflectionPad2d / Conv2dRT /
        BatchNorm2d / LeakyReLU / Upsample in the order models/skip.py composes them)."""
        P = Program()
        vi_iter = iter(self._vi)
        zin = P.tensor(cin, H, W)

        def fail(what):
            raise NotImplementedError("module pattern outside the skip() family: " + what)

        def run(m, v):
            # v = dict(tid, pad, up): pending ReflectionPad2d amount / pending x2 upsample (None or its mode)
            def flush_up(v):
                """A pending Upsample with no Concat after it (skip() with num_channels_skip == 0): materialise it."""
                if v['up']:
                    t = P.tensors[v['tid']]
                    out = P.tensor(t['C'], 2 * t['H'], 2 * t['W'])
                    P.concat_up(None, v['tid'], out, v['up'])
                    v = dict(tid=out, pad=0, up=None)
                return v
            if isinstance(m, nn.Sequential):
                for c in m._modules.values():
                    v = run(c, v)
                return v
            if isinstance(m, Concat):
                kids = list(m._modules.values())
                if m.dim != 1 or len(kids) != 2:
                    fail("Concat with dim != 1 or != 2 branches")
                a, b = run(kids[0], dict(v)), run(kids[1], dict(v))
                if a['pad'] or a['up'] or b['pad'] or not b['up']:
                    fail("Concat expects (skip branch, up-sampled deeper branch)")
                ta, tb = P.tensors[a['tid']], P.tensors[b['tid']]
                # Concat centre-crops to the smaller branch (models/common.py:31-41): in skip() that is the up-sampled branch losing its last row /
                # column at an odd size (crop offset 0); any other size relation is outside the family
                if not (0 <= 2 * tb['H'] - ta['H'] <= 1 and 0 <= 2 * tb['W'] - ta['W'] <= 1):
                    raise NotImplementedError("Concat of %dx%d with the x2 up-sampling of %dx%d: only the crop of the up-sampled branch by one "
                                              "row / column is built" % (ta['H'], ta['W'], tb['H'], tb['W']))
                out = P.tensor(ta['C'] + tb['C'], ta['H'], ta['W'])
                P.concat_up(a['tid'], b['tid'], out, b['up'])
                return dict(tid=out, pad=0, up=None)
            if isinstance(m, nn.ReflectionPad2d):
                p = m.padding
                if len(set(p)) != 1:
                    fail("asymmetric ReflectionPad2d")
                if v['pad'] or v['up']:
                    fail("ReflectionPad2d after a pending pad/upsample")
                return dict(v, pad=p[0])
            if isinstance(m, (Conv2dRT, nn.Conv2d)):
                if m is not next(vi_iter):
                    fail("conv execution order differs from module order")
                k = m.kernel_size[0]
                if m.kernel_size[0] != m.kernel_size[1] or v['pad'] != k // 2 or v['up']:
                    fail("Conv2d must follow ReflectionPad2d(k//2)")
                if k not in (1, 3, 5):
                    fail("kernel size %d (1, 3, 5 are built)" % k)
                stride = m.kwargs['stride'] if isinstance(m, Conv2dRT) else m.stride
                stride = stride[0] if isinstance(stride, (tuple, list)) else stride
                out = P.tensor(m.out_channels, *P.conv_out_hw(v['tid'], k, stride))
                P.conv(v['tid'], out, k, stride, bias=m._has_bias, lrt=self._lrt)
                lay = P.layers[-1]
                if lay['w_off'] != m._w_off or lay['b_off'] != m._b_off:
                    fail("internal: parameter offsets out of sync")
                return dict(tid=out, pad=0, up=None, fresh_conv=True)
            if isinstance(m, nn.Dropout2d):
                # models/common.py:125-131: conv -> Dropout2d -> (BatchNorm): folded into the BatchNorm that follows the conv output
                if not v.get('fresh_conv') or v['pad'] or v['up']:
                    fail("Dropout2d must directly follow a convolution")
                if m.inplace:
                    fail("in-place Dropout2d")
                P.set_dropout(v['tid'], float(m.p))
                return dict(v, fresh_conv=False)
            if isinstance(m, nn.BatchNorm2d):
                v = flush_up(v)
                if v['pad']:
                    fail("BatchNorm2d after a pending pad")
                P.set_bn(v['tid'], act=False, eps=m.eps)
                if P.bns[-1]['off'] != m._bn_off:
                    fail("internal: BatchNorm offsets out of sync")
                return dict(v, fresh_conv=False)
            if isinstance(m, nn.LeakyReLU):
                t = P.tensors[v['tid']]
                if not t['has_bn'] or t['has_act'] or v['pad'] or v['up']:
                    fail("LeakyReLU must directly follow a BatchNorm2d")
                t['has_act'], t['slope'] = 1, float(m.negative_slope)
                return v
            if isinstance(m, nn.Upsample):
                if m.mode not in ('bilinear', 'nearest') or float(m.scale_factor) != 2.0 or m.align_corners:
                    raise NotImplementedError("Upsample(mode=%r, scale=%r): only bilinear / nearest x2 (align_corners=False) are built" % (m.mode, m.scale_factor))
                if v['pad'] or v['up']:
                    fail("Upsample after a pending pad/upsample")
                return dict(v, up=m.mode)
            fail(type(m).__name__)

        res = run(self.net, dict(tid=zin, pad=0, up=None))
        for t in P.tensors:
            if t["drop_p"] > 0 and not t["has_bn"]:
                fail("Dropout2d without a BatchNorm behind it")
        if res['pad'] or res['up'] or P.tensors[res['tid']]['has_bn']:
            fail("the net must end with a convolution")
        if P.n_vi != self.n_vi or P.n_bn != self.n_bn:
            fail("internal: parameter count mismatch")
        return P, zin, res['tid']

    def _plan_for(self, cin, H, W, n):
        key = (cin, H, W, n)
        if key not in self._plans:
            P, zin, zout = self._compile(cin, H, W)
            self._plans[key] = P.compile(zin, zout, n)
        return self._plans[key]

    def _blocks(self):
        if not self._views_intact():
            self._reflatten()
        n = self.n_vi
        return self._flat[:n], self._flat[n:2 * n], self._flat[2 * n:]

    # ------------------------------------------------------------------ nn.Module surface
    def forward(self, x):
        modes = {b.training for b in self._bn}
        if len(modes) > 1:
            raise NotImplementedError("BatchNorm layers in mixed train / eval mode")
        if getattr(self, "_flat_mode", False):
            return _NetFunctionFlat.apply(self, x, next(self.parameters()))
        return _NetFunction.apply(self, x, *self._param_list)

    def set_sampling(self, enabled=True):
        """enabled=False reproduces RTLayer's eval branch (w = mu, b = mu_b: reparam_layers.py:33-35) while BatchNorm keeps
        using batch statistics."""
        self._sampling = bool(enabled)
        return self

    def kl(self):
        """Sum of the layers' KL(prior || posterior) as a tensor of shape [1] (freq_to_bayes.py:43-48)."""
        m = self._vi[0]
        if getattr(self, "_flat_mode", False):
            return _KLFunctionFlat.apply(self, 0, self.n_vi, m.prior_mu, m.prior_sigma, next(self.parameters()))
        vi_params = [p for q in self._vi for p in ([q.W_mu, q.W_rho] + ([q.bias_mu, q.bias_rho] if q._has_bias else []))]
        return _KLFunction.apply(self, 0, self.n_vi, m.prior_mu, m.prior_sigma, *vi_params)

    def _layer_kl(self, m):
        hi = (m._b_off + m.out_channels) if m._has_bias else m._w_off + m.W_mu.numel()
        return self._hip_kl(m._w_off, hi, m.prior_mu, m.prior_sigma)[0]

    # ------------------------------------------------------------------ HIP calls
    def _hip_forward(self, x):
        if x.dim() == 4:
            if x.shape[0] != 1:
                raise NotImplementedError("batch size %d: the deep-image-prior nets run on one image (batch 1); MC samples are "
                                          "requested with n_samples" % x.shape[0])
            x3 = x[0]
        else:
            x3 = x
        x3 = x3.contiguous().float()
        cin, H, W = x3.shape
        plan = self._plan_for(cin, H, W, self.n_samples)
        mu, rho, bn = self._blocks()
        if self._autotune and not getattr(plan, "tuned", False):      # first use of this plan: pick the fastest tiling per layer (one-time)
            plan.autotune(mu, rho, bn, x3, self.n_samples)
        sample = bool(self.training and self._sampling)
        step = self._step
        dropping = any(d.training for d in self._drops)      # nn.Dropout2d is the identity in eval mode
        if self._drops:
            L.check(L.lib().mfvi_plan_set_dropout(plan.handle, int(dropping)))
        if self._bn and not self._running_intact():
            self._bind_running()
        bn_eval = bool(self._bn) and not self._bn[0].training
        if bn_eval and not all(b.track_running_stats for b in self._bn):
            raise NotImplementedError("BatchNorm2d(track_running_stats=False) in eval mode")
        # (a plain `net.eval(); y = net(x)` outside torch.no_grad() is ordinary notebook usage and works forward-only; what is not built — the
        #  backward through eval-mode BatchNorm — is refused where it is asked for: _check_token, called by both wrappers' backward)
        # module.eval(): running statistics instead of batch statistics (forward only; the reference never trains in eval mode)
        L.check(L.lib().mfvi_plan_set_bn_eval(plan.handle, L.ptr(self._running) if bn_eval else None))
        out = plan.forward(mu, rho, bn, x3, self.seed, step, 0, self.n_samples, sample)
        if self._bn and not bn_eval and all(b.track_running_stats for b in self._bn):
            # the running statistics a reference run leaves in its state_dict: one update per batch-1 forward, sample by sample
            mom = {b.momentum for b in self._bn}
            if len(mom) != 1 or None in mom:
                raise NotImplementedError("BatchNorm2d momentum must be one number for all layers (the skip() nets use the default 0.1)")
            L.check(L.lib().mfvi_plan_bn_update_running(plan.handle, L.ptr(plan.workspace), self.n_samples, float(mom.pop()), L.ptr(self._running), L.stream_ptr()))
            torch._foreach_add_([b.num_batches_tracked for b in self._bn], self.n_samples)      # one launch for the ~30 counters
        self._token += 1
        if sample or dropping:
            self._step += 1                  # fresh eps / dropout masks for every call, like randn_like in VIModule.rsample
        return out, (self._token, plan, x3, step, sample, bn_eval)

    def _check_token(self, token):
        """The forward a backward call belongs to must be the latest one (one workspace per plan) and a training-mode one."""
        tok, plan, x3, step, sample, bn_eval = token
        if tok != self._token:
            raise RuntimeError("backward through a MeanFieldVI / FusedNet forward that is no longer the latest one: the activations live in one "
                               "workspace per plan.  Draw K Monte-Carlo samples in ONE call (n_samples=K) instead of K calls.")
        if bn_eval:
            raise NotImplementedError("backward through eval-mode BatchNorm is not built (the reference never trains in eval mode): "
                                      "call .train() before a forward whose gradients are needed")
        return plan, x3, step, sample

    def _hip_backward(self, token, dout, want_dz, flat=False):
        plan, x3, step, sample = self._check_token(token)
        mu, rho, bn = self._blocks()
        g = torch.zeros(2 * self.n_vi + max(self.n_bn, 1), dtype=torch.float32, device=self.device)
        n = self.n_vi
        dz = torch.empty((self.n_samples,) + tuple(x3.shape), device=self.device) if want_dz else None
        plan.backward(mu, rho, bn, x3, self.seed, step, 0, self.n_samples, dout, g[:n], g[n:2 * n], g[2 * n:], sample, dz=dz)
        if flat:
            if want_dz:
                dz = dz.sum(0, keepdim=True) if dz.shape[0] > 1 else dz
            return g[:2 * n + self.n_bn], dz
        grads = []
        for m in self._vi:
            nw = m.W_mu.numel()
            grads += [g[m._w_off:m._w_off + nw].view_as(m.W_mu), g[n + m._w_off:n + m._w_off + nw].view_as(m.W_rho)]
            if m._has_bias:
                c = m.out_channels
                grads += [g[m._b_off:m._b_off + c], g[n + m._b_off:n + m._b_off + c]]
        for b in self._bn:
            c = b.num_features; o = 2 * n + b._bn_off
            grads += [g[o:o + c], g[o + c:o + 2 * c]]
        if want_dz:
            dz = dz.sum(0, keepdim=True) if dz.shape[0] > 1 else dz
        return grads, dz

    def _hip_kl(self, lo, hi, prior_mu, prior_sigma):
        mu, rho, _ = self._blocks()
        acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        L.check(L.lib().mfvi_kl(L.ptr(mu[lo:hi]), L.ptr(rho[lo:hi]), hi - lo, prior_mu, float(torch.tensor(prior_sigma, dtype=torch.float32)),
                                L.ptr(acc), L.stream_ptr()))
        return acc.float()                                   # FloatTensor of shape [1], as the reference returns

    def _hip_kl_backward(self, lo, hi, prior_mu, prior_sigma, gout, flat=False):
        mu, rho, _ = self._blocks()
        n = self.n_vi
        g = torch.zeros(2 * n + (self.n_bn if flat else 0), dtype=torch.float32, device=self.device)
        scale = float(gout.reshape(-1)[0])
        L.check(L.lib().mfvi_kl_backward(L.ptr(mu[lo:hi]), L.ptr(rho[lo:hi]), hi - lo, prior_mu,
                                         float(torch.tensor(prior_sigma, dtype=torch.float32)), scale, L.ptr(g[lo:hi]), L.ptr(g[n + lo:n + hi]),
                                         L.stream_ptr()))
        if flat:
            return g
        grads = []
        for m in self._vi:
            nw = m.W_mu.numel()
            grads += [g[m._w_off:m._w_off + nw].view_as(m.W_mu), g[n + m._w_off:n + m._w_off + nw].view_as(m.W_rho)]
            if m._has_bias:
                c = m.out_channels
                grads += [g[m._b_off:m._b_off + c], g[n + m._b_off:n + m._b_off + c]]
        return grads


class FusedNet(MeanFieldVI):
    """The reference's PLAIN skip() nets — what run_*_dip / run_*_mcd / run_*_sgld optimise (bayesian_optimization.py:1140-1166,
    1526-1567, 1739-1766) — executed by the same HIP layer program: nn.Conv2d weights are used as they are (w = weight, no
    sampling, no rho), nn.Dropout2d layers of the MC-dropout nets are folded into the BatchNorm that follows them.

        net = FusedNet(get_net(input_depth, 'skip', pad, ..., dropout_mode_down='2d', dropout_p_down=p, ...), device=device)
        out = net(net_input); loss = mse(out[:, :1], target); loss.backward(); optimizer.step()

    The reference runs these nets as ordinary torch modules, so this wrapper is the one extra line a caller adds; parameters keep
    the reference's names below the `net.` prefix (net.<path>.weight / .bias and the BatchNorm keys) and are views of one flat buffer."""

    def __init__(self, net, device=torch.device('cuda'), seed=None, n_samples=1, autotune=True):
        nn.Module.__init__(self)
        self._autotune = bool(autotune)
        device = torch.device(device)
        if device.type != 'cuda':
            raise NotImplementedError("this implementation runs on the GPU only (device=%s); there is no CPU path" % device)
        L.lib()
        self.device = device
        self.net = net
        self.n_samples = int(n_samples)
        self.seed = int(torch.initial_seed() if seed is None else seed) & ((1 << 63) - 1)
        self._step = 0
        self._token = 0
        self._plans = {}
        self._sampling = False                                   # w = weight: RTLayer's eval branch of the kernels
        self._lrt = False
        self._drops = [m for m in net.modules() if isinstance(m, nn.Dropout2d)]
        for m in net.modules():
            if isinstance(m, (nn.Linear, nn.Conv3d, nn.Dropout)) and not isinstance(m, nn.Dropout2d):
                raise NotImplementedError("%s is outside the skip() family" % type(m).__name__)
        self._flatten(None)

    def _flatten(self, posteriors):
        """Conv weights / biases become views of the MU block of one flat fp32 buffer [MU | RHO (unused, zero) | BN]; the values
        nn.Conv2d initialised them with (kaiming-uniform) are kept."""
        self._vi = [m for m in self.net.modules() if isinstance(m, nn.Conv2d)]
        self._bn = [m for m in self.net.modules() if isinstance(m, nn.BatchNorm2d)]
        n_vi = 0
        for m in self._vi:
            if m.groups != 1 or m.dilation != (1, 1) or m.padding != (0, 0):
                raise NotImplementedError("Conv2d with groups/dilation/own padding is outside the skip() family")
            kh, kw = m.kernel_size
            m._has_bias = m.bias is not None
            m._w_off = n_vi; n_vi += m.out_channels * m.in_channels * kh * kw
            m._b_off = -1
            if m._has_bias:
                m._b_off = n_vi; n_vi += m.out_channels
        n_bn = sum(2 * b.num_features for b in self._bn)
        self.n_vi, self.n_bn = n_vi, n_bn
        flat = torch.zeros(2 * n_vi + n_bn, dtype=torch.float32, device=self.device)
        self._flat = flat
        self._fill(flat, lambda m: m.weight.data, lambda m: m.bias.data, lambda b: (b.weight.data, b.bias.data))
        self._rebind(flat)
        for b in self._bn:
            if not b.affine:
                raise NotImplementedError("non-affine BatchNorm")
            b.to(self.device)
            self._rebind_bn(flat, b)
        self._bind_running()
        self._param_list = []
        for m in self._vi:
            self._param_list += [m.weight] + ([m.bias] if m._has_bias else [])
        for b in self._bn:
            self._param_list += [b.weight, b.bias]

    def _fill(self, flat, w_of, b_of, bn_of):
        off = 0
        for m in self._vi:
            nw = m.weight.numel()
            flat[m._w_off:m._w_off + nw] = w_of(m).reshape(-1).to(flat)
            if m._has_bias:
                flat[m._b_off:m._b_off + m.out_channels] = b_of(m).to(flat)
        for b in self._bn:
            c = b.num_features; o = 2 * self.n_vi + off
            g, be = bn_of(b)
            flat[o:o + c] = g.to(flat); flat[o + c:o + 2 * c] = be.to(flat)
            b._bn_off = off
            off += 2 * c

    def _rebind(self, flat):
        for m in self._vi:
            nw = m.weight.numel()
            w = flat[m._w_off:m._w_off + nw].view(m.out_channels, m.in_channels, *m.kernel_size)
            if isinstance(m.weight, nn.Parameter) and m.weight.device == flat.device:
                m.weight.data = w
            else:
                m.weight = nn.Parameter(w)
            if m._has_bias:
                bv = flat[m._b_off:m._b_off + m.out_channels]
                if isinstance(m.bias, nn.Parameter) and m.bias.device == flat.device:
                    m.bias.data = bv
                else:
                    m.bias = nn.Parameter(bv)

    def _rebind_bn(self, flat, b):
        c = b.num_features; o = 2 * self.n_vi + b._bn_off
        b.weight.data = flat[o:o + c]; b.bias.data = flat[o + c:o + 2 * c]

    def _views_intact(self):
        base = self._flat.data_ptr()
        m = self._vi[-1]
        ok = m.weight.data_ptr() == base + 4 * m._w_off
        if self._bn:
            b = self._bn[-1]
            ok = ok and b.weight.data_ptr() == base + 4 * (2 * self.n_vi + b._bn_off)
        return ok

    def _reflatten(self):
        new = torch.zeros_like(self._flat)
        self._fill(new, lambda m: m.weight.data, lambda m: m.bias.data, lambda b: (b.weight.data, b.bias.data))
        self._flat = new
        self._rebind(new)
        for b in self._bn:
            self._rebind_bn(new, b)

    def kl(self):
        raise AttributeError("FusedNet wraps a deterministic net: there is no KL term (use MeanFieldVI for the Bayesian path)")

    def set_sampling(self, enabled=True):
        if enabled:
            raise ValueError("a deterministic net has nothing to sample")
        return self

    def _hip_backward(self, token, dout, want_dz):
        plan, x3, step, sample = self._check_token(token)
        mu, rho, bn = self._blocks()
        g = torch.zeros(2 * self.n_vi + max(self.n_bn, 1), dtype=torch.float32, device=self.device)
        n = self.n_vi
        dz = torch.empty((self.n_samples,) + tuple(x3.shape), device=self.device) if want_dz else None
        plan.backward(mu, rho, bn, x3, self.seed, step, 0, self.n_samples, dout, g[:n], g[n:2 * n], g[2 * n:], False, dz=dz)
        grads = []
        for m in self._vi:
            nw = m.weight.numel()
            grads.append(g[m._w_off:m._w_off + nw].view_as(m.weight))
            if m._has_bias:
                grads.append(g[m._b_off:m._b_off + m.out_channels])
        for b in self._bn:
            c = b.num_features; o = 2 * n + b._bn_off
            grads += [g[o:o + c], g[o + c:o + 2 * c]]
        if want_dz:
            dz = dz.sum(0, keepdim=True) if dz.shape[0] > 1 else dz
        return grads, dz


class _GaussianNLL(torch.autograd.Function):
    """Forward and backward of the two NLLs on the HIP kernels (mfvi_gaussian_nll_tensors[_backward]); the upstream gradient stays on
    the device."""

    @staticmethod
    def forward(ctx, mu, neg_logvar, target, mask, mean):
        mu_c, s_c = mu.contiguous().float(), neg_logvar.contiguous().float()
        C, hw = (mu_c.shape[1], mu_c[0, 0].numel()) if mu_c.dim() >= 3 else (1, mu_c.numel())
        if mu_c.dim() >= 3 and mu_c.shape[0] != 1:
            raise NotImplementedError("batch size %d: the deep-image-prior losses act on one image" % mu_c.shape[0])
        t_c = target.expand_as(mu_c).contiguous().float()
        Cs = s_c.numel() // hw
        if s_c.numel() != Cs * hw or Cs not in (1, C):
            raise NotImplementedError("neg_logvar of shape %s does not broadcast over mu of shape %s along the channel axis" % (tuple(s_c.shape), tuple(mu_c.shape)))
        m_c, Cm = None, 1
        if mask is not None:
            m_c = mask.contiguous().float(); Cm = m_c.numel() // hw
            if m_c.numel() != Cm * hw or Cm not in (1, C):
                raise NotImplementedError("mask of shape %s does not broadcast over mu of shape %s along the channel axis" % (tuple(m_c.shape), tuple(mu_c.shape)))
        acc = torch.empty(1, dtype=torch.float64, device=mu_c.device)
        L.check(L.lib().mfvi_gaussian_nll_tensors(L.ptr(mu_c), L.ptr(s_c), L.ptr(t_c), L.ptr(m_c), C, Cs, Cm, hw, int(mean), L.ptr(acc), L.stream_ptr()))
        ctx.save_for_backward(mu_c, s_c, t_c, m_c)
        ctx.dims = (C, Cs, Cm, hw, int(mean), mu.shape, neg_logvar.shape)
        return acc.float()[0]

    @staticmethod
    def backward(ctx, g):
        mu_c, s_c, t_c, m_c = ctx.saved_tensors
        C, Cs, Cm, hw, mean, mu_shape, s_shape = ctx.dims
        dmu, ds = torch.empty_like(mu_c), torch.empty_like(s_c)
        L.check(L.lib().mfvi_gaussian_nll_tensors_backward(L.ptr(mu_c), L.ptr(s_c), L.ptr(t_c), L.ptr(m_c), C, Cs, Cm, hw, mean,
                                                           L.ptr(g.reshape(1).contiguous().float()), L.ptr(dmu), L.ptr(ds), L.stream_ptr()))
        return dmu.view(mu_shape), ds.view(s_shape), None, None, None


def _reduction(reduction):
    if reduction not in ('mean', 'sum'):
        raise ValueError("reduction=%r: 'mean' or 'sum'" % (reduction,))
    return reduction == 'mean'


def gaussian_nll(mu, neg_logvar, target, reduction='mean'):
    """Drop-in for utils/bayesian_utils.py:29-32 on the HIP kernels: same arguments, a 0-dim differentiable tensor back."""
    return _GaussianNLL.apply(mu, neg_logvar, target, None, _reduction(reduction))


def gaussian_nll_inpainting(mu, neg_logvar, target, mask, reduction='mean'):
    """Drop-in for utils/bayesian_utils.py:35-39 (masked; neg_logvar broadcasts over the colour channels) on the HIP kernels."""
    return _GaussianNLL.apply(mu, neg_logvar, target, mask, _reduction(reduction))
