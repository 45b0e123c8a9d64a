"""Layer program = the fused-op description of a skip()-family network that libmfvi_hip executes.

`Program` is built by walking a module tree (bayes.py) or by hand (tests); `Plan` owns the compiled
plan handle plus its device workspace and exposes forward/backward on torch CUDA tensors.
"""
import ctypes as C

from . import _lib as L


class Program:
    def __init__(self):
        self.tensors = []          # TensorDesc fields as dicts
        self.ops = []
        self.n_vi = 0
        self.n_bn = 0
        self.layers = []           # per VI layer: dict(cin, cout, k, stride, w_off, b_off)
        self.bns = []              # per BN: dict(C, off, tensor)

    def tensor(self, C_, H, W, bn=False, act=False, slope=0.2, eps=1e-5):
        off = -1
        if bn:
            off = self.n_bn
            self.n_bn += 2 * C_
            self.bns.append(dict(C=C_, off=off, tensor=len(self.tensors)))
        self.tensors.append(dict(C=C_, H=H, W=W, has_bn=int(bn), has_act=int(act), slope=slope, eps=eps, drop_p=0.0, bn_off=off))
        return len(self.tensors) - 1

    def set_bn(self, tid, act, slope=0.2, eps=1e-5):
        """Attach the BatchNorm (and optional LeakyReLU) that FOLLOWS tensor `tid` in the module order."""
        t = self.tensors[tid]
        if t["has_bn"]:
            raise ValueError("tensor %d already has a BatchNorm" % tid)
        if act and not 0.0 <= slope <= 1.0:          # the kernels form LeakyReLU as max(v, slope * v) (mfvi_plan_create checks it too)
            raise ValueError("LeakyReLU slope %r outside [0, 1]" % (slope,))
        t.update(has_bn=1, has_act=int(act), slope=slope, eps=eps, bn_off=self.n_bn)
        self.bns.append(dict(C=t["C"], off=self.n_bn, tensor=tid))
        self.n_bn += 2 * t["C"]

    def set_dropout(self, tid, p):
        """nn.Dropout2d(p) directly after the conv that produces tensor `tid` (before its BatchNorm)."""
        if not 0.0 <= p < 1.0:
            raise ValueError("dropout probability %r outside [0, 1)" % (p,))
        self.tensors[tid]["drop_p"] = float(p)

    def conv(self, in_id, out_id, ksize, stride=1, bias=True, lrt=False):
        """Conv2dRT (default) or, with lrt=True, Conv2dLRT (local reparameterisation: BayTorch/modules/reparam_layers.py:39-72)."""
        cin, cout = self.tensors[in_id]["C"], self.tensors[out_id]["C"]
        w_off = self.n_vi
        self.n_vi += cout * cin * ksize * ksize
        b_off = -1
        if bias:
            b_off = self.n_vi
            self.n_vi += cout
        lid = len(self.layers)
        self.layers.append(dict(cin=cin, cout=cout, k=ksize, stride=stride, w_off=w_off, b_off=b_off, lrt=bool(lrt)))
        self.ops.append(dict(type=L.OP_CONV_LRT if lrt else L.OP_CONV, in0=in_id, in1=-1, out=out_id, ksize=ksize, stride=stride, layer_id=lid,
                             up_mode=0, w_off=w_off, b_off=b_off))
        return lid

    def concat_up(self, in0, in1, out_id, mode="bilinear"):
        """Concat(in0, Upsample(x2, mode)(in1)); in0=None is a plain upsample (skip() with num_channels_skip == 0)."""
        self.ops.append(dict(type=L.OP_CONCAT_UP, in0=-1 if in0 is None else in0, in1=in1, out=out_id, ksize=0, stride=0,
                             layer_id=0, up_mode={"bilinear": 0, "nearest": 1}[mode], w_off=0, b_off=-1))

    def conv_out_hw(self, in_id, ksize, stride):
        t = self.tensors[in_id]
        p = ksize // 2
        return (t["H"] + 2 * p - ksize) // stride + 1, (t["W"] + 2 * p - ksize) // stride + 1

    def compile(self, input_id, output_id, max_samples, param_dtype="f32"):
        return Plan(self, input_id, output_id, max_samples, param_dtype)


class Plan:
    """Compiled program + workspace on the current CUDA device."""

    def __init__(self, prog, input_id, output_id, max_samples, param_dtype="f32"):
        import torch
        self.prog, self.input_id, self.output_id, self.max_samples = prog, input_id, output_id, max_samples
        if param_dtype not in ("f32", "bf16"):
            raise ValueError("param_dtype %r: 'f32' or 'bf16'" % (param_dtype,))
        self.param_dtype = param_dtype
        td = (L.TensorDesc * len(prog.tensors))(*[L.TensorDesc(**t) for t in prog.tensors])
        od = (L.OpDesc * len(prog.ops))(*[L.OpDesc(**o) for o in prog.ops])
        h = C.c_void_p()
        L.check(L.lib().mfvi_plan_create(td, len(prog.tensors), od, len(prog.ops), input_id, output_id, prog.n_vi, prog.n_bn,
                                         max_samples, C.byref(h)))
        self.handle = h
        if param_dtype == "bf16":       # mu / rho handed to forward / backward are torch.bfloat16 tensors
            L.check(L.lib().mfvi_plan_set_param_dtype(h, L.PARAM_BF16))
        self.workspace_bytes = L.lib().mfvi_plan_workspace_bytes(h)
        self.workspace = torch.empty(max(self.workspace_bytes, 16), dtype=torch.uint8, device="cuda")
        self.in_shape = tuple(prog.tensors[input_id][k] for k in ("C", "H", "W"))
        self.out_shape = tuple(prog.tensors[output_id][k] for k in ("C", "H", "W"))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                L.lib().mfvi_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward(self, mu, rho, bn, z, seed, step, k0, n_samples, sample_weights=True, out=None):
        import torch
        if out is None:
            out = torch.empty((n_samples,) + self.out_shape, dtype=torch.float32, device="cuda")
        assert z.is_contiguous() and z.dtype == torch.float32 and z.numel() == self.in_shape[0] * self.in_shape[1] * self.in_shape[2]
        self._check_params(mu, rho)
        L.check(L.lib().mfvi_forward(self.handle, L.ptr(mu), L.ptr(rho), L.ptr(bn), L.ptr(z), seed, step, k0, n_samples,
                                     int(bool(sample_weights)), L.ptr(self.workspace), L.ptr(out), L.stream_ptr()))
        return out

    def _check_params(self, mu, rho):
        import torch
        want = torch.bfloat16 if self.param_dtype == "bf16" else torch.float32
        if mu.dtype != want or rho.dtype != want:
            raise TypeError("this plan takes %s mu / rho, got %s / %s" % (want, mu.dtype, rho.dtype))

    def backward(self, mu, rho, bn, z, seed, step, k0, n_samples, dout, dmu, drho, dbn, sample_weights=True, dz=None):
        self._check_params(mu, rho)
        assert dout.is_contiguous() and dout.numel() == n_samples * self.out_shape[0] * self.out_shape[1] * self.out_shape[2]
        L.check(L.lib().mfvi_backward(self.handle, L.ptr(mu), L.ptr(rho), L.ptr(bn), L.ptr(z), seed, step, k0, n_samples,
                                      int(bool(sample_weights)), L.ptr(self.workspace), L.ptr(dout), L.ptr(dmu), L.ptr(drho),
                                      L.ptr(dbn), L.ptr(dz), L.stream_ptr()))

    def autotune(self, mu, rho, bn, z, n_samples=None, cache=None):
        """Time the valid MFMA tilings of every conv op on this device and keep the fastest (results unchanged).
        cache: optional JSON path — tilings found earlier for the same program / sample count are reused (and a fresh
        search is written there), e.g. to keep a profiled run free of the search's launches."""
        import json, os, torch
        n = n_samples or self.max_samples
        T = self.prog.tensors
        sig = "%d:%s" % (n, ";".join("%d,%d,%d,%d>%d,%dx%d" % (o["type"], o["ksize"], o["stride"], T[o["in1"] if o["in0"] < 0 else o["in0"]]["C"],
                                                               T[o["out"]]["C"], T[o["out"]]["H"], T[o["out"]]["W"]) for o in self.prog.ops))
        cache = cache or os.environ.get("MFVI_TUNE_CACHE")
        if cache and os.path.exists(cache):
            try:
                saved = json.load(open(cache))
                if saved.get("sig") == sig:
                    for i, tri in saved["tunes"].items():
                        for w, v in enumerate(tri):
                            L.check(L.lib().mfvi_plan_set_tune(self.handle, int(i), w, int(v)))
                    self.tuned = True
                    return
            except (ValueError, KeyError, OSError):
                pass
        out_scratch = torch.empty(2 * n * self.out_shape[0] * self.out_shape[1] * self.out_shape[2], dtype=torch.float32, device="cuda")
        grad_scratch = torch.empty(2 * self.prog.n_vi + self.prog.n_bn, dtype=torch.float32, device="cuda")
        L.check(L.lib().mfvi_plan_autotune(self.handle, L.ptr(mu), L.ptr(rho), L.ptr(bn), L.ptr(z), n, L.ptr(self.workspace),
                                           L.ptr(out_scratch), L.ptr(grad_scratch), L.stream_ptr()))
        self.tuned = True
        if cache:
            lib = L.lib()
            raw = {str(i): [max(lib.mfvi_plan_get_tune(self.handle, i, w), 0) for w in range(3)]
                   for i, o in enumerate(self.prog.ops) if o["type"] == L.OP_CONV}
            try:
                json.dump({"sig": sig, "tunes": raw}, open(cache, "w"))
            except OSError:
                pass

    def tunes(self):
        """-> {op index: (forward, backward-data, backward-weight tiling)} as triples; None = heuristic."""
        dec = lambda v: None if v <= 0 else (v & 255, (v >> 8) & 255, (v >> 16) & 255)
        lib = L.lib()
        return {i: tuple(dec(lib.mfvi_plan_get_tune(self.handle, i, w)) for w in range(3))
                for i, o in enumerate(self.prog.ops) if o["type"] == L.OP_CONV}

    def side_stream(self, enabled):
        """Backward-weight kernels on the plan's side stream (default) or on the caller's stream."""
        L.check(L.lib().mfvi_plan_set_side_stream(self.handle, int(bool(enabled))))

    def grad_split(self, first_op, stream=None):
        """mfvi_plan_set_grad_split: the weight gradients of the ops >= first_op are reduced on `stream` (a torch.cuda.Stream) as soon as
        the backward pass has enqueued their kernels; first_op < 0 removes the split."""
        L.check(L.lib().mfvi_plan_set_grad_split(self.handle, int(first_op), None if first_op < 0 else C.c_void_p(stream.cuda_stream)))

    def grad_split_offset(self, first_op):
        """First parameter index of the flat-layout tail owned by the ops >= first_op."""
        off = C.c_int64(0)
        L.check(L.lib().mfvi_plan_grad_split_offset(self.handle, int(first_op), C.byref(off)))
        return off.value

    def choose_grad_split(self, tail_fraction=0.9):
        """-> (first_op, offset): the LAST op (the earliest point of the backward pass, which runs the ops last to first) whose tail of
        the flat layout still holds at least tail_fraction of the variational parameters."""
        best = None
        for i, o in enumerate(self.prog.ops):
            if o["type"] != L.OP_CONV:
                continue
            if self.prog.n_vi - o["w_off"] >= tail_fraction * self.prog.n_vi:
                best = i
        if best is None:
            raise ValueError("no op owns a tail of %.2f of the parameters" % tail_fraction)
        return best, self.grad_split_offset(best)

    def profile(self, mode, op=-1, pass_=-1):
        L.check(L.lib().mfvi_plan_profile(self.handle, mode, op, pass_))

    def profile_read(self, capacity=65536):
        """-> list of (op index, pass, milliseconds) recorded since the last read."""
        n = C.c_int(0)
        ops = (C.c_int * capacity)(); passes = (C.c_int * capacity)(); ms = (C.c_float * capacity)()
        L.check(L.lib().mfvi_plan_profile_read(self.handle, capacity, C.byref(n), ops, passes, ms))
        k = min(n.value, capacity)
        return [(ops[i], passes[i], ms[i]) for i in range(k)]

    def read_tensor(self, tid, sample=0, which=0):
        import torch
        t = self.prog.tensors[tid]
        if which in (0, 1):
            dst = torch.empty((t["C"], t["H"], t["W"]), dtype=torch.float32, device="cuda")
        else:
            dst = torch.empty((t["C"], 2), dtype=torch.float64, device="cuda")
        L.check(L.lib().mfvi_plan_read_tensor(self.handle, L.ptr(self.workspace), tid, sample, which, L.ptr(dst), L.stream_ptr()))
        return dst


def skip_program(H, W, input_depth=16, n_out=2, nd=(16, 32, 64, 128, 128), nu=(16, 32, 64, 128, 128), ns=(4, 4, 4, 4, 4),
                 fd=3, fu=3, fs=1, need1x1_up=True, upsample_mode="bilinear", drop_down=0.0, drop_up=0.0, lrt=False):
    """The skip() hour-glass of the reference (models/skip.py:58-134) as a layer program, in module order:
    per scale  [skip-conv/BN/act], down-conv(s2)/BN/act, conv/BN/act, [deeper scale], Upsample, [Concat], BN,
    up-conv/BN/act, [1x1-conv/BN/act]; then the final 1x1 conv.  ns[i] == 0 drops the skip branch and its Concat
    (models/skip.py:62-66), need1x1_up / filter sizes / upsample_mode as in the inpainting runner
    (bayesian_optimization.py:2970-2998).  drop_down / drop_up > 0 put nn.Dropout2d(p) after the deeper / up convolutions
    (dropout_mode_down = dropout_mode_up = '2d' of the MC-dropout runners, bayesian_optimization.py:1526-1549).
    lrt=True builds every convolution as a local-reparameterisation layer (MeanFieldVI(reparam='local')).
    Returns (program, input_id, output_id, tensor-id map)."""
    P = Program()
    _conv = P.conv
    P.conv = lambda *a, **kw: _conv(*a, lrt=lrt, **kw)
    names = {}
    zin = P.tensor(input_depth, H, W)

    def scale(i, x):
        h, w = P.tensors[x]["H"], P.tensors[x]["W"]
        s = None
        if ns[i]:
            s = P.tensor(ns[i], *P.conv_out_hw(x, fs, 1)); P.conv(x, s, fs, 1); P.set_bn(s, act=True)
        d1 = P.tensor(nd[i], *P.conv_out_hw(x, fd, 2)); P.conv(x, d1, fd, 2); P.set_bn(d1, act=True)
        d2 = P.tensor(nd[i], *P.conv_out_hw(d1, fd, 1)); P.conv(d1, d2, fd, 1); P.set_bn(d2, act=True)
        deep, kk = d2, nd[i]
        if i < len(nd) - 1:
            deep, kk = scale(i + 1, d2), nu[i + 1]
        # Concat centre-crops the up-sampled branch to the skip branch (models/common.py:31-41): at an odd h the 2*ceil(h/2) rows lose the last
        # one.  Without a skip branch there is no Concat and the up-sampled size stands (models/skip.py:62-66).
        ch, cw = (h, w) if s is not None else (2 * P.tensors[deep]["H"], 2 * P.tensors[deep]["W"])
        cat = P.tensor(ns[i] + kk, ch, cw); P.concat_up(s, deep, cat, upsample_mode); P.set_bn(cat, act=False)
        u = P.tensor(nu[i], *P.conv_out_hw(cat, fu, 1)); P.conv(cat, u, fu, 1); P.set_bn(u, act=True)
        top_i = u
        if need1x1_up:
            top_i = P.tensor(nu[i], ch, cw); P.conv(u, top_i, 1, 1); P.set_bn(top_i, act=True)
        for t, p in ((d1, drop_down), (d2, drop_down), (u, drop_up)) + (((top_i, drop_up),) if need1x1_up else ()):
            if p:
                P.set_dropout(t, p)
        names[i] = dict(skip=s, d1=d1, d2=d2, cat=cat, up=u, up1=top_i)
        return top_i

    top = scale(0, zin)
    out = P.tensor(n_out, P.tensors[top]["H"], P.tensors[top]["W"]); P.conv(top, out, 1, 1)
    del P.conv
    return P, zin, out, names
