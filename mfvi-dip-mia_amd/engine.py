"""Fused tempered-ELBO iteration on one GPU (or one rank of a K-sharded job).

One `step()` is the body of the reference's hot loop (bayesian_optimization.py:1360-1381 for denoising,
:2176-2191 SR, :568-582 CT) with K Monte-Carlo forwards instead of one (K sequential
batch-1 forwards on the same perturbed input, loss averaged):

    z      = z0 + 0.1 * N(0,1)                              (RNG domain INPUT, step)
    out_k  = net_k(z),  k = 0..K-1                          (eps keyed by the GLOBAL sample index)
    loss   = 1/K sum_k NLL(out_k) + temp * KL
    grads  -> [all-reduce over ranks when K is sharded] -> Adam

All arithmetic runs in libmfvi_hip; torch only owns the buffers, the stream and (optionally) the process group.
"""
import math

from . import _lib as L
from .program import skip_program
from .sharding import allreduce_sum_, shard_samples

TASK_DEN, TASK_SR, TASK_CT, TASK_INP = "den", "sr", "ct", "inp"
# the inpainting runner's net (bayesian_optimization.py:2970-2998): 6 scales, no skip branches, 5x5 down filters, nearest upsampling,
# no 1x1 up convolutions, 3 colour logits + 1 log-precision channel
INP_NET = dict(nd=(16, 32, 64, 128, 128, 128), nu=(16, 32, 64, 128, 128, 128), ns=(0, 0, 0, 0, 0, 0), fd=5, fu=3, need1x1_up=False,
               upsample_mode="nearest")


class ElboEngine:
    def __init__(self, H, W, task=TASK_DEN, K=1, input_depth=16, temp=1.0, sigma=0.1, lr=1e-3, seed=1, sr_factor=4,
                 theta_deg=None, rank=0, world_size=1, process_group=None, samples_per_launch=None, net_kwargs=None,
                 autotune=True, param_dtype="f32"):
        import torch
        self.torch = torch
        self.task, self.K, self.H, self.W = task, int(K), H, W
        self.rank, self.world = rank, world_size
        self.pg = process_group
        self.k0, self.K_local = shard_samples(self.K, rank, world_size)
        self.temp, self.lr, self.seed = float(temp), float(lr), int(seed)
        # prior scale exactly as bayesian_optimization.py:1335-1336 + modules/module.py:38, rounded to fp32
        import numpy as np
        self.prior_sigma = float(np.float32(math.sqrt(temp) * sigma + 1e-6))
        n_out = {TASK_CT: 1, TASK_INP: 4}.get(task, 2)
        kw = dict(INP_NET if task == TASK_INP else {})
        kw.update(net_kwargs or {})
        self.prog, self.zin, self.zout, self.names = skip_program(H, W, input_depth, n_out, **kw)
        self.chunk = min(self.K_local, samples_per_launch or self.K_local)
        self.param_dtype = param_dtype
        self.plan = self.prog.compile(self.zin, self.zout, self.chunk, param_dtype=param_dtype)
        P = self.prog
        self.n_vi, self.n_bn = P.n_vi, P.n_bn
        self.n_params = 2 * P.n_vi + P.n_bn
        dev = "cuda"
        # gradients (+8 scalars riding the all-reduce) and the float64 loss accumulators in ONE allocation: one fill launch clears both per iteration
        n_g = (self.n_params + 8 + 1) // 2 * 2                                            # the accumulators start 8-byte aligned
        self._gbuf = torch.zeros(4 * n_g + 32, dtype=torch.uint8, device=dev)
        self.grads = self._gbuf[:4 * (self.n_params + 8)].view(torch.float32)
        self.m = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        if param_dtype == "bf16":
            # BASELINE configs[4]: mu / rho live in bfloat16 (no float32 master copy; stochastic rounding in the update), BN stays float32
            if task == TASK_CT or type(self) is not ElboEngine:
                raise NotImplementedError("bf16 parameter storage is built for the MFVI den / sr / inpainting engines")
            pad = (P.n_vi + 7) // 8 * 8                                                   # the RHO block starts 8-byte aligned
            self.params16 = torch.zeros(2 * pad, dtype=torch.bfloat16, device=dev)
            self.mu, self.rho = self.params16[:P.n_vi], self.params16[pad:pad + P.n_vi]
            self.bn = torch.empty(max(P.n_bn, 1), dtype=torch.float32, device=dev)[:P.n_bn]
            self.params = None
        else:
            self.params = torch.empty(self.n_params, dtype=torch.float32, device=dev)
            self.mu, self.rho, self.bn = self.params[:P.n_vi], self.params[P.n_vi:2 * P.n_vi], self.params[2 * P.n_vi:]
        self.dmu, self.drho, self.dbn = self.grads[:P.n_vi], self.grads[P.n_vi:2 * P.n_vi], self.grads[2 * P.n_vi:self.n_params]
        self.z0 = torch.empty((input_depth, H, W), dtype=torch.float32, device=dev)
        self.z = torch.empty_like(self.z0)
        self.out = torch.empty((self.chunk, n_out, H, W), dtype=torch.float32, device=dev)
        self.dout = torch.empty_like(self.out)
        self.acc = self._gbuf[4 * n_g:].view(torch.float64)                # [0] nll sum, [1] kl
        self.upd_scratch = torch.zeros(L.lib().mfvi_elbo_update_scratch_bytes(), dtype=torch.uint8, device=dev)
        self.t_applied = torch.zeros(1, dtype=torch.int32, device=dev)    # CT: optimizer steps actually taken (the NaN guard skips some)
        self.sr_factor = sr_factor
        self.theta = None
        if task == TASK_CT:
            th = theta_deg if theta_deg is not None else list(range(0, 180, 4))        # bayesian_optimization.py:545
            self.theta = torch.tensor(th, dtype=torch.float32, device=dev)
            self.ct_scratch = torch.empty(self.chunk * len(th) * W, dtype=torch.float32, device=dev)
        self.t = 0
        self.step_dev = None             # device-resident iteration (enable_device_step / enable_graph): the RNG step counter on the device
        self._graph = None
        self.target = None
        self.sample_weights = True       # w = mu + softplus(rho) * eps; the non-Bayesian siblings run w = mu
        self.init_params()
        if autotune:      # one-time: pick the fastest kernel tiling per layer on this device (results unchanged)
            self.plan.autotune(self.mu, self.rho, self.bn, self.z0, self.chunk)

    # -------------------------------------------------------------------------------------------
    def init_params(self):
        """mu ~ N(0, 0.1), rho ~ N(-3, 0.1) (BayTorch/modules/module.py:26-30,56-62), BN gamma=1, beta=0;
        z0 = 0.1*U(0,1) (utils/common_utils.py:134-162).  Drawn from the RNG spec (domain INIT / UNIFORM)."""
        lib, sp = L.lib(), L.stream_ptr()
        if self.param_dtype == "bf16":      # the float32 draw of the spec, rounded to nearest-even bf16
            tmp = self.torch.empty(self.n_vi, dtype=self.torch.float32, device="cuda")
            for stream_id, (a, b), dst in ((0, (0.0, 0.1), self.mu), (1, (-3.0, 0.1), self.rho)):
                L.check(lib.mfvi_normal_fill(self.seed, L.DOMAIN_INIT, stream_id, 0, 0, self.n_vi, a, b, L.ptr(tmp), sp))
                L.check(lib.mfvi_f32_to_bf16(L.ptr(tmp), self.n_vi, L.ptr(dst), sp))
        else:
            L.check(lib.mfvi_normal_fill(self.seed, L.DOMAIN_INIT, 0, 0, 0, self.n_vi, 0.0, 0.1, L.ptr(self.mu), sp))
            L.check(lib.mfvi_normal_fill(self.seed, L.DOMAIN_INIT, 1, 0, 0, self.n_vi, -3.0, 0.1, L.ptr(self.rho), sp))
        self.bn.zero_()
        for b in self.prog.bns:
            self.bn[b["off"]:b["off"] + b["C"]] = 1.0
        L.check(lib.mfvi_uniform_fill(self.seed, 0, 0, 0, self.z0.numel(), 0.1, L.ptr(self.z0), sp))
        self.m.zero_(); self.v.zero_(); self.t = 0; self.t_applied.zero_()

    def set_target(self, target, mask=None):
        """den: noisy image [H][W]; sr: low-res image [H/f][W/f]; ct: sinogram [T][W]; inp: colour image [3][H][W] with
        mask [1|3][H][W] (1 = known pixel, rounded like bayesian_optimization.py:3024)."""
        self.target = target.contiguous().float().cuda()
        if self.task == TASK_INP:
            if mask is None:
                raise ValueError("the inpainting task needs a mask")
            self.mask = mask.contiguous().float().cuda().round()
            if self.mask.dim() == 2:
                self.mask = self.mask[None]

    # -------------------------------------------------------------------------------------------
    def forward_only(self, step=None, perturb=True):
        """K_local MC forward passes (no loss/backward): the 'MC-forward-passes/s' leg of the metric."""
        lib, sp = L.lib(), L.stream_ptr()
        step = self.t if step is None else step
        if perturb:
            self._perturb(step)
        if self.step_dev is not None:
            step = 0                      # an offset to the device counter
        for c0 in range(0, self.K_local, self.chunk):
            n = min(self.chunk, self.K_local - c0)
            self.plan.forward(self.mu, self.rho, self.bn, self.z if perturb else self.z0, self.seed, step, self.k0 + c0, n, self.sample_weights,
                              self.out)
        return self.out

    def _perturb(self, step):
        lib, sp = L.lib(), L.stream_ptr()
        if self.step_dev is not None:      # counter = *step_dev (+ 0) read on the device
            L.check(lib.mfvi_perturb_input_dev(L.ptr(self.z0), self.seed, L.ptr(self.step_dev), 0, self.z0.numel(), 0.1, L.ptr(self.z), sp))
        else:
            L.check(lib.mfvi_perturb_input(L.ptr(self.z0), self.seed, step, self.z0.numel(), 0.1, L.ptr(self.z), sp))

    def _loss_and_dout(self, n):
        lib, sp = L.lib(), L.stream_ptr()
        scale = 1.0 / self.K
        if self.task == TASK_DEN:
            L.check(lib.mfvi_gaussian_nll(L.ptr(self.out), L.ptr(self.target), n, self.H, self.W, 1, scale, L.ptr(self.dout), L.ptr(self.acc), sp))
        elif self.task == TASK_SR:
            L.check(lib.mfvi_gaussian_nll(L.ptr(self.out), L.ptr(self.target), n, self.H, self.W, self.sr_factor, scale, L.ptr(self.dout), L.ptr(self.acc), sp))
        elif self.task == TASK_INP:
            L.check(lib.mfvi_gaussian_nll_inpainting(L.ptr(self.out), L.ptr(self.target), L.ptr(self.mask), self.mask.shape[0], n, self.H, self.W,
                                                     scale, L.ptr(self.dout), L.ptr(self.acc), sp))
        else:
            L.check(lib.mfvi_radon_mse(L.ptr(self.out), L.ptr(self.target), L.ptr(self.theta), n, self.H, self.W, self.theta.numel(), scale,
                                       L.ptr(self.ct_scratch), L.ptr(self.dout), L.ptr(self.acc), sp))

    def grad_only(self, step=None, perturb=True, with_kl=True, after_forward=None):
        """Everything of one iteration except the optimizer update; returns nothing (grads, acc hold the result)."""
        lib, sp = L.lib(), L.stream_ptr()
        step = self.t if step is None else step
        self._gbuf.zero_()                     # grads and acc
        exchange = self.world > 1 or getattr(self, "_force_exchange", False)
        zsrc = self.z0
        if perturb:
            self._perturb(step)
            zsrc = self.z
        if self.step_dev is not None:
            step = 0                      # an offset to the device counter
        for c0 in range(0, self.K_local, self.chunk):
            n = min(self.chunk, self.K_local - c0)
            self.plan.forward(self.mu, self.rho, self.bn, zsrc, self.seed, step, self.k0 + c0, n, self.sample_weights, self.out)
            self._loss_and_dout(n)
            if after_forward is not None and c0 + n >= self.K_local:
                after_forward(n)       # self.out holds the n samples of the LAST launch, final in stream order here: work that only reads it can overlap the backward pass
            split = exchange and getattr(self, "_ov", None) is not None and c0 + n >= self.K_local      # the gradients are final after the LAST launch
            if split:
                self.plan.grad_split(self._ov["op"], self._ov["stream"])
            try:
                self.plan.backward(self.mu, self.rho, self.bn, zsrc, self.seed, step, self.k0 + c0, n, self.dout, self.dmu, self.drho, self.dbn,
                                   self.sample_weights)
            finally:
                if split:
                    self.plan.grad_split(-1)
        if exchange:
            if self.grads.is_cuda:      # HIP events around the exchange (bench.py's allreduce_ms): on the caller's stream — with the overlap on, what is left exposed
                ev = self._ar_events = getattr(self, "_ar_events", [])
                if len(ev) >= 64:
                    ev.pop(0)
                a_, b_ = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
                a_.record(); self._exchange(); b_.record()
                ev.append((a_, b_))
            else:
                self._exchange()
        if with_kl:
            # KL and its gradient are deterministic: every rank computes them redundantly (no communication)
            mu, rho = self.mu, self.rho
            if self.param_dtype == "bf16":      # the KL of the float32 values the bf16 parameters denote (step() fuses this into the update)
                mu, rho = self.params_f32()[:2]
            L.check(lib.mfvi_kl(L.ptr(mu), L.ptr(rho), self.n_vi, 0.0, self.prior_sigma, L.ptr(self.acc[1:]), sp))
            L.check(lib.mfvi_kl_backward(L.ptr(mu), L.ptr(rho), self.n_vi, 0.0, self.prior_sigma, self.temp,
                                         L.ptr(self.dmu), L.ptr(self.drho), sp))

    def set_allreduce_overlap(self, enabled, tail_fraction=0.9):
        """K sharded over ranks: start the exchange under the tail of the backward pass.  The weight gradients of the deep / up-path ops —
        a tail of the flat layout holding >= tail_fraction of the parameters — are complete when the backward pass (last op first) reaches
        the top scales of the down path; the plan reduces them on a second stream there (mfvi_plan_set_grad_split) and their all-reduce is
        enqueued on that stream, beside the rest of the pass.  The head of the layout, d BN and the NLL scalar follow in one packed
        all-reduce after the pass.  The same element-wise sums either way: with two ranks (one addition per element) the update is
        bit-identical (tests/test_gpu_multirank.py); with more ranks a ring / tree collective cuts the three buffers of the split
        schedule into other chunks than the one flat buffer, so the per-element order of the additions — and with it the last bit — may
        differ between the two schedules.  bench.py records which schedule ran (`allreduce_overlap`); MFVI_AR_OVERLAP=0|1 pins it."""
        self._ov = None
        if not enabled:
            return
        op, off = self.plan.choose_grad_split(tail_fraction)
        n_rest = self.grads.numel() - 2 * self.n_vi               # d BN + the 8 scalars
        self._ov = dict(op=op, off=off, stream=self.torch.cuda.Stream(),
                        head=self.torch.empty(2 * off + n_rest, dtype=self.torch.float32, device=self.grads.device))

    def _exchange(self):
        """The single exchange of the K-sharded step: grads (+ the NLL scalar) summed over ranks."""
        torch, ov, n_vi = self.torch, getattr(self, "_ov", None), self.n_vi
        self.grads[self.n_params] = self.acc[0].float()
        force = getattr(self, "_force_exchange", False)
        if ov is None:
            allreduce_sum_(self.grads, self.pg, force)
            return
        off, head, main = ov["off"], ov["head"], torch.cuda.current_stream()
        with torch.cuda.stream(ov["stream"]):       # behind the plan's early gradient reduction (stream order), beside the rest of the pass
            allreduce_sum_(self.dmu[off:], self.pg, force)
            allreduce_sum_(self.drho[off:], self.pg, force)
        pieces = (self.dmu[:off], self.drho[:off], self.grads[2 * n_vi:])
        torch.cat(pieces, out=head)
        allreduce_sum_(head, self.pg, force)
        o = 0
        for p in pieces:
            p.copy_(head[o:o + p.numel()]); o += p.numel()
        main.wait_stream(ov["stream"])

    def allreduce_ms(self):
        """Mean duration (ms) of the gradient all-reduce over the last <= 64 iterations, from events on the stream it was enqueued on
        (synchronises); None on a single rank."""
        ev = getattr(self, "_ar_events", [])
        if not ev:
            return None
        self.torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def params_f32(self):
        """(mu, rho, bn) as float32 tensors: views of the parameter buffer, or the expansion of the bf16 blocks (mfvi_bf16_to_f32)."""
        if self.param_dtype != "bf16":
            return self.mu, self.rho, self.bn
        lib, sp = L.lib(), L.stream_ptr()
        out = self.torch.empty(2 * self.n_vi, dtype=self.torch.float32, device="cuda")
        L.check(lib.mfvi_bf16_to_f32(L.ptr(self.mu), self.n_vi, L.ptr(out), sp))
        L.check(lib.mfvi_bf16_to_f32(L.ptr(self.rho), self.n_vi, L.ptr(out[self.n_vi:]), sp))
        return out[:self.n_vi], out[self.n_vi:], self.bn

    def enable_device_step(self):
        """Device-resident iteration state: the RNG step counter (the reference's loop index i, bayesian_optimization.py:1360) and Adam's
        update count live in device memory — the kernels read them there (mfvi_plan_set_step_source; the update through the device counter
        of mfvi_elbo_update_guarded) and a one-element add inside the iteration advances the step.  Same arithmetic as the host-driven path;
        what it buys is that an iteration no longer depends on any host value: it can be captured once and replayed (enable_graph)."""
        if self.param_dtype == "bf16" or type(self) is not ElboEngine:
            raise NotImplementedError("the device-resident iteration is built for the float32 MFVI engine")
        if self.step_dev is None:
            torch = self.torch
            self.step_dev = torch.full((1,), self.t, dtype=torch.int32, device="cuda")
            self.t_applied.fill_(self.t)                      # Adam's count of applied updates (CT: the NaN guard may hold it back)
            L.check(L.lib().mfvi_plan_set_step_source(self.plan.handle, L.ptr(self.step_dev)))
        return self

    def _device_iteration(self, after_forward=None):
        lib, sp = L.lib(), L.stream_ptr()
        self.grad_only(0, with_kl=False, after_forward=after_forward)
        guard = self._guard() if self.task == TASK_CT else (None, None)
        L.check(lib.mfvi_elbo_update_guarded(L.ptr(self.params), L.ptr(self.grads), L.ptr(self.m), L.ptr(self.v), self.n_vi, self.n_bn, 0.0,
                                             self.prior_sigma, self.temp, self.lr, 0.9, 0.999, 1e-8, L.ptr(self.t_applied), *guard,
                                             L.ptr(self.acc[1:]), L.ptr(self.upd_scratch), sp))
        self.step_dev.add_(1)

    def enable_graph(self, warmup=3, side_stream=None):
        """Capture ONE iteration (both streams of the plan, fork / join events as graph edges) into a HIP graph and replay it from then on:
        step() becomes one hipGraphLaunch.  `warmup` un-captured iterations run first (tables uploaded, side stream and events created —
        nothing may be allocated under capture).  The loop being matched: bayesian_optimization.py:1360-1372."""
        torch = self.torch
        self.enable_device_step()
        if self.world > 1:
            raise NotImplementedError("graph capture of the K-sharded iteration (the all-reduce inside the graph) is not built")
        if side_stream is None:
            import os
            side_stream = os.environ.get("MFVI_GRAPH_SIDE", "1") != "0"
        if not side_stream:      # one linear chain of kernel nodes (no fork / join edges)
            self.plan.side_stream(False)
        for _ in range(max(1, warmup)):
            self._device_iteration(); self.t += 1
        torch.cuda.synchronize()
        L.check(L.lib().mfvi_plan_set_capture_mode(self.plan.handle, 1))
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                self._device_iteration()
        finally:
            L.check(L.lib().mfvi_plan_set_capture_mode(self.plan.handle, 0))
        # the capture itself executes nothing: the counters still say "iteration self.t"
        self._graph = g
        return self

    def step(self, after_forward=None):
        """One ELBO iteration: K forwards + NLL + backward + (all-reduce), then KL + its gradient + Adam in one fused launch
        (identical on every rank: no communication).  after_forward: optional callable f(n) run once the last forward (n samples in self.out) + data term of the
        iteration are enqueued (the runners start their per-iteration bookkeeping there, on a second stream beside the backward pass)."""
        lib, sp = L.lib(), L.stream_ptr()
        if self._graph is not None and after_forward is None:
            self._graph.replay(); self.t += 1
            return
        if self.step_dev is not None:
            self._device_iteration(after_forward); self.t += 1
            return
        self.grad_only(self.t, with_kl=False, after_forward=after_forward)
        self.t += 1
        if self.param_dtype == "bf16":
            L.check(lib.mfvi_elbo_update_bf16(L.ptr(self.mu), L.ptr(self.rho), L.ptr(self.bn), L.ptr(self.grads), L.ptr(self.m), L.ptr(self.v), self.n_vi,
                                              self.n_bn, 0.0, self.prior_sigma, self.temp, self.lr, 0.9, 0.999, 1e-8, self.t, self.seed,
                                              L.ptr(self.acc[1:]), L.ptr(self.upd_scratch), sp))
            return
        if self.task == TASK_CT:
            # `if not torch.isnan(loss): optimizer.step()` (bayesian_optimization.py:581-582) decided on the device: no host sync
            L.check(lib.mfvi_elbo_update_guarded(L.ptr(self.params), L.ptr(self.grads), L.ptr(self.m), L.ptr(self.v), self.n_vi, self.n_bn, 0.0,
                                                 self.prior_sigma, self.temp, self.lr, 0.9, 0.999, 1e-8, L.ptr(self.t_applied), *self._guard(),
                                                 L.ptr(self.acc[1:]), L.ptr(self.upd_scratch), sp))
            return
        L.check(lib.mfvi_elbo_update(L.ptr(self.params), L.ptr(self.grads), L.ptr(self.m), L.ptr(self.v), self.n_vi, self.n_bn, 0.0,
                                     self.prior_sigma, self.temp, self.lr, 0.9, 0.999, 1e-8, self.t, L.ptr(self.acc[1:]),
                                     L.ptr(self.upd_scratch), sp))

    def _guard(self):
        """(loss_d, loss_f) of the NaN guard: the local double accumulator, or the float that rode the all-reduce when K is sharded
        (every rank then takes the same decision)."""
        if self.world > 1:
            return None, L.ptr(self.grads[self.n_params:])
        return L.ptr(self.acc), None

    def losses(self):
        """(nll, kl, loss) of the last grad_only/step — forces a device sync."""
        if self.world > 1:
            nll = float(self.grads[self.n_params]) / self.K
        else:
            nll = float(self.acc[0]) / self.K
        kl = float(self.acc[1])
        return nll, kl, nll + self.temp * kl


METHOD_DIP, METHOD_MCD, METHOD_SGLD = "dip", "mcd", "sgld"


class SiblingEngine(ElboEngine):
    """The reference's non-Bayesian comparison methods on the same layer program and kernels:

    dip   plain deep image prior: w = mu, MSE on the image channel, AdamW(wd=0)      (bayesian_optimization.py:1064-1237)
    mcd   MC dropout: Dropout2d(p) after the deeper / up convolutions (train mode throughout), heteroscedastic NLL,
          AdamW(weight_decay)                                                          (bayesian_optimization.py:1447-1655)
    sgld  N(0, (2*lr0)^2) noise on the 4-D conv weights before every forward (add_noise, :166-170), MSE (den) / NLL (sr),
          AdamW(weight_decay) with ExponentialLR(gamma) while lr > 1e-8               (bayesian_optimization.py:1658-1860)

    Parameters live in the MU block (RHO is unused, zero); nn.Conv2d's default initialisation (kaiming-uniform with a = sqrt(5):
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias) is drawn from the RNG spec."""

    def __init__(self, H, W, method=METHOD_DIP, task=TASK_DEN, weight_decay=0.0, dropout_p=0.3, gamma=0.996, param_noise_sigma=2.0,
                 net_kwargs=None, **kw):
        if method not in (METHOD_DIP, METHOD_MCD, METHOD_SGLD):
            raise ValueError("method %r: 'dip', 'mcd' or 'sgld'" % (method,))
        self.method = method
        self.weight_decay, self.gamma, self.param_noise_sigma = float(weight_decay), float(gamma), float(param_noise_sigma)
        nk = dict(net_kwargs or {})
        if method == METHOD_MCD:
            nk.update(drop_down=float(dropout_p), drop_up=float(dropout_p))
        kw.pop("temp", None); kw.pop("sigma", None)
        super().__init__(H, W, task=task, temp=0.0, sigma=0.0, net_kwargs=nk, **kw)
        self.lr0 = self.lr                    # add_noise keeps using the initial rate (LR, not the scheduler's)

    def init_params(self):
        lib, sp = L.lib(), L.stream_ptr()
        self.sample_weights = False
        self.params.zero_()
        for lid, lay in enumerate(self.prog.layers):
            bound = 1.0 / math.sqrt(lay["cin"] * lay["k"] * lay["k"])
            n = lay["cout"] * lay["cin"] * lay["k"] * lay["k"] + (lay["cout"] if lay["b_off"] >= 0 else 0)
            L.check(lib.mfvi_uniform_fill_range(self.seed, 16 + lid, 0, 0, n, -bound, bound, L.ptr(self.mu[lay["w_off"]:]), sp))
        for b in self.prog.bns:
            self.bn[b["off"]:b["off"] + b["C"]] = 1.0
        L.check(lib.mfvi_uniform_fill(self.seed, 0, 0, 0, self.z0.numel(), 0.1, L.ptr(self.z0), sp))
        self.m.zero_(); self.v.zero_(); self.t = 0; self.t_applied.zero_()

    def _loss_and_dout(self, n):
        lib, sp = L.lib(), L.stream_ptr()
        mse_image = self.method == METHOD_DIP or (self.method == METHOD_SGLD and self.task == TASK_DEN)
        if self.task in (TASK_DEN, TASK_SR) and mse_image:      # F.mse_loss(out[:, :1], target)  (:1177, :1780, :1985)
            f = self.sr_factor if self.task == TASK_SR else 1
            L.check(lib.mfvi_mse_channel(L.ptr(self.out), L.ptr(self.target), n, self.out.shape[1], self.H, self.W, 0, f, 1.0 / self.K,
                                         L.ptr(self.dout), L.ptr(self.acc), sp))
        elif self.task == TASK_INP and self.method == METHOD_DIP:  # mse_loss(out[:, :3].sigmoid() * mask, img * mask)  (:2824-2826)
            L.check(lib.mfvi_mse_sigmoid_masked(L.ptr(self.out), L.ptr(self.target), L.ptr(self.mask), self.mask.shape[0], n, self.H, self.W,
                                                1.0 / self.K, L.ptr(self.dout), L.ptr(self.acc), sp))
        else:                                                    # gaussian_nll / radon MSE / masked NLL as in the MFVI runners
            super()._loss_and_dout(n)

    def add_noise(self, step):
        """add_noise(net, param_noise_sigma, LR): 4-D parameters (the conv weights) only, std = sigma * lr0 (RNG domain 4)."""
        lib, sp = L.lib(), L.stream_ptr()
        std = self.param_noise_sigma * self.lr0
        for lid, lay in enumerate(self.prog.layers):
            n = lay["cout"] * lay["cin"] * lay["k"] * lay["k"]
            L.check(lib.mfvi_add_normal(L.ptr(self.mu[lay["w_off"]:]), self.seed, lid, step, n, std, sp))

    def step(self, after_forward=None):
        lib, sp = L.lib(), L.stream_ptr()
        if self.method == METHOD_SGLD:
            self.add_noise(self.t)
        self.grad_only(self.t, with_kl=False, after_forward=after_forward)
        self.t += 1
        n = self.n_vi
        for lo, hi in ((0, n), (2 * n, self.n_params)):          # MU block and BN block; RHO does not exist for these methods
            if self.task == TASK_CT:                              # NaN guard of the CT runners (bayesian_optimization.py:380, 792, 994)
                L.check(lib.mfvi_adamw_step_guarded(L.ptr(self.params[lo:]), L.ptr(self.grads[lo:]), L.ptr(self.m[lo:]), L.ptr(self.v[lo:]), hi - lo,
                                                    self.lr, 0.9, 0.999, 1e-8, L.ptr(self.t_applied), *self._guard(), self.weight_decay, sp))
            else:
                L.check(lib.mfvi_adamw_step(L.ptr(self.params[lo:]), L.ptr(self.grads[lo:]), L.ptr(self.m[lo:]), L.ptr(self.v[lo:]), hi - lo,
                                            self.lr, 0.9, 0.999, 1e-8, self.t, self.weight_decay, sp))
        if self.task == TASK_CT:
            L.check(lib.mfvi_step_advance(L.ptr(self.t_applied), *self._guard(), sp))
        if self.method == METHOD_SGLD and self.lr > 1e-8:        # scheduler.step() while get_last_lr() > 1e-8 (:1784-1785)
            self.lr *= self.gamma

    def losses(self):
        loss, _, _ = super().losses()
        return loss, 0.0, loss
