#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): MC-forward-passes/s and ELBO-iterations/s of the MFVI deep-image-prior fit.

Default workload = configs[1]: test_configs/mfvi_den.json values, 256x256 skip net, K = 16 MC samples per GPU.
One step = one tempered-ELBO iteration: input perturbation, weight draw, K MC forwards, data term, backward, KL(+grad), Adam (and, for
N > 1 ranks, the single all-reduce of the flat gradient buffer).

    python bench.py [--gpus N --steps K --warmup W] [--config cfg1|cfg2|cfg3|cfg4|cfg5|inp] [--scaling weak|strong] [--k K] [--mode engine|dropin]
      N > 1: one rank per GPU over RCCL — under torch.distributed.run (the driver's launch line: RANK / LOCAL_RANK / WORLD_SIZE /
      MASTER_* come from the environment), or started plainly, in which case this process spawns the N ranks itself (self_launch).

--mode engine (default): the fused runner (ElboEngine: ~10 library calls per iteration, no autograd).
--mode dropin: the loop a user of the reference runs after INTEGRATION.md's three-line switch — get_net + MeanFieldVI(n_samples=K) +
  gaussian_nll + temp * net.kl() + loss.backward() + torch.optim.AdamW over the wrapper's Parameters (bayesian_optimization.py:1356-1372;
  the reference's own loop is K = 1: `--mode dropin --k 1`).  Same JSON shape; den task, one GPU.

--scaling weak (default): every rank evaluates its own K samples (eps keyed by the global sample index): K * N samples per iteration.
--scaling strong: the config's K is the job's total, split over the ranks (cfg3 as BASELINE states it: K = 32 over 4 GPUs).
With N > 1 and the default weak mode the JSON line also carries a `strong_scaling` object (K total = the config's K) measured right after
the timed region, so one driver run records both curves.

Prints ONE JSON line on rank 0.  Data are synthetic (seeded phantom + noise from the package's own generator), weights random-init;
inputs are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak
F32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: FP32 vector == FP32-input MFMA peak
BF16_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak; a bf16x6 kernel spends 6 bf16 instructions per fp32 product: ceiling 2500 / 6
PASS_NAMES = {0: "fwd", 1: "bwd_weight", 2: "bwd_data", 3: "fold", 4: "concat_bwd", 5: "grad_finalize", 6: "sample_weights"}

# hyper-parameters: test_configs/mfvi_{den,sr,ct}.json:5,9,15-18 of the reference; inpainting: configs/mfvi_inp.json
DEN = dict(temp=5.656911698337764e-07, sigma=1.4616642493692077e-05, lr=1e-3, seed=1, p_sigma=0.1)
SR = dict(temp=4.381719802264805e-07, sigma=4.9e-08, lr=1e-3, seed=2, p_sigma=0.1)
CT = dict(temp=2.2e-10, sigma=1.7e-7, lr=1e-3, seed=1, p_sigma=0.1)
INP = dict(temp=1e-12, sigma=0.0006506274185234421, lr=2e-3, seed=2, p_sigma=0.1)
# BASELINE.json configs (K = MC samples per GPU in weak mode / per job in strong mode; spl = samples per launch)
CONFIGS = {
    "cfg1": dict(task="den", size=128, k=4, input_depth=16, hp=DEN, cpu_samples=64,
                 what="configs[0]: mfvi_den.json on one 128x128 grayscale slice, K=4 (the reference's CPU-runnable plumbing case)"),
    "cfg2": dict(task="den", size=256, k=16, input_depth=16, hp=DEN, cpu_samples=32,
                 what="configs[1]: mfvi_den.json hyper-parameters, 256x256 grayscale, K=16 MC samples per GPU, 26-layer skip net, single fused-HIP path"),
    "cfg3": dict(task="sr", size=512, k=8, k_strong=32, input_depth=32, hp=SR, cpu_samples=8,
                 what="configs[2]: mfvi_sr.json, 4x super-resolution, 512x512 target / 128x128 observation, input depth 32; K=32 over 4 GPUs = 8 per GPU"),
    "cfg4": dict(task="ct", size=256, k=16, input_depth=16, hp=CT, cpu_samples=32,
                 what="configs[3]: mfvi_ct.json, 256x256, 45-angle Radon forward / back-projection in HIP, one image per GPU, K=16"),
    "cfg5": dict(task="den", size=512, k=64, spl=16, input_depth=16, hp=DEN, cpu_samples=8, param_dtype="bf16",
                 what="configs[4]: one 512x512 denoising fit per GPU, K=64 (4 launches of 16), bf16 mu/rho with fp32 KL terms summed in fp64"),
    "inp": dict(task="inp", size=256, k=16, input_depth=16, hp=INP, cpu_samples=0,
                what="inpainting MFVI variant (SURVEY 8f rank 2): 6-scale no-skip net, 5x5 down filters, nearest upsampling, 256x256 colour image, K=16"),
}


def conv_cost(prog, op_index, n_samples):
    """Algorithmic bytes / FLOPs of one launch of the conv kernels of op `op_index`:
    input read once + output written once per sample, mu and rho read once; 2*MAC FLOPs."""
    o = prog.ops[op_index]
    if o["type"] != 1:
        return None
    ti, to = prog.tensors[o["in0"]], prog.tensors[o["out"]]
    k = o["ksize"]
    nw = to["C"] * ti["C"] * k * k
    bytes_ = 4 * n_samples * (ti["C"] * ti["H"] * ti["W"] + to["C"] * to["H"] * to["W"]) + 8 * (nw + to["C"])
    flops = 2.0 * n_samples * nw * to["H"] * to["W"]
    return dict(bytes=bytes_, flops=flops, desc="%dx%d conv %d->%d @%dx%d s%d" % (k, k, ti["C"], to["C"], to["H"], to["W"], o["stride"]))


def iteration_cost(prog, K):
    """Algorithmic FLOPs / bytes of ONE ELBO iteration with K MC samples: every conv forward once, backward twice
    (data + weight) — 3x the forward's FLOPs and bytes — plus KL (mu, rho read once) and Adam (4 tensors read, 3 written per parameter)."""
    fl = by = 0.0
    n_par = 0
    for i, o in enumerate(prog.ops):
        c = conv_cost(prog, i, K)
        if c:
            fl += c["flops"]; by += c["bytes"]
            ti, to = prog.tensors[o["in0"]], prog.tensors[o["out"]]
            n_par += 2 * (to["C"] * ti["C"] * o["ksize"] ** 2 + to["C"])
    n_par += prog.n_bn
    return dict(flops=3.0 * fl, bytes=3.0 * by + 4.0 * n_par + 28.0 * n_par)


# The reference's OWN PyTorch CPU path, measured once by the survey in the build container (the Python reference cannot
# travel to the GPU box, so this is a committed number, not something this script measures): K = 1 ELBO iterations of run_den_mfvi's
# loop (MeanFieldVI + skip + gaussian_nll + torch AdamW) at 256x256.
REFERENCE_CPU_PROBE = dict(value=5.61, unit="ELBO-iterations/s (K = 1 MC pass each)", ms_per_iteration=178.0, cores=8, threads=8,
                           software="torch 2.10.0 CPU, float32", workload="mfvi_den 256x256 skip net, K = 1",
                           source="BASELINE.md section 3 (survey probe in the build container, not measured by this run)")


def csrc_sha256():
    """sha256 over the kernel sources the running library is built from (csrc/*.hip, csrc/*.h, include/mfvi_hip.h, sorted by name).
    hipcc's output is not byte-reproducible (two builds of one source differ), so the SOURCES identify a build: __graft_entry__.build()
    recompiles them in place, and mfvi-dip-mia_amd/_build.py rebuilds whenever one of them is newer than the .so."""
    import glob
    import hashlib
    from mfvi_dip_mia_amd import _lib as L
    d = os.path.join(os.path.dirname(os.path.abspath(L.__file__)), "csrc")
    files = sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))) + [os.path.join(ROOT, "include", "mfvi_hip.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def traffic_lookup(key):
    """HBM bytes per launch of a kernel from the separate rocprofv3 --pmc passes (profiles/traffic.json), or None when the kernels that
    produced those counters are not the ones running now: the file carries the sha256 of the kernel sources it was measured on, and
    a library set with MFVI_LIB_PATH (an experimental build) never matches."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(tfile))
        if os.environ.get("MFVI_LIB_PATH") or t.get("csrc_sha256") != csrc_sha256():
            return None
        return t["entries"].get(key)
    except Exception:
        return None


def cpu_baseline(cfg, n_samples):
    """The CPU oracle (oracle/, the restatement of the reference path; kind = "port") timed on the host cores for a bounded sample of
    the same workload: n_samples MC passes (forward + data term + backward, then KL and one Adam step) of the config's net."""
    import numpy as np
    from oracle import oracle as O
    S, hp, task = cfg["size"], cfg["hp"], cfg["task"]
    n_out = 1 if task == "ct" else 2
    net = O.make_net(S, S, input_depth=cfg["input_depth"], n_out=n_out)
    mu, rho, bnp = O.init_params(net, hp["seed"])
    z = (0.1 * O.uniform_fill(hp["seed"], 0, 0, 0, cfg["input_depth"] * S * S)).reshape(cfg["input_depth"], S, S)
    img = O.phantom(S, S, hp["seed"])
    tgt = O.noisy(img, hp["p_sigma"], hp["seed"]); theta = None; tcode = 0
    if task == "sr":
        tgt = np.ascontiguousarray(img[::4, ::4]); tcode = 1
    elif task == "ct":
        theta = np.arange(0, 180., 4., dtype=np.float32); tgt = O.radon_fwd(img, theta); tcode = 2
    ps = float(np.float32(np.sqrt(hp["temp"]) * hp["sigma"] + 1e-6))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = O.set_threads(min(avail, 16))          # the GPU box gives one GPU a 16-core CPU share
    t0 = time.perf_counter()
    r = O.elbo_grad(net, mu, rho, bnp, z, tgt, task=tcode, factor=4, theta_deg=theta, seed=hp["seed"], step=0, K=n_samples, temp=hp["temp"], prior_sigma=ps)
    p = np.concatenate([mu, rho, bnp]); g = np.concatenate([r["dmu"], r["drho"], r["dbn"]])
    O.adam(p, g, np.zeros_like(p), np.zeros_like(p), hp["lr"], 1)
    dt = time.perf_counter() - t0
    return dict(value=n_samples / dt, unit="MC-forward-passes/s", cores=cores, kind="port",
                sample="%d MC passes (fwd + data term + bwd) + KL + Adam of the %dx%d %s net, C oracle with OpenMP on %d host threads, %.1f s"
                       % (n_samples, S, S, task, cores, dt))


def unfused_gpu_baseline(cfg, K, torch, steps=20, warmup=5, max_passes=480, device="cuda"):
    """SURVEY.md 8(d)'s same-node comparison row: the SAME iteration out of stock PyTorch-ROCm ops (MIOpen / ATen kernels, autograd,
    torch.optim.AdamW), float32, on the GPU this benchmark runs on.  The net is this package's own plain-torch module tree
    (nets.get_net, executable as is); every nn.Conv2d in it is swapped for a small module written here that holds (mu, rho) and runs
    w = mu + softplus(rho) * randn_like(mu), F.conv2d(x, w, b) — what BayTorch/modules/reparam_layers.py:26-37 does, restated, not the
    reference's file.  One iteration = the reference's loop body (bayesian_optimization.py:1360-1372) with K sequential batch-1 forwards on
    the same perturbed input and the loss averaged, the closed-form reverse KL of every layer, backward, AdamW(wd = 0).
    Untimed by the contract; `steps` iterations after `warmup` (MIOpen's kernel search runs in the warm-up), fewer when K is large."""
    import math
    import numpy as np
    import torch.nn as nn
    import torch.nn.functional as F
    from mfvi_dip_mia_amd import nets
    from mfvi_dip_mia_amd.engine import INP_NET
    from mfvi_dip_mia_amd.runner import phantom
    S, hp, task = cfg["size"], cfg["hp"], cfg["task"]
    dev = torch.device(device)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    prior_sigma = float(np.float32(math.sqrt(hp["temp"]) * hp["sigma"] + 1e-6))

    class ReparamConv(nn.Module):
        def __init__(self, conv):
            super().__init__()
            self.stride = conv.stride
            self.W_mu = nn.Parameter(torch.randn_like(conv.weight) * 0.1)
            self.W_rho = nn.Parameter(torch.randn_like(conv.weight) * 0.1 - 3.0)
            self.bias_mu = nn.Parameter(torch.randn_like(conv.bias) * 0.1)
            self.bias_rho = nn.Parameter(torch.randn_like(conv.bias) * 0.1 - 3.0)

        def forward(self, x):
            w = self.W_mu + F.softplus(self.W_rho) * torch.randn_like(self.W_mu)
            b = self.bias_mu + F.softplus(self.bias_rho) * torch.randn_like(self.bias_mu)
            return F.conv2d(x, w, b, self.stride)

        def kl(self):      # KL(N(0, s0^2) || N(mu, s^2)) summed: log(s / s0) + (s0^2 + mu^2) / (2 s^2) - 1/2
            t = 0.0
            for m, r in ((self.W_mu, self.W_rho), (self.bias_mu, self.bias_rho)):
                s = F.softplus(r)
                t = t + (torch.log(s / prior_sigma) + (prior_sigma ** 2 + m * m) / (2 * s * s) - 0.5).sum()
            return t

    def swap(mod):
        for name, ch in list(mod._modules.items()):
            if isinstance(ch, nn.Conv2d):
                mod._modules[name] = ReparamConv(ch)
            else:
                swap(ch)

    torch.manual_seed(hp["seed"])
    n_out = {"ct": 1, "inp": 4}.get(task, 2)
    if task == "inp":
        kw = INP_NET
        net = nets.skip(cfg["input_depth"], n_out, num_channels_down=list(kw["nd"]), num_channels_up=list(kw["nu"]), num_channels_skip=list(kw["ns"]),
                        filter_size_down=kw["fd"], filter_size_up=kw["fu"], need1x1_up=kw["need1x1_up"], upsample_mode=kw["upsample_mode"],
                        need_sigmoid=False, need_bias=True, pad='reflection', dropout_mode_down='None', dropout_mode_up='None')
    else:
        net = nets.get_net(cfg["input_depth"], 'skip', 'reflection', 'bilinear', n_channels=n_out, skip_n33d=[16, 32, 64, 128, 128],
                           skip_n33u=[16, 32, 64, 128, 128], skip_n11=4, num_scales=5)
    net = net.to(dev)
    swap(net)
    layers = [m for m in net.modules() if isinstance(m, ReparamConv)]
    opt = torch.optim.AdamW(net.parameters(), lr=hp["lr"], weight_decay=0)
    rng = np.random.default_rng(hp["seed"] + 1)
    img = torch.from_numpy(phantom(S, S, hp["seed"])).to(dev)[None, None]
    z0 = 0.1 * torch.rand((1, cfg["input_depth"], S, S), device=dev)
    if task == "den":
        target = torch.clamp(img + hp["p_sigma"] * torch.randn_like(img), 0, 1)
    elif task == "sr":
        target = img[..., ::4, ::4].contiguous()
    elif task == "ct":      # 45 rotations, bilinear sampling with zero padding, summed over the rows (radon/radon.py:23-55 restated)
        th = torch.deg2rad(torch.arange(0., 180., 4., device=dev))
        rot = torch.stack([torch.stack([th.cos(), -th.sin(), torch.zeros_like(th)], 1), torch.stack([th.sin(), th.cos(), torch.zeros_like(th)], 1)], 1)
        grid = F.affine_grid(rot, (th.numel(), 1, S, S), align_corners=False)
        radon = lambda x: F.grid_sample(x.expand(th.numel(), -1, -1, -1), grid, mode='bilinear', padding_mode='zeros', align_corners=False).sum(2).transpose(0, 1)[None]
        target = radon(img)
    else:
        img = torch.from_numpy(np.stack([phantom(S, S, hp["seed"] + c) for c in range(3)])).to(dev)[None]
        mask = torch.from_numpy((rng.random((1, 1, S, S)) > 0.12).astype(np.float32)).to(dev)

    def data_term(out):
        if task == "ct":
            return F.mse_loss(radon(out), target)
        if task == "inp":      # utils/bayesian_utils.py:35-39: masked heteroscedastic NLL on the sigmoid of the colour logits
            s = torch.clamp(out[:, 3:], -20, 20)
            return ((torch.exp(s) * (img - torch.sigmoid(out[:, :3])) ** 2 - s) * mask).mean()
        o = out[..., ::4, ::4] if task == "sr" else out
        s = torch.clamp(o[:, 1:], -20, 20)
        return (torch.exp(s) * (target - o[:, :1]) ** 2 - s).mean()

    def step():
        opt.zero_grad(set_to_none=True)
        z = z0 + 0.1 * torch.randn_like(z0)
        nll = 0.0
        for _ in range(K):      # K sequential batch-1 forwards; each backward frees its graph (the sum of the gradients is the gradient of the mean)
            l = data_term(net(z)) / K
            l.backward()
            nll = nll + l.detach()
        kl = sum(m.kl() for m in layers)
        (hp["temp"] * kl).backward()
        opt.step()
        return nll, kl

    steps = max(3, min(steps, max_passes // max(K, 1)))
    for _ in range(warmup):
        step()
    sync(); t0 = time.perf_counter()
    for _ in range(steps):
        nll, kl = step()
    sync(); dt = time.perf_counter() - t0
    n_launch = None
    try:      # launches per iteration, from the profiler's kernel records of ONE more iteration (information only)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step(); sync()
        n_launch = sum(e.count for e in prof.key_averages() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower())
    except Exception:
        n_launch = None
    return dict(ms_per_iteration=1e3 * dt / steps, value=K * steps / dt, unit="MC-forward-passes/s", elbo_iters_per_sec=steps / dt,
                steps=steps, warmup=warmup, dtype="f32", kernel_launches_per_iteration=n_launch,
                final_nll=float(nll), final_kl=float(kl.detach()),
                what="stock PyTorch-ROCm ops on the same GPU: nets.get_net module tree, plain-torch reparam conv (w = mu + softplus(rho) * randn), "
                     "K sequential batch-1 forwards + autograd + closed-form KL + torch.optim.AdamW, MIOpen warm (%d warm-up iterations)" % warmup,
                software="torch %s" % torch.__version__)


def make_engine(cfg, K, rank, world, torch):
    """Engine + synthetic target of a config, inputs from the package's own generator (runner.phantom)."""
    import numpy as np
    from mfvi_dip_mia_amd import _lib as L
    from mfvi_dip_mia_amd.engine import ElboEngine
    from mfvi_dip_mia_amd.runner import phantom
    S, hp, task = cfg["size"], cfg["hp"], cfg["task"]
    kw = dict(param_dtype=cfg["param_dtype"]) if "param_dtype" in cfg else {}
    eng = ElboEngine(S, S, task=task, K=K, input_depth=cfg["input_depth"], temp=hp["temp"], sigma=hp["sigma"], lr=hp["lr"], seed=hp["seed"],
                     rank=rank, world_size=world, samples_per_launch=cfg.get("spl"), **kw)
    rng = np.random.default_rng(hp["seed"] + 1)
    if task == "inp":
        img = np.stack([phantom(S, S, hp["seed"] + c) for c in range(3)])
        mask = (rng.random((1, S, S)) > 0.12).astype(np.float32)
        eng.set_target(torch.from_numpy(img), torch.from_numpy(mask))
        return eng
    img = phantom(S, S, hp["seed"])
    if task == "den":
        tgt = torch.from_numpy(np.clip(img + rng.normal(scale=hp["p_sigma"], size=img.shape), 0, 1).astype(np.float32))
    elif task == "sr":
        tgt = torch.from_numpy(np.ascontiguousarray(img[::4, ::4]))
    else:
        tgt = torch.empty((eng.theta.numel(), S), device="cuda")
        L.check(L.lib().mfvi_radon_forward(L.ptr(torch.from_numpy(img).cuda()), L.ptr(eng.theta), 1, S, S, eng.theta.numel(), L.ptr(tgt), L.stream_ptr()))
    eng.set_target(tgt)
    return eng


class DropInLoop:
    """The reference's own training loop (bayesian_optimization.py:1356-1372) on the drop-in classes, exactly as INTEGRATION.md's
    three-line switch leaves it: get_net -> MeanFieldVI -> torch.optim.AdamW; per iteration the input perturbation with torch's RNG
    (:1363-1364), net(z), gaussian_nll + temp * net.kl(), loss.backward(), optimizer.step().  K MC samples come from ONE call
    (MeanFieldVI(n_samples=K)); the reference's loop is K = 1.  Exposes what main() needs from an engine."""

    def __init__(self, cfg, K, torch, flat=False):
        import numpy as np
        import mfvi_dip_mia_amd as M
        from mfvi_dip_mia_amd.runner import phantom
        self.torch, self.M = torch, M
        S, hp = cfg["size"], cfg["hp"]
        if cfg["task"] != "den":
            sys.exit("--mode dropin is the denoising loop (run_den_mfvi)")
        dev = torch.device("cuda")
        torch.manual_seed(hp["seed"])
        net = M.get_net(cfg["input_depth"], 'skip', 'reflection', 'bilinear', n_channels=2, skip_n33d=[16, 32, 64, 128, 128],
                        skip_n33u=[16, 32, 64, 128, 128], skip_n11=4, num_scales=5)
        self.net = M.MeanFieldVI(net, prior={'mu': 0.0, 'sigma': float(np.sqrt(hp["temp"]) * hp["sigma"])}, replace_layers='all', device=dev,
                                 reparam='', n_samples=K, seed=hp["seed"], flat_parameters=flat)
        self.opt = torch.optim.AdamW(self.net.parameters(), lr=hp["lr"], weight_decay=0)
        self.temp = hp["temp"]
        self.K_local = self.chunk = K
        rng = np.random.default_rng(hp["seed"] + 1)
        img = phantom(S, S, hp["seed"])
        self.target = torch.from_numpy(np.clip(img + rng.normal(scale=hp["p_sigma"], size=img.shape), 0, 1).astype(np.float32)).to(dev)[None, None]
        self.z0 = 0.1 * torch.rand((1, cfg["input_depth"], S, S), device=dev)
        self.last = None
        self.step()                                          # compiles (and autotunes) the plan of this input shape
        self.plan = next(iter(self.net._plans.values()))
        self.prog = self.plan.prog

    def _loss(self, out):
        M = self.M
        nll = sum(M.gaussian_nll(out[i:i + 1, :1], out[i:i + 1, 1:], self.target) for i in range(out.shape[0])) / out.shape[0]
        kl = self.net.kl()
        return nll + self.temp * kl, nll, kl

    def step(self, after_forward=None):
        torch = self.torch
        z = self.z0 + 0.1 * torch.randn_like(self.z0)
        self.opt.zero_grad(set_to_none=True)
        out = self.net(z)
        loss, nll, kl = self._loss(out)
        loss.backward()
        self.opt.step()
        self.last = (nll.detach(), kl.detach(), loss.detach())

    def forward_only(self, step=0):
        with self.torch.no_grad():
            self.net(self.z0)

    def losses(self):
        nll, kl, loss = self.last
        return float(nll), float(kl), float(loss)

    def allreduce_ms(self):
        return None


def child_commands(n, argv, env, port=None):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): the N per-rank child processes the parent starts,
    as a list of (argv, env) — one fresh interpreter per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the same command line.
    Same shape as the reference's fan-out, which spawns its own children one per device (bayesian_optimization.py:3760-3775)."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = []
    for r in range(n):
        e = dict(env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out.append(([sys.executable, os.path.abspath(__file__)] + list(argv), e))
    return out


def self_launch(n, argv):
    """Parent of a plain `python bench.py --gpus N`: starts the ranks as child processes (never exec, never touches the GPU itself),
    relays rank 0's stdout (the ONE JSON line) and every rank's stderr, waits for all of them; exit code = first non-zero child code.
    A rank that dies takes the others down (they would otherwise wait in the rendezvous / a collective for ever)."""
    import subprocess
    procs = []
    for r, (cmd, env) in enumerate(child_commands(n, argv, os.environ)):
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for q in pending:
                    procs[q].terminate()
        if pending:
            time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--size", type=int, default=None, help="override the config's image size")
    ap.add_argument("--k", type=int, default=None, help="override the config's MC samples (per GPU in weak mode, per job in strong mode)")
    ap.add_argument("--mode", default="engine", choices=["engine", "dropin"])
    ap.add_argument("--flat-parameters", action="store_true", help="--mode dropin: MeanFieldVI(..., flat_parameters=True) (one flat Parameter for the optimizer)")
    ap.add_argument("--graph", action="store_true", help="--mode engine, one GPU: capture one iteration into a HIP graph (device-resident step counters) and time its replays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpu-baseline", action="store_true", help="skip the unfused PyTorch-ROCm leg (SURVEY 8d's same-node row)")
    ap.add_argument("--profile-all", action="store_true", help="print the per-kernel time table of one iteration to stderr")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.size:
        cfg["size"] = args.size
    if args.k:
        cfg["k"] = args.k; cfg.pop("k_strong", None)

    # started plainly with --gpus N > 1 (no launcher in front): this process becomes the parent of N ranks — decided before torch is
    # imported or the GPU touched (a process that has initialised the GPU must not be replaced, and is not: children are spawned)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d inside a %d-rank job (WORLD_SIZE): the two must agree" % (args.gpus, world))
    # MFVI_BENCH_BACKEND=gloo rehearses the N>1 path with several ranks on ONE GPU (RCCL refuses duplicate devices); the
    # driver's runs use the default: one rank per GPU over RCCL
    backend = os.environ.get("MFVI_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)         # "nccl" is RCCL on ROCm

    import mfvi_dip_mia_amd as M
    M._lib.lib()       # no library, no benchmark: fail loudly

    S = cfg["size"]
    strong = args.scaling == "strong"
    k_strong = cfg.get("k_strong", cfg["k"])
    K_total = k_strong if strong else cfg["k"] * world
    if K_total % world:
        sys.exit("K = %d MC samples do not split over %d ranks" % (K_total, world))
    if args.mode == "dropin":
        if world > 1:
            sys.exit("--mode dropin runs on one GPU (the reference's loop has no sharding)")
        eng = DropInLoop(cfg, K_total, torch, flat=args.flat_parameters)
    else:
        eng = make_engine(cfg, K_total, rank, world, torch)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed: warm-up, then one fully instrumented iteration to find the dominant kernel ----
    for _ in range(args.warmup):
        eng.step()
    torch.cuda.synchronize()
    # (the instrumented iteration runs every kernel ALONE on the stream: in the timed region the backward-weight kernels overlap the
    #  backward-data / concat chain on the plan's side stream, which stretches the launches that share the chip)
    eng.plan.side_stream(False)
    eng.plan.profile(1)
    runs = []
    for _ in range(5):          # five instrumented iterations, the median per kernel: one sample of a 100 us launch moves by 10 % from run to run
        eng.step()
        torch.cuda.synchronize()
        one = {}
        for op, ps_, ms in eng.plan.profile_read():
            one[(op, ps_)] = one.get((op, ps_), 0.0) + ms
        runs.append(one)
    eng.plan.profile(0)
    eng.plan.side_stream(True)
    by = {k_: sorted(r_.get(k_, 0.0) for r_ in runs)[len(runs) // 2] for k_ in runs[0]}
    n_launch = max(1, (eng.K_local + eng.chunk - 1) // eng.chunk)        # launches of a kernel per iteration (cfg5: 4)
    (dom_op, dom_pass), dom_ms = max(((k_, v) for k_, v in by.items() if k_[1] in (0, 1, 2) and k_[0] >= 0 and conv_cost(eng.prog, k_[0], 1)), key=lambda kv: kv[1])
    if args.profile_all and rank == 0:
        tot = sum(by.values())
        cls = {}
        for (op, ps_), ms in by.items():
            c = conv_cost(eng.prog, op, eng.chunk) if op >= 0 else None
            kind = PASS_NAMES[ps_] + (" " + c["desc"].split(" ")[0] + ("/s2" if c["desc"].endswith("s2") else "") if c else "")
            cls[kind] = cls.get(kind, 0.0) + ms
        for kind, ms in sorted(cls.items(), key=lambda kv: -kv[1]):
            sys.stderr.write("  %-22s %8.3f ms %5.1f%%\n" % (kind, ms, 100 * ms / tot))
        for (op, ps_), ms in sorted(by.items(), key=lambda kv: -kv[1])[:(200 if os.environ.get('MFVI_PROFILE_FULL') else 24)]:
            c = conv_cost(eng.prog, op, eng.chunk) if op >= 0 else None
            ms1 = ms / n_launch
            rate = "  %6.1f TFLOP/s %6.0f GB/s(alg)" % (c["flops"] / ms1 / 1e9, c["bytes"] / ms1 / 1e6) if c and ps_ in (0, 1, 2) else ""
            sys.stderr.write("op %2d %-10s %8.3f ms %5.1f%%  %s%s\n" % (op, PASS_NAMES[ps_], ms, 100 * ms / tot, c["desc"] if c else ("all layers" if op < 0 else "concat_up"), rate))
        sys.stderr.write("sum of kernel times in one iteration: %.3f ms\n" % tot)

    # forward-only rate (extra information, untimed region)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5):
        eng.forward_only(step=1000 + i)
    torch.cuda.synchronize(); fwd_only = 5 * eng.K_local * world / (time.perf_counter() - t0)

    # ELBO iterations WITH the reference loop's per-iteration bookkeeping (EMA, clips, ring buffers, 2 MSE + 3 PSNR + 3 SSIM:
    # bayesian_optimization.py:1374-1406) — SURVEY 8(d)(ii) asks for the rate with and without it (extra information, untimed region)
    with_book = None
    if world == 1 and cfg["task"] == "den" and args.mode == "engine":
        from mfvi_dip_mia_amd.runner import _Book, phantom
        gt = phantom(S, S, cfg["hp"]["seed"])
        book = _Book(eng, 16, gt, eng.target.cpu().numpy())
        for i in range(3):
            eng.step(after_forward=book.hook(eng, i, eng.chunk))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(3, 13):
            eng.step(after_forward=book.hook(eng, i, eng.chunk))
        torch.cuda.synchronize(); with_book = 10 / (time.perf_counter() - t0)
        book.wait()          # the side-stream bookkeeping has read eng.out before the timed steps overwrite it

    # ---- timed region: exactly --steps iterations, only the dominant kernel carries events ----
    def timed(e, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            e.step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
        return dt
    # N > 1: the exchange after the backward pass, or started under its tail (ElboEngine.set_allreduce_overlap)?  Both schedules give the
    # same update; which one is faster depends on the node's links, so it is timed here (untimed by the contract) like a kernel tiling.
    overlap = None
    if world > 1 and args.mode == "engine":
        mode = os.environ.get("MFVI_AR_OVERLAP", "auto")
        if mode == "auto":
            ab = {}
            for ov in (False, True, False, True):
                eng.set_allreduce_overlap(ov)
                for _ in range(3):
                    eng.step()
                ab[ov] = min(ab.get(ov, 1e9), timed(eng, 10) / 10)      # (max over ranks: every rank takes the same decision)
            use = ab[True] < ab[False]
            overlap = {"used": bool(use), "chosen": "timed", "ms_per_step_after_backward": 1e3 * ab[False], "ms_per_step_overlapped": 1e3 * ab[True]}
        else:
            use = mode == "1"
            overlap = {"used": bool(use), "chosen": "MFVI_AR_OVERLAP=" + mode}
        eng.set_allreduce_overlap(use)
        if use:
            overlap["split"] = {"first_op": eng._ov["op"], "tail_floats": 2 * (eng.n_vi - eng._ov["off"]), "head_floats": int(eng._ov["head"].numel())}
        eng._ar_events = []
        for _ in range(2):
            eng.step()
    graph_info = None
    if args.graph:
        if world > 1 or args.mode != "engine":
            sys.exit("--graph: the fused engine on one GPU")
        eng.enable_graph(warmup=3)
        for _ in range(3):
            eng.step()
        graph_info = {"captured": True, "what": "one ELBO iteration (both streams of the plan) as ONE hipGraphLaunch; RNG step counter and Adam's update count read from device memory"}
    eng.plan.profile(0 if args.graph else 2, dom_op, dom_pass)      # (a replayed graph carries no per-kernel events)
    dt = timed(eng, args.steps)
    recs = eng.plan.profile_read()
    eng.plan.profile(0)
    nll, kl, loss = eng.losses()

    # second curve of a multi-GPU weak run: the config's K split over the ranks (untimed by the contract's value, reported beside it)
    strong_extra = None
    if world > 1 and not strong and k_strong % world == 0:
        eng2 = make_engine(cfg, k_strong, rank, world, torch)
        eng2.set_allreduce_overlap(bool(overlap and overlap["used"]))
        for _ in range(max(args.warmup, 6)):      # a collective's first calls on a new buffer set up staging / channels (gloo: ~0.25 s once)
            eng2.step()
        dt2 = timed(eng2, args.steps)
        strong_extra = dict(k_total=k_strong, k_per_rank=k_strong // world, ms_per_step=1e3 * dt2 / args.steps, value=k_strong * args.steps / dt2,
                            unit="MC-forward-passes/s", elbo_iters_per_sec=args.steps / dt2)
        del eng2

    if rank == 0:
        kms = [ms for _, _, ms in recs]
        avg_ms = sum(kms) / max(len(kms), 1) if kms else by[(dom_op, dom_pass)] / n_launch      # --graph: the instrumented iteration's duration
        cost = conv_cost(eng.prog, dom_op, eng.chunk)
        ai = cost["flops"] / cost["bytes"]
        ridge = F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        if ai >= ridge:      # fp32 arithmetic: above the ridge the matrix/vector fp32 rate bounds the kernel
            roof = dict(bound="mfma", achieved=cost["flops"] / (avg_ms * 1e-3) / 1e12, peak=F32_PEAK_TFLOPS, unit="TFLOP/s")
        else:
            roof = dict(bound="hbm", achieved=cost["bytes"] / (avg_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["traffic"] = None
        # the same kernel timed alone (untimed instrumented iteration, side stream off): what the kernel achieves when it has the chip
        iso_ms = by[(dom_op, dom_pass)] / n_launch
        roof["alone"] = dict(avg_launch_ms=iso_ms, achieved=(cost["flops"] / 1e12 if roof["bound"] == "mfma" else cost["bytes"] / 1e9) / (iso_ms * 1e-3),
                             frac=(cost["flops"] / 1e12 if roof["bound"] == "mfma" else cost["bytes"] / 1e9) / (iso_ms * 1e-3) / roof["peak"])
        if world > 1 and backend != "nccl":      # the one-GPU rehearsal (MFVI_BENCH_BACKEND=gloo): the ranks share the chip, no kernel runs alone
            roof["alone"] = None
        roof["note"] = ("achieved/frac: launch duration inside the timed region, where the backward-weight kernels run on the plan's low-priority side "
                        "stream and share the chip with the backward-data / concat chain of the caller's stream (the overlap shortens the "
                        "iteration and stretches the individual launches); 'alone': the same kernel in untimed instrumented iterations (median of five), side "
                        "stream off, chip to itself.  algorithmic_bytes counts the layer's input, output, mu and rho once; the sampled-weight slab "
                        "the kernel actually reads its weights from ([K][n_vi] floats written once per pass by sample_weights_kernel, DESIGN.md §5) "
                        "adds slab_bytes to what crosses HBM / L2")
        o = eng.prog.ops[dom_op]
        nw = eng.prog.tensors[o["out"]]["C"] * eng.prog.tensors[o["in0"]]["C"] * o["ksize"] ** 2 + eng.prog.tensors[o["out"]]["C"]
        roof.update(kernel="%s of op %d: %s, %d samples/launch" % (PASS_NAMES[dom_pass], dom_op, cost["desc"], eng.chunk),
                    avg_launch_ms=avg_ms, launches=len(kms), algorithmic_bytes=cost["bytes"], algorithmic_flops=cost["flops"], slab_bytes=4 * nw * eng.chunk,
                    hbm_gbs_algorithmic=cost["bytes"] / (avg_ms * 1e-3) / 1e9, hbm_frac_algorithmic=cost["bytes"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
        roof["traffic"] = traffic_lookup("%s:%s" % (PASS_NAMES[dom_pass], cost["desc"]))     # None unless measured on THIS build of the library
        ic = iteration_cost(eng.prog, eng.K_local)
        sec = dt / args.steps
        roof["iteration"] = dict(algorithmic_flops=ic["flops"], algorithmic_bytes=ic["bytes"], tflops=ic["flops"] / sec / 1e12,
                                 frac_of_f32_mfma_peak=ic["flops"] / sec / 1e12 / F32_PEAK_TFLOPS, hbm_gbs=ic["bytes"] / sec / 1e9,
                                 frac_of_hbm_peak=ic["bytes"] / sec / 1e9 / HBM_PEAK_GBS,
                                 note="whole ELBO iteration per GPU: conv FLOPs / bytes of 1 forward + 2 backward passes over K samples, KL, Adam")
        total_samples = eng.K_local * world * args.steps
        res = {
            "metric": "MC-forward-passes/sec (each inside a full ELBO iteration: fwd+data term+bwd+KL+Adam), %dx%d skip MFVI %s" % (S, S, cfg["task"]),
            "value": total_samples / dt, "unit": "MC-forward-passes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32" if cfg.get("param_dtype", "f32") == "f32" else "f32 arithmetic on bf16-stored mu/rho", "data": "synthetic",
            "config": {"workload": cfg["what"] + ("" if cfg["k"] == CONFIGS[args.config]["k"] else " [--k %d]" % cfg["k"]), "name": args.config,
                       "mc_samples_per_iteration": eng.K_local * world, "mc_samples_per_gpu": eng.K_local,
                       "parallelism": "mc-sample sharding x%d, 1 all-reduce/iter" % world,
                       "mode": "engine: fused runner (ElboEngine), no autograd" if args.mode == "engine" else
                               "dropin: get_net + MeanFieldVI%s + gaussian_nll + net.kl() + loss.backward() + torch.optim.AdamW (INTEGRATION.md loop)" % ("(flat_parameters=True)" if args.flat_parameters else "")},
            "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": backend if world > 1 else None,
            "elbo_iters_per_sec": args.steps / dt, "elbo_iters_per_sec_with_bookkeeping": with_book, "fwd_only_mc_passes_per_sec": fwd_only,
            "final_loss": loss, "final_nll": nll, "final_kl": kl,
            "roofline": roof,
        }
        # convolutions that ran on the bf16 matrix instruction with three-way split fp32 operands (same fp32 results: DESIGN.md section 1)
        try:
            from mfvi_dip_mia_amd import _lib as L_
            fam = lambda i, w: L_.lib().mfvi_plan_last_kernel(eng.plan.handle, i, w)
            x6 = []
            for i, o_ in enumerate(eng.prog.ops):
                if o_["type"] != 1:
                    continue
                for which, ps_, n_ in ((0, 0, "fwd"), (1, 2, "bwd_data"), (2, 1, "bwd_weight")):       # (family query index, profile pass id)
                    if fam(i, which) != 3:
                        continue
                    c = conv_cost(eng.prog, i, eng.chunk)
                    e_ = {"kernel": "%s:%s" % (n_, c["desc"])}
                    if (i, ps_) in by:       # duration alone on the chip, from the untimed instrumented iteration (side stream off)
                        ms1 = by[(i, ps_)] / n_launch
                        tf = c["flops"] / ms1 / 1e9
                        e_.update(alone_ms=ms1, tflops=tf, frac_of_f32_mfma_peak=tf / F32_PEAK_TFLOPS, frac_of_bf16_peak_over_6=tf / (BF16_PEAK_TFLOPS / 6.0))
                    x6.append(e_)
            res["bf16x6_kernels"] = x6
        except Exception:
            res["bf16x6_kernels"] = None
        if graph_info:
            res["graph"] = graph_info
        if strong_extra:
            res["strong_scaling"] = strong_extra
        if world > 1:
            res["allreduce_ms"] = eng.allreduce_ms()        # mean time of the exchange on the caller's stream (HIP events); overlapped: what is left exposed
            res["allreduce_overlap"] = overlap
        if cfg["task"] == "den" and S == 256:
            res["reference_cpu_probe"] = REFERENCE_CPU_PROBE
        k_local = eng.K_local
        if world == 1 and not args.no_gpu_baseline:
            try:
                del eng
                torch.cuda.empty_cache()
                res["unfused_gpu_baseline"] = unfused_gpu_baseline(cfg, K_total, torch)
                res["unfused_gpu_baseline"]["speedup_of_this_path"] = res["unfused_gpu_baseline"]["ms_per_iteration"] / res["ms_per_step"]
            except Exception as ex:      # a baseline leg must never take the benchmark line down
                res["unfused_gpu_baseline"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if args.mode == "dropin":
            res["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(cfg, min(cfg["cpu_samples"], 8 * max(1, k_local)))
        elif world == 1 and not args.no_cpu_baseline and cfg["cpu_samples"]:
            res["cpu_baseline"] = cpu_baseline(cfg, cfg["cpu_samples"])
        elif world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = None       # the oracle has no 5x5 / no-skip net driver: the inpainting variant is pinned by reference goldens only
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
