"""Task runners: the reference's run_{den,sr,ct,inp}_mfvi loops (bayesian_optimization.py:1240-1444, 2048-2263, 442-648, 2892-3114)
and their non-Bayesian siblings run_{den,sr,ct,inp}_{dip,mcd,sgld} (:261-439, 651-1237, 1447-2045, 2266-2692, 3117-3544)
on the fused engine, producing the same artefacts (`save.npz` with the dict-of-'mfvi' object arrays that
eval_denoising.ipynb / eval_sr.ipynb / eval_ct.ipynb read, `locals.txt`).

MI355X-first differences (none changes a stored number's meaning):
  * the whole per-iteration bookkeeping (EMA, clip, ring buffers, 2 MSE + 3 PSNR + 3 SSIM) runs in HIP kernels that
    write into device arrays; the host reads them once at the end instead of ~10 `.item()` syncs per iteration;
  * K Monte-Carlo samples per iteration (K = 1 reproduces the reference); `out` in the bookkeeping is the sample mean;
  * images come from the caller (array / .npy / image file); the reference's data/ folder is git-ignored upstream, so a
    seeded synthetic phantom is available (`img='phantom'`).
CLI:  python -m mfvi_dip_mia_amd.runner --task denoising --bayes mfvi --config configs/mfvi_den.json [--img phantom --imsize 256 --k 16]
"""
import argparse
import json
import os
import time

import numpy as np

from . import _lib as L
from . import artifacts as A
from .engine import ElboEngine, SiblingEngine

MC_ITER = 25            # ring-buffer length (bayesian_optimization.py:1314)
EXP_WEIGHT = 0.99       # EMA weight (:1292)


def phantom(H, W, seed):
    """Seeded synthetic ground truth in [0,1]: soft ellipses, two hard bars, a faint texture."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    img = np.zeros((H, W))
    for _ in range(6):
        cx, cy = rng.uniform(-0.6, 0.6, 2); ax, ay = rng.uniform(0.1, 0.5, 2); th = rng.uniform(0, np.pi); amp = rng.uniform(0.2, 0.6)
        xr = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); yr = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        img += amp / (1.0 + np.exp(np.minimum(((xr / ax) ** 2 + (yr / ay) ** 2 - 1.0) * 8.0, 60.0)))
    img[int(0.2 * H):int(0.25 * H), int(0.1 * W):int(0.9 * W)] += 0.3
    img[int(0.1 * H):int(0.9 * H), int(0.7 * W):int(0.74 * W)] += 0.25
    img += 0.03 * np.sin(9 * xx) * np.cos(7 * yy)
    img -= img.min(); img /= img.max()
    return img.astype(np.float32)


def _load_image(img, imsize, seed):
    if isinstance(img, np.ndarray):
        a = img.astype(np.float32)
    elif img == "phantom":
        a = phantom(imsize[0], imsize[1], seed)
    elif isinstance(img, str) and img.endswith(".npy"):
        a = np.load(img).astype(np.float32)
    elif isinstance(img, str):
        from PIL import Image
        a = np.asarray(Image.open(img).convert("L"), np.float32) / 255.0       # utils/common_utils.py:179-191
    else:
        raise FileNotFoundError("img=%r: the reference's data/ images are not distributed (git-ignored upstream); pass an array, "
                                "a path, or img='phantom'" % (img,))
    a = np.squeeze(a)
    if a.ndim != 2:
        raise ValueError("expected one grayscale image, got shape %s" % (a.shape,))
    h, w = a.shape
    return np.ascontiguousarray(a[:h - h % 32, :w - w % 32])                     # crop to a multiple of 32 (get_image: d=32)


class _AsyncBook:
    """The per-iteration bookkeeping only READS the iteration's forward output, so it runs on a second stream beside the backward pass:
    `eng.step(after_forward=book.hook(eng, i, n))` enqueues it behind the forward, `book.wait()` (called by the next hook and by snapshot /
    results) orders the caller's stream behind it before `eng.out` is written again."""
    _side = None; _done = None

    def hook(self, eng, i, n=None):
        """-> the callable for eng.step(after_forward=...).  The engine passes the sample count of its LAST launch (eng.out holds only
        those; with K_local not a multiple of the launch size the last launch is partial), which overrides `n`."""
        def run(n_last=None):
            n_ = n if n_last is None else n_last
            t = self.t
            if self._side is None:
                self._side = t.cuda.Stream(); self._done = t.cuda.Event()
            main = t.cuda.current_stream()
            ready = t.cuda.Event(); ready.record(main)
            with t.cuda.stream(self._side):
                self._side.wait_event(ready)
                self.iteration(eng, i, n_)
                self._done.record(self._side)
        self.wait()          # the previous iteration's bookkeeping has read eng.out before this iteration's forward overwrites it
        return run

    def wait(self):
        if self._done is not None:
            self.t.cuda.current_stream().wait_event(self._done)


class _Book(_AsyncBook):
    """Device-side bookkeeping state of one den / sr / ct fit (bayesian_optimization.py:1374-1416, :2190-2236, :584-626).
    Column 0 of mse_noisy / psnrs / ssims is the 'corrupted' reference of the task: the noisy image (den), the low-resolution image
    against the [::f, ::f] projection of the output (sr: downsampler(out_avg) / out_lr, :2203-2218), the ground truth (ct)."""

    def __init__(self, eng, num_iter, gt, noisy_or_none, task="den", factor=4):
        import torch
        self.t = torch
        H, W, C = eng.H, eng.W, eng.out.shape[1]
        dev = "cuda"
        self.C, self.H, self.W, self.task, self.f = C, H, W, task, int(factor)
        self.ema = torch.zeros((C, H, W), device=dev)
        self.out_clip = torch.zeros((H, W), device=dev); self.ale_clip = torch.zeros((H, W), device=dev); self.avg_clip = torch.zeros((H, W), device=dev)
        self.ring_epi = torch.zeros((MC_ITER, H, W), device=dev); self.ring_ale = torch.zeros((MC_ITER, H, W), device=dev)
        self.metrics = torch.zeros((num_iter, 8), dtype=torch.float64, device=dev)     # mse_noisy, mse_gt, 3 psnr-mse, 3 ssim sums
        self.gt = torch.from_numpy(np.ascontiguousarray(gt, np.float32)).to(dev)
        self.noisy = None if noisy_or_none is None else torch.from_numpy(np.ascontiguousarray(noisy_or_none, np.float32)).to(dev)
        self.var = torch.zeros((H, W), device=dev); self.ale_mean = torch.zeros((H, W), device=dev)
        if task == "sr":      # img_small_torch = downsampler(img_torch) and the projections of out_avg / out (:2101, :2203, :2207)
            h, w = H // self.f, W // self.f
            self.gt_lr = self.gt[::self.f, ::self.f].contiguous()
            self.avg_lr = torch.zeros((h, w), device=dev); self.out_lr_clip = torch.zeros((h, w), device=dev)

    def iteration(self, eng, i, n, lr_view=None):
        lib, sp, p = L.lib(), L.stream_ptr(), L.ptr
        slot = i % MC_ITER
        L.check(lib.mfvi_bookkeep(p(eng.out), n, self.C, self.H, self.W, p(self.ema), EXP_WEIGHT, int(i == 0), p(self.out_clip), p(self.ale_clip),
                                  p(self.avg_clip), p(self.ring_epi[slot]), p(self.ring_ale[slot]) if self.C > 1 else None, sp))
        m = self.metrics[i]
        hw = self.H * self.W
        avg0 = self.ema[0]
        if self.task == "sr":
            h, w = self.gt_lr.shape
            L.check(lib.mfvi_decimate(p(avg0), self.H, self.W, self.f, p(self.avg_lr), sp))
            L.check(lib.mfvi_decimate(p(self.out_clip), self.H, self.W, self.f, p(self.out_lr_clip), sp))     # clip and [::f, ::f] commute
            L.check(lib.mfvi_sq_err_sum(p(self.avg_lr), p(self.gt_lr), h * w, p(m[0:]), sp))         # mse(downsampler(out_avg)[:, :1], img_small)  :2203
            L.check(lib.mfvi_sq_err_sum(p(self.gt_lr), p(self.out_lr_clip), h * w, p(m[2:]), sp))    # psnr_lr                                       :2214
            L.check(lib.mfvi_ssim_sum(p(self.gt_lr), p(self.out_lr_clip), h, w, p(m[5:]), sp))       # ssim_lr                                       :2217
        else:
            ref_noisy = self.noisy if self.noisy is not None else self.gt                            # ct: both columns against the ground truth (:594-606)
            L.check(lib.mfvi_sq_err_sum(p(avg0), p(ref_noisy), hw, p(m[0:]), sp))                    # mse(out_avg[:, :1], noisy)  :1389
            L.check(lib.mfvi_sq_err_sum(p(ref_noisy), p(self.out_clip), hw, p(m[2:]), sp))           # psnr_corrupted              :1398
            L.check(lib.mfvi_ssim_sum(p(ref_noisy), p(self.out_clip), self.H, self.W, p(m[5:]), sp))
        L.check(lib.mfvi_sq_err_sum(p(avg0), p(self.gt), hw, p(m[1:]), sp))              # mse(out_avg[:, :1], gt)     :1390
        L.check(lib.mfvi_sq_err_sum(p(self.gt), p(self.out_clip), hw, p(m[3:]), sp))     # psnr_gt
        L.check(lib.mfvi_sq_err_sum(p(self.gt), p(self.avg_clip), hw, p(m[4:]), sp))     # psnr_gt_sm
        L.check(lib.mfvi_ssim_sum(p(self.gt), p(self.out_clip), self.H, self.W, p(m[6:]), sp))
        L.check(lib.mfvi_ssim_sum(p(self.gt), p(self.avg_clip), self.H, self.W, p(m[7:]), sp))

    def snapshot(self):
        self.wait()
        lib, sp, p = L.lib(), L.stream_ptr(), L.ptr
        L.check(lib.mfvi_ring_stats(p(self.ring_epi), MC_ITER, self.H, self.W, p(self.var), None, sp))
        if self.C > 1:
            L.check(lib.mfvi_ring_stats(p(self.ring_ale), MC_ITER, self.H, self.W, None, p(self.ale_mean), sp))
        return self.var.cpu().numpy(), self.ale_mean.cpu().numpy(), self.avg_clip.cpu().numpy()

    def results(self):
        self.wait()
        m = self.metrics.cpu().numpy(); hw = float(self.H * self.W)
        n = np.full(8, hw)
        if self.task == "sr":
            n[[0, 2, 5]] = float(self.gt_lr.numel())
        m = m / n
        with np.errstate(divide="ignore"):
            psnrs = 10.0 * np.log10(1.0 / m[:, 2:5])
        return m[:, 0], m[:, 1], psnrs, m[:, 5:8]


class _BookInp(_AsyncBook):
    """Device-side bookkeeping of the inpainting runner (bayesian_optimization.py:3039-3090): sigmoid colour channels, masked PSNR / SSIM."""

    def __init__(self, eng, num_iter, img, mask):
        import torch
        self.t = torch
        dev = "cuda"
        H, W = eng.H, eng.W
        self.H, self.W = H, W
        self.img = torch.from_numpy(np.ascontiguousarray(img, np.float32)).to(dev)
        self.mask = torch.from_numpy(np.ascontiguousarray(mask, np.float32)).to(dev).round()
        self.mc = self.mask.shape[0]
        self.ema = torch.zeros((4, H, W), device=dev)
        self.out_clip = torch.zeros((3, H, W), device=dev); self.avg_clip = torch.zeros_like(self.out_clip); self.ale_clip = torch.zeros((H, W), device=dev)
        self.img_m = torch.zeros_like(self.out_clip); self.out_m = torch.zeros_like(self.out_clip); self.avg_m = torch.zeros_like(self.out_clip)
        self.ring_epi = torch.zeros((MC_ITER, 3, H, W), device=dev); self.ring_ale = torch.zeros((MC_ITER, H, W), device=dev)
        self.metrics = torch.zeros((num_iter, 10, 3), dtype=torch.float64, device=dev)      # per channel: mse, 3 psnr-mse, 3 ssim sums (+ spare)
        self.var = torch.zeros((3, H, W), device=dev); self.ale_mean = torch.zeros((H, W), device=dev)

    def iteration(self, eng, i, n):
        lib, p, sp = L.lib(), L.ptr, L.stream_ptr()
        H, W = self.H, self.W; HW = H * W; slot = i % MC_ITER
        L.check(lib.mfvi_bookkeep_inpainting(p(eng.out), n, H, W, p(self.img), p(self.mask), self.mc, p(self.ema), EXP_WEIGHT, int(i == 0),
                                             p(self.out_clip), p(self.ale_clip), p(self.avg_clip), p(self.img_m), p(self.out_m), p(self.avg_m),
                                             p(self.ring_epi[slot]), p(self.ring_ale[slot]), sp))
        m = self.metrics[i]
        for c in range(3):                                            # channel-wise sums; means over the 3 channels taken on the host
            L.check(lib.mfvi_sq_err_sum(p(self.ema[c]), p(self.img[c]), HW, p(m[0, c:]), sp))            # mse(out_avg[:, :3], img)        :3051-3052
            L.check(lib.mfvi_sq_err_sum(p(self.img[c]), p(self.out_clip[c]), HW, p(m[1, c:]), sp))       # psnr_corrupted                  :3062
            L.check(lib.mfvi_sq_err_sum(p(self.img_m[c]), p(self.out_m[c]), HW, p(m[2, c:]), sp))        # psnr_gt (masked)                :3063
            L.check(lib.mfvi_sq_err_sum(p(self.img_m[c]), p(self.avg_m[c]), HW, p(m[3, c:]), sp))        # psnr_gt_sm
            L.check(lib.mfvi_ssim_sum(p(self.img[c]), p(self.out_clip[c]), H, W, p(m[4, c:]), sp))
            L.check(lib.mfvi_ssim_sum(p(self.img_m[c]), p(self.out_m[c]), H, W, p(m[5, c:]), sp))
            L.check(lib.mfvi_ssim_sum(p(self.img_m[c]), p(self.avg_m[c]), H, W, p(m[6, c:]), sp))

    def snapshot(self):
        self.wait()
        lib, p, sp = L.lib(), L.ptr, L.stream_ptr()
        for c in range(3):
            L.check(lib.mfvi_ring_stats(p(self.ring_epi[:, c].contiguous()), MC_ITER, self.H, self.W, p(self.var[c]), None, sp))
        L.check(lib.mfvi_ring_stats(p(self.ring_ale), MC_ITER, self.H, self.W, None, p(self.ale_mean), sp))
        return self.var.cpu().numpy(), self.ale_mean.cpu().numpy(), self.avg_clip.cpu().numpy()

    def results(self):
        self.wait()
        mt = self.metrics.cpu().numpy().mean(axis=2) / (self.H * self.W)      # mean over the colour channels == mean over all 3*H*W elements
        with np.errstate(divide="ignore"):
            psnrs = 10.0 * np.log10(1.0 / mt[:, 1:4])
        return mt[:, 0], mt[:, 0], psnrs, mt[:, 4:7]                          # the reference computes both MSEs against img_torch (:3051-3052)


def _make_engine(method, H, W, task, K, input_depth, temp, sigma, lr, seed, net_kwargs, sib, param_dtype="f32", **kw):
    if param_dtype != "f32":
        kw["param_dtype"] = param_dtype             # bf16 mu / rho (BASELINE configs[4]); the MFVI engines only
    if method == "mfvi":
        return ElboEngine(H, W, task=task, K=K, input_depth=input_depth, temp=temp, sigma=sigma, lr=lr, seed=seed, net_kwargs=net_kwargs, **kw)
    return SiblingEngine(H, W, method=method, task=task, K=K, input_depth=input_depth, lr=lr, seed=seed, net_kwargs=net_kwargs, **sib, **kw)


def _run(task, img, imsize, p_sigma, num_iter, lr, temp, sigma, input_depth, seed, show_every, plot, save, save_path, K, factor=4,
         theta_step=4.0, verbose=False, net_kwargs=None, method="mfvi", weight_decay=0.0, dropout_p=0.3, gamma=0.996, param_dtype="f32", **unused):
    import torch
    sib = dict(weight_decay=weight_decay, dropout_p=dropout_p, gamma=gamma)
    timestamp = str(time.time())
    run_dir = os.path.join(save_path, timestamp)
    if save:
        os.makedirs(run_dir, exist_ok=False)
        with open(os.path.join(run_dir, "locals.txt"), "w") as f:
            hyper = dict(temp=temp, sigma=sigma) if method == "mfvi" else {"dip": {}, "mcd": dict(dropout_p=dropout_p, weight_decay=weight_decay),
                                                                          "sgld": dict(gamma=gamma, weight_decay=weight_decay)}[method]
            for key, val in dict(task=task, method=method, img=img if not isinstance(img, np.ndarray) else "<array>", imsize=imsize, p_sigma=p_sigma,
                                 num_iter=num_iter, lr=lr, input_depth=input_depth, seed=seed, show_every=show_every,
                                 K=K, save_path=save_path, **hyper).items():
                print(key, "=", val, file=f)
    img_np = _load_image(img, imsize, seed)
    H, W = img_np.shape
    num_iter += 1                                                    # bayesian_optimization.py:1291
    extra = {}
    noisy = None
    if task == "den":
        rng = np.random.default_rng(seed + 1)
        noisy = np.clip(img_np + rng.normal(scale=p_sigma, size=img_np.shape), 0, 1).astype(np.float32)      # denoising_utils.py:11
        target = noisy
        eng = _make_engine(method, H, W, "den", K, input_depth, temp, sigma, lr, seed, net_kwargs, sib, param_dtype=param_dtype)
    elif task == "sr":
        target = np.ascontiguousarray(img_np[::factor, ::factor])    # nearest /factor decimation (:2095-2099)
        noisy = None
        eng = _make_engine(method, H, W, "sr", K, input_depth, temp, sigma, lr, seed, net_kwargs, sib, sr_factor=factor, param_dtype=param_dtype)
        extra["img_lr"] = target
    else:
        theta = np.arange(0, 180.0, theta_step, dtype=np.float32)     # :545
        eng = _make_engine(method, H, W, "ct", K, input_depth, temp, sigma, lr, seed, net_kwargs, sib, theta_deg=theta.tolist())
        sino = torch.empty((len(theta), W), device="cuda")
        gt_d = torch.from_numpy(img_np).cuda()
        L.check(L.lib().mfvi_radon_forward(L.ptr(gt_d), L.ptr(eng.theta), 1, H, W, len(theta), L.ptr(sino), L.stream_ptr()))
        target = sino
        extra["img_radon"] = sino.cpu().numpy()[None, None]
    eng.set_target(torch.from_numpy(target) if isinstance(target, np.ndarray) else target)
    book = _Book(eng, num_iter, img_np, noisy, task=task, factor=factor)
    n_snap = num_iter // show_every + 1
    recons = np.zeros((n_snap, 1, H, W)); uncerts_epi = np.zeros((n_snap, 1, H, W)); uncerts_ale = np.zeros((n_snap, 1, H, W))
    t0 = time.perf_counter()
    for i in range(num_iter):
        # one fused ELBO iteration (a non-finite loss skips the update on the device: engine.step); the bookkeeping of :1374-1406 reads only the
        # iteration's forward output and runs on a second stream beside the backward pass
        eng.step(after_forward=book.hook(eng, i, eng.chunk))
        if i % show_every == 0:
            var, ale, recon = book.snapshot()
            uncerts_epi[i // show_every, 0] = var; uncerts_ale[i // show_every, 0] = ale; recons[i // show_every, 0] = recon
            if verbose:
                nll, kl, loss = eng.losses()
                print("iter %6d  loss %.5f  nll %.5f  kl %.4e  (%.1f it/s)" % (i, loss, nll, kl, (i + 1) / (time.perf_counter() - t0)))
            if plot and save:                                        # loss_<method>.png, out_avg.png, out_var.png, out_ale.png (:1418-1422)
                mn, mg, ps_, _ = book.results()
                A.snapshot_pngs(run_dir, method, i, mn, mg, ps_, recon[None], None if method == "dip" else var[None],
                                None if (method == "dip" or task == "ct") else ale[None])
    torch.cuda.synchronize()
    mse_noisy, mse_gt, psnrs, ssims = book.results()
    if save:
        # first two keys as the reference writes them per task: den img_gt / img_noisy (:1438), sr img_hr / img_lr (:2258), ct img_gt / img_radon (:643)
        # with the reference's shapes: get_image() arrays are (1, H, W), img_lr is squeezed, the CT tensors keep (1, 1, ., .)
        head = dict(den=(img_np[None], None if noisy is None else noisy[None]), sr=(img_np[None], extra.get("img_lr")),
                    ct=(img_np[None, None], extra.get("img_radon")))[task]
        A.save_npz(run_dir, task, method, head, mse_noisy, mse_gt, recons, uncerts_epi, uncerts_ale, psnrs, ssims)
        with open(os.path.join(run_dir, "locals.txt"), "a") as f:
            print("max psnr_gt_sm = %.4f, max ssim_gt_sm = %.4f" % (np.nanmax(psnrs[:, 2]), np.nanmax(ssims[:, 2])), file=f)
            if plot:                                                  # mse_noisy / mse_gt / psnrs / ssims .png + the maxima lines (:201-258, :1431-1433)
                A.plot_results({method: mse_noisy}, {method: mse_gt}, {method: psnrs}, {method: ssims}, run_dir, f)
        if plot and task == "sr":
            A.sr_input_png(run_dir, img_np[None], extra["img_lr"], factor)
    return dict(psnr=float(psnrs[-1, 2]), run_dir=run_dir if save else None, psnrs=psnrs, ssims=ssims, mse_noisy=mse_noisy, mse_gt=mse_gt,
                recons=recons, uncerts=uncerts_epi, uncerts_ale=uncerts_ale, seconds=time.perf_counter() - t0, engine=eng)


def run_den_mfvi(img="phantom", imsize=(256, 256), p_sigma=0.1, num_iter=5000, lr=3e-4, temp=4e-6, sigma=0.01, input_depth=16, seed=42,
                 show_every=100, plot=False, save=True, save_path="../logs", K=1, **kw):
    """bayesian_optimization.py:1240-1444.  Returns the dict of results; ['psnr'] is the reference's return value."""
    return _run("den", img, imsize, p_sigma, num_iter, lr, temp, sigma, input_depth, seed, show_every, plot, save, save_path, K, **kw)


def run_sr_mfvi(img="phantom", imsize=(512, 512), factor=4, num_iter=5000, lr=3e-4, temp=4e-6, sigma=0.01, input_depth=32, seed=42,
                show_every=100, plot=False, save=True, save_path="../logs", K=1, **kw):
    """bayesian_optimization.py:2048-2263 (p_sigma is swallowed by **kwargs there too)."""
    kw.pop("p_sigma", None)
    return _run("sr", img, imsize, 0.0, num_iter, lr, temp, sigma, input_depth, seed, show_every, plot, save, save_path, K, factor=factor, **kw)


def run_ct_mfvi(img="phantom", imsize=(256, 256), num_iter=5000, lr=3e-4, temp=4e-6, sigma=0.01, input_depth=16, seed=42,
                show_every=100, plot=False, save=True, save_path="../logs", K=1, **kw):
    """bayesian_optimization.py:442-648."""
    kw.pop("p_sigma", None)
    return _run("ct", img, imsize, 0.0, num_iter, lr, temp, sigma, input_depth, seed, show_every, plot, save, save_path, K, **kw)


def _sibling(task, method, defaults):
    """run_<task>_<method> with the reference's keyword names and defaults; everything else as the MFVI runner of the task."""
    def run(img="phantom", imsize=defaults.get("imsize", (256, 256)), num_iter=5000, lr=defaults.get("lr", 3e-4), input_depth=defaults.get("input_depth", 16),
            seed=42, show_every=100, plot=False, save=True, save_path="../logs", K=1, p_sigma=0.1, **kw):
        hyper = {k: kw.pop(k, v) for k, v in defaults.items() if k in ("weight_decay", "dropout_p", "gamma")}
        kw.pop("temp", None); kw.pop("sigma", None)
        if task == "inp":
            hyper.setdefault("weight_decay", 0.0)                     # run_inp_dip: weight_decay = 0 (bayesian_optimization.py:2756)
            return run_inp_mfvi(img=img, imsize=imsize, num_iter=num_iter, lr=lr, input_depth=input_depth, seed=seed, show_every=show_every,
                                plot=plot, save=save, save_path=save_path, K=K, method=method, **hyper, **kw)
        return _run(task, img, imsize, p_sigma if task == "den" else 0.0, num_iter, lr, 0.0, 0.0, input_depth, seed, show_every, plot, save, save_path, K,
                    method=method, **hyper, **kw)
    run.__name__ = "run_%s_%s" % (task, method)
    run.__doc__ = "bayesian_optimization.py run_%s_%s; returns the dict of results (['psnr'] is the reference's return value)." % (task, method)
    return run


# keyword defaults of the reference's signatures (bayesian_optimization.py:261-280, 651-670, 862-881, 1064-1083, 1447-1466, 1658-1677,
# 1863-1883, 2266-2286, 2484-2504, 3117-3135, 3328-3346)
run_den_dip = _sibling("den", "dip", dict())
run_den_mcd = _sibling("den", "mcd", dict(dropout_p=0.3, weight_decay=3e-4))
run_den_sgld = _sibling("den", "sgld", dict(gamma=0.996, weight_decay=5e-8))
run_sr_dip = _sibling("sr", "dip", dict(imsize=(512, 512), input_depth=32))
run_sr_mcd = _sibling("sr", "mcd", dict(imsize=(512, 512), input_depth=32, dropout_p=0.2, weight_decay=1e-4))
run_sr_sgld = _sibling("sr", "sgld", dict(imsize=(512, 512), input_depth=32, gamma=0.996, weight_decay=1e-4))
run_ct_dip = _sibling("ct", "dip", dict())
run_ct_mcd = _sibling("ct", "mcd", dict(dropout_p=0.3, weight_decay=3e-4))
run_ct_sgld = _sibling("ct", "sgld", dict(gamma=0.996, weight_decay=5e-8))


def run_inp_mfvi(img="phantom", mask=None, imsize=(256, 256), num_iter=5000, lr=2e-3, temp=4e-6, sigma=0.01, input_depth=32, seed=42,
                 show_every=100, plot=False, save=True, save_path="../logs", K=1, net_kwargs=None, verbose=False, method="mfvi",
                 weight_decay=1e-4, dropout_p=0.2, gamma=0.996, **unused):
    """bayesian_optimization.py:2892-3114: inpainting with the 6-scale no-skip net (5x5 down filters, nearest up-sampling), sigmoid on
    the colour channels, masked Gaussian NLL.  img: (3, H, W) array in [0, 1] or 'phantom' (three synthetic planes); mask: (1|3, H, W),
    1 = known pixel (the reference ships its masks in data/inpainting/).  save.npz carries the reference's keys for this task
    (img_inpainting, img_mask, mse_corrupted, mse_gt, recons, uncerts, uncerts_ale, psnrs, ssims)."""
    import torch
    timestamp = str(time.time())
    run_dir = os.path.join(save_path, timestamp)
    if save:
        os.makedirs(run_dir, exist_ok=False)
        with open(os.path.join(run_dir, "locals.txt"), "w") as f:
            for key, val in dict(task="inp", imsize=imsize, num_iter=num_iter, lr=lr, temp=temp, sigma=sigma, input_depth=input_depth,
                                 seed=seed, show_every=show_every, K=K, save_path=save_path).items():
                print(key, "=", val, file=f)
    if isinstance(img, np.ndarray):
        img_np = np.ascontiguousarray(img, np.float32)
    else:
        img_np = np.stack([_load_image(img, imsize, seed + c) for c in range(3)]).astype(np.float32)
    _, H, W = img_np.shape
    if mask is None:                                                  # synthetic scratches: a seeded random-walk mask, ~12 % missing
        rng = np.random.default_rng(seed)
        mask = np.ones((1, H, W), np.float32)
        for _ in range(24):
            y, x = rng.integers(0, H), rng.integers(0, W)
            for _ in range(H):
                mask[0, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2] = 0
                y = int(np.clip(y + rng.integers(-1, 2), 0, H - 1)); x = int(np.clip(x + rng.integers(-1, 2), 0, W - 1))
    mask_np = np.ascontiguousarray(mask, np.float32)
    if mask_np.ndim == 2:
        mask_np = mask_np[None]
    num_iter += 1                                                     # bayesian_optimization.py:2935
    eng = _make_engine(method, H, W, "inp", K, input_depth, temp, sigma, lr, seed, net_kwargs, dict(weight_decay=weight_decay, dropout_p=dropout_p, gamma=gamma))
    eng.set_target(torch.from_numpy(img_np), torch.from_numpy(mask_np))
    book = _BookInp(eng, num_iter, img_np, mask_np)
    n_snap = num_iter // show_every + 1
    recons = np.zeros((n_snap, 3, H, W)); uncerts_epi = np.zeros((n_snap, 3, H, W)); uncerts_ale = np.zeros((n_snap, 1, H, W))
    t0 = time.perf_counter()
    for i in range(num_iter):
        eng.step(after_forward=book.hook(eng, i, eng.chunk))
        if i % show_every == 0:
            var, ale, recon = book.snapshot()
            uncerts_epi[i // show_every] = var; uncerts_ale[i // show_every, 0] = ale; recons[i // show_every] = recon
            if verbose:
                nll, kl, loss = eng.losses()
                print("iter %6d  loss %.5f  nll %.5f  kl %.4e  (%.1f it/s)" % (i, loss, nll, kl, (i + 1) / (time.perf_counter() - t0)))
    torch.cuda.synchronize()
    mse_corrupted, mse_gt, psnrs, ssims = book.results()
    if save:
        A.save_npz(run_dir, "inp", method, (img_np, mask_np), mse_corrupted, mse_gt, recons, uncerts_epi, uncerts_ale, psnrs, ssims)
        with open(os.path.join(run_dir, "locals.txt"), "a") as f:
            print("max psnr_gt_sm = %.4f, max ssim_gt_sm = %.4f" % (np.nanmax(psnrs[:, 2]), np.nanmax(ssims[:, 2])), file=f)
            if plot:
                A.plot_results({method: mse_corrupted}, {method: mse_gt}, {method: psnrs}, {method: ssims}, run_dir, f)
        if plot:
            A.snapshot_pngs(run_dir, method, num_iter - 1, mse_corrupted, mse_gt, psnrs, recons[-1], None if method == "dip" else uncerts_epi[-1],
                            None if method == "dip" else uncerts_ale[-1])
    return dict(psnr=float(psnrs[-1, 2]), run_dir=run_dir if save else None, psnrs=psnrs, ssims=ssims, mse_corrupted=mse_corrupted, mse_gt=mse_gt,
                recons=recons, uncerts=uncerts_epi, uncerts_ale=uncerts_ale, seconds=time.perf_counter() - t0, engine=eng)


run_inp_dip = _sibling("inp", "dip", dict(input_depth=32, lr=2e-3))
run_inp_mcd = _sibling("inp", "mcd", dict(input_depth=32, lr=2e-3, dropout_p=0.2, weight_decay=1e-4))
run_inp_sgld = _sibling("inp", "sgld", dict(input_depth=32, lr=2e-3, gamma=0.996, weight_decay=1e-4))

# which two bo_params a method's candidates are (bayesian_optimization.py:3715-3718)
BO_KEYS = {"mfvi": ("temp", "sigma"), "mcd": ("dropout_p", "weight_decay"), "sgld": ("gamma", "weight_decay"), "dip": ()}


def load_config(path, bayes="mfvi", with_devices=False):
    """The reference's JSON schema {bo_params:{<name>:{candidates}, ...}, run_params:{...}} (bayesian_optimization.py:3901-3909 reads it
    through pandas; plain json is equivalent).  Returns ([candidate dicts], run_params[, devices]); `devices` is run_params['devices']
    of the reference's configs (eval_result.py:21-22), e.g. ["cuda:0", ..., "cuda:7"]."""
    cfg = json.load(open(path))
    rp = dict(cfg["run_params"])
    devices = rp.pop("devices", None)
    rp.pop("bo_results_path", None)
    keys = BO_KEYS[bayes]
    if not keys:
        cands = [dict()]
    else:
        a, b = keys
        cands = [{a: x, b: y} for x in cfg["bo_params"][a]["candidates"] for y in cfg["bo_params"][b]["candidates"]]
    return (cands, rp, devices) if with_devices else (cands, rp)


def fit_job(fn_name, **kw):
    """One independent fit in a worker process of the fan-out: run_<task>_<method>(**kw) -> PSNR (the reference's return value)."""
    return globals()[fn_name](**kw)["psnr"]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="denoising", choices=["denoising", "super-resolution", "ct", "inpainting"])
    ap.add_argument("--bayes", default="mfvi", choices=["mfvi", "dip", "mcd", "sgld"])
    ap.add_argument("--config", required=True)
    ap.add_argument("--img", default=None, help="one image, or a comma-separated list: every (image, candidate) pair is one independent fit")
    ap.add_argument("--imsize", type=int, default=None)
    ap.add_argument("--k", type=int, default=1); ap.add_argument("--num-iter", type=int, default=None); ap.add_argument("--save-path", default=None)
    ap.add_argument("--devices", default=None, help="comma-separated devices (cuda:0,cuda:1,...): the independent fits are dealt round-robin to one "
                                                    "fresh worker process per device (default: the config's run_params.devices; absent: this process)")
    ap.add_argument("--param-dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--bo-rounds", type=int, default=0, help="> 0: the reference's Gaussian-process outer loop over the method's two hyper-parameters "
                                                             "(bo_params.<name>.logbounds of the config) for this many rounds, each round's candidates "
                                                             "run as independent fits (bo.py; parity unpinned: gpytorch is not available here)")
    a = ap.parse_args(argv)
    cands, rp, cfg_devices = load_config(a.config, a.bayes, with_devices=True)
    short = {"denoising": "den", "super-resolution": "sr", "ct": "ct", "inpainting": "inp"}[a.task]
    fn_name = "run_%s_%s" % (short, a.bayes)
    fn = globals().get(fn_name)                # f(): bayesian_optimization.py:3709-3724
    if fn is None:
        raise NotImplementedError("run_%s_%s is not built" % (short, a.bayes))
    if a.imsize:
        rp["imsize"] = (a.imsize, a.imsize)
    if a.num_iter is not None:
        rp["num_iter"] = a.num_iter
    if a.save_path:
        rp["save_path"] = a.save_path
    rp["plot"] = False
    if a.param_dtype != "f32":
        rp["param_dtype"] = a.param_dtype
    imgs = a.img.split(",") if a.img else [rp.pop("img", "phantom")]
    rp.pop("img", None)
    jobs = [dict(cand, img=im) for im in imgs for cand in cands]
    devices = a.devices.split(",") if a.devices else cfg_devices
    if a.bo_rounds > 0:
        # bo() of the reference (bayesian_optimization.py:3727-3880): rounds of [fits of the candidates over the devices -> GP -> new candidates]
        from . import bo as _bo
        keys = BO_KEYS[a.bayes]
        if not keys or len(imgs) != 1:
            raise ValueError("--bo-rounds needs a method with two hyper-parameters (mfvi, mcd, sgld) and one image")
        bo_params = {k: json.load(open(a.config))["bo_params"][k] for k in keys}

        def evaluate(cand_list):
            jb = [dict(zip(keys, c), img=imgs[0]) for c in cand_list]
            if devices:
                from .fanout import run_jobs
                res, _ = run_jobs(jb, devices, "mfvi_dip_mia_amd.runner:fit_job", dict(rp, fn_name=fn_name, K=a.k))
                return [(tuple(job[k] for k in keys), y) for _, job, y in res]
            return [(c, fn(K=a.k, verbose=False, **j, **rp)["psnr"]) for c, j in zip(cand_list, jb)]
        return _bo.bo(bo_params, evaluate, n_rounds=a.bo_rounds)
    if devices:
        # independent fits over the node's GPUs (bayesian_optimization.py:3760-3781, eval_result.py:27-53): no per-step communication,
        # one final gather of (candidate, psnr), NaNs dropped
        from .fanout import run_jobs, print_table
        results, dropped = run_jobs(jobs, devices, "mfvi_dip_mia_amd.runner:fit_job", dict(rp, fn_name=fn_name, K=a.k))
        print_table(results, list(BO_KEYS[a.bayes]))
        for i, job, why in dropped:
            print("dropped fit %d %s: %s" % (i, {k: v for k, v in job.items() if k != "img"}, why))
        return results
    out = []
    for job in jobs:
        r = fn(K=a.k, verbose=True, **job, **rp)
        cand = {k: v for k, v in job.items() if k != "img"}
        print("%s -> PSNR %.3f dB in %.1f s (%s)" % (" ".join("%s %.3e" % kv for kv in cand.items()) or "dip", r["psnr"], r["seconds"], r["run_dir"]))
        out.append(r)
    return out


if __name__ == "__main__":
    main()
