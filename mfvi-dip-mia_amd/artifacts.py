"""Host-side artefacts of a fit, as the reference's runners write them (no GPU, no HIP: numpy + PIL + matplotlib only):

  save.npz     dict-of-method object arrays under the reference's per-task keys       bayesian_optimization.py:1436-1440 (den), :2256-2260 (sr),
                                                                                      :641-645 (ct), :3107-3111 (inpainting)
  locals.txt   arguments + the PSNR / SSIM maxima plot_results prints                :1261-1266, :246-258
  PNGs (plot=True)  out_avg / out_var / out_ale per snapshot, loss_<method>.png      :1418-1422, :172-199 (plot_loss)
                    mse_noisy / mse_gt / psnrs / ssims at the end                    :201-258 (plot_results)
                    input.png for SR (ground truth beside the nearest-upsampled observation; PIL instead of cv2)   :2103-2108

`read_like_notebooks` replays the reads eval_denoising.ipynb / eval_sr.ipynb / eval_ct.ipynb / eval_inp.ipynb perform on a save.npz, so a
test can prove a file written here opens in them unchanged."""
import os

import numpy as np

HEAD_KEYS = {"den": ("img_gt", "img_noisy"), "sr": ("img_hr", "img_lr"), "ct": ("img_gt", "img_radon"), "inp": ("img_inpainting", "img_mask")}
MSE_KEY = {"den": "mse_noisy", "sr": "mse_noisy", "ct": "mse_noisy", "inp": "mse_corrupted"}


def np_to_pil(img_np):
    """C x H x W in [0, 1] -> PIL image (utils/common_utils.py:194-206)."""
    from PIL import Image
    ar = np.clip(np.asarray(img_np) * 255, 0, 255).astype(np.uint8)
    ar = ar[0] if ar.shape[0] == 1 else ar.transpose(1, 2, 0)
    return Image.fromarray(ar)


def save_npz(run_dir, task, method, head, mse_corrupted, mse_gt, recons, uncerts_epi, uncerts_ale, psnrs, ssims):
    """head: the two task-specific arrays in the reference's shapes — den (1,H,W),(1,H,W); sr (1,H,W),(h,w); ct (1,1,H,W),(1,1,T,W);
    inp (3,H,W),(1|3,H,W).  DIP runs leave the uncertainty dicts empty like the reference (:1126-1127, :1230-1232)."""
    wrap = lambda a: {method: a}
    unc = (lambda a: {}) if method == "dip" else wrap
    k0, k1 = HEAD_KEYS[task]
    np.savez(os.path.join(run_dir, "save.npz"), **{k0: head[0], k1: head[1], MSE_KEY[task]: wrap(mse_corrupted)}, mse_gt=wrap(mse_gt), recons=wrap(recons),
             uncerts=unc(uncerts_epi), uncerts_ale=unc(uncerts_ale), psnrs=wrap(psnrs), ssims=wrap(ssims))


def write_locals(run_dir, **kw):
    with open(os.path.join(run_dir, "locals.txt"), "w") as f:
        for key, val in kw.items():
            print(key, "=", val, file=f)


def _plt():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


# loss_<method>.png — the figure the reference's loop rewrites every show_every iterations (bayesian_optimization.py:172-199).  Its
# LAYOUT is the artefact contract (people compare these PNGs across methods), so it is written down as data: a left axis with the two
# MSE curves (default colour cycle, y range 0 ... 0.03, grid), a twin right axis with column 2 of the PSNR table (smoothed output vs
# ground truth) in green.
LOSS_FIGURE = dict(left=dict(xlabel="iteration", ylabel="mse", ylim=(0, 0.03)), psnr_column=2, psnr_style="g")


def plot_loss(mse_corrupted, mse_gt, psnrs, it, path, title="MSE", y_label="psnr_gt_sm"):
    """Draw LOSS_FIGURE for the first `it` iterations and save it to `path` (same argument list as the reference's plot_loss)."""
    plt = _plt()
    curves = [np.asarray(c)[:it] for c in (mse_corrupted, mse_gt)]
    psnr = np.asarray(psnrs)[:it, LOSS_FIGURE["psnr_column"]]
    fig, left = plt.subplots()
    for c in curves:
        left.plot(np.arange(len(c)), c)
    left.set(title=title, **LOSS_FIGURE["left"])
    left.grid(True)
    right = left.twinx()
    right.plot(np.arange(len(psnr)), psnr, LOSS_FIGURE["psnr_style"])
    right.set_ylabel(y_label)
    fig.tight_layout()
    fig.savefig(path)
    plt.close("all")


def plot_results(MSE_CORRUPTED, MSE_GT, PSNRS, SSIMS, run_dir, file):
    """bayesian_optimization.py:201-258: mse_noisy.png, mse_gt.png, psnrs.png, ssims.png; prints '<method> PSNR_max / SSIM_max' into `file`."""
    plt = _plt()
    for name, D, title, ylim in (("mse_noisy", MSE_CORRUPTED, 'MSE noisy', 0.03), ("mse_gt", MSE_GT, 'MSE GT', 0.01)):
        fig, ax = plt.subplots(1, 1)
        for key, loss in D.items():
            ax.plot(range(len(loss)), loss, label=key)
        ax.set_title(title); ax.set_xlabel('iteration'); ax.set_ylabel('mse loss'); ax.set_ylim(0, ylim); ax.grid(True); ax.legend()
        plt.tight_layout(); plt.savefig(os.path.join(run_dir, name + ".png"))
    for name, D, labels, word in (("psnrs", PSNRS, ["psnr_noisy", "psnr_gt", "psnr_gt_sm"], "PSNR"), ("ssims", SSIMS, ["ssim_noisy", "ssim_gt", "ssim_gt_sm"], "SSIM")):
        fig, axs = plt.subplots(1, 3, constrained_layout=True)
        for key, v in D.items():
            v = np.array(v)
            print("%s %s_max: %s" % (key, word, np.max(v)), file=file)
            for i in range(v.shape[1]):
                axs[i].plot(range(v.shape[0]), v[:, i], label=key)
                axs[i].set_title(labels[i]); axs[i].set_xlabel('iteration'); axs[i].set_ylabel(word.lower()); axs[i].legend()
        plt.savefig(os.path.join(run_dir, name + ".png"))
    plt.close('all')


def snapshot_pngs(run_dir, method, it, mse_corrupted, mse_gt, psnrs, out_avg, var, ale):
    """What every show_every-th iteration writes with plot=True (bayesian_optimization.py:1418-1422): out_avg (C,H,W), var / ale (C,H,W) or None."""
    plot_loss(mse_corrupted, mse_gt, psnrs, it, os.path.join(run_dir, "loss_%s.png" % method), "MSE " + method.upper())
    np_to_pil(out_avg).save(os.path.join(run_dir, "out_avg.png"), "PNG")
    for name, a in (("out_var", var), ("out_ale", ale)):
        if a is not None:
            m = float(np.max(a))
            np_to_pil(a / m if m > 0 else a).save(os.path.join(run_dir, name + ".png"), "PNG")


def sr_input_png(run_dir, img_hr, img_lr, factor):
    """input.png of the SR runner (:2103-2108): ground truth beside the nearest-neighbour enlargement of the observation."""
    up = np.repeat(np.repeat(np.asarray(img_lr), factor, axis=0), factor, axis=1)[None]
    both = np.concatenate([np.asarray(img_hr), up[:, :img_hr.shape[1], :img_hr.shape[2]]], axis=2)
    np_to_pil(both).save(os.path.join(run_dir, "input.png"), "PNG")


def read_like_notebooks(path, task, method):
    """The reads of eval_{denoising,sr,ct,inp}.ipynb on one save.npz (np.load(allow_pickle=True), `.flat[0]` dicts, the image keys, the
    shape arithmetic of their error / uncertainty cells).  Returns a dict of the derived arrays; raises if anything does not fit."""
    run = np.load(path, allow_pickle=True)
    psnrs = run['psnrs'].flat[0]; ssims = run['ssims'].flat[0]; losses = run['mse_gt'].flat[0]
    out = dict(psnr_curve=psnrs[method][::100], ssim_curve=ssims[method][::100], mse_curve=losses[method][::100])
    recons = run['recons'].flat[0][method]
    out["final_recon"] = recons[-1][0]                                      # io.imsave(..., img_as_ubyte(recons[-1][0]))
    k0, k1 = HEAD_KEYS[task]
    gt, other = run[k0], run[k1]
    if method != "dip":
        uncert = run['uncerts'].flat[0][method][-1]; uncert_ale = run['uncerts_ale'].flat[0][method][-1]
        if task == "inp":
            imgs_mc = recons[-25:]
            errvar = ((imgs_mc - gt[None, ].repeat(imgs_mc.shape[0], axis=0)) ** 2).mean(axis=(0,)) * other
        else:
            imgs_mc = recons[-25:].transpose(1, 0, 2, 3)
            errvar = ((imgs_mc - gt.repeat(imgs_mc.shape[1], axis=0)) ** 2).mean(axis=(0, 1))
        out.update(errvar=errvar, uncerts=uncert + uncert_ale)
    else:
        assert run['uncerts'].flat[0] == {} and run['uncerts_ale'].flat[0] == {}
    out["head"] = (gt, other)
    return out
