"""ctypes/numpy front-end of the CPU oracle (oracle/mfvi_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmfvi_oracle.so")

MAXS = 8
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "mfvi_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmfvi_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OracleNet(C.Structure):
    _fields_ = [("n_scales", C.c_int), ("nd", C.c_int * MAXS), ("nu", C.c_int * MAXS), ("ns", C.c_int * MAXS),
                ("input_depth", C.c_int), ("n_out", C.c_int), ("fd", C.c_int), ("fu", C.c_int), ("fs", C.c_int),
                ("H", C.c_int), ("W", C.c_int), ("drop_down", C.c_float), ("drop_up", C.c_float)]


def make_net(H, W, input_depth=16, n_out=2, nd=(16, 32, 64, 128, 128), nu=(16, 32, 64, 128, 128), ns=(4, 4, 4, 4, 4),
             fd=3, fu=3, fs=1, drop_down=0.0, drop_up=0.0):
    """drop_down / drop_up: Dropout2d(p) after the deeper / up convolutions (the MC-dropout sibling's net)."""
    n = OracleNet()
    n.drop_down, n.drop_up = drop_down, drop_up
    n.n_scales = len(nd)
    for i in range(len(nd)):
        n.nd[i], n.nu[i], n.ns[i] = nd[i], nu[i], ns[i]
    n.input_depth, n.n_out, n.fd, n.fu, n.fs, n.H, n.W = input_depth, n_out, fd, fu, fs, H, W
    return n


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    L.oracle_net_forward.restype = C.c_void_p
    L.oracle_tape_conv_out.restype = C.c_long
    L.oracle_tape_cat.restype = C.c_long
    L.oracle_tape_conv_grad.restype = C.c_long
    for name in ("oracle_kl", "oracle_gaussian_nll", "oracle_mse", "oracle_psnr", "oracle_ssim", "oracle_elbo_grad"):
        getattr(L, name).restype = C.c_double
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n=0):
    return lib().oracle_set_threads(C.c_int(n))


# ---- RNG spec ------------------------------------------------------------------------
def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32); k = np.asarray(key, dtype=np.uint32); o = np.zeros(4, np.uint32)
    lib().oracle_philox(_p(c), _p(k), _p(o))
    return o


def normal_fill(seed, domain, stream, sample, step, n):
    out = np.zeros(n, np.float32)
    lib().oracle_normal_fill(C.c_uint64(seed), C.c_uint32(domain), C.c_uint32(stream), C.c_uint32(sample),
                             C.c_uint32(step), C.c_long(n), _p(out))
    return out


def uniform_fill(seed, stream, sample, step, n):
    out = np.zeros(n, np.float32)
    lib().oracle_uniform_fill(C.c_uint64(seed), C.c_uint32(stream), C.c_uint32(sample), C.c_uint32(step), C.c_long(n), _p(out))
    return out


def eps(seed, step, sample, layer, tensor, n):
    out = np.zeros(n, np.float32)
    lib().oracle_eps(C.c_uint64(seed), C.c_uint32(step), C.c_uint32(sample), C.c_int(layer), C.c_int(tensor), C.c_long(n), _p(out))
    return out


# ---- elementary ops ------------------------------------------------------------------
def softplus(x):
    x = _f(x); o = np.empty_like(x); lib().oracle_softplus(_p(x), C.c_long(x.size), _p(o)); return o


def reparam(mu, rho, e):
    mu, rho, e = _f(mu), _f(rho), _f(e); o = np.empty_like(mu)
    lib().oracle_reparam(_p(mu), _p(rho), _p(e), C.c_long(mu.size), _p(o)); return o


def conv_fwd(x, w, b, stride):
    x, w = _f(x), _f(w); b = None if b is None else _f(b)
    cin, H, W = x.shape; cout, _, k, _ = w.shape; p = k // 2
    Ho, Wo = (H + 2 * p - k) // stride + 1, (W + 2 * p - k) // stride + 1
    y = np.empty((cout, Ho, Wo), np.float32)
    lib().oracle_conv_fwd(_p(x), cin, H, W, _p(w), _p(b), cout, k, stride, _p(y))
    return y


def conv_bwd(x, w, stride, dy, need_dx=True):
    x, w, dy = _f(x), _f(w), _f(dy)
    cin, H, W = x.shape; cout, _, k, _ = w.shape
    dx = np.empty_like(x) if need_dx else None
    dw = np.zeros(w.shape, np.float64); db = np.zeros(cout, np.float64)
    lib().oracle_conv_bwd(_p(x), cin, H, W, _p(w), cout, k, stride, _p(dy), _p(dx), _p(dw), _p(db))
    return dx, dw, db


def bn_fwd(x, gamma, beta, eps_=1e-5):
    x, gamma, beta = _f(x), _f(gamma), _f(beta); c = x.shape[0]; hw = x[0].size
    y = np.empty_like(x); m = np.empty(c, np.float32); r = np.empty(c, np.float32)
    lib().oracle_bn_fwd(_p(x), c, hw, _p(gamma), _p(beta), C.c_float(eps_), _p(y), _p(m), _p(r))
    return y, m, r


def bn_bwd(x, gamma, mean, rstd, dy):
    x, gamma, mean, rstd, dy = _f(x), _f(gamma), _f(mean), _f(rstd), _f(dy); c = x.shape[0]; hw = x[0].size
    dx = np.empty_like(x); dg = np.zeros(c, np.float64); db = np.zeros(c, np.float64)
    lib().oracle_bn_bwd(_p(x), c, hw, _p(gamma), _p(mean), _p(rstd), _p(dy), _p(dx), _p(dg), _p(db))
    return dx, dg, db


def lrelu_fwd(x, slope=0.2):
    x = _f(x); y = np.empty_like(x); lib().oracle_lrelu_fwd(_p(x), C.c_long(x.size), C.c_float(slope), _p(y)); return y


def upsample2_fwd(x):
    x = _f(x); c, H, W = x.shape; y = np.empty((c, 2 * H, 2 * W), np.float32)
    lib().oracle_upsample2_fwd(_p(x), c, H, W, _p(y)); return y


def upsample2_bwd(dy):
    dy = _f(dy); c, Ho, Wo = dy.shape; dx = np.empty((c, Ho // 2, Wo // 2), np.float32)
    lib().oracle_upsample2_bwd(_p(dy), c, Ho // 2, Wo // 2, _p(dx)); return dx


def upsample2_nearest_fwd(x):
    x = _f(x); c, H, W = x.shape; y = np.empty((c, 2 * H, 2 * W), np.float32)
    lib().oracle_upsample2_nearest_fwd(_p(x), c, H, W, _p(y)); return y


def upsample2_nearest_bwd(dy):
    dy = _f(dy); c, Ho, Wo = dy.shape; dx = np.empty((c, Ho // 2, Wo // 2), np.float32)
    lib().oracle_upsample2_nearest_bwd(_p(dy), c, Ho // 2, Wo // 2, _p(dx)); return dx


def gaussian_nll_inp(out4, target3, mask, scale=1.0, want_grad=False):
    """sigmoid on out4[:3] + gaussian_nll_inpainting; mask (1|3, H, W)."""
    out4, target3, mask = _f(out4), _f(target3), _f(mask)
    HW = out4.shape[1] * out4.shape[2]
    d = np.empty_like(out4) if want_grad else None
    lib().oracle_gaussian_nll_inp.restype = C.c_double
    v = lib().oracle_gaussian_nll_inp(_p(out4), _p(target3), _p(mask), C.c_int(mask.shape[0]), C.c_long(HW), C.c_double(scale), _p(d))
    return (v, d) if want_grad else v


def mse_sigmoid_masked(out4, target3, mask, scale=1.0, want_grad=False):
    """run_inp_dip's loss: F.mse_loss(out4[:3].sigmoid() * mask, target3 * mask); mask (1|3, H, W)."""
    out4, target3, mask = _f(out4), _f(target3), _f(mask)
    HW = out4.shape[-1] * out4.shape[-2]
    d = np.empty_like(out4) if want_grad else None
    lib().oracle_mse_sigmoid_masked.restype = C.c_double
    v = lib().oracle_mse_sigmoid_masked(_p(out4), _p(target3), _p(mask), C.c_int(mask.shape[0]), C.c_long(HW), C.c_double(scale), _p(d))
    return (v, d) if want_grad else v


def kl(mu, rho, prior_sigma, prior_mu=0.0, scale=0.0, want_grad=False):
    mu, rho = _f(mu).ravel(), _f(rho).ravel()
    dmu = np.zeros_like(mu) if want_grad else None; drho = np.zeros_like(rho) if want_grad else None
    v = lib().oracle_kl(_p(mu), _p(rho), C.c_long(mu.size), C.c_float(prior_mu), C.c_float(prior_sigma),
                        C.c_double(scale), _p(dmu), _p(drho))
    return (v, dmu, drho) if want_grad else v


def gaussian_nll(mu, s, target, scale=1.0, want_grad=False):
    mu, s, target = _f(mu), _f(s), _f(target)
    dmu = np.empty_like(mu) if want_grad else None; ds = np.empty_like(s) if want_grad else None
    v = lib().oracle_gaussian_nll(_p(mu), _p(s), _p(target), C.c_long(mu.size), C.c_double(scale), _p(dmu), _p(ds))
    return (v, dmu, ds) if want_grad else v


def mse(a, b, scale=1.0, want_grad=False):
    a, b = _f(a), _f(b); da = np.empty_like(a) if want_grad else None
    v = lib().oracle_mse(_p(a), _p(b), C.c_long(a.size), C.c_double(scale), _p(da))
    return (v, da) if want_grad else v


def adam(p, g, m, v, lr, t):
    """In-place on float32 contiguous arrays."""
    lib().oracle_adam(_p(p), _p(g), _p(m), _p(v), C.c_long(p.size), C.c_float(lr), C.c_int(t))


def adamw(p, g, m, v, lr, t, wd):
    """torch.optim.AdamW with decoupled weight decay, in place."""
    lib().oracle_adamw(_p(p), _p(g), _p(m), _p(v), C.c_long(p.size), C.c_float(lr), C.c_int(t), C.c_float(wd))


def dropout_mask(seed, step, sample, layer, p, n_channels):
    """Dropout2d factors (0 or 1/(1-p)) of conv layer `layer` for one forward (RNG domain 5)."""
    d = np.zeros(n_channels, np.float32)
    lib().oracle_dropout_mask(C.c_uint64(seed), C.c_uint32(step), C.c_uint32(sample), C.c_int(layer), C.c_float(p), C.c_int(n_channels), _p(d))
    return d


def psnr(a, b):
    a, b = _f(a), _f(b); return lib().oracle_psnr(_p(a), _p(b), C.c_long(a.size))


def ssim(a, b):
    a, b = _f(a), _f(b); H, W = a.shape[-2:]; return lib().oracle_ssim(_p(a), _p(b), H, W)


def radon_fwd(img, theta_deg):
    img, theta_deg = _f(img), _f(theta_deg); H, W = img.shape[-2:]; T = theta_deg.size
    s = np.empty((T, W), np.float32); lib().oracle_radon_fwd(_p(img), H, W, _p(theta_deg), T, _p(s)); return s


def radon_adj(dsino, theta_deg, H, W):
    dsino, theta_deg = _f(dsino), _f(theta_deg); T = theta_deg.size
    d = np.empty((H, W), np.float32); lib().oracle_radon_adj(_p(dsino), H, W, _p(theta_deg), T, _p(d)); return d


# ---- whole net -----------------------------------------------------------------------
def net_table(net):
    nc = C.c_int(); nb = C.c_int(); nvi = C.c_long(); nbn = C.c_long()
    conv = np.zeros((64, 6), np.int64); bn = np.zeros((64, 2), np.int64)
    lib().oracle_net_table(C.byref(net), C.byref(nc), _p(conv), C.byref(nb), _p(bn), C.byref(nvi), C.byref(nbn))
    return conv[:nc.value].copy(), bn[:nb.value].copy(), nvi.value, nbn.value


class Tape:
    def __init__(self, net, handle):
        self.net, self.h = net, C.c_void_p(handle)

    def conv_out(self, conv_id):
        n = lib().oracle_tape_conv_out(self.h, conv_id, None, None, None)
        conv, _, _, _ = net_table(self.net)
        cout = int(conv[conv_id][1])
        y = np.empty(n, np.float32); m = np.zeros(cout, np.float32); r = np.zeros(cout, np.float32)
        lib().oracle_tape_conv_out(self.h, conv_id, _p(y), _p(m), _p(r))
        return y.reshape(cout, -1), m, r

    def conv_grad(self, conv_id, which=0):
        n = lib().oracle_tape_conv_grad(self.h, conv_id, which, None)
        if n < 0:
            return None
        y = np.empty(n, np.float32); lib().oracle_tape_conv_grad(self.h, conv_id, which, _p(y)); return y

    def cat(self, scale):
        n = lib().oracle_tape_cat(self.h, scale, None)
        y = np.empty(n, np.float32); lib().oracle_tape_cat(self.h, scale, _p(y)); return y

    def backward(self, dout, n_vi, n_bnp, want_dz=False):
        dout = _f(dout)
        dmu = np.zeros(n_vi, np.float64); drho = np.zeros(n_vi, np.float64); dbn = np.zeros(n_bnp, np.float64)
        dz = np.empty((self.net.input_depth, self.net.H, self.net.W), np.float32) if want_dz else None
        lib().oracle_net_backward(self.h, _p(dout), _p(dmu), _p(drho), _p(dbn), _p(dz))
        return dmu, drho, dbn, dz

    def free(self):
        if self.h:
            lib().oracle_tape_free(self.h); self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def net_forward(net, mu, rho, bn, z, seed, step, sample, sample_weights=True):
    """One MC forward (MeanFieldVI.forward).  mu/rho/bn/z must stay alive while the tape is used."""
    mu, rho, bn, z = _f(mu), _f(rho), _f(bn), _f(z)
    out = np.empty((net.n_out, net.H, net.W), np.float32)
    h = lib().oracle_net_forward(C.byref(net), _p(mu), _p(rho), _p(bn), _p(z), C.c_uint64(seed), C.c_uint32(step),
                                 C.c_uint32(sample), C.c_int(1 if sample_weights else 0), _p(out))
    t = Tape(net, h); t._keep = (mu, rho, bn, z)
    return out, t


def elbo_grad(net, mu, rho, bn, z, target, task=0, factor=4, theta_deg=None, seed=1, step=0, k0=0, K=1, K_total=None,
              temp=1.0, prior_sigma=0.1, with_kl=True, want_out=False):
    mu, rho, bn, z, target = _f(mu), _f(rho), _f(bn), _f(z), _f(target)
    K_total = K if K_total is None else K_total
    th = None if theta_deg is None else _f(theta_deg)
    dmu = np.zeros_like(mu); drho = np.zeros_like(rho); dbn = np.zeros_like(bn)
    outs = np.empty((K, net.n_out, net.H, net.W), np.float32) if want_out else None
    nll = C.c_double(); klv = C.c_double()
    loss = lib().oracle_elbo_grad(C.byref(net), _p(mu), _p(rho), _p(bn), _p(z), _p(target), C.c_int(task), C.c_int(factor),
                                  _p(th), C.c_int(0 if th is None else th.size), C.c_uint64(seed), C.c_uint32(step),
                                  C.c_int(k0), C.c_int(K), C.c_int(K_total), C.c_float(temp), C.c_float(prior_sigma),
                                  C.c_int(1 if with_kl else 0), _p(dmu), _p(drho), _p(dbn), _p(outs),
                                  C.byref(nll), C.byref(klv))
    return dict(loss=loss, nll=nll.value, kl=klv.value, dmu=dmu, drho=drho, dbn=dbn, out=outs)


def sibling_grad(net, mu, bn, z, target, loss="mse0", seed=1, step=0, k0=0, K=1, K_total=None, want_out=False, factor=1, theta_deg=None):
    """Gradient of the non-Bayesian siblings' loss (DIP / SGLD: F.mse_loss(out[:, :1], y), bayesian_optimization.py:1177,1780;
    MC dropout: gaussian_nll, :1578) averaged over K forwards of the deterministic-weight net (w = mu); Dropout2d masks, when
    the net has them, come from RNG domain 5 keyed by (step, sample)."""
    mu, bn, z, target = _f(mu), _f(bn), _f(z), _f(target)
    K_total = K if K_total is None else K_total
    _, _, n_vi, n_bnp = net_table(net)
    rho = np.zeros_like(mu)
    dmu = np.zeros(n_vi, np.float64); dbn = np.zeros(n_bnp, np.float64)
    total = 0.0; outs = []
    for k in range(k0, k0 + K):
        out, tape = net_forward(net, mu, rho, bn, z, seed, step, k, sample_weights=False)
        dout = np.zeros_like(out)
        f = factor                                     # SR projection out[..., ::f, ::f] (bayesian_optimization.py:2095-2099)
        if loss == "mse0":
            v, g = mse(out[0, ::f, ::f], target, 1.0 / K_total, want_grad=True)
            dout[0, ::f, ::f] = g
        elif loss == "gnll":
            v, g0, g1 = gaussian_nll(out[0, ::f, ::f], out[1, ::f, ::f], target, 1.0 / K_total, want_grad=True)
            dout[0, ::f, ::f] = g0; dout[1, ::f, ::f] = g1
        elif loss == "radon":                          # mse_loss(radon(out), sinogram) of the CT runs (:377, :789, :991)
            sino = radon_fwd(out[0], theta_deg)
            v, ds = mse(sino, target, 1.0 / K_total, want_grad=True)
            dout[0] = radon_adj(ds, theta_deg, net.H, net.W)
        else:
            raise ValueError(loss)
        total += v / K_total
        g = tape.backward(dout, n_vi, n_bnp)
        dmu += g[0]; dbn += g[2]
        tape.free(); outs.append(out)
    r = dict(loss=total, dmu=dmu.astype(np.float32), dbn=dbn.astype(np.float32))
    if want_out:
        r["out"] = np.stack(outs)
    return r


# ---- deterministic synthetic inputs ----------------------------------
def init_params(net, seed):
    """mu ~ N(0, 0.1^2), rho ~ N(-3, 0.1^2) (modules/module.py:26-30,56-62) from RNG domain INIT;
    BN gamma = 1, beta = 0 (torch defaults)."""
    conv, bn, n_vi, n_bnp = net_table(net)
    mu = 0.1 * normal_fill(seed, 2, 0, 0, 0, n_vi)
    rho = -3.0 + 0.1 * normal_fill(seed, 2, 1, 0, 0, n_vi)
    bnp = np.zeros(n_bnp, np.float32)
    for c, off in bn:
        bnp[off:off + c] = 1.0
    return mu.astype(np.float32), rho.astype(np.float32), bnp


def phantom(H, W, seed):
    """Synthetic ground-truth image in [0,1]: soft ellipses + hard bars + smooth texture."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    img = np.zeros((H, W), np.float64)
    for _ in range(6):
        cx, cy = rng.uniform(-0.6, 0.6, 2); ax, ay = rng.uniform(0.1, 0.5, 2); th = rng.uniform(0, np.pi)
        amp = rng.uniform(0.2, 0.6)
        xr = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); yr = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        img += amp / (1.0 + np.exp(np.minimum(((xr / ax) ** 2 + (yr / ay) ** 2 - 1.0) * 8.0, 60.0)))
    img[int(0.2 * H):int(0.25 * H), int(0.1 * W):int(0.9 * W)] += 0.3
    img[int(0.1 * H):int(0.9 * H), int(0.7 * W):int(0.74 * W)] += 0.25
    img += 0.03 * np.sin(9 * xx) * np.cos(7 * yy)
    img -= img.min(); img /= img.max()
    return img.astype(np.float32)


def noisy(img, p_sigma, seed):
    rng = np.random.default_rng(seed + 1)
    return np.clip(img + rng.normal(scale=p_sigma, size=img.shape), 0, 1).astype(np.float32)


# ---- the runners' per-iteration bookkeeping (numpy float32 restatement) ------------------------
def bookkeeping_raw(task, seed, i, img, C, k=0):
    """Synthetic raw network output of iteration i, MC sample k (values leave [0, 1] so the clips matter)."""
    img = np.asarray(img, np.float32)
    H, W = img.shape[-2:]
    nz = normal_fill(seed, 2, 20 + i, k, 0, C * H * W).reshape(C, H, W)
    out = np.empty((C, H, W), np.float32)
    if task == "inp":
        out[:3] = 3.0 * (img - 0.5) + 0.5 * nz[:3]
        out[3] = 1.0 + nz[3]
    else:
        out[0] = img + 0.2 * nz[0]
        if C > 1:
            out[1] = 1.5 + nz[1]
    return out


class Bookkeeper:
    """EMA / clips / 25-slot ring buffers / metrics of the MFVI runners, float32 like the reference's torch code, with the build's
    K-sample generalisation (`out` := mean over the K samples of [out_k[:1], exp(-out_k[1:])]; K = 1 is the reference):
    den bayesian_optimization.py:1374-1416, sr :2190-2236, ct :584-626, inp :3039-3090."""

    def __init__(self, task, H, W, gt, noisy=None, mask=None, factor=4, weight=0.99, mc_iter=25):
        self.task, self.H, self.W, self.f, self.w, self.R = task, H, W, factor, np.float32(weight), mc_iter
        self.gt = np.asarray(gt, np.float32); self.noisy = None if noisy is None else np.asarray(noisy, np.float32)
        self.mask = None if mask is None else np.asarray(mask, np.float32)
        nc = 3 if task == "inp" else 1
        self.ring_epi = np.zeros((mc_iter, nc, H, W), np.float32); self.ring_ale = np.zeros((mc_iter, H, W), np.float32)
        self.ema = None; self.i = 0

    @staticmethod
    def _psnr(a, b):
        return 10.0 * np.log10(1.0 / np.mean((a.astype(np.float32) - b.astype(np.float32)).astype(np.float64) ** 2))

    @staticmethod
    def _ssim(a, b):
        a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
        if a.ndim == 2:
            return ssim(a, b)
        return float(np.mean([ssim(a[c], b[c]) for c in range(a.shape[0])]))

    def step(self, outs):
        """outs: [K][C][H][W] raw network outputs of this iteration -> the 8 numbers the runner stores
        (mse_corrupted, mse_gt, 3 PSNR, 3 SSIM)."""
        outs = np.asarray(outs, np.float32); K = outs.shape[0]
        one = np.float32(1.0)
        nim = 3 if self.task == "inp" else 1
        tr = (lambda x: (one / (one + np.exp(-x))).astype(np.float32)) if self.task == "inp" else (lambda x: x)
        m = np.zeros((nim, self.H, self.W), np.float32)
        for k in range(K):
            m += tr(outs[k, :nim])
        m /= np.float32(K)
        chans = [m]
        a = None
        if outs.shape[1] > nim:
            a = np.zeros((1, self.H, self.W), np.float32)
            for k in range(K):
                a += np.exp(-outs[k, nim:nim + 1]).astype(np.float32)
            a /= np.float32(K)
            chans.append(a)
        out = np.concatenate(chans, 0)
        self.ema = out.copy() if self.ema is None else (self.ema * self.w + out * (one - self.w)).astype(np.float32)
        _out = np.clip(out[:nim], 0, 1); _avg = np.clip(self.ema[:nim], 0, 1)
        slot = self.i % self.R
        self.ring_epi[slot] = _out
        if a is not None:
            self.ring_ale[slot] = np.clip(a[0], 0, 1)
        self.i += 1
        mse_ = lambda x, y: float(np.mean((x.astype(np.float32) - y.astype(np.float32)).astype(np.float64) ** 2))
        gt = self.gt if self.task == "inp" else self.gt[None]
        if self.task == "den":
            nz = self.noisy[None]
            return [mse_(self.ema[:1], nz), mse_(self.ema[:1], gt), self._psnr(nz, _out), self._psnr(gt, _out), self._psnr(gt, _avg),
                    self._ssim(nz[0], _out[0]), self._ssim(gt[0], _out[0]), self._ssim(gt[0], _avg[0])]
        if self.task == "sr":
            f = self.f
            small = gt[:, ::f, ::f]
            # out_lr is the projection of the RAW output (channel 0 is untouched by the exp): mean over samples of out_k[0, ::f, ::f]
            _lr = np.clip(m[:, ::f, ::f], 0, 1)
            return [mse_(self.ema[:1, ::f, ::f], small), mse_(self.ema[:1], gt), self._psnr(small, _lr), self._psnr(gt, _out), self._psnr(gt, _avg),
                    self._ssim(small[0], _lr[0]), self._ssim(gt[0], _out[0]), self._ssim(gt[0], _avg[0])]
        if self.task == "ct":
            m0 = mse_(self.ema[:1], gt); p0 = self._psnr(gt, _out); s0 = self._ssim(gt[0], _out[0])
            return [m0, m0, p0, p0, self._psnr(gt, _avg), s0, s0, self._ssim(gt[0], _avg[0])]
        mk = self.mask
        m0 = mse_(self.ema[:3], gt)
        return [m0, m0, self._psnr(gt, _out), self._psnr(gt * mk, _out * mk), self._psnr(gt * mk, _avg * mk),
                self._ssim(gt, _out), self._ssim(gt * mk, _out * mk), self._ssim(gt * mk, _avg * mk)]

    def snapshot(self):
        """-> (unbiased variance of the epistemic ring over all R slots, mean of the aleatoric ring, clipped EMA image)."""
        nim = 3 if self.task == "inp" else 1
        var = np.var(self.ring_epi.astype(np.float64), axis=0, ddof=1).astype(np.float32)
        ale = np.mean(self.ring_ale.astype(np.float64), axis=0).astype(np.float32)
        return (var if nim == 3 else var[0]), ale, np.clip(self.ema[:nim], 0, 1) if nim == 3 else np.clip(self.ema[0], 0, 1)


# ---- bfloat16 parameter storage (BASELINE configs[4]) ------------------------------------------
def bf16_round(x):
    """float32 -> nearest bfloat16 (ties to even) -> float32; NaN-free inputs."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def bf16_bits(x):
    """float32 array holding bf16-representable values -> uint16 bit patterns."""
    return (np.ascontiguousarray(x, np.float32).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(b):
    return (np.ascontiguousarray(b, np.uint16).astype(np.uint32) << 16).view(np.float32)


def adam_bf16_sr(p_bits, g, m, v, lr, t, seed, stream):
    """In place: Adam in float32 + stochastic rounding to bf16 (the build's update rule for bf16 mu / rho); p_bits uint16, g / m / v float32."""
    assert p_bits.dtype == np.uint16 and p_bits.flags.c_contiguous
    lib().oracle_adam_bf16_sr(_p(p_bits), _p(g), _p(m), _p(v), C.c_long(p_bits.size), C.c_float(lr), C.c_int(t), C.c_uint64(seed), C.c_uint32(stream))


# ---- local reparameterisation (Conv2dLRT) on the oracle's conv primitives ----------------------
def lrt_eps(seed, step, sample, layer, n):
    """eps of an LRT layer in OUTPUT space: element j of [Cout][Ho][Wo] (RNG domain 7, stream = layer)."""
    return normal_fill(seed, 7, layer, sample, step, n)


def lrt_conv(x, w_mu, w_rho, b_mu, b_rho, eps, stride, dy=None):
    """LRTLayer.forward with layer_fn = conv2d behind ReflectionPad2d(k // 2) (BayTorch/modules/reparam_layers.py:59-72,
    models/common.py:100-135) and its autograd:  y = conv(x, mu, b_mu) + sqrt(1e-16 + conv(x^2, softplus(rho)^2, softplus(b_rho)^2)) * eps.
    Returns y, or (y, dx, dw_mu, dw_rho, db_mu, db_rho) when dy is given."""
    x = _f(x); w_mu = _f(w_mu); w_rho = _f(w_rho); b_mu = _f(b_mu); b_rho = _f(b_rho); eps = _f(eps)
    sw = softplus(w_rho.ravel()).reshape(w_rho.shape); sb = softplus(b_rho)
    a = conv_fwd(x, w_mu, b_mu, stride)
    s2 = conv_fwd(x * x, sw * sw, sb * sb, stride)
    std = np.sqrt(np.float32(1e-16) + s2)
    y = a + std * eps.reshape(a.shape)
    if dy is None:
        return y
    dy = _f(dy)
    ds2 = dy * eps.reshape(a.shape) / (2.0 * std)
    dxa, dwm, dbm = conv_bwd(x, w_mu, stride, dy)
    dxb, dws, dbs = conv_bwd(x * x, sw * sw, stride, ds2)
    dx = dxa + 2.0 * x * dxb
    sg = lambda r: 1.0 / (1.0 + np.exp(-r.astype(np.float64)))
    return y, dx, dwm, dws * 2.0 * sw * sg(w_rho), dbm, dbs * 2.0 * sb * sg(b_rho)
