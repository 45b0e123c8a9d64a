"""The reference's Gaussian-process outer loop over two hyper-parameters (bayesian_optimization.py:3545-3880), host side only: it
proposes candidates, the fits themselves run through `fanout.run_jobs` (one worker per device) on the HIP path.

PARITY UNPINNED: the reference builds this on gpytorch (ExactGP + ScaleKernel(RBFKernel) + ConstantMean + GaussianLikelihood,
ExactMarginalLogLikelihood) and skimage.feature.peak_local_max, neither of which exists in this image, so no golden vectors can be
generated.  What follows restates those published models with plain torch (CPU, float64) and scipy:

  model        f ~ GP(c, s * exp(-|x - x'|^2 / (2 l^2))),  y = f + N(0, noise)                      :3546-3562
               l = softplus(raw), set to 0.3; s = softplus(raw), raw 0; c raw 0; noise = softplus(raw) + 1e-4, raw 0
               priors  c ~ N(15, 4),  noise ~ Gamma(concentration 0.01, rate 100)                     :3550-3551, :3567-3569
  training     Adam(lr 0.05) on -(log N(y | c, K + noise I) + log-priors) / n, 2000 steps             :3565-3600
  acquisition  expected improvement over max(posterior mean at the evaluated points), or UCB          :3603-3646
  candidates   <= 4 local maxima of the acquisition on the 100 x 100 grid (min distance 5, >= 0.1 * max) plus the global one, each refined
               with L-BFGS (strong Wolfe) through a sigmoid parametrisation of [0, 1]^2               :3649-3684
  loop         evaluate candidates (fan-out), drop NaNs, refit, propose; 20 rounds                    :3727-3880
  scaling      normalize_X / unnormalize_X (:3687-3706) work on log10 of the hyper-parameters: log10, then the affine map of the log
               bounds onto [0, 1]; back: the inverse map, then 10 ** x.  The helpers below restate exactly that (the log10 / pow(10)
               live inside them, as in the reference).
"""
import itertools
import math

import numpy as np


def _softplus_inv(y):
    return math.log(math.expm1(y))


class ExactGP:
    """ExactGPModel of the reference (:3546-3562) with gpytorch's default constraints and the priors named above."""

    def __init__(self, X, Y):
        import torch
        self.torch = torch
        self.X = X.double(); self.Y = Y.double()
        z = lambda v: torch.tensor(float(v), dtype=torch.float64, requires_grad=True)
        self.raw_mean = z(0.0)
        self.raw_lengthscale = z(_softplus_inv(3e-1))
        self.raw_outputscale = z(0.0)
        self.raw_noise = z(0.0)
        self._chol = None

    def parameters(self):
        return [self.raw_mean, self.raw_lengthscale, self.raw_outputscale, self.raw_noise]

    @property
    def lengthscale(self):
        return self.torch.nn.functional.softplus(self.raw_lengthscale)

    @property
    def outputscale(self):
        return self.torch.nn.functional.softplus(self.raw_outputscale)

    @property
    def noise(self):
        return self.torch.nn.functional.softplus(self.raw_noise) + 1e-4

    def kernel(self, A, B):
        d2 = ((A[:, None, :] - B[None, :, :]) / self.lengthscale).pow(2).sum(-1)
        return self.outputscale * self.torch.exp(-0.5 * d2)

    def neg_mll(self):
        """-(marginal log likelihood + log-priors) / n, ExactMarginalLogLikelihood's normalisation."""
        torch = self.torch
        n = self.X.shape[0]
        K = self.kernel(self.X, self.X) + self.noise * torch.eye(n, dtype=torch.float64)
        L = torch.linalg.cholesky(K)
        r = (self.Y - self.raw_mean).unsqueeze(-1)
        alpha = torch.cholesky_solve(r, L)
        mll = -0.5 * (r * alpha).sum() - torch.log(torch.diagonal(L)).sum() - 0.5 * n * math.log(2 * math.pi)
        mll = mll + torch.distributions.Normal(15.0, 4.0).log_prob(self.raw_mean)
        mll = mll + torch.distributions.Gamma(0.01, 100.0).log_prob(self.noise)
        return -mll / n

    def eval(self):
        torch = self.torch
        with torch.no_grad():
            n = self.X.shape[0]
            K = self.kernel(self.X, self.X) + self.noise * torch.eye(n, dtype=torch.float64)
            self._chol = torch.linalg.cholesky(K)
            self._alpha = torch.cholesky_solve((self.Y - self.raw_mean).unsqueeze(-1), self._chol)
        return self

    def posterior(self, Xs):
        """Latent posterior mean / variance at Xs (what `gp(X)` returns in eval mode: no likelihood noise added)."""
        torch = self.torch
        Ks = self.kernel(Xs.double(), self.X)
        mean = self.raw_mean + (Ks @ self._alpha).squeeze(-1)
        v = torch.cholesky_solve(Ks.transpose(0, 1), self._chol)
        var = self.outputscale - (Ks * v.transpose(0, 1)).sum(-1)
        return mean, var


def train_gp(X_train, Y_train, iter_max=2000, verbose=False):
    """:3565-3600."""
    import torch
    gp = ExactGP(X_train, Y_train)
    opt = torch.optim.Adam(gp.parameters(), lr=0.05)
    for i in range(iter_max):
        opt.zero_grad()
        loss = gp.neg_mll()
        loss.backward()
        if verbose and i % 100 == 0:
            print("Iter %4d/%d - Loss: %.4f   lengthscale: %.3f   noise: %.4f" % (i + 1, iter_max, loss.item(), gp.lengthscale.item(), gp.noise.item()))
        opt.step()
    return gp.eval()


def expected_improvement(gp, X, X_train):
    """:3603-3632."""
    import torch
    mu, var = gp.posterior(X)
    mu_sample, _ = gp.posterior(X_train)
    sigma = var.clamp_min(1e-9).sqrt().reshape(-1, 1)
    u = (mu - mu_sample.max()).reshape(-1, 1) / sigma
    normal = torch.distributions.Normal(torch.zeros_like(u), torch.ones_like(u))
    ei = sigma * (torch.exp(normal.log_prob(u)) + u * normal.cdf(u))
    return ei.clamp_min(0)


def upper_confidence_bound(gp, X, kappa=2):
    """:3635-3637."""
    mu, var = gp.posterior(X)
    return mu + kappa * var.clamp_min(0).sqrt()


def acquisition_fun(gp, X, X_train, acq_fn, *args):
    assert acq_fn in ("ei", "ucb")
    return expected_improvement(gp, X, X_train) if acq_fn == "ei" else upper_confidence_bound(gp, X, *args)


def peak_local_max(img, min_distance=1, threshold_rel=None, num_peaks=np.inf):
    """skimage.feature.peak_local_max with its defaults (exclude_border = min_distance): coordinates of the local maxima of a
    (2*min_distance+1)^2 neighbourhood above threshold_rel * max, strongest first."""
    from scipy import ndimage
    img = np.asarray(img, dtype=np.float64)
    size = 2 * min_distance + 1
    mx = ndimage.maximum_filter(img, size=size, mode="nearest")
    thr = img.min() if threshold_rel is None else threshold_rel * img.max()
    mask = (img == mx) & (img > thr)
    b = min_distance
    if b > 0:
        mask[:b, :] = False; mask[-b:, :] = False; mask[:, :b] = False; mask[:, -b:] = False
    coords = np.argwhere(mask)
    order = np.argsort(-img[mask], kind="stable")
    coords = coords[order]
    if np.isfinite(num_peaks):
        coords = coords[:int(num_peaks)]
    return coords


def find_candidates(gp, X_, samples, acq_fn="ei", grid=100):
    """:3649-3684: peaks of the acquisition on the grid, each refined by L-BFGS in the sigmoid parametrisation of [0, 1]^2."""
    import torch
    with torch.no_grad():
        acq = acquisition_fun(gp, X_, samples, acq_fn)
    acq = acq.cpu().numpy().reshape(grid, grid)
    peaks = peak_local_max(acq, min_distance=5, threshold_rel=0.1, num_peaks=4)
    global_max = np.array(np.unravel_index(np.argmax(acq, axis=None), acq.shape)).reshape(1, -1)
    peaks = np.unique(np.append(peaks.reshape(-1, 2), global_max, axis=0), axis=0)
    X_init = X_[np.ravel_multi_index(peaks.transpose(), acq.shape)]
    candidates, improvement = [], []
    for i in range(len(X_init[:4])):
        x0 = X_init[i].unsqueeze(0).double().clamp(1e-6, 1 - 1e-6)
        u = torch.log(x0) - torch.log1p(-x0)                      # inverse of the sigmoid transform (transform_to(interval(0, 1)).inv)
        u = u.clone().detach().requires_grad_(True)
        minimizer = torch.optim.LBFGS([u], line_search_fn="strong_wolfe")

        def closure():
            minimizer.zero_grad()
            y = -acquisition_fun(gp, torch.sigmoid(u), samples, acq_fn).sum()
            y.backward()
            return y

        minimizer.step(closure)
        X = torch.sigmoid(u)
        with torch.no_grad():
            improvement.append(float(acquisition_fun(gp, X, samples, acq_fn).sum()))
        candidates.append(X.detach().cpu())
    return candidates, improvement, acq


def normalize_X(X_unnorm, x1_logbounds, x2_logbounds):
    """:3687-3695: log10 of the hyper-parameters, then the affine map of the log-bounds box onto [0, 1]^2."""
    X = X_unnorm.clone().log10()
    X[:, 0] -= x1_logbounds[0]; X[:, 0] /= (x1_logbounds[1] - x1_logbounds[0])
    X[:, 1] -= x2_logbounds[0]; X[:, 1] /= (x2_logbounds[1] - x2_logbounds[0])
    return X


def unnormalize_X(X_norm, x1_logbounds, x2_logbounds):
    """:3698-3706: the inverse map, then 10 ** x."""
    X = X_norm.clone()
    X[:, 0] *= (x1_logbounds[1] - x1_logbounds[0]); X[:, 0] += x1_logbounds[0]
    X[:, 1] *= (x2_logbounds[1] - x2_logbounds[0]); X[:, 1] += x2_logbounds[0]
    return 10.0 ** X


def bo(bo_params, evaluate, n_rounds=20, gp_iters=2000, acq_fn="ei", verbose=True):
    """:3727-3880 without the plots.  bo_params: {name1: {logbounds: [lo, hi], candidates: [...]}, name2: {...}} (the reference's JSON);
    evaluate(list of (p1, p2) tuples) -> list of (candidate, psnr) in any order, e.g. a closure over fanout.run_jobs.  Returns the
    history (X, Y) and the candidates proposed last."""
    import torch
    (n1, v1), (n2, v2) = list(bo_params.items())
    lb1, lb2 = v1["logbounds"], v2["logbounds"]
    g1 = torch.logspace(lb1[0], lb1[1], 100, dtype=torch.double)
    g2 = torch.logspace(lb2[0], lb2[1], 100, dtype=torch.double)
    G1, G2 = torch.meshgrid(g1, g2, indexing="ij")
    X_ = torch.stack([G1.reshape(-1), G2.reshape(-1)]).transpose(1, 0)
    candidates = list(itertools.product(v1["candidates"], v2["candidates"]))
    X, Y = [], []
    for rnd in range(n_rounds):
        got = [(tuple(c), float(y)) for c, y in evaluate(list(candidates)) if not math.isnan(float(y))]       # NaN fits are dropped (:3777-3781)
        if verbose:
            print(); print("%s      %s       psnr" % (n1, n2))
            for c, y in got:
                print("%.6f  %.6f  %.6f" % (c[0], c[1], y))
        X += [c for c, _ in got]; Y += [y for _, y in got]
        if not X:
            raise RuntimeError("bo: every fit of the first round failed")
        X_train = normalize_X(torch.tensor(np.array(X), dtype=torch.float64), lb1, lb2)
        Y_train = torch.tensor(np.array(Y), dtype=torch.float64)
        gp = train_gp(X_train, Y_train, gp_iters, verbose=False)
        X_test = normalize_X(X_, lb1, lb2)
        cands, exp_imp, _ = find_candidates(gp, X_test, X_train, acq_fn)
        cands = torch.unique(torch.cat(cands).cpu(), dim=0)
        candidates = [tuple(map(float, c)) for c in unnormalize_X(cands, lb1, lb2).numpy()]
        if verbose:
            print("round %d: best %.4f at %s; next %s" % (rnd, max(Y), X[int(np.argmax(Y))], candidates))
    return X, Y, candidates
