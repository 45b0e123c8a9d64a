"""KL reported by the fused update against mfvi_kl on the same parameters (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mfvi_dip_mia_amd as M
L = M._lib
eng = M.engine.ElboEngine(64, 64, task="den", K=1, input_depth=16, seed=3, autotune=False, temp=5.6e-7, sigma=1.46e-5, lr=1e-3)
eng.set_target(torch.rand(64, 64))
lib = L.lib()
N = int(os.environ.get("N", "6")); bad = 0; worst = 0.0
for it in range(N):
    ref = torch.zeros(1, dtype=torch.float64, device="cuda")
    L.check(lib.mfvi_kl(L.ptr(eng.mu), L.ptr(eng.rho), eng.n_vi, 0.0, eng.prior_sigma, L.ptr(ref), L.stream_ptr()))
    eng.step()
    nll, kl, loss = eng.losses()
    rel = abs(kl - float(ref)) / abs(float(ref)); worst = max(worst, rel)
    if rel > 1e-9:
        bad += 1; print(it, "kl fused %.10g  mfvi_kl %.10g  rel %.3e" % (kl, float(ref), rel))
print("iterations %d, mismatches %d, worst rel %.3e" % (N, bad, worst))
