import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import mfvi_dip_mia_amd as M
def relerr(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
for mode, H, W in [("nearest", 35, 45), ("nearest", 33, 47), ("nearest", 37, 43), ("nearest", 35, 45), ("bilinear", 35, 45)]:
    for seed in (3, 4, 5, 6):
        torch.manual_seed(seed)
        mk = lambda: M.get_net(8, 'skip', 'reflection', mode, n_channels=2, skip_n33d=[8, 16, 16], skip_n33u=[8, 16, 16], skip_n11=4, num_scales=3)
        ref = mk(); base = mk(); base.load_state_dict(ref.state_dict())
        hip = M.FusedNet(base, device=torch.device('cuda'), seed=1)
        x = torch.rand(1, 8, H, W) * 0.1
        o_ref = ref(x); o_hip = hip(x.cuda())
        g = torch.randn_like(o_ref)
        o_ref.backward(g); o_hip.backward(g.cuda())
        cr = [m for m in ref.modules() if isinstance(m, torch.nn.Conv2d)]
        ch = [m for m in hip.modules() if isinstance(m, torch.nn.Conv2d)]
        errs = [relerr(b.weight.grad.cpu().numpy(), a.weight.grad.numpy()) for a, b in zip(cr, ch)]
        # float64 reference of the same net: how far is torch fp32 itself from it?
        ref64 = mk().double(); ref64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
        o64 = ref64(x.double()); o64.backward(g.double())
        c64 = [m for m in ref64.modules() if isinstance(m, torch.nn.Conv2d)]
        e_t = [relerr(a.weight.grad.numpy().astype(np.float64), c.weight.grad.numpy()) for a, c in zip(cr, c64)]
        e_h = [relerr(b.weight.grad.cpu().numpy().astype(np.float64), c.weight.grad.numpy()) for b, c in zip(ch, c64)]
        print(mode, H, W, "seed", seed, "fwd %.1e" % relerr(o_hip.detach().cpu().numpy(), o_ref.detach().numpy()),
              "max grad err hip-vs-torch32 %.1e | torch32-vs-f64 %.1e | hip-vs-f64 %.1e" % (max(errs), max(e_t), max(e_h)))
