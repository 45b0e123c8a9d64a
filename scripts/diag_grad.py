"""Diagnostic (GPU box): per-layer gradient error of the full den net vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
import mfvi_dip_mia_amd as M

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = O.make_net(size, size)
seed, step = 1, 3
mu, rho, bnp = O.init_params(net, seed)
conv, bn, n_vi, n_bnp = O.net_table(net)
g = O.normal_fill(seed, 2, 7, 0, 0, n_bnp)
for c, off in bn:
    bnp[off:off + c] = 1.0 + 0.1 * g[off:off + c]; bnp[off + c:off + 2 * c] = 0.1 * g[off + c:off + 2 * c]
P, zin, out_id, names = M.skip_program(size, size)
plan = P.compile(zin, out_id, 1)
z = (0.1 * O.uniform_fill(seed, 0, 0, 0, 16 * size * size)).reshape(16, size, size)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
d_mu, d_rho, d_bn, d_z = dev(mu), dev(rho), dev(bnp), dev(z)
out = plan.forward(d_mu, d_rho, d_bn, d_z, seed, step, 0, 1)
ref, tape = O.net_forward(net, mu, rho, bnp, z, seed, step, 0)
rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))
print("out err", rel(out.cpu().numpy()[0], ref))
dout = O.normal_fill(seed, 2, 9, 0, 0, out.numel()).reshape(tuple(out.shape))
dmu = torch.zeros_like(d_mu); drho = torch.zeros_like(d_rho); dbn = torch.zeros_like(d_bn)
plan.backward(d_mu, d_rho, d_bn, d_z, seed, step, 0, 1, dev(dout), dmu, drho, dbn)
a, b, c_, _ = tape.backward(dout[0], n_vi, n_bnp)
gm, gr, gb = dmu.cpu().numpy(), drho.cpu().numpy(), dbn.cpu().numpy()
# tensor ids per layer in module order
order = []
def walk(i):
    order.extend([names[i]["skip"], names[i]["d1"], names[i]["d2"]])
    if i + 1 in names: walk(i + 1)
    order.extend([names[i]["up"], names[i]["up1"]])
walk(0)
for lid, row in enumerate(conv):
    cin, cout, k, s, wo, bo = [int(v) for v in row]
    sl = slice(wo, bo + cout)
    fe = ""
    if lid < len(order):
        y_ref, m_ref, r_ref = tape.conv_out(lid)
        y_gpu = plan.read_tensor(order[lid], 0, 0).cpu().numpy().reshape(y_ref.shape)
        fe = "fwd %.1e" % rel(y_gpu, y_ref)
    print("L%02d %3d->%3d k%d s%d  dmu %.1e  drho %.1e  db %.1e  %s" % (lid, cin, cout, k, s, rel(gm[sl], a[sl]), rel(gr[sl], b[sl]), rel(gm[bo:bo+cout], a[bo:bo+cout]), fe))
for i, (c, off) in enumerate(bn):
    print("BN%02d C%3d dgamma %.1e dbeta %.1e" % (i, c, rel(gb[off:off+c], c_[off:off+c]), rel(gb[off+c:off+2*c], c_[off+c:off+2*c])))
print("---- intermediate gradients: ga (d/d BN output) per conv output tensor, and dy formed from GPU data in float64 ----")
for lid in range(len(order)):
    tid = order[lid]
    ga_ref = tape.conv_grad(lid, 1); dy_ref = tape.conv_grad(lid, 0)
    if ga_ref is None: continue
    ga = plan.read_tensor(tid, 0, 1).cpu().numpy().astype(np.float64).reshape(P.tensors[tid]["C"], -1)
    y = plan.read_tensor(tid, 0, 0).cpu().numpy().astype(np.float64).reshape(ga.shape)
    st = plan.read_tensor(tid, 0, 2).cpu().numpy(); bs = plan.read_tensor(tid, 0, 3).cpu().numpy()
    n = ga.shape[1]
    mean = st[:, 0] / n; var = st[:, 1] / n - mean ** 2; rstd = 1 / np.sqrt(var + 1e-5)
    xh = (y - mean[:, None]) * rstd[:, None]
    bdesc = [b for b in P.bns if b["tensor"] == tid][0]
    gam = bnp[bdesc["off"]:bdesc["off"] + bdesc["C"]].astype(np.float64)
    dy = gam[:, None] * rstd[:, None] * (ga - bs[:, 0:1] / n - xh * bs[:, 1:2] / n)
    dy_exact_sums = gam[:, None] * rstd[:, None] * (ga - ga.mean(1, keepdims=True) - xh * (ga * xh).mean(1, keepdims=True))
    ga_ref = ga_ref.reshape(ga.shape); dy_ref = dy_ref.reshape(ga.shape)
    print("L%02d ga %.1e  bsum0 %.1e bsum1 %.1e  dy %.1e  dy(exact sums) %.1e | max|ga| %.2e" % (
        lid, rel(ga, ga_ref), rel(bs[:, 0], ga.sum(1)), rel(bs[:, 1], (ga * xh).sum(1)), rel(dy, dy_ref), rel(dy_exact_sums, dy_ref), np.abs(ga_ref).max()))
print("---- where are the ga errors of the last BN'd tensor? ----")
for lid in (len(order) - 1, len(order) - 2, 0):
    tid = order[lid]
    C_ = P.tensors[tid]["C"]; Hh = P.tensors[tid]["H"]; Ww = P.tensors[tid]["W"]
    ga = plan.read_tensor(tid, 0, 1).cpu().numpy().astype(np.float64)
    y = plan.read_tensor(tid, 0, 0).cpu().numpy().astype(np.float64)
    st = plan.read_tensor(tid, 0, 2).cpu().numpy(); n = Hh * Ww
    mean = st[:, 0] / n; var = st[:, 1] / n - mean ** 2; rstd = 1 / np.sqrt(var + 1e-5)
    bdesc = [b for b in P.bns if b["tensor"] == tid][0]
    gam = bnp[bdesc["off"]:bdesc["off"] + C_].astype(np.float64); bet = bnp[bdesc["off"] + C_:bdesc["off"] + 2 * C_].astype(np.float64)
    v = (y - mean[:, None, None]) * (rstd * gam)[:, None, None] + bet[:, None, None]
    ga_ref = tape.conv_grad(lid, 1).reshape(ga.shape)
    err = np.abs(ga - ga_ref); thr = 1e-4 * np.abs(ga_ref).max()
    bad = np.argwhere(err > thr)
    print("L%02d: %d of %d elements off by > 1e-4 max; " % (lid, len(bad), ga.size))
    for (c, r, q) in bad[:12]:
        print("   c=%d r=%d q=%d  gpu %.5f ref %.5f ratio %.3f  v(BN out)=%.3e" % (c, r, q, ga[c, r, q], ga_ref[c, r, q], ga[c, r, q] / ga_ref[c, r, q] if ga_ref[c, r, q] else 0, v[c, r, q]))
