#!/usr/bin/env python3
"""Markdown rows of DESIGN.md's results table from profiles/r04_bench_*.json (one bench line each)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [("cfg1", "den 128², K=4"), ("cfg2", "den 256², K=16"), ("cfg3", "SR ×4 512², depth 32, K=8"), ("cfg4", "CT 256², 45 angles, K=16"),
        ("cfg5", "den 512², K=64 (4 launches of 16), bf16 μ/ρ"), ("inp", "inpainting 256² (5×5 filters, no skips), K=16"),
        ("cfg2_k1", "cfg2 `--k 1`: the reference's own loop shape"), ("dropin", "`--mode dropin --k 1`: INTEGRATION.md loop, `torch.optim.AdamW` over 254 Parameters"),
        ("dropin_flat", "`--mode dropin --k 1 --flat-parameters`: `MeanFieldVI(flat_parameters=True)`")]
for key, what in rows:
    f = os.path.join(ROOT, "profiles", "r04_bench_%s.json" % key)
    if not os.path.exists(f):
        continue
    d = json.loads(open(f).read().strip().split("\n")[-1])
    r = d["roofline"]
    extra = "%d it/s" % round(d["elbo_iters_per_sec"])
    if d.get("elbo_iters_per_sec_with_bookkeeping"):
        extra += "; %d with bookkeeping" % round(d["elbo_iters_per_sec_with_bookkeeping"])
    if d.get("fwd_only_mc_passes_per_sec") and key.startswith("cfg") and "k1" not in key:
        extra += "; forward-only %d" % round(d["fwd_only_mc_passes_per_sec"])
    dom = "latency-bound" if r.get("alone", {}).get("frac", 0) < 0.2 else "%s: %.2f / %.2f" % ((r["kernel"].split(" of ")[0].replace("bwd_data", "bwd-data").replace("bwd_weight", "bwd-weight") + " " + r["kernel"].split(":")[1].split(",")[0].strip().replace("3x3 conv ", "")), r["alone"]["frac"], r["frac"])
    x6 = d.get("bf16x6_kernels")
    print("| %s | %s | **%.2f** | %d (%s) | %s | %s |" % (key, what, d["ms_per_step"], round(d["value"]), extra, dom, len(x6) if x6 is not None else "—"))
