#!/usr/bin/env python3
"""profiles/traffic.json from a layer-traffic table (scripts/dev/layer_traffic.sh): HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate
rocprofv3 --pmc passes) of the forward / backward-data / backward-weight kernel of each 3x3 layer measured, stamped with the sha256 of the
kernel sources they were measured on (bench.py reports roofline.traffic only when the running library is built from the same sources).
usage: make_traffic_json.py profiles/r03_layer_traffic.txt [lib.so] > profiles/traffic.json"""
import hashlib, json, os, re, sys, glob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256():
    d = os.path.join(ROOT, "mfvi-dip-mia_amd", "csrc")
    files = sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))) + [os.path.join(ROOT, "include", "mfvi_hip.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def kernel_pass(name, ks):
    """which pass of the kxk layer under test a kernel name of the table belongs to (None: one of the 1x1 helper layers)"""
    if name.startswith("conv_fwd_x6_kernel"): return "fwd"
    if name.startswith("conv_bwd_x6_kernel") or name.startswith("conv_bwd_x6s_kernel"): return "bwd_data"
    if name.startswith("conv_bww_x6_kernel") or name.startswith("conv_bww_split_kernel"): return "bwd_weight"
    m = re.match(r"conv_rp_kernel<(\d)", name)
    if m: return "fwd" if m.group(1) == "0" else "bwd_data"
    m = re.match(r"conv_bww_mfma_kernel<(\d)", name)
    if m: return "bwd_weight" if int(m.group(1)) == ks else None
    m = re.match(r"conv_mfma_kernel<(\d), (\d), \d+, \d+, (\d)", name)
    if m and int(m.group(1)) == ks: return "fwd" if m.group(3) == "0" else "bwd_data"
    return None


entries = {}
desc = None
for line in open(sys.argv[1]):
    m = re.match(r"== layer (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)", line)
    if m:
        cin, cout, ks, st, H, W = [int(v) for v in m.groups()]
        desc = "%dx%d conv %d->%d @%dx%d s%d" % (ks, ks, cin, cout, H // st, W // st, st); cur_ks = ks
        continue
    m = re.match(r"(\S.*?)\s+\d+\s+rd\s+([\d.]+)\s+wr\s+([\d.]+)", line)
    if m and desc:
        p = kernel_pass(m.group(1).strip(), cur_ks)
        if p:
            entries.setdefault("%s:%s" % (p, desc), int(round((float(m.group(2)) + float(m.group(3))) * 1e6)))
out = {"csrc_sha256": csrc_sha256(),
       "source": "%s (scripts/dev/layer_traffic.sh, every kernel alone on the stream: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with "
                 "--kernel-trace only + a plain trace, joined by scripts/hbm_traffic.py; FETCH_SIZE doubled per MI355X_MICROARCH.md); bytes per launch, K = 16, "
                 "autotuned tilings.  csrc_sha256 = sha256 over csrc/*.hip, csrc/*.h and include/mfvi_hip.h of the build these counters were measured on.  "
                 "backward-data of the 3x3 stride-1 layers INCLUDES its fused fold (reads the raw input tensor, writes ga)" % os.path.relpath(sys.argv[1], ROOT),
       "entries": entries}
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    out["lib_sha256_of_the_measured_build"] = hashlib.sha256(open(sys.argv[2], "rb").read()).hexdigest()
print(json.dumps(out, indent=1))
