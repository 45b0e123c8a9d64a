"""Builds libmfvi_hip.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmfvi_hip.so")
SOURCES = ["conv_fwd.hip", "conv_bwd_data.hip", "conv_bwd_weight.hip", "conv_mfma.hip", "conv_rp.hip", "conv_x6.hip", "conv_bwd_x6.hip", "conv_bwd_x6s.hip", "conv_small.hip", "conv_1x1.hip", "conv_bww_mfma.hip", "conv_bww_x6.hip", "elementwise.hip", "losses.hip", "radon.hip", "plan.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file extras.  conv_bwd_x6: no SLP vectorisation — packed fp32 instructions (v_pk_add / v_pk_fma) beside a matrix stream cost more issue time
# than the scalar ones they replace (MI355X_MICROARCH.md; measured on this kernel: 68->32 @128^2 88.0 -> 78.8 us); the explicit pairs of the
# bf16 split (common.h) stay packed
# (the other two bf16x6 kernels: 1-5 % on the same A/B — backward-weight 36->16 85.6 -> 81.5 us, forward 132->64 59.5 -> 58.6)
FILE_FLAGS = {"conv_bwd_x6.hip": ["-fno-slp-vectorize"], "conv_bwd_x6s.hip": ["-fno-slp-vectorize"], "conv_bww_x6.hip": ["-fno-slp-vectorize"], "conv_x6.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mfvi_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [cc] + FLAGS + FILE_FLAGS.get(src, []) + os.environ.get("MFVI_EXTRA_FLAGS", "").split() + ["-c", os.path.join(CSRC, src), "-o", obj]     # e.g. -DPRODUCER_PRIO=1 for experiments
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return obj
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, SOURCES))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
