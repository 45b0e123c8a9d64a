"""Public surface."""
from . import _lib
from ._build import build
from .program import Plan, Program, skip_program
from . import engine, sharding
from . import runner
from .nets import Concat, get_net, skip

__all__ = ["build", "Plan", "Program", "skip_program", "_lib", "engine", "sharding", "runner", "Concat", "get_net", "skip", "MeanFieldVI", "FusedNet", "Conv2dRT", "Conv2dLRT",
           "gaussian_nll", "gaussian_nll_inpainting"]


def __getattr__(name):          # bayes.py needs torch.nn at import: keep `import mfvi_dip_mia_amd` light
    if name in ("MeanFieldVI", "FusedNet", "Conv2dRT", "Conv2dLRT", "gaussian_nll", "gaussian_nll_inpainting"):
        from . import bayes
        return getattr(bayes, name)
    raise AttributeError(name)
