"""Public surface."""
from . import _lib
from ._build import build
from .program import Plan, Program, skip_program
from . import engine

__all__ = ["build", "Plan", "Program", "skip_program", "_lib", "engine"]
