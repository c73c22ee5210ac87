from . import lib  # noqa: F401  (raises if libmcmc_hip.so is missing: there is no CPU fallback)
