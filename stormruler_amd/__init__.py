"""stormruler_amd -- MI355X-native Krylov backend for StormRuler's Storm::Solvers hot path.

``stormruler_amd.mesh`` (host-side face graphs) imports without a GPU or the HIP library;
everything else goes through ``libstorm_hip.so`` and fails loudly when it is missing.
"""
from . import mesh  # noqa: F401

__all__ = ["mesh", "api", "load"]


def load():
    """Import the device API (raises ImportError if libstorm_hip.so is not built)."""
    from . import api

    return api


def __getattr__(name):
    if name == "api":
        import importlib

        return importlib.import_module(".api", __name__)
    raise AttributeError(name)
