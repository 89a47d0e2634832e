"""muscato_amd -- MI355X-native seed-and-extend hot path of muscato (screen + confirm).

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/muscato_hip.h) and
the host-side mirror of the reference interface (api.Config / api.Engine).  Importing the
package does not load the HIP library; creating an Engine does, and fails loudly if
libmuscato_hip.so is missing (no CPU fallback).
"""
from .api import Config, Engine, MuscatoError, gather, sorted_hits  # noqa: F401

__all__ = ["Config", "Engine", "MuscatoError", "gather", "sorted_hits"]
