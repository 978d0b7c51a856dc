"""grand_plus_amd -- MI355X-native GFPush propagation-matrix precompute (GRAND+ hot path).

Only what the path needs lives here: `csrc/` (HIP kernels + the C ABI of
include/grandplus.h), the ctypes binding (`_native`), the host-side mirror of the
reference's `propagation.Graph` (`api.Graph`), the caller-side recipe helpers (`recipes`),
the multi-GPU seed-sharding driver (`sharded`), the tie-aware parity comparator (`parity`)
and the synthetic workload generator (`synth`).
"""
from .api import Graph, algorithmic_bytes          # noqa: F401
from .recipes import RECIPES, Recipe, make_coef    # noqa: F401

__all__ = ["Graph", "algorithmic_bytes", "RECIPES", "Recipe", "make_coef"]
