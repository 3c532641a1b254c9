"""MI355X-native (gfx950) implementation of the Season-NeRF per-ray hot path behind the reference's own Python
call boundary: `T_NeRF` (network), `All_in_One_Eval` (ray evaluator).  All arithmetic runs in the HIP kernels of
`csrc/` through the C ABI of `include/season_nerf_hip.h`; importing this package never falls back to a CPU path."""
from . import _lib
from .network import T_NeRF, SineLayer
from .evaluator import All_in_One_Eval, sample_parameters

__all__ = ["T_NeRF", "SineLayer", "All_in_One_Eval", "sample_parameters", "_lib"]
