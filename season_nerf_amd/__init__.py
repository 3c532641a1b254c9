"""MI355X-native (gfx950) implementation of the Season-NeRF per-ray hot path behind the reference's own Python
call boundary: `T_NeRF` (network), `All_in_One_Eval` (ray evaluator).  All arithmetic runs in the HIP kernels of
`csrc/` through the C ABI of `include/season_nerf_hip.h`; importing this package never falls back to a CPU path."""
from . import _lib
from . import ops
from . import parallel
from . import raytable
from .training import FusedAdam, TrainEngine, create_solor_rays_uniform
from .network import T_NeRF, SineLayer
from .evaluator import All_in_One_Eval, sample_parameters, get_PV
from .render import (Quick_Run_Net, component_render_by_dir, component_render_by_P, get_imgs_from_Img_Dict, get_imgs_from_Img_Dict_t_step,
                     render_season_sweep, world_angle_2_local_vec, encode_time)

__all__ = ["T_NeRF", "SineLayer", "All_in_One_Eval", "sample_parameters", "get_PV", "Quick_Run_Net", "component_render_by_dir", "component_render_by_P",
           "get_imgs_from_Img_Dict", "get_imgs_from_Img_Dict_t_step", "render_season_sweep", "world_angle_2_local_vec",
           "encode_time", "parallel", "raytable", "FusedAdam", "TrainEngine", "create_solor_rays_uniform", "_lib"]
from .adaptive_loss import AdaptiveLossFunction  # noqa: E402,F401
from . import validation  # noqa: E402,F401
from .validation import DSM_Distance, eval_img, image_error  # noqa: E402,F401
from .trainer import GraphedTrainStep, Net_tool, T_NeRF_Net_Tool  # noqa: E402,F401
from .model_io import load_model, load_t_nerf, load_args_from_json, parse_time, render_novel_view  # noqa: E402,F401
from .synthetic import synthetic_state_dict, per_point_layer_shapes  # noqa: E402,F401
