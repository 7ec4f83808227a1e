"""Host-side numerics the hot path needs (device-agnostic re-statements of gaussiansplatting/utils/*)."""
from .graphics import BasicPointCloud, focal2fov, fov2focal, getProjectionMatrix  # noqa: F401
from .general import (build_rotation, build_scaling_rotation, get_expon_lr_func, inverse_sigmoid,  # noqa: F401
                      strip_lowerdiag, strip_symmetric)
from .sh import C0, RGB2SH, SH2RGB, eval_sh  # noqa: F401
