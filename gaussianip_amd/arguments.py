"""Pipeline / optimisation parameter groups with the reference's field names and defaults.

Reference: gaussiansplatting/arguments/__init__.py:63-68 (PipelineParams), :70-88 (OptimizationParams); instantiated
with a throw-away ArgumentParser at threestudio/systems/GaussianIP.py:103-104 and :571 — the parser argument is
accepted and ignored (defaults only), optionally registering the same `--flag` options for CLI use.
Pinned by tests/golden/lr_schedule.npz (`opt`, `pipe` arrays).
"""


class _Group:
    _defaults = {}

    def __init__(self, parser=None, name=None, **overrides):
        for k, v in self._defaults.items():
            setattr(self, k, overrides.get(k, v))
        if parser is not None:
            grp = parser.add_argument_group(name or type(self).__name__)
            for k, v in self._defaults.items():
                try:
                    if isinstance(v, bool):
                        grp.add_argument("--" + k, default=v, action="store_true")
                    else:
                        grp.add_argument("--" + k, default=v, type=type(v))
                except Exception:   # option already registered on this parser
                    pass

    def extract(self, args):
        out = type(self)()
        for k in self._defaults:
            if hasattr(args, k):
                setattr(out, k, getattr(args, k))
        return out


class PipelineParams(_Group):
    _defaults = dict(convert_SHs_python=False, compute_cov3D_python=False, debug=False)

    def __init__(self, parser=None, **kw):
        super().__init__(parser, "Pipeline Parameters", **kw)


class OptimizationParams(_Group):
    _defaults = dict(iterations=3_200, position_lr_init=0.00005, position_lr_final=0.000025,
                     position_lr_delay_mult=0.5, position_lr_max_steps=30_000, feature_lr=0.0125, opacity_lr=0.01,
                     scaling_lr=0.005, rotation_lr=0.001, percent_dense=0.01, lambda_dssim=0.2,
                     densification_interval=100, opacity_reset_interval=3000, densify_from_iter=500,
                     densify_until_iter=15_000, densify_grad_threshold=0.0002)

    def __init__(self, parser=None, **kw):
        super().__init__(parser, "Optimization Parameters", **kw)
