"""ctypes binding of libgip_raster.so / libgip_knn.so (the C-ABI declared in include/gip_raster.h, include/gip_knn.h).

The product path has NO CPU fallback: if the HIP library is missing, loading raises ImportError with build
instructions (`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

GIP_MAX_VIEWS = 16
GIP_RECORD_BYTES = 64
GIP_PARTIAL_FLOATS = 12
GIP_OK = 0

_vp = ctypes.c_void_p


class GipRasterConfig(ctypes.Structure):
    _fields_ = [("P", ctypes.c_int32), ("V", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("sh_degree", ctypes.c_int32), ("sh_coeffs", ctypes.c_int32), ("prefiltered", ctypes.c_int32),
                ("debug", ctypes.c_int32), ("scale_modifier", ctypes.c_float),
                ("tanfovx", ctypes.c_float * GIP_MAX_VIEWS), ("tanfovy", ctypes.c_float * GIP_MAX_VIEWS),
                ("capacity", ctypes.c_uint64), ("exact_lists", ctypes.c_int32), ("forward_only", ctypes.c_int32),
                ("sh_scalar", ctypes.c_int32)]


class GipRasterInputs(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                                   "cov3D_precomp", "viewmatrix", "projmatrix", "campos", "bg")]


class GipRasterOutputs(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("color", "radii", "depth", "alpha", "host_header")]


class GipRasterGradsIn(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("dL_dcolor", "dL_ddepth", "dL_dalpha", "alpha", "color", "depth")]


class GipRasterGradsOut(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("dL_dmeans3D", "dL_dmeans2D", "dL_dshs", "dL_dcolors_precomp", "dL_dopacities",
                                   "dL_dscales", "dL_drotations", "dL_dcov3D_precomp")]


class GipRasterStateLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_size_t) for n in ("header", "records", "inst_offset", "tile_count", "tile_start",
                                                "tile_cursor", "tile_count_b", "inst_slot", "block_sums", "block_offset", "keys", "n_contrib",
                                                "final_T", "tile_order", "seg_start", "ckpt_start", "seg_tile", "checkpoints", "sh_colors", "total")] + \
               [(n, ctypes.c_uint32) for n in ("tiles_x", "tiles_y", "num_blocks", "reserved")]


_raster = None
_knn = None


def _missing(name):
    return ImportError(
        "gaussianip_amd: %s not found in %s. The HIP extension must be built for gfx950 first "
        "(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gaussianip_amd/csrc`). "
        "There is no CPU fallback for the product path." % (name, LIB_DIR))


call_counts = {}       # C-ABI entry point -> number of calls through this module (tests assert that the HIP path ran)


class _Counted:
    """The loaded library with a per-symbol call counter in front of every entry point."""

    def __init__(self, lib):
        self._lib = lib
        self._wrapped = {}

    def __getattr__(self, name):
        w = self._wrapped.get(name)
        if w is None:
            fn = getattr(self._lib, name)

            def w(*args, _fn=fn, _name=name):
                call_counts[_name] = call_counts.get(_name, 0) + 1
                return _fn(*args)
            self._wrapped[name] = w
        return w


def raster_lib():
    global _raster
    if _raster is None:
        path = os.path.join(LIB_DIR, os.environ.get("GIP_RASTER_LIB", "libgip_raster.so"))
        if not os.path.exists(path):
            raise _missing("libgip_raster.so")
        lib = ctypes.CDLL(path)
        lib.gip_abi_version.restype = ctypes.c_int
        lib.gip_status_string.restype = ctypes.c_char_p
        lib.gip_status_string.argtypes = [ctypes.c_int]
        lib.gip_raster_state_bytes.restype = ctypes.c_size_t
        lib.gip_raster_state_bytes.argtypes = [ctypes.POINTER(GipRasterConfig)]
        lib.gip_raster_scratch_bytes.restype = ctypes.c_size_t
        lib.gip_raster_scratch_bytes.argtypes = [ctypes.POINTER(GipRasterConfig)]
        lib.gip_raster_state_layout.restype = ctypes.c_int
        lib.gip_raster_state_layout.argtypes = [ctypes.POINTER(GipRasterConfig), ctypes.POINTER(GipRasterStateLayout)]
        lib.gip_raster_forward.restype = ctypes.c_int
        lib.gip_raster_forward.argtypes = [ctypes.POINTER(GipRasterConfig), ctypes.POINTER(GipRasterInputs),
                                           ctypes.POINTER(GipRasterOutputs), _vp, ctypes.c_size_t, _vp]
        lib.gip_raster_backward.restype = ctypes.c_int
        lib.gip_raster_backward.argtypes = [ctypes.POINTER(GipRasterConfig), ctypes.POINTER(GipRasterInputs),
                                            ctypes.POINTER(GipRasterGradsIn), _vp, ctypes.c_size_t, _vp,
                                            ctypes.c_size_t, ctypes.POINTER(GipRasterGradsOut), _vp]
        lib.gip_raster_forward_profiled.restype = ctypes.c_int
        lib.gip_raster_forward_profiled.argtypes = lib.gip_raster_forward.argtypes + [ctypes.POINTER(ctypes.c_float)]
        lib.gip_raster_backward_profiled.restype = ctypes.c_int
        lib.gip_raster_backward_profiled.argtypes = lib.gip_raster_backward.argtypes + [ctypes.POINTER(ctypes.c_float)]
        lib.gip_raster_read_header.restype = ctypes.c_int
        lib.gip_raster_read_header.argtypes = [_vp, _vp, _vp]
        lib.gip_raster_mark_visible.restype = ctypes.c_int
        lib.gip_raster_mark_visible.argtypes = [ctypes.c_int32, _vp, _vp, _vp, _vp, _vp]
        _raster = _Counted(lib)
    return _raster


def knn_lib():
    global _knn
    if _knn is None:
        path = os.path.join(LIB_DIR, "libgip_knn.so")
        if not os.path.exists(path):
            raise _missing("libgip_knn.so")
        lib = ctypes.CDLL(path)
        lib.gip_knn_workspace_bytes.restype = ctypes.c_size_t
        lib.gip_knn_workspace_bytes.argtypes = [ctypes.c_int32]
        lib.gip_knn_mean_dist2.restype = ctypes.c_int
        lib.gip_knn_mean_dist2.argtypes = [ctypes.c_int32, _vp, _vp, _vp, ctypes.c_size_t, _vp]
        lib.gip_knn_workspace_bytes_mode.restype = ctypes.c_size_t
        lib.gip_knn_workspace_bytes_mode.argtypes = [ctypes.c_int32, ctypes.c_int32]
        lib.gip_knn_mean_dist2_mode.restype = ctypes.c_int
        lib.gip_knn_mean_dist2_mode.argtypes = [ctypes.c_int32, _vp, _vp, _vp, ctypes.c_size_t, ctypes.c_int32, _vp]
        _knn = _Counted(lib)
    return _knn


_model = None


class GipAdamGroup(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("step", ctypes.c_void_p), ("n", ctypes.c_int64), ("lr", ctypes.c_float), ("reserved", ctypes.c_int32)]


class GipGatherTensor(ctypes.Structure):
    _fields_ = [("old_rows", ctypes.c_void_p), ("new_rows", ctypes.c_void_p), ("dst", ctypes.c_void_p),
                ("row_bytes", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def model_lib():
    global _model
    if _model is None:
        path = os.path.join(LIB_DIR, "libgip_model.so")
        if not os.path.exists(path):
            raise _missing("libgip_model.so")
        lib = ctypes.CDLL(path)
        lib.gip_gather_rows.restype = ctypes.c_int
        lib.gip_gather_rows.argtypes = [ctypes.POINTER(GipGatherTensor), ctypes.c_int32, _vp, ctypes.c_int64, ctypes.c_int64, _vp]
        lib.gip_adam_step.restype = ctypes.c_int
        lib.gip_adam_step.argtypes = [ctypes.POINTER(GipAdamGroup), ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp, _vp]
        lib.gip_openpose_draw.restype = ctypes.c_int
        lib.gip_openpose_draw.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t, _vp]
        lib.gip_openpose_workspace_bytes.restype = ctypes.c_size_t
        lib.gip_openpose_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
        lib.gip_pack_bucket.restype = ctypes.c_int
        lib.gip_pack_bucket.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.c_int32, _vp,
                                        ctypes.c_int32, ctypes.c_int64, _vp, _vp]
        lib.gip_max_bucket.restype = ctypes.c_int
        lib.gip_max_bucket.argtypes = [_vp, ctypes.c_int32, ctypes.c_int64, _vp, ctypes.c_int64, _vp, _vp]
        lib.gip_unpack_bucket.restype = ctypes.c_int
        lib.gip_unpack_bucket.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.c_int32, _vp,
                                          ctypes.c_int64, _vp, ctypes.c_float, _vp]
        lib.gip_sparsity_workspace_bytes.restype = ctypes.c_size_t
        lib.gip_sparsity_workspace_bytes.argtypes = []
        lib.gip_sparsity_loss_forward.restype = ctypes.c_int
        lib.gip_sparsity_loss_forward.argtypes = [_vp, ctypes.c_int64, _vp, _vp]
        lib.gip_sparsity_loss_backward.restype = ctypes.c_int
        lib.gip_sparsity_loss_backward.argtypes = [_vp, ctypes.c_int64, _vp, ctypes.c_float, _vp, _vp, _vp]
        lib.gip_activate_gaussians.restype = ctypes.c_int
        lib.gip_activate_gaussians.argtypes = [_vp, _vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp]
        lib.gip_activate_gaussians_backward.restype = ctypes.c_int
        lib.gip_activate_gaussians_backward.argtypes = [_vp] * 6 + [ctypes.c_int64, _vp, _vp, _vp, _vp]
        lib.gip_densify_stats.restype = ctypes.c_int
        lib.gip_densify_stats.argtypes = [_vp, ctypes.c_int32, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp]
        _model = _Counted(lib)
    return _model


_nn = None


def nn_lib():
    global _nn
    if _nn is None:
        path = os.path.join(LIB_DIR, os.environ.get("GIP_NN_LIB", "libgip_nn.so"))
        if not os.path.exists(path):
            raise _missing("libgip_nn.so")
        lib = ctypes.CDLL(path)
        lib.gip_gn_workspace_bytes.restype = ctypes.c_size_t
        lib.gip_gn_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
        lib.gip_gn_silu_forward.restype = ctypes.c_int
        lib.gip_gn_silu_forward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_float, ctypes.c_int32, _vp, ctypes.c_int32, _vp,
                                            ctypes.c_size_t, _vp]
        lib.gip_gn_silu_backward.restype = ctypes.c_int
        lib.gip_gn_silu_backward.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int64,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32, _vp,
                                             ctypes.c_size_t, _vp]
        lib.gip_gn_silu_backward_accum.restype = ctypes.c_int
        lib.gip_gn_silu_backward_accum.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int64,
                                                   ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32, _vp, _vp,
                                                   ctypes.c_size_t, _vp]
        lib.gip_conv3x3s2_dgrad_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3s2_dgrad_nhwc_f16.argtypes = [_vp, _vp, _vp] + [ctypes.c_int32] * 5 + [_vp]
        lib.gip_upsample2x_conv3x3_nhwc_f16.restype = ctypes.c_int
        lib.gip_upsample2x_conv3x3_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 5 + [_vp]
        lib.gip_winograd_input_f16.restype = ctypes.c_int
        lib.gip_winograd_input_f16.argtypes = [_vp, _vp] + [ctypes.c_int32] * 4 + [_vp]
        lib.gip_winograd_output_f16.restype = ctypes.c_int
        lib.gip_winograd_output_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 4 + [_vp]
        lib.gip_winograd_output_stats_f16.restype = ctypes.c_int
        lib.gip_winograd_output_stats_f16.argtypes = [_vp, _vp, _vp, _vp, _vp] + [ctypes.c_int32] * 4 + [_vp]
        lib.gip_conv3x3_fewch_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3_fewch_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 7 + [_vp]
        lib.gip_conv3x3_c3_fwd_stats_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3_c3_fwd_stats_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 4 + [_vp, _vp]
        lib.gip_conv3x3_c3_fwd_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3_c3_fwd_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 4 + [_vp]
        lib.gip_conv3x3_c3_dgrad_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3_c3_dgrad_nhwc_f16.argtypes = [_vp, _vp, _vp] + [ctypes.c_int32] * 4 + [_vp]
        lib.gip_gn_silu_backward_sums.restype = ctypes.c_int
        lib.gip_gn_silu_backward_sums.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int64,
                                                  ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32, _vp, _vp,
                                                  ctypes.c_int32, _vp, ctypes.c_size_t, _vp]
        lib.gip_conv3x3_gnbwd_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3_gnbwd_nhwc_f16.argtypes = [_vp, _vp, _vp] + [ctypes.c_int32] * 5 + [_vp, _vp, _vp, _vp, _vp, ctypes.c_int32,
                                                   ctypes.c_int32, _vp, ctypes.c_int32, _vp, _vp]
        lib.gip_gn_silu_forward_stats.restype = ctypes.c_int
        lib.gip_gn_silu_forward_stats.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
                                                  ctypes.c_int32, ctypes.c_float, ctypes.c_int32, _vp, ctypes.c_int32, _vp,
                                                  ctypes.c_int32, _vp]
        lib.gip_conv3x3_stats_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3_stats_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp, _vp] + [ctypes.c_int32] * 5 + [_vp, _vp]
        lib.gip_conv3x3_stats_ws_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3_stats_ws_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp, _vp] + [ctypes.c_int32] * 5 + [_vp, ctypes.c_int32, _vp, ctypes.c_size_t, _vp]
        lib.gip_linear_stats_f16.restype = ctypes.c_int
        lib.gip_linear_stats_f16.argtypes = [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _vp, _vp]
        lib.gip_add_bias_residual.restype = ctypes.c_int
        lib.gip_add_bias_residual.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int32, _vp]
        lib.gip_geglu.restype = ctypes.c_int
        lib.gip_geglu.argtypes = [_vp, _vp, ctypes.c_int64, ctypes.c_int32, _vp]
        lib.gip_cat2_stats_f16.restype = ctypes.c_int
        lib.gip_cat2_stats_f16.argtypes = [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _vp]
        lib.gip_layernorm_f16.restype = ctypes.c_int
        lib.gip_layernorm_f16.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, _vp]
        lib.gip_lpips_layer_blocks.restype = ctypes.c_int32
        lib.gip_lpips_layer_blocks.argtypes = [ctypes.c_int32, ctypes.c_int64]
        lib.gip_lpips_layer_forward.restype = ctypes.c_int
        lib.gip_lpips_layer_forward.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _vp]
        lib.gip_lpips_layer_backward.restype = ctypes.c_int
        lib.gip_lpips_layer_backward.argtypes = [_vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, _vp]
        lib.gip_attention_fwd_strided2_f16.restype = ctypes.c_int
        lib.gip_attention_fwd_strided2_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 5 + [ctypes.c_float, _vp, _vp, ctypes.c_int32,
                                                       ctypes.c_float, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp]
        lib.gip_attention_fwd_f16.restype = ctypes.c_int
        lib.gip_attention_fwd_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 5 + [ctypes.c_float, _vp, _vp, ctypes.c_int32, ctypes.c_float, _vp]
        lib.gip_attention_fwd_strided_f16.restype = ctypes.c_int
        lib.gip_attention_fwd_strided_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 5 + [ctypes.c_float, _vp, _vp, ctypes.c_int32, ctypes.c_float, ctypes.c_int32, ctypes.c_int32, _vp]
        lib.gip_conv3x3s2_stats_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3s2_stats_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 7 + [_vp, _vp]
        lib.gip_conv3x3s2_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3s2_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp] + [ctypes.c_int32] * 7 + [_vp, ctypes.c_size_t, _vp]
        lib.gip_linear_row_parts.restype = ctypes.c_int32
        lib.gip_linear_row_parts.argtypes = [ctypes.c_int64, ctypes.c_int32]
        lib.gip_linear_rows_f16.restype = ctypes.c_int
        lib.gip_linear_rows_f16.argtypes = [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _vp, _vp]
        lib.gip_linear_ln_f16.restype = ctypes.c_int
        lib.gip_linear_ln_f16.argtypes = [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp,
                                          ctypes.c_int32, ctypes.c_float, _vp]
        lib.gip_linear_f16.restype = ctypes.c_int
        lib.gip_linear_f16.argtypes = [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp]
        lib.gip_conv3x3_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp, _vp] + [ctypes.c_int32] * 5 + [_vp, ctypes.c_size_t, _vp]
        _i64p = ctypes.POINTER(ctypes.c_int64)
        lib.gip_image_prep_f16.restype = ctypes.c_int
        lib.gip_image_prep_f16.argtypes = [_vp] + [ctypes.c_int32] * 4 + [_vp, _vp]
        lib.gip_image_prep_backward_f16.restype = ctypes.c_int
        lib.gip_image_prep_backward_f16.argtypes = [_vp] + [ctypes.c_int32] * 4 + [_vp, _vp]
        lib.gip_latent_sample_f16.restype = ctypes.c_int
        lib.gip_latent_sample_f16.argtypes = [_vp, _i64p, _vp, _vp, _vp, _vp, ctypes.c_float] + [ctypes.c_int32] * 5 + [_vp, _vp, _vp]
        lib.gip_latent_sample_backward_f16.restype = ctypes.c_int
        lib.gip_latent_sample_backward_f16.argtypes = [_vp, _i64p, _vp, _vp, ctypes.c_float] + [ctypes.c_int32] * 4 + [_vp, _vp]
        lib.gip_anpg_loss_f16.restype = ctypes.c_int
        lib.gip_anpg_loss_f16.argtypes = [_vp, _i64p, _vp, _i64p, _vp, _vp] + [ctypes.c_int32] * 4 + [ctypes.c_float, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_float, _vp, _vp, _vp, _vp, _vp]
        lib.gip_timestep_embedding_f16.restype = ctypes.c_int
        lib.gip_timestep_embedding_f16.argtypes = [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _vp, _vp]
        lib.gip_winograd_input_gn_f16.restype = ctypes.c_int
        lib.gip_winograd_input_gn_f16.argtypes = [_vp, _vp] + [ctypes.c_int32] * 4 + [_vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp,
                                                  ctypes.c_int32, _vp]
        lib.gip_gn_stats_from_partials.restype = ctypes.c_int
        lib.gip_gn_stats_from_partials.argtypes = [_vp, _vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                                   _vp, ctypes.c_int32, _vp, ctypes.c_int32, _vp]
        lib.gip_conv3x3_gnin_nhwc_f16.restype = ctypes.c_int
        lib.gip_conv3x3_gnin_nhwc_f16.argtypes = [_vp, _vp, _vp, _vp, _vp] + [ctypes.c_int32] * 5 + [_vp, _vp, _vp, _vp, ctypes.c_int32,
                                                  ctypes.c_int32, _vp, ctypes.c_int32, _vp, _vp]
        lib.gip_linear_batched_f16.restype = ctypes.c_int
        lib.gip_linear_batched_f16.argtypes = [_vp, _vp, _vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _vp]
        lib.gip_softmax_rows_f16.restype = ctypes.c_int
        lib.gip_softmax_rows_f16.argtypes = [_vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, _vp]
        lib.gip_softmax_rows_backward_f16.restype = ctypes.c_int
        lib.gip_softmax_rows_backward_f16.argtypes = [_vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, _vp]
        lib.gip_scale_cast_f16.restype = ctypes.c_int
        lib.gip_scale_cast_f16.argtypes = [_vp, _vp, ctypes.c_float, _vp, ctypes.c_int64, _vp]
        _nn = _Counted(lib)
    return _nn


def status_string(rc):
    return raster_lib().gip_status_string(int(rc)).decode()


RASTER_SYMBOLS = ["gip_abi_version", "gip_status_string", "gip_raster_state_bytes", "gip_raster_scratch_bytes",
                  "gip_raster_state_layout", "gip_raster_forward", "gip_raster_backward", "gip_raster_read_header",
                  "gip_raster_mark_visible", "gip_raster_forward_profiled", "gip_raster_backward_profiled"]
