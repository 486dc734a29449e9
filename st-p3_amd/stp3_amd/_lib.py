"""ctypes binding of libstp3hip.so (the C ABI declared in include/stp3_hip.h).

The library is the product: there is no CPU or PyTorch fallback for the operators it
exports.  ``lib()`` raises if the shared object is missing or lacks a symbol, and every
operator in ``stp3_amd.ops`` raises if it is handed non-GPU tensors.
"""
import ctypes
import os

from . import profiling

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libstp3hip.so')

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int32 = ctypes.c_int32
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t


class LiftDims(ctypes.Structure):
    """struct stp3_lift_dims (include/stp3_hip.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in ('B', 'T', 'N', 'D', 'fH', 'fW', 'C', 'X', 'Y', 'Z')]

    @property
    def BT(self):
        return self.B * self.T

    @property
    def NPIX(self):
        return self.N * self.fH * self.fW

    @property
    def P(self):
        return self.NPIX * self.D

    @property
    def V(self):
        return self.X * self.Y * self.Z


class DwConvDims(ctypes.Structure):
    """struct stp3_dwconv_dims (include/stp3_hip.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in ('N', 'H', 'W', 'C', 'Ho', 'Wo', 'K', 'stride', 'pad_top', 'pad_left',
                                              'dtype')]


class WgradJob(ctypes.Structure):
    """struct stp3_wgrad_job (include/stp3_hip.h)."""
    _fields_ = [('partials', ctypes.c_void_p), ('dw', ctypes.c_void_p), ('numel', ctypes.c_int64), ('splits', ctypes.c_int32),
                ('reserved', ctypes.c_int32)]


class LayerNormDims(ctypes.Structure):
    """struct stp3_layernorm_dims (include/stp3_hip.h)."""
    _fields_ = [('rows', ctypes.c_int64), ('C', ctypes.c_int32), ('ldx', ctypes.c_int32), ('ldy', ctypes.c_int32),
                ('dtype', ctypes.c_int32), ('act', ctypes.c_int32), ('eps', ctypes.c_float)]


class GruDims(ctypes.Structure):
    """struct stp3_gru_dims (include/stp3_hip.h)."""
    _fields_ = [('rows', ctypes.c_int64), ('Cx', ctypes.c_int32), ('C', ctypes.c_int32), ('dtype', ctypes.c_int32),
                ('bias_init', ctypes.c_float)]


class BnDims(ctypes.Structure):
    """struct stp3_bn_dims (include/stp3_hip.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in ('N', 'rows', 'C', 'ldx', 'ldy', 'ldr', 'dtype', 'act', 'res_mode',
                                              'has_sbias', 'has_oscale', 'cpad')]


class ConvDims(ctypes.Structure):
    """struct stp3_conv_dims (include/stp3_hip.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in ('N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'KH', 'KW', 'stride', 'pad_h',
                                              'pad_w', 'dil_h', 'dil_w', 'ldx', 'ldy', 'out_dtype', 'has_bias')]


class SeDims(ctypes.Structure):
    """struct stp3_se_dims (include/stp3_hip.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in ('N', 'rows', 'C', 'ld', 'dtype')]


class WprepEntry(ctypes.Structure):
    """struct stp3_wprep_entry (include/stp3_hip.h)."""
    _fields_ = [('src', ctypes.c_void_p), ('fwd', ctypes.c_void_p), ('flip', ctypes.c_void_p),
                ('stride_co', ctypes.c_int64), ('stride_ci', ctypes.c_int64), ('stride_kh', ctypes.c_int64),
                ('stride_kw', ctypes.c_int64), ('first_block', ctypes.c_int64),
                ('cout', ctypes.c_int32), ('cin', ctypes.c_int32), ('kh', ctypes.c_int32), ('kw', ctypes.c_int32),
                ('dst_cout', ctypes.c_int32), ('dst_cin', ctypes.c_int32), ('co_off', ctypes.c_int32),
                ('ci_off', ctypes.c_int32), ('fwd_f32', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class SeMlpDims(ctypes.Structure):
    """struct stp3_se_mlp_dims (include/stp3_hip.h)."""
    _fields_ = [('N', ctypes.c_int32), ('C', ctypes.c_int32), ('S', ctypes.c_int32), ('inv_rows', ctypes.c_float)]


class PairDims(ctypes.Structure):
    """struct stp3_pair_dims (include/stp3_hip.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in ('frames', 'T', 'rows', 'C', 'ldx', 'dtype')]


class UpsampleDims(ctypes.Structure):
    """struct stp3_upsample_dims (include/stp3_hip.h)."""
    _fields_ = [(k, ctypes.c_int32) for k in ('N', 'H', 'W', 'C', 'scale', 'ldx', 'ldy', 'dtype')]


class CeDims(ctypes.Structure):
    """struct stp3_ce_dims (include/stp3_hip.h)."""
    _fields_ = [('rows', ctypes.c_int32), ('P', ctypes.c_int32), ('C', ctypes.c_int32), ('k', ctypes.c_int32),
                ('ignore_index', ctypes.c_int32), ('dtype', ctypes.c_int32), ('stride_row', ctypes.c_int64),
                ('stride_c', ctypes.c_int64), ('stride_p', ctypes.c_int64)]


class PlanDims(ctypes.Structure):
    """struct stp3_plan_dims (include/stp3_hip.h)."""
    _fields_ = ([(k, ctypes.c_int32) for k in ('B', 'N', 'T', 'H', 'W', 'K0', 'KL')] +
                [(k, ctypes.c_float) for k in ('dx0', 'dx1', 'bx0', 'bx1', 'safety', 'headway', 'lrdivider', 'comfort',
                                               'progress', 'volume', 'rule', 'w0', 'w1', 'headway_dist', 'lr_dist')])


class ImageDims(ctypes.Structure):
    """struct stp3_image_dims (include/stp3_hip.h)."""
    _fields_ = ([(k, ctypes.c_int32) for k in ('N', 'H', 'W', 'Wr', 'Hr', 'left', 'top', 'Wo', 'Ho', 'ksize_h', 'ksize_v',
                                               'out_dtype')] + [('mean', ctypes.c_float * 3), ('std', ctypes.c_float * 3)])


class Poly(ctypes.Structure):
    """struct stp3_poly (include/stp3_hip.h)."""
    _fields_ = [('map', ctypes.c_int32), ('nv', ctypes.c_int32), ('value', ctypes.c_float), ('reserved', ctypes.c_int32),
                ('xy', ctypes.c_int32 * 16)]


class OptimBucket(ctypes.Structure):
    """struct stp3_optim_bucket (include/stp3_hip.h)."""
    _fields_ = [('grad', ctypes.c_void_p), ('param', ctypes.c_void_p), ('exp_avg', ctypes.c_void_p),
                ('exp_avg_sq', ctypes.c_void_p), ('numel', ctypes.c_int64), ('first_block', ctypes.c_int64)]


DTYPE_F32 = 0
DTYPE_BF16 = 1

VOX_REFERENCE = 0
VOX_PIXELMAJOR = 1
BEV_CHANNELS_FIRST = 0
BEV_CHANNELS_LAST = 1
BEV_CHANNELS_LAST_BF16 = 2

ACT_NONE, ACT_RELU, ACT_SWISH = 0, 1, 2
ACT_GELU = 3                  # stp3_layernorm_* only
RES_NONE, RES_BEFORE_ACT, RES_AFTER_ACT = 0, 1, 2

_DIMS_P = ctypes.POINTER(LiftDims)
_BN_P = ctypes.POINTER(BnDims)
c_double = ctypes.c_double
_DW_P = ctypes.POINTER(DwConvDims)
_LN_P = ctypes.POINTER(LayerNormDims)
_GRU_P = ctypes.POINTER(GruDims)

# name -> (restype, argtypes); mirrors include/stp3_hip.h one to one
SIGNATURES = {
    'stp3_version': (ctypes.c_char_p, []),
    'stp3_voxel_index': (c_int, [_DIMS_P] + [c_void_p] * 9 + [c_int, c_void_p, c_void_p, c_void_p]),
    'stp3_lift_plan_bytes': (c_int, [_DIMS_P, ctypes.POINTER(c_size_t)]),
    'stp3_lift_plan_build': (c_int, [_DIMS_P] + [c_void_p] * 12 + [c_size_t, c_void_p]),
    'stp3_depth_softmax': (c_int, [_DIMS_P, c_void_p, c_void_p, c_void_p]),
    'stp3_lift_workspace_bytes': (c_int, [_DIMS_P, ctypes.POINTER(c_size_t)]),
    'stp3_lift_splat_fwd': (c_int, [_DIMS_P, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_size_t,
                                    c_void_p, c_void_p, c_void_p]),
    'stp3_lift_bwd_needs_prob': (c_int, [_DIMS_P, ctypes.POINTER(c_int)]),
    'stp3_lift_splat_bwd': (c_int, [_DIMS_P, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_float, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'stp3_gru_reset_cat_fwd': (c_int, [_GRU_P, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_gru_output_fwd': (c_int, [_GRU_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_gru_output_bwd': (c_int, [_GRU_P, c_void_p, ctypes.c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_gru_reset_cat_bwd': (c_int, [_GRU_P, c_void_p, ctypes.c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_layernorm_fwd': (c_int, [_LN_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_layernorm_bwd_workspace': (c_int, [_LN_P, ctypes.POINTER(c_size_t)]),
    'stp3_layernorm_bwd': (c_int, [_LN_P] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    'stp3_dwconv2d_fwd': (c_int, [_DW_P, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_dwconv2d_fwd_bias': (c_int, [_DW_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_dwconv2d_bwd_data': (c_int, [_DW_P, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_dwconv2d_bwd_weight_workspace': (c_int, [_DW_P, ctypes.POINTER(c_size_t)]),
    'stp3_dwconv2d_bwd_weight': (c_int, [_DW_P, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'stp3_dwconv2d_bwd_weight_oihw': (c_int, [_DW_P, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'stp3_bn_workspace_bytes': (c_int, [_BN_P, ctypes.POINTER(c_size_t)]),
    'stp3_bn_stats': (c_int, [_BN_P, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    'stp3_bn_apply_fwd': (c_int, [_BN_P] + [c_void_p] * 5 + [c_double, c_void_p, c_void_p, c_float, c_float]
                          + [c_void_p] * 6),
    'stp3_bn_bwd_reduce': (c_int, [_BN_P] + [c_void_p] * 10 + [c_size_t, c_void_p, c_void_p, c_void_p]),
    'stp3_bn_apply_bwd': (c_int, [_BN_P] + [c_void_p] * 10 + [c_double, c_void_p, c_void_p, c_void_p]),
    'stp3_bn_fwd_train': (c_int, [_BN_P] + [c_void_p] * 6 + [c_float, c_float] + [c_void_p] * 4 + [c_size_t, c_void_p,
                                                                                                 c_void_p]),
    'stp3_bn_bwd_train': (c_int, [_BN_P] + [c_void_p] * 10 + [c_size_t] + [c_void_p] * 4),
    'stp3_bn_dsbias': (c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_sum_n': (c_int, [c_int32, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    'stp3_linear_fwd': (c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    'stp3_linear_bwd': (c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                                c_void_p]),
    'stp3_sum_n_plane': (c_int, [c_int32, ctypes.c_int64, c_int32, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_void_p, c_void_p]),
    'stp3_conv2d_fwd_workspace': (c_int, [ctypes.POINTER(ConvDims), ctypes.POINTER(c_size_t)]),
    'stp3_conv2d_fwd': (c_int, [ctypes.POINTER(ConvDims)] + [c_void_p] * 6 + [c_size_t, c_void_p]),
    'stp3_conv2d_fwd_add': (c_int, [ctypes.POINTER(ConvDims), c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    'stp3_conv2d_fwd_stats': (c_int, [ctypes.POINTER(ConvDims), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'stp3_conv2d_fwd_bnact': (c_int, [ctypes.POINTER(ConvDims), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    'stp3_conv2d_bn_bwd_reduce': (c_int, [ctypes.POINTER(ConvDims), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    'stp3_conv2d_bn_bwd_apply': (c_int, [ctypes.POINTER(ConvDims), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                                         c_void_p, c_double, c_void_p, c_void_p]),
    'stp3_se_workspace_bytes': (c_int, [ctypes.POINTER(SeDims), ctypes.POINTER(c_size_t)]),
    'stp3_se_pool': (c_int, [ctypes.POINTER(SeDims), c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    'stp3_se_scale': (c_int, [ctypes.POINTER(SeDims), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_conv2d_bn_bwd_apply_dx': (c_int, [ctypes.POINTER(ConvDims), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p,
                                            ctypes.c_double, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    'stp3_conv2d_wgrad_workspace': (c_int, [ctypes.POINTER(ConvDims), ctypes.POINTER(c_size_t)]),
    'stp3_conv2d_wgrad': (c_int, [ctypes.POINTER(ConvDims), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'stp3_conv2d_wgrad_partials': (c_int, [ctypes.POINTER(ConvDims), c_void_p, c_void_p, c_void_p, c_size_t,
                                           ctypes.POINTER(c_int32), c_void_p]),
    'stp3_conv2d_wgrad_reduce_batch': (c_int, [c_int32, ctypes.POINTER(WgradJob), c_void_p]),
    'stp3_conv2d_prep_weights': (c_int, [c_void_p, c_int32, ctypes.c_int64, c_void_p]),
    'stp3_conv2d_scatter_weight_grads': (c_int, [c_void_p, c_int32, ctypes.c_int64, c_void_p]),
    'stp3_se_mlp_fwd': (c_int, [c_void_p] * 9),
    'stp3_se_mlp_bwd': (c_int, [c_void_p] * 15),
    'stp3_optim_workspace_bytes': (c_int, [ctypes.c_int64, ctypes.POINTER(c_size_t)]),
    'stp3_optim_clip_adam': (c_int, [c_void_p, c_int32, ctypes.c_int64] + [c_float] * 6 + [c_void_p, c_void_p, c_size_t,
                                                                                          c_void_p]),
    'stp3_causal_pair_fwd': (c_int, [ctypes.POINTER(PairDims), c_void_p, c_void_p, c_void_p]),
    'stp3_causal_pair_bwd': (c_int, [ctypes.POINTER(PairDims), c_void_p, c_void_p, c_void_p]),
    'stp3_upsample_bilinear_fwd': (c_int, [ctypes.POINTER(UpsampleDims), c_void_p, c_void_p, c_void_p]),
    'stp3_upsample_bilinear_bwd': (c_int, [ctypes.POINTER(UpsampleDims), c_void_p, c_void_p, c_void_p]),
    'stp3_dwconv2d_fwd_stats_workspace': (c_int, [_DW_P, ctypes.POINTER(c_size_t)]),
    'stp3_dwconv2d_fwd_stats': (c_int, [_DW_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'stp3_bn_finalize': (c_int, [c_void_p, c_int32, c_double, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    'stp3_mbconv_workspace_bytes': (c_int, [ctypes.POINTER(SeDims), ctypes.POINTER(c_size_t)]),
    'stp3_se_pool_act': (c_int, [ctypes.POINTER(SeDims), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p,
                                 c_void_p]),
    'stp3_mbconv_scale_act': (c_int, [ctypes.POINTER(SeDims), c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                      c_void_p, c_void_p]),
    'stp3_mbconv_bwd_reduce': (c_int, [ctypes.POINTER(SeDims), c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                       c_size_t, c_void_p, c_void_p]),
    'stp3_mbconv_bwd_coef': (c_int, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_mbconv_bwd_apply': (c_int, [ctypes.POINTER(SeDims), c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                      c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    'stp3_dwconv2d_fwd_stats_bn': (c_int, [_DW_P, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_float,
                                           c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'stp3_ce_topk_workspace_bytes': (c_int, [ctypes.POINTER(CeDims), ctypes.POINTER(c_size_t)]),
    'stp3_ce_topk_fwd': (c_int, [ctypes.POINTER(CeDims)] + [c_void_p] * 6 + [c_double, c_int32, c_void_p, c_void_p, c_size_t,
                                                                           c_void_p]),
    'stp3_ce_topk_bwd': (c_int, [ctypes.POINTER(CeDims)] + [c_void_p] * 7 + [c_double, c_void_p, c_void_p]),
    'stp3_reg_loss_workspace_bytes': (c_int, [ctypes.POINTER(c_size_t)]),
    'stp3_reg_loss_fwd': (c_int, [c_int32] * 4 + [c_float, c_int32] + [c_void_p] * 5 + [c_size_t, c_void_p]),
    'stp3_reg_loss_bwd': (c_int, [c_int32] * 4 + [c_float, c_int32] + [c_void_p] * 7),
    'stp3_warp_nearest': (c_int, [c_int32] * 4 + [c_void_p] * 5),
    'stp3_traj_cost_fwd': (c_int, [ctypes.POINTER(PlanDims)] + [c_void_p] * 14),
    'stp3_traj_cost_bwd': (c_int, [ctypes.POINTER(PlanDims)] + [c_void_p] * 5),
    'stp3_image_prep_rows_per_workgroup': (c_int, []),
    'stp3_image_prep_lds_bytes': (c_int, [ctypes.POINTER(ImageDims), c_int32, ctypes.POINTER(c_size_t)]),
    'stp3_image_prep': (c_int, [ctypes.POINTER(ImageDims)] + [c_void_p] * 5 + [c_int32, c_void_p, c_void_p]),
    'stp3_fill_polygons': (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'stp3_instance_labels_workspace_bytes': (c_int, [c_int32, c_int32, ctypes.POINTER(c_size_t)]),
    'stp3_instance_labels': (c_int, [c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    'stp3_voxels_sum_fwd': (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    'stp3_voxels_sum_bwd': (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
}

_lib = None
_timed = None


class Stp3HipError(RuntimeError):
    pass


def lib():
    """Load libstp3hip.so once; fail loudly if it is absent or incomplete."""
    global _lib, _timed
    if _lib is not None:
        if profiling.ENABLED:                 # bench.py's roofline steps: every call bracketed by stream events
            if _timed is None:
                _timed = profiling.TimedLib(_lib)
            return _timed
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Stp3HipError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C st-p3_amd/csrc`).  There is no fallback path.')
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise Stp3HipError(f'libstp3hip.so does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


_ERRORS = {-10001: 'STP3_EINVAL (bad dimension / null pointer)',
           -10002: 'STP3_EUNSUP (unsupported configuration)',
           -10003: 'STP3_ENOSPACE (workspace too small)'}


def check(rc, what):
    if rc != 0:
        raise Stp3HipError(f'{what} failed: {_ERRORS.get(rc, f"hipError {-rc}")}')
