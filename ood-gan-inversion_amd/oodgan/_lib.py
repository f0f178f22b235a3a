"""ctypes binding of liboodgan_hip.so (the C ABI declared in include/oodgan.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (the reference's pybind ops raise RuntimeError from TORCH_CHECK the same
way, src/ops/op/fused_bias_act.cpp:7,13-14)."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# OODGAN_LIB: load another build of the same ABI (the diagnostic stamp build of tools/clock_probe.py); default = the product
LIB_PATH = os.environ.get('OODGAN_LIB') or os.path.join(_HERE, 'liboodgan_hip.so')

CONV_S1, CONV_T2, CONV_S2 = 0, 1, 2
ACT_NONE, ACT_LRELU, ACT_PRELU = 0, 1, 2

P = c_void_p  # device pointers travel as integers


ABI_VERSION = 109


class ConvArgs(Structure):
    _fields_ = [
        ('x', P), ('wpk', P), ('in_scale', P), ('in_shift', P), ('out_scale', P), ('bias', P), ('noise', P),
        ('noise_w', P), ('slope', P), ('dotx', P), ('dot_part', P), ('y', P),
        ('B', c_int), ('K', c_int), ('M', c_int), ('Hin', c_int), ('Win', c_int),
        ('in_pitch', c_int), ('out_pitch', c_int), ('in_scale_stride', c_int), ('out_scale_stride', c_int),
        ('noise_batch', c_int), ('mode', c_int), ('act', c_int), ('dot_nparts', c_int), ('in_mul2', P),
        ('x_sform', c_int), ('ys', P), ('ys_scale', P), ('ys_scale_stride', c_int),
        ('rgb_w', P), ('rgb_s', P), ('rgb_y', P), ('rgb_s_stride', c_int), ('rgb_scale', c_float), ('fuse', P), ('dot_actgrad', c_int), ('groups', c_int), ('y_fform', c_int), ('x_fform', c_int), ('dotx_fform', c_int), ('dotx_sform', c_int), ('dotx_scale', P), ('dotx_scale_stride', c_int), ('workspace', P), ('workspace_bytes', c_long), ('ys_vmax', P), ('x_hi_only', c_int),
    ]


class ActBwdFuse(Structure):
    """oodgan_actbwd_fuse (include/oodgan.h): fused activation backward of the stride-2 input-gradient conv."""
    _fields_ = [('g_rgb', P), ('w_rgb', P), ('s_rgb', P), ('noise', P), ('noise_w', P), ('bias', P), ('dscale', P), ('mul2', P), ('ys', P),
                ('part_r', P), ('part_t', P), ('part_max', P), ('s_rgb_stride', c_int), ('noise_batch', c_int), ('dscale_stride', c_int),
                ('rgb_scale', c_float), ('nmax', c_long), ('ys_hi_only', c_int)]


class ReduceJob(Structure):
    _fields_ = [('part', P), ('out', P), ('B', c_int), ('C', c_int), ('nparts', c_int), ('out_stride', c_int), ('accumulate', c_int),
                ('part2', P), ('scale2', P), ('nparts2', c_int), ('scale2_stride', c_int)]


class DemodBwdJob(Structure):
    _fields_ = [('s', P), ('wsq', P), ('d', P), ('r', P), ('gs', P), ('s_stride', c_int), ('d_stride', c_int), ('gs_stride', c_int),
                ('B', c_int), ('Ci', c_int), ('Co', c_int), ('scale', c_float)]


class DemodFwdJob(Structure):
    _fields_ = [('s', P), ('wsq', P), ('d', P), ('s_stride', c_int), ('d_stride', c_int), ('B', c_int), ('Ci', c_int), ('Co', c_int),
                ('scale', c_float)]


class ScaleCheckJob(Structure):
    _fields_ = [('part', P), ('n', c_long), ('state', P)]


_SIGS = {
    'oodgan_version': (c_int, []),
    'oodgan_last_error': (c_char_p, []),
    'oodgan_device_count': (c_int, []),
    'oodgan_set_tunable': (c_int, [c_char_p, c_long]),
    'oodgan_get_tunable': (c_long, [c_char_p]),
    'oodgan_dispatch_count': (c_long, [c_char_p]),
    'oodgan_dispatch_reset': (c_int, []),
    'oodgan_zero': (c_int, [P, c_long, P]),
    'oodgan_bias_act_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_long, c_int, c_float, c_float, P]),
    'oodgan_bias_act_bwd': (c_int, [P, P, P, P, c_int, c_int, c_long, c_float, c_float, P]),
    'oodgan_upfirdn2d': (c_int, [P, P, P] + [c_int] * 15 + [P]),
    'oodgan_blur_bias_act': (c_int, [P, P, P] + [c_int] * 9 + [P, P, c_int, P, c_int, P]),
    'oodgan_style_affine_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    'oodgan_style_affine_bwd': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'oodgan_equal_linear': (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, c_float, c_int, P]),
    'oodgan_equal_linear_grouped': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, P]),
    'oodgan_pixel_norm': (c_int, [P, P, c_int, c_int, P]),
    'oodgan_weight_sqsum': (c_int, [P, P, c_int, c_int, c_int, P]),
    'oodgan_demod_fwd': (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'oodgan_demod_bwd': (c_int, [P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'oodgan_pack_conv3x3': (c_int, [P, P, c_int, c_int, c_float, c_int, c_int, P]),
    'oodgan_conv3x3': (c_int, [POINTER(ConvArgs), P]),
    'oodgan_conv3x3_nparts': (c_int, [c_int, c_int, c_int]),
    'oodgan_pack_conv3x3_f16s_bytes': (c_long, [c_int, c_int, c_int]),
    'oodgan_pack_conv3x3_f16s': (c_int, [P, P, P, c_int, c_int, c_float, c_int, c_int, P]),
    'oodgan_conv3x3_f16s': (c_int, [POINTER(ConvArgs), P, P]),
    'oodgan_conv3x3_f16s_nparts': (c_int, [c_int, c_int, c_int]),
    'oodgan_conv3x3_f16s_nparts2': (c_int, [c_int, c_int, c_int, c_int]),
    'oodgan_conv3x3_s2_fuse_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'oodgan_conv3x3_s1_ys_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'oodgan_conv3x3_s2_grouped_supported': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'oodgan_conv3x3_s1_actgrad_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'oodgan_sform_bytes': (c_long, [c_int, c_int, c_int, c_int]),
    'oodgan_sform_phases_bytes': (c_long, [c_int, c_int, c_int, c_int]),
    'oodgan_blurT_to_sform_phases': (c_int, [P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_from_sform': (c_int, [P, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_to_sform_phases': (c_int, [P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_to_sform_phases_padtl': (c_int, [P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_to_sform': (c_int, [P, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    'oodgan_resize_bicubic_ac': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_avgpool': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_blur_act_sform': (c_int, [P, P, P, P, P, c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    'oodgan_blur_act_sform_sep': (c_int, [P, P, P, P, P, c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    'oodgan_blur_act_fform': (c_int, [P, P, P, P, c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    'oodgan_conv3x3_xf_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'oodgan_upconv_vblur_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'oodgan_upconv_vblur_fform': (c_int, [P, P, c_long, P, P, P, c_int, P, P, c_int, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_conv3x3_xf_nparts': (c_int, [c_int, c_int, c_int]),
    'oodgan_conv3x3_tiny_workspace': (c_long, [c_int, c_int, c_int, c_int, c_int, c_int]),

    'oodgan_absmax_scaled': (c_int, [P, P, c_int, P, c_int, c_int, c_long, P]),
    'oodgan_fwd_range_update': (c_int, [P, P, P, c_int, P]),
    'oodgan_fwd_range_plan': (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_act_bwd_sform_nparts': (c_int, [c_int, c_int]),
    'oodgan_act_bwd_sform': (c_int, [P, P, P, c_int, P, P, P, P, P, c_int, c_float, P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_act_bwd_sform_f': (c_int, [P, P, c_int, P, P, P, P, P, c_int, c_float, P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_from_fform': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_act_bwd_blurT_nparts': (c_int, [c_int, c_int]),
    'oodgan_act_bwd_blurT_pre_supported': (c_int, [c_int, c_int]),
    'oodgan_act_bwd_blurT_sform_phases': (c_int, [P, P, P, c_int, P, P, P, P, P, c_int, c_float, P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_act_bwd_blurT_hi_supported': (c_int, [c_int, c_int]),
    'oodgan_conv3x3_s1_xh_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'oodgan_act_bwd_blurT_sform_phases_hi': (c_int, [P, P, P, c_int, P, P, P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_absmax_scale_check': (c_int, [P, c_long, P, P, P]),
    'oodgan_reduce_batch': (c_int, [P, c_int, P]),
    'oodgan_demod_bwd_batch': (c_int, [P, c_int, P]),
    'oodgan_demod_fwd_batch': (c_int, [P, c_int, P]),
    'oodgan_absmax_scale_check_batch': (c_int, [P, c_int, P, P]),
    'oodgan_hform_bytes': (c_long, [c_int, c_int, c_int, c_int]),
    'oodgan_to_hform': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_from_hform': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_modconv_f16_wbytes': (c_long, [c_int, c_int, c_int]),
    'oodgan_modconv_f16_pack': (c_int, [P, P, c_int, c_float, c_int, c_int, P, c_int, c_int, c_int, P]),
    'oodgan_modconv_f16_pack_affine': (c_int, [P, P, c_int, P, P, c_int, c_float, c_int, c_int, P, c_int, c_int, c_int, P]),
    'oodgan_modconv_f16': (c_int, [P, P, P, c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_reduce_parts': (c_int, [P, P, c_long, c_int, c_int, P]),
    'oodgan_reduce_parts_cols': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_torgb_fwd': (c_int, [P, P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    'oodgan_rgb_finish': (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    'oodgan_rgb_finish_parts': (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, P]),
    'oodgan_torgb_fwd_sform': (c_int, [P, P, P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P, P]),
    'oodgan_act_bwd_fused': (c_int, [P, P, P, c_int, P, P, P, P, P, c_int, c_float, P, P, P, c_int, c_int, c_long, P]),
    'oodgan_act_bwd_nparts': (c_int, [c_long]),
    'oodgan_feature_modulation': (c_int, [P, P, P, P, c_long, c_int, P]),
    'oodgan_act_bwd_fused_max': (c_int, [P, P, P, c_int, P, P, P, P, P, c_int, c_float, P, P, P, P, P, c_int, c_int, c_int, c_long, P]),
    'oodgan_absmax_scale': (c_int, [P, c_long, P, P]),
    'oodgan_absmax_scale_clear': (c_int, [P, c_long, P, P]),
    'oodgan_mse_fwd_bwd': (c_int, [P, P, P, P, P, c_int, c_long, c_float, P]),
    'oodgan_mse_fwd_bwd_row': (c_int, [P, P, P, P, P, P, c_int, c_int, c_long, c_float, P]),
    'oodgan_conv2d_s1': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_add_mask': (c_int, [P, P, P, c_long, P]),
    'oodgan_maxpool3s2_fwd': (c_int, [P, P, P, c_long, c_int, c_int, P]),
    'oodgan_maxpool3s2_bwd': (c_int, [P, P, P, P, P, c_long, c_int, c_int, P]),
    'oodgan_lpips_prep': (c_int, [P, P, c_int, c_int, c_int, c_float, c_float, POINTER(c_float), POINTER(c_float), P]),
    'oodgan_lpips_img_grad': (c_int, [P, P, c_int, c_int, c_int, c_float, c_float, POINTER(c_float), P]),
    'oodgan_lpips_head': (c_int, [P, P, P, P, P, c_int, c_int, c_long, c_float, c_int, P]),
    'oodgan_lpips_head_nparts': (c_int, [c_long]),
    'oodgan_lpips_finish': (c_int, [POINTER(P), POINTER(c_int), POINTER(c_long), c_int, P, P, c_int, c_int, P]),
    'oodgan_plan_create': (P, []),
    'oodgan_plan_destroy': (c_int, [P]),
    'oodgan_plan_record_begin': (c_int, [P]),
    'oodgan_plan_record_end': (c_long, [P]),
    'oodgan_plan_size': (c_long, [P]),
    'oodgan_plan_run': (c_int, [P, c_int]),
    'oodgan_plan_set_null_launch': (c_int, [c_int]),
    'oodgan_mse_nparts': (c_int, [c_long]),
    'oodgan_adam_step': (c_int, [P, P, P, P, c_long, c_float, c_float, c_float, c_float, c_int, P]),
    'oodgan_adam_step_dev': (c_int, [P, P, P, P, c_long, c_float, c_float, c_float, c_float, P, P]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises RuntimeError if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950); '
                'there is no CPU or PyTorch fallback for the hot path')
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name, None)
            if fn is None:
                continue  # optional groups are bound lazily by bind_extra()
            fn.restype = res
            fn.argtypes = args
        if h.oodgan_version() != ABI_VERSION:        # the argument structs of this mirror (ConvArgs: ys_vmax since 106) must match the library's
            raise RuntimeError(f'{LIB_PATH} has ABI version {h.oodgan_version()}, this host layer needs exactly {ABI_VERSION} (the argument structs must match): rebuild the library')
        _lib = h
    return _lib


def bind_extra(sigs):
    h = lib()
    for name, (res, args) in sigs.items():
        fn = getattr(h, name)
        fn.restype = res
        fn.argtypes = args
    _SIGS.update(sigs)


def check(rc, what=''):
    if rc != 0:
        msg = lib().oodgan_last_error()
        raise RuntimeError(f'liboodgan_hip {what} failed (rc={rc}): {msg.decode() if msg else ""}')


def set_tunable(name, value):
    """Dispatch tunable of the library (include/oodgan.h, oodgan_set_tunable); returns the previous value."""
    h = lib()
    old = h.oodgan_get_tunable(name.encode())
    check(h.oodgan_set_tunable(name.encode(), int(value)), 'set_tunable')
    return old


def dispatch_count(name):
    """Calls of oodgan_conv3x3_f16s routed to kernel family ``name`` since load / dispatch_reset() (include/oodgan.h)."""
    n = lib().oodgan_dispatch_count(name.encode())
    if n < 0:
        raise RuntimeError(f'unknown dispatch counter {name!r}')
    return n


def dispatch_reset():
    check(lib().oodgan_dispatch_reset(), 'dispatch_reset')


def exported_symbols():
    return sorted(_SIGS)
