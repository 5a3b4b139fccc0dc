"""ctypes binding of libtdr_hip.so (the C ABI declared in include/tdr.h).

The product path has NO fallback: if the shared library is missing or a call
fails, this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C textualdegremoval_amd/csrc`.
"""
import ctypes as C
import os

# torch bundles its own libamdhip64; it must be the HIP runtime already resident
# when libtdr_hip.so is dlopen'ed so both share one runtime (one device context,
# one set of streams).  Loading in the other order gives two runtimes.
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TDR_LIB_PATH', os.path.join(_HERE, 'libtdr_hip.so'))   # override: profiling probe builds

ABI_VERSION = 107      # csrc/tdr_error.cpp: bumped with every incompatible change of include/tdr.h
c_fp = C.c_void_p      # device pointers travel as integers
i32, i64, f32 = C.c_int, C.c_int64, C.c_float


class TdrConvDesc(C.Structure):
    _fields_ = [
        ('N', i32), ('Cin', i32), ('H', i32), ('W', i32),
        ('Cout', i32), ('OH', i32), ('OW', i32),
        ('KH', i32), ('stride', i32), ('dil', i32), ('pad', i32),
        ('inp', c_fp), ('in_ns', i64),
        ('gate', i32),
        ('kscale', c_fp), ('kscale_ns', i64),
        ('wp', c_fp), ('wp_ns', i64), ('Mpad', i32), ('wp_fmt', i32),
        ('out', c_fp), ('out_ns', i64),
        ('epi', i32),
        ('bias', c_fp), ('bias_ns', i64),
        ('scale', c_fp), ('scale_ns', i64),
        ('bias2', c_fp), ('bias2_ns', i64), ('bias2_mul', f32),
        ('res', c_fp), ('res_ns', i64),
        ('mask', c_fp), ('mask_ns', i64),
        ('aux', c_fp), ('aux_ns', i64),
        ('relu', i32),
    ]


class TdrConvP16Desc(C.Structure):
    _fields_ = [
        ('N', i32), ('Cin', i32), ('H', i32), ('W', i32), ('Cout', i32),
        ('inp', c_fp),
        ('wp', c_fp), ('Mpad', i32), ('wp_fmt', i32),
        ('bias', c_fp),
        ('res32', c_fp), ('res32_ns', i64),
        ('res16', c_fp),
        ('mask32', c_fp), ('mask32_ns', i64),
        ('mask16', c_fp),
        ('relu', i32),
        ('out32', c_fp), ('out32_ns', i64),
        ('out16', c_fp),
    ]


class TdrWgradP16Desc(C.Structure):
    _fields_ = [('N', i32), ('Cin', i32), ('H', i32), ('W', i32), ('Cout', i32),
                ('in16', c_fp), ('dout16', c_fp), ('g', c_fp), ('db', c_fp), ('ws', c_fp), ('ws_floats', i64), ('fmt', i32)]


class TdrPackJob(C.Structure):
    _fields_ = [
        ('w', c_fp), ('wp', c_fp),
        ('Cout', i32), ('Cin', i32), ('KH', i32), ('mode', i32), ('fmt', i32),
        ('M', i32), ('Kch', i32), ('KHe', i32), ('CK', i32), ('Mx', i32),
        ('total', i64), ('first_block', i64),
    ]


class TdrWgradDesc(C.Structure):
    _fields_ = [
        ('N', i32), ('Cin', i32), ('H', i32), ('W', i32), ('Cout', i32), ('OH', i32), ('OW', i32),
        ('KH', i32), ('stride', i32), ('pad', i32),
        ('inp', c_fp), ('in_ns', i64), ('gate', i32),
        ('dout', c_fp), ('dout_ns', i64),
        ('g', c_fp),
        ('db', c_fp),
        ('per_image', i32),
        ('ws', c_fp), ('ws_floats', i64),
        ('math', i32),
    ]


class TdrNafTailDesc(C.Structure):
    _fields_ = [('N', i32), ('C', i32), ('HW', i32), ('w_fmt', i32), ('c_out', i32), ('eps', f32),
                ('g', c_fp), ('g_ns', i64), ('sca', c_fp), ('x', c_fp), ('x_ns', i64),
                ('w3', c_fp), ('w4', c_fp), ('w5', c_fp),
                ('b3', c_fp), ('beta', c_fp), ('lnw', c_fp), ('lnb', c_fp), ('b4', c_fp), ('b5', c_fp), ('gamma', c_fp),
                ('y', c_fp), ('y_ns', i64), ('mu', c_fp), ('rs', c_fp), ('yn', c_fp), ('yn_ns', i64),
                ('t4', c_fp), ('t4_ns', i64), ('out', c_fp), ('out_ns', i64)]


class TdrNafTailBwdDesc(C.Structure):
    _fields_ = [('N', i32), ('C', i32), ('HW', i32), ('w_fmt', i32), ('c_out', i32),
                ('dout', c_fp), ('dout_ns', i64), ('gamma', c_fp), ('t4', c_fp), ('t4_ns', i64), ('y', c_fp), ('y_ns', i64),
                ('mu', c_fp), ('rs', c_fp), ('lnw', c_fp), ('w5t', c_fp), ('w4t', c_fp),
                ('dt4', c_fp), ('dt4_ns', i64), ('dy', c_fp), ('dy_ns', i64), ('gw', c_fp), ('gb', c_fp), ('ws', c_fp),
                ('w3t', c_fp), ('beta', c_fp), ('sca', c_fp), ('dgp', c_fp), ('dgp_ns', i64)]


class TdrNafHeadBwdDesc(C.Structure):
    _fields_ = [('N', i32), ('C', i32), ('HW', i32), ('w_fmt', i32),
                ('dt1', c_fp), ('dt1_ns', i64), ('x', c_fp), ('x_ns', i64), ('mu', c_fp), ('rs', c_fp), ('lnw', c_fp),
                ('w1t', c_fp), ('res', c_fp), ('res_ns', i64), ('dx', c_fp), ('dx_ns', i64), ('gw', c_fp), ('gb', c_fp), ('ws', c_fp)]


class TdrNafHeadFwdDesc(C.Structure):
    _fields_ = [('N', i32), ('C', i32), ('HW', i32), ('w_fmt', i32),
                ('x', c_fp), ('x_ns', i64), ('lnw', c_fp), ('lnb', c_fp), ('eps', f32), ('w1', c_fp), ('b1', c_fp),
                ('mu', c_fp), ('rs', c_fp), ('xn', c_fp), ('xn_ns', i64), ('t1', c_fp), ('t1_ns', i64)]


class TdrSfDynVecDesc(C.Structure):
    _fields_ = [('N', i32), ('c', i32), ('GK', i32), ('KK', i32), ('d', i32), ('eps', f32), ('momentum', f32)] + \
               [(k, c_fp) for k in ('ap', 'wconv', 'bn_w', 'bn_b', 'fc_w', 'fc_b', 'f0_w', 'f0_b', 'f1_w', 'f1_b', 'run_mean', 'run_var', 'nbt',
                                    'taps', 'ah', 'al', 'xhat', 'rstd', 'z', 'att')] + [('use_running', i32)]


class TdrSfDynVecBwdDesc(C.Structure):
    _fields_ = [('N', i32), ('c', i32), ('GK', i32), ('KK', i32), ('d', i32)] + \
               [(k, c_fp) for k in ('ap', 'wconv', 'bn_w', 'fc_w', 'f0_w', 'f1_w', 'taps', 'xhat', 'rstd', 'z', 'att', 'dtaps', 'dah', 'dal',
                                    'dap', 'g_wconv', 'g_bn_w', 'g_bn_b', 'g_fc_w', 'g_fc_b', 'g_f0_w', 'g_f0_b', 'g_f1_w', 'g_f1_b', 'ws')]


class TdrStepGuard(C.Structure):
    _fields_ = [('scale', f32), ('inv_scale', f32), ('max_scale', f32), ('good', i32), ('growth_interval', i32),
                ('step', i32), ('skipped', i32), ('finite', i32), ('bc1', f32), ('bc2_sqrt', f32)]


# name -> (restype, argtypes); every symbol include/tdr.h declares
SIGNATURES = {
    'tdr_version': (i32, []),
    'tdr_last_error': (C.c_char_p, []),
    'tdr_conv_forward': (i32, [C.POINTER(TdrConvDesc), c_fp]),
    'tdr_conv_ck': (i32, [i32]),
    'tdr_ssim_y64_ws_doubles': (i64, [i32, i32]),
    'tdr_ssim_y64': (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_local_avgpool_ws_floats': (i64, [i32, i32, i32, i32]),
    'tdr_local_avgpool': (i32, [c_fp, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_p16_bytes': (i64, [i32, i32, i32, i32]),
    'tdr_p16_bytes_fmt': (i64, [i32, i32, i32, i32, i32]),
    'tdr_p16_from_f32_fmt': (i32, [c_fp, i64, i32, i32, i32, i32, c_fp, i32, c_fp]),
    'tdr_p16_to_f32_fmt': (i32, [c_fp, i32, i32, i32, i32, c_fp, i64, i32, c_fp]),
    'tdr_p16_from_f32': (i32, [c_fp, i64, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_p16_to_f32': (i32, [c_fp, i32, i32, i32, i32, c_fp, i64, c_fp]),
    'tdr_conv3x3_p16': (i32, [C.POINTER(TdrConvP16Desc), c_fp]),
    'tdr_conv3x3_p16_force_cfg': (i32, [i32]),
    'tdr_wgrad3x3_p16_ws_floats': (i64, [C.POINTER(TdrWgradP16Desc)]),
    'tdr_wgrad3x3_p16': (i32, [C.POINTER(TdrWgradP16Desc), c_fp]),
    'tdr_packed_weight_floats': (i64, [i32, i32, i32]),
    'tdr_pack_weights': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_packed_weight_bytes_bx3': (i64, [i32, i32, i32]),
    'tdr_pack_weights_bx3': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_pack_job_init': (i32, [C.POINTER(TdrPackJob), c_fp, i32, i32, i32, i32, i32, c_fp]),
    'tdr_pack_weights_multi': (i32, [c_fp, i32, i64, c_fp]),
    'tdr_pack_patches': (i32, [c_fp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_wgrad_ws_floats': (i64, [C.POINTER(TdrWgradDesc)]),
    'tdr_conv_wgrad': (i32, [C.POINTER(TdrWgradDesc), c_fp]),
    'tdr_wgrad1x1_group_supported': (i32, [C.POINTER(TdrWgradDesc)]),
    'tdr_wgrad1x1_group_ws_floats': (i64, [C.POINTER(TdrWgradDesc), i32]),
    'tdr_wgrad1x1_group': (i32, [C.POINTER(TdrWgradDesc), i32, c_fp, c_fp]),
    'tdr_layernorm2d_fwd': (i32, [c_fp, i64, c_fp, c_fp, f32, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_ln_ws_floats': (i64, [i32, i32, i32]),
    'tdr_layernorm2d_bwd': (i32, [c_fp, c_fp, i64, c_fp, c_fp, c_fp, c_fp, i64, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp,
                                  c_fp, c_fp]),
    'tdr_dwsg_ws_floats': (i64, [i32, i32, i32, i32]),
    'tdr_dwsg_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_dwsg_bwd': (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_dwsg_bwd_biased': (i32, [c_fp, c_fp, f32, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_dwsg_bwd_parts_supported': (i32, [i32]),
    'tdr_dw_param_finish': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_dwgelu_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_dwgelu_bwd': (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_dwconv_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_dwconv_act_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_dwconv_act_bwd': (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_dwpair_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, i64, c_fp]),
    'tdr_dwpair_bwd': (i32, [c_fp, i64, c_fp, i64, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_dwconv_halves_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, c_fp, i64, c_fp]),
    'tdr_dwconv_halves_bwd': (i32, [c_fp, c_fp, i64, c_fp, c_fp, i64, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_dwconv_bwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_row_sumsq': (i32, [c_fp, i64, i32, i32, i32, c_fp, c_fp]),
    'tdr_mdta_pad': (i32, [i32]),
    'tdr_mdta_softmax': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_mdta_bwd': (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_axpby_dev': (i32, [c_fp, c_fp, c_fp, i64, c_fp, c_fp]),
    'tdr_dot': (i32, [c_fp, c_fp, i64, c_fp, c_fp, c_fp]),
    'tdr_pixel_shuffle2': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_leaky_relu_fwd': (i32, [c_fp, i64, f32, c_fp, c_fp]),
    'tdr_leaky_relu_bwd': (i32, [c_fp, c_fp, i64, f32, c_fp, c_fp]),
    'tdr_gather_col': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_mapper_combine': (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_mapper_combine_bwd': (i32, [c_fp, i32, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_cross_attention_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i32, i32, f32, c_fp, c_fp, c_fp]),
    'tdr_cross_attention_bwd': (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i32, i32, f32, c_fp, c_fp,
                                      c_fp, c_fp, c_fp]),
    'tdr_transpose_pad': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_conv_force_cfg': (i32, [i32, i32]),
    'tdr_packed_weight_bytes_hx2': (i64, [i32, i32, i32]),
    'tdr_pack_weights_hx2': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_pack_weights_bx3_batch': (i32, [c_fp, i64, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_sca_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, c_fp, c_fp]),
    'tdr_sca_bwd': (i32, [c_fp] * 8 + [i32, i32] + [c_fp] * 7 + [c_fp]),
    # ---- un-guided SFNet (csrc/tdr_sfnet.hip)
    'tdr_gelu_fwd': (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp, i64, c_fp]),
    'tdr_gelu_bwd': (i32, [c_fp, c_fp, c_fp, i64, c_fp]),
    'tdr_subsample2': (i32, [c_fp, i32, i32, i32, c_fp, c_fp]),
    'tdr_instnorm_fwd': (i32, [c_fp, c_fp, c_fp, f32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_instnorm_bwd': (i32, [c_fp] * 5 + [i32, i32, i32] + [c_fp] * 5),
    'tdr_region_affine_fwd': (i32, [c_fp, i64, c_fp, c_fp, f32, i32, i32, i32, i32, i32, c_fp, i64, c_fp, c_fp]),
    'tdr_region_affine_bwd': (i32, [c_fp, i64, c_fp, i64, c_fp, c_fp, f32, c_fp, i32, i32, i32, i32, i32, c_fp, i64, c_fp, c_fp, c_fp, c_fp]),
    'tdr_sf_dyn_vec_fwd': (i32, [C.POINTER(TdrSfDynVecDesc), c_fp]),
    'tdr_sf_dyn_vec_bwd': (i32, [C.POINTER(TdrSfDynVecBwdDesc), c_fp]),
    'tdr_sf_dynfilt_fwd': (i32, [c_fp, i64, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_sf_dynfilt_bwd_reduce': (i32, [c_fp, c_fp, i64, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_sf_dynfilt_bwd_dx': (i32, [c_fp] * 5 + [i32] * 6 + [c_fp, i64, c_fp]),
    'tdr_convt4_weight_to_3x3': (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_convt4_grad_from_3x3': (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_sf_region_split': (i32, [c_fp, i64, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_sf_local_affine': (i32, [c_fp, i64, c_fp, c_fp, c_fp, f32, i32, i32, i32, i32, i32, c_fp, i64, c_fp]),
    'tdr_sf_emerge': (i32, [c_fp, i64, c_fp, i32, i32, i32, c_fp, c_fp]),
    'tdr_sf_softmax_mix': (i32, [c_fp, i64, c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp]),
    'tdr_scaled_conv_param_grads': (i32, [c_fp] * 5 + [i32, i32] + [c_fp] * 3 + [c_fp]),
    'tdr_pair_sum_partials_multi': (i32, [c_fp, i32, i32, c_fp]),
    'tdr_dw_param_finish_nb': (i32, [i32, i32]),
    'tdr_dw_param_finish_multi': (i32, [c_fp, i32, i32, c_fp]),
    'tdr_scaled_conv_param_grads_multi': (i32, [c_fp, i32, i32, c_fp]),
    'tdr_chansum_ws_floats': (i64, [i32, i32, i32]),
    'tdr_channel_sum': (i32, [c_fp, i64, i32, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_copy_rows': (i32, [c_fp, i64, c_fp, i64, i32, i64, c_fp]),
    'tdr_add_rows': (i32, [c_fp, i64, c_fp, i64, i32, i64, c_fp]),
    'tdr_pixel_unshuffle2': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_pad_crop': (i32, [c_fp, i32, i32, i32, i32, c_fp, i32, i32, c_fp]),
    'tdr_relu_bwd': (i32, [c_fp, c_fp, i64, c_fp, c_fp]),
    'tdr_l1_loss': (i32, [c_fp, c_fp, i64, f32, f32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_lr_blocks_fwd': (i32, [c_fp] + [i32] * 8 + [c_fp, c_fp]),
    'tdr_lr_blocks_bwd': (i32, [c_fp] + [i32] * 8 + [c_fp, c_fp]),
    'tdr_patch_inv_norm': (i32, [c_fp] + [i32] * 10 + [c_fp, c_fp]),
    'tdr_coarse_argmax_box': (i32, [c_fp, c_fp, c_fp] + [i32] * 6 + [c_fp, c_fp, c_fp, c_fp]),
    'tdr_gather_ref_block': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp, i32, i32, i32, c_fp, c_fp]),
    'tdr_scatter_ref_block': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp, i32, i32, c_fp, c_fp]),
    'tdr_fine_argmax': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_fine_search_bwd': (i32, [c_fp] * 7 + [i32] * 4 + [c_fp, c_fp, c_fp]),
    'tdr_transfer_fwd': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, i64, c_fp]),
    'tdr_transfer_ws_floats': (i64, [i32, i32, i32, i32, i32, i32, i32, i32]),
    'tdr_transfer_bwd': (i32, [c_fp, i64, c_fp, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i32,
                               c_fp, c_fp, c_fp, c_fp]),
    'tdr_resize_bilinear': (i32, [c_fp, i32, i32, i32, c_fp, i32, i32, c_fp]),
    'tdr_unfold_windows': (i32, [c_fp, i32, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_patchify': (i32, [c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_vit_assemble': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp]),
    'tdr_attention_fwd_math': (i32, [c_fp, i32, i32, i32, i32, i32, f32, i32, i32, c_fp, c_fp]),
    'tdr_attention_fwd': (i32, [c_fp, i32, i32, i32, i32, i32, f32, c_fp, c_fp]),
    'tdr_token_match': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i64, c_fp, c_fp, c_fp, c_fp]),
    'tdr_transpose_f32': (i32, [c_fp, i32, i32, i32, c_fp, c_fp]),
    'tdr_tok_layernorm': (i32, [c_fp, c_fp, c_fp, i64, i32, f32, i32, c_fp, c_fp]),
    'tdr_tok16_gemm': (i32, [c_fp, c_fp, c_fp, i64, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_tok16x2_gemm': (i32, [c_fp, c_fp, c_fp, i64, i32, i32, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_cm_to_tok16x2': (i32, [c_fp, i32, i64, c_fp, c_fp]),
    'tdr_tok16x3_gemm': (i32, [c_fp, c_fp, c_fp, i64, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_cm_to_tok16x3': (i32, [c_fp, i32, i64, c_fp, c_fp]),
    'tdr_tok16_attention': (i32, [c_fp, i32, i32, i32, i32, i32, f32, c_fp, c_fp]),
    'tdr_optim_chunk': (i32, []),
    'tdr_multi_copy': (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, f32, c_fp]),
    'tdr_grad_sumsq': (i32, [c_fp, c_fp, c_fp, c_fp, i32, c_fp, c_fp, c_fp]),
    'tdr_adamw_step': (i32, [c_fp] * 8 + [i32, c_fp, C.POINTER(f32), i32, f32, i32, f32, f32, f32, f32, i32, c_fp]),
    'tdr_adamw_step_dev': (i32, [c_fp] * 8 + [i32, c_fp, c_fp, f32, i32, f32, f32, f32, f32, c_fp]),
    'tdr_naf_tail_supported': (i32, [i32, i32]),
    'tdr_naf_tail_fwd': (i32, [C.POINTER(TdrNafTailDesc), c_fp]),
    'tdr_naf_tail_bwd_ws_floats': (i64, [i32, i32, i32]),
    'tdr_naf_tail_bwd': (i32, [C.POINTER(TdrNafTailBwdDesc), c_fp]),
    'tdr_naf_head_bwd': (i32, [C.POINTER(TdrNafHeadBwdDesc), c_fp]),
    'tdr_naf_head_fwd': (i32, [C.POINTER(TdrNafHeadFwdDesc), c_fp]),
    'tdr_absmax_bits': (i32, [c_fp, i64, i32, i64, c_fp, c_fp]),
    'tdr_pair_sum_partials': (i32, [c_fp, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_pair_sum_mid_floats': (i64, [i32, i32]),
    'tdr_tksa_softmax': (i32, [c_fp, c_fp, c_fp, c_fp, C.POINTER(i32), i32, i32, i32, c_fp, c_fp, c_fp]),
    'tdr_tksa_bwd': (i32, [c_fp, c_fp, c_fp, c_fp, C.POINTER(i32), c_fp, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_dwk_fwd': (i32, [c_fp, i64, c_fp, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, c_fp, i64, c_fp]),
    'tdr_dwk_bwd_can_accumulate': (i32, [i32, i32, i32, i64, i64, i64, i64]),
    'tdr_dwk_bwd_acc': (i32, [c_fp, i64, c_fp, i64, c_fp, i64, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp, i64, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_dwk_bwd': (i32, [c_fp, i64, c_fp, i64, c_fp, i64, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp, i64, c_fp, c_fp, c_fp, c_fp]),
    'tdr_avgpool3': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_add_relu': (i32, [c_fp, c_fp, i64, c_fp, c_fp]),
    'tdr_linear_small_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_linear_small_bwd': (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_softmax_rows': (i32, [c_fp, c_fp, i32, i32, c_fp, c_fp]),
    'tdr_scale_copy': (i32, [c_fp, i64, c_fp, i32, i32, i64, c_fp, i64, c_fp]),
    'tdr_rows_dot': (i32, [c_fp, i64, c_fp, i64, i32, i64, c_fp, i32, c_fp, c_fp]),
    'tdr_dwk_bwd_ws_floats': (i64, [i32, i32, i32, i32, i32, i32]),
    'tdr_crop_augment': (i32, [c_fp, i64, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, c_fp]),
    'tdr_ssim3d_ws_floats': (i64, [i32, i32]),
    'tdr_ssim3d': (i32, [c_fp, c_fp, i32, i32, i32, f32, c_fp, c_fp, c_fp]),
    'tdr_plane_mean': (i32, [c_fp, i64, i32, i32, i32, c_fp, c_fp]),
    'tdr_plane_add': (i32, [c_fp, i64, c_fp, f32, i32, i32, i32, c_fp]),
    'tdr_prompt_weights_fwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp]),
    'tdr_prompt_weights_bwd': (i32, [c_fp, c_fp, c_fp, c_fp, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_prompt_mix_fwd': (i32, [c_fp, c_fp, i32, i32, i64, c_fp, c_fp]),
    'tdr_prompt_mix_bwd_ws_floats': (i64, [i32, i32]),
    'tdr_prompt_mix_bwd': (i32, [c_fp, c_fp, c_fp, i32, i32, i64, c_fp, c_fp, c_fp, c_fp]),
    'tdr_resize_bilinear_bwd': (i32, [c_fp, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_group_ln_act_fwd': (i32, [c_fp, c_fp, c_fp, f32, f32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_group_ln_ws_floats': (i64, [i32, i32, i32]),
    'tdr_group_ln_act_bwd': (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, f32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_mapper_combine_all': (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_mapper_combine_all_bwd': (i32, [c_fp, i32, i32, i32, i32, i32, c_fp, c_fp, c_fp, c_fp]),
    'tdr_gather_col_strided': (i32, [c_fp, i32, i32, i64, i64, i32, c_fp, c_fp]),
    'tdr_splitk_finish': (i32, [c_fp, i32, i32, i64, c_fp, c_fp, c_fp, i32, c_fp, c_fp]),
    'tdr_text_inject_fwd': (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_text_inject_bwd': (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_add_noise': (i32, [c_fp, c_fp, c_fp, c_fp, i32, i64, c_fp, c_fp]),
    'tdr_pool_time': (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_upsample_nearest_add': (i32, [c_fp, i32, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_pool_sum': (i32, [c_fp, i32, i32, i32, i32, c_fp, c_fp]),
    'tdr_comm_available': (i32, []),
    'tdr_comm_unique_id_bytes': (i32, []),
    'tdr_comm_unique_id': (i32, [c_fp]),
    'tdr_comm_init': (i32, [C.POINTER(c_fp), i32, i32, c_fp]),
    'tdr_comm_allreduce': (i32, [c_fp, c_fp, i64, i32, c_fp]),
    'tdr_comm_reduce': (i32, [c_fp, c_fp, i64, i32, c_fp]),
    'tdr_comm_broadcast': (i32, [c_fp, c_fp, i64, i32, c_fp]),
    'tdr_comm_rank': (i32, [c_fp]),
    'tdr_comm_world': (i32, [c_fp]),
    'tdr_comm_destroy': (i32, [c_fp]),
    'tdr_multi_ema': (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, f32, c_fp]),
    'tdr_pixel_loss': (i32, [i32, c_fp, c_fp, i32, i64, i64, f32, f32, f32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_l1_loss_guarded': (i32, [c_fp, c_fp, i64, f32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'tdr_multi_copy_guarded': (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, c_fp]),
    'tdr_grad_sumsq_guarded': (i32, [c_fp] * 5 + [i32, c_fp, c_fp, c_fp, f32, f32, c_fp]),
    'tdr_adamw_step_guarded': (i32, [c_fp] * 8 + [i32, c_fp, c_fp, c_fp, f32, i32, i32, f32, f32, f32, f32, c_fp]),
}

_lib = None


class TdrError(RuntimeError):
    pass


def load():
    """Load libtdr_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TdrError(f'{LIB_PATH} is missing: build it first (python -c "import __graft_entry__ as g; g.build()"). '
                       'There is no CPU fallback on the product path.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    have = lib.tdr_version()
    if have != ABI_VERSION:
        raise TdrError(f'{LIB_PATH} exports C-ABI version {have}, this package binds version {ABI_VERSION}: a stale build would be '
                       'called with shifted arguments -- rebuild it (python -c "import __graft_entry__ as g; g.build()")')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().tdr_last_error().decode('utf-8', 'replace')
        raise TdrError(f'{what} failed (code {rc}): {msg}')
