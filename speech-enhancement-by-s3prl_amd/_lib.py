"""ctypes binding of libse_amd.so (the C ABI declared in include/se_amd.h).

There is NO CPU fallback: if the library is missing, or a launch is requested for a tensor that does not
live on a gfx950 device, this module raises.  Pointers handed to the library are `tensor.data_ptr()` of
contiguous torch tensors; the stream is torch's current HIP stream.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint16, c_void_p

import torch

from . import checkpoint as _checkpoint      # noqa: F401  (registers argparse.Namespace with torch's weights-only unpickler: checkpoint.py)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SE_AMD_LIB: developer switch for A/B and stamp builds (tools/): another build of the SAME library; never a fallback
LIB_PATH = os.environ.get('SE_AMD_LIB') or os.path.join(_HERE, 'libse_amd.so')

SE_ACT = {'Identity': 0, 'ReLU': 1, 'Sigmoid': 2, 'GELU': 3, 'Exp': 4}


class SEError(RuntimeError):
    pass


class Geometry(ctypes.Structure):
    _fields_ = [('sample_rate', c_int), ('win', c_int), ('hop', c_int), ('n_freq', c_int), ('n_mels', c_int)]


class EncoderConfig(ctypes.Structure):
    _fields_ = [('input_dim', c_int), ('hidden', c_int), ('layers', c_int), ('heads', c_int), ('intermediate', c_int),
                ('ln_eps', c_float), ('spec_out', c_int), ('fused_ln_min_rows', c_int)]


_FP = POINTER(c_float)
_FPP = POINTER(_FP)


class EncoderWeights(ctypes.Structure):
    _fields_ = ([(n, _FP) for n in ('in_w', 'in_b', 'in_ln_w', 'in_ln_b')] +
                [(n, _FPP) for n in ('q_w', 'q_b', 'k_w', 'k_b', 'v_w', 'v_b', 'ao_w', 'ao_b', 'aln_w', 'aln_b',
                                     'ff1_w', 'ff1_b', 'ff2_w', 'ff2_b', 'oln_w', 'oln_b')] +
                [(n, _FP) for n in ('sh_dense_w', 'sh_dense_b', 'sh_ln_w', 'sh_ln_b', 'sh_out_w', 'sh_out_b')])


class EncoderGrads(ctypes.Structure):
    _fields_ = ([(n, _FP) for n in ('in_w', 'in_b', 'in_ln_w', 'in_ln_b')] +
                [(n, _FPP) for n in ('q_w', 'q_b', 'k_w', 'k_b', 'v_w', 'v_b', 'ao_w', 'ao_b', 'aln_w', 'aln_b',
                                     'ff1_w', 'ff1_b', 'ff2_w', 'ff2_b', 'oln_w', 'oln_b')])


_P = c_void_p
LAYER_DONE_CB = ctypes.CFUNCTYPE(None, c_int, c_void_p)      # se_encoder_bwd_cb_bf16's host callback
# name -> (restype, argtypes); mirrors include/se_amd.h one to one
SIGNATURES = {
    'se_last_error': (c_char_p, []),
    'se_version': (c_char_p, []),
    'se_device_available': (c_int, []),
    'se_plan_create': (c_int, [POINTER(Geometry), POINTER(_P)]),
    'se_plan_destroy': (None, [_P]),
    'se_plan_tables': (c_int, [_P, _P, _P]),
    'se_num_frames': (c_int, [_P, c_int]),
    'se_stft_f32': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    'se_stft2_f32': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    'se_stft_tphase_f32': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P, _P]),
    'se_istft_tphase_f32': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, c_int, _P, _P]),
    'se_features_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'se_features_f32': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, c_size_t, _P]),
    'se_features2_f32': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, c_size_t, _P, c_int, _P, _P]),
    'se_features3_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'se_features3_colstats_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'se_features3_f32': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, c_size_t, _P, c_int, _P, _P, c_float, _P]),
    'se_istft_f32': (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, c_int, _P, _P, _P]),
    'se_masked_sumsq_f32': (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    'se_dbnorm_f32': (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, c_float, c_float, _P]),
    'se_length_masks_i64': (c_int, [_P, c_int, c_int, _P, _P]),
    'se_head_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'se_head_linear_f32': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    'se_head_colstats_f32': (c_int, [_P, c_int, c_int, c_int, c_float, _P, _P]),
    'se_head_w3_bytes': (c_size_t, [c_int, c_int]),
    'se_head_split_weights_f32': (c_int, [_P, c_int, c_int, _P, _P]),
    'se_head_linear_pre_f32': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'se_head_sisdr_scratch_doubles': (c_size_t, [c_int, c_int, c_int]),
    'se_head_linear_sisdr_f32': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_float, _P, _P, _P, _P, _P]),
    'se_sisdr_head_mean_f32': (c_int, [_P, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    'se_head_linear_bwd_f32': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    'se_head_dx_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'se_head_linear_dx_f32': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, c_size_t, _P]),
    'se_l1_masked_f32': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P]),
    'se_l1_scratch_doubles': (c_size_t, [c_int]),
    'se_l1_masked_loss_f32': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P]),
    'se_encoder_create': (c_int, [POINTER(EncoderConfig), POINTER(EncoderWeights), POINTER(_P)]),
    'se_encoder_destroy': (None, [_P]),
    'se_encoder_workspace_bytes': (c_size_t, [_P, c_int, c_int]),
    'se_encoder_fwd_bf16': (c_int, [_P, _P, _P, c_int, c_int, _P, _P, c_size_t, _P]),
    'se_encoder_fwd2_bf16': (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, c_size_t, _P]),
    'se_spechead_fwd_bf16': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, c_size_t, _P]),
    'se_spechead_fwd2_bf16': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, c_size_t, c_int, _P]),
    'se_valid_lengths_i32': (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    'se_gemm_bf16': (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
    'se_gemm_res_ln_bf16': (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P]),
    'se_gemm_res24_scratch_bytes': (c_size_t, []),
    'se_gemm_res24_lo_bytes': (c_size_t, [c_int]),
    'se_gemm_res24_ln_bf16': (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    'se_mhsa_fwd_bf16': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    'se_mhsa_fwd_prescaled_bf16': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    'se_mhsa_fwd_prescaled_variant_bf16': (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    'se_layernorm_f32': (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, _P]),
    'se_cast_f32_bf16': (c_int, [_P, c_size_t, _P, _P]),
    'se_gemm_f32': (c_int, [_P, ctypes.c_long, _P, ctypes.c_long, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, ctypes.c_long, c_int, c_int,
                            ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, _P]),
    'se_softmax_rows_f32': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'se_split3_bf16': (c_int, [_P, ctypes.c_long, c_int, c_int, c_int, c_int, _P, _P]),
    'se_gemm_x3out_bf16': (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    'se_layernorm_x3_f32': (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, c_int, _P]),
    'se_mhsa_fwd_x3_split_f32': (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    'se_gemm_res_ln_x3_bf16': (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P]),
    'se_mhsa_fwd_x3_f32': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    'se_transpose_bf16': (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P]),
    'se_transpose_f32_bf16': (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P]),
    'se_wgrad_bf16': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_size_t, _P]),
    'se_wgrad_tn_bf16': (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_size_t, _P]),
    'se_wgrad_tn_slabs_bf16': (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    'se_colsum_f32': (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P]),
    'se_colsum_groups': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    'se_layernorm_bwd_groups_f32': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P, _P, _P, _P, _P]),
    'se_layernorm_bwd_f32': (c_int, [_P, _P, _P, c_int, c_int, c_float, c_int, _P, _P, _P, _P, c_int, _P]),
    'se_gelu_layernorm_f32': (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, _P]),
    'se_spec_epilogue_f32': (c_int, [_P, c_size_t, c_int, c_int, c_float, _P, _P, _P]),
    'se_spec_epilogue_bwd_f32': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P]),
    'se_mhsa_fwd_lse_bf16': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_float, ctypes.c_uint64, ctypes.c_uint32, _P]),
    'se_mhsa_bwd_bf16': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_float, ctypes.c_uint64, ctypes.c_uint32, _P]),
    'se_gelu_bf16': (c_int, [_P, c_size_t, _P, _P]),
    'se_gelu_bwd_bf16': (c_int, [_P, _P, c_size_t, _P, _P]),
    'se_encoder_refresh_bf16': (c_int, [_P, POINTER(EncoderWeights), _P]),
    'se_encoder_saved_bytes': (c_size_t, [_P, c_int, c_int]),
    'se_encoder_train_workspace_bytes': (c_size_t, [_P, c_int, c_int]),
    'se_encoder_fwd_train_bf16': (c_int, [_P, _P, _P, c_int, c_int, _P, _P, c_size_t, _P, c_size_t, c_float, ctypes.c_uint64, _P]),
    'se_encoder_bwd_bf16': (c_int, [_P, _P, c_int, c_int, _P, _P, c_size_t, POINTER(EncoderGrads), _P, c_size_t, c_float, ctypes.c_uint64, _P]),
    'se_encoder_bwd_cb_bf16': (c_int, [_P, _P, c_int, c_int, _P, _P, c_size_t, POINTER(EncoderGrads), _P, c_size_t, c_float, ctypes.c_uint64, _P, _P, _P]),
    'se_multi_sumsq_f32': (c_int, [_P, _P, c_int, _P, _P]),
    'se_bertadam_step_f32': (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_double, c_double, c_double, c_double, c_double, c_double, _P]),
    'se_multi_copy_f32': (c_int, [_P, _P, _P, c_int, _P]),
    'se_mix_f32': (c_int, [_P, c_int, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P]),
    'se_sisdr_f32': (c_int, [_P, _P, c_int, _P, c_int, c_float, _P, _P, _P]),
    'se_lstm_fwd_bf16': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    'se_lstm_bwd_bf16': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    'se_colsum_bf16': (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    'se_sisdr_spec_loss_scratch_doubles': (c_size_t, [c_int, c_int, c_int]),
    'se_sisdr_spec_loss_f32': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P]),
    'se_sisdr_spec_f32': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P]),
    'se_wsd_energy_f32': (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    'se_wsd_f32': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, _P, _P, _P]),
    'se_prof_enable': (c_int, [c_int]),
    'se_prof_reset': (c_int, []),
    'se_prof_read': (c_int, [c_int, POINTER(c_double), POINTER(c_double), POINTER(ctypes.c_longlong)]),
}

_lib = None


def load():
    """Loads the shared library (once). Raises SEError if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SEError(f'{LIB_PATH} is missing: build it with `python speech-enhancement-by-s3prl_amd/build.py` '
                      '(or __graft_entry__.build()); the MI355X path has no CPU fallback')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name) and os.environ.get('SE_AMD_ALLOW_PARTIAL') == '1':
            continue                 # bring-up only; the test-suite checks every symbol is exported
        fn = getattr(lib, name)      # AttributeError here == header / library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().se_last_error().decode()


def check(status, what):
    if status != 0:
        raise SEError(f'{what} failed with status {status}: {last_error()}')


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    mat = getattr(t, 'materialize', None)
    if mat is not None:              # a LazyTensor (preprocessor.py) handed to a kernel: `.contiguous()` / `.float()` on it are no-ops that return the
        t = mat()                    # wrapper itself, whose data_ptr() is NULL -- the kernel gets the value (kept alive by the wrapper)
    if not t.is_cuda:
        raise SEError('libse_amd kernels take device tensors only (got a CPU tensor); there is no CPU fallback')
    if not t.is_contiguous():
        raise SEError('libse_amd kernels take contiguous tensors')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
