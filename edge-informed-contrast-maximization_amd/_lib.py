"""ctypes binding of libeincm_hip.so (C-ABI: include/eincm.h).  No CPU fallback: a missing library raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EINCM_LIB') or os.path.join(_HERE, 'libeincm_hip.so')      # EINCM_LIB: a developer's variant build (tools/build_variant.sh)

MAX_REFS = 16
N_STAGES = 10
STAGE_NAMES = ('clear', 'theta', 'splat', 'stats', 'imgrad', 'gather', 'tv', 'project', 'final', 'copy')

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_NONFINITE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
CONTRAST_GRAD_MAG, CONTRAST_VARIANCE = 0, 1
METHODS = {'linear': 0, 'bilinear': 0, 'triangle': 0, 'lanczos3': 1, 'lanczos5': 2, 'cubic': 3, 'bicubic': 3}
PF_FULL_AUX = 1
PF_NO_TV_GRAD = 2
SW_DEFER_CONSTANTS = 1
CF_TIMING = 1
CF_TIMING_DOMINANT = 2


class Params(C.Structure):
    _fields_ = [('alpha', C.c_double), ('beta', C.c_double), ('gamma', C.c_double), ('delta', C.c_double),
                ('cur_pyr_lvl', C.c_int32), ('method', C.c_int32), ('contrast_kind', C.c_int32), ('flags', C.c_uint32)]


class Aux(C.Structure):
    _fields_ = [('final_loss', C.c_double), ('mean_rel_corr', C.c_double), ('mean_rel_contrast', C.c_double),
                ('mean_rel_iwe_divergence', C.c_double), ('theta_total_variation', C.c_double)]


_A = C.c_double * MAX_REFS


class ObjectivesOut(C.Structure):
    _fields_ = [('n_refs', C.c_int32), ('_pad', C.c_int32),
                ('correlations', _A), ('zero_correlations', _A), ('rel_correlations', _A),
                ('contrasts', _A), ('zero_contrast', C.c_double), ('rel_contrasts', _A),
                ('theta_total_variation', C.c_double), ('theta_divergence', C.c_double),
                ('iwe_divergences', _A), ('zero_iwe_divergence', C.c_double), ('rel_iwe_divergences', _A),
                ('flow_warp_losses', _A), ('multi_ref_weights', _A), ('variances', _A), ('zero_variance', C.c_double)]


class TiledOut(C.Structure):
    _fields_ = [('n_refs', C.c_int32), ('n_tiles', C.c_int32),
                ('adaptive_mean_gradient_magnitude', _A), ('adaptive_variance', _A), ('adaptive_mean_squared_error', _A),
                ('sum_squared_error', _A), ('mean_hadamard_product', _A), ('sum_hadamard_product', _A), ('joint_contrast', _A)]


EDT_FORMULATIONS = {'exponential': 0, 'linear': 1, 'linear-bound': 2, 'logarithmic': 3}


class Timings(C.Structure):
    _fields_ = [('ms', C.c_float * N_STAGES), ('total_ms', C.c_float)]


# every symbol include/eincm.h declares: (name, restype, argtypes)
_P = C.c_void_p
_D = C.POINTER(C.c_double)
SIGNATURES = [
    ('eincm_abi_version', C.c_int, []),
    ('eincm_last_error', C.c_char_p, [_P]),
    ('eincm_create', _P, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint32]),
    ('eincm_destroy', None, [_P]),
    ('eincm_set_windows', C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int16),
                                    C.POINTER(C.c_int16), _D, _D, _D]),
    # the hot calls take raw addresses (c_void_p): ndarray.ctypes.data costs 1 us, data_as(POINTER(c_double)) 2.2 us per argument
    ('eincm_loss_grad', C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(Params), _P, _P, C.POINTER(Aux)]),
    ('eincm_loss_grad_async', C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(Params), C.c_int]),
    ('eincm_loss_grad_wait', C.c_int, [_P, _P, _P, C.POINTER(Aux)]),
    ('eincm_handover_loss_grad', C.c_int, [_P, _D, _D, _D, C.c_int, C.c_int, C.POINTER(Params), _D, _D]),
    ('eincm_objectives', C.c_int, [_P, _D, C.POINTER(ObjectivesOut)]),
    ('eincm_get_iwes', C.c_int, [_P, C.POINTER(C.c_float)]),
    ('eincm_get_zero_iwe', C.c_int, [_P, C.POINTER(C.c_float)]),
    ('eincm_get_image_grad', C.c_int, [_P, C.POINTER(C.c_float)]),
    ('eincm_get_scaled_theta', C.c_int, [_P, _D]),
    ('eincm_get_count_images', C.c_int, [_P, C.POINTER(C.c_uint32)]),
    ('eincm_loss_grad_device', C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_double, C.POINTER(C.c_double), C.c_void_p,
                                        C.POINTER(Aux)]),
    ('eincm_set_timing_period', C.c_int, [_P, C.c_int]),
    ('eincm_get_warped_events', C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ('eincm_multi_ref_weights', C.c_int, [C.c_int, _D]),
    ('eincm_resample_matrix', C.c_int, [C.c_int, C.c_int, C.c_int, _D]),
    ('eincm_get_timings', C.c_int, [_P, C.POINTER(Timings)]),
    ('eincm_set_timed_kernels', C.c_int, [_P, C.c_int, C.c_int]),
    ('eincm_loss_grad_masked_async', C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_void_p, C.c_int]),
    ('eincm_loss_grad_masked', C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('eincm_get_host_profile', C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    ('eincm_get_launch_policy', C.c_int, [_P, C.POINTER(C.c_double)]),
    ('eincm_get_timings_total', C.c_int, [_P, C.POINTER(Timings), C.POINTER(C.c_int64), C.c_int]),
    ('eincm_set_windows_ex', C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int16),
                                       C.POINTER(C.c_int16), _D, _D, _D, C.c_uint32]),
    ('eincm_set_windows_ptrs', C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_double), C.c_uint32]),
    ('eincm_forward_iwe', C.c_int, [_P, _D, C.c_int, C.c_int, C.POINTER(Params), C.c_int]),
    ('eincm_finish_loss_grad', C.c_int, [_P, _D, _D, C.POINTER(Aux)]),
    ('eincm_finish_constants', C.c_int, [_P]),
    ('eincm_set_device_results', C.c_int, [_P, C.c_int]),
    ('eincm_finish_launch', C.c_int, [_P]),
    ('eincm_grad_device_ptr', C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    ('eincm_finish_collect', C.c_int, [_P, _D, _D, C.POINTER(Aux)]),
    ('eincm_iwe_device_ptr', C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    ('eincm_mask_device_ptr', C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    ('eincm_inv_dist_transform', C.c_int, [_P, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_double, C.c_double, _D,
                                           C.POINTER(C.c_int32)]),
    ('eincm_gaussian_blur', C.c_int, [_P, _D, C.c_int, C.c_double, _D]),
    ('eincm_tiled_objectives', C.c_int, [_P, C.c_int, C.c_int, C.POINTER(TiledOut)]),
]

_lib = None


class EincmLibraryMissing(ImportError):
    pass


def load():
    """Load libeincm_hip.so (built in-tree by __graft_entry__.build()).  Raises if absent: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EincmLibraryMissing(
            f'{LIB_PATH} not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950). This engine has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)           # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
