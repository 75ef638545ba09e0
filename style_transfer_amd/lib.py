"""ctypes binding of libstx.so (the C ABI declared in include/stx.h).

There is deliberately no fallback: if the shared object is missing or a call fails, an
exception is raised.  Build it with ``python -m style_transfer_amd.build``.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('STX_LIB') or os.path.join(_HERE, 'csrc', 'libstx.so')

HOST, DEVICE = 0, 1
LAYER_INPUT, LAYER_CONV, LAYER_RELU, LAYER_POOL = 0, 1, 2, 3
POOL_MAX, POOL_AVE = 0, 1
(Q_SHARED_ENGINES, Q_TARGET_UPLOADS, Q_TARGET_BYTES, Q_WEIGHT_BYTES, Q_TILE_EVALS,
 Q_PEERS_WITHOUT_ACCESS) = range(6)

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)


class StxError(RuntimeError):
    """A libstx call returned a negative status."""

    def __init__(self, func, code, message):
        super().__init__('%s failed with status %d: %s' % (func, code, message))
        self.code = code


class LayerDesc(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('type', ctypes.c_int), ('bottom', ctypes.c_char_p),
                ('top', ctypes.c_char_p), ('num_output', ctypes.c_int),
                ('kernel_size', ctypes.c_int), ('pad', ctypes.c_int), ('stride', ctypes.c_int),
                ('pool_mode', ctypes.c_int)]


class ContentTarget(ctypes.Structure):
    _fields_ = [('content_index', ctypes.c_int), ('layer', ctypes.c_char_p),
                ('channels', ctypes.c_int), ('height', ctypes.c_int), ('width', ctypes.c_int),
                ('features', ctypes.c_void_p), ('mem', ctypes.c_int)]


class StyleTarget(ctypes.Structure):
    _fields_ = [('style_index', ctypes.c_int), ('layer', ctypes.c_char_p),
                ('channels', ctypes.c_int), ('gram', ctypes.c_void_p), ('mem', ctypes.c_int)]


class Tap(ctypes.Structure):
    _fields_ = [('layer', ctypes.c_char_p), ('layer_weight', ctypes.c_double),
                ('is_content', ctypes.c_int), ('content_weight', ctypes.c_double),
                ('is_style', ctypes.c_int), ('style_weight', ctypes.c_double),
                ('is_dd', ctypes.c_int), ('dd_weight', ctypes.c_double)]


# name -> argtypes; every function returns int status except the three noted below.
_vp, _i, _sz, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_double
SIGNATURES = {
    'stx_reread_env': [],
    'stx_device_count': [c_int_p],
    'stx_device_name': [_i, ctypes.c_char_p, _sz],
    'stx_engine_create': [_i, ctypes.POINTER(LayerDesc), _i, ctypes.POINTER(_vp)],
    'stx_engine_create_shared': [_vp, ctypes.POINTER(_vp)],
    'stx_engine_wait': [_vp, _vp],
    'stx_engine_query': [_vp, _i, c_double_p],
    'stx_set_conv_weights': [_vp, ctypes.c_char_p, _vp, _vp, _i],
    'stx_sync': [_vp],
    'stx_fence': [_vp, ctypes.POINTER(ctypes.c_ulonglong)],
    'stx_fence_wait': [_vp, ctypes.c_ulonglong],
    'stx_engine_device': [_vp, c_int_p],
    'stx_engine_stream': [_vp, ctypes.POINTER(_vp)],
    'stx_malloc': [_vp, _sz, ctypes.POINTER(_vp)],
    'stx_free': [_vp, _vp],
    'stx_memset_async': [_vp, _vp, _i, _sz],
    'stx_memcpy_async': [_vp, _vp, _i, _vp, _i, _sz],
    'stx_set_contents_and_styles': [_vp, ctypes.POINTER(ContentTarget), _i,
                                    ctypes.POINTER(StyleTarget), _i],
    'stx_features_tile': [_vp, _vp, _i, _i, _i, ctypes.POINTER(ctypes.c_char_p), _i,
                          ctypes.POINTER(_vp), _i],
    'stx_sc_grad_tile': [_vp, _vp, _i, _i, _i, c_int_p, c_int_p, ctypes.POINTER(Tap), _i,
                         c_double_p, _vp, _i, _i],
    'stx_tile_buffers': [_vp, _i, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp)],
    'stx_gram_matrix': [_vp, _vp, _i, _i, _i, _vp, _i],
    'stx_image_cut_tile': [_vp, _vp, _i, _i, c_int_p, _i, _i, _i, _i, _vp],
    'stx_image_put_tile': [_vp, _vp, _i, _i, c_int_p, _i, _i, _i, _i, _vp],
    'stx_map_place': [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i],
    'stx_map_roll_add': [_vp, _vp, _vp, _i, _i, _i, c_int_p, _d, _d],
    'stx_image_resample': [_vp, _vp, _i, _i, _i, _vp, _i, _i, c_int_p, c_double_p, _i, c_int_p,
                           c_double_p, _i, _i],
    'stx_image_regularizers': [_vp, _vp, _vp, _i, _i, c_float_p, _d, _d, _d, _d, _vp, _d,
                               c_int_p, c_double_p],
    'stx_image_swt_haar': [_vp, _vp, _vp, _i, _i, c_int_p, _d, _d, c_double_p],
    'stx_adam_step': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _d, _d, _d, _d, _d, _d, _d],
    'stx_vec_dot': [_vp, _vp, _vp, _sz, c_double_p],
    'stx_vec_axpy': [_vp, _d, _vp, _vp, _sz],
    'stx_vec_scale': [_vp, _d, _vp, _sz],
    'stx_vec_mean_abs': [_vp, _vp, _sz, c_double_p],
    'stx_vec_dot_async': [_vp, _vp, _vp, _sz, _vp],
    'stx_vec_abs_sum_async': [_vp, _vp, _sz, _vp],
    'stx_vec_axpy_dev': [_vp, _d, _vp, _d, _d, _vp, _d, _vp, _vp, _sz],
    'stx_vec_scale_dev': [_vp, _d, _vp, _d, _vp, _sz],
    'stx_vec_axpy_dot_dev': [_vp, _d, _vp, _d, _d, _vp, _d, _d, _vp, _d, _vp, _vp, _vp, _vp, _sz, _vp],
    'stx_vec_lbfgs_pair': [_vp, _vp, _vp, _vp, _vp, _sz, _vp, c_double_p],
    'stx_vec_scale2_axpy': [_vp, _d, _d, _vp, _vp, _sz],
    'stx_image_step_stats': [_vp, _vp, _vp, _i, _i, c_double_p],
    'stx_image_step_stats_async': [_vp, _vp, _vp, _i, _i, c_double_p],
    'stx_image_to_u8': [_vp, _vp, _i, _i, c_float_p, _vp],
    'stx_op_conv_forward': [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp],
    'stx_op_conv_backward_data': [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp],
    'stx_op_pool_forward': [_vp, _vp, _i, _i, _i, _i, _vp],
    'stx_op_pool_backward': [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    'stx_op_style_terms': [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, c_double_p, c_double_p],
    'stx_op_content_terms': [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, c_int_p, _vp, c_double_p],
    'stx_last_tile_ms': [_vp, c_float_p],
    'stx_last_tile_flops': [_vp, c_double_p, c_double_p],
    'stx_clock_marks': [_vp, ctypes.c_int],
    'stx_clock_marks_read': [_vp, c_double_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)],
    'stx_profile_enable': [_vp, _i],
    'stx_profile_read': [_vp, ctypes.c_char_p, _sz, ctypes.POINTER(_sz)],
}
NON_STATUS = {'stx_version': (ctypes.c_char_p, []), 'stx_last_error': (ctypes.c_char_p, []),
              'stx_engine_destroy': (None, [_vp])}

_lib = None


def load():
    """Loads libstx.so once and declares every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s is missing: build the HIP extension first '
                          '(python -m style_transfer_amd.build)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, ctypes.c_int
    for name, (restype, argtypes) in NON_STATUS.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, restype
    _lib = lib
    return lib


def call(name, *args):
    """Calls a status-returning libstx function and raises StxError on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise StxError(name, rc, lib.stx_last_error().decode('utf-8', 'replace'))


def reread_env():
    """The library reads its STX_* switches from a snapshot of the environment (stx_reread_env): call this after
    changing one inside a running process.  A no-op while the library is not loaded (its first use takes the
    snapshot)."""
    if _lib is not None:
        call('stx_reread_env')


def device_count():
    n = ctypes.c_int(0)
    call('stx_device_count', ctypes.byref(n))
    return n.value


def device_name(device):
    buf = ctypes.create_string_buffer(256)
    call('stx_device_name', device, buf, 256)
    return buf.value.decode()


def version():
    return load().stx_version().decode()
