"""ctypes binding of libmdgat_hip.so (C ABI in include/mdgat_hip.h).

The library is hand-written HIP for gfx950; there is no CPU or PyTorch fallback: if the shared
object is missing, importing the product path raises immediately."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MDGAT_HIP_LIB') or os.path.join(_HERE, 'libmdgat_hip.so')   # override: kernel experiments only

MAX_LAYERS = 64

EXTRACT_DUSTBIN = 0
EXTRACT_DUSTBIN_MUTUAL = 1
EXTRACT_THRESHOLD = 2
EXTRACT_THRESHOLD_MUTUAL = 3

PROF_CLASSES = ('encoder', 'layer', 'attention_full', 'attention_topk', 'scores', 'sinkhorn', 'extract', 'layer_first', 'layer_last',
                'f64_gemm', 'f64_attention_full', 'f64_attention_topk', 'f64_other')

ARITH_FP32 = 0
ARITH_FP64 = 1
F64_ENCODERS_ONLY = -1

OK = 0
ERR_BAD_ARG = -1
ERR_HIP = -2
ERR_UNSUPPORTED = -3
ERR_NO_WEIGHTS = -4


class MdgatConfig(C.Structure):
    _fields_ = [
        ('L', C.c_int32),
        ('sinkhorn_iters', C.c_int32),
        ('topk', C.c_int32 * MAX_LAYERS),
        ('extract_mode', C.c_int32),
        ('match_threshold', C.c_float),
        ('attention_mode', C.c_int32),
        ('arithmetic', C.c_int32),
        ('f64_layers', C.c_int32),
        ('f64_sinkhorn', C.c_int32),
    ]


class MdgatTaps(C.Structure):
    _fields_ = [('x_enc', C.c_void_p), ('x_layers', C.c_void_p), ('mdesc', C.c_void_p), ('scores', C.c_void_p),
                ('topk_sel', C.c_void_p)]


TAP_NAMES = ('x_enc', 'x_layers', 'mdesc', 'scores', 'topk_sel')


# name -> (restype, argtypes); every symbol include/mdgat_hip.h declares
SIGNATURES = {
    'mdgat_create': (C.c_int, [C.POINTER(MdgatConfig), C.c_int, C.POINTER(C.c_void_p)]),
    'mdgat_load_weights': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    'mdgat_load_weights_f64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    'mdgat_weights_f64_device_ptr': (C.c_void_p, [C.c_void_p]),
    'mdgat_blob_floats': (C.c_size_t, [C.c_int]),
    'mdgat_weights_device_ptr': (C.c_void_p, [C.c_void_p]),
    'mdgat_destroy': (None, [C.c_void_p]),
    'mdgat_last_error': (C.c_char_p, []),
    'mdgat_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    'mdgat_forward': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_void_p] * 4 +
                      [C.c_void_p, C.POINTER(MdgatTaps), C.c_void_p, C.c_size_t, C.c_void_p]),
    'mdgat_forward_f64': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_void_p] * 4 +
                          [C.c_void_p, C.POINTER(MdgatTaps), C.c_void_p, C.c_size_t, C.c_void_p]),
    'mdgat_forward_frames': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 +
                             [C.c_void_p, C.POINTER(MdgatTaps), C.c_void_p, C.c_size_t, C.c_void_p]),
    'mdgat_async_status': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    'mdgat_last_token': (C.c_uint, [C.c_void_p]),
    'mdgat_matched_any': (C.c_int, [C.c_void_p, C.c_uint, C.POINTER(C.c_uint)]),
    'mdgat_profile': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    'mdgat_set_lanes': (C.c_int, [C.c_void_p, C.c_int]),
    'mdgat_set_layer_split_tiles': (C.c_int, [C.c_int]),
    'mdgat_set_f64_layer_fusion': (C.c_int, [C.c_int]),
    'mdgat_set_f64_attention_form': (C.c_int, [C.c_int]),
    'mdgat_set_f64_sinkhorn_form': (C.c_int, [C.c_int]),
    'mdgat_sinkhorn_f64': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mdgat_sinkhorn_f64_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'mdgat_sinkhorn_f64_extract': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mdgat_sinkhorn': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
    'mdgat_sinkhorn_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'mdgat_extract': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 4 + [C.c_void_p]),
    'mdgat_attention': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    'mdgat_attention_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'mdgat_mfma_probe': (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_void_p]),
    'mdgat_mfma_f64_probe': (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_void_p]),
    'mdgat_pointwise_f64': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'mdgat_attention_f64': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mdgat_attention_qk_probe': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mdgat_attention_qk_probe_sets': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'mdgat_attention_sel': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    'mdgat_topk_sel_words': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'mdgat_pointwise': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'mdgat_pose': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                             C.c_void_p, C.c_void_p]),
    'mdgat_gt_matches': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mdgat_knn': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_size_t, C.c_void_p]),
    'mdgat_knn_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built - there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C mdgat_matcher_amd/csrc`.  mdgat_matcher_amd has no CPU / PyTorch fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().mdgat_last_error().decode('utf-8', 'replace')


def check(rc: int, what: str = 'libmdgat_hip'):
    """Raise like the reference would: RuntimeError carrying the library's message."""
    if rc != OK:
        raise RuntimeError(f'{what} failed (status {rc}): {last_error()}')
