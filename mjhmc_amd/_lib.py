"""ctypes binding of libmjhmc_hip.so (include/mjhmc_hip.h).

There is deliberately NO CPU fallback: if the HIP library is missing or no MI355X is visible the
product path raises.  (The NumPy restatement under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
KERNEL_HEADERS = os.path.join(_HERE, 'csrc')     # elementwise.hpp / philox.hpp: what hipRTC compiles user expressions against
LIB_PATH = os.environ.get('MJHMC_HIP_LIB') or os.path.join(_HERE, 'lib', 'libmjhmc_hip.so')

# enums of include/mjhmc_hip.h
E_ISO_GAUSS, E_DIAG_GAUSS, E_ROUGH_WELL, E_MM_GAUSS, E_FUNNEL_NEAL, E_FUNNEL_REF, E_PRODUCT_OF_T, E_SPARSE_CODE, E_USER_EXPR, E_HOST = range(10)
F64, F32, BF16 = 0, 1, 2
MODE_MJHMC, MODE_CONTROL, MODE_CTHMC = 0, 1, 2
F_X, F_V, F_EX, F_EV, F_DEDX, F_HFLF, F_CACHE, F_DWELL, F_TRANS = range(9)
ERR_NO_DEVICE = -4
ERR_NONFINITE = -5
OP_SUM, OP_MIN, OP_MAX = 0, 1, 2
COMM_ID_BYTES = 128


class IterStats(ctypes.Structure):
    _fields_ = [('l', ctypes.c_int64), ('f', ctypes.c_int64), ('r', ctypes.c_int64), ('fl', ctypes.c_int64),
                ('n_cold', ctypes.c_int64), ('E_evals', ctypes.c_int64), ('dEdX_evals', ctypes.c_int64),
                ('nonfinite', ctypes.c_int32), ('L_used', ctypes.c_int32), ('eps_used', ctypes.c_double),
                ('n_flf_run', ctypes.c_int64)]


class EngineError(RuntimeError):
    pass


_P = ctypes.c_void_p
_dp = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes): the complete export list of include/mjhmc_hip.h
PROTOTYPES = {
    'mjhmc_last_error': (ctypes.c_char_p, []),
    'mjhmc_abi_version': (ctypes.c_int, []),
    'mjhmc_ctx_create': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_P)]),
    'mjhmc_ctx_destroy': (ctypes.c_int, [_P]),
    'mjhmc_ctx_info': (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int),
                                      ctypes.POINTER(ctypes.c_uint64)]),
    'mjhmc_energy_create': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_size_t, ctypes.POINTER(_P)]),
    'mjhmc_energy_create_expr': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, _P, ctypes.c_size_t,
                                                ctypes.c_char_p, ctypes.POINTER(_P)]),
    'mjhmc_expr_check': (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]),
    'mjhmc_energy_create_expr_coupled': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                                        ctypes.c_char_p, _P, ctypes.c_size_t, ctypes.c_char_p,
                                                        ctypes.POINTER(_P)]),
    'mjhmc_expr_check_coupled': (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                                ctypes.c_char_p, ctypes.c_char_p]),
    'mjhmc_energy_destroy': (ctypes.c_int, [_P]),
    'mjhmc_eval': (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_int64, _P, _P]),
    'mjhmc_sampler_create': (ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _P, _P,
                                            ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(_P)]),
    'mjhmc_sampler_destroy': (ctypes.c_int, [_P]),
    'mjhmc_set_hparams': (ctypes.c_int, [_P, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_double]),
    'mjhmc_iterate': (ctypes.c_int, [_P, ctypes.c_int, _P, _P, _P, ctypes.c_int, ctypes.POINTER(IterStats),
                                     ctypes.POINTER(ctypes.c_int)]),
    'mjhmc_host_set_energy': (ctypes.c_int, [_P, _P, _P]),
    'mjhmc_traj_begin': (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int64)]),
    'mjhmc_traj_step': (ctypes.c_int, [_P, _P, ctypes.c_int, _P]),
    'mjhmc_traj_finish': (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int, ctypes.POINTER(IterStats)]),
    'mjhmc_checkpoint': (ctypes.c_int, [_P]),
    'mjhmc_restore': (ctypes.c_int, [_P]),
    'mjhmc_rollback': (ctypes.c_int, [_P]),
    'mjhmc_get_tick': (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint64)]),
    'mjhmc_set_tick': (ctypes.c_int, [_P, ctypes.c_uint64]),
    'mjhmc_advance_tick': (ctypes.c_int, [_P, ctypes.c_int64]),
    'mjhmc_reset_flf_cache': (ctypes.c_int, [_P]),
    'mjhmc_read': (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_size_t]),
    'mjhmc_write': (ctypes.c_int, [_P, ctypes.c_int, _P, ctypes.c_size_t]),
    'mjhmc_ring_alloc': (ctypes.c_int, [_P, ctypes.c_int]),
    'mjhmc_ring_slot_bytes': (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint64)]),
    'mjhmc_mem_info': (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    'mjhmc_iterate_download': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.POINTER(IterStats), ctypes.POINTER(ctypes.c_int)]),
    'mjhmc_ring_read_dwell': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P]),
    'mjhmc_ring_gather': (ctypes.c_int, [_P, _P, ctypes.c_int64, _P]),
    'mjhmc_ring_read': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    'mjhmc_ring_moments': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_double, _dp, _dp]),
    'mjhmc_leapfrog': (ctypes.c_int, [_P, ctypes.c_int, _P, _P, ctypes.c_int64, ctypes.c_double, ctypes.c_int, _P, _P, _P, _P, _P]),
    'mjhmc_ring_autocor': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    'mjhmc_autocor': (ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _P]),
    'mjhmc_draw_from': (ctypes.c_int, [_P, _P, _P, ctypes.c_int64, _P, ctypes.POINTER(ctypes.c_int64)]),
    'mjhmc_min_idx': (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int64, _P]),
    'mjhmc_comm_unique_id': (ctypes.c_int, [_P]),
    'mjhmc_comm_create': (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, _P, ctypes.POINTER(_P)]),
    'mjhmc_comm_destroy': (ctypes.c_int, [_P]),
    'mjhmc_comm_available': (ctypes.c_int, []),
    'mjhmc_comm_count': (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int)]),
    'mjhmc_comm_allreduce_i64': (ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.c_int]),
    'mjhmc_comm_allreduce_f64': (ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.c_int]),
    'mjhmc_comm_bcast': (ctypes.c_int, [_P, _P, ctypes.c_size_t, ctypes.c_int]),
    'mjhmc_comm_allgatherv': (ctypes.c_int, [_P, _P, _P, _P]),
    'mjhmc_comm_allgather_ring': (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    'mjhmc_comm_allgather_columns': (ctypes.c_int, [_P, _P, _P, ctypes.c_int64, _P, _P]),
    'mjhmc_last_timing': (ctypes.c_int, [_P, _dp, _dp, ctypes.POINTER(ctypes.c_int)]),
    'mjhmc_set_timing': (ctypes.c_int, [_P, ctypes.c_int]),
    'mjhmc_sync': (ctypes.c_int, [_P]),
}

# libmjhmc_hip_test.so only (csrc/Makefile: test_hooks; built with -DMJHMC_TEST_HOOKS): the SAME sources plus the
# environment A/B switches of the launch strategies, the failure-placing hook and the single-GPU gather hooks
TEST_HOOKS_PATH = os.path.join(_HERE, 'lib', 'libmjhmc_hip_test.so')
TEST_HOOK_PROTOTYPES = {
    'mjhmc_test_gather_ring_local': (ctypes.c_int, [ctypes.POINTER(_P), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P]),
    'mjhmc_test_gather_columns_local': (ctypes.c_int, [ctypes.POINTER(_P), ctypes.c_int, _P, _P, _P]),
}

_lib = None
_hooks = None


def _attach(lib, protos):
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load the shared library (once) and attach prototypes.  Raises EngineError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError('libmjhmc_hip.so not found at %s -- build it with `make -C mjhmc_amd/csrc` '
                          '(or python -c "import __graft_entry__ as g; g.build()"). There is no CPU fallback.'
                          % LIB_PATH)
    _lib = _attach(ctypes.CDLL(LIB_PATH), PROTOTYPES)
    return _lib


def load_test_hooks():
    """The test build of the library (tests and measurement tools only; the product never loads it): pass it to
    engine.Context(device, lib=...)."""
    global _hooks
    if _hooks is None:
        if not os.path.exists(TEST_HOOKS_PATH):
            raise EngineError('libmjhmc_hip_test.so not found at %s -- `make -C mjhmc_amd/csrc test_hooks`' % TEST_HOOKS_PATH)
        _hooks = _attach(_attach(ctypes.CDLL(TEST_HOOKS_PATH), PROTOTYPES), TEST_HOOK_PROTOTYPES)
    return _hooks


def check(rc, lib=None):
    if rc != 0:
        msg = (lib or load()).mjhmc_last_error()
        raise EngineError('libmjhmc_hip: %s (status %d)' % (msg.decode() if msg else '?', rc))


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError('expected shape %r, got %r' % (tuple(shape), a.shape))
    return a


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)
