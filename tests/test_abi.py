"""CPU: the C-ABI library builds, loads and exports every symbol include/mjhmc_hip.h declares;
argument validation paths that need no GPU."""
import ctypes
import os
import re

import pytest

from mjhmc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'mjhmc_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mjhmc_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree(lib):
    names = declared_functions()
    assert len(names) >= 20
    assert set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)
    for n in names:
        assert hasattr(lib, n), n


def test_abi_version_and_no_device_error(lib):
    assert lib.mjhmc_abi_version() == 2
    import torch
    if torch.cuda.device_count() == 0:
        h = ctypes.c_void_p()
        rc = lib.mjhmc_ctx_create(0, ctypes.byref(h))
        assert rc == _lib.ERR_NO_DEVICE
        assert b'no HIP device' in lib.mjhmc_last_error()


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: constructing a sampler without a device raises, it does not degrade."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip('GPU present')
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import TestGaussian
    with pytest.raises(_lib.EngineError):
        MarkovJumpHMC(distribution=TestGaussian(ndims=2, nbatch=10), epsilon=0.1, beta=0.1)


def test_product_never_imports_the_oracle():
    import subprocess
    import sys
    code = ("import sys; import mjhmc_amd, mjhmc_amd.samplers.markov_jump_hmc, mjhmc_amd.misc.distributions; "
            "bad=[m for m in sys.modules if m.split('.')[0]=='oracle']; assert not bad, bad")
    subprocess.check_call([sys.executable, '-c', code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'mjhmc_amd')):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), f


def test_hyperparameter_validation_is_host_side():
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import TestGaussian
    with pytest.raises(ValueError):
        MarkovJumpHMC(distribution=TestGaussian(ndims=2, nbatch=10), beta=1.0)
    with pytest.raises(NotImplementedError):
        MarkovJumpHMC(Xinit=None, E=None, dEdX=None)


def test_user_expression_energies_compile_without_a_device(lib):
    """mjhmc_expr_check: hipRTC builds the engine's kernel templates around a pair of C expressions for gfx950 with no
    GPU present (LambdaDistribution(device_expr=...), include/mjhmc_hip.h: mjhmc_energy_create_expr)."""
    inc = _lib.KERNEL_HEADERS.encode()
    assert lib.mjhmc_expr_check(10, b"log(1.0 + x*x/p[d])", b"2.0*x/(p[d] + x*x)", inc) == 0, lib.mjhmc_last_error()
    assert lib.mjhmc_expr_check(700, b"0.5*x*x + p[0]*cos(x)", b"x - p[0]*sin(x)", inc) == 0, lib.mjhmc_last_error()
    rc = lib.mjhmc_expr_check(4, b"0.5*x*y", b"x", inc)
    assert rc == -1 and b"undeclared identifier 'y'" in lib.mjhmc_last_error()
    # coordinates coupled through per-particle statistics S[k] (Neal's funnel as expressions)
    assert lib.mjhmc_expr_check_coupled(
        32, b"d == 0 ? x : 0.0; d == 0 ? 0.0 : x*x", b"0.0",
        b"S[0]*S[0]/(2*p[0]*p[0]) + 0.5*exp(-S[0])*S[1] + 0.5*(p[1]-1)*S[0]",
        b"d == 0 ? x/(p[0]*p[0]) - 0.5*exp(-x)*S[1] + 0.5*(p[1]-1) : x*exp(-S[0])", inc) == 0, lib.mjhmc_last_error()
    rc = lib.mjhmc_expr_check_coupled(8, b"x*x", b"S[0]*x", None, b"S[1]*x + T", inc)
    assert rc == -1 and b"undeclared identifier 'T'" in lib.mjhmc_last_error()


def test_unloadable_rccl_and_hiprtc_are_errors_not_crashes():
    """librccl / libhiprtc are dlopen'ed on first use; when that fails the entry points return MJHMC_ERR_COMM /
    MJHMC_ERR_INVALID with the loader's message (dlerror() is consumed ONCE -- a second call returns NULL)."""
    import subprocess
    import sys
    code = """
import ctypes, sys
sys.path.insert(0, %r)
from mjhmc_amd import _lib
lib = _lib.load()
assert lib.mjhmc_comm_available() == -6, lib.mjhmc_last_error()
msg = lib.mjhmc_last_error()
assert b'librccl.so could not be loaded' in msg and b'/nonexistent/librccl.so' in msg, msg
buf = ctypes.create_string_buffer(128)
assert lib.mjhmc_comm_unique_id(buf) == -6
rc = lib.mjhmc_expr_check(4, b"0.5*x*x", b"x", _lib.KERNEL_HEADERS.encode())
assert rc != 0 and b'libhiprtc.so could not be loaded' in lib.mjhmc_last_error(), lib.mjhmc_last_error()
print('load failures ok')
""" % ROOT
    env = dict(os.environ, MJHMC_RCCL_LIB='/nonexistent/librccl.so', MJHMC_HIPRTC_LIB='/nonexistent/libhiprtc.so')
    env.pop('MJHMC_HIP_LIB', None)
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0 and 'load failures ok' in p.stdout.decode(), p.stdout.decode()[-2000:]


def test_rccl_loads_here(lib):
    assert lib.mjhmc_comm_available() == 0, lib.mjhmc_last_error()


def test_shipped_library_reads_no_switches_test_build_does():
    """The A/B environment switches and the failure-placing hook are compiled into libmjhmc_hip_test.so only."""
    hooks = _lib.TEST_HOOKS_PATH
    if not os.path.exists(hooks):
        import __graft_entry__ as g
        g.build()
    ship = open(_lib.LIB_PATH, 'rb').read()
    test = open(hooks, 'rb').read()
    for name in (b'MJHMC_DEBUG_POISON', b'MJHMC_NO_FUSE', b'MJHMC_NO_SPLIT', b'MJHMC_NO_COMPACT', b'MJHMC_CHUNKS_PER_LANE',
                 b'MJHMC_AUTOCOR_STAGING_MB', b'mjhmc_test_gather_ring_local'):
        assert name not in ship, name
        assert name in test, name
    envs = set(re.findall(rb'MJHMC_[A-Z_]{3,}', ship)) - {b'MJHMC_ERR_', b'MJHMC_E_'}
    assert {e for e in envs if not e.startswith((b'MJHMC_E_', b'MJHMC_ERR', b'MJHMC_F', b'MJHMC_MODE', b'MJHMC_OP', b'MJHMC_BF'))} \
        <= {b'MJHMC_RCCL_LIB', b'MJHMC_HIPRTC_LIB', b'MJHMC_HIPFFT_LIB', b'MJHMC_JUMP_WAVES', b'MJHMC_ABI_VERSION', b'MJHMC_COMM_ID_BYTES'}, envs
    t = _lib.load_test_hooks()
    assert hasattr(t, 'mjhmc_test_gather_ring_local') and hasattr(t, 'mjhmc_iterate')


def test_host_side_under_address_sanitizer(tmp_path):
    """`make asan`: the translation units that hold host logic (handles, argument checks, RCCL / hipRTC plumbing) built
    with AddressSanitizer on the host side and driven through their error paths and the hipRTC compile (no device
    needed; GPU ASan is not available on this pool)."""
    import subprocess
    import sys
    clang = '/opt/rocm/lib/llvm/bin/clang'
    if not os.path.exists(clang):
        pytest.skip('no ROCm clang')
    rt = subprocess.check_output([clang, '-print-file-name=libclang_rt.asan-x86_64.so']).decode().strip()
    if not os.path.exists(rt):
        pytest.skip('no ASan runtime')
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'mjhmc_amd', 'csrc'), 'asan', '-j8'], stdout=subprocess.DEVNULL)
    drive = tmp_path / 'drive.py'
    drive.write_text('''
import ctypes, sys
sys.path.insert(0, %r)
from mjhmc_amd import _lib
lib = _lib.load()
h = ctypes.c_void_p()
lib.mjhmc_ctx_create(99, ctypes.byref(h))
inc = _lib.KERNEL_HEADERS.encode()
assert lib.mjhmc_expr_check(10, b"0.5*x*x/p[0]", b"x/p[0]", inc) == 0
assert lib.mjhmc_expr_check(4, b"0.5*x*y", b"x", inc) == -1 and b"undeclared" in lib.mjhmc_last_error()
assert lib.mjhmc_expr_check(0, b"x", b"x", inc) == -1
assert lib.mjhmc_energy_create(None, 0, 2, None, 0, None) == -1
assert lib.mjhmc_iterate(None, 1, None, None, None, -1, None, None) == -1
assert lib.mjhmc_comm_create(None, 0, 1, None, None) == -1
assert lib.mjhmc_rollback(None) == -1 and lib.mjhmc_get_tick(None, None) == -1
print("asan drive ok")
''' % ROOT)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0',
               MJHMC_HIP_LIB=os.path.join(ROOT, 'mjhmc_amd', 'lib', 'libmjhmc_hip_asan.so'))
    p = subprocess.run([sys.executable, str(drive)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and 'asan drive ok' in out and 'AddressSanitizer' not in out, out[-3000:]
