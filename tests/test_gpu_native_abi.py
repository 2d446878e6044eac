"""The host-array, opaque-handle C ABI (include/pyscf_amd.h: PAMD_df_create, PAMD_df_get_jk, PAMD_df_export_cderi ...): driven
from plain numpy in a process that never imports torch.  Reference interface replaced: pyscf/df/df.py:147-267 +
pyscf/df/df_jk.py:280-413 with the caller-owned-buffer convention of df_jk.py:373-379."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _release_cached_hbm():
    """The worker is a separate process that hipMallocs for itself: hand it what this process's torch allocator has cached."""
    if 'torch' in sys.modules:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    yield


@pytest.mark.gpu
def test_host_array_abi_from_numpy_without_torch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_native_abi_worker.py')], capture_output=True, text=True,
                       timeout=900)
    try:                                               # keep the worker's timing lines where a GPU-box run can be inspected
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'native_abi_worker.log'), 'w') as f:
            f.write(p.stdout + p.stderr[-2000:])
    except OSError:
        pass
    assert p.returncode == 0 and 'NATIVE_ABI_OK' in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_native_wrapper_imports_without_torch_and_library_exports_the_handle_api():
    code = ("import sys; sys.path.insert(0, %r); from pyscf_amd.df import native; lib = native.load(); "
            "[getattr(lib, n) for n in ('PAMD_df_create', 'PAMD_df_get_jk', 'PAMD_df_naux', 'PAMD_df_nao', "
            "'PAMD_df_export_cderi', 'PAMD_df_destroy')]; assert 'torch' not in sys.modules; print('ok')" % ROOT)
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'ok' in p.stdout, p.stderr[-2000:]
