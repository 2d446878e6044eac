"""BASELINE config 5 WHOLE on one GPU - (H2O)_128 cc-pVDZ, all 14 848 aux rows (560 GB) through the out-of-core C handle - and its
ENERGY against the oracle-only golden.  This module sorts FIRST on purpose: the worker process page-locks
~285 GB of host memory, and the GPU boxes of this pool run in a container limited to 300 GiB (cgroup `memory.max`; /proc/meminfo
shows the host's 3 TB) - r05 lost two boxes to the kernel's OOM killer when this test ran late in a whole-suite process whose
parent had grown by then.  At the start of the run the parent is small; the test also skips unless the container leaves the room,
and the handle itself refuses (PAMD_df_create: ...container's memory limit...) before the kernel would kill."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _container_memory_left_gb():
    """min(MemAvailable, cgroup limit - current usage) in GB: what this process tree may still take."""
    left = float('inf')
    try:
        with open('/proc/meminfo') as f:
            for line in f:
                if line.startswith('MemAvailable:'):
                    left = float(line.split()[1]) * 1e-6
    except OSError:
        pass
    for lim_p, cur_p in (('/sys/fs/cgroup/memory.max', '/sys/fs/cgroup/memory.current'),
                         ('/sys/fs/cgroup/memory/memory.limit_in_bytes', '/sys/fs/cgroup/memory/memory.usage_in_bytes')):
        try:
            lim = open(lim_p).read().strip()
            if lim == 'max':
                break
            lim, cur = int(lim), int(open(cur_p).read().strip())
            try:                                            # file cache is charged as well but reclaimed before anything is killed
                stat = dict(line.split() for line in open(os.path.join(os.path.dirname(lim_p), 'memory.stat')))
                cur -= min(cur, int(stat.get('inactive_file', 0)) + int(stat.get('active_file', 0)))
            except (OSError, ValueError):
                pass
            if 0 < lim < 1 << 60:
                left = min(left, (lim - cur) * 1e-9)
            break
        except (OSError, ValueError):
            continue
    return left


@pytest.mark.gpu
def test_config5_whole_tensor_out_of_core_on_one_gpu_vs_oracle_goldens():
    """All 14 848 aux rows (560 GB) on ONE GPU: resident rows + ~285 GB streamed from page-locked host memory per build.
      * J / K of a seeded local density against the sum of two oracle-only goldens covering every row (1e-9);
      * r05: J rows / (K C) rows and the energy functional at the orbitals of tests/golden/h2o128_ccpvdz_rhf_orbitals.npz against
        the oracle-only ENERGY golden (tools/gen_golden_energy_sweep.py), and the product's OWN SCF from its minao guess through
        the handle converging to that energy - 1e-8 Eh, pyscf/df/test/test_df_jk.py:57-59 at config-5 size."""
    # r06: ONE skip path - memory.  The goldens are committed (a missing one FAILS), a worker that
    # does not finish in 25 minutes FAILS (normal: 4-5; a hang in the part workers or the pinned pool is a product bug, not an
    # environment), and of the worker's error texts only the handle's own explicit refusals of the host allocation skip.  A skip
    # prints CONFIG5_SKIPPED in the suite summary (tests/conftest.py), so that a silent skip on a smaller box cannot pass for N3.
    for name in ('h2o128_ccpvdz_rows0-7424_local_oracle.json', 'h2o128_ccpvdz_rows7424-14848_local_oracle.json',
                 'h2o128_ccpvdz_energy_oracle.json'):
        assert os.path.exists(os.path.join(ROOT, 'tests', 'golden', name)), '%s is not in tests/golden' % name
    left = _container_memory_left_gb()
    if left < 305:
        pytest.skip('CONFIG5_SKIPPED memory: needs ~285 GB of page-locked host memory + margin inside the container limit, '
                    '%.0f GB left here' % left)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_native_cfg45_worker.py'), 'config5'], capture_output=True,
                       text=True, timeout=1500)                       # TimeoutExpired = failure
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', '_native_cfg45_worker_config5.log'), 'w') as f:
            f.write(p.stdout + p.stderr[-3000:])
    except OSError:
        pass
    out = p.stdout + p.stderr
    if p.returncode != 0 and 'AssertionError' not in out and any(k in out for k in (
            "container's memory limit", 'page-locked host memory could not be allocated')):
        # the handle itself refused the host rows (PAMD_df_create: ...): the environment cannot hold the case - not a parity statement
        pytest.skip('CONFIG5_SKIPPED memory: ' + out.strip().splitlines()[-1][-300:])
    assert p.returncode == 0 and 'NATIVE_CONFIG5_OK' in p.stdout, p.stdout[-3000:] + p.stderr[-4000:]
    # the energy leg (oracle-only golden + converged SCF, 1e-8 Eh) must have RUN, not been skipped
    assert 'NATIVE_CONFIG5_ENERGY_OK' in p.stdout, p.stdout[-3000:]
