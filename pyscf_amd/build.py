"""In-tree build of libpyscf_amd.so (hipcc, gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'lib', 'libpyscf_amd.so')
SOURCES = ['capi.hip', 'df_handle.hip', 'df_jk.hip', 'int3c2e.hip', 'int2e.hip', 'int1e.hip', 'grid.hip', 'xc.hip', 'xc_sparse.hip', 'xc_handle.hip']
# (source, extra flags, object tag): the int3c2e family is compiled once per aux angular momentum
VARIANTS = [('int3c2e_lk.hip', ['-DPAMD_LK=%d' % lk], 'lk%d' % lk) for lk in range(7)] + \
           [('int3c2e_grad_lk.hip', ['-DPAMD_LK=%d' % lk], 'lk%d' % lk) for lk in range(7)]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, v[0]) for v in VARIANTS] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.inc'))]
    if (not force) and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest(deps):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    jobs = [(s, [], '') for s in srcs] + [(os.path.join(CSRC, s), fl, tag) for s, fl, tag in VARIANTS]
    for s, flags, tag in jobs:
        o = os.path.join(HERE, 'build', os.path.basename(s) + tag + '.o')
        objs.append(o)
        if (not force) and os.path.exists(o) and os.path.getmtime(o) >= _newest(
                [s] + [d for d in deps if d.endswith(('.h', '.inc'))]):
            continue
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + flags + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('build failed: ' + ' '.join(cmd))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
