#!/bin/bash
# second J pass: overlapped beside the plain SYRK, or in line before the re-tiled SYRK; auto-tuned per shape
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_df_jk.py tests/test_gpu_native_abi.py -q -m gpu -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
run() { timeout 300 python tools/kbench.py --steps 6 "$@" 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/kbench_j2_policy.log; }
run --j2-policy overlap --tag "config 3 J+K overlap"
run --j2-policy serial --tag "config 3 J+K serial"
run --j2-policy overlap --no-square --tag "config 3 packed J+K overlap"
run --j2-policy serial --no-square --tag "config 3 packed J+K serial"
run --nao 1856 --naux 556 --nocc 160 --j2-policy overlap --tag "config 3 1/8 shard overlap"
run --nao 1856 --naux 556 --nocc 160 --j2-policy serial --tag "config 3 1/8 shard serial"
timeout 900 python bench.py --molecule taxol --no-cpu-baseline --xc '' --steps 3 > $O/bench_taxol_1gpu.json 2> $O/bench_taxol_1gpu.err; cut -c1-420 $O/bench_taxol_1gpu.json; tail -2 $O/bench_taxol_1gpu.err
timeout 600 python bench.py --no-cpu-baseline --xc '' > $O/bench_cfg3.json 2> $O/bench_cfg3.err; tail -2 $O/bench_cfg3.err
python - <<'P'
import json
for n in ('bench_taxol_1gpu', 'bench_cfg3'):
    d = json.load(open('gpurun_out/r03z/%s.json' % n))
    print(n, d['value'], d['jk_schedule'], d['config']['workload'][-120:])
    print('  ', {k: v['ms_total'] for k, v in d['kernels'].items()})
P
