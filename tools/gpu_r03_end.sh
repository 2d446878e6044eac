#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r03end; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bench_launch.py -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pk_final -o p -- python $R/tools/kbench.py --steps 2 --no-j --no-square > $O/pk_final.log 2>&1
cd $R
python - <<'P'
import csv, glob, collections
f = glob.glob('gpurun_out/r03end/pk_final/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(float)
for row in csv.DictReader(open(f[0])):
    if 'e2_pk' in row['Kernel_Name']:
        acc[row['Counter_Name']] += float(row['Counter_Value'])
print('final e2_pk', dict(acc))
P
tail -1 $O/pk_final.log | cut -c1-250
