#!/bin/bash
# e2_pk: transposed-layout B fragments by single ds_read_b64 (tools/ab build) vs hipcc's merged ds_read2_b64 (product build)
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r03ab2; mkdir -p $O
run() { timeout 300 python tools/kbench.py --steps 6 --no-j --no-square "$@" 2>/dev/null | tail -1 | cut -c1-250 | tee -a $O/kbench_pk_trb64.log; }
for rep in 1 2; do
  unset PAMD_LIBRARY; run --tag "read2 K-only packed"
  export PAMD_LIBRARY=$PWD/tools/ab/libpyscf_amd_trb64.so; run --tag "trb64 K-only packed"
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/lds_trb64 -o p -- python $R/tools/kbench.py --steps 2 --no-j --no-square > $O/lds_trb64.log 2>&1
cd $R; unset PAMD_LIBRARY
python - <<'P'
import csv, glob, collections
f = glob.glob('gpurun_out/r03ab2/lds_trb64/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(float)
for row in csv.DictReader(open(f[0])):
    if 'e2_pk' in row['Kernel_Name']:
        acc[row['Counter_Name']] += float(row['Counter_Value'])
print('trb64 e2_pk', dict(acc))
P
