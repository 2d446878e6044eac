#!/bin/bash
# PMC comparison of the two half transforms (square image vs packed + diagonal blocks), K only, one counter group per pass
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03pmc; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -oE "\b(SQ_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TA_[A-Z_0-9]+)\b" $O/avail.txt | sort -u > $O/avail_names.txt; wc -l $O/avail_names.txt
pass() { # name, counters..., then mode flag
  local name=$1; shift
  for mode in sq pk; do
    local fl=""; [ $mode = pk ] && fl="--no-square"
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/${name}_$mode -o p -- python $R/tools/kbench.py --steps 2 --no-j $fl > $O/${name}_$mode.log 2>&1
  done
}
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS
pass wait SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA
pass tcp TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum
pass ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_DATA_STALL_CYCLES_sum
cd $R
python - <<'P'
import csv, glob, os, collections
O = 'gpurun_out/r03pmc'
for d in sorted(glob.glob(O + '/*_*/')):
    files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not files:
        print(os.path.basename(d.rstrip('/')), 'no counters:', open(d.rstrip('/') + '.log').read()[-300:].replace('\n', ' | '))
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row['Kernel_Name']
        if 'e2_' not in k: continue
        k = k.split('<')[0].split('(')[0][-24:]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    for k, v in acc.items():
        print(os.path.basename(d.rstrip('/')), k, {c: '%.4g' % x for c, x in v.items()})
P
