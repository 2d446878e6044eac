"""Per-kernel means of every counter found under <dir>/*/x_counter_collection.csv (rocprofv3 --pmc passes of tools/gpu_job.sh):
    python tools/pmc_table.py gpurun_out/xcpmc [kernel-name-substring ...]"""
import csv, glob, os, sys
root = sys.argv[1]
want = sys.argv[2:] or ['sub_vmat', 'sub_orb_dot', 'sub_scale']
rows = {}
for f in sorted(glob.glob(os.path.join(root, '*', '*counter_collection.csv'))):
    tag = os.path.basename(os.path.dirname(f))
    agg = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        short = next((w for w in want if w in k), None)
        if not short:
            continue
        d = agg.setdefault((short, r['Dispatch_Id']), {})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        if 'Start_Timestamp' in r:
            d['_dur_us'] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3
    per = {}
    for (short, _), d in agg.items():
        for c, v in d.items():
            per.setdefault(short, {}).setdefault(c, []).append(v)
    for short, cs in per.items():
        rows[(tag, short)] = {c: sum(v) / len(v) for c, v in cs.items()}
        rows[(tag, short)]['_n'] = len(next(iter(cs.values())))
for (tag, short), d in sorted(rows.items()):
    print('%-14s %-12s n=%-4d %s' % (tag, short, d['_n'], '  '.join('%s=%.4g' % (c, v) for c, v in sorted(d.items()) if c != '_n')))
