"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[-34:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(agg.items()):
    if 'hj::' not in k: continue
    out = {c: f"{x:.3g}" for c, x in v.items()}
    d = {}
    if 'SQ_THREAD_CYCLES_VALU' in v and v.get('SQ_INSTS_VALU'):
        d['lanes/inst'] = round(v['SQ_THREAD_CYCLES_VALU'] / v['SQ_INSTS_VALU'], 1)
    if 'SQ_WAVE_CYCLES' in v:
        for c in ('SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_ANY'):
            if c in v: d[c.replace('SQ_', '') + '/wavecyc'] = round(v[c] / v['SQ_WAVE_CYCLES'], 3)
    print(k, out, d)
