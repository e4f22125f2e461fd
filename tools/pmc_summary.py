"""Aggregate a rocprofv3 --pmc run (…_counter_collection.csv) into {kernel: {counter: {dispatches, mean, max}}} JSON."""
import sys, csv, json, glob, collections
root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: {'dispatches': len(v), 'mean': sum(v) / len(v), 'max': max(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(res, open(out, 'w'), indent=1)
for k, cs in res.items():
  if 'step_kernel' in k:
    print(k)
    for c, v in cs.items():
      print('   %-24s mean %.4g  (%d dispatches)' % (c, v['mean'], v['dispatches']))
    w = cs.get('SQ_WAVE_CYCLES', {}).get('mean')
    if w:
      for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU'):
        if c in cs: print('   %s / WAVE_CYCLES = %.3f' % (c, cs[c]['mean'] / w))
      if 'SQ_THREAD_CYCLES_VALU' in cs and 'SQ_ACTIVE_INST_VALU' in cs:
        print('   lane utilisation = %.3f' % (cs['SQ_THREAD_CYCLES_VALU']['mean'] / (64 * cs['SQ_ACTIVE_INST_VALU']['mean'])))
