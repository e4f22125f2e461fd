"""Aggregate a rocprofv3 --pmc run (…_counter_collection.csv, one directory per counter pass) into JSON.

usage: pmc_summary.py <root of the passes> <out.json> [<launch log>]      (default log: <root>/<pass>/launch.log)

{kernel: {counter: {dispatches, mean, max}}} as before, plus - when the library's launch log of the same command is given
(BLCD_LAUNCH_LOG=<file>: one line per step_kernel dispatch, "env-steps world-steps slots cohort grid") -
  kernel -> "by_env_steps" -> {"<env-steps per launch>": {counter: {dispatches, mean, max}, "slots": mean slots per dispatch}}
so that counters of launches of different lengths (a jointed class's first rollout runs 50-step launches, the later ones
200-step launches) are never averaged together.  The k-th step_kernel dispatch of every pass is the k-th line of the log: the
command is deterministic and every pass runs it once.

"_meta" carries what bench.py needs to decide whether the summary still describes the code: a content hash of
boxlcd_amd/csrc + include + the build switches (tools/csrc_rev.py - unchanged by commits that do not touch the kernels) and the FETCH_SIZE note of MI355X_MICROARCH.md (gfx950 tallies a 128-B request at 64 B: wide coalesced reads are
under-reported by exactly 2x; other widths are uncalibrated - tools/micro/fetch_calib.hip measures this path's own pattern).
"""
import sys, csv, json, glob, collections, os

root, out = sys.argv[1], sys.argv[2]
log = sys.argv[3] if len(sys.argv) > 3 else None
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tree_rev():
  sys.path.insert(0, REPO)
  from tools.csrc_rev import csrc_rev
  return csrc_rev()


def stats(v):
  return {'dispatches': len(v), 'mean': sum(v) / len(v), 'max': max(v)}


def read_log(path):
  out_ = []
  if path and os.path.exists(path):
    for line in open(path):
      f = line.split()
      if len(f) >= 5:
        out_.append(tuple(int(x) for x in f[:5]))
  return out_


acc = collections.defaultdict(lambda: collections.defaultdict(list))
by_steps = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
slots = collections.defaultdict(lambda: collections.defaultdict(list))
mismatch, any_log = False, False
for f in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
  rows = list(csv.DictReader(open(f)))
  # this pass's own launch log: <root>/<pass>/launch.log (tools/profile.sh), or the one given on the command line
  rel = os.path.relpath(f, root).split(os.sep)
  launches = read_log(os.path.join(root, rel[0], 'launch.log')) or read_log(log)
  any_log = any_log or bool(launches)
  ids = sorted({int(r['Dispatch_Id']) for r in rows if 'step_kernel' in r['Kernel_Name']})   # dispatch order of the step kernels in this pass
  ordinal = {d: i for i, d in enumerate(ids)}
  if launches and len(launches) != len(ids):
    mismatch = True
    launches = []
  for r in rows:
    k = r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    if launches and 'step_kernel' in k:
      L = launches[ordinal[int(r['Dispatch_Id'])]]
      by_steps[k][str(L[0])][r['Counter_Name']].append(float(r['Counter_Value']))
      slots[k][str(L[0])].append(L[2])

res = {k: {c: stats(v) for c, v in cs.items()} for k, cs in acc.items()}
for k, groups in by_steps.items():
  res[k]['by_env_steps'] = {g: dict({c: stats(v) for c, v in cs.items()}, slots=sum(slots[k][g]) / len(slots[k][g])) for g, cs in groups.items()}
res['_meta'] = {'csrc_rev': tree_rev(), 'launch_log': any_log and not mismatch,
                'fetch_size_note': 'gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B: exactly 1/2 of the bytes of wide (16 B/lane) coalesced reads; '
                                   '4 B/lane reads (this path\'s state loads) per tools/micro/fetch_calib.hip',
                'units': 'FETCH_SIZE / WRITE_SIZE in KB per dispatch; SQ_* in quad-cycles / instructions per dispatch'}
if mismatch:
  print('WARNING: launch log does not match the dispatch sequence; by_env_steps dropped', file=sys.stderr)
  for k in by_steps:
    res[k].pop('by_env_steps', None)
json.dump(res, open(out, 'w'), indent=1)
for k, cs in res.items():
  if 'step_kernel' in k:
    print(k)
    for c, v in cs.items():
      if c != 'by_env_steps':
        print('   %-24s mean %.4g  (%d dispatches)' % (c, v['mean'], v['dispatches']))
    for g, gc in cs.get('by_env_steps', {}).items():
      print('   launches of %s env-steps, %.0f slots:' % (g, gc['slots']), ', '.join('%s %.4g (%d)' % (c, v['mean'], v['dispatches']) for c, v in gc.items() if c != 'slots'))
    w = cs.get('SQ_WAVE_CYCLES', {}).get('mean')
    if w:
      for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU'):
        if c in cs: print('   %s / WAVE_CYCLES = %.3f' % (c, cs[c]['mean'] / w))
      if 'SQ_THREAD_CYCLES_VALU' in cs and 'SQ_ACTIVE_INST_VALU' in cs:
        print('   lane utilisation = %.3f' % (cs['SQ_THREAD_CYCLES_VALU']['mean'] / (64 * cs['SQ_ACTIVE_INST_VALU']['mean'])))
