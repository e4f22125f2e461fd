"""Where a kept-ISA kernel's spill code sits: spill / reload instructions per innermost loop (backward-branch loops).
usage: python tools/spill_loops.py <file.s>"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN4blcd11step_kernel')][0]
end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith('s_endpgm')][0]
body = lines[start:end]
lab = {}
for i, l in enumerate(body):
  m = re.match(r'^(\.LBB\d+_\d+):', l)
  if m: lab[m.group(1)] = i
loops = []
for i, l in enumerate(body):
  m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
  if m:
    t = m.group(1) or m.group(2)
    if t in lab and lab[t] < i: loops.append((lab[t], i, t))
spill = [i for i, l in enumerate(body) if 'Folded Spill' in l or 'Folded Reload' in l]
print('spill/reload instructions', len(spill), 'of', len(body))
c = Counter()
for s in spill:
  inner = None
  for a, b, t in loops:
    if a <= s <= b and (inner is None or (b - a) < (inner[1] - inner[0])): inner = (a, b, t)
  c[inner] += 1
for k, v in c.most_common(12):
  print('  %4d  %s' % (v, 'outside any loop' if k is None else '%s (%d instructions, lines %d-%d)' % (k[2], k[1] - k[0], k[0], k[1])))
