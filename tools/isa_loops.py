"""List the loops (backward branches) of one kernel in a kept gfx950 ISA file with instruction-class counts per loop body.
  python tools/isa_loops.py boxlcd_amd/csrc/_obj/cfg_4_3_16_0/blcd_cfg-hip-amdgcn-amd-amdhsa-gfx950.s 'step_kernelILi4ELi3ELi16ELi0ELb0' [min_insts]
A loop body = the lines between a label and the last backward branch to it (inner loops are counted inside outer ones)."""
import re
import sys
from collections import Counter


def classify(op):
  if op.startswith('v_accvgpr'):
    return 'acc'
  if op.startswith('v_cndmask'):
    return 'cnd'
  if op.startswith('v_mov'):
    return 'vmov'
  if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
    return 'lane'
  if op.startswith('v_cmp'):
    return 'vcmp'
  if op.startswith('v_'):
    return 'valu'
  if op.startswith(('s_waitcnt', 's_nop')):
    return 'wait'
  if op.startswith(('s_cbranch', 's_branch')):
    return 'br'
  if op.startswith('s_'):
    return 'salu'
  if op.startswith('scratch_'):
    return 'scratch'
  if op.startswith('ds_'):
    return 'lds'
  if op.startswith(('global_', 'buffer_', 'flat_')):
    return 'vmem'
  return 'other'


def main():
  path, kern = sys.argv[1], sys.argv[2]
  min_insts = int(sys.argv[3]) if len(sys.argv) > 3 else 200
  lines = open(path).read().split('\n')
  start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(kern) + r'\S*:', l))
  end = next(i for i in range(start, len(lines)) if lines[i].startswith('\t.end_amdhsa_kernel') or lines[i].startswith('.Lfunc_end'))
  body = lines[start:end]
  labels = {}
  insts = []   # (op, text)
  for l in body:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
      labels[m.group(1)] = len(insts)
      continue
    t = l.strip()
    if not t or t.startswith(('.', ';')):
      continue
    insts.append((t.split()[0], t))
  print(f'{kern}: {len(insts)} instructions, {len(labels)} labels')
  loops = {}
  for i, (op, t) in enumerate(insts):
    if op.startswith(('s_cbranch', 's_branch')):
      tgt = t.split()[-1]
      if tgt in labels and labels[tgt] <= i:
        loops[tgt] = max(loops.get(tgt, 0), i)
  for tgt, last in sorted(loops.items(), key=lambda kv: labels[kv[0]]):
    lo = labels[tgt]
    n = last - lo + 1
    if n < min_insts:
      continue
    cnt = Counter(classify(op) for op, _ in insts[lo:last + 1])
    print(f'{tgt:>12} [{lo:6d},{last:6d}] n={n:6d}  ' + ' '.join(f'{k}={v}' for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])))


if __name__ == '__main__':
  main()
