"""Scan gfx950 ISA text (hipcc -S --cuda-device-only) for a miscompile seen with ROCm 7.2's clang on the 512-VGPR step
kernels: a join block `L:` reached by `s_and_saveexec_b64 S, c ; s_cbranch_execz L` whose exec restore `s_or_b64 exec, exec, S`
is preceded, inside L, by exec-dependent instructions (live-range-split copies `v_accvgpr_write`, spill stores ...).  Lanes that
skipped the region do not execute those copies and later read stale registers (observed: a scratch address register holding
a float -> memory fault).  usage: python tools/scan_endcf.py file.s [...]   exit code 1 if any hit."""
import re
import sys


def copy_src(l):
  """source VGPR(s) of a save-type instruction (AGPR write / spill store / plain register move), else None"""
  m = re.match(r'v_accvgpr_write_b32 a\d+, (v\d+)$', l)
  if m:
    return [m.group(1)]
  m = re.match(r'scratch_store_dword\w* \w+, (v\[?[\d:]+\]?), .*Folded Spill', l)
  if m:
    return regs(m.group(1))
  return None


def regs(tok):
  m = re.match(r'v\[(\d+):(\d+)\]', tok)
  if m:
    return [f'v{k}' for k in range(int(m.group(1)), int(m.group(2)) + 1)]
  return [tok] if re.match(r'v\d+$', tok) else []


def defs(l):
  """VGPRs written by instruction l (first operand of VALU / load instructions)"""
  t = l.replace(',', ' ').split()
  if len(t) < 2 or not t[0].startswith(('v_', 'scratch_load', 'ds_read', 'flat_load', 'global_load', 'buffer_load')):
    return []
  return regs(t[1])


def scan(path):
  lines = open(path).read().split('\n')
  label = {}
  for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
      label[m.group(1)] = i
  hits = []
  for i, l in enumerate(lines):
    m = re.match(r'\s+s_and_saveexec_b64 (s\[\d+:\d+\]),', l)
    if not m:
      continue
    S = m.group(1)
    # the branch over the region follows within a few instructions
    tgt = None
    for j in range(i + 1, min(i + 4, len(lines))):
      b = re.match(r'\s+s_cbranch_execz (\.LBB\d+_\d+)', lines[j])
      if b:
        tgt = b.group(1)
        break
    if tgt is None or tgt not in label:
      continue
    pre = []
    for j in range(label[tgt] + 1, len(lines)):
      t = lines[j].strip()
      if not t or t.startswith(';'):
        continue
      if t.startswith('.'):
        if re.match(r'^\.LBB', lines[j]):
          break
        continue
      if t.startswith('s_or_b64 exec, exec, '):
        if t.split(', ')[-1].split()[0] == S:
          # registers defined inside the region (then-body) or earlier in the join block: copies of THOSE are phi-style
          # moves that only the region's lanes need; a save of a register defined outside the region is the bug
          inside = set()
          for q in lines[i + 1:label[tgt]]:
            inside.update(defs(q.strip()))
          bad = []
          for q in pre:
            src = copy_src(q)
            if src and not all(r in inside for r in src):
              bad.append(q)
            inside.update(defs(q))
          if bad:
            hits.append((i + 1, label[tgt] + 1, bad))
        break
      if re.match(r's_\w+ .*\bexec\b', t) or t.startswith(('s_cbranch', 's_branch', 's_setpc', 's_swappc', 's_endpgm')):
        break
      pre.append(t)
  return hits


if __name__ == '__main__':
  bad = 0
  for p in sys.argv[1:]:
    h = scan(p)
    bad += len(h)
    print(f'{p}: {len(h)} join block(s) that save an outer register before their own exec restore')
    for a, b, pre in h[:8]:
      print(f'   saveexec at line {a}, join at line {b}: {pre[:4]}')
  sys.exit(1 if bad else 0)
