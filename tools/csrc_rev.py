"""Revision of the device code: sha256 over the step kernel's sources (boxlcd_amd/csrc/*.{h,hip} minus blcd_api.hip / blcd_render_ex.h) and the per-class build switches of
__graft_entry__.py (works without .git, so the GPU box and this container compute the same value).  tools/pmc_summary.py stamps
every profile summary with it; bench.py drops profile-derived blocks whose stamp differs from the code it is running."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_rev():
  h = hashlib.sha256()
  csrc = os.path.join(ROOT, 'boxlcd_amd', 'csrc')
  # the step kernel's sources: everything blcd_cfg.hip includes.  blcd_api.hip (host code, launch policy and the small
  # re-bin / reset / render kernels) and the any-size renderer are left out on purpose: an API-only change does not make
  # step_kernel's counters stale
  files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.h', '.hip')) and f not in ('blcd_api.hip', 'blcd_render_ex.h'))
  for p in files:
    h.update(os.path.basename(p).encode() + b'\0')
    h.update(open(p, 'rb').read())
  # build switches: the lines of __graft_entry__.py that define flags
  src = open(os.path.join(ROOT, '__graft_entry__.py')).read()
  for m in re.finditer(r'^(HIP_FLAGS|CFG_FLAGS|CONFIGS|CLASS_FLAGS)\s*=.*?(?=^\S)', src, re.M | re.S):
    h.update(m.group(0).encode())
  return h.hexdigest()[:16]


if __name__ == '__main__':
  print(csrc_rev())
