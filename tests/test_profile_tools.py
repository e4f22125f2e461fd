"""The tooling behind bench.py's profile-derived blocks: counters grouped per launch length, revision stamp,
and bench.py refusing a stale or mismatching summary.  CPU only."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _write_pass(d, counter, values, kernel='void blcd::step_kernel<4, 3, 16, 0, false>(blcd::DevScene const*, float*)'):
  os.makedirs(d, exist_ok=True)
  with open(os.path.join(d, 'pmc_counter_collection.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Counter_Name', 'Counter_Value'])
    did = 10
    for v in values:                       # step_kernel dispatches interleaved with another kernel's
      w.writerow([did, kernel, 65344, counter, v]); did += 1
      w.writerow([did, 'reset_kernel(blcd::DevScene const*)', 64, counter, 1.0]); did += 1
  with open(os.path.join(d, 'launch.log'), 'w') as f:
    for steps in (50, 50, 50, 50, 200, 200):
      f.write(f'{steps} 0 50000 -1 1021\n')


def test_pmc_summary_groups_counters_by_launch_length_and_stamps_the_revision(tmp_path):
  from tools.csrc_rev import csrc_rev
  root = tmp_path / 'prof_pmc'
  _write_pass(str(root / 'FETCH_SIZE'), 'FETCH_SIZE', [10, 10, 10, 10, 40, 44])
  _write_pass(str(root / 'WRITE_SIZE'), 'WRITE_SIZE', [14, 14, 14, 14, 56, 60])
  out = tmp_path / 'sum.json'
  subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), str(root), str(out)], stdout=subprocess.DEVNULL)
  d = json.load(open(out))
  k = [x for x in d if 'step_kernel' in x][0]
  g = d[k]['by_env_steps']
  assert sorted(g) == ['200', '50']
  assert g['50']['FETCH_SIZE'] == {'dispatches': 4, 'mean': 10.0, 'max': 10.0} and g['200']['FETCH_SIZE']['mean'] == 42.0
  assert g['200']['WRITE_SIZE']['mean'] == 58.0 and g['200']['slots'] == 50000.0
  assert d[k]['FETCH_SIZE']['mean'] == 124 / 6                     # the undifferentiated mean is still there - and is NOT what bench.py uses
  assert d['_meta']['csrc_rev'] == csrc_rev() and d['_meta']['launch_log'] is True
  # a log that does not describe the dispatch sequence drops the groups instead of mis-assigning them
  with open(root / 'FETCH_SIZE' / 'launch.log', 'a') as f:
    f.write('200 0 50000 -1 1021\n')
  subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), str(root), str(out)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
  d = json.load(open(out))
  assert 'by_env_steps' not in d[k] and d['_meta']['launch_log'] is False


def test_bench_drops_profile_blocks_that_do_not_describe_the_running_code(tmp_path, monkeypatch):
  import bench
  from tools.csrc_rev import csrc_rev
  k = 'void blcd::step_kernel<4, 3, 16, 0, false>'
  grp = {'FETCH_SIZE': {'dispatches': 3, 'mean': 1.0e6, 'max': 1.0e6}, 'WRITE_SIZE': {'dispatches': 3, 'mean': 2.0e6, 'max': 2.0e6},
         'SQ_WAVE_CYCLES': {'mean': 100.0}, 'SQ_ACTIVE_INST_VALU': {'mean': 40.0}, 'SQ_WAIT_ANY': {'mean': 30.0}, 'SQ_THREAD_CYCLES_VALU': {'mean': 800.0},
         'SQ_WAVES': {'mean': 1021.0}, 'slots': 50000.0}
  prof = {k: dict(grp, by_env_steps={'200': grp}), '_meta': {'csrc_rev': csrc_rev(), 'launch_log': True}}
  path = tmp_path / 'p.json'
  json.dump(prof, open(path, 'w'))
  monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
  monkeypatch.setattr(bench, 'PMC_PROFILES', {('Urchin', 50000): 'p.json'})
  traffic, detail, valu = bench.pmc_blocks('Urchin', 50000, 1, 200.0)
  assert traffic == (2.0 * 1.0e6 + 2.0e6) * 1024.0              # 2 x FETCH_SIZE + WRITE_SIZE, KB -> B
  assert detail['traffic_uncorrected'] == 3.0e6 * 1024.0 and abs(valu['lanes_per_valu_inst'] - 20.0) < 1e-12
  # launches of a length the profile does not hold: no number rather than a wrong one
  traffic, detail, valu = bench.pmc_blocks('Urchin', 50000, 1, 50.0)
  assert traffic is None and valu is None and 'no profiled launches of 50' in detail['dropped']
  # another revision of the device code
  prof['_meta']['csrc_rev'] = 'deadbeefdeadbeef'
  json.dump(prof, open(path, 'w'))
  traffic, detail, valu = bench.pmc_blocks('Urchin', 50000, 1, 200.0)
  assert traffic is None and valu is None and 're-profile' in detail['dropped']


def test_committed_profiles_carry_the_revision_of_the_committed_kernels():
  """The r04 summaries bench.py reads were taken on exactly the step-kernel sources and build switches in the tree."""
  import bench
  from tools.csrc_rev import csrc_rev
  import pytest
  stale = []
  for key, path in bench.PMC_PROFILES.items():
    d = json.load(open(os.path.join(ROOT, path)))
    assert d['_meta']['launch_log'] is True
    if d['_meta']['csrc_rev'] != csrc_rev():
      stale.append(path)
  if stale:     # not an error (bench.py then reports `traffic: null` with the reason) - but it should be seen
    pytest.skip(f'profiles taken on other kernel sources than the tree holds ({csrc_rev()}): re-run tools/profile_all.sh: {stale}')
