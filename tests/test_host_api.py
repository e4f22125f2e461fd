"""Host-side logic (no GPU): env catalogue, observation/action tables, spaces, sampling, scene lowering, C-ABI surface."""
import ctypes
import os
import re
import numpy as np
import pytest
import boxlcd_amd as B
from boxlcd_amd import utils


def test_env_map_matches_reference_catalogue():
  names = {'Dropbox', 'Bounce', 'Bounce2', 'Object2', 'Object3', 'Urchin', 'Luxo', 'UrchinCube', 'LuxoCube', 'UrchinBall',
           'LuxoBall', 'UrchinBalls', 'LuxoBalls', 'UrchinCubes', 'LuxoCubes', 'Crab', 'CrabCube', 'SpiderCube'}
  assert set(B.env_map) == names                       # reference boxLCD/envs.py:17-137
  assert B.ENV_DG.fps == 10 and B.ENV_DG.lcd_base == 16 and B.ENV_DG.wh_ratio == 2.0


def test_obs_keys_sorted_and_sizes():
  e = B.envs.LuxoBall()
  assert e.obs_keys == sorted(e.obs_keys) and e.obs_size == 20 and e.act_size == 3
  assert e.obs_keys[:4] == ['luxo0:lfoot:cos', 'luxo0:lfoot:sin', 'luxo0:lfoot:x:p', 'luxo0:lfoot:y:p']
  assert e.pobs_keys == [k for k in e.obs_keys if not k.startswith('object')] and e.pobs_size == 16
  assert e.WIDTH == 7 and e.observation_space.spaces['lcd'].shape == (16, 24)      # int(1.5*5) = 7
  d = B.envs.Dropbox()
  assert d.obs_keys == ['object0:cos', 'object0:sin', 'object0:x:p', 'object0:y:p'] and d.act_keys == ['dummy']
  assert d.observation_space.spaces['proprio'].shape == (1,) and d.G.ep_len == 25
  u = B.envs.Urchin()
  assert u.act_keys == ['urchin0:aleg:speed', 'urchin0:bleg:speed', 'urchin0:cleg:speed'] and u.obs_size == 16
  assert u.observation_space.spaces['lcd'].shape == (16, 32) and u.action_space.shape == (3,)
  c = B.envs.Crab()
  assert c.scene.desc.n_bodies == 17 and c.scene.desc.n_joints == 16 and c.act_size == 12   # 4 fixed claw tips
  with pytest.raises(NotImplementedError):
    B.envs.Urchin({'use_speed': 0})                    # the reference's non-default branches are broken (App. E)


def test_G_overrides_and_namespace():
  import argparse
  e = B.envs.Bounce({'ep_len': 7})
  assert e.G.ep_len == 7 and e.G.wh_ratio == 1.0
  e = B.envs.Bounce(argparse.Namespace(fps=30))
  assert e.scene.desc.substeps == 1 and abs(e.scene.desc.dt - np.float32(1 / 30)) < 1e-9


def test_mapto_rmapto_roundtrip_and_namedarray():
  lo_hi = utils.A[0, 7]
  x = np.linspace(-1, 1, 11)
  assert np.allclose(utils.rmapto(utils.mapto(x, lo_hi), lo_hi), x)
  info = {'a:x:p': utils.A[0, 10], 'a:cos': utils.A[-1, 1]}
  na = utils.NamedArray(np.zeros(2), info)
  na['a:x:p'] = 7.5
  assert na.arr[0] == 0.5 and na['a:x:p'] == 7.5 and na('a:cos') == 0.0
  na['a:x:p', 'a:cos'] = np.array([10.0, -1.0])
  assert (na.arr == [1.0, -1.0]).all() and (na[['a:x:p', 'a:cos']] == [10.0, -1.0]).all()


def test_gym_compatible_seeding_and_sampling_ranges():
  e1, e2 = B.envs.Object2(), B.envs.Object2()
  e1.seed(3); e2.seed(3)
  p1, _ = e1._sample_poses(lambda lo, hi: np.array([e1.np_random.uniform(lo, hi)]), 1)
  p2, _ = e2._sample_poses(lambda lo, hi: np.array([e2.np_random.uniform(lo, hi)]), 1)
  assert (p1 == p2).all()
  # known value of gym 0.17.3's seeding.np_random(0): RandomState seeded with sha512('0')[:8] as uint32 list
  from boxlcd_amd.world_env import np_random
  rng, s = np_random(0)
  ref = np.random.RandomState(); 
  import hashlib, struct
  h = hashlib.sha512(b'0').digest()[:8] + b'\0' * 4
  big = sum(v << (32 * i) for i, v in enumerate(struct.unpack('3I', h)))
  ints = []
  while big > 0:
    big, m = divmod(big, 2**32); ints.append(m)
  ref.seed(ints)
  assert rng.uniform() == ref.uniform() and s == 0
  be = B.BatchedWorldEnv('Urchin', 500, seed=1)
  poses, sel = be.sample_initial(500)
  assert poses.shape == (500, 4, 3) and poses.dtype == np.float32
  assert poses[:, 0, 0].min() >= 1.25 - 1e-5 and poses[:, 0, 0].max() <= 8.75 + 1e-5 and np.allclose(poses[:, 0, 1], 1.25)
  # anchors coincide: leg position = root + R(leg)*(0,-0.6667)
  legs = poses[:, 1:, :]
  exp = poses[:, :1, :2] - np.stack([-np.sin(legs[..., 2]), np.cos(legs[..., 2])], -1) * (40 / 30.0 / 2)
  assert np.abs(legs[..., :2] - exp).max() < 1e-5
  bo = B.BatchedWorldEnv('Object2', 2000, seed=2)
  poses, sel = bo.sample_initial(2000)
  assert set(np.unique(sel)) == {0, 1} and poses[..., :2].min() >= 0.5 - 1e-6 and poses[..., :2].max() <= 4.5 + 1e-6


def test_state_to_poses_inverts_obs_normalisation():
  e = B.BatchedWorldEnv('LuxoBall', 3)
  fs = np.zeros((3, e.obs_size))
  na = utils.NamedArray(fs, e.obs_info)
  na['object0:x:p'] = 3.5; na['object0:y:p'] = 1.0; na['object0:cos'] = np.cos(0.7); na['object0:sin'] = np.sin(0.7)
  poses = e._state_to_poses(fs)
  i = e.scene.body_index['object0']
  assert np.allclose(poses[:, i], [3.5, 1.0, 0.7], atol=1e-6)


def test_scene_lowering_invariants():
  e = B.envs.LuxoBall()
  d = e.scene.desc
  assert (d.n_bodies, d.n_joints, d.n_obs, d.n_act) == (5, 3, 20, 3)
  assert [b.name for b in e.scene.bodies] == ['luxo0:root', 'luxo0:lhip', 'luxo0:lknee', 'luxo0:lfoot', 'object0']
  assert d.bodies[0].density == np.float32(0.1) and d.bodies[1].density == 1.0 and d.bodies[4].density == np.float32(0.2)
  assert d.bodies[0].mask_bits == 0x011 and d.bodies[4].category_bits == 0x0110 and d.bodies[4].mask_bits == 0xFFFF
  assert (d.joints[1].body_a, d.joints[1].body_b) == (1, 2) and d.joints[0].enable_limit == 1
  kinds = [d.obs[i].kind for i in range(d.n_obs)]
  assert kinds[:4] == [4, 5, 0, 1] and kinds[12:16] == [2, 3, 0, 1]      # links use transform.angle, root uses body.angle
  o2 = B.envs.Object2().scene.desc
  assert o2.bodies[0].n_choices == 2 and o2.shapes[o2.bodies[0].shape[0]].type == 0 and o2.shapes[o2.bodies[0].shape[1]].is_box == 1


def test_header_symbols_exported_by_library():
  """Every function include/boxlcd.h declares is exported by libboxlcd_hip.so and bound by the ctypes layer (no compute)."""
  from boxlcd_amd import _lib
  hdr = open('include/boxlcd.h').read()
  declared = set(re.findall(r'\b(blcd_[a-z_0-9]+)\s*\(', hdr))
  assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
  if not os.path.exists(_lib.LIB_PATH):
    pytest.skip('libboxlcd_hip.so not built (run __graft_entry__.build())')
  lib = ctypes.CDLL(_lib.LIB_PATH)
  for name in declared:
    assert hasattr(lib, name), name
  lib.blcd_version.restype = ctypes.c_int
  assert lib.blcd_version() == int(re.search(r"#define BLCD_VERSION (\d+)", hdr).group(1)) == 101


def test_scene_struct_layout_matches_c_header():
  from boxlcd_amd.scene import SceneDesc, ShapeDef, BodyDef, JointDef, ObsDef
  assert ctypes.sizeof(ShapeDef) == 16 + 64 and ctypes.sizeof(BodyDef) == 48 and ctypes.sizeof(JointDef) == 48
  assert ctypes.sizeof(ObsDef) == 16
  assert ctypes.sizeof(SceneDesc) == 64 + 24 * 80 + 20 * 48 + 20 * 48 + 96 * 16


def test_no_gpu_means_loud_failure():
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from boxlcd_amd import _lib
  if not os.path.exists(_lib.LIB_PATH):
    with pytest.raises(RuntimeError):
      B.envs.Dropbox().reset()
  else:
    with pytest.raises(RuntimeError, match='no HIP device|hip'):
      B.envs.Dropbox().reset()


def test_product_never_imports_oracle():
  import glob
  for path in glob.glob('boxlcd_amd/**/*', recursive=True):
    if os.path.isfile(path) and path.endswith(('.py', '.h', '.hip', '.cpp')):
      txt = open(path, errors='ignore').read()
      assert 'b2o_' not in txt and 'pyb2o' not in txt and 'oracle/' not in txt.replace('the parity oracle', ''), path


def test_goal_wrapper_column_selection():
  """BodyGoalEnv compares every '(x|y):p' entry of proprio (body_goal.py:63-64), CubeGoalEnv the objects' (cube_goal.py:12-13);
  host logic only (no GPU until reset())."""
  from boxlcd_amd.goal import BodyGoalEnv, CubeGoalEnv
  env = B.envs.UrchinCube()
  body = BodyGoalEnv(env, {})
  assert [env.obs_keys[c] for c in body._cols] == [k for k in env.pobs_keys if k.endswith(('x:p', 'y:p'))]
  assert all(not env.obs_keys[c].startswith('object') for c in body._cols) and len(body._cols) == 8
  cube = CubeGoalEnv(env, {})
  assert cube.keys == ['object0:x:p', 'object0:y:p'] and [env.obs_keys[i] for i in cube.root_idxs] == ['urchin0:root:x:p', 'urchin0:root:y:p']
  assert set(body.observation_space.spaces) >= {'goal:lcd', 'goal:proprio'}
  assert cube.observation_space.spaces['goal:object'].shape == (2,)


def test_step_kernel_isa_passes_the_exec_restore_scan(tmp_path):
  """Guard against a ROCm 7.2 clang miscompile of the 512-VGPR step kernels: a join block that saves an outer
  register (AGPR write / spill store) BEFORE restoring its own exec mask.  build() keeps each class's device ISA
  (csrc/_obj/cfg_*/...s) and rebuilds a flagged class with the workaround; here the scanner is checked on a synthetic hit
  and every kept ISA file must be clean."""
  import sys
  import __graft_entry__ as G
  sys.path.insert(0, os.path.join(G.ROOT, 'tools'))
  import scan_endcf
  bad = tmp_path / 'bad.s'
  bad.write_text('\n'.join([
      '\tv_add_u32_e32 v131, 24, v40', '\ts_and_saveexec_b64 s[0:1], s[16:17]', '\ts_cbranch_execz .LBB1_2', '; %bb.1:',
      '\tv_mov_b32_e32 v14, 6', '.LBB1_2:', '\tv_accvgpr_write_b32 a43, v131', '\ts_or_b64 exec, exec, s[0:1]', '\ts_endpgm']))
  assert len(scan_endcf.scan(str(bad))) == 1
  good = tmp_path / 'good.s'
  good.write_text(bad.read_text().replace('\tv_accvgpr_write_b32 a43, v131\n\ts_or_b64 exec, exec, s[0:1]',
                                           '\ts_or_b64 exec, exec, s[0:1]\n\tv_accvgpr_write_b32 a43, v131'))
  assert scan_endcf.scan(str(good)) == []
  objdir = os.path.join(G.ROOT, 'boxlcd_amd', 'csrc', '_obj')
  isas = [os.path.join(objdir, d, 'blcd_cfg-hip-amdgcn-amd-amdhsa-gfx950.s') for d in sorted(os.listdir(objdir)) if d.startswith('cfg_')] \
      if os.path.isdir(objdir) else []
  isas = [p for p in isas if os.path.exists(p)]
  if not isas:
    pytest.skip('no kept ISA (library was built elsewhere)')
  for p in isas:
    assert scan_endcf.scan(p) == [], p


def test_reset_stream_is_counter_based_and_sharding_invariant():
  """The batched reset stream: Philox4x32-10 keyed by the seed with counter (env id, reset count, variable, 0) - BASELINE.md §3's
  per-env counter-based RNG.  Known-answer vectors of the generator (Random123 kat_vectors), and the property that matters: an
  environment's start depends on (seed, env id, reset count) only - not on the batch size, not on which shard holds it."""
  import boxlcd_amd as B
  V = B.BatchedWorldEnv
  # KAT through the public helper: counter (0,0,0,0), key 0 -> first two words 6627e8d5 e169c58d
  u = V._philox_u01(0, [0], [0], 0)[0]
  assert u == ((0x6627e8d5 >> 5) * 67108864.0 + (0xe169c58d >> 6)) / 9007199254740992.0
  # counter word 3 = the high half of the (64-bit) global env id: the all-ones vector -> 408f276d 41c83b0e
  u = V._philox_u01(0xffffffffffffffff, [0xffffffffffffffff], [0xffffffff], 0xffffffff)
  assert u[0] == ((0x408f276d >> 5) * 67108864.0 + (0x41c83b0e >> 6)) / 9007199254740992.0
  # counter (243f6a88 85a308d3 13198a2e 03707344), key (a4093822 299f31d0) -> d16cfe09 94fdcceb (the digits-of-pi vector)
  u = V._philox_u01(0x299f31d0a4093822, [0x03707344243f6a88], [0x85a308d3], 0x13198a2e)
  assert u[0] == ((0xd16cfe09 >> 5) * 67108864.0 + (0x94fdcceb >> 6)) / 9007199254740992.0
  for name in ('Urchin', 'LuxoBall', 'Object2', 'Crab'):
    a = V(name, 64, seed=9)
    b = V(name, 7, seed=9)
    pa, sa = a.sample_initial(64)
    pb, sb = b.mirror_poses([5, 3], [0, 0])
    assert (pa[[5, 3]] == pb).all() and (sa[[5, 3]] == sb).all()
    pa2, _ = a.sample_initial(64)                      # second reset of the same environments: new draws
    assert (pa2 != pa).any() and (a.mirror_poses([5], [1])[0] == pa2[5]).all()
    c = V(name, 64, seed=10)
    assert (c.sample_initial(64)[0] != pa).any()
    # shards: same seed, env_id_base = first global id of the shard -> the shard IS that part of the batch
    s1 = V(name, 32, seed=9, env_id_base=32)
    p1, q1 = s1.sample_initial(32)
    assert (p1 == pa[32:]).all() and (q1 == sa[32:]).all()
