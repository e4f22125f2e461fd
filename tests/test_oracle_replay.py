"""The CPU oracle against the reference's published recordings, replayed from the RECORDER'S OWN INPUTS: env.seed(S) reset
sample + RandomState(A) action tape (reference research/scripts/evaluations/demo_imgs.py:58-72; tests/replay.py).  No start
pose is fitted anywhere.  LCD frames use the oracle's raster (variant 2); the 8x RGB view is drawn with the installed Pillow.

What reproduces them (DESIGN.md §2 has the full switch table; tools/replay_gifs.py --table regenerates it):
  pybox2d 2.3.10 bundles **Box2D 2.3.0** — first-order damping `v *= clamp(1 - h*c, 0, 1)`, `b2Sweep::Advance` as a lerp,
  hill-climbing b2FindMaxSeparation with the 0.98/0.001 reference-face rule — and the recordings below marked EXACT were made
  on a **glibc <= 2.27 libm** (sinf/cosf correctly rounded but for ~1e-7 of inputs).  Child links are placed with pybox2d's
  float32 b2Vec2 arithmetic (world_env.py:250)."""
import numpy as np
import pytest
import replay as R

EXACT = ['Dropbox', 'Bounce', 'Bounce2', 'Object2', 'Object2_circles', 'Object2_cubes', 'UrchinBall']


@pytest.mark.parametrize('gif', EXACT)
def test_recording_reproduced_exactly_from_recorder_inputs(oracle, gif):
  """Every LCD frame AND every 8x RGB frame (25.6-27.4 px/unit) of the recording, 26-150 frames, chaotic for the robots."""
  bad_lcd, bad_rgb = R.replay(gif, oracle)
  assert sum(bad_lcd) == 0, R.summary(bad_lcd)
  assert sum(bad_rgb) == 0, R.summary(bad_rgb)


def test_urchin_cube_recording(oracle):
  """UrchinCube.gif (robot + damped box, limb-box b2CollidePolygons): all 150 LCD frames; the 8x view differs in 4 late frames."""
  bad_lcd, bad_rgb = R.replay('UrchinCube', oracle)
  assert sum(bad_lcd) == 0, R.summary(bad_lcd)
  s = R.summary(bad_rgb)
  assert s['exact'] >= 146 and s['first_miss'] >= 142, s


def test_box2d_230_forms_are_what_the_recordings_need(oracle):
  """Each Box2D >= 2.3.1 form, switched on alone, breaks a recording that the 2.3.0 form reproduces exactly."""
  with oracle.variants(damping=0):
    assert R.summary(R.replay('UrchinCube', oracle, want_rgb=False)[0])['first_miss'] == 4     # cube falls with Pade damping
  with oracle.variants(polygons=0):
    assert R.summary(R.replay('UrchinCube', oracle, want_rgb=False)[0])['exact'] < 150
  with oracle.variants(advance=0):
    assert R.summary(R.replay('Object2_cubes', oracle, want_rgb=False)[0])['exact'] < 50
    assert R.summary(R.replay('UrchinBall', oracle, want_rgb=False)[0])['exact'] < 150
  with oracle.variants(sincos=0):
    assert R.summary(R.replay('UrchinBall', oracle, want_rgb=False)[0])['exact'] < 150


def test_four_recordings_pin_the_oracle_only_for_a_prefix(oracle):
  """PARITY IS OPEN HERE, and this test states exactly how far it is closed: Urchin.gif, Luxo.gif, LuxoBall.gif and
  LuxoCube.gif are reproduced at 25.6 px/unit for their first 17 / 66 / 57 / 33 frames ONLY; past those frames the oracle
  (and therefore the HIP path) is NOT pinned by them.  BASELINE configs[2] (Urchin) and [3] (LuxoBall) are among the four.
  Every candidate tried in rounds 2-3 (all 48 combinations of the five oracle switches, velocity/position iteration counts,
  a correctly rounded sincos, four FMA-contracting builds with GCC and clang at -O2/-O3) is tabulated in
  profiles/r03_open_recordings_sweep.md and DESIGN.md 2.1: none reproduces any of the four to its end.
  Inside those prefixes every LCD frame is exact - including Luxo frame 37 / LuxoBall frame 38, whose foot truncates to the one-pixel-high
  polygon (0,0),(5,0),(5,0),(0,0): until round 4 a tracked 6-pixel difference, now reproduced by the scan-position rule of raster variant 2
  (test_one_pixel_high_polygons_of_the_recordings below)."""
  for gif, first_miss in (('Urchin', 17), ('Luxo', 66), ('LuxoBall', 57), ('LuxoCube', 33)):
    bl, br = R.replay(gif, oracle)
    assert R.summary(br)['first_miss'] == first_miss, (gif, R.summary(br))     # pinned for exactly this prefix
    assert all(b == 0 for b in bl[:first_miss]), (gif, [i for i, b in enumerate(bl[:first_miss]) if b])


def test_one_pixel_high_polygons_of_the_recordings(oracle):
  """Eleven robot links of the exactly replayed recording frames truncate to a single pixel row (tools/degenerate_polys.py).  The
  recordings' Pillow draws exactly the two whose row begins at x = 0 and none of the other nine: raster variant 2 draws a row's
  horizontal edges from its scan position (which starts at 0) and skips one that begins to the right of it - the behaviour of the
  first Pillow releases that draw each polygon pixel once (9.0.x, the release the reference pins), fixed upstream later (Pillow 12
  draws all eleven).  The integer polygons below are the recordings' own (profiles/r04_param_sweep.md)."""
  W, H = 32, 16
  drawn = [[(0, 0), (5, 0), (5, 0), (0, 0)]]
  undrawn = [[(17, 0), (17, 0), (12, 0), (12, 0)], [(12, 0), (12, 0), (8, 0), (8, 0)], [(10, 3), (10, 3), (12, 3), (12, 3)], [(11, 2), (11, 2), (14, 2), (14, 2)],
             [(9, 2), (9, 2), (11, 2), (11, 2)], [(4, 3), (4, 3), (6, 3), (6, 3)], [(5, 0), (10, 0), (10, 0), (5, 0)], [(4, 0), (10, 0), (10, 0), (4, 0)]]
  for poly in drawn + undrawn:
    img = oracle.raster_polygon(np.array(poly, np.int32), W, H, 2)
    want = np.ones((H, W), np.uint8)
    if poly in drawn:
      want[0, 0:6] = 0
    assert (img == want).all(), poly
  # an ordinary quad with horizontal top and bottom edges: the spans already cover them, nothing is drawn twice or beyond
  quad = np.array([(3, 2), (9, 2), (9, 6), (3, 6)], np.int32)
  img = oracle.raster_polygon(quad, W, H, 2)
  want = np.ones((H, W), np.uint8)
  want[2:7, 3:10] = 0
  assert (img == want).all()


def test_seed7_sample_is_the_recordings_start(oracle):
  """One env step from the unrefined seed-S sample equals frame 0 of every recording at 8x resolution."""
  for gif in R.GIFS:
    bad_lcd, bad_rgb = R.replay(gif, oracle, frames=1)
    assert bad_lcd == [0] and bad_rgb == [0], gif
