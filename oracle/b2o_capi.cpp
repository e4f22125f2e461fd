// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under boxlcd_amd/ may include, link or call this.
//
// Environment level of the CPU oracle: restates boxLCD/world_env.py `reset` (:306-317 world construction),
// `step` (:431-458), `_get_obs` (:387-429) and `lcd_render` (:460-512) on top of the Box2D restatement in
// b2o_world.h and the Pillow restatement in b2o_raster.h.  Parity status: "parity unpinned" against pybox2d
// (not installable here; the reference has no tests/golden vectors) — pinned instead by Pillow-12.2.0 goldens
// (rasteriser), the reference's published GIF frames and analytic known answers (tests/test_oracle_*.py).
#include <chrono>
#include <cstdio>
#include <thread>
#include "b2o_api.h"
#include "b2o_raster.h"
#include "b2o_world.h"

using namespace b2o;

struct b2o_env {
  b2o_scene_desc scene;
  World world;
  std::vector<int> shape_sel;
  std::vector<std::pair<int, int>> pairs;  // canonical pair slots (proxy ids)
};

static Shape BuildShape(const b2o_shape_def& d) {
  Shape s;
  std::memset(&s, 0, sizeof(s));
  if (d.type == 0) {
    ShapeSetCircle(&s, d.radius);
  } else if (d.is_box) {
    ShapeSetAsBox(&s, d.verts[0][0], d.verts[0][1]);
  } else {
    Vec2 vs[B2O_MAX_POLY_VERTS];
    for (int i = 0; i < d.n_verts; ++i) vs[i] = V2(d.verts[i][0], d.verts[i][1]);
    ShapeSetPolygon(&s, vs, d.n_verts);
  }
  return s;
}

static void BuildPairs(b2o_env* e) {
  // static part of b2ContactManager::AddPair: not both static, not joint-connected, category/mask filter
  e->pairs.clear();
  int nb = (int)e->world.bodies.size();
  for (int i = 0; i < nb; ++i)
    for (int j = i + 1; j < nb; ++j)
      if (e->world.ShouldCollide(j, i)) e->pairs.emplace_back(i, j);
}

extern "C" {

b2o_env* b2o_create(const b2o_scene_desc* scene) {
  b2o_env* e = new b2o_env();
  e->scene = *scene;
  return e;
}
void b2o_destroy(b2o_env* e) { delete e; }

void b2o_reset(b2o_env* e, const float* poses, const int32_t* shape_sel) {
  const b2o_scene_desc& S = e->scene;
  e->world = World();
  World& w = e->world;
  w.gravity = V2(S.gravity[0], S.gravity[1]);
  // world_env.py:311-314 — four static edge bodies: bottom, left, right, top (pybox2d fixture defaults)
  float W = S.world_w, H = S.world_h;
  Vec2 ev[4][2] = {{V2(0, 0), V2(W, 0)}, {V2(0, 0), V2(0, H)}, {V2(W, 0), V2(W, H)}, {V2(0, H), V2(W, H)}};
  for (int i = 0; i < 4; ++i) {
    Shape s;
    std::memset(&s, 0, sizeof(s));
    ShapeSetEdge(&s, ev[i][0], ev[i][1]);
    w.CreateBody(kStaticBody, V2(0.0f, 0.0f), 0.0f, s, 0.0f, 0.2f, 0.0f, 0x0001, 0xFFFF, 0.0f, 0.0f);
  }
  e->shape_sel.assign(S.n_bodies, 0);
  for (int i = 0; i < S.n_bodies; ++i) {
    const b2o_body_def& bd = S.bodies[i];
    int sel = shape_sel ? shape_sel[i] : 0;
    if (sel < 0 || sel >= bd.n_choices) sel = 0;
    e->shape_sel[i] = sel;
    Shape s = BuildShape(S.shapes[bd.shape[sel]]);
    w.CreateBody(kDynamicBody, V2(poses[3 * i + 0], poses[3 * i + 1]), poses[3 * i + 2], s, bd.density, bd.friction,
                 bd.restitution, bd.category_bits, bd.mask_bits, bd.linear_damping, bd.angular_damping);
  }
  for (int i = 0; i < S.n_joints; ++i) {
    const b2o_joint_def& jd = S.joints[i];
    w.CreateRevoluteJoint(4 + jd.body_a, 4 + jd.body_b, V2(jd.anchor_a[0], jd.anchor_a[1]), V2(jd.anchor_b[0], jd.anchor_b[1]),
                          jd.enable_limit != 0, jd.lower, jd.upper, jd.max_motor_torque);
  }
  BuildPairs(e);
}

void b2o_set_poses(b2o_env* e, const float* poses, const uint8_t* mask) {
  // reset(full_state=): body.position = ..., then body.angle = ... (two SetTransform calls per body, world_env.py:333-380)
  for (int i = 0; i < e->scene.n_bodies; ++i) {
    if (mask && !mask[i]) continue;
    Body& b = e->world.bodies[4 + i];
    e->world.SetTransform(4 + i, V2(poses[3 * i], poses[3 * i + 1]), b.sweep.a);
    e->world.SetTransform(4 + i, e->world.bodies[4 + i].xf.p, poses[3 * i + 2]);
  }
}

void b2o_set_motor_speeds(b2o_env* e, const float* action) {
  const b2o_scene_desc& S = e->scene;
  for (int i = 0; i < S.n_joints; ++i) {
    const b2o_joint_def& jd = S.joints[i];
    if (jd.action_index < 0) continue;
    // utils.py:117 mapto + world_env.py:441 (float64 glue; the value crosses into Box2D as float32)
    double a = (double)action[jd.action_index];
    double m = ((a + 1.0) / 2.0 * (double)(1 - (-1))) + (double)(-1);
    double c = m < -1.0 ? -1.0 : (m > 1.0 ? 1.0 : m);
    float speed = (float)((double)jd.speed * c);
    e->world.SetMotorSpeed(i, speed);
  }
}

void b2o_world_step(b2o_env* e) { e->world.Step(e->scene.dt, e->scene.vel_iters, e->scene.pos_iters); }

void b2o_env_step(b2o_env* e, const float* action) {
  if (action) b2o_set_motor_speeds(e, action);
  for (int i = 0; i < e->scene.substeps; ++i) b2o_world_step(e);
}

void b2o_get_obs(b2o_env* e, double* out) {
  const b2o_scene_desc& S = e->scene;
  for (int i = 0; i < S.n_obs; ++i) {
    const b2o_obs_def& od = S.obs[i];
    const Body& b = e->world.bodies[4 + od.body];
    double val = 0.0;
    switch (od.kind) {
      case 0: val = (double)b.xf.p.x; break;
      case 1: val = (double)b.xf.p.y; break;
      case 2: val = cos((double)b.sweep.a); break;
      case 3: val = sin((double)b.sweep.a); break;
      case 4: val = cos((double)atan2f(b.xf.q.s, b.xf.q.c)); break;
      case 5: val = sin((double)atan2f(b.xf.q.s, b.xf.q.c)); break;
    }
    double lo = (double)od.lo, hi = (double)od.hi;
    out[i] = ((val - lo) / (hi - lo) * 2.0) + -1.0;  // utils.py:119 rmapto
  }
}

static void RenderBodies(const b2o_scene_desc& S, int n, const Shape* shapes, const Transform* xfs, uint8_t* lcd) {
  int W = S.lcd_w, H = S.lcd_h;
  std::vector<uint8_t> img((size_t)W * H, 1);
  Canvas cv{W, H, img.data()};
  double WIDTH = (double)S.world_w, width = (double)W;
  for (int i = 0; i < n; ++i) {
    const Shape& sh = shapes[i];
    const Transform& xf = xfs[i];
    if (sh.type == kCircle) {
      double px = (double)xf.p.x, py = (double)xf.p.y, rad = (double)sh.radius;
      double tlx = (px - rad) / WIDTH * width, tly = (py - rad) / WIDTH * width;
      double brx = (px + rad) / WIDTH * width, bry = (py + rad) / WIDTH * width;
      if (!draw_ellipse(cv, (int)tlx, (int)tly, (int)brx, (int)bry)) {
        fprintf(stderr, "b2o: ellipse bbox outside LUT range\n");
      }
    } else {
      int xy[2 * kMaxPolygonVertices];
      for (int k = 0; k < sh.count; ++k) {
        Vec2 p = Mul(xf, sh.v[k]);
        xy[2 * k] = (int)((double)p.x / WIDTH * width);
        xy[2 * k + 1] = (int)((double)p.y / WIDTH * width);
      }
      draw_polygon(cv, xy, sh.count, S.raster_variant);
    }
  }
  for (int r = 0; r < H; ++r) std::memcpy(lcd + (size_t)r * W, img.data() + (size_t)(H - 1 - r) * W, W);
}

// lcd_render(width, height, lcd_mode) — reference world_env.py:460-512 with explicit size and mode: mode 0 = '1' (uint8 [H][W],
// 1 = background, fill only), mode 1 = 'RGB' (uint8 [H][W][3] AFTER the reference's `255 - lcd`: fill by body.color1, 1-px
// outline by body.color2; colours world_env.py:201,303,482-483).  Both axes scale by width / WIDTH (world_env.py:495-503).
static bool RenderBodiesEx(const b2o_scene_desc& S, int n, const Shape* shapes, const Transform* xfs, int W, int H, int mode, uint8_t* out) {
  const int C = mode ? 3 : 1;
  std::vector<uint8_t> img((size_t)W * H * C, 1);
  Canvas cv{W, H, img.data()};
  cv.C = C;
  double WIDTH = (double)S.world_w, width = (double)W;
  bool ok = true;
  for (int i = 0; i < n; ++i) {
    const Shape& sh = shapes[i];
    const Transform& xf = xfs[i];
    const bool robot = S.bodies[i].kind != 0;
    // int(255.0 * (1 - x)) of color1 / color2
    const uint8_t fillc[3] = {(uint8_t)(robot ? 25 : 127), 153, (uint8_t)(robot ? 153 : 25)};
    const uint8_t outc[3] = {(uint8_t)(robot ? 127 : 178), 178, 127};
    const uint8_t zero[3] = {0, 0, 0};
    if (sh.type == kCircle) {
      double px = (double)xf.p.x, py = (double)xf.p.y, rad = (double)sh.radius;
      double tlx = (px - rad) / WIDTH * width, tly = (py - rad) / WIDTH * width;
      double brx = (px + rad) / WIDTH * width, bry = (py + rad) / WIDTH * width;
      ok = draw_ellipse_rgb(cv, (int)tlx, (int)tly, (int)brx, (int)bry, mode ? fillc : zero, mode ? outc : nullptr) && ok;
    } else {
      int xy[2 * kMaxPolygonVertices];
      for (int k = 0; k < sh.count; ++k) {
        Vec2 p = Mul(xf, sh.v[k]);
        xy[2 * k] = (int)((double)p.x / WIDTH * width);
        xy[2 * k + 1] = (int)((double)p.y / WIDTH * width);
      }
      std::memcpy(cv.ink, mode ? fillc : zero, 3);
      draw_polygon(cv, xy, sh.count, S.raster_variant);
      if (mode) {
        std::memcpy(cv.ink, outc, 3);
        draw_polygon_outline(cv, xy, sh.count);
      }
    }
  }
  for (int r = 0; r < H; ++r)
    for (int x = 0; x < W * C; ++x) {
      uint8_t v = img[(size_t)(H - 1 - r) * W * C + x];
      out[(size_t)r * W * C + x] = mode ? (uint8_t)(255 - v) : v;
    }
  return ok;
}

void b2o_set_ellipse_rgb_lut(const uint8_t* lut, int32_t amax) {
  g_ellipse_rgb_lut = lut;
  g_ellipse_rgb_amax = amax;
}

int32_t b2o_render_ex(b2o_env* e, int32_t width, int32_t height, int32_t mode, uint8_t* out) {
  const b2o_scene_desc& S = e->scene;
  Shape shapes[B2O_MAX_BODIES];
  Transform xfs[B2O_MAX_BODIES];
  for (int i = 0; i < S.n_bodies; ++i) {
    shapes[i] = e->world.bodies[4 + i].shape;
    xfs[i] = e->world.bodies[4 + i].xf;
  }
  return RenderBodiesEx(S, S.n_bodies, shapes, xfs, width, height, mode, out) ? 0 : -1;
}

int32_t b2o_render_poses_ex(const b2o_scene_desc* scene, const float* poses, const int32_t* shape_sel, int32_t n, int32_t width,
                            int32_t height, int32_t mode, uint8_t* out) {
  const b2o_scene_desc& S = *scene;
  bool ok = true;
  for (int k = 0; k < n; ++k) {
    Shape shapes[B2O_MAX_BODIES];
    Transform xfs[B2O_MAX_BODIES];
    for (int i = 0; i < S.n_bodies; ++i) {
      int sel = shape_sel ? shape_sel[k * S.n_bodies + i] : 0;
      shapes[i] = BuildShape(S.shapes[S.bodies[i].shape[sel]]);
      const float* p = poses + ((size_t)k * S.n_bodies + i) * 3;
      xfs[i].p = V2(p[0], p[1]);
      xfs[i].q.Set(p[2]);
    }
    ok = RenderBodiesEx(S, S.n_bodies, shapes, xfs, width, height, mode, out + (size_t)k * width * height * (mode ? 3 : 1)) && ok;
  }
  return ok ? 0 : -1;
}

void b2o_render(b2o_env* e, uint8_t* lcd) {
  const b2o_scene_desc& S = e->scene;
  Shape shapes[B2O_MAX_BODIES];
  Transform xfs[B2O_MAX_BODIES];
  for (int i = 0; i < S.n_bodies; ++i) {
    shapes[i] = e->world.bodies[4 + i].shape;
    xfs[i] = e->world.bodies[4 + i].xf;
  }
  RenderBodies(S, S.n_bodies, shapes, xfs, lcd);
}

void b2o_render_poses(const b2o_scene_desc* scene, const float* poses, const int32_t* shape_sel, int32_t n, uint8_t* lcd) {
  const b2o_scene_desc& S = *scene;
  for (int k = 0; k < n; ++k) {
    Shape shapes[B2O_MAX_BODIES];
    Transform xfs[B2O_MAX_BODIES];
    for (int i = 0; i < S.n_bodies; ++i) {
      int sel = shape_sel ? shape_sel[k * S.n_bodies + i] : 0;
      shapes[i] = BuildShape(S.shapes[S.bodies[i].shape[sel]]);
      const float* p = poses + ((size_t)k * S.n_bodies + i) * 3;
      xfs[i].p = V2(p[0], p[1]);
      xfs[i].q.Set(p[2]);
    }
    RenderBodies(S, S.n_bodies, shapes, xfs, lcd + (size_t)k * S.lcd_w * S.lcd_h);
  }
}

int32_t b2o_num_pairs(b2o_env* e) { return (int32_t)e->pairs.size(); }
void b2o_pair_table(b2o_env* e, int32_t* pairs) {
  for (size_t i = 0; i < e->pairs.size(); ++i) {
    pairs[2 * i] = e->pairs[i].first;
    pairs[2 * i + 1] = e->pairs[i].second;
  }
}

void b2o_dump(b2o_env* e, float* bodies, float* joints, float* pairs) {
  const b2o_scene_desc& S = e->scene;
  World& w = e->world;
  if (bodies)
    for (int i = 0; i < S.n_bodies; ++i) {
      const Body& b = w.bodies[4 + i];
      float* o = bodies + i * B2O_BODY_STATE_FLOATS;
      o[0] = b.sweep.c.x; o[1] = b.sweep.c.y; o[2] = b.sweep.a; o[3] = b.v.x; o[4] = b.v.y; o[5] = b.w;
      o[6] = b.sleepTime; o[7] = b.awake ? 1.0f : 0.0f;
      o[8] = b.fat.lo.x; o[9] = b.fat.lo.y; o[10] = b.fat.hi.x; o[11] = b.fat.hi.y;
    }
  if (joints)
    for (int i = 0; i < S.n_joints; ++i) {
      const Joint& j = w.joints[i];
      float* o = joints + i * B2O_JOINT_STATE_FLOATS;
      o[0] = j.impulse.x; o[1] = j.impulse.y; o[2] = j.impulse.z; o[3] = j.motorImpulse; o[4] = (float)j.limitState;
    }
  if (pairs)
    for (size_t k = 0; k < e->pairs.size(); ++k) {
      float* o = pairs + k * B2O_PAIR_STATE_FLOATS;
      for (int q = 0; q < B2O_PAIR_STATE_FLOATS; ++q) o[q] = 0.0f;
      int cid = w.FindContact(e->pairs[k].first, e->pairs[k].second);
      if (cid < 0) continue;
      const Contact& c = w.contacts[cid];
      o[0] = 1.0f;
      o[1] = c.touching ? 1.0f : 0.0f;
      o[3] = (float)c.m.pointCount;
      if (c.m.pointCount > 0) {
        o[2] = (float)c.m.type;
        o[4] = c.m.localNormal.x; o[5] = c.m.localNormal.y; o[6] = c.m.localPoint.x; o[7] = c.m.localPoint.y;
        for (int p = 0; p < c.m.pointCount; ++p) {
          const ManifoldPoint& mp = c.m.points[p];
          float* q = o + 8 + 4 * p;
          q[0] = mp.localPoint.x; q[1] = mp.localPoint.y; q[2] = mp.normalImpulse; q[3] = mp.tangentImpulse;
          o[16 + p] = (float)(mp.id.cf.indexA + 16 * mp.id.cf.indexB + 256 * mp.id.cf.typeA + 512 * mp.id.cf.typeB);
        }
      }
    }
}

// body-origin transforms (xf.p, xf.q.s, xf.q.c) and world-frame polygon vertices b2Mul(xf, v) — what lcd_render reads
// (world_env.py:479,501-502); out: [nb,4] and [nb, 1 + 2*kMaxPolygonVertices] (count, then x,y pairs; count 0 = circle, then radius)
void b2o_body_xf(b2o_env* e, float* xf_out, float* verts_out) {
  const b2o_scene_desc& S = e->scene;
  const int VS = 1 + 2 * kMaxPolygonVertices;
  for (int i = 0; i < S.n_bodies; ++i) {
    const Body& b = e->world.bodies[4 + i];
    xf_out[4 * i] = b.xf.p.x; xf_out[4 * i + 1] = b.xf.p.y; xf_out[4 * i + 2] = b.xf.q.s; xf_out[4 * i + 3] = b.xf.q.c;
    float* v = verts_out + (size_t)i * VS;
    if (b.shape.type == kCircle) { v[0] = 0.0f; v[1] = b.shape.radius; continue; }
    v[0] = (float)b.shape.count;
    for (int k = 0; k < b.shape.count; ++k) {
      Vec2 p = Mul(b.xf, b.shape.v[k]);
      v[1 + 2 * k] = p.x; v[2 + 2 * k] = p.y;
    }
  }
}

// diagnostic: move one state scalar of one body by `ulps` float32 steps (0:c.x 1:c.y 2:a 3:v.x 4:v.y 5:w); used by
// tools/nudge_search.py to localise WHEN a replay departs from a recording.
void b2o_nudge(b2o_env* e, int32_t body, int32_t field, int32_t ulps) {
  Body& b = e->world.bodies[4 + body];
  float* f = field == 0 ? &b.sweep.c.x : field == 1 ? &b.sweep.c.y : field == 2 ? &b.sweep.a : field == 3 ? &b.v.x : field == 4 ? &b.v.y : &b.w;
  for (int i = 0; i < (ulps < 0 ? -ulps : ulps); ++i) *f = nextafterf(*f, ulps > 0 ? INFINITY : -INFINITY);
  if (field < 3) { b.sweep.c0 = b.sweep.c; b.sweep.a0 = b.sweep.a; b.SynchronizeTransform(); }
}

// variant switches of b2o_math.h (process-wide; tests that change them restore the defaults)
void b2o_set_variant(int32_t key, int32_t value) { if (key >= 0 && key < kNumVariants) g_variant[key] = value; }
int32_t b2o_get_variant(int32_t key) { return key >= 0 && key < kNumVariants ? g_variant[key] : -1; }

void b2o_track_sweeps(b2o_env* e, int32_t on) { e->world.stats.trackSweeps = on != 0; }
int32_t b2o_solve_log(b2o_env* e, int32_t* out, int32_t cap) {   // copies and clears; returns the number of ints
  auto& v = e->world.stats.solveLog; int n = (int)v.size() < cap ? (int)v.size() : cap; for (int i = 0; i < n; ++i) out[i] = v[i]; v.clear(); return n; }
void b2o_nic_hist(b2o_env* e, int64_t* out16) { for (int i = 0; i < 16; ++i) out16[i] = e->world.stats.nicHist[i]; }
void b2o_nic_hist_free(b2o_env* e, int64_t* out16) { for (int i = 0; i < 16; ++i) out16[i] = e->world.stats.nicHistFree[i]; }
int32_t b2o_last_solve_sweeps(b2o_env* e) { int v = e->world.stats.lastSolveSweeps; e->world.stats.lastSolveSweeps = 0; return v; }   // read-and-reset
void b2o_sweep_hist(b2o_env* e, int64_t* out182) {
  for (int i = 0; i < 182; ++i) out182[i] = e->world.stats.sweepHist[i];
}
void b2o_period_hist(b2o_env* e, int64_t* out36) {
  for (int i = 0; i < 34; ++i) out36[i] = e->world.stats.periodHist[i];
  out36[34] = e->world.stats.cycleAtSum;
  out36[35] = e->world.stats.cycleCount;
}
void b2o_pos_fix_hist(b2o_env* e, int64_t* out64) {
  for (int i = 0; i < 64; ++i) out64[i] = e->world.stats.posFixHist[i];
}
int64_t b2o_toi_iters(b2o_env* e) { return e->world.stats.toiIters; }
void b2o_pos_iter_hist(b2o_env* e, int64_t* out62) {
  for (int i = 0; i < 62; ++i) out62[i] = e->world.stats.posIterHist[i];
}
void b2o_stats(b2o_env* e, int64_t* out6) {
  const Stats& s = e->world.stats;
  out6[0] = s.steps; out6[1] = s.toiEvents; out6[2] = s.toiCalls; out6[3] = s.islands; out6[4] = s.contactsCreated;
  out6[5] = s.contactsDestroyed;
}

int32_t b2o_contact_order(b2o_env* e, int32_t* out, int32_t cap) {
  int n = 0;
  for (int cid : e->world.contactList) {
    const Contact& c = e->world.contacts[cid];
    int a = std::min(c.bodyA, c.bodyB), b = std::max(c.bodyA, c.bodyB);
    for (size_t k = 0; k < e->pairs.size(); ++k)
      if (e->pairs[k].first == a && e->pairs[k].second == b && n < cap) out[n++] = (int32_t)k;
  }
  return n;
}

void b2o_raster_polygon(const int32_t* xy, int32_t count, int32_t w, int32_t h, int32_t variant, uint8_t* img) {
  Canvas cv{w, h, img};
  draw_polygon(cv, xy, count, variant);
}
void b2o_raster_ellipse(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t w, int32_t h, uint8_t* img) {
  Canvas cv{w, h, img};
  draw_ellipse(cv, x0, y0, x1, y1);
}
void b2o_sincos(const float* x, int64_t n, float* s, float* c) {
  for (int64_t i = 0; i < n; ++i) b2o_sincosf(x[i], &s[i], &c[i]);
}
void b2o_mass_data(const b2o_scene_desc* scene, int32_t shape, float density, float* out) {
  Shape s = BuildShape(scene->shapes[shape]);
  MassData md;
  ShapeComputeMass(&s, &md, density);
  out[0] = md.mass; out[1] = md.center.x; out[2] = md.center.y; out[3] = md.I; out[4] = (float)s.count;
  for (int i = 0; i < s.count && i < 8; ++i) {
    out[5 + 2 * i] = s.v[i].x;
    out[6 + 2 * i] = s.v[i].y;
  }
}

// Narrow phase by itself, for tests/test_oracle_narrowphase.py (geometry-based properties of the manifolds: an algorithmic check
// that does not depend on this restatement being compared with a copy of itself).  Shape = {kind 0 circle r | 1 box hx hy | 2 edge
// x1 y1 x2 y2 | 3 polygon n x0 y0 ...}, pose = {x, y, angle}; dispatch as b2Contact's s_registers (A = the lower shape type).
// out = {pointCount, type, localNormal.xy, localPoint.xy, [lp.xy, id.key] x2, world normal.xy, [world point.xy, separation] x2, swapped}
static Shape ShapeFromSpec(const float* sp) {
  Shape s;
  int kind = (int)sp[0];
  if (kind == 0) ShapeSetCircle(&s, sp[1]);
  else if (kind == 1) ShapeSetAsBox(&s, sp[1], sp[2]);
  else if (kind == 2) ShapeSetEdge(&s, V2(sp[1], sp[2]), V2(sp[3], sp[4]));
  else {
    Vec2 vs[16];
    int n = (int)sp[1];
    for (int i = 0; i < n && i < 16; ++i) vs[i] = V2(sp[2 + 2 * i], sp[3 + 2 * i]);
    ShapeSetPolygon(&s, vs, n);
  }
  return s;
}
int32_t b2o_collide(const float* specA, const float* poseA, const float* specB, const float* poseB, float* out) {
  Shape a = ShapeFromSpec(specA), b = ShapeFromSpec(specB);
  Transform xa, xb;
  xa.p = V2(poseA[0], poseA[1]); xa.q.Set(poseA[2]);
  xb.p = V2(poseB[0], poseB[1]); xb.q.Set(poseB[2]);
  // b2Contact::s_registers: the fixture with the "smaller" collision routine row comes first; (edge, circle), (edge, polygon),
  // (polygon, circle) keep the non-circle / edge shape as A
  int swapped = 0;
  auto rank = [](const Shape& s) { return s.type == kEdge ? 0 : (s.type == kPolygon ? 1 : 2); };
  if (rank(a) > rank(b)) { std::swap(a, b); std::swap(xa, xb); swapped = 1; }
  Manifold m;
  m.pointCount = 0;
  m.type = 0;
  if (a.type == kCircle && b.type == kCircle) CollideCircles(&m, &a, xa, &b, xb);
  else if (a.type == kPolygon && b.type == kCircle) CollidePolygonAndCircle(&m, &a, xa, &b, xb);
  else if (a.type == kPolygon && b.type == kPolygon) CollidePolygons(&m, &a, xa, &b, xb);
  else if (a.type == kEdge && b.type == kCircle) CollideEdgeAndCircle(&m, &a, xa, &b, xb);
  else if (a.type == kEdge && b.type == kPolygon) CollideEdgeAndPolygon(&m, &a, xa, &b, xb);
  else return -1;
  for (int i = 0; i < 24; ++i) out[i] = 0.0f;
  out[0] = (float)m.pointCount;
  out[1] = (float)m.type;
  out[2] = m.localNormal.x; out[3] = m.localNormal.y; out[4] = m.localPoint.x; out[5] = m.localPoint.y;
  for (int j = 0; j < m.pointCount && j < 2; ++j) {
    out[6 + 3 * j] = m.points[j].localPoint.x; out[7 + 3 * j] = m.points[j].localPoint.y; out[8 + 3 * j] = (float)m.points[j].id.key;
  }
  if (m.pointCount > 0) {
    WorldManifold wm;
    wm.Initialize(&m, xa, a.radius, xb, b.radius);
    out[12] = wm.normal.x; out[13] = wm.normal.y;
    for (int j = 0; j < m.pointCount && j < 2; ++j) {
      out[14 + 3 * j] = wm.points[j].x; out[15 + 3 * j] = wm.points[j].y; out[16 + 3 * j] = wm.separations[j];
    }
  }
  out[20] = (float)swapped;
  return m.pointCount;
}

double b2o_rollout(const b2o_scene_desc* scene, int32_t n, int32_t T, int32_t threads, const float* poses,
                   const int32_t* shape_sel, const float* actions, float* obs_out, uint8_t* lcd_out, float* state_out,
                   int32_t render_every_step) {
  const b2o_scene_desc& S = *scene;
  if (threads < 1) threads = 1;
  auto work = [&](int lo, int hi) {
    std::vector<double> obs(S.n_obs > 0 ? S.n_obs : 1);
    std::vector<uint8_t> lcd((size_t)S.lcd_w * S.lcd_h);
    std::vector<float> zero(S.n_act > 0 ? S.n_act : 1, 0.0f);
    for (int k = lo; k < hi; ++k) {
      b2o_env* e = b2o_create(scene);
      b2o_reset(e, poses + (size_t)k * S.n_bodies * 3, shape_sel ? shape_sel + (size_t)k * S.n_bodies : nullptr);
      for (int t = 0; t < T; ++t) {
        const float* a = actions ? actions + ((size_t)t * n + k) * S.n_act : zero.data();
        b2o_env_step(e, a);
        if (render_every_step) {
          b2o_get_obs(e, obs.data());
          b2o_render(e, lcd.data());
        }
      }
      if (obs_out) {
        b2o_get_obs(e, obs.data());
        for (int i = 0; i < S.n_obs; ++i) obs_out[(size_t)k * S.n_obs + i] = (float)obs[i];
      }
      if (lcd_out) b2o_render(e, lcd_out + (size_t)k * S.lcd_w * S.lcd_h);
      if (state_out) b2o_dump(e, state_out + (size_t)k * S.n_bodies * B2O_BODY_STATE_FLOATS, nullptr, nullptr);
      b2o_destroy(e);
    }
  };
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int i = 0; i < threads; ++i) {
    int lo = (int)((int64_t)n * i / threads), hi = (int)((int64_t)n * (i + 1) / threads);
    th.emplace_back(work, lo, hi);
  }
  for (auto& t : th) t.join();
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

// The same rollout, keeping what _get_obs / lcd_render return after EVERY env-step (world_env.py:458): obs_out float32
// [T][n][n_obs], lcd_out uint8 [T][n][h][w] - the layout of the product's blcd_rollout outputs - plus the final state.
double b2o_rollout_frames(const b2o_scene_desc* scene, int32_t n, int32_t T, int32_t threads, const float* poses,
                          const int32_t* shape_sel, const float* actions, float* obs_out, uint8_t* lcd_out, float* state_out) {
  const b2o_scene_desc& S = *scene;
  if (threads < 1) threads = 1;
  const size_t px = (size_t)S.lcd_w * S.lcd_h;
  auto work = [&](int lo, int hi) {
    std::vector<double> obs(S.n_obs > 0 ? S.n_obs : 1);
    std::vector<float> zero(S.n_act > 0 ? S.n_act : 1, 0.0f);
    for (int k = lo; k < hi; ++k) {
      b2o_env* e = b2o_create(scene);
      b2o_reset(e, poses + (size_t)k * S.n_bodies * 3, shape_sel ? shape_sel + (size_t)k * S.n_bodies : nullptr);
      for (int t = 0; t < T; ++t) {
        b2o_env_step(e, actions ? actions + ((size_t)t * n + k) * S.n_act : zero.data());
        if (obs_out) {
          b2o_get_obs(e, obs.data());
          for (int i = 0; i < S.n_obs; ++i) obs_out[((size_t)t * n + k) * S.n_obs + i] = (float)obs[i];
        }
        if (lcd_out) b2o_render(e, lcd_out + ((size_t)t * n + k) * px);
      }
      if (state_out) b2o_dump(e, state_out + (size_t)k * S.n_bodies * B2O_BODY_STATE_FLOATS, nullptr, nullptr);
      b2o_destroy(e);
    }
  };
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int i = 0; i < threads; ++i) th.emplace_back(work, (int)((int64_t)n * i / threads), (int)((int64_t)n * (i + 1) / threads));
  for (auto& t : th) t.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
