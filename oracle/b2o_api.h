/* ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under boxlcd_amd/ may include, link or call this.
 *
 * C interface of the CPU oracle (one environment per handle, scalar float32, Box2D-2.3.x restatement).
 * Callers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The scene description has the same
 * memory layout as include/boxlcd.h's blcd_scene_desc so that the Python scene compiler fills one ctypes struct.
 */
#ifndef B2O_API_H
#define B2O_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define B2O_MAX_POLY_VERTS 8
#define B2O_MAX_BODIES 20
#define B2O_MAX_JOINTS 20
#define B2O_MAX_SHAPES 24
#define B2O_MAX_OBS 96

typedef struct b2o_shape_def {
  int32_t type;    /* 0 circle, 1 polygon */
  int32_t n_verts; /* polygon: vertices as given to polygonShape(vertices=...) / box corners from SetAsBox */
  float radius;    /* circle radius */
  int32_t is_box;  /* polygon created with polygonShape(box=(hx,hy)): verts[0] = (hx,hy) */
  float verts[B2O_MAX_POLY_VERTS][2];
} b2o_shape_def;

typedef struct b2o_body_def {
  int32_t n_choices; /* 1, or 2 for boxLCD's shape='random' objects (world_env.py:273-274) */
  int32_t shape[2];  /* indices into shapes[] */
  float density, friction, restitution;
  uint32_t category_bits, mask_bits;
  float linear_damping, angular_damping;
  int32_t kind; /* 0 object, 1 robot root, 2 robot link (informational) */
  int32_t _pad;
} b2o_body_def;

typedef struct b2o_joint_def {
  int32_t body_a, body_b; /* indices into bodies[] */
  float anchor_a[2], anchor_b[2];
  int32_t enable_limit;
  float lower, upper;
  float max_motor_torque;
  float speed;          /* Joint.speed: action scale (world_defs.py:40) */
  int32_t action_index; /* index into the action vector, -1 if limits[0]==limits[1] (world_env.py:438) */
} b2o_joint_def;

/* kind: 0 x:p  1 y:p  2 cos(body.angle)  3 sin(body.angle)  4 cos(transform.angle)  5 sin(transform.angle) */
typedef struct b2o_obs_def {
  int32_t kind, body;
  float lo, hi;
} b2o_obs_def;

typedef struct b2o_scene_desc {
  int32_t n_bodies, n_joints, n_shapes, n_obs, n_act;
  int32_t lcd_w, lcd_h, raster_variant; /* 0 legacy, 1 modern, 2 recording era */
  float world_w, world_h;               /* WIDTH = int(wh_ratio*base_dim), HEIGHT = base_dim */
  float gravity[2];
  float dt;                             /* float32(1/(fps*3)) */
  int32_t substeps, vel_iters, pos_iters;
  b2o_shape_def shapes[B2O_MAX_SHAPES];
  b2o_body_def bodies[B2O_MAX_BODIES];
  b2o_joint_def joints[B2O_MAX_JOINTS];
  b2o_obs_def obs[B2O_MAX_OBS];
} b2o_scene_desc;

/* Canonical per-env dump used by the parity tests (same field order as blcd_debug_dump in include/boxlcd.h). */
#define B2O_BODY_STATE_FLOATS 12 /* cx cy a vx vy w sleepTime awake fat.lo.x fat.lo.y fat.hi.x fat.hi.y */
#define B2O_JOINT_STATE_FLOATS 5 /* impulse.x impulse.y impulse.z motorImpulse limitState */
#define B2O_PAIR_STATE_FLOATS 18 /* exists touching type pointCount ln.x ln.y lp.x lp.y {p.x p.y ni ti}x2 id0 id1 ; id = iA + 16 iB + 256 tA + 512 tB */

typedef struct b2o_env b2o_env;

b2o_env* b2o_create(const b2o_scene_desc* scene);
void b2o_destroy(b2o_env* e);
/* poses: [n_bodies][3] = x, y, angle of each body definition (as CreateDynamicBody(position, angle)); shape_sel: [n_bodies] */
void b2o_reset(b2o_env* e, const float* poses, const int32_t* shape_sel);
/* overwrite poses through b2Body::SetTransform as reset(full_state=) does (world_env.py:333-380); mask[i]!=0 selects bodies */
void b2o_set_poses(b2o_env* e, const float* poses, const uint8_t* mask);
void b2o_env_step(b2o_env* e, const float* action);        /* WorldEnv.step: motor speeds + `substeps` world steps */
void b2o_world_step(b2o_env* e);                           /* one b2World::Step(dt, vel_iters, pos_iters) */
void b2o_set_motor_speeds(b2o_env* e, const float* action); /* the action -> motorSpeed half of step() only */
void b2o_get_obs(b2o_env* e, double* full_state);          /* normalised float64 [n_obs] as _get_obs() */
void b2o_render(b2o_env* e, uint8_t* lcd);                 /* [lcd_h][lcd_w] 1 = background 0 = body, after FLIP_TOP_BOTTOM */
int32_t b2o_num_pairs(b2o_env* e);                         /* canonical pair-slot count (static filter) */
void b2o_pair_table(b2o_env* e, int32_t* pairs);           /* [n][2] proxy ids (0..3 walls, 4.. bodies) */
void b2o_dump(b2o_env* e, float* bodies, float* joints, float* pairs);
void b2o_stats(b2o_env* e, int64_t* out6);
void b2o_track_sweeps(b2o_env* e, int32_t on);   /* diagnostic: histogram of the sweep at which velocities reach a fixed point */
void b2o_sweep_hist(b2o_env* e, int64_t* out182);
void b2o_period_hist(b2o_env* e, int64_t* out36);
int32_t b2o_contact_order(b2o_env* e, int32_t* out_pairs, int32_t cap); /* world contact list, newest first, as pair-slot ids */

/* stateless helpers */
void b2o_render_poses(const b2o_scene_desc* scene, const float* poses, const int32_t* shape_sel, int32_t n, uint8_t* lcd);
void b2o_raster_polygon(const int32_t* xy, int32_t count, int32_t w, int32_t h, int32_t variant, uint8_t* img);
void b2o_raster_ellipse(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t w, int32_t h, uint8_t* img);
void b2o_sincos(const float* x, int64_t n, float* s, float* c);
void b2o_mass_data(const b2o_scene_desc* scene, int32_t shape, float density, float* out /* mass cx cy I n v.. */);
/* narrow phase by itself (tests/test_oracle_narrowphase.py): shape spec {kind, params..}, pose {x, y, angle}; out float[24], returns pointCount */
int32_t b2o_collide(const float* specA, const float* poseA, const float* specB, const float* poseB, float* out);

/* CPU-baseline rollout: n envs x T env-steps, envs statically partitioned over `threads`; returns seconds.
 * poses [n][nb][3], shape_sel [n][nb], actions [T][n][n_act] (may be NULL -> zeros); outputs (may be NULL):
 * obs_out float32 [n][n_obs] (final), lcd_out [n][h][w] (final), state_out [n][nb][12] (final). */
double b2o_rollout(const b2o_scene_desc* scene, int32_t n, int32_t T, int32_t threads, const float* poses,
                   const int32_t* shape_sel, const float* actions, float* obs_out, uint8_t* lcd_out, float* state_out,
                   int32_t render_every_step);
/* the same, keeping every env-step's observation and frame: obs_out [T][n][n_obs], lcd_out [T][n][h][w] */
double b2o_rollout_frames(const b2o_scene_desc* scene, int32_t n, int32_t T, int32_t threads, const float* poses,
                          const int32_t* shape_sel, const float* actions, float* obs_out, uint8_t* lcd_out, float* state_out);

#ifdef __cplusplus
}
#endif
#endif
