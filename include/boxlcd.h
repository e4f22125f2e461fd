/* boxlcd.h — C ABI of the MI355X-native batched boxLCD hot path (libboxlcd_hip.so).
 *
 * The reference (matwilso/boxLCD) has no FFI layer: its hot path is Python calling two un-vendored native
 * libraries through their Python bindings.  Each entry point below therefore cites the reference *call site*
 * it replaces (paths relative to the reference root); INTEGRATION.md shows the ctypes stub a boxLCD maintainer
 * would add.
 *
 * Conventions: every function returns 0 on success and a negative blcd_status on failure (message via
 * blcd_last_error(), thread-local).  The caller owns every buffer.  Buffers marked "host|device" may be plain
 * host memory or HIP device memory on the handle's device (detected with hipPointerGetAttributes); device buffers
 * are used in place (zero copy) on the handle's stream.  A handle is bound to one device; all handles of a device share ONE HIP stream
 * (a second hardware queue doubles the scratch reservation of the large scene classes, see blcd_api.hip), so calls on
 * different handles serialise on the device.  A handle is
 * not thread-safe; different handles may be driven from different threads.  No exceptions cross the ABI.
 * There is NO CPU fallback: blcd_create fails if no HIP device is usable.
 */
#ifndef BOXLCD_H
#define BOXLCD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BLCD_VERSION 101 /* 0.1.1: state snapshots carry the reset sampler's counters; blcd_build_features, blcd_sample_set_base, blcd_get_shape_sel */

#define BLCD_MAX_POLY_VERTS 8
#define BLCD_MAX_BODIES 20
#define BLCD_MAX_JOINTS 20
#define BLCD_MAX_SHAPES 24
#define BLCD_MAX_OBS 96

typedef enum blcd_status {
  BLCD_OK = 0,
  BLCD_ERR_INVALID = -1,     /* bad argument / scene outside supported limits */
  BLCD_ERR_NO_DEVICE = -2,   /* no usable HIP device (there is no CPU path) */
  BLCD_ERR_HIP = -3,         /* a HIP runtime call failed */
  BLCD_ERR_UNSUPPORTED = -4, /* scene needs a feature this build lacks */
  BLCD_ERR_ENV_FAULT = -5    /* a per-env guard tripped on device (NaN state, ellipse outside LUT, pair overflow) */
} blcd_status;

/* ---- scene description: the data form of WorldEnv.__init__ + _reset_bodies (boxLCD/world_env.py:47-142,197-304),
 *      world_defs.py:11-59 and envs.py:17-137.  Filled by boxlcd_amd/scene.py. ---- */
typedef struct blcd_shape_def {
  int32_t type;    /* 0 circleShape(radius), 1 polygonShape(...)  (world_env.py:273, world_defs.py:82-83,103-108) */
  int32_t n_verts; /* polygonShape(vertices=...): count, goes through the b2PolygonShape::Set hull */
  float radius;
  int32_t is_box;  /* polygonShape(box=(hx,hy)): verts[0] = (hx, hy) */
  float verts[BLCD_MAX_POLY_VERTS][2];
} blcd_shape_def;

typedef struct blcd_body_def {
  int32_t n_choices; /* 2 for Object(shape='random') (world_env.py:273-274), else 1 */
  int32_t shape[2];
  float density, friction, restitution;
  uint32_t category_bits, mask_bits;
  float linear_damping, angular_damping;
  int32_t kind; /* 0 object, 1 robot root, 2 robot link */
  int32_t _pad;
} blcd_body_def;

typedef struct blcd_joint_def { /* revoluteJointDef(...) at world_env.py:255-266 */
  int32_t body_a, body_b;
  float anchor_a[2], anchor_b[2];
  int32_t enable_limit;
  float lower, upper;
  float max_motor_torque;
  float speed;          /* Joint.speed, the action scale at world_env.py:441 */
  int32_t action_index; /* -1: joint is not actuated (world_env.py:438) */
} blcd_joint_def;

/* one entry per sorted observation key (world_env.py:120-122).
 * kind: 0 x:p  1 y:p  2 cos(body.angle)  3 sin(body.angle)  4 cos(transform.angle)  5 sin(transform.angle) */
typedef struct blcd_obs_def {
  int32_t kind, body;
  float lo, hi;
} blcd_obs_def;

typedef struct blcd_scene_desc {
  int32_t n_bodies, n_joints, n_shapes, n_obs, n_act;
  int32_t lcd_w, lcd_h;   /* int(lcd_base*wh_ratio), lcd_base (world_env.py:467-469) */
  int32_t raster_variant; /* Pillow polygon scan rule: 0 = 9.0.x (the reference's pin), 1 = >= 12 (corner joining),
                             2 = 8.2-8.4 (what the reference's demo GIFs were recorded with); SURVEY.md App. C.4b */
  float world_w, world_h; /* WIDTH = int(wh_ratio*base_dim), HEIGHT (world_env.py:144-150) */
  float gravity[2];
  float dt;               /* float32(1/(fps*3)) (world_env.py:448) */
  int32_t substeps, vel_iters, pos_iters; /* 3, 180, 60 */
  blcd_shape_def shapes[BLCD_MAX_SHAPES];
  blcd_body_def bodies[BLCD_MAX_BODIES];
  blcd_joint_def joints[BLCD_MAX_JOINTS];
  blcd_obs_def obs[BLCD_MAX_OBS];
} blcd_scene_desc;

typedef struct blcd_handle_s* blcd_handle;

/* library */
int blcd_version(void);
/* Optional parts compiled into this library (BLCD_DEFS at build time): bit 0 per-wave timers (blcd_debug_wave_times),
 * bit 1 the environment-level schedulers of DESIGN.md 4.4 (BLCD_ASYNC / BLCD_WAVE_BATCH / BLCD_YIELD_PASSES, blcd_sched_stats).
 * Neither is in the default build; blcd_create fails with BLCD_ERR_UNSUPPORTED when a knob asks for a part that is absent. */
#define BLCD_FEATURE_WAVETIMES 1
#define BLCD_FEATURE_SCHED 2
int blcd_build_features(void);
const char* blcd_last_error(void);
int blcd_device_count(void);

/* Replaces N x `envs.<Name>(G)` construction + `Box2D.b2World(gravity=...)` (world_env.py:47-67; the vector form
 * research/wrappers/async_vector_env.py:98-109 forks one process per env instead). */
int blcd_create(const blcd_scene_desc* scene, int32_t n_envs, int32_t device, blcd_handle* out);
int blcd_destroy(blcd_handle h);
int blcd_num_envs(blcd_handle h);

/* Static contact-pair slots (proxy ids: 0..3 = walls bottom,left,right,top; 4+i = body i) in (A,B)-sorted order:
 * the pairs b2ContactManager::AddPair can ever accept for this scene. */
int blcd_num_pairs(blcd_handle h);
int blcd_pair_table(blcd_handle h, int32_t* pairs /* host [n_pairs][2] */);

/* Replaces WorldEnv.reset()'s world construction (world_env.py:306-317 + _reset_bodies :197-304) for the envs in
 * `idxs` (NULL = all, then n must equal n_envs).  poses = position/angle handed to CreateDynamicBody for every body,
 * sampled on the host by boxlcd_amd/world_env.py exactly as the reference samples them (float64 numpy).
 * idxs host; poses host|device float32 [n][n_bodies][3]; shape_sel host|device int32 [n][n_bodies] or NULL. */
int blcd_reset(blcd_handle h, const int32_t* idxs, int32_t n, const float* poses, const int32_t* shape_sel);

/* Device-side `_reset_bodies` sampling (world_env.py:197-304): reset n environments from a counter-based random stream instead
 * of host-sampled poses, so that a training loop that resets environments never touches the host.
 * Stream: Philox4x32-10, key = seed (lo, hi), counter = (env id, that env's reset count since blcd_sample_reseed, variable
 * index, 0); u = ((x0 >> 5) * 2^26 + (x1 >> 6)) * 2^-53; draw = lr + (ur - lr) * u; value = mapto(draw, (lo, hi)) - all float64,
 * the reference's expressions (utils.py:117) - so an environment's start depends on (seed, env id, reset count) only, never on
 * the batch size or on which rank holds it.  `ops` is the scene's sampling program in the reference's draw order (the Python
 * shim builds it from the world definition; boxlcd_amd/world_env.py mirrors it in numpy as the checker):
 *   kind 0 DRAW   r[d] = mapto(uniform(f[0], f[1]), (f[2], f[3]))
 *   kind 1 SEL    shape_sel[body] = uniform(0, 1) < 0.5 ? 0 : 1            ('random' object shapes)
 *   kind 2 ATAN2  r[d] = atan2(r[a], r[b])      kind 3 ZERO r[d] = 0
 *   kind 4 BODY   pose[body] = (float) (r[a], r[b], r[d]); remembers position (float32) and angle (float64) of `body`
 *   kind 5 LINK   child link `body` of `parent` on the robot rooted at body `a`: f = {joint angle, anchorA.xy, anchorB.xy}
 *                 (world_env.py:235-253: float64 angles, float32 b2Vec2 position arithmetic)
 * idxs host|device or NULL (= all, n == n_envs). */
typedef struct blcd_sample_op {
  int32_t kind, d, a, b, body, parent;
  double f[5];
} blcd_sample_op;
int blcd_reset_sampled(blcd_handle h, const int32_t* idxs, int32_t n, uint64_t seed, const blcd_sample_op* ops, int32_t n_ops);
int blcd_sample_reseed(blcd_handle h);   /* reset counts back to 0 (env.seed()) */
/* Sharded batches: the Philox counter's env id is env_id_base + (index inside this handle), so rank r of a batch cut into
 * contiguous shards of n environments passes r * n and - with the SAME seed on every rank - the gathered batch is the batch one
 * handle of world x n environments would have sampled (research/wrappers/async_vector_env.py has one seed per env process
 * instead).  Default 0. */
int blcd_sample_set_base(blcd_handle h, uint64_t env_id_base);
/* The shape each body currently has (Object(shape='random'): world_env.py:273-274 picks one per reset): int32
 * [n_envs][n_bodies], host|device.  What blcd_render_poses_ex needs as shape_sel to draw the batch's current states. */
int blcd_get_shape_sel(blcd_handle h, int32_t* shape_sel);

/* Replaces the `body.position = ...; body.angle = ...` overwrite of reset(full_state=/proprio=) (world_env.py:319-380):
 * b2Body::SetTransform semantics (velocities untouched, broad-phase proxy refreshed).  mask host uint8 [n_bodies] or NULL. */
int blcd_set_poses(blcd_handle h, const int32_t* idxs, int32_t n, const float* poses, const uint8_t* mask);

/* Replaces WorldEnv.step(action) (world_env.py:431-458) for all envs, n_steps times with the same actions:
 * action -> motorSpeed (utils.py:117 + world_env.py:441), then `substeps` x b2World.Step(dt, vel_iters, pos_iters).
 * actions host|device float32 [n_envs][n_act] (NULL = zeros).
 * Returns BLCD_ERR_ENV_FAULT (after completing the step) when any environment carries a fault flag: see blcd_get_faults. */
int blcd_step(blcd_handle h, const float* actions, int32_t n_steps);

/* blcd_step(actions, 1) + blcd_get_obs(float32) as ONE call with one stream synchronisation - `obs, rew, done, info =
 * venv.step(actions)` as the reference's policy loops issue it (research/rl/ppo.py:127-133, rl/sac.py:200-214;
 * async_vector_env.py:191-242 returns the observations with the step).  full_state float32 [n_envs][n_obs], lcd uint8
 * [n_envs][lcd_h][lcd_w]; either may be NULL; host|device.  Same results as the two calls (tests/test_gpu_api.py). */
int blcd_step_obs(blcd_handle h, const float* actions, float* full_state, uint8_t* lcd);
/* The same step WITHOUT the synchronisation, for consumers that stay on the device (a policy network between two steps): device
 * buffers only; the call returns when the work is queued on blcd_stream(h).  Ordering in both directions is the caller's - make
 * blcd_stream(h) wait for the producer of `actions` before the call and the consumer of the outputs wait for blcd_stream(h) after it
 * (hipEventRecord + hipStreamWaitEvent; the Python layer does both for torch: BatchedWorldEnv.step_torch(sync=False)).  A fault
 * raised by the step is not reported here: it stays readable through blcd_get_faults and is returned by the next synchronising
 * step / rollout call. */
int blcd_step_obs_async(blcd_handle h, const float* actions, float* full_state, uint8_t* lcd);
/* The stream blcd_step_obs_async queues on: adopt = 0 (default) = the handle's own stream; adopt = 1 = `stream`, a hipStream_t of the
 * caller (NULL = the device's default stream, which is what torch uses unless told otherwise): the step runs ON that stream, between the caller's kernels, with no hand-off between streams (a Bounce-100k step is 61 us of kernel; two cross-stream
 * waits cost half as much again).  Every other entry point keeps the handle's own stream and is ordered behind steps queued on the adopted
 * one.  Opt-in: each hardware queue reserves scratch for the largest kernel it has run, so adopt a stream for the small
 * scene classes; the caller keeps the stream alive until blcd_sync / blcd_destroy.  Drains pending asynchronous steps. */
int blcd_set_async_stream(blcd_handle h, void* stream, int32_t adopt);

/* Fused rollout, replaces the inner loop of research/data.py:56-61 (`for j in range(ep_len): venv.step(act)`):
 * T env-steps with per-step actions [T][n_envs][n_act]; per-step outputs (any may be NULL):
 * lcd_out uint8 [T][n_envs][lcd_h][lcd_w], obs_out float32 [T][n_envs][n_obs].  All host|device.
 * The library cuts the T steps into fused launches itself (joint-free scenes in batches larger than 64 environments per
 * SIMD: 20 env-steps with slot re-binning in between; everything else: up to 200, sized from the previous rollout's time
 * per step; BLCD_CHUNK=<n> pins it) - results do not depend on the cut. */
int blcd_rollout(blcd_handle h, const float* actions, int32_t T, uint8_t* lcd_out, float* obs_out);
/* The same rollout with the frames at ONE BIT per pixel - what `lcd` is in the reference (a bool array, world_env.py:508-509)
 * and what north_star calls the "1-bit framebuffer": lcd_bits_out uint8 [T][n_envs][lcd_h][lcd_w / 8], pixel x of a row =
 * bit (x % 8) of byte x / 8 (numpy.unpackbits(..., bitorder='little') gives blcd_rollout's bytes), 1 = background.
 * 8x less store traffic for the tensor that is ~94 % of a chunk's bytes.  lcd_w must be a multiple of 8 (every catalogue env). */
int blcd_rollout_bits(blcd_handle h, const float* actions, int32_t T, uint8_t* lcd_bits_out, float* obs_out);

/* Replaces WorldEnv._get_obs() (world_env.py:387-429) incl. lcd_render() (:460-512).
 * full_state: normalised [n_envs][n_obs], dtype 0 = float32 (vector-env layout, async_vector_env.py:374-378),
 * 1 = float64 (single-env layout, world_env.py:388,427); lcd uint8 [n_envs][lcd_h][lcd_w], 1 = background, 0 = body.
 * Either pointer may be NULL.  host|device. */
int blcd_get_obs(blcd_handle h, void* full_state, int32_t dtype, uint8_t* lcd);

/* Goal-conditioned reward / done epilogue, evaluated on device from the handle's current state.
 * Replaces BodyGoalEnv.comp_rew_done (research/wrappers/body_goal.py:58-88) and CubeGoalEnv.comp_rew_done
 * (research/wrappers/cube_goal.py:64-86), which the reference runs per process on host arrays:
 *   mode 0 (state):  delta = mean_k |goal_full_state[idx_k] - full_state[idx_k]|   (float64, numpy's summation order)
 *                    rew = diff_delt ? -0.05 + 10 (last_delta - delta) : -delta;  delta < thresh -> rew += 1, done
 *   mode 1 (LCD):    similarity = mean(lcd == 0 & lcd == goal_lcd) / mean(lcd == 0);  rew = -1 + similarity;
 *                    similarity > thresh -> rew = 0, done;  delta := similarity
 *   finally rew *= rew_scale.  last_delta is the previous evaluation's delta (the reference's `last_obs`), kept per env. */
typedef struct blcd_goal_desc {
  int32_t mode;      /* 0 state distance, 1 LCD similarity */
  int32_t diff_delt; /* mode 0 only */
  int32_t n_idx;     /* mode 0: number of full_state columns compared (BodyGoalEnv: '.*(x|y):p' of proprio; CubeGoalEnv:
                        'object.*(x|y):p') */
  int32_t idxs[BLCD_MAX_OBS];
  double thresh;     /* G.goal_thresh (body_goal.py:75), 0.05 (cube_goal.py:80) or 0.70 (body_goal.py:84) */
  double rew_scale;  /* G.rew_scale (body_goal.py:98, cube_goal.py:59) */
} blcd_goal_desc;

/* Installs the reward definition and the goals of n environments (idxs NULL = environments 0..n-1).
 * goal_full_state float64 [n][n_obs] (the goal observation's 'full_state'), goal_lcd uint8 [n][lcd_h][lcd_w] or NULL
 * (needed for mode 1).  host|device. */
int blcd_goal_set(blcd_handle h, const blcd_goal_desc* g, const int32_t* idxs, int32_t n, const double* goal_full_state,
                  const uint8_t* goal_lcd);
/* After a reset: last_delta := delta of the current state for the listed environments (idxs NULL = all). */
int blcd_goal_seed(blcd_handle h, const int32_t* idxs, int32_t n);
/* After a step: rew float64 [n_envs], done uint8 [n_envs] (the wrapper's own `_done`; the caller ORs the time limit),
 * delta float64 [n_envs] (info['delta']); any may be NULL; updates last_delta.  host|device. */
int blcd_goal_eval(blcd_handle h, double* rew, uint8_t* done, double* delta);

/* Replaces the `env.reset(proprio=s)['lcd']` state->LCD use (research/nets/autoencoders/_base.py:69,76):
 * renders m pose sets without touching env state.  poses float32 [m][n_bodies][3] (x, y, angle of each body's
 * transform), shape_sel int32 [m][n_bodies] or NULL; lcd uint8 [m][lcd_h][lcd_w].  host|device. */
int blcd_render_poses(blcd_handle h, const float* poses, const int32_t* shape_sel, int32_t m, uint8_t* lcd);

/* Replaces `WorldEnv.lcd_render(width, height, lcd_mode)` (boxLCD/world_env.py:460-512) for explicit canvas sizes and for
 * lcd_mode='RGB' - i.e. also the 8x view `render(mode='human')` composes (world_env.py:514-535).  mode 0 = '1': out uint8
 * [m][height][width], 1 = background (fill only); mode 1 = 'RGB': out uint8 [m][height][width][3], body.color1 fill + 1-px
 * body.color2 outline, already `255 - image` as the reference returns it.  Both axes scale by width / WIDTH, as upstream.
 * Circles come from Pillow's ellipse span table, which the caller hands over once per process with
 * blcd_set_ellipse_rgb_lut (file boxlcd_amd/ellipse_rgb_lut.bin: uint8 [amax+1][5][amax+3][6], amax = 100).  host|device. */
int blcd_set_ellipse_rgb_lut(const uint8_t* lut /* host */, int32_t amax);
int blcd_render_poses_ex(blcd_handle h, const float* poses, const int32_t* shape_sel, int32_t m, int32_t width, int32_t height,
                         int32_t mode, uint8_t* out);

/* Transport form of LCD frames (the reference's frames are bool arrays, world_env.py:506-509): n_pixels 0/1
 * bytes <-> n_pixels/8 bytes, bit k of byte i = pixel 8i+k.  Device pointers, asynchronous on `stream` (a hipStream_t; NULL =
 * the default stream); n_pixels a multiple of 8, the byte-per-pixel buffer 8-byte aligned.  Used by the multi-GPU gather
 * (boxlcd_amd/dist.py) so that rollouts cross xGMI at one bit per pixel and are delivered as uint8 again. */
int blcd_pack_bits(const uint8_t* src, uint8_t* dst, int64_t n_pixels, void* stream);
int blcd_unpack_bits(const uint8_t* src, uint8_t* dst, int64_t n_pixels, void* stream);

/* Body poses for host-side consumers: float32 [n_envs][n_bodies][4] = transform.position.x, .y, body.angle, awake. */
int blcd_get_poses(blcd_handle h, float* poses);

/* Full snapshot (positions, velocities, impulses, contacts, sleep state, and the reset sampler's per-environment reset counts
 * + env-id base, so a resumed run draws the same reset stream) — the reference never checkpoints env state
 * (SURVEY.md §5); needed for mid-rollout parity tests and exact resume.  *size in/out (query with blob == NULL).
 * blcd_set_state refuses snapshots written by another BLCD_VERSION or another build's state layout. */
int blcd_get_state(blcd_handle h, void* blob, size_t* size);
int blcd_set_state(blcd_handle h, const void* blob, size_t size);

/* Per-env fault flags raised on device since the last reset (0 = healthy): int32 [n_envs], host.
 * 1 non-finite state, 2 circle larger than the ellipse table, 4 more simultaneous contacts than the scene class holds. */
int blcd_get_faults(blcd_handle h, int32_t* flags);

/* Blocks until the handle's stream is idle; blcd_stream returns the hipStream_t so that a caller can order work against it.
 * Ordering contract for DEVICE buffers: the library launches on its own non-blocking stream and every entry point returns after
 * synchronising that stream, so results are complete on return.  The other direction is the caller's: a device buffer handed to a
 * call must be complete - and an output buffer no longer in use - on the caller's side, either by synchronising the producing stream
 * or, without a host synchronisation, by making blcd_stream(h) wait for it (hipEventRecord on the producer's stream +
 * hipStreamWaitEvent(blcd_stream(h), event)) before the call.  The shipped Python layer does the latter for torch tensors
 * (boxlcd_amd/_lib.py Handle._after_torch; tests/test_gpu_api.py::test_device_tensors_are_ordered_behind_torchs_stream). */
int blcd_sync(blcd_handle h);
void* blcd_stream(blcd_handle h);

/* Kernel-only timing of the last blcd_step/blcd_rollout launch sequence, measured with hipEvents on the handle's
 * stream (ms); used by bench.py's roofline block. */
int blcd_last_kernel_ms(blcd_handle h, float* ms, int32_t* launches);

/* ---- parity-test hooks (sub-step granularity + canonical dump; same layout as the parity oracle's dump) ---- */
#define BLCD_BODY_STATE_FLOATS 12 /* cx cy a vx vy w sleepTime awake fat.lo.x fat.lo.y fat.hi.x fat.hi.y */
#define BLCD_JOINT_STATE_FLOATS 5 /* impulse.x impulse.y impulse.z motorImpulse limitState */
#define BLCD_PAIR_STATE_FLOATS 18 /* exists touching type pointCount ln.xy lp.xy {p.xy ni ti}x2 id0 id1 */
/* Environment-level scheduling of fused chunks (joint-free scene classes; no reference counterpart - it only decides WHEN an
 * environment's world step runs, never its result): counters since the last call, then reset.
 * out8 = { first passes: lanes entered live, lanes that left suspended, waves, passes per chunk;
 *          later passes: lanes entered live, lanes that left suspended, waves, lane threshold }. */
int blcd_sched_stats(blcd_handle h, uint64_t* out8);
int blcd_debug_wave_times(blcd_handle h, uint64_t* out, int32_t cap);      /* per-wave ticks (100 MHz) of the last step launch; needs a -DBLCD_WAVETIMES build and BLCD_WAVETIMES=1 */
int blcd_debug_world_step(blcd_handle h, int32_t n_world_steps);           /* n x b2World::Step only */
int blcd_debug_set_motor_speeds(blcd_handle h, const float* actions);       /* the action half of step() only */
int blcd_debug_dump(blcd_handle h, float* bodies, float* joints, float* pairs); /* host [n_envs][..][FLOATS] */
int blcd_debug_sincos(const float* x, int64_t n, float* s, float* c, int32_t device); /* device sincosf on host arrays */
int blcd_debug_mass_data(const blcd_scene_desc* scene, int32_t shape, float density, float* out16);
/* The device narrow phase by itself (b2CollideCircles / b2CollidePolygonAndCircle / b2CollidePolygons and the wall forms of
 * b2CollideEdgeAndCircle / b2EPCollider), n independent configurations, host arrays.  spec[n][BLCD_COLLIDE_SPEC_FLOATS] =
 * {0, r} circle | {1, hx, hy} box | {2, x1, y1, x2, y2} edge (its pose must be 0, 0, 0) | {3, k, x0, y0, ...} polygon (k <= 8);
 * pose[n][3] = x, y, angle; out[n][24] = pointCount, type, localNormal.xy, localPoint.xy, {localPoint.xy, id.key} x 2,
 * world normal.xy, {world point.xy, -} x 2, swapped (A and B exchanged as b2Contact::Create does).  Test hook. */
#define BLCD_COLLIDE_SPEC_FLOATS 18
int blcd_debug_collide(int32_t device, int32_t n, const float* specA, const float* poseA, const float* specB, const float* poseB, float* out);

#ifdef __cplusplus
}
#endif
#endif /* BOXLCD_H */
