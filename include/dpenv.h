/*
 * dpenv.h - C ABI of libdpenv.so: the MI355X-native batched ReVolt dynamic-positioning
 * environment (env.step hot path of simensov/ml4ca as one HIP kernel for gfx950).
 *
 * This is the drop-in boundary.  The reference is pure Python with no FFI of its own; the
 * interface this library replaces is (paths relative to the reference root,
 * WW = src/rl/windows_workspace):
 *   - upper boundary, what the PPO loop calls: Revolt.reset / Revolt.step
 *       WW/specific/customEnv.py:135-194, :92-133  (consumed at WW/spinup/algos/tf1/ppo/ppo.py:286-322)
 *   - lower boundary, the plant plug-in seam it swallows: DigiTwin.val / DigiTwin.step
 *       WW/specific/digitwin.py:50-114, :213-219   (py4j RPC into the Cybersea simulator)
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - Plain C: opaque handle, raw pointers and sizes, int status codes.  No torch types.
 *   - Every bulk-data pointer is a DEVICE pointer owned by the caller (e.g. tensor.data_ptr());
 *     config structs and vessel parameter vectors are HOST memory.  A NULL optional pointer
 *     means "not requested".
 *   - The library owns the per-env state block in HBM behind the handle.  No allocation,
 *     no host synchronisation inside reset/step: calls are ordered on the hipStream_t passed in
 *     and are graph-capturable.
 *   - One handle per (process, GPU).  A handle is not thread-safe; different handles are independent.
 *   - Thruster order everywhere: 0 = bow (THR1), 1 = stern port (THR2), 2 = stern starboard (THR3)
 *     (customEnv.py:48-50).  The ROS/QP code uses port, star, bow (qp_allocator.py:69-70).
 *   - Return value: DPENV_OK or a negative DPENV_E*; dpenv_last_error() gives the message.
 *     There is NO CPU fallback: without a usable gfx950 device dpenv_create fails with DPENV_ENODEV.
 */
#ifndef DPENV_H
#define DPENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPENV_ABI_VERSION 5

typedef struct dpenv_s* dpenv_handle;
typedef void* dpenv_stream; /* hipStream_t; NULL = the null stream */

enum {
    DPENV_OK = 0,
    DPENV_EINVAL = -1,  /* bad argument / unsupported combination */
    DPENV_ENODEV = -2,  /* no usable HIP device */
    DPENV_ENOMEM = -3,  /* device allocation failed */
    DPENV_EHIP = -4,    /* HIP runtime error (launch, copy) */
};

/* env variants, customEnv.py:11 (Revolt, name 'full'), :327 RevoltSimple, :351 RevoltLimited, :373 RevoltFinal */
enum { DPENV_FULL = 0, DPENV_SIMPLE = 1, DPENV_LIMITED = 2, DPENV_FINAL = 3 };
/* memory layout of action / observation batches */
enum { DPENV_AOS = 0 /* [n_envs][dim] row-major (torch-native) */, DPENV_SOA = 1 /* [dim][n_envs] */ };
/* yaw wrap in the error frame: REFERENCE replicates errorFrame.py:29,31 calling wrap_angle with its
 * default deg=True on radians (mathematics.py:14); RADIANS wraps to [-pi,pi) like the ROS node. */
enum { DPENV_WRAP_REFERENCE = 0, DPENV_WRAP_RADIANS = 1 };
enum { DPENV_F32 = 0, DPENV_BF16 = 1 };
/* bits of the per-env done byte */
enum { DPENV_DONE_TERMINAL = 1 /* is_terminal, customEnv.py:207-213 */,
       DPENV_DONE_TIMELIMIT = 2 /* traj_len == max_ep_len, ppo.py:304 */,
       DPENV_DONE_FAULT = 4 /* non-finite state or action (also sets TERMINAL) */ };

/* canonical state exchange format for get/set_state: float state[DPENV_NSTATE][n_envs] */
enum {
    DPENV_S_N = 0, DPENV_S_E, DPENV_S_PSI, DPENV_S_U, DPENV_S_V, DPENV_S_R,
    DPENV_S_REF_N, DPENV_S_REF_E, DPENV_S_REF_PSI,
    DPENV_S_PT_BOW, DPENV_S_PT_PORT, DPENV_S_PT_STAR, /* previous thrust command, percent (customEnv.py:126) */
    DPENV_S_A_BOW, DPENV_S_A_PORT, DPENV_S_A_STAR,    /* azimuth command in force, rad (customEnv.py:122) */
    DPENV_NSTATE
};
/* int32 counters[2][n_envs]: [0] steps taken in the running episode, [1] episodes sampled so far */

/* vessel parameter vector: float[DPENV_NPARAM] per vessel class.  Hull terms are BUILD-OWNED (the
 * reference's plant is the closed Cybersea simulator); thruster terms come from the reference
 * (K: qp_allocator.py:51-55 / SupervisedTau.py:69-71, lever arms: qp_allocator.py:69-70). */
enum {
    DPENV_P_M11 = 0, DPENV_P_M22, DPENV_P_M23, DPENV_P_M33, /* rigid-body + added mass, symmetric */
    DPENV_P_XU, DPENV_P_XUU, DPENV_P_YV, DPENV_P_YVV, DPENV_P_YR, DPENV_P_NV, DPENV_P_NR, DPENV_P_NRR, /* damping >= 0 */
    DPENV_P_KF_BOW, DPENV_P_KF_PORT, DPENV_P_KF_STAR, /* F = K n|n|, n >= 0 */
    DPENV_P_KR_BOW, DPENV_P_KR_PORT, DPENV_P_KR_STAR, /* n < 0 */
    DPENV_P_LX_BOW, DPENV_P_LX_PORT, DPENV_P_LX_STAR,
    DPENV_P_LY_BOW, DPENV_P_LY_PORT, DPENV_P_LY_STAR,
    DPENV_P_NUV, DPENV_P_YUR, /* speed-proportional cross-flow terms: yaw moment -N_uv u v (adds to the Munk moment -(m22-m11) u v; N_uv < -(m22-m11) would make the hull weathervane-stable), sway force -Y_ur u r */
    /* BUILD-OWNED inflow thrust loss (the linear open-water characteristic, Fossen 2011 eq. 9.7): F = K n|n| - Kl |n| u_a, u_a = the speed through
     * the water of the thruster's position along its axis at the start of the env step, never past zero thrust; [N per percent per m/s], >= 0.
     * 0 (the default hull) is the reference's law F = K n|n| (SupervisedTau.py:42-83) exactly.  Non-zero coefficients are carried by the SINGLE
     * class of dpenv_create (kernel arguments: the shared training form, the default's cost) or by per-env blocks (the general per-env kernels);
     * vessel CLASSES (n_classes > 1) may not have them. */
    DPENV_P_KLF_BOW, DPENV_P_KLF_PORT, DPENV_P_KLF_STAR, /* n >= 0 */
    DPENV_P_KLR_BOW, DPENV_P_KLR_PORT, DPENV_P_KLR_STAR, /* n < 0 */
    DPENV_NPARAM = 32,
    DPENV_NPARAM_USED = 32
};
#define DPENV_MAX_CLASSES 64   /* classes share LDS-staged tables; for more distinct hulls than that - one per env - see dpenv_set_vessel_params */

typedef struct dpenv_config {
    uint32_t struct_size;    /* sizeof(dpenv_config), ABI check */
    int32_t n_envs;
    int32_t device;          /* HIP device ordinal, -1 = current */
    int32_t variant;         /* DPENV_FULL .. DPENV_FINAL */
    int32_t extended_state;  /* obs dim 9 (1) or 6 (0), customEnv.py:44,201-205 */
    int32_t cont_ang;        /* FINAL only: 7 actions with sin/cos azimuth heads, customEnv.py:227-235 */
    int32_t n_substeps;      /* plant sub-steps per env step, 20 (customEnv.py:79-80) */
    float substep_dt;        /* 0.01 s (customEnv.py:81) */
    int32_t wrap_mode;       /* DPENV_WRAP_* */
    int32_t terminate;       /* 1: evaluate is_terminal bounds; 0: never terminal */
    int32_t max_ep_len;      /* time limit in env steps (train.py:70-73 -> 400); 0 = none */
    int32_t auto_reset;      /* re-sample finished envs inside step (ppo.py:305-322 batched) */
    int32_t action_layout;   /* DPENV_AOS / DPENV_SOA */
    int32_t obs_layout;
    int32_t obs_dtype;       /* DPENV_F32 / DPENV_BF16 */
    int32_t current_enabled; /* per-env constant irrotational current, see dpenv_set_current */
    uint64_t seed;           /* Philox key of the reset sampler */
    int64_t env_id_base;     /* global id of local env 0: results do not depend on the rank count */
    float reset_fraction;    /* 0.8 (customEnv.py:135; curriculum hook ppo.py:286,319) */
    int32_t hold_plant;      /* 1: hull state is held while the step runs, like Hull.StateResetOn=1
                                (customEnv.py:164-167); lets parity tests replay the reference's scripted-plant
                                fixtures through the kernel.  0 in production. */
    int32_t current_drift;   /* 1: V_c and beta_c follow a first-order Gauss-Markov process around the values given to
                                dpenv_set_current (config 5's slowly varying disturbance; build-defined) */
    float current_tau;       /* correlation time [s], 100 */
    float current_sigma_v;   /* stationary std of V_c [m/s], 0.02 */
    float current_sigma_beta;/* stationary std of beta_c [rad], 5 deg */
    int32_t reset_acts;      /* 1: an episode starts with previous thrust clip(100 * N(0, 0.1)) instead of zero (the reference's
                                reset_acts constructor flag, customEnv.py:30,179-188); drawn in the kernel by every kind of reset,
                                Philox keyed (seed; global env id, episode) like the pose sample */
    int32_t step_one_wave;   /* 0 (default): dpenv_step with auto_reset on launches a second wave per 64 envs that prepares the re-draw of
                                finished envs beside the plant loop (up to 256 envs per CU), and dpenv_rollout runs an env wave and a row
                                wave per 64 envs (up to 384 envs per CU) - same rows bit for bit, the one-wave kernels above those sizes
                                (DESIGN.md section 4).  1: the one-wave kernels at every size - the A/B switch of tools/ and tests;
                                was `reserved` (0) before round 4 */
    int32_t per_env_lds;     /* per-env parameter blocks in dpenv_step: 0 (default) a lane loads its block straight into registers; 1: by
                                LDS-DMA (global_load_lds_dwordx4) into a [group][lane] LDS image that the step reads back - the A/B SURVEY
                                section 7 asks for ("LDS [matrix_elem][lane] vs plain VGPRs"); same rows bit for bit, slower (DESIGN.md
                                section 4).  The T-step kernels load the block once per launch into registers either way. */
} dpenv_config;

/* Optional outputs / inputs of one step beyond the Gym tuple.  All device pointers, any may be NULL. */
typedef struct dpenv_step_io {
    uint32_t struct_size;
    const float* action;     /* [n][act_dim] or [act_dim][n] per action_layout */
    const float* new_ref;    /* [3][n]: applied AFTER obs/reward/done of this step (customEnv.py:131) */
    void* obs;               /* [n][obs_dim] or [obs_dim][n], f32 or bf16 */
    float* reward;           /* [n] */
    uint8_t* done;           /* [n], DPENV_DONE_* bits */
    float* reward_parts;     /* [4][n]: vel, pose gaussian, thrust penalty, derivative penalty */
    void* final_obs;         /* same layout as obs: terminal observation of envs that auto-reset */
} dpenv_step_io;

/* Fill *cfg with the shipped training configuration (RevoltFinal, extended state, continuous
 * angles, 20 x 0.01 s, T = 400, train.py:47-54).  n_envs is left 0. */
int dpenv_default_config(dpenv_config* cfg);
/* Default ReVolt parameter vector. */
int dpenv_default_vessel(float params[DPENV_NPARAM]);
/* The presets of the build-owned plant, one per set of steady full-thrust speeds the reference records (customEnv.py:13-18):
 * NO_LOSS = dpenv_default_vessel (+2.20 m/s ahead, 0.60 rad/s: "no thrust losses activated"); THRUST_LOSS = the same hull with stern
 * thrusters that meet BOTH sets: their reverse gain from -1.60 m/s astern without losses, their inflow-loss coefficients (DPENV_P_KLF_* /
 * DPENV_P_KLR_*) from +1.4 / -1.1 m/s ahead / astern "with thrust losses" (the velocity bounds the reference trains with, customEnv.py:26);
 * yaw then comes out at 0.505 rad/s against the recorded 0.52 (tests/calibration/fit_thrust_loss_preset.py).  Pass the vector to
 * dpenv_create (one class), in a dpenv_set_vessel_params block, or as `nominal` to dpenv_set_vessel_randomisation. */
enum { DPENV_VESSEL_NO_LOSS = 0, DPENV_VESSEL_THRUST_LOSS = 1,
       /* round 6, a FLAG (combine with THRUST_LOSS: 3): the sway-yaw part of the hull (m22, Yv, Yvv, Yr, Nv, Nr, Nrr, Yur) refitted jointly to
        * what the default hull is fitted to AND to the reference's 32 recorded Cybersea station-keeping runs in a current from 16 directions
        * (results/all_plots/dyn_pos/) AND to the recorded steady sway speed: sway 0.350 m/s (recorded 0.35; default hull 0.29), yaw 0.605, the
        * station-keeping yaw-moment residual halved, at 0.01-0.02 m on the free-drift / box-test / replay rows
        * (tests/calibration/fit_dynpos_preset.py).  Not the default: every row the default hull has produced stays. */
       DPENV_VESSEL_DYNPOS_FIT = 2 };
int dpenv_default_vessel_ex(int32_t kind, float params[DPENV_NPARAM]);
/* Derived sizes for a config. */
int dpenv_act_dim(const dpenv_config* cfg);
int dpenv_obs_dim(const dpenv_config* cfg);

/* Create an environment batch.  vessel_params: host float[n_classes][DPENV_NPARAM], NULL = one
 * default class.  With n_classes > 1 call dpenv_set_vessel_class to assign envs to classes.
 * A single class with thrust-loss coefficients (dpenv_default_vessel_ex(DPENV_VESSEL_THRUST_LOSS, ...)) runs on kernels that take hull AND
 * coefficients from their arguments - dpenv_step, dpenv_rollout, and dpenv_policy_rollout's two-wave form in the shipped configuration (final
 * variant, continuous angles, extended state, leaky-relu): the traffic of the default hull; the other closed-loop forms read that hull from a
 * per-env image (every env the same block).  dpenv_get_vessel_params works; dpenv_set_vessel_params(h, NULL, s) returns to this class WITH
 * its loss.  Classes (n_classes > 1) with coefficients are refused. */
int dpenv_create(const dpenv_config* cfg, const float* vessel_params, int32_t n_classes, dpenv_handle* out);
int dpenv_destroy(dpenv_handle h);
/* Message of the last failure on this handle (h == NULL: last failure of dpenv_create in this thread). */
const char* dpenv_last_error(dpenv_handle h);

/* Change the pose/velocity fraction of the training reset sampler (curriculum hook, ppo.py:286,319-322). */
int dpenv_set_reset_fraction(dpenv_handle h, float fraction);
/* class_id: device int32[n_envs], values in [0, n_classes).  Copied. */
int dpenv_set_vessel_class(dpenv_handle h, const int32_t* class_id, dpenv_stream s);

/* ---- per-env vessel parameter blocks (north_star: "per-env 3x3 mass / Coriolis / damping blocks"; SURVEY appendix D: "M, D as per-env
 * SoA parameter arrays (domain randomisation) with a shared-default fast path") -----------------------------------------------------
 * Every env gets its OWN hull and thruster parameters - the constants the reference hard-codes once for its one vessel
 * (qp_allocator.py:51-55,69-70 K and lever arms, SupervisedTau.py:35-36,69-71) and the build-owned mass / damping terms of the plant
 * that stands in for customEnv.py:124.  params: DEVICE float[DPENV_NPARAM][n_envs] (structure of arrays: row p = parameter DPENV_P_p of
 * every env), copied on the stream: one kernel derives each env's mass-matrix inverse (the same float operations
 * as for a class, so an env given its class's numbers reproduces the class path bit for bit) and packs the block as eight float4
 * streams (+ two for the thrust-loss coefficients).  dpenv_step then reads 128 B more per env-step (SURVEY 8d accounts 108 B: 285 B per
 * env-step); the T-step kernels (dpenv_rollout, dpenv_policy_rollout) load the block once per launch.  A block that is not a vessel (mass
 * matrix not positive definite, non-finite entry, negative loss coefficient) is not rejected here: that env reports DPENV_DONE_FAULT at
 * its first step.  If ANY env has a thrust-loss coefficient the general per-env kernels apply it (they read 32 B more per env-step; envs
 * without a coefficient get the rows of the plain per-env kernels bit for bit).  Whether any env has one is a word the packing kernel leaves
 * behind the table: the setter does NOT wait for it - it is stream-ordered and may be recorded into a HIP graph.  Outside a capture the word
 * travels to the host behind an event and the next launch on this handle takes it from there (waiting for that one event if need be); a
 * launch that is itself being recorded, and every launch after a RECORDED setter, runs the general kernels with the coefficients applied -
 * zeros where there are none, which change no row.  A refused call (bad flags, a HIP error) leaves the handle's switches - per-env blocks,
 * randomisation, thrust loss - as they were.
 * params == NULL: back to the vessel classes / the single class of dpenv_create (the shared-default fast path: parameters in SGPRs).
 * Switching between the paths voids HIP graphs captured before (the table's address is a kernel argument).
 * flags (dpenv_set_vessel_params_ex): DPENV_VESSEL_KEEP_RANDOMISATION - the table is installed while the domain randomisation STAYS in force
 * (dpenv_set_vessel_randomisation must have been called): the restore path of a checkpoint taken mid-episode - fresh handle, same config;
 * dpenv_set_vessel_randomisation(nominal, range); dpenv_set_vessel_params_ex(saved table, KEEP); dpenv_set_state(saved state, counters) - the
 * run continues bit for bit, hulls re-drawn at every later reset.  dpenv_set_vessel_params(h, p, s) = _ex(h, p, 0, s): ends the re-draws. */
enum { DPENV_VESSEL_KEEP_RANDOMISATION = 1 };
int dpenv_set_vessel_params_ex(dpenv_handle h, const float* params, uint32_t flags, dpenv_stream s);
int dpenv_set_vessel_params(dpenv_handle h, const float* params, dpenv_stream s);
/* The parameter vectors in force: DEVICE float[DPENV_NPARAM][n_envs] - the per-env blocks, or, with ONE class and none in force, that class's
 * vector in every column.  Refused with vessel classes (n_classes > 1) and no per-env blocks. */
int dpenv_get_vessel_params(dpenv_handle h, float* params_out, dpenv_stream s);
/* Domain randomisation through the reset path: from this call on EVERY reset of an env - dpenv_reset (also with explicit init),
 * auto-reset inside dpenv_step / dpenv_rollout / dpenv_policy_rollout, reset_at_end - starts the new episode on a freshly drawn hull:
 *   parameter p = nominal[p] * (1 + rel_range[p] * u),  u uniform in [-1, 1) (16 bits),
 * Philox4x32-10 keyed by config.seed with counter (global env id, episode counter, tag 0x48000000 | block) - parameter p takes the
 * 16-bit half (q & 1) of word (q & 7) >> 1 of block q >> 3, q = its slot in the order m11 m22 m23 m33 Xu (0..4) | Xuu Yv Yvv Yr Nv Nr Nrr
 * Nuv (8..15) | Yur Kf[3] Kr[3] lx_bow (16..23) | lx_port lx_star ly[3] Klf[3] (24..31), Klr[3] (5..7); u = h / 32768 - 1: a function of the env and of
 * its episode like the pose sample, so hulls do not depend on the rank count or on the launch form, and a checkpoint (dpenv_get_state
 * counters + dpenv_get_vessel_params) restores them through dpenv_set_vessel_params_ex(..., DPENV_VESSEL_KEEP_RANDOMISATION).  nominal: HOST float[DPENV_NPARAM], NULL = class 0 of dpenv_create; rel_range:
 * HOST float[DPENV_NPARAM], entries in [0, 1), 0 = that parameter is not randomised; every hull of the range must have a positive
 * definite mass matrix (checked).  Until its first reset an env runs on the nominal hull.  Implies per-env blocks;
 * rel_range == NULL stops the re-draws (the hulls in force stay); dpenv_set_vessel_params(h, NULL / table, s) ends it as well.
 * With the randomisation on, a dpenv_reset with explicit init advances the episode counter too (it consumes random numbers). */
int dpenv_set_vessel_randomisation(dpenv_handle h, const float* nominal, const float* rel_range, dpenv_stream s);
/* vc, beta: device float[n_envs] current speed [m/s] and NED direction [rad].  Copied; they are both the
 * present value and the mean the drift process reverts to. */
int dpenv_set_current(dpenv_handle h, const float* vc, const float* beta, dpenv_stream s);
/* Only the PRESENT values (the drift's state), leaving the means alone: restores what dpenv_get_current returned (checkpoints). */
int dpenv_set_current_present(dpenv_handle h, const float* vc, const float* beta, dpenv_stream s);
/* present current of every env (differs from the set values only with current_drift / the per-episode randomisation) */
int dpenv_get_current(dpenv_handle h, float* vc_out, float* beta_out, dpenv_stream s);
/* the means the drift reverts to (= the values given to dpenv_set_current until a randomised reset re-draws them): with dpenv_get_current the
 * current's part of a checkpoint - restore with dpenv_set_current(means) followed by dpenv_set_current_present(present values) */
int dpenv_get_current_mean(dpenv_handle h, float* vc_out, float* beta_out, dpenv_stream s);
/* Per-episode randomisation of the current through the reset path (config 5 widened; the reference's one operating point is 0.2 m/s towards
 * 135 deg, results/all_plots/current_box_test/plot_pos.py:78): from this call on EVERY reset of an env - dpenv_reset (also with explicit
 * init), auto-reset inside dpenv_step / dpenv_rollout / dpenv_policy_rollout, reset_at_end - starts the new episode in a freshly drawn current
 *   V_c = max(0, vc_nominal[i] + vc_range * u1),   beta_c = beta_nominal[i] + beta_range * u2,   u1, u2 uniform in [-1, 1) (24 bits),
 * words 0 and 1 of Philox4x32-10 keyed by config.seed with counter (global env id, episode counter, tag 3): a function of the env and of
 * its episode like the pose sample and the hull draw - independent of the rank count and of the launch form.  The drawn values become the
 * present current AND the mean the drift (config.current_drift) reverts to.  vc_nominal, beta_nominal: DEVICE float[n_envs], copied; NULL =
 * the means in force (what dpenv_set_current gave; a checkpoint restore passes the ORIGINAL nominals explicitly: by then the means are drawn
 * values).  Until its first reset an env keeps the current it has.  Needs config.current_enabled.
 * The re-draw lives in the kernels that carry re-draws: with ONE class, the shared training form (hull and thrust-loss coefficients - zero for
 * a hull without a loss - as kernel arguments: the default's memory traffic, the default's rows until a reset draws); with per-env blocks in
 * force, the general per-env kernels; vessel classes (n_classes > 1 without per-env blocks) are refused, and so is returning to them with
 * dpenv_set_vessel_params(h, NULL, s) while it is on.  Both ranges 0: off (currents stay as they are).  Like the hull randomisation, a dpenv_reset
 * with explicit init then advances the episode counter too.  Stream-ordered, may be recorded into a graph. */
int dpenv_set_current_randomisation(dpenv_handle h, const float* vc_nominal, const float* beta_nominal, float vc_range, float beta_range,
                                    dpenv_stream s);

/* Revolt.reset (customEnv.py:135-194) for the envs selected by mask (device uint8[n], NULL = all).
 * init: device float[6][n] = N, E, psi, u, v, r (the **init override, customEnv.py:141,152), NULL =
 * training sample (simtools.py:109-123).  ref: device float[3][n] new setpoints, NULL = keep.
 * obs_out (optional) receives the observation of EVERY env. */
int dpenv_reset(dpenv_handle h, const uint8_t* mask, const float* init, const float* ref, void* obs_out,
                dpenv_stream s);

/* Revolt.step (customEnv.py:92-133) for all envs. */
int dpenv_step(dpenv_handle h, const float* action, const float* new_ref, void* obs_out, float* rew_out,
               uint8_t* done_out, dpenv_stream s);
int dpenv_step_ex(dpenv_handle h, const dpenv_step_io* io, dpenv_stream s);

/* Fused rollout: T env steps in ONE launch with the state resident in registers.  Exactly the semantics of T
 * successive dpenv_step calls with actions[t] as the action and, at t == switch_step[k], refs[k] as new_ref
 * (the setpoint-sequence form of test_policy.py:127,148-153 / results/all_plots/box_test/plot_pos.py:55-59).
 * obs[t] is the observation returned by step t.  Open loop: the action block must exist before the launch
 * (recorded command sequences, pre-sampled exploration noise, benchmark input). */
#define DPENV_MAX_SWITCH 8
typedef struct dpenv_rollout_io {
    uint32_t struct_size;
    int32_t T;
    const float* actions;    /* [T][n][act_dim] (AOS) or [T][act_dim][n] (SOA) */
    void* obs;               /* [T][n][obs_dim] or [T][obs_dim][n]; f32 or bf16 */
    float* reward;           /* [T][n] */
    uint8_t* done;           /* [T][n], DPENV_DONE_* bits */
    int32_t n_switch;        /* 0..DPENV_MAX_SWITCH, switch_step strictly increasing */
    int32_t switch_step[DPENV_MAX_SWITCH];
    const float* refs;       /* [n_switch][3][n] */
} dpenv_rollout_io;
int dpenv_rollout(dpenv_handle h, const dpenv_rollout_io* io, dpenv_stream s);

/* ---- actor-critic in the loop -------------------------------------------------
 * The PPO actor-critic of the reference (mlp_gaussian_policy / mlp_actor_critic, spinup/algos/tf1/ppo/core.py:29-33,
 * 80-107; shipped model 9-80-80-80-7 + 9-80-80-80-1, leaky_relu 0.2, config.json) evaluated inside the rollout
 * launch on the matrix cores (f16 weights/activations, f32 accumulate), so that one launch produces T rows of the
 * trajectory buffer (o, a, r, v, logp) of ppo.py:298 for every env. */
enum { DPENV_ACT_LEAKY_RELU = 0, DPENV_ACT_TANH = 1 };
typedef struct dpenv_mlp {
    int32_t n_layers;        /* dense layers = hidden layers + 1, in [2, 5] */
    int32_t sizes[6];        /* n_layers + 1 widths, e.g. {9, 80, 80, 80, 7}; hidden widths equal and <= 96 */
    const float* W[5];       /* W[l][in][out] row-major (tf.layers.dense kernel layout); host or device, see dpenv_policy_desc */
    const float* b[5];       /* b[l][out] */
} dpenv_mlp;
/* Arithmetic of the in-kernel networks.  F16: f16 weights and activations, f32 accumulation - the fast mode, within ~5e-4 of
 * the output scale of an fp32 evaluation.  F32: "fp32-faithful" split-f16 arithmetic (W = Wh + Wl, x = xh + xl, three MFMAs per
 * product, activations in f32): mu, v, logp within 1e-5 of an fp32 evaluation of core.py:29-33,80-107 - the mode parity with
 * the reference's fp32 TF1 networks is claimed on; three times the matrix work of F16.
 * F32_ACTOR: the actor (mu, and with it the sampled action and logp) in the F32 arithmetic, the critic in the F16 arithmetic: what a
 * PPO update needs exactly is the log-likelihood (the ratio exp(logp_new - logp_old) then starts at 1); values carry the F16 mode's
 * ~5e-4 and are bit-identical to the F16 mode's.  Two thirds of the matrix work of F32. */
enum { DPENV_POLICY_F16 = 0, DPENV_POLICY_F32 = 1, DPENV_POLICY_F32_ACTOR = 2 };
/* Launch form of dpenv_policy_rollout.  TWO_WAVE: every 64 envs get an env wave and a network wave (pair-level LDS hand-over;
 * 256-env workgroups with both waves of a pair on one SIMD, or - while one round of them fits the chip, n_envs <= 128 x CUs -
 * 128-env workgroups with a SIMD per wave); ONE_WAVE: one wave does both.  Both write identical rows.  AUTO picks TWO_WAVE where
 * it exists (every arithmetic with leaky-relu / relu, F16 also with tanh) and its LDS footprint (the weight images its network wave
 * reads + 40-50 KiB of mailboxes) fits the 160 KiB, else ONE_WAVE. */
enum { DPENV_LAUNCH_AUTO = 0, DPENV_LAUNCH_ONE_WAVE = 1, DPENV_LAUNCH_TWO_WAVE = 2 };
typedef struct dpenv_policy_desc {
    uint32_t struct_size;
    const dpenv_mlp* pi;     /* obs_dim -> act_dim */
    const dpenv_mlp* v;      /* obs_dim -> 1, same hidden shape */
    const float* log_std;    /* float[act_dim] (core.py:83) */
    int32_t activation;      /* DPENV_ACT_* */
    float leak;              /* leaky-relu slope in [0, 1] */
    int32_t precision;       /* DPENV_POLICY_* */
    int32_t launch_form;     /* DPENV_LAUNCH_* */
    int32_t device_pointers; /* 0: W, b, log_std are HOST pointers (copied on the stream, then packed on the device);
                                1: they are DEVICE pointers (e.g. the optimiser's own parameter tensors): packed by one kernel
                                on the stream - no host copy, no synchronisation, graph-capturable.  Re-upload after every
                                PPO update (ppo.py:260-280) costs one small launch. */
    int32_t reserved;
} dpenv_policy_desc;
/* Pack the networks into the image the rollout kernels stage into LDS.  Launches issued (on any stream) before this call keep
 * the weights they were given: the library holds two images and writes them alternately, and the packing waits (on `s`, by event)
 * for the last launch that read the image it reuses.  Launches issued afterwards read the new image; they are ordered behind the
 * packing if they are issued on `s` (or on a stream the caller orders behind `s`).  device_pointers = 1: stream-ordered, no host
 * synchronisation, graph-capturable.  device_pointers = 0 (host arrays): the call synchronises `s` before it returns, so the
 * arrays may be freed or changed at once.  Fails with DPENV_EINVAL if the requested launch form cannot hold the networks in the
 * 160 KiB LDS; after a failed DPENV_ENOMEM no policy is in force.
 * Graphs: a dpenv_policy_rollout / dpenv_policy_forward RECORDED INTO A HIP GRAPH has the address of the image that was current at
 * capture time baked into its kernel node.  From that capture on, every eager upload is written IN PLACE into that image (no more
 * alternation), so a replay always runs the weights of the latest upload - the usual PPO pattern "upload each epoch, replay the rollout
 * graph" works with any number of uploads between replays.  Ordering of an in-place upload: behind eager readers of the image by their
 * event (any stream), behind graph replays by STREAM ORDER - replay and upload on one stream, or order the two streams yourself.  An
 * upload recorded into the graph itself (device pointers) re-packs from the weight tensors at every replay.  Growing the network shape
 * (a larger image) voids graphs captured before. */
int dpenv_set_policy_desc(dpenv_handle h, const dpenv_policy_desc* d, dpenv_stream s);
/* What DPENV_LAUNCH_AUTO resolved to for the policy in force: *two_wave_out = 1 for the two-wave form, *envs_per_workgroup_out = 256 or
 * 128 (host ints, either may be NULL). */
int dpenv_get_policy_launch(dpenv_handle h, int32_t* two_wave_out, int32_t* envs_per_workgroup_out);
/* The same and more, as the library resolved it: out[0] two-wave form (0 / 1), out[1] envs per workgroup, out[2] waves per 64 envs (1 one-wave
 * form, 2 env + network wave, 3 env + actor + critic wave), out[3] the arithmetic (DPENV_POLICY_*). */
int dpenv_get_policy_launch_ex(dpenv_handle h, int32_t out[4]);
/* Ends the pinning described above: call it when every HIP graph that recorded a dpenv_policy_rollout / dpenv_policy_forward of this handle
 * has been destroyed (or will not be replayed again).  Until then an upload whose image LAYOUT differs from the one the graphs were captured
 * with - another hidden shape, precision, activation, leak or launch form; the kernel nodes hold those by value next to the image's address -
 * is refused with DPENV_EINVAL (it would be read with the old layout: wrong weights, no error); new weights of the same layout are what the
 * in-place upload is for.  After the call uploads alternate between the two images again and any layout is accepted. */
int dpenv_release_policy_graphs(dpenv_handle h);
/* Convenience forms (host pointers, F16, AUTO, null stream):
 * pi: obs_dim -> act_dim, v: obs_dim -> 1 (same hidden shape); log_std: host float[act_dim]; leak: hidden
 * leaky-relu slope (0.2 = tf.nn.leaky_relu default; 0 = relu).  Packs and uploads; may be called again after
 * every PPO update. */
int dpenv_set_policy(dpenv_handle h, const dpenv_mlp* pi, const dpenv_mlp* v, const float* log_std, float leak);
/* The same with the hidden activation named: the reference's --activation {leaky, relu, tanh} (train.py:24,31;
 * spinup core.py:29-33 takes any activation, tanh being Spinning Up's default).  relu = DPENV_ACT_LEAKY_RELU with
 * leak 0; leak is ignored for DPENV_ACT_TANH. */
int dpenv_set_policy_ex(dpenv_handle h, const dpenv_mlp* pi, const dpenv_mlp* v, const float* log_std, int32_t activation,
                        float leak);
/* mu_out [n][act_dim], v_out [n] for obs [n][obs_dim] (all device, row-major): the deterministic policy of
 * test_policy.py:90 and the critic. */
int dpenv_policy_forward(dpenv_handle h, const float* obs, float* mu_out, float* v_out, int32_t n, dpenv_stream s);

typedef struct dpenv_policy_rollout_io {
    uint32_t struct_size;
    int32_t T;
    const float* noise;      /* [T][n][act_dim] N(0,1) draws (a = mu + exp(log_std) * noise, core.py:85); NULL: see `sample` */
    void* obs;               /* [T][n][obs_dim]  policy input of step t; f32 or bf16 per config.obs_dtype (the actor
                                always sees the full-precision observation, only the stored row is rounded) */
    float* act;              /* [T][n][act_dim] */
    float* reward;           /* [T][n] */
    float* value;            /* [T][n]  V(obs[t]) */
    float* logp;             /* [T][n]  log-likelihood of act[t] (core.py:42-46) */
    uint8_t* done;           /* [T][n]  DPENV_DONE_* bits */
    float* boot;             /* [T][n]  value appended at a path end (ppo.py:311): 0 if terminal, V(next obs) if only the
                                time limit or the end of the launch cut the path; 0 elsewhere.  Feed to dpenv_gae. */
    void* last_obs;          /* [n][obs_dim] policy input of the next launch; f32 or bf16 */
    float* last_value;       /* [n] */
    int32_t n_switch;
    int32_t switch_step[DPENV_MAX_SWITCH];
    const float* refs;       /* [n_switch][3][n] */
    int32_t sample;          /* with noise == NULL: 0 = deterministic policy a = mu (test_policy.py:90); 1 = the exploration noise is
                                drawn INSIDE the kernel like the reference's tf.random_normal (core.py:85): Philox4x32-10 +
                                Box-Muller keyed (config.seed; global env id, number of actions that env has sampled so far), so a
                                trajectory does not depend on the rank count or launch geometry and no [T][n][act_dim] noise block
                                is generated, stored or read (28 B per env-step).  Ignored when noise != NULL. */
    int32_t reset_at_end;    /* 1: the reference's epoch boundary (ppo.py:305-322, `t == local_steps_per_epoch - 1`): after step T-1
                                EVERY env is cut - boot[T-1] = V(its last observation), or 0 if it terminated at that step
                                (ppo.py:311) - and re-drawn with the training sampler (episode counter + 1, step counter 0), so the
                                next launch starts T-step-aligned fresh episodes like the reference's next epoch; last_obs /
                                last_value are those of the NEW episodes.  Needs config.auto_reset.  0: episodes continue across
                                launches (the block still ends with a bootstrap value, like a cut-off path). */
} dpenv_policy_rollout_io;
/* Requires AOS layouts.  Vessel classes, drifting current, auto-reset (with reset_acts) and bf16 observation rows all work
 * here as in dpenv_step.  Launch form and arithmetic: dpenv_policy_desc. */
int dpenv_policy_rollout(dpenv_handle h, const dpenv_policy_rollout_io* io, dpenv_stream s);

/* Parity/test access to the library-owned state in the canonical format above. */
int dpenv_get_state(dpenv_handle h, float* state_out, int32_t* counters_out, dpenv_stream s);
int dpenv_set_state(dpenv_handle h, const float* state_in, const int32_t* counters_in, dpenv_stream s);
/* The two per-env draw counters that key the streams drawn every env step: exploration noise (sample = 1 rollouts) and the
 * Gauss-Markov current drift; device uint32[n_envs] each, either pointer may be NULL.  get_state + get_current + these =
 * everything a checkpoint needs: restoring them reproduces the sampled rollouts that followed the checkpoint, bit for bit. */
int dpenv_get_rng_counters(dpenv_handle h, uint32_t* noise_ctr_out, uint32_t* drift_ctr_out, dpenv_stream s);
int dpenv_set_rng_counters(dpenv_handle h, const uint32_t* noise_ctr_in, const uint32_t* drift_ctr_in, dpenv_stream s);
/* The observation of step t carries the thrust command of step t-1 (customEnv.py:196-205 fills state_ext before :126 updates
 * prev_thrust); the state block holds the command of step t.  A closed-loop launch that CONTINUES an episode therefore starts from
 * the observation its predecessor ended with: the library keeps that observation's thrust columns (device float[n_envs][4]: o[6], o[7],
 * o[8], unused).  Every call that changes the state keeps them (round 4): dpenv_policy_rollout and dpenv_rollout leave the columns of the
 * last observation they returned, dpenv_set_state those of an observation rebuilt from the state (previous thrust / 100), dpenv_reset
 * those of the envs it re-draws (a masked reset leaves the other envs' columns alone), and dpenv_step those of the observation it returns -
 * but only while a policy is in force (16 bytes per env-step that the plain step path does not pay): after a dpenv_step WITHOUT a policy
 * the columns are stale, and a closed-loop launch that follows an upload rebuilds its first observation from the state block (its thrust
 * columns are then the command of the last step, not of the one before).  get fails with DPENV_EINVAL while they are stale; set is for
 * restoring a mid-episode checkpoint (after dpenv_set_state).  Whether a launch continues or rebuilds is decided on the host when
 * dpenv_policy_rollout is CALLED: inside a captured graph the first closed-loop launch keeps the decision made at capture time on
 * every replay (capture a graph that starts with a continuing launch after one such launch has run).  Likewise a dpenv_step RECORDED INTO A
 * GRAPH before the first policy upload has "no policy in force" baked in (it does not write the columns): re-capture step graphs after the
 * first upload if closed-loop launches are to continue from their observations. */
int dpenv_get_obs_thrust(dpenv_handle h, float* out, dpenv_stream s);
int dpenv_set_obs_thrust(dpenv_handle h, const float* in, dpenv_stream s);

/* Stateless thruster force map tau = B(alpha) F(n) (SupervisedTau.py:42-83) for n items:
 * n_pct, alpha, tau_out are device float[3][n] (bow, port, star / Fx, Fy, Mz); params host float[DPENV_NPARAM]. */
int dpenv_thrust_map(const float* params, const float* n_pct, const float* alpha, float* tau_out, int32_t n,
                     dpenv_stream s);

/* GAE-lambda over a [T][n] rollout (TrajectoryBuffer.finish_path, ppo.py:65-91, batched): a path ends
 * after step t of env i where end[t][i] != 0 and always after T-1.  Bootstrap value at a path end:
 * boot[t][i] if boot != NULL, else 0 at inner ends and last_val[i] (NULL = 0) at the final row. */
int dpenv_gae(const float* rew, const float* val, const uint8_t* end, const float* boot, const float* last_val,
              int32_t T, int32_t n, float gamma, float lam, float* adv_out, float* ret_out, dpenv_stream s);
/* The same scan, and in the same pass the statistics the normalisation needs: stats_out[0] = sum of adv, stats_out[1] = sum of
 * adv^2 over the block (device doubles; accumulated in double in a fixed order, so two runs give the same bits).  workspace:
 * dpenv_gae_workspace_bytes(n) bytes of device memory (per-workgroup partials), needed when stats_out is given.
 * A lane owns two adjacent env columns when n % 2 == 0 and the blocks are 8-byte aligned (8-byte row accesses), and rows
 * are fetched two 8-row groups ahead of the recurrence: the scan is bound by HBM (17 B per env-step, 21 B with boot). */
int64_t dpenv_gae_workspace_bytes(int32_t n);
int dpenv_gae_stats(const float* rew, const float* val, const uint8_t* end, const float* boot, const float* last_val,
                    int32_t T, int32_t n, float gamma, float lam, float* adv_out, float* ret_out, void* workspace,
                    double* stats_out, dpenv_stream s);
/* Advantage normalisation (TrajectoryBuffer.get, ppo.py:99-103 + mpi_tools.py:71-92):
 * adv = (adv - mean) / (std + 1e-8), mean and population std over ALL ranks' samples.
 * One-pass form: all-reduce (sum) the two doubles of dpenv_gae_stats and the sample count over the ranks, then
 *   dpenv_adv_apply_stats(adv, count, stats, total_count): mean = stats[0] / total_count, std = sqrt(stats[1] / total_count - mean^2).
 * Three-pass form (the reference's own order, two all-reduces): sum -> [mean = sum/count] -> sum of squared deviations ->
 * [std = sqrt(sumsq/count)] -> apply; sum_out, sumsq_out, mean, std are device float scalars.  The two reductions are
 * deterministic (double partials added in a fixed order) and share one scratch buffer per device: do not run them
 * concurrently on two streams of one device. */
int dpenv_adv_apply_stats(float* adv, int64_t count, const double* stats, double total_count, dpenv_stream s);
int dpenv_adv_sum(const float* adv, int64_t count, float* sum_out, dpenv_stream s);
int dpenv_adv_sumsq(const float* adv, int64_t count, const float* mean, float* sumsq_out, dpenv_stream s);
int dpenv_adv_apply(float* adv, int64_t count, const float* mean, const float* std, dpenv_stream s);

int dpenv_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DPENV_H */
