// dpenv_dev.h - internal contract between the host API (dpenv_api.hip) and the kernels
// (dpenv_kernels.hip) of libdpenv.so.  Not part of the public ABI (that is include/dpenv.h).
#ifndef DPENV_DEV_H
#define DPENV_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef DPENV_BLOCK
#define DPENV_BLOCK 64    // threads per workgroup = one wave64; one lane per environment; LDS staging is wave-private
#endif

namespace dpenv {

constexpr int BLOCK = DPENV_BLOCK;
constexpr int RBLOCK = 64;        // rollout kernel: one wave per workgroup (wave-private LDS transposes)
constexpr int MAX_SWITCH = 8;
constexpr int MAX_CLASSES = 64;

// kernel specialisations: variant x azimuth-head style (customEnv.py:11,327,351,373 + cont_ang :90)
enum { MODE_FULL = 0, MODE_SIMPLE = 1, MODE_LIMITED = 2, MODE_FINAL_WRAP = 3, MODE_FINAL_CONT = 4 };
enum { LAYOUT_AOS = 0, LAYOUT_SOA = 1 };
enum { WRAP_REFERENCE = 0, WRAP_RADIANS = 1 };
enum { DONE_TERMINAL = 1u, DONE_TIMELIMIT = 2u, DONE_FAULT = 4u };

// derived per-class parameter block (host precomputes the mass-matrix inverse)
enum {
    VD_M11 = 0, VD_M22, VD_M23, VD_INV11, VD_I22, VD_I23, VD_I33,
    VD_XU, VD_XUU, VD_YV, VD_YVV, VD_YR, VD_NV, VD_NR, VD_NRR, VD_NUV, VD_YUR,
    VD_KF, VD_KR = VD_KF + 3, VD_LX = VD_KR + 3, VD_LY = VD_LX + 3,
    VD_COUNT = VD_LY + 3
};

struct VesselDev {
    float p[VD_COUNT];
};

// Per-ENV parameter blocks (dpenv_set_vessel_params / dpenv_set_vessel_randomisation; north_star: "per-env 3x3 mass / Coriolis /
// damping blocks", SURVEY appendix D: domain randomisation).  One block = the VD_COUNT derived floats above + the raw m33 (so that the
// public parameter vector can be read back) + 2 pad = ENV_GROUPS float4.  HBM layout float4 ET[ENV_GROUPS][stride] - group g of env i at
// ET[g * stride + i], the same structure-of-float4-streams form as the state (S0..S3): a wave-level access is one coalesced 1 KiB
// dwordx4 transaction.  128 B per env-step read by dpenv_step (SURVEY 8d accounts 3 x 9 x 4 = 108 B for it: 285 B per env-step).
constexpr int VD_M33 = VD_COUNT;          // slot of the raw m33 in a per-env block
constexpr int ENV_GROUPS = 8;
constexpr int ENV_BLOCK_FLOATS = 4 * ENV_GROUPS;
static_assert(VD_COUNT + 1 <= ENV_BLOCK_FLOATS, "per-env block: derived parameters + m33 must fit eight float4");
constexpr int RAND_NPARAM = 32;           // public parameters DPENV_P_M11 .. DPENV_P_KLR_STAR (include/dpenv.h): the whole vector
// The six inflow thrust-loss coefficients (DPENV_P_KLF_* / DPENV_P_KLR_*) are NOT part of the Vessel the kernels carry in registers: they
// are two more rows of the per-env table, ET[ENV_GROUPS] = (Klf bow, port, star, Klr bow) and ET[ENV_GROUPS + 1] = (Klr port, star, -, -),
// read at the force map of an env step ONLY by the GENERAL per-env kernels (VES_ENV_RND below; the closed loop's RND instantiations and its
// one-wave forms) and only while StepArgs.loss_on says some env has one.  Every other kernel passes a compile-time "no table" and is, to
// the instruction, what it was before the loss existed: built both ways (round 5) - with the coefficients inside the Vessel the 256-env f16
// closed loop went from 76 to 100 B of scratch, behind a run-time flag in every kernel from 76 to 92 B and the headline step kernel from
// 63 to 66 VGPRs (one wave per SIMD less).  In the randomisation's draw they are groups 8 and 9.
constexpr int LOSS_GROUPS = 2;
constexpr int DRAW_GROUPS = ENV_GROUPS + LOSS_GROUPS;
constexpr int RAND_TAB_FLOATS = 64;       // device table of the randomisation: [0..31] nominal public parameters, [32..63] relative half-ranges

// where step_kernel takes a lane's vessel from (template argument VES)
enum { VES_ARGS = 0,      // one class: kernel arguments (SGPRs)
       VES_CLASS_LDS = 1, // vessel classes: [class][param] table staged into LDS as [param][class]
       VES_ENV_VGPR = 2,  // per-env blocks: eight coalesced float4 loads per lane straight into registers
       VES_ENV_LDS = 3,   // per-env blocks: LDS-DMA (global_load_lds_dwordx4) into a [group][lane] image, read back when needed (the A/B of SURVEY 7)
       VES_ENV_RND = 4,   // the GENERAL per-env form: VES_ENV_VGPR + the domain randomisation's hull re-draw in the reset paths (while StepArgs.rand_tab)
                          //   + the inflow thrust loss (while StepArgs.loss_on).  Its own instantiation: the draw's four Philox blocks cost the register
                          //   allocation of the other forms 15-70 VGPRs when they share the code, the loss 10-16
       VES_ARGS_LOSS = 5 }; // the SHARED training form (round 6): one class, hull AND the six thrust-loss coefficients as kernel arguments (StepArgs.v0,
                          //   StepArgs.kl; SGPRs; zeros = no loss: the default's rows bit for bit) + the per-episode current re-draw in the reset
                          //   paths (while StepArgs.cur_nom) - the regime the reference trains in (customEnv.py:17,26) and a randomised current at
                          //   the default's memory traffic, not at 160 B per env-step of identical per-env blocks
// StepArgs.loss_on
enum { LOSS_NONE = 0,     // no thrust loss anywhere
       LOSS_TABLE = 1,    // the per-env table's coefficients are applied: some env has one - or the host does not know yet (a setter recorded into a
                          //   graph, an answer still on its way): zero coefficients are neutral bit for bit, so assuming a loss is always right
       LOSS_SHARED = 3 }; // the single class's coefficients in StepArgs.kl (possibly all zero): only kernels instantiated for it are launched with this value

struct StepArgs {
    // library-owned state streams (see dpenv_kernels.hip header)
    float4* S0;
    float4* S1;
    float4* S2;
    float4* RF;
    int32_t* episode;
    // per-call I/O (caller-owned device memory)
    const float* action;
    const float* new_ref;
    void* obs;
    float* rew;
    uint8_t* done;
    float* parts;
    void* final_obs;
    // optional per-env inputs owned by the library
    float* cur_vc;            // present current speed / direction (read-write with drift)
    float* cur_beta;
    float* cur_vc0;           // means the drift reverts to (written by a reset when the current is randomised per episode)
    float* cur_beta0;
    uint32_t* drift_ctr;      // per-env draw counter of the drift noise
    int32_t current_drift;
    float drift_a;            // dt / tau
    float drift_sv;           // sigma_v * sqrt(2 dt / tau)
    float drift_sb;           // sigma_beta * sqrt(2 dt / tau)
    const int32_t* class_id;
    const float* class_tab;   // [n_classes][VD_COUNT]
    int32_t n_classes;
    VesselDev v0;             // class 0 by value -> SGPRs on the single-class path
    int32_t n;
    int32_t n_substeps;
    float h;
    float inv_dt;             // 1 / (n_substeps * h)
    int32_t max_ep_len;
    int32_t terminate;
    int32_t auto_reset;
    int32_t wrap_mode;
    int32_t action_layout;
    int32_t obs_layout;
    int32_t obs_bf16;
    int32_t hold_plant;
    uint32_t seed_lo, seed_hi;
    int64_t env_id_base;
    float reset_fraction;
    int32_t reset_acts;       // customEnv.py:179-188: previous thrust drawn at reset
    uint32_t* noise_ctr;      // per-env count of exploration-noise draws made so far (in-kernel sampling)
    float4* S3;               // thrust columns (o[6..8]) of the observation the LAST closed-loop launch ended with, see PolicyArgs.use_lag
    float4* env_tab;          // per-env parameter blocks ET[ENV_GROUPS][env_stride], NULL = classes / the single class (written by the kernels only
                              //   when the randomisation re-draws a hull)
    int32_t env_stride;
    int32_t loss_on;          // LOSS_* below the VES_ enum: whether / where from the kernels that carry the inflow thrust loss apply it
    const float* rand_tab;    // domain randomisation on: device float[RAND_TAB_FLOATS] (nominal | relative half-range); every reset - explicit,
                              //   auto, reset_at_end - re-draws the env's hull for the new episode, Philox keyed (seed; global env id, episode)
    // ---- round 6 (appended: the fields above keep their offsets) ----
    float kl[8];              // LOSS_SHARED: Klf bow, port, star | Klr bow, port, star of the single class (two pad)
    const float* cur_nom;     // per-episode randomisation of the current on (dpenv_set_current_randomisation): device float[2][cur_nom_stride], the
                              //   nominal V_c | beta_c of every env; every reset draws the new episode's current around them (current_redraw)
    int32_t cur_nom_stride;
    float cur_range_v;        // half-ranges of the draw: V_c [m/s], beta_c [rad]
    float cur_range_b;
};

// fused T-step rollout (dpenv_rollout)
struct RolloutArgs {
    int32_t T;
    const float* actions;     // [T][n][A] or [T][A][n]
    void* obs;                // [T][n][OD] or [T][OD][n]
    float* rew;               // [T][n]
    uint8_t* done;            // [T][n]
    int32_t n_switch;
    int32_t switch_step[MAX_SWITCH];
    const float* refs;        // [n_switch][3][n]
};

// actor-critic evaluated in-kernel (dpenv_policy.hip, dpenv_policy_x.hip)
struct PolicyArgs {
    const uint4* frags;       // LDS image, first part: [2 nets][nent] x 16 B MFMA A-operand fragment entries (f16), actor then critic;
                              // split arithmetic: followed by the same for the LOW parts of the weights (W = hi + lo)
    const float* bias;        // [2][nblk][32] f32: bias tiles of the row-blocks after the first layer, accumulator layout
    const float* consts;      // [24] f32 written by the packing kernel: exp(log_std) (core.py:84) | 1 / (exp(log_std) + 1e-8)
                              // (core.py:45) | -log_std - 0.5 log(2 pi) (core.py:45), 8 slots each
    int32_t nent;             // 16-byte entries of one weight image of one network (compact fragment layout, FragAddr in dpenv_policy_dev.h)
    int32_t nblk;             // bias blocks per net = 3 (n_hidden - 1) + 1
    int32_t ks;               // k-steps of 16 hidden features: 5 (width <= 80) or 6 (width <= 96)
    int32_t act;              // hidden activation: 0 leaky-relu(leak), 1 tanh
    int32_t ws;               // rollout launch form: 0 = one wave per 64 envs, 1 = an env wave and a network wave per 64 envs (policy_rollout_ws_kernel)
    int32_t split;            // 1 = split-f16 arithmetic (DPENV_POLICY_F32): weights and activations as hi + lo f16 pairs
    int32_t critic_f16;       // with split: the critic is evaluated in f16 arithmetic on the high image (DPENV_POLICY_F32_ACTOR)
    int32_t n_hidden;
    float leak;               // leaky-relu slope (0.2)
    int32_t sample;           // 1: noise == NULL means "draw the exploration noise in the kernel" (policy_noise), not "a = mu"
    int32_t reset_at_end;     // 1: every env is cut and re-drawn after step T-1 (ppo.py:305-322), boot = V(last obs) unless terminal
    int32_t use_lag;          // 1: the state was last touched by a closed-loop launch: its first policy input takes the thrust columns
                              // from S3.  The observation of step t carries the thrust command of step t-1 (customEnv.py:196-205: state_ext is
                              // filled BEFORE prev_thrust is updated, :126), the state block only the command of step t; without the lagged
                              // copy a launch that continues an episode would start from an observation the reference never shows its policy.
    int32_t ws_groups;        // two-wave form: 4 = 512-thread workgroups of 256 envs, 2 = 256-thread workgroups of 128 envs (a SIMD per wave)
    // rollout I/O
    int32_t T;
    const float* noise;       // [T][n][A] standard normal draws, NULL = deterministic (a = mu)
    void* obs_out;            // [T][n][OD]  policy input of step t   (ppo.py:298 'o'); f32 or bf16
    float* act_out;           // [T][n][A]   action taken             ('a')
    float* rew;               // [T][n]
    float* val;               // [T][n]      V(o_t)                   ('v_t')
    float* logp;              // [T][n]                               ('logp_t')
    uint8_t* done;            // [T][n]
    float* boot;              // [T][n]      bootstrap value where a path ends (ppo.py:311), else 0
    void* last_obs;           // [n][OD]     policy input of the next launch; f32 or bf16
    float* last_val;          // [n]
    int32_t n_switch;
    int32_t switch_step[MAX_SWITCH];
    const float* refs;
    uint32_t* dbg;            // diagnostic builds only (DPENV_WS_SELFCHECK): event records, NULL otherwise
};

// device-side weight packing (pack_policy_kernel): one dense network, DEVICE pointers
struct PackNet {
    const float* W[5];        // W[l][in][out] row-major (tf.layers.dense kernel layout)
    const float* b[5];
    int32_t n_layers, in_dim, H, out_dim;
};

// which arithmetics get a critic wave of their own (ROLES = 3) in the 128-env geometry of the two-wave closed loop: bit 0 f16, bit 1 all
// exact, bit 2 exact actor (dpenv_policy_ws.h has the measurements); here because dpenv_get_policy_launch_ex reports the resolved form
#ifndef DPENV_WS_CRITIC_WAVE
#define DPENV_WS_CRITIC_WAVE 6
#endif
constexpr int POLICY_WS_MAILBOX_BYTES = 4 * (64 * 9 * 5 + 64 * 4 + 64) * 4;   // two-wave form: four groups of mailboxes
constexpr int POLICY_WS_MAILBOX_X_BYTES = 4 * (64 * 9 * 4 + 64 * 4 + 64) * 4; // the same for the split arithmetics (no row staging area)
constexpr int PREC_F16 = 0, PREC_F32 = 1, PREC_F32_ACTOR = 2;                 // = DPENV_POLICY_* of include/dpenv.h
constexpr int POLICY_STAGING_BYTES = 4 * 64 * 9 * 4;                           // one-wave form: four wave-private row areas

}  // namespace dpenv

extern "C" {
hipError_t dpenv_dev_launch_pack_policy(const dpenv::PackNet* pi, const dpenv::PackNet* v, const float* log_std, int adim, int ks,
                                        int nent, int nblk, int split, void* frags, float* bias, float* consts, hipStream_t s);
hipError_t dpenv_dev_launch_policy_forward_x(const dpenv::PolicyArgs* pa, int od, int adim, const float* obs, float* mu,
                                             float* v, int n, hipStream_t s);
hipError_t dpenv_dev_launch_policy_rollout_x(const dpenv::StepArgs* a, const dpenv::PolicyArgs* pa, int mode, int ext,
                                             hipStream_t s);
hipError_t dpenv_dev_launch_policy_rollout_ws(const dpenv::StepArgs* a, const dpenv::PolicyArgs* pa, int mode, int ext, hipStream_t s);
// two-wave form of the split arithmetics (dpenv_policy_xws1.hip: PREC_F32, dpenv_policy_xws2.hip: PREC_F32_ACTOR)
hipError_t dpenv_dev_launch_policy_rollout_xws_f32(const dpenv::StepArgs* a, const dpenv::PolicyArgs* pa, int mode, int ext,
                                                   hipStream_t s);
hipError_t dpenv_dev_launch_policy_rollout_xws_f32_actor(const dpenv::StepArgs* a, const dpenv::PolicyArgs* pa, int mode, int ext,
                                                         hipStream_t s);
hipError_t dpenv_dev_launch_policy_forward(const dpenv::PolicyArgs* pa, int od, int adim, const float* obs, float* mu,
                                           float* v, int n, hipStream_t s);
hipError_t dpenv_dev_launch_policy_rollout(const dpenv::StepArgs* a, const dpenv::PolicyArgs* pa, int mode, int ext,
                                           hipStream_t s);
hipError_t dpenv_dev_launch_rollout(const dpenv::StepArgs* a, const dpenv::RolloutArgs* ra, int mode, int ext, int ves, int two_wave,
                                    hipStream_t s);
// ves: VES_* (where the vessel of a lane comes from)
hipError_t dpenv_dev_launch_step(const dpenv::StepArgs* a, int mode, int ext, int ves, int reset_wave, hipStream_t s);
// raw public parameters -> per-env blocks: raw[p * p_stride + i * i_stride] (SoA block: p_stride = n, i_stride = 1; one vector for every env:
// p_stride = 1, i_stride = 0); and back (out[p * n + i])
// tab: ET[DRAW_GROUPS][stride] (vessel block + thrust-loss rows); loss_flag (device word, may be NULL) is OR-ed with 1 if any env's
// thrust-loss coefficient is non-zero
hipError_t dpenv_dev_launch_pack_env_vessels(const float* raw, int64_t p_stride, int64_t i_stride, float4* tab, uint32_t* loss_flag,
                                             int stride, int n, hipStream_t s);
hipError_t dpenv_dev_launch_unpack_env_vessels(const float4* tab, int stride, float* out, int n, hipStream_t s);
hipError_t dpenv_dev_launch_reset(const dpenv::StepArgs* a, int mode, int ext, const uint8_t* mask, const float* init,
                                  const float* ref, hipStream_t s);
hipError_t dpenv_dev_launch_get_state(const dpenv::StepArgs* a, float* st, int32_t* ctr, hipStream_t s);
hipError_t dpenv_dev_launch_set_state(const dpenv::StepArgs* a, const float* st, const int32_t* ctr, hipStream_t s);
hipError_t dpenv_dev_launch_thrust_map(const dpenv::VesselDev* vd, const float* n_pct, const float* alpha, float* tau,
                                       int n, hipStream_t s);
int64_t dpenv_dev_gae_workspace_bytes(int n);
hipError_t dpenv_dev_launch_gae(const float* rew, const float* val, const uint8_t* end, const float* boot,
                                const float* last_val, int T, int n, float gamma, float lam, float* adv, float* ret,
                                double* workspace, double* stats, hipStream_t s);
hipError_t dpenv_dev_launch_sum(const float* x, int64_t count, const float* mean, float* out, hipStream_t s);
hipError_t dpenv_dev_launch_adv_apply(float* x, int64_t count, const float* mean, const float* std, const double* stats,
                                      double total_count, hipStream_t s);
}

#endif
