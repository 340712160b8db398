/*
 * dpenv_oracle.h - CPU oracle for the ReVolt dynamic-positioning env.step path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (ml4ca_amd/, libdpenv.so)
 * includes, links or calls this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it - as the checker, never as the thing shipped.
 *
 * It is a scalar, one-env-at-a-time restatement in plain C of the reference's
 * Python (simensov/ml4ca, paths relative to the reference root; WW =
 * src/rl/windows_workspace):
 *   action decode / clip / command map   WW/specific/customEnv.py:102-122,215-244
 *   pose-error observation               WW/specific/errorFrame.py:25-37,
 *                                        WW/specific/misc/mathematics.py:7-17,
 *                                        WW/specific/customEnv.py:196-205
 *   reward                               WW/specific/customEnv.py:253-325
 *   termination                          WW/specific/customEnv.py:207-213
 *   reset                                WW/specific/customEnv.py:135-194,
 *                                        WW/specific/misc/simtools.py:109-123
 *   thruster force map tau = B(alpha)F   src/sl/SupervisedTau.py:42-83,
 *                                        src/qp/ROS/qp_allocator/src/qp_allocator.py:44-55,69-70
 *   GAE buffer                           WW/spinup/algos/tf1/ppo/ppo.py:65-105,
 *                                        WW/spinup/algos/tf1/ppo/core.py:48-63,
 *                                        WW/spinup/utils/mpi_tools.py:71-92
 *
 * PINNING: the float64 build of every stage above is checked against golden
 * vectors generated from the imported reference (tools/gen_golden.py ->
 * tests/golden/ fixtures; tests/test_oracle_golden.py).
 *
 * PARITY UNPINNED for one stage: the hull/thruster plant (customEnv.py:124 ->
 * digitwin.py:213-219 -> closed-source Cybersea simulator, absent from the
 * reference tree).  dpo_plant_* below is a BUILD-OWNED 3-DOF model (DESIGN.md
 * section 3); for that stage the oracle only pins HIP-vs-CPU agreement.
 *
 * Every function exists twice: suffix _f64 (REAL = double) and _f32 (REAL = float).
 */
#ifndef DPENV_ORACLE_H
#define DPENV_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DPO_FULL = 0, DPO_SIMPLE = 1, DPO_LIMITED = 2, DPO_FINAL = 3 };
enum { DPO_WRAP_REFERENCE = 0, DPO_WRAP_RADIANS = 1 };

/* canonical state field order, SoA: state[field * n_envs + env] */
enum {
    DPO_S_N = 0, DPO_S_E, DPO_S_PSI, DPO_S_U, DPO_S_V, DPO_S_R,
    DPO_S_REF_N, DPO_S_REF_E, DPO_S_REF_PSI,
    DPO_S_PT_BOW, DPO_S_PT_PORT, DPO_S_PT_STAR,   /* previous thrust command, percent */
    DPO_S_A_BOW, DPO_S_A_PORT, DPO_S_A_STAR,      /* current azimuth command, rad */
    DPO_NSTATE
};
/* counters[0*n+env] = steps taken in this episode, counters[1*n+env] = episodes sampled so far */

/* vessel parameter vector (env thruster order: bow, port, star) */
enum {
    DPO_P_M11 = 0, DPO_P_M22, DPO_P_M23, DPO_P_M33,
    DPO_P_XU, DPO_P_XUU, DPO_P_YV, DPO_P_YVV, DPO_P_YR, DPO_P_NV, DPO_P_NR, DPO_P_NRR,
    DPO_P_KF_BOW, DPO_P_KF_PORT, DPO_P_KF_STAR,
    DPO_P_KR_BOW, DPO_P_KR_PORT, DPO_P_KR_STAR,
    DPO_P_LX_BOW, DPO_P_LX_PORT, DPO_P_LX_STAR,
    DPO_P_LY_BOW, DPO_P_LY_PORT, DPO_P_LY_STAR,
    DPO_P_NUV, DPO_P_YUR,                          /* lift-type cross-flow terms N_uv u v, Y_ur u r */
    DPO_P_KLF_BOW, DPO_P_KLF_PORT, DPO_P_KLF_STAR, /* inflow thrust loss, n >= 0: F = K n|n| - Kl |n| u_a (build-owned) */
    DPO_P_KLR_BOW, DPO_P_KLR_PORT, DPO_P_KLR_STAR, /* n < 0 */
    DPO_NPARAM = 32
};

typedef struct dpo_config {
    int32_t variant;          /* DPO_FULL.. */
    int32_t extended_state;   /* obs 9 (1) or 6 (0) */
    int32_t cont_ang;         /* final only: 7 actions, sin/cos azimuth heads */
    int32_t n_substeps;       /* 20 (customEnv.py:79-80) */
    double  substep_dt;       /* 0.01 (customEnv.py:81) */
    int32_t wrap_mode;        /* DPO_WRAP_REFERENCE replicates quirk Q1 */
    int32_t terminate;        /* 1: evaluate is_terminal; 0: never terminal */
    int32_t max_ep_len;       /* time-limit in env steps, 0 = none (ppo.py:304) */
    int32_t auto_reset;       /* resample finished envs inside step */
    int32_t current_enabled;  /* constant irrotational current per env */
    uint64_t seed;
    int64_t env_id_base;      /* global id of local env 0 (rank-count invariance) */
    double  reset_fraction;   /* 0.8 (customEnv.py:135, ppo.py:286) */
    int32_t current_drift;    /* build-defined (config 5): Gauss-Markov drift of the current, once per env step */
    double  current_tau;      /* s */
    double  current_sigma_v;  /* m/s */
    double  current_sigma_beta; /* rad */
    int32_t reset_acts;       /* customEnv.py:30,179-188: previous thrust clip(100 N(0, 0.1)) at reset */
} dpo_config;

#define DPO_DECL(suffix, REAL)                                                                         \
    int  dpo_act_dim_##suffix(const dpo_config* c);                                                    \
    int  dpo_obs_dim_##suffix(const dpo_config* c);                                                    \
    void dpo_default_vessel_##suffix(REAL* p);                                                         \
    void dpo_thrust_loss_vessel_##suffix(REAL* p);                                                     \
    void dpo_dynpos_fit_vessel_##suffix(REAL* p);                                                      \
    void dpo_decode_##suffix(const dpo_config* c, const REAL* action, const REAL ang_in[3],            \
                             REAL thrust_out[3], REAL ang_out[3]);                                     \
    void dpo_thrust_map_##suffix(const REAL* vessel, const REAL n_pct[3], const REAL alpha[3],         \
                                 REAL tau[3]);                                                         \
    void dpo_plant_##suffix(const dpo_config* c, const REAL* vessel, REAL eta[3], REAL nu[3],          \
                            const REAL n_pct[3], const REAL alpha[3], const REAL* current);            \
    void dpo_obs_##suffix(const dpo_config* c, const REAL eta[3], const REAL nu[3], const REAL ref[3], \
                          const REAL prev_thrust[3], REAL* obs);                                       \
    void dpo_reward_##suffix(const dpo_config* c, const REAL* obs, const REAL thrust_now[3],           \
                             const REAL ang_cur[3], const REAL ang_prev[3], REAL parts[4]);            \
    int  dpo_done_##suffix(const dpo_config* c, const REAL* obs);                                      \
    void dpo_sample_reset_##suffix(const dpo_config* c, int64_t env_gid, uint32_t episode,             \
                                   REAL eta[3], REAL nu[3]);                                           \
    void dpo_policy_noise_##suffix(const dpo_config* c, int64_t env_gid, uint32_t draw, int32_t adim, REAL* xi); \
    void dpo_draw_vessel_##suffix(const dpo_config* c, const REAL* rand_tab, int64_t env_gid,          \
                                  uint32_t episode, REAL* params);                                     \
    void dpo_draw_current_##suffix(const dpo_config* c, int64_t env_gid, uint32_t episode, REAL nom_v, \
                                   REAL nom_b, REAL range_v, REAL range_b, REAL* out);                 \
    void dpo_reset_##suffix(const dpo_config* c, int32_t n, REAL* state, int32_t* counters,            \
                            const uint8_t* mask, const REAL* init, const REAL* ref, REAL* obs,         \
                            REAL* vessel_env, const REAL* rand_tab, REAL* current, REAL* current_mean, \
                            const REAL* cur_rand);                                                     \
    void dpo_step_##suffix(const dpo_config* c, const REAL* vessel, int32_t n, REAL* state,            \
                           int32_t* counters, const REAL* action, const REAL* new_ref,                 \
                           const REAL* plant_override, REAL* current, REAL* obs, REAL* rew,            \
                           uint8_t* done, REAL* parts, REAL* final_obs, REAL* current_mean,            \
                           uint32_t* drift_ctr, REAL* vessel_env, const REAL* rand_tab,                \
                           const REAL* cur_rand);                                                      \
    void dpo_discount_cumsum_##suffix(const REAL* x, int32_t n, REAL discount, REAL* y);               \
    void dpo_gae_##suffix(const REAL* rew, const REAL* val, const uint8_t* end, const REAL* boot,      \
                          const REAL* last_val, int32_t T, int32_t n, REAL gamma, REAL lam,            \
                          REAL* adv, REAL* ret);                                                       \
    void dpo_normalize_adv_##suffix(REAL* adv, int64_t count, REAL* mean_std);

DPO_DECL(f64, double)
DPO_DECL(f32, float)

int dpo_set_threads(int n);
void dpo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
