// dpenv_policy.hip - the PPO actor-critic evaluated INSIDE the rollout launch.
//
// Reference: mlp_gaussian_policy / mlp_actor_critic, src/rl/windows_workspace/spinup/algos/tf1/ppo/core.py:29-33,
// 80-107 (dense layers y = x W + b with W[in][out], hidden activation leaky_relu(0.2) for the shipped model -
// train.py:24,31, config.json - linear output, log_std parameter, pi = mu + N(0,1) exp(log_std),
// gaussian_likelihood core.py:42-46), consumed by the rollout loop ppo.py:289-322 which stores
// (o, a, r, v, logp) per step (ppo.py:298).
//
// Mapping to CDNA4.  A wave owns 64 environments (one per lane) and evaluates the MLP for all of them at
// once on the matrix cores in the TRANSPOSED orientation  H_next^T [features x envs] = W^T [out x in] . H^T:
//   * A operand  = weights, pre-permuted on the host into MFMA fragments, read from LDS (one 16-byte
//                  ds_read_b128 per lane per fragment); one LDS copy per 256-thread workgroup.
//   * B operand  = activations.  The f32 accumulator tile of v_mfma_f32_32x32x16_f16 has the env on the lane and
//                  the feature in the register index, which is exactly what the next layer's B operand wants:
//                  leaky-relu + convert to f16 in place, no lane movement, no LDS (cdna_hip_programming.md
//                  section 3, "an accumulator tile as the next MFMA's operand"; the k-order permutation that
//                  comes with it is folded into the host-side weight packing).
//   * biases     = first layer: a constant-1 input (slot 15) whose weight column holds the bias; later layers: the
//                  accumulator's INITIAL value (srcC of the first MFMA of a row-block is a bias tile read from LDS,
//                  one 64-byte broadcast read per lane half), so hidden width 80 needs K = 80 = 5 k-steps, not 96.
//   * in/out     = the env-per-lane <-> (32-env tile, lane-half) exchange at both ends is one
//                  v_permlane32_swap per register.
// Precision: f16 weights and activations, f32 accumulation.  This is the only MFMA use in the library; the
// environment itself stays scalar fp32 physics.
#include "dpenv_policy_dev.h"

namespace dpenv {

// =============================================================================================
//  standalone forward pass: mu [n][A], v [n] for obs [n][OD]   (deterministic policy / validation)
// =============================================================================================
template <int OD, int A, int KA>
__global__ __launch_bounds__(PBLOCK) void policy_forward_kernel(const PolicyArgs pa, const float* obs, float* mu_out,
                                                                 float* v_out, int n)
{
    extern __shared__ uint4 lds_dyn[];
    uint4* lds_w = lds_dyn;                                                     // [2 * nfrag][64] fragments
    const float* lds_b = (const float*)(lds_dyn + 2 * pa.nfrag * 64);                       // [2][nblk][32] bias floats
    float* lds_io = (float*)lds_dyn + policy_lds_io_offset_floats(pa) + (threadIdx.x >> 6) * (64 * 9);   // wave-private staging
    stage_weights(lds_w, pa);
    const int lane = threadIdx.x & 63;
    const int wave0 = blockIdx.x * PBLOCK + (threadIdx.x & ~63);               // first env of this wave
    if (wave0 >= n) return;
    const int i = wave0 + lane;
    const bool live = i < n;
    float pre[OD], o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    load_rows<OD, 64>(obs + (int64_t)wave0 * OD, (int64_t)(n - wave0) * OD, lane, pre);
    wave_rows_from_regs<OD>(lds_io, pre, o, lane);
    half8 in0, in1;
    obs_to_frags<OD>(o, in0, in1);
    float mu[8], vv[8];
#if DPENV_JOINT_EVAL
    mlp_eval2<KA>(lds_w, lds_w + pa.nfrag * 64, lds_b, lds_b + pa.nblk * 32, pa.n_hidden, in0, in1, (_Float16)pa.leak, mu, vv);
#else
    mlp_eval<KA>(lds_w, lds_b, pa.n_hidden, in0, in1, (_Float16)pa.leak, mu);
    mlp_eval<KA>(lds_w + pa.nfrag * 64, lds_b + pa.nblk * 32, pa.n_hidden, in0, in1, (_Float16)pa.leak, vv);
#endif
    wave_store_rows<A>(lds_io, mu_out, (int64_t)wave0 * A, (int64_t)(n - wave0) * A, mu, lane);
    if (live) v_out[i] = vv[0];
}

// =============================================================================================
//  policy-in-the-loop rollout: T steps of (policy -> sample -> env.step -> value) per launch,
//  writing the PPO trajectory rows (ppo.py:298) straight into [T][n][.] blocks.
// =============================================================================================
template <int MODE, bool EXT, int KA>
__global__ __launch_bounds__(PBLOCK) void policy_rollout_kernel(const StepArgs a, const PolicyArgs pa)
{
    constexpr int A = ModeTraits<MODE>::A;
    constexpr int OD = EXT ? 9 : 6;
    extern __shared__ uint4 lds_dyn[];
    uint4* lds_w = lds_dyn;
    float* lds_io = (float*)lds_dyn + policy_lds_io_offset_floats(pa) + (threadIdx.x >> 6) * (64 * 9);
    stage_weights(lds_w, pa);
    const uint4* Wpi = lds_w;
    const uint4* Wv = lds_w + pa.nfrag * 64;
    const float* Bpi = (const float*)(lds_dyn + 2 * pa.nfrag * 64);
    const float* Bv = Bpi + pa.nblk * 32;
    const _Float16 leak = (_Float16)pa.leak;

    const int lane = threadIdx.x & 63;
    const int n = a.n;
    const int wave0 = blockIdx.x * PBLOCK + (threadIdx.x & ~63);
    if (wave0 >= n) return;                                      // whole wave out of range (uniform)
    const int i = wave0 + lane;
    const bool live = i < n;
    const int il = live ? i : n - 1;

    Env s;
    load_env(a, il, s);
    sincos_lean(s.psi, s.sn, s.cs);
    Current cur = {0.0f, 0.0f, 0.0f, 0.0f, 0u};
    float vc0 = 0.0f, beta0 = 0.0f;
    if (a.cur_vc) {
        cur.vc = a.cur_vc[il]; cur.beta = a.cur_beta[il];
        if (a.current_drift) { vc0 = a.cur_vc0[il]; beta0 = a.cur_beta0[il]; cur.ctr = a.drift_ctr[il]; }
        current_components(cur);
    }
    // single class: SGPR-resident (the VGPRs are needed by the MLP); classes: per lane from the table, once per launch
    const Vessel ve = (a.n_classes > 1) ? vessel_from_table(a.class_tab, a.class_id[il]) : vessel_from_args(a.v0);
    uint32_t episode = a.auto_reset ? (uint32_t)a.episode[il] : 0u;
    bool ep_dirty = false, rf_dirty = (MODE == MODE_FULL);
    const PolicyConsts<A> pc = load_policy_consts<A>(pa);
    const bool draw = pa.noise == nullptr && pa.sample != 0;
    uint32_t nctr = draw ? a.noise_ctr[il] : 0u;

    const int64_t stride_a = (int64_t)n * A, stride_o = (int64_t)n * OD;
    const int64_t w_a = (int64_t)wave0 * A, w_o = (int64_t)wave0 * OD;
    const int64_t rem_a = stride_a - w_a, rem_o = stride_o - w_o;

    // observation of the current state = policy input of step 0 (ENV:196-205), and its value
    float o[9];
    {
        float sr_, cr_;
        bool same_;
        make_obs(s.N, s.E, s.psi, s.u, s.v, s.r, s.refN, s.refE, s.refPsi, s.pt, a.wrap_mode == WRAP_REFERENCE, o, sr_, cr_,
                 same_);
    }
    half8 in0, in1;
    obs_to_frags<OD>(o, in0, in1);
    float vout[8], mu[8];
#if DPENV_JOINT_EVAL
    mlp_eval2<KA>(Wpi, Wv, Bpi, Bv, pa.n_hidden, in0, in1, leak, mu, vout);     // actor and critic of o_0
#else
    mlp_eval<KA>(Wv, Bv, pa.n_hidden, in0, in1, leak, vout);
#endif
    float v_t = vout[0];

    float pre[A];
    if (pa.noise) load_rows<A, 64>(pa.noise + w_a, rem_a, lane, pre);
    int next_switch = 0;
    for (int t = 0; t < pa.T; ++t) {
        // ---- store the policy input row; the actor's mean for it is already there (joint evaluation) -------------
        wave_store_rows<OD>(lds_io, pa.obs_out, (int64_t)t * stride_o + w_o, rem_o, o, lane, a.obs_bf16 != 0);
#if !DPENV_JOINT_EVAL
        mlp_eval<KA>(Wpi, Bpi, pa.n_hidden, in0, in1, leak, mu);
#endif
        // ---- sample: a = mu + std * xi (core.py:85), log-likelihood (core.py:42-46) -----------
        float act[A];
        float logp;
        if (pa.noise) {
            float xi[A];
            wave_rows_from_regs<A>(lds_io, pre, xi, lane);
            if (t + 1 < pa.T) load_rows<A, 64>(pa.noise + (int64_t)(t + 1) * stride_a + w_a, rem_a, lane, pre);
            logp = sample_action<A>(pc, mu, xi, act);
        } else if (draw) {
            float xi[A];
            policy_noise<A>(a, a.env_id_base + i, nctr, xi);
            ++nctr;
            logp = sample_action<A>(pc, mu, xi, act);
        } else {
            logp = mean_action<A>(pc, mu, act);
        }
        wave_store_rows<A>(lds_io, pa.act_out, (int64_t)t * stride_a + w_a, rem_a, act, lane);

        // ---- env.step ------------------------------------------------------------------------
        bool has_ref = false;
        float nrN = 0.0f, nrE = 0.0f, nrP = 0.0f;
        if (next_switch < pa.n_switch && pa.switch_step[next_switch] == t) {
            const float* rp = pa.refs + (int64_t)next_switch * 3 * n;
            nrN = rp[il]; nrE = rp[(int64_t)n + il]; nrP = rp[2 * (int64_t)n + il];
            has_ref = true; rf_dirty = true;
            ++next_switch;
        }
        StepOut out;
        env_step<MODE, EXT>(a, ve, s, act, has_ref, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, out);
        if (a.current_drift) current_drift_step(a, cur, vc0, beta0, a.env_id_base + i);

#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = out.o[k];
        const bool do_reset = a.auto_reset && out.d != 0u && live;
#if DPENV_JOINT_EVAL
        // ---- value of the observation this step produced, and the next policy input ------------------------------
        // No env of the wave finished (the common case): the next policy input IS that observation, so one joint
        // evaluation gives V(o') for the bootstrap and the actor's mean for the next step.  Otherwise the critic is
        // run once more on the pre-reset observation of the wave before the finished envs are re-drawn.
        float v_pre = 0.0f;
        if (__ballot(do_reset) != 0ull) {                       // wave-uniform
            // only a CUT episode (time limit) bootstraps with V of its last observation; a terminated one with 0 (ppo.py:311)
            if (__ballot(do_reset && (out.d & DONE_TERMINAL) == 0u) != 0ull) {
                obs_to_frags<OD>(o, in0, in1);
                mlp_eval<KA>(Wv, Bv, pa.n_hidden, in0, in1, leak, vout);
                v_pre = vout[0];
            }
            if (do_reset) {
                env_auto_reset<MODE>(a, s, a.env_id_base + i, episode, o);
                ++episode; ep_dirty = true; rf_dirty = true;
            }
        }
        obs_to_frags<OD>(o, in0, in1);
        mlp_eval2<KA>(Wpi, Wv, Bpi, Bv, pa.n_hidden, in0, in1, leak, mu, vout);
        const float v_next = do_reset ? v_pre : vout[0];
        const float v_new = vout[0];
#else
        // ---- critic on the observation this step produced (pre-reset) --------------------------
        obs_to_frags<OD>(out.o, in0, in1);
        mlp_eval<KA>(Wv, Bv, pa.n_hidden, in0, in1, leak, vout);
        const float v_next = vout[0];
        float v_new = v_next;
        if (__ballot(do_reset) != 0ull) {                       // wave-uniform: rare
            if (do_reset) {
                env_auto_reset<MODE>(a, s, a.env_id_base + i, episode, o);
                ++episode; ep_dirty = true; rf_dirty = true;
            }
            obs_to_frags<OD>(o, in0, in1);
            mlp_eval<KA>(Wv, Bv, pa.n_hidden, in0, in1, leak, vout);
            v_new = do_reset ? vout[0] : v_new;
        }
#endif
        // bootstrap value at a path end (ppo.py:311): 0 if the env terminated, V(o) if only the time limit or the
        // end of this launch cut the path
        const bool terminal = (out.d & DONE_TERMINAL) != 0u;
        const bool ended = (out.d != 0u) || (t == pa.T - 1);
        const float boot = (ended && !terminal) ? v_next : 0.0f;

        if (live) {
            (pa.rew + (int64_t)t * n)[(unsigned)i] = out.reward;
            (pa.done + (int64_t)t * n)[(unsigned)i] = (uint8_t)out.d;
            (pa.val + (int64_t)t * n)[(unsigned)i] = v_t;
            (pa.logp + (int64_t)t * n)[(unsigned)i] = logp;
            (pa.boot + (int64_t)t * n)[(unsigned)i] = boot;
        }
        v_t = v_new;
    }
    // observation after the last step (policy input of the next launch) and final state
    wave_store_rows<OD>(lds_io, pa.last_obs, w_o, rem_o, o, lane, a.obs_bf16 != 0);
    if (live) {
        pa.last_val[i] = v_t;
        store_env(a, i, s, rf_dirty);
        if (ep_dirty) a.episode[i] = (int)episode;
        if (a.current_drift) { a.cur_vc[i] = cur.vc; a.cur_beta[i] = cur.beta; a.drift_ctr[i] = cur.ctr; }
        if (draw) a.noise_ctr[i] = nctr;
    }
}

// =============================================================================================
//  The same rollout with the work of every 64 envs split over TWO waves (wave specialisation).
//
//  At 65 536 envs policy_rollout_kernel puts one wave on each SIMD, and one wave alone issues an instruction every
//  ~2.3 ns however idle the SIMD is (tools/issue_rate.hip).  Here a 512-thread workgroup owns 256 envs with eight waves:
//  E-wave g (0..3) carries the environments of group g, M-wave 4+g their networks, so every SIMD holds an E and an M
//  wave.  The pair hands over through LDS mailboxes and sequence words (no workgroup barrier in the loop).  Per step t:
//      network wave                                          env wave
//      wait o_t; fragments; xi_t -> mailbox                  (rows of step t-1, drift of step t)
//      actor(o_t) -> mu_t, post                              wait mu_t
//      critic(o_t) (+ critic of the pre-reset observation    a_t = mu_t + std xi_t; env.step; reset of finished envs;
//         where step t-1 cut an episode), post V             o_t+1 -> mailbox, post            <- the serial chain ends here
//      draw xi_t+1 (Philox + Box-Muller) while waiting       logp, action / reward / done / observation rows; wait V; val, boot rows
//  The chain actor(o_t) -> env.step(t) -> actor(o_t+1) stays serial; the critic - half of the network work - and the exploration
//  noise run beside the env step, the row bookkeeping beside the actor.  Same mlp_eval chains and same env_step as
//  policy_rollout_kernel: every row is bit-identical.  (ROLES = 2 is the product; ROLES = 3 below is a measured
//  alternative kept behind -DDPENV_WS3.)
// =============================================================================================
constexpr int WSBLOCK = 512;
constexpr int WS_GROUP_FLOATS = 64 * 9 * 5 + 64 * 4 + 64;       // io | obs | pre[2] | mu | v[2] | vpre[2] | sequence words (+ pad)
static_assert(4 * WS_GROUP_FLOATS * 4 == POLICY_WS_MAILBOX_BYTES, "dpenv_dev.h: POLICY_WS_MAILBOX_BYTES out of step with the mailbox layout");

//  ROLES = 3 (768-thread workgroups, three waves per SIMD) splits the NETWORK wave by env tile: wave 4 + g evaluates actor and
//  critic for envs 0..31 of group g, wave 8 + g for envs 32..63 (mlp_eval_tile: half the MFMAs, half the packing, half the
//  registers per wave, no cross-lane moves: a tile's fragments are read from the mailbox in operand layout and its outputs go back
//  in accumulator layout).  The serial chain env.step(t) -> actor(o_t+1) -> env.step(t+1) then holds HALF an actor evaluation,
//  and the two network streams of a SIMD fill each other's MFMA / dependency stalls.  (An env / actor / critic split of the
//  three waves was measured first and is slower than two roles: each network wave still runs a full two-tile evaluation on the
//  chain, at 168 VGPRs it spills; profiles/r02_closed_loop_forms.txt.)
// pair-level hand-over inside a workgroup: a sequence word in LDS, released by one wave and acquired by its partner.
// Both waves of a pair belong to the same workgroup, so they are always co-resident; the waiter sleeps between polls.
__device__ __forceinline__ void ws_post(int* p, int v, int lane)
{
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ws_wait(int* p, int v)
{
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < v)
        __builtin_amdgcn_s_sleep(2);
}
#ifdef DPENV_WS_PROFILE
#define WS_WAIT_T(acc, p, v) do { const uint64_t t0_ = __builtin_amdgcn_s_memtime(); ws_wait(p, v); acc += __builtin_amdgcn_s_memtime() - t0_; } while (0)
#define WS_TIC(t_) const uint64_t t_ = __builtin_amdgcn_s_memtime()
#define WS_TOC(acc, t_) acc += __builtin_amdgcn_s_memtime() - t_
#else
#define WS_WAIT_T(acc, p, v) ws_wait(p, v)
#define WS_TIC(t_)
#define WS_TOC(acc, t_)
#endif

#ifdef DPENV_WS_DEBUG_NOMFMA
#define WS_EVAL(W_, B_, f0_, f1_)                                                   \
    do {                                                                           \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) outv[k_] = 0.01f * (float)(f0_[k_ & 7] + f1_[k_ & 7]); \
        for (int it_ = 0; it_ < 600; ++it_) { _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) outv[k_] = fmaf(outv[k_], 0.999f, 1e-4f); } \
    } while (0)
#else
#define WS_EVAL(W_, B_, f0_, f1_) mlp_eval<KA>(W_, B_, pa.n_hidden, f0_, f1_, leak, outv)
#endif
template <int MODE, bool EXT, int KA, int ROLES>
__global__ __launch_bounds__(256 * ROLES) void policy_rollout_ws_kernel(const StepArgs a, const PolicyArgs pa)
{
    constexpr int A = ModeTraits<MODE>::A;
    constexpr int OD = EXT ? 9 : 6;
    constexpr int THREADS = 256 * ROLES;
    extern __shared__ uint4 lds_dyn[];
    uint4* lds_w = lds_dyn;
    {
        const int total = 2 * pa.nfrag * 64;
        for (int k = threadIdx.x; k < total; k += THREADS) lds_w[k] = pa.frags[k];
        float* lb = (float*)(lds_w + total);
        for (int k = threadIdx.x; k < 2 * pa.nblk * 32; k += THREADS) lb[k] = pa.bias[k];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef DPENV_WS_SWAP_ROLES
    const int role = (ROLES - 1) - (wave >> 2);     // diagnostic: the network waves are the first-dispatched (older) ones
#else
    const int role = wave >> 2;                     // 0 = env wave, 1 = network wave (ROLES 2) / network wave of tile 0 (ROLES 3), 2 = of tile 1
#endif
    const int g = wave & 3;
    constexpr int OBS_SLOTS = 1;
    float* grp = (float*)lds_dyn + policy_lds_io_offset_floats(pa) + g * WS_GROUP_FLOATS;
    float* lds_io = grp;                         // E-wave row staging
    float* obs_mb = grp + 64 * 9;                // [OBS_SLOTS][64][9] o_t (by step parity): one row of 9 per lane (stride 9 is conflict-free)
    float* pre_mb = grp + 64 * 9 * (1 + OBS_SLOTS);   // [2][64][9] pre-reset observation of a cut episode, by step parity
    float* mu_mb = pre_mb + 64 * 9 * 2;          // actor mean, stride 9
    float* v_mb = mu_mb + 64 * 9;                // [2][64] V(o_t), by step parity
    float* vpre_mb = v_mb + 128;                 // [2][64] V(pre-reset o_t), by step parity
    int* seq = (int*)(vpre_mb + 128);            // [0] observations posted, [1] means posted, [2] values posted, [4..5] pre flags,
                                                 // [6] / [7] means / values posted by the network wave of tile 1 (ROLES 3)
    int* flag = seq + 4;
    // Two roles: the NETWORK wave draws the exploration noise (Philox + Box-Muller, ~300 VALU per step) while it waits for the
    // next observation - with the noise in the env wave that wave was the busy one (tools/ws_profile.py).  xi_t travels in the
    // observation mailbox: once the network wave has turned o_t into fragments the rows are free until the env wave writes
    // o_t+1, which it does after it has waited for mu_t and read xi_t.
    constexpr bool M_NOISE = (ROLES == 2);
    float* xi_mb = obs_mb;
    const uint4* Wpi = lds_w;
    const uint4* Wv = lds_w + pa.nfrag * 64;
    const float* Bpi = (const float*)(lds_dyn + 2 * pa.nfrag * 64);
    const float* Bv = Bpi + pa.nblk * 32;
    const _Float16 leak = (_Float16)pa.leak;
    const int n = a.n;
    const int wave0 = blockIdx.x * 256 + g * 64;
    const int i = wave0 + lane;
    const bool live = i < n;
    const int il = live ? i : n - 1;
    if (role == 0 && lane < 8) seq[lane] = 0;
    __syncthreads();                                                         // weights staged, sequence words cleared
    if (wave0 >= n) return;                                                  // a group without envs: all its waves leave

    if (role != 0) {
        // ------------------------------------------------------------------------------------ network wave(s)
        // the actor is on the serial chain of the step: let the SIMD's instruction arbiter prefer it; a critic-only wave
        // trails and takes what is left
#ifndef DPENV_WS_M_PRIO
#define DPENV_WS_M_PRIO 3
#endif
#ifndef DPENV_WS_NO_SETPRIO
        __builtin_amdgcn_s_setprio(DPENV_WS_M_PRIO);
#endif
        if constexpr (ROLES == 3) {
            const int tile = role - 1, env = tile * 32 + (lane & 31), hh = lane >> 5;
            int* s_mu = &seq[tile ? 6 : 1];
            int* s_v = &seq[tile ? 7 : 2];
            float ov[4];
            for (int t = 0; t <= pa.T; ++t) {
                ws_wait(&seq[0], t + 1);                                     // o_t posted (and step t-1's pre flag)
                const half8 in = tile_frag_from_mailbox<OD>(obs_mb, tile, lane);
                if (t < pa.T) {
                    mlp_eval_tile<KA>(Wpi, Bpi, pa.n_hidden, in, leak, ov);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * hh + j < A) mu_mb[env * 9 + 4 * hh + j] = ov[j];
                    ws_post(s_mu, t + 1, lane);                              // mu_t of this tile posted
                }
                mlp_eval_tile<KA>(Wv, Bv, pa.n_hidden, in, leak, ov);
                if (hh == 0) v_mb[(t & 1) * 64 + env] = ov[0];
                if (t > 0 && flag[(t - 1) & 1] != 0) {                       // step t-1 cut an episode that was re-drawn
                    const half8 pin = tile_frag_from_mailbox<OD>(pre_mb + ((t - 1) & 1) * (64 * 9), tile, lane);
                    mlp_eval_tile<KA>(Wv, Bv, pa.n_hidden, pin, leak, ov);
                    if (hh == 0) vpre_mb[(t & 1) * 64 + env] = ov[0];
                }
                ws_post(s_v, t + 1, lane);                                   // V(o_t) (and V of the pre-reset o_t) of this tile posted
            }
            return;
        }
        const bool do_actor = true, do_critic = true;
        half8 in0, in1;
        float o[9], outv[8];
        const bool draw_m = M_NOISE && pa.noise == nullptr && pa.sample != 0;
        uint32_t nctr_m = draw_m ? a.noise_ctr[il] : 0u;
        float xin[A];                                                        // xi of the step whose observation is awaited
        if (draw_m) { policy_noise<A>(a, a.env_id_base + i, nctr_m, xin); ++nctr_m; }
        uint64_t w_obs = 0, t_act = 0, t_cri = 0; const uint64_t t_start = __builtin_amdgcn_s_memtime(); (void)t_start; (void)w_obs; (void)t_act; (void)t_cri;
        auto frags_from = [&](const float* mb, half8& f0, half8& f1) {
#pragma unroll
            for (int k = 0; k < 9; ++k) o[k] = k < OD ? mb[lane * 9 + k] : 0.0f;
            obs_to_frags<OD>(o, f0, f1);
        };
        for (int t = 0; t <= pa.T; ++t) {
            if (!do_critic && t == pa.T) break;                              // the actor is not asked for mu_T
            WS_WAIT_T(w_obs, &seq[0], t + 1);                                // o_t posted (and step t-1's pre flag)
            frags_from(obs_mb + (t & (OBS_SLOTS - 1)) * (64 * 9), in0, in1);
            if (draw_m && t < pa.T) {                                        // o_t is in registers: its rows now carry xi_t
#pragma unroll
                for (int k = 0; k < A; ++k) xi_mb[lane * 9 + k] = xin[k];
            }
            if (do_actor && t < pa.T) {
                WS_TIC(ta_);
#ifdef DPENV_WS_M_PRIO_CRITIC
                __builtin_amdgcn_s_setprio(DPENV_WS_M_PRIO);
#endif
                WS_EVAL(Wpi, Bpi, in0, in1);
#pragma unroll
                for (int k = 0; k < A; ++k) mu_mb[lane * 9 + k] = outv[k];
                ws_post(&seq[1], t + 1, lane);                               // mu_t posted
#ifdef DPENV_WS_M_PRIO_CRITIC
                __builtin_amdgcn_s_setprio(DPENV_WS_M_PRIO_CRITIC);
#endif
                WS_TOC(t_act, ta_);
            }
            if (do_critic) {
                WS_TIC(tc_);
                WS_EVAL(Wv, Bv, in0, in1);
                v_mb[(t & 1) * 64 + lane] = outv[0];
                if (t > 0 && flag[(t - 1) & 1] != 0) {                       // step t-1 cut an episode that was re-drawn
                    half8 p0, p1;
                    frags_from(pre_mb + ((t - 1) & 1) * (64 * 9), p0, p1);
                    WS_EVAL(Wv, Bv, p0, p1);
                    vpre_mb[(t & 1) * 64 + lane] = outv[0];
                }
                ws_post(&seq[2], t + 1, lane);                               // V(o_t) (and V of the pre-reset o_t) posted
                WS_TOC(t_cri, tc_);
            }
            if (draw_m && t + 1 < pa.T) { policy_noise<A>(a, a.env_id_base + i, nctr_m, xin); ++nctr_m; }   // while the env wave steps
        }
#ifdef DPENV_WS_PROFILE
        if (live && pa.T >= 10 && do_actor) {
            (pa.logp + (int64_t)3 * n)[(unsigned)i] = (float)w_obs; (pa.logp + (int64_t)4 * n)[(unsigned)i] = (float)(__builtin_amdgcn_s_memtime() - t_start);
            (pa.logp + (int64_t)5 * n)[(unsigned)i] = (float)t_act; (pa.logp + (int64_t)6 * n)[(unsigned)i] = (float)t_cri;
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------------------------- E-wave
#ifdef DPENV_WS_E_PRIO
    __builtin_amdgcn_s_setprio(DPENV_WS_E_PRIO);
#endif
    Env s;
    Current cur = {0.0f, 0.0f, 0.0f, 0.0f, 0u};
    float vc0 = 0.0f, beta0 = 0.0f;
    bool ep_dirty = false, rf_dirty = (MODE == MODE_FULL);
    float o[9];
    const Vessel ve = launch_vessel(a, il);                                  // in VGPRs: this wave has them to spare
    const PolicyConsts<A> pc = load_policy_consts<A>(pa);
    const bool draw = pa.noise == nullptr && pa.sample != 0;
    uint32_t nctr = draw ? a.noise_ctr[il] : 0u;
    const int64_t stride_a = (int64_t)n * A, stride_o = (int64_t)n * OD;
    const int64_t w_a = (int64_t)wave0 * A, w_o = (int64_t)wave0 * OD;
    const int64_t rem_a = stride_a - w_a, rem_o = stride_o - w_o;
    float pre[A];
    load_env(a, il, s);
    sincos_lean(s.psi, s.sn, s.cs);
    if (a.cur_vc) {
        cur.vc = a.cur_vc[il]; cur.beta = a.cur_beta[il];
        if (a.current_drift) { vc0 = a.cur_vc0[il]; beta0 = a.cur_beta0[il]; cur.ctr = a.drift_ctr[il]; }
        current_components(cur);
    }
    uint32_t episode = a.auto_reset ? (uint32_t)a.episode[il] : 0u;
    {
        float sr_, cr_;
        bool same_;
        make_obs(s.N, s.E, s.psi, s.u, s.v, s.r, s.refN, s.refE, s.refPsi, s.pt, a.wrap_mode == WRAP_REFERENCE, o, sr_, cr_, same_);
    }
#pragma unroll
    for (int k = 0; k < OD; ++k) obs_mb[lane * 9 + k] = o[k];               // parity 0
    ws_post(&seq[0], 1, lane);                                               // o_0 posted
    if (pa.noise) load_rows<A, 64>(pa.noise + w_a, rem_a, lane, pre);
    int next_switch = 0;
    uint64_t w_mu = 0, w_v = 0, t_env = 0, t_noi = 0; const uint64_t t_start = __builtin_amdgcn_s_memtime(); (void)t_start; (void)w_mu; (void)w_v; (void)t_env; (void)t_noi;
    uint64_t t_pre = 0, t_post = 0, t_off = 0; (void)t_pre; (void)t_post; (void)t_off;
    bool boot_wanted = false, was_reset = false;                             // of the step whose boot row is still owed
    // Between "mu_t has arrived" and "o_t+1 is posted" the env wave is on the serial chain of the rollout (the network wave waits
    // for that observation), so only what o_t+1 needs is done there: a_t = mu_t + std xi_t, env.step, the reset of finished
    // envs.  The rows of step t (action, log-likelihood, reward, done, the observation row of t+1) are written after the
    // hand-over, while the network wave evaluates mu_t+1.
    wave_store_rows<OD>(lds_io, pa.obs_out, w_o, rem_o, o, lane, a.obs_bf16 != 0);
    for (int t = 0; t < pa.T; ++t) {
        const bool q_boot_wanted = boot_wanted, q_was_reset = was_reset;     // flags of step t-1
        // the exploration noise of this step does not depend on the actor's answer: it is drawn while the network wave is
        // still evaluating mu_t (the env wave would otherwise only poll)
        float xi[A];
        WS_TIC(tn_);
        if (pa.noise) {
            wave_rows_from_regs<A>(lds_io, pre, xi, lane);
            if (t + 1 < pa.T) load_rows<A, 64>(pa.noise + (int64_t)(t + 1) * stride_a + w_a, rem_a, lane, pre);
        } else if (draw) {
            if (!M_NOISE) policy_noise<A>(a, a.env_id_base + i, nctr, xi);
            ++nctr;
        }
        WS_TOC(t_noi, tn_);
        WS_WAIT_T(w_mu, &seq[1], t + 1);                                     // mu_t posted
        if (ROLES == 3) ws_wait(&seq[6], t + 1);
        WS_TIC(tp_);
#ifdef DPENV_WS_DYN_PRIO
        __builtin_amdgcn_s_setprio(3);                                       // on the chain until o_t+1 is posted
#endif
        float act[A], mu[A];
        float logp;
#pragma unroll
        for (int k = 0; k < A; ++k) mu[k] = mu_mb[lane * 9 + k];
        if (M_NOISE && draw) {
#pragma unroll
            for (int k = 0; k < A; ++k) xi[k] = xi_mb[lane * 9 + k];
        }
#pragma unroll
        for (int k = 0; k < A; ++k) act[k] = (pa.noise || draw) ? fmaf(pc.std[k], xi[k], mu[k]) : mu[k];      // core.py:85
        bool has_ref = false;
        float nrN = 0.0f, nrE = 0.0f, nrP = 0.0f;
        if (next_switch < pa.n_switch && pa.switch_step[next_switch] == t) {
            const float* rp = pa.refs + (int64_t)next_switch * 3 * n;
            nrN = rp[il]; nrE = rp[(int64_t)n + il]; nrP = rp[2 * (int64_t)n + il];
            has_ref = true; rf_dirty = true;
            ++next_switch;
        }
        StepOut out;
#ifdef DPENV_WS_SELFCHECK
        // Diagnostic build only (tools/ws_selfcheck.py): the step is evaluated a second time from opaque copies of the same
        // inputs AFTER the partner wave has finished its critic (it then only polls), and every field of the two results
        // is compared bit for bit.  Both evaluations are the same deterministic IEEE arithmetic, so a difference is a
        // transient fault of the first evaluation (the one that runs beside the partner's MFMAs) - and the record says
        // which quantity, which lane, by how much.
        Env sB = s;
        const float pre6[6] = {s.N, s.E, s.psi, s.u, s.v, s.r};
        float actB[A];
#pragma unroll
        for (int k = 0; k < A; ++k) { actB[k] = act[k]; asm volatile("" : "+v"(actB[k])); }
        asm volatile("" : "+v"(sB.N), "+v"(sB.E), "+v"(sB.psi), "+v"(sB.u), "+v"(sB.v), "+v"(sB.r), "+v"(sB.sn), "+v"(sB.cs));
        asm volatile("" : "+v"(sB.refN), "+v"(sB.refE), "+v"(sB.refPsi), "+v"(sB.pt[0]), "+v"(sB.pt[1]), "+v"(sB.pt[2]),
                          "+v"(sB.ang[0]), "+v"(sB.ang[1]), "+v"(sB.ang[2]), "+v"(sB.steps));
#endif
        WS_TOC(t_pre, tp_);
        WS_TIC(te_);
        env_step<MODE, EXT>(a, ve, s, act, has_ref, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, out);
        WS_TOC(t_env, te_);
        WS_TIC(tq_);
#ifdef DPENV_WS_SELFCHECK
        {
            ws_wait(&seq[2], t + 1);                                         // critic(o_t) done: the partner is idle from here
            if (ROLES == 3) ws_wait(&seq[7], t + 1);
            StepOut outB;
            env_step<MODE, EXT>(a, ve, sB, actB, has_ref, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, outB);
            const float fa[20] = {s.N, s.E, s.psi, s.u, s.v, s.r, s.sn, s.cs, out.reward, out.o[0], out.o[1], out.o[2], out.o[3],
                                  out.o[4], out.o[5], out.o[6], out.o[7], out.o[8], __uint_as_float(out.d), s.ang[1]};
            const float fb[20] = {sB.N, sB.E, sB.psi, sB.u, sB.v, sB.r, sB.sn, sB.cs, outB.reward, outB.o[0], outB.o[1], outB.o[2],
                                  outB.o[3], outB.o[4], outB.o[5], outB.o[6], outB.o[7], outB.o[8], __uint_as_float(outB.d), sB.ang[1]};
            uint32_t mask = 0;
#pragma unroll
            for (int k = 0; k < 20; ++k) mask |= (__float_as_uint(fa[k]) != __float_as_uint(fb[k])) ? (1u << k) : 0u;
            if (mask != 0u && pa.dbg != nullptr) {
                const uint32_t slot = atomicAdd(pa.dbg, 1u);
                if (slot < 2000u) {
                    uint32_t* rec = pa.dbg + 4 + (size_t)slot * 60;
                    rec[0] = (uint32_t)i; rec[1] = (uint32_t)t; rec[2] = mask; rec[3] = (uint32_t)lane;
#pragma unroll
                    for (int k = 0; k < 20; ++k) { rec[4 + k] = __float_as_uint(fa[k]); rec[24 + k] = __float_as_uint(fb[k]); }
#pragma unroll
                    for (int k = 0; k < 6; ++k) rec[44 + k] = __float_as_uint(pre6[k]);
#pragma unroll
                    for (int k = 0; k < A; ++k) rec[50 + k] = __float_as_uint(actB[k]);
                }
            }
        }
#endif
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = out.o[k];
        const bool do_reset = a.auto_reset && out.d != 0u && live;
        const bool terminal = (out.d & DONE_TERMINAL) != 0u;
        const bool ended = (out.d != 0u) || (t == pa.T - 1);
        boot_wanted = ended && !terminal;                                    // ppo.py:311
        was_reset = do_reset;
        // the critic is owed the PRE-reset observation only where a cut (not terminated) episode is re-drawn
        const bool post_pre = __ballot(do_reset && boot_wanted) != 0ull;
        if (post_pre) {
            float* pm = pre_mb + (t & 1) * (64 * 9);
#pragma unroll
            for (int k = 0; k < OD; ++k) pm[lane * 9 + k] = o[k];
        }
        if (lane == 0) flag[t & 1] = post_pre ? 1 : 0;
        if (__ballot(do_reset) != 0ull) {
            if (do_reset) {
                env_auto_reset<MODE>(a, s, a.env_id_base + i, episode, o);
                ++episode; ep_dirty = true; rf_dirty = true;
            }
        }
        // o_{t+1} replaces o_t in the mailbox: the network wave read o_t right after it saw seq[0] = t + 1 and BEFORE it posted
        // mu_t, which this wave has waited for.  (Three roles: o_{t+1} goes into the slot of its parity, which last held o_{t-1};
        // the actor read that before posting mu_{t-1} and the critic before posting V(o_{t-1}), both waited for in step t-1.)
        {
            float* om = obs_mb + ((t + 1) & (OBS_SLOTS - 1)) * (64 * 9);
#pragma unroll
            for (int k = 0; k < OD; ++k) om[lane * 9 + k] = o[k];           // the next policy input
        }
        ws_post(&seq[0], t + 2, lane);                                       // o_{t+1} (and this step's pre flag) posted
        WS_TOC(t_post, tq_);
        WS_TIC(tr_);
#ifdef DPENV_WS_DYN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);                                   // nothing of the rows below moves up into the chain
        logp = action_logp<A>(pc, mu, act);                                  // core.py:42-46 on (a_t, mu_t)
        if (a.current_drift) current_drift_step(a, cur, vc0, beta0, a.env_id_base + i);   // the current of step t+1: not needed by o_t+1
        wave_store_rows<A>(lds_io, pa.act_out, (int64_t)t * stride_a + w_a, rem_a, act, lane);
        if (t + 1 < pa.T) wave_store_rows<OD>(lds_io, pa.obs_out, (int64_t)(t + 1) * stride_o + w_o, rem_o, o, lane, a.obs_bf16 != 0);
        if (live) {
            (pa.rew + (int64_t)t * n)[(unsigned)i] = out.reward;
            (pa.done + (int64_t)t * n)[(unsigned)i] = (uint8_t)out.d;
            (pa.logp + (int64_t)t * n)[(unsigned)i] = logp;
        }
        WS_WAIT_T(w_v, &seq[2], t + 1);                                      // V(o_t), V(pre-reset o_t) posted
        if (ROLES == 3) ws_wait(&seq[7], t + 1);
        WS_TOC(t_off, tr_);
        if (live) {
            const float v_t = v_mb[(t & 1) * 64 + lane];
            (pa.val + (int64_t)t * n)[(unsigned)i] = v_t;
            if (t > 0) (pa.boot + (int64_t)(t - 1) * n)[(unsigned)i] = q_boot_wanted ? (q_was_reset ? vpre_mb[(t & 1) * 64 + lane] : v_t) : 0.0f;
        }
    }
    ws_wait(&seq[2], pa.T + 1);                                              // V(o_T) posted
    if (ROLES == 3) ws_wait(&seq[7], pa.T + 1);
#ifdef DPENV_WS_PROFILE
    if (live && pa.T >= 12) {
        (pa.logp + (int64_t)0 * n)[(unsigned)i] = (float)w_mu; (pa.logp + (int64_t)1 * n)[(unsigned)i] = (float)w_v; (pa.logp + (int64_t)2 * n)[(unsigned)i] = (float)(__builtin_amdgcn_s_memtime() - t_start);
        (pa.logp + (int64_t)7 * n)[(unsigned)i] = (float)t_env; (pa.logp + (int64_t)8 * n)[(unsigned)i] = (float)t_noi;
        (pa.logp + (int64_t)9 * n)[(unsigned)i] = (float)t_pre; (pa.logp + (int64_t)10 * n)[(unsigned)i] = (float)t_post;
        (pa.logp + (int64_t)11 * n)[(unsigned)i] = (float)t_off;
    }
#endif
    wave_store_rows<OD>(lds_io, pa.last_obs, w_o, rem_o, o, lane, a.obs_bf16 != 0);
    if (live) {
        const float v_T = v_mb[(pa.T & 1) * 64 + lane];
        (pa.boot + (int64_t)(pa.T - 1) * n)[(unsigned)i] = boot_wanted ? (was_reset ? vpre_mb[(pa.T & 1) * 64 + lane] : v_T) : 0.0f;
        pa.last_val[i] = v_T;
        store_env(a, i, s, rf_dirty);
        if (ep_dirty) a.episode[i] = (int)episode;
        if (a.current_drift) { a.cur_vc[i] = cur.vc; a.cur_beta[i] = cur.beta; a.drift_ctr[i] = cur.ctr; }
        if (draw) a.noise_ctr[i] = nctr;
    }
}

#ifdef DPENV_WS_SELFCHECK
#include "dpenv_diag.inc"      // diagnostic builds only: pk_probe_kernel (tools/ws_pk_probe.py)
#endif

// =============================================================================================
//  Weight packing ON THE DEVICE: fp32 dense kernels W[l][in][out] / biases (the reference's variable layout, core.py:29-33)
//  -> the LDS image the evaluation kernels stage (MFMA A-operand fragments, bias tiles, std / logp constants).
//  One thread per f16 fragment element.  Fragment f, lane l = (r = l & 31, h = l >> 5), element j holds
//  W^T[out row 32 mo + r][input slot k(f, h, j)]:
//    first layer : k = 8 h + j                                   (slots 0..in-1 = inputs, slot 15 = the layer's bias)
//    later layers: k = 32 mt + 16 s + 8 (j >> 2) + 4 h + (j & 3)  (the accumulator-as-operand order), ks = 2 mt + s;
//                  their biases go into `bias` as accumulator-layout tiles: block b, half h, register r16
//                  -> b[out row 32 mo + 8 (r16 >> 2) + 4 h + (r16 & 3)]
//  split: a second image of the LOW parts, lo = f16(w - f32(f16(w))), behind the first (dpenv_policy_dev.h, mlp_eval_x).
//  Stream-ordered, no host round trip: a PPO loop re-packs after every update without synchronising (ppo.py:260-280).
// =============================================================================================
__device__ __forceinline__ float pack_weight(const PackNet& m, int ks_n, int f, int lane, int j)
{
    const int r = lane & 31, hh = lane >> 5, H = m.H, nh = m.n_layers - 1;
    if (f < 3) {                                                        // first layer
        const int k = 8 * hh + j, row = 32 * f + r;
        if (row >= H) return 0.0f;
        return k < m.in_dim ? m.W[0][(size_t)k * H + row] : (k == 15 ? m.b[0][row] : 0.0f);
    }
    const int idx = f - 3, per_layer = 3 * ks_n;
    if (idx < per_layer * (nh - 1)) {                                   // hidden -> hidden
        const int l = 1 + idx / per_layer, rem = idx % per_layer, mo = rem / ks_n, ks = rem % ks_n;
        const int fk = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * hh + (j & 3), row = 32 * mo + r;
        return (row < H && fk < H) ? m.W[l][(size_t)fk * H + row] : 0.0f;
    }
    const int ks = idx - per_layer * (nh - 1);                          // output layer
    const int fk = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * hh + (j & 3);
    return (r < m.out_dim && fk < H) ? m.W[m.n_layers - 1][(size_t)fk * m.out_dim + r] : 0.0f;
}

__global__ __launch_bounds__(256) void pack_policy_kernel(const PackNet pi, const PackNet v, const float* log_std, int adim, int ks_n,
                                                          int nfrag, int nblk, int split, _Float16* frags, float* bias, float* consts)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int per_net = nfrag * 64 * 8;
    if (tid < 2 * per_net) {
        const int net = tid / per_net, e = tid % per_net, f = e / 512, lane = (e >> 3) & 63, j = e & 7;
        const float w = pack_weight(net ? v : pi, ks_n, f, lane, j);
        const _Float16 hi = (_Float16)w;                                // round to nearest even
        frags[tid] = hi;
        if (split) frags[2 * per_net + tid] = (_Float16)(w - (float)hi);
    }
    if (tid < 2 * nblk * 32) {
        const int net = tid / (nblk * 32), e = tid % (nblk * 32), blk = e / 32, hh = (e >> 4) & 1, r16 = e & 15;
        const PackNet& m = net ? v : pi;
        const int nh = m.n_layers - 1, rr = 8 * (r16 >> 2) + 4 * hh + (r16 & 3);
        float b;
        if (blk < 3 * (nh - 1)) {
            const int l = 1 + blk / 3, row = 32 * (blk % 3) + rr;
            b = row < m.H ? m.b[l][row] : 0.0f;
        } else {
            b = rr < m.out_dim ? m.b[m.n_layers - 1][rr] : 0.0f;
        }
        bias[tid] = b;
    }
    if (tid < 8) {
        const float ls = tid < adim ? log_std[tid] : 0.0f;
        const float sd = expf(ls);
        consts[tid] = sd;                                               // core.py:84
        consts[8 + tid] = 1.0f / (sd + 1e-8f);                          // core.py:45, EPS = 1e-8
        consts[16 + tid] = tid < adim ? (-ls - 0.5f * logf(2.0f * 3.14159265358979323846f)) : 0.0f;
    }
}

}  // namespace dpenv

using namespace dpenv;

#ifdef DPENV_WS_SELFCHECK
#define DPENV_DIAG_LAUNCHERS
#include "dpenv_diag.inc"
#endif

extern "C" hipError_t dpenv_dev_launch_pack_policy(const PackNet* pi, const PackNet* v, const float* log_std, int adim, int ks, int nfrag,
                                                   int nblk, int split, void* frags, float* bias, float* consts, hipStream_t s)
{
    const int total = 2 * nfrag * 64 * 8;
    hipLaunchKernelGGL(pack_policy_kernel, dim3((total + 255) / 256), dim3(256), 0, s, *pi, *v, log_std, adim, ks, nfrag, nblk, split,
                       (_Float16*)frags, bias, consts);
    return hipGetLastError();
}

static size_t policy_lds_bytes(const PolicyArgs& pa)
{
    return (size_t)2 * pa.nfrag * 64 * 16 + (size_t)2 * pa.nblk * 32 * 4 + (size_t)PWAVES * 64 * 9 * 4;
}

extern "C" hipError_t dpenv_dev_launch_policy_forward(const PolicyArgs* pa, int od, int adim, const float* obs, float* mu,
                                                      float* v, int n, hipStream_t s)
{
    const dim3 grid((n + PBLOCK - 1) / PBLOCK), block(PBLOCK);
    const size_t lds = policy_lds_bytes(*pa);
#define FWD_K(OD_, A_, KS_)                                                                                              \
    do {                                                                                                                 \
        hipError_t e = hipFuncSetAttribute((const void*)policy_forward_kernel<OD_, A_, KS_>,                              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
        if (e != hipSuccess) return e;                                                                                   \
        hipLaunchKernelGGL((policy_forward_kernel<OD_, A_, KS_>), grid, block, lds, s, *pa, obs, mu, v, n);               \
        return hipGetLastError();                                                                                        \
    } while (0)
#define FWD(OD_, A_)                                                                                                     \
    do {                                                                                                                 \
        if (ka == 5) FWD_K(OD_, A_, 5);                                                                                  \
        if (ka == 6) FWD_K(OD_, A_, 6);                                                                                  \
        if (ka == 21) FWD_K(OD_, A_, 21);                                                                                \
        FWD_K(OD_, A_, 22);                                                                                              \
    } while (0)
    if ((pa->ks != 5 && pa->ks != 6) || (pa->act != 0 && pa->act != 1)) return hipErrorInvalidValue;
    const int ka = pa->ks + 16 * pa->act;
    if (od == 9 && adim == 7) FWD(9, 7);
    if (od == 9 && adim == 5) FWD(9, 5);
    if (od == 9 && adim == 6) FWD(9, 6);
    if (od == 6 && adim == 7) FWD(6, 7);
    if (od == 6 && adim == 5) FWD(6, 5);
    if (od == 6 && adim == 6) FWD(6, 6);
    if (od == 6 && adim == 3) FWD(6, 3);
#undef FWD_K
#undef FWD
    return hipErrorInvalidValue;
}

template <int MODE, bool EXT, int KA>
static hipError_t launch_policy_rollout_one(const StepArgs& a, const PolicyArgs& pa, hipStream_t s)
{
    if (pa.ws) {
        // two waves per 64 envs: 512-thread workgroups of 256 envs, one LDS image of the weights + four mailbox groups
        const dim3 grid((a.n + 255) / 256);
#ifdef DPENV_WS3
        constexpr int ROLES = 3;
        const size_t lds = (size_t)2 * pa.nfrag * 64 * 16 + (size_t)2 * pa.nblk * 32 * 4 + (size_t)4 * WS_GROUP_FLOATS * 4;
#else
        constexpr int ROLES = 2;
        const size_t lds = (size_t)2 * pa.nfrag * 64 * 16 + (size_t)2 * pa.nblk * 32 * 4 + (size_t)4 * WS_GROUP_FLOATS * 4;
#endif
        hipError_t e = hipFuncSetAttribute((const void*)policy_rollout_ws_kernel<MODE, EXT, KA, ROLES>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((policy_rollout_ws_kernel<MODE, EXT, KA, ROLES>), grid, dim3(256 * ROLES), lds, s, a, pa);
        return hipGetLastError();
    }
    const dim3 grid((a.n + PBLOCK - 1) / PBLOCK), block(PBLOCK);
    const size_t lds = policy_lds_bytes(pa);
    hipError_t e = hipFuncSetAttribute((const void*)policy_rollout_kernel<MODE, EXT, KA>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((policy_rollout_kernel<MODE, EXT, KA>), grid, block, lds, s, a, pa);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_policy_rollout_mode(const StepArgs& a, const PolicyArgs& pa, bool ext, hipStream_t s)
{
    if ((pa.ks != 5 && pa.ks != 6) || (pa.act != 0 && pa.act != 1)) return hipErrorInvalidValue;
    switch (pa.ks + 16 * pa.act) {
    case 5: return ext ? launch_policy_rollout_one<MODE, true, 5>(a, pa, s) : launch_policy_rollout_one<MODE, false, 5>(a, pa, s);
    case 6: return ext ? launch_policy_rollout_one<MODE, true, 6>(a, pa, s) : launch_policy_rollout_one<MODE, false, 6>(a, pa, s);
    case 21: return ext ? launch_policy_rollout_one<MODE, true, 21>(a, pa, s) : launch_policy_rollout_one<MODE, false, 21>(a, pa, s);
    default: return ext ? launch_policy_rollout_one<MODE, true, 22>(a, pa, s) : launch_policy_rollout_one<MODE, false, 22>(a, pa, s);
    }
}

extern "C" hipError_t dpenv_dev_launch_policy_rollout(const StepArgs* a, const PolicyArgs* pa, int mode, int ext,
                                                      hipStream_t s)
{
    switch (mode) {
    case MODE_FULL: return launch_policy_rollout_mode<MODE_FULL>(*a, *pa, ext, s);
    case MODE_SIMPLE: return launch_policy_rollout_mode<MODE_SIMPLE>(*a, *pa, ext, s);
    case MODE_LIMITED: return launch_policy_rollout_mode<MODE_LIMITED>(*a, *pa, ext, s);
    case MODE_FINAL_WRAP: return launch_policy_rollout_mode<MODE_FINAL_WRAP>(*a, *pa, ext, s);
    case MODE_FINAL_CONT: return launch_policy_rollout_mode<MODE_FINAL_CONT>(*a, *pa, ext, s);
    }
    return hipErrorInvalidValue;
}
