// dpenv_policy_ws.h - the closed-loop rollout with the work of every 64 envs split over two waves (an env wave and a network
// wave per SIMD), for all three network arithmetics.  Included by dpenv_policy.hip (PREC_F16) and dpenv_policy_xws*.hip
// (PREC_F32, PREC_F32_ACTOR: one translation unit each, they take minutes to compile).
#ifndef DPENV_POLICY_WS_H
#define DPENV_POLICY_WS_H
#include <type_traits>
#include "dpenv_policy_dev.h"

namespace dpenv {

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
//  policy_rollout_kernel: every row is bit-identical.  (ROLES = 2: an env wave and a network wave.  Two three-wave forms - a critic wave
//  of its own; the network wave split by env tile - were built and measured in round 2, were slower, and are gone: DESIGN.md section 4.)
// =============================================================================================
//  Arithmetic (PREC): PREC_F16 - the fast mode; PREC_F32 - actor and critic in the split-f16 arithmetic of mlp_eval_x (the network
//  wave reads a second, LOW image of the weights); PREC_F32_ACTOR - actor split, critic plain f16 on the high image.  The split
//  modes have no LDS left for the env wave's row staging: their rows go out per lane (store_row_direct), the mailbox group is
//  2 304 bytes smaller.  Rows are bit-identical to the one-wave kernels of the same arithmetic (tests).
//  Geometry (GROUPS): 4 - a 512-thread workgroup of 256 envs, an env and a network wave on every SIMD (65 536 envs fill the chip);
//  2 - a 256-thread workgroup of 128 envs, ONE wave per SIMD: for batches that leave SIMDs free (n <= 128 x CUs, e.g. config 4's
//  32 768 envs per GPU) neither wave shares its issue port.
constexpr int WSBLOCK = 512;
constexpr int WS_GROUP_FLOATS = 64 * 9 * 5 + 64 * 4 + 64;       // io | obs | pre[2] | mu | v[2] | vpre[2] | sequence words (+ pad)
constexpr int WS_GROUP_FLOATS_X = 64 * 9 * 4 + 64 * 4 + 64;     // the same without the io area
static_assert(4 * WS_GROUP_FLOATS * 4 == POLICY_WS_MAILBOX_BYTES, "dpenv_dev.h: POLICY_WS_MAILBOX_BYTES out of step with the mailbox layout");
static_assert(4 * WS_GROUP_FLOATS_X * 4 == POLICY_WS_MAILBOX_X_BYTES, "dpenv_dev.h: POLICY_WS_MAILBOX_X_BYTES out of step with the mailbox layout");
__host__ __device__ constexpr int ws_images(int prec) { return prec == PREC_F16 ? 2 : (prec == PREC_F32_ACTOR ? 3 : 4); }

// pair-level hand-over inside a workgroup: a sequence word in LDS, released by one wave and acquired by its partner.
// Both waves of a pair belong to the same workgroup, so they are always co-resident; the waiter sleeps between polls.
// Every lane's mailbox rows are released by a workgroup-scope fence that ALL lanes execute (the sequence word is stored by lane 0 alone);
// the waiter acquires with a fence after its poll loop.  On gfx950 (waves of a workgroup share a CU, no tgsplit) the release fence is the
// s_waitcnt lgkmcnt(0) that used to sit inside the lane-0 branch, the acquire fence emits nothing: same instructions, but ordered by the
// memory model instead of by in-order LDS issue and the compiler's good will.
__device__ __forceinline__ void ws_post(int* p, int v, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ws_wait(int* p, int v)
{
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < v)
        __builtin_amdgcn_s_sleep(2);       // x 64 cycles between polls (A/B'd in round 3: 0 / 1 / 2 / 4 within noise in both geometries)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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

#define WS_EVAL(W_, B_, f0_, f1_) mlp_eval<KA>(W_, B_, pa.n_hidden, f0_, f1_, leak, outv)
//  RND: the domain randomisation's hull re-draw compiled into the reset branch (instantiated for the shipped training configuration only -
//  final variant, continuous angles, extended state, leaky-relu - see dpenv_ws_launch::pick and dpenv_env_dev.h redraw_vessel_cold).
//  SLOSS (round 6): the SHARED training form - the single class's thrust-loss coefficients from the kernel arguments (StepArgs.kl, LOSS_SHARED;
//  zeros for a hull without a loss: the rows of the default kernels bit for bit) and the per-episode current re-draw: the thrust-loss preset and /
//  or dpenv_set_current_randomisation on the shared hull, without per-env blocks; same configurations as RND.
template <int MODE, bool EXT, int KA, int ROLES, int PREC = PREC_F16, int GROUPS = 4, bool RND = false, bool SLOSS = false>
__global__ __launch_bounds__(64 * GROUPS * ROLES) void policy_rollout_ws_kernel(const StepArgs a, const PolicyArgs pa)
{
    static_assert(!(RND && SLOSS), "per-env blocks carry their own coefficients");
    constexpr int IL = SLOSS ? IL_SHARED : IL_NONE;
    constexpr bool CURR = RND || SLOSS;            // the forms that re-draw the current with the episode
    constexpr int A = ModeTraits<MODE>::A;
    constexpr int OD = EXT ? 9 : 6;
    constexpr int THREADS = 64 * GROUPS * ROLES;
    constexpr bool SPLIT = PREC != PREC_F16;
    // env-wave rows through LDS transposes (else per lane).  Not in the three-role form: its env wave shares a SIMD with the critic wave and has
    // 256 registers, not 512 - with the staging code the f16 env wave spilled 116 B per lane there (round 4; without it 243 registers, no
    // scratch - and still 1-3 % slower than two roles for f16, so DPENV_WS_CRITIC_WAVE leaves f16 out: profiles/r04_critic_wave.txt)
    constexpr bool STAGE = !SPLIT && ROLES == 2;
    constexpr int NIMG = ws_images(PREC);           // weight images staged: pi_hi, v_hi (, pi_lo (, v_lo))
    // All-exact arithmetic with a SIMD per wave (GROUPS = 2): the CRITIC runs on the ENV wave (round 3).  Two exact evaluations one after
    // the other in the network wave bound the step at 10.8 us while the env wave idles for most of it; with the critic behind the env
    // wave's rows the two networks are evaluated at the same time on two SIMDs' matrix pipes (10.3 -> 9.7 us at 32 768 envs, same call).
    // Same mlp_eval_x on the same fragments: rows unchanged.  NOT for two waves per SIMD (GROUPS = 4), measured: there the critic's
    // MFMAs beside the actor's contend for ONE matrix pipe (today's order pairs matrix work with vector work: critic beside env.step,
    // rows beside the actor) and env state + evaluation do not fit 256 registers (10.0 -> 11.4 us exact actor, 13.5 -> 17.2 all exact,
    // with the vessel block and policy constants re-fetched per step and 268 / 412 B of scratch left); and not for the exact-actor
    // mode with its f16 critic, which gains nothing (7.36 vs 7.34 us).
#ifdef DPENV_WS_SELFCHECK
    constexpr bool ECRITIC_ON = false;  // the diagnostic double evaluation waits for the network wave's critic
#else
    constexpr bool ECRITIC_ON = true;
#endif
    constexpr bool ECRITIC = PREC == PREC_F32 && GROUPS == 2 && ROLES == 2 && ECRITIC_ON;
    // ROLES = 3 (round 4, 128-env workgroups only): a CRITIC WAVE of its own per 64 envs - six waves on the four SIMDs of a CU, in the
    // order E0 E1 A0 A1 C0 C1, so that the actor waves keep a SIMD each and a critic wave shares one with its env wave (matrix work beside
    // vector work, the pairing that nets; MI355X_MICROARCH.md "Two waves per SIMD").  V(o_t) is then evaluated while the actor wave
    // evaluates mu_t, by a wave whose registers hold nothing but the evaluation (the env wave's own copy - ECRITIC above - is compiled
    // around ~100 registers of env state).  The observation mailbox gets a second slot (by step parity): the critic may still be reading
    // o_t when the env wave posts o_t+1.  (Round 2's three-wave forms were 256-env workgroups with THREE waves on every SIMD: 168
    // registers per wave, spills, three streams per issue port - measured slower and removed.  Here no SIMD holds more than two.)
    static_assert(ROLES == 2 || (ROLES == 3 && GROUPS == 2), "an env wave and a network wave per 64 envs; a critic wave of its own only with a SIMD per wave to spare");
    static_assert(GROUPS == 4 || GROUPS == 2, "workgroups of 256 or 128 envs");
    extern __shared__ uint4 lds_dyn[];
    uint4* lds_w = lds_dyn;
    const int img_floats = NIMG * pa.nent * 4 + 2 * pa.nblk * 32;      // images | bias tiles, then the mailbox groups
    {
        const int total = NIMG * pa.nent;
        for (int k = threadIdx.x; k < total; k += THREADS) lds_w[k] = pa.frags[k];
        float* lb = (float*)(lds_w + total);
        for (int k = threadIdx.x; k < 2 * pa.nblk * 32; k += THREADS) lb[k] = pa.bias[k];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = wave / GROUPS;                 // 0 = env wave, 1 = network wave
    const int g = wave % GROUPS;
    constexpr int OBS_SLOTS = ROLES == 3 ? 2 : 1;
    float* grp = (float*)lds_dyn + img_floats + g * ((STAGE ? WS_GROUP_FLOATS : WS_GROUP_FLOATS_X) + (OBS_SLOTS - 1) * 64 * 9);
    float* lds_io = grp;                         // E-wave row staging (STAGE only)
    float* obs_mb = grp + (STAGE ? 64 * 9 : 0);  // [OBS_SLOTS][64][9] o_t (by step parity): one row of 9 per lane (stride 9 is conflict-free)
    float* pre_mb = obs_mb + 64 * 9 * OBS_SLOTS;   // [2][64][9] pre-reset observation of a cut episode, by step parity
    float* mu_mb = pre_mb + 64 * 9 * 2;          // actor mean, stride 9
    float* v_mb = mu_mb + 64 * 9;                // [2][64] V(o_t), by step parity
    float* vpre_mb = v_mb + 128;                 // [2][64] V(pre-reset o_t), by step parity
    int* seq = (int*)(vpre_mb + 128);            // [0] observations posted, [1] means posted, [2] values posted, [4..5] pre flags,
    int* flag = seq + 4;
    // Two roles: the NETWORK wave draws the exploration noise (Philox + Box-Muller, ~300 VALU per step) while it waits for the
    // next observation - with the noise in the env wave that wave was the busy one (tools/ws_profile.py).  xi_t travels in the
    // observation mailbox: once the network wave has turned o_t into fragments the rows are free until the env wave writes
    // o_t+1, which it does after it has waited for mu_t and read xi_t.
    // Split arithmetics: the ENV wave draws (it idles ~4 us per step while the network wave evaluates the actor) - unless it carries the critic
    // (ECRITIC): then the network wave (actor only) has the time.  f16 with a SIMD per wave: the network wave (two evaluations per step) is the
    // busy one there, the env wave draws.
    constexpr bool M_NOISE = (ROLES == 2) && ((!SPLIT && GROUPS == 4) || ECRITIC);
    float* xi_mb = obs_mb;
    const uint4* Wpi = lds_w;
    const uint4* Wv = lds_w + pa.nent;
    const uint4* Wpi_l = lds_w + 2 * pa.nent;                         // SPLIT only
    const uint4* Wv_l = lds_w + 3 * pa.nent;                          // PREC_F32 only
    const float* Bpi = (const float*)(lds_dyn + NIMG * pa.nent);
    const float* Bv = Bpi + pa.nblk * 32;
    const _Float16 leak = (_Float16)pa.leak;
    const int n = a.n;
    const int wave0 = blockIdx.x * (64 * GROUPS) + g * 64;
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
        // (the critic wave of the three-role form shares its SIMD with the env wave; same-call A/B at 32 768 envs, all exact: priority 0 8.4 us
        // per step, 1 8.05, 2 7.85-8.1, 3 8.1 - profiles/r04_critic_wave.txt)
        if (ROLES == 3 && role == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(3);
        // one copy of the loop per role, so that a wave's registers hold only what ITS evaluation needs (with the roles as run-time flags
        // the three-role kernel kept the union of both and spilled)
        auto net_wave = [&](auto ACT_, auto CRI_) __attribute__((always_inline)) {
        constexpr bool do_actor = decltype(ACT_)::value, do_critic = decltype(CRI_)::value;
        half8 in0, in1;                                                      // PREC_F16: first-layer fragments of o_t
        SplitIn inx;                                                         // SPLIT: their high and low parts
        float o[9], outv[8];
        const bool draw_m = M_NOISE && pa.noise == nullptr && pa.sample != 0;
        uint32_t nctr_m = draw_m ? a.noise_ctr[il] : 0u;
        float xin[A];                                                        // xi of the step whose observation is awaited
        if (draw_m) { policy_noise<A>(a, a.env_id_base + i, nctr_m, xin); ++nctr_m; }
        uint64_t w_obs = 0, t_act = 0, t_cri = 0; const uint64_t t_start = __builtin_amdgcn_s_memtime(); (void)t_start; (void)w_obs; (void)t_act; (void)t_cri;
        auto row_from = [&](const float* mb) {
#pragma unroll
            for (int k = 0; k < 9; ++k) o[k] = k < OD ? mb[lane * 9 + k] : 0.0f;
        };
        auto frags_from = [&](const float* mb, half8& f0, half8& f1) {
            row_from(mb);
            obs_to_frags<OD>(o, f0, f1);
        };
        // the critic in this launch's arithmetic: split like the actor (PREC_F32), or plain f16 on the high image and the high parts
        // of the input (PREC_F32_ACTOR: exactly the f16 mode's critic)
        auto critic_x = [&](const SplitIn& f) __attribute__((always_inline)) {
            if constexpr (PREC == PREC_F32) mlp_eval_x<KA>(Wv, Wv_l, Bv, pa.n_hidden, f, pa.leak, outv);
            else mlp_eval<KA>(Wv, Bv, pa.n_hidden, f.h0, f.h1, leak, outv);
        };
        for (int t = 0; t <= pa.T; ++t) {
            if (!do_critic && t == pa.T) break;                              // the actor is not asked for mu_T
            WS_WAIT_T(w_obs, &seq[0], t + 1);                                // o_t posted (and step t-1's pre flag)
            if constexpr (SPLIT) { row_from(obs_mb + (t & (OBS_SLOTS - 1)) * (64 * 9)); obs_to_frags_x<OD>(o, inx); }
            else frags_from(obs_mb + (t & (OBS_SLOTS - 1)) * (64 * 9), in0, in1);
            if (draw_m && t < pa.T) {                                        // o_t is in registers: its rows now carry xi_t
#pragma unroll
                for (int k = 0; k < A; ++k) xi_mb[lane * 9 + k] = xin[k];
            }
            if (do_actor && t < pa.T) {
                WS_TIC(ta_);
                if constexpr (SPLIT) mlp_eval_x<KA>(Wpi, Wpi_l, Bpi, pa.n_hidden, inx, pa.leak, outv);
                else WS_EVAL(Wpi, Bpi, in0, in1);
#pragma unroll
                for (int k = 0; k < A; ++k) mu_mb[lane * 9 + k] = outv[k];
                if constexpr (SPLIT && !M_NOISE && do_critic) {
                    // The critic needs the fragments of o_t again.  Carried across the actor's evaluation they are 16 registers the
                    // evaluation does not have (the 256-env geometry leaves a wave 256 registers and no AGPRs: they were spilled to
                    // scratch and reloaded, ~30 scratch loads per step).  The row is still in the mailbox - the env wave overwrites it
                    // with o_t+1 only after it has been given mu_t, which is posted below - so it is read and split a second time
                    // here.  (The compiler barrier keeps the two reads two: the mailbox belongs to both waves.)
                    asm volatile("" ::: "memory");
                    row_from(obs_mb + (t & (OBS_SLOTS - 1)) * (64 * 9));
                    obs_to_frags_x<OD>(o, inx);
                }
                ws_post(&seq[1], t + 1, lane);                               // mu_t posted
                WS_TOC(t_act, ta_);
            }
            if (do_critic) {
                WS_TIC(tc_);
                if constexpr (SPLIT) critic_x(inx);
                else WS_EVAL(Wv, Bv, in0, in1);
                v_mb[(t & 1) * 64 + lane] = outv[0];
                if (t > 0 && flag[(t - 1) & 1] != 0) {                       // step t-1 cut an episode that was re-drawn
                    if constexpr (SPLIT) {
                        SplitIn pin;
                        row_from(pre_mb + ((t - 1) & 1) * (64 * 9));
                        obs_to_frags_x<OD>(o, pin);
                        critic_x(pin);
                    } else {
                        half8 p0, p1;
                        frags_from(pre_mb + ((t - 1) & 1) * (64 * 9), p0, p1);
                        WS_EVAL(Wv, Bv, p0, p1);
                    }
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
        };
        if constexpr (ROLES == 3) {
            if (role == 1) net_wave(std::true_type{}, std::false_type{});
            else net_wave(std::false_type{}, std::true_type{});
        } else {
            net_wave(std::true_type{}, std::integral_constant<bool, !ECRITIC>{});
        }
        return;
    }

    // ---------------------------------------------------------------------------------------- E-wave
    Env s;
    Current cur = {0.0f, 0.0f, 0.0f, 0.0f, 0u};
    float vc0 = 0.0f, beta0 = 0.0f;
    bool ep_dirty = false, rf_dirty = (MODE == MODE_FULL);
    float o[9];
    Vessel ve = launch_vessel(a, il);                                        // in VGPRs: this wave has them to spare; re-drawn with the episode
                                                                             // when the randomisation is on
    // With the randomisation compiled in, two waves per SIMD (GROUPS = 4: 256 registers) have no room for the 29 parameters ACROSS the step
    // next to the re-draw: the block is re-read from the table at the top of every step instead (eight 16-byte loads, L2-resident, issued
    // while this wave waits for the actor's answer) and a reset only rewrites the table.
    constexpr bool VE_RELOAD = RND && (GROUPS == 4 || ROLES == 3);          // (the three-role form: a critic wave shares the env wave's SIMD)
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
    if (EXT && pa.use_lag) {                                     // continue the episode with the observation the last launch ended with
        const float4 lg = a.S3[il];
        o[6] = lg.x; o[7] = lg.y; o[8] = lg.z;
    }
#pragma unroll
    for (int k = 0; k < OD; ++k) obs_mb[lane * 9 + k] = o[k];               // parity 0
    ws_post(&seq[0], 1, lane);                                               // o_0 posted
    if (STAGE && pa.noise) load_rows<A, 64>(pa.noise + w_a, rem_a, lane, pre);
    int next_switch = 0;
    // a 64-env slice of rows: through the wave's LDS staging area (coalesced stores), or per lane where the LDS has no room for it
    auto put_rows_o = [&](void* dst, int64_t t_off, const float* v) __attribute__((always_inline)) {
        if constexpr (STAGE) wave_store_rows<OD>(lds_io, dst, t_off + w_o, rem_o, v, lane, a.obs_bf16 != 0);
        else if (live) store_row_direct<OD>(dst, (t_off + w_o) / OD + lane, v, a.obs_bf16 != 0);
    };
    auto put_rows_a = [&](void* dst, int64_t t_off, const float* v) __attribute__((always_inline)) {
        if constexpr (STAGE) wave_store_rows<A>(lds_io, dst, t_off + w_a, rem_a, v, lane);
        else if (live) store_row_direct<A>(dst, (t_off + w_a) / A + lane, v, false);
    };
    // ECRITIC: V of an observation this wave holds (one env per lane, like the one-wave kernels)
    auto critic_here = [&](const float* ob) __attribute__((always_inline)) -> float {
        float oc[9], outv[8];
#pragma unroll
        for (int k = 0; k < 9; ++k) oc[k] = k < OD ? ob[k] : 0.0f;
        SplitIn f;
        obs_to_frags_x<OD>(oc, f);
        if constexpr (PREC == PREC_F32) mlp_eval_x<KA>(Wv, Wv_l, Bv, pa.n_hidden, f, pa.leak, outv);
        else mlp_eval<KA>(Wv, Bv, pa.n_hidden, f.h0, f.h1, leak, outv);
        return outv[0];
    };
    float v_cur = 0.0f;
    bool pre_owed = false;                                                   // step t-1 left a pre-reset observation in pre_mb
    uint64_t w_mu = 0, w_v = 0, t_env = 0, t_noi = 0; const uint64_t t_start = __builtin_amdgcn_s_memtime(); (void)t_start; (void)w_mu; (void)w_v; (void)t_env; (void)t_noi;
    uint64_t t_pre = 0, t_post = 0, t_off = 0; (void)t_pre; (void)t_post; (void)t_off;
    bool boot_wanted = false, was_reset = false;                             // of the step whose boot row is still owed
    // Between "mu_t has arrived" and "o_t+1 is posted" the env wave is on the serial chain of the rollout (the network wave waits
    // for that observation), so only what o_t+1 needs is done there: a_t = mu_t + std xi_t, env.step, the reset of finished
    // envs.  The rows of step t (action, log-likelihood, reward, done, the observation row of t+1) are written after the
    // hand-over, while the network wave evaluates mu_t+1.
    put_rows_o(pa.obs_out, 0, o);
    // Pre-drawn reset sample.  A reset sits on the serial chain of the step (o_t+1 of a re-drawn env is its first observation), and with
    // termination on some env of a 64-env wave ends in about every fourth step, so the whole wave pays the Philox draw there.  The
    // draw is a pure function of (seed, global env id, episode): it is made while this wave waits for the actor's answer, for the episode
    // that would start next, and a reset on the chain is an assignment plus the first observation.  Same values, same rows.
    constexpr bool PREDRAW = true;        // measured (round 3, same call): 1-2 % in every form, e.g. f16 7.28 -> 7.18 us at 65 536 envs
    ResetDraw rdraw;
    bool need_draw = PREDRAW && (a.auto_reset || pa.reset_at_end);
    for (int t = 0; t <= pa.T; ++t) {
        const bool q_boot_wanted = boot_wanted, q_was_reset = was_reset;     // flags of step t-1
        if constexpr (ECRITIC) {
            // V(o_t) - and V of the pre-reset observation where step t-1 cut an episode that was re-drawn - while the network wave
            // evaluates mu_t.  One copy of the evaluation: first the row this wave left in pre_mb, then o_t.
            float v_pre = 0.0f;
            for (int pass = pre_owed ? 0 : 1; pass < 2; ++pass) {            // wave-uniform
                const float* rm = pre_mb + ((t - 1) & 1) * (64 * 9);
                float orow[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) orow[k] = k < OD ? (pass == 0 ? rm[lane * 9 + k] : o[k]) : 0.0f;
                const float vv = critic_here(orow);
                if (pass == 0) v_pre = vv; else v_cur = vv;
            }
            if (live) {
                if (t < pa.T) (pa.val + (int64_t)t * n)[(unsigned)i] = v_cur;
                if (t > 0) (pa.boot + (int64_t)(t - 1) * n)[(unsigned)i] = q_boot_wanted ? (q_was_reset ? v_pre : v_cur) : 0.0f;
            }
        }
        if (t == pa.T) break;
        // the exploration noise of this step does not depend on the actor's answer: it is drawn while the network wave is
        // still evaluating mu_t (the env wave would otherwise only poll)
        float xi[A];
        WS_TIC(tn_);
        if (pa.noise) {
            if constexpr (STAGE) {
                wave_rows_from_regs<A>(lds_io, pre, xi, lane);
                if (t + 1 < pa.T) load_rows<A, 64>(pa.noise + (int64_t)(t + 1) * stride_a + w_a, rem_a, lane, pre);
            } else {
#pragma unroll
                for (int k = 0; k < A; ++k) xi[k] = pa.noise[((int64_t)t * n + il) * A + k];
            }
        } else if (draw) {
            if (!M_NOISE) policy_noise<A>(a, a.env_id_base + i, nctr, xi);
            ++nctr;
        }
        if constexpr (VE_RELOAD) ve = vessel_from_env(a.env_tab, a.env_stride, il);
        if (PREDRAW && __ballot(need_draw) != 0ull) {                        // wave-uniform; lanes whose episode did not move redraw the same values
            reset_draw<MODE>(a, a.env_id_base + i, episode, rdraw);
            need_draw = false;
        }
        WS_TOC(t_noi, tn_);
        WS_WAIT_T(w_mu, &seq[1], t + 1);                                     // mu_t posted
        WS_TIC(tp_);
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
        // only what o_t+1 and the reset decision depend on stays on the chain; the reward and the azimuth bookkeeping follow the hand-over
#ifdef DPENV_WS_SELFCHECK
        constexpr bool DEFER = false;    // the diagnostic double evaluation compares whole steps
#else
        constexpr bool DEFER = true;
#endif
        StepRest rest;
        env_step_chain<MODE, EXT, DEFER>(a, ve, s, act, has_ref, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, out, rest, RND ? il : IL);
        if constexpr (!DEFER) env_step_finish<MODE, EXT, false>(a, s, act, rest, true, out);
        WS_TOC(t_env, te_);
        WS_TIC(tq_);
#ifdef DPENV_WS_SELFCHECK
        {
            ws_wait(&seq[2], t + 1);                                         // critic(o_t) done: the partner is idle from here
                StepOut outB;
            env_step<MODE, EXT>(a, ve, sB, actB, has_ref, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, outB, RND ? il : IL);
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
        // ppo.py:305-322 with reset_at_end: after the LAST step of the block every env is cut and re-drawn, ended or not
        const bool do_reset = ((a.auto_reset && out.d != 0u) || (pa.reset_at_end && t == pa.T - 1)) && live;
        float new_vc = 0.0f, new_beta = 0.0f;                               // CURR: the re-drawn env's new current (dpenv_set_current_randomisation)
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
        pre_owed = post_pre;
        if (__ballot(do_reset) != 0ull) {
            if (do_reset) {
                if constexpr (PREDRAW) { reset_apply<MODE>(a, s, rdraw, o); need_draw = true; }
                else env_auto_reset<MODE>(a, s, a.env_id_base + i, episode, o);
                // domain randomisation: the new episode runs on a new hull (the RND instantiation also serves fixed hulls with a thrust loss)
                if constexpr (VE_RELOAD) { if (a.rand_tab) redraw_vessel_table_call(a.rand_tab, a.seed_lo, a.seed_hi, a.env_tab, a.env_stride, a.env_id_base + i, i, episode); }
                else if constexpr (RND) { if (a.rand_tab) redraw_vessel_cold(a, i, episode, ve); }
                // ... in a new current: drawn here, put in force behind this step's drift update below (the drift of step t belongs to the episode
                // that ended - dpenv_step applies it before the reset -, the new episode starts exactly on the drawn values)
                if constexpr (CURR) { if (a.cur_nom) { const float2 cd = current_redraw_call(a.cur_nom, a.cur_nom_stride, a.cur_range_v, a.cur_range_b, a.seed_lo, a.seed_hi, a.env_id_base + i, i, episode); new_vc = cd.x; new_beta = cd.y; } }
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
        __builtin_amdgcn_sched_barrier(0);                                   // nothing of the rows below moves up into the chain
        if constexpr (DEFER) env_step_finish<MODE, EXT, true>(a, s, act, rest, !do_reset, out);   // reward, azimuths of a continuing env
        logp = action_logp<A>(pc, mu, act);                                  // core.py:42-46 on (a_t, mu_t)
        if (a.current_drift) current_drift_step(a, cur, vc0, beta0, a.env_id_base + i);   // the current of step t+1: not needed by o_t+1
        if constexpr (CURR) { if (a.cur_nom && do_reset) { cur.vc = new_vc; cur.beta = new_beta; vc0 = new_vc; beta0 = new_beta; current_components(cur); } }
        put_rows_a(pa.act_out, (int64_t)t * stride_a, act);
        if (t + 1 < pa.T) put_rows_o(pa.obs_out, (int64_t)(t + 1) * stride_o, o);
        if (live) {
            (pa.rew + (int64_t)t * n)[(unsigned)i] = out.reward;
            (pa.done + (int64_t)t * n)[(unsigned)i] = (uint8_t)out.d;
            (pa.logp + (int64_t)t * n)[(unsigned)i] = logp;
        }
        if constexpr (ECRITIC) {
            WS_TOC(t_off, tr_);
        } else {
            WS_WAIT_T(w_v, &seq[2], t + 1);                                  // V(o_t), V(pre-reset o_t) posted
            WS_TOC(t_off, tr_);
            if (live) {
                const float v_t = v_mb[(t & 1) * 64 + lane];
                (pa.val + (int64_t)t * n)[(unsigned)i] = v_t;
                if (t > 0) (pa.boot + (int64_t)(t - 1) * n)[(unsigned)i] = q_boot_wanted ? (q_was_reset ? vpre_mb[(t & 1) * 64 + lane] : v_t) : 0.0f;
            }
        }
    }
    if constexpr (!ECRITIC) ws_wait(&seq[2], pa.T + 1);                      // V(o_T) posted
#ifdef DPENV_WS_PROFILE
    if (live && pa.T >= 12) {
        (pa.logp + (int64_t)0 * n)[(unsigned)i] = (float)w_mu; (pa.logp + (int64_t)1 * n)[(unsigned)i] = (float)w_v; (pa.logp + (int64_t)2 * n)[(unsigned)i] = (float)(__builtin_amdgcn_s_memtime() - t_start);
        (pa.logp + (int64_t)7 * n)[(unsigned)i] = (float)t_env; (pa.logp + (int64_t)8 * n)[(unsigned)i] = (float)t_noi;
        (pa.logp + (int64_t)9 * n)[(unsigned)i] = (float)t_pre; (pa.logp + (int64_t)10 * n)[(unsigned)i] = (float)t_post;
        (pa.logp + (int64_t)11 * n)[(unsigned)i] = (float)t_off;
    }
#endif
    put_rows_o(pa.last_obs, 0, o);
    if (live) {
        if constexpr (ECRITIC) pa.last_val[i] = v_cur;                       // V(o_T); boot[T-1] went out with it
        else {
            const float v_T = v_mb[(pa.T & 1) * 64 + lane];
            (pa.boot + (int64_t)(pa.T - 1) * n)[(unsigned)i] = boot_wanted ? (was_reset ? vpre_mb[(pa.T & 1) * 64 + lane] : v_T) : 0.0f;
            pa.last_val[i] = v_T;
        }
        store_env(a, i, s, rf_dirty);
        if (EXT) a.S3[i] = make_float4(o[6], o[7], o[8], 0.0f);
        if (ep_dirty) a.episode[i] = (int)episode;
        if (a.current_drift) { a.cur_vc[i] = cur.vc; a.cur_beta[i] = cur.beta; a.drift_ctr[i] = cur.ctr; }
        if constexpr (CURR) { if (a.cur_nom && ep_dirty) store_current(a, i, cur, vc0, beta0, true); }
        if (draw) a.noise_ctr[i] = nctr;
    }
}


}  // namespace dpenv

// ---- host side: one launcher per arithmetic, instantiated by the translation unit that owns it -------------------------------
// Two-wave launches exist for hidden activation leaky-relu / relu in every arithmetic; tanh only in PREC_F16 with four groups
// (dpenv_set_policy_desc routes the others to the one-wave kernels).  -DDPENV_DEV_FAST (development builds only, never shipped)
// instantiates the shipped configuration alone: final / continuous angles / extended state, width <= 80.
namespace dpenv_ws_launch {
using namespace dpenv;

// which arithmetics get the critic wave (ROLES = 3) in the 128-env geometry: bit 0 f16, bit 1 all exact, bit 2 exact actor.  Measured
// (same call, bit-identical rows, profiles/r04_critic_wave.txt; 32 768 / 8 192 envs): all exact 9.43 -> 7.85 / 9.24 -> 7.30 us per step, exact
// actor 7.4-7.7 -> 7.3-7.45 / 7.23 -> 7.01; f16 5.20 -> 5.50 / 4.76 -> 5.03 with row staging (116 B of scratch at the 256 registers two waves
// on a SIMD leave), 5.27 -> 5.33 / 4.79 -> 4.92 without it (no scratch): the f16 step is its chain already - so the two split arithmetics
// get the critic wave, f16 keeps two roles.
// (DPENV_WS_CRITIC_WAVE: dpenv_dev.h, default 6)
template <int MODE, bool EXT, int KA, int PREC, int GROUPS, bool RND = false, bool SLOSS = false>
static hipError_t go(const StepArgs& a, const PolicyArgs& pa, hipStream_t s)
{
    constexpr int ROLES = (GROUPS == 2 && ((DPENV_WS_CRITIC_WAVE >> PREC) & 1)) ? 3 : 2;
    const dim3 grid((a.n + 64 * GROUPS - 1) / (64 * GROUPS));
    const size_t lds = (size_t)ws_images(PREC) * pa.nent * 16 + (size_t)2 * pa.nblk * 32 * 4 +
                       (size_t)GROUPS * ((((PREC == PREC_F16 && ROLES == 2)) ? WS_GROUP_FLOATS : WS_GROUP_FLOATS_X) +
                                         (ROLES == 3 ? 64 * 9 : 0)) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)policy_rollout_ws_kernel<MODE, EXT, KA, ROLES, PREC, GROUPS, RND, SLOSS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((policy_rollout_ws_kernel<MODE, EXT, KA, ROLES, PREC, GROUPS, RND, SLOSS>), grid, dim3(64 * GROUPS * ROLES), lds, s, a, pa);
    return hipGetLastError();
}

// the randomisation's instantiation exists for the shipped training configuration (train.py:47-54: final, continuous angles, extended
// state) with leaky-relu / relu networks; dpenv_policy_rollout refuses the others while the randomisation is on (dpenv.h)
template <int MODE, bool EXT, int KA, int PREC, int GROUPS>
static hipError_t pick(const StepArgs& a, const PolicyArgs& pa, hipStream_t s)
{
    if (a.loss_on == LOSS_SHARED) {                       // the single class's coefficients as kernel arguments
        if (a.env_tab) return hipErrorInvalidValue;
        if constexpr (MODE == MODE_FINAL_CONT && EXT && KA < 16) return go<MODE, EXT, KA, PREC, GROUPS, false, true>(a, pa, s);
        else return hipErrorNotSupported;
    }
    if (a.rand_tab || a.loss_on != LOSS_NONE || a.cur_nom) {   // the general per-env form: hull / current re-draws, the table's thrust loss
        if constexpr (MODE == MODE_FINAL_CONT && EXT && KA < 16) return go<MODE, EXT, KA, PREC, GROUPS, true>(a, pa, s);
        else return hipErrorNotSupported;
    }
    return go<MODE, EXT, KA, PREC, GROUPS, false>(a, pa, s);
}

template <int MODE, bool EXT, int PREC>
static hipError_t by_shape(const StepArgs& a, const PolicyArgs& pa, hipStream_t s)
{
    const int ka = pa.ks + 16 * pa.act;
    const bool two = pa.ws_groups == 2;
#ifdef DPENV_DEV_FAST
    if (ka != 5) return hipErrorInvalidValue;
    return two ? pick<MODE, EXT, 5, PREC, 2>(a, pa, s) : pick<MODE, EXT, 5, PREC, 4>(a, pa, s);
#else
    switch (ka) {
    case 5: return two ? pick<MODE, EXT, 5, PREC, 2>(a, pa, s) : pick<MODE, EXT, 5, PREC, 4>(a, pa, s);
    case 6: return two ? pick<MODE, EXT, 6, PREC, 2>(a, pa, s) : pick<MODE, EXT, 6, PREC, 4>(a, pa, s);
    }
    if constexpr (PREC == PREC_F16) {
        if (two) return hipErrorInvalidValue;
        if (ka == 21) return pick<MODE, EXT, 21, PREC, 4>(a, pa, s);
        if (ka == 22) return pick<MODE, EXT, 22, PREC, 4>(a, pa, s);
    }
    return hipErrorInvalidValue;
#endif
}

template <int PREC>
static hipError_t launch(const StepArgs& a, const PolicyArgs& pa, int mode, int ext, hipStream_t s)
{
    if ((pa.ks != 5 && pa.ks != 6) || (pa.act != 0 && pa.act != 1) || (pa.ws_groups != 2 && pa.ws_groups != 4)) return hipErrorInvalidValue;
#ifdef DPENV_DEV_FAST
    if (mode != MODE_FINAL_CONT || !ext) return hipErrorInvalidValue;
    return by_shape<MODE_FINAL_CONT, true, PREC>(a, pa, s);
#else
    switch (mode) {
    case MODE_FULL: return ext ? by_shape<MODE_FULL, true, PREC>(a, pa, s) : by_shape<MODE_FULL, false, PREC>(a, pa, s);
    case MODE_SIMPLE: return ext ? by_shape<MODE_SIMPLE, true, PREC>(a, pa, s) : by_shape<MODE_SIMPLE, false, PREC>(a, pa, s);
    case MODE_LIMITED: return ext ? by_shape<MODE_LIMITED, true, PREC>(a, pa, s) : by_shape<MODE_LIMITED, false, PREC>(a, pa, s);
    case MODE_FINAL_WRAP: return ext ? by_shape<MODE_FINAL_WRAP, true, PREC>(a, pa, s) : by_shape<MODE_FINAL_WRAP, false, PREC>(a, pa, s);
    case MODE_FINAL_CONT: return ext ? by_shape<MODE_FINAL_CONT, true, PREC>(a, pa, s) : by_shape<MODE_FINAL_CONT, false, PREC>(a, pa, s);
    }
    return hipErrorInvalidValue;
#endif
}
}  // namespace dpenv_ws_launch

#endif
