// dpenv_policy_x.hip - the PPO actor-critic inside the rollout launch in split-f16 ("fp32-faithful") arithmetic:
// DPENV_POLICY_F32.  Reference: mlp_gaussian_policy / mlp_actor_critic, spinup/algos/tf1/ppo/core.py:29-33,80-107 (fp32 TF1
// dense layers), gaussian_likelihood core.py:42-46, rollout loop ppo.py:289-322.  This is the mode parity with the reference's
// fp32 networks is claimed on (mu, v and logp within 1e-5 of an fp32 evaluation; tests/test_gpu_policy.py); dpenv_policy.hip
// holds the f16 fast mode.  See dpenv_policy_dev.h (mlp_eval_x) for the arithmetic.
//
// Launch form: one wave per 64 envs, 256-thread workgroups.  The LDS image of both networks is twice the f16 one (high and
// low fragments: 152 KiB for the shipped 9-80-80-80 shape), which leaves no room for the two-wave form's mailboxes nor for row
// staging: observation / action / noise rows are accessed per lane (36 / 28-byte rows; slower stores, same bytes).
#include "dpenv_policy_dev.h"

namespace dpenv {

struct SplitNets {
    const uint4 *Wpi_h, *Wv_h, *Wpi_l, *Wv_l;
    const float *Bpi, *Bv;
};

__device__ __forceinline__ SplitNets split_nets(const uint4* lds_w, const PolicyArgs& pa)
{
    SplitNets s;
    s.Wpi_h = lds_w;
    s.Wv_h = lds_w + pa.nent;
    s.Wpi_l = lds_w + 2 * pa.nent;
    s.Wv_l = lds_w + 3 * pa.nent;
    s.Bpi = (const float*)(lds_w + 4 * pa.nent);
    s.Bv = s.Bpi + pa.nblk * 32;
    return s;
}

// the critic in the arithmetic the descriptor asked for: split-f16 like the actor (DPENV_POLICY_F32), or plain f16 on the HIGH
// image and the high parts of the input - exactly the F16 mode's critic, bit for bit (DPENV_POLICY_F32_ACTOR)
template <int KA>
__device__ __forceinline__ void critic_eval(const SplitNets& nets, const PolicyArgs& pa, const SplitIn& in, float leak, float out[8])
{
    if (pa.critic_f16) mlp_eval<KA>(nets.Wv_h, nets.Bv, pa.n_hidden, in.h0, in.h1, (_Float16)leak, out);
    else mlp_eval_x<KA>(nets.Wv_h, nets.Wv_l, nets.Bv, pa.n_hidden, in, leak, out);
}

template <int OD, int A, int KA>
__global__ __launch_bounds__(PBLOCK) void policy_forward_x_kernel(const PolicyArgs pa, const float* obs, float* mu_out, float* v_out, int n)
{
    extern __shared__ uint4 lds_dyn[];
    stage_weights(lds_dyn, pa);
    const SplitNets nets = split_nets(lds_dyn, pa);
    const int i = blockIdx.x * PBLOCK + threadIdx.x;
    if ((int)(blockIdx.x * PBLOCK + (threadIdx.x & ~63)) >= n) return;          // whole wave out of range (uniform)
    const bool live = i < n;
    const int il = live ? i : n - 1;
    float o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < OD; ++k) o[k] = obs[(int64_t)il * OD + k];
    SplitIn in;
    obs_to_frags_x<OD>(o, in);
    float mu[8], vv[8];
    mlp_eval_x<KA>(nets.Wpi_h, nets.Wpi_l, nets.Bpi, pa.n_hidden, in, pa.leak, mu);
    critic_eval<KA>(nets, pa, in, pa.leak, vv);
    if (live) {
        store_row_direct<A>(mu_out, i, mu, false);
        v_out[i] = vv[0];
    }
}

// the rollout loop of policy_rollout_kernel (dpenv_policy.hip) with the exact network evaluation; rows, bootstrap values and
// auto-reset handling are identical
template <int MODE, bool EXT, int KA>
__global__ __launch_bounds__(PBLOCK) void policy_rollout_x_kernel(const StepArgs a, const PolicyArgs pa)
{
    constexpr int A = ModeTraits<MODE>::A;
    constexpr int OD = EXT ? 9 : 6;
    extern __shared__ uint4 lds_dyn[];
    stage_weights(lds_dyn, pa);
    const SplitNets nets = split_nets(lds_dyn, pa);
    const float leak = pa.leak;
    const int n = a.n;
    const int wave0 = blockIdx.x * PBLOCK + (threadIdx.x & ~63);
    if (wave0 >= n) return;
    const int i = blockIdx.x * PBLOCK + threadIdx.x;
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
    Vessel ve = launch_vessel_plain(a, il);                      // re-drawn with the episode when the randomisation is on
    uint32_t episode = a.auto_reset ? (uint32_t)a.episode[il] : 0u;
    bool ep_dirty = false, rf_dirty = (MODE == MODE_FULL);
    const PolicyConsts<A> pc = load_policy_consts<A>(pa);
    const bool draw = pa.noise == nullptr && pa.sample != 0;
    uint32_t nctr = draw ? a.noise_ctr[il] : 0u;

    float o[9];
    {
        float sr_, cr_;
        bool same_;
        make_obs(s.N, s.E, s.psi, s.u, s.v, s.r, s.refN, s.refE, s.refPsi, s.pt, a.wrap_mode == WRAP_REFERENCE, o, sr_, cr_, same_);
    }
    if (EXT && pa.use_lag) {                                     // continue the episode with the observation the last launch ended with
        const float4 lg = a.S3[il];
        o[6] = lg.x; o[7] = lg.y; o[8] = lg.z;
    }
    SplitIn in;
    float vout[8], mu[8];
    obs_to_frags_x<OD>(o, in);
    mlp_eval_x<KA>(nets.Wpi_h, nets.Wpi_l, nets.Bpi, pa.n_hidden, in, leak, mu);
    critic_eval<KA>(nets, pa, in, leak, vout);
    float v_t = vout[0];

    int next_switch = 0;
    for (int t = 0; t < pa.T; ++t) {
        const int64_t row = (int64_t)t * n + i;
        if (live) store_row_direct<OD>(pa.obs_out, row, o, a.obs_bf16 != 0);
        float act[A];
        float logp;
        if (pa.noise || draw) {
            float xi[A];
            if (pa.noise) {
#pragma unroll
                for (int k = 0; k < A; ++k) xi[k] = pa.noise[((int64_t)t * n + il) * A + k];
            } else {
                policy_noise<A>(a, a.env_id_base + i, nctr, xi);
                ++nctr;
            }
            logp = sample_action<A>(pc, mu, xi, act);
        } else {
            logp = mean_action<A>(pc, mu, act);
        }
        if (live) store_row_direct<A>(pa.act_out, row, act, false);

        bool has_ref = false;
        float nrN = 0.0f, nrE = 0.0f, nrP = 0.0f;
        if (next_switch < pa.n_switch && pa.switch_step[next_switch] == t) {
            const float* rp = pa.refs + (int64_t)next_switch * 3 * n;
            nrN = rp[il]; nrE = rp[(int64_t)n + il]; nrP = rp[2 * (int64_t)n + il];
            has_ref = true; rf_dirty = true;
            ++next_switch;
        }
        StepOut out;
        env_step<MODE, EXT>(a, ve, s, act, has_ref, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, out, il);
        if (a.current_drift) current_drift_step(a, cur, vc0, beta0, a.env_id_base + i);
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = out.o[k];
        // ppo.py:305-322 with reset_at_end: after the LAST step of the block every env is cut and re-drawn, ended or not
        const bool do_reset = ((a.auto_reset && out.d != 0u) || (pa.reset_at_end && t == pa.T - 1)) && live;
        float v_pre = 0.0f;
        if (__ballot(do_reset) != 0ull) {                       // wave-uniform
            // the critic once more, on the pre-reset observation - only if an env of the wave was CUT (time limit): a terminated
            // env bootstraps with 0 (ppo.py:311), and with termination on most finished envs are terminated ones
            if (__ballot(do_reset && (out.d & DONE_TERMINAL) == 0u) != 0ull) {
                obs_to_frags_x<OD>(o, in);
                critic_eval<KA>(nets, pa, in, leak, vout);
                v_pre = vout[0];
            }
            if (do_reset) {
                env_auto_reset<MODE>(a, s, a.env_id_base + i, episode, o);
                if (a.rand_tab) redraw_vessel_cold(a, i, episode, ve);    // domain randomisation: the new episode runs on a new hull
                if (a.cur_nom) current_redraw(a, i, episode, cur, vc0, beta0);    // ... in a new current (stored with the final state)
                ++episode; ep_dirty = true; rf_dirty = true;
            }
        }
        obs_to_frags_x<OD>(o, in);
        mlp_eval_x<KA>(nets.Wpi_h, nets.Wpi_l, nets.Bpi, pa.n_hidden, in, leak, mu);
        critic_eval<KA>(nets, pa, in, leak, vout);
        const float v_next = do_reset ? v_pre : vout[0];
        const float v_new = vout[0];
        const bool terminal = (out.d & DONE_TERMINAL) != 0u;
        const bool ended = (out.d != 0u) || (t == pa.T - 1);
        const float boot = (ended && !terminal) ? v_next : 0.0f;          // ppo.py:311
        if (live) {
            pa.rew[row] = out.reward;
            pa.done[row] = (uint8_t)out.d;
            pa.val[row] = v_t;
            pa.logp[row] = logp;
            pa.boot[row] = boot;
        }
        v_t = v_new;
    }
    if (live) {
        store_row_direct<OD>(pa.last_obs, i, o, a.obs_bf16 != 0);
        pa.last_val[i] = v_t;
        store_env(a, i, s, rf_dirty);
        if (EXT) a.S3[i] = make_float4(o[6], o[7], o[8], 0.0f);
        if (ep_dirty) a.episode[i] = (int)episode;
        if (a.current_drift) { a.cur_vc[i] = cur.vc; a.cur_beta[i] = cur.beta; a.drift_ctr[i] = cur.ctr; }
        if (a.cur_nom && ep_dirty) store_current(a, i, cur, vc0, beta0, true);
        if (draw) a.noise_ctr[i] = nctr;
    }
}

}  // namespace dpenv

using namespace dpenv;

static size_t lds_bytes_x(const PolicyArgs& pa) { return (size_t)4 * pa.nent * 16 + (size_t)2 * pa.nblk * 32 * 4; }

extern "C" hipError_t dpenv_dev_launch_policy_forward_x(const PolicyArgs* pa, int od, int adim, const float* obs, float* mu, float* v,
                                                        int n, hipStream_t s)
{
    const dim3 grid((n + PBLOCK - 1) / PBLOCK), block(PBLOCK);
    const size_t lds = lds_bytes_x(*pa);
#define FWD_K(OD_, A_, KS_)                                                                                              \
    do {                                                                                                                 \
        hipError_t e = hipFuncSetAttribute((const void*)policy_forward_x_kernel<OD_, A_, KS_>,                            \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
        if (e != hipSuccess) return e;                                                                                   \
        hipLaunchKernelGGL((policy_forward_x_kernel<OD_, A_, KS_>), grid, block, lds, s, *pa, obs, mu, v, n);             \
        return hipGetLastError();                                                                                        \
    } while (0)
#define FWD(OD_, A_)                                                                                                     \
    do {                                                                                                                 \
        if (ka == 5) FWD_K(OD_, A_, 5);                                                                                  \
        if (ka == 6) FWD_K(OD_, A_, 6);                                                                                  \
        if (ka == 21) FWD_K(OD_, A_, 21);                                                                                \
        FWD_K(OD_, A_, 22);                                                                                              \
    } while (0)
    if ((pa->ks != 5 && pa->ks != 6) || (pa->act != 0 && pa->act != 1) || !pa->split) return hipErrorInvalidValue;
    const int ka = pa->ks + 16 * pa->act;
#ifdef DPENV_DEV_FAST
    if (od == 9 && adim == 7 && ka == 5) FWD_K(9, 7, 5);
#else
    if (od == 9 && adim == 7) FWD(9, 7);
    if (od == 9 && adim == 5) FWD(9, 5);
    if (od == 9 && adim == 6) FWD(9, 6);
    if (od == 6 && adim == 7) FWD(6, 7);
    if (od == 6 && adim == 5) FWD(6, 5);
    if (od == 6 && adim == 6) FWD(6, 6);
    if (od == 6 && adim == 3) FWD(6, 3);
#endif
#undef FWD_K
#undef FWD
    return hipErrorInvalidValue;
}

template <int MODE, bool EXT, int KA>
static hipError_t launch_x_one(const StepArgs& a, const PolicyArgs& pa, hipStream_t s)
{
    const dim3 grid((a.n + PBLOCK - 1) / PBLOCK), block(PBLOCK);
    const size_t lds = lds_bytes_x(pa);
    hipError_t e = hipFuncSetAttribute((const void*)policy_rollout_x_kernel<MODE, EXT, KA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((policy_rollout_x_kernel<MODE, EXT, KA>), grid, block, lds, s, a, pa);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_x_mode(const StepArgs& a, const PolicyArgs& pa, bool ext, hipStream_t s)
{
    if ((pa.ks != 5 && pa.ks != 6) || (pa.act != 0 && pa.act != 1) || !pa.split) return hipErrorInvalidValue;
#ifdef DPENV_DEV_FAST
    if (pa.ks != 5 || pa.act != 0 || !ext) return hipErrorInvalidValue;
    return launch_x_one<MODE, true, 5>(a, pa, s);
#else
    switch (pa.ks + 16 * pa.act) {
    case 5: return ext ? launch_x_one<MODE, true, 5>(a, pa, s) : launch_x_one<MODE, false, 5>(a, pa, s);
    case 6: return ext ? launch_x_one<MODE, true, 6>(a, pa, s) : launch_x_one<MODE, false, 6>(a, pa, s);
    case 21: return ext ? launch_x_one<MODE, true, 21>(a, pa, s) : launch_x_one<MODE, false, 21>(a, pa, s);
    default: return ext ? launch_x_one<MODE, true, 22>(a, pa, s) : launch_x_one<MODE, false, 22>(a, pa, s);
    }
#endif
}

extern "C" hipError_t dpenv_dev_launch_policy_rollout_x(const StepArgs* a, const PolicyArgs* pa, int mode, int ext, hipStream_t s)
{
    if (pa->ws) return pa->critic_f16 ? dpenv_dev_launch_policy_rollout_xws_f32_actor(a, pa, mode, ext, s)
                                      : dpenv_dev_launch_policy_rollout_xws_f32(a, pa, mode, ext, s);
#ifdef DPENV_DEV_FAST
    return mode == MODE_FINAL_CONT ? launch_x_mode<MODE_FINAL_CONT>(*a, *pa, ext, s) : hipErrorInvalidValue;
#else
    switch (mode) {
    case MODE_FULL: return launch_x_mode<MODE_FULL>(*a, *pa, ext, s);
    case MODE_SIMPLE: return launch_x_mode<MODE_SIMPLE>(*a, *pa, ext, s);
    case MODE_LIMITED: return launch_x_mode<MODE_LIMITED>(*a, *pa, ext, s);
    case MODE_FINAL_WRAP: return launch_x_mode<MODE_FINAL_WRAP>(*a, *pa, ext, s);
    case MODE_FINAL_CONT: return launch_x_mode<MODE_FINAL_CONT>(*a, *pa, ext, s);
    }
    return hipErrorInvalidValue;
#endif
}
