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
    uint4* lds_w = lds_dyn;                                                     // [2][nent] fragment entries
    const float* lds_b = (const float*)(lds_dyn + 2 * pa.nent);                       // [2][nblk][32] bias floats
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
    mlp_eval2<KA>(lds_w, lds_w + pa.nent, lds_b, lds_b + pa.nblk * 32, pa.n_hidden, in0, in1, (_Float16)pa.leak, mu, vv);
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
    const uint4* Wv = lds_w + pa.nent;
    const float* Bpi = (const float*)(lds_dyn + 2 * pa.nent);
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
    Vessel ve = launch_vessel_plain(a, il);                      // re-drawn with the episode when the randomisation is on
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
    if (EXT && pa.use_lag) {                                     // continue the episode with the observation the last launch ended with
        const float4 lg = a.S3[il];
        o[6] = lg.x; o[7] = lg.y; o[8] = lg.z;
    }
    half8 in0, in1;
    obs_to_frags<OD>(o, in0, in1);
    float vout[8], mu[8];
    mlp_eval2<KA>(Wpi, Wv, Bpi, Bv, pa.n_hidden, in0, in1, leak, mu, vout);     // actor and critic of o_0
    float v_t = vout[0];

    float pre[A];
    if (pa.noise) load_rows<A, 64>(pa.noise + w_a, rem_a, lane, pre);
    int next_switch = 0;
    for (int t = 0; t < pa.T; ++t) {
        // ---- store the policy input row; the actor's mean for it is already there (joint evaluation) -------------
        wave_store_rows<OD>(lds_io, pa.obs_out, (int64_t)t * stride_o + w_o, rem_o, o, lane, a.obs_bf16 != 0);
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
        env_step<MODE, EXT>(a, ve, s, act, has_ref, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, out, il);
        if (a.current_drift) current_drift_step(a, cur, vc0, beta0, a.env_id_base + i);

#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = out.o[k];
        // ppo.py:305-322 with reset_at_end: after the LAST step of the block every env is cut and re-drawn, ended or not
        const bool do_reset = ((a.auto_reset && out.d != 0u) || (pa.reset_at_end && t == pa.T - 1)) && live;
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
                if (a.rand_tab) redraw_vessel_cold(a, i, episode, ve);    // domain randomisation: the new episode runs on a new hull
                if (a.cur_nom) current_redraw(a, i, episode, cur, vc0, beta0);    // ... in a new current (stored with the final state)
                ++episode; ep_dirty = true; rf_dirty = true;
            }
        }
        obs_to_frags<OD>(o, in0, in1);
        mlp_eval2<KA>(Wpi, Wv, Bpi, Bv, pa.n_hidden, in0, in1, leak, mu, vout);
        const float v_next = do_reset ? v_pre : vout[0];
        const float v_new = vout[0];
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
        if (EXT) a.S3[i] = make_float4(o[6], o[7], o[8], 0.0f);
        if (ep_dirty) a.episode[i] = (int)episode;
        if (a.current_drift) { a.cur_vc[i] = cur.vc; a.cur_beta[i] = cur.beta; a.drift_ctr[i] = cur.ctr; }
        if (a.cur_nom && ep_dirty) store_current(a, i, cur, vc0, beta0, true);
        if (draw) a.noise_ctr[i] = nctr;
    }
}

// =============================================================================================
//  Weight packing ON THE DEVICE: fp32 dense kernels W[l][in][out] / biases (the reference's variable layout, core.py:29-33)
//  -> the LDS image the evaluation kernels stage (MFMA A-operand fragments, bias tiles, std / logp constants).
//  One thread per f16 element of the COMPACT image (FragAddr: block-2 fragments of an 80-wide layer and the output layer's fragments
//  keep only the rows anybody reads).  Logical fragment f, lane l = (r = l & 31, h = l >> 5), element j holds
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
                                                          int nent, int nblk, int split, _Float16* frags, float* bias, float* consts)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int per_net = nent * 8;
    if (tid < 2 * per_net) {
        const int net = tid / per_net, el = tid % per_net, j = el & 7;
        int e = el >> 3;
        const PackNet& m = net ? v : pi;
        // entry of the compact image (FragAddr, dpenv_policy_dev.h) -> logical fragment f and a lane that reads this entry
        const int nh = m.n_layers - 1, B2 = ks_n == 5 ? 32 : 64, L0 = 128 + B2, LH = ks_n * (128 + B2);
        auto lane_of = [](int k, int per_half) { return (k / per_half) * 32 + (k % per_half); };      // entry k of a 2 x per_half fragment
        int f, lane;
        if (e < L0) {
            if (e < 128) { f = e >> 6; lane = e & 63; }
            else { f = 2; lane = lane_of(e - 128, B2 / 2); }
        } else if (e < L0 + (nh - 1) * LH) {
            e -= L0;
            const int l = e / LH, r = e % LH;
            if (r < 2 * ks_n * 64) { f = 3 + l * 3 * ks_n + (r >> 6); lane = r & 63; }
            else { const int r2 = r - 2 * ks_n * 64; f = 3 + l * 3 * ks_n + 2 * ks_n + r2 / B2; lane = lane_of(r2 % B2, B2 / 2); }
        } else {
            e -= L0 + (nh - 1) * LH;
            f = 3 + (nh - 1) * 3 * ks_n + (e >> 4);
            lane = lane_of(e & 15, 8);
        }
        const float w = pack_weight(m, ks_n, f, lane, j);
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

extern "C" hipError_t dpenv_dev_launch_pack_policy(const PackNet* pi, const PackNet* v, const float* log_std, int adim, int ks, int nent,
                                                   int nblk, int split, void* frags, float* bias, float* consts, hipStream_t s)
{
    const int total = 2 * nent * 8;
    hipLaunchKernelGGL(pack_policy_kernel, dim3((total + 255) / 256), dim3(256), 0, s, *pi, *v, log_std, adim, ks, nent, nblk, split,
                       (_Float16*)frags, bias, consts);
    return hipGetLastError();
}

static size_t policy_lds_bytes(const PolicyArgs& pa)
{
    return (size_t)2 * pa.nent * 16 + (size_t)2 * pa.nblk * 32 * 4 + (size_t)PWAVES * 64 * 9 * 4;
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
static hipError_t launch_policy_rollout_one(const StepArgs& a, const PolicyArgs& pa, hipStream_t s)
{
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
#ifdef DPENV_DEV_FAST
    if (pa.ks != 5 || pa.act != 0 || !ext) return hipErrorInvalidValue;
    return launch_policy_rollout_one<MODE, true, 5>(a, pa, s);
#else
    switch (pa.ks + 16 * pa.act) {
    case 5: return ext ? launch_policy_rollout_one<MODE, true, 5>(a, pa, s) : launch_policy_rollout_one<MODE, false, 5>(a, pa, s);
    case 6: return ext ? launch_policy_rollout_one<MODE, true, 6>(a, pa, s) : launch_policy_rollout_one<MODE, false, 6>(a, pa, s);
    case 21: return ext ? launch_policy_rollout_one<MODE, true, 21>(a, pa, s) : launch_policy_rollout_one<MODE, false, 21>(a, pa, s);
    default: return ext ? launch_policy_rollout_one<MODE, true, 22>(a, pa, s) : launch_policy_rollout_one<MODE, false, 22>(a, pa, s);
    }
#endif
}

extern "C" hipError_t dpenv_dev_launch_policy_rollout(const StepArgs* a, const PolicyArgs* pa, int mode, int ext,
                                                      hipStream_t s)
{
    if (pa->ws) return dpenv_dev_launch_policy_rollout_ws(a, pa, mode, ext, s);      // dpenv_policy_ws.hip
#ifdef DPENV_DEV_FAST
    return mode == MODE_FINAL_CONT ? launch_policy_rollout_mode<MODE_FINAL_CONT>(*a, *pa, ext, s) : hipErrorInvalidValue;
#else
    switch (mode) {
    case MODE_FULL: return launch_policy_rollout_mode<MODE_FULL>(*a, *pa, ext, s);
    case MODE_SIMPLE: return launch_policy_rollout_mode<MODE_SIMPLE>(*a, *pa, ext, s);
    case MODE_LIMITED: return launch_policy_rollout_mode<MODE_LIMITED>(*a, *pa, ext, s);
    case MODE_FINAL_WRAP: return launch_policy_rollout_mode<MODE_FINAL_WRAP>(*a, *pa, ext, s);
    case MODE_FINAL_CONT: return launch_policy_rollout_mode<MODE_FINAL_CONT>(*a, *pa, ext, s);
    }
    return hipErrorInvalidValue;
#endif
}
