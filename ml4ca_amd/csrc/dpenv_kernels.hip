// dpenv_kernels.hip - gfx950 (CDNA4 / MI355X) kernels of libdpenv.so.
//
// Hot path: Revolt.step (reference src/rl/windows_workspace/specific/customEnv.py:92-133, "ENV")
// for a batch of independent environments, ONE WAVEFRONT LANE PER ENVIRONMENT:
//   action decode + clip        ENV:104-110,215-244
//   command map                 ENV:117-122
//   thruster force map          src/sl/SupervisedTau.py:42-83, qp_allocator.py:51-55,69-70
//   plant, 20 x 10 ms           BUILD-OWNED (ENV:124 calls the closed Cybersea simulator)
//   pose-error observation      errorFrame.py:25-37, mathematics.py:7-17, ENV:196-205
//   reward                      ENV:253-325
//   termination                 ENV:207-213
//   new_ref                     ENV:131
//   auto-reset                  ENV:135-194 + simtools.py:109-123 (batched form of ppo.py:305-322)
//
// Memory plan (HBM): library-owned state as four float4 streams so that every wave-level access
// is one 1 KiB coalesced dwordx4 transaction (16 B per lane):
//   S0[i] = (N, E, psi, u)            read + written every step
//   S1[i] = (v, r, a_port, a_star)    read + written every step
//   S2[i] = (pt_bow, pt_port, pt_star, step_count bits)   read + written every step
//   RF[i] = (ref_N, ref_E, ref_psi, a_bow)   read every step, written only on new_ref / reset / FULL
// => 64 B read + 48 B written of state per env-step; with a 7-float action, 9-float observation,
// reward and done byte: 92 B read + 89 B written (SURVEY 8d accounts 88 + 89 = 177 B).
// Caller-facing action/observation batches are [n][dim] (torch-native); they are transposed
// through LDS (odd row strides 7 / 9 / 5 / 3 -> conflict-free ds_read/ds_write) so that global
// traffic stays fully coalesced.  Per-vessel-class parameter blocks (mass, damping, thruster
// geometry) are staged in LDS as [param][class] when more than one class exists; with a single
// class they arrive as kernel arguments (SGPRs); per-ENV blocks (dpenv_set_vessel_params, domain
// randomisation) are eight more float4 streams ET[g][i], loaded straight into registers (or, for the
// A/B, by LDS-DMA into a [group][lane] image).  No MFMA: this is batched 3x3 physics.
#include "dpenv_env_dev.h"

namespace dpenv {

#ifdef DPENV_STEP_TRACE
// Diagnostic builds only (tools/build_kernels_variant.sh, tools/step_placement.py; never in the product library): where and when
// every workgroup of step_kernel ran.  Ring of DPENV_STEP_TRACE_RING launches, slot = the env's own step counter, record per
// workgroup = {HW_REG_XCC_ID, HW_REG_HW_ID, s_memrealtime at entry, s_memrealtime after the last store was issued}.
__device__ uint32_t* g_step_trace = nullptr;
constexpr int STEP_TRACE_RING = 8;
#endif

// =============================================================================================
//  env.step: one launch = one step of every env
// =============================================================================================
// RESETW (launched when config.auto_reset is on; 64-env workgroups only): the workgroup has a SECOND wave that does nothing but prepare
// the re-draw of every env of the group - episode counter and setpoint loaded, the two Philox blocks, the sampler, the new episode's
// first observation (reset_draw + reset_apply: ~250 VALU instructions, a pure function of (seed, global env id, episode, setpoint)) -
// and leaves the result in LDS.  It runs beside the env wave's plant loop on another SIMD of the CU, so a finished env costs the env
// wave a barrier and seventeen LDS reads instead of 0.6 us of lone-wave instruction issue (round 4: with termination on, some wave of
// every launch holds a finished env, so every launch paid for the draw; tools/step_placement.py).  Same
// functions on the same inputs: every row is bit-identical to the one-wave form.
constexpr int RESETW_FIELDS = 18;       // N, E, psi | o[0..8] | pt[0..2] | sin psi, cos psi | episode counter (bits)
constexpr int RESETW_FIELDS_RND = 20;   // the general per-env form and the shared training form (VES 4, 5): | the new episode's current V_c, beta_c

template <int MODE>
__device__ __forceinline__ void reset_wave(const StepArgs& a, float* lds, int lane, int blk)
{
    const int n = a.n;
    const int i = blk * 64 + lane;
    const int il = i < n ? i : n - 1;
    const uint32_t ep = (uint32_t)a.episode[il];
    const float4 rf = a.RF[il];
    Env s;
    s.refN = rf.x; s.refE = rf.y; s.refPsi = rf.z;
    if (a.new_ref) { s.refN = a.new_ref[il]; s.refE = a.new_ref[(int64_t)n + il]; s.refPsi = a.new_ref[2 * (int64_t)n + il]; }   // ENV:131 precedes the reset
    ResetDraw d;
    reset_draw<MODE>(a, a.env_id_base + i, ep, d);
    float o[9];
    reset_apply<MODE>(a, s, d, o);
    lds[0 * 64 + lane] = s.N; lds[1 * 64 + lane] = s.E; lds[2 * 64 + lane] = s.psi;
#pragma unroll
    for (int k = 0; k < 9; ++k) lds[(3 + k) * 64 + lane] = o[k];
    lds[12 * 64 + lane] = s.pt[0]; lds[13 * 64 + lane] = s.pt[1]; lds[14 * 64 + lane] = s.pt[2];
    lds[15 * 64 + lane] = s.sn; lds[16 * 64 + lane] = s.cs;
    lds[17 * 64 + lane] = __uint_as_float(ep);      // the env wave only has to store ep + 1: no load on its tail
}

template <int MODE>
__device__ __forceinline__ void reset_from_lds(const float* lds, int lane, Env& s, float o_new[9])
{
    s.N = lds[0 * 64 + lane]; s.E = lds[1 * 64 + lane]; s.psi = lds[2 * 64 + lane];
#pragma unroll
    for (int k = 0; k < 9; ++k) o_new[k] = lds[(3 + k) * 64 + lane];
    s.u = o_new[3]; s.v = o_new[4]; s.r = o_new[5];
    s.pt[0] = lds[12 * 64 + lane]; s.pt[1] = lds[13 * 64 + lane]; s.pt[2] = lds[14 * 64 + lane];
    default_angles<MODE>(s.ang[0], s.ang[1], s.ang[2]);
    s.steps = 0;
    s.sn = lds[15 * 64 + lane]; s.cs = lds[16 * 64 + lane];
}

// VES (dpenv_dev.h VES_*): where a lane's vessel comes from - kernel arguments, the LDS-staged class table, or its own per-env block
// (straight into registers, or through an LDS image filled by LDS-DMA: the A/B SURVEY section 7 asks for, bench.py `vessel_classes.per_env`).
// (Rejected forms of this kernel - loads not hoisted above the first branch, observation rows before the state stores, no early stores,
// the state streams repeated as preloaded scalar arguments - were A/B switches until round 6: tools/ab/README.md, profiles/LAB_NOTES.md.)
template <int MODE, bool EXT, int VES, bool RESETW = false>
__global__ __launch_bounds__(RESETW ? 2 * BLOCK : BLOCK) void step_kernel(const StepArgs a)
{
    const float4 *pS0 = a.S0, *pS1 = a.S1, *pS2 = a.S2, *pRF = a.RF;
    const float* paction = a.action;
    const int pn = a.n, playout = a.action_layout;
    constexpr int A = ModeTraits<MODE>::A;
    constexpr int OD = EXT ? 9 : 6;
    constexpr bool PER_CLASS = VES == VES_CLASS_LDS;
    constexpr bool RND = VES == VES_ENV_RND;            // domain randomisation: a reset re-draws the hull (and the current)
    constexpr bool ENV_VGPR = VES == VES_ENV_VGPR || RND;
    constexpr int IL = VES == VES_ARGS_LOSS ? IL_SHARED : IL_NONE;    // (RND: the lane's own row of the table)
    constexpr bool CURR = RND || VES == VES_ARGS_LOSS;                // the forms that re-draw the current with the episode (dpenv_set_current_randomisation)
    static_assert(!RESETW || BLOCK == 64, "the reset wave pairs with ONE env wave");
    static_assert(VES != VES_ENV_LDS || BLOCK == 64, "the LDS-DMA image is [group][lane] of one wave");
    __shared__ float lds_io[BLOCK * 9];
    __shared__ float lds_cls[PER_CLASS ? VD_COUNT * MAX_CLASSES : 1];
    __shared__ float4 lds_pe[VES == VES_ENV_LDS ? ENV_GROUPS * 64 : 1];
    __shared__ float lds_rst[RESETW ? (CURR ? RESETW_FIELDS_RND : RESETW_FIELDS) * 64 : 1];
    __shared__ uint32_t lds_fin[RESETW ? 64 : 1];      // env wave -> reset wave: this env finished and is being re-drawn

    if (RESETW && threadIdx.x >= BLOCK) {
        const int lane = threadIdx.x - BLOCK;
        reset_wave<MODE>(a, lds_rst, lane, blockIdx.x);
        // domain randomisation: the hull of the episode that would start now, drawn beside the plant loop as well; the env wave says
        // which envs finished, and this wave - whose registers hold the block - puts it into the table, off the env wave's way
        float4 hull[DRAW_GROUPS];                       // the vessel block's groups and the thrust-loss table's two
        const int i = blockIdx.x * 64 + lane;
        const bool rnd = RND && a.rand_tab != nullptr;  // (the general per-env kernel also serves fixed hulls with a thrust loss)
        if (rnd)
            draw_env_groups(hull_key(a), a.env_id_base + i, __float_as_uint(lds_rst[17 * 64 + lane]), [&](int g, const float4& q) { hull[g] = q; });
        if (CURR && a.cur_nom) {                        // the current of the episode that would start now
            const float2 cd = current_draw(a, i, i < a.n ? i : a.n - 1, __float_as_uint(lds_rst[17 * 64 + lane]));
            lds_rst[18 * 64 + lane] = cd.x; lds_rst[19 * 64 + lane] = cd.y;
        }
        __syncthreads();
        if (rnd && lds_fin[lane] != 0u) {
#pragma unroll
            for (int g = 0; g < DRAW_GROUPS; ++g) store_draw_group(a.env_tab, a.env_stride, i, g, hull[g]);
        }
        return;
    }
    const int tid = threadIdx.x;
    const int i = blockIdx.x * BLOCK + tid;
    const int n = pn;
    const bool live = i < n;
    const int il = live ? i : n - 1;   // dead lanes shadow the last env and never store
#ifdef DPENV_STEP_TRACE
    const uint64_t trace_t0 = wall_clock64();
#endif

    // ---- issue all global loads up front ------------------------------------------------------
    float act[A];
    if (playout == LAYOUT_AOS) {
        const int64_t base = (int64_t)blockIdx.x * (BLOCK * A);
        load_rows<A, BLOCK>(paction + base, (int64_t)n * A - base, tid, act);
    } else {
#pragma unroll
        for (int k = 0; k < A; ++k) act[k] = paction[(int64_t)k * n + il];
    }
    Env s;
    load_env(pS0, pS1, pS2, pRF, il, s);          // (the state streams BEFORE the action rows, so that the heading arrives first: measured, 0.4 % slower;
                                                  //  the third stream - not needed before the observation - issued LAST, outside the first wait: no change)
    float nrN = 0.0f, nrE = 0.0f, nrP = 0.0f;
    if (a.new_ref) { nrN = a.new_ref[il]; nrE = a.new_ref[(int64_t)n + il]; nrP = a.new_ref[2 * (int64_t)n + il]; }
    Current cur = {0.0f, 0.0f, 0.0f, 0.0f, 0u};
    float vc0 = 0.0f, beta0 = 0.0f;
    if (a.cur_vc) {
        cur.vc = a.cur_vc[il]; cur.beta = a.cur_beta[il];
        if (a.current_drift) { vc0 = a.cur_vc0[il]; beta0 = a.cur_beta0[il]; cur.ctr = a.drift_ctr[il]; }
    }
    int cls = 0;
    if (PER_CLASS) cls = a.class_id[il];
    Vessel ve_env;
    if (ENV_VGPR) ve_env = vessel_from_env(a.env_tab, a.env_stride, il);                 // eight more 16-byte loads in the same burst
    if (VES == VES_ENV_LDS) {
        // LDS-DMA: eight global_load_lds_dwordx4, each a lane's 16 bytes -> lds_pe[g][lane] (wave-uniform base + lane x 16), no VGPR
        // destination; landed when vmcnt reaches 0 (below, where the step needs its inputs anyway)
#pragma unroll
        for (int g = 0; g < ENV_GROUPS; ++g)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.env_tab + ((int64_t)g * a.env_stride + il)),
                                             (__attribute__((address_space(3))) void*)(lds_pe + g * 64), 16, 0, 0);
    }
    // Every load of the step is in flight before the first loaded value is looked at: the sine / cosine of the heading (and of the current's
    // direction) start with a range test - a branch - and whatever load the compiler leaves behind that branch waits a whole memory round
    // trip for psi first (round 5: read off the ISA - the fourth state stream, the setpoint / current loads and the per-env block each
    // started only after an earlier s_waitcnt vmcnt had drained).
    __builtin_amdgcn_sched_barrier(0);
    sincos_lean(s.psi, s.sn, s.cs);
    if (a.cur_vc) current_components(cur);
    if (PER_CLASS) {
        for (int k = tid; k < VD_COUNT * a.n_classes; k += BLOCK) {
            const int c = k / VD_COUNT, p = k - c * VD_COUNT;   // global table is [class][param]
            lds_cls[p * a.n_classes + c] = a.class_tab[k];
        }
    }
    if (playout == LAYOUT_AOS) {
#pragma unroll
        for (int j = 0; j < A; ++j) lds_io[j * BLOCK + tid] = act[j];
    }
    if (playout == LAYOUT_AOS || PER_CLASS) lds_order<BLOCK>();
    if (playout == LAYOUT_AOS) {
#pragma unroll
        for (int k = 0; k < A; ++k) act[k] = lds_io[tid * A + k];
    }
    if (VES == VES_ENV_LDS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the DMA writes have landed in this wave's image
        __builtin_amdgcn_wave_barrier();
    }
    const Vessel ve = ENV_VGPR ? ve_env : VES == VES_ENV_LDS ? vessel_from_env_lds(lds_pe, tid)
                    : PER_CLASS ? vessel_from_lds(lds_cls, a.n_classes, cls) : vessel_from_args(a.v0);

    StepOut out;
    StepRest rest;
    env_step_chain<MODE, EXT, false>(a, ve, s, act, a.new_ref != nullptr, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, out, rest, RND ? il : IL);
    bool rf_dirty = (a.new_ref != nullptr) || (MODE == MODE_FULL);
    // Without auto-reset the state, the observation and the termination bits are final HERE, ~100 instructions (exp, three square roots, the
    // penalties) before the reward: their stores - 85 of the 89 bytes an env-step writes - go out now and travel while the reward is computed;
    // the launch ends when its last store has landed, not when it was issued (round 5: 4.930 -> 4.897 us, four interleaved runs).  Storing the
    // third state stream (commands and step counter: functions of the action alone) BEFORE the plant as well makes it 5.06 us: vmcnt counts
    // in order, so every later wait for a load then waits for that store too.  With auto-reset a finished env's state and row are the new
    // episode's: everything is stored after the re-draw, below.
#ifdef DPENV_STEP_TRACE
    const bool early = false;                              // (the diagnostic build takes its end-of-wave time stamp on the common path)
#else
    const bool early = !RESETW && !a.auto_reset;           // launch-uniform
#endif
    if (early) {
        if (live) {
            if (EXT && a.S3) a.S3[i] = make_float4(out.o[6], out.o[7], out.o[8], 0.0f);      // (see the stores below)
            store_env(a, i, s, rf_dirty);
            a.done[i] = (uint8_t)out.d;
        }
        __builtin_amdgcn_sched_barrier(0);
        store_obs<OD>(a, a.obs, out.o, i, live, lds_io);
        __builtin_amdgcn_sched_barrier(0);
    }
    env_step_finish<MODE, EXT, false>(a, s, act, rest, true, out);
    if (a.current_drift) {
        current_drift_step(a, cur, vc0, beta0, a.env_id_base + i);
        if (live) { a.cur_vc[i] = cur.vc; a.cur_beta[i] = cur.beta; a.drift_ctr[i] = cur.ctr; }
    }
    if (early) {
        if (live) {
            a.rew[i] = out.reward;
            if (a.parts) {
                a.parts[i] = out.parts[0]; a.parts[(int64_t)n + i] = out.parts[1];
                a.parts[2 * (int64_t)n + i] = out.parts[2]; a.parts[3 * (int64_t)n + i] = out.parts[3];
            }
        }
        return;
    }

    float o_next[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) o_next[k] = out.o[k];

    // ---- auto-reset (divergent, rare): the batched form of ppo.py:305-322 -------------------------
    if (RESETW) {
        lds_fin[tid] = (a.auto_reset && out.d != 0u && live) ? 1u : 0u;
        __syncthreads();                // the reset wave's results are in LDS (it finished long ago: ~250 instructions against ~1 000)
    }
    if (a.auto_reset && out.d != 0u && live) {
        if (a.final_obs) {
            // scattered 36-byte rows, only from lanes that finished an episode
            for (int k = 0; k < OD; ++k) {
                const int64_t idx = (a.obs_layout == LAYOUT_SOA) ? (int64_t)k * n + i : (int64_t)i * OD + k;
                if (a.obs_bf16) ((uint16_t*)a.final_obs)[idx] = f2bf(out.o[k]);
                else ((float*)a.final_obs)[idx] = out.o[k];
            }
        }
        const uint32_t ep = RESETW ? __float_as_uint(lds_rst[17 * 64 + tid]) : (uint32_t)a.episode[i];
        a.episode[i] = (int)(ep + 1u);
        if (RESETW) {
            reset_from_lds<MODE>(lds_rst, tid, s, o_next);
            if (CURR && a.cur_nom) {                                 // the new episode's current, drawn by the reset wave: present value and drift mean
                const float v = lds_rst[18 * 64 + tid], b = lds_rst[19 * 64 + tid];
                a.cur_vc[i] = v; a.cur_beta[i] = b; a.cur_vc0[i] = v; a.cur_beta0[i] = b;
            }
        } else {
            env_auto_reset<MODE>(a, s, a.env_id_base + i, ep, o_next);
            if (RND && a.rand_tab) redraw_vessel_table(a, i, ep);   // domain randomisation: the new episode's hull (the reset wave does this in the two-wave form)
            if (CURR && a.cur_nom) {
                const float2 cd = current_draw(a, i, i, ep);
                a.cur_vc[i] = cd.x; a.cur_beta[i] = cd.y; a.cur_vc0[i] = cd.x; a.cur_beta0[i] = cd.y;
            }
        }
        rf_dirty = true;
    }

    // ---- stores ---------------------------------------------------------------------------------
    if (live) {
        // the thrust columns of the observation just returned: a closed-loop launch that follows continues from THIS observation, not
        // from one rebuilt from the state block (which holds the command of this step, not of the one before: customEnv.py:196-205,126).
        // Only when a policy is in force (a.S3 is NULL otherwise): 16 more bytes per env-step that the plain step path does not pay.
        if (EXT && a.S3) a.S3[i] = make_float4(o_next[6], o_next[7], o_next[8], 0.0f);
        store_env(a, i, s, rf_dirty);
        a.rew[i] = out.reward;
        a.done[i] = (uint8_t)out.d;
        if (a.parts) {
            a.parts[i] = out.parts[0]; a.parts[(int64_t)n + i] = out.parts[1];
            a.parts[2 * (int64_t)n + i] = out.parts[2]; a.parts[3 * (int64_t)n + i] = out.parts[3];
        }
    }
    // the state / reward / done stores go out BEFORE the observation rows take their turn through the LDS transposition: they are in flight
    // a few dozen instructions and two LDS round trips earlier and the launch ends that much sooner (round 5: 4.950 -> 4.927 us, four interleaved runs; with the row image
    // written to LDS first and the state stores issued while it lands: 4.941)
    __builtin_amdgcn_sched_barrier(0);
    store_obs<OD>(a, a.obs, o_next, i, live, lds_io);
#ifdef DPENV_STEP_TRACE
    if (tid == 0 && g_step_trace) {
        const uint64_t t1 = wall_clock64();
        const uint32_t slot = (uint32_t)(s.steps - 1) % STEP_TRACE_RING;      // envs of a launch without resets share their counter
        uint32_t* rec = g_step_trace + ((size_t)slot * gridDim.x + blockIdx.x) * 4;
        rec[0] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                  // HW_REG_XCC_ID
        rec[1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                   // HW_REG_HW_ID
        rec[2] = (uint32_t)trace_t0;
        rec[3] = (uint32_t)t1;
    }
#endif
}

// =============================================================================================
//  fused rollout: one launch = T steps of every env, state resident in registers.
//  Same semantics as T successive step_kernel launches with actions[t] (and refs[k] passed as
//  new_ref at step switch_step[k]); per env-step only the action row is read and the
//  observation / reward / done row written (69 B instead of 177 B).
//  One wave per workgroup: the LDS transposes are wave-private, barriers are free.
// =============================================================================================
template <int MODE, bool EXT, int VES>
__global__ __launch_bounds__(RBLOCK) void rollout_kernel(const StepArgs a, const RolloutArgs ra)
{
    constexpr int A = ModeTraits<MODE>::A;
    constexpr int OD = EXT ? 9 : 6;
    constexpr bool PER_CLASS = VES == VES_CLASS_LDS, RND = VES == VES_ENV_RND, PER_ENV = VES == VES_ENV_VGPR || RND;
    constexpr int IL = VES == VES_ARGS_LOSS ? IL_SHARED : IL_NONE;
    constexpr bool CURR = RND || VES == VES_ARGS_LOSS;
    __shared__ float lds_act[RBLOCK * 7];
    __shared__ float lds_obs[RBLOCK * 9];
    __shared__ float lds_cls[PER_CLASS ? VD_COUNT * MAX_CLASSES : 1];

    const int tid = threadIdx.x;
    const int i = blockIdx.x * RBLOCK + tid;
    const int n = a.n;
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
    int cls = 0;
    if (PER_CLASS) {
        cls = a.class_id[il];
        for (int k = tid; k < VD_COUNT * a.n_classes; k += RBLOCK) {
            const int c = k / VD_COUNT, p = k - c * VD_COUNT;
            lds_cls[p * a.n_classes + c] = a.class_tab[k];
        }
        lds_order<RBLOCK>();
    }
    // per-env blocks (a.env_tab): a lane's own block, once per launch, straight into registers - for a T-step kernel the register file
    // is the staging area
    Vessel ve = PER_ENV ? vessel_from_env(a.env_tab, a.env_stride, il)
                        : (PER_CLASS ? vessel_from_lds(lds_cls, a.n_classes, cls) : vessel_from_args(a.v0));
    if (!PER_CLASS && !PER_ENV) pin_vessel_in_vgprs(ve);
    uint32_t episode = a.auto_reset ? (uint32_t)a.episode[il] : 0u;
    bool ep_dirty = false, rf_dirty = (MODE == MODE_FULL);

    const int64_t step_stride_act = (int64_t)n * A;
    const int64_t step_stride_obs = (int64_t)n * OD;
    const int64_t blk_act = (int64_t)blockIdx.x * (RBLOCK * A);
    const int64_t blk_obs = (int64_t)blockIdx.x * (RBLOCK * OD);

    // software pipeline: the action row of step t+1 is in flight while step t computes
    float pre[A];
    auto fetch = [&](int t) {
        const float* src = ra.actions + (int64_t)t * step_stride_act;   // uniform
        if (a.action_layout == LAYOUT_AOS) {
            load_rows<A, RBLOCK>(src + blk_act, step_stride_act - blk_act, tid, pre);
        } else {
#pragma unroll
            for (int k = 0; k < A; ++k) pre[k] = src[(int64_t)k * n + il];
        }
    };
    fetch(0);
    int next_switch = 0;
    bool cur_dirty = false;                      // a reset drew a new current (RND only): present value and drift mean go back at the end
    float lag[3] = {0.0f, 0.0f, 0.0f};
    for (int t = 0; t < ra.T; ++t) {
        float act[A];
        if (a.action_layout == LAYOUT_AOS) {
            lds_order<RBLOCK>();
#pragma unroll
            for (int j = 0; j < A; ++j) lds_act[j * RBLOCK + tid] = pre[j];
            lds_order<RBLOCK>();
#pragma unroll
            for (int k = 0; k < A; ++k) act[k] = lds_act[tid * A + k];
        } else {
#pragma unroll
            for (int k = 0; k < A; ++k) act[k] = pre[k];
        }
        if (t + 1 < ra.T) fetch(t + 1);

        bool has_ref = false;
        float nrN = 0.0f, nrE = 0.0f, nrP = 0.0f;
        if (next_switch < ra.n_switch && ra.switch_step[next_switch] == t) {   // wave-uniform
            const float* rp = ra.refs + (int64_t)next_switch * 3 * n;
            nrN = rp[il]; nrE = rp[(int64_t)n + il]; nrP = rp[2 * (int64_t)n + il];
            has_ref = true; rf_dirty = true;
            ++next_switch;
        }
        StepOut out;
        env_step<MODE, EXT>(a, ve, s, act, has_ref, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, out, RND ? il : IL);
        if (a.current_drift) current_drift_step(a, cur, vc0, beta0, a.env_id_base + i);
        float o_next[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) o_next[k] = out.o[k];
        if (a.auto_reset && out.d != 0u && live) {
            env_auto_reset<MODE>(a, s, a.env_id_base + i, episode, o_next);
            if (RND && a.rand_tab) redraw_vessel(a, i, episode, ve);   // domain randomisation: the new episode runs on a new hull
            if (CURR && a.cur_nom) { current_redraw_inline(a, i, episode, cur, vc0, beta0); cur_dirty = true; }  // ... in a new current
            ++episode; ep_dirty = true; rf_dirty = true;
        }
        lag[0] = o_next[6]; lag[1] = o_next[7]; lag[2] = o_next[8];
        if (live) {
            (ra.rew + (int64_t)t * n)[(unsigned)i] = out.reward;          // uniform row base + lane offset
            (ra.done + (int64_t)t * n)[(unsigned)i] = (uint8_t)out.d;
        }
        // observation row -> [T][n][OD] through the wave-private LDS transpose
        if (a.obs_layout == LAYOUT_SOA) {
            if (live) {
#pragma unroll
                for (int k = 0; k < OD; ++k) {
                    const int64_t idx = (int64_t)t * step_stride_obs + (int64_t)k * n + i;
                    if (a.obs_bf16) ((uint16_t*)ra.obs)[idx] = f2bf(o_next[k]);
                    else ((float*)ra.obs)[idx] = o_next[k];
                }
            }
        } else {
            lds_order<RBLOCK>();
#pragma unroll
            for (int k = 0; k < OD; ++k) lds_obs[tid * OD + k] = o_next[k];
            lds_order<RBLOCK>();
            store_rows<OD, RBLOCK>(ra.obs, (int64_t)t * step_stride_obs + blk_obs, step_stride_obs - blk_obs, a.obs_bf16,
                                   lds_obs, tid);
        }
    }
    if (live) {
        store_env(a, i, s, rf_dirty);
        if (EXT) a.S3[i] = make_float4(lag[0], lag[1], lag[2], 0.0f);          // thrust columns of the last observation returned (see step_kernel)
        if (ep_dirty) a.episode[i] = (int)episode;
        if (a.current_drift) { a.cur_vc[i] = cur.vc; a.cur_beta[i] = cur.beta; a.drift_ctr[i] = cur.ctr; }
        if (CURR && cur_dirty) store_current(a, i, cur, vc0, beta0, true);
    }
}

// =============================================================================================
//  fused rollout, two waves per 64 envs (round 4): an ENV wave and a ROW wave.
//
//  rollout_kernel above is bound by the instruction issue of ONE wave per SIMD: 910 VALU + the LDS transposes and row stores of every
//  step in one stream (2.65 us per step at 65 536 envs).  About a sixth of that stream needs nothing the plant loop produces until
//  the step is over, or nothing at all: the action row fetch and its transposition, the azimuth bookkeeping (two atan2), the reward,
//  the observation / reward / done row stores, the re-draw of finished envs.  Here a 128-thread workgroup owns 64 envs with two waves:
//      row wave (H)                                           env wave (E)
//      act_t+2 row: global -> LDS mailbox, post               wait act_t; a_t from the mailbox
//      wait post(t): o_t+1 (pre-reset), done bits              env_step_chain (decode, force map, plant, observation, termination)
//      commands / azimuths of a_t, reward, rows of step t      re-draw of finished envs from the prepared record
//      prepared re-draw record for the next episode            post o_t+1, done bits                       -> step t+1 at once
//  E never waits for H on its way (H runs one step behind on the rows and two ahead on the actions); the hand-over is LDS mailboxes with
//  sequence words (release / acquire at workgroup scope, no barrier in the loop).  Same device functions on the same values
//  (env_decode_cmd, env_reward, reset_draw / reset_apply, env_step_chain): every row and the final state are bit-identical to
//  rollout_kernel and to T single steps (tests).  config.step_one_wave keeps the one-wave kernel (A/B, tests).
//  (The same split for the ONE-step launch - step_ws_kernel, round 4 - was built, bit-identical and slower: a wave that lives 3 us does not
//  earn back a second wave's start-up and two hand-overs: 5.10 -> 5.19 us at 65 536 envs, profiles/r04_step_forms.txt.  Removed; dpenv_step
//  keeps one wave per 64 envs, plus the reset wave when auto-reset is on.)
// =============================================================================================
// all lanes release their mailbox rows with a workgroup-scope fence, lane 0 stores the sequence word; the waiter acquires after its poll
// loop (on gfx950: the same s_waitcnt lgkmcnt(0) as before, now outside the lane-0 branch, and nothing on the acquire side)
__device__ __forceinline__ void mb_post(int* p, int v, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void mb_wait(int* p, int v)
{
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < v)
        __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

constexpr int RW_REC = 18;              // re-draw record: N, E, psi | o[0..8] | pt[0..2] | sin psi, cos psi | the new episode's index (bits; RND only)
template <int MODE, bool EXT, int VES>
__global__ __launch_bounds__(128) void rollout_ws_kernel(const StepArgs a, const RolloutArgs ra)
{
    constexpr int A = ModeTraits<MODE>::A;
    constexpr int OD = EXT ? 9 : 6;
    constexpr bool PER_CLASS = VES == VES_CLASS_LDS, RND = VES == VES_ENV_RND, PER_ENV = VES == VES_ENV_VGPR || RND;
    constexpr int IL = VES == VES_ARGS_LOSS ? IL_SHARED : IL_NONE;
    constexpr bool CURR = RND || VES == VES_ARGS_LOSS;
    __shared__ float act_mb[2][64 * 7];          // a_t rows by step parity, [lane * A + k]
    __shared__ float post_mb[2][64 * 9];         // o_t+1 rows (pre-reset) by step parity, [lane * OD + k]: also the staged image of the row store
    __shared__ uint32_t done_mb[2][64];
    __shared__ float rec_mb[RW_REC * 64];        // the prepared re-draw, [field][lane]
    __shared__ float hull_mb[RND ? 4 * DRAW_GROUPS * 64 : 1]; // the prepared re-draw's hull (domain randomisation), [field][lane]; part of the record
    __shared__ float ang_mb[2 * 64];             // stern azimuths in force after the last step (MODE_FINAL_CONT: the row wave keeps them)
    __shared__ int seq[8];                       // [0] actions posted, [1] steps posted, [2] re-draw record version, [3] final
    __shared__ float lds_cls[PER_CLASS ? VD_COUNT * MAX_CLASSES : 1];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + lane;
    const int n = a.n;
    const bool live = i < n;
    const int il = live ? i : n - 1;
    if (threadIdx.x < 8) seq[threadIdx.x] = 0;
    int cls = 0;
    if (PER_CLASS) {
        cls = a.class_id[il];
        for (int k = threadIdx.x; k < VD_COUNT * a.n_classes; k += 128) {
            const int c = k / VD_COUNT, p = k - c * VD_COUNT;
            lds_cls[p * a.n_classes + c] = a.class_tab[k];
        }
    }
    __syncthreads();
    const int64_t step_stride_act = (int64_t)n * A;
    const int64_t step_stride_obs = (int64_t)n * OD;
    const int64_t blk_act = (int64_t)blockIdx.x * (64 * A);
    const int64_t blk_obs = (int64_t)blockIdx.x * (64 * OD);
    const bool resets = a.auto_reset != 0;

    if (wave == 1) {
        // ------------------------------------------------------------------------------------------------ row wave
        float pt[3], ang[3], refN, refE, refPsi;
        {
            const float4 s1 = a.S1[il], s2 = a.S2[il], rf = a.RF[il];
            pt[0] = s2.x; pt[1] = s2.y; pt[2] = s2.z;
            ang[0] = rf.w; ang[1] = s1.z; ang[2] = s1.w;
            refN = rf.x; refE = rf.y; refPsi = rf.z;
        }
        uint32_t episode = resets ? (uint32_t)a.episode[il] : 0u;
        bool ep_dirty = false;
        int next_switch = 0, version = 0;
        float rec_o[9], rec_pt[3];
        const bool rnd = RND && a.rand_tab != nullptr;   // domain randomisation: the record carries the hull of the episode that would start now
        // the re-draw an env would get if it finished NOW: (seed, global env id, episode) and the setpoint in force
        auto prepare = [&]() __attribute__((always_inline)) {
            Env s;
            s.refN = refN; s.refE = refE; s.refPsi = refPsi;
            ResetDraw d;
            reset_draw<MODE>(a, a.env_id_base + i, episode, d);
            reset_apply<MODE>(a, s, d, rec_o);
            if (rnd)
                draw_env_groups(hull_key(a), a.env_id_base + i, episode, [&](int g, const float4& q) {
                    hull_mb[(4 * g + 0) * 64 + lane] = q.x; hull_mb[(4 * g + 1) * 64 + lane] = q.y;
                    hull_mb[(4 * g + 2) * 64 + lane] = q.z; hull_mb[(4 * g + 3) * 64 + lane] = q.w;
                });
            rec_mb[0 * 64 + lane] = s.N; rec_mb[1 * 64 + lane] = s.E; rec_mb[2 * 64 + lane] = s.psi;
#pragma unroll
            for (int k = 0; k < 9; ++k) rec_mb[(3 + k) * 64 + lane] = rec_o[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) { rec_pt[k] = s.pt[k]; rec_mb[(12 + k) * 64 + lane] = s.pt[k]; }
            rec_mb[15 * 64 + lane] = s.sn; rec_mb[16 * 64 + lane] = s.cs;
            if (CURR) rec_mb[17 * 64 + lane] = __uint_as_float(episode);       // the env wave draws the new episode's current from it
            ++version;
            mb_post(&seq[2], version, lane);
        };
        // a setpoint handed over at step t is in force BEFORE that step's re-draw (ENV:131 precedes the reset): the record is
        // re-made with it while the env wave is still in the plant loop of step t
        auto look_ahead = [&](int t) __attribute__((always_inline)) {
            if (next_switch < ra.n_switch && ra.switch_step[next_switch] == t) {   // wave-uniform
                const float* rp = ra.refs + (int64_t)next_switch * 3 * n;
                refN = rp[il]; refE = rp[(int64_t)n + il]; refPsi = rp[2 * (int64_t)n + il];
                ++next_switch;
                if (resets) prepare();
            }
        };
        float pre[A];
        float lag[3] = {0.0f, 0.0f, 0.0f};
        auto fetch = [&](int t) __attribute__((always_inline)) {
            const float* src = ra.actions + (int64_t)t * step_stride_act;   // uniform
            if (a.action_layout == LAYOUT_AOS) {
                load_rows<A, 64>(src + blk_act, step_stride_act - blk_act, lane, pre);
            } else {
#pragma unroll
                for (int k = 0; k < A; ++k) pre[k] = src[(int64_t)k * n + il];
            }
        };
        auto post_actions = [&](int t) __attribute__((always_inline)) {     // pre holds row t
            float* mb = act_mb[t & 1];
            if (a.action_layout == LAYOUT_AOS) {
#pragma unroll
                for (int j = 0; j < A; ++j) mb[j * 64 + lane] = pre[j];      // element j * 64 + lane of the 64 x A block IS its row-major place
            } else {
#pragma unroll
                for (int k = 0; k < A; ++k) mb[lane * A + k] = pre[k];
            }
            mb_post(&seq[0], t + 1, lane);
        };
        if (resets) prepare();                                               // version 1: the first episode's successor
        look_ahead(0);
        fetch(0); post_actions(0);
        if (ra.T > 1) { fetch(1); post_actions(1); }
        if (ra.T > 2) fetch(2);
        for (int t = 0; t < ra.T; ++t) {
            mb_wait(&seq[1], t + 1);                                         // step t posted
            const float* pm = post_mb[t & 1];
            float o[9], act[A];
#pragma unroll
            for (int k = 0; k < 9; ++k) o[k] = k < OD ? pm[lane * OD + k] : 0.0f;
            const uint32_t d = done_mb[t & 1][lane];
#pragma unroll
            for (int k = 0; k < A; ++k) act[k] = act_mb[t & 1][lane * A + k];
            // ENV:102-126 for the books: commands in force before, commands of this step
            float thr[3], pt_old[3], ang_prev[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { ang_prev[k] = ang[k]; pt_old[k] = pt[k]; }
            env_decode_cmd<MODE, false>(ang, act, thr);
            StepOut out;
            env_reward<MODE, EXT>(a, o, thr, pt_old, ang, ang_prev, out);
            if (live) {
                (ra.rew + (int64_t)t * n)[(unsigned)i] = out.reward;
                (ra.done + (int64_t)t * n)[(unsigned)i] = (uint8_t)d;
            }
            const bool do_reset = resets && d != 0u && live;
            const bool any_reset = __ballot(do_reset) != 0ull;
#pragma unroll
            for (int k = 0; k < 3; ++k) lag[k] = do_reset ? rec_o[6 + k] : o[6 + k];            // thrust columns of the observation this step returns
#pragma unroll
            for (int k = 0; k < 3; ++k) pt[k] = do_reset ? rec_pt[k] : thr[k];                   // ENV:126 / ENV:190
            if (do_reset) default_angles<MODE>(ang[0], ang[1], ang[2]);
            // observation row of step t: the new episode's first observation where the env was re-drawn
            if (a.obs_layout == LAYOUT_SOA) {
                if (live) {
#pragma unroll
                    for (int k = 0; k < OD; ++k) {
                        const float v = do_reset ? rec_o[k] : o[k];
                        const int64_t idx = (int64_t)t * step_stride_obs + (int64_t)k * n + i;
                        if (a.obs_bf16) ((uint16_t*)ra.obs)[idx] = f2bf(v);
                        else ((float*)ra.obs)[idx] = v;
                    }
                }
            } else {
                if (any_reset) {
                    if (do_reset) {
#pragma unroll
                        for (int k = 0; k < OD; ++k) post_mb[t & 1][lane * OD + k] = rec_o[k];
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                store_rows<OD, 64>(ra.obs, (int64_t)t * step_stride_obs + blk_obs, step_stride_obs - blk_obs, a.obs_bf16, pm, lane);
            }
            if (any_reset) {
                if (do_reset) {
                    if (rnd) {                                               // the hull the env wave took from the record goes into the table
#pragma unroll
                        for (int g = 0; g < ENV_GROUPS; ++g)              // (the two thrust-loss rows: the ENV wave writes them, it reads them back)
                            store_draw_group(a.env_tab, a.env_stride, i, g,
                                             make_float4(hull_mb[(4 * g + 0) * 64 + lane], hull_mb[(4 * g + 1) * 64 + lane],
                                                         hull_mb[(4 * g + 2) * 64 + lane], hull_mb[(4 * g + 3) * 64 + lane]));
                    }
                    ++episode; ep_dirty = true;
                }
                prepare();                                                   // lanes whose episode did not move re-make the same record
            }
            look_ahead(t + 1);
            // a_t+2 takes the place of a_t (read by the env wave before it posted step t, and by this wave above)
            if (t + 2 < ra.T) {
                post_actions(t + 2);
                if (t + 3 < ra.T) fetch(t + 3);
            }
        }
        if (MODE == MODE_FINAL_CONT) { ang_mb[lane] = ang[1]; ang_mb[64 + lane] = ang[2]; }
        mb_post(&seq[3], 1, lane);
        if (live && ep_dirty) a.episode[i] = (int)episode;
        if (EXT && live) a.S3[i] = make_float4(lag[0], lag[1], lag[2], 0.0f);  // see step_kernel
        return;
    }

    // ---------------------------------------------------------------------------------------------------- env wave
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
    Vessel ve = PER_ENV ? vessel_from_env(a.env_tab, a.env_stride, il)
                        : (PER_CLASS ? vessel_from_lds(lds_cls, a.n_classes, cls) : vessel_from_args(a.v0));
    if (!PER_CLASS && !PER_ENV) pin_vessel_in_vgprs(ve);
    bool rf_dirty = (MODE == MODE_FULL), cur_dirty = false;
    int next_switch = 0, need = 1;               // re-draw record version this wave may read: 1 + setpoint switches so far + re-draw events so far
    for (int t = 0; t < ra.T; ++t) {
        mb_wait(&seq[0], t + 1);                                             // a_t posted (long ago: the row wave runs two steps ahead)
        float act[A];
#pragma unroll
        for (int k = 0; k < A; ++k) act[k] = act_mb[t & 1][lane * A + k];
        bool has_ref = false;
        float nrN = 0.0f, nrE = 0.0f, nrP = 0.0f;
        if (next_switch < ra.n_switch && ra.switch_step[next_switch] == t) {   // wave-uniform
            const float* rp = ra.refs + (int64_t)next_switch * 3 * n;
            nrN = rp[il]; nrE = rp[(int64_t)n + il]; nrP = rp[2 * (int64_t)n + il];
            has_ref = true; rf_dirty = true;
            ++next_switch;
            if (resets) ++need;
        }
        StepOut out;
        StepRest rest;
        env_step_chain<MODE, EXT, true>(a, ve, s, act, has_ref, nrN, nrE, nrP, a.cur_vc != nullptr, cur.vcN, cur.vcE, out, rest, RND ? il : IL);
        if (a.current_drift) current_drift_step(a, cur, vc0, beta0, a.env_id_base + i);
        const bool do_reset = resets && out.d != 0u && live;
        float* pm = post_mb[t & 1];
#pragma unroll
        for (int k = 0; k < OD; ++k) pm[lane * OD + k] = out.o[k];           // the row wave's reward is that of the TERMINAL observation
        done_mb[t & 1][lane] = out.d;
        if (__ballot(do_reset) != 0ull) {                                    // wave-uniform
            mb_wait(&seq[2], need);                                          // the record for (episode, setpoint) as they are now
            if (do_reset) {
                float o_new[9];
                reset_from_lds<MODE>(rec_mb, lane, s, o_new);
                if (RND && a.rand_tab) {                                     // the record's hull (the row wave keeps the table)
                    VesselDev hd;
#pragma unroll
                    for (int k = 0; k < VD_COUNT; ++k) hd.p[k] = hull_mb[k * 64 + lane];
                    ve = vessel_from_args(hd);
                    // the thrust-loss coefficients of the new episode: this wave reads them from the table at its next force map, so it
                    // is this wave that puts them there (a wave's own store is ahead of its own later load; the row wave runs a step behind)
#pragma unroll
                    for (int g = ENV_GROUPS; g < DRAW_GROUPS; ++g)
                        store_draw_group(a.env_tab, a.env_stride, i, g,
                                         make_float4(hull_mb[(4 * g + 0) * 64 + lane], hull_mb[(4 * g + 1) * 64 + lane],
                                                     hull_mb[(4 * g + 2) * 64 + lane], hull_mb[(4 * g + 3) * 64 + lane]));
                }
                if (CURR && a.cur_nom) { current_redraw_inline(a, i, __float_as_uint(rec_mb[17 * 64 + lane]), cur, vc0, beta0); cur_dirty = true; }
                rf_dirty = true;
            }
            ++need;
        }
        mb_post(&seq[1], t + 1, lane);                                       // o_t+1, done bits posted; the record (if read) is free again
    }
    mb_wait(&seq[3], 1);
    if (MODE == MODE_FINAL_CONT) {
        // the azimuth bookkeeping of the continuous-angle variant lives in the row wave (two atan2 per step that nothing on this
        // wave's way needs); a re-drawn env's defaults are the same on both sides
        s.ang[1] = ang_mb[lane]; s.ang[2] = ang_mb[64 + lane];
    }
    if (live) {
        store_env(a, i, s, rf_dirty);
        if (a.current_drift) { a.cur_vc[i] = cur.vc; a.cur_beta[i] = cur.beta; a.drift_ctr[i] = cur.ctr; }
        if (CURR && cur_dirty) store_current(a, i, cur, vc0, beta0, true);
    }
}

// =============================================================================================
//  env.reset  (ENV:135-194)
// =============================================================================================
template <int MODE, bool EXT>
__global__ __launch_bounds__(BLOCK) void reset_kernel(const StepArgs a, const uint8_t* mask, const float* init,
                                                      const float* ref)
{
    constexpr int OD = EXT ? 9 : 6;
    __shared__ float lds_io[BLOCK * 9];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * BLOCK + tid;
    const int n = a.n;
    const bool live = i < n;
    const int il = live ? i : n - 1;

    float4 s0 = a.S0[il], s1 = a.S1[il], s2 = a.S2[il], rf = a.RF[il];
    const bool sel = live && (mask == nullptr || mask[il] != 0);
    if (sel) {
        float eta[3], nu[3], pt[3] = {0.0f, 0.0f, 0.0f};
        // the episode counter advances with every reset that consumes random numbers (sampled pose, or drawn thrust)
        const uint32_t ep = (uint32_t)a.episode[i];
        if (!init || a.reset_acts || a.rand_tab || a.cur_nom) a.episode[i] = (int)(ep + 1u);
        if (init) {
            // explicit **init (ENV:141,152,159-161); the 50 held sub-steps (ENV:164-167) keep it in place
            for (int k = 0; k < 3; ++k) { eta[k] = init[(int64_t)k * n + i]; nu[k] = init[(int64_t)(3 + k) * n + i]; }
        } else {
            sample_reset<MODE>(a, a.env_id_base + i, ep, eta, nu);
        }
        if (a.reset_acts) sample_reset_thrust(a, a.env_id_base + i, ep, pt);   // ENV:179-188
        float ab, ap, as;
        default_angles<MODE>(ab, ap, as);
        s0 = make_float4(eta[0], eta[1], eta[2], nu[0]);
        s1 = make_float4(nu[1], nu[2], ap, as);
        s2 = make_float4(pt[0], pt[1], pt[2], __int_as_float(0));
        rf.w = ab;
        if (ref) { rf.x = ref[i]; rf.y = ref[(int64_t)n + i]; rf.z = ref[2 * (int64_t)n + i]; }
        a.S0[i] = s0; a.S1[i] = s1; a.S2[i] = s2; a.RF[i] = rf;
        // the lagged thrust columns a closed-loop launch starts from (PolicyArgs.use_lag): a new episode's observation carries the
        // reset's own previous thrust (ENV:190,196-205) - written per env, so that a MASKED reset leaves the other envs' lag alone
        a.S3[i] = make_float4(pt[0] * 0.01f, pt[1] * 0.01f, pt[2] * 0.01f, 0.0f);
        if (a.rand_tab) redraw_vessel_table(a, i, ep);      // domain randomisation: every reset starts its episode on a freshly drawn hull
        if (a.cur_nom) {                                    // ... and in a freshly drawn current
            const float2 cd = current_draw(a, i, i, ep);
            a.cur_vc[i] = cd.x; a.cur_beta[i] = cd.y; a.cur_vc0[i] = cd.x; a.cur_beta0[i] = cd.y;
        }
    }
    if (a.obs) {
        const float pt[3] = {s2.x, s2.y, s2.z};
        float o[9];
        float sr_, cr_;
        bool same_;
        make_obs(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, rf.x, rf.y, rf.z, pt, a.wrap_mode == WRAP_REFERENCE, o, sr_, cr_, same_);
        store_obs<OD>(a, a.obs, o, i, live, lds_io);
    }
}

// ---- canonical state <-> packed float4 streams (parity tests, checkpointing) ---------------------
__global__ __launch_bounds__(BLOCK) void get_state_kernel(const StepArgs a, float* st, int32_t* ctr)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const int64_t n = a.n;
    if (i >= a.n) return;
    const float4 s0 = a.S0[i], s1 = a.S1[i], s2 = a.S2[i], rf = a.RF[i];
    if (st) {
        st[0 * n + i] = s0.x; st[1 * n + i] = s0.y; st[2 * n + i] = s0.z;
        st[3 * n + i] = s0.w; st[4 * n + i] = s1.x; st[5 * n + i] = s1.y;
        st[6 * n + i] = rf.x; st[7 * n + i] = rf.y; st[8 * n + i] = rf.z;
        st[9 * n + i] = s2.x; st[10 * n + i] = s2.y; st[11 * n + i] = s2.z;
        st[12 * n + i] = rf.w; st[13 * n + i] = s1.z; st[14 * n + i] = s1.w;
    }
    if (ctr) { ctr[i] = __float_as_int(s2.w); ctr[n + i] = a.episode[i]; }
}

__global__ __launch_bounds__(BLOCK) void set_state_kernel(const StepArgs a, const float* st, const int32_t* ctr)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const int64_t n = a.n;
    if (i >= a.n) return;
    float4 s2 = a.S2[i];
    if (st) {
        a.S0[i] = make_float4(st[0 * n + i], st[1 * n + i], st[2 * n + i], st[3 * n + i]);
        a.S1[i] = make_float4(st[4 * n + i], st[5 * n + i], st[13 * n + i], st[14 * n + i]);
        a.RF[i] = make_float4(st[6 * n + i], st[7 * n + i], st[8 * n + i], st[12 * n + i]);
        s2.x = st[9 * n + i]; s2.y = st[10 * n + i]; s2.z = st[11 * n + i];
        // the lagged thrust columns a closed-loop launch starts from: what an observation rebuilt from this state carries (make_obs: pt / 100)
        a.S3[i] = make_float4(s2.x * 0.01f, s2.y * 0.01f, s2.z * 0.01f, 0.0f);
    }
    if (ctr) { s2.w = __int_as_float(ctr[i]); a.episode[i] = ctr[n + i]; }
    a.S2[i] = s2;
}

// ---- per-env parameter blocks: public parameter vectors <-> the packed float4 streams (dpenv_set / get_vessel_params) -------------
__global__ __launch_bounds__(BLOCK) void pack_env_vessels_kernel(const float* raw, int64_t p_stride, int64_t i_stride, float4* tab, uint32_t* loss_flag,
                                                                 int stride, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float r[RAND_NPARAM], d[ENV_BLOCK_FLOATS];
#pragma unroll
    for (int p = 0; p < RAND_NPARAM; ++p) r[p] = raw[p * p_stride + i * i_stride];
    derive_env_block(r, d);
    store_env_block(tab, stride, i, d);
    // the six thrust-loss coefficients: two rows behind the vessel block; the device flag tells the host whether any env has one (the
    // general per-env kernels then load and apply them: dpenv_env_dev.h).  A block that is not a vessel carries NaN coefficients too.
    const bool bad = !(d[0] == d[0]);
    const float nanv = __builtin_nanf("");
    tab[(int64_t)ENV_GROUPS * stride + i] = bad ? make_float4(nanv, nanv, nanv, nanv) : make_float4(r[26], r[27], r[28], r[29]);
    tab[(int64_t)(ENV_GROUPS + 1) * stride + i] = bad ? make_float4(nanv, nanv, 0.0f, 0.0f) : make_float4(r[30], r[31], 0.0f, 0.0f);
    bool any = false;
#pragma unroll
    for (int p = 26; p < RAND_NPARAM; ++p) any = any || (r[p] != 0.0f);
    if (loss_flag && __ballot(any) != 0ull && (threadIdx.x & 63) == __builtin_ctzll(__ballot(true))) atomicOr(loss_flag, 1u);
}

__global__ __launch_bounds__(BLOCK) void unpack_env_vessels_kernel(const float4* tab, int stride, float* out, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float d[ENV_BLOCK_FLOATS];
#pragma unroll
    for (int g = 0; g < ENV_GROUPS; ++g) {
        const float4 q = tab[(int64_t)g * stride + i];
        d[4 * g] = q.x; d[4 * g + 1] = q.y; d[4 * g + 2] = q.z; d[4 * g + 3] = q.w;
    }
    float r[32];
    r[0] = d[VD_M11]; r[1] = d[VD_M22]; r[2] = d[VD_M23]; r[3] = d[VD_M33];
    r[4] = d[VD_XU]; r[5] = d[VD_XUU]; r[6] = d[VD_YV]; r[7] = d[VD_YVV]; r[8] = d[VD_YR];
    r[9] = d[VD_NV]; r[10] = d[VD_NR]; r[11] = d[VD_NRR];
#pragma unroll
    for (int k = 0; k < 3; ++k) { r[12 + k] = d[VD_KF + k]; r[15 + k] = d[VD_KR + k]; r[18 + k] = d[VD_LX + k]; r[21 + k] = d[VD_LY + k]; }
    r[24] = d[VD_NUV]; r[25] = d[VD_YUR];
    {
        const float4 q0 = tab[(int64_t)ENV_GROUPS * stride + i], q1 = tab[(int64_t)(ENV_GROUPS + 1) * stride + i];
        r[26] = q0.x; r[27] = q0.y; r[28] = q0.z; r[29] = q0.w; r[30] = q1.x; r[31] = q1.y;
    }
#pragma unroll
    for (int p = 0; p < 32; ++p) out[(int64_t)p * n + i] = r[p];
}

// ---- stateless force map (SupervisedTau.py:42-83) -------------------------------------------------
__global__ __launch_bounds__(BLOCK) void thrust_map_kernel(const VesselDev vd, const float* n_pct, const float* alpha,
                                                           float* tau, int n)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Vessel ve = vessel_from_args(vd);
    const float nn[3] = {n_pct[i], n_pct[(int64_t)n + i], n_pct[2 * (int64_t)n + i]};
    const float al[3] = {alpha[i], alpha[(int64_t)n + i], alpha[2 * (int64_t)n + i]};
    float tx, ty, tn;
    thrust_map(ve, nn, al, tx, ty, tn);
    tau[i] = tx; tau[(int64_t)n + i] = ty; tau[2 * (int64_t)n + i] = tn;
}

// ---- GAE-lambda reverse scan (ppo.py:65-91, core.py:48-63) + advantage statistics (ppo.py:99-103, mpi_tools.py:71-92) ----
// A lane owns V adjacent env columns (V = 4: every row access is a 16-byte load / store, 1 KiB per wave-instruction) and walks
// them from t = T-1 down to 0.  The recurrence is serial in t, but none of the LOADS depends on it: rows are fetched GAE_U at a
// time into one of two register buffers while the other is being consumed, so 2 GAE_U rows of every column are in flight and
// the scan runs at memory speed instead of one HBM latency per row (round 1: 0.13 of the HBM roof).  The same pass sums adv and
// adv^2 per lane in double; a fixed-order wave reduction leaves one partial pair per workgroup and gae_finalize_kernel adds
// them in index order, so the statistics are bit-reproducible (no float atomics).

template <int V> struct GaeVec;
template <> struct GaeVec<2> { typedef float2 F; typedef uint16_t E; };
template <> struct GaeVec<1> { typedef float F; typedef uint8_t E; };

template <int V> struct GaeRow {
    typename GaeVec<V>::F r, v, b;
    typename GaeVec<V>::E e;
};

__device__ __forceinline__ float gae_get(const float2& x, int k) { return k == 0 ? x.x : x.y; }
__device__ __forceinline__ float gae_get(const float& x, int) { return x; }
__device__ __forceinline__ void gae_set(float2& x, int k, float v) { if (k == 0) x.x = v; else x.y = v; }
__device__ __forceinline__ void gae_set(float& x, int, float v) { x = v; }
__device__ __forceinline__ uint32_t gae_end(uint16_t e, int k) { return ((uint32_t)e >> (8 * k)) & 0xffu; }
__device__ __forceinline__ uint32_t gae_end(uint8_t e, int) { return e; }

__device__ __forceinline__ double wave_sum_fixed(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);      // same pairing every run
    return x;
}

template <int V, int GAE_U>
__global__ __launch_bounds__(64) void gae_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                 const uint8_t* __restrict__ end, const float* __restrict__ boot,
                                                 const float* __restrict__ last_val, int T, int n, float gamma, float lam,
                                                 float* __restrict__ adv, float* __restrict__ ret, double* __restrict__ partials)
{
    typedef typename GaeVec<V>::F F;
    typedef typename GaeVec<V>::E E;
    const int c0 = (blockIdx.x * 64 + threadIdx.x) * V;
    const bool live = c0 < n;
    const int cl = live ? c0 : 0;                      // dead lanes shadow column 0 and never store
    const float gl = gamma * lam;
    float acc[V], g[V], vnext[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { acc[k] = 0.0f; g[k] = 0.0f; vnext[k] = 0.0f; }
    double s1 = 0.0, s2 = 0.0;
    F lv;
#pragma unroll
    for (int k = 0; k < V; ++k) gae_set(lv, k, 0.0f);
    if (last_val) lv = *(const F*)(last_val + cl);

    GaeRow<V> A[GAE_U], B[GAE_U];
    auto load = [&](GaeRow<V> (&buf)[GAE_U], int j) {
#pragma unroll
        for (int u = 0; u < GAE_U; ++u) {
            int t = T - 1 - j * GAE_U - u;
            t = t < 0 ? 0 : t;                         // past the start: re-read row 0 (never consumed)
            const int64_t k = (int64_t)t * n + cl;
            buf[u].r = *(const F*)(rew + k);
            buf[u].v = *(const F*)(val + k);
            if (boot) buf[u].b = *(const F*)(boot + k);
            if (end) buf[u].e = *(const E*)(end + k); else buf[u].e = 0;
        }
    };
    auto consume = [&](const GaeRow<V> (&buf)[GAE_U], int j) {
#pragma unroll
        for (int u = 0; u < GAE_U; ++u) {
            const int t = T - 1 - j * GAE_U - u;
            if (t < 0) break;
            F a_out, r_out;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const bool is_end = (t == T - 1) || (gae_end(buf[u].e, k) != 0u);
                if (is_end) {
                    const float l = boot ? gae_get(buf[u].b, k) : ((t == T - 1) ? gae_get(lv, k) : 0.0f);
                    acc[k] = 0.0f; g[k] = l; vnext[k] = l;
                }
                const float rk = gae_get(buf[u].r, k), vk = gae_get(buf[u].v, k);
                const float delta = rk + gamma * vnext[k] - vk;
                acc[k] = delta + gl * acc[k];
                g[k] = rk + gamma * g[k];
                vnext[k] = vk;
                gae_set(a_out, k, acc[k]);
                gae_set(r_out, k, g[k]);
                s1 += (double)acc[k];
                s2 += (double)acc[k] * (double)acc[k];
            }
            if (live) {
                const int64_t k = (int64_t)t * n + c0;
                *(F*)(adv + k) = a_out;
                *(F*)(ret + k) = r_out;
            }
        }
    };
    const int nb = (T + GAE_U - 1) / GAE_U;
    load(A, 0);
    for (int j = 0; j < nb; j += 2) {
        load(B, j + 1);
        consume(A, j);
        load(A, j + 2);
        consume(B, j + 1);
    }
    if (partials) {
        if (!live) { s1 = 0.0; s2 = 0.0; }
        s1 = wave_sum_fixed(s1);
        s2 = wave_sum_fixed(s2);
        if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s1; partials[2 * blockIdx.x + 1] = s2; }
    }
}

// partial pairs -> {sum, sum of squares} in a fixed order: thread k adds partials k, k + 256, ... then a fixed tree over LDS
__global__ __launch_bounds__(256) void gae_finalize_kernel(const double* __restrict__ partials, int nparts, double* __restrict__ out)
{
    __shared__ double red[2][256];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) { a += partials[2 * k]; b += partials[2 * k + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = red[0][0]; out[1] = red[1][0]; }
}

// per-workgroup partial sums of (x - m) or (x - m)^2, float4 grid-stride reads, double accumulation, fixed order
constexpr int SUMB = 256;
__global__ __launch_bounds__(SUMB) void sum_partial_kernel(const float* __restrict__ x, int64_t count, const float* mean, int squared,
                                                           double* __restrict__ partials)
{
    __shared__ double red[SUMB / 64];
    const float m = mean ? mean[0] : 0.0f;
    double s = 0.0;
    const int64_t nvec = ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) ? count / 4 : 0;
    for (int64_t k = (int64_t)blockIdx.x * SUMB + threadIdx.x; k < nvec; k += (int64_t)gridDim.x * SUMB) {
        const float4 q = ((const float4*)x)[k];
        const float d0 = q.x - m, d1 = q.y - m, d2 = q.z - m, d3 = q.w - m;
        s += squared ? ((double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2) + (double)(d3 * d3))
                     : ((double)d0 + (double)d1 + (double)d2 + (double)d3);
    }
    for (int64_t k = nvec * 4 + (int64_t)blockIdx.x * SUMB + threadIdx.x; k < count; k += (int64_t)gridDim.x * SUMB) {
        const float d = x[k] - m;
        s += squared ? (double)(d * d) : (double)d;
    }
    s = wave_sum_fixed(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < SUMB / 64; ++w) t += red[w];
        partials[2 * blockIdx.x] = t; partials[2 * blockIdx.x + 1] = 0.0;
    }
}

__global__ __launch_bounds__(256) void sum_finalize_kernel(const double* __restrict__ partials, int nparts, float* __restrict__ out)
{
    __shared__ double red[256];
    double a = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) a += partials[2 * k];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

// adv <- (adv - mean) / (std + 1e-8)   (ppo.py:103).  STATS: mean and std from {sum, sum of squares} and the global count,
// computed per thread in double (mean = s1 / N, std = sqrt(s2 / N - mean^2)); otherwise from device float scalars.
template <bool STATS>
__global__ __launch_bounds__(SUMB) void adv_apply_kernel(float* __restrict__ x, int64_t count, const float* mean, const float* std,
                                                         const double* stats, double total_count)
{
    float m, inv;
    if (STATS) {
        const double mu = stats[0] / total_count;
        double var = stats[1] / total_count - mu * mu;
        var = var > 0.0 ? var : 0.0;
        m = (float)mu;
        inv = 1.0f / ((float)sqrt(var) + 1e-8f);
    } else {
        m = mean[0];
        inv = 1.0f / (std[0] + 1e-8f);
    }
    const int64_t nvec = ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) ? count / 4 : 0;
    for (int64_t k = (int64_t)blockIdx.x * SUMB + threadIdx.x; k < nvec; k += (int64_t)gridDim.x * SUMB) {
        float4 q = ((float4*)x)[k];
        q.x = (q.x - m) * inv; q.y = (q.y - m) * inv; q.z = (q.z - m) * inv; q.w = (q.w - m) * inv;
        ((float4*)x)[k] = q;
    }
    for (int64_t k = nvec * 4 + (int64_t)blockIdx.x * SUMB + threadIdx.x; k < count; k += (int64_t)gridDim.x * SUMB)
        x[k] = (x[k] - m) * inv;
}

// scratch of the legacy three-pass entry points (dpenv_adv_sum / dpenv_adv_sumsq): per-workgroup partials.  One buffer per
// device code object: those two calls must not run concurrently on two streams of one device (dpenv.h says so);
// dpenv_gae_stats takes an explicit workspace instead.
constexpr int SUM_MAXGRID = 2048;
__device__ double g_sum_partials[2 * SUM_MAXGRID];

}  // namespace dpenv

// =============================================================================================
//  launchers (called from dpenv_api.cpp through dpenv_dev.h)
// =============================================================================================
using namespace dpenv;

template <int MODE, int VES>
static hipError_t launch_step_ves(const StepArgs& a, bool ext, bool reset_wave, hipStream_t s)
{
    const dim3 grid((a.n + BLOCK - 1) / BLOCK), block(BLOCK);
    if (a.auto_reset && BLOCK == 64 && reset_wave) {
        // auto-reset on: a second wave per workgroup prepares the re-draws beside the plant loop (RESETW above)
        const dim3 block2(2 * BLOCK);
        if (ext) hipLaunchKernelGGL((step_kernel<MODE, true, VES, BLOCK == 64>), grid, block2, 0, s, a);
        else hipLaunchKernelGGL((step_kernel<MODE, false, VES, BLOCK == 64>), grid, block2, 0, s, a);
        return hipGetLastError();
    }
    if (ext) hipLaunchKernelGGL((step_kernel<MODE, true, VES>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((step_kernel<MODE, false, VES>), grid, block, 0, s, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_step_mode(const StepArgs& a, bool ext, int ves, bool reset_wave, hipStream_t s)
{
    switch (ves) {
    case VES_ARGS: return launch_step_ves<MODE, VES_ARGS>(a, ext, reset_wave, s);
    case VES_CLASS_LDS: return launch_step_ves<MODE, VES_CLASS_LDS>(a, ext, reset_wave, s);
    case VES_ENV_VGPR: return launch_step_ves<MODE, VES_ENV_VGPR>(a, ext, reset_wave, s);
    case VES_ENV_LDS: return launch_step_ves<MODE, (BLOCK == 64 ? VES_ENV_LDS : VES_ENV_VGPR)>(a, ext, reset_wave, s);
    case VES_ENV_RND: return launch_step_ves<MODE, VES_ENV_RND>(a, ext, reset_wave, s);
    case VES_ARGS_LOSS: return launch_step_ves<MODE, VES_ARGS_LOSS>(a, ext, reset_wave, s);
    }
    return hipErrorInvalidValue;
}

#ifdef DPENV_STEP_TRACE
extern "C" int dpenv_debug_set_step_trace(void* p)      // device buffer of STEP_TRACE_RING x workgroups x 4 dwords (or NULL)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(dpenv::g_step_trace), &p, sizeof p);
}
#endif

extern "C" hipError_t dpenv_dev_launch_step(const StepArgs* a, int mode, int ext, int ves, int reset_wave, hipStream_t s)
{
    if (ves >= VES_ENV_VGPR && ves <= VES_ENV_RND && !a->env_tab) return hipErrorInvalidValue;
    if ((ves == VES_ARGS_LOSS) != (a->loss_on == LOSS_SHARED)) return hipErrorInvalidValue;
    switch (mode) {
    case MODE_FULL: return launch_step_mode<MODE_FULL>(*a, ext, ves, reset_wave != 0, s);
    case MODE_SIMPLE: return launch_step_mode<MODE_SIMPLE>(*a, ext, ves, reset_wave != 0, s);
    case MODE_LIMITED: return launch_step_mode<MODE_LIMITED>(*a, ext, ves, reset_wave != 0, s);
    case MODE_FINAL_WRAP: return launch_step_mode<MODE_FINAL_WRAP>(*a, ext, ves, reset_wave != 0, s);
    case MODE_FINAL_CONT: return launch_step_mode<MODE_FINAL_CONT>(*a, ext, ves, reset_wave != 0, s);
    }
    return hipErrorInvalidValue;
}

extern "C" hipError_t dpenv_dev_launch_pack_env_vessels(const float* raw, int64_t p_stride, int64_t i_stride, float4* tab, uint32_t* loss_flag,
                                                        int stride, int n, hipStream_t s)
{
    hipLaunchKernelGGL(pack_env_vessels_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, raw, p_stride, i_stride, tab, loss_flag, stride, n);
    return hipGetLastError();
}

extern "C" hipError_t dpenv_dev_launch_unpack_env_vessels(const float4* tab, int stride, float* out, int n, hipStream_t s)
{
    hipLaunchKernelGGL(unpack_env_vessels_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, tab, stride, out, n);
    return hipGetLastError();
}

template <int MODE, int VES>
static hipError_t launch_rollout_ves(const StepArgs& a, const RolloutArgs& ra, bool ext, bool two_wave, hipStream_t s)
{
    const dim3 grid((a.n + RBLOCK - 1) / RBLOCK), block(RBLOCK);
    if (two_wave) {
        const dim3 block2(128);
        if (ext) hipLaunchKernelGGL((rollout_ws_kernel<MODE, true, VES>), grid, block2, 0, s, a, ra);
        else hipLaunchKernelGGL((rollout_ws_kernel<MODE, false, VES>), grid, block2, 0, s, a, ra);
        return hipGetLastError();
    }
    if (ext) hipLaunchKernelGGL((rollout_kernel<MODE, true, VES>), grid, block, 0, s, a, ra);
    else hipLaunchKernelGGL((rollout_kernel<MODE, false, VES>), grid, block, 0, s, a, ra);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_rollout_mode(const StepArgs& a, const RolloutArgs& ra, bool ext, int ves, bool two_wave, hipStream_t s)
{
    switch (ves) {
    case VES_ARGS: return launch_rollout_ves<MODE, VES_ARGS>(a, ra, ext, two_wave, s);
    case VES_CLASS_LDS: return launch_rollout_ves<MODE, VES_CLASS_LDS>(a, ra, ext, two_wave, s);
    case VES_ENV_VGPR: case VES_ENV_LDS: return launch_rollout_ves<MODE, VES_ENV_VGPR>(a, ra, ext, two_wave, s);   // a T-step kernel's staging area is the register file
    case VES_ENV_RND: return launch_rollout_ves<MODE, VES_ENV_RND>(a, ra, ext, two_wave, s);
    case VES_ARGS_LOSS: return launch_rollout_ves<MODE, VES_ARGS_LOSS>(a, ra, ext, two_wave, s);
    }
    return hipErrorInvalidValue;
}

extern "C" hipError_t dpenv_dev_launch_rollout(const StepArgs* a, const RolloutArgs* ra, int mode, int ext, int ves, int two_wave,
                                               hipStream_t s)
{
    if (ves >= VES_ENV_VGPR && ves <= VES_ENV_RND && !a->env_tab) return hipErrorInvalidValue;
    if ((ves == VES_ARGS_LOSS) != (a->loss_on == LOSS_SHARED)) return hipErrorInvalidValue;
    switch (mode) {
    case MODE_FULL: return launch_rollout_mode<MODE_FULL>(*a, *ra, ext, ves, two_wave != 0, s);
    case MODE_SIMPLE: return launch_rollout_mode<MODE_SIMPLE>(*a, *ra, ext, ves, two_wave != 0, s);
    case MODE_LIMITED: return launch_rollout_mode<MODE_LIMITED>(*a, *ra, ext, ves, two_wave != 0, s);
    case MODE_FINAL_WRAP: return launch_rollout_mode<MODE_FINAL_WRAP>(*a, *ra, ext, ves, two_wave != 0, s);
    case MODE_FINAL_CONT: return launch_rollout_mode<MODE_FINAL_CONT>(*a, *ra, ext, ves, two_wave != 0, s);
    }
    return hipErrorInvalidValue;
}

template <int MODE>
static hipError_t launch_reset_mode(const StepArgs& a, bool ext, const uint8_t* mask, const float* init, const float* ref,
                                    hipStream_t s)
{
    const dim3 grid((a.n + BLOCK - 1) / BLOCK), block(BLOCK);
    if (ext) hipLaunchKernelGGL((reset_kernel<MODE, true>), grid, block, 0, s, a, mask, init, ref);
    else hipLaunchKernelGGL((reset_kernel<MODE, false>), grid, block, 0, s, a, mask, init, ref);
    return hipGetLastError();
}

extern "C" hipError_t dpenv_dev_launch_reset(const StepArgs* a, int mode, int ext, const uint8_t* mask, const float* init,
                                             const float* ref, hipStream_t s)
{
    switch (mode) {
    case MODE_FULL: return launch_reset_mode<MODE_FULL>(*a, ext, mask, init, ref, s);
    case MODE_SIMPLE: return launch_reset_mode<MODE_SIMPLE>(*a, ext, mask, init, ref, s);
    case MODE_LIMITED: return launch_reset_mode<MODE_LIMITED>(*a, ext, mask, init, ref, s);
    case MODE_FINAL_WRAP: return launch_reset_mode<MODE_FINAL_WRAP>(*a, ext, mask, init, ref, s);
    case MODE_FINAL_CONT: return launch_reset_mode<MODE_FINAL_CONT>(*a, ext, mask, init, ref, s);
    }
    return hipErrorInvalidValue;
}

extern "C" hipError_t dpenv_dev_launch_get_state(const StepArgs* a, float* st, int32_t* ctr, hipStream_t s)
{
    hipLaunchKernelGGL(get_state_kernel, dim3((a->n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, *a, st, ctr);
    return hipGetLastError();
}

extern "C" hipError_t dpenv_dev_launch_set_state(const StepArgs* a, const float* st, const int32_t* ctr, hipStream_t s)
{
    hipLaunchKernelGGL(set_state_kernel, dim3((a->n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, *a, st, ctr);
    return hipGetLastError();
}

extern "C" hipError_t dpenv_dev_launch_thrust_map(const VesselDev* vd, const float* n_pct, const float* alpha, float* tau,
                                                  int n, hipStream_t s)
{
    hipLaunchKernelGGL(thrust_map_kernel, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, *vd, n_pct, alpha, tau, n);
    return hipGetLastError();
}

extern "C" int64_t dpenv_dev_gae_workspace_bytes(int n)
{
    return (((int64_t)n + 63) / 64) * 2 * (int64_t)sizeof(double);      // sized for the scalar form (one lane per column)
}

extern "C" hipError_t dpenv_dev_launch_gae(const float* rew, const float* val, const uint8_t* end, const float* boot,
                                           const float* last_val, int T, int n, float gamma, float lam, float* adv,
                                           float* ret, double* workspace, double* stats, hipStream_t s)
{
    // two columns per lane (8-byte rows) need n % 2 == 0 and 8-byte aligned bases (torch allocations are); anything else takes the
    // scalar form.  Measured at T = 400, n = 65 536 (tools/gae_bench.py; 21 B per env-step): two columns per lane with rows fetched
    // two groups of 8 ahead of the recurrence 104 us = 5.3 TB/s (groups of 4 / 6 / 10 / 12 / 16 / 24: 129 / 112 / 104 / 105 / 109 /
    // 115 us); four columns per lane (256 waves for 1 024 SIMDs) 118 us; one column per lane 185-198 us.
    const uintptr_t al = reinterpret_cast<uintptr_t>(rew) | reinterpret_cast<uintptr_t>(val) | reinterpret_cast<uintptr_t>(adv) |
                         reinterpret_cast<uintptr_t>(ret) | reinterpret_cast<uintptr_t>(boot) | reinterpret_cast<uintptr_t>(last_val);
    const bool vec = (n % 2 == 0) && ((al & 7u) == 0) && ((reinterpret_cast<uintptr_t>(end) & 1u) == 0);
    const int lanes = vec ? n / 2 : n;
    const int grid = (lanes + 63) / 64;
    double* parts = stats ? workspace : nullptr;
    if (vec) hipLaunchKernelGGL((gae_kernel<2, 8>), dim3(grid), dim3(64), 0, s, rew, val, end, boot, last_val, T, n, gamma, lam, adv, ret, parts);
    else hipLaunchKernelGGL((gae_kernel<1, 8>), dim3(grid), dim3(64), 0, s, rew, val, end, boot, last_val, T, n, gamma, lam, adv, ret, parts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !stats) return e;
    // the vector and the scalar form size their grids differently; the workspace is sized for the larger (scalar) one
    hipLaunchKernelGGL(gae_finalize_kernel, dim3(1), dim3(256), 0, s, (const double*)parts, grid, stats);
    return hipGetLastError();
}

static int reduce_grid(int64_t count)
{
    int64_t g = (count / 4 + SUMB - 1) / SUMB;
    if (g > SUM_MAXGRID) g = SUM_MAXGRID;   // 256 CUs x 8 blocks: grid-stride the rest
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" hipError_t dpenv_dev_launch_sum(const float* x, int64_t count, const float* mean, float* out, hipStream_t s)
{
    double* parts = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&parts, HIP_SYMBOL(g_sum_partials));
    if (e != hipSuccess) return e;
    const int grid = reduce_grid(count);
    hipLaunchKernelGGL(sum_partial_kernel, dim3(grid), dim3(SUMB), 0, s, x, count, mean, mean != nullptr ? 1 : 0, parts);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, s, (const double*)parts, grid, out);
    return hipGetLastError();
}

extern "C" hipError_t dpenv_dev_launch_adv_apply(float* x, int64_t count, const float* mean, const float* std,
                                                 const double* stats, double total_count, hipStream_t s)
{
    if (stats) hipLaunchKernelGGL(adv_apply_kernel<true>, dim3(reduce_grid(count)), dim3(SUMB), 0, s, x, count, mean, std, stats, total_count);
    else hipLaunchKernelGGL(adv_apply_kernel<false>, dim3(reduce_grid(count)), dim3(SUMB), 0, s, x, count, mean, std, stats, total_count);
    return hipGetLastError();
}
