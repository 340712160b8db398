/*
 * dpenv_oracle_impl.h - body of the CPU oracle, included twice by dpenv_oracle.c
 * with REAL / SFX defined (double/f64 and float/f32).  TEST INFRASTRUCTURE ONLY,
 * see dpenv_oracle.h for scope, reference citations and the pinning statement.
 *
 * Reference paths: ENV = src/rl/windows_workspace/specific/customEnv.py,
 * EF = .../specific/errorFrame.py, MATH = .../specific/misc/mathematics.py,
 * SIMT = .../specific/misc/simtools.py, PPO = .../spinup/algos/tf1/ppo/ppo.py,
 * CORE = .../spinup/algos/tf1/ppo/core.py, STAU = src/sl/SupervisedTau.py,
 * QPROS = src/qp/ROS/qp_allocator/src/qp_allocator.py.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

#define R(x) ((REAL)(x))
#define PI_R R(3.14159265358979323846)

/* numpy's np.mod for floats: result takes the sign of the divisor (MATH:17 uses np.mod) */
static REAL FN(np_mod)(REAL a, REAL b)
{
    REAL r = M_FMOD(a, b);
    if (r != R(0)) {
        if ((r < R(0)) != (b < R(0))) r += b;
    } else {
        r = M_COPYSIGN(R(0), b);
    }
    return r;
}

/* MATH:14-17  wrap_angle(angle, deg=True): ref = 180 if deg else pi; mod(angle+ref, 2ref) - ref.
 * The double build evaluates it literally.  The float build uses the algebraically identical
 * x - 2ref*floor((x+ref)/(2ref)) because adding 180 to a radian-sized float and subtracting it
 * again would throw away ~8 bits (the double reference does not suffer from that). */
static REAL FN(wrap_angle)(REAL angle, int deg)
{
    REAL ref = deg ? R(180.0) : PI_R;
#if WRAP_LITERAL
    return FN(np_mod)(angle + ref, R(2) * ref) - ref;
#else
    (void)FN(np_mod);
    REAL k = M_FLOOR((angle + ref) / (R(2) * ref));
    return angle - k * (R(2) * ref);
#endif
}

int FN(dpo_act_dim)(const dpo_config* c)
{
    switch (c->variant) {
    case DPO_FULL: return 6;                    /* ENV:24 */
    case DPO_SIMPLE: return 3;                  /* ENV:332 */
    case DPO_LIMITED: return 5;                 /* ENV:356 */
    default: return c->cont_ang ? 7 : 5;        /* ENV:379 */
    }
}

int FN(dpo_obs_dim)(const dpo_config* c) { return c->extended_state ? 9 : 6; } /* ENV:44 */

/* BUILD-OWNED default hull (NOT from the reference) + thruster
 * constants that ARE in the reference: K and F_max "as currently set in the simulator"
 * QPROS:51-55, lever arms QPROS:69-70 / STAU:35-36 (reordered port,star,bow -> bow,port,star). */
void FN(dpo_default_vessel)(REAL* p)
{
    for (int i = 0; i < DPO_NPARAM; ++i) p[i] = R(0);
    /* fitted to the reference's recorded Cybersea runs by tests/calibration/calibrate_plant.py */
    p[DPO_P_M11] = R(263.93); p[DPO_P_M22] = R(300.9); p[DPO_P_M23] = R(7.0); p[DPO_P_M33] = R(300.0);
    p[DPO_P_XU] = R(3.0);  p[DPO_P_XUU] = R(7.1);
    p[DPO_P_YV] = R(19.8); p[DPO_P_YVV] = R(80.3);
    p[DPO_P_YR] = R(-1.1); p[DPO_P_NV] = R(19.7);
    p[DPO_P_NR] = R(77.8); p[DPO_P_NRR] = R(24.9);
    p[DPO_P_NUV] = R(40.0); p[DPO_P_YUR] = R(30.0);
    p[DPO_P_KF_BOW] = R(0.0009); p[DPO_P_KF_PORT] = R(0.00205); p[DPO_P_KF_STAR] = R(0.00205);
    p[DPO_P_KR_BOW] = R(0.0009); p[DPO_P_KR_PORT] = R(0.00205); p[DPO_P_KR_STAR] = R(0.00205);
    p[DPO_P_LX_BOW] = R(1.08); p[DPO_P_LX_PORT] = R(-1.12); p[DPO_P_LX_STAR] = R(-1.12);
    p[DPO_P_LY_BOW] = R(0.0);  p[DPO_P_LY_PORT] = R(-0.15); p[DPO_P_LY_STAR] = R(0.15);
}

/* BUILD-OWNED second preset (the steady speeds "with thrust losses", ENV:17; tests/calibration/fit_thrust_loss_preset.py): the same hull,
 * reverse gain of the stern thrusters from the no-loss astern speed (ENV:14: -1.60 m/s), inflow-loss coefficients from +1.4 / -1.1 m/s */
void FN(dpo_thrust_loss_vessel)(REAL* p)
{
    FN(dpo_default_vessel)(p);
    p[DPO_P_KR_PORT] = p[DPO_P_KR_STAR] = R(THRUST_LOSS_KR_STERN);
    p[DPO_P_KLF_PORT] = p[DPO_P_KLF_STAR] = R(THRUST_LOSS_KLF_STERN);
    p[DPO_P_KLR_PORT] = p[DPO_P_KLR_STAR] = R(THRUST_LOSS_KLR_STERN);
    p[DPO_P_KLF_BOW] = p[DPO_P_KLR_BOW] = R(THRUST_LOSS_KL_BOW);
}

/* BUILD-OWNED third preset (round 6; tests/calibration/fit_dynpos_preset.py): the sway-yaw part of the hull refitted jointly to the default's
 * records, the 32 recorded Cybersea station-keeping runs (results/all_plots/dyn_pos/) and the recorded steady sway speed (ENV:14: 0.35 m/s) */
void FN(dpo_dynpos_fit_vessel)(REAL* p)
{
    FN(dpo_default_vessel)(p);
    p[DPO_P_M22] = R(317.3); p[DPO_P_M33] = R(300.0);
    p[DPO_P_YV] = R(21.4); p[DPO_P_YVV] = R(54.3); p[DPO_P_YR] = R(-4.9);
    p[DPO_P_NV] = R(11.4); p[DPO_P_NR] = R(57.0); p[DPO_P_NRR] = R(59.6);
    p[DPO_P_NUV] = R(40.0); p[DPO_P_YUR] = R(25.3);
}

/* per-variant action bounds ENV:63,339,362,390 */
static void FN(action_bounds)(const dpo_config* c, REAL bnd[6], int* n)
{
    bnd[0] = bnd[1] = bnd[2] = R(100);
    switch (c->variant) {
    case DPO_FULL: bnd[3] = bnd[4] = bnd[5] = PI_R; *n = 6; break;
    case DPO_SIMPLE: *n = 3; break;
    case DPO_LIMITED: bnd[3] = bnd[4] = PI_R / R(2); *n = 5; break;
    default: bnd[3] = bnd[4] = PI_R; *n = 5; break;
    }
}

/*
 * ENV:102-122.  Input: raw policy action (act_dim values) and the azimuth commands in force
 * (bow, port, star).  Output: thrust commands in percent (THR1 bow, THR2 port, THR3 star) and
 * the updated azimuth commands.  The caller keeps ang_in as prev_angles (ENV:102).
 */
void FN(dpo_decode)(const dpo_config* c, const REAL* action, const REAL ang_in[3], REAL thrust_out[3],
                    REAL ang_out[3])
{
    REAL a[6] = {0, 0, 0, 0, 0, 0};
    REAL bnd[6];
    int nb;
    FN(action_bounds)(c, bnd, &nb);
    if (c->variant == DPO_FINAL) {
        a[0] = action[0]; a[1] = action[1]; a[2] = action[2];
        if (c->cont_ang) {
            /* ENV:227-235 handle_continuous_angles: atan2(sin_head, cos_head) / bounds[3] */
            a[3] = M_ATAN2(action[3], action[4]) / bnd[3];
            a[4] = M_ATAN2(action[5], action[6]) / bnd[3];
        } else {
            /* ENV:237-244 wrap_stern_angles: wrap_angle(a*bnd, deg=False)/bnd */
#if WRAP_LITERAL
            a[3] = FN(wrap_angle)(action[3] * bnd[3], 0) / bnd[3];
            a[4] = FN(wrap_angle)(action[4] * bnd[3], 0) / bnd[3];
#else
            /* float build: same map in units of pi, a - 2*floor((a+1)/2), exact in float */
            a[3] = action[3] - R(2) * M_FLOOR((action[3] + R(1)) * R(0.5));
            a[4] = action[4] - R(2) * M_FLOOR((action[4] + R(1)) * R(0.5));
#endif
        }
    } else {
        for (int i = 0; i < nb; ++i) a[i] = action[i];
    }
    /* ENV:215-225 scale_and_clip */
    for (int i = 0; i < nb; ++i) {
        REAL v = a[i] * bnd[i];
        if (v < -bnd[i]) v = -bnd[i];
        if (v > bnd[i]) v = bnd[i];
        a[i] = v;
    }
    thrust_out[0] = a[0]; thrust_out[1] = a[1]; thrust_out[2] = a[2];
    ang_out[0] = ang_in[0]; ang_out[1] = ang_in[1]; ang_out[2] = ang_in[2];
    /* ENV:117-122 with the per-variant valid_action_indices / act_2_act_map (ENV:61,348,363-364,391-392) */
    if (c->variant == DPO_FULL) {
        ang_out[0] = a[3]; ang_out[1] = a[4]; ang_out[2] = a[5];
    } else if (c->variant == DPO_LIMITED || c->variant == DPO_FINAL) {
        ang_out[1] = a[3]; ang_out[2] = a[4];
    }
}

/* STAU:42-83 / QPROS:156-158: tau = B(alpha) F, F_i = K_i n_i |n_i| (K may differ fwd/reverse: STAU:69-71, QPROS:45-49) */
void FN(dpo_thrust_map)(const REAL* p, const REAL n[3], const REAL alpha[3], REAL tau[3])
{
    REAL tx = R(0), ty = R(0), tn = R(0);
    for (int i = 0; i < 3; ++i) {
        REAL K = (n[i] >= R(0)) ? p[DPO_P_KF_BOW + i] : p[DPO_P_KR_BOW + i];
        REAL F = K * M_FABS(n[i]) * n[i];
        REAL ca = M_COS(alpha[i]), sa = M_SIN(alpha[i]);
        tx += ca * F;
        ty += sa * F;
        tn += (p[DPO_P_LX_BOW + i] * sa - p[DPO_P_LY_BOW + i] * ca) * F;
    }
    tau[0] = tx; tau[1] = ty; tau[2] = tn;
}

/*
 * BUILD-OWNED plant (replaces ENV:124 dTwin.step(20) -> Cybersea; PARITY UNPINNED).
 * 3-DOF manoeuvring model M nu_r' + C(nu_r) nu_r + D(nu_r) nu_r = tau, eta' = R(psi) nu_r + v_c,
 * n_substeps semi-implicit Euler steps of substep_dt; heading advanced by dt*r, its sin/cos by a
 * second-order rotation re-seeded from the exact values every env step.  Ideal actuators: commanded
 * n, alpha act immediately.  current = {V_c, beta_c} (NED, constant, irrotational) or NULL.
 */
void FN(dpo_plant)(const dpo_config* c, const REAL* p, REAL eta[3], REAL nu[3], const REAL n_pct[3],
                   const REAL alpha[3], const REAL* current)
{
    const REAL h = R(c->substep_dt);
    const REAL m11 = p[DPO_P_M11], m22 = p[DPO_P_M22], m23 = p[DPO_P_M23], m33 = p[DPO_P_M33];
    const REAL inv11 = R(1) / m11;
    const REAL det = m22 * m33 - m23 * m23;
    const REAL i22 = m33 / det, i23 = -m23 / det, i33 = m22 / det;
    REAL tau[3];

    REAL N = eta[0], E = eta[1], psi = eta[2];
    REAL u = nu[0], v = nu[1], r = nu[2];
    REAL cs = M_COS(psi), sn = M_SIN(psi);
    REAL vcN = R(0), vcE = R(0);
    if (current) {
        vcN = current[0] * M_COS(current[1]);
        vcE = current[0] * M_SIN(current[1]);
        /* relative velocity nu_r = nu - R(psi)^T v_c */
        u -= cs * vcN + sn * vcE;
        v -= -sn * vcN + cs * vcE;
    }
    /* tau = B(alpha) F with the BUILD-OWNED inflow thrust loss: F_i = K_i n_i|n_i| - Kl_i |n_i| u_a,i, u_a,i the velocity through the water
     * of thruster i's position along its axis at the start of the env step (the linear open-water characteristic, Fossen 2011 eq. 9.7),
     * never past zero thrust; Kl = 0 (default hull) is the reference's law (STAU:42-83) exactly */
    {
        REAL tx = R(0), ty = R(0), tn = R(0);
        for (int i = 0; i < 3; ++i) {
            const int ahead = n_pct[i] >= R(0);
            REAL K = ahead ? p[DPO_P_KF_BOW + i] : p[DPO_P_KR_BOW + i];
            REAL Kl = ahead ? p[DPO_P_KLF_BOW + i] : p[DPO_P_KLR_BOW + i];
            REAL F = K * M_FABS(n_pct[i]) * n_pct[i];
            REAL ca = M_COS(alpha[i]), sa = M_SIN(alpha[i]);
            if (Kl != R(0)) {
                REAL ua = (u - p[DPO_P_LY_BOW + i] * r) * ca + (v + p[DPO_P_LX_BOW + i] * r) * sa;
                F -= Kl * M_FABS(n_pct[i]) * ua;
                if (ahead ? (F < R(0)) : (F > R(0))) F = R(0);
            }
            tx += ca * F;
            ty += sa * F;
            tn += (p[DPO_P_LX_BOW + i] * sa - p[DPO_P_LY_BOW + i] * ca) * F;
        }
        tau[0] = tx; tau[1] = ty; tau[2] = tn;
    }
    for (int k = 0; k < c->n_substeps; ++k) {
        REAL c13 = -(m22 * v + m23 * r);
        REAL c23 = m11 * u;
        REAL fx = tau[0] - c13 * r - (p[DPO_P_XU] + p[DPO_P_XUU] * M_FABS(u)) * u;
        REAL fy = tau[1] - c23 * r - ((p[DPO_P_YV] + p[DPO_P_YVV] * M_FABS(v)) * v + (p[DPO_P_YR] + p[DPO_P_YUR] * u) * r);
        REAL fn = tau[2] + (c13 * u + c23 * v) - ((p[DPO_P_NV] + p[DPO_P_NUV] * u) * v + (p[DPO_P_NR] + p[DPO_P_NRR] * M_FABS(r)) * r);
        u += h * (fx * inv11);
        v += h * (i22 * fy + i23 * fn);
        r += h * (i23 * fy + i33 * fn);
        N += h * (cs * u - sn * v + vcN);
        E += h * (sn * u + cs * v + vcE);
        /* heading by d = h r; its sin/cos by the second-order rotation (1 - d^2/2, d), re-seeded from cos/sin(psi) at
         * every env step (one order above the Euler integrator itself) */
        REAL d = h * r;
        REAL cd = R(1) - R(0.5) * d * d;
        psi += d;
        REAL c2 = cs * cd - sn * d;
        REAL s2 = sn * cd + cs * d;
        cs = c2; sn = s2;
    }
    if (current) {
        /* back to velocity over ground with the heading reached */
        REAL ce = M_COS(psi), se = M_SIN(psi);
        u += ce * vcN + se * vcE;
        v += -se * vcN + ce * vcE;
    }
    eta[0] = N; eta[1] = E; eta[2] = psi;
    nu[0] = u; nu[1] = v; nu[2] = r;
}

/*
 * ENV:196-205 + EF:25-32 + SIMT:52-59.  obs = [x~, y~, psi~, u, v, r, prev_thrust/100].
 * Reference mode reproduces quirk Q1: wrap_angle is called with its default deg=True on radians
 * (EF:29,31; MATH:14).  prev_thrust is the command of the PREVIOUS step (ENV:125-126, quirk Q2).
 */
void FN(dpo_obs)(const dpo_config* c, const REAL eta[3], const REAL nu[3], const REAL ref[3],
                 const REAL prev_thrust[3], REAL* obs)
{
    int deg = (c->wrap_mode == DPO_WRAP_REFERENCE);
    REAL eN = eta[0] - ref[0], eE = eta[1] - ref[1], ePsi = eta[2] - ref[2];   /* EF:28 */
    REAL rot = FN(wrap_angle)(eta[2], deg);                                     /* EF:29 */
    REAL cr = M_COS(rot), sr = M_SIN(rot);
    obs[0] = cr * eN + sr * eE;                                                 /* EF:30, MATH:7-9 transposed */
    obs[1] = -sr * eN + cr * eE;
    obs[2] = FN(wrap_angle)(ePsi, deg);                                         /* EF:31 */
    obs[3] = nu[0]; obs[4] = nu[1]; obs[5] = nu[2];                             /* SIMT:57-59 */
    if (c->extended_state) {
        obs[6] = prev_thrust[0] / R(100);                                       /* ENV:204 */
        obs[7] = prev_thrust[1] / R(100);
        obs[8] = prev_thrust[2] / R(100);
    }
}

/*
 * ENV:253-325.  parts = {vel_reward, multivariate_gaussian, thrust_penalty, action_derivative_penalty}.
 * thrust_now = the command just written (ENV:126 stores it in prev_thrust before reward() runs);
 * obs[6:9]*100 = the command before it (ENV:311).
 */
void FN(dpo_reward)(const dpo_config* c, const REAL* obs, const REAL thrust_now[3], const REAL ang_cur[3],
                    const REAL ang_prev[3], REAL parts[4])
{
    const REAL dt = R(c->substep_dt) * R(c->n_substeps);                        /* ENV:81 */
    /* ENV:267-273 with vel_rew_coeffs = [0.5,0.5,1.0] (ENV:78) */
    parts[0] = -M_SQRT(obs[3] * obs[3] * R(0.5) + obs[4] * obs[4] * R(0.5) + obs[5] * obs[5] * R(1.0));
    /* ENV:275-290; covar = diag(1^2, 5^2) (ENV:86-88) */
    {
        REAL rr = M_SQRT(obs[0] * obs[0] + obs[1] * obs[1]);
        REAL yaw = obs[2] * R(180) / PI_R;
        REAL q = rr * rr * R(1.0) + yaw * yaw * (R(1) / R(25));
        REAL multivar = R(2) * M_EXP(R(-0.5) * q);
        REAL special = M_SQRT(rr * rr + (yaw * R(0.25)) * (yaw * R(0.25)));
        REAL anti = R(1) - R(0.1) * special;
        if (anti < R(-1)) anti = R(-1);
        parts[1] = multivar + anti + R(0.5);
    }
    /* ENV:292-302 with pen_coeff [0.20,0.30,0.30] (ENV:263) */
    {
        const REAL pc[3] = {R(0.20), R(0.30), R(0.30)};
        REAL pen = R(0);
        for (int i = 0; i < 3; ++i) pen -= M_FABS(thrust_now[i]) / R(100) * pc[i];
        parts[2] = pen;
    }
    /* ENV:304-325 with pen_coeff [0.05]*3, ang_coeff [0,0.01,0.01] (ENV:263) */
    parts[3] = R(0);
    if (c->extended_state) {
        const REAL pc[3] = {R(0.05), R(0.05), R(0.05)};
        const REAL ac[3] = {R(0.00), R(0.01), R(0.01)};
        REAL pen = R(0);
        for (int i = 0; i < 3; ++i) {
            REAL dT = (thrust_now[i] - obs[6 + i] * R(100)) / dt;
            pen -= M_FABS(dT / R(100)) * pc[i];
        }
        /* bnd = real_action_bounds[4] (ENV:319): pi for full/final, pi/2 for limited; simple has no
         * such element and the reference raises IndexError (fixture simple_ext_raises_indexerror) -
         * callers reject simple+extended before reaching here. */
        REAL bnd = (c->variant == DPO_LIMITED) ? PI_R / R(2) : PI_R;
        REAL angpen = R(0);
        for (int i = 0; i < 3; ++i) {
            REAL dA = (ang_cur[i] - ang_prev[i]) / dt;      /* no wrap: quirk Q3 */
            angpen -= M_FABS(dA / bnd) * ac[i];
        }
        if (angpen < R(-1)) angpen = R(-1);
        parts[3] = pen + angpen;
    }
}

/* ENV:207-213 with per-variant real_ss_bounds ENV:26,337,361,386 (intended values, quirk Q9) */
int FN(dpo_done)(const dpo_config* c, const REAL* obs)
{
    REAL b[6] = {R(8.0), R(8.0), PI_R / R(2), R(1.4), R(0.30), R(0.52)};
    if (c->variant == DPO_SIMPLE) { b[3] = R(1.75); b[5] = R(0.51); }
    if (c->variant == DPO_LIMITED || c->variant == DPO_FINAL) b[2] = R(45) * PI_R / R(180);
    if (!c->terminate) return 0;
    for (int i = 0; i < 6; ++i)
        if (M_FABS(obs[i]) > b[i]) return 1;
    return 0;
}

static void FN(default_angles)(const dpo_config* c, REAL a[3])
{
    /* default_actions ENV:58,341-346,366-371,394-399 */
    a[0] = a[1] = a[2] = R(0);
    if (c->variant == DPO_SIMPLE) { a[0] = PI_R / R(2); a[1] = R(-3) * PI_R / R(4); a[2] = R(3) * PI_R / R(4); }
    if (c->variant == DPO_LIMITED || c->variant == DPO_FINAL) a[0] = PI_R / R(2);
}

static REAL FN(u01_sym)(uint32_t w)
{
    /* 24-bit uniform mapped to [-1, 1): exact in float and double */
    REAL U = R(w >> 8) * R(1.0 / 16777216.0);
    return R(2) * U - R(1);
}

/*
 * Training reset sampler, ENV:143-145 + SIMT:109-123: pose ~ U(+-fraction*bounds[0:3]),
 * velocity ~ U(+-0.30*fraction*bounds[3:]).  The reference draws from numpy's global RNG seeded by
 * wall-clock (quirk Q8) so only the distribution can match; the build uses Philox4x32-10 keyed by
 * seed, counter = (global env id, episode index, draw index).
 */
void FN(dpo_sample_reset)(const dpo_config* c, int64_t gid, uint32_t episode, REAL eta[3], REAL nu[3])
{
    REAL b[6] = {R(8.0), R(8.0), PI_R / R(2), R(1.4), R(0.30), R(0.52)};
    if (c->variant == DPO_SIMPLE) { b[3] = R(1.75); b[5] = R(0.51); }
    if (c->variant == DPO_LIMITED || c->variant == DPO_FINAL) b[2] = R(45) * PI_R / R(180);
    const REAL fr = R(c->reset_fraction);
    const REAL fv = R(0.30) * fr;
    uint32_t key[2] = {(uint32_t)(c->seed & 0xffffffffu), (uint32_t)(c->seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), episode, 0u};
    uint32_t w0[4], w1[4];
    dpo_philox4x32_10(ctr, key, w0);
    ctr[3] = 1u;
    dpo_philox4x32_10(ctr, key, w1);
    eta[0] = (b[0] * fr) * FN(u01_sym)(w0[0]);
    eta[1] = (b[1] * fr) * FN(u01_sym)(w0[1]);
    eta[2] = (b[2] * fr) * FN(u01_sym)(w0[2]);
    nu[0] = (b[3] * fv) * FN(u01_sym)(w0[3]);
    nu[1] = (b[4] * fv) * FN(u01_sym)(w1[0]);
    nu[2] = (b[5] * fv) * FN(u01_sym)(w1[1]);
}

/* Two standard normals from two Philox words (Box-Muller), u1 in (0, 1), u2 in [0, 1), both 24-bit: the build's normal
 * generator (current drift, exploration noise, reset_acts). */
static void FN(box_muller)(uint32_t wa, uint32_t wb, REAL* z0, REAL* z1)
{
    const REAL u1 = (R(wa >> 8) + R(0.5)) * R(1.0 / 16777216.0);
    const REAL u2 = R(wb >> 8) * R(1.0 / 16777216.0);
    const REAL rad = M_SQRT(R(-2) * M_LOG(u1));
    const REAL ang = R(2) * PI_R * u2;
    *z0 = rad * M_COS(ang);
    *z1 = rad * M_SIN(ang);
}

/*
 * Exploration noise of the policy: the reference samples pi = mu + N(0,1) exp(log_std) inside the TF graph (core.py:85) from an
 * unseeded-by-design stream (quirk Q8), so only the distribution can match; the build draws xi for action number `draw` of env
 * `gid` from Philox keyed by the seed with counter (global env id, draw, tag 0xA0000000 | block), four normals per block.
 */
void FN(dpo_policy_noise)(const dpo_config* c, int64_t gid, uint32_t draw, int32_t adim, REAL* xi)
{
    uint32_t key[2] = {(uint32_t)(c->seed & 0xffffffffu), (uint32_t)(c->seed >> 32)};
    REAL z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < (adim > 4 ? 2 : 1); ++b) {
        uint32_t ctr[4] = {(uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), draw, 0xA0000000u | (uint32_t)b};
        uint32_t w[4];
        dpo_philox4x32_10(ctr, key, w);
        FN(box_muller)(w[0], w[1], &z[4 * b], &z[4 * b + 1]);
        FN(box_muller)(w[2], w[3], &z[4 * b + 2], &z[4 * b + 3]);
    }
    for (int k = 0; k < adim && k < 8; ++k) xi[k] = z[k];
}

/*
 * Build-defined ("M, D as per-env SoA parameter arrays (domain randomisation)"): the hull of episode `episode` of env
 * `gid`.  rand_tab = { nominal[32] | relative half-range[32] } in the public parameter order; parameter p = nominal[p] * (1 + range[p] * u),
 * u uniform in [-1, 1) with 16 bits, taken from four Philox4x32-10 blocks keyed by the seed with counter (global env id, episode,
 * tag 0x48000000 | block).  The 16 bits of a parameter are half (q & 1) of word (q & 7) >> 1 of block q >> 3 for its slot q: the
 * parameters numbered in the order the kernels pack a per-env block (m11 m22 m23 m33 Xu Klr[3] | Xuu Yv Yvv Yr Nv Nr Nrr Nuv | Yur Kf Kr lx_bow |
 * lx_port lx_star ly Klf[3]), so that one Philox block fills two float4 groups there.  The parameters it randomises are the constants the
 * reference hard-codes once for its one vessel (QPROS:51-55,69-70, STAU:35-36,69-71) and the build-owned hull terms.
 */
static const int FN(RAND_SLOT)[32] = {0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 13, 14, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 15, 16,
                                      29, 30, 31, 5, 6, 7};      /* ... Klf[3] (slots 29-31), Klr[3] (5-7) */

void FN(dpo_draw_vessel)(const dpo_config* c, const REAL* rand_tab, int64_t gid, uint32_t episode, REAL* p)
{
    uint32_t key[2] = {(uint32_t)(c->seed & 0xffffffffu), (uint32_t)(c->seed >> 32)};
    uint32_t w[4][4];
    for (int b = 0; b < 4; ++b) {
        uint32_t ctr[4] = {(uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), episode, 0x48000000u | (uint32_t)b};
        dpo_philox4x32_10(ctr, key, w[b]);
    }
    for (int k = 0; k < DPO_NPARAM; ++k) p[k] = R(0);
    for (int k = 0; k < 32; ++k) {
        const int q = FN(RAND_SLOT)[k];
        const uint32_t h16 = (w[q >> 3][(q & 7) >> 1] >> (16 * (q & 1))) & 0xffffu;
        const REAL u = R(h16) * R(1.0 / 32768.0) - R(1);
        const REAL sc = R(1) + rand_tab[32 + k] * u;
        p[k] = rand_tab[k] * sc;
    }
}

/*
 * Build-defined (round 6; config 5 widened): per-episode randomisation of the current.  cur_rand = { range_v, range_b | nominal V_c [n] |
 * nominal beta_c [n] }: the current of episode `episode` of env `gid` is V_c = max(0, V_nom + range_v u1), beta_c = beta_nom + range_b u2,
 * u1, u2 = u01_sym of words 0, 1 of Philox4x32-10 keyed by the seed with counter (global env id, episode, tag 3) - beside the pose sample
 * (tags 0, 1) and the reset thrust (tag 2).  The reference's one operating point: 0.2 m/s towards 135 deg (current_box_test/plot_pos.py:78).
 */
void FN(dpo_draw_current)(const dpo_config* c, int64_t gid, uint32_t episode, REAL nom_v, REAL nom_b, REAL range_v, REAL range_b, REAL* out)
{
    uint32_t key[2] = {(uint32_t)(c->seed & 0xffffffffu), (uint32_t)(c->seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), episode, 3u};
    uint32_t w[4];
    dpo_philox4x32_10(ctr, key, w);
    const REAL pv = range_v * FN(u01_sym)(w[0]);
    const REAL pb = range_b * FN(u01_sym)(w[1]);
    const REAL v = nom_v + pv;
    out[0] = v > R(0) ? v : R(0);
    out[1] = nom_b + pb;
}

static void FN(reset_one)(const dpo_config* c, int32_t n, int32_t i, REAL* state, int32_t* counters,
                          const REAL* init, const REAL* ref, REAL* vessel_env, const REAL* rand_tab,
                          REAL* current, REAL* current_mean, const REAL* cur_rand)
{
    REAL eta[3], nu[3], ang[3], pt[3] = {R(0), R(0), R(0)};
    /* the episode counter advances with every reset that consumes random numbers (sampled pose, or drawn thrust) */
    const uint32_t ep = (uint32_t)counters[n + i];
    const int redraw_current = cur_rand && current;
    if (!init || c->reset_acts || (rand_tab && vessel_env) || redraw_current) counters[n + i] += 1;
    if (redraw_current) {
        /* every reset starts its episode in a freshly drawn current: present value and the mean the drift reverts to */
        REAL d[2];
        FN(dpo_draw_current)(c, c->env_id_base + i, ep, cur_rand[2 + i], cur_rand[2 + n + i], cur_rand[0], cur_rand[1], d);
        current[i] = d[0]; current[n + i] = d[1];
        if (current_mean) { current_mean[i] = d[0]; current_mean[n + i] = d[1]; }
    }
    if (rand_tab && vessel_env) {
        /* domain randomisation: every reset starts its episode on a freshly drawn hull */
        REAL pv[DPO_NPARAM];
        FN(dpo_draw_vessel)(c, rand_tab, c->env_id_base + i, ep, pv);
        for (int k = 0; k < DPO_NPARAM; ++k) vessel_env[(int64_t)k * n + i] = pv[k];
    }
    if (init) {
        /* explicit **init (ENV:141,152,159-161) */
        for (int k = 0; k < 3; ++k) { eta[k] = init[k * n + i]; nu[k] = init[(3 + k) * n + i]; }
    } else {
        FN(dpo_sample_reset)(c, c->env_id_base + i, ep, eta, nu);
    }
    if (c->reset_acts) {
        /* ENV:179-188: action[0:3] = np.random.normal(0, 0.1); scale_and_clip -> x 100, clip to +-100; thrust only.
         * Draw keyed (seed; global env id, episode, tag 2) next to the pose / velocity draws (quirk Q8: distribution only). */
        const int64_t gid = c->env_id_base + i;
        uint32_t key[2] = {(uint32_t)(c->seed & 0xffffffffu), (uint32_t)(c->seed >> 32)};
        uint32_t ctr[4] = {(uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), ep, 2u};
        uint32_t w[4];
        REAL z[4];
        dpo_philox4x32_10(ctr, key, w);
        FN(box_muller)(w[0], w[1], &z[0], &z[1]);
        FN(box_muller)(w[2], w[3], &z[2], &z[3]);
        for (int k = 0; k < 3; ++k) {
            REAL a = (R(0.1) * z[k]) * R(100);
            pt[k] = a > R(100) ? R(100) : (a < R(-100) ? R(-100) : a);
        }
    }
    /* the 50 held sub-steps with StateResetOn (ENV:164-167) leave the written state in place */
    for (int k = 0; k < 3; ++k) { state[(DPO_S_N + k) * n + i] = eta[k]; state[(DPO_S_U + k) * n + i] = nu[k]; }
    if (ref) for (int k = 0; k < 3; ++k) state[(DPO_S_REF_N + k) * n + i] = ref[k * n + i];
    FN(default_angles)(c, ang);
    for (int k = 0; k < 3; ++k) {
        state[(DPO_S_PT_BOW + k) * n + i] = pt[k];       /* ENV:190, or ENV:179-188 with reset_acts */
        state[(DPO_S_A_BOW + k) * n + i] = ang[k];       /* ENV:173-177,192 */
    }
    counters[i] = 0;
}

static void FN(obs_of_state)(const dpo_config* c, int32_t n, int32_t i, const REAL* state, REAL* obs)
{
    REAL eta[3], nu[3], ref[3], pt[3];
    for (int k = 0; k < 3; ++k) {
        eta[k] = state[(DPO_S_N + k) * n + i]; nu[k] = state[(DPO_S_U + k) * n + i];
        ref[k] = state[(DPO_S_REF_N + k) * n + i]; pt[k] = state[(DPO_S_PT_BOW + k) * n + i];
    }
    FN(dpo_obs)(c, eta, nu, ref, pt, obs);
}

/* ENV:135-194.  mask NULL = all envs; init [6][n] SoA or NULL = sample; ref [3][n] or NULL = keep. */
void FN(dpo_reset)(const dpo_config* c, int32_t n, REAL* state, int32_t* counters, const uint8_t* mask,
                   const REAL* init, const REAL* ref, REAL* obs, REAL* vessel_env, const REAL* rand_tab,
                   REAL* current, REAL* current_mean, const REAL* cur_rand)
{
    const int od = FN(dpo_obs_dim)(c);
    for (int32_t i = 0; i < n; ++i) {
        if (!mask || mask[i]) FN(reset_one)(c, n, i, state, counters, init, ref, vessel_env, rand_tab, current, current_mean, cur_rand);
        if (obs) FN(obs_of_state)(c, n, i, state, obs + (int64_t)i * od);
    }
}

/*
 * ENV:92-133 for n independent envs.  action [n][act_dim]; new_ref [3][n] or NULL;
 * plant_override [6][n] (eta, nu after the plant step; replaces dpo_plant - used to replay the
 * scripted plant of the golden fixtures) or NULL; current [2][n] or NULL; obs [n][obs_dim];
 * parts [n][4] or NULL; final_obs [n][obs_dim] or NULL (terminal obs of envs that auto-reset).
 * done bits: 1 = is_terminal, 2 = time limit reached (ppo.py:304), 4 = non-finite state.
 */
/* Build-defined (config 5): first-order Gauss-Markov drift of one env's current, applied once per env
 * step AFTER the plant step: x <- x + (dt/tau)(x0 - x) + sigma sqrt(2 dt/tau) xi, xi ~ N(0,1) by Box-Muller on a
 * Philox draw keyed (seed; global env id, draw index, tag 0xC0000000). */
static void FN(current_drift)(const dpo_config* c, int64_t gid, uint32_t* ctr, REAL* vc, REAL* beta, REAL vc0, REAL beta0)
{
    const REAL dt = R(c->substep_dt) * R(c->n_substeps);
    const REAL a = dt / R(c->current_tau);
    const REAL sv = R(c->current_sigma_v) * M_SQRT(R(2) * dt / R(c->current_tau));
    const REAL sb = R(c->current_sigma_beta) * M_SQRT(R(2) * dt / R(c->current_tau));
    uint32_t key[2] = {(uint32_t)(c->seed & 0xffffffffu), (uint32_t)(c->seed >> 32)};
    uint32_t ctrv[4] = {(uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), *ctr, 0xC0000000u};
    uint32_t w[4];
    dpo_philox4x32_10(ctrv, key, w);
    *ctr += 1u;
    REAL zc, zs;
    FN(box_muller)(w[0], w[1], &zc, &zs);
    *vc = *vc + a * (vc0 - *vc) + sv * zc;
    *beta = *beta + a * (beta0 - *beta) + sb * zs;
}

void FN(dpo_step)(const dpo_config* c, const REAL* vessel, int32_t n, REAL* state, int32_t* counters,
                  const REAL* action, const REAL* new_ref, const REAL* plant_override, REAL* current,
                  REAL* obs, REAL* rew, uint8_t* done, REAL* parts_out, REAL* final_obs, REAL* current_mean,
                  uint32_t* drift_ctr, REAL* vessel_env, const REAL* rand_tab, const REAL* cur_rand)
{
    /* cur_rand { range_v, range_b | nominal V_c [n] | nominal beta_c [n] } or NULL: an auto-reset re-draws the env's current (after the drift
     * step of the episode that ended: the new episode starts exactly on the drawn values) */
    /* vessel_env [DPO_NPARAM][n] or NULL: every env's OWN parameter vector (the kernels' per-env blocks) instead of the shared `vessel`;
     * rand_tab { nominal[32] | range[32] } or NULL: domain randomisation - an auto-reset re-draws the env's column of vessel_env */
    const int ad = FN(dpo_act_dim)(c), od = FN(dpo_obs_dim)(c);
    /* envs are independent (trainer.py:61-75: one simulator per env); threads only split the loop */
#pragma omp parallel for schedule(static) if (n >= 4096)
    for (int32_t i = 0; i < n; ++i) {
        REAL eta[3], nu[3], ref[3], pt[3], ang_prev[3], ang_cur[3], thrust[3], parts[4];
        REAL* o = obs + (int64_t)i * od;
        for (int k = 0; k < 3; ++k) {
            eta[k] = state[(DPO_S_N + k) * n + i]; nu[k] = state[(DPO_S_U + k) * n + i];
            ref[k] = state[(DPO_S_REF_N + k) * n + i]; pt[k] = state[(DPO_S_PT_BOW + k) * n + i];
            ang_prev[k] = state[(DPO_S_A_BOW + k) * n + i];               /* ENV:102 */
        }
        FN(dpo_decode)(c, action + (int64_t)i * ad, ang_prev, thrust, ang_cur);   /* ENV:104-122 */
        if (plant_override) {
            for (int k = 0; k < 3; ++k) { eta[k] = plant_override[k * n + i]; nu[k] = plant_override[(3 + k) * n + i]; }
        } else {
            REAL cur[2];
            if (current) { cur[0] = current[i]; cur[1] = current[n + i]; }
            REAL pv[DPO_NPARAM];
            if (vessel_env) for (int k = 0; k < DPO_NPARAM; ++k) pv[k] = vessel_env[(int64_t)k * n + i];
            FN(dpo_plant)(c, vessel_env ? pv : vessel, eta, nu, thrust, ang_cur, current ? cur : (const REAL*)0);   /* ENV:124 */
        }
        if (current && c->current_drift && current_mean && drift_ctr)
            FN(current_drift)(c, c->env_id_base + i, &drift_ctr[i], &current[i], &current[n + i], current_mean[i],
                              current_mean[n + i]);
        FN(dpo_obs)(c, eta, nu, ref, pt, o);                               /* ENV:125 */
        FN(dpo_reward)(c, o, thrust, ang_cur, ang_prev, parts);            /* ENV:126-128 */
        uint8_t d = (uint8_t)FN(dpo_done)(c, o);                           /* ENV:129 */
        int finite = 1;
        for (int k = 0; k < 3; ++k) finite = finite && isfinite(eta[k]) && isfinite(nu[k]);
        for (int k = 0; k < ad; ++k) finite = finite && isfinite(action[(int64_t)i * ad + k]);   /* a non-finite action is a fault too */
        if (!finite) d |= 5;
        if (new_ref) for (int k = 0; k < 3; ++k) ref[k] = new_ref[k * n + i];   /* ENV:131, quirk Q4 */
        counters[i] += 1;
        if (c->max_ep_len > 0 && counters[i] >= c->max_ep_len) d |= 2;      /* ppo.py:304 */
        rew[i] = parts[0] + parts[1] + parts[2] + parts[3];                /* ENV:263 */
        done[i] = d;
        if (parts_out) for (int k = 0; k < 4; ++k) parts_out[(int64_t)i * 4 + k] = parts[k];
        for (int k = 0; k < 3; ++k) {
            state[(DPO_S_N + k) * n + i] = eta[k]; state[(DPO_S_U + k) * n + i] = nu[k];
            state[(DPO_S_REF_N + k) * n + i] = ref[k];
            state[(DPO_S_PT_BOW + k) * n + i] = thrust[k];                 /* ENV:126 */
            state[(DPO_S_A_BOW + k) * n + i] = ang_cur[k];
        }
        if (c->auto_reset && d) {
            /* ppo.py:305-322: finished envs are reset and the next policy input is the reset obs */
            if (final_obs) for (int k = 0; k < od; ++k) final_obs[(int64_t)i * od + k] = o[k];
            FN(reset_one)(c, n, i, state, counters, (const REAL*)0, (const REAL*)0, vessel_env, rand_tab, current, current_mean, cur_rand);
            FN(obs_of_state)(c, n, i, state, o);
        }
    }
}

/* CORE:48-63: y[t] = x[t] + discount * y[t+1] (scipy.signal.lfilter on the reversed vector) */
void FN(dpo_discount_cumsum)(const REAL* x, int32_t n, REAL discount, REAL* y)
{
    REAL acc = R(0);
    for (int32_t t = n - 1; t >= 0; --t) { acc = x[t] + discount * acc; y[t] = acc; }
}

/*
 * PPO:65-91 batched over n env columns of a [T][n] rollout.  A path ends after step t of env i where
 * end[t][i] != 0 (and always after T-1).  The bootstrap value appended at a path end (PPO:82-83) is
 * boot[t][i] when boot != NULL, else 0 for inner ends and last_val[i] (or 0) for the final row.
 * deltas = r_t + gamma v_{t+1} - v_t; adv = discount_cumsum(deltas, gamma*lam);
 * ret = discount_cumsum(rews + [last_val], gamma)[:-1].
 */
void FN(dpo_gae)(const REAL* rew, const REAL* val, const uint8_t* end, const REAL* boot, const REAL* last_val,
                 int32_t T, int32_t n, REAL gamma, REAL lam, REAL* adv, REAL* ret)
{
    for (int32_t i = 0; i < n; ++i) {
        REAL a = R(0), g = R(0), vnext = R(0);
        for (int32_t t = T - 1; t >= 0; --t) {
            int64_t k = (int64_t)t * n + i;
            int is_end = (t == T - 1) || (end && end[k]);
            if (is_end) {
                REAL lv = boot ? boot[k] : ((t == T - 1 && last_val) ? last_val[i] : R(0));
                a = R(0); g = lv; vnext = lv;
            }
            REAL delta = rew[k] + gamma * vnext - val[k];
            a = delta + (gamma * lam) * a;
            g = rew[k] + gamma * g;
            adv[k] = a; ret[k] = g;
            vnext = val[k];
        }
    }
}

/* PPO:99-103 + mpi_tools.py:71-92 (single rank): mean, std (population), adv = (adv-mean)/(std+1e-8) */
void FN(dpo_normalize_adv)(REAL* adv, int64_t count, REAL* mean_std)
{
    double s = 0.0;
    for (int64_t i = 0; i < count; ++i) s += (double)adv[i];
    REAL mean = R(s / (double)count);
    double q = 0.0;
    for (int64_t i = 0; i < count; ++i) { double d = (double)adv[i] - (double)mean; q += d * d; }
    REAL std = R(sqrt(q / (double)count));
    for (int64_t i = 0; i < count; ++i) adv[i] = (adv[i] - mean) / (std + R(1e-8));
    if (mean_std) { mean_std[0] = mean; mean_std[1] = std; }
}

#undef CAT_
#undef CAT
#undef FN
#undef R
#undef PI_R
