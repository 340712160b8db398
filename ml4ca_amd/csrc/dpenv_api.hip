// dpenv_api.hip - host side of libdpenv.so: the C ABI declared in include/dpenv.h.
// Owns the per-env state block in HBM, validates arguments, fills the kernel argument block and
// launches on the caller's stream.  No CPU compute path exists here by design: if HIP or the
// device is unavailable every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/dpenv.h"
#include "dpenv_dev.h"

using namespace dpenv;

struct dpenv_s {
    dpenv_config cfg;
    StepArgs args;          // persistent part of the kernel argument block
    int mode;
    int n_classes;
    float* blob;            // one allocation: S0 | S1 | S2 | RF | episode | vc | beta | class_id | class_tab
    size_t blob_bytes;
    float* cur_vc;
    float* cur_beta;
    float* cur_vc0;
    float* cur_beta0;
    uint32_t* drift_ctr;
    int32_t* class_id;
    float4* env_tab;        // per-env parameter blocks + thrust-loss rows ET[DRAW_GROUPS][env_stride] (dpenv_dev.h), always allocated
    int env_stride;
    uint32_t* loss_flag;    // device word BEHIND the table (ET[DRAW_GROUPS][stride] | flag): the packing kernel reports "some env has a thrust-loss
                            //   coefficient" through it; while the host does not know the answer (LOSS_UNKNOWN) the general per-env kernels run with
                            //   the loss applied: zero coefficients are neutral bit for bit
    int loss_state;         // LOSS_OFF / LOSS_ON / LOSS_UNKNOWN -> StepArgs.loss_on (per-env blocks in force only)
    bool loss_pending;      // the flag word is on its way to loss_host behind loss_ev: resolved (without blocking a capture) by the next launch
    hipEvent_t loss_ev;
    uint32_t* loss_host;    // pinned host word
    bool rand_loss;         // the randomisation's nominal hull has a thrust-loss coefficient (then every drawn hull has one)
    bool shared_loss;       // the SINGLE class of dpenv_create has thrust-loss coefficients: kernels with the shared-loss instantiation take them from
                            //   the arguments (StepArgs.kl, VES_ARGS_LOSS); for the others env_tab holds that hull for every env while no
                            //   per-env table is in force (repacked from raw0_dev when the caller returns to the class)
    float* raw0_dev;        // device copy of raw0
    bool cur_rand;          // per-episode randomisation of the current (dpenv_set_current_randomisation): needs per-env blocks in force
    float* cur_nom;         // device float[2][env_stride]: nominal V_c | beta_c of every env
    float* rand_tab;        // device float[RAND_TAB_FLOATS]: nominal | relative half-range of the domain randomisation
    float rand_host[RAND_TAB_FLOATS];   // its host image (the source of the stream-ordered upload must outlive the call)
    float raw0[DPENV_NPARAM];           // public parameter vector of class 0 (the randomisation's default nominal hull)
    bool per_env;           // the kernels take every env's vessel from env_tab (dpenv_set_vessel_params / dpenv_set_vessel_randomisation)
    bool randomise;         // every reset re-draws the env's hull
    bool classes_assigned;
    bool current_set;
    uint32_t* noise_ctr;
    PolicyArgs pol;         // persistent part (weights, std) of the policy kernel arguments
    void* pol_buf;          // fragments | bias tiles | constants, as the kernels stage them: TWO images of pol_buf_bytes / 2 each,
    size_t pol_buf_bytes;   //   written alternately, so that an upload never touches the image the launches before it read
    int pol_slot;           // image the NEXT alternating upload writes (0 / 1)
    int pol_cur;            // image the LAST upload wrote = the one launches read
    hipEvent_t pol_read[2]; // recorded behind the last launch that read image k: the upload that reuses it waits for that
    bool pol_read_valid[2];
    hipStream_t pol_read_stream[2];   // the stream that event was last recorded on (a reader on another stream chains behind it)
    int pol_pinned;         // >= 0: a captured graph reads this image (PolicyArgs.frags is baked into the graph's kernel nodes by value):
                            //   every later upload is written IN PLACE into it, so that a replay sees the latest weights
    PolicyArgs pol_pin_layout;   // ... and the kernel node ALSO holds the image's layout and the launch form by value (nent, nblk, ks, n_hidden,
                            //   act, leak, split, critic_f16, ws, ws_groups): what an in-place upload must not change (layout_differs)
    int n_cus;
    float* pol_raw;         // device staging of raw fp32 weights for the host-pointer form of set_policy
    size_t pol_raw_bytes;
    int pol_form;           // DPENV_LAUNCH_* requested
    bool has_policy;
    bool lag_valid;         // the state block was last written by a closed-loop launch (PolicyArgs.use_lag)
    int device;
    std::string err;
};

enum { LOSS_OFF = 0, LOSS_ON = 1, LOSS_UNKNOWN = 2 };

static thread_local std::string g_create_err;

static int fail(dpenv_handle h, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_err = buf;
    return code;
}

// Entry points launch on the handle's device even if the caller's current device is another one.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int want)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != want) switched = (hipSetDevice(want) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

#define HIP_TRY(h, expr)                                                                               \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(h, DPENV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int dpenv_abi_version(void) { return DPENV_ABI_VERSION; }

extern "C" int dpenv_default_config(dpenv_config* c)
{
    if (!c) return DPENV_EINVAL;
    std::memset(c, 0, sizeof *c);
    c->struct_size = (uint32_t)sizeof *c;
    c->n_envs = 0;
    c->device = -1;
    c->variant = DPENV_FINAL;      // train.py:47 'final'
    c->extended_state = 1;         // train.py:52
    c->cont_ang = 1;               // train.py:54
    c->n_substeps = 20;            // customEnv.py:79-80
    c->substep_dt = 0.01f;         // customEnv.py:81
    c->wrap_mode = DPENV_WRAP_REFERENCE;
    c->terminate = 1;
    c->max_ep_len = 400;           // customEnv.py:83 with max_ep_len=800, n_steps=20
    c->auto_reset = 0;
    c->action_layout = DPENV_AOS;
    c->obs_layout = DPENV_AOS;
    c->obs_dtype = DPENV_F32;
    c->current_enabled = 0;
    c->seed = 0;
    c->env_id_base = 0;
    c->reset_fraction = 0.8f;      // customEnv.py:135
    c->current_drift = 0;
    c->current_tau = 100.0f;                           // SURVEY 8(d) config 5 (build-defined)
    c->current_sigma_v = 0.02f;
    c->current_sigma_beta = 5.0f * 3.14159265358979f / 180.0f;
    c->reset_acts = 0;             // customEnv.py:30 reset_acts=False
    return DPENV_OK;
}

// BUILD-OWNED hull + the reference's thruster constants
// (qp_allocator.py:51-55 K "as currently set in the simulator", :69-70 lever arms; env order bow,port,star).
extern "C" int dpenv_default_vessel(float* p)
{
    if (!p) return DPENV_EINVAL;
    for (int i = 0; i < DPENV_NPARAM; ++i) p[i] = 0.0f;
    // fitted to the reference's recorded Cybersea runs by tests/calibration/calibrate_plant.py
    p[DPENV_P_M11] = 263.93f; p[DPENV_P_M22] = 300.9f; p[DPENV_P_M23] = 7.0f; p[DPENV_P_M33] = 300.0f;
    p[DPENV_P_XU] = 3.0f;  p[DPENV_P_XUU] = 7.1f;
    p[DPENV_P_YV] = 19.8f; p[DPENV_P_YVV] = 80.3f;
    p[DPENV_P_YR] = -1.1f; p[DPENV_P_NV] = 19.7f;
    p[DPENV_P_NR] = 77.8f; p[DPENV_P_NRR] = 24.9f;
    p[DPENV_P_NUV] = 40.0f; p[DPENV_P_YUR] = 30.0f;
    p[DPENV_P_KF_BOW] = 0.0009f; p[DPENV_P_KF_PORT] = 0.00205f; p[DPENV_P_KF_STAR] = 0.00205f;
    p[DPENV_P_KR_BOW] = 0.0009f; p[DPENV_P_KR_PORT] = 0.00205f; p[DPENV_P_KR_STAR] = 0.00205f;
    p[DPENV_P_LX_BOW] = 1.08f; p[DPENV_P_LX_PORT] = -1.12f; p[DPENV_P_LX_STAR] = -1.12f;
    p[DPENV_P_LY_BOW] = 0.0f;  p[DPENV_P_LY_PORT] = -0.15f; p[DPENV_P_LY_STAR] = 0.15f;
    return DPENV_OK;
}

// The same hull with the thrust gains of the reference's SECOND set of steady full-thrust speeds - "with thrust losses" +1.4 / -1.1 m/s
// ahead / astern (customEnv.py:17), which are also the velocity bounds it trains with (customEnv.py:26) - derived by
// tests/calibration/fit_thrust_loss_preset.py as an INFLOW loss of the stern thrusters, F = K n|n| - Kl |n| u_a (u_a: the water's speed
// along the thruster axis; never past zero thrust), their reverse gain from the no-loss astern speed (-1.60 m/s, customEnv.py:14); bow
// unchanged (sway 0.29 m/s against the recorded 0.30); the hull is untouched, so the free-drift record is reproduced as before.  Yaw comes
// out at 0.505 rad/s against the recorded 0.52 (a constant gain reduced to meet +1.4 m/s - round 5's first form of this preset - gave 0.35).
// Not the default: the loss code lives in the general per-env kernels only (DESIGN.md section 3 for what it costs and what it changes).
extern "C" int dpenv_default_vessel_ex(int32_t kind, float* p)
{
    if (!p || kind < 0 || kind > (DPENV_VESSEL_THRUST_LOSS | DPENV_VESSEL_DYNPOS_FIT)) return DPENV_EINVAL;
    dpenv_default_vessel(p);
    if (kind & DPENV_VESSEL_DYNPOS_FIT) {
        // tests/calibration/fit_dynpos_preset.py (round 6): the sway-yaw part of the hull refitted JOINTLY to what the default is fitted to (free
        // drift, box test, steady surge / yaw speeds) AND to the reference's 32 recorded station-keeping runs in a current from 16 directions
        // (results/all_plots/dyn_pos/) AND to the recorded steady sway speed 0.35 m/s (customEnv.py:14), which the default hull misses (0.29)
        p[DPENV_P_M22] = 317.3f; p[DPENV_P_M33] = 300.0f;
        p[DPENV_P_YV] = 21.4f; p[DPENV_P_YVV] = 54.3f; p[DPENV_P_YR] = -4.9f;
        p[DPENV_P_NV] = 11.4f; p[DPENV_P_NR] = 57.0f; p[DPENV_P_NRR] = 59.6f;
        p[DPENV_P_NUV] = 40.0f; p[DPENV_P_YUR] = 25.3f;
    }
    if (kind & DPENV_VESSEL_THRUST_LOSS) {
        // tests/calibration/fit_thrust_loss_preset.py: stern reverse gain from -1.60 m/s astern without losses, inflow-loss coefficients from
        // +1.4 / -1.1 m/s with losses (customEnv.py:14,17); the bow thruster keeps its gain and has no loss.  (Surge only: the same numbers on
        // either hull - Xu, Xuu and m11 are not part of the dyn_pos fit.)
        p[DPENV_P_KR_PORT] = p[DPENV_P_KR_STAR] = 0.001149f;
        p[DPENV_P_KLF_PORT] = p[DPENV_P_KLF_STAR] = 0.08173f;
        p[DPENV_P_KLR_PORT] = p[DPENV_P_KLR_STAR] = 0.05039f;
    }
    return DPENV_OK;
}

static int mode_of(const dpenv_config* c)
{
    switch (c->variant) {
    case DPENV_FULL: return MODE_FULL;
    case DPENV_SIMPLE: return MODE_SIMPLE;
    case DPENV_LIMITED: return MODE_LIMITED;
    case DPENV_FINAL: return c->cont_ang ? MODE_FINAL_CONT : MODE_FINAL_WRAP;
    }
    return -1;
}

extern "C" int dpenv_act_dim(const dpenv_config* c)
{
    if (!c) return DPENV_EINVAL;
    switch (mode_of(c)) {
    case MODE_FULL: return 6;          // customEnv.py:24
    case MODE_SIMPLE: return 3;        // :332
    case MODE_LIMITED: return 5;       // :356
    case MODE_FINAL_WRAP: return 5;    // :379
    case MODE_FINAL_CONT: return 7;
    }
    return DPENV_EINVAL;
}

extern "C" int dpenv_obs_dim(const dpenv_config* c)
{
    if (!c) return DPENV_EINVAL;
    return c->extended_state ? 9 : 6;   // customEnv.py:44
}

// allow_loss: the caller deals with the inflow thrust-loss coefficients (parameters 26-31), which are not part of a VesselDev
static int derive_vessel(const float* p, VesselDev* d, std::string* why, bool allow_loss = false)
{
    const double m11 = p[DPENV_P_M11], m22 = p[DPENV_P_M22], m23 = p[DPENV_P_M23], m33 = p[DPENV_P_M33];
    const double det = m22 * m33 - m23 * m23;
    if (!(m11 > 0.0) || !(det > 0.0) || !(m22 > 0.0)) {
        *why = "mass matrix is not positive definite";
        return DPENV_EINVAL;
    }
    for (int i = 0; i < DPENV_NPARAM; ++i)
        if (!std::isfinite(p[i])) { *why = "vessel parameter is not finite"; return DPENV_EINVAL; }
    d->p[VD_M11] = (float)m11; d->p[VD_M22] = (float)m22; d->p[VD_M23] = (float)m23;
    // same float operations as the fp32 oracle so that both sides integrate with identical constants
    const float fm11 = (float)m11, fm22 = (float)m22, fm23 = (float)m23, fm33 = (float)m33;
    const float fdet = fm22 * fm33 - fm23 * fm23;
    d->p[VD_INV11] = 1.0f / fm11;
    d->p[VD_I22] = fm33 / fdet; d->p[VD_I23] = -fm23 / fdet; d->p[VD_I33] = fm22 / fdet;
    d->p[VD_XU] = p[DPENV_P_XU]; d->p[VD_XUU] = p[DPENV_P_XUU]; d->p[VD_YV] = p[DPENV_P_YV];
    d->p[VD_YVV] = p[DPENV_P_YVV]; d->p[VD_YR] = p[DPENV_P_YR]; d->p[VD_NV] = p[DPENV_P_NV];
    d->p[VD_NR] = p[DPENV_P_NR]; d->p[VD_NRR] = p[DPENV_P_NRR];
    d->p[VD_NUV] = p[DPENV_P_NUV]; d->p[VD_YUR] = p[DPENV_P_YUR];
    for (int i = 0; i < 3; ++i) {
        d->p[VD_KF + i] = p[DPENV_P_KF_BOW + i]; d->p[VD_KR + i] = p[DPENV_P_KR_BOW + i];
        d->p[VD_LX + i] = p[DPENV_P_LX_BOW + i]; d->p[VD_LY + i] = p[DPENV_P_LY_BOW + i];
        if (!(p[DPENV_P_KLF_BOW + i] >= 0.0f) || !(p[DPENV_P_KLR_BOW + i] >= 0.0f)) { *why = "thrust-loss coefficients must be >= 0"; return DPENV_EINVAL; }
        if (!allow_loss && (p[DPENV_P_KLF_BOW + i] != 0.0f || p[DPENV_P_KLR_BOW + i] != 0.0f)) {
            *why = "thrust-loss coefficients (parameters 26-31) are not available to vessel CLASSES: one class, per-env blocks (dpenv_set_vessel_params) or the nominal hull of dpenv_set_vessel_randomisation";
            return DPENV_EINVAL;
        }
    }
    return DPENV_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" int dpenv_create(const dpenv_config* cfg, const float* vessel_params, int32_t n_classes, dpenv_handle* out)
{
    if (!cfg || !out) return fail(nullptr, DPENV_EINVAL, "dpenv_create: NULL argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(dpenv_config))
        return fail(nullptr, DPENV_EINVAL, "dpenv_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size,
                    sizeof(dpenv_config));
    if (cfg->n_envs <= 0) return fail(nullptr, DPENV_EINVAL, "n_envs must be positive");
    const int mode = mode_of(cfg);
    if (mode < 0) return fail(nullptr, DPENV_EINVAL, "unknown variant %d", cfg->variant);
    if (cfg->cont_ang && cfg->variant != DPENV_FINAL)
        return fail(nullptr, DPENV_EINVAL, "cont_ang is only defined for the final variant (customEnv.py:228)");
    if (cfg->variant == DPENV_SIMPLE && cfg->extended_state)
        return fail(nullptr, DPENV_EINVAL,
                    "simple + extended_state is not runnable in the reference either (IndexError at customEnv.py:319)");
    if (cfg->n_substeps <= 0 || !(cfg->substep_dt > 0.0f)) return fail(nullptr, DPENV_EINVAL, "bad sub-step settings");
    if (cfg->action_layout < 0 || cfg->action_layout > 1 || cfg->obs_layout < 0 || cfg->obs_layout > 1)
        return fail(nullptr, DPENV_EINVAL, "bad layout");
    if (cfg->obs_dtype != DPENV_F32 && cfg->obs_dtype != DPENV_BF16) return fail(nullptr, DPENV_EINVAL, "bad obs_dtype");
    if (cfg->max_ep_len < 0) return fail(nullptr, DPENV_EINVAL, "max_ep_len < 0");
    if (cfg->current_drift && (!cfg->current_enabled || !(cfg->current_tau > 0.0f) || cfg->current_sigma_v < 0.0f ||
                               cfg->current_sigma_beta < 0.0f))
        return fail(nullptr, DPENV_EINVAL, "current_drift needs current_enabled, tau > 0 and non-negative sigmas");
    if (vessel_params == nullptr) n_classes = 1;
    if (n_classes < 1 || n_classes > MAX_CLASSES)
        return fail(nullptr, DPENV_EINVAL, "n_classes must be in [1, %d]", MAX_CLASSES);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, DPENV_ENODEV, "no HIP device available (%s); libdpenv has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    int dev = cfg->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) return fail(nullptr, DPENV_ENODEV, "hipGetDevice failed");
    } else if (dev >= ndev) {
        return fail(nullptr, DPENV_ENODEV, "device %d out of range (%d devices)", dev, ndev);
    }
    if (hipSetDevice(dev) != hipSuccess) return fail(nullptr, DPENV_ENODEV, "hipSetDevice(%d) failed", dev);

    dpenv_s* h = new (std::nothrow) dpenv_s();
    if (!h) return fail(nullptr, DPENV_ENOMEM, "host allocation failed");
    h->cfg = *cfg;
    h->mode = mode;
    h->n_classes = n_classes;
    h->device = dev;
    h->classes_assigned = false;
    h->per_env = false;
    h->randomise = false;
    h->loss_state = LOSS_OFF;
    h->loss_pending = false;
    h->loss_ev = nullptr;
    h->loss_host = nullptr;
    h->rand_loss = false;
    h->shared_loss = false;
    h->cur_rand = false;
    h->current_set = false;
    h->pol_buf = nullptr; h->pol_buf_bytes = 0;
    h->pol_slot = 0;
    h->pol_cur = 0;
    h->pol_read[0] = h->pol_read[1] = nullptr;
    h->pol_read_valid[0] = h->pol_read_valid[1] = false;
    h->pol_read_stream[0] = h->pol_read_stream[1] = nullptr;
    h->pol_pinned = -1;
    {
        hipDeviceProp_t prop;
        h->n_cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    h->pol_raw = nullptr; h->pol_raw_bytes = 0;
    h->pol_form = DPENV_LAUNCH_AUTO;
    h->has_policy = false;
    h->lag_valid = false;
    std::memset(&h->pol, 0, sizeof h->pol);

    VesselDev tab[MAX_CLASSES];
    float defp[DPENV_NPARAM];
    dpenv_default_vessel(defp);
    for (int c = 0; c < n_classes; ++c) {
        std::string why;
        const float* p = vessel_params ? vessel_params + (size_t)c * DPENV_NPARAM : defp;
        if (derive_vessel(p, &tab[c], &why, n_classes == 1) != DPENV_OK) {
            delete h;
            return fail(nullptr, DPENV_EINVAL, "vessel class %d: %s", c, why.c_str());
        }
        if (c == 0) std::memcpy(h->raw0, p, sizeof h->raw0);
    }

    const size_t n = (size_t)cfg->n_envs;
    const size_t npad = align_up(n, 256);
    size_t off = 0;
    const size_t o_s0 = off; off += npad * 16;
    const size_t o_s1 = off; off += npad * 16;
    const size_t o_s2 = off; off += npad * 16;
    const size_t o_rf = off; off += npad * 16;
    const size_t o_ep = off; off += npad * 4;
    const size_t o_vc = off; off += npad * 4;
    const size_t o_be = off; off += npad * 4;
    const size_t o_v0 = off; off += npad * 4;
    const size_t o_b0 = off; off += npad * 4;
    const size_t o_dc = off; off += npad * 4;
    const size_t o_ci = off; off += npad * 4;
    const size_t o_nc = off; off += npad * 4;
    const size_t o_s3 = off; off += npad * 16;
    const size_t o_ct = off; off += align_up(sizeof(VesselDev) * MAX_CLASSES, 256);
    const size_t o_rt = off; off += align_up(sizeof(float) * RAND_TAB_FLOATS, 256);
    const size_t o_r0 = off; off += 256;                                 // raw0_dev
    const size_t o_cn = off; off += npad * 4 * 2;                        // cur_nom
    const size_t o_et = off; off += npad * 16 * DRAW_GROUPS + 256;      // the per-env table and, behind it, its thrust-loss flag word
    h->blob_bytes = off;
    void* blob = nullptr;
    e = hipMalloc(&blob, off);
    if (e != hipSuccess) {
        delete h;
        return fail(nullptr, DPENV_ENOMEM, "hipMalloc(%zu) failed: %s", off, hipGetErrorString(e));
    }
    h->blob = (float*)blob;
    char* b = (char*)blob;
    e = hipEventCreateWithFlags(&h->loss_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->loss_host, sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        if (h->loss_ev) (void)hipEventDestroy(h->loss_ev);
        (void)hipFree(blob);
        delete h;
        return fail(nullptr, DPENV_EHIP, "event / pinned word of the per-env table's flag: %s", hipGetErrorString(e));
    }
    *h->loss_host = 0u;
    e = hipMemset(blob, 0, off);
    if (e == hipSuccess) e = hipMemcpy(b + o_ct, tab, sizeof(VesselDev) * n_classes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipEventDestroy(h->loss_ev);
        (void)hipHostFree(h->loss_host);
        (void)hipFree(blob);
        delete h;
        return fail(nullptr, DPENV_EHIP, "state initialisation failed: %s", hipGetErrorString(e));
    }

    StepArgs& a = h->args;
    std::memset(&a, 0, sizeof a);
    a.S0 = (float4*)(b + o_s0); a.S1 = (float4*)(b + o_s1); a.S2 = (float4*)(b + o_s2); a.RF = (float4*)(b + o_rf);
    a.episode = (int32_t*)(b + o_ep);
    h->cur_vc = (float*)(b + o_vc); h->cur_beta = (float*)(b + o_be);
    h->cur_vc0 = (float*)(b + o_v0); h->cur_beta0 = (float*)(b + o_b0);
    h->drift_ctr = (uint32_t*)(b + o_dc);
    h->class_id = (int32_t*)(b + o_ci);
    h->noise_ctr = (uint32_t*)(b + o_nc);
    h->env_tab = (float4*)(b + o_et);
    h->env_stride = (int)npad;
    h->rand_tab = (float*)(b + o_rt);
    h->raw0_dev = (float*)(b + o_r0);
    h->cur_nom = (float*)(b + o_cn);
    h->loss_flag = (uint32_t*)(b + o_et + npad * 16 * DRAW_GROUPS);     // = env_tab + DRAW_GROUPS * env_stride: where thrust_loss_on() looks (dpenv_env_dev.h)
    a.class_tab = (const float*)(b + o_ct);
    a.n_classes = n_classes;
    a.v0 = tab[0];
    a.n = cfg->n_envs;
    a.n_substeps = cfg->n_substeps;
    a.h = cfg->substep_dt;
    a.inv_dt = 1.0f / (cfg->substep_dt * (float)cfg->n_substeps);
    a.max_ep_len = cfg->max_ep_len;
    a.terminate = cfg->terminate;
    a.auto_reset = cfg->auto_reset;
    a.wrap_mode = cfg->wrap_mode;
    a.action_layout = cfg->action_layout;
    a.obs_layout = cfg->obs_layout;
    a.obs_bf16 = (cfg->obs_dtype == DPENV_BF16);
    a.hold_plant = cfg->hold_plant;
    {
        const float dt = cfg->substep_dt * (float)cfg->n_substeps;
        a.current_drift = cfg->current_drift;
        a.drift_a = cfg->current_drift ? dt / cfg->current_tau : 0.0f;
        a.drift_sv = cfg->current_drift ? cfg->current_sigma_v * sqrtf(2.0f * dt / cfg->current_tau) : 0.0f;
        a.drift_sb = cfg->current_drift ? cfg->current_sigma_beta * sqrtf(2.0f * dt / cfg->current_tau) : 0.0f;
    }
    a.seed_lo = (uint32_t)(cfg->seed & 0xffffffffu);
    a.seed_hi = (uint32_t)(cfg->seed >> 32);
    a.env_id_base = cfg->env_id_base;
    a.reset_fraction = cfg->reset_fraction;
    a.reset_acts = cfg->reset_acts ? 1 : 0;
    a.noise_ctr = h->noise_ctr;
    a.S3 = (float4*)(b + o_s3);
    {
        // a single class WITH thrust-loss coefficients (the thrust-loss preset): the step / fused-rollout kernels and the two-wave closed loop of
        // the shipped configuration have an instantiation that takes hull AND coefficients from the kernel arguments (VES_ARGS_LOSS, round 6);
        // the remaining kernels (one-wave closed loop, other env variants' two-wave forms) carry the loss code in their general per-env form
        // only - for them the table holds this hull for every env
        bool any = false;
        for (int p = 0; p < 3; ++p) {
            a.kl[p] = h->raw0[DPENV_P_KLF_BOW + p]; a.kl[3 + p] = h->raw0[DPENV_P_KLR_BOW + p];
            any = any || a.kl[p] != 0.0f || a.kl[3 + p] != 0.0f;
        }
        e = hipMemcpy(h->raw0_dev, h->raw0, sizeof h->raw0, hipMemcpyHostToDevice);
        if (e == hipSuccess) {       // the table starts as the image of class 0 for every env (what the kernels without a shared form read)
            e = dpenv_dev_launch_pack_env_vessels(h->raw0_dev, 1, 0, h->env_tab, nullptr, h->env_stride, cfg->n_envs, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (e != hipSuccess) {
            (void)hipEventDestroy(h->loss_ev);
            (void)hipHostFree(h->loss_host);
            (void)hipFree(blob);
            delete h;
            return fail(nullptr, DPENV_EHIP, "the class hull's device copy / per-env image: %s", hipGetErrorString(e));
        }
        h->shared_loss = any && n_classes == 1;
    }
    *out = h;
    return DPENV_OK;
}

extern "C" int dpenv_destroy(dpenv_handle h)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    if (h->blob) (void)hipFree(h->blob);
    if (h->pol_buf) (void)hipFree(h->pol_buf);
    if (h->pol_raw) (void)hipFree(h->pol_raw);
    for (int k = 0; k < 2; ++k) if (h->pol_read[k]) (void)hipEventDestroy(h->pol_read[k]);
    if (h->loss_ev) (void)hipEventDestroy(h->loss_ev);
    if (h->loss_host) (void)hipHostFree(h->loss_host);
    delete h;
    return DPENV_OK;
}

extern "C" const char* dpenv_last_error(dpenv_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int dpenv_set_reset_fraction(dpenv_handle h, float fraction)
{
    if (!h) return DPENV_EINVAL;
    if (!(fraction >= 0.0f) || !std::isfinite(fraction)) return fail(h, DPENV_EINVAL, "bad reset fraction");
    h->cfg.reset_fraction = fraction;
    h->args.reset_fraction = fraction;
    return DPENV_OK;
}

extern "C" int dpenv_set_vessel_class(dpenv_handle h, const int32_t* class_id, dpenv_stream s)
{
    if (!h || !class_id) return fail(h, DPENV_EINVAL, "dpenv_set_vessel_class: NULL argument");
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, hipMemcpyAsync(h->class_id, class_id, sizeof(int32_t) * (size_t)h->cfg.n_envs, hipMemcpyDeviceToDevice,
                              (hipStream_t)s));
    h->classes_assigned = true;
    return DPENV_OK;
}

extern "C" int dpenv_set_current(dpenv_handle h, const float* vc, const float* beta, dpenv_stream s)
{
    if (!h || !vc || !beta) return fail(h, DPENV_EINVAL, "dpenv_set_current: NULL argument");
    DeviceGuard dev_guard(h->device);
    if (!h->cfg.current_enabled) return fail(h, DPENV_EINVAL, "config.current_enabled is 0");
    const size_t bytes = sizeof(float) * (size_t)h->cfg.n_envs;
    HIP_TRY(h, hipMemcpyAsync(h->cur_vc, vc, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    HIP_TRY(h, hipMemcpyAsync(h->cur_beta, beta, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    HIP_TRY(h, hipMemcpyAsync(h->cur_vc0, vc, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    HIP_TRY(h, hipMemcpyAsync(h->cur_beta0, beta, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    h->current_set = true;
    return DPENV_OK;
}

extern "C" int dpenv_set_current_present(dpenv_handle h, const float* vc, const float* beta, dpenv_stream s)
{
    if (!h || !vc || !beta) return fail(h, DPENV_EINVAL, "dpenv_set_current_present: NULL argument");
    DeviceGuard dev_guard(h->device);
    if (!h->cfg.current_enabled) return fail(h, DPENV_EINVAL, "config.current_enabled is 0");
    const size_t bytes = sizeof(float) * (size_t)h->cfg.n_envs;
    HIP_TRY(h, hipMemcpyAsync(h->cur_vc, vc, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    HIP_TRY(h, hipMemcpyAsync(h->cur_beta, beta, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_get_current(dpenv_handle h, float* vc_out, float* beta_out, dpenv_stream s)
{
    if (!h || !vc_out || !beta_out) return fail(h, DPENV_EINVAL, "dpenv_get_current: NULL argument");
    DeviceGuard dev_guard(h->device);
    if (!h->cfg.current_enabled) return fail(h, DPENV_EINVAL, "config.current_enabled is 0");
    const size_t bytes = sizeof(float) * (size_t)h->cfg.n_envs;
    HIP_TRY(h, hipMemcpyAsync(vc_out, h->cur_vc, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    HIP_TRY(h, hipMemcpyAsync(beta_out, h->cur_beta, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_get_current_mean(dpenv_handle h, float* vc_out, float* beta_out, dpenv_stream s)
{
    if (!h || !vc_out || !beta_out) return fail(h, DPENV_EINVAL, "dpenv_get_current_mean: NULL argument");
    DeviceGuard dev_guard(h->device);
    if (!h->cfg.current_enabled) return fail(h, DPENV_EINVAL, "config.current_enabled is 0");
    const size_t bytes = sizeof(float) * (size_t)h->cfg.n_envs;
    HIP_TRY(h, hipMemcpyAsync(vc_out, h->cur_vc0, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    HIP_TRY(h, hipMemcpyAsync(beta_out, h->cur_beta0, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_set_current_randomisation(dpenv_handle h, const float* vc_nominal, const float* beta_nominal, float vc_range,
                                               float beta_range, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    if (!h->cfg.current_enabled) return fail(h, DPENV_EINVAL, "config.current_enabled is 0");
    if (!(vc_range >= 0.0f) || !(beta_range >= 0.0f) || !std::isfinite(vc_range) || !std::isfinite(beta_range))
        return fail(h, DPENV_EINVAL, "the half-ranges must be finite and >= 0");
    if (vc_range == 0.0f && beta_range == 0.0f) {             // off: every env keeps the current (and the drift mean) it has
        h->cur_rand = false;
        h->args.cur_range_v = h->args.cur_range_b = 0.0f;
        return DPENV_OK;
    }
    if (!h->per_env && h->n_classes > 1)
        return fail(h, DPENV_EINVAL, "the re-draw has no vessel-class form: with vessel classes give every env its block first (dpenv_set_vessel_params)");
    const size_t bytes = sizeof(float) * (size_t)h->cfg.n_envs;
    // NULL: the means in force - unless the randomisation is already on (then the means are drawn values: the nominals stay what they are)
    if (vc_nominal || !h->cur_rand)
        HIP_TRY(h, hipMemcpyAsync(h->cur_nom, vc_nominal ? vc_nominal : h->cur_vc0, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    if (beta_nominal || !h->cur_rand)
        HIP_TRY(h, hipMemcpyAsync(h->cur_nom + h->env_stride, beta_nominal ? beta_nominal : h->cur_beta0, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    h->cur_rand = true;
    h->args.cur_range_v = vc_range;
    h->args.cur_range_b = beta_range;
    return DPENV_OK;
}

// The packing kernel's flag word is on its way to the host (dpenv_set_vessel_params): take it if it has arrived; wait for it unless the launch
// that asks is being RECORDED into a graph - a recorded launch (and every launch after a setter that was itself recorded) runs the general
// per-env kernels with the table's coefficients applied (LOSS_TABLE): where no env has one they are zeros, which leave every row as it is
static void resolve_loss(dpenv_handle h, hipStream_t s)
{
    if (!h->loss_pending) return;
    hipError_t q = hipEventQuery(h->loss_ev);
    if (q == hipErrorNotReady) {
        (void)hipGetLastError();                                        // (not an error: the launch below must not report it)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (cap != hipStreamCaptureStatusNone) return;
        q = hipEventSynchronize(h->loss_ev);
    }
    if (q != hipSuccess) { (void)hipGetLastError(); return; }            // unknown stays unknown: the loss stays applied
    h->loss_state = *(volatile uint32_t*)h->loss_host != 0u ? LOSS_ON : LOSS_OFF;
    h->loss_pending = false;
}

static void bind_optional(dpenv_handle h, StepArgs& a, hipStream_t s)
{
    // current_enabled without dpenv_set_current = zero current (the block is zero-initialised)
    a.cur_vc = h->cfg.current_enabled ? h->cur_vc : nullptr;
    a.cur_beta = h->cfg.current_enabled ? h->cur_beta : nullptr;
    a.cur_vc0 = h->cur_vc0;
    a.cur_beta0 = h->cur_beta0;
    a.drift_ctr = h->drift_ctr;
    a.class_id = h->class_id;
    resolve_loss(h, s);
    a.env_tab = h->per_env ? h->env_tab : nullptr;
    a.env_stride = h->env_stride;
    // the per-episode current re-draw lives in the general per-env kernels and in the shared training form (one class)
    const bool cr = h->cur_rand && h->cfg.current_enabled && (h->per_env || h->n_classes == 1);
    a.loss_on = h->per_env ? (h->loss_state == LOSS_OFF ? (int)LOSS_NONE : (int)LOSS_TABLE) : ((h->shared_loss || cr) ? (int)LOSS_SHARED : (int)LOSS_NONE);
    a.rand_tab = (h->per_env && h->randomise) ? h->rand_tab : nullptr;
    a.cur_nom = cr ? h->cur_nom : nullptr;
    a.cur_nom_stride = h->env_stride;
}

// a kernel without a shared training form (the one-wave closed loop; the two-wave closed loop of other env variants) on a handle that runs on it:
// the thrust loss through the table image of the class hull (every env the same block) in its general per-env form, or no loss at all; the
// current re-draw is a run-time switch in those kernels whatever the vessel's source
static void bind_shared_loss_as_table(dpenv_handle h, StepArgs& a)
{
    if (a.loss_on != LOSS_SHARED) return;
    if (h->shared_loss) { a.env_tab = h->env_tab; a.loss_on = LOSS_TABLE; }
    else a.loss_on = LOSS_NONE;
}

// where the kernels take a lane's vessel from (dpenv_dev.h VES_*); call after bind_optional (which settles loss_state)
static int vessel_source(dpenv_handle h)
{
    if (h->per_env)
        return (h->randomise || h->loss_state != LOSS_OFF || (h->cur_rand && h->cfg.current_enabled)) ? VES_ENV_RND
               : (h->cfg.per_env_lds ? VES_ENV_LDS : VES_ENV_VGPR);
    if (h->n_classes > 1) return VES_CLASS_LDS;
    return (h->shared_loss || (h->cur_rand && h->cfg.current_enabled)) ? VES_ARGS_LOSS : VES_ARGS;
}

static bool classes_missing(dpenv_handle h) { return !h->per_env && h->n_classes > 1 && !h->classes_assigned; }

// ---- per-env parameter blocks ------------------------------------------------------------------------------------------------
extern "C" int dpenv_set_vessel_params_ex(dpenv_handle h, const float* params, uint32_t flags, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    // everything that can refuse the call comes first; the handle's switches (per_env, randomise, loss_state) change only after the last
    // call that can fail (ADVICE / VERDICT r05: a refused call used to leave the thrust loss and the re-draws silently off)
    if (flags & ~(uint32_t)DPENV_VESSEL_KEEP_RANDOMISATION) return fail(h, DPENV_EINVAL, "dpenv_set_vessel_params_ex: unknown flag bits 0x%x", flags);
    const bool keep = (flags & DPENV_VESSEL_KEEP_RANDOMISATION) != 0;
    if (keep && (!params || !h->randomise))
        return fail(h, DPENV_EINVAL, "DPENV_VESSEL_KEEP_RANDOMISATION needs a table and the randomisation in force (dpenv_set_vessel_randomisation first)");
    if (!params) {
        if (h->cur_rand && h->n_classes > 1)
            return fail(h, DPENV_EINVAL, "the per-episode current randomisation has no vessel-class form: switch it off first "
                                         "(dpenv_set_current_randomisation with both ranges 0)");
        // back to the classes / the single class; a single class WITH thrust-loss coefficients keeps them (kernel arguments), and the table
        // becomes the image of that hull again for the kernels that read it there
        if (h->per_env)
            HIP_TRY(h, dpenv_dev_launch_pack_env_vessels(h->raw0_dev, 1, 0, h->env_tab, nullptr, h->env_stride, h->cfg.n_envs, (hipStream_t)s));
        h->per_env = false;
        h->randomise = false;
        h->loss_pending = false;
        h->loss_state = LOSS_OFF;
        return DPENV_OK;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    HIP_TRY(h, hipStreamIsCapturing((hipStream_t)s, &cap));
    // stream-ordered, no read-back: the packing kernel leaves "some env has a thrust-loss coefficient" in the word behind the table
    HIP_TRY(h, hipMemsetAsync(h->loss_flag, 0, sizeof(uint32_t), (hipStream_t)s));
    HIP_TRY(h, dpenv_dev_launch_pack_env_vessels(params, (int64_t)h->cfg.n_envs, 1, h->env_tab, h->loss_flag, h->env_stride, h->cfg.n_envs, (hipStream_t)s));
    // which kernels run from here on depends on that word (the general per-env form applies the loss, the plain one does not carry the code).
    // Outside a capture it also travels to a pinned host word behind an event: the next launch takes the answer from there (resolve_loss);
    // recorded into a graph the setter leaves the answer unknown: the general kernels then apply the table's coefficients whatever they are.
    bool pending = false;
    if (cap == hipStreamCaptureStatusNone) {
        HIP_TRY(h, hipMemcpyAsync(h->loss_host, h->loss_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)s));
        HIP_TRY(h, hipEventRecord(h->loss_ev, (hipStream_t)s));
        pending = true;
    }
    h->per_env = true;
    if (!keep) h->randomise = false;
    if (h->randomise && h->rand_loss) { h->loss_state = LOSS_ON; h->loss_pending = false; }     // every drawn hull has a coefficient: nothing to ask
    else { h->loss_state = LOSS_UNKNOWN; h->loss_pending = pending; }
    return DPENV_OK;
}

extern "C" int dpenv_set_vessel_params(dpenv_handle h, const float* params, dpenv_stream s)
{
    return dpenv_set_vessel_params_ex(h, params, 0u, s);
}

extern "C" int dpenv_get_vessel_params(dpenv_handle h, float* params_out, dpenv_stream s)
{
    if (!h || !params_out) return fail(h, DPENV_EINVAL, "dpenv_get_vessel_params: NULL argument");
    if (!h->per_env && h->n_classes > 1)
        return fail(h, DPENV_EINVAL, "no per-env parameter blocks in force (dpenv_set_vessel_params / dpenv_set_vessel_randomisation)");
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dpenv_dev_launch_unpack_env_vessels(h->env_tab, h->env_stride, params_out, h->cfg.n_envs, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_set_vessel_randomisation(dpenv_handle h, const float* nominal, const float* rel_range, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    if (!rel_range) { h->randomise = false; return DPENV_OK; }         // hulls stay as they are, no more re-draws
    const float* nom = nominal ? nominal : h->raw0;
    for (int p = 0; p < DPENV_NPARAM; ++p) {
        const float r = p < RAND_NPARAM ? rel_range[p] : 0.0f;
        if (!std::isfinite(nom[p]) || !(r >= 0.0f && r < 1.0f) || (p >= DPENV_P_KLF_BOW && nom[p] < 0.0f))
            return fail(h, DPENV_EINVAL, "parameter %d: nominal must be finite (thrust-loss coefficients >= 0) and the relative half-range in [0, 1)", p);
        h->rand_host[p] = nom[p];
        h->rand_host[32 + p] = r;
    }
    {
        // every hull the draw can produce must be a vessel: the corner of the range with the lightest diagonal and the largest coupling
        const double m11 = (double)nom[DPENV_P_M11] * (1.0 - rel_range[DPENV_P_M11]);
        const double m22 = (double)nom[DPENV_P_M22] * (1.0 - rel_range[DPENV_P_M22]);
        const double m33 = (double)nom[DPENV_P_M33] * (1.0 - rel_range[DPENV_P_M33]);
        const double m23 = std::fabs((double)nom[DPENV_P_M23]) * (1.0 + rel_range[DPENV_P_M23]);
        if (!(m11 > 0.0) || !(m22 > 0.0) || !(m22 * m33 - m23 * m23 > 0.0))
            return fail(h, DPENV_EINVAL, "the range admits a mass matrix that is not positive definite");
    }
    HIP_TRY(h, hipMemcpyAsync(h->rand_tab, h->rand_host, sizeof h->rand_host, hipMemcpyHostToDevice, (hipStream_t)s));
    // until its first reset every env runs on the nominal hull
    HIP_TRY(h, dpenv_dev_launch_pack_env_vessels(h->rand_tab, 1, 0, h->env_tab, nullptr, h->env_stride, h->cfg.n_envs, (hipStream_t)s));
    h->per_env = true;
    h->randomise = true;
    h->rand_loss = false;                               // a draw scales the nominal value: no coefficient there, none in any hull
    for (int p = DPENV_P_KLF_BOW; p <= DPENV_P_KLR_STAR; ++p) h->rand_loss = h->rand_loss || nom[p] != 0.0f;
    h->loss_state = h->rand_loss ? LOSS_ON : LOSS_OFF;
    h->loss_pending = false;
    return DPENV_OK;
}

extern "C" int dpenv_reset(dpenv_handle h, const uint8_t* mask, const float* init, const float* ref, void* obs_out,
                           dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    StepArgs a = h->args;
    bind_optional(h, a, (hipStream_t)s);
    a.obs = obs_out;
    HIP_TRY(h, dpenv_dev_launch_reset(&a, h->mode, h->cfg.extended_state, mask, init, ref, (hipStream_t)s));
    // the reset kernel writes the lagged thrust columns (S3) of the envs it re-draws: a full reset makes them valid everywhere, a
    // masked one leaves the validity of the other envs' columns as it was (it used to drop the lag of every env)
    if (!mask) h->lag_valid = true;
    return DPENV_OK;
}

extern "C" int dpenv_step_ex(dpenv_handle h, const dpenv_step_io* io, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    if (!io || io->struct_size != sizeof(dpenv_step_io)) return fail(h, DPENV_EINVAL, "dpenv_step_io ABI mismatch");
    if (!io->action || !io->obs || !io->reward || !io->done)
        return fail(h, DPENV_EINVAL, "action, obs, reward and done buffers are required");
    if (classes_missing(h))
        return fail(h, DPENV_EINVAL, "n_classes > 1 but dpenv_set_vessel_class was never called");
    StepArgs a = h->args;
    bind_optional(h, a, (hipStream_t)s);
    a.action = io->action;
    a.new_ref = io->new_ref;
    a.obs = io->obs;
    a.rew = io->reward;
    a.done = io->done;
    a.parts = io->reward_parts;
    a.final_obs = io->final_obs;
    // the lagged thrust columns (S3) are kept by every kernel that changes the state; the one-launch-per-step kernel writes them only
    // while a policy is in force (16 B per env-step more) - without one the columns are marked stale and a later closed-loop launch
    // rebuilds its first observation from the state block
    if (!h->has_policy) a.S3 = nullptr;
    // the reset wave pays while the batch leaves it a SIMD slot beside its env wave (<= one env wave per SIMD = 256 envs per CU: 65 536 on
    // MI355X); beyond that the extra waves queue behind env waves and the one-wave kernel is faster (98 304 envs: 6.92 vs 7.19 us, 1 M:
    // 34.7 vs 39.2; profiles/r04_reset_wave.txt)
    const int reset_wave = (!h->cfg.step_one_wave && (int64_t)h->cfg.n_envs <= (int64_t)256 * h->n_cus) ? 1 : 0;
    HIP_TRY(h, dpenv_dev_launch_step(&a, h->mode, h->cfg.extended_state, vessel_source(h), reset_wave, (hipStream_t)s));
    h->lag_valid = h->has_policy;
    return DPENV_OK;
}

extern "C" int dpenv_step(dpenv_handle h, const float* action, const float* new_ref, void* obs_out, float* rew_out,
                          uint8_t* done_out, dpenv_stream s)
{
    dpenv_step_io io;
    std::memset(&io, 0, sizeof io);
    io.struct_size = (uint32_t)sizeof io;
    io.action = action; io.new_ref = new_ref; io.obs = obs_out; io.reward = rew_out; io.done = done_out;
    return dpenv_step_ex(h, &io, s);
}

extern "C" int dpenv_rollout(dpenv_handle h, const dpenv_rollout_io* io, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    if (!io || io->struct_size != sizeof(dpenv_rollout_io)) return fail(h, DPENV_EINVAL, "dpenv_rollout_io ABI mismatch");
    if (io->T <= 0 || !io->actions || !io->obs || !io->reward || !io->done)
        return fail(h, DPENV_EINVAL, "T > 0 and action, obs, reward, done blocks are required");
    if (io->n_switch < 0 || io->n_switch > DPENV_MAX_SWITCH || (io->n_switch > 0 && !io->refs))
        return fail(h, DPENV_EINVAL, "bad setpoint schedule");
    for (int k = 0; k < io->n_switch; ++k)
        if (io->switch_step[k] < 0 || io->switch_step[k] >= io->T || (k > 0 && io->switch_step[k] <= io->switch_step[k - 1]))
            return fail(h, DPENV_EINVAL, "switch_step must be strictly increasing within [0, T)");
    if (classes_missing(h))
        return fail(h, DPENV_EINVAL, "n_classes > 1 but dpenv_set_vessel_class was never called");
    StepArgs a = h->args;
    bind_optional(h, a, (hipStream_t)s);
    RolloutArgs ra;
    std::memset(&ra, 0, sizeof ra);
    ra.T = io->T; ra.actions = io->actions; ra.obs = io->obs; ra.rew = io->reward; ra.done = io->done;
    ra.n_switch = io->n_switch; ra.refs = io->refs;
    for (int k = 0; k < io->n_switch; ++k) ra.switch_step[k] = io->switch_step[k];
    // env wave + row wave while the chip has issue slots to spare (measured: 16 384 ... 98 304 envs 5-20 % faster, 131 072 equal, 262 144 and
    // above 10-15 % slower than one wave per 64 envs: profiles/r04_fused_two_wave.txt, r04_batch_sweep.txt)
    // (the randomisation's instantiation needs 201 VGPRs: two waves per SIMD, i.e. up to 256 envs per CU)
    const int two_wave = (!h->cfg.step_one_wave && (int64_t)h->cfg.n_envs <= (int64_t)(vessel_source(h) == VES_ENV_RND ? 256 : 384) * h->n_cus) ? 1 : 0;
    HIP_TRY(h, dpenv_dev_launch_rollout(&a, &ra, h->mode, h->cfg.extended_state, vessel_source(h), two_wave, (hipStream_t)s));
    // the rollout kernels leave the thrust columns of their last observation in S3 - with the extended state (without it no observation has
    // thrust columns and S3 is never written: the flag keeps what it was)
    if (h->cfg.extended_state) h->lag_valid = true;
    return DPENV_OK;
}

// ---- actor-critic ------------------------------------------------------------------------------------------
static int check_net(const dpenv_mlp* m, int in_dim, int out_dim, std::string* why)
{
    const int nl = m->n_layers;
    if (nl < 2 || nl > 5) { *why = "n_layers must be in [2, 5]"; return DPENV_EINVAL; }
    const int H = m->sizes[1];
    if (m->sizes[0] != in_dim || m->sizes[nl] != out_dim) { *why = "network input/output width does not match the env"; return DPENV_EINVAL; }
    if (in_dim > 15 || out_dim > 8 || H < 1 || H > 96) { *why = "limits: obs_dim <= 15, out <= 8, hidden width <= 96"; return DPENV_EINVAL; }
    for (int l = 1; l < nl; ++l) if (m->sizes[l] != H) { *why = "hidden widths must be equal"; return DPENV_EINVAL; }
    for (int l = 0; l < nl; ++l) if (!m->W[l] || !m->b[l]) { *why = "NULL weight pointer"; return DPENV_EINVAL; }
    return DPENV_OK;
}

extern "C" int dpenv_set_policy_desc(dpenv_handle h, const dpenv_policy_desc* d, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    if (!d || d->struct_size != sizeof(dpenv_policy_desc)) return fail(h, DPENV_EINVAL, "dpenv_policy_desc ABI mismatch");
    if (d->activation != DPENV_ACT_LEAKY_RELU && d->activation != DPENV_ACT_TANH)
        return fail(h, DPENV_EINVAL, "activation must be DPENV_ACT_LEAKY_RELU or DPENV_ACT_TANH");
    if (d->precision != DPENV_POLICY_F16 && d->precision != DPENV_POLICY_F32 && d->precision != DPENV_POLICY_F32_ACTOR)
        return fail(h, DPENV_EINVAL, "bad precision");
    if (d->launch_form < DPENV_LAUNCH_AUTO || d->launch_form > DPENV_LAUNCH_TWO_WAVE) return fail(h, DPENV_EINVAL, "bad launch_form");
    if (d->activation == DPENV_ACT_LEAKY_RELU && !(d->leak >= 0.0f && d->leak <= 1.0f))
        return fail(h, DPENV_EINVAL, "leaky-relu slope must be in [0, 1] (evaluated as max(x, leak x))");
    DeviceGuard dev_guard(h->device);
    const dpenv_mlp* pi = d->pi;
    const dpenv_mlp* v = d->v;
    if (!pi || !v || !d->log_std) return fail(h, DPENV_EINVAL, "dpenv_set_policy: NULL argument");
    if (pi->n_layers != v->n_layers || pi->sizes[1] != v->sizes[1])
        return fail(h, DPENV_EINVAL, "actor and critic must have the same hidden shape");
    const int od = dpenv_obs_dim(&h->cfg), ad = dpenv_act_dim(&h->cfg);
    std::string why;
    if (check_net(pi, od, ad, &why) != DPENV_OK) return fail(h, DPENV_EINVAL, "actor: %s", why.c_str());
    if (check_net(v, od, 1, &why) != DPENV_OK) return fail(h, DPENV_EINVAL, "critic: %s", why.c_str());
    const int nl = pi->n_layers, n_hidden = nl - 1, H = pi->sizes[1];
    const int ks = H <= 80 ? 5 : 6;
    const int nent = (128 + (ks == 5 ? 32 : 64)) + (n_hidden - 1) * ks * (128 + (ks == 5 ? 32 : 64)) + ks * 16;   // FragAddr, dpenv_policy_dev.h
    const int nblk = 3 * (n_hidden - 1) + 1;
    const int split = d->precision != DPENV_POLICY_F16;
    const size_t bytes_frags = (size_t)2 * nent * 16 * (1 + split), bytes_bias = (size_t)2 * nblk * 32 * sizeof(float);
    // the launch form decides the LDS footprint: refuse here what the rollout could not launch.  The two-wave form stages the
    // images its network wave reads (f16: actor + critic; F32_ACTOR: + the actor's low image; F32: all four) and a mailbox group
    // per 64 envs; it exists for leaky-relu / relu in every arithmetic and for tanh in f16.
    const size_t per_image = (size_t)nent * 16, lds_max = 160 * 1024;
    const int n_img_two = !split ? 2 : (d->precision == DPENV_POLICY_F32_ACTOR ? 3 : 4);
    const size_t lds_image = bytes_frags + bytes_bias;
    const size_t lds_two = n_img_two * per_image + bytes_bias + (split ? POLICY_WS_MAILBOX_X_BYTES : POLICY_WS_MAILBOX_BYTES);
    const bool form_two = !split || d->activation == DPENV_ACT_LEAKY_RELU;
    const bool fits_two = form_two && lds_two <= lds_max;
    const bool fits_one = lds_image + (split ? 0 : POLICY_STAGING_BYTES) <= lds_max;
    if (!fits_one) return fail(h, DPENV_EINVAL, "networks do not fit the 160 KiB LDS (%zu bytes of fragments and biases)", lds_image);
    if (d->launch_form == DPENV_LAUNCH_TWO_WAVE && !fits_two)
        return fail(h, DPENV_EINVAL, form_two ? "networks + the two-wave form's mailboxes exceed the 160 KiB LDS (%zu bytes): use DPENV_LAUNCH_ONE_WAVE"
                                              : "the split arithmetics have a two-wave form for leaky-relu / relu only (%zu)", lds_two);
    // two images, written alternately: launches issued before this call keep reading the one they were given (the header's
    // promise), whatever stream they run on; the image written now was last read by launches issued before the PREVIOUS upload,
    // and this stream waits for the last of them (an event recorded behind every launch that reads an image).
    const size_t need = (bytes_frags + bytes_bias + 24 * sizeof(float) + 255) / 256 * 256;
    if (h->pol_buf && h->pol_buf_bytes < 2 * need) {
        // growing the images: a launch that still reads the old ones may be in flight
        HIP_TRY(h, hipDeviceSynchronize());
        (void)hipFree(h->pol_buf); h->pol_buf = nullptr; h->pol_buf_bytes = 0;
        h->has_policy = false;
        std::memset(&h->pol, 0, sizeof h->pol);
        h->pol_read_valid[0] = h->pol_read_valid[1] = false;
        h->pol_pinned = -1;                                   // graphs captured on the old images are void (dpenv.h)
    }
    if (!h->pol_buf) {
        if (hipMalloc(&h->pol_buf, 2 * need) != hipSuccess) {
            h->pol_buf = nullptr; h->pol_buf_bytes = 0; h->has_policy = false;
            std::memset(&h->pol, 0, sizeof h->pol);
            return fail(h, DPENV_ENOMEM, "hipMalloc of the policy images failed");
        }
        h->pol_buf_bytes = 2 * need;
        h->pol_slot = 0;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing((hipStream_t)s, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;      // inside a captured graph the graph's own edges order everything
    // Which image this upload writes.  Eager launches read the image the last upload wrote (h->pol), so uploads alternate and a launch
    // in flight keeps its weights.  A launch RECORDED INTO A GRAPH has the image's address baked into its kernel node: once that has
    // happened (pol_pinned) an alternating upload would leave every second set of weights invisible to the replays - so from then on
    // every eager upload goes in place into the pinned image, ordered behind the eager readers by their event and behind graph
    // replays by stream order (the caller replays and uploads on one stream, or orders them itself: dpenv.h).
    //  An upload recorded INTO a graph after an image has been pinned goes into the pinned image as well: alternating there would
    // pin the other image at the next captured launch, and the graphs captured first would stop seeing uploads.
    const int slot = (h->pol_pinned >= 0) ? h->pol_pinned : h->pol_slot;
    const int ws_new = d->launch_form == DPENV_LAUNCH_TWO_WAVE ? 1 : (d->launch_form == DPENV_LAUNCH_ONE_WAVE ? 0 : (fits_two ? 1 : 0));
    const int groups_new = (d->activation == DPENV_ACT_LEAKY_RELU && (h->cfg.n_envs + 127) / 128 <= h->n_cus) ? 2 : 4;
    if (h->pol_pinned >= 0) {
        // the captured kernel nodes hold nent, nblk, ks, n_hidden, act, leak, split, critic_f16, ws, ws_groups BY VALUE next to the image's
        // address: an in-place upload of another shape, arithmetic, activation or launch form would be read with the old layout - wrong
        // weights and no error.  Refused; dpenv_release_policy_graphs() ends the pin once the graphs are gone.
        const PolicyArgs& q = h->pol_pin_layout;
        if (q.nent != nent || q.nblk != nblk || q.ks != ks || q.n_hidden != n_hidden || q.act != d->activation || q.leak != d->leak ||
            q.split != split || q.critic_f16 != (d->precision == DPENV_POLICY_F32_ACTOR ? 1 : 0) || q.ws != ws_new || q.ws_groups != groups_new)
            return fail(h, DPENV_EINVAL, "a captured graph reads the policy image with another layout (hidden shape, precision, activation, leak or launch "
                                         "form differ): destroy the graphs and call dpenv_release_policy_graphs() before uploading this policy");
    }
    if (!capturing && h->pol_read_valid[slot]) HIP_TRY(h, hipStreamWaitEvent((hipStream_t)s, h->pol_read[slot], 0));
    PackNet pn[2];
    const float* ls_dev = d->log_std;
    if (!d->device_pointers) {
        // host weights: stage them raw in device memory (stream-ordered copies), then pack on the device like the other form
        size_t raw = 8;
        for (int k = 0; k < 2; ++k) {
            const dpenv_mlp* m = k ? v : pi;
            for (int l = 0; l < nl; ++l) raw += (size_t)m->sizes[l] * m->sizes[l + 1] + m->sizes[l + 1];
        }
        // the staging area is reused by every host-pointer upload: the previous one's packing kernel must be done with it (this
        // form is synchronous anyway, see the end of this function)
        if (h->pol_raw && h->pol_raw_bytes < raw * sizeof(float)) { HIP_TRY(h, hipDeviceSynchronize()); (void)hipFree(h->pol_raw); h->pol_raw = nullptr; }
        if (!h->pol_raw) {
            void* p = nullptr;
            if (hipMalloc(&p, raw * sizeof(float)) != hipSuccess) return fail(h, DPENV_ENOMEM, "hipMalloc of the weight staging failed");
            h->pol_raw = (float*)p; h->pol_raw_bytes = raw * sizeof(float);
        }
        float* cursor = h->pol_raw;
        HIP_TRY(h, hipMemcpyAsync(cursor, d->log_std, sizeof(float) * ad, hipMemcpyHostToDevice, (hipStream_t)s));
        ls_dev = cursor; cursor += 8;
        for (int k = 0; k < 2; ++k) {
            const dpenv_mlp* m = k ? v : pi;
            for (int l = 0; l < nl; ++l) {
                const size_t nw = (size_t)m->sizes[l] * m->sizes[l + 1], nb = m->sizes[l + 1];
                HIP_TRY(h, hipMemcpyAsync(cursor, m->W[l], nw * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)s));
                pn[k].W[l] = cursor; cursor += nw;
                HIP_TRY(h, hipMemcpyAsync(cursor, m->b[l], nb * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)s));
                pn[k].b[l] = cursor; cursor += nb;
            }
        }
    } else {
        for (int k = 0; k < 2; ++k) {
            const dpenv_mlp* m = k ? v : pi;
            for (int l = 0; l < nl; ++l) { pn[k].W[l] = m->W[l]; pn[k].b[l] = m->b[l]; }
        }
    }
    for (int k = 0; k < 2; ++k) {
        pn[k].n_layers = nl; pn[k].in_dim = od; pn[k].H = H; pn[k].out_dim = k ? 1 : ad;
        for (int l = nl; l < 5; ++l) { pn[k].W[l] = nullptr; pn[k].b[l] = nullptr; }
    }
    char* base = (char*)h->pol_buf + (size_t)slot * (h->pol_buf_bytes / 2);
    float* bias = (float*)(base + bytes_frags);
    float* consts = (float*)(base + bytes_frags + bytes_bias);
    HIP_TRY(h, dpenv_dev_launch_pack_policy(&pn[0], &pn[1], ls_dev, ad, ks, nent, nblk, split, base, bias, consts, (hipStream_t)s));
    PolicyArgs& pa = h->pol;
    std::memset(&pa, 0, sizeof pa);
    pa.frags = (const uint4*)base;
    pa.bias = bias;
    pa.consts = consts;
    pa.nent = nent;
    pa.nblk = nblk;
    pa.ks = ks;
    pa.act = d->activation;
    pa.split = split;
    pa.critic_f16 = d->precision == DPENV_POLICY_F32_ACTOR;
    pa.n_hidden = n_hidden;
    pa.leak = d->leak;
    // AUTO: the two-wave form where it exists and fits (the faster one), else one wave per 64 envs
    pa.ws = ws_new;
    // two-wave geometry: 128-env workgroups (a SIMD per wave) while one round of them fits the chip, else 256-env workgroups;
    // tanh has the 256-env form only
    pa.ws_groups = groups_new;
    h->pol_form = d->launch_form;
    h->has_policy = true;
    h->pol_cur = slot;                                       // "current" = the image just written (pinned or not)
    h->pol_slot = slot ^ 1;
    // host pointers: the caller's arrays (and the shared staging area) must be done with before this returns
    if (!d->device_pointers && !capturing) HIP_TRY(h, hipStreamSynchronize((hipStream_t)s));
    return DPENV_OK;
}

// behind every launch that reads the current image: the event an upload into that image waits for
static int mark_policy_read(dpenv_handle h, dpenv_stream s)
{
    const int cur = h->pol_cur;                             // the image the last upload wrote
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing((hipStream_t)s, &cap);
    if (cap != hipStreamCaptureStatusNone) {
        h->pol_pinned = cur;                                // a graph now holds this image's address: uploads go in place from here on
        h->pol_pin_layout = h->pol;                         // ... and its layout / launch form, by value
        return DPENV_OK;
    }
    if (!h->pol_read[cur] && hipEventCreateWithFlags(&h->pol_read[cur], hipEventDisableTiming) != hipSuccess) {
        h->pol_read[cur] = nullptr;
        return fail(h, DPENV_EHIP, "hipEventCreate failed");
    }
    // one event per image: a reader on ANOTHER stream than the last one first waits (on its stream) for the event as it stands, so
    // that the re-recorded event completes only when both readers are done - the chain covers every reader on any stream
    if (h->pol_read_valid[cur] && h->pol_read_stream[cur] != (hipStream_t)s)
        HIP_TRY(h, hipStreamWaitEvent((hipStream_t)s, h->pol_read[cur], 0));
    HIP_TRY(h, hipEventRecord(h->pol_read[cur], (hipStream_t)s));
    h->pol_read_valid[cur] = true;
    h->pol_read_stream[cur] = (hipStream_t)s;
    return DPENV_OK;
}

extern "C" int dpenv_set_policy_ex(dpenv_handle h, const dpenv_mlp* pi, const dpenv_mlp* v, const float* log_std, int32_t activation,
                                   float leak)
{
    dpenv_policy_desc d;
    std::memset(&d, 0, sizeof d);
    d.struct_size = (uint32_t)sizeof d;
    d.pi = pi; d.v = v; d.log_std = log_std; d.activation = activation; d.leak = leak;
    d.precision = DPENV_POLICY_F16; d.launch_form = DPENV_LAUNCH_AUTO; d.device_pointers = 0;
    const int rc = dpenv_set_policy_desc(h, &d, nullptr);     // host pointers: synchronises the null stream before it returns
    if (rc != DPENV_OK) return rc;
    // the convenience forms keep the contract they always had: when they return the new weights are in force for launches on
    // ANY stream (non-blocking streams do not order themselves behind the null stream)
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, hipDeviceSynchronize());
    return DPENV_OK;
}

extern "C" int dpenv_set_policy(dpenv_handle h, const dpenv_mlp* pi, const dpenv_mlp* v, const float* log_std, float leak)
{
    return dpenv_set_policy_ex(h, pi, v, log_std, DPENV_ACT_LEAKY_RELU, leak);
}

extern "C" int dpenv_get_policy_launch(dpenv_handle h, int32_t* two_wave_out, int32_t* envs_per_workgroup_out)
{
    if (!h) return DPENV_EINVAL;
    if (!h->has_policy) return fail(h, DPENV_EINVAL, "dpenv_set_policy has not been called");
    if (two_wave_out) *two_wave_out = h->pol.ws ? 1 : 0;
    if (envs_per_workgroup_out) *envs_per_workgroup_out = h->pol.ws ? 64 * h->pol.ws_groups : 256;
    return DPENV_OK;
}

extern "C" int dpenv_get_policy_launch_ex(dpenv_handle h, int32_t out[4])
{
    if (!h || !out) return fail(h, DPENV_EINVAL, "dpenv_get_policy_launch_ex: NULL argument");
    if (!h->has_policy) return fail(h, DPENV_EINVAL, "dpenv_set_policy has not been called");
    const PolicyArgs& pa = h->pol;
    const int prec = !pa.split ? PREC_F16 : (pa.critic_f16 ? PREC_F32_ACTOR : PREC_F32);
    out[0] = pa.ws ? 1 : 0;
    out[1] = pa.ws ? 64 * pa.ws_groups : 256;
    out[2] = !pa.ws ? 1 : ((pa.ws_groups == 2 && ((DPENV_WS_CRITIC_WAVE >> prec) & 1)) ? 3 : 2);      // waves per 64 envs (dpenv_policy_ws.h: go<>)
    out[3] = prec;
    return DPENV_OK;
}

extern "C" int dpenv_release_policy_graphs(dpenv_handle h)
{
    if (!h) return DPENV_EINVAL;
    h->pol_pinned = -1;                                      // uploads alternate between the two images again
    return DPENV_OK;
}

extern "C" int dpenv_policy_forward(dpenv_handle h, const float* obs, float* mu_out, float* v_out, int32_t n, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    if (!h->has_policy) return fail(h, DPENV_EINVAL, "dpenv_set_policy has not been called");
    if (!obs || !mu_out || !v_out || n <= 0) return fail(h, DPENV_EINVAL, "dpenv_policy_forward: bad argument");
    if (h->pol.split)
        HIP_TRY(h, dpenv_dev_launch_policy_forward_x(&h->pol, dpenv_obs_dim(&h->cfg), dpenv_act_dim(&h->cfg), obs, mu_out, v_out, n,
                                                     (hipStream_t)s));
    else
        HIP_TRY(h, dpenv_dev_launch_policy_forward(&h->pol, dpenv_obs_dim(&h->cfg), dpenv_act_dim(&h->cfg), obs, mu_out, v_out, n,
                                                   (hipStream_t)s));
    return mark_policy_read(h, s);
}

#ifdef DPENV_WS_SELFCHECK
// diagnostic builds only (tools/ws_selfcheck.py): device buffer of mismatch records, see policy_rollout_ws_kernel
static uint32_t* g_selfcheck_buf = nullptr;
extern "C" int dpenv_debug_set_buffer(void* p) { g_selfcheck_buf = (uint32_t*)p; return 0; }
extern "C" hipError_t dpenv_dev_launch_pk_probe(const PolicyArgs* pa, uint32_t* out, float* sink, int blocks, int iters, int variant, int partner,
                                                hipStream_t s);
extern "C" int dpenv_debug_pk_probe(dpenv_handle h, void* out, void* sink, int blocks, int iters, int variant, int partner)
{
    if (!h || !h->has_policy) return DPENV_EINVAL;
    HIP_TRY(h, dpenv_dev_launch_pk_probe(&h->pol, (uint32_t*)out, (float*)sink, blocks, iters, variant, partner, nullptr));
    return DPENV_OK;
}
#endif

extern "C" int dpenv_policy_rollout(dpenv_handle h, const dpenv_policy_rollout_io* io, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    if (!h->has_policy) return fail(h, DPENV_EINVAL, "dpenv_set_policy has not been called");
    if (!io || io->struct_size != sizeof(dpenv_policy_rollout_io)) return fail(h, DPENV_EINVAL, "dpenv_policy_rollout_io ABI mismatch");
    if (io->T <= 0 || !io->obs || !io->act || !io->reward || !io->value || !io->logp || !io->done || !io->boot ||
        !io->last_obs || !io->last_value)
        return fail(h, DPENV_EINVAL, "T > 0 and every output block are required");
    if (h->cfg.action_layout != DPENV_AOS || h->cfg.obs_layout != DPENV_AOS)
        return fail(h, DPENV_EINVAL, "policy rollout needs AOS layouts");
    if (classes_missing(h))
        return fail(h, DPENV_EINVAL, "n_classes > 1 but dpenv_set_vessel_class was never called");
    if (io->n_switch < 0 || io->n_switch > DPENV_MAX_SWITCH || (io->n_switch > 0 && !io->refs))
        return fail(h, DPENV_EINVAL, "bad setpoint schedule");
    for (int k = 0; k < io->n_switch; ++k)
        if (io->switch_step[k] < 0 || io->switch_step[k] >= io->T || (k > 0 && io->switch_step[k] <= io->switch_step[k - 1]))
            return fail(h, DPENV_EINVAL, "switch_step must be strictly increasing within [0, T)");
    StepArgs a = h->args;
    bind_optional(h, a, (hipStream_t)s);
    PolicyArgs pa = h->pol;
    pa.T = io->T; pa.noise = io->noise; pa.obs_out = io->obs; pa.act_out = io->act; pa.rew = io->reward; pa.val = io->value;
    pa.logp = io->logp; pa.done = io->done; pa.boot = io->boot; pa.last_obs = io->last_obs; pa.last_val = io->last_value;
    pa.n_switch = io->n_switch; pa.refs = io->refs;
    pa.sample = io->sample ? 1 : 0;
    pa.reset_at_end = io->reset_at_end ? 1 : 0;
    pa.use_lag = h->lag_valid ? 1 : 0;
    if (pa.reset_at_end && !h->cfg.auto_reset)
        return fail(h, DPENV_EINVAL, "reset_at_end re-draws every env with the training sampler: it needs config.auto_reset");
    for (int k = 0; k < io->n_switch; ++k) pa.switch_step[k] = io->switch_step[k];
#ifdef DPENV_WS_SELFCHECK
    pa.dbg = g_selfcheck_buf;
#endif
    // the two-wave kernels carry the randomisation's hull re-draw in an instantiation of their own, built for the shipped training configuration
    // (final / continuous angles / extended state, leaky-relu or relu networks); everything else runs the one-wave kernels while the
    // randomisation is on - the same rows bit for bit, the draw a run-time switch there
    const bool shipped = h->mode == MODE_FINAL_CONT && h->cfg.extended_state && pa.act == DPENV_ACT_LEAKY_RELU;
    if ((a.rand_tab || a.loss_on != LOSS_NONE || a.cur_nom) && pa.ws && !shipped) pa.ws = 0;
    // the one-wave kernels (and only they, after the line above) know the thrust loss through the table alone
    if (!pa.ws) bind_shared_loss_as_table(h, a);
    if (pa.split) HIP_TRY(h, dpenv_dev_launch_policy_rollout_x(&a, &pa, h->mode, h->cfg.extended_state, (hipStream_t)s));
    else HIP_TRY(h, dpenv_dev_launch_policy_rollout(&a, &pa, h->mode, h->cfg.extended_state, (hipStream_t)s));
    h->lag_valid = true;
    return mark_policy_read(h, s);
}

extern "C" int dpenv_get_state(dpenv_handle h, float* state_out, int32_t* counters_out, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dpenv_dev_launch_get_state(&h->args, state_out, counters_out, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_set_state(dpenv_handle h, const float* state_in, const int32_t* counters_in, dpenv_stream s)
{
    if (!h) return DPENV_EINVAL;
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dpenv_dev_launch_set_state(&h->args, state_in, counters_in, (hipStream_t)s));
    if (state_in) h->lag_valid = true;                     // the kernel wrote the columns a state-rebuilt observation carries (pt / 100)
    return DPENV_OK;
}

// exploration-noise and current-drift draw counters: with them a restored state reproduces sampled rollouts (dpenv.h)
extern "C" int dpenv_get_rng_counters(dpenv_handle h, uint32_t* noise_ctr_out, uint32_t* drift_ctr_out, dpenv_stream s)
{
    if (!h || (!noise_ctr_out && !drift_ctr_out)) return fail(h, DPENV_EINVAL, "dpenv_get_rng_counters: NULL argument");
    DeviceGuard dev_guard(h->device);
    const size_t bytes = sizeof(uint32_t) * (size_t)h->cfg.n_envs;
    if (noise_ctr_out) HIP_TRY(h, hipMemcpyAsync(noise_ctr_out, h->noise_ctr, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    if (drift_ctr_out) HIP_TRY(h, hipMemcpyAsync(drift_ctr_out, h->drift_ctr, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_set_rng_counters(dpenv_handle h, const uint32_t* noise_ctr_in, const uint32_t* drift_ctr_in, dpenv_stream s)
{
    if (!h || (!noise_ctr_in && !drift_ctr_in)) return fail(h, DPENV_EINVAL, "dpenv_set_rng_counters: NULL argument");
    DeviceGuard dev_guard(h->device);
    const size_t bytes = sizeof(uint32_t) * (size_t)h->cfg.n_envs;
    if (noise_ctr_in) HIP_TRY(h, hipMemcpyAsync(h->noise_ctr, noise_ctr_in, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    if (drift_ctr_in) HIP_TRY(h, hipMemcpyAsync(h->drift_ctr, drift_ctr_in, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return DPENV_OK;
}

// the thrust columns of the observation the last closed-loop launch ended with (PolicyArgs.use_lag): part of a mid-episode checkpoint
extern "C" int dpenv_get_obs_thrust(dpenv_handle h, float* out, dpenv_stream s)
{
    if (!h || !out) return fail(h, DPENV_EINVAL, "dpenv_get_obs_thrust: NULL argument");
    if (!h->lag_valid) return fail(h, DPENV_EINVAL, "the lagged thrust columns are stale (dpenv_step ran without a policy in force, or a masked reset on stale columns): the next closed-loop launch starts from the state block alone");
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, hipMemcpyAsync(out, h->args.S3, sizeof(float4) * (size_t)h->cfg.n_envs, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_set_obs_thrust(dpenv_handle h, const float* in, dpenv_stream s)
{
    if (!h || !in) return fail(h, DPENV_EINVAL, "dpenv_set_obs_thrust: NULL argument");
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, hipMemcpyAsync(h->args.S3, in, sizeof(float4) * (size_t)h->cfg.n_envs, hipMemcpyDeviceToDevice, (hipStream_t)s));
    h->lag_valid = true;
    return DPENV_OK;
}

extern "C" int dpenv_thrust_map(const float* params, const float* n_pct, const float* alpha, float* tau_out, int32_t n,
                                dpenv_stream s)
{
    if (!n_pct || !alpha || !tau_out || n <= 0) return fail(nullptr, DPENV_EINVAL, "dpenv_thrust_map: bad argument");
    float defp[DPENV_NPARAM];
    if (!params) { dpenv_default_vessel(defp); params = defp; }
    VesselDev vd;
    std::string why;
    if (derive_vessel(params, &vd, &why, true) != DPENV_OK) return fail(nullptr, DPENV_EINVAL, "%s", why.c_str());
    HIP_TRY(nullptr, dpenv_dev_launch_thrust_map(&vd, n_pct, alpha, tau_out, n, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int64_t dpenv_gae_workspace_bytes(int32_t n) { return n > 0 ? dpenv_dev_gae_workspace_bytes(n) : 0; }

extern "C" int dpenv_gae_stats(const float* rew, const float* val, const uint8_t* end, const float* boot, const float* last_val,
                               int32_t T, int32_t n, float gamma, float lam, float* adv_out, float* ret_out, void* workspace,
                               double* stats_out, dpenv_stream s)
{
    if (!rew || !val || !adv_out || !ret_out || T <= 0 || n <= 0)
        return fail(nullptr, DPENV_EINVAL, "dpenv_gae: bad argument");
    if (stats_out && !workspace) return fail(nullptr, DPENV_EINVAL, "dpenv_gae_stats: statistics need the workspace (dpenv_gae_workspace_bytes)");
    HIP_TRY(nullptr, dpenv_dev_launch_gae(rew, val, end, boot, last_val, T, n, gamma, lam, adv_out, ret_out, (double*)workspace,
                                          stats_out, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_gae(const float* rew, const float* val, const uint8_t* end, const float* boot, const float* last_val,
                         int32_t T, int32_t n, float gamma, float lam, float* adv_out, float* ret_out, dpenv_stream s)
{
    return dpenv_gae_stats(rew, val, end, boot, last_val, T, n, gamma, lam, adv_out, ret_out, nullptr, nullptr, s);
}

extern "C" int dpenv_adv_sum(const float* adv, int64_t count, float* sum_out, dpenv_stream s)
{
    if (!adv || !sum_out || count <= 0) return fail(nullptr, DPENV_EINVAL, "dpenv_adv_sum: bad argument");
    HIP_TRY(nullptr, dpenv_dev_launch_sum(adv, count, nullptr, sum_out, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_adv_sumsq(const float* adv, int64_t count, const float* mean, float* sumsq_out, dpenv_stream s)
{
    if (!adv || !mean || !sumsq_out || count <= 0) return fail(nullptr, DPENV_EINVAL, "dpenv_adv_sumsq: bad argument");
    HIP_TRY(nullptr, dpenv_dev_launch_sum(adv, count, mean, sumsq_out, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_adv_apply(float* adv, int64_t count, const float* mean, const float* std, dpenv_stream s)
{
    if (!adv || !mean || !std || count <= 0) return fail(nullptr, DPENV_EINVAL, "dpenv_adv_apply: bad argument");
    HIP_TRY(nullptr, dpenv_dev_launch_adv_apply(adv, count, mean, std, nullptr, 0.0, (hipStream_t)s));
    return DPENV_OK;
}

extern "C" int dpenv_adv_apply_stats(float* adv, int64_t count, const double* stats, double total_count, dpenv_stream s)
{
    if (!adv || !stats || count <= 0 || !(total_count >= 1.0)) return fail(nullptr, DPENV_EINVAL, "dpenv_adv_apply_stats: bad argument");
    HIP_TRY(nullptr, dpenv_dev_launch_adv_apply(adv, count, nullptr, nullptr, stats, total_count, (hipStream_t)s));
    return DPENV_OK;
}
