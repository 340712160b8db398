/* c_abi_demo.c - libdpenv.so from plain C: no Python, no torch.  Allocates the I/O buffers with the HIP runtime,
 * resets 65 536 environments, steps them 2 000 times with a fixed action block, and prints throughput and a
 * checksum.  Then the FUSED entry points from the same C program (round 5): dpenv_rollout (50 env steps in one launch, one setpoint
 * switch inside it) and, on the rows it wrote, dpenv_gae_stats + dpenv_adv_apply_stats (TrajectoryBuffer.finish_path / get of
 * spinup/algos/tf1/ppo/ppo.py:65-105), and the per-env vessel entry points (the thrust-loss preset through dpenv_create, explicit per-env
 * blocks with dpenv_set_vessel_params / dpenv_get_vessel_params, dpenv_set_vessel_randomisation) - with checksums that
 * tests/test_gpu_c_abi.py compares with the same calls made through the Python binding.  Build (see tests/test_gpu_c_abi.py):
 *   gcc -std=c11 -O2 -I include -I /opt/rocm/include examples/c_abi_demo.c -L ml4ca_amd/lib -ldpenv -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/ml4ca_amd/lib -Wl,-rpath,/opt/rocm/lib -o build/c_abi_demo
 */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "dpenv.h"

#define CHECK(x)                                                                     \
    do {                                                                             \
        int rc_ = (x);                                                               \
        if (rc_ != DPENV_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, dpenv_last_error(h));        \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 65536;
    const int steps = argc > 2 ? atoi(argv[2]) : 2000;
    dpenv_handle h = NULL;
    dpenv_config cfg;
    dpenv_default_config(&cfg);               /* RevoltFinal, extended state, continuous-angle heads, T = 400 */
    cfg.n_envs = n;
    cfg.auto_reset = 1;
    cfg.seed = 7;
    CHECK(dpenv_create(&cfg, NULL, 1, &h));
    const int ad = dpenv_act_dim(&cfg), od = dpenv_obs_dim(&cfg);

    float *act, *obs, *rew;
    uint8_t* done;
    if (hipMalloc((void**)&act, sizeof(float) * n * ad) != hipSuccess || hipMalloc((void**)&obs, sizeof(float) * n * od) != hipSuccess ||
        hipMalloc((void**)&rew, sizeof(float) * n) != hipSuccess || hipMalloc((void**)&done, n) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    float* hact = (float*)malloc(sizeof(float) * n * ad);
    unsigned s = 12345u;
    for (int i = 0; i < n * ad; ++i) {
        s = s * 1664525u + 1013904223u;
        hact[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 1.2f;
    }
    hipMemcpy(act, hact, sizeof(float) * n * ad, hipMemcpyHostToDevice);

    hipStream_t stream;
    hipStreamCreate(&stream);
    CHECK(dpenv_reset(h, NULL, NULL, NULL, obs, stream));
    for (int t = 0; t < 50; ++t) CHECK(dpenv_step(h, act, NULL, obs, rew, done, stream));
    hipStreamSynchronize(stream);
    const double t0 = now();
    for (int t = 0; t < steps; ++t) CHECK(dpenv_step(h, act, NULL, obs, rew, done, stream));
    hipStreamSynchronize(stream);
    const double dt = now() - t0;

    float* hrew = (float*)malloc(sizeof(float) * n);
    uint8_t* hdone = (uint8_t*)malloc(n);
    hipMemcpy(hrew, rew, sizeof(float) * n, hipMemcpyDeviceToHost);
    hipMemcpy(hdone, done, n, hipMemcpyDeviceToHost);
    double sum = 0.0;
    int faults = 0;
    for (int i = 0; i < n; ++i) {
        sum += hrew[i];
        faults += (hdone[i] & DPENV_DONE_FAULT) != 0;
    }
    printf("c_abi_demo: %d envs x %d eager steps in %.3f ms = %.3e env-steps/s (%.2f us per step); mean reward %.4f; faults %d\n", n,
           steps, dt * 1e3, (double)n * steps / dt, dt / steps * 1e6, sum / n, faults);
    CHECK(dpenv_destroy(h));
    if (faults) return 2;

    /* ---- fused entry points: dpenv_rollout + dpenv_gae_stats + dpenv_adv_apply_stats ------------------------------------------- */
    const int n2 = argc > 3 ? atoi(argv[3]) : 4096, T = 50, SW = 20;
    dpenv_default_config(&cfg);
    cfg.n_envs = n2;
    cfg.auto_reset = 1;
    cfg.seed = 7;
    h = NULL;
    CHECK(dpenv_create(&cfg, NULL, 1, &h));
    const size_t na = (size_t)T * n2 * ad, no = (size_t)T * n2 * od, nr = (size_t)T * n2;
    float *acts, *obsT, *rewT, *valT, *advT, *retT, *refs;
    uint8_t* doneT;
    double* stats;
    void* ws;
    const size_t ws_bytes = (size_t)dpenv_gae_workspace_bytes(n2);
    if (hipMalloc((void**)&acts, 4 * na) != hipSuccess || hipMalloc((void**)&obsT, 4 * no) != hipSuccess || hipMalloc((void**)&rewT, 4 * nr) != hipSuccess ||
        hipMalloc((void**)&valT, 4 * nr) != hipSuccess || hipMalloc((void**)&advT, 4 * nr) != hipSuccess || hipMalloc((void**)&retT, 4 * nr) != hipSuccess ||
        hipMalloc((void**)&doneT, nr) != hipSuccess || hipMalloc((void**)&refs, 4 * 3 * (size_t)n2) != hipSuccess ||
        hipMalloc((void**)&stats, 2 * sizeof(double)) != hipSuccess || hipMalloc(&ws, ws_bytes) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    float* hacts = (float*)malloc(4 * na);
    s = 2024u;
    for (size_t i = 0; i < na; ++i) {
        s = s * 1664525u + 1013904223u;
        hacts[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 1.6f;
    }
    hipMemcpy(acts, hacts, 4 * na, hipMemcpyHostToDevice);
    float* hrefs = (float*)malloc(4 * 3 * (size_t)n2);
    for (int i = 0; i < n2; ++i) { hrefs[i] = 1.5f; hrefs[n2 + i] = -0.5f; hrefs[2 * n2 + i] = 0.1f; }      /* [1][3][n]: N, E, psi */
    hipMemcpy(refs, hrefs, 4 * 3 * (size_t)n2, hipMemcpyHostToDevice);
    CHECK(dpenv_reset(h, NULL, NULL, NULL, NULL, stream));
    dpenv_rollout_io rio;
    memset(&rio, 0, sizeof rio);
    rio.struct_size = (uint32_t)sizeof rio;
    rio.T = T; rio.actions = acts; rio.obs = obsT; rio.reward = rewT; rio.done = doneT;
    rio.n_switch = 1; rio.switch_step[0] = SW; rio.refs = refs;            /* the setpoint handed over at step 20, visible from step 21 (customEnv.py:131) */
    CHECK(dpenv_rollout(h, &rio, stream));
    hipStreamSynchronize(stream);
    float* hobs = (float*)malloc(4 * no);
    float* hrewT = (float*)malloc(4 * nr);
    uint8_t* hdoneT = (uint8_t*)malloc(nr);
    hipMemcpy(hobs, obsT, 4 * no, hipMemcpyDeviceToHost);
    hipMemcpy(hrewT, rewT, 4 * nr, hipMemcpyDeviceToHost);
    hipMemcpy(hdoneT, doneT, nr, hipMemcpyDeviceToHost);
    double obs_sum = 0.0, rew_sum = 0.0;
    long done_count = 0;
    for (size_t i = 0; i < no; ++i) obs_sum += hobs[i];
    for (size_t i = 0; i < nr; ++i) { rew_sum += hrewT[i]; done_count += hdoneT[i] != 0; }
    /* a stand-in critic: V = r / 2 (the scan only needs SOME value row; the closed-loop entry point writes the real one) */
    for (size_t i = 0; i < nr; ++i) hrewT[i] *= 0.5f;
    hipMemcpy(valT, hrewT, 4 * nr, hipMemcpyHostToDevice);
    if (dpenv_gae_stats(rewT, valT, doneT, NULL, NULL, T, n2, 0.99f, 0.97f, advT, retT, ws, stats, stream) != DPENV_OK ||
        dpenv_adv_apply_stats(advT, (int64_t)nr, stats, (double)nr, stream) != DPENV_OK) {
        fprintf(stderr, "gae / normalisation failed: %s\n", dpenv_last_error(NULL));
        return 1;
    }
    hipStreamSynchronize(stream);
    double hstats[2], adv_sum = 0.0, adv_sq = 0.0, ret_sum = 0.0;
    hipMemcpy(hstats, stats, sizeof hstats, hipMemcpyDeviceToHost);
    hipMemcpy(hrewT, advT, 4 * nr, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < nr; ++i) { adv_sum += hrewT[i]; adv_sq += (double)hrewT[i] * hrewT[i]; }
    hipMemcpy(hrewT, retT, 4 * nr, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < nr; ++i) ret_sum += hrewT[i];
    printf("c_abi_demo fused: %d envs x %d steps in one dpenv_rollout launch, setpoint switch at step %d; checksums obs %.17g rew %.17g done %ld "
           "gae_stats %.17g %.17g adv_norm_sum %.17g adv_norm_sq %.17g ret %.17g\n", n2, T, SW, obs_sum, rew_sum, done_count, hstats[0], hstats[1], adv_sum,
           adv_sq, ret_sum);
    CHECK(dpenv_destroy(h));

    /* ---- per-env vessels from plain C: a preset through dpenv_create, explicit per-env blocks, the randomisation ------------------ */
    const int n3 = 2048, S3 = 20;
    float preset[DPENV_NPARAM], range[DPENV_NPARAM];
    CHECK(dpenv_default_vessel_ex(DPENV_VESSEL_THRUST_LOSS, preset));          /* the hull with the inflow thrust loss (customEnv.py:13-18, second set of speeds) */
    dpenv_default_config(&cfg);
    cfg.n_envs = n3;
    cfg.auto_reset = 1;
    cfg.max_ep_len = 8;
    cfg.seed = 11;
    h = NULL;
    CHECK(dpenv_create(&cfg, preset, 1, &h));                                  /* one class WITH loss coefficients: hull and coefficients as kernel arguments */
    float *tab, *act3, *obs3, *rew3;
    uint8_t* done3;
    const size_t nt = (size_t)DPENV_NPARAM * n3;
    if (hipMalloc((void**)&tab, 4 * nt) != hipSuccess || hipMalloc((void**)&act3, 4 * (size_t)n3 * ad) != hipSuccess ||
        hipMalloc((void**)&obs3, 4 * (size_t)n3 * od) != hipSuccess || hipMalloc((void**)&rew3, 4 * (size_t)n3) != hipSuccess ||
        hipMalloc((void**)&done3, n3) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    float* htab = (float*)malloc(4 * nt);
    CHECK(dpenv_get_vessel_params(h, tab, stream));
    hipStreamSynchronize(stream);
    hipMemcpy(htab, tab, 4 * nt, hipMemcpyDeviceToHost);
    int same = 1;
    for (int p = 0; p < DPENV_NPARAM; ++p)
        for (int i = 0; i < n3; ++i) same = same && htab[(size_t)p * n3 + i] == preset[p];
    /* every env its own hull AND its own loss coefficients: +-10 % around the preset, [param][env] */
    s = 4711u;
    for (int p = 0; p < DPENV_NPARAM; ++p)
        for (int i = 0; i < n3; ++i) {
            s = s * 1664525u + 1013904223u;
            htab[(size_t)p * n3 + i] = preset[p] * (1.0f + 0.2f * ((float)(s >> 8) / 16777216.0f - 0.5f));
        }
    hipMemcpy(tab, htab, 4 * nt, hipMemcpyHostToDevice);
    /* the error path: a refused call (unknown flag bits; KEEP_RANDOMISATION with no randomisation in force) returns DPENV_EINVAL with a message and
     * leaves the handle as it was - the preset's thrust loss stays in force (tests/test_gpu_round6.py checks the rows) */
    int refused = dpenv_set_vessel_params_ex(h, tab, 0x80u, stream) == DPENV_EINVAL && strstr(dpenv_last_error(h), "unknown flag") != NULL &&
                  dpenv_set_vessel_params_ex(h, tab, DPENV_VESSEL_KEEP_RANDOMISATION, stream) == DPENV_EINVAL &&
                  strstr(dpenv_last_error(h), "KEEP_RANDOMISATION") != NULL;
    CHECK(dpenv_get_vessel_params(h, tab, stream));                             /* still the preset, every env */
    hipStreamSynchronize(stream);
    hipMemcpy(htab, tab, 4 * nt, hipMemcpyDeviceToHost);
    for (int p = 0; p < DPENV_NPARAM; ++p)
        for (int i = 0; i < n3; ++i) refused = refused && htab[(size_t)p * n3 + i] == preset[p];
    s = 4711u;
    for (int p = 0; p < DPENV_NPARAM; ++p)
        for (int i = 0; i < n3; ++i) {
            s = s * 1664525u + 1013904223u;
            htab[(size_t)p * n3 + i] = preset[p] * (1.0f + 0.2f * ((float)(s >> 8) / 16777216.0f - 0.5f));
        }
    hipMemcpy(tab, htab, 4 * nt, hipMemcpyHostToDevice);
    CHECK(dpenv_set_vessel_params(h, tab, stream));                             /* stream-ordered: no read-back (the kernels learn on the device whether any env carries a loss) */
    float* hact3 = (float*)malloc(4 * (size_t)n3 * ad);
    float* hobs3 = (float*)malloc(4 * (size_t)n3 * od);
    float* hrew3 = (float*)malloc(4 * (size_t)n3);
    double sums[2][3];
    for (int phase = 0; phase < 2; ++phase) {
        if (phase == 1) {                                                       /* hulls re-drawn by every reset, +-15 % around the preset, coefficients included */
            for (int p = 0; p < DPENV_NPARAM; ++p) range[p] = 0.15f;
            CHECK(dpenv_set_vessel_randomisation(h, preset, range, stream));
        }
        CHECK(dpenv_reset(h, NULL, NULL, NULL, NULL, stream));
        double osum = 0.0, rsum = 0.0;
        s = 99u + (unsigned)phase;
        for (int t = 0; t < S3; ++t) {
            for (size_t i = 0; i < (size_t)n3 * ad; ++i) {
                s = s * 1664525u + 1013904223u;
                hact3[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 1.6f;
            }
            hipMemcpy(act3, hact3, 4 * (size_t)n3 * ad, hipMemcpyHostToDevice);
            CHECK(dpenv_step(h, act3, NULL, obs3, rew3, done3, stream));
            hipStreamSynchronize(stream);
            hipMemcpy(hobs3, obs3, 4 * (size_t)n3 * od, hipMemcpyDeviceToHost);
            hipMemcpy(hrew3, rew3, 4 * (size_t)n3, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < (size_t)n3 * od; ++i) osum += hobs3[i];
            for (int i = 0; i < n3; ++i) rsum += hrew3[i];
        }
        CHECK(dpenv_get_vessel_params(h, tab, stream));
        hipStreamSynchronize(stream);
        hipMemcpy(htab, tab, 4 * nt, hipMemcpyDeviceToHost);
        double tsum = 0.0;
        for (size_t i = 0; i < nt; ++i) tsum += htab[i];
        sums[phase][0] = osum; sums[phase][1] = rsum; sums[phase][2] = tsum;
    }
    printf("c_abi_demo vessels: %d envs x %d steps on the thrust-loss preset; preset read back %s; refused calls %s; per-env blocks: checksums obs %.17g rew %.17g table %.17g; "
           "randomised: obs %.17g rew %.17g table %.17g\n", n3, S3, same ? "exactly" : "DIFFERENT", refused ? "left the handle alone" : "CHANGED THE HANDLE", sums[0][0],
           sums[0][1], sums[0][2], sums[1][0], sums[1][1], sums[1][2]);
    CHECK(dpenv_destroy(h));
    return same && refused ? 0 : 3;
}
