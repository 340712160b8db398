/* c_abi_demo.c - libdpenv.so from plain C: no Python, no torch.  Allocates the I/O buffers with the HIP runtime,
 * resets 65 536 environments, steps them 2 000 times with a fixed action block, and prints throughput and a
 * checksum.  Build (see tests/test_gpu_c_abi.py):
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
    return faults ? 2 : 0;
}
