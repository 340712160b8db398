// Floor of a dependent one-launch-per-step chain at the benchmark geometry (1024 workgroups x 64 threads):
// (a) empty kernels, (b) kernels that only move the step's bytes (4 float4 in, 3 float4 + 9 floats + 1 float + 1 byte
// out per thread, 7 floats of action in), captured 50-per-graph like bench.py.  Prints us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int) {}
__global__ void k_move(const float4* __restrict__ s_in, float4* __restrict__ s_out, const float* __restrict__ act,
                       float* __restrict__ obs, float* __restrict__ rew, unsigned char* __restrict__ done, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    float4 a = s_in[i], b = s_in[n + i], c = s_in[2 * n + i], d = s_in[3 * n + i];
    float acc = 0.f;
    for (int j = 0; j < 7; ++j) acc += act[(size_t)blockIdx.x * 448 + j * 64 + threadIdx.x];
    a.x += acc; b.y += d.x; c.z += d.y;
    s_out[i] = a; s_out[n + i] = b; s_out[2 * n + i] = c;
    for (int j = 0; j < 9; ++j) obs[(size_t)blockIdx.x * 576 + j * 64 + threadIdx.x] = a.x + j;
    rew[i] = b.x; done[i] = (unsigned char)(c.x > 0);
}
template <typename F> float run(F launch)
{
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int k = 0; k < 50; ++k) launch(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 40; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / (40 * 50);
}
int main()
{
    const int n = 65536;
    float4 *sa, *sb; float *act, *obs, *rew; unsigned char* done;
    hipMalloc(&sa, 4 * n * 16); hipMalloc(&sb, 4 * n * 16); hipMalloc(&act, n * 28); hipMalloc(&obs, n * 36);
    hipMalloc(&rew, n * 4); hipMalloc(&done, n);
    hipMemset(sa, 0, 4 * n * 16); hipMemset(sb, 0, 4 * n * 16); hipMemset(act, 0, n * 28);
    printf("empty kernel chain      : %.2f us per launch\n", run([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, s, 0); }));
    printf("byte-moving kernel chain: %.2f us per launch (in-place state: each launch reads what the previous wrote)\n",
           run([&](hipStream_t s) { hipLaunchKernelGGL(k_move, dim3(1024), dim3(64), 0, s, sa, sa, act, obs, rew, done, n); }));
    return 0;
}
