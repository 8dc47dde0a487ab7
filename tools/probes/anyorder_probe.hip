// anyorder_probe.hip -- does hipExtAnyOrderLaunch let INDEPENDENT kernels of ONE stream overlap on gfx950 (AQL barrier bit cleared)?
// hip_ext.h says "not supported on AMD GFX9xx boards" for the module variant; this measures it.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/anyorder_probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
// Three arms, each NREP times: N small kernels (G workgroups spinning ~T us) back to back on one stream
//   (a) flags = 0, (b) flags = hipExtAnyOrderLaunch on all but the first, (c) one kernel per stream over N streams (the upper bound).
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(float* out, long clocks) {
    const long t0 = wall_clock64();
    float v = (float)threadIdx.x;
    while (wall_clock64() - t0 < clocks) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) out[blockIdx.x] = v;
}
// dependent chain check: kernel i adds 1 to every element after reading it (ordering errors show up as a wrong sum)
__global__ void inc_kernel(float* x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += 1.f;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4, G = argc > 2 ? atoi(argv[2]) : 16;
    const double us = argc > 3 ? atof(argv[3]) : 20.0;
    const int NREP = 200;
    CK(hipSetDevice(0));
    float* out = nullptr;
    CK(hipMalloc(&out, 1 << 20));
    const long clocks = (long)(us * 100.0);   // wall_clock64 ticks at 100 MHz
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipStream_t> ss(N);
    for (auto& x : ss) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int arm = 0; arm < 3; ++arm) {
        float best = 1e30f, sum = 0.f;
        for (int rep = 0; rep < NREP + 10; ++rep) {
            CK(hipStreamSynchronize(s));
            for (auto& x : ss) CK(hipStreamSynchronize(x));
            CK(hipEventRecord(e0, s));
            if (arm < 2) {
                for (int i = 0; i < N; ++i)
                    hipExtLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), 0, s, nullptr, nullptr, (arm == 1 && i > 0) ? hipExtAnyOrderLaunch : 0, out, clocks);
            } else {
                hipEvent_t ev[64];
                CK(hipEventRecord(e1, s));
                for (int i = 0; i < N; ++i) {
                    CK(hipStreamWaitEvent(ss[i], e1, 0));
                    hipLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), 0, ss[i], out, clocks);
                    CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
                    CK(hipEventRecord(ev[i], ss[i]));
                    CK(hipStreamWaitEvent(s, ev[i], 0));
                }
                for (int i = 0; i < N; ++i) CK(hipEventDestroy(ev[i]));
            }
            // a dependent (ordinary) launch closes the group, as the next network layer would
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, out, 100L);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 10) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("arm %d (%s): N=%d kernels of %d workgroups x %.0f us: best %.1f us, mean %.1f us\n", arm,
               arm == 0 ? "in order" : arm == 1 ? "hipExtAnyOrderLaunch" : "one stream each", N, G, us, best * 1e3f, sum / NREP * 1e3f);
    }
    // ordering: a dependent launch after any-order launches must still wait for all of them
    float* x = nullptr;
    const int n = 1 << 20;
    CK(hipMalloc(&x, n * 4));
    CK(hipMemsetAsync(x, 0, n * 4, s));
    for (int i = 0; i < 50; ++i) {
        hipExtLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, out, clocks);
        hipLaunchKernelGGL(inc_kernel, dim3(n / 256), dim3(256), 0, s, x, n);
    }
    std::vector<float> h(n);
    CK(hipMemcpyAsync(h.data(), x, n * 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    long bad = 0;
    for (int i = 0; i < n; ++i) bad += h[i] != 50.f;
    printf("dependent chain interleaved with any-order launches: %ld wrong of %d\n", bad, n);
    return 0;
}
