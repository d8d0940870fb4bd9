// graph_launch.hip - how fast does a captured hipGraph replay a long dependent chain of small kernels, alone and
// with several graphs replayed concurrently on separate streams?  (basis for the device-side sweep schedule)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(long long cycles, double *sink)
{
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) { }
    if (sink && threadIdx.x == 9999) sink[0] = 1.0;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 3000;
    double *sink;
    CK(hipMalloc(&sink, 8));
    for (long long cyc : {0ll, 1000ll, 10000ll, 25000ll}) {   // s_memtime ticks at 100 MHz: 1000 = 10 us ? (printed below)
        for (int nstreams : {1, 2, 4}) {
            std::vector<hipStream_t> st(nstreams);
            std::vector<hipGraph_t> g(nstreams);
            std::vector<hipGraphExec_t> ge(nstreams);
            for (int i = 0; i < nstreams; ++i) {
                CK(hipStreamCreate(&st[i]));
                CK(hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < N; ++k) hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 0, st[i], cyc, sink);
                CK(hipStreamEndCapture(st[i], &g[i]));
                CK(hipGraphInstantiate(&ge[i], g[i], nullptr, nullptr, 0));
            }
            // direct launches
            CK(hipDeviceSynchronize());
            double t0 = now();
            for (int k = 0; k < N; ++k)
                for (int i = 0; i < nstreams; ++i) hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 0, st[i], cyc, sink);
            double t_issue = now() - t0;
            CK(hipDeviceSynchronize());
            double t_direct = now() - t0;
            // graph replay (second replay timed)
            for (int i = 0; i < nstreams; ++i) CK(hipGraphLaunch(ge[i], st[i]));
            CK(hipDeviceSynchronize());
            t0 = now();
            for (int i = 0; i < nstreams; ++i) CK(hipGraphLaunch(ge[i], st[i]));
            double t_gissue = now() - t0;
            CK(hipDeviceSynchronize());
            double t_graph = now() - t0;
            printf("cycles %6lld streams %d launches/stream %d | direct: issue %.2f us/launch/stream, total %.2f us | graph: issue %.3f ms, total %.2f us/launch/stream\n",
                   cyc, nstreams, N, t_issue / N * 1e6, t_direct / N * 1e6, t_gissue * 1e3, t_graph / N * 1e6);
            for (int i = 0; i < nstreams; ++i) {
                CK(hipGraphExecDestroy(ge[i])); CK(hipGraphDestroy(g[i])); CK(hipStreamDestroy(st[i]));
            }
        }
    }
    return 0;
}
