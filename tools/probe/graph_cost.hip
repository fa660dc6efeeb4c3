// Probe: does a hipGraph shorten the GPU-side gap between dependent kernels?  N empty (and N tiny-work) kernels in one stream,
// launched (a) one by one, (b) as a captured graph.  us per kernel in both forms.
// hipcc --offload-arch=gfx950 -O3 tools/probe/graph_cost.hip -o tools/probe/graph_cost.bin && tools/probe/graph_cost.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty(float* p) { if (p && threadIdx.x == 9999) *p = 0.f; }
__global__ void k_spin(float* p, long long cycles) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (p && threadIdx.x == 9999) *p = 0.f; }
__global__ void k_small(float* p, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
    float* buf; CK(hipMalloc(&buf, 1 << 24)); CK(hipMemset(buf, 0, 1 << 24));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 500;
    for (int form = 0; form < 5; ++form) {          // 0 empty 1 block, 1 empty 1024 blocks, 2 small work 1024 blocks, 3 / 4: 1024 blocks spinning 10 / 30 us (100 MHz wall clock)
        auto launch = [&]() {
            if (form == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, buf);
            else if (form == 1) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, buf);
            else if (form == 2) hipLaunchKernelGGL(k_small, dim3(1024), dim3(256), 0, s, buf, 1024 * 256);
            else hipLaunchKernelGGL(k_spin, dim3(1024), dim3(256), 0, s, buf, form == 3 ? 1000LL : 3000LL);
        };
        for (int i = 0; i < 50; ++i) launch();
        CK(hipStreamSynchronize(s));
        float ms_stream = 0, ms_graph = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) launch();
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_stream, e0, e1));
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) launch();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_graph, e0, e1));
        }
        printf("form %d: stream %.2f us/kernel, graph %.2f us/kernel\n", form, ms_stream * 1000 / N, ms_graph * 1000 / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
