// Probe: what does a "last workgroup finalizes" scheme with deterministic (integer) device-scope atomics cost?
// N workgroups each add C 64-bit values into C accumulators, then take a ticket; the last one reads the C sums.
// hipcc --offload-arch=gfx950 -O3 tools/probe/atomics.hip -o /tmp/atomics && /tmp/atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty(unsigned long long* acc, unsigned* ticket, float* out, int C) {}
__global__ void k_atom(unsigned long long* acc, unsigned* ticket, float* out, int C) {
    __shared__ int last;
    const int tid = threadIdx.x;
    for (int c = tid; c < C; c += blockDim.x) atomicAdd(&acc[c], (unsigned long long)(blockIdx.x * 131 + c));
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (last) {
        __threadfence();
        for (int c = tid; c < C; c += blockDim.x) {
            out[c] = (float)__hip_atomic_load(&acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc[c] = 0;
        }
        if (tid == 0) *ticket = 0;
    }
}
int main() {
    unsigned long long* acc; unsigned* ticket; float* out;
    hipMalloc(&acc, 4096 * 8); hipMalloc(&ticket, 4); hipMalloc(&out, 4096 * 4);
    hipMemset(acc, 0, 4096 * 8); hipMemset(ticket, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int Ns[] = {256, 512, 1024, 2048}, Cs[] = {32, 96, 192, 1152};
    for (int N : Ns) for (int C : Cs) {
        float ms[2];
        for (int v = 0; v < 2; ++v) {
            for (int i = 0; i < 5; ++i) { if (v) hipLaunchKernelGGL(k_atom, dim3(N), dim3(256), 0, 0, acc, ticket, out, C); else hipLaunchKernelGGL(k_empty, dim3(N), dim3(256), 0, 0, acc, ticket, out, C); }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) { if (v) hipLaunchKernelGGL(k_atom, dim3(N), dim3(256), 0, 0, acc, ticket, out, C); else hipLaunchKernelGGL(k_empty, dim3(N), dim3(256), 0, 0, acc, ticket, out, C); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[v], e0, e1);
        }
        printf("N=%4d workgroups x C=%4d atomics: %.2f us per launch (empty kernel %.2f us)\n", N, C, ms[1] * 1000 / 50, ms[0] * 1000 / 50);
    }
    return 0;
}
