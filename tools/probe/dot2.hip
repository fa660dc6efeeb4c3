// Throughput probe: v_dot2c_f32_bf16 vs v_fma_f32 (issue rate per wave), 8 independent accumulators, 256 CUs x 8 waves.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/dot2.hip -o tools/probe/dot2_bin ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
    float acc[8];
    unsigned a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    for (int j = 0; j < 8; ++j) acc[j] = (float)j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) acc[j] = __builtin_amdgcn_fdot2_f32_bf16(*(bf2*)&a, *(bf2*)&b, acc[j], false);
            else acc[j] = __builtin_fmaf(__uint_as_float(a), __uint_as_float(b), acc[j]);
        }
        a += 0x10001u;
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d, iters, 0x3f803f80u);
            else hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, d, iters, 0x3f803f80u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts = 2048.0 * 4 * iters * 8;            // wave-instructions
            printf("%s: %.3f ms, %.2f wave-instr/ns chip-wide, %.2f cycles per wave-instr per SIMD (2.4 GHz, 1024 SIMDs)\n",
                   mode == 0 ? "v_dot2c_f32_bf16" : "v_fma_f32", ms, insts / (ms * 1e6), ms * 1e6 * 2.4 * 1024 / insts);
        }
    }
    return 0;
}
