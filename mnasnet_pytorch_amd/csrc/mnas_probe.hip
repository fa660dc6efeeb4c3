// Box calibration probes (bench.py "box" object): what this GPU sustains for a plain streaming copy and for a pure packed-FMA
// loop, measured next to the timed step windows.  They exist because boxes of one pool differ (shader clock / power state): a
// step rate only means something next to the copy rate and the vector-ALU rate of the box it was taken on.  Not on the step path.
#include "mnas_common.h"

typedef float pf2 __attribute__((ext_vector_type(2)));

// dst[i] = src[i], 16 bytes per lane, grid-stride (the pattern of tools/probe/bw.hip's 1:1 mix)
__global__ __launch_bounds__(256) void k_probe_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) dst[i] = src[i];
}

// 8 independent v_pk_fma_f32 chains per lane, nothing else in the loop: flops = blocks * 256 * iters * 8 * 2 lanes * 2.
// At one packed FMA per SIMD per 4 cycles the chip does 256 CUs * 4 SIMDs * 16 lanes * 2 * 2 = 65 536 flop per clock, so
// TFLOP/s / 65.536 = the shader clock in GHz this box sustains under vector load.
__global__ __launch_bounds__(256) void k_probe_valu(float* __restrict__ out, int iters) {
    pf2 acc[8];
    const pf2 a = {1.0000001f, 0.9999999f}, b = {1e-7f, -1e-7f};
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[j].x = (float)(threadIdx.x + j); acc[j].y = (float)(blockIdx.x + j); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_elementwise_fma(acc[j], a, b);
    }
    pf2 s = acc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) s += acc[j];
    if (s.x + s.y == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s.x;      // never true: keeps the chains alive
}

// a kernel that does nothing: what one dependent launch costs on this box (the floor under the 138 bookkeeping launches of a step)
__global__ void k_probe_empty(int* p) { if (p && threadIdx.x == 1024) *p = 0; }
extern "C" int mnas_probe_empty(int blocks, int threads, void* stream) {
    if (blocks < 1 || threads < 1 || threads > 1024) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_probe_empty, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (int*)nullptr);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_probe_copy(const void* src, void* dst, int64_t bytes, void* stream) {
    if (!src || !dst || bytes < 16 || (bytes & 15)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_probe_copy, dim3(8192), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, (size_t)(bytes >> 4));
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_probe_valu(float* out, int blocks, int iters, void* stream) {
    if (!out || blocks < 1 || iters < 1) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_probe_valu, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
