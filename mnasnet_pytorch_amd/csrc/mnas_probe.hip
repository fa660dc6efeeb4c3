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

// The same copy built the way the streaming kernels of the step are built: FOUR independent 16-byte loads in flight per lane
// (a workgroup moves 16 KB per trip, each wave four consecutive 1 KB lines), nontemporal loads and stores (nothing is re-read).
// This -- not the plain grid-stride copy above, which keeps one load per lane in flight -- is the rate a kernel of the step may
// be compared with (round 6: k_pw_bwd 16->48 at 112^2 sustained more algorithmic bytes than k_probe_copy did).
__global__ __launch_bounds__(256) void k_probe_copy4(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, size_t n) {
    const size_t st = (size_t)gridDim.x * 1024;
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + 768 < n; i += st) {
        const u32x4_t a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256);
        const u32x4_t c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + 256);
        __builtin_nontemporal_store(c, dst + i + 512);
        __builtin_nontemporal_store(d, dst + i + 768);
    }
}

// read-only stream: four independent 16-byte loads in flight per lane, folded with XOR; the sink is never written (the
// comparison value cannot occur), it only keeps the loads alive.  rate = bytes / time.
__global__ __launch_bounds__(256) void k_probe_read(const u32x4_t* __restrict__ src, uint32_t* __restrict__ sink, size_t n) {
    const size_t st = (size_t)gridDim.x * 1024;
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    for (; i + 768 < n; i += st) {
        const u32x4_t a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256);
        const u32x4_t c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
        acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y;
        acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && sink) sink[blockIdx.x] = acc.x;
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

extern "C" int mnas_probe_copy4(const void* src, void* dst, int64_t bytes, int blocks, void* stream) {
    if (!src || !dst || bytes < 16384 || (bytes & 16383) || blocks < 0) return MNAS_EINVAL;      // whole 16 KB trips only
    const size_t n = (size_t)(bytes >> 4);
    const size_t trips = n / 1024;
    const int g = blocks > 0 ? blocks : (int)(trips < 65536 ? trips : 65536);
    hipLaunchKernelGGL(k_probe_copy4, dim3(g), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)src, (u32x4_t*)dst, n);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_probe_read(const void* src, void* sink, int64_t bytes, int blocks, void* stream) {
    if (!src || !sink || bytes < 16384 || (bytes & 16383) || blocks < 0) return MNAS_EINVAL;     // whole 16 KB trips only
    const size_t n = (size_t)(bytes >> 4);
    const size_t trips = n / 1024;
    const int g = blocks > 0 ? blocks : (int)(trips < 65536 ? trips : 65536);
    hipLaunchKernelGGL(k_probe_read, dim3(g), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)src, (uint32_t*)sink, n);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_probe_valu(float* out, int blocks, int iters, void* stream) {
    if (!out || blocks < 1 || iters < 1) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_probe_valu, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
