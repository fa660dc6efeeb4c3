// Squeeze-and-excitation for the MBConv block (BASELINE config 4: "5x5-depthwise + SE-block variant").  BUILD-DEFINED: the
// reference has no SE block anywhere, so the definition is this repo's (MnasNet-A1 form), restated in
// oracle/mnasnet_oracle.py::se_apply -- "parity unpinned by the reference":
//     a2 = relu(bn2(y2))                       the activated depthwise output (N,H,W,E)
//     z  = mean_hw a2                          squeeze          -> mnas_pool_act (existing)
//     h  = relu(fc1 z + b1),  u = fc2 h + b2   excite (R << E)  -> mnas_head_linear_fwd (existing GEMM)
//     a2s = a2 * sigmoid(u)[n][e]              scale            -> k_se_scale: the project conv reads a2s as a plain activation
// Backward (gs = dL/da2s from the project conv's input gradient):
//     du[n][e] = (sum_hw gs * a2) * s (1 - s)                   -> k_se_bwd_reduce
//     MLP backward                                              -> mnas_head_linear_bwd_w / _bwd_x (existing)
//     g_a2 = gs * s + dz[n][e] / HW                             -> k_se_bwd_apply  (dz = dL/dz from the MLP backward)
// All three kernels are elementwise / per-image reductions over the E-wide tensor: HBM-bound (read y2 [+ gs], write one tensor).
#include "mnas_common.h"

__device__ __forceinline__ float se_sigmoid(float u) { return 1.f / (1.f + __expf(-u)); }

// one thread per 16-byte channel group of one pixel
__global__ __launch_bounds__(256) void k_se_scale(MnasActIn a, const float* __restrict__ u, int N, int HW, int C, uint4* __restrict__ out) {
    const int G = C >> 3;
    const size_t total = (size_t)N * HW * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % G);
        const size_t pix = i / G;
        const int n = (int)(pix / HW);
        float f[8];
        unpack8(((const uint4*)a.data)[i], f);
        const float* up = u + (size_t)n * C + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = f[j];
            if (a.scale) v = fmaxf(fmaf(v, a.scale[cg * 8 + j], a.shift[cg * 8 + j]), 0.f);
            f[j] = v * se_sigmoid(up[j]);
        }
        out[i] = pack8(f);
    }
}

// one workgroup per (image, 64-channel block): thread (pg = tid >> 3 in 0..31 walks the pixels, cg = tid & 7 one channel group)
__global__ __launch_bounds__(256) void k_se_bwd_reduce(const uint4* __restrict__ gs, MnasActIn a, const float* __restrict__ u, int HW,
                                                       int C, float* __restrict__ du) {
    __shared__ float red[32][65];
    const int n = blockIdx.x, cb = blockIdx.y * 64;
    const int cg = threadIdx.x & 7, pg = threadIdx.x >> 3;
    const int c0 = cb + cg * 8;
    const int G = C >> 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
        float s[8], t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = a.scale ? a.scale[c0 + j] : 1.f; t[j] = a.scale ? a.shift[c0 + j] : 0.f; }
        for (int p = pg; p < HW; p += 32) {
            const size_t idx = ((size_t)n * HW + p) * G + (c0 >> 3);
            float g[8], y[8];
            unpack8(gs[idx], g);
            unpack8(((const uint4*)a.data)[idx], y);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = a.scale ? fmaxf(fmaf(y[j], s[j], t[j]), 0.f) : y[j];
                acc[j] = fmaf(g[j], v, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pg][cg * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64 && cb + threadIdx.x < C) {
        float v = 0.f;
        for (int r = 0; r < 32; ++r) v += red[r][threadIdx.x];         // fixed order: deterministic
        const float sg = se_sigmoid(u[(size_t)n * C + cb + threadIdx.x]);
        du[(size_t)n * C + cb + threadIdx.x] = v * sg * (1.f - sg);
    }
}

__global__ __launch_bounds__(256) void k_se_bwd_apply(const uint4* __restrict__ gs, const float* __restrict__ u, const float* __restrict__ dz,
                                                      int N, int HW, int C, uint4* __restrict__ out) {
    const int G = C >> 3;
    const size_t total = (size_t)N * HW * G;
    const float inv = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % G);
        const int n = (int)((i / G) / HW);
        float g[8];
        unpack8(gs[i], g);
        const float* up = u + (size_t)n * C + cg * 8;
        const float* zp = dz + (size_t)n * C + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = fmaf(g[j], se_sigmoid(up[j]), zp[j] * inv);
        out[i] = pack8(g);
    }
}

static int se_grid(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
extern "C" int mnas_se_scale(const MnasActIn* a, const float* u, int N, int HW, int C, void* out_bf16, void* stream) {
    if (!a || !a->data || !u || !out_bf16 || N < 1 || HW < 1 || C < 8 || (C & 7)) return MNAS_EINVAL;
    if ((a->scale == nullptr) != (a->shift == nullptr)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_scale, dim3(se_grid((size_t)N * HW * (C >> 3))), dim3(256), 0, (hipStream_t)stream, *a, u, N, HW, C, (uint4*)out_bf16);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
extern "C" int mnas_se_bwd_reduce(const void* gs, const MnasActIn* a, const float* u, int N, int HW, int C, float* du, void* stream) {
    if (!gs || !a || !a->data || !u || !du || N < 1 || HW < 1 || C < 8 || (C & 7)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_bwd_reduce, dim3(N, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const uint4*)gs, *a, u, HW, C, du);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
extern "C" int mnas_se_bwd_apply(const void* gs, const float* u, const float* dz, int N, int HW, int C, void* out_bf16, void* stream) {
    if (!gs || !u || !dz || !out_bf16 || N < 1 || HW < 1 || C < 8 || (C & 7)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_bwd_apply, dim3(se_grid((size_t)N * HW * (C >> 3))), dim3(256), 0, (hipStream_t)stream, (const uint4*)gs, u, dz, N, HW,
                       C, (uint4*)out_bf16);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
