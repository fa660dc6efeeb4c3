// BatchNorm bookkeeping, residual add, layout conversions, weight packing, wgrad reductions, fused Adam.
// All of these are HBM- or latency-bound helpers around the conv kernels; see include/mnas.h for the ABI.
#include "mnas_common.h"
#ifndef MNAS_FIN_HOIST
#define MNAS_FIN_HOIST 1
#endif

// ------------------------------------------------------------------------------------------------
// BatchNorm forward finalize  (ATen native_batch_norm's statistics step; mnasnet.py:55,60)
// partial layout: float[2][C][nparts] (channel-major: one wave reads one channel's partials contiguously)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum one channel's two partial rows with TPC threads (64: one wave per channel, 4 channels per block; 256: the whole
// block on one channel -- the early layers have 16-48 channels x 1024-2048 partials, where a wave per channel leaves
// the chip empty and walks 32 dependent-latency rounds).  Result valid in every thread with (threadIdx.x % TPC) == 0.
template <int TPC>
__device__ __forceinline__ void bn_partial_sums(const float* __restrict__ partial, int nparts, int C, int c, double& s1,
                                                double& s2) {
    const int t = threadIdx.x % TPC;
    const float* p1 = partial + (size_t)c * nparts;
    const float* p2 = partial + ((size_t)C + c) * nparts;
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    // every pass issues its 8 loads together (predicated, no dependent tail loop: these launches are a chain of
    // memory round trips and little else)
    for (int p = t; p < nparts; p += 4 * TPC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pj = p + j * TPC;
            const bool ok = pj < nparts;
            a1[j] += ok ? p1[pj] : 0.f;
            a2[j] += ok ? p2[pj] : 0.f;
        }
    }
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1 += (double)a1[j]; s2 += (double)a2[j]; }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (TPC == 256) {
        __shared__ double red[2][4];
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
        __syncthreads();
        s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

template <int TPC>
__global__ __launch_bounds__(256) void k_bn_fwd_finalize(
    const float* __restrict__ partial, int nparts, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, int64_t* nbt, float momentum,
    float eps, int training, float* bnbuf) {
    const int lane = threadIdx.x % TPC;
    const int c = (TPC == 256) ? blockIdx.x : blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c < C) {
        float s, t;
        if (training) {
            // everything the tail needs is loaded BEFORE the partial sums: these launches are chains of memory round trips, and
            // gamma / beta / the running statistics after the reduce were a second one (MNAS_FIN_HOIST=0: A/B builds)
#if MNAS_FIN_HOIST
            const float g0 = gamma[c], b0 = beta[c], rm0 = running_mean[c], rv0 = running_var[c];
#endif
            double s1, s2;
            bn_partial_sums<TPC>(partial, nparts, C, c, s1, s2);
#if !MNAS_FIN_HOIST
            const float g0 = gamma[c], b0 = beta[c];
#endif
            const double mean = s1 / count;
            double var = s2 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const double invstd = 1.0 / sqrt(var + (double)eps);
            s = (float)((double)g0 * invstd);
            t = (float)((double)b0 - mean * (double)g0 * invstd);
            if (lane == 0) {
#if !MNAS_FIN_HOIST
                const float rm0 = running_mean[c], rv0 = running_var[c];
#endif
                bnbuf[5 * C + c] = (float)mean;
                bnbuf[6 * C + c] = (float)invstd;
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                running_mean[c] = (float)((1.0 - momentum) * (double)rm0 + momentum * mean);
                running_var[c] = (float)((1.0 - momentum) * (double)rv0 + momentum * unbiased);
            }
        } else {
            const float invstd = 1.0f / sqrtf(running_var[c] + eps);
            s = gamma[c] * invstd;
            t = beta[c] - running_mean[c] * s;
        }
        if (lane == 0) {
            bnbuf[0 * C + c] = s;
            bnbuf[1 * C + c] = t;
        }
    }
    if (training && nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

extern "C" int mnas_bn_fwd_finalize(const float* partial, int nparts, int C, double count, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var,
                                    int64_t* num_batches_tracked, float momentum, float eps, int training,
                                    float* bnbuf, void* stream) {
    if (C <= 0 || (training && (!partial || nparts <= 0))) return MNAS_EINVAL;
    if (training && nparts > 256)
        hipLaunchKernelGGL(k_bn_fwd_finalize<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nparts, C,
                           count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, training,
                           bnbuf);
    else
        hipLaunchKernelGGL(k_bn_fwd_finalize<64>, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial, nparts, C,
                           count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, training,
                           bnbuf);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward reduce: per-channel sum(dz), sum(dz*xhat)   (native_batch_norm_backward, pass 1)
// rows x C bf16; thread = (row lane, channel group) with the channel group fixed per thread so the
// coefficients stay in registers across the persistent row loop.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const uint4* __restrict__ g, const uint4* __restrict__ y,
                                                       const float* __restrict__ bnbuf, int64_t rows, int C,
                                                       float* __restrict__ partial) {
    extern __shared__ float red[];   // [R][2][C]: one slot per (row lane, channel), summed in row-lane order (deterministic)
    const int G = C >> 3;
    const int R = 256 / G;
    const int tid = threadIdx.x;
    const int64_t chunk = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * chunk;
    const int64_t r1 = min(rows, r0 + chunk);
    if (tid < R * G) {
        const int cg = tid % G, rl = tid / G;
        float s[8], t[8], mu[8], is[8], a1[8], a2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s[j] = bnbuf[0 * C + cg * 8 + j];
            t[j] = bnbuf[1 * C + cg * 8 + j];
            is[j] = bnbuf[6 * C + cg * 8 + j];
            mu[j] = -bnbuf[5 * C + cg * 8 + j] * is[j];
            a1[j] = 0.f;
            a2[j] = 0.f;
        }
        for (int64_t r = r0 + rl; r < r1; r += R) {
            const uint4 gv = g[r * G + cg];
            const uint4 yv = y[r * G + cg];
            float gf[8], yf[8];
            unpack8(gv, gf);
            unpack8(yv, yf);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dz = (fmaf(yf[j], s[j], t[j]) > 0.f) ? gf[j] : 0.f;
                a1[j] += dz;
                a2[j] = fmaf(dz, fmaf(yf[j], is[j], mu[j]), a2[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[(rl * 2 + 0) * C + cg * 8 + j] = a1[j];
            red[(rl * 2 + 1) * C + cg * 8 + j] = a2[j];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) {
        float v = red[i];
        for (int rl = 1; rl < R; ++rl) v += red[rl * 2 * C + i];
        partial[(size_t)i * gridDim.x + blockIdx.x] = v;   // [2][C][P]
    }
}

extern "C" int mnas_bn_bwd_reduce(const void* g, const void* y, const float* bnbuf, int64_t rows, int C, int nparts,
                                  float* partial, void* stream) {
    if (C <= 0 || (C & 7) || C > 2048 || nparts <= 0) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3(nparts), dim3(256), (size_t)(256 / (C >> 3)) * 2 * C * sizeof(float), (hipStream_t)stream,
                       (const uint4*)g, (const uint4*)y, bnbuf, rows, C, partial);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ------------------------------------------------------------------------------------------------
// dy = c1*(g*[s*y+t>0]) + c2*y + c3 materialised as bf16 (rows x C).  The dense 3x3 convs gather every dy element 9 times
// (stride 1) / 2.25 times (stride 2) in their input gradient and once more in their weight gradient: forming dy on load
// there means two tensor reads and the 5-coefficient transform per gathered 16 bytes, and those launches are VALU-issue
// bound (k_igemm<dgrad, stride 2> at 112x112: 26 k VALU instructions per wave).  One elementwise pass over the (small,
// post-stride) tensor instead.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dy_mat(const uint4* __restrict__ g, const uint4* __restrict__ y,
                                                const float* __restrict__ coef, int64_t rows, int C, uint4* __restrict__ out) {
    const int G = C >> 3, R = 256 / G;
    const int tid = threadIdx.x;
    if (tid >= R * G) return;
    const int cg = tid % G, rl = tid / G;
    float cf[5][8];
#pragma unroll
    for (int r = 0; r < 5; ++r) {          // 16-byte loads (C % 8 == 0): 10 instead of 40 per thread -- most launches are one row per thread
        *(float4*)&cf[r][0] = *(const float4*)(coef + (size_t)r * C + cg * 8);
        *(float4*)&cf[r][4] = *(const float4*)(coef + (size_t)r * C + cg * 8 + 4);
    }
    for (int64_t r = (int64_t)blockIdx.x * R + rl; r < rows; r += (int64_t)gridDim.x * R) {
        float o[8];
        dy8(g[r * G + cg], y[r * G + cg], cf[0], cf[1], cf[2], cf[3], cf[4], o);
        out[r * G + cg] = pack8(o);
    }
}
extern "C" int mnas_dy_materialize(const MnasGradIn* d, int64_t rows, int C, void* out_bf16, void* stream) {
    if (!d || !d->g || !d->y || !d->coef || !out_bf16 || rows < 1 || C <= 0 || (C & 7) || C > 2048) return MNAS_EINVAL;
    const int R = 256 / (C >> 3);
    int64_t blocks = (rows + R - 1) / R;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_dy_mat, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)d->g, (const uint4*)d->y,
                       d->coef, rows, C, (uint4*)out_bf16);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// dgamma, dbeta and the dy-on-load coefficients:
//   dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) = c1*dz + c2*y + c3
template <int TPC>
__device__ __forceinline__ void bn_bwd_finalize_body(const float* __restrict__ partial, int nparts, int C, double count, float* bnbuf,
                                                     float* dgamma, float* dbeta, int accumulate, int blk) {
    const int lane = threadIdx.x % TPC;
    const int c = (TPC == 256) ? blk : blk * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
#if MNAS_FIN_HOIST
    // (loaded ahead of the partial sums: see k_bn_fwd_finalize)
    const float fs = bnbuf[0 * C + c], fmean = bnbuf[5 * C + c], finv = bnbuf[6 * C + c];
    const float dg0 = (dgamma && accumulate) ? dgamma[c] : 0.f, db0 = (dbeta && accumulate) ? dbeta[c] : 0.f;
#endif
    double s1, s2;
    bn_partial_sums<TPC>(partial, nparts, C, c, s1, s2);
    if (lane == 0) {
#if !MNAS_FIN_HOIST
        const float fs = bnbuf[0 * C + c], fmean = bnbuf[5 * C + c], finv = bnbuf[6 * C + c];
        const float dg0 = (dgamma && accumulate) ? dgamma[c] : 0.f, db0 = (dbeta && accumulate) ? dbeta[c] : 0.f;
#endif
        const double s = fs, mean = fmean, invstd = finv;
        const double md = s1 / count, mx = s2 / count;
        bnbuf[2 * C + c] = (float)s;
        bnbuf[3 * C + c] = (float)(-s * invstd * mx);
        bnbuf[4 * C + c] = (float)(s * (mean * invstd * mx - md));
        if (dgamma) dgamma[c] = dg0 + (float)s2;
        if (dbeta) dbeta[c] = db0 + (float)s1;
    }
}
template <int TPC>
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ partial, int nparts, int C,
                                                         double count, float* bnbuf, float* dgamma, float* dbeta,
                                                         int accumulate) {
    bn_bwd_finalize_body<TPC>(partial, nparts, C, count, bnbuf, dgamma, dbeta, accumulate, blockIdx.x);
}

extern "C" int mnas_bn_bwd_finalize(const float* partial, int nparts, int C, double count, float* bnbuf, float* dgamma,
                                    float* dbeta, int accumulate, void* stream) {
    if (C <= 0 || nparts <= 0) return MNAS_EINVAL;
    if (nparts > 256)
        hipLaunchKernelGGL(k_bn_bwd_finalize<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nparts, C,
                           count, bnbuf, dgamma, dbeta, accumulate);
    else
        hipLaunchKernelGGL(k_bn_bwd_finalize<64>, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial, nparts, C,
                           count, bnbuf, dgamma, dbeta, accumulate);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ------------------------------------------------------------------------------------------------
// out = act(a) + act(b): the MBConv_block residual (mnasnet.py:133) and the features-output conversion
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_add_act(MnasActIn a, MnasActIn b, int64_t rows, int C, uint4* __restrict__ out,
                                                 float* __restrict__ out_nchw, int HW, int nt) {
    const int G = C >> 3;
    const int R = 256 / G;
    const int tid = threadIdx.x;
    const bool ha = a.scale != nullptr, hb = (b.data != nullptr) && (b.scale != nullptr);
    // coefficients through LDS: one 4-byte load per thread and table instead of 32 scalar loads per thread (on the 14x14 / 7x7
    // tensors a thread handles ONE 16-byte chunk, and the coefficient loads were 10x the instructions of the actual work)
    __shared__ __attribute__((aligned(16))) float coef[4][2048];
    for (int i = tid; i < C; i += 256) {
        coef[0][i] = ha ? a.scale[i] : 1.f; coef[1][i] = ha ? a.shift[i] : 0.f;
        coef[2][i] = hb ? b.scale[i] : 1.f; coef[3][i] = hb ? b.shift[i] : 0.f;
    }
    __syncthreads();
    if (tid >= R * G) return;
    const int cg = tid % G, rl = tid / G;
    float sa[8], ta[8], sb[8], tb[8];
    *(float4*)&sa[0] = *(const float4*)&coef[0][cg * 8]; *(float4*)&sa[4] = *(const float4*)&coef[0][cg * 8 + 4];
    *(float4*)&ta[0] = *(const float4*)&coef[1][cg * 8]; *(float4*)&ta[4] = *(const float4*)&coef[1][cg * 8 + 4];
    *(float4*)&sb[0] = *(const float4*)&coef[2][cg * 8]; *(float4*)&sb[4] = *(const float4*)&coef[2][cg * 8 + 4];
    *(float4*)&tb[0] = *(const float4*)&coef[3][cg * 8]; *(float4*)&tb[4] = *(const float4*)&coef[3][cg * 8 + 4];
    const uint4* pa = (const uint4*)a.data;
    const uint4* pb = (const uint4*)b.data;
    for (int64_t r = (int64_t)blockIdx.x * R + rl; r < rows; r += (int64_t)gridDim.x * R) {
        float fa[8], fb[8];
        unpack8(pa[r * G + cg], fa);
        if (ha) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fa[j] = fmaxf(fmaf(fa[j], sa[j], ta[j]), 0.f);
        }
        if (pb) {
            unpack8(pb[r * G + cg], fb);
            if (hb) {
#pragma unroll
                for (int j = 0; j < 8; ++j) fb[j] = fmaxf(fmaf(fb[j], sb[j], tb[j]), 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) fa[j] += fb[j];
        }
        if (out) st_u4(out + r * G + cg, pack8(fa), nt);
        if (out_nchw) {
            const int64_t n = r / HW, hw = r % HW;
#pragma unroll
            for (int j = 0; j < 8; ++j) out_nchw[(n * C + cg * 8 + j) * HW + hw] = fa[j];
        }
    }
}

extern "C" int mnas_add_act(const MnasActIn* a, const MnasActIn* b, int64_t rows, int C, void* out_bf16,
                            float* out_nchw_f32, int HW, void* stream) {
    if (!a || !a->data || C <= 0 || (C & 7) || C > 2048) return MNAS_EINVAL;
    MnasActIn bb = {nullptr, nullptr, nullptr};
    if (b) bb = *b;
    const int R = 256 / (C >> 3);
    int64_t blocks = (rows + R - 1) / R;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_add_act, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a, bb, rows, C,
                       (uint4*)out_bf16, out_nchw_f32, HW, (mnas_nt_mask() & MNAS_NT_ADD_ACT) ? 1 : 0);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ------------------------------------------------------------------------------------------------
// Global average pool of the features output, fused with the last ConvBlock's BatchNorm+ReLU: out[n][c] = mean_hw act(a)
// (classifiers.py:49,109: AdaptiveAvgPool2d(1) on features(x)) -- the 16 MB fp32 NCHW feature map is never written.
// One workgroup per image; thread = (channel group of 8, row lane); row lanes combined through LDS in fixed order.
// Backward: g[n][hw][c] = gpool[n][c] / HW as bf16 NHWC (the gradient of the features output the engine's backward starts from).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pool_act(MnasActIn a, int HW, int C, float* __restrict__ out) {
    extern __shared__ float red[];            // [R][C]
    const int G = C >> 3, R = 256 / G;
    const int tid = threadIdx.x, n = blockIdx.x;
    const int cg = tid % G, rl = tid / G;
    if (tid < R * G) {
        float sa[8], ta[8], acc[8];
        const bool ha = a.scale != nullptr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sa[j] = ha ? a.scale[cg * 8 + j] : 1.f;
            ta[j] = ha ? a.shift[cg * 8 + j] : 0.f;
            acc[j] = 0.f;
        }
        const uint4* p = (const uint4*)a.data + (size_t)n * HW * G;
        // 8 row loads in flight per thread (32 KB per CU), added in the original row order: the squeeze of the SE variant walks
        // the 112x112 / 56x56 depthwise outputs with ONE workgroup per image (186 us for 308 MB with one load in flight, round 4)
        for (int r0 = rl; r0 < HW; r0 += 8 * R) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u * R;
                v[u] = make_uint4(0, 0, 0, 0);
                if (r < HW) v[u] = p[(size_t)r * G + cg];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (r0 + u * R >= HW) break;
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += ha ? fmaxf(fmaf(f[j], sa[j], ta[j]), 0.f) : f[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[rl * C + cg * 8 + j] = acc[j];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float v = red[c];
        for (int rl2 = 1; rl2 < R; ++rl2) v += red[rl2 * C + c];
        out[(size_t)n * C + c] = v / (float)HW;
    }
}
extern "C" int mnas_pool_act(const MnasActIn* a, int N, int HW, int C, float* out, void* stream) {
    if (!a || !a->data || !out || N < 1 || HW < 1 || C <= 0 || (C & 7) || C > 2048) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_pool_act, dim3(N), dim3(256), (size_t)(256 / (C >> 3)) * C * sizeof(float), (hipStream_t)stream, *a, HW,
                       C, out);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
__global__ __launch_bounds__(256) void k_pool_bwd(const float* __restrict__ gpool, int HW, int C, uint4* __restrict__ g,
                                                  int64_t total) {
    const int G = C >> 3;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cg = (int)(i % G);
        const int64_t n = i / ((int64_t)G * HW);
        const float* src = gpool + n * C + cg * 8;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = src[j] * inv;
        g[i] = pack8(f);
    }
}
extern "C" int mnas_pool_bwd(const float* gpool, int N, int HW, int C, void* g_bf16, void* stream) {
    if (!gpool || !g_bf16 || N < 1 || HW < 1 || C <= 0 || (C & 7)) return MNAS_EINVAL;
    const int64_t total = (int64_t)N * HW * (C >> 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pool_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gpool, HW, C, (uint4*)g_bf16, total);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ src, uint4* __restrict__ dst, int N,
                                                      int C, int HW) {
    const int G = C >> 3;
    const int64_t total = (int64_t)N * HW * G;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cg = (int)(i % G);
        const int64_t p = i / G;
        const int64_t n = p / HW, hw = p % HW;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = src[(n * C + cg * 8 + j) * HW + hw];
        dst[i] = pack8(f);
    }
}
extern "C" int mnas_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int N, int C, int HW, void* stream) {
    if (C <= 0 || (C & 7)) return MNAS_EINVAL;
    int64_t total = (int64_t)N * HW * (C >> 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (uint4*)dst, N, C, HW);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int64_t mnas_packed_bytes(int kind, int Co, int Ci, int kh, int kw) {
    if (kind == MNAS_PACK_FWD) return (int64_t)round_up(Co, 16) * round_up(kh * kw * Ci, 32) * 2;
    if (kind == MNAS_PACK_DGRAD) return (int64_t)round_up(Ci, 16) * round_up(kh * kw * Co, 32) * 2;
    if (kind == MNAS_PACK_DW) return (int64_t)kh * kw * Co * 4;
    if (kind == MNAS_PACK_TCONV) return (int64_t)round_up(4 * Ci, 16) * round_up(4 * Co, 32) * 2;
    return -1;
}

// dst[r][k] bf16, rows = round_up(R,16), cols = round_up(T*S,32); kind FWD: r=co, (t,s)=(tap,ci); DGRAD: r=ci, s=co
__global__ void k_pack_gemm(const float* __restrict__ w, int kind, int Co, int Ci, int taps, uint16_t* __restrict__ dst,
                            int rows_pad, int kpad) {
    const int total = rows_pad * kpad;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / kpad, k = i % kpad;
        const int S = (kind == MNAS_PACK_FWD) ? Ci : Co;
        const int R = (kind == MNAS_PACK_FWD) ? Co : Ci;
        float v = 0.f;
        if (r < R && k < taps * S) {
            const int tap = k / S, s = k % S;
            const int co = (kind == MNAS_PACK_FWD) ? r : s;
            const int ci = (kind == MNAS_PACK_FWD) ? s : r;
            v = w[((size_t)co * Ci + ci) * taps + tap];
        }
        dst[i] = f_to_bf(v);
    }
}
__global__ void k_pack_dw(const float* __restrict__ w, int C, int taps, float* __restrict__ dst) {
    const int total = C * taps;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int tap = i / C, c = i % C;
        dst[i] = w[(size_t)c * taps + tap];
    }
}
// MNAS_PACK_TCONV element (r, k) of the [round16(4Ci)][round32(4Co)] matrix: row = parity class (ph,pw) x ci, column = dy
// neighbour (dh,dw) x co; class ph = 0 only sees kh = 1 from dh = 0; ph = 1 sees kh = 2 from dh = 0 and kh = 0 from dh = 1.
__device__ __forceinline__ float tconv_elem(const float* __restrict__ w, int Co, int Ci, int r, int k) {
    if (r >= 4 * Ci || k >= 4 * Co) return 0.f;
    const int cls = r / Ci, ci = r - cls * Ci, nb = k / Co, co = k - nb * Co;
    const int ph = cls >> 1, pw = cls & 1, dh = nb >> 1, dw = nb & 1;
    const int kh = ph == 0 ? (dh == 0 ? 1 : -1) : (dh == 0 ? 2 : 0);
    const int kw = pw == 0 ? (dw == 0 ? 1 : -1) : (dw == 0 ? 2 : 0);
    return (kh >= 0 && kw >= 0) ? w[(((size_t)co * Ci + ci) * 3 + kh) * 3 + kw] : 0.f;
}
__global__ void k_pack_tconv(const float* __restrict__ w, int Co, int Ci, uint16_t* __restrict__ dst) {
    const int rows = (4 * Ci + 15) / 16 * 16, kpad = (4 * Co + 31) / 32 * 32;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * kpad; i += gridDim.x * 256) dst[i] = f_to_bf(tconv_elem(w, Co, Ci, i / kpad, i % kpad));
}
// All layers of a network in ONE launch: blockIdx.y selects the descriptor (device array), blockIdx.x strides over its
// elements.  A training step re-packs ~80 small tensors (the weights change with every optimizer step); as separate
// launches they cost ~4 us each at the head of the forward.
__global__ __launch_bounds__(256) void k_pack_batch(const MnasPackDesc* __restrict__ descs) {
    const MnasPackDesc d = descs[blockIdx.y];
    const int taps = d.taps;
    if (d.kind == MNAS_PACK_DW) {
        const int total = d.Co * taps;
        float* dst = (float*)d.dst;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            const int tap = i / d.Co, c = i % d.Co;
            dst[i] = d.w[(size_t)c * taps + tap];
        }
        return;
    }
    if (d.kind == MNAS_PACK_TCONV) {
        const int rows = (4 * d.Ci + 15) / 16 * 16, kpad = (4 * d.Co + 31) / 32 * 32;
        uint16_t* dst = (uint16_t*)d.dst;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * kpad; i += gridDim.x * 256)
            dst[i] = f_to_bf(tconv_elem(d.w, d.Co, d.Ci, i / kpad, i % kpad));
        return;
    }
    const int S = (d.kind == MNAS_PACK_FWD) ? d.Ci : d.Co;
    const int R = (d.kind == MNAS_PACK_FWD) ? d.Co : d.Ci;
    const int rows_pad = (R + 15) / 16 * 16, kpad = (taps * S + 31) / 32 * 32;
    const int total = rows_pad * kpad;
    uint16_t* dst = (uint16_t*)d.dst;
    // 4 gathers in flight per thread and pass (the largest tensors -- 192x320x3x3 in two layouts -- are ~70 dependent
    // passes per thread otherwise, which is most of this launch's 34 us)
    const int stride = gridDim.x * 256;
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * stride;
            v[j] = 0.f;
            if (i < total) {
                const int r = i / kpad, k = i % kpad;
                if (r < R && k < taps * S) {
                    const int tap = k / S, s = k % S;
                    const int co = (d.kind == MNAS_PACK_FWD) ? r : s;
                    const int ci = (d.kind == MNAS_PACK_FWD) ? s : r;
                    v[j] = d.w[((size_t)co * d.Ci + ci) * taps + tap];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * stride;
            if (i < total) dst[i] = f_to_bf(v[j]);
        }
    }
}
extern "C" int mnas_pack_weights_batch(const MnasPackDesc* descs_device, int n, void* stream) {
    if (!descs_device || n < 1 || n > 65535) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_pack_batch, dim3(128, n), dim3(256), 0, (hipStream_t)stream, descs_device);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_pack_weights(const float* w, int kind, int Co, int Ci, int kh, int kw, void* dst, void* stream) {
    const int taps = kh * kw;
    if (kind == MNAS_PACK_FWD || kind == MNAS_PACK_DGRAD) {
        const int R = (kind == MNAS_PACK_FWD) ? Co : Ci, S = (kind == MNAS_PACK_FWD) ? Ci : Co;
        const int rows_pad = round_up(R, 16), kpad = round_up(taps * S, 32);
        int blocks = (rows_pad * kpad + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_pack_gemm, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, kind, Co, Ci, taps,
                           (uint16_t*)dst, rows_pad, kpad);
    } else if (kind == MNAS_PACK_TCONV) {
        if (kh != 3 || kw != 3) return MNAS_EINVAL;
        hipLaunchKernelGGL(k_pack_tconv, dim3(32), dim3(256), 0, (hipStream_t)stream, w, Co, Ci, (uint16_t*)dst);
    } else if (kind == MNAS_PACK_DW) {
        int blocks = (Co * taps + 255) / 256;
        hipLaunchKernelGGL(k_pack_dw, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Co, taps, (float*)dst);
    } else {
        return MNAS_EINVAL;
    }
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ------------------------------------------------------------------------------------------------
// wgrad reductions (+ relayout to the reference's [Co][Ci][kh][kw])
// ------------------------------------------------------------------------------------------------
// 32 outputs x 8 split lanes per block, 16 independent loads per thread and pass.  Split ranges longer than FIN_SPLITS
// rows are reduced in two deterministic levels: level 1 (FOLD) cuts the range over grid.y and writes each chunk's sum
// back into the chunk's FIRST partial row (in place: a block only touches its own 32 columns); level 2 sums those rows
// (row stride FIN_SPLITS) and applies the relayout to the reference's weight layout.  No float atomics.
#define FIN_SPLITS 128
template <bool DW, bool FOLD>
__device__ __forceinline__ void wgrad_finalize_body(float* __restrict__ partial, int nsplit, int pstride, int Co, int Ci, int taps,
                                                    float* __restrict__ grad, int accumulate, int bx, int by) {
    __shared__ float red[8][33];
    const int K = taps * Ci;
    const int total = DW ? Co * taps : Co * K;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = bx * 32 + tx;
    const int p0 = FOLD ? by * FIN_SPLITS : 0, p1 = FOLD ? min(nsplit, p0 + FIN_SPLITS) : nsplit;
    float s = 0.f;
#if MNAS_FIN_HOIST
    // the destination (two divisions) and its old value are formed BEFORE the partial rows are read (see k_bn_fwd_finalize)
    float* d = nullptr;
    float dold = 0.f;
    if (!FOLD && ty == 0 && i < total) {
        if (DW) {          // wpartial rows are [taps][C]; reference layout [C][1][kh][kw]
            const int tap = i / Co, c = i % Co;
            d = grad + (size_t)c * taps + tap;
        } else {           // partial rows are [Co][taps*Ci]; reference layout [Co][Ci][kh][kw]
            const int co = i / K, k = i % K;
            const int tap = k / Ci, ci = k % Ci;
            d = grad + ((size_t)co * Ci + ci) * taps + tap;
        }
        if (accumulate) dold = *d;
    }
#endif
    if (i < total) {
        const float* src = partial + i;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        // 16 rows per lane and pass, all loads of a pass in flight together (predicated: no dependent tail loop)
        for (int p = p0 + ty; p < p1; p += 128) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int pj = p + 8 * j;
                v[j] = pj < p1 ? src[(size_t)pj * pstride * total] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += (v[j] + v[j + 4]) + (v[j + 8] + v[j + 12]);
        }
        s += (a[0] + a[1]) + (a[2] + a[3]);
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < total) {
#pragma unroll
        for (int j = 1; j < 8; ++j) s += red[j][tx];
        if (FOLD) {
            partial[(size_t)p0 * total + i] = s;
            return;
        }
#if MNAS_FIN_HOIST
        *d = dold + s;
#else
        float* d;
        if (DW) {          // wpartial rows are [taps][C]; reference layout [C][1][kh][kw]
            const int tap = i / Co, c = i % Co;
            d = grad + (size_t)c * taps + tap;
        } else {           // partial rows are [Co][taps*Ci]; reference layout [Co][Ci][kh][kw]
            const int co = i / K, k = i % K;
            const int tap = k / Ci, ci = k % Ci;
            d = grad + ((size_t)co * Ci + ci) * taps + tap;
        }
        *d = (accumulate ? *d : 0.f) + s;
#endif
    }
}
template <bool DW, bool FOLD>
__global__ __launch_bounds__(256) void k_wgrad_finalize(float* __restrict__ partial, int nsplit, int pstride, int Co, int Ci,
                                                         int taps, float* __restrict__ grad, int accumulate) {
    wgrad_finalize_body<DW, FOLD>(partial, nsplit, pstride, Co, Ci, taps, grad, accumulate, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// One launch for everything that sits between two dependent backward kernels: the BatchNorm-backward finalize of the NEXT
// layer and the weight-gradient reductions of the kernel that just ran (and the second level of the one before it).  As
// separate launches these were 2-3 tiny dependent kernels (5-9 us each + launch boundaries) in every gap of the main stream.
// Blocks [0, bn.nblk) run the BatchNorm part, the next w1.nx*w1.ny the first weight-gradient task, the rest the second.
// ------------------------------------------------------------------------------------------------
struct PostWg { float* partial; float* grad; int nsplit, Co, Ci, taps, dw, level, nx, ny; };
struct PostArgs {
    const float* bn_partial; float* bnbuf; float* dgamma; float* dbeta; double count;
    int bn_nparts, bn_C, bn_nblk, bn_wide;
    PostWg w1, w2;
};
__device__ __forceinline__ void post_wg(const PostWg& w, int b) {
    const int bx = b % w.nx, by = b / w.nx;
    // level 1: single-level finalize; 2: fold (first level of two); 3: second level over the folded rows (stride FIN_SPLITS)
    if (w.level == 2) {
        if (w.dw) wgrad_finalize_body<true, true>(w.partial, w.nsplit, 1, w.Co, w.Ci, w.taps, w.grad, 1, bx, by);
        else wgrad_finalize_body<false, true>(w.partial, w.nsplit, 1, w.Co, w.Ci, w.taps, w.grad, 1, bx, by);
    } else {
        const int ns = w.level == 3 ? (w.nsplit + FIN_SPLITS - 1) / FIN_SPLITS : w.nsplit;
        const int ps = w.level == 3 ? FIN_SPLITS : 1;
        if (w.dw) wgrad_finalize_body<true, false>(w.partial, ns, ps, w.Co, w.Ci, w.taps, w.grad, 1, bx, by);
        else wgrad_finalize_body<false, false>(w.partial, ns, ps, w.Co, w.Ci, w.taps, w.grad, 1, bx, by);
    }
}
__global__ __launch_bounds__(256) void k_bwd_post(PostArgs a) {
    int b = blockIdx.x;
    if (b < a.bn_nblk) {
        if (a.bn_wide) bn_bwd_finalize_body<256>(a.bn_partial, a.bn_nparts, a.bn_C, a.count, a.bnbuf, a.dgamma, a.dbeta, 1, b);
        else bn_bwd_finalize_body<64>(a.bn_partial, a.bn_nparts, a.bn_C, a.count, a.bnbuf, a.dgamma, a.dbeta, 1, b);
        return;
    }
    b -= a.bn_nblk;
    const int n1 = a.w1.level ? a.w1.nx * a.w1.ny : 0;
    if (b < n1) { post_wg(a.w1, b); return; }
    post_wg(a.w2, b - n1);
}
static bool post_fill(PostWg* w, const MnasPostWgrad& t) {
    w->level = t.level;
    if (!t.level) { w->nx = w->ny = 0; return true; }
    if (t.level < 1 || t.level > 3 || !t.partial || !t.grad || t.nsplit < 1 || t.Co < 1 || t.Ci < 1 || t.taps < 1) return false;
    if (t.level == 1 && t.nsplit > 2 * FIN_SPLITS) return false;          // single level only for short split ranges
    w->partial = t.partial; w->grad = t.grad; w->nsplit = t.nsplit; w->Co = t.Co; w->Ci = t.Ci; w->taps = t.taps; w->dw = t.dw;
    const int total = t.dw ? t.Co * t.taps : t.Co * t.Ci * t.taps;
    w->nx = (total + 31) / 32;
    w->ny = t.level == 2 ? (t.nsplit + FIN_SPLITS - 1) / FIN_SPLITS : 1;
    return true;
}
extern "C" int mnas_bwd_post(const MnasBwdPost* p, void* stream) {
    if (!p) return MNAS_EINVAL;
    PostArgs a = {};
    if (p->bn_C > 0) {
        if (!p->bn_partial || !p->bnbuf || p->bn_nparts < 1) return MNAS_EINVAL;
        a.bn_partial = p->bn_partial; a.bnbuf = p->bnbuf; a.dgamma = p->dgamma; a.dbeta = p->dbeta; a.count = p->count;
        a.bn_nparts = p->bn_nparts; a.bn_C = p->bn_C;
        a.bn_wide = p->bn_nparts > 256 ? 1 : 0;
        a.bn_nblk = a.bn_wide ? p->bn_C : (p->bn_C + 3) / 4;
    }
    if (!post_fill(&a.w1, p->w1) || !post_fill(&a.w2, p->w2)) return MNAS_EINVAL;
    const int blocks = a.bn_nblk + a.w1.nx * a.w1.ny + a.w2.nx * a.w2.ny;
    if (blocks < 1) return MNAS_OK;
    hipLaunchKernelGGL(k_bwd_post, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

template <bool DW>
static int launch_wgrad_finalize(float* partial, int nsplit, int Co, int Ci, int taps, float* grad, int accumulate,
                                 void* stream) {
    if (!partial || !grad || nsplit < 1 || Co < 1 || Ci < 1 || taps < 1) return MNAS_EINVAL;
    const int total = DW ? Co * taps : Co * Ci * taps;
    const int nx = (total + 31) / 32;
    if (nsplit > 2 * FIN_SPLITS) {
        const int ny = (nsplit + FIN_SPLITS - 1) / FIN_SPLITS;
        hipLaunchKernelGGL((k_wgrad_finalize<DW, true>), dim3(nx, ny), dim3(256), 0, (hipStream_t)stream, partial, nsplit, 1,
                           Co, Ci, taps, grad, accumulate);
        hipLaunchKernelGGL((k_wgrad_finalize<DW, false>), dim3(nx, 1), dim3(256), 0, (hipStream_t)stream, partial, ny,
                           FIN_SPLITS, Co, Ci, taps, grad, accumulate);
    } else {
        hipLaunchKernelGGL((k_wgrad_finalize<DW, false>), dim3(nx, 1), dim3(256), 0, (hipStream_t)stream, partial, nsplit, 1,
                           Co, Ci, taps, grad, accumulate);
    }
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
extern "C" int mnas_wgrad_finalize(float* partial, int nsplit, int Co, int Ci, int taps, float* grad,
                                   int accumulate, void* stream) {
    return launch_wgrad_finalize<false>(partial, nsplit, Co, Ci, taps, grad, accumulate, stream);
}
extern "C" int mnas_dw_wgrad_finalize(float* wpartial, int nparts, int C, int k, float* grad, int accumulate,
                                      void* stream) {
    return launch_wgrad_finalize<true>(wpartial, nparts, C, 1, k * k, grad, accumulate, stream);
}

// ------------------------------------------------------------------------------------------------
// fused Adam over a flat buffer (torch.optim.Adam semantics, train.py:219-221)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                              float wd, float bc1, float bc2_sqrt, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}
// torch.optim.RMSprop (train.py:222-224; centered = False) and torch.optim.SGD (train.py:226-228) over the same flat buffers
__global__ __launch_bounds__(256) void k_rmsprop(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                 float* __restrict__ buf, int64_t n, float lr, float alpha, float eps, float wd,
                                                 float momentum, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float s = fmaf(alpha, sq[i], (1.f - alpha) * gi * gi);
        sq[i] = s;
        const float avg = sqrtf(s) + eps;
        if (momentum > 0.f) {
            const float b = fmaf(momentum, buf[i], gi / avg);
            buf[i] = b;
            p[i] = pi - lr * b;
        } else {
            p[i] = pi - lr * (gi / avg);
        }
    }
}
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int64_t n,
                                             float lr, float momentum, float dampening, float wd, int nesterov, int first,
                                             float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        if (momentum != 0.f) {
            const float b = first ? gi : fmaf(momentum, buf[i], (1.f - dampening) * gi);
            buf[i] = b;
            gi = nesterov ? fmaf(momentum, b, gi) : b;
        }
        p[i] = pi - lr * gi;
    }
}
static unsigned opt_blocks(int64_t n) {
    int64_t blocks = (n + 255) / 256;
    return (unsigned)(blocks > 2048 ? 2048 : blocks);
}
extern "C" int mnas_rmsprop_step(float* p, const float* g, float* square_avg, float* momentum_buf, int64_t n, float lr, float alpha,
                                 float eps, float weight_decay, float momentum, float grad_scale, void* stream) {
    if (n < 0 || !p || !g || !square_avg || (momentum > 0.f && !momentum_buf)) return MNAS_EINVAL;
    if (n == 0) return MNAS_OK;
    hipLaunchKernelGGL(k_rmsprop, dim3(opt_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, square_avg, momentum_buf, n, lr, alpha,
                       eps, weight_decay, momentum, grad_scale);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
extern "C" int mnas_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                             float weight_decay, int nesterov, int step, float grad_scale, void* stream) {
    if (n < 0 || step < 1 || !p || !g || (momentum != 0.f && !momentum_buf)) return MNAS_EINVAL;
    if (n == 0) return MNAS_OK;
    hipLaunchKernelGGL(k_sgd, dim3(opt_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, momentum_buf, n, lr, momentum, dampening,
                       weight_decay, nesterov, step == 1 ? 1 : 0, grad_scale);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

extern "C" int mnas_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    if (step < 1 || n < 0) return MNAS_EINVAL;
    if (n == 0) return MNAS_OK;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                       eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
