// BatchNorm statistics of a 1x1 convolution WITHOUT computing its output: the expand conv of MBConv_block
// (mnasnet.py:116-121) is linear in its input a = act(x) (C channels), y = W a + b, so over the batch
//     mean(y_c) = w_c . mean(a) + b_c          var(y_c) = w_c^T Cov(a) w_c,   Cov(a) = E[a a^T] - mean(a) mean(a)^T.
// k_gram accumulates the C x C second-moment matrix G = sum_pix a a^T and the column sums S = sum_pix a on the matrix
// cores (a pass over the SMALL tensor: C channels per pixel instead of t*C); k_gram_reduce / k_gram_bn turn them into the
// (scale, shift) of the expand conv's BatchNorm2d (mnasnet.py:55,60; ATen native_batch_norm statistics step), exactly as
// mnas_bn_fwd_finalize does from (sum y, sum y^2).  With the statistics known up front, the fused expand+depthwise kernel
// (mnas_dw.hip, EXP forms) never has to materialise the expanded tensor in HBM.
//
// k_gram: GEMM D[ci][cj] with the reduction over pixels; both operands are the same pixel-major tile, read with the LDS
// transpose read (ds_read_b64_tr_b16) as in mnas_wgrad.hip.  A workgroup owns a 64 x 64 slab of G and a contiguous pixel
// range; its 4 waves split every 128-pixel chunk and are combined through LDS in wave order (deterministic); slabs of the
// pixel splits go to gpart[split][C][C] / spart[split][C].
#include "mnas_common.h"

typedef __attribute__((ext_vector_type(4))) short gr_s4_t;
typedef __attribute__((address_space(3))) gr_s4_t* gr_lds_s4_ptr;

struct GramArgs {
    int M, C, chunk;          // pixels, channels, pixels per split (multiple of 128)
    MnasActIn x;
    float* gpart;             // [nsplit][C][C]
    float* spart;             // [nsplit][C]
};

__device__ __forceinline__ bf16x8_t gr_tr_frag(const uint16_t* tile, int ld, int row0, int col0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const uint16_t* p = tile + (row0 + g * 8 + (i >> 2)) * ld + col0 + (i & 3) * 4;
    const gr_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gr_lds_s4_ptr)p);
    const gr_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gr_lds_s4_ptr)(p + 4 * ld));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 2) void k_gram(GramArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BPX = 128, LD = 72;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.z * 64;
    const int rn = min(64, a.C - r0), cn = min(64, a.C - c0);         // valid rows / columns of the slab (multiples of 8)
    const int rtn = (rn + 15) >> 4, ctn = (cn + 15) >> 4;
    const bool same = r0 == c0;                                        // diagonal slab: one tile serves both operands
    float* lds_cx = (float*)smem;                                      // [2][2][64]  scale/shift for the row / column range
    uint16_t* tile_r = (uint16_t*)(lds_cx + 4 * 64);                   // [128][LD]   channels r0..r0+63
    uint16_t* tile_c = tile_r + BPX * LD;                              // [128][LD]   channels c0..c0+63
    float* lds_out = (float*)tile_r;                                   // reused at the end: [64][65] (+ [64] sums)

    const bool hasx = a.x.scale != nullptr;
    for (int i = tid; i < 4 * 64; i += 256) {
        const int which = i >> 7, r = (i >> 6) & 1, c = (which ? c0 : r0) + (i & 63);
        lds_cx[i] = (hasx && c < a.C) ? (r == 0 ? a.x.scale[c] : a.x.shift[c]) : 0.f;
    }
    for (int i = tid; i < 2 * BPX * LD / 8; i += 256) ((uint4*)tile_r)[i] = make_uint4(0, 0, 0, 0);

    const int cwr = rn >> 3, cwc = cn >> 3;
    int pr[4], kr[4], pc[4], kcn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        pr[i] = q / cwr; kr[i] = q - pr[i] * cwr;
        if (pr[i] >= BPX) pr[i] = -1;
        pc[i] = q / cwc; kcn[i] = q - pc[i] * cwc;
        if (pc[i] >= BPX || same) pc[i] = -1;
    }
    f32x4_t acc[4][4], acc1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc1[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3f80;              // bf16 1.0

    const int p_begin = blockIdx.x * a.chunk, p_end = min(a.M, p_begin + a.chunk);
    uint4 vr[4], vc[4];
    auto issue = [&](int pc0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vr[i] = make_uint4(0, 0, 0, 0); vc[i] = make_uint4(0, 0, 0, 0);
            if (pr[i] >= 0 && pc0 + pr[i] < p_end)
                vr[i] = *(const uint4*)((const uint16_t*)a.x.data + (size_t)(pc0 + pr[i]) * a.C + r0 + kr[i] * 8);
            if (pc[i] >= 0 && pc0 + pc[i] < p_end)
                vc[i] = *(const uint4*)((const uint16_t*)a.x.data + (size_t)(pc0 + pc[i]) * a.C + c0 + kcn[i] * 8);
        }
    };
    auto xform = [&](uint4 v, int which, int k8, bool ok) -> uint4 {
        if (!ok) return make_uint4(0, 0, 0, 0);
        if (!hasx) return v;
        float s[8], t[8];
        const float* cx = lds_cx + which * 128;
        *(float4*)&s[0] = *(const float4*)(cx + k8 * 8);
        *(float4*)&s[4] = *(const float4*)(cx + k8 * 8 + 4);
        *(float4*)&t[0] = *(const float4*)(cx + 64 + k8 * 8);
        *(float4*)&t[4] = *(const float4*)(cx + 64 + k8 * 8 + 4);
        return act8(v, s, t);
    };
    if (p_begin < p_end) issue(p_begin);
    for (int pc0 = p_begin; pc0 < p_end; pc0 += BPX) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (pr[i] >= 0) *(uint4*)(tile_r + pr[i] * LD + kr[i] * 8) = xform(vr[i], 0, kr[i], pc0 + pr[i] < p_end);
            if (pc[i] >= 0) *(uint4*)(tile_c + pc[i] * LD + kcn[i] * 8) = xform(vc[i], 1, kcn[i], pc0 + pc[i] < p_end);
        }
        __syncthreads();
        if (pc0 + BPX < p_end) issue(pc0 + BPX);
        const uint16_t* tc = same ? tile_r : tile_c;
        bf16x8_t bf[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            if (ct < ctn) bf[ct] = gr_tr_frag(tc, LD, wave * 32, ct * 16, lane);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if (rt >= rtn) continue;
            const bf16x8_t af = gr_tr_frag(tile_r, LD, wave * 32, rt * 16, lane);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                if (ct < ctn) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[ct], acc[rt][ct], 0, 0, 0);
            if (blockIdx.z == 0) acc1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, ones, acc1[rt], 0, 0, 0);
        }
    }
    // combine the 4 waves through LDS in wave order (deterministic); column 64 of the scratch holds the channel sums
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave != w) continue;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if (rt >= rtn) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + lg * 4 + r;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    if (ct >= ctn) continue;
                    float* d = &lds_out[row * 66 + ct * 16 + l15];
                    *d = (w == 0 ? 0.f : *d) + acc[rt][ct][r];
                }
                if (l15 == 0) {
                    float* d = &lds_out[row * 66 + 64];
                    *d = (w == 0 ? 0.f : *d) + acc1[rt][r];
                }
            }
        }
    }
    __syncthreads();
    float* gdst = a.gpart + (size_t)blockIdx.x * a.C * a.C;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        if (r < rn && c < cn) gdst[(size_t)(r0 + r) * a.C + c0 + c] = lds_out[r * 66 + c];
    }
    if (blockIdx.z == 0)
        for (int r = tid; r < rn; r += 256) a.spart[(size_t)blockIdx.x * a.C + r0 + r] = lds_out[r * 66 + 64];
}

extern "C" int mnas_gram(const MnasActIn* x, int64_t M, int C, int nsplit, float* gpart, float* spart, void* stream) {
    if (!x || !x->data || M < 1 || M > 0x7fffffff || C < 8 || (C & 7) || nsplit < 1 || !gpart || !spart) return MNAS_EINVAL;
    GramArgs a;
    a.M = (int)M; a.C = C;
    const int per = (int)((M + nsplit - 1) / nsplit);
    a.chunk = (per + 127) / 128 * 128;
    a.x = *x; a.gpart = gpart; a.spart = spart;
    const int nb = (C + 63) / 64;
    const size_t lds = (size_t)4 * 64 * sizeof(float) + (size_t)2 * 128 * 72 * 2;
    hipLaunchKernelGGL(k_gram, dim3(nsplit, nb, nb), dim3(256), lds, (hipStream_t)stream, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ---- (G, S) partials -> BatchNorm coefficients of y = W a + b ----------------------------------------------------------
// level 1: gsum[i] = sum over splits (fp64) of the C*C + C partial entries
__global__ __launch_bounds__(256) void k_gram_reduce(const float* __restrict__ gpart, const float* __restrict__ spart, int nsplit,
                                                     int C, double* __restrict__ gsum) {
    const int total = C * C + C;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    double s = 0.0;
    if (i < C * C) { for (int p = 0; p < nsplit; ++p) s += (double)gpart[(size_t)p * C * C + i]; }
    else { for (int p = 0; p < nsplit; ++p) s += (double)spart[(size_t)p * C + (i - C * C)]; }
    gsum[i] = s;
}
// level 2: one wave per output channel: mean = w.mu + b, var = w^T Cov w with the bf16-ROUNDED weights the conv kernels use
__global__ __launch_bounds__(256) void k_gram_bn(const double* __restrict__ gsum, int C, int Co, double count,
                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                                                 float* bnbuf) {
    const int lane = threadIdx.x & 63;
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (co < Co) {
        const float* wr = w + (size_t)co * C;
        double m = 0.0, q = 0.0;
        for (int i = lane; i < C; i += 64) {
            const double wi = (double)bf_to_f(f_to_bf(wr[i]));
            m += wi * gsum[C * C + i];
            double row = 0.0;
            for (int j = 0; j < C; ++j) row += (double)bf_to_f(f_to_bf(wr[j])) * gsum[(size_t)i * C + j];
            q += wi * row;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m += __shfl_xor(m, o, 64); q += __shfl_xor(q, o, 64); }
        if (lane == 0) {
            const double mean_lin = m / count;                       // w . mean(a)
            double var = q / count - mean_lin * mean_lin;            // w^T (G/n - mu mu^T) w
            if (var < 0.0) var = 0.0;
            const double mean = mean_lin + (bias ? (double)bias[co] : 0.0);
            const double invstd = 1.0 / sqrt(var + (double)eps);
            bnbuf[0 * Co + co] = (float)((double)gamma[co] * invstd);
            bnbuf[1 * Co + co] = (float)((double)beta[co] - mean * (double)gamma[co] * invstd);
            bnbuf[5 * Co + co] = (float)mean;
            bnbuf[6 * Co + co] = (float)invstd;
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[co] = (float)((1.0 - momentum) * (double)running_mean[co] + momentum * mean);
            running_var[co] = (float)((1.0 - momentum) * (double)running_var[co] + momentum * unbiased);
        }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

extern "C" int mnas_gram_bn_finalize(const float* gpart, const float* spart, int nsplit, int C, int Co, double count,
                                     const float* w, const float* bias, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                     float eps, double* scratch, float* bnbuf, void* stream) {
    if (!gpart || !spart || nsplit < 1 || C < 1 || Co < 1 || !w || !gamma || !beta || !running_mean || !running_var ||
        !scratch || !bnbuf) return MNAS_EINVAL;
    const int total = C * C + C;
    hipLaunchKernelGGL(k_gram_reduce, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, gpart, spart, nsplit, C, scratch);
    hipLaunchKernelGGL(k_gram_bn, dim3((Co + 3) / 4), dim3(256), 0, (hipStream_t)stream, scratch, C, Co, count, w, bias, gamma,
                       beta, running_mean, running_var, num_batches_tracked, momentum, eps, bnbuf);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
