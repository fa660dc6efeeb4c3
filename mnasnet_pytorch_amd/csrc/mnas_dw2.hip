// Depthwise k x k convolution (k in {3,5}, pad k/2) with STRIDE 2, NHWC bf16, fp32 math: forward, input gradient, weight gradient.
// Replaces ATen's grouped conv2d fwd/bwd for the depthwise ConvBlock of SepConv(reduce=True) (mnasnet.py:73-81: stride = 2 if reduce).
// Mnasnet itself never builds that form (mnasnet.py:180: SepConv(32, 16, 3)), so these are plain direct kernels -- correct,
// deterministic, coalesced, not tuned: thread = (pixel lane, channel pair), consecutive lanes = consecutive channel pairs (4-byte
// loads / stores of one pixel's channels are contiguous), a grid-stride walk over the pixels.  Same rounding points as the stride-1
// sweeps of mnas_dw.hip (act-on-read and dy-on-read in fp32, results rounded to bf16 once), so oracle/bf16_mirror.py needs no
// special case.  Entry points: mnas_dw_fwd / mnas_dw_bwd with stride == 2 (include/mnas.h).
#include "mnas_common.h"

struct Dw2Args {
    int N, H, W, C, k, Ho, Wo;
    int cps;        // channel pairs
    int cpb;        // channel pairs per block (<= 256, whole block when cps <= 256)
    int pl;         // pixel lanes per block = 256 / cpb
};
static bool dw2_plan(int N, int H, int W, int C, int k, Dw2Args* a) {
    if (N < 1 || H < 1 || W < 1 || C < 8 || (C & 7) || (k != 3 && k != 5)) return false;
    a->N = N; a->H = H; a->W = W; a->C = C; a->k = k;
    a->Ho = (H + 2 * (k / 2) - k) / 2 + 1; a->Wo = (W + 2 * (k / 2) - k) / 2 + 1;
    a->cps = C / 2;
    a->cpb = a->cps < 256 ? a->cps : 256;
    a->pl = 256 / a->cpb;
    return a->Ho >= 1 && a->Wo >= 1 && (long long)N * H * W * C < 0x7fffffffLL;
}
static int dw2_cblocks(const Dw2Args& a) { return (a.cps + a.cpb - 1) / a.cpb; }

// ---- forward: out[n,oy,ox,c] = bias[c] + sum w[ky,kx,c] * act(x)[n, 2oy+ky-p, 2ox+kx-p, c]; stats float[2][C][P] (P = gridDim.x)
template <int KS>
__global__ __launch_bounds__(256) void k_dw2_fwd(Dw2Args a, MnasActIn in, const float* __restrict__ w, const float* __restrict__ bias,
                                                 uint32_t* __restrict__ out, float* __restrict__ stats) {
    __shared__ float red[2][256][2];
    constexpr int PAD = KS / 2;
    const int cpl = threadIdx.x % a.cpb, pln = threadIdx.x / a.cpb;
    const int cp = blockIdx.y * a.cpb + cpl;
    const bool ok = pln < a.pl && cp < a.cps;
    const int ch = 2 * cp;
    const bool has_coef = in.scale != nullptr;
    mnas_f2 wt[KS * KS], b2 = {0.f, 0.f}, cs = {1.f, 1.f}, ct = {0.f, 0.f}, s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
    if (ok) {
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) wt[t] = mnas_ld2(w + (size_t)t * a.C + ch);
        if (bias) b2 = mnas_ld2(bias + ch);
        if (has_coef) { cs = mnas_ld2(in.scale + ch); ct = mnas_ld2(in.shift + ch); }
    }
    const uint32_t* x = (const uint32_t*)in.data;
    const int Mo = a.N * a.Ho * a.Wo;
    if (ok) {
        for (int m = blockIdx.x * a.pl + pln; m < Mo; m += gridDim.x * a.pl) {
            const int n = m / (a.Ho * a.Wo), r = m - n * a.Ho * a.Wo, oy = r / a.Wo, ox = r - oy * a.Wo;
            mnas_f2 acc = b2;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = 2 * oy + ky - PAD;
                if (iy < 0 || iy >= a.H) continue;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int ix = 2 * ox + kx - PAD;
                    if (ix < 0 || ix >= a.W) continue;
                    mnas_f2 v = mnas_bf2(x[(((size_t)n * a.H + iy) * a.W + ix) * a.cps + cp]);
                    if (has_coef) { v = mnas_f2fma(v, cs, ct); v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
                    acc = mnas_f2fma(wt[ky * KS + kx], v, acc);
                }
            }
            s1 += acc;
            s2 = mnas_f2fma(acc, acc, s2);
            out[(size_t)m * a.cps + cp] = pack_bf16(acc.x, acc.y);
        }
    }
    if (stats) {        // pixel lanes of one channel pair added in lane order: deterministic, every (channel, column) written once
        red[0][threadIdx.x][0] = s1.x; red[0][threadIdx.x][1] = s1.y; red[1][threadIdx.x][0] = s2.x; red[1][threadIdx.x][1] = s2.y;
        __syncthreads();
        if (pln == 0 && cp < a.cps) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v0 = 0.f, v1 = 0.f;
                for (int p = 0; p < a.pl; ++p) { v0 += red[q][p * a.cpb + cpl][0]; v1 += red[q][p * a.cpb + cpl][1]; }
                stats[((size_t)q * a.C + ch) * gridDim.x + blockIdx.x] = v0;
                stats[((size_t)q * a.C + ch + 1) * gridDim.x + blockIdx.x] = v1;
            }
        }
    }
}

// dy = c1*(g*[s*y+t>0]) + c2*y + c3 of one channel pair (fp32, as the stride-1 sweeps form it on read)
__device__ __forceinline__ mnas_f2 dw2_dy(uint32_t gu, uint32_t yu, const mnas_f2 (&cf)[5]) {
    const mnas_f2 g = mnas_bf2(gu), y = mnas_bf2(yu);
    const mnas_f2 z = mnas_f2fma(y, cf[0], cf[1]);
    mnas_f2 dz;
    dz.x = (z.x > 0.f) ? g.x : 0.f;
    dz.y = (z.y > 0.f) ? g.y : 0.f;
    return mnas_f2fma(cf[2], dz, mnas_f2fma(cf[3], y, cf[4]));
}

// ---- input gradient: gin[n,iy,ix,c] = sum_{ky,kx: (iy+p-ky), (ix+p-kx) even} w[ky,kx,c] * dy[n,(iy+p-ky)/2,(ix+p-kx)/2,c]
template <int KS>
__global__ __launch_bounds__(256) void k_dw2_dgrad(Dw2Args a, MnasGradIn d, const float* __restrict__ w, uint32_t* __restrict__ gin) {
    constexpr int PAD = KS / 2;
    const int cpl = threadIdx.x % a.cpb, pln = threadIdx.x / a.cpb;
    const int cp = blockIdx.y * a.cpb + cpl;
    if (!(pln < a.pl && cp < a.cps)) return;
    const int ch = 2 * cp;
    mnas_f2 wt[KS * KS], cf[5];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) wt[t] = mnas_ld2(w + (size_t)t * a.C + ch);
#pragma unroll
    for (int r = 0; r < 5; ++r) cf[r] = mnas_ld2(d.coef + (size_t)r * a.C + ch);
    const uint32_t *g = (const uint32_t*)d.g, *y = (const uint32_t*)d.y;
    const int Mi = a.N * a.H * a.W;
    for (int m = blockIdx.x * a.pl + pln; m < Mi; m += gridDim.x * a.pl) {
        const int n = m / (a.H * a.W), r = m - n * a.H * a.W, iy = r / a.W, ix = r - iy * a.W;
        mnas_f2 acc = {0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int ty = iy + PAD - ky;
            if (ty < 0 || (ty & 1) || (ty >> 1) >= a.Ho) continue;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int tx = ix + PAD - kx;
                if (tx < 0 || (tx & 1) || (tx >> 1) >= a.Wo) continue;
                const size_t o = (((size_t)n * a.Ho + (ty >> 1)) * a.Wo + (tx >> 1)) * a.cps + cp;
                acc = mnas_f2fma(wt[ky * KS + kx], dw2_dy(g[o], y[o], cf), acc);
            }
        }
        gin[(size_t)m * a.cps + cp] = pack_bf16(acc.x, acc.y);
    }
}

// ---- weight gradient: wpartial[row][tap][c] = sum over this block's output pixels of dy * act(x)[2oy+ky-p, 2ox+kx-p]
template <int KS>
__global__ __launch_bounds__(256) void k_dw2_wgrad(Dw2Args a, MnasActIn x, MnasGradIn d, float* __restrict__ wpartial) {
    extern __shared__ float red2[];          // [256][2] per tap pass
    constexpr int PAD = KS / 2;
    const int cpl = threadIdx.x % a.cpb, pln = threadIdx.x / a.cpb;
    const int cp = blockIdx.y * a.cpb + cpl;
    const bool ok = pln < a.pl && cp < a.cps;
    const int ch = 2 * cp;
    const bool has_coef = x.scale != nullptr;
    mnas_f2 acc[KS * KS], cf[5], cs = {1.f, 1.f}, ct = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) acc[t] = mnas_f2{0.f, 0.f};
    if (ok) {
#pragma unroll
        for (int r = 0; r < 5; ++r) cf[r] = mnas_ld2(d.coef + (size_t)r * a.C + ch);
        if (has_coef) { cs = mnas_ld2(x.scale + ch); ct = mnas_ld2(x.shift + ch); }
        const uint32_t *g = (const uint32_t*)d.g, *y = (const uint32_t*)d.y, *xi = (const uint32_t*)x.data;
        const int Mo = a.N * a.Ho * a.Wo;
        for (int m = blockIdx.x * a.pl + pln; m < Mo; m += gridDim.x * a.pl) {
            const int n = m / (a.Ho * a.Wo), r = m - n * a.Ho * a.Wo, oy = r / a.Wo, ox = r - oy * a.Wo;
            const size_t o = (size_t)m * a.cps + cp;
            const mnas_f2 dy = dw2_dy(g[o], y[o], cf);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = 2 * oy + ky - PAD;
                if (iy < 0 || iy >= a.H) continue;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int ix = 2 * ox + kx - PAD;
                    if (ix < 0 || ix >= a.W) continue;
                    mnas_f2 v = mnas_bf2(xi[(((size_t)n * a.H + iy) * a.W + ix) * a.cps + cp]);
                    if (has_coef) { v = mnas_f2fma(v, cs, ct); v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
                    acc[ky * KS + kx] = mnas_f2fma(dy, v, acc[ky * KS + kx]);
                }
            }
        }
    }
    // pixel lanes of a channel pair summed in lane order, tap by tap
    for (int t = 0; t < KS * KS; ++t) {
        __syncthreads();
        red2[threadIdx.x * 2] = acc[t].x; red2[threadIdx.x * 2 + 1] = acc[t].y;
        __syncthreads();
        if (pln == 0 && cp < a.cps) {
            float v0 = 0.f, v1 = 0.f;
            for (int p = 0; p < a.pl; ++p) { v0 += red2[(p * a.cpb + cpl) * 2]; v1 += red2[(p * a.cpb + cpl) * 2 + 1]; }
            float* dst = wpartial + ((size_t)blockIdx.x * KS * KS + t) * a.C + ch;
            dst[0] = v0; dst[1] = v1;
        }
    }
}

// ---- host side (called by mnas_dw_fwd / mnas_dw_bwd / mnas_dw_rows when stride == 2) ---------------------------------------------
static int dw2_grid(const Dw2Args& a, int nparts, int pixels) {
    int g = (pixels + a.pl - 1) / a.pl;
    if (g > nparts) g = nparts;
    return g < 1 ? 1 : g;
}
// rows / columns of the partial tables: which = 0 forward statistics float[2][C][rows]; which = 3 wpartial float[rows][k*k][C]
int mnas_dw2_rows(int N, int H, int W, int C, int k, int nparts, int which) {
    Dw2Args a;
    if (!dw2_plan(N, H, W, C, k, &a) || nparts < 1 || (which != 0 && which != 3)) return -1;
    return dw2_grid(a, nparts, N * a.Ho * a.Wo);
}
int mnas_dw2_fwd(const MnasDwFwd* c, void* stream) {
    Dw2Args a;
    if (!dw2_plan(c->N, c->H, c->W, c->C, c->k, &a) || !c->in.data || !c->w || !c->out) return MNAS_EINVAL;
    const dim3 grid(dw2_grid(a, c->nparts, a.N * a.Ho * a.Wo), dw2_cblocks(a));
    hipStream_t s = (hipStream_t)stream;
    if (c->k == 3) hipLaunchKernelGGL(k_dw2_fwd<3>, grid, dim3(256), 0, s, a, c->in, c->w, c->bias, (uint32_t*)c->out, c->stats);
    else hipLaunchKernelGGL(k_dw2_fwd<5>, grid, dim3(256), 0, s, a, c->in, c->w, c->bias, (uint32_t*)c->out, c->stats);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
int mnas_dw2_bwd(const MnasDwBwd* c, void* stream) {
    Dw2Args a;
    if (!dw2_plan(c->N, c->H, c->W, c->C, c->k, &a) || !c->x.data || !c->dy.g || !c->dy.y || !c->dy.coef || !c->w) return MNAS_EINVAL;
    if (c->phase < 1 || c->phase > 2 || c->g_masked || c->red_partial) return MNAS_EINVAL;   // two plain launches, no fused reduce
    hipStream_t s = (hipStream_t)stream;
    if (c->phase == 1) {
        if (!c->gin) return MNAS_EINVAL;
        const dim3 grid(dw2_grid(a, c->nparts, a.N * a.H * a.W), dw2_cblocks(a));
        if (c->k == 3) hipLaunchKernelGGL(k_dw2_dgrad<3>, grid, dim3(256), 0, s, a, c->dy, c->w, (uint32_t*)c->gin);
        else hipLaunchKernelGGL(k_dw2_dgrad<5>, grid, dim3(256), 0, s, a, c->dy, c->w, (uint32_t*)c->gin);
    } else {
        if (!c->wpartial) return MNAS_EINVAL;
        const dim3 grid(dw2_grid(a, c->nparts, a.N * a.Ho * a.Wo), dw2_cblocks(a));
        if (c->k == 3) hipLaunchKernelGGL(k_dw2_wgrad<3>, grid, dim3(256), 256 * 2 * sizeof(float), s, a, c->x, c->dy, c->wpartial);
        else hipLaunchKernelGGL(k_dw2_wgrad<5>, grid, dim3(256), 256 * 2 * sizeof(float), s, a, c->x, c->dy, c->wpartial);
    }
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
