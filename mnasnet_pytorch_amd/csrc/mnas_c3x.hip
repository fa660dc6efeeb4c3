// Dense 3x3 stride-2 forward of the first stage transitions (16 -> 24 at 112x112 -> 56x56, 24 -> 40 at 56x56 -> 28x28; ConvBlock of
// mnasnet.py:48-62 as used at :157-158) as a weight-stationary, barrier-free kernel.  Same contract as mnas_conv_gemm mode 0 (raw
// bf16 output, per-workgroup BatchNorm partial statistics, producer's BatchNorm+ReLU applied on load); behind mnas_conv_gemm, in
// front of k_igemm's im2col staging.
//
// k_igemm stages a 128-pixel im2col tile global -> registers -> LDS per K chunk, with a barrier pair each: 65 us for 141 MB at
// 112x112 (2.2 TB/s), 42 us at 56x56.  The weights here are tiny (7-17 KB), so:
//   * every WAVE keeps the whole [Co][9*Ci] weight matrix as MFMA A fragments in registers (40-84 VGPRs) and works on its own
//     16-pixel groups -- nothing is shared between waves, there is no LDS tile and no barrier in the loop;
//   * a group's B fragments are gathered straight from global memory, one group ahead: lane (pixel, k-chunk) reads the 16 bytes
//     = 8 consecutive input channels of tap (kh, kw) of its output pixel (k = tap*Ci + ci, the packed-weight order); neighbouring
//     output pixels re-read 2.25x of the input from L1 / L2; BatchNorm+ReLU in registers, out-of-image taps stay zero;
//   * the [16 pixels][Co] result leaves through a wave-private LDS stage as 16-byte stores; statistics in registers across all
//     groups, the four waves of a workgroup combined once at the end.
// Roofline: HBM (input read once from HBM, output written once).
#include "mnas_common.h"

struct C3xArgs {
    int N, Hi, Wi, Ci, Ho, Wo, Co;
    int stride;
    int M;                   // output pixels
    int Kpad;                // 9*Ci rounded up to 32
    int co_pad16;
    float rcp_hw, rcp_wo;
    MnasActIn act;
    const uint16_t* w;       // MNAS_PACK_FWD: [co_pad16][Kpad], k = tap*Ci + ci
    const float* bias;
    void* out;
    float* stats;            // [2][Co][gridDim.x] or NULL
};

__device__ __forceinline__ int c3x_fdiv(int n, int d, float rcp) {
    if (rcp == 0.f) return n / d;
    int q = (int)((float)n * rcp);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

template <int TPW, int KS>
__global__ __launch_bounds__(256) void k_c3x(C3xArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WC = TPW * 16, SP = WC + 8;
    float* lds_coef = (float*)smem;                                // [2][Ci] act-on-load scale / shift
    float* lds_bias = lds_coef + 2 * a.Ci;                         // [WC]
    uint16_t* stage = (uint16_t*)(lds_bias + WC);                  // [4][16][SP]
    float* lds_fin = (float*)(stage + 4 * 16 * SP);                // end of kernel: [4][2][WC]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const bool has_coef = a.act.scale != nullptr;

    for (int i = tid; i < 2 * a.Ci; i += 256) lds_coef[i] = has_coef ? (i < a.Ci ? a.act.scale[i] : a.act.shift[i - a.Ci]) : 0.f;
    for (int i = tid; i < WC; i += 256) lds_bias[i] = (a.bias && i < a.Co) ? a.bias[i] : 0.f;
    // ---- the whole weight matrix: A fragments [cout l15][k = ks*32 + lg*8 ..]; this lane's tap / channel per k-step
    bf16x8_t wf[TPW][KS];
    int koff[KS];                                                  // (kh*Wi + kw)*Ci + ci, -1 past 9*Ci
    unsigned tbits = 0;                                            // 4 bits per k-step: kh (2), kw (2)
    int cik[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 32 + lg * 8;
        const int tap = k / a.Ci, ci = k - tap * a.Ci;
        const bool kok = tap < 9;
        const int kh = kok ? tap / 3 : 0, kw = kok ? tap - kh * 3 : 0;
        koff[ks] = kok ? (kh * a.Wi + kw) * a.Ci + ci : -1;
        cik[ks] = ci;
        tbits |= (unsigned)(kh | (kw << 2)) << (4 * ks);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int row = t * 16 + l15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < a.co_pad16 && k < a.Kpad) v = *(const uint4*)(a.w + (size_t)row * a.Kpad + k);
            wf[t][ks] = *(const bf16x8_t*)&v;
        }
    }
    float s1[TPW][4], s2[TPW][4];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
    __syncthreads();                                               // coefficient / bias tables visible

    uint16_t* st = stage + wave * 16 * SP;
    const int hw = a.Ho * a.Wo;
    const int ngroups = (a.M + 15) >> 4;
    const int gstride = gridDim.x * 4, g0 = blockIdx.x * 4 + wave;
    const int cpp = a.Co >> 3;                                     // 16-byte chunks per output pixel
    constexpr int NPASS = (16 * (WC / 8) + 63) / 64;
    uint4 v0[KS];
    unsigned inb = 0, inb_n = 0;                                   // per k-step: the tap lies inside the image
    auto issue = [&](int g) {
        const int m = g * 16 + l15;
        const bool mok = m < a.M;
        const int n = mok ? c3x_fdiv(m, hw, a.rcp_hw) : 0, rem = m - n * hw;
        const int oy = mok ? c3x_fdiv(rem, a.Wo, a.rcp_wo) : 0, ox = rem - oy * a.Wo;
        const int iy0 = oy * a.stride - 1, ix0 = ox * a.stride - 1;
        const uint16_t* base = (const uint16_t*)a.act.data + (((ptrdiff_t)n * a.Hi + iy0) * a.Wi + ix0) * (ptrdiff_t)a.Ci;
        inb_n = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kh = (tbits >> (4 * ks)) & 3, kw = (tbits >> (4 * ks + 2)) & 3;
            const int iy = iy0 + kh, ix = ix0 + kw;
            const bool ok = mok && koff[ks] >= 0 && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
            v0[ks] = make_uint4(0, 0, 0, 0);
            if (ok) { v0[ks] = *(const uint4*)(base + koff[ks]); inb_n |= 1u << ks; }
        }
    };
    if (g0 < ngroups) issue(g0);
    for (int g = g0; g < ngroups; g += gstride) {
        const int m0 = g * 16;
        inb = inb_n;
        bf16x8_t bf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 v = v0[ks];
            if (has_coef && ((inb >> ks) & 1u)) {
                const int c = cik[ks];
                float s[8], t[8];
                *(float4*)&s[0] = *(const float4*)(lds_coef + c); *(float4*)&s[4] = *(const float4*)(lds_coef + c + 4);
                *(float4*)&t[0] = *(const float4*)(lds_coef + a.Ci + c); *(float4*)&t[4] = *(const float4*)(lds_coef + a.Ci + c + 4);
                v = act8(v, s, t);
            }
            bf[ks] = *(const bf16x8_t*)&v;
        }
        if (g + gstride < ngroups) issue(g + gstride);             // next group's fragments fly under the MFMAs / stores
        const bool mok = m0 + l15 < a.M;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][ks], bf[ks], acc, 0, 0, 0);
            const float4 bb = *(const float4*)(lds_bias + t * 16 + lg * 4);
            const mnas_f2 a0 = {acc[0] + bb.x, acc[1] + bb.y}, a1 = {acc[2] + bb.z, acc[3] + bb.w};
            if (mok) {
                mnas_stat2(a0, &s1[t][0], &s2[t][0]);
                mnas_stat2(a1, &s1[t][2], &s2[t][2]);
            }
            uint2 pk;
            pk.x = pack_bf16(a0.x, a0.y);
            pk.y = pack_bf16(a1.x, a1.y);
            *(uint2*)(st + l15 * SP + t * 16 + lg * 4) = pk;
        }
        __builtin_amdgcn_wave_barrier();                           // (LDS operations of one wave execute in order)
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int q = p * 64 + lane;
            const int px = q / cpp, ch = q - px * cpp;
            if (q < 16 * cpp && m0 + px < a.M)
                st_u4((uint16_t*)a.out + (size_t)(m0 + px) * a.Co + ch * 8, *(const uint4*)(st + px * SP + ch * 8), true);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (a.stats) {
        // 16 pixel lanes -> one value per cout (shuffle tree), then the four waves in order
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x1 = s1[t][r], x2 = s2[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
                if (l15 == 0) {
                    lds_fin[(wave * 2 + 0) * WC + t * 16 + lg * 4 + r] = x1;
                    lds_fin[(wave * 2 + 1) * WC + t * 16 + lg * 4 + r] = x2;
                }
            }
        __syncthreads();
        for (int i = tid; i < 2 * WC; i += 256) {
            const int r = i / WC, c = i - r * WC;
            const float v = ((lds_fin[(0 * 2 + r) * WC + c] + lds_fin[(1 * 2 + r) * WC + c]) + lds_fin[(2 * 2 + r) * WC + c]) +
                            lds_fin[(3 * 2 + r) * WC + c];
            if (c < a.Co) a.stats[((size_t)r * a.Co + c) * gridDim.x + blockIdx.x] = v;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct C3xPlan { int tpw, ks; size_t lds; };

int mnas_c3x_enabled() {
    static int on = -1;
    if (on < 0) on = mnas_diag_env("MNAS_C3X", 1);
    return on;
}
static bool c3x_plan(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad, C3xPlan* p) {
    if (!mnas_c3x_enabled() || kh != 3 || kw != 3 || pad != 1 || stride != 2) return false;
    if ((Ci & 7) || (Co & 7) || Ci < 8 || Co < 8 || Ho != (Hi - 1) / 2 + 1 || Wo != (Wi - 1) / 2 + 1) return false;
    if ((long long)N * Ho * Wo < 100000 || (long long)N * Hi * Wi * Ci >= (1ll << 31)) return false;      // the large maps only
    const int ks = (9 * Ci + 31) / 32, tiles = (Co + 15) / 16;
    // instantiated: 16 -> 24 (2 tiles x 5 k-steps), 24 -> 40 (3 x 7)
    if (tiles <= 2 && ks <= 5) { p->tpw = 2; p->ks = 5; }
    else if (tiles <= 3 && ks <= 7) { p->tpw = 3; p->ks = 7; }
    else return false;
    const int wc = p->tpw * 16;
    p->lds = (size_t)2 * Ci * 4 + (size_t)wc * 4 + (size_t)4 * 16 * (wc + 8) * 2 + (size_t)8 * wc * 4;
    return true;
}
int mnas_c3x_ok(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad) {
    C3xPlan p;
    return c3x_plan(N, Hi, Wi, Ci, Ho, Wo, Co, kh, kw, stride, pad, &p) ? 1 : 0;
}

int mnas_c3x_run(const MnasConvGemm* c, void* stream) {
    C3xPlan p;
    if (c->mode != 0 || c->resid || c->gate || !c->act.data || !c->out || c->nparts < 1 ||
        !c3x_plan(c->N, c->Hi, c->Wi, c->Ci, c->Ho, c->Wo, c->Co, c->kh, c->kw, c->stride, c->pad, &p)) return MNAS_EINVAL;
    C3xArgs a;
    a.N = c->N; a.Hi = c->Hi; a.Wi = c->Wi; a.Ci = c->Ci; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co;
    a.stride = c->stride;
    a.M = c->N * c->Ho * c->Wo;
    a.Kpad = (9 * c->Ci + 31) / 32 * 32;
    a.co_pad16 = (c->Co + 15) / 16 * 16;
    a.rcp_hw = a.M < (1 << 24) ? 1.0f / (float)(c->Ho * c->Wo) : 0.f;
    a.rcp_wo = a.M < (1 << 24) ? 1.0f / (float)c->Wo : 0.f;
    a.act = c->act; a.w = (const uint16_t*)c->w; a.bias = c->bias; a.out = c->out; a.stats = c->stats;
    hipStream_t s = (hipStream_t)stream;
#define MNAS_C3X(T_, K_) \
    if (p.tpw == T_ && p.ks == K_) { \
        hipLaunchKernelGGL((k_c3x<T_, K_>), dim3(c->nparts), dim3(256), p.lds, s, a); \
        MNAS_CHECK_LAUNCH(); \
        return MNAS_OK; \
    }
    MNAS_C3X(2, 5) MNAS_C3X(3, 7)
#undef MNAS_C3X
    return MNAS_EINVAL;
}
