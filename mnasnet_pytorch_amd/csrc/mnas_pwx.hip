// 1x1 (pointwise) forward of the WIDENING convs on the small / mid feature maps (the expand convs of MBConv_block, mnasnet.py:116-121:
// 40 -> 240 at 28x28, 80 -> 480 and 96 -> 576 at 14x14) as a weight-stationary, output-streaming kernel:
//     out[M][Co] = act(in)[M][Ci] * W[Co][Ci]^T + bias,      Ci <= 96, Co = 6 Ci, M = 50 k .. 200 k pixels
// Same contract as mnas_conv_gemm mode 0 (raw bf16 output, per-workgroup BatchNorm partial statistics, producer's BatchNorm+ReLU
// applied on load).  Behind mnas_conv_gemm, in front of k_pwf (csrc/mnas_pwf.hip).
//
// k_pwf walks (64-pixel tile, 96-cout block) phases with one barrier each: 18 MFMAs per wave and phase, and a launch on these maps
// is a few hundred such phases per CU -- a chain of barrier / DMA-wait latencies (measured 1.75-2.1 TB/s for a write-dominated
// stream, 38 us for 67 MB at 96 -> 576).  Here:
//   * the WHOLE weight matrix is register-resident in a workgroup: the cout tiles are split over the NW waves (TPW tiles each, every
//     wave the full K = 2-3 k-steps: 16-72 VGPRs of MFMA A fragments), loaded once, so there is no reduction across waves and no
//     barrier in the pixel loop at all;
//   * workgroups walk 16-pixel groups persistently; a group's activation fragments (8 consecutive input channels of one pixel = 16
//     contiguous bytes) go from global memory straight to registers one group ahead, BatchNorm+ReLU applied in registers; all waves
//     of a workgroup read the same fragments (the small operand: L1 / L2 hits);
//   * a wave's [16 pixels][TPW*16 couts] result goes through a wave-private LDS stage and leaves as 16-byte stores of contiguous
//     TPW*32-byte row segments; statistics stay in registers across all groups (each cout has exactly one owner lane group).
// Measured in the bs-256 step (same call as k_pwf): 96 -> 576 at 14x14 36 -> 26 us, 80 -> 480 30-37 -> 24-29 us, 40 -> 240 at 28x28
// 52 -> 36 us.  Roofline: HBM writes (the output is 6x the input).
#include "mnas_common.h"

struct PwxArgs {
    int M, Ci, Co;
    int Kpad;                // Ci rounded up to 32
    int co_pad16;
    MnasActIn act;
    const uint16_t* w;       // MNAS_PACK_FWD: [co_pad16][Kpad]
    const float* bias;
    void* out;
    float* stats;            // [2][Co][gridDim.x] or NULL
};

template <int TPW, int KS, int NW>
__global__ __launch_bounds__(64 * NW) void k_pwx(PwxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTH = 64 * NW, WC = TPW * 16, SP = WC + 8;       // couts per wave, stage row pitch (bf16 elements)
    float* lds_coef = (float*)smem;                                // [2][Kpad] act-on-load scale / shift
    float* lds_bias = lds_coef + 2 * a.Kpad;                       // [NW*WC]
    uint16_t* stage = (uint16_t*)(lds_bias + NW * WC);             // [NW][16][SP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const bool has_coef = a.act.scale != nullptr;
    const int cb0 = blockIdx.y * NW * WC;                          // first cout of this workgroup's block (grid.y: wide outputs)
    const int cw0 = cb0 + wave * WC;                               // first cout of this wave

    uint16_t* st = stage + wave * 16 * SP;
    const int ngroups = (a.M + 15) >> 4;
    uint4 v0[KS];
    auto issue = [&](int g) {
        const int m = g * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 32 + lg * 8;
            v0[ks] = make_uint4(0, 0, 0, 0);
            if (m < a.M && k < a.Ci) v0[ks] = *(const uint4*)((const uint16_t*)a.act.data + (size_t)m * a.Ci + k);
        }
    };
    if (MNAS_EARLY && (int)blockIdx.x < ngroups) issue(blockIdx.x);    // ahead of the setup: (data, weights, tables) share one round trip
    // ---- this wave's cout tiles, full K: A fragments [cout l15][k = ks*32 + lg*8 ..]
    bf16x8_t wf[TPW][KS];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int row = cw0 + t * 16 + l15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < a.co_pad16) v = *(const uint4*)(a.w + (size_t)row * a.Kpad + ks * 32 + lg * 8);
            wf[t][ks] = *(const bf16x8_t*)&v;
        }
    mnas_fill_table(lds_coef, 2 * a.Kpad, tid, NTH, [&](int i) {
        const int r = i / a.Kpad, c = i - r * a.Kpad;
        return (has_coef && c < a.Ci) ? (r == 0 ? a.act.scale[c] : a.act.shift[c]) : 0.f;
    });
    mnas_fill_table(lds_bias, NW * WC, tid, NTH, [&](int i) { return (a.bias && cb0 + i < a.Co) ? a.bias[cb0 + i] : 0.f; });
    float s1[TPW][4], s2[TPW][4];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
    __syncthreads();                                               // coefficient / bias tables visible (the only barrier)
    if (!MNAS_EARLY && (int)blockIdx.x < ngroups) issue(blockIdx.x);

    constexpr int CPP = TPW * 2;                                   // 16-byte chunks per pixel of this wave's cout range
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int m0 = g * 16;
        bf16x8_t bf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 32 + lg * 8;
            uint4 v = v0[ks];
            if (has_coef && k < a.Ci && m0 + l15 < a.M) {
                float s[8], t[8];
                *(float4*)&s[0] = *(const float4*)(lds_coef + k); *(float4*)&s[4] = *(const float4*)(lds_coef + k + 4);
                *(float4*)&t[0] = *(const float4*)(lds_coef + a.Kpad + k); *(float4*)&t[4] = *(const float4*)(lds_coef + a.Kpad + k + 4);
                v = act8(v, s, t);
            }
            bf[ks] = *(const bf16x8_t*)&v;
        }
        if (g + (int)gridDim.x < ngroups) issue(g + gridDim.x);   // next group's fragments fly under the MFMAs / stores
        const bool mok = m0 + l15 < a.M;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][ks], bf[ks], acc, 0, 0, 0);
            // lane holds couts cw0 + t*16 + lg*4 + {0..3} of pixel m0 + l15
            const float4 bb = *(const float4*)(lds_bias + wave * WC + t * 16 + lg * 4);
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
        // wave-private stage -> 16-byte stores of whole row segments (LDS operations of one wave execute in order)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < (16 * CPP + 63) / 64; ++i) {
            const int q = i * 64 + lane;
            const int px = q / CPP, ch = q - px * CPP;
            const int co = cw0 + ch * 8;
            if (q < 16 * CPP && m0 + px < a.M && co < a.Co)
                st_u4((uint16_t*)a.out + (size_t)(m0 + px) * a.Co + co, *(const uint4*)(st + px * SP + ch * 8), true);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (a.stats) {
        // per-cout sums: the 16 pixel lanes of a cout quad combined by a shuffle tree; every cout has one owner -> no atomics,
        // no cross-wave step, fixed order
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x1 = s1[t][r], x2 = s2[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
                const int c = cw0 + t * 16 + lg * 4 + r;
                if (l15 == 0 && c < a.Co) {
                    a.stats[((size_t)0 * a.Co + c) * gridDim.x + blockIdx.x] = x1;
                    a.stats[((size_t)1 * a.Co + c) * gridDim.x + blockIdx.x] = x2;
                }
            }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct PwxPlan { int tpw, ks, nw, nblocks; size_t lds; };

int mnas_pwx_enabled() {
    static int on = -1;
    if (on < 0) on = mnas_diag_env("MNAS_PWX", 1);
    return on;
}
static bool pwx_plan(int M, int Ci, int Co, PwxPlan* p) {
    if (!mnas_pwx_enabled() || (Ci & 7) || (Co & 7) || Ci < 8 || Co < 8 || M < 1) return false;
    // widening convs on the <= 28x28 maps (bs 256).  192 -> 1152 at 7x7 (three cout blocks over grid.y, 6 waves x 4 tiles x 6 k-steps)
    // was measured too: 32.7 us against 30.4 for the K-streaming kernel (csrc/mnas_pws.hip) -- not instantiated
    if (Ci > 96 || Co < 4 * Ci || M > 250000) return false;
    const int ks = (Ci + 31) / 32;
    int tiles = (Co + 15) / 16;
    p->nblocks = 1;
    if (ks > 3) {                                                  // 192 -> 1152: three blocks of 24 cout tiles over grid.y (6 waves x 4 tiles)
        if (tiles % 24) return false;
        p->nblocks = tiles / 24; tiles = 24;
    }
    // waves x tiles-per-wave covering the cout tiles with the least padding; fragments TPW*KS*4 <= 72 VGPRs
    int best = 1 << 30;
    p->nw = 0;
    static const int nws[] = {4, 6, 8};
    for (int nw : nws) {
        const int tpw = (tiles + nw - 1) / nw;
        if (tpw < 1 || tpw > 6 || tpw * ks > 24) continue;
        const int waste = nw * tpw - tiles;
        if (waste < best) { best = waste; p->nw = nw; p->tpw = tpw; }
    }
    if (!p->nw) return false;
    p->ks = ks;
    // instantiated: 96 -> 576 (6 waves x 6 tiles), 80 -> 480 (6 x 5), 40 -> 240 (4 x 4); anything else stays on k_pwf
    if (!((p->tpw == 6 && ks == 3 && p->nw == 6) || (p->tpw == 5 && ks == 3 && p->nw == 6) || (p->tpw == 4 && ks == 2 && p->nw == 4)))
        return false;
    const int Kpad = ks * 32, wc = p->tpw * 16;
    p->lds = (size_t)2 * Kpad * 4 + (size_t)p->nw * wc * 4 + (size_t)p->nw * 16 * (wc + 8) * 2;
    return p->lds <= 64 * 1024;
}
// persistent workgroups for a launch of this shape, or -1 (not this kernel's case)
int mnas_pwx_parts(int M, int Ci, int Co) {
    PwxPlan p;
    if (!pwx_plan(M, Ci, Co, &p)) return -1;
    const int ngroups = (M + 15) / 16;
    // one workgroup per CU for the 6-wave forms (178 / 158 VGPRs: a second one would not be resident, and every workgroup loads
    // the whole weight matrix): 96 -> 576 36 -> 26 us with 256, 34 with 512, 41 with 768 workgroups; two per CU for the 4-wave form
    // (40 -> 240: 52 -> 36 us with 512, 39-47 with 256)
    static int wgs = -1;
    if (wgs < 0) wgs = mnas_diag_env("MNAS_PWX_WGS", 0);
    int want = wgs > 0 ? wgs : (p.nw == 6 ? 256 : 512);
    want = want / p.nblocks;
    if (want < 32) want = 32;
    return ngroups < want ? ngroups : want;
}

int mnas_pwx_forward(const MnasConvGemm* c, void* stream) {
    PwxPlan p;
    const int M = c->N * c->Ho * c->Wo;
    if (!pwx_plan(M, c->Ci, c->Co, &p) || c->resid || c->gate || !c->act.data || !c->out) return MNAS_EINVAL;
    PwxArgs a;
    a.M = M; a.Ci = c->Ci; a.Co = c->Co;
    a.Kpad = p.ks * 32;
    a.co_pad16 = (c->Co + 15) / 16 * 16;
    a.act = c->act; a.w = (const uint16_t*)c->w; a.bias = c->bias; a.out = c->out; a.stats = c->stats;
    hipStream_t s = (hipStream_t)stream;
#define MNAS_PWX(T_, K_, W_) \
    if (p.tpw == T_ && p.ks == K_ && p.nw == W_) { \
        hipLaunchKernelGGL((k_pwx<T_, K_, W_>), dim3(c->nparts, p.nblocks), dim3(64 * W_), p.lds, s, a); \
        MNAS_CHECK_LAUNCH(); \
        return MNAS_OK; \
    }
    MNAS_PWX(6, 3, 6) MNAS_PWX(5, 3, 6) MNAS_PWX(4, 2, 4)
#undef MNAS_PWX
    return MNAS_EINVAL;
}
