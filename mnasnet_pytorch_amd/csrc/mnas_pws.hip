// Long-K / few-output-channel 1x1 GEMMs as a K-STREAMING kernel: the narrowing (project) convs forward, and the input
// gradient of the widening (expand) convs, on the 28x28 / 14x14 / 7x7 stages:
//     out[M][N] = f(in)[M][K] * W[N][K]^T         K = 240..1152, N = 40..192, M = 12 k..200 k pixels
// (ConvBlock(kernel_size=1), mnasnet.py:48-62 / its autograd mirror).  Same contract as mnas_conv_gemm modes 0 / 1.
//
// k_igemm walks K in LDS chunks: global -> VGPR -> transform -> LDS -> barrier -> fragments, one chunk in flight; at these
// sizes a launch is a chain of exposed latencies (0.5-1.9 TB/s measured).  Here nothing is staged:
//   * the K range is split over the NW waves of a workgroup (k-steps of 32 interleaved: wave w takes steps w, w+NW, ...);
//     each wave keeps ITS slice of the weight block in registers for the whole kernel (NT x KSW MFMA A-fragments);
//   * the activation / gradient fragments (8 consecutive k of one pixel = 16 contiguous bytes in NHWC) are loaded straight
//     from global memory into registers, one 16-pixel group ahead; the BatchNorm+ReLU ("act-on-load") or BatchNorm/ReLU
//     backward ("dy-on-load") transform runs in registers on the fragment (coefficients from LDS);
//   * per 16-pixel group a wave issues NT x KSW MFMAs into NT accumulators, parks its partial sums in LDS (double-buffered,
//     one barrier per group) and 16 x N/8 threads add the NW partials in wave order, apply the epilogue (bias + BatchNorm
//     partial statistics, or residual gradient + fused BatchNorm-backward reduce) and store 16 bytes each.
// Bytes in flight: KSW KB per wave and tensor -> 40-80 KB per CU with two workgroups resident: enough to cover HBM latency
// without any software pipeline.  Roofline: HBM.
#include "mnas_common.h"
#ifndef MNAS_PWS_D2
#define MNAS_PWS_D2 1        // fragments loaded TWO pixel groups ahead (0: one, A/B builds)
#endif

struct PwsArgs {
    int M, K, N;             // pixels, reduction length (input channels of this GEMM), outputs
    int Kpad, ksteps;        // K rounded to 32, k-steps
    int n_pad16;
    MnasActIn act;           // MODE 0
    MnasGradIn grad;         // MODE 1
    const uint16_t* w;       // [n_pad16][Kpad] (MNAS_PACK_FWD for mode 0, MNAS_PACK_DGRAD for mode 1)
    const float* bias;       // MODE 0
    const void* resid;       // MODE 1: bf16 [M][N] or NULL
    void* out;               // bf16 [M][N]
    float* stats;            // [2][N][gridDim.x] or NULL
    const void* red_y;       // MODE 1 fused reduce target (raw output of the ConvBlock whose gradient `out` is), or NULL
    const float* red_bn;
    const float* gate;       // MODE 0 GATE instantiations: float[images][K] multiplier applied after the activation (csrc/mnas_se.hip)
    int hw;                  // pixels per image (gate row = pixel / hw)
};

template <int MODE, int NT, int KSW, int NW, bool GATE = false>
__global__ __launch_bounds__(64 * NW) void k_pws(PwsArgs a) {
    static_assert(!GATE || MODE == 0, "the gate is an act-on-load extension of the forward");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NB = NT * 16, CROWS = MODE == 1 ? 5 : 2;
    constexpr int PP = NB + 4;                               // partial row pitch (floats)
    constexpr int NTH = 64 * NW;
    float* lds_coef = (float*)smem;                          // [CROWS][Kpad]
    float* lds_part = lds_coef + CROWS * a.Kpad;             // [2][NW][16][PP]
    float* lds_redc = lds_part + 2 * NW * 16 * PP;           // [4][NB]   fused-reduce coefficients (MODE 1)
    float* lds_fin = lds_part;                               // reused at the very end: [16][2][NB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.y * NB;
    const int nb_valid = min(NB, a.N - n0);                  // valid outputs of this block (multiple of 8)
    const int chunks = nb_valid >> 3;                        // 16-byte output chunks per pixel
    const bool has_coef = MODE == 1 || a.act.scale != nullptr;
    const bool do_red = MODE == 1 && a.red_y != nullptr;

    const int ngroups = (a.M + 15) >> 4;
    uint4 v0[KSW], v1[MODE == 1 ? KSW : 1];
    // D2 (round 6): a second register set, filled TWO groups ahead.  A group's arithmetic (<= 18 MFMAs, the partial exchange) is
    // ~1 us, a global load under load 2-4 us, and with 8 waves x ~200 VGPRs there is one workgroup per CU: with the loads only
    // one group ahead every iteration waited for them (SQ: the waves wait 60 % of their cycles, profiles/r06_sq_counters.txt).
    // Input-gradient mode only: the forward forms (4 waves, two workgroups per CU) LOSE with it (class 1.79 vs 1.75 ms).
    constexpr bool D2 = MNAS_PWS_D2 && !GATE && MODE == 1;
    uint4 w0[D2 ? KSW : 1], w1[(D2 && MODE == 1) ? KSW : 1];
    float gq[GATE ? KSW : 1][8];                             // GATE: the fragment's 8 multipliers, fetched with it
    auto issue_to = [&](int g, uint4* d0, uint4* d1) {
        const int m = g * 16 + l15;
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int k = (wave + NW * j) * 32 + lg * 8;
            d0[j] = make_uint4(0, 0, 0, 0);
            if (MODE == 1) d1[j] = make_uint4(0, 0, 0, 0);
            if (m < a.M && k < a.K) {
                const size_t off = (size_t)m * a.K + k;
                if (MODE == 0) d0[j] = *(const uint4*)((const uint16_t*)a.act.data + off);
                else {
                    d0[j] = *(const uint4*)((const uint16_t*)a.grad.g + off);
                    d1[j] = *(const uint4*)((const uint16_t*)a.grad.y + off);
                }
            }
        }
    };
    auto issue = [&](int g) {
        const int m = g * 16 + l15;
        const float* grow = GATE ? a.gate + (size_t)(m / a.hw) * a.K : nullptr;
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int k = (wave + NW * j) * 32 + lg * 8;
            v0[j] = make_uint4(0, 0, 0, 0);
            if (MODE == 1) v1[j] = make_uint4(0, 0, 0, 0);
            if (m < a.M && k < a.K) {
                const size_t off = (size_t)m * a.K + k;
                if constexpr (GATE) {
                    const float4 a0 = *(const float4*)(grow + k), a1 = *(const float4*)(grow + k + 4);
                    gq[j][0] = a0.x; gq[j][1] = a0.y; gq[j][2] = a0.z; gq[j][3] = a0.w;
                    gq[j][4] = a1.x; gq[j][5] = a1.y; gq[j][6] = a1.z; gq[j][7] = a1.w;
                }
                if (MODE == 0) v0[j] = *(const uint4*)((const uint16_t*)a.act.data + off);
                else {
                    v0[j] = *(const uint4*)((const uint16_t*)a.grad.g + off);
                    v1[j] = *(const uint4*)((const uint16_t*)a.grad.y + off);
                }
            }
        }
    };
    int it = 0;
    auto first_issue = [&]() {
        if ((int)blockIdx.x < ngroups) { if constexpr (D2) issue_to(blockIdx.x, v0, v1); else issue(blockIdx.x); }
        if constexpr (D2) { if ((int)(blockIdx.x + gridDim.x) < ngroups) issue_to(blockIdx.x + gridDim.x, w0, w1); }
    };
    if (MNAS_EARLY) first_issue();           // the first groups' loads, the weights and the tables below share one round trip
    // ---- this wave's slice of the weight block: A fragments [cout l15][k = ks*32 + lg*8 ..], k-steps ks = wave + NW*j
    bf16x8_t wf[NT][KSW];
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        const int ks = wave + NW * j;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int row = n0 + nt * 16 + l15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ks < a.ksteps && row < a.n_pad16) v = *(const uint4*)(a.w + (size_t)row * a.Kpad + ks * 32 + lg * 8);
            wf[nt][j] = *(const bf16x8_t*)&v;
        }
    }
    mnas_fill_table(lds_coef, CROWS * a.Kpad, tid, NTH, [&](int i) {
        const int r = i / a.Kpad, c = i - r * a.Kpad;
        float v = 0.f;
        if (c < a.K) {
            if (MODE == 1) v = a.grad.coef[(size_t)r * a.K + c];
            else if (has_coef) v = r == 0 ? a.act.scale[c] : a.act.shift[c];
        }
        return v;
    });
    if (do_red)
        mnas_fill_table(lds_redc, 4 * NB, tid, NTH, [&](int i) {
            const int r = i / NB, co = n0 + i % NB;
            return co < a.N ? mnas_red_coef(a.red_bn, a.N, r, co) : 0.f;
        });
    __syncthreads();
    if (!MNAS_EARLY) first_issue();

    // ---- epilogue role: thread te = p * chunks + c8 (p = pixel of the group, c8 = 8-channel chunk), fixed for the whole kernel
    const bool epi = tid < 16 * chunks;
    const int ep = epi ? tid / chunks : 0, ec8 = epi ? tid - ep * chunks : 0;
    float bias8[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bias8[j] = (MODE == 0 && epi && a.bias) ? a.bias[n0 + ec8 * 8 + j] : 0.f;
        s1[j] = 0.f; s2[j] = 0.f;
    }

    for (int g = blockIdx.x; g < ngroups; g += gridDim.x, ++it) {
        const int m0 = g * 16;
        // ---- fragments of this group (transform in registers), then the next group's loads go out
        bf16x8_t bf[KSW];
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int k = (wave + NW * j) * 32 + lg * 8;
            uint4 v = v0[j];
            if (has_coef && k < a.K && m0 + l15 < a.M) {
                float cf[CROWS][8];
#pragma unroll
                for (int r = 0; r < CROWS; ++r) {
                    *(float4*)&cf[r][0] = *(const float4*)(lds_coef + r * a.Kpad + k);
                    *(float4*)&cf[r][4] = *(const float4*)(lds_coef + r * a.Kpad + k + 4);
                }
                if (MODE == 0) {
                    if constexpr (GATE) v = act8g(v, cf[0], cf[1], gq[j]);
                    else v = act8(v, cf[0], cf[1]);
                } else {
                    float o[8];
                    dy8(v, v1[j], cf[0], cf[1], cf[2 % CROWS], cf[3 % CROWS], cf[4 % CROWS], o);
                    v = pack8(o);
                }
            }
            bf[j] = *(const bf16x8_t*)&v;
        }
        // epilogue operands of this group (residual gradient / reduce target): issued now, used after the barrier
        uint4 rpre = make_uint4(0, 0, 0, 0), ypre = make_uint4(0, 0, 0, 0);
        if (MODE == 1 && epi && m0 + ep < a.M) {
            const size_t o = (size_t)(m0 + ep) * a.N + n0 + ec8 * 8;
            if (a.resid) rpre = *(const uint4*)((const uint16_t*)a.resid + o);
            if (do_red) ypre = *(const uint4*)((const uint16_t*)a.red_y + o);
        }
        if constexpr (D2) {                                  // the set filled one group ago becomes current; refill two groups ahead
#pragma unroll
            for (int j = 0; j < KSW; ++j) { v0[j] = w0[j]; if (MODE == 1) v1[j] = w1[j]; }
            if (g + 2 * (int)gridDim.x < ngroups) issue_to(g + 2 * gridDim.x, w0, w1);
        } else if (g + (int)gridDim.x < ngroups) issue(g + gridDim.x);
        f32x4_t acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KSW; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][j], bf[j], acc[nt], 0, 0, 0);
        // ---- park the partial sums: [buf][wave][pixel l15][cout nt*16 + lg*4 ..]
        float* part = lds_part + ((it & 1) * NW + wave) * 16 * PP + l15 * PP + lg * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *(float4*)(part + nt * 16) = *(const float4*)&acc[nt];
        __syncthreads();
        if (epi && m0 + ep < a.M) {
            float v[8];
            const float* src = lds_part + (it & 1) * NW * 16 * PP + ep * PP + ec8 * 8;
            *(float4*)&v[0] = *(const float4*)src;
            *(float4*)&v[4] = *(const float4*)(src + 4);
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const float4 x0 = *(const float4*)(src + w * 16 * PP), x1 = *(const float4*)(src + w * 16 * PP + 4);
                v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w;
                v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
            }
            const size_t o = (size_t)(m0 + ep) * a.N + n0 + ec8 * 8;
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {                // channel pairs in float2 (v_pk_fma_f32)
                    const mnas_f2 w2 = mnas_ld2(v + 2 * j) + mnas_ld2(bias8 + 2 * j);
                    v[2 * j] = w2.x; v[2 * j + 1] = w2.y;
                    mnas_stat2(w2, s1 + 2 * j, s2 + 2 * j);
                }
                *(uint4*)((uint16_t*)a.out + o) = pack8(v);
            } else {
                if (a.resid) {
                    float r[8];
                    unpack8(rpre, r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += r[j];
                }
                const uint4 pk = pack8(v);
                *(uint4*)((uint16_t*)a.out + o) = pk;
                if (do_red) {
                    // dz = g*[s*y+t>0] with g as stored (bf16), xhat = y*invstd - mean*invstd   (as k_igemm's fused reduce)
                    const uint32_t gu[4] = {pk.x, pk.y, pk.z, pk.w}, yu[4] = {ypre.x, ypre.y, ypre.z, ypre.w};
                    const int cl = ec8 * 8;
#pragma unroll
                    for (int j = 0; j < 4; ++j)              // channel pairs in float2 (v_pk_fma_f32)
                        mnas_red2(gu[j], yu[j], mnas_ld2(lds_redc + cl + 2 * j), mnas_ld2(lds_redc + NB + cl + 2 * j),
                                  mnas_ld2(lds_redc + 2 * NB + cl + 2 * j), mnas_ld2(lds_redc + 3 * NB + cl + 2 * j), s1 + 2 * j, s2 + 2 * j);
                }
            }
        }
    }
    if ((MODE == 0 || do_red) && a.stats) {
        // the 16 pixel-threads of a chunk are combined in pixel order (deterministic)
        __syncthreads();
        if (epi) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lds_fin[(ep * 2 + 0) * NB + ec8 * 8 + j] = s1[j];
                lds_fin[(ep * 2 + 1) * NB + ec8 * 8 + j] = s2[j];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * nb_valid; i += NTH) {
            const int r = i / nb_valid, cl = i - r * nb_valid;
            float v = lds_fin[r * NB + cl];
            for (int p = 1; p < 16; ++p) v += lds_fin[(p * 2 + r) * NB + cl];
            a.stats[((size_t)r * a.N + n0 + cl) * gridDim.x + blockIdx.x] = v;           // [2][N][P]
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct PwsPlan { int nt, ksw, nw, nblocks; size_t lds; };
static bool pws_plan(int mode, int M, int K, int N, PwsPlan* p) {
    if ((K & 7) || (N & 7) || K < 192 || N < 8 || M < 1) return false;       // long reductions only
    // input-gradient mode: with 4 waves (KSW = 5) the dy-on-load transform (two tensors, five coefficient rows) pushed the kernel
    // to 256 VGPRs = one 4-wave workgroup per CU and the step lost 0.47 ms against k_igemm; 8 waves per workgroup (below) fix that
    // round 2 measured this form neutral in the step although faster per launch (dgrad class 1.05 -> 0.99 ms): the side stream's
    // weight-gradient kernels took what it freed.  With the weight gradients on the main stream (round 3 default) it shows:
    // 14x14 expand convs 58 -> 49 us, 80->480 51 -> 41 us, 7x7 108 -> 65 us, step 11.20 -> 11.09 ms.  MNAS_PWS=1 (diagnosis
    // build) restores k_igemm for the input gradients.
    if (mode == 1 && mnas_pws_enabled() < 2) return false;
    if (M > 250000) return false;                                            // the 112x112 / 56x56 layers stay on k_igemm / k_pw_bwd
    const int ksteps = (K + 31) / 32;
    const int tiles = (N + 15) / 16;
    int nt = tiles <= 3 ? 3 : 6;
    const int nblocks = (tiles + nt - 1) / nt;
    if (nblocks * nt - tiles > 3) return false;
    // input-gradient mode carries two tensors and five coefficient rows per fragment: 8 waves (KSW <= 3) keep it under 200 VGPRs
    const int nw = (ksteps > 20 || (mode == 1 && ksteps >= 12)) ? 8 : 4;
    const int ksw_need = (ksteps + nw - 1) / nw;
    if (ksw_need > (mode == 1 ? 3 : 5)) return false;
    p->nt = nt; p->nw = nw; p->ksw = ksw_need <= 2 ? 2 : (ksw_need == 3 ? 3 : 5); p->nblocks = nblocks;
    const int NB = nt * 16, Kpad = ksteps * 32, crows = mode == 1 ? 5 : 2;
    size_t part = (size_t)2 * nw * 16 * (NB + 4) * 4, fin = (size_t)16 * 2 * NB * 4;
    p->lds = (size_t)crows * Kpad * 4 + (part > fin ? part : fin) + (size_t)4 * NB * 4;
    return p->lds <= 160 * 1024;
}
int mnas_pws_parts(int mode, int M, int K, int N) {
    PwsPlan p;
    if (!mnas_pws_enabled() || !pws_plan(mode, M, K, N, &p)) return -1;
    const int ngroups = (M + 15) / 16;
    const int base = mnas_diag_env("MNAS_PWS_WGS", 512);      // (diagnosis build: the sweep of DESIGN_HISTORY.md round 5)
    int want = base / p.nblocks;                      // two workgroups per CU
    if (p.nw == 8) want = base / 2 / p.nblocks;
    if (want < 32) want = 32;
    return ngroups < want ? ngroups : want;
}

template <int MODE, int NT, int KSW, int NW>
static int pws_launch(const PwsArgs& a, const PwsPlan& p, int nparts, hipStream_t s) {
    if (a.gate) {       // gated act-on-load: forward, short per-wave K slices only (registers: 8 more per k-step of the slice)
        if constexpr (MODE == 0 && KSW <= 3) {
            hipLaunchKernelGGL((k_pws<MODE, NT, KSW, NW, true>), dim3(nparts, p.nblocks), dim3(64 * NW), p.lds, s, a);
            MNAS_CHECK_LAUNCH();
            return MNAS_OK;
        }
        return MNAS_EINVAL;
    }
    hipLaunchKernelGGL((k_pws<MODE, NT, KSW, NW>), dim3(nparts, p.nblocks), dim3(64 * NW), p.lds, s, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
template <int MODE>
static int pws_dispatch(const PwsArgs& a, const PwsPlan& p, int nparts, hipStream_t s) {
#define MNAS_PWS(NT_, KSW_, NW_) if (p.nt == NT_ && p.ksw == KSW_ && p.nw == NW_) return pws_launch<MODE, NT_, KSW_, NW_>(a, p, nparts, s);
    MNAS_PWS(3, 2, 4) MNAS_PWS(3, 5, 4) MNAS_PWS(6, 2, 4) MNAS_PWS(6, 5, 4) MNAS_PWS(3, 5, 8) MNAS_PWS(6, 5, 8)
    MNAS_PWS(3, 2, 8) MNAS_PWS(6, 2, 8) MNAS_PWS(3, 3, 8) MNAS_PWS(6, 3, 8) MNAS_PWS(3, 3, 4) MNAS_PWS(6, 3, 4)
#undef MNAS_PWS
    return MNAS_EINVAL;
}

// 1 when the forward launch of this shape runs here AND takes a gate (mnas_conv_gemm_gate_ok)
int mnas_pws_gate_ok(int M, int K, int N) {
    PwsPlan p;
    return (mnas_pws_enabled() && pws_plan(0, M, K, N, &p) && p.ksw <= 3) ? 1 : 0;
}

// called by mnas_conv_gemm for 1x1 convs when mnas_pws_parts(...) > 0
int mnas_pws_run(const MnasConvGemm* c, void* stream) {
    const int M = c->N * c->Ho * c->Wo;
    PwsPlan p;
    if (!pws_plan(c->mode, M, c->Ci, c->Co, &p) || (c->mode == 1 && !c->grad.coef)) return MNAS_EINVAL;
    PwsArgs a;
    a.M = M; a.K = c->Ci; a.N = c->Co;
    a.ksteps = (c->Ci + 31) / 32; a.Kpad = a.ksteps * 32;
    a.n_pad16 = (c->Co + 15) / 16 * 16;
    a.act = c->act; a.grad = c->grad; a.w = (const uint16_t*)c->w; a.bias = c->bias; a.resid = c->resid; a.out = c->out;
    a.stats = c->stats; a.red_y = c->mode == 1 ? c->red_y : nullptr; a.red_bn = c->red_bn;
    a.gate = c->mode == 0 ? c->gate : nullptr; a.hw = c->Ho * c->Wo;
    if (a.gate && (!c->act.scale || p.ksw > 3)) return MNAS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    return c->mode == 0 ? pws_dispatch<0>(a, p, c->nparts, s) : pws_dispatch<1>(a, p, c->nparts, s);
}
