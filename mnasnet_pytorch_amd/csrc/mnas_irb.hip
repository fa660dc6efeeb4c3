// Fused inverted-residual block (MBConv_block, mnasnet.py:105-137) on the SMALL feature maps (the 14x14 and 7x7 stages:
// features.5/6/7 of MNASNet-1.0, 7 block applications per step), forward.  The backward kernels live in mnas_irb_bwd.hip.
//
//     x (N,H,W,C) --1x1 expand--> y1 (N,H,W,E) --BN1+ReLU--> a1 --k x k depthwise--> y2 --BN2+ReLU--> a2 --1x1 project--> y3
//
// What is fused and why.  On these maps the unfused launches are latency-bound (tens of MB per launch, 1.1-2.0 TB/s) and the
// t-times expanded tensors y1, g1, g2 round-trip HBM three times each.  Here the EXPANDED TENSOR y1 NEVER REACHES HBM:
//   * forward (k_irb_fwd): one (image, 32-channel slice of E) unit at a time -- y1 slice = W1[slice] . act(x) on the matrix
//     cores (the activation fragments come straight from global memory into registers, the weight slice lives in registers for
//     the whole workgroup), bias + bf16 rounding + BatchNorm1 + ReLU in the MFMA epilogue, written as the zero-padded LDS image
//     the depthwise stage wants; k x k depthwise on the vector ALU from LDS (a thread owns one output row of one channel pair:
//     every LDS value is read once per row pass and reused across the k taps in registers); raw y2 + its BatchNorm partial
//     statistics leave the chip.  BatchNorm1's batch statistics are known BEFORE the launch from the covariance of x
//     (mnas_gram + mnas_gram_bn_finalize: y1 is linear in act(x)).
//   * backward RECOMPUTES y1 from x instead of reading it (mnas_irb_bwd.hip).
// y2 (the depthwise output) IS stored: recomputing it would redo the depthwise conv on the vector ALU (25 FMA per element,
// 9 us per image at the ALU peak) to save a 15 us read.
//
// Roofline: the depthwise stage binds (fp32 vector ALU: 2 x k x k flop per element against 157 TFLOP/s), then HBM (y2 written
// once); the matrix-core share is ~10 % of the unit time.
#include "mnas_common.h"

struct IrbFwdArgs {
    int N, H, W, C, E;
    int HW, NI;                 // pixels per image, images per pass (2 for 7x7 maps: 98 pixels, the same 14 task rows)
    int Kpad;                   // C rounded up to 32 (row pitch of the packed expand weights)
    int npt;                    // 16-pixel tiles per pass
    int ipg;                    // image passes per workgroup (grid.y groups)
    MnasActIn x;
    const uint16_t* w1;         // MNAS_PACK_FWD [E_pad16][Kpad]
    const float* b1;
    const float* bn1;           // bnbuf of the expand conv: rows 0,1 = scale, shift
    const float* wdw;           // [k*k][E]
    const float* bdw;
    uint16_t* y1;               // optional (N,H,W,E)
    uint16_t* y2;               // (N,H,W,E)
    float* stats;               // [2][E][gridDim.y] or NULL
};

// One workgroup = 256 threads = (one 32-channel slice of E) x (a group of image passes).
//   MFMA phase : wave w owns the 16-pixel tiles w, w+4, ...; both 16-channel tiles of the slice.
//   depthwise  : thread (pair = tid & 15, slot = tid >> 4): output row `slot` (image slot / H, row slot % H) of channel pair `pair`.
// LDS image: [NI][(H + 2p)][RW][32] bf16 with RW = W + 2p rounded up to an ODD number of pixels: a pixel is 16 dwords, so two
// rows of a 32-lane read group fall into different bank halves (conflict-free ds_read_b32).
template <int KS, int WW, int KST>
__global__ __launch_bounds__(256, 2) void k_irb_fwd(IrbFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = KS / 2;
    constexpr int RW = (WW + 2 * P) | 1;
    constexpr int NPTW = 4;                                        // pixel tiles per wave (<= 13 tiles per pass)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int e0 = blockIdx.x * 32;
    const int RH = a.H + 2 * P;
    uint32_t* a1p = (uint32_t*)smem;                               // [NI*RH*RW][16] channel pairs
    const int img_dw = a.NI * RH * RW * 16;
    float* lds_x = (float*)(a1p + img_dw);                         // [2][Kpad] act-on-load coefficients of x
    float* lds_red = lds_x + 2 * a.Kpad;                           // [16 slots][2][32]
    const bool hasx = a.x.scale != nullptr;

    for (int i = tid; i < img_dw; i += 256) a1p[i] = 0u;           // zero borders (interiors are rewritten every pass)
    for (int i = tid; i < 2 * a.Kpad; i += 256) {
        const int c = i < a.Kpad ? i : i - a.Kpad;
        lds_x[i] = (hasx && c < a.C) ? (i < a.Kpad ? a.x.scale[c] : a.x.shift[c]) : 0.f;
    }
    // ---- per-workgroup constants in registers
    const int ksteps = a.Kpad >> 5;                                // <= KST (3: C <= 96, 6: C <= 192)
    bf16x8_t wfrag[2][KST];
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            wfrag[et][ks] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            if (ks < ksteps) wfrag[et][ks] = *(const bf16x8_t*)(a.w1 + (size_t)(e0 + et * 16 + l15) * a.Kpad + ks * 32 + lg * 8);
        }
    float b1r[2][4], s1r[2][4], t1r[2][4];
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = e0 + et * 16 + lg * 4 + r;
            b1r[et][r] = a.b1 ? a.b1[e] : 0.f;
            s1r[et][r] = a.bn1[e];
            t1r[et][r] = a.bn1[a.E + e];
        }
    // this lane's pixels in the MFMA phase: LDS dword offset of the pixel inside the padded image, or -1
    int poff[NPTW], ppix[NPTW];
#pragma unroll
    for (int i = 0; i < NPTW; ++i) {
        const int pt = wave + 4 * i;
        const int p = pt * 16 + l15;
        const bool ok = pt < a.npt && p < a.NI * a.HW;
        const int im = ok ? p / a.HW : 0, pp = ok ? p - im * a.HW : 0;
        const int py = pp / a.W, px = pp - py * a.W;
        ppix[i] = ok ? p : -1;
        poff[i] = ((im * RH + py + P) * RW + px + P) * 16;
    }
    // depthwise task of this thread
    const int pair = tid & 15, slot = tid >> 4;
    const bool task = slot < a.NI * a.H;
    const int t_im = task ? slot / a.H : 0, t_row = task ? slot - t_im * a.H : 0;
    float2 wt[KS * KS];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) wt[t] = *(const float2*)(a.wdw + (size_t)t * a.E + e0 + pair * 2);
    float2 bd = make_float2(0.f, 0.f);
    if (a.bdw) bd = *(const float2*)(a.bdw + e0 + pair * 2);
    float2 st1 = make_float2(0.f, 0.f), st2 = make_float2(0.f, 0.f);
    __syncthreads();

    const int npass = (a.N + a.NI - 1) / a.NI;
    for (int pi = 0; pi < a.ipg; ++pi) {
        const int pass = blockIdx.y * a.ipg + pi;
        if (pass >= npass) break;                                   // uniform
        const int n0 = pass * a.NI;
        const int nimg = min(a.NI, a.N - n0);
        const uint16_t* xb = (const uint16_t*)a.x.data + (size_t)n0 * a.HW * a.C;
        // ---- expand: D[e][pix] = W1[e][:] . act(x)[pix][:]  (all activation fragments of the pass are loaded up front)
        uint4 xv[NPTW][KST];
#pragma unroll
        for (int i = 0; i < NPTW; ++i) {
            const bool live = ppix[i] >= 0 && ppix[i] < nimg * a.HW;
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                xv[i][ks] = make_uint4(0, 0, 0, 0);
                const int c = ks * 32 + lg * 8;
                if (ks < ksteps && live && c < a.C) xv[i][ks] = *(const uint4*)(xb + (size_t)ppix[i] * a.C + c);
            }
        }
#pragma unroll
        for (int i = 0; i < NPTW; ++i) {
            if (wave + 4 * i >= a.npt) break;                       // uniform
            const bool live = ppix[i] >= 0 && ppix[i] < nimg * a.HW;
            f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                if (ks >= ksteps) break;
                uint4 u = xv[i][ks];
                if (hasx) {
                    const int c = ks * 32 + lg * 8;
                    float s[8], t[8];
                    *(float4*)&s[0] = *(const float4*)(lds_x + c); *(float4*)&s[4] = *(const float4*)(lds_x + c + 4);
                    *(float4*)&t[0] = *(const float4*)(lds_x + a.Kpad + c); *(float4*)&t[4] = *(const float4*)(lds_x + a.Kpad + c + 4);
                    u = act8(u, s, t);
                    if (!live) u = make_uint4(0, 0, 0, 0);
                }
                const bf16x8_t bfrag = *(const bf16x8_t*)&u;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[0][ks], bfrag, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[1][ks], bfrag, acc[1], 0, 0, 0);
            }
            if (live) {
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[et][r] + b1r[et][r];
                    uint2 yr;
                    yr.x = pack_bf16(v[0], v[1]); yr.y = pack_bf16(v[2], v[3]);
                    if (a.y1) *(uint2*)(a.y1 + ((size_t)n0 * a.HW + ppix[i]) * a.E + e0 + et * 16 + lg * 4) = yr;
                    const float q[4] = {bf_lo(yr.x), bf_hi(yr.x), bf_lo(yr.y), bf_hi(yr.y)};
                    uint2 ar;
                    ar.x = pack_bf16(fmaxf(fmaf(q[0], s1r[et][0], t1r[et][0]), 0.f), fmaxf(fmaf(q[1], s1r[et][1], t1r[et][1]), 0.f));
                    ar.y = pack_bf16(fmaxf(fmaf(q[2], s1r[et][2], t1r[et][2]), 0.f), fmaxf(fmaf(q[3], s1r[et][3], t1r[et][3]), 0.f));
                    *(uint2*)(a1p + poff[i] + et * 8 + lg * 2) = ar;
                }
            }
        }
        __syncthreads();
        // ---- depthwise: one output row of one channel pair per thread
        if (task && t_im < nimg) {
            float2 acc[WW];
#pragma unroll
            for (int j = 0; j < WW; ++j) acc[j] = bd;
            const uint32_t* rowp = a1p + ((t_im * RH + t_row) * RW) * 16 + pair;
#pragma unroll
            for (int dr = 0; dr < KS; ++dr) {
                float2 in[WW + 2 * P];
#pragma unroll
                for (int c = 0; c < WW + 2 * P; ++c) {
                    const uint32_t u = rowp[(dr * RW + c) * 16];
                    in[c] = make_float2(bf_lo(u), bf_hi(u));
                }
#pragma unroll
                for (int j = 0; j < WW; ++j)
#pragma unroll
                    for (int dc = 0; dc < KS; ++dc) {
                        acc[j].x = fmaf(wt[dr * KS + dc].x, in[j + dc].x, acc[j].x);
                        acc[j].y = fmaf(wt[dr * KS + dc].y, in[j + dc].y, acc[j].y);
                    }
            }
            uint16_t* yo = a.y2 + ((size_t)(n0 + t_im) * a.HW + t_row * a.W) * a.E + e0 + pair * 2;
#pragma unroll
            for (int j = 0; j < WW; ++j) {                          // W == WW (dispatch)
                st1.x += acc[j].x; st1.y += acc[j].y;
                st2.x = fmaf(acc[j].x, acc[j].x, st2.x); st2.y = fmaf(acc[j].y, acc[j].y, st2.y);
                *(uint32_t*)(yo + (size_t)j * a.E) = pack_bf16(acc[j].x, acc[j].y);
            }
        }
        __syncthreads();                                            // image consumed before the next pass overwrites it
    }
    if (a.stats) {
        // deterministic: every task slot parks its sums, 64 threads add the 16 slots in order
        lds_red[(slot * 2 + 0) * 32 + pair * 2] = st1.x; lds_red[(slot * 2 + 0) * 32 + pair * 2 + 1] = st1.y;
        lds_red[(slot * 2 + 1) * 32 + pair * 2] = st2.x; lds_red[(slot * 2 + 1) * 32 + pair * 2 + 1] = st2.y;
        __syncthreads();
        if (tid < 64) {
            const int r = tid >> 5, c = tid & 31;
            float v = 0.f;
            for (int s = 0; s < 16; ++s) v += lds_red[(s * 2 + r) * 32 + c];
            a.stats[((size_t)r * a.E + e0 + c) * gridDim.y + blockIdx.y] = v;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// Shapes the fused block kernels take: a whole image (two 7x7 images) is one pass: H*W*NI <= 208 pixels, W in {14, 7} (a
// thread keeps a full output row in registers), H <= 16/NI task rows, C <= 192 and C % 8 == 0, E % 32 == 0, k in {3, 5}.
static bool irb_shape_ok(int N, int H, int W, int C, int E, int k, int* ni) {
    if (N < 1 || (k != 3 && k != 5) || (C & 7) || C < 16 || C > 192 || (E & 31) || E < 32) return false;
    if (W == 14 && H >= 1 && H <= 14) { *ni = 1; return true; }
    if (W == 7 && H >= 1 && H <= 7) { *ni = 2; return true; }
    return false;
}
extern "C" int mnas_irb_supported(int N, int H, int W, int C, int E, int k) {
    int ni;
    return irb_shape_ok(N, H, W, C, E, k, &ni) ? 1 : 0;
}
// number of image groups (= columns of the statistics table, grid.y) the forward launch uses for a requested upper bound
extern "C" int mnas_irb_fwd_parts(int N, int H, int W, int C, int E, int k, int want) {
    int ni;
    if (!irb_shape_ok(N, H, W, C, E, k, &ni)) return -1;
    const int npass = (N + ni - 1) / ni;
    if (want < 1) want = 1;
    if (want > npass) want = npass;
    const int ipg = (npass + want - 1) / want;
    return (npass + ipg - 1) / ipg;
}

extern "C" int mnas_irb_fwd(const MnasIrbFwd* c, void* stream) {
    int ni;
    if (!c || !irb_shape_ok(c->N, c->H, c->W, c->C, c->E, c->k, &ni)) return MNAS_EINVAL;
    if (!c->x.data || !c->w1 || !c->bn1 || !c->wdw || !c->y2 || c->nparts < 1) return MNAS_EINVAL;
    if ((c->x.scale == nullptr) != (c->x.shift == nullptr)) return MNAS_EINVAL;
    IrbFwdArgs a;
    a.N = c->N; a.H = c->H; a.W = c->W; a.C = c->C; a.E = c->E;
    a.HW = c->H * c->W; a.NI = ni;
    a.Kpad = (c->C + 31) / 32 * 32;
    a.npt = (a.NI * a.HW + 15) / 16;
    const int npass = (c->N + ni - 1) / ni;
    const int groups = mnas_irb_fwd_parts(c->N, c->H, c->W, c->C, c->E, c->k, c->nparts);
    if (groups != c->nparts) return MNAS_EINVAL;                     // the caller sizes the statistics table with mnas_irb_fwd_parts
    a.ipg = (npass + groups - 1) / groups;
    a.x = c->x; a.w1 = (const uint16_t*)c->w1; a.b1 = c->b1; a.bn1 = c->bn1; a.wdw = c->wdw; a.bdw = c->bdw;
    a.y1 = (uint16_t*)c->y1; a.y2 = (uint16_t*)c->y2; a.stats = c->stats;
    const int P = c->k / 2;
    const int RW = (c->W + 2 * P) | 1, RH = c->H + 2 * P;
    const size_t lds = (size_t)a.NI * RH * RW * 16 * 4 + (size_t)2 * a.Kpad * 4 + (size_t)16 * 2 * 32 * 4;
    const dim3 grid(c->E / 32, groups);
    hipStream_t s = (hipStream_t)stream;
    const int kst = a.Kpad <= 96 ? 3 : 6;
#define MNAS_IRB_FWD(K_, W_, T_) \
    if (c->k == K_ && c->W == W_ && kst == T_) { hipLaunchKernelGGL((k_irb_fwd<K_, W_, T_>), grid, dim3(256), lds, s, a); MNAS_CHECK_LAUNCH(); return MNAS_OK; }
    MNAS_IRB_FWD(3, 14, 3) MNAS_IRB_FWD(5, 14, 3) MNAS_IRB_FWD(3, 7, 3) MNAS_IRB_FWD(5, 7, 3)
    MNAS_IRB_FWD(3, 14, 6) MNAS_IRB_FWD(5, 14, 6) MNAS_IRB_FWD(3, 7, 6) MNAS_IRB_FWD(5, 7, 6)
#undef MNAS_IRB_FWD
    return MNAS_EINVAL;
}
