// Input gradient of the stride-2 dense 3x3 convs (mnasnet.py:157-161 with reduce=True) as a transposed convolution, WEIGHT-STATIONARY
// and barrier-free.  Same contract and weight packing as k_tconv (csrc/mnas_tconv.hip, which it takes over from behind
// mnas_tconv_dgrad):
//     gin[n, 2i+ph, 2j+pw, ci] = sum over the taps of parity class (ph, pw) of dy[n, i+dh, j+dw, :] . W[:, ci, kh, kw]
// over a MATERIALISED dy; the four classes of the 2x2 output block (2i.., 2j..) read the dy pixels d00 = dy(i,j), d01 = dy(i,j+1),
// d10 = dy(i+1,j), d11 = dy(i+1,j+1):  class (0,0) needs d00; (0,1) d00, d01; (1,0) d00, d10; (1,1) all four.
//
// k_tconv / k_igemm's parity form walk LDS phases (DMA'd dy tile, weights, out-stage; a barrier each): 110 us for 24 -> 16 at 112x112
// (1.3 TB/s), 79 us for 40 -> 24 at 56x56 -- chains of exposed latencies on 50-250 MB of traffic.  Here (105 / 52 us) a
// workgroup is FOUR WAVES = THE FOUR PARITY CLASSES:
//   * a wave keeps the weight rows of its class -- [Ci][its 1 / 2 / 2 / 4 neighbours x Co], compact K, as MFMA A fragments -- in
//     registers for the whole kernel (12-120 VGPRs);
//   * workgroups walk groups of 16 super-pixels (i, j) persistently; a wave loads the dy fragments of the neighbours its class needs
//     straight from global memory to registers, one group ahead (8 consecutive dy channels of one pixel = 16 contiguous bytes; the
//     four waves hit the same lines in L1);
//   * its [16 pixels][Ci] result leaves through a wave-private LDS stage as 16-byte stores to the class's pixel of each 2x2 block,
//     with the fused BatchNorm-backward reduce of the layer below on the same 16-byte pattern (its operand red_y loaded with the
//     stores' addresses);
//   * no barrier inside the loop, none between the classes.
// Roofline: HBM (gin written once, red_y read once, dy read once + L1/L2 re-reads by the neighbouring classes).
#include "mnas_common.h"

struct TcxArgs {
    int M2;                  // super-pixels = N * Ho * Wo (dy pixels)
    int Ho, Wo, Co, Ci;      // dy plane / channels, gin channels; gin plane = 2Ho x 2Wo
    int Kpad;                // row length of the packed weights: 4*Co rounded up to 32
    int rows_pad;            // 4*Ci rounded up to 16
    int cib;                 // result channels per workgroup block (grid.y blocks of cib channels; multiple of 8)
    float rcp_hw, rcp_wo;
    const uint16_t* dy;      // bf16 (N,Ho,Wo,Co), materialised
    const uint16_t* w;       // MNAS_PACK_TCONV: [rows_pad][Kpad], row = class*Ci + ci, column = neighbour*Co + co
    void* out;               // bf16 (N,2Ho,2Wo,Ci)
    float* stats;            // fused reduce partials [2][Ci][gridDim.x] or NULL
    const void* red_y;
    const float* red_bn;
};

__device__ __forceinline__ int tcx_fdiv(int n, int d, float rcp) {
    if (rcp == 0.f) return n / d;
    int q = (int)((float)n * rcp);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// TPW: 16-row tiles covering the block's result channels; KSM: k-steps of the four-neighbour class (compact K = 4*Co)
template <int TPW, int KSM>
__global__ __launch_bounds__(256) void k_tcx(TcxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WC = TPW * 16, SP = WC + 8;                      // stage row pitch (bf16 elements)
    uint16_t* stage = (uint16_t*)smem;                             // [4][16][SP]
    float* lds_rc = (float*)(stage + 4 * 16 * SP);                 // [4][WC] reduce coefficients (s, t, invstd, -mean*invstd)
    float* lds_fin = lds_rc + 4 * WC;                              // end of kernel: [256][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int cls = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = parity class (ph, pw)
    const int ph = cls >> 1, pw = cls & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int ci0 = blockIdx.y * a.cib;                            // first result channel of this block
    const int cin = min(a.cib, a.Ci - ci0);                        // valid channels (multiple of 8)
    const bool do_red = a.red_y != nullptr;
    const int nslots = (1 + ph) * (1 + pw);                        // neighbours this class reads
    const int kc = nslots * a.Co;                                  // compact K of the class

    if (do_red)
        for (int i = tid; i < 4 * WC; i += 256) {
            const int r = i / WC, c = i - r * WC, cc = ci0 + c;
            float v = 0.f;
            if (c < cin) {
                if (r == 0) v = a.red_bn[cc];
                else if (r == 1) v = a.red_bn[a.Ci + cc];
                else if (r == 2) v = a.red_bn[6 * a.Ci + cc];
                else v = -a.red_bn[5 * a.Ci + cc] * a.red_bn[6 * a.Ci + cc];
            }
            lds_rc[i] = v;
        }
    // ---- this lane's k positions: k-step ks covers compact k = ks*32 + lg*8 .. +7 = 8 channels co.. of neighbour slot `slot`;
    // slot -> neighbour (dh, dw): class (0,1) reads (0,0),(0,1); (1,0) reads (0,0),(1,0); (1,1) all four in packing order
    int koff[KSM];                                                 // pixel / channel offset into dy, -1: past the class's K
    unsigned nbits = 0;                                            // bit 2*ks: dh, bit 2*ks+1: dw
    bf16x8_t wf[TPW][KSM];
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) {
        const int k = ks * 32 + lg * 8;
        const int slot = k / a.Co, co = k - slot * a.Co;
        const bool kok = k < kc;
        int dh = 0, dw = 0;
        if (kok) {
            if (ph && pw) { dh = slot >> 1; dw = slot & 1; }
            else if (ph) dh = slot;
            else if (pw) dw = slot;
        }
        koff[ks] = kok ? (dh * a.Wo + dw) * a.Co + co : -1;
        nbits |= (unsigned)(dh | (dw << 1)) << (2 * ks);
        const int col = (dh * 2 + dw) * a.Co + co;                 // column in the packed matrix
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int rl = t * 16 + l15;                           // channel within the block
            const int row = cls * a.Ci + ci0 + rl;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kok && rl < cin && row < a.rows_pad) v = *(const uint4*)(a.w + (size_t)row * a.Kpad + col);
            wf[t][ks] = *(const bf16x8_t*)&v;
        }
    }
    if (do_red) __syncthreads();                                   // coefficient table visible (the only barrier before the end)

    uint16_t* st = stage + cls * 16 * SP;
    const int hw = a.Ho * a.Wo;
    const int ngroups = (a.M2 + 15) >> 4;
    const int cpp = cin >> 3;                                      // 16-byte chunks per output pixel of this block
    constexpr int NPASS = (16 * (WC / 8) + 63) / 64;
    float r1[NPASS][8], r2[NPASS][8];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) { r1[p][j] = 0.f; r2[p][j] = 0.f; }

    uint4 v0[KSM];
    // copy-out role of this lane in pass p: chunk q = p*64 + lane -> (pixel px of the group, 16-byte channel chunk ch); the
    // output offset (elements; -1: nothing) and the fused reduce's operand are fetched WITH the group's dy fragments, one group
    // ahead: loaded in the copy-out itself, the operand's latency was exposed once per group and wave (122 vs 110 us at 112x112)
    int ooff[NPASS], ooff_n[NPASS];
    uint4 yv[NPASS], yv_n[NPASS];
    auto issue = [&](int g) {
        const int m = g * 16 + l15;
        const bool mok = m < a.M2;
        const int n = mok ? tcx_fdiv(m, hw, a.rcp_hw) : 0, rem = m - n * hw;
        const int i = mok ? tcx_fdiv(rem, a.Wo, a.rcp_wo) : 0, j = rem - i * a.Wo;
        const uint16_t* base = a.dy + (size_t)(mok ? m : 0) * a.Co;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            const int dh = (nbits >> (2 * ks)) & 1, dw = (nbits >> (2 * ks + 1)) & 1;
            v0[ks] = make_uint4(0, 0, 0, 0);
            if (mok && koff[ks] >= 0 && i + dh < a.Ho && j + dw < a.Wo) v0[ks] = *(const uint4*)(base + koff[ks]);
        }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int q = p * 64 + lane;
            const int px = q / cpp, ch = q - px * cpp;
            const int mm = g * 16 + px;
            ooff_n[p] = -1;
            yv_n[p] = make_uint4(0, 0, 0, 0);
            if (q < 16 * cpp && mm < a.M2) {
                const int n2 = tcx_fdiv(mm, hw, a.rcp_hw), rem2 = mm - n2 * hw;
                const int i2 = tcx_fdiv(rem2, a.Wo, a.rcp_wo), j2 = rem2 - i2 * a.Wo;
                ooff_n[p] = ((n2 * 2 * a.Ho + 2 * i2 + ph) * (2 * a.Wo) + 2 * j2 + pw) * a.Ci + ci0 + ch * 8;
                if (do_red) yv_n[p] = *(const uint4*)((const uint16_t*)a.red_y + ooff_n[p]);
            }
        }
    };
    if ((int)blockIdx.x < ngroups) issue(blockIdx.x);
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int m0 = g * 16;
        bf16x8_t bf[KSM];
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) bf[ks] = *(const bf16x8_t*)&v0[ks];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) { ooff[p] = ooff_n[p]; yv[p] = yv_n[p]; }
        if (g + (int)gridDim.x < ngroups) issue(g + gridDim.x);   // next group's fragments fly under the MFMAs / stores
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][ks], bf[ks], acc, 0, 0, 0);
            // lane holds channels ci0 + t*16 + lg*4 + {0..3} of super-pixel m0 + l15
            uint2 pk;
            pk.x = pack_bf16(acc[0], acc[1]);
            pk.y = pack_bf16(acc[2], acc[3]);
            *(uint2*)(st + l15 * SP + t * 16 + lg * 4) = pk;
        }
        __builtin_amdgcn_wave_barrier();                           // (LDS operations of one wave execute in order)
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int q = p * 64 + lane;
            const int px = q / cpp, ch = q - px * cpp;
            if (ooff[p] >= 0) {
                const uint4 pk = *(const uint4*)(st + px * SP + ch * 8);
                st_u4((uint16_t*)a.out + ooff[p], pk, true);
                if (do_red) {
                    const uint32_t gu[4] = {pk.x, pk.y, pk.z, pk.w}, yu[4] = {yv[p].x, yv[p].y, yv[p].z, yv[p].w};
                    const int cl = ch * 8;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        mnas_red2(gu[jj], yu[jj], mnas_ld2(lds_rc + cl + 2 * jj), mnas_ld2(lds_rc + WC + cl + 2 * jj),
                                  mnas_ld2(lds_rc + 2 * WC + cl + 2 * jj), mnas_ld2(lds_rc + 3 * WC + cl + 2 * jj), &r1[p][2 * jj], &r2[p][2 * jj]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (do_red && a.stats) {
        // per-(thread, pass) sums -> per-channel: all threads / passes holding a channel chunk added in a fixed order
        float* outp = a.stats;
        for (int p = 0; p < NPASS; ++p) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) { lds_fin[tid * 16 + j] = r1[p][j]; lds_fin[tid * 16 + 8 + j] = r2[p][j]; }
            __syncthreads();
            for (int i = tid; i < 2 * cin; i += 256) {
                const int r = i / cin, c = i - r * cin, cc = c >> 3, j = c & 7;
                float v = 0.f;
                for (int th = 0; th < 256; ++th)
                    if (((p * 64 + (th & 63)) % cpp) == cc && p * 64 + (th & 63) < 16 * cpp) v += lds_fin[th * 16 + r * 8 + j];
                float* d = outp + ((size_t)r * a.Ci + ci0 + c) * gridDim.x + blockIdx.x;
                *d = (p == 0 ? 0.f : *d) + v;
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct TcxPlan { int tpw, ksm, cib, nblocks; size_t lds; };

int mnas_tcx_enabled() {
    static int on = -1;
    if (on < 0) on = mnas_diag_env("MNAS_TCX", 1);
    return on;
}
static bool tcx_plan(int Ho, int Wo, int Co, int Ci, TcxPlan* p) {
    if (!mnas_tcx_enabled() || (Co & 7) || (Ci & 7) || Co < 8 || Ci < 8 || Ho < 1 || Wo < 1) return false;
    const int ksm = (4 * Co + 31) / 32;
    // instantiated: 24 -> 16 (112x112), 40 -> 24 (56x56), 80 -> 40 (28x28); fragments TPW*KSM*4 VGPRs + 2*KSM*4 of dy in flight
    if (Ci <= 16 && ksm <= 3) { p->tpw = 1; p->ksm = 3; }
    else if (Ci <= 32 && ksm <= 5) { p->tpw = 2; p->ksm = 5; }
    // (80 -> 40 at 28x28, K = 320: measured as three 16-channel blocks over grid.y -- 57 us at best against 47 for k_igemm's parity
    // form -- and as one block with 312 VGPRs: 184 us.  Not instantiated.)
    else return false;
    p->cib = p->tpw * 16 < Ci ? p->tpw * 16 : Ci;
    p->nblocks = (Ci + p->cib - 1) / p->cib;
    const int wc = p->tpw * 16;
    p->lds = (size_t)4 * 16 * (wc + 8) * 2 + (size_t)4 * wc * 4 + (size_t)256 * 16 * 4;
    return true;
}
int mnas_tcx_ok(int Ho, int Wo, int Co, int Ci) {
    TcxPlan p;
    return tcx_plan(Ho, Wo, Co, Ci, &p) ? 1 : 0;
}
int mnas_tcx_parts(int N, int Ho, int Wo, int Co, int Ci) {
    TcxPlan p;
    if (!tcx_plan(Ho, Wo, Co, Ci, &p)) return -1;
    if ((long long)N * Ho * Wo * 4 * Ci > 0x7fffffff) return -1;     // 32-bit element offsets: the caller falls back at BUILD time
    const long long groups = ((long long)N * Ho * Wo + 15) / 16;
    // persistent workgroups: 24 -> 16 at 112x112 105 us with 1024 (148 with 512, 123 with 2048; k_tconv 110); 40 -> 24 at 56x56
    // 52 us with 768 (61 with 512, 66 with 1024, 75 with 2048; k_igemm's parity form 79)
    static int wgs = -1;
    if (wgs < 0) wgs = mnas_diag_env("MNAS_TCX_WGS", 0);
    int want = (wgs > 0 ? wgs : (p.tpw == 1 ? 1024 : 768)) / p.nblocks;
    if (want < 64) want = 64;
    return (int)(groups < want ? groups : want);
}

int mnas_tcx_dgrad(const MnasTconvDgrad* c, void* stream) {
    TcxPlan p;
    const long long M2 = (long long)c->N * c->Ho * c->Wo;
    if (M2 * 4 * c->Ci > 0x7fffffff || !tcx_plan(c->Ho, c->Wo, c->Co, c->Ci, &p)) return MNAS_EINVAL;
    TcxArgs a;
    a.M2 = (int)M2; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co; a.Ci = c->Ci;
    a.Kpad = (4 * c->Co + 31) / 32 * 32; a.rows_pad = (4 * c->Ci + 15) / 16 * 16; a.cib = p.cib;
    a.rcp_hw = M2 < (1 << 24) ? 1.0f / (float)(c->Ho * c->Wo) : 0.f;
    a.rcp_wo = M2 < (1 << 24) ? 1.0f / (float)c->Wo : 0.f;
    a.dy = (const uint16_t*)c->dy; a.w = (const uint16_t*)c->w; a.out = c->out; a.stats = c->stats;
    a.red_y = c->red_y; a.red_bn = c->red_bn;
    hipStream_t s = (hipStream_t)stream;
#define MNAS_TCX(T_, K_) \
    if (p.tpw == T_ && p.ksm == K_) { \
        hipLaunchKernelGGL((k_tcx<T_, K_>), dim3(c->nparts, p.nblocks), dim3(256), p.lds, s, a); \
        MNAS_CHECK_LAUNCH(); \
        return MNAS_OK; \
    }
    MNAS_TCX(1, 3) MNAS_TCX(2, 5)
#undef MNAS_TCX
    return MNAS_EINVAL;
}

// =====================================================================================================================================
// k_tcr: the same transposed convolution for the WEIGHT-HEAVY transition onto the 7x7 plane (96 -> 192 stride 2: its input gradient
// reads dy (N,7,7,192) and writes gin (N,14,14,96); the packed weight matrix is 4*96 x 4*192 x 2 B = 590 KB against 19 KB of dy per
// image).  k_igemm's parity form: 68 us for 14 MB of traffic.  Structure of k_c3r (csrc/mnas_c3r.hip):
//   * a workgroup (8 waves) owns 16 result channels and walks images persistently; the weight rows of ALL FOUR parity classes for
//     those channels are register-resident, each class's compact K (its 1/2/2/4 neighbours x Co) split four ways over the waves
//     (wave w: k-steps (w & 3) + 4j of every class): 14 k-steps = 56 VGPRs for Co = 192;
//   * the image's dy plane is staged once in LDS with a zero row / column at the bottom / right ([Ho+1][Wo+1][Co+8]); the output
//     pixels are taken class-major, each class padded to PC = 64 slots, so a 16-pixel tile belongs to one class and needs only that
//     class's k-steps; the two wave groups (w >> 2) split a class's tiles;
//   * the four K-partials meet in LDS, 512 threads combine them, store 8 bytes per (pixel, 4 channels) to the class's pixel of the
//     2x2 block and run the fused BatchNorm-backward reduce (operand prefetched before the MFMA phase).
// =====================================================================================================================================
struct TcrArgs {
    int N, Ho, Wo, Co, Ci;   // dy plane / channels, gin channels
    int Kpad, rows_pad;
    int LW, Cp;              // LDS image: (Ho+1) rows x LW = Wo+1 pixels x Cp = Co+8 elements
    const uint16_t* dy;
    const uint16_t* w;
    void* out;
    float* stats;
    const void* red_y;
    const float* red_bn;
};

// KQ0..KQ3: per-wave k-step counts (upper bounds) of the classes (0,0), (0,1), (1,0), (1,1); PC = 64 pixel slots per class
template <int KQ0, int KQ1, int KQ3>
__global__ __launch_bounds__(512) void k_tcr(TcrArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NB = 16, PPW = NB + 4, NSLOT = 256, MAXS = 6;
    constexpr int KQ[4] = {KQ0, KQ1, KQ1, KQ3};
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wave & 3, phalf = wave >> 2;
    const int l15 = lane & 15, lg = lane >> 4;
    const int img_elems = ((a.Ho + 1) * a.LW * a.Cp + 7) & ~7;
    uint16_t* img = (uint16_t*)smem;                               // [img_elems]
    float* part = (float*)(img + img_elems);                       // [4][NSLOT][PPW]
    float* lds_rc = part + 4 * NSLOT * PPW;                        // [4][NB]
    float* lds_fin = part;                                         // end of kernel: [128][2][NB]
    const int ci0 = blockIdx.y * NB;
    const bool do_red = a.red_y != nullptr;
    const int npc = a.Ho * a.Wo;                                   // pixels per class

    for (int i = tid; i < img_elems >> 3; i += 512) ((uint4*)img)[i] = make_uint4(0, 0, 0, 0);           // zero border (and interior)
    if (do_red)
        for (int i = tid; i < 4 * NB; i += 512) {
            const int r = i / NB, cc = ci0 + i % NB;
            float v = 0.f;
            if (cc < a.Ci) {
                if (r == 0) v = a.red_bn[cc];
                else if (r == 1) v = a.red_bn[a.Ci + cc];
                else if (r == 2) v = a.red_bn[6 * a.Ci + cc];
                else v = -a.red_bn[5 * a.Ci + cc] * a.red_bn[6 * a.Ci + cc];
            }
            lds_rc[i] = v;
        }
    // ---- weight fragments: class c, this wave's k-steps ks = kq + 4j of the class's compact K; LDS offset of the lane's 8 channels
    bf16x8_t wf0[KQ0], wf1[KQ1], wf2[KQ1], wf3[KQ3];
    int to0[KQ0], to1[KQ1], to2[KQ1], to3[KQ3];
    auto load_class = [&](auto& wf, auto& to, int cls, int kqn) {
        const int ph = cls >> 1, pw = cls & 1;
        const int kc = (1 + ph) * (1 + pw) * a.Co;
#pragma unroll
        for (int j = 0; j < kqn; ++j) {
            const int k = (kq + 4 * j) * 32 + lg * 8;
            const bool kok = k < kc;
            const int slot = kok ? k / a.Co : 0, co = kok ? k - slot * a.Co : 0;
            int dh = 0, dw = 0;
            if (ph && pw) { dh = slot >> 1; dw = slot & 1; }
            else if (ph) dh = slot;
            else if (pw) dw = slot;
            to[j] = kok ? (dh * a.LW + dw) * a.Cp + co : -1;
            const int row = cls * a.Ci + ci0 + l15, col = (dh * 2 + dw) * a.Co + co;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kok && ci0 + l15 < a.Ci && row < a.rows_pad) v = *(const uint4*)(a.w + (size_t)row * a.Kpad + col);
            wf[j] = *(const bf16x8_t*)&v;
        }
    };
    load_class(wf0, to0, 0, KQ0); load_class(wf1, to1, 1, KQ1); load_class(wf2, to2, 2, KQ1); load_class(wf3, to3, 3, KQ3);
    // ---- this lane's pixels: tile t (0, 1) of its wave group within a class -> super-pixel p = (phalf*2 + t)*16 + l15
    int pbase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = (phalf * 2 + t) * 16 + l15;
        const int pp = p < npc ? p : 0;
        const int i = pp / a.Wo, j = pp - i * a.Wo;
        pbase[t] = (i * a.LW + j) * a.Cp;
    }
    // ---- epilogue role: thread -> 4 channels c4 of slots srow + 128*e
    const int c4 = tid & 3, srow = tid >> 2;
    const int coe = ci0 + c4 * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int co8 = a.Co >> 3, in_slots = npc * co8;
    uint4 vimg[MAXS];
    auto fetch = [&](int n) {
        const uint16_t* src = a.dy + (size_t)n * npc * a.Co;
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            const int q = tid + 512 * j;
            vimg[j] = make_uint4(0, 0, 0, 0);
            if (q < in_slots) vimg[j] = *(const uint4*)(src + (size_t)q * 8);
        }
    };
    auto place = [&]() {
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            const int q = tid + 512 * j;
            if (q >= in_slots) continue;
            const int pixq = q / co8, c8 = q - pixq * co8;
            const int iy = pixq / a.Wo, ix = pixq - iy * a.Wo;
            *(uint4*)(img + (iy * a.LW + ix) * a.Cp + c8 * 8) = vimg[j];
        }
    };
    __syncthreads();
    if ((int)blockIdx.x < a.N) { fetch(blockIdx.x); place(); }
    for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
        __syncthreads();                                           // image n published; partials of image n-1 consumed
        const int nn = n + gridDim.x;
        if (nn < a.N) fetch(nn);
        // epilogue operands (output offset, reduce operand) of this thread's two slots, fetched under the MFMA phase
        int ooff[2];
        uint2 ypre[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int slot = srow + 128 * e, cls = slot >> 6, p = slot & 63;
            ooff[e] = -1; ypre[e] = make_uint2(0, 0);
            if (p < npc && coe < a.Ci) {
                const int i = p / a.Wo, j = p - i * a.Wo;
                ooff[e] = (((n * 2 * a.Ho) + 2 * i + (cls >> 1)) * (2 * a.Wo) + 2 * j + (cls & 1)) * a.Ci + coe;
                if (do_red) ypre[e] = *(const uint2*)((const uint16_t*)a.red_y + ooff[e]);
            }
        }
        f32x4_t acc[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[c][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        auto run_class = [&](const auto& wf, const auto& to, int cls, int kqn, f32x4_t (&ac)[2]) {
#pragma unroll
            for (int j = 0; j < kqn; ++j) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    uint4 bv = make_uint4(0, 0, 0, 0);
                    if (to[j] >= 0) bv = *(const uint4*)(img + pbase[t] + to[j]);
                    ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], *(const bf16x8_t*)&bv, ac[t], 0, 0, 0);
                }
            }
        };
        run_class(wf0, to0, 0, KQ0, acc[0]); run_class(wf1, to1, 1, KQ1, acc[1]);
        run_class(wf2, to2, 2, KQ1, acc[2]); run_class(wf3, to3, 3, KQ3, acc[3]);
        // ---- park the K-partials: part[kq][slot = cls*64 + (phalf*2 + t)*16 + l15][channel lg*4 ..]
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                *(float4*)(part + ((size_t)kq * NSLOT + c * 64 + (phalf * 2 + t) * 16 + l15) * PPW + lg * 4) = *(const float4*)&acc[c][t];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (ooff[e] < 0) continue;
            const int slot = srow + 128 * e;
            const float* src = part + (size_t)slot * PPW + c4 * 4;
            float4 v = *(const float4*)src;
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const float4 x = *(const float4*)(src + (size_t)q * NSLOT * PPW);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            uint2 pk;
            pk.x = pack_bf16(v.x, v.y);
            pk.y = pack_bf16(v.z, v.w);
            *(uint2*)((uint16_t*)a.out + ooff[e]) = pk;
            if (do_red) {
                const int cl = c4 * 4;
                mnas_red2(pk.x, ypre[e].x, mnas_ld2(lds_rc + cl), mnas_ld2(lds_rc + NB + cl), mnas_ld2(lds_rc + 2 * NB + cl),
                          mnas_ld2(lds_rc + 3 * NB + cl), s1, s2);
                mnas_red2(pk.y, ypre[e].y, mnas_ld2(lds_rc + cl + 2), mnas_ld2(lds_rc + NB + cl + 2), mnas_ld2(lds_rc + 2 * NB + cl + 2),
                          mnas_ld2(lds_rc + 3 * NB + cl + 2), s1 + 2, s2 + 2);
            }
        }
        if (nn < a.N) place();                                     // all MFMA reads of the tile finished before the barrier above
    }
    if (do_red && a.stats) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            lds_fin[(srow * 2 + 0) * NB + c4 * 4 + r] = s1[r];
            lds_fin[(srow * 2 + 1) * NB + c4 * 4 + r] = s2[r];
        }
        __syncthreads();
        for (int i = tid; i < 2 * NB; i += 512) {
            const int r = i / NB, cl = i - r * NB, cc = ci0 + cl;
            float v = lds_fin[r * NB + cl];
            for (int q = 1; q < 128; ++q) v += lds_fin[(q * 2 + r) * NB + cl];
            if (cc < a.Ci) a.stats[((size_t)r * a.Ci + cc) * gridDim.x + blockIdx.x] = v;
        }
    }
}

static bool tcr_plan(int Ho, int Wo, int Co, int Ci, int* parts_max, size_t* lds) {
    if (!mnas_tcx_enabled() || (Co & 31) || (Ci & 15) || Ho * Wo > 64 || Ho < 1 || Wo < 1) return false;
    if (Co != 192) return false;                                   // instantiated for 192 dy channels (k-steps per wave: 2, 3, 3, 6)
    if ((long long)Co * Ci < 16 * 1024) return false;
    const size_t img = (((size_t)(Ho + 1) * (Wo + 1) * (Co + 8) + 7) & ~(size_t)7) * 2;
    *lds = img + (size_t)4 * 256 * 20 * 4 + (size_t)4 * 16 * 4;
    if ((size_t)Ho * Wo * (Co / 8) > 512 * 6 || *lds > 160 * 1024) return false;
    const int slices = Ci / 16;
    *parts_max = 256 / slices < 1 ? 1 : 256 / slices;
    return true;
}
int mnas_tcr_ok(int Ho, int Wo, int Co, int Ci) {
    int pm; size_t lds;
    return tcr_plan(Ho, Wo, Co, Ci, &pm, &lds) ? 1 : 0;
}
int mnas_tcr_parts(int N, int Ho, int Wo, int Co, int Ci) {
    int pm; size_t lds;
    if (N < 32 || !tcr_plan(Ho, Wo, Co, Ci, &pm, &lds)) return -1;
    if ((long long)N * 4 * Ho * Wo * Ci > 0x7fffffff) return -1;     // 32-bit element offsets (as mnas_tcr_dgrad checks)
    return pm < N ? pm : N;
}
int mnas_tcr_dgrad(const MnasTconvDgrad* c, void* stream) {
    int pm; size_t lds;
    if (c->N < 32 || !tcr_plan(c->Ho, c->Wo, c->Co, c->Ci, &pm, &lds)) return MNAS_EINVAL;
    if ((long long)c->N * 4 * c->Ho * c->Wo * c->Ci > 0x7fffffff) return MNAS_EINVAL;
    TcrArgs a;
    a.N = c->N; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co; a.Ci = c->Ci;
    a.Kpad = (4 * c->Co + 31) / 32 * 32; a.rows_pad = (4 * c->Ci + 15) / 16 * 16;
    a.LW = c->Wo + 1; a.Cp = c->Co + 8;
    a.dy = (const uint16_t*)c->dy; a.w = (const uint16_t*)c->w; a.out = c->out; a.stats = c->stats;
    a.red_y = c->red_y; a.red_bn = c->red_bn;
    hipLaunchKernelGGL((k_tcr<2, 3, 6>), dim3(c->nparts, c->Ci / 16), dim3(512), lds, (hipStream_t)stream, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
