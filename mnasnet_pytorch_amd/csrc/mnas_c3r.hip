// Dense 3x3 convolution (pad 1) onto the 7x7 feature maps with the WEIGHTS REGISTER-RESIDENT: forward (stride 1 / 2) and
// stride-1 input gradient of ConvBlock(kernel_size=3) where the weight block is the big operand (192 -> 320 at 7x7: 1.1 MB of
// bf16 weights against 19 KB of activations per image; 96 -> 192 stride 2; mnasnet.py:48-62, the stage transitions :160-161).
// Behind mnas_conv_gemm, in front of k_dimg (csrc/mnas_dimg.hip), which streams the whole weight block through LDS once per
// pair of images: 128 passes x 1.1 MB from L2, 72 us forward / 92 us input gradient for 14 GFLOP.
//
// Here a workgroup (8 waves) owns a SLICE of NT*16 output channels and walks images persistently:
//   * its [NT*16][9*Ci] weight slice lives in registers for the whole kernel, the reduction dimension split four ways over the
//     waves (wave w: k-steps (w & 3) + 4j) as MFMA A fragments -- loaded once per workgroup, never again;
//   * the two wave groups (w >> 2) split the 16-pixel tiles of the output plane;
//   * an image is staged once in LDS as the zero-bordered activated tile ([Hi+2][Wi+2][Ci+8] bf16, as k_dimg): a B fragment is
//     one 16-byte LDS read per lane (8 consecutive input channels of one tap; k = tap*Ci + ci, the packed-weight order); the
//     next image's chunks are fetched to registers before the MFMA phase and written to the other tile after it;
//   * the four K-partials of every (pixel, cout) meet in LDS (one barrier), 512 threads add them in wave order, apply the
//     epilogue (bias + BatchNorm partial statistics, or the fused BatchNorm-backward reduce) and store 8 bytes each.
// MODE 0: forward.  MODE 1: input gradient = the same correlation with mirrored taps over the materialised dy and the
// [Ci][tap*Co+co] packing (as k_dimg MODE 1).  Measured in the bs-256 step (same call as k_dimg / k_igemm): 192 -> 320 forward
// 72 -> 46 us, its input gradient 92 -> 76 us, 96 -> 192 stride 2 forward 33 -> 29 us.  Roofline: neither HBM nor MFMA peak --
// MFMA issue of two waves per SIMD plus the LDS partial exchange; the memory system sees every tensor once.
#include "mnas_common.h"

struct C3rArgs {
    int N, Hi, Wi, Ci, Ho, Wo, Co;   // Ci = reduction channels (dy channels in MODE 1), Co = result channels
    int stride;                      // 1, or 2 (MODE 0 only)
    int Kpad, ksteps;                // 9*Ci rounded up to 32, k-steps of 32
    int LW, Cp;                      // LDS image: (Hi+2) rows x LW = Wi+2 pixels x Cp = Ci+8 elements
    int co_pad16;
    MnasActIn act;                   // input (MODE 1: the materialised dy, no coefficients)
    const uint16_t* w;               // [co_pad16][Kpad]
    const float* bias;
    void* out;
    float* stats;                    // [2][Co][gridDim.x]
    const void* red_y;
    const float* red_bn;
};

// KQ (round 6): ways the reduction dimension is split over the 8 waves.  4: two wave groups split the four pixel tiles (PTW = 2).
// 8: every wave takes an eighth of K for all four tiles (PTW = 4) -- the input gradient of 192 -> 320 (K = 2880) then keeps TWO cout
// tiles of weights in registers (2 x 12 fragments) where the 4-way split could hold one (24): a B fragment read from LDS feeds two
// MFMAs, the launch needs half as many passes over the images (6 channel slices instead of 12); the image tile is single-buffered
// (the next image is written after the barrier that ends the MFMA phase anyway) to make room for the eight partial tiles.
template <int MODE, int NT, int KSW, int PTW, int KQ = 4>
__global__ __launch_bounds__(512) void k_c3r(C3rArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert((8 / KQ) * PTW == 4, "four 16-pixel tiles per image");
    constexpr int NB = NT * 16, NCH = NB / 4, PROWS = 512 / NCH, PT = (8 / KQ) * PTW, PP = NB + 4;
    constexpr int NIMG = KQ == 8 ? 1 : 2;                          // image tiles in LDS
    constexpr int MAXS = 6;                                        // image chunks (16 B) per thread
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wave % KQ, ph = wave / KQ;
    const int l15 = lane & 15, lg = lane >> 4;
    const int img_elems = ((a.Hi + 2) * a.LW * a.Cp + 7) & ~7;
    uint16_t* img = (uint16_t*)smem;                               // [NIMG][img_elems]
    float* part = (float*)(img + NIMG * img_elems);                // [KQ][PT*16][PP]
    float* lds_coef = part + KQ * PT * 16 * PP;                    // [2][Ci] act-on-load scale / shift
    float* lds_rc = lds_coef + 2 * a.Ci;                           // MODE 1: [4][NB] reduce coefficients
    float* lds_fin = part;                                         // end of kernel: [PROWS][2][NB]
    const int co0 = blockIdx.y * NB;
    const bool has_coef = MODE == 0 && a.act.scale != nullptr;
    const bool do_red = MODE == 1 && a.red_y != nullptr;
    const int npix = a.Ho * a.Wo;

    for (int i = tid; i < 2 * a.Ci; i += 512)
        lds_coef[i] = has_coef ? (i < a.Ci ? a.act.scale[i] : a.act.shift[i - a.Ci]) : 0.f;
    for (int i = tid; i < (NIMG * img_elems) >> 3; i += 512) ((uint4*)img)[i] = make_uint4(0, 0, 0, 0);  // zero borders (and interiors)
    if (do_red)
        for (int i = tid; i < 4 * NB; i += 512) {
            const int r = i / NB, cc = co0 + i % NB;
            float v = 0.f;
            if (cc < a.Co) {
                if (r == 0) v = a.red_bn[cc];
                else if (r == 1) v = a.red_bn[a.Co + cc];
                else if (r == 2) v = a.red_bn[6 * a.Co + cc];
                else v = -a.red_bn[5 * a.Co + cc] * a.red_bn[6 * a.Co + cc];
            }
            lds_rc[i] = v;
        }
    // ---- this wave's quarter of the weight slice: A fragments [cout l15][k = ks*32 + lg*8 ..], k-steps ks = kq + 4j
    bf16x8_t wf[NT][KSW];
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        const int ks = kq + KQ * j;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int row = co0 + nt * 16 + l15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ks < a.ksteps && row < a.co_pad16) v = *(const uint4*)(a.w + (size_t)row * a.Kpad + ks * 32 + lg * 8);
            wf[nt][j] = *(const bf16x8_t*)&v;
        }
    }
    // ---- this lane's output pixels (one per 16-pixel tile of its wave group) -> base offset in the LDS image.  Pixels past the
    // plane compute on whatever (0, 0) + tap holds; they are never stored or counted.
    int pbase[PTW];
#pragma unroll
    for (int t = 0; t < PTW; ++t) {
        const int p = (ph * PTW + t) * 16 + l15;
        const int pp = p < npix ? p : 0;
        const int oy = pp / a.Wo, ox = pp - oy * a.Wo;
        pbase[t] = (oy * a.stride * a.LW + ox * a.stride) * a.Cp;
    }
    // ---- epilogue role: thread -> 4 consecutive couts c4 of pixel rows prow + PROWS*i, fixed for the whole kernel
    const int c4 = tid % NCH, prow = tid / NCH;
    const bool epi_ok = prow < PROWS;                              // (NB = 48: 12 chunks x 42 rows = 504 of the 512 threads)
    const int coe = co0 + c4 * 4;
    float bias4[4], s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bias4[r] = (MODE == 0 && a.bias && coe + r < a.Co) ? a.bias[coe + r] : 0.f;
        s1[r] = 0.f; s2[r] = 0.f;
    }

    // ---- image staging: 16-byte chunks (pixel, 8 channels), contiguous in NHWC
    const int ci8 = a.Ci >> 3;
    const int in_slots = a.Hi * a.Wi * ci8;
    uint4 vimg[MAXS];
    auto fetch = [&](int n) {
        const uint16_t* src = (const uint16_t*)a.act.data + (size_t)n * a.Hi * a.Wi * a.Ci;
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            const int q = tid + 512 * j;
            vimg[j] = make_uint4(0, 0, 0, 0);
            if (q < in_slots) vimg[j] = *(const uint4*)(src + (size_t)q * 8);
        }
    };
    auto place = [&](int buf) {
        uint16_t* dst = img + buf * img_elems;
#pragma unroll
        for (int j = 0; j < MAXS; ++j) {
            const int q = tid + 512 * j;
            if (q >= in_slots) continue;
            const int pixq = q / ci8, c8 = q - pixq * ci8;
            const int iy = pixq / a.Wi, ix = pixq - iy * a.Wi;
            uint4 u = vimg[j];
            if (has_coef) {
                float s[8], t[8];
                *(float4*)&s[0] = *(const float4*)(lds_coef + c8 * 8); *(float4*)&s[4] = *(const float4*)(lds_coef + c8 * 8 + 4);
                *(float4*)&t[0] = *(const float4*)(lds_coef + a.Ci + c8 * 8); *(float4*)&t[4] = *(const float4*)(lds_coef + a.Ci + c8 * 8 + 4);
                u = act8(u, s, t);
            }
            *(uint4*)(dst + ((iy + 1) * a.LW + ix + 1) * a.Cp + c8 * 8) = u;
        }
    };

    __syncthreads();                                               // coefficients / zero fill visible
    if ((int)blockIdx.x < a.N) { fetch(blockIdx.x); place(0); }
    int it = 0;
    for (int n = blockIdx.x; n < a.N; n += gridDim.x, ++it) {
        __syncthreads();                                           // image n published; partials of image n-1 consumed
        const int nn = n + gridDim.x;
        if (nn < a.N) fetch(nn);                                   // lands under the MFMA phase
        // MODE 1: the fused reduce's operand of this thread's epilogue row(s), fetched here: loaded in the epilogue itself its
        // latency was exposed once per image
        constexpr int NEP = (PT * 16 + PROWS - 1) / PROWS;
        uint2 ypre[NEP];
        if (do_red) {
#pragma unroll
            for (int e = 0; e < NEP; ++e) {
                const int p = prow + e * PROWS;
                ypre[e] = make_uint2(0, 0);
                if (epi_ok && p < npix && coe < a.Co) ypre[e] = *(const uint2*)((const uint16_t*)a.red_y + (size_t)n * npix * a.Co + (size_t)p * a.Co + coe);
            }
        }
        const uint16_t* im = img + (NIMG == 2 ? (it & 1) : 0) * img_elems;
        f32x4_t acc[PTW][NT];
#pragma unroll
        for (int t = 0; t < PTW; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const int ks = kq + KQ * j;
            if (ks >= a.ksteps) break;                             // uniform per wave
            // this lane's 8 consecutive k: channels ci0.. of tap `tap` (k = tap*Ci + ci)
            const int k = ks * 32 + lg * 8;
            const int tap = k / a.Ci, ci0 = k - tap * a.Ci;
            int toff = 0;
            bool kok = tap < 9;
            if (kok) {
                const int th = tap / 3, tw = tap - th * 3;
                const int dh = MODE == 1 ? 2 - th : th, dw = MODE == 1 ? 2 - tw : tw;
                toff = (dh * a.LW + dw) * a.Cp + ci0;
            }
#pragma unroll
            for (int t = 0; t < PTW; ++t) {
                uint4 bv = make_uint4(0, 0, 0, 0);
                if (kok) bv = *(const uint4*)(im + pbase[t] + toff);
                const bf16x8_t bfrag = *(const bf16x8_t*)&bv;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][j], bfrag, acc[t][nt], 0, 0, 0);
            }
        }
        // ---- park the K-partials: part[kq][pixel][cout nt*16 + lg*4 ..]
#pragma unroll
        for (int t = 0; t < PTW; ++t) {
            float* dst = part + ((size_t)kq * PT * 16 + (ph * PTW + t) * 16 + l15) * PP + lg * 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) *(float4*)(dst + nt * 16) = *(const float4*)&acc[t][nt];
        }
        __syncthreads();
        // ---- combine + epilogue
        const size_t obase = (size_t)n * npix * a.Co;
#pragma unroll
        for (int e = 0; e < NEP; ++e) {
            const int p = prow + e * PROWS;
            if (!epi_ok || p >= npix || coe >= a.Co) continue;
            const float* src = part + (size_t)p * PP + c4 * 4;
            float4 v = *(const float4*)src;
#pragma unroll
            for (int q = 1; q < KQ; ++q) {
                const float4 x = *(const float4*)(src + (size_t)q * PT * 16 * PP);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            const size_t o = obase + (size_t)p * a.Co + coe;
            if (MODE == 0) {
                const mnas_f2 a0 = {v.x + bias4[0], v.y + bias4[1]}, a1 = {v.z + bias4[2], v.w + bias4[3]};
                mnas_stat2(a0, s1, s2);
                mnas_stat2(a1, s1 + 2, s2 + 2);
                uint2 pk;
                pk.x = pack_bf16(a0.x, a0.y);
                pk.y = pack_bf16(a1.x, a1.y);
                *(uint2*)((uint16_t*)a.out + o) = pk;
            } else {
                uint2 pk;
                pk.x = pack_bf16(v.x, v.y);
                pk.y = pack_bf16(v.z, v.w);
                *(uint2*)((uint16_t*)a.out + o) = pk;
                if (do_red) {
                    const uint2 yv = ypre[e];
                    const int cl = c4 * 4;
                    mnas_red2(pk.x, yv.x, mnas_ld2(lds_rc + cl), mnas_ld2(lds_rc + NB + cl), mnas_ld2(lds_rc + 2 * NB + cl),
                              mnas_ld2(lds_rc + 3 * NB + cl), s1, s2);
                    mnas_red2(pk.y, yv.y, mnas_ld2(lds_rc + cl + 2), mnas_ld2(lds_rc + NB + cl + 2), mnas_ld2(lds_rc + 2 * NB + cl + 2),
                              mnas_ld2(lds_rc + 3 * NB + cl + 2), s1 + 2, s2 + 2);
                }
            }
        }
        if (nn < a.N) place(NIMG == 2 ? ((it + 1) & 1) : 0);       // that tile was last read in an MFMA phase that ended before the barrier above
    }
    if ((MODE == 0 || do_red) && a.stats) {
        // the PROWS pixel-rows of a cout chunk combined in row order (deterministic)
        __syncthreads();
        if (epi_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lds_fin[(prow * 2 + 0) * NB + c4 * 4 + r] = s1[r];
                lds_fin[(prow * 2 + 1) * NB + c4 * 4 + r] = s2[r];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * NB; i += 512) {
            const int r = i / NB, cl = i - r * NB, cc = co0 + cl;
            float v = lds_fin[r * NB + cl];
            for (int q = 1; q < PROWS; ++q) v += lds_fin[(q * 2 + r) * NB + cl];
            if (cc < a.Co) a.stats[((size_t)r * a.Co + cc) * gridDim.x + blockIdx.x] = v;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
struct C3rPlan { int nt, ksw, ptw, kq, slices, parts; size_t lds; };

int mnas_c3r_enabled() {
    static int on = -1;
    if (on < 0) on = mnas_diag_env("MNAS_C3R", 1);
    return on;
}
static bool c3r_plan(int mode, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad, C3rPlan* p) {
    if (!mnas_c3r_enabled() || kh != 3 || kw != 3 || pad != 1 || (stride != 1 && !(stride == 2 && mode == 0))) return false;
    if ((Ci & 7) || (Co & 7) || Ci < 64 || N < 32 || Ho != (Hi - 1) / stride + 1 || Wo != (Wi - 1) / stride + 1) return false;
    const int npix = Ho * Wo;
    if (npix > 64) return false;                                   // 7x7 output planes (four 16-pixel tiles); larger planes stay on k_dimg
    // weight-heavy layers only: the slice a workgroup keeps in registers must be worth more than the partial exchange
    // (192 -> 320 and its input gradient; 96 -> 192 stride 2 forward)
    if ((long long)9 * Ci * Co < 128 * 1024) return false;
    const int ksteps = (9 * Ci + 31) / 32;
    int ksw = (ksteps + 3) / 4;
    p->ptw = 2; p->kq = 4;
    // cout tiles per workgroup: 2 while the fragments fit (NT*KSW*4 registers), else 1 -- or (round 6, MNAS_C3R_KQ8) the
    // reduction split eight ways with two cout tiles (192 -> 320 input gradient: 90 k-steps = 8 x 12)
    p->nt = ksw <= 14 ? 2 : 1;
    if (p->nt == 1 && (ksteps + 7) / 8 <= 12 && mnas_diag_env("MNAS_C3R_KQ8", 1)) {
        p->kq = 8; p->ptw = 4; p->nt = 2; ksw = (ksteps + 7) / 8; p->ksw = 12;
    } else if (p->nt == 2 && (ksteps + 7) / 8 <= 7 && (Co + 47) / 48 < (Co + 31) / 32 && stride == 1 && mnas_diag_env("MNAS_C3R_KQ8", 1) >= 1 &&
               mnas_diag_env("MNAS_C3R_NT3", 1)) {
        // three cout tiles with the 8-way split where that saves passes (192 -> 320 forward: 7 slices of 48 instead of 10 of 32)
        p->kq = 8; p->ptw = 4; p->nt = 3; ksw = (ksteps + 7) / 8; p->ksw = 7;
    } else {
        if (p->nt * ksw > 28 && !(p->nt == 1 && ksw <= 24)) return false;
        p->ksw = ksw <= 14 ? 14 : 24;
    }
    const int nb = p->nt * 16;
    p->slices = (Co + nb - 1) / nb;
    int parts = 256 / p->slices;                                   // one workgroup per CU (139 KB of LDS at most), one round
    if (parts < 1) parts = 1;
    if (parts > N) parts = N;
    p->parts = parts;
    const size_t img = (((size_t)(Hi + 2) * (Wi + 2) * (Ci + 8) + 7) & ~(size_t)7) * 2;
    const size_t partb = (size_t)p->kq * 64 * (nb + 4) * 4;        // [KQ][4 tiles x 16 pixels][NB + 4]
    p->lds = (p->kq == 8 ? 1 : 2) * img + partb + (size_t)2 * Ci * 4 + (size_t)4 * nb * 4;
    const size_t fin = (size_t)(512 / (nb / 4)) * 2 * nb * 4;     // lives in the partial area
    if (fin > partb) return false;
    if ((size_t)Hi * Wi * (Ci / 8) > 512 * 6) return false;        // image chunks per thread
    return p->lds <= 160 * 1024;
}
int mnas_c3r_parts(int mode, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad) {
    C3rPlan p;
    if (!c3r_plan(mode, N, Hi, Wi, Ci, Ho, Wo, Co, kh, kw, stride, pad, &p)) return -1;
    return p.parts;
}

int mnas_c3r_run(const MnasConvGemm* c, void* stream) {
    C3rPlan p;
    if (!c3r_plan(c->mode, c->N, c->Hi, c->Wi, c->Ci, c->Ho, c->Wo, c->Co, c->kh, c->kw, c->stride, c->pad, &p)) return MNAS_EINVAL;
    if (c->mode == 1 && (c->grad.coef || c->grad.y)) return MNAS_EINVAL;          // materialised dy only
    if (c->resid || c->gate || c->nparts < 1) return MNAS_EINVAL;
    C3rArgs a = {};
    a.N = c->N; a.Hi = c->Hi; a.Wi = c->Wi; a.Ci = c->Ci; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co;
    a.stride = c->stride;
    a.Kpad = (9 * c->Ci + 31) / 32 * 32; a.ksteps = a.Kpad / 32;
    a.LW = c->Wi + 2; a.Cp = c->Ci + 8;
    a.co_pad16 = (c->Co + 15) / 16 * 16;
    if (c->mode == 0) a.act = c->act;
    else { a.act.data = c->grad.g; a.act.scale = nullptr; a.act.shift = nullptr; }
    a.w = (const uint16_t*)c->w; a.bias = c->mode == 0 ? c->bias : nullptr; a.out = c->out; a.stats = c->stats;
    a.red_y = c->mode == 1 ? c->red_y : nullptr; a.red_bn = c->red_bn;
    if (a.red_y && (!a.red_bn || !a.stats)) return MNAS_EINVAL;
    const dim3 grid(c->nparts, p.slices);                // workgroups beyond N only write their (zero) statistics column
    hipStream_t s = (hipStream_t)stream;
#define MNAS_C3R(M_, NT_, K_) \
    if (c->mode == M_ && p.kq == 4 && p.nt == NT_ && p.ksw == K_) { \
        hipLaunchKernelGGL((k_c3r<M_, NT_, K_, 2>), grid, dim3(512), p.lds, s, a); \
        MNAS_CHECK_LAUNCH(); \
        return MNAS_OK; \
    }
    MNAS_C3R(0, 2, 14) MNAS_C3R(1, 2, 14) MNAS_C3R(0, 1, 24) MNAS_C3R(1, 1, 24)
#undef MNAS_C3R
    if (p.kq == 8 && p.nt == 3 && p.ksw == 7) {
        if (c->mode == 0) hipLaunchKernelGGL((k_c3r<0, 3, 7, 4, 8>), grid, dim3(512), p.lds, s, a);
        else hipLaunchKernelGGL((k_c3r<1, 3, 7, 4, 8>), grid, dim3(512), p.lds, s, a);
        MNAS_CHECK_LAUNCH();
        return MNAS_OK;
    }
    if (p.kq == 8 && p.nt == 2 && p.ksw == 12) {
        if (c->mode == 0) hipLaunchKernelGGL((k_c3r<0, 2, 12, 4, 8>), grid, dim3(512), p.lds, s, a);
        else hipLaunchKernelGGL((k_c3r<1, 2, 12, 4, 8>), grid, dim3(512), p.lds, s, a);
        MNAS_CHECK_LAUNCH();
        return MNAS_OK;
    }
    return MNAS_EINVAL;
}
