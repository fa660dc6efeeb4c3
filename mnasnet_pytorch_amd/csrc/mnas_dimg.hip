// Dense 3x3 convolution (pad 1, stride 1 / 2) on the SMALL feature maps (output plane <= 256 pixels: the 14x14 and 7x7
// stages), forward and -- for stride 1 -- input gradient, with one whole image per workgroup.  Replaces, behind
// mnas_conv_gemm, k_igemm's im2col staging for ConvBlock(kernel_size=3) (mnasnet.py:48-62; the stage-transition convs
// 80->96, 96->192, 192->320 of mnasnet.py:157-161).
//
// Why: on these layers k_igemm is a chain of exposed latencies -- 64-pixel tiles (196 .. 980 workgroups), 14-27 K chunks
// per tile, each chunk a gather of 9 shifted copies of the same pixels from global memory plus a re-staged weight block
// (measured 32-95 us for 4-40 MB of traffic).  Here:
//   * the (padded, activated) input image is staged ONCE in LDS ([Hi+2][Wi+2][Ci+8] bf16, zero border), so the 9 taps are
//     9 offsets into the same tile: a B fragment (16 pixels x 32 k) is ONE 16-byte LDS read per lane (8 consecutive input
//     channels of one tap; k = tap*Ci + ci, the packed-weight order);
//   * the accumulators of the WHOLE output image live in registers (waves split the 16-pixel tiles, every wave covers the
//     workgroup's cout tiles), so the weights are streamed exactly once per image: K chunks of [couts][KC] through a
//     double-buffered LDS block, the next chunk's global loads in flight under the MFMAs, one barrier per chunk;
//   * wide outputs are cut into cout groups over grid.y (the image is re-staged per group: it is the small operand).
// MODE 0: forward (act-on-load input, bias, BatchNorm partial statistics).  MODE 1: input gradient of a stride-1 conv =
// the same correlation with mirrored tap offsets over the materialised dy and the [Ci][tap*Co+co] weight packing, plus the
// optional fused BatchNorm-backward reduce (as k_igemm MODE 1).  Roofline: HBM / L2 (weights are re-read per image from L2).
#include "mnas_common.h"

struct DimgArgs {
    int N, Hi, Wi, Ci, Ho, Wo, Co;   // Ci = reduction channels (dy channels in MODE 1), Co = result channels
    int stride;
    int Ktot, Kpad;                  // 9*Ci, rounded up to 32
    int KC, nkc;                     // K elements per weight chunk (multiple of 32), number of chunks
    int LW, Cp;                      // LDS image: (Hi+2) rows x LW = Wi+2 pixels x Cp = Ci+8 elements
    int cg;                          // cout tiles per workgroup (grid.y groups of cg tiles)
    int ni;                          // images per workgroup pass (7x7 planes: 2, so that a weight chunk is used twice)
    int co_pad16;
    MnasActIn act;                   // input (MODE 1: the materialised dy, no coefficients)
    const uint16_t* w;
    const float* bias;
    const void* resid;
    void* out;
    float* stats;                    // [2][Co][gridDim.x]
    const void* red_y;
    const float* red_bn;
};

// (PXW 4, CG 3: the 14x14 stride-1 layers with their couts cut into 48-channel groups -- 244-255 VGPRs, two workgroups per CU)
template <int MODE, int PXW, int CG>
__global__ __launch_bounds__(256, (PXW == 4 && CG == 3 ? 2 : 1)) void k_dimg(DimgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int img_elems = (a.Hi + 2) * a.LW * a.Cp;
    const int wpitch = a.KC + 8;                                   // weight chunk row pitch (elements)
    const int rows = a.cg * 16;                                    // weight rows of this workgroup
    uint16_t* img = (uint16_t*)smem;                               // [(Hi+2)*LW][Cp]
    const int img_pitch = (img_elems + 7) & ~7;                    // elements between the tiles of consecutive images
    uint16_t* wbuf = img + a.ni * img_pitch;                       // [2][rows][wpitch]
    float* lds_coef = (float*)(wbuf + 2 * rows * wpitch);          // [2][Ci] act-on-load scale / shift
    float* lds_red = lds_coef + 2 * a.Ci;                          // [4 waves][2][rows] (end of kernel)
    float* lds_rc = lds_red + 8 * rows;                            // MODE 1: [4][rows] reduce coefficients (s, t, invstd, -mean*invstd)
    const int ct0 = blockIdx.y * a.cg;                             // first cout tile of this group
    const int ctn = min(a.cg, (a.co_pad16 >> 4) - ct0);            // valid tiles in the group
    const bool has_coef = MODE == 0 && a.act.scale != nullptr;
    const bool do_red = MODE == 1 && a.red_y != nullptr;
    const int npix = a.Ho * a.Wo;

    for (int i = tid; i < 2 * a.Ci; i += 256)
        lds_coef[i] = has_coef ? (i < a.Ci ? a.act.scale[i] : a.act.shift[i - a.Ci]) : 0.f;
    for (int i = tid; i < ((a.ni * img_pitch) >> 3); i += 256) ((uint4*)img)[i] = make_uint4(0, 0, 0, 0);   // zero borders (and interiors)
    if (do_red)
        for (int i = tid; i < 4 * rows; i += 256) {
            const int r = i / rows, cc = ct0 * 16 + i % rows;
            float v = 0.f;
            if (cc < a.Co) {
                if (r == 0) v = a.red_bn[cc];
                else if (r == 1) v = a.red_bn[a.Co + cc];
                else if (r == 2) v = a.red_bn[6 * a.Co + cc];
                else v = -a.red_bn[5 * a.Co + cc] * a.red_bn[6 * a.Co + cc];
            }
            lds_rc[i] = v;
        }

    // this lane's output pixels (one per 16-pixel tile of the wave) and their base offset in the LDS image
    int pbase[PXW], pix[PXW];
#pragma unroll
    for (int i = 0; i < PXW; ++i) {
        const int p = (wave + 4 * i) * 16 + l15;                   // pixel of the ni consecutive images, taken as one plane
        const bool ok = p < a.ni * npix;
        pix[i] = ok ? p : -1;
        const int im = ok ? p / npix : 0, pp = ok ? p - im * npix : 0;
        const int oy = pp / a.Wo, ox = pp - oy * a.Wo;
        // MODE 0: padded row = oy*s + kh; MODE 1 (stride 1): padded row = oy + 2 - kh  (tap offsets applied below)
        pbase[i] = im * img_pitch + (oy * a.stride * a.LW + ox * a.stride) * a.Cp;
    }
    float bias_r[CG][4], s1[CG][4], s2[CG][4];
#pragma unroll
    for (int ct = 0; ct < CG; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = (ct0 + ct) * 16 + lg * 4 + r;
            bias_r[ct][r] = (a.bias && ct < ctn && co < a.Co) ? a.bias[co] : 0.f;
            s1[ct][r] = 0.f; s2[ct][r] = 0.f;
        }
    // weight chunk staging plan: 16-byte pieces (row, piece) of the [rows][KC] block
    constexpr int NWMAX = 6;
    const int ppr = a.KC >> 3;                                     // pieces per row
    const int npieces = rows * ppr;
    const int nw = (npieces + 255) >> 8;
    uint4 wregA[NWMAX], wregB[NWMAX];                              // chunks c and c+1 in flight (loads issued two chunks ahead)
    auto load_w = [&](uint4 (&wreg)[NWMAX], int c) {
        if (c >= a.nkc) return;
        const int kc0 = c * a.KC;
#pragma unroll
        for (int i = 0; i < NWMAX; ++i) {
            if (i >= nw) break;
            const int q = tid + 256 * i, row = q / ppr, pc = q - row * ppr;
            const int k = kc0 + pc * 8, gr = ct0 * 16 + row;
            wreg[i] = make_uint4(0, 0, 0, 0);
            if (q < npieces && gr < a.co_pad16 && k < a.Kpad) wreg[i] = *(const uint4*)(a.w + (size_t)gr * a.Kpad + k);
        }
    };
    auto store_w = [&](const uint4 (&wreg)[NWMAX], int buf) {
        uint16_t* dst = wbuf + buf * rows * wpitch;
#pragma unroll
        for (int i = 0; i < NWMAX; ++i) {
            if (i >= nw) break;
            const int q = tid + 256 * i, row = q / ppr, pc = q - row * ppr;
            if (q < npieces) *(uint4*)(dst + row * wpitch + pc * 8) = wreg[i];
        }
    };

    const int ci8 = a.Ci >> 3;
    const int in_slots = a.Hi * a.Wi * ci8;
    const int ngroups = (a.N + a.ni - 1) / a.ni;
    for (int ng = blockIdx.x; ng < ngroups; ng += gridDim.x) {
        const int n = ng * a.ni, nimg = min(a.ni, a.N - n);          // images n .. n+nimg-1
        __syncthreads();                                           // previous image consumed (first pass: zero fill / coefficients visible)
        // ---- stage the image: 16-byte chunks, act-on-load, into the padded tile
        const uint16_t* src = (const uint16_t*)a.act.data + (size_t)n * a.Hi * a.Wi * a.Ci;
        const int tot_slots = nimg * in_slots;
        for (int q0 = 0; q0 < tot_slots; q0 += 256 * 8) {          // 8 loads in flight per thread, then transform + store
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = q0 + tid + 256 * j;
                v[j] = make_uint4(0, 0, 0, 0);
                if (q < tot_slots) v[j] = *(const uint4*)(src + (size_t)q * 8);         // (image, pixel, chunk) slots are contiguous in NHWC
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = q0 + tid + 256 * j;
                if (q >= tot_slots) continue;
                const int imq = q / in_slots, qq = q - imq * in_slots;
                const int pixq = qq / ci8, c8 = qq - pixq * ci8;
                const int iy = pixq / a.Wi, ix = pixq - iy * a.Wi;
                uint4 u = v[j];
                if (has_coef) {
                    float s[8], t[8];
                    *(float4*)&s[0] = *(const float4*)(lds_coef + c8 * 8); *(float4*)&s[4] = *(const float4*)(lds_coef + c8 * 8 + 4);
                    *(float4*)&t[0] = *(const float4*)(lds_coef + a.Ci + c8 * 8); *(float4*)&t[4] = *(const float4*)(lds_coef + a.Ci + c8 * 8 + 4);
                    u = act8(u, s, t);
                }
                *(uint4*)(img + imq * img_pitch + ((iy + 1) * a.LW + ix + 1) * a.Cp + c8 * 8) = u;
            }
        }
        load_w(wregA, 0);
        load_w(wregB, 1);
        f32x4_t acc[PXW][CG];
#pragma unroll
        for (int i = 0; i < PXW; ++i)
#pragma unroll
            for (int ct = 0; ct < CG; ++ct) acc[i][ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // this lane's k position: 8 consecutive k = 8 consecutive channels ci0.. of tap `tap`
        int tap = (lg * 8) / a.Ci, ci0 = lg * 8 - tap * a.Ci;
        auto chunk = [&](uint4 (&wreg)[NWMAX], int c) {
            if (c >= a.nkc) return;                                // uniform
            store_w(wreg, c & 1);
            __syncthreads();                                       // chunk c (and, c == 0, the image) published; buffer (c+1)&1 is free
            load_w(wreg, c + 2);
            const uint16_t* wb = wbuf + (c & 1) * rows * wpitch;
            const int ksteps = min(a.KC, a.Kpad - c * a.KC) >> 5;
            for (int ks = 0; ks < ksteps; ++ks) {
                int toff = 0;
                if (tap < 9) {
                    const int th = tap / 3, tw = tap - th * 3;
                    const int dh = MODE == 1 ? 2 - th : th, dw = MODE == 1 ? 2 - tw : tw;
                    toff = (dh * a.LW + dw) * a.Cp + ci0;
                }
                bf16x8_t bfrag[PXW];
#pragma unroll
                for (int i = 0; i < PXW; ++i) bfrag[i] = *(const bf16x8_t*)(img + pbase[i] + toff);
#pragma unroll
                for (int ct = 0; ct < CG; ++ct) {
                    if (ct < ctn) {                                // uniform
                        const bf16x8_t afrag = *(const bf16x8_t*)(wb + (ct * 16 + l15) * wpitch + ks * 32 + lg * 8);
#pragma unroll
                        for (int i = 0; i < PXW; ++i)
                            acc[i][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[i], acc[i][ct], 0, 0, 0);
                    }
                }
                ci0 += 32;                                         // next K step of this lane
                while (ci0 >= a.Ci) { ci0 -= a.Ci; ++tap; }
            }
        };
        for (int c = 0; c < a.nkc; c += 2) {
            chunk(wregA, c);
            chunk(wregB, c + 1);
        }
        // ---- epilogue: lane holds couts (ct0+ct)*16 + lg*4 + {0..3} of pixel pix[i]
#pragma unroll
        for (int i = 0; i < PXW; ++i) {
            if (pix[i] < 0 || pix[i] >= nimg * npix) continue;
            const size_t obase = ((size_t)n * npix + pix[i]) * a.Co;
            uint2 yreg[CG];
            if (do_red) {
#pragma unroll
                for (int ct = 0; ct < CG; ++ct) {                  // all reduce operands of this pixel in flight together
                    const int co = (ct0 + ct) * 16 + lg * 4;
                    yreg[ct] = make_uint2(0, 0);
                    if (ct < ctn && co < a.Co) yreg[ct] = *(const uint2*)((const uint16_t*)a.red_y + obase + co);
                }
            }
#pragma unroll
            for (int ct = 0; ct < CG; ++ct) {
                const int co = (ct0 + ct) * 16 + lg * 4;
                if (ct >= ctn || co >= a.Co) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][ct][r] + bias_r[ct][r];
                if (a.resid) {
                    const uint2 rv = *(const uint2*)((const uint16_t*)a.resid + obase + co);
                    v[0] += bf_lo(rv.x); v[1] += bf_hi(rv.x); v[2] += bf_lo(rv.y); v[3] += bf_hi(rv.y);
                }
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s1[ct][r] += v[r]; s2[ct][r] = fmaf(v[r], v[r], s2[ct][r]); }
                }
                uint2 pk;
                pk.x = pack_bf16(v[0], v[1]);
                pk.y = pack_bf16(v[2], v[3]);
                *(uint2*)((uint16_t*)a.out + obase + co) = pk;
                if (do_red) {
                    // fused BN-backward reduce (as k_igemm MODE 1): dz = g*[s*y+t>0] with g as stored, xhat = (y-mean)*invstd
                    const uint2 yv = yreg[ct];
                    const float gq[4] = {bf_lo(pk.x), bf_hi(pk.x), bf_lo(pk.y), bf_hi(pk.y)};
                    const float yq[4] = {bf_lo(yv.x), bf_hi(yv.x), bf_lo(yv.y), bf_hi(yv.y)};
                    const int cl = ct * 16 + lg * 4;
                    const float4 cs = *(const float4*)(lds_rc + cl), ctt = *(const float4*)(lds_rc + rows + cl);
                    const float4 ci_ = *(const float4*)(lds_rc + 2 * rows + cl), cm = *(const float4*)(lds_rc + 3 * rows + cl);
                    const float rs_[4] = {cs.x, cs.y, cs.z, cs.w}, rt_[4] = {ctt.x, ctt.y, ctt.z, ctt.w};
                    const float ri_[4] = {ci_.x, ci_.y, ci_.z, ci_.w}, rm_[4] = {cm.x, cm.y, cm.z, cm.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dz = (fmaf(yq[r], rs_[r], rt_[r]) > 0.f) ? gq[r] : 0.f;
                        s1[ct][r] += dz;
                        s2[ct][r] = fmaf(dz, fmaf(yq[r], ri_[r], rm_[r]), s2[ct][r]);
                    }
                }
            }
        }
    }
    if ((MODE == 0 || do_red) && a.stats) {
        // deterministic workgroup reduction (as k_igemm): 16-lane shuffle tree, one LDS slot per (wave, channel), waves in order
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < CG; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x1 = s1[ct][r], x2 = s2[ct][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
                if (l15 == 0 && ct < ctn) {
                    lds_red[(wave * 2 + 0) * rows + ct * 16 + lg * 4 + r] = x1;
                    lds_red[(wave * 2 + 1) * rows + ct * 16 + lg * 4 + r] = x2;
                }
            }
        __syncthreads();
        for (int i = tid; i < 2 * ctn * 16; i += 256) {
            const int r = i / (ctn * 16), cl = i - r * ctn * 16, cc = ct0 * 16 + cl;
            const float v = ((lds_red[(0 * 2 + r) * rows + cl] + lds_red[(1 * 2 + r) * rows + cl]) + lds_red[(2 * 2 + r) * rows + cl]) +
                            lds_red[(3 * 2 + r) * rows + cl];
            if (cc < a.Co) a.stats[((size_t)r * a.Co + cc) * gridDim.x + blockIdx.x] = v;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
struct DimgPlan { int pxw, cg, groups, kc, nkc, ni; size_t lds; };
int mnas_dimg_parts(int mode, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad);

int mnas_dimg_enabled() {
    static int on = -1;
    if (on < 0) on = mnas_diag_env("MNAS_DIMG", 1);
    return on;
}
static bool dimg_plan(int mode, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad,
                      DimgPlan* p) {
    if (!mnas_dimg_enabled() || kh != 3 || kw != 3 || pad != 1 || (stride != 1 && stride != 2)) return false;
    if (mode == 1 && stride != 1) return false;
    if ((Ci & 7) || (Co & 7) || Ci < 16 || N < 32) return false;
    if (Ho != (Hi - 1) / stride + 1 || Wo != (Wi - 1) / stride + 1) return false;
    const int npix = Ho * Wo, tiles = (Co + 15) / 16;
    if (npix > 256) return false;
    // stride 2 onto a 7x7 plane (96 -> 192): two 14x14 input tiles leave room for 64-wide weight chunks only, 46 vs k_igemm's 32 us
    if (stride == 2 && npix <= 64 && mnas_dimg_enabled() < 2) return false;
    // 7x7 planes: two images per pass (98 pixels, 2 tiles per wave): every weight chunk is used twice -- the weights are
    // re-read per pass from L2, 283 MB per launch for 192->320 with one image, which is what bounded it
    static int wgs = -1, nimax = -1;
    if (wgs < 0) { wgs = mnas_diag_env("MNAS_DIMG_WGS", 256); nimax = mnas_diag_env("MNAS_DIMG_NI", 2); }
    const int Kpad = (9 * Ci + 31) / 32 * 32;
    for (p->ni = (npix <= 64 ? nimax : 1); p->ni >= 1; --p->ni) {
        p->pxw = p->ni * npix > 128 ? 4 : (p->ni * npix > 64 ? 2 : 1);
        const int cgmax = p->pxw == 4 ? 6 : 12;
        p->groups = (tiles + cgmax - 1) / cgmax;
        // enough workgroups to cover the CUs: cout groups over grid.y (each re-stages the image).  14x14 planes, stride 1: twice
        // as many, smaller groups (<= 3 cout tiles: the 2-workgroups-per-CU instantiation): 80 -> 96 forward 33 -> 26 us, its input
        // gradient 45 -> 33 us (round 4); the same split loses on the 7x7 planes (72 -> 82 us) and on the stride-2 layers
        const int want = (npix > 64 && stride == 1) ? 2 * wgs : wgs;
        while ((long long)(N / p->ni) * p->groups < want && p->groups < tiles) ++p->groups;
        p->cg = (tiles + p->groups - 1) / p->groups;
        p->groups = (tiles + p->cg - 1) / p->cg;
        const size_t img = (((size_t)(Hi + 2) * (Wi + 2) * (Ci + 8) * 2 + 15) & ~(size_t)15) * p->ni;
        // largest K chunk (multiple of 32, <= 256) whose double buffer fits next to the image tiles -- two workgroups per CU
        // (78 KB) if possible --, 6 staging registers per thread and chunk at most
        for (int budget = 78; budget <= 150; budget += 72)
            for (int kc = 256; kc >= 32; kc -= 32) {
                const size_t wb = (size_t)2 * p->cg * 16 * (kc + 8) * 2;
                const size_t lds = img + wb + (size_t)2 * Ci * 4 + (size_t)12 * p->cg * 16 * 4;
                const int pieces = p->cg * 16 * (kc / 8);
                if (lds <= (size_t)budget * 1024 && pieces <= 6 * 256) {
                    p->kc = kc < Kpad ? kc : Kpad; p->nkc = (Kpad + p->kc - 1) / p->kc; p->lds = lds;
                    return true;
                }
            }
    }
    return false;
}
// > 0: this problem runs on k_dimg with that many persistent workgroups along grid.x (the caller passes it as nparts)
extern "C" int mnas_conv_img_parts(int mode, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int k, int stride, int pad) {
    const int r = mnas_c3r_parts(mode, N, Hi, Wi, Ci, Ho, Wo, Co, k, k, stride, pad);      // (mnas_conv_gemm tries that kernel first)
    if (r > 0) return r;
    return mnas_dimg_parts(mode, N, Hi, Wi, Ci, Ho, Wo, Co, k, k, stride, pad);
}
int mnas_dimg_parts(int mode, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad) {
    DimgPlan p;
    if (!dimg_plan(mode, N, Hi, Wi, Ci, Ho, Wo, Co, kh, kw, stride, pad, &p)) return -1;
    return (N + p.ni - 1) / p.ni;
}

int mnas_dimg_run(const MnasConvGemm* c, void* stream) {
    DimgPlan p;
    if (!dimg_plan(c->mode, c->N, c->Hi, c->Wi, c->Ci, c->Ho, c->Wo, c->Co, c->kh, c->kw, c->stride, c->pad, &p)) return MNAS_EINVAL;
    if (c->mode == 1 && (c->grad.coef || c->grad.y)) return MNAS_EINVAL;          // materialised dy only
    if (c->mode == 0 && c->resid) return MNAS_EINVAL;
    DimgArgs a = {};
    a.N = c->N; a.Hi = c->Hi; a.Wi = c->Wi; a.Ci = c->Ci; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co;
    a.stride = c->stride;
    a.Ktot = 9 * c->Ci; a.Kpad = (a.Ktot + 31) / 32 * 32;
    a.KC = p.kc; a.nkc = p.nkc;
    a.LW = c->Wi + 2; a.Cp = c->Ci + 8;
    a.cg = p.cg; a.ni = p.ni; a.co_pad16 = (c->Co + 15) / 16 * 16;
    if (c->mode == 0) a.act = c->act;
    else { a.act.data = c->grad.g; a.act.scale = nullptr; a.act.shift = nullptr; }
    a.w = (const uint16_t*)c->w; a.bias = c->bias; a.resid = c->resid; a.out = c->out; a.stats = c->stats;
    a.red_y = c->mode == 1 ? c->red_y : nullptr; a.red_bn = c->red_bn;
    if (a.red_y && (!a.red_bn || !a.stats)) return MNAS_EINVAL;
    const dim3 grid(c->nparts, p.groups);            // workgroups beyond N only write their (zero) statistics column
    hipStream_t s = (hipStream_t)stream;
#define MNAS_DIMG(M_, P_, C_) \
    if (c->mode == M_ && p.pxw == P_ && p.cg <= C_) { \
        hipLaunchKernelGGL((k_dimg<M_, P_, C_>), grid, dim3(256), p.lds, s, a); \
        MNAS_CHECK_LAUNCH(); \
        return MNAS_OK; \
    }
    MNAS_DIMG(0, 4, 3) MNAS_DIMG(1, 4, 3)            // (smallest fitting cout-tile bound first: the register arrays are sized by it)
    MNAS_DIMG(0, 4, 6) MNAS_DIMG(0, 2, 12) MNAS_DIMG(0, 1, 12) MNAS_DIMG(1, 4, 6) MNAS_DIMG(1, 2, 12) MNAS_DIMG(1, 1, 12)
#undef MNAS_DIMG
    return MNAS_EINVAL;
}
