// 1x1 (pointwise) convolution forward as a DMA-pipelined MFMA GEMM: out[M][Co] = act(in)[M][Ci] * W[Co][Ci]^T + bias.
// Replaces ATen conv2d forward for ConvBlock(kernel_size=1) (mnasnet.py:48-62, the expand / project convs of
// MBConv_block :116-129 and SepConv's pointwise conv :82-95).  Same contract as mnas_conv_gemm mode 0 (raw bf16 output,
// per-workgroup BatchNorm partial statistics, producer's BatchNorm+ReLU applied to the input on the way in).
//
// Why a second GEMM kernel: k_igemm stages its tiles global -> VGPR -> LDS, one (tile, K-chunk) ahead at most, and a
// workgroup lives for one or two tiles.  On the 14x14 / 7x7 / 28x28 stages (50-200 k pixels) every launch is then a
// chain of exposed memory latencies (measured 0.5-1.4 TB/s of 4.7 achievable); on the 112x112 / 56x56 expand convs the
// 8-byte-per-lane epilogue stores cap the write rate.  Here:
//   * persistent workgroups walk (pixel tile, K chunk) PHASES; the raw activation chunk of phase n+1 (and the weight
//     chunk, unless the whole weight block is LDS-resident) is copied HBM/L2 -> LDS by `global_load_lds` while phase n
//     computes: no staging registers, one barrier per phase;
//   * the LDS image is the padded row layout the fragments want ([row][nch+1] 16-byte slots, the pad slot is a lane
//     the DMA masks off); the thread that issued a slot's DMA also applies relu(scale*x+shift) to it IN PLACE
//     (its own vmcnt wait covers its own slots), so the transform runs once per element and needs no extra barrier;
//   * MFMA orientation as k_igemm: A = weights [16 cout][32 k], B = activations [16 pixels][32 k], D[cout][pixel];
//   * epilogue: bias, statistics in registers across all tiles of the workgroup, bf16 pack, then the tile goes through an
//     LDS out-stage and leaves as 16-byte-per-lane stores of whole row segments (NB*2 contiguous bytes per pixel),
//     issued one phase later under the next tile's MFMAs (the out-stage is double-buffered, so it rides on the phase
//     barrier).
// Work split: grid.x persistent workgroups over 64*PT-pixel tiles, grid.y over blocks of NB = 16*NT output channels
// (the activation tile of an expand conv is re-read per block -- from L2 / the Infinity Cache: it is the small tensor).
// Roofline: HBM (AI 11-165 flop/B, left of the 312 flop/B bf16 ridge).
#include "mnas_common.h"
#ifdef MNAS_DIAG
#define PWF_ABL(a_, bit_) ((a_).abl & (bit_))
#else
#define PWF_ABL(a_, bit_) false
#endif
#include <cstdlib>

typedef __attribute__((address_space(3))) void* pwf_lds_ptr;
typedef const __attribute__((address_space(1))) void* pwf_gbl_ptr;

struct PwfArgs {
    int M, Ci, Co;
    int Kpad;                // Ci rounded up to 32 (row length of the packed weights)
    int kc, nkc;             // K elements per chunk (multiple of 32, <= 128), number of chunks
    int nch;                 // kc / 8: data slots (16 B) per LDS row; row pitch = nch + 1 slots
    int co_pad16;
    int nt_store;
    int abl;                 // ablation bits (MNAS_PWF_ABL, diagnosis): 1 no stores, 2 no transform, 4 no MFMA, 8 no activation DMA
    MnasActIn act;
    const uint16_t* w;       // MNAS_PACK_FWD: [co_pad16][Kpad]
    const float* bias;
    void* out;
    float* stats;            // [2][Co][gridDim.x] or NULL
    // MODE 1 (input gradient of a WIDENING-in-backward 1x1 conv, i.e. the project conv: few dy channels, many outputs)
    MnasGradIn grad;         // dy-on-load operand (M, Ci); coef == NULL: g is dy
    const void* red_y;       // fused BatchNorm-backward reduce target (M, Co) or NULL
    const float* red_bn;
};

// MODE 0: forward (see above).  MODE 1: out[M][Co] = dy[M][Ci] * Wd[Co][Ci]^T with dy = c1*(g*[s*y+t>0]) + c2*y + c3 formed
// in place in LDS from the DMA'd raw (g, y) tiles, no bias / statistics of the output; instead the optional fused
// BatchNorm-backward reduce of the layer the result is the gradient of (as k_igemm MODE 1), done on the copy-out path where a
// thread owns one 16-byte channel column: the reduce operand red_y is loaded one phase ahead.
template <int MODE, int NT, int PT, bool WRES>
__global__ __launch_bounds__(256) void k_pwf(PwfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BP = 64 * PT, NB = NT * 16;
    constexpr int OPITCH = NB / 8 + 1;                       // out-stage row pitch (16-byte slots)
    constexpr int MAXA = (BP * 17 + 255) / 256;              // DMA slots per thread and chunk, activation tile (pitch <= 17)
    constexpr int MAXW = (NB * 17 + 255) / 256;              //                                  weight block
    constexpr int NO = (BP * (NB / 8) + 255) / 256;          // out-stage copy slots per thread
    const int pitch = a.nch + 1;
    constexpr int CROWS = MODE == 1 ? 5 : 2;
    constexpr int ASLOTS = MODE == 1 ? 3 : 2;                // MODE 1: slot 2 is the (single-buffered) raw-y tile
    float* lds_coef = (float*)smem;                                      // [CROWS][Kpad]
    float* lds_redc = lds_coef + CROWS * a.Kpad;                         // MODE 1: [4][NB] reduce coefficients
    uint4* lds_a = (uint4*)(lds_redc + (MODE == 1 ? 4 * NB : 0));        // [ASLOTS][BP * pitch]
    uint4* lds_w = lds_a + ASLOTS * BP * pitch;                          // [WRES ? nkc : 2][NB * pitch]
    uint4* lds_o = lds_w + (WRES ? a.nkc : 2) * NB * pitch;              // [2][BP * OPITCH]
    float* lds_red = (float*)lds_o;                                      // reused at the very end: [4][2][NB] / MODE 1 [256][16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.y * NB;
    const bool has_coef = MODE == 1 ? a.grad.coef != nullptr : a.act.scale != nullptr;
    const bool do_red = MODE == 1 && a.red_y != nullptr;
    const int na = (BP * pitch + 255) >> 8, nw = (NB * pitch + 255) >> 8;   // DMA rounds (uniform)

    // ---- slot plans (tile-invariant): slot q = 256*i + tid = row * pitch + j
    int pa[MAXA], ja[MAXA], pw_[MAXW], jw[MAXW];
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
        const int q = 256 * i + tid;
        pa[i] = q / pitch; ja[i] = q - pa[i] * pitch;
        if (pa[i] >= BP || ja[i] >= a.nch) ja[i] = -1;      // beyond the tile, or the pad slot: never written
    }
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
        const int q = 256 * i + tid;
        pw_[i] = q / pitch; jw[i] = q - pw_[i] * pitch;
        if (pw_[i] >= NB || jw[i] >= a.nch || n0 + pw_[i] >= a.co_pad16) jw[i] = -1;
    }
    // ---- one-time setup: zero the tiles (K padding and never-written slots must read as 0), coefficient table
    {
        const int nz = (ASLOTS * BP + (WRES ? a.nkc : 2) * NB) * pitch;
        for (int i = tid; i < nz; i += 256) lds_a[i] = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < CROWS * a.Kpad; i += 256) {
            const int r = i / a.Kpad, c = i - r * a.Kpad;
            float v = 0.f;
            if (has_coef && c < a.Ci) {
                if (MODE == 1) v = a.grad.coef[r * a.Ci + c];
                else v = r == 0 ? a.act.scale[c] : a.act.shift[c];
            }
            lds_coef[i] = v;
        }
        if (do_red)
            for (int i = tid; i < 4 * NB; i += 256) {      // (s, t, invstd, -mean*invstd) of the reduce target, as k_igemm
                const int r = i / NB, co = n0 + i % NB;
                float v = 0.f;
                if (co < a.Co) {
                    if (r == 0) v = a.red_bn[0 * a.Co + co];
                    else if (r == 1) v = a.red_bn[1 * a.Co + co];
                    else if (r == 2) v = a.red_bn[6 * a.Co + co];
                    else v = -a.red_bn[5 * a.Co + co] * a.red_bn[6 * a.Co + co];
                }
                lds_redc[i] = v;
            }
    }
    __syncthreads();

    auto dma_a = [&](int slot, int tile0, int k0) {
        uint4* dst = lds_a + slot * BP * pitch;
        uint4* dsty = lds_a + 2 * BP * pitch;
        const uint16_t* src = (const uint16_t*)(MODE == 1 ? a.grad.g : a.act.data);
        const uint16_t* srcy = (const uint16_t*)a.grad.y;
        if (PWF_ABL(a, 8)) return;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            if (i >= na) break;
            const int k = k0 + ja[i] * 8;
            if (ja[i] >= 0 && k < a.Ci && tile0 + pa[i] < a.M) {
                const size_t off = (size_t)(tile0 + pa[i]) * a.Ci + k;
                __builtin_amdgcn_global_load_lds((pwf_gbl_ptr)(src + off), (pwf_lds_ptr)(dst + 256 * i + wave * 64), 16, 0, 0);
                if (MODE == 1 && has_coef)
                    __builtin_amdgcn_global_load_lds((pwf_gbl_ptr)(srcy + off), (pwf_lds_ptr)(dsty + 256 * i + wave * 64), 16, 0, 0);
            }
        }
    };
    auto dma_w = [&](int slot, int k0) {
        uint4* dst = lds_w + slot * NB * pitch;
#pragma unroll
        for (int i = 0; i < MAXW; ++i) {
            if (i >= nw) break;
            const int k = k0 + jw[i] * 8;
            if (jw[i] >= 0 && k < a.Kpad)
                __builtin_amdgcn_global_load_lds((pwf_gbl_ptr)(a.w + (size_t)(n0 + pw_[i]) * a.Kpad + k),
                                                 (pwf_lds_ptr)(dst + 256 * i + wave * 64), 16, 0, 0);
        }
    };
    // Single-k-step layers (Ci <= 32: 16 -> 48 at 112x112, 24 -> 72 at 56x56 -- bandwidth-bound, 5.4 TB/s on a materialised input):
    // the activation is applied to the MFMA B fragment in REGISTERS after the LDS read (a fragment is read by exactly one wave, its
    // 8 coefficient pairs are per-lane constants for the whole kernel) instead of in place in LDS, where the read-modify-write sat
    // between the DMA wait and the barrier of every phase: 97 -> 77 us for 16 -> 48 on a virtual input (round 5, tools/kbench.py).
    // Same act8 on the same raw chunk -> bit-identical.
    const bool frag_xf = MODE == 0 && has_coef && a.nkc == 1 && a.kc == 32 && !PWF_ABL(a, 2);
    float fx_s[8], fx_t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { fx_s[j] = frag_xf ? lds_coef[lg * 8 + j] : 0.f; fx_t[j] = frag_xf ? lds_coef[a.Kpad + lg * 8 + j] : 0.f; }
    // relu(scale*x+shift) in place on the slots THIS thread's DMA wrote (after its own vmcnt wait)
    auto transform_a = [&](int slot, int tile0, int k0) {
        if (!has_coef || frag_xf || PWF_ABL(a, 2)) return;
        uint4* t = lds_a + slot * BP * pitch;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            if (i >= na) break;
            const int k = k0 + ja[i] * 8;
            if (ja[i] >= 0 && k < a.Ci && tile0 + pa[i] < a.M) {
                float s[8], sh[8];
                *(float4*)&s[0] = *(const float4*)(lds_coef + k);
                *(float4*)&s[4] = *(const float4*)(lds_coef + k + 4);
                *(float4*)&sh[0] = *(const float4*)(lds_coef + a.Kpad + k);
                *(float4*)&sh[4] = *(const float4*)(lds_coef + a.Kpad + k + 4);
                if constexpr (MODE == 1) {
                    float c1[8], c2[8], c3[8], d[8];
                    *(float4*)&c1[0] = *(const float4*)(lds_coef + 2 * a.Kpad + k);
                    *(float4*)&c1[4] = *(const float4*)(lds_coef + 2 * a.Kpad + k + 4);
                    *(float4*)&c2[0] = *(const float4*)(lds_coef + 3 * a.Kpad + k);
                    *(float4*)&c2[4] = *(const float4*)(lds_coef + 3 * a.Kpad + k + 4);
                    *(float4*)&c3[0] = *(const float4*)(lds_coef + 4 * a.Kpad + k);
                    *(float4*)&c3[4] = *(const float4*)(lds_coef + 4 * a.Kpad + k + 4);
                    dy8(t[256 * i + tid], lds_a[2 * BP * pitch + 256 * i + tid], s, sh, c1, c2, c3, d);
                    t[256 * i + tid] = pack8(d);
                } else {
                    t[256 * i + tid] = act8(t[256 * i + tid], s, sh);
                }
            }
        }
    };
    auto copy_out = [&](int obuf, int tile0) {
        const uint4* o = lds_o + obuf * BP * OPITCH;
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int q = 256 * i + tid;
            const int p = q / (NB / 8), c8 = q - p * (NB / 8);
            const int m = tile0 + p, co = n0 + c8 * 8;
            if (p < BP && m < a.M && co < a.Co && !PWF_ABL(a, 1))
                st_u4((uint16_t*)a.out + (size_t)m * a.Co + co, o[p * OPITCH + c8], a.nt_store);
        }
    };

    // MODE 1 copy-out role: thread -> fixed 16-byte channel column of the out-stage rows orow0, orow0 + orows, ...
    constexpr int NCH8 = NB / 8;
    constexpr int TCOLS = (256 / NCH8) * NCH8, OROWS = TCOLS / NCH8;
    constexpr int MAXR = MODE == 1 ? (BP + OROWS - 1) / OROWS : 1;
    const int oc8 = tid % NCH8, orow0 = tid / NCH8;
    float r1[8], r2[8];
    int ooff[MAXR];
    uint4 yreg[MAXR];
#pragma unroll
    for (int j = 0; j < 8; ++j) { r1[j] = 0.f; r2[j] = 0.f; }
    auto plan_out = [&](int tile0) {          // runs in tile t's own phase; copy_out1 consumes it one phase later
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            const int p = orow0 + k * OROWS, m = tile0 + p, co = n0 + oc8 * 8;
            ooff[k] = -1;
            yreg[k] = make_uint4(0, 0, 0, 0);
            if (tid < TCOLS && p < BP && m < a.M && co < a.Co) {
                ooff[k] = m * a.Co + co;
                if (do_red) yreg[k] = *(const uint4*)((const uint16_t*)a.red_y + ooff[k]);
            }
        }
    };
    auto copy_out1 = [&](int obuf) {
        const uint4* o = lds_o + obuf * BP * OPITCH;
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            if (ooff[k] < 0) continue;
            const int p = orow0 + k * OROWS;
            const uint4 pk = o[p * OPITCH + oc8];
            if (!PWF_ABL(a, 1)) st_u4((uint16_t*)a.out + ooff[k], pk, a.nt_store);
            if (do_red) {
                const uint32_t gu[4] = {pk.x, pk.y, pk.z, pk.w}, yu[4] = {yreg[k].x, yreg[k].y, yreg[k].z, yreg[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {                // channel pairs in float2 (v_pk_fma_f32)
                    const int c = oc8 * 8 + 2 * j;
                    mnas_red2(gu[j], yu[j], mnas_ld2(lds_redc + c), mnas_ld2(lds_redc + NB + c), mnas_ld2(lds_redc + 2 * NB + c),
                              mnas_ld2(lds_redc + 3 * NB + c), r1 + 2 * j, r2 + 2 * j);
                }
            }
        }
    };

    float bias_r[NT][4], s1[NT][4], s2[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = n0 + nt * 16 + lg * 4 + r;
            bias_r[nt][r] = (a.bias && co < a.Co) ? a.bias[co] : 0.f;
            s1[nt][r] = 0.f; s2[nt][r] = 0.f;
        }

    const int ntiles = (a.M + BP - 1) / BP;
    if (WRES) {
        for (int kc = 0; kc < a.nkc; ++kc) dma_w(kc, kc * a.kc);
    }
    // phase n = (tile, K chunk); its operands live in slot n & 1
    int ph = 0;
    if ((int)blockIdx.x < ntiles) {
        dma_a(0, blockIdx.x * BP, 0);
        if (!WRES) dma_w(0, 0);
    }
    int prev_tile0 = -1;          // tile whose out-stage buffer is waiting to be copied out
    int obuf = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tile0 = t * BP;
        f32x4_t acc[PT][NT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[pt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < a.nkc; ++kc, ++ph) {
            const int slot = ph & 1, k0 = kc * a.kc;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this thread's DMA of phase ph (and older stores) landed
            transform_a(slot, tile0, k0);
            __syncthreads();                                       // phase ph operands + previous tile's out-stage published;
                                                                   // every wave is done with phase ph-1 (slot ^ 1 is free)
            {   // operands of the next phase of this workgroup
                int nk = kc + 1, ntile = t;
                if (nk == a.nkc) { nk = 0; ntile = t + gridDim.x; }
                if (ntile < ntiles) {
                    dma_a(slot ^ 1, ntile * BP, nk * a.kc);
                    if (!WRES) dma_w(slot ^ 1, nk * a.kc);
                }
            }
            if (prev_tile0 >= 0) {
                if constexpr (MODE == 1) copy_out1(obuf ^ 1); else copy_out(obuf ^ 1, prev_tile0);
                prev_tile0 = -1;
            }
            if (MODE == 1 && kc == 0) plan_out(tile0);
            const uint4* ta = lds_a + slot * BP * pitch;
            const uint4* tw = lds_w + (WRES ? kc : slot) * NB * pitch;
            const int ksteps = PWF_ABL(a, 4) ? 0 : (min(a.kc, a.Kpad - k0) >> 5);
            for (int ks = 0; ks < ksteps; ++ks) {
                bf16x8_t bfrag[PT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    bfrag[pt] = *(const bf16x8_t*)(ta + ((wave * PT + pt) * 16 + l15) * pitch + ks * 4 + lg);
                    if (frag_xf) {                           // (ks == 0: the layer has one k-step)
                        const uint4 v = act8(*(const uint4*)&bfrag[pt], fx_s, fx_t);
                        bfrag[pt] = *(const bf16x8_t*)&v;
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8_t afrag = *(const bf16x8_t*)(tw + (nt * 16 + l15) * pitch + ks * 4 + lg);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        acc[pt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[pt], acc[pt][nt], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: lane holds couts n0 + nt*16 + lg*4 + {0..3} of pixel tile0 + (wave*PT+pt)*16 + l15
        uint4* o = lds_o + obuf * BP * OPITCH;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int p = (wave * PT + pt) * 16 + l15;
            const bool mok = tile0 + p < a.M;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = MODE == 1 ? acc[pt][nt][r] : acc[pt][nt][r] + bias_r[nt][r];
                if (MODE == 0 && mok) {                      // channel pairs in float2 (v_pk_fma_f32)
                    mnas_stat2((mnas_f2){v[0], v[1]}, &s1[nt][0], &s2[nt][0]);
                    mnas_stat2((mnas_f2){v[2], v[3]}, &s1[nt][2], &s2[nt][2]);
                }
                uint2 pk;
                pk.x = pack_bf16(v[0], v[1]);
                pk.y = pack_bf16(v[2], v[3]);
                *(uint2*)((unsigned char*)(o + p * OPITCH) + nt * 32 + lg * 8) = pk;
            }
        }
        prev_tile0 = tile0;
        obuf ^= 1;
    }
    __syncthreads();
    if (prev_tile0 >= 0) {
        if constexpr (MODE == 1) copy_out1(obuf ^ 1); else copy_out(obuf ^ 1, prev_tile0);
    }

    if constexpr (MODE == 1) {
        if (do_red && a.stats) {
            // per-thread sums -> per-channel: the threads of one channel column added in thread order (deterministic)
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) { lds_red[tid * 16 + j] = r1[j]; lds_red[tid * 16 + 8 + j] = r2[j]; }
            __syncthreads();
            for (int i = tid; i < 2 * NB; i += 256) {
                const int r = i / NB, cl = i % NB, c = n0 + cl;
                float v = 0.f;
                for (int th = cl >> 3; th < TCOLS; th += NCH8) v += lds_red[th * 16 + r * 8 + (cl & 7)];
                if (c < a.Co) a.stats[((size_t)r * a.Co + c) * gridDim.x + blockIdx.x] = v;
            }
        }
        return;
    }
    if (a.stats) {
        // deterministic workgroup reduction (as k_igemm): 16-lane shuffle tree, one LDS slot per (wave, channel), waves in order
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x1 = s1[nt][r], x2 = s2[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x1 += __shfl_xor(x1, o, 64); x2 += __shfl_xor(x2, o, 64); }
                if (l15 == 0) {
                    lds_red[(wave * 2 + 0) * NB + nt * 16 + lg * 4 + r] = x1;
                    lds_red[(wave * 2 + 1) * NB + nt * 16 + lg * 4 + r] = x2;
                }
            }
        __syncthreads();
        for (int i = tid; i < 2 * NB; i += 256) {
            const int r = i / NB, cl = i % NB, c = n0 + cl;
            const float v = ((lds_red[(0 * 2 + r) * NB + cl] + lds_red[(1 * 2 + r) * NB + cl]) + lds_red[(2 * 2 + r) * NB + cl]) +
                            lds_red[(3 * 2 + r) * NB + cl];
            if (c < a.Co) a.stats[((size_t)r * a.Co + c) * gridDim.x + blockIdx.x] = v;      // [2][Co][P]
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct PwfPlan { int nt, nblocks, pt, kc, nkc, wres; size_t lds; };

static size_t pwf_lds_bytes(int mode, int nt, int pt, int Kpad, int kc, int nkc, int wres) {
    const int BP = 64 * pt, NB = nt * 16, pitch = kc / 8 + 1;
    size_t b = (size_t)(mode == 1 ? 5 : 2) * Kpad * 4;
    if (mode == 1) b += (size_t)4 * NB * 4;
    b += (size_t)(mode == 1 ? 3 : 2) * BP * pitch * 16;
    b += (size_t)(wres ? nkc : 2) * NB * pitch * 16;
    size_t o = (size_t)2 * BP * (NB / 8 + 1) * 16, red = mode == 1 ? (size_t)256 * 16 * 4 : (size_t)8 * NB * 4;
    b += o > red ? o : red;
    return b;
}

// cout tiles per block: whole output when it fits 6 tiles, otherwise the block size (<= 6) with the least padding
static bool pwf_plan(int M, int Ci, int Co, PwfPlan* p) {
    if ((Ci & 7) || (Co & 7) || Ci < 8 || Co < 8 || M < 1) return false;
    // widening convs with a single K chunk only (the expand convs up to 120 input channels): measured against k_igemm at
    // bs 256 (tools/kbench.py) 120 vs 166 us (16->48 @112^2), 50 vs 76 (40->240 @28^2), 38 vs 55 (80->480 @14^2), 46 vs 73
    // (96->576); the narrowing convs (long K, few output channels) and 192->1152 stay on k_igemm, which is 15-40 % faster there
    if (mnas_pwf_enabled() < 2 && !(Co > Ci && Ci <= 128)) return false;
    const int tiles = (Co + 15) / 16, Kpad = (Ci + 31) / 32 * 32;
    int nt = tiles, blocks = 1;
    if (tiles > 6) {
        int best = 1 << 30;
        for (int c = 6; c >= 4; --c) {
            const int b = (tiles + c - 1) / c, waste = b * c - tiles;
            if (waste < best) { best = waste; nt = c; blocks = b; }
        }
    }
    p->nt = nt; p->nblocks = blocks;
    p->kc = Kpad <= 128 ? Kpad : 128;
    p->nkc = (Kpad + p->kc - 1) / p->kc;
    // pixels per tile: 128 where the tensor is large (fewer, longer phases), 64 on the small-M stages (more workgroups in flight)
    p->pt = ((int64_t)M >= 400000) ? 2 : 1;
    // whole weight block resident when it is small (single-chunk layers, and multi-chunk ones up to 48 KB)
    const size_t wbytes = (size_t)p->nkc * nt * 16 * (p->kc / 8 + 1) * 16;
    p->wres = wbytes <= 48 * 1024 ? 1 : 0;
    p->lds = pwf_lds_bytes(0, nt, p->pt, Kpad, p->kc, p->nkc, p->wres);
    if (p->lds > 96 * 1024 && p->wres) { p->wres = 0; p->lds = pwf_lds_bytes(0, nt, p->pt, Kpad, p->kc, p->nkc, 0); }
    if (p->lds > 96 * 1024 && p->pt == 2) { p->pt = 1; p->lds = pwf_lds_bytes(0, nt, 1, Kpad, p->kc, p->nkc, p->wres); }
    return p->lds <= 160 * 1024;
}

// MODE 1: K = dy channels (<= 128, one chunk, weights resident), Co = conv input channels > K.  Channel block = the tile
// count (<= 6) with the least padding among those that leave room for two workgroups per CU (80 KB each).
static bool pwd_plan(int M, int Ci, int Co, PwfPlan* p) {
    if ((Ci & 7) || (Co & 7) || Ci < 8 || Co < 8 || M < 1 || !mnas_pwf_enabled()) return false;
    if (!(Co > Ci && Ci <= 128) || (int64_t)M * Co >= (1ll << 31)) return false;
    const int tiles = (Co + 15) / 16, Kpad = (Ci + 31) / 32 * 32;
    p->kc = Kpad; p->nkc = 1; p->wres = 1;
    p->pt = ((int64_t)M >= 400000) ? 2 : 1;
    for (; p->pt >= 1; --p->pt) {
        int best = 1 << 30;
        p->nt = 0;
        for (int c = tiles < 6 ? tiles : 6; c >= 1; --c) {
            const size_t lds = pwf_lds_bytes(1, c, p->pt, Kpad, Kpad, 1, 1);
            if (lds > 80 * 1024) continue;
            const int b = (tiles + c - 1) / c, waste = b * c - tiles;
            if (waste < best) { best = waste; p->nt = c; p->nblocks = b; p->lds = lds; }
        }
        if (p->nt) return true;
    }
    return false;
}

static int pwf_grid(const PwfPlan& p, int M) {
    const int ntiles = (M + 64 * p.pt - 1) / (64 * p.pt);
    const int per_cu = (int)(160 * 1024 / p.lds) < 1 ? 1 : (int)(160 * 1024 / p.lds);
    int want = 256 * (per_cu > 3 ? 3 : per_cu) / p.nblocks;      // fill the chip once with resident workgroups
    if (want < 64) want = 64;
    if (want > 1024) want = 1024;
    return ntiles < want ? ntiles : want;
}

// persistent workgroups along the pixel dimension for a 1x1 forward with M pixels (host-side, no launch)
int mnas_pwf_parts(int M, int Ci, int Co) {
    PwfPlan p;
    if (!pwf_plan(M, Ci, Co, &p)) return -1;
    return pwf_grid(p, M);
}
// same for the input gradient (Ci = dy channels, Co = channels of the result)
int mnas_pwd_parts(int M, int Ci, int Co) {
    PwfPlan p;
    if (!pwd_plan(M, Ci, Co, &p)) return -1;
    // exactly the resident capacity (two workgroups per CU), never a second partial round of workgroups; a multiple of 8:
    // the channel blocks of one pixel tile then land on the same XCD (shared L2)
    const int ntiles = (M + 64 * p.pt - 1) / (64 * p.pt);
    int g = 256 * (int)(160 * 1024 / p.lds) / p.nblocks;
    g = mnas_diag_env("MNAS_PWD_GRID", g);
    if (g > ntiles) g = ntiles;
    if (g > 8) g &= ~7;
    return g < 1 ? 1 : g;
}

template <int MODE, int NT, int PT>
static int pwf_launch2(const PwfArgs& a, const PwfPlan& p, int nparts, hipStream_t s) {
    if (p.wres) hipLaunchKernelGGL((k_pwf<MODE, NT, PT, true>), dim3(nparts, p.nblocks), dim3(256), p.lds, s, a);
    else if (MODE == 0) hipLaunchKernelGGL((k_pwf<0, NT, PT, false>), dim3(nparts, p.nblocks), dim3(256), p.lds, s, a);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// called by mnas_conv_gemm for mode 0, 1x1, stride 1
int mnas_pwf_forward(const MnasConvGemm* c, void* stream) {
    PwfPlan p;
    const int M = c->N * c->Ho * c->Wo;
    if (!pwf_plan(M, c->Ci, c->Co, &p)) return MNAS_EINVAL;
    PwfArgs a;
    a.M = M; a.Ci = c->Ci; a.Co = c->Co;
    a.Kpad = (c->Ci + 31) / 32 * 32;
    a.kc = p.kc; a.nkc = p.nkc; a.nch = p.kc / 8;
    a.co_pad16 = (c->Co + 15) / 16 * 16;
    a.nt_store = (mnas_nt_mask() & MNAS_NT_PWF) ? 1 : 0;
    a.abl = mnas_diag_env("MNAS_PWF_ABL", 0);
    a.act = c->act; a.w = (const uint16_t*)c->w; a.bias = c->bias; a.out = c->out; a.stats = c->stats;
    a.grad.g = nullptr; a.grad.y = nullptr; a.grad.coef = nullptr; a.red_y = nullptr; a.red_bn = nullptr;
    hipStream_t s = (hipStream_t)stream;
#define MNAS_PWF(NT_) if (p.nt == NT_) return p.pt == 2 ? pwf_launch2<0, NT_, 2>(a, p, c->nparts, s) : pwf_launch2<0, NT_, 1>(a, p, c->nparts, s);
    MNAS_PWF(1) MNAS_PWF(2) MNAS_PWF(3) MNAS_PWF(4) MNAS_PWF(5) MNAS_PWF(6)
#undef MNAS_PWF
    return MNAS_EINVAL;
}

// called by mnas_conv_gemm for mode 1, 1x1, stride 1, no residual (MNAS_EINVAL: not this kernel's shape, caller falls back)
int mnas_pwd_dgrad(const MnasConvGemm* c, void* stream) {
    PwfPlan p;
    const int M = c->N * c->Ho * c->Wo;
    if (c->resid || c->bias || !pwd_plan(M, c->Ci, c->Co, &p)) return MNAS_EINVAL;
    if ((c->grad.coef == nullptr) != (c->grad.y == nullptr)) return MNAS_EINVAL;
    PwfArgs a;
    a.M = M; a.Ci = c->Ci; a.Co = c->Co;
    a.Kpad = (c->Ci + 31) / 32 * 32;
    a.kc = p.kc; a.nkc = 1; a.nch = p.kc / 8;
    a.co_pad16 = (c->Co + 15) / 16 * 16;
    a.nt_store = (mnas_nt_mask() & MNAS_NT_PWF) ? 1 : 0;
    a.abl = mnas_diag_env("MNAS_PWF_ABL", 0);
    a.act.data = nullptr; a.act.scale = nullptr; a.act.shift = nullptr;
    a.w = (const uint16_t*)c->w; a.bias = nullptr; a.out = c->out;
    a.grad = c->grad; a.red_y = c->red_y; a.red_bn = c->red_bn;
    a.stats = c->red_y ? c->stats : nullptr;
    hipStream_t s = (hipStream_t)stream;
#define MNAS_PWD(NT_) if (p.nt == NT_) return p.pt == 2 ? pwf_launch2<1, NT_, 2>(a, p, c->nparts, s) : pwf_launch2<1, NT_, 1>(a, p, c->nparts, s);
    MNAS_PWD(1) MNAS_PWD(2) MNAS_PWD(3) MNAS_PWD(4) MNAS_PWD(5) MNAS_PWD(6)
#undef MNAS_PWD
    return MNAS_EINVAL;
}
