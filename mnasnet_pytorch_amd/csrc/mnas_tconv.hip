// Input gradient of the stride-2 dense 3x3 convs (mnasnet.py:157-161 with reduce=True: the stage-transition ConvBlock) as a
// TRANSPOSED convolution on the matrix cores, without any per-element gather:
//     gin[n, 2i+ph, 2j+pw, ci] = sum over the taps of parity class (ph, pw) of dy[n, i+dh, j+dw, :] . W[:, ci, kh, kw]
// An output pixel only receives 1, 2, 2 or 4 of the 9 taps, and all four classes of the 2x2 output block (2i.., 2j..) read the SAME
// four dy pixels d00 = dy(i,j), d01 = dy(i,j+1), d10 = dy(i+1,j), d11 = dy(i+1,j+1):
//     (0,0): d00 W11            (0,1): d01 W10 + d00 W12            (1,0): d10 W01 + d00 W21
//     (1,1): d11 W00 + d10 W02 + d01 W20 + d00 W22
// So ONE GEMM per "super-pixel" (i, j):  [d00 | d01 | d10 | d11] (K = 4 Co)  x  Wt[K][4 Ci]  ->  the 2x2 block of gin, with Wt the
// block matrix above (7 of its 16 blocks are zero: MFMA work is not what bounds these layers).  dy must be MATERIALISED
// (mnas_dy_materialize): the four K segments of a row are four contiguous Co-channel runs of dy and go HBM/L2 -> LDS by
// global_load_lds, exactly like k_pwf's activation tile (same phase pipeline: DMA of tile n+1 under the MFMAs of tile n, weights
// LDS-resident, LDS out-stage).  The out-stage leaves as 16-byte stores scattered to the four output pixels of each block; the
// fused BatchNorm-backward reduce of the ConvBlock that produced the conv's input reads its raw output with the same 16-byte
// pattern.  k_igemm's parity-class form did the same arithmetic with a per-slot address decode: 26 k VALU instructions per wave,
// 195 us for 24->16 at 112x112 against a 52 us bandwidth floor.
#include "mnas_common.h"

typedef __attribute__((address_space(3))) void* tc_lds_ptr;
typedef const __attribute__((address_space(1))) void* tc_gbl_ptr;

struct TconvArgs {
    int M2;                  // super-pixels = N * Ho * Wo (dy pixels)
    int Ho, Wo, Co, Ci;      // dy plane / channels, gin channels; gin plane = 2Ho x 2Wo
    int K, Kpad, nch;        // 4*Co, rounded to 32, Kpad/8
    int N4;                  // 4*Ci
    float rcp_hw, rcp_wo;
    const uint16_t* dy;      // bf16 (N,Ho,Wo,Co), materialised
    const uint16_t* w;       // MNAS_PACK_TCONV: [N4 rounded to 16][Kpad]
    void* out;               // bf16 (N,2Ho,2Wo,Ci)
    float* stats;            // fused reduce partials [2][Ci][gridDim.x] or NULL
    const void* red_y;
    const float* red_bn;
};

__device__ __forceinline__ int tc_fdiv(int n, int d, float rcp) {
    if (rcp == 0.f) return n / d;
    int q = (int)((float)n * rcp);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

template <int NT, int PT>
__global__ __launch_bounds__(256) void k_tconv(TconvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BP = 64 * PT, NB = NT * 16;
    constexpr int OPITCH = NB / 8 + 1;
    constexpr int MAXA = (BP * 33 + 255) / 256;              // pitch <= 33 (K <= 256)
    const int pitch = a.nch + 1;
    uint4* lds_a = (uint4*)smem;                             // [2][BP * pitch]
    uint4* lds_w = lds_a + 2 * BP * pitch;                   // [NB * pitch]
    uint4* lds_o = lds_w + NB * pitch;                       // [2][BP * OPITCH]
    float* lds_redc = (float*)(lds_o + 2 * BP * OPITCH);     // [4][Ci]
    float* lds_fin = (float*)lds_o;                          // reused at the end: [256][16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int co8 = a.Co >> 3, ci8 = a.Ci >> 3, nch8 = a.N4 >> 3;
    const bool do_red = a.red_y != nullptr;
    const int na = (BP * pitch + 255) >> 8;

    // ---- slot plan: slot q = 256*i + tid = row * pitch + j;  j -> (neighbour nb = j / (Co/8), channel chunk)
    int pa[MAXA], ja[MAXA];
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
        const int q = 256 * i + tid;
        pa[i] = q / pitch; ja[i] = q - pa[i] * pitch;
        if (pa[i] >= BP || ja[i] * 8 >= a.K) ja[i] = -1;
    }
    {
        const int nz = (2 * BP + NB) * pitch;
        for (int i = tid; i < nz; i += 256) lds_a[i] = make_uint4(0, 0, 0, 0);
        if (do_red)
            for (int i = tid; i < 4 * a.Ci; i += 256) {
                const int r = i / a.Ci, c = i - r * a.Ci;
                float v;
                if (r == 0) v = a.red_bn[0 * a.Ci + c];
                else if (r == 1) v = a.red_bn[1 * a.Ci + c];
                else if (r == 2) v = a.red_bn[6 * a.Ci + c];
                else v = -a.red_bn[5 * a.Ci + c] * a.red_bn[6 * a.Ci + c];
                lds_redc[i] = v;
            }
    }
    __syncthreads();
    // weights: resident
    for (int q = tid; q < NB * pitch; q += 256) {
        const int r = q / pitch, j = q - r * pitch;
        if (j < a.nch && r < ((a.N4 + 15) & ~15)) lds_w[q] = *(const uint4*)(a.w + (size_t)r * a.Kpad + j * 8);
    }

    // validity of the DMA of phase `slot` (bit i = slot plan entry i landed real data), per thread
    unsigned ok0 = 0u, ok1 = 0u;          // two scalars, not an array: a run-time index would send it to scratch
    auto dma_a = [&](int slot, int tile0) {
        uint4* dst = lds_a + slot * BP * pitch;
        unsigned ok = 0;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            if (i >= na) break;
            if (ja[i] < 0) continue;
            const int m = tile0 + pa[i];
            if (m >= a.M2) continue;
            const int hw = a.Ho * a.Wo;
            const int n = tc_fdiv(m, hw, a.rcp_hw), rem = m - n * hw;
            const int ii = tc_fdiv(rem, a.Wo, a.rcp_wo), jj = rem - ii * a.Wo;
            const int nb = ja[i] / co8, c = ja[i] - nb * co8;
            const int y = ii + (nb >> 1), x = jj + (nb & 1);
            if (y < a.Ho && x < a.Wo) {
                ok |= 1u << i;
                __builtin_amdgcn_global_load_lds((tc_gbl_ptr)(a.dy + (((size_t)n * a.Ho + y) * a.Wo + x) * a.Co + c * 8),
                                                 (tc_lds_ptr)(dst + 256 * i + wave * 64), 16, 0, 0);
            }
        }
        if (slot) ok1 = ok; else ok0 = ok;
    };
    // slots whose neighbour lies outside the dy plane read as zero (the DMA skipped them: clear what an older tile left there)
    auto clear_a = [&](int slot) {
        uint4* t = lds_a + slot * BP * pitch;
        const unsigned ok = slot ? ok1 : ok0;
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            if (i >= na) break;
            if (ja[i] >= 0 && !((ok >> i) & 1u)) t[256 * i + tid] = make_uint4(0, 0, 0, 0);
        }
    };
    // copy-out role: thread -> fixed 16-byte column c8 of the out-stage row (class = c8 / (Ci/8), channel chunk = c8 % (Ci/8))
    const int tcols = (256 / nch8) * nch8;                   // threads that take part
    const int oc8 = tid % nch8, orow0 = tid / nch8, orows = tcols / nch8;
    const int ocls = oc8 / ci8, occ = oc8 - ocls * ci8;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    // The output offsets of the rows this thread copies out, and the fused-reduce operand (raw output of the target layer at
    // those offsets), are prepared ONE PHASE AHEAD: plan_out(t) runs in tile t's own phase (decode + 16-byte loads in flight
    // under the MFMAs), copy_out consumes them in the next phase -- no load latency and no address decode on the copy-out path.
    constexpr int MAXR = PT == 1 ? 4 : 7;                    // rows per thread: ceil(BP / (tcols / nch8)), nch8 <= 12
    int ooff[MAXR];
    uint4 yreg[MAXR];
    auto plan_out = [&](int tile0) {
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            const int p = orow0 + k * orows, m = tile0 + p;
            ooff[k] = -1;
            yreg[k] = make_uint4(0, 0, 0, 0);
            if (tid < tcols && p < BP && m < a.M2) {
                const int hw = a.Ho * a.Wo;
                const int n = tc_fdiv(m, hw, a.rcp_hw), rem = m - n * hw;
                const int ii = tc_fdiv(rem, a.Wo, a.rcp_wo), jj = rem - ii * a.Wo;
                ooff[k] = (((n * 2 * a.Ho + 2 * ii + (ocls >> 1)) * 2 * a.Wo + 2 * jj + (ocls & 1)) * a.Ci + occ * 8);
                if (do_red) yreg[k] = *(const uint4*)((const uint16_t*)a.red_y + ooff[k]);
            }
        }
    };
    auto copy_out = [&](int obuf) {
        const uint4* o = lds_o + obuf * BP * OPITCH;
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            if (ooff[k] < 0) continue;
            const int p = orow0 + k * orows;
            const uint4 pk = o[p * OPITCH + oc8];
            *(uint4*)((uint16_t*)a.out + ooff[k]) = pk;
            if (do_red) {
                float gq[8], yq[8];
                unpack8(pk, gq);
                unpack8(yreg[k], yq);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = occ * 8 + j;
                    const float dz = (fmaf(yq[j], lds_redc[c], lds_redc[a.Ci + c]) > 0.f) ? gq[j] : 0.f;
                    s1[j] += dz;
                    s2[j] = fmaf(dz, fmaf(yq[j], lds_redc[2 * a.Ci + c], lds_redc[3 * a.Ci + c]), s2[j]);
                }
            }
        }
    };

    const int ntiles = (a.M2 + BP - 1) / BP;
    if ((int)blockIdx.x < ntiles) dma_a(0, blockIdx.x * BP);
    int prev_tile0 = -1, obuf = 0, ph = 0;
    const int ksteps = a.Kpad >> 5;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++ph) {
        const int slot = ph & 1, tile0 = t * BP;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        clear_a(slot);
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) dma_a(slot ^ 1, (t + gridDim.x) * BP);
        if (prev_tile0 >= 0) { copy_out(obuf ^ 1); prev_tile0 = -1; }
        plan_out(tile0);
        f32x4_t acc[PT][NT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[pt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const uint4* ta = lds_a + slot * BP * pitch;
        for (int ks = 0; ks < ksteps; ++ks) {
            bf16x8_t bfrag[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                bfrag[pt] = *(const bf16x8_t*)(ta + ((wave * PT + pt) * 16 + l15) * pitch + ks * 4 + lg);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8_t afrag = *(const bf16x8_t*)(lds_w + (nt * 16 + l15) * pitch + ks * 4 + lg);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[pt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[pt], acc[pt][nt], 0, 0, 0);
            }
        }
        uint4* o = lds_o + obuf * BP * OPITCH;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int p = (wave * PT + pt) * 16 + l15;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                uint2 pk;
                pk.x = pack_bf16(acc[pt][nt][0], acc[pt][nt][1]);
                pk.y = pack_bf16(acc[pt][nt][2], acc[pt][nt][3]);
                *(uint2*)((unsigned char*)(o + p * OPITCH) + nt * 32 + lg * 8) = pk;
            }
        }
        prev_tile0 = tile0;
        obuf ^= 1;
    }
    __syncthreads();
    if (prev_tile0 >= 0) copy_out(obuf ^ 1);

    if (do_red && a.stats) {
        // per-thread sums -> per-channel: the threads of one channel chunk (all classes, all row lanes) added in thread order
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) { lds_fin[tid * 16 + j] = s1[j]; lds_fin[tid * 16 + 8 + j] = s2[j]; }
        __syncthreads();
        for (int i = tid; i < 2 * a.Ci; i += 256) {
            const int r = i / a.Ci, c = i - r * a.Ci;
            const int cc = c >> 3, j = c & 7;
            float v = 0.f;
            for (int th = 0; th < tcols; ++th)
                if ((th % nch8) % ci8 == cc) v += lds_fin[th * 16 + r * 8 + j];
            a.stats[((size_t)r * a.Ci + c) * gridDim.x + blockIdx.x] = v;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
static bool tconv_ok(int Ho, int Wo, int Co, int Ci, int* nt, int* pt, size_t* lds, long long M2) {
    if ((Co & 7) || (Ci & 7) || Co < 8 || Ci < 8 || 4 * Co > 256 || 4 * Ci > 96 || Ho < 1 || Wo < 1) return false;
    const int Kpad = (4 * Co + 31) / 32 * 32, pitch = Kpad / 8 + 1;
    *nt = (4 * Ci + 15) / 16;
    for (*pt = (M2 >= 400000 ? 2 : 1); *pt >= 1; --*pt) {          // 128-pixel tiles on the large layers when they fit two per CU
        const int BP = 64 * *pt, NB = *nt * 16;
        const size_t o = (size_t)2 * BP * (NB / 8 + 1) * 16, fin = (size_t)256 * 16 * 4;
        *lds = (size_t)(2 * BP + NB) * pitch * 16 + (o > fin ? o : fin) + (size_t)4 * Ci * 4;
        // two workgroups per CU or nothing: with one (40->24 at 56x56: 102 KB) the phase pipeline is 107 us against k_igemm's 81
        if (*lds <= 80 * 1024) return true;
    }
    return false;
}
extern "C" int mnas_tconv_supported(int Ho, int Wo, int Co, int Ci) {
    if (mnas_tcx_ok(Ho, Wo, Co, Ci) || mnas_tcr_ok(Ho, Wo, Co, Ci)) return 1;      // the weight-stationary kernels (csrc/mnas_tcx.hip)
    int nt, pt; size_t lds;
    return tconv_ok(Ho, Wo, Co, Ci, &nt, &pt, &lds, 1 << 20) ? 1 : 0;
}
extern "C" int mnas_tconv_parts(int N, int Ho, int Wo, int Co, int Ci) {
    {
        int r = mnas_tcx_parts(N, Ho, Wo, Co, Ci);
        if (r > 0) return r;
        r = mnas_tcr_parts(N, Ho, Wo, Co, Ci);
        if (r > 0) return r;
    }
    int nt, pt; size_t lds;
    const long long M2 = (long long)N * Ho * Wo;
    if (M2 * 4 * Ci > 0x7fffffff || !tconv_ok(Ho, Wo, Co, Ci, &nt, &pt, &lds, M2)) return -1;
    const int ntiles = (int)((M2 + 64 * pt - 1) / (64 * pt));
    int per_cu = (int)(160 * 1024 / lds);
    if (per_cu > 3) per_cu = 3;
    if (per_cu < 1) per_cu = 1;
    const int want = 256 * per_cu;
    return ntiles < want ? ntiles : want;
}

extern "C" int mnas_tconv_dgrad(const MnasTconvDgrad* c, void* stream) {
    if (!c || !c->dy || !c->w || !c->out || c->nparts < 1) return MNAS_EINVAL;
    if (c->red_y && (!c->red_bn || !c->stats)) return MNAS_EINVAL;
    if (mnas_tcx_parts(c->N, c->Ho, c->Wo, c->Co, c->Ci) > 0) return mnas_tcx_dgrad(c, stream);    // (incl. the 32-bit offset limit)
    if (mnas_tcr_parts(c->N, c->Ho, c->Wo, c->Co, c->Ci) > 0) return mnas_tcr_dgrad(c, stream);
    int nt, pt; size_t lds;
    const long long M2 = (long long)c->N * c->Ho * c->Wo;
    if (M2 * 4 * c->Ci > 0x7fffffff || !tconv_ok(c->Ho, c->Wo, c->Co, c->Ci, &nt, &pt, &lds, M2)) return MNAS_EINVAL;
    TconvArgs a;
    a.M2 = (int)M2; a.Ho = c->Ho; a.Wo = c->Wo; a.Co = c->Co; a.Ci = c->Ci;
    a.K = 4 * c->Co; a.Kpad = (a.K + 31) / 32 * 32; a.nch = a.Kpad / 8; a.N4 = 4 * c->Ci;
    a.rcp_hw = M2 < (1 << 24) ? 1.0f / (float)(c->Ho * c->Wo) : 0.f;
    a.rcp_wo = M2 < (1 << 24) ? 1.0f / (float)c->Wo : 0.f;
    a.dy = (const uint16_t*)c->dy; a.w = (const uint16_t*)c->w; a.out = c->out; a.stats = c->stats;
    a.red_y = c->red_y; a.red_bn = c->red_bn;
    hipStream_t s = (hipStream_t)stream;
#define MNAS_TC(NT_) if (nt == NT_) { \
        if (pt == 2) hipLaunchKernelGGL((k_tconv<NT_, 2>), dim3(c->nparts), dim3(256), lds, s, a); \
        else hipLaunchKernelGGL((k_tconv<NT_, 1>), dim3(c->nparts), dim3(256), lds, s, a); \
        MNAS_CHECK_LAUNCH(); return MNAS_OK; }
    MNAS_TC(2) MNAS_TC(3) MNAS_TC(4) MNAS_TC(5) MNAS_TC(6)
#undef MNAS_TC
    return MNAS_EINVAL;
}
