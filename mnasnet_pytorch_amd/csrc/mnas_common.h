// Device-side helpers shared by the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mnas.h"

#define MNAS_WAVE 64
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16 = 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 C/D fragment

// Diagnosis switches (A/B runs of kernel variants, ablations that skip work) exist only in a -DMNAS_DIAG build
// (tools/build_alt.sh -> libmnas_hip_alt.so).  The shipped library reads NO environment variable and has no mutable
// global state beyond these compile-time constants (include/mnas.h contract).
#ifdef MNAS_DIAG
#include <cstdlib>
static inline int mnas_diag_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
#define mnas_diag_env(name, dflt) (dflt)
#endif

#define MNAS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// MNAS_EARLY (1): a kernel's FIRST tile / group loads are issued before its one-time setup (coefficient tables, resident weights)
// instead of behind the setup's barrier: one exposed memory round trip less per workgroup (0: A/B builds)
#ifndef MNAS_EARLY
#define MNAS_EARLY 1
#endif
// dst[i] = f(i), i = tid, tid + nth, ... < n with FOUR values fetched before the first is written: a setup table of a few
// iterations per thread costs one memory round trip instead of one per iteration (MNAS_EARLY = 0: the plain loop)
template <class F>
__device__ __forceinline__ void mnas_fill_table(float* dst, int n, int tid, int nth, F f) {
#if MNAS_EARLY
    for (int i0 = tid; i0 < n; i0 += 4 * nth) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i0 + j * nth < n) ? f(i0 + j * nth) : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i0 + j * nth < n) dst[i0 + j * nth] = v[j];
    }
#else
    for (int i = tid; i < n; i += nth) dst[i] = f(i);
#endif
}
// row r of a fused-reduce coefficient table (s, t, invstd, -mean*invstd of channel c, from a bnbuf of C channels): both loads
// unconditional -- as a chain of if / else every row was its own branch with its own wait
__device__ __forceinline__ float mnas_red_coef(const float* red_bn, int C, int r, int c) {
    const int row = r == 0 ? 0 : (r == 1 ? 1 : (r == 2 ? 6 : 5));
    const float v = red_bn[(size_t)row * C + c], w = red_bn[(size_t)6 * C + c];
    return r == 3 ? -v * w : v;
}
// the same for 16-byte items: store(i, load(i)), i = tid, tid + nth, ... < n, four loads in flight before the first store
template <class L, class S>
__device__ __forceinline__ void mnas_copy_items(int n, int tid, int nth, L load, S store) {
#if MNAS_EARLY
    for (int i0 = tid; i0 < n; i0 += 4 * nth) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i0 + j * nth < n) ? load(i0 + j * nth) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i0 + j * nth < n) store(i0 + j * nth, v[j]);
    }
#else
    for (int i = tid; i < n; i += nth) store(i, load(i));
#endif
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf_to_f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// two fp32 -> packed bf16x2, round-to-nearest-even (v_cvt_pk_bf16_f32: lo = src0, hi = src1; verified
// on gfx950 by tools/probe).
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ uint16_t f_to_bf(float f) { return (uint16_t)(pack_bf16(f, 0.f) & 0xffffu); }

// unpack 8 bf16 (one 16-byte channel group) to fp32
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16(f[0], f[1]); v.y = pack_bf16(f[2], f[3]);
    v.z = pack_bf16(f[4], f[5]); v.w = pack_bf16(f[6], f[7]);
    return v;
}

// Channel-pair math in explicit float2: v_pk_fma_f32 does two FMAs per issue slot.  The build runs with -fno-slp-vectorize (the
// SLP pass packs arbitrary neighbours and pays for it in register shuffles), so packing has to be written out where it pays:
// the staging transforms below run once per 16 bytes in every GEMM-class kernel (round 3 SQ counters: 20-25 % of their wave
// cycles are vector-ALU instructions, most of them these transforms and the fused reduces).  Same IEEE operations as the scalar
// form -> bit-identical results.
typedef float mnas_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mnas_f2 mnas_f2fma(mnas_f2 a, mnas_f2 b, mnas_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ mnas_f2 mnas_bf2(uint32_t u) { mnas_f2 r; r.x = bf_lo(u); r.y = bf_hi(u); return r; }
__device__ __forceinline__ mnas_f2 mnas_ld2(const float* p) { mnas_f2 r; r.x = p[0]; r.y = p[1]; return r; }

// BatchNorm partial statistics of one channel pair: s1 += v, s2 += v*v
__device__ __forceinline__ void mnas_stat2(mnas_f2 v, float* s1, float* s2) {
    const mnas_f2 a = mnas_ld2(s1) + v, b = mnas_f2fma(v, v, mnas_ld2(s2));
    s1[0] = a.x; s1[1] = a.y; s2[0] = b.x; s2[1] = b.y;
}
// fused BatchNorm-backward reduce of one channel pair: dz = g*[s*y+t>0] (g, y: packed bf16 pairs as stored), r1 += dz,
// r2 += dz*(y*inv + m)   (inv = invstd, m = -mean*invstd)
__device__ __forceinline__ void mnas_red2(uint32_t gu, uint32_t yu, mnas_f2 s, mnas_f2 t, mnas_f2 inv, mnas_f2 m, float* r1, float* r2) {
    const mnas_f2 g = mnas_bf2(gu), y = mnas_bf2(yu);
    const mnas_f2 z = mnas_f2fma(y, s, t);
    mnas_f2 dz;
    dz.x = (z.x > 0.f) ? g.x : 0.f;
    dz.y = (z.y > 0.f) ? g.y : 0.f;
    const mnas_f2 a = mnas_ld2(r1) + dz, b = mnas_f2fma(dz, mnas_f2fma(y, inv, m), mnas_ld2(r2));
    r1[0] = a.x; r1[1] = a.y; r2[0] = b.x; r2[1] = b.y;
}

// act-on-load for one channel group: relu(s*x+t)
__device__ __forceinline__ uint4 act8(const uint4& raw, const float* s, const float* t) {
    const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const mnas_f2 z = mnas_f2fma(mnas_bf2(u[j]), mnas_ld2(s + 2 * j), mnas_ld2(t + 2 * j));
        o[j] = pack_bf16(fmaxf(z.x, 0.f), fmaxf(z.y, 0.f));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// gated act-on-load (squeeze-excite applied on load, csrc/mnas_se.hip): relu(s*x+t) * g, g = the per-(image, channel) excitation
__device__ __forceinline__ uint4 act8g(const uint4& raw, const float* s, const float* t, const float* g) {
    const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const mnas_f2 z = mnas_f2fma(mnas_bf2(u[j]), mnas_ld2(s + 2 * j), mnas_ld2(t + 2 * j));
        const mnas_f2 r = {fmaxf(z.x, 0.f), fmaxf(z.y, 0.f)};
        const mnas_f2 q = r * mnas_ld2(g + 2 * j);
        o[j] = pack_bf16(q.x, q.y);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// dy-on-load for one channel group: c1*(g*[s*y+t>0]) + c2*y + c3 ; coef rows are s,t,c1,c2,c3
__device__ __forceinline__ void dy8(const uint4& graw, const uint4& yraw, const float* s, const float* t,
                                    const float* c1, const float* c2, const float* c3, float* out) {
    const uint32_t gu[4] = {graw.x, graw.y, graw.z, graw.w}, yu[4] = {yraw.x, yraw.y, yraw.z, yraw.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const mnas_f2 g = mnas_bf2(gu[j]), y = mnas_bf2(yu[j]);
        const mnas_f2 z = mnas_f2fma(y, mnas_ld2(s + 2 * j), mnas_ld2(t + 2 * j));
        mnas_f2 dz;
        dz.x = (z.x > 0.f) ? g.x : 0.f;
        dz.y = (z.y > 0.f) ? g.y : 0.f;
        const mnas_f2 d = mnas_f2fma(mnas_ld2(c1 + 2 * j), dz, mnas_f2fma(mnas_ld2(c2 + 2 * j), y, mnas_ld2(c3 + 2 * j)));
        out[2 * j] = d.x; out[2 * j + 1] = d.y;
    }
}

// Barrier that PUBLISHES LDS-DMA data (global_load_lds): every wave first waits for its own outstanding DMA (vmcnt(0)),
// then joins the barrier.  A bare __syncthreads() is not enough: hipcc only inserts the vmcnt wait in front of the wave's
// own first LDS read AFTER the barrier, so a wave could read chunks whose DMA -- issued by ANOTHER wave -- had not landed
// yet (seen as run-to-run differences / NaNs from stale LDS at full problem sizes, never in small tests).
__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// Stores of the large activation / gradient tensors.  nt = nontemporal (streaming) store: the written tensor is not read
// again by this kernel; measured on MI355X (tools/probe/bw.hip) a 1-read : 3-write stream sustains 3.3 TB/s with plain
// stores and 4.5-5.0 TB/s with nontemporal ones (1:1 4.7 -> 5.1; read-heavy mixes unchanged).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_u1(void* p, uint32_t v, bool nt) {
    if (nt) __builtin_nontemporal_store(v, (uint32_t*)p); else *(uint32_t*)p = v;
}
__device__ __forceinline__ void st_u2(void* p, uint2 v, bool nt) {
    if (nt) __builtin_nontemporal_store((u32x2_t){v.x, v.y}, (u32x2_t*)p); else *(uint2*)p = v;
}
__device__ __forceinline__ void st_u4(void* p, uint4 v, bool nt) {
    if (nt) __builtin_nontemporal_store((u32x4_t){v.x, v.y, v.z, v.w}, (u32x4_t*)p); else *(uint4*)p = v;
}
// MNAS_NT environment bit mask (diagnosis / A-B runs; unset = the tuned default): which kernel classes store nontemporally
#define MNAS_NT_IGEMM_FWD 1
#define MNAS_NT_IGEMM_DGRAD 2
#define MNAS_NT_DW_FWD 4
#define MNAS_NT_DW_BWD 8
#define MNAS_NT_PW_BWD 16
#define MNAS_NT_ADD_ACT 32
#define MNAS_NT_STEM 64
#define MNAS_NT_PWF 128
int mnas_nt_mask();
// DMA-pipelined 1x1 forward (mnas_pwf.hip): mnas_conv_gemm dispatches mode 0 / 1x1 here unless MNAS_PWF=0
int mnas_pwf_enabled();
int mnas_pwf_parts(int M, int Ci, int Co);
int mnas_pwf_forward(const MnasConvGemm* c, void* stream);
int mnas_stem_fwd_band(const MnasStemFwd* c, void* stream);      // csrc/mnas_stem.hip; MNAS_EINVAL = not a band shape
int mnas_stem_wgrad_band(const MnasStemWgrad* c, void* stream);
int mnas_dimg_parts(int mode, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad);
int mnas_dimg_run(const MnasConvGemm* c, void* stream);          // csrc/mnas_dimg.hip: dense 3x3 on the small feature maps
int mnas_c3r_parts(int mode, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad);
int mnas_c3r_run(const MnasConvGemm* c, void* stream);           // csrc/mnas_c3r.hip: weight-heavy dense 3x3 on the 7x7 planes, weights register-resident
int mnas_pwd_enabled();      // MNAS_PWD (default 1): DMA-pipelined 1x1 input gradient for the few-dy-channel convs
int mnas_pwd_parts(int M, int Ci, int Co);
int mnas_pwd_dgrad(const MnasConvGemm* c, void* stream);
// K-streaming 1x1 GEMM (mnas_pws.hip): long-K forward / input gradient on the small-M stages (diagnosis build: MNAS_PWS=0 off, 1 forward only)
int mnas_dw2_rows(int N, int H, int W, int C, int k, int nparts, int which);      // csrc/mnas_dw2.hip: stride-2 depthwise (SepConv reduce)
int mnas_dw2_fwd(const MnasDwFwd* c, void* stream);
int mnas_dw2_bwd(const MnasDwBwd* c, void* stream);
int mnas_pws_enabled();
int mnas_pws_parts(int mode, int M, int K, int N);
int mnas_pws_run(const MnasConvGemm* c, void* stream);
int mnas_pws_gate_ok(int M, int K, int N);
int mnas_c3x_ok(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad);
int mnas_c3x_run(const MnasConvGemm* c, void* stream);           // csrc/mnas_c3x.hip: stride-2 3x3 forward on the large maps, weight-stationary
int mnas_tcx_ok(int Ho, int Wo, int Co, int Ci);                 // csrc/mnas_tcx.hip: stride-2 3x3 input gradient, weight-stationary
int mnas_tcx_parts(int N, int Ho, int Wo, int Co, int Ci);
int mnas_tcx_dgrad(const MnasTconvDgrad* c, void* stream);
int mnas_tcr_ok(int Ho, int Wo, int Co, int Ci);
int mnas_tcr_parts(int N, int Ho, int Wo, int Co, int Ci);
int mnas_tcr_dgrad(const MnasTconvDgrad* c, void* stream);
int mnas_pwx_parts(int M, int Ci, int Co);                       // csrc/mnas_pwx.hip: widening 1x1 forward, weight-stationary
int mnas_pwx_forward(const MnasConvGemm* c, void* stream);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware remap of a 1-D block id: consecutive LOGICAL ids share an XCD (and its L2).
// Hardware places block b on XCD b % 8 (observed, speed only).  Bijective for any n.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
    const unsigned q = n >> 3, r = n & 7u;
    const unsigned xcd = b & 7u, slot = b >> 3;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
