// Squeeze-and-excitation for the MBConv block (BASELINE config 4: "5x5-depthwise + SE-block variant").  BUILD-DEFINED: the
// reference has no SE block anywhere, so the definition is this repo's (MnasNet-A1 form), restated in
// oracle/mnasnet_oracle.py::se_apply -- "parity unpinned by the reference":
//     a2 = relu(bn2(y2))                       the activated depthwise output (N,H,W,E)
//     z  = mean_hw a2                          squeeze          -> mnas_pool_act (existing)
//     h  = relu(fc1 z + b1),  u = fc2 h + b2   excite (R << E)  -> mnas_head_linear_fwd (existing GEMM)
//     a2s = a2 * sigmoid(u)[n][e]              scale            -> k_se_scale: the project conv reads a2s as a plain activation
// Backward (gs = dL/da2s from the project conv's input gradient):
//     du[n][e] = (sum_hw gs * a2) * s (1 - s)                   -> k_se_bwd_reduce
//     MLP backward                                              -> mnas_head_linear_bwd_w / _bwd_x (existing)
//     g_a2 = gs * s + dz[n][e] / HW                             -> k_se_bwd_apply  (dz = dL/dz from the MLP backward)
// All three kernels are elementwise / per-image reductions over the E-wide tensor: HBM-bound (read y2 [+ gs], write one tensor).
#include "mnas_common.h"

__device__ __forceinline__ float se_sigmoid(float u) { return 1.f / (1.f + __expf(-u)); }

// Common geometry: a workgroup = (image n, pixel range); a thread keeps ONE 16-byte channel group for its whole range, so the
// BatchNorm coefficients and sigmoid(u[n][c]) of its 8 channels are registers computed once (the naive form re-evaluated 8
// exponentials per pixel and channel group).  R = 256 / G pixel rows are in flight per workgroup step (G = C / 8).
struct SeGeom { int G, R, HW, C, chunk; };      // chunk = pixels per workgroup (multiple of R)

__global__ __launch_bounds__(256) void k_se_scale(MnasActIn a, const float* __restrict__ u, SeGeom g, uint4* __restrict__ out) {
    const int n = blockIdx.x, cg = threadIdx.x % g.G, pr = threadIdx.x / g.G;
    if (pr >= g.R) return;
    float s[8], t[8], sg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s[j] = a.scale ? a.scale[cg * 8 + j] : 1.f;
        t[j] = a.scale ? a.shift[cg * 8 + j] : 0.f;
        sg[j] = se_sigmoid(u[(size_t)n * g.C + cg * 8 + j]);
    }
    const int p0 = blockIdx.y * g.chunk, p1 = min(g.HW, p0 + g.chunk);
    const uint4* src = (const uint4*)a.data + (size_t)n * g.HW * g.G + cg;
    uint4* dst = out + (size_t)n * g.HW * g.G + cg;
    for (int p = p0 + pr; p < p1; p += g.R) {
        float f[8];
        unpack8(src[(size_t)p * g.G], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = a.scale ? fmaxf(fmaf(f[j], s[j], t[j]), 0.f) : f[j];
            f[j] = v * sg[j];
        }
        dst[(size_t)p * g.G] = pack8(f);
    }
}

// du partial sums: every workgroup reduces its pixel range in registers, its R rows through LDS in fixed order, and writes
// dupart[split][n][c]; k_se_bwd_finish adds the splits in order and applies s (1 - s).  Deterministic.
__global__ __launch_bounds__(256) void k_se_bwd_reduce(const uint4* __restrict__ gs, MnasActIn a, SeGeom g, float* __restrict__ dupart,
                                                       int N) {
    extern __shared__ float red[];                                  // [R][C]
    const int n = blockIdx.x, cg = threadIdx.x % g.G, pr = threadIdx.x / g.G;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pr < g.R) {
        float s[8], t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = a.scale ? a.scale[cg * 8 + j] : 1.f; t[j] = a.scale ? a.shift[cg * 8 + j] : 0.f; }
        const int p0 = blockIdx.y * g.chunk, p1 = min(g.HW, p0 + g.chunk);
        const size_t base = (size_t)n * g.HW * g.G + cg;
        for (int p = p0 + pr; p < p1; p += g.R) {
            float gq[8], y[8];
            unpack8(gs[base + (size_t)p * g.G], gq);
            unpack8(((const uint4*)a.data)[base + (size_t)p * g.G], y);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = a.scale ? fmaxf(fmaf(y[j], s[j], t[j]), 0.f) : y[j];
                acc[j] = fmaf(gq[j], v, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[pr * g.C + cg * 8 + j] = acc[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < g.C; c += 256) {
        float v = 0.f;
        for (int r = 0; r < g.R; ++r) v += red[r * g.C + c];
        dupart[((size_t)blockIdx.y * N + n) * g.C + c] = v;
    }
}
__global__ __launch_bounds__(256) void k_se_bwd_finish(const float* __restrict__ dupart, const float* __restrict__ u, int splits, int NC,
                                                       float* __restrict__ du) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NC) return;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += dupart[(size_t)s * NC + i];
    const float sg = se_sigmoid(u[i]);
    du[i] = v * sg * (1.f - sg);
}

// RED: also the BatchNorm-backward reduce of the depthwise conv whose activated output `out` is the gradient of (red_y = its raw
// output, red_bn = its bnbuf): partial (sum dz, sum dz*xhat) per workgroup, dz = out * [s*y+t > 0] with out AS STORED (bf16) --
// exactly what mnas_bn_bwd_reduce would compute from (out, red_y, red_bn) in a separate pass over both tensors.
template <bool RED>
__global__ __launch_bounds__(256) void k_se_bwd_apply(const uint4* __restrict__ gs, const float* __restrict__ u, const float* __restrict__ dz,
                                                      SeGeom g, uint4* __restrict__ out, const uint4* __restrict__ red_y,
                                                      const float* __restrict__ red_bn, float* __restrict__ red_partial) {
    extern __shared__ float red[];                                  // RED: [R][2][C]
    const int n = blockIdx.x, cg = threadIdx.x % g.G, pr = threadIdx.x / g.G;
    float r1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, r2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pr < g.R) {
        float sg[8], zz[8], bs[8], bt[8], bi[8], bm[8];
        const float inv = 1.f / (float)g.HW;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            sg[j] = se_sigmoid(u[(size_t)n * g.C + c]);
            zz[j] = dz[(size_t)n * g.C + c] * inv;
            if (RED) {
                bs[j] = red_bn[c]; bt[j] = red_bn[g.C + c]; bi[j] = red_bn[6 * g.C + c];
                bm[j] = -red_bn[5 * g.C + c] * red_bn[6 * g.C + c];
            }
        }
        const int p0 = blockIdx.y * g.chunk, p1 = min(g.HW, p0 + g.chunk);
        const size_t base = (size_t)n * g.HW * g.G + cg;
        for (int p = p0 + pr; p < p1; p += g.R) {
            float gq[8];
            unpack8(gs[base + (size_t)p * g.G], gq);
#pragma unroll
            for (int j = 0; j < 8; ++j) gq[j] = fmaf(gq[j], sg[j], zz[j]);
            const uint4 o = pack8(gq);
            out[base + (size_t)p * g.G] = o;
            if (RED) {
                float y[8], q[8];
                unpack8(red_y[base + (size_t)p * g.G], y);
                unpack8(o, q);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = (fmaf(y[j], bs[j], bt[j]) > 0.f) ? q[j] : 0.f;
                    r1[j] += d;
                    r2[j] = fmaf(d, fmaf(y[j], bi[j], bm[j]), r2[j]);
                }
            }
        }
    }
    if (RED) {
        if (pr < g.R) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[(pr * 2 + 0) * g.C + cg * 8 + j] = r1[j]; red[(pr * 2 + 1) * g.C + cg * 8 + j] = r2[j]; }
        }
        __syncthreads();
        const int col = blockIdx.x * gridDim.y + blockIdx.y, ncols = gridDim.x * gridDim.y;
        for (int i = threadIdx.x; i < 2 * g.C; i += 256) {
            const int r = i / g.C, c = i - r * g.C;
            float v = 0.f;
            for (int k = 0; k < g.R; ++k) v += red[(k * 2 + r) * g.C + c];           // fixed order: deterministic
            red_partial[((size_t)r * g.C + c) * ncols + col] = v;
        }
    }
}

// ---- excitation applied ON LOAD (round 4): the project conv reads relu(bn2(y2)) * sigmoid(u)[n][c] through MnasConvGemm.gate
// instead of a materialised a2s (k_se_scale's read + write and the project conv's second read of an E-wide tensor go away).
__global__ __launch_bounds__(256) void k_se_gate(const float* __restrict__ u, int NC, float* __restrict__ gate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NC) gate[i] = se_sigmoid(u[i]);
}
extern "C" int mnas_se_gate(const float* u, int N, int C, float* gate, void* stream) {
    if (!u || !gate || N < 1 || C < 1) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_gate, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, u, N * C, gate);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// Backward of the gated project conv WITHOUT a pass over the E-wide tensors for du.  mnas_pw_bwd ran in segment mode on the
// UNGATED activation a2 = relu(bn2(y2)), `kseg` workgroups per image: wpartial[n*kseg + j][o][c] = sum over that pixel range of
// dy3[pix][o] * a2[pix][c].  With gs = dy3 . W3 (the project conv's input gradient, = dL/d(a2*s)):
//     du_pre[n][c] = sum_pix gs[pix][c] * a2[pix][c] = sum_o W3[o][c] * (sum_j wpartial[n*kseg+j][o][c])
//     dW3[o][c]    = sum_n s[n][c] * (sum_j wpartial[n*kseg+j][o][c])                      (a2s = a2 * s)
// k_se_proj_du: one workgroup per image; folds the image's kseg slabs into slab n*kseg scaled by s (what k_se_proj_dw then adds
// over the images) and writes du = du_pre * s (1 - s).  Both sums run in a fixed order: deterministic.
__global__ __launch_bounds__(256) void k_se_proj_du(float* __restrict__ wpartial, int kseg, int Co, int Ci, const float* __restrict__ u,
                                                    const float* __restrict__ W, float* __restrict__ du) {
    // thread = (input channel c, output-channel group og): all 256 threads work whatever Ci is (the squeeze-excite project convs
    // have Ci = 48 / 72 / 240: one thread per channel left 19-94 % of the workgroup idle on a launch that sits on the critical
    // path); the og partial sums of a channel meet in LDS and are added in group order: deterministic.  W is the fp32 master
    // weight (gs was formed with its bf16 rounding: the difference is inside the gradient tolerances, include/mnas.h says so).
    __shared__ float part[256];
    const int n = blockIdx.x;
    const size_t slab = (size_t)Co * Ci;
    float* base = wpartial + (size_t)n * kseg * slab;
    if (Ci <= 256) {
        const int G = 256 / Ci;
        const int c = threadIdx.x % Ci, og = threadIdx.x / Ci;
        const float sg = se_sigmoid(u[(size_t)n * Ci + c]);
        float acc = 0.f;
        if (og < G) {
            for (int o = og; o < Co; o += G) {
                float p = base[(size_t)o * Ci + c];
                for (int j = 1; j < kseg; ++j) p += base[j * slab + (size_t)o * Ci + c];
                acc = fmaf(W[(size_t)o * Ci + c], p, acc);
                base[(size_t)o * Ci + c] = p * sg;
            }
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        if (og == 0) {
            float a = part[c];
            for (int g = 1; g < G; ++g) a += part[g * Ci + c];
            du[(size_t)n * Ci + c] = a * sg * (1.f - sg);
        }
        return;
    }
    for (int c = threadIdx.x; c < Ci; c += 256) {
        const float sg = se_sigmoid(u[(size_t)n * Ci + c]);
        float acc = 0.f;
        for (int o = 0; o < Co; ++o) {
            float p = base[(size_t)o * Ci + c];
            for (int j = 1; j < kseg; ++j) p += base[j * slab + (size_t)o * Ci + c];
            acc = fmaf(W[(size_t)o * Ci + c], p, acc);
            base[(size_t)o * Ci + c] = p * sg;
        }
        du[(size_t)n * Ci + c] = acc * sg * (1.f - sg);
    }
}
// dW[i] (+)= sum_n wpartial[n*kseg][i]: thread column tx = element, 8 row groups ty combined through LDS in fixed order
__global__ __launch_bounds__(256) void k_se_proj_dw(const float* __restrict__ wpartial, int N, size_t nstride, int total,
                                                    float* __restrict__ dW, int accumulate) {
    __shared__ float red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (i < total) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int n = ty; n < N; n += 32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += (n + 8 * j < N) ? wpartial[(size_t)(n + 8 * j) * nstride + i] : 0.f;
        }
        s = (a[0] + a[1]) + (a[2] + a[3]);
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < total) {
#pragma unroll
        for (int j = 1; j < 8; ++j) s += red[j][tx];
        dW[i] = (accumulate ? dW[i] : 0.f) + s;
    }
}
extern "C" int mnas_se_proj_finalize(float* wpartial, int N, int kseg, int Co, int Ci, const float* u, const float* W, float* dW,
                                     int accumulate, float* du, void* stream) {
    if (!wpartial || !u || !W || !dW || !du || N < 1 || kseg < 1 || Co < 1 || Ci < 1) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_proj_du, dim3(N), dim3(256), 0, (hipStream_t)stream, wpartial, kseg, Co, Ci, u, W, du);
    const int total = Co * Ci;
    hipLaunchKernelGGL(k_se_proj_dw, dim3((total + 31) / 32), dim3(256), 0, (hipStream_t)stream, wpartial,
                       N, (size_t)kseg * total, total, dW, accumulate);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// ---- the excitation MLP in one launch per direction (round 5, ABI 7) -----------------------------------------------------------
// z [N][E] -> hb = relu(W1 z + b1) [N][R] -> u = W2 hb + b2 [N][E] (-> gate = sigmoid(u)), R = 8..48 hidden units.  As
// mnas_head_linear_* calls these were 2 (+ the gate) forward and 4 backward launches per block of 9-13 us each on problems of a
// few hundred kFLOP -- 102 launches, 1.17 ms of the squeeze-excite variant's step.  These kernels are bound by LOAD LATENCY, not by
// bytes or flops: every loop over global memory is written as explicit batches of 16-40 independent loads (a plain loop with a
// run-time trip count waits for each load in turn: the first version of k_se_fc_fwd took 100 us at E = 1152).  Parameters are
// views of the trainer's flat buffer, 4-byte aligned only: scalar loads throughout.  Every sum runs in a fixed order.

#define SE_NB 2              // images per workgroup of the per-image kernels (they share every weight load)
#define SE_TR 256            // W2 rows per LDS tile
#define SE_FC_MAXR 48
__device__ __forceinline__ float se_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// W2 [E][R] is read one ROW per channel: SE_TR rows at a time go through LDS (coalesced flat copy, odd row pitch -> the row reads
// are conflict-free) -- a thread walking its own row in global memory touches 64 different lines per load instruction of its wave.
__device__ __forceinline__ void se_stage_rows(float* __restrict__ t_s, const float* __restrict__ W2, int e0, int E, int R, int ld, float rcp_r) {
    const int rows = min(SE_TR, E - e0), total = rows * R;
    const float* src = W2 + (size_t)e0 * R;
    constexpr int U = SE_TR * SE_FC_MAXR / 256 / 2;                  // two batches of U loads cover the largest tile
    for (int base = 0; base < total; base += 256 * U) {
        float v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int i = base + threadIdx.x + 256 * j;
            v[j] = i < total ? src[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int i = base + threadIdx.x + 256 * j;
            if (i < total) {
                int row = (int)(((float)i + 0.5f) * rcp_r);          // i / R for i < 2^20 (R <= 48)
                t_s[row * ld + (i - row * R)] = v[j];
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_se_fc_fwd(const float* __restrict__ z, const float* __restrict__ W1, const float* __restrict__ b1,
                                                   const float* __restrict__ W2, const float* __restrict__ b2, int N, int E, int R,
                                                   float* __restrict__ hb, float* __restrict__ u, float* __restrict__ gate) {
    extern __shared__ float se_s[];                                 // [NB][E] z, [NB][R] hidden, [SE_TR][R|1] W2 rows
    float* z_s = se_s;
    float* h_s = se_s + SE_NB * E;
    float* t_s = h_s + SE_NB * R;
    const int ld = R | 1;
    const float rcp_r = 1.f / (float)R;
    const int n0 = blockIdx.x * SE_NB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        float v[SE_NB * 5];                                          // E <= 1280: one batch
#pragma unroll
        for (int j = 0; j < SE_NB * 5; ++j) {
            const int i = tid + 256 * j;
            v[j] = (i < SE_NB * E && n0 + (i >= E ? 1 : 0) < N) ? z[(size_t)n0 * E + i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < SE_NB * 5; ++j) {
            const int i = tid + 256 * j;
            if (i < SE_NB * E) z_s[i] = v[j];
        }
    }
    __syncthreads();
    // hidden rows: a wave takes rows r, r+4, r+8, r+12 at once (4 x 10 loads in flight per lane), lanes stride the channels
    for (int r = wave; r < R; r += 16) {
        float acc[4][SE_NB];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < SE_NB; ++b) acc[q][b] = 0.f;
        for (int e0 = 0; e0 < E; e0 += 640) {
            float w[4][10];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const int e = e0 + lane + 64 * j;
                    w[q][j] = (e < E && r + 4 * q < R) ? W1[(size_t)(r + 4 * q) * E + e] : 0.f;
                }
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int e = e0 + lane + 64 * j;
                if (e < E) {
#pragma unroll
                    for (int b = 0; b < SE_NB; ++b) {
                        const float zz = z_s[b * E + e];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q][b] = fmaf(w[q][j], zz, acc[q][b]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < SE_NB; ++b) {
                const float v = se_wave_sum(acc[q][b]);
                const int rr = r + 4 * q;
                if (lane == 0 && rr < R) {
                    const float h = fmaxf(v + b1[rr], 0.f);
                    h_s[b * R + rr] = h;
                    if (n0 + b < N) hb[(size_t)(n0 + b) * R + rr] = h;
                }
            }
    }
    for (int e0 = 0; e0 < E; e0 += SE_TR) {
        __syncthreads();                                             // hidden rows written / previous tile consumed
        se_stage_rows(t_s, W2, e0, E, R, ld, rcp_r);
        __syncthreads();
        const int e = e0 + tid;
        if (e < E) {
            float acc[SE_NB];
            const float bb = b2[e];
#pragma unroll
            for (int b = 0; b < SE_NB; ++b) acc[b] = bb;
            const float* w = t_s + tid * ld;
#pragma unroll 8                                                     // (LDS latency: eight rows' reads in flight)
            for (int r = 0; r < R; ++r) {
                const float wr = w[r];
#pragma unroll
                for (int b = 0; b < SE_NB; ++b) acc[b] = fmaf(wr, h_s[b * R + r], acc[b]);
            }
#pragma unroll
            for (int b = 0; b < SE_NB; ++b) {
                if (n0 + b < N) {
                    u[(size_t)(n0 + b) * E + e] = acc[b];
                    if (gate) gate[(size_t)(n0 + b) * E + e] = se_sigmoid(acc[b]);
                }
            }
        }
    }
}
// dh = (du W2) * [hb > 0],  dz = dh W1   (SE_NB images per workgroup; thread = channel e, W2 rows through LDS as above)
__global__ __launch_bounds__(256) void k_se_fc_bwd_x(const float* __restrict__ du, const float* __restrict__ hb, const float* __restrict__ W1,
                                                     const float* __restrict__ W2, int N, int E, int R, float* __restrict__ dh,
                                                     float* __restrict__ dz) {
    extern __shared__ float se_s[];
    float* part = se_s;                                             // [S slices][NB][R] partial dh
    float* h_s = part + 256;                                        // [NB][R] dh
    float* d_s = h_s + SE_NB * R;                                   // [NB][SE_TR] du of the tile's channels
    float* t_s = d_s + SE_NB * SE_TR;                               // [SE_TR][R|1] W2 rows
    const int ld = R | 1;
    const float rcp_r = 1.f / (float)R;
    const int n0 = blockIdx.x * SE_NB, tid = threadIdx.x;
    // dh: thread = (hidden unit r, image b, row slice sl): the S = 256 / (NB R) slices walk the tile rows sl, sl+S, ... (du
    // broadcast, W2 row elements of consecutive r adjacent: conflict-free); slices are added in order at the end
    const int S = 256 / (SE_NB * R);
    const int pr = tid % (SE_NB * R), sl = tid / (SE_NB * R);
    const int rb = pr / R, rr = pr - rb * R;
    float dacc = 0.f;
    for (int e0 = 0; e0 < E; e0 += SE_TR) {
        __syncthreads();
        se_stage_rows(t_s, W2, e0, E, R, ld, rcp_r);
        {
            float v[SE_NB];
#pragma unroll
            for (int b = 0; b < SE_NB; ++b) v[b] = (e0 + tid < E && n0 + b < N) ? du[(size_t)(n0 + b) * E + e0 + tid] : 0.f;
#pragma unroll
            for (int b = 0; b < SE_NB; ++b) d_s[b * SE_TR + tid] = v[b];
        }
        __syncthreads();
        if (sl < S) {
            const int rows = min(SE_TR, E - e0);
#pragma unroll 8
            for (int row = sl; row < rows; row += S) dacc = fmaf(d_s[rb * SE_TR + row], t_s[row * ld + rr], dacc);
        }
    }
    if (sl < S) part[sl * SE_NB * R + pr] = dacc;
    __syncthreads();
    for (int i = tid; i < SE_NB * R; i += 256) {
        const int b = i / R, r = i - b * R;
        float v = part[i];
        for (int q = 1; q < S; ++q) v += part[q * SE_NB * R + i];
        float g = 0.f;
        if (n0 + b < N) {
            g = hb[(size_t)(n0 + b) * R + r] > 0.f ? v : 0.f;
            dh[(size_t)(n0 + b) * R + r] = g;
        }
        h_s[i] = g;
    }
    __syncthreads();
    // dz: thread = channels tid + 256 k (k < 5: E <= 1280); 8 hidden rows x 5 channels of W1 in flight per batch
    float acc[5][SE_NB];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int b = 0; b < SE_NB; ++b) acc[k][b] = 0.f;
    for (int r0 = 0; r0 < R; r0 += 8) {
        float w[8][5];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int e = tid + 256 * k;
                w[q][k] = (e < E && r0 + q < R) ? W1[(size_t)(r0 + q) * E + e] : 0.f;
            }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (r0 + q < R) {
#pragma unroll
                for (int b = 0; b < SE_NB; ++b) {
                    const float hh = h_s[b * R + r0 + q];
#pragma unroll
                    for (int k = 0; k < 5; ++k) acc[k][b] = fmaf(hh, w[q][k], acc[k][b]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int e = tid + 256 * k;
        if (e < E) {
#pragma unroll
            for (int b = 0; b < SE_NB; ++b)
                if (n0 + b < N) dz[(size_t)(n0 + b) * E + e] = acc[k][b];
        }
    }
}
// dW2[e][r] (+)= sum_n du[n][e] hb[n][r], db2[e] (+)= sum_n du[n][e], dW1[r][e] (+)= sum_n dh[n][r] z[n][e], db1[r] (+)= sum_n dh[n][r]
// workgroup (x, y) = 16 channels x 4 hidden units.  The images are dealt over 16 "phases" (4 lane groups x 4 waves): lane = channel
// + 16 * group, so a lane walks N / 16 images -- ONE batch of loads for N = 256 (a thread walking all N images alone spent 14 us
// waiting for 16 batches in turn).  hb / dh of the four units come from LDS; the phases' sums are added in a fixed order (lane
// groups by shuffle, waves through LDS).
#define SE_WCH 512           // images per LDS chunk of (hb, dh)
__global__ __launch_bounds__(256) void k_se_fc_bwd_w(const float* __restrict__ du, const float* __restrict__ z, const float* __restrict__ hb,
                                                     const float* __restrict__ dh, int N, int E, int R, float* __restrict__ dW1,
                                                     float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2, int accumulate) {
    __shared__ float hd_s[SE_WCH][8];                               // [image][hb of 4 units, dh of 4 units]
    __shared__ float red_s[3][16][13];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ex = lane & 15, ph = wave * 4 + (lane >> 4);          // image phase 0..15
    const int r0 = blockIdx.y * 4, e = blockIdx.x * 16 + ex;
    const int ec = e < E ? e : E - 1;
    float a2[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2 = 0.f;
    for (int c0 = 0; c0 < N; c0 += SE_WCH) {
        const int cn = min(SE_WCH, N - c0);
        __syncthreads();
        for (int base = 0; base < cn * 8; base += 256 * 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + tid + 256 * j, nn = i >> 3, q = i & 7, r = r0 + (q & 3);
                v[j] = (i < cn * 8 && r < R) ? (q < 4 ? hb : dh)[(size_t)(c0 + nn) * R + r] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = base + tid + 256 * j;
                if (i < cn * 8) hd_s[i >> 3][i & 7] = v[j];
            }
        }
        __syncthreads();
        for (int nb = ph; nb < cn; nb += 256) {                      // 16 images of this phase per batch
            float d[16], zz[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int nn = nb + 16 * j;
                d[j] = nn < cn ? du[(size_t)(c0 + nn) * E + ec] : 0.f;
                zz[j] = nn < cn ? z[(size_t)(c0 + nn) * E + ec] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int nn = nb + 16 * j;
                if (nn < cn) {
                    s2 += d[j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float h = hd_s[nn][q], g = hd_s[nn][4 + q];
                        a2[q] = fmaf(d[j], h, a2[q]);
                        a1[q] = fmaf(g, zz[j], a1[q]);
                        s1[q] += g;
                    }
                }
            }
        }
    }
    // lane groups (phases 4w .. 4w+3) of a wave: butterfly over lanes ^16, ^32 -- the same tree for every run
    float vals[13];
#pragma unroll
    for (int q = 0; q < 4; ++q) { vals[q] = a2[q]; vals[4 + q] = a1[q]; vals[8 + q] = s1[q]; }
    vals[12] = s2;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        vals[k] += __shfl_xor(vals[k], 16, 64);
        vals[k] += __shfl_xor(vals[k], 32, 64);
    }
    if (wave > 0 && lane < 16) {
#pragma unroll
        for (int k = 0; k < 13; ++k) red_s[wave - 1][lane][k] = vals[k];
    }
    __syncthreads();
    if (wave != 0 || lane >= 16) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int k = 0; k < 13; ++k) vals[k] += red_s[w][lane][k];
    if (e < E) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q;
            if (r < R) {
                float* q2 = dW2 + (size_t)e * R + r;
                float* q1 = dW1 + (size_t)r * E + e;
                *q2 = (accumulate ? *q2 : 0.f) + vals[q];
                *q1 = (accumulate ? *q1 : 0.f) + vals[4 + q];
            }
        }
        if (blockIdx.y == 0) db2[e] = (accumulate ? db2[e] : 0.f) + vals[12];
    }
    if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (r0 + q < R) db1[r0 + q] = (accumulate ? db1[r0 + q] : 0.f) + vals[8 + q];
    }
}
// shapes the fused MLP kernels take: <= 48 hidden units, <= 1280 channels (k_se_fc_bwd_x's register tile), 64 KB of LDS
extern "C" int mnas_se_fc_supported(int E, int R) {
    return E >= 1 && E <= 1280 && R >= 1 && R <= SE_FC_MAXR &&
           (size_t)(SE_NB * E + 256 + SE_NB * R + SE_NB * SE_TR + SE_TR * (R | 1)) * sizeof(float) <= 64 * 1024;
}
extern "C" int mnas_se_fc_fwd(const float* z, const float* W1, const float* b1, const float* W2, const float* b2, int N, int E, int R,
                              float* hb, float* u, float* gate, void* stream) {
    if (!z || !W1 || !b1 || !W2 || !b2 || !hb || !u || N < 1 || !mnas_se_fc_supported(E, R)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_fc_fwd, dim3((N + SE_NB - 1) / SE_NB), dim3(256), (size_t)(SE_NB * (E + R) + SE_TR * (R | 1)) * sizeof(float),
                       (hipStream_t)stream, z, W1, b1, W2, b2, N, E, R, hb, u, gate);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
extern "C" int mnas_se_fc_bwd(const float* du, const float* z, const float* hb, const float* W1, const float* W2, int N, int E, int R,
                              float* dh, float* dz, float* dW1, float* db1, float* dW2, float* db2, int accumulate, void* stream) {
    if (!du || !z || !hb || !W1 || !W2 || !dh || !dz || !dW1 || !db1 || !dW2 || !db2) return MNAS_EINVAL;
    if (N < 1 || !mnas_se_fc_supported(E, R)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_fc_bwd_x, dim3((N + SE_NB - 1) / SE_NB), dim3(256), (size_t)(256 + SE_NB * R + SE_NB * SE_TR + SE_TR * (R | 1)) * sizeof(float),
                       (hipStream_t)stream, du, hb, W1, W2, N, E, R, dh, dz);
    hipLaunchKernelGGL(k_se_fc_bwd_w, dim3((E + 15) / 16, (R + 3) / 4), dim3(256), 0, (hipStream_t)stream, du, z, hb, dh, N, E, R, dW1, db1,
                       dW2, db2, accumulate ? 1 : 0);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}

// pixel splits per image: enough workgroups to fill the chip (>= ~2048), whole multiples of R pixels each
static bool se_geom(int N, int HW, int C, SeGeom* g, int* splits) {
    if (N < 1 || HW < 1 || C < 8 || (C & 7) || C > 2048) return false;
    g->G = C >> 3; g->R = 256 / g->G; g->HW = HW; g->C = C;
    if (g->R < 1) return false;
    int sp = 2048 / N;                                               // N * splits <= 2048: the fused reduce's table has one column per workgroup
    const int maxsp = (HW + g->R * 4 - 1) / (g->R * 4);              // at least 4 pixels per thread
    if (sp > maxsp) sp = maxsp;
    if (sp < 1) sp = 1;
    int chunk = (HW + sp - 1) / sp;
    chunk = (chunk + g->R - 1) / g->R * g->R;
    g->chunk = chunk;
    *splits = (HW + chunk - 1) / chunk;
    return true;
}
// bytes of scratch mnas_se_bwd_reduce needs (partial sums float[splits][N][C])
extern "C" int64_t mnas_se_scratch_bytes(int N, int HW, int C) {
    SeGeom g; int sp;
    if (!se_geom(N, HW, C, &g, &sp)) return -1;
    return (int64_t)sp * N * C * sizeof(float);
}
extern "C" int mnas_se_scale(const MnasActIn* a, const float* u, int N, int HW, int C, void* out_bf16, void* stream) {
    SeGeom g; int sp;
    if (!a || !a->data || !u || !out_bf16 || !se_geom(N, HW, C, &g, &sp)) return MNAS_EINVAL;
    if ((a->scale == nullptr) != (a->shift == nullptr)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_scale, dim3(N, sp), dim3(256), 0, (hipStream_t)stream, *a, u, g, (uint4*)out_bf16);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
extern "C" int mnas_se_bwd_reduce(const void* gs, const MnasActIn* a, const float* u, int N, int HW, int C, float* du, float* scratch,
                                  void* stream) {
    SeGeom g; int sp;
    if (!gs || !a || !a->data || !u || !du || !scratch || !se_geom(N, HW, C, &g, &sp)) return MNAS_EINVAL;
    hipLaunchKernelGGL(k_se_bwd_reduce, dim3(N, sp), dim3(256), (size_t)g.R * C * sizeof(float), (hipStream_t)stream, (const uint4*)gs, *a, g,
                       scratch, N);
    hipLaunchKernelGGL(k_se_bwd_finish, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, u, sp, N * C, du);
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
extern "C" int mnas_se_bwd_apply(const void* gs, const float* u, const float* dz, int N, int HW, int C, void* out_bf16,
                                 const void* red_y, const float* red_bn, float* red_partial, void* stream) {
    SeGeom g; int sp;
    if (!gs || !u || !dz || !out_bf16 || !se_geom(N, HW, C, &g, &sp)) return MNAS_EINVAL;
    if (red_partial) {
        if (!red_y || !red_bn) return MNAS_EINVAL;
        hipLaunchKernelGGL(k_se_bwd_apply<true>, dim3(N, sp), dim3(256), (size_t)g.R * 2 * C * sizeof(float), (hipStream_t)stream,
                           (const uint4*)gs, u, dz, g, (uint4*)out_bf16, (const uint4*)red_y, red_bn, red_partial);
    } else {
        hipLaunchKernelGGL(k_se_bwd_apply<false>, dim3(N, sp), dim3(256), 0, (hipStream_t)stream, (const uint4*)gs, u, dz, g,
                           (uint4*)out_bf16, (const uint4*)nullptr, (const float*)nullptr, (float*)nullptr);
    }
    MNAS_CHECK_LAUNCH();
    return MNAS_OK;
}
// columns of the fused reduce table mnas_se_bwd_apply writes (= N * pixel splits), or -1
extern "C" int mnas_se_bwd_apply_cols(int N, int HW, int C) {
    SeGeom g; int sp;
    if (!se_geom(N, HW, C, &g, &sp)) return -1;
    return N * sp;
}
