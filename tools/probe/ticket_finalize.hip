// Probe (round 6): is a BatchNorm finalize done by the LAST workgroup of its producer cheaper than the separate
// finalize launch?  Measured INSIDE a realistic producer -- a persistent streaming kernel (NWG workgroups copy `bytes` with four
// 16-byte loads in flight per lane, ~100 us, workgroups retire staggered) that ends by writing its per-channel partial sums
// float[2][C][NWG], exactly what the conv kernels of the step do -- followed by a dependent consumer launch that reads the
// per-channel (scale, shift).  Chains of (producer [, finalize], consumer) run back to back in one stream:
//   mode 0  producer + consumer only (no finalize at all: the floor)
//   mode 1  producer (plain partial stores) + k_finalize (C workgroups, fixed-order fp64 sums) + consumer   -- what the step does
//   mode 2  ticketed: partials as agent-scope RELAXED atomic stores (sc1: no L2 write-back fence), s_waitcnt vmcnt(0), barrier, ONE
//           relaxed agent-scope fetch_add per workgroup; the workgroup that draws NWG-1 reduces the table (agent-scope loads, the
//           same fixed order -> bit-identical to mode 1) and writes (scale, shift); no __threadfence anywhere
//   mode 3  as 2 with plain stores + __threadfence() before the ticket and after it in the last workgroup (the classic form)
// hipcc --offload-arch=gfx950 -O3 tools/probe/ticket_finalize.hip -o tools/probe/ticket_finalize.bin && tools/probe/ticket_finalize.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum_fixed(double v) {          // fixed butterfly: deterministic
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// the table reduce: sums s = 0 .. 2C-1, each over NWG partials; wave w takes s = w, w+4, ...; lane l adds partials l, l+64, ...
template <bool AGENT>
__device__ __forceinline__ void reduce_table(const float* part, int C, int nwg, double count, float* coef) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < C; c += 4) {
        double s1 = 0.0, s2 = 0.0;
        for (int p = lane; p < nwg; p += 64) {
            float a, b;
            if (AGENT) {
                a = __hip_atomic_load(part + (size_t)c * nwg + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = __hip_atomic_load(part + (size_t)(C + c) * nwg + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                a = part[(size_t)c * nwg + p];
                b = part[(size_t)(C + c) * nwg + p];
            }
            s1 += a; s2 += b;
        }
        s1 = wave_sum_fixed(s1); s2 = wave_sum_fixed(s2);
        if (lane == 0) {
            const double mean = s1 / count;
            double var = s2 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const double inv = 1.0 / sqrt(var + 1e-5);
            coef[c] = (float)inv;
            coef[C + c] = (float)(-mean * inv);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_producer(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n, float* part, int C,
                                                 unsigned* ticket, double count, float* coef) {
    __shared__ int last;
    const int tid = threadIdx.x;
    const size_t st = (size_t)gridDim.x * 1024;
    size_t i = (size_t)blockIdx.x * 1024 + tid;
    float acc1 = 0.f, acc2 = 0.f;
    for (; i + 768 < n; i += st) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256);
        const u32x4 c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
        acc1 += __uint_as_float(a.x & 0x3fffffffu) + __uint_as_float(c.y & 0x3fffffffu);
        acc2 += __uint_as_float(b.z & 0x3fffffffu) + __uint_as_float(d.w & 0x3fffffffu);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + 256);
        __builtin_nontemporal_store(c, dst + i + 512);
        __builtin_nontemporal_store(d, dst + i + 768);
    }
    if (MODE == 0) { if (acc1 + acc2 == -1.f) part[0] = acc1; return; }
    // per-channel partials of this workgroup (channel = thread, strided): [2][C][NWG]
    for (int c = tid; c < C; c += 256) {
        const float v1 = acc1 + (float)c, v2 = acc2 + 2.f * (float)c;
        if (MODE == 2) {
            __hip_atomic_store(part + (size_t)c * gridDim.x + blockIdx.x, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part + (size_t)(C + c) * gridDim.x + blockIdx.x, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            part[(size_t)c * gridDim.x + blockIdx.x] = v1;
            part[(size_t)(C + c) * gridDim.x + blockIdx.x] = v2;
        }
    }
    if (MODE == 1) return;
    if (MODE == 3) __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    if (MODE == 3) __threadfence();
    if (MODE == 2) reduce_table<true>(part, C, gridDim.x, count, coef); else reduce_table<false>(part, C, gridDim.x, count, coef);
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_finalize(const float* part, int C, int nwg, double count, float* coef) {
    // one workgroup per 4 channels would mirror k_bn_fwd_finalize<64>; C workgroups x 256 threads mirrors <256>: take the latter
    const int c = blockIdx.x, tid = threadIdx.x;
    __shared__ double red[2][4];
    double s1 = 0.0, s2 = 0.0;
    for (int p = tid; p < nwg; p += 256) { s1 += part[(size_t)c * nwg + p]; s2 += part[(size_t)(C + c) * nwg + p]; }
    s1 = wave_sum_fixed(s1); s2 = wave_sum_fixed(s2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = 1.0 / sqrt(var + 1e-5);
        coef[c] = (float)inv;
        coef[C + c] = (float)(-mean * inv);
    }
}

// dependent consumer: every workgroup reads the coefficients first (as act-on-load does), then streams a little
__global__ __launch_bounds__(256) void k_consumer(const float* coef, int C, const u32x4* __restrict__ src, float* sink, size_t n) {
    float s = 0.f;
    for (int c = threadIdx.x; c < 2 * C; c += 256) s += coef[c];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) s += __uint_as_float(src[i].x & 0x3fffffffu);
    if (s == -1.f) sink[0] = s;
}

template <int MODE>
static float chain(int reps, int nwg, int C, const u32x4* src, u32x4* dst, size_t n, float* part, unsigned* ticket, float* coef, float* sink,
                   hipEvent_t e0, hipEvent_t e1) {
    auto once = [&]() {
        hipLaunchKernelGGL(k_producer<MODE>, dim3(nwg), dim3(256), 0, 0, src, dst, n, part, C, ticket, 3.0e6, coef);
        if (MODE == 1) hipLaunchKernelGGL(k_finalize, dim3(C), dim3(256), 0, 0, part, C, nwg, 3.0e6, coef);
        hipLaunchKernelGGL(k_consumer, dim3(1024), dim3(256), 0, 0, coef, C, src, sink, n);
    };
    for (int i = 0; i < 3; ++i) once();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) once();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 384) << 20;      // MiB copied per producer launch
    const size_t n = bytes / 16;
    u32x4 *src, *dst; float *part, *coef, *sink; unsigned* ticket;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
    hipMemset(src, 0x11, bytes);
    hipMalloc(&part, 2 * 2048 * 2048 * sizeof(float)); hipMalloc(&coef, 2 * 2048 * sizeof(float) * 4); hipMalloc(&sink, 64); hipMalloc(&ticket, 4);
    hipMemset(ticket, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 30;
    const int NW[] = {1024, 512, 256}, CS[] = {48, 240, 576};
    printf("producer copies %zu MiB (reads + writes %zu MiB) per launch; us per (producer [, finalize], consumer) link, %d links per measurement\n",
           bytes >> 20, bytes >> 19, reps);
    for (int nwg : NW) for (int C : CS) {
        float t[4];
        for (int r = 0; r < 2; ++r) {                // second pass reported (first warms clocks)
            t[0] = chain<0>(reps, nwg, C, src, dst, n, part, ticket, coef, sink, e0, e1);
            t[1] = chain<1>(reps, nwg, C, src, dst, n, part, ticket, coef, sink, e0, e1);
            t[2] = chain<2>(reps, nwg, C, src, dst, n, part, ticket, coef, sink, e0, e1);
            t[3] = chain<3>(reps, nwg, C, src, dst, n, part, ticket, coef, sink, e0, e1);
        }
        // bit-equality of the coefficients of mode 1 and mode 2
        std::vector<float> c1(2 * C), c2(2 * C);
        chain<1>(1, nwg, C, src, dst, n, part, ticket, coef, sink, e0, e1); hipDeviceSynchronize();
        hipMemcpy(c1.data(), coef, 2 * C * 4, hipMemcpyDeviceToHost);
        hipMemset(coef, 0, 2 * C * 4);
        chain<2>(1, nwg, C, src, dst, n, part, ticket, coef, sink, e0, e1); hipDeviceSynchronize();
        hipMemcpy(c2.data(), coef, 2 * C * 4, hipMemcpyDeviceToHost);
        const bool same = memcmp(c1.data(), c2.data(), 2 * C * 4) == 0;
        printf("NWG=%4d C=%3d: no finalize %7.2f | separate launch %7.2f (+%5.2f) | ticket, sc1 stores %7.2f (+%5.2f) | ticket, __threadfence %7.2f (+%5.2f) | coefficients %s\n",
               nwg, C, t[0], t[1], t[1] - t[0], t[2], t[2] - t[0], t[3], t[3] - t[0], same ? "bit-equal" : "DIFFER");
    }
    return 0;
}
