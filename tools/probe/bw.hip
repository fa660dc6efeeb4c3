// HBM bandwidth ceilings on this pool: read-only, write-only, and mixed read:write ratios, plain vs nontemporal,
// 2 GiB working sets (beyond the 256 MiB Infinity Cache) and a 96 MiB producer->consumer pair (inside it).
// build: hipcc --offload-arch=gfx950 -O3 -o bw_bin bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float nf4;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

template <int U>
__global__ void k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x * U + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x * U;
    float s = 0.f;
    for (; i + (size_t)(U - 1) * blockDim.x < n; i += st) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 12345.678f) out[blockIdx.x] = s;
}
template <bool NT>
__global__ void k_fill(float4* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (; i < n; i += st) { if (NT) __builtin_nontemporal_store(nf4{v.x, v.y, v.z, v.w}, (nf4*)(b + i)); else b[i] = v; }
}
// R reads of n elements each from a (consecutive regions), W writes of n elements each to b
template <int R, int W, bool NT>
__global__ void k_mix(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        float4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = a[i + (size_t)r * n];
        float4 s = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) { s.x += v[r].x; s.y += v[r].y; s.z += v[r].z; s.w += v[r].w; }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            float4 o = s; o.x += (float)w;
            if (NT) __builtin_nontemporal_store(nf4{o.x, o.y, o.z, o.w}, (nf4*)(b + i + (size_t)w * n)); else b[i + (size_t)w * n] = o;
        }
    }
}
template <typename F>
static float timeit(F f, int iters = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}
int main() {
    const size_t bytes = (size_t)3 << 30;             // a: 3 GiB, b: 3 GiB
    float4 *a, *b; float* o; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 1 << 20));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    const size_t n2g = ((size_t)2 << 30) / 16;
    for (int grid : {1024, 2048, 4096, 16384}) {
        float ms = timeit([&] { k_read<4><<<grid, 256>>>(a, o, n2g); });
        printf("read  2GiB grid=%5d U=4: %.3f ms -> %.2f TB/s\n", grid, ms, 2.147483648 / ms);
        ms = timeit([&] { k_read<8><<<grid, 256>>>(a, o, n2g); });
        printf("read  2GiB grid=%5d U=8: %.3f ms -> %.2f TB/s\n", grid, ms, 2.147483648 / ms);
    }
    for (int grid : {1024, 2048, 4096, 16384, 65536}) {
        float ms = timeit([&] { k_fill<false><<<grid, 256>>>(b, n2g); });
        printf("fill  2GiB grid=%5d plain: %.3f ms -> %.2f TB/s\n", grid, ms, 2.147483648 / ms);
        ms = timeit([&] { k_fill<true><<<grid, 256>>>(b, n2g); });
        printf("fill  2GiB grid=%5d nt   : %.3f ms -> %.2f TB/s\n", grid, ms, 2.147483648 / ms);
    }
    const size_t n = ((size_t)768 << 20) / 16;        // 768 MiB per stream
    for (int grid : {2048, 8192}) {
#define MIX(R, W) { float ms = timeit([&] { k_mix<R, W, false><<<grid, 256>>>(a, b, n); }); \
        float ms2 = timeit([&] { k_mix<R, W, true><<<grid, 256>>>(a, b, n); }); \
        double gb = (R + W) * 0.805306368; \
        printf("mix r%d:w%d grid=%5d: plain %.3f ms %.2f TB/s (write part %.2f) | nt %.3f ms %.2f TB/s\n", R, W, grid, ms, gb / ms, W * 0.805306368 / ms, ms2, gb / ms2); }
        MIX(1, 1) MIX(2, 1) MIX(3, 1) MIX(1, 3) MIX(4, 1) MIX(1, 2)
    }
    // producer -> consumer inside the Infinity Cache: fill 96 MiB then read it back, alternating
    {
        const size_t ns = ((size_t)96 << 20) / 16;
        float ms = timeit([&] { k_fill<false><<<4096, 256>>>(b, ns); k_read<4><<<4096, 256>>>(b, o, ns); }, 10);
        printf("fill+read 96MiB pair: %.3f ms (fill alone %.3f, read alone %.3f)\n", ms,
               timeit([&] { k_fill<false><<<4096, 256>>>(b, ns); }, 10), timeit([&] { k_read<4><<<4096, 256>>>(b, o, ns); }, 10));
        const size_t nb = ((size_t)1 << 30) / 16;
        ms = timeit([&] { k_fill<false><<<4096, 256>>>(b, nb); k_read<4><<<4096, 256>>>(b, o, nb); }, 5);
        printf("fill+read 1GiB pair: %.3f ms\n", ms);
    }
    return 0;
}
