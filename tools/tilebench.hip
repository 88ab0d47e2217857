// tilebench.hip -- access-pattern microbenchmark behind lq_conv_tile.hpp (round 3).
//
// A set of NT tensors [hw][ci][co] fp32 (default 5 x 9 x 512 x 512 = 47 MB, the size of the ResNet-18-like weight set) is read
// (and optionally a second stream of the same shape, and optionally written) by kernels that differ ONLY in the access pattern:
//   rows      contiguous: block b reads 256 consecutive float4 (x U), the shape of the generic row traversals
//   tile      conv tile: wave = 8 channels x 32 o x 9 taps, lane (c_lw, o4) reads one float4 per tap (9 independent loads)
//   tile64    conv tile with 64 output channels per tile: wave = 4 channels x 64 o x 9 taps (256-byte row segments)
//   tileR     `tile`, the 9 taps in R rounds (fewer loads in flight per wave)
// Build:  hipcc -O3 --offload-arch=gfx950 -o tools/tilebench tools/tilebench.hip      Run: tools/tilebench [nt] [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int HW = 9, CI = 512, CO = 512;
constexpr size_t TEN = (size_t)HW * CI * CO;

struct Set { const float* a[16]; const float* b[16]; float* o[16]; float* part; };

__device__ __forceinline__ float red(float4 v) { return v.x + v.y + v.z + v.w; }

template <int U, int STREAMS, int WRITE>
__global__ __launch_bounds__(256) void k_rows(Set s, int blocks_per_tensor) {
    const int t = blockIdx.x / blocks_per_tensor, b = blockIdx.x % blocks_per_tensor;
    const float4* A = reinterpret_cast<const float4*>(s.a[t]);
    const float4* B = reinterpret_cast<const float4*>(s.b[t]);
    float4 x[U], y[U];
    const size_t base = (size_t)b * 256 * U + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        x[u] = A[base + u * 256];
        if (STREAMS == 2) y[u] = B[base + u * 256];
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        acc += red(x[u]);
        if (STREAMS == 2) acc += red(y[u]);
        if (WRITE) reinterpret_cast<float4*>(s.o[t])[base + u * 256] = x[u];
    }
    if (acc == 12345.678f) s.part[blockIdx.x] = acc;
}

// wave tile: CW channels x (OT) output channels x 9 taps; block = 4 waves = 4*CW channels
template <int OT, int ROUNDS, int STREAMS, int WRITE>
__global__ __launch_bounds__(256) void k_tile(Set s, int tiles_per_tensor) {
    constexpr int LPR = OT / 4;            // lanes per row
    constexpr int CW = 64 / LPR;           // channels per wave and pass
    const int t = blockIdx.x / tiles_per_tensor, b = blockIdx.x % tiles_per_tensor;
    constexpr int NTO = CO / OT;
    const int tci = b / NTO, to = b % NTO;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = tci * (4 * CW) + w * CW + lane / LPR, o = to * OT + (lane % LPR) * 4;
    const float* A = s.a[t];
    const float* B = s.b[t];
    float acc = 0.f;
    constexpr int PER = HW / ROUNDS;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        float4 x[PER], y[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int h = r * PER + q;
            const size_t i = ((size_t)h * CI + c) * CO + o;
            x[q] = *reinterpret_cast<const float4*>(A + i);
            if (STREAMS == 2) y[q] = *reinterpret_cast<const float4*>(B + i);
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int h = r * PER + q;
            const size_t i = ((size_t)h * CI + c) * CO + o;
            acc += red(x[q]);
            if (STREAMS == 2) acc += red(y[q]);
            if (WRITE) *reinterpret_cast<float4*>(s.o[t] + i) = x[q];
        }
    }
    if (acc == 12345.678f) s.part[blockIdx.x] = acc;
}

template <class F>
static double time_us(F launch, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 5;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    Set s;
    for (int t = 0; t < nt; ++t) {
        float *a, *b, *o;
        CK(hipMalloc(&a, TEN * 4));
        CK(hipMalloc(&b, TEN * 4));
        CK(hipMalloc(&o, TEN * 4));
        CK(hipMemset(a, 0, TEN * 4));
        CK(hipMemset(b, 0, TEN * 4));
        s.a[t] = a;
        s.b[t] = b;
        s.o[t] = o;
    }
    CK(hipMalloc(&s.part, 1 << 20));
    const double mb = nt * TEN * 4 / 1e6;
    printf("# %d tensors of %d x %d x %d fp32 = %.1f MB per stream; microseconds per launch and TB/s over all streams moved\n", nt, HW, CI, CO, mb);
#define RUN(name, streams, write, expr, grid) do { \
        const double us = time_us([&] { expr; }, iters); \
        printf("%-34s grid %6d  %8.2f us  %6.2f TB/s\n", name, (int)(grid), us, mb * ((streams) + (write)) / us); } while (0)
    {
        const int bpt1 = (int)(TEN / 4 / 256), bpt2 = bpt1 / 2, bpt4 = bpt1 / 4, bpt9 = bpt1 / 9;
        RUN("rows U=1 read1", 1, 0, (k_rows<1, 1, 0><<<dim3(nt * bpt1), dim3(256), 0, 0>>>(s, bpt1)), nt * bpt1);
        RUN("rows U=1 read2", 2, 0, (k_rows<1, 2, 0><<<dim3(nt * bpt1), dim3(256), 0, 0>>>(s, bpt1)), nt * bpt1);
        RUN("rows U=2 read2", 2, 0, (k_rows<2, 2, 0><<<dim3(nt * bpt2), dim3(256), 0, 0>>>(s, bpt2)), nt * bpt2);
        RUN("rows U=4 read2", 2, 0, (k_rows<4, 2, 0><<<dim3(nt * bpt4), dim3(256), 0, 0>>>(s, bpt4)), nt * bpt4);
        RUN("rows U=9 read2", 2, 0, (k_rows<9, 2, 0><<<dim3(nt * bpt9), dim3(256), 0, 0>>>(s, bpt9)), nt * bpt9);
        RUN("rows U=9 read1", 1, 0, (k_rows<9, 1, 0><<<dim3(nt * bpt9), dim3(256), 0, 0>>>(s, bpt9)), nt * bpt9);
        RUN("rows U=1 read1+write", 1, 1, (k_rows<1, 1, 1><<<dim3(nt * bpt1), dim3(256), 0, 0>>>(s, bpt1)), nt * bpt1);
        RUN("rows U=4 read1+write", 1, 1, (k_rows<4, 1, 1><<<dim3(nt * bpt4), dim3(256), 0, 0>>>(s, bpt4)), nt * bpt4);
        RUN("rows U=9 read1+write", 1, 1, (k_rows<9, 1, 1><<<dim3(nt * bpt9), dim3(256), 0, 0>>>(s, bpt9)), nt * bpt9);
    }
    {
        const int tpt32 = (CI / 32) * (CO / 32), tpt64 = (CI / 16) * (CO / 64), tpt128 = (CI / 8) * (CO / 128);
        RUN("tile 32o 9 in flight read1", 1, 0, (k_tile<32, 1, 1, 0><<<dim3(nt * tpt32), dim3(256), 0, 0>>>(s, tpt32)), nt * tpt32);
        RUN("tile 32o 9 in flight read2", 2, 0, (k_tile<32, 1, 2, 0><<<dim3(nt * tpt32), dim3(256), 0, 0>>>(s, tpt32)), nt * tpt32);
        RUN("tile 32o 3x3 rounds read2", 2, 0, (k_tile<32, 3, 2, 0><<<dim3(nt * tpt32), dim3(256), 0, 0>>>(s, tpt32)), nt * tpt32);
        RUN("tile 64o 9 in flight read2", 2, 0, (k_tile<64, 1, 2, 0><<<dim3(nt * tpt64), dim3(256), 0, 0>>>(s, tpt64)), nt * tpt64);
        RUN("tile 128o 9 in flight read2", 2, 0, (k_tile<128, 1, 2, 0><<<dim3(nt * tpt128), dim3(256), 0, 0>>>(s, tpt128)), nt * tpt128);
        RUN("tile 32o read1+write", 1, 1, (k_tile<32, 1, 1, 1><<<dim3(nt * tpt32), dim3(256), 0, 0>>>(s, tpt32)), nt * tpt32);
        RUN("tile 64o read1+write", 1, 1, (k_tile<64, 1, 1, 1><<<dim3(nt * tpt64), dim3(256), 0, 0>>>(s, tpt64)), nt * tpt64);
    }
    return 0;
}
