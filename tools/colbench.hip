// colbench.hip -- access-pattern microbenchmark behind lq_batch_cols.hpp (round 4).
//
// NT matrices [R][C] fp32 (default 5 x 512 x 4608 = 47 MB per stream: the 3x3x512x512 kernels of the ResNet-18-like set, stored OIHW)
// are READ TWICE (P and dy) by math-free kernels that differ only in how a 256-thread block walks its tile:
//   stack   four waves stacked on the rows of a 256-column tile (wave w: rows w, w+4, ...; 1 KB per wave and row)  -- lq_batch_cols.hpp
//   side    four waves side by side on a 1024-column tile (each wave walks every row of the block; 4 KB contiguous per block and row)
//   pair    2 x 2: two waves side by side (512 columns), two stacked
//   flat    contiguous: the block reads RB*256 consecutive float4 (what a row stream does; no column structure) -- the ceiling
// U rows (float4 per lane and stream) in flight per wave and stage; PIPE: next stage issued before the current one is consumed.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/colbench tools/colbench.hip     Run: tools/colbench [nt] [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int R = 512, C = 4608;
constexpr size_t TEN = (size_t)R * C;
struct Set { const float* a[16]; const float* b[16]; float* part; };

__device__ __forceinline__ float red(float4 v) { return v.x + v.y + v.z + v.w; }

// WX waves side by side, WY = 4 / WX stacked; tile = (64*4*WX) columns x RB rows
template <int WX, int U, int PIPE>
__global__ __launch_bounds__(256) void k_tile(Set s, int RB, int nbx, int blocks_per_tensor) {
    constexpr int WY = 4 / WX;
    const int t = blockIdx.x / blocks_per_tensor, b = blockIdx.x % blocks_per_tensor;
    const int by = b / nbx, bx = b % nbx;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wx = w % WX, wy = w / WX;
    const int col = (bx * WX + wx) * 256 + lane * 4;
    const bool active = col < C;
    unsigned voff = (active ? col : 0) * 4u;
    const int r0 = by * RB, r1 = min(r0 + RB, R);
    const char* A = reinterpret_cast<const char*>(s.a[t]);
    const char* B = reinterpret_cast<const char*>(s.b[t]);
    const size_t pitch = (size_t)C * 4, step = pitch * WY;
    const char* lastA = A + (size_t)(r1 - 1) * pitch;
    const char* lastB = B + (size_t)(r1 - 1) * pitch;
    float4 x0[U], y0[U], x1[PIPE ? U : 1], y1[PIPE ? U : 1];
    auto load = [&](float4* x, float4* y, int r, const char* pa, const char* pb) {
        asm volatile("" : "+v"(voff));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = r + WY * u < r1;
            const char* qa = in ? pa + (size_t)u * step : lastA;
            const char* qb = in ? pb + (size_t)u * step : lastB;
            x[u] = *reinterpret_cast<const float4*>(qa + voff);
            y[u] = *reinterpret_cast<const float4*>(qb + voff);
        }
    };
    float acc = 0.f;
    auto eat = [&](const float4* x, const float4* y, int r) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + WY * u < r1) acc += red(x[u]) * red(y[u]);
    };
    int r = r0 + wy;
    const char* pa = A + (size_t)r * pitch;
    const char* pb = B + (size_t)r * pitch;
    const size_t stage = (size_t)U * step;
    load(x0, y0, r, pa, pb);
    if (r < r1) {
        if (PIPE) {
            for (;;) {
                load(x1, y1, r + WY * U, pa + stage, pb + stage);
                __builtin_amdgcn_sched_barrier(0);
                eat(x0, y0, r);
                r += WY * U; pa += stage; pb += stage;
                if (r >= r1) break;
                load(x0, y0, r + WY * U, pa + stage, pb + stage);
                __builtin_amdgcn_sched_barrier(0);
                eat(x1, y1, r);
                r += WY * U; pa += stage; pb += stage;
                if (r >= r1) break;
            }
        } else {
            for (;;) {
                eat(x0, y0, r);
                r += WY * U; pa += stage; pb += stage;
                if (r >= r1) break;
                load(x0, y0, r, pa, pb);
            }
        }
    }
    if (acc == 12345.678f) s.part[blockIdx.x] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void k_flat(Set s, int blocks_per_tensor) {
    const int t = blockIdx.x / blocks_per_tensor, b = blockIdx.x % blocks_per_tensor;
    const float4* A = reinterpret_cast<const float4*>(s.a[t]);
    const float4* B = reinterpret_cast<const float4*>(s.b[t]);
    float4 x[U], y[U];
    const size_t base = (size_t)b * 256 * U + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        x[u] = A[base + u * 256];
        y[u] = B[base + u * 256];
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc += red(x[u]) * red(y[u]);
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
    const int same = argc > 3 ? atoi(argv[3]) : 0;      // 1: allocate the second stream 1 MB + 4 KB further (breaks identical channel / bank mapping of P and dy)
    Set s;
    for (int t = 0; t < nt; ++t) {
        float *a, *b;
        CK(hipMalloc(&a, TEN * 4));
        CK(hipMalloc(&b, TEN * 4 + (2 << 20)));
        CK(hipMemset(a, 0, TEN * 4));
        CK(hipMemset(b, 0, TEN * 4 + (2 << 20)));
        s.a[t] = a;
        s.b[t] = same ? b + ((1 << 20) + 4096) / 4 : b;
    }
    CK(hipMalloc(&s.part, 1 << 22));
    const double mb = 2.0 * nt * TEN * 4 / 1e6;
    printf("# %d matrices of %d x %d fp32 read twice = %.1f MB per launch; second stream offset %s; us per launch (hipEvent over %d back-to-back launches), TB/s\n",
           nt, R, C, mb, same ? "1 MB + 4 KB" : "0 (same alignment)", iters);
#define RUNT(WX, U, PIPE, RB) do { \
        const int nbx = (C + 256 * WX - 1) / (256 * WX), nby = (R + (RB) - 1) / (RB), bpt = nbx * nby; \
        const double us = time_us([&] { k_tile<WX, U, PIPE><<<dim3(nt * bpt), dim3(256), 0, 0>>>(s, RB, nbx, bpt); }, iters); \
        printf("tile %4d cols  WX=%d U=%d pipe=%d RB=%3d  grid %6d  %8.2f us  %6.2f TB/s\n", 256 * WX, WX, U, PIPE, RB, nt * bpt, us, mb / us); } while (0)
    {
        const int b1 = (int)(TEN / 4 / 256);
        double us = time_us([&] { k_flat<1><<<dim3(nt * b1), dim3(256), 0, 0>>>(s, b1); }, iters);
        printf("flat U=1                                  grid %6d  %8.2f us  %6.2f TB/s\n", nt * b1, us, mb / us);
        us = time_us([&] { k_flat<4><<<dim3(nt * b1 / 4), dim3(256), 0, 0>>>(s, b1 / 4); }, iters);
        printf("flat U=4                                  grid %6d  %8.2f us  %6.2f TB/s\n", nt * b1 / 4, us, mb / us);
        us = time_us([&] { k_flat<8><<<dim3(nt * b1 / 8), dim3(256), 0, 0>>>(s, b1 / 8); }, iters);
        printf("flat U=8                                  grid %6d  %8.2f us  %6.2f TB/s\n", nt * b1 / 8, us, mb / us);
    }
    RUNT(1, 4, 0, 16); RUNT(1, 4, 0, 32); RUNT(1, 4, 0, 48); RUNT(1, 4, 0, 64);
    RUNT(1, 2, 1, 16); RUNT(1, 2, 1, 32); RUNT(1, 2, 1, 48); RUNT(1, 2, 1, 64);
    RUNT(1, 8, 0, 32); RUNT(1, 1, 1, 32); RUNT(1, 4, 1, 32);
    RUNT(2, 4, 0, 8); RUNT(2, 4, 0, 16); RUNT(2, 4, 0, 24); RUNT(2, 2, 1, 16); RUNT(2, 2, 1, 24);
    RUNT(4, 4, 0, 4); RUNT(4, 4, 0, 8); RUNT(4, 4, 0, 12); RUNT(4, 8, 0, 8); RUNT(4, 2, 1, 8); RUNT(4, 2, 1, 12); RUNT(4, 4, 1, 8); RUNT(4, 4, 1, 16);
    return 0;
}
