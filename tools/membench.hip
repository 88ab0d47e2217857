// tools/membench.hip -- development microbenchmark (not part of the product): what can two/one
// read streams and read+write streams sustain on this MI355X with the traversal geometries the
// library uses?  Build: hipcc -O3 --offload-arch=gfx950 -o membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kBlock = 256;

typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ float4 ld(const float4* p) {
    if (NT) { v4f v = __builtin_nontemporal_load((const v4f*)p); return make_float4(v.x, v.y, v.z, v.w); }
    return *p;
}
template <int NT>
__device__ __forceinline__ void st(float4* p, float4 v) {
    if (NT) { v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (v4f*)p); } else *p = v;
}

// MODE 0: read A            (sum)
// MODE 1: read A, read B    (sum)
// MODE 2: read A, write C   (copy*2)
// MODE 3: read A, read B, write C
template <int MODE, int U, int NT>
__global__ __launch_bounds__(kBlock) void k_chunk(const float* A, const float* B, float* C, float* sink, int CH) {
    const int64_t base = (int64_t)blockIdx.x * CH;
    const float4* A4 = (const float4*)(A + base);
    const float4* B4 = (const float4*)(B + base);
    float4* C4 = (float4*)(C + base);
    const int n4 = CH / 4;
    float acc = 0.f;
    for (int j0 = 0; j0 < n4; j0 += U * kBlock) {
        float4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int j = j0 + u * kBlock + threadIdx.x;
            a[u] = ld<NT>(A4 + j);
            if (MODE == 1 || MODE == 3) b[u] = ld<NT>(B4 + j);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int j = j0 + u * kBlock + threadIdx.x;
            float4 r = a[u];
            if (MODE == 1 || MODE == 3) { r.x += b[u].x; r.y += b[u].y; r.z += b[u].z; r.w += b[u].w; }
            if (MODE >= 2) st<NT>(C4 + j, r); else acc += r.x + r.y + r.z + r.w;
        }
    }
    if (MODE < 2 && acc == 123.456f) sink[0] = acc;
}

// persistent grid-stride over chunks
template <int MODE, int U, int NT>
__global__ __launch_bounds__(kBlock) void k_persist(const float* A, const float* B, float* C, float* sink, int CH, int nchunks) {
    float acc = 0.f;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t base = (int64_t)c * CH;
        const float4* A4 = (const float4*)(A + base);
        const float4* B4 = (const float4*)(B + base);
        float4* C4 = (float4*)(C + base);
        const int n4 = CH / 4;
        for (int j0 = 0; j0 < n4; j0 += U * kBlock) {
            float4 a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int j = j0 + u * kBlock + threadIdx.x;
                a[u] = ld<NT>(A4 + j);
                if (MODE == 1 || MODE == 3) b[u] = ld<NT>(B4 + j);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int j = j0 + u * kBlock + threadIdx.x;
                float4 r = a[u];
                if (MODE == 1 || MODE == 3) { r.x += b[u].x; r.y += b[u].y; r.z += b[u].z; r.w += b[u].w; }
                if (MODE >= 2) st<NT>(C4 + j, r); else acc += r.x + r.y + r.z + r.w;
            }
        }
    }
    if (MODE < 2 && acc == 123.456f) sink[0] = acc;
}

// persistent, software-pipelined: prefetch chunk k+1 while processing chunk k (U float4 per thread per stream)
template <int MODE, int U, int NT>
__global__ __launch_bounds__(kBlock) void k_pipe(const float* A, const float* B, float* C, float* sink, int nchunks) {
    constexpr int CH = U * kBlock * 4;
    float acc = 0.f;
    int c = blockIdx.x;
    if (c >= nchunks) return;
    float4 a[U], b[U];
    {
        const float4* A4 = (const float4*)(A + (int64_t)c * CH);
        const float4* B4 = (const float4*)(B + (int64_t)c * CH);
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = ld<NT>(A4 + u * kBlock + threadIdx.x); if (MODE == 1 || MODE == 3) b[u] = ld<NT>(B4 + u * kBlock + threadIdx.x); }
    }
    for (;;) {
        const int cn = c + gridDim.x;
        float4 an[U], bn[U];
        if (cn < nchunks) {
            const float4* A4 = (const float4*)(A + (int64_t)cn * CH);
            const float4* B4 = (const float4*)(B + (int64_t)cn * CH);
#pragma unroll
            for (int u = 0; u < U; ++u) { an[u] = ld<NT>(A4 + u * kBlock + threadIdx.x); if (MODE == 1 || MODE == 3) bn[u] = ld<NT>(B4 + u * kBlock + threadIdx.x); }
        }
        float4* C4 = (float4*)(C + (int64_t)c * CH);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 r = a[u];
            if (MODE == 1 || MODE == 3) { r.x += b[u].x; r.y += b[u].y; r.z += b[u].z; r.w += b[u].w; }
            if (MODE >= 2) st<NT>(C4 + u * kBlock + threadIdx.x, r); else acc += r.x + r.y + r.z + r.w;
        }
        if (cn >= nchunks) break;
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = an[u]; b[u] = bn[u]; }
        c = cn;
    }
    if (MODE < 2 && acc == 123.456f) sink[0] = acc;
}

template <int MODE, int BS, int NT>
__global__ __launch_bounds__(BS) void k_one(const float* A, const float* B, float* C, float* sink) {
    const int64_t j = (int64_t)blockIdx.x * BS + threadIdx.x;
    float4 a = ld<NT>((const float4*)A + j), b;
    if (MODE == 1 || MODE == 3) { b = ld<NT>((const float4*)B + j); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    if (MODE >= 2) st<NT>((float4*)C + j, a);
    else if (a.x + a.y + a.z + a.w == 123.456f) sink[0] = a.x;
}

// XCD-aware variant of k_one: workgroup w runs on XCD w % 8 (round-robin dispatch); XMAP 1 gives every XCD one contiguous
// eighth of the buffer (logical block = (w % 8) * (nb / 8) + w / 8) instead of every eighth block.
template <int MODE, int BS, int NT, int XMAP>
__global__ __launch_bounds__(BS) void k_one_x(const float* A, const float* B, float* C, float* sink, int nb) {
    int w = blockIdx.x;
    if (XMAP) { const int per = nb >> 3; w = (w & 7) * per + (w >> 3); }
    const int64_t j = (int64_t)w * BS + threadIdx.x;
    float4 a = ld<NT>((const float4*)A + j), b;
    if (MODE == 1 || MODE == 3) { b = ld<NT>((const float4*)B + j); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    if (MODE >= 2) st<NT>((float4*)C + j, a);
    else if (a.x + a.y + a.z + a.w == 123.456f) sink[0] = a.x;
}

// Cache-policy probe (gfx950 global_load/global_store modifiers sc0 / sc1 / nt), one float4 per thread per stream.
#define LQ_POL_KERNEL(NAME, LDPOL, STPOL)                                                                              \
    __global__ __launch_bounds__(512) void NAME(const float* A, const float* B, float* C, float* sink, int mode) {    \
        const int64_t j = (int64_t)blockIdx.x * 512 + threadIdx.x;                                                     \
        const float4* pa = (const float4*)A + j;                                                                       \
        const float4* pb = (const float4*)B + j;                                                                       \
        float4* pc = (float4*)C + j;                                                                                   \
        v4f a, b;                                                                                                      \
        if (mode == 1) {                                                                                               \
            asm volatile("global_load_dwordx4 %0, %2, off " LDPOL "\n\tglobal_load_dwordx4 %1, %3, off " LDPOL          \
                         "\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb) : "memory");                 \
            if (a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w == 123.456f) sink[0] = a.x;                              \
        } else {                                                                                                       \
            asm volatile("global_load_dwordx4 %0, %1, off " LDPOL "\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(pa) : "memory"); \
            a.x += 1.0f;                                                                                               \
            asm volatile("global_store_dwordx4 %0, %1, off " STPOL : : "v"(pc), "v"(a) : "memory");                    \
        }                                                                                                              \
    }
LQ_POL_KERNEL(k_pol_plain, "", "")
LQ_POL_KERNEL(k_pol_nt, "nt", "nt")
LQ_POL_KERNEL(k_pol_sc1nt, "sc1 nt", "sc1 nt")
LQ_POL_KERNEL(k_pol_sc0sc1nt, "sc0 sc1 nt", "sc0 sc1 nt")
LQ_POL_KERNEL(k_pol_sc1, "sc1", "sc1")
LQ_POL_KERNEL(k_pol_sc0sc1, "sc0 sc1", "sc0 sc1")
LQ_POL_KERNEL(k_pol_ldnt_stsc, "nt", "sc0 sc1 nt")
LQ_POL_KERNEL(k_pol_ldsc_stnt, "sc0 sc1 nt", "nt")

int main(int argc, char** argv) {
    const int64_t N = 256ll * 3 * 224 * 224;
    const int SETS = 4;
    const size_t pad = argc > 1 ? (size_t)atol(argv[1]) : 0;   // extra bytes between buffers (de-alias test)
    std::vector<float*> A(SETS), B(SETS), C(SETS);
    char* pool;
    size_t each = N * 4 + pad;
    each = (each + 255) / 256 * 256;
    CK(hipMalloc(&pool, each * 3 * SETS + 4096));
    CK(hipMemset(pool, 0, each * 3 * SETS));
    for (int i = 0; i < SETS; ++i) { A[i] = (float*)(pool + each * (3 * i)); B[i] = (float*)(pool + each * (3 * i + 1)); C[i] = (float*)(pool + each * (3 * i + 2)); }
    float* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 40;
    auto report = [&](const char* name, double bytes, float ms) {
        printf("%-44s %8.1f us  %7.0f GB/s\n", name, ms * 1000.0 / iters, bytes * iters / (ms * 1e-3) / 1e9);
    };
    auto run = [&](const char* name, double bytes, auto launch) {
        for (int w = 0; w < 5; ++w) launch(w % SETS);
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) launch(it % SETS);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGetLastError()); report(name, bytes, ms);
    };
    const double b2 = N * 8.0, b3 = N * 12.0;
#define TB(label, MODE, bytes, BS) { char nm[96]; const int nb = (int)(N / 4 / BS); \
        snprintf(nm, 96, "one BS%d %s", BS, label); \
        run(nm, bytes, [&](int k){ hipLaunchKernelGGL((k_one<MODE, BS, 0>), dim3(nb), dim3(BS), 0, 0, A[k], B[k], C[k], sink); }); \
        snprintf(nm, 96, "one BS%d %s nt", BS, label); \
        run(nm, bytes, [&](int k){ hipLaunchKernelGGL((k_one<MODE, BS, 1>), dim3(nb), dim3(BS), 0, 0, A[k], B[k], C[k], sink); }); }
    for (int rep = 0; rep < 2; ++rep) {
    TB("read2", 1, b2, 256) TB("read2", 1, b2, 512) TB("read2", 1, b2, 1024)
    TB("read1+write1", 2, b2, 256) TB("read1+write1", 2, b2, 512) TB("read1+write1", 2, b2, 1024)
    TB("read2+write1", 3, b3, 256) TB("read2+write1", 3, b3, 512) TB("read2+write1", 3, b3, 1024)
    }
#define POL(K, label) { const int nb = (int)(N / 4 / 512); \
        run("policy " label "  read1+write1", b2, [&](int k){ hipLaunchKernelGGL(K, dim3(nb), dim3(512), 0, 0, A[k], B[k], C[k], sink, 2); }); \
        run("policy " label "  read2", b2, [&](int k){ hipLaunchKernelGGL(K, dim3(nb), dim3(512), 0, 0, A[k], B[k], C[k], sink, 1); }); }
    for (int rep = 0; rep < 2; ++rep) {
        POL(k_pol_plain, "(none)      ") POL(k_pol_nt, "nt          ") POL(k_pol_sc1nt, "sc1 nt      ") POL(k_pol_sc0sc1nt, "sc0 sc1 nt  ")
        POL(k_pol_sc1, "sc1         ") POL(k_pol_sc0sc1, "sc0 sc1     ") POL(k_pol_ldnt_stsc, "ld nt/st sc*") POL(k_pol_ldsc_stnt, "ld sc*/st nt")
    }
    for (int rep = 0; rep < 2; ++rep) {     // nb = N/4/512 = 18816 is a multiple of 8
        const int nb = (int)(N / 4 / 512);
        run("xcd round-robin  BS512 read1+write1 nt", b2, [&](int k){ hipLaunchKernelGGL((k_one_x<2, 512, 1, 0>), dim3(nb), dim3(512), 0, 0, A[k], B[k], C[k], sink, nb); });
        run("xcd contiguous   BS512 read1+write1 nt", b2, [&](int k){ hipLaunchKernelGGL((k_one_x<2, 512, 1, 1>), dim3(nb), dim3(512), 0, 0, A[k], B[k], C[k], sink, nb); });
        run("xcd round-robin  BS512 read2 nt", b2, [&](int k){ hipLaunchKernelGGL((k_one_x<1, 512, 1, 0>), dim3(nb), dim3(512), 0, 0, A[k], B[k], C[k], sink, nb); });
        run("xcd contiguous   BS512 read2 nt", b2, [&](int k){ hipLaunchKernelGGL((k_one_x<1, 512, 1, 1>), dim3(nb), dim3(512), 0, 0, A[k], B[k], C[k], sink, nb); });
    }
    {
        const int nb = (int)(N / 1024); char nm[96];
        snprintf(nm, 96, "ALTERNATE BS256 nt: R1W1 then R2 (616MB)");
        run(nm, N * 16.0, [&](int k){
            hipLaunchKernelGGL((k_one<2, 256, 1>), dim3(nb), dim3(256), 0, 0, A[k], B[k], C[k], sink);
            hipLaunchKernelGGL((k_one<1, 256, 1>), dim3(nb), dim3(256), 0, 0, A[k], B[k], C[k], sink); });
    }
    return 0;
}
