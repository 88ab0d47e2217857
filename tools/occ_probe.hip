#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
// how many blocks of 256 threads with BYTES of static LDS are resident per CU at once?  Every block stamps its start and end.
template <int BYTES> __global__ __launch_bounds__(256) void k(unsigned long long* ts, float* o) {
    __shared__ float s[BYTES / 4];
    if (threadIdx.x == 0) ts[2 * blockIdx.x] = wall_clock64();
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float acc = 0.f;
    for (int i = 0; i < 3000; ++i) { acc += s[(threadIdx.x * 7 + i) % (BYTES / 4)]; __builtin_amdgcn_s_sleep(8); }
    o[blockIdx.x * 256 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ts[2 * blockIdx.x + 1] = wall_clock64();
}
template <int BYTES> void run(int nb) {
    unsigned long long* ts; float* o;
    hipMalloc(&ts, nb * 16); hipMalloc(&o, nb * 256 * 4); hipMemset(ts, 0, nb * 16);
    hipLaunchKernelGGL(k<BYTES>, dim3(nb), dim3(256), 0, 0, ts, o); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<BYTES>, dim3(nb), dim3(256), 0, 0, ts, o); hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * nb); hipMemcpy(h.data(), ts, nb * 16, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull; for (int i = 0; i < nb; ++i) t0 = std::min(t0, h[2 * i]);
    unsigned long long first_end = ~0ull; for (int i = 0; i < nb; ++i) first_end = std::min(first_end, h[2 * i + 1]);
    int early = 0; for (int i = 0; i < nb; ++i) if (h[2 * i] < first_end) ++early;      // blocks that started before any block ended
    printf("%6d B LDS: %d of %d blocks resident at once = %.2f per CU\n", BYTES, early, nb, early / 256.0);
    hipFree(ts); hipFree(o);
}
int main() { run<37376>(2048); run<32768>(2048); run<32256>(2048); run<31744>(2048); run<30720>(2048); run<28672>(2048); run<27136>(2048); return 0; }
