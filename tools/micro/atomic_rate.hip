// Microbenchmark: cost of same-address device-scope atomics issued by one lane per wave,
// versus the same number of atomics spread over many addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_same(unsigned* ctr, unsigned long long* ctr64, int reps, int spread, unsigned* sink) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned acc = 0;
    for (int r = 0; r < reps; r++) {
        unsigned v = 0;
        if (lane == 0) {
            if (spread >= 0) v = atomicAdd(ctr + (size_t)(spread ? (wave % spread) * 32u : 0u), 1u);
            else v = (unsigned)atomicAdd(ctr64, 0x100000001ull);
        }
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        acc += v;
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main() {
    unsigned* ctr; unsigned long long* c64; unsigned* sink;
    CK(hipMalloc(&ctr, 1 << 20)); CK(hipMalloc(&c64, 64)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(ctr, 0, 1 << 20)); CK(hipMemset(c64, 0, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 2048;
    struct { int reps, spread; const char* name; } cases[] = {
        {1, 0, "8192 waves x 1 atomic, same address (u32)"},
        {4, 0, "8192 waves x 4 atomics, same address (u32)"},
        {16, 0, "8192 waves x 16 atomics, same address (u32)"},
        {16, -1, "8192 waves x 16 atomics, same address (u64)"},
        {16, 8, "8192 waves x 16 atomics, 8 addresses"},
        {16, 64, "8192 waves x 16 atomics, 64 addresses"},
        {16, 1024, "8192 waves x 16 atomics, 1024 addresses"},
    };
    for (auto& c : cases) {
        for (int it = 0; it < 2; it++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_same, dim3(blocks), dim3(256), 0, 0, ctr, c64, c.reps, c.spread, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = 8192.0 * c.reps;
        printf("%-48s %8.1f us  %7.1f ns/atomic\n", c.name, ms * 1e3, ms * 1e6 / n);
    }
    // empty launch for reference
    for (int it = 0; it < 2; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_same, dim3(blocks), dim3(256), 0, 0, ctr, c64, 0, 0, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-48s %8.1f us\n", "empty 2048x256 launch", ms * 1e3);
    return 0;
}
