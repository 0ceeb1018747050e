// Microbenchmark: dependent-load latency (pointer chase over 128-byte records) for working
// sets in L2 / MALL / HBM, with one wave alone on the device and with the device full, plus the
// ratio of clock64() ticks to the 100 MHz wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// every lane chases its own chain (lanes start at different records)
__global__ void __launch_bounds__(64) chase(const unsigned* __restrict__ next, unsigned n, int hops, unsigned long long* out) {
    unsigned cur = (unsigned)(((unsigned long long)(blockIdx.x * 64 + threadIdx.x) * 2654435761ull) % n);
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
    for (int h = 0; h < hops; h++) cur = next[(size_t)cur * 32u];
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned long long)(c1 - c0); out[1] = (unsigned long long)(w1 - w0); }
    if (cur == 0xFFFFFFFFu) out[2] = cur;
}

int main() {
    const size_t max_recs = (size_t)4 << 20;            // 4M records x 128 B = 512 MB
    unsigned* d_next; unsigned long long* d_out;
    CK(hipMalloc(&d_next, max_recs * 128)); CK(hipMalloc(&d_out, 64));
    std::mt19937 rng(1);
    for (size_t recs : {(size_t)1 << 13, (size_t)1 << 17, (size_t)1 << 20, max_recs}) {
        std::vector<unsigned> perm(recs); std::iota(perm.begin(), perm.end(), 0u);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<unsigned> host(recs * 32, 0u);
        for (size_t i = 0; i < recs; i++) host[(size_t)perm[i] * 32] = perm[(i + 1) % recs];   // one big cycle
        CK(hipMemcpy(d_next, host.data(), recs * 128, hipMemcpyHostToDevice));
        for (int blocks : {1, 256 * 16}) {
            const int hops = 2000;
            unsigned long long out[2];
            for (int it = 0; it < 2; it++) {
                hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d_next, (unsigned)recs, hops, d_out);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
            printf("set %7zu KB, %5d waves: %7.1f ticks/hop, %7.1f ns/hop (clock64 runs at %.0f MHz)\n", recs * 128 >> 10, blocks,
                   (double)out[0] / hops, (double)out[1] * 10.0 / hops, (double)out[0] / ((double)out[1] * 10.0) * 1e3);
        }
    }
    return 0;
}
