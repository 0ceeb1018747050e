// Microbenchmark: vector-memory instruction throughput of one CU's L1 (TCP) for the access
// shapes a BVH walk produces.  Every wave issues independent loads in a loop; the figure of
// merit is CU cycles per wave-level load instruction with 16 waves resident per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: coalesced 16 B/lane; 1: 8 lanes share a random 128-B line; 2: every lane its own random
// line (16 B of it); 3: like 1 but LDS-DMA; 4: every lane its own random line, 8 B; 5: 2 lanes share 32 B
template <int MODE>
__global__ void __launch_bounds__(256) k(const char* __restrict__ buf, unsigned lines_mask, int reps, double* sink) {
    __shared__ __attribute__((aligned(16))) char lds[4][1024];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned gw = blockIdx.x * 4 + wave;
    double acc = 0.0;
    for (int r = 0; r < reps; r++) {
        size_t off;
        if (MODE == 0) off = (size_t)((hash(gw * 7919u + r) & lines_mask) & ~7u) * 128u + lane * 16u;
        else if (MODE == 1 || MODE == 3) off = (size_t)(hash((gw * 64u + (lane >> 3)) * 2654435761u + r) & lines_mask) * 128u + (lane & 7u) * 16u;
        else if (MODE == 5) off = (size_t)(hash((gw * 64u + (lane >> 1)) * 2654435761u + r) & lines_mask) * 128u + (lane & 1u) * 16u;
        else off = (size_t)(hash((gw * 64u + lane) * 2654435761u + r) & lines_mask) * 128u;
        if (MODE == 3) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + off),
                                             (__attribute__((address_space(3))) void*)(&lds[wave][0]), 16, 0, 0);
        } else if (MODE == 4) {
            acc += *(const double*)(buf + off);
        } else {
            const double2 v = *(const double2*)(buf + off);
            acc += v.x + v.y;
        }
    }
    if (MODE == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc = *(double*)&lds[wave][lane * 16]; }
    if (acc == 1.2345) sink[0] = acc;
}

template <int MODE>
int run(const char* name, const char* buf, size_t bytes, double* sink, int cus) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 2000, blocks = cus * 4;      // 16 waves per CU, one round
    const unsigned mask = (unsigned)(bytes / 128 - 1);
    float ms = 0;
    for (int it = 0; it < 2; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, mask, reps, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double instr_per_cu = 16.0 * reps;
    printf("%-52s set %8zu KB: %8.1f us  %6.1f ns per wave-instr per CU (%.1f clk @2.4GHz)  %.0f GB/s useful\n", name, bytes >> 10,
           ms * 1e3, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4,
           (double)cus * instr_per_cu * 64 * (MODE == 4 ? 8 : 16) / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    char* buf; double* sink;
    const size_t big = (size_t)512 << 20;
    CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big)); CK(hipMalloc(&sink, 64));
    for (size_t bytes : {(size_t)16 << 10, (size_t)2 << 20, (size_t)128 << 20, big}) {
        run<0>("coalesced 16 B/lane (1 KB contiguous)", buf, bytes, sink, cus);
        run<1>("8 lanes share a random 128-B line", buf, bytes, sink, cus);
        run<3>("8 lanes share a random 128-B line, LDS-DMA", buf, bytes, sink, cus);
        run<5>("2 lanes share 32 B of a random line", buf, bytes, sink, cus);
        run<2>("every lane 16 B of its own random line", buf, bytes, sink, cus);
        run<4>("every lane 8 B of its own random line", buf, bytes, sink, cus);
    }
    return 0;
}
