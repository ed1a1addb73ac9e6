// Micro-benchmark: how many bytes per second can ONE compute unit pull from HBM with 1 KiB-per-wave loads
// (buffer_load_dwordx4, 16 B per lane) as a function of waves per CU and loads in flight per wave?
// Build: hipcc --offload-arch=gfx950 -O3 -o cu_stream cu_stream.hip ; run: ./cu_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(1024) void k(const u32x4* __restrict__ src, u32x4* sink, size_t blocks_per_wave, int waves_total) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const u32x4* p = src + (size_t)wave * blocks_per_wave * 64 + lane;
    u32x4 r[D];
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < D; d++) r[d] = __builtin_nontemporal_load(p + (size_t)d * 64);
    for (size_t i = D; i < blocks_per_wave; i += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            acc += r[d];
            r[d] = __builtin_nontemporal_load(p + (i + d) * 64);
        }
    }
#pragma unroll
    for (int d = 0; d < D; d++) acc += r[d];
    if (acc.x == 0x12345678u) sink[wave * 64 + lane] = acc;
}
template <int D>
void run(const u32x4* src, u32x4* sink, int waves_per_cu, int cus, size_t blocks_per_wave) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int wt = waves_per_cu * cus;
    k<D><<<cus, 64 * waves_per_cu>>>(src, sink, blocks_per_wave, wt);
    hipEventRecord(a);
    k<D><<<cus, 64 * waves_per_cu>>>(src, sink, blocks_per_wave, wt);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double gb = (double)wt * blocks_per_wave * 1024 / 1e9;
    printf("CUs %3d waves/CU %2d in-flight %2d : %8.1f GB/s total  %7.1f GB/s per CU  %6.1f GB/s per wave\n", cus,
           waves_per_cu, D, gb / (ms * 1e-3), gb / (ms * 1e-3) / cus, gb / (ms * 1e-3) / wt);
}
int main() {
    const size_t bytes = size_t(4) << 30;
    u32x4 *src, *sink;
    hipMalloc(&src, bytes);
    hipMalloc(&sink, 1 << 24);
    hipMemset(src, 1, bytes);
    for (int cus : {1, 256}) {
        const size_t bpw = cus == 1 ? 40000 : 8000;
        for (int w : {1, 2, 4, 8}) {
            if ((size_t)w * cus * bpw * 1024 > bytes) continue;
            run<8>(src, sink, w, cus, bpw);
            run<16>(src, sink, w, cus, bpw);
            run<32>(src, sink, w, cus, bpw);
            run<60>(src, sink, w, cus, bpw);
        }
    }
    return 0;
}
