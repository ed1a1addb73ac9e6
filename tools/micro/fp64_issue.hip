// How fast can ONE wavefront issue fp64 mul / add (and DPP moves) on gfx950, as a function of the number of independent
// dependency chains?  Prints ns per instruction (4 cycles at 2.4 GHz = 1.67 ns).  Build: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off fp64_issue.hip -o fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP, int KIND>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* t, int n, double x, double y) {
    double a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = x + i + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (KIND == 0) a[i] = a[i] * x;          // v_mul_f64
                if (KIND == 1) a[i] = a[i] + y;          // v_add_f64
                if (KIND == 2) a[i] = a[i] * x + y;      // mul, add (no contraction)
                if (KIND == 3) {                         // DPP move pair + add
                    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a[i]), 0x138, 0xf, 0xf, true);
                    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a[i]), 0x138, 0xf, 0xf, true);
                    a[i] = __hiloint2double(hi, lo) + y;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += a[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}

template <int ILP, int KIND>
void run(const char* name, int per) {
    double* out;
    unsigned long long* t;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&t, 8);
    const int n = 20000;
    hipLaunchKernelGGL((k<ILP, KIND>), dim3(1), dim3(64), 0, 0, out, t, n, 1.0000001, 1e-9);
    hipLaunchKernelGGL((k<ILP, KIND>), dim3(1), dim3(64), 0, 0, out, t, n, 1.0000001, 1e-9);
    unsigned long long h = 0;
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    const double ns = (double)h * 10.0;
    printf("%-10s ILP %d: %.2f ns per instruction\n", name, ILP, ns / ((double)n * 8 * ILP * per));
    hipFree(out);
    hipFree(t);
}

int main() {
    run<1, 0>("mul", 1); run<2, 0>("mul", 1); run<4, 0>("mul", 1); run<8, 0>("mul", 1);
    run<1, 1>("add", 1); run<2, 1>("add", 1); run<4, 1>("add", 1); run<8, 1>("add", 1);
    run<1, 2>("mul+add", 2); run<2, 2>("mul+add", 2); run<4, 2>("mul+add", 2); run<8, 2>("mul+add", 2);
    run<1, 3>("dpp2+add", 3); run<2, 3>("dpp2+add", 3); run<4, 3>("dpp2+add", 3); run<8, 3>("dpp2+add", 3);
    return 0;
}
