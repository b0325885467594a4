// Micro-benchmark: what the fp64 vector pipe of this GPU sustains for register-only work, as a function of waves per SIMD:
//   mode 0: fused multiply-add      acc[i] = fma(-a, acc[i+8], acc[i])
//   mode 1: multiply then subtract  p = a * acc[i+8]; acc[i] = acc[i] - p   (the reference's arithmetic, -ffp-contract=off)
//   mode 2: the same with the 16 products computed first and the 16 subtractions after them
//   mode 3: mode 1 with the common factor in an SGPR pair (v_mul_f64 v, s, v): the operand form of lu_wavepanel_kernel's U entries
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o fp64_peak fp64_peak.hip ; run: ./fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, const double* in, int iters) {
    double acc[16];
    double a = in[threadIdx.x & 7];
    if (MODE == 3) {  // wave-uniform, in SGPRs
        const int lo = __builtin_amdgcn_readfirstlane(__double2loint(a)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(a));
        a = __hiloint2double(hi, lo);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i] = in[(threadIdx.x + i) & 63];
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(-a, acc[(i + 8) & 15], acc[i]);
            } else if (MODE == 1 || MODE == 3) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = acc[i] - a * acc[(i + 8) & 15];
            } else {
                double p[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) p[i] = a * acc[(i + 8) & 15];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = acc[i] - p[i];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int wgs_per_cu, int cus, double* out, double* in) {
    const int iters = 20000;
    const int grid = cus * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, in, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)grid * 256 * iters * 64;  // elementary updates (one multiply and one add/subtract each)
    const double inst = (double)grid * 4 * iters * 64 * (MODE == 0 ? 1 : 2);  // wave instructions
    const double simd_seconds = (double)cus * 4 * ms * 1e-3;
    printf("mode %d  waves/SIMD %d  %8.2f ms  %6.2f TFLOP/s (2 flops per update)  %5.2f ns per wave instruction per SIMD\n", MODE, wgs_per_cu, ms,
           2 * ops / (ms * 1e-3) / 1e12, simd_seconds / (inst / 1.0) * 1e9);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
    double *out, *in;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    hipMalloc(&in, 64 * 8);
    std::vector<double> h(64);
    for (int i = 0; i < 64; ++i) h[i] = 1.0 + 1e-9 * i;
    hipMemcpy(in, h.data(), 64 * 8, hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; ++w) {
        run<0>(w, cus, out, in);
        run<1>(w, cus, out, in);
        run<2>(w, cus, out, in);
        run<3>(w, cus, out, in);
    }
    return 0;
}
