// micro-benchmark: cycles per wave reduction / per dependent DP op / per LDS round trip on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../rust-ida_amd/csrc/common.hpp"
using namespace idahip;

__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__global__ void k_reduce(double* out, unsigned long long* cyc, int iters) {
    double v = (double)((threadIdx.x * 2654435761u) & 1023);
    __shared__ double sh[64];
    const unsigned long long t0 = now();
    for (int i = 0; i < iters; ++i) { v = wave_max_f64(v) + (double)(threadIdx.x & 3); }
    const unsigned long long t1 = now();
    int p = threadIdx.x;
    for (int i = 0; i < iters; ++i) { p = wave_min_i32(p) + (threadIdx.x & 3); }
    const unsigned long long t2 = now();
    double x = v;
    for (int i = 0; i < iters; ++i) { x = x * 1.0000001; x = x + 1e-9; }   // dependent DP chain: mul, add
    const unsigned long long t3 = now();
    double acc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[j] = acc[j] * 1.0000001; acc[j] = acc[j] + 1e-9; }  // 8 independent chains
    }
    const unsigned long long t4 = now();
    sh[threadIdx.x & 63] = x;
    __syncthreads();
    double y = 0;
    int idx = threadIdx.x & 63;
    for (int i = 0; i < iters; ++i) { y += sh[idx]; idx = (idx + (int)y) & 63; }    // dependent LDS round trips
    const unsigned long long t5 = now();
    for (int i = 0; i < iters; ++i) { __syncthreads(); }
    const unsigned long long t6 = now();
    double s = x + y + p;
    for (int j = 0; j < 8; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; }
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 64);
    for (int blocks : {1, 256, 2048}) for (int threads : {64, 256, 512}) {
        const int iters = 200;
        hipLaunchKernelGGL(k_reduce, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost);
        printf("blocks %4d threads %3d: wave_max_f64 %.0f cyc, wave_min_i32 %.0f cyc, dep mul+add %.1f cyc, 8x indep (mul+add) %.1f cyc, LDS round trip %.0f cyc, barrier %.0f cyc\n",
               blocks, threads, h[0] / (double)iters, h[1] / (double)iters, h[2] / (double)iters, h[3] / (double)iters, h[4] / (double)iters, h[5] / (double)iters);
    }
    return 0;
}
