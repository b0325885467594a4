// Micro-benchmark: what a plain streaming read / copy sustains from HBM on this GPU (the ceiling the memory-bound kernels
// are measured against in practice). build: hipcc -O3 --offload-arch=gfx950 -o hbm_stream hbm_stream.hip ; run: ./hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>

template <int UNROLL>
__global__ __launch_bounds__(256) void read_k(const double2* __restrict__ in, double* __restrict__ out, size_t n2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        double2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 123.456) out[0] = acc;
}

__global__ __launch_bounds__(256) void copy_k(const double2* __restrict__ in, double2* __restrict__ out, size_t n2) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) out[i] = in[i];
}

int main() {
    const size_t bytes = (size_t)8 << 30;  // 8 GiB
    double2 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t n2 = bytes / 16;
    for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64}) {
        float ms;
        hipLaunchKernelGGL(read_k<8>, dim3(grid), dim3(256), 0, 0, a, (double*)b, n2);
        hipEventRecord(e0);
        hipLaunchKernelGGL(read_k<8>, dim3(grid), dim3(256), 0, 0, a, (double*)b, n2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("read  grid %6d: %7.1f GB/s\n", grid, bytes / (ms * 1e-3) / 1e9);
        hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n2);
        hipEventRecord(e0);
        hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("copy  grid %6d: %7.1f GB/s (read + write)\n", grid, 2.0 * bytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
