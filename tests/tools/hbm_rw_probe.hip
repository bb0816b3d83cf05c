// Probe (GPU box): achievable HBM rates of plain streaming kernels on MI355X -- read-only, write-only, copy, and a 1 : 10 read : write
// mix (the fan-out launch's pattern: one tile in, ten masked copies out).  16-byte accesses, fully coalesced, grid-stride.
//   hipcc --offload-arch=gfx950 -O3 tests/tools/hbm_rw_probe.hip -o .ab/hbm_rw_probe && .ab/hbm_rw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ void k_read(const u32x4* __restrict__ in, u32x4* out, size_t n) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= in[i];
    if (acc.x == 0x12345678u) out[0] = acc;
}
__global__ void k_write(u32x4* out, size_t n, int nt) {
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (nt) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
__global__ void k_copy(const u32x4* __restrict__ in, u32x4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_fan(const u32x4* __restrict__ in, u32x4* out, size_t n, size_t stride) {      // 1 read, 10 writes `stride` apart
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = in[i];
#pragma unroll
        for (int s = 0; s < 10; ++s) out[i + s * stride] = v & (0x9E3779B9u * (s + 1));
    }
}

template <typename F> static double time_ms(F f, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const size_t GB = 1ull << 30;
    const size_t bytes = 20 * GB, n = bytes / 16, n1 = (2 * GB) / 16;
    u32x4 *in = nullptr, *out = nullptr;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes + 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
    const int blocks = 256 * 8, threads = 256;
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(threads), 0, 0, in, out, n); }, 3);
    printf("read  20 GB: %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(threads), 0, 0, out, n, 0); }, 3);
    printf("write 20 GB: %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(threads), 0, 0, out, n, 1); }, 3);
    printf("write 20 GB (nt): %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(threads), 0, 0, in, out, n / 2); }, 3);
    printf("copy 10 -> 10 GB: %.3f ms  %.2f TB/s (read + write)\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_fan, dim3(blocks), dim3(threads), 0, 0, in, out, n1, n1); }, 3);
    printf("fan 2 GB -> 10 x 2 GB: %.3f ms  %.2f TB/s written\n", ms, 10.0 * 2 * GB / ms / 1e9);
    return 0;
}
