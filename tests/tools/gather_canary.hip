// Gather canary (development probe): every lane sums 16-byte and 8-byte records gathered from a static table at hashed
// indices -- the access pattern of post_fuse_kernel (one 64-byte Welford record + one 40-byte parameter record per kept anchor) without its
// arithmetic.  The host compares a run beside OTHER kernels with a run alone, bit for bit.
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tests/tools/gather_canary.hip -o tests/tools/libgather_canary.so
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ __launch_bounds__(256) void gather_canary_kernel(const float4* __restrict__ t4, const float2* __restrict__ t2, unsigned n4, unsigned n2, int iters, float* out) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    unsigned h = t * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const unsigned r = (h >> 4) % (n4 / 4);                 // a 64-byte record
        const float4 a = t4[r * 4], b = t4[r * 4 + 1], c = t4[r * 4 + 2], d = t4[r * 4 + 3];
        const unsigned s = (h >> 7) % (n2 / 5);                 // a 40-byte record
        float e = 0.f;
#pragma unroll
        for (int q = 0; q < 5; ++q) { const float2 v = t2[s * 5 + q]; e += v.x - v.y; }
        acc += (a.x + a.y + a.z + a.w) + (b.x - b.y) + (c.z * 0.5f + c.w) + (d.x + d.y) + e;
    }
    out[t] = acc;
}
extern "C" int gather_canary_run(int blocks, int iters, float* host_out) {
    static hipStream_t st = nullptr;
    static float4* t4 = nullptr; static float2* t2 = nullptr; static float* d = nullptr; static int cap = 0;
    const unsigned n4 = 4u << 20, n2 = 5u << 20;             // 64 MB + 40 MB of records
    if (!st) {
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 1;
        if (hipMalloc(&t4, (size_t)n4 * 16) != hipSuccess || hipMalloc(&t2, (size_t)n2 * 8) != hipSuccess) return 2;
        float* h = (float*)malloc((size_t)n4 * 16);
        unsigned x = 1u;
        for (size_t i = 0; i < (size_t)n4 * 4; ++i) { x = x * 1103515245u + 12345u; h[i] = (float)((x >> 9) & 0xFFFF) * (1.0f / 65536.0f); }
        if (hipMemcpy(t4, h, (size_t)n4 * 16, hipMemcpyHostToDevice) != hipSuccess) return 3;
        if (hipMemcpy(t2, h, (size_t)n2 * 8, hipMemcpyHostToDevice) != hipSuccess) return 3;
        free(h);
    }
    if (cap < blocks) { if (d) (void)hipFree(d); if (hipMalloc(&d, (size_t)blocks * 256 * 4) != hipSuccess) return 4; cap = blocks; }
    hipLaunchKernelGGL(gather_canary_kernel, dim3(blocks), dim3(256), 0, st, t4, t2, n4, n2, iters, d);
    if (hipMemcpyAsync(host_out, d, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return 5;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : 6;
}
