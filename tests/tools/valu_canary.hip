// VALU canary (development probe): every lane runs a long chain of fp32 fma / sqrt / reciprocal / division / exp operations on its own
// seed and stores the result -- no LDS, no loads in the chain.  The host compares the results of a run beside OTHER kernels with the results
// of a run alone, bit for bit, and reports which lanes differ (posterior kernels beside this library's convolutions write wrong
// values for aligned runs of 16 slots).
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tests/tools/valu_canary.hip -o tests/tools/libvalu_canary.so
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ __launch_bounds__(256) void valu_canary_kernel(int iters, int mode, float* out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float x = 1.0f + (float)(t & 1023) * 0.001f, y = 0.5f + (float)(t & 255) * 0.002f;
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {                      // plain fma chain
            x = __builtin_fmaf(x, 0.9990234f, y); y = __builtin_fmaf(y, 0.5f, 0.25f * x);
        } else if (mode == 1) {               // transcendental unit: sqrt, rcp, exp
            const float s = sqrtf(x * x + 1.0f);
            y = 1.0f / (s + y * y);
            x = expf(-y) + 0.5f * x;
        } else {                              // IEEE division (v_div_scale / v_div_fmas / v_div_fixup, denormal mode switches)
            y = (x + 1.0f) / (y + 2.0f);
            x = (y + 3.0f) / (x + 1.5f);
        }
    }
    out[t] = x + y;
}
extern "C" int valu_canary_run(int blocks, int iters, int mode, float* host_out) {
    static hipStream_t st = nullptr;
    static float* d = nullptr; static int cap = 0;
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 1;
    if (cap < blocks) { if (d) hipFree(d); if (hipMalloc(&d, (size_t)blocks * 256 * 4) != hipSuccess) return 2; cap = blocks; }
    hipLaunchKernelGGL(valu_canary_kernel, dim3(blocks), dim3(256), 0, st, iters, mode, d);
    if (hipMemcpyAsync(host_out, d, (size_t)blocks * 256 * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return 3;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : 4;
}
