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
        } else if (mode == 2) {               // IEEE division (v_div_scale / v_div_fmas / v_div_fixup, denormal mode switches)
            y = (x + 1.0f) / (y + 2.0f);
            x = (y + 3.0f) / (x + 1.5f);
        } else if (mode == 3) {               // integer multiplies: v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 (address arithmetic)
            unsigned a = __float_as_uint(x), b = __float_as_uint(y);
            unsigned long long z = (unsigned long long)a * 2654435761ull + b;
            a = (unsigned)(z >> 13) * 40503u + (unsigned)z; b = __umulhi(a, b | 1u) + (unsigned)(z >> 32);
            x = __uint_as_float((a & 0x007FFFFFu) | 0x3F800000u); y = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
        } else if (mode == 4) {               // fp64 fma chain
            double dx = x, dy = y;
            dx = __builtin_fma(dx, 0.99902343751, dy); dy = __builtin_fma(dy, 0.5, 0.25 * dx);
            x = (float)dx; y = (float)dy;
        } else if (mode == 5) {               // packed fp32: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (what the 4x4 matrix code of the posterior compiles to)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p = {x, y}, q = {y, x};
            p = p * (f2){0.9990234f, 0.5f} + q * (f2){0.25f, 0.125f};
            q = q * p + (f2){0.001f, 0.002f};
            p = p - q * (f2){0.5f, 0.25f};
            x = p.x * 0.5f + 0.5f; y = q.y * 0.25f + p.y * 0.25f + 0.3f;
            x = x - __builtin_floorf(x) + 0.5f; y = y - __builtin_floorf(y) + 0.5f;
        } else {                              // 64-bit address arithmetic: v_lshl_add_u64 / v_mov_b64
            unsigned long long pa = (unsigned long long)__float_as_uint(x), pb_ = (unsigned long long)__float_as_uint(y) << 7;
            pa = (pa << 4) + pb_;  pb_ = (pb_ << 2) + pa;  pa = (pa << 3) + (pb_ >> 5);
            x = __uint_as_float(((unsigned)(pa >> 9) & 0x007FFFFFu) | 0x3F800000u); y = __uint_as_float(((unsigned)(pb_ >> 17) & 0x007FFFFFu) | 0x3F800000u);
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
