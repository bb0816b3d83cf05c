// Synthetic company for tests/tools/side_race.py: long-running kernels of ONE instruction class each, to find which
// class of this library's backbone kernels disturbs a posterior wave on the same SIMD.
//   mode 0: v_cvt_pk_bf16_f32      mode 1: packed 16-bit integer ops (v_pk_max_i16 / v_pk_min_u16 / v_pk_mul_lo_u16 / v_pk_sub_u16)
//   mode 2: v_mfma_f32_32x32x16_bf16     mode 3: LDS-DMA (global_load_lds_dwordx4) + ds_read_b128     mode 4: v_bitop3_b32 + v_perm_b32
//   mode 5: the transcendental unit (v_rcp_f32 / v_rcp_iflag_f32 / v_exp_f32 / v_sqrt_f32)
//   mode 6: packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)     mode 7: packed fp32 between bf16 MFMAs (round 6)
//   modes 8-12: ONE kind of instruction between bf16 MFMAs: v_pk_max_i16 / v_cvt_pk_bf16_f32 / v_mov_b64 / v_pk_mul_f32 / v_fma_f64
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tests/tools/noise_kernels.hip -o tests/tools/libnoise_kernels.so
#include <hip/hip_runtime.h>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) short s16x2;
__global__ __launch_bounds__(256) void noise_kernel(int mode, int iters, const float* __restrict__ src, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = blockIdx.x * 256 + threadIdx.x;
    float x = 1.0f + (t & 255) * 0.001f, y = 0.5f + (t & 63) * 0.002f;
    unsigned u = (unsigned)t * 2654435761u, v = u ^ 0x9E3779B9u;
    f32x16 acc = {0};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { unsigned p; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y)); x += __uint_as_float((p << 16)) * 1e-6f; y += 1e-6f; }
        } else if (mode == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                unsigned p, q, r, s_;
                asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(p) : "v"(u), "v"(v));
                asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(q) : "v"(p), "v"(v));
                asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(q), "v"(u));
                asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(s_) : "v"(r), "v"(p));
                u = s_ + 1u; v ^= r;
            }
        } else if (mode == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else if (mode == 3) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)(t * 4 + ((i * 4 + k) & 255) * 262144) & 0xFFFFFF)),
                                                 (__attribute__((address_space(3))) void*)(lds + (k * 256 + (threadIdx.x & ~63)) * 16), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4 r = *reinterpret_cast<const float4*>(lds + threadIdx.x * 16);
            x += r.x * 1e-9f;
        } else if (mode == 6 || mode == 7) {  // 6: packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32); 7: the same between MFMAs (a conv epilogue's mix)
            typedef __attribute__((ext_vector_type(2))) float f2;
            f2 p = {x, y}, k1 = {0.99993f, 0.99991f}, k2 = {7.0e-5f, 9.0e-5f};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(k1), "v"(k2));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(k1));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(k2));
                if (mode == 7) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
            x = p.x; y = p.y;
        } else if (mode >= 8 && mode <= 12) { // ONE kind of instruction between bf16 MFMAs: 8 v_pk_max_i16, 9 v_cvt_pk_bf16_f32, 10 v_mov_b64,
            typedef __attribute__((ext_vector_type(2))) float f2;     // 11 v_pk_mul_f32 alone, 12 v_fma_f64 (64-bit datapath, not packed)
            f2 p = {x, y}, k1 = {0.99993f, 0.99991f};
            double dd = x;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (mode == 8) { unsigned q; asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(q) : "v"(u), "v"(v)); u = q + 1u; }
                else if (mode == 9) { unsigned q; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(q) : "v"(x), "v"(y)); x += __uint_as_float(q << 16) * 1e-6f; }
                else if (mode == 10) { f2 q; asm volatile("v_mov_b64 %0, %1" : "=v"(q) : "v"(p)); p = q; p.x += 1e-6f; }
                else if (mode == 11) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(k1)); }
                else { asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(dd) : "v"(dd), "v"(0.99993), "v"(1e-5)); }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
            x = p.x + (float)dd * 1e-9f; y = p.y;
        } else if (mode == 5) {               // transcendental unit: v_rcp_f32 / v_rcp_iflag_f32 / v_exp_f32 / v_sqrt_f32 / v_rsq_f32
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float p, q, r, s_;
                asm volatile("v_rcp_f32 %0, %1" : "=v"(p) : "v"(x));
                asm volatile("v_rcp_iflag_f32 %0, %1" : "=v"(q) : "v"(y));
                asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(p));
                asm volatile("v_sqrt_f32 %0, %1" : "=v"(s_) : "v"(q));
                x = x + r * 1e-7f + 1e-6f; y = y + s_ * 1e-7f + 1e-6f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                unsigned p, q;
                asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(p) : "v"(u), "v"(v), "v"(u + 7u));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(q) : "v"(p), "v"(v), "v"(0x07060302u));
                u = q + 1u; v ^= p;
            }
        }
    }
    out[t] = x + y + (float)(u ^ v) + acc[0];
}
extern "C" int noise_run(int mode, int blocks, int iters) {
    static hipStream_t st = nullptr;
    static float* src = nullptr; static float* out = nullptr;
    if (!st) {
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 1;
        if (hipMalloc(&src, (size_t)(64u << 20) + 4096) != hipSuccess || hipMalloc(&out, (size_t)65536 * 256 * 4) != hipSuccess) return 2;
        (void)hipMemset(src, 0, (size_t)(64u << 20) + 4096);
    }
    hipLaunchKernelGGL(noise_kernel, dim3(blocks), dim3(256), 16384, st, mode, iters, src, out);
    return hipStreamSynchronize(st) == hipSuccess ? 0 : 3;
}
