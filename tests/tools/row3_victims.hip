// WHICH instruction class of a victim wave is miscomputed in its last 16-lane row while a convolution kernel of
// the library shares the compute unit?  Micro-victims, one instruction class each: every thread computes the same chain TWICE (the
// compiler is kept from merging the two copies by opaque register barriers) and logs a difference with HW_REG_HW_ID / HW_REG_XCC_ID.
// Built by tests/tools/row3_probe.py (hipcc -O3 --offload-arch=gfx950 -shared); not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>

struct Rec { uint32_t hw_id, xcc_id, mode, lane, block, iter; float first, second; };
__device__ unsigned int g_count;
__device__ unsigned long long g_waves;
__device__ Rec g_recs[4096];

__device__ __forceinline__ void opaque(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void report(int mode, int it, float a, float b) {
    if (__float_as_uint(a) == __float_as_uint(b)) return;
    const unsigned int k = atomicAdd(&g_count, 1u);
    if (k < 4096u) {
        Rec r;
        r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4); r.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        r.mode = mode; r.lane = threadIdx.x & 63; r.block = blockIdx.x; r.iter = it; r.first = a; r.second = b;
        g_recs[k] = r;
    }
}

// the chains: x is a per-lane seed in (1, 2)
__device__ __forceinline__ float chain_fma(float x) {
    float y = x;
#pragma unroll 16
    for (int i = 0; i < 256; ++i) y = fmaf(y, 0.99993f, x * 7.0e-5f);
    return y;
}
__device__ __forceinline__ float chain_div(float x) {          // IEEE fp32 division: v_div_scale / v_rcp / v_div_fmas / v_div_fixup + denormal-mode toggles
    float y = x;
#pragma unroll 4
    for (int i = 0; i < 64; ++i) y = (y + 3.0f) / (x + 1.0f + 1e-3f * y);
    return y;
}
__device__ __forceinline__ float chain_trans(float x) {        // the transcendental unit: exp, sqrt, rcp, rsq, log
    float y = x;
#pragma unroll 4
    for (int i = 0; i < 64; ++i) y = __expf(-y) + sqrtf(y + 1.0f) + __frcp_rn(y + 2.0f) + __logf(y + 1.5f) * 0.1f;
    return y;
}
__device__ __forceinline__ float chain_int(float x) {          // integer multiplies / shifts / bit ops (quarter-rate v_mul_lo_u32)
    uint32_t u = __float_as_uint(x);
#pragma unroll 16
    for (int i = 0; i < 256; ++i) u = u * 1664525u + (u >> 7) + 1013904223u;
    return __uint_as_float((u >> 9) | 0x3F800000u);
}
__device__ __forceinline__ float chain_f64(float x) {
    double y = x;
#pragma unroll 8
    for (int i = 0; i < 96; ++i) y = fma(y, 0.99993, (double)x * 7.0e-5);
    return (float)y;
}
__device__ __forceinline__ float chain_xlane(float x) {        // cross-lane: DPP / ds_bpermute / readlane paths
    float y = x;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
        y += __shfl_xor(y, 1) * 1e-3f;
        y += __shfl_xor(y, 16) * 1e-3f;
        y += __shfl(y, (threadIdx.x + 17) & 63) * 1e-3f;
    }
    return y;
}
__device__ __noinline__ float call_fma(float x) { return chain_fma(x); }                       // the same chain behind a function call
__device__ __noinline__ void call_fma_scratch(float x, float* __restrict__ out8) {             // ... writing its results through private memory
    float y = x;
    for (int k = 0; k < 8; ++k) { y = chain_fma(y); out8[k] = y; }
}

// mode 17: the arithmetic in which the library saw the fault -- the posterior's prior fusion (csrc/post_kernels.hip: two Cholesky inverses
// of a 4x4 SPD matrix and two matrix-vector products), plain C++ that the SLP vectoriser packs into v_pk_mul_f32 / v_pk_add_f32 with
// op_sel / neg modifiers.  This file is compiled with the vectoriser ON (the library is not, since round 6).
struct M4 { float m[4][4]; };
__device__ __forceinline__ M4 inv_spd4_v(const M4& a) {
    float g[4][4], h[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[i][j] = 0.f; h[i][j] = 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float d = a.m[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= g[j][k] * g[j][k];
        const float dj = sqrtf(d);
        g[j][j] = dj;
        const float inv = 1.0f / dj;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            float s = a.m[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= g[i][k] * g[j][k];
            g[i][j] = s * inv;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j][j] = 1.0f / g[j][j];
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            float s = 0.f;
#pragma unroll
            for (int k = j; k < i; ++k) s += g[i][k] * h[k][j];
            h[i][j] = -s / g[i][i];
        }
    }
    M4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = i; k < 4; ++k) s += h[k][i] * h[k][j];
            r.m[i][j] = s; r.m[j][i] = s;
        }
    return r;
}
__device__ __noinline__ float prior_fusion(float x, float pp) {
    M4 lik;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) lik.m[i][j] = i == j ? 2.0f + x * (float)(i + 1) : 0.05f * x * (float)(i + j + 1);
    const float mu[4] = {100.f * x, 120.f * x, 40.f + x, 30.f + x}, am[4] = {101.f * x, 119.f * x, 41.f, 29.f};
    const M4 prec = inv_spd4_v(lik);
    M4 post = prec;
#pragma unroll
    for (int i = 0; i < 4; ++i) post.m[i][i] += pp;
    const M4 pcov = inv_spd4_v(post);
    float inter[4], pm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += prec.m[i][k] * mu[k];
        inter[i] = pp * am[i] + s;
    }
    float out = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += pcov.m[i][k] * inter[k];
        pm[i] = s;
        out += pm[i] * (float)(i + 1) + pcov.m[i][i] + pcov.m[i][(i + 1) & 3];
    }
    return out;
}

template <int MODE>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ seeds, float* __restrict__ sink, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float x = seeds[t & 65535];
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_waves, (unsigned long long)iters);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float xa = x, xb = x;
        opaque(xa); opaque(xb);
        float a, b;
        if constexpr (MODE == 0) { a = chain_fma(xa); b = chain_fma(xb); }
        else if constexpr (MODE == 1) { a = chain_div(xa); b = chain_div(xb); }
        else if constexpr (MODE == 2) { a = chain_trans(xa); b = chain_trans(xb); }
        else if constexpr (MODE == 3) { a = chain_int(xa); b = chain_int(xb); }
        else if constexpr (MODE == 4) { a = chain_f64(xa); b = chain_f64(xb); }
        else if constexpr (MODE == 5) { a = chain_xlane(xa); b = chain_xlane(xb); }
        else if constexpr (MODE == 6) { a = call_fma(xa); b = call_fma(xb); }
        else if constexpr (MODE == 7) {
            float r1[8], r2[8];
            call_fma_scratch(xa, r1); call_fma_scratch(xb, r2);
            a = r1[7] + r1[3]; b = r2[7] + r2[3];
        } else if constexpr (MODE == 8) {          // memory only -- the same gathered global reads twice (volatile: two loads)
            const volatile float* s = seeds;
            a = 0.f; b = 0.f;
            for (int k = 0; k < 16; ++k) { const int i = (t * 7 + k * 4099 + it * 13) & 65535; a += s[i]; }
            for (int k = 0; k < 16; ++k) { const int i = (t * 7 + k * 4099 + it * 13) & 65535; b += s[i]; }
        } else if constexpr (MODE == 9 || MODE == 10) {   // 16-byte (9) / 8-byte (10) loads per lane, twice: global_load_dwordx4 / dwordx2
            a = 0.f; b = 0.f;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                float acc2 = 0.f;
#pragma unroll 4
                for (int k = 0; k < 16; ++k) {
                    const int i = ((t * 5 + k * 4099 + it * 13) & 16383) * 4;
                    if constexpr (MODE == 9) {
                        float4 v;
                        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(seeds + i) : "memory");
                        acc2 += v.x + 2.f * v.y + 3.f * v.z + 5.f * v.w;
                    } else {
                        float2 v;
                        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(seeds + i) : "memory");
                        acc2 += v.x + 2.f * v.y;
                    }
                }
                if (pass == 0) a = acc2; else b = acc2;
            }
        } else if constexpr (MODE == 12 || MODE == 13) {  // packed fp32 (12: v_pk_mul_f32 + v_pk_add_f32; 13: v_pk_fma_f32): 64-bit datapath ops
            typedef __attribute__((ext_vector_type(2))) float f2;
            a = 0.f; b = 0.f;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                f2 y = {x, x * 1.5f}, k1 = {0.99993f, 0.99991f}, k2 = {x * 7.0e-5f, x * 9.0e-5f};
                asm volatile("" : "+v"(y));
#pragma unroll 16
                for (int i = 0; i < 128; ++i) {
                    if constexpr (MODE == 12) {
                        f2 t;
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(y), "v"(k1));
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(t), "v"(k2));
                    } else {
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(y), "v"(k1), "v"(k2));
                    }
                }
                if (pass == 0) a = y.x + y.y; else b = y.x + y.y;
            }
        } else if constexpr (MODE >= 20 && MODE <= 23) {   // which op_sel form: 20 src1 high half to BOTH lanes; 21 src0 halves swapped; 22 both swapped; 23 v_pk_add_f32 with src1 swapped
            typedef __attribute__((ext_vector_type(2))) float f2;
            a = 0.f; b = 0.f;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                f2 y = {x, x * 1.5f}, k1 = {0.99993f, 0.99991f}, k2 = {x * 7.0e-5f, x * 9.0e-5f};
                asm volatile("" : "+v"(y));
#pragma unroll 16
                for (int i = 0; i < 128; ++i) {
                    f2 t;
                    if constexpr (MODE == 20) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(y), "v"(k1)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(t), "v"(k2)); }
                    else if constexpr (MODE == 21) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(y), "v"(k1)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(t), "v"(k2)); }
                    else if constexpr (MODE == 22) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(t) : "v"(y), "v"(k1)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(t), "v"(k2)); }
                    else { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(y), "v"(k1)); asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(y) : "v"(t), "v"(k2)); }
                }
                if (pass == 0) a = y.x + y.y; else b = y.x + y.y;
            }
        } else if constexpr (MODE == 18 || MODE == 19) {   // packed fp32 with op_sel (a result half fed from the OTHER half of a source pair), as in mode 17's code
            typedef __attribute__((ext_vector_type(2))) float f2;
            a = 0.f; b = 0.f;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                f2 y = {x, x * 1.5f}, k1 = {0.99993f, 0.99991f}, k2 = {x * 7.0e-5f, x * 9.0e-5f};
                asm volatile("" : "+v"(y));
#pragma unroll 16
                for (int i = 0; i < 128; ++i) {
                    f2 t;
                    if constexpr (MODE == 18) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(y), "v"(k1));
                    else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(y), "v"(k1));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(t), "v"(k2));
                }
                if (pass == 0) a = y.x + y.y; else b = y.x + y.y;
            }
        } else if constexpr (MODE == 17) {
            a = prior_fusion(xa, 1e-5f); b = prior_fusion(xb, 1e-5f);
        } else if constexpr (MODE >= 14 && MODE <= 16) {  // packed fp32 WITH operand modifiers, as the SLP vectoriser emits them for 4x4 inverses:
            // 14: op_sel_hi:[1,0] (the low half of src1 broadcast to both halves); 15: neg_lo / neg_hi (a packed subtraction);
            // 16: both, plus a scalar consumer of each half right behind the packed instruction (the dot-product pattern)
            typedef __attribute__((ext_vector_type(2))) float f2;
            a = 0.f; b = 0.f;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                f2 y = {x, x * 1.5f}, k1 = {0.99993f, 0.99991f}, k2 = {x * 7.0e-5f, x * 9.0e-5f};
                asm volatile("" : "+v"(y));
                float s_ = 0.f;
#pragma unroll 16
                for (int i = 0; i < 128; ++i) {
                    f2 t;
                    if constexpr (MODE == 14) {
                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(y), "v"(k1));
                        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(t), "v"(k2));
                    } else if constexpr (MODE == 15) {
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(y), "v"(k1));
                        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(y) : "v"(t), "v"(k2));
                    } else {
                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(y), "v"(k1));
                        s_ += t.x * 1e-3f; s_ -= t.y * 1e-3f;
                        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(y) : "v"(t), "v"(k2));
                    }
                }
                if (pass == 0) a = y.x + y.y + s_; else b = y.x + y.y + s_;
            }
        } else {                                   // 11: private memory round trip (scratch_store / scratch_load of 16-byte pieces), twice
            a = 0.f; b = 0.f;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                float arr[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) arr[k] = x * (float)(k + 1) + (float)it;
                float acc2 = 0.f;
#pragma unroll 1
                for (int k = 0; k < 32; ++k) acc2 += arr[(k * 7 + (t & 31) + it) & 31] * (float)(k + 1);        // (dynamic index: the array lives in scratch)
                if (pass == 0) a = acc2; else b = acc2;
            }
        }
        report(MODE, it, a, b);
        acc += a;
        x = 1.0f + (a - floorf(a)) * 0.5f + 1e-3f * (float)(it & 7);
    }
    if (acc == 12345.678f) sink[t & 65535] = acc;          // (keeps the chains alive)
}

static float* g_seeds = nullptr; static float* g_sink = nullptr; static hipStream_t g_stream = nullptr;
extern "C" int victim_run(int mode, int blocks, int iters, int cu_lo, int cu_hi) {
    if (!g_seeds) {
        if (hipMalloc(&g_seeds, 65536 * 4) != hipSuccess || hipMalloc(&g_sink, 65536 * 4) != hipSuccess) return 1;
        static float h[65536];
        uint32_t u = 12345u;
        for (int i = 0; i < 65536; ++i) { u = u * 1664525u + 1013904223u; h[i] = 1.0f + (float)(u >> 8) / 16777216.0f; }
        if (hipMemcpy(g_seeds, h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    if (!g_stream) {
        if (cu_hi > cu_lo) {            // CU slots [lo, hi) of every XCD (mask bit i = slot i / 8 of XCD i % 8: tests/tools/cu_mask_probe.hip)
            uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int s = cu_lo; s < cu_hi; ++s) for (int x = 0; x < 8; ++x) { const int bit = s * 8 + x; m[bit >> 5] |= 1u << (bit & 31); }
            if (hipExtStreamCreateWithCUMask(&g_stream, 8, m) != hipSuccess) return 2;
        } else if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return 2;
    }
#define LAUNCH(M) case M: hipLaunchKernelGGL(victim_kernel<M>, dim3(blocks), dim3(256), 0, g_stream, g_seeds, g_sink, iters); break;
    switch (mode) { LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8) LAUNCH(9) LAUNCH(10) LAUNCH(11) LAUNCH(12) LAUNCH(13) LAUNCH(14) LAUNCH(15) LAUNCH(16) LAUNCH(17) LAUNCH(18) LAUNCH(19) LAUNCH(20) LAUNCH(21) LAUNCH(22) LAUNCH(23) default: return 3; }
    if (hipGetLastError() != hipSuccess) return 4;
    return hipStreamSynchronize(g_stream) == hipSuccess ? 0 : 5;
}
// ---- a synthetic COMPANY: workgroups that do nothing but LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction) and read the
// landed bytes back, for `us` microseconds; mode 1: the same bytes through plain global_load_dwordx4 + ds_write (no LDS-DMA)
__global__ __launch_bounds__(256) void dma_company_kernel(const float* __restrict__ src, float* __restrict__ sink, int mode, long long ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, wave = tid >> 6;
    float acc = 0.f;
    unsigned int it = blockIdx.x * 977u;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        for (int k = 0; k < 8; ++k) {
            const float* g = src + (((it + k) * 256u + tid) & 16383u) * 4;
            if (mode == 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                             (__attribute__((address_space(3))) void*)(smem + (k * 4 + wave) * 1024), 16, 0, 0);
            else *reinterpret_cast<float4*>(smem + (k * 256 + tid) * 16) = *reinterpret_cast<const float4*>(g);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<const float*>(smem + ((tid * 52 + it) & 32764));
        __syncthreads();
        ++it;
    }
    if (acc == 12345.678f) sink[tid] = acc;
}
static hipStream_t g_cstream = nullptr;
extern "C" int company_run(int mode, int blocks, int microseconds) {
    if (!g_seeds) return 1;
    if (!g_cstream && hipStreamCreateWithFlags(&g_cstream, hipStreamNonBlocking) != hipSuccess) return 2;
    hipLaunchKernelGGL(dma_company_kernel, dim3(blocks), dim3(256), 32768, g_cstream, g_seeds, g_sink, mode, (long long)microseconds * 100);
    if (hipGetLastError() != hipSuccess) return 4;
    return hipStreamSynchronize(g_cstream) == hipSuccess ? 0 : 5;
}
extern "C" int victim_read(unsigned int* count, unsigned long long* waves, void* recs, int max) {
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_count), 4) != hipSuccess || hipMemcpyFromSymbol(waves, HIP_SYMBOL(g_waves), 8) != hipSuccess) return 1;
    const unsigned int m = n < (unsigned)max ? (n < 4096u ? n : 4096u) : (unsigned)max;
    if (m && hipMemcpyFromSymbol(recs, HIP_SYMBOL(g_recs), (size_t)m * sizeof(Rec)) != hipSuccess) return 1;
    const unsigned int z = 0; const unsigned long long z8 = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_count), &z, 4) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_waves), &z8, 8) != hipSuccess) return 1;
    *count = n;
    return 0;
}
