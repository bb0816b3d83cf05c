// Implicit-GEMM convolution on gfx950 MFMA (v_mfma_f32_32x32x16_bf16), the kernel behind every
// conv of the RetinaNet forward except the 3-channel stem:
//   ResNet-50 bottlenecks   src/retina_net/models/feature_extractor.py:195-213,283-309
//   FPN laterals / outputs  src/retina_net/models/feature_decoder.py:136-171
//   head towers + dropout   src/retina_net/models/multitask_headers.py:98-123,209-230,318-342
//
// Orientation: D[cout][pixel] = W[cout][k] * X[pixel][k].  Weights are the MFMA "A" operand and
// pixels the "B" operand, so a lane's accumulator registers hold 4 CONSECUTIVE output channels of
// one pixel: the NHWC epilogue store is 8 contiguous bytes per lane and one Philox4x32 call yields
// exactly the 4 dropout decisions the lane needs.
//
// Both operands are K-contiguous (OHWI weights, NHWC activations), staged global->LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4) into 128-byte rows.  The LDS image is lane-linear, so
// the bank-conflict swizzle is applied on the SOURCE chunk index and again on the fragment read
// (chunk ^= (row>>1)&7: conflict-free for ds_read_b128's 16-lane groups over 128-B rows).
// Zero padding is physical (padded planes), so the gather needs no bounds checks.
#include "kernels.h"
#include "philox.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);          // round to nearest even (finite inputs)
    return u >> 16;
}

template <int BC, int BP, int WC, int WP>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int BK = 64;                       // bf16 per K-tile row (128 B)
    constexpr int ROWB = BK * 2;
    constexpr int W_BYTES = BC * ROWB, X_BYTES = BP * ROWB, STAGE = W_BYTES + X_BYTES;
    constexpr int NW = BC * 8 / 256, NX = BP * 8 / 256;   // 16-B chunks per thread per tile
    constexpr int WTC = BC / WC, WTP = BP / WP;
    constexpr int FC = WTC / 32, FP = WTP / 32;
    static_assert(WC * WP == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const ConvGroup& G = a.g[blockIdx.z];
    const int bp0 = blockIdx.x * BP, bc0 = blockIdx.y * BC;
    const int cpt = a.cin / BK;                  // K-tiles per tap
    const int KT = a.taps * cpt;

    // ---- per-thread staging descriptors
    const int ldrow = tid >> 3;                              // 0..31 (+32*i)
    const int ldchunk = (tid & 7) ^ ((tid >> 4) & 7);        // source chunk (pre-swizzled)
    const char* xsrc[NX];
    int xpitch[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        int m = bp0 + i * 32 + ldrow;
        m = m < a.M ? m : a.M - 1;
        const int2 e = *reinterpret_cast<const int2*>(&a.rows[m]);
        xsrc[i] = reinterpret_cast<const char*>(G.in) +
                  ((size_t)e.x * a.in_cstride + G.in_coff + ldchunk * 8) * 2;
        xpitch[i] = e.y * a.in_cstride * 2;
    }
    const char* wsrc[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int co = bc0 + i * 32 + ldrow;
        wsrc[i] = reinterpret_cast<const char*>(G.w) + ((size_t)co * KT * BK + ldchunk * 8) * 2;
    }

    auto issue = [&](int stage, int kt, int ky, int kx, int cc) {
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wsrc[i] + (size_t)kt * ROWB),
                                             LDS_PTR(sb + (i * 256 + wave * 64) * 16), 16, 0, 0);
        const int tapoff = (kx * a.in_cstride + cc * BK) * 2;
#pragma unroll
        for (int i = 0; i < NX; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(xsrc[i] + ky * xpitch[i] + tapoff),
                                             LDS_PTR(sb + W_BYTES + (i * 256 + wave * 64) * 16), 16, 0, 0);
    };

    f32x16 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31;
    const int fswz = (frow >> 1) & 7;
    const int fhalf = lane >> 5;

    int ky = 0, kx = 0, cc = 0;
    issue(0, 0, 0, 0, 0);
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < KT) {
            if (++cc == cpt) { cc = 0; if (++kx == a.KW) { kx = 0; ++ky; } }
            issue(cur ^ 1, kt + 1, ky, kx, cc);
        }
        const char* wb = smem + cur * STAGE + (wc * WTC + frow) * ROWB;
        const char* xb = smem + cur * STAGE + W_BYTES + (wp * WTP + frow) * ROWB;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int ch = ((ks * 2 + fhalf) ^ fswz) << 4;
            bf16x8 af[FC], bfr[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) af[i] = *reinterpret_cast<const bf16x8*>(wb + i * 32 * ROWB + ch);
#pragma unroll
            for (int j = 0; j < FP; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(xb + j * 32 * ROWB + ch);
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        cur ^= 1;
    }

    // ---- epilogue: bias (+residual) (+ReLU) (+dropout) -> bf16 / fp32 store
    const bool relu = a.flags & CONV_RELU, drop = a.flags & CONV_DROPOUT, of32 = a.flags & CONV_OUT_F32;
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const int m = bp0 + wp * WTP + j * 32 + frow;
        if (m >= a.M) continue;
        const int4 e0 = *reinterpret_cast<const int4*>(&a.rows[m]);          // in_off, pitch, out_off, res_off
        const int2 e1 = *(reinterpret_cast<const int2*>(&a.rows[m]) + 2);    // rng_p, rng_zs
#pragma unroll
        for (int i = 0; i < FC; ++i) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int co = bc0 + wc * WTC + i * 32 + g4 * 8 + fhalf * 4;
                if (co >= a.cout_valid) continue;
                const float4 bv = *reinterpret_cast<const float4*>(G.bias + co);
                float v[4] = {acc[i][j][g4 * 4 + 0] + bv.x, acc[i][j][g4 * 4 + 1] + bv.y,
                              acc[i][j][g4 * 4 + 2] + bv.z, acc[i][j][g4 * 4 + 3] + bv.w};
                if (G.res) {
                    const uint2 r = *reinterpret_cast<const uint2*>(G.res + (size_t)e0.w * a.res_cstride + co);
                    v[0] += bf16_to_f32(r.x & 0xFFFFu); v[1] += bf16_to_f32(r.x >> 16);
                    v[2] += bf16_to_f32(r.y & 0xFFFFu); v[3] += bf16_to_f32(r.y >> 16);
                }
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                if (of32) {
                    float* o = reinterpret_cast<float*>(G.out) + (size_t)e0.z * a.out_cstride + co;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (co + q < a.cout_valid) o[q] = v[q];
                } else if (!drop) {
                    uint2 pk;
                    pk.x = f32_to_bf16(v[0]) | (f32_to_bf16(v[1]) << 16);
                    pk.y = f32_to_bf16(v[2]) | (f32_to_bf16(v[3]) << 16);
                    uint16_t* o = reinterpret_cast<uint16_t*>(G.out) + (size_t)e0.z * a.out_cstride + co;
                    *reinterpret_cast<uint2*>(o) = pk;
                    if (G.out_relu) {
                        uint2 pr;
                        pr.x = f32_to_bf16(fmaxf(v[0], 0.f)) | (f32_to_bf16(fmaxf(v[1], 0.f)) << 16);
                        pr.y = f32_to_bf16(fmaxf(v[2], 0.f)) | (f32_to_bf16(fmaxf(v[3], 0.f)) << 16);
                        *reinterpret_cast<uint2*>(G.out_relu + (size_t)e0.z * a.out_cstride + co) = pr;
                    }
                } else {
                    const uint32_t img = a.image_base + ((uint32_t)e1.y >> 16);
                    const int fan = a.fan_count > 1 ? a.fan_count : 1;
                    for (int n = 0; n < fan; ++n) {
                        const uint32_t sample = a.fan_count > 1 ? (uint32_t)n : ((uint32_t)e1.y & 0xFFFFu);
                        const Philox4 r = philox4x32_10((uint32_t)e1.x, (uint32_t)co >> 2,
                                                        sample | ((uint32_t)G.layer_id << 16), img,
                                                        a.seed_lo, a.seed_hi);
                        const float w0 = r.x >= a.drop_threshold ? v[0] * a.drop_scale : 0.f;
                        const float w1 = r.y >= a.drop_threshold ? v[1] * a.drop_scale : 0.f;
                        const float w2 = r.z >= a.drop_threshold ? v[2] * a.drop_scale : 0.f;
                        const float w3 = r.w >= a.drop_threshold ? v[3] * a.drop_scale : 0.f;
                        uint2 pk;
                        pk.x = f32_to_bf16(w0) | (f32_to_bf16(w1) << 16);
                        pk.y = f32_to_bf16(w2) | (f32_to_bf16(w3) << 16);
                        uint16_t* o = reinterpret_cast<uint16_t*>(G.out) +
                                      ((size_t)e0.z + (size_t)n * a.fan_stride) * a.out_cstride + co;
                        *reinterpret_cast<uint2*>(o) = pk;
                    }
                }
            }
        }
    }
}

template <int BC, int BP, int WC, int WP>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
    constexpr int STAGE = (BC + BP) * 128;
    static bool attr_set = false;
    auto kern = conv_igemm_kernel<BC, BP, WC, WP>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((a.M + BP - 1) / BP, a.cout_pad / BC, a.groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), 2 * STAGE, s, a);
    return hipGetLastError();
}

hipError_t launch_conv_igemm(const ConvArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if (a.cin % 64 != 0 || a.cout_pad % 64 != 0) return hipErrorInvalidValue;
    if (a.cout_pad % 128 == 0) return launch_cfg<128, 128, 2, 2>(a, s);
    return launch_cfg<64, 128, 1, 4>(a, s);
}
