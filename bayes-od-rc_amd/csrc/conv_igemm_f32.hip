// fp32 twin of conv_igemm.hip for the library's "fp32" precision mode: same implicit-GEMM structure,
// row tables, padded planes, LDS-DMA staging and swizzle, but fp32 activations / weights and the
// exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bitwise an fmaf
// chain, 1/16 of the bf16 MFMA rate).  This is the mode in which the whole forward pass agrees with
// the reference's fp32 arithmetic to ~1e-5 (tests/test_gpu_forward.py::test_fp32_mode_end_to_end);
// the bf16 kernel is the throughput path.
//
// A 128-byte LDS row holds 32 fp32 (BK = 32).  One ds_read_b128 per operand brings 4 consecutive k
// of a row; lane half h reads chunk 2*ks+h, and MFMA q of the 4 that follow pairs element q of both
// halves, i.e. k = {8ks+q, 8ks+4+q}: any pairing is valid as long as A and B use the same one.
#include "kernels.h"
#include "philox.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int BC, int BP, int WC, int WP>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs a) {
    constexpr int BK = 32;                       // fp32 per K-tile row (128 B)
    constexpr int ROWB = 128;
    constexpr int W_BYTES = BC * ROWB, X_BYTES = BP * ROWB, STAGE = W_BYTES + X_BYTES;
    constexpr int NW = BC * 8 / 256, NX = BP * 8 / 256;
    constexpr int WTC = BC / WC, WTP = BP / WP;
    constexpr int FC = WTC / 32, FP = WTP / 32;
    static_assert(WC * WP == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const ConvGroup& G = a.g[blockIdx.z];
    int bx = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, idx = bx >> 3;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bp0 = bx * BP, bc0 = blockIdx.y * BC;
    const int cpt = a.cin / BK;
    const int KT = a.taps * cpt;
    const int KH = a.taps / a.KW;

    const int ldrow = tid >> 3;
    const int ldchunk = (tid & 7) ^ ((tid >> 4) & 7);
    const char* xsrc[NX];
    int xpitch[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        int m = bp0 + i * 32 + ldrow;
        m = m < a.M ? m : a.M - 1;
        const int2 e = *reinterpret_cast<const int2*>(&a.rows[m]);
        xsrc[i] = reinterpret_cast<const char*>(G.in) + ((size_t)e.x * a.in_cstride + G.in_coff + ldchunk * 4) * 4;
        xpitch[i] = e.y * a.in_cstride * 4;
    }
    const char* wsrc[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int co = bc0 + i * 32 + ldrow;
        wsrc[i] = reinterpret_cast<const char*>(G.w) + ((size_t)co * a.taps * a.cin + ldchunk * 4) * 4;
    }
    auto issue = [&](int stage, int ky, int kx, int cc) {
        char* sb = smem + stage * STAGE;
        const int woff = ((ky * a.KW + kx) * a.cin + cc * BK) * 4;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wsrc[i] + woff), LDS_PTR(sb + (i * 256 + wave * 64) * 16), 16, 0, 0);
        const int tapoff = (kx * a.in_cstride + cc * BK) * 4;
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
    issue(0, 0, 0, 0);
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < KT) {
            if (++kx == a.KW) { kx = 0; if (++ky == KH) { ky = 0; ++cc; } }
            issue(cur ^ 1, ky, kx, cc);
        }
        const char* wb = smem + cur * STAGE + (wc * WTC + frow) * ROWB;
        const char* xb = smem + cur * STAGE + W_BYTES + (wp * WTP + frow) * ROWB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ((ks * 2 + fhalf) ^ fswz) << 4;
            float4 af[FC], bfr[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) af[i] = *reinterpret_cast<const float4*>(wb + i * 32 * ROWB + ch);
#pragma unroll
            for (int j = 0; j < FP; ++j) bfr[j] = *reinterpret_cast<const float4*>(xb + j * 32 * ROWB + ch);
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bfr[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bfr[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bfr[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bfr[j].w, acc[i][j], 0, 0, 0);
                }
        }
        cur ^= 1;
    }

    // ---- epilogue: bias (+residual) (+ReLU) (+dropout, optional N-way fan-out) -> fp32; CONV_ACCUM adds to what the output
    // already holds (the input-gradient GEMMs of the fp32 training handle)
    const bool relu = a.flags & CONV_RELU, drop = a.flags & CONV_DROPOUT, accum = a.flags & CONV_ACCUM;
    const float* res = reinterpret_cast<const float*>(G.res);
    float* out = reinterpret_cast<float*>(G.out);
    float* out_relu = reinterpret_cast<float*>(G.out_relu);
    const bool vec_ok = (a.out_cstride % 4 == 0) && (a.cout_valid % 4 == 0);
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const int m = bp0 + wp * WTP + j * 32 + frow;
        if (m >= a.M) continue;
        const int4 e0 = *reinterpret_cast<const int4*>(&a.rows[m]);
        const int2 e1 = *(reinterpret_cast<const int2*>(&a.rows[m]) + 2);
#pragma unroll
        for (int i = 0; i < FC; ++i) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int co = bc0 + wc * WTC + i * 32 + g4 * 8 + fhalf * 4;
                if (co >= a.cout_valid) continue;
                const float4 bv = *reinterpret_cast<const float4*>(G.bias + co);
                float v[4] = {acc[i][j][g4 * 4 + 0] + bv.x, acc[i][j][g4 * 4 + 1] + bv.y,
                              acc[i][j][g4 * 4 + 2] + bv.z, acc[i][j][g4 * 4 + 3] + bv.w};
                if (res) {
                    const float4 r = *reinterpret_cast<const float4*>(res + (size_t)e0.w * a.res_cstride + co);
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                }
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                const int fan = (drop && a.fan_count > 1) ? a.fan_count : 1;
                for (int n = 0; n < fan; ++n) {
                    float w[4] = {v[0], v[1], v[2], v[3]};
                    if (drop) {
                        const uint32_t img = a.image_base + ((uint32_t)e1.y >> 16);
                        const uint32_t sample = a.sample_base + (a.fan_count > 1 ? (uint32_t)n : ((uint32_t)e1.y & 0xFFFFu));
                        const Philox4 r = philox4x32_10((uint32_t)e1.x, dropout_group16((uint32_t)co),
                                                        sample | ((uint32_t)G.layer_id << 16), img, a.seed_lo, a.seed_hi);
                        const DropPair dw = dropout_run_windows(r, ((co >> 4) & 1) * 2 + ((co >> 3) & 1));     // contract v3 (philox.h)
                        const uint32_t w0 = dw.x, w1 = dw.y;
                        w[0] = (w0 & 0xFFFFu) >= a.drop_threshold ? v[0] * a.drop_scale : 0.f;
                        w[1] = (w0 >> 16) >= a.drop_threshold ? v[1] * a.drop_scale : 0.f;
                        w[2] = (w1 & 0xFFFFu) >= a.drop_threshold ? v[2] * a.drop_scale : 0.f;
                        w[3] = (w1 >> 16) >= a.drop_threshold ? v[3] * a.drop_scale : 0.f;
                    }
                    const size_t o = ((size_t)e0.z + (size_t)n * a.fan_stride) * a.out_cstride + co;
                    if (accum) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (co + q < a.cout_valid) w[q] += out[o + q];
                    }
                    if (vec_ok) {
                        *reinterpret_cast<float4*>(out + o) = make_float4(w[0], w[1], w[2], w[3]);
                        if (out_relu)
                            *reinterpret_cast<float4*>(out_relu + o) =
                                make_float4(fmaxf(w[0], 0.f), fmaxf(w[1], 0.f), fmaxf(w[2], 0.f), fmaxf(w[3], 0.f));
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (co + q < a.cout_valid) out[o + q] = w[q];
                    }
                }
            }
        }
    }
}

template <int BC, int BP, int WC, int WP>
static hipError_t launch_f32_cfg(const ConvArgs& a, hipStream_t s) {
    constexpr int LDS = 2 * (BC + BP) * 128;
    static PerDeviceOnce once;
    bool& attr_set = *once.slot();
    auto kern = conv_igemm_f32_kernel<BC, BP, WC, WP>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((a.M + BP - 1) / BP, a.cout_pad / BC, a.groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, a);
    return hipGetLastError();
}

hipError_t launch_conv_igemm_f32(const ConvArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if (a.cin % 32 != 0 || a.cout_pad % 64 != 0) return hipErrorInvalidValue;
    if (a.cout_pad % 128 == 0) return launch_f32_cfg<128, 128, 2, 2>(a, s);
    return launch_f32_cfg<64, 128, 1, 4>(a, s);
}
