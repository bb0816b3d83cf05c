// Stem of the ResNet-50 backbone: 7x7 s2 VALID conv + folded BN + ReLU, then
// ZeroPadding2D((1,2)) + MaxPool 3x3 s2 VALID   (src/retina_net/models/feature_extractor.py:17-33,
// :107-111).  The 3-channel input stays fp32 (no bf16 rounding
// of pixels); everything after is bf16 storage.
#include "kernels.h"

__device__ __forceinline__ uint32_t f32_to_bf16_a(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_to_f32_a(uint32_t v) { return __uint_as_float(v << 16); }

// Block: 4 output rows x 32 output cols x 64 channels. thread: co = tid&63, row = tid>>6.
// LDS: weights [7][7][3][64] fp32 (37.6 KB) + input patch [13][69*3] fp32 (10.8 KB).
constexpr int ST_TR = 4, ST_TC = 32, ST_PR = ST_TR * 2 + 5, ST_PC = ST_TC * 2 + 5;

__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ img,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        uint16_t* __restrict__ out,
                                                        int H, int W, int oh, int ow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sw = reinterpret_cast<float*>(smem);               // 7*7*3*64
    float* sp = sw + 7 * 7 * 3 * 64;                          // ST_PR * ST_PC * 3
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * ST_TR, ox0 = blockIdx.x * ST_TC;
    for (int i = tid; i < 7 * 7 * 3 * 64; i += 256) sw[i] = w[i];
    const float* im = img + (size_t)b * H * W * 3;
    const int iy0 = oy0 * 2, ix0 = ox0 * 2;
    for (int i = tid; i < ST_PR * ST_PC * 3; i += 256) {
        const int r = i / (ST_PC * 3), c = i % (ST_PC * 3);
        const int iy = iy0 + r, ixc = ix0 * 3 + c;
        sp[i] = (iy < H && ixc < W * 3) ? im[(size_t)iy * W * 3 + ixc] : 0.f;
    }
    __syncthreads();
    const int co = tid & 63, row = tid >> 6;
    const int oy = oy0 + row;
    const float bv = bias[co];
    for (int seg = 0; seg < ST_TC / 8; ++seg) {
        float acc[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[p] = 0.f;
        for (int ky = 0; ky < 7; ++ky) {
            const float* prow = sp + (row * 2 + ky) * (ST_PC * 3) + seg * 16 * 3;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                float in[21];
#pragma unroll
                for (int t = 0; t < 21; ++t) in[t] = prow[t * 3 + ci];
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const float wv = sw[((ky * 7 + kx) * 3 + ci) * 64 + co];
#pragma unroll
                    for (int p = 0; p < 8; ++p) acc[p] = fmaf(in[2 * p + kx], wv, acc[p]);
                }
            }
        }
        if (oy < oh) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int ox = ox0 + seg * 8 + p;
                if (ox < ow)
                    out[(((size_t)b * oh + oy) * ow + ox) * 64 + co] =
                        (uint16_t)f32_to_bf16_a(fmaxf(acc[p] + bv, 0.f));
            }
        }
    }
}

hipError_t launch_stem_conv(const float* img, const float* w, const float* bias, uint16_t* out,
                            int B, int H, int W, int oh, int ow, hipStream_t s) {
    const size_t lds = (7 * 7 * 3 * 64 + ST_PR * ST_PC * 3) * sizeof(float);
    dim3 grid((ow + ST_TC - 1) / ST_TC, (oh + ST_TR - 1) / ST_TR, B);
    hipLaunchKernelGGL(stem_conv_kernel, grid, dim3(256), lds, s, img, w, bias, out, H, W, oh, ow);
    return hipGetLastError();
}

// thread = (output pixel, 8-channel group); 16-byte loads/stores. Pads are zeros: inputs are
// post-ReLU (>= 0) so a zero pad value is what ZeroPadding2D + max produces.
__global__ __launch_bounds__(256) void stem_pool_kernel(const uint16_t* __restrict__ in,
                                                        uint16_t* __restrict__ out, int B, int ih,
                                                        int iw, int oh, int ow, int out_pitch,
                                                        int out_plane) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int total = B * oh * ow * 8;
    if (gid >= total) return;
    const int cg = gid & 7;
    int p = gid >> 3;
    const int ox = p % ow; p /= ow;
    const int oy = p % oh;
    const int b = p / oh;
    float m[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q] = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        if (iy < 0 || iy >= ih) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox + kx - 2;
            if (ix < 0 || ix >= iw) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(in + (((size_t)b * ih + iy) * iw + ix) * 64 + cg * 8);
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                m[2 * q] = fmaxf(m[2 * q], bf16_to_f32_a(u[q] & 0xFFFFu));
                m[2 * q + 1] = fmaxf(m[2 * q + 1], bf16_to_f32_a(u[q] >> 16));
            }
        }
    }
    uint4 o;
    o.x = f32_to_bf16_a(m[0]) | (f32_to_bf16_a(m[1]) << 16);
    o.y = f32_to_bf16_a(m[2]) | (f32_to_bf16_a(m[3]) << 16);
    o.z = f32_to_bf16_a(m[4]) | (f32_to_bf16_a(m[5]) << 16);
    o.w = f32_to_bf16_a(m[6]) | (f32_to_bf16_a(m[7]) << 16);
    const size_t opix = (size_t)b * out_plane + (size_t)(oy + 1) * out_pitch + (ox + 1);
    *reinterpret_cast<uint4*>(out + opix * 64 + cg * 8) = o;
}

hipError_t launch_stem_pool(const uint16_t* in, uint16_t* out, int B, int ih, int iw, int oh, int ow,
                            int out_pitch, int out_plane, hipStream_t s) {
    const int total = B * oh * ow * 8;
    hipLaunchKernelGGL(stem_pool_kernel, dim3((total + 255) / 256), dim3(256), 0, s, in, out, B, ih,
                       iw, oh, ow, out_pitch, out_plane);
    return hipGetLastError();
}
