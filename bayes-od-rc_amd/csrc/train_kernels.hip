// Building blocks of the training step.  This round: the weight gradient of one
// convolution as a pixel-reduction GEMM on the forward implicit-GEMM kernel,
//
//     dW[(tap, ci)][co] = sum_m  Xcol^T[(tap, ci)][m] * dY^T[co][m]        (m = output pixel)
//
// Both operands must be K(= m)-contiguous rows for the kernel's LDS-DMA staging, so the activations are
// gathered-and-transposed through the FORWARD row table (im2col^T) and dY is transposed; the reduction over
// pixels is split over workgroups with the kernel's split-K path.
#include "kernels.h"

// out[(tap*C + c)][m] = in[(rows[m].in_off + ky*rows[m].in_pitch + kx) * cstride + c]   (rows == nullptr: in_off = m)
// for m < M, zero for M <= m < Kpad.  64 x 64 (m x c) tiles through LDS: reads are contiguous in c, writes in m.
template <typename T> struct OneOf;
template <> struct OneOf<uint16_t> { static __device__ uint16_t v() { return (uint16_t)0x3F80; } };      // bf16 1.0
template <> struct OneOf<float> { static __device__ float v() { return 1.0f; } };
template <typename T>
__global__ __launch_bounds__(256) void gather_transpose_kernel(const T* in, const RowEnt* rows, T* out,
                                                               int M, int Kpad, int C, int cstride, int KW, int ones_row) {
    __shared__ T tile[64][65 + (sizeof(T) == 2)];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
    if (ones_row >= 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 64 && m0 + (int)threadIdx.x < Kpad)
        out[(size_t)ones_row * Kpad + m0 + threadIdx.x] = m0 + (int)threadIdx.x < M ? OneOf<T>::v() : (T)0;      // 1.0: the bias-gradient row
    const int ky = tap / KW, kx = tap - ky * KW;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;           // 4 rows of 64 threads
    for (int r = ty; r < 64; r += 4) {
        const int m = m0 + r, c = c0 + tx;
        T v = (T)0;
        if (m < M && c < C) {
            long pix = m;
            if (rows) { const RowEnt e = rows[m]; pix = (long)e.in_off + (long)ky * e.in_pitch + kx; }
            v = in[pix * cstride + c];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, m = m0 + tx;
        if (c < C && m < Kpad) out[((size_t)tap * C + c) * Kpad + m] = tile[tx][r];
    }
}

// Vector form for channel counts / pixel strides that are multiples of 8: 16-byte reads along the channels, 16-byte
// writes along the pixels.
__global__ __launch_bounds__(256) void gather_transpose_vec_kernel(const uint16_t* in, const RowEnt* rows, uint16_t* out,
                                                                   int M, int Kpad, int C, int cstride, int KW, int ones_row) {
    __shared__ uint16_t tile[64][72];                                  // [m][c], 144-byte rows keep the 16-byte row writes aligned
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
    if (ones_row >= 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 64 && m0 + (int)threadIdx.x < Kpad)
        out[(size_t)ones_row * Kpad + m0 + threadIdx.x] = m0 + (int)threadIdx.x < M ? (uint16_t)0x3F80 : (uint16_t)0;
    const int ky = tap / KW, kx = tap - ky * KW;
    const int sub = threadIdx.x & 7, row = threadIdx.x >> 3;            // 8 threads x 16 B per 64-channel row, 32 rows per pass
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = row + p * 32, m = m0 + r, c = c0 + sub * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (m < M && c < C) {
            long pix = m;
            if (rows) { const RowEnt e = rows[m]; pix = (long)e.in_off + (long)ky * e.in_pitch + kx; }
            v = *reinterpret_cast<const uint4*>(in + pix * cstride + c);
        }
        *reinterpret_cast<uint4*>(&tile[r][sub * 8]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int cr = row + p * 32, c = c0 + cr, m = m0 + sub * 8;     // this thread: channel c, pixels m .. m+7
        if (c < C && m < Kpad) {
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = (uint32_t)tile[sub * 8 + 2 * q][cr] | ((uint32_t)tile[sub * 8 + 2 * q + 1][cr] << 16);
            *reinterpret_cast<uint4*>(out + ((size_t)tap * C + c) * Kpad + m) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// Few channels per pixel (the stem's image, C = 3): tile over the joint row index n = tap * C + c instead of wasting
// 61 of 64 channel lanes.
__global__ __launch_bounds__(256) void gather_transpose_smallc_kernel(const uint16_t* in, const RowEnt* rows, uint16_t* out,
                                                                      int M, int Kpad, int C, int cstride, int KW, int NR, int ones_row) {
    __shared__ uint16_t tile[64][66];                                  // [n][m]
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    if (ones_row >= 0 && blockIdx.y == 0 && threadIdx.x < 64 && m0 + (int)threadIdx.x < Kpad)
        out[(size_t)ones_row * Kpad + m0 + threadIdx.x] = m0 + (int)threadIdx.x < M ? (uint16_t)0x3F80 : (uint16_t)0;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int m = m0 + tx;
    RowEnt e{};
    if (m < M && rows) e = rows[m];
    for (int r = ty; r < 64; r += 4) {
        const int n = n0 + r;
        uint16_t v = 0;
        if (m < M && n < NR) {
            const int tap = n / C, c = n - tap * C;
            const int ky = tap / KW, kx = tap - ky * KW;
            const long pix = rows ? (long)e.in_off + (long)ky * e.in_pitch + kx : m;
            v = in[pix * cstride + c];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int n = n0 + r;
        if (n < NR && m < Kpad) out[(size_t)n * Kpad + m] = tile[r][tx];
    }
}

__global__ void fill_row_bf16_kernel(uint16_t* row, int n_set, int n_total, uint16_t value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_total) row[i] = i < n_set ? value : (uint16_t)0;
}

hipError_t launch_gather_transpose(const void* in, const RowEnt* rows, void* out, int M, int Kpad, int C, int cstride,
                                   int taps, int KW, bool append_ones_row, hipStream_t s, bool f32) {
    const int ones_row = append_ones_row ? taps * C : -1;
    dim3 grid((Kpad + 63) / 64, (C + 63) / 64, taps);
    if (f32) {                                    // fp32 training handle (gradient verification mode): the generic form only
        hipLaunchKernelGGL(gather_transpose_kernel<float>, grid, dim3(256), 0, s, reinterpret_cast<const float*>(in), rows,
                           reinterpret_cast<float*>(out), M, Kpad, C, cstride, KW, ones_row);
        return hipGetLastError();
    }
    if (C % 8 == 0 && cstride % 8 == 0 && Kpad % 8 == 0) {
        hipLaunchKernelGGL(gather_transpose_vec_kernel, grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(in), rows,
                           reinterpret_cast<uint16_t*>(out), M, Kpad, C, cstride, KW, ones_row);
        return hipGetLastError();
    }
    if (C < 16) {
        const int NR = taps * C;
        hipLaunchKernelGGL(gather_transpose_smallc_kernel, dim3((Kpad + 63) / 64, (NR + 63) / 64), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(in), rows,
                           reinterpret_cast<uint16_t*>(out), M, Kpad, C, cstride, KW, NR, ones_row);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gather_transpose_kernel<uint16_t>, grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(in), rows,
                       reinterpret_cast<uint16_t*>(out), M, Kpad, C, cstride, KW, ones_row);
    return hipGetLastError();
}

hipError_t launch_fill_row_bf16(void* row, int n_set, int n_total, float value, hipStream_t s) {
    uint32_t u; __builtin_memcpy(&u, &value, 4);
    hipLaunchKernelGGL(fill_row_bf16_kernel, dim3((n_total + 255) / 256), dim3(256), 0, s, reinterpret_cast<uint16_t*>(row), n_set, n_total,
                       (uint16_t)(u >> 16));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The rest of the backward pass: everything except the two GEMMs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f_dev(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf_dev(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <typename T> struct El;
template <> struct El<uint16_t> { static __device__ __forceinline__ float ld(uint16_t v) { return bf2f_dev(v); } static __device__ __forceinline__ uint16_t st(float f) { return f2bf_dev(f); } };
template <> struct El<float> { static __device__ __forceinline__ float ld(float v) { return v; } static __device__ __forceinline__ float st(float f) { return f; } };

// Per step: fp32 master weights (HWIO) + BatchNorm parameters -> the forward packing [cout_pad][taps][cin] (bf16), the
// folded bias (fp32) and the input-gradient packing [(tap, ci)][cout_pad] (bf16, no flip: it is used as a plain GEMM).
// one folded weight into the layer's packings (i = its index in the forward layout [cout_pad][taps][cin]); fp32 packings for
// the fp32 training handle, bf16 otherwise
__device__ __forceinline__ void fold_store(const FoldArgs& a, long i, int t, int ci, int co, float v) {
    const size_t ib = ((size_t)t * a.cin + ci) * a.cout_pad + co, ifl = ((size_t)ci * a.taps + (a.taps - 1 - t)) * a.cout_pad + co;
    if (a.w_fwd32 && co < a.cout) a.w_fwd32[((size_t)t * a.cin + ci) * a.cout + co] = v;      // stem: fp32 [tap*cin][cout]
    if (a.f32) {
        if (a.w_fwd) reinterpret_cast<float*>(a.w_fwd)[i] = v;
        if (a.w_bwd) reinterpret_cast<float*>(a.w_bwd)[ib] = v;
        if (a.w_flip) reinterpret_cast<float*>(a.w_flip)[ifl] = v;
    } else {
        if (a.w_fwd) a.w_fwd[i] = f2bf_dev(v);
        if (a.w_bwd) a.w_bwd[ib] = f2bf_dev(v);
        if (a.w_flip) a.w_flip[ifl] = f2bf_dev(v);
    }
}

__global__ __launch_bounds__(256) void fold_pack_kernel(FoldArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)a.cout_pad * a.taps * a.cin;
    if (i < n) {
        const int ci = (int)(i % a.cin);
        const int t = (int)((i / a.cin) % a.taps);
        const int co = (int)(i / ((long)a.cin * a.taps));
        float v = 0.f;
        if (co < a.cout) {
            const float s = a.gamma ? a.gamma[co] / sqrtf(a.var[co] + a.eps) : 1.0f;
            v = a.kernel[((size_t)t * a.cin + ci) * a.cout + co] * s;
        }
        fold_store(a, i, t, ci, co, v);
    }
    if (i < a.cout_pad) {
        float b = 0.f;
        if (i < a.cout) {
            b = a.bias ? a.bias[i] : 0.f;
            if (a.gamma) { const float s = a.gamma[i] / sqrtf(a.var[i] + a.eps); b = (b - a.mean[i]) * s + a.beta[i]; }
        }
        a.b_fwd[i] = b;
    }
}

// dZ[m][co] (dense bf16, cout_pad columns) = dOut[out_off(m)][co] * (mask from the stored output); the same value is
// what the residual input of the layer receives (out = act(conv + res)), scattered with atomics because a nearest-
// upsampled residual (FPN) is read by several output pixels.
template <typename T>
__global__ __launch_bounds__(256) void act_backward_gather_kernel(ActBwdArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.M * a.cout_pad) return;
    const int m = (int)(i / a.cout_pad), co = (int)(i % a.cout_pad);
    const T* out_act = reinterpret_cast<const T*>(a.out_bf16);
    T* dz = reinterpret_cast<T*>(a.dz);
    T* dzp = reinterpret_cast<T*>(a.dzp);
    float g = 0.f;
    if (co < a.cout) {
        const RowEnt e = a.rows[m];
        const size_t o = (size_t)e.out_off * a.out_cstride + co;
        g = a.dout[o];
        if (out_act) { if (El<T>::ld(out_act[o]) == 0.f) g = 0.f; else g *= a.scale; }
        if (a.dres && g != 0.f) atomicAdd(a.dres + (size_t)e.res_off * a.res_cstride + co, g);
        if (dzp) dzp[o] = El<T>::st(g);
    }
    dz[i] = El<T>::st(g);
}

// 8 channels per thread (cout, cout_pad and both pixel strides multiples of 8)
__global__ __launch_bounds__(256) void act_backward_gather_vec_kernel(ActBwdArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = a.cout_pad / 8;
    if (i >= (long)a.M * c8) return;
    const int m = (int)(i / c8), co = (int)(i % c8) * 8;
    uint4 packed = make_uint4(0u, 0u, 0u, 0u);
    if (co < a.cout) {
        const RowEnt e = a.rows[m];
        const size_t o = (size_t)e.out_off * a.out_cstride + co;
        const float4 g0 = *reinterpret_cast<const float4*>(a.dout + o), g1 = *reinterpret_cast<const float4*>(a.dout + o + 4);
        float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        if (a.out_bf16) {
            const uint4 y = *reinterpret_cast<const uint4*>(a.out_bf16 + o);
            const uint32_t yw[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint16_t h = (uint16_t)(yw[k >> 1] >> ((k & 1) * 16));
                g[k] = (h & 0x7FFFu) == 0 ? 0.f : g[k] * a.scale;
            }
        }
        if (a.dres) {
            float* r = a.dres + (size_t)e.res_off * a.res_cstride + co;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (g[k] != 0.f) atomicAdd(r + k, g[k]);
        }
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (uint32_t)f2bf_dev(g[2 * k]) | ((uint32_t)f2bf_dev(g[2 * k + 1]) << 16);
        packed = make_uint4(w[0], w[1], w[2], w[3]);
        if (a.dzp) *reinterpret_cast<uint4*>(a.dzp + o) = packed;
    }
    *reinterpret_cast<uint4*>(a.dz + (size_t)m * a.cout_pad + co) = packed;
}

// 64 pixels x 64 channels per block: the vector form above plus, through an LDS tile, the transposed copy dZ^T that
// the weight-gradient GEMM reads (saves the separate transpose launch); residual gradients are staged through LDS so
// that a wavefront's atomics fall on consecutive channels of one pixel.
__global__ __launch_bounds__(256) void act_backward_tile_kernel(ActBwdArgs a) {
    __shared__ uint16_t tT[64][66];                                    // [channel][pixel]
    __shared__ float tR[32][65];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tm = tid >> 3, cl = (tid & 7) * 8, co = c0 + cl;
    for (int pass = 0; pass < 2; ++pass) {
        const int ml = pass * 32 + tm, m = m0 + ml;
        float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const bool live = m < a.M && co < a.cout;
        size_t o = 0;
        if (live) {
            const RowEnt e = a.rows[m];
            o = (size_t)e.out_off * a.out_cstride + co;
            const float4 g0 = *reinterpret_cast<const float4*>(a.dout + o), g1 = *reinterpret_cast<const float4*>(a.dout + o + 4);
            g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
            if (a.out_bf16) {
                const uint4 y = *reinterpret_cast<const uint4*>(a.out_bf16 + o);
                const uint32_t yw[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint16_t hv = (uint16_t)(yw[k >> 1] >> ((k & 1) * 16));
                    g[k] = (hv & 0x7FFFu) == 0 ? 0.f : g[k] * a.scale;
                }
            }
        }
        uint16_t hb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { hb[k] = f2bf_dev(g[k]); tT[cl + k][ml] = hb[k]; }
        if (m < a.M) {
            const uint4 packed = make_uint4((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16),
                                            (uint32_t)hb[4] | ((uint32_t)hb[5] << 16), (uint32_t)hb[6] | ((uint32_t)hb[7] << 16));
            *reinterpret_cast<uint4*>(a.dz + (size_t)m * a.cout_pad + co) = packed;
            if (live && a.dzp) *reinterpret_cast<uint4*>(a.dzp + o) = packed;
        }
        if (a.dres) {                                                  // (uniform across the block)
#pragma unroll
            for (int k = 0; k < 8; ++k) tR[tm][cl + k] = g[k];
            __syncthreads();
            const int cc = tid & 63;
            for (int j = tid >> 6; j < 32; j += 4) {
                const int mm = m0 + pass * 32 + j;
                const float v = tR[j][cc];
                if (mm < a.M && v != 0.f) atomicAdd(a.dres + (size_t)a.rows[mm].res_off * a.res_cstride + c0 + cc, v);
            }
            __syncthreads();
        }
    }
    __syncthreads();
    const int row = tid >> 2, mq = (tid & 3) * 16;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&tT[row][mq]);
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = src[k];
    uint4* dst = reinterpret_cast<uint4*>(a.dzt + (size_t)(c0 + row) * a.Kpad + m0 + mq);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// second consumer of a layer's output through its ReLU'd copy (P6 -> relu -> P7): dOut += dOutRelu * [out > 0]
template <typename T>
__global__ __launch_bounds__(256) void relu_merge_kernel(const float* dout_relu, const T* out, float* dout, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && El<T>::ld(out[i]) > 0.f) dout[i] += dout_relu[i];
}

// col2im: dIn[(in_off(m) + ky*pitch + kx)][ci] += dXcol[m][(tap, ci)]
__global__ __launch_bounds__(256) void col2im_kernel(const float* dxcol, const RowEnt* rows, float* din, int M, int taps, int KW, int cin,
                                                     int in_cstride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)M * taps * cin;
    if (i >= n) return;
    const int ci = (int)(i % cin);
    const int t = (int)((i / cin) % taps);
    const int m = (int)(i / ((long)cin * taps));
    const float v = dxcol[i];
    if (v == 0.f) return;
    const RowEnt e = rows[m];
    const int ky = t / KW, kx = t - ky * KW;
    atomicAdd(din + ((size_t)e.in_off + (size_t)ky * e.in_pitch + kx) * in_cstride + ci, v);
}

// ZeroPadding2D((1,2)) + MaxPool 3x3 s2 backward: the gradient goes to the first maximum of the window in row-major
// scan order (the element the forward kernel's strict '>' scan keeps); masked by the stem's ReLU.
template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<uint16_t>(const uint16_t* p, float v[8]) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __uint_as_float((w[k >> 1] >> ((k & 1) * 16)) << 16);
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float g[8]);
template <> __device__ __forceinline__ void store8<uint16_t>(uint16_t* p, const float g[8]) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (uint32_t)f2bf_dev(g[2 * k]) | ((uint32_t)f2bf_dev(g[2 * k + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float g[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(g[0], g[1], g[2], g[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(g[4], g[5], g[6], g[7]);
}
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_backward_kernel(const T* stem_out, const float* dpool, T* dz, int B, int ih, int iw,
                                                                 int oh, int ow, int pool_pitch, int pool_plane) {
    // one thread per stem pixel and 8 channels: sum the pooled gradients of the (up to 4) windows whose arg-max it is
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)B * ih * iw * 8;
    if (i >= n) return;
    const int c = (int)(i & 7) * 8;
    const int x = (int)((i >> 3) % iw), y = (int)((i >> 3) / iw % ih), b = (int)((i >> 3) / ((long)iw * ih));
    float v[8], g[8];
    load8<T>(stem_out + i * 8, v);
    bool any = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) { g[k] = 0.f; any |= v[k] > 0.f; }
    if (any) {
        // windows (oy, ox) cover padded rows 2oy..2oy+2 (pad 1 on top) and padded cols 2ox..2ox+2 (pad 2 on the left)
        for (int oy = y / 2; oy <= (y + 1) / 2 && oy < oh; ++oy) {
            if (2 * oy > y + 1 || 2 * oy + 2 < y + 1) continue;
            for (int ox = (x + 1) / 2; ox <= (x + 2) / 2 && ox < ow; ++ox) {
                if (2 * ox > x + 2 || 2 * ox + 2 < x + 2) continue;
                bool first[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) first[k] = v[k] > 0.f;
                for (int wy = 0; wy < 3; ++wy)
                    for (int wx = 0; wx < 3; ++wx) {
                        const int sy = 2 * oy + wy - 1, sx = 2 * ox + wx - 2;
                        float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // zero padding (inputs are post-ReLU)
                        if (sy >= 0 && sy < ih && sx >= 0 && sx < iw)
                            load8<T>(stem_out + (((size_t)b * ih + sy) * iw + sx) * 64 + c, u);
                        const bool before = sy < y || (sy == y && sx < x);
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (u[k] > v[k] || (u[k] == v[k] && before)) first[k] = false;
                    }
                const float* dp = dpool + ((size_t)b * pool_plane + (size_t)(oy + 1) * pool_pitch + (ox + 1)) * 64 + c;
                const float4 d0 = *reinterpret_cast<const float4*>(dp), d1 = *reinterpret_cast<const float4*>(dp + 4);
                const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) if (first[k]) g[k] += d[k];
            }
        }
    }
    store8<T>(dz + i * 8, g);
}

// dW'[n = (tap, ci)][co] and db'[co] (row N-1) of the folded layer -> gradients of the master parameters:
//   dKernel = dW' * s,  dBias = db' * s,  dGamma = (sum_n dW' * K + db' * (bias - mean)) / sigma,  dBeta = db'      (s = gamma / sigma)
// Pass 1: one block per (64 channels, 32 rows): coalesced over channels, partial dot products by atomics.  Pass 2: per channel.
__global__ __launch_bounds__(256) void unfold_grad_kernel(UnfoldArgs a) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int co = blockIdx.x * 64 + c;
    const int nk = a.taps * a.cin;
    float dot = 0.f, sq = 0.f;
    if (co < a.cout) {
        const float s = a.gamma ? a.gamma[co] / sqrtf(a.var[co] + a.eps) : 1.0f;
        const int n1 = min(nk, (int)(blockIdx.y + 1) * 32);
        for (int n = blockIdx.y * 32 + r; n < n1; n += 4) {
            const size_t i = (size_t)n * a.cout + co;
            const float g = a.dwp[i], k = a.kernel[i];
            a.d_kernel[i] += g * s + 2.0f * a.l2 * k;
            dot += g * k;
            sq += k * k;
        }
    }
    red[r][c] = dot;
    __syncthreads();
    if (r == 0 && co < a.cout && a.gamma) atomicAdd(a.dot + co, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    if (a.l2 > 0.f) {                                                  // (uniform) regularisation loss of this block's slice
        __syncthreads();
        red[r][c] = sq;
        __syncthreads();
        if (threadIdx.x < 64) {
            float v = red[0][c] + red[1][c] + red[2][c] + red[3][c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (threadIdx.x == 0 && v != 0.f) atomicAdd(a.l2_loss, a.l2 * v);
        }
    }
}

__global__ __launch_bounds__(256) void unfold_grad_final_kernel(UnfoldArgs a) {
    const int co = blockIdx.x * 256 + threadIdx.x;
    if (co >= a.cout) return;
    const int nk = a.taps * a.cin;
    const float sigma = a.gamma ? sqrtf(a.var[co] + a.eps) : 1.0f;
    const float s = a.gamma ? a.gamma[co] / sigma : 1.0f;
    const float dbp = a.dwp[(size_t)nk * a.cout + co];
    if (a.d_bias) a.d_bias[co] += dbp * s;
    if (a.gamma) {
        const float b = a.bias ? a.bias[co] : 0.f;
        a.d_gamma[co] += (a.dot[co] + dbp * (b - a.mean[co])) / sigma;
        a.d_beta[co] += dbp;
        a.dot[co] = 0.f;                       // ready for the next layer
    }
}

__global__ __launch_bounds__(256) void l2_grad_kernel(const float* w, float* g, long n, float rate, float* loss_acc) {
    __shared__ float red[256];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    if (i < n) { const float v = w[i]; g[i] += 2.0f * rate * v; s = rate * v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0 && red[0] != 0.f) atomicAdd(loss_acc, red[0]);
}

// Squared global gradient norm, deterministic (fixed partition, fixed reduction order, no atomics): data-parallel replicas
// must derive the same clip factor from the same all-reduced gradients, bit for bit.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long n, float* partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float v = g[i]; s += v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* partial, int nblocks, float* acc) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) *acc = red[0];
}

// tf.clip_by_global_norm(5.0) + keras Adam(epsilon): g *= clip / max(norm, clip); m, v updates; w -= lr_t * m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* w, const float* g, float* m, float* v, long n, const float* sumsq, float clip,
                                                   const float* lr_t_ptr, float beta1, float beta2, float eps) {
    const float lr_t = *lr_t_ptr;              // device scalar: the launch can be replayed from a hipGraph with a new rate
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float norm = sqrtf(*sumsq);
    const float gi = g[i] * (clip / fmaxf(norm, clip));
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    w[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

// every convolution of the model in ONE launch: blockIdx.y selects the layer's descriptor (device array)
__global__ __launch_bounds__(256) void fold_pack_all_kernel(const FoldArgs* all) {
    __shared__ uint16_t tile[64][66];                                  // [ci][co]
    const FoldArgs a = all[blockIdx.y];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0)
        for (int i = tid; i < a.cout_pad; i += 256) {
            float b = 0.f;
            if (i < a.cout) {
                b = a.bias ? a.bias[i] : 0.f;
                if (a.gamma) { const float s = a.gamma[i] / sqrtf(a.var[i] + a.eps); b = (b - a.mean[i]) * s + a.beta[i]; }
            }
            a.b_fwd[i] = b;
        }
    if (a.w_fwd32 || a.f32 || (a.cin & 63) || (a.cout_pad & 63)) {             // the stem (fp32 [tap*cin][cout], 3 input channels): element-wise
        const long n = (long)a.cout_pad * a.taps * a.cin;
        for (long i = (long)blockIdx.x * 256 + tid; i < n; i += (long)gridDim.x * 256) {
            const int ci = (int)(i % a.cin);
            const int t = (int)((i / a.cin) % a.taps);
            const int co = (int)(i / ((long)a.cin * a.taps));
            float v = 0.f;
            if (co < a.cout) {
                const float s = a.gamma ? a.gamma[co] / sqrtf(a.var[co] + a.eps) : 1.0f;
                v = a.kernel[((size_t)t * a.cin + ci) * a.cout + co] * s;
            }
            fold_store(a, i, t, ci, co, v);
        }
        return;
    }
    // 64 (ci) x 64 (co) tiles of one tap: the master kernel [tap][ci][co] is read along co; the two gradient-side
    // layouts are co-contiguous as well, the forward layout [co][tap][ci] is written along ci through the LDS tile
    const int ct = a.cin / 64, ot = a.cout_pad / 64;
    const int ntiles = a.taps * ct * ot;
    const int tx = tid & 63, ty = tid >> 6;
    for (int ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const int o_t = ti % ot, c_t = (ti / ot) % ct, t = ti / (ot * ct);
        const int co = o_t * 64 + tx;
        const float s = co < a.cout ? (a.gamma ? a.gamma[co] / sqrtf(a.var[co] + a.eps) : 1.0f) : 0.f;
        for (int r = ty; r < 64; r += 4) {
            const int ci = c_t * 64 + r;
            const float v = co < a.cout ? a.kernel[((size_t)t * a.cin + ci) * a.cout + co] * s : 0.f;
            const uint16_t hv = f2bf_dev(v);
            if (a.w_bwd) a.w_bwd[((size_t)t * a.cin + ci) * a.cout_pad + co] = hv;
            if (a.w_flip) a.w_flip[((size_t)ci * a.taps + (a.taps - 1 - t)) * a.cout_pad + co] = hv;
            tile[r][tx] = hv;
        }
        __syncthreads();
        if (a.w_fwd)
            for (int r = ty; r < 64; r += 4)
                a.w_fwd[((size_t)(o_t * 64 + r) * a.taps + t) * a.cin + c_t * 64 + tx] = tile[tx][r];
        __syncthreads();
    }
}
hipError_t launch_fold_pack_all(const FoldArgs* device_array, int count, long max_elems, hipStream_t s) {
    const long want = (max_elems + 4095) / 4096;                      // tiles of the largest layer
    hipLaunchKernelGGL(fold_pack_all_kernel, dim3((unsigned)(want < 128 ? want : 128), count), dim3(256), 0, s, device_array);
    return hipGetLastError();
}

hipError_t launch_fold_pack(const FoldArgs& a, hipStream_t s) {
    const long n = (long)a.cout_pad * a.taps * a.cin;
    hipLaunchKernelGGL(fold_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_act_backward_gather(const ActBwdArgs& a, hipStream_t s, bool* wrote_transpose) {
    if (wrote_transpose) *wrote_transpose = false;
    if (a.f32) {                                  // fp32 training handle: one thread per element, fp32 dZ (the caller transposes)
        const long n = (long)a.M * a.cout_pad;
        hipLaunchKernelGGL(act_backward_gather_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (a.dzt && a.cout % 8 == 0 && a.cout_pad % 64 == 0 && a.out_cstride % 8 == 0 && a.Kpad % 64 == 0) {
        hipLaunchKernelGGL(act_backward_tile_kernel, dim3(a.Kpad / 64, a.cout_pad / 64), dim3(256), 0, s, a);
        if (wrote_transpose) *wrote_transpose = true;
        return hipGetLastError();
    }
    // layers with a residual input keep the one-thread-per-element form: its atomics are coalesced across the wavefront
    if (!a.dres && a.cout % 8 == 0 && a.cout_pad % 8 == 0 && a.out_cstride % 8 == 0) {
        const long nv = (long)a.M * (a.cout_pad / 8);
        hipLaunchKernelGGL(act_backward_gather_vec_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const long n = (long)a.M * a.cout_pad;
    hipLaunchKernelGGL(act_backward_gather_kernel<uint16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_relu_merge(const float* dout_relu, const void* out, float* dout, long n, hipStream_t s, bool f32) {
    if (f32) hipLaunchKernelGGL(relu_merge_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dout_relu, reinterpret_cast<const float*>(out), dout, n);
    else hipLaunchKernelGGL(relu_merge_kernel<uint16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dout_relu, reinterpret_cast<const uint16_t*>(out), dout, n);
    return hipGetLastError();
}
hipError_t launch_col2im(const float* dxcol, const RowEnt* rows, float* din, int M, int taps, int KW, int cin, int in_cstride, hipStream_t s) {
    const long n = (long)M * taps * cin;
    hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dxcol, rows, din, M, taps, KW, cin, in_cstride);
    return hipGetLastError();
}
hipError_t launch_stem_pool_backward(const void* stem_out, const float* dpool, void* dz, int B, int ih, int iw, int oh, int ow, int pool_pitch,
                                     int pool_plane, hipStream_t s, bool f32) {
    const long n = (long)B * ih * iw * 8;
    if (f32) hipLaunchKernelGGL(stem_pool_backward_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float*>(stem_out), dpool,
                                reinterpret_cast<float*>(dz), B, ih, iw, oh, ow, pool_pitch, pool_plane);
    else hipLaunchKernelGGL(stem_pool_backward_kernel<uint16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(stem_out), dpool,
                       reinterpret_cast<uint16_t*>(dz), B, ih, iw, oh, ow, pool_pitch, pool_plane);
    return hipGetLastError();
}
hipError_t launch_unfold_grad(const UnfoldArgs& a, hipStream_t s) {
    const int nk = a.taps * a.cin;
    hipLaunchKernelGGL(unfold_grad_kernel, dim3((a.cout + 63) / 64, (nk + 31) / 32), dim3(256), 0, s, a);
    hipLaunchKernelGGL(unfold_grad_final_kernel, dim3((a.cout + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_l2_grad(const float* w, float* g, long n, float rate, float* loss_acc, hipStream_t s) {
    hipLaunchKernelGGL(l2_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, g, n, rate, loss_acc);
    return hipGetLastError();
}
hipError_t launch_sumsq(const float* g, long n, float* acc, float* partial1024, hipStream_t s) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(1024), dim3(256), 0, s, g, n, partial1024);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, partial1024, 1024, acc);
    return hipGetLastError();
}
hipError_t launch_adam(float* w, const float* g, float* m, float* v, long n, const float* sumsq, float clip, const float* lr_t, float beta1, float beta2,
                       float eps, hipStream_t s) {
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, g, m, v, n, sumsq, clip, lr_t, beta1, beta2, eps);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* in, uint16_t* out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = f2bf_dev(in[i]);
}
hipError_t launch_f32_to_bf16(const float* in, void* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, reinterpret_cast<uint16_t*>(out), n);
    return hipGetLastError();
}

// row table of the input-gradient GEMM of a 1x1 layer: row m reads dZ row m and accumulates into input pixel in_off(m)
__global__ __launch_bounds__(256) void make_dgrad_rows_kernel(const RowEnt* fwd, RowEnt* out, int M) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    RowEnt e{};
    e.in_off = m; e.out_off = fwd[m].in_off;
    out[m] = e;
}
hipError_t launch_make_dgrad_rows(const RowEnt* fwd, RowEnt* out, int M, hipStream_t s) {
    hipLaunchKernelGGL(make_dgrad_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, s, fwd, out, M);
    return hipGetLastError();
}
