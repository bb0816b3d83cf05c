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
__global__ __launch_bounds__(256) void gather_transpose_kernel(const uint16_t* in, const RowEnt* rows, uint16_t* out,
                                                               int M, int Kpad, int C, int cstride, int KW) {
    __shared__ uint16_t tile[64][66];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
    const int ky = tap / KW, kx = tap - ky * KW;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;           // 4 rows of 64 threads
    for (int r = ty; r < 64; r += 4) {
        const int m = m0 + r, c = c0 + tx;
        uint16_t v = 0;
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

__global__ void fill_row_bf16_kernel(uint16_t* row, int n_set, int n_total, uint16_t value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_total) row[i] = i < n_set ? value : (uint16_t)0;
}

hipError_t launch_gather_transpose(const void* in, const RowEnt* rows, void* out, int M, int Kpad, int C, int cstride,
                                   int taps, int KW, hipStream_t s) {
    dim3 grid((Kpad + 63) / 64, (C + 63) / 64, taps);
    hipLaunchKernelGGL(gather_transpose_kernel, grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(in), rows,
                       reinterpret_cast<uint16_t*>(out), M, Kpad, C, cstride, KW);
    return hipGetLastError();
}

hipError_t launch_fill_row_bf16(void* row, int n_set, int n_total, float value, hipStream_t s) {
    uint32_t u; __builtin_memcpy(&u, &value, 4);
    hipLaunchKernelGGL(fill_row_bf16_kernel, dim3((n_total + 255) / 256), dim3(256), 0, s, reinterpret_cast<uint16_t*>(row), n_set, n_total,
                       (uint16_t)(u >> 16));
    return hipGetLastError();
}
