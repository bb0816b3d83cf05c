// Training-loss FORWARD on the device (BASELINE config 5):
// RetinaNetModel.get_loss for 'classification' (softmax focal loss, src/core/losses.py:30-61),
// 'regression' (Huber), 'regression_var' / 'regression_covar'
// (src/retina_net/models/retinanet_model.py:183-323).  One thread per (image, anchor); block partial
// sums are written out and added on the host in double, so the result is run-to-run deterministic.
#include "kernels.h"
#include <math.h>

#define LOSS_BLOCK 256

__device__ __forceinline__ float huber1(float e) {
    const float a = fabsf(e);
    return a <= 1.0f ? 0.5f * e * e : a - 0.5f;
}

template <int C>
__global__ __launch_bounds__(LOSS_BLOCK) void loss_kernel(LossArgs a, float* __restrict__ partial) {
    __shared__ float red[4][LOSS_BLOCK];
    const int tid = threadIdx.x;
    const long long idx = (long long)blockIdx.x * LOSS_BLOCK + tid;
    float s_cls = 0.f, s_cmp = 0.f, s_reg = 0.f, s_pos = 0.f;
    if (idx < (long long)a.B * a.A) {
        const int an = (int)(idx % a.A);
        const float pos = a.pos[idx] ? 1.f : 0.f, neg = a.neg[idx] ? 1.f : 0.f;
        s_pos = pos;
        if (a.do_cls && (pos + neg) > 0.f) {
            const float* x = a.cls + idx * C;
            const float* y = a.cls_t + idx * C;
            float v[C], mx = x[0];
#pragma unroll
            for (int j = 0; j < C; ++j) { v[j] = x[j]; mx = fmaxf(mx, v[j]); }
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) se += expf(v[j] - mx);
            const float lse = logf(se);
            float pt = 0.f, ce = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const float ls = v[j] - mx - lse;
                pt += expf(ls) * y[j];
                ce -= (y[j] * (1.0f - a.label_smoothing) + a.label_smoothing / (float)C) * ls;
            }
            const float ngm = y[C - 1];
            const float alpha = 0.5f * (1.0f - ngm) + 0.5f * ngm;
            const float f = 1.0f - pt;
            s_cls = alpha * f * f * ce * (pos + neg);
        }
        if (a.reg_kind && pos > 0.f) {
            const float4 p = reinterpret_cast<const float4*>(a.box)[idx];
            const float4 t = reinterpret_cast<const float4*>(a.box_t)[idx];
            if (a.reg_kind == 1) {                                   // plain Huber, mean over the 4 coordinates
                s_cmp = 0.25f * (huber1(p.x - t.x) + huber1(p.y - t.y) + huber1(p.z - t.z) + huber1(p.w - t.w));
            } else {
                const float4 anc = reinterpret_cast<const float4*>(a.anchors)[an];
                float pb[4], tb[4];
                pb[0] = anc.z * p.x / 10.0f + anc.x; tb[0] = anc.z * t.x / 10.0f + anc.x;
                pb[1] = anc.w * p.y / 10.0f + anc.y; tb[1] = anc.w * t.y / 10.0f + anc.y;
                pb[2] = anc.z * fminf(fmaxf(expf(p.z / 5.0f), 1e-4f), 1e4f); tb[2] = anc.z * fminf(fmaxf(expf(t.z / 5.0f), 1e-4f), 1e4f);
                pb[3] = anc.w * fminf(fmaxf(expf(p.w / 5.0f), 1e-4f), 1e4f); tb[3] = anc.w * fminf(fmaxf(expf(t.w / 5.0f), 1e-4f), 1e4f);
                const float* c = a.cov + idx * 10;                   // fill_triangular params: diag = (x4,x9,x5,x0)
                const float ld[4] = {c[4], c[9], c[5], c[0]};
                float cmp = 0.f, reg = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { cmp += expf(-ld[k]) * huber1(pb[k] - tb[k]); reg += ld[k]; }
                if (a.reg_kind == 3) {                               // x ||L_inv||_F, unit diagonal
                    const float fro = sqrtf(4.0f + c[8] * c[8] + c[7] * c[7] + c[6] * c[6] + c[3] * c[3] + c[2] * c[2] + c[1] * c[1]);
                    cmp *= fro;
                }
                s_cmp = cmp; s_reg = 0.5f * reg;
            }
        }
    }
    red[0][tid] = s_cls; red[1][tid] = s_cmp; red[2][tid] = s_reg; red[3][tid] = s_pos;
    __syncthreads();
    for (int s = LOSS_BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + s];
        }
        __syncthreads();
    }
    if (tid < 4) partial[(size_t)blockIdx.x * 4 + tid] = red[tid][0];
}

hipError_t launch_loss(const LossArgs& a, float* partial, int nblocks, hipStream_t s) {
    if (a.C == 8) hipLaunchKernelGGL(loss_kernel<8>, dim3(nblocks), dim3(LOSS_BLOCK), 0, s, a, partial);
    else if (a.C == 4) hipLaunchKernelGGL(loss_kernel<4>, dim3(nblocks), dim3(LOSS_BLOCK), 0, s, a, partial);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Loss BACKWARD: gradients of
//     total = w_cls * S_cls / max(n_pos, 1) + w_reg * (S_cmp + S_reg) / max(n_pos, 1)
// with respect to the raw head outputs (class logits, box regression targets, the 10 covariance parameters),
// exactly the terms loss_kernel sums; the focal modulating factor is differentiated too (the reference
// applies no stop_gradient, src/core/losses.py:44-61).  sums[3] = n_pos comes from the forward pass on the device.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float huber1_grad(float e) { return fminf(fmaxf(e, -1.0f), 1.0f); }

template <int C>
__global__ __launch_bounds__(LOSS_BLOCK) void loss_backward_kernel(LossArgs a, const float* __restrict__ sums, float w_cls, float w_reg,
                                                                   float* __restrict__ dcls, float* __restrict__ dbox, float* __restrict__ dcov) {
    const long long idx = (long long)blockIdx.x * LOSS_BLOCK + threadIdx.x;
    if (idx >= (long long)a.B * a.A) return;
    const int an = (int)(idx % a.A);
    const float inv_n = 1.0f / fmaxf(sums[3], 1.0f);
    const float pos = a.pos[idx] ? 1.f : 0.f, neg = a.neg[idx] ? 1.f : 0.f;
    if (dcls) {
        float g[C];
#pragma unroll
        for (int j = 0; j < C; ++j) g[j] = 0.f;
        if (a.do_cls && (pos + neg) > 0.f) {
            const float* x = a.cls + idx * C;
            const float* y = a.cls_t + idx * C;
            float v[C], mx = x[0];
#pragma unroll
            for (int j = 0; j < C; ++j) { v[j] = x[j]; mx = fmaxf(mx, v[j]); }
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) se += expf(v[j] - mx);
            const float lse = logf(se);
            float p[C], q[C], pt = 0.f, ce = 0.f, Q = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const float ls = v[j] - mx - lse;
                p[j] = expf(ls);
                q[j] = y[j] * (1.0f - a.label_smoothing) + a.label_smoothing / (float)C;
                pt += p[j] * y[j]; ce -= q[j] * ls; Q += q[j];
            }
            const float ngm = y[C - 1];
            const float alpha = 0.5f * (1.0f - ngm) + 0.5f * ngm;
            const float f = 1.0f - pt;
            const float k = alpha * (pos + neg) * w_cls * inv_n;
#pragma unroll
            for (int j = 0; j < C; ++j)
                g[j] = k * (-2.0f * f * ce * p[j] * (y[j] - pt) + f * f * (p[j] * Q - q[j]));
        }
#pragma unroll
        for (int j = 0; j < C; ++j) dcls[idx * C + j] = g[j];
    }
    float gb[4] = {0.f, 0.f, 0.f, 0.f};
    float gc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.reg_kind && pos > 0.f) {
        const float4 p = reinterpret_cast<const float4*>(a.box)[idx];
        const float4 t = reinterpret_cast<const float4*>(a.box_t)[idx];
        const float k = w_reg * inv_n;
        if (a.reg_kind == 1) {
            gb[0] = 0.25f * k * huber1_grad(p.x - t.x); gb[1] = 0.25f * k * huber1_grad(p.y - t.y);
            gb[2] = 0.25f * k * huber1_grad(p.z - t.z); gb[3] = 0.25f * k * huber1_grad(p.w - t.w);
        } else {
            const float4 anc = reinterpret_cast<const float4*>(a.anchors)[an];
            const float ez = expf(p.z / 5.0f), ew = expf(p.w / 5.0f);
            float pb[4], tb[4], dpb[4];
            pb[0] = anc.z * p.x / 10.0f + anc.x; tb[0] = anc.z * t.x / 10.0f + anc.x; dpb[0] = anc.z / 10.0f;
            pb[1] = anc.w * p.y / 10.0f + anc.y; tb[1] = anc.w * t.y / 10.0f + anc.y; dpb[1] = anc.w / 10.0f;
            pb[2] = anc.z * fminf(fmaxf(ez, 1e-4f), 1e4f); tb[2] = anc.z * fminf(fmaxf(expf(t.z / 5.0f), 1e-4f), 1e4f);
            pb[3] = anc.w * fminf(fmaxf(ew, 1e-4f), 1e4f); tb[3] = anc.w * fminf(fmaxf(expf(t.w / 5.0f), 1e-4f), 1e4f);
            dpb[2] = (ez >= 1e-4f && ez <= 1e4f) ? anc.z * ez / 5.0f : 0.f;           // clip_by_value passes no gradient outside
            dpb[3] = (ew >= 1e-4f && ew <= 1e4f) ? anc.w * ew / 5.0f : 0.f;
            const float* c = a.cov + idx * 10;
            const int di[4] = {4, 9, 5, 0};                                           // fill_triangular diagonal = (x4, x9, x5, x0)
            float eld[4], hub[4], cmp = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { eld[q] = expf(-c[di[q]]); hub[q] = huber1(pb[q] - tb[q]); cmp += eld[q] * hub[q]; }
            float fro = 1.0f;
            if (a.reg_kind == 3) fro = sqrtf(4.0f + c[8] * c[8] + c[7] * c[7] + c[6] * c[6] + c[3] * c[3] + c[2] * c[2] + c[1] * c[1]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gb[q] = k * fro * eld[q] * huber1_grad(pb[q] - tb[q]) * dpb[q];
                gc[di[q]] = k * (-fro * eld[q] * hub[q] + 0.5f);
            }
            if (a.reg_kind == 3) {
                const int od[6] = {8, 7, 6, 3, 2, 1};
#pragma unroll
                for (int q = 0; q < 6; ++q) gc[od[q]] = k * cmp * c[od[q]] / fro;
            }
        }
    }
    if (dbox) reinterpret_cast<float4*>(dbox)[idx] = make_float4(gb[0], gb[1], gb[2], gb[3]);
    if (dcov) {
#pragma unroll
        for (int q = 0; q < 10; ++q) dcov[idx * 10 + q] = gc[q];
    }
}

// sums[0..3] = (S_cls, S_cmp, S_reg, n_pos) from the block partials, on the device (one block)
__global__ __launch_bounds__(LOSS_BLOCK) void loss_reduce_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ sums) {
    __shared__ double red[4][LOSS_BLOCK];
    double acc[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblocks; i += LOSS_BLOCK)
        for (int q = 0; q < 4; ++q) acc[q] += (double)partial[(size_t)i * 4 + q];
    for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int s = LOSS_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) sums[threadIdx.x] = (float)red[threadIdx.x][0];
}

hipError_t launch_loss_reduce(const float* partial, int nblocks, float* sums, hipStream_t s) {
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(LOSS_BLOCK), 0, s, partial, nblocks, sums);
    return hipGetLastError();
}

hipError_t launch_loss_backward(const LossArgs& a, const float* sums, float w_cls, float w_reg, float* dcls, float* dbox, float* dcov,
                                hipStream_t s) {
    const int nblocks = (int)(((long long)a.B * a.A + LOSS_BLOCK - 1) / LOSS_BLOCK);
    if (a.C == 8) hipLaunchKernelGGL(loss_backward_kernel<8>, dim3(nblocks), dim3(LOSS_BLOCK), 0, s, a, sums, w_cls, w_reg, dcls, dbox, dcov);
    else if (a.C == 4) hipLaunchKernelGGL(loss_backward_kernel<4>, dim3(nblocks), dim3(LOSS_BLOCK), 0, s, a, sums, w_cls, w_reg, dcls, dbox, dcov);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
