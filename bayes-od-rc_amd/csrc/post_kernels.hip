// Bayesian post-processing on the device: everything `bayes_od_inference` does after the model
// call and all of `bayes_od_clustering` (src/retina_net/experiments/inference_utils.py:25-215,
// :285-364).  fp32 throughout, like the reference.
//
//   K1 post_sample_kernel   softmax per MC sample -> mean -> Categorical.sample(30) via Philox
//                           uniforms -> counts -> background filter flag            (:31-51)
//   K2 post_scan_kernel     ordered compaction offsets (tf.boolean_mask keeps anchor order)
//   K3 post_fuse_kernel     per kept anchor: decode, mean / 4x4 sample covariance over MC,
//                           aleatoric L D L^T, mixing, Dirichlet + Gaussian prior fusion,
//                           KITTI rescale, score ranking, corners                    (:25-29,:53-202)
//   K4 nms_kernel           NonMaxSuppressionV5 (soft-NMS), one wavefront per image   (:204-212)
//   K5 cluster_fuse_kernel  bayes_od_clustering, one workgroup per (image, centre)   (:285-364)
//   K6 iou_matrix_kernel    box_utils.bbox_iou_vuvu (only for API compatibility)     (:214-215)
#include "kernels.h"
#include "philox.h"
#include <algorithm>
#include <math.h>

#define POST_BLOCK 256
#define MAXC 16

// ------------------------------------------------------------------------------------------------
// small dense helpers (all indices compile-time after unrolling)
// ------------------------------------------------------------------------------------------------
struct Mat4 { float m[4][4]; };

// inverse of a symmetric positive definite 4x4 through Cholesky A = G G^T, inv = G^-T G^-1
__device__ __forceinline__ Mat4 inv_spd4_once(const Mat4& a) {
    float g[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) g[i][j] = 0.f;
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
    // h = G^-1 (lower)
    float h[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) h[i][j] = 0.f;
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
    Mat4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = i; k < 4; ++k) s += h[k][i] * h[k][j];
            r.m[i][j] = s;
            r.m[j][i] = s;
        }
    return r;
}

// DESIGN.md 8.4 (round 6): this file -- like every source of the library -- is compiled WITHOUT the SLP vectoriser (bayes_od_rc_amd/build.py: -fno-slp-vectorize), i.e. without
// packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).  With them, post_fuse_kernel and cluster_fuse_kernel -- the 4x4
// inverses and matrix-vector products, which the vectoriser packs two floats at a time -- returned wrong results in lanes 48-63 of a wave
// whenever a convolution kernel of the library shared the compute unit (0.2-10 % of the waves in the self-check build,
// tests/tools/selfcheck_probe.py; never alone; never on disjoint CUs); without them: 0 of 56 million waves, and the canary test
// (tests/test_gpu_zz_canary.py) passes.  Same IEEE operations either way: results are bit-identical to the packed build's when nothing
// runs beside it.  (A round-6 interim form evaluated the inverse until two consecutive evaluations agreed: it removed 85 % of the events,
// not all -- the products behind the inverse are packed too.)
__device__ __forceinline__ Mat4 inv_spd4(const Mat4& a) { return inv_spd4_once(a); }

__device__ __forceinline__ void decode_box(const float4 anc, const float4 t, float o[4]) {
    // box_utils.box_from_anchor_and_target_bnms (:171-192): anchors (v,u,h,w)
    o[0] = anc.z * t.x / 10.0f + anc.x;
    o[1] = anc.w * t.y / 10.0f + anc.y;
    o[2] = anc.z * fminf(fmaxf(expf(t.z / 5.0f), 1e-4f), 1e4f);
    o[3] = anc.w * fminf(fmaxf(expf(t.w / 5.0f), 1e-4f), 1e4f);
}

// ------------------------------------------------------------------------------------------------
// K1: mean softmax + categorical sampling + filter flag
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(POST_BLOCK) void post_sample_kernel(PostCfg c, PostBuffers pb, int nblocks) {
    const int a = blockIdx.x * POST_BLOCK + threadIdx.x;
    const int b = blockIdx.y;
    bool keep = false;
    if (a < c.A) {
        float mp[C];
#pragma unroll
        for (int j = 0; j < C; ++j) mp[j] = 0.f;
        if (c.aggregated) {                 // sum_n softmax from the head epilogue (conv_igemm.hip agg_reduce_cls: the loop below, fused)
            const float* l = pb.agg_cls + ((size_t)b * c.A + a) * C;
#pragma unroll
            for (int j = 0; j < C; ++j) mp[j] = l[j];
        }
        for (int n = 0; n < (c.aggregated ? 0 : c.N); ++n) {
            const float* l = pb.cls + (((size_t)b * c.N + n) * c.A + a) * C;
            float v[C];
            if (C % 4 == 0) {
#pragma unroll
                for (int q = 0; q < C / 4; ++q) {
                    const float4 t = reinterpret_cast<const float4*>(l)[q];
                    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < C; ++j) v[j] = l[j];
            }
            float mx = v[0];
#pragma unroll
            for (int j = 1; j < C; ++j) mx = fmaxf(mx, v[j]);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) { v[j] = expf(v[j] - mx); s += v[j]; }
#pragma unroll
            for (int j = 0; j < C; ++j) mp[j] += v[j] / s;
        }
        float cdf[C];
        float accum = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) { mp[j] = mp[j] / (float)c.N; accum += mp[j]; cdf[j] = accum; }
        const float total = cdf[C - 1];
        int cnt[C];
#pragma unroll
        for (int j = 0; j < C; ++j) cnt[j] = 0;
        const int groups = (c.draws + 3) / 4;
        for (int g = 0; g < groups; ++g) {
            const Philox4 r = philox4x32_10((uint32_t)a, (uint32_t)g, BOD_CAT_TAG, c.image_base + b,
                                            c.seed_lo, c.seed_hi);
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (4 * g + q < c.draws) {
                    const float t = (float)(w[q] >> 8) * 5.9604644775390625e-8f * total;
                    int k = 0;
#pragma unroll
                    for (int j = 0; j < C; ++j) k += (cdf[j] <= t) ? 1 : 0;
                    k = k < C - 1 ? k : C - 1;
#pragma unroll
                    for (int j = 0; j < C; ++j) cnt[j] += (j == k) ? 1 : 0;
                }
            }
        }
        int best = 0, bestc = cnt[0];
#pragma unroll
        for (int j = 1; j < C; ++j)
            if (cnt[j] > bestc) { bestc = cnt[j]; best = j; }
        keep = best != C - 1;
        pb.keep[(size_t)b * c.A + a] = keep ? 1 : 0;
        float* dc = pb.d_counts + ((size_t)b * c.A + a) * C;
#pragma unroll
        for (int j = 0; j < C; ++j) dc[j] = (float)cnt[j];
    }
    const int n = __syncthreads_count(keep ? 1 : 0);
    if (threadIdx.x == 0) pb.block_counts[(size_t)b * nblocks + blockIdx.x] = n;
}

// K2: exclusive scan of block counts (in place), one block per image
__global__ __launch_bounds__(POST_BLOCK) void post_scan_kernel(PostBuffers pb, int nblocks) {
    __shared__ int part[POST_BLOCK];
    const int b = blockIdx.x, tid = threadIdx.x;
    int* bc = pb.block_counts + (size_t)b * nblocks;
    const int per = (nblocks + POST_BLOCK - 1) / POST_BLOCK;
    int local = 0;
    for (int i = 0; i < per; ++i) {
        const int idx = tid * per + i;
        if (idx < nblocks) local += bc[idx];
    }
    part[tid] = local;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < POST_BLOCK; ++i) { const int t = part[i]; part[i] = run; run += t; }
        pb.num_kept[b] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int i = 0; i < per; ++i) {
        const int idx = tid * per + i;
        if (idx < nblocks) { const int t = bc[idx]; bc[idx] = run; run += t; }
    }
}

// K3a (round 4): the ordered compaction alone -- every kept anchor writes its index to its slot.  The per-anchor work below then runs
// over the COMPACT list: with ~2 % of the anchors kept, a wave of the one-thread-per-anchor form ran the whole posterior (three 4x4
// inverses, the aleatoric L D L^T, the prior fusion) for one or two active lanes -- 0.61 ms per 512 frames on the main stream.
__global__ __launch_bounds__(POST_BLOCK) void post_compact_kernel(PostCfg c, PostBuffers pb, int nblocks) {
    __shared__ int wave_off[POST_BLOCK / 64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = blockIdx.x * POST_BLOCK + tid;
    const int b = blockIdx.y;
    const bool keep = (a < c.A) && pb.keep[(size_t)b * c.A + a];
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_off[wave + 1] = __popcll(bal);
    if (tid == 0) wave_off[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int w = 1; w <= POST_BLOCK / 64; ++w) wave_off[w] += wave_off[w - 1];
    __syncthreads();
    if (!keep) return;
    const int slot = pb.block_counts[(size_t)b * nblocks + blockIdx.x] + wave_off[wave] + before;
    pb.anchor_index[(size_t)b * c.A + slot] = a;
}

// K3: per kept anchor fusion, written straight to its compacted slot
// BOD_POST_SELFCHECK (development build, tests/tools/build_variant.sh + selfcheck_probe.py): every slot is computed TWICE
// in the same thread through ONE out-of-line copy of the code -- the same machine instructions on the same inputs -- and the two
// results are compared bit for bit; a difference is logged with where the wave ran (HW_REG_HW_ID: SIMD / CU / SE, HW_REG_XCC_ID).
#if defined(BOD_POST_SELFCHECK)
#define POST_FUSE_INLINE __noinline__
#if BOD_POST_SELFCHECK == 2
// variant 2: the slot's inputs are loaded ONCE by the kernel and handed to the out-of-line copy in registers; the copy neither loads nor
// stores (pure VALU between call and return) -- does the difference between two computations survive without memory operations?
struct FuseIn { float4 anc, ab[4]; float2 cv[5]; float dc[8]; };
#define POST_FUSE_OUT , float* __restrict__ out20, const FuseIn& fin
#else
#define POST_FUSE_OUT , float* __restrict__ out20
#endif
struct SelfcheckRec { uint32_t hw_id, xcc_id, image, slot, lane, elem; float first, second; };
__device__ unsigned int g_selfcheck_count;
__device__ unsigned long long g_selfcheck_waves;          // waves that ran the check (the rate's denominator)
__device__ SelfcheckRec g_selfcheck_recs[4096];
#else
#define POST_FUSE_INLINE __forceinline__
#define POST_FUSE_OUT
#endif
template <int C>
__device__ POST_FUSE_INLINE void post_fuse_anchor(const PostCfg& c, const PostBuffers& pb, const int b, const int slot POST_FUSE_OUT) {
    const size_t o = (size_t)b * c.A + slot;
#if defined(BOD_POST_SELFCHECK) && BOD_POST_SELFCHECK == 2
    const int a = 0; (void)a; (void)o;
    const float4 anc = fin.anc;
#else
    const int a = pb.anchor_index[o];

    const float4 anc = reinterpret_cast<const float4*>(pb.anchors)[a];
#endif
    // ---- epistemic: two-pass mean / unbiased covariance over MC samples (:220-244)
    float mu[4] = {0.f, 0.f, 0.f, 0.f};
    Mat4 epi;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) epi.m[i][j] = 0.f;
    const int n_raw = c.aggregated ? 0 : c.N;
    if (c.aggregated) {                     // Welford mean and co-moment sums from the head epilogue (agg_reduce_box)
#if defined(BOD_POST_SELFCHECK) && BOD_POST_SELFCHECK == 2
        const float4* ab = fin.ab;
#else
        const float4* ab = reinterpret_cast<const float4*>(pb.agg_box) + ((size_t)b * c.A + a) * 4;
#endif
        const float4 m = ab[0], q0 = ab[1], q1 = ab[2], q2 = ab[3];
        mu[0] = m.x; mu[1] = m.y; mu[2] = m.z; mu[3] = m.w;
        epi.m[0][0] = q0.x; epi.m[1][0] = q0.y; epi.m[1][1] = q0.z; epi.m[2][0] = q0.w;
        epi.m[2][1] = q1.x; epi.m[2][2] = q1.y; epi.m[3][0] = q1.z; epi.m[3][1] = q1.w;
        epi.m[3][2] = q2.x; epi.m[3][3] = q2.y;
    }
    for (int n = 0; n < n_raw; ++n) {
        const float4 t = reinterpret_cast<const float4*>(pb.box)[((size_t)b * c.N + n) * c.A + a];
        float bx[4];
        decode_box(anc, t, bx);
#pragma unroll
        for (int i = 0; i < 4; ++i) mu[i] += bx[i];
    }
    if (!c.aggregated) {
#pragma unroll
        for (int i = 0; i < 4; ++i) mu[i] = mu[i] / (float)c.N;
    }
    for (int n = 0; n < n_raw; ++n) {
        const float4 t = reinterpret_cast<const float4*>(pb.box)[((size_t)b * c.N + n) * c.A + a];
        float bx[4];
        decode_box(anc, t, bx);
#pragma unroll
        for (int i = 0; i < 4; ++i) bx[i] -= mu[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) epi.m[i][j] += bx[i] * bx[j];
    }
    const float nm1 = (float)c.N - 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) { epi.m[i][j] = epi.m[i][j] / nm1; epi.m[j][i] = epi.m[i][j]; }

#if defined(BOD_POST_SELFCHECK_PART) && BOD_POST_SELFCHECK_PART == 1          // (bisection of the self-check: stop behind the epistemic block)
    for (int i = 0; i < 4; ++i) { out20[i] = mu[i]; for (int j = 0; j < 4; ++j) out20[4 + i * 4 + j] = epi.m[i][j]; }
    return;
#endif
    // ---- aleatoric (:62-84): mean of the raw lower-triangular params, D = exp(diag), L = inv(unit lower)
    Mat4 al;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) al.m[i][j] = 0.f;
    if (c.has_covar) {
        float x[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) x[q] = 0.f;
        if (c.aggregated) {                 // sum_n of the raw parameters from the head epilogue (agg_reduce_cov)
#if defined(BOD_POST_SELFCHECK) && BOD_POST_SELFCHECK == 2
            const float* p = reinterpret_cast<const float*>(fin.cv);
#else
            const float* p = pb.agg_cov + ((size_t)b * c.A + a) * 10;
#endif
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const float2 t = reinterpret_cast<const float2*>(p)[q];
                x[2 * q] = t.x; x[2 * q + 1] = t.y;
            }
        }
        for (int n = 0; n < n_raw; ++n) {
            const float* p = pb.cov + (((size_t)b * c.N + n) * c.A + a) * 10;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const float2 t = reinterpret_cast<const float2*>(p)[q];
                x[2 * q] += t.x; x[2 * q + 1] += t.y;
            }
        }
#pragma unroll
        for (int q = 0; q < 10; ++q) x[q] = x[q] / (float)c.N;
        // tfp.math.fill_triangular (retinanet_model.py:110)
        const float d0 = expf(x[4]), d1 = expf(x[9]), d2 = expf(x[5]), d3 = expf(x[0]);
        if (c.use_full_covar) {
            const float l10 = x[8], l20 = x[7], l21 = x[6], l30 = x[3], l31 = x[2], l32 = x[1];
            // inverse of the unit lower matrix by forward substitution
            float li[4][4];
            li[0][0] = 1.f; li[0][1] = 0.f; li[0][2] = 0.f; li[0][3] = 0.f;
            li[1][0] = -l10; li[1][1] = 1.f; li[1][2] = 0.f; li[1][3] = 0.f;
            li[2][0] = -(l20 * 1.f + l21 * li[1][0]); li[2][1] = -l21; li[2][2] = 1.f; li[2][3] = 0.f;
            li[3][0] = -(l30 * 1.f + l31 * li[1][0] + l32 * li[2][0]);
            li[3][1] = -(l31 * 1.f + l32 * li[2][1]);
            li[3][2] = -l32; li[3][3] = 1.f;
            const float d[4] = {d0, d1, d2, d3};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int k = 0; k <= j; ++k) s += li[i][k] * d[k] * li[j][k];
                    al.m[i][j] = s; al.m[j][i] = s;
                }
        } else {
            al.m[0][0] = d0; al.m[1][1] = d1; al.m[2][2] = d2; al.m[3][3] = d3;
        }
    }
#if defined(BOD_POST_SELFCHECK_PART) && BOD_POST_SELFCHECK_PART == 2          // ... behind the aleatoric block (exp, unit-lower inverse, L D L^T)
    for (int i = 0; i < 4; ++i) { out20[i] = mu[i]; for (int j = 0; j < 4; ++j) out20[4 + i * 4 + j] = al.m[i][j]; }
    return;
#endif
    Mat4 lik;                                                     // (:86-87)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) lik.m[i][j] = (10.0f * al.m[i][j] + 1.0f * epi.m[i][j]) / 11.0f;

    // ---- Dirichlet posterior (:90-97)
#if defined(BOD_POST_SELFCHECK) && BOD_POST_SELFCHECK == 2
    const float* dc = fin.dc;
#else
    const float* dc = pb.d_counts + ((size_t)b * c.A + a) * C;
#endif
    float pc[C];
    float csum = 0.f;
    const float alpha = c.dirichlet ? 1.0f / (float)C : 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) { pc[j] = dc[j] + alpha; csum += pc[j]; }
    float best = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const float s = pc[j] / csum;
#if !(defined(BOD_POST_SELFCHECK) && BOD_POST_SELFCHECK == 2)
        pb.counts[o * C + j] = pc[j];
        pb.score[o * C + j] = s;
#endif
        best = j == 0 ? s : fmaxf(best, s);
    }

#if defined(BOD_POST_SELFCHECK_PART) && BOD_POST_SELFCHECK_PART == 3          // ... behind the likelihood mix and the Dirichlet scores
    for (int i = 0; i < 4; ++i) { out20[i] = mu[i] + best; for (int j = 0; j < 4; ++j) out20[4 + i * 4 + j] = lik.m[i][j]; }
    return;
#endif
    // ---- Gaussian prior fusion (:100-145): prior mean = the anchor, prior cov = iso_var * I
    Mat4 pcov;
    float pm[4];
    if (c.gaussian_iso) {
        const Mat4 prec = inv_spd4(lik);
        const float pp = 1.0f / c.iso_var;
        Mat4 post_prec = prec;
#pragma unroll
        for (int i = 0; i < 4; ++i) post_prec.m[i][i] += pp;
        pcov = inv_spd4(post_prec);
        const float am[4] = {anc.x, anc.y, anc.z, anc.w};
        float inter[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += prec.m[i][k] * mu[k];
            inter[i] = pp * am[i] + s;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += pcov.m[i][k] * inter[k];
            pm[i] = s;
        }
    } else {
        pcov = lik;
#pragma unroll
        for (int i = 0; i < 4; ++i) pm[i] = mu[i];
    }
    if (c.kitti_sh > 0.f) {                                       // (:147-167) S mu, S Sigma S^T
        const float sc[4] = {c.kitti_sh, c.kitti_sw, c.kitti_sh, c.kitti_sw};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pm[i] *= sc[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) pcov.m[i][j] = sc[i] * pcov.m[i][j] * sc[j];
        }
    }
#if !(defined(BOD_POST_SELFCHECK) && BOD_POST_SELFCHECK == 2)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pb.means[o * 4 + i] = pm[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) pb.covs[o * 16 + i * 4 + j] = pcov.m[i][j];
    }
#endif
#if defined(BOD_POST_SELFCHECK)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out20[i] = pm[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) out20[4 + i * 4 + j] = pcov.m[i][j];
    }
#endif
#if defined(BOD_POST_SELFCHECK) && BOD_POST_SELFCHECK == 2
    out20[0] += best * 0.f;
    return;
#else
    pb.ranking[o] = best;                                         // ranking_method 'score' (:202)
    // box_utils.vuhw_to_vuvu (:5-23)
    pb.corners[o * 4 + 0] = pm[0] - pm[2] / 2.0f;
    pb.corners[o * 4 + 1] = pm[1] - pm[3] / 2.0f;
    pb.corners[o * 4 + 2] = pm[0] + pm[2] / 2.0f;
    pb.corners[o * 4 + 3] = pm[1] + pm[3] / 2.0f;
#endif
}

// a few blocks per image walk its compact list (POST_FUSE_BLOCKS x 256 slots per pass: one pass up to 2 048 kept anchors)
#define POST_FUSE_BLOCKS 8
template <int C>
__global__ __launch_bounds__(POST_BLOCK) void post_fuse_kernel(PostCfg c, PostBuffers pb, int nblocks) {
    const int b = blockIdx.y;
    const int m = pb.num_kept[b];
#if defined(BOD_POST_SELFCHECK)
    for (int slot = blockIdx.x * POST_BLOCK + threadIdx.x; slot < m; slot += gridDim.x * POST_BLOCK) {
        float r1[20], r2[20];
#if BOD_POST_SELFCHECK == 2
        FuseIn fin;
        {
            const size_t o = (size_t)b * c.A + slot;
            const int a = pb.anchor_index[o];
            fin.anc = reinterpret_cast<const float4*>(pb.anchors)[a];
            for (int q = 0; q < 4; ++q) fin.ab[q] = (reinterpret_cast<const float4*>(pb.agg_box) + ((size_t)b * c.A + a) * 4)[q];
            for (int q = 0; q < 5; ++q) fin.cv[q] = reinterpret_cast<const float2*>(pb.agg_cov + ((size_t)b * c.A + a) * 10)[q];
            for (int q = 0; q < 8; ++q) fin.dc[q] = q < C ? pb.d_counts[((size_t)b * c.A + a) * C + q] : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        post_fuse_anchor<C>(c, pb, b, slot, r1, fin);
        post_fuse_anchor<C>(c, pb, b, slot, r2, fin);
#else
        post_fuse_anchor<C>(c, pb, b, slot, r1);
        post_fuse_anchor<C>(c, pb, b, slot, r2);
#endif
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_selfcheck_waves, 1ull);
        int bad = -1;
        for (int i = 19; i >= 0; --i) if (__float_as_uint(r1[i]) != __float_as_uint(r2[i])) bad = i;
        if (bad >= 0) {
            const unsigned int k = atomicAdd(&g_selfcheck_count, 1u);
            if (k < 4096u) {
                SelfcheckRec r;
                r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID, all 32 bits
                r.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);        // HW_REG_XCC_ID
                r.image = (uint32_t)b; r.slot = (uint32_t)slot; r.lane = threadIdx.x & 63; r.elem = (uint32_t)bad;
                r.first = r1[bad]; r.second = r2[bad];
                g_selfcheck_recs[k] = r;
            }
        }
    }
#else
    for (int slot = blockIdx.x * POST_BLOCK + threadIdx.x; slot < m; slot += gridDim.x * POST_BLOCK) post_fuse_anchor<C>(c, pb, b, slot);
#endif
}
#if defined(BOD_POST_SELFCHECK)
// (development builds only: not in include/bayesod.h)  count of mismatching slots since the last call, the waves that ran the check, up to
// `max` records of 8 dwords {hw_id, xcc_id, image, slot, lane, element, first, second}; resets the log.
extern "C" int bod_debug_selfcheck_read(unsigned int* count, unsigned long long* waves, void* recs, int max) {
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_selfcheck_count), 4) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(waves, HIP_SYMBOL(g_selfcheck_waves), 8) != hipSuccess) return 1;
    const unsigned int m = n < (unsigned)max ? n : (unsigned)max;
    if (m && hipMemcpyFromSymbol(recs, HIP_SYMBOL(g_selfcheck_recs), (size_t)(m < 4096u ? m : 4096u) * sizeof(SelfcheckRec)) != hipSuccess) return 1;
    const unsigned int z = 0; const unsigned long long z8 = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_selfcheck_count), &z, 4) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_selfcheck_waves), &z8, 8) != hipSuccess) return 1;
    *count = n;
    return 0;
}
#endif

hipError_t launch_posterior(const PostCfg& c, const PostBuffers& b, hipStream_t s) {
    const int nblocks = (c.A + POST_BLOCK - 1) / POST_BLOCK;
    dim3 grid(nblocks, c.B);
    if (c.C == 8) {
        hipLaunchKernelGGL(post_sample_kernel<8>, grid, dim3(POST_BLOCK), 0, s, c, b, nblocks);
        hipLaunchKernelGGL(post_scan_kernel, dim3(c.B), dim3(POST_BLOCK), 0, s, b, nblocks);
        hipLaunchKernelGGL(post_compact_kernel, grid, dim3(POST_BLOCK), 0, s, c, b, nblocks);
        hipLaunchKernelGGL(post_fuse_kernel<8>, dim3(std::min(nblocks, POST_FUSE_BLOCKS), c.B), dim3(POST_BLOCK), 0, s, c, b, nblocks);
    } else if (c.C == 4) {
        hipLaunchKernelGGL(post_sample_kernel<4>, grid, dim3(POST_BLOCK), 0, s, c, b, nblocks);
        hipLaunchKernelGGL(post_scan_kernel, dim3(c.B), dim3(POST_BLOCK), 0, s, b, nblocks);
        hipLaunchKernelGGL(post_compact_kernel, grid, dim3(POST_BLOCK), 0, s, c, b, nblocks);
        hipLaunchKernelGGL(post_fuse_kernel<4>, dim3(std::min(nblocks, POST_FUSE_BLOCKS), c.B), dim3(POST_BLOCK), 0, s, c, b, nblocks);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Validation post-processing (src/retina_net/experiments/validation_utils.py:10-77): deterministic single
// forward pass -> softmax -> drop anchors whose arg-max class is background -> rank by the top score -> the
// same soft-NMS kernel.  The compacted candidates land in the posterior buffers (score = counts = softmax row,
// means = decoded box, covs = 0), so bod_nms and the getters serve both paths.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(POST_BLOCK) void val_flag_kernel(PostCfg c, PostBuffers pb, int nblocks) {
    const int a = blockIdx.x * POST_BLOCK + threadIdx.x;
    const int b = blockIdx.y;
    bool keep = false;
    if (a < c.A) {
        const float* l = pb.cls + (((size_t)b * c.N) * c.A + a) * C;        // sample 0
        float v[C];
#pragma unroll
        for (int j = 0; j < C; ++j) v[j] = l[j];
        float mx = v[0];
#pragma unroll
        for (int j = 1; j < C; ++j) mx = fmaxf(mx, v[j]);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) { v[j] = expf(v[j] - mx); s += v[j]; }
        int best = 0; float bestp = v[0] / s;
        float* dc = pb.d_counts + ((size_t)b * c.A + a) * C;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float pj = v[j] / s;
            dc[j] = pj;
            if (j > 0 && pj > bestp) { bestp = pj; best = j; }                // tf.argmax: first maximum
        }
        keep = best != C - 1;
        pb.keep[(size_t)b * c.A + a] = keep ? 1 : 0;
    }
    const int n = __syncthreads_count(keep ? 1 : 0);
    if (threadIdx.x == 0) pb.block_counts[(size_t)b * nblocks + blockIdx.x] = n;
}

template <int C>
__global__ __launch_bounds__(POST_BLOCK) void val_fuse_kernel(PostCfg c, PostBuffers pb, int nblocks) {
    __shared__ int wave_off[POST_BLOCK / 64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = blockIdx.x * POST_BLOCK + tid;
    const int b = blockIdx.y;
    const bool keep = (a < c.A) && pb.keep[(size_t)b * c.A + a];
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_off[wave + 1] = __popcll(bal);
    if (tid == 0) wave_off[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int w = 1; w <= POST_BLOCK / 64; ++w) wave_off[w] += wave_off[w - 1];
    __syncthreads();
    if (!keep) return;
    const int slot = pb.block_counts[(size_t)b * nblocks + blockIdx.x] + wave_off[wave] + before;
    const size_t o = (size_t)b * c.A + slot;
    const float4 anc = reinterpret_cast<const float4*>(pb.anchors)[a];
    const float4 t = reinterpret_cast<const float4*>(pb.box)[((size_t)b * c.N) * c.A + a];
    float bx[4];
    decode_box(anc, t, bx);                                                   // box_utils.box_from_anchor_and_target (:149-168)
    const float* dc = pb.d_counts + ((size_t)b * c.A + a) * C;
    float top = dc[0];
#pragma unroll
    for (int j = 0; j < C; ++j) { pb.counts[o * C + j] = dc[j]; pb.score[o * C + j] = dc[j]; top = fmaxf(top, dc[j]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) pb.means[o * 4 + i] = bx[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) pb.covs[o * 16 + i] = 0.f;
    pb.ranking[o] = top;
    // vuhw_to_vuvu (box_utils.py:5-23)
    reinterpret_cast<float4*>(pb.corners)[o] = make_float4(bx[0] - bx[2] / 2.0f, bx[1] - bx[3] / 2.0f, bx[0] + bx[2] / 2.0f, bx[1] + bx[3] / 2.0f);
    pb.anchor_index[o] = a;
}

hipError_t launch_validation_post(const PostCfg& c, const PostBuffers& b, hipStream_t s) {
    const int nblocks = (c.A + POST_BLOCK - 1) / POST_BLOCK;
    dim3 grid(nblocks, c.B);
    if (c.C == 8) {
        hipLaunchKernelGGL(val_flag_kernel<8>, grid, dim3(POST_BLOCK), 0, s, c, b, nblocks);
        hipLaunchKernelGGL(post_scan_kernel, dim3(c.B), dim3(POST_BLOCK), 0, s, b, nblocks);
        hipLaunchKernelGGL(val_fuse_kernel<8>, grid, dim3(POST_BLOCK), 0, s, c, b, nblocks);
    } else if (c.C == 4) {
        hipLaunchKernelGGL(val_flag_kernel<4>, grid, dim3(POST_BLOCK), 0, s, c, b, nblocks);
        hipLaunchKernelGGL(post_scan_kernel, dim3(c.B), dim3(POST_BLOCK), 0, s, b, nblocks);
        hipLaunchKernelGGL(val_fuse_kernel<4>, grid, dim3(POST_BLOCK), 0, s, c, b, nblocks);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// joint-entropy ranking (:169-200): min-max normalised information gains, one block per image
// ------------------------------------------------------------------------------------------------
// log det of an SPD 4x4 (a posterior covariance) through its Cholesky factor, in double: the cofactor expansion in fp32 loses
// cond(Sigma) x 2^-24 of the determinant to cancellation, which the min-max normalisation over the image's M boxes then
// stretches to ~5e-3 of the ranking; log det = 2 sum log L_ii has no cancellation.  (M ~ 10^3 boxes per image: the cost is nil.)
__device__ __forceinline__ double logdet4_spd(const float* m) {
    double L[4][4];
    double ld = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = (double)m[i * 4 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) { L[i][i] = sqrt(s); ld += log(s); }      // log L_ii^2
            else L[i][j] = s / L[j][j];
        }
    }
    return ld;
}

__global__ __launch_bounds__(POST_BLOCK) void joint_entropy_kernel(PostCfg c, PostBuffers pb) {
    __shared__ float red[4][POST_BLOCK];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int M = pb.num_kept[b];
    const float two_pi_term = 2.0f + 2.0f * logf(2.0f * 3.14159265358979323846f);
    const float iv = c.iso_var;
    const float prior_ent = two_pi_term + 0.5f * logf(iv * iv * iv * iv);
    const float cat_prior = logf((float)c.C);          // entropy of the uniform prior score
    float gmin = INFINITY, gmax = -INFINITY, cmin = INFINITY, cmax = -INFINITY;
    for (int m = tid; m < M; m += POST_BLOCK) {
        const size_t o = (size_t)b * c.A + m;
        const float g = (float)((double)prior_ent - ((double)two_pi_term + 0.5 * logdet4_spd(pb.covs + o * 16)));
        float ce = 0.f;
        for (int j = 0; j < c.C; ++j) { const float p = pb.score[o * c.C + j]; ce -= p * logf(p); }
        const float cg = cat_prior - ce;
        gmin = fminf(gmin, g); gmax = fmaxf(gmax, g);
        cmin = fminf(cmin, cg); cmax = fmaxf(cmax, cg);
    }
    red[0][tid] = gmin; red[1][tid] = gmax; red[2][tid] = cmin; red[3][tid] = cmax;
    __syncthreads();
    for (int s = POST_BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] = fminf(red[0][tid], red[0][tid + s]);
            red[1][tid] = fmaxf(red[1][tid], red[1][tid + s]);
            red[2][tid] = fminf(red[2][tid], red[2][tid + s]);
            red[3][tid] = fmaxf(red[3][tid], red[3][tid + s]);
        }
        __syncthreads();
    }
    gmin = red[0][0]; gmax = red[1][0]; cmin = red[2][0]; cmax = red[3][0];
    const float gden = fmaxf(1.0f, gmax - gmin), cden = fmaxf(0.001f, cmax - cmin);
    for (int m = tid; m < M; m += POST_BLOCK) {
        const size_t o = (size_t)b * c.A + m;
        const float g = (float)((double)prior_ent - ((double)two_pi_term + 0.5 * logdet4_spd(pb.covs + o * 16)));
        float ce = 0.f;
        for (int j = 0; j < c.C; ++j) { const float p = pb.score[o * c.C + j]; ce -= p * logf(p); }
        const float cg = cat_prior - ce;
        pb.ranking[o] = (cg - cmin) / cden + (g - gmin) / gden;
    }
}

hipError_t launch_joint_entropy_rank(const PostCfg& c, const PostBuffers& b, hipStream_t s) {
    hipLaunchKernelGGL(joint_entropy_kernel, dim3(c.B), dim3(POST_BLOCK), 0, s, c, b);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K4: soft-NMS (NonMaxSuppressionV5). One wavefront per image walks the exact greedy order of
// the TF op: pop the best live candidate, decay it by every box selected since it was last
// examined (newest first), select iff unchanged, else re-queue.  The priority queue is replaced
// by a per-lane cached arg-max over a strided slice + a 6-step wave butterfly (same order:
// score desc, index asc).
// ------------------------------------------------------------------------------------------------
#define NMS_LDS_CAP 6144
#define NMS_MAX_OUT 512

__device__ __forceinline__ float nms_iou(const float4 bi, const float4 bj) {
    const float ymin_i = fminf(bi.x, bi.z), xmin_i = fminf(bi.y, bi.w);
    const float ymax_i = fmaxf(bi.x, bi.z), xmax_i = fmaxf(bi.y, bi.w);
    const float ymin_j = fminf(bj.x, bj.z), xmin_j = fminf(bj.y, bj.w);
    const float ymax_j = fmaxf(bj.x, bj.z), xmax_j = fmaxf(bj.y, bj.w);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.f || area_j <= 0.f) return 0.f;
    const float iy0 = fmaxf(ymin_i, ymin_j), ix0 = fmaxf(xmin_i, xmin_j);
    const float iy1 = fminf(ymax_i, ymax_j), ix1 = fminf(xmax_i, xmax_j);
    const float inter = fmaxf(iy1 - iy0, 0.f) * fmaxf(ix1 - ix0, 0.f);
    return inter / ((area_i + area_j) - inter);
}

// Wave-wide arg-max of (score desc, index asc) without LDS traffic: 4 DPP steps reduce each row of 16
// lanes (quad xor 1, xor 2, half-row mirror, row mirror), then the four row results are read out through
// SGPRs.  idx < 0 marks "no candidate".
__device__ __forceinline__ bool nms_better(float os, int oi, float bs, int bi) {
    return (oi >= 0) && (bi < 0 || os > bs || (os == bs && oi < bi));
}
template <int CTRL>
__device__ __forceinline__ void nms_dpp_step(float& bs, int& bi) {
    const float os = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bs), CTRL, 0xF, 0xF, false));
    const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xF, 0xF, false);
    if (nms_better(os, oi, bs, bi)) { bs = os; bi = oi; }
}
__device__ __forceinline__ void nms_wave_argmax(float& bs, int& bi) {
    nms_dpp_step<0xB1>(bs, bi);     // quad_perm [1,0,3,2]
    nms_dpp_step<0x4E>(bs, bi);     // quad_perm [2,3,0,1]
    nms_dpp_step<0x141>(bs, bi);    // row_half_mirror
    nms_dpp_step<0x140>(bs, bi);    // row_mirror
    float rs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bs), 0));
    int ri = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float os = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bs), r));
        const int oi = __builtin_amdgcn_readlane(bi, r);
        if (nms_better(os, oi, rs, ri)) { rs = os; ri = oi; }
    }
    bs = rs; bi = ri;
}

#define NMS_BOX_CAP 4096
__global__ __launch_bounds__(64) void nms_kernel(NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char nms_smem[];
    float4* s_box = reinterpret_cast<float4*>(nms_smem);                                  // [NMS_BOX_CAP]
    float4* s_selbox = s_box + NMS_BOX_CAP;                                               // [NMS_MAX_OUT]
    float* s_score = reinterpret_cast<float*>(s_selbox + NMS_MAX_OUT);                    // [NMS_LDS_CAP]
    int* s_begin = reinterpret_cast<int*>(s_score + NMS_LDS_CAP);                         // [NMS_LDS_CAP]
    const int b = blockIdx.x, lane = threadIdx.x;
    const int M = a.num_kept[b];
    const float4* gboxes = reinterpret_cast<const float4*>(a.corners) + (size_t)b * a.A;
    float* score = (M <= NMS_LDS_CAP) ? s_score : a.work_scores + (size_t)b * ((a.A + 511) & ~511);
    int* begin = (M <= NMS_LDS_CAP) ? s_begin : a.work_begin + (size_t)b * ((a.A + 511) & ~511);
    const bool box_lds = M <= NMS_BOX_CAP;
    int* sel = a.selected + (size_t)b * a.max_out;
    const float scale = a.sigma > 0.f ? -0.5f / a.sigma : 0.f;
    const bool soft = a.sigma > 0.f;
    const bool always_soft = a.variant == 1 && soft;
    const float NEG_INF = -INFINITY;

    // score[] is padded to a multiple of 64*8 with -inf; a candidate that leaves the queue is marked by
    // score = -inf as well (the op never queues a -inf score: it only admits score > score_threshold = -inf),
    // so the lane-local rescan is a branch-free strided max with 8 independent LDS reads in flight.
    const int Mpad = (M + 511) & ~511;
    for (int i = lane; i < Mpad; i += 64) {
        score[i] = i < M ? a.ranking[(size_t)b * a.A + i] : NEG_INF;
        if (i < M) {
            begin[i] = 0;
            if (box_lds) s_box[i] = gboxes[i];
        }
    }
    __syncthreads();

    // lane-local best over its strided slice
    float lbest = NEG_INF; int lidx = -1;
    auto rescan = [&]() {
        lbest = NEG_INF; lidx = -1;
        for (int i0 = lane; i0 < Mpad; i0 += 512) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = score[i0 + q * 64];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (v[q] > lbest) { lbest = v[q]; lidx = i0 + q * 64; }    // ascending i => lowest index wins ties
        }
    };
    rescan();
    int nsel = 0;
    while (nsel < a.max_out) {
        float bs = lbest; int bi = lidx;
        nms_wave_argmax(bs, bi);
        if (bi < 0) break;
        const int idx = bi;
        const float original = bs;
        const int beg = begin[idx];
        const float4 cb = box_lds ? s_box[idx] : gboxes[idx];
        // Decay by every box selected since the candidate was last examined, newest first (the TF op's
        // loop order; fp32 products do not commute in rounding).  Weights are evaluated lane-parallel, 64
        // selected boxes at a time with lane l <-> j = hi-1-l; a factor of exactly 1.0f (no overlap) leaves
        // the product bit-identical, so only the lanes with w != 1 enter the serial chain.
        float s = original;
        for (int hi = nsel; hi > beg; hi -= 64) {
            const int j = hi - 1 - lane;
            float w = 1.0f;
            if (j >= beg) {
                const float sim = nms_iou(cb, s_selbox[j]);
                if (sim != 0.f) w = (float)exp((double)(scale * (sim * sim)));
                if (!always_soft && !(sim <= a.iou_thr)) w = 0.f;
            }
            unsigned long long m = __ballot(w != 1.0f);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                s = s * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), l));
            }
        }
        const bool owner = (idx & 63) == lane;
        if (s == original) {
            if (lane == 0) { sel[nsel] = idx; s_selbox[nsel] = cb; }
            if (owner) score[idx] = NEG_INF;
            ++nsel;
        } else if (s > NEG_INF) {
            if (owner) { score[idx] = s; begin[idx] = nsel; }
        } else {
            if (owner) score[idx] = NEG_INF;
        }
        __syncthreads();
        if (owner) rescan();
    }
    if (lane == 0) a.num_selected[b] = nsel;
}

hipError_t launch_nms(const NmsArgs& a, hipStream_t s) {
    if (a.max_out > NMS_MAX_OUT) return hipErrorInvalidValue;
    constexpr int LDS = (NMS_BOX_CAP + NMS_MAX_OUT) * 16 + NMS_LDS_CAP * 8;
    static PerDeviceOnce once;
    bool& attr_set = *once.slot();
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(nms_kernel, dim3(a.B), dim3(64), LDS, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K5: cluster-and-fuse. One workgroup per (centre, image); the M x M affinity matrix of the
// reference (:214-215) is never built: IoU against the centre is evaluated on the fly.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float iou_plus1(const float4 p, const float4 q) {
    // box_utils.bbox_iou_vuvu (:117-146) including the (min-max+1) area quirk
    const float xi1 = fmaxf(p.y, q.y), yi1 = fmaxf(p.x, q.x);
    const float xi2 = fminf(p.w, q.w), yi2 = fminf(p.z, q.z);
    const float inter = fmaxf(xi2 - xi1 + 1.0f, 0.f) * fmaxf(yi2 - yi1 + 1.0f, 0.f);
    const float a1 = (p.y - p.w + 1.0f) * (p.x - p.z + 1.0f);
    const float a2 = (q.y - q.w + 1.0f) * (q.x - q.z + 1.0f);
    const float uni = (a1 + a2) - inter;
    return inter / (uni + 0.00001f);
}

#define CL_BLOCK 256
template <int C>
__global__ __launch_bounds__(CL_BLOCK) void cluster_fuse_kernel(ClusterArgs a) {
    __shared__ float red[CL_BLOCK];
    __shared__ float top_kl[4 * 3];              // the four waves' three best (kl, anchor) each
    __shared__ int top_ix[4 * 3];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (k >= a.num_selected[b]) return;
    const int M = a.num_kept[b];
    const int centre = a.selected[(size_t)b * a.max_out + k];
    const size_t base = (size_t)b * a.A;
    const float4* boxes = reinterpret_cast<const float4*>(a.corners) + base;
    const float4 cbox = boxes[centre];
    float cs[C];
    {
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) { cs[j] = a.counts[(base + centre) * C + j]; sum += cs[j]; }
#pragma unroll
        for (int j = 0; j < C; ++j) cs[j] = cs[j] / sum;
        // scipy.stats.entropy renormalises pk
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) s2 += cs[j];
#pragma unroll
        for (int j = 0; j < C; ++j) cs[j] = cs[j] / s2;
    }
    float psum[10], pmsum[4], ssum[C], csum[C];
#pragma unroll
    for (int q = 0; q < 10; ++q) psum[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) pmsum[q] = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) { ssum[j] = 0.f; csum[j] = 0.f; }
    float tk[3] = {INFINITY, INFINITY, INFINITY};
    int ti[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    int cnt = 0;
    const float* aff = (a.affinity && b == a.affinity_img) ? a.affinity + (size_t)k * a.A : nullptr;
    for (int i = tid; i < M; i += CL_BLOCK) {
        if (!((aff ? aff[i] : iou_plus1(boxes[i], cbox)) > a.thr)) continue;
        ++cnt;
        Mat4 cv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) cv.m[r][q] = a.covs[(base + i) * 16 + r * 4 + q];
        const Mat4 pr = inv_spd4(cv);
        int t = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q <= r; ++q) psum[t++] += pr.m[r][q];
        float mu[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mu[r] = a.means[(base + i) * 4 + r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) s += pr.m[r][q] * mu[q];
            pmsum[r] += s;
        }
        float row[C], rs = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) { row[j] = a.counts[(base + i) * C + j]; rs += row[j]; }
        float kl = 0.f, qs = 0.f;
        float sc[C];
#pragma unroll
        for (int j = 0; j < C; ++j) { sc[j] = row[j] / rs; qs += sc[j]; }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float q = sc[j] / qs;
            if (cs[j] > 0.f) kl += (q > 0.f) ? cs[j] * logf(cs[j] / q) : INFINITY;
            ssum[j] += sc[j];
            csum[j] += row[j];
        }
        // insert into this thread's sorted top-3 (kl asc, index asc)
        if (kl < tk[2] || (kl == tk[2] && i < ti[2])) {
            tk[2] = kl; ti[2] = i;
            if (tk[2] < tk[1] || (tk[2] == tk[1] && ti[2] < ti[1])) {
                float f = tk[1]; tk[1] = tk[2]; tk[2] = f; int g = ti[1]; ti[1] = ti[2]; ti[2] = g;
                if (tk[1] < tk[0] || (tk[1] == tk[0] && ti[1] < ti[0])) {
                    f = tk[0]; tk[0] = tk[1]; tk[1] = f; g = ti[0]; ti[0] = ti[1]; ti[1] = g;
                }
            }
        }
    }
    // ---- block reductions (round 4): wave-wide butterflies + one LDS exchange of the four waves' partials -- two barriers for all 15
    // sums (the tree reduction per value they replace: 140), the global top-3 as three rounds of wave-wide arg-min + a 12-entry merge
    // (replaced: one thread scanning 768 entries).  Sums are re-associated (fp32 round-off); the top-3 selection is exact.
    const int wave = tid >> 6, lane = tid & 63;
    auto wsum = [](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    {
        float vals[15];
#pragma unroll
        for (int q = 0; q < 10; ++q) vals[q] = psum[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) vals[10 + q] = pmsum[q];
        vals[14] = (float)cnt;                                  // (at most A < 2^24: exact)
#pragma unroll
        for (int q = 0; q < 15; ++q) {
            const float r = wsum(vals[q]);
            if (lane == 0) red[wave * 16 + q] = r;
        }
    }
    // this wave's three best (kl asc, index asc) of its threads' sorted candidates
    {
        int p = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float ck = p == 0 ? tk[0] : p == 1 ? tk[1] : p == 2 ? tk[2] : INFINITY;
            const int ci = p == 0 ? ti[0] : p == 1 ? ti[1] : p == 2 ? ti[2] : 0x7fffffff;
            float bk = ck; int bi = ci;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ok = __shfl_xor(bk, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (oi != 0x7fffffff && (bi == 0x7fffffff || ok < bk || (ok == bk && oi < bi))) { bk = ok; bi = oi; }
            }
            if (lane == 0) { top_kl[wave * 3 + r] = bk; top_ix[wave * 3 + r] = bi; }
            if (bi != 0x7fffffff && ci == bi) ++p;              // (an anchor belongs to exactly one thread: the winner advances)
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 10; ++q) psum[q] = (red[q] + red[16 + q]) + (red[32 + q] + red[48 + q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) pmsum[q] = (red[10 + q] + red[26 + q]) + (red[42 + q] + red[58 + q]);
    const int total = (int)((red[14] + red[30]) + (red[46] + red[62]));
    if (total <= 3) {                                           // (block-uniform)
        __syncthreads();                                        // every thread has read the first exchange
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float r0 = wsum(ssum[j]), r1 = wsum(csum[j]);
            if (lane == 0) { red[wave * 16 + j] = r0; red[64 + wave * 16 + j] = r1; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < C; ++j) {
            ssum[j] = (red[j] + red[16 + j]) + (red[32 + j] + red[48 + j]);
            csum[j] = (red[64 + j] + red[80 + j]) + (red[96 + j] + red[112 + j]);
        }
    }
    if (tid != 0) return;
    const size_t ob = (size_t)b * a.max_out + k;
    // fused Gaussian: cov = inv(sum prec), mean = cov * sum(prec*mean)  (:321-331), x70 (:361)
    Mat4 ps;
    {
        int t = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q <= r; ++q) { ps.m[r][q] = psum[t]; ps.m[q][r] = psum[t]; ++t; }
    }
    const Mat4 fc = inv_spd4(ps);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += fc.m[r][q] * pmsum[q];
        a.out_means[ob * 4 + r] = s;
#pragma unroll
        for (int q = 0; q < 4; ++q) a.out_covs[ob * 16 + r * 4 + q] = fc.m[r][q] * 70.0f;
    }
    if (total > 3) {
        // global top-3 over the per-thread candidates (:338-349)
        float bk[3] = {INFINITY, INFINITY, INFINITY};
        int bx[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        for (int t = 0; t < 4 * 3; ++t) {                       // the four waves' three best each
            const float kl = top_kl[t]; const int ix = top_ix[t];
            if (ix == 0x7fffffff) continue;
            if (kl < bk[2] || (kl == bk[2] && ix < bx[2])) {
                bk[2] = kl; bx[2] = ix;
                if (bk[2] < bk[1] || (bk[2] == bk[1] && bx[2] < bx[1])) {
                    float f = bk[1]; bk[1] = bk[2]; bk[2] = f; int g = bx[1]; bx[1] = bx[2]; bx[2] = g;
                    if (bk[1] < bk[0] || (bk[1] == bk[0] && bx[1] < bx[0])) {
                        f = bk[0]; bk[0] = bk[1]; bk[1] = f; g = bx[0]; bx[0] = bx[1]; bx[1] = g;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) { ssum[j] = 0.f; csum[j] = 0.f; }
        for (int t = 0; t < 3; ++t) {
            float row[C], rs = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) { row[j] = a.counts[(base + bx[t]) * C + j]; rs += row[j]; }
#pragma unroll
            for (int j = 0; j < C; ++j) { ssum[j] += row[j] / rs; csum[j] += row[j]; }
        }
    }
    const float denom = (float)(total > 3 ? 3 : total);
#pragma unroll
    for (int j = 0; j < C; ++j) {
        a.out_scores[ob * C + j] = ssum[j] / denom;
        a.out_counts[ob * C + j] = csum[j];
    }
}

hipError_t launch_cluster_fuse(const ClusterArgs& a, hipStream_t s) {
    dim3 grid(a.max_out, a.B);
    if (a.C == 8) hipLaunchKernelGGL(cluster_fuse_kernel<8>, grid, dim3(CL_BLOCK), 0, s, a);
    else if (a.C == 4) hipLaunchKernelGGL(cluster_fuse_kernel<4>, grid, dim3(CL_BLOCK), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

__global__ void iou_matrix_kernel(const float4* boxes, int M, float* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < M) out[(size_t)i * M + j] = iou_plus1(boxes[i], boxes[j]);
}

// ------------------------------------------------------------------------------------------------
// Detection records for the path's one multi-GPU exchange: per image K padded rows of
// W = 1 + 4 + 16 + 2C floats -- [valid, mean (v,u,h,w), covariance row-major, score[C], counts[C]] -- zero beyond the image's
// detection count.  The layout of distributed.pack_records, written by one kernel instead of five torch ops.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_records_kernel(const int32_t* __restrict__ num, const float* __restrict__ scores,
                                                           const float* __restrict__ means, const float* __restrict__ covs,
                                                           const float* __restrict__ counts, float* __restrict__ rec, int B, int K, int C) {
    const int W = 21 + 2 * C;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * K * W) return;
    const int w = (int)(i % W);
    const long row = i / W;                      // b * K + k
    const int b = (int)(row / K), k = (int)(row % K);
    float v = 0.f;
    if (k < num[b]) {
        if (w == 0) v = 1.f;
        else if (w < 5) v = means[row * 4 + (w - 1)];
        else if (w < 21) v = covs[row * 16 + (w - 5)];
        else if (w < 21 + C) v = scores[row * C + (w - 21)];
        else v = counts[row * C + (w - 21 - C)];
    }
    rec[i] = v;
}

hipError_t launch_pack_records(const int32_t* num, const float* scores, const float* means, const float* covs, const float* counts,
                               float* rec, int B, int K, int C, hipStream_t s) {
    const long n = (long)B * K * (21 + 2 * C);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, num, scores, means, covs, counts, rec, B, K, C);
    return hipGetLastError();
}

hipError_t launch_iou_matrix(const float* corners, int M, float* out, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(iou_matrix_kernel, dim3((M + 255) / 256, M), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(corners), M, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Frame preprocessing on the device (the dataset handlers' work between image decode and
// sample_dict['image_normalized']): uint8 RGB -> [optional KITTI bilinear resize + centred crop / zero pad]
// -> float32, mean subtraction, RGB -> BGR  (bdd_dataset_handler.py:128-139, kitti_dataset_handler.py:120-148).
// One thread per output pixel; fp32 operation order identical to oracle/preprocess.py (this file is built
// with -ffp-contract=off), so results are bit-exact.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_kernel(PreprocArgs a) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, b = blockIdx.z;
    if (x >= a.W) return;
    const int ry = y - a.pad_y + a.crop_y, rx = x - a.pad_x + a.crop_x;     // coordinates in the (resized) source
    float v[3] = {0.f, 0.f, 0.f};                                            // zero padding (before mean subtraction)
    const bool inside = y >= a.pad_y && x >= a.pad_x && ry < a.rh && rx < a.rw && ry >= 0 && rx >= 0 &&
                        y - a.pad_y < a.vis_h && x - a.pad_x < a.vis_w;
    if (inside) {
        const uint8_t* src = a.src + (size_t)b * a.sh * a.sw * 3;
        if (!a.resize) {
            const uint8_t* p = src + ((size_t)ry * a.sw + rx) * 3;
            v[0] = (float)p[0]; v[1] = (float)p[1]; v[2] = (float)p[2];
        } else {
            const float fy = ((float)ry + 0.5f) * a.scale_y - 0.5f;
            const float fx = ((float)rx + 0.5f) * a.scale_x - 0.5f;
            const float fy0 = floorf(fy), fx0 = floorf(fx);
            const float ly = fy - fy0, lx = fx - fx0;
            const int y0 = min(max((int)fy0, 0), a.sh - 1), y1 = min(max((int)ceilf(fy), 0), a.sh - 1);
            const int x0 = min(max((int)fx0, 0), a.sw - 1), x1 = min(max((int)ceilf(fx), 0), a.sw - 1);
            const uint8_t* p00 = src + ((size_t)y0 * a.sw + x0) * 3;
            const uint8_t* p01 = src + ((size_t)y0 * a.sw + x1) * 3;
            const uint8_t* p10 = src + ((size_t)y1 * a.sw + x0) * 3;
            const uint8_t* p11 = src + ((size_t)y1 * a.sw + x1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float tl = (float)p00[c], tr = (float)p01[c], bl = (float)p10[c], br = (float)p11[c];
                const float top = tl + (tr - tl) * lx;
                const float bot = bl + (br - bl) * lx;
                v[c] = top + (bot - top) * ly;
            }
        }
    }
    float* o = a.dst + (((size_t)b * a.H + y) * a.W + x) * 3;
    o[0] = v[2] - a.mean[2]; o[1] = v[1] - a.mean[1]; o[2] = v[0] - a.mean[0];
}

hipError_t launch_preprocess(const PreprocArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(preprocess_kernel, dim3((a.W + 255) / 256, a.H, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}
