// Host side of libbayesod_hip.so: handle, HBM plan, weight folding/packing, launch sequencing and
// the C ABI of include/bayesod.h.  Mirrors, stage for stage, the call stack of
// src/retina_net/experiments/run_inference.py:137-149 -> inference_utils.bayes_od_inference
// -> RetinaNetModel.call.
#include "../../include/bayesod.h"
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <tuple>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <dlfcn.h>

namespace {

constexpr float BN_EPS = 1e-3f;   // keras BatchNormalization default

thread_local std::string g_create_error;

struct HostTensor { std::vector<int64_t> shape; std::vector<float> data; };

struct Plane {               // NHWC view (bf16, or fp32 in fp32 precision mode) with a 1-pixel zero border
    char* d = nullptr;
    int64_t base = 0;        // pixel offset of this view inside d
    int64_t bstride = 0;     // pixels between consecutive batch items
    int h = 0, w = 0, C = 0, pitch = 0;
};

struct PackedConv { char* w = nullptr; float* bias = nullptr; int cout = 0, cout_pad = 0, taps = 0, kw = 0, cin = 0; };

struct Op {
    enum Kind { STEM, POOL, CONV } kind;
    ConvArgs conv;
    bool is_head3x3 = false;
    double flops = 0;
    std::string name;
    std::string wname[3], bnname[3];     // per group: conv / batch-norm layer names (training: parameter lookup)
    bool same_geom = false;              // 3x3 stride-1 SAME with identically laid-out input / output planes (training: input gradient as a convolution)
    // The last tower layers exist in two flavours when the MC aggregation is fused into their epilogue: FLAVOUR_RAW writes the
    // per-sample head outputs [B,N,A,.] (bod_forward, sample sharding, parity tests), FLAVOUR_AGG reduces them over the
    // samples inside the tile (bod_infer).  0 = the op belongs to both.
    int flavour = 0;
    int head[3] = {-1, -1, -1};          // per group: BOD_HEAD_* whose raw buffer the fused 1x1 writes (raw buffers are allocated lazily)
    bool hx_pyramid = false;             // f16mx: this launch (the first tower layer) reads the pyramid as hx rows -- converted right in front of it
};
enum { FLAVOUR_BOTH = 0, FLAVOUR_RAW = 1, FLAVOUR_AGG = 2 };

uint16_t f2bf(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; std::memcpy(&f, &u, 4); return f; }

// ---- f16mx precision: the "hx" row of conv_igemm.hip (header there), packed on the host for weights (and for the test entry
// point's activations): float -> IEEE half, round to nearest even, subnormals kept (the f16 MFMA honours them:
// profiles/round5_mx_probe.txt)
uint16_t f2h(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u >= 0x47800000u) return (uint16_t)(sign | (u > 0x7F800000u ? 0x7E00u : 0x7C00u));      // >= 65536: inf (NaN stays NaN)
    if (u < 0x38800000u) {                         // below 2^-14: subnormal half = round(|f| * 2^24)
        float a; std::memcpy(&a, &u, 4);
        return (uint16_t)(sign | (uint32_t)std::nearbyint(a * 16777216.0f));
    }
    u += 0xFFFu + ((u >> 13) & 1u);                // round the 13 dropped mantissa bits to nearest even
    return (uint16_t)(sign | ((u - 0x38000000u) >> 13));      // (a mantissa carry runs into the exponent, up to inf, correctly)
}
float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    float f;
    if (e == 0) { f = (float)m * (1.0f / 16777216.0f); uint32_t u; std::memcpy(&u, &f, 4); u |= sign; std::memcpy(&f, &u, 4); return f; }
    const uint32_t u = sign | ((e == 31 ? 255u : e + 112u) << 23) | (m << 13);
    std::memcpy(&f, &u, 4);
    return f;
}
// e2m3 (OCP fp6, bias 1): round to nearest even, saturating at 7.5
uint32_t f2e2m3(float x) {
    const uint32_t sign = std::signbit(x) ? 32u : 0u;
    const float a = std::fabs(x);
    uint32_t code;
    if (a < 1.0f) code = (uint32_t)std::nearbyint(a * 8.0f);
    else if (a < 2.0f) code = 8u + (uint32_t)std::nearbyint((a - 1.0f) * 8.0f);
    else if (a < 4.0f) code = 16u + (uint32_t)std::nearbyint((a - 2.0f) * 4.0f);
    else code = 24u + (uint32_t)std::nearbyint((std::min(a, 7.5f) - 4.0f) * 2.0f);
    return sign | std::min(code, 31u);
}
float e2m3_to_f(uint32_t c) {
    const uint32_t e = (c >> 3) & 3u, m = c & 7u;
    const float r = e == 0 ? m / 8.0f : std::ldexp(1.0f + m / 8.0f, (int)e - 1);
    return (c & 32u) ? -r : r;
}
// One hx row: C channels (multiple of 64) -> 4 C bytes.  Per 64 channels: H chunk = 64 f16 hi; X chunk = four 32-byte slots (m, b),
// pieces at X offsets 64m + 16b and 64m + 32 + 16b; a slot = the 16 channels 32m + 8 g4 + 4b + r as 32 e2m3 elements under one scale
// 2^(eb - 127), eb = biased exponent of (max(|v|, 2^-14) * 16/15) - 2: element 2k = hi6 (weights: lo6'), 2k + 1 = lo6' (weights:
// hi6), lo' = (v - hi) * 2^11; scale byte at byte 28 of the slot (weights: eb - 11, which undoes the 2^11 of both cross products).
void pack_hx_row(const float* v, int C, uint8_t* dst, bool weights) {
    std::memset(dst, 0, (size_t)C * 4);
    for (int q = 0; q < C / 64; ++q) {
        uint8_t* H = dst + (size_t)q * 256;
        uint8_t* X = H + 128;
        for (int c = 0; c < 64; ++c) { const uint16_t hb = f2h(v[q * 64 + c]); std::memcpy(H + 2 * c, &hb, 2); }
        for (int m = 0; m < 2; ++m)
            for (int b = 0; b < 2; ++b) {
                float hi[16], lo[16], mx = 6.103515625e-05f;
                for (int k = 0; k < 16; ++k) {
                    const float x = v[q * 64 + 32 * m + 8 * (k >> 2) + 4 * b + (k & 3)];
                    hi[k] = h2f(f2h(x)); lo[k] = (x - hi[k]) * 2048.0f;
                    mx = std::max(mx, std::fabs(x));
                }
                const float mxs = mx * 1.0666667f;
                uint32_t ub; std::memcpy(&ub, &mxs, 4);
                const uint32_t eb = (ub >> 23) - 2u;
                const float inv = std::ldexp(1.0f, 127 - (int)eb);
                uint8_t slot[32] = {0};
                for (int k = 0; k < 16; ++k) {
                    const uint32_t h6 = f2e2m3(hi[k] * inv), l6 = f2e2m3(lo[k] * inv);
                    const uint32_t e0 = weights ? l6 : h6, e1 = weights ? h6 : l6;
                    const int bit0 = 12 * k;
                    for (int t = 0; t < 6; ++t) {
                        if ((e0 >> t) & 1u) slot[(bit0 + t) >> 3] |= (uint8_t)(1u << ((bit0 + t) & 7));
                        if ((e1 >> t) & 1u) slot[(bit0 + 6 + t) >> 3] |= (uint8_t)(1u << ((bit0 + 6 + t) & 7));
                    }
                }
                slot[28] = (uint8_t)(weights ? eb - 11u : eb);
                std::memcpy(X + 64 * m + 16 * b, slot, 16);
                std::memcpy(X + 64 * m + 32 + 16 * b, slot + 16, 16);
            }
    }
}
// ... and back (test entry point: the epilogue's hx output): value = hi + lo' * 2^-11 with lo' from the slot's odd elements
void unpack_hx_row(const uint8_t* src, int C, float* v) {
    for (int q = 0; q < C / 64; ++q) {
        const uint8_t* H = src + (size_t)q * 256;
        const uint8_t* X = H + 128;
        for (int m = 0; m < 2; ++m)
            for (int b = 0; b < 2; ++b) {
                uint8_t slot[32];
                std::memcpy(slot, X + 64 * m + 16 * b, 16);
                std::memcpy(slot + 16, X + 64 * m + 32 + 16 * b, 16);
                const float sc = std::ldexp(1.0f, (int)slot[28] - 127);
                for (int k = 0; k < 16; ++k) {
                    const int c = 32 * m + 8 * (k >> 2) + 4 * b + (k & 3);
                    uint16_t hb; std::memcpy(&hb, H + 2 * c, 2);
                    uint32_t l6 = 0;
                    for (int t = 0; t < 6; ++t) l6 |= (uint32_t)((slot[(12 * k + 6 + t) >> 3] >> ((12 * k + 6 + t) & 7)) & 1u) << t;
                    v[q * 64 + c] = h2f(hb) + e2m3_to_f(l6) * sc * (1.0f / 2048.0f);
                }
            }
    }
}

// e2m1 (OCP MX fp4): round to nearest even, saturating at 6; code = sign(1) exp(2) mant(1), bias 1, subnormal step 0.5
uint32_t f2e2m1(float x) {
    const uint32_t s = std::signbit(x) ? 8u : 0u;
    float a = std::fabs(x);
    if (!(a == a)) return s;
    if (a >= 6.0f) return s | 7u;
    static const float grid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    uint32_t best = 0;
    for (uint32_t c = 0; c < 7; ++c) {
        const float mid = 0.5f * (grid[c] + grid[c + 1]);
        if (a > mid || (a == mid && (c & 1u))) best = c + 1;          // ties to the even code
    }
    return s | best;
}
float e2m1_to_f(uint32_t c) {
    static const float grid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    return (c & 8u) ? -grid[c & 7u] : grid[c & 7u];
}
// One h4 row (f16mx4): C = 256 channels -> 1 024 bytes in eight 128-byte chunks: [H0 H1 X0 H2 H3 X1 S -].  Hq = the 64 f16 hi of channels
// 64q..; Xx = the cross-term elements of channels 128x..: piece 2 ks + half (16 bytes) = the MX block of the 16 channels
// 128x + 32 ks + 8 g4 + 4 half + r as 32 e2m1 nibbles, nibble 2k = hi4 (weights: lo4'), 2k + 1 = lo4' (weights: hi4), k = 4 g4 + r,
// lo' = (v - hi) * 2^11, under the block's scale 2^(eb - 127), eb = biased exponent of (max(|hi|, 2^-14) * 4/3) - 2; S: byte
// 8x + 4 half + ks = eb (weights: eb - 11).
void pack_h4_row(const float* v, int C, uint8_t* dst, bool weights) {
    std::memset(dst, 0, (size_t)C * 4);
    for (int q = 0; q < C / 64; ++q) {
        uint8_t* H = dst + (size_t)(3 * (q >> 1) + (q & 1)) * 128;
        for (int c = 0; c < 64; ++c) { const uint16_t hb = f2h(v[q * 64 + c]); std::memcpy(H + 2 * c, &hb, 2); }
    }
    for (int x = 0; x < C / 128; ++x)
        for (int ks = 0; ks < 4; ++ks)
            for (int b = 0; b < 2; ++b) {
                float hi[16], lo[16], mx = 6.103515625e-05f;
                for (int k = 0; k < 16; ++k) {
                    const float xv = v[128 * x + 32 * ks + 8 * (k >> 2) + 4 * b + (k & 3)];
                    hi[k] = h2f(f2h(xv)); lo[k] = (xv - hi[k]) * 2048.0f;
                    mx = std::max(mx, std::fabs(hi[k]));
                }
                const float mxs = mx * 1.3333334f;
                uint32_t ub; std::memcpy(&ub, &mxs, 4);
                const uint32_t eb = (ub >> 23) - 2u;
                const float inv = std::ldexp(1.0f, 127 - (int)eb);
                uint8_t* piece = dst + (size_t)(3 * x + 2) * 128 + (2 * ks + b) * 16;
                for (int k = 0; k < 16; ++k) {
                    const uint32_t h4 = f2e2m1(hi[k] * inv), l4 = f2e2m1(lo[k] * inv);
                    piece[k] = (uint8_t)(weights ? (l4 | (h4 << 4)) : (h4 | (l4 << 4)));
                }
                dst[6 * 128 + 8 * x + 4 * b + ks] = (uint8_t)(weights ? eb - 11u : eb);
            }
}
void unpack_h4_row(const uint8_t* src, int C, float* v) {
    for (int x = 0; x < C / 128; ++x)
        for (int ks = 0; ks < 4; ++ks)
            for (int b = 0; b < 2; ++b) {
                const uint8_t* piece = src + (size_t)(3 * x + 2) * 128 + (2 * ks + b) * 16;
                const float sc = std::ldexp(1.0f, (int)src[6 * 128 + 8 * x + 4 * b + ks] - 127);
                for (int k = 0; k < 16; ++k) {
                    const int c = 128 * x + 32 * ks + 8 * (k >> 2) + 4 * b + (k & 3);
                    const int q = c >> 6;
                    uint16_t hb; std::memcpy(&hb, src + (size_t)(3 * (q >> 1) + (q & 1)) * 128 + 2 * (c & 63), 2);
                    v[c] = h2f(hb) + e2m1_to_f((uint32_t)piece[k] >> 4) * sc * (1.0f / 2048.0f);
                }
            }
}

}  // namespace

struct bod_context {
    bod_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    std::map<const void*, size_t> alloc_bytes;
    int64_t device_bytes = 0;
    struct TrainState* train = nullptr;                 // training mode (train_impl.inc)

    // geometry
    int sh = 0, sw = 0, ph = 0, pw = 0;                 // stem / pool output
    int ch[6] = {0}, cw[6] = {0};                       // stage 2..5 sizes (index = stage)
    int nlev = 0; int lh[8] = {0}, lw[8] = {0};         // pyramid levels p3..p7
    int64_t lvl_off[8] = {0};                           // padded pixel offset of each level
    int64_t lvl_p0[8] = {0};                            // dense pixel offset of each level
    int64_t Ppad = 0; int P = 0, A = 0;

    // weights
    std::map<std::string, HostTensor> host_w;           // "name/kind"
    std::map<std::string, PackedConv> packed;
    float* stem_w = nullptr; float* stem_b = nullptr;
    bool weights_ready = false, anchors_ready = false, forward_done = false, posterior_done = false,
         nms_done = false, cluster_done = false;

    // activations
    float* d_images = nullptr;
    float* splitk_partial = nullptr; size_t splitk_elems = 0;        // fp32 partial sums of the split-K layers
    uint8_t* d_frames_u8 = nullptr; size_t frames_u8_cap = 0;     // staging for bod_upload_frames_u8 (plain hipMalloc, grows)
    // pipelined input path (bod_upload_frames_u8_async): two image buffers (0 = d_images), their uint8 staging, a copy stream;
    // ev_img_ready: preprocess of a buffer finished (the forward waits on it); ev_img_free: the stem that read it finished
    // (the next upload into the same buffer waits on it)
    float* d_images_b[2] = {nullptr, nullptr}; uint8_t* d_u8_b[2] = {nullptr, nullptr}; size_t u8_cap_b[2] = {0, 0};
    hipStream_t copy = nullptr;
    hipEvent_t ev_img_ready[2] = {nullptr, nullptr}, ev_img_free[2] = {nullptr, nullptr};
    bool img_ready_pending[2] = {false, false}, img_free_pending[2] = {false, false};
    int cur_img_buf = -1;
    char* stem_out = nullptr;
    int es = 2;                                          // bytes per activation / weight CHANNEL (2 = bf16; 4 = fp32, or a (hi, lo) bf16 pair)
    bool split = false;                                  // bf16x3 precision: (hi, lo) bf16 pairs, three MFMA products (conv_igemm.hip)
    char* pyr_hx = nullptr;                              // f16mx: the pyramid as hx rows [B][Ppad][1 KiB] (pairs_to_hx_kernel in front of the first tower layer)
    int pyr_fmt = 1;                                     // format of `pyr_hx` (1 hx, 2 h4)
    int plan_mx = 0;                                     // ... and the plan really runs them that way (BOD_TOWER_MX=0 / BOD_CONV_XREUSE=0: bf16x3 towers)
    int mx = 0;                                          // 1 = f16mx, 2 = f16mx4 (the cross terms as ONE e2m1 product of twice the channels; h4 rows).  f16mx precision: bf16x3 everywhere but the head towers, which run one f16 + half a block-scaled e2m3 product per multiplication (conv_igemm.hip header)
    Plane pyramid;                                       // all levels, [B][Ppad][256]
    char* head_act[3][2] = {{nullptr}};              // [B][N][Ppad][256]
    char* head_act_t[3][4] = {{nullptr}};            // training: one buffer per tower layer
    float* raw[3] = {nullptr};                           // cls [B,N,P,9C] box [B,N,P,36] cov [B,N,P,90]
    // MC aggregation fused into the last tower layers (ConvGroup.agg_kind): per-anchor statistics instead of raw[]
    float* agg[3] = {nullptr};                           // sum softmax [B,A,C], Welford box [B,A,16], sum cov params [B,A,10]
    bool agg_plan = false;                               // the plan holds FLAVOUR_AGG ops
    bool plan_fused_out = false, plan_xreuse = false, plan_xreuse0 = false;     // bod_plan_info
    bool raw_valid = false, agg_valid = false;           // which of the two the last forward produced
    uint64_t last_seed = 0; uint32_t last_first_image = 0;
    std::map<std::string, RowEnt*> tables;
    struct XrTable { RowEnt* rows = nullptr; int2* ext = nullptr; int m = 0; };
    std::map<std::string, XrTable> xr_tables;          // row-reuse tilings of plane -> plane 3x3 layers (add_conv)
    std::vector<Op> ops;
    const float* cur_images = nullptr;

    // post
    float* d_anchors = nullptr;
    PostBuffers pb{};
    float* nms_scores = nullptr; int32_t* nms_begin = nullptr;
    // detection records are double-buffered ("slots") so the latency-bound NMS + cluster-fuse of one
    // batch can run on a side stream underneath the next batch's convolutions
    int32_t* nms_sel_s[2] = {nullptr, nullptr}; int32_t* nms_nsel_s[2] = {nullptr, nullptr};
    float* out_scores_s[2] = {nullptr, nullptr}; float* out_means_s[2] = {nullptr, nullptr};
    float* out_covs_s[2] = {nullptr, nullptr}; float* out_counts_s[2] = {nullptr, nullptr};
    int32_t* nms_sel = nullptr; int32_t* nms_nsel = nullptr;          // = current slot
    float* out_scores = nullptr; float* out_means = nullptr; float* out_covs = nullptr; float* out_counts = nullptr;
    int slot = 0;
    hipStream_t side = nullptr;
    // ---- CU-partitioned pipeline overlap (bod_config.pipeline_overlap; bod_infer_async only).  The memory-bound front of batch i+1
    // (stem, backbone, FPN) runs on its own stream, masked to the last `ov_front_slots` CU slots of every XCD, while the MFMA-bound
    // back of batch i (fan-out layer, towers, posterior) runs on the other slots: `back` = the main stream of such a call (h->stream
    // points at it for the call's duration), `full` = the unmasked main stream every other entry point uses.  The only tensor that
    // crosses the partition is the pyramid, double-buffered (pyr_d); ev_front_done[p]: front of the batch with parity p finished,
    // ev_l0_done[p]: the fan-out layer that read pyramid p finished (the front two batches later waits for it).
    int overlap_mode = 0;                               // 0 off, 1 CU-masked streams, 2 plain streams (A/B)
    int ov_front_slots = 0, ov_slots = 0;
    hipStream_t full = nullptr, front = nullptr, back = nullptr;
    hipEvent_t ev_front_done[2] = {nullptr, nullptr}, ev_l0_done[2] = {nullptr, nullptr}, ev_join = nullptr;
    bool l0_pending[2] = {false, false};
    bool ov_dirty = false;                              // front / back hold work no `full`-stream call has been ordered behind yet
    char* pyr_d[2] = {nullptr, nullptr};                // pyr_d[0] == pyramid.d; [1] only on overlap handles
    int fwd_parity = 0, pyr_last = 0;
    int first_head_op = 0;                              // index of the first head launch in ops: everything before it is the front
    bool in_overlap_call = false;
    int n_cu = 256;                                     // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    hipEvent_t ev_posterior = nullptr; hipEvent_t ev_done[2] = {nullptr, nullptr};
    int dev_op_lo = 0, dev_op_hi = 1 << 30;                   // BOD_FORWARD_OPS=lo:hi (development: tests/tools/selfcheck_probe.py): forward runs ops [lo, hi) only
    hipStream_t done_stream[2] = {nullptr, nullptr};          // the stream that finished a slot's records (bod_infer_async): a ticket's gather follows on it
    bool side_pending[2] = {false, false};
    char* host_stage[2] = {nullptr, nullptr};     // pinned host copy of a slot's records (filled on the side stream)
    float* rec_send = nullptr; float* rec_recv = nullptr; size_t rec_recv_elems = 0;      // bod_gather_detections: packed records, gathered blocks
    hipEvent_t ev_gather = nullptr; hipStream_t gather_stream = nullptr;                  // last use of rec_send / rec_recv and the stream it went to
    void select_slot(int sidx) {
        slot = sidx;
        nms_sel = nms_sel_s[sidx]; nms_nsel = nms_nsel_s[sidx];
        out_scores = out_scores_s[sidx]; out_means = out_means_s[sidx];
        out_covs = out_covs_s[sidx]; out_counts = out_counts_s[sidx];
    }
    float* iou_scratch = nullptr; int64_t iou_cap = 0;
    float* affinity = nullptr; int affinity_img = -1;   // bod_set_affinity: centre columns of a caller-supplied affinity matrix (one-shot)

    // profiling
    bool profiling = false; int prof_which = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_head, ev_post;
    double prof_flops = 0;

    bod_status fail(bod_status s, const char* fmt, ...) {
        char buf[1024];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return s;
    }
    template <typename T> bod_status dalloc(T** p, size_t n_elems, bool zero = true) {
        void* q = nullptr;
        const size_t bytes = std::max<size_t>(n_elems * sizeof(T), 256);
        hipError_t e = hipMalloc(&q, bytes);
        if (e != hipSuccess) return fail(BOD_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        if (zero) {
            e = hipMemsetAsync(q, 0, bytes, stream);
            if (e != hipSuccess) return fail(BOD_ERR_HIP, "hipMemset: %s", hipGetErrorString(e));
        }
        allocs.push_back(q);
        alloc_bytes[q] = bytes;
        device_bytes += (int64_t)bytes;
        *p = reinterpret_cast<T*>(q);
        return BOD_OK;
    }
};

#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return (h)->fail(BOD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                               \
    } while (0)
#define BODCHK(expr) do { bod_status _s = (expr); if (_s != BOD_OK) return _s; } while (0)

namespace {

const char* kHeadPrefix[3] = {"pyramid_classification", "pyramid_regression", "pyramid_cov"};
const int kHeadConvs[3] = {4, 3, 4};     // RegHeader.call uses 3 towers convs (multitask_headers.py:209-230)

// CU mask of hipExtStreamCreateWithCUMask for CU slots [lo, hi) of every XCD.  On MI355X (8 XCDs x 32 CUs) mask bit i selects slot
// i / 8 of XCD i % 8 (tests/tools/cu_mask_probe.hip; an XCD without any bit set keeps all its CUs), so a slot range is symmetric
// over the XCDs and workgroup b still lands on XCD b % 8 -- what the tower kernel's XCD-aware tile order assumes.
void cu_slot_mask(int lo, int hi, uint32_t mask[8]) {
    for (int w = 0; w < 8; ++w) mask[w] = 0u;
    for (int slot = lo; slot < hi; ++slot)
        for (int xcd = 0; xcd < 8; ++xcd) { const int bit = slot * 8 + xcd; mask[bit >> 5] |= 1u << (bit & 31); }
}

int same_pad_before(int in, int k, int s) {
    const int out = (in + s - 1) / s;
    const int total = std::max((out - 1) * s + k - in, 0);
    return total / 2;
}

bod_status new_plane(bod_context* h, Plane* p, int B, int hh, int ww, int C) {
    p->h = hh; p->w = ww; p->C = C; p->pitch = ww + 2; p->base = 0;
    p->bstride = (int64_t)(hh + 2) * (ww + 2);
    return h->dalloc(&p->d, (size_t)B * p->bstride * C * h->es);
}

// row table for plane -> plane convolutions. org_* = padded coordinate of the window origin of
// output (0,0); res optional (nearest-upsampled when its size differs).
bod_status make_table(bod_context* h, const std::string& key, int B, const Plane& in, const Plane& out,
                      int stride, int org_y, int org_x, const Plane* res, const RowEnt** tbl) {
    auto it = h->tables.find(key);
    if (it != h->tables.end()) { *tbl = it->second; return BOD_OK; }
    std::vector<RowEnt> rows((size_t)B * out.h * out.w);
    size_t r = 0;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < out.h; ++y)
            for (int x = 0; x < out.w; ++x) {
                RowEnt e{};
                e.in_off = (int32_t)(in.base + b * in.bstride + (int64_t)(y * stride + org_y) * in.pitch + (x * stride + org_x));
                e.in_pitch = in.pitch;
                e.out_off = (int32_t)(out.base + b * out.bstride + (int64_t)(y + 1) * out.pitch + (x + 1));
                if (res) {
                    int ry = y, rx = x;
                    if (res->h != out.h || res->w != out.w) {
                        ry = std::min((int)std::floor((y + 0.5) * ((double)res->h / out.h)), res->h - 1);
                        rx = std::min((int)std::floor((x + 0.5) * ((double)res->w / out.w)), res->w - 1);
                    }
                    e.res_off = (int32_t)(res->base + b * res->bstride + (int64_t)(ry + 1) * res->pitch + (rx + 1));
                }
                rows[r++] = e;
            }
    RowEnt* d = nullptr;
    BODCHK(h->dalloc(&d, rows.size(), false));
    HIPCHK(h, hipMemcpyAsync(d, rows.data(), rows.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->tables[key] = d;
    *tbl = d;
    return BOD_OK;
}

const HostTensor* find_w(bod_context* h, const std::string& name, int kind) {
    auto it = h->host_w.find(name + "/" + std::to_string(kind));
    return it == h->host_w.end() ? nullptr : &it->second;
}

// Fold BN (double), pack OHWI bf16 padded to cout_pad, upload.
bod_status pack_conv(bod_context* h, const std::string& name, const std::string& bn, int cout_pad_to,
                     PackedConv* out, int hx = 0) {          // hx: 0 = the data path's own form, 1 = hx rows (f16mx), 2 = h4 rows (f16mx4)
    const std::string ckey = hx == 2 ? name + ":h4" : hx ? name + ":hx" : name;             // (f16mx: tower layers 1.. are packed as hx rows, everything else as pairs)
    auto it = h->packed.find(ckey);
    if (it != h->packed.end()) { *out = it->second; return BOD_OK; }
    const HostTensor* k = find_w(h, name, 0);
    if (!k || k->shape.size() != 4) return h->fail(BOD_ERR_NOT_READY, "missing conv kernel '%s'", name.c_str());
    const int kh = (int)k->shape[0], kw = (int)k->shape[1], cin = (int)k->shape[2], cout = (int)k->shape[3];
    const HostTensor* b = find_w(h, name, 1);
    std::vector<double> scale(cout, 1.0), shift(cout, 0.0);
    for (int o = 0; o < cout; ++o) shift[o] = b ? (double)b->data[o] : 0.0;
    if (!bn.empty()) {
        const HostTensor *g = find_w(h, bn, 2), *be = find_w(h, bn, 3), *mu = find_w(h, bn, 4), *var = find_w(h, bn, 5);
        if (!g || !be || !mu || !var) return h->fail(BOD_ERR_NOT_READY, "missing batch-norm '%s'", bn.c_str());
        for (int o = 0; o < cout; ++o) {
            const double s = (double)g->data[o] / std::sqrt((double)var->data[o] + (double)BN_EPS);
            scale[o] = s;
            shift[o] = (shift[o] - (double)mu->data[o]) * s + (double)be->data[o];
        }
    }
    PackedConv pc;
    pc.cout = cout; pc.taps = kh * kw; pc.kw = kw; pc.cin = cin;
    pc.cout_pad = ((cout + cout_pad_to - 1) / cout_pad_to) * cout_pad_to;
    const size_t nw = (size_t)pc.cout_pad * pc.taps * cin;
    const bool as_f32 = h->es == 4 && !h->split;
    std::vector<uint16_t> w(as_f32 ? 0 : nw * (h->split ? 2 : 1), 0);
    std::vector<float> w32(as_f32 ? nw : 0, 0.f);
    std::vector<float> bias(pc.cout_pad, 0.f);
    for (int o = 0; o < cout; ++o) {
        bias[o] = (float)shift[o];
        for (int t = 0; t < pc.taps; ++t)
            for (int c = 0; c < cin; ++c) {
                const double v = (double)k->data[((size_t)t * cin + c) * cout + o] * scale[o];
                const size_t idx = ((size_t)o * pc.taps + t) * cin + c;
                if (h->split) {            // (hi, lo) pair of the fp32 weight: 32 hi then 32 lo per 64-slot group
                    const float f = (float)v;
                    const uint16_t hi = f2bf(f);
                    const size_t slot = ((size_t)o * pc.taps + t) * 2 * cin + (size_t)(c >> 5) * 64 + (c & 31);
                    w[slot] = hi; w[slot + 32] = f2bf(f - bf2f(hi));
                } else if (h->es == 2) w[idx] = f2bf((float)v);
                else w32[idx] = (float)v;
            }
    }
    if (hx) {                                  // one hx row per (cout, tap): same bytes as the pair form
        if (!h->split || cin % 64 != 0) return h->fail(BOD_ERR_INVALID_ARG, "hx weights need the (hi, lo) data path and cin %% 64 == 0");
        std::vector<float> row(cin);
        uint8_t* wb = reinterpret_cast<uint8_t*>(w.data());
        for (int o = 0; o < cout; ++o)
            for (int t = 0; t < pc.taps; ++t) {
                for (int c = 0; c < cin; ++c) row[c] = (float)((double)k->data[((size_t)t * cin + c) * cout + o] * scale[o]);
                if (hx == 2) {
                    if (cin != 256) return h->fail(BOD_ERR_INVALID_ARG, "h4 weights: 256 input channels");
                    pack_h4_row(row.data(), cin, wb + ((size_t)o * pc.taps + t) * cin * 4, true);
                } else pack_hx_row(row.data(), cin, wb + ((size_t)o * pc.taps + t) * cin * 4, true);
            }
    }
    size_t w_extra = 0;
    if (hx == 2) {          // ... + a compact copy of the scale bytes behind the rows: [tap][x][cout][half][ks], what the kernel's scale pieces read
        if (pc.cout_pad != 256) return h->fail(BOD_ERR_INVALID_ARG, "h4 weights: one 256-cout tile");
        w_extra = (size_t)pc.taps * 2 * 2048;
        w.resize(w.size() + w_extra / 2, 0);
        uint8_t* wb = reinterpret_cast<uint8_t*>(w.data());
        uint8_t* sc = wb + nw * h->es;
        for (int t = 0; t < pc.taps; ++t)
            for (int x = 0; x < 2; ++x)
                for (int o = 0; o < 256; ++o)
                    std::memcpy(sc + ((size_t)(t * 2 + x) * 256 + o) * 8, wb + ((size_t)o * pc.taps + t) * cin * 4 + 768 + 8 * x, 8);
    }
    BODCHK(h->dalloc(&pc.w, nw * h->es + w_extra, false));
    BODCHK(h->dalloc(&pc.bias, bias.size(), false));
    HIPCHK(h, hipMemcpyAsync(pc.w, as_f32 ? (const void*)w32.data() : (const void*)w.data(), nw * h->es + w_extra,
                             hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(pc.bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->packed[ckey] = pc;
    *out = pc;
    return BOD_OK;
}

ConvArgs base_args(const PackedConv& pc, const RowEnt* rows, int M, int in_cstride, int out_cstride) {
    ConvArgs a{};
    a.rows = rows; a.M = M; a.taps = pc.taps; a.KW = pc.kw; a.cin = pc.cin;
    a.in_cstride = in_cstride; a.cout_pad = pc.cout_pad; a.cout_valid = pc.cout;
    a.out_cstride = out_cstride; a.res_cstride = out_cstride; a.groups = 1;
    a.fan_count = 1; a.fan_stride = 0;
    return a;
}

// bf16x3 precision: the kernel counts (hi, lo) SLOTS wherever the bf16 mode counts channels of a pixel (kernels.h, ConvArgs::split)
void to_split_args(ConvArgs* a) {
    a->split = 1;
    a->cin *= 2; a->in_cstride *= 2; a->res_cstride *= 2;
    if (!(a->flags & CONV_OUT_F32)) a->out_cstride *= 2;
    for (int g = 0; g < 3; ++g) a->g[g].in_coff *= 2;
}

// conv + folded BN (+residual) (+ReLU) between two planes
bod_status add_conv(bod_context* h, const std::string& name, const std::string& bn, const Plane& in,
                    const Plane& out, int stride, bool same, bool relu, const Plane* res,
                    char* out_relu = nullptr) {
    PackedConv pc;
    BODCHK(pack_conv(h, name, bn, 64, &pc));
    if (pc.cin != in.C || pc.cout != out.C)
        return h->fail(BOD_ERR_INVALID_ARG, "conv '%s': weight shape [%d->%d] does not match planes [%d->%d]",
                       name.c_str(), pc.cin, pc.cout, in.C, out.C);
    const int kh = pc.taps / pc.kw;
    int oy = 1, ox = 1;
    if (same) { oy = 1 - same_pad_before(in.h, kh, stride); ox = 1 - same_pad_before(in.w, pc.kw, stride); }
    // the table depends on plane geometry only (offsets are relative to each buffer's base)
    char key[256];
    snprintf(key, sizeof key, "pp:%lld,%lld,%d:%lld,%lld,%d,%d,%d:%d:%d:%d:%lld,%lld,%d,%d,%d", (long long)in.base,
             (long long)in.bstride, in.pitch, (long long)out.base, (long long)out.bstride, out.pitch, out.h, out.w,
             stride, oy, ox, res ? (long long)res->base : -1LL, res ? (long long)res->bstride : 0LL,
             res ? res->pitch : 0, res ? res->h : 0, res ? res->w : 0);
    const RowEnt* tbl = nullptr;
    const int B = h->cfg.batch;
    BODCHK(make_table(h, key, B, in, out, stride, oy, ox, res, &tbl));
    Op op; op.kind = Op::CONV;
    op.conv = base_args(pc, tbl, B * out.h * out.w, in.C, out.C);
    op.conv.g[0] = ConvGroup{in.d, pc.w, pc.bias, out.d, res ? res->d : nullptr, out_relu, 0, 0, nullptr, nullptr, nullptr, 0, 0};
    op.conv.flags = relu ? CONV_RELU : 0;
    op.flops = 2.0 * op.conv.M * pc.cout * pc.taps * pc.cin;
    op.name = name; op.wname[0] = name; op.bnname[0] = bn;
    op.same_geom = stride == 1 && same && pc.taps == 9 && pc.kw == 3 && in.base == out.base && in.bstride == out.bstride && in.pitch == out.pitch &&
                   in.h == out.h && in.w == out.w;
    if (stride == 1 && same && pc.taps == 9 && pc.kw == 3 && in.h == out.h && in.w == out.w) { op.conv.plane_h = out.h; op.conv.plane_w = out.w; }
    // Split-K for layers with too few output tiles to fill the chip and a long reduction (P6 always; most of
    // stage 3-5 at batch 1): enough splits for >= ~256 workgroups, each keeping >= 4 K-tiles.
    static const bool splitk_on = [] { const char* e = getenv("BOD_CONV_SPLITK"); return !e || atoi(e) != 0; }();
    if (splitk_on && (h->es == 2 || h->split)) {
        const long tiles = (long)((op.conv.M + 127) / 128) * (pc.cout_pad % 128 == 0 ? pc.cout_pad / 128 : pc.cout_pad / 64);
        const int chunks = pc.cin * (h->split ? 2 : 1) / 64;          // K-tiles per tap (bf16x3: 32 channels per K-tile)
        int S = 1;
        while (tiles * S < 256 && S * 2 <= 16 && chunks % (S * 2) == 0 && (long)pc.taps * (chunks / (S * 2)) >= 4) S *= 2;
        if (S > 1) {
            op.conv.ksplit = S;
            h->splitk_elems = std::max(h->splitk_elems, (size_t)S * op.conv.M * pc.cout_pad);
        }
    }
    // Activation row reuse for the plane -> plane 3x3 stride-1 SAME convolutions of 256 -> 256 channels (stage 4's `2b` layers, P3-P5
    // of the FPN): the tower kernel's loop -- every (chunk, ky) staged once, the three kx taps read the same rows -- on this layer's own
    // tiles of x-adjacent runs (plan_tables.h).  Round 4, tests/tools/op_table.py: these launches ran the generic loop at 1 100-1 130
    // TFLOP/s against the towers' 1 270.  BOD_PLANE_XREUSE=0: the generic launches (different fp32 summation order: not bit-identical).
    static const bool plane_xr_on = [] { const char* e = getenv("BOD_PLANE_XREUSE"); return !e || atoi(e) != 0; }();
    static const bool xr_on = [] { const char* e = getenv("BOD_CONV_XREUSE"); return !e || atoi(e) != 0; }();
    if (plane_xr_on && xr_on && (h->es == 2 || h->split) && !h->cfg.training && op.conv.ksplit <= 1 && stride == 1 && same && pc.taps == 9 && pc.kw == 3 &&
        pc.cin == 256 && pc.cout == 256 && pc.cout_pad == 256 && !res && !out_relu && in.h == out.h && in.w == out.w) {
        ConvArgs probe = op.conv;
        if (conv_igemm_uses_full_cout_tile(probe)) {
            const std::string xkey = std::string(key) + ":xr";
            auto it = h->xr_tables.find(xkey);
            if (it == h->xr_tables.end()) {
                std::vector<RowEnt> rows((size_t)op.conv.M), tiled;
                std::vector<ExtRow> ext;
                HIPCHK(h, hipMemcpy(rows.data(), tbl, rows.size() * sizeof(RowEnt), hipMemcpyDeviceToHost));
                if (!xr_tile_rows(rows, tiled, ext)) return h->fail(BOD_ERR_INVALID_ARG, "row-reuse tiling of '%s': extended rows out of order", name.c_str());
                bod_context::XrTable t;
                t.m = (int)tiled.size();
                BODCHK(h->dalloc(&t.rows, tiled.size(), false));
                BODCHK(h->dalloc(&t.ext, ext.size(), false));
                HIPCHK(h, hipMemcpy(t.rows, tiled.data(), tiled.size() * sizeof(RowEnt), hipMemcpyHostToDevice));
                HIPCHK(h, hipMemcpy(t.ext, ext.data(), ext.size() * sizeof(int2), hipMemcpyHostToDevice));
                it = h->xr_tables.emplace(xkey, t).first;
            }
            op.conv.rows = it->second.rows; op.conv.ext = it->second.ext; op.conv.M = it->second.m; op.conv.xreuse = 2;
        }
    }
    h->ops.push_back(op);
    return BOD_OK;
}

bod_status build_geometry(bod_context* h) {
    const bod_config& c = h->cfg;
    const int H = c.image_h, W = c.image_w;
    if (H < 64 || W < 64) return h->fail(BOD_ERR_INVALID_ARG, "image size %dx%d too small", H, W);
    h->sh = (H - 7) / 2 + 1; h->sw = (W - 7) / 2 + 1;
    h->ph = (h->sh + 2 - 3) / 2 + 1; h->pw = (h->sw + 4 - 3) / 2 + 1;
    h->ch[2] = h->ph; h->cw[2] = h->pw;
    for (int s = 3; s <= 5; ++s) { h->ch[s] = (h->ch[s - 1] - 1) / 2 + 1; h->cw[s] = (h->cw[s - 1] - 1) / 2 + 1; }
    if (c.min_level != 3 || c.max_level != 7)
        return h->fail(BOD_ERR_INVALID_ARG, "only pyramid levels 3..7 are supported (got %d..%d)", c.min_level, c.max_level);
    h->nlev = 5;
    h->lh[0] = h->ch[3]; h->lw[0] = h->cw[3];
    h->lh[1] = h->ch[4]; h->lw[1] = h->cw[4];
    h->lh[2] = h->ch[5]; h->lw[2] = h->cw[5];
    h->lh[3] = (h->lh[2] + 1) / 2; h->lw[3] = (h->lw[2] + 1) / 2;
    h->lh[4] = (h->lh[3] + 1) / 2; h->lw[4] = (h->lw[3] + 1) / 2;
    int64_t pp = 0; int p = 0;
    for (int l = 0; l < 5; ++l) {
        const int stride = 1 << (l + 3);
        const int ah = (H + stride - 1) / stride, aw = (W + stride - 1) / stride;   // anchor grid (fpn_anchor_generator.py:28-29)
        if (ah != h->lh[l] || aw != h->lw[l])
            return h->fail(BOD_ERR_INVALID_ARG,
                           "pyramid level p%d is %dx%d but the anchor grid is %dx%d for a %dx%d image "
                           "(the reference's tf.concat of head outputs with anchors would mismatch too)",
                           l + 3, h->lh[l], h->lw[l], ah, aw, H, W);
        h->lvl_off[l] = pp; h->lvl_p0[l] = p;
        pp += (int64_t)(h->lh[l] + 2) * (h->lw[l] + 2);
        p += h->lh[l] * h->lw[l];
    }
    h->Ppad = pp; h->P = p; h->A = p * c.anchors_per_location;
    return BOD_OK;
}

Plane level_view(bod_context* h, int l) {
    Plane v = h->pyramid;
    v.base = h->lvl_off[l]; v.h = h->lh[l]; v.w = h->lw[l]; v.pitch = h->lw[l] + 2;
    return v;
}

// Raw head outputs [B,N,A,.] fp32.  Allocated with the plan, or on first use by a handle that never loads
// weights (a "post-only" handle whose raw buffers are filled by bod_set_raw / through bod_device_raw).
bod_status ensure_raw(bod_context* h) {
    const bod_config& c = h->cfg;
    const int out_ch[3] = {c.anchors_per_location * c.num_classes, c.anchors_per_location * 4, c.anchors_per_location * 10};
    for (int hd = 0; hd < 3; ++hd) {
        if ((hd == 2 && !c.has_covar_head) || h->raw[hd]) continue;
        BODCHK(h->dalloc(&h->raw[hd], (size_t)c.batch * c.mc_samples * h->P * out_ch[hd]));
    }
    // plans with the fused MC aggregation allocate the raw tensors on first use: point the raw-flavour ops at them
    for (Op& o : h->ops)
        if (o.kind == Op::CONV && o.flavour == FLAVOUR_RAW)
            for (int g = 0; g < o.conv.groups; ++g)
                if (o.conv.g[g].w2 && o.head[g] >= 0) o.conv.g[g].out2 = h->raw[o.head[g]];
    return BOD_OK;
}

bod_status train_init(bod_context* h);          // train_impl.inc
bod_status train_forward_only(bod_context* h, const float* dev, uint64_t seed, uint32_t first_image_id);
const uint32_t* train_dyn_rng(bod_context* h);
void train_destroy(bod_context* h);

bod_status build_plan(bod_context* h) {
    const bod_config& c = h->cfg;
    const int B = c.batch, N = c.mc_samples;
    const bool train_mode = c.training != 0;
    h->ops.clear();
    // ---------------- stem
    {
        const HostTensor* k = find_w(h, "conv1", 0);
        if (!k || k->shape.size() != 4 || k->shape[0] != 7 || k->shape[2] != 3 || k->shape[3] != 64)
            return h->fail(BOD_ERR_NOT_READY, "missing / malformed stem kernel 'conv1' [7,7,3,64]");
        const HostTensor* b = find_w(h, "conv1", 1);
        const HostTensor *g = find_w(h, "bn_conv1", 2), *be = find_w(h, "bn_conv1", 3), *mu = find_w(h, "bn_conv1", 4), *var = find_w(h, "bn_conv1", 5);
        if (!g || !be || !mu || !var) return h->fail(BOD_ERR_NOT_READY, "missing batch-norm 'bn_conv1'");
        std::vector<float> w(7 * 7 * 3 * 64), bias(64);
        for (int o = 0; o < 64; ++o) {
            const double s = (double)g->data[o] / std::sqrt((double)var->data[o] + (double)BN_EPS);
            bias[o] = (float)((((b ? (double)b->data[o] : 0.0) - (double)mu->data[o]) * s) + (double)be->data[o]);
            for (int t = 0; t < 147; ++t) w[(size_t)t * 64 + o] = (float)((double)k->data[(size_t)t * 64 + o] * s);
        }
        BODCHK(h->dalloc(&h->stem_w, w.size(), false));
        BODCHK(h->dalloc(&h->stem_b, bias.size(), false));
        HIPCHK(h, hipMemcpyAsync(h->stem_w, w.data(), w.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->stem_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        BODCHK(h->dalloc(&h->stem_out, (size_t)B * h->sh * h->sw * 64 * h->es));
        Op s; s.kind = Op::STEM; s.name = "conv1(stem)"; s.flops = 2.0 * B * h->sh * h->sw * 64.0 * 147.0; h->ops.push_back(s);
        Op p; p.kind = Op::POOL; p.name = "pool1"; h->ops.push_back(p);
    }
    // ---------------- ResNet-50 stages (feature_extractor.py:104-139)
    Plane x;
    BODCHK(new_plane(h, &x, B, h->ph, h->pw, 64));
    Plane pool_out = x;
    h->ops[1].conv.g[0].out = pool_out.d;   // remember pool destination
    const char* blocks[6] = {"", "", "abc", "abcd", c.backbone_depth == 101 ? "abcdefghijklmnopqrstuvw" : "abcdef", "abc"};
    const int f1s[6] = {0, 0, 64, 128, 256, 512};
    Plane taps[6];
    for (int st = 2; st <= 5; ++st) {
        const int f1 = f1s[st], f3 = f1 * 4, hh = h->ch[st], ww = h->cw[st];
        const int first_stride = st == 2 ? 1 : 2;
        Plane t1, t2, sc, oa, ob;
        BODCHK(new_plane(h, &t1, B, hh, ww, f1));
        BODCHK(new_plane(h, &t2, B, hh, ww, f1));
        BODCHK(new_plane(h, &sc, B, hh, ww, f3));
        BODCHK(new_plane(h, &oa, B, hh, ww, f3));
        BODCHK(new_plane(h, &ob, B, hh, ww, f3));
        bool use_a = true;
        // Bottleneck chain (stages 2 and 3, bf16 inference): a block's 3x3 conv `2b` keeps its tile in LDS and runs the block's 1x1
        // expansion `2c` (+ shortcut + ReLU) and the NEXT block's 1x1 reduction `2a` on it (conv_igemm.hip, ConvGroup.ch_*): the two
        // narrow intermediates never reach HBM -- 1.3 KB instead of 2 KB per pixel and block, two launches fewer.  The 3x3 comes FIRST
        // in the fused chain, so only the narrow t1 plane is re-read with a halo; t1 is double-buffered (the fused launch reads one
        // t1 plane with halos while it writes the next block's).  BOD_CHAIN_FUSION=0: the three separate launches (A/B; bit-identical).
        // Measured per block at 256 frames of 512x512 (BOD_TRACE_OPS): stage 2 (64 -> 256 channels) 2.23 -> 1.99 ms, its last block
        // 1.67 -> 1.45; stage 3 (128 -> 512) 1.33 -> 1.59 ms -- four shortcut passes per 128-pixel tile, each opening with an exposed
        // LDS-DMA round trip, cost more than the bytes they save -- so stage 3 is chained only on request (BOD_CHAIN_FUSION=3).
        // Later in round 3 the separate launches got their own streaming kernels (conv_pointwise.hip: the 1x1 expansion with the next
        // tile's rows in flight, the 64 -> 64 3x3 as a sliding window): per 256-frame step the chained stage 2 measures 23.1-23.6 ms of
        // backbone against 22.4-23.1 unchained on the same boxes, so the chain is OFF by default now (BOD_CHAIN_FUSION=2: stage 2,
        // =3: stages 2 and 3; the kernel, its guard entries and its bit-identity test stay).
        int chain_stages = 0;
        if (const char* e = getenv("BOD_CHAIN_FUSION")) chain_stages = atoi(e);
        bool chain_ok = h->es == 2 && !h->split && !train_mode && ((st == 2 && chain_stages >= 1) || (st == 3 && chain_stages >= 3));
        Plane t1alt;
        if (chain_ok) BODCHK(new_plane(h, &t1alt, B, hh, ww, f1));
        bool t1_ready = false;                  // the previous block's fused launch already produced this block's t1
        bool pw_t1_ready = false;               // (same, through the pointwise kernel's fused 2a)
        for (const char* bl = blocks[st]; *bl; ++bl) {
            char cb[64], bb[64];
            snprintf(cb, sizeof cb, "res%d%c_branch", st, *bl);
            snprintf(bb, sizeof bb, "bn%d%c_branch", st, *bl);
            const std::string c_(cb), b_(bb);
            if (chain_ok) {
                Plane& out = use_a ? oa : ob;
                const bool first = *bl == 'a';
                if (!t1_ready) BODCHK(add_conv(h, c_ + "2a", b_ + "2a", x, t1, first ? first_stride : 1, false, true, nullptr));
                if (first) BODCHK(add_conv(h, c_ + "1", b_ + "1", x, sc, first_stride, false, false, nullptr));
                BODCHK(add_conv(h, c_ + "2b", b_ + "2b", t1, t2, 1, true, true, nullptr));
                Op& op2b = h->ops.back();
                const Plane& shortcut = first ? sc : x;
                const bool can = op2b.conv.ksplit <= 1 && !conv_igemm_uses_big_tile(op2b.conv) && op2b.conv.cout_pad == f1 && (f1 == 64 || f1 == 128);
                if (!can) {                      // (split-K at batch 1, ...): the separate launches
                    BODCHK(add_conv(h, c_ + "2c", b_ + "2c", t2, out, 1, false, true, &shortcut));
                    t1_ready = false;
                } else {
                    PackedConv p2c;
                    BODCHK(pack_conv(h, c_ + "2c", b_ + "2c", 64, &p2c));
                    if (p2c.cin != f1 || p2c.cout != f3 || p2c.taps != 1 || p2c.cout_pad != f3)
                        return h->fail(BOD_ERR_INVALID_ARG, "conv '%s2c' must be 1x1 %d->%d", cb, f1, f3);
                    ConvGroup& G = op2b.conv.g[0];
                    G.ch_w2 = p2c.w; G.ch_b2 = p2c.bias; G.ch_res = shortcut.d; G.ch_out = out.d; G.ch_c2 = f3;
                    op2b.flops += 2.0 * op2b.conv.M * (double)f3 * f1;
                    op2b.name += "+2c"; op2b.wname[1] = c_ + "2c"; op2b.bnname[1] = b_ + "2c";
                    t1_ready = false;
                    if (bl[1] != 0) {            // the next block of the stage: its 2a (stride 1) rides along, into the other t1 plane
                        char nb[64], nbb[64];
                        snprintf(nb, sizeof nb, "res%d%c_branch2a", st, bl[1]);
                        snprintf(nbb, sizeof nbb, "bn%d%c_branch2a", st, bl[1]);
                        PackedConv p2a;
                        BODCHK(pack_conv(h, nb, nbb, 64, &p2a));
                        if (p2a.cin != f3 || p2a.cout != f1 || p2a.taps != 1 || p2a.cout_pad != f1)
                            return h->fail(BOD_ERR_INVALID_ARG, "conv '%s' must be 1x1 %d->%d", nb, f3, f1);
                        G.ch_w3 = p2a.w; G.ch_b3 = p2a.bias; G.ch_out3 = t1alt.d;
                        op2b.flops += 2.0 * op2b.conv.M * (double)f1 * f3;
                        op2b.name += std::string("+") + nb; op2b.wname[2] = nb; op2b.bnname[2] = nbb;
                        std::swap(t1, t1alt);
                        t1_ready = true;
                    }
                }
                if (first && (st == 3 || st == 4)) {
                    taps[st] = out;
                    Plane fresh;
                    BODCHK(new_plane(h, &fresh, B, hh, ww, f3));
                    x = out;
                    if (use_a) oa = fresh; else ob = fresh;
                    use_a = !use_a;
                    continue;
                }
                x = out;
                use_a = !use_a;
                continue;
            }
            // Stage 2 on the streaming pointwise kernel (conv_pointwise.hip, BC = 256): the 1x1 expansion just added carries the NEXT
            // block's 1x1 reduction -- computed from the finished tile in LDS, written to the t1 plane this block's 3x3 has already
            // consumed -- when the launch is certain to run on that kernel (conv_pointwise_can_fuse_next).  BOD_PW_FUSE_NEXT=0: off.
            auto fuse_next_2a = [&](const char* blk) -> bod_status {
                if (st != 2 || blk[1] == 0 || train_mode || h->es != 2 || h->split) return BOD_OK;
                Op& op2c = h->ops.back();
                if (!conv_pointwise_can_fuse_next(op2c.conv) || op2c.conv.ksplit > 1) return BOD_OK;
                char nb[64], nbb[64];
                snprintf(nb, sizeof nb, "res%d%c_branch2a", st, blk[1]);
                snprintf(nbb, sizeof nbb, "bn%d%c_branch2a", st, blk[1]);
                PackedConv p2a;
                BODCHK(pack_conv(h, nb, nbb, 64, &p2a));
                if (p2a.cin != f3 || p2a.cout != f1 || p2a.taps != 1 || p2a.cout_pad != 64) return BOD_OK;
                ConvGroup& G = op2c.conv.g[0];
                G.ch_w3 = p2a.w; G.ch_b3 = p2a.bias; G.ch_out3 = t1.d;
                op2c.flops += 2.0 * op2c.conv.M * (double)f1 * f3;
                op2c.name += std::string("+") + nb; op2c.wname[2] = nb; op2c.bnname[2] = nbb;
                pw_t1_ready = true;
                return BOD_OK;
            };
            if (train_mode) {                   // training keeps every activation: fresh planes per block
                BODCHK(new_plane(h, &t1, B, hh, ww, f1));
                BODCHK(new_plane(h, &t2, B, hh, ww, f1));
                BODCHK(new_plane(h, &sc, B, hh, ww, f3));
                BODCHK(new_plane(h, &oa, B, hh, ww, f3));
                BODCHK(new_plane(h, &ob, B, hh, ww, f3));
            }
            Plane& out = use_a ? oa : ob;
            if (*bl == 'a') {
                // Stage 2's ConvBlock: `branch1` (64 -> 256) and `2a` (64 -> 64 + ReLU) are both 1x1 over the pooled plane
                // (feature_extractor.py:283-309).  On the pointwise kernel's 256-channel tile the 2a rides on branch1's input tile
                // (conv_pointwise.hip, dual form): one launch, the plane read once.  BOD_PW_FUSE_DUAL=0: the two launches.
                bool dual = false;
                if (st == 2 && !train_mode && h->es == 2 && !h->split) {
                    BODCHK(add_conv(h, c_ + "1", b_ + "1", x, sc, first_stride, false, false, nullptr));
                    Op& op1 = h->ops.back();
                    PackedConv p2a;
                    BODCHK(pack_conv(h, c_ + "2a", b_ + "2a", 64, &p2a));
                    if (conv_pointwise_can_fuse_dual(op1.conv) && op1.conv.ksplit <= 1 && p2a.cin == 64 && p2a.cout == f1 && f1 == 64 &&
                        p2a.taps == 1 && p2a.cout_pad == 64) {
                        ConvGroup& G = op1.conv.g[0];
                        G.ch_w3 = p2a.w; G.ch_b3 = p2a.bias; G.ch_out3 = t1.d; G.ch_dual = 1;
                        op1.flops += 2.0 * op1.conv.M * (double)f1 * 64.0;
                        op1.name += std::string("+") + c_ + "2a"; op1.wname[2] = c_ + "2a"; op1.bnname[2] = b_ + "2a";
                        dual = true;
                    } else {
                        h->ops.pop_back();            // planned below in the reference's order
                    }
                }
                if (!dual) BODCHK(add_conv(h, c_ + "2a", b_ + "2a", x, t1, first_stride, false, true, nullptr));
                BODCHK(add_conv(h, c_ + "2b", b_ + "2b", t1, t2, 1, true, true, nullptr));
                if (!dual) BODCHK(add_conv(h, c_ + "1", b_ + "1", x, sc, first_stride, false, false, nullptr));
                BODCHK(add_conv(h, c_ + "2c", b_ + "2c", t2, out, 1, false, true, &sc));
                BODCHK(fuse_next_2a(bl));
                // the C3 / C4 taps are the block-'a' outputs (:119-120,:126-127): keep them alive
                if (st == 3 || st == 4) {
                    taps[st] = out;
                    Plane fresh;
                    BODCHK(new_plane(h, &fresh, B, hh, ww, f3));
                    x = out;
                    if (use_a) oa = fresh; else ob = fresh;
                    use_a = !use_a;
                    continue;
                }
            } else {
                if (!pw_t1_ready) BODCHK(add_conv(h, c_ + "2a", b_ + "2a", x, t1, 1, false, true, nullptr));
                pw_t1_ready = false;
                BODCHK(add_conv(h, c_ + "2b", b_ + "2b", t1, t2, 1, true, true, nullptr));
                BODCHK(add_conv(h, c_ + "2c", b_ + "2c", t2, out, 1, false, true, &x));
                BODCHK(fuse_next_2a(bl));
            }
            x = out;
            use_a = !use_a;
        }
    }
    const Plane c5 = x, c4 = taps[4], c3 = taps[3];

    // ---------------- FPN (feature_decoder.py:136-171)
    h->pyramid.C = 256; h->pyramid.bstride = h->Ppad;
    BODCHK(h->dalloc(&h->pyramid.d, (size_t)B * h->Ppad * 256 * h->es));
    Plane c5r, m4, m3, p6relu;
    BODCHK(new_plane(h, &c5r, B, h->lh[2], h->lw[2], 256));
    BODCHK(new_plane(h, &m4, B, h->lh[1], h->lw[1], 256));
    BODCHK(new_plane(h, &m3, B, h->lh[0], h->lw[0], 256));
    // relu(P6) shares the pyramid layout so one row table serves both outputs of the P6 conv
    char* p6relu_buf = nullptr;
    BODCHK(h->dalloc(&p6relu_buf, (size_t)B * h->Ppad * 256 * h->es));
    p6relu = h->pyramid; p6relu.d = p6relu_buf;
    BODCHK(add_conv(h, "C5_reduced", "", c5, c5r, 1, false, false, nullptr));
    BODCHK(add_conv(h, "P5", "", c5r, level_view(h, 2), 1, true, false, nullptr));
    BODCHK(add_conv(h, "P6", "", c5, level_view(h, 3), 2, true, false, nullptr, p6relu_buf));
    {
        Plane v = p6relu; v.base = h->lvl_off[3]; v.h = h->lh[3]; v.w = h->lw[3]; v.pitch = h->lw[3] + 2;
        BODCHK(add_conv(h, "P7", "", v, level_view(h, 4), 2, true, false, nullptr));
    }
    BODCHK(add_conv(h, "C4_reduced", "", c4, m4, 1, false, false, &c5r));     // + nearest-up(c5r)
    BODCHK(add_conv(h, "P4", "", m4, level_view(h, 1), 1, true, false, nullptr));
    BODCHK(add_conv(h, "C3_reduced", "", c3, m3, 1, false, false, &m4));      // + nearest-up(m4) (:162-167)
    BODCHK(add_conv(h, "P3", "", m3, level_view(h, 0), 1, true, false, nullptr));

    // ---------------- heads (multitask_headers.py; retinanet_model.py:78-109)
    const size_t act_elems = (size_t)B * N * h->Ppad * 256 * h->es;
    for (int hd = 0; hd < 3; ++hd) {
        if (hd == 2 && !c.has_covar_head) continue;
        if (train_mode) {                       // every tower activation is kept for the backward pass
            for (int l = 0; l < kHeadConvs[hd]; ++l) BODCHK(h->dalloc(&h->head_act_t[hd][l], act_elems));
            continue;
        }
        BODCHK(h->dalloc(&h->head_act[hd][0], act_elems));
        BODCHK(h->dalloc(&h->head_act[hd][1], act_elems));
    }
    const int out_ch[3] = {c.anchors_per_location * c.num_classes, c.anchors_per_location * 4, c.anchors_per_location * 10};
    // row tables: layer 1 (pyramid -> N dropout variants), layers 2.. (per sample), output 1x1
    std::vector<RowEnt> t1, t2, t3;
    {
        PyramidGeometry pg = pyramid_geometry(h->lh, h->lw);      // (plan_tables.h; bod_create laid the planes out with the same function's formulas)
        if (pg.Ppad != h->Ppad || pg.P != h->P) return h->fail(BOD_ERR_INVALID_ARG, "pyramid geometry mismatch");
        head_row_tables(pg, B, N, t1, t2, t3);
    }
    RowEnt *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
    BODCHK(h->dalloc(&d1, t1.size(), false));
    BODCHK(h->dalloc(&d2, t2.size(), false));
    BODCHK(h->dalloc(&d3, t3.size(), false));
    HIPCHK(h, hipMemcpyAsync(d1, t1.data(), t1.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d2, t2.data(), t2.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d3, t3.data(), t3.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));

    // Activation row reuse for the per-sample 3x3 tower layers: re-pack the rows into 256-slot tiles made
    // of runs of x-adjacent pixels and list each tile's extended input rows (kernels.h, ConvArgs::ext).
    RowEnt* d2x = nullptr; int2* dext = nullptr; int m2x = 0;
    bool xreuse = (h->es == 2 || h->split) && !train_mode;      // bf16, bf16x3 and f16mx: the per-sample tower layers (and, `xreuse0` below, the fan-out layer) on the row-reuse loop
    if (const char* e = getenv("BOD_CONV_XREUSE")) xreuse = xreuse && atoi(e) != 0;
    // f16mx precision: the towers' arithmetic exists in the row-reuse kernel only, which therefore runs them at every size (the tile
    // heuristics below choose between kernels of equal results; here the kernel IS the arithmetic).  BOD_TOWER_MX=0: plain bf16x3 towers.
    bool mx_plan = h->mx != 0 && xreuse;
    const int mxf = h->mx;                 // tower row format: 1 = hx, 2 = h4
    if (const char* e = getenv("BOD_TOWER_MX")) mx_plan = mx_plan && atoi(e) != 0;
    {
        ConvArgs probe{};
        probe.M = B * N * h->P; probe.cout_pad = 256; probe.fan_count = 1; probe.flags = CONV_RELU;
        xreuse = xreuse && (mx_plan || conv_igemm_uses_full_cout_tile(probe));
    }
    // rows -> 256-slot tiles of x-adjacent runs + each tile's extended input rows (kernels.h, ConvArgs::ext)
    static_assert(sizeof(ExtRow) == sizeof(int2), "ExtRow is the host-side twin of int2");
    auto make_xr_tiles = [&](const std::vector<RowEnt>& src, RowEnt** d_rows, int2** d_ext, int* m_out) -> bod_status {
        std::vector<RowEnt> tiled;
        std::vector<ExtRow> ext;
        if (!xr_tile_rows(src, tiled, ext)) return h->fail(BOD_ERR_INVALID_ARG, "row-reuse tiling: extended rows out of order");
        *m_out = (int)tiled.size();
        BODCHK(h->dalloc(d_rows, tiled.size(), false));
        BODCHK(h->dalloc(d_ext, ext.size(), false));
        HIPCHK(h, hipMemcpyAsync(*d_rows, tiled.data(), tiled.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(*d_ext, ext.data(), ext.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return BOD_OK;
    };
    if (xreuse) BODCHK(make_xr_tiles(t2, &d2x, &dext, &m2x));
    // the first tower layer (one convolution per image, N-way dropout fan-out epilogue) takes the row-reuse loop too once
    // its launch is on the 256x256 tile (conv_igemm.hip: from 1 024 tiles on): activation reads x4.5 -> x1.2 of the
    // algorithmic bytes (profiles/round1_head_conv_pmc.json, launch 0).  BOD_FAN_XREUSE=0: A/B aid.
    RowEnt* d1x = nullptr; int2* dext1 = nullptr; int m1x = 0;
    bool xreuse0 = xreuse;          // (N = 1: no fan-out, a plain three-head launch over the pyramid -- on the same loop since round 4: 1.07 -> 1.23 PFLOP/s)
    if (const char* e = getenv("BOD_FAN_XREUSE")) xreuse0 = xreuse0 && atoi(e) != 0;
    {
        ConvArgs probe{};
        probe.M = B * h->P; probe.cout_pad = 256; probe.fan_count = N; probe.flags = CONV_RELU | CONV_DROPOUT; probe.groups = c.has_covar_head ? 3 : 2;
        xreuse0 = xreuse0 && conv_igemm_uses_full_cout_tile(probe);
    }
    if (mx_plan) xreuse0 = true;           // (the first tower layer writes the hx rows the next one reads)
    // ... and reads hx rows itself: the pyramid, which the FPN's kernels write as (hi, lo) pairs, is converted once per forward (1.5 GB
    // in, 1.5 GB out per 256 frames: 0.7 ms) so that the three first-layer convs run at 1.5 instead of 3 products too.  Not on overlap
    // handles (two pyramid buffers).  BOD_MX_LAYER0=0: the bf16x3 loop on the pair rows (conv_igemm_mx_kernel<2>).
    bool mx_l0 = mx_plan && !c.pipeline_overlap;
    if (const char* e = getenv("BOD_MX_LAYER0")) mx_l0 = mx_l0 && atoi(e) != 0;
    if (mx_l0) BODCHK(h->dalloc(&h->pyr_hx, (size_t)B * h->Ppad * 256 * 4));
    if (xreuse0) BODCHK(make_xr_tiles(t1, &d1x, &dext1, &m1x));

    const bool mc = train_mode || std::max(N, c.mc_ensemble_size) > 1;    // mc_dropout_enabled (retinanet_model.py:74-77); training: dropout on (:113-129)
    const uint32_t thr = (uint32_t)std::floor((double)c.dropout_rate * 65536.0);
    const float dscale = (float)(1.0 / (1.0 - (double)c.dropout_rate));
    const int nheads = c.has_covar_head ? 3 : 2;
    // which heads are still running a tower conv at `layer`
    // Fuse each head's 1x1 output conv into the epilogue of its last tower layer whenever that layer runs
    // with the full 256-channel cout tile (bf16 mode, enough rows): the last tower activation then never
    // goes to HBM and three launches disappear.  BOD_FUSE_HEAD_OUTPUT=0 keeps the separate launches.
    // (bf16x3: the fused form exists in the row-reuse kernel's epilogue only -- `xreuse` below -- with the same products in the
    // same order as the separate 1x1 launches: bit-identical head outputs either way)
    bool fuse_out = (h->es == 2 || (h->split && xreuse)) && !train_mode;
    if (const char* e = getenv("BOD_FUSE_HEAD_OUTPUT")) fuse_out = fuse_out && atoi(e) != 0;
    {
        ConvArgs probe{};
        probe.M = B * N * h->P; probe.cout_pad = 256; probe.fan_count = 1; probe.flags = CONV_RELU;
        fuse_out = fuse_out && (mx_plan || conv_igemm_uses_full_cout_tile(probe));
    }
    // ---- MC aggregation fused into the last tower layers' epilogues (inference_utils.py:31-60,
    // :220-244): a tile of those layers must hold ALL N samples of its pixels, so they get their own row table -- tiles of
    // Q <= 256 / N pixel slots, row = slot * N + sample, made of runs of x-adjacent pixels whose extended rows (run + 2,
    // once per sample) fit the 320 staged rows.  BOD_FUSE_AGGREGATION=0 keeps the raw tensors + the posterior's own loops.
    bool agg = xreuse && fuse_out && N >= 2 && 256 / N >= 1 && 320 / N - 2 >= 1 && c.mc_ensemble_size <= N;
    if (const char* e = getenv("BOD_FUSE_AGGREGATION")) agg = agg && atoi(e) != 0;
    RowEnt* d2a = nullptr; int2* dexta = nullptr; int m2a = 0;
    if (agg) {
        std::vector<RowEnt> tiled;
        std::vector<ExtRow> ext;
        const int rc = xr_tile_rows_aggregated(t2, B, N, h->P, tiled, ext);
        if (rc == 1) return h->fail(BOD_ERR_INVALID_ARG, "aggregated tiling: no pixel fits a tile (N = %d)", N);
        if (rc == 2) return h->fail(BOD_ERR_INVALID_ARG, "aggregated tiling: extended rows out of order");
        m2a = (int)tiled.size();
        BODCHK(h->dalloc(&d2a, tiled.size(), false));
        BODCHK(h->dalloc(&dexta, ext.size(), false));
        HIPCHK(h, hipMemcpyAsync(d2a, tiled.data(), tiled.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(dexta, ext.data(), ext.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const size_t BA = (size_t)B * h->A;
        BODCHK(h->dalloc(&h->agg[0], BA * c.num_classes));
        BODCHK(h->dalloc(&h->agg[1], BA * 16));
        if (c.has_covar_head) BODCHK(h->dalloc(&h->agg[2], BA * 10));
    } else {
        BODCHK(ensure_raw(h));                 // the ops below reference the raw tensors directly
    }
    h->agg_plan = agg; h->plan_fused_out = fuse_out; h->plan_xreuse = xreuse; h->plan_xreuse0 = xreuse0; h->plan_mx = mx_plan ? mxf : 0;
    // A layer's launch takes the sample-complete ("aggregated") tiling only for the heads that END there (fused 1x1 + MC aggregation): such
    // a tile holds 25 pixels x 10 samples = 250 of its 256 rows (240 at N = 30), so every other head's conv of that layer -- the
    // classification and covariance towers at layer 2 -- would pay 2.8 % (6 %) more MFMA work for nothing.  Round 4: those heads run the
    // layer as their own launch on the plain tiling (part 0); the ending heads' launch (part 1) exists in the two flavours.
    // f16mx4: all towers on h4 rows (e2m1 cross terms).  BOD_MX4_BOX_HX=1 keeps the box-regression tower on hx rows (e2m3) -- the epistemic
    // covariance is a sample variance of N nearly equal boxes and amplifies the box outputs' rounding error; measured on the BASELINE-size
    // frames: fused covariance entries 2.5e-3 -> 1.8e-3 (f16mx: 5e-4), 1 024 -> 1 000 frames/s (f16mx: 944) -- as separate launches from
    // layer 1 on; layer 0 then reads ONE pyramid (hx) and writes each head's format.
    int hfmt[3] = {mxf, mxf, mxf};
    if (mxf == 2) { const char* e = getenv("BOD_MX4_BOX_HX"); if (e && atoi(e) != 0) hfmt[1] = 1; }
    // f16mx, round 6: the CLASSIFICATION tower alone runs on h4 rows (e2m1 cross terms: three quarters of the staged bytes and K-tiles of 3
    // of the 8 per-sample convs) -- its logits go through a softmax and a 30-draw categorical and lose nothing measurable (raw class
    // logits 1.2e-4 against 9.3e-5 on hx rows; the bench line's gate: every clause unchanged, categorical draw flips 1 -> 5 of 1 600
    // detections), while the box and covariance towers, whose rounding reaches the fused covariance entries, stay on hx rows.  The same
    // mixed plan as above: the towers run as one launch per format from layer 1 on.  +1.6-2.1 % frames/s.  BOD_MX_CLS_H4=0: all on hx rows.
    if (mxf == 1) { static const bool cls_h4 = [] { const char* e = getenv("BOD_MX_CLS_H4"); return !e || atoi(e) != 0; }(); if (cls_h4) hfmt[0] = 2; }
    const bool mixed = mx_plan && hfmt[0] != hfmt[1];
    const int pyr_fmt = mixed ? 1 : mxf;
    h->pyr_fmt = pyr_fmt;
    for (int layer = 0; layer < 4; ++layer)
    for (int fpass = 0; fpass < 2; ++fpass)
    for (int part = 0; part < 2; ++part)
    for (int flav = (agg && layer >= 2) ? FLAVOUR_RAW : FLAVOUR_BOTH; flav <= ((agg && layer >= 2) ? FLAVOUR_AGG : FLAVOUR_BOTH); ++flav) {
        static const bool split_on = [] { const char* e = getenv("BOD_SPLIT_AGG_LAUNCH"); return !e || atoi(e) != 0; }();     // (=0: one launch per layer on the aggregated tiling, A/B aid)
        const bool split_launch = agg && layer >= 2 && split_on;       // part 0: heads that continue, part 1: heads that end at this layer
        if (fpass == 1 && (!mixed || layer == 0)) continue;
        const int lfmt = !mx_plan ? 0 : layer == 0 ? pyr_fmt : mixed ? (fpass == 0 ? 2 : 1) : mxf;       // row format this launch READS
        if (!split_launch && part == 1) continue;
        if (split_launch && part == 0 && flav == FLAVOUR_AGG) continue;      // the continuing heads' launch is the same in both flavours: planned once
        const bool both = split_launch && part == 0;
        Op op; op.kind = Op::CONV; op.is_head3x3 = true; op.flavour = both ? FLAVOUR_BOTH : flav;
        int g = 0; PackedConv pc0{};
        double fused_flops = 0;
        for (int hd = 0; hd < nheads; ++hd) {
            if (layer >= kHeadConvs[hd]) continue;
            if (split_launch && (part == 1) != (layer == kHeadConvs[hd] - 1)) continue;
            if (mixed && layer > 0 && hfmt[hd] != lfmt) continue;
            PackedConv pc;
            BODCHK(pack_conv(h, std::string(kHeadPrefix[hd]) + "_" + std::to_string(layer), "", 128, &pc, (mx_plan && (layer > 0 || mx_l0)) ? lfmt : 0));
            if (pc.cin != 256 || pc.cout != 256 || pc.taps != 9)
                return h->fail(BOD_ERR_INVALID_ARG, "head conv %s_%d must be 3x3 256->256", kHeadPrefix[hd], layer);
            op.wname[g] = std::string(kHeadPrefix[hd]) + "_" + std::to_string(layer);
            ConvGroup cg{};
            cg.in = layer == 0 ? (mx_l0 ? h->pyr_hx : h->pyramid.d) : (train_mode ? h->head_act_t[hd][layer - 1] : h->head_act[hd][(layer + 1) & 1]);
            cg.w = pc.w; cg.bias = pc.bias;
            cg.out = train_mode ? h->head_act_t[hd][layer] : h->head_act[hd][layer & 1];
            cg.layer_id = hd * 4 + layer;
            cg.out_hx = (mx_plan && layer < kHeadConvs[hd] - 1) ? hfmt[hd] : 0;      // read by the head's next tower layer: hx rows; a last layer feeds the 1x1: pairs
            if (fuse_out && layer == kHeadConvs[hd] - 1) {
                PackedConv po;
                BODCHK(pack_conv(h, kHeadPrefix[hd], "", 32, &po));
                if (po.cin != 256 || po.cout != out_ch[hd] || po.taps != 1)
                    return h->fail(BOD_ERR_INVALID_ARG, "head output conv %s must be 1x1 256->%d (got %d->%d)", kHeadPrefix[hd], out_ch[hd], po.cin, po.cout);
                cg.w2 = po.w; cg.bias2 = po.bias; cg.out2 = h->raw[hd]; cg.cout2 = po.cout; cg.out2_cstride = out_ch[hd];
                op.head[g] = hd;
                if (flav == FLAVOUR_AGG) {             // reduce over the MC samples inside the tile instead of writing [B,N,A,.]
                    cg.out2 = nullptr;
                    cg.agg_kind = hd == 0 ? AGG_CLS : hd == 1 ? AGG_BOX : AGG_COV;
                    cg.agg_n = N; cg.agg_P = h->P; cg.agg_C = c.num_classes; cg.agg_out = h->agg[hd]; cg.anchors = h->d_anchors;
                }
                fused_flops += 2.0 * ((double)B * N * h->P) * 256.0 * out_ch[hd];      // the 1x1 output conv runs inside this launch
            }
            op.conv.g[g] = cg;
            if (g == 0) pc0 = pc;
            ++g;
        }
        if (g == 0) continue;                                // (no head of this part at this layer)
        const int M = layer == 0 ? B * h->P : B * N * h->P;
        ConvArgs a = base_args(pc0, layer == 0 ? d1 : d2, M, 256, 256);
        for (int q = 0; q < g; ++q) a.g[q] = op.conv.g[q];
        a.groups = g;
        a.flags = CONV_RELU | (mc ? CONV_DROPOUT : 0);
        a.fan_count = layer == 0 ? N : 1;
        a.fan_stride = (int32_t)h->Ppad;
        a.drop_threshold = thr; a.drop_scale = dscale;
        a.mx = mx_plan ? ((layer == 0 && !mx_l0) ? 2 : (lfmt == 2 ? 3 : 1)) : 0;
        op.hx_pyramid = mx_l0 && layer == 0;
        {   // BOD_MX_LOADER=0|1|2: which waves of the f16mx loop issue the weight pieces (conv_igemm.hip: all / lower four / upper four)
            // (same-box A/B at 256 frames, two rounds each: towers 198.0 / 196.2 / 199.7 ms with 0 / 1 / 2)
            static const int mx_loader = getenv("BOD_MX_LOADER") ? atoi(getenv("BOD_MX_LOADER")) : 1;
            // BOD_TOWER_LOADER=1: the same pairing in the bf16 tower loop (A/B switch; measured 0.8 % SLOWER there: 211.6 against 209.9 ms per 512 frames)
            static const int tower_loader = getenv("BOD_TOWER_LOADER") ? atoi(getenv("BOD_TOWER_LOADER")) : 0;
            a.mx_loader = mx_plan ? mx_loader : ((h->es == 2 && layer > 0) ? tower_loader : 0);
        }
        if (xreuse0 && layer == 0) { a.rows = d1x; a.M = m1x; a.ext = dext1; a.xreuse = 2; }
        if (xreuse && layer > 0) {
            a.rows = d2x; a.M = m2x; a.ext = dext;
            if (agg && layer >= 2 && !(split_launch && part == 0)) { a.rows = d2a; a.M = m2a; a.ext = dexta; }       // sample-complete tiles (both flavours)
            a.xreuse = 2;       // 32-bit activation offsets against the tile's first extended row: any buffer size
        }
        op.conv = a;
        op.flops = 2.0 * M * 256.0 * 2304.0 * g + fused_flops;
        op.name = "head_tower_layer_" + std::to_string(layer) + (both ? "" : flav == FLAVOUR_AGG ? "(aggregating)" : flav == FLAVOUR_RAW ? "(raw)" : "");
        op.same_geom = N == 1;            // pyramid [B][Ppad] and head planes [B*N][Ppad] coincide at N = 1
        h->ops.push_back(op);
    }
    for (int hd = 0; hd < nheads && !fuse_out; ++hd) {
        PackedConv pc;
        BODCHK(pack_conv(h, kHeadPrefix[hd], "", 64, &pc));
        if (pc.cin != 256 || pc.cout != out_ch[hd] || pc.taps != 1)
            return h->fail(BOD_ERR_INVALID_ARG, "head output conv %s must be 1x1 256->%d (got %d->%d)", kHeadPrefix[hd], out_ch[hd], pc.cin, pc.cout);
        Op op; op.kind = Op::CONV;
        ConvArgs a = base_args(pc, d3, B * N * h->P, 256, out_ch[hd]);
        a.g[0] = ConvGroup{train_mode ? h->head_act_t[hd][kHeadConvs[hd] - 1] : h->head_act[hd][(kHeadConvs[hd] - 1) & 1], pc.w, pc.bias, h->raw[hd], nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0};
        a.flags = CONV_OUT_F32;
        op.conv = a;
        op.flops = 2.0 * a.M * pc.cout * 256.0;
        op.name = kHeadPrefix[hd]; op.wname[0] = kHeadPrefix[hd];
        h->ops.push_back(op);
    }
    if (h->splitk_elems) {
        BODCHK(h->dalloc(&h->splitk_partial, h->splitk_elems));
        for (Op& o : h->ops) if (o.kind == Op::CONV && o.conv.ksplit > 1) o.conv.partial = h->splitk_partial;
    }
    if (h->split) for (Op& o : h->ops) if (o.kind == Op::CONV) to_split_args(&o.conv);
    h->first_head_op = (int)h->ops.size();
    for (size_t i = 0; i < h->ops.size(); ++i) if (h->ops[i].is_head3x3) { h->first_head_op = (int)i; break; }
    h->pyr_d[0] = h->pyramid.d;
    // BOD_DUMP_OPS=1: the plan, one line per op on stderr, with the ALGORITHMIC bytes of the launch (input pixels read once, output and
    // shortcut once, weights once) -- tests/tools/op_table.py joins it with a rocprofv3 kernel trace into a per-op roofline table
    if (const char* e = getenv("BOD_DUMP_OPS")) if (atoi(e) != 0) {
        fprintf(stderr, "# ops: index name kind flavour M taps cin cout groups fan has_res fused_next flops bytes\n");
        for (size_t i = 0; i < h->ops.size(); ++i) {
            const Op& o = h->ops[i];
            const ConvArgs& a = o.conv;
            double bytes = 0;
            if (o.kind == Op::STEM)                    // fp32 frames in, pooled 64-channel plane out (the fused stem + pool kernel)
                bytes = (double)B * c.image_h * c.image_w * 3 * 4 + (double)B * h->ph * h->pw * 64 * h->es;
            if (o.kind == Op::CONV) {
                const double es = (a.flags & CONV_OUT_F32) ? 4.0 : (double)h->es;
                for (int g = 0; g < a.groups; ++g) {
                    bytes += (double)a.M * a.cin * h->es;                                                    // input pixels (stride-2 3x3: ~4x that)
                    bytes += (double)a.M * std::max(1, a.fan_count) * a.cout_valid * es;                     // output(s)
                    if (a.g[g].res) bytes += (double)a.M * a.cout_valid * h->es;
                    if (a.g[g].ch_w3) bytes += (double)a.M * 64 * h->es;
                    bytes += (double)a.taps * a.cin * a.cout_pad * h->es;
                }
            }
            fprintf(stderr, "# op %zu %s %d %d %d %d %d %d %d %d %d %d %.6g %.6g\n", i, o.name.c_str(), (int)o.kind, o.flavour, a.M, a.taps, a.cin,
                    a.cout_valid, a.groups, a.fan_count, o.kind == Op::CONV && a.g[0].res ? 1 : 0, o.kind == Op::CONV && a.g[0].ch_w3 ? 1 : 0, o.flops, bytes);
        }
    }
    if (h->overlap_mode) BODCHK(h->dalloc(&h->pyr_d[1], (size_t)B * h->Ppad * 256 * h->es));      // zero borders like the first
    return BOD_OK;
}

bod_status alloc_post(bod_context* h) {
    const bod_config& c = h->cfg;
    const size_t BA = (size_t)c.batch * h->A;
    const int nblocks = (h->A + 255) / 256;
    PostBuffers& p = h->pb;
    BODCHK(h->dalloc(&p.keep, BA));
    BODCHK(h->dalloc(&p.d_counts, BA * c.num_classes));
    BODCHK(h->dalloc(&p.block_counts, (size_t)c.batch * nblocks));
    BODCHK(h->dalloc(&p.num_kept, (size_t)c.batch));
    BODCHK(h->dalloc(&p.counts, BA * c.num_classes));
    BODCHK(h->dalloc(&p.score, BA * c.num_classes));
    BODCHK(h->dalloc(&p.means, BA * 4));
    BODCHK(h->dalloc(&p.covs, BA * 16));
    BODCHK(h->dalloc(&p.ranking, BA));
    BODCHK(h->dalloc(&p.corners, BA * 4));
    BODCHK(h->dalloc(&p.anchor_index, BA));
    BODCHK(h->dalloc(&h->d_anchors, (size_t)h->A * 4));
    const size_t BApad = (size_t)h->cfg.batch * ((h->A + 511) & ~(size_t)511);   // nms_kernel pads its queue to 512
    BODCHK(h->dalloc(&h->nms_scores, BApad));
    BODCHK(h->dalloc(&h->nms_begin, BApad));
    const size_t BK = (size_t)c.batch * c.nms_max_output_size;
    for (int sidx = 0; sidx < 2; ++sidx) {
        BODCHK(h->dalloc(&h->nms_sel_s[sidx], BK));
        BODCHK(h->dalloc(&h->nms_nsel_s[sidx], (size_t)c.batch));
        BODCHK(h->dalloc(&h->out_scores_s[sidx], BK * c.num_classes));
        BODCHK(h->dalloc(&h->out_means_s[sidx], BK * 4));
        BODCHK(h->dalloc(&h->out_covs_s[sidx], BK * 16));
        BODCHK(h->dalloc(&h->out_counts_s[sidx], BK * c.num_classes));
        const size_t stage_bytes = ((size_t)c.batch + BK * (2 * (size_t)c.num_classes + 20)) * 4;
        if (hipHostMalloc(reinterpret_cast<void**>(&h->host_stage[sidx]), stage_bytes, hipHostMallocDefault) != hipSuccess)
            return h->fail(BOD_ERR_OOM, "pinned host staging buffer (%zu bytes)", stage_bytes);
    }
    h->select_slot(0);
    BODCHK(h->dalloc(&h->d_images, (size_t)c.batch * c.image_h * c.image_w * 3));
    h->d_images_b[0] = h->d_images;
    return BOD_OK;
}

PostCfg post_cfg(bod_context* h, uint64_t seed, uint32_t first_image) {
    const bod_config& c = h->cfg;
    PostCfg p{};
    p.B = c.batch; p.N = c.mc_samples; p.A = h->A; p.C = c.num_classes; p.draws = c.num_categorical_draws;
    p.use_full_covar = c.use_full_covar; p.has_covar = c.has_covar_head; p.dirichlet = c.dirichlet_non_informative;
    p.gaussian_iso = c.gaussian_isotropic; p.ranking_method = c.ranking_method; p.iso_var = c.isotropic_variance;
    p.kitti_sh = c.kitti_scale_h; p.kitti_sw = c.kitti_scale_w;
    p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32); p.image_base = first_image;
    return p;
}

// flavour: FLAVOUR_RAW = per-sample head outputs into raw[] (RetinaNetModel.call's tensors), FLAVOUR_AGG = MC statistics
// reduced inside the last tower layers' tiles (plans with agg_plan only).  only_flavoured: run just the ops that differ
// between the two (materialise_raw re-runs the raw flavour of the last layers on the activations still in HBM).
// ---- rocprofv3 markers (tracing): with BOD_ROCTX=1 every stage of a step is bracketed by a roctx range --
// bod:stem, bod:res2 .. bod:res5, bod:fpn, bod:head_tower_layer_k, bod:posterior, bod:nms, bod:cluster_fuse, bod:collect,
// bod:upload -- so that `rocprofv3 --marker-trace --kernel-trace` lays the kernels of a step out by stage.  The roctx library is
// opened at run time (no link dependency; without the variable not a single call is made).  The ranges bracket the ENQUEUE of a
// stage on the host; the stage timers of the C ABI (bod_profile_begin / bod_profile_end) are the device-side figures.
struct Markers {
    bool on = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Markers() {
        const char* e = getenv("BOD_ROCTX");
        if (!e || !atoi(e)) return;
        void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { fprintf(stderr, "bayesod: BOD_ROCTX=1 but no roctx library could be opened (%s)\n", dlerror()); return; }
        push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        on = push && pop;
    }
};
static Markers& markers() { static Markers m; return m; }
struct MarkerRange {                    // scope = one range
    bool on;
    explicit MarkerRange(const char* name) : on(markers().on) { if (on) markers().push(name); }
    ~MarkerRange() { if (on) markers().pop(); }
    MarkerRange(const MarkerRange&) = delete;
    MarkerRange& operator=(const MarkerRange&) = delete;
};
// stage of a forward op, from its name: conv1 / pool1 -> stem, resK* -> resK, C?_reduced / P? -> fpn, head_tower_layer_k(...) -> itself
static std::string op_stage(const std::string& name) {
    if (name.rfind("conv1", 0) == 0 || name.rfind("pool1", 0) == 0) return "bod:stem";
    if (name.rfind("res", 0) == 0 && name.size() > 3) return "bod:" + name.substr(0, 4);
    if (name.rfind("head_tower_layer_", 0) == 0) return "bod:" + name.substr(0, name.find('(') == std::string::npos ? name.size() : name.find('('));
    if (name.rfind("pyramid_", 0) == 0) return "bod:head_outputs";      // separate 1x1 output convs (plans without the fused epilogue)
    return "bod:fpn";
}

// Records ev_img_free for the image buffer the current call reads (bod_device_images_buffer(k)): everything enqueued on the main
// stream so far has finished with the frames once the event fires; bod_upload_frames_u8_async makes the copy stream wait on it.
bod_status mark_images_consumed(bod_context* h, hipStream_t st = nullptr) {
    if (h->cur_img_buf >= 0 && h->copy) {
        HIPCHK(h, hipEventRecord(h->ev_img_free[h->cur_img_buf], st ? st : h->stream));
        h->img_free_pending[h->cur_img_buf] = true;
    }
    return BOD_OK;
}

// Every entry point but the pipelined ones (bod_infer_async, bod_collect, bod_gather_detections of a ticket, the asynchronous upload)
// works on the unmasked `full` stream: it first waits for whatever the partition's two streams still hold.
bod_status join_overlap(bod_context* h) {
    if (!h->ov_dirty) return BOD_OK;
    HIPCHK(h, hipStreamSynchronize(h->front));
    HIPCHK(h, hipStreamSynchronize(h->back));
    h->ov_dirty = false;
    return BOD_OK;
}

// overlapped (bod_infer_async on a pipeline_overlap handle; h->stream == h->back for the call): the ops in front of the first head
// launch go to h->front, the pyramid they write is buffer `fwd_parity` of two, and two events hand it to the back and take it back.
bod_status run_forward(bod_context* h, const float* dev_images, uint64_t seed, uint32_t first_image, int flavour = FLAVOUR_RAW,
                       bool only_flavoured = false, bool overlapped = false) {
    const bod_config& c = h->cfg;
    if (flavour == FLAVOUR_RAW && !only_flavoured) BODCHK(ensure_raw(h));
    const bool ov = overlapped && h->overlap_mode != 0 && !only_flavoured;
    hipStream_t fs = ov ? h->front : h->stream;
    int par = 0;
    if (ov) {
        par = h->fwd_parity; h->fwd_parity ^= 1;
        if (h->l0_pending[par]) { HIPCHK(h, hipStreamWaitEvent(fs, h->ev_l0_done[par], 0)); h->l0_pending[par] = false; }
    }
    if (!only_flavoured) h->pyr_last = par;
    // Kernel CHOICES (sliding-window / pointwise / fused-stem eligibility: launch_conv_igemm, workgroups against compute units) are made
    // against the WHOLE chip on every stream: a CU-masked front stream that chose by its own 32 CUs ran other kernels than bod_infer
    // and a serial handle at mid-size batches (64 frames at 512x512: 64 column strips >= 32 take slide3x3_c128, which re-associates one
    // fp32 add per output, < 256 do not) and broke the header's "bit-identical to the serial pipeline" (round-4 advisor finding).  The
    // overlap mode is an opt-in A/B switch that measured slower: the contract wins over its planner.
    const int cus_front = h->n_cu;
    const int cus_back = h->n_cu;
    // BOD_TRACE_OPS=k: the k-th forward call is traced op by op (HIP events on the engine stream) and a table
    // is printed to stderr -- a development aid (tests/tools), off by default.
    static const int trace_call = getenv("BOD_TRACE_OPS") ? atoi(getenv("BOD_TRACE_OPS")) : 0;
    static int call_no = 0;
    const bool trace = trace_call > 0 && ++call_no == trace_call && !ov;
    std::vector<hipEvent_t> tev;
    if (trace) {
        tev.resize(h->ops.size() + 1);
        for (hipEvent_t& e : tev) HIPCHK(h, hipEventCreate(&e));
        HIPCHK(h, hipEventRecord(tev[0], h->stream));
    }
    size_t op_i = 0;
    bool stem_pool_fused = false;
    std::string cur_stage;                                        // open marker range (BOD_ROCTX=1)
    struct StageCloser { std::string& s; ~StageCloser() { if (!s.empty()) markers().pop(); } } stage_closer{cur_stage};
    for (Op& op : h->ops) {
        if (trace && op_i > 0) HIPCHK(h, hipEventRecord(tev[op_i], h->stream));
        const bool is_front = (int)op_i < h->first_head_op;
        const bool first_back = (int)op_i == h->first_head_op;
        hipStream_t st = is_front ? fs : h->stream;
        ++op_i;
        if ((op.flavour != FLAVOUR_BOTH && op.flavour != flavour) || (only_flavoured && op.flavour == FLAVOUR_BOTH)) continue;
        if ((int)op_i - 1 < h->dev_op_lo || (int)op_i - 1 >= h->dev_op_hi) continue;          // (BOD_FORWARD_OPS: development)
        if (ov && first_back) {                          // the pyramid is complete: hand it to the back
            HIPCHK(h, hipEventRecord(h->ev_front_done[par], fs));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_front_done[par], 0));
        }
        if (markers().on) {
            std::string st = op_stage(op.name);
            if (st != cur_stage) {
                if (!cur_stage.empty()) markers().pop();
                markers().push(st.c_str());
                cur_stage.swap(st);
            }
        }
        switch (op.kind) {
            case Op::STEM: {
                // bf16 inference on rows of <= 256 stem pixels: stem + zero-pad + max-pool in one kernel (the stem plane is never
                // written; aux_kernels.hip, BOD_STEM_POOL_FUSED=0: the two launches).  Training handles keep the plane (pool backward).
                const Op* pool = op_i < h->ops.size() && h->ops[op_i].kind == Op::POOL ? &h->ops[op_i] : nullptr;
                // Round 6: the (hi, lo) precisions take the same walk on three bf16 products (their stem was the exact fp32 kernel + a
                // pooling launch: 5.2 ms per 256 frames against 0.8); BOD_STEM_SPLIT_FUSED=0: as before.
                static const bool split_fused = [] { const char* e = getenv("BOD_STEM_SPLIT_FUSED"); return !e || atoi(e) != 0; }();
                stem_pool_fused = pool && !h->train && (h->split ? split_fused : h->es == 2) && !trace &&
                                  stem_pool_fused_applies(dev_images, c.batch, c.image_w, h->sw, is_front ? cus_front : cus_back);
                if (stem_pool_fused)
                    HIPCHK(h, launch_stem_pool_fused(dev_images, h->stem_w, h->stem_b, pool->conv.g[0].out, h->split ? 1 : 0, c.batch, c.image_h,
                                                     c.image_w, h->sh, h->sw, h->ph, h->pw, h->pw + 2, (h->ph + 2) * (h->pw + 2), st));
                else
                HIPCHK(h, launch_stem_conv(dev_images, h->stem_w, h->stem_b, h->stem_out, h->es == 4, c.batch, c.image_h,
                                           c.image_w, h->sh, h->sw, st));
                // the frames are consumed: the copy stream may refill this buffer.  (Training handles read the frames again in the
                // backward pass -- bf16 copy + stem weight gradient -- and record the event at the end of the step instead:
                // mark_images_consumed, train_impl.inc)
                if (!h->train) BODCHK(mark_images_consumed(h, st));
                break;
            }
            case Op::POOL:
                if (stem_pool_fused) break;
                HIPCHK(h, launch_stem_pool(h->stem_out, op.conv.g[0].out, h->split ? 2 : (h->es == 4 ? 1 : 0), c.batch,
                                           h->sh, h->sw, h->ph, h->pw, h->pw + 2, (h->ph + 2) * (h->pw + 2), st));
                break;
            case Op::CONV: {
                if (op.hx_pyramid) HIPCHK(h, launch_pairs_to_hx(h->pyramid.d, h->pyr_hx, (long)c.batch * h->Ppad, 256, st, h->pyr_fmt));
                op.conv.seed_lo = (uint32_t)seed; op.conv.seed_hi = (uint32_t)(seed >> 32);
                op.conv.image_base = first_image;
                op.conv.sample_base = (uint32_t)c.mc_sample_base;
                op.conv.dyn_rng = train_dyn_rng(h);              // training: device copy of {seed, image id} (graph replay)
                const bool is_tower = op.conv.xreuse != 0 && op.conv.fan_count <= 1;        // per-sample tower launches (one kernel symbol)
                const bool timed = h->profiling && op.is_head3x3 && (h->prof_which == 0 || (h->prof_which == 1) == is_tower);
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (timed) {
                    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
                    HIPCHK(h, hipEventRecord(e0, st));
                }
                op.conv.n_cu = is_front ? cus_front : cus_back;          // compute units this launch may fill (planner: workgroups vs CUs)
                {   // fan-out launch: CU de-phasing (conv_igemm.hip); BOD_FAN_STAGGER_US=t: quarter-tile delay in microseconds (0 = off)
                    static const int stagger_us = getenv("BOD_FAN_STAGGER_US") ? atoi(getenv("BOD_FAN_STAGGER_US")) : 0;
                    op.conv.stagger_ticks = (op.conv.fan_count > 1 && op.conv.xreuse) ? stagger_us * 100 : 0;
                }
                if (par != 0) {                                  // the second pyramid buffer of an overlap handle
                    ConvArgs a = op.conv;
                    for (int g = 0; g < a.groups; ++g) {
                        if (a.g[g].in == h->pyr_d[0]) a.g[g].in = h->pyr_d[1];
                        if (a.g[g].out == h->pyr_d[0]) a.g[g].out = h->pyr_d[1];
                    }
                    HIPCHK(h, (h->es == 4 && !h->split) ? launch_conv_igemm_f32(a, st) : launch_conv_igemm(a, st));
                } else
                HIPCHK(h, (h->es == 4 && !h->split) ? launch_conv_igemm_f32(op.conv, st) : launch_conv_igemm(op.conv, st));
                if (timed) {
                    HIPCHK(h, hipEventRecord(e1, st));
                    h->ev_head.emplace_back(e0, e1);
                    h->prof_flops += op.flops;
                }
                if (ov && first_back) {                          // the fan-out layer has read pyramid `par`
                    HIPCHK(h, hipEventRecord(h->ev_l0_done[par], h->stream));
                    h->l0_pending[par] = true;
                }
                break;
            }
        }
    }
    if (trace) {
        HIPCHK(h, hipEventRecord(tev[h->ops.size()], h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        fprintf(stderr, "# op trace: name kind M taps cin cout_pad groups fan ms TFLOP/s\n");
        for (size_t i = 0; i < h->ops.size(); ++i) {
            float ms = 0.f;
            HIPCHK(h, hipEventElapsedTime(&ms, tev[i], tev[i + 1]));
            const Op& o = h->ops[i];
            fprintf(stderr, "%-28s %d %8d %2d %5d %5d %d %2d %9.4f %8.1f\n", o.name.c_str(), (int)o.kind, o.conv.M, o.conv.taps,
                    o.conv.cin, o.conv.cout_pad, o.conv.groups, o.conv.fan_count, ms, o.flops / (ms * 1e-3) / 1e12);
        }
        for (hipEvent_t& e : tev) hipEventDestroy(e);
    }
    h->forward_done = true; h->posterior_done = h->nms_done = h->cluster_done = false; h->affinity_img = -1;
    h->last_seed = seed; h->last_first_image = first_image;
    if (flavour == FLAVOUR_AGG) { h->agg_valid = true; h->raw_valid = false; }
    else { h->raw_valid = true; if (!only_flavoured) h->agg_valid = false; }
    return BOD_OK;
}

// The raw head outputs of the last forward, for callers that ask for them after an aggregating bod_infer: the last tower
// layers' raw flavour is re-run on the activations still in HBM (same seed, same Philox streams: exactly the tensors a
// raw-flavour forward would have produced).
bod_status materialise_raw(bod_context* h) {
    if (h->raw_valid || !h->agg_valid) { BODCHK(ensure_raw(h)); return BOD_OK; }
    BODCHK(ensure_raw(h));
    const bool prof = h->profiling;
    h->profiling = false;
    const bool fd = h->forward_done, pd = h->posterior_done, nd = h->nms_done, cd = h->cluster_done;
    const int aff = h->affinity_img;
    const bod_status st = run_forward(h, h->cur_images, h->last_seed, h->last_first_image, FLAVOUR_RAW, true);
    h->profiling = prof;
    h->forward_done = fd; h->posterior_done = pd; h->nms_done = nd; h->cluster_done = cd; h->affinity_img = aff;
    return st;
}

bod_status run_posterior(bod_context* h, uint64_t seed, uint32_t first_image) {
    if (!h->anchors_ready) return h->fail(BOD_ERR_NOT_READY, "bod_set_anchors has not been called");
    for (int sidx = 0; sidx < 2; ++sidx)
        if (h->side_pending[sidx]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_done[sidx], 0));
    MarkerRange mr("bod:posterior");
    PostCfg pc = post_cfg(h, seed, first_image);
    PostBuffers pb = h->pb;
    pb.cls = h->raw[0]; pb.box = h->raw[1]; pb.cov = h->raw[2]; pb.anchors = h->d_anchors;
    pc.aggregated = (h->agg_valid && !h->raw_valid) ? 1 : 0;        // statistics from the conv epilogue, no [B,N,A,.] tensors
    pb.agg_cls = h->agg[0]; pb.agg_box = h->agg[1]; pb.agg_cov = h->agg[2];
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profiling) {
        HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    HIPCHK(h, launch_posterior(pc, pb, h->stream));
    if (h->cfg.ranking_method == BOD_RANK_JOINT_ENTROPY && h->cfg.gaussian_isotropic && h->cfg.dirichlet_non_informative)
        HIPCHK(h, launch_joint_entropy_rank(pc, pb, h->stream));
    if (h->profiling) { HIPCHK(h, hipEventRecord(e1, h->stream)); h->ev_post.emplace_back(e0, e1); }
    h->posterior_done = true; h->nms_done = h->cluster_done = false; h->affinity_img = -1;
    return BOD_OK;
}

bod_status run_nms(bod_context* h, hipStream_t st) {
    MarkerRange mr("bod:nms");
    const bod_config& c = h->cfg;
    NmsArgs a{};
    a.B = c.batch; a.A = h->A; a.num_kept = h->pb.num_kept; a.corners = h->pb.corners; a.ranking = h->pb.ranking;
    a.work_scores = h->nms_scores; a.work_begin = h->nms_begin; a.selected = h->nms_sel; a.num_selected = h->nms_nsel;
    a.max_out = c.nms_max_output_size; a.iou_thr = c.nms_iou_threshold; a.sigma = c.nms_soft_sigma; a.variant = c.nms_variant;
    HIPCHK(h, launch_nms(a, st));
    h->nms_done = true; h->cluster_done = false; h->affinity_img = -1;      // new centres: a pending bod_set_affinity was sized for the old ones
    return BOD_OK;
}

bod_status run_cluster(bod_context* h, hipStream_t st) {
    MarkerRange mr("bod:cluster_fuse");
    const bod_config& c = h->cfg;
    ClusterArgs a{};
    a.B = c.batch; a.A = h->A; a.C = c.num_classes; a.max_out = c.nms_max_output_size;
    a.num_kept = h->pb.num_kept; a.selected = h->nms_sel; a.num_selected = h->nms_nsel;
    a.corners = h->pb.corners; a.counts = h->pb.counts; a.means = h->pb.means; a.covs = h->pb.covs;
    a.thr = c.nms_iou_threshold;
    a.affinity = h->affinity_img >= 0 ? h->affinity : nullptr; a.affinity_img = h->affinity_img;
    h->affinity_img = -1;                         // consumed by this call
    a.out_scores = h->out_scores; a.out_means = h->out_means; a.out_covs = h->out_covs; a.out_counts = h->out_counts;
    HIPCHK(h, launch_cluster_fuse(a, st));
    h->cluster_done = true;
    return BOD_OK;
}

bod_status stage_images(bod_context* h, const float* images, int on_device, const float** dev, hipStream_t st = nullptr) {
    if (!images) return h->fail(BOD_ERR_INVALID_ARG, "images is NULL");
    if (!st) st = h->stream;                            // the stream the stem will run on
    h->cur_img_buf = -1;
    if (on_device) {
        for (int k = 0; k < 2; ++k)
            if (images == h->d_images_b[k] && h->d_images_b[k]) {
                h->cur_img_buf = k;
                if (h->img_ready_pending[k]) {           // filled by bod_upload_frames_u8_async on the copy stream
                    HIPCHK(h, hipStreamWaitEvent(st, h->ev_img_ready[k], 0));
                    h->img_ready_pending[k] = false;
                }
            }
        *dev = images;
        return BOD_OK;
    }
    const size_t bytes = (size_t)h->cfg.batch * h->cfg.image_h * h->cfg.image_w * 3 * sizeof(float);
    HIPCHK(h, hipMemcpyAsync(h->d_images, images, bytes, hipMemcpyHostToDevice, st));
    *dev = h->d_images;
    return BOD_OK;
}

// bod_infer reduces over the MC samples inside the conv epilogue whenever the plan has that flavour and this handle holds the
// whole ensemble (a handle computing a shard of a larger ensemble keeps the raw tensors: the exchange needs them)
int infer_flavour(bod_context* h) {
    const bod_config& c = h->cfg;
    const bool whole = c.mc_sample_base == 0 && (c.mc_ensemble_size == 0 || c.mc_ensemble_size == c.mc_samples);
    return (h->agg_plan && whole && h->anchors_ready) ? FLAVOUR_AGG : FLAVOUR_RAW;
}

template <typename T>
bod_status d2h(bod_context* h, T* dst, const T* src, size_t n) {
    if (!dst || n == 0) return BOD_OK;
    HIPCHK(h, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    return BOD_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* bod_version(void) { return "bayesod-hip 0.1.0 (gfx950)"; }

const char* bod_last_error(bod_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

bod_status bod_create(const bod_config* cfg, bod_handle* out) {
    if (!cfg || !out) { g_create_error = "bod_create: NULL argument"; return BOD_ERR_INVALID_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_create_error = "no HIP device visible: libbayesod_hip has no CPU fallback";
        return BOD_ERR_NO_DEVICE;
    }
    std::unique_ptr<bod_context> h(new bod_context());
    h->cfg = *cfg;
    auto bail = [&](bod_status s) { g_create_error = h->err; return s; };
    const bod_config& c = h->cfg;
    if (c.device < 0 || c.device >= ndev) return bail(h->fail(BOD_ERR_INVALID_ARG, "device %d out of range [0,%d)", c.device, ndev));
    if (c.batch < 1 || c.batch > 4096 || c.mc_samples < 1 || c.mc_samples > 4096)
        return bail(h->fail(BOD_ERR_INVALID_ARG, "batch=%d / mc_samples=%d out of range", c.batch, c.mc_samples));
    if (c.mc_sample_base < 0 || c.mc_sample_base + c.mc_samples > 65535 || c.mc_ensemble_size < 0 ||
        (c.mc_ensemble_size > 0 && c.mc_sample_base + c.mc_samples > c.mc_ensemble_size))
        return bail(h->fail(BOD_ERR_INVALID_ARG, "mc_sample_base=%d / mc_samples=%d / mc_ensemble_size=%d inconsistent",
                            c.mc_sample_base, c.mc_samples, c.mc_ensemble_size));
    if (c.training && ((c.precision != BOD_PRECISION_BF16 && c.precision != BOD_PRECISION_FP32) || c.mc_samples != 1))
        return bail(h->fail(BOD_ERR_INVALID_ARG, "training handles run in bf16 (or, for gradient verification, fp32) precision with mc_samples = 1 (dropout stays on)"));
    if (c.backbone_depth != 0 && c.backbone_depth != 50 && c.backbone_depth != 101)
        return bail(h->fail(BOD_ERR_INVALID_ARG, "backbone_depth must be 50 or 101, got %d", c.backbone_depth));
    if (c.num_classes != 4 && c.num_classes != 8)
        return bail(h->fail(BOD_ERR_INVALID_ARG, "num_classes (incl. background) must be 4 or 8, got %d", c.num_classes));
    if (c.anchors_per_location < 1 || (c.anchors_per_location * c.num_classes) % 4 != 0)
        return bail(h->fail(BOD_ERR_INVALID_ARG, "anchors_per_location=%d unsupported", c.anchors_per_location));
    if (!(c.dropout_rate >= 0.f && c.dropout_rate < 1.f)) return bail(h->fail(BOD_ERR_INVALID_ARG, "dropout_rate must be in [0,1)"));
    if (c.num_categorical_draws < 1 || c.num_categorical_draws > 1024) return bail(h->fail(BOD_ERR_INVALID_ARG, "num_categorical_draws out of range"));
    if (c.nms_max_output_size < 1 || c.nms_max_output_size > 512) return bail(h->fail(BOD_ERR_INVALID_ARG, "nms_max_output_size must be in [1,512]"));
    if (c.precision != BOD_PRECISION_BF16 && c.precision != BOD_PRECISION_FP32 && c.precision != BOD_PRECISION_BF16X3 && c.precision != BOD_PRECISION_F16MX && c.precision != BOD_PRECISION_F16MX4)
        return bail(h->fail(BOD_ERR_INVALID_ARG, "precision must be BOD_PRECISION_BF16 (0), BOD_PRECISION_FP32 (1), BOD_PRECISION_BF16X3 (2), BOD_PRECISION_F16MX (3) or BOD_PRECISION_F16MX4 (4)"));
    h->es = c.precision == BOD_PRECISION_BF16 ? 2 : 4;
    h->split = c.precision == BOD_PRECISION_BF16X3 || c.precision == BOD_PRECISION_F16MX || c.precision == BOD_PRECISION_F16MX4;
    h->mx = c.precision == BOD_PRECISION_F16MX ? 1 : c.precision == BOD_PRECISION_F16MX4 ? 2 : 0;
    if (hipSetDevice(c.device) != hipSuccess) return bail(h->fail(BOD_ERR_HIP, "hipSetDevice(%d) failed", c.device));
    // Development switches of the section-8.4 probes (tests/tools/selfcheck_probe.py), read when the handle is created:
    //   BOD_CU_MASK_SLOTS=lo:hi  the handle's main stream may use CU slots [lo, hi) of every XCD only (hipExtStreamCreateWithCUMask);
    //   BOD_FORWARD_OPS=lo:hi    bod_forward runs ops [lo, hi) of the plan only (a company of chosen kernels; the outputs are garbage).
    if (const char* e = getenv("BOD_FORWARD_OPS")) { int lo = 0, hi = 0; if (sscanf(e, "%d:%d", &lo, &hi) == 2 && lo >= 0 && hi > lo) { h->dev_op_lo = lo; h->dev_op_hi = hi; } }
    bool masked_main = false;
    // (BOD_MAIN_CUS_PER_XCD=k, the older spelling: slots [0, k))
    const std::string mask_env = getenv("BOD_CU_MASK_SLOTS") ? std::string(getenv("BOD_CU_MASK_SLOTS")) :
                                 getenv("BOD_MAIN_CUS_PER_XCD") ? "0:" + std::to_string(std::max(1, std::min(32, atoi(getenv("BOD_MAIN_CUS_PER_XCD"))))) : std::string();
    if (!mask_env.empty()) {
        const char* e = mask_env.c_str();
        int lo = 0, hi = 0;
        hipDeviceProp_t prop;
        if (sscanf(e, "%d:%d", &lo, &hi) == 2 && hipGetDeviceProperties(&prop, c.device) == hipSuccess && prop.multiProcessorCount % 8 == 0 &&
            lo >= 0 && hi > lo && hi <= prop.multiProcessorCount / 8) {
            uint32_t m[8];
            cu_slot_mask(lo, hi, m);
            if (hipExtStreamCreateWithCUMask(&h->stream, 8, m) != hipSuccess) return bail(h->fail(BOD_ERR_HIP, "hipExtStreamCreateWithCUMask failed"));
            masked_main = true;
        } else return bail(h->fail(BOD_ERR_INVALID_ARG, "BOD_CU_MASK_SLOTS=%s: want lo:hi within the XCD's CU slots", e));
    }
    if (!masked_main && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(h->fail(BOD_ERR_HIP, "hipStreamCreate failed"));
    if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) return bail(h->fail(BOD_ERR_HIP, "hipStreamCreate failed"));
    h->full = h->stream;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c.device) != hipSuccess) return bail(h->fail(BOD_ERR_HIP, "hipGetDeviceProperties failed"));
        h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        // pipeline overlap: BOD_OVERLAP=0 forces it off, =1 on (CU-masked streams), =2 on with two plain streams (A/B aid);
        // BOD_OVERLAP_FRONT_SLOTS=k: CU slots per XCD the front owns (default 4; the workgroup dispatcher balances over the four
        // shader engines of an XCD, so only multiples of 4 change anything: tests/tools/cu_mask_probe.hip)
        int want = c.pipeline_overlap ? 1 : 0;
        if (const char* e = getenv("BOD_OVERLAP")) want = atoi(e);
        if (c.training) want = 0;
        // Round 6: the mode runs kernels of this library beside each other by design, and those can miscompute a 16-lane row
        //  -- it is slower than one stream anyway (8.3).  An experiment, never a default: refused without the switch.
        if (want && !(getenv("BOD_OVERLAP_EXPERIMENTAL") && atoi(getenv("BOD_OVERLAP_EXPERIMENTAL")) == 1))
            return bail(h->fail(BOD_ERR_INVALID_ARG, "pipeline_overlap is experimental (slower than one stream and not bit-reproducible: include/bayesod.h, "
                                                     "DESIGN.md 8.3-8.4): set BOD_OVERLAP_EXPERIMENTAL=1 to create such a handle"));
        if (want) {
            const int slots = h->n_cu / 8;
            int fs = getenv("BOD_OVERLAP_FRONT_SLOTS") ? atoi(getenv("BOD_OVERLAP_FRONT_SLOTS")) : 4;
            fs = std::max(1, std::min(fs, slots - 1));
            const bool masks = want == 1 && h->n_cu % 8 == 0 && slots >= 8 && slots <= 32;
            if (masks) {
                uint32_t mf[8], mb[8];
                cu_slot_mask(slots - fs, slots, mf);
                cu_slot_mask(0, slots - fs, mb);
                if (hipExtStreamCreateWithCUMask(&h->front, 8, mf) != hipSuccess || hipExtStreamCreateWithCUMask(&h->back, 8, mb) != hipSuccess)
                    return bail(h->fail(BOD_ERR_HIP, "hipExtStreamCreateWithCUMask failed"));
                h->overlap_mode = 1; h->ov_front_slots = fs; h->ov_slots = slots;
            } else {
                if (hipStreamCreateWithFlags(&h->front, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&h->back, hipStreamNonBlocking) != hipSuccess)
                    return bail(h->fail(BOD_ERR_HIP, "hipStreamCreate failed"));
                h->overlap_mode = 2; h->ov_front_slots = 0; h->ov_slots = slots;
            }
            for (int k = 0; k < 2; ++k)
                if (hipEventCreateWithFlags(&h->ev_front_done[k], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&h->ev_l0_done[k], hipEventDisableTiming) != hipSuccess)
                    return bail(h->fail(BOD_ERR_HIP, "hipEventCreate failed"));
            if (hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) return bail(h->fail(BOD_ERR_HIP, "hipEventCreate failed"));
        }
    }
    if (hipEventCreateWithFlags(&h->ev_posterior, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_done[1], hipEventDisableTiming) != hipSuccess)
        return bail(h->fail(BOD_ERR_HIP, "hipEventCreate failed"));
    bod_status s = build_geometry(h.get());
    if (s != BOD_OK) return bail(s);
    if ((int64_t)c.batch * c.mc_samples * h->Ppad >= (1LL << 31))
        return bail(h->fail(BOD_ERR_INVALID_ARG, "batch*mc_samples*pixels exceeds 2^31 rows"));
    s = alloc_post(h.get());
    if (s != BOD_OK) return bail(s);
    *out = h.release();
    return BOD_OK;
}

bod_status bod_destroy(bod_handle h) {
    if (!h) return BOD_OK;
    hipSetDevice(h->cfg.device);
    h->stream = h->full;
    if (h->front) { hipStreamSynchronize(h->front); hipStreamDestroy(h->front); }
    if (h->back) { hipStreamSynchronize(h->back); hipStreamDestroy(h->back); }
    for (int k = 0; k < 2; ++k) { if (h->ev_front_done[k]) hipEventDestroy(h->ev_front_done[k]); if (h->ev_l0_done[k]) hipEventDestroy(h->ev_l0_done[k]); }
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->side) { hipStreamSynchronize(h->side); hipStreamDestroy(h->side); }
    if (h->ev_posterior) hipEventDestroy(h->ev_posterior);
    for (int sidx = 0; sidx < 2; ++sidx) if (h->ev_done[sidx]) hipEventDestroy(h->ev_done[sidx]);
    train_destroy(h);
    for (int sidx = 0; sidx < 2; ++sidx) if (h->host_stage[sidx]) hipHostFree(h->host_stage[sidx]);
    if (h->rec_recv && h->rec_recv != h->rec_send) hipFree(h->rec_recv);
    if (h->rec_send) hipFree(h->rec_send);
    if (h->ev_gather) hipEventDestroy(h->ev_gather);
    for (void* p : h->allocs) hipFree(p);
    if (h->iou_scratch) hipFree(h->iou_scratch);
    if (h->affinity) hipFree(h->affinity);
    if (h->d_frames_u8) hipFree(h->d_frames_u8);
    if (h->copy) { hipStreamSynchronize(h->copy); hipStreamDestroy(h->copy); }
    for (int k = 0; k < 2; ++k) {
        if (h->d_u8_b[k]) hipFree(h->d_u8_b[k]);
        if (h->ev_img_ready[k]) hipEventDestroy(h->ev_img_ready[k]);
        if (h->ev_img_free[k]) hipEventDestroy(h->ev_img_free[k]);
    }
    for (auto& e : h->ev_head) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& e : h->ev_post) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return BOD_OK;
}

bod_status bod_query_sizes(bod_handle h, bod_sizes* out) {
    if (!h || !out) return BOD_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    out->num_pixels = h->P; out->num_anchors = h->A; out->num_levels = h->nlev;
    for (int l = 0; l < h->nlev; ++l) { out->level_h[l] = h->lh[l]; out->level_w[l] = h->lw[l]; }
    out->max_detections = h->cfg.nms_max_output_size;
    out->device_bytes = h->device_bytes;
    return BOD_OK;
}

bod_status bod_update_config(bod_handle h, const bod_config* cfg) {
    if (!h || !cfg) return BOD_ERR_INVALID_ARG;
    const bod_config& o = h->cfg;
    if (cfg->device != o.device || cfg->image_h != o.image_h || cfg->image_w != o.image_w || cfg->batch != o.batch ||
        cfg->mc_samples != o.mc_samples || cfg->num_classes != o.num_classes ||
        cfg->anchors_per_location != o.anchors_per_location || cfg->min_level != o.min_level ||
        cfg->max_level != o.max_level || cfg->has_covar_head != o.has_covar_head || cfg->dropout_rate != o.dropout_rate ||
        cfg->precision != o.precision || cfg->training != o.training || cfg->backbone_depth != o.backbone_depth || cfg->pipeline_overlap != o.pipeline_overlap || (std::max(cfg->mc_ensemble_size, cfg->mc_samples) > 1) != (std::max(o.mc_ensemble_size, o.mc_samples) > 1))
        return h->fail(BOD_ERR_INVALID_ARG, "bod_update_config: geometry / model fields cannot change on a live handle");
    if (cfg->nms_max_output_size != o.nms_max_output_size)
        return h->fail(BOD_ERR_INVALID_ARG, "bod_update_config: nms_max_output_size sizes device buffers and cannot change");
    if (cfg->num_categorical_draws < 1 || cfg->num_categorical_draws > 1024)
        return h->fail(BOD_ERR_INVALID_ARG, "num_categorical_draws out of range");
    if (cfg->mc_sample_base < 0 || cfg->mc_sample_base + cfg->mc_samples > 65535)
        return h->fail(BOD_ERR_INVALID_ARG, "mc_sample_base + mc_samples must stay below 65536 (16-bit sample field of the RNG counter)");
    h->cfg = *cfg;
    return BOD_OK;
}

bod_status bod_load_weight(bod_handle h, const char* name, int32_t kind, const int64_t* shape, int32_t ndim, const float* data) {
    if (!h) return BOD_ERR_INVALID_ARG;
    if (!name || !shape || !data || ndim < 1 || ndim > 4 || kind < 0 || kind > 5)
        return h->fail(BOD_ERR_INVALID_ARG, "bod_load_weight: bad argument");
    if ((kind == 0) != (ndim == 4)) return h->fail(BOD_ERR_INVALID_ARG, "bod_load_weight('%s'): kind %d needs ndim %d", name, kind, kind == 0 ? 4 : 1);
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] <= 0) return h->fail(BOD_ERR_INVALID_ARG, "non-positive dim"); t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(data, data + n);
    h->host_w[std::string(name) + "/" + std::to_string(kind)] = std::move(t);
    h->weights_ready = false;
    return BOD_OK;
}

bod_status bod_finalize_weights(bod_handle h) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->weights_ready) return BOD_OK;
    if (!h->ops.empty()) return h->fail(BOD_ERR_INVALID_ARG, "weights were already finalized; create a new handle to reload");
    BODCHK(build_plan(h));
    if (h->cfg.training) BODCHK(train_init(h));
    h->host_w.clear();
    h->weights_ready = true;
    return BOD_OK;
}

bod_status bod_set_anchors(bod_handle h, const float* anchors, int32_t n) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!anchors || n != h->A) return h->fail(BOD_ERR_INVALID_ARG, "bod_set_anchors: expected %d anchors, got %d", h->A, n);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_anchors, anchors, (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->anchors_ready = true;
    return BOD_OK;
}

bod_status bod_upload_images(bod_handle h, const float* host_images) {
    if (!h || !host_images) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t bytes = (size_t)h->cfg.batch * h->cfg.image_h * h->cfg.image_w * 3 * sizeof(float);
    HIPCHK(h, hipMemcpyAsync(h->d_images, host_images, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BOD_OK;
}

static bod_status preproc_geometry(bod_handle h, int32_t src_h, int32_t src_w, const float* rgb_means, int32_t aspect_resize, PreprocArgs* out) {
    const bod_config& c = h->cfg;
    PreprocArgs a{};
    a.B = c.batch; a.sh = src_h; a.sw = src_w; a.H = c.image_h; a.W = c.image_w; a.resize = aspect_resize ? 1 : 0;
    a.rh = src_h; a.rw = src_w;
    if (aspect_resize) {
        // tf.image.resize(..., preserve_aspect_ratio=True): scale = min(H/sh, W/sw) in float32, size = round(s * in)
        const float fh = (float)c.image_h / (float)src_h, fw = (float)c.image_w / (float)src_w;
        const float sc = fh < fw ? fh : fw;
        a.rh = (int32_t)std::nearbyint(sc * (float)src_h);
        a.rw = (int32_t)std::nearbyint(sc * (float)src_w);
        if (a.rh < 1 || a.rw < 1) return h->fail(BOD_ERR_INVALID_ARG, "bod_upload_frames_u8: degenerate resize %dx%d", a.rh, a.rw);
    } else if (src_h != c.image_h || src_w != c.image_w) {
        return h->fail(BOD_ERR_INVALID_ARG, "bod_upload_frames_u8: frames are %dx%d but the handle expects %dx%d (pass aspect_resize=1 for "
                       "the KITTI-style resize + crop/pad)", src_h, src_w, c.image_h, c.image_w);
    }
    a.scale_y = (float)src_h / (float)a.rh; a.scale_x = (float)src_w / (float)a.rw;
    const int dh = c.image_h - a.rh, dw = c.image_w - a.rw;                 // resize_with_crop_or_pad (floor division like Python)
    auto fdiv2 = [](int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };
    a.crop_y = std::max(fdiv2(-dh), 0); a.crop_x = std::max(fdiv2(-dw), 0);
    a.pad_y = std::max(fdiv2(dh), 0); a.pad_x = std::max(fdiv2(dw), 0);
    a.vis_h = std::min(a.rh, c.image_h); a.vis_w = std::min(a.rw, c.image_w);
    for (int k = 0; k < 3; ++k) a.mean[k] = rgb_means[k];
    *out = a;
    return BOD_OK;
}

bod_status bod_upload_frames_u8(bod_handle h, const uint8_t* rgb, int32_t src_h, int32_t src_w, const float* rgb_means,
                                int32_t aspect_resize) {
    MarkerRange mr_api("bod:upload");
    if (!h || !rgb || !rgb_means || src_h < 1 || src_w < 1) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    const bod_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device));
    PreprocArgs a{};
    BODCHK(preproc_geometry(h, src_h, src_w, rgb_means, aspect_resize, &a));
    const size_t bytes = (size_t)c.batch * src_h * src_w * 3;
    if (bytes > h->frames_u8_cap) {
        if (h->d_frames_u8) { HIPCHK(h, hipStreamSynchronize(h->stream)); hipFree(h->d_frames_u8); h->d_frames_u8 = nullptr; h->frames_u8_cap = 0; }
        if (hipMalloc(reinterpret_cast<void**>(&h->d_frames_u8), bytes) != hipSuccess)
            return h->fail(BOD_ERR_OOM, "bod_upload_frames_u8: %zu bytes of staging", bytes);
        h->frames_u8_cap = bytes;
    }
    a.src = h->d_frames_u8; a.dst = h->d_images;
    HIPCHK(h, hipMemcpyAsync(h->d_frames_u8, rgb, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_preprocess(a, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));       // the caller may reuse `rgb` on return
    return BOD_OK;
}

bod_status bod_upload_frames_u8_async(bod_handle h, const uint8_t* rgb, int32_t src_h, int32_t src_w, const float* rgb_means,
                                      int32_t aspect_resize, int32_t buffer) {
    MarkerRange mr_api("bod:upload");
    if (!h || !rgb || !rgb_means || src_h < 1 || src_w < 1) return BOD_ERR_INVALID_ARG;
    if (buffer < 0 || buffer > 1) return h->fail(BOD_ERR_INVALID_ARG, "bod_upload_frames_u8_async: buffer must be 0 or 1");
    const bod_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device));
    PreprocArgs a{};
    BODCHK(preproc_geometry(h, src_h, src_w, rgb_means, aspect_resize, &a));
    if (!h->copy) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->copy, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_img_ready[k], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_img_free[k], hipEventDisableTiming));
        }
    }
    // (no zero fill: dalloc's memset would run on the main stream, behind the forward in flight, and land on the frames)
    if (!h->d_images_b[buffer]) BODCHK(h->dalloc(&h->d_images_b[buffer], (size_t)c.batch * c.image_h * c.image_w * 3, false));
    const size_t bytes = (size_t)c.batch * src_h * src_w * 3;
    if (bytes > h->u8_cap_b[buffer]) {
        HIPCHK(h, hipStreamSynchronize(h->copy));
        if (h->d_u8_b[buffer]) hipFree(h->d_u8_b[buffer]);
        h->d_u8_b[buffer] = nullptr; h->u8_cap_b[buffer] = 0;
        if (hipMalloc(reinterpret_cast<void**>(&h->d_u8_b[buffer]), bytes) != hipSuccess)
            return h->fail(BOD_ERR_OOM, "bod_upload_frames_u8_async: %zu bytes of staging", bytes);
        h->u8_cap_b[buffer] = bytes;
    }
    // do not overwrite frames a forward pass still has to read (its stem records ev_img_free), nor frames of an upload
    // nobody consumed yet (same stream: ordered)
    if (h->img_free_pending[buffer]) { HIPCHK(h, hipStreamWaitEvent(h->copy, h->ev_img_free[buffer], 0)); h->img_free_pending[buffer] = false; }
    else if (buffer == 0) {                                                   // buffer 0 doubles as the synchronous d_images
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->front) HIPCHK(h, hipStreamSynchronize(h->front));
    }
    a.src = h->d_u8_b[buffer]; a.dst = h->d_images_b[buffer];
    HIPCHK(h, hipMemcpyAsync(h->d_u8_b[buffer], rgb, bytes, hipMemcpyHostToDevice, h->copy));
    HIPCHK(h, launch_preprocess(a, h->copy));
    HIPCHK(h, hipEventRecord(h->ev_img_ready[buffer], h->copy));
    h->img_ready_pending[buffer] = true;
    return BOD_OK;
}

const float* bod_device_images_buffer(bod_handle h, int32_t buffer) {
    return (h && buffer >= 0 && buffer <= 1) ? h->d_images_b[buffer] : nullptr;
}

const float* bod_device_images(bod_handle h) { return h ? h->d_images : nullptr; }

bod_status bod_synchronize(bod_handle h) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (h->copy) HIPCHK(h, hipStreamSynchronize(h->copy));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->side));
    h->side_pending[0] = h->side_pending[1] = false;
    return BOD_OK;
}

bod_status bod_forward(bod_handle h, const float* images, int32_t on_device, uint64_t seed, uint32_t first_image_id) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->weights_ready) return h->fail(BOD_ERR_NOT_READY, "weights not finalized");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const float* dev = nullptr;
    BODCHK(stage_images(h, images, on_device, &dev));
    h->cur_images = dev;
    if (h->train) return train_forward_only(h, dev, seed, first_image_id);     // model(x, 'training'): dropout on, N = 1
    return run_forward(h, dev, seed, first_image_id);
}

bod_status bod_get_raw(bod_handle h, float* cls, float* box, float* cov) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->forward_done) return h->fail(BOD_ERR_NOT_READY, "bod_forward has not run");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    BODCHK(materialise_raw(h));
    const bod_config& c = h->cfg;
    const size_t n = (size_t)c.batch * c.mc_samples * h->A;
    BODCHK(d2h(h, cls, h->raw[0], n * c.num_classes));
    BODCHK(d2h(h, box, h->raw[1], n * 4));
    if (cov) {
        if (!c.has_covar_head) return h->fail(BOD_ERR_INVALID_ARG, "model has no covariance head");
        BODCHK(d2h(h, cov, h->raw[2], n * 10));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BOD_OK;
}

bod_status bod_set_raw(bod_handle h, const float* cls, const float* box, const float* cov) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    HIPCHK(h, hipSetDevice(h->cfg.device));
    BODCHK(ensure_raw(h));
    const bod_config& c = h->cfg;
    const size_t n = (size_t)c.batch * c.mc_samples * h->A;
    if (cls) HIPCHK(h, hipMemcpyAsync(h->raw[0], cls, n * c.num_classes * 4, hipMemcpyHostToDevice, h->stream));
    if (box) HIPCHK(h, hipMemcpyAsync(h->raw[1], box, n * 16, hipMemcpyHostToDevice, h->stream));
    if (cov && c.has_covar_head) HIPCHK(h, hipMemcpyAsync(h->raw[2], cov, n * 40, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->forward_done = true; h->posterior_done = h->nms_done = h->cluster_done = false; h->affinity_img = -1;
    h->raw_valid = true; h->agg_valid = false;
    return BOD_OK;
}

bod_status bod_get_pyramid(bod_handle h, int32_t l, float* out) {
    if (!h || !out) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->forward_done || !h->pyramid.d) return h->fail(BOD_ERR_NOT_READY, "bod_forward has not run");
    if (l < 0 || l >= h->nlev) return h->fail(BOD_ERR_INVALID_ARG, "level index %d out of range", l);
    const int B = h->cfg.batch;
    std::vector<char> tmp((size_t)B * h->Ppad * 256 * h->es);
    HIPCHK(h, hipMemcpyAsync(tmp.data(), h->pyr_d[h->pyr_last] ? h->pyr_d[h->pyr_last] : h->pyramid.d, tmp.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int hh = h->lh[l], ww = h->lw[l], pitch = ww + 2;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < ww; ++x) {
                const size_t pix = ((size_t)b * h->Ppad + h->lvl_off[l] + (size_t)(y + 1) * pitch + (x + 1)) * 256;
                float* d = out + (((size_t)b * hh + y) * ww + x) * 256;
                if (h->split) {
                    const uint16_t* s = reinterpret_cast<const uint16_t*>(tmp.data()) + pix * 2;
                    for (int ch = 0; ch < 256; ++ch) { const int slot = (ch >> 5) * 64 + (ch & 31); d[ch] = bf2f(s[slot]) + bf2f(s[slot + 32]); }
                } else if (h->es == 2) {
                    const uint16_t* s = reinterpret_cast<const uint16_t*>(tmp.data()) + pix;
                    for (int ch = 0; ch < 256; ++ch) d[ch] = bf2f(s[ch]);
                } else {
                    std::memcpy(d, reinterpret_cast<const float*>(tmp.data()) + pix, 256 * sizeof(float));
                }
            }
    return BOD_OK;
}

bod_status bod_posterior(bod_handle h, uint64_t seed, uint32_t first_image_id) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->forward_done) return h->fail(BOD_ERR_NOT_READY, "bod_forward / bod_set_raw has not run");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return run_posterior(h, seed, first_image_id);
}

bod_status bod_validation_post(bod_handle h) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->forward_done) return h->fail(BOD_ERR_NOT_READY, "bod_forward / bod_set_raw has not run");
    if (!h->anchors_ready) return h->fail(BOD_ERR_NOT_READY, "bod_set_anchors has not been called");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    BODCHK(materialise_raw(h));
    for (int sidx = 0; sidx < 2; ++sidx)
        if (h->side_pending[sidx]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_done[sidx], 0));
    PostCfg pc = post_cfg(h, 0, 0);
    PostBuffers pb = h->pb;
    pb.cls = h->raw[0]; pb.box = h->raw[1]; pb.cov = h->raw[2]; pb.anchors = h->d_anchors;
    HIPCHK(h, launch_validation_post(pc, pb, h->stream));
    h->posterior_done = true; h->nms_done = h->cluster_done = false; h->affinity_img = -1;
    return BOD_OK;
}

bod_status bod_get_num_kept(bod_handle h, int32_t* out) {
    if (!h || !out) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->posterior_done) return h->fail(BOD_ERR_NOT_READY, "bod_posterior has not run");
    BODCHK(d2h(h, out, h->pb.num_kept, (size_t)h->cfg.batch));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BOD_OK;
}

static bod_status image_m(bod_handle h, int32_t img, int32_t* m) {
    if (img < 0 || img >= h->cfg.batch) return h->fail(BOD_ERR_INVALID_ARG, "image index %d out of range", img);
    HIPCHK(h, hipMemcpyAsync(m, h->pb.num_kept + img, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BOD_OK;
}

bod_status bod_get_posterior(bod_handle h, int32_t img, float* counts, float* score, float* means, float* covs, float* ranking, int32_t* anchor_index) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->posterior_done) return h->fail(BOD_ERR_NOT_READY, "bod_posterior has not run");
    int32_t m = 0;
    BODCHK(image_m(h, img, &m));
    const size_t o = (size_t)img * h->A, C = h->cfg.num_classes;
    BODCHK(d2h(h, counts, h->pb.counts + o * C, m * C));
    BODCHK(d2h(h, score, h->pb.score + o * C, m * C));
    BODCHK(d2h(h, means, h->pb.means + o * 4, (size_t)m * 4));
    BODCHK(d2h(h, covs, h->pb.covs + o * 16, (size_t)m * 16));
    BODCHK(d2h(h, ranking, h->pb.ranking + o, (size_t)m));
    BODCHK(d2h(h, anchor_index, h->pb.anchor_index + o, (size_t)m));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BOD_OK;
}

bod_status bod_set_posterior(bod_handle h, int32_t img, int32_t m, const float* counts, const float* means, const float* covs, const float* ranking) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (img < 0 || img >= h->cfg.batch || m < 0 || m > h->A) return h->fail(BOD_ERR_INVALID_ARG, "bod_set_posterior: bad image index / M");
    if (m > 0 && (!counts || !means || !covs || !ranking)) return h->fail(BOD_ERR_INVALID_ARG, "bod_set_posterior: NULL array");
    const size_t o = (size_t)img * h->A, C = h->cfg.num_classes;
    std::vector<float> corners((size_t)m * 4);
    for (int i = 0; i < m; ++i) {
        const float v = means[i * 4], u = means[i * 4 + 1], hh = means[i * 4 + 2], ww = means[i * 4 + 3];
        corners[i * 4] = v - hh / 2.0f; corners[i * 4 + 1] = u - ww / 2.0f;
        corners[i * 4 + 2] = v + hh / 2.0f; corners[i * 4 + 3] = u + ww / 2.0f;
    }
    if (m > 0) {
        HIPCHK(h, hipMemcpyAsync(h->pb.counts + o * C, counts, m * C * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->pb.means + o * 4, means, (size_t)m * 16, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->pb.covs + o * 16, covs, (size_t)m * 64, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->pb.ranking + o, ranking, (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->pb.corners + o * 4, corners.data(), (size_t)m * 16, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipMemcpyAsync(h->pb.num_kept + img, &m, 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->posterior_done = true; h->nms_done = h->cluster_done = false; h->affinity_img = -1;
    return BOD_OK;
}

bod_status bod_nms(bod_handle h) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->posterior_done) return h->fail(BOD_ERR_NOT_READY, "bod_posterior has not run");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return run_nms(h, h->stream);
}

bod_status bod_get_nms(bod_handle h, int32_t img, int32_t* indices, int32_t* num) {
    if (!h || !num) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->nms_done) return h->fail(BOD_ERR_NOT_READY, "bod_nms has not run");
    if (img < 0 || img >= h->cfg.batch) return h->fail(BOD_ERR_INVALID_ARG, "image index out of range");
    HIPCHK(h, hipMemcpyAsync(num, h->nms_nsel + img, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    BODCHK(d2h(h, indices, h->nms_sel + (size_t)img * h->cfg.nms_max_output_size, (size_t)*num));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BOD_OK;
}

bod_status bod_set_nms(bod_handle h, int32_t img, const int32_t* indices, int32_t n) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->posterior_done) return h->fail(BOD_ERR_NOT_READY, "bod_posterior / bod_set_posterior has not run");
    if (img < 0 || img >= h->cfg.batch || n < 0 || n > h->cfg.nms_max_output_size || (n > 0 && !indices))
        return h->fail(BOD_ERR_INVALID_ARG, "bod_set_nms: bad image index or count (max %d)", h->cfg.nms_max_output_size);
    int32_t m = 0;
    BODCHK(image_m(h, img, &m));
    for (int i = 0; i < n; ++i)
        if (indices[i] < 0 || indices[i] >= m) return h->fail(BOD_ERR_INVALID_ARG, "cluster centre %d out of range [0,%d)", indices[i], m);
    if (n > 0)
        HIPCHK(h, hipMemcpyAsync(h->nms_sel + (size_t)img * h->cfg.nms_max_output_size, indices, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->nms_nsel + img, &n, 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->nms_done = true; h->cluster_done = false; h->affinity_img = -1;      // new centres: a pending bod_set_affinity was sized for the old ones
    return BOD_OK;
}

bod_status bod_get_iou_matrix(bod_handle h, int32_t img, float* iou) {
    if (!h || !iou) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->posterior_done) return h->fail(BOD_ERR_NOT_READY, "bod_posterior has not run");
    int32_t m = 0;
    BODCHK(image_m(h, img, &m));
    if (m == 0) return BOD_OK;
    const int64_t need = (int64_t)m * m;
    if (need > h->iou_cap) {
        if (h->iou_scratch) hipFree(h->iou_scratch);
        h->iou_scratch = nullptr; h->iou_cap = 0;
        if (hipMalloc((void**)&h->iou_scratch, (size_t)need * 4) != hipSuccess)
            return h->fail(BOD_ERR_OOM, "cannot allocate %lld-element IoU matrix", (long long)need);
        h->iou_cap = need;
    }
    HIPCHK(h, launch_iou_matrix(h->pb.corners + (size_t)img * h->A * 4, m, h->iou_scratch, h->stream));
    BODCHK(d2h(h, iou, h->iou_scratch, (size_t)need));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BOD_OK;
}

bod_status bod_set_affinity(bod_handle h, int32_t img, const float* centre_columns, int32_t k, int32_t m) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->nms_done) return h->fail(BOD_ERR_NOT_READY, "bod_nms / bod_set_nms has not run");
    if (img < 0 || img >= h->cfg.batch || !centre_columns) return h->fail(BOD_ERR_INVALID_ARG, "bod_set_affinity: bad image index / NULL");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int32_t mm = 0, kk = 0;
    BODCHK(image_m(h, img, &mm));
    HIPCHK(h, hipMemcpyAsync(&kk, h->nms_nsel + img, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (m != mm || k != kk)
        return h->fail(BOD_ERR_INVALID_ARG, "bod_set_affinity: got %d columns of %d rows, the image has %d centres and %d boxes", k, m, kk, mm);
    if (!h->affinity && hipMalloc((void**)&h->affinity, (size_t)h->cfg.nms_max_output_size * h->A * 4) != hipSuccess)
        return h->fail(BOD_ERR_OOM, "bod_set_affinity: %zu bytes", (size_t)h->cfg.nms_max_output_size * h->A * 4);
    for (int r = 0; r < k && m > 0; ++r)          // row r -> affinity[r][0..m): one plain copy per centre (k <= max_detections)
        HIPCHK(h, hipMemcpyAsync(h->affinity + (size_t)r * h->A, centre_columns + (size_t)r * m, (size_t)m * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->affinity_img = img;
    return BOD_OK;
}

bod_status bod_cluster_fuse(bod_handle h) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->nms_done) return h->fail(BOD_ERR_NOT_READY, "bod_nms has not run");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return run_cluster(h, h->stream);
}

bod_status bod_get_detections(bod_handle h, int32_t img, int32_t* num, float* scores, float* means, float* covs, float* counts) {
    if (!h || !num) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->cluster_done) return h->fail(BOD_ERR_NOT_READY, "bod_cluster_fuse has not run");
    if (img < 0 || img >= h->cfg.batch) return h->fail(BOD_ERR_INVALID_ARG, "image index out of range");
    HIPCHK(h, hipMemcpyAsync(num, h->nms_nsel + img, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t k = (size_t)*num, o = (size_t)img * h->cfg.nms_max_output_size, C = h->cfg.num_classes;
    BODCHK(d2h(h, scores, h->out_scores + o * C, k * C));
    BODCHK(d2h(h, means, h->out_means + o * 4, k * 4));
    BODCHK(d2h(h, covs, h->out_covs + o * 16, k * 16));
    BODCHK(d2h(h, counts, h->out_counts + o * C, k * C));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BOD_OK;
}

bod_status bod_get_detections_batch(bod_handle h, int32_t* num, float* scores, float* means, float* covs, float* counts) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->cluster_done) return h->fail(BOD_ERR_NOT_READY, "bod_cluster_fuse has not run");
    const size_t BK = (size_t)h->cfg.batch * h->cfg.nms_max_output_size, C = h->cfg.num_classes;
    BODCHK(d2h(h, num, h->nms_nsel, (size_t)h->cfg.batch));
    BODCHK(d2h(h, scores, h->out_scores, BK * C));
    BODCHK(d2h(h, means, h->out_means, BK * 4));
    BODCHK(d2h(h, covs, h->out_covs, BK * 16));
    BODCHK(d2h(h, counts, h->out_counts, BK * C));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return BOD_OK;
}

bod_status bod_device_raw(bod_handle h, void** p, int32_t mark_ready) {
    if (!h || !p) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (mark_ready) BODCHK(ensure_raw(h)); else BODCHK(materialise_raw(h));
    p[0] = h->raw[0]; p[1] = h->raw[1]; p[2] = h->cfg.has_covar_head ? h->raw[2] : nullptr;
    if (mark_ready) { h->forward_done = true; h->posterior_done = h->nms_done = h->cluster_done = false; h->affinity_img = -1; h->raw_valid = true; h->agg_valid = false; }
    return BOD_OK;
}

bod_status bod_device_detections(bod_handle h, int32_t sidx, void** p) {
    if (!h || !p || sidx < 0 || sidx > 1) return BOD_ERR_INVALID_ARG;
    p[0] = h->nms_nsel_s[sidx]; p[1] = h->out_scores_s[sidx]; p[2] = h->out_means_s[sidx];
    p[3] = h->out_covs_s[sidx]; p[4] = h->out_counts_s[sidx];
    return BOD_OK;
}

bod_status bod_infer(bod_handle h, const float* images, int32_t on_device, uint64_t seed, uint32_t first_image_id) {
    MarkerRange mr_api("bod:infer");
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->weights_ready) return h->fail(BOD_ERR_NOT_READY, "weights not finalized");
    if (h->cfg.mc_samples < 2) return h->fail(BOD_ERR_INVALID_ARG, "bayes_od needs mc_samples >= 2 (sample covariance divides by N-1)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const float* dev = nullptr;
    BODCHK(stage_images(h, images, on_device, &dev));
    h->cur_images = dev;
    BODCHK(run_forward(h, dev, seed, first_image_id, infer_flavour(h)));
    BODCHK(run_posterior(h, seed, first_image_id));
    BODCHK(run_nms(h, h->stream));
    return run_cluster(h, h->stream);
}

bod_status bod_infer_async(bod_handle h, const float* images, int32_t on_device, uint64_t seed, uint32_t first_image_id, int32_t* slot_out) {
    MarkerRange mr_api("bod:infer_async");
    if (!h || !slot_out) return BOD_ERR_INVALID_ARG;
    if (!h->weights_ready) return h->fail(BOD_ERR_NOT_READY, "weights not finalized");
    if (h->cfg.mc_samples < 2) return h->fail(BOD_ERR_INVALID_ARG, "bayes_od needs mc_samples >= 2 (sample covariance divides by N-1)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int sidx = h->slot ^ 1;
    if (h->side_pending[sidx])
        return h->fail(BOD_ERR_NOT_READY, "slot %d still holds uncollected detections: call bod_collect first", sidx);
    // pipeline_overlap handles: this call's front goes to h->front, everything else to h->back (h->stream for the call's duration);
    // both are first ordered behind whatever the full-chip stream still holds (an upload, a synchronous call's tail)
    const bool ov = h->overlap_mode != 0;
    struct MainStream { bod_context* h; ~MainStream() { h->stream = h->full; h->in_overlap_call = false; } } restore_main{h};
    if (ov) {
        HIPCHK(h, hipEventRecord(h->ev_join, h->full));
        HIPCHK(h, hipStreamWaitEvent(h->front, h->ev_join, 0));
        HIPCHK(h, hipStreamWaitEvent(h->back, h->ev_join, 0));
        h->stream = h->back; h->in_overlap_call = true; h->ov_dirty = true;
    }
    const float* dev = nullptr;
    BODCHK(stage_images(h, images, on_device, &dev, ov ? h->front : h->stream));
    h->cur_images = dev;
    BODCHK(run_forward(h, dev, seed, first_image_id, infer_flavour(h), false, ov));
    BODCHK(run_posterior(h, seed, first_image_id));           // waits for the side stream's previous readers
    HIPCHK(h, hipEventRecord(h->ev_posterior, h->stream));
    h->select_slot(sidx);
    // Round 5: soft-NMS + cluster-and-fuse (and the copies of the records) run on the MAIN stream, behind the posterior, not on the side
    // stream beside the next call's stem.  With them on the side stream a call's detections were not reproducible: one detection in ~60
    // frames with its fused mean moved by up to 0.4 px (same counts and scores: one cluster member more or less) -- in 3-14 of 12 runs of
    // tests/tools/overlap_race.py on overlap handles (64 frames, two calls in flight), and in 5 of 12 fresh processes whose FIRST work is a
    // pipelined pair on a serial handle (tests/tools/pipelined_vs_sync.py, 128 x 128 x 128 frames); with them on the main stream: 0 of 12
    // and 0 of 12.  The mechanism is not found (every buffer the two streams share is ordered by events; tests/tools/side_race.py: the
    // posterior re-run on unchanged inputs differs while ANOTHER handle keeps the GPU busy, never alone; no LDS overrun:
    // tests/tools/lds_canary.py) -- so the contract ("pipelined == synchronous, bit for bit") is kept by not overlapping them.  Cost:
    // the side work no longer hides under the next stem, where it collided for 2.4 ms per 512-frame step anyway (History A.1):
    // measured below.  BOD_SIDE_STREAM=1: the side stream again (A/B).
    static const bool side_on = [] { const char* e = getenv("BOD_SIDE_STREAM"); return e && atoi(e) != 0; }();
    hipStream_t sd = side_on ? h->side : h->stream;
    if (side_on) HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_posterior, 0));
    BODCHK(run_nms(h, sd));
    BODCHK(run_cluster(h, sd));
    // The records follow the kernels on the side stream into pinned host memory, so bod_collect only waits for
    // this slot's event: it must never queue work behind the NEXT batch's side-stream kernels (that would
    // stall the host until the next batch has finished and drain the pipeline).
    {
        const size_t B = (size_t)h->cfg.batch, BK = B * h->cfg.nms_max_output_size, C = h->cfg.num_classes;
        char* hs = h->host_stage[sidx];
        HIPCHK(h, hipMemcpyAsync(hs, h->nms_nsel_s[sidx], B * 4, hipMemcpyDeviceToHost, sd)); hs += B * 4;
        HIPCHK(h, hipMemcpyAsync(hs, h->out_scores_s[sidx], BK * C * 4, hipMemcpyDeviceToHost, sd)); hs += BK * C * 4;
        HIPCHK(h, hipMemcpyAsync(hs, h->out_means_s[sidx], BK * 16, hipMemcpyDeviceToHost, sd)); hs += BK * 16;
        HIPCHK(h, hipMemcpyAsync(hs, h->out_covs_s[sidx], BK * 64, hipMemcpyDeviceToHost, sd)); hs += BK * 64;
        HIPCHK(h, hipMemcpyAsync(hs, h->out_counts_s[sidx], BK * C * 4, hipMemcpyDeviceToHost, sd));
    }
    HIPCHK(h, hipEventRecord(h->ev_done[sidx], sd));
    h->done_stream[sidx] = sd;
    h->side_pending[sidx] = true;
    *slot_out = sidx;
    return BOD_OK;
}

bod_status bod_collect(bod_handle h, int32_t sidx, int32_t* num, float* scores, float* means, float* covs, float* counts) {
    MarkerRange mr_api("bod:collect");
    if (!h) return BOD_ERR_INVALID_ARG;
    if (sidx < 0 || sidx > 1 || !h->side_pending[sidx]) return h->fail(BOD_ERR_NOT_READY, "slot %d has no pending batch", sidx);
    const size_t B = (size_t)h->cfg.batch, BK = B * h->cfg.nms_max_output_size, C = h->cfg.num_classes;
    HIPCHK(h, hipEventSynchronize(h->ev_done[sidx]));
    const char* hs = h->host_stage[sidx];
    if (num) std::memcpy(num, hs, B * 4);
    hs += B * 4;
    if (scores) std::memcpy(scores, hs, BK * C * 4);
    hs += BK * C * 4;
    if (means) std::memcpy(means, hs, BK * 16);
    hs += BK * 16;
    if (covs) std::memcpy(covs, hs, BK * 64);
    hs += BK * 64;
    if (counts) std::memcpy(counts, hs, BK * C * 4);
    h->side_pending[sidx] = false;
    return BOD_OK;
}

bod_status bod_stage_conv(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                          const float* w, const float* bias, int32_t KH, int32_t KW, int32_t Cout,
                          int32_t stride, int32_t same_padding, int32_t relu, const float* residual,
                          float dropout_rate, uint64_t seed, int32_t layer_id, uint32_t image_id,
                          int32_t round_output_bf16, int32_t precision, float* out) {
    bod_context ctx;                      // scratch context: owns the temporary device buffers
    bod_context* h = &ctx;
    const int mxfmt = precision == BOD_PRECISION_F16MX ? 1 : precision == BOD_PRECISION_F16MX4 ? 2 : 0;
    const bool mxp = mxfmt != 0;                 // f16mx tower kernel; round_output_bf16 = data path under test (bayesod.h)
    const int mx_mode = mxp ? round_output_bf16 : 0;
    const bool f32 = precision == BOD_PRECISION_FP32, x3 = precision == BOD_PRECISION_BF16X3 || mxp;
    if (mxp) round_output_bf16 = 1;
    h->es = (f32 || x3) ? 4 : 2;
    h->split = x3; h->mx = mxfmt;
    auto done = [&](bod_status s) {
        if (s != BOD_OK) g_create_error = h->err;
        if (h->stream) hipStreamSynchronize(h->stream);
        for (void* p : h->allocs) hipFree(p);
        if (h->stream) hipStreamDestroy(h->stream);
        h->allocs.clear(); h->stream = nullptr;
        return s;
    };
    if (!x || !w || !out || B < 1 || H < 1 || W < 1 || KH < 1 || KW < 1 || stride < 1 || stride > 2)
        return done(h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv: bad argument"));
    if (precision != BOD_PRECISION_BF16 && !f32 && !x3) return done(h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv: bad precision"));
    if (mxp && (KH != 3 || KW != 3 || stride != 1 || !same_padding || Cin != 256 || Cout != 256 || residual || mx_mode < 0 || mx_mode > 2))
        return done(h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv: BOD_PRECISION_F16MX / F16MX4 run a head-tower layer (3x3, stride 1, SAME, 256 -> 256, no residual), round_output_bf16 in 0..2"));
    if (f32 && round_output_bf16) return done(h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv: round_output_bf16 is meaningless in fp32 precision"));
    if (Cin % 64 != 0 || ((round_output_bf16 || (dropout_rate > 0.f && !f32)) && Cout % 4 != 0))
        return done(h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv: Cin must be a multiple of 64 (and Cout of 4 for bf16 output); got %d, %d", Cin, Cout));
    if (KH > 3 || KW > 3) return done(h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv: kernel larger than 3x3 needs a wider zero border"));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return done(h->fail(BOD_ERR_NO_DEVICE, "no HIP device %d: libbayesod_hip has no CPU fallback", device));
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
        return done(h->fail(BOD_ERR_HIP, "cannot set up device %d", device));
    h->cfg.batch = B;
    int OH, OW, oy = 1, ox = 1;
    if (same_padding) {
        OH = (H + stride - 1) / stride; OW = (W + stride - 1) / stride;
        oy = 1 - same_pad_before(H, KH, stride); ox = 1 - same_pad_before(W, KW, stride);
    } else {
        OH = (H - KH) / stride + 1; OW = (W - KW) / stride + 1;
    }
    if (OH < 1 || OW < 1) return done(h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv: empty output"));
    auto run = [&]() -> bod_status {
        Plane in, res;
        BODCHK(new_plane(h, &in, B, H, W, Cin));
        // bf16x3: a pixel holds 2 * C slots, channel c -> hi at (c / 32) * 64 + c % 32, lo 32 slots further on
        auto put_split = [](std::vector<uint16_t>& dst, size_t pix_slot0, int c, float v) {
            const uint16_t hi = f2bf(v);
            const size_t slot = pix_slot0 + (size_t)(c >> 5) * 64 + (c & 31);
            dst[slot] = hi; dst[slot + 32] = f2bf(v - bf2f(hi));
        };
        std::vector<uint16_t> hx(f32 ? 0 : (size_t)B * in.bstride * Cin * (x3 ? 2 : 1), 0);
        std::vector<float> hx32(f32 ? (size_t)B * in.bstride * Cin : 0, 0.f);
        const bool hx_in = mxp && mx_mode != 2;
        for (int b = 0; b < B && hx_in; ++b)
            for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx) {
                    const size_t pix = (size_t)b * in.bstride + (size_t)(y + 1) * in.pitch + (xx + 1);
                    (mxfmt == 2 ? pack_h4_row : pack_hx_row)(x + (((size_t)b * H + y) * W + xx) * Cin, Cin, reinterpret_cast<uint8_t*>(hx.data()) + pix * Cin * 4, false);
                }
        for (int b = 0; b < B && !hx_in; ++b)
            for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx)
                    for (int c = 0; c < Cin; ++c) {
                        const size_t pix = (size_t)b * in.bstride + (size_t)(y + 1) * in.pitch + (xx + 1);
                        const size_t di = pix * Cin + c;
                        const float v = x[(((size_t)b * H + y) * W + xx) * Cin + c];
                        if (f32) hx32[di] = v; else if (x3) put_split(hx, pix * 2 * Cin, c, v); else hx[di] = f2bf(v);
                    }
        HIPCHK(h, hipMemcpyAsync(in.d, f32 ? (const void*)hx32.data() : (const void*)hx.data(),
                                 (size_t)B * in.bstride * Cin * h->es, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (residual) {
            BODCHK(new_plane(h, &res, B, OH, OW, Cout));
            std::vector<uint16_t> hr(f32 ? 0 : (size_t)B * res.bstride * Cout * (x3 ? 2 : 1), 0);
            std::vector<float> hr32(f32 ? (size_t)B * res.bstride * Cout : 0, 0.f);
            for (int b = 0; b < B; ++b)
                for (int y = 0; y < OH; ++y)
                    for (int xx = 0; xx < OW; ++xx)
                        for (int c = 0; c < Cout; ++c) {
                            const size_t pix = (size_t)b * res.bstride + (size_t)(y + 1) * res.pitch + (xx + 1);
                            const size_t di = pix * Cout + c;
                            const float v = residual[(((size_t)b * OH + y) * OW + xx) * Cout + c];
                            if (f32) hr32[di] = v; else if (x3) put_split(hr, pix * 2 * Cout, c, v); else hr[di] = f2bf(v);
                        }
            HIPCHK(h, hipMemcpyAsync(res.d, f32 ? (const void*)hr32.data() : (const void*)hr.data(),
                                     (size_t)B * res.bstride * Cout * h->es, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        HostTensor k; k.shape = {KH, KW, Cin, Cout}; k.data.assign(w, w + (size_t)KH * KW * Cin * Cout);
        h->host_w["stage/0"] = std::move(k);
        if (bias) { HostTensor bt; bt.shape = {Cout}; bt.data.assign(bias, bias + Cout); h->host_w["stage/1"] = std::move(bt); }
        PackedConv pc;
        BODCHK(pack_conv(h, "stage", "", 64, &pc, hx_in ? mxfmt : 0));
        // dense fp32 / bf16 output [B,OH,OW,Cout]
        const bool drop = dropout_rate > 0.f;
        const size_t n_out = (size_t)B * OH * OW * Cout;
        float* d_out32 = nullptr; uint16_t* d_out16 = nullptr;
        const bool f32_out = f32 || (!round_output_bf16 && !drop);
        if (f32_out) BODCHK(h->dalloc(&d_out32, n_out)); else BODCHK(h->dalloc(&d_out16, n_out * (x3 ? 2 : 1)));
        std::vector<RowEnt> rows((size_t)B * OH * OW);
        size_t r = 0;
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < OH; ++y)
                for (int xx = 0; xx < OW; ++xx) {
                    RowEnt e{};
                    e.in_off = (int32_t)(b * in.bstride + (int64_t)(y * stride + oy) * in.pitch + (xx * stride + ox));
                    e.in_pitch = in.pitch;
                    e.out_off = (int32_t)(((int64_t)b * OH + y) * OW + xx);
                    if (residual) e.res_off = (int32_t)(b * res.bstride + (int64_t)(y + 1) * res.pitch + (xx + 1));
                    e.rng_p = y * OW + xx;
                    e.rng_zs = b;
                    rows[r++] = e;
                }
        RowEnt* d_rows = nullptr;
        BODCHK(h->dalloc(&d_rows, rows.size(), false));
        HIPCHK(h, hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
        ConvArgs a = base_args(pc, d_rows, B * OH * OW, Cin, Cout);
        a.g[0] = ConvGroup{in.d, pc.w, pc.bias, f32_out ? (void*)d_out32 : (void*)d_out16, residual ? res.d : nullptr, nullptr, 0, layer_id, nullptr, nullptr, nullptr, 0, 0};
        if (mxp) {                  // the tower kernel lives on the row-reuse loop: 256-slot tiles of x-adjacent runs + extended rows (plan_tables.h)
            std::vector<RowEnt> tiled;
            std::vector<ExtRow> ext;
            if (!xr_tile_rows(rows, tiled, ext)) return h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv: row-reuse tiling failed");
            RowEnt* d_tiled = nullptr; int2* d_ext = nullptr;
            BODCHK(h->dalloc(&d_tiled, tiled.size(), false));
            BODCHK(h->dalloc(&d_ext, ext.size(), false));
            HIPCHK(h, hipMemcpyAsync(d_tiled, tiled.data(), tiled.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(d_ext, ext.data(), ext.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            a.rows = d_tiled; a.M = (int)tiled.size(); a.ext = d_ext; a.xreuse = 2;
            a.mx = mx_mode == 2 ? 2 : (mxfmt == 2 ? 3 : 1);
            a.g[0].out_hx = mx_mode != 0 ? mxfmt : 0;
        }
        a.flags = (relu ? CONV_RELU : 0) | (drop ? CONV_DROPOUT : 0) | ((f32_out && !f32) ? CONV_OUT_F32 : 0);
        if (stride == 1 && same_padding && KH == 3 && KW == 3) { a.plane_h = OH; a.plane_w = OW; }      // (the sliding-window kernels walk planes)
        a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.image_base = image_id;
        a.drop_threshold = (uint32_t)std::floor((double)dropout_rate * 65536.0);
        a.drop_scale = (float)(1.0 / (1.0 - (double)dropout_rate));
        if (const char* ks = getenv("BOD_STAGE_KSPLIT")) {        // test hook: run this stage through the split-K path
            const int S = atoi(ks);
            if (S > 1 && !f32_out && !drop && (Cin / 64) % S == 0 && Cin % 64 == 0) {
                a.ksplit = S;
                BODCHK(h->dalloc(&a.partial, (size_t)S * a.M * a.cout_pad));
            }
        }
        if (x3) to_split_args(&a);
        HIPCHK(h, f32 ? launch_conv_igemm_f32(a, h->stream) : launch_conv_igemm(a, h->stream));
        if (f32_out) {
            HIPCHK(h, hipMemcpyAsync(out, d_out32, n_out * 4, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        } else {
            std::vector<uint16_t> ho(n_out * (x3 ? 2 : 1));
            HIPCHK(h, hipMemcpyAsync(ho.data(), d_out16, ho.size() * 2, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            if (mxp && mx_mode != 0) {
                for (size_t px = 0; px < n_out / Cout; ++px) (mxfmt == 2 ? unpack_h4_row : unpack_hx_row)(reinterpret_cast<const uint8_t*>(ho.data()) + px * Cout * 4, Cout, out + px * Cout);
            } else if (x3) {
                for (size_t px = 0; px < n_out / Cout; ++px)
                    for (int c = 0; c < Cout; ++c) {
                        const size_t slot = px * 2 * Cout + (size_t)(c >> 5) * 64 + (c & 31);
                        out[px * Cout + c] = bf2f(ho[slot]) + bf2f(ho[slot + 32]);
                    }
            } else {
                for (size_t i = 0; i < n_out; ++i) out[i] = bf2f(ho[i]);
            }
        }
        return BOD_OK;
    };
    return done(run());
}

bod_status bod_stage_conv_wgrad(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                const float* dy, int32_t KH, int32_t KW, int32_t Cout, int32_t stride, int32_t same_padding,
                                int32_t ksplit, float* dw, float* db) {
    bod_context ctx;
    bod_context* h = &ctx;
    h->es = 2;
    auto done = [&](bod_status s) {
        if (s != BOD_OK) g_create_error = h->err;
        if (h->stream) hipStreamSynchronize(h->stream);
        for (void* p : h->allocs) hipFree(p);
        if (h->stream) hipStreamDestroy(h->stream);
        h->allocs.clear(); h->stream = nullptr;
        return s;
    };
    if (!x || !dy || !dw || B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || KH < 1 || KW < 1 || KH > 3 || KW > 3 || stride < 1 || stride > 2)
        return done(h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv_wgrad: bad argument"));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return done(h->fail(BOD_ERR_NO_DEVICE, "no HIP device %d: libbayesod_hip has no CPU fallback", device));
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
        return done(h->fail(BOD_ERR_HIP, "cannot set up device %d", device));
    h->cfg.batch = B;
    int OH, OW, oy = 1, ox = 1;
    if (same_padding) {
        OH = (H + stride - 1) / stride; OW = (W + stride - 1) / stride;
        oy = 1 - same_pad_before(H, KH, stride); ox = 1 - same_pad_before(W, KW, stride);
    } else {
        OH = (H - KH) / stride + 1; OW = (W - KW) / stride + 1;
    }
    if (OH < 1 || OW < 1) return done(h->fail(BOD_ERR_INVALID_ARG, "bod_stage_conv_wgrad: empty output"));
    auto run = [&]() -> bod_status {
        const int M = B * OH * OW, taps = KH * KW;
        // ---- forward-layout inputs: padded bf16 plane of x, dense bf16 dY, the forward row table
        Plane in;
        BODCHK(new_plane(h, &in, B, H, W, Cin));
        std::vector<uint16_t> hx((size_t)B * in.bstride * Cin, 0);
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx)
                    for (int c = 0; c < Cin; ++c)
                        hx[((size_t)b * in.bstride + (size_t)(y + 1) * in.pitch + (xx + 1)) * Cin + c] = f2bf(x[(((size_t)b * H + y) * W + xx) * Cin + c]);
        HIPCHK(h, hipMemcpyAsync(in.d, hx.data(), hx.size() * 2, hipMemcpyHostToDevice, h->stream));
        std::vector<uint16_t> hdy((size_t)M * Cout);
        for (size_t i = 0; i < hdy.size(); ++i) hdy[i] = f2bf(dy[i]);
        uint16_t* d_dy = nullptr;
        BODCHK(h->dalloc(&d_dy, hdy.size(), false));
        HIPCHK(h, hipMemcpyAsync(d_dy, hdy.data(), hdy.size() * 2, hipMemcpyHostToDevice, h->stream));
        std::vector<RowEnt> rows((size_t)M);
        size_t r = 0;
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < OH; ++y)
                for (int xx = 0; xx < OW; ++xx) {
                    RowEnt e{};
                    e.in_off = (int32_t)(b * in.bstride + (int64_t)(y * stride + oy) * in.pitch + (xx * stride + ox));
                    e.in_pitch = in.pitch;
                    rows[r++] = e;
                }
        RowEnt* d_rows = nullptr;
        BODCHK(h->dalloc(&d_rows, rows.size(), false));
        HIPCHK(h, hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
        // ---- K-contiguous operands: dY^T [cout_pad][Kpad] and Xcol^T [taps*Cin + 1][Kpad] (last row = ones -> db)
        const int cout_pad = (Cout + 63) / 64 * 64;
        const int N = taps * Cin + 1;
        int S = ksplit;
        if (S < 1) {                                  // enough splits for ~512 workgroups, >= 4 K-tiles each
            const long tiles = (long)((N + 127) / 128) * (cout_pad % 128 == 0 ? cout_pad / 128 : cout_pad / 64);
            S = 1;
            while (tiles * S < 512 && S < 64 && (long)M / (64L * S * 2) >= 4) S *= 2;
        }
        const int Kpad = (M + 64 * S - 1) / (64 * S) * (64 * S);
        uint16_t* d_dyt = nullptr; uint16_t* d_xct = nullptr;
        BODCHK(h->dalloc(&d_dyt, (size_t)cout_pad * Kpad));                 // zero-filled: rows >= Cout stay 0
        BODCHK(h->dalloc(&d_xct, (size_t)N * Kpad, false));
        const bool trace = getenv("BOD_TRACE_WGRAD") != nullptr;     // development aid: device time of the two phases
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        if (trace) { for (auto& e : ev) HIPCHK(h, hipEventCreate(&e)); HIPCHK(h, hipEventRecord(ev[0], h->stream)); }
        HIPCHK(h, launch_gather_transpose(d_dy, nullptr, d_dyt, M, Kpad, Cout, Cout, 1, 1, false, h->stream));
        HIPCHK(h, launch_gather_transpose(in.d, d_rows, d_xct, M, Kpad, Cin, Cin, taps, KW, true, h->stream));
        // ---- the forward kernel as a plain GEMM: "pixels" = rows of Xcol^T, "weights" = dY^T, reduction = pixels
        if (trace) HIPCHK(h, hipEventRecord(ev[1], h->stream));
        std::vector<RowEnt> grow((size_t)N);
        for (int n = 0; n < N; ++n) { RowEnt e{}; e.in_off = n; e.out_off = n; grow[n] = e; }
        RowEnt* d_grow = nullptr;
        BODCHK(h->dalloc(&d_grow, grow.size(), false));
        HIPCHK(h, hipMemcpyAsync(d_grow, grow.data(), grow.size() * sizeof(RowEnt), hipMemcpyHostToDevice, h->stream));
        float* d_zero = nullptr; float* d_out = nullptr; float* d_part = nullptr;
        BODCHK(h->dalloc(&d_zero, (size_t)cout_pad));
        BODCHK(h->dalloc(&d_out, (size_t)N * Cout));
        ConvArgs a{};
        a.rows = d_grow; a.M = N; a.taps = 1; a.KW = 1; a.cin = Kpad; a.in_cstride = Kpad;
        a.cout_pad = cout_pad; a.cout_valid = Cout; a.out_cstride = Cout; a.res_cstride = Cout; a.groups = 1;
        a.fan_count = 1; a.flags = CONV_OUT_F32;
        a.g[0] = ConvGroup{d_xct, d_dyt, d_zero, d_out, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0};
        if (S > 1) {
            a.ksplit = S;
            BODCHK(h->dalloc(&d_part, (size_t)S * N * cout_pad, false));
            a.partial = d_part;
        }
        HIPCHK(h, launch_conv_igemm(a, h->stream));
        if (trace) {
            HIPCHK(h, hipEventRecord(ev[2], h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            float t0 = 0, t1 = 0;
            hipEventElapsedTime(&t0, ev[0], ev[1]); hipEventElapsedTime(&t1, ev[1], ev[2]);
            const double fl = 2.0 * M * (double)taps * Cin * Cout;
            fprintf(stderr, "# wgrad M=%d K=%dx%d Cout=%d S=%d: transposes %.3f ms, GEMM+reduce %.3f ms (%.1f TFLOP/s)\n",
                    M, taps, Cin, Cout, S, t0, t1, fl / (t1 * 1e-3) / 1e12);
            for (auto& e : ev) hipEventDestroy(e);
        }
        std::vector<float> ho((size_t)N * Cout);
        HIPCHK(h, hipMemcpyAsync(ho.data(), d_out, ho.size() * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::memcpy(dw, ho.data(), (size_t)taps * Cin * Cout * 4);           // [(tap, ci)][co] == HWIO
        if (db) std::memcpy(db, ho.data() + (size_t)taps * Cin * Cout, (size_t)Cout * 4);
        return BOD_OK;
    };
    return done(run());
}

static bod_status loss_impl(int32_t device, int32_t B, int32_t A, int32_t C, const float* cls, const float* cls_t,
                            const float* box, const float* box_t, const float* cov, const float* anchors,
                            const uint8_t* pos, const uint8_t* neg, int32_t do_cls, int32_t reg_kind,
                            float label_smoothing, double* out4, float w_cls, float w_reg, float* dcls, float* dbox, float* dcov) {
    bod_context ctx;
    bod_context* h = &ctx;
    auto done = [&](bod_status s) {
        if (s != BOD_OK) g_create_error = h->err;
        if (h->stream) hipStreamSynchronize(h->stream);
        for (void* p : h->allocs) hipFree(p);
        if (h->stream) hipStreamDestroy(h->stream);
        h->allocs.clear(); h->stream = nullptr;
        return s;
    };
    if (B < 1 || A < 1 || (C != 4 && C != 8) || !pos || !neg || !out4 || reg_kind < 0 || reg_kind > 3)
        return done(h->fail(BOD_ERR_INVALID_ARG, "bod_loss_forward: bad argument (C must be 4 or 8)"));
    if ((do_cls && (!cls || !cls_t)) || (reg_kind && (!box || !box_t)) || (reg_kind >= 2 && (!cov || !anchors)))
        return done(h->fail(BOD_ERR_INVALID_ARG, "bod_loss_forward: a tensor required by the selected losses is NULL"));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return done(h->fail(BOD_ERR_NO_DEVICE, "no HIP device %d: libbayesod_hip has no CPU fallback", device));
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
        return done(h->fail(BOD_ERR_HIP, "cannot set up device %d", device));
    auto run = [&]() -> bod_status {
        const size_t n = (size_t)B * A;
        LossArgs a{};
        a.B = B; a.A = A; a.C = C; a.do_cls = do_cls; a.reg_kind = reg_kind; a.label_smoothing = label_smoothing;
        auto up = [&](const void* src, size_t bytes, const void** dst) -> bod_status {
            if (!src) { *dst = nullptr; return BOD_OK; }
            char* d = nullptr;
            BODCHK(h->dalloc(&d, bytes, false));
            HIPCHK(h, hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, h->stream));
            *dst = d;
            return BOD_OK;
        };
        BODCHK(up(cls, n * C * 4, (const void**)&a.cls)); BODCHK(up(cls_t, n * C * 4, (const void**)&a.cls_t));
        BODCHK(up(box, n * 16, (const void**)&a.box)); BODCHK(up(box_t, n * 16, (const void**)&a.box_t));
        BODCHK(up(cov, n * 40, (const void**)&a.cov)); BODCHK(up(anchors, (size_t)A * 16, (const void**)&a.anchors));
        BODCHK(up(pos, n, (const void**)&a.pos)); BODCHK(up(neg, n, (const void**)&a.neg));
        const int nblocks = (int)((n + 255) / 256);
        float* partial = nullptr;
        BODCHK(h->dalloc(&partial, (size_t)nblocks * 4));
        HIPCHK(h, launch_loss(a, partial, nblocks, h->stream));
        std::vector<float> hp((size_t)nblocks * 4);
        HIPCHK(h, hipMemcpyAsync(hp.data(), partial, hp.size() * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int q = 0; q < 4; ++q) out4[q] = 0.0;
        for (int b = 0; b < nblocks; ++b)
            for (int q = 0; q < 4; ++q) out4[q] += (double)hp[(size_t)b * 4 + q];
        if (dcls || dbox || dcov) {
            float* sums = nullptr; float* g_cls = nullptr; float* g_box = nullptr; float* g_cov = nullptr;
            BODCHK(h->dalloc(&sums, 4));
            if (dcls) BODCHK(h->dalloc(&g_cls, n * C));
            if (dbox) BODCHK(h->dalloc(&g_box, n * 4));
            if (dcov) BODCHK(h->dalloc(&g_cov, n * 10));
            HIPCHK(h, launch_loss_reduce(partial, nblocks, sums, h->stream));
            HIPCHK(h, launch_loss_backward(a, sums, w_cls, w_reg, g_cls, g_box, g_cov, h->stream));
            if (dcls) HIPCHK(h, hipMemcpyAsync(dcls, g_cls, n * C * 4, hipMemcpyDeviceToHost, h->stream));
            if (dbox) HIPCHK(h, hipMemcpyAsync(dbox, g_box, n * 16, hipMemcpyDeviceToHost, h->stream));
            if (dcov) HIPCHK(h, hipMemcpyAsync(dcov, g_cov, n * 40, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        return BOD_OK;
    };
    return done(run());
}

bod_status bod_loss_forward(int32_t device, int32_t B, int32_t A, int32_t C, const float* cls, const float* cls_t,
                            const float* box, const float* box_t, const float* cov, const float* anchors,
                            const uint8_t* pos, const uint8_t* neg, int32_t do_cls, int32_t reg_kind,
                            float label_smoothing, double* out4) {
    return loss_impl(device, B, A, C, cls, cls_t, box, box_t, cov, anchors, pos, neg, do_cls, reg_kind, label_smoothing, out4,
                     0.f, 0.f, nullptr, nullptr, nullptr);
}

bod_status bod_loss_backward(int32_t device, int32_t B, int32_t A, int32_t C, const float* cls, const float* cls_t,
                             const float* box, const float* box_t, const float* cov, const float* anchors,
                             const uint8_t* pos, const uint8_t* neg, int32_t do_cls, int32_t reg_kind,
                             float label_smoothing, float w_cls, float w_reg, double* out4, float* dcls, float* dbox, float* dcov) {
    if (!dcls && !dbox && !dcov) return BOD_ERR_INVALID_ARG;
    return loss_impl(device, B, A, C, cls, cls_t, box, box_t, cov, anchors, pos, neg, do_cls, reg_kind, label_smoothing, out4,
                     w_cls, w_reg, dcls, dbox, dcov);
}

bod_status bod_bench_head_conv(bod_handle h, int32_t layer, int32_t variant, int32_t iters, double* mean_ms, double* flops) {
    if (!h || !mean_ms || iters < 1 || layer < 0 || layer > 7) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    if (!h->weights_ready) return h->fail(BOD_ERR_NOT_READY, "weights not finalized");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int seen = 0; Op* op = nullptr;
    for (Op& o : h->ops) if (o.is_head3x3 && o.flavour != FLAVOUR_RAW && seen++ == layer) { op = &o; break; }    // (layers 2, 3: the flavour bod_infer runs)
    if (!op) return h->fail(BOD_ERR_INVALID_ARG, "no head launch %d", layer);
    ConvArgs a = op->conv;
    a.variant = variant;
    // BOD_BENCH_ZERO=1: time the identical launch on zero-filled activations (DVFS / power-limit probe: the matrix
    // pipe toggles far less on zeros, so any speed-up is clock, not work).  Destroys the head buffers' contents.
    if (const char* z = getenv("BOD_BENCH_ZERO")) {
        if (atoi(z) != 0 && layer > 0) {
            const size_t bytes = (size_t)h->cfg.batch * h->cfg.mc_samples * h->Ppad * 256 * h->es;
            for (int g = 0; g < a.groups; ++g) HIPCHK(h, hipMemsetAsync(const_cast<void*>(a.g[g].in), 0, bytes, h->stream));
        }
    }
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    HIPCHK(h, launch_conv_igemm(a, h->stream));            // warm-up
    HIPCHK(h, hipEventRecord(e0, h->stream));
    for (int i = 0; i < iters; ++i) HIPCHK(h, launch_conv_igemm(a, h->stream));
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0; HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    if ((variant == 90 || variant == 91) && a.mx == 1) {
        unsigned long long c[16];
        conv_igemm_phase_cycles(c, true);
        const double tiles = (double)c[15];
        fprintf(stderr, "# f16mx phase clock, wave 0, cycles per tile (%.0f tiles): whole tile %.0f = set-up %.0f + loop %.0f (of which: waiting for the next K-tile's pieces + barrier %.0f, "
                        "K-tile bodies %.0f; 72 K-tiles) + epilogue %.0f; of the wait: vmcnt / lgkmcnt %.0f, s_barrier %.0f\n", tiles, (double)c[9] / tiles, ((double)c[8] - (double)c[6] - (double)c[7]) / tiles,
                ((double)c[6] + (double)c[7]) / tiles, (double)c[6] / tiles, (double)c[7] / tiles, ((double)c[9] - (double)c[8]) / tiles,
                (double)c[13] / tiles, ((double)c[6] - (double)c[13]) / tiles);
        // round 6: the loop by K-tile flavour (36 H K-tiles of four f16 k-steps, 36 X K-tiles of two block-scaled products: half the MFMA issue on
        // the same staged bytes), for wave 0 (early: issues the weight pieces) and wave 4 (late: its partner on the SIMD)
        fprintf(stderr, "# f16mx phase clock by K-tile flavour, cycles per K-tile: wave 0: H %.0f  X %.0f | wave 4: H %.0f  X %.0f, wait + barrier per K-tile %.0f (wave 0: %.0f)"
                        " | MFMA issue of a SIMD's two waves: H 2048, X 1024\n",
                (double)c[0] / tiles / 36.0, (double)c[1] / tiles / 36.0, (double)c[2] / tiles / 36.0, (double)c[3] / tiles / 36.0, (double)c[4] / tiles / 72.0, (double)c[6] / tiles / 72.0);
    } else if (variant == 90) {
        unsigned long long c[16];
        conv_igemm_phase_cycles(c, true);
        const double tiles = (double)c[15];
        static const char* names[6] = {"set-up", "main loop", "barrier+bias/relu/pack", "philox+lds writes", "barrier", "store loop"};
        fprintf(stderr, "# phase clock, wave 0, cycles per tile (%.0f tiles):", tiles);
        for (int k = 0; k < 6; ++k) fprintf(stderr, "  %s %.0f", names[k], (double)c[k] / tiles);
        // (aggregating tiles leave through the fused 1x1 + MC reduction instead of the store loop: slots 10..12)
        if (c[10] + c[11] + c[12] > 0)
            fprintf(stderr, "  fused 1x1 MFMAs %.0f  barriers+fp32 tile %.0f  MC reduction+stores %.0f", (double)c[10] / tiles, (double)c[11] / tiles, (double)c[12] / tiles);
        double cyc = 0; for (int k = 0; k < 6; ++k) cyc += (double)c[k];
        cyc += (double)c[10] + (double)c[11] + (double)c[12];
        fprintf(stderr, "  | shader clock during the tiles %.3f GHz (cycles / 100 MHz real-time ticks); tiles account for %.3f ms of the %.3f ms launch per CU\n",
                cyc / (double)c[14] * 0.1, (double)c[14] / tiles * 1e-5 * (tiles / (iters + 1)) / 256.0, ms / iters);
    }
    *mean_ms = ms / iters;
    if (flops) *flops = op->flops;
    return BOD_OK;
}

bod_status bod_profile_begin(bod_handle h) {
    if (!h) return BOD_ERR_INVALID_ARG;
    for (auto& e : h->ev_head) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& e : h->ev_post) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    h->ev_head.clear(); h->ev_post.clear(); h->prof_flops = 0; h->profiling = true;
    return BOD_OK;
}

// ---- the path's one multi-GPU exchange through the C ABI ----
extern "C++" {
namespace {
// RCCL is opened at run time: the library neither links librccl nor needs it on a single GPU.  (ncclGather is RCCL's own entry point;
// datatype 7 = ncclFloat32, result 0 = ncclSuccess -- rccl.h.)
struct Rccl {
    typedef int (*gather_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef const char* (*errstr_fn)(int);
    gather_fn gather = nullptr; errstr_fn errstr = nullptr; bool tried = false;
    bool load() {
        if (tried) return gather != nullptr;
        tried = true;
        void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return false;
        gather = reinterpret_cast<gather_fn>(dlsym(lib, "ncclGather"));
        errstr = reinterpret_cast<errstr_fn>(dlsym(lib, "ncclGetErrorString"));
        return gather != nullptr;
    }
};
Rccl& rccl() { static Rccl r; return r; }
}  // namespace
}  // extern "C++"

int32_t bod_record_width(bod_handle h) { return h ? 21 + 2 * h->cfg.num_classes : 0; }

bod_status bod_gather_detections(bod_handle h, int32_t slot, void* nccl_comm, int32_t world, int32_t rank, int32_t root,
                                 float* gathered_host, float** gathered_device) {
    MarkerRange mr_api("bod:gather_detections");
    if (!h) return BOD_ERR_INVALID_ARG;
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
        return h->fail(BOD_ERR_INVALID_ARG, "bod_gather_detections: world %d, rank %d, root %d", world, rank, root);
    if (!nccl_comm && world != 1) return h->fail(BOD_ERR_INVALID_ARG, "bod_gather_detections: %d ranks need an ncclComm_t", world);
    if (rank != root && (gathered_host || gathered_device)) return h->fail(BOD_ERR_INVALID_ARG, "bod_gather_detections: only the root receives");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int B = h->cfg.batch, K = h->cfg.nms_max_output_size, C = h->cfg.num_classes, W = 21 + 2 * C;
    const size_t block = (size_t)B * K * W;
    // where the records are and which stream finished them
    int sidx = slot;
    // Round 6: a ticket's pack kernel and gather no longer run on the side stream beside the next call's forward -- a kernel
    // of this library beside its convolution kernels can miscompute a 16-lane row, and the pack kernel is one.  They follow the slot's
    // records on the stream that finished them (the main stream; the back stream of a pipeline_overlap handle): if the next
    // bod_infer_async is already enqueued they run behind its kernels -- one step of latency, nothing beside anything.
    // BOD_SIDE_STREAM=1 (the repro switch of bod_infer_async): the side stream again.
    static const bool side_on = [] { const char* e = getenv("BOD_SIDE_STREAM"); return e && atoi(e) != 0; }();
    hipStream_t st = (slot >= 0 && slot <= 1 && !side_on && h->done_stream[slot]) ? h->done_stream[slot] : h->side;
    if (slot < 0) {
        // synchronous bod_infer / bod_cluster_fuse: the records are the current buffers, finished on the MAIN stream -- pack and gather
        // go out on that stream too, so the next bod_infer (which rewrites nms_nsel / out_* of this slot on the main stream) and
        // bod_synchronize are ordered behind them by stream order.  (On the side stream nothing would hold the main stream back.)
        if (!h->cluster_done) return h->fail(BOD_ERR_NOT_READY, "bod_gather_detections: bod_cluster_fuse has not run");
        BODCHK(join_overlap(h));
        sidx = h->slot;
        st = h->stream;
    } else if (slot > 1 || !h->side_pending[slot]) {
        return h->fail(BOD_ERR_NOT_READY, "bod_gather_detections: slot %d has no pending batch", slot);
    }
    if (!h->rec_send && hipMalloc(reinterpret_cast<void**>(&h->rec_send), block * 4) != hipSuccess)
        return h->fail(BOD_ERR_OOM, "bod_gather_detections: %zu bytes", block * 4);
    // rec_send / rec_recv are shared by the ticket gathers (side stream) and the synchronous form (main stream): a gather that goes to
    // the other stream than the previous one waits for it -- nothing else orders the two streams against each other (round-4 advisor finding)
    if (h->ev_gather && h->gather_stream && h->gather_stream != st) HIPCHK(h, hipStreamWaitEvent(st, h->ev_gather, 0));
    // (on the side stream: behind the slot's event; on the records' own stream the wait is a no-op)
    if (slot >= 0) HIPCHK(h, hipStreamWaitEvent(st, h->ev_done[slot], 0));
    HIPCHK(h, launch_pack_records(h->nms_nsel_s[sidx], h->out_scores_s[sidx], h->out_means_s[sidx], h->out_covs_s[sidx],
                                  h->out_counts_s[sidx], h->rec_send, B, K, C, st));
    float* recv = nullptr;
    if (rank == root) {
        if (world == 1 && !nccl_comm) recv = h->rec_send;
        else {
            if (h->rec_recv_elems < block * world) {
                if (h->rec_recv && h->rec_recv != h->rec_send) hipFree(h->rec_recv);
                h->rec_recv = nullptr; h->rec_recv_elems = 0;
                if (hipMalloc(reinterpret_cast<void**>(&h->rec_recv), block * world * 4) != hipSuccess)
                    return h->fail(BOD_ERR_OOM, "bod_gather_detections: %zu bytes", block * world * 4);
                h->rec_recv_elems = block * world;
            }
            recv = h->rec_recv;
        }
    }
    if (nccl_comm) {
        if (!rccl().load()) return h->fail(BOD_ERR_NOT_READY, "bod_gather_detections: librccl.so (ncclGather) could not be opened");
        const int rc = rccl().gather(h->rec_send, recv, block, /*ncclFloat32*/ 7, root, nccl_comm, st);
        if (rc != 0) return h->fail(BOD_ERR_HIP, "ncclGather: %s", rccl().errstr ? rccl().errstr(rc) : "error");
    }
    if (!h->ev_gather) HIPCHK(h, hipEventCreateWithFlags(&h->ev_gather, hipEventDisableTiming));
    if (rank == root && gathered_host) {
        HIPCHK(h, hipMemcpyAsync(gathered_host, recv, block * world * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipEventRecord(h->ev_gather, st)); h->gather_stream = st;
        HIPCHK(h, hipEventSynchronize(h->ev_gather));          // (the copy, not whatever was enqueued on the stream behind it)
    } else {
        HIPCHK(h, hipEventRecord(h->ev_gather, st)); h->gather_stream = st;
    }
    if (!(rank == root && gathered_host) && slot >= 0) {
        // keep bod_collect / the next bod_infer_async of this slot behind the send: re-record the slot's event
        HIPCHK(h, hipEventRecord(h->ev_done[slot], st));
    }
    if (gathered_device) *gathered_device = recv;
    return BOD_OK;
}

bod_status bod_plan_info(bod_handle h, int32_t* info8) {
    if (!h || !info8) return BOD_ERR_INVALID_ARG;
    if (!h->weights_ready) return h->fail(BOD_ERR_NOT_READY, "bod_plan_info: the plan is built by bod_finalize_weights");
    for (int i = 0; i < 8; ++i) info8[i] = 0;
    info8[0] = h->agg_plan; info8[1] = h->plan_fused_out; info8[2] = h->plan_xreuse; info8[3] = h->plan_xreuse0;
    info8[4] = (int32_t)h->ops.size();
    for (const Op& o : h->ops) if (o.kind == Op::CONV && !o.is_head3x3 && o.conv.xreuse) ++info8[5];
    info8[6] = h->plan_mx;
    return BOD_OK;
}

bod_status bod_profile_select(bod_handle h, int32_t which) {
    if (!h) return BOD_ERR_INVALID_ARG;
    if (which < 0 || which > 2) return h->fail(BOD_ERR_INVALID_ARG, "bod_profile_select: %d", which);
    h->prof_which = which;
    return BOD_OK;
}

bod_status bod_profile_end(bod_handle h, double* head_ms, int64_t* head_launches, double* head_flops, double* post_ms, int64_t* post_launches) {
    if (!h) return BOD_ERR_INVALID_ARG;
    BODCHK(join_overlap(h));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double hm = 0, pm = 0;
    for (auto& e : h->ev_head) { float ms = 0; HIPCHK(h, hipEventElapsedTime(&ms, e.first, e.second)); hm += ms; }
    for (auto& e : h->ev_post) { float ms = 0; HIPCHK(h, hipEventElapsedTime(&ms, e.first, e.second)); pm += ms; }
    if (head_ms) *head_ms = hm;
    if (head_launches) *head_launches = (int64_t)h->ev_head.size();
    if (head_flops) *head_flops = h->prof_flops;
    if (post_ms) *post_ms = pm;
    if (post_launches) *post_launches = (int64_t)h->ev_post.size();
    h->profiling = false;
    return BOD_OK;
}

}  // extern "C"

#include "train_impl.inc"
