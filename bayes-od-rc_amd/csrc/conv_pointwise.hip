// Pointwise (1x1) convolutions of the backbone as a STREAMING kernel (round 3).
//
// The generic implicit-GEMM kernel (conv_igemm.hip) runs a 1x1 layer as a chain of dependent memory phases per tile -- K-tile DMA,
// MFMAs, shortcut DMA into the tile image, add, store -- on a CU that holds nothing else (one 256x256 tile, its staging LDS
// aliased by the output tile): res3's `2c` layers move 2.4 GB in 0.68 ms = 3.5 TB/s, 21 us per tile for 3.6 us of MFMAs.  For
// reductions of at most 256 channels this kernel turns the tile loop inside out:
//
//   * persistent workgroups of 4 waves; a workgroup keeps ONE 128-channel cout tile for its whole life, so its weights live in
//     registers as MFMA A fragments (K/4 VGPRs per lane) and are never read again;
//   * it walks pixel tiles of BP pixels; the input rows and the shortcut rows of tile t+1 are in flight (LDS-DMA into the other
//     half of two double buffers) while tile t is multiplied, finished in LDS (the shortcut tile is overwritten in place) and
//     stored as whole 256-byte pieces of pixel rows; two or three workgroups share a CU;
//   * same instruction (v_mfma_f32_32x32x16_bf16), same k order, same epilogue arithmetic (fma with the bias, + shortcut, hardware
//     round-to-nearest-even pack, ReLU on the packed words) as the generic kernel: outputs are bit-identical
//     (tests/test_gpu_forward.py::test_pointwise_kernel_is_bit_identical, BOD_POINTWISE=0 plans the generic launches).
//
// LDS-DMA completion is invisible to the compiler: the loop waits with explicit s_waitcnt vmcnt(n) where n counts exactly the
// vector-memory instructions issued behind the DMA pieces it needs (the previous tile's output stores, the next tile's pieces, the
// row-table loads): vmcnt retires in issue order on gfx9 -- the same model the compiler's own wait insertion uses -- so the stores of
// tile t-1 stay in flight under tile t's MFMAs.
#include "kernels.h"
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) short pw_bf16x8;
typedef __attribute__((ext_vector_type(16))) float pw_f32x16;

#define PW_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PW_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ uint4 pw_sink[64];                     // where the stores of a tail tile's invalid pixels go (one 16-byte slot per lane)

__device__ __forceinline__ uint32_t pw_pack(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ uint32_t pw_relu(uint32_t w) {
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ float pw_bf2f(uint32_t v) { return __uint_as_float(v << 16); }

// Chunk swizzle of a staged [pixel][channels] tile for the MFMA B fragments (ds_read_b128: four 16-lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ... over 64 banks = one 256-byte bank row per LDS cycle; MI355X_MICROARCH.md, LDS).  The 16 pixels of a lane
// group read the SAME logical chunk, so their physical 16-byte slots must be 16 different ones of the bank row: rows of 256 bytes
// or more XOR the chunk's low four bits with pixel & 15 (a group's pixels are a complete residue system mod 16, whatever the tap
// shift); rows of 128 bytes lie two to a bank row, the row's parity is in the address, and the chunk's three bits take
// (pixel >> 1) & 7.  Round 3's `pixel & 7` left every slot twice in a group (2-way conflicts).  Round 4, per kernel: the 64-channel
// sliding window 660 -> 632 us and the 512-channel reductions 537 -> 506 us per 512 frames with this form; the 128-channel sliding
// window keeps `pixel & 7` (see there).
template <int CPR> __device__ __forceinline__ int pw_swz(int px) { return CPR >= 16 ? (px & 15) : ((px >> 1) & 7); }

template <int N> __device__ __forceinline__ void pw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BC = 128 (four waves) is the kernel described above.  BC = 256 with NEXT (eight waves, one workgroup per CU) is the fused form of
// ResNet stage 2: the cout tile spans ALL 256 channels of a pixel, so the NEXT block's 1x1 reduction `2a` (256 -> 64, + bias + ReLU,
// ConvGroup.ch_w3 / ch_b3 / ch_out3) is computed from the finished tile while it is still in LDS -- the 2.1 GB block output is
// written once and read once (by the next shortcut) instead of twice, and the next block's `2a` launch disappears.  The `2a` MFMAs
// read the stored bf16 values of the tile in ascending k order like the generic kernel would from memory: bit-identical.
// NEXT = 2 ("dual", round 4): the second 64-cout convolution reads the INPUT tile instead of the finished one -- a ConvBlock's first
// two launches, `branch1` (64 -> 256, the projection shortcut) and `2a` (64 -> 64 + ReLU), both 1x1 over the same pooled plane
// (feature_extractor.py:283-309), become one: the plane is read once, and the 2a launch -- a one-K-tile layer that ran 4.16 M
// pixels through the generic kernel's per-tile latency chain at 1.1 TB/s, 0.97 ms per 256 frames -- disappears.  Same MFMA, same k
// order (four k-steps of 16 ascending), same epilogue arithmetic as the generic 64x128 tile: bit-identical.
template <int K, int BP, bool RES, int WGS, int BC = 128, int NEXT = 0>
__global__ __launch_bounds__(BC * 2, WGS) void pw_conv_kernel(const ConvArgs a, int ptiles, int ctiles, int pstride) {
    constexpr int THREADS = BC * 2;               // one wave per 32 couts
    constexpr int CPX = K * 2 / 16;               // 16-byte chunks per input pixel row
    constexpr int RCH = BC / 8;                   // 16-byte chunks per row of the shortcut / output tile
    constexpr int XB = BP * K * 2;                // bytes of one input buffer
    constexpr int RB = BP * BC * 2;               // bytes of one shortcut / output buffer (BC channels per pixel)
    constexpr int NXP = BP * CPX / THREADS;       // input DMA pieces per thread
    constexpr int NRP = BP * RCH / THREADS;       // shortcut pieces (and output stores) per thread
    constexpr int FP = BP / 32, KS = K / 16;
    constexpr int C2 = 64, KS2 = NEXT == 2 ? K / 16 : BC / 16;       // NEXT: couts and k-steps of the fused 2a (dual: over the input's channels)
    constexpr int K2 = NEXT == 2 ? K : BC;        // channels the fused 2a reduces
    constexpr int NTP = NEXT ? BP * (C2 / 8) / THREADS : 0;     // NEXT: stores of the 2a tile per thread
    static_assert(NXP >= 1 && NRP >= 1 && BP % 32 == 0 && (CPX & 7) == 0 && (!NEXT || (BC == 256 && BP == 64 && NTP == 1)), "tile shape");
    extern __shared__ __attribute__((aligned(16))) char pw_smem[];
    char* const xbuf = pw_smem;                   // [2][BP][K] bf16, chunk-swizzled
    constexpr int NRB = RES ? 2 : 1;              // shortcut / output buffers: double-buffered only when a shortcut is DMA'd into them
    char* const rbuf = pw_smem + 2 * XB;          // [NRB][BP][BC] bf16, chunk-swizzled
    int4* const meta = reinterpret_cast<int4*>(pw_smem + 2 * XB + NRB * RB);    // [3][BP] {in_off, out_off, res_off, valid}
    char* const tbuf = pw_smem + 2 * XB + NRB * RB + 3 * BP * 16;               // NEXT: [BP][64] bf16 tile of the fused 2a

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const ConvGroup& G = a.g[0];
    // workgroups that share a pixel tile (one per cout tile) sit on the SAME XCD (block b runs on XCD b % 8): the input rows they all
    // read are fetched into one L2 once
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int ct = jb % ctiles, p0 = (jb / ctiles) * 8 + xcd;
    const int c0 = ct * BC;
    const int nt = p0 < ptiles ? (ptiles - p0 + pstride - 1) / pstride : 0;
    if (nt == 0) return;

    auto load_ent = [&](int t) {                  // row-table entry of this thread's pixel (tid % BP) of the workgroup's tile t
        const int tt = t < nt ? t : nt - 1;
        int m = (p0 + tt * pstride) * BP + (tid % BP);
        const int valid = m < a.M;
        m = valid ? m : a.M - 1;
        const int4 e = *reinterpret_cast<const int4*>(&a.rows[m]);
        return make_int4(e.x, e.z, e.w, valid);   // in_off, out_off, res_off
    };
    auto issue = [&](int t) {                     // LDS-DMA of tile t's input rows (and shortcut rows) into buffer t & 1
        const int4* mt = meta + (t % 3) * BP;
        char* xb = xbuf + (t & 1) * XB;
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int q = i * THREADS + tid, row = q / CPX, cp = q % CPX;
            const int c = cp ^ pw_swz<CPX>(row);
            const char* src = reinterpret_cast<const char*>(G.in) + ((size_t)mt[row].x * a.in_cstride + G.in_coff + c * 8) * 2;
            __builtin_amdgcn_global_load_lds(PW_GLOBAL_PTR(src), PW_LDS_PTR(xb + (i * THREADS + wave * 64) * 16), 16, 0, 0);
        }
        if (RES) {
            char* rb = rbuf + (t & 1) * RB;
#pragma unroll
            for (int i = 0; i < NRP; ++i) {
                const int q = i * THREADS + tid, row = q / RCH, cp = q % RCH;
                const int c = cp ^ (row & 7);
                const char* src = reinterpret_cast<const char*>(G.res) + ((size_t)mt[row].z * a.res_cstride + c0 + c * 8) * 2;
                __builtin_amdgcn_global_load_lds(PW_GLOBAL_PTR(src), PW_LDS_PTR(rb + (i * THREADS + wave * 64) * 16), 16, 0, 0);
            }
        }
    };

    // ---- prologue: row tables of tiles 0 and 1, tile 0's DMA, the weights and the bias into registers
    {
        const int4 e0 = load_ent(0), e1 = load_ent(1);
        if (tid < BP) { meta[tid] = e0; meta[BP + tid] = e1; }
    }
    __syncthreads();
    issue(0);
    asm volatile("" ::: "memory");
    pw_bf16x8 wf[KS];
    {
        const char* wrow = reinterpret_cast<const char*>(G.w) + ((size_t)(c0 + wave * 32 + frow) * K + fhalf * 8) * 2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const pw_bf16x8*>(wrow + ks * 32);
    }
    float4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4*>(G.bias + c0 + wave * 32 + g * 8 + fhalf * 4);
    const bool relu = a.flags & CONV_RELU;
    // NEXT: waves 0-3 compute the fused 2a -- cout block (wave & 1) x pixel fragment (wave >> 1) -- with its 16 k-steps of weights in registers
    pw_bf16x8 wf2[NEXT ? KS2 : 1];
    float4 bv2[4];
    if (NEXT) {
        const char* wrow = reinterpret_cast<const char*>(G.ch_w3) + ((size_t)((wave & 1) * 32 + frow) * K2 + fhalf * 8) * 2;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) wf2[ks] = *reinterpret_cast<const pw_bf16x8*>(wrow + ks * 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) bv2[g] = *reinterpret_cast<const float4*>(G.ch_b3 + (wave & 1) * 32 + g * 8 + fhalf * 4);
    }

    for (int t = 0; t < nt; ++t) {
        __syncthreads();                          // meta of tile t+1 visible; the buffers of tile t+1 (= of tile t-1) are free
        const bool more = t + 1 < nt;
        if (more) issue(t + 1);
        asm volatile("" ::: "memory");
        const int4 ent = load_ent(t + 2);
        asm volatile("" ::: "memory");
        // tile t's pieces have landed once at most the instructions issued BEHIND them are outstanding (vmcnt retires in issue order on
        // gfx9, the model the compiler's own wait insertion uses): this iteration's pieces + row-table load and, from the second tile
        // on, the previous iteration's row-table load and output stores -- which thereby stay in flight under this tile's MFMAs
        if (more) { if (t > 0) pw_wait_vm<1 + NRP + NTP + NXP + (RES ? NRP : 0) + 1>(); else pw_wait_vm<NXP + (RES ? NRP : 0) + 1>(); }
        else pw_wait_vm<1>();
        __syncthreads();

        const char* xb = xbuf + (t & 1) * XB;
        char* rb = rbuf + (RES ? (t & 1) : 0) * RB;
        pw_f32x16 acc[FP];
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int px = j * 32 + frow;
                const pw_bf16x8 b = *reinterpret_cast<const pw_bf16x8*>(xb + px * (K * 2) + (((ks * 2 + fhalf) ^ pw_swz<CPX>(px)) << 4));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b, acc[j], 0, 0, 0);
            }
        // ---- finish in LDS: the lane's four consecutive channels of a pixel are 8 bytes of the (shortcut) tile, overwritten in place
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int px = j * 32 + frow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                char* p = rb + px * (BC * 2) + (((wave * 4 + g) ^ (px & 7)) << 4) + fhalf * 8;
                float v0 = __builtin_fmaf(acc[j][g * 4 + 0], 1.0f, bv[g].x), v1 = __builtin_fmaf(acc[j][g * 4 + 1], 1.0f, bv[g].y);
                float v2 = __builtin_fmaf(acc[j][g * 4 + 2], 1.0f, bv[g].z), v3 = __builtin_fmaf(acc[j][g * 4 + 3], 1.0f, bv[g].w);
                if (RES) {
                    const uint2 r = *reinterpret_cast<const uint2*>(p);
                    v0 += pw_bf2f(r.x & 0xFFFFu) * 1.0f; v1 += pw_bf2f(r.x >> 16) * 1.0f;
                    v2 += pw_bf2f(r.y & 0xFFFFu) * 1.0f; v3 += pw_bf2f(r.y >> 16) * 1.0f;
                }
                uint2 o;
                o.x = pw_pack(v0, v1); o.y = pw_pack(v2, v3);
                if (relu) { o.x = pw_relu(o.x); o.y = pw_relu(o.y); }
                *reinterpret_cast<uint2*>(p) = o;
            }
        }
        __syncthreads();
        {
            const int4* mt = meta + (t % 3) * BP;
#pragma unroll
            for (int i = 0; i < NRP; ++i) {
                const int q = i * THREADS + tid, row = q / RCH, cp = q % RCH;
                const int c = cp ^ (row & 7);
                const uint4 v = *reinterpret_cast<const uint4*>(rb + q * 16);
                const int4 e = mt[row];
                uint4* dst = e.w ? reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.out) + (size_t)e.y * a.out_cstride + c0 + c * 8) : &pw_sink[lane];
                *dst = v;
            }
        }
        if (NEXT) {
            // ---- the next block's 2a on the finished tile (its stored bf16 values): D[64 couts][BP pixels] over k = 0 .. BC-1
            if (wave < 4) {
                const int px = (wave >> 1) * 32 + frow;
                pw_f32x16 acc2;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    const pw_bf16x8 b = NEXT == 2 ? *reinterpret_cast<const pw_bf16x8*>(xb + px * (K * 2) + (((ks * 2 + fhalf) ^ pw_swz<CPX>(px)) << 4))
                                                  : *reinterpret_cast<const pw_bf16x8*>(rb + px * (BC * 2) + (((ks * 2 + fhalf) ^ (px & 7)) << 4));
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[ks], b, acc2, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 o;
                    o.x = pw_relu(pw_pack(__builtin_fmaf(acc2[g * 4 + 0], 1.0f, bv2[g].x), __builtin_fmaf(acc2[g * 4 + 1], 1.0f, bv2[g].y)));
                    o.y = pw_relu(pw_pack(__builtin_fmaf(acc2[g * 4 + 2], 1.0f, bv2[g].z), __builtin_fmaf(acc2[g * 4 + 3], 1.0f, bv2[g].w)));
                    *reinterpret_cast<uint2*>(tbuf + px * 128 + ((((wave & 1) * 4 + g) ^ (px & 7)) << 4) + fhalf * 8) = o;
                }
            }
            __syncthreads();
            {
                const int4* mt = meta + (t % 3) * BP;
                const int q = tid, row = q >> 3, cp = q & 7;           // BP * 8 = THREADS pieces: one per thread
                const int c = cp ^ (row & 7);
                const uint4 v = *reinterpret_cast<const uint4*>(tbuf + q * 16);
                const int4 e = mt[row];
                uint4* dst = e.w ? reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.ch_out3) + (size_t)e.y * C2 + c * 8) : &pw_sink[lane];
                *dst = v;
            }
        }
        asm volatile("" ::: "memory");
        if (tid < BP) meta[((t + 2) % 3) * BP + tid] = ent;     // (the compiler waits for the load; visible after the next barrier)
    }
}

template <int K, int BP, bool RES, int WGS, int BC = 128, int NEXT = 0>
static hipError_t pw_launch_cfg(const ConvArgs& a, hipStream_t s) {
    constexpr int LDS = 2 * BP * K * 2 + (RES ? 2 : 1) * BP * BC * 2 + 3 * BP * 16 + (NEXT ? BP * 128 : 0);
    static PerDeviceOnce once;
    bool& attr_set = *once.slot();
    auto kern = pw_conv_kernel<K, BP, RES, WGS, BC, NEXT>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int ptiles = (a.M + BP - 1) / BP, ctiles = a.cout_pad / BC;
    const int slots = launch_cus(a) * WGS;                  // persistent workgroups: WGS per compute unit of the launch's share
    int pstride = std::min(slots / ctiles, ptiles) / 8 * 8; // pixel-tile lanes: a multiple of the 8 XCDs
    if (pstride < 8) pstride = 8;
    hipLaunchKernelGGL(kern, dim3(pstride * ctiles), dim3(BC * 2), LDS, s, a, ptiles, ctiles, pstride);
    return hipGetLastError();
}

// true when the launch was taken by the pointwise kernel
bool conv_pointwise_eligible(const ConvArgs& a) {
    static const bool on = [] { const char* e = getenv("BOD_POINTWISE"); return !e || atoi(e) != 0; }();
    if (!on) return false;
    const ConvGroup& g = a.g[0];
    if (a.variant != 0 || a.split || a.xreuse || a.ksplit > 1 || a.groups != 1 || a.taps != 1 || a.fan_count > 1) return false;
    if (a.flags & (CONV_DROPOUT | CONV_OUT_F32 | CONV_ACCUM)) return false;
    if (g.w2 || g.ch_w2 || g.out_relu || g.agg_kind) return false;
    if (g.ch_w3 && !(a.cin == 64 && a.cout_pad == 256 && (g.res != nullptr) != (g.ch_dual != 0) && g.ch_b3 && g.ch_out3)) return false;   // fused 2a: stage 2's shapes only
    // (512-channel reductions WITH shortcut need two output buffers and fit one workgroup per CU only: +0.55 ms per 256-frame step
    //  against the generic kernel, not kept)
    // 512-channel reductions WITHOUT shortcut (stage 3's `2a`, stage 4's first block, the C3 lateral of the FPN): 128 weight registers,
    // 32-pixel tiles and a single output buffer keep two workgroups on a CU; -0.1 ms per 256-frame step (BOD_PW_K512=0: generic)
    static const bool k512 = [] { const char* e = getenv("BOD_PW_K512"); return !e || atoi(e) != 0; }();
    if (a.cin != 64 && a.cin != 128 && a.cin != 256 && !(k512 && a.cin == 512 && !g.res)) return false;
    // (64-channel cout tiles -- stage 2's `2a` reductions -- measured on this kernel: no difference to the generic 64x128 tiles, three
    //  workgroups per CU, that run them now; not kept)
    if (a.cout_pad % 128 != 0 || a.cout_valid != a.cout_pad) return false;
    if ((a.in_cstride & 7) || (a.out_cstride & 7) || (g.in_coff & 7) || (g.res && (a.res_cstride & 7))) return false;
    // persistent workgroups need a few tiles each: at least 128 output pixels per compute unit of the launch's share (32 768 on the
    // whole chip; tests/tools/planner_sweep.py: no batch from 3 frames on where the generic kernel is faster).  BOD_POINTWISE_MIN_M
    // overrides the floor (tests lower it).
    static const int min_m = [] { const char* e = getenv("BOD_POINTWISE_MIN_M"); return e ? atoi(e) : -1; }();
    if (a.M < (min_m >= 0 ? min_m : 128 * launch_cus(a))) return false;
    return true;
}

hipError_t launch_conv_pointwise(const ConvArgs& a, hipStream_t s) {
    const bool res = a.g[0].res != nullptr;
    if (a.g[0].ch_w3 && a.g[0].ch_dual) return pw_launch_cfg<64, 64, false, 1, 256, 2>(a, s);   // branch1 + the same block's 2a on one input tile (stage 2's ConvBlock)
    if (a.g[0].ch_w3) return pw_launch_cfg<64, 64, true, 1, 256, 1>(a, s);         // 2c + the next block's 2a (stage 2)
    // (32-pixel tiles with 5 / 4 workgroups per CU for the 64- / 128-channel reductions measured: +0.3 ms per 256-frame step)
    if (a.cin == 64) return res ? pw_launch_cfg<64, 64, true, 3>(a, s) : pw_launch_cfg<64, 64, false, 3>(a, s);
    if (a.cin == 128) return res ? pw_launch_cfg<128, 64, true, 2>(a, s) : pw_launch_cfg<128, 64, false, 2>(a, s);
    if (a.cin == 512) return pw_launch_cfg<512, 32, false, 2>(a, s);
    return res ? pw_launch_cfg<256, 32, true, 3>(a, s) : pw_launch_cfg<256, 32, false, 3>(a, s);
}


// ------------------------------------------------------------------------------------------------------------------------------
// Sliding-window 3x3 (stride 1, SAME), 64 -> 64 channels: ResNet stage 2's `2b` layers.  The generic kernel stages a 128-pixel tile's
// input rows once per TAP (nine K-tiles, each a fresh LDS-DMA of the same pixels shifted by one): 4.8 GB through L2 for a layer
// that reads 0.53 GB, 0.68 ms.  Here a workgroup owns a 64-pixel-wide column strip of one image and walks DOWN it: the three input
// rows of an output row live in a four-slot LDS ring, so moving one row down loads ONE new input row (64 + 2 halo pixels) while
// the current row is multiplied -- every input pixel is staged once (+3 % halo).  All 36 weight fragments of a wave (32 couts x
// 576 k) live in registers; four waves = 2 cout halves x 2 pixel fragments.  Same MFMA (32x32x16), same k order (taps outer, 16-
// channel steps inner within the single 64-channel chunk) and epilogue as the generic kernel: bit-identical.
// ------------------------------------------------------------------------------------------------------------------------------
template <int LEAD>
__global__ __launch_bounds__(256, 2) void slide3x3_c64_kernel(const ConvArgs a, int nstrips, int xsegs) {
    constexpr int SLOT = 66 * 128;                // bytes per ring slot: 64 + 2 halo pixels x 64 channels
    constexpr int RING = LEAD == 1 ? 4 : 6;       // rows y .. y+2 in use + LEAD rows landing / in flight (a power of two for one row ahead)
    constexpr int OB = 64 * 128;                  // one output row tile [64 px][64 ch] bf16
    extern __shared__ __attribute__((aligned(16))) char sl_smem[];
    char* const ring = sl_smem;                   // [RING][SLOT]
    char* const obuf = sl_smem + RING * SLOT;     // [2][OB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int wc = wave & 1, wp = wave >> 1;      // cout half, pixel fragment
    const ConvGroup& G = a.g[0];
    const int H = a.plane_h, W = a.plane_w;

    // weights: A fragments of couts wc*32 .. +31 for all 36 k-steps (tap-major, 4 x 16 channels per tap)
    pw_bf16x8 wf[36];
    {
        const char* wrow = reinterpret_cast<const char*>(G.w) + ((size_t)(wc * 32 + frow) * 576 + fhalf * 8) * 2;
#pragma unroll
        for (int k = 0; k < 36; ++k) wf[k] = *reinterpret_cast<const pw_bf16x8*>(wrow + k * 32);
    }
    float4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4*>(G.bias + wc * 32 + g * 8 + fhalf * 4);
    const bool relu = a.flags & CONV_RELU;

    for (int strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
        const int b = strip / xsegs, xs = strip - b * xsegs;
        const int x0 = xs * 64, nv = min(64, W - x0);                  // valid pixels of this strip's rows
        const int m00 = (b * H) * W + x0;
        const int4 e0 = *reinterpret_cast<const int4*>(&a.rows[m00]);  // in_off, in_pitch, out_off, res_off
        const int out_pitch = H > 1 ? a.rows[m00 + W].out_off - e0.z : 0;
        const long in0 = (long)e0.x;                                   // pixel index of the window origin of output (b, 0, x0)
        const int in_pitch = e0.y;
        // DMA of input row `ir` (relative to the strip's first input row) into ring slot ir % RING: 528 16-byte pieces, pixel = piece / 8,
        // chunk swizzled by the pixel.  Waves 1-3 issue two instructions per row, wave 0 a third one for the last 16 pieces.
        auto issue_row = [&](int ir) {
            char* slot = ring + (ir % RING) * SLOT;
            auto piece = [&](int i) {
                const int q = i * 256 + tid;
                int px = q >> 3;
                const int cp = q & 7;
                const int c = cp ^ pw_swz<8>(px);
                px = px < nv + 2 ? px : 0;                             // (pieces past the row's halo re-read its first pixel)
                const char* src = reinterpret_cast<const char*>(G.in) + (((size_t)(in0 + (long)ir * in_pitch + px)) * a.in_cstride + G.in_coff + c * 8) * 2;
                __builtin_amdgcn_global_load_lds(PW_GLOBAL_PTR(src), PW_LDS_PTR(slot + (i * 256 + wave * 64) * 16), 16, 0, 0);
            };
            piece(0); piece(1);
            if (wave == 0 && lane < 16) piece(2);
        };
        __syncthreads();                          // the previous strip's last reads of the ring / output tiles are done
        issue_row(0); issue_row(1); issue_row(2);
        if (LEAD == 2 && H >= 2) issue_row(3);
        for (int y = 0; y < H; ++y) {
            const bool pre = y + LEAD < H;        // row y+2+LEAD exists (the padded plane has H+2 rows)
            if (pre) issue_row(y + 2 + LEAD);
            asm volatile("" ::: "memory");
            // rows y .. y+2 have landed once at most the instructions issued BEHIND row y+2's pieces are outstanding (vmcnt retires in
            // issue order): the pieces of rows y+3 and y+4 (D each: 3 on wave 0, 2 elsewhere) and the two stores of each of the last
            // two output rows.  The last two rows of a strip wait for everything.
            if (!pre) pw_wait_vm<0>();
            else if (LEAD == 2) {
                if (wave == 0) { if (y == 0) pw_wait_vm<6>(); else if (y == 1) pw_wait_vm<8>(); else pw_wait_vm<10>(); }
                else { if (y == 0) pw_wait_vm<4>(); else if (y == 1) pw_wait_vm<6>(); else pw_wait_vm<8>(); }
            } else {
                if (wave == 0) { if (y == 0) pw_wait_vm<3>(); else pw_wait_vm<5>(); }
                else { if (y == 0) pw_wait_vm<2>(); else pw_wait_vm<4>(); }
            }
            __syncthreads();
            pw_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const char* slot = ring + ((y + ky) % RING) * SLOT;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int px = wp * 32 + frow + kx;
                        const pw_bf16x8 bf = *reinterpret_cast<const pw_bf16x8*>(slot + px * 128 + (((ks * 2 + fhalf) ^ pw_swz<8>(px)) << 4));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(ky * 3 + kx) * 4 + ks], bf, acc, 0, 0, 0);
                    }
            }
            char* ob = obuf + (y & 1) * OB;
            {
                const int px = wp * 32 + frow;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 o;
                    o.x = pw_pack(__builtin_fmaf(acc[g * 4 + 0], 1.0f, bv[g].x), __builtin_fmaf(acc[g * 4 + 1], 1.0f, bv[g].y));
                    o.y = pw_pack(__builtin_fmaf(acc[g * 4 + 2], 1.0f, bv[g].z), __builtin_fmaf(acc[g * 4 + 3], 1.0f, bv[g].w));
                    if (relu) { o.x = pw_relu(o.x); o.y = pw_relu(o.y); }
                    *reinterpret_cast<uint2*>(ob + px * 128 + (((wc * 4 + g) ^ (px & 7)) << 4) + fhalf * 8) = o;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 256 + tid, px = q >> 3, cp = q & 7;
                const int c = cp ^ (px & 7);
                const uint4 v = *reinterpret_cast<const uint4*>(ob + q * 16);
                uint4* dst = px < nv ? reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.out) + ((size_t)e0.z + (size_t)y * out_pitch + px) * a.out_cstride + c * 8)
                                     : &pw_sink[lane];
                *dst = v;
            }
            asm volatile("" ::: "memory");
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Sliding-window 3x3 (stride 1, SAME), 128 -> 128 channels: ResNet stage 3's `2b` layers (round 4).  The generic 128x128 tile ran them
// at 0.79 PFLOP/s (tests/tools/op_table.py: 780 us per 512 frames for 247 us of MFMA work) -- LDS-bound: every K-tile re-stages the
// tile's pixels for one tap, and a 64x64 wave tile reads four fragments per four MFMAs.  Same walk as the 64-channel kernel above --
// a workgroup owns a 64-pixel column strip, three input rows of an output row in a four-slot ring, every input pixel staged once --
// but a wave cannot hold 32 couts x 1152 k of weights (288 registers): EIGHT waves = 4 cout blocks x 2 K-HALVES (input channels
// 0-63 / 64-127 of every tap, 36 weight fragments = 144 registers each, two waves per SIMD); a wave multiplies both 32-pixel fragments
// of the row with its half of K, the two halves of a cout block trade one fragment's partial sums through LDS and each finishes one
// (bias, ReLU, pack).  The sum of a pixel is (channels 0-63, taps in order) + (channels 64-127, taps in order): the generic kernel's
// k order with ONE fp32 addition re-associated -- not bit-identical to it (tests/test_gpu_forward.py compares at the rounding floor),
// deterministic.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void slide3x3_c128_kernel(const ConvArgs a, int nstrips, int xsegs) {
    constexpr int SLOT = 66 * 256;                // bytes per ring slot: 64 + 2 halo pixels x 128 channels
    constexpr int RING = 4;
    constexpr int OB = 64 * 256;                  // one output row tile [64 px][128 ch] bf16
    extern __shared__ __attribute__((aligned(16))) char sl_smem[];
    char* const ring = sl_smem;                   // [RING][SLOT]
    char* const obuf = sl_smem + RING * SLOT;     // [2][OB]
    float4* const xch = reinterpret_cast<float4*>(sl_smem + RING * SLOT + 2 * OB);      // [8 waves][4 groups][64 lanes] float4: a fragment's partial sums
    float* const sbias = reinterpret_cast<float*>(sl_smem + RING * SLOT + 2 * OB + 8 * 4096);   // [128] (registers are short: 144 hold the weights)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhalf = lane >> 5;
    const int cb = wave & 3, kh = wave >> 2;      // cout block of 32, K half (input channels kh*64 .. +63); this wave finishes pixel fragment kh
    const ConvGroup& G = a.g[0];
    const int H = a.plane_h, W = a.plane_w;

    // weights: A fragments of couts cb*32 .. +31 for the 36 k-steps of this K half (tap-major, 4 x 16 channels per tap)
    pw_bf16x8 wf[36];
    {
        const char* wrow = reinterpret_cast<const char*>(G.w) + ((size_t)(cb * 32 + frow) * 1152 + kh * 64 + fhalf * 8) * 2;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[t * 4 + ks] = *reinterpret_cast<const pw_bf16x8*>(wrow + (t * 128 + ks * 16) * 2);
    }
    if (tid < 128) sbias[tid] = G.bias[tid];
    const bool relu = a.flags & CONV_RELU;

    for (int strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
        const int b = strip / xsegs, xs = strip - b * xsegs;
        const int x0 = xs * 64, nv = min(64, W - x0);
        const int m00 = (b * H) * W + x0;
        const int4 e0 = *reinterpret_cast<const int4*>(&a.rows[m00]);  // in_off, in_pitch, out_off, res_off
        const int out_pitch = H > 1 ? a.rows[m00 + W].out_off - e0.z : 0;
        const long in0 = (long)e0.x;
        const int in_pitch = e0.y;
        // DMA of input row `ir` into ring slot ir % RING: 1056 16-byte pieces, pixel = piece / 16, chunk swizzled by the pixel.
        // Every wave issues two instructions per row, the lower half of wave 0 a third one for the last 32 pieces.
        auto issue_row = [&](int ir) {
            char* slot = ring + (ir % RING) * SLOT;
            auto piece = [&](int i) {
                const int q = i * 512 + tid;
                int px = q >> 4;
                const int cp = q & 15;
                // (pixel & 7, not pw_swz: the conflict-free form measured SLOWER here -- 590 -> 636 us per launch with 95 % fewer bank
                //  conflicts and 42 % fewer LDS cycles; the kernel is not LDS-bound, and its waves wait longer on fragment reads that way)
                const int c = cp ^ (px & 7);
                px = px < nv + 2 ? px : 0;
                const char* src = reinterpret_cast<const char*>(G.in) + (((size_t)(in0 + (long)ir * in_pitch + px)) * a.in_cstride + G.in_coff + c * 8) * 2;
                __builtin_amdgcn_global_load_lds(PW_GLOBAL_PTR(src), PW_LDS_PTR(slot + (i * 512 + wave * 64) * 16), 16, 0, 0);
            };
            piece(0); piece(1);
            if (wave == 0 && lane < 32) piece(2);
        };
        __syncthreads();                          // the previous strip's last reads of the ring / output tiles are done
        issue_row(0); issue_row(1); issue_row(2);
        for (int y = 0; y < H; ++y) {
            const bool pre = y + 1 < H;           // row y+3 exists (the padded plane has H+2 rows)
            if (pre) issue_row(y + 3);
            asm volatile("" ::: "memory");
            // rows y .. y+2 have landed once at most the instructions issued BEHIND row y+2's pieces are outstanding (vmcnt retires in
            // issue order): the pieces of row y+3 (3 on wave 0, 2 elsewhere) and the two stores of the previous output row
            if (!pre) pw_wait_vm<0>();
            else if (wave == 0) { if (y == 0) pw_wait_vm<3>(); else pw_wait_vm<5>(); }
            else { if (y == 0) pw_wait_vm<2>(); else pw_wait_vm<4>(); }
            __syncthreads();
            pw_f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const char* slot = ring + ((y + ky) % RING) * SLOT;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int px = j * 32 + frow + kx;
                            const pw_bf16x8 bf = *reinterpret_cast<const pw_bf16x8*>(slot + px * 256 + (((kh * 8 + ks * 2 + fhalf) ^ (px & 7)) << 4));
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(ky * 3 + kx) * 4 + ks], bf, acc[j], 0, 0, 0);
                        }
            }
            // the fragment the OTHER K half finishes goes to it through LDS
            {
                float4* mine = xch + wave * 256;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const pw_f32x16& v = acc[1 - kh];
                    mine[g * 64 + lane] = make_float4(v[g * 4 + 0], v[g * 4 + 1], v[g * 4 + 2], v[g * 4 + 3]);
                }
            }
            __syncthreads();
            char* ob = obuf + (y & 1) * OB;
            {
                const float4* theirs = xch + (wave ^ 4) * 256;
                const int px = kh * 32 + frow;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t = theirs[g * 64 + lane];
                    const pw_f32x16& v = acc[kh];
                    // (channels 0-63) + (channels 64-127), whichever half this wave holds
                    const float s0 = kh ? t.x + v[g * 4 + 0] : v[g * 4 + 0] + t.x, s1 = kh ? t.y + v[g * 4 + 1] : v[g * 4 + 1] + t.y;
                    const float s2 = kh ? t.z + v[g * 4 + 2] : v[g * 4 + 2] + t.z, s3 = kh ? t.w + v[g * 4 + 3] : v[g * 4 + 3] + t.w;
                    const float4 bvg = *reinterpret_cast<const float4*>(sbias + cb * 32 + g * 8 + fhalf * 4);
                    uint2 o;
                    o.x = pw_pack(__builtin_fmaf(s0, 1.0f, bvg.x), __builtin_fmaf(s1, 1.0f, bvg.y));
                    o.y = pw_pack(__builtin_fmaf(s2, 1.0f, bvg.z), __builtin_fmaf(s3, 1.0f, bvg.w));
                    if (relu) { o.x = pw_relu(o.x); o.y = pw_relu(o.y); }
                    *reinterpret_cast<uint2*>(ob + px * 256 + (((cb * 4 + g) ^ (px & 7)) << 4) + fhalf * 8) = o;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 512 + tid, px = q >> 4, cp = q & 15;
                const int c = cp ^ (px & 7);
                const uint4 v = *reinterpret_cast<const uint4*>(ob + q * 16);
                uint4* dst = px < nv ? reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.out) + ((size_t)e0.z + (size_t)y * out_pitch + px) * a.out_cstride + c * 8)
                                     : &pw_sink[lane];
                *dst = v;
            }
            asm volatile("" ::: "memory");
        }
    }
}

bool conv_slide3x3_eligible(const ConvArgs& a) {
    static const bool on = [] { const char* e = getenv("BOD_SLIDE3X3"); return !e || atoi(e) != 0; }();
    if (!on) return false;
    const ConvGroup& g = a.g[0];
    if (a.variant != 0 || a.split || a.xreuse || a.ksplit > 1 || a.groups != 1 || a.taps != 9 || a.KW != 3 || a.fan_count > 1) return false;
    if (a.flags & (CONV_DROPOUT | CONV_OUT_F32 | CONV_ACCUM)) return false;
    if (g.w2 || g.ch_w2 || g.out_relu || g.agg_kind || g.res) return false;
    const bool c64 = a.cin == 64 && a.cout_pad == 64 && a.cout_valid == 64;
    // (128 -> 128: BOD_SLIDE3X3_C128=0 plans the generic launches)
    static const bool c128_on = [] { const char* e = getenv("BOD_SLIDE3X3_C128"); return !e || atoi(e) != 0; }();
    const bool c128 = c128_on && a.cin == 128 && a.cout_pad == 128 && a.cout_valid == 128;
    if ((!c64 && !c128) || a.plane_h < 1 || a.plane_w < 1) return false;
    if ((a.in_cstride & 7) || (a.out_cstride & 7) || (g.in_coff & 7)) return false;
    if (a.M % (a.plane_h * a.plane_w) != 0) return false;
    // A workgroup owns one 64-pixel column strip of one image for the whole launch and walks down it: the grid is images x strips,
    // two workgroups per compute unit.  The rule is therefore in WORKGROUPS against compute units, not in pixels: round 3's pixel floor
    // sent 3 frames of 512x512 here -- 6 workgroups on 256 CUs, 172 us per launch against ~18 for the generic kernel -- and every batch
    // up to ~100 frames paid for it.  Measured per N=1 forward (tests/tools/planner_sweep.py, sliding window vs generic): 512x512
    // (2 strips) 3 / 8 / 32 / 128 / 256 frames +33 / +20 / +7.5 / -0.3 / -1.8 %, 384x1248 (5 strips) +18 / +9 / +2 / -1 / -1.8 %:
    // it pays from 1.5 workgroups per compute unit on.  BOD_POINTWISE_MIN_M >= 0 (tests) replaces the rule by that pixel floor.
    static const int min_m = [] { const char* e = getenv("BOD_POINTWISE_MIN_M"); return e ? atoi(e) : -1; }();
    if (min_m >= 0) return a.M >= min_m;
    const long nstrips = (long)(a.M / (a.plane_h * a.plane_w)) * ((a.plane_w + 63) / 64);
    // (the 128-channel kernel: one workgroup of eight waves per compute unit -- from one strip per CU on)
    if (c128) return nstrips >= (long)launch_cus(a);
    return 2 * nstrips >= 3 * (long)launch_cus(a);
}

hipError_t launch_conv_slide3x3(const ConvArgs& a, hipStream_t s) {
    if (a.cin == 128) {
        constexpr int LDS128 = 4 * 66 * 256 + 2 * 64 * 256 + 8 * 4096 + 512;
        static PerDeviceOnce once128;
        bool& set128 = *once128.slot();
        if (!set128) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(slide3x3_c128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
            if (e != hipSuccess) return e;
            set128 = true;
        }
        const int xsegs = (a.plane_w + 63) / 64;
        const int nstrips = a.M / (a.plane_h * a.plane_w) * xsegs;
        hipLaunchKernelGGL(slide3x3_c128_kernel, dim3(std::min(nstrips, launch_cus(a))), dim3(512), LDS128, s, a, nstrips, xsegs);
        return hipGetLastError();
    }
    constexpr int LDS = 6 * 66 * 128 + 2 * 64 * 128;
    static PerDeviceOnce once;
    bool& attr_set = *once.slot();
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(slide3x3_c64_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(slide3x3_c64_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int xsegs = (a.plane_w + 63) / 64;
    const int nstrips = a.M / (a.plane_h * a.plane_w) * xsegs;
    // rows of prefetch: one (default) or two (BOD_SLIDE_LEAD=2).  Measured per 256-frame step, same box: backbone 22.96 ms with one row
    // ahead, 23.11-23.17 with two (a sixth ring slot and longer wait chains for nothing: the row in flight is not what a row waits for),
    // 23.69 on the generic kernel
    static const int lead = [] { const char* e = getenv("BOD_SLIDE_LEAD"); return e ? atoi(e) : 1; }();
    const int wgs = std::min(nstrips, 2 * launch_cus(a));             // two workgroups per compute unit; the rest of the strips in further rounds
    if (lead == 2) hipLaunchKernelGGL(slide3x3_c64_kernel<2>, dim3(wgs), dim3(256), LDS, s, a, nstrips, xsegs);
    else hipLaunchKernelGGL(slide3x3_c64_kernel<1>, dim3(wgs), dim3(256), 4 * 66 * 128 + 2 * 64 * 128, s, a, nstrips, xsegs);
    return hipGetLastError();
}

// Plan-time question of engine.hip: may this 1x1 expansion (64 -> 256 with shortcut, ResNet stage 2) carry the next block's 2a?  Only
// when the launch is certain to run on the pointwise kernel (the generic kernel knows nothing of ConvGroup.ch_w3 on its own).
bool conv_pointwise_can_fuse_next(const ConvArgs& a) {
    static const bool on = [] { const char* e = getenv("BOD_PW_FUSE_NEXT"); return !e || atoi(e) != 0; }();
    static const bool forced = [] { const char* e = getenv("BOD_FORCE_CONV_TILE"); return e && atoi(e) != 0; }();
    static const bool nt = [] { const char* e = getenv("BOD_NT_STORES"); return e && (atoi(e) & 1); }();
    static const bool res_reg = [] { const char* e = getenv("BOD_RES_REGISTER"); return e && atoi(e) == 1; }();   // (sends the launch to a variant build)
    if (!on || forced || nt || res_reg) return false;
    return a.cin == 64 && a.cout_pad == 256 && a.g[0].res && conv_pointwise_eligible(a);
}

// ... and may a 64 -> 256 projection shortcut (no residual of its own) carry its block's 2a on the same input tile (dual form)?
bool conv_pointwise_can_fuse_dual(const ConvArgs& a) {
    static const bool on = [] { const char* e = getenv("BOD_PW_FUSE_DUAL"); return !e || atoi(e) != 0; }();
    static const bool forced = [] { const char* e = getenv("BOD_FORCE_CONV_TILE"); return e && atoi(e) != 0; }();
    static const bool nt = [] { const char* e = getenv("BOD_NT_STORES"); return e && (atoi(e) & 1); }();
    static const bool res_reg = [] { const char* e = getenv("BOD_RES_REGISTER"); return e && atoi(e) == 1; }();
    if (!on || forced || nt || res_reg) return false;
    return a.cin == 64 && a.cout_pad == 256 && !a.g[0].res && !(a.flags & CONV_RELU) && conv_pointwise_eligible(a);
}
