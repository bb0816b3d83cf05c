// Implicit-GEMM convolution on gfx950 MFMA (v_mfma_f32_32x32x16_bf16), the kernel behind every
// conv of the RetinaNet forward except the 3-channel stem:
//   ResNet-50 bottlenecks   src/retina_net/models/feature_extractor.py:195-213,283-309
//   FPN laterals / outputs  src/retina_net/models/feature_decoder.py:136-171
//   head towers + dropout   src/retina_net/models/multitask_headers.py:98-123,209-230,318-342
//
// Orientation: D[cout][pixel] = W[cout][k] * X[pixel][k].  Weights are the MFMA "A" operand and
// pixels the "B" operand, so a lane's accumulator registers hold 4 CONSECUTIVE output channels of
// one pixel; one Philox4x32 call yields the 16 dropout decisions of four such groups (contract v3, philox.h).
//
// Both operands are K-contiguous (OHWI weights, NHWC activations), staged global->LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4) into 128-byte rows, double-buffered, one barrier per
// K-tile.  The LDS image is lane-linear, so the bank-conflict swizzle is applied on the SOURCE
// chunk index and again on the fragment read (chunk ^= (row>>1)&7: conflict-free for
// ds_read_b128's 16-lane groups over 128-B rows).  Zero padding is physical (padded planes), so the
// gather needs no bounds checks.
//
// The kernel is L2->LDS-bandwidth bound at small tiles (measured: loads alone 1.3 ms vs MFMA 0.8 ms
// per head-tower launch at 128x128), hence the 256(cout) x 256(pixel) 8-wave configuration for the
// 256-channel layers: every staged byte feeds twice the MACs.
//
// Epilogue (bf16 outputs): bias (+residual) (+ReLU) (+Philox dropout) in registers, transposed
// through LDS as a [pixel][cout] tile (chunk-swizzled), then stored as whole 16-byte pieces of
// contiguous NHWC pixel rows (512 B per pixel at 256 channels) instead of 8-byte scatters.
//
// bf16x3 ("split") precision mode (ConvArgs.split, template flag SPLIT): every value is stored as TWO bf16, x = hi + lo
// (exact to 2^-17 relative), 32 channels of hi followed by their 32 lo halves in every 64-slot (128-byte) group, for
// activations and weights alike.  One K-tile row is then 32 channels and the staging code is untouched (cin and all
// pixel strides are given in slots); the product is hi*hi + hi*lo + lo*hi on the same MFMA (fp32 accumulate, the
// lo*lo term is below 2^-16), i.e. six k-steps per K-tile instead of four: three times the MFMA work of bf16 mode for
// twice the bytes, and fp32-class results (end-to-end 1e-3 against the float64 oracle) on the bf16 matrix pipe.
//
// f16mx precision (ConvArgs.mx, template value MXK; round 5) -- the parity mode's head towers at HALF the matrix-pipe time of bf16x3.
// x = hi + lo with hi = f16(x) (11 bits): hi*hi is ONE f16 product, exact in the fp32 accumulator; the two cross terms hi*lo + lo*hi
// need 3-4 bits only and go through the block-scaled MX pipe (v_mfma_scale_f32_32x32x64_f8f6f4 on e2m3 operands: four times the
// f16 rate) as one product: per 16 channels a block of 32 elements {hi6, lo6'} (activations) against {lo6', hi6} (weights),
// lo' = lo * 2^11 (|lo'| <= |hi|: one shared block scale 2^e, e = floor(log2(max * 16/15)) - 2, nothing saturates; the 2^-11 is
// folded into the weights' scale byte).  1.5 bf16-product equivalents per multiplication instead of 3; per-layer error 1.3e-5 of the
// output RMS against bf16x3's 4e-6 and the 1e-3 gate (tests/tools/tower_numerics.py; operand layout and sustained rates:
// tests/tools/mx_probe.hip, profiles/round5_mx_probe.txt).  The "hx" row of a pixel / of a weight (cout, tap) keeps the bf16x3 row's
// size and slot counts (4 bytes per channel): per 64 channels a 128-byte H chunk (64 f16 hi, natural order: the k-steps of an
// f16 K-tile) and a 128-byte X chunk of four 32-byte slots (m, b) -- m = 32-channel half, b = MFMA K block = lane >> 5 -- split in
// two 16-byte pieces at X offsets 64m + 16b and 64m + 32 + 16b (k-steps 2m, 2m+1 of the K-tile: the fragment reads of the loop are
// the bf16 ones); a slot holds the 16 channels 32m + 8*g4 + 4b + r (the accumulator layout of a lane) as elements 2k = hi6 / lo6',
// 2k+1 = lo6' / hi6 (k = 4*g4 + r; activations / weights), element e at bits [6e, 6e+6), and its scale byte at byte 28.
//
// f16mx4 precision (ConvArgs.mx == 3, MXK = 3): the cross terms as e2m1 (fp4) elements -- 32 of them are FOUR operand registers at the
// e2m3 rate, so an X K-tile of 128 bytes per row covers 128 channels: per 256 channels 4 H + 2 X K-tiles per tap instead of 4 + 4, three
// quarters of the staged bytes.  The "h4" row keeps the 1 024-byte pitch: chunks [H0 H1 X0 H2 H3 X1 S -]; Hq = 64 f16 hi of channels
// 64q..; Xx piece 2 ks + half = the block of channels 128x + 32 ks + 8 g4 + 4 half + r as 32 nibbles, nibble 2k = hi4 / lo4',
// 2k + 1 = lo4' / hi4 (k = 4 g4 + r; activations / weights); S byte 8x + 4 half + ks = the block's E8M0 scale, 2^e with
// e = floor(log2(max |hi| * 4/3)) - 2 (weights: - 11).  Per-layer error 5.7e-5 rms / 2.9e-4 max of the output RMS: raw head outputs
// 6e-4 end to end, fused covariance entries 2.5e-3 -- an opt-in mode between bf16 and f16mx.
#include "kernels.h"
#include "philox.h"
#include <cstdlib>
#include <string>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(6))) int i32x6;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16v;
// the two 16-byte pieces of an MX operand (32 bytes per lane: 24 of e2m3 elements, the block's E8M0 scale in byte 28)
__device__ __forceinline__ i32x8 mx_cat(const bf16x8& p0, const bf16x8& p1) {
    return __builtin_shufflevector(__builtin_bit_cast(i32x4, p0), __builtin_bit_cast(i32x4, p1), 0, 1, 2, 3, 4, 5, 6, 7);
}

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);          // round to nearest even (finite inputs)
    return u >> 16;
}
// two fp32 -> packed bf16x2 (lo | hi << 16), round-to-nearest-even in ONE instruction on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#else
    return f32_to_bf16(lo) | (f32_to_bf16(hi) << 16);
#endif
}
// ReLU of two packed bf16 as ONE v_pk_max_i16: a negative bf16 is a negative int16 (sign bit), max with 0 clears it
// (-0 becomes +0), positive values are unchanged.  round(max(x, 0)) == max(round(x), 0) for round-to-nearest.
__device__ __forceinline__ uint32_t relu_bf16x2_pk(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(w));
    return r;
#else
    return ((w & 0x8000u) ? 0u : (w & 0xFFFFu)) | ((w & 0x80000000u) ? 0u : (w & 0xFFFF0000u));
#endif
}
// Dropout keep mask of two 16-bit uniform words at once: 0xFFFF where word >= threshold, else 0 (threshold >= 1).
// saturating (word - (threshold-1)) is non-zero exactly when the word is kept; min(.,1) * 0xFFFF spreads it.
__device__ __forceinline__ uint32_t keep_mask_u16x2(uint32_t w, uint32_t thr_m1_x2) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d, m;
    const uint32_t one = 0x00010001u, ones = 0xFFFFFFFFu;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(w), "v"(thr_m1_x2));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(d), "v"(one));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(m) : "v"(d), "v"(ones));
    return m;
#else
    const uint32_t t = (thr_m1_x2 & 0xFFFFu) + 1u;
    return ((w & 0xFFFFu) >= t ? 0x0000FFFFu : 0u) | ((w >> 16) >= t ? 0xFFFF0000u : 0u);
#endif
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w) {
    const uint32_t lo = (w & 0x8000u) ? 0u : (w & 0xFFFFu);
    const uint32_t hi = (w & 0x80000000u) ? 0u : (w & 0xFFFF0000u);
    return lo | hi;
}

template <int BC, int BP, int WC, int WP, bool XR = false>
struct ConvCfg {
    static constexpr int THREADS = 64 * WC * WP;
    static constexpr int STAGE = (BC + BP) * 128;
    static constexpr int EP_BYTES = BP * BC * 2;
    // row-reuse layout: 2 weight stages of BC rows + 2 extended-activation buffers of XR_EXT_ROWS rows
    static constexpr int XR_BYTES = 2 * BC * 128 + 2 * XR_EXT_ROWS * 128;
    static constexpr int STAGED = XR ? XR_BYTES : 2 * STAGE;
    static constexpr int MAIN = (STAGED > EP_BYTES) ? STAGED : EP_BYTES;
    // after the staging / epilogue area: per-tile metadata read by the epilogue (kept out of registers
    // during the K loop and out of global memory in the epilogue)
    static constexpr int OFF_OUT = MAIN;                 // int   [BP] output pixel (-1 = invalid row)
    static constexpr int OFF_RES = OFF_OUT + BP * 4;     // int   [BP] residual pixel
    static constexpr int OFF_RNG = OFF_RES + BP * 4;     // int2  [BP] dropout counters
    static constexpr int OFF_BIAS = OFF_RNG + BP * 8;    // float [BC]
    static constexpr int OFF_OUT2 = OFF_BIAS + BC * 4;   // int   [BP] dense row of the fused 1x1 output
    static constexpr int LDS = OFF_OUT2 + BP * 4;
};

// ABL: 0 = production (5 = the same code under its own symbol for the fan-out launch); 1 = no epilogue; 2 = no global->LDS traffic after the first tile;
//      3 = no MFMA / LDS fragment reads; 4 = dropout without the Philox call
//      (ablation builds for tests/tools/bench_head_conv.py)
// SPLIT: only the upper half of the waves issues the global->LDS staging (2x the pieces each), so the
//      lower half starts its MFMAs right after the barrier and the two waves of every SIMD run
//      out of phase (the matrix pipe stays fed while the other wave issues loads / waits).
// Phase clock of the instrumented build (variant 90, tests/tools/bench_head_conv.py): wave 0 of every
// workgroup adds the shader-clock cycles it spent between consecutive stamps; slot 15 counts tiles.
__device__ unsigned long long g_phase_cycles[16];
template <int ABL>
__device__ __forceinline__ void phase_stamp(unsigned long long& t, int slot) {
    if constexpr (ABL == 90) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[slot], now - t);
        t = now;
        __builtin_amdgcn_sched_barrier(0);
    }
}
int device_cu_count() {
    static int cus[64] = {0};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    if (cus[d] == 0) {
        hipDeviceProp_t prop;
        cus[d] = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[d];
}

void conv_igemm_phase_cycles(unsigned long long* out16, bool reset) {
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z); }
}

// One K-tile (BK = 64, four k-steps of 16) of MFMAs for a wave tile of FC x FP 32x32 fragments, software-
// pipelined in registers.  `wa` / `xb[j]` are LDS byte offsets of the lane's A / B fragment rows INCLUDING the
// swizzled chunk of k-step 0; every row base is a multiple of 128 B, so k-step ks only flips offset bits 5..6.
// A fragment i lives at wa + i*32 rows.  Every ds_read_b128 is issued >= 2-4 MFMAs before its first use and
// one read is slotted per MFMA, so the only exposed LDS wait is the first fragment set after the block barrier.
#define SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SGB_DSRD(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
// D = C tied in place: with the builtin the register allocator renames the 32 four-register accumulators of the 16x16x32 loop
// through fresh tuples on every MFMA and runs out of registers; the tied form keeps each accumulator where it is.  The
// instruction stream is then the program order (volatile), which is how the loop below is written anyway.  (Same-accumulator
// MFMAs are 32 instructions apart; the epilogue's first read of an accumulator comes after s_nops + a workgroup barrier.)
__device__ __forceinline__ void mfma16_inplace(f32x4& c, const bf16x8& a, const bf16x8& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#endif
}

// The block-scaled product of the f16mx X tiles, D = C tied in place like mfma16_inplace: the builtin form leaves the choice of the
// destination to the register allocator, which moves the 32x32 accumulators through fresh tuples and spills them (844 bytes per
// lane in the first build).  Sources come straight from LDS reads (no VALU write in front of the MFMA: kernel_guard checks).
#if defined(BOD_MX_ABL_READS)
typedef i32x4 mxop_t;          // timing ablation: fp4-format operands (same MFMA rate as fp6), ONE 16-byte read each
#else
typedef i32x6 mxop_t;
#endif
__device__ __forceinline__ void mfma_mx6_inplace(f32x16& c, const i32x4& a, const i32x4& b, const int sa, const int sb) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
#endif
}
__device__ __forceinline__ void mfma_mx6_inplace(f32x16& c, const i32x6& a, const i32x6& b, const int sa, const int sb) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
#endif
}

// (the H tiles' f16 product in the same tied form: with the builtin in one branch and tied asm in the other the allocator spills)
__device__ __forceinline__ void mfma_f16_32_inplace(f32x16& c, const bf16x8& a, const bf16x8& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#endif
}

// f16mx: an H tile (64 f16 hi per row: four f16 k-steps) or an X tile (the 64 channels' cross terms: two block-scaled e2m3 products of
// K = 64, operands = the pieces of k-steps (2m, 2m+1), scale bytes inside the operands) of the row-reuse loop.  `wa` / `xb[j]`: LDS byte
// offsets of the lane's A / B fragment rows incl. the swizzled chunk of k-step 0 (KTilePipe); `wa_n` / `xb_n`: the NEXT K-tile's.
//  * The MFMAs are tied inline asm (program order).  The A operand of the next cout-fragment step is requested in front of the current
//    step's MFMAs (double-buffered), the next k-step's B operands behind the MFMAs of the current k-step's last step, one by one as each dies.
//  * ONE barrier per K-tile, in front of the LAST step's MFMAs (`sync()`: the caller's vmcnt wait + s_barrier): by then every LDS read of
//    this K-tile has returned -- its weight stage is free for the K-tile after next -- and the next K-tile's pieces have landed, so its
//    first operands (`HxCarry`) are requested right behind the barrier, under the last step's MFMAs, and the K-tile boundary itself has
//    neither a barrier nor an exposed LDS round trip (phase clock of the top-of-K-tile form: 845 of 2 380 cycles per K-tile with no wave of
//    the SIMD issuing an MFMA; the bf16 tower loop's "barrier two fragment steps before the K-tile's end", compiler-scheduled reads here).
//  * `slot(k, wr)`, k = 0..7, behind the MFMAs of every (second, in an H tile) step -- k = 7 behind the barrier: the caller issues one
//    LDS-DMA piece there (destination = wr + offset), so that the issue (60-185 cycles with the wave parked) runs under in-flight MFMAs.
//  * `rd` and `wr` are the SAME LDS block, declared __restrict__: every read of this function goes through `rd`, every LDS-DMA destination
//    through `wr`, and they never overlap between two barriers (the stage / extended-row buffer being read against the ones being filled)
//    -- without that the compiler puts `s_waitcnt vmcnt(0)` in front of every LDS read that follows an LDS-DMA issue (it cannot tell the
//    stages apart): a full L2 / HBM round trip with the matrix pipe idle per piece.
// (f16mx4: the cross terms as ONE block-scaled e2m1 product -- four-register operands, the block's E8M0 scale = byte KS of the lane's
// scale registers: op_sel / op_sel_hi carry bit 0 / bit 1 of the byte index of A and B)
template <int KS>
__device__ __forceinline__ void mfma_mx4_inplace(f32x16& c, const bf16x8& a, const bf16x8& b, int sa, int sb) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (KS == 0) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    else if constexpr (KS == 1) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,1,0] op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    else if constexpr (KS == 2) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[0,0,0] op_sel_hi:[1,1,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    else asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,1,0] op_sel_hi:[1,1,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
#endif
}
template <int FP>
struct HxCarry {                 // the first step's operands of a K-tile, requested during the K-tile before
    bf16x8 hA, hB[FP];
    mxop_t xA, xB[FP];
    int xsa, xsb[FP];
    int sa4[4];                  // f16mx4: the K-tile's A scales (dword i = the four k-steps' bytes of cout fragment i); B scales in xsb
};
// F4 (f16mx4): X tiles are the cross terms of 128 channels as four e2m1 products of K = 64 -- operands = the 16-byte pieces of k-step ks
// like an H tile's, scales from the two scale arrays in LDS: `ws` = the lane's dword of cout fragment 0 (fragment i: + 256 i bytes),
// `xs[j]` = the dword of the lane's pixel row of fragment j; `ws_n` / `xs_n`: the next K-tile's.
template <int FC, int FP, int ROWB, bool XT, bool NXT, bool F4 = false, class SYNC, class SLOT>
__device__ __forceinline__ void hx_ktile(f32x16 (&acc)[FC][FP], const char* __restrict__ rd, char* __restrict__ wr, const int wa, const int (&xb)[FP],
                                         const int wa_n, const int (&xb_n)[FP], const bool has_next, HxCarry<FP>& c, SYNC&& sync, SLOT&& slot,
                                         const int ws_n = 0, const int* xs_n = nullptr) {
    auto ldH = [&](int base, int ks) { return *reinterpret_cast<const bf16x8*>(rd + (base ^ (ks << 5))); };
    // an X operand = 24 bytes of elements (first piece + 8 bytes of the second) + the scale byte (second piece, byte 12): read straight
    // into a 6-register tuple and one scale register.  The 8- and 4-byte reads run 2- / 4-way bank conflicts (rows r and r + 16 of a
    // 32-lane group: 42 % of the kernel's LDS cycles); the conflict-free alternative -- two 16-byte reads, registers 4, 5 copied into
    // the SIX-register operand the MFMA wants, `s_nop 1` in front of the MFMA for the copies' two wait states -- measured 3 % SLOWER on
    // the towers on two boxes (199 / 205 ms against 193.6 per 256 frames): the copies and their waits cost more than the conflicts.
    auto ld6 = [&](int base, int m, int& sc) {
#if defined(BOD_MX_ABL_READS)
        // timing ablation (wrong results): BOD_MX_ABL_READS=1 one 16-byte read per operand (fp4-format MFMA: the fp6 rate), =2 two 16-byte
        // reads (the second one's registers only kept alive), constant scale -- what conflict-free X operand reads would cost
        const i32x4 a = *reinterpret_cast<const i32x4*>(rd + (base ^ ((2 * m) << 5)));
        sc = 0x7b7b7b7b;
        if (BOD_MX_ABL_READS == 2) {
            i32x4 b = *reinterpret_cast<const i32x4*>(rd + (base ^ ((2 * m + 1) << 5)));
            asm volatile("" :: "v"(b));
        }
        return a;
#else
        const char* p1 = rd + (base ^ ((2 * m + 1) << 5));
        const i32x4 a = *reinterpret_cast<const i32x4*>(rd + (base ^ ((2 * m) << 5)));
        const int2 b = *reinterpret_cast<const int2*>(p1);
        sc = *reinterpret_cast<const int*>(p1 + 12);
        i32x6 r;
        r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b.x; r[5] = b.y;
        return r;
#endif
    };
    // X tiles: the A operand double-buffered (requested one step ahead; single-buffered measured 3 % slower on the towers)
    constexpr bool XA2 = true;
    auto ldS = [&](int off) { return *reinterpret_cast<const int*>(rd + off); };
    auto next_A = [&]() {
        if constexpr (NXT && !F4) c.xA = ld6(wa_n, 0, c.xsa);
        else {
            c.hA = ldH(wa_n, 0);
            if constexpr (NXT && F4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) c.sa4[i] = ldS(ws_n + i * 256);
            }
        }
    };
    auto next_B = [&](int j) {
        if constexpr (NXT && !F4) c.xB[j] = ld6(xb_n[j], 0, c.xsb[j]);
        else { c.hB[j] = ldH(xb_n[j], 0); if constexpr (NXT && F4) c.xsb[j] = ldS(xs_n[j]); }
    };
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!XT || F4) {
        constexpr int NIT = 4 * FC;
        static_assert(!F4 || FC == 4, "f16mx4: four cout fragments per wave");
        bf16x8 Bf[FP], Af[2];
        int sa[4] = {0, 0, 0, 0}, sb[FP];
#pragma unroll
        for (int j = 0; j < FP; ++j) { Bf[j] = c.hB[j]; sb[j] = (XT && F4) ? c.xsb[j] : 0; }
        Af[0] = c.hA;
        if constexpr (XT && F4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sa[i] = c.sa4[i];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int ks = it / FC, i = it % FC, cur = it & 1;
            if (it + 1 < NIT) Af[cur ^ 1] = ldH(wa + ((it + 1) % FC) * 32 * ROWB, (it + 1) / FC);
            if (it == NIT - 1) { sync(); if (has_next) next_A(); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                if constexpr (XT && F4) {
                    if (ks == 0) mfma_mx4_inplace<0>(acc[i][j], Af[cur], Bf[j], sa[i & 3], sb[j]);
                    else if (ks == 1) mfma_mx4_inplace<1>(acc[i][j], Af[cur], Bf[j], sa[i & 3], sb[j]);
                    else if (ks == 2) mfma_mx4_inplace<2>(acc[i][j], Af[cur], Bf[j], sa[i & 3], sb[j]);
                    else mfma_mx4_inplace<3>(acc[i][j], Af[cur], Bf[j], sa[i & 3], sb[j]);
                } else mfma_f16_32_inplace(acc[i][j], Af[cur], Bf[j]);
                if (i == FC - 1 && ks + 1 < 4) Bf[j] = ldH(xb[j], ks + 1);
                if (it == NIT - 1 && has_next) next_B(j);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (it & 1) slot(it >> 1, wr);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        constexpr int NIT = 2 * FC;
        mxop_t B6[FP], A6[2];
        int sb[FP], sa[2];
#pragma unroll
        for (int j = 0; j < FP; ++j) { B6[j] = c.xB[j]; sb[j] = c.xsb[j]; }
        A6[0] = c.xA; sa[0] = c.xsa;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m = it / FC, i = it % FC, cur = XA2 ? (it & 1) : 0;
            if (XA2) { if (it + 1 < NIT) A6[cur ^ 1] = ld6(wa + ((it + 1) % FC) * 32 * ROWB, (it + 1) / FC, sa[cur ^ 1]); }
            else if (it > 0) A6[0] = ld6(wa + i * 32 * ROWB, m, sa[0]);
            if (it == NIT - 1) { sync(); if (has_next) next_A(); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                mfma_mx6_inplace(acc[i][j], A6[cur], B6[j], sa[cur], sb[j]);
                if (i == FC - 1 && m == 0) B6[j] = ld6(xb[j], 1, sb[j]);
                if (it == NIT - 1 && has_next) next_B(j);
            }
            __builtin_amdgcn_sched_barrier(0);
            slot(it, wr);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int FC, int FP, int ROWB>
struct KTilePipe {
    const char* smem; int wa; int xb[FP];
    bf16x8 Ac[FC == 4 ? 2 : FC], Bc[FP];            // fragments of k-step 0, loaded by first_loads()
    __device__ __forceinline__ bf16x8 ldA(int i, int ks) const { return *reinterpret_cast<const bf16x8*>(smem + ((wa ^ (ks << 5)) + i * 32 * ROWB)); }
    __device__ __forceinline__ bf16x8 ldB(int j, int ks) const { return *reinterpret_cast<const bf16x8*>(smem + (xb[j] ^ (ks << 5))); }

    // Issue the first fragment set right after the block barrier; the caller then issues the next tile's
    // LDS-DMA (between two sched_barriers) under this one exposed LDS wait, and calls run().
    __device__ __forceinline__ void first_loads() {
        Ac[0] = ldA(0, 0);
#pragma unroll
        for (int j = 0; j < FP; ++j) Bc[j] = ldB(j, 0);
#pragma unroll
        for (int i = 1; i < (FC == 4 ? 2 : FC); ++i) Ac[i] = ldA(i, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // bf16x3: a K-tile row holds [hi 0..15 | hi 16..31 | lo 0..15 | lo 16..31] (k-steps 0..3 of the bf16 layout); per
    // 16-channel half h the three products hi*lo, lo*hi, hi*hi (small terms first) go into the same accumulators.
    __device__ __forceinline__ void run_split(f32x16 (&acc)[FC][FP]) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 Ah[FC], Al[FC], Bh[FP], Bl[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) { if (h == 0 && i < (FC == 4 ? 2 : FC)) Ah[i] = Ac[i]; else Ah[i] = ldA(i, h); }
#pragma unroll
            for (int j = 0; j < FP; ++j) { if (h == 0) Bh[j] = Bc[j]; else Bh[j] = ldB(j, h); }
#pragma unroll
            for (int j = 0; j < FP; ++j) Bl[j] = ldB(j, 2 + h);
#pragma unroll
            for (int i = 0; i < FC; ++i) Al[i] = ldA(i, 2 + h);
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[i], Bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[i], Bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[i], Bh[j], acc[i][j], 0, 0, 0);
        }
    }

    // The same products in the same order per accumulator with a smaller live set (B fragments of the half + one A pair at a
    // time: 32 fragment registers instead of 48) for the row-reuse loop, whose staging state leaves fewer registers.
    __device__ __forceinline__ void run_split_lean(f32x16 (&acc)[FC][FP]) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 Bh[FP], Bl[FP];
#pragma unroll
            for (int j = 0; j < FP; ++j) { if (h == 0) Bh[j] = Bc[j]; else Bh[j] = ldB(j, h); }
#pragma unroll
            for (int j = 0; j < FP; ++j) Bl[j] = ldB(j, 2 + h);
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                bf16x8 Ah, Al;
                if (h == 0 && i < (FC == 4 ? 2 : FC)) Ah = Ac[i]; else Ah = ldA(i, h);
                Al = ldA(i, 2 + h);
#pragma unroll
                for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    __device__ __forceinline__ void run(f32x16 (&acc)[FC][FP]) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FC == 4 && FP == 2) {
            // A fragments double-buffered in pairs (i = 0,1 | 2,3), B fragments across k-steps: 32 fragment registers
            bf16x8 A23[2], Bn[2];
#define MFMA_ROW(I, AF) \
    acc[I][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF, Bc[0], acc[I][0], 0, 0, 0); \
    acc[I][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF, Bc[1], acc[I][1], 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                A23[0] = ldA(2, ks); A23[1] = ldA(3, ks);          // phase P: MFMAs of A0,A1 | loads A2,A3, next B0,B1
                Bn[0] = ldB(0, ks + 1); Bn[1] = ldB(1, ks + 1);
                MFMA_ROW(0, Ac[0]) MFMA_ROW(1, Ac[1])
                SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1);
                Ac[0] = ldA(0, ks + 1); Ac[1] = ldA(1, ks + 1);    // phase Q: MFMAs of A2,A3 | loads next A0,A1
                MFMA_ROW(2, A23[0]) MFMA_ROW(3, A23[1])
                SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(2);
                Bc[0] = Bn[0]; Bc[1] = Bn[1];
            }
            A23[0] = ldA(2, 3); A23[1] = ldA(3, 3);
            MFMA_ROW(0, Ac[0]) MFMA_ROW(1, Ac[1])
            SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(2);
            MFMA_ROW(2, A23[0]) MFMA_ROW(3, A23[1])
            SGB_MFMA(4);
#undef MFMA_ROW
        } else {
            // other wave tiles (FC = 2, FP = 1 or 2 in production): all fragments of the next k-step in a second register set
            bf16x8 An[FC], Bn[FP];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) {
#pragma unroll
                    for (int i = 0; i < FC; ++i) An[i] = ldA(i, ks + 1);
#pragma unroll
                    for (int j = 0; j < FP; ++j) Bn[j] = ldB(j, ks + 1);
                }
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac[i], Bc[j], acc[i][j], 0, 0, 0);
                if (ks < 3) {
                    if constexpr (FC == 2 && FP == 2) { SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); }
                    else if constexpr (FC == 2 && FP == 1) { SGB_MFMA(1); SGB_DSRD(2); SGB_MFMA(1); SGB_DSRD(1); }
#pragma unroll
                    for (int i = 0; i < FC; ++i) Ac[i] = An[i];
#pragma unroll
                    for (int j = 0; j < FP; ++j) Bc[j] = Bn[j];
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// MC aggregation behind the fused 1x1 head output conv (ConvGroup.agg_kind; SURVEY.md section 7 step 4, north_star "Welford
// covariance reduction in LDS").  The tile's fp32 head outputs sit in LDS as ytile[row][ystride], row = pixel_slot * N +
// sample; one thread per (pixel slot, anchor) walks the N samples.  The arithmetic of AGG_CLS / AGG_COV is, operation for
// operation, that of post_sample_kernel / post_fuse_kernel on the raw [B,N,A,.] tensors (post_kernels.hip), so the sums are
// bit-identical to the unfused path; AGG_BOX is Welford's update of the mean and the co-moment matrix of the decoded boxes
// (inference_utils.py:220-244 computes the same two-pass).
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
template <int C>
__device__ __forceinline__ void agg_reduce_cls(const ConvGroup& G, const float* ytile, int ystride, const int* s_off, const int* s_off2,
                                               int tid, int nthreads, int rows) {
    const int N = G.agg_n, AN = G.cout2 / C, Q = rows / N;
    for (int item = tid; item < Q * AN; item += nthreads) {
        const int q = item / AN, an = item - q * AN, r0 = q * N;
        if (s_off[r0] < 0) continue;
        const int o0 = s_off2[r0];                       // (image * N + 0) * P + pixel
        const int b = o0 / (N * G.agg_P), p = o0 - b * N * G.agg_P;
        float mp[C];
#pragma unroll
        for (int j = 0; j < C; ++j) mp[j] = 0.f;
        for (int n = 0; n < N; ++n) {
            const float* l = ytile + (size_t)(r0 + n) * ystride + an * C;
            float v[C];
#pragma unroll
            for (int k = 0; k < C / 4; ++k) {
                const float4 t = reinterpret_cast<const float4*>(l)[k];
                v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
            }
            float mx = v[0];
#pragma unroll
            for (int j = 1; j < C; ++j) mx = fmaxf(mx, v[j]);
            float sden = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) { v[j] = expf(v[j] - mx); sden += v[j]; }
#pragma unroll
            for (int j = 0; j < C; ++j) mp[j] += v[j] / sden;
        }
        float* o = G.agg_out + (((size_t)b * G.agg_P + p) * AN + an) * C;
#pragma unroll
        for (int k = 0; k < C / 4; ++k) reinterpret_cast<float4*>(o)[k] = make_float4(mp[4 * k], mp[4 * k + 1], mp[4 * k + 2], mp[4 * k + 3]);
    }
}

__device__ __forceinline__ void agg_reduce_box(const ConvGroup& G, const float* ytile, int ystride, const int* s_off, const int* s_off2,
                                               int tid, int nthreads, int rows) {
    const int N = G.agg_n, AN = G.cout2 / 4, Q = rows / N;
    for (int item = tid; item < Q * AN; item += nthreads) {
        const int q = item / AN, an = item - q * AN, r0 = q * N;
        if (s_off[r0] < 0) continue;
        const int o0 = s_off2[r0];
        const int b = o0 / (N * G.agg_P), p = o0 - b * N * G.agg_P;
        const float4 anc = reinterpret_cast<const float4*>(G.anchors)[(size_t)p * AN + an];
        float mean[4] = {0.f, 0.f, 0.f, 0.f}, m2[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) m2[k] = 0.f;
        for (int n = 0; n < N; ++n) {
            const float4 t = *reinterpret_cast<const float4*>(ytile + (size_t)(r0 + n) * ystride + an * 4);
            // box_utils.box_from_anchor_and_target_bnms (:171-192), as post_kernels.hip decode_box
            float x[4];
            x[0] = anc.z * t.x / 10.0f + anc.x;
            x[1] = anc.w * t.y / 10.0f + anc.y;
            x[2] = anc.z * fminf(fmaxf(expf(t.z / 5.0f), 1e-4f), 1e4f);
            x[3] = anc.w * fminf(fmaxf(expf(t.w / 5.0f), 1e-4f), 1e4f);
            // Welford: d = x - mean; mean += d / k; M2[i][j] += d_i * (x_j - mean_j)
            const float inv = 1.0f / (float)(n + 1);
            float d[4], e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { d[i] = x[i] - mean[i]; mean[i] += d[i] * inv; e[i] = x[i] - mean[i]; }
            int k = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) m2[k++] += d[i] * e[j];
        }
        float* o = G.agg_out + (((size_t)b * G.agg_P + p) * AN + an) * 16;
        reinterpret_cast<float4*>(o)[0] = make_float4(mean[0], mean[1], mean[2], mean[3]);
        reinterpret_cast<float4*>(o)[1] = make_float4(m2[0], m2[1], m2[2], m2[3]);
        reinterpret_cast<float4*>(o)[2] = make_float4(m2[4], m2[5], m2[6], m2[7]);
        reinterpret_cast<float4*>(o)[3] = make_float4(m2[8], m2[9], 0.f, 0.f);
    }
}

__device__ __forceinline__ void agg_reduce_cov(const ConvGroup& G, const float* ytile, int ystride, const int* s_off, const int* s_off2,
                                               int tid, int nthreads, int rows) {
    const int N = G.agg_n, CH = G.cout2, Q = rows / N;
    for (int item = tid; item < Q * CH; item += nthreads) {
        const int q = item / CH, c = item - q * CH, r0 = q * N;
        if (s_off[r0] < 0) continue;
        const int o0 = s_off2[r0];
        const int b = o0 / (N * G.agg_P), p = o0 - b * N * G.agg_P;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += ytile[(size_t)(r0 + n) * ystride + c];
        G.agg_out[((size_t)b * G.agg_P + p) * CH + c] = acc;
    }
}
#pragma clang fp contract(fast)

template <int BC, int BP, int WC, int WP, int ABL, bool XR, bool SPLIT = false, int MXK = 0>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, const int gz, const int bx, const int by, char* smem) {
    static_assert(!(SPLIT && ABL != 0), "the bf16x3 mode exists as production build only");
    static_assert(MXK == 0 || (SPLIT && XR), "f16mx: the row-reuse loop of the (hi, lo) data path only");
    // the f16mx loop; 11 = its phase-clock twin (tests/tools/bench_head_conv.py variant 90), 12 = the same without the loop's LDS-DMA (variant 91)
    constexpr bool MXL = MXK == 1 || MXK == 11 || MXK == 12 || MXK == 3;
    constexpr bool MX4 = MXK == 3;          // f16mx4: h4 rows in (4 H chunks + 2 X chunks of e2m1 cross terms + the scale bytes)
    constexpr bool MXI = MXK == 11 || MXK == 12;
    unsigned long long mx_t_wait = 0, mx_t_body = 0, mx_t0 = 0, mx_t_cnt = 0, mx_t_h = 0, mx_t_x = 0;          // (_h / _x, round 6: whole H / X K-tiles)
    if constexpr (MXI) mx_t0 = __builtin_amdgcn_s_memtime();
    using Cfg = ConvCfg<BC, BP, WC, WP, XR>;
    constexpr int THREADS = Cfg::THREADS;
    constexpr int LTHREADS = THREADS;            // threads that stage
    constexpr int RPI = LTHREADS / 8;            // tile rows covered by one staging instruction
    constexpr int BK = 64;                       // bf16 per K-tile row (128 B)
    constexpr int ROWB = BK * 2;
    constexpr int W_BYTES = BC * ROWB, STAGE = Cfg::STAGE;
    constexpr int NW = BC / RPI, NX = BP / RPI;  // 16-B chunks per thread per tile
    constexpr int WTC = BC / WC, WTP = BP / WP;
    constexpr int FC = WTC / 32, FP = WTP / 32;
    constexpr int CPR = BC / 8;                  // 16-B chunks per pixel row of the epilogue tile
    static_assert(BC % RPI == 0 && BP % RPI == 0 && RPI % 16 == 0, "tile / thread mismatch");
    int* s_off = reinterpret_cast<int*>(smem + Cfg::OFF_OUT);
    int* s_res = reinterpret_cast<int*>(smem + Cfg::OFF_RES);
    int2* s_rng = reinterpret_cast<int2*>(smem + Cfg::OFF_RNG);
    float* s_bias = reinterpret_cast<float*>(smem + Cfg::OFF_BIAS);
    int* s_off2 = reinterpret_cast<int*>(smem + Cfg::OFF_OUT2);

    unsigned long long tstamp = 0;
    unsigned long long treal = 0;
    if constexpr (ABL == 90) { tstamp = __builtin_amdgcn_s_memtime(); treal = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[15], 1ull); }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    // split-K exists only in the 4-wave configurations (the 256x256 tile has no registers to spare for it)
    constexpr bool CAN_SPLITK = (BC <= 128) && !XR;
    const int nsplit = (CAN_SPLITK && a.ksplit > 1) ? a.ksplit : 1;
    const int grp = gz / nsplit, kpart = gz - grp * nsplit;          // split-K: this workgroup reduces chunks [c_begin, c_begin + cpt)
    const ConvGroup& G = a.g[grp];
    const int bp0 = bx * BP, bc0 = by * BC;
    const int cpt = a.cin / BK / nsplit;         // K-tiles per tap (of this split)
    const int c_begin = kpart * cpt;
    const int KT = MX4 ? a.taps * 6 : a.taps * cpt;          // (h4 rows: six of the eight 128-byte chunks are K-tiles)
    const int wrow = a.taps * a.cin;             // elements per weight row [cout][taps][cin]

    // ---- per-thread staging descriptors
    constexpr bool loader = true;
    const int ltid = tid;
    const int lwave = __builtin_amdgcn_readfirstlane(ltid >> 6);
    const int ldrow = ltid >> 3;                             // 0..RPI-1 (+RPI*i)
    const int ldchunk = (ltid & 7) ^ ((ltid >> 4) & 7);      // source chunk (pre-swizzled)
    const char* xsrc[NX];
    int xpitch[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        int m = bp0 + i * RPI + ldrow;
        m = m < a.M ? m : a.M - 1;
        const int2 e = *reinterpret_cast<const int2*>(&a.rows[m]);
        xsrc[i] = reinterpret_cast<const char*>(G.in) +
                  ((size_t)e.x * a.in_cstride + G.in_coff + c_begin * BK + ldchunk * 8) * 2;
        xpitch[i] = e.y * a.in_cstride * 2;
    }
    const char* wsrc[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int co = bc0 + i * RPI + ldrow;
        wsrc[i] = reinterpret_cast<const char*>(G.w) + ((size_t)co * wrow + c_begin * BK + ldchunk * 8) * 2;
    }
    // Row-reuse loop: K-tile 0's weights depend on nothing but the tile's cout range -- their LDS-DMA goes out FIRST, in front of
    // the row-table loads below, whose L2 / HBM round trip it then shares instead of following it (a fresh workgroup's LDS is free;
    // the persistent form ends every tile with a workgroup barrier).
    constexpr bool EARLY_W = XR && ABL != 81;
    if constexpr (EARLY_W) {
        const int wrs0 = RPI * wrow * 2;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            int off = i * wrs0;
            asm volatile("" : "+s"(off));
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wsrc[0] + off), LDS_PTR(smem + (i * THREADS + lwave * 64) * 16), 16, 0, 0);
        }
    }
    for (int i = tid; i < BP; i += THREADS) {
        const int m = bp0 + i;
        const int mm = m < a.M ? m : a.M - 1;
        const int4 e0 = *reinterpret_cast<const int4*>(&a.rows[mm]);
        s_off[i] = m < a.M ? e0.z : -1;
        s_res[i] = e0.w;
        s_rng[i] = *(reinterpret_cast<const int2*>(&a.rows[mm]) + 2);
        s_off2[i] = a.rows[mm].pad0;
    }
    // bias pre-multiplied by the dropout keep scale: the bf16 epilogue computes max(fma(acc, scale, bias*scale), 0),
    // i.e. relu(acc + bias) * scale with one instruction less per element (scale = 1 without dropout: acc + bias exactly)
    const float epi_scale = (a.flags & CONV_DROPOUT) && !(a.flags & CONV_OUT_F32) ? a.drop_scale : 1.0f;
    for (int i = tid; i < BC; i += THREADS) s_bias[i] = G.bias[bc0 + i] * epi_scale;

    // K order: channel chunk OUTER, taps inner -- the 9 taps of one 64-channel chunk re-read (shifted)
    // the same activation rows in 9 consecutive K-tiles, so the re-reads hit L1/L2 instead of MALL/HBM
    auto issue_w = [&](int stage, int ky, int kx, int cc) {
        char* sb = smem + stage * STAGE;
        const int woff = ((ky * a.KW + kx) * a.cin + cc * BK) * 2;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wsrc[i] + woff),
                                             LDS_PTR(sb + (i * LTHREADS + lwave * 64) * 16), 16, 0, 0);
    };
    auto issue_x = [&](int stage, int ky, int kx, int cc) {
        char* sb = smem + stage * STAGE;
        const int tapoff = (kx * a.in_cstride + cc * BK) * 2;
#pragma unroll
        for (int i = 0; i < NX; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(xsrc[i] + ky * xpitch[i] + tapoff),
                                             LDS_PTR(sb + W_BYTES + (i * LTHREADS + lwave * 64) * 16), 16, 0, 0);
    };
    auto issue = [&](int stage, int ky, int kx, int cc) { issue_w(stage, ky, kx, cc); issue_x(stage, ky, kx, cc); };

    f32x16 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // M16: the row-reuse tower kernel multiplies with v_mfma_f32_16x16x32_bf16.  Under the power limit of a launch that
    // keeps all 1 024 matrix pipes busy on random data, a register-resident loop of that instruction sustains 2.1 PF/s on
    // MI355X, the 32x32x16 form 1.3 PF/s (tests/tools/mfma_power.hip) -- the rate at which the 32x32 loop and the vendor
    // library's GEMMs level off.  Same LDS image, same swizzle (conflict-free for the 16-row fragments too), same bytes per
    // FLOP (wave tile 128 x 64 = 8 x 4 fragments of 16 x 16): a lane then holds, per fragment, 4 consecutive couts
    // (fc*16 + (lane>>4)*4 + r) of pixel fp*16 + (lane & 15).
    constexpr bool M16 = XR && !SPLIT && ABL != 81 && BC == 256 && BP == 256 && WC == 2 && WP == 4;
    constexpr int FC16 = M16 ? 8 : 1, FP16 = M16 ? 4 : 1;
    f32x4 acc4[FC16][FP16];
#pragma unroll
    for (int i = 0; i < FC16; ++i)
#pragma unroll
        for (int j = 0; j < FP16; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The loop's MFMAs are inline asm: the compiler does not give them the two wait states an MFMA needs behind a VALU write of one of
    // its sources (tests/tools/mfma_war_probe.hip: 18 % wrong results without them).  Left alone it may sink the zeroing of an
    // accumulator to just in front of its first MFMA; naming the accumulators as operands of an asm statement pins the zeroing here,
    // far ahead of the loop.  (tests/test_kernel_resources.py checks the disassembly for this and for copies behind an MFMA.)
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < FC16; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(acc4[i][0]), "+v"(acc4[i][1]), "+v"(acc4[i][2]), "+v"(acc4[i][3]));
#endif
        }
    }
    const int l15 = lane & 15, q4 = lane >> 4;

    const int frow = lane & 31;
    const int fswz = (frow >> 1) & 7;
    const int fhalf = lane >> 5;
    int ky = 0, kx = 0, cc = 0;
    const int KH = a.taps / a.KW;
    phase_stamp<ABL>(tstamp, 0);            // tile set-up (row table, bias, pointers)
    if (loader && !XR) issue(0, 0, 0, 0);
    int cur = 0;
    // Dropout decisions drawn INSIDE the main loop (row-reuse tower kernel, per-sample layers): the 16 Philox calls of a lane's
    // tile -- their counters are known before the first K-tile -- run one round at a time in the MFMA shadows of the first 8
    // (chunk, ky) groups and leave 128 keep bits (call q = j*8 + i*2 + p -> one byte); the epilogue only expands them.
    // Same generator, same counters, same threshold test: bit-identical to drawing them in the epilogue.
    uint32_t ph_bits[4] = {0u, 0u, 0u, 0u};
    bool ph_inloop = false;
    if constexpr (XR && ABL != 81) {
        // Row-reuse loop, second generation: (a) compact staging state -- one weight pointer plus scalar
        // row strides, 32-bit activation offsets against the group's base pointer, advanced incrementally
        // per (chunk, ky) group -- frees the registers for (b) a software-pipelined fragment schedule: per
        // k-step the A fragments are double-buffered in pairs (i = 0,1 | 2,3) and the B fragments across
        // k-steps, every ds_read_b128 is issued >= 4 MFMAs before its first use, and the only exposed LDS
        // wait is the first fragment set after each block barrier (covered by issuing the LDS-DMA there).
        static_assert(BC == 256 && BP == 256 && WC == 2 && WP == 4, "written for the 256x256 8-wave tile");
        static_assert(M16 || !(!SPLIT && FP == 2 && ABL != 81), "decisions are drawn in the loop by the 16x16x32 builds only");
        constexpr int WST = BC * ROWB, XBUF = XR_EXT_ROWS * ROWB, NXE = XR_EXT_ROWS * 8 / THREADS;   // 5 pieces / thread
        // 32-bit offsets are taken against the tile's FIRST extended row (the smallest pixel index of the tile: the
        // host builds the list in increasing order and pads it with that row), so the activation buffer may be of any size
        const int ext_first = __builtin_amdgcn_readfirstlane(a.ext[(size_t)bx * XR_EXT_ROWS].x);
        const char* in_base = reinterpret_cast<const char*>(G.in) + ((size_t)ext_first * a.in_cstride + G.in_coff) * 2;
        const char* wbase = wsrc[0];
        // M16: the LDS-DMA pieces of the loop are buffer_load_dwordx4 ... lds, not global_load_lds_dwordx4: a buffer resource in SGPRs
        // (weights: one for the tile + a scalar byte offset per piece; extended rows: the tile's first row), the lane's 32-bit byte
        // offset in one VGPR.  Same bytes, same landing -- but the issue parks the wave for fewer cycles: tests/tools/loop_anatomy.hip,
        // 3404 -> 3111 cycles per K-tile (profiles/round2_loop_anatomy.txt)
        const char* wuni;
        {
            const uint64_t wu = (uint64_t)(uintptr_t)(reinterpret_cast<const char*>(G.w) + ((size_t)bc0 * wrow + c_begin * BK) * 2);
            wuni = reinterpret_cast<const char*>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wu >> 32)) << 32) |
                                                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wu)));
        }
        const uint32_t wlane = (uint32_t)((ldrow * wrow + ldchunk * 8) * 2);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wuni), 0, -1, 0x00020000);      // raw, unbounded
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in_base), 0, -1, 0x00020000);
        const int wrs = RPI * wrow * 2;                                     // bytes between the rows of two weight pieces
        uint32_t xo[NXE];
        int xp[NXE];
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            const int2 e = a.ext[(size_t)bx * XR_EXT_ROWS + i * (THREADS / 8) + (tid >> 3)];
            // M16: the extended rows are chunk-ROTATED -- physical chunk = (chunk + (row & 6)) & 7 -- not XOR-swizzled: a B fragment's 16
            // rows start at ANY extended row, and the rotation keeps every ds_read_b128 lane group (16 lanes = rows {0-3,12-15} at chunk
            // c and rows {4-11} at c+1, or the reverse) on 16 distinct 4-bank slots for every start row; the XOR form only does for
            // starts that are multiples of 16 (20 % of the LDS cycles of the loop were bank conflicts, profiles/round2_head_conv_counters.json)
            const int xchunk = M16 ? ((ltid & 7) - ((ltid >> 3) & 6)) & 7 : ldchunk;
            xo[i] = ((uint32_t)(e.x - ext_first) * (uint32_t)a.in_cstride + xchunk * 8) * 2u;
            xp[i] = e.y * a.in_cstride * 2;
        }
        int xrow[FP];
#pragma unroll
        for (int j = 0; j < FP; ++j) xrow[j] = a.rows[bp0 + wp * WTP + j * 32 + frow].pad1;
        int xrow16[FP16];                                                    // M16: extended-row index of the lane's pixel in each 16-pixel fragment
#pragma unroll
        for (int j = 0; j < FP16; ++j) xrow16[j] = M16 ? a.rows[bp0 + wp * WTP + j * 16 + l15].pad1 : 0;
        auto issue_wx = [&](int stage, int ky_, int kx_, int cc_) {
            const int woff = ((ky_ * 3 + kx_) * a.cin + cc_ * BK) * 2;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                int off = woff + i * wrs;
                asm volatile("" : "+s"(off));                               // keep the sum scalar, do not hoist 4 pointers
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wbase + off), LDS_PTR(smem + stage * WST + (i * THREADS + wave * 64) * 16), 16, 0, 0);
            }
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        (void)issue_wx;                                                      // (K-tile 0's weights went out at the top of the tile: EARLY_W)
#pragma unroll
        for (int i = 0; i < NXE; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(in_base + xo[i]), LDS_PTR(smem + 2 * WST + (i * THREADS + wave * 64) * 16), 16, 0, 0);
        const int a_c0 = (fhalf ^ fswz) << 4;                                // chunk byte offset of k-step 0 (k-step ks: ^ (ks << 5))
        const int a_row = (wc * WTC + frow) * ROWB + a_c0;
        const int NG = KT / 3;
        constexpr bool PH_BUILD = !SPLIT && FP == 2 && (ABL == 0 || ABL == 9 || ABL == 90 || ABL == 30 || ABL == 31 || ABL == 1 || ABL == 2);
        const bool ph_on = PH_BUILD && (a.flags & CONV_DROPOUT) && !(a.flags & CONV_OUT_F32) && a.fan_count <= 1 && NG >= 8 && bc0 == 0 &&
                           a.drop_threshold >= 1 && a.variant != 83;            // (variant 83: decisions drawn in the epilogue, A/B)
        ph_inloop = ph_on;
        uint32_t ph_k0 = a.seed_lo, ph_k1 = a.seed_hi, ph_img = a.image_base;
        PhiloxState ph{0u, 0u, 0u, 0u, 0u, 0u};
        if (ph_on) {
            if (a.dyn_rng) { ph_k0 = a.dyn_rng[0]; ph_k1 = a.dyn_rng[1]; ph_img = a.dyn_rng[2]; }
        }
        const uint32_t ph_thr = a.drop_threshold;
        auto ph_compress = [&]() -> uint32_t {                               // 16 keep bits: bit 4u + r = channel r of run u (contract v3)
            const Philox4 rr{ph.c0, ph.c1, ph.c2, ph.c3};
            uint32_t b = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const DropPair w = dropout_run_windows(rr, u);
                b |= (((w.x & 0xFFFFu) >= ph_thr) ? 1u : 0u) << (4 * u) | (((w.x >> 16) >= ph_thr) ? 1u : 0u) << (4 * u + 1) |
                     (((w.y & 0xFFFFu) >= ph_thr) ? 1u : 0u) << (4 * u + 2) | (((w.y >> 16) >= ph_thr) ? 1u : 0u) << (4 * u + 3);
            }
            return b;
        };
        // ------------------------------------------------------------------------------------------------------------------
        // The tower loop with the K-tile barrier TWO FRAGMENT STEPS BEFORE THE K-TILE'S END (round 3; every M16 build but ABL 9).  A K-tile is 16
        // steps (k-step ks = step >> 3, cout fragment fc = step & 7) of four MFMAs.  The barrier in front of step 14 comes after
        // every read of this K-tile's weight stage has returned (the A fragments of steps 14, 15 are fetched two steps ahead), so
        // behind it the NEXT K-tile's first fragments are read -- A(0), A(1), and the B set behind step 15's MFMAs -- while
        // steps 14 and 15 still multiply: the LDS round trip and the barrier skew that opened every K-tile with the matrix pipe
        // idle (tests/tools/loop_anatomy.hip: 3111 -> 2714 cycles per K-tile) disappear behind eight MFMAs.  The compiled forms of
        // round 2 lost that gain to `s_waitcnt lgkmcnt(0)` instructions the compiler put behind the prefetch (it cannot count
        // loads it mixes with hand-written waits).  Here EVERY LDS read of the loop is inline asm and every wait is written out
        // with its exact count -- LDS returns in order, so `lgkmcnt(n)` in front of an MFMA leaves exactly the n younger reads in
        // flight -- the compiler sees no LDS access at all and adds no wait of its own.  bayes_od_rc_amd/kernel_guard.py checks
        // the disassembly: no instruction may touch a fragment register between its ds_read and the wait that covers it.
        // DMA schedule: a stage is free from the barrier in front of step 14 of the K-tile that read it, so the weight pieces of
        // K-tile t+1 go out behind steps 15 (of t-1), 1, 3, 5 (of t) and are waited for in front of step 14 of t; the next
        // group's extended rows behind steps 7, 9, 11 of the group's first K-tile and 7, 9 of its second.
        // ------------------------------------------------------------------------------------------------------------------
        constexpr bool MB = M16 && ABL != 9;                  // every build of the 16x16x32 loop but ABL 9 (the round-2 loop, top-of-K-tile barrier: A/B)
        [[maybe_unused]] constexpr int AHEAD = 2;             // A fragments fetched this many steps ahead (ring of four; three ahead measured 0.3-0.5 % slower)
        if constexpr (MB) {
#if defined(__HIP_DEVICE_COMPILE__)
            auto lgkm = [](int n) {
                switch (n) {
                    case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); break;
                    case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); break;
                }
            };
            // A fragment fc of the K-tile whose stage / k-step is folded into `addr`: immediate offset fc * 16 rows
            auto rdA = [](bf16x8& d, uint32_t addr, int fc) {
                switch (fc) {
                    case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); break;
                    case 1: asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(d) : "v"(addr)); break;
                    case 2: asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(d) : "v"(addr)); break;
                    case 3: asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(d) : "v"(addr)); break;
                    case 4: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(d) : "v"(addr)); break;
                    case 5: asm volatile("ds_read_b128 %0, %1 offset:10240" : "=v"(d) : "v"(addr)); break;
                    case 6: asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(d) : "v"(addr)); break;
                    default: asm volatile("ds_read_b128 %0, %1 offset:14336" : "=v"(d) : "v"(addr)); break;
                }
            };
            auto rdB = [](bf16x8& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); };
            static_assert(ROWB * 16 == 2048 && WST == 32768, "immediate offsets of rdA");
            const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
            // A address of fragment 0, k-step 0 in the stage of the CURRENT K-tile (k-step 1: ^ 64; other stage: ^ WST)
            uint32_t a_cur = lds0 + (uint32_t)((wc * WTC + l15) * ROWB + ((q4 ^ ((l15 >> 1) & 7)) << 4));
            auto b_addr = [&](int xbuf, int j, int kxc_) -> uint32_t {
                const int r = xrow16[j] + kxc_;
                return lds0 + (uint32_t)(2 * WST + xbuf * XBUF + r * ROWB + (((q4 + (r & 6)) & 7) << 4));
            };
            // fragments: A in a ring of four (three steps ahead; 16 % 4 == 0, the slots repeat per K-tile), B in TWO sets -- set ks holds
            // the fragments of k-step ks; the set not in use is refilled half a K-tile ahead: k-step 1's fragments behind steps 2..5,
            // the next K-tile's k-step-0 fragments behind steps 9..12 (same staged rows, one further: resident since the group's
            // first barrier) or, at a group's last K-tile, behind the barrier (steps 14, 15: the next group's rows)
            bf16x8 Ar[4], Bc[2][4];
            // uniform inputs of the in-loop Philox set-up, pinned in scalar registers: a kernel-argument load inside the loop makes the
            // compiler wait lgkmcnt(0) behind it, which also drains the fragment reads in flight
            uint32_t ph_lid16 = (uint32_t)G.layer_id << 16, ph_sbase = a.sample_base;
            asm volatile("" : "+s"(ph_lid16), "+s"(ph_sbase));
            // weight offsets of the K-tiles one and two ahead: K-tile t is tap t % 9 of channel chunk t / 9
            int wo1 = a.cin * 2, tap1 = 1, wo2 = 2 * a.cin * 2, tap2 = 2;
            auto w_advance = [&](int& wo, int& tap) { if (++tap == 9) { tap = 0; wo += BK * 2 - 8 * a.cin * 2; } else wo += a.cin * 2; };
            // ConvArgs.mx_loader == 1 (round 5, A/B switch BOD_TOWER_LOADER): the lower four waves -- one per SIMD -- issue the weight pieces of
            // both waves of their SIMD (their own 8 rows of a piece and the rows + 32 of wave + 4), the upper four none (f16mx loop: -0.9 %)
            const bool w_pair = a.mx_loader == 1;
            auto dma_w = [&](int piece, int wo, int stage_) {
                int so_ = wo + piece * wrs;
                asm volatile("" : "+s"(so_));
                if (!w_pair) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(smem + stage_ * WST + (piece * THREADS + wave * 64) * 16), 16, (int)wlane, so_, 0, 0);
                    return;
                }
                if (wave >= 4) return;
                int so2 = so_ + 32 * wrow * 2;
                asm volatile("" : "+s"(so2));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(smem + stage_ * WST + (piece * THREADS + wave * 64) * 16), 16, (int)wlane, so_, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(smem + stage_ * WST + (piece * THREADS + (wave + 4) * 64) * 16), 16, (int)wlane, so2, 0, 0);
            };
            // prologue: K-tile 0's weights and group 0's rows are on their way (issued above); add weight piece 0 of K-tile 1, then
            // wait for everything but that piece
            if (KT > 1) dma_w(0, wo1, 1);
            if (KT > 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
            // (the order a group's last K-tile leaves behind: A(0), A(1), B0, B1, A(2), B2, B3)
            rdA(Ar[0], a_cur, 0);
            if constexpr (AHEAD == 3) rdA(Ar[1], a_cur, 1);
            rdB(Bc[0][0], b_addr(0, 0, 0)); rdB(Bc[0][1], b_addr(0, 1, 0));
            rdA(Ar[AHEAD - 1], a_cur, AHEAD - 1); rdB(Bc[0][2], b_addr(0, 2, 0)); rdB(Bc[0][3], b_addr(0, 3, 0));
            __builtin_amdgcn_sched_barrier(0);
            for (int g = 0; g < NG; ++g) {
                const bool xnext = g + 1 < NG;
                const bool next_row = ky + 1 < 3;
                const int xdst = 2 * WST + ((g + 1) & 1) * XBUF;
                const bool ph_act = ph_on && g < 8;
                // group g < 8 draws ONE call in slots 0..9 (contract v3: 16 decisions per call): pixel fragment fp = g >> 1, cout fragment
                // pair tt = (g & 1) * 2 + (q4 >> 1) -- the lane pair (l, l ^ 32) shares the calls of a fragment pair's two halves: the lower
                // lane half draws the even pairs, the upper half the odd ones, and they swap their 128 keep bits after the loop
                auto ph_slot = [&](int sidx) {
                    if (!PH_BUILD || !ph_act || sidx > 9) return;
                    if (sidx == 0) {
                        long long rr;
                        const uint32_t ra = (uint32_t)(uintptr_t)LDS_PTR(&s_rng[wp * WTP + (g >> 1) * 16 + l15]);
                        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rr) : "v"(ra) : "memory");
                        ph.c1 = (uint32_t)((wc * 4 + (g & 1) * 2 + (q4 >> 1)) * 2 + (q4 & 1));     // dropout_group16(wc*128 + tt*32 + (q4&1)*4)
                        ph.c0 = (uint32_t)rr;
                        const uint32_t ry = (uint32_t)((unsigned long long)rr >> 32);
                        ph.c2 = (ph_sbase + (ry & 0xFFFFu)) | ph_lid16;
                        ph.c3 = ph_img + (ry >> 16);
                        ph.k0 = ph_k0; ph.k1 = ph_k1;
                        philox_rounds(ph, 1);
                        return;
                    }
                    philox_rounds(ph, 1);
                    if (sidx == 9) {
                        const uint32_t n16 = ph_compress();
                        ph_bits[3] = (ph_bits[3] << 16) | (ph_bits[2] >> 16);
                        ph_bits[2] = (ph_bits[2] << 16) | (ph_bits[1] >> 16);
                        ph_bits[1] = (ph_bits[1] << 16) | (ph_bits[0] >> 16);
                        ph_bits[0] = (ph_bits[0] << 16) | n16;
                    }
                };
#pragma unroll
                for (int kxc = 0; kxc < 3; ++kxc) {
                    const int kt = g * 3 + kxc;
                    const bool w1 = kt + 1 < KT, w2 = kt + 2 < KT;
                    // B fragments: k-step 1 of this K-tile; k-step 0 of the next (same rows one further, or the next group's rows)
                    uint32_t xb1[4], xbn[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        xb1[j] = b_addr(g & 1, j, kxc) ^ 64u;
                        xbn[j] = kxc < 2 ? b_addr(g & 1, j, kxc + 1) : b_addr((g + 1) & 1, j, 0);
                    }
                    const uint32_t a_k1 = a_cur ^ 64u, a_nxt = a_cur ^ (uint32_t)WST;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int st = 0; st < 16; ++st) {
                        const int fc = st & 7, ks = st >> 3;
                        if (st == 14) {
                            // the next K-tile's weights (and, behind a group's last K-tile, the next group's rows) have landed; my reads of
                            // this K-tile's stage have returned
                            if (xnext && kxc == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                            else if (xnext && kxc == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            lgkm(0);
                            __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
                        }
                        // ---- this step's reads, in the order the wait table below assumes
                        // A fragments THREE steps ahead (ring of four): the last one of the K-tile, A(15), goes out in step 12, so the
                        // lgkmcnt(0) in front of the barrier finds it landed; the next K-tile's A(0), A(1) follow the barrier, A(2) step 15
                        if (st + AHEAD <= 15) rdA(Ar[(st + AHEAD) & 3], (st + AHEAD) < 8 ? a_cur : a_k1, (st + AHEAD) & 7);
                        else if (st == 14) { rdA(Ar[0], a_nxt, 0); if constexpr (AHEAD == 3) rdA(Ar[1], a_nxt, 1); }
                        else if (st == 15) rdA(Ar[AHEAD - 1], a_nxt, AHEAD - 1);
                        if (st >= 2 && st <= 5) rdB(Bc[1][st - 2], xb1[st - 2]);
                        if (kxc < 2 && st >= 9 && st <= 12) rdB(Bc[0][st - 9], xbn[st - 9]);
                        if (kxc == 2 && st >= 14) { rdB(Bc[0][2 * (st - 14)], xbn[2 * (st - 14)]); rdB(Bc[0][2 * (st - 14) + 1], xbn[2 * (st - 14) + 1]); }
                        // ---- reads younger than the operands of this step's MFMAs (LDS returns in order).  Step 0: A(0) and the B set were
                        // issued by the previous K-tile -- behind its barrier if that was a group's last one (kxc == 0 here): then only
                        // A(AHEAD) is younger than the last B fragment.
                        {
                            // (tables: tests/tools/lds_wait_tables.py replays the issue order below and counts)
                            constexpr int W_SAME3[16] = {3, 3, 4, 5, 6, 7, 6, 5, 3, 4, 5, 6, 7, 5, -1, -1};     // kxc < 2: next B set read in steps 9..12
                            constexpr int W_LAST3[16] = {3, 3, 4, 5, 6, 7, 6, 5, 3, 3, 3, 3, 3, 2, -1, -1};     // kxc == 2: next B set read behind the barrier
                            constexpr int W_SAME2[16] = {2, 2, 3, 4, 5, 5, 4, 3, 2, 3, 4, 5, 5, 4, -1, -1};
                            constexpr int W_LAST2[16] = {2, 2, 3, 4, 5, 5, 4, 3, 2, 2, 2, 2, 2, 2, -1, -1};
                            int n = AHEAD == 3 ? (kxc < 2 ? W_SAME3[st] : W_LAST3[st]) : (kxc < 2 ? W_SAME2[st] : W_LAST2[st]);
                            if (st == 0 && kxc == 0) n = 1;
                            if (n >= 0) lgkm(n);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) mfma16_inplace(acc4[fc][j], Ar[st & 3], Bc[ks][j]);
                        if (st & 1) {                        // 8 slots per K-tile, behind steps 1, 3, .. 15
                            const int slot = st >> 1;
                            __builtin_amdgcn_sched_barrier(0);
                            // (two pieces right behind the barrier and the others a slot earlier: no difference, A/B on one box)
                            if (ABL == 2) {}                     // (ablation: no staging after the first K-tile)
                            else if (slot < 3) { if (w1) dma_w(slot + 1, wo1, (kt + 1) & 1); }
                            else if (slot == 7) { if (w2) dma_w(0, wo2, kt & 1); }
                            else if ((kxc == 0 && slot < 6) || (kxc == 1 && slot < 5)) {
                                const int I = kxc == 0 ? slot - 3 : slot;          // pieces 0..2 in the group's first K-tile, 3..4 in its second
                                if (xnext) {
                                    xo[I] += next_row ? (uint32_t)xp[I] : (uint32_t)(BK * 2 - 2 * xp[I]);
                                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, LDS_PTR(smem + xdst + (I * THREADS + wave * 64) * 16), 16, (int)xo[I], 0, 0, 0);
                                }
                            }
                            if (slot < 7) ph_slot(kxc * 7 + slot);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    a_cur = a_nxt;
                    w_advance(wo1, tap1); w_advance(wo2, tap2);
                }
                if (++ky == 3) { ky = 0; ++cc; }
            }
            // the reads issued for a K-tile that does not exist: their registers stay allocated until they have landed
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Ar[0]), "+v"(Ar[1]), "+v"(Ar[2]), "+v"(Ar[3]), "+v"(Bc[0][0]), "+v"(Bc[0][1]), "+v"(Bc[0][2]), "+v"(Bc[0][3]),
                         "+v"(Bc[1][0]), "+v"(Bc[1][1]), "+v"(Bc[1][2]), "+v"(Bc[1][3]) :: "memory");
#endif
        } else if constexpr (MXL) {
            // ---- f16mx: the K-tile flavour is fixed per loop -- chunk cc = 2p is the H chunk of 64-channel group p (f16 hi: four f16
            // k-steps), cc = 2p + 1 its X chunk (two block-scaled e2m3 products): three groups (ky) of one flavour, then three of the other, as
            // two loops in sequence (one body with a flavour branch per K-tile makes the register allocator spill accumulators: 792 bytes
            // per lane).  Barrier inside the K-tile, first operands carried across its boundary, LDS-DMA pieces in the slots of the MFMA
            // stream: hx_ktile.  Schedule of the pieces (buffer_load_dwordx4 ... lds: resource in SGPRs, one 32-bit lane offset, scalar piece
            // offset -- the issue parks the wave for fewer cycles than global_load_lds_dwordx4): the stage K-tile kt read is free behind
            // ITS barrier, so weight piece 0 of K-tile kt + 2 goes out in slot 7 of kt and pieces 1..3 in slots 0..2 of kt + 1; the next
            // group's extended rows in slots 3..5 -- pieces 0..2 in the group's first K-tile, 3..4 in its second (HBM / Infinity Cache: a
            // microsecond under load).  The wait in front of the barrier of kt needs the weights of kt + 1, not the extended rows issued
            // behind them in this K-tile: loads retire in order, vmcnt(3) / vmcnt(2) leaves exactly those outstanding.
            auto woff_of = [&](const int kt) {          // K-tile kt = tap kt % 9 of chunk kt / 9 (chunk outer, ky, kx inner)
                const int ch = kt / 9, tap = kt - ch * 9;
                return (tap * a.cin + ch * BK) * 2;
            };
            // Who issues the weight pieces (ConvArgs.mx_loader): 0 = every wave its own eighth of a piece (8 rows); 1 / 2 = the lower / upper four
            // waves -- one per SIMD -- issue their own rows AND those of the wave they share the SIMD with (rows + 32: 16 KiB further in the
            // source, 4 KiB in LDS), the other four none.  The two waves of a SIMD share one matrix pipe and a wave parks 60-185 cycles per
            // piece it issues: with every wave issuing, one wave of each SIMD reaches the K-tile's barrier ~620 cycles before the other
            // (phase clock: `s_barrier` 44 k of 223 k cycles per tile) and idles there while its partner's parks leave the pipe empty.
            // (round 6, measured and not kept: the four late waves issuing their LDS-DMA piece IN FRONT of a step's MFMAs instead of behind
            // them, so that the two waves of a SIMD alternate between parking on a piece and issuing MFMAs -- 81.5 against 78.9 ms per
            // layer-1 launch on one box, profiles/round6_mx_ablations.txt)
            const int mx_loader = a.mx_loader & 3;
            const bool w_issuer = mx_loader == 0 || (mx_loader == 1 ? wave < 4 : wave >= 4);
            auto dma_w = [&](const int piece, const int kt_, char* __restrict__ wr) {
                if (!w_issuer) return;
                int off = woff_of(kt_) + piece * wrs;
                asm volatile("" : "+s"(off));
                if (mx_loader == 0) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(wr + (kt_ & 1) * WST + (piece * THREADS + wave * 64) * 16), 16, (int)wlane, off, 0, 0);
                    return;
                }
                // (lane offsets of the lower wave of the pair: wave & 3; this wave's own rows are + 0 or + 32 accordingly)
                const int w3 = wave & 3;
                const int lane_off = (int)wlane - (wave >= 4 ? 32 * wrow * 2 : 0);
                int off2 = off + 32 * wrow * 2;
                asm volatile("" : "+s"(off2));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(wr + (kt_ & 1) * WST + (piece * THREADS + w3 * 64) * 16), 16, lane_off, off, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(wr + (kt_ & 1) * WST + (piece * THREADS + (w3 + 4) * 64) * 16), 16, lane_off, off2, 0, 0);
            };
            // f16mx4: the scale bytes of an X K-tile's weights (chunk 6 of the tap's 1 024 bytes: [x][half][ks]) and of an X group's extended
            // rows (chunk 6 of the row) travel as 4-byte LDS-DMA pieces -- one lane per (row, half): [row][half] dwords in LDS, the dword's
            // byte ks = the block's E8M0 scale (op_sel of the MFMA).  Weights: with weight piece 1 (older than the extended rows' pieces:
            // the wait in front of the barrier covers it); rows: in slot 6 of the group's first K-tile, behind the rows' first three pieces.
            constexpr int OFF_WS = Cfg::LDS, OFF_XS = OFF_WS + 2 * 2048, XSB = XR_EXT_ROWS * 8;
            uint32_t wslane = 0, xso[2] = {0u, 0u};
            int xsp[2] = {0, 0};
            if constexpr (MX4) {
                wslane = (uint32_t)(tid * 16);
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) {
                    const int r = p_ * 256 + (tid >> 1);
                    const int2 e = a.ext[(size_t)bx * XR_EXT_ROWS + (r < XR_EXT_ROWS ? r : 0)];
                    xso[p_] = ((uint32_t)(e.x - ext_first) * (uint32_t)a.in_cstride) * 2u + 768u + (uint32_t)(tid & 1) * 4u;
                    xsp[p_] = e.y * a.in_cstride * 2;
                }
            }
            auto dma_ws = [&](const int kt_, char* __restrict__ wr) {          // the scales of X K-tile kt_ (chunk 2 or 5 of its tap)
                // (a compact copy behind the 256 weight rows: [tap][x][cout][half][ks], 2 KiB per X K-tile = sixteen 128-byte lines; the
                //  same bytes out of the rows' own scale chunks are 256 lines -- as many L2 requests as the K-tile's weights)
                if (wave >= 2) return;
                const int ch = kt_ / 9, tap = kt_ - ch * 9;
                int off = 256 * wrow * 2 + (tap * 2 + (ch >= 3 ? 1 : 0)) * 2048;
                asm volatile("" : "+s"(off));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(wr + OFF_WS + (kt_ & 1) * 2048 + wave * 1024), 16, (int)wslane, off, 0, 0);
            };
            auto dma_xs = [&](const int g_, const int ky_, const int x_, char* __restrict__ wr) {      // the scales of X group g_'s extended rows
                const int v0 = (int)xso[0] + ky_ * xsp[0] + x_ * 8;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, LDS_PTR(wr + OFF_XS + (g_ & 1) * XSB + wave * 256), 4, v0, 0, 0, 0);
                if (wave < (XR_EXT_ROWS - 256) / 32) {
                    const int v1 = (int)xso[1] + ky_ * xsp[1] + x_ * 8;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, LDS_PTR(wr + OFF_XS + (g_ & 1) * XSB + 2048 + wave * 256), 4, v1, 0, 0, 0);
                }
            };
            HxCarry<FP> carry;
            // (round 6, measured and not kept: s_setprio 1 for the four late waves -- which reach every K-tile's barrier last -- 73.3-73.6 against
            // 72.2-72.4 ms per layer-1 launch; for the four early ones 72.4-72.5: profiles/round6_mx_ablations.txt)
            // prologue: K-tile 0's weights and group 0's rows are on their way (issued above); piece 0 of K-tile 1 behind them
            if (KT > 1 && MXK != 12) dma_w(0, 1, smem);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
            {
                carry.hA = *reinterpret_cast<const bf16x8*>(smem + a_row);
#pragma unroll
                for (int j = 0; j < FP; ++j) { const int r = xrow[j]; carry.hB[j] = *reinterpret_cast<const bf16x8*>(smem + 2 * WST + r * ROWB + ((fhalf ^ ((r >> 1) & 7)) << 4)); }
            }
            auto group = [&](const int g, auto XT, auto LASTB) {
                constexpr bool xt = decltype(XT)::value;
                constexpr bool last_of_block = decltype(LASTB)::value;          // the next group has the other flavour
                constexpr bool xt_next_group = last_of_block ? !xt : xt;
                const bool xnext = g + 1 < NG && MXK != 12;
                const bool next_row = ky + 1 < 3;
                const int xdst = 2 * WST + ((g + 1) & 1) * XBUF;
                auto ktile = [&](auto KXC) {
                    constexpr int kxc = decltype(KXC)::value;
                    constexpr bool nxt = (kxc == 2 && last_of_block) ? !xt : xt;
                    const int kt = g * 3 + kxc;
                    const bool has_next = kt + 1 < KT;
                    // (the lane's row offsets re-derived behind an opaque barrier per K-tile: otherwise every (stage, buffer, tap, k-step)
                    // variant of the fragment addresses -- loop-invariant, some seventy registers -- is hoisted out of the loop and the
                    // kernel spills)
                    int a_row_ = a_row, xrow_[FP];
#pragma unroll
                    for (int j = 0; j < FP; ++j) xrow_[j] = xrow[j];
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : "+v"(a_row_));
#pragma unroll
                    for (int j = 0; j < FP; ++j) asm volatile("" : "+v"(xrow_[j]));
#endif
                    const int wa_ = (kt & 1) * WST + a_row_, wa_n = ((kt + 1) & 1) * WST + a_row_;
                    const int xbase = 2 * WST + (g & 1) * XBUF, xbase_n = kxc < 2 ? xbase : 2 * WST + ((g + 1) & 1) * XBUF;
                    int xb_[FP], xb_n[FP];
#pragma unroll
                    for (int j = 0; j < FP; ++j) {
                        const int r = xrow_[j] + kxc, rn = kxc < 2 ? r + 1 : xrow_[j];
                        xb_[j] = xbase + r * ROWB + ((fhalf ^ ((r >> 1) & 7)) << 4);
                        xb_n[j] = xbase_n + rn * ROWB + ((fhalf ^ ((rn >> 1) & 7)) << 4);
                    }
                    // f16mx4: scale dwords of the NEXT K-tile (weights: stage of kt + 1; rows: the group's array, the tap's row)
                    int ws_n = 0, xs_n[FP];
#pragma unroll
                    for (int j = 0; j < FP; ++j) xs_n[j] = 0;
                    if constexpr (MX4 && nxt) {
                        ws_n = OFF_WS + ((kt + 1) & 1) * 2048 + ((wc * WTC + frow) * 2 + fhalf) * 4;
#pragma unroll
                        for (int j = 0; j < FP; ++j) {
                            const int rn = kxc < 2 ? xrow_[j] + kxc + 1 : xrow_[j];
                            xs_n[j] = OFF_XS + ((kxc < 2 ? g : g + 1) & 1) * XSB + (rn * 2 + fhalf) * 4;
                        }
                    }
                    // the extended rows' scale pieces of the next group leave in this group's first K-tile (slot 6), when it is an X group
                    constexpr bool xs_here = MX4 && xt_next_group && kxc == 0;
                    unsigned long long mx_tb = 0, mx_tk = 0;
                    if constexpr (MXI) { mx_tb = __builtin_amdgcn_s_memtime(); mx_tk = mx_tb; }
                    hx_ktile<FC, FP, ROWB, xt, nxt, MX4>(acc, smem, smem, wa_, xb_, wa_n, xb_n, has_next, carry,
                        [&]() {
                            unsigned long long ta = 0;
                            if constexpr (MXI) ta = __builtin_amdgcn_s_memtime();
                            if (xs_here && xnext) {
                                if (wave < (XR_EXT_ROWS - 256) / 32) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
                                else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                            }
                            else if (xnext && kxc == 0) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
                            else if (xnext && kxc == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                            unsigned long long tm = 0;
                            if constexpr (MXI) tm = __builtin_amdgcn_s_memtime();
                            __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
                            if constexpr (MXI) { const unsigned long long tb = __builtin_amdgcn_s_memtime(); mx_t_wait += tb - ta; mx_t_cnt += tm - ta; mx_tb += tb - ta; }
                        },
                        [&](const int k, char* __restrict__ wr) {
                            if (MXK == 12) return;
                            if (k == 7) { if (kt + 2 < KT) dma_w(0, kt + 2, wr); }
                            else if (k < 3) {
                                if (kt + 1 < KT) { dma_w(k + 1, kt + 1, wr); if constexpr (MX4 && nxt) { if (k == 0) dma_ws(kt + 1, wr); } }
                            }
                            else if (xs_here && k == 6) { if (xnext) dma_xs(g + 1, next_row ? ky + 1 : 0, (next_row ? cc : cc + 1) >= 3 ? 1 : 0, wr); }
                            else if (xnext && k < 6 && kxc < 2) {
                                const int i = kxc == 0 ? k - 3 : k;          // pieces 0..2 | 3..4
                                if (i < NXE) {
                                    xo[i] += next_row ? (uint32_t)xp[i] : (uint32_t)(BK * 2 - 2 * xp[i]);
                                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, LDS_PTR(wr + xdst + (i * THREADS + wave * 64) * 16), 16, (int)xo[i], 0, 0, 0);
                                }
                            }
                        }, ws_n, xs_n);
                    if constexpr (MXI) {
                        const unsigned long long te = __builtin_amdgcn_s_memtime();
                        mx_t_body += te - mx_tb;
                        if (xt) mx_t_x += te - mx_tk; else mx_t_h += te - mx_tk;
                    }
                };
                ktile(std::integral_constant<int, 0>{}); ktile(std::integral_constant<int, 1>{}); ktile(std::integral_constant<int, 2>{});
                if (++ky == 3) { ky = 0; ++cc; }
            };
            if constexpr (MX4) {          // per 128 channels: two H chunks, then the X chunk of their cross terms
                for (int g0 = 0; g0 < NG; g0 += 9) {
                    group(g0, std::false_type{}, std::false_type{}); group(g0 + 1, std::false_type{}, std::false_type{}); group(g0 + 2, std::false_type{}, std::false_type{});
                    group(g0 + 3, std::false_type{}, std::false_type{}); group(g0 + 4, std::false_type{}, std::false_type{}); group(g0 + 5, std::false_type{}, std::true_type{});
                    group(g0 + 6, std::true_type{}, std::false_type{}); group(g0 + 7, std::true_type{}, std::false_type{}); group(g0 + 8, std::true_type{}, std::true_type{});
                }
            } else
            for (int g0 = 0; g0 < NG; g0 += 6) {
                group(g0, std::false_type{}, std::false_type{}); group(g0 + 1, std::false_type{}, std::false_type{}); group(g0 + 2, std::false_type{}, std::true_type{});
                group(g0 + 3, std::true_type{}, std::false_type{}); group(g0 + 4, std::true_type{}, std::false_type{}); group(g0 + 5, std::true_type{}, std::true_type{});
            }
            if constexpr (MXI) {
                // (round 6: whole K-tiles by flavour, wave 0 -- an early wave, which issues the weight pieces -- and wave 4, its late partner on the SIMD)
                if (threadIdx.x == 256) { atomicAdd(&g_phase_cycles[2], mx_t_h); atomicAdd(&g_phase_cycles[3], mx_t_x); atomicAdd(&g_phase_cycles[4], mx_t_wait); }
                if (threadIdx.x == 0) {
                    const unsigned long long now = __builtin_amdgcn_s_memtime();
                    atomicAdd(&g_phase_cycles[0], mx_t_h); atomicAdd(&g_phase_cycles[1], mx_t_x);
                    atomicAdd(&g_phase_cycles[6], mx_t_wait); atomicAdd(&g_phase_cycles[7], mx_t_body); atomicAdd(&g_phase_cycles[13], mx_t_cnt);
                    atomicAdd(&g_phase_cycles[8], now - mx_t0); atomicAdd(&g_phase_cycles[15], 1ull);
                    mx_t0 = now;
                }
            }
        } else
        for (int g = 0; g < NG; ++g) {
            const bool xnext = g + 1 < NG;
            const bool next_row = ky + 1 < 3;
            const int xdst = 2 * WST + ((g + 1) & 1) * XBUF;
            // group g < 8 draws one call in slots 0..9 (see the mid-tile-barrier loop above)
            const bool ph_act = ph_on && g < 8;
            auto ph_slot = [&](int sidx) {
                if (!PH_BUILD || !ph_act || sidx > 9) return;
                if (sidx == 0) {
                    int2 r = make_int2(0, 0);
                    if constexpr (M16) {
                        // read by hand: the compiler puts an s_waitcnt vmcnt(0) in front of a plain LDS load that follows an LDS-DMA issue
                        // (it cannot tell the table from the staging buffers) -- a full L2 round trip with the matrix pipe idle
                        long long rr;
                        const uint32_t ra = (uint32_t)(uintptr_t)LDS_PTR(&s_rng[wp * WTP + (g >> 1) * 16 + l15]);
                        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rr) : "v"(ra) : "memory");
                        r.x = (int)(uint32_t)rr; r.y = (int)(uint32_t)((unsigned long long)rr >> 32);
                        ph.c1 = (uint32_t)((wc * 4 + (g & 1) * 2 + (q4 >> 1)) * 2 + (q4 & 1));     // dropout_group16(wc*128 + tt*32 + (q4&1)*4)
                    }
                    ph.c0 = (uint32_t)r.x;
                    ph.c2 = (a.sample_base + ((uint32_t)r.y & 0xFFFFu)) | ((uint32_t)G.layer_id << 16);
                    ph.c3 = ph_img + ((uint32_t)r.y >> 16);
                    ph.k0 = ph_k0; ph.k1 = ph_k1;
                    philox_rounds(ph, 1);
                    return;
                }
                philox_rounds(ph, 1);
                if (sidx == 9) {
                    const uint32_t n16 = ph_compress();
                    ph_bits[3] = (ph_bits[3] << 16) | (ph_bits[2] >> 16);
                    ph_bits[2] = (ph_bits[2] << 16) | (ph_bits[1] >> 16);
                    ph_bits[1] = (ph_bits[1] << 16) | (ph_bits[0] >> 16);
                    ph_bits[0] = (ph_bits[0] << 16) | n16;
                }
            };
            bf16x8 Bc16[4];          // M16: the B fragments live across the K-tiles of a group (see the body below)
#pragma unroll
            for (int kxc = 0; kxc < 3; ++kxc) {
                const int kt = g * 3 + kxc;
                // This K-tile needs its weights and, in a group's first K-tile, the group's extended rows -- NOT the extended rows of
                // the next group that went out behind the weights during the previous K-tile (HBM / Infinity Cache: they land a
                // microsecond later).  Loads retire in order, so the wait leaves exactly those pieces outstanding: the three issued in
                // the group's first K-tile (kxc 0 -> 1) or the two of its second (kxc 1 -> 2; the first three are older than that
                // K-tile's weights and long landed).
                if (M16 && ABL != 2 && xnext && kxc == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (M16 && ABL != 2 && xnext && kxc == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // (bare s_barrier: __syncthreads() is a fence, before which the compiler waits for every outstanding load.  What the
                // barrier orders here is LDS-DMA landings, waited for just above, against LDS reads whose data the MFMAs have consumed.)
                if constexpr (M16) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } else __syncthreads();
                const bool wnext = kt + 1 < KT && ABL != 2;
                // weight tile kt+1 = tap (ky, kxc+1) of this chunk, or tap (ky+1 | 0, 0) of the next group's chunk
                int woff;
                if (kxc < 2) woff = ((ky * 3 + kxc + 1) * a.cin + cc * BK) * 2;
                else woff = next_row ? (((ky + 1) * 3) * a.cin + cc * BK) * 2 : ((cc + 1) * BK) * 2;
                const int wdst = ((kt + 1) & 1) * WST;
                KTilePipe<FC, FP, ROWB> pipe;
                pipe.smem = smem;
                pipe.wa = (kt & 1) * WST + a_row;
                const int xbase = 2 * WST + (g & 1) * XBUF;
                if constexpr (M16) {
                    // ---- 16x16x32 body: two k-steps of 32, per k-step 8 A fragments x 4 B fragments = 32 MFMAs.  A fragments in a
                    // ring of four (loaded two cout fragments ahead), the next k-step's B fragments while the first four cout
                    // fragments multiply; one ds_read_b128 slotted per MFMA or two; DMA pieces and Philox rounds after every
                    // second cout fragment.
                    const int wa16 = (kt & 1) * WST + (wc * WTC + l15) * ROWB + ((q4 ^ ((l15 >> 1) & 7)) << 4);
                    int xb16[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int r = xrow16[j] + kxc; xb16[j] = xbase + r * ROWB + (((q4 + (r & 6)) & 7) << 4); }
                    auto ldA16 = [&](int fc, int ks) { return *reinterpret_cast<const bf16x8*>(smem + ((wa16 ^ (ks << 6)) + fc * 16 * ROWB)); };
                    auto ldB16 = [&](int fp, int ks) { return *reinterpret_cast<const bf16x8*>(smem + (xb16[fp] ^ (ks << 6))); };
                    bf16x8 Ar[3];                            // A ring of three (the fragment in use and the next two); ONE set of B fragments
                    bf16x8 (&Bc)[4] = Bc16;                  // (Bc16): the next k-step's replace the current one's one by one behind the last cout
                    // fragment's MFMAs -- also across the K-tiles of a group, whose extended rows are all in LDS since the group's first
                    // barrier: only the weights' fragments (and, in a group's first K-tile, the activations') wait behind the barrier
                    if (kxc == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) Bc[j] = ldB16(j, 0);
                    }
                    Ar[0] = ldA16(0, 0); Ar[1] = ldA16(1, 0);
                    __builtin_amdgcn_sched_barrier(0);
#define DMA_W(I) if (wnext) { int so_ = woff + (I) * wrs; asm volatile("" : "+s"(so_)); \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(smem + wdst + ((I) * THREADS + wave * 64) * 16), 16, (int)wlane, so_, 0, 0); }
#define DMA_X(I) if ((I) < NXE && xnext && ABL != 2) { xo[(I) < NXE ? (I) : 0] += next_row ? (uint32_t)xp[(I) < NXE ? (I) : 0] : (uint32_t)(BK * 2 - 2 * xp[(I) < NXE ? (I) : 0]); \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, LDS_PTR(smem + xdst + (((I) < NXE ? (I) : 0) * THREADS + wave * 64) * 16), 16, (int)xo[(I) < NXE ? (I) : 0], 0, 0, 0); }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                        for (int fc = 0; fc < 8; ++fc) {
                            int reads = 0;
                            const int seq = ks * 8 + fc;             // fragment sequence number 0..15 -> ring slot seq % 3
                            if (fc + 2 < 8) { Ar[(seq + 2) % 3] = ldA16(fc + 2, ks); ++reads; }
                            else if (ks == 0) { Ar[(seq + 2) % 3] = ldA16(fc + 2 - 8, 1); ++reads; }
                            if (ks == 0 && fc == 7) {
                                // last cout fragment of k-step 0: B fragment j is dead once its MFMA has issued -> its k-step-1 value follows
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    mfma16_inplace(acc4[fc][j], Ar[seq % 3], Bc[j]);
                                    Bc[j] = ldB16(j, 1);
                                }
                            } else if (ks == 1 && fc == 7 && kxc < 2) {
                                // last cout fragment of the K-tile: the next K-tile of the group reads the same staged rows one extended row further
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    mfma16_inplace(acc4[fc][j], Ar[seq % 3], Bc[j]);
                                    const int r = xrow16[j] + kxc + 1;
                                    Bc[j] = *reinterpret_cast<const bf16x8*>(smem + xbase + r * ROWB + (((q4 + (r & 6)) & 7) << 4));
                                }
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) mfma16_inplace(acc4[fc][j], Ar[seq % 3], Bc[j]);
                            }
                            (void)reads;
                            if ((fc & 1) && !(ks == 1 && fc == 7)) {                 // 7 slots per K-tile, after 8, 16, ... 56 MFMAs
                                const int slot = ks * 4 + (fc >> 1);
                                __builtin_amdgcn_sched_barrier(0);
                                // next K-tile's weights (L2-resident) first; the next GROUP's extended rows (HBM / Infinity Cache: a microsecond
                                // under load) as early in the group as possible -- pieces 0..2 in its first K-tile, 3..4 in the second -- so that
                                // every piece has more than a K-tile to land before the vmcnt(0) that precedes its first reader
                                if (slot == 0) { DMA_W(0) } else if (slot == 1) { DMA_W(1) } else if (slot == 2) { DMA_W(2) } else if (slot == 3) { DMA_W(3) }
                                else if (kxc == 0) { DMA_X(slot - 4) } else if (kxc == 1 && slot < 6) { DMA_X(slot - 1) }
                                ph_slot(kxc * 7 + slot);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
#undef DMA_W
#undef DMA_X
                    continue;
                }
#pragma unroll
                for (int j = 0; j < FP; ++j) { const int r = xrow[j] + kxc; pipe.xb[j] = xbase + r * ROWB + ((fhalf ^ ((r >> 1) & 7)) << 4); }
                pipe.first_loads();
                if constexpr (SPLIT) {
                    // bf16x3 on the row-reuse staging: the next tile's pieces go out under the first fragment loads, then the
                    // six k-steps of the (hi, lo) products (compiler-scheduled: three times the MFMAs per staged tile)
                    if (wnext) {
#pragma unroll
                        for (int i = 0; i < NW; ++i) {
                            int off = woff + i * wrs;
                            asm volatile("" : "+s"(off));
                            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wbase + off), LDS_PTR(smem + wdst + (i * THREADS + wave * 64) * 16), 16, 0, 0);
                        }
                    }
                    if (xnext) {
#pragma unroll
                        for (int i = 0; i < NXE; ++i)
                            if (i == 2 * kxc || i == 2 * kxc + 1) {
                                xo[i] += next_row ? (uint32_t)xp[i] : (uint32_t)(BK * 2 - 2 * xp[i]);
                                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(in_base + xo[i]), LDS_PTR(smem + xdst + (i * THREADS + wave * 64) * 16), 16, 0, 0);
                            }
                    }
                    pipe.run_split_lean(acc);
                    continue;
                }
                bf16x8 A23[2], Bn[2];
#define MFMA_ROW(I, AF) \
    acc[I][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF, pipe.Bc[0], acc[I][0], 0, 0, 0); \
    acc[I][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF, pipe.Bc[1], acc[I][1], 0, 0, 0);
#define DMA_W(I) if (wnext) { int off = woff + (I) * wrs; asm volatile("" : "+s"(off)); \
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wbase + off), LDS_PTR(smem + wdst + ((I) * THREADS + wave * 64) * 16), 16, 0, 0); }
#define DMA_X(I) if ((I) < NXE && xnext && ABL != 2) { xo[(I) < NXE ? (I) : 0] += next_row ? (uint32_t)xp[(I) < NXE ? (I) : 0] : (uint32_t)(BK * 2 - 2 * xp[(I) < NXE ? (I) : 0]); \
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(in_base + xo[(I) < NXE ? (I) : 0]), LDS_PTR(smem + xdst + (((I) < NXE ? (I) : 0) * THREADS + wave * 64) * 16), 16, 0, 0); }
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    A23[0] = pipe.ldA(2, ks); A23[1] = pipe.ldA(3, ks);
                    Bn[0] = pipe.ldB(0, ks + 1); Bn[1] = pipe.ldB(1, ks + 1);
                    MFMA_ROW(0, pipe.Ac[0]) MFMA_ROW(1, pipe.Ac[1])
                    SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0) { DMA_W(0) } else if (ks == 1) { DMA_W(2) } else { DMA_X(2 * kxc) }
                    ph_slot(kxc * 7 + ks * 2);
                    __builtin_amdgcn_sched_barrier(0);
                    pipe.Ac[0] = pipe.ldA(0, ks + 1); pipe.Ac[1] = pipe.ldA(1, ks + 1);
                    MFMA_ROW(2, A23[0]) MFMA_ROW(3, A23[1])
                    SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0) { DMA_W(1) } else if (ks == 1) { DMA_W(3) } else { DMA_X(2 * kxc + 1) }
                    ph_slot(kxc * 7 + ks * 2 + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    pipe.Bc[0] = Bn[0]; pipe.Bc[1] = Bn[1];
                }
                A23[0] = pipe.ldA(2, 3); A23[1] = pipe.ldA(3, 3);
                MFMA_ROW(0, pipe.Ac[0]) MFMA_ROW(1, pipe.Ac[1])
                SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(1); SGB_DSRD(1); SGB_MFMA(2);
                __builtin_amdgcn_sched_barrier(0);
                ph_slot(kxc * 7 + 6);
                __builtin_amdgcn_sched_barrier(0);
                MFMA_ROW(2, A23[0]) MFMA_ROW(3, A23[1])
                SGB_MFMA(4);
#undef MFMA_ROW
#undef DMA_W
#undef DMA_X
            }
            if (++ky == 3) { ky = 0; ++cc; }
        }
    } else if constexpr (XR) {
        // Activation row reuse: per (channel chunk, ky) the tile's extended rows are staged ONCE and the
        // three kx taps read them at row offsets 0/1/2; only the weights stream every K-tile.
        static_assert(BC == 256 && BP == 256 && WC * WP == 8, "row reuse is written for the 256x256 8-wave tile");
        constexpr int WST = BC * ROWB, XBUF = XR_EXT_ROWS * ROWB, NXE = XR_EXT_ROWS * 8 / THREADS;   // 5 pieces / thread
        const char* xe[NXE];
        int xep[NXE];
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            const int2 e = a.ext[(size_t)bx * XR_EXT_ROWS + i * (THREADS / 8) + (tid >> 3)];
            xe[i] = reinterpret_cast<const char*>(G.in) + ((size_t)e.x * a.in_cstride + G.in_coff + ldchunk * 8) * 2;
            xep[i] = e.y * a.in_cstride * 2;
        }
        int xrow[FP];
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            int m = bp0 + wp * WTP + j * 32 + frow;
            xrow[j] = a.rows[m].pad1;
        }
        auto issue_wx = [&](int stage, int ky_, int kx_, int cc_) {
            const int woff = ((ky_ * 3 + kx_) * a.cin + cc_ * BK) * 2;
#pragma unroll
            for (int i = 0; i < NW; ++i)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wsrc[i] + woff), LDS_PTR(smem + stage * WST + (i * THREADS + wave * 64) * 16), 16, 0, 0);
        };
        auto issue_xe = [&](int buf, int ky_, int cc_, int i0, int i1) {
#pragma unroll
            for (int i = 0; i < NXE; ++i)
                if (i >= i0 && i < i1)
                    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(xe[i] + ky_ * xep[i] + cc_ * (BK * 2)),
                                                     LDS_PTR(smem + 2 * WST + buf * XBUF + (i * THREADS + wave * 64) * 16), 16, 0, 0);
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the weight / activation pieces issued by the generic prologue
        __syncthreads();
        issue_wx(0, 0, 0, 0);
        issue_xe(0, 0, 0, 0, NXE);
        for (int kt = 0; kt < KT; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int g = cc * 3 + ky;
            int nkx = kx + 1, nky = ky, ncc = cc;
            if (nkx == 3) { nkx = 0; if (++nky == 3) { nky = 0; ++ncc; } }
            if (kt + 1 < KT && ABL != 2) {
                issue_wx((kt + 1) & 1, nky, nkx, ncc);
                if (g + 1 < KT / 3) {                            // next (chunk, ky) group, a third per K-tile
                    const int gky = ky + 1 < 3 ? ky + 1 : 0, gcc = ky + 1 < 3 ? cc : cc + 1;
                    if (kx == 0) issue_xe((g + 1) & 1, gky, gcc, 0, 2);
                    else if (kx == 1) issue_xe((g + 1) & 1, gky, gcc, 2, 4);
                    else issue_xe((g + 1) & 1, gky, gcc, 4, NXE);
                }
            }
            const char* wb = smem + (kt & 1) * WST + (wc * WTC + frow) * ROWB;
            const char* xbase = smem + 2 * WST + (g & 1) * XBUF;
            const char* xbj[FP];
            int xsw[FP];
#pragma unroll
            for (int j = 0; j < FP; ++j) { const int r = xrow[j] + kx; xbj[j] = xbase + r * ROWB; xsw[j] = (r >> 1) & 7; }
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int ch = ((ks * 2 + fhalf) ^ fswz) << 4;
                bf16x8 af[FC], bfr[FP];
#pragma unroll
                for (int i = 0; i < FC; ++i) af[i] = *reinterpret_cast<const bf16x8*>(wb + i * 32 * ROWB + ch);
#pragma unroll
                for (int j = 0; j < FP; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(xbj[j] + (((ks * 2 + fhalf) ^ xsw[j]) << 4));
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            kx = nkx; ky = nky; cc = ncc;
        }
    } else {
        const int a_c0 = (fhalf ^ fswz) << 4;                                // chunk byte offset of k-step 0
        for (int kt = 0; kt < KT; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const bool more = kt + 1 < KT;
            if (more) { if (++kx == a.KW) { kx = 0; if (++ky == KH) { ky = 0; ++cc; } } }
            KTilePipe<FC, FP, ROWB> pipe;
            pipe.smem = smem;
            pipe.wa = cur * STAGE + (wc * WTC + frow) * ROWB + a_c0;
#pragma unroll
            for (int j = 0; j < FP; ++j) pipe.xb[j] = cur * STAGE + W_BYTES + (wp * WTP + j * 32 + frow) * ROWB + a_c0;
            pipe.first_loads();
            if (more && ABL != 2 && loader) issue(cur ^ 1, ky, kx, cc);
            if constexpr (SPLIT) pipe.run_split(acc); else pipe.run(acc);
            cur ^= 1;
        }
    }

    if constexpr (M16 || MXL) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last in-place MFMAs have retired before any VALU reads an accumulator
#endif
    }
    phase_stamp<ABL>(tstamp, 1);            // main loop
    if (ABL == 1) {
#pragma unroll
        for (int i = 0; i < FC16; ++i)
#pragma unroll
            for (int j = 0; j < FP16; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::"v"(acc4[i][j]));
#endif
            }
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int j = 0; j < FP; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" ::"v"(acc[i][j]));
#endif
            }
        return;
    }

    if (CAN_SPLITK && nsplit > 1) {
        // ---- split-K: raw fp32 accumulators of this split; conv_splitk_reduce_kernel finishes the layer
        float* part = a.partial + (size_t)gz * a.M * a.cout_pad;
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int m = bp0 + wp * WTP + j * 32 + frow;
            if (m >= a.M) continue;
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int co = bc0 + wc * WTC + i * 32 + g4 * 8 + fhalf * 4;
                    *reinterpret_cast<float4*>(part + (size_t)m * a.cout_pad + co) =
                        make_float4(acc[i][j][g4 * 4 + 0], acc[i][j][g4 * 4 + 1], acc[i][j][g4 * 4 + 2], acc[i][j][g4 * 4 + 3]);
                }
        }
        return;
    }
    const bool relu = a.flags & CONV_RELU, drop = a.flags & CONV_DROPOUT, of32 = !XR && (a.flags & CONV_OUT_F32);   // (row-reuse launches: head towers, bf16 / pair outputs only)
    if (of32) {
        // ---- fp32 outputs.  Channel counts that are multiples of 4 (input gradients, weight gradients): 16-byte
        // accesses, and with CONV_ACCUM the four old values of a fragment row are loaded before the first store
        const bool accum = a.flags & CONV_ACCUM;
        if ((a.cout_valid & 3) == 0) {
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int m = bp0 + wp * WTP + j * 32 + frow;
                if (m >= a.M) continue;
                float* orow = reinterpret_cast<float*>(G.out) + (size_t)a.rows[m].out_off * a.out_cstride;
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    const int co0 = bc0 + wc * WTC + i * 32 + fhalf * 4;
                    float4 old[4];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        old[g4] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (accum && co0 + g4 * 8 < a.cout_valid) old[g4] = *reinterpret_cast<const float4*>(orow + co0 + g4 * 8);
                    }
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int co = co0 + g4 * 8;
                        if (co >= a.cout_valid) continue;
                        const float4 bv = *reinterpret_cast<const float4*>(G.bias + co);
                        float4 v = make_float4(acc[i][j][g4 * 4 + 0] + bv.x, acc[i][j][g4 * 4 + 1] + bv.y,
                                               acc[i][j][g4 * 4 + 2] + bv.z, acc[i][j][g4 * 4 + 3] + bv.w);
                        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                        // (old + r, in this order: the same rounding as the per-channel path)
                        if (accum) v = make_float4(old[g4].x + v.x, old[g4].y + v.y, old[g4].z + v.z, old[g4].w + v.w);
                        *reinterpret_cast<float4*>(orow + co) = v;
                    }
                }
            }
            return;
        }
        // per-channel validity (head 1x1 convs: 36 / 90 / 720 channels)
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int m = bp0 + wp * WTP + j * 32 + frow;
            if (m >= a.M) continue;
            const int out_off = a.rows[m].out_off;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int co = bc0 + wc * WTC + i * 32 + g4 * 8 + fhalf * 4;
                    if (co >= a.cout_valid) continue;
                    const float4 bv = *reinterpret_cast<const float4*>(G.bias + co);
                    float v[4] = {acc[i][j][g4 * 4 + 0] + bv.x, acc[i][j][g4 * 4 + 1] + bv.y,
                                  acc[i][j][g4 * 4 + 2] + bv.z, acc[i][j][g4 * 4 + 3] + bv.w};
                    float* o = reinterpret_cast<float*>(G.out) + (size_t)out_off * a.out_cstride + co;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (co + q >= a.cout_valid) continue;
                        const float r = relu ? fmaxf(v[q], 0.f) : v[q];
                        o[q] = accum ? o[q] + r : r;
                    }
                }
            }
        }
        return;
    }

    if constexpr (SPLIT) {
        // ---- bf16x3 outputs: every value leaves as a (hi, lo) bf16 pair, 32 hi then 32 lo per 64-slot group of the NHWC
        // pixel row.  Same route as the bf16 tile below -- registers -> chunk-swizzled [pixel][slot] LDS tile -> 16-byte
        // stores of whole pixel rows -- but a pixel row is 4 bytes per channel, so the 256x256 tile goes in FP passes of
        // one pixel fragment per wave (128 pixels x 1 KiB = the 128 KiB the staging buffers leave).
        constexpr int EPASS = (BP * BC * 4 > Cfg::MAIN) ? FP : 1;
        constexpr int JP = FP / EPASS;               // pixel fragments per wave per pass
        constexpr int WPP = JP * 32;                 // pixels per wave per pass
        constexpr int PPASS = WP * WPP;              // pixels per pass
        constexpr int ROW2 = BC * 4;                 // bytes per pixel row of the tile
        constexpr int CPR2 = ROW2 / 16;
        static_assert(PPASS * ROW2 <= Cfg::MAIN, "split epilogue tile does not fit the staging area");
        uint32_t rng_seed_lo = a.seed_lo, rng_seed_hi = a.seed_hi, rng_image_base = a.image_base;
        if (a.dyn_rng) { rng_seed_lo = a.dyn_rng[0]; rng_seed_hi = a.dyn_rng[1]; rng_image_base = a.dyn_rng[2]; }
        // (uniform: the Philox key schedule -- 20 values -- stays in scalar registers)
        rng_seed_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)rng_seed_lo);
        rng_seed_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)rng_seed_hi);
        // (the fan-out launch of this mode runs on the row-reuse loop too since round 4 -- xreuse == 2 with fan_count > 1: the sample
        // loop below re-runs both passes per sample; kernel_guard's no-spill check covers that build, <256,256,2,4,0,true,true>, and
        // the f16mx kernels)
        const int fan = (drop && a.fan_count > 1) ? a.fan_count : 1;
        const uint32_t thr_m1 = a.drop_threshold > 0 ? a.drop_threshold - 1u : 0u;
        const uint32_t thr_m1_x2 = thr_m1 | (thr_m1 << 16);
        uint16_t* out16 = reinterpret_cast<uint16_t*>(G.out);

        // pass PASS of the tile: bias / residual / ReLU / (hi, lo) split / dropout mask in registers -> LDS tile [PPASS][ROW2]
        const float clamp_lo = relu ? 0.f : -65504.0f;            // (uniform: hx / h4 epilogues)
        auto write_pass = [&](auto PASS, int n) {
            constexpr int pass = decltype(PASS)::value;
#pragma unroll
            for (int jj = 0; jj < JP; ++jj) {
                constexpr int dummy = 0; (void)dummy;
                const int j = pass * JP + jj;
                const int pixl = wp * WTP + j * 32 + frow;
                const int2 rg = s_rng[pixl];
                const int ro = s_res[pixl];
                const uint32_t img = rng_image_base + ((uint32_t)rg.y >> 16);
                const uint32_t sample = a.sample_base + (a.fan_count > 1 ? (uint32_t)n : ((uint32_t)rg.y & 0xFFFFu));
                if constexpr (MXK != 0) {
                    if (G.out_hx == 2) {
                        // ---- f16mx4: the row leaves in the h4 format (header of this file): f16 hi to the H chunks (chunk 3 (q >> 1) + (q & 1) of
                        // 64-channel group q), the lane's 16 channels of cout fragment i = block (x = wc, ks = i, half = fhalf) of X chunk
                        // 3 wc + 2: sixteen v_cvt_scalef32_pk_fp4_f32 pack {hi4, lo4'} pairs (byte k = channel k of the block) under the
                        // block's scale 2^e, e = floor(log2(max * 4/3)) - 2 (the largest |hi| lands in [3, 6], |lo'| <= 4); the four scale
                        // bytes of the lane's blocks = one dword of the scale chunk (chunk 6, byte 8 wc + 4 fhalf + i).
                        static_assert(BC == 256 && WC == 2, "h4 epilogue: 256-cout tile, two cout halves");
                        int fr_i = frow, fh_i = fhalf;
#if defined(__HIP_DEVICE_COMPILE__)
                        asm volatile("" : "+v"(fr_i), "+v"(fh_i));
#endif
                        const int lr = wp * WPP + jj * 32 + fr_i;
                        char* prow = smem + lr * ROW2;
                        uint32_t sdw = 0u;
#pragma unroll
                        for (int i = 0; i < FC; ++i) {
                            const int col0 = wc * WTC + i * 32 + fh_i * 4;            // channel of (g4 = 0, r = 0)
                            Philox4 rr{0u, 0u, 0u, 0u};
                            if (drop) rr = philox4x32_10((uint32_t)rg.x, dropout_group16(bc0 + col0), sample | ((uint32_t)G.layer_id << 16), img, rng_seed_lo, rng_seed_hi);
                            float hv[16], lv[16];
                            float mx = 6.103515625e-05f;                              // 2^-14
                            const int q = col0 >> 6;
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const int col = col0 + g4 * 8;
                                const float4 bv = *reinterpret_cast<const float4*>(s_bias + col);
                                float v[4] = {__builtin_fmaf(acc[i][j][g4 * 4 + 0], epi_scale, bv.x), __builtin_fmaf(acc[i][j][g4 * 4 + 1], epi_scale, bv.y),
                                              __builtin_fmaf(acc[i][j][g4 * 4 + 2], epi_scale, bv.z), __builtin_fmaf(acc[i][j][g4 * 4 + 3], epi_scale, bv.w)};
                                // ReLU and the f16 clamp are ONE v_med3_f32 between clamp_lo (0 with ReLU, -65504 without) and 65504 (round 6: the
                                // separate maximum + select + median were three of the twelve vector instructions per element; same values: the
                                // clamp commutes with the dropout's zeroing, and max(x, 0) then min(., 65504) is the median of the three)
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], clamp_lo, 65504.0f);
                                if (drop) {
                                    const DropPair dw = dropout_run_windows(rr, g4);
                                    const uint32_t thr = a.drop_threshold;
                                    v[0] = (dw.x & 0xFFFFu) >= thr ? v[0] : 0.f; v[1] = (dw.x >> 16) >= thr ? v[1] : 0.f;
                                    v[2] = (dw.y & 0xFFFFu) >= thr ? v[2] : 0.f; v[3] = (dw.y >> 16) >= thr ? v[3] : 0.f;
                                }
                                uint16_t hb[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const _Float16 hh = (_Float16)v[r];
                                    hb[r] = __builtin_bit_cast(uint16_t, hh);
                                    const float hf = (float)hh;
                                    hv[g4 * 4 + r] = hf;
                                    lv[g4 * 4 + r] = (v[r] - hf) * 2048.0f;
                                    mx = fmaxf(mx, fabsf(hf));
                                }
                                const int c64 = col & 63, chH = (3 * (q >> 1) + (q & 1)) * 8 + (c64 >> 3);
                                *reinterpret_cast<uint2*>(prow + (((chH ^ lr) & (CPR2 - 1)) << 4) + (c64 & 7) * 2) =
                                    make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
                            }
                            const uint32_t eb = (__float_as_uint(mx * 1.3333334f) >> 23) - 2u;      // biased exponent = the E8M0 byte
                            const float sc = __uint_as_float(eb << 23);
                            uint32_t pk[4];
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                uint32_t w_ = 0u;
                                w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w_, hv[4 * d + 0], lv[4 * d + 0], sc, 0);
                                w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w_, hv[4 * d + 1], lv[4 * d + 1], sc, 1);
                                w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w_, hv[4 * d + 2], lv[4 * d + 2], sc, 2);
                                w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w_, hv[4 * d + 3], lv[4 * d + 3], sc, 3);
                                pk[d] = w_;
                            }
                            const int chX = (3 * wc + 2) * 8 + 2 * i + fh_i;
                            *reinterpret_cast<uint4*>(prow + (((chX ^ lr) & (CPR2 - 1)) << 4)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                            sdw |= eb << (8 * i);
                            __builtin_amdgcn_sched_barrier(0);           // one fragment at a time (register pressure)
                        }
                        *reinterpret_cast<uint32_t*>(prow + (((48 ^ lr) & (CPR2 - 1)) << 4) + wc * 8 + fh_i * 4) = sdw;
                        // the rest of the scale chunk and the unused eighth chunk are part of the 1 KiB row the store loop copies: zeros, not
                        // whatever the staging buffers left there (the activation buffers stay byte-for-byte reproducible; never read)
                        {
                            const int zq = wc * 2 + fh_i;
#pragma unroll
                            for (int z = 0; z < 4; ++z) {
                                const int pz = 49 + zq + 4 * z;
                                if (pz < 64) *reinterpret_cast<uint4*>(prow + (((pz ^ lr) & (CPR2 - 1)) << 4)) = make_uint4(0u, 0u, 0u, 0u);
                            }
                        }
                        continue;
                    }
                    if (G.out_hx) {
                        // ---- f16mx: the row leaves in the hx format (header of this file).  A lane's 16 channels of cout fragment i --
                        // 32i + 8*g4 + 4*fhalf + r -- are exactly one MX block: f16 hi to the H chunk (4 x 8 bytes), then ONE
                        // v_cvt_scalef32_2xpk16_fp6_f32 packs {hi6, lo6'} of the 16 channels (it interleaves its two sources) under the
                        // block's scale, 24 bytes + the scale byte to the lane's slot of the X chunk.  Dropout zeroes the VALUES (the
                        // fields of a 6-bit stream cannot be masked), same decisions as the pair form.
                        static_assert(BC == 256 && WC == 2, "hx epilogue: 256-cout tile, two cout halves");
#pragma unroll
                        for (int i = 0; i < FC; ++i) {
                            int fr_i = frow, fh_i = fhalf;
#if defined(__HIP_DEVICE_COMPILE__)
                            asm volatile("" : "+v"(fr_i), "+v"(fh_i));
#endif
                            const int lr = wp * WPP + jj * 32 + fr_i;
                            char* prow = smem + lr * ROW2;
                            const int col0 = wc * WTC + i * 32 + fh_i * 4;            // channel of (g4 = 0, r = 0)
                            Philox4 rr{0u, 0u, 0u, 0u};
                            if (drop) rr = philox4x32_10((uint32_t)rg.x, dropout_group16(bc0 + col0), sample | ((uint32_t)G.layer_id << 16), img, rng_seed_lo, rng_seed_hi);
                            f32x16v hv, lv;
                            float mx = 6.103515625e-05f;                              // 2^-14: |lo'| of an f16-subnormal value stays below it
                            const int q = col0 >> 6;
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const int col = col0 + g4 * 8;
                                const float4 bv = *reinterpret_cast<const float4*>(s_bias + col);
                                float v[4] = {__builtin_fmaf(acc[i][j][g4 * 4 + 0], epi_scale, bv.x), __builtin_fmaf(acc[i][j][g4 * 4 + 1], epi_scale, bv.y),
                                              __builtin_fmaf(acc[i][j][g4 * 4 + 2], epi_scale, bv.z), __builtin_fmaf(acc[i][j][g4 * 4 + 3], epi_scale, bv.w)};
                                // ReLU and the f16 clamp are ONE v_med3_f32 between clamp_lo (0 with ReLU, -65504 without) and 65504 (round 6: the
                                // separate maximum + select + median were three of the twelve vector instructions per element; same values: the
                                // clamp commutes with the dropout's zeroing, and max(x, 0) then min(., 65504) is the median of the three)
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], clamp_lo, 65504.0f);
                                if (drop) {
                                    const DropPair dw = dropout_run_windows(rr, g4);
                                    const uint32_t thr = a.drop_threshold;
                                    v[0] = (dw.x & 0xFFFFu) >= thr ? v[0] : 0.f; v[1] = (dw.x >> 16) >= thr ? v[1] : 0.f;
                                    v[2] = (dw.y & 0xFFFFu) >= thr ? v[2] : 0.f; v[3] = (dw.y >> 16) >= thr ? v[3] : 0.f;
                                }
                                uint16_t hb[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    // (f16 range: a tower activation beyond +-65 504 -- none in a trained RetinaNet, whose tower outputs stay below a
                                    // few hundred -- is clamped instead of becoming an infinity; the bf16x3 mode has no such limit)
                                    const _Float16 hh = (_Float16)v[r];
                                    hb[r] = __builtin_bit_cast(uint16_t, hh);
                                    const float hf = (float)hh;
                                    hv[g4 * 4 + r] = hf;
                                    lv[g4 * 4 + r] = (v[r] - hf) * 2048.0f;
                                    mx = fmaxf(mx, fabsf(v[r]));
                                }
                                const int c64 = col & 63, chH = q * 16 + (c64 >> 3);
                                *reinterpret_cast<uint2*>(prow + (((chH ^ lr) & (CPR2 - 1)) << 4) + (c64 & 7) * 2) =
                                    make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
                            }
                            // block scale 2^e, e = floor(log2(max * 16/15)) - 2: the largest element lands in [2, 7.5]
                            const uint32_t eb = (__float_as_uint(mx * 1.0666667f) >> 23) - 2u;      // biased exponent = the E8M0 byte
                            const i32x6 pk = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(hv, lv, __uint_as_float(eb << 23));
                            const int chX = q * 16 + 8 + ((col0 >> 5) & 1) * 4 + fh_i;
                            *reinterpret_cast<uint4*>(prow + (((chX ^ lr) & (CPR2 - 1)) << 4)) = make_uint4((uint32_t)pk[0], (uint32_t)pk[1], (uint32_t)pk[2], (uint32_t)pk[3]);
                            *reinterpret_cast<uint4*>(prow + ((((chX + 2) ^ lr) & (CPR2 - 1)) << 4)) = make_uint4((uint32_t)pk[4], (uint32_t)pk[5], 0u, eb);
                            __builtin_amdgcn_sched_barrier(0);           // one fragment at a time (register pressure)
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    Philox4 rr{0u, 0u, 0u, 0u};
                    // the lane's row / half re-derived behind an opaque barrier per cout fragment: the tile addresses of this
                    // fragment are then computed HERE -- otherwise the whole unrolled epilogue's address arithmetic is scheduled
                    // to the top of the block, where all 128 accumulators are live, and spills by the hundred
                    int fr_i = frow, fh_i = fhalf;
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : "+v"(fr_i), "+v"(fh_i));
#endif
                    const int lr = wp * WPP + jj * 32 + fr_i;
                    char* prow = smem + lr * ROW2;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int col = wc * WTC + i * 32 + g4 * 8 + fh_i * 4;
                        const int slot = (col >> 5) * 64 + (col & 31);          // hi half; the lo half sits 32 slots on
                        const float4 bv = *reinterpret_cast<const float4*>(s_bias + col);
                        float v[4] = {__builtin_fmaf(acc[i][j][g4 * 4 + 0], epi_scale, bv.x), __builtin_fmaf(acc[i][j][g4 * 4 + 1], epi_scale, bv.y),
                                      __builtin_fmaf(acc[i][j][g4 * 4 + 2], epi_scale, bv.z), __builtin_fmaf(acc[i][j][g4 * 4 + 3], epi_scale, bv.w)};
                        if (G.res) {
                            const uint16_t* rp = reinterpret_cast<const uint16_t*>(G.res) + (size_t)ro * a.res_cstride + bc0 * 2 + slot;
                            const uint2 rh = *reinterpret_cast<const uint2*>(rp), rl = *reinterpret_cast<const uint2*>(rp + 32);
                            v[0] += (bf16_to_f32(rh.x & 0xFFFFu) + bf16_to_f32(rl.x & 0xFFFFu)) * epi_scale;
                            v[1] += (bf16_to_f32(rh.x >> 16) + bf16_to_f32(rl.x >> 16)) * epi_scale;
                            v[2] += (bf16_to_f32(rh.y & 0xFFFFu) + bf16_to_f32(rl.y & 0xFFFFu)) * epi_scale;
                            v[3] += (bf16_to_f32(rh.y >> 16) + bf16_to_f32(rl.y >> 16)) * epi_scale;
                        }
                        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        uint2 hi, lo;
                        hi.x = pack_bf16x2(v[0], v[1]); hi.y = pack_bf16x2(v[2], v[3]);
                        lo.x = pack_bf16x2(v[0] - bf16_to_f32(hi.x & 0xFFFFu), v[1] - bf16_to_f32(hi.x >> 16));
                        lo.y = pack_bf16x2(v[2] - bf16_to_f32(hi.y & 0xFFFFu), v[3] - bf16_to_f32(hi.y >> 16));
                        if (drop) {                       // dropout contract v3, as in the bf16 epilogue below: the fragment's four runs share one call
                            if (g4 == 0)
                                rr = philox4x32_10((uint32_t)rg.x, dropout_group16(bc0 + col), sample | ((uint32_t)G.layer_id << 16), img,
                                                   rng_seed_lo, rng_seed_hi);
                            const DropPair dw = dropout_run_windows(rr, g4);
                            const uint32_t m0 = keep_mask_u16x2(dw.x, thr_m1_x2);
                            const uint32_t m1 = keep_mask_u16x2(dw.y, thr_m1_x2);
                            hi.x &= m0; lo.x &= m0; hi.y &= m1; lo.y &= m1;
                        }
                        const int ch = slot >> 3;
                        *reinterpret_cast<uint2*>(prow + (((ch ^ lr) & (CPR2 - 1)) << 4) + (slot & 7) * 2) = hi;
                        *reinterpret_cast<uint2*>(prow + ((((ch + 4) ^ lr) & (CPR2 - 1)) << 4) + (slot & 7) * 2) = lo;
                        __builtin_amdgcn_sched_barrier(0);           // one 4-channel run at a time (register pressure)
                    }
                }
            }
        };

        // Fused 1x1 head output conv (+ MC aggregation) of the bf16x3 mode, row-reuse tower kernel only: after each pass the
        // wave multiplies its share of the pass's 128 pixels -- cout2 fragment wave>>1, pixel fragments 2(wave&1), 2(wave&1)+1 --
        // with the (hi, lo) weight pairs read straight from global memory (96 KB per head, L2-resident): per 16-channel half the
        // products hi*lo, lo*hi, hi*hi in the main loop's order, i.e. the sums of a separate bf16x3 1x1 launch, bit for bit.  The
        // fp32 outputs of both passes stay in registers (the accumulators are dead by then) until the whole tile is done; then
        // they go out as [B,N,A,.] rows or, aggregating, through LDS into agg_reduce_* exactly like the bf16 mode's tile.
        // (Its own straight-line branch of the epilogue: the four output fragments are defined on every path that reads them.)
        constexpr bool CAN_FUSE_S = XR && BC == 256 && BP == 256 && WC * WP == 8 && EPASS == 2 && JP == 1;
        if constexpr (CAN_FUSE_S) {
            if (G.w2 != nullptr) {
                const int f2 = wave >> 1;
                const bool fuse_active = f2 * 32 < G.cout2;
                (void)fuse_active;
#if defined(__HIP_DEVICE_COMPILE__)
                // weight fragments by buffer loads: resource in SGPRs, ONE 32-bit lane offset, batch / fragment offsets as scalar +
                // immediate (sixteen 64-bit lane pointers would not fit beside the accumulators); the next batch of fragments is in flight
                // while the current one multiplies
                const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(G.w2), 0, -1, 0x00020000);
                auto one_by_one = [&](f32x16& y0, f32x16& y1, u32x4 (&wn)[4]) {
                    const int lr0 = (wave & 1) * 64 + frow, lr1 = lr0 + 32;
                    const char* prow0 = smem + lr0 * ROW2;
                    const char* prow1 = smem + lr1 * ROW2;
                    int w2lane = ((f2 * 32 + frow) * 512 + fhalf * 8) * 2;
                    asm volatile("" : "+v"(w2lane));
#pragma unroll 1
                    for (int cb = 0; cb < 8; ++cb) {          // a batch = 32 channels = the (hi, lo) fragments of two 16-channel halves
                        bf16x8 wc[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) wc[k] = __builtin_bit_cast(bf16x8, wn[k]);
                        if (cb < 7) {
                            const int so = (cb + 1) * 128;
                            wn[0] = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2lane, so, 0);      wn[1] = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2lane + 64, so, 0);
                            wn[2] = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2lane + 32, so, 0); wn[3] = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2lane + 96, so, 0);
                        }
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            // half k of the batch: the lane's 8 channels c8 = (cb*2 + k)*2 + fhalf -> chunk of their hi half
                            const int ch = cb * 8 + k * 2 + fhalf;
                            const bf16x8 Bh0 = *reinterpret_cast<const bf16x8*>(prow0 + (((ch ^ lr0) & (CPR2 - 1)) << 4));
                            const bf16x8 Bl0 = *reinterpret_cast<const bf16x8*>(prow0 + ((((ch + 4) ^ lr0) & (CPR2 - 1)) << 4));
                            const bf16x8 Bh1 = *reinterpret_cast<const bf16x8*>(prow1 + (((ch ^ lr1) & (CPR2 - 1)) << 4));
                            const bf16x8 Bl1 = *reinterpret_cast<const bf16x8*>(prow1 + ((((ch + 4) ^ lr1) & (CPR2 - 1)) << 4));
                            y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[2 * k], Bl0, y0, 0, 0, 0);     y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[2 * k], Bl1, y1, 0, 0, 0);
                            y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[2 * k + 1], Bh0, y0, 0, 0, 0); y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[2 * k + 1], Bh1, y1, 0, 0, 0);
                            y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[2 * k], Bh0, y0, 0, 0, 0);     y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[2 * k], Bh1, y1, 0, 0, 0);
                        }
                    }
                };
                auto first_batch = [&](u32x4 (&wn)[4]) {             // requested before the pass's barrier: L2 latency under the wait
                    int w2lane = ((f2 * 32 + frow) * 512 + fhalf * 8) * 2;
                    asm volatile("" : "+v"(w2lane));
                    wn[0] = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2lane, 0, 0);      wn[1] = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2lane + 64, 0, 0);
                    wn[2] = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2lane + 32, 0, 0); wn[3] = __builtin_amdgcn_raw_buffer_load_b128(w2rsrc, w2lane + 96, 0, 0);
                };
                __syncthreads();                          // all waves are done with the staging buffers
                f32x16 y00, y01, y10, y11;                // [pass][pixel fragment 2(wave&1) + h]
                u32x4 wn[4];
                // ---- pass 0
                write_pass(std::integral_constant<int, 0>{}, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) wn[k] = u32x4{0u, 0u, 0u, 0u};
                if (fuse_active) first_batch(wn);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) { y00[r] = 0.f; y01[r] = 0.f; }
                if (fuse_active) one_by_one(y00, y01, wn);
                __syncthreads();                          // the pass's tile has been read: the next pass may overwrite it
                // ---- pass 1
                write_pass(std::integral_constant<int, 1>{}, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) wn[k] = u32x4{0u, 0u, 0u, 0u};
                if (fuse_active) first_batch(wn);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) { y10[r] = 0.f; y11[r] = 0.f; }
                if (fuse_active) one_by_one(y10, y11, wn);
                __syncthreads();                          // ... or the fp32 output tile
                // ---- outputs: cout2 rows f2*32 + g4*8 + fhalf*4 + r of tile pixels (2(wave&1) + h)*64 + pass*32 + frow
                int frow_o = frow, fhalf_o = fhalf;
                asm volatile("" : "+v"(frow_o), "+v"(fhalf_o));      // output addresses are computed HERE (hoisted, they spill)
                auto each_fragment = [&](auto&& fn) {
                    fn(y00, ((wave & 1) * 2 + 0) * 64 + 0 * 32 + frow_o); fn(y01, ((wave & 1) * 2 + 1) * 64 + 0 * 32 + frow_o);
                    fn(y10, ((wave & 1) * 2 + 0) * 64 + 1 * 32 + frow_o); fn(y11, ((wave & 1) * 2 + 1) * 64 + 1 * 32 + frow_o);
                };
                if (G.agg_kind != AGG_NONE) {
                    const int ystride = ((G.cout2 + 31) & ~31) + 4;            // floats per row: = 4 mod 32, conflict-free 16-byte writes
                    float* ytile = reinterpret_cast<float*>(smem);
                    if (fuse_active)
                        each_fragment([&](const f32x16& y, int row) {
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const int co2 = f2 * 32 + g4 * 8 + fhalf_o * 4;
                                if (co2 >= G.cout2) continue;
                                const float4 b2 = *reinterpret_cast<const float4*>(G.bias2 + co2);
                                *reinterpret_cast<float4*>(ytile + (size_t)row * ystride + co2) =
                                    make_float4(y[g4 * 4 + 0] + b2.x, y[g4 * 4 + 1] + b2.y, y[g4 * 4 + 2] + b2.z, y[g4 * 4 + 3] + b2.w);
                            }
                        });
                    __syncthreads();
                    if (G.agg_kind == AGG_CLS) {
                        if (G.agg_C == 8) agg_reduce_cls<8>(G, ytile, ystride, s_off, s_off2, tid, THREADS, BP);
                        else agg_reduce_cls<4>(G, ytile, ystride, s_off, s_off2, tid, THREADS, BP);
                    } else if (G.agg_kind == AGG_BOX) {
                        agg_reduce_box(G, ytile, ystride, s_off, s_off2, tid, THREADS, BP);
                    } else {
                        agg_reduce_cov(G, ytile, ystride, s_off, s_off2, tid, THREADS, BP);
                    }
                    return;
                }
                if (fuse_active)
                    each_fragment([&](const f32x16& y, int pixl) {
                        if (s_off[pixl] < 0) return;
                        float* orow = G.out2 + (size_t)s_off2[pixl] * G.out2_cstride;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int co2 = f2 * 32 + g4 * 8 + fhalf_o * 4;
                            if (co2 >= G.cout2) continue;
                            const float4 b2 = *reinterpret_cast<const float4*>(G.bias2 + co2);
                            const float v0 = y[g4 * 4 + 0] + b2.x, v1 = y[g4 * 4 + 1] + b2.y;
                            const float v2 = y[g4 * 4 + 2] + b2.z, v3 = y[g4 * 4 + 3] + b2.w;
                            if (co2 + 1 < G.cout2) *reinterpret_cast<float2*>(orow + co2) = make_float2(v0, v1);
                            else orow[co2] = v0;
                            if (co2 + 3 < G.cout2) *reinterpret_cast<float2*>(orow + co2 + 2) = make_float2(v2, v3);
                            else if (co2 + 2 < G.cout2) orow[co2 + 2] = v2;
                        }
                    });
#endif
                return;                               // the tile itself has no other consumer (fused groups never fan out)
            }
        }

        auto store_pass = [&](int pass, int n) {      // LDS tile -> 16-byte stores of (hi chunk, lo chunk) pairs of whole pixel rows
            for (int q = tid; q < PPASS * (CPR2 / 2); q += THREADS) {
                const int lr = q / (CPR2 / 2), hc = q % (CPR2 / 2);
                const int ch = (hc >> 2) * 8 + (hc & 3);
                const int rem = lr % WPP;
                const int pixl = (lr / WPP) * WTP + (pass * JP + rem / 32) * 32 + (rem & 31);
                const int off = s_off[pixl];
                if (off < 0) continue;
                const char* prow = smem + lr * ROW2;
                const uint4 vh = *reinterpret_cast<const uint4*>(prow + (((ch ^ lr) & (CPR2 - 1)) << 4));
                const uint4 vl = *reinterpret_cast<const uint4*>(prow + ((((ch + 4) ^ lr) & (CPR2 - 1)) << 4));
                const size_t e = ((size_t)off + (size_t)n * a.fan_stride) * a.out_cstride + bc0 * 2 + ch * 8;
                if (MXK != 0 && (a.flags & CONV_NT_OUT)) {        // f16mx towers, BOD_NT_STORES bits 3 / 4: streaming stores (A/B switch)
                    __builtin_nontemporal_store(u32x4{vh.x, vh.y, vh.z, vh.w}, reinterpret_cast<u32x4*>(out16 + e));
                    __builtin_nontemporal_store(u32x4{vl.x, vl.y, vl.z, vl.w}, reinterpret_cast<u32x4*>(out16 + e + 32));
                    continue;
                }
                *reinterpret_cast<uint4*>(out16 + e) = vh;
                *reinterpret_cast<uint4*>(out16 + e + 32) = vl;
                if (G.out_relu) {                 // relu(hi + lo): the pair survives iff hi is not negative
                    auto keep = [](uint32_t h) { return ~(((h >> 15) & 0x00010001u) * 0xFFFFu); };
                    const uint32_t k0 = keep(vh.x), k1 = keep(vh.y), k2 = keep(vh.z), k3 = keep(vh.w);
                    uint16_t* o2 = reinterpret_cast<uint16_t*>(G.out_relu);
                    *reinterpret_cast<uint4*>(o2 + e) = make_uint4(vh.x & k0, vh.y & k1, vh.z & k2, vh.w & k3);
                    *reinterpret_cast<uint4*>(o2 + e + 32) = make_uint4(vl.x & k0, vl.y & k1, vl.z & k2, vl.w & k3);
                }
            }
        };
        __syncthreads();                              // all waves are done with the staging buffers
        for (int n = 0; n < fan; ++n) {
            write_pass(std::integral_constant<int, 0>{}, n);
            __syncthreads();
            store_pass(0, n);
            if constexpr (EPASS > 1) {
                static_assert(EPASS <= 2, "two passes at most");
                __syncthreads();
                write_pass(std::integral_constant<int, EPASS - 1>{}, n);
                __syncthreads();
                store_pass(EPASS - 1, n);
            }
            if (n + 1 < fan) __syncthreads();
        }
        return;
    }

    // ---- bf16 outputs: registers -> swizzled [pixel][cout] LDS tile -> 16-byte row-contiguous stores
    __syncthreads();                                  // all waves are done with the staging buffers
    // the row-reuse launches (head towers) have neither a residual nor a second relu output: compiled out there, which
    // keeps the kernel at the 256-register budget without a spill
    constexpr bool CAN_RES = !XR;
    // streaming outputs: activations of 2-22 GB per launch that the next launch reads from HBM anyway are stored non-temporally, so
    // that they do not evict what the kernels re-read through L2 / the Infinity Cache (weights, halo rows) -- ConvArgs.flags & CONV_NT_OUT
    const bool nt_out = (a.flags & CONV_NT_OUT) != 0;
    int2 rng[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) rng[j] = s_rng[wp * WTP + j * 32 + frow];    // rng_p, rng_zs
    // Residual (bottleneck shortcuts, FPN merges): the tile's residual pixels come in by LDS-DMA -- whole 16-byte pieces of the
    // NHWC rows, every lane of a wave on consecutive pieces (two 512-byte rows per instruction at 256 channels) -- straight into
    // the [pixel][cout] tile image the outputs will overwrite: piece (pixel, physical chunk pc) holds the row's logical chunk
    // pc ^ pixel, like the output tile, so a lane reads its four residual channels from the position it later writes its result
    // to.  (The first form read them from global memory in the accumulator layout: 8 bytes per lane, 32 rows per instruction --
    // eight requests per 128-byte line; the memory-bound 1x1 layers of res2 / res3 spent their epilogue in the texture path.)
    const bool res_lds = CAN_RES && G.res != nullptr && ABL == 0 && a.variant != 82;      // (variant 82: the register form, A/B)
    if (CAN_RES && res_lds) {
        const char* rbase = reinterpret_cast<const char*>(G.res) + (size_t)bc0 * 2;
#pragma unroll
        for (int it = 0; it < BP * CPR / THREADS; ++it) {
            const int q = it * THREADS + tid;
            const int pixl = q / CPR, pc = q % CPR;
            const int cp = (pc ^ pixl) & (CPR - 1);
            const char* src = rbase + ((size_t)s_res[pixl] * a.res_cstride + cp * 8) * 2;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(smem + (it * THREADS + wave * 64) * 16), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    int res_off[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) res_off[j] = (CAN_RES && !res_lds) ? s_res[wp * WTP + j * 32 + frow] : 0;
    // phase A: finish the arithmetic and pack to bf16 (halves the live registers before the RNG):
    // rounding(x*scale) then zeroing == zeroing then rounding, so the mask is applied on packed words
    uint2 pk16[FC16][FP16];                           // M16: (cout fragment, pixel fragment) -> the lane's 4 packed channels
    if constexpr (M16) {
#pragma unroll
        for (int fp = 0; fp < FP16; ++fp)
#pragma unroll
            for (int fc = 0; fc < FC16; ++fc) {
                const float4 bv = *reinterpret_cast<const float4*>(s_bias + wc * WTC + fc * 16 + q4 * 4);
                pk16[fc][fp].x = pack_bf16x2(__builtin_fmaf(acc4[fc][fp][0], epi_scale, bv.x), __builtin_fmaf(acc4[fc][fp][1], epi_scale, bv.y));
                pk16[fc][fp].y = pack_bf16x2(__builtin_fmaf(acc4[fc][fp][2], epi_scale, bv.z), __builtin_fmaf(acc4[fc][fp][3], epi_scale, bv.w));
                if (relu) { pk16[fc][fp].x = relu_bf16x2_pk(pk16[fc][fp].x); pk16[fc][fp].y = relu_bf16x2_pk(pk16[fc][fp].y); }
            }
    }
    uint2 pk[FC][FP][4];
#pragma unroll
    for (int j = 0; j < (M16 ? 0 : FP); ++j) {
#pragma unroll
        for (int i = 0; i < FC; ++i) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = wc * WTC + i * 32 + g4 * 8 + fhalf * 4;
                const int co = bc0 + col;
                const float4 bv = *reinterpret_cast<const float4*>(s_bias + col);
                float v[4] = {__builtin_fmaf(acc[i][j][g4 * 4 + 0], epi_scale, bv.x), __builtin_fmaf(acc[i][j][g4 * 4 + 1], epi_scale, bv.y),
                              __builtin_fmaf(acc[i][j][g4 * 4 + 2], epi_scale, bv.z), __builtin_fmaf(acc[i][j][g4 * 4 + 3], epi_scale, bv.w)};
                if (CAN_RES && G.res) {
                    uint2 r;
                    if (res_lds) {
                        const int pixl = wp * WTP + j * 32 + frow;
                        r = *reinterpret_cast<const uint2*>(smem + pixl * (BC * 2) + ((((col >> 3) ^ pixl) & (CPR - 1)) << 4) + (col & 7) * 2);
                    } else {
                        r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(G.res) + (size_t)res_off[j] * a.res_cstride + co);
                    }
                    v[0] += bf16_to_f32(r.x & 0xFFFFu) * epi_scale; v[1] += bf16_to_f32(r.x >> 16) * epi_scale;
                    v[2] += bf16_to_f32(r.y & 0xFFFFu) * epi_scale; v[3] += bf16_to_f32(r.y >> 16) * epi_scale;
                }
                pk[i][j][g4].x = pack_bf16x2(v[0], v[1]);
                pk[i][j][g4].y = pack_bf16x2(v[2], v[3]);
                if (relu) { pk[i][j][g4].x = relu_bf16x2_pk(pk[i][j][g4].x); pk[i][j][g4].y = relu_bf16x2_pk(pk[i][j][g4].y); }
            }
        }
    }
    phase_stamp<ABL>(tstamp, 2);            // barrier + bias/ReLU/scale/pack
    __builtin_amdgcn_sched_barrier(0);
    // fused 1x1 head conv: wave w owns cout2 fragment w/2 and pixel fragments (w&1)*HALFP.. of the tile;
    // its 16 weight fragments (64 VGPRs) are fetched after the tile barrier (L2-resident: 48 KB per head)
    constexpr bool CAN_FUSE = (BC == 256) && (BP % 64 == 0) && (WC * WP == 8);
    constexpr int HALFP = BP / 64;               // pixel fragments per wave in the fused conv
    const bool fuse = CAN_FUSE && G.w2 != nullptr;
    const int f2 = wave >> 1;
    const bool fuse_active = fuse && f2 * 32 < G.cout2;
    bf16x8 w2f[16];
    const uint16_t* wp2 = reinterpret_cast<const uint16_t*>(G.w2) + (size_t)(f2 * 32 + frow) * 256 + fhalf * 8;
    if (ABL == 31) {
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" ::"v"(pk[i][j][g4].x), "v"(pk[i][j][g4].y));
#endif
                }
        return;
    }
    uint32_t rng_seed_lo = a.seed_lo, rng_seed_hi = a.seed_hi, rng_image_base = a.image_base;
    if (a.dyn_rng) { rng_seed_lo = a.dyn_rng[0]; rng_seed_hi = a.dyn_rng[1]; rng_image_base = a.dyn_rng[2]; }      // uniform: scalar loads
    const int fan = (drop && a.fan_count > 1) ? a.fan_count : 1;
    const uint32_t thr_m1 = a.drop_threshold > 0 ? a.drop_threshold - 1u : 0u;        // (dropout layers have threshold >= 1)
    const uint32_t thr_m1_x2 = thr_m1 | (thr_m1 << 16);
    if (fan > 1 && !fuse && !(CAN_RES && G.out_relu) && (ABL == 0 || ABL == 5)) {
        // ---- N-way dropout fan-out (first tower layer: one convolution, N masked copies).  The unmasked tile goes
        // through LDS ONCE; each thread then keeps (pixel, 16-channel group) items in registers and, per sample, draws
        // ONE Philox call (contract v3: a call decides the runs {x..x+3, x+8.., x+16.., x+24..} of a 32-channel block, i.e. half of
        // this item and half of its neighbour's -- the neighbour draws the other call and they trade mask halves through DPP),
        // masks and stores 2 x 16 bytes -- no LDS traffic and no barrier inside the sample loop.
        if constexpr (M16) {
#pragma unroll
            for (int fp = 0; fp < FP16; ++fp) {
                const int pixl = wp * WTP + fp * 16 + l15;
                char* prow = smem + pixl * (BC * 2);
#pragma unroll
                for (int fc = 0; fc < FC16; ++fc) {
                    const int col = wc * WTC + fc * 16 + q4 * 4;
                    *reinterpret_cast<uint2*>(prow + ((((col >> 3) ^ pixl) & (CPR - 1)) << 4) + (col & 7) * 2) = pk16[fc][fp];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < (M16 ? 0 : FP); ++j) {
            const int pixl = wp * WTP + j * 32 + frow;
            char* prow = smem + pixl * (BC * 2);
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int col = wc * WTC + i * 32 + g4 * 8 + fhalf * 4;
                    *reinterpret_cast<uint2*>(prow + ((((col >> 3) ^ pixl) & (CPR - 1)) << 4) + (col & 7) * 2) = pk[i][j][g4];
                }
        }
        __syncthreads();
        constexpr int GPR = BC / 16;                                   // 16-channel groups per pixel row
        auto keep2 = [thr_m1_x2](uint32_t w) { return keep_mask_u16x2(w, thr_m1_x2); };
        // A lane owns one (pixel, 16-channel group) item: its two Philox calls decide exactly those 32 bytes.  Stored as they are,
        // every 64-lane store instruction would write 16-byte pieces 32 bytes apart (two half-covered instructions per row, 64
        // separate write requests each); ten samples make this launch the chip's largest store stream (21.5 GB per 256 frames).
        // Lanes 2k / 2k+1 therefore own items k and k + 32 of the wave's 64 and trade halves through DPP (lane ^ 1): the first
        // store instruction of a sample then writes items 0..31 -- lane 2k the first 16 bytes of item k, lane 2k+1 the second --
        // i.e. 1 KiB of CONTIGUOUS pixel rows, the second one items 32..63.  Same values to the same addresses.
        static_assert((BP * GPR) % THREADS == 0 && GPR <= 32 && 64 % GPR == 0, "fan-out item mapping");
        const int odd = lane & 1;
#pragma unroll 1
        for (int it = 0; it < BP * GPR / THREADS; ++it) {
            const int base = it * THREADS + wave * 64;
            const int it_e = base + (lane >> 1), it_o = it_e + 32;      // the pair's two items; this lane owns it_e (even lane) or it_o
            const int q = odd ? it_o : it_e;
            const int pixl = q / GPR, gq = q % GPR;
            const int off_e = s_off[it_e / GPR], off_o = s_off[it_o / GPR];
            const char* prow = smem + pixl * (BC * 2);
            const uint4 va = *reinterpret_cast<const uint4*>(prow + ((((2 * gq) ^ pixl) & (CPR - 1)) << 4));
            const uint4 vb = *reinterpret_cast<const uint4*>(prow + ((((2 * gq + 1) ^ pixl) & (CPR - 1)) << 4));
            const int2 r = s_rng[pixl];
            const uint32_t img = rng_image_base + ((uint32_t)r.y >> 16);
            const int c0 = bc0 + gq * 16;
            // the 32-channel block of items gq & ~1, gq | 1 has two calls: bit 2 of the channel = 0 (runs at +0, +8, +16, +24) and = 1 (+4,
            // +12, +20, +28).  The even item's lane draws the first, the odd item's lane (lane ^ 2: same pixel) the second; an item needs
            // runs u0 = (gq & 1) * 2, u0 + 1 of BOTH calls.
            const bool gq_odd = (gq & 1) != 0;
            const uint32_t gmine = dropout_group16(c0 & ~31) + (gq_odd ? 1u : 0u);
            uint16_t* const obase = reinterpret_cast<uint16_t*>(G.out) + bc0 + odd * 8;
            uint16_t* oe = obase + (size_t)(off_e < 0 ? 0 : off_e) * a.out_cstride + (it_e % GPR) * 16;
            uint16_t* oo = obase + (size_t)(off_o < 0 ? 0 : off_o) * a.out_cstride + (it_o % GPR) * 16;
            const size_t sample_stride = (size_t)a.fan_stride * a.out_cstride;
#pragma unroll 1
            for (int n = 0; n < fan; ++n) {
                const uint32_t key = (a.sample_base + (uint32_t)n) | ((uint32_t)G.layer_id << 16);
                const Philox4 pm = philox4x32_10((uint32_t)r.x, gmine, key, img, rng_seed_lo, rng_seed_hi);
                // masks of my call's four runs; runs u0, u0+1 are mine, the other two the neighbour item's
                const DropPair w0 = dropout_run_windows(pm, 0), w1 = dropout_run_windows(pm, 1), w2 = dropout_run_windows(pm, 2), w3 = dropout_run_windows(pm, 3);
                const uint32_t k0x = keep2(w0.x), k0y = keep2(w0.y), k1x = keep2(w1.x), k1y = keep2(w1.y);
                const uint32_t k2x = keep2(w2.x), k2y = keep2(w2.y), k3x = keep2(w3.x), k3y = keep2(w3.y);
                const uint32_t mine0x = gq_odd ? k2x : k0x, mine0y = gq_odd ? k2y : k0y, mine1x = gq_odd ? k3x : k1x, mine1y = gq_odd ? k3y : k1y;
                uint32_t got0x = gq_odd ? k0x : k2x, got0y = gq_odd ? k0y : k2y, got1x = gq_odd ? k1x : k3x, got1y = gq_odd ? k1y : k3y;    // (what I give)
#if defined(__HIP_DEVICE_COMPILE__)
                // quad_perm [2,3,0,1]: the value of lane ^ 2 (executed by all lanes, outside any branch)
                got0x = (uint32_t)__builtin_amdgcn_mov_dpp((int)got0x, 0x4E, 0xF, 0xF, true); got0y = (uint32_t)__builtin_amdgcn_mov_dpp((int)got0y, 0x4E, 0xF, 0xF, true);
                got1x = (uint32_t)__builtin_amdgcn_mov_dpp((int)got1x, 0x4E, 0xF, 0xF, true); got1y = (uint32_t)__builtin_amdgcn_mov_dpp((int)got1y, 0x4E, 0xF, 0xF, true);
#endif
                // channels +0..3 and +8..11 of the item belong to the first call (A), +4..7 and +12..15 to the second (B)
                const uint32_t a0x = gq_odd ? got0x : mine0x, a0y = gq_odd ? got0y : mine0y, a1x = gq_odd ? got1x : mine1x, a1y = gq_odd ? got1y : mine1y;
                const uint32_t b0x = gq_odd ? mine0x : got0x, b0y = gq_odd ? mine0y : got0y, b1x = gq_odd ? mine1x : got1x, b1y = gq_odd ? mine1y : got1y;
                const uint4 oa = make_uint4(va.x & a0x, va.y & a0y, va.z & b0x, va.w & b0y);
                const uint4 ob = make_uint4(vb.x & a1x, vb.y & a1y, vb.z & b1x, vb.w & b1y);
                // d1: even lane keeps its first half, odd lane takes the even lane's second half (item it_e, bytes 0..15 | 16..31);
                // d2: odd lane keeps its second half, even lane takes the odd lane's first half (item it_o).  DPP quad_perm [1,0,3,2]
                // = the value of lane ^ 1.
                uint4 d1, d2;
#if defined(__HIP_DEVICE_COMPILE__)
                // (the exchange is executed by ALL lanes, outside any branch: a DPP read of an inactive lane returns 0)
                auto other = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); };
                const uint4 sa = make_uint4(other(oa.x), other(oa.y), other(oa.z), other(oa.w));
                const uint4 sb = make_uint4(other(ob.x), other(ob.y), other(ob.z), other(ob.w));
                d1.x = odd ? sb.x : oa.x; d1.y = odd ? sb.y : oa.y; d1.z = odd ? sb.z : oa.z; d1.w = odd ? sb.w : oa.w;
                d2.x = odd ? ob.x : sa.x; d2.y = odd ? ob.y : sa.y; d2.z = odd ? ob.z : sa.z; d2.w = odd ? ob.w : sa.w;
#else
                d1 = oa; d2 = ob;
#endif
                if (nt_out) {
                    if (off_e >= 0) __builtin_nontemporal_store(u32x4{d1.x, d1.y, d1.z, d1.w}, reinterpret_cast<u32x4*>(oe + (size_t)n * sample_stride));
                    if (off_o >= 0) __builtin_nontemporal_store(u32x4{d2.x, d2.y, d2.z, d2.w}, reinterpret_cast<u32x4*>(oo + (size_t)n * sample_stride));
                } else {
                    if (off_e >= 0) *reinterpret_cast<uint4*>(oe + (size_t)n * sample_stride) = d1;
                    if (off_o >= 0) *reinterpret_cast<uint4*>(oo + (size_t)n * sample_stride) = d2;
                }
            }
        }
        return;
    }
    // M16: the partner lane (l ^ 32) drew the calls of the other cout fragment of each pair: fetch its 128 keep bits once
    uint32_t ph_oth[4] = {0u, 0u, 0u, 0u};
    if constexpr (M16) {
        if (drop && ph_inloop) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ph_oth[k] = (uint32_t)__shfl_xor((int)ph_bits[k], 32, 64);
        }
    }
    // M16, decisions drawn in the loop: group g = fp*2 + (fc>>2) left the 16 keep bits of a call at bit (7 - g)*16 of the 128 -- drawn by
    // the lower lane half for the even fragment pairs (fc>>1 even), by the upper half for the odd ones (T0 / T1 below) -- and of that
    // call the lane's run is u = (fc&1)*2 + (q4>>1): bits 4u .. 4u+3.  Source and the (q4>>1) part of the shift are selected ONCE per
    // word here, so that the per-fragment expansion below is four constant-position bit extracts.
    uint32_t ph_En[4] = {0u, 0u, 0u, 0u}, ph_On[4] = {0u, 0u, 0u, 0u};
    if constexpr (M16) {
        if (drop && ph_inloop) {
            const bool up = (q4 >> 1) != 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t t0 = up ? ph_oth[k] : ph_bits[k], t1 = up ? ph_bits[k] : ph_oth[k];
                ph_En[k] = up ? t0 >> 4 : t0;
                ph_On[k] = up ? t1 >> 4 : t1;
            }
        }
    }
    for (int n = 0; n < fan; ++n) {
        if constexpr (M16) {
#pragma unroll
            for (int fp = 0; fp < FP16; ++fp) {
                const int pixl = wp * WTP + fp * 16 + l15;
                char* prow = smem + pixl * (BC * 2);
                const int2 rg = s_rng[pixl];
                const uint32_t img = rng_image_base + ((uint32_t)rg.y >> 16);
                const uint32_t sample = a.sample_base + (a.fan_count > 1 ? (uint32_t)n : ((uint32_t)rg.y & 0xFFFFu));
                Philox4 rr16{0u, 0u, 0u, 0u};
#pragma unroll
                for (int fc = 0; fc < FC16; ++fc) {
                    const int col = wc * WTC + fc * 16 + q4 * 4;
                    uint2 o = pk16[fc][fp];
                    if (drop && ph_inloop) {
                        // the call of fragment pair fc>>1 left its 16 bits at a fixed position; the lane's nibble (run (fc&1)*2 + (q4>>1)) sits
                        // at bit (fc&1)*8 of that field in ph_En (even pairs) / ph_On (odd pairs).  Bit r -> 16-bit lane r of the 4 packed channels.
                        const int bitpos = (7 - (fp * 2 + (fc >> 2))) * 16 + (fc & 1) * 8;
                        const int P = bitpos & 31;
                        const uint32_t srcw = ((fc >> 1) & 1) ? ph_On[bitpos >> 5] : ph_En[bitpos >> 5];
#if defined(__HIP_DEVICE_COMPILE__)
                        // (v_bfe_i32 of one bit = 0 / ~0; v_perm_b32 takes the low half of one and the high half of the other)
                        o.x &= __builtin_amdgcn_perm((uint32_t)__builtin_amdgcn_sbfe((int)srcw, P + 1, 1), (uint32_t)__builtin_amdgcn_sbfe((int)srcw, P, 1), 0x07060100u);
                        o.y &= __builtin_amdgcn_perm((uint32_t)__builtin_amdgcn_sbfe((int)srcw, P + 3, 1), (uint32_t)__builtin_amdgcn_sbfe((int)srcw, P + 2, 1), 0x07060100u);
#else
                        (void)P; (void)srcw;
#endif
                    } else if (drop) {
                        // decisions drawn here: the call of this lane's fragment pair (its partner lane draws the same one), run (fc&1)*2 + (q4>>1)
                        if ((fc & 1) == 0) {
                            if (ABL == 4) rr16 = Philox4{(uint32_t)col * 0x9E3779B9u, (uint32_t)rg.x * 0x85EBCA6Bu, sample * 0xC2B2AE35u, img};
                            else rr16 = philox4x32_10((uint32_t)rg.x, dropout_group16(bc0 + col), sample | ((uint32_t)G.layer_id << 16), img, rng_seed_lo, rng_seed_hi);
                        }
                        const DropPair dw = dropout_run_windows(rr16, (fc & 1) * 2 + (q4 >> 1));
                        o.x &= keep_mask_u16x2(dw.x, thr_m1_x2);
                        o.y &= keep_mask_u16x2(dw.y, thr_m1_x2);
                    }
                    *reinterpret_cast<uint2*>(prow + ((((col >> 3) ^ pixl) & (CPR - 1)) << 4) + (col & 7) * 2) = o;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < (M16 ? 0 : FP); ++j) {
            const int pixl = wp * WTP + j * 32 + frow;
            char* prow = smem + pixl * (BC * 2);
            const uint32_t img = rng_image_base + ((uint32_t)rng[j].y >> 16);
            const uint32_t sample = a.sample_base + (a.fan_count > 1 ? (uint32_t)n : ((uint32_t)rng[j].y & 0xFFFFu));
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                Philox4 rr{0u, 0u, 0u, 0u};
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int col = wc * WTC + i * 32 + g4 * 8 + fhalf * 4;
                    uint2 o = pk[i][j][g4];
                    if (drop) {
                        // dropout contract v3: one Philox call decides 16 channels -- the lane's four runs g4 = 0..3 of this 32-channel
                        // fragment share the call keyed by (col>>5, fhalf), run g4 reads output word g4 (philox.h)
                        if (g4 == 0) {
                            if (ABL == 4) rr = Philox4{(uint32_t)col * 0x9E3779B9u, (uint32_t)rng[j].x * 0x85EBCA6Bu, sample * 0xC2B2AE35u, img};
                            else rr = philox4x32_10((uint32_t)rng[j].x, dropout_group16(bc0 + col),
                                                    sample | ((uint32_t)G.layer_id << 16), img, rng_seed_lo, rng_seed_hi);
                        }
                        const DropPair dw = dropout_run_windows(rr, g4);
                        o.x &= keep_mask_u16x2(dw.x, thr_m1_x2);
                        o.y &= keep_mask_u16x2(dw.y, thr_m1_x2);
                    }
                    *reinterpret_cast<uint2*>(prow + ((((col >> 3) ^ pixl) & (CPR - 1)) << 4) + (col & 7) * 2) = o;
                }
                if (drop) __builtin_amdgcn_sched_barrier(0);   // keep the Philox chains from interleaving (registers)
            }
        }
        phase_stamp<ABL>(tstamp, 3);        // Philox mask + LDS tile writes
        __syncthreads();
        phase_stamp<ABL>(tstamp, 4);        // barrier
        if constexpr (ABL == 10) {
            // ---- bottleneck chain (ConvGroup.ch_*): the tile in LDS is t2 = relu(2b) [128 pixels][CM channels].  Per pass of 128 output
            // channels: the shortcut's pixels come in by LDS-DMA into the pass tile Y [128][128] while the wave multiplies its 32 pixels
            // with the 2c weights (A fragments straight from global memory: L2-resident, 32-128 KB per layer); relu(acc + bias +
            // shortcut) is written over the shortcut in place, the pass goes out as 16-byte row pieces, and -- on the same bf16 values,
            // read back from the wave's own rows -- the next block's 2a accumulates its partial sums over this pass's 128 channels.
            // Same MFMA shape, k order and epilogue arithmetic as the separate 2c / 2a launches: bit-identical planes.
            static_assert(!XR && !SPLIT && BP == 128 && (BC == 64 || BC == 128) && THREADS == 256, "chain epilogue: 64x128 / 128x128 tiles");
            constexpr int CM = BC, CPRM = CM / 8, T2B = BP * CM * 2, KS2 = CM / 16, F3 = CM / 32;
            static_assert(T2B + 128 * 256 <= Cfg::MAIN, "chain epilogue does not fit the staging area");
            char* Yt = smem + T2B;
            const int C2 = G.ch_c2, npass = C2 / 128;
            const int mypix = wave * 32 + frow;
            bf16x8 b2[KS2];
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
                b2[ks] = *reinterpret_cast<const bf16x8*>(smem + mypix * (CM * 2) + ((((ks * 2 + fhalf) ^ mypix) & (CPRM - 1)) << 4));
            const bool has3 = G.ch_w3 != nullptr;
            f32x16 acc3[F3];
#pragma unroll
            for (int f = 0; f < F3; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[f][r] = 0.f;
            const uint16_t* w2 = reinterpret_cast<const uint16_t*>(G.ch_w2);
            const uint16_t* w3 = reinterpret_cast<const uint16_t*>(G.ch_w3);
            const char* resb = reinterpret_cast<const char*>(G.ch_res);
            uint16_t* outb = reinterpret_cast<uint16_t*>(G.ch_out);
            for (int q = 0; q < npass; ++q) {
                // shortcut pixels of this pass -> Y (piece (pixel, physical chunk pc) holds logical chunk pc ^ pixel)
#pragma unroll 2
                for (int it = 0; it < 8; ++it) {
                    const int qi = it * THREADS + tid, pixl = qi >> 4, pc = qi & 15, cp = (pc ^ pixl) & 15;
                    const int off = s_off[pixl] < 0 ? 0 : s_off[pixl];
                    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(resb + ((size_t)off * C2 + q * 128 + cp * 8) * 2),
                                                     LDS_PTR(Yt + (it * THREADS + wave * 64) * 16), 16, 0, 0);
                }
                // one 32-channel fragment at a time (16 accumulator registers live: the kernel keeps the base build's residency); the
                // first fragment's MFMAs run while the shortcut pieces are in flight
                auto gemm2 = [&](int f) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS2; ++ks) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(w2 + (size_t)(q * 128 + f * 32 + frow) * CM + ks * 16 + fhalf * 8);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b2[ks], acc, 0, 0, 0);
                    }
                    return acc;
                };
                auto finish = [&](int f, const f32x16& acc) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int col = f * 32 + g4 * 8 + fhalf * 4;
                        char* pp = Yt + mypix * 256 + ((((col >> 3) ^ mypix) & 15) << 4) + (col & 7) * 2;
                        const uint2 r = *reinterpret_cast<const uint2*>(pp);
                        const float4 bv = *reinterpret_cast<const float4*>(G.ch_b2 + q * 128 + col);
                        // (the separate 2c launch: fma(acc, 1, bias), += shortcut * 1, ReLU, round)
                        float v0 = __builtin_fmaf(acc[g4 * 4 + 0], 1.0f, bv.x), v1 = __builtin_fmaf(acc[g4 * 4 + 1], 1.0f, bv.y);
                        float v2 = __builtin_fmaf(acc[g4 * 4 + 2], 1.0f, bv.z), v3 = __builtin_fmaf(acc[g4 * 4 + 3], 1.0f, bv.w);
                        v0 += bf16_to_f32(r.x & 0xFFFFu) * 1.0f; v1 += bf16_to_f32(r.x >> 16) * 1.0f;
                        v2 += bf16_to_f32(r.y & 0xFFFFu) * 1.0f; v3 += bf16_to_f32(r.y >> 16) * 1.0f;
                        uint2 o;
                        o.x = relu_bf16x2_pk(pack_bf16x2(v0, v1)); o.y = relu_bf16x2_pk(pack_bf16x2(v2, v3));
                        *reinterpret_cast<uint2*>(pp) = o;
                    }
                };
                // (requesting the next 2a's 16 weight fragments of the pass here, ahead of their use, needs two workgroups per CU instead
                // of three -- 64 more registers -- and measured 0.7 ms slower per 256 frames than fetching them three at a time)
                constexpr bool PRE3 = false;
                bf16x8 a3[1];
                {
                    const f32x16 a0 = gemm2(0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();                      // every wave's shortcut pieces have landed
                    finish(0, a0);
                }
#pragma unroll
                for (int f = 1; f < 4; ++f) { const f32x16 af_ = gemm2(f); finish(f, af_); }
                if (has3) {                               // next block's 2a on this pass's 128 channels (own rows: no barrier needed)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const bf16x8 yb = *reinterpret_cast<const bf16x8*>(Yt + mypix * 256 + ((((ks * 2 + fhalf) ^ mypix) & 15) << 4));
#pragma unroll
                        for (int f = 0; f < F3; ++f) {
                            bf16x8 af;
                            if constexpr (PRE3) af = a3[ks * 2 + f];
                            else af = *reinterpret_cast<const bf16x8*>(w3 + (size_t)(f * 32 + frow) * C2 + q * 128 + ks * 16 + fhalf * 8);
                            acc3[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, yb, acc3[f], 0, 0, 0);
                        }
                    }
                }
                __syncthreads();                          // the pass tile is complete
#pragma unroll 2
                for (int it = 0; it < 8; ++it) {
                    const int qi = it * THREADS + tid, pixl = qi >> 4, pc = qi & 15, cp = (pc ^ pixl) & 15;
                    const int off = s_off[pixl];
                    if (off < 0) continue;
                    *reinterpret_cast<uint4*>(outb + (size_t)off * C2 + q * 128 + cp * 8) = *reinterpret_cast<const uint4*>(Yt + qi * 16);
                }
                __syncthreads();                          // ... and read: the next pass may overwrite it
            }
            if (has3) {
                // t1' = relu(acc3 + bias) -> bf16 -> the t2 region (every wave read its t2 fragments long ago) -> row pieces
#pragma unroll
                for (int f = 0; f < F3; ++f)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int col = f * 32 + g4 * 8 + fhalf * 4;
                        const float4 bv = *reinterpret_cast<const float4*>(G.ch_b3 + col);
                        uint2 o;
                        o.x = relu_bf16x2_pk(pack_bf16x2(__builtin_fmaf(acc3[f][g4 * 4 + 0], 1.0f, bv.x), __builtin_fmaf(acc3[f][g4 * 4 + 1], 1.0f, bv.y)));
                        o.y = relu_bf16x2_pk(pack_bf16x2(__builtin_fmaf(acc3[f][g4 * 4 + 2], 1.0f, bv.z), __builtin_fmaf(acc3[f][g4 * 4 + 3], 1.0f, bv.w)));
                        *reinterpret_cast<uint2*>(smem + mypix * (CM * 2) + ((((col >> 3) ^ mypix) & (CPRM - 1)) << 4) + (col & 7) * 2) = o;
                    }
                __syncthreads();
                uint16_t* out3 = reinterpret_cast<uint16_t*>(G.ch_out3);
#pragma unroll
                for (int it = 0; it < BP * CPRM / THREADS; ++it) {
                    const int qi = it * THREADS + tid, pixl = qi / CPRM, pc = qi % CPRM, cp = (pc ^ pixl) & (CPRM - 1);
                    const int off = s_off[pixl];
                    if (off < 0) continue;
                    *reinterpret_cast<uint4*>(out3 + (size_t)off * CM + cp * 8) = *reinterpret_cast<const uint4*>(smem + qi * 16);
                }
            }
            return;
        }
        if (fuse && G.agg_kind != AGG_NONE) {
            // ---- fused 1x1 + MC aggregation: all of a wave's output fragments stay in registers until every wave has read
            // the activation tile; its LDS then becomes the fp32 output tile [row][ystride] the reduction walks
            f32x16 yy[HALFP];
            if (fuse_active) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) w2f[ks] = *reinterpret_cast<const bf16x8*>(wp2 + ks * 16);
#pragma unroll
                for (int h4 = 0; h4 < HALFP; ++h4) {
                    const int pixl = ((wave & 1) * HALFP + h4) * 32 + frow;
                    const char* prow = smem + pixl * (BC * 2);
#pragma unroll
                    for (int r = 0; r < 16; ++r) yy[h4][r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const bf16x8 xb = *reinterpret_cast<const bf16x8*>(prow + ((((ks * 2 + fhalf) ^ pixl) & (CPR - 1)) << 4));
                        yy[h4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[ks], xb, yy[h4], 0, 0, 0);
                    }
                }
            }
            phase_stamp<ABL>(tstamp, 10);       // (aggregating tiles) fused 1x1 MFMAs
            __syncthreads();
            const int ystride = ((G.cout2 + 31) & ~31) + 4;            // floats per row: = 4 mod 32, conflict-free 16-byte writes
            float* ytile = reinterpret_cast<float*>(smem);
            if (fuse_active) {
#pragma unroll
                for (int h4 = 0; h4 < HALFP; ++h4) {
                    const int row = ((wave & 1) * HALFP + h4) * 32 + frow;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int co2 = f2 * 32 + g4 * 8 + fhalf * 4;
                        if (co2 >= G.cout2) continue;
                        const float4 b2 = *reinterpret_cast<const float4*>(G.bias2 + co2);
                        *reinterpret_cast<float4*>(ytile + (size_t)row * ystride + co2) =
                            make_float4(yy[h4][g4 * 4 + 0] + b2.x, yy[h4][g4 * 4 + 1] + b2.y, yy[h4][g4 * 4 + 2] + b2.z, yy[h4][g4 * 4 + 3] + b2.w);
                    }
                }
            }
            __syncthreads();
            phase_stamp<ABL>(tstamp, 11);       // barrier + fp32 output tile to LDS + barrier
            if (G.agg_kind == AGG_CLS) {
                if (G.agg_C == 8) agg_reduce_cls<8>(G, ytile, ystride, s_off, s_off2, tid, THREADS, BP);
                else agg_reduce_cls<4>(G, ytile, ystride, s_off, s_off2, tid, THREADS, BP);
            } else if (G.agg_kind == AGG_BOX) {
                agg_reduce_box(G, ytile, ystride, s_off, s_off2, tid, THREADS, BP);
            } else {
                agg_reduce_cov(G, ytile, ystride, s_off, s_off2, tid, THREADS, BP);
            }
            phase_stamp<ABL>(tstamp, 12);       // MC reduction over the samples + statistics stores
            if constexpr (ABL == 90) { if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[14], __builtin_amdgcn_s_memrealtime() - treal); }
            return;                                   // (fused groups never fan out: one pass, and `pk` dies here)
        }
        if (fuse) {
            if (fuse_active) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) w2f[ks] = *reinterpret_cast<const bf16x8*>(wp2 + ks * 16);
#pragma unroll 1
                for (int pf = (wave & 1) * HALFP; pf < (wave & 1) * HALFP + HALFP; ++pf) {
                    const int pixl = pf * 32 + frow;
                    const char* prow = smem + pixl * (BC * 2);
                    f32x16 y;
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const bf16x8 xb = *reinterpret_cast<const bf16x8*>(prow + ((((ks * 2 + fhalf) ^ pixl) & (CPR - 1)) << 4));
                        y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[ks], xb, y, 0, 0, 0);
                    }
                    if (s_off[pixl] < 0) continue;
                    float* orow = G.out2 + (size_t)s_off2[pixl] * G.out2_cstride;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int co2 = f2 * 32 + g4 * 8 + fhalf * 4;
                        if (co2 >= G.cout2) continue;
                        const float4 b2 = *reinterpret_cast<const float4*>(G.bias2 + co2);
                        const float v0 = y[g4 * 4 + 0] + b2.x, v1 = y[g4 * 4 + 1] + b2.y;
                        const float v2 = y[g4 * 4 + 2] + b2.z, v3 = y[g4 * 4 + 3] + b2.w;
                        if (co2 + 1 < G.cout2) *reinterpret_cast<float2*>(orow + co2) = make_float2(v0, v1);
                        else orow[co2] = v0;
                        if (co2 + 3 < G.cout2) *reinterpret_cast<float2*>(orow + co2 + 2) = make_float2(v2, v3);
                        else if (co2 + 2 < G.cout2) orow[co2 + 2] = v2;
                    }
                }
            }
            return;                                   // the tile itself has no other consumer (fused groups never fan out)
        }
        if constexpr (M16) {
            // Tower launches (row reuse, one pass, plain stores): all sixteen 16-byte pieces of a thread are read from the LDS tile
            // up front -- the accumulators are dead, there are registers for them -- and leave as buffer stores against the tile's
            // first extended input row (planes of equal geometry: no output pixel of the tile lies below it), so a piece costs one
            // 32-bit offset instead of a 64-bit address, and an invalid slot is an out-of-range offset the hardware drops instead of
            // a branch.  (The rolled loop read the tile four pieces at a time, each batch behind its own LDS round trip.)
            if (fan == 1 && !nt_out && ABL != 30 && a.fan_count <= 1) {
#if defined(__HIP_DEVICE_COMPILE__)
                const int prow0 = tid >> 5, cp = tid & 31;
                const int first = __builtin_amdgcn_readfirstlane(a.ext[(size_t)bx * XR_EXT_ROWS].x);
                char* obase = reinterpret_cast<char*>(G.out) + ((size_t)first * a.out_cstride + bc0) * 2;
                obase = reinterpret_cast<char*>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)(uintptr_t)obase >> 32)) << 32) |
                                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)obase)));
                const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0x7FFFFFFF, 0x00020000);      // raw, 2 GiB window
                int offs[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) offs[k] = s_off[prow0 + 16 * k];
                u32x4 v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const u32x4*>(smem + (prow0 + 16 * k) * (BC * 2) + cp * 16);
                const int rowb = a.out_cstride * 2;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int c16 = (cp ^ (prow0 + 16 * k)) & (CPR - 1);
                    const uint32_t vo = offs[k] < 0 ? 0xFFFFFFF0u : (uint32_t)(offs[k] - first) * (uint32_t)rowb + (uint32_t)(c16 * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(v[k], orsrc, (int)vo, 0, 0);
                }
#endif
                phase_stamp<ABL>(tstamp, 5);
                if constexpr (ABL == 90) { if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[14], __builtin_amdgcn_s_memrealtime() - treal); }
                return;
            }
        }
#pragma unroll 4
        for (int q = tid; q < BP * CPR; q += THREADS) {
            const int pixl = q / CPR, cp = q % CPR;
            const int off = s_off[pixl];
            if (off < 0) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(smem + pixl * (BC * 2) + cp * 16);
            const int c16 = (cp ^ pixl) & (CPR - 1);
            const size_t e = ((size_t)off + (size_t)n * a.fan_stride) * a.out_cstride + bc0 + c16 * 8;
            if (ABL == 30) { if (v.x == 0x12345678u && e == 0) *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.out)) = v; continue; }
            // (variant 86, BOD_NT_STORES=1: non-temporal stores for the generic kernel's outputs -- A/B)
            if (nt_out) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(G.out) + e));
            else *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.out) + e) = v;
            if (CAN_RES && G.out_relu) {
                uint4 r;
                r.x = relu_bf16x2(v.x); r.y = relu_bf16x2(v.y); r.z = relu_bf16x2(v.z); r.w = relu_bf16x2(v.w);
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.out_relu) + e) = r;
            }
        }
        phase_stamp<ABL>(tstamp, 5);        // store loop (issue)
        if constexpr (ABL == 90) { if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[14], __builtin_amdgcn_s_memrealtime() - treal); }   // 100 MHz ticks
        if (n + 1 < fan) __syncthreads();
    }
}

// One tile per workgroup.  XCD-aware tile order: block b runs on XCD b%8; give every XCD a contiguous
// range of pixel tiles so that neighbouring tiles (which share their 3x3 halo rows) share an L2.
// (the bottleneck-chain build, ABL 10, is memory-bound and must keep the base kernel's residency: 3 / 2 workgroups per CU at 64 / 128 couts)
template <int BC, int BP, int WC, int WP, int ABL, bool XR, bool SPLIT = false>
__global__ __launch_bounds__(64 * WC * WP, (ABL == 10 ? (BC == 64 ? 3 : 2) : 1)) void conv_igemm_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bx = blockIdx.x, by = blockIdx.y, gz = blockIdx.z;
    const int ny_tiles = a.cout_pad / BC;
    if (!XR && gridDim.y == 1 && ny_tiles > 1) {
        // Several cout tiles per pixel tile, launched as ONE grid row with the cout tile as the fast index INSIDE an XCD (round 4): the
        // ny workgroups of a pixel tile run side by side on one XCD and read its activation rows from HBM once, through that L2.  The
        // (nx, ny) grid ran a whole plane of pixel tiles per cout tile: stage 5's 512 -> 2048 expansions re-read their input eight
        // times (tests/tools/op_table.py: 2.2 TB/s algorithmic at 540 us for 263 us of bytes).  Workgroup b: XCD b % 8, slot b / 8 ->
        // cout tile slot % ny, the XCD's (slot / ny)-th pixel tile; XCDs with one pixel tile less retire their last ny slots at once.
        const int nx = (a.M + BP - 1) / BP, q = nx >> 3, r = nx & 7, xcd = bx & 7, slot = bx >> 3;
        const int pj = slot / ny_tiles;
        by = slot - pj * ny_tiles;
        if (pj >= q + (xcd < r ? 1 : 0)) return;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pj;
    } else {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, idx = bx >> 3;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // Groups that READ THE SAME INPUT (the first tower layer: three heads on one pyramid) are launched interleaved in x --
    // grid (tiles * groups, ny, 1), work item = tile * groups + group -- so that a pixel tile's three workgroups run back to
    // back on one XCD and the second and third find the pyramid rows in that XCD's L2 (grid.z-major order re-read the
    // 0.78 GB pyramid from HBM once per head: 3.2 GB fetched for 0.72 GB algorithmic, profiles/round2_head_conv_pmc.json)
    if (gridDim.z == 1 && a.groups > 1 && a.ksplit <= 1) {
        if (a.fan_chunk > 0) {
            const int nxt = (a.M + BP - 1) / BP, T = a.fan_chunk, per = T * a.groups;
            const int c = bx / per, w = bx - c * per, here = min(T, nxt - c * T);
            gz = w / here; bx = c * T + (w - gz * here);
        } else { gz = bx % a.groups; bx = bx / a.groups; }
    }
    if constexpr (ABL == 5) {
        // De-phase the CUs of the fan-out launch.  Every CU holds one workgroup, all tiles take the same time, and a launch starts
        // all CUs together: the epilogues -- ten masked copies of the tile, 1.3 MB per CU -- would all store at the same moment
        // (8 TB/s demanded for 40 us, then nothing for 70 us) and back up into the CUs' store queues, behind which the Philox
        // draws of the next samples wait.  The FIRST workgroup of a CU (blockIdx < n_cu: workgroup b starts on CU slot b / 8 of
        // XCD b % 8) sleeps (slot & 3) quarter-tiles, after which the four phases keep their distance for the whole launch.
        if (a.stagger_ticks > 0 && (int)blockIdx.x < (a.n_cu > 0 ? a.n_cu : 256) && blockIdx.y == 0 && blockIdx.z == 0) {
            const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 3) * (unsigned long long)a.stagger_ticks;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(64);
        }
    }
    conv_tile<BC, BP, WC, WP, ABL, XR, SPLIT>(a, gz, bx, by, smem);
}

// f16mx precision (header of this file): the head towers' launches under their own symbol.  MXK 1: hx rows in -- per multiplication one
// f16 product + half a block-scaled e2m3 product; MXK 2: (hi, lo) bf16 pairs in (the bf16x3 loop: the first tower layer reads the
// pyramid).  Either writes hx rows or (hi, lo) pairs per group (ConvGroup.out_hx), fans out, fuses the 1x1 + aggregation.
template <int MXK>
__global__ __launch_bounds__(512, 1) void conv_igemm_mx_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bx = blockIdx.x;
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, idx = bx >> 3;      // XCD x owns a contiguous range of pixel tiles
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    unsigned long long t0 = 0;
    if constexpr (MXK == 11 || MXK == 12) t0 = __builtin_amdgcn_s_memtime();
    conv_tile<256, 256, 2, 4, 0, true, true, MXK>(a, blockIdx.z, bx, 0, smem);
    if constexpr (MXK == 11 || MXK == 12) { if (threadIdx.x == 0) atomicAdd(&g_phase_cycles[9], __builtin_amdgcn_s_memtime() - t0); }      // whole tile incl. epilogue
}

template <int MXK>
static hipError_t launch_mx(const ConvArgs& a, hipStream_t s) {
    using Cfg = ConvCfg<256, 256, 2, 4, true>;
    constexpr int LDS = Cfg::LDS + (MXK == 3 ? 2 * 2048 + 2 * XR_EXT_ROWS * 8 : 0);          // f16mx4: + the two scale arrays, double-buffered
    static_assert(LDS <= 160 * 1024, "LDS of a compute unit");
    static PerDeviceOnce once;
    bool& attr_set = *once.slot();
    auto kern = conv_igemm_mx_kernel<MXK>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((a.M + 255) / 256, 1, a.groups), dim3(Cfg::THREADS), LDS, s, a);
    return hipGetLastError();
}

// f16mx: (hi, lo) bf16 pair rows -> hx rows (header of this file), one thread per (pixel, 16-channel slot): the pyramid the generic FPN
// kernels wrote as pairs becomes the first tower layer's input, so that layer too runs on the f16 + MX-fp6 loop.  The slot's channels
// 64q + 32m + 8 g4 + 4b + r are four runs of four: four 8-byte reads of hi and of lo, four 8-byte H-chunk writes, two 16-byte slot pieces.
__global__ __launch_bounds__(256) void pairs_to_hx_kernel(const uint16_t* __restrict__ in, uint8_t* __restrict__ out, const long npix, const int C) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int spp = C / 16;                       // slots per pixel
    if (t >= npix * spp) return;
    const long pix = t / spp;
    const int slot = (int)(t - pix * spp), q = slot >> 2, m = (slot >> 1) & 1, b = slot & 1;
    const uint16_t* src = in + pix * 2 * C + (2 * q + m) * 64 + 4 * b;      // 32-channel block (2q + m): 32 hi then 32 lo
    uint8_t* dst = out + pix * 4 * C + q * 256;
    f32x16v hv, lv;
    float mx = 6.103515625e-05f;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const uint2 h2 = *reinterpret_cast<const uint2*>(src + 8 * g4), l2 = *reinterpret_cast<const uint2*>(src + 32 + 8 * g4);
        float v[4] = {bf16_to_f32(h2.x & 0xFFFFu) + bf16_to_f32(l2.x & 0xFFFFu), bf16_to_f32(h2.x >> 16) + bf16_to_f32(l2.x >> 16),
                      bf16_to_f32(h2.y & 0xFFFFu) + bf16_to_f32(l2.y & 0xFFFFu), bf16_to_f32(h2.y >> 16) + bf16_to_f32(l2.y >> 16)};
        uint16_t hb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = __builtin_amdgcn_fmed3f(v[r], -65504.0f, 65504.0f);
            const _Float16 hh = (_Float16)v[r];
            hb[r] = __builtin_bit_cast(uint16_t, hh);
            const float hf = (float)hh;
            hv[g4 * 4 + r] = hf;
            lv[g4 * 4 + r] = (v[r] - hf) * 2048.0f;
            mx = fmaxf(mx, fabsf(v[r]));
        }
        *reinterpret_cast<uint2*>(dst + 2 * (32 * m + 8 * g4 + 4 * b)) = make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
    }
    const uint32_t eb = (__float_as_uint(mx * 1.0666667f) >> 23) - 2u;
    const i32x6 pk = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(hv, lv, __uint_as_float(eb << 23));
    *reinterpret_cast<uint4*>(dst + 128 + 64 * m + 16 * b) = make_uint4((uint32_t)pk[0], (uint32_t)pk[1], (uint32_t)pk[2], (uint32_t)pk[3]);
    *reinterpret_cast<uint4*>(dst + 128 + 64 * m + 32 + 16 * b) = make_uint4((uint32_t)pk[4], (uint32_t)pk[5], 0u, eb);
}

// f16mx4: (hi, lo) bf16 pair rows -> h4 rows (engine.hip, pack_h4_row), one thread per (pixel, MX block): block (x, ks, b) = channels
// 128x + 32 ks + 8 g4 + 4b + r -- the same four runs of four as above.
__global__ __launch_bounds__(256) void pairs_to_h4_kernel(const uint16_t* __restrict__ in, uint8_t* __restrict__ out, const long npix) {
    constexpr int C = 256;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= npix * 16) return;
    const long pix = t >> 4;
    const int slot = (int)(t & 15), x = slot >> 3, ks = (slot >> 1) & 3, b = slot & 1;
    const uint16_t* src = in + pix * 2 * C + (4 * x + ks) * 64 + 4 * b;      // 32-channel block 4x + ks: 32 hi then 32 lo
    uint8_t* dst = out + pix * 4 * C;
    const int q = 2 * x + (ks >> 1);
    uint8_t* H = dst + (3 * (q >> 1) + (q & 1)) * 128 + 2 * ((32 * ks) & 63);
    float hv[16], lv[16];
    float mx = 6.103515625e-05f;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const uint2 h2 = *reinterpret_cast<const uint2*>(src + 8 * g4), l2 = *reinterpret_cast<const uint2*>(src + 32 + 8 * g4);
        float v[4] = {bf16_to_f32(h2.x & 0xFFFFu) + bf16_to_f32(l2.x & 0xFFFFu), bf16_to_f32(h2.x >> 16) + bf16_to_f32(l2.x >> 16),
                      bf16_to_f32(h2.y & 0xFFFFu) + bf16_to_f32(l2.y & 0xFFFFu), bf16_to_f32(h2.y >> 16) + bf16_to_f32(l2.y >> 16)};
        uint16_t hb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = __builtin_amdgcn_fmed3f(v[r], -65504.0f, 65504.0f);
            const _Float16 hh = (_Float16)v[r];
            hb[r] = __builtin_bit_cast(uint16_t, hh);
            const float hf = (float)hh;
            hv[g4 * 4 + r] = hf;
            lv[g4 * 4 + r] = (v[r] - hf) * 2048.0f;
            mx = fmaxf(mx, fabsf(hf));
        }
        *reinterpret_cast<uint2*>(H + 2 * (8 * g4 + 4 * b)) = make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
    }
    const uint32_t eb = (__float_as_uint(mx * 1.3333334f) >> 23) - 2u;
    const float sc = __uint_as_float(eb << 23);
    uint32_t pk[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t w_ = 0u;
        w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w_, hv[4 * d + 0], lv[4 * d + 0], sc, 0);
        w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w_, hv[4 * d + 1], lv[4 * d + 1], sc, 1);
        w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w_, hv[4 * d + 2], lv[4 * d + 2], sc, 2);
        w_ = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w_, hv[4 * d + 3], lv[4 * d + 3], sc, 3);
        pk[d] = w_;
    }
    *reinterpret_cast<uint4*>(dst + (3 * x + 2) * 128 + (2 * ks + b) * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    dst[6 * 128 + 8 * x + 4 * b + ks] = (uint8_t)eb;
}

hipError_t launch_pairs_to_hx(const void* in, void* out, long npix, int C, hipStream_t s, int fmt) {
    if (C % 64 != 0 || npix <= 0) return hipErrorInvalidValue;
    if (fmt == 2) {
        if (C != 256) return hipErrorInvalidValue;
        hipLaunchKernelGGL(pairs_to_h4_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(in),
                           reinterpret_cast<uint8_t*>(out), npix);
        return hipGetLastError();
    }
    const long total = npix * (C / 16);
    hipLaunchKernelGGL(pairs_to_hx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(in),
                       reinterpret_cast<uint8_t*>(out), npix, C);
    return hipGetLastError();
}

// Persistent form of the row-reuse kernel: one workgroup per CU walks a contiguous range of (head, pixel tile) work
// items, XCD x owning a contiguous eighth of them (neighbouring tiles share halo rows in that XCD's L2).  Removes the
// workgroup retire / dispatch gap between tiles (measured with the phase clock: tiles cover 1.26-1.30 ms of a
// 1.35-1.38 ms launch per CU).
template <int ABL>
__global__ __launch_bounds__(512) void conv_igemm_xr_persistent_kernel(const ConvArgs a, const int nx, const int total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nb = gridDim.x;                    // multiple of 8
    const int v = __builtin_amdgcn_readfirstlane((blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3));
    const int per = total / nb, extra = total - per * nb;
    int t = __builtin_amdgcn_readfirstlane(v * per + (v < extra ? v : extra));
    const int t1 = __builtin_amdgcn_readfirstlane(t + per + (v < extra ? 1 : 0));
    for (; t < t1; ++t) {
        const int z = t / nx;
        conv_tile<256, 256, 2, 4, ABL, true>(a, z, t - z * nx, 0, smem);
        __syncthreads();                         // LDS (epilogue tile + metadata) is reused by the next tile
    }
}

template <int ABL>
static hipError_t launch_xr_persistent(const ConvArgs& a, hipStream_t s) {
    using Cfg = ConvCfg<256, 256, 2, 4, true>;
    static PerDeviceOnce once;
    bool& attr_set = *once.slot();
    static int n_cu = 0;
    auto kern = conv_igemm_xr_persistent_kernel<ABL>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        n_cu = prop.multiProcessorCount & ~7;
        attr_set = true;
    }
    const int nx = (a.M + 255) / 256, total = nx * a.groups;
    const int nb = total < n_cu ? ((total + 7) & ~7) : n_cu;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(512), Cfg::LDS, s, a, nx, total);
    return hipGetLastError();
}

template <int BC, int BP, int WC, int WP, int ABL, bool XR = false, bool SPLIT = false>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
    using Cfg = ConvCfg<BC, BP, WC, WP, XR>;
    static PerDeviceOnce once;
    bool& attr_set = *once.slot();
    const int nx = (a.M + BP - 1) / BP, ny = a.cout_pad / BC;
    auto kern = conv_igemm_kernel<BC, BP, WC, WP, ABL, XR, SPLIT>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(nx, ny, a.groups * (a.ksplit > 1 ? a.ksplit : 1));
    if (ABL == 5 && a.groups > 1 && a.ksplit <= 1) grid = dim3(nx * a.groups, ny, 1);      // shared-input groups, interleaved (see the kernel)
    // cout tile as the fast index inside an XCD (see the kernel).  BOD_COUT_INNER=0: the (nx, ny) grid, A/B aid -- same tiles, same results
    static const bool cout_inner = [] { const char* e = getenv("BOD_COUT_INNER"); return !e || atoi(e) != 0; }();
    // (from 64 pixel tiles on: with fewer, the (nx, ny) grid spreads a pixel tile's cout tiles over all XCDs, which small launches need more)
    if (cout_inner && !XR && ny > 1 && nx >= 64) grid = dim3(8 * ((nx >> 3) + ((nx & 7) ? 1 : 0)) * ny, 1, grid.z);
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::THREADS), Cfg::LDS, s, a);
    return hipGetLastError();
}

// Split-K reduce: out[m][co] = act(sum_s partial[s][m][co] + bias[co] (+ residual)) -> bf16 through the row table.
// One thread per 8 output channels (16-byte stores); the same fp32 operation order for every split count.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvArgs a) {
    const int c8 = a.cout_pad / 8;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const int grp = blockIdx.y;
    if (q >= (long)a.M * c8) return;
    const int m = (int)(q / c8), co = (int)(q % c8) * 8;
    if (co >= a.cout_valid) return;
    const ConvGroup& G = a.g[grp];
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.f;
    for (int s = 0; s < a.ksplit; ++s) {
        const float* p = a.partial + ((size_t)(grp * a.ksplit + s) * a.M + m) * a.cout_pad + co;
        const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
        v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w; v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
    }
    const RowEnt e = a.rows[m];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += G.bias[co + k];
    const int slot = a.split ? (co >> 5) * 64 + (co & 31) : co;      // bf16x3: hi half of the (hi, lo) pair, lo 32 slots on
    if (G.res) {
        const uint16_t* rp = reinterpret_cast<const uint16_t*>(G.res) + (size_t)e.res_off * a.res_cstride + slot;
        const uint4 r = *reinterpret_cast<const uint4*>(rp);
        const uint32_t rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] += bf16_to_f32(rw[k] & 0xFFFFu); v[2 * k + 1] += bf16_to_f32(rw[k] >> 16); }
        if (a.split) {
            const uint4 rl = *reinterpret_cast<const uint4*>(rp + 32);
            const uint32_t lw[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[2 * k] += bf16_to_f32(lw[k] & 0xFFFFu); v[2 * k + 1] += bf16_to_f32(lw[k] >> 16); }
        }
    }
    if (a.flags & CONV_RELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    if (a.flags & CONV_OUT_F32) {                    // fp32 result (weight gradients): per-channel validity
        float* o32 = reinterpret_cast<float*>(G.out) + (size_t)e.out_off * a.out_cstride + co;
        if (a.flags & CONV_ACCUM) {                  // input gradients of the training step accumulate in place
#pragma unroll
            for (int k = 0; k < 8; ++k) if (co + k < a.cout_valid) o32[k] += v[k];
            return;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (co + k < a.cout_valid) o32[k] = v[k];
        return;
    }
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    const size_t off = (size_t)e.out_off * a.out_cstride + slot;
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.out) + off) = o;
    if (a.split) {
        const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
        uint4 l;
        uint32_t* lw = &l.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) lw[k] = pack_bf16x2(v[2 * k] - bf16_to_f32(ow[k] & 0xFFFFu), v[2 * k + 1] - bf16_to_f32(ow[k] >> 16));
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.out) + off + 32) = l;
        if (G.out_relu) {
            auto keep = [](uint32_t h) { return ~(((h >> 15) & 0x00010001u) * 0xFFFFu); };
            const uint32_t k0 = keep(o.x), k1 = keep(o.y), k2 = keep(o.z), k3 = keep(o.w);
            uint16_t* o2 = reinterpret_cast<uint16_t*>(G.out_relu);
            *reinterpret_cast<uint4*>(o2 + off) = make_uint4(o.x & k0, o.y & k1, o.z & k2, o.w & k3);
            *reinterpret_cast<uint4*>(o2 + off + 32) = make_uint4(l.x & k0, l.y & k1, l.z & k2, l.w & k3);
        }
        return;
    }
    if (G.out_relu) {
        uint4 r;
        r.x = relu_bf16x2(o.x); r.y = relu_bf16x2(o.y); r.z = relu_bf16x2(o.z); r.w = relu_bf16x2(o.w);
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(G.out_relu) + off) = r;
    }
}

// 256x256 tiles (one 8-wave workgroup per CU) pay once there are at least ~1.5 workgroups per CU; below that
// half the chip idles and the 128x128 configuration (two workgroups per CU, 4x the tiles) is 20-25 % faster
// (measured on the stage-4 / FPN layers: 128 big tiles for 256 CUs).
// The N-way dropout fan-out layer takes the big tile too (from 1 024 tiles on) since its epilogue keeps the tile in
// registers across the samples (1.9-2.0 ms on 256x256 tiles vs 2.2 ms on 128x128 at 64 frames; the older epilogue, an
// LDS round trip and two barriers per sample, preferred two co-resident 128x128 workgroups).  BOD_FAN_SMALL=1: A/B aid.
static bool fan_out_small_tile() {
    static const bool small = getenv("BOD_FAN_SMALL") && atoi(getenv("BOD_FAN_SMALL")) == 1;
    return small;
}

static bool conv_big_tile_pays(const ConvArgs& a) {
    // fp32 accumulate-in-place launches (training: input gradients) take the big tile from 96 tiles on (measured on the
    // training step, batch 8 and 32); other fp32-output launches never do.  BOD_F32_BIG_TILES=0 disables, =n moves the threshold
    static const int f32_min_tiles = getenv("BOD_F32_BIG_TILES") ? atoi(getenv("BOD_F32_BIG_TILES")) : 96;
    if (a.cout_pad % 256 != 0) return false;
    static const bool old_rule = getenv("BOD_TILE_RULE_OLD") != nullptr;          // A/B aid
    if (old_rule) return a.M >= 16384 && !(a.flags & CONV_OUT_F32);
    const long tiles = (long)((a.M + 255) / 256) * (a.cout_pad / 256) * (a.groups > 0 ? a.groups : 1);
    if (a.flags & CONV_OUT_F32) return (a.flags & CONV_ACCUM) && f32_min_tiles > 0 && tiles >= f32_min_tiles;
    if (a.fan_count > 1) return tiles >= 1024;        // (fan-out layer: -1 % at 513 tiles, +0.5 % from 1 026 on)
    // BOD_SMALL_TILE_MAXK=k (A/B aid): layers whose reduction is at most k (taps * cin: the memory-bound 1x1 layers of res2 / res3
    // have 64 / 128) take the 128x128 tile, two workgroups per CU, whatever their tile count
    static const int small_maxk = getenv("BOD_SMALL_TILE_MAXK") ? atoi(getenv("BOD_SMALL_TILE_MAXK")) : 0;
    if (small_maxk > 0 && a.taps > 0 && a.cin > 0 && a.taps * a.cin <= small_maxk && !a.xreuse && a.fan_count <= 1) return false;
    return tiles >= 384;
}

bool conv_igemm_uses_big_tile(const ConvArgs& a) {
    static const int forced = [] { const char* e = getenv("BOD_FORCE_CONV_TILE"); return e ? atoi(e) : 0; }();
    bool big = conv_big_tile_pays(a);
    if (forced == 256) big = a.cout_pad % 256 == 0 && !(a.flags & CONV_OUT_F32);
    if (forced == 128) big = false;
    return big;
}

bool conv_igemm_uses_full_cout_tile(const ConvArgs& a) {
    // 256x256 tiles once there are enough pixel tiles to fill the chip; BOD_FORCE_CONV_TILE=256|128
    // overrides the size heuristic (tests exercise both configurations on small inputs)
    static const int forced = [] { const char* e = getenv("BOD_FORCE_CONV_TILE"); return e ? atoi(e) : 0; }();
    bool big = conv_big_tile_pays(a);
    if (a.fan_count > 1 && fan_out_small_tile()) big = false;
    if (forced == 256) big = a.cout_pad % 256 == 0 && !(a.flags & CONV_OUT_F32);
    if (forced == 128) big = false;
    return big && a.cout_pad == 256;
}

hipError_t launch_conv_igemm(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a_local = a_in;
    // a threshold of 0 keeps every element (rate < 2^-16): the packed keep-mask needs threshold >= 1, so run without dropout
    if ((a_local.flags & CONV_DROPOUT) && a_local.drop_threshold == 0) a_local.flags &= ~CONV_DROPOUT;
    // BOD_RES_REGISTER=1: residuals read from global memory in the accumulator layout instead of through the LDS tile (A/B aid)
    static const bool res_register = getenv("BOD_RES_REGISTER") && atoi(getenv("BOD_RES_REGISTER")) == 1;
    if (res_register && a_local.variant == 0 && !a_local.xreuse && !a_local.split && a_local.ksplit <= 1) a_local.variant = 82;
    // BOD_NT_STORES (default 4; 0 = plain stores everywhere): bit 0 = non-temporal output stores in the generic kernel (backbone / FPN), bit 1 = in
    // the row-reuse kernels (head towers, fan-out launch).  Measured on one box, 256 frames: backbone -0.35 ms with bit 0, the tower
    // launches +0.5 ms beside it (net 0); bit 1 costs the towers 1 % and the fan-out launch 4 %: the outputs are re-read by the next
    // launch's neighbouring tiles through L2 after all.  Kept as an A/B switch.
    static const int nt_stores = getenv("BOD_NT_STORES") ? atoi(getenv("BOD_NT_STORES")) : 4;      // (round 5: bit 2 on -- the fan-out launch, 13.5 -> 13.2 ms per 512 frames in two same-box A/B pairs)
    // bit 2 (round 5): the fan-out launch ALONE -- its ten masked copies (42.9 GB per 512 frames) are what pushes the three heads' weights
    // and the shared pyramid rows out of L2 (11.2 GB fetched for 1.4 GB algorithmic, profiles/round4_head_conv_pmc.json launch 0)
    if (!(a_local.flags & CONV_OUT_F32) && !a_local.split && a_local.ksplit <= 1 &&
        (((nt_stores & 1) && !a_local.xreuse) || ((nt_stores & 2) && a_local.xreuse) || ((nt_stores & 4) && a_local.xreuse && a_local.fan_count > 1)))
        a_local.flags |= CONV_NT_OUT;
    // bits 3 / 4 (f16mx towers): the per-sample tower layers' hx outputs / the first layer's ten-fold hx outputs
    if (a_local.mx && (((nt_stores & 8) && a_local.fan_count <= 1) || ((nt_stores & 16) && a_local.fan_count > 1))) a_local.flags |= CONV_NT_OUT;
    {   // Work-item order of the fan-out launch (kernels.h ConvArgs.fan_chunk). Measured round 6 (profiles/round6_mx_ablations.txt,
        // "fan-out work-item order"): chunks of 8..64 tiles are 0.9 % faster than the interleaved order (12.62 against 12.73 ms at
        // 512 frames), all the same within noise; 16 is the default, BOD_FAN_CHUNK=0 gives the interleaved order back.
        static const int fan_chunk = getenv("BOD_FAN_CHUNK") ? atoi(getenv("BOD_FAN_CHUNK")) : 16;
        a_local.fan_chunk = (a_local.fan_count > 1 && a_local.groups > 1 && !a_local.mx && !a_local.split) ? fan_chunk : 0;
    }
    const ConvArgs& a = a_local;
    if (a.M <= 0) return hipSuccess;
    if (a.cin % 64 != 0 || a.cout_pad % 64 != 0) return hipErrorInvalidValue;
    if (a.mx) {                                      // f16mx precision: head-tower launches on the row-reuse loop, whatever the tile heuristics say
        if ((a.mx < 1 || a.mx > 3) || !a.split || a.xreuse != 2 || a.cout_pad != 256 || a.cin != 512 || a.taps != 9 || a.KW != 3 || !a.ext || a.M % 256 != 0 ||
            (a.flags & CONV_OUT_F32) || a.ksplit > 1 || (a.variant != 0 && a.variant != 90 && a.variant != 91) || a.groups < 1 || a.groups > 3)
            return hipErrorInvalidValue;
        for (int g = 0; g < a.groups; ++g)
            if (a.g[g].res || a.g[g].out_relu || a.g[g].ch_w2 || a.g[g].ch_w3 || (a.g[g].w2 && (a.g[g].out_hx || a.fan_count > 1))) return hipErrorInvalidValue;
#ifndef BOD_DEV_MX_ONLY
        if (a.variant == 90 && a.mx == 1) return launch_mx<11>(a, s);          // phase clock (tests/tools/bench_head_conv.py)
        if (a.variant == 91 && a.mx == 1) return launch_mx<12>(a, s);          // ... without the loop's LDS-DMA
#endif
#ifdef BOD_DEV_MX_ONLY
        return a.mx == 1 ? launch_mx<1>(a, s) : hipErrorInvalidValue;
#else
        if (a.mx == 3) return a.variant == 0 ? launch_mx<3>(a, s) : hipErrorInvalidValue;          // f16mx4: h4 rows in
        return a.mx == 1 ? launch_mx<1>(a, s) : launch_mx<2>(a, s);
#endif
    }
#ifdef BOD_DEV_MX_ONLY
    // developer build (hipcc -DBOD_DEV_MX_ONLY -c conv_igemm.hip): conv_igemm_mx_kernel<1> alone, for a fast compile / disassemble loop
    return hipErrorInvalidValue;
#else
    static const int forced = [] { const char* e = getenv("BOD_FORCE_CONV_TILE"); return e ? atoi(e) : 0; }();
    if (!forced && !(a.flags & CONV_NT_OUT) && conv_pointwise_eligible(a)) return launch_conv_pointwise(a, s);   // streaming 1x1 kernel (bit-identical)
    if (!forced && !(a.flags & CONV_NT_OUT) && conv_slide3x3_eligible(a)) return launch_conv_slide3x3(a, s);     // sliding-window 3x3, 64 -> 64 (bit-identical)
    // a fused "next block's 2a" (ch_w3 without the chain's ch_w2) exists only in the pointwise kernel: a plan that carries one must
    // never fall through to the generic kernel, which would run the expansion and silently skip the reduction
    if (a.g[0].ch_w3 && !a.g[0].ch_w2) return hipErrorInvalidValue;
    bool big = conv_big_tile_pays(a);
    if (a.fan_count > 1 && fan_out_small_tile()) big = false;
    if (forced == 256) big = a.cout_pad % 256 == 0 && !(a.flags & CONV_OUT_F32);
    if (forced == 128) big = false;
    for (int g = 0; g < a.groups; ++g)
        if (a.g[g].w2 && !(big && a.cout_pad == 256)) return hipErrorInvalidValue;   // fusion needs the full-cout tile
    if (a.split) {                                   // bf16x3: fused 1x1 (+ aggregation) on the row-reuse loop only, no ablation builds, no fan-out on the row-reuse loop
        if (a.variant != 0 || a.cin % 128 != 0 || (a.xreuse && a.xreuse != 2)) return hipErrorInvalidValue;
        for (int g = 0; g < a.groups; ++g) if (a.g[g].w2 && a.fan_count > 1) return hipErrorInvalidValue;    // fused groups never fan out
        for (int g = 0; g < a.groups; ++g) if (a.g[g].w2 && !a.xreuse) return hipErrorInvalidValue;   // the fused 1x1 lives in the row-reuse kernel's epilogue
    }
    // bottleneck chain (ConvGroup.ch_w2): one cout tile holding all couts, 128-pixel tiles, bf16, no split-K
    if (a.g[0].ch_w2) {
        if (a.groups != 1 || big || a.split || a.xreuse || a.ksplit > 1 || a.variant != 0 || a.fan_count > 1 || (a.flags & (CONV_OUT_F32 | CONV_DROPOUT)) ||
            !(a.flags & CONV_RELU) || a.g[0].res || a.g[0].out_relu || a.g[0].w2 || a.g[0].ch_c2 % 128 != 0 || a.cout_valid != a.cout_pad)
            return hipErrorInvalidValue;
        if (a.cout_pad == 64) return launch_cfg<64, 128, 1, 4, 10>(a, s);
        if (a.cout_pad == 128) return launch_cfg<128, 128, 2, 2, 10>(a, s);
        return hipErrorInvalidValue;
    }
    if (a.ksplit > 1) {
        if ((a.cin / 64) % a.ksplit != 0 || !a.partial || a.xreuse || a.fan_count > 1 || (a.flags & CONV_DROPOUT) || a.variant != 0)
            return hipErrorInvalidValue;
        for (int g = 0; g < a.groups; ++g) if (a.g[g].w2) return hipErrorInvalidValue;
        hipError_t e;
        if (a.split) e = a.cout_pad % 128 == 0 ? launch_cfg<128, 128, 2, 2, 0, false, true>(a, s) : launch_cfg<64, 128, 1, 4, 0, false, true>(a, s);
        else e = a.cout_pad % 128 == 0 ? launch_cfg<128, 128, 2, 2, 0>(a, s) : launch_cfg<64, 128, 1, 4, 0>(a, s);
        if (e != hipSuccess) return e;
        const long q = (long)a.M * (a.cout_pad / 8);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((q + 255) / 256), a.groups), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (a.split) {
        if (a.xreuse) {
            if (!(big && a.cout_pad == 256 && a.taps == 9 && a.KW == 3 && a.ext && a.M % 256 == 0) || (a.flags & CONV_OUT_F32)) return hipErrorInvalidValue;
            return launch_cfg<256, 256, 2, 4, 0, true, true>(a, s);
        }
        if (big) return launch_cfg<256, 256, 2, 4, 0, false, true>(a, s);
        if (a.cout_pad % 128 == 0) return launch_cfg<128, 128, 2, 2, 0, false, true>(a, s);
        return launch_cfg<64, 128, 1, 4, 0, false, true>(a, s);
    }
    const int variant = a.variant;
    if (a.xreuse) {
        if (!(big && a.cout_pad == 256 && a.taps == 9 && a.KW == 3 && a.ext && a.M % 256 == 0)) return hipErrorInvalidValue;
        for (int g = 0; g < a.groups; ++g) if (a.g[g].res || a.g[g].out_relu) return hipErrorInvalidValue;   // compiled out of the row-reuse kernel
        if (a.flags & CONV_OUT_F32) return hipErrorInvalidValue;                                              // likewise
        // xreuse == 2: compact-state, software-pipelined loop (32-bit byte offsets against the tile's first extended row);
        // otherwise (or variant 81, for A/B timing) the first-generation loop with 64-bit pointers
        if (a.xreuse != 2 || a.variant == 81) return launch_cfg<256, 256, 2, 4, 81, true>(a, s);
        if (a.variant == 1) return launch_cfg<256, 256, 2, 4, 1, true>(a, s);     // no epilogue
        if (a.variant == 2) return launch_cfg<256, 256, 2, 4, 2, true>(a, s);     // no staging after tile 0
        if (a.variant == 4) return launch_cfg<256, 256, 2, 4, 4, true>(a, s);     // cheap hash instead of Philox
        if (a.variant == 30) return launch_cfg<256, 256, 2, 4, 30, true>(a, s);   // no global stores
        if (a.variant == 90) return launch_cfg<256, 256, 2, 4, 90, true>(a, s);   // phase clock
        if (a.variant == 96) return launch_xr_persistent<0>(a, s);                                      // persistent, one workgroup per CU
        if (a.variant == 31) return launch_cfg<256, 256, 2, 4, 31, true>(a, s);   // epilogue ends after bias/ReLU/pack
        if (a.variant != 0 && a.variant != 83) return hipErrorInvalidValue;       // 83: production build, dropout decisions drawn in the epilogue (A/B)
        // the N-way fan-out launch of the first tower layer is its own kernel symbol (ABL = 5: the production code, of which
        // it runs the loop and the register-resident fan-out epilogue), so that a kernel trace lists the per-sample tower
        // launches -- the roofline kernel of bench.py -- and the fan-out launch separately
        if (a.fan_count > 1) return launch_cfg<256, 256, 2, 4, 5, true>(a, s);
        // ... and so are the plane -> plane 3x3 layers of the backbone / FPN that the engine plans on this loop (ABL = 6: the production
        // code without dropout -- no in-loop Philox build)
        if (a.plane_h > 0 && !(a.flags & CONV_DROPOUT) && a.variant == 0) return launch_cfg<256, 256, 2, 4, 6, true>(a, s);
        // BOD_TOWER_MIDBAR=0: the per-sample tower launches on the round-2 loop (barrier at the top of the K-tile, compiler-placed
        // LDS waits; ABL = 9) -- A/B aid, bit-identical results
        static const bool old_loop = getenv("BOD_TOWER_MIDBAR") && atoi(getenv("BOD_TOWER_MIDBAR")) == 0;
        if (old_loop && a.variant == 0) return launch_cfg<256, 256, 2, 4, 9, true>(a, s);
        return launch_cfg<256, 256, 2, 4, 0, true>(a, s);
    }
    switch (variant) {          // ablation builds of the generic loop (tests/tools/bench_head_conv.py)
        case 0: break;
        case 82: break;                                             // production build, residual read in the accumulator layout (A/B)
        case 1: return launch_cfg<256, 256, 2, 4, 1>(a, s);        // no epilogue
        case 2: return launch_cfg<256, 256, 2, 4, 2>(a, s);        // no staging after tile 0
        case 4: return launch_cfg<256, 256, 2, 4, 4>(a, s);        // cheap hash instead of Philox
        case 30: return launch_cfg<256, 256, 2, 4, 30>(a, s);      // no global stores
        case 31: return launch_cfg<256, 256, 2, 4, 31>(a, s);      // epilogue ends after bias/ReLU/pack
        case 91: return launch_cfg<128, 128, 2, 2, 90>(a, s);      // 128x128: phase clock
        case 92: return launch_cfg<128, 128, 2, 2, 1>(a, s);       // 128x128: no epilogue
        case 93: return launch_cfg<128, 128, 2, 2, 4>(a, s);       // 128x128: cheap hash instead of Philox
        case 94: return launch_cfg<256, 256, 2, 4, 0>(a, s);       // fan-out layer on the 256x256 tile
        case 95: return launch_cfg<128, 128, 2, 2, 30>(a, s);      // 128x128: no global stores
        default: return hipErrorInvalidValue;
    }
    if (big) return launch_cfg<256, 256, 2, 4, 0>(a, s);
    if (a.cout_pad % 128 == 0) return launch_cfg<128, 128, 2, 2, 0>(a, s);
    return launch_cfg<64, 128, 1, 4, 0>(a, s);
#endif
}
