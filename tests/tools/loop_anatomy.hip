// Anatomy of the tower kernel's K-tile loop: the same per-wave work -- 64 x v_mfma_f32_16x16x32_bf16 per K-tile on a 128 x 64 wave tile,
// 8 waves per CU -- rebuilt from its parts, which are switched on one at a time: the K-tile barrier, the 24 ds_read_b128 of the A/B
// fragments, the 6 LDS-DMA pieces per wave (4 weight + 2 activation), and Philox-like integer VALU work in the MFMA shadows.
// Prints per configuration: time per K-tile per CU, the shader clock (s_memtime cycles / s_memrealtime ticks) and the MFMA duty
// (2048 matrix-pipe cycles per K-tile per SIMD / measured cycles), and a hash of the output (configurations that read the same
// LDS contents in the same order must agree: the mid-tile-barrier form equals the top-barrier form without DMA).
// usage (GPU box): hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tests/tools/loop_anatomy.hip -o /tmp/loop_anatomy && /tmp/loop_anatomy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

enum { F_BAR = 1, F_LDS = 2, F_DMA = 4, F_VALU = 8, F_DMA_DWORD = 16, F_DMA_BURST = 32, F_DMA_NOWAIT = 64, F_PRIO = 128, F_RING5 = 256, F_MIDBAR = 512, F_AGPR = 1024, F_DMA_ASYM = 2048, F_DMA_BUF = 4096 };
constexpr int ROWB = 128, WST = 256 * ROWB, XROWS = 320, XBUF = XROWS * ROWB, LDS_BYTES = 2 * WST + 2 * XBUF;

template <bool AGPR>
__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));      // accumulators in AGPRs
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__device__ unsigned long long g_cycles[4];

template <int F>
__global__ __launch_bounds__(512) void loop_kernel(const char* __restrict__ wsrc, const char* __restrict__ xsrc, float* __restrict__ out, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 2, wp = wave & 3, l15 = lane & 15, q4 = lane >> 4;
    // fill LDS with the (random) source once: every stage holds real data
    for (int i = tid; i < LDS_BYTES / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(wsrc)[i];
    __syncthreads();
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int RN = (F & F_RING5) ? 5 : (F & F_MIDBAR) ? 4 : 3;      // A ring (across K-tiles: 16 % RN must be 0)
    bf16x8 Ar[RN], Bc[4];
    #pragma unroll
    for (int i = 0; i < RN; ++i) Ar[i] = reinterpret_cast<const bf16x8*>(smem)[i * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) Bc[j] = reinterpret_cast<const bf16x8*>(smem)[(4 + j) * 64 + lane];
    const uint32_t wlane = (uint32_t)(((tid >> 3) * 2304 + (tid & 7) * 8) * 2);      // a weight row is 2304 elements; 64 rows per piece
    const uint32_t xlane = (uint32_t)(tid * 16);
    uint32_t ph0 = tid * 2654435761u, ph1 = tid ^ 0x9E3779B9u, ph2 = 12345u, ph3 = tid + 77u;
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int AHEAD = (F & F_RING5) ? 4 : 2, BS = (F & F_MIDBAR) ? 16 - AHEAD : 0;      // the barrier sits in front of step BS of the K-tile's 16
    auto a_addr = [&](int kt_, int st) {          // A fragment of step st (= ks * 8 + fc) of K-tile kt_
        const int wa = (kt_ & 1) * WST + (wc * 128 + l15) * ROWB + ((q4 ^ ((l15 >> 1) & 7)) << 4);
        return smem + ((wa ^ ((st >> 3) << 6)) + (st & 7) * 16 * ROWB);
    };
    auto b_addr = [&](int kt_, int j, int ks) {
        const int kxc = kt_ % 3, xb = (kt_ / 3) & 1;
        const int r = wp * 64 + j * 16 + l15 + kxc;
        return smem + ((2 * WST + xb * XBUF + r * ROWB + (((q4 + (r & 6)) & 7) << 4)) ^ (ks << 6));
    };
    if ((F & F_MIDBAR) && (F & F_LDS)) {          // the fragments K-tile 0 starts with (later K-tiles: fetched behind the previous one's barrier)
#pragma unroll
        for (int j = 0; j < 4; ++j) Bc[j] = *reinterpret_cast<const bf16x8*>(b_addr(0, j, 0));
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) Ar[i] = *reinterpret_cast<const bf16x8*>(a_addr(0, i));
    }
    for (int kt = 0; kt < ktiles; ++kt) {
        const int stage = kt & 1, kxc = kt % 3, xb = (kt / 3) & 1;
        const char* wg = wsrc + (size_t)((kt % 36) * 64) * 2;                        // K-tile kt of the 256 x 2304 weight matrix (L2-resident)
        const char* xg = xsrc + (size_t)(blockIdx.x % 64) * 65536 + (size_t)(kt % 12) * 40960;
        const int wdst = (stage ^ 1) * WST, xdst = 2 * WST + (xb ^ 1) * XBUF;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int ks = st >> 3, fc = st & 7;
            if (st == BS) {
                if (F & F_DMA) { if (!(F & F_DMA_NOWAIT)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                if ((F & F_MIDBAR) && (F & F_LDS)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage's last readers have their data
                if (F & F_BAR) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
                if (!(F & F_MIDBAR) && (F & F_LDS)) {
                    if (kxc == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) Bc[j] = *reinterpret_cast<const bf16x8*>(b_addr(kt, j, 0));
                    }
#pragma unroll
                    for (int i = 0; i < AHEAD; ++i) Ar[i] = *reinterpret_cast<const bf16x8*>(a_addr(kt, i));
                }
                if ((F & F_DMA) && (F & F_DMA_BURST)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const char* p = wg + (size_t)i * 64 * 2304 * 2; uint32_t wl = wlane; asm volatile("" : "+s"(p), "+v"(wl));
                        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p + wl), LDS_PTR(smem + wdst + (i * 512 + wave * 64) * 16), 16, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const char* p = xg + (size_t)i * 8192; uint32_t xl = xlane; asm volatile("" : "+s"(p), "+v"(xl));
                        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p + xl), LDS_PTR(smem + xdst + (i * 512 + wave * 64) * 16), 16, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (F & F_LDS) {
                if (st + AHEAD < 16) Ar[(st + AHEAD) % RN] = *reinterpret_cast<const bf16x8*>(a_addr(kt, st + AHEAD));
                else if (F & F_MIDBAR) Ar[(st + AHEAD) % RN] = *reinterpret_cast<const bf16x8*>(a_addr(kt + 1, st + AHEAD - 16));
            }
            if (F & F_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mfma16<(F & F_AGPR) != 0>(acc[fc][j], Ar[st % RN], Bc[j]);
                if ((F & F_LDS) && st == 7) Bc[j] = *reinterpret_cast<const bf16x8*>(b_addr(kt, j, 1));
                if ((F & F_LDS) && st == 15 && (kxc < 2 || (F & F_MIDBAR))) Bc[j] = *reinterpret_cast<const bf16x8*>(b_addr(kt + 1, j, 0));
            }
            if (F & F_PRIO) __builtin_amdgcn_s_setprio(0);
            if ((fc & 1) && st != 15) {
                const int slot = st >> 1;
                __builtin_amdgcn_sched_barrier(0);
                if ((F & F_DMA) && (F & F_DMA_ASYM)) {
                    // all pieces issued by waves 0..3 (one per SIMD), two per slot: their SIMD partners (waves 4..7) never sit in a DMA issue
                    if (wave < 4 && slot < 6) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int pc = slot * 2 + h;                      // 0..7 weight half-pieces, 8..11 activation half-pieces
                            const char* p = pc < 8 ? wg + (size_t)pc * 32 * 2304 * 2 : xg + (size_t)(pc - 8) * 4096;
                            uint32_t vl = pc < 8 ? (uint32_t)((((tid & 255) >> 3) * 2304 + (tid & 7) * 8) * 2) : (uint32_t)((tid & 255) * 16);
                            asm volatile("" : "+s"(p), "+v"(vl));
                            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p + vl), LDS_PTR(smem + (pc < 8 ? wdst : xdst - 8 * 4096) + (pc * 256 + wave * 64) * 16), 16, 0, 0);
                        }
                    }
                } else if ((F & F_DMA) && (F & F_DMA_BUF)) {
                    if (slot < 6) {
                        const char* p = slot < 4 ? wg + (size_t)slot * 64 * 2304 * 2 : xg + (size_t)(slot - 4) * 8192;
                        asm volatile("" : "+s"(p));
                        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, 0x7fffffff, 0x00020000);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (slot < 4 ? wdst + slot * 8192 : xdst + (slot - 4) * 8192) + wave * 1024), 16, slot < 4 ? wlane : xlane, 0, 0, 0);
                    }
                } else if ((F & F_DMA) && !(F & F_DMA_BURST)) {
                    if (slot < 4) {
                        const char* p = wg + (size_t)slot * 64 * 2304 * 2; uint32_t wl = wlane; asm volatile("" : "+s"(p), "+v"(wl));
                        if (F & F_DMA_DWORD) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p + wl), LDS_PTR(smem + wdst + (slot * 512 + wave * 64) * 16), 4, 0, 0);
                        else __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p + wl), LDS_PTR(smem + wdst + (slot * 512 + wave * 64) * 16), 16, 0, 0);
                    } else if (slot < 6) {
                        const char* p = xg + (size_t)(slot - 4) * 8192; uint32_t xl = xlane; asm volatile("" : "+s"(p), "+v"(xl));
                        if (F & F_DMA_DWORD) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p + xl), LDS_PTR(smem + xdst + ((slot - 4) * 512 + wave * 64) * 16), 4, 0, 0);
                        else __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p + xl), LDS_PTR(smem + xdst + ((slot - 4) * 512 + wave * 64) * 16), 16, 0, 0);
                    }
                }
                if (F & F_VALU) {
                    // one Philox4x32 round
                    const unsigned long long p0 = (unsigned long long)ph0 * 0xD2511F53u, p1 = (unsigned long long)ph2 * 0xCD9E8D57u;
                    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ph1 ^ 0x12345u, n2 = (uint32_t)(p0 >> 32) ^ ph3 ^ 0x6789u;
                    ph1 = (uint32_t)p1; ph3 = (uint32_t)p0; ph0 = n0; ph2 = n2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (tid == 0) {
        atomicAdd(&g_cycles[0], __builtin_amdgcn_s_memtime() - t0);
        atomicAdd(&g_cycles[1], __builtin_amdgcn_s_memrealtime() - r0);
    }
    float s = (float)(ph0 ^ ph1 ^ ph2 ^ ph3);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = s;
}


// ------------------------------------------------------------------------------------------------------------------------------
// The structural alternative (DESIGN.md section 10.1b, round-3 review item 1): FOUR waves per CU, one per SIMD, wave tile 128 x 128 =
// 8 x 8 fragments of 16 x 16 (256 accumulator registers -- in AGPRs with F_AGPR -- of the 512 a lone wave may use).  Same CU tile
// (256 couts x 256 pixels), same LDS image, same K-tile: per wave 128 MFMAs, 32 ds_read_b128 (16 A + 16 B: two thirds of the fragment
// reads per MFMA of the 8-wave form), 12 LDS-DMA pieces, 14 Philox rounds.  With a single wave on the SIMD nothing else covers an issue
// gap, so the fillers are placed by hand, one small group behind each MFMA (sched_barrier between groups): a 16 x 16 x 32 MFMA occupies
// the pipe for 16 cycles = 4 issue slots, i.e. up to three other instructions per MFMA are free IF they are spread.
// F_SPREAD = 0 puts a step's fillers in one clump behind its eight MFMAs instead (what a compiler-scheduled port would do).
enum { G_SPREAD = 1 << 16 };
template <int F>
__global__ __launch_bounds__(256) void loop_kernel_4w(const char* __restrict__ wsrc, const char* __restrict__ xsrc, float* __restrict__ out, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1, l15 = lane & 15, q4 = lane >> 4;
    for (int i = tid; i < LDS_BYTES / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(wsrc)[i];
    __syncthreads();
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int RN = (F & F_MIDBAR) ? 4 : 3, BS = (F & F_MIDBAR) ? 14 : 0;
    bf16x8 Ar[RN], Bc[8];
#pragma unroll
    for (int i = 0; i < RN; ++i) Ar[i] = reinterpret_cast<const bf16x8*>(smem)[i * 64 + lane];
#pragma unroll
    for (int j = 0; j < 8; ++j) Bc[j] = reinterpret_cast<const bf16x8*>(smem)[(4 + j) * 64 + lane];
    const uint32_t wlane = (uint32_t)(((tid >> 3) * 2304 + (tid & 7) * 8) * 2);      // 32 weight rows per piece (256 threads)
    const uint32_t xlane = (uint32_t)(tid * 16);
    uint32_t ph0 = tid * 2654435761u, ph1 = tid ^ 0x9E3779B9u, ph2 = 12345u, ph3 = tid + 77u;
    unsigned long long p0 = 0, p1 = 0;
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    auto a_addr = [&](int kt_, int st) {
        const int wa = (kt_ & 1) * WST + (wc * 128 + l15) * ROWB + ((q4 ^ ((l15 >> 1) & 7)) << 4);
        return smem + ((wa ^ ((st >> 3) << 6)) + (st & 7) * 16 * ROWB);
    };
    auto b_addr = [&](int kt_, int j, int ks) {
        const int kxc = kt_ % 3, xb = (kt_ / 3) & 1;
        const int r = wp * 128 + j * 16 + l15 + kxc;
        return smem + ((2 * WST + xb * XBUF + r * ROWB + (((q4 + (r & 6)) & 7) << 4)) ^ (ks << 6));
    };
    if ((F & F_MIDBAR) && (F & F_LDS)) {          // the fragments K-tile 0 starts with (later K-tiles: fetched behind the previous one's barrier)
#pragma unroll
        for (int j = 0; j < 8; ++j) Bc[j] = *reinterpret_cast<const bf16x8*>(b_addr(0, j, 0));
        Ar[0] = *reinterpret_cast<const bf16x8*>(a_addr(0, 0));
        Ar[1] = *reinterpret_cast<const bf16x8*>(a_addr(0, 1));
    }
    for (int kt = 0; kt < ktiles; ++kt) {
        const int stage = kt & 1, kxc = kt % 3, xb = (kt / 3) & 1;
        const char* wg = wsrc + (size_t)((kt % 36) * 64) * 2;
        const char* xg = xsrc + (size_t)(blockIdx.x % 64) * 65536 + (size_t)(kt % 12) * 40960;
        const int wdst = (stage ^ 1) * WST, xdst = 2 * WST + (xb ^ 1) * XBUF;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int fc = st & 7;
            if (st == BS) {
                if (F & F_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((F & F_MIDBAR) && (F & F_LDS)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (F & F_BAR) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
                if (!(F & F_MIDBAR) && (F & F_LDS)) {
                    if (kxc == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) Bc[j] = *reinterpret_cast<const bf16x8*>(b_addr(kt, j, 0));
                    }
                    Ar[0] = *reinterpret_cast<const bf16x8*>(a_addr(kt, 0));
                    Ar[1] = *reinterpret_cast<const bf16x8*>(a_addr(kt, 1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the DMA pieces of K-tile kt+1 go out in the 12 steps BEHIND this K-tile's barrier: steps BS .. BS+11 (mod 16), i.e. with
            // the mid-tile barrier steps 14, 15 of this K-tile and 0..9 of the next
            const int dslot = (st - BS + 16) % 16;
            auto filler = [&](int part) {          // the step's non-MFMA work in four parts
                if (part == 0) {
                    if (F & F_LDS) {
                        if (st + 2 < 16) Ar[(st + 2) % RN] = *reinterpret_cast<const bf16x8*>(a_addr(kt, st + 2));
                        else if (F & F_MIDBAR) Ar[(st + 2) % RN] = *reinterpret_cast<const bf16x8*>(a_addr(kt + 1, st + 2 - 16));
                    }
                } else if (part == 1) {
                    if ((F & F_DMA) && dslot < 12) {
                        // (mid-tile barrier: steps 0..9 issue the pieces 2..11 of THIS K-tile's successor, whose first two went out in steps 14, 15 of the previous one)
                        const bool late = (F & F_MIDBAR) && st < BS;
                        const char* wgn = late ? wg : wsrc + (size_t)(((kt + 1) % 36) * 64) * 2;
                        const char* xgn = late ? xg : xsrc + (size_t)(blockIdx.x % 64) * 65536 + (size_t)((kt + 1) % 12) * 40960;
                        const int wd = late ? wdst : ((F & F_MIDBAR) ? (stage) * WST : wdst), xd = late ? xdst : ((F & F_MIDBAR) ? 2 * WST + (((kt + 1) / 3) & 1 ^ 1) * XBUF : xdst);
                        const char* p = dslot < 8 ? wgn + (size_t)dslot * 32 * 2304 * 2 : xgn + (size_t)(dslot - 8) * 4096;
                        asm volatile("" : "+s"(p));
                        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, 0x7fffffff, 0x00020000);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + (dslot < 8 ? wd + dslot * 4096 : xd + (dslot - 8) * 4096) + wave * 1024), 16, dslot < 8 ? wlane : xlane, 0, 0, 0);
                    }
                } else if (part == 2) {
                    if ((F & F_VALU) && st < 14) { p0 = (unsigned long long)ph0 * 0xD2511F53u; p1 = (unsigned long long)ph2 * 0xCD9E8D57u; }
                } else {
                    if ((F & F_VALU) && st < 14) {
                        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ph1 ^ 0x12345u, n2 = (uint32_t)(p0 >> 32) ^ ph3 ^ 0x6789u;
                        ph1 = (uint32_t)p1; ph3 = (uint32_t)p0; ph0 = n0; ph2 = n2;
                    }
                }
            };
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mfma16<(F & F_AGPR) != 0>(acc[fc][j], Ar[st % RN], Bc[j]);
                if ((F & F_LDS) && st == 7) Bc[j] = *reinterpret_cast<const bf16x8*>(b_addr(kt, j, 1));
                if ((F & F_LDS) && st == 15 && (kxc < 2 || (F & F_MIDBAR))) Bc[j] = *reinterpret_cast<const bf16x8*>(b_addr(kt + 1, j, 0));
                if (F & G_SPREAD) {
                    if (j == 0) filler(0); else if (j == 2) filler(1); else if (j == 4) filler(2); else if (j == 6) filler(3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!(F & G_SPREAD)) {
                __builtin_amdgcn_sched_barrier(0);
                filler(0); filler(1); filler(2); filler(3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (tid == 0) {
        atomicAdd(&g_cycles[0], __builtin_amdgcn_s_memtime() - t0);
        atomicAdd(&g_cycles[1], __builtin_amdgcn_s_memrealtime() - r0);
    }
    float s = (float)(ph0 ^ ph1 ^ ph2 ^ ph3);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = acc[i][j];
            if constexpr ((F & F_AGPR) != 0) asm volatile("" : "+v"(v));
            s += v[0] + v[3];
        }
    out[blockIdx.x * 512 + tid] = s;
}

template <int F>
static void run4(const char* name, const char* w, const char* x, float* o, int ktiles) {
    auto kern = loop_kernel_4w<F>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e30; unsigned long long c[4] = {0, 0, 0, 0}, cb[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        unsigned long long z[4] = {0, 0, 0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_cycles), z, sizeof z);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), LDS_BYTES, 0, w, x, o, ktiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cycles), sizeof c);
        if (rep > 0 && ms < best) { best = ms; memcpy(cb, c, sizeof c); }
    }
    const double cyc = (double)cb[0] / 256 / ktiles, ghz = (double)cb[0] / (double)cb[1] * 0.1;
    const double tf = 256.0 * 4 * ktiles * 128 * 16384 / (best * 1e-3) / 1e12;
    printf("[4 waves x 128x128 ] %-58s %7.3f us/K-tile  %6.0f cycles/K-tile  clock %.3f GHz  MFMA duty %.3f  %6.0f TFLOP/s\n", name, best * 1e3 / ktiles, cyc, ghz,
           2048.0 / cyc, tf);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int F>
static void run(const char* name, const char* w, const char* x, float* o, int ktiles) {
    auto kern = loop_kernel<F>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e30; unsigned long long c[4] = {0, 0, 0, 0}, cb[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        unsigned long long z[4] = {0, 0, 0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_cycles), z, sizeof z);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), LDS_BYTES, 0, w, x, o, ktiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cycles), sizeof c);
        if (rep > 0 && ms < best) { best = ms; memcpy(cb, c, sizeof c); }
    }
    const double cyc = (double)cb[0] / 256 / ktiles, ghz = (double)cb[0] / (double)cb[1] * 0.1;
    const double tf = 256.0 * 8 * ktiles * 64 * 16384 / (best * 1e-3) / 1e12;
    { std::vector<float> ho(256 * 512); hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost); unsigned long long hsh = 1469598103934665603ull; for (float v : ho) { unsigned u; memcpy(&u, &v, 4); hsh = (hsh ^ u) * 1099511628211ull; } printf("[%016llx] ", hsh); }
    printf("%-58s %7.3f us/K-tile  %6.0f cycles/K-tile  clock %.3f GHz  MFMA duty %.3f  %6.0f TFLOP/s\n", name, best * 1e3 / ktiles, cyc, ghz, 2048.0 / cyc, tf);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const int ktiles = 3600;
    const size_t wbytes = (size_t)256 * 2304 * 2 + (1 << 20), xbytes = (size_t)64 * 65536 + 12 * 40960 + (1 << 20);
    std::vector<unsigned short> h((wbytes + xbytes) / 2);
    srand(1);
    for (auto& v : h) { float f = (rand() % 100 < 50) ? 0.f : ((rand() % 2001) - 1000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    char* d; float* o;
    hipMalloc(&d, wbytes + xbytes); hipMalloc(&o, 256 * 512 * 4);
    hipMemcpy(d, h.data(), wbytes + xbytes, hipMemcpyHostToDevice);
    const char* w = d; const char* x = d + wbytes;
    run<0>("MFMA only (operands in registers)", w, x, o, ktiles);
    run<F_AGPR>("MFMA only, accumulators in AGPRs", w, x, o, ktiles);
    run<F_BAR>("+ barrier per K-tile", w, x, o, ktiles);
    run<F_LDS | F_BAR>("+ 24 ds_read_b128 + barrier", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_RING5>("   same, A ring of five (four fragments ahead)", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_PRIO>("   same, s_setprio 1 around the MFMAs", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_AGPR>("   same, AGPR accumulators", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_MIDBAR>("   same, barrier two steps before the K-tile's end", w, x, o, ktiles);
    run<F_DMA | F_BAR>("+ 6 LDS-DMA pieces + barrier (no ds_read)", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA>("+ ds_read + barrier + 6 LDS-DMA pieces in slots", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_DMA_NOWAIT>("   same, DMA never waited for", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_DMA_DWORD>("   same, 4-byte pieces (same instruction count)", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_DMA_BURST>("   same, pieces in a burst after the barrier", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_DMA_ASYM>("   same, all pieces issued by one wave per SIMD", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_DMA_BUF>("   same, buffer_load ... lds instead of global_load_lds", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_VALU>("+ ds_read + barrier + 7 Philox rounds", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_VALU>("+ ds_read + barrier + DMA + 7 Philox rounds (= the loop)", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_VALU | F_AGPR>("   the loop, AGPR accumulators", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_VALU | F_PRIO>("   the loop, s_setprio", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_VALU | F_MIDBAR>("   the loop, barrier two steps early", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_VALU | F_DMA_BUF>("   the loop, buffer_load ... lds", w, x, o, ktiles);
    run<F_LDS | F_BAR | F_DMA | F_VALU | F_DMA_BUF | F_MIDBAR>("   the loop, buffer_load ... lds + barrier two steps early", w, x, o, ktiles);
    // ---- round 3: four waves per CU, one per SIMD, 128 x 128 wave tiles (see loop_kernel_4w)
    run4<F_BAR | F_AGPR>("MFMA + barrier, AGPR accumulators", w, x, o, ktiles);
    run4<F_BAR>("MFMA + barrier, VGPR accumulators", w, x, o, ktiles);
    run4<F_LDS | F_BAR | F_AGPR | G_SPREAD>("+ 32 ds_read_b128", w, x, o, ktiles);
    run4<F_LDS | F_BAR | F_AGPR | F_DMA | G_SPREAD>("+ ds_read + 12 buffer_load ... lds pieces", w, x, o, ktiles);
    run4<F_LDS | F_BAR | F_AGPR | F_VALU | G_SPREAD>("+ ds_read + 14 Philox rounds", w, x, o, ktiles);
    run4<F_LDS | F_BAR | F_AGPR | F_DMA | F_VALU | G_SPREAD>("the loop (ds_read + DMA + Philox), fillers spread", w, x, o, ktiles);
    run4<F_LDS | F_BAR | F_AGPR | F_DMA | F_VALU>("the loop, fillers in one clump per step", w, x, o, ktiles);
    run4<F_LDS | F_BAR | F_DMA | F_VALU | G_SPREAD>("the loop, fillers spread, VGPR accumulators", w, x, o, ktiles);
    run4<F_LDS | F_BAR | F_AGPR | G_SPREAD | F_MIDBAR>("ds_read only, barrier two steps before the K-tile's end", w, x, o, ktiles);
    run4<F_LDS | F_BAR | F_AGPR | F_DMA | F_VALU | G_SPREAD | F_MIDBAR>("the loop, fillers spread, barrier two steps early", w, x, o, ktiles);
    return 0;
}
