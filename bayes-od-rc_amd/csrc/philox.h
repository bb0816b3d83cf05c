// Philox4x32-10 (Salmon et al. SC'11; same generator as rocRAND's philox4x32_10), inlined so the
// counter layout is ours to define (DESIGN.md "RNG contract"; oracle twin: oracle/philox.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BOD_HD __host__ __device__ __forceinline__
#else
#define BOD_HD inline
#endif

struct Philox4 { uint32_t x, y, z, w; };

// a ^ b ^ c: ONE v_bitop3_b32 (truth table 0x96) on gfx950 -- the compiler emits two v_xor_b32 for the round's three-way xors, a
// third of the round's vector instructions (2 multiplies + 4 xors -> 2 + 2)
BOD_HD uint32_t philox_xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

BOD_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                             uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = philox_xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = philox_xor3((uint32_t)(p0 >> 32), c3, k1);
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// The same generator as a resumable state, so a call can be spread over several schedule slots.
struct PhiloxState { uint32_t c0, c1, c2, c3, k0, k1; };
BOD_HD void philox_rounds(PhiloxState& s, int n) {
#pragma unroll
    for (int r = 0; r < n; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * s.c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * s.c2;
        const uint32_t n0 = philox_xor3((uint32_t)(p1 >> 32), s.c1, s.k0);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = philox_xor3((uint32_t)(p0 >> 32), s.c3, s.k1);
        const uint32_t n3 = (uint32_t)p0;
        s.c0 = n0; s.c1 = n1; s.c2 = n2; s.c3 = n3;
        s.k0 += 0x9E3779B9u;
        s.k1 += 0xBB67AE85u;
    }
}

// Dropout contract v3 (DESIGN.md "RNG contract"; oracle twin oracle/philox.py): a Philox call decides 16 channels.  Its 128 bits
// (word x least significant) are read as 16 overlapping 16-bit windows at byte stride, wrapping from w back to x.  Channel c belongs
// to call group (c>>5)*2 + ((c>>2)&1) -- the four 4-channel runs {c0, c0+8, c0+16, c0+24} one lane of the 32x32 MFMA layout holds in a
// 32-channel fragment -- and run u = ((c>>4)&1)*2 + ((c>>3)&1) of the call looks at output word u: channel c&3 = 0, 1 at its two halves,
// 2 at bits 8..23, 3 at bits 24..31 with bits 0..7 of word (u+1)&3 on top.  keep iff window >= threshold.  Every window is a uniform
// 16-bit number (exact keep probability to 2^-16, as with contract v2's eight half-words per call, at half the Philox calls); two
// windows that share a byte are independent unless the more significant byte of one ties with the threshold's (probability 2^-8).
BOD_HD uint32_t dropout_group16(uint32_t c) { return (c >> 5) * 2u + ((c >> 2) & 1u); }
// the two packed window pairs of run u: .x = channels 0, 1 (low, high half), .y = channels 2, 3
struct DropPair { uint32_t x, y; };
BOD_HD DropPair dropout_run_windows(const Philox4& r, int u) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    const uint32_t a = w[u & 3], b = w[(u + 1) & 3];
    return DropPair{a, (a >> 8) | (b << 24)};
}

#define BOD_CAT_TAG 0x00CA7E60u
