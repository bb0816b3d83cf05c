// Probe for the parity mode's "f16 + MX cross terms" tower arithmetic (tests/tools/tower_numerics.py):
//   part 1  operand / scale layout of v_mfma_scale_f32_{32x32x64,16x16x128}_f8f6f4 (bf8 and fp6 operands) against a host GEMM,
//           the packing of v_cvt_scalef32_2xpk16_fp6_f32, rounding / saturation of v_cvt_pk_bf8_f32, f16 subnormals in the f16 MFMA;
//   part 2  sustained rates of register-resident loops under the board's power limit (2 waves per SIMD, random operands, no memory
//           traffic): f16 / bf16 / MX-bf8 / MX-fp6 / MX-fp4 in both tile shapes, and the mixed sequence of a tower K-tile pair
//           (64 channels: two f16 products + one MX product per 16x16 fragment pair).
// usage (GPU box): hipcc --offload-arch=gfx950 -O3 tests/tools/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(6))) int i32x6;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

// ---------------------------------------------------------------- part 1 kernels
template <int SHAPE, int FMT>       // SHAPE 0: 32x32x64, 1: 16x16x128; FMT 1 = bf8 (e5m2), 2 = fp6 (e2m3), 4 = fp4 (e2m1)
__global__ void mx_once(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* out) {
    const int l = threadIdx.x;
    if constexpr (SHAPE == 0) {
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, FMT, FMT, 0, sa[l], 0, sb[l]);
        for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
    } else {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, FMT, FMT, 0, sa[l], 0, sb[l]);
        for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
    }
}
__global__ void cvt_probe(const float* f, int* out) {
    // fp6: 2 x 16 floats -> 32 e2m3 in 6 dwords
    typedef __attribute__((ext_vector_type(16))) float v16;
    v16 v0, v1;
    for (int i = 0; i < 16; ++i) { v0[i] = f[i]; v1[i] = f[16 + i]; }
    const i32x6 p = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(v0, v1, 1.0f);
    for (int i = 0; i < 6; ++i) out[i] = p[i];
    const i32x6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(v0, v1, 4.0f);       // what does the scale operand do? (divide by it?)
    for (int i = 0; i < 6; ++i) out[6 + i] = q[i];
    // bf8: rounding and saturation
    for (int i = 0; i < 8; ++i) out[12 + i] = __builtin_amdgcn_cvt_pk_bf8_f32(f[32 + 2 * i], f[33 + 2 * i], 0, false);
}
__global__ void f16_denorm_probe(float* out) {
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // A[row 0][k 0] = 2^-20 (f16 subnormal), B[k 0][col 0] = 2^10 -> D[0][0] = 2^-10 if subnormal inputs are honoured
    if (l == 0) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1024.f; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (l == 0) out[0] = c[0];
    f32x16 c2;
    for (int r = 0; r < 16; ++r) c2[r] = 0.f;
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    if (l == 0) out[1] = c2[0];
}

// ---------------------------------------------------------------- host-side formats
static float bf8_to_f(uint8_t v) {                 // e5m2, bias 15
    const int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
    float r = e == 0 ? ldexpf((float)m, -16) : ldexpf(1.0f + m / 4.0f, e - 15);
    return s ? -r : r;
}
static float fp6_to_f(uint8_t v) {                 // e2m3, bias 1
    const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
    float r = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
    return s ? -r : r;
}
static float fp4_to_f(uint8_t v) {                 // e2m1, bias 1
    const int s = (v >> 3) & 1, e = (v >> 1) & 3, m = v & 1;
    float r = e == 0 ? m * 0.5f : ldexpf(1.0f + m * 0.5f, e - 1);
    return s ? -r : r;
}

template <int SHAPE, int FMT>
static void layout_test(const char* name) {
    constexpr int R = SHAPE == 0 ? 32 : 16, KB = SHAPE == 0 ? 2 : 4, K = KB * 32;
    // hypothesis H1: lane l holds row/col l % R, K block l / R, element j of the block at bits of the lane's operand in order; the
    // lane's scale byte 0 scales exactly that (row, block)
    std::vector<int> ha(64 * 8, 0), hb(64 * 8, 0), hsa(64), hsb(64);
    std::vector<float> A(R * K), B(K * R), SA(R * KB), SB(R * KB);
    srand(7 + SHAPE * 10 + FMT);
    for (int l = 0; l < 64; ++l) {
        const int r = l % R, kb = l / R;
        uint8_t* pa = reinterpret_cast<uint8_t*>(&ha[l * 8]);
        uint8_t* pb = reinterpret_cast<uint8_t*>(&hb[l * 8]);
        for (int j = 0; j < 32; ++j) {
            uint8_t ea, eb; float fa, fb;
            if (FMT == 1) {
                do { ea = rand() & 0xFF; } while (((ea >> 2) & 31) == 31 || ((ea >> 2) & 31) < 10 || ((ea >> 2) & 31) > 20);
                do { eb = rand() & 0xFF; } while (((eb >> 2) & 31) == 31 || ((eb >> 2) & 31) < 10 || ((eb >> 2) & 31) > 20);
                fa = bf8_to_f(ea); fb = bf8_to_f(eb); pa[j] = ea; pb[j] = eb;
            } else if (FMT == 2) {
                ea = rand() & 0x3F; eb = rand() & 0x3F; fa = fp6_to_f(ea); fb = fp6_to_f(eb);
                // element j at bits [6j, 6j+6) of the lane's 192-bit string
                for (int bit = 0; bit < 6; ++bit) {
                    if ((ea >> bit) & 1) pa[(6 * j + bit) >> 3] |= 1u << ((6 * j + bit) & 7);
                    if ((eb >> bit) & 1) pb[(6 * j + bit) >> 3] |= 1u << ((6 * j + bit) & 7);
                }
            } else {
                ea = rand() & 0xF; eb = rand() & 0xF; fa = fp4_to_f(ea); fb = fp4_to_f(eb);
                pa[j >> 1] |= ea << ((j & 1) * 4); pb[j >> 1] |= eb << ((j & 1) * 4);
            }
            A[r * K + kb * 32 + j] = fa; B[(kb * 32 + j) * R + r] = fb;
        }
        const int ea8 = 120 + rand() % 12, eb8 = 120 + rand() % 12;
        hsa[l] = ea8 | 0x55AA00; hsb[l] = eb8 | 0x33CC00;          // garbage in the other bytes: op_sel 0 must read byte 0 only
        SA[r * KB + kb] = ldexpf(1.0f, ea8 - 127); SB[r * KB + kb] = ldexpf(1.0f, eb8 - 127);
    }
    i32x8 *da, *db; int *dsa, *dsb; float* dout;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((mx_once<SHAPE, FMT>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
    std::vector<float> out(64 * 16);
    hipMemcpy(out.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int row = 0; row < R; ++row)
        for (int col = 0; col < R; ++col) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)A[row * K + k] * SA[row * KB + k / 32] * (double)B[k * R + col] * SB[col * KB + k / 32];
            float got;
            if (SHAPE == 0) {          // col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
                const int hi = (row >> 2) & 1, reg = (row & 3) + 4 * (row >> 3);
                got = out[(hi * 32 + col) * 16 + reg];
            } else {                   // col = lane & 15, row = (lane >> 4) * 4 + reg
                got = out[((row >> 2) * 16 + col) * 4 + (row & 3)];
            }
            maxerr = fmax(maxerr, fabs(got - ref)); maxref = fmax(maxref, fabs(ref));
        }
    printf("layout %-28s H1 (lane = row + R*block, element j in order, scale byte 0 per lane): max |err| %.3g of max |ref| %.3g -> %s\n", name, maxerr,
           maxref, maxerr <= 1e-4 * maxref ? "CONFIRMED" : "MISMATCH");
    hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dout);
}

// ---------------------------------------------------------------- part 2: sustained rates
// MODE 0 bf16 32x32x16 | 1 bf16 16x16x32 | 2 f16 32x32x16 | 3 f16 16x16x32 | 4 MX 32x32x64 | 5 MX 16x16x128 | 6 mixed 32x32 (4 f16 + 2 MX per
// fragment pair and 64 channels) | 7 mixed 16x16 (2 f16 + 1 MX)
template <int MODE, int FMT>
__global__ __launch_bounds__(512) void rate_loop(const i32x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    i32x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[i * 64 + lane]; b[i] = src[(4 + i) * 64 + lane]; }
    const int sc = 127;
    auto h16 = [](const i32x8& v, int half) { f16x8 r; const f16x8* p = reinterpret_cast<const f16x8*>(&v); r = p[half]; return r; };
    auto b16 = [](const i32x8& v, int half) { bf16x8 r; const bf16x8* p = reinterpret_cast<const bf16x8*>(&v); r = p[half]; return r; };
    constexpr bool S32 = MODE == 0 || MODE == 2 || MODE == 4 || MODE == 6;
    f32x16 acc32[8];
    f32x4 acc16[32];
    if constexpr (S32) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if constexpr (MODE == 0) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b16(a[(i + k) & 3], k & 1), b16(b[(i * 3 + k) & 3], (k >> 1) & 1), acc32[i], 0, 0, 0);
                    else acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h16(a[(i + k) & 3], k & 1), h16(b[(i * 3 + k) & 3], (k >> 1) & 1), acc32[i], 0, 0, 0);
                }
        } else if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    if constexpr (MODE == 1) acc16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b16(a[(i + k) & 3], k & 1), b16(b[(i * 3 + k) & 3], (k >> 1) & 1), acc16[i], 0, 0, 0);
                    else acc16[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h16(a[(i + k) & 3], k & 1), h16(b[(i * 3 + k) & 3], (k >> 1) & 1), acc16[i], 0, 0, 0);
                }
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc32[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(i + k) & 3], b[(i * 3 + k) & 3], acc32[i], FMT, FMT, 0, sc, 0, sc);
        } else if constexpr (MODE == 5) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc16[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[(i + k) & 3], b[(i * 3 + k) & 3], acc16[i], FMT, FMT, 0, sc, 0, sc);
        } else if constexpr (MODE == 6) {
            // a K-tile pair of the tower loop per 32x32 fragment pair: H tile = 4 f16 k-steps, X tile = 2 MX products
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h16(a[(i + k) & 3], k & 1), h16(b[(i * 3 + k) & 3], (k >> 1) & 1), acc32[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc32[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(i + k) & 3], b[(i * 3 + k) & 3], acc32[i], FMT, FMT, 0, sc, 0, sc);
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc16[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h16(a[(i + k) & 3], k & 1), h16(b[(i * 3 + k) & 3], (k >> 1) & 1), acc16[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 32; ++i) acc16[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 3], b[(i * 3) & 3], acc16[i], FMT, FMT, 0, sc, 0, sc);
        }
    }
    float s = 0.f;
    if constexpr (S32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc32[i][0] + acc32[i][7];
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) s += acc16[i][0] + acc16[i][3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int FMT>
static void rate(const char* name, const i32x8* d, float* o, double macs_per_iter_per_wave, double direct_macs_per_iter_per_wave) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256;
    int iters = 2000;
    double best_ms = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rate_loop<MODE, FMT>), dim3(blocks), dim3(512), 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) { iters = (int)(iters * 60.0 / ms); continue; }          // ~60 ms per timed launch: the power limiter has settled
        if (ms < best_ms) best_ms = ms;
    }
    const double waves = blocks * 8.0;
    const double tmacs = waves * iters * macs_per_iter_per_wave / (best_ms * 1e-3) / 1e12;
    printf("rate %-44s %7.0f T MAC-slots/s issued (%5.0f TFLOP/s)", name, tmacs, 2 * tmacs);
    if (direct_macs_per_iter_per_wave > 0)
        printf("   = %6.0f T direct MACs/s (bf16x3 at 1.3 PFLOP/s issued: 217)", waves * iters * direct_macs_per_iter_per_wave / (best_ms * 1e-3) / 1e12);
    printf("\n");
}

int main() {
    // ---- part 1
    layout_test<0, 1>("32x32x64 bf8");
    layout_test<1, 1>("16x16x128 bf8");
    layout_test<0, 2>("32x32x64 fp6 (e2m3)");
    layout_test<1, 2>("16x16x128 fp6 (e2m3)");
    layout_test<0, 4>("32x32x64 fp4");
    layout_test<1, 4>("16x16x128 fp4");
    {
        std::vector<float> f(48);
        const float v6[32] = {0.125f, 0.25f, 0.5f, 1.0f, 1.125f, 2.0f, 3.5f, 7.5f, -0.125f, -1.0f, 0.0f, 0.0625f, 0.1875f, 8.0f, 100.f, -7.5f,
                              1.0f, 2.0f, 3.0f, 4.0f, 5.0f, 6.0f, 7.0f, 0.375f, 0.625f, 0.75f, 0.875f, 1.25f, 1.5f, 1.75f, 2.5f, -3.0f};
        for (int i = 0; i < 32; ++i) f[i] = v6[i];
        const float v8[16] = {1.0f, 1.125f, 1.375f, 1.25f, 60000.f, 1e6f, -1e6f, 6.1e-5f, 1.5e-5f, 7.6e-6f, 3.0e-6f, 0.3f, -0.3f, 1.874f, 1.876f, 0.f};
        for (int i = 0; i < 16; ++i) f[32 + i] = v8[i];
        float* df; int* dout;
        hipMalloc(&df, 48 * 4); hipMalloc(&dout, 20 * 4);
        hipMemcpy(df, f.data(), 48 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(1), 0, 0, df, dout);
        int o[20];
        hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
        for (int pass = 0; pass < 2; ++pass) {
            printf("cvt_scalef32_2xpk16_fp6_f32 scale %s: dwords", pass == 0 ? "1.0" : "4.0");
            for (int i = 0; i < 6; ++i) printf(" %08x", (unsigned)o[pass * 6 + i]);
            printf("\n   decoded as element j at bits [6j, 6j+6):");
            const uint8_t* p = reinterpret_cast<const uint8_t*>(&o[pass * 6]);
            for (int j = 0; j < 32; ++j) {
                unsigned e = 0;
                for (int bit = 0; bit < 6; ++bit) e |= ((p[(6 * j + bit) >> 3] >> ((6 * j + bit) & 7)) & 1u) << bit;
                printf(" %g", fp6_to_f((uint8_t)e));
            }
            printf("\n   inputs:                                 ");
            for (int j = 0; j < 32; ++j) printf(" %g", v6[j]);
            printf("\n");
        }
        printf("cvt_pk_bf8_f32 (in -> out):");
        for (int i = 0; i < 8; ++i) printf("  [%g, %g] -> [%g, %g] (%04x)", v8[2 * i], v8[2 * i + 1], bf8_to_f(o[12 + i] & 0xFF), bf8_to_f((o[12 + i] >> 8) & 0xFF), o[12 + i] & 0xFFFF);
        printf("\n");
        float* dd; hipMalloc(&dd, 8);
        hipLaunchKernelGGL(f16_denorm_probe, dim3(1), dim3(64), 0, 0, dd);
        float r[2]; hipMemcpy(r, dd, 8, hipMemcpyDeviceToHost);
        printf("f16 MFMA with a subnormal input 2^-20 x 2^10: 16x16x32 -> %g, 32x32x16 -> %g (2^-10 = %g if subnormals are honoured)\n", r[0], r[1], ldexp(1.0, -10));
    }
    // ---- part 2
    std::vector<int> h(8 * 64 * 8);
    i32x8* d; float* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 256 * 512 * 4);
    auto fill = [&](int kind) {        // 0: random f16 in [-1, 1]; 1: random bf16 in [-1, 1]; 2: random bytes with exponent bits kept sane (bf8 / fp6 / fp4 alike)
        srand(3);
        for (auto& w : h) {
            if (kind == 2) { w = (int)(((unsigned)rand() ^ ((unsigned)rand() << 16)) & 0xBBBBBBBBu); continue; }
            unsigned short hw[2];
            for (int q = 0; q < 2; ++q) {
                const float f = ((rand() % 2001) - 1000) / 1000.0f;
                if (kind == 0) { _Float16 x = (_Float16)f; memcpy(&hw[q], &x, 2); }
                else { unsigned u; memcpy(&u, &f, 4); hw[q] = u >> 16; }
            }
            w = hw[0] | (hw[1] << 16);
        }
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    };
    const double M32 = 32.0 * 32, M16 = 16.0 * 16;
    fill(1);
    rate<0, 0>("bf16 32x32x16", d, o, 64 * M32 * 16, 0);
    rate<1, 0>("bf16 16x16x32", d, o, 128 * M16 * 32, 0);
    fill(0);
    rate<2, 0>("f16 32x32x16", d, o, 64 * M32 * 16, 0);
    rate<3, 0>("f16 16x16x32", d, o, 128 * M16 * 32, 0);
    fill(2);
    rate<4, 1>("MX bf8 32x32x64", d, o, 32 * M32 * 64, 0);
    rate<5, 1>("MX bf8 16x16x128", d, o, 64 * M16 * 128, 0);
    rate<4, 2>("MX fp6 32x32x64", d, o, 32 * M32 * 64, 0);
    rate<5, 2>("MX fp6 16x16x128", d, o, 64 * M16 * 128, 0);
    rate<4, 4>("MX fp4 32x32x64", d, o, 32 * M32 * 64, 0);
    rate<5, 4>("MX fp4 16x16x128", d, o, 64 * M16 * 128, 0);
    // mixed: per iteration a wave covers 64 channels of its 8 (32 x 32) / 32 (16 x 16) fragment pairs = 8192 x 64 direct MACs
    rate<6, 1>("tower K-tile pair, 32x32: 4 f16 + 2 MX-bf8", d, o, 32 * M32 * 16 + 16 * M32 * 64, 8 * M32 * 64);
    rate<7, 1>("tower K-tile pair, 16x16: 2 f16 + 1 MX-bf8", d, o, 64 * M16 * 32 + 32 * M16 * 128, 32 * M16 * 64);
    rate<6, 2>("tower K-tile pair, 32x32: 4 f16 + 2 MX-fp6", d, o, 32 * M32 * 16 + 16 * M32 * 64, 8 * M32 * 64);
    rate<7, 2>("tower K-tile pair, 16x16: 2 f16 + 1 MX-fp6", d, o, 64 * M16 * 32 + 32 * M16 * 128, 32 * M16 * 64);
    rate<6, 4>("tower K-tile pair, 32x32: 4 f16 + 2 MX-fp4", d, o, 32 * M32 * 16 + 16 * M32 * 64, 8 * M32 * 64);
    rate<7, 4>("tower K-tile pair, 16x16: 2 f16 + 1 MX-fp4", d, o, 64 * M16 * 32 + 32 * M16 * 128, 32 * M16 * 64);
    return 0;
}
