// Probe of the two register hazards around an INLINE-ASM MFMA (the compiler inserts wait states for a real MFMA, none for inline asm):
//  (1) WAR: an LDS load that RETURNS into the MFMA's SrcA / SrcB right behind it -- the tower loop refreshes a B fragment in place
//      behind the MFMA that reads it (conv_igemm.hip).  Per iteration every wave (8 per CU, two per SIMD: the matrix pipe is contended)
//      issues FILL independent MFMAs, then one MFMA reading ones, then -- NOPS slots later -- a ds_read_b128 that overwrites the source
//      with twos, and checks the result (32 per element if the MFMA read ones).  Measured: never wrong.
//  (2) RAW: a VALU instruction that WRITES a source register NOPS slots in front of the MFMA.  Measured: 18 % wrong results with 0 or
//      1 slot, none from 2 on -- the MFMA needs two wait states behind a VALU write of its sources.  This is what made several
//      re-schedules of the tower loop "almost right": the compiler sank an accumulator's zeroing, or a register copy, to just in front
//      of an inline-asm MFMA.  tests/test_kernel_resources.py checks the disassembly for it; conv_igemm.hip pins the zeroing.
// usage (GPU box): hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tests/tools/mfma_war_probe.hip -o /tmp/war && /tmp/war
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int FILL, int NOPS, int SRC>      // SRC 0: overwrite SrcB, 1: overwrite SrcA
__global__ __launch_bounds__(512) void probe(int iters, unsigned* bad) {
    __shared__ __attribute__((aligned(16))) short twos[64 * 8];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 8; i += 512) twos[i] = 0x4000;          // bf16 2.0
    __syncthreads();
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = 0x3F80;                             // bf16 1.0
    bf16x8 a = ones, b = ones;
    f32x4 fill[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) fill[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned wrong = 0;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(twos + lane * 8);
    for (int it = 0; it < iters; ++it) {
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FILL; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(fill[i & 7]) : "v"(ones), "v"(ones));
        if (SRC == 0) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, %0\n\t"
                         ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                         "ds_read_b128 %1, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_nop 15\n\ts_nop 15"
                         : "+v"(t), "+v"(b) : "v"(a), "v"(addr), "n"(NOPS) : "memory");
            b = ones;
        } else {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                         ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                         "ds_read_b128 %1, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_nop 15\n\ts_nop 15"
                         : "+v"(t), "+v"(a) : "v"(b), "v"(addr), "n"(NOPS) : "memory");
            a = ones;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) wrong += (t[r] != 32.0f) ? 1u : 0u;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += fill[i][0];
    if (s == -1.f) wrong += 1;                     // keep the fill MFMAs alive
    if (wrong) atomicAdd(bad, wrong);
}

// RAW probe: a VALU instruction writes the MFMA's SrcB (v_mov: twos -> ones) NOPS slots before the MFMA reads it
template <int NOPS>
__global__ __launch_bounds__(512) void probe_raw(int iters, unsigned* bad) {
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = 0x3F80;
    unsigned wrong = 0;
    for (int it = 0; it < iters; ++it) {
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("v_mov_b32 v100, %2\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, %2\n\t"      // B = twos
                     "s_nop 7\n\ts_nop 7\n\t"
                     "v_mov_b32 v100, %3\n\t"                                    // VALU write of the first SrcB register (two of eight elements -> 1.0)
                     ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                     "v_mfma_f32_16x16x32_bf16 %0, %1, v[100:103], %0\n\t"
                     "s_nop 15\n\ts_nop 15"
                     : "+v"(t) : "v"(ones), "v"(0x40004000u), "v"(0x3F803F80u), "n"(NOPS) : "v100", "v101", "v102", "v103");
        // every lane's first two k elements became 1.0, the other six stayed 2.0: 2*1 + 6*2 = 14 per lane, 4 lanes per k-row ... = 56
#pragma unroll
        for (int r = 0; r < 4; ++r) wrong += (t[r] != 56.0f) ? 1u : 0u;
    }
    if (wrong) atomicAdd(bad, wrong);
}

template <int NOPS>
static void run_raw(unsigned* d_bad) {
    unsigned zero = 0, h = 0;
    hipMemcpy(d_bad, &zero, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((probe_raw<NOPS>), dim3(256), dim3(512), 0, 0, 20000, d_bad);
    hipDeviceSynchronize();
    hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    printf("VALU write of SrcB %d slot(s) before the MFMA: %u wrong results of %llu\n", NOPS, h, 256ull * 512 * 20000 * 4);
}

template <int FILL, int NOPS, int SRC>
static void run(unsigned* d_bad) {
    unsigned zero = 0, h = 0;
    hipMemcpy(d_bad, &zero, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((probe<FILL, NOPS, SRC>), dim3(256), dim3(512), 0, 0, 20000, d_bad);
    hipDeviceSynchronize();
    hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost);
    printf("overwrite Src%c, %2d MFMAs queued ahead, ds_read %d slot(s) behind the MFMA: %u wrong results of %llu\n", SRC ? 'A' : 'B', FILL, NOPS, h,
           256ull * 512 * 20000 * 4);
}

int main() {
    unsigned* d_bad; hipMalloc(&d_bad, 4);
    run<4, 0, 0>(d_bad); run<8, 0, 0>(d_bad); run<16, 0, 0>(d_bad); run<32, 0, 0>(d_bad);
    run<8, 1, 0>(d_bad); run<8, 4, 0>(d_bad); run<32, 4, 0>(d_bad);
    run<8, 0, 1>(d_bad); run<32, 0, 1>(d_bad); run<32, 4, 1>(d_bad);
    run_raw<0>(d_bad); run_raw<1>(d_bad); run_raw<2>(d_bad); run_raw<3>(d_bad); run_raw<4>(d_bad); run_raw<6>(d_bad); run_raw<8>(d_bad);
    return 0;
}
