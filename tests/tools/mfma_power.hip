// Sustained MFMA rate under the power limit: register-resident loops of v_mfma_f32_32x32x16_bf16 and
// v_mfma_f32_16x16x32_bf16 on random vs zero operands (no memory traffic at all), 2 waves per SIMD.
// usage (GPU box): hipcc --offload-arch=gfx950 -O3 tests/tools/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>   // 0: 32x32x16, 8 accumulators; 1: 16x16x32, 32 accumulators (the same 128 accumulator registers)
__global__ __launch_bounds__(512) void mfma_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(i * 64 + lane)]; b[i] = src[((8 + i) * 64 + lane)]; }
    if constexpr (SHAPE == 0) {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + k) & 7], b[(i * 3 + k) & 7], acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + k) & 7], b[(i * 3 + k) & 7], acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

int main() {
    const int blocks = 256 * 1, iters = 4000;
    std::vector<unsigned short> h(16 * 64 * 8);
    bf16x8* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, blocks * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int data = 0; data < 3; ++data) {
        srand(1);
        for (auto& v : h) {
            if (data == 0) v = 0;
            else if (data == 1) { float f = ((rand() % 2001) - 1000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }     // random in [-1, 1]
            else { float f = (rand() % 100 < 65) ? 0.f : (rand() % 1000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; } // 65 % zeros, positive (post-ReLU/dropout like)
        }
        hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int shape = 0; shape < 2; ++shape) {
            double best = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(512), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(512), 0, 0, d, o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // per wave and iteration: shape 0: 64 MFMAs x 32*32*16*2 FLOP; shape 1: 128 MFMAs x 16*16*32*2 FLOP
                const double flops = (double)blocks * 8 * iters * (shape == 0 ? 64.0 * 32768 : 128.0 * 16384);
                const double tf = flops / (ms * 1e-3) / 1e12;
                if (rep > 0 && tf > best) best = tf;
            }
            printf("data=%s shape=%s : %.0f TFLOP/s\n", data == 0 ? "zeros" : data == 1 ? "random[-1,1]" : "65%-zero-positive", shape == 0 ? "32x32x16" : "16x16x32", best);
        }
    }
    return 0;
}
