// Probe (GPU box): which compute units does bit i of a hipExtStreamCreateWithCUMask mask select on MI355X (8 XCDs x 32 CUs)?
//   hipcc --offload-arch=gfx950 -O2 tests/tools/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
// For a handful of masks it launches many one-wave workgroups that spin for ~20 us and record (XCC_ID, SE, SH, CU) from the
// hardware-id registers, then prints the distinct CUs per XCD the mask reached.  Used to lay out the engine's front / tower
// partition (engine.hip, BOD_OVERLAP) symmetrically over the XCDs: the tower kernel's XCD-aware tile order assumes workgroup
// b lands on XCD b % 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <vector>

__global__ void where_am_i(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char* label, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", label, hipGetErrorString(e)); return; }
    const int nb = 8192;
    uint32_t* d = nullptr;
    hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(where_am_i, dim3(nb), dim3(64), 0, s, d, 2000);      // 2000 ticks of 100 MHz = 20 us
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per_xcc;       // xcc -> {se * 32 + sh * 16 + cu}
    std::map<int, int> first_wg_xcc;
    for (int b = 0; b < nb; ++b) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
        const int cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per_xcc[(int)xcc].insert(se * 32 + sh * 16 + cu);
        if (b < 16) first_wg_xcc[b] = (int)xcc;
    }
    int total = 0;
    printf("%s:", label);
    for (auto& kv : per_xcc) { printf(" xcc%d=%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  total %d CUs; wg->xcc of the first 16 workgroups:", total);
    for (auto& kv : first_wg_xcc) printf(" %d", kv.second);
    printf("\n");
    if (total <= 64) {
        for (auto& kv : per_xcc) {
            printf("    xcc%d:", kv.first);
            for (int id : kv.second) printf(" se%d.sh%d.cu%d", id / 32, (id / 16) & 1, id & 15);
            printf("\n");
        }
    }
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
    const int words = 8;                         // 256 bits
    auto mk = [&](auto pred) { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < 256; ++i) if (pred(i)) m[i >> 5] |= 1u << (i & 31); return m; };
    run("all 256 bits", mk([](int) { return true; }));
    run("bits 0..31", mk([](int i) { return i < 32; }));
    run("bits 0..7", mk([](int i) { return i < 8; }));
    run("bits 8..15", mk([](int i) { return i >= 8 && i < 16; }));
    run("bits i%8==0", mk([](int i) { return i % 8 == 0; }));
    run("bits i%32<4", mk([](int i) { return i % 32 < 4; }));
    run("bits 0..63", mk([](int i) { return i < 64; }));
    run("bits 32..255", mk([](int i) { return i >= 32; }));
    run("bits 224..255", mk([](int i) { return i >= 224; }));
    return 0;
}
