// CU-mask lab (not part of the product): which (XCC, SE, CU) run the blocks of a kernel launched on a stream made with
// hipExtStreamCreateWithCUMask, for a few mask patterns?  Prints the number of distinct CUs used per XCC.
//   hipcc --offload-arch=gfx950 -O3 tools/cumask_lab.hip -o tools/_bin/cumask_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <map>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the block alive a little so that the grid spreads over every enabled CU
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

int main() {
    const int nblk = 4096;
    unsigned* d;
    CK(hipMalloc(&d, nblk * 8));
    std::vector<unsigned> h(2 * nblk);
    struct Pat { const char* name; unsigned m[8]; };
    std::vector<Pat> pats;
    { Pat p{"all 256", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}}; pats.push_back(p); }
    { Pat p{"low 64 bits", {~0u, ~0u, 0, 0, 0, 0, 0, 0}}; pats.push_back(p); }
    { Pat p{"high 192 bits", {0, 0, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}}; pats.push_back(p); }
    { Pat p{"every 4th bit", {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u}}; pats.push_back(p); }
    { Pat p{"low byte of each word", {0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu}}; pats.push_back(p); }
    for (const Pat& p : pats) {
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, 8, p.m));
        hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, s, d, 2000);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost));
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int i = 0; i < nblk; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        int total = 0;
        printf("%-24s:", p.name);
        for (auto& kv : per_xcc) { printf(" xcc%u=%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
        printf("  total %d CUs\n", total);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
