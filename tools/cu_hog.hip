// CU hog lab (not part of the product): occupy N compute units -- one 1024-thread block with all 160 KB of LDS each, so
// that no LDS-using block of another kernel fits beside it -- until a flag is raised or a deadline passes, while the
// product's decode runs on the rest of the chip.  Answers: how many CUs does the HBM-bound decode need?  (tools/ab_cu_hog.py)
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/cu_hog.hip -o tools/_bin/libcuhog.so
#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(1024) hog_kernel(const int* stop, long long max_ticks, unsigned* where) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[2 * blockIdx.x] = hw;
        where[2 * blockIdx.x + 1] = xcc;
        lds[0] = 1;
    }
    // one lane of the block polls (host memory: every poll crosses PCIe -- 512 polling waves starved the whole device in
    // the first version of this lab), every 100 us; the other waves wait at the barrier and cost nothing.
    // Every wave reaches the exit: the deadline is on the constant 100 MHz clock whatever the flag does.
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < max_ticks) {
            if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
            for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
    __syncthreads();
}

extern "C" int hog_launch(void* stream, int n_blocks, double max_seconds, const int* stop_flag_dev, unsigned* where_dev) {
    static bool attr = false;
    const int lds = 160 * 1024;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
        attr = true;
    }
    hipLaunchKernelGGL(hog_kernel, dim3(n_blocks), dim3(1024), lds, (hipStream_t)stream, stop_flag_dev,
                       (long long)(max_seconds * 1e8), where_dev);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
