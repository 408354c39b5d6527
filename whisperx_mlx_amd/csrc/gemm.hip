// Dense fp16 GEMM on the gfx950 matrix cores (MFMA 16x16x32 f16, fp32 accumulate).
//
//   out[y][x] = epilogue( sum_k X[x][k] * Y[y][k] )
//
// X and Y are both K-contiguous row-major fp16 matrices ([rows][K] with a row
// stride), which is exactly the MFMA operand layout: 8 consecutive k per lane.
// X plays the MFMA "A" role (its row index lands in the accumulator's register
// dimension: 4 consecutive x per lane -> one 8-byte store), Y the "B" role.
//   * activations @ weight^T, row-major output:  X = weight [N][K], Y = act [M][K]
//   * transposed output (V^T, K^T for attention): X = act,  Y = weight
//   * grouped convolution (wav2vec2 positional conv): Y rows are gathered, K = (tap, channel)
//
// Tile 128(x) x 128(y) x 64(k), 256 threads = 2x2 waves, each wave 64x64 as 4x4
// MFMA tiles.  LDS rows are 128 B with the 16-byte chunk XOR-swizzled by (row&7)
// so every ds_read_b128 fragment read is bank-conflict free (MI355X LDS: 64
// dword banks for b128).  Global->register->LDS staging with the next K-tile's
// loads in flight under the current tile's MFMAs; one barrier per K-tile.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

namespace {

constexpr int BX = 128, BY = 128, BK = 64;
constexpr int TILE_BYTES = BX * BK * 2;   // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * (BK * 2) + ((chunk ^ (row & 7)) << 4);
}

template <bool GELU>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[4][4], int x0, int y0, int wx, int wy,
                                              int fr, int fq, int bz) {
    // epilogue: lane holds x = xb + 0..3 (contiguous), y = yb
    h16* __restrict__ out = p.out + (long)bz * p.strideOut;
    const h16* __restrict__ R = p.R ? p.R + (long)bz * p.strideR : nullptr;
    const h16* __restrict__ bias = p.bias ? p.bias + (long)bz * p.strideBias : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + wy * 64 + j * 16 + fr;
        if (y >= p.RY) continue;
        const float by_ = (bias && p.bias_on_y) ? (float)bias[y] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = x0 + wx * 64 + i * 16 + fq * 4;
            if (x >= p.RX) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + by_;
            if (x + 3 < p.RX) {
                if (bias && !p.bias_on_y) {
                    const half4 b4 = *reinterpret_cast<const half4*>(bias + x);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)b4[r];
                }
                if (GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
                }
                if (R) {
                    const half4 r4 = *reinterpret_cast<const half4*>(R + (long)y * p.ldr + x);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)r4[r];
                }
                half4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (h16)v[r];
                long oaddr = (long)y * p.ldo + x;
                if (p.hs_T > 0) {
                    // head-split store: [part][batch][head][t][64] so that one (batch, head) is contiguous
                    const int bb = y / p.hs_T, tt = y - bb * p.hs_T;
                    const int part = x / p.hs_d, xr = x - part * p.hs_d;
                    oaddr = (long)part * p.hs_part_stride + (((long)bb * p.hs_H + (xr >> 6)) * p.hs_T + tt) * 64 + (xr & 63);
                }
                *reinterpret_cast<half4*>(out + oaddr) = o;
            } else {
                for (int r = 0; r < 4 && x + r < p.RX; ++r) {
                    float t = v[r];
                    if (bias && !p.bias_on_y) t += (float)bias[x + r];
                    if (GELU) t = gelu_f(t);
                    if (R) t += (float)R[(long)y * p.ldr + x + r];
                    out[(long)y * p.ldo + x + r] = (h16)t;
                }
            }
        }
    }
}

template <bool GELU, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wx = wave & 1, wy = wave >> 1;

    const int ntx = (p.RX + BX - 1) / BX;
    const int nty = (p.RY + BY - 1) / BY;
    // L2-aware tile order.  Blocks are dealt round-robin over the 8 XCDs (private 4 MiB L2 each);
    // xcd_remap gives every XCD one contiguous run of logical tiles, and inside a run tiles walk
    // GY consecutive y-tiles (activation rows) for one x-tile before moving to the next x-tile:
    // the ~64 blocks an XCD has in flight then share GY activation panels and a sliding window
    // of weight panels that fit its L2, instead of cycling through every weight panel per row
    // of tiles (measured: FETCH_SIZE 15x the algorithmic bytes with the plain row-major order).
    const int tile = xcd_remap(blockIdx.x, ntx * nty);
    constexpr int GY = 8;
    const int per_group = GY * ntx;
    const int grp = tile / per_group, rem = tile - grp * per_group;
    const int gcnt = min(GY, nty - grp * GY);
    const int tx = rem / gcnt, ty = grp * GY + rem - tx * gcnt;
    const int x0 = tx * BX, y0 = ty * BY;
    const int bz = blockIdx.z;

    const h16* __restrict__ X = p.X + (long)bz * p.strideX;
    const h16* __restrict__ Y = p.Y + (long)bz * p.strideY;

    // staging assignment: 4 chunks (16 B) per operand per thread
    int srow[4], sch[4];
    const h16* gx[4];
    const h16* gy[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = tid + 256 * q;
        srow[q] = c >> 3;
        sch[q] = c & 7;
        const int rx = min(x0 + srow[q], p.RX - 1);
        const int ry = min(y0 + srow[q], p.RY - 1);
        gx[q] = X + (long)rx * p.ldx + sch[q] * 8;
        gy[q] = Y + (long)ry * p.ldy + (GATHER ? 0 : sch[q] * 8);
    }

    half8 rx_[4], ry_[4];
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = (k0 + sch[q] * 8) < p.K;
            rx_[q] = ok ? *reinterpret_cast<const half8*>(gx[q] + k0) : zero8;
            if (GATHER) {
                // K index = (tap, channel-in-group): taps are `y_gather_step` elements apart
                const int kc = (k0 >> 3) + sch[q];
                const int tap = kc / p.y_gather_group;
                const long off = (long)tap * p.y_gather_step + (kc - tap * p.y_gather_group) * 8;
                ry_[q] = ok ? *reinterpret_cast<const half8*>(gy[q] + off) : zero8;
            } else {
                ry_[q] = ok ? *reinterpret_cast<const half8*>(gy[q] + k0) : zero8;
            }
        }
    };
    auto sstore = [&](int buf) {
        char* bx = smem + buf * 2 * TILE_BYTES;
        char* by = bx + TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<half8*>(bx + lds_off(srow[q], sch[q])) = rx_[q];
            *reinterpret_cast<half8*>(by + lds_off(srow[q], sch[q])) = ry_[q];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    sstore(0);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const char* bx = smem + (kt & 1) * 2 * TILE_BYTES;
        const char* by = bx + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const half8*>(bx + lds_off(wx * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const half8*>(by + lds_off(wy * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore((kt + 1) & 1);
        __syncthreads();
    }

    gemm_epilogue<GELU>(p, acc, x0, y0, wx, wy, fr, fq, bz);
}


// ---------------------------------------------------------------------------------------
// LDS-DMA variant: the K-tile staging uses global_load_lds (16 B per lane, straight into LDS,
// no VGPR round trip and no ds_write pass).  The LDS destination of one wave instruction is
// lane-linear (wave-uniform base + lane * 16 B = 8 rows x 128 B), so the XOR swizzle that keeps
// the fragment reads conflict free is applied to the per-lane SOURCE address instead: LDS slot
// (row, c') receives global chunk c' ^ (row & 7), and the reads keep using lds_off().
__device__ __attribute__((aligned(16))) const unsigned wx_zero16[4] = {0u, 0u, 0u, 0u};

template <bool GELU, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wx = wave & 1, wy = wave >> 1;

    const int ntx = (p.RX + BX - 1) / BX;
    const int nty = (p.RY + BY - 1) / BY;
    const int tile = xcd_remap(blockIdx.x, ntx * nty);
    constexpr int GY = 8;
    const int per_group = GY * ntx;
    const int grp = tile / per_group, rem = tile - grp * per_group;
    const int gcnt = min(GY, nty - grp * GY);
    const int tx = rem / gcnt, ty = grp * GY + rem - tx * gcnt;
    const int x0 = tx * BX, y0 = ty * BY;
    const int bz = blockIdx.z;
    const h16* __restrict__ X = p.X + (long)bz * p.strideX;
    const h16* __restrict__ Y = p.Y + (long)bz * p.strideY;

    // this wave stages rows [wave*32, wave*32+32) of both operand tiles: 4 DMA instructions each
    const int r_in = lane >> 3, cs = lane & 7;
    const int csrc = cs ^ r_in;                       // source chunk for LDS slot (row, cs): row & 7 == r_in
    const h16* gx[4];
    const h16* gy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + r_in;
        gx[j] = X + (long)min(x0 + row, p.RX - 1) * p.ldx;
        gy[j] = Y + (long)min(y0 + row, p.RY - 1) * p.ldy;
    }
    const h16* zsrc = reinterpret_cast<const h16*>(wx_zero16);
    auto stage = [&](int k0, int buf) {
        char* bx = smem + buf * 2 * TILE_BYTES + wave * 32 * (BK * 2);
        char* by = bx + TILE_BYTES;
        const int kel = k0 + csrc * 8;
        const bool ok = kel < p.K;
        long yoff = kel;
        if (GATHER) {
            const int kc = kel >> 3;
            const int tap = kc / p.y_gather_group;
            yoff = (long)tap * p.y_gather_step + (kc - tap * p.y_gather_group) * 8;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const h16* sx = ok ? gx[j] + kel : zsrc;
            const h16* sy = ok ? gy[j] + yoff : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sx,
                                             (__attribute__((address_space(3))) void*)(bx + j * 8 * (BK * 2)), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sy,
                                             (__attribute__((address_space(3))) void*)(by + j * 8 * (BK * 2)), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage((kt + 1) * BK, (kt + 1) & 1);
        const char* bx = smem + (kt & 1) * 2 * TILE_BYTES;
        const char* by = bx + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const half8*>(bx + lds_off(wx * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const half8*>(by + lds_off(wy * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    gemm_epilogue<GELU>(p, acc, x0, y0, wx, wy, fr, fq, bz);
}

}  // namespace

hipError_t launch_gemm_f16(const GemmArgs& a, int batch, bool gelu, hipStream_t s) {
    const int ntx = (a.RX + BX - 1) / BX, nty = (a.RY + BY - 1) / BY;
    dim3 grid(ntx * nty, 1, batch), block(256);
    const size_t lds = 4 * TILE_BYTES;
    static const bool regstage = getenv("WX_GEMM_REGSTAGE") != nullptr;
    if (!regstage) {
        if (a.y_gather_group > 0) {
            if (gelu)
                hipLaunchKernelGGL((gemm_glds_kernel<true, true>), grid, block, lds, s, a);
            else
                hipLaunchKernelGGL((gemm_glds_kernel<false, true>), grid, block, lds, s, a);
        } else if (gelu)
            hipLaunchKernelGGL((gemm_glds_kernel<true, false>), grid, block, lds, s, a);
        else
            hipLaunchKernelGGL((gemm_glds_kernel<false, false>), grid, block, lds, s, a);
        return hipGetLastError();
    }
    if (a.y_gather_group > 0) {
        if (gelu)
            hipLaunchKernelGGL((gemm_f16_kernel<true, true>), grid, block, lds, s, a);
        else
            hipLaunchKernelGGL((gemm_f16_kernel<false, true>), grid, block, lds, s, a);
    } else if (gelu)
        hipLaunchKernelGGL((gemm_f16_kernel<true, false>), grid, block, lds, s, a);
    else
        hipLaunchKernelGGL((gemm_f16_kernel<false, false>), grid, block, lds, s, a);
    return hipGetLastError();
}
