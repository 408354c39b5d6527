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
// Two kernels, chosen by shape in launch_gemm_f16:
//   gemm_8phase_kernel  256 x 256 x 64 tile, 8 waves, one block per CU, LDS-DMA staged two to four phases ahead
//                       with counted vmcnt and raw barriers, LDS-transposed epilogue (the encoder's GEMMs)
//   gemm_glds_kernel    128 x 128 x 64 tile, 4 waves, LDS-DMA double buffer (K % 64 != 0, narrow N)
// LDS rows are 128 B with the 16-byte chunk XOR-swizzled by (row&7) so every ds_read_b128 fragment read is
// bank-conflict free (MI355X LDS: 64 dword banks for b128).
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace {

constexpr int BX = 128, BY = 128, BK = 64;
constexpr int TILE_BYTES = BX * BK * 2;   // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * (BK * 2) + ((chunk ^ (row & 7)) << 4);
}

template <bool GELU, int NI = 4, int NJ = 4>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[NI][NJ], int xw, int yw,
                                              int fr, int fq, int bz) {
    // epilogue: lane holds x = xb + 0..3 (contiguous), y = yb
    h16* __restrict__ out = p.out + (long)bz * p.strideOut;
    const h16* __restrict__ R = p.R ? p.R + (long)bz * p.strideR : nullptr;
    const h16* __restrict__ bias = p.bias ? p.bias + (long)bz * p.strideBias : nullptr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int y = yw + j * 16 + fr;
        if (y >= p.RY) continue;
        const float by_ = (bias && p.bias_on_y) ? (float)bias[y] : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int x = xw + i * 16 + fq * 4;
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
                if (x < p.xscale_cols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= p.xscale;
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
                    if (x + r < p.xscale_cols) t *= p.xscale;
                    if (GELU) t = gelu_f(t);
                    if (R) t += (float)R[(long)y * p.ldr + x + r];
                    out[(long)y * p.ldo + x + r] = (h16)t;
                }
            }
        }
    }
}



// ---------------------------------------------------------------------------------------
// LDS-DMA variant: the K-tile staging uses global_load_lds (16 B per lane, straight into LDS,
// no VGPR round trip and no ds_write pass).  The LDS destination of one wave instruction is
// lane-linear (wave-uniform base + lane * 16 B = 8 rows x 128 B), so the XOR swizzle that keeps
// the fragment reads conflict free is applied to the per-lane SOURCE address instead: LDS slot
// (row, c') receives global chunk c' ^ (row & 7), and the reads keep using lds_off().
__device__ __attribute__((aligned(16))) const unsigned wx_zero16[4] = {0u, 0u, 0u, 0u};

// NARROW (RX <= 64: the grouped positional conv of wav2vec2 has 48 output channels per group): the tile is 64 x 128, the four
// waves sit side by side along y (64 x 32 each) instead of 2 x 2 -- with the 2 x 2 layout the two waves of the upper x half
// multiplied rows that do not exist (round 4: 3.3 -> 1.8 ms for the 16 groups of 64 x 30 s).
template <bool GELU, bool GATHER, bool NARROW = false>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wx = NARROW ? 0 : (wave & 1), wy = NARROW ? wave : (wave >> 1);
    constexpr int NJ = NARROW ? 2 : 4;          // 16-row y blocks per wave
    const int ywave = NARROW ? wave * 32 : wy * 64;

    const int ntx = (p.RX + BX - 1) / BX;
    const int nty = (p.RY + BY - 1) / BY;
    const int tile = xcd_remap(blockIdx.x, ntx * nty);
    constexpr int GY = 8;
    const int per_group = GY * ntx;
    const int grp = tile / per_group, rem = tile - grp * per_group;
    const int gcnt = min(GY, nty - grp * GY);
    const int tx = rem / gcnt, ty = grp * GY + rem - tx * gcnt;
    const int x0 = tx * BX, y0 = ty * BY;
    int bz = blockIdx.z;
    if (p.zsplit > 0) {          // two-level batch: fold the second level into the base pointers (p is this block's own copy)
        const int z2 = bz / p.zsplit;
        bz -= z2 * p.zsplit;
        p.X += (long)z2 * p.strideX2;
        p.Y += (long)z2 * p.strideY2;
        if (p.bias) p.bias += (long)z2 * p.strideBias2;
        if (p.R) p.R += (long)z2 * p.strideR2;
        p.out += (long)z2 * p.strideOut2;
    }
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
            if (!NARROW || wave < 2)        // NARROW: only x rows 0..63 are read (wave-uniform)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sx,
                                                 (__attribute__((address_space(3))) void*)(bx + j * 8 * (BK * 2)), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sy,
                                             (__attribute__((address_space(3))) void*)(by + j * 8 * (BK * 2)), 16, 0, 0);
        }
    };

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
            half8 a[4], b[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const half8*>(bx + lds_off(wx * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = *reinterpret_cast<const half8*>(by + lds_off(ywave + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    gemm_epilogue<GELU, 4, NJ>(p, acc, x0 + wx * 64, y0 + ywave, fr, fq, bz);
}


// Epilogue of the 256^2 kernel: the accumulator layout (4 consecutive x per lane, 16 rows per
// instruction) gives 8-byte stores in 32-byte runs, which is store-issue bound (15-20 us per tile
// measured, a third of a K = 1280 GEMM).  Each wave instead transposes its 128 x 64 block through
// its own 17 KiB of the (now idle) staging LDS, 64 x-columns at a time in fp32, and writes full
// 128-byte lines with one 16-byte store per lane; bias / GELU / residual are applied on the read
// side in fp32 exactly as in gemm_epilogue (single rounding to fp16).
constexpr int EPI_ROW = 272;                 // 64 fp32 + 16 B pad: conflict-free b128 writes
constexpr int EPI_WAVE = 64 * EPI_ROW;       // 17 KiB per wave

template <bool GELU>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs& p, f32x4 (&acc)[8][4], int xw, int yw, int lane,
                                                  char* region, int bz) {
    h16* __restrict__ out = p.out + (long)bz * p.strideOut;
    const h16* __restrict__ R = p.R ? p.R + (long)bz * p.strideR : nullptr;
    const h16* __restrict__ bias = p.bias ? p.bias + (long)bz * p.strideBias : nullptr;
    const int fr = lane & 15, fq = lane >> 4;
    const int rrow = lane >> 3, c8 = lane & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(region + (j * 16 + fr) * EPI_ROW + (i * 16 + fq * 4) * 4) = acc[h * 4 + i][j];
        const int x = xw + h * 64 + c8 * 8;
        float bx[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bx[r] = 0.f;
        if (bias && !p.bias_on_y && x + 7 < p.RX) {
            const half8 b8 = *reinterpret_cast<const half8*>(bias + x);
#pragma unroll
            for (int r = 0; r < 8; ++r) bx[r] = (float)b8[r];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + rrow;
            const int y = yw + row;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(region + row * EPI_ROW + c8 * 32);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(region + row * EPI_ROW + c8 * 32 + 16);
            if (y >= p.RY || x >= p.RX) continue;
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const float by_ = (bias && p.bias_on_y) ? (float)bias[y] : 0.f;
            if (x + 7 < p.RX) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += by_ + bx[r];
                if (x < p.xscale_cols) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] *= p.xscale;
                }
                if (GELU) {
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const wx_f2 g = gelu_f2((wx_f2){v[r], v[r + 1]});
                        v[r] = g[0];
                        v[r + 1] = g[1];
                    }
                }
                if (R) {
                    const half8 r8 = *reinterpret_cast<const half8*>(R + (long)y * p.ldr + x);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)r8[r];
                }
                half8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = (h16)v[r];
                long oaddr = (long)y * p.ldo + x;
                if (p.hs_T > 0) {
                    const int bb = y / p.hs_T, tt = y - bb * p.hs_T;
                    const int part = x / p.hs_d, xr = x - part * p.hs_d;
                    oaddr = (long)part * p.hs_part_stride + (((long)bb * p.hs_H + (xr >> 6)) * p.hs_T + tt) * 64 + (xr & 63);
                }
                *reinterpret_cast<half8*>(out + oaddr) = o;
            } else {
                for (int r = 0; r < 8 && x + r < p.RX; ++r) {
                    float t = v[r] + by_;
                    if (bias && !p.bias_on_y) t += (float)bias[x + r];
                    if (x + r < p.xscale_cols) t *= p.xscale;
                    if (GELU) t = gelu_f(t);
                    if (R) t += (float)R[(long)y * p.ldr + x + r];
                    out[(long)y * p.ldo + x + r] = (h16)t;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves as 2 (x) x 4 (y), 128 x 64 of the output per wave, one block per
// CU, 128 KiB of LDS: two K-tile buffers, each four 16 KiB half-tiles (X rows 0-127 / 128-255,
// Y rows 0-127 / 128-255).  A K-tile is four phases of 16 MFMAs (one quadrant of the wave's
// output x K = 64); every phase also issues one half-tile of LDS-DMA (2 instructions per lane).
// The two wave groups (x half 0 / 1 = one wave of each SIMD) run one barrier apart, so while one
// group is in its MFMA cluster the other issues its ds_reads and DMA.
//
// Staging order, K-tile t:  P0 X_lo(t+1)  P1 X_hi(t+1)  P2 Y_lo(t+2)  P3 Y_hi(t+2), then
// s_waitcnt vmcnt(4): everything up to X_hi(t+1) has landed, the two Y half-tiles of t+2 stay in
// flight across the barriers (never vmcnt(0) in the loop).  Hazards:
//   RAW  the wait sits before P3's first barrier, the reads of tile t+1 start in P0(t+1), after a
//        barrier both groups reached behind their waits.
//   WAR  Y(t) is read in P0 only and an explicit lgkmcnt(0) before P1's first barrier retires the
//        reads, so Y(t+2) may overwrite it from P2 on; X_lo(t) / X_hi(t) are last read in P3 by
//        group 0 / group 1 and consumed by P3's MFMAs, which end before the slot in which the
//        first wave stages X_lo(t+2) (P0 of t+1) / X_hi(t+2) (P1 of t+1).
// The four phases are (x rows 0-63 | 64-127) x (k 0-31 | 32-63) against all 64 y columns: 12 + 4 +
// 4 + 4 ds_read_b128 per wave.
// Tiles past the end of K are staged from the last K-tile again (never read), which keeps every
// vmcnt count static.  Requires K % 64 == 0.
constexpr int B8 = 256;
constexpr int HALF8 = 128 * BK * 2;          // 16 KiB half-tile
constexpr int BUF8 = 4 * HALF8;              // 64 KiB per K-tile buffer
constexpr int LDS8 = 8 * EPI_WAVE;            // 136 KiB: 128 KiB of staging, reused (+8 KiB) by the epilogue

// PERSIST (GemmArgs::max_blocks): the grid is capped and every block walks tiles blockIdx.x, + gridDim.x, ... -- a launch
// that occupies at most gridDim.x CUs for its whole duration (the encoder beside other passes' decode).  The tile body is
// the same code; a block-wide barrier between tiles hands the LDS from one tile's epilogue to the next tile's staging.
template <bool GELU, bool GATHER, bool PERSIST = false>
__global__ __launch_bounds__(512) void gemm_8phase_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int ntx = (p.RX + B8 - 1) / B8;
    const int nty = (p.RY + B8 - 1) / B8;
  for (int tile_lin = blockIdx.x; tile_lin < ntx * nty; tile_lin += PERSIST ? (int)gridDim.x : ntx * nty) {
    const int tile = xcd_remap(tile_lin, ntx * nty);
    constexpr int GY = 4;
    const int per_group = GY * ntx;
    const int grp = tile / per_group, rem = tile - grp * per_group;
    const int gcnt = min(GY, nty - grp * GY);
    const int tx = rem / gcnt, ty = grp * GY + rem - tx * gcnt;
    const int x0 = tx * B8, y0 = ty * B8;
    const int bz = blockIdx.z;
    const char* __restrict__ Xb = reinterpret_cast<const char*>(p.X + (long)bz * p.strideX);
    const char* __restrict__ Yb = reinterpret_cast<const char*>(p.Y + (long)bz * p.strideY);

    // per-lane byte offsets of the 16-B piece each DMA instruction fetches (source-side swizzle)
    const int r_in = lane >> 3, csrc = (lane & 7) ^ r_in;
    unsigned voff[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = h * 128 + j * 64 + wave * 8 + r_in;
            voff[h][j] = (unsigned)(((long)min(x0 + row, p.RX - 1) * p.ldx + csrc * 8) * 2);
            voff[2 + h][j] = (unsigned)(((long)min(y0 + row, p.RY - 1) * p.ldy + (GATHER ? 0 : csrc * 8)) * 2);
        }
    const int nk = p.K / BK;
    auto stage = [&](auto WHICH, int T, int buf) {
        constexpr int which = decltype(WHICH)::value;
        const int k0 = min(T, nk - 1) * BK;
        char* dst = smem + buf * BUF8 + which * HALF8 + wave * 1024;
        const char* src = (which < 2 ? Xb : Yb);
        unsigned koff = (unsigned)k0 * 2;
        if (GATHER && which >= 2) {
            const int kc = (k0 >> 3) + csrc;
            const int tap = kc / p.y_gather_group;
            koff = (unsigned)(((long)tap * p.y_gather_step + (kc - tap * p.y_gather_group) * 8) * 2);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (voff[which][j] + koff)),
                                             (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    // fragment byte offsets inside a half-tile: row*128 + ((chunk ^ (row&7)) << 4), chunk = ks*4 + fq
    const int sw0 = ((fq ^ (fr & 7)) << 4), sw1 = (((4 + fq) ^ (fr & 7)) << 4);
    const int aoff = wr * HALF8 + fr * 128;
    const int boff = 2 * HALF8 + (wc >> 1) * HALF8 + ((wc & 1) * 64 + fr) * 128;

    stage(I0{}, 0, 0);
    stage(I1{}, 0, 0);
    stage(I2{}, 0, 0);
    stage(I3{}, 0, 0);
    stage(I2{}, 1, 1);
    stage(I3{}, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // run the second wave group one barrier behind

    half8 a[4], b[4][2];
    auto ktile = [&](auto BUFC, int t) {
        constexpr int BUF = decltype(BUFC)::value;
        const char* base = smem + BUF * BUF8;
        auto read_a = [&](int half, int sw) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(base + aoff + (half * 4 + i) * 2048 + sw);
        };
        auto mfma16 = [&](int half, int ks) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j][ks], acc[half * 4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        // ---- P0: X rows 0-63 k 0-31 and ALL of Y (k 0-31 first); stage X_lo(t+1)
        read_a(0, sw0);
        {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j][0] = *reinterpret_cast<const half8*>(base + boff + j * 2048 + sw0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j][1] = *reinterpret_cast<const half8*>(base + boff + j * 2048 + sw1);
        }
        stage(I0{}, t + 1, BUF ^ 1);
        __builtin_amdgcn_s_barrier();
        mfma16(0, 0);
        __builtin_amdgcn_s_barrier();
        // ---- P1: X rows 0-63 k 32-63; every Y read is back (lgkmcnt 0) before Y(t+2) is staged in P2
        read_a(0, sw1);
        stage(I1{}, t + 1, BUF ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma16(0, 1);
        __builtin_amdgcn_s_barrier();
        // ---- P2: X rows 64-127 k 0-31; stage Y_lo(t+2) over this buffer's Y_lo
        read_a(1, sw0);
        stage(I2{}, t + 2, BUF);
        __builtin_amdgcn_s_barrier();
        mfma16(1, 0);
        __builtin_amdgcn_s_barrier();
        // ---- P3: X rows 64-127 k 32-63; stage Y_hi(t+2); retire everything up to X_hi(t+1)
        read_a(1, sw1);
        stage(I3{}, t + 2, BUF);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma16(1, 1);
        __builtin_amdgcn_s_barrier();
    };
    for (int t = 0; t < nk; t += 2) {
        ktile(I0{}, t);
        if (t + 1 < nk) ktile(I1{}, t + 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's DMA has landed and every fragment read is done: LDS is free
    gemm_epilogue_lds<GELU>(p, acc, x0 + wr * 128, y0 + wc * 64, lane, smem + wave * EPI_WAVE, bz);
    if (PERSIST) __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------
// The 256 x 256 kernel as ONE staging pipeline over all the tiles of a block (round 5).  One block per CU walks tiles
// blockIdx.x, + gridDim.x, ...; the k-loop of gemm_8phase_kernel runs on, tile after tile, and the stagings that kernel
// issues past the end of K -- "t + 1" and "t + 2" in a tile's last two k-tiles, dummies there -- fetch the NEXT tile's
// k-tiles 0 and 1 instead.  A tile's prologue (six half-tiles of DMA with nothing to overlap: an HBM round trip and
// 96 KiB through one CU's memory path, plus a workgroup launch per tile) disappears under the previous tile's MFMAs.
//
// What that costs: while a tile's epilogue runs, 96 KiB of the staging LDS already hold the next tile's operands
// (buffer 0 whole, the Y half of buffer 1).  The epilogue therefore transposes through the 64 KiB that are free -- the X
// half of buffer 1, last read in the tile's last phase, and 32 KiB behind the staging buffers (160 KiB of LDS in all) --
// 8 KiB per wave: fp32 [64 y][32 x], four passes over the wave's 128 x-columns, stores of 16 bytes per lane in 64-byte
// runs.  The arithmetic of a pass is gemm_epilogue_lds' (bias, GELU, residual in fp32, one rounding to fp16): the output
// is bit-identical to gemm_8phase_kernel's.  Requires an even number of k-tiles (every tile starts in buffer 0).
constexpr int LDSP = 160 * 1024;
constexpr int EPIP_WAVE = 8192;

// The same epilogue for a wave whose 128 x 64 piece lies inside the output (every tile but the ragged edge), bias along x or
// none: nothing is decided per row, the rows' offsets (and the head-split layout's two divisions) are taken once per tile
// instead of once per (pass, row), a pass's residual loads are issued in front of its LDS round trip, and the passes are
// straight-line code the compiler can overlap.  Same fp32 arithmetic in the same order as the general form below: same bits.
// (Timeline of a K = 1280 tile, tools/lab_gemm_timeline.py: the k-loop 46 800 cycles, the general epilogue 9 800 -- 23 000 with
// GELU -- of which the stores and the LDS round trip explain less than half.)
template <bool GELU, bool HAS_R, bool HS>
__device__ __forceinline__ void gemm_epilogue_fast32(const GemmArgs& p, f32x4 (&acc)[8][4], int xw, int yw, int lane,
                                                     char* region) {
    h16* __restrict__ out = p.out;
    const h16* __restrict__ R = p.R;
    const h16* __restrict__ bias = p.bias;
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = lane >> 2, c4 = lane & 3;
    long orow[4], rrow[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int y = yw + it * 16 + rr;
        if (HS) {
            const int bb = y / p.hs_T, tt = y - bb * p.hs_T;
            orow[it] = ((long)bb * p.hs_H * p.hs_T + tt) * 64;
        } else {
            orow[it] = (long)y * p.ldo;
        }
        rrow[it] = HAS_R ? (long)y * p.ldr : 0;
    }
    const bool has_xs = p.xscale_cols > 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int x = xw + q * 32 + c4 * 8;
        half8 r8[4];
        if (HAS_R) {
#pragma unroll
            for (int it = 0; it < 4; ++it) r8[it] = *reinterpret_cast<const half8*>(R + rrow[it] + x);
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = j * 16 + fr;
                *reinterpret_cast<f32x4*>(region + row * 128 + (((i2 * 4 + fq) ^ ((row >> 1) & 7)) << 4)) = acc[q * 2 + i2][j];
            }
        float bx[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bx[r] = 0.f;
        if (bias) {
            // (requesting the bias -- or the residual -- one pass ahead costs 3-10 spilled registers in this 256-register kernel)
            const half8 b8 = *reinterpret_cast<const half8*>(bias + x);
#pragma unroll
            for (int r = 0; r < 8; ++r) bx[r] = 0.f + (float)b8[r];      // (the general form adds `by_ + bx`, by_ = 0 here)
        }
        long xoff = x;
        if (HS) {
            const int part = x / p.hs_d, xr = x - part * p.hs_d;
            xoff = (long)part * p.hs_part_stride + (long)(xr >> 6) * p.hs_T * 64 + (xr & 63);
        }
        const bool scaled = has_xs && x < p.xscale_cols;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 16 + rr;
            const int sw = (row >> 1) & 7;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(region + row * 128 + (((2 * c4) ^ sw) << 4));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(region + row * 128 + (((2 * c4 + 1) ^ sw) << 4));
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += bx[r];
            if (scaled) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] *= p.xscale;
            }
            if (GELU) {
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const wx_f2 g = gelu_f2((wx_f2){v[r], v[r + 1]});
                    v[r] = g[0];
                    v[r + 1] = g[1];
                }
            }
            if (HAS_R) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += (float)r8[it][r];
            }
            // the fp32 value is pinned in a register before its conversion: straight-line code lets the compiler fold the GELU's
            // (or the scale's) last fma and the conversion into ONE v_fma_mixlo_f16 -- one rounding where the general form,
            // which converts behind a branch, rounds twice (17-21 of 262 144 outputs one fp16 ulp apart before the pin)
            half8 o;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float t = v[r];
                asm volatile("" : "+v"(t));
                o[r] = (h16)t;
            }
            *reinterpret_cast<half8*>(out + orow[it] + xoff) = o;
        }
    }
}

template <bool GELU>
__device__ __forceinline__ void gemm_epilogue_lds32(const GemmArgs& p, f32x4 (&acc)[8][4], int xw, int yw, int lane,
                                                    char* region) {
    if (xw + 128 <= p.RX && yw + 64 <= p.RY && !(p.bias && p.bias_on_y) && !p.nt_stores && !p.general_epilogue) {      // (wave-uniform)
        if (p.hs_T > 0) {
            if (!p.R) return gemm_epilogue_fast32<GELU, false, true>(p, acc, xw, yw, lane, region);
        } else if (p.R) {
            return gemm_epilogue_fast32<GELU, true, false>(p, acc, xw, yw, lane, region);
        } else {
            return gemm_epilogue_fast32<GELU, false, false>(p, acc, xw, yw, lane, region);
        }
    }
    h16* __restrict__ out = p.out;
    const h16* __restrict__ R = p.R;
    const h16* __restrict__ bias = p.bias;
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = lane >> 2, c4 = lane & 3;
    // 128-byte rows of eight 16-byte chunks, chunk ^= (row >> 1) & 7: the 16 lanes of a b128 access (16 rows of one chunk
    // on the write side, 4 rows x 4 alternate chunks on the read side) touch 16 different 16-byte bank groups
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = j * 16 + fr;
                *reinterpret_cast<f32x4*>(region + row * 128 + (((i2 * 4 + fq) ^ ((row >> 1) & 7)) << 4)) = acc[q * 2 + i2][j];
            }
        const int x = xw + q * 32 + c4 * 8;
        float bx[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bx[r] = 0.f;
        if (bias && !p.bias_on_y && x + 7 < p.RX) {
            const half8 b8 = *reinterpret_cast<const half8*>(bias + x);
#pragma unroll
            for (int r = 0; r < 8; ++r) bx[r] = (float)b8[r];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 16 + rr;
            const int y = yw + row;
            const int sw = (row >> 1) & 7;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(region + row * 128 + (((2 * c4) ^ sw) << 4));
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(region + row * 128 + (((2 * c4 + 1) ^ sw) << 4));
            if (y >= p.RY || x >= p.RX) continue;
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const float by_ = (bias && p.bias_on_y) ? (float)bias[y] : 0.f;
            if (x + 7 < p.RX) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += by_ + bx[r];
                if (x < p.xscale_cols) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] *= p.xscale;
                }
                if (GELU) {
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const wx_f2 g = gelu_f2((wx_f2){v[r], v[r + 1]});
                        v[r] = g[0];
                        v[r + 1] = g[1];
                    }
                }
                if (R) {
                    const half8 r8 = *reinterpret_cast<const half8*>(R + (long)y * p.ldr + x);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)r8[r];
                }
                half8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = (h16)v[r];
                long oaddr = (long)y * p.ldo + x;
                if (p.hs_T > 0) {
                    const int bb = y / p.hs_T, tt = y - bb * p.hs_T;
                    const int part = x / p.hs_d, xr = x - part * p.hs_d;
                    oaddr = (long)part * p.hs_part_stride + (((long)bb * p.hs_H + (xr >> 6)) * p.hs_T + tt) * 64 + (xr & 63);
                }
                if (p.nt_stores)
                    __builtin_nontemporal_store(o, reinterpret_cast<half8*>(out + oaddr));
                else
                    *reinterpret_cast<half8*>(out + oaddr) = o;
            } else {
                for (int r = 0; r < 8 && x + r < p.RX; ++r) {
                    float t = v[r] + by_;
                    if (bias && !p.bias_on_y) t += (float)bias[x + r];
                    if (x + r < p.xscale_cols) t *= p.xscale;
                    if (GELU) t = gelu_f(t);
                    if (R) t += (float)R[(long)y * p.ldr + x + r];
                    out[(long)y * p.ldo + x + r] = (h16)t;
                }
            }
        }
    }
}

template <bool GELU>
__global__ __launch_bounds__(512) void gemm_pipe_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int ntx = (p.RX + B8 - 1) / B8;
    const int nty = (p.RY + B8 - 1) / B8;
    const int ntiles = ntx * nty;
    const char* __restrict__ Xb = reinterpret_cast<const char*>(p.X);
    const char* __restrict__ Yb = reinterpret_cast<const char*>(p.Y);
    const int r_in = lane >> 3, csrc = (lane & 7) ^ r_in;
    auto tile_xy = [&](int tile_lin, int& x0, int& y0) {
        const int tile = xcd_remap(tile_lin, ntiles);
        constexpr int GY = 4;
        const int per_group = GY * ntx;
        const int grp = tile / per_group, rem = tile - grp * per_group;
        const int gcnt = min(GY, nty - grp * GY);
        const int tx = rem / gcnt, ty = grp * GY + rem - tx * gcnt;
        x0 = tx * B8;
        y0 = ty * B8;
    };
    // per-lane byte offsets of the 16-B piece each DMA instruction fetches (source-side swizzle), for one tile
    auto offsets = [&](int x0, int y0, unsigned (&v)[4][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = h * 128 + j * 64 + wave * 8 + r_in;
                v[h][j] = (unsigned)(((long)min(x0 + row, p.RX - 1) * p.ldx + csrc * 8) * 2);
                v[2 + h][j] = (unsigned)(((long)min(y0 + row, p.RY - 1) * p.ldy + csrc * 8) * 2);
            }
    };
    int tile_lin = blockIdx.x, x0, y0;
    tile_xy(tile_lin, x0, y0);
    // ONE set of staging offsets: a tile's stagings are, in program order, all of its own k-tiles and then -- from P2 of
    // its k-tile nk - 2 on: Y(nk), X(nk), Y(nk + 1) -- only the next tile's, so the offsets switch to the next tile at that
    // point (`vst`, with `tsub` = nk taken off the k-tile index) and nothing is selected per staging
    unsigned vst[4][2];
    offsets(x0, y0, vst);
    int tsub = 0, xn = x0, yn = y0;
    bool has_next = false;

    const int nk = p.K / BK;
    // before the switch: this tile's k-tile T.  After it: the NEXT tile's k-tile T - nk -- its prologue, riding in this
    // tile's last two k-tiles; the block's last tile never switches and stages its own last k-tile again (never read;
    // keeps every vmcnt count static)
    auto stage = [&](auto WHICH, int T, int buf) {
        constexpr int which = decltype(WHICH)::value;
        const int kt = min(T - tsub, nk - 1);
        char* dst = smem + buf * BUF8 + which * HALF8 + wave * 1024;
        const char* src = (which < 2 ? Xb : Yb);
        const unsigned koff = (unsigned)kt * (BK * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (vst[which][j] + koff)),
                                             (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    const int fr = lane & 15, fq = lane >> 4;
    const int sw0 = ((fq ^ (fr & 7)) << 4), sw1 = (((4 + fq) ^ (fr & 7)) << 4);
    const int aoff = wr * HALF8 + fr * 128;
    const int boff = 2 * HALF8 + (wc >> 1) * HALF8 + ((wc & 1) * 64 + fr) * 128;
    // the epilogue's 8 KiB of this wave: waves 0-3 in the X half of buffer 1, waves 4-7 behind the staging buffers
    char* const epi = smem + (wave < 4 ? BUF8 + wave * EPIP_WAVE : 2 * BUF8 + (wave - 4) * EPIP_WAVE);

    if (p.stagger_ticks > 0) {
        // spread the blocks' tile boundaries over `stagger_ticks`: identical tiles keep the CUs in lock step otherwise, and
        // every round of epilogues is one burst of 256 x 128 KiB of stores
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)p.stagger_ticks * blockIdx.x / gridDim.x;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    stage(I0{}, 0, 0);
    stage(I1{}, 0, 0);
    stage(I2{}, 0, 0);
    stage(I3{}, 0, 0);
    stage(I2{}, 1, 1);
    stage(I3{}, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // run the second wave group one barrier behind

    f32x4 acc[8][4];
    half8 a[4], b[4][2];
    auto ktile = [&](auto BUFC, int t, bool to_next) {
        constexpr int BUF = decltype(BUFC)::value;
        const char* base = smem + BUF * BUF8;
        auto read_a = [&](int half, int sw) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(base + aoff + (half * 4 + i) * 2048 + sw);
        };
        auto mfma16 = [&](int half, int ks) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j][ks], acc[half * 4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        // the four phases of gemm_8phase_kernel (hazards argued there)
        read_a(0, sw0);
        {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j][0] = *reinterpret_cast<const half8*>(base + boff + j * 2048 + sw0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j][1] = *reinterpret_cast<const half8*>(base + boff + j * 2048 + sw1);
        }
        stage(I0{}, t + 1, BUF ^ 1);
        __builtin_amdgcn_s_barrier();
        mfma16(0, 0);
        __builtin_amdgcn_s_barrier();
        read_a(0, sw1);
        stage(I1{}, t + 1, BUF ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma16(0, 1);
        __builtin_amdgcn_s_barrier();
        if (BUF == 0 && to_next) {
            // k-tile nk - 2, between P1 and P2: every staging from here on belongs to the next tile of this block
            has_next = tile_lin + (int)gridDim.x < ntiles;
            if (has_next) {
                tile_xy(tile_lin + (int)gridDim.x, xn, yn);
                offsets(xn, yn, vst);
                tsub = nk;
            }
        }
        read_a(1, sw0);
        stage(I2{}, t + 2, BUF);
        __builtin_amdgcn_s_barrier();
        mfma16(1, 0);
        __builtin_amdgcn_s_barrier();
        read_a(1, sw1);
        stage(I3{}, t + 2, BUF);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma16(1, 1);
        __builtin_amdgcn_s_barrier();
    };
#ifdef WX_LAB_ENV
    int lab_tile = 0;
    const int lab_sel = (blockIdx.x == 0 ? 0 : blockIdx.x == 128 ? 1 : -1), lab_w = (wave == 0 ? 0 : wave == 4 ? 1 : wave == 7 ? 2 : -1);
    auto stamp = [&](int k) {
        if (p.lab_stamps && lab_sel >= 0 && lab_w >= 0 && lab_tile < 16 && lane == 0)
            p.lab_stamps[((lab_sel * 3 + lab_w) * 16 + lab_tile) * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };
#define WX_STAMP(k) stamp(k)
#else
#define WX_STAMP(k)
#endif
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        WX_STAMP(0);
#ifdef WX_LAB_ENV
        const unsigned long long lab_c0 = __builtin_amdgcn_s_memtime();
#endif
        for (int t = 0; t < nk; t += 2) {
            ktile(I0{}, t, t == nk - 2);
            ktile(I1{}, t + 1, false);
        }
        WX_STAMP(1);
#ifdef WX_LAB_ENV
        if (p.lab_stamps && lab_sel >= 0 && lab_w >= 0 && lab_tile < 16 && lane == 0)
            p.lab_stamps[((lab_sel * 3 + lab_w) * 16 + lab_tile) * 8 + 6] = __builtin_amdgcn_s_memtime() - lab_c0;     // core-clock cycles of the k-loop
#endif
        if (wr == 0) __builtin_amdgcn_s_barrier();   // the two wave groups meet again
        // every fragment read of this tile is done: the X half of buffer 1 is free for the epilogue.  The next tile's k-tile
        // 0 (buffer 0) and its Y(1) (the Y half of buffer 1) may still be in flight -- they land in LDS the epilogue does not
        // touch, and waiting for them HERE put 96 KiB per CU of exposed DMA in front of every epilogue
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        WX_STAMP(2);
        gemm_epilogue_lds32<GELU>(p, acc, x0 + wr * 128, y0 + wc * 64, lane, epi);
        WX_STAMP(3);
        if (!has_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the block's last, never-read stagings have landed before its LDS is released)
            break;
        }
        tile_lin += (int)gridDim.x;
        x0 = xn;
        y0 = yn;
        tsub = 0;             // `vst` already holds this tile's offsets
        // the epilogue's LDS reads are retired before any wave stages X(1) of the new tile over them (P0 / P1 of its k-tile 0),
        // and everything staged for the new tile so far -- its k-tile 0 and Y(1) -- has landed (the epilogue's stores too)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        WX_STAMP(4);
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();   // and the second group falls one barrier behind again
        WX_STAMP(5);
#ifdef WX_LAB_ENV
        ++lab_tile;
#endif
    }
}
#undef WX_STAMP


#ifdef WX_LAB_ENV
// ---------------------------------------------------------------------------------------
// LAB (round 5): 256 (x) x 128 (y) x 32 tile, FOUR waves (2 x 2, 128 x 64 of the output each: the same per-wave shape and
// the same order of k-steps as the kernels above, so the same bits), 72 KiB of LDS as a three-stage ring of 24 KiB k-tiles
// -- TWO blocks per CU.  The point: a tile's prologue and epilogue (7-14 us of a 36-43 us K = 1280 tile, which the
// one-block-per-CU kernels cannot hide: no registers to park a finished tile, no LDS beside the operands) overlap the OTHER
// block's main loop, the two blocks drifting out of phase by themselves.
// LDS rows are 64 B (32 halves); slot (row, c') of a 1 KiB DMA image (16 rows) holds global chunk c' ^ ((row >> 2) & 3), so the
// 16 lanes of a fragment read (16 rows, one chunk) land on 16 different 16-byte bank groups.
constexpr int T4X = 256, T4Y = 128, T4K = 32;
constexpr int T4_STAGE = (T4X + T4Y) * T4K * 2;          // 24 KiB
constexpr int T4_LDS = 3 * T4_STAGE;
// slot of chunk c in row r: c ^ T4_SWZ((r >> 2) & 3), the permutation 0 2 3 1.  With the identity (first two versions of the lab)
// every ds_read_b128 of a fragment was a 2-way bank conflict: the hardware serves a b128 read in the lane groups {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), not in runs of 16 lanes, and in such a group rows r and r + 4 of
// neighbouring k-chunks met on one 16-byte bank group.  This permutation gives the four rows of a residue mod 4 four different slots
// in every lane group.
#define T4_SWZ(h) ((0x78 >> (2 * (h))) & 3)                      // 72 KiB (>= 4 x EPI_WAVE for the epilogue)

template <bool GELU>
__global__ __launch_bounds__(256, 2) void gemm_4w_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;            // wr: x half (128 rows), wc: y half (64 rows)

    const int ntx = (p.RX + T4X - 1) / T4X;
    const int nty = (p.RY + T4Y - 1) / T4Y;
    const int tile = xcd_remap(blockIdx.x, ntx * nty);
    constexpr int GY = 8;
    const int per_group = GY * ntx;
    const int grp = tile / per_group, rem = tile - grp * per_group;
    const int gcnt = min(GY, nty - grp * GY);
    const int tx = rem / gcnt, ty = grp * GY + rem - tx * gcnt;
    const int x0 = tx * T4X, y0 = ty * T4Y;
    const char* __restrict__ Xb = reinterpret_cast<const char*>(p.X);
    const char* __restrict__ Yb = reinterpret_cast<const char*>(p.Y);

    // DMA: a wave instruction moves 16 rows x 64 B (lane l: row l >> 2, slot l & 3).  X: 256 rows = 16 instructions, Y: 128
    // rows = 8; wave w issues X row blocks w, w + 4, w + 8, w + 12 and Y row blocks w, w + 4.
    const int r_in = lane >> 2, cslot = lane & 3;
    unsigned vx[4], vy[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave + 4 * j) * 16 + r_in;
        vx[j] = (unsigned)(((long)min(x0 + row, p.RX - 1) * p.ldx + ((cslot ^ T4_SWZ((row >> 2) & 3)) * 8)) * 2);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave + 4 * j) * 16 + r_in;
        vy[j] = (unsigned)(((long)min(y0 + row, p.RY - 1) * p.ldy + ((cslot ^ T4_SWZ((row >> 2) & 3)) * 8)) * 2);
    }
    const int nk = p.K / T4K;
    auto stage = [&](int t) {
        const int kt = min(t, nk - 1);            // past the end: the last k-tile again (never read; static vmcnt counts)
        char* base = smem + (t % 3) * T4_STAGE;
        const unsigned koff = (unsigned)kt * (T4K * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xb + (vx[j] + koff)),
                                             (__attribute__((address_space(3))) void*)(base + (wave + 4 * j) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Yb + (vy[j] + koff)),
                                             (__attribute__((address_space(3))) void*)(base + T4X * 64 + (wave + 4 * j) * 1024), 16, 0, 0);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    // fragment (16 rows starting at r0, k-chunk fq): row r0 + fr, slot fq ^ ((row >> 2) & 3); r0 is a multiple of 16
    const int foff = fr * 64 + ((fq ^ T4_SWZ((fr >> 2) & 3)) << 4);
    const int aoff = wr * 128 * 64 + foff;
    const int boff = T4X * 64 + wc * 64 * 64 + foff;

    // Fragments double-buffered in registers (second version of the lab): the 12 fragment reads of k-tile t + 1 are issued in
    // front of the 32 MFMAs of k-tile t and land under them, so a wave's matrix pipe does not wait for its own LDS reads --
    // the first version read and multiplied in turn and left that to the other block's wave.  Ring: while k-tile t is
    // multiplied from registers, t + 1 is read, t + 2 is in flight and t + 3 is issued into the buffer t was read from.
    half8 a0[8], b0[4], a1[8], b1[4];
    auto frag = [&](int t, half8 (&a)[8], half8 (&b)[4]) {
        const char* base = smem + (t % 3) * T4_STAGE;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const half8*>(base + aoff + i * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8*>(base + boff + j * 1024);
    };
    // the first MFMAs of a k-tile go in FRONT of the next tile's fragment reads: the compiler cannot see through the asm wait
    // at the top and puts its own lgkmcnt(0) in front of the first MFMA -- placed there it waits for nothing, placed behind the
    // twelve reads it would wait for all of them
    auto mm = [&](half8 (&a)[8], half8 (&b)[4], int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    stage(0);
    stage(1);
    stage(2);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // k-tile 0 is in the LDS for every wave
    frag(0, a0, b0);
    unsigned long long lab_wait = 0, lab_bar = 0, lab_t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < nk; t += 2) {             // nk is even (checked at launch)
        // k-tile t + 1 has landed (this wave's share; the barrier makes it everybody's) and the fragments of t are in registers:
        // every wave is then past its reads of t, whose buffer takes k-tile t + 3
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        const unsigned long long c2 = __builtin_amdgcn_s_memtime();
        lab_wait += c1 - c0;
        lab_bar += c2 - c1;
        stage(t + 3);
        __builtin_amdgcn_s_setprio(1);
        mm(a0, b0, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        frag(t + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0, 2, 8);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage(t + 4);
        __builtin_amdgcn_s_setprio(1);
        mm(a1, b1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        frag(t + 2, a0, b0);                      // past the end: the re-staged last k-tile, never multiplied
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b1, 2, 8);
        __builtin_amdgcn_s_setprio(0);
    }
    if (p.lab_stamps && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 700) && (wave == 0 || wave == 3)) {
        unsigned long long* o = p.lab_stamps + ((blockIdx.x == 0 ? 0 : 1) * 2 + (wave == 0 ? 0 : 1)) * 4;
        o[0] = __builtin_amdgcn_s_memtime() - lab_t0;      // the k-loop
        o[1] = lab_wait;                                    // of which: waiting for the DMA / fragments at the top of even k-tiles
        o[2] = lab_bar;                                     // of which: at the barrier behind it
        o[3] = nk;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's DMA has landed and every fragment read is done: the LDS is the epilogue's
    gemm_epilogue_lds<GELU>(p, acc, x0 + wr * 128, y0 + wc * 64, lane, smem + wave * EPI_WAVE, 0);
}

#endif   // WX_LAB_ENV

}  // namespace

#ifdef WX_LAB_ENV
// lab: the timeline stamps of the LAST tile-pipelined launch (2 blocks x 3 waves x 16 tiles x 8 stamps, 10 ns ticks)
static constexpr size_t LAB_STAMP_WORDS = 2 * 3 * 16 * 8;
static unsigned long long* g_lab_stamps = nullptr;
extern "C" int wx_lab_read_gemm_stamps(unsigned long long* out) {
    if (!g_lab_stamps) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpy(out, g_lab_stamps, LAB_STAMP_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif

hipError_t launch_gemm_f16(const GemmArgs& a, int batch, bool gelu, hipStream_t s) {
    const int ntx = (a.RX + BX - 1) / BX, nty = (a.RY + BY - 1) / BY;
    dim3 grid(ntx * nty, 1, batch), block(256);
    const size_t lds = 4 * TILE_BYTES;
#ifdef LAB_GEMM128            // lab builds only (tools/build_lab.py): every GEMM on the 128 x 128 tile, which leaves room on its CU
    constexpr bool lab128 = true;
#else
    constexpr bool lab128 = false;
#endif
    if (a.zsplit > 0 && (a.RX >= 512 || batch % a.zsplit)) return hipErrorInvalidValue;     // two-level batch: the 128 x 128 kernel only
    if (a.xscale_cols & 7) return hipErrorInvalidValue;
    if (!lab128 && a.zsplit <= 0 && a.RX >= 512 && a.RY >= 512 && a.K % BK == 0 && a.K >= 2 * BK) {
        const int n8x = (a.RX + B8 - 1) / B8, n8y = (a.RY + B8 - 1) / B8;
        dim3 grid8(n8x * n8y, 1, batch), block8(512);
        // several host threads (one per engine context) launch GEMMs concurrently: raise the LDS limit exactly once
        static std::once_flag attr_once;
        std::call_once(attr_once, [] {
            (void)hipFuncSetAttribute((const void*)gemm_8phase_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
            (void)hipFuncSetAttribute((const void*)gemm_8phase_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
            (void)hipFuncSetAttribute((const void*)gemm_8phase_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
            (void)hipFuncSetAttribute((const void*)gemm_8phase_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
        });
#ifdef WX_LAB_ENV
        static const int lab_4w = WX_LAB_GETENV_INT("WX_GEMM_4W", 0);      // lab: the four-wave, two-blocks-per-CU kernel
        if (lab_4w && a.max_blocks == 0 && a.y_gather_group <= 0 && batch == 1 && a.K % (2 * T4K) == 0 && a.K >= 4 * T4K) {
            static std::once_flag attr_once_4w;
            std::call_once(attr_once_4w, [] {
                (void)hipFuncSetAttribute((const void*)gemm_4w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS);
                (void)hipFuncSetAttribute((const void*)gemm_4w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS);
            });
            const dim3 grid4(((a.RX + T4X - 1) / T4X) * ((a.RY + T4Y - 1) / T4Y), 1, 1);
            GemmArgs a4 = a;
            if (WX_LAB_GETENV_INT("WX_GEMM_STAMPS", 0)) {
                if (!g_lab_stamps) (void)hipMalloc(reinterpret_cast<void**>(&g_lab_stamps), LAB_STAMP_WORDS * sizeof(unsigned long long));
                if (g_lab_stamps) (void)hipMemsetAsync(g_lab_stamps, 0, LAB_STAMP_WORDS * sizeof(unsigned long long), s);
                a4.lab_stamps = g_lab_stamps;
            }
            if (gelu)
                hipLaunchKernelGGL((gemm_4w_kernel<true>), grid4, dim3(256), T4_LDS, s, a4);
            else
                hipLaunchKernelGGL((gemm_4w_kernel<false>), grid4, dim3(256), T4_LDS, s, a4);
            return hipGetLastError();
        }
#endif
        // the tile-pipelined kernel (one staging pipeline over all the tiles of a block): plain operands, an even number of
        // k-tiles, one batch.  max_blocks > 0 caps its grid (a launch confined to that many CUs), < 0 asks for the
        // one-tile-per-block kernel (tests: the two are bit-identical)
        if (a.max_blocks >= 0 && a.y_gather_group <= 0 && batch == 1 && (a.K / BK) % 2 == 0) {
            static std::once_flag attr_once_pipe;
            std::call_once(attr_once_pipe, [] {
                (void)hipFuncSetAttribute((const void*)gemm_pipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
                (void)hipFuncSetAttribute((const void*)gemm_pipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
            });
            static int n_cu = 0;
            if (!n_cu) {
                int dev = 0, cu = 0;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
                n_cu = cu > 0 ? (cu & ~7) : 256;
            }
            const int cap = a.max_blocks > 0 ? a.max_blocks : n_cu;
            dim3 gridp(n8x * n8y < cap ? n8x * n8y : cap, 1, 1);
            GemmArgs ap = a;
            static const int lab_stagger = WX_LAB_GETENV_INT("WX_GEMM_STAGGER_US", -1) >= 0 ? WX_LAB_GETENV_INT("WX_GEMM_STAGGER_US", -1) * 100 : -1;   // lab builds
            static const int lab_nt = WX_LAB_GETENV_INT("WX_GEMM_NT", -1);
            if (lab_stagger >= 0) ap.stagger_ticks = (n8x * n8y > (int)gridp.x) ? lab_stagger : 0;
            if (lab_nt >= 0) ap.nt_stores = lab_nt;
            static const int lab_epi = WX_LAB_GETENV_INT("WX_GEMM_EPI", 0);
            if (lab_epi) ap.general_epilogue = 1;                 // lab builds: A/B of the two epilogues (WX_GEMM_EPI=1)
#ifdef WX_LAB_ENV
            static const int lab_stamps_on = WX_LAB_GETENV_INT("WX_GEMM_STAMPS", 0);     // tools/lab_gemm_timeline.py
            if (lab_stamps_on) {
                if (!g_lab_stamps) {
                    (void)hipMalloc(reinterpret_cast<void**>(&g_lab_stamps), LAB_STAMP_WORDS * sizeof(unsigned long long));
                }
                if (g_lab_stamps) (void)hipMemsetAsync(g_lab_stamps, 0, LAB_STAMP_WORDS * sizeof(unsigned long long), s);
                ap.lab_stamps = g_lab_stamps;
            }
#endif
            if (gelu)
                hipLaunchKernelGGL((gemm_pipe_kernel<true>), gridp, block8, LDSP, s, ap);
            else
                hipLaunchKernelGGL((gemm_pipe_kernel<false>), gridp, block8, LDSP, s, ap);
            return hipGetLastError();
        }
        if (a.max_blocks > 0 && a.max_blocks < n8x * n8y && a.y_gather_group <= 0 && batch == 1) {
            static std::once_flag attr_once_p;
            std::call_once(attr_once_p, [] {
                (void)hipFuncSetAttribute((const void*)gemm_8phase_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
                (void)hipFuncSetAttribute((const void*)gemm_8phase_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
            });
            dim3 gridp(a.max_blocks, 1, batch);
            if (gelu)
                hipLaunchKernelGGL((gemm_8phase_kernel<true, false, true>), gridp, block8, LDS8, s, a);
            else
                hipLaunchKernelGGL((gemm_8phase_kernel<false, false, true>), gridp, block8, LDS8, s, a);
            return hipGetLastError();
        }
        if (a.y_gather_group > 0) {
            if (gelu)
                hipLaunchKernelGGL((gemm_8phase_kernel<true, true>), grid8, block8, LDS8, s, a);
            else
                hipLaunchKernelGGL((gemm_8phase_kernel<false, true>), grid8, block8, LDS8, s, a);
        } else if (gelu)
            hipLaunchKernelGGL((gemm_8phase_kernel<true, false>), grid8, block8, LDS8, s, a);
        else
            hipLaunchKernelGGL((gemm_8phase_kernel<false, false>), grid8, block8, LDS8, s, a);
        return hipGetLastError();
    }
    if (a.y_gather_group > 0) {
        if (gelu && a.RX <= 64)
            hipLaunchKernelGGL((gemm_glds_kernel<true, true, true>), grid, block, lds, s, a);       // 64 x 128 tile: no waves on absent rows
        else if (gelu)
            hipLaunchKernelGGL((gemm_glds_kernel<true, true>), grid, block, lds, s, a);
        else
            hipLaunchKernelGGL((gemm_glds_kernel<false, true>), grid, block, lds, s, a);
    } else if (gelu)
        hipLaunchKernelGGL((gemm_glds_kernel<true, false>), grid, block, lds, s, a);
    else
        hipLaunchKernelGGL((gemm_glds_kernel<false, false>), grid, block, lds, s, a);
    return hipGetLastError();
}
