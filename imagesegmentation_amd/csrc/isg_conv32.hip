// The encoder's C -> C channel convolutions, C = 32 or 16 (3 x 3 and 5 x 5 taps, SAME padding; reference
// polyrnn/RNNConvExperiment's second and third tf.layers.conv2d and the 16-channel layers of
// polyrnn/History2LargeExperiment.py:15-22) as MFMA kernels, bf16 with float accumulation.  With 16 channels
// the 32-row MFMA tile is half zero rows: these layers move 64 bytes per pixel, the matrix pipe has room.
//   forward / data gradient:  D[row][pixel] = sum_k A[row][k] * X[k][pixel]; the caller's weight image
//       decides which of the two it is (data gradient = flipped taps, transposed channels);
//       optional bias and ReLU in the epilogue - or, for a data gradient that lands on the ReLU output of the
//       layer below, that layer's ReLU mask and its bias gradient (the masked values summed per channel),
//       which saves the separate mask-and-sum pass over the tensor.
//   weight gradient:          D[cin][cout] = sum_pixels X[pixel + tap][cin] * dZ[pixel][cout].
// These layers move 150 - 200 bytes per pixel against 18 - 51 kFLOP: HBM-bound, so the kernels are
// plain persistent tiles with the weight image resident in LDS and the next tile in flight in
// registers while the current one is multiplied.
#include <hip/hip_bf16.h>

#include "isg_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int MROWS = 32;       // rows of one MFMA tile: the output channels, zero rows above the real ones
constexpr int TW = 32;
constexpr int kThreads = 512;   // 8 waves

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

int g_cus = 0;
int num_cus() {
    if (g_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_cus = prop.multiProcessorCount;
        if (g_cus <= 0) g_cus = 256;
    }
    return g_cus;
}

// ---------------------------------------------------------------------------------------------
// forward / data gradient
// ---------------------------------------------------------------------------------------------
#ifndef ISG_C32_NPT5
#define ISG_C32_NPT5 4
#endif
// tile rows per wave: the more rows, the fewer LDS reads per MFMA (see the k-loop); the 5 x 5, 32-channel kernel holds
// one workgroup per CU whatever its tile (50 KB of weights), so it takes the largest tile that fits beside them
template <int KS, int CH>
constexpr int rows_per_wave() { return KS == 5 && CH == 32 ? ISG_C32_NPT5 : 2; }

template <int KS, int CH>
struct CF {
    static_assert(CH == 16 || CH == 32, "channels in = channels out = 16 or 32");
    static constexpr int NPT = rows_per_wave<KS, CH>();
    static constexpr int TH = (kThreads / 64) * NPT;  // tile rows
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int SPT = CH / 16;               // k-steps per tap: 16-channel groups
    static constexpr int NCHK = CH / 8;               // 16-byte chunks per pixel
    static constexpr int NS = KS * KS * SPT;
    static constexpr int HH = TH + KS - 1, HWP = TW + KS - 1;
    static constexpr int NPIX = HH * HWP;
    static constexpr int PSTRIDE = CH * 2 + 16;       // pad spreads the 16-byte operand reads over the banks
    static constexpr int IN_BYTES = NPIX * PSTRIDE;
    static constexpr int W_BYTES = NS * 1024;
    static constexpr int NLOAD = (NPIX * NCHK + kThreads - 1) / kThreads;
    static constexpr int LDS_BYTES = W_BYTES + IN_BYTES + MROWS * 4;
};

struct Conv32Params {
    const void* x;     // [B,H,W,CH] bf16
    const void* wimg;  // [NS][64][8] bf16
    const float* bias; // [CH] or null
    void* y;           // [B,H,W,CH] bf16
    int B, H, W, tiles_x, tiles_y, relu;
    int y_pstride;     // bytes from one output pixel to the next (CH * 2, or more: y is a channel slice of wider pixels)
    const void* mask;      // MASK: [B,H,W,CH] bf16, the output keeps its value where mask > 0 and is zero elsewhere
    float* bias_partial;   // MASK: [grid][32] sums of the kept values per channel
};

constexpr int kMaskedMaxBlocks = 1024;

// MASK: no bias, no ReLU; the epilogue applies `mask` and sums what it keeps (see the top of the file).
// waves per SIMD the register allocation aims at: two workgroups of 8 waves per CU wherever the LDS image leaves room
// for two (128 registers per lane), else one
template <int KS, int CH>
constexpr int conv32_min_blocks() { return (160 * 1024) / CF<KS, CH>::LDS_BYTES >= 2 ? 4 : 2; }

template <int KS, int CH, bool MASK>
__global__ __launch_bounds__(kThreads, (conv32_min_blocks<KS, CH>())) void conv32_kernel(Conv32Params p) {
    using K = CF<KS, CH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_w = smem;
    char* lds_in = smem + K::W_BYTES;
    float* lds_bias = reinterpret_cast<float*>(smem + K::W_BYTES + K::IN_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px_l = lane & 31, half = lane >> 5;
    {
        const uint4* src = static_cast<const uint4*>(p.wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds_w);
        for (int i = tid; i < K::W_BYTES / 16; i += kThreads) dst[i] = src[i];
        if (tid < MROWS) lds_bias[tid] = p.bias && tid < CH ? p.bias[tid] : 0.0f;
    }
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.B;
    const int n_iter = (ntiles + gridDim.x - 1) / gridDim.x;

    auto tile_origin = [&](int k, int& b, int& ty0, int& tx0) -> bool {
        const int tile = k * gridDim.x + blockIdx.x;
        if (k >= n_iter || tile >= ntiles) return false;
        b = tile / tiles_per_img;
        const int rem = tile - b * tiles_per_img;
        const int tyi = rem / p.tiles_x;
        ty0 = tyi * K::TH, tx0 = (rem - tyi * p.tiles_x) * TW;
        return true;
    };
    uint4 st[K::NLOAD];
    auto stage_load = [&](int k) {
        int b, ty0, tx0;
        if (!tile_origin(k, b, ty0, tx0)) return;
#ifdef ISG_C32_NO_STAGE
        if (k > 0) return;
#endif
#pragma unroll
        for (int j = 0; j < K::NLOAD; ++j) {
            const int idx = tid + kThreads * j;
            const int px = idx / K::NCHK, ch = idx % K::NCHK;
            const int hy = px / K::HWP, hx = px - hy * K::HWP;
            const int gy = ty0 + hy - K::PAD, gx = tx0 + hx - K::PAD;
            const bool inside = px < K::NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            st[j] = make_uint4(0u, 0u, 0u, 0u);
            if (inside) st[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.x) +
                                                                (((size_t)b * p.H + gy) * p.W + gx) * (CH * 2) + ch * 16);
        }
    };
    auto stage_store = [&](int k) {
        const int tile = k * gridDim.x + blockIdx.x;
        if (k >= n_iter || tile >= ntiles) return;
#pragma unroll
        for (int j = 0; j < K::NLOAD; ++j) {
            const int idx = tid + kThreads * j;
            const int px = idx / K::NCHK, ch = idx % K::NCHK;
            if (px < K::NPIX) *reinterpret_cast<uint4*>(lds_in + px * K::PSTRIDE + ch * 16) = st[j];
        }
    };

    float bsum[4][4];  // MASK: this lane's kept values summed over its pixels, rows 8 g + 4 half + i
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) bsum[g][i] = 0.0f;

    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int k = 0; k < n_iter; ++k) {
        int b, ty0, tx0;
        const bool valid = tile_origin(k, b, ty0, tx0);
        stage_load(k + 1);
        if (valid) {
            constexpr int NPT = K::NPT;
            uint2 mk[NPT][2 * K::SPT];  // MASK: the mask values of this lane's rows
            // Requested ahead of the k-loop where one workgroup runs per CU (nothing else covers their latency); behind it
            // where two do and the 16 registers they would hold across the loop are what keeps the kernel above 128.
#ifndef ISG_C32_MASK_EARLY
            constexpr bool MASK_LATE = conv32_min_blocks<KS, CH>() == 4 && CH == 32;
#else
            constexpr bool MASK_LATE = false;
#endif
            auto load_mask = [&]() {
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
                    const int cy = min(ty0 + NPT * wave + pt, p.H - 1), cx = min(tx0 + px_l, p.W - 1);
                    const char* src = static_cast<const char*>(p.mask) + (((size_t)b * p.H + cy) * p.W + cx) * (CH * 2);
#pragma unroll
                    for (int g = 0; g < 2 * K::SPT; ++g) mk[pt][g] = *reinterpret_cast<const uint2*>(src + (8 * g + 4 * half) * 2);
                }
            };
            if constexpr (MASK && !MASK_LATE) load_mask();
            f32x16 acc[NPT];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pt][r] = 0.0f;
            const char* bb = lds_in + ((NPT * wave) * K::HWP + px_l) * K::PSTRIDE + half * 16;
            const char* aw = lds_w + lane * 16;
            // The k-loop goes column by column: a column = one tap column tx and one 16-channel group q.  Its KS taps
            // (ty) on the wave's NPT pixel rows read NPT + KS - 1 rows of the tile, each fetched ONCE and used by
            // every (ty, pt) pair with ty + pt = row: NPT + 2 KS - 1 LDS reads of 1 KB per NPT KS MFMAs in place of
            // the KS (NPT + 1) of a loop that fetches per tap (5 x 5: 1.1 KB per MFMA at 2 rows, 0.65 at 4, from 1.5;
            // four SIMDs' MFMAs take 1 KB per 8 LDS cycles... the LDS delivers 128 B a cycle, so 1 KB per MFMA is
            // where the matrix cores and the LDS are level).
            // A sub-step = one tap of a column (NPT MFMAs); its fragments are requested AHEAD sub-steps before.
#ifndef ISG_C32_AHEAD
#define ISG_C32_AHEAD 2
#endif
            constexpr int AHEAD = NPT >= 4 ? 1 : ISG_C32_AHEAD, NB = AHEAD + 1;
            constexpr int NR = NPT + KS - 1, NU = KS * K::SPT * KS;
            static_assert(AHEAD < KS, "two column buffers");
            bf16x8 af[NB], rows[2][NR];
            auto fetch = [&](int u) {
                const int col = u / KS, ty = u - col * KS;
                const int tx = col / K::SPT, q = col - tx * K::SPT;
                af[u % NB] = *reinterpret_cast<const bf16x8*>(aw + ((ty * KS + tx) * K::SPT + q) * 1024);
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (r == ty + NPT - 1 || (ty == 0 && r < NPT - 1))
                        rows[col & 1][r] = *reinterpret_cast<const bf16x8*>(bb + (r * K::HWP + tx) * K::PSTRIDE + q * 32);
            };
#ifndef ISG_C32_NO_KLOOP  // (diagnostic builds leave a part out to see what the others take)
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) fetch(u);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + AHEAD < NU) fetch(u + AHEAD);
                const int col = u / KS, ty = u - col * KS;
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt)
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u % NB], rows[col & 1][ty + pt], acc[pt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            if constexpr (MASK && MASK_LATE) load_mask();
            // epilogue: lane = pixel; its 16 accumulators are rows 8g + 4 half + i (g = reg >> 2, i = reg & 3).
            // The two halves swap two row groups so that each lane stores whole 16-byte runs of 8 channels.
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                unsigned pk[4][2];
                const bool px_in = ty0 + NPT * wave + pt < p.H && tx0 + px_l < p.W;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (MASK) {
                            v[i] = acc[pt][4 * g + i];
                            if (g < 2 * K::SPT) {
                                // (a ReLU output is never negative: positive = any bit below the sign)
                                const unsigned word = i < 2 ? mk[pt][g].x : mk[pt][g].y;
                                const bool keep = (word & ((i & 1) ? 0x7FFF0000u : 0x00007FFFu)) != 0u;
                                v[i] = keep ? v[i] : 0.0f;
                                bsum[g][i] += px_in ? v[i] : 0.0f;
                            }
                        } else {
                            v[i] = acc[pt][4 * g + i] + lds_bias[8 * g + 4 * half + i];
                            if (p.relu) v[i] = fmaxf(v[i], 0.0f);
                        }
                    }
                    pk[g][0] = pack2(v[0], v[1]), pk[g][1] = pack2(v[2], v[3]);
                }
                unsigned recv[K::SPT][2];
#pragma unroll
                for (int c = 0; c < K::SPT; ++c)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const unsigned send = half ? pk[2 * c][w] : pk[2 * c + 1][w];
                        recv[c][w] = __shfl_xor(send, 32);
                    }
                const int gy = ty0 + NPT * wave + pt, gx = tx0 + px_l;
                unsigned out[K::SPT][4];
#pragma unroll
                for (int c = 0; c < K::SPT; ++c) {
                    // (the row group this lane keeps, 2 c + half, by selects: a register array indexed with a
                    // per-lane value would live in scratch memory)
                    const unsigned k0 = half ? pk[2 * c + 1][0] : pk[2 * c][0], k1 = half ? pk[2 * c + 1][1] : pk[2 * c][1];
                    out[c][0] = half ? recv[c][0] : k0, out[c][1] = half ? recv[c][1] : k1;
                    out[c][2] = half ? k0 : recv[c][0], out[c][3] = half ? k1 : recv[c][1];
                }
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifdef ISG_C32_NO_STORE  // (diagnostic build: the epilogue without its stores)
                if (gy < p.H && gx < p.W && p.relu == 12345) {
#else
                if (gy < p.H && gx < p.W) {
#endif
                    char* dst = static_cast<char*>(p.y) + (((size_t)b * p.H + gy) * p.W + gx) * (size_t)(unsigned)p.y_pstride;
#pragma unroll
                    for (int c = 0; c < K::SPT; ++c)
                        *reinterpret_cast<u32x4*>(dst + (2 * c + half) * 16) = u32x4{out[c][0], out[c][1], out[c][2], out[c][3]};
                }
            }
        }
        __syncthreads();  // every wave is done reading the tile
        stage_store(k + 1);
        __syncthreads();
    }
    if constexpr (MASK) {
        // per row: the 32 pixels of a half-wave, then the 8 waves through LDS (the tile buffer is free now)
        float* red = reinterpret_cast<float*>(lds_in);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = bsum[g][i];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) s += __shfl_xor(s, o);
                if (px_l == 0) red[(wave * 2 + half) * 16 + 4 * g + i] = s;
            }
        __syncthreads();
        if (tid < MROWS) {
            const int g = tid >> 3, hf = (tid >> 2) & 1, i = tid & 3;
            float s = 0.0f;
            for (int w = 0; w < kThreads / 64; ++w) s += red[(w * 2 + hf) * 16 + 4 * g + i];
            p.bias_partial[(size_t)blockIdx.x * MROWS + tid] = s;
        }
    }
}

// dbias[c] = sum over workgroups of bias_partial[workgroup][c], in a fixed order
__global__ __launch_bounds__(256) void conv32_bias_reduce_kernel(const float* __restrict__ partial, int nblocks, int CH,
                                                                 float* __restrict__ dbias) {
    __shared__ float red[8][MROWS];
    const int c = threadIdx.x & (MROWS - 1), part = threadIdx.x >> 5;
    float s = 0.0f;
    for (int b = part; b < nblocks; b += 8) s += partial[(size_t)b * MROWS + c];
    red[part][c] = s;
    __syncthreads();
    if (part == 0 && c < CH) {
        float v = 0.0f;
        for (int k = 0; k < 8; ++k) v += red[k][c];
        dbias[c] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------
constexpr int kWgBlocks = 256;

template <int KS, int CH>
struct CWG {
    static constexpr int NCHK = CH / 8;                 // real 16-byte chunks per pixel (the LDS rows keep 4)
    static constexpr int TH = 8;
    static constexpr int PAD = (KS - 1) / 2;
    static constexpr int NT = KS * KS;                  // one 32 x 32 output tile per tap
    static constexpr int BASE = NT / 4, EXTRA = NT % 4; // tiles per wave group: BASE (+1 for the first EXTRA groups)
    static constexpr int TPW = BASE + (EXTRA ? 1 : 0);
    static constexpr int HH = TH + KS - 1, HWP = TW + KS - 1;
    static constexpr int NPIX = HH * HWP;
    static constexpr int IN_BYTES = NPIX * 64;          // unpadded 64-byte pixel rows: conflict-free transposing reads
    static constexpr int DZ_BYTES = TH * TW * 64;
    static constexpr int BUF_BYTES = IN_BYTES + DZ_BYTES;
    static constexpr int NLX = (NPIX * 4 + kThreads - 1) / kThreads;
    static constexpr int NLZ = (TH * TW * 4) / kThreads;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static_assert(4 * 16 * 64 * 4 <= LDS_BYTES, "reduction scratch");
};

struct Conv32WgradParams {
    const void* x;   // [B,H,W,CH] bf16
    const void* dz;  // [B,H,W,CH] bf16
    float* slab;     // [blocks][NT][32][32]
    int B, H, W, tiles_x, tiles_y;
};

template <int KS, int CH>
__global__ __launch_bounds__(kThreads) void conv32_wgrad_kernel(Conv32WgradParams p) {
    using K = CWG<KS, CH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int kh = wave & 1, grp = wave >> 1;  // pixel half x tap group; the halves are summed through LDS at the end
    const int t0 = grp * K::BASE + min(grp, K::EXTRA);
    const int nt = K::BASE + (grp < K::EXTRA ? 1 : 0);

    f32x16 acc[K::TPW];
#pragma unroll
    for (int i = 0; i < K::TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.B;
    const int n_iter = (ntiles + gridDim.x - 1) / gridDim.x;

    uint4 stx[K::NLX], stz[K::NLZ];
    auto stage_load = [&](int k) {
        const int tile = k * gridDim.x + blockIdx.x;
        if (k >= n_iter || tile >= ntiles) return;
        const int b = tile / tiles_per_img;
        const int rem = tile - b * tiles_per_img;
        const int tyi = rem / p.tiles_x;
        const int ty0 = tyi * K::TH, tx0 = (rem - tyi * p.tiles_x) * TW;
#pragma unroll
        for (int j = 0; j < K::NLX; ++j) {
            const int idx = tid + kThreads * j;
            const int px = idx >> 2, ch = idx & 3;
            const int hy = px / K::HWP, hx = px - hy * K::HWP;
            const int gy = ty0 + hy - K::PAD, gx = tx0 + hx - K::PAD;
            // (16 channels: the LDS rows keep their 64 bytes, the upper half zero - same reads, half-empty tiles)
            const bool inside = px < K::NPIX && ch < K::NCHK && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            stx[j] = make_uint4(0u, 0u, 0u, 0u);
            if (inside) stx[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.x) +
                                                                 (((size_t)b * p.H + gy) * p.W + gx) * (CH * 2) + ch * 16);
        }
#pragma unroll
        for (int j = 0; j < K::NLZ; ++j) {
            const int idx = tid + kThreads * j;
            const int px = idx >> 2, ch = idx & 3;
            const int gy = ty0 + px / TW, gx = tx0 + (px & (TW - 1));
            stz[j] = make_uint4(0u, 0u, 0u, 0u);
            if (gy < p.H && gx < p.W && ch < K::NCHK)
                stz[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.dz) +
                                                         (((size_t)b * p.H + gy) * p.W + gx) * (CH * 2) + ch * 16);
        }
    };
    auto stage_store = [&](int k, char* buf) {
        const int tile = k * gridDim.x + blockIdx.x;
        if (k >= n_iter || tile >= ntiles) return;
#pragma unroll
        for (int j = 0; j < K::NLX; ++j) {
            const int idx = tid + kThreads * j;
            if (idx < K::NPIX * 4) *reinterpret_cast<uint4*>(buf + idx * 16) = stx[j];
        }
#pragma unroll
        for (int j = 0; j < K::NLZ; ++j) {
            const int idx = tid + kThreads * j;
            *reinterpret_cast<uint4*>(buf + K::IN_BYTES + idx * 16) = stz[j];
        }
    };
    auto compute = [&](const char* buf) {
        const char* lds_in = buf;
        const char* lds_dz = buf + K::IN_BYTES;
        const int li = lane & 15, g16 = lane >> 4;
        const int q = li >> 2, pp4 = li & 3;
        const int csub = 16 * (g16 & 1) + 4 * pp4;  // channel offset inside the 32 rows
        const int pj = 16 * kh + 8 * half + q;
        auto tr_pair = [](const char* a0) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * 64));
            const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, both);
        };
        // (the fragments of the next tile row are requested before this row's MFMAs)
        const bool last_full = nt == K::TPW;
        bf16x8 bq[2], aq[2][K::TPW];
        auto fetch = [&](int r, int slot) {
            bq[slot] = tr_pair(lds_dz + (r * TW + pj) * 64 + csub * 2);
#pragma unroll
            for (int i = 0; i < K::TPW; ++i)
                if (i < K::BASE || last_full) {  // (only the last slot depends on the wave group)
                    const int tap = t0 + i;
                    const int ty = tap / KS, tx = tap - ty * KS;
                    aq[slot][i] = tr_pair(lds_in + ((r + ty) * K::HWP + pj + tx) * 64 + csub * 2);
                }
        };
        fetch(0, 0);
#pragma unroll
        for (int r = 0; r < K::TH; ++r) {
            if (r + 1 < K::TH) fetch(r + 1, (r + 1) & 1);
#pragma unroll
            for (int i = 0; i < K::TPW; ++i)
                if (i < K::BASE || last_full)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[r & 1][i], bq[r & 1], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    stage_load(0);
    stage_store(0, smem);
    __syncthreads();
    for (int k = 0; k < n_iter; ++k) {
        stage_load(k + 1);
        if (k * (int)gridDim.x + (int)blockIdx.x < ntiles) compute(smem + (k & 1) * K::BUF_BYTES);
        stage_store(k + 1, smem + ((k + 1) & 1) * K::BUF_BYTES);
        __syncthreads();
    }
    float* red = reinterpret_cast<float*>(smem) + (size_t)grp * (16 * 64);
#pragma unroll
    for (int i = 0; i < K::TPW; ++i) {
        if (kh == 1 && i < nt) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) red[reg * 64 + lane] = acc[i][reg];
        }
        __syncthreads();
        if (kh == 0 && i < nt) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[i][reg] += red[reg * 64 + lane];
        }
        __syncthreads();
    }
    if (kh != 0) return;
    float* slab = p.slab + (size_t)blockIdx.x * K::NT * 32 * 32;
#pragma unroll
    for (int i = 0; i < K::TPW; ++i) {
        if (i < nt) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                slab[((size_t)(t0 + i) * 32 + row) * 32 + (lane & 31)] = acc[i][reg];
            }
        }
    }
}

// dw[tap][cin][cout] = sum over blocks of slab[block][tap][cin][cout] (32 x 32 per tap, of which ch x ch are real).
// A workgroup owns 64 outputs; 4 groups of 64 threads sum a quarter of the slabs each.
__global__ __launch_bounds__(256) void conv32_wgrad_reduce_kernel(const float* __restrict__ slab, int nblocks, int ntaps,
                                                                int ch, float* __restrict__ dw) {
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o, n = ntaps * ch * ch;
    float s = 0.0f;
    if (idx < n) {
        const int tap = idx / (ch * ch), r = idx - tap * ch * ch, cin = r / ch, cout = r - cin * ch;
        const size_t at = ((size_t)tap * 32 + cin) * 32 + cout, per_block = (size_t)ntaps * 32 * 32;
        for (int b = sg; b < nblocks; b += 4) s += slab[(size_t)b * per_block + at];
    }
    red[sg][o] = s;
    __syncthreads();
    if (sg == 0 && idx < n) dw[idx] = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
}

template <int KS, int CH, bool MASK>
int launch_conv32(Conv32Params& p, float* dbias, hipStream_t s) {
    using K = CF<KS, CH>;
    static_assert(K::LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(K::IN_BYTES >= 2 * (kThreads / 64) * 16 * 4, "the closing bias reduction borrows the tile buffer");
    p.tiles_x = (p.W + TW - 1) / TW, p.tiles_y = (p.H + K::TH - 1) / K::TH;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const int per_cu = (160 * 1024) / K::LDS_BYTES >= 2 ? 2 : 1;
    int grid = ntiles < per_cu * num_cus() ? ntiles : per_cu * num_cus();
    if (MASK && grid > kMaskedMaxBlocks) grid = kMaskedMaxBlocks;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv32_kernel<KS, CH, MASK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((conv32_kernel<KS, CH, MASK>), dim3(grid), dim3(kThreads), K::LDS_BYTES, s, p);
    int rc = isg::check_launch("conv32_kernel");
    if (!MASK || rc != ISG_OK) return rc;
    hipLaunchKernelGGL(conv32_bias_reduce_kernel, dim3(1), dim3(256), 0, s, p.bias_partial, grid, CH, dbias);
    return isg::check_launch("conv32_bias_reduce_kernel");
}

template <int KS, int CH>
int launch_conv32_wgrad(Conv32WgradParams& p, float* dw, hipStream_t s) {
    using K = CWG<KS, CH>;
    static_assert(K::LDS_BYTES <= 160 * 1024, "LDS budget");
    p.tiles_x = (p.W + TW - 1) / TW, p.tiles_y = (p.H + K::TH - 1) / K::TH;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const int grid = ntiles < kWgBlocks ? ntiles : kWgBlocks;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv32_wgrad_kernel<KS, CH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((conv32_wgrad_kernel<KS, CH>), dim3(grid), dim3(kThreads), K::LDS_BYTES, s, p);
    int rc = isg::check_launch("conv32_wgrad_kernel");
    if (rc != ISG_OK) return rc;
    const int n = K::NT * CH * CH;
    hipLaunchKernelGGL(conv32_wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, s, p.slab, grid, K::NT, CH, dw);
    return isg::check_launch("conv32_wgrad_reduce_kernel");
}

int conv_same(const char* who, int CH, const void* x, const void* wimg, const float* bias, int B, int H, int W, int KS,
              int relu, void* y, void* stream, long long y_pixel_stride = 0) {
    ISG_REQUIRE(KS == 3 || KS == 5, ISG_EINVAL, "%s: kernel size %d has no instantiation (3, 5)", who, KS);
    ISG_REQUIRE(B >= 1 && H >= 1 && W >= 1, ISG_EINVAL, "%s: bad shape", who);
    ISG_REQUIRE(x && wimg && y, ISG_EINVAL, "%s: null pointer", who);
    if (y_pixel_stride == 0) y_pixel_stride = CH;
    ISG_REQUIRE(y_pixel_stride >= CH && y_pixel_stride % 8 == 0 && y_pixel_stride <= (1 << 20) && (uintptr_t)y % 16 == 0,
                ISG_EINVAL, "%s: y_pixel_stride %lld (a multiple of 8, at least %d) on a 16-byte aligned y", who,
                y_pixel_stride, CH);
    Conv32Params p = {};
    p.x = x, p.wimg = wimg, p.bias = bias, p.y = y, p.B = B, p.H = H, p.W = W, p.relu = relu;
    p.y_pstride = (int)y_pixel_stride * 2;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (CH == 32) return KS == 3 ? launch_conv32<3, 32, false>(p, nullptr, s) : launch_conv32<5, 32, false>(p, nullptr, s);
    return KS == 3 ? launch_conv32<3, 16, false>(p, nullptr, s) : launch_conv32<5, 16, false>(p, nullptr, s);
}

int conv_same_masked(const char* who, int CH, const void* x, const void* wimg, const void* mask, int B, int H, int W,
                     int KS, void* y, float* dbias, void* workspace, void* stream) {
    ISG_REQUIRE(KS == 3 || KS == 5, ISG_EINVAL, "%s: kernel size %d has no instantiation (3, 5)", who, KS);
    ISG_REQUIRE(B >= 1 && H >= 1 && W >= 1, ISG_EINVAL, "%s: bad shape", who);
    ISG_REQUIRE(x && wimg && mask && y && dbias && workspace, ISG_EINVAL, "%s: null pointer", who);
    Conv32Params p = {};
    p.x = x, p.wimg = wimg, p.y = y, p.B = B, p.H = H, p.W = W, p.y_pstride = CH * 2;
    p.mask = mask, p.bias_partial = static_cast<float*>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (CH == 32) return KS == 3 ? launch_conv32<3, 32, true>(p, dbias, s) : launch_conv32<5, 32, true>(p, dbias, s);
    return KS == 3 ? launch_conv32<3, 16, true>(p, dbias, s) : launch_conv32<5, 16, true>(p, dbias, s);
}

int conv_same_wgrad(const char* who, int CH, const void* x, const void* dz, int B, int H, int W, int KS, void* workspace,
                    float* dw, void* stream) {
    ISG_REQUIRE(KS == 3 || KS == 5, ISG_EINVAL, "%s: kernel size %d has no instantiation (3, 5)", who, KS);
    ISG_REQUIRE(B >= 1 && H >= 1 && W >= 1, ISG_EINVAL, "%s: bad shape", who);
    ISG_REQUIRE(x && dz && workspace && dw, ISG_EINVAL, "%s: null pointer", who);
    Conv32WgradParams p = {};
    p.x = x, p.dz = dz, p.slab = static_cast<float*>(workspace), p.B = B, p.H = H, p.W = W;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (CH == 32) return KS == 3 ? launch_conv32_wgrad<3, 32>(p, dw, s) : launch_conv32_wgrad<5, 32>(p, dw, s);
    return KS == 3 ? launch_conv32_wgrad<3, 16>(p, dw, s) : launch_conv32_wgrad<5, 16>(p, dw, s);
}

}  // namespace

extern "C" long long isg_conv32_packed_elems(int KS) {
    if (KS != 3 && KS != 5) return -1;
    return (long long)KS * KS * 2 * 64 * 8;
}

extern "C" int isg_conv32(const void* x, const void* wimg, const float* bias, int B, int H, int W, int KS, int relu,
                          void* y, void* stream) {
    return conv_same("isg_conv32", 32, x, wimg, bias, B, H, W, KS, relu, y, stream);
}

extern "C" int isg_conv32_strided(const void* x, const void* wimg, const float* bias, int B, int H, int W, int KS,
                                  int relu, void* y, long long y_pixel_stride, void* stream) {
    return conv_same("isg_conv32_strided", 32, x, wimg, bias, B, H, W, KS, relu, y, stream, y_pixel_stride);
}

extern "C" long long isg_conv32_masked_workspace_bytes(void) { return (long long)kMaskedMaxBlocks * MROWS * (long long)sizeof(float); }

extern "C" int isg_conv32_masked(const void* x, const void* wimg, const void* mask, int B, int H, int W, int KS, void* y,
                                 float* dbias, void* workspace, void* stream) {
    return conv_same_masked("isg_conv32_masked", 32, x, wimg, mask, B, H, W, KS, y, dbias, workspace, stream);
}

extern "C" int isg_conv16_masked(const void* x, const void* wimg, const void* mask, int B, int H, int W, int KS, void* y,
                                 float* dbias, void* workspace, void* stream) {
    return conv_same_masked("isg_conv16_masked", 16, x, wimg, mask, B, H, W, KS, y, dbias, workspace, stream);
}

extern "C" long long isg_conv32_wgrad_workspace_bytes(int KS) {
    if (KS != 3 && KS != 5) return -1;
    return (long long)kWgBlocks * KS * KS * 32 * 32 * (long long)sizeof(float);
}

extern "C" int isg_conv32_wgrad(const void* x, const void* dz, int B, int H, int W, int KS, void* workspace, float* dw,
                                void* stream) {
    return conv_same_wgrad("isg_conv32_wgrad", 32, x, dz, B, H, W, KS, workspace, dw, stream);
}

// the same four for 16 -> 16 channels (one k-step per tap; dw is [KS][KS][16][16])
extern "C" long long isg_conv16_packed_elems(int KS) {
    if (KS != 3 && KS != 5) return -1;
    return (long long)KS * KS * 64 * 8;
}

extern "C" int isg_conv16(const void* x, const void* wimg, const float* bias, int B, int H, int W, int KS, int relu,
                          void* y, void* stream) {
    return conv_same("isg_conv16", 16, x, wimg, bias, B, H, W, KS, relu, y, stream);
}

extern "C" long long isg_conv16_wgrad_workspace_bytes(int KS) { return isg_conv32_wgrad_workspace_bytes(KS); }

extern "C" int isg_conv16_wgrad(const void* x, const void* dz, int B, int H, int W, int KS, void* workspace, float* dw,
                                void* stream) {
    return conv_same_wgrad("isg_conv16_wgrad", 16, x, dz, B, H, W, KS, workspace, dw, stream);
}
