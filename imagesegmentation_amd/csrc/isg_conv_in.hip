// The encoder's first convolution: KS x KS taps over an image with at most 4 channels -> CO = 32 or 16
// channels, TensorFlow SAME padding, + bias + ReLU in the epilogue (reference polyrnn/RNNConvExperiment's
// 14 x 14 -> 32 layer and the 5 x 5 -> 16 layer of polyrnn/History2LargeExperiment.py:15-22;
// tf.layers.conv2d(..., activation=tf.nn.relu)).  With 16 output channels half the MFMA rows are zero.
// A library implicit GEMM spends most of its K on padding when the image has 3 channels; here the
// reduction index is (ky, kx, c) with c padded to 4 only, so one 16-deep MFMA step covers 4
// horizontal taps, and the B operand of a lane is 16 contiguous bytes of the staged image tile.
//   D[out channel][pixel] = sum_k W[out channel][k] * X[k][pixel],   32x32x16 bf16 MFMA
// Persistent workgroups, weight image resident in LDS, image tile double-buffered through registers.
#include <hip/hip_bf16.h>

#include <type_traits>

#include "isg_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 32, MROWS = 32, kThreads = 256;

template <int KS>
struct CI {
    static constexpr int PAD = (KS - 1) / 2;          // SAME: the smaller half goes before
    static constexpr int KXP = (KS + 3) & ~3;         // horizontal taps padded to whole k-steps (zero weights)
    static constexpr int SPK = KXP / 4;               // k-steps per kernel row: 4 taps x 4 channels
    static constexpr int NS = KS * SPK;
    static constexpr int HH = TH + KS - 1;            // staged rows
    static constexpr int ROWPX = TW + KXP;            // staged pixels per row (covers px + kx, even count)
    static constexpr int NPIX = HH * ROWPX;
    static constexpr int IN_BYTES = NPIX * 8;         // 4 channels x bf16 per pixel
    static constexpr int W_BYTES = NS * 1024;
    static constexpr int NLOAD = (NPIX + kThreads - 1) / kThreads;
    static constexpr int LDS_BYTES = W_BYTES + 2 * IN_BYTES + MROWS * 4;
};

struct ConvInParams {
    const void* x4;    // [B,H,W,4] bf16, channel 3 = 0
    const void* wimg;  // [NS][64 lanes][8] bf16 A fragments
    const float* bias; // [CO]
    void* y;           // [B,H,W,CO] bf16
    int B, H, W, tiles_x, tiles_y, relu;
};

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

template <int KS, int CO>
__global__ __launch_bounds__(kThreads) void conv_in_fwd_kernel(ConvInParams p) {
    static_assert(CO == 16 || CO == 32, "output channels");
    using K = CI<KS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_w = smem;
    char* lds_in = smem + K::W_BYTES;
    float* lds_bias = reinterpret_cast<float*>(smem + K::W_BYTES + 2 * K::IN_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px_l = lane & 31, half = lane >> 5;
    {
        const uint4* src = static_cast<const uint4*>(p.wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds_w);
        for (int i = tid; i < K::W_BYTES / 16; i += kThreads) dst[i] = src[i];
        if (tid < MROWS) lds_bias[tid] = tid < CO ? p.bias[tid] : 0.0f;
    }
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.B;
    const int n_iter = (ntiles + gridDim.x - 1) / gridDim.x;

    uint2 st[K::NLOAD];
    auto tile_origin = [&](int k, int& b, int& ty0, int& tx0) -> bool {
        const int tile = k * gridDim.x + blockIdx.x;
        if (k >= n_iter || tile >= ntiles) return false;
        b = tile / tiles_per_img;
        const int rem = tile - b * tiles_per_img;
        const int tyi = rem / p.tiles_x;
        ty0 = tyi * TH, tx0 = (rem - tyi * p.tiles_x) * TW;
        return true;
    };
    auto stage_load = [&](int k) {
        int b, ty0, tx0;
        if (!tile_origin(k, b, ty0, tx0)) return;
#pragma unroll
        for (int j = 0; j < K::NLOAD; ++j) {
            const int idx = tid + kThreads * j;
            const int hy = idx / K::ROWPX, hx = idx - hy * K::ROWPX;
            const int gy = ty0 + hy - K::PAD, gx = tx0 + hx - K::PAD;
            const bool inside = idx < K::NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            st[j] = make_uint2(0u, 0u);
            if (inside) st[j] = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.x4) +
                                                                (((size_t)b * p.H + gy) * p.W + gx) * 8);
        }
    };
    auto stage_store = [&](int k, char* buf) {
        const int tile = k * gridDim.x + blockIdx.x;
        if (k >= n_iter || tile >= ntiles) return;
#pragma unroll
        for (int j = 0; j < K::NLOAD; ++j) {
            const int idx = tid + kThreads * j;
            if (idx < K::NPIX) *reinterpret_cast<uint2*>(buf + idx * 8) = st[j];
        }
    };

    stage_load(0);
    stage_store(0, lds_in);
    __syncthreads();
    for (int k = 0; k < n_iter; ++k) {
        int b, ty0, tx0;
        const bool valid = tile_origin(k, b, ty0, tx0);
        stage_load(k + 1);
        if (valid) {
            const char* buf = lds_in + (k & 1) * K::IN_BYTES;
            f32x16 acc[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pt][r] = 0.0f;
            const char* bb = buf + ((2 * wave) * K::ROWPX + px_l + 2 * half) * 8;
            const char* aw = lds_w + lane * 16;
            // (the fragments are requested two k-steps ahead of their MFMAs)
#ifndef ISG_CIN_AHEAD
#define ISG_CIN_AHEAD 2
#endif
            constexpr int AHEAD = ISG_CIN_AHEAD, NB = AHEAD + 1;
            bf16x8 af[NB], bfr[NB][2];
            auto fetch = [&](int ks, int slot) {
                const int ky = ks / K::SPK, s = ks - ky * K::SPK;
                af[slot] = *reinterpret_cast<const bf16x8*>(aw + ks * 1024);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const char* src = bb + ((ky + pt) * K::ROWPX + 4 * s) * 8;
                    const uint2 lo = *reinterpret_cast<const uint2*>(src);
                    const uint2 hi = *reinterpret_cast<const uint2*>(src + 8);
                    bfr[slot][pt] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                }
            };
#pragma unroll
            for (int ks = 0; ks < AHEAD; ++ks) fetch(ks, ks);
#pragma unroll
            for (int ks = 0; ks < K::NS; ++ks) {
                if (ks + AHEAD < K::NS) fetch(ks + AHEAD, (ks + AHEAD) % NB);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks % NB], bfr[ks % NB][pt], acc[pt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue: lane = pixel; its 16 accumulators are channels 8g + 4 half + i (g = reg >> 2, i = reg & 3).
            // The two halves swap two channel groups so that each lane stores whole 16-byte runs of 8 channels.
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                unsigned pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = acc[pt][4 * g + i] + lds_bias[8 * g + 4 * half + i];
                        if (p.relu) v[i] = fmaxf(v[i], 0.0f);
                    }
                    pk[g][0] = pack2(v[0], v[1]), pk[g][1] = pack2(v[2], v[3]);
                }
                // half 0 keeps groups 0 and 2 (stores chunks 0 and 2), half 1 keeps 1 and 3 (chunks 1 and 3)
                unsigned recv[CO / 16][2];
#pragma unroll
                for (int c = 0; c < CO / 16; ++c)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const unsigned send = half ? pk[2 * c][w] : pk[2 * c + 1][w];
                        recv[c][w] = __shfl_xor(send, 32);
                    }
                const int gy = ty0 + 2 * wave + pt, gx = tx0 + px_l;
                unsigned out[CO / 16][4];
#pragma unroll
                for (int c = 0; c < CO / 16; ++c) {
                    // (the group this lane keeps, 2 c + half, by selects: a register array indexed with a per-lane
                    // value would live in scratch memory)
                    const unsigned k0 = half ? pk[2 * c + 1][0] : pk[2 * c][0], k1 = half ? pk[2 * c + 1][1] : pk[2 * c][1];
                    out[c][0] = half ? recv[c][0] : k0, out[c][1] = half ? recv[c][1] : k1;
                    out[c][2] = half ? k0 : recv[c][0], out[c][3] = half ? k1 : recv[c][1];
                }
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                if (gy < p.H && gx < p.W) {
                    char* dst = static_cast<char*>(p.y) + (((size_t)b * p.H + gy) * p.W + gx) * (CO * 2);
#pragma unroll
                    for (int c = 0; c < CO / 16; ++c)
                        *reinterpret_cast<u32x4*>(dst + (2 * c + half) * 16) = u32x4{out[c][0], out[c][1], out[c][2], out[c][3]};
                }
            }
        }
        stage_store(k + 1, lds_in + ((k + 1) & 1) * K::IN_BYTES);
        __syncthreads();
    }
}

// [B,H,W,3] -> [B,H,W,4] with a zero fourth channel (8-byte pixels for the staging loads)
__global__ __launch_bounds__(256) void pad_channels_3to4_kernel(const __bf16* __restrict__ x, long long npix,
                                                               __bf16* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 v = {x[3 * i], x[3 * i + 1], x[3 * i + 2], (__bf16)0.0f};
        *reinterpret_cast<bf16x4*>(out + 4 * i) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dW[ky][kx][c][o] = sum over pixels of X[pixel + (ky, kx)][c] * dZ[pixel][o]
// ---------------------------------------------------------------------------------------------
// For one kernel row ky the 16 x 4 (kx, c) pairs are 64 "virtual channels" of the staged image
// tile: virtual channel v of pixel p is channel v % 4 of pixel p + v / 4, i.e. 2 bytes at
// (p * 8 + v * 2) - the same address arithmetic as a real 64-channel pixel row with an 8-byte pixel
// stride.  Both MFMA operands are then read with the transposing LDS read (the reduction runs over
// pixels), D[virtual channel][out channel], 2 row tiles x 14 kernel rows = 28 output tiles shared by
// 8 waves: wave = (pixel half kh) x (7 tiles), the two halves summed through LDS at the end.
constexpr int kWgThreads = 512;
constexpr int kWgBlocks = 256;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int KS>
struct CW {
    using F = CI<KS>;
    static constexpr int MTW = F::KXP * 4 / 32;          // 32-row tiles per kernel row: (kx, c) pairs / 32
    static constexpr int NT = MTW * KS;                  // output tiles (ky, row tile)
    static constexpr int TPW = (NT + 3) / 4;             // tiles per wave group
    static constexpr int DZ_BYTES = TH * TW * 64;
    static constexpr int BUF_BYTES = F::IN_BYTES + DZ_BYTES;
    static constexpr int NLX = (F::NPIX + kWgThreads - 1) / kWgThreads;
    static constexpr int NLZ = (TH * TW * 4) / kWgThreads;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static_assert(F::KXP % 8 == 0, "whole 32-row tiles of 8 taps x 4 channels");
    static_assert((TH * TW * 4) % kWgThreads == 0, "dZ tile must fill whole passes");
    static_assert(4 * 16 * 64 * 4 <= LDS_BYTES, "reduction scratch");
};

struct ConvInWgradParams {
    const void* x4;  // [B,H,W,4] bf16
    const void* dz;  // [B,H,W,CO] bf16
    float* slab;     // [blocks][NT][32][32]
    int B, H, W, tiles_x, tiles_y;
};

template <int KS, int CO>
__global__ __launch_bounds__(kWgThreads) void conv_in_wgrad_kernel(ConvInWgradParams p) {
    using K = CW<KS>;
    using F = CI<KS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int kh = wave & 1, grp = wave >> 1;
    const int t0 = grp * K::TPW;
    const int nt = min(K::TPW, K::NT - t0);

    f32x16 acc[K::TPW];
#pragma unroll
    for (int i = 0; i < K::TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.B;
    const int n_iter = (ntiles + gridDim.x - 1) / gridDim.x;

    uint2 stx[K::NLX];
    uint4 stz[K::NLZ];
    auto stage_load = [&](int k) {
        const int tile = k * gridDim.x + blockIdx.x;
        if (k >= n_iter || tile >= ntiles) return;
        const int b = tile / tiles_per_img;
        const int rem = tile - b * tiles_per_img;
        const int tyi = rem / p.tiles_x;
        const int ty0 = tyi * TH, tx0 = (rem - tyi * p.tiles_x) * TW;
#pragma unroll
        for (int j = 0; j < K::NLX; ++j) {
            const int idx = tid + kWgThreads * j;
            const int hy = idx / F::ROWPX, hx = idx - hy * F::ROWPX;
            const int gy = ty0 + hy - F::PAD, gx = tx0 + hx - F::PAD;
            const bool inside = idx < F::NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            stx[j] = make_uint2(0u, 0u);
            if (inside) stx[j] = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.x4) +
                                                                 (((size_t)b * p.H + gy) * p.W + gx) * 8);
        }
#pragma unroll
        for (int j = 0; j < K::NLZ; ++j) {
            const int idx = tid + kWgThreads * j;
            const int px = idx >> 2, ch = idx & 3;
            const int gy = ty0 + px / TW, gx = tx0 + (px & (TW - 1));
            stz[j] = make_uint4(0u, 0u, 0u, 0u);
            if (gy < p.H && gx < p.W && ch < CO / 8)  // (16 channels: the upper half of the 64-byte LDS row stays zero)
                stz[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.dz) +
                                                         (((size_t)b * p.H + gy) * p.W + gx) * (CO * 2) + ch * 16);
        }
    };
    auto stage_store = [&](int k, char* buf) {
        const int tile = k * gridDim.x + blockIdx.x;
        if (k >= n_iter || tile >= ntiles) return;
#pragma unroll
        for (int j = 0; j < K::NLX; ++j) {
            const int idx = tid + kWgThreads * j;
            if (idx < F::NPIX) *reinterpret_cast<uint2*>(buf + idx * 8) = stx[j];
        }
#pragma unroll
        for (int j = 0; j < K::NLZ; ++j) {
            const int idx = tid + kWgThreads * j;
            *reinterpret_cast<uint4*>(buf + F::IN_BYTES + idx * 16) = stz[j];
        }
    };
    auto compute = [&](const char* buf) {
        const char* lds_in = buf;
        const char* lds_dz = buf + F::IN_BYTES;
        const int li = lane & 15, g16 = lane >> 4;
        const int q = li >> 2, pp4 = li & 3;
        const int csub = 16 * (g16 & 1) + 4 * pp4;  // (virtual) channel offset inside a 32-row tile
        const int pj = 16 * kh + 8 * half + q;
        auto tr_pair = [](const char* a0, auto second) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(a0 + decltype(second)::value));
            const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, both);
        };
        // (the fragments of the next tile row are requested before this row's MFMAs; with 28 tiles every wave
        // group has all its slots in use and nothing in the loop is conditional)
        constexpr bool kAllFull = K::NT == 4 * K::TPW;
        bf16x8 bq[2], aq[2][K::TPW];
        auto fetch = [&](int r, int slot) {
            bq[slot] = tr_pair(lds_dz + (r * TW + pj) * 64 + csub * 2, std::integral_constant<int, 4 * 64>{});
#pragma unroll
            for (int i = 0; i < K::TPW; ++i)
                if (kAllFull || i < nt) {
                    const int t = t0 + i;
                    const int ky = t / K::MTW, mt = t % K::MTW;
                    aq[slot][i] = tr_pair(lds_in + ((r + ky) * F::ROWPX + pj) * 8 + (32 * mt + csub) * 2,
                                          std::integral_constant<int, 4 * 8>{});
                }
        };
        fetch(0, 0);
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            if (r + 1 < TH) fetch(r + 1, (r + 1) & 1);
#pragma unroll
            for (int i = 0; i < K::TPW; ++i)
                if (kAllFull || i < nt)
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
    // sum the two pixel halves (waves 2g and 2g + 1), one tile slot at a time
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

// dw[ky][kx][c][o] = sum over blocks of slab[block][MTW ky + (4 kx + c) / 32][(4 kx + c) % 32][o].
// A workgroup owns 64 outputs; 4 groups of 64 threads sum a quarter of the slabs each.
template <int KS, int CO>
__global__ __launch_bounds__(256) void conv_in_wgrad_reduce_kernel(const float* __restrict__ slab, int nblocks,
                                                                 float* __restrict__ dw) {
    using K = CW<KS>;
    __shared__ float red[4][64];
    const int t = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + t;
    const bool has = idx < KS * KS * 3 * CO;
    float s = 0.0f;
    if (has) {
        const int o = idx % CO, c = (idx / CO) % 3, kx = (idx / (CO * 3)) % KS, ky = idx / (CO * 3 * KS);
        const int v = 4 * kx + c;
        const size_t off = ((size_t)(K::MTW * ky + v / 32) * 32 + v % 32) * 32 + o;
        for (int b = sg; b < nblocks; b += 4) s += slab[(size_t)b * K::NT * 1024 + off];
    }
    red[sg][t] = s;
    __syncthreads();
    if (sg == 0 && has) dw[idx] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
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

}  // namespace

namespace {
bool conv_in_has(int KS, int CO) { return (KS == 14 || KS == 5) && (CO == 32 || CO == 16); }

template <int KS, int CO>
int launch_conv_in_fwd(ConvInParams& p, hipStream_t s) {
    using K = CI<KS>;
    static_assert(K::LDS_BYTES <= 160 * 1024, "LDS budget");
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const int grid = ntiles < 2 * num_cus() ? ntiles : 2 * num_cus();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_in_fwd_kernel<KS, CO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((conv_in_fwd_kernel<KS, CO>), dim3(grid), dim3(kThreads), K::LDS_BYTES, s, p);
    return isg::check_launch("conv_in_fwd_kernel");
}

template <int KS, int CO>
int launch_conv_in_wgrad(ConvInWgradParams& p, float* dw, hipStream_t s) {
    using K = CW<KS>;
    static_assert(K::LDS_BYTES <= 160 * 1024, "LDS budget");
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const int grid = ntiles < kWgBlocks ? ntiles : kWgBlocks;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_in_wgrad_kernel<KS, CO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((conv_in_wgrad_kernel<KS, CO>), dim3(grid), dim3(kWgThreads), K::LDS_BYTES, s, p);
    int rc = isg::check_launch("conv_in_wgrad_kernel");
    if (rc != ISG_OK) return rc;
    const int n_out = KS * KS * 3 * CO;
    hipLaunchKernelGGL((conv_in_wgrad_reduce_kernel<KS, CO>), dim3((n_out + 63) / 64), dim3(256), 0, s, p.slab, grid, dw);
    return isg::check_launch("conv_in_wgrad_reduce_kernel");
}
}  // namespace

extern "C" long long isg_conv_in_packed_elems(int KS, int cout) {
    if (!conv_in_has(KS, cout)) return -1;
    return (long long)(KS == 14 ? CI<14>::NS : CI<5>::NS) * 64 * 8;
}

extern "C" int isg_conv_in_pad_image(const void* x3, long long npix, void* x4, void* stream) {
    ISG_REQUIRE(npix >= 0, ISG_EINVAL, "isg_conv_in_pad_image: bad size");
    if (npix == 0) return ISG_OK;
    ISG_REQUIRE(x3 && x4, ISG_EINVAL, "isg_conv_in_pad_image: null pointer");
    const int blocks = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
    hipLaunchKernelGGL(pad_channels_3to4_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(x3), npix, static_cast<__bf16*>(x4));
    return isg::check_launch("pad_channels_3to4_kernel");
}

extern "C" int isg_conv_in_fwd(const void* x4, const void* wimg, const float* bias, int B, int H, int W, int KS, int cout,
                               int relu, void* y, void* stream) {
    ISG_REQUIRE(conv_in_has(KS, cout), ISG_EINVAL, "isg_conv_in_fwd: no instantiation for %d x %d taps -> %d channels "
                "(14 or 5 taps, 32 or 16 channels)", KS, KS, cout);
    ISG_REQUIRE(B >= 1 && H >= 1 && W >= 1, ISG_EINVAL, "isg_conv_in_fwd: bad shape");
    ISG_REQUIRE(x4 && wimg && bias && y, ISG_EINVAL, "isg_conv_in_fwd: null pointer");
    ConvInParams p = {};
    p.x4 = x4, p.wimg = wimg, p.bias = bias, p.y = y, p.B = B, p.H = H, p.W = W, p.relu = relu;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + TH - 1) / TH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (KS == 14) return cout == 32 ? launch_conv_in_fwd<14, 32>(p, s) : launch_conv_in_fwd<14, 16>(p, s);
    return cout == 32 ? launch_conv_in_fwd<5, 32>(p, s) : launch_conv_in_fwd<5, 16>(p, s);
}

extern "C" long long isg_conv_in_wgrad_workspace_bytes(int KS, int cout) {
    if (!conv_in_has(KS, cout)) return -1;
    return (long long)kWgBlocks * (KS == 14 ? CW<14>::NT : CW<5>::NT) * 32 * 32 * (long long)sizeof(float);
}

extern "C" int isg_conv_in_wgrad(const void* x4, const void* dz, int B, int H, int W, int KS, int cout, void* workspace,
                                 float* dw, void* stream) {
    ISG_REQUIRE(conv_in_has(KS, cout), ISG_EINVAL, "isg_conv_in_wgrad: no instantiation for %d x %d taps -> %d channels "
                "(14 or 5 taps, 32 or 16 channels)", KS, KS, cout);
    ISG_REQUIRE(B >= 1 && H >= 1 && W >= 1, ISG_EINVAL, "isg_conv_in_wgrad: bad shape");
    ISG_REQUIRE(x4 && dz && workspace && dw, ISG_EINVAL, "isg_conv_in_wgrad: null pointer");
    ConvInWgradParams p = {};
    p.x4 = x4, p.dz = dz, p.slab = static_cast<float*>(workspace), p.B = B, p.H = H, p.W = W;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + TH - 1) / TH;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (KS == 14) return cout == 32 ? launch_conv_in_wgrad<14, 32>(p, dw, s) : launch_conv_in_wgrad<14, 16>(p, dw, s);
    return cout == 32 ? launch_conv_in_wgrad<5, 32>(p, dw, s) : launch_conv_in_wgrad<5, 16>(p, dw, s);
}
