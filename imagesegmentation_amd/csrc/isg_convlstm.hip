// ConvLSTM cell kernels for gfx950: fused gate convolution (implicit GEMM on MFMA) + gate math,
// and the three backward pieces (pointwise gate backward, data gradient, weight gradient).
//
// Orientation of every MFMA product here:  D[row][pixel] = sum_k A[row][k] * B[k][pixel]
//   A = packed weights (rows = gate columns or input channels), B = activations staged in LDS,
//   so one lane owns ONE pixel column of the 32x32 result and, through the D register map
//   row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), groups of 4 consecutive rows.
// The forward weight image orders its 64 gate rows so that those 4 consecutive rows are the four
// gates (j, i, f, o) of one channel: the LSTM update is then lane-local and each lane stores 8
// contiguous channels of its pixel.
//
// Tile: 8 x 32 output pixels per 256-thread workgroup (4 waves, 64 pixels x 64 rows each), input
// tile with a 1-pixel halo staged in LDS once and reused by all 9 taps; workgroups are persistent
// over tiles so the weight image is loaded into LDS once per workgroup (bf16) or streamed from L2
// (f32, whose MFMA is 16x slower and needs no LDS residency).
#include "isg_convlstm_shared.h"

using namespace isgc;

namespace {

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
// Image layout for both kinds: [k-step s][row tile t][lane 0..63][KG] with the lane's row
// R = 32 t + (lane & 31) and its k-group g = 2 s + (lane >> 5); k-group g covers KG consecutive
// channels of one tap: tap = g / GPT, first channel = (g % GPT) * KG.
//   forward image: rows = 64 permuted gate columns, channels = [x (Cs) | h (F)]
//   dgrad image  : rows = input channels [x (Cs) | h (F)] padded to 32*MT, channels of a tap = the
//                  64 gate columns in the kernels' (channel, gate) order; taps are flipped.  This one
//                  is stored row-tile-major, [t][s][lane][KG], so that a launch can take one tile.
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w_x, const float* __restrict__ w_h, int C, int Cw, int Cs,
                                    int kind, int MT, T* __restrict__ out, long long total) {
    constexpr int KG = Tr<T>::KG;
    const int cpt = kind == 0 ? Cs + F_ : NG;  // channels per tap on the K side
    const int gpt = cpt / KG;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % KG);
        long long r = idx / KG;
        const int lane = (int)(r % 64);
        r /= 64;
        int t, s;
        if (kind == 0) {
            t = (int)(r % MT);
            s = (int)(r / MT);
        } else {
            const long long ns = total / ((long long)KG * 64 * MT);
            s = (int)(r % ns);
            t = (int)(r / ns);
        }
        const int R = 32 * t + (lane & 31);
        const int g = 2 * s + (lane >> 5);
        const int tap = g / gpt;
        const int kc = (g - tap * gpt) * KG + j;  // channel within the tap, K side
        float v = 0.0f;
        if (tap < 9) {
            if (kind == 0) {
                const int col = fwd_row_to_col(R);
                if (kc < C) v = w_x[((size_t)tap * Cw + kc) * NG + col];
                else if (kc >= Cs) v = w_h[((size_t)tap * F_ + (kc - Cs)) * NG + col];
            } else {
                const int ch = kc >> 2, gate = kc & 3;  // dY channel order: [channel][gate]
                const int col = gate * F_ + ch;
                const int ftap = 8 - tap;               // flipped tap
                const int ci = dgrad_row_to_channel(R);
                if (ci < C) v = w_x[((size_t)ftap * Cw + ci) * NG + col];
                else if (ci >= Cs && ci < Cs + F_) v = w_h[((size_t)ftap * F_ + (ci - Cs)) * NG + col];
            }
        }
        out[idx] = Tr<T>::from_f(v);
    }
}

// ---------------------------------------------------------------------------------------------
// shared pieces of the two convolution-shaped kernels
// ---------------------------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ void stage_tile(const void* in0, const void* in1, int c0, int c1, int H, int W, int b,
                                           int ty0, int tx0, char* lds_in, int pstride) {
    constexpr int CH = Tr<T>::CHUNK;
    const int n0 = c0 / CH, nch = n0 + c1 / CH;
    for (int idx = threadIdx.x; idx < NPX * nch; idx += kThreads) {
        const int px = idx / nch, ch = idx - px * nch;
        const int hy = px / HALO_W, hx = px - hy * HALO_W;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W && (ch < n0 || in1 != nullptr)) {  // (in1 null: all zero)
            const size_t pixel = ((size_t)b * H + gy) * W + gx;
            const T* src = ch < n0 ? static_cast<const T*>(in0) + pixel * c0 + ch * CH
                                   : static_cast<const T*>(in1) + pixel * c1 + (ch - n0) * CH;
            v = *reinterpret_cast<const uint4*>(src);
        }
        char* dst = lds_in + px * pstride + ch * 16;
        if constexpr (Tr<T>::PAD % 16 == 0) {
            *reinterpret_cast<uint4*>(dst) = v;
        } else {
            uint32_t* d = reinterpret_cast<uint32_t*>(dst);
            d[0] = v.x;
            d[1] = v.y;
            d[2] = v.z;
            d[3] = v.w;
        }
    }
}

// acc[t][pt] += sum over 9 taps and all channel groups.  Wave handles tile rows 2*wave, 2*wave+1.
template <typename T, int MT>
__device__ __forceinline__ void conv_mainloop(const char* lds_in, int pstride, const char* lds_w, const T* gw, int spt,
                                              int a_ss, int a_ts, f32x16 (&acc)[MT][2]) {
    using frag = typename Tr<T>::frag;
    constexpr int KG = Tr<T>::KG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, half = lane >> 5;
    const char* b_base = lds_in + ((2 * wave) * HALO_W + p) * pstride + half * KG * (int)sizeof(T);
    int s = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const char* b_tap = b_base + (dy * HALO_W + dx) * pstride;
        for (int q = 0; q < spt; ++q, ++s) {
            frag a[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if constexpr (Tr<T>::W_IN_LDS)
                    a[t] = *reinterpret_cast<const frag*>(lds_w + ((size_t)(s * a_ss + t * a_ts) * 64 + lane) * sizeof(frag));
                else
                    a[t] = *reinterpret_cast<const frag*>(gw + ((size_t)(s * a_ss + t * a_ts) * 64 + lane) * KG);
            }
            const frag b0 = *reinterpret_cast<const frag*>(b_tap + q * 2 * KG * (int)sizeof(T));
            const frag b1 = *reinterpret_cast<const frag*>(b_tap + HALO_W * pstride + q * 2 * KG * (int)sizeof(T));
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                acc[t][0] = Tr<T>::mma(a[t], b0, acc[t][0]);
                acc[t][1] = Tr<T>::mma(a[t], b1, acc[t][1]);
            }
        }
    }
}

// MODE 0: LSTM forward (rows = 64 permuted gate columns).  MODE 1: data gradient (rows = channels).
template <typename T, int MT, int MODE>
__global__ __launch_bounds__(kThreads) void conv3x3_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cpt = p.c0 + p.c1;
    const int pstride = cpt * (int)sizeof(T) + Tr<T>::PAD;
    const size_t wbytes = Tr<T>::W_IN_LDS ? (size_t)9 * p.spt * MT * 64 * sizeof(typename Tr<T>::frag) : 0;
    char* lds_w = smem;
    char* lds_in = smem + wbytes;
    float* lds_bias = reinterpret_cast<float*>(lds_in + ((NPX * pstride + 15) & ~15));

    if constexpr (Tr<T>::W_IN_LDS) {
        const uint4* src = static_cast<const uint4*>(p.wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds_w);
        for (size_t i = threadIdx.x; i < wbytes / 16; i += kThreads) dst[i] = src[i];
    }
    if (MODE == 0 && threadIdx.x < NG) lds_bias[threadIdx.x] = p.bias[threadIdx.x];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px_l = lane & 31, half = lane >> 5;
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.B;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int rem = tile - b * tiles_per_img;
        const int tyi = rem / p.tiles_x;
        const int ty0 = tyi * TH, tx0 = (rem - tyi * p.tiles_x) * TW;
        const bool live = p.durations == nullptr || p.t < p.durations[b];

        if (!live) {  // block-uniform branch: no barrier below is skipped by part of the block
            if (MODE == 0) lstm_copy_through_tile<T>(p, b, ty0, tx0, threadIdx.x, kThreads);
            else dgrad_zero_dx_tile<T>(p, b, ty0, tx0, threadIdx.x, kThreads);
            continue;
        }

        __syncthreads();  // previous tile's LDS reads are done (and the weight image is in place)
        stage_tile<T>(p.in0, p.in1, p.c0, p.c1, p.H, p.W, b, ty0, tx0, lds_in, pstride);
        __syncthreads();

        f32x16 acc[MT][2];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][pt][r] = 0.0f;
        conv_mainloop<T, MT>(lds_in, pstride, lds_w, static_cast<const T*>(p.wimg), p.spt, p.a_sstride, p.a_tstride, acc);

#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int gy = ty0 + 2 * wave + pt, gx = tx0 + px_l;
            if (gy >= p.H || gx >= p.W) continue;
            if constexpr (MODE == 0) {
                static_assert(MODE != 0 || MT == 2, "forward image has 64 rows");
                float z[32];
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            z[4 * (4 * t + a) + g] = acc[t][pt][4 * a + g] + lds_bias[g * F_ + 8 * half + 4 * t + a];
                lstm_finish_pixel<T, false>(p, b, gy, gx, half, z, p.hist_table);
            } else {
                const size_t pixel = ((size_t)b * p.H + gy) * p.W + gx;
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        float v[4] = {acc[t][pt][4 * a], acc[t][pt][4 * a + 1], acc[t][pt][4 * a + 2],
                                      acc[t][pt][4 * a + 3]};
                        dgrad_store4<T>(p, pixel, dgrad_frag_channel(p.row_base + 32 * t, a, half), v);
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pointwise gate backward
// ---------------------------------------------------------------------------------------------
struct PointwiseParams {
    const void* gates;     // [B,H,W,F,4] (g,i,f,o)
    const float* c_prev;   // [B,H,W,F]
    const float* c_cur;    // [B,H,W,F]
    const void* dh_above;  // grad of h as the layer's OUTPUT, sample stride dh_above_bstride, or null
    long long dh_above_bstride;
    int pool;              // > 1: dh_above is on the grid pooled by pool x pool windows (gradient of an average pool)
    int W;
    const void* dh_state;  // grad of h as carried STATE [B,H,W,F], or null
    const void* dc_state;  // [B,H,W,F] or null, float or bf16 (the kernel's DC type)
    void* dc_prev;         // [B,H,W,F], same type (may alias dc_state)
    void* dy;              // [B,H,W,F,4] pre-activation grads, (channel, gate) order
    const int* durations;
    int t, B, HW;
};

// CPT consecutive values as one vector load (CPT is 1 or 2; the address is CPT-element aligned)
template <int CPT>
__device__ __forceinline__ void load_run(const float* src, float (&v)[CPT]) {
    if constexpr (CPT == 2) {
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        const f32x2v x = *reinterpret_cast<const f32x2v*>(src);
        v[0] = x.x, v[1] = x.y;
    } else {
        v[0] = src[0];
    }
}
template <int CPT>
__device__ __forceinline__ void load_run(const __bf16* src, float (&v)[CPT]) {
    if constexpr (CPT == 2) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 x = *reinterpret_cast<const bf16x2*>(src);
        v[0] = (float)x[0], v[1] = (float)x[1];
    } else {
        v[0] = (float)src[0];
    }
}

// One thread takes the channels whose four gate values fill 16 bytes (2 for bf16, 1 for float), so
// that every load and store of a wave covers whole, contiguous cache lines: a narrower split makes
// the dy stores partial-sector writes, which cost twice the HBM write traffic.
template <typename T, typename DC>
__global__ __launch_bounds__(256) void lstm_pointwise_bwd_kernel(PointwiseParams p) {
    constexpr int CPT = sizeof(T) == 2 ? 2 : 1;  // channels per thread
    constexpr int TPP = F_ / CPT;                // threads per pixel
    const long long total = (long long)p.B * p.HW * TPP;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % TPP);
        const long long pixel = idx / TPP;
        const int b = (int)(pixel / p.HW);
        const size_t so = (size_t)pixel * F_ + q * CPT;
        const bool live = p.durations == nullptr || p.t < p.durations[b];
        float dc[CPT], gv[4 * CPT], out[4 * CPT], dcp[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) dc[k] = 0.0f;
        if (p.dc_state) load_run<CPT>(static_cast<const DC*>(p.dc_state) + so, dc);
        T* dy = static_cast<T*>(p.dy) + (size_t)pixel * NG + q * 4 * CPT;
        if (!live) {
#pragma unroll
            for (int k = 0; k < 4 * CPT; ++k) out[k] = 0.0f;
#pragma unroll
            for (int k = 0; k < CPT; ++k) dcp[k] = dc[k];
        } else {
            float dh[CPT], cpv[CPT], ccv[CPT];
            const T* above = static_cast<const T*>(p.dh_above);
            const T* state = static_cast<const T*>(p.dh_state);
            const int pix = (int)(pixel - (long long)b * p.HW);
            size_t ao = (size_t)b * p.dh_above_bstride + q * CPT;
            if (p.pool > 1) {
                const int y = pix / p.W, x = pix - y * p.W;
                ao += ((size_t)(y / p.pool) * (p.W / p.pool) + x / p.pool) * F_;
            } else {
                ao += (size_t)pix * F_;
            }
            float da[CPT], ds[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) da[k] = 0.0f, ds[k] = 0.0f;
            if (above) {
                load_run<CPT>(above + ao, da);
                if (p.pool > 1) {
                    const float inv = 1.0f / (float)(p.pool * p.pool);
#pragma unroll
                    for (int k = 0; k < CPT; ++k) da[k] *= inv;
                }
            }
            if (state) load_run<CPT>(state + so, ds);
            load_run<CPT>(p.c_prev + so, cpv);
            load_run<CPT>(p.c_cur + so, ccv);
#pragma unroll
            for (int k = 0; k < CPT; ++k) dh[k] = da[k] + ds[k];
            const T* gp = static_cast<const T*>(p.gates) + (size_t)pixel * NG + q * 4 * CPT;
            if constexpr (CPT == 2) {
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(gp);
#pragma unroll
                for (int k = 0; k < 8; ++k) gv[k] = (float)x[k];
            } else {
                float v[4];
                load4(gp, v);
#pragma unroll
                for (int k = 0; k < 4; ++k) gv[k] = v[k];
            }
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const float g = gv[4 * k], ig = gv[4 * k + 1], fg = gv[4 * k + 2], og = gv[4 * k + 3];
                const float tc = tanh_t<T>(ccv[k]);
                const float d_o = dh[k] * tc;
                const float dct = dc[k] + dh[k] * og * (1.0f - tc * tc);
                out[4 * k] = dct * ig * (1.0f - g * g);
                out[4 * k + 1] = dct * g * ig * (1.0f - ig);
                out[4 * k + 2] = dct * cpv[k] * fg * (1.0f - fg);
                out[4 * k + 3] = d_o * og * (1.0f - og);
                dcp[k] = dct * fg;
            }
        }
        DC* dcd = static_cast<DC*>(p.dc_prev) + so;
        if constexpr (CPT == 2) {
            store8(dy, out);
            if constexpr (sizeof(DC) == 4) {
                *reinterpret_cast<float2*>(dcd) = float2{dcp[0], dcp[1]};
            } else {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<bf16x2*>(dcd) = bf16x2{(__bf16)dcp[0], (__bf16)dcp[1]};
            }
        } else {
            store4(dy, reinterpret_cast<const float(&)[4]>(out));
            dcd[0] = (DC)dcp[0];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  slab[tap][cc][col] = sum_pixels XH[pixel + tap][cc] * dY[pixel][col]
// ---------------------------------------------------------------------------------------------

template <typename T, int MT>
__global__ __launch_bounds__(kThreads) void conv_wgrad_kernel(WgradParams p) {
    constexpr int NSLOT = MT == 2 ? 9 : 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cpt = p.c0 + p.c1;
    const int pstride = cpt * (int)sizeof(T) + Tr<T>::PAD;
    const int dstride = NG * (int)sizeof(T) + Tr<T>::PAD;
    char* lds_in = smem;
    char* lds_dy = smem + ((NPX * pstride + 15) & ~15);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    // work split: MT == 2 -> (mt, nt) = (wave >> 1, wave & 1), all 9 taps
    //             MT == 1 -> nt = wave & 1, taps 0..4 (wave < 2) or 5..8 (+ one idle slot)
    const int mt = MT == 2 ? (wave >> 1) : 0;
    const int nt = wave & 1;
    const int tap0 = MT == 2 ? 0 : (wave >> 1) * 5;
    const bool does_bias = MT == 2 ? (mt == 0) : (wave < 2);

    f32x16 acc[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float bsum = 0.0f;

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.B;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int rem = tile - b * tiles_per_img;
        const int tyi = rem / p.tiles_x;
        const int ty0 = tyi * TH, tx0 = (rem - tyi * p.tiles_x) * TW;
        if (p.durations != nullptr && p.t >= p.durations[b]) continue;  // dY is zero there

        __syncthreads();
        stage_tile<T>(p.in0, p.in1, p.c0, p.c1, p.H, p.W, b, ty0, tx0, lds_in, pstride);
        {
            constexpr int CH = Tr<T>::CHUNK;
            constexpr int nch = NG / CH;
            for (int idx = threadIdx.x; idx < TH * TW * nch; idx += kThreads) {
                const int px = idx / nch, ch = idx - px * nch;
                const int gy = ty0 + px / TW, gx = tx0 + (px & (TW - 1));
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (gy < p.H && gx < p.W)
                    v = *reinterpret_cast<const uint4*>(static_cast<const T*>(p.dy) +
                                                        (((size_t)b * p.H + gy) * p.W + gx) * NG + ch * CH);
                char* dst = lds_dy + px * dstride + ch * 16;
                if constexpr (Tr<T>::PAD % 16 == 0) {
                    *reinterpret_cast<uint4*>(dst) = v;
                } else {
                    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
                    d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
                }
            }
        }
        __syncthreads();

        if constexpr (sizeof(T) == 2) {
            // k-step = 16 consecutive pixels of one tile row; operands come out of the pixel-major
            // LDS images through the transposing read (4 pixels x 16 channels per 16-lane group).
            const int li = lane & 15, g16 = lane >> 4;
            const int q = li >> 2, pp = li & 3;
            const int csub = 16 * (g16 & 1) + 4 * pp;  // channel offset inside the 32-row tile
            for (int ks = 0; ks < 16; ++ks) {
                const int r = ks >> 1, x0 = (ks & 1) * 16;
                const int pj = x0 + 8 * half + q;  // this lane's pixel (first read; +4 for the second)
                bf16x8 bfrag;
                {
                    const char* a0 = lds_dy + (r * TW + pj) * dstride + (32 * nt + csub) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a0 + 4 * dstride));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bfrag = __builtin_bit_cast(bf16x8, both);
                }
                if (does_bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum += (float)bfrag[j];
                }
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) {
                    const int tap = tap0 + i;
                    if (tap < 9) {
                        const int dy = tap / 3, dx = tap - dy * 3;
                        const char* a0 = lds_in + ((r + dy) * HALO_W + pj + dx) * pstride + (32 * mt + csub) * 2;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(a0));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(a0 + 4 * pstride));
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), bfrag,
                                                                         acc[i], 0, 0, 0);
                    }
                }
            }
        } else {
            // f32: one pixel per lane half per MFMA; rows/cols are contiguous in the pixel-major images
            const int rc = lane & 31;
            for (int ks = 0; ks < TH * TW / 2; ++ks) {
                const int pix = 2 * ks + half;
                const int r = pix / TW, x = pix & (TW - 1);
                const float bv = *reinterpret_cast<const float*>(lds_dy + pix * dstride + (32 * nt + rc) * 4);
                if (does_bias) bsum += bv;
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) {
                    const int tap = tap0 + i;
                    if (tap < 9) {
                        const int dy = tap / 3, dx = tap - dy * 3;
                        const float av = *reinterpret_cast<const float*>(
                            lds_in + ((r + dy) * HALO_W + x + dx) * pstride + (32 * mt + rc) * 4);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }

    // write this workgroup's partial sums
    float* slab = p.slab + (size_t)blockIdx.x * 9 * (MT * 32) * NG;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int tap = tap0 + i;
        if (tap < 9) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int cc = 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                float* dst = slab + ((size_t)tap * (MT * 32) + cc) * NG + 32 * nt + (lane & 31);
                *dst = p.slab_accumulate ? *dst + acc[i][reg] : acc[i][reg];
            }
        }
    }
    if (does_bias) {
        bsum += __shfl_xor(bsum, 32);
        if (half == 0) {
            float* dst = p.bias_slab + (size_t)blockIdx.x * NG + 32 * nt + lane;
            *dst = p.slab_accumulate ? *dst + bsum : bsum;
        }
    }
}

// dw_x[tap][c][gate*F+ch] (+)= sum_blocks slab[blk][tap][c][ch*4+gate], same for dw_h and dbias.
// A workgroup owns 64 outputs; its 4 groups of 64 threads each sum a quarter of the slabs, LDS adds the four.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab,
                                                         const float* __restrict__ bias_slab, int nblk, int MT, int C,
                                                         int Cw, int Cs, int accumulate, float* __restrict__ dw_x,
                                                         float* __restrict__ dw_h, float* __restrict__ dbias) {
    __shared__ float red[4][64];
    const int n_x = 9 * C * NG, n_h = 9 * F_ * NG;
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;
    const size_t per_blk = (size_t)9 * (MT * 32) * NG;
    float sum = 0.0f;
    float* dst = nullptr;
    if (idx < n_x + n_h) {
        int tap, c, col;
        if (idx < n_x) {
            tap = idx / (C * NG);
            const int rem = idx - tap * C * NG;
            c = rem / NG;
            col = rem - c * NG;
            dst = dw_x + ((size_t)tap * Cw + c) * NG + col;
        } else {
            const int j = idx - n_x;
            tap = j / (F_ * NG);
            const int rem = j - tap * F_ * NG;
            c = Cs + rem / NG;
            col = rem % NG;
            dst = dw_h + j;
        }
        const int gate = col / F_, ch = col - gate * F_;
        const size_t off = ((size_t)tap * (MT * 32) + c) * NG + ch * 4 + gate;
        for (int b = sg; b < nblk; b += 4) sum += slab[b * per_blk + off];
    } else if (idx < n_x + n_h + NG) {
        const int col = idx - n_x - n_h;
        const int gate = col / F_, ch = col - gate * F_;
        for (int b = sg; b < nblk; b += 4) sum += bias_slab[(size_t)b * NG + ch * 4 + gate];
        dst = dbias + col;
    }
    red[sg][o] = sum;
    __syncthreads();
    if (sg == 0 && dst) {
        sum = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
        *dst = accumulate ? *dst + sum : sum;
    }
}

// ---------------------------------------------------------------------------------------------
// one-hot history channels: weight-tap table (forward) and their weight gradient (backward)
// ---------------------------------------------------------------------------------------------
// table[tap][i][ch][gate] = w_x[tap][C + i][gate * F + ch]
__global__ void hist_table_kernel(const float* __restrict__ w_x, int C, int hist, float* __restrict__ table) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 9 * hist * NG) return;
    const int k = idx % NG, i = (idx / NG) % hist, tap = idx / (NG * hist);
    const int ch = k >> 2, gate = k & 3;
    table[idx] = w_x[((size_t)tap * (C + hist) + C + i) * NG + gate * F_ + ch];
}

// dw_x[tap][C + i][:] (+)= sum over live samples of dY at the one pixel that tap reaches from the point
constexpr int kHistGroups = 16;  // sample groups per workgroup
constexpr int kHistChunk = 2048;  // samples located per pass
template <typename T>
__global__ __launch_bounds__(kHistGroups * NG) void hist_wgrad_kernel(const T* __restrict__ dy,
                                                                     const int* __restrict__ pts, long long pts_bstride,
                                                                     int hist, int B, int H, int W, int C,
                                                                     const int* __restrict__ durations, int t,
                                                                     int accumulate, float* __restrict__ dw_x) {
    __shared__ float red[kHistGroups][NG];
    __shared__ long long where[kHistChunk];  // element offset of the one pixel a sample contributes, or -1
    const int tap = blockIdx.x / hist, i = blockIdx.x - tap * hist;
    const int k = threadIdx.x & (NG - 1);  // dY channel, (channel, gate) order
    const int sg = threadIdx.x / NG;       // the samples are dealt to kHistGroups groups of threads
    const int ty = tap / 3, tx = tap - ty * 3;
    float sum = 0.0f;
    for (int b0 = 0; b0 < B; b0 += kHistChunk) {
        const int nb = min(kHistChunk, B - b0);
        // first where every sample's pixel is (coalesced reads of lengths and points, one sample per thread) ...
        __syncthreads();
        for (int j = threadIdx.x; j < nb; j += kHistGroups * NG) {
            const int b = b0 + j;
            const bool live = durations == nullptr || t < durations[b];
            const int px = pts[(size_t)b * pts_bstride + 2 * i], py = pts[(size_t)b * pts_bstride + 2 * i + 1];
            const int gy = py - ty + 1, gx = px - tx + 1;
            const bool hit = live && px >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            where[j] = hit ? (((long long)b * H + gy) * W + gx) * NG : -1;
        }
        __syncthreads();
        // ... then the gather itself: independent loads, eight in flight per thread
#pragma unroll 8
        for (int j = sg; j < nb; j += kHistGroups) {
            const long long at = where[j];
            const float v = (float)dy[(at >= 0 ? at : 0) + k];
            sum += at >= 0 ? v : 0.0f;
        }
    }
    red[sg][k] = sum;
    __syncthreads();
    if (sg == 0) {
        sum = 0.0f;
#pragma unroll
        for (int g = 0; g < kHistGroups; ++g) sum += red[g][k];
        float* dst = dw_x + ((size_t)tap * (C + hist) + C + i) * NG + (k & 3) * F_ + (k >> 2);
        *dst = accumulate ? *dst + sum : sum;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int g_num_cus = 0;
int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

template <typename K>
int set_lds(K kernel, size_t lds) {
    ISG_REQUIRE(lds <= 160 * 1024, ISG_EINVAL, "LDS request %zu exceeds 160 KiB", lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    return ISG_OK;
}

int check_channels(const char* who, int Cs, int dtype) {
    ISG_REQUIRE(dtype == ISG_F32 || dtype == ISG_BF16, ISG_EDTYPE, "%s: dtype %d", who, dtype);
    const int mult = dtype == ISG_BF16 ? 16 : 4;
    ISG_REQUIRE(Cs >= mult && Cs % mult == 0 && Cs <= 112, ISG_EINVAL,
                "%s: input channel stride %d must be a multiple of %d (pad with zero channels) and <= 112", who, Cs,
                mult);
    return ISG_OK;
}

template <typename T>
size_t conv_lds_bytes(int cpt, int spt, int MT, bool with_bias) {
    const size_t w = Tr<T>::W_IN_LDS ? (size_t)9 * spt * MT * 64 * sizeof(typename Tr<T>::frag) : 0;
    const size_t in = ((size_t)NPX * (cpt * sizeof(T) + Tr<T>::PAD) + 15) & ~(size_t)15;
    return w + in + (with_bias ? NG * sizeof(float) : 0);
}

template <typename T, int MT, int MODE>
int launch_conv(ConvParams& p, hipStream_t stream) {
    const int cpt = p.c0 + p.c1;
    const size_t lds = conv_lds_bytes<T>(cpt, p.spt, MT, MODE == 0);
    int rc = set_lds(&conv3x3_kernel<T, MT, MODE>, lds);
    if (rc != ISG_OK) return rc;
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int grid = ntiles < num_cus() * per_cu ? ntiles : num_cus() * per_cu;
    hipLaunchKernelGGL((conv3x3_kernel<T, MT, MODE>), dim3(grid), dim3(kThreads), lds, stream, p);
    return isg::check_launch("conv3x3_kernel");
}

}  // namespace

extern "C" long long isg_convlstm_packed_elems(int Cs, int F, int kind) {
    if (F != F_ || Cs <= 0) return -1;
    if (kind == 0) return (long long)9 * (Cs + F_) * NG;  // forward image: 64 rows x K
    const int MT = (Cs + F_ + 31) / 32;
    return (long long)9 * NG * MT * 32;  // dgrad image: MT*32 rows x (9 * 64)
}

extern "C" int isg_convlstm_pack_weights(const float* w_x, const float* w_h, int C, int Cs, int hist, int F, int dtype,
                                         void* packed_fwd, void* packed_dgrad, float* hist_table, void* stream) {
    ISG_REQUIRE(F == F_, ISG_EINVAL, "isg_convlstm: filters must be %d (got %d)", F_, F);
    ISG_REQUIRE(w_x && w_h && C >= 1 && C <= Cs && hist >= 0 && (hist == 0 || hist_table), ISG_EINVAL,
                "isg_convlstm_pack_weights: bad arguments");
    int rc = check_channels("isg_convlstm_pack_weights", Cs, dtype);
    if (rc != ISG_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int MT = (Cs + F_ + 31) / 32;
    ISG_REQUIRE(MT <= 4, ISG_EINVAL, "isg_convlstm: too many input channels");
    for (int kind = 0; kind < 2; ++kind) {
        void* out = kind == 0 ? packed_fwd : packed_dgrad;
        if (!out) continue;
        const long long total = isg_convlstm_packed_elems(Cs, F, kind);
        const int mt = kind == 0 ? 2 : MT;
        const int blocks = (int)((total + 255) / 256);
        if (dtype == ISG_F32)
            hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, s, w_x, w_h, C, C + hist, Cs, kind,
                               mt, static_cast<float*>(out), total);
        else
            hipLaunchKernelGGL(pack_weights_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, w_x, w_h, C, C + hist, Cs,
                               kind, mt, static_cast<__bf16*>(out), total);
        rc = isg::check_launch("pack_weights_kernel");
        if (rc != ISG_OK) return rc;
    }
    if (hist > 0) {
        hipLaunchKernelGGL(hist_table_kernel, dim3((9 * hist * NG + 255) / 256), dim3(256), 0, s, w_x, C, hist,
                           hist_table);
        rc = isg::check_launch("hist_table_kernel");
    }
    return rc;
}

namespace {
// the fused 8 x 8 pooled output exists in the bf16 ping-pong kernels only, on grids they tile exactly
bool pooled_fwd_supported(int Cs, int F, int dtype, int B, int H, int W, int pool, int hist) {
    return F == F_ && dtype == ISG_BF16 && (Cs == 16 || Cs == 32) && pool == 8 && H % TH == 0 && W % TW == 0 &&
           hist >= 0 && hist <= 2 && (long long)B * H * W * Cs * 2 < (1LL << 32);
}

// the cell state stored as bf16: only the bf16 ping-pong kernels read and write it that way
bool state_bf16_supported(int Cs, int F, int dtype, int B, int H, int W) {
    return F == F_ && dtype == ISG_BF16 && (Cs == 16 || Cs == 32 || Cs == 48) && B >= 1 && H >= 1 && W >= 1 &&
           (long long)B * H * W * (Cs > F_ ? Cs : F_) * 2 < (1LL << 32);
}

int cell_fwd_impl(const void* x, const void* h_prev, const void* c_prev_v, const void* packed_fwd, const float* bias,
                  int B, int H, int W, int Cs, int F, int dtype, const int32_t* durations, int t,
                  const int32_t* hist_points, int hist, long long hist_batch_stride, const float* hist_table,
                  void* h_out, void* c_out_v, void* out, long long out_batch_stride, void* gates, int save_layout,
                  void* pool_out, long long pool_batch_stride, void* stream, int h_pixel_stride = F_,
                  int out_pixel_stride = F_) {
    const bool state_bf16 = (save_layout & ISG_STATE_BF16) != 0;
    save_layout &= ~ISG_STATE_BF16;
    // (typed as float in the kernels' parameter block; the bf16 instantiations cast it back)
    const float* c_prev = static_cast<const float*>(c_prev_v);
    float* c_out = static_cast<float*>(c_out_v);
    ISG_REQUIRE(F == F_, ISG_EINVAL, "isg_convlstm: filters must be %d (got %d)", F_, F);
    ISG_REQUIRE(!state_bf16 || state_bf16_supported(Cs, F, dtype, B, H, W), ISG_EINVAL,
                "isg_convlstm_cell_fwd: ISG_STATE_BF16 needs bf16 activations and Cs 16, 32 or 48 (got Cs=%d dtype=%d): "
                "ask isg_convlstm_state_bf16_supported first", Cs, dtype);
    ISG_REQUIRE(x && packed_fwd && bias && h_out && c_out, ISG_EINVAL, "isg_convlstm_cell_fwd: null pointer");
    ISG_REQUIRE((h_prev == nullptr) == (c_prev == nullptr), ISG_EINVAL,
                "isg_convlstm_cell_fwd: h_prev and c_prev are both given or both NULL (the zero state)");
    ISG_REQUIRE(B >= 1 && H >= 1 && W >= 1, ISG_EINVAL, "isg_convlstm_cell_fwd: bad shape");
    ISG_REQUIRE(h_pixel_stride >= F_ && h_pixel_stride % 8 == 0 && out_pixel_stride >= F_ && out_pixel_stride % 8 == 0 &&
                    (long long)B * H * W < (1LL << 31),
                ISG_EINVAL, "isg_convlstm_cell_fwd_strided: pixel strides are multiples of 8 elements and at least %d "
                "(got %d and %d)", F_, h_pixel_stride, out_pixel_stride);
    ISG_REQUIRE(hist >= 0 && (hist == 0 || (hist_points && hist_table)), ISG_EINVAL,
                "isg_convlstm_cell_fwd: history points and table are needed when hist > 0");
    ISG_REQUIRE(save_layout == ISG_SAVE_GATES || (save_layout == ISG_SAVE_FACTORS && (dtype == ISG_BF16 || !gates)),
                ISG_EINVAL, "isg_convlstm_cell_fwd: save_layout %d (the backward factors exist in bf16 only)", save_layout);
    int rc = check_channels("isg_convlstm_cell_fwd", Cs, dtype);
    if (rc != ISG_OK) return rc;
    ConvParams p = {};
    p.save_layout = save_layout;
    p.in0 = x, p.in1 = h_prev, p.c0 = Cs, p.c1 = F_;
    p.B = B, p.H = H, p.W = W;
    p.wimg = packed_fwd;
    p.a_sstride = 2, p.a_tstride = 1;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + TH - 1) / TH;
    p.durations = durations, p.t = t;
    p.bias = bias, p.c_prev = c_prev, p.c_out = c_out, p.h_out = h_out, p.out = out;
    p.state_bf16 = state_bf16 ? 1 : 0;
    p.out_bstride = out_batch_stride, p.gates = gates;
    p.h_pstride = h_pixel_stride, p.out_pstride = out_pixel_stride;
    p.pool_out = pool_out, p.pool_bstride = pool_batch_stride;
    p.hist_points = hist_points, p.hist = hist, p.hist_bstride = hist_batch_stride, p.hist_table = hist_table;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32) {
        p.spt = (Cs + F_) / 2;
        return launch_conv<float, 2, 0>(p, s);
    }
    p.spt = (Cs + F_) / 16;
    rc = launch_conv_pp(p, 2, 0, num_cus(), s);
    if (rc != kNotSupported) return rc;
    ISG_REQUIRE(!state_bf16, ISG_EINVAL, "isg_convlstm_cell_fwd: no kernel with a bf16 cell state for this shape");
    ISG_REQUIRE(pool_out == nullptr, ISG_EINVAL, "isg_convlstm_cell_fwd_pooled: no kernel with a pooled output for this shape");
    return launch_conv<__bf16, 2, 0>(p, s);
}
}  // namespace

extern "C" int isg_convlstm_state_bf16_supported(int Cs, int F, int dtype, int B, int H, int W) {
    return state_bf16_supported(Cs, F, dtype, B, H, W) ? 1 : 0;
}

extern "C" int isg_convlstm_cell_fwd(const void* x, const void* h_prev, const void* c_prev, const void* packed_fwd,
                                     const float* bias, int B, int H, int W, int Cs, int F, int dtype,
                                     const int32_t* durations, int t, const int32_t* hist_points, int hist,
                                     long long hist_batch_stride, const float* hist_table, void* h_out, void* c_out,
                                     void* out, long long out_batch_stride, void* gates, int save_layout,
                                     void* stream) {
    return cell_fwd_impl(x, h_prev, c_prev, packed_fwd, bias, B, H, W, Cs, F, dtype, durations, t, hist_points, hist,
                         hist_batch_stride, hist_table, h_out, c_out, out, out_batch_stride, gates, save_layout, nullptr,
                         0, stream);
}

extern "C" int isg_convlstm_cell_fwd_strided(const void* x, const void* h_prev, const void* c_prev,
                                             const void* packed_fwd, const float* bias, int B, int H, int W, int Cs,
                                             int F, int dtype, const int32_t* durations, int t,
                                             const int32_t* hist_points, int hist, long long hist_batch_stride,
                                             const float* hist_table, void* h_out, int h_pixel_stride, void* c_out,
                                             void* out, long long out_batch_stride, int out_pixel_stride, void* gates,
                                             int save_layout, void* stream) {
    return cell_fwd_impl(x, h_prev, c_prev, packed_fwd, bias, B, H, W, Cs, F, dtype, durations, t, hist_points, hist,
                         hist_batch_stride, hist_table, h_out, c_out, out, out_batch_stride, gates, save_layout, nullptr,
                         0, stream, h_pixel_stride, out_pixel_stride);
}

extern "C" int isg_convlstm_pooled_fwd_supported(int Cs, int F, int dtype, int B, int H, int W, int pool, int hist) {
    return pooled_fwd_supported(Cs, F, dtype, B, H, W, pool, hist) ? 1 : 0;
}

extern "C" int isg_convlstm_cell_fwd_pooled(const void* x, const void* h_prev, const void* c_prev,
                                            const void* packed_fwd, const float* bias, int B, int H, int W, int Cs,
                                            int F, int dtype, const int32_t* durations, int t,
                                            const int32_t* hist_points, int hist, long long hist_batch_stride,
                                            const float* hist_table, void* h_out, void* c_out, void* gates,
                                            int save_layout, int pool, void* pool_out, long long pool_batch_stride,
                                            void* stream) {
    ISG_REQUIRE(pool_out, ISG_EINVAL, "isg_convlstm_cell_fwd_pooled: null pooled output");
    ISG_REQUIRE(pooled_fwd_supported(Cs, F, dtype, B, H, W, pool, hist), ISG_EINVAL,
                "isg_convlstm_cell_fwd_pooled: needs bf16, Cs 16 or 32, pool 8, H %% 8 == 0 and W %% 32 == 0 "
                "(got Cs=%d dtype=%d pool=%d %dx%d): ask isg_convlstm_pooled_fwd_supported first",
                Cs, dtype, pool, H, W);
    return cell_fwd_impl(x, h_prev, c_prev, packed_fwd, bias, B, H, W, Cs, F, dtype, durations, t, hist_points, hist,
                         hist_batch_stride, hist_table, h_out, c_out, nullptr, 0, gates, save_layout, pool_out,
                         pool_batch_stride, stream);
}

extern "C" int isg_convlstm_pointwise_bwd(const void* gates, const float* c_prev, const float* c_cur,
                                          const void* dh_above, long long dh_above_batch_stride, int dh_above_pool,
                                          const void* dh_state, const void* dc_state, int dc_dtype, int B, int H,
                                          int W, int F, int dtype, const int32_t* durations, int t, void* dy,
                                          void* dc_prev, void* stream) {
    ISG_REQUIRE(F == F_, ISG_EINVAL, "isg_convlstm: filters must be %d (got %d)", F_, F);
    ISG_REQUIRE(gates && c_prev && c_cur && dy && dc_prev, ISG_EINVAL, "isg_convlstm_pointwise_bwd: null pointer");
    ISG_REQUIRE(dtype == ISG_F32 || dtype == ISG_BF16, ISG_EDTYPE, "isg_convlstm_pointwise_bwd: dtype %d", dtype);
    ISG_REQUIRE(dc_dtype == ISG_F32 || dc_dtype == ISG_BF16, ISG_EDTYPE, "isg_convlstm_pointwise_bwd: dc_dtype %d",
                dc_dtype);
    PointwiseParams p = {};
    p.gates = gates, p.c_prev = c_prev, p.c_cur = c_cur;
    ISG_REQUIRE(dh_above_pool >= 1 && H % dh_above_pool == 0 && W % dh_above_pool == 0, ISG_EINVAL,
                "isg_convlstm_pointwise_bwd: pool %d must divide the %dx%d grid", dh_above_pool, H, W);
    p.dh_above = dh_above, p.dh_above_bstride = dh_above_batch_stride, p.pool = dh_above_pool, p.W = W;
    p.dh_state = dh_state, p.dc_state = dc_state, p.dc_prev = dc_prev, p.dy = dy;
    p.durations = durations, p.t = t, p.B = B, p.HW = H * W;
    const long long total = (long long)B * H * W * (dtype == ISG_F32 ? F_ : F_ / 2);
    // grid-stride beyond 65,536 workgroups (measured at 100 k workgroups' worth of pixels: 16 k workgroups 3.5 %
    // slower, one workgroup per 256 items 1 % slower; non-temporal loads and stores 7 % slower)
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32 && dc_dtype == ISG_F32)
        hipLaunchKernelGGL((lstm_pointwise_bwd_kernel<float, float>), dim3(blocks), dim3(256), 0, s, p);
    else if (dtype == ISG_F32)
        hipLaunchKernelGGL((lstm_pointwise_bwd_kernel<float, __bf16>), dim3(blocks), dim3(256), 0, s, p);
    else if (dc_dtype == ISG_F32)
        hipLaunchKernelGGL((lstm_pointwise_bwd_kernel<__bf16, float>), dim3(blocks), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((lstm_pointwise_bwd_kernel<__bf16, __bf16>), dim3(blocks), dim3(256), 0, s, p);
    return isg::check_launch("lstm_pointwise_bwd_kernel");
}

extern "C" int isg_convlstm_dgrad(const void* dy, const void* packed_dgrad, int B, int H, int W, int Cs, int F,
                                  int dtype, const int32_t* durations, int t, void* dx, int dx_channels,
                                  int dx_channel_stride, int dx_accumulate, void* dh_prev, void* stream) {
    ISG_REQUIRE(F == F_, ISG_EINVAL, "isg_convlstm: filters must be %d (got %d)", F_, F);
    ISG_REQUIRE(dy && packed_dgrad && dh_prev, ISG_EINVAL, "isg_convlstm_dgrad: null pointer");
    ISG_REQUIRE(B >= 1 && H >= 1 && W >= 1, ISG_EINVAL, "isg_convlstm_dgrad: bad shape");
    int rc = check_channels("isg_convlstm_dgrad", Cs, dtype);
    if (rc != ISG_OK) return rc;
    ISG_REQUIRE(!dx || (dx_channels >= 4 && dx_channels % 4 == 0 && dx_channels <= Cs &&
                        dx_channel_stride % 4 == 0 && dx_channel_stride >= dx_channels),
                ISG_EINVAL, "isg_convlstm_dgrad: dx channels/stride must be multiples of 4 within the input width");
    ConvParams p = {};
    p.in0 = dy, p.in1 = nullptr, p.c0 = NG, p.c1 = 0;
    p.B = B, p.H = H, p.W = W;
    p.wimg = packed_dgrad;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + TH - 1) / TH;
    p.durations = durations, p.t = t;
    p.dx = dx, p.dx_channels = dx ? dx_channels : 0, p.dx_cstride = dx_channel_stride, p.dx_accumulate = dx_accumulate;
    p.dh_prev = dh_prev;
    p.row_split = Cs;
    p.row_base = 0;
    p.spt = dtype == ISG_F32 ? NG / 2 : NG / 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int MT = (Cs + F_ + 31) / 32;
    p.a_sstride = 1, p.a_tstride = 9 * p.spt;  // row-tile-major image
    if (dtype == ISG_BF16) {
        // fast paths: all row tiles in one ping-pong launch on 4-row pixel tiles (two row tiles), else one
        // launch per 32-row tile of the image (each re-stages dY)
        rc = launch_conv_pp(p, MT, 1, num_cus(), s);
        if (rc != kNotSupported) return rc;
        bool fast = true;
        for (int tile = 0; tile < MT && fast; ++tile) {
            ConvParams q = p;
            q.wimg = static_cast<const char*>(packed_dgrad) + (size_t)tile * 9 * p.spt * 1024;
            q.row_base = 32 * tile;
            rc = launch_conv_pp(q, 1, 1, num_cus(), s);
            if (rc == kNotSupported) fast = false;
            else if (rc != ISG_OK) return rc;
        }
        if (fast) return ISG_OK;
    }
    if (dtype == ISG_F32) {
        if (MT == 1) return launch_conv<float, 1, 1>(p, s);
        if (MT == 2) return launch_conv<float, 2, 1>(p, s);
        if (MT == 3) return launch_conv<float, 3, 1>(p, s);
        return launch_conv<float, 4, 1>(p, s);
    }
    if (MT == 1) return launch_conv<__bf16, 1, 1>(p, s);
    if (MT == 2) return launch_conv<__bf16, 2, 1>(p, s);
    if (MT == 3) return launch_conv<__bf16, 3, 1>(p, s);
    return launch_conv<__bf16, 4, 1>(p, s);
}

extern "C" long long isg_convlstm_wgrad_workspace_bytes(int Cs, int F) {
    if (F != F_ || Cs <= 0 || Cs + F_ > 64) return -1;
    const int MT = (Cs + F_ + 31) / 32;
    return (long long)kWgradMaxBlocks * ((long long)9 * MT * 32 * NG + NG) * (long long)sizeof(float);
}

extern "C" int isg_convlstm_wgrad(const void* x, const void* h_prev, const void* dy, int B, int H, int W, int C,
                                  int Cs, int F, int dtype, const int32_t* durations, int t,
                                  const int32_t* hist_points, int hist, long long hist_batch_stride, void* workspace,
                                  int mode, float* dw_x, float* dw_h, float* dbias, void* stream) {
    ISG_REQUIRE(F == F_, ISG_EINVAL, "isg_convlstm: filters must be %d (got %d)", F_, F);
    ISG_REQUIRE(x && h_prev && dy && workspace && dw_x && dw_h && dbias, ISG_EINVAL,
                "isg_convlstm_wgrad: null pointer");
    ISG_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1 && C <= Cs, ISG_EINVAL, "isg_convlstm_wgrad: bad shape");
    ISG_REQUIRE(mode >= ISG_WGRAD_ASSIGN && mode <= ISG_WGRAD_DEFER, ISG_EINVAL, "isg_convlstm_wgrad: bad mode %d",
                mode);
    const bool deferred = mode == ISG_WGRAD_DEFER_FIRST || mode == ISG_WGRAD_DEFER;
    const int accumulate = mode == ISG_WGRAD_ADD || mode == ISG_WGRAD_DEFER;
    int rc = check_channels("isg_convlstm_wgrad", Cs, dtype);
    if (rc != ISG_OK) return rc;
    ISG_REQUIRE(Cs + F_ <= 64, ISG_EINVAL, "isg_convlstm_wgrad: at most %d input channels (got %d)", 64 - F_, Cs);
    const int MT = (Cs + F_ + 31) / 32;
    WgradParams p = {};
    p.in0 = x, p.in1 = h_prev, p.c0 = Cs, p.c1 = F_, p.dy = dy;
    p.B = B, p.H = H, p.W = W;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + TH - 1) / TH;
    p.durations = durations, p.t = t;
    const int ntiles = p.tiles_x * p.tiles_y * B;
    const int grid = ntiles < kWgradMaxBlocks ? ntiles : kWgradMaxBlocks;
    p.slab = static_cast<float*>(workspace);
    p.bias_slab = p.slab + (size_t)kWgradMaxBlocks * 9 * MT * 32 * NG;
    p.slab_accumulate = mode == ISG_WGRAD_DEFER;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t esz = dtype == ISG_F32 ? 4 : 2, pad = dtype == ISG_F32 ? 4 : 16;
    const size_t lds = (((size_t)NPX * ((Cs + F_) * esz + pad) + 15) & ~(size_t)15) + (size_t)TH * TW * (NG * esz + pad);
    bool launched = false;
    if (dtype == ISG_BF16) {
        rc = launch_wgrad_pp(p, MT, grid, s);
        if (rc != kNotSupported) {
            if (rc != ISG_OK) return rc;
            launched = true;
        }
    }
    if (launched) {
    } else if (dtype == ISG_F32) {
        if (MT == 1) {
            if ((rc = set_lds(&conv_wgrad_kernel<float, 1>, lds)) != ISG_OK) return rc;
            hipLaunchKernelGGL((conv_wgrad_kernel<float, 1>), dim3(grid), dim3(kThreads), lds, s, p);
        } else {
            if ((rc = set_lds(&conv_wgrad_kernel<float, 2>, lds)) != ISG_OK) return rc;
            hipLaunchKernelGGL((conv_wgrad_kernel<float, 2>), dim3(grid), dim3(kThreads), lds, s, p);
        }
    } else {
        if (MT == 1) {
            if ((rc = set_lds(&conv_wgrad_kernel<__bf16, 1>, lds)) != ISG_OK) return rc;
            hipLaunchKernelGGL((conv_wgrad_kernel<__bf16, 1>), dim3(grid), dim3(kThreads), lds, s, p);
        } else {
            if ((rc = set_lds(&conv_wgrad_kernel<__bf16, 2>, lds)) != ISG_OK) return rc;
            hipLaunchKernelGGL((conv_wgrad_kernel<__bf16, 2>), dim3(grid), dim3(kThreads), lds, s, p);
        }
    }
    if ((rc = isg::check_launch("conv_wgrad_kernel")) != ISG_OK) return rc;
    if (!deferred) {
        const int n_out = 9 * C * NG + 9 * F_ * NG + NG;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n_out + 63) / 64), dim3(256), 0, s, p.slab, p.bias_slab, grid,
                           MT, C, C + hist, Cs, accumulate, dw_x, dw_h, dbias);
        if ((rc = isg::check_launch("wgrad_reduce_kernel")) != ISG_OK) return rc;
    }
    if (hist > 0) {
        ISG_REQUIRE(hist_points, ISG_EINVAL, "isg_convlstm_wgrad: history points are needed when hist > 0");
        if (dtype == ISG_F32)
            hipLaunchKernelGGL(hist_wgrad_kernel<float>, dim3(9 * hist), dim3(kHistGroups * NG), 0, s, static_cast<const float*>(dy),
                               hist_points, hist_batch_stride, hist, B, H, W, C, durations, t, accumulate, dw_x);
        else
            hipLaunchKernelGGL(hist_wgrad_kernel<__bf16>, dim3(9 * hist), dim3(kHistGroups * NG), 0, s,
                               static_cast<const __bf16*>(dy), hist_points, hist_batch_stride, hist, B, H, W, C,
                               durations, t, accumulate, dw_x);
        rc = isg::check_launch("hist_wgrad_kernel");
    }
    return rc;
}

extern "C" int isg_convlstm_wgrad_finish(const void* workspace, int B, int H, int W, int C, int Cs, int hist, int F,
                                         int accumulate, float* dw_x, float* dw_h, float* dbias, void* stream) {
    ISG_REQUIRE(F == F_, ISG_EINVAL, "isg_convlstm: filters must be %d (got %d)", F_, F);
    ISG_REQUIRE(workspace && dw_x && dw_h && dbias, ISG_EINVAL, "isg_convlstm_wgrad_finish: null pointer");
    ISG_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1 && C <= Cs && hist >= 0 && Cs + F_ <= 64, ISG_EINVAL,
                "isg_convlstm_wgrad_finish: bad shape");
    const int MT = (Cs + F_ + 31) / 32;
    const int ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * B;
    const int grid = ntiles < kWgradMaxBlocks ? ntiles : kWgradMaxBlocks;
    const float* slab = static_cast<const float*>(workspace);
    const float* bias_slab = slab + (size_t)kWgradMaxBlocks * 9 * MT * 32 * NG;
    const int n_out = 9 * C * NG + 9 * F_ * NG + NG;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n_out + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream),
                       slab, bias_slab, grid, MT, C, C + hist, Cs, accumulate, dw_x, dw_h, dbias);
    return isg::check_launch("wgrad_reduce_kernel");
}
