// Shared declarations of the ConvLSTM kernels (isg_convlstm.hip: generic kernels + host API;
// isg_convlstm_pp.hip: the bf16 ping-pong fast path).
#pragma once
#include <hip/hip_bf16.h>

#include "isg_common.h"

namespace isgc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TH = 8, TW = 32;             // output tile
constexpr int HALO_W = TW + 2, HALO_H = TH + 2;
constexpr int NPX = HALO_W * HALO_H;       // staged input pixels per tile
constexpr int F_ = 16;                     // hidden channels (every runnable reference net)
constexpr int NG = 4 * F_;                 // gate columns
constexpr int kWgradMaxBlocks = 256;       // partial-sum slabs per weight-gradient launch

template <typename T>
struct Tr;

template <>
struct Tr<float> {
    static constexpr int KG = 1;      // k values a lane feeds per MFMA
    static constexpr int CHUNK = 4;   // elements per 16-byte staging chunk
    static constexpr int PAD = 4;     // bytes added to an LDS pixel row (bank spread)
    static constexpr int DTYPE = ISG_F32;
    static constexpr bool W_IN_LDS = false;
    using frag = float;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
    static __device__ __forceinline__ float ex(float v) { return expf(v); }
};

template <>
struct Tr<__bf16> {
    static constexpr int KG = 8;
    static constexpr int CHUNK = 8;
    static constexpr int PAD = 16;
    static constexpr int DTYPE = ISG_BF16;
    static constexpr bool W_IN_LDS = true;
    using frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
    static __device__ __forceinline__ __bf16 from_f(float v) { return (__bf16)v; }
    static __device__ __forceinline__ float ex(float v) { return __expf(v); }
};

template <typename T>
__device__ __forceinline__ float sigmoid_t(float z) {
    return 1.0f / (1.0f + Tr<T>::ex(-z));
}
template <typename T>
__device__ __forceinline__ float tanh_t(float z) {
    // 1 - 2 / (exp(2z) + 1); saturates cleanly at +-1 for large |z|
    return 1.0f - 2.0f / (Tr<T>::ex(2.0f * z) + 1.0f);
}

// Reference gate column (j|i|f|o blocks of F) held by row R of the forward weight image.
__host__ __device__ inline int fwd_row_to_col(int R) {
    const int t = R >> 5, r = R & 31;
    const int gate = r & 3, hh = (r >> 2) & 1, a = r >> 3;
    return gate * F_ + 8 * hh + 4 * t + a;
}

// Input channel of the [x | h_prev] concatenation held by row R of the data-gradient weight image.  The first 32-row
// tile is stored permuted: the 16 accumulator registers of an MFMA lane (rows 8 a + 4 half + i, a = 0..3, i = 0..3)
// are then the 16 CONSECUTIVE channels 16 half + 4 a + i, which an epilogue stores as two 16-byte pieces instead of
// four 8-byte ones.  Later row tiles keep the identity (the fused backward reads the second one through a 16-row
// MFMA whose lanes already hold consecutive channels).
__host__ __device__ inline int dgrad_row_to_channel(int R) {
    return R < 32 ? 16 * ((R >> 2) & 1) + 4 * (R >> 3) + (R & 3) : R;
}
// First of the four consecutive channels in accumulator registers 4 a .. 4 a + 3 of a lane in half `half` of the
// wave, for the row tile that starts at image row `row0` (a multiple of 32).
__device__ __forceinline__ int dgrad_frag_channel(int row0, int a, int half) {
    return row0 == 0 ? 16 * half + 4 * a : row0 + 8 * a + 4 * half;
}

struct ConvParams {
    const void* in0;  // [B,H,W,c0]
    const void* in1;  // [B,H,W,c1] or null (c1 = 0)
    int c0, c1;
    int B, H, W;
    const void* wimg;  // packed image
    int spt;           // k-steps per tap
    int tiles_x, tiles_y;
    const int* durations;  // [B] or null
    int t;                 // current time step (compared with durations)
    // LSTM epilogue
    const float* bias;     // [4F] reference order
    const float* c_prev;   // [B,H,W,F]
    float* c_out;          // [B,H,W,F]
    int state_bf16;        // c_prev / c_out hold bf16 instead of float (bf16 ping-pong kernels only)
    void* h_out;           // [B,H,W,F]; pixel stride h_pstride elements (F = dense; wider: a channel slice of a
                           // joint buffer, e.g. the next step's [x | h] input)
    void* out;             // masked output or null; sample stride out_bstride elements, pixel stride out_pstride
    long long out_bstride;
    int h_pstride, out_pstride;
    void* gates;           // [B,H,W,F,4] (g,i,f,o), or the backward factors [B,H,W,F/2,4,2]; or null
    int save_layout;       // ISG_SAVE_GATES / ISG_SAVE_FACTORS: what `gates` receives
    // pooled output of the ping-pong kernels: means of the masked output over 8 x 8 windows,
    // [H/8][W/8][F] per sample with sample stride pool_bstride elements; null = none
    void* pool_out;
    long long pool_bstride;
    // dgrad epilogue
    void* dx;              // [B,H,W,dx_cstride] or null
    int dx_channels, dx_cstride, dx_accumulate;
    void* dh_prev;         // [B,H,W,F]; left untouched for a masked sample (state gradient passes through)
    int row_split;         // rows [0, row_split) -> dx, [row_split, row_split + F) -> dh_prev
    int row_base;          // input channel of image row 0 (a launch may cover one 32-row tile of the image)
    int a_sstride, a_tstride;  // weight image: fragment block index = s * a_sstride + t * a_tstride
    // one-hot history inputs applied as a 9-tap weight scatter in the LSTM epilogue (not in the GEMM)
    const int* hist_points;    // [B][hist][2] (x, y) of this step, sample stride hist_bstride ints; x < 0 = empty
    int hist;
    long long hist_bstride;
    const float* hist_table;   // [9 taps][hist][F][4 gates] float
};

template <typename T>
__device__ __forceinline__ void store4(T* dst, const float (&v)[4]) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
        bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<bf16x4*>(dst) = o;
    }
}
template <typename T>
__device__ __forceinline__ void load4(const T* src, float (&v)[4]) {
    if constexpr (sizeof(T) == 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(src);
        v[0] = x[0], v[1] = x[1], v[2] = x[2], v[3] = x[3];
    } else {
        const bf16x4 x = *reinterpret_cast<const bf16x4*>(src);
        v[0] = (float)x[0], v[1] = (float)x[1], v[2] = (float)x[2], v[3] = (float)x[3];
    }
}


// ---- LSTM epilogue of one pixel, shared by the generic and the ping-pong kernels -------------
// z[4 * cl + gate]: pre-activations of the lane's 8 channels (cl) in (j, i, f, o) order, bias added.

// Adds the one-hot history inputs: a frame whose point sits at (px, py) contributes weight tap
// (py - gy + 1, px - gx + 1) to the pixel (gx, gy) when that tap exists.
__device__ __forceinline__ void add_history_taps(float (&z)[32], const int* pts, int hist, const float* table, int gx,
                                                 int gy, int half) {
    // the points of a sample are wave-uniform and read-only: constant address space = scalar loads
    const __attribute__((address_space(4))) int* cpts =
        (const __attribute__((address_space(4))) int*)(unsigned long long)(pts);
    for (int i = 0; i < hist; ++i) {
        const int px = cpts[2 * i], py = cpts[2 * i + 1];
        const int ty = py - gy + 1, tx = px - gx + 1;
        if (px >= 0 && (unsigned)ty < 3u && (unsigned)tx < 3u) {
            const float* tb = table + ((ty * 3 + tx) * hist + i) * NG + 32 * half;
#pragma unroll
            for (int k = 0; k < 32; ++k) z[k] += tb[k];
        }
    }
}

template <bool FAST>
__device__ __forceinline__ float act_sigmoid(float z) {
#ifdef ISG_DIAG_NOACT  // diagnostic build: what the launch takes without the transcendental instructions
    if constexpr (FAST) return 0.25f * z + 0.5f;
#endif
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * z));
    else return 1.0f / (1.0f + expf(-z));
}
template <bool FAST>
__device__ __forceinline__ float act_tanh(float z) {
#ifdef ISG_DIAG_NOACT
    if constexpr (FAST) return 0.5f * z;
#endif
    if constexpr (FAST) return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.88539008f * z) + 1.0f), 1.0f);
    else return 1.0f - 2.0f / (expf(2.0f * z) + 1.0f);
}

// The gate backward of one channel is linear in the incoming gradients (dh, dc):
//   dct = dc + dh E;   dY = (dct A, dct B, dct C, dh D) for the gates (j, i, f, o);   dc_prev = dct f
// with A = i (1 - j^2), B = j i (1 - i), C = c_prev f (1 - f), D = tanh(c') o (1 - o) = h' (1 - o),
// E = o (1 - tanh(c')^2) = o - h'^2 / o (the derivative of reference convLSTM.py:40-43).  With c_prev and h' at
// hand - the forward pass stores both anyway - everything follows from (A, B, f, o) without a hyperbolic tangent
// and without the new cell state.  The forward kernels can save these four instead of the gate activations
// (ISG_SAVE_FACTORS), in the same 128 bytes per pixel: per channel pair A0 A1 B0 B1 f0 f1 o0 o1 (a 32-bit word
// holds one value of both channels; the kernels finish channels in pairs).
__device__ __forceinline__ void lstm_backward_factors(float g, float ig, float& a, float& b) {
    // (written as the fused operations they become, so that every instantiation rounds them the same way)
    a = ig * __builtin_fmaf(-g, g, 1.0f);
    b = g * __builtin_fmaf(-ig, ig, ig);
}

// c' = sig(f) c + sig(i) tanh(j);  h' = sig(o) tanh(c')   (reference convLSTM.py:40-43)
template <bool FAST>
__device__ __forceinline__ void lstm_pixel(const float (&z)[32], const float (&cp)[8], float (&hn)[8], float (&cn)[8],
                                           float (&gs)[32]) {
#pragma unroll
    for (int cl = 0; cl < 8; ++cl) {
        const float g = act_tanh<FAST>(z[4 * cl]);
        const float ig = act_sigmoid<FAST>(z[4 * cl + 1]);
        const float fg = act_sigmoid<FAST>(z[4 * cl + 2]);
        const float og = act_sigmoid<FAST>(z[4 * cl + 3]);
        const float c = __builtin_fmaf(fg, cp[cl], ig * g);  // (one fused form in every instantiation)
        cn[cl] = c;
        hn[cl] = og * act_tanh<FAST>(c);
        gs[4 * cl] = g, gs[4 * cl + 1] = ig, gs[4 * cl + 2] = fg, gs[4 * cl + 3] = og;
    }
}

template <typename T>
__device__ __forceinline__ void store8(T* dst, const float* v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        const bf16x8 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3],
                          (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
        *reinterpret_cast<bf16x8*>(dst) = o;
    }
}

// Everything after the GEMM for one live pixel: history taps, gate math, state / output / gate stores.
// `table` is the history tap table (LDS or global); z already holds accumulator + bias.
template <typename T, bool FAST>
__device__ __forceinline__ void lstm_finish_pixel_with_state(const ConvParams& p, int b, int gy, int gx, int half,
                                                             float (&z)[32], const float (&cp)[8],
                                                             const float* table) {
    const size_t pixel = ((size_t)b * p.H + gy) * p.W + gx;
    const size_t so = pixel * F_ + 8 * half;
    if (p.hist > 0) add_history_taps(z, p.hist_points + (size_t)b * p.hist_bstride, p.hist, table, gx, gy, half);
    float hn[8], cn[8], gs[32];
    lstm_pixel<FAST>(z, cp, hn, cn, gs);
    *reinterpret_cast<f32x4*>(p.c_out + so) = f32x4{cn[0], cn[1], cn[2], cn[3]};
    *reinterpret_cast<f32x4*>(p.c_out + so + 4) = f32x4{cn[4], cn[5], cn[6], cn[7]};
    store8(static_cast<T*>(p.h_out) + pixel * p.h_pstride + 8 * half, hn);
    if (p.out)
        store8(static_cast<T*>(p.out) + (size_t)b * p.out_bstride + ((size_t)gy * p.W + gx) * p.out_pstride + 8 * half, hn);
    if (p.gates && p.save_layout == ISG_SAVE_FACTORS) {
        T* fd = static_cast<T*>(p.gates) + pixel * NG + 32 * half;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {  // this lane's four channel pairs
            float v[8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int cl = 2 * pr + k;
                lstm_backward_factors(gs[4 * cl], gs[4 * cl + 1], v[k], v[2 + k]);
                v[4 + k] = gs[4 * cl + 2], v[6 + k] = gs[4 * cl + 3];
            }
            store8(fd + 8 * pr, v);
        }
    } else if (p.gates) {
        T* gd = static_cast<T*>(p.gates) + pixel * NG + 32 * half;
#pragma unroll
        for (int k = 0; k < 4; ++k) store8(gd + 8 * k, gs + 8 * k);
    }
}

// Same, loading the previous cell state itself.
template <typename T, bool FAST>
__device__ __forceinline__ void lstm_finish_pixel(const ConvParams& p, int b, int gy, int gx, int half, float (&z)[32],
                                                  const float* table) {
    const size_t so = (((size_t)b * p.H + gy) * p.W + gx) * F_ + 8 * half;
    f32x4 c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = c0;  // (c_prev null: the zero state)
    if (p.c_prev) {
        c0 = *reinterpret_cast<const f32x4*>(p.c_prev + so);
        c1 = *reinterpret_cast<const f32x4*>(p.c_prev + so + 4);
    }
    const float cp[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    lstm_finish_pixel_with_state<T, FAST>(p, b, gy, gx, half, z, cp, table);
}

// dynamic_rnn past the end of a sequence: the state of the tile's pixels is copied through and the
// emitted output is zero.  Called by `nthreads` threads with ids 0..nthreads-1.  CT: the cell state's type.
template <typename T, typename CT = float>
__device__ __forceinline__ void lstm_copy_through_tile(const ConvParams& p, int b, int ty0, int tx0, int tid,
                                                       int nthreads, int tile_rows = TH) {
    for (int i = tid; i < tile_rows * TW * (F_ / 4); i += nthreads) {
        const int q4 = i & 3, pix = i >> 2;
        const int gy = ty0 + pix / TW, gx = tx0 + (pix & (TW - 1));
        if (gy < p.H && gx < p.W) {
            const size_t pixel = ((size_t)b * p.H + gy) * p.W + gx;
            const size_t o = pixel * F_ + q4 * 4;
            float hv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            float cv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (p.c_prev) {  // (null: the zero state passes through as zeros)
                load4(reinterpret_cast<const CT*>(p.c_prev) + o, cv);
                load4(static_cast<const T*>(p.in1) + o, hv);
            }
            store4(reinterpret_cast<CT*>(p.c_out) + o, cv);
            store4(static_cast<T*>(p.h_out) + pixel * p.h_pstride + q4 * 4, hv);
            if (p.out) {
                const float zero[4] = {0.f, 0.f, 0.f, 0.f};
                store4(static_cast<T*>(p.out) + (size_t)b * p.out_bstride + ((size_t)gy * p.W + gx) * p.out_pstride + q4 * 4,
                       zero);
            }
        }
    }
}

// Data-gradient epilogue of one pixel and one group of 4 consecutive rows (input channels cc0..cc0+3).
template <typename T>
__device__ __forceinline__ void dgrad_store4(const ConvParams& p, size_t pixel, int cc0, float (&v)[4]) {
    if (cc0 < p.row_split) {  // x part of the concatenated input
        if (p.dx && cc0 < p.dx_channels) {
            T* d = static_cast<T*>(p.dx) + pixel * p.dx_cstride + cc0;
            if (p.dx_accumulate) {
                float old[4];
                load4(d, old);
                v[0] += old[0], v[1] += old[1], v[2] += old[2], v[3] += old[3];
            }
            store4(d, v);
        }
    } else if (cc0 < p.row_split + F_) {
        store4(static_cast<T*>(p.dh_prev) + pixel * F_ + (cc0 - p.row_split), v);
    }
}

// A masked sample's dY is zero: dh_prev keeps the carried gradient; dx must still be defined.
template <typename T>
__device__ __forceinline__ void dgrad_zero_dx_tile(const ConvParams& p, int b, int ty0, int tx0, int tid, int nthreads,
                                                   int tile_rows = TH) {
    if (!p.dx || p.dx_accumulate) return;
    const int nq = p.dx_channels / 4;
    for (int i = tid; i < tile_rows * TW * nq; i += nthreads) {
        const int q4 = i % nq, pix = i / nq;
        const int gy = ty0 + pix / TW, gx = tx0 + (pix & (TW - 1));
        if (gy < p.H && gx < p.W) {
            const float zero[4] = {0.f, 0.f, 0.f, 0.f};
            store4(static_cast<T*>(p.dx) + (((size_t)b * p.H + gy) * p.W + gx) * p.dx_cstride + q4 * 4, zero);
        }
    }
}

struct WgradParams {
    const void* in0;  // x [B,H,W,c0]
    const void* in1;  // h_prev [B,H,W,F]
    int c0, c1;
    const void* dy;   // [B,H,W,64] (channel, gate) order
    int B, H, W, tiles_x, tiles_y;
    const int* durations;
    int t;
    float* slab;       // [gridDim.x][9][MT*32][64]
    float* bias_slab;  // [gridDim.x][64]
    int slab_accumulate;  // add to what the slabs hold (sums deferred over several launches)
};

// bf16 ping-pong fast path (isg_convlstm_pp.hip).  Returns ISG_OK, an error, or kNotSupported when
// the shape has no fast instantiation (the caller then uses the generic kernel).
constexpr int kNotSupported = 1;
int launch_conv_pp(const ConvParams& p, int mt, int mode, int num_cus, hipStream_t stream);
// bf16 weight-gradient fast path: double-buffered LDS, 8 waves sharing the 9 x MT x 2 output tiles.
int launch_wgrad_pp(const WgradParams& p, int mt, int grid, hipStream_t stream);

}  // namespace isgc
