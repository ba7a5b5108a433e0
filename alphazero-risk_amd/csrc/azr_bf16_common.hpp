// azr_bf16_common.hpp — constants, packed-weight layout and small device helpers shared by the bf16 MFMA tower kernels
// (azr_net_bf16.hip: 1..3 boards per workgroup, double-buffered; azr_tower_sb.hip: 4 boards, single LDS buffer).
#pragma once
#include <stdint.h>
#include <string.h>

#include "azr_internal.hpp"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace azr {
constexpr int ROWB = 544;                 // LDS bytes per activation row: 256 bf16 + 32 B pad: 16-B slot = (2*row + kgroup) mod 16 -> ds_read_b128 lane groups conflict-free
constexpr int FROWB = 32;                 // LDS bytes per stem-feature row: 16 bf16 (13 planes + 3 zero)
constexpr int KS_PER_TAP = 8;             // 256 input channels / 32 per MFMA
constexpr int STEM_KS = 5;                // 9 taps x 16 channels = 144 -> 5 k-steps of 32 (last half zero)
constexpr size_t FRAGS_PER_KSTEP = 16;    // 8 waves x 2 n-tiles, 64 lanes x 16 B each
constexpr size_t TOWER_LAYER_HALFS = (size_t)9 * KS_PER_TAP * FRAGS_PER_KSTEP * 64 * 8;  // = 2304 * 256
constexpr size_t STEM_HALFS = (size_t)STEM_KS * FRAGS_PER_KSTEP * 64 * 8;

__host__ __device__ inline uint16_t f2bf(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint16_t bf_rne(float f)
{
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32, round-to-nearest-even
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// The 16-bit element type of a tower: bf16 (AZR_NET_BF16) or fp16 (AZR_NET_F16).  Same kernels, same packed-fragment layout, same MFMA
// rate (v_mfma_f32_16x16x32_bf16 / _f16); fp16 keeps 11 significand bits instead of 8 — the tower's error against an exact evaluation
// drops ~7x (tools/net_precision.py) — and pays for it with range: conv weights are packed as 2^k w per layer (the exact inverse goes
// into the folded BN scale, as for NET_F32X) and activations saturate at 65504 instead of overflowing.
template <bool F16> struct El;
template <> struct El<false> {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    template <typename TB, typename TA>   // any 16-byte register types (a ring slot is a u32x4, an LDS fragment an s16x8)
    static __device__ __forceinline__ f32x4 mfma(const TB& b, const TA& a, const f32x4& c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint16_t rne(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
    static __device__ __forceinline__ float tof(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
    // two packed elements (low half first) <-> two floats
    static __device__ __forceinline__ float lo_of(uint32_t u) { return __uint_as_float(u << 16); }
    static __device__ __forceinline__ float hi_of(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
    template <typename V2f> static __device__ __forceinline__ uint32_t pack_relu(const V2f& x)   // RNE, then ReLU on the rounded pair
    {
        typedef __attribute__((ext_vector_type(2))) short s16x2_t;
        const s16x2_t z = {0, 0};
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, __builtin_convertvector(x, v2)), z));
    }
};
template <> struct El<true> {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    template <typename TB, typename TA>
    static __device__ __forceinline__ f32x4 mfma(const TB& b, const TA& a, const f32x4& c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, b), __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint16_t rne(float f) { return __builtin_bit_cast(uint16_t, (_Float16)fminf(f, 65504.0f)); }   // (stem features: once per launch)
    static __device__ __forceinline__ float tof(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
    static __device__ __forceinline__ float lo_of(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu)); }
    static __device__ __forceinline__ float hi_of(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
    template <typename V2f> static __device__ __forceinline__ uint32_t pack_relu(const V2f& x)   // RNE, ReLU and saturation on the rounded pair
    {
        // as signed 16-bit integers the non-negative fp16 bit patterns are ordered like their values, +inf = 0x7c00 just above the
        // largest finite 0x7bff: max with 0 is the ReLU (-0 -> +0), min with 0x7bff the saturation (v_pk_max_i16, v_pk_min_i16)
        typedef __attribute__((ext_vector_type(2))) short s16x2_t;
        const s16x2_t z = {0, 0}, m = {0x7bff, 0x7bff};
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(s16x2_t, __builtin_convertvector(x, v2)), z), m));
    }
};

// setInStateTensor (alphazero_nn.cpp:31-67) for one cell/plane, from the 88-byte NNInputData image in LDS
__device__ __forceinline__ float plane_value(const uint8_t* in88, int pos, int c)
{
    const uint32_t b = in88[pos];
    const int army = b & 63, owner = b >> 6, cur = in88[42], enemy = cur == 0 ? 1 : 0;
    const float fa = (float)army * 0.03125f;   // = army / 32 exactly (a power of two), without the correctly-rounded division sequence
    const float* f = reinterpret_cast<const float*>(in88 + 48);
    switch (c) {
    case 0: return owner == cur ? fa : 0.0f;
    case 1: return owner == enemy ? fa : 0.0f;
    case 2: return owner == 2 ? fa : 0.0f;
    case 3: return f[9];
    case 4: return f[0];
    case 5: return f[1];
    case 6: return f[2];
    default: return c < 13 ? f[3 + (c - 7)] : 0.0f;
    }
}


constexpr size_t KSTRIDE = FRAGS_PER_KSTEP * 64;   // s16x8 units between consecutive k-steps
constexpr size_t KBYTES = KSTRIDE * 16;            // bytes per k-step of packed weights (all 16 column tiles)
constexpr int MAX_RING = 16;                       // deepest weight ring any kernel runs ahead (k-steps): run-off padding

struct Bf16Net {
    bool f16 = false;                 // AZR_NET_F16: the packed weights and the activations are fp16 (El<true>), `fold16` carries the weight scales
    float* fold16 = nullptr;          // [14 + 2B * 2 * 256] folded BN with 2^-k of the layer's weight scale in the scale rows
    uint16_t* stem_wp = nullptr;
    uint16_t* tower_wp = nullptr;
    unsigned long long* diag = nullptr;  // set only by azr_debug_tower_clock
    int sb_mode = 1;   // use of the single-image tiles: 0 never, 1 plan, 2 / 3 / 4 force 4 / 2 / 3 boards (AZR_TOWER_SB, read once at creation)
    int sc_mode = 1;   // launches of <= 128 boards on the split-channel tower (azr_tower_sc.hip); 0 = one board per workgroup (AZR_TOWER_SC, read once at creation)
    uint16_t* sc_ex = nullptr;        // split-channel tower: the exchange images [2 parities][128 pairs][96 rows][256] bf16
    unsigned* sc_counters = nullptr;  // ... the pairs' arrival counters and XCC words
    unsigned sc_spin_limit = 0;       // ... polls of a hand-off before it gives up (a constant in the product library; a test hook otherwise)
    unsigned sc_serial = 0, sc_tag = 0;  // ... launches so far; (serial << 2) of the last one = what its give-up word is compared with
    int sc_force_wt = 0;              // ... test hook: never the plain-store form of a same-XCD pair
};
// words of sc_counters behind the per-pair words: the running launch's give-up word (raised by a hand-off that ran out of polls; the
// guarded k_tower_bf16<1> launch queued behind every k_tower_sc launch recomputes the batch when it is up) and the count of such launches
constexpr int SC_W_GIVEUP = 128, SC_W_FALLBACKS = 129, SC_WORDS = 132;
inline Bf16Net* bf16net(azr_engine* h) { return reinterpret_cast<Bf16Net*>(h->net.bf16ctx); }
const float* net_head_params(azr_engine* h);
const float* net_fold(azr_engine* h);
// azr_tower_sb.hip
int tower_sb_init(azr_engine* h);
int tower_sb_launch(azr_engine* h, int nb, int wgs, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st);
// azr_tower_sc.hip
int tower_sc_init(azr_engine* h);
void tower_sc_free(azr_engine* h);
int tower_sc_launch(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st,
                    const int* n_dev = nullptr, const int* n_other = nullptr);
int tower_sc_fallbacks(azr_engine* h, unsigned long long* out);
}  // namespace azr
