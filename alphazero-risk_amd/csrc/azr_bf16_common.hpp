// azr_bf16_common.hpp — constants, packed-weight layout and small device helpers shared by the bf16 MFMA tower kernels
// (azr_net_bf16.hip: 1..3 boards per workgroup, double-buffered; azr_tower_sb.hip: 4 boards, single LDS buffer).
#pragma once
#include <stdint.h>
#include <string.h>

#include "azr_internal.hpp"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
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
    uint16_t* stem_wp = nullptr;
    uint16_t* tower_wp = nullptr;
    unsigned long long* diag = nullptr;  // set only by azr_debug_tower_clock
    int sb_mode = 1;   // use of the single-image tiles: 0 never, 1 plan, 2 / 3 / 4 force 4 / 2 / 3 boards (AZR_TOWER_SB, read once at creation)
    int sc_mode = 1;   // launches of <= 256 boards on the split-channel tower (azr_tower_sc.hip); 0 = one board per workgroup (AZR_TOWER_SC, read once at creation)
    uint16_t* sc_ex = nullptr;        // split-channel tower: the exchange images [2 parities][128 pairs][96 rows][256] bf16
    unsigned* sc_counters = nullptr;  // ... the pairs' arrival counters [128] and the error word
};
inline Bf16Net* bf16net(azr_engine* h) { return reinterpret_cast<Bf16Net*>(h->net.bf16ctx); }
const float* net_head_params(azr_engine* h);
const float* net_fold(azr_engine* h);
// azr_tower_sb.hip
int tower_sb_init(azr_engine* h);
int tower_sb_launch(azr_engine* h, int nb, int wgs, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st);
// azr_tower_sc.hip
int tower_sc_init(azr_engine* h);
void tower_sc_free(azr_engine* h);
int tower_sc_launch(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st);
int tower_sc_check(azr_engine* h);
}  // namespace azr
