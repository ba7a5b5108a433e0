// azr_tower_fx.hip — k_tower_fx<NB>: NET_F32X, the whole policy/value net (python/src/build_graph.py:63-90) at the precision of
// the reference's fp32 evaluation (alphazero_nn.cpp:247-248) on the fp16 MFMA of gfx950.
//
// Why.  The reference evaluates the net in fp32 and plays argmax-N (alphazero_player.cpp:11-12).  The bf16 tower is 16 x the
// fp32 rate but differs from an fp32 evaluation by up to 1e-2 in pi / v; the fp32 VALU path is exact and slow.  This kernel is
// the fp32-equivalent path at matrix-core rate.
//
// How.  Every 3x3 conv operand x (activation or weight) is held as an fp16 PAIR  x = hi + lo,  hi = rne16(x), lo = rne16(x - hi):
// 22 significand bits per operand (the low part of a value below 2^-3 is an fp16 subnormal, which the matrix core takes at full
// value — profiles/r03_mfma_round_probe.txt — so its absolute error is 2^-25: the size of rounding the operand to fp32).  A product
// a * w = ah*wh + ah*wl + al*wh + al*wl and the last term is dropped.  tools/f32x_split_study.py (float64 emulation of the split
// alone, B = 20, the 128 boards of the parity test): max |d pi| 2.3e-7, max |d v| 3.5e-7 (with the low parts pre-scaled by 2^11, the
// round-3 form: 1.6e-7 / 2.9e-7 — no difference that matters against the 2e-5 gate); bf16 pairs with the same 3 passes stop at
// 2.2e-5 / 3.4e-5, six bf16 passes would be needed for what three fp16 passes give.
// Per layer, ONE set of fp32 accumulators and ONE pass over the 72 k-steps, 12 MFMAs per row tile and k-step:
//     acc += wh * ah ;  acc += wh * al ;  acc += wl * ah          (products of fp16 values are exact in the fp32 accumulate)
//     epilogue: folded BN (fp32 fma), shortcut add in fp32 (the block input stays in fp32 REGISTERS), ReLU, split into the pair
// i.e. 3 x the MFMAs of the bf16 tower on v_mfma_f32_16x16x32_f16.  (Round 3 ran the cross terms as a phase of their own, at 2^11
// times their weight, and re-read the wh half of the stream for the main term: 12 weight fragments per tile and k-step pair where
// this form loads 8 — the kernel was held by the weight stream through the CU's L1, 42 B/clk of 64 — and 144 k-steps of loop
// overhead where this has 72.)
// The stem (K = 9 x 13) runs on the fp32-input MFMA v_mfma_f32_16x16x4_f32 from fp32 planes and fp32 weights: exact fp32.
// The heads read the reconstructed fp32 activations (hi + lo is exact in fp32) and run the fp32 fma chains of k_heads.
//
// Structure = k_tower_sb (azr_tower_sb.hip): one workgroup of 4 waves (one per SIMD) owns NB boards for the whole net, the
// activations never leave LDS (two planes: hi image, lo' image), waves split the 256 output channels (4 tiles of 16 each), weight
// fragments stream global -> registers through a ring that never drains, border-class row order with skipped (tile, tap)
// pairs, a layer's 72 k-steps fully unrolled.  Packed weights: per layer and k-step [16 column tiles of wh | 16 of wl], one
// linear stream over all layers.
// Range: fp16 holds |x| < 65504.  Weights are checked when they are packed (azr_nn_set_weights fails loudly); an activation
// beyond the range would turn into inf -> NaN outputs, which the search counts as rule errors (never silently wrong).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "azr_internal.hpp"
#include "azr_bf16_common.hpp"
#include "azr_rowclass.hpp"

using namespace azr;

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (void)hipGetLastError(); /* the runtime's last-error slot is sticky: clear it */    \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return AZR_E_HIP;                                                                   \
        }                                                                                       \
    } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

#ifndef FX_DROP
#define FX_DROP 0     /* diagnostics: 1 / 2 = leave out the wh x al' / wl' x ah cross term */
#endif
namespace {
constexpr int NT = 4;                  // 16-channel column tiles per wave (4 waves x 64 channels)
constexpr int WAVES = 4, THREADS = 256;
constexpr int STEPS = 72;              // k-steps of a layer (9 taps x 8 slices of 32 input channels)
constexpr uint32_t STEP_BYTES = 2 * (uint32_t)KBYTES;            // packed bytes per k-step: wh tiles | wl tiles
constexpr uint32_t LAYER_BYTES = STEPS * STEP_BYTES;
constexpr int STEM_K = 9 * 13, STEM_KS4 = (STEM_K + 3) / 4;      // stem on the 16x16x4 fp32 MFMA: 30 k-steps

template <int NB_>
struct FX {
    static constexpr int NB = NB_;
    static constexpr int ROWS = 42 * NB;
    static constexpr int MT = (ROWS + 15) / 16;
    static constexpr int ZR = MT * 16;                       // index of the shared zero row
    static constexpr int RING = 4;                           // weight ring depth in k-steps (144 = 0 mod RING)
    static constexpr int PLANE = (ZR + 1) * ROWB;            // one activation image (hi, or lo') incl. its zero row
    static constexpr int FEAT_OFF = 2 * PLANE;               // stem features, fp32 [(ZR + 1)][16]
    static constexpr int HEAD_OFF = FEAT_OFF + (ZR + 1) * 64;
    static constexpr int IN88_OFF = HEAD_OFF + NB * 448 * 4;
    static constexpr int ROWOF_OFF = IN88_OFF + NB * 96;     // u8 [ZR]: cell (board * 42 + pos) -> row
    static constexpr int TAPROW_OFF = ROWOF_OFF + ZR;        // u8 [10][ZR]: source row of (tap, row); tap 9 = all zero row
    static constexpr int ROWCELL_OFF = TAPROW_OFF + 10 * ZR; // u16 [ZR]: row -> y | x << 4 | board << 8, 0xffff = pad row
    static constexpr int FOLD_OFF = (ROWCELL_OFF + 2 * ZR + 15) / 16 * 16;   // float [2][256]: the layer's folded BN scale | shift
    static constexpr int LDS_BYTES = FOLD_OFF + 2 * NF * 4;
    static_assert(NB == 2, "tile shapes (the row order of azr_rowclass.hpp)");
    static_assert(STEPS % RING == 0, "ring depth");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(PLANE % 16 == 0 && FEAT_OFF % 16 == 0 && HEAD_OFF % 16 == 0 && TAPROW_OFF % 4 == 0 && ROWCELL_OFF % 2 == 0, "alignment");
    static_assert(3 * NF * 4 <= (ZR + 1) * 64, "the heads stage 3 x 256 floats in the stem feature image");
};

__device__ __forceinline__ f16x8 lds16h(const uint8_t* p) { return *reinterpret_cast<const f16x8*>(p); }

// the pair of an fp32 value: hi = rne16(y), lo = rne16(y - hi)
__device__ __forceinline__ void split_pair(float y, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)y;
    lo = (_Float16)(y - (float)hi);
}

// layer epilogue for 4 consecutive channels of one board cell: folded BN (fp32 fma), optional shortcut add (fp32), ReLU;
// returns the fp32 result and its pair images (4 x fp16 each)
template <bool SHORTCUT>
__device__ __forceinline__ void bn_relu_split(const f32x4& acc, const float4& s, const float4& h, f32x4& x, uint2& ohi, uint2& olo)
{
    f32x4 y;
    y[0] = fmaf(acc[0], s.x, h.x); y[1] = fmaf(acc[1], s.y, h.y); y[2] = fmaf(acc[2], s.z, h.z); y[3] = fmaf(acc[3], s.w, h.w);
    if (SHORTCUT) y += x;
#pragma unroll
    for (int i = 0; i < 4; i++) y[i] = y[i] > 0.0f ? y[i] : 0.0f;
    x = y;
    _Float16 hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; i++) split_pair(y[i], hi[i], lo[i]);
    ohi = uint2{__builtin_bit_cast(uint32_t, f16x2{hi[0], hi[1]}), __builtin_bit_cast(uint32_t, f16x2{hi[2], hi[3]})};
    olo = uint2{__builtin_bit_cast(uint32_t, f16x2{lo[0], lo[1]}), __builtin_bit_cast(uint32_t, f16x2{lo[2], lo[3]})};
}

#define MFMA16(W, A, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, W), A, ACC, 0, 0, 0)

// One tap (8 k-steps of 32 input channels) of a 3x3 conv layer for the MT row tiles x 4 column tiles of a wave: per tile and k-step
// 12 MFMAs (wh x ah, wh x al, wl x ah).  Everything that depends on the tap — which tiles run, ring slots, wait counts — is a
// compile-time constant; `ws` = byte offset of the layer's k-step 0 in the packed stream (the stream is linear over the layers, so the
// ring's look-ahead simply runs into the next layer).  The non-MFMA work of a k-step (the 8 refill loads of the ring slot the previous
// k-step freed, the LDS fragment reads, next-tap table lookups) is dealt out between the MFMAs as in k_tower_sb.
template <int NB, int TAP>
__device__ __forceinline__ void conv_tap(const uint8_t* bufH, const uint8_t* bufL, const uint8_t* tr_c, uint32_t g16,
                                         const __amdgpu_buffer_rsrc_t wsrc, uint32_t loff, uint32_t ws, u32x4 (&bq)[FX<NB>::RING][2 * NT],
                                         f32x4 (&acc)[FX<NB>::MT][NT], f16x8 (&ah)[FX<NB>::MT], f16x8 (&al)[FX<NB>::MT], uint32_t (&ap)[FX<NB>::MT])
{
    constexpr int MT = FX<NB>::MT, ZR = FX<NB>::ZR, RING = FX<NB>::RING;
    constexpr int NTAP = TAP < 8 ? TAP + 1 : 9;                   // the tap that runs next (9 = none: the layer ends)
    constexpr uint32_t sk = skip_mask<NB>(TAP), skn = skip_mask<NB>(NTAP);
    constexpr int active = MT - __builtin_popcount(sk & ((1u << MT) - 1u));   // tiles that run this tap
    constexpr int SLOTS = 2 * active;                                          // refill slots of a k-step (MFMA gaps that take a load)
    constexpr int NL = 2 * NT;                                                 // refill loads of a k-step
    uint32_t np[MT];
#pragma unroll
    for (int ks = 0; ks < KS_PER_TAP; ks++) {
        const int gk = TAP * KS_PER_TAP + ks;                       // k-step inside the layer
        const int cur = gk % RING, ref = (gk + RING - 1) % RING;    // ring slot in use / slot freed by the previous k-step
        const uint32_t fut = ws + (uint32_t)(gk + RING - 1) * STEP_BYTES;   // the k-step whose fragments go into `ref` now
        // this k-step's fragments have landed; the loads of the RING - 2 k-steps behind it stay in flight: vmcnt(16) (6-bit field:
        // low 4 bits in [3:0], high 2 in [15:14]); lgkmcnt / expcnt untouched
        static_assert(RING == 4 && NL == 8, "the wait count below is that of a 4-deep ring of 8 loads per k-step");
        __builtin_amdgcn_s_waitcnt(0x0F70 | (16 & 15) | ((16 >> 4) << 14));
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            if (!((sk >> mt) & 1u)) {
                const int j = __builtin_popcount(~sk & ((1u << mt) - 1u));   // index among the active tiles
                // refill load q of the 8 goes into gap `slot q` = (q * SLOTS) / NL
#define REFILL(SLOT)                                                                                                                       \
    _Pragma("unroll") for (int q = 0; q < NL; q++) if ((SLOT) == (q * SLOTS) / NL)                                                         \
        bq[ref][q] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, loff + (q & 3) * 1024, (int)(fut + (q >> 2) * (uint32_t)KBYTES), 0)
                MFMA16(bq[cur][0], ah[mt], acc[mt][0]);
                REFILL(2 * j);
                __builtin_amdgcn_sched_barrier(0);
                MFMA16(bq[cur][1], ah[mt], acc[mt][1]);
                MFMA16(bq[cur][2], ah[mt], acc[mt][2]);
                MFMA16(bq[cur][3], ah[mt], acc[mt][3]);
#if FX_DROP != 1
                MFMA16(bq[cur][0], al[mt], acc[mt][0]);
                MFMA16(bq[cur][1], al[mt], acc[mt][1]);
                MFMA16(bq[cur][2], al[mt], acc[mt][2]);
                MFMA16(bq[cur][3], al[mt], acc[mt][3]);
#endif
                if (ks < KS_PER_TAP - 1) al[mt] = lds16h(bufL + ap[mt] + (ks + 1) * 64);
                __builtin_amdgcn_sched_barrier(0);
#if FX_DROP != 2
                MFMA16(bq[cur][4], ah[mt], acc[mt][0]);
#endif
                REFILL(2 * j + 1);
                __builtin_amdgcn_sched_barrier(0);
#if FX_DROP != 2
                MFMA16(bq[cur][5], ah[mt], acc[mt][1]);
                MFMA16(bq[cur][6], ah[mt], acc[mt][2]);
                MFMA16(bq[cur][7], ah[mt], acc[mt][3]);
#endif
                if (ks < KS_PER_TAP - 1) ah[mt] = lds16h(bufH + ap[mt] + (ks + 1) * 64);
#undef REFILL
            }
            // the next tap's source rows: one table byte per tile slot two k-steps before the tap ends, its address arithmetic
            // one k-step later, the first fragments of the next tap in the last k-step
            if (NTAP < 9) {
                if (ks == KS_PER_TAP - 3) { if (!((skn >> mt) & 1u)) np[mt] = (uint32_t)tr_c[NTAP * ZR + mt * 16]; }
                if (ks == KS_PER_TAP - 2) { if (!((skn >> mt) & 1u)) np[mt] = np[mt] * ROWB + g16; }
                if (ks == KS_PER_TAP - 1) {
                    if (!((skn >> mt) & 1u)) {
                        ah[mt] = lds16h(bufH + np[mt]);
                        al[mt] = lds16h(bufL + np[mt]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (NTAP < 9) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) ap[mt] = np[mt];
    }
}

template <int NB>
__global__ __launch_bounds__(THREADS, 1) void k_tower_fx(const uint8_t* __restrict__ in88, int in_stride, int n,
                                                          const float* __restrict__ stem_w, const uint16_t* __restrict__ tower_wp,
                                                          const float* __restrict__ fold, int blocks, const float* __restrict__ hp,
                                                          float* __restrict__ pi_out, float* __restrict__ v_out,
                                                          const int* __restrict__ slot_map)
{
    using G = FX<NB>;
    constexpr int ROWS = G::ROWS, MT = G::MT, ZR = G::ZR, RING = G::RING;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* bufH = lds;
    uint8_t* bufL = lds + G::PLANE;
    float* featF = reinterpret_cast<float*>(lds + G::FEAT_OFF);
    uint8_t* in_l = lds + G::IN88_OFF;
    uint8_t* rowof = lds + G::ROWOF_OFF;
    uint8_t* taprow = lds + G::TAPROW_OFF;
    uint16_t* rowcell = reinterpret_cast<uint16_t*>(lds + G::ROWCELL_OFF);
    float* foldl = reinterpret_cast<float*>(lds + G::FOLD_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;        // MFMA fragment coordinates: board cell (column) c of a tile, k-group g
    const int board0 = blockIdx.x * NB;

    // ---- weight ring: the first RING - 1 k-steps of layer 0 fly while the tables and the stem are built
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(tower_wp), (short)0, 0x7fffffff, 0x00020000);
    const uint32_t loff = (uint32_t)((wave * NT) * 64 + lane) * 16u;   // this lane's fragment bytes inside a 16-tile block
    uint32_t wl = 0;                                                     // byte offset of the current layer (wave-uniform)
    u32x4 bq[RING][2 * NT];
#pragma unroll
    for (int ks = 0; ks < RING - 1; ks++)
#pragma unroll
        for (int q = 0; q < 2 * NT; q++)
            bq[ks][q] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, loff + (q & 3) * 1024, (int)((uint32_t)ks * STEP_BYTES + (q >> 2) * (uint32_t)KBYTES), 0);

    // ---- stage the NNInputData images, build the row tables
    for (int i = tid; i < NB * 96; i += THREADS) {
        const int b = i / 96, o = i % 96;
        const int slot = (board0 + b < n) ? (slot_map ? slot_map[board0 + b] : board0 + b) : 0;
        in_l[i] = (board0 + b < n && o < 88) ? in88[(size_t)slot * in_stride + o] : (uint8_t)0;
    }
    for (int i = tid; i < ROWB / 4; i += THREADS) {
        reinterpret_cast<uint32_t*>(bufH + ZR * ROWB)[i] = 0;
        reinterpret_cast<uint32_t*>(bufL + ZR * ROWB)[i] = 0;
    }
    for (int i = tid; i < ZR; i += THREADS) rowcell[i] = 0xffffu;
    __syncthreads();
    for (int i = tid; i < ROWS; i += THREADS) {
        const int b = i / 42, pos = i - b * 42, r = row_of<NB>(b, pos);
        rowof[i] = (uint8_t)r;
        rowcell[r] = (uint16_t)((pos / 6) | ((pos % 6) << 4) | (b << 8));
    }
    __syncthreads();
    for (int i = tid; i < 10 * ZR; i += THREADS) {   // source row of row r under tap t (pad rows and out-of-board taps: the zero row)
        const int t = i / ZR, r = i - t * ZR;
        const int ci = rowcell[r];
        int src = ZR;
        if (t < 9 && ci != 0xffff) {
            const int y = (ci & 15) + t / 3 - 1, x = ((ci >> 4) & 15) + t % 3 - 1;
            if ((unsigned)y < 7u && (unsigned)x < 6u) src = rowof[(ci >> 8) * 42 + y * 6 + x];
        }
        taprow[i] = (uint8_t)src;
    }
    // stem features in fp32: featF as [ZR + 1][16] (row ZR = zero row); planes 13..15 are zero
    for (int i = tid; i < (ZR + 1) * 16; i += THREADS) {
        const int r = i >> 4, ch = i & 15;
        float v = 0.0f;
        const int ci = r < ZR ? rowcell[r] : 0xffff;
        if (ci != 0xffff) v = plane_value(in_l + (ci >> 8) * 96, (ci & 15) * 6 + ((ci >> 4) & 15), ch);
        featF[i] = v;
    }
    __syncthreads();

    f32x4 acc[MT][NT];
    f32x4 res[MT][NT];      // the block input of this wave's (cell, 4-channel) elements in fp32 = the residual operand
    const uint32_t eoff = (uint32_t)(c * ROWB + (wave * 64 + g * 4) * 2);   // epilogue store address of tile 0 / column tile 0

    // ---- stem: 3x3 conv 13 -> 256 on the fp32-input MFMA (exact fp32 products and sums): D[channel][cell], k = tap * 13 + plane
    {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0, 0, 0, 0};
#pragma unroll 2
        for (int ks = 0; ks < STEM_KS4; ks++) {
            const int kk = 4 * ks + g;                       // this lane's k of the k-step
            const bool kv = kk < STEM_K;
            const int tap = kv ? kk / 13 : 9, ch = kv ? kk - tap * 13 : 0;
            float wv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) wv[nt] = kv ? stem_w[(size_t)kk * NF + (wave * NT + nt) * 16 + c] : 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int row = taprow[tap * ZR + mt * 16 + c];
                const float xv = featF[row * 16 + ch];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt], xv, acc[mt][nt], 0, 0, 0);
            }
        }
        // conv_bn over the board ROW (build_graph.py:68 axis=1) + ReLU
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int ci = rowcell[mt * 16 + c];
            const int y = ci == 0xffff ? 0 : (ci & 15);
            const float sc = fold[y], sh = fold[7 + y];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                uint2 ohi, olo;
                bn_relu_split<false>(acc[mt][nt], float4{sc, sc, sc, sc}, float4{sh, sh, sh, sh}, res[mt][nt], ohi, olo);
                if (c < pad_from<NB>(mt)) {
                    *reinterpret_cast<uint2*>(bufH + eoff + mt * 16 * ROWB + nt * 32) = ohi;
                    *reinterpret_cast<uint2*>(bufL + eoff + mt * 16 * ROWB + nt * 32) = olo;
                }
            }
        }
    }
    __syncthreads();

    // ---- residual tower: 2B conv layers, activations resident in the two LDS planes
    const uint8_t* tr_c = taprow + c;           // this lane's column of the (tap, row) -> source-row table
    const uint32_t g16 = (uint32_t)g * 16u;
    const int layers = 2 * blocks;
    for (int L = 0; L < layers; L++) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0, 0, 0, 0};
        uint32_t ap[MT];        // LDS byte address of this lane's fragment of tile mt at k-step 0 of the current tap
        f16x8 ah[MT], al[MT];   // ... and the fragments (hi plane, lo' plane) of the k-step about to run
        {
            const uint32_t sk0 = skip_mask<NB>(0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                ap[mt] = (uint32_t)tr_c[mt * 16] * ROWB + g16;
                if (!((sk0 >> mt) & 1u)) { ah[mt] = lds16h(bufH + ap[mt]); al[mt] = lds16h(bufL + ap[mt]); }
            }
        }
        conv_tap<NB, 0>(bufH, bufL, tr_c, g16, wsrc, loff, wl, bq, acc, ah, al, ap);
        conv_tap<NB, 1>(bufH, bufL, tr_c, g16, wsrc, loff, wl, bq, acc, ah, al, ap);
        conv_tap<NB, 2>(bufH, bufL, tr_c, g16, wsrc, loff, wl, bq, acc, ah, al, ap);
        conv_tap<NB, 3>(bufH, bufL, tr_c, g16, wsrc, loff, wl, bq, acc, ah, al, ap);
        // this layer's folded BN scale / shift (see the epilogue): requested 40 k-steps ahead of their use, not a whole layer
        const float2 fnext = *reinterpret_cast<const float2*>(fold + 14 + (size_t)L * 2 * NF + 2 * tid);
        conv_tap<NB, 4>(bufH, bufL, tr_c, g16, wsrc, loff, wl, bq, acc, ah, al, ap);
        conv_tap<NB, 5>(bufH, bufL, tr_c, g16, wsrc, loff, wl, bq, acc, ah, al, ap);
        conv_tap<NB, 6>(bufH, bufL, tr_c, g16, wsrc, loff, wl, bq, acc, ah, al, ap);
        conv_tap<NB, 7>(bufH, bufL, tr_c, g16, wsrc, loff, wl, bq, acc, ah, al, ap);
        conv_tap<NB, 8>(bufH, bufL, tr_c, g16, wsrc, loff, wl, bq, acc, ah, al, ap);
        wl += LAYER_BYTES;
        // this layer's folded BN scale / shift go through LDS (2 registers per lane over the k-steps instead of 32)
        *reinterpret_cast<float2*>(foldl + 2 * tid) = fnext;
        __syncthreads();        // every wave has read the planes for the last time
        // (scale / shift of a column tile are read where they are used: 8 live registers instead of 32)
        if (L & 1) {    // second conv of a block: + shortcut (the block's input, fp32 in registers); the output is the next block's input
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const float4 sc = *reinterpret_cast<const float4*>(foldl + wave * 64 + g * 4 + nt * 16);
                const float4 sh = *reinterpret_cast<const float4*>(foldl + NF + wave * 64 + g * 4 + nt * 16);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    uint2 ohi, olo;
                    bn_relu_split<true>(acc[mt][nt], sc, sh, res[mt][nt], ohi, olo);
                    if (c < pad_from<NB>(mt)) {
                        *reinterpret_cast<uint2*>(bufH + eoff + mt * 16 * ROWB + nt * 32) = ohi;
                        *reinterpret_cast<uint2*>(bufL + eoff + mt * 16 * ROWB + nt * 32) = olo;
                    }
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const float4 sc = *reinterpret_cast<const float4*>(foldl + wave * 64 + g * 4 + nt * 16);
                const float4 sh = *reinterpret_cast<const float4*>(foldl + NF + wave * 64 + g * 4 + nt * 16);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    uint2 ohi, olo;
                    f32x4 t;
                    bn_relu_split<false>(acc[mt][nt], sc, sh, t, ohi, olo);
                    if (c < pad_from<NB>(mt)) {
                        *reinterpret_cast<uint2*>(bufH + eoff + mt * 16 * ROWB + nt * 32) = ohi;
                        *reinterpret_cast<uint2*>(bufL + eoff + mt * 16 * ROWB + nt * 32) = olo;
                    }
                }
            }
        }
        __syncthreads();        // the new planes are complete
    }

    // ---- both heads (build_graph.py:76-90) on the reconstructed fp32 activations: the fma chains of k_heads (azr_net.hip)
    // (The thread index goes through an opaque move first: left alone, the compiler computes the heads' thread-dependent addresses
    //  ahead of the layer loop and carries them across it in registers the loop needs — three of them ended in scratch, written once
    //  and read once per thread: 1.5 MB of the launch's 1.7 MB of HBM writes.)
    {
        int tid_h = threadIdx.x;
        asm volatile("" : "+v"(tid_h));
        const int tid = tid_h, lane = tid & 63, wave = tid >> 6;
        const float* wpi = hp;              // [256][2]
        const float* bnpi = wpi + NF * 2;   // g[2] b[2] m[2] v[2]
        const float* wd = bnpi + 8;         // [84][43]
        const float* bd = wd + 84 * 43;     // [43]
        const float* wv = bd + 43;          // [256]
        const float* bnv = wv + NF;         // g b m v
        const float* w1 = bnv + 4;          // [42][256]
        const float* b1 = w1 + 42 * 256;    // [256]
        const float* w2 = b1 + 256;         // [256]
        const float* b2 = w2 + 256;         // [1]
        float* feat = reinterpret_cast<float*>(lds + G::HEAD_OFF);   // [NB][128]: 84 policy features, then 42 value features
        float* hid = feat + NB * 128;                             // [NB][256]
        float* logit = hid + NB * 256;                            // [NB][64]
        float* wl3 = featF;                                       // [3][256]: the three 1x1-conv weight columns
        for (int i = tid; i < 3 * NF; i += THREADS) wl3[i] = i < 2 * NF ? wpi[(i & (NF - 1)) * 2 + (i >> 8)] : wv[i - 2 * NF];
        __syncthreads();
        for (int idx = tid; idx < NB * 126; idx += THREADS) {  // 42 cells x {pi0, pi1, v} per board
            const int bb = idx / 126, t = idx % 126, pos = t / 3, ch = t % 3;
            const uint32_t ro = (uint32_t)rowof[bb * 42 + pos] * ROWB;
            const f16x8* h8 = reinterpret_cast<const f16x8*>(bufH + ro);
            const f16x8* l8 = reinterpret_cast<const f16x8*>(bufL + ro);
            const float* w = wl3 + ch * NF;
            float sacc = 0.0f;
            for (int q = 0; q < NF / 8; q++) {
                const f16x8 hh = h8[q], ll = l8[q];
#pragma unroll
                for (int e = 0; e < 8; e++) sacc = fmaf((float)hh[e] + (float)ll[e], w[8 * q + e], sacc);
            }
            const float* bnp = ch < 2 ? bnpi : bnv;
            const int nc = ch < 2 ? 2 : 1, kk = ch < 2 ? ch : 0;
            float y = (sacc - bnp[2 * nc + kk]) * (bnp[kk] / sqrtf(bnp[3 * nc + kk] + 1e-3f)) + bnp[nc + kk];
            y = y > 0.0f ? y : 0.0f;
            if (ch < 2) feat[bb * 128 + pos * 2 + ch] = y;  // NHWC flatten: (y*6+x)*2 + c
            else feat[bb * 128 + 84 + pos] = y;
        }
        __syncthreads();
        for (int idx = tid; idx < NB * 43; idx += THREADS) {
            const int bb = idx / 43, t = idx % 43;
            float sacc = 0.0f;
            for (int i = 0; i < 84; i++) sacc = fmaf(feat[bb * 128 + i], wd[i * 43 + t], sacc);
            logit[bb * 64 + t] = sacc + bd[t];
        }
        for (int idx = tid; idx < NB * 256; idx += THREADS) {
            const int bb = idx >> 8, t = idx & 255;
            float sacc = 0.0f;
            for (int i = 0; i < 42; i++) sacc = fmaf(feat[bb * 128 + 84 + i], w1[i * 256 + t], sacc);
            sacc += b1[t];
            hid[idx] = (sacc > 0.0f ? sacc : 0.0f) * w2[t];
        }
        __syncthreads();
        // one wave per (board, head): softmax over the 43 logits / tanh of the 256-term value sum
        for (int job = wave; job < NB * 2; job += WAVES) {
            const int bb = job >> 1;
            if (board0 + bb >= n) continue;
            const int slot = slot_map ? slot_map[board0 + bb] : board0 + bb;
            if ((job & 1) == 0) {
                const float lv = lane < 43 ? logit[bb * 64 + lane] : -INFINITY;
                float mx = lv;
                for (int sft = 32; sft >= 1; sft >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sft));
                const float e = lane < 43 ? expf(lv - mx) : 0.0f;
                float se = e;
                for (int sft = 32; sft >= 1; sft >>= 1) se += __shfl_xor(se, sft);
                if (lane < 43) pi_out[(size_t)slot * PI_STRIDE + lane] = e / se;
                if (lane == 43) pi_out[(size_t)slot * PI_STRIDE + 43] = 0.0f;
            } else {
                const float* hb = hid + bb * 256;
                float sacc = hb[lane] + hb[lane + 64] + hb[lane + 128] + hb[lane + 192];
                for (int sft = 32; sft >= 1; sft >>= 1) sacc += __shfl_xor(sacc, sft);
                if (lane == 0) v_out[slot] = tanhf(sacc + b2[0]);
            }
        }
    }
}

struct FxNet {
    uint16_t* tower_wp = nullptr;   // packed fp16 pairs: per layer 72 k-steps x [16 tiles wh | 16 tiles wl'] x 64 lanes x 8
    float* fold = nullptr;          // the folded BN table of net_fold() with every conv layer's scale divided by that layer's weight scale
};
FxNet* fx(azr_engine* h) { return reinterpret_cast<FxNet*>(h->net.fxctx); }
size_t fx_stream_bytes(int blocks) { return (size_t)2 * blocks * LAYER_BYTES + (size_t)MAX_RING * STEP_BYTES; }   // + ring run-off
}  // namespace

namespace azr {

int net_fx_alloc(azr_engine* h)
{
    FxNet* x = new FxNet();
    h->net.fxctx = x;
    const size_t bytes = fx_stream_bytes(h->net.blocks);
    HIPCHK(h, hipMalloc((void**)&x->tower_wp, bytes));
    HIPCHK(h, hipMemsetAsync(x->tower_wp, 0, bytes, h->stream));
    HIPCHK(h, hipMalloc((void**)&x->fold, (14 + (size_t)2 * h->net.blocks * 2 * NF) * sizeof(float)));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_tower_fx<2>), hipFuncAttributeMaxDynamicSharedMemorySize, FX<2>::LDS_BYTES));
    return AZR_OK;
}

void net_fx_free(azr_engine* h)
{
    if (!h->net.fxctx) return;
    FxNet* x = fx(h);
    if (x->tower_wp) hipFree(x->tower_wp);
    if (x->fold) hipFree(x->fold);
    delete x;
    h->net.fxctx = nullptr;
}

// pack the HWIO fp32 kernels of the AZRW vector into fp16 pairs in MFMA operand fragment order:
// fragment (layer, tap, ks, half, ctile), lane l, element j  <-  pair(s_L * W[tap][ci = ks*32 + 8*(l>>4) + j][co = ctile*16 + (l&15)])
// s_L = the power of two that brings the layer's largest |w| into [2^13, 2^14): the low part of an fp16 pair is worth 11 more bits only
// while it is a NORMAL fp16 number (>= 6.1e-5), and the low parts of unscaled Glorot weights of +-0.036 (|lo| <= 1.8e-5) are all
// subnormal: they keep 2^-24 absolute, i.e. 13 - 14 bits of the weight in all — measured: 1.3e-4 in pi / v at B = 20 against 3e-6 with
// the scale.  (The matrix core does NOT flush subnormal operands: profiles/r03_mfma_round_probe.txt; the loss is the subnormals' own
// precision.)  With the scale a weight down to 2^-11 of the layer's largest keeps a normal low part.  1 / s_L goes into the layer's
// folded BN scale: powers of two, exact.  Activations are not scaled: their low parts are subnormal below 2^-3 and then carry 2^-25
// absolute, the size of rounding the activation to fp32 (header).
// fold_host = the folded BN table of net_upload (stem scale[7] shift[7]; per conv layer scale[256] shift[256]).
int net_fx_upload(azr_engine* h, const float* fold_host)
{
    FxNet* x = fx(h);
    const int B = h->net.blocks;
    const float* flat = h->flat.data();
    const size_t layer_floats = (size_t)9 * NF * NF + 4 * NF;
    const float* t0 = flat + 9 * 13 * NF + 28;
    const size_t layer_halfs = LAYER_BYTES / 2, kstep_halfs = KBYTES / 2;
    std::vector<uint16_t> tower((size_t)2 * B * layer_halfs);
    std::vector<float> fold(fold_host, fold_host + 14 + (size_t)2 * B * 2 * NF);
    for (int L = 0; L < 2 * B; L++) {
        const float* W = t0 + (size_t)L * layer_floats;
        float worst = 0.0f;
        for (size_t i = 0; i < (size_t)9 * NF * NF; i++) {
            const float aw = W[i] < 0 ? -W[i] : W[i];
            if (!(aw <= worst)) worst = aw;   // (also catches NaN)
        }
        if (!(worst < 65504.0f)) { h->err = "NET_F32X: a conv weight is outside the fp16 range (|w| must be < 65504) or not a number"; return AZR_E_INVALID_ARGUMENT; }
        int e = 0;
        if (worst > 0.0f) {
            int we;
            frexpf(worst, &we);           // worst = f * 2^we, f in [0.5, 1)
            e = 14 - we;                  // worst * 2^e in [2^13, 2^14)
            if (e > 24) e = 24;
            if (e < -2) e = -2;
        }
        const float sL = ldexpf(1.0f, e), inv = ldexpf(1.0f, -e);
        float* fs = fold.data() + 14 + (size_t)L * 2 * NF;
        for (int i = 0; i < NF; i++) fs[i] *= inv;
        uint16_t* dst = tower.data() + (size_t)L * layer_halfs;
        for (int tap = 0; tap < 9; tap++)
            for (int ks = 0; ks < 8; ks++)
                for (int ct = 0; ct < 16; ct++)
                    for (int l = 0; l < 64; l++) {
                        const int ci0 = ks * 32 + 8 * (l >> 4), co = ct * 16 + (l & 15);
                        uint16_t* dh = dst + ((size_t)(tap * 8 + ks) * 2) * kstep_halfs + ((size_t)ct * 64 + l) * 8;
                        uint16_t* dl = dh + kstep_halfs;
                        for (int j = 0; j < 8; j++) {
                            const float w = W[((size_t)tap * NF + ci0 + j) * NF + co] * sL;
                            const _Float16 hi = (_Float16)w;
                            const _Float16 lo = (_Float16)(w - (float)hi);
                            memcpy(dh + j, &hi, 2);
                            memcpy(dl + j, &lo, 2);
                        }
                    }
    }
    HIPCHK(h, hipMemcpyAsync(x->tower_wp, tower.data(), tower.size() * 2, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(x->fold, fold.data(), fold.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return AZR_OK;
}

int net_fx_forward(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st)
{
    FxNet* x = fx(h);
    const int wgs = (n + 1) / 2;
    if (h->pe_tower0) hipEventRecord(h->pe_tower0, st);
    hipLaunchKernelGGL(k_tower_fx<2>, dim3(wgs), dim3(THREADS), FX<2>::LDS_BYTES, st, d_in88, in_stride, n, h->net.stem_w, x->tower_wp, x->fold,
                       h->net.blocks, net_head_params(h), d_pi, d_v, d_map);
    if (h->pe_tower1) hipEventRecord(h->pe_tower1, st);
    HIPCHK(h, hipGetLastError());
    return AZR_OK;
}

}  // namespace azr
