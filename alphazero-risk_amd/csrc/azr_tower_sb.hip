// azr_tower_sb.hip — k_tower_sb<NB>: the whole policy/value net (python/src/build_graph.py:63-90) for NB = 2, 3 or 4 boards per
// workgroup with ONE LDS activation image (gfx950).
//
// Why a second tower kernel.  k_tower_bf16 (azr_net_bf16.hip) keeps two ping-pong activation images in LDS, which caps a
// workgroup at 3 boards (M = 128 GEMM rows), and its 8 waves x 32 channels read every activation fragment for 2 MFMAs only.
// Here a workgroup owns up to 4 boards (168 cells, 11 MFMA row tiles) in a SINGLE LDS image:
//   * a layer's whole output lives in accumulator registers until the layer's last k-step (it always did); the image is
//     overwritten in place between two barriers (all waves done reading | epilogue stores | all stores visible);
//   * the residual input of a block — the values this wave itself stored two layers earlier — stays packed (bf16) in
//     registers instead of in a second image (at 4 boards 6 of the 11 row tiles keep theirs in the LDS the single image
//     leaves free: 176 accumulators + 88 residual registers + ring + fragments overflow the 512-register file);
//   * 4 waves (one per SIMD, up to 512 registers each) split the 256 output channels, 64 (four 16-wide tiles) each, so an
//     activation fragment read from LDS feeds 4 MFMAs: LDS fragment traffic per MFMA is half that of the 8-wave tiles;
//   * per k-step a wave issues, tile by tile, [4 MFMAs | 1 ds_read_b128 of that tile's NEXT k-step fragment]; the 4 buffer
//     loads that refill the weight-ring slot freed by the previous k-step, and the next tap's row-table lookups, are dealt
//     out over the k-step, one per MFMA gap (ring of 4 slots, 3 k-steps ahead, never drained; the packed stream of all
//     layers is contiguous, exactly as for k_tower_bf16 — both kernels read the same packed weights);
//   * the layer loop is rolled (conv1 and conv2 of a block share the code; only the epilogue looks at the parity) and a
//     layer's 72 k-steps are fully unrolled with compile-time skip masks: ~40 KB of straight-line code, inside the
//     64 KB instruction cache;
//   * a layer's folded BN scale / shift reach the epilogue through LDS (2 registers per lane over the k-steps, not 32).
// Row order (border classes with the corners in the column classes, azr_rowclass.hpp): 18 of 99 / 12 of 72 / 9 of 54
// (tile, tap) pairs are entirely out of board and are skipped.
// Arithmetic is the k_tower_bf16 arithmetic — same k order, same fp32 epilogue, same RNE points — so results are
// bit-identical across tile shapes (tests/test_gpu_net.py checks it).
#include <stdlib.h>
#include <string.h>

#include <string>

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

namespace {
constexpr int NT = 4;                  // 16-channel column tiles per wave (4 waves x 64 channels)
constexpr int WAVES = 4, THREADS = 256;

// Geometry of a workgroup of NB boards (NB = 4: the product tile; NB = 2: the tile for 257..512 boards per launch).
template <int NB_>
struct SB {
    static constexpr int NB = NB_;                       // boards per workgroup
    static constexpr int ROWS = 42 * NB;                 // board cells
    static constexpr int MT = (ROWS + 15) / 16;          // 16-row MFMA tiles: 11 (8 pad rows) / 6 (12 pad rows)
    static constexpr int ZR = MT * 16;                   // index of the shared zero row
    static constexpr int RING = 4;                         // weight ring depth in k-steps (72 = 0 mod RING)
    // LDS map
    static constexpr int BUF = (ZR + 1) * ROWB;                 // the activation image incl. its zero row     96 288 B at NB = 4
    static constexpr int FEAT_OFF = BUF;                        // stem features [(ZR + 1)][16 bf16]             5 664 B
    static constexpr int HEAD_OFF = FEAT_OFF + (ZR + 1) * FROWB;  // heads scratch: NB x (128 + 256 + 64) floats 7 168 B
    static constexpr int IN88_OFF = HEAD_OFF + NB * 448 * 4;    // NB x 96 B NNInputData images
    static constexpr int ROWOF_OFF = IN88_OFF + NB * 96;        // u8 [ZR]: cell (board * 42 + pos) -> row
    static constexpr int TAPROW_OFF = ROWOF_OFF + ZR;           // u8 [10][ZR]: source row of (tap, row); tap 9 = all zero row
    static constexpr int ROWCELL_OFF = TAPROW_OFF + 10 * ZR;    // u16 [ZR]: row -> y | x << 4 | board << 8, 0xffff = pad row
    static constexpr int FOLD_OFF = (ROWCELL_OFF + 2 * ZR + 15) / 16 * 16;   // float [2][256]: the current layer's folded BN scale | shift
    // the residual operand of the first RES_LDS row tiles lives here instead of in registers (NB = 4 only: 176 accumulators +
    // 88 residual registers + ring + fragments do not fit the 512-register file without spills): uint2 [RES_LDS * NT][256 lanes]
    static constexpr int RES_LDS = NB == 4 ? 6 : 0;
    static constexpr int RES_OFF = FOLD_OFF + 2 * NF * 4;
    static constexpr int LDS_BYTES = RES_OFF + RES_LDS * NT * THREADS * 8;
    static_assert(NB >= 2 && NB <= 4, "tile shapes");
    static_assert(72 % RING == 0 && RING <= MAX_RING, "ring depth");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(BUF % 16 == 0 && FEAT_OFF % 16 == 0 && HEAD_OFF % 16 == 0 && FOLD_OFF % 16 == 0 && TAPROW_OFF % 4 == 0 && ROWCELL_OFF % 2 == 0, "alignment");
    static_assert(3 * NF * 4 <= (ZR + 1) * FROWB, "the heads stage 3 x 256 floats in the stem feature image");
};

// s_waitcnt lgkmcnt(N) alone (vmcnt / expcnt untouched)
template <int N>
__device__ __forceinline__ void wait_lgkm()
{
    if constexpr (N >= 0 && N <= 14) __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
}

__device__ __forceinline__ s16x8 lds16(const uint8_t* p) { return *reinterpret_cast<const s16x8*>(p); }
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;

// Layer epilogue for 4 consecutive channels of one board cell: folded BN (fp32 fma), optional shortcut add (the packed
// bf16 block input), ReLU, round-to-nearest-even to bf16.  Packed forms: v_pk_fma_f32 / v_pk_add_f32 / v_cvt_pk_bf16_f32 /
// v_pk_max_i16 — ReLU is applied to the ROUNDED value as a signed 16-bit max with 0 (rounding is monotonic and odd, so
// relu(rne(v)) == rne(relu(v)); -0 becomes +0 like `v > 0 ? v : 0`).
template <bool SHORTCUT, bool F16>
__device__ __forceinline__ uint2 bn_relu_pack(const f32x4& acc, const float4& s, const float4& h, const uint2& x)
{
    // (the two instantiations are kept apart on purpose: merged by the optimiser, the common fma of all 44 tiles is hoisted
    //  above the parity branch and 176 intermediate values have to be parked)
    f32x4 t = acc;
    if (SHORTCUT) asm volatile("; epilogue with shortcut" : "+v"(t));
    else asm volatile("; epilogue" : "+v"(t));
    f32x2 lo = __builtin_elementwise_fma(f32x2{t[0], t[1]}, f32x2{s.x, s.y}, f32x2{h.x, h.y});
    f32x2 hi = __builtin_elementwise_fma(f32x2{t[2], t[3]}, f32x2{s.z, s.w}, f32x2{h.z, h.w});
    if (SHORTCUT) {
        lo += f32x2{El<F16>::lo_of(x.x), El<F16>::hi_of(x.x)};
        hi += f32x2{El<F16>::lo_of(x.y), El<F16>::hi_of(x.y)};
    }
    return uint2{El<F16>::pack_relu(lo), El<F16>::pack_relu(hi)};
}

// One tap (8 k-steps of 32 input channels) of a 3x3 conv layer for the 11 row tiles x 4 column tiles of a wave.  Everything
// that depends on the tap — which tiles run (skip masks), ring slots, wait counts — is a compile-time constant.
//
// Issue order.  A lone wave issues one instruction per ~4 cycles and an MFMA occupies the issue port for 8 of its 16 cycles:
// whatever else sits in one MFMA gap beyond ~8 cycles delays the matrix pipe.  So the non-MFMA work of a k-step is dealt out
// one piece per gap instead of in bursts:
//   k-step start:  one counted vmcnt wait for all four weight fragments of this k-step (the loads of the two k-steps in
//                  flight behind them stay outstanding)
//   per tile:      M0 | [one refill load] | M1 | [wait for the NEXT tile's fragment] | M2 | M3 | re-read of this tile's
//                  fragment for the next k-step
// The compiler's own waitcnt insertion still runs afterwards and stays the safety net: an explicit wait only moves a wait
// to an earlier, cheaper place.
template <int NB, int TAP, bool F16>
__device__ __forceinline__ void conv_tap(const uint8_t* bufX, const uint8_t* tr_c, uint32_t g16, const __amdgpu_buffer_rsrc_t wsrc,
                                         uint32_t loff, uint32_t& wk, u32x4 (&bq)[SB<NB>::RING][NT], f32x4 (&acc)[SB<NB>::MT][NT],
                                         s16x8 (&a)[SB<NB>::MT], uint32_t (&ap)[SB<NB>::MT])
{
    constexpr int MT = SB<NB>::MT, ZR = SB<NB>::ZR, RING = SB<NB>::RING;
    constexpr uint32_t sk = skip_mask<NB>(TAP), skn = skip_mask<NB>(TAP + 1);
    constexpr int active = MT - __builtin_popcount(sk & ((1u << MT) - 1u));   // tiles that run this tap
    uint32_t np[MT];
#pragma unroll
    for (int ks = 0; ks < KS_PER_TAP; ks++) {
        const int gk = TAP * KS_PER_TAP + ks;                       // k-step inside the layer (72 = 0 mod RING)
        const int cur = gk % RING, ref = (gk + RING - 1) % RING;    // ring slot in use / slot freed by the previous k-step
        {   // vmcnt((RING - 2) * NT): this k-step's four fragments have landed, the younger k-steps stay in flight
            constexpr int VM = (RING - 2) * NT;
            __builtin_amdgcn_s_waitcnt(0x0F70 | (VM & 15) | ((VM >> 4) << 14));   // lgkmcnt / expcnt untouched
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            if (!((sk >> mt) & 1u)) {
                const int j = __builtin_popcount(~sk & ((1u << mt) - 1u));   // index among the active tiles
                acc[mt][0] = El<F16>::mfma(bq[cur][0], a[mt], acc[mt][0]);
                // refill of the slot the previous k-step freed, RING - 1 k-steps ahead: one of its 4 loads after the first
                // MFMA of 4 tiles spread over the k-step (also across layer boundaries)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    if (j == ((nt + 1) * active) / NT - 1)
                        bq[ref][nt] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, loff + nt * 1024, (int)(wk + (RING - 1) * KBYTES), 0);
                __builtin_amdgcn_sched_barrier(0);
                acc[mt][1] = El<F16>::mfma(bq[cur][1], a[mt], acc[mt][1]);
                wait_lgkm<active - 2>();                            // the next tile's fragment was requested `active - 1` reads ago
                __builtin_amdgcn_sched_barrier(0);
                acc[mt][2] = El<F16>::mfma(bq[cur][2], a[mt], acc[mt][2]);
                acc[mt][3] = El<F16>::mfma(bq[cur][3], a[mt], acc[mt][3]);
                if (ks < KS_PER_TAP - 1) a[mt] = lds16(bufX + ap[mt] + (ks + 1) * 64);
            }
            // the next tap's source rows, dealt out like everything else: one table byte per tile slot two k-steps before
            // the tap ends, its address arithmetic one k-step later (as a burst in front of a k-step they cost ~3 % of a layer)
            if (ks == KS_PER_TAP - 3) { if (!((skn >> mt) & 1u)) np[mt] = (uint32_t)tr_c[(TAP + 1) * ZR + mt * 16]; }
            if (ks == KS_PER_TAP - 2) { if (!((skn >> mt) & 1u)) np[mt] = np[mt] * ROWB + g16; }
            if (ks == KS_PER_TAP - 1) { if (!((skn >> mt) & 1u)) a[mt] = lds16(bufX + np[mt]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        wk += (uint32_t)KBYTES;
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) ap[mt] = np[mt];
}

template <int NB, bool F16>
__global__ __launch_bounds__(THREADS, 1) void k_tower_sb(const uint8_t* __restrict__ in88, int in_stride, int n,
                                                          const uint16_t* __restrict__ stem_wp, const uint16_t* __restrict__ tower_wp,
                                                          const float* __restrict__ fold, int blocks, const float* __restrict__ hp,
                                                          float* __restrict__ pi_out, float* __restrict__ v_out,
                                                          unsigned long long* __restrict__ diag, const int* __restrict__ slot_map)
{
    using G = SB<NB>;
    constexpr int ROWS = G::ROWS, MT = G::MT, ZR = G::ZR, RING = G::RING;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* bufX = lds;
    uint8_t* bufF = lds + G::FEAT_OFF;
    uint8_t* in_l = lds + G::IN88_OFF;
    uint8_t* rowof = lds + G::ROWOF_OFF;
    uint8_t* taprow = lds + G::TAPROW_OFF;
    uint16_t* rowcell = reinterpret_cast<uint16_t*>(lds + G::ROWCELL_OFF);
    float* foldl = reinterpret_cast<float*>(lds + G::FOLD_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;        // MFMA fragment coordinates: board cell (column) c of a tile, k-group g
    const int board0 = blockIdx.x * NB;
    // diagnostics (azr_debug_tower_clock / azr_debug_tower_trace; `diag` is null in every product launch): workgroup 0's
    // shader-clock / real-time stamps around the tower, and per workgroup 8 words from diag[8 + 8 * blockIdx]: real-time
    // at kernel start, tower start, tower end, kernel end, and the XCC the workgroup ran on
    if (diag && blockIdx.x == 0 && tid == 0) { diag[0] = __builtin_amdgcn_s_memtime(); diag[1] = __builtin_amdgcn_s_memrealtime(); }
    if (diag && tid == 0) {
        diag[8 + 8 * (size_t)blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        diag[8 + 8 * (size_t)blockIdx.x + 4] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11)) & 15u;
    }

    // ---- weight ring: the first RING - 1 k-steps of layer 0 fly while the tables and the stem are built
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(tower_wp), (short)0, 0x7fffffff, 0x00020000);
    const uint32_t loff = (uint32_t)((wave * NT) * 64 + lane) * 16u;   // this lane's fragment bytes inside a k-step block
    uint32_t wk = 0;                                                     // byte offset of the current k-step (wave-uniform)
    u32x4 bq[RING][NT];
#pragma unroll
    for (int ks = 0; ks < RING - 1; ks++)   // (slot RING - 1 is filled during k-step 0, and so on: the refill of a slot is
                                            //  issued during the k-step after the one that used it)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) bq[ks][nt] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, loff + nt * 1024, ks * (int)KBYTES, 0);

    // ---- stage the NNInputData images, build the row tables
    for (int i = tid; i < NB * 96; i += THREADS) {
        const int b = i / 96, o = i % 96;
        const int slot = (board0 + b < n) ? (slot_map ? slot_map[board0 + b] : board0 + b) : 0;
        in_l[i] = (board0 + b < n && o < 88) ? in88[(size_t)slot * in_stride + o] : (uint8_t)0;
    }
    for (int i = tid; i < ROWB / 4; i += THREADS) reinterpret_cast<uint32_t*>(bufX + ZR * ROWB)[i] = 0;
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
    // stem features: bufF as [ZR + 1][16] bf16 (row ZR = zero row); planes 13..15 are zero
    for (int i = tid; i < (ZR + 1) * 16; i += THREADS) {
        const int r = i >> 4, ch = i & 15;
        float v = 0.0f;
        const int ci = r < ZR ? rowcell[r] : 0xffff;
        if (ci != 0xffff) v = plane_value(in_l + (ci >> 8) * 96, (ci & 15) * 6 + ((ci >> 4) & 15), ch);
        reinterpret_cast<uint16_t*>(bufF)[i] = El<F16>::rne(v);
    }
    __syncthreads();

    f32x4 acc[MT][NT];
    // the block input of this wave's (cell, 4-channel) elements, packed bf16 = the residual operand: tiles RL.. in registers,
    // tiles 0..RL-1 in LDS (one conflict-free 8-byte slot per lane, tile and column tile)
    constexpr int RL = G::RES_LDS;
    uint2 res[MT - RL][NT];
    uint2* resl = reinterpret_cast<uint2*>(lds + G::RES_OFF) + tid;
    const uint32_t eoff = (uint32_t)(c * ROWB + (wave * 64 + g * 4) * 2);   // epilogue store address of tile 0 / column tile 0

    // ---- stem: 3x3 conv 13 -> 256, two taps per 32-deep k-step (tap = 2*ks + (g >> 1), channels (g & 1)*8 ..), weights as
    //      the MFMA "A" operand: D[channel][cell], a lane ends up with 4 consecutive channels of one board cell
    {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0, 0, 0, 0};
        const s16x8* wp = reinterpret_cast<const s16x8*>(stem_wp) + (size_t)(wave * NT) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < STEM_KS; ks++) {
            const int tap = 2 * ks + (g >> 1);   // 0..9; tap 9 (second half of the last k-step) is the zero row
            s16x8 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b[nt] = wp[(size_t)ks * FRAGS_PER_KSTEP * 64 + nt * 64];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int row = taprow[tap * ZR + mt * 16 + c];
                const s16x8 av = lds16(bufF + row * FROWB + (g & 1) * 16);
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc[mt][nt] = El<F16>::mfma(b[nt], av, acc[mt][nt]);
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
                const uint2 o = bn_relu_pack<false, F16>(acc[mt][nt], float4{sc, sc, sc, sc}, float4{sh, sh, sh, sh}, uint2{0, 0});
                if (mt < RL) resl[(mt * NT + nt) * THREADS] = o; else res[mt < RL ? 0 : mt - RL][nt] = o;
                if (c < pad_from<NB>(mt)) *reinterpret_cast<uint2*>(bufX + eoff + mt * 16 * ROWB + nt * 32) = o;
            }
        }
    }
    __syncthreads();

    if (diag && tid == 0) { diag[8 + 8 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); diag[8 + 8 * (size_t)blockIdx.x + 5] = __builtin_amdgcn_s_memtime(); }
    // ---- residual tower: 2B conv layers, activations resident in the one LDS image
    const uint8_t* tr_c = taprow + c;           // this lane's column of the (tap, row) -> source-row table
    const uint32_t g16 = (uint32_t)g * 16u;
    const int layers = 2 * blocks;
    for (int L = 0; L < layers; L++) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0, 0, 0, 0};
        const float2 fnext = *reinterpret_cast<const float2*>(fold + 14 + (size_t)L * 2 * NF + 2 * tid);   // (see the epilogue)
        uint32_t ap[MT];        // LDS byte address of this lane's fragment of tile mt at k-step 0 of the current tap
        s16x8 a[MT];            // ... and the fragment of the k-step about to run
        {
            const uint32_t sk0 = skip_mask<NB>(0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                ap[mt] = (uint32_t)tr_c[mt * 16] * ROWB + g16;
                if (!((sk0 >> mt) & 1u)) a[mt] = lds16(bufX + ap[mt]);
            }
        }
        conv_tap<NB, 0, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
        conv_tap<NB, 1, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
        conv_tap<NB, 2, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
        conv_tap<NB, 3, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
        conv_tap<NB, 4, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
        conv_tap<NB, 5, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
        conv_tap<NB, 6, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
        conv_tap<NB, 7, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
        conv_tap<NB, 8, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
        // this layer's folded BN scale / shift go through LDS: 2 registers per lane over the k-steps instead of 32 (which the
        // register allocator parked in scratch), written before the barrier, read back 16 bytes at a time after it
        *reinterpret_cast<float2*>(foldl + 2 * tid) = fnext;
        __syncthreads();        // every wave has read the image for the last time
        float4 sc[NT], sh[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            sc[nt] = *reinterpret_cast<const float4*>(foldl + wave * 64 + g * 4 + nt * 16);
            sh[nt] = *reinterpret_cast<const float4*>(foldl + NF + wave * 64 + g * 4 + nt * 16);
        }
        if (L & 1) {    // second conv of a block: + shortcut (the block's input, kept packed in registers), and this
                        // output is the next block's input
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const uint2 x = mt < RL ? resl[(mt * NT + nt) * THREADS] : res[mt < RL ? 0 : mt - RL][nt];
                    const uint2 o = bn_relu_pack<true, F16>(acc[mt][nt], sc[nt], sh[nt], x);
                    if (mt < RL) resl[(mt * NT + nt) * THREADS] = o; else res[mt < RL ? 0 : mt - RL][nt] = o;
                    if (c < pad_from<NB>(mt)) *reinterpret_cast<uint2*>(bufX + eoff + mt * 16 * ROWB + nt * 32) = o;
                }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const uint2 o = bn_relu_pack<false, F16>(acc[mt][nt], sc[nt], sh[nt], uint2{0, 0});
                    if (c < pad_from<NB>(mt)) *reinterpret_cast<uint2*>(bufX + eoff + mt * 16 * ROWB + nt * 32) = o;
                }
        }
        __syncthreads();        // the new image is complete
    }

    if (diag && blockIdx.x == 0 && tid == 0) { diag[2] = __builtin_amdgcn_s_memtime(); diag[3] = __builtin_amdgcn_s_memrealtime(); }
    if (diag && tid == 0) { diag[8 + 8 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime(); diag[8 + 8 * (size_t)blockIdx.x + 6] = __builtin_amdgcn_s_memtime(); }
    // ---- both heads (build_graph.py:76-90; the arithmetic of k_tower_bf16's fused heads, same order)
    {
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
        // 1x1 convs (256 -> 2 policy + 1 value channel per cell): the three weight columns are staged in LDS (the stem's
        // feature image is free by now) and the activations are read 8 channels at a time — the same fma chain over
        // ci = 0..255 as before, so the same bits, without 512 two-byte LDS reads and 512 global loads per output
        float* wl = reinterpret_cast<float*>(bufF);              // [3][256]
        for (int i = tid; i < 3 * NF; i += THREADS) wl[i] = i < 2 * NF ? wpi[(i & (NF - 1)) * 2 + (i >> 8)] : wv[i - 2 * NF];
        __syncthreads();
        for (int idx = tid; idx < NB * 126; idx += THREADS) {  // 42 cells x {pi0, pi1, v} per board
            const int bb = idx / 126, t = idx % 126, pos = t / 3, ch = t % 3;
            const s16x8* x8 = reinterpret_cast<const s16x8*>(bufX + rowof[bb * 42 + pos] * ROWB);
            const float4* w4 = reinterpret_cast<const float4*>(wl + ch * NF);
            float sacc = 0.0f;
            for (int q = 0; q < NF / 8; q++) {
                const s16x8 xx = x8[q];
                const float4 wa = w4[2 * q], wb = w4[2 * q + 1];
                sacc = fmaf(El<F16>::tof((uint16_t)xx[0]), wa.x, sacc); sacc = fmaf(El<F16>::tof((uint16_t)xx[1]), wa.y, sacc);
                sacc = fmaf(El<F16>::tof((uint16_t)xx[2]), wa.z, sacc); sacc = fmaf(El<F16>::tof((uint16_t)xx[3]), wa.w, sacc);
                sacc = fmaf(El<F16>::tof((uint16_t)xx[4]), wb.x, sacc); sacc = fmaf(El<F16>::tof((uint16_t)xx[5]), wb.y, sacc);
                sacc = fmaf(El<F16>::tof((uint16_t)xx[6]), wb.z, sacc); sacc = fmaf(El<F16>::tof((uint16_t)xx[7]), wb.w, sacc);
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
    if (diag && tid == 0) diag[8 + 8 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace

namespace azr {

template <int NB, bool F16>
static int sb_attr(azr_engine* h)
{
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_tower_sb<NB, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, SB<NB>::LDS_BYTES));
    return AZR_OK;
}
int tower_sb_init(azr_engine* h)
{
    int rc = bf16net(h)->f16 ? (sb_attr<4, true>(h) || sb_attr<3, true>(h) || sb_attr<2, true>(h)) : (sb_attr<4, false>(h) || sb_attr<3, false>(h) || sb_attr<2, false>(h));
    return rc ? AZR_E_HIP : AZR_OK;
}

template <int NB, bool F16>
static void sb_go(azr_engine* h, int wgs, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st)
{
    Bf16Net* x = bf16net(h);
    hipLaunchKernelGGL((k_tower_sb<NB, F16>), dim3(wgs), dim3(THREADS), SB<NB>::LDS_BYTES, st, d_in88, in_stride, n, x->stem_wp, x->tower_wp,
                       F16 ? (const float*)x->fold16 : net_fold(h), h->net.blocks, net_head_params(h), d_pi, d_v, x->diag, d_map);
}

// `wgs` workgroups of nb (2, 3 or 4) boards: boards [0, n) of the launch (the last workgroup may be partly filled)
int tower_sb_launch(azr_engine* h, int nb, int wgs, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st)
{
    if (nb < 2 || nb > 4 || wgs < 1 || (long long)wgs * nb < n) { h->err = "tower_sb_launch: bad tiling"; return AZR_E_INVALID_ARGUMENT; }
    if (bf16net(h)->f16) {
        if (nb == 4) sb_go<4, true>(h, wgs, d_in88, in_stride, n, d_pi, d_v, d_map, st);
        else if (nb == 3) sb_go<3, true>(h, wgs, d_in88, in_stride, n, d_pi, d_v, d_map, st);
        else sb_go<2, true>(h, wgs, d_in88, in_stride, n, d_pi, d_v, d_map, st);
    } else {
        if (nb == 4) sb_go<4, false>(h, wgs, d_in88, in_stride, n, d_pi, d_v, d_map, st);
        else if (nb == 3) sb_go<3, false>(h, wgs, d_in88, in_stride, n, d_pi, d_v, d_map, st);
        else sb_go<2, false>(h, wgs, d_in88, in_stride, n, d_pi, d_v, d_map, st);
    }
    HIPCHK(h, hipGetLastError());
    return AZR_OK;
}

}  // namespace azr
