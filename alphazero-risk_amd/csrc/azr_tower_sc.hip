// azr_tower_sc.hip — k_tower_sc: the bf16 net for launches of up to 128 boards with the OUTPUT CHANNELS of a board pair split over
// 4 workgroups of one persistent launch (gfx950).
//
// Why.  One workgroup that owns a board for the whole net (k_tower_bf16<1>) streams all 47.3 MB of packed weights through ONE CU:
// 0.45 - 0.48 ms per pass whatever the batch, bound by that CU's L1 (16 KB of weight fragments per k-step at <= 64 B/clk) with most
// of the chip idle when fewer than 256 boards wait — the regime of the 100-game arena, the benchmark games, `-m play`, every quota
// tail, and of a rank's share of them on 8 GPUs (game/game.cpp:277-312; alphazero_trainer.cpp:121-132,147).  Cutting the net into
// one launch per layer made it slower (profiles/r03_layer_by_layer_experiment.txt).  Here the launch stays persistent — every
// workgroup runs all layers with a weight ring that never restarts — and a pair of boards is shared by CGN = 4 workgroups that each
// compute 64 of the 256 output channels: a workgroup streams a quarter of the weights.
//
// Inside a workgroup the 96 x 64 tile of a layer is split 2 x 2 over the four waves (3 row tiles x 2 column tiles each): per k-step
// the CU reads 12 KB of activations from LDS (96 clocks), 8 KB of weights through L1 (128 clocks) and issues 24 MFMAs (96 clocks per
// SIMD) — the L1 term binds, 3.7 us per layer measured (1 x 4, every wave on all 96 rows, read 24 KB of LDS per k-step: 5.8 us).
//
// Every layer ends with an all-to-all among the pair's workgroups, through memory:
//     epilogue -> own channels of the new image: write-through (sc1) stores to the exchange image of this layer's parity
//     every wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds 1 to the pair's counter (agent scope)
//     own channels -> LDS image; the next layer's first weight fragments are requested;  one lane polls the counter until all CGN
//     workgroups of the pair have arrived at this layer (relaxed sc1 loads, s_sleep, bounded: a spin that runs out raises the
//     launch's GIVE-UP word in device memory — the launch then ends with garbage, never hangs, and the guarded k_tower_bf16<1> launch
//     queued right behind it on the same stream recomputes the whole batch, see tower_sc_launch), workgroup barrier, the partners'
//     channels are fetched with sc1 loads
//     (they never hit this CU's L1: no acquire fence needed for write-through, drained stores) into the LDS image, barrier.
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": the drained-sc1-stores + counter form.)
// 3.2 us per layer; where the four workgroups of a pair run on one XCD (checked at run time through HW_REG_XCC_ID; the block ids are laid
// out for it) the layer images stay in that XCD's L2 — plain stores instead of write-through ones, 3 - 4 % of a launch.  Tried and left out (profiles/r03_small_batch_tower.txt): a data-is-the-flag form (epoch tag in the sign bits of
// the post-ReLU bf16, no drain / counter / poll) — 4.1 us, the first sweep always comes too early and a sweep is a full round trip;
// a fifth wave touching the next layers' weights into L2 — slower, the L1
// path is the bound and the touches double its traffic.
// Exchange images are double-buffered by layer parity: to overwrite parity p a workgroup must have finished layer L + 1, which
// needed every partner's layer-L + 1 slice, which they produced after reading layer L — no reader can be behind.  All workgroups of
// a launch must be co-resident: <= 256 workgroups of 64 KB LDS and <= 256 VGPRs (two fit on a CU, at 0.66 ms instead of 0.33: the two networks
// of an arena, side by side, take this kernel only while their workgroups have a CU each — n_other below).
// Block ids are laid out so that a pair's workgroups are 8 ids apart (same XCD under round-robin dispatch:
// speed only).  Arithmetic = k_tower_sb<2>'s: same packed weights, row order, skipped (tile, tap) pairs, k order, fp32 epilogue and
// RNE points — bit-identical results (tests/test_gpu_net.py::test_tile_shapes_agree_bit_for_bit).
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
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;

namespace {
constexpr int NB = 2, ROWS = 84, MT = 6, ZR = 96, THREADS = 256;
constexpr int MAX_PAIRS = 64;                        // board pairs of one launch (128 boards)
constexpr int CUS = 256;                              // compute units of the part (MI355X)
constexpr int CGN = 4;                                // workgroups per board pair: 64 output channels each
constexpr int MTW = 3, NTW = 2;                       // a wave's tile: 3 row tiles (one half of the pair's 96 rows) x 2 column tiles (32 channels)
constexpr int RING = 12;                              // weight ring depth in k-steps (72 = 0 mod RING): 22 KB in flight per wave
constexpr int IMG = (ZR + 1) * ROWB;                  // the pair's activation image in LDS, incl. the zero row
constexpr uint32_t EX_PAIR_BYTES = ZR * NF * 2;       // one pair's exchange image: [96 rows][256] bf16, no pad
constexpr int AUX_SC1 = 16;                           // cache-policy bits of the raw buffer builtins: sc1 (write-through store / L1-bypassing load)
constexpr unsigned SPIN_LIMIT = 1u << 21;             // polls of a hand-off before the launch gives up (~1 s)
constexpr int W_GIVEUP = 2 * MAX_PAIRS;               // word of the counter block: raised by a hand-off that ran out of polls: (serial of the launch << 2) | 1; | 2 = more boards than 128
constexpr int W_FALLBACKS = 2 * MAX_PAIRS + 1;        // ... launches the guarded one-board-per-workgroup kernel had to recompute (never zeroed)

// LDS map (dynamic): image | stem features | NNInputData images | tables | heads scratch
constexpr int FEAT_OFF = IMG;
constexpr int IN88_OFF = FEAT_OFF + (ZR + 1) * FROWB;
constexpr int ROWOF_OFF = IN88_OFF + NB * 96;
constexpr int TAPROW_OFF = ROWOF_OFF + ZR;
constexpr int ROWCELL_OFF = TAPROW_OFF + 10 * ZR;
constexpr int HEAD_OFF = (ROWCELL_OFF + 2 * ZR + 15) / 16 * 16;
constexpr int LDS_BYTES = HEAD_OFF + (3 * NF + NB * 128 + NB * 256 + NB * 64) * 4;
static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");

__device__ __forceinline__ s16x8 lds16(const uint8_t* p) { return *reinterpret_cast<const s16x8*>(p); }

// k_tower_sb's epilogue, verbatim: folded BN (fp32 fma), optional shortcut add (packed bf16 block input), ReLU on the rounded value
template <bool SHORTCUT, bool F16>
__device__ __forceinline__ uint2 bn_relu_pack(const f32x4& acc, const float4& s, const float4& h, const uint2& x)
{
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

// the row tiles of a wave half: 0 - 2 (the board-edge tiles, which skip 3 of their 9 taps each) and 3 - 5.  Uneven on purpose: dealing
// the edge tiles over both halves (24 and 21 tile-taps instead of 18 and 27) measured 3 - 5 % slower — a k-step is bound by the
// weight fragments through L1, not by the matrix pipe, and the half that finishes early leaves L1 to the other
template <int MH> __host__ __device__ constexpr int tile_of(int i) { return MH * 3 + i; }

// One tap of a layer for this wave's tile (three row tiles, two column tiles); compile-time skip masks and ring
// slots; wk = byte offset of the layer's k-step 0 in the packed stream.  A k-step is 6 MFMAs (96 cycles of the matrix pipe) against
// 3 LDS fragments and 2 weight fragments: with the rows split over wave pairs the workgroup reads 12 KB of LDS per k-step (96 clocks
// at 128 B/clk) and 8 KB through L1 (128 clocks) — with all four waves on all 96 rows it was 24 KB of LDS, 192 clocks, the bound.
// Everything that is not an MFMA is dealt out one piece per MFMA gap, and scheduling regions (sched_barrier) pin that order: left to
// itself the compiler sinks the refills until the ring has drained and then runs two loads deep.
template <int MH, int TAP, bool F16>
__device__ __forceinline__ void sc_tap(const uint8_t* bufX, const uint8_t* tr_c, uint32_t g16, const __amdgpu_buffer_rsrc_t wsrc, uint32_t loff, uint32_t wk,
                                       u32x4 (&bq)[RING][NTW], f32x4 (&acc)[MTW][NTW], s16x8 (&a)[MTW], uint32_t (&ap)[MTW])
{
    constexpr uint32_t m = skip_mask<NB>(TAP), mn = skip_mask<NB>(TAP + 1);
    constexpr uint32_t sk = ((m >> tile_of<MH>(0)) & 1u) | (((m >> tile_of<MH>(1)) & 1u) << 1) | (((m >> tile_of<MH>(2)) & 1u) << 2);
    constexpr uint32_t skn = ((mn >> tile_of<MH>(0)) & 1u) | (((mn >> tile_of<MH>(1)) & 1u) << 1) | (((mn >> tile_of<MH>(2)) & 1u) << 2);
    uint32_t np[MTW];
#pragma unroll
    for (int ks = 0; ks < KS_PER_TAP; ks++) {
        const int gk = TAP * KS_PER_TAP + ks, cur = gk % RING, ref = (gk + RING - 1) % RING;
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            if (!((sk >> i) & 1u)) {
#pragma unroll
                for (int nt = 0; nt < NTW; nt++)
                    acc[i][nt] = El<F16>::mfma(bq[cur][nt], a[i], acc[i][nt]);
                if (ks < KS_PER_TAP - 1) a[i] = lds16(bufX + ap[i] + (ks + 1) * 64);
            }
            if (ks == KS_PER_TAP - 3) { if (!((skn >> i) & 1u)) np[i] = (uint32_t)tr_c[(TAP + 1) * ZR + tile_of<MH>(i) * 16]; }
            if (ks == KS_PER_TAP - 2) { if (!((skn >> i) & 1u)) np[i] = np[i] * ROWB + g16; }
            if (ks == KS_PER_TAP - 1) { if (!((skn >> i) & 1u)) a[i] = lds16(bufX + np[i]); }
            // refill of the slot the previous k-step freed, RING - 1 k-steps ahead, one fragment per row tile — inside the layer only:
            // the first RING - 1 k-steps of the NEXT layer are requested after this layer's slice has been published (in front of the
            // stores they would sit in the wave's in-order memory counter and the publish would wait for them), and fly during the hand-off
            if (i < NTW && gk + RING - 1 < 9 * KS_PER_TAP)
                bq[ref][i] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, loff + i * 1024, (int)(wk + (uint32_t)(gk + RING - 1) * (uint32_t)KBYTES), 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < MTW; i++) ap[i] = np[i];
}

// The all-to-all at the end of a layer (epoch = 1 for the stem, 2 .. 2B + 1 for the conv layers).  The pieces that depend on the wave's
// row half (MH: which row tiles, which pad rows) contain NO barrier; the barriers, the count and the poll live in hand_off() below,
// code that all four waves of the workgroup execute at the same program counter.
// publish_stores: this wave's part o[i][nt] of the workgroup's channels goes to the exchange image of the epoch's parity.
template <int MH>
__device__ __forceinline__ void publish_stores(const uint2 (&o)[MTW][NTW], int epoch, int pair, int ct0, int c, int g, const __amdgpu_buffer_rsrc_t ex, bool same_xcd)
{
    const uint32_t img_off = (uint32_t)(pair * 2 + (epoch & 1)) * EX_PAIR_BYTES;
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        const int mt = tile_of<MH>(i);
        if (!(c < pad_from<NB>(mt))) continue;
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            const uint32_t off = img_off + (uint32_t)((mt * 16 + c) * NF + (ct0 + nt) * 16 + g * 4) * 2u;
            // all four workgroups of the pair on ONE XCD (checked, below): the image stays in that XCD's L2 — plain stores, drained the
            // same way; the partners' sc1 loads are served by the same L2.  Otherwise write-through, as the visibility rules demand.
            if (same_xcd) __builtin_amdgcn_raw_buffer_store_b64(u32x2{o[i][nt].x, o[i][nt].y}, ex, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b64(u32x2{o[i][nt].x, o[i][nt].y}, ex, off, 0, AUX_SC1);
        }
    }
}
// own_to_lds: the own channels go into the LDS image (nobody reads the old one any more: called behind hand_off's first barrier)
template <int MH>
__device__ __forceinline__ void own_to_lds(uint8_t* bufX, const uint2 (&o)[MTW][NTW], int ct0, int c, int g)
{
    const uint32_t eoff = (uint32_t)(c * ROWB + (ct0 * 16 + g * 4) * 2);
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        const int mt = tile_of<MH>(i);
        if (!(c < pad_from<NB>(mt))) continue;
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) *reinterpret_cast<uint2*>(bufX + eoff + mt * 16 * ROWB + nt * 32) = o[i][nt];
    }
}
// count_in: every storing wave has drained its stores (s_waitcnt vmcnt(0)) before this barrier; behind it ONE lane counts the workgroup in
__device__ __forceinline__ void count_in(int tid, unsigned* counter)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its stores
    __syncthreads();                                    // ... and every wave has read the old image for the last time
    if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// gather: ONE lane polls the pair's counter (relaxed agent loads, bounded), barrier, the partners' channels come into the LDS image
// through sc1 loads — loads that never hit this CU's L1, of bytes that were stored write-through and drained before the count: the
// form that needs no acquire fence (MI355X_MICROARCH.md, visibility section, valid forms, first row of the table).  Pad rows are
// stored by nobody and fetched by nobody.  A poll that runs out of spins raises the launch's give-up word (device memory, agent scope)
// and goes on with whatever the image holds; once the word is up nobody waits any more (the launch is lost: it only has to end).
__device__ __forceinline__ void gather(uint8_t* bufX, int epoch, int pair, int cg, int tid, const __amdgpu_buffer_rsrc_t ex, unsigned* counter, unsigned* giveup,
                                       unsigned tag, unsigned spin_limit)
{
    constexpr int UNITS = ZR * 32 / THREADS;   // 16-byte units per thread over the whole [96][512 B] image: rows (tid >> 5) + 8 i, segment tid & 31
    const uint32_t img_off = (uint32_t)(pair * 2 + (epoch & 1)) * EX_PAIR_BYTES;
    if (tid == 0) {
        const unsigned want = (unsigned)(CGN * epoch);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if (spins >= spin_limit) { __hip_atomic_store(giveup, tag | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if ((spins & 255u) == 255u && (__hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ tag) < 4u) break;
            spins++;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    const int seg = tid & 31, r0 = tid >> 5;
    const bool own = (seg >> 3) == cg;          // 8 segments (64 channels) of a 512-byte row are this workgroup's own
    bool act[UNITS];
#pragma unroll
    for (int i = 0; i < UNITS; i++) act[i] = !own && (r0 + 8 * (i & 1)) < pad_from<NB>(i >> 1);
    u32x4 st[UNITS];
#pragma unroll
    for (int i = 0; i < UNITS; i++)
        if (act[i]) st[i] = __builtin_amdgcn_raw_buffer_load_b128(ex, img_off + (uint32_t)(tid + THREADS * i) * 16u, 0, AUX_SC1);
#pragma unroll
    for (int i = 0; i < UNITS; i++)
        if (act[i]) *reinterpret_cast<u32x4*>(bufX + (r0 + 8 * i) * ROWB + seg * 16) = st[i];
    __syncthreads();
}

// ---- the two pieces of a wave's work that depend on its row half MH (compile-time tile lists and skip masks); no barrier inside
// stem: 3x3 conv 13 -> 256 (this wave's rows and channels), conv_bn over the board row + ReLU -> o (= the first block's input, res)
template <int MH, bool F16>
__device__ __forceinline__ void stem_half(const uint8_t* bufF, const uint8_t* taprow, const uint16_t* rowcell, const uint16_t* __restrict__ stem_wp,
                                          const float* __restrict__ fold, int ct0, int lane, int c, int g, f32x4 (&acc)[MTW][NTW], uint2 (&o)[MTW][NTW],
                                          uint2 (&res)[MTW][NTW])
{
#pragma unroll
    for (int i = 0; i < MTW; i++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) acc[i][nt] = f32x4{0, 0, 0, 0};
    const s16x8* wp = reinterpret_cast<const s16x8*>(stem_wp) + (size_t)ct0 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < STEM_KS; ks++) {
        const int tap = 2 * ks + (g >> 1);
        s16x8 b[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) b[nt] = wp[(size_t)ks * FRAGS_PER_KSTEP * 64 + nt * 64];
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            const int row = taprow[tap * ZR + tile_of<MH>(i) * 16 + c];
            const s16x8 av = lds16(bufF + row * FROWB + (g & 1) * 16);
#pragma unroll
            for (int nt = 0; nt < NTW; nt++)
                acc[i][nt] = El<F16>::mfma(b[nt], av, acc[i][nt]);
        }
    }
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        const int ci = rowcell[tile_of<MH>(i) * 16 + c];
        const int y = ci == 0xffff ? 0 : (ci & 15);
        const float sc = fold[y], sh = fold[7 + y];
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            o[i][nt] = bn_relu_pack<false, F16>(acc[i][nt], float4{sc, sc, sc, sc}, float4{sh, sh, sh, sh}, uint2{0, 0});
            res[i][nt] = o[i][nt];
        }
    }
}

// one conv layer of the tower for this wave's tile: 72 k-steps (the ring runs ahead inside the layer) and the epilogue -> o (and res
// behind the second conv of a block)
template <int MH, bool F16>
__device__ __forceinline__ void layer_half(const uint8_t* bufX, const uint8_t* tr_c, uint32_t g16, const __amdgpu_buffer_rsrc_t wsrc, uint32_t loff, uint32_t wk,
                                           bool second, const float4 (&sc)[NTW], const float4 (&sh)[NTW], u32x4 (&bq)[RING][NTW],
                                           f32x4 (&acc)[MTW][NTW], uint2 (&o)[MTW][NTW], uint2 (&res)[MTW][NTW])
{
#pragma unroll
    for (int i = 0; i < MTW; i++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) acc[i][nt] = f32x4{0, 0, 0, 0};
    uint32_t ap[MTW];
    s16x8 a[MTW];
    {
        constexpr uint32_t m0 = skip_mask<NB>(0);
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            ap[i] = (uint32_t)tr_c[tile_of<MH>(i) * 16] * ROWB + g16;
            if (!((m0 >> tile_of<MH>(i)) & 1u)) a[i] = lds16(bufX + ap[i]);
        }
    }
    sc_tap<MH, 0, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
    sc_tap<MH, 1, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
    sc_tap<MH, 2, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
    sc_tap<MH, 3, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
    sc_tap<MH, 4, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
    sc_tap<MH, 5, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
    sc_tap<MH, 6, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
    sc_tap<MH, 7, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
    sc_tap<MH, 8, F16>(bufX, tr_c, g16, wsrc, loff, wk, bq, acc, a, ap);
    if (second) {    // second conv of a block: + shortcut (the block's input), and this output is the next block's input
#pragma unroll
        for (int i = 0; i < MTW; i++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) { o[i][nt] = bn_relu_pack<true, F16>(acc[i][nt], sc[nt], sh[nt], res[i][nt]); res[i][nt] = o[i][nt]; }
    } else {
#pragma unroll
        for (int i = 0; i < MTW; i++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) o[i][nt] = bn_relu_pack<false, F16>(acc[i][nt], sc[nt], sh[nt], uint2{0, 0});
    }
}

// stem + residual tower of the workgroup.  Waves 0, 1 take the row tiles 0 - 2, waves 2, 3 the tiles 3 - 5 (mh; the per-half pieces
// above are selected by a wave-uniform branch); everything that synchronises the workgroup — count_in's and gather's barriers, the
// count, the poll — is HERE, in code common to all four waves.  false = this workgroup is done (not channel group 0).
template <bool F16>
__device__ __forceinline__ bool sc_run(uint8_t* lds, int pair, int cg, int blocks, const uint16_t* __restrict__ stem_wp, const uint16_t* __restrict__ tower_wp,
                                       uint32_t tower_bytes, const float* __restrict__ fold, uint8_t* __restrict__ ex_base, uint32_t ex_bytes_total,
                                       unsigned* counter, unsigned* giveup, unsigned tag, unsigned spin_limit, int force_wt)
{
    // which XCD this workgroup runs on: every workgroup of the pair adds 1 to the 3-bit field of its XCC in the pair's word before its
    // stem stores are drained and counted; behind the stem's hand-off the word says whether all four share one XCD (speed only: the
    // layers' images then stay in that L2 — profiles/r03_small_batch_tower.txt: 5 % of a launch)
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11)) & 7u;
    unsigned* xccw = counter + MAX_PAIRS;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(xccw, 1u << (3u * xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint8_t* bufX = lds;
    const uint8_t* bufF = lds + FEAT_OFF;
    const uint8_t* taprow = lds + TAPROW_OFF;
    const uint16_t* rowcell = reinterpret_cast<const uint16_t*>(lds + ROWCELL_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const bool mh = (wave >> 1) != 0;                   // wave-uniform: this wave's row half
    const int ct0 = cg * 4 + (wave & 1) * NTW;          // this wave's first 16-channel column tile
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(tower_wp), (short)0, (int)tower_bytes, 0x00020000);
    const uint32_t loff = (uint32_t)(ct0 * 64 + lane) * 16u;
    u32x4 bq[RING][NTW];
    // ---- weight ring: the first RING - 1 k-steps of layer 0 fly while the stem runs
#pragma unroll
    for (int ks = 0; ks < RING - 1; ks++)
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) bq[ks][nt] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, loff + nt * 1024, ks * (int)KBYTES, 0);

    f32x4 acc[MTW][NTW];
    uint2 res[MTW][NTW];     // the block input of this wave's (cell, 4-channel) elements, packed bf16 = the residual operand
    uint2 o[MTW][NTW];
    const __amdgpu_buffer_rsrc_t ex = __builtin_amdgcn_make_buffer_rsrc(ex_base, (short)0, (int)ex_bytes_total, 0x00020000);

    if (mh) stem_half<1, F16>(bufF, taprow, rowcell, stem_wp, fold, ct0, lane, c, g, acc, o, res);
    else stem_half<0, F16>(bufF, taprow, rowcell, stem_wp, fold, ct0, lane, c, g, acc, o, res);
    if (mh) publish_stores<1>(o, 1, pair, ct0, c, g, ex, false); else publish_stores<0>(o, 1, pair, ct0, c, g, ex, false);
    count_in(tid, counter);
    if (mh) own_to_lds<1>(bufX, o, ct0, c, g); else own_to_lds<0>(bufX, o, ct0, c, g);
    gather(bufX, 1, pair, cg, tid, ex, counter, giveup, tag, spin_limit);
    // (thread 0's add above was drained with the stem's stores — vmcnt(0) in count_in — before this workgroup was counted in)
    const bool same_xcd = !force_wt &&
        __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(xccw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == (int)((unsigned)CGN << (3u * xcc));

    // ---- residual tower
    const uint8_t* tr_c = taprow + c;
    const uint32_t g16 = (uint32_t)g * 16u;
    const int layers = 2 * blocks;
    uint32_t wk = 0;
    for (int L = 0; L < layers; L++) {
        float4 sc[NTW], sh[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) {
            sc[nt] = *reinterpret_cast<const float4*>(fold + 14 + (size_t)L * 2 * NF + (ct0 + nt) * 16 + g * 4);
            sh[nt] = *reinterpret_cast<const float4*>(fold + 14 + (size_t)L * 2 * NF + NF + (ct0 + nt) * 16 + g * 4);
        }
        if (mh) layer_half<1, F16>(bufX, tr_c, g16, wsrc, loff, wk, (L & 1) != 0, sc, sh, bq, acc, o, res);
        else layer_half<0, F16>(bufX, tr_c, g16, wsrc, loff, wk, (L & 1) != 0, sc, sh, bq, acc, o, res);
        wk += (uint32_t)(9 * KS_PER_TAP) * (uint32_t)KBYTES;
        if (mh) publish_stores<1>(o, L + 2, pair, ct0, c, g, ex, same_xcd); else publish_stores<0>(o, L + 2, pair, ct0, c, g, ex, same_xcd);
        count_in(tid, counter);
        if (mh) own_to_lds<1>(bufX, o, ct0, c, g); else own_to_lds<0>(bufX, o, ct0, c, g);
        if (L == layers - 1) {
            if (cg != 0) return false;   // after the last layer only channel group 0 goes on (the heads)
        } else {
            // the next layer's first RING - 1 k-steps: requested now, they fly during the hand-off
#pragma unroll
            for (int ks = 0; ks < RING - 1; ks++)
#pragma unroll
                for (int nt = 0; nt < NTW; nt++)
                    bq[ks][nt] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, loff + nt * 1024, (int)(wk + (uint32_t)ks * (uint32_t)KBYTES), 0);
        }
        gather(bufX, L + 2, pair, cg, tid, ex, counter, giveup, tag, spin_limit);
    }
    return true;
}

template <bool F16>
__global__ __launch_bounds__(THREADS) void k_tower_sc(const uint8_t* __restrict__ in88, int in_stride, int n, int pairs,
                                                       const uint16_t* __restrict__ stem_wp, const uint16_t* __restrict__ tower_wp, uint32_t tower_bytes,
                                                       const float* __restrict__ fold, int blocks, const float* __restrict__ hp,
                                                       float* __restrict__ pi_out, float* __restrict__ v_out, const int* __restrict__ slot_map,
                                                       uint8_t* __restrict__ ex_base, uint32_t ex_bytes_total, unsigned* __restrict__ counters,
                                                       unsigned spin_limit, int force_wt, const int* __restrict__ n_dev, unsigned tag, const int* __restrict__ n_other)
{
    // n_dev != null: the batch size is a word in device memory (written by the tree step ahead of this launch in stream order; the grid
    // was sized for the largest batch).  More boards than this kernel takes: the give-up word is raised with the value 2 — the guarded
    // one-board-per-workgroup launch behind this one computes the batch, and does not count it as a hand-off that gave up.
    // n_other != null: the batch of ANOTHER network's launch that runs side by side with this one (the two-net arena).  Four workgroups per
    // board pair pay only while every workgroup has a CU to itself (two per CU: 0.70 ms against 0.43, profiles/r03_small_batch_tower.txt —
    // measured again in the arena's trace, profiles/r04_arena_passes.txt: 0.66 ms for the second of two 100-board launches).  So both
    // launches take this kernel only if their workgroups fit the 256 CUs together; else both go one board per workgroup (letting the
    // smaller batch keep this kernel beside the larger one's one-board workgroups measured slower: it slows the larger, which the pass
    // waits for).  Both launches evaluate the same rule on the same two words.
    if (n_dev) {
        n = __builtin_amdgcn_readfirstlane(*n_dev);
        pairs = (n + 1) / 2;
        bool here = n <= 2 * MAX_PAIRS;
        if (here && n_other) {
            const int m = __builtin_amdgcn_readfirstlane(*n_other);
            const int wg_n = CGN * ((n + 1) / 2), wg_m = CGN * ((m + 1) / 2);
            if (m > 0 && wg_n + wg_m > CUS) here = false;
        }
        if (!here) {
            if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(counters + W_GIVEUP, tag | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    // block id -> (pair, channel group): the CGN workgroups of a pair are 8 ids apart
    const int grp = blockIdx.x / (8 * CGN), r = blockIdx.x % (8 * CGN), cg = r / 8, pair = grp * 8 + (r & 7);
    if (pair >= pairs) return;   // (padding of the last group of 8 pairs: belongs to no pair, waits for nobody)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* bufX = lds;
    uint8_t* bufF = lds + FEAT_OFF;
    uint8_t* in_l = lds + IN88_OFF;
    uint8_t* rowof = lds + ROWOF_OFF;
    uint8_t* taprow = lds + TAPROW_OFF;
    uint16_t* rowcell = reinterpret_cast<uint16_t*>(lds + ROWCELL_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int board0 = pair * NB;

    // ---- NNInputData images, zero row, row tables, stem features (as k_tower_sb)
    for (int i = tid; i < NB * 96; i += THREADS) {
        const int b = i / 96, o = i % 96;
        const int slot = (board0 + b < n) ? (slot_map ? slot_map[board0 + b] : board0 + b) : 0;
        in_l[i] = (board0 + b < n && o < 88) ? in88[(size_t)slot * in_stride + o] : (uint8_t)0;
    }
    for (int i = tid; i < ROWB / 4; i += THREADS) reinterpret_cast<uint32_t*>(bufX + ZR * ROWB)[i] = 0;
    for (int i = tid; i < ZR; i += THREADS) rowcell[i] = 0xffffu;
    __syncthreads();
    for (int i = tid; i < ROWS; i += THREADS) {
        const int b = i / 42, pos = i - b * 42, rr = row_of<NB>(b, pos);
        rowof[i] = (uint8_t)rr;
        rowcell[rr] = (uint16_t)((pos / 6) | ((pos % 6) << 4) | (b << 8));
    }
    __syncthreads();
    for (int i = tid; i < 10 * ZR; i += THREADS) {
        const int t = i / ZR, rr = i - t * ZR;
        const int ci = rowcell[rr];
        int src = ZR;
        if (t < 9 && ci != 0xffff) {
            const int y = (ci & 15) + t / 3 - 1, x = ((ci >> 4) & 15) + t % 3 - 1;
            if ((unsigned)y < 7u && (unsigned)x < 6u) src = rowof[(ci >> 8) * 42 + y * 6 + x];
        }
        taprow[i] = (uint8_t)src;
    }
    for (int i = tid; i < (ZR + 1) * 16; i += THREADS) {
        const int rr = i >> 4, ch = i & 15;
        float v = 0.0f;
        const int ci = rr < ZR ? rowcell[rr] : 0xffff;
        if (ci != 0xffff) v = plane_value(in_l + (ci >> 8) * 96, (ci & 15) * 6 + ((ci >> 4) & 15), ch);
        reinterpret_cast<uint16_t*>(bufF)[i] = El<F16>::rne(v);
    }
    // pad rows of the image are MFMA operands of nobody (their tile rows read the zero row) but the image is also fetched whole:
    // keep them defined
    for (int i = tid; i < ZR * (ROWB / 4); i += THREADS) reinterpret_cast<uint32_t*>(bufX)[i] = 0;
    __syncthreads();

    // ---- stem and tower
    if (!sc_run<F16>(lds, pair, cg, blocks, stem_wp, tower_wp, tower_bytes, fold, ex_base, ex_bytes_total, counters + pair, counters + W_GIVEUP, tag, spin_limit, force_wt))
        return;

    // ---- both heads for the pair (k_tower_sb's fused heads), channel group 0 only
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
        float* wl = reinterpret_cast<float*>(lds + HEAD_OFF);   // [3][256]
        float* feat = wl + 3 * NF;                               // [NB][128]
        float* hid = feat + NB * 128;                            // [NB][256]
        float* logit = hid + NB * 256;                           // [NB][64]
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
            if (ch < 2) feat[bb * 128 + pos * 2 + ch] = y;
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
        for (int job = wave; job < NB * 2; job += 4) {
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

}  // namespace

namespace azr {

int tower_sc_init(azr_engine* h)
{
    Bf16Net* x = bf16net(h);
    const size_t ex_bytes = (size_t)2 * MAX_PAIRS * EX_PAIR_BYTES;   // [pair][parity of the epoch][96 rows][256] bf16
    HIPCHK(h, hipMalloc((void**)&x->sc_ex, ex_bytes));
    HIPCHK(h, hipMemsetAsync(x->sc_ex, 0, ex_bytes, h->stream));
    // [MAX_PAIRS] arrival counters | [MAX_PAIRS] XCC words | give-up word of the running launch | count of recomputed launches
    static_assert(W_GIVEUP == SC_W_GIVEUP && W_FALLBACKS == SC_W_FALLBACKS, "the guarded fallback launch (azr_net_bf16.hip) reads these words");
    HIPCHK(h, hipMalloc((void**)&x->sc_counters, SC_WORDS * sizeof(unsigned)));
    HIPCHK(h, hipMemsetAsync(x->sc_counters, 0, SC_WORDS * sizeof(unsigned), h->stream));
    // test hooks (libazr_hip_test.so only): AZR_TOWER_SC_SPIN = polls before a hand-off gives up (0: every hand-off whose partners are
    // not there yet gives up at once — forces the recompute path); AZR_TOWER_SC_WT=1: never the plain-store form of a same-XCD pair
    x->sc_spin_limit = (unsigned)hook_env_int("AZR_TOWER_SC_SPIN", (int)SPIN_LIMIT);
    x->sc_force_wt = hook_env_int("AZR_TOWER_SC_WT", 0);
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_tower_sc<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_tower_sc<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    return AZR_OK;
}

void tower_sc_free(azr_engine* h)
{
    Bf16Net* x = bf16net(h);
    if (!x) return;
    if (x->sc_ex) hipFree(x->sc_ex);
    if (x->sc_counters) hipFree(x->sc_counters);
    x->sc_ex = nullptr;
    x->sc_counters = nullptr;
}

// the whole net for n <= 128 boards in one persistent launch of 4 workgroups per board pair.  The caller queues the guarded
// one-board-per-workgroup launch right behind it (net_bf16_forward): if a hand-off of this launch gave up, that one recomputes the batch.
// n_dev != null: the batch size is read from that word of device memory by the launch itself and n is only its upper bound (the grid);
// n_other (optional): the batch-size word of another network's launch running beside this one (see the kernel).
int tower_sc_launch(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st,
                    const int* n_dev, const int* n_other)
{
    if (n_dev && n > 2 * MAX_PAIRS) n = 2 * MAX_PAIRS;
    if (n < 1 || n > 2 * MAX_PAIRS) { h->err = "tower_sc_launch: 1..128 boards"; return AZR_E_INVALID_ARGUMENT; }
    Bf16Net* x = bf16net(h);
    const int pairs = (n + 1) / 2, B = h->net.blocks;
    const int wgs = ((pairs + 7) / 8) * 8 * CGN;     // whole groups of 8 pairs (ids of a pair's workgroups are 8 apart)
    const uint32_t tower_bytes = (uint32_t)(((size_t)2 * B * TOWER_LAYER_HALFS + MAX_RING * KSTRIDE * 8) * 2);
    const uint32_t ex_bytes = (uint32_t)((size_t)2 * MAX_PAIRS * EX_PAIR_BYTES);
    // The pairs' words count within ONE launch: the guarded launch the caller queues behind this one zeroes them again (workgroup 0, before
    // anything else) — no memset in front of a launch.  The give-up word carries the launch's serial number in its upper 30 bits: what an
    // earlier launch left there is nobody's business.
    x->sc_tag = (++x->sc_serial & 0x3fffffffu) << 2;
    const unsigned tag = x->sc_tag;
    if (x->f16)
        hipLaunchKernelGGL(k_tower_sc<true>, dim3(wgs), dim3(THREADS), LDS_BYTES, st, d_in88, in_stride, n, pairs, x->stem_wp, x->tower_wp, tower_bytes, (const float*)x->fold16, B,
                           net_head_params(h), d_pi, d_v, d_map, reinterpret_cast<uint8_t*>(x->sc_ex), ex_bytes, x->sc_counters, x->sc_spin_limit, x->sc_force_wt, n_dev, tag, n_other);
    else
        hipLaunchKernelGGL(k_tower_sc<false>, dim3(wgs), dim3(THREADS), LDS_BYTES, st, d_in88, in_stride, n, pairs, x->stem_wp, x->tower_wp, tower_bytes, net_fold(h), B,
                           net_head_params(h), d_pi, d_v, d_map, reinterpret_cast<uint8_t*>(x->sc_ex), ex_bytes, x->sc_counters, x->sc_spin_limit, x->sc_force_wt, n_dev, tag, n_other);
    HIPCHK(h, hipGetLastError());
    return AZR_OK;
}

// launches of the split-channel tower that gave up and were recomputed by the guarded launch behind them, since the handle was created
int tower_sc_fallbacks(azr_engine* h, unsigned long long* out)
{
    Bf16Net* x = bf16net(h);
    unsigned v = 0;
    if (x && x->sc_counters) {
        HIPCHK(h, hipMemcpyAsync(&v, x->sc_counters + W_FALLBACKS, sizeof v, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    *out = v;
    return AZR_OK;
}

}  // namespace azr
