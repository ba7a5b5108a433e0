// azr_net_bf16.hip — the residual tower of python/src/build_graph.py:63-74 as ONE persistent MFMA kernel (gfx950).
//
// MI355X-first design (not a per-layer conv library call):
//   * A workgroup (8 waves, 512 threads) owns NB whole boards (NB = 1, 2 or 3 -> M = 48 / 96 / 128 GEMM rows) for the
//     ENTIRE stem + 2B conv layers.  The boards' activations (42 x 256 bf16 each) never leave LDS: two ping-pong
//     buffers [rows + 1 zero row][256 + 16 pad] bf16, 138 KB of the CU's 160 KB at NB = 3.  HBM sees the 88-byte
//     inputs, the weights, and the final activation only.
//   * Implicit GEMM per layer: M = board cells, N = 256 output channels, K = 9 taps x 256 input channels, on
//     v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  The A operand of tap (dy,dx) is the SAME LDS image read at a
//     per-lane row offset (out-of-board neighbours read the zero row) — no im2col, no halo copies.
//   * Waves split N (32 channels = two 16-wide tiles each), so weight fragments are private to a wave and stream
//     global -> VGPR with no LDS staging, pre-packed on the host in exactly the lane order of the MFMA B operand
//     (one coalesced 1-KiB global_load_dwordx4 per fragment), double-buffered 4 k-steps ahead.
//   * Epilogue per layer in registers: folded BN (fp32 scale/shift per channel = per lane), residual add (read from
//     the LDS image being replaced), ReLU, round-to-nearest-even bf16, written straight back into LDS.
//   * conv_bn of the stem normalises over the board ROW (build_graph.py:68 axis=1): per-row scale/shift.
// One barrier per layer.  The heads (1x1 convs + dense layers, 47 k MAC/board) run at the end of the same launch on the
// LDS-resident tower output: one kernel = one whole net forward, HBM sees 88 B in and 45 floats out per board.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "azr_internal.hpp"
#include "azr_bf16_common.hpp"

using namespace azr;

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (void)hipGetLastError(); /* the runtime's last-error slot is sticky: clear it */        \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return AZR_E_HIP;                                                                   \
        }                                                                                       \
    } while (0)


namespace {
// ---------------------------------------------------------------------------------------------------------------
// Row order inside a workgroup.  A 3x3 SAME conv on the 7x6 board has 304 valid (cell, tap) pairs of 378: a fifth of a
// dense tiling's MFMAs multiply the zero row.  Cells are therefore ordered by border class so that whole 16-row MFMA
// tiles are out of board for a tap and both the MFMAs and the LDS fragment reads of that (tile, tap) are skipped:
//   3 boards (126 cells, 8 tiles):  tile 0 = 16 cells with y = 0 (no dy = -1 taps), tile 1 = 16 cells with y = 6 (no
//     dy = +1), tile 2 = the 15 cells x = 0, y = 1..5 + 1 pad row (no dx = -1), tile 3 = the 15 cells x = 5 (no dx = +1),
//     tiles 4..7 = the 2 + 2 left-over y = 0 / y = 6 cells and the 60 interior cells          -> 12 of 72 tile-taps skipped
//   2 boards (84 cells, 6 tiles):   tile 0 = 12 cells y = 0 + 4 pad rows, tile 1 = 12 cells y = 6 + 4 pad rows,
//     tiles 2..5 = the other 60 cells + 4 pad rows                                             -> 6 of 54 skipped
//   1 board: identity (that shape is bound by the weight stream, not by the MFMA pipe).
// The skipped products are exact zeros, so the results do not change.  row_of() is the only definition of the order; the
// kernel derives its LDS tables (cell -> row, row -> cell) from it.
// ---------------------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ int row_of(int b, int pos)
{
    const int y = pos / 6, x = pos - y * 6;
    if constexpr (NB == 3) {
        if (y == 0) { const int q = b * 6 + x; return q < 16 ? q : 64 + (q - 16); }
        if (y == 6) { const int q = b * 6 + x; return q < 16 ? 16 + q : 66 + (q - 16); }
        if (x == 0) return 32 + b * 5 + (y - 1);
        if (x == 5) return 48 + b * 5 + (y - 1);
        return 68 + b * 20 + (y - 1) * 4 + (x - 1);
    } else if constexpr (NB == 2) {
        if (y == 0) return b * 6 + x;
        if (y == 6) return 16 + b * 6 + x;
        return 32 + b * 30 + (y - 1) * 6 + x;
    } else {
        return b * 42 + pos;
    }
}
// is row r a board cell (not one of the pad rows of the order above)?
template <int NB>
__device__ __forceinline__ bool row_valid(int r)
{
    if constexpr (NB == 3) return r != 47 && r != 63;
    else if constexpr (NB == 2) return r < 32 ? (r & 15) < 12 : r < 92;
    else return r < 42;
}
// bit mt set = M tile mt has no in-board cell for this tap
template <int NB>
constexpr uint32_t skip_mask(int tap)
{
    if (tap < 0 || tap > 8) return 0xffffffffu;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    uint32_t m = 0;
    if (NB >= 2) m |= (dy == -1 ? 1u : 0u) | (dy == 1 ? 2u : 0u);
    if (NB == 3) m |= (dx == -1 ? 4u : 0u) | (dx == 1 ? 8u : 0u);
    return m;
}

// LDS row a lane reads for cell info `ri` (y | x << 4 | board << 8; 0xffff for a pad row) under tap (dy,dx); taps
// outside the board read the zero row
__device__ __forceinline__ int tap_row(int ri, int dy, int dx, const uint8_t* __restrict__ rowof, int zero_row)
{
    const int y = (ri & 15) + dy, x = ((ri >> 4) & 15) + dx;
    const bool ok = (unsigned)y < 7u && (unsigned)x < 6u;
    const int cell = ok ? ((ri >> 8) & 3) * 42 + y * 6 + x : 0;
    const int r = rowof[cell];
    return ok ? r : zero_row;
}

template <int NB>
struct Geo {
    static constexpr int ROWS = 42 * NB;
    static constexpr int MT = (ROWS + 15) / 16;
    static constexpr int ZR = MT * 16;              // index of the shared zero row
    static constexpr int BUF = (ZR + 1) * ROWB;     // one activation buffer incl. its zero row
    static constexpr int IN88_OFF = 2 * BUF;        // NB x 96 B of NNInputData images
    static constexpr int ROWOF_OFF = IN88_OFF + NB * 96;   // u8 [NB * 42 (+ pad to 128)]: cell -> row
    static constexpr int ROWCELL_OFF = ROWOF_OFF + 128;    // u16 [MT * 16]: row -> y | x << 4 | board << 8, 0xffff = pad
    static constexpr int LDS_BYTES = ROWCELL_OFF + 2 * MT * 16;
};


// ---------------------------------------------------------------------------------------------------------------
// Wave tiling: the 16 column tiles (16 channels each) of N = 256 are split over WAVES = 16 / NT waves, NT tiles per
// wave.  NT = 2 -> 8 waves (2 per SIMD, <= 256 VGPRs);  NT = 4 -> 4 waves (ONE per SIMD, up to 512 VGPRs): every A
// fragment read from LDS then feeds 4 MFMAs instead of 2, halving the LDS traffic that co-limits the 8-wave shape.
// The packed weight stream is identical for both (fragment index = column tile).
// ---------------------------------------------------------------------------------------------------------------

// ring depth in taps (8 k-steps each)
template <int NB, int NT> struct RingTaps { static constexpr int value = (NB == 1) ? 2 : 1; };

// Operand orientation.  Swapped (1-board tile): weights are the MFMA "A" operand, so D = [channel][cell] and the
// epilogue stores 4 consecutive channels per lane (8-byte LDS ops, 4x fewer).  The swapped epilogue needs 16 VGPRs of
// BN constants instead of 4, which the 3-board tile (already at the 256-VGPR limit) cannot afford: it keeps D = [cell][channel].
template <int MT> struct Swap { static constexpr bool value = MT <= 3; };
__device__ __forceinline__ float bn_x(float v) { return v; }
__device__ __forceinline__ float bn_x(const float4& v) { return v.x; }
__device__ __forceinline__ float4 bn_4(float v) { return float4{v, v, v, v}; }
__device__ __forceinline__ float4 bn_4(const float4& v) { return v; }
template <bool SWAP> struct BnConst;                                  // folded-BN constants a lane needs per column tile
template <> struct BnConst<true> { typedef float4 type; };           // 4 consecutive channels
template <> struct BnConst<false> { typedef float type; };           // 1 channel

// one tap = 8 k-steps against ring slots SB .. SB+7.  `wb` is the wave-UNIFORM byte pointer to the current
// k-step's 16-KiB fragment block (advanced with scalar adds); `loff` is this lane's byte offset inside a block.
// SK / SKN: skip_mask of this tap / of the tap whose first fragments are prefetched at the end (all ones = none)
template <int MT, int NT, int RT, int SB, uint32_t SK, uint32_t SKN, bool F16>
__device__ __forceinline__ void conv_tap(const uint8_t* IN, int tap, const char* __restrict__& wb, uint32_t loff,
                                         s16x8 (&bq)[RT * 8][NT], f32x4 (&acc)[MT][NT], s16x8 (&a)[2][MT], int (&aoff)[MT],
                                         const int (&rinfo)[MT], int g16, const uint8_t* __restrict__ rowof, int zero_row)
{
    const int ntap = tap < 8 ? tap + 1 : 8;
    const int ndy = ntap / 3 - 1, ndx = ntap % 3 - 1;
    int noff[MT];
#pragma unroll
    for (int ks = 0; ks < KS_PER_TAP; ks++) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks == KS_PER_TAP - 2) {  // the next tap's rows, one k-step before they are needed (short live range)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
                if (!((SKN >> mt) & 1u)) {
                    // (3-board tile: at the 256-VGPR limit the cell info is re-read from LDS instead of held in registers)
                    const int ri = MT == 8 ? (int)reinterpret_cast<const uint16_t*>(rowof + 128)[mt * 16 + (threadIdx.x & 15)] : rinfo[mt];
                    noff[mt] = tap_row(ri, ndy, ndx, rowof, zero_row) * ROWB + g16;
                }
        }
        // (1) LDS reads of the NEXT k-step's A fragments go out first ...
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            if (ks < KS_PER_TAP - 1) { if (!((SK >> mt) & 1u)) a[nxt][mt] = *reinterpret_cast<const s16x8*>(IN + aoff[mt] + (ks + 1) * 64); }
            else if (!((SKN >> mt) & 1u)) a[nxt][mt] = *reinterpret_cast<const s16x8*>(IN + noff[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (2) ... and fly under this k-step's MFMAs; then the freed ring slot is refilled one ring ahead
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
                if ((SK >> mt) & 1u) {
                    // every row of this tile is out of board for this tap: the products are exact zeros
                } else if constexpr (Swap<MT>::value)
                    // weights as the MFMA "A" operand, activations as "B": D[channel][cell], so a lane ends up with 4
                    // CONSECUTIVE CHANNELS of one board cell (row = 4*(lane>>4)+j, col = lane&15) -> 8-byte LDS stores
                    acc[mt][nt] = El<F16>::mfma(bq[SB + ks][nt], a[cur][mt], acc[mt][nt]);
                else
                    acc[mt][nt] = El<F16>::mfma(a[cur][mt], bq[SB + ks][nt], acc[mt][nt]);
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            bq[SB + ks][nt] = *reinterpret_cast<const s16x8*>(wb + RT * 8 * KBYTES + loff + nt * 1024);
        wb += KBYTES;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
        if (!((SKN >> mt) & 1u)) aoff[mt] = noff[mt];
}

// One 3x3 conv layer F->F: acc[mt][nt] = sum over 9 taps x 256 channels (72 k-steps of 32).
// Weight fragments come from a RING of RT*8 k-steps held in VGPRs that never drains: the packed tower weights of
// all layers are one contiguous stream in exactly consumption order, so a slot is refilled with the k-step one ring
// ahead right after its MFMAs issue — also across layer boundaries, where the epilogue + barrier then overlap the
// next layer's weight latency.  A fragments are double-buffered one k-step ahead so their LDS latency hides under
// the current MFMAs; scheduling regions (sched_barrier) keep that order.
// PAR = parity of the layer's first tap in the global tap sequence (9 taps per layer: it alternates per layer).
template <int MT, int NT, int RT, int PAR, bool F16>
__device__ __forceinline__ void conv_tower_layer(const uint8_t* IN, const char* __restrict__& wb, uint32_t loff,
                                                 s16x8 (&bq)[RT * 8][NT], f32x4 (&acc)[MT][NT], const int (&rinfo)[MT], int g16,
                                                 const uint8_t* __restrict__ rowof, int zero_row, const float* __restrict__ fs, typename BnConst<Swap<MT>::value>::type (&sc)[NT],
                                                 typename BnConst<Swap<MT>::value>::type (&sh)[NT])
{
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0, 0, 0, 0};
    int aoff[MT];
    s16x8 a[2][MT];
    constexpr int NBX = MT == 8 ? 3 : MT == 6 ? 2 : 1;   // boards per workgroup of this tile count
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        if ((skip_mask<NBX>(0) >> mt) & 1u) { aoff[mt] = zero_row * ROWB + g16; continue; }
        const int ri = MT == 8 ? (int)reinterpret_cast<const uint16_t*>(rowof + 128)[mt * 16 + (threadIdx.x & 15)] : rinfo[mt];
        aoff[mt] = tap_row(ri, -1, -1, rowof, zero_row) * ROWB + g16;
        a[0][mt] = *reinterpret_cast<const s16x8*>(IN + aoff[mt]);
    }
    // this layer's folded BN (4 consecutive channels per lane and tile) is requested one tap before the epilogue:
    // early enough not to wait behind the weight ring, late enough not to hold 16 VGPRs through the layer
    auto load_bn = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            if constexpr (Swap<MT>::value) {
                sc[nt] = *reinterpret_cast<const typename BnConst<Swap<MT>::value>::type*>(fs + nt * 16);
                sh[nt] = *reinterpret_cast<const typename BnConst<Swap<MT>::value>::type*>(fs + NF + nt * 16);
            } else {  // one channel per lane and tile
                sc[nt] = fs[nt * 16];
                sh[nt] = fs[NF + nt * 16];
            }
        }
    };
    if constexpr (RT == 1) {
        load_bn();  // (register allocation at the 256-VGPR limit of the 3-board tile is best with the early load)
#define AZR_TAP(T) conv_tap<MT, NT, 1, 0, skip_mask<NBX>(T), skip_mask<NBX>(T + 1), F16>(IN, T, wb, loff, bq, acc, a, aoff, rinfo, g16, rowof, zero_row)
        AZR_TAP(0); AZR_TAP(1); AZR_TAP(2); AZR_TAP(3); AZR_TAP(4); AZR_TAP(5); AZR_TAP(6); AZR_TAP(7); AZR_TAP(8);
#undef AZR_TAP
    } else {
        constexpr int S0 = PAR ? 8 : 0, S1 = PAR ? 0 : 8;
        for (int tap = 0; tap < 8; tap += 2) {
            conv_tap<MT, NT, 2, S0, 0u, 0u, F16>(IN, tap, wb, loff, bq, acc, a, aoff, rinfo, g16, rowof, zero_row);
            conv_tap<MT, NT, 2, S1, 0u, 0u, F16>(IN, tap + 1, wb, loff, bq, acc, a, aoff, rinfo, g16, rowof, zero_row);
        }
        load_bn();
        conv_tap<MT, NT, 2, S0, 0u, 0u, F16>(IN, 8, wb, loff, bq, acc, a, aoff, rinfo, g16, rowof, zero_row);
    }
}

// the whole network for the NB boards [board0, board0 + NB) of one workgroup
template <int NB, int NT, bool F16>
__device__ __forceinline__ void tower_body(uint8_t* __restrict__ lds, const int board0, const uint8_t* __restrict__ in88, int in_stride,
                                           int n, const uint16_t* __restrict__ stem_wp, const uint16_t* __restrict__ tower_wp,
                                           const float* __restrict__ fold, int blocks, const float* __restrict__ hp,
                                           float* __restrict__ pi_out, float* __restrict__ v_out,
                                           unsigned long long* __restrict__ diag, const int* __restrict__ slot_map)
{
    using G = Geo<NB>;
    constexpr int ROWS = G::ROWS, MT = G::MT, ZR = G::ZR, THREADS = 1024 / NT, WCOLS = NT * 16;
    uint8_t* bufX = lds;
    uint8_t* bufT = lds + G::BUF;
    uint8_t* in_l = lds + G::IN88_OFF;
    uint8_t* rowof = lds + G::ROWOF_OFF;                                        // cell (board * 42 + pos) -> row
    uint16_t* rowcell = reinterpret_cast<uint16_t*>(lds + G::ROWCELL_OFF);      // row -> y | x << 4 | board << 8
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    // clock diagnostic (azr_debug_tower_clock): shader-clock and 100 MHz real-time stamps around the whole tower of
    // workgroup 0; `diag` is null in every product launch
    if (diag && blockIdx.x == 0 && tid == 0) { diag[0] = __builtin_amdgcn_s_memtime(); diag[1] = __builtin_amdgcn_s_memrealtime(); }

    // ---- stage the NNInputData images, zero the zero rows and the stem feature image
    for (int i = tid; i < NB * 96; i += THREADS) {
        const int b = i / 96, o = i % 96;
        // slot_map (optional): board i of this launch is leaf slot slot_map[i] (two-net arena: each net sees its own leaves)
        const int slot = (board0 + b < n) ? (slot_map ? slot_map[board0 + b] : board0 + b) : 0;
        in_l[i] = (board0 + b < n && o < 88) ? in88[(size_t)slot * in_stride + o] : (uint8_t)0;
    }
    for (int i = tid; i < ROWB / 4; i += THREADS) {
        reinterpret_cast<uint32_t*>(bufX + ZR * ROWB)[i] = 0;
        reinterpret_cast<uint32_t*>(bufT + ZR * ROWB)[i] = 0;
    }
    // pad rows of the row order hold zeros too (their accumulator rows are never stored, but they are MFMA operands)
    for (int i = tid; i < (ZR - ROWS) * (ROWB / 4); i += THREADS) {
        int pr = i / (ROWB / 4), k = 0;
        for (int r = 0; r < ZR; r++)
            if (!row_valid<NB>(r)) { if (k == pr) { pr = r; break; } k++; }
        reinterpret_cast<uint32_t*>(bufX + pr * ROWB)[i % (ROWB / 4)] = 0;
        reinterpret_cast<uint32_t*>(bufT + pr * ROWB)[i % (ROWB / 4)] = 0;
    }
    for (int i = tid; i < ZR; i += THREADS) rowcell[i] = 0xffffu;
    __syncthreads();
    for (int i = tid; i < ROWS; i += THREADS) {
        const int b = i / 42, pos = i - b * 42, r = row_of<NB>(b, pos);
        rowof[i] = (uint8_t)r;
        rowcell[r] = (uint16_t)((pos / 6) | ((pos % 6) << 4) | (b << 8));
    }
    __syncthreads();
    // stem features: bufT as [ZR + 1][16] bf16 (row ZR = zero row); planes 13..15 are zero
    for (int i = tid; i < (ZR + 1) * 16; i += THREADS) {
        const int r = i >> 4, c = i & 15;
        float v = 0.0f;
        const int ci = r < ZR ? rowcell[r] : 0xffff;
        if (ci != 0xffff) v = plane_value(in_l + (ci >> 8) * 96, (ci & 15) * 6 + ((ci >> 4) & 15), c);
        reinterpret_cast<uint16_t*>(bufT)[i] = El<F16>::rne(v);
    }
    __syncthreads();

    // ---- per-lane geometry of the rows this lane feeds as MFMA A operand (row = mt*16 + m)
    int rinfo[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) rinfo[mt] = rowcell[mt * 16 + m];   // 0xffff (y = x = 15) for a pad row
    f32x4 acc[MT][NT];

    // ---- start the weight ring: the first ring of k-steps of layer 0 flies while the stem runs
    const char* __restrict__ wb = reinterpret_cast<const char*>(tower_wp);      // wave-uniform, scalar-advanced
    const uint32_t loff = (uint32_t)((wave * NT) * 64 + lane) * 16u;             // this lane's fragment bytes in a k-step
    constexpr int RT = RingTaps<NB, NT>::value;
    s16x8 bq[RT * 8][NT];
#pragma unroll
    for (int ks = 0; ks < RT * 8; ks++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) bq[ks][nt] = *reinterpret_cast<const s16x8*>(wb + ks * KBYTES + loff + nt * 1024);

    // ---- stem: 3x3 conv 13 -> 256, two taps per 32-deep k-step (tap slot = 2*ks + (g >> 1), channels (g & 1)*8 ..)
    {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0, 0, 0, 0};
        const s16x8* wp = reinterpret_cast<const s16x8*>(stem_wp) + (size_t)(wave * NT) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < STEM_KS; ks++) {
            const int tap = 2 * ks + (g >> 1);
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            s16x8 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b[nt] = wp[(size_t)ks * FRAGS_PER_KSTEP * 64 + nt * 64];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int row = tap < 9 ? tap_row(rinfo[mt], dy, dx, rowof, ZR) : ZR;
                const s16x8 av = *reinterpret_cast<const s16x8*>(bufT + row * FROWB + (g & 1) * 16);
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    if constexpr (Swap<MT>::value)
                        acc[mt][nt] = El<F16>::mfma(b[nt], av, acc[mt][nt]);
                    else
                        acc[mt][nt] = El<F16>::mfma(av, b[nt], acc[mt][nt]);
            }
        }
        if constexpr (Swap<MT>::value) {
        // conv_bn over the board row + ReLU -> bufX.  D layout: col = lane & 15 = board cell, row = 4*(lane>>4)+j = channel
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int r = mt * 16 + m;
            if (rinfo[mt] != 0xffff) {
                const int y = rinfo[mt] & 15;
                const float sc = fold[y], sh = fold[7 + y];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const int c0 = wave * WCOLS + nt * 16 + g * 4;
                    uint16_t o4[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float v = fmaf(acc[mt][nt][j], sc, sh);
                        o4[j] = El<F16>::rne(v > 0.0f ? v : 0.0f);
                    }
                    *reinterpret_cast<uint2*>(bufX + r * ROWB + c0 * 2) = uint2{(uint32_t)o4[0] | ((uint32_t)o4[1] << 16), (uint32_t)o4[2] | ((uint32_t)o4[3] << 16)};
                }
            }
        }
        } else {
            // D layout: col = lane & 15 = channel, row = 4*(lane>>4)+j = board cell
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int r = mt * 16 + g * 4 + j;
                    const int ci = rowcell[r];
                    if (ci != 0xffff) {
                        const int y = ci & 15;
                        const float sc = fold[y], sh = fold[7 + y];
#pragma unroll
                        for (int nt = 0; nt < NT; nt++) {
                            const float v = fmaf(acc[mt][nt][j], sc, sh);
                            reinterpret_cast<uint16_t*>(bufX + r * ROWB)[wave * WCOLS + nt * 16 + m] = El<F16>::rne(v > 0.0f ? v : 0.0f);
                        }
                    }
                }
        }
    }
    __syncthreads();

    // ---- residual tower: 2 conv layers per block, activations resident in LDS
    const int g16 = g * 16;
    typedef typename BnConst<Swap<MT>::value>::type bn_t;
    auto epilogue = [&](bool second, uint8_t* OUT, const bn_t (&sc)[NT], const bn_t (&sh)[NT]) {
        if constexpr (!Swap<MT>::value) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int r = mt * 16 + g * 4 + j;
                    if (row_valid<NB>(r)) {
#pragma unroll
                        for (int nt = 0; nt < NT; nt++) {
                            uint16_t* o = reinterpret_cast<uint16_t*>(OUT + r * ROWB) + wave * WCOLS + nt * 16 + m;
                            float v = fmaf(acc[mt][nt][j], bn_x(sc[nt]), bn_x(sh[nt]));
                            if (second) v += El<F16>::tof(*o);  // shortcut: OUT still holds the block's input at this element
                            *o = El<F16>::rne(v > 0.0f ? v : 0.0f);
                        }
                    }
                }
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int r = mt * 16 + m;
            if (row_valid<NB>(r)) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    uint2* o = reinterpret_cast<uint2*>(OUT + r * ROWB + (wave * WCOLS + nt * 16 + g * 4) * 2);
                    const float4 s4 = bn_4(sc[nt]), h4 = bn_4(sh[nt]);
                    float v0 = fmaf(acc[mt][nt][0], s4.x, h4.x), v1 = fmaf(acc[mt][nt][1], s4.y, h4.y);
                    float v2 = fmaf(acc[mt][nt][2], s4.z, h4.z), v3 = fmaf(acc[mt][nt][3], s4.w, h4.w);
                    if (second) {  // shortcut: OUT still holds the block's input at these 4 channels of this cell
                        const uint2 x = *o;
                        v0 += El<F16>::tof((uint16_t)(x.x & 0xffffu)); v1 += El<F16>::tof((uint16_t)(x.x >> 16));
                        v2 += El<F16>::tof((uint16_t)(x.y & 0xffffu)); v3 += El<F16>::tof((uint16_t)(x.y >> 16));
                    }
                    const uint32_t lo = (uint32_t)El<F16>::rne(v0 > 0.0f ? v0 : 0.0f) | ((uint32_t)El<F16>::rne(v1 > 0.0f ? v1 : 0.0f) << 16);
                    const uint32_t hi = (uint32_t)El<F16>::rne(v2 > 0.0f ? v2 : 0.0f) | ((uint32_t)El<F16>::rne(v3 > 0.0f ? v3 : 0.0f) << 16);
                    *o = uint2{lo, hi};
                }
            }
        }
    };
    for (int blk = 0; blk < blocks; blk++) {
        bn_t sc[NT], sh[NT];
        const float* fs = fold + 14 + (size_t)(2 * blk) * 2 * NF + wave * WCOLS + (Swap<MT>::value ? g * 4 : m);
        conv_tower_layer<MT, NT, RT, 0, F16>(bufX, wb, loff, bq, acc, rinfo, g16, rowof, ZR, fs, sc, sh);
        epilogue(false, bufT, sc, sh);
        __syncthreads();
        conv_tower_layer<MT, NT, RT, 1, F16>(bufT, wb, loff, bq, acc, rinfo, g16, rowof, ZR, fs + 2 * NF, sc, sh);
        epilogue(true, bufX, sc, sh);
        __syncthreads();
    }

    if (diag && blockIdx.x == 0 && tid == 0) { diag[2] = __builtin_amdgcn_s_memtime(); diag[3] = __builtin_amdgcn_s_memrealtime(); }
    // ---- both heads, fused (build_graph.py:76-90; same arithmetic order as k_heads in azr_net.hip).  The tower output
    // stays in LDS (bufX); bufT is free and holds the head features.  47 k MAC per board: VALU work.
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
        float* feat = reinterpret_cast<float*>(bufT);   // [NB][128]: 84 policy features, then 42 value features
        float* hid = feat + NB * 128;                   // [NB][256]
        float* logit = hid + NB * 256;                  // [NB][64]
        // 1x1 convs (256 -> 2 policy + 1 value channel per cell): the three weight columns are staged in LDS (bufT is free;
        // the dense scratch above takes its first NB * 1792 bytes) and the activations are read 8 channels at a time — the
        // same fma chain over ci = 0..255 as ever, so the same bits (the fp32 path's k_heads and the 4-board kernel agree)
        float* wl = reinterpret_cast<float*>(bufT + 8192);       // [3][256]
        for (int i = tid; i < 3 * NF; i += THREADS) wl[i] = i < 2 * NF ? wpi[(i & (NF - 1)) * 2 + (i >> 8)] : wv[i - 2 * NF];
        __syncthreads();
        for (int idx = tid; idx < NB * 126; idx += THREADS) {  // 42 cells x {pi0, pi1, v} per board
            const int bb = idx / 126, t = idx % 126, pos = t / 3, c = t % 3;
            const s16x8* x8 = reinterpret_cast<const s16x8*>(bufX + rowof[bb * 42 + pos] * ROWB);
            const float4* w4 = reinterpret_cast<const float4*>(wl + c * NF);
            float sacc = 0.0f;
            for (int q = 0; q < NF / 8; q++) {
                const s16x8 xx = x8[q];
                const float4 wa = w4[2 * q], wb = w4[2 * q + 1];
                sacc = fmaf(El<F16>::tof((uint16_t)xx[0]), wa.x, sacc); sacc = fmaf(El<F16>::tof((uint16_t)xx[1]), wa.y, sacc);
                sacc = fmaf(El<F16>::tof((uint16_t)xx[2]), wa.z, sacc); sacc = fmaf(El<F16>::tof((uint16_t)xx[3]), wa.w, sacc);
                sacc = fmaf(El<F16>::tof((uint16_t)xx[4]), wb.x, sacc); sacc = fmaf(El<F16>::tof((uint16_t)xx[5]), wb.y, sacc);
                sacc = fmaf(El<F16>::tof((uint16_t)xx[6]), wb.z, sacc); sacc = fmaf(El<F16>::tof((uint16_t)xx[7]), wb.w, sacc);
            }
            const float* bnp = c < 2 ? bnpi : bnv;
            const int nc = c < 2 ? 2 : 1, kk = c < 2 ? c : 0;
            float y = (sacc - bnp[2 * nc + kk]) * (bnp[kk] / sqrtf(bnp[3 * nc + kk] + 1e-3f)) + bnp[nc + kk];
            y = y > 0.0f ? y : 0.0f;
            if (c < 2) feat[bb * 128 + pos * 2 + c] = y;  // NHWC flatten: (y*6+x)*2 + c
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
        for (int job = wave; job < NB * 2; job += THREADS / 64) {
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

// Workgroups [0, n_full) carry NB boards, the rest NB - 1: a batch that is not a whole number of 256-workgroup waves of
// NB boards is split into whole waves of mixed size instead (2048 boards = 512 x 3 + 256 x 2: each CU slot runs 3 + 3 + 2
// boards rather than a 2.67-wave tail).  NB - 1 runs the NB - 1 instantiation of the same body inside this kernel's LDS.
template <int NB, int NT, bool F16>
__global__ __launch_bounds__(1024 / NT, NT == 2 ? 2 : 1) void k_tower_bf16(const uint8_t* __restrict__ in88, int in_stride, int n,
                                                                           const uint16_t* __restrict__ stem_wp,
                                                                           const uint16_t* __restrict__ tower_wp,
                                                                           const float* __restrict__ fold, int blocks,
                                                                           const float* __restrict__ hp,
                                                                           float* __restrict__ pi_out, float* __restrict__ v_out,
                                                                           unsigned long long* __restrict__ diag, int n_full,
                                                                           const int* __restrict__ slot_map, unsigned* __restrict__ guard,
                                                                           const int* __restrict__ n_dev, unsigned tag)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int bid = blockIdx.x;
    // guard != null: this launch stands behind a k_tower_sc launch of the same batch and runs only if that one raised its give-up word
    // (1: a hand-off ran out of polls, its results are garbage; 2: more boards than it takes) — every workgroup ends here otherwise;
    // workgroup 0 counts the recompute of a launch that gave up
    // (the word holds (serial of the launch << 2) | reason; tag = this launch pair's serial << 2).  Workgroup 0 also zeroes the pairs' arrival
    // counters and XCC words for the next k_tower_sc launch — this kernel is what runs between two of them in stream order.
    if (guard) {
        if (bid == 0 && threadIdx.x < SC_W_GIVEUP) guard[threadIdx.x] = 0u;
        const unsigned word = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(guard + SC_W_GIVEUP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int why = (word ^ tag) < 4u ? (int)(word & 3u) : 0;
        if (why == 0) return;
        if (why == 1 && bid == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(guard + SC_W_FALLBACKS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // n_dev != null (NB = 1 only): the batch size is a word in device memory, the grid covers the largest batch
    if (n_dev) {
        n = n_full = __builtin_amdgcn_readfirstlane(*n_dev);
        if (bid >= n) return;
    }
    if constexpr (NB >= 2) {
        if (bid >= n_full) {
            tower_body<NB - 1, NT, F16>(lds, n_full * NB + (bid - n_full) * (NB - 1), in88, in_stride, n, stem_wp, tower_wp, fold, blocks, hp,
                                   pi_out, v_out, diag, slot_map);
            return;
        }
    }
    tower_body<NB, NT, F16>(lds, bid * NB, in88, in_stride, n, stem_wp, tower_wp, fold, blocks, hp, pi_out, v_out, diag, slot_map);
}

Bf16Net* bn(azr_engine* h) { return bf16net(h); }
}  // namespace

namespace azr {

int net_bf16_alloc(azr_engine* h)
{
    Bf16Net* x = new Bf16Net();
    h->net.bf16ctx = x;
    const int B = h->net.blocks;
    x->f16 = h->cfg.net_dtype == AZR_NET_F16;
    if (x->f16) HIPCHK(h, hipMalloc((void**)&x->fold16, (14 + (size_t)2 * B * 2 * NF) * sizeof(float)));
    HIPCHK(h, hipMalloc((void**)&x->stem_wp, STEM_HALFS * 2));
    HIPCHK(h, hipMalloc((void**)&x->tower_wp, ((size_t)2 * B * TOWER_LAYER_HALFS + MAX_RING * KSTRIDE * 8) * 2));  // + ring run-off
    HIPCHK(h, hipMemsetAsync(x->tower_wp, 0, ((size_t)2 * B * TOWER_LAYER_HALFS + MAX_RING * KSTRIDE * 8) * 2, h->stream));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_tower_bf16<1, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo<1>::LDS_BYTES));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_tower_bf16<1, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo<1>::LDS_BYTES));
    // test hooks (libazr_hip_test.so only, azr_internal.hpp; read ONCE, here, never in the launch path): AZR_TOWER_SB = 0: the two-image
    // kernel (one board per workgroup) for every launch — the independently written implementation the single-image tiles are compared
    // with bit for bit; 1 (default, and the product): plan; 2 / 3 / 4: force the 4- / 2- / 3-board single-image tile
    x->sb_mode = hook_env_int("AZR_TOWER_SB", 1);
    // AZR_TOWER_SC=0: launches of <= 128 boards on k_tower_bf16<1> instead of the split-channel tower (A/B measurements, tests)
    x->sc_mode = hook_env_int("AZR_TOWER_SC", 1);
    int rc = tower_sc_init(h);
    if (rc) return rc;
    return tower_sb_init(h);
}

void net_bf16_free(azr_engine* h)
{
    if (!h->net.bf16ctx) return;
    Bf16Net* x = bn(h);
    tower_sc_free(h);
    if (x->stem_wp) hipFree(x->stem_wp);
    if (x->tower_wp) hipFree(x->tower_wp);
    if (x->fold16) hipFree(x->fold16);
    delete x;
    h->net.bf16ctx = nullptr;
}

// pack the HWIO fp32 kernels of the AZRW vector into MFMA B-operand fragment order (bf16 or fp16, RNE):
// fragment (layer, tap, ks, wave, nt), lane l, element j  <-  W[tap][ci = ks*32 + 8*(l>>4) + j][co = wave*32 + nt*16 + (l&15)]

// power-of-two scale of one fp16-packed weight tensor: max |2^e w| in [2^13, 2^14) (e clipped to [-2, 24]); false = a weight is
// outside the fp16 range or not a number
static bool f16_scale(const float* W, size_t n, int& e)
{
    float worst = 0.0f;
    for (size_t i = 0; i < n; i++) {
        const float aw = W[i] < 0 ? -W[i] : W[i];
        if (!(aw <= worst)) worst = aw;   // (also catches NaN)
    }
    if (!(worst < 65504.0f)) return false;
    e = 0;
    if (worst > 0.0f) {
        int we;
        frexpf(worst, &we);           // worst = f * 2^we, f in [0.5, 1)
        e = 14 - we;
        if (e > 24) e = 24;
        if (e < -2) e = -2;
    }
    return true;
}
static inline uint16_t f2h(float f) { const _Float16 v = (_Float16)f; uint16_t u; memcpy(&u, &v, 2); return u; }

// packs the stem and tower conv weights into MFMA fragments: bf16 (NET_BF16), or fp16 of 2^k w with k per layer and 2^-k folded into
// the layer's BN scale (NET_F16; `fold_host` = the fp32 fold of net_upload)
int net_bf16_upload(azr_engine* h, const float* fold_host)
{
    Bf16Net* x = bn(h);
    const int B = h->net.blocks;
    const bool f16 = x->f16;
    const float* flat = h->flat.data();
    std::vector<uint16_t> stem(STEM_HALFS, 0), tower((size_t)2 * B * TOWER_LAYER_HALFS);
    std::vector<float> fold(fold_host, fold_host + 14 + (size_t)2 * B * 2 * NF);
    float sS = 1.0f;
    if (f16) {
        int e = 0;
        if (!f16_scale(flat, (size_t)9 * 13 * NF, e)) { h->err = "NET_F16: a stem weight is outside the fp16 range (|w| must be < 65504) or not a number"; return AZR_E_INVALID_ARGUMENT; }
        sS = ldexpf(1.0f, e);
        for (int y = 0; y < 7; y++) fold[y] *= ldexpf(1.0f, -e);
    }
    for (int ks = 0; ks < STEM_KS; ks++)
        for (int w = 0; w < 8; w++)
            for (int nt = 0; nt < 2; nt++)
                for (int l = 0; l < 64; l++)
                    for (int j = 0; j < 8; j++) {
                        const int g = l >> 4, tap = 2 * ks + (g >> 1), ch = (g & 1) * 8 + j, co = w * 32 + nt * 16 + (l & 15);
                        float v = (tap < 9 && ch < 13) ? flat[((size_t)tap * 13 + ch) * NF + co] : 0.0f;
                        stem[((((size_t)ks * 8 + w) * 2 + nt) * 64 + l) * 8 + j] = f16 ? f2h(v * sS) : f2bf(v);
                    }
    const size_t layer_floats = (size_t)9 * NF * NF + 4 * NF;
    const float* t0 = flat + 9 * 13 * NF + 28;
    for (int L = 0; L < 2 * B; L++) {
        const float* W = t0 + (size_t)L * layer_floats;
        float sL = 1.0f;
        if (f16) {
            int e = 0;
            if (!f16_scale(W, (size_t)9 * NF * NF, e)) { h->err = "NET_F16: a conv weight is outside the fp16 range (|w| must be < 65504) or not a number"; return AZR_E_INVALID_ARGUMENT; }
            sL = ldexpf(1.0f, e);
            float* fs = fold.data() + 14 + (size_t)L * 2 * NF;
            for (int i = 0; i < NF; i++) fs[i] *= ldexpf(1.0f, -e);
        }
        uint16_t* dst = tower.data() + (size_t)L * TOWER_LAYER_HALFS;
        for (int tap = 0; tap < 9; tap++)
            for (int ks = 0; ks < 8; ks++)
                for (int w = 0; w < 8; w++)
                    for (int nt = 0; nt < 2; nt++)
                        for (int l = 0; l < 64; l++) {
                            const int ci0 = ks * 32 + 8 * (l >> 4), co = w * 32 + nt * 16 + (l & 15);
                            uint16_t* d = dst + (((((size_t)tap * 8 + ks) * 8 + w) * 2 + nt) * 64 + l) * 8;
                            for (int j = 0; j < 8; j++) {
                                const float wv = W[((size_t)tap * NF + ci0 + j) * NF + co];
                                d[j] = f16 ? f2h(wv * sL) : f2bf(wv);
                            }
                        }
    }
    HIPCHK(h, hipMemcpyAsync(x->stem_wp, stem.data(), stem.size() * 2, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(x->tower_wp, tower.data(), tower.size() * 2, hipMemcpyHostToDevice, h->stream));
    if (f16) HIPCHK(h, hipMemcpyAsync(x->fold16, fold.data(), fold.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return AZR_OK;
}

// Single-image tiles (azr_tower_sb.hip) for 2, 3 or 4 boards per workgroup: boards per workgroup for a launch of n boards,
// or 0 = the two-image kernels (1..3 boards).  AZR_TOWER_SB: 0 = never, 1 = plan (default), 2 / 3 / 4 = force the 4- / 2- /
// 3-board tile for every launch (tests, measurements).
static int plan_sb(int sb_mode, int n)
{
    int snb = sb_mode == 2 ? 4 : sb_mode == 3 ? 2 : sb_mode == 4 ? 3 : 0;
    if (sb_mode == 1 && n > 256) {
        // Relative time of one 256-workgroup round of 2 / 3 / 4 boards per workgroup.  One workgroup per CU is resident, so a launch
        // of w workgroups takes ceil(w / 256) rounds, and only the RATIOS of the round times enter the choice: the three tiles are
        // the same MFMA-bound code, so a box that clocks lower stretches all three alike.  The ratios are those of the tiles'
        // measured shader cycles per workgroup (1.115 / 1.415 / 1.938 M cycles); launch times measured on three
        // boxes of the pool (1.77 - 2.0 GHz sustained): 0.61 / 0.80 / 1.05 ms, 0.63 / 0.82 / 1.07 and 0.68 / 0.89 / 1.18 — the same
        // ratios to 2 %, far inside the margins the plan turns on (the closest call, 768 boards: 3 x 256 at 0.73 against 2 rounds of
        // 2-board tiles at 1.15).
        static const float T[5] = {0.0f, 0.0f, 0.575f, 0.730f, 1.0f};
        float best = 0.0f;
        for (int c = 4; c >= 2; c--) {
            const int w = (n + c - 1) / c;
            const float t = (float)((w + 255) / 256) * T[c];
            if (snb == 0 || t < best) { snb = c; best = t; }
        }
    }
    return n >= snb ? snb : 0;
}

int net_bf16_forward(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st)
{
    Bf16Net* x = bn(h);
    const float* fold = net_fold(h);
    const int B = h->net.blocks;
    if (h->pe_tower0) hipEventRecord(h->pe_tower0, st);
    if (const int snb = plan_sb(x->sb_mode, n)) {   // more than 256 boards: 2, 3 or 4 per workgroup in one LDS image (azr_tower_sb.hip)
        int rc = tower_sb_launch(h, snb, (n + snb - 1) / snb, d_in88, in_stride, n, d_pi, d_v, d_map, st);
        if (h->pe_tower1) hipEventRecord(h->pe_tower1, st);
        return rc;
    }
    unsigned* guard = nullptr;
    if (x->sb_mode != 0 && x->sc_mode != 0 && n <= 128) {   // up to 128 boards: a board pair's channels split over 4 workgroups (azr_tower_sc.hip)
        int rc = tower_sc_launch(h, d_in88, in_stride, n, d_pi, d_v, d_map, st);
        if (h->pe_tower1) hipEventRecord(h->pe_tower1, st);
        if (rc) return rc;
        // ... and right behind it, in stream order, the one-board-per-workgroup kernel under the launch's give-up word: it recomputes the
        // batch if (and only if) a hand-off of the persistent launch ran out of polls; otherwise its n workgroups end at their first
        // instruction.  Later tree steps on this stream therefore never see the garbage of a launch that gave up.
        guard = x->sc_counters;
    }
    // AZR_TOWER_SB=0 / AZR_TOWER_SC=0: one board per workgroup for the whole net, two ping-pong images, 8 waves x 32 channels
    if (x->f16)
        hipLaunchKernelGGL((k_tower_bf16<1, 2, true>), dim3(n), dim3(512), Geo<1>::LDS_BYTES, st, d_in88, in_stride, n, x->stem_wp, x->tower_wp,
                           (const float*)x->fold16, B, net_head_params(h), d_pi, d_v, x->diag, n, d_map, guard, nullptr, x->sc_tag);
    else
        hipLaunchKernelGGL((k_tower_bf16<1, 2, false>), dim3(n), dim3(512), Geo<1>::LDS_BYTES, st, d_in88, in_stride, n, x->stem_wp, x->tower_wp, fold, B,
                           net_head_params(h), d_pi, d_v, x->diag, n, d_map, guard, nullptr, x->sc_tag);
    if (h->pe_tower1 && !guard) hipEventRecord(h->pe_tower1, st);
    HIPCHK(h, hipGetLastError());
    return AZR_OK;
}

// Small batches whose size only the device knows (the arena's and the emptying self-play tail's waiting leaves, counted by the tree step
// into *n_dev ahead of this call in stream order): no read-back, no host synchronisation per pass.  Both launches are sized for n_max
// (<= 256) boards and read the count themselves: the split-channel tower takes up to 128 boards, above that — or when the launch of another
// network beside it (n_other: that one's count) leaves it no CU per workgroup — it raises the give-up word with the value 2 and the guarded
// one-board-per-workgroup launch computes the batch; a count of 0 ends every workgroup at once.
bool net_bf16_counted_ok(azr_engine* h, int n_max)
{
    Bf16Net* x = bn(h);
    return x && x->sb_mode == 1 && x->sc_mode != 0 && n_max >= 1 && n_max <= 256;   // the product's plan (no forced tile of the test build)
}

int net_bf16_forward_counted(azr_engine* h, const uint8_t* d_in88, int in_stride, int n_max, const int* n_dev, const int* n_other, float* d_pi, float* d_v,
                             const int* d_map, hipStream_t st)
{
    if (!net_bf16_counted_ok(h, n_max) || !n_dev) { h->err = "net_bf16_forward_counted: 1..256 boards on the split-channel tower"; return AZR_E_INVALID_ARGUMENT; }
    Bf16Net* x = bn(h);
    const int B = h->net.blocks;
    if (h->pe_tower0) hipEventRecord(h->pe_tower0, st);
    int rc = tower_sc_launch(h, d_in88, in_stride, n_max, d_pi, d_v, d_map, st, n_dev, n_other);
    if (h->pe_tower1) hipEventRecord(h->pe_tower1, st);
    if (rc) return rc;
    if (x->f16)
        hipLaunchKernelGGL((k_tower_bf16<1, 2, true>), dim3(n_max), dim3(512), Geo<1>::LDS_BYTES, st, d_in88, in_stride, n_max, x->stem_wp, x->tower_wp,
                           (const float*)x->fold16, B, net_head_params(h), d_pi, d_v, (unsigned long long*)nullptr, n_max, d_map, x->sc_counters, n_dev, x->sc_tag);
    else
        hipLaunchKernelGGL((k_tower_bf16<1, 2, false>), dim3(n_max), dim3(512), Geo<1>::LDS_BYTES, st, d_in88, in_stride, n_max, x->stem_wp, x->tower_wp,
                           net_fold(h), B, net_head_params(h), d_pi, d_v, (unsigned long long*)nullptr, n_max, d_map, x->sc_counters, n_dev, x->sc_tag);
    HIPCHK(h, hipGetLastError());
    return AZR_OK;
}


}  // namespace azr

// Diagnostics (not part of the product path).  azr_debug_tower_clock: sustained shader clock of the tower kernel under load
// = d(s_memtime) / d(s_memrealtime) x 100 MHz around workgroup 0's whole tower, after `warm` back-to-back launches on the
// leaf buffers.  azr_debug_tower_trace: per workgroup of that launch (k_tower_sb4 only) 8 words: 100 MHz real-time at kernel
// start, tower start, tower end, kernel end, the XCC id, and the shader-clock counter at tower start / end.
static int tower_diag_run(azr_engine* h, int n, int warm, std::vector<unsigned long long>& v)
{
    if (!h || !h->net.bf16ctx || !h->weights_set) return AZR_E_STATE;
    Bf16Net* x = bn(h);
    unsigned long long* d = nullptr;
    const size_t words = 8 + 8 * (size_t)(n > 0 ? n : 1);
    HIPCHK(h, hipMalloc((void**)&d, words * sizeof(unsigned long long)));
    HIPCHK(h, hipMemsetAsync(d, 0, words * sizeof(unsigned long long), h->stream));
    for (int i = 0; i < warm; i++) net_bf16_forward(h, h->d.leaf_in, LEAF_STRIDE, n, h->d.net_pi, h->d.net_v, nullptr, h->stream);
    x->diag = d;
    int rc = net_bf16_forward(h, h->d.leaf_in, LEAF_STRIDE, n, h->d.net_pi, h->d.net_v, nullptr, h->stream);
    x->diag = nullptr;
    v.assign(words, 0);
    HIPCHK(h, hipMemcpyAsync(v.data(), d, words * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipFree(d);
    return rc;
}

extern "C" int azr_debug_tower_plan(azr_engine* h, int n, int* boards_per_wg, int* wgs)
{
    if (!h || !h->net.bf16ctx || n < 1) return AZR_E_STATE;
    const int snb = plan_sb(bn(h)->sb_mode, n);
    if (snb) { *boards_per_wg = snb; *wgs = (n + snb - 1) / snb; return AZR_OK; }
    if (bn(h)->sb_mode != 0 && bn(h)->sc_mode != 0 && n <= 128) {   // split-channel tower: 4 workgroups of 64 channels per board pair
        *wgs = ((n + 1) / 2) * 4;
        *boards_per_wg = 2;
        return AZR_OK;
    }
    *wgs = n;   // one board per workgroup (k_tower_bf16<1>)
    *boards_per_wg = 1;
    return AZR_OK;
}

extern "C" int azr_debug_tower_clock(azr_engine* h, int n, int warm, double* ghz_out, double* tower_ms_out)
{
    std::vector<unsigned long long> v;
    int rc = tower_diag_run(h, n, warm, v);
    if (rc) return rc;
    const double cyc = (double)(v[2] - v[0]), rt = (double)(v[3] - v[1]);
    if (ghz_out) *ghz_out = rt > 0 ? cyc / rt * 0.1 : 0.0;
    if (tower_ms_out) *tower_ms_out = rt * 1e-5;  // 100 MHz ticks -> ms
    return AZR_OK;
}

extern "C" int azr_debug_tower_trace(azr_engine* h, int n, int warm, unsigned long long* out8, int cap_wgs, int* wgs_out)
{
    std::vector<unsigned long long> v;
    int rc = tower_diag_run(h, n, warm, v);
    if (rc) return rc;
    int wgs = 0;
    for (int i = 0; i < n && i < cap_wgs; i++) {
        if (v[8 + 8 * (size_t)i] == 0) break;
        for (int k = 0; k < 8; k++) out8[8 * (size_t)i + k] = v[8 + 8 * (size_t)i + k];
        wgs++;
    }
    if (wgs_out) *wgs_out = wgs;
    return AZR_OK;
}
