// azr_rowclass.hpp — border-class row order of the board cells of a workgroup tile (shared by the inference tower,
// azr_tower_sb.hip, and the training conv kernels, azr_train.hip).
//
// A 3x3 SAME conv on the 6x7 board is an implicit GEMM whose M rows are board cells; a tap (dy, dx) has no in-board source
// for the cells of one board edge.  If the rows of a workgroup's tile are ordered so that whole 16-row MFMA tiles lie on one
// edge, the (tile, tap) pairs without any source are skipped altogether — exact zeros left out, results unchanged.
#pragma once
#include <stdint.h>

namespace azr {

// Row order = border classes, corners counted with the COLUMN classes: a cell with x = 0 has no in-board source under the
// three taps with dx = -1 whatever its y, so all 7 x = 0 cells of a board (corners included) share tiles, likewise x = 5;
// the y = 0 / y = 6 classes are the 4 non-corner cells.  Pad rows (no cell: zero under every tap) top up class tiles.
//   NB = 4 (176 rows):  tile 0 = 16 cells x=0 | 1 = 16 cells x=5 | 2 = the other 12 x=0 + 4 pads | 3 = the other 12 x=5 + 4 pads
//                       | 4 = 16 cells y=0 | 5 = 16 cells y=6 | 6..10 = the 80 interior cells          => 18 of 99 tile-taps skipped
//   NB = 3 (128 rows):  the corners go where they complete a tile: tile 0 = 16 of the 18 cells y=0 (all but board 0's corners)
//                       | 1 = 16 of the 18 cells y=6 | 2 = the 15 cells x=0, y=1..5 + board 0's (0,0) | 3 = same for x=5 with (5,0)
//                       | 4..7 = board 0's (0,6) and (5,6), the 60 interior cells, 2 pads              => 12 of 72
//   NB = 2 (96 rows):   tile 0 = 14 cells x=0 + 2 pads | 1 = 14 cells x=5 + 2 pads | 2 = 8 cells y=0 + 8 pads
//                       | 3 = 8 cells y=6 + 8 interior | 4, 5 = interior                               =>  9 of 54
template <int NB>
__host__ __device__ __forceinline__ constexpr int row_of(int b, int pos)
{
    const int y = pos / 6, x = pos - y * 6;
    if (NB == 4) {
        if (x == 0) { const int q = b * 7 + y; return q < 16 ? q : 32 + (q - 16); }
        if (x == 5) { const int q = b * 7 + y; return q < 16 ? 16 + q : 48 + (q - 16); }
        if (y == 0) return 64 + b * 4 + (x - 1);
        if (y == 6) return 80 + b * 4 + (x - 1);
        return 96 + b * 20 + (y - 1) * 4 + (x - 1);
    } else if (NB == 3) {
        const bool corner0 = b == 0 && (x == 0 || x == 5);          // board 0's corners complete the column tiles
        if (y == 0) return corner0 ? (x == 0 ? 47 : 63) : (b == 0 ? x - 1 : 4 + (b - 1) * 6 + x);
        if (y == 6) return corner0 ? (x == 0 ? 64 : 65) : 16 + (b == 0 ? x - 1 : 4 + (b - 1) * 6 + x);
        if (x == 0) return 32 + b * 5 + (y - 1);
        if (x == 5) return 48 + b * 5 + (y - 1);
        return 66 + b * 20 + (y - 1) * 4 + (x - 1);
    } else {
        if (x == 0) return b * 7 + y;
        if (x == 5) return 16 + b * 7 + y;
        if (y == 0) return 32 + b * 4 + (x - 1);
        if (y == 6) return 48 + b * 4 + (x - 1);
        return 56 + b * 20 + (y - 1) * 4 + (x - 1);
    }
}
// bit mt set = tile mt has no in-board source cell under this tap (tap 9 = "no tap": everything skipped)
template <int NB>
__host__ __device__ constexpr uint32_t skip_mask(int tap)
{
    if (tap > 8) return 0xffffffffu;
    const int ty = tap / 3, tx = tap - 3 * ty;
    if (NB == 4) return (tx == 0 ? 0x5u : tx == 2 ? 0xAu : 0u) | (ty == 0 ? 0x10u : ty == 2 ? 0x20u : 0u);
    if (NB == 3) return (ty == 0 ? 0x1u : ty == 2 ? 0x2u : 0u) | (tx == 0 ? 0x4u : tx == 2 ? 0x8u : 0u);
    return (tx == 0 ? 0x1u : tx == 2 ? 0x2u : 0u) | (ty == 0 ? 0x4u : 0u);
}
// first pad lane (fragment column) of tile mt; 16 = the tile has no pad rows
template <int NB>
__host__ __device__ constexpr int pad_from(int mt)
{
    if (NB == 4) return (mt == 2 || mt == 3) ? 12 : 16;
    if (NB == 3) return mt == 7 ? 14 : 16;
    return mt < 2 ? 14 : mt == 2 ? 8 : 16;
}

}  // namespace azr
