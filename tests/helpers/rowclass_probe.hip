// prints the border-class row tables of csrc/azr_rowclass.hpp (host side of the same constexpr functions the kernels use):
//   NB MT | row_of[NB*42] | skip_mask[9] | pad_from[MT]
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "azr_rowclass.hpp"

template <int NB>
static void dump()
{
    const int MT = (42 * NB + 15) / 16;
    printf("%d %d |", NB, MT);
    for (int b = 0; b < NB; b++)
        for (int pos = 0; pos < 42; pos++) printf(" %d", azr::row_of<NB>(b, pos));
    printf(" |");
    for (int t = 0; t < 9; t++) printf(" %u", azr::skip_mask<NB>(t));
    printf(" |");
    for (int mt = 0; mt < MT; mt++) printf(" %d", azr::pad_from<NB>(mt));
    printf("\n");
}

int main()
{
    dump<2>();
    dump<3>();
    dump<4>();
    return 0;
}
