// Launchers of the column-owner backward-data kernels of a residual block (dz, dx, dx masked by the LeakyReLU in front of the stack).
// The kernel and its design notes: wn_col_dev.h.
#include "wn_col_dev.h"

namespace wn {

hipError_t launch_hcol(int prec, int epi, const HColArgs& a_in, hipStream_t st) {
    if (a_in.nunit <= 0 || a_in.nks <= 0) return hipSuccess;
    if (prec != HP_BF16 && prec != HP_F16) return hipErrorInvalidValue;
    HColArgs a = a_in;
    static const int dbg = getenv("WN_COL_DBG") ? atoi(getenv("WN_COL_DBG")) : 0;   // measurement: 1 = no stores
    a.dbg = dbg;
    a.nwg = (a.nunit + 3) / 4;
    const int nwg = a.nwg;
    const unsigned grid = (unsigned)(((nwg + 7) / 8) * 8);
    const bool bf = prec == HP_BF16;
    // dz: K = skip tiles [+ z tiles] (both = NT at equal padded widths): NKS = 2 NT or 4 NT;  dx: K = 4 or 5 segments of NT tiles: 8 NT or 10 NT
#define WN_LAUNCH_COL(NT_, NKS_, EPI_)                                                                                    \
    do {                                                                                                                  \
        if (bf) hipLaunchKernelGGL((hcol_kernel<true, NT_, NKS_, EPI_>), dim3(grid), dim3(256), 0, st, a);                \
        else hipLaunchKernelGGL((hcol_kernel<false, NT_, NKS_, EPI_>), dim3(grid), dim3(256), 0, st, a);                  \
        return hipGetLastError();                                                                                         \
    } while (0)
    const int f = a.nks / a.nt;         // k-steps per row tile: 2, 4 (dz) or 8, 10 (dx)
    if (a.nks != f * a.nt) return hipErrorInvalidValue;
    if (epi == HEPI_DGATE && (f == 2 || f == 4)) {
        switch (a.nt * 16 + f) {
            case 1 * 16 + 2: WN_LAUNCH_COL(1, 2, HEPI_DGATE);  case 1 * 16 + 4: WN_LAUNCH_COL(1, 4, HEPI_DGATE);
            case 2 * 16 + 2: WN_LAUNCH_COL(2, 4, HEPI_DGATE);  case 2 * 16 + 4: WN_LAUNCH_COL(2, 8, HEPI_DGATE);
            case 3 * 16 + 2: WN_LAUNCH_COL(3, 6, HEPI_DGATE);  case 3 * 16 + 4: WN_LAUNCH_COL(3, 12, HEPI_DGATE);
            case 4 * 16 + 2: WN_LAUNCH_COL(4, 8, HEPI_DGATE);  case 4 * 16 + 4: WN_LAUNCH_COL(4, 16, HEPI_DGATE);
        }
    } else if (epi == HEPI_LEAKY && (f == 8 || f == 10)) {      // dx of the stack's first block, masked by the LeakyReLU in front of it
        switch (a.nt * 16 + f) {
            case 1 * 16 + 8: WN_LAUNCH_COL(1, 8, kCEpiLeakyBwd);   case 1 * 16 + 10: WN_LAUNCH_COL(1, 10, kCEpiLeakyBwd);
            case 2 * 16 + 8: WN_LAUNCH_COL(2, 16, kCEpiLeakyBwd);  case 2 * 16 + 10: WN_LAUNCH_COL(2, 20, kCEpiLeakyBwd);
            case 3 * 16 + 8: WN_LAUNCH_COL(3, 24, kCEpiLeakyBwd);  case 3 * 16 + 10: WN_LAUNCH_COL(3, 30, kCEpiLeakyBwd);
            case 4 * 16 + 8: WN_LAUNCH_COL(4, 32, kCEpiLeakyBwd);  case 4 * 16 + 10: WN_LAUNCH_COL(4, 40, kCEpiLeakyBwd);
        }
    } else if (epi == HEPI_STORE && (f == 8 || f == 10)) {
        switch (a.nt * 16 + f) {
            case 1 * 16 + 8: WN_LAUNCH_COL(1, 8, HEPI_STORE);   case 1 * 16 + 10: WN_LAUNCH_COL(1, 10, HEPI_STORE);
            case 2 * 16 + 8: WN_LAUNCH_COL(2, 16, HEPI_STORE);  case 2 * 16 + 10: WN_LAUNCH_COL(2, 20, HEPI_STORE);
            case 3 * 16 + 8: WN_LAUNCH_COL(3, 24, HEPI_STORE);  case 3 * 16 + 10: WN_LAUNCH_COL(3, 30, HEPI_STORE);
            case 4 * 16 + 8: WN_LAUNCH_COL(4, 32, HEPI_STORE);  case 4 * 16 + 10: WN_LAUNCH_COL(4, 40, HEPI_STORE);
        }
    }
#undef WN_LAUNCH_COL
    return hipErrorInvalidValue;
}

}  // namespace wn
