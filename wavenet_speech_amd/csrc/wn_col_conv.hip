// Launcher of the column-owner kernel for the 1x1 convolutions that stay in the half series (output block, feature layer:
// reference modules/wavenet.py:67-71, raw_ctcnet.py:57-61,89-93): forward with the LeakyReLU fused into the epilogue, backward-data
// with its derivative.  <= 128 channels on both sides, one-plane modes.  The kernel and its design notes: wn_col_dev.h.
#include "wn_col_dev.h"

namespace wn {

hipError_t launch_hcol_conv(int prec, bool backward, const HColArgs& a_in, hipStream_t st) {
    if (a_in.nunit <= 0 || a_in.nks <= 0) return hipSuccess;
    if (prec != HP_BF16 && prec != HP_F16) return hipErrorInvalidValue;
    HColArgs a = a_in;
    a.dbg = 0;
    a.nwg = (a.nunit + 3) / 4;
    const unsigned grid = (unsigned)(((a.nwg + 7) / 8) * 8);
    const bool bf = prec == HP_BF16;
#define WN_LAUNCH_CONV(NT_, NKS_)                                                                                                 \
    do {                                                                                                                          \
        if (backward) {                                                                                                           \
            if (bf) hipLaunchKernelGGL((hcol_kernel<true, NT_, NKS_, kCEpiLeakyBwd>), dim3(grid), dim3(256), 0, st, a);           \
            else hipLaunchKernelGGL((hcol_kernel<false, NT_, NKS_, kCEpiLeakyBwd>), dim3(grid), dim3(256), 0, st, a);             \
        } else {                                                                                                                  \
            if (bf) hipLaunchKernelGGL((hcol_kernel<true, NT_, NKS_, kCEpiLeakyFwd>), dim3(grid), dim3(256), 0, st, a);           \
            else hipLaunchKernelGGL((hcol_kernel<false, NT_, NKS_, kCEpiLeakyFwd>), dim3(grid), dim3(256), 0, st, a);             \
        }                                                                                                                         \
        return hipGetLastError();                                                                                                 \
    } while (0)
    if (a.nt < 1 || a.nt > 4 || a.nks < 2 || a.nks > 8 || (a.nks & 1)) return hipErrorInvalidValue;
    switch (a.nt * 16 + a.nks) {
        case 1 * 16 + 2: WN_LAUNCH_CONV(1, 2);  case 1 * 16 + 4: WN_LAUNCH_CONV(1, 4);  case 1 * 16 + 6: WN_LAUNCH_CONV(1, 6);  case 1 * 16 + 8: WN_LAUNCH_CONV(1, 8);
        case 2 * 16 + 2: WN_LAUNCH_CONV(2, 2);  case 2 * 16 + 4: WN_LAUNCH_CONV(2, 4);  case 2 * 16 + 6: WN_LAUNCH_CONV(2, 6);  case 2 * 16 + 8: WN_LAUNCH_CONV(2, 8);
        case 3 * 16 + 2: WN_LAUNCH_CONV(3, 2);  case 3 * 16 + 4: WN_LAUNCH_CONV(3, 4);  case 3 * 16 + 6: WN_LAUNCH_CONV(3, 6);  case 3 * 16 + 8: WN_LAUNCH_CONV(3, 8);
        case 4 * 16 + 2: WN_LAUNCH_CONV(4, 2);  case 4 * 16 + 4: WN_LAUNCH_CONV(4, 4);  case 4 * 16 + 6: WN_LAUNCH_CONV(4, 6);  case 4 * 16 + 8: WN_LAUNCH_CONV(4, 8);
    }
#undef WN_LAUNCH_CONV
    return hipErrorInvalidValue;
}

}  // namespace wn
