// Launcher of the column-owner kernel for skips_sum = sum_l W_skip,l' z_l written as leaky(S) * scale into the half series (the input
// of the output block, reference modules/wavenet.py:100,103 / raw_ctcnet.py:143-151): one long-K product, K = the z of up to 16
// equally wide blocks (<= 128 channels), streamed through the 16-fragment register window.  The tiled hgemm_kernel runs this product
// at 2.9 TB/s of its 402 MB at cfg2 (one round of 544 tiles, each a serial 88-step K loop); the kernel and its notes: wn_col_dev.h.
#include "wn_col_dev.h"

namespace wn {

hipError_t launch_hcol_skipsum(int prec, const HColArgs& a_in, hipStream_t st) {
    if (a_in.nunit <= 0 || a_in.nks <= 0) return hipSuccess;
    if (prec != HP_BF16 && prec != HP_F16) return hipErrorInvalidValue;
    HColArgs a = a_in;
    a.dbg = 0;
    a.nwg = (a.nunit + 3) / 4;
    const unsigned grid = (unsigned)(((a.nwg + 7) / 8) * 8);
    const bool bf = prec == HP_BF16;
#define WN_LAUNCH_SKIP(NT_, NB_)                                                                                                  \
    case NB_:                                                                                                                     \
        if (bf) hipLaunchKernelGGL((hcol_kernel<true, NT_, 2 * NT_ * NB_, kCEpiLeakyFwd>), dim3(grid), dim3(256), 0, st, a);      \
        else hipLaunchKernelGGL((hcol_kernel<false, NT_, 2 * NT_ * NB_, kCEpiLeakyFwd>), dim3(grid), dim3(256), 0, st, a);        \
        return hipGetLastError();
#define WN_LAUNCH_SKIP_ALL(NT_)                                                                                                   \
    switch (nb) {                                                                                                                 \
        WN_LAUNCH_SKIP(NT_, 1) WN_LAUNCH_SKIP(NT_, 2) WN_LAUNCH_SKIP(NT_, 3) WN_LAUNCH_SKIP(NT_, 4) WN_LAUNCH_SKIP(NT_, 5)        \
        WN_LAUNCH_SKIP(NT_, 6) WN_LAUNCH_SKIP(NT_, 7) WN_LAUNCH_SKIP(NT_, 8) WN_LAUNCH_SKIP(NT_, 9) WN_LAUNCH_SKIP(NT_, 10)       \
        WN_LAUNCH_SKIP(NT_, 11) WN_LAUNCH_SKIP(NT_, 12) WN_LAUNCH_SKIP(NT_, 13) WN_LAUNCH_SKIP(NT_, 14) WN_LAUNCH_SKIP(NT_, 15)   \
        WN_LAUNCH_SKIP(NT_, 16)                                                                                                   \
    }
    if (a.nt != 2 && a.nt != 4) return hipErrorInvalidValue;
    if (a.nks % (2 * a.nt) != 0) return hipErrorInvalidValue;
    const int nb = a.nks / (2 * a.nt);
    if (a.nt == 4) { WN_LAUNCH_SKIP_ALL(4) } else { WN_LAUNCH_SKIP_ALL(2) }
#undef WN_LAUNCH_SKIP_ALL
#undef WN_LAUNCH_SKIP
    return hipErrorInvalidValue;
}

}  // namespace wn
