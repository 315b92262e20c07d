"""
wavenet_speech_amd -- MI355X-native WaveNet dilated residual-block stack behind the nn.Module surface of
paultsw/wavenet-speech (modules.wavenet.WaveNet, modules.raw_ctcnet.RawCTCNet, modules.block.ResidualBlock,
modules.classifier.WaveNetClassifier, modules.conv_ops.*).

    from wavenet_speech_amd.modules.wavenet import WaveNet      # instead of: from modules.wavenet import WaveNet

All arithmetic of the hot path runs in hand-written HIP kernels (csrc/, gfx950) reached through the C ABI of
libwavenet_amd.so (include/wavenet_amd.h).  No CPU fallback exists.
"""
from . import functional, graphs, modules, series  # noqa: F401
from .modules import (CausalConv1d, NonCausalConv1d, RawCTCNet, ResidualBlock, WaveNet,  # noqa: F401
                      WaveNetClassifier)
from .modules.block import freeze_for_inference, set_precision  # noqa: F401
from ._flags import check_device_flags  # noqa: F401
from .graphs import GraphedStep  # noqa: F401
from .functional_half import check_fp16_overflow  # noqa: F401

__version__ = "0.1.0"
