from .block import GatedActivationUnit, ResidualBlock  # noqa: F401
from .classifier import WaveNetClassifier  # noqa: F401
from .conv_ops import CausalConv1d, NonCausalConv1d, autopad, reshape_in, reshape_out  # noqa: F401
from .raw_ctcnet import RawCTCNet  # noqa: F401
from .wavenet import WaveNet  # noqa: F401
