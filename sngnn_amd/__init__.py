"""sngnn_amd - MI355X-native implementation of SNGNN's similarity-navigated
aggregation path behind the reference's own nn.Module API.

The aggregation runs in hand-written HIP (libsngnn_hip.so, C ABI in
include/sngnn_hip.h); there is no CPU fallback."""
from .conv import AGNNConv, SNConv, SNConv_plus, SNConv_plus_plus
from .ggcn import GGCNlayer_SP
from .models import AGNN, SNGNN, SNGNN_Plus, SNGNN_Plus_Plus
from .synth import Data

__all__ = ["SNConv", "SNConv_plus", "SNConv_plus_plus", "SNGNN", "SNGNN_Plus",
           "SNGNN_Plus_Plus", "AGNNConv", "AGNN", "GGCNlayer_SP", "Data"]
