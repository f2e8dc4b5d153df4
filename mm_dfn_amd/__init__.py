"""mm_dfn_amd: MI355X-native (gfx950) implementation of the MM-DFN hot path.

Drop-in ``nn.Module`` counterparts of the reference's GraphConvolution /
GCNII_lyc (model_GCN.py), MM_GCN (model_mm.py), DialogueGNNModel (model.py) and
FocalLoss (loss.py) whose compute is hand-written HIP behind a C ABI
(include/mmdfn_hip.h).  See DESIGN.md.
"""
from .layout import DialogueLayout, BlockTileAdjacency  # noqa: F401
from .graph_conv import GraphConvolution, GCNII_lyc, GCNII  # noqa: F401
from .mm_gcn import MM_GCN  # noqa: F401
from .dialogue_model import DialogueGNNModel  # noqa: F401
from .loss import FocalLoss  # noqa: F401
from .fusion import MFN, MMGatedAttention  # noqa: F401
from .multistream import MultiStreamGraphModel  # noqa: F401
