"""unimm_amd -- MI355X-native (gfx950) implementation of the UniMM-UL forward/backward hot path.

Drop-in surface (same names / signatures as the reference):
    from unimm_amd import VisualDialogEncoder, BertForMultiModalPreTraining, BertConfig
"""
from .config import BertConfig
from .modeling import BertForMultiModalPreTraining, VisualDialogEncoder

__version__ = "0.1.0"
__all__ = ["BertConfig", "BertForMultiModalPreTraining", "VisualDialogEncoder"]
