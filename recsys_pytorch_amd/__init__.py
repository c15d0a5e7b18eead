"""recsys_pytorch_amd -- MI355X-native BPR-MF training step + Top-N scoring,
behind the model interface of yoongi0428/RecSys_PyTorch (models/BaseModel.py,
models/MF.py).  Arithmetic lives in librsx.so (hand-written HIP, include/rsx.h);
this package is the host-side mirror of the reference interface."""
__version__ = "0.1.0"

from .mf import BaseModel, MF            # noqa: E402,F401  (registry: getattr(recsys_pytorch_amd, 'MF'))
from .lightgcn import LightGCN           # noqa: E402,F401
from .evaluator import Evaluator         # noqa: E402,F401
from .data import InteractionData        # noqa: E402,F401
