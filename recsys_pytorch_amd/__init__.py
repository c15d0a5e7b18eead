"""recsys_pytorch_amd -- MI355X-native BPR-MF training step + Top-N scoring,
behind the model interface of yoongi0428/RecSys_PyTorch (models/BaseModel.py,
models/MF.py).  Arithmetic lives in librsx.so (hand-written HIP, include/rsx.h);
this package is the host-side mirror of the reference interface."""
__version__ = "0.1.0"
