"""hyperpri_amd -- MI355X-native (gfx950) implementation of the HyperPRI segmentation hot path.

Drop-in ``torch.nn.Module`` replacements for the reference's ``src/Experiments/models.py`` and
``model_parts.py`` whose forward/backward run in hand-written HIP kernels.  See DESIGN.md.
"""
from .model_parts import DoubleConv, Down, OutConv, Up, set_precision  # noqa: F401
from .models import (CubeNET, SpectralUNET, UNet, initialize_model, set_parameter_requires_grad,  # noqa: F401
                     translate_load_dir)
from .trainer import (BCEWithLogitsLoss, forward_loss, FusedAdam, FusedSGD, PRCurve, SegCounts, SegmentationModel,  # noqa: F401
                      average_precision, load_checkpoint, network_state_dict)

__version__ = "0.1.0"
