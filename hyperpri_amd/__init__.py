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


def _warn_removed_switches():
    """Environment switches of earlier rounds that no longer exist (folded into HPRI_FUSIONS or dropped): setting one is a silent
    no-op otherwise."""
    import os
    import sys
    gone = ("HPRI_FUSE_BN_REDUCE", "HPRI_FUSE_BN_REDUCE_BF16", "HPRI_CONVT_PLANES", "HPRI_GRAD_BF16_INNER", "HPRI_PLANES_CAT1", "HPRI_YR_BF16",
            "HPRI_WINO4", "HPRI_BUCKET_MB", "HPRI_TAIL_MB")
    hit = [v for v in gone if v in os.environ]
    if hit:
        print("hyperpri_amd: " + ", ".join(hit) + " no longer exist" + ("s" if len(hit) == 1 else "") + ": the per-feature switches are module "
              "attributes of hyperpri_amd.engine under the HPRI_FUSIONS master switch (INTEGRATION.md); GradSync takes bucket_mb / tail_mb "
              "as arguments.", file=sys.stderr)


_warn_removed_switches()
