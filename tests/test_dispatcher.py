"""The hot path as PyTorch custom operators (BASELINE.json north_star: "registered as PyTorch-ROCm custom ops"): one
``torch.ops.hyperpri.<module>`` per reference module, registered for the CUDA (ROCm) key only.  CPU part: the operators exist in the
dispatcher with the documented schema, and there is no CPU kernel behind them (no silent fallback)."""
import pytest
import torch

import hyperpri_amd  # noqa: F401  (registers the operators)
from hyperpri_amd import autograd as A


def test_operators_are_registered_with_the_dispatcher():
    for name in A.OP_NAMES:
        op = getattr(torch.ops.hyperpri, name)
        schema = str(op.default._schema)
        assert schema.startswith(f"hyperpri::{name}(Tensor[] inputs, Tensor[] params, SymInt program, bool grad_mode, SymInt input_planes) -> Tensor"), schema
        # a kernel for the CUDA dispatch key (ROCm devices are "cuda" to PyTorch) and an autograd formula; nothing for the CPU key
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"hyperpri::{name}", "CUDA")
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"hyperpri::{name}", "Autograd")
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f"hyperpri::{name}", "CPU")


def test_no_cpu_kernel_behind_the_operators():
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(NotImplementedError):
        torch.ops.hyperpri.unet([x], [], 0, False, 0)
    # ... and the modules say so in their own words before they get that far
    net = hyperpri_amd.UNet(3, 1, bilinear=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(x)
