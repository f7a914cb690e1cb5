"""Golden fixtures for SpectralUNET(bnorm=False) from the REAL reference (models.py:72,105-110: Linear -> ReLU stages without
BatchNorm1d -- the constructor argument no HyperPRI experiment sets, still part of the class API):
  net_spectral_nobn_tiny   SpectralUNET(10, 1, 4, bnorm=False)   @ (3,10,7,9)
  net_spectral_nobn_f50    SpectralUNET(22, 1, 50, bnorm=False)  @ (2,22,9,14)   (a width that is no multiple of 4 / 8 / 32)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_nobn.py      (build container only)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as MG  # noqa: E402  (sets sys.path for the oracle and the reference)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m = (MG.u(4322, (3, 1, 7, 9)) > 0.7).float()
    MG.net_fixture("net_spectral_nobn_tiny", MG.RM.SpectralUNET(10, 1, 4, bnorm=False), MG.u(1237, (3, 10, 7, 9)), m)
    m = (MG.u(4324, (2, 1, 9, 14)) > 0.7).float()
    MG.net_fixture("net_spectral_nobn_f50", MG.RM.SpectralUNET(22, 1, 50, bnorm=False), MG.u(1242, (2, 22, 9, 14)), m)


if __name__ == "__main__":
    main()
