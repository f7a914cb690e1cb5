"""Golden fixtures at the BASELINE configs' exact channel widths but reduced spatial size, from the REAL reference:
  net_spectral1650_small   SpectralUNET(238, 1, 1650) (config C3's widths: 1650 -> padded 1664, concat 3300) @ (2,238,16,24)
  net_cubenet128_300_small CubeNET(300, 1, first_depth=128) (config C5's widths: first conv K = 2700) @ (2,1,300,32,48)
  net_spectral_3class      SpectralUNET(10, 3, 4): n_classes != 1, where models.py:144 reshapes the (R*C, n_classes) result
                           as (n_classes, R, C) -- the drop-in must reproduce exactly that element order @ (2,10,7,9)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_widths.py      (build container only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as MG  # noqa: E402  (sets sys.path for the oracle and the reference)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m = (MG.u(4330, (2, 1, 16, 24)) > 0.8).float()
    MG.net_fixture("net_spectral1650_small", MG.RM.SpectralUNET(238, 1, 1650), MG.u(1250, (2, 238, 16, 24)), m)
    m = (MG.u(4332, (2, 3, 7, 9)) > 0.7).float()
    MG.net_fixture("net_spectral_3class", MG.RM.SpectralUNET(10, 3, 4), MG.u(1252, (2, 10, 7, 9)), m)
    m = (MG.u(4331, (2, 1, 32, 48)) > 0.9).float()
    MG.net_fixture("net_cubenet128_300_small", MG.RM.CubeNET(300, 1, first_depth=128, bilinear=False),
                   MG.u(1251, (2, 1, 300, 32, 48)), m)


if __name__ == "__main__":
    main()
