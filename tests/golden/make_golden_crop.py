"""Golden fixtures for ``Up`` when the skip is SMALLER than the upsampled tensor: ``F.pad`` with negative widths crops
(``src/Experiments/model_parts.py:73-80``; left = diff // 2 with Python floor division, so -1 -> crop 1 left / 0 right).

Run only in the build container (imports the real reference from /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_crop.py

Same recipe as make_golden.py's block fixtures (generator-defined weights and inputs, train-mode forward + backward,
eval-mode forward); only data is stored.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import make_golden as G  # noqa: E402  (helpers; importing it also puts the reference on sys.path)


def main():
    import torch
    torch.manual_seed(0)
    u, RP = G.u, G.RP
    # both axes cropped: 2*(6,10) = (12,20) against an (11,19) skip -> diff -1,-1: one row / column dropped at the top / left
    G.block_fixture("block_up_crop", RP.Up(16, 8, bilinear=False), [u(31, (2, 16, 6, 10)), u(32, (2, 8, 11, 19))], 3100)
    # rows cropped, columns padded: 2*(5,9) = (10,18) against (9,21) -> diffY -1, diffX +3 (1 left, 2 right)
    G.block_fixture("block_up_crop_mixed", RP.Up(16, 8, bilinear=False), [u(33, (1, 16, 5, 9)), u(34, (1, 8, 9, 21))], 3200)
    # bilinear upsampling, uneven crops: (12,20) against (10,17) -> diffY -2 (1,1), diffX -3 (2 left, 1 right)
    G.block_fixture("block_up_bilinear_crop", RP.Up(16, 8, bilinear=True), [u(35, (2, 8, 6, 10)), u(36, (2, 8, 10, 17))], 3300)


if __name__ == "__main__":
    main()
