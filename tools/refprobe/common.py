"""Shared setup for the golden-vector generators.

These scripts IMPORT the reference (osudrl/apex at /root/reference) in this container only, run its own
functions on seeded inputs and write small .npz fixtures (inputs + expected outputs) to tests/golden/.
Nothing from the reference is copied; the GPU box never sees /root/reference.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")


def setup_reference_path():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; golden generators only run in the build container")
    sys.path[:0] = [os.path.join(HERE, "stubs"), REF, os.path.join(REPO, "tests")]      # tests/: golden_util (seeded inputs, slim records)
    os.makedirs(GOLD, exist_ok=True)


MIRRORED_OBS_FULL_CLOCK = [0.1, 1, -2, 3, -4, -10, -11, 12, 13, 14, -5, -6, 7, 8, 9, 15, -16, 17, -18, 19, -20,
                           -26, -27, 28, 29, 30, -21, -22, 23, 24, 25, 31, -32, 33, 37, 38, 39, 34, 35, 36,
                           43, 44, 45, 40, 41, 42, 46, 47, 48, 49]
MIRRORED_ACTS = [-5, -6, 7, 8, 9, -0.1, -1, 2, 3, 4]
