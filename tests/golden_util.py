"""Helpers shared by the golden generators (tools/refprobe/gen_golden_*.py, this container only) and the tests that replay them.

The BASELINE-size fixtures (LSTM 2 x 128, TD3 256-unit nets) would weigh several MB each if every parameter tensor were stored, so they
hold (a) the SEED of their input parameters, drawn here from numpy's frozen legacy generator (RandomState streams are guaranteed stable
across numpy versions), and (b) a strided subsample + sum + L2 norm of every large output tensor.  Inputs are regenerated on the test
side with the same function the generator used to load the reference's modules."""
import numpy as np

STRIDE = 16


def seeded_params(shapes, seed):
    """Parameter list for `shapes` (state_dict order): U(-k, k) with k = 1 / sqrt(fan_in of the last weight seen)."""
    rs = np.random.RandomState(seed)
    out, k = [], 1.0
    for s in shapes:
        s = tuple(int(x) for x in s)
        if len(s) == 2:
            k = 1.0 / np.sqrt(s[1])
        out.append(rs.uniform(-k, k, s).astype(np.float32))
    return out


def seeded_noise(shapes, seed, scale):
    rs = np.random.RandomState(seed)
    return [(rs.randn(*[int(x) for x in s]) * scale).astype(np.float32) for s in shapes]


def slim(x):
    """[sum, l2, strided subsample...] of a tensor as one float64 vector."""
    x = np.asarray(x, dtype=np.float64).ravel()
    return np.concatenate([[x.sum(), np.sqrt((x * x).sum())], x[::STRIDE]])


def check_slim(got, ref_slim, atol, frac_tol=None, frac=0.0, err_msg=""):
    """Compare a full tensor with its slim record: the subsample element-wise (at most `frac` of the entries may exceed `frac_tol`,
    none may exceed `atol`), sum and norm within the accumulated tolerance."""
    got = np.asarray(got, dtype=np.float64).ravel()
    sub = got[::STRIDE]
    d = np.abs(sub - ref_slim[2:])
    assert d.max() <= atol, (err_msg, "max", d.max(), atol)
    if frac_tol is not None:
        assert (d > frac_tol).mean() <= frac, (err_msg, "frac", (d > frac_tol).mean(), frac)
    n = got.size
    assert abs(got.sum() - ref_slim[0]) <= atol * n * 0.05 + 1e-6 * max(1.0, abs(ref_slim[0])), (err_msg, "sum", got.sum(), ref_slim[0])
    assert abs(np.sqrt((got * got).sum()) - ref_slim[1]) <= atol * np.sqrt(n) + 1e-6 * max(1.0, ref_slim[1]), (err_msg, "l2")
