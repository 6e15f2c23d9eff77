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


# ---- G4b (tools/refprobe/gen_golden_epoch.py): an epoch of small-minibatch PPO steps on the 2 x 256 networks; (mirror, minibatch, steps, Adam step count of the first)
EPOCH_CASES = [(True, 64, 8, 1), (False, 64, 6, 1), (True, 32, 4, 7), (True, 128, 3, 1), (False, 16, 5, 2049)]
_ACTOR_SHAPES = [(256, 50), (256,), (256, 256), (256,), (10, 256), (10,)]
_CRITIC_SHAPES = [(256, 50), (256,), (256, 256), (256,), (1, 256), (1,)]


def epoch_case_inputs(c):
    """Inputs of G4b case c, regenerated from its seed: parameter lists (state_dict order) of the policy, the old policy (policy - noise) and the critic,
    the normaliser, a batch of 1.25 x the epoch's rows (so that the sample order is a proper gather) and the epoch's sample order."""
    mirror, mb, nb, adam_t0 = EPOCH_CASES[c]
    rs = np.random.RandomState(4100 + c)
    B = (nb * mb * 5) // 4
    actor = seeded_params(_ACTOR_SHAPES, 4200 + c)
    critic = seeded_params(_CRITIC_SHAPES, 4300 + c)
    actor[4] = actor[4] * 0.5; critic[4] = critic[4] * 0.5
    old = [w - nz for w, nz in zip(actor, seeded_noise(_ACTOR_SHAPES, 4400 + c, 0.004))]
    obs = rs.randn(B, 50).astype(np.float32)
    ph = rs.rand(B) * 2 * np.pi
    obs[:, 46] = np.sin(ph); obs[:, 47] = np.cos(ph)      # clock columns are valid sines (arcsin in mirror_clock_observation)
    return dict(actor=actor, old=old, critic=critic, obs=obs, act=(rs.randn(B, 10) * 0.3).astype(np.float32),
                ret=rs.randn(B).astype(np.float32), adv=rs.randn(B).astype(np.float32),
                obs_mean=(rs.randn(50) * 0.3).astype(np.float32), obs_std=(rs.rand(50) + 0.5).astype(np.float32),
                perm=rs.permutation(B)[:nb * mb].astype(np.int64))
