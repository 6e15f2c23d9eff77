"""CPU: the ENV kernels' sources (apex_amd/csrc/env.hip with cassie_lane.h, cassie_complete.h, estimator_lane.h: reset, env step, substep, one-launch rollout) compiled for
the host under the lane-exact wave emulation of tools/hipemu - 64 fibers per wave, every DPP operand / ds_bpermute / readlane / ballot an exchange between them, the
gfx950 inline-assembly dialect restated per lane in tools/hipemu/gfx950/lane_ops.h - and driven through the SAME host code (apex_amd/vecenv.py, the C ABI of env.hip) and
the SAME test bodies as the GPU parity tests of tests/test_gpu_env.py, against the fp64 oracle.  The redirection lives in this file only (a fixture swaps the loaded
library handle and the device hooks for the duration of a test); the product has no CPU path.

What this pins without a GPU: the lane map, every index and LDS offset, the tree / factor / sweep / finish stages as written, the contact detection, the complete-row
path, the estimator, reward, reset images, the in-kernel restart of the one-launch rollout.  What it cannot see: gfx950 code generation (hazard distances of the
hand-spaced DPP sequences, register allocation), v_rcp_f32 / fast-math rounding, speed."""
import contextlib
import os

import numpy as np
import pytest
import torch

from test_kernel_emulation_learner import CLANG, _NoStream, emulated_library

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def dev(monkeypatch):
    """torch.device('cpu') with apex_amd.vecenv / engine talking to the emulated kernel sources for this test only"""
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++")
    from apex_amd import _lib, engine, vecenv
    lib = emulated_library()
    lib.apx_emul_set_workgroups(0)
    monkeypatch.setattr(_lib, "_lib", lib)
    monkeypatch.setattr(engine, "_need_gpu", lambda *ts: None)
    monkeypatch.setattr(engine, "_stream", lambda: None)
    monkeypatch.setattr(vecenv, "_stream", lambda: None)
    monkeypatch.setattr(vecenv, "_device", lambda index: torch.device("cpu"))
    monkeypatch.setattr(vecenv, "_on_device", lambda t: True)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _NoStream())
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: _NoStream())
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: _NoStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return torch.device("cpu")


def test_emulated_smoke_reset_and_one_teacher_forced_step(dev):
    """the body of __graft_entry__.smoke()'s env half: reset observation against the oracle, then one teacher-forced env step (kernel state overwritten with the
    oracle's) at the parity suite's fixed tolerances"""
    import __graft_entry__ as G
    G._smoke_env(dev)
