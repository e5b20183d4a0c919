"""The oracles of the SURVEY.md 8f rank-3 / rank-4 tasks reproduce the committed golden vectors (tests/golden/task_*.npz, written
by oracle/gen_golden_tasks.py from the reference's own method source) bit for bit -- runs without /root/reference."""
import os

import pytest
import torch

import golden_tasks as gt


@pytest.mark.parametrize("name", list(gt.FAMILIES))
def test_oracle_reproduces_task_golden(name):
    fam = gt.FAMILIES[name]
    paths = gt.files(name)
    assert len(paths) == 2, paths
    for path in paths:
        g = gt.load(path)
        st, bf, want = gt.unflatten(g, "st"), gt.unflatten(g, "b"), gt.unflatten(g, "out")
        b = fam.oracle_post(st, bf)
        got = fam.outputs(b)
        assert set(want) == set(got)
        for k, w in want.items():
            if isinstance(w, dict):
                for kk in w:
                    assert torch.equal(got[k][kk], w[kk]), (os.path.basename(path), k, kk)
            else:
                assert torch.equal(got[k], w) and got[k].dtype == w.dtype, (os.path.basename(path), k)
        assert 0 < int(want["reset_buf"].sum()) < want["reset_buf"].numel()
