"""CPU: the oracle reproduces every golden vector (tests/golden, produced by executing the
reference's function source).  Integer outputs bit-exact; fp32 outputs bit-exact on the CPU
that generated them and within the north_star tolerance elsewhere."""
import os

import pytest
import torch

import golden_io
from helpers import assert_close, assert_equal
from oracle import cartpole_oracle as co
from oracle import locomotion_oracle as lo


@pytest.mark.parametrize("kind", ["ant", "humanoid"])
def test_locomotion_oracle_reproduces_golden(kind):
    paths = golden_io.files(kind)
    assert len(paths) >= 2
    for path in paths:
        g = golden_io.load(path)
        spec = golden_io.spec_for(kind, g)
        n = g["in_pos"].shape[0]
        b = lo.LocomotionBuffers.fresh(spec, n)
        b.actions[:] = g["in_actions"]
        b.potentials[:] = g["in_potentials"]
        b.progress_buf[:] = g["in_progress"]
        b.reset_buf[:] = g["in_reset"]
        lo.post_physics(spec, b, g["in_pos"], g["in_rot"], g["in_vel"], g["in_dof_pos"], g["in_dof_vel"], g["in_sensors"])
        tag = os.path.basename(path)
        assert_equal(b.reset_buf, g["out_reset"], f"{tag} reset")
        assert_equal(b.progress_buf, g["out_progress"], f"{tag} progress")
        assert_equal(b.prev_potentials, g["out_prev_potentials"], f"{tag} prev_potentials")
        assert_close(b.obs_buf, g["out_obs"], f"{tag} obs")
        assert_close(b.potentials, g["out_potentials"], f"{tag} potentials", rtol=1e-7, atol=0)
        assert_close(b.up_vec, g["out_up_vec"], f"{tag} up_vec")
        assert_close(b.heading_vec, g["out_heading_vec"], f"{tag} heading_vec")
        pot_ulp_ok = b.potentials == g["out_potentials"]  # reward inherits 1 ULP(60000) = 4e-3 otherwise
        assert_close(b.rew_buf, g["out_rew"], f"{tag} rew", mask=pot_ulp_ok)
        assert_equal(lo.dof_at_limit_cost(spec, g["out_obs"]), g["out_limit_cost"], f"{tag} limit cost")


def test_golden_holds_the_edge_cases():
    g = golden_io.load(golden_io.files("ant")[1])
    th = 0.31
    z = g["in_pos"][:3, 2]
    assert z[0].item() == pytest.approx(th) and z[1] < z[0] < z[2]          # termination height +-1 ULP
    assert g["out_reset"][1] == 1 and g["out_rew"][1] == -2.0                 # just below -> death cost
    assert list(g["out_reset"][15:19]) == [0, 1, 1, 1]                        # progress 998 (no), 999, 1000, 1001
    assert torch.isfinite(g["out_obs"]).all()
    assert g["out_obs"][12, 11] == 0.0                                        # to_target == 0 -> heading_proj 0
    assert g["out_limit_cost"].dtype == torch.int64 and g["out_limit_cost"][24] == 8  # all eight dofs at +1.0


def test_cartpole_oracle_reproduces_golden():
    paths = golden_io.files("cartpole")
    assert len(paths) >= 2
    for path in paths:
        g = golden_io.load(path)
        spec = co.CartpoleSpec()
        n = g["in_dof_pos"].shape[0]
        obs, rew = torch.zeros(n, 4), torch.zeros(n)
        reset, progress = torch.ones(n, dtype=torch.long), g["in_progress"].clone()
        co.post_physics(spec, g["in_dof_pos"], g["in_dof_vel"], obs, rew, reset, progress)
        assert_equal(obs, g["out_obs"], "obs")
        assert_equal(rew, g["out_rew"], "rew")
        assert_equal(reset, g["out_reset"], "reset")
        assert_equal(progress, g["out_progress"], "progress")
        q, qd = co.reset_idx(spec, g["reset_ids"], g["reset_u"], reset, progress)
        assert_equal(q, g["reset_dof_pos"], "reset dof_pos")
        assert_equal(qd, g["reset_dof_vel"], "reset dof_vel")
        assert_equal(reset, g["reset_reset_buf"], "reset_buf after reset")
        assert_equal(progress, g["reset_progress_buf"], "progress_buf after reset")
        # on-threshold rows: |x| == 3.0 and |theta| == pi/2 are NOT resets (strict >); progress 500 is
        assert list(g["out_reset"][:4]) == [0, 1, 0, 1]
        assert list(g["out_reset"][4:8].tolist()) == [max(int(r), 0) for r in g["out_reset"][4:8]]
        assert g["out_reset"][6] == 1 and g["out_reset"][7] == 1  # progress 499+1 = 500 and 501 -> done
