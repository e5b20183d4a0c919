"""Container only (needs /root/reference): the oracle's restatement equals the reference's own
function source, executed verbatim under local torch, bit for bit on random + edge inputs."""
import pytest
import torch

from oracle import cartpole_oracle as co
from oracle import gen_golden
from oracle import locomotion_oracle as lo
from oracle import ref_exec

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
LOCO = "tasks/shared/locomotion.py"


@pytest.mark.parametrize("kind", ["ant", "humanoid"])
@pytest.mark.parametrize("n,seed", [(33, 3), (1000, 4)])
def test_locomotion_restatement_equals_reference_source(kind, n, seed):
    spec = lo.ant_spec() if kind == "ant" else lo.humanoid_spec()
    g = gen_golden.locomotion_golden(kind, n, seed)  # reference source -> expected
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    b = lo.LocomotionBuffers.fresh(spec, n)
    b.actions[:], b.potentials[:] = t["in_actions"], t["in_potentials"]
    b.progress_buf[:], b.reset_buf[:] = t["in_progress"], t["in_reset"]
    lo.post_physics(spec, b, t["in_pos"], t["in_rot"], t["in_vel"], t["in_dof_pos"], t["in_dof_vel"], t["in_sensors"])
    for mine, want in ((b.obs_buf, "out_obs"), (b.potentials, "out_potentials"), (b.prev_potentials, "out_prev_potentials"),
                       (b.up_vec, "out_up_vec"), (b.heading_vec, "out_heading_vec"), (b.rew_buf, "out_rew"),
                       (b.reset_buf, "out_reset"), (b.progress_buf, "out_progress")):
        assert torch.equal(mine, t[want]), want


def test_normalize_angle_and_wrap_to_pi_verbatim():
    f = ref_exec.load_function(LOCO, "normalize_angle")
    x = torch.linspace(-20, 20, 4001)
    assert torch.equal(f(x), lo.normalize_angle(x))


def test_reference_reset_idx_arithmetic_equals_restatement():
    """LocomotionTask.reset_idx (:133-158) executed on a stub self with the helpers restated;
    torch.rand replayed from the same seed."""
    spec = lo.ant_spec()
    n, d = 50, spec.num_dof
    g = torch.Generator().manual_seed(0)
    init_pos = torch.randn(n, 3, generator=g)
    init_rot = torch.randn(n, 4, generator=g)
    init_dof = torch.rand(n, d, generator=g) - 0.5
    ids = torch.tensor([0, 3, 4, 17, 49])
    captured = {}

    class _View:
        num_dof = d

        def set_joint_positions(self, v, indices=None): captured["dof_pos"] = v.clone()
        def set_joint_velocities(self, v, indices=None): captured["dof_vel"] = v.clone()
        def set_world_poses(self, p, r, indices=None): captured["root"] = (p.clone(), r.clone())
        def set_velocities(self, v, indices=None): captured["root_vel"] = v.clone()

    from oracle import torch_utils as tu

    def torch_rand_float(lower, upper, shape, device):
        return (upper - lower) * torch.rand(*shape, device=device) + lower

    b = lo.LocomotionBuffers.fresh(spec, n)
    stub = ref_exec.Stub(
        _robots=_View(), _device="cpu", initial_dof_pos=init_dof, initial_root_pos=init_pos, initial_root_rot=init_rot,
        dof_limits_lower=spec.dof_limits_lower, dof_limits_upper=spec.dof_limits_upper,
        targets=torch.tensor(spec.target).repeat((n, 1)), dt=spec.dt, potentials=b.potentials.clone(),
        prev_potentials=b.prev_potentials.clone(), reset_buf=b.reset_buf.clone(), progress_buf=b.progress_buf.clone() + 7)
    fn = ref_exec.load_function(LOCO, "reset_idx", cls="LocomotionTask", helpers=False,
                                extra={"torch_rand_float": torch_rand_float, "tensor_clamp": tu.tensor_clamp})
    torch.manual_seed(77)
    fn(stub, ids)
    torch.manual_seed(77)
    u_pos, u_vel = torch.rand(len(ids), d), torch.rand(len(ids), d)
    b.progress_buf += 7
    w = lo.reset_idx(spec, b, ids, init_pos, init_rot, init_dof, u_pos, u_vel)
    assert torch.equal(w.dof_pos, captured["dof_pos"]) and torch.equal(w.dof_vel, captured["dof_vel"])
    assert torch.equal(w.root_pos, captured["root"][0]) and torch.equal(w.root_rot, captured["root"][1])
    assert torch.equal(w.root_vel, captured["root_vel"])
    for k in ("potentials", "prev_potentials", "reset_buf", "progress_buf"):
        assert torch.equal(getattr(b, k), getattr(stub, k)), k


def test_cartpole_restatement_equals_reference_source():
    g = gen_golden.cartpole_golden(300, 9)
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    spec = co.CartpoleSpec()
    n = 300
    obs, rew = torch.zeros(n, 4), torch.zeros(n)
    reset, progress = torch.ones(n, dtype=torch.long), t["in_progress"].clone()
    co.post_physics(spec, t["in_dof_pos"], t["in_dof_vel"], obs, rew, reset, progress)
    assert torch.equal(rew, t["out_rew"]) and torch.equal(reset, t["out_reset"])
    q, qd = co.reset_idx(spec, t["reset_ids"], t["reset_u"], reset, progress)
    assert torch.equal(q, t["reset_dof_pos"]) and torch.equal(qd, t["reset_dof_vel"])
