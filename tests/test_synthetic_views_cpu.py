"""CPU: the synthetic PhysX stand-in's setters (sim/synthetic.py) -- host logic only."""
import torch

from omniisaacgymenvs_amd.sim.synthetic import SyntheticArticulationView


def test_aliased_hand_over_is_skipped_only_for_identity_indices():
    """a full-size scatter whose values alias the view's buffer was dropped whatever the indices said."""
    v = SyntheticArticulationView("robots", 4, 2, device="cpu")
    v.apply_resets = True
    v.velocities[:] = torch.arange(24, dtype=torch.float32).reshape(4, 6)
    want = v.velocities.clone()
    v.set_velocities(v.velocities, indices=torch.arange(4))  # identity: nothing moves
    assert torch.equal(v.velocities, want)
    perm = torch.tensor([1, 0, 3, 2])
    v.set_velocities(v.velocities.clone(), indices=perm)  # a real permutation of a copy lands
    assert torch.equal(v.velocities, want[perm])
    cur = v.velocities.clone()
    v.set_velocities(v.velocities, indices=perm)  # the same through an alias: NOT skipped any more
    assert torch.equal(v.velocities[perm], cur)
