"""Container only (needs /root/reference): the DR-noise oracle equals the reference's own Randomizer methods
(utils/domain_randomization/randomize.py:283-410), compiled from their source and run on a stub ``self`` under CPU
torch with torch's RNG replayed from the same seed -- bit for bit, for every distribution / operation pair."""
import copy

import pytest
import torch

from oracle import dr_noise_oracle as dro
from oracle import ref_exec

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
F = "utils/domain_randomization/randomize.py"
METHODS = ["apply_observations_randomization", "apply_actions_randomization", "_apply_uncorrelated_noise",
           "_apply_correlated_noise"]

SHADOW_HAND_OBS = {  # cfg/task/ShadowHand.yaml:139-148
    "on_reset": {"operation": "additive", "distribution": "gaussian", "distribution_parameters": [0, 0.0001]},
    "on_interval": {"frequency_interval": 1, "operation": "additive", "distribution": "gaussian",
                    "distribution_parameters": [0, 0.002]},
}
CASES = {
    "shadow_hand_obs": SHADOW_HAND_OBS,
    "uniform_scaling_every_3": {
        "on_reset": {"operation": "scaling", "distribution": "uniform", "distribution_parameters": [0.9, 1.1]},
        "on_interval": {"frequency_interval": 3, "operation": "additive", "distribution": "uniform",
                        "distribution_parameters": [-0.05, 0.05]},
    },
    "loguniform_reset_only": {
        "on_reset": {"operation": "scaling", "distribution": "loguniform", "distribution_parameters": [0.5, 2.0]},
    },
    "interval_only": {
        "on_interval": {"frequency_interval": 2, "operation": "scaling", "distribution": "log_uniform",
                        "distribution_parameters": [0.8, 1.25]},
    },
}


def _stub(params, n, w, kind):
    cls = type("RefRandomizer", (), {m: ref_exec.load_function(F, m, cls="Randomizer") for m in METHODS})
    o = cls()
    o._config = {"rl_device": "cpu"}
    setattr(o, f"_{kind}_dr_params", copy.deepcopy(params))
    setattr(o, f"_{kind}_counter_buffer", torch.zeros(n, dtype=torch.int))
    setattr(o, f"_{kind}_correlated_noise", torch.zeros(n, w))
    return o


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("kind", ["observations", "actions"])
def test_oracle_equals_reference_randomizer_source(case, kind):
    params = CASES[case]
    n, w = 203, 42 if kind == "observations" else 20
    ref = _stub(params, n, w, kind)
    counter, corr = torch.zeros(n, dtype=torch.int), torch.zeros(n, w)
    g = torch.Generator().manual_seed(5)
    torch.manual_seed(11)
    state = torch.get_rng_state()
    for step in range(7):
        buf = torch.randn(n, w, generator=g)
        reset = (torch.rand(n, generator=g) < (1.0 if step == 0 else 0.0 if step == 3 else 0.2)).long()
        torch.set_rng_state(state)
        fn = getattr(ref, f"apply_{kind}_randomization")
        want = fn(buf.clone(), reset) if kind == "observations" else fn(buf.clone(), reset)
        after = torch.get_rng_state()
        torch.set_rng_state(state)
        got = dro.apply_randomization(buf.clone(), reset, counter, corr, params)
        assert torch.equal(torch.get_rng_state(), after), "RNG consumption differs"
        state = after
        assert torch.equal(got, want), (case, kind, step)
        assert torch.equal(counter, getattr(ref, f"_{kind}_counter_buffer"))
        assert torch.equal(corr, getattr(ref, f"_{kind}_correlated_noise"))
