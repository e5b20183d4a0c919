"""Container only (needs /root/reference): the observer oracle logs exactly what the reference's own RLGPUAlgoObserver method
source (utils/rlgames/rlgames_utils.py) logs on the same stream of infos."""
import types

import pytest
import torch

from observer_cases import Writer, infos_stream
from oracle import observer_oracle as oo
from oracle import ref_exec

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
F = "utils/rlgames/rlgames_utils.py"


def _reference(writer):
    cls = type("RefObserver", (), {m: ref_exec.load_function(F, m, cls="RLGPUAlgoObserver") for m in ("process_infos", "after_print_stats")})
    o = cls()
    o.algo = types.SimpleNamespace(device="cpu")
    o.writer, o.ep_infos, o.direct_info = writer, [], {}
    o.mean_scores = types.SimpleNamespace(current_size=0)
    return o


@pytest.mark.parametrize("steps,seed", [(1, 0), (40, 1), (333, 2)])
def test_observer_oracle_equals_reference_source(steps, seed):
    wr, wo = Writer(), Writer()
    ref, orc = _reference(wr), oo.ObserverOracle(wo)
    for epoch in range(3):
        for a, b in zip(infos_stream(steps, seed + epoch), infos_stream(steps, seed + epoch)):
            ref.process_infos(a, None)
            orc.process_infos(b, None)
        ref.after_print_stats(100 * epoch, epoch, 1.5 * epoch)
        orc.after_print_stats(100 * epoch, epoch, 1.5 * epoch)
    assert wr.calls == wo.calls and len(wr.calls) > 0
    assert any(t.startswith("Episode/") for t, _, _ in wr.calls)
