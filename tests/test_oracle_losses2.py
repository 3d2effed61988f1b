"""oracle/losses_ref.py ``lambda_loss`` / ``weighted_pointwise`` against values and autograd gradients of the REFERENCE
(tests/golden/losses2.npz from make_losses2_golden.py), incl. the reference's own __main__ demo answers 0.0127 / 0.0110."""
import json
import os

import numpy as np
import pytest

from oracle import losses_ref as L

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses2.npz"))
META = json.loads(str(G["meta"]))


@pytest.mark.parametrize("name", sorted(META, key=lambda s: int(s[4:])))
def test_lambda_loss_cases(name):
    kw = META[name]
    v, g = L.lambda_loss(G["y_pred"], G["y_true"], **kw)
    # the reference's discounts are float32 (torch.log2 of a float tensor) whatever the input dtype: schemes that use them agree to
    # a float32 ulp of log2 (numpy's and torch's float32 log2 differ in the last bit), the others to float64 round-off
    uses_d = kw["weighing_scheme"] in ("ndcgLoss1_scheme", "ndcgLoss2_scheme", "lambdaRank_scheme", "ndcgLoss2PP_scheme")
    rel = 5e-8 if uses_d else 1e-9
    assert v == pytest.approx(float(G[name + ".value"]), rel=rel, abs=1e-12), kw
    ref = G[name + ".grad"]
    assert np.allclose(g, ref, rtol=10 * rel, atol=(1e-7 * np.abs(ref).max()) if uses_d else 1e-12), kw


def test_reference_demo_answers():
    for name, t, printed in (("demo1", "demo.t1", 0.0127), ("demo2", "demo.t2", 0.0110)):
        v, g = L.lambda_loss(G["demo.y_pred"], G[t], weighing_scheme="ndcgLoss1_scheme", reduction_log="natural")
        assert round(v, 4) == printed                                   # what the reference's __main__ prints
        assert v == pytest.approx(float(G[name + ".value"]), rel=2e-5)    # the golden ran in fp32
        assert np.allclose(g, G[name + ".grad"], rtol=5e-3, atol=1e-7)


def test_weighted_pointwise():
    for i in (0, 1):
        v, g = L.weighted_pointwise(G[f"wp.demo{i}.pred"], G["wp.weight"])
        assert v == pytest.approx(float(G[f"wp.demo{i}.value"]), rel=1e-9) and np.allclose(g, G[f"wp.demo{i}.grad"], rtol=1e-8, atol=1e-14)
    v, g = L.weighted_pointwise(G["y_pred"], G["wp.rand.weight"], T=0.7)
    assert v == pytest.approx(float(G["wp.rand.value"]), rel=1e-9) and np.allclose(g, G["wp.rand.grad"], rtol=1e-8, atol=1e-14)
