"""The oracle's loss restatements against golden vectors captured from the reference
(tests/golden/losses.npz, made by tests/golden/make_golden.py) and the reference's own printed
known answers."""
import os

import numpy as np
import pytest

from oracle import losses_ref as L

from conftest import GOLDEN

G = np.load(os.path.join(GOLDEN, "losses.npz"))
NAMES = [str(n) for n in G["names"]]


def run_oracle(name):
    kind = str(G[name + "/kind"])
    yp, yt = G[name + "/y_pred"], G[name + "/y_true"]
    kw = {k.split("/kw_")[1]: G[k] for k in G.files if k.startswith(name + "/kw_")}
    if kind == "kl":
        return L.kl_div(yp, yt, T=float(kw["T"]))
    if kind == "mse":
        return L.margin_mse(yp, yt)
    red = str(kw["reduction"]) if "reduction" in kw else "mean"
    if kind == "ranknet":
        return L.ranknet(yp, yt, reduction=red)
    if kind == "lambda":
        return L.lambda_mrr(yp, yt, reduction=red)
    return L.bweight_lambda_mrr(yp, yt, kw["batch_weight"], reduction=red)


@pytest.mark.parametrize("name", NAMES)
def test_loss_value_and_grad_match_reference(name):
    val, grad = run_oracle(name)
    ref_val, ref_grad = float(G[name + "/value"]), G[name + "/grad"]
    # reference computes in fp32; oracle in fp64
    assert val == pytest.approx(ref_val, rel=2e-5, abs=1e-6), name
    scale = max(1e-12, float(np.abs(ref_grad).max()))
    assert np.allclose(grad, ref_grad, rtol=2e-4, atol=2e-5 * scale + 1e-9), name


def test_reference_known_answers():
    """Values printed by the reference's __main__ demos (kl_div.py:27-31 -> 0.0209, margin_mse.py:22-26 ->
    0.1111, ranknet.py:48-67 -> 0.7022, lambda_rank.py:107-114 -> 0.5343 / 0.1537)."""
    assert run_oracle("demo_kl")[0] == pytest.approx(0.02089791, abs=1e-7)
    assert run_oracle("demo_mse")[0] == pytest.approx(0.11111111, abs=1e-7)
    assert run_oracle("demo_ranknet")[0] == pytest.approx(0.7022, abs=5e-5)
    assert run_oracle("demo_bweight3")[0] == pytest.approx(0.5343, abs=5e-5)
    assert run_oracle("demo_bweight4")[0] == pytest.approx(0.1537, abs=5e-5)
    kl_g = run_oracle("demo_kl")[1].ravel()
    assert np.allclose(kl_g, [0.0348183, 0.0128089, -0.0476272, 0.0200345, -0.0321860, 0.0121515], atol=1e-6)
    assert np.allclose(run_oracle("demo_mse")[1].ravel(), np.array([1, 1, -2, 1, -2, 1]) / 9.0, atol=1e-7)


def test_padded_entries_get_zero_grad():
    name = "lambda_pad_4x8"
    _, grad = run_oracle(name)
    pad = G[name + "/y_true"] == -1
    assert np.all(grad[pad] == 0)
    assert np.all(G[name + "/grad"][pad] == 0)      # and so does the reference


def test_invalid_reduction_raises():
    with pytest.raises(ValueError):
        L.lambda_mrr(np.zeros((1, 3)), np.array([[1.0, 0.5, 0.0]]), reduction="max")


def test_ranknet_is_permutation_invariant():
    name = "ranknet_4x8"
    yp, yt = G[name + "/y_pred"], G[name + "/y_true"]
    perm = np.array([3, 1, 7, 0, 2, 6, 5, 4])
    assert L.ranknet(yp[:, perm], yt[:, perm])[0] == pytest.approx(L.ranknet(yp, yt)[0], rel=1e-12)


def test_train_mrr_recall():
    logits = np.array([[0.1, 0.9, 0.3], [0.5, 0.2, 0.1]])
    labels = np.array([[1.0, 0.5, -0.5], [1.0, 0.5, -0.5]])
    mrr, rec = L.train_mrr_recall(logits, labels, 10)
    assert mrr == pytest.approx((1 / 3 + 1.0) / 2)
    assert rec == 1.0
