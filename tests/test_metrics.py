"""reproducibility_metric against the reference's own outputs (tests/golden/metrics_ref.npz, recorded by
tests/golden/gen_golden.py from /root/reference/src/pcgym/evaluation_metrics.py:81-327): every (dispersion, performance)
pair, every method, every component, on shapes that include the ones where the reference's MAD raises or broadcasts
(:127-130, quirk Q14).  CPU here; the same check on device tensors in tests/test_gpu_metrics.py."""
import numpy as np
import pytest
import torch

import helpers as H
from pcgym_amd.rollout import reproducibility_metric


def metric_cases():
    g = H.gold("metrics_ref")
    names = sorted({k.split("__")[0] for k in g.files})
    return g, names


def check_against_reference(device):
    g, names = metric_cases()
    n_val = n_raise = 0
    for name in names:
        comps = [k.split("__")[2] for k in g.files if k.startswith(name + "__in__")]
        data = {"pi": {c: torch.tensor(g[f"{name}__in__{c}"], device=device) for c in comps}}
        for disp in ("std", "mad"):
            for perf in ("mean", "median"):
                m = reproducibility_metric(disp, perf, -1.25)
                for kind, fn in (("perf", m.policy_performance_metric), ("disp", m.policy_dispersion_metric),
                                 ("scal", m.scalarised_performance)):
                    for c in comps:
                        key = f"{name}__{disp}__{perf}__{kind}__{c}"
                        if key + "_raises" in g.files:
                            with pytest.raises(ValueError):
                                fn(data, c)
                            n_raise += 1
                            continue
                        got = fn(data, c)["pi"][c]
                        assert got.device.type == torch.device(device).type
                        want = g[key]
                        assert tuple(got.shape) == want.shape, (key, tuple(got.shape), want.shape)
                        assert np.allclose(got.cpu().numpy(), want, rtol=1e-13, atol=1e-13), key
                        n_val += 1
    assert n_val >= 150 and n_raise >= 20, (n_val, n_raise)


def test_reproducibility_metric_matches_the_reference_outputs():
    check_against_reference("cpu")


def test_all_components_at_once_and_evaluate():
    g, _ = metric_cases()
    comps = ["r", "x", "u", "g"]
    data = {"pi": {c: torch.tensor(g[f"n7_reps5__in__{c}"]) for c in comps}}
    m = reproducibility_metric("std", "median", -1.25)
    out = m.scalarised_performance(data)
    assert list(out["pi"].keys()) == comps
    for c in comps:
        assert np.allclose(out["pi"][c].numpy(), g[f"n7_reps5__std__median__scal__{c}"], rtol=1e-13, atol=1e-13)

    class Evaluator:  # evaluation_metrics.py:232-237: .data if present, else get_rollouts()
        def get_rollouts(self):
            return data

    ev = m.evaluate(Evaluator(), "r")
    assert np.allclose(ev["pi"]["r"].numpy(), g["n7_reps5__std__median__scal__r"], rtol=1e-13, atol=1e-13)


def test_mad_about_each_rows_own_median_when_compat_is_off():
    rng = np.random.default_rng(0)
    for shape in ((1, 7, 5), (3, 7, 5), (2, 6, 6), (9,)):
        x = rng.normal(size=shape)
        want = np.median(np.abs(x - np.median(x, axis=-1, keepdims=True)), axis=-1)
        got = reproducibility_metric("mad", "mean", 1.0, reference_compat=False).policy_dispersion_metric(
            {"pi": {"x": torch.tensor(x)}}, "x")["pi"]["x"].numpy()
        assert np.allclose(got, want, rtol=1e-13, atol=1e-13)


def test_bad_names_raise_like_reference():
    with pytest.raises(ValueError, match="Invalid dispersion metric"):
        reproducibility_metric("iqr", "mean", 1.0)
    with pytest.raises(ValueError, match="Invalid performance metric"):
        reproducibility_metric("std", "mode", 1.0)
