"""GPU: reproducibility_metric on device tensors against the reference's recorded outputs (row f-1, metrics half)."""
import pytest

from test_metrics import check_against_reference

pytestmark = pytest.mark.gpu


def test_reproducibility_metric_on_device_tensors_matches_the_reference_outputs():
    import torch

    assert torch.cuda.is_available(), "GPU test needs a GPU"
    check_against_reference("cuda")
