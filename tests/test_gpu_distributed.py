"""Two ranks (gloo for the exchange, each with its own context on cuda:0) run the time-sharded stream and
the split track batch through the real C ABI and compare with the unsharded result and the oracle."""
import pytest

from test_distributed_cpu import run_workers


@pytest.mark.gpu
def test_two_ranks_one_gpu():
    run_workers("gpu", 2)
