"""GPU: activation recomputation of the trunk on the kernels (float32 and bf16 autocast): bit-equal to the plain run --
the kernels are deterministic, the re-run forward leaves running statistics and counters alone (tests/test_recompute_cpu.py)."""
import pytest

from tests.test_recompute_cpu import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('autocast', [False, True])
def test_recomputed_trunk_equals_the_plain_one_on_the_kernels(autocast):
    check('cuda', autocast)
