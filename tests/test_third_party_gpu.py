"""tests/third_party_cases.py on the GPU box (``-m gpu``): the other machine where torch-scatter /
torch-sparse / torch-geometric could exist.  Skips while they are absent; the pytest summary prints
"third-party packages: ... absent" either way (tests/conftest.py)."""
import pytest

from tests import third_party_cases as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", T.CASES, ids=[c.__name__[5:] for c in T.CASES])
def test_third_party_on_the_gpu_box(case):
    case()
