"""tests/third_party_cases.py on the build container's CPU (skips while the packages are absent;
the pytest summary says so)."""
import pytest

from tests import third_party_cases as T


@pytest.mark.parametrize("case", T.CASES, ids=[c.__name__[5:] for c in T.CASES])
def test_third_party(case):
    case()


def test_the_probe_itself_runs_against_the_restatements():
    """The cases are code that only executes where the packages exist: keep them honest here by
    running every comparison with the oracle's own functions bound to the packages' names."""
    import sys
    import types

    import torch

    from oracle import sngnn_oracle as O
    if any(m in sys.modules for m in ("torch_scatter", "torch_sparse", "torch_geometric")):
        pytest.skip("the real packages are installed: nothing to emulate")
    ts = types.ModuleType("torch_scatter")
    ts.scatter_max = lambda src, index, dim=0: O.scatter_max(src, index)
    ts.scatter = lambda msg, index, dim=-2, dim_size=None, reduce="mean": O.scatter_mean(msg, index, dim_size)
    ts.scatter_mean = lambda src, index, dim=0: O.scatter_mean_1d(src, index)
    sys.modules["torch_scatter"] = ts
    try:
        for case in (T.case_scatter_max_first_occurrence_and_empty_groups,
                     T.case_scatter_mean_counts_every_edge_and_clamps,
                     T.case_appendix_b_kat_through_the_real_packages,
                     T.case_exact_tie_fixtures_through_the_real_scatter_max):
            case()
    finally:
        del sys.modules["torch_scatter"]
    assert torch.equal(O.scatter_max(torch.tensor([1., 1.]), torch.tensor([0, 0]))[1], torch.tensor([0]))
