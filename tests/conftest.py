import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _third_party_status():
    """Whether the packages the reference's hot path calls (requirements.txt:66-69) import here."""
    import importlib
    out = {}
    for name in ("torch_scatter", "torch_sparse", "torch_geometric"):
        try:
            m = importlib.import_module(name)
            out[name] = "present " + str(getattr(m, "__version__", "?"))
        except Exception as ex:      # noqa: BLE001
            out[name] = "absent (" + type(ex).__name__ + ")"
    return out


def pytest_terminal_summary(terminalreporter):
    """What the parity tests tolerated or could not check, counted and printed instead of hidden:
    the third-party packages' presence (tests/test_third_party.py runs against them when they
    import), the rows whose selection differed from the oracle's at operator level (same inputs:
    gate TIE_ULPS) and at model level (each side's own ``lin`` result: MODEL_TIE_ULPS), with the
    largest gap any of them needed."""
    from tests import helpers
    tr = terminalreporter
    tr.write_line("third-party packages: " + ", ".join(f"{k} {v}" for k, v in _third_party_status().items()))
    for line in helpers.REPORT_LINES:
        tr.write_line(line)
    if helpers.OPERATOR_TIE_LOG:
        tot = sum(d for _, d, _ in helpers.OPERATOR_TIE_LOG)
        rows = sum(r for _, _, r in helpers.OPERATOR_TIE_LOG)
        mx = max(helpers.NEAR_TIE_GAPS, default=0.0)
        tr.write_line(f"near-tie selection rows tolerated at operator level: {tot} of {rows} rows compared; "
                      f"largest gap needed {mx:.3e} = {mx / helpers.ULP32:.2f} ulp (gate {helpers.TIE_ULPS} ulp)")
        for label, d, r in helpers.OPERATOR_TIE_LOG:
            if d:
                tr.write_line(f"  {label}: {d} of {r}")
    if helpers.DEEP_OPERATOR_LOG:
        tot = sum(d for _, d, _ in helpers.DEEP_OPERATOR_LOG)
        rows = sum(r for _, _, r in helpers.DEEP_OPERATOR_LOG)
        mx = max(helpers.DEEP_OPERATOR_GAPS, default=0.0)
        mx64 = max(helpers.DEEP_EXACT_GAPS, default=0.0)
        tr.write_line(f"second-layer operator level (the oracle's own layer input on both sides): {tot} of {rows} rows, "
                      f"largest gap in EXACT (float64) cosines of the same unit rows {mx64:.3e} = {mx64 / helpers.ULP32:.2f} ulp "
                      f"(gate {helpers.TIE_ULPS} ulp: the kernel's own rounding); in the oracle's fp32 scores "
                      f"{mx:.3e} = {mx / helpers.ULP32:.2f} ulp (gate {helpers.DEEP_GATE_ULPS} ulp: + the oracle's)")
        for label, d, r in helpers.DEEP_OPERATOR_LOG:
            tr.write_line(f"  {label}: {d} of {r}")
    if helpers.NEAR_TIE_LOG:
        tot = sum(d for _, d, _ in helpers.NEAR_TIE_LOG)
        rows = sum(r for _, _, r in helpers.NEAR_TIE_LOG)
        mx = max(helpers.MODEL_TIE_GAPS, default=0.0)
        tr.write_line(f"near-tie selection rows tolerated at model level: {tot} of {rows} rows compared; "
                      f"largest gap needed {mx:.3e} = {mx / helpers.ULP32:.2f} ulp (gate {helpers.MODEL_TIE_ULPS} ulp at a model's first layer, + 1 per layer in front)")
        for label, d, r in helpers.NEAR_TIE_LOG:
            tr.write_line(f"  {label}: {d} of {r}")
