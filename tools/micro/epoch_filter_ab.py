"""A/B of the fp16 filter inside a graphed epoch: tools/bench_epochs.py <model> with sngnn_filter_enable(MODE)
(MODE from the environment, default 1)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sngnn_amd import _lib  # noqa: E402

_lib.load().sngnn_filter_enable(int(os.environ.get("MODE", "1")))
sys.argv = ["bench_epochs.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_epochs.py"), run_name="__main__")
