"""GPU: tools/batch_size_perf.py writes the reference's log/batch-size-perf.txt (alphazero_gpu_cluster.cpp:54-65) in the format
python/src/log_chart.py:87-96 parses (csv rows "batch, ns_per_sample")."""
import csv
import importlib.util
import os

import pytest

from gpu_common import ROOT, pkg

pytestmark = pytest.mark.gpu


def test_batch_size_perf_log_format(tmp_path):
    spec = importlib.util.spec_from_file_location("batch_size_perf", os.path.join(ROOT, "tools", "batch_size_perf.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    P = pkg()
    rows = tool.measure(P, 2, P.NET_BF16, batches=[1, 4, 32, 256], warm=2, timed=5)
    path = str(tmp_path / "log" / "batch-size-perf.txt")
    tool.write_log(path, rows)
    # the reference's own reader (build_NN_batch_speed_chart), minus the plot
    batch_size, avg_time = [], []
    with open(path) as f:
        for row in csv.reader(f, delimiter=','):
            batch_size.append(int(row[0]))
            avg_time.append(int(row[1]))
    assert batch_size == [1, 4, 32, 256] and all(t > 0 for t in avg_time)
    assert avg_time[-1] < avg_time[0]          # a batch of 256 costs less per sample than a batch of 1
    assert set(tool.REFERENCE_NS) == set(tool.BATCHES)
