"""GPU: data-parallel optimiser step (azr_nn_train_dp) = the single-GPU step (azr_nn_train).

Two (and four) ranks over gloo share this box's one GPU; each takes its slice of every minibatch, batch statistics /
losses / gradients are all-reduced through the callback, every rank takes the same Adam step.  The reference trains on
GPU 0 only (alphazero_gpu_cluster.cpp:221-231); what is being replaced is its weight hand-over through temp.bin.
Tolerances are those of tests/test_gpu_train.py for the step itself: losses 2e-5, per-tensor gradients 2e-3 of the
tensor's largest gradient, weights after Adam 2e-6 absolute per step taken."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,blocks,bs", [(2, 2, 64), (4, 1, 64), (4, 2, 256)])
def test_data_parallel_step_equals_single_gpu_step(tmp_path, world, blocks, bs):
    """(4, 2, 256): 64-record shares — the per-rank shape of configs[4] (BATCH_SIZE 512 over 8 GPUs): the small-batch conv kernel and
    8-board weight-gradient slices in four PROCESSES over gloo (world 8 x 64 records in one process: tests/test_gpu_rehearsal.py)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp.npz")
    env = dict(os.environ, DP_BLOCKS=str(blocks), DP_BS=str(bs), DP_N=str(3 * bs + 8))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "helpers", "dp_train_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    d = np.load(out)
    L = 2 * blocks + 1
    per_step = 2 * (L + 1) + 2     # 2 all-reduces per batch-norm layer (L conv layers + the heads), the loss pair, the gradient
    for tag, steps in (("one", 1), ("multi", 6)):
        assert int(d[f"{tag}_steps"]) == steps and int(d["world"]) == world and int(d[f"{tag}_refused"]) == 1
        # every rank ends with the same weights, bit for bit (same reduced sums, same Adam step); same shuffle stream consumed
        for k in range(1, world):
            assert (d[f"{tag}_w_all"][k].view(np.uint32) == d[f"{tag}_w_all"][0].view(np.uint32)).all(), (tag, k)
        assert int(d[f"{tag}_state_dp"]) == int(d[f"{tag}_state_1"])
        calls = d[f"{tag}_calls"]
        assert len(calls) == steps * per_step, (len(calls), per_step)
        assert (calls[:, 0] == len(d["w0"])).sum() == steps      # one gradient all-reduce per step: the whole AZRW vector
    # ---- ONE step against the single-GPU step: same losses, same gradients tensor by tensor, same Adam update
    assert np.abs(d["one_hist_dp"] - d["one_hist_1"]).max() <= 2e-5, (d["one_hist_dp"], d["one_hist_1"])
    worst = 0.0
    for name, off, n in T.net_layout(blocks):
        a, b = d["one_g_dp"][off:off + n], d["one_g_1"][off:off + n]
        if name.endswith("_bn"):   # gamma, beta (moving statistics carry no gradient)
            a, b = a[:n // 2], b[:n // 2]
        scale = np.abs(b).max()
        if scale > 0:
            worst = max(worst, np.abs(a - b).max() / scale)
    assert worst <= 2e-3, worst
    # Adam's first step moves every weight by ~lr * sign(g): the two runs agree except where a gradient is so close to
    # zero that summation order decides its sign
    dw = np.abs(d["one_w_dp"] - d["one_w_1"])
    moved = np.abs(d["one_w_1"] - d["w0"]) > 0
    assert np.median(dw[moved]) <= 1e-7 and (dw[moved] > 1e-5).mean() <= 2e-3, (np.median(dw[moved]), (dw[moved] > 1e-5).mean())
    # ---- 2 epochs x 3 steps: the trajectories stay together (epoch losses within 1e-3 relative)
    rel = np.abs(d["multi_hist_dp"] - d["multi_hist_1"]) / np.abs(d["multi_hist_1"])
    assert rel.max() <= 1e-3, (d["multi_hist_dp"], d["multi_hist_1"])
    assert np.abs(d["multi_w_1"] - d["w0"]).max() > 1e-3
    print(f"world {world}: one step max rel grad diff {worst:.2e}, median |dw| {np.median(dw[moved]):.1e}, "
          f"share of weights off by > 1e-5: {(dw[moved] > 1e-5).mean():.1e}; multi-step loss rel diff {rel.max():.1e}; {per_step} all-reduces per step")
