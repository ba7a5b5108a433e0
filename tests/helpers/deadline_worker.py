"""TEST INFRASTRUCTURE (tests/test_host_threads.py): one of two gloo ranks on the CPU.  Rank 1 leaves before the collective; rank 0
arms learn.py's Deadline watchdog and enters an all_reduce that can never complete.  argv[1] = path of alphazero-risk_amd/learn.py
(only the watchdog class is taken from it: no engine binding, no GPU)."""
import os
import sys
import time

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
if rank == 1:
    os._exit(0)                       # the partner is gone before the collective
src = open(sys.argv[1]).read()
ns = {"os": os, "sys": sys}
exec(src[src.index("class Deadline:"):src.index("def learn(")], ns)
wd = ns["Deadline"](2.0, rank)
wd.arm("training and weight hand-over")
t = torch.ones(4)
try:
    dist.all_reduce(t)                # rank 1 never joins
except Exception:   # noqa: BLE001
    pass                              # (gloo may notice the closed socket: then wait like a stream synchronisation would)
time.sleep(60)
sys.exit(0)
