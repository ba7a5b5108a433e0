"""TEST INFRASTRUCTURE: one rank of the data-parallel optimiser-step test (tests/test_gpu_train_dp.py), started as
`python -m torch.distributed.run --nproc-per-node W tests/helpers/dp_train_worker.py OUT.npz` with the gloo backend
(the ranks share this box's one GPU; on an 8-GPU node the same code runs over RCCL).

Every rank: same weights, same records, same shuffle stream; azr_nn_train_dp with its slice of every minibatch.  Rank 0
additionally runs the single-GPU azr_nn_train on the same inputs in a second engine.  Saved for the test: both weight
vectors, both gradient vectors of the last step, both loss histories, and every rank's weights (they must be equal)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out = sys.argv[1]
    blocks, bs, n, epochs = int(os.environ.get("DP_BLOCKS", "2")), int(os.environ.get("DP_BS", "64")), int(os.environ.get("DP_N", "200")), 2
    torch.cuda.set_device(0)
    torch.cuda.init()
    backend = os.environ.get("DP_BACKEND", "gloo")   # "nccl" = RCCL: the all-reduces run on the engine's device buffers themselves
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import azr_testlib as T
    P = importlib.import_module("alphazero-risk_amd")
    shard = importlib.import_module("alphazero-risk_amd.shard")
    flat = T.make_net_flat(blocks, seed=9, perturb_bn=True)
    rng = np.random.default_rng(5)
    in88 = np.load(os.path.join(T.GOLDEN, "encode.npz"))["in88"]
    rec = np.zeros((n, 265), np.uint8)
    rec[:, 0] = rng.integers(0, 2, n)
    rec[:, 1:89] = in88[rng.integers(0, len(in88), n)]
    rec[:, 89:93] = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), n).view(np.uint8).reshape(n, 4)
    pi = rng.random((n, 43)).astype(np.float32) ** 3
    pi /= pi.sum(1, keepdims=True)
    rec[:, 93:265] = pi.view(np.uint8).reshape(n, 172)

    ar = shard.make_allreduce(dist, on_device=(backend == "nccl"), device_index=0)
    native = os.environ.get("DP_NATIVE", "0") == "1"   # the engine's own RCCL communicator: every sum in stream order, no callback
    res = {}
    for tag, nn, ep in (("one", bs, 1), ("multi", n, epochs)):   # one single step (tight comparison), then 2 epochs x 3 steps
        eng = P.Engine(4, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
        eng.set_weights(flat)
        calls = []

        def counted(ptr, count, dtype):
            calls.append((count, dtype))
            ar(ptr, count, dtype)

        if native:
            shard.native_dp_init(eng, dist, rank, world, torch.device("cuda", 0))
        hist, state = eng.train_dp(rec[:nn], ep, None if native else counted, rank, world, batch_size=bs, rng_state=4321)
        w_dp, g_dp = eng.get_weights(), eng.train_grads()
        bad = 0
        try:   # a minibatch that does not split evenly over the ranks is refused
            if world > 1:
                eng.train_dp(rec[:nn], 1, counted, rank, world, rng_state=1, batch_size=bs + 1)
        except P.AzrError as e:
            bad += e.code == 1
        eng.close()
        cdev = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
        allw = [torch.zeros(len(w_dp), device=cdev) for _ in range(world)]
        dist.all_gather(allw, torch.from_numpy(w_dp).to(cdev))
        allw = [x.cpu() for x in allw]
        if rank == 0:
            ref = P.Engine(4, blocks=blocks, sims=1, dtype=P.NET_F32, node_capacity=64)
            ref.set_weights(flat)
            hist1, state1 = ref.train(rec[:nn], ep, batch_size=bs, rng_state=4321)
            w_1, g_1 = ref.get_weights(), ref.train_grads()
            ref.close()
            res.update({f"{tag}_w_dp": w_dp, f"{tag}_g_dp": g_dp, f"{tag}_w_1": w_1, f"{tag}_g_1": g_1, f"{tag}_hist_dp": np.array(hist),
                        f"{tag}_hist_1": np.array(hist1), f"{tag}_state_dp": state, f"{tag}_state_1": state1,
                        f"{tag}_w_all": np.stack([x.numpy() for x in allw]), f"{tag}_calls": np.array(calls, np.int64), f"{tag}_refused": bad,
                        f"{tag}_steps": ep * (nn // bs)})
    if rank == 0:
        np.savez(out, w0=flat, world=world, backend=backend, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
