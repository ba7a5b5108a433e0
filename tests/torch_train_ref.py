"""PyTorch cross-check of the optimiser step.

The product's optimiser step is `azr_nn_train` (csrc/azr_train.hip: hand-written HIP forward-train / backward / Adam on
the AZRW vector).  This module is the same graph — python/src/build_graph.py:54-103, driven by AlphaZeroNN::train,
neural_network/alphazero_nn.cpp:351-410 — in PyTorch autograd over the same AZRW flat parameter vector.  It is what the
native step is tested against (tests/test_gpu_train.py, float64 on the CPU), the timing yard-stick of
tools/train_bench.py (PyTorch-ROCm / MIOpen).the float64 reference of tests/test_gpu_train.py.  Inference, search and self-play never
touch it.  Parity: "unpinned" (no TensorFlow here); its forward pass is checked against the oracle's fp32 restatement
and its update rule against a NumPy Adam in tests/test_train.py.

Semantics kept from the reference graph:
  loss = softmax-CE(target pi, logits) [batch mean] + MSE(z, v) [batch mean] + 1e-3 * sum ||kernel||^2
         over the 2B+3 conv kernels and the 3 dense kernels (build_graph.py:30,60,92-98)
  Adam(lr 1e-3, beta1 .9, beta2 .999, eps 1e-8) (build_graph.py:31,103; torch.optim.Adam places eps slightly
  differently from tf.train.AdamOptimizer, which the native step follows); BN momentum 0.99, eps 1e-3, batch statistics
  in training; the stem's conv_bn normalises over the board ROW (axis=1, build_graph.py:68)
  per epoch: shuffle, floor(N / BATCH_SIZE) minibatches, epoch-average policy / value loss (alphazero_nn.cpp:366-408)
"""
import numpy as np
import torch
import torch.nn.functional as F

F_ = 256
BN_EPS = 1e-3
BN_MOMENTUM_TORCH = 0.01  # TF momentum 0.99
L2_C = 1e-3


def layout(blocks):
    """(name, offset, shape) of every tensor in the AZRW flat vector"""
    off, out = 0, []

    def add(name, *shape):
        nonlocal off
        n = int(np.prod(shape))
        out.append((name, off, shape))
        off += n

    add("stem_w", 3, 3, 13, F_)
    add("stem_bn", 4, 7)
    for b in range(blocks):
        for ab in "ab":
            add(f"b{b}{ab}_w", 3, 3, F_, F_)
            add(f"b{b}{ab}_bn", 4, F_)
    add("pi_w", F_, 2); add("pi_bn", 4, 2); add("pd_w", 84, 43); add("pd_b", 43)
    add("v_w", F_, 1); add("v_bn", 4, 1); add("v1_w", 42, 256); add("v1_b", 256); add("v2_w", 256, 1); add("v2_b", 1)
    return out, off


def planes_from_in88(in88):
    """setInStateTensor (alphazero_nn.cpp:31-67), vectorised: uint8 [n,88] -> float32 [n,13,7,6] (NCHW)"""
    x = np.asarray(in88, np.uint8)
    n = x.shape[0]
    army = (x[:, :42] & 63).astype(np.float32) / 32.0
    owner = x[:, :42] >> 6
    cur = x[:, 42:43]
    f = x[:, 48:88].copy().view(np.float32).reshape(n, 10)
    p = np.zeros((n, 13, 42), np.float32)
    p[:, 0] = np.where(owner == cur, army, 0)
    p[:, 1] = np.where(owner == 1 - cur, army, 0)
    p[:, 2] = np.where(owner == 2, army, 0)
    p[:, 3] = f[:, 9:10]   # army share
    p[:, 4] = f[:, 0:1]    # reinforcement share
    p[:, 5] = f[:, 1:2]    # attacks during turn
    p[:, 6] = f[:, 2:3]    # can draw card
    for k in range(6):
        p[:, 7 + k] = f[:, 3 + k:4 + k]
    return p.reshape(n, 13, 7, 6)


def unpack_records(rec265):
    """265-byte records -> (in88 [n,88] u8, pi [n,43] f32, z [n] f32)"""
    r = np.asarray(rec265, np.uint8).reshape(-1, 265)
    return r[:, 1:89].copy(), r[:, 93:265].copy().view(np.float32).reshape(-1, 43), r[:, 89:93].copy().view(np.float32).reshape(-1)


class AzrNet(torch.nn.Module):
    """the graph of build_graph.py:54-90 with parameters bound to the AZRW layout"""

    def __init__(self, blocks, flat):
        super().__init__()
        self.blocks = blocks
        self.lay, self.count = layout(blocks)
        flat = np.asarray(flat, np.float32)
        assert flat.size == self.count
        self.p = torch.nn.ParameterDict()
        self.buf = {}
        for name, off, shape in self.lay:
            t = torch.from_numpy(flat[off:off + int(np.prod(shape))].reshape(shape).copy())
            if name.endswith("_bn"):
                key = name[:-3]
                self.p[key + "_g"] = torch.nn.Parameter(t[0].clone())
                self.p[key + "_b"] = torch.nn.Parameter(t[1].clone())
                self.register_buffer(key + "_m", t[2].clone())
                self.register_buffer(key + "_v", t[3].clone())
            else:
                self.p[name] = torch.nn.Parameter(t)

    def kernels(self):
        return [v for k, v in self.p.items() if k.endswith("_w")]

    def _conv(self, x, w):  # HWIO -> OIHW
        return F.conv2d(x, w.permute(3, 2, 0, 1), padding=w.shape[0] // 2)

    def _bn(self, x, key, dim=1):
        """batch norm over `dim` (1 = channel; 2 = board row for the stem, build_graph.py:68)"""
        g, b = self.p[key + "_g"], self.p[key + "_b"]
        m, v = getattr(self, key + "_m"), getattr(self, key + "_v")
        if dim != 1:
            x = x.transpose(1, dim)
        y = F.batch_norm(x, m, v, g, b, self.training, BN_MOMENTUM_TORCH, BN_EPS)
        return y.transpose(1, dim) if dim != 1 else y

    def forward(self, x):  # x [n,13,7,6] -> logits [n,43], v [n]
        h = F.relu(self._bn(self._conv(x, self.p["stem_w"]), "stem", dim=2))
        for b in range(self.blocks):
            t = F.relu(self._bn(self._conv(h, self.p[f"b{b}a_w"]), f"b{b}a"))
            t = self._bn(self._conv(t, self.p[f"b{b}b_w"]), f"b{b}b")
            h = F.relu(t + h)
        n = x.shape[0]
        pi = F.relu(self._bn(torch.einsum("nchw,co->nohw", h, self.p["pi_w"]), "pi"))
        pi = pi.permute(0, 2, 3, 1).reshape(n, 84)  # NHWC flatten: (y*6+x)*2 + c
        logits = pi @ self.p["pd_w"] + self.p["pd_b"]
        v = F.relu(self._bn(torch.einsum("nchw,co->nohw", h, self.p["v_w"]), "v"))
        v = v.reshape(n, 42)
        v = F.relu(v @ self.p["v1_w"] + self.p["v1_b"])
        v = torch.tanh(v @ self.p["v2_w"] + self.p["v2_b"]).reshape(n)
        return logits, v

    def losses(self, x, pi_t, z_t):
        logits, v = self(x)
        loss_pi = -(pi_t * F.log_softmax(logits, dim=1)).sum(1).mean()
        loss_v = F.mse_loss(v, z_t)
        l2 = L2_C * sum((w * w).sum() for w in self.kernels())
        return loss_pi, loss_v, l2

    def to_flat(self):
        flat = np.zeros(self.count, np.float32)
        for name, off, shape in self.lay:
            n = int(np.prod(shape))
            if name.endswith("_bn"):
                key = name[:-3]
                t = torch.stack([self.p[key + "_g"].detach(), self.p[key + "_b"].detach(), getattr(self, key + "_m"),
                                 getattr(self, key + "_v")])
            else:
                t = self.p[name].detach()
            flat[off:off + n] = t.cpu().numpy().reshape(-1)
        return flat


class Trainer:
    """AlphaZeroNNId::train (alphazero_gpu_cluster.h:31) for one net; keeps the Adam state across iterations like the
    TF session does."""

    def __init__(self, blocks, flat, device="cpu", batch_size=512, seed=0):
        self.device = torch.device(device)
        self.net = AzrNet(blocks, flat).to(self.device)
        self.opt = torch.optim.Adam(self.net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
        self.batch_size = batch_size
        self.rng = np.random.default_rng(seed)

    def load_flat(self, flat):
        fresh = AzrNet(self.net.blocks, flat).to(self.device)
        self.net.load_state_dict(fresh.state_dict())

    def flat(self):
        return self.net.to_flat()

    def train(self, rec265, epochs, log=None):
        """returns [(avg policy loss, avg value loss)] per epoch; remainder records of an epoch are dropped"""
        in88, pi, z = unpack_records(rec265)
        x = torch.from_numpy(planes_from_in88(in88))
        pi_t, z_t = torch.from_numpy(pi), torch.from_numpy(z)
        n, bs = x.shape[0], self.batch_size
        out = []
        self.net.train()
        for _ in range(epochs):
            order = torch.from_numpy(self.rng.permutation(n))
            lp = lv = 0.0
            nb = n // bs
            for c in range(nb):
                idx = order[c * bs:(c + 1) * bs]
                xb, pb, zb = x[idx].to(self.device), pi_t[idx].to(self.device), z_t[idx].to(self.device)
                self.opt.zero_grad(set_to_none=True)
                loss_pi, loss_v, l2 = self.net.losses(xb, pb, zb)
                (loss_pi + loss_v + l2).backward()
                self.opt.step()
                lp += loss_pi.item()
                lv += loss_v.item()
            if nb:
                out.append((lp / nb, lv / nb))
                if log:
                    log.write(f"{lp / nb}, {lv / nb}, ")
        if log:
            log.write("\n")
            log.flush()
        self.net.eval()
        return out
