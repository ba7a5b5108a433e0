"""CPU: pins the SHAPE side of the net rows to the graph the reference ships.

tests/golden/graph_v2_5.json is extracted by tests/golden/make_graph_golden.py from the reference's own exported
GraphDef text (python/model/model_txt_V2_5.pb: V2 planes, 5 blocks).  Checked here, without TensorFlow:
  * the AZRW flat vector (azr_nn_param_count, the oracle's and the torch reference's layout) is exactly
    the graph's variable list in graph order, shape by shape; the initialiser is the graph's (Glorot bounds, BN identity)
  * the oracle's fp32 forward equals a small numpy interpreter that walks the graph's inference subgraph node by node
    (Conv2D NHWC/SAME, the `cond` batch-norm layers with THEIR data_format — conv_bn is NCHW on an NHWC tensor, i.e.
    normalised over the board row — and epsilon, Relu / Add / Reshape / MatMul / BiasAdd / Softmax / Tanh)
  * the constants of the optimiser step (Adam, L2 scale, BN momentum / epsilon) used by the float64 reference the HIP
    step is tested against (tests/torch_train_ref.py) are the graph's
Numeric values of the reference's TensorFlow kernels stay "parity unpinned" (DESIGN.md §5): this pins structure."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import azr_testlib as T
from gpu_common import pkg

G = json.load(open(os.path.join(T.GOLDEN, "graph_v2_5.json")))
BLOCKS = 5


def graph_layout():
    """[(variable name, offset, shape)] in graph order"""
    off, out = 0, []
    for v in G["variables"]:
        out.append((v["name"], off, tuple(v["shape"])))
        off += int(np.prod(v["shape"]))
    return out, off


def test_azrw_layout_is_the_graphs_variable_list():
    lay, total = graph_layout()
    assert total == T.oracle().orc_net_param_count(BLOCKS) == pkg().load_library().azr_nn_param_count(BLOCKS)
    # the test library's view: kernels one entry each, a batch-norm layer = gamma, beta, moving_mean, moving_variance
    it = iter(lay)
    for name, off, n in T.net_layout(BLOCKS):
        if name.endswith("_bn"):
            c = n // 4
            for suffix in ("gamma", "beta", "moving_mean", "moving_variance"):
                vname, voff, shape = next(it)
                assert vname.endswith("/" + suffix) and shape == (c,) and voff == off, (name, vname)
                off += c
        else:
            vname, voff, shape = next(it)
            assert voff == off and int(np.prod(shape)) == n, (name, vname, shape)
            assert vname.endswith("/kernel" if name.endswith("_w") else "/bias"), (name, vname)
    assert next(it, None) is None
    # the float64 torch reference of the optimiser-step tests uses the same offsets
    import torch_train_ref as R
    rl, rtotal = R.layout(BLOCKS)
    assert rtotal == total
    starts = {off for _, off, _ in lay}
    assert {off for _, off, _ in rl} <= starts
    # shapes the kernels assume: HWIO 3x3 convs, 1x1 head convs, the stem's BN has one parameter per board ROW
    byname = {v["name"]: v for v in G["variables"]}
    assert byname["conv/kernel"]["shape"] == [3, 3, 13, 256] and byname["conv_bn/gamma"]["shape"] == [7]
    assert byname["res0a_branch2a/kernel"]["shape"] == [3, 3, 256, 256] and byname["bn0a_branch2a/gamma"]["shape"] == [256]
    assert byname["pi/kernel"]["shape"] == [1, 1, 256, 2] and byname["v/kernel"]["shape"] == [1, 1, 256, 1]
    assert byname["dense/kernel"]["shape"] == [84, 43] and byname["dense_1/kernel"]["shape"] == [42, 256]
    assert byname["dense_2/kernel"]["shape"] == [256, 1]
    assert len(G["conv2d"]) == 2 * BLOCKS + 3
    assert all(c["strides"] == [1, 1, 1, 1] and c["padding"] == "SAME" and c["data_format"] == "NHWC" for c in G["conv2d"])


def test_random_init_is_the_graphs_initializer():
    flat = T.make_net_flat(BLOCKS, seed=123)
    for (name, off, shape), v in zip(graph_layout()[0], G["variables"]):
        w = flat[off:off + int(np.prod(shape))]
        init = v["init"]
        if init["kind"] == "uniform":   # glorot_uniform: +-sqrt(6 / (fan_in + fan_out)), as the graph's min / max constants
            fan_in = int(np.prod(shape[:-1]))
            fan_out = int(np.prod(shape[:-2])) * shape[-1] if len(shape) > 2 else shape[-1]
            assert abs(init["max"] - np.sqrt(6.0 / (fan_in + fan_out))) < 1e-7 and init["min"] == -init["max"]
            assert w.min() >= init["min"] - 1e-9 and w.max() <= init["max"] + 1e-9
            if w.size > 500:
                assert w.max() > 0.97 * init["max"] and w.min() < 0.97 * init["min"] and abs(w.mean()) < 0.1 * init["max"]
        elif init["kind"] == "ones":
            assert (w == 1).all(), name
        else:
            assert (w == 0).all(), name


def conv2d_same_nhwc(x, k):
    n, h, w, ci = x.shape
    kh, kw, _, co = k.shape
    xp = np.zeros((n, h + kh - 1, w + kw - 1, ci), np.float64)
    xp[:, kh // 2:kh // 2 + h, kw // 2:kw // 2 + w] = x
    y = np.zeros((n, h, w, co), np.float64)
    for dy in range(kh):
        for dx in range(kw):
            y += np.einsum("nhwc,co->nhwo", xp[:, dy:dy + h, dx:dx + w], k[dy, dx].astype(np.float64))
    return y


def interpret(in88, flat):
    """evaluate graph_v2_5.json's inference subgraph (training = false) in float64 numpy"""
    orc = T.oracle()
    n = len(in88)
    planes = np.zeros((n, 7, 6, 13), np.float32)
    for i in range(n):
        orc.orc_planes(T.ptr(in88[i]), T.ptr(planes[i]))       # setInStateTensor (alphazero_nn.cpp:31-67)
    var = {name: flat[off:off + int(np.prod(shape))].reshape(shape).astype(np.float64) for name, off, shape in graph_layout()[0]}
    fwd, memo = G["forward_graph"], {}

    def ev(name):
        if name in memo:
            return memo[name]
        e = fwd[name]
        op, ins = e["op"], e["inputs"]
        if op == "Placeholder":
            r = planes.astype(np.float64) if name == "input_state" else False
        elif op == "VarHandleOp":
            r = var[name]
        elif op in ("ReadVariableOp", "Identity"):
            r = ev(ins[0])
        elif op == "Squeeze":
            r = None
        elif op == "Const":
            r = e["value"]
        elif op == "Conv2D":
            assert e["data_format"] == "NHWC" and e["padding"] == "SAME"
            r = conv2d_same_nhwc(ev(ins[0]), ev(ins[1]))
        elif op == "BatchNormCond":   # inputs: pred, gamma, beta, moving_mean, moving_variance, x
            gamma, beta, mean, variance, x = (ev(i) for i in ins[1:6])
            inf = e["inference"]
            assert not inf["is_training"]
            axis = 1 if inf["data_format"] == "NCHW" else 3
            shp = [1, 1, 1, 1]
            shp[axis] = x.shape[axis]
            assert gamma.shape == (x.shape[axis],), (name, gamma.shape, x.shape)
            r = (x - mean.reshape(shp)) / np.sqrt(variance.reshape(shp) + inf["epsilon"]) * gamma.reshape(shp) + beta.reshape(shp)
        elif op == "Relu":
            r = np.maximum(ev(ins[0]), 0)
        elif op in ("Add", "AddV2"):
            r = ev(ins[0]) + ev(ins[1])
        elif op == "Reshape":
            r = ev(ins[0]).reshape(ev(ins[1]))
        elif op == "MatMul":
            assert not e["transpose_a"] and not e["transpose_b"]
            r = ev(ins[0]) @ ev(ins[1])
        elif op == "BiasAdd":
            r = ev(ins[0]) + ev(ins[1])
        elif op == "Softmax":
            z = ev(ins[0])
            z = np.exp(z - z.max(-1, keepdims=True))
            r = z / z.sum(-1, keepdims=True)
        elif op == "Tanh":
            r = np.tanh(ev(ins[0]))
        else:
            raise AssertionError(f"unhandled op {op} at {name}")
        memo[name] = r
        return r

    return ev("output_policy"), ev("output_value").reshape(-1)


def test_oracle_forward_equals_the_reference_graph_structure(orc):
    x = np.load(os.path.join(T.GOLDEN, "encode.npz"))["in88"][::97][:12].copy()
    flat = T.make_net_flat(BLOCKS, seed=5, perturb_bn=True)
    pi, v = interpret(x, flat)
    net = T.OrcNet(BLOCKS, flat.ctypes.data_as(T.f32p))
    opi = np.zeros((len(x), 43), np.float32)
    ov = np.zeros(len(x), np.float32)
    orc.orc_net_forward(C.byref(net), T.ptr(x), len(x), T.ptr(opi), T.ptr(ov))
    dpi, dv = np.abs(opi - pi).max(), np.abs(ov - v).max()
    print(f"oracle fp32 forward vs float64 interpretation of the reference graph: max|dpi|={dpi:.2e} max|dv|={dv:.2e}")
    assert dpi <= 2e-5 and dv <= 2e-5
    # the stem's batch norm really acts per board ROW: scaling row 3's gamma changes the outputs, and an NHWC (per
    # channel) reading of the same 7 parameters is impossible (256 channels)
    lay = {nm: off for nm, off, _ in graph_layout()[0]}
    f2 = flat.copy()
    f2[lay["conv_bn/gamma"] + 3] *= 1.5
    pi2, v2 = interpret(x, f2)
    orc.orc_net_forward(C.byref(T.OrcNet(BLOCKS, f2.ctypes.data_as(T.f32p))), T.ptr(x), len(x), T.ptr(opi), T.ptr(ov))
    assert np.abs(pi2 - pi).max() > 1e-4 and np.abs(opi - pi2).max() <= 2e-5 and np.abs(ov - v2).max() <= 2e-5


def test_optimiser_constants_are_the_graphs():
    import torch_train_ref as R
    adam = G["adam"]
    f32 = np.float32
    assert f32(adam["learning_rate"]) == f32(1e-3) and f32(adam["beta1"]) == f32(0.9) and f32(adam["beta2"]) == f32(0.999)
    assert f32(adam["epsilon"]) == f32(1e-8) and adam["use_nesterov"] is False
    # one ResourceApplyAdam per trainable tensor: kernels, gamma/beta of every BN, dense biases (moving statistics are not trained)
    trainable = [v for v in G["variables"] if not v["name"].split("/")[-1].startswith("moving_")]
    assert adam["apply_count"] == len(trainable) == 45
    l2 = {k: v for k, v in G["float_consts"].items() if k.endswith("/Regularizer/mul/x")}
    assert len(l2) == 2 * BLOCKS + 3 + 3 and all(f32(v) == f32(1e-3) for v in l2.values())   # every conv + dense KERNEL, no bias / BN
    assert f32(R.L2_C) == f32(1e-3) and f32(R.BN_EPS) == f32(1e-3) and abs((1 - R.BN_MOMENTUM_TORCH) - 0.99) < 1e-12
    for bn in G["batch_norm"]:
        assert f32(bn["momentum"]) == f32(0.99), bn["layer"]
        for mode in ("training", "inference"):
            assert len(bn[mode]) == 1 and f32(bn[mode][0]["epsilon"]) == f32(1e-3)
            assert bn[mode][0]["data_format"] == ("NCHW" if bn["layer"] == "conv_bn" else "NHWC")
    ops = {n["op"] for n in G["loss_ops"]}
    assert {"SoftmaxCrossEntropyWithLogits", "SquaredDifference"} <= ops
    assert [p["shape"] for p in G["placeholders"] if p["name"] in ("input_state", "target_policy", "target_value")] == \
        [[-1, 7, 6, 13], [-1, 43], [-1, 1]]
