#!/usr/bin/env python3
"""Generates tests/golden/graph_v2_5.json from the reference's exported text GraphDef
(/root/reference/python/model/model_txt_V2_5.pb, written by python/src/build_graph.py:122-127 for the V2 feature
set and 5 residual blocks).  The file is DATA the reference ships (a protobuf text dump, no code); this script reads it
as text — TensorFlow is neither needed nor available — and keeps only what pins the net's shape side:

  * every trainable / moving-statistics variable: name, shape, dtype, initializer (uniform bounds, zeros, ones)
  * every Conv2D: strides, padding, data_format, input / kernel
  * every batch-norm layer: which function bodies its `cond` calls, their FusedBatchNormV3 epsilon / data_format /
    is_training, the moving-average decay, and — for conv_bn (built with axis=1 on NHWC, build_graph.py:68) — the
    transposes / reshapes that put the board ROW on the channel axis
  * dense layers (MatMul + BiasAdd), activations, the loss ops and the Adam / L2 constants

Run in the build container only:  python tests/golden/make_graph_golden.py
"""
import json
import os
import re
import struct
import sys

SRC = "/root/reference/python/model/model_txt_V2_5.pb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_v2_5.json")

TOKEN = re.compile(r'\s*(?:([A-Za-z_][A-Za-z0-9_]*)\s*(:)?|(\{)|(\})|("(?:[^"\\]|\\.)*")|([-+0-9.eE]+[A-Za-z]*|inf|-inf|nan))')


def parse(text):
    """protobuf text format -> nested {field: [values]} (every field is a list; messages are dicts)"""
    pos, n = 0, len(text)
    root = {}
    stack = [root]
    key = None
    while True:
        m = TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip():
                raise ValueError("parse error at %d: %r" % (pos, text[pos:pos + 40]))
            break
        pos = m.end()
        ident, colon, lb, rb, string, number = m.groups()
        if ident is not None and key is None:
            key = ident
            continue
        if lb:
            d = {}
            stack[-1].setdefault(key, []).append(d)
            stack.append(d)
            key = None
        elif rb:
            stack.pop()
        else:
            if string is not None:
                val = string[1:-1]
            elif number is not None:
                try:
                    val = int(number)
                except ValueError:
                    val = float(number)
            else:
                val = ident  # enum / bool literal
            stack[-1].setdefault(key, []).append(val)
            key = None
    return root


def unescape(s):
    """protobuf text string -> bytes"""
    out = bytearray()
    i = 0
    while i < len(s):
        c = s[i]
        if c != "\\":
            out.append(ord(c)); i += 1; continue
        i += 1
        c = s[i]
        if c in "01234567":
            j = i
            while j < len(s) and j < i + 3 and s[j] in "01234567":
                j += 1
            out.append(int(s[i:j], 8)); i = j
        else:
            out.append({"n": 10, "r": 13, "t": 9, "\\": 92, '"': 34, "'": 39}[c]); i += 1
    return bytes(out)


def attrs(node):
    return {a["key"][0]: a["value"][0] for a in node.get("attr", [])}


def shape_of(v):
    return [d["size"][0] for d in v["shape"][0].get("dim", [])]


def const_value(node):
    t = attrs(node)["value"]["tensor"][0]
    if "float_val" in t:
        return t["float_val"][0]
    if "int_val" in t:
        return t["int_val"][0]
    if "tensor_content" in t:
        raw = unescape(t["tensor_content"][0])
        if t["dtype"][0] == "DT_INT32":
            return list(struct.unpack("<%di" % (len(raw) // 4), raw))
        if t["dtype"][0] == "DT_FLOAT":
            return list(struct.unpack("<%df" % (len(raw) // 4), raw))
    return None


def main():
    g = parse(open(SRC).read())
    nodes = {n["name"][0]: n for n in g["node"]}
    funcs = {f["signature"][0]["name"][0]: f for f in g["library"][0]["function"]}

    variables = []
    for n in g["node"]:
        name = n["name"][0]
        if n["op"][0] != "VarHandleOp" or "/optimize" in name or name in ("beta1_power", "beta2_power"):
            continue
        a = attrs(n)
        init = None
        if name + "/Initializer/random_uniform/min" in nodes:
            init = {"kind": "uniform", "min": const_value(nodes[name + "/Initializer/random_uniform/min"]),
                    "max": const_value(nodes[name + "/Initializer/random_uniform/max"])}
        elif name + "/Initializer/zeros" in nodes:
            init = {"kind": "zeros"}
        elif name + "/Initializer/ones" in nodes:
            init = {"kind": "ones"}
        variables.append({"name": name, "shape": shape_of(a["shape"]), "dtype": a["dtype"]["type"][0], "init": init})

    convs = []
    for n in g["node"]:
        if n["op"][0] == "Conv2D":
            a = attrs(n)
            convs.append({"name": n["name"][0], "input": n["input"][0], "kernel": n["input"][1].split("/")[0] + "/kernel",
                          "strides": a["strides"]["list"][0]["i"], "padding": a["padding"]["s"][0],
                          "data_format": a["data_format"]["s"][0],
                          "dilations": a["dilations"]["list"][0]["i"]})

    def fn_bn(fname):
        out = []
        for nd in funcs[fname].get("node_def", []):
            if nd["op"][0] == "FusedBatchNormV3":
                a = attrs(nd)
                out.append({"epsilon": a["epsilon"]["f"][0], "data_format": a["data_format"]["s"][0],
                            "is_training": a["is_training"]["b"][0] == "true",
                            "exponential_avg_factor": a.get("exponential_avg_factor", {}).get("f", [None])[0]})
        return out

    bns = []
    for n in g["node"]:
        name = n["name"][0]
        if n["op"][0] in ("If", "StatelessIf") and name.endswith("_bn/cond") or (n["op"][0] in ("If", "StatelessIf") and re.fullmatch(r"bn[^/]*/cond", name)):
            a = attrs(n)
            tb, eb = a["then_branch"]["func"][0]["name"][0], a["else_branch"]["func"][0]["name"][0]
            layer = name.split("/")[0]
            decay = None
            for cand in (layer + "/AssignMovingAvg/sub/x", layer + "/cond_1/Identity"):
                pass
            # moving-average decay: cond_1 selects the constant momentum when training, 1.0 otherwise
            a1 = attrs(nodes[layer + "/cond_1"])
            tb1 = a1["then_branch"]["func"][0]["name"][0]
            for nd in funcs[tb1].get("node_def", []):
                if nd["op"][0] == "Const":
                    t = attrs(nd)["value"]["tensor"][0]
                    if "float_val" in t:
                        decay = t["float_val"][0]
            data_in = n["input"][1] if len(n["input"]) > 1 else None
            bns.append({"layer": layer, "data_input": data_in, "training": fn_bn(tb), "inference": fn_bn(eb), "momentum": decay})

    # conv_bn is built with axis=1 on an NHWC tensor (build_graph.py:68): the layer itself reshapes so that the board row
    # is the normalised axis; record the parameter shape and the reshape constants around it
    conv_bn_reshapes = {}
    for n in g["node"]:
        name = n["name"][0]
        if name.startswith("conv_bn/") and n["op"][0] in ("Reshape", "Transpose"):
            shp = n["input"][1]
            conv_bn_reshapes[name] = {"op": n["op"][0], "input": n["input"][0],
                                      "shape": const_value(nodes[shp]) if shp in nodes and nodes[shp]["op"][0] == "Const" else shp}

    dense = []
    for n in g["node"]:
        if n["op"][0] == "MatMul" and not n["name"][0].startswith("gradients"):
            a = attrs(n)
            dense.append({"name": n["name"][0], "input": n["input"][0], "kernel": n["input"][1].split("/")[0] + "/kernel",
                          "transpose_a": a["transpose_a"]["b"][0] == "true", "transpose_b": a["transpose_b"]["b"][0] == "true"})
    acts = [{"name": n["name"][0], "op": n["op"][0], "input": n["input"][0]} for n in g["node"]
            if n["op"][0] in ("Relu", "Tanh", "Softmax", "LogSoftmax") and not n["name"][0].startswith("gradients")]
    adds = [{"name": n["name"][0], "inputs": n["input"]} for n in g["node"]
            if n["op"][0] in ("Add", "AddV2") and not n["name"][0].startswith(("gradients", "optimize")) and re.match(r"(add|res)", n["name"][0])]
    reshapes = {}
    for n in g["node"]:
        if n["op"][0] == "Reshape" and not n["name"][0].startswith(("gradients", "conv_bn", "optimize")):
            shp = n["input"][1]
            if shp in nodes and nodes[shp]["op"][0] == "Const":
                reshapes[n["name"][0]] = {"input": n["input"][0], "shape": const_value(nodes[shp])}

    adam = {}
    for n in g["node"]:
        if n["op"][0] == "ResourceApplyAdam":
            adam.setdefault("apply_count", 0)
            adam["apply_count"] += 1
            if "inputs" not in adam:
                adam["inputs"] = n["input"]
                a = attrs(n)
                adam["use_nesterov"] = a["use_nesterov"]["b"][0] == "true"
    for key in ("optimize/learning_rate", "optimize/beta1", "optimize/beta2", "optimize/epsilon"):
        if key in nodes:
            adam[key.split("/")[1]] = const_value(nodes[key])
    consts = {}
    for n in g["node"]:
        if n["op"][0] == "Const" and not n["name"][0].startswith(("gradients", "save")):
            v = const_value(n)
            if isinstance(v, float) and ("mul" in n["name"][0].lower() or "l2" in n["name"][0].lower() or n["name"][0].endswith("/x") or n["name"][0].endswith("/y")):
                consts[n["name"][0]] = v
    loss_ops = [{"name": n["name"][0], "op": n["op"][0], "inputs": n["input"]} for n in g["node"]
                if n["op"][0] in ("SoftmaxCrossEntropyWithLogits", "SquaredDifference", "L2Loss", "AddN", "Square", "Sum")
                and not n["name"][0].startswith(("gradients", "optimize"))]
    placeholders = [{"name": n["name"][0], "op": n["op"][0],
                     "shape": shape_of(attrs(n)["shape"]) if "dim" in attrs(n)["shape"]["shape"][0] else None}
                    for n in g["node"] if n["op"][0] in ("Placeholder", "PlaceholderWithDefault")]

    # the inference subgraph, node by node, from the two outputs back to the placeholders / variables: enough for a
    # small interpreter (tests/test_graph_pin.py) to evaluate the reference's own graph structure in numpy
    fwd = {}

    def walk(ref):
        name = ref.split(":")[0]
        if name in fwd:
            return
        n = nodes[name]
        op = n["op"][0]
        ins = [i for i in n.get("input", []) if not i.startswith("^")]
        ent = {"op": op, "inputs": [i.split(":")[0] for i in ins]}
        fwd[name] = ent
        if op in ("VarHandleOp", "Placeholder"):
            ent["inputs"] = []
            return
        a = attrs(n)
        if op == "Conv2D":
            ent.update(strides=a["strides"]["list"][0]["i"], padding=a["padding"]["s"][0], data_format=a["data_format"]["s"][0])
        elif op == "If":   # a batch-norm layer: else-branch (training == false) = FusedBatchNormV3 on the moving statistics
            ent["op"] = "BatchNormCond"
            ent["inference"] = fn_bn(a["else_branch"]["func"][0]["name"][0])[0]
            ent["training"] = fn_bn(a["then_branch"]["func"][0]["name"][0])[0]
        elif op == "Const":
            ent["value"] = const_value(n)
        elif op == "MatMul":
            ent.update(transpose_a=a["transpose_a"]["b"][0] == "true", transpose_b=a["transpose_b"]["b"][0] == "true")
        elif op == "BiasAdd":
            ent["data_format"] = a.get("data_format", {"s": ["NHWC"]})["s"][0]
        for i in ent["inputs"]:
            walk(i)

    walk("output_policy")
    walk("output_value")

    out = {"forward_graph": fwd, "source": "python/model/model_txt_V2_5.pb (reference data file; V2 input planes, BLOCKS=5)",
           "node_count": len(g["node"]), "function_count": len(funcs), "placeholders": placeholders, "variables": variables,
           "conv2d": convs, "batch_norm": bns, "conv_bn_reshapes": conv_bn_reshapes, "dense": dense, "activations": acts,
           "residual_adds": adds, "reshapes": reshapes, "adam": adam, "float_consts": consts, "loss_ops": loss_ops}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT, "variables:", len(variables), "convs:", len(convs), "bn layers:", len(bns))


if __name__ == "__main__":
    sys.exit(main())
