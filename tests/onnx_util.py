"""Test helper: real ONNX bytes from torch's TorchScript exporter without the `onnx` package.

torch serialises the ModelProto in C++; only a post-processing hook imports `onnx`.  The hook is a no-op for models without
onnxscript functions, so it is bypassed here.  (Nothing from the reference is involved: these are independent models that
exercise the parser and the compiler on a real exporter's output.)"""
import io
import warnings


def export(model, args, opset=17, input_names=("x",), output_names=("y",), dynamic_axes=None):
    import torch
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    saved = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto
    try:
        f = io.BytesIO()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(model.eval(), args, f, opset_version=opset, input_names=list(input_names), output_names=list(output_names),
                              dynamic_axes=dynamic_axes, dynamo=False)
        return f.getvalue()
    finally:
        onnx_proto_utils._add_onnxscript_fn = saved


def splice_if(then_bytes, else_bytes, shared_inputs, cond_nodes, cond, inputs, initializers=(), opset=17):
    """Two exported models with the same input / output signature -> one model `If(cond) {then} else {else}`: every name of a
    branch gets a prefix (except the shared graph inputs, which the branches read from the enclosing scope), its initializers
    move into the branch graph.  This is the shape of Silero VAD's ONNX file (one network per sample rate under an `If` on sr)."""
    from lele_amd.compiler import onnx_pb as pb
    arms, out_names = [], None
    for prefix, data in (("t_", then_bytes), ("e_", else_bytes)):
        g = pb.load(data).graph
        ren = lambda n, p=prefix: n if (n == "" or n in shared_inputs) else p + n  # noqa: E731
        for t in g.initializer:
            t.name = ren(t.name)
        for n in g.node:
            n.input, n.output = [ren(i) for i in n.input], [ren(o) for o in n.output]
            n.name = prefix + n.name
        outs = [v.name for v in g.output]
        out_names = out_names or outs
        assert outs == out_names, "the branches must produce the same outputs"
        for v in g.output:
            v.name = ren(v.name)
        arms.append(pb.Graph(g.node, [], g.output, g.initializer, prefix + "branch"))
    nodes = list(cond_nodes) + [pb.Node("If", [cond], out_names, then_branch=arms[0], else_branch=arms[1])]
    g = pb.Graph(nodes, inputs, [pb.ValueInfo(o, pb.FLOAT, None) for o in out_names], list(initializers))
    return pb.Model(g, opset=opset).serialize()
