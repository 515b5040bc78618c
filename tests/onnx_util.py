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
