"""ONNX -> device plan compiler.  See lower.py; plans run with `lele_amd.plan.Runner`."""
from .lower import CompileError, compile_model  # noqa: F401
