"""kfunca_amd — MI355X-native (gfx950) implementation of kfunca's tensor-kernel hot path.

    import kfunca_amd as kfunca        # same surface as the reference's `kfunca` module (src/register.cpp)

Layers:
    kfunca_amd._C            pybind11 module: Tensor / autograd / operator API (host C++, kfunca_amd/csrc/core)
    libkfunca_hip.so         the C-ABI device library (include/kfunca_hip.h): hand-written HIP kernels + RCCL
    kfunca_amd.hip_abi       ctypes binding of that C ABI (what a foreign host would bind)
    kfunca_amd.parallel      batch sharding + gradient all-reduce for 1..8 GPUs of one node

The native module is loaded on first use; if it has not been built the import fails loudly — there is
no Python or CPU fallback for any operator.
"""
import importlib

__all__ = ["device_info", "memstat", "dtype", "empty", "empty_like", "from_numpy", "to_numpy", "zeros",
           "causal_attention", "gemm", "cat", "tensor",
           # extensions over the reference surface
           "rms_norm", "layer_norm", "embedding", "causal_attention_qkv", "gemm_fused", "qkv_linear", "from_numpy_bf16", "device_count", "synchronize", "memstat_dict", "graph_begin", "graph_end", "graph_launch", "graph_destroy"]

_native = None


def _load():
    global _native
    if _native is None:
        try:
            _native = importlib.import_module("kfunca_amd._C")
        except ImportError as e:  # loud: no fallback
            raise ImportError("kfunca_amd._C is not built (run `python -m kfunca_amd._build`); "
                              "the operator API has no fallback path") from e
    return _native


_SUBMODULES = ("_build", "_C", "hip_abi", "parallel")


def __getattr__(name):
    # `from kfunca_amd import _build` asks the package for the attribute first: submodule names must fall through to the
    # import system (AttributeError), or a clean checkout could never build the native module it is about to be told is missing
    if name.startswith("__") or name in _SUBMODULES:
        raise AttributeError(name)
    return getattr(_load(), name)


def __dir__():
    return sorted(set(__all__) | set(dir(_load())))
