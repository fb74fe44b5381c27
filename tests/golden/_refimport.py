"""Import harness for the *reference* implementation (generation-time only).

Used only by tests/golden/make_golden.py inside the build container, where
/root/reference exists.  Nothing under tests/ that runs on the GPU box imports
this file.  It contains no reference code: it only arranges for
``import ultralytics`` to resolve to /root/reference with permissive stubs for
the third-party packages that are absent offline (cv2, torchvision, timm, ...),
following the recipe recorded in SURVEY.md section 8(c).
"""
import importlib.abc
import importlib.machinery
import os
import socket
import sys
import tempfile
import types

REF = os.environ.get("DEALYOLO_REFERENCE", "/root/reference")
_STUB_ROOTS = {
    "cv2", "torchvision", "timm", "thop", "seaborn", "mmcv", "mmengine", "pywt",
    "efficientnet_pytorch", "torch_dct", "natten", "lancedb", "cpuinfo", "torchmetrics",
    "albumentations", "dill", "hub_sdk", "pycocotools", "lap", "wandb", "clearml", "comet_ml",
    "mlflow", "neptune", "ray", "dvclive", "tensorboard", "DCNv3", "DCNv4", "swattention",
    "mamba_ssm", "selective_scan_cuda", "causal_conv1d", "causal_conv1d_cuda", "triton_never",
}


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _Dummy(metaclass=_Meta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__") and name not in ("__version__",):
            raise AttributeError(name)
        if name == "__version__":
            return "0.0.0"
        cls = type(name, (_Dummy,), {})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def install():
    """Make ``import ultralytics`` resolve to the reference tree."""
    if any(isinstance(f, _Finder) for f in sys.meta_path):
        return
    os.environ.setdefault("YOLO_CONFIG_DIR", tempfile.mkdtemp(prefix="yolo_cfg_"))
    os.environ.setdefault("YOLO_VERBOSE", "False")
    real = socket.create_connection

    def _offline(*a, **k):
        raise OSError("offline")

    socket.create_connection = _offline
    sys.meta_path.insert(0, _Finder())
    # the product's drop-in package is also called ``ultralytics``: make sure the reference wins here
    sys.path[:] = [p for p in sys.path if "experiment-yolo_amd" not in p]
    sys.path.insert(0, REF)
    try:
        import ultralytics  # noqa: F401
    finally:
        socket.create_connection = real
    assert ultralytics.__file__.startswith(REF), ultralytics.__file__
