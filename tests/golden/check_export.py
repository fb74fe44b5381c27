"""Build-container check (needs /root/reference; not part of the automated suites): a checkpoint written by THIS package's
``save_reference_format`` must load and run in the REFERENCE.

    python tests/golden/check_export.py write /tmp/x.pt     # this package writes (deterministic state, seed 33)
    python tests/golden/check_export.py read  /tmp/x.pt     # the reference loads it and compares with its own model

Two processes because both packages are called ``ultralytics``."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import graph as og  # noqa: E402

NAME, SEED = "yolov8n-ASF-P2P2", 33
mode, path = sys.argv[1], sys.argv[2]
if mode == "write":
    sys.path.insert(0, os.path.join(ROOT, "experiment-yolo_amd"))
    from ultralytics.nn.tasks import DetectionModel, save_reference_format
    cfg = os.path.join(ROOT, "experiment-yolo_amd", "ultralytics", "cfg", "models", NAME + ".yaml")
    m = DetectionModel(cfg, ch=3, verbose=False)
    m.load_state_dict(og.fill_state(og.state_layout(og.build_graph(og.load_yaml(cfg))), SEED), strict=True)
    save_reference_format(path, m, updates=5, epoch=1, train_args={"imgsz": 640})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
else:
    sys.path.insert(0, HERE)
    import _refimport
    _refimport.install()
    from ultralytics.nn.tasks import DetectionModel, attempt_load_one_weight
    cfg = os.path.join(_refimport.REF, "ultralytics/cfg/models", NAME + ".yaml")
    native = DetectionModel(cfg, ch=3, verbose=False)
    state = og.fill_state(og.state_layout(og.build_graph(og.load_yaml(cfg))), SEED)
    native.load_state_dict({k: (v.half().float() if v.is_floating_point() else v) for k, v in state.items()}, strict=True)
    native.eval()
    loaded, ckpt = attempt_load_one_weight(path)  # the reference's own loader: ckpt["model"].float().eval() + attribute fix-ups
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ya, fa = loaded(x)
        yb, fb = native(x)
    print("type:", type(loaded).__module__, type(loaded).__name__, "| max |dy|:", float((ya - yb).abs().max()),
          "| feats equal:", all(torch.equal(p, q) for p, q in zip(fa, fb)))
    assert torch.equal(ya, yb)
    print("reference loaded and ran the checkpoint written by this package: outputs identical")
