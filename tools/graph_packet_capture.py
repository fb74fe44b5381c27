#!/usr/bin/env python3
"""Evidence for the hipGraph defect of this ROCm that hip/__init__.py works around.

A training step captured into a hipGraph is replayed after a burst of ordinary launches issued by the SAME thread between two
replays: N eval-mode forwards of the model (~170 launches each), the eager traces of N other launch lists (~540 each), N tiny
launches of one library kernel, N torch element-wise launches, N torch multi-tensor launches (fat argument blocks: does the size of
the kernel arguments decide, or whose kernels they are?), N launches of the same tiny kernel from a second copy of the library.  With the runtime's graph packet capture on
(DEBUG_CLR_GRAPH_PACKET_CAPTURE=1, the default of ROCm 7.2) the replay after ~1,000 library launches returns non-finite gradients
while the same recorded launch list issued eagerly is fine; with the flag at 0 every case is finite.  Also shown: StepPlan's
capture check (two eager passes over the list, then the verification replay) catches the defect at capture time.

    python tools/graph_packet_capture.py            -> table on stdout (run on the GPU box; ~1 minute)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("fwd", 3), ("fwd", 6), ("fwd", 20), ("trace", 1), ("trace", 2), ("trace", 3), ("crit", 4), ("burst", 5000), ("torchburst", 20000),
         ("torchfat", 5000), ("torchfat", 20000), ("burstcopy", 5000), ("burstcopy", 20000)]  # round 3: torch launches with a FAT argument block (multi-tensor apply: ~4 KB of kernarg each)

CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(ROOT, "experiment-yolo_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel
mode, N = sys.argv[1], int(sys.argv[2])
def batch(B, S, nb, seed):
    rng = np.random.default_rng(seed)
    return dict(img=torch.from_numpy(rng.random((B, 3, S, S), dtype=np.float32)), batch_idx=torch.arange(B).repeat_interleave(nb).float(),
                cls=torch.from_numpy(rng.integers(0, 6, (B * nb, 1)).astype(np.float32)),
                bboxes=torch.from_numpy(np.concatenate([rng.random((B * nb, 2)) * 0.6 + 0.2, rng.random((B * nb, 2)) * 0.2 + 0.1], 1).astype(np.float32)))
torch.manual_seed(0)
m = DetectionModel(os.path.join(ROOT, "experiment-yolo_amd", "ultralytics", "cfg", "models", "yolov8n-ASF-P2P2.yaml"), verbose=False).cuda().train()
B, S = 4, 64
try:
    plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=True, dynamic_scale=False)
    def step(i):
        plan.set_hyper([1e-3] * 3, 0.9, [0.0, 5e-4, 0.0])
        plan.forward_backward(batch(B, S, 3, i)); plan.optimizer_step(); torch.cuda.synchronize()
        return bool(torch.isfinite(plan.rt.flat_g).all())
    step(0); step(1)
except RuntimeError as e:
    print("capture check raised:", str(e)[:110]); sys.exit(0)
if mode == "crit":
    for k in range(N): plan.crit(plan.ho, batch(B, S, 3, 100 + k))
elif mode == "trace":
    keep = []
    for k in range(N):
        s2 = 32 * (1 + k % 3)
        o = StepPlan(m, B, s2, nmax=8, init_scale=1.0, use_graph=False, dynamic_scale=False, share=plan)
        o.forward_backward(batch(B, s2, 3, 200 + k)); keep.append(o)
elif mode == "burst":
    t = [torch.ones(16, device="cuda") for _ in range(4)] + [torch.zeros(64, device="cuda")]
    for k in range(N): plan.eng.call("dy_bn_eval_coef", *[x.data_ptr() for x in t], 16, 1e-3)
elif mode == "torchburst":
    x = torch.zeros(1024, device="cuda")
    for k in range(N): x.add_(1.0)
elif mode == "burstcopy":  # the SAME small kernel, launched from a second copy of the shared library (its own code object / hipModule)
    import ctypes as C, shutil, tempfile
    from ultralytics.hip import LIB_PATH
    cp = os.path.join(tempfile.mkdtemp(), "libdealyolo_hip_copy.so"); shutil.copy(LIB_PATH, cp)
    L2 = C.CDLL(cp)
    L2.dy_bn_eval_coef.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_float, C.c_void_p]
    t = [torch.ones(16, device="cuda") for _ in range(4)] + [torch.zeros(64, device="cuda")]
    st = torch.cuda.current_stream().cuda_stream
    for k in range(N): L2.dy_bn_eval_coef(*[x.data_ptr() for x in t], 16, 1e-3, st)
elif mode == "torchfat":
    xs = [torch.zeros(256, device="cuda") for _ in range(100)]
    for k in range(N): torch._foreach_add_(xs, 1.0)   # one multi_tensor_apply launch: a table of 100 addresses + sizes by value
elif mode == "fwd":
    m.eval()
    with torch.no_grad():
        for k in range(N): m(torch.rand(B, 3, S, S, device="cuda"))
    m.train()
torch.cuda.synchronize()
ok = step(2)
plan.eng.replay(plan.rec_fb); torch.cuda.synchronize()
print("graph replay after the burst:", "finite" if ok else "NON-FINITE gradients", "| same list issued eagerly:", "finite" if bool(torch.isfinite(plan.rt.flat_g).all()) else "NON-FINITE")
'''


def run(env, mode, n):
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD, mode, str(n)], capture_output=True, text=True, env=e, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    return lines[-1] if lines else ("error: " + p.stderr.strip().splitlines()[-1][:150] if p.stderr.strip() else "no output")


if __name__ == "__main__":
    print("A. packet capture ON (the runtime's default), StepPlan's capture check as shipped:")
    print("   ", run({"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1", "DY_ALLOW_UNSAFE_GRAPHS": "1"}, "fwd", 0))
    for name, env in (("B. packet capture ON, capture check without its burst (so that the later corruption can be seen)",
                       {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1", "DY_ALLOW_UNSAFE_GRAPHS": "1", "DY_VERIFY_BURST": "0"}),
                      ("C. packet capture OFF (what importing ultralytics sets), capture check as shipped", {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"})):
        print(name + ":")
        for mode, n in CASES:
            print(f"    {mode:10s} x {n:<6d} {run(env, mode, n)}", flush=True)
