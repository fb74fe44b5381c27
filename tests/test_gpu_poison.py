"""-m gpu: no kernel of the training step may read memory that nothing wrote.

Every buffer the engine obtains uninitialised (activation / gradient twins, scratch, weight-gradient slabs, loss workspace)
is filled with 0xFF bytes (NaN as fp16/fp32) when ``ultralytics.hip.engine.POISON`` is on, and the shared scratch is
re-poisoned on every request.  A step traced that way must produce the same losses, gradients and updated weights as a
clean one: a read-before-write shows up as a NaN (or as a difference) here instead of depending on what the caching
allocator returns -- the failure class behind a training run whose optimizer steps are all skipped (found_inf) without
any exception."""
import os

import numpy as np
import pytest
import torch

from conftest import CFG_DIR
from oracle import graph as og

pytestmark = pytest.mark.gpu


def _batch(B, S, nb, seed):
    rng = np.random.default_rng(seed)
    return dict(img=torch.from_numpy(rng.random((B, 3, S, S), dtype=np.float32)),
                batch_idx=torch.arange(B).repeat_interleave(nb).float(),
                cls=torch.from_numpy(rng.integers(0, 6, (B * nb, 1)).astype(np.float32)),
                bboxes=torch.from_numpy(np.concatenate([rng.random((B * nb, 2)) * 0.6 + 0.2, rng.random((B * nb, 2)) * 0.2 + 0.03], 1).astype(np.float32)))


def _steps(name, B, S, poison, accumulate, zero_p=False):
    import ultralytics.hip.engine as E
    from ultralytics.hip.train import StepPlan
    from ultralytics.nn.tasks import DetectionModel
    cfg = os.path.join(CFG_DIR, name + ".yaml")
    g = og.build_graph(og.load_yaml(cfg))
    m = DetectionModel(cfg, verbose=False)
    sd = og.fill_state(og.state_layout(g), 11)
    if zero_p:  # the reference's initialisation regime (p_conv.weight = 0, |offset| < 1): no far samples, so LDConv's input gradient
        for k in sd:  # takes no fp32 atomics and the whole step is bit-reproducible
            if k.endswith("p_conv.weight"):
                sd[k] = torch.zeros_like(sd[k])
            elif k.endswith("p_conv.bias"):
                sd[k] = sd[k].clamp(-0.9, 0.9)
    m.load_state_dict(sd, strict=True)
    E.POISON = poison
    try:
        m.cuda().train()
        plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=True)
        out = []
        for it in range(3):  # traced step, then two graph replays
            plan.set_hyper([1e-3, 1e-4, 1e-4], 0.9, [0.0, 5e-4, 0.0])
            plan.forward_backward(_batch(B, S, 4, it))
            if accumulate:
                plan.accumulate()
            plan.optimizer_step()
            torch.cuda.synchronize()
            out.append((plan.crit.scalars.cpu().clone(), plan.rt.flat_g.cpu().clone(), plan.rt.flat_p.cpu().clone(), plan.state.cpu().clone()))
    finally:
        E.POISON = False
    return out


@pytest.mark.parametrize("name,B,S,zero_p", [("yolov8n-ASF-P2P2", 2, 64, False), ("yolov8n-LD-P2", 2, 64, True), ("yolov8n-LD-P2", 2, 640, True),
                                             ("yolov8n-LD-P2", 2, 640, False), ("yolov8n-ASF-P2P2", 2, 640, False), ("yolov8n-ASF-P2", 2, 128, False)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_step_reads_nothing_uninitialised(name, B, S, zero_p, accumulate):
    clean = _steps(name, B, S, False, accumulate, zero_p)
    dirty = _steps(name, B, S, True, accumulate, zero_p)
    for it, ((s0, g0, p0, st0), (s1, g1, p1, st1)) in enumerate(zip(clean, dirty)):
        assert torch.isfinite(s1[5:9]).all(), f"step {it}: loss items {s1[5:9].tolist()} with poisoned buffers"
        assert torch.isfinite(g1).all(), f"step {it}: {int((~torch.isfinite(g1)).sum())} non-finite gradient words with poisoned buffers"
        assert st1[6] == 0 and st1[5] == it + 1, f"step {it}: optimizer state {st1.tolist()} (steps taken / skipped)"
        # With random p_conv weights LDConv's far-sample side pass uses fp32 atomics: the order-dependent rounding shows as ~1e-3 of
        # the largest gradient from one clean run to the next and feeds back through the updated weights, so that case only has to
        # stay finite and take every step (the poison is NaN: a real read shows as NaN).  Everything else is bit-reproducible.
        if "LD" in name and not zero_p:
            continue
        assert float((s0[5:9] - s1[5:9]).abs().max()) <= 0.0, (it, s0[5:9].tolist(), s1[5:9].tolist())
        assert float((g0 - g1).abs().max()) <= 0.0, f"step {it}: gradients differ with poisoned buffers"
        assert float((p0 - p1).abs().max()) <= 0.0, f"step {it}: weights differ with poisoned buffers"
