"""-m gpu: the data-parallel step itself.  Two rank processes share the one GPU and exchange over gloo --
the same StepPlan code path a multi-GPU run takes with RCCL: forward/backward on the rank's shard, accumulate(), ONE all-reduce
of [gradients | float buffers], optimizer_step().  No multi-GPU oracle exists (the reference's DDP path needs CUDA devices), so
the result is compared with the single-process emulation the survey defines: the W shards go through ONE replica in train mode
one after another (per-shard BatchNorm statistics), their gradients of L_r * b are summed, one optimizer step follows, and the
running-statistic buffers must be rank 0's."""
import os
import subprocess
import sys

import pytest
import torch

from dp_common import STEPS, build_model, global_batch, hyper

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emulate(world, b):
    from ultralytics.hip.dist import shard_batch
    from ultralytics.hip.train import StepPlan
    m = build_model().cuda().train()
    plan = StepPlan(m, b, 64, nmax=8, optimizer="SGD", world_size=1, use_graph=False, init_scale=1.0, dynamic_scale=False)
    rt = plan.rt
    for it, accumulate in enumerate(STEPS):
        buf_rank0 = rt.flat_b.clone()  # rank 0's running statistics evolve through ITS shards only
        start = rt.flat_b.clone()
        for r in range(world):  # every rank starts the iteration from the same buffers (they were exchanged at the last step)
            rt.flat_b.copy_(start)
            for micro in range(accumulate):
                plan.set_hyper(*hyper(it))
                plan.forward_backward(shard_batch(global_batch(it, micro, b * world), r, world))
                plan.accumulate()
            if r == 0:
                buf_rank0 = rt.flat_b.clone()
        rt.flat_b.copy_(buf_rank0)
        plan.optimizer_step()
    torch.cuda.synchronize()
    return rt.flat_p.cpu(), rt.flat_b.cpu(), plan.ema.cpu()


@pytest.mark.parametrize("steps,buckets", [("1,1,1", 1), ("1,2,1", 1), ("1,1,1", 2), ("1,2,1", 2)])
def test_two_rank_step_equals_the_single_process_emulation(tmp_path, steps, buckets, monkeypatch):
    """steps = micro-batches per optimizer step.  Without accumulation the two ranks' sum g0 + g1 has ONE order: the replicas must
    equal the emulation bit for bit.  With an accumulating step the rank-wise sum (a + b) + (c + d) and the emulation's chain
    ((a + b) + c) + d differ by fp32 rounding (measured 4e-9 on the weights right after such a step); the step that follows runs on
    fp16 activations, which turn that into isolated last-bit flips: 5e-10 ... 1.5e-6 depending on the trajectory (round 3: the same
    schedule gave 5e-10 with the imported stem input and 1.5e-6 with the direct stem, 4e-9 for '2,1' either way).
    buckets = 2 (DY_DP_BUCKETS): the backward list is cut where the neck + head end and their gradients are all-reduced on a side stream
    while the backbone's backward runs (hip/train.py, _start_bucket1) -- the same sums, so the same bits."""
    import dp_common
    world, b = 2, 2
    port = 29500 + os.getpid() % 2000
    monkeypatch.setenv("DY_TEST_DP_STEPS", steps)
    monkeypatch.setenv("DY_DP_BUCKETS", str(buckets))
    monkeypatch.setattr(dp_common, "STEPS", [int(v) for v in steps.split(",")])
    monkeypatch.setattr(sys.modules[__name__], "STEPS", dp_common.STEPS)
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    ranks = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert torch.equal(ranks[0]["p"], ranks[1]["p"]) and torch.equal(ranks[0]["b"], ranks[1]["b"]), "replicas diverged"
    p, bufs, ema = _emulate(world, b)
    # same kernels, same per-shard arithmetic; only the order of the fp32 gradient sum differs (gloo ring vs axpy chain)
    dp = float((ranks[0]["p"] - p).abs().max() / p.abs().max())
    db = float((ranks[0]["b"] - bufs).abs().max() / bufs.abs().max())
    de = float((ranks[0]["ema"] - ema).abs().max() / ema.abs().max())
    print(f"2 ranks vs emulation: weights {dp:.2e}, buffers {db:.2e}, EMA {de:.2e} (relative to the largest entry)")
    if "2" not in steps:
        assert dp == 0.0 and db == 0.0 and de == 0.0
    assert dp < 1e-5 and db < 5e-5 and de < 1e-5


@pytest.mark.parametrize("buckets", [1, 2])
def test_rccl_runs_the_collective_path_on_the_gpu(tmp_path, buckets):
    """The box has one GPU and RCCL refuses two ranks on one device, so the N > 1 replicas above talk over gloo through a host copy.
    This runs the SAME step over RCCL itself: a one-rank "nccl" group whose StepPlan is told world_size = 2, so that every
    collective of the data-parallel path is issued on device tensors -- the flat [gradients | buffers] all-reduce, and with
    DY_DP_BUCKETS=2 the first bucket on the side stream beside the backbone's backward -- and must leave exactly the weights the
    gloo form of the same one-rank run leaves (a one-rank all-reduce is the identity in both)."""
    port = 29500 + (os.getpid() + 7 * buckets) % 2000
    res = {}
    for backend in ("nccl", "gloo"):
        out = tmp_path / backend
        out.mkdir()
        env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", DY_TEST_DP_BACKEND=backend, DY_TEST_DP_PLAN_WORLD="2", DY_TEST_DP_STEPS="1,2,1",
                   DY_DP_BUCKETS=str(buckets))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(out)], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=300)
        assert p.returncode == 0, p.stdout
        res[backend] = torch.load(os.path.join(out, "rank0.pt"))
    for k in ("p", "b", "ema"):
        assert torch.equal(res["nccl"][k], res["gloo"][k]), k


def test_gradient_buckets_are_contiguous_views_of_the_flat_buffer():
    """The flat parameter / gradient buffers are laid out [neck + head: bias | decayed | norm][backbone: bias | decayed | norm] (hip/runtime.py),
    so that each gradient bucket of the bucketed all-reduce is ONE slice reduced in place (DDP's reducer works on contiguous buckets:
    reference engine/trainer.py:694-695) -- no index gathers, no staging copies between the two graphs -- and the optimizer kernel
    takes the six segment bounds."""
    from conftest import CFG_DIR
    from ultralytics.hip.train import StepPlan
    from ultralytics.nn.tasks import DetectionModel
    torch.manual_seed(0)
    m = DetectionModel(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml"), verbose=False).cuda().train()
    plan = StepPlan(m, 2, 64, nmax=8)
    rt = plan.rt
    nb = len(m.yaml["backbone"])
    b = [0] + list(rt.seg_bounds) + [rt.n_params_flat]
    assert b == sorted(b) and rt.bucket_split == b[3] and 0 < rt.bucket_split < rt.n_params_flat
    norm = tuple(v for k, v in torch.nn.__dict__.items() if "Norm" in k and isinstance(v, type))
    mods = dict(m.named_modules())
    for name, p in m.named_parameters():
        o, late = rt.param_off[name], int(name.split(".")[1]) >= nb
        grp = 0 if "bias" in name else (2 if isinstance(mods[name.rsplit(".", 1)[0]], norm) else 1)
        seg = next(k for k in range(6) if b[k] <= o < b[k + 1])
        assert seg == (0 if late else 3) + grp and rt.param_group[name] == grp, name
        assert (o < rt.bucket_split) == late
        assert p.data.data_ptr() == rt.flat_p.data_ptr() + 4 * o and p.grad.data_ptr() == rt.flat_gb.data_ptr() + 4 * o
    assert not hasattr(plan, "_bucket_stage") and not hasattr(plan, "_bucket_positions")
    # the two buckets tile the exchange buffer [gradients | buffer tail] exactly
    b1, b2 = rt.flat_gb[:rt.bucket_split], rt.flat_gb[rt.bucket_split:]
    assert b1.is_contiguous() and b2.is_contiguous() and b1.numel() + b2.numel() == rt.flat_gb.numel()
    assert b2.data_ptr() == rt.flat_gb.data_ptr() + 4 * rt.bucket_split
