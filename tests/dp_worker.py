"""Rank process of tests/test_gpu_dp.py: one of W ranks that share GPU 0 and talk over gloo (the rehearsal path of bench.py and
DetectionTrainer): runs the data-parallel StepPlan -- forward/backward on its shard, accumulate(), the ONE all-reduce,
optimizer_step() -- and writes its final weights / buffers for the parent to compare with the single-process emulation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "experiment-yolo_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

from dp_common import STEPS, build_model, global_batch, hyper


def main(out_dir):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("DY_TEST_DP_BACKEND", "gloo")
    if backend == "nccl":  # RCCL: device tensors go into the collective as they are (hip/dist.py's non-gloo branch)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    # a one-rank group whose plan believes in `plan_world` ranks: every collective of the N > 1 path is issued (and is the identity)
    plan_world = int(os.environ.get("DY_TEST_DP_PLAN_WORLD", world))
    from ultralytics.hip.dist import shard_batch
    from ultralytics.hip.train import StepPlan
    m = build_model().cuda().train()
    b = 2
    plan = StepPlan(m, b, 64, nmax=8, optimizer="SGD", world_size=plan_world, use_graph=True, init_scale=1.0, dynamic_scale=False)
    for it, accumulate in enumerate(STEPS):
        for micro in range(accumulate):
            plan.set_hyper(*hyper(it))
            plan.forward_backward(shard_batch(global_batch(it, micro, b * world), rank, world), exchange=accumulate == 1)
            if accumulate > 1:
                plan.accumulate()
        plan.all_reduce()
        plan.optimizer_step()
    torch.cuda.synchronize()
    if os.environ.get("DY_DP_BUCKETS") == "2":
        assert plan.buckets == 2 and plan.fb_cut is not None and plan.graph_fb2 is not None, "the bucketed path was asked for and not taken"
    taken, skipped, _ = plan.check_progress()
    assert (taken, skipped) == (len(STEPS), 0)
    torch.save({"p": plan.rt.flat_p.cpu(), "b": plan.rt.flat_b.cpu(), "ema": plan.ema.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
