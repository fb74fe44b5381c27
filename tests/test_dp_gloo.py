"""CPU, world_size 2 over gloo: the data-parallel exchange of the training step.

The reference has no multi-GPU oracle, so DP is pinned by the equivalence of section 8e: each rank
back-propagates its own shard (own BN batch statistics, own target_scores_sum) and the applied gradient is the SUM over
ranks.  Here the two ranks run the CPU oracle on their shards, exchange with the product's collective helpers
(ultralytics/hip/dist.py: one all-reduce of ONE flat buffer + rank-0 buffer broadcast), and rank 0 checks the result
against the single-process emulation."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import CFG_DIR, PKG, ROOT


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from golden.cases import synth_batch
    from oracle import graph as og, loss as ol, nn as onn
    from ultralytics.hip.dist import all_reduce_flat, broadcast_buffers, shard_batch
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml")))
    sd = og.fill_state(og.state_layout(g), 5)
    names = [k for k in sd if og.is_param(k) and ".dfl." not in k]
    full = synth_batch(77, 4, 3, g.nc)
    mine = shard_batch(full, rank, world)

    def local_grads(state, batch):
        leaves = {k: state[k].clone().requires_grad_(True) for k in names}
        st = dict(state)
        st.update(leaves)
        feats = onn.forward(g, st, batch["img"], training=True)
        loss, _ = ol.detection_loss(feats, batch, g.strides, g.nc)
        gr = torch.autograd.grad(loss, [leaves[k] for k in names])
        return torch.cat([x.reshape(-1) for x in gr])

    # rank-0 buffers win before the forward (DDP broadcast_buffers)
    bufs = [k for k in sd if k.endswith(("running_mean", "running_var"))]
    flat_b = torch.cat([sd[k].reshape(-1) for k in bufs]) + rank  # make ranks disagree first
    broadcast_buffers(flat_b, 0)
    o = 0
    for k in bufs:
        sd[k] = flat_b[o:o + sd[k].numel()].view(sd[k].shape).clone()
        o += sd[k].numel()
    flat = local_grads(sd, mine)
    all_reduce_flat(flat, world)
    if rank == 0:
        sd0 = og.fill_state(og.state_layout(g), 5)
        ref = sum(local_grads({k: v.clone() for k, v in sd0.items()}, shard_batch(full, r, world)) for r in range(world))
        q.put((float((flat - ref).abs().max()), float(ref.abs().max()), float(flat_b.sum() - torch.cat([sd0[k].reshape(-1) for k in bufs]).sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_sum_equals_single_process_emulation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale, bdiff = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert err <= 1e-5 * scale, (err, scale)  # identical up to fp32 summation order
    assert abs(bdiff) < 1e-3  # rank 0's buffers were broadcast


def test_shard_batch_partitions_targets():
    from golden.cases import synth_batch
    from ultralytics.hip.dist import shard_batch
    full = synth_batch(3, 4, 5, 6)
    parts = [shard_batch(full, r, 2) for r in range(2)]
    assert sum(p["cls"].shape[0] for p in parts) == full["cls"].shape[0]
    assert all(p["img"].shape[0] == 2 and set(p["batch_idx"].tolist()) <= {0.0, 1.0} for p in parts)
    np.testing.assert_array_equal(parts[1]["img"].numpy(), full["img"][2:].numpy())
