#!/usr/bin/env python3
"""Throughput benchmark of the DEAL-YOLO training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full training iteration of DEAL-YOLO-N (yolov8n-ASF-P2P2) at 640x640, per-GPU batch 64 (BASELINE.json
configs[1]): image import, forward, detection loss (TAL + CIoU + DFL + BCE), hand-written backward, gradient all-reduce
(RCCL, N>1), SGD-nesterov + EMA -- all through libdealyolo_hip.so.  Inputs are synthetic and already resident in HBM.
Prints ONE JSON line on rank 0.  Extra objects:
  roofline     the kernel with the largest total time per step among ALL launches (whatever it is): the bytes its launch cannot
               avoid (operands read once + results written once at fp16; for a convolution that IS SURVEY.md 8(d)'s algorithmic
               bytes) / its average launch duration, measured live with HIP events on the launch stream, against 8 TB/s HBM;
               `traffic` = its PMC HBM bytes per launch from profiles/; `step` = the whole step priced by 8(d) alone, where
               BatchNorm / activation passes count as fused away = 0 bytes (images/s x 367 MB / 8 TB/s);
               `top_conv` = the most expensive convolution instantiation, for comparison with earlier rounds;
  cpu_baseline the CPU oracle (a port of the reference's PyTorch CPU path) timed on the host cores in BASELINE.md section 2's protocol
               (bs=2, median of 8 steps after 2 warm-up); `bs16` = the same at bs=16 (about 25 s of CPU work: the bounded sample).
--loss ciou|wiou|ciou+nwd|wiou+nwd selects the box loss: the default is the north-star's wiou+nwd; the reference-default ciou rate is
measured beside it (config.ciou_images_per_s: the same launch list replayed eagerly with the other loss mode).
--gpus N from a plain `python bench.py` start re-launches itself under torch.distributed.run (one rank per GPU) as a CHILD process
before anything touches the GPU, like YOLO.train(device='0,1,..') does (reference engine/trainer.py:607-627, utils/dist.py:47-65).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "experiment-yolo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ultralytics.hip  # noqa: E402,F401  (first: sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before anything initialises HIP)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG = os.path.join(ROOT, "experiment-yolo_amd", "ultralytics", "cfg", "models", "yolov8n-ASF-P2P2.yaml")
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per image of one TRAINING step (3 x forward conv in+out at fp16), by model YAML stem
ALG_BYTES_PER_IMAGE = {"yolov8n-ASF-P2P2": 367.1e6, "yolov8n-LD-P2": 407.8e6, "yolov8n-ASF-P2": 392.1e6}


def synth_batch(seed, B, imgsz, nc, n_per=8):
    """SURVEY.md 8(d) recipe: U[0,1) image, 8 boxes/img, wh in [0.01,0.09), xy in [0.1,0.9), sorted batch_idx."""
    rng = np.random.default_rng(seed)
    n = B * n_per
    return dict(img=torch.from_numpy(rng.random((B, 3, imgsz, imgsz), dtype=np.float32)),
                batch_idx=torch.arange(B).repeat_interleave(n_per).float(),
                cls=torch.from_numpy(rng.integers(0, nc, (n, 1)).astype(np.float32)),
                bboxes=torch.from_numpy(np.concatenate([rng.random((n, 2)) * 0.8 + 0.1, rng.random((n, 2)) * 0.08 + 0.01], 1).astype(np.float32)))


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota (the GPU box gives 16 per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max") and txt[0] != "max":
                n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            elif path.endswith("quota_us") and int(txt[0]) > 0:
                n = min(n, max(1, int(int(txt[0]) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
        except Exception:
            pass
    return max(1, min(n, 32))


def cpu_baseline(imgsz):
    """CPU oracle (port of the reference PyTorch CPU trainer step: forward + loss + autograd backward + SGD + EMA).
    `value`: BASELINE.md section 2's protocol (bs=2, median of 8 steps after 2 warm-up steps); `bs16`: 10 steps of bs=16."""
    from oracle import graph as og, trainer as otr
    g = og.build_graph(og.load_yaml(CFG))
    torch.set_num_threads(host_cores())

    def run(bs, steps, warm, budget):
        ts = otr.TrainState(g, og.default_init_state(g, 0), otr.Hyp(), batch_size=bs, nb=8)
        t = []
        for i in range(warm + steps):
            batch = synth_batch(100 + i, bs, imgsz, g.nc)
            t0 = time.perf_counter()
            otr.train_step(ts, batch)
            t.append(time.perf_counter() - t0)
            print(f"[bench] cpu_baseline bs={bs} step {i + 1}/{warm + steps}: {t[-1]:.2f} s ({torch.get_num_threads()} threads)", file=sys.stderr, flush=True)
            if sum(t) > budget:  # bounded sample: never let the reported baseline stall the run
                break
        return bs / float(np.median(t[warm:] if len(t) > warm else t))

    v2 = run(2, 8, 2, 60)
    v16 = run(16, 10, 2, 90)
    return {"value": v2, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"8 steps of bs=2 {imgsz}x{imgsz} after 2 warm-up (median), fp32, oracle/trainer.py -- BASELINE.md section 2's protocol; "
                      f"bs16 = 10 steps of bs=16 after 2 warm-up (about 25 s of CPU work)",
            "bs16": v16}


def relaunch(a):
    """--gpus N without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child
    process -- nothing in this process has touched the GPU yet -- and relay its output (rank 0 prints the JSON line)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(p.stdout)
    sys.stdout.flush()
    raise SystemExit(p.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--graph", type=int, default=1, help="capture the step into hipGraphs")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--probe", type=int, default=1, help="time the dominant kernel with HIP events")
    ap.add_argument("--model", default="yolov8n-ASF-P2P2", help="model YAML stem (yolov8n-LD-P2 = BASELINE.json configs[3])")
    ap.add_argument("--lr", type=float, default=0.01, help="learning rate of the three parameter groups (timing does not depend on it; "
                    "tiny test configurations pass a smaller one so that no step overflows)")
    ap.add_argument("--loss", default="wiou+nwd", choices=["ciou", "wiou", "ciou+nwd", "wiou+nwd"],
                    help="box loss mode (north-star: wiou+nwd; the reference's cfg default is ciou)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("DY_REHEARSE_ON_ONE_GPU") == "1"  # N ranks on one GPU over gloo: exercises the N>1 code path only
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ultralytics.hip.train import StepPlan
    from ultralytics.nn.tasks import DetectionModel

    torch.manual_seed(0)
    cfg_path = CFG if a.model == "yolov8n-ASF-P2P2" else os.path.join(os.path.dirname(CFG), a.model + ".yaml")
    model = DetectionModel(cfg_path, verbose=False).to(dev).train()
    if "LD" in a.model:  # zero-initialised p_conv weights (reference conv.py:357) would make every offset identical
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith("p_conv.weight"):
                    p_.normal_(0, 0.02)
    for k, v in model.named_parameters():
        v.requires_grad = ".dfl" not in k
    plan = StepPlan(model, a.batch, a.imgsz, nmax=8, optimizer="SGD", world_size=world, use_graph=bool(a.graph))
    plan.crit.bbox_loss.use_wiseiou, plan.crit.bbox_loss.nwd_loss = a.loss.startswith("wiou"), a.loss.endswith("nwd")
    batch = {k: v.to(dev) for k, v in synth_batch(1 + rank, a.batch, a.imgsz, 6).items()}
    # the synthetic images live in the plan's static input buffer (what a loader's H2D copy would target): the step then starts
    # at the import kernel instead of with a 315 MB device-to-device staging copy
    plan.img.copy_(batch["img"])
    batch["img"] = plan.img
    lr, mom, wd = [a.lr] * 3, 0.937, [0.0, 0.0005 * a.batch * world / 64 if a.batch * world < 64 else 0.0005, 0.0]

    def one_step():
        plan.set_hyper(lr, mom, wd)
        plan.forward_backward(batch)
        if world > 1:
            plan.all_reduce()
        plan.optimizer_step()

    for _ in range(a.warmup):
        one_step()
    # The dynamic loss scale starts at 65536 (GradScaler's policy) and halves on every overflow: until the search ends some
    # optimizer steps are skipped ones.  Keep stepping (untimed) until the skip counter has stood still for eight steps, so that
    # every TIMED step is a full one whatever --warmup was.
    settle, still, last = 0, 0, float(plan.state[6])
    while still < 8 and settle < 100:
        one_step()
        settle += 1
        now = float(plan.state[6])
        still, last = (still + 1, last) if now == last else (0, now)
    if world > 1:  # every rank leaves the settling loop after the same number of steps (all-reduced gradients overflow together)
        dist.barrier()

    def timed_region():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # A timed region in which the scale search skipped an optimizer step anyway (the fixed synthetic batch is being over-fitted at
    # the reference's lr0, gradient spikes do happen) is thrown away and timed again, so that the line below never holds a skipped step.
    retimed = 0
    while True:
        skipped_before = float(plan.state[6])
        dt = timed_region()
        if float(plan.state[6]) == skipped_before or retimed == 3:  # state is all-reduced with the gradients: the same on every rank
            break
        retimed += 1
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt
    plan.check_progress()  # a step whose optimizer launches do not take effect (counter stuck, everything skipped) is not a number
    skipped_timed = float(plan.state[6]) - skipped_before

    # the reference-default CIoU loss beside the north-star's WIoU+NWD (or the other way round): same launch list, replayed
    # eagerly (the loss mode is a by-value launch argument, so it cannot change inside the captured graph)
    other = "ciou" if a.loss != "ciou" else "wiou+nwd"
    other_rate = None
    if world == 1:
        g_fb, plan.graph_fb = plan.graph_fb, None
        try:
            plan.crit.bbox_loss.use_wiseiou, plan.crit.bbox_loss.nwd_loss = other.startswith("wiou"), other.endswith("nwd")
            for _ in range(3):
                one_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                one_step()
            torch.cuda.synchronize()
            other_rate = a.batch * a.steps / (time.perf_counter() - t1)
        finally:
            plan.crit.bbox_loss.use_wiseiou, plan.crit.bbox_loss.nwd_loss = a.loss.startswith("wiou"), a.loss.endswith("nwd")
            plan.crit.sync_modes()
            plan.graph_fb = g_fb

    roof = None
    if rank == 0 and a.probe:
        pr = plan.probe_dominant_kernel(batch, reps=max(5, min(a.steps, 20)))
        if pr:
            try:  # HBM bytes per launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed under profiles/
                tj = json.load(open(os.path.join(ROOT, "profiles", "kernel_traffic.json")))
            except Exception:
                tj = {}

            def traffic(k):
                by = tj.get("traffic_bytes_by_kernel", {})
                return by.get(k, next((v for kk, v in by.items() if kk.startswith(k.split(" ")[0])), None)) if k else None

            alg = ALG_BYTES_PER_IMAGE.get(a.model)
            step_gbs = value * alg / 1e9 if alg else None
            tc = pr["top_conv"]
            roof = {"bound": "hbm", "achieved": pr["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pr["gbs"] / HBM_PEAK_GBS,
                    "traffic": traffic(pr["kernel"]), "kernel": pr["kernel"], "avg_us": pr["us"], "bytes_per_launch": pr["bytes"],
                    "bytes_definition": "each operand of the launch read once + each result written once at fp16 (for a convolution this IS "
                                        "SURVEY.md 8(d)'s algorithmic bytes; 8(d) prices a BatchNorm/activation pass at 0 -- the step figure below does)",
                    "algorithmic_bytes_per_launch_8d": pr["algorithmic_bytes"],
                    "launches_per_step": pr["launches_per_step"], "ms_per_step": pr["ms_per_step"],
                    "step": {"achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS if step_gbs else None, "unit": "GB/s",
                             "algorithmic_bytes_per_image": alg, "device_ms_sum_of_launches": pr["step_device_ms"],
                             "traffic_bytes_per_step": tj.get("step_traffic_bytes")},
                    "top_conv": None if tc is None else {"kernel": tc["kernel"], "achieved": tc["gbs"], "frac": tc["gbs"] / HBM_PEAK_GBS,
                                                         "avg_us": tc["us"], "bytes_per_launch": tc["bytes"], "traffic": traffic(tc["kernel"]),
                                                         "launches_per_step": tc["launches_per_step"]},
                    "ranking": pr["ranking"]}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        cpu = cpu_baseline(a.imgsz)
    if rank == 0:
        s = plan.crit.scalars.cpu()
        out = {"metric": "images/sec (train) DEAL-YOLO-N 640x640 bs=64/GPU", "value": value, "unit": "images/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": f"{'DEAL-YOLO-N' if a.model == 'yolov8n-ASF-P2P2' else a.model} ({a.model}.yaml) train step {a.imgsz}x{a.imgsz}, per-GPU batch {a.batch}, "
                                      f"fwd+TAL/{a.loss.upper()}/DFL/BCE loss+bwd+SGD+EMA, BASELINE.json configs[{3 if 'LD' in a.model else 1}]",
                          "loss_mode": a.loss, f"{other.replace('+', '_')}_images_per_s": other_rate,
                          "settle_steps": settle, "skipped_in_timed_steps": skipped_timed, "timed_regions_discarded": retimed,
                          "global_batch": a.batch * world, "parallelism": f"dp{world}", "hipgraph": bool(a.graph),
                          "loss_items_last": [float(x) for x in s[5:8]], "loss_scale": float(plan.state[0]),
                          "skipped_steps": float(plan.state[6]),
                          "device_ms_by_call": plan.breakdown() if roof else None},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # rank 0 may still be probing / printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
