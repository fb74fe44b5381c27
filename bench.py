#!/usr/bin/env python3
"""Throughput benchmark of the DEAL-YOLO training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full training iteration of DEAL-YOLO-N (yolov8n-ASF-P2P2) at 640x640, per-GPU batch 64 (BASELINE.json
configs[1]): image import, forward, detection loss (TAL + CIoU + DFL + BCE), hand-written backward, gradient all-reduce
(RCCL, N>1), SGD-nesterov + EMA -- all through libdealyolo_hip.so.  Inputs are synthetic and already resident in HBM.
Prints ONE JSON line on rank 0.  Extra objects:
  roofline     SURVEY.md 8(d) throughout: `achieved` / `frac` = ALGORITHMIC bytes per launch / average launch duration of the dominant
               kernel -- the kernel group with the largest total time per step among the launches that have algorithmic bytes
               (convolutions: forward, input gradient, weight gradient, stem) -- measured live with HIP events on the launch stream,
               against 8 TB/s HBM; `traffic` = its PMC HBM bytes per launch from profiles/; `step` = the whole step priced the same
               way (images/s x 367 MB / 8 TB/s; BatchNorm / activation passes count as fused away = 0 bytes);
               `dominant_launch` = the kernel group with the largest total time among ALL launches (a BatchNorm pass: 0 algorithmic
               bytes) priced with the bytes its launch cannot avoid (`own_frac`) -- what the top level carried in round 3;
  config.secondary  BASELINE.json configs[3] and configs[4] timed in the same run: the LD model's training step (images/s) and
               yolov8n-p2 fused inference at 1280x1280, batch 32 (get_FPS.py's protocol shortened: 20 + 100 forwards), each with its
               8(d) step fraction;
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
P2_1280_FWD_BYTES_PER_IMAGE = 528.0e6  # yolov8n-p2 forward at 1280x1280 (conv in + out at fp16)


def secondary_ld(dev, steps=10):
    """BASELINE.json configs[3]: yolov8n-LD-P2 (LDConv-heavy) training step, 640x640, batch 64 -- the same StepPlan protocol as the
    headline (hipGraph, loss-scale search settled before the timed steps)."""
    from ultralytics.hip.train import StepPlan
    from ultralytics.nn.tasks import DetectionModel
    torch.manual_seed(0)
    model = DetectionModel(os.path.join(os.path.dirname(CFG), "yolov8n-LD-P2.yaml"), verbose=False).to(dev).train()
    with torch.no_grad():  # zero-initialised p_conv weights (reference conv.py:357) would make every offset identical
        for n_, p_ in model.named_parameters():
            if n_.endswith("p_conv.weight"):
                p_.normal_(0, 0.02)
    for k, v in model.named_parameters():
        v.requires_grad = ".dfl" not in k
    plan = StepPlan(model, 64, 640, nmax=8, optimizer="SGD", use_graph=True)
    plan.crit.bbox_loss.use_wiseiou, plan.crit.bbox_loss.nwd_loss = True, True
    batch = {k: v.to(dev) for k, v in synth_batch(7, 64, 640, 6).items()}
    plan.img.copy_(batch["img"])
    batch["img"] = plan.img

    def one():
        plan.set_hyper([0.01] * 3, 0.937, [0.0, 0.0005, 0.0])
        plan.forward_backward(batch)
        plan.optimizer_step()

    for _ in range(3):
        one()
    still, last, n = 0, float(plan.state[6]), 0
    while still < 6 and n < 60:
        one()
        n += 1
        now = float(plan.state[6])
        still, last = (still + 1, last) if now == last else (0, now)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    plan.check_progress()
    rate = 64 / dt
    return {"workload": "yolov8n-LD-P2 train step 640x640 batch 64 (BASELINE.json configs[3]), wiou+nwd, hipGraph", "images_per_s": rate,
            "ms_per_step": dt * 1e3, "steps": steps, "algorithmic_bytes_per_image": ALG_BYTES_PER_IMAGE["yolov8n-LD-P2"],
            "step_frac": rate * ALG_BYTES_PER_IMAGE["yolov8n-LD-P2"] / 1e9 / HBM_PEAK_GBS}


def secondary_p2_1280(dev, warmup=20, iters=100):
    """BASELINE.json configs[4]: yolov8n-p2 (nc 80) fused inference at 1280x1280, batch 32, get_FPS.py's protocol (reference
    get_FPS.py:42-87: fuse, warm-up forwards, then forwards timed one by one with a synchronisation after each) shortened from
    200 + 1000 to 20 + 100 forwards; no NMS, as there."""
    from ultralytics import YOLO
    torch.manual_seed(0)
    model = YOLO("yolov8n-p2.yaml").model.to(dev).eval()
    model.fuse()
    x = torch.rand(32, 3, 1280, 1280, device=dev)
    ts = []
    with torch.no_grad():
        for _ in range(warmup):
            model(x)
        torch.cuda.synchronize()
        for _ in range(iters):
            t0 = time.perf_counter()
            model(x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    t = float(np.mean(ts))
    fps = 32 / t
    return {"workload": "yolov8n-p2 (nc 80) fused inference 1280x1280 batch 32 (BASELINE.json configs[4]), get_FPS.py protocol 20 + 100 forwards, no NMS",
            "fps": fps, "ms_per_forward": t * 1e3, "latency_ms_per_image": t / 32 * 1e3, "algorithmic_bytes_per_image": P2_1280_FWD_BYTES_PER_IMAGE,
            "step_frac": fps * P2_1280_FWD_BYTES_PER_IMAGE / 1e9 / HBM_PEAK_GBS}


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
    ap.add_argument("--secondary", type=int, default=1, help="also time BASELINE configs[3] (LD train) and configs[4] (p2 1280 inference)")
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

    roof = breakdown = None
    scalars_last, loss_scale, skipped_total = plan.crit.scalars.cpu(), float(plan.state[0]), float(plan.state[6])
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
            tc, ta = pr["top_conv"], pr["top_alg"]
            roof = {"bound": "hbm", "achieved": ta["alg_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ta["alg_gbs"] / HBM_PEAK_GBS,
                    "traffic": traffic(ta["kernel"]), "kernel": ta["kernel"], "avg_us": ta["us"], "bytes_per_launch": ta["algorithmic_bytes"],
                    "bytes_definition": "SURVEY.md 8(d) algorithmic bytes (layer input read once + output written once at fp16) per launch of the "
                                        "DOMINANT ALGORITHMIC KERNEL: the kernel group with the largest total time per step among the launches that "
                                        "have algorithmic bytes (convolution forward / input gradient / weight gradient / stem); BatchNorm and "
                                        "activation passes are priced at 0 by 8(d) -- the group that is largest among ALL launches is under "
                                        "dominant_launch with its own least traffic, the whole step under step",
                    "launches_per_step": ta["launches_per_step"], "ms_per_step": ta["ms_per_step"],
                    "step": {"achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS if step_gbs else None, "unit": "GB/s",
                             "algorithmic_bytes_per_image": alg, "device_ms_sum_of_launches": pr["step_device_ms"],
                             "traffic_bytes_per_step": tj.get("step_traffic_bytes")},
                    "dominant_launch": {"kernel": pr["kernel"], "avg_us": pr["us"], "own_bytes_per_launch": pr["bytes"], "own_achieved": pr["gbs"],
                                        "own_frac": pr["gbs"] / HBM_PEAK_GBS, "algorithmic_bytes_per_launch_8d": pr["algorithmic_bytes"],
                                        "traffic": traffic(pr["kernel"]), "launches_per_step": pr["launches_per_step"], "ms_per_step": pr["ms_per_step"],
                                        "own_bytes_definition": "each operand of the launch read once + each result written once at fp16"},
                    "top_conv": None if tc is None else {"kernel": tc["kernel"], "achieved": tc["gbs"], "frac": tc["gbs"] / HBM_PEAK_GBS,
                                                         "avg_us": tc["us"], "bytes_per_launch": tc["bytes"], "traffic": traffic(tc["kernel"]),
                                                         "launches_per_step": tc["launches_per_step"]},
                    "ranking": pr["ranking"]}
            breakdown = plan.breakdown()
    secondary = None
    if rank == 0 and world == 1 and a.secondary and a.model == "yolov8n-ASF-P2P2" and a.batch == 64 and a.imgsz == 640:
        secondary = {}  # (the headline plan's buffers stay resident: 288 GB hold all three workloads)
        for key, fn in (("ld_train", secondary_ld), ("p2_1280_infer", secondary_p2_1280)):
            try:
                secondary[key] = fn(dev)
            except Exception as e:  # a secondary figure must never cost the headline line
                secondary[key] = {"error": f"{type(e).__name__}: {e}"}
            print(f"[bench] secondary {key}: {secondary[key]}", file=sys.stderr, flush=True)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        cpu = cpu_baseline(a.imgsz)
    if rank == 0:
        s = scalars_last
        out = {"metric": "images/sec (train) DEAL-YOLO-N 640x640 bs=64/GPU", "value": value, "unit": "images/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": f"{'DEAL-YOLO-N' if a.model == 'yolov8n-ASF-P2P2' else a.model} ({a.model}.yaml) train step {a.imgsz}x{a.imgsz}, per-GPU batch {a.batch}, "
                                      f"fwd+TAL/{a.loss.upper()}/DFL/BCE loss+bwd+SGD+EMA, BASELINE.json configs[{3 if 'LD' in a.model else 1}]",
                          "loss_mode": a.loss, f"{other.replace('+', '_')}_images_per_s": other_rate,
                          "settle_steps": settle, "skipped_in_timed_steps": skipped_timed, "capture_retries": plan.capture_retries, "timed_regions_discarded": retimed,
                          "global_batch": a.batch * world, "parallelism": f"dp{world}", "hipgraph": bool(a.graph),
                          "loss_items_last": [float(x) for x in s[5:8]], "loss_scale": loss_scale,
                          "skipped_steps": skipped_total,
                          "device_ms_by_call": breakdown, "secondary": secondary},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # rank 0 may still be probing / printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
