"""-m gpu: whole DEAL-YOLO models through the traced/replayed StepPlan against the reference-generated fixtures
(tests/golden/models.npz): head outputs, loss items, per-parameter gradient norms, BN running statistics, and that a
replayed step reproduces the traced one bit for bit.  fp16 activations vs fp32 reference -> tolerances stated inline."""
import os

import numpy as np
import pytest
import torch

from conftest import CFG_DIR
from golden.cases import MODES
from gpu_util import l2err, relerr
from oracle import graph as og

pytestmark = pytest.mark.gpu
MODELS = ["yolov8n-ASF-P2P2", "yolov8n-LD-P2"]


def _build(name, mi):
    from ultralytics.nn.tasks import DetectionModel
    m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
    m.load_state_dict(og.fill_state(og.state_layout(g), 7 + mi), strict=True)
    return m.cuda().train(), g


@pytest.mark.parametrize("mi,name", list(enumerate(MODELS)))
@pytest.mark.parametrize("mode", list(MODES))
def test_model_step_vs_golden(golden, mi, name, mode):
    from ultralytics.hip.train import StepPlan
    G = golden("models")
    m, g = _build(name, mi)
    plan = StepPlan(m, 2, 64, nmax=8, init_scale=1024.0)
    plan.crit.bbox_loss.use_wiseiou, plan.crit.bbox_loss.nwd_loss = MODES[mode]
    batch = {k: G.t(f"{name}/{k}") for k in ("img", "batch_idx", "cls", "bboxes")}
    plan.forward_backward(batch)
    torch.cuda.synchronize()
    ho = plan.ho
    ld = "LD" in name  # LDConv floors its sampling coordinates: an fp16 rounding can move a sample to the next pixel
    for l, f in enumerate(ho.as_reference_list()):
        e, e2 = relerr(f.float(), G.t(f"{name}/feat{l}")), l2err(f.float(), G.t(f"{name}/feat{l}"))
        print(f"feat{l} relerr {e:.2e} l2err {e2:.2e}")
        assert (e2 < 5e-2) if ld else (e < 3e-2)
    s = plan.crit.scalars.cpu()
    ref_items = G.t(f"{name}/{mode}/items")
    per_item = float(((s[5:8] - ref_items).abs() / ref_items.abs()).max())
    print("items", s[5:8].tolist(), ref_items.tolist(), f"worst per-item relative error {per_item:.2e}")
    # fp16 activation storage against the fp32 reference at this 64x64 fixture (randomly filled weights and BN statistics): measured
    # 5.7e-3 on the box / dfl items of N, 5e-5 on LD; at 640x640 (tests/test_gpu_configs.py) 6e-5 .. 6e-4.  Tolerance = 2x measured.
    assert per_item < 1.2e-2
    assert abs(float(s[8]) - float(G[f"{name}/{mode}/loss"])) < 5e-3 * float(G[f"{name}/{mode}/loss"])
    names = list(G[f"{name}/{mode}/grad_names"])
    params = dict(m.named_parameters())
    scale = float(plan.state[0])
    l2 = torch.stack([params[k].grad.float().norm() / scale for k in names]).cpu()
    ref = G.t(f"{name}/{mode}/grad_l2")
    rel = ((l2 - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())).numpy()
    print("grad-l2 rel err: median %.2e max %.2e (%s)" % (np.median(rel), rel.max(), names[int(rel.argmax())]))
    # measured: N median 1.7e-3 / max 1.9e-2, LD median 4.2e-3 / max 8.5e-2 (p_conv.bias: gradients through floor()-ed coordinates)
    assert np.median(rel) < (1e-2 if ld else 4e-3) and rel.max() < (0.2 if ld else 4e-2)
    if mode == "ciou":
        first = params[names[0]].grad.float().cpu() / scale
        print("grad_first relerr %.3e l2err %.3e" % (relerr(first, G.t(f"{name}/{mode}/grad_first")), l2err(first, G.t(f"{name}/{mode}/grad_first"))))
        assert (l2err if ld else relerr)(first, G.t(f"{name}/{mode}/grad_first")) < (0.15 if ld else 5e-2)
        sd = m.state_dict()
        rm = list(G[f"{name}/run_mean_names"])
        assert relerr(torch.stack([sd[k].sum() for k in rm]).cpu(), G.t(f"{name}/run_mean_sum")) < 5e-3
        assert relerr(torch.stack([sd[k.replace("mean", "var")].sum() for k in rm]).cpu(), G.t(f"{name}/run_var_sum")) < 5e-3


def test_asf_p2_forward_vs_golden(golden):
    """ASF-P2 (Zoom_cat fusion, two ScalSeq+Add stages, 4 detection levels): train-mode head outputs."""
    from ultralytics.hip.train import StepPlan
    G = golden("models")
    name = "yolov8n-ASF-P2"
    m, g = _build(name, 2)
    plan = StepPlan(m, 2, 64, nmax=8, init_scale=1024.0)
    batch = {k: G.t(f"{name}/{k}") for k in ("img", "batch_idx", "cls", "bboxes")}
    plan.forward_backward(batch)
    torch.cuda.synchronize()
    for l, f in enumerate(plan.ho.as_reference_list()):
        e = l2err(f.float(), G.t(f"{name}/feat{l}"))
        print(f"ASF-P2 feat{l} l2err {e:.2e}")
        assert e < 3.5e-2  # fp16 storage noise through BatchNorm over 2 x 8 x 8 samples: 2.9e-2 / 3.06e-2 measured (imported / direct stem)
    assert torch.isfinite(plan.rt.flat_g).all()


def test_replay_is_bitwise_identical(golden):
    from ultralytics.hip.train import StepPlan
    G = golden("models")
    name = MODELS[0]
    m, g = _build(name, 0)
    plan = StepPlan(m, 2, 64, nmax=8, init_scale=1024.0)
    batch = {k: G.t(f"{name}/{k}") for k in ("img", "batch_idx", "cls", "bboxes")}
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    plan.forward_backward(batch)
    g1, s1 = plan.rt.flat_g.clone(), plan.crit.scalars.clone()
    m.load_state_dict(sd0)
    plan.forward_backward(batch)  # replayed launch list
    torch.cuda.synchronize()
    assert torch.equal(plan.crit.scalars[5:9], s1[5:9])
    assert torch.equal(plan.rt.flat_g, g1)


@pytest.mark.parametrize("variant", ["side_wgrad", "head_branches"])
def test_stream_variants_of_the_step_give_the_same_gradients(golden, variant, monkeypatch):
    """Two ways the recorded step can use more than one stream: weight gradients on a side stream (StepPlan(side_wgrad=True) /
    DY_SIDE_WGRAD=1: the head then takes its DENSE kernels -- decided before the forward is traced, so the loss zeroes what a dense
    backward reads) and Detect's levels as branches (DY_HEAD_STREAMS=1).  Same loss items; gradients to the arithmetic of the other
    kernel forms (rows vs dense sums, group vs single BatchNorm launches: 1e-3 of the largest gradient)."""
    from ultralytics.hip import engine as E
    from ultralytics.hip.train import StepPlan
    G = golden("models")
    name = MODELS[0]
    batch = {k: G.t(f"{name}/{k}") for k in ("img", "batch_idx", "cls", "bboxes")}
    m0, _ = _build(name, 0)
    base = StepPlan(m0, 2, 64, nmax=8, init_scale=1024.0)
    base.forward_backward(batch)
    g0, s0 = base.rt.flat_g.clone(), base.crit.scalars.clone()
    m1, _ = _build(name, 0)
    if variant == "head_branches":
        monkeypatch.setattr(E, "HEAD_STREAMS", True)
    plan = StepPlan(m1, 2, 64, nmax=8, init_scale=1024.0, side_wgrad=(variant == "side_wgrad"))
    plan.forward_backward(batch)
    if variant == "side_wgrad":
        assert any(o[3] == 1 for o in plan.rec_fb.ops if o[0] is not None), "no launch was recorded on the side stream"
        assert not any(o[2].startswith(("dy_conv1x1_rows", "dy_head_box_decode", "dy_cls_head")) for o in plan.rec_fb.ops)
    else:
        assert {o[3] for o in plan.rec_fb.ops if o[0] is not None} == {0, 1, 2}, "Detect's levels were not recorded as branches"
    plan.forward_backward(batch)  # and the replayed list (fork / join markers included) repeats it
    torch.cuda.synchronize()
    assert relerr(plan.crit.scalars[5:9], s0[5:9]) < 1e-5
    e = relerr(plan.rt.flat_g, g0)
    print(f"{variant}: gradients vs the default step {e:.2e}")
    assert torch.isfinite(plan.rt.flat_g).all() and e < 1e-3


@pytest.mark.parametrize("name", [MODELS[0], "yolov8n-LD-P2"])
def test_cv1_halves_as_planes_are_the_same_step_bit_for_bit(golden, name, monkeypatch):
    """C2f.cv1's two halves as tensors of their own (Engine.new_planes, DY_PLANAR_CV1; reference nn/modules/block.py:223) against the
    2c-wide tensor with a sliced second half: the apply / reduce / weight-gradient kernels compute every element and every sum in
    the same order from another base pointer, so loss items and ALL gradients must be EQUAL -- and the two-plane launches must be
    the ones recorded."""
    from ultralytics.hip import engine as E
    from ultralytics.hip.train import StepPlan
    G = golden("models")
    batch = {k: G.t(f"{name}/{k}") for k in ("img", "batch_idx", "cls", "bboxes")}
    outs = []
    for on in (False, True):
        monkeypatch.setattr(E, "PLANAR_CV1", on)
        m, _ = _build(name, MODELS.index(name))
        plan = StepPlan(m, 2, 64, nmax=8, init_scale=1024.0)
        plan.forward_backward(batch)
        plan.forward_backward(batch)
        torch.cuda.synchronize()
        names = [o[2] for o in plan.rec_fb.ops if o[0] is not None]
        n_c2f = sum(1 for mod in m.modules() if type(mod).__name__ == "C2f")
        for k in ("dy_bn_act_apply_acc_split", "dy_bn_act_bwd_reduce_acc_split", "dy_conv1x1_wgrad_bn_planes"):
            assert names.count(k) == (n_c2f if on else 0), (k, names.count(k), n_c2f)
        outs.append((plan.rt.flat_g.clone(), plan.crit.scalars.clone()))
    assert torch.isfinite(outs[0][0]).all() and float(outs[0][0].abs().max()) > 0
    if "LD" in name:  # LDConv's far-sample scatter adds with fp32 atomics: the order of those sums differs from run to run
        assert relerr(outs[0][1][5:9], outs[1][1][5:9]) < 1e-6 and relerr(outs[0][0], outs[1][0]) < 1e-4
        return
    assert torch.equal(outs[0][1][5:9], outs[1][1][5:9])
    assert torch.equal(outs[0][0], outs[1][0]), f"max diff {float((outs[0][0] - outs[1][0]).abs().max()):.3e}"


@pytest.mark.parametrize("name", MODELS)
def test_upsample_left_to_the_concat_consumer_is_the_same_step_bit_for_bit(golden, name, monkeypatch):
    """nn.Upsample in front of a Concat never executed in TRAINING either (engine.UpAct, DY_UPSEG_TRAIN): the segmented 1x1 conv and its
    weight-gradient kernel read the low-resolution tensor at (y >> 1, x >> 1), the member's input gradient lands in a full-resolution
    gradient tensor that Upsample's backward folds as before -- the same values in the same order: loss items and ALL gradients EQUAL."""
    from ultralytics.hip import engine as E
    from ultralytics.hip.train import StepPlan
    G = golden("models")
    batch = {k: G.t(f"{name}/{k}") for k in ("img", "batch_idx", "cls", "bboxes")}
    outs = []
    for on in (False, True):
        monkeypatch.setattr(E, "UPSEG_TRAIN", on)
        m, _ = _build(name, MODELS.index(name))
        plan = StepPlan(m, 2, 64, nmax=8, init_scale=1024.0)
        plan.forward_backward(batch)
        plan.forward_backward(batch)
        torch.cuda.synchronize()
        ups = [o for o in plan.rec_fb.ops if o[0] is not None and o[2] == "dy_upsample2x"]
        assert len(ups) == (2 if on else 4), len(ups)  # two Upsample layers: forward + backward launches, or the backward ones alone
        outs.append((plan.rt.flat_g.clone(), plan.crit.scalars.clone()))
    assert torch.isfinite(outs[0][0]).all() and float(outs[0][0].abs().max()) > 0
    if "LD" in name:
        assert relerr(outs[0][1][5:9], outs[1][1][5:9]) < 1e-6 and relerr(outs[0][0], outs[1][0]) < 1e-4
        return
    assert torch.equal(outs[0][1][5:9], outs[1][1][5:9])
    assert torch.equal(outs[0][0], outs[1][0]), f"max diff {float((outs[0][0] - outs[1][0]).abs().max()):.3e}"


@pytest.mark.parametrize("name", [MODELS[0], "yolov8n-p2"])
def test_per_layer_capture_does_not_touch_the_gradients(golden, name):
    """``model._capture = []`` (the per-layer outputs some parity tests read) takes inspection copies of concatenations / lazily
    up-sampled tensors (Engine.snapshot): they never join the backward pass, so the step's gradients are the bits of the step without
    a capture.  (A copy made through Engine.dense registered a closure that handed an unwritten gradient on.)"""
    from golden.cases import synth_batch
    from ultralytics.hip.train import StepPlan
    from ultralytics.nn.tasks import DetectionModel
    p = os.path.join(CFG_DIR, name + ".yaml")
    g = og.build_graph(og.load_yaml(p))
    batch = synth_batch(5, 2, 4, g.nc)
    outs = []
    for cap in (False, True):
        m = DetectionModel(p, ch=3, verbose=False)
        m.load_state_dict(og.fill_state(og.state_layout(g), 10), strict=True)
        m.cuda().train()
        plan = StepPlan(m, 2, 64, nmax=8, init_scale=1.0)
        if cap:
            m._capture = []
        plan.forward_backward(batch)
        torch.cuda.synchronize()
        if cap:
            assert len(m._capture) == len(m.model)
            m._capture = None
        outs.append(plan.rt.flat_g.clone())
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0], outs[1]), f"max diff {float((outs[0] - outs[1]).abs().max()):.3e}"


def test_optimizer_trace_vs_golden(golden):
    """5 SGD-nesterov steps (warm-up lr/momentum, clip 10, EMA) against the reference's own optimizer_step trace."""
    from golden.cases import synth_batch
    from ultralytics.hip.train import StepPlan
    G = golden("trainer")
    name = "yolov8n-ASF-P2P2"
    from ultralytics.nn.tasks import DetectionModel
    m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
    m.load_state_dict(og.fill_state(og.state_layout(g), 11), strict=True)
    m.cuda().train()
    for k, v in m.named_parameters():
        v.requires_grad = ".dfl" not in k
    plan = StepPlan(m, 2, 64, nmax=8, init_scale=1024.0)
    ref = G["SGD/trace"]
    nw, wd = 100, 0.0005 * 2 * 32 / 64
    for ni in range(5):
        batch = synth_batch(900 + ni, 2, 4, g.nc)
        lr = [float(np.interp(ni, [0, nw], [0.1 if j == 0 else 0.0, 0.01])) for j in range(3)]
        mom = float(np.interp(ni, [0, nw], [0.8, 0.937]))
        row = ref[ni]
        plan.set_hyper(lr, mom, [0.0, wd, 0.0])
        plan.forward_backward(batch)
        plan.accumulate()  # the reference steps only when ni - last_opt_step >= accumulate (warm-up ramps it 1 -> 32)
        if row[5]:
            plan.optimizer_step()
        loss, items = plan.loss_items()
        print(ni, loss, row[0], float(plan.state[3]), row[4], row[5])
        assert abs(loss - row[0]) < 1e-2 * row[0]
        if row[5]:
            # golden grad_norm is that of the accumulated gradient at the time of the step
            assert abs(float(plan.state[3]) - row[4]) < 5e-2 * row[4], "grad norm"
        fl = torch.cat([plan.rt.flat_p, plan.rt.flat_b]).double().abs().sum()
        assert abs(float(fl) - row[10]) < 2e-4 * row[10], "abs-sum of the state after the step"
    assert relerr(m.state_dict()["model.0.conv.weight"], G.t("SGD/final_w0")) < 2e-2


def test_reference_checkpoint_runs_like_a_native_model():
    """A model rebuilt from the reference-format checkpoint (tests/golden/ref_ckpt.pt) computes exactly what a natively built
    model with the same (fp16-rounded) state computes."""
    import os
    from conftest import ROOT
    from ultralytics.nn.tasks import DetectionModel, attempt_load_weights
    a = attempt_load_weights(os.path.join(ROOT, "tests", "golden", "ref_ckpt.pt"), device="cuda:0")
    b = DetectionModel(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml"), ch=3, verbose=False)
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml")))
    b.load_state_dict({k: (v.half().float() if v.is_floating_point() else v) for k, v in og.fill_state(og.state_layout(g), 21).items()})
    b.cuda().eval()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(4)).cuda()
    ya, _ = a(x)
    yb, _ = b(x)
    assert torch.equal(ya, yb)


def test_steps_queued_behind_a_busy_device_keep_their_own_scalars():
    """The host can enqueue several steps while the device is still busy.  Per-step scalars (target count, hyper-parameters)
    therefore travel by value, never through a reused pinned word: two different batches queued back to back behind a parked
    device must accumulate exactly the gradients they produce one at a time."""
    from ultralytics.hip.train import StepPlan
    from ultralytics.nn.tasks import DetectionModel
    g = torch.Generator().manual_seed(12)
    B, S = 2, 128

    def batch(n_per):
        n = B * n_per
        return dict(img=torch.rand(B, 3, S, S, generator=g).cuda(), batch_idx=torch.arange(B).repeat_interleave(n_per).float().cuda(),
                    cls=torch.randint(0, 6, (n, 1), generator=g).float().cuda(),
                    bboxes=torch.cat([torch.rand(n, 2, generator=g) * 0.6 + 0.2, torch.rand(n, 2, generator=g) * 0.2 + 0.05], 1).cuda())

    a, b = batch(2), batch(7)
    torch.manual_seed(0)
    m = DetectionModel("yolov8n-ASF-P2P2.yaml", verbose=False).cuda().train()
    plan = StepPlan(m, B, S, nmax=8, optimizer="SGD", use_graph=True, init_scale=1.0)
    plan.set_hyper([0.0] * 3, 0.9, [0.0] * 3)
    ref = None
    for bt in (a, b):  # one at a time, synchronised
        plan.forward_backward(bt)
        torch.cuda.synchronize()
        ref = plan.rt.flat_g.clone() if ref is None else ref + plan.rt.flat_g
    torch.cuda._sleep(400_000_000)  # park the device: everything below is enqueued before any of it runs
    plan.forward_backward(a)
    plan.accumulate()
    plan.forward_backward(b)
    plan.accumulate()
    torch.cuda.synchronize()
    assert torch.equal(plan.gsum[:plan.rt.n_params_flat], ref)


@pytest.mark.parametrize("mi,name", list(enumerate(MODELS)))
def test_layer_outputs_vs_the_fp16_storage_oracle(golden, mi, name):
    """Separates storage rounding from kernel error.  The engine keeps activations in fp16; against the fp32 reference that alone
    costs ~1e-3 per layer and 1e-2 at the head, which is what the model-level tolerances above absorb.  Here the oracle rounds to
    fp16 at exactly the points where the engine stores (oracle.nn.STORAGE_FP16: raw conv output, activation, sums / pools; batch
    statistics from the un-rounded conv output), so what is left is the kernels' own arithmetic: every layer of the training-mode
    forward must agree to a few 1e-4 (relative L2) -- a regression of 1e-3 in any kernel fails here although it would pass the
    comparisons with the fp32 goldens."""
    import oracle.nn as onn
    from gpu_util import l2err
    from ultralytics.hip.runtime import Runtime
    G = golden("models")
    m, g = _build(name, mi)
    img = G.t(f"{name}/img")
    sd = og.fill_state(og.state_layout(g), 7 + mi)
    sd = {k: (v.half().float() if v.dim() >= 4 else v) for k, v in sd.items()}  # conv weights reach the MFMAs as fp16 packs
    rt = m._runtime(torch.device("cuda", 0))
    rt.eng.training = True
    m._capture = []
    with torch.no_grad():
        rt.ensure_packed()
        m.forward_act(rt.to_act(img.cuda()))
    torch.cuda.synchronize()
    acts, m._capture = m._capture, None
    outs = [Runtime.to_tensor(a).float().cpu() if not isinstance(a, (list, tuple)) and hasattr(a, "st") else None for a in acts]
    ld = "LD" in name
    worst = 0.0
    onn.STORAGE_FP16 = True
    try:
        for i, L in enumerate(g.layers[:-1]):  # all but Detect (its fp32 maps are compared by the golden tests)
            # teacher forcing: the oracle layer gets the ENGINE's inputs, so one layer's arithmetic is compared at a time and
            # fp16 rounding decisions that flip in an earlier layer do not pile up
            at = lambda j: outs[i + j] if j < 0 else outs[j]  # noqa: E731  (negative 'from' indices are relative to this layer)
            src = img.half().float() if i == 0 else (at(L.f) if isinstance(L.f, int) else [at(j) for j in L.f])
            with torch.no_grad():
                r = onn.apply_layer(L, {k: v.clone() for k, v in sd.items()}, src, True, g.strides)
            e = l2err(outs[i], r)
            fp32 = l2err(outs[i], G.t(f"{name}/layer{i}")) if f"{name}/layer{i}" in G else float("nan")
            worst = max(worst, e)
            print(f"{name} layer {i:2d} {L.kind:12s} vs fp16-storage oracle on the same input {e:.2e}   vs fp32 reference (whole chain) {fp32:.2e}")
            # measured: 1e-7 .. 5e-5 for single convolutions / LDConv / SPPF / ScalSeq, 2e-5 .. 4.4e-4 for C2f (a chain of up to six
            # convolutions whose intermediate fp16 roundings can flip); copies (Upsample, Concat, Add of two maps) exactly 0
            assert e < 1e-3, f"layer {i} ({L.kind})"
    finally:
        onn.STORAGE_FP16 = False
    print(f"{name}: worst layer {worst:.2e}")
