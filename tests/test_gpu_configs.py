"""-m gpu: the BASELINE.json configurations at their real sizes, against reference-generated fixtures.

configs[1] / configs[3]: DEAL-YOLO-N and the LD variant at 640x640 -- tests/golden/fullsize.npz holds the reference's step-0
loss in the four loss modes and its per-parameter gradient norms for a batch of 2; the batch of 64 is 32 copies of that batch
(every BatchNorm sees the same statistics, every normalised loss term the same value), so the same fixture pins the bs-64
step.  configs[4]: yolov8n-p2 (nc = 80, four levels) -- models.npz layer outputs at 64x64, and the 1280x1280 batch-32 fused
forward with the validator's NMS settings, where more than max_nms = 30000 candidates per image reach the pre-sort."""
import os

import numpy as np
import pytest
import torch

from conftest import CFG_DIR
from golden.cases import MODES
from gpu_util import l2err, relerr
from oracle import graph as og
from oracle import nms as onms

pytestmark = pytest.mark.gpu


def _model(name, seed):
    from ultralytics.nn.tasks import DetectionModel
    m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
    m.load_state_dict(og.fill_state(og.state_layout(g), seed), strict=True)
    return m, g


# ---- configs[1], configs[3]: 640x640 training step ---------------------------------------------------------------------------
def _fullsize_batch(G, name, mi, copies):
    img = torch.from_numpy(np.random.default_rng(5 + mi).random((2, 3, 640, 640), dtype=np.float32))
    b = {k: G.t(f"{name}/{k}") for k in ("batch_idx", "cls", "bboxes")}
    if copies > 1:
        b = dict(batch_idx=torch.cat([b["batch_idx"] + 2 * c for c in range(copies)]), cls=b["cls"].repeat(copies, 1),
                 bboxes=b["bboxes"].repeat(copies, 1))
        img = img.repeat(copies, 1, 1, 1)
    return dict(img=img, **b)


@pytest.mark.parametrize("mi,name", [(0, "yolov8n-ASF-P2P2"), (1, "yolov8n-LD-P2")])
@pytest.mark.parametrize("copies", [1, 32], ids=["bs2", "bs64"])
def test_fullsize_640_step_vs_reference(golden, mi, name, copies):
    from ultralytics.hip.train import StepPlan
    G = golden("fullsize")
    ld = "LD" in name
    B = 2 * copies
    batch = _fullsize_batch(G, name, mi, copies)
    for mode, (wiou, nwd) in MODES.items():
        m, g = _model(name, 21 + mi)
        m.cuda().train()
        plan = StepPlan(m, B, 640, nmax=8, init_scale=64.0)
        plan.crit.bbox_loss.use_wiseiou, plan.crit.bbox_loss.nwd_loss = wiou, nwd
        plan.forward_backward(batch)
        loss, items = plan.loss_items()
        ref_items, ref_loss = G.t(f"{name}/{mode}/items"), float(G[f"{name}/{mode}/loss"]) * copies  # loss = sum(items) * B
        e_items, e_loss = relerr(items, ref_items), abs(loss - ref_loss) / ref_loss
        print(f"{name} bs{B} {mode}: items {items.tolist()} ref {ref_items.tolist()} relerr {e_items:.2e} loss relerr {e_loss:.2e}")
        # fp16 activation storage against the fp32 reference.  Measured (round 3): N 6.5e-4 at batch 2, 8.3e-6 at batch 64; LD 5.7e-4 /
        # 6.1e-5.  Bound: the north-star's 1e-3 for both models
        assert e_items < 1e-3 and e_loss < 1e-3
        if mode == "ciou":
            assert float(plan.state[2]) == 0.0 and torch.isfinite(plan.rt.flat_g).all()
        if mode == "ciou" and not ld:  # LD gradients: test_fullsize_640_ld_gradients_in_the_init_regime (see there)
            names = [str(k) for k in G[f"{name}/grad_names"]]
            params = dict(m.named_parameters())
            l2 = torch.stack([params[k].grad.float().norm() for k in names]).cpu() / (float(plan.state[0]) * copies)
            ref = G.t(f"{name}/grad_l2")
            rel = ((l2 - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())).numpy()
            print(f"  grad-l2 rel err: median {np.median(rel):.2e} max {rel.max():.2e} ({names[int(rel.argmax())]})")
            assert np.median(rel) < 5e-3 and rel.max() < 0.12  # measured 8e-4 / 7.8e-2
        del plan, m
        torch.cuda.empty_cache()


@pytest.mark.parametrize("copies", [1, 32], ids=["bs2", "bs64"])
def test_fullsize_640_ld_gradients_in_the_init_regime(golden, copies):
    """LD variant, 640x640, p_conv.weight = 0 as LDConv.__init__ leaves it (|offset| < 1): fullsize_ld0.npz.  With the random
    p_conv weights of fullsize.npz the offsets reach tens of pixels and the REFERENCE's gradients are not a stable function of
    its inputs -- perturbing the image by 1e-4 relative moves the reference's own per-parameter gradient norms by a median of
    34 % (max 150 %), by 0.13 % (max 3.5 %) in this regime (oracle, fp32, CPU) -- so that fixture pins the LD losses and this
    one the LD gradients."""
    from ultralytics.hip.train import StepPlan
    G = golden("fullsize_ld0")
    name, mi = "yolov8n-LD-P2", 1
    from ultralytics.nn.tasks import DetectionModel
    m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
    sd = og.fill_state(og.state_layout(g), 21 + mi)
    for k in sd:
        if k.endswith("p_conv.weight"):
            sd[k] = torch.zeros_like(sd[k])
        elif k.endswith("p_conv.bias"):
            sd[k] = sd[k].clamp(-0.9, 0.9)
    m.load_state_dict(sd, strict=True)
    m.cuda().train()
    B = 2 * copies
    plan = StepPlan(m, B, 640, nmax=8, init_scale=64.0)
    plan.forward_backward(_fullsize_batch(G, name, mi, copies))
    loss, items = plan.loss_items()
    e = relerr(items, G.t(f"{name}/ciou/items"))
    print(f"LD init regime bs{B}: items {items.tolist()} ref {G.t(f'{name}/ciou/items').tolist()} relerr {e:.2e}")
    assert e < 2e-4 and float(plan.state[2]) == 0.0  # measured 9.6e-5 (batch 2), 3.0e-5 (batch 64)
    names = [str(k) for k in G[f"{name}/grad_names"]]
    params = dict(m.named_parameters())
    l2 = torch.stack([params[k].grad.float().norm() for k in names]).cpu() / (float(plan.state[0]) * copies)
    ref = G.t(f"{name}/grad_l2")
    rel = ((l2 - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())).numpy()
    print(f"  grad-l2 rel err: median {np.median(rel):.2e} max {rel.max():.2e} ({names[int(rel.argmax())]})")
    assert np.median(rel) < 3e-3 and rel.max() < 6e-2  # measured 1.0e-3 / 2.9e-2


# ---- configs[4]: yolov8n-p2 -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mi,name", [(3, "yolov8n-p2"), (2, "yolov8n-ASF-P2")])
def test_p2_models_vs_golden(golden, mi, name):
    """Four-level models at the fixture size: train-mode head maps, eval-mode decode, and the fused (BN-folded) eval output."""
    from ultralytics.hip.train import StepPlan
    G = golden("models")
    m, g = _model(name, 7 + mi)
    m.cuda().train()
    plan = StepPlan(m, 2, 64, nmax=8, init_scale=1.0)
    m._capture = []
    plan.forward_backward({k: G.t(f"{name}/{k}") for k in ("img", "batch_idx", "cls", "bboxes")})
    torch.cuda.synchronize()
    acts, m._capture = m._capture, None
    print(f"{name}: non-finite gradient words {int((~torch.isfinite(plan.rt.flat_g)).sum())}, loss items {plan.crit.scalars[5:9].tolist()}")
    feats = [f.float() for f in plan.ho.as_reference_list()]
    # (1) the kernels: Detect in the fp16-storage oracle (oracle.nn.STORAGE_FP16 rounds where the engine stores) fed with the ENGINE's
    # own input maps -- what is left is the head kernels' arithmetic, whatever the batch statistics of a 2 x 2 x 2 map amplify upstream
    import oracle.nn as onn
    from oracle import graph as og
    from ultralytics.hip.runtime import Runtime
    sd = og.fill_state(og.state_layout(g), 7 + mi)
    sd = {k: (v.half().float() if v.dim() >= 4 else v) for k, v in sd.items()}
    det = g.layers[-1]
    src = [Runtime.to_tensor(acts[j]).float().cpu() for j in det.f]
    onn.STORAGE_FP16 = True
    try:
        with torch.no_grad():
            ref_feats = onn.apply_layer(det, sd, src, True, g.strides)
    finally:
        onn.STORAGE_FP16 = False
    for l, (f, r) in enumerate(zip(feats, ref_feats)):
        e = l2err(f, r)
        print(f"{name} feat{l} vs fp16-storage oracle on the engine's inputs {e:.2e}")
        assert e < 2e-3, f"level {l}"
    # (2) the whole chain against the fp32 reference: informative at this fixture size (measured 0.9e-2 .. 4.4e-2: the coarse levels'
    # batch statistics come from 2 x 4 x 4 and 2 x 2 x 2 samples); the eval-mode comparison below is the tight end-to-end one
    for l, f in enumerate(feats):
        e = l2err(f, G.t(f"{name}/feat{l}"))
        print(f"{name} feat{l} l2err vs fp32 reference (whole chain) {e:.2e}")
        assert e < (4e-2 if l == 0 else 0.15)
    assert torch.isfinite(plan.rt.flat_g).all() and float(plan.state[2]) == 0.0
    m2, _ = _model(name, 7 + mi)
    m2.cuda().eval()
    x = G.t(f"{name}/img").cuda()
    y, feats = m2(x)
    ref = G.t(f"{name}/y_eval")
    assert y.shape == ref.shape and len(feats) == 4
    eb, ec = relerr(y[:, :4].cpu(), ref[:, :4]), relerr(y[:, 4:].cpu(), ref[:, 4:])
    print(f"{name} eval: box relerr {eb:.2e} cls relerr {ec:.2e}")
    assert eb < 2e-2 and ec < 2e-2
    m2.fuse()
    yf, _ = m2(x)
    reff = G.t(f"{name}/y_eval_fused")
    eb, ec = relerr(yf[:, :4].cpu(), reff[:, :4]), relerr(yf[:, 4:].cpu(), reff[:, 4:])
    print(f"{name} fused eval: box relerr {eb:.2e} cls relerr {ec:.2e}")
    assert eb < 2e-2 and ec < 2e-2
    assert sum(p.numel() for p in m2.parameters()) == int(G[f"{name}/n_params_fused"])


def test_p2_1280_batch32_fused_forward_and_validator_nms():
    """BASELINE configs[4] at full size: yolov8n-p2, 1280x1280, batch 32, fused, then the validator's NMS (conf 0.001, IoU 0.7,
    multi_label) where every image has more than max_nms = 30000 candidates, compared with the oracle's restatement of the
    reference on the SAME decoded tensor (two of the 32 images: the oracle's sequential loop takes seconds per image)."""
    from ultralytics.utils.ops import non_max_suppression
    m, g = _model("yolov8n-p2", 10)
    m.cuda().eval()
    m.fuse()
    B, S = 32, 1280
    x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        y, feats = m(x)
    A = sum((S // s) ** 2 for s in (4, 8, 16, 32))
    assert y.shape == (B, 84, A) and A == 136000 and torch.isfinite(y).all()
    # a randomly initialised head scores every class ~1e-5: spread the class scores with ONE monotone affine map in logit space
    # (applied to the tensor that both sides then read) placed so that image 0 has 100,000 (anchor, class) pairs above conf = 0.001
    # and 3,000 above soft-NMS's fixed 0.25
    s = y[:, 4:].clamp(1e-7, 1 - 1e-7)
    z = torch.log(s) - torch.log1p(-s)
    top = torch.topk(z[0].flatten(), 100000).values
    z1, z2 = float(top[-1]), float(top[2999])
    l1, l2 = float(np.log(0.001 / 0.999)), float(np.log(0.25 / 0.75))
    l3, z3 = float(np.log(0.99 / 0.01)), float(z.max())  # piecewise linear above z2 so that the top scores do not saturate into ties
    zz = torch.where(z <= z2, l1 + (z - z1) * ((l2 - l1) / (z2 - z1)), l2 + (z - z2) * ((l3 - l2) / (z3 - z2)))
    y = torch.cat((y[:, :4], torch.sigmoid(zz)), 1).contiguous()
    del s, z, zz, top
    out = non_max_suppression(y, conf_thres=0.001, iou_thres=0.7, multi_label=True, max_det=300)
    torch.cuda.synchronize()
    assert len(out) == B and all(o.shape[1] == 6 and o.shape[0] <= 300 for o in out)
    ncand = (y[:2, 4:] > 0.001).sum((1, 2)).tolist()
    print("candidates per image", ncand, "above 0.25", (y[:2, 4:] > 0.25).sum((1, 2)).tolist(), "kept", [int(o.shape[0]) for o in out[:4]])
    assert min(ncand) > 30000, "the fixture must exercise the n > max_nms pre-sort"
    ref = onms.non_max_suppression(y[:2].cpu(), conf_thres=0.001, iou_thres=0.7, multi_label=True, max_det=300)
    for b in range(2):
        got = out[b].cpu()
        assert got.shape == ref[b].shape, (got.shape, ref[b].shape)
        bad = ((got[:, :4] != ref[b][:, :4]).any(1) | (got[:, 5] != ref[b][:, 5])).nonzero().view(-1)
        if len(bad):
            k = int(bad[0])
            print(f"image {b}: {len(bad)} of {len(got)} rows differ, first at {k}: got {got[k].tolist()} ref {ref[b][k].tolist()}; "
                  f"max |box diff| {float((got[:, :4] - ref[b][:, :4]).abs().max()):.3e}")
        # the reference's pre-sort is an unstable argsort (utils/ops.py:396): candidates of EQUAL confidence (they occur: image 1 of this
        # fixture holds anchors 60 px apart with bit-identical scores) come out in an unspecified order there, in candidate order
        # here -- so rows are compared after putting both lists into one canonical order
        def canon(t):
            for c in (3, 2, 1, 0, 5, 4):
                t = t[torch.sort(t[:, c], stable=True).indices]
            return t
        cg, cr = canon(got), canon(ref[b])
        assert torch.equal(cg[:, 5], cr[:, 5]) and torch.equal(cg[:, :4], cr[:, :4]), f"image {b}: kept boxes / classes"
        assert float((cg[:, 4] - cr[:, 4]).abs().max()) <= 1e-6, f"image {b}: decayed confidences"
        assert len(bad) <= 0.05 * len(got), f"image {b}: {len(bad)} rows out of place -- more than score ties explain"
