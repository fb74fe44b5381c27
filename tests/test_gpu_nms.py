"""-m gpu: decode + soft-NMS kernels against the reference-generated fixtures tests/golden/nms.npz.  Kept indices must be
bit-exact; decayed scores may differ by one fp32 ulp (device exp vs the host's vectorised expf)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_soft_nms_cases(golden):
    from ultralytics.utils.ops import soft_nms
    G = golden("nms")
    names = sorted({k.split("/")[1] for k in G.keys("soft/")})
    for n in names:
        boxes = G.t(f"soft/{n}/boxes").view(-1, 4).cuda()
        scores = G.t(f"soft/{n}/scores_in").clone().cuda()
        keep = soft_nms(boxes, scores, float(G[f"soft/{n}/thr"]))
        assert keep.tolist() == G[f"soft/{n}/keep"].tolist(), n
        ref = G.t(f"soft/{n}/scores_out")
        assert torch.allclose(scores.cpu(), ref, rtol=3e-7, atol=0), n


@pytest.mark.parametrize("tag,kw", [
    ("predict", dict(conf_thres=0.25, iou_thres=0.7, max_det=300)),
    ("val", dict(conf_thres=0.001, iou_thres=0.7, multi_label=True, max_det=300)),
    ("agnostic", dict(conf_thres=0.25, iou_thres=0.45, agnostic=True, max_det=300)),
    ("classes", dict(conf_thres=0.2, iou_thres=0.6, classes=[1, 4], max_det=20)),
])
def test_nms_full(golden, tag, kw):
    from ultralytics.utils.ops import non_max_suppression
    G = golden("nms")
    out = non_max_suppression(G.t("nms/pred").cuda(), **kw)
    for i, o in enumerate(out):
        ref = G.t(f"nms/{tag}/img{i}")
        assert o.shape == ref.shape, (tag, i, o.shape, ref.shape)
        assert torch.equal(o[:, :4].cpu(), ref[:, :4]) and torch.equal(o[:, 5].cpu(), ref[:, 5]), (tag, i)
        assert torch.allclose(o[:, 4].cpu(), ref[:, 4], rtol=3e-7, atol=0)


def test_decode_and_eval_paths(golden):
    """Eval forward of DEAL-YOLO-N: unfused BN (running statistics) and fused (BN folded) against the reference."""
    import os
    from conftest import CFG_DIR
    from gpu_util import relerr
    from oracle import graph as og
    from ultralytics.nn.tasks import DetectionModel
    G = golden("models")
    name = "yolov8n-ASF-P2P2"
    m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
    m.load_state_dict(og.fill_state(og.state_layout(g), 7), strict=True)
    m.cuda().eval()
    y, feats = m(G.t(f"{name}/img").cuda())
    ref = G.t(f"{name}/y_eval")
    assert y.shape == ref.shape
    assert relerr(y[:, :4].cpu(), ref[:, :4]) < 2e-2 and relerr(y[:, 4:].cpu(), ref[:, 4:]) < 2e-2
    m.fuse()
    yf, _ = m(G.t(f"{name}/img").cuda())
    reff = G.t(f"{name}/y_eval_fused")
    assert relerr(yf[:, :4].cpu(), reff[:, :4]) < 2e-2 and relerr(yf[:, 4:].cpu(), reff[:, 4:]) < 2e-2
    assert sum(p.numel() for p in m.parameters()) == int(G[f"{name}/n_params_fused"])


@pytest.mark.parametrize("max_nms", [300, 1000, 30000])
def test_candidate_cap_presort_kernel(max_nms):
    """``x = x[x[:, 4].argsort(descending=True)[:max_nms]]`` (reference utils/ops.py:395-396) as dy_nms_presort: for every image with
    more than max_nms candidates the max_nms most confident ones in descending confidence, ties in candidate order (= a STABLE
    descending sort; the reference's unstable argsort leaves the order of equal confidences open), other images untouched.
    Confidences are drawn from a few hundred distinct values, so ties straddle the cut."""
    from ultralytics.hip import check, lib
    torch.manual_seed(max_nms)
    L = lib()
    B = 5
    counts = [max_nms * 3 + 17, max_nms, max_nms + 1, 7, max_nms * 2]
    cap = max(counts)
    sc = (torch.randint(1, 400, (B, cap)).float() / 400).cuda()          # many exact ties
    sc[4] = torch.rand(cap).cuda() * 0.9 + 0.05                           # and one image without
    bx = torch.rand(B, cap, 4).cuda()
    cl = torch.randint(0, 6, (B, cap)).float().cuda()
    cnt = torch.tensor(counts, dtype=torch.int32).cuda()
    ob, osc, ocl = torch.full((B, max_nms, 4), -1.0).cuda(), torch.full((B, max_nms), -1.0).cuda(), torch.full((B, max_nms), -1.0).cuda()
    ws = torch.empty(L.dy_nms_presort_workspace(B, max_nms), dtype=torch.uint8, device="cuda")
    check(L.dy_nms_presort(bx.data_ptr(), sc.data_ptr(), cl.data_ptr(), cnt.data_ptr(), B, cap, max_nms, ob.data_ptr(), osc.data_ptr(),
                           ocl.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "dy_nms_presort")
    torch.cuda.synchronize()
    assert cnt.tolist() == [min(c, max_nms) for c in counts]
    for b, n in enumerate(counts):
        k = min(n, max_nms)
        idx = torch.sort(sc[b, :n], descending=True, stable=True).indices[:k] if n > max_nms else torch.arange(k, device="cuda")
        assert torch.equal(osc[b, :k], sc[b, idx]) and torch.equal(ocl[b, :k], cl[b, idx]) and torch.equal(ob[b, :k], bx[b, idx]), b
