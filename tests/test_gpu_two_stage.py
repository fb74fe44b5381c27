"""-m gpu: two-stage inference kernels (crop + letterbox, refinement choice, per-class hard NMS) against the oracle and the
reference-generated fixture tests/golden/two_stage.npz, and the batched flow end to end."""
import os

import numpy as np
import pytest
import torch

from oracle import two_stage as ots

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "two_stage.npz"))


def test_crop_letterbox_kernel_matches_float_bilinear_reference():
    from ultralytics.utils.double_inference import prepare_cropped_images
    rng = np.random.default_rng(0)
    H, W = 300, 420
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    rects = [[0, 0, 40, 30], [100, 50, 420, 300], [200, 100, 232, 132], [5, 7, 16, 290], [380, 0, 420, 11], [0, 0, 420, 300], [17, 19, 18, 20]]
    crops, geos = prepare_cropped_images(torch.from_numpy(img).cuda(), [dict(x1=r[0], y1=r[1], x2=r[2], y2=r[3]) for r in rects])
    out = crops.cpu().numpy()
    for k, r in enumerate(rects):
        ref = ots.crop_letterbox(img, r)
        ratio, new, px, py = ots.crop_geometry(r)
        assert (geos[k]["ratio"], geos[k]["new_size"], geos[k]["pad_x"], geos[k]["pad_y"]) == (ratio, new, px, py)
        d = np.abs(out[k].astype(int) - ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, (k, d.max(), (d > 0).mean())  # same formula; a rounding tie may differ
        assert (out[k][:py] == 114).all() and (out[k][:, :px] == 114).all()


def test_refine_select_kernel_matches_reference():
    from ultralytics.hip import check, lib
    W, H = [int(v) for v in G["crop/wh"]]
    K = int(G["ref/n"])
    dets, off, orig, rects, scale = [], [0], [], [], []
    for k in range(K):
        c, lab, cf = G[f"ref/{k}/cand"], G[f"ref/{k}/labels"], G[f"ref/{k}/confs"]
        dets.append(np.concatenate([c.reshape(-1, 4), cf.reshape(-1, 1), lab.reshape(-1, 1).astype(np.float32)], 1).astype(np.float32))
        off.append(off[-1] + len(c))
        orig.append(G[f"ref/{k}/orig"])
        rects.append(G[f"ref/{k}/rect"])
        scale.append(G[f"ref/{k}/geom"])
    t = lambda a, dt: torch.tensor(np.asarray(a), dtype=dt).cuda().contiguous()  # noqa: E731
    dets_d, off_d = t(np.concatenate(dets, 0), torch.float32), t(off, torch.int32)
    orig_d, rects_d, scale_d = t(orig, torch.float32), t(rects, torch.int32), t(scale, torch.float32)
    out = torch.zeros((K, 6), device="cuda")
    found = torch.zeros(K, dtype=torch.int32, device="cuda")
    check(lib().dy_refine_select(dets_d.data_ptr(), off_d.data_ptr(), orig_d.data_ptr(), rects_d.data_ptr(), scale_d.data_ptr(), K,
                                 float(W), float(H), out.data_ptr(), found.data_ptr(), None), "dy_refine_select")
    torch.cuda.synchronize()
    out, found = out.cpu().numpy(), found.cpu().numpy()
    n_hit = 0
    for k in range(K):
        ref = G[f"ref/{k}/out"]
        assert bool(found[k]) == (ref.size > 0), k
        if ref.size:
            n_hit += 1
            np.testing.assert_array_equal(out[k], ref.astype(np.float32))
    assert n_hit >= 5


def test_hard_nms_kernel_matches_reference():
    from ultralytics.utils.double_inference import torchvision_nms
    for k in range(int(G["nms/n"])):
        b, s, lab = G[f"nms/{k}/boxes"], G[f"nms/{k}/scores"], G[f"nms/{k}/labels"]
        kb, ks, kl = torchvision_nms(b.tolist(), s.tolist(), lab.tolist(), 0.45)
        assert np.array_equal(np.array(kb, np.float32).reshape(-1, 4), G[f"nms/{k}/kept_boxes"])
        assert np.array_equal(np.array(ks, np.float32), G[f"nms/{k}/kept_scores"])
        assert np.array_equal(np.array(kl, np.int64), G[f"nms/{k}/kept_labels"])
    assert torchvision_nms([], [], []) == ([], [], [])


def test_batched_flow_equals_the_reference_flow_on_the_same_second_stage(monkeypatch):
    """The whole device flow (crops -> second pass -> refinement -> merge) against the oracle's restatement of the script's
    per-detection loop.  The second pass is replaced by a seeded stand-in so that every branch is exercised whatever the
    (untrained) network outputs; the real second pass is run once for its contract (one (k,6) tensor per crop, clipped)."""
    from ultralytics.nn.tasks import DetectionModel
    from ultralytics.utils import double_inference as di
    torch.manual_seed(0)
    model = DetectionModel("yolov8n-ASF-P2P2.yaml", verbose=False).cuda().eval()
    rng = np.random.default_rng(5)
    H, W = 480, 640
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    n = 16
    c = np.stack([rng.uniform(40, W - 40, n), rng.uniform(40, H - 40, n)], 1)
    wh = rng.uniform(8, 120, (n, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1)
    pred = {"boxes": boxes.tolist(), "scores": rng.uniform(0.1, 0.6, n).tolist(), "labels": rng.integers(0, 3, n).tolist()}
    seen = {}
    real = di._second_stage

    def fake(model_, crops, conf, iou, bs):
        seen["real"] = real(model_, crops, conf, iou, bs)
        g = np.random.default_rng(9)
        out = []
        for k in range(crops.shape[0]):
            m = int(g.integers(0, 9))
            cxy = g.uniform(200, 440, (m, 2))
            half = g.uniform(60, 260, (m, 2))
            b = np.concatenate([cxy - half, cxy + half], 1).clip(0, 640)
            out.append(torch.tensor(np.concatenate([b, g.uniform(0.25, 1, (m, 1)), g.integers(0, 3, (m, 1))], 1), dtype=torch.float32).cuda())
        seen["preds"] = out
        return out

    monkeypatch.setattr(di, "_second_stage", fake)
    replaced = 0
    for aligned in (True, False):
        out, dt = di.double_inference(torch.from_numpy(img), model, pred, aligned=aligned)
        idxs = [i for i, s in enumerate(pred["scores"]) if s >= 0.25]
        assert len(seen["real"]) == len(idxs) and all(p.shape[1] == 6 and (p[:, :4] >= 0).all() and (p[:, :4] <= 640).all() for p in seen["real"])
        rects = ots.optimal_crops([pred["boxes"][i] for i in idxs], W, H)
        cur = {k: list(v) for k, v in pred.items()}
        results = []
        for j, i in enumerate(idxs):
            p = seen["preds"][j].cpu().numpy()
            ratio, _, px, py = ots.crop_geometry(rects[j])
            sc = ots.scale_boxes(p[:, :4], px, py, rects[j], ratio)
            results.append(ots.refine(sc, p[:, 5].astype(int), p[:, 4], np.array(pred["boxes"][i], np.float32), pred["scores"][i],
                                      pred["labels"][i], W, H) if len(p) else None)
        replaced += sum(r is not None for r in results)
        pairs = zip(results, idxs) if aligned else zip([r for r in results if r is not None], idxs)
        for r, i in pairs:
            if r is not None:
                cur["boxes"][i], cur["scores"][i], cur["labels"][i] = r
        keep = ots.nms_per_class(cur["boxes"], cur["scores"], cur["labels"], 0.45)
        np.testing.assert_allclose(np.array(out["boxes"]), np.array(cur["boxes"], np.float32)[keep], rtol=0, atol=1e-3)
        np.testing.assert_allclose(np.array(out["scores"]), np.array(cur["scores"], np.float32)[keep], rtol=0, atol=1e-6)
        assert out["labels"] == [cur["labels"][k] for k in keep]
    assert replaced >= 2, "no refinement happened: the stand-in second pass does not exercise the replacement branch"
