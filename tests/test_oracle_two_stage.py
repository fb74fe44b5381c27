"""CPU: oracle/two_stage.py against tests/golden/two_stage.npz (outputs of the reference script's own functions)."""
import os

import numpy as np

from oracle import two_stage as ots

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "two_stage.npz"))


def test_crop_rectangles_match_reference():
    W, H = [int(v) for v in G["crop/wh"]]
    boxes = [[float(v) for v in b] for b in G["crop/boxes"]]
    assert np.array_equal(ots.optimal_crops(boxes, W, H), G["crop/rects"])


def test_scale_and_refine_match_reference():
    W, H = [int(v) for v in G["crop/wh"]]
    hits = 0
    for k in range(int(G["ref/n"])):
        cand, labels, confs = G[f"ref/{k}/cand"], G[f"ref/{k}/labels"], G[f"ref/{k}/confs"]
        orig, rect, (ratio, px, py) = G[f"ref/{k}/orig"], G[f"ref/{k}/rect"], G[f"ref/{k}/geom"]
        r2, _, px2, py2 = ots.crop_geometry(rect)
        assert (r2, px2, py2) == (ratio, px, py)
        scaled = ots.scale_boxes(cand, int(px), int(py), rect, float(ratio))
        assert np.array_equal(scaled, G[f"ref/{k}/scaled"])
        res = ots.refine(scaled, labels, confs, orig[:4].astype(np.float32), float(orig[4]), int(orig[5]), W, H) if len(cand) else None
        ref = G[f"ref/{k}/out"]
        if ref.size == 0:
            assert res is None, k
        else:
            hits += 1
            assert res is not None, k
            np.testing.assert_array_equal(np.array(res[0] + [res[1], res[2]], np.float64), ref)
    assert hits >= 5  # the cases do exercise the replacement branch


def test_per_class_nms_matches_reference():
    for k in range(int(G["nms/n"])):
        b, s, lab = G[f"nms/{k}/boxes"], G[f"nms/{k}/scores"], G[f"nms/{k}/labels"]
        keep = ots.nms_per_class(b, s, lab, 0.45)
        assert np.array_equal(b[keep], G[f"nms/{k}/kept_boxes"])
        assert np.array_equal(s[keep], G[f"nms/{k}/kept_scores"])
        assert np.array_equal(lab[keep], G[f"nms/{k}/kept_labels"])
