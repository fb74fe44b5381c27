"""-m gpu: dy_detection_loss (target packing, DFL decode, task-aligned assignment, BCE + CIoU|WIoU + NWD + DFL, analytic
gradients) against the reference-generated fixtures tests/golden/loss.npz.  All arithmetic is fp32 on both sides:
losses within 1e-4 relative, assignment tensors exact, gradients within 2e-3 of the largest entry (they are emitted
in fp16)."""
import pytest
import torch

from golden.cases import MODES, loss_cases
from gpu_util import relerr

pytestmark = pytest.mark.gpu
SHAPES = [(16, 16), (8, 8), (4, 4)]


def _headout(G, case, grads=True):
    from ultralytics.nn.modules.head import HeadOut
    box, cls = [], []
    for l in range(3):
        f = G.t(f"{case}/feat{l}").cuda().permute(0, 2, 3, 1).contiguous()
        box.append(f[..., :64].contiguous())
        c = torch.zeros(*f.shape[:3], 8, device="cuda")
        c[..., :6] = f[..., 64:]
        cls.append(c)
    ho = HeadOut(box, cls, 6, [4.0, 8.0, 16.0])
    if grads:
        ho.alloc_grads()
    return ho


class _M:  # minimal stand-in exposing what v8DetectionLoss reads from a model
    class _Det:
        stride = torch.tensor([4.0, 8.0, 16.0])
        nc, no, reg_max = 6, 70, 16

    def __init__(self):
        self.model = [self._Det()]
        self._p = torch.zeros(1, device="cuda")

    def parameters(self):
        yield self._p


@pytest.mark.parametrize("case", list(loss_cases()))
@pytest.mark.parametrize("mode", list(MODES))
def test_loss_vs_golden(golden, case, mode):
    from ultralytics.utils.loss import v8DetectionLoss
    G = golden("loss")
    crit = v8DetectionLoss(_M())
    crit.bbox_loss.use_wiseiou, crit.bbox_loss.nwd_loss = MODES[mode]
    batch = {k: G.t(f"{case}/{k}") for k in ("batch_idx", "cls", "bboxes")}
    n_calls = 3 if (MODES[mode][0] and case == "random5") else 1
    for call in range(n_calls):
        ho = _headout(G, case)
        loss, items = crit(ho, batch)
        torch.cuda.synchronize()
        tag = f"{case}/{mode}" + (f"/call{call}" if n_calls > 1 else "")
        assert abs(float(loss) - float(G[f"{tag}/loss"])) <= 1e-4 * abs(float(G[f"{tag}/loss"])) + 1e-5, (float(loss), float(G[f"{tag}/loss"]))
        assert relerr(items, G.t(f"{tag}/items")) < 1e-4
        for l in range(3):
            ref = G.t(f"{tag}/gfeat{l}").permute(0, 2, 3, 1)
            got = torch.cat((ho.dbox[l].float(), ho.dcls[l][..., :6].float()), -1).cpu()
            if float(ref.abs().max()) == 0:
                assert float(got.abs().max()) == 0
            else:
                assert relerr(got, ref) < 2e-3, f"grad level {l}"
        if MODES[mode][0]:
            assert abs(float(crit.bbox_loss.wiou_loss.iou_mean) - float(G[f"{tag}/iou_mean"])) < 1e-5
    if mode == "ciou":  # assignment: exact
        A = sum(h * w for h, w in SHAPES)
        ws = crit  # per-anchor state lives in the workspace; re-derive through the public scalars + gradients:
        fg_ref = G.t(f"{case}/fg_mask").bool()
        assert int(round(float(crit.scalars[3]))) == int(fg_ref.sum()), "number of foreground anchors"
        tss_ref = max(float(G.t(f"{case}/target_scores").sum()), 1.0)
        assert abs(float(crit.scalars[1]) - tss_ref) < 1e-4 * tss_ref
        gt, ts, _ = crit.debug_assignment()
        gt, ts = gt.cpu(), ts.cpu()
        assert torch.equal(gt >= 0, fg_ref), "fg_mask"
        tgi = G.t(f"{case}/target_gt_idx").long()
        assert torch.equal(gt[fg_ref].long(), tgi[fg_ref]), "target_gt_idx"
        assert relerr(ts, G.t(f"{case}/target_scores").sum(-1)) < 1e-5, "target_scores"
        _ = (A, ws)
