"""DetectionValidator with the reference's method names (reference models/yolo/detect/val.py, engine/validator.py).

``update_metrics`` handles a whole batch with ONE launch of dy_match_predictions (label/prediction rescaling, IoU, greedy
matching at the ten IoU thresholds); the statistics stay on the device until ``get_stats``.  Dataset construction, JSON /
txt dumps, plots and the confusion matrix are control plane or data pipeline."""
import ctypes as C

import numpy as np
import torch

from ....hip import check, lib
from ....utils import LOGGER, ops
from ....utils.metrics import DetMetrics
from ....hip.engine import dev_empty


class DetectionValidator:
    def __init__(self, dataloader=None, save_dir=None, pbar=None, args=None, _callbacks=None):
        from ....cfg import get_cfg
        self.args = get_cfg(overrides=args if isinstance(args, dict) else None) if not hasattr(args, "conf") else args
        self.dataloader = dataloader
        self.device = None
        self.nc, self.names, self.seen = 0, {}, 0
        self.iouv = torch.linspace(0.5, 0.95, 10)  # mAP@0.5:0.95 thresholds (reference val.py:37)
        self.niou = self.iouv.numel()
        self.metrics = DetMetrics()
        self.stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[])
        self.training = False

    # ---- reference validator hooks -------------------------------------------------------------------------------
    def init_metrics(self, model):
        names = getattr(model, "names", None) or {i: str(i) for i in range(model.model[-1].nc)}
        self.names = dict(enumerate(names)) if isinstance(names, (list, tuple)) else names
        self.nc = len(self.names)
        self.metrics.names = self.names
        self.seen = 0
        self.stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[])

    def preprocess(self, batch):
        dev = self.device
        img = batch["img"].to(dev, non_blocking=True)
        if img.dtype == torch.uint8 and img.shape[-1] == 3 and img.shape[1] != 3:  # the loader's NHWC layout
            img = img.permute(0, 3, 1, 2)
        batch["img"] = img.float() / 255 if img.dtype == torch.uint8 else img.float()
        for k in ("batch_idx", "cls", "bboxes"):
            batch[k] = batch[k].to(dev)
        return batch

    def postprocess(self, preds):
        a = self.args
        conf = 0.001 if a.conf is None else a.conf  # reference engine/validator.py: default validation confidence
        return ops.non_max_suppression(preds, conf, a.iou, multi_label=True, agnostic=bool(a.single_cls), max_det=a.max_det)

    @staticmethod
    def _geometry(batch, B, imgsz):
        """(B,5) gain, padw, padh, ori_h, ori_w: what scale_boxes derives from ori_shape / ratio_pad (utils/ops.py)."""
        g = np.zeros((B, 5), np.float32)
        oris, rps = batch.get("ori_shape"), batch.get("ratio_pad")
        for i in range(B):
            oh, ow = (oris[i] if oris is not None else imgsz)
            rp = rps[i] if rps is not None else None
            if rp is None:
                gain = min(imgsz[0] / oh, imgsz[1] / ow)
                pad = (round((imgsz[1] - ow * gain) / 2 - 0.1), round((imgsz[0] - oh * gain) / 2 - 0.1))
            else:
                gain, pad = rp[0][0], rp[1]
            g[i] = (gain, pad[0], pad[1], oh, ow)
        return g

    def update_metrics(self, preds, batch):
        """preds: the NMS output list; batch: img / batch_idx / cls / bboxes (+ ori_shape, ratio_pad when letterboxed)."""
        dev = preds[0].device if len(preds) else batch["cls"].device
        B = len(preds)
        imgsz = tuple(batch["img"].shape[2:])
        counts = [int(p.shape[0]) for p in preds]
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum(counts)
        ntot = int(off[-1])
        packed = torch.cat([p.reshape(-1, 6).float() for p in preds], 0).contiguous() if ntot else torch.zeros((0, 6), device=dev)
        tcls = batch["cls"].reshape(-1).float().contiguous()
        tidx = batch["batch_idx"].reshape(-1).float().contiguous()
        tbox = batch["bboxes"].reshape(-1, 4).float().contiguous()
        tp = torch.zeros((ntot, self.niou), dtype=torch.uint8, device=dev)
        predn = dev_empty((ntot, 6), torch.float32, dev)
        if ntot:
            geom = torch.from_numpy(self._geometry(batch, B, imgsz)).to(dev)
            offd = torch.from_numpy(off).to(dev)
            if getattr(self, "_status", None) is None or self._status.device != dev:
                self._status = torch.zeros(1, dtype=torch.int32, device=dev)  # ONE word for the whole run: every batch ORs into it
            status = self._status
            iouv = self.iouv.to(dev)
            check(lib().dy_match_predictions(packed.data_ptr(), offd.data_ptr(), tidx.data_ptr(), tcls.data_ptr(), tbox.data_ptr(),
                                             tcls.numel(), geom.data_ptr(), iouv.data_ptr(), self.niou, B, imgsz[0], imgsz[1],
                                             tp.data_ptr(), predn.data_ptr(), status.data_ptr(),
                                             torch.cuda.current_stream(dev).cuda_stream), "dy_match_predictions")
        self.seen += B
        # images with neither predictions nor labels contribute nothing; label-only images contribute their target classes
        self.stats["tp"].append(tp.bool())
        self.stats["conf"].append(predn[:, 4])
        self.stats["pred_cls"].append(predn[:, 5])
        self.stats["target_cls"].append(tcls)
        self.last_predn = predn
        return tp

    def get_stats(self):
        if getattr(self, "_status", None) is not None and int(self._status.item()) & 1:
            self._status.zero_()
            raise RuntimeError("an image carried more than 1024 labels (dy_match_predictions capacity)")
        stats = {k: torch.cat(v, 0).cpu().numpy() for k, v in self.stats.items() if len(v)}
        if len(stats) and stats["tp"].any():
            self.metrics.process(**stats)
        self.nt_per_class = np.bincount(stats["target_cls"].astype(int), minlength=self.nc) if len(stats) else np.zeros(self.nc, int)
        return self.metrics.results_dict

    def print_results(self):
        mp, mr, m50, m = self.metrics.mean_results()
        LOGGER.info(("%22s" + "%11i" * 2 + "%11.3g" * 4) % ("all", self.seen, int(self.nt_per_class.sum()), mp, mr, m50, m))

    # ---- driver (reference engine/validator.py:104-216, inference part) -------------------------------------------
    @torch.no_grad()
    def __call__(self, trainer=None, model=None):
        from ....utils.torch_utils import select_device
        model = model if model is not None else trainer.model
        self.device = next(model.parameters()).device if next(model.parameters()).is_cuda else select_device("0")
        model.to(self.device).eval()
        self.init_metrics(model)
        for batch in self.dataloader:
            batch = self.preprocess(batch)
            preds = self.postprocess(model(batch["img"]))
            self.update_metrics(preds, batch)
        stats = self.get_stats()
        self.print_results()
        return stats
