"""Dataset / loader builders (drop-in for reference data/build.py:84-124) and the loader that feeds the HIP step.

``HipDataLoader`` replaces torch's multi-process DataLoader with what the MI355X step needs at >4,000 images/s per GPU:
sample decoding on a thread pool (numpy / PIL release the GIL), batches assembled straight into a ring of PINNED uint8 NHWC
buffers (78.6 MB for 64x640x640x3, a quarter of the fp32 NCHW batch the reference ships over PCIe), the host-to-device copy
issued on its own HIP stream one batch ahead so that it overlaps the previous step's kernels, and an event the consumer's
stream waits on.  Epoch order reproduces the reference's: torch's RandomSampler drawing from a generator seeded
6148914691236517205 + RANK (build.py:110-111) after the DataLoader iterator's one base-seed draw, or DistributedSampler's
``randperm(seed = epoch)`` strided by rank when world_size > 1.

``cache='hbm'`` (training, fixed canvas size): the decoded, letterboxed dataset is uploaded ONCE into one device tensor -- a
100k-image 640x640 set is 123 GB of the MI355X's 288 GB -- and a batch is just (pool, int32 slot indices, flip bits): the import
kernel gathers straight from the pool, so steady-state training moves no pixel over PCIe and the host only builds labels."""
from __future__ import annotations

import math
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ..utils import RANK
from .dataset import YOLODataset

AUGMENT_KEYS = ("mosaic", "mixup", "copy_paste", "hsv_h", "hsv_s", "hsv_v", "degrees", "translate", "scale", "shear", "perspective",
                "flipud", "fliplr")


def build_yolo_dataset(cfg, img_path, batch, data, mode="train", rect=False, stride=32, layout="nhwc", flip_on_device=False):
    flips = {k: float(getattr(cfg, k, 0.0) or 0.0) for k in ("flipud", "fliplr", "mosaic", "degrees", "translate", "scale", "shear", "hsv_h", "hsv_s", "hsv_v",
                                                             "perspective", "mixup", "copy_paste")} if mode == "train" else {}
    return YOLODataset(img_path=img_path, imgsz=cfg.imgsz, batch_size=batch, augment=mode == "train", flip_on_device=flip_on_device, **flips,
                       rect=bool(getattr(cfg, "rect", False)) or rect, stride=int(stride), pad=0.0 if mode == "train" else 0.5,
                       data=data, fraction=getattr(cfg, "fraction", 1.0) if mode == "train" else 1.0,
                       cache=(getattr(cfg, "cache", False) or False) if mode == "train" or getattr(cfg, "cache", False) != "hbm" else False,
                       layout=layout, prefix=f"{mode}: ")


def build_dataloader(dataset, batch, workers, shuffle=True, rank=-1, world_size=1, device=None, drop_last=False):
    return HipDataLoader(dataset, min(batch, len(dataset)), workers, shuffle, rank, world_size, device, drop_last)


class HipDataLoader:
    def __init__(self, dataset, batch_size, workers=8, shuffle=True, rank=-1, world_size=1, device=None, drop_last=False, prefetch=2):
        self.dataset, self.batch_size, self.shuffle, self.rank, self.world_size = dataset, batch_size, shuffle, rank, max(world_size, 1)
        self.device = torch.device(device) if device is not None else None
        self.drop_last, self.prefetch = drop_last, prefetch
        self.workers = max(1, min(workers, os.cpu_count() or 1))
        self.generator = torch.Generator()
        self.generator.manual_seed(6148914691236517205 + RANK)
        self._base_seed_drawn = False
        self.epoch = 0
        self._pinned, self._copy_stream = {}, None

    # ---- order ---------------------------------------------------------------------------------------------------
    def set_epoch(self, epoch):
        self.epoch = epoch

    def _indices(self):
        n = len(self.dataset)
        if self.world_size > 1:  # torch DistributedSampler (seed 0, drop_last False): pad by wrapping, stride by rank
            if self.shuffle:
                g = torch.Generator()
                g.manual_seed(0 + self.epoch)
                idx = torch.randperm(n, generator=g).tolist()
            else:
                idx = list(range(n))
            total = math.ceil(n / self.world_size) * self.world_size
            pad = total - len(idx)
            idx += idx[:pad] if pad <= len(idx) else (idx * math.ceil(pad / len(idx)))[:pad]
            return idx[max(self.rank, 0):total:self.world_size]
        if not self.shuffle:
            return list(range(n))
        if not self._base_seed_drawn:  # _BaseDataLoaderIter.__init__ draws the workers' base seed from the same generator, once
            torch.empty((), dtype=torch.int64).random_(generator=self.generator)
            self._base_seed_drawn = True
        idx = torch.randperm(n, generator=self.generator).tolist()
        torch.randperm(n, generator=self.generator)  # RandomSampler's tail draw for num_samples % n (empty, still drawn)
        return idx

    def __len__(self):
        n = len(self.dataset) if self.world_size == 1 else math.ceil(len(self.dataset) / self.world_size)
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    # ---- HBM-resident dataset ------------------------------------------------------------------------------------
    def _build_pool(self, workers):
        ds = self.dataset
        if not (ds.augment and ds.flip_on_device and ds.layout == "nhwc" and not ds.rect):
            raise ValueError("cache='hbm' needs the training dataset in NHWC layout with flip_on_device=True")
        shape = tuple(ds.get(0, 0)["img"].shape)
        if ds.geometric and shape[0] != shape[1]:
            raise ValueError("mosaic / affine need square training canvases")
        n, chunk = len(ds), 256
        self._pool = torch.empty((n, *shape), dtype=torch.uint8, device=self.device)
        stage = torch.empty((min(chunk, n), *shape), dtype=torch.uint8).pin_memory()
        stage_np = stage.numpy()

        def load(j, i):
            np.copyto(stage_np[j], ds.get(i, 0)["img"].numpy())

        for lo in range(0, n, chunk):
            hi = min(lo + chunk, n)
            list(workers.map(lambda j: load(j - lo, j), range(lo, hi)))
            self._pool[lo:hi].copy_(stage[:hi - lo], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()  # the staging buffer is reused by the next chunk

    def _assemble_pool(self, workers, idx):
        flips = [self.dataset.draw_augment(i) for i in idx]
        # label arithmetic is a few dozen tiny numpy calls per sample: GIL-bound, so worker threads only add contention
        samples = [self.dataset.get(i, f, pixels=False) for i, f in zip(idx, flips)]
        batch = self.dataset.collate_fn(samples)
        batch["img"] = self._pool
        if "warp" not in batch:  # plain pool batch: slot indices (+ flip bits, HSV gains); mosaic / affine: the warp records
            batch["index"] = torch.tensor(idx, dtype=torch.int32)
        return batch  # host tensors: the consumer's thread moves them (see the threading rule below)

    # ---- batches -------------------------------------------------------------------------------------------------
    # Threading rule ("Threads and hipGraphs"): ONLY the consumer's thread calls into HIP.  The producer
    # thread and its decode workers do host work (decode, label arithmetic, memcpy into pinned slots the consumer allocated);
    # every host->device copy, event and device allocation is issued by the thread that also launches the step's hipGraphs.
    # Measured on this ROCm: a second thread enqueuing copies while hipGraphLaunch runs can break the ordering INSIDE the
    # replayed graph (a gradient buffer came back non-finite, every optimizer step was skipped, the run trained nothing).
    def _to_device(self, batch, keys):
        """Host -> device for the listed entries on the loader's own non-blocking stream (called from the consumer's thread),
        then one event the consumer's stream waits on.  The copy of batch k+1 is enqueued before batch k is handed out, so it
        runs on the copy engine under step k's kernels."""
        with torch.cuda.stream(self._copy_stream):
            for k in keys:
                if k in batch and torch.is_tensor(batch[k]) and not batch[k].is_cuda:
                    batch[k] = batch[k].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        batch["_ready"] = ev
        batch["_moved"] = [k for k in keys if k in batch]
        return ev

    def _assemble(self, pool, idx, slot):
        """One batch, host side only.  Each worker decodes a contiguous chunk of samples and copies the pixels straight into the
        batch buffer (memcpy, GIL released) -- the pinned ring slot the consumer allocated, or fresh pageable memory when the
        shape is not the ring's (or no device is set); only the label tensors go through collate_fn."""
        n, W = len(idx), min(self.workers, len(idx))
        flips = [self.dataset.draw_augment(i) for i in idx]  # RNG consumed here, in sample order, whatever the worker schedule
        first = self.dataset.get(idx[0], flips[0])
        shape = tuple(first["img"].shape)
        buf = self._pinned.get(slot)
        if buf is None or tuple(buf.shape) != (n, *shape):
            buf = torch.empty((n, *shape), dtype=torch.uint8)  # pageable: never a HIP call from this thread
        samples = [None] * n
        buf_np = buf.numpy()  # plain memcpy per image from the worker threads (torch's copy_ would nest its own thread pool)

        def work(lo, hi):
            for j in range(lo, hi):
                s = first if j == 0 else self.dataset.get(idx[j], flips[j])
                if tuple(s["img"].shape) != shape:
                    raise ValueError(f"images of one batch differ in shape: {tuple(s['img'].shape)} vs {shape} ({s['im_file']})")
                np.copyto(buf_np[j], s["img"].numpy())
                s["img"] = buf[j]
                samples[j] = s

        bounds = [n * k // W for k in range(W + 1)]
        list(pool.map(lambda k: work(bounds[k], bounds[k + 1]), range(W)))
        imgs = [s.pop("img") for s in samples]
        for s, v in zip(samples, imgs):  # keep the reference's key order: collate_fn walks batch[0].keys()
            s["img"] = v[:0]
        batch = self.dataset.collate_fn(samples)
        batch["img"] = buf
        return batch

    def __iter__(self):
        idx = self._indices()
        nb = len(self)
        chunks = [idx[i * self.batch_size:(i + 1) * self.batch_size] for i in range(nb)]
        dev = self.device is not None
        if dev and self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        q: queue.Queue = queue.Queue(maxsize=self.prefetch)
        free: queue.Queue = queue.Queue()  # pinned ring slots the producer may fill
        stop = threading.Event()

        hbm = dev and getattr(self.dataset, "cache_mode", False) == "hbm"
        if hbm and getattr(self, "_pool", None) is None:
            with ThreadPoolExecutor(self.workers) as pool:
                self._build_pool(pool)
        nslots = self.prefetch + 3
        if dev and not hbm:
            ds = self.dataset
            if ds.augment and not ds.rect and ds.layout == "nhwc":  # fixed canvas: pin the ring here, in the consumer's thread
                shape = (self.batch_size, int(ds.imgsz), int(ds.imgsz), 3)
                for s in range(nslots):
                    if s not in self._pinned or tuple(self._pinned[s].shape) != shape:
                        self._pinned[s] = torch.empty(shape, dtype=torch.uint8).pin_memory()
        for s in range(nslots):
            free.put(s)

        def produce():  # host work only -- see the threading rule above
            try:
                with ThreadPoolExecutor(self.workers) as pool:
                    for c in chunks:
                        slot = None
                        while not hbm and slot is None:
                            if stop.is_set():
                                return
                            try:
                                slot = free.get(timeout=0.05)
                            except queue.Empty:
                                pass
                        if stop.is_set():
                            return
                        q.put((self._assemble_pool(pool, c) if hbm else self._assemble(pool, c, slot), slot))
                q.put(None)
            except BaseException as e:  # surfaced in the consumer
                q.put(e)

        keys = ("warp", "index", "flip", "hsv") if hbm else ("img", "flip", "hsv")

        def stage(item):
            """Consumer's thread: enqueue the host->device copies of a produced batch on the copy stream."""
            if isinstance(item, BaseException):
                raise item
            if item is None:
                return None
            batch, slot = item
            ev = self._to_device(batch, keys) if dev else None
            return batch, slot, ev

        def release(slot, ev):
            if slot is not None:
                if ev is not None:
                    ev.synchronize()  # the copy out of this pinned slot was enqueued a whole step ago
                free.put(slot)

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            cur_item = stage(q.get())
            done = cur_item is None
            while not done:
                nxt, have_next = None, False
                try:  # batch k+1 already produced: its copy goes out before step k starts and overlaps it
                    nxt, have_next = stage(q.get_nowait()), True
                except queue.Empty:
                    pass
                b, slot, ev = cur_item
                ready = b.pop("_ready", None)
                if ready is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ready)
                    # device tensors allocated under the copy stream: tell the caching allocator that the consumer's stream
                    # reads them, or a block could be handed to a later copy while queued kernels still need it
                    for k in b.pop("_moved", ()):
                        if torch.is_tensor(b[k]) and b[k].is_cuda and b[k] is not getattr(self, "_pool", None):
                            b[k].record_stream(cur)
                yield b
                if not have_next:
                    nxt = stage(q.get())
                release(slot, ev)
                cur_item, done = nxt, nxt is None
        finally:
            stop.set()
            while t.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(0.05)
