"""CPU: the YOLO-format dataset reader + loader against tests/golden/data.npz, which holds what the
REFERENCE's YOLODataset + build_dataloader produced for the fixture dataset of golden.cases.write_dataset (train mode with
all augmentation gains zero, two shuffled epochs; val mode with rectangular batches).  Images, labels, order, shapes and
ratio_pad must be identical; plus the host-side behaviours the golden cannot hold (PNG decoding without the *.npy caches,
interpolated sizes, DDP sharding, data YAML checks)."""
import os
import shutil
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from golden.cases import DATASET_IMGSZ, write_dataset

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data.npz")


@pytest.fixture(scope="module")
def root(tmp_path_factory):
    r = str(tmp_path_factory.mktemp("dataset"))
    write_dataset(r)
    return r


def _loader(root, mode, layout="nchw", fliplr=0.0, flipud=0.0, geo=None, **kw):
    from ultralytics.data import build_dataloader, build_yolo_dataset, check_det_dataset
    data = check_det_dataset(os.path.join(root, "data.yaml"))
    cfg = SimpleNamespace(imgsz=DATASET_IMGSZ, rect=False, cache=False, fraction=1.0, fliplr=fliplr, flipud=flipud, **(geo or {}))
    ds = build_yolo_dataset(cfg, data[mode], 4, data, mode=mode, rect=mode == "val", stride=32, layout=layout)
    return build_dataloader(ds, 4, 2, shuffle=mode == "train", rank=-1, **kw)


def _check(G, tag, batch):
    assert [os.path.basename(f) for f in batch["im_file"]] == list(G[f"{tag}/files"]), tag
    assert torch.equal(batch["img"], torch.from_numpy(G[f"{tag}/img"])), tag
    for k in ("cls", "bboxes", "batch_idx"):
        ref = torch.from_numpy(G[f"{tag}/{k}"])
        assert batch[k].shape == ref.shape and torch.equal(batch[k], ref), (tag, k, batch[k], ref)
    assert np.array_equal(np.array(batch["ori_shape"]), G[f"{tag}/ori_shape"])
    assert np.array_equal(np.array(batch["resized_shape"]), G[f"{tag}/resized_shape"])
    if f"{tag}/ratio_pad" in G:
        rp = np.array([[r[0][0], r[0][1], r[1][0], r[1][1]] for r in batch["ratio_pad"]], dtype=np.float64)
        assert np.array_equal(rp, G[f"{tag}/ratio_pad"])


def test_train_batches_identical_to_reference_over_two_epochs(root):
    G = np.load(GOLD)
    loader = _loader(root, "train")
    assert len(loader) == int(G["train/nb"])
    assert len(loader.dataset) == 9  # 13 files, 4 dropped as corrupt
    for ep in range(2):
        for i, batch in enumerate(loader):
            _check(G, f"train/e{ep}/b{i}", batch)


def test_flipped_batches_identical_to_reference(root):
    """fliplr 0.5 / flipud 0.25: same images flipped as in the reference run (Python ``random`` seeded alike and consumed in the
    pipeline's order), pixels and labels identical."""
    import random
    G = np.load(GOLD)
    loader = _loader(root, "train", fliplr=0.5, flipud=0.25)
    random.seed(7)
    flipped = 0
    for ep in range(2):
        for i, batch in enumerate(loader):
            _check(G, f"flip/e{ep}/b{i}", batch)
            flipped += int(not torch.equal(batch["img"], torch.from_numpy(G[f"train/e{ep}/b{i}/img"])))
    assert flipped >= 3  # the fixture does contain flipped batches


def test_mosaic_affine_flip_labels_identical_to_reference(root):
    """mosaic 1.0, degrees 5, translate 0.1, scale 0.5, shear 2, fliplr 0.5, flipud 0.1 -- the whole geometric pipeline of the
    reference (Mosaic partners drawn from its image buffer, mosaic centre, affine matrix, box transform, clipping, candidate
    filter, flips) reproduced label for label over three epochs; Python's ``random`` seeded as in the reference run."""
    import random
    G = np.load(GOLD)
    loader = _loader(root, "train", geo=dict(mosaic=1.0, degrees=5.0, translate=0.1, scale=0.5, shear=2.0), fliplr=0.5, flipud=0.1)
    ds = loader.dataset
    random.seed(11)
    n_boxes = 0
    for ep in range(3):
        idx = loader._indices()
        for i in range(len(loader)):
            chunk = idx[i * 4:(i + 1) * 4]
            augs = [ds.draw_augment(j) for j in chunk]
            batch = ds.collate_fn([ds.get(j, a, pixels=False) for j, a in zip(chunk, augs)])
            tag = f"geo/e{ep}/b{i}"
            assert [os.path.basename(f) for f in batch["im_file"]] == list(G[f"{tag}/files"]), tag
            for k in ("cls", "bboxes", "batch_idx"):
                ref = torch.from_numpy(G[f"{tag}/{k}"])
                assert batch[k].shape == ref.shape and torch.equal(batch[k], ref), (tag, k, batch[k][:6], ref[:6])
            n_boxes += len(batch["cls"])
    assert n_boxes > 50


def test_val_rect_batches_identical_to_reference(root):
    G = np.load(GOLD)
    loader = _loader(root, "val")
    assert len(loader) == int(G["val/nb"])
    for i, batch in enumerate(loader):
        _check(G, f"val/b{i}", batch)


def test_nhwc_layout_and_png_decoding_agree_with_the_cached_arrays(root, tmp_path):
    a = [b for b in _loader(root, "val")]
    nhwc = [b for b in _loader(root, "val", layout="nhwc")]
    bare = str(tmp_path / "bare")
    shutil.copytree(root, bare)
    for split in ("train", "val"):
        for f in os.listdir(os.path.join(bare, "images", split)):
            if f.endswith(".npy"):
                os.remove(os.path.join(bare, "images", split, f))
    png = [b for b in _loader(bare, "val")]
    for x, y, z in zip(a, nhwc, png):
        assert torch.equal(x["img"], y["img"].permute(0, 3, 1, 2)) and torch.equal(x["img"], z["img"])
        assert torch.equal(x["bboxes"], z["bboxes"])


def test_interpolated_sizes_keep_geometry(tmp_path):
    """Long side != imgsz: load_image resizes (bilinear; the reference's cv2 fixed-point INTER_LINEAR is not pinned) -- shapes,
    padding and label geometry still follow the reference's formulas."""
    from PIL import Image
    from ultralytics.data import YOLODataset
    r = tmp_path / "ds"
    (r / "images" / "train").mkdir(parents=True)
    (r / "labels" / "train").mkdir(parents=True)
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, (30, 100, 3), dtype=np.uint8)).save(r / "images" / "train" / "a.png")
    (r / "labels" / "train" / "a.txt").write_text("1 0.5 0.5 0.4 0.6\n")
    s = YOLODataset(str(r / "images" / "train"), imgsz=64, augment=True, data={"nc": 2})[0]
    assert tuple(s["img"].shape) == (64, 64, 3) and s["ori_shape"] == (30, 100)
    # resized to 20x64 (ceil(30*0.64) = 20), centred: rows 22..41 hold the image, the rest is the 114 border
    assert (s["img"][:22] == 114).all() and (s["img"][42:] == 114).all() and not (s["img"][22:42] == 114).all()
    np.testing.assert_allclose(s["bboxes"].numpy(), [[0.5, 0.5, 0.4, 0.6 * 20 / 64]], rtol=1e-6)


def test_ddp_sharding_matches_torch_distributed_sampler(root):
    from torch.utils.data.distributed import DistributedSampler
    full = _loader(root, "train")
    n = len(full.dataset)
    for epoch in (0, 3):
        for rank in range(2):
            ld = _loader(root, "train", world_size=2)
            ld.rank = rank
            ld.set_epoch(epoch)
            ref = DistributedSampler(range(n), num_replicas=2, rank=rank, shuffle=True)
            ref.set_epoch(epoch)
            assert ld._indices() == list(ref)


def test_data_yaml_errors(tmp_path):
    from ultralytics.data import check_det_dataset
    p = tmp_path / "d.yaml"
    p.write_text("train: images/train\nnc: 2\n")
    with pytest.raises(SyntaxError):
        check_det_dataset(str(p))
    p.write_text("train: images/train\nval: images/val\nnc: 2\nnames: [a]\n")
    with pytest.raises(SyntaxError):
        check_det_dataset(str(p))
    p.write_text("train: images/train\nval: images/val\nnc: 2\n")
    with pytest.raises(FileNotFoundError):
        check_det_dataset(str(p))
    with pytest.raises(FileNotFoundError):
        check_det_dataset(str(tmp_path / "missing.yaml"))


def test_mixup_perspective_labels_identical_to_reference(root):
    """MixUp 0.5 + perspective 5e-4 (+ copy_paste 0.3: a no-op for box-only labels in the reference too) on top of mosaic 0.7, the
    affine gains and the flips: the nested draw order of the reference's pipeline -- Mosaic and RandomPerspective of the sample,
    MixUp's probability, its partner index, the partner's own Mosaic / RandomPerspective draws, numpy's beta(32, 32) -- and the
    perspective divide in the box transform, label for label over three epochs (tests/golden/data_mix.npz)."""
    import random
    G = np.load(os.path.join(os.path.dirname(GOLD), "data_mix.npz"))
    loader = _loader(root, "train", geo=dict(mosaic=0.7, mixup=0.5, copy_paste=0.3, degrees=5.0, translate=0.1, scale=0.5, shear=2.0, perspective=0.0005),
                     fliplr=0.5, flipud=0.1)
    ds = loader.dataset
    random.seed(13)
    np.random.seed(5)
    n_boxes = n_mix = 0
    for ep in range(3):
        idx = loader._indices()
        for i in range(len(loader)):
            chunk = idx[i * 4:(i + 1) * 4]
            augs = [ds.draw_augment(j) for j in chunk]
            n_mix += sum("mix" in a for a in augs)
            batch = ds.collate_fn([ds.get(j, a, pixels=False) for j, a in zip(chunk, augs)])
            tag = f"mix/e{ep}/b{i}"
            assert [os.path.basename(f) for f in batch["im_file"]] == list(G[f"{tag}/files"]), tag
            for k in ("cls", "bboxes", "batch_idx"):
                ref = torch.from_numpy(G[f"{tag}/{k}"])
                assert batch[k].shape == ref.shape and torch.equal(batch[k], ref), (tag, k, batch[k][:6], ref[:6])
            assert batch["warp"].shape == (len(chunk), 96)
            n_boxes += len(batch["cls"])
    assert n_boxes > 50 and n_mix >= 5
