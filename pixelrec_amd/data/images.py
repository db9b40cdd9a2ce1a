"""Item images for the PixelNet models, kept RESIDENT IN HBM (reference: LMDB of pickled `LMDB_Image` objects read
per sample by 10 DataLoader workers, code/generate_lmdb.py:20-71, data/dataset/trainset.py:79-167, batchset.py).

An MI355X has 288 GB of HBM: the whole Pixel200K image set (96 K covers x 224x224x3 uint8 = 14.5 GB; PixelRec/8M:
408 K covers = 61 GB) fits next to the model, so a training batch is assembled ON THE DEVICE from item ids by one
kernel (pxr_image_u8_to_f32: gather + /255 + Normalize(0.5,0.5) + HWC->CHW; id 0 -> the all-zero pad image) instead of
352 decoded float images (212 MB) crossing PCIe every step.

Backends for filling the store (indexed by INTERNAL item id, row 0 = padding):
  * `<path>.npy`  uint8 [n, H, W, 3] + `<path>.tokens.txt` (one item token per line, same order)   -- native format;
  * LMDB in the reference's format -- key = ascii item token, value = pickle of an object with `.image` bytes,
    `.size`, `.channels` (generate_lmdb.py / data/utils.py:192-201); read through the `lmdb` module when it is
    installed, otherwise -- with a warning -- through the build's own EXPERIMENTAL read-only parser (data/lmdb_read.py:
    never checked against a file written by liblmdb itself);
  * `synthetic:<H>`  deterministic pseudo-random images (benchmarks / tests; no dataset is available offline).
"""
from __future__ import annotations

import io
import os
import pickle

import numpy as np
import torch


class _LmdbImage:  # stand-in for the reference's pickled LMDB_Image (any module path)
    def get_image(self):
        return np.frombuffer(self.image, dtype=np.uint8).reshape(*self.size, self.channels)


class _Unpickler(pickle.Unpickler):
    """Restricted unpickler for LMDB values: the reference's payload is `LMDB_Image{channels, size, image, id}`
    (generate_lmdb.py:20-40) whose attributes are ints, tuples, bytes and strings -- no other global is ever needed,
    so every other class / function reference in a value is refused (a pickle can otherwise run arbitrary code)."""

    def find_class(self, module, name):
        if name == "LMDB_Image":
            return _LmdbImage
        raise pickle.UnpicklingError(f"image LMDB value references {module}.{name}: only LMDB_Image is allowed")


class ImageStore:
    def __init__(self, images_u8: torch.Tensor):
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
        self.images = images_u8
        self.n, self.H, self.W = images_u8.shape[:3]

    def to(self, device):
        self.images = self.images.to(device)
        return self

    @staticmethod
    def synthetic(item_num: int, size: int = 224, seed: int = 0, device="cpu"):
        g = torch.Generator().manual_seed(seed)
        img = torch.randint(0, 256, (item_num, size, size, 3), generator=g, dtype=torch.uint8)
        img[0] = 0
        return ImageStore(img.to(device))

    @staticmethod
    def from_config(config, dataload, device):
        path = str(config["image_path"])
        id2token = dataload.id2token["item_id"]
        if path.startswith("synthetic:"):
            return ImageStore.synthetic(dataload.item_num, int(path.split(":")[1]), int(config["seed"] or 0), device)
        if path.endswith(".npy"):
            arr = np.load(path, mmap_mode="r")
            tokens = [t.strip() for t in open(path[:-4] + ".tokens.txt")]
            pos = {t: i for i, t in enumerate(tokens)}
            out = torch.zeros(dataload.item_num, *arr.shape[1:], dtype=torch.uint8)
            for iid in range(1, dataload.item_num):
                out[iid] = torch.from_numpy(np.ascontiguousarray(arr[pos[str(id2token[iid])]]))
            return ImageStore(out.to(device))
        try:
            import lmdb
        except ImportError:
            lmdb = None
        if lmdb is not None:
            env = lmdb.open(path, subdir=os.path.isdir(path), readonly=True, lock=False, readahead=False, meminit=False)
            txn = env.begin()
            fetch = txn.get
        else:   # no liblmdb binding: the build's own read-only parser of the same file
            import logging

            from .lmdb_read import LmdbFile

            # EXPERIMENTAL: written from liblmdb's published file layout and checked only against files laid out by this
            # repo's own test fixture -- no byte produced by liblmdb was available offline (parity unpinned, see
            # INTEGRATION.md).  The `.npy + .tokens.txt` store is the supported format; `pip install lmdb` the safe one.
            logging.getLogger().warning("image_path %s: the `lmdb` module is not installed; reading the file with the "
                                        "EXPERIMENTAL built-in parser (pixelrec_amd/data/lmdb_read.py)", path)
            fetch = LmdbFile(path).get
        out = None
        for iid in range(1, dataload.item_num):
            token = str(id2token[iid])
            blob = fetch(token.encode("ascii"))
            if blob is None:
                raise KeyError(f"image_path {path!r}: no image stored for item {token!r}")
            obj = _Unpickler(io.BytesIO(bytes(blob))).load()
            img = obj.get_image()[..., :3]
            if out is None:
                out = torch.zeros(dataload.item_num, *img.shape, dtype=torch.uint8)
            out[iid] = torch.from_numpy(np.array(img))      # copy: the pickled bytes are read-only
        if out is None:
            raise ValueError(f"image_path {path!r}: the catalogue has no items")
        return ImageStore(out.to(device))

    def batch(self, ids: torch.Tensor) -> torch.Tensor:
        """ids int64 [...] on the store's device -> fp32 [..., 3, H, W], reference-normalised, on the device."""
        from .. import ops

        return ops.image_u8_to_f32(self.images, ids.contiguous())


def interleave_pos_neg(items: torch.Tensor) -> torch.Tensor:
    """items [B, 2, L+1] (positives | negatives) -> ids [B, 2(L+1)] ordered pos_0, neg_0, pos_1, neg_1, ... as
    MOSEQTrainDataset stacks the images (trainset.py:145-165)."""
    B = items.shape[0]
    return items.permute(0, 2, 1).reshape(B, -1).contiguous()
