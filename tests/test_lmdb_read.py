"""The build's read-only LMDB parser (pixelrec_amd/data/lmdb_read.py) against files laid out by tests/lmdb_fixture.py,
and the ImageStore filled through it from a database in the reference's image format (code/generate_lmdb.py:20-71:
key = ascii item token, value = pickle of an LMDB_Image {channels, size, image bytes, id}).  No liblmdb exists offline:
these tests pin the parser to the published page layout, not to the real library (see the module header)."""
import pickle

import numpy as np
import pytest
import torch

from pixelrec_amd.data.lmdb_read import LmdbError, LmdbFile
from tests.lmdb_fixture import write_lmdb


class LMDB_Image:  # same attribute names as the reference's class (data/utils.py:192-201); pickled under this module's path
    def __init__(self, image, iid):
        self.channels = image.shape[2]
        self.size = image.shape[:2]
        self.image = image.tobytes()
        self.id = iid


def _items(n, seed=0, big_every=5):
    rng = np.random.default_rng(seed)
    out = {}
    for i in range(n):
        size = int(rng.integers(70_000, 160_000)) if big_every and i % big_every == 0 else int(rng.integers(0, 300))
        out[str(int(rng.integers(0, 10**9))).encode("ascii") + b"k%d" % i] = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
    return out


@pytest.mark.parametrize("n,max_nodes,current_meta", [(7, None, 1), (200, 3, 0), (200, 3, 1), (1500, None, 1), (64, 2, 1)])
def test_lookup_and_iteration(tmp_path, n, max_nodes, current_meta):
    items = _items(n, seed=n)
    depth = write_lmdb(str(tmp_path / "db.mdb"), items, max_nodes=max_nodes, current_meta=current_meta)
    if max_nodes:
        assert depth >= 4                                    # really walks several branch levels
    with LmdbFile(str(tmp_path / "db.mdb")) as db:
        assert len(db) == n and db.depth == depth and db.txnid == 2
        for k, v in items.items():
            assert bytes(db.get(k)) == v                      # small values inline, > 2 KB values through overflow pages
        assert db.get(b"") is None and db.get(b"\xff" * 9) is None and db.get(b"5") is None
        some = sorted(items)[n // 2]
        assert db.get(some[:-1]) is None and db.get(some + b"0") is None and db.get(some, b"x") != b"x"
        got = list(db.items())
        assert [k for k, _ in got] == sorted(items)          # bytewise key order, a prefix sorting first
        assert all(bytes(v) == items[k] for k, v in got)


def test_subdir_and_empty_database(tmp_path):
    write_lmdb(str(tmp_path / "env"), {b"a": b"1", b"ab": b"2"}, subdir=True)
    with LmdbFile(str(tmp_path / "env")) as db:              # lmdb.open(path, subdir=True): <path>/data.mdb
        assert bytes(db.get(b"a")) == b"1" and bytes(db.get(b"ab")) == b"2" and db.get(b"b") is None
    write_lmdb(str(tmp_path / "empty.mdb"), {})
    with LmdbFile(str(tmp_path / "empty.mdb")) as db:
        assert len(db) == 0 and db.get(b"a") is None and list(db.items()) == []


def test_rejects_what_it_does_not_parse(tmp_path):
    p = tmp_path / "junk.mdb"
    p.write_bytes(b"\x00" * 8192)
    with pytest.raises(LmdbError, match="not an LMDB data file"):
        LmdbFile(str(p))
    (tmp_path / "zero.mdb").write_bytes(b"")
    with pytest.raises(LmdbError):
        LmdbFile(str(tmp_path / "zero.mdb"))
    write_lmdb(str(tmp_path / "dup.mdb"), {b"a": b"1"})
    raw = bytearray((tmp_path / "dup.mdb").read_bytes())
    for meta in (0, 4096):
        raw[meta + 16 + 24 + 48 + 4] = 0x04                  # main database flags: MDB_DUPSORT
    (tmp_path / "dup.mdb").write_bytes(bytes(raw))
    with pytest.raises(LmdbError, match="not supported"):
        LmdbFile(str(tmp_path / "dup.mdb"))
    raw[4096 + 16 + 4] = 9                                   # data format version of the current meta
    (tmp_path / "ver.mdb").write_bytes(bytes(raw))
    with pytest.raises(LmdbError, match="version"):
        LmdbFile(str(tmp_path / "ver.mdb"))


def test_image_store_from_reference_format_lmdb(tmp_path):
    """ImageStore.from_config on an LMDB of pickled LMDB_Image objects keyed by item token -> uint8 [N, 224, 224, 3]
    rows indexed by the INTERNAL item id, row 0 = the all-zero pad image (trainset.py:147-165, batchset.py:58-60)."""
    from pixelrec_amd.data.images import ImageStore

    rng = np.random.default_rng(4)
    tokens = ["[PAD]", "9007", "12", "555001", "7", "88"]
    imgs = {t: rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for t in tokens[1:]}
    db = {t.encode("ascii"): pickle.dumps(LMDB_Image(im, t)) for t, im in imgs.items()}
    db[b"__keys__"] = pickle.dumps([t.encode("ascii") for t in imgs])       # generate_lmdb.py:69-71
    db[b"__len__"] = pickle.dumps(len(imgs))
    write_lmdb(str(tmp_path / "covers.lmdb"), db)

    class DL:
        item_num = len(tokens)
        id2token = {"item_id": np.array(tokens)}

    store = ImageStore.from_config({"image_path": str(tmp_path / "covers.lmdb"), "seed": 0}, DL(), torch.device("cpu"))
    assert store.images.shape == (6, 224, 224, 3) and store.images.dtype == torch.uint8
    assert int(store.images[0].sum()) == 0
    for iid, t in enumerate(tokens[1:], start=1):
        assert np.array_equal(store.images[iid].numpy(), imgs[t])

    class DLMissing(DL):
        id2token = {"item_id": np.array(tokens[:-1] + ["404"])}

    with pytest.raises(KeyError, match="404"):
        ImageStore.from_config({"image_path": str(tmp_path / "covers.lmdb"), "seed": 0}, DLMissing(), torch.device("cpu"))
