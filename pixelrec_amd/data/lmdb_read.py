"""Read-only access to an LMDB data file without liblmdb (the reference stores the item images in one,
code/generate_lmdb.py:42-71, and reads it with the `lmdb` module, data/dataset/trainset.py:100,147-160; that module is
not installed in this image).

What is parsed: the on-disk layout liblmdb 0.9.x writes on a 64-bit little-endian machine (data format version 1), as
published in its source documentation:

  page      = 16-byte header {pgno u64, pad u16, flags u16, lower u16, upper u16 | n_overflow_pages u32}, then u16 node
              offsets (from the page start) up to `lower`; #keys = (lower - 16) / 2
  meta page = header + {magic 0xBEEFC0DE u32, version u32, address u64, mapsize u64, 2 x db record (48 bytes: pad u32 --
              the page size in the first record --, flags u16, depth u16, branch/leaf/overflow page counts, #entries,
              root page; all u64), last page u64, transaction id u64}; pages 0 and 1 are metas, the one with the larger
              transaction id is current; record 1 is the main (unnamed) database
  node      = {lo u16, hi u16, flags u16, key size u16, key bytes, data}; on a branch page lo|hi<<16|flags<<32 is the
              child page and node 0 carries no key (it sorts first); on a leaf lo|hi<<16 is the data size and the data
              follow the key, unless flag 0x01 says they live on overflow pages: then a u64 page number follows the
              key and the bytes start 16 bytes into that page, contiguous over as many pages as they need
  key order = bytewise, a prefix sorting first (liblmdb's default comparator) -- Python's `bytes` ordering.

Only what the reference's files need is built: the unnamed database with default flags (no dup-sort, no integer or
reversed keys, no named sub-databases), lookups and in-order iteration.  Anything else raises LmdbError.

PINNING: no liblmdb build, `lmdb` module or `.mdb` file exists offline, so this parser is checked only against files
laid out by tests/lmdb_fixture.py from the same published description ("parity unpinned").  When the `lmdb` module is
importable, pixelrec_amd.data.images uses it instead of this file.
"""
from __future__ import annotations

import mmap
import os
import struct

MAGIC = 0xBEEFC0DE
DATA_VERSION = 1
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
DB_REVERSEKEY, DB_DUPSORT, DB_INTEGERKEY = 0x02, 0x04, 0x08
PAGEHDR = 16
P_INVALID = (1 << 64) - 1


class LmdbError(RuntimeError):
    pass


class LmdbFile:
    """`LmdbFile(path)` (a directory holding data.mdb, or the file itself -- lmdb.open's subdir=True/False);
    `.get(key) -> bytes | None`, `.items()` in key order, `len()` = number of entries."""

    def __init__(self, path: str):
        if os.path.isdir(path):
            path = os.path.join(path, "data.mdb")
        self.path = path
        self._f = open(path, "rb")
        try:
            self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError as e:   # empty file
            self._f.close()
            raise LmdbError(f"{path}: not an LMDB data file (empty)") from e
        first = self._read_meta(0)
        if first is None:
            raise LmdbError(f"{path}: not an LMDB data file (no valid meta page 0)")
        self.page_size = first["psize"]
        second = self._read_meta(self.page_size)
        meta = first if second is None or first["txnid"] >= second["txnid"] else second
        if meta["flags"] & (DB_REVERSEKEY | DB_DUPSORT | DB_INTEGERKEY):
            raise LmdbError(f"{path}: main database flags {meta['flags']:#x} (dup-sort / integer / reversed keys) "
                            "are not supported by this reader")
        self.entries, self.depth, self._root, self.txnid = meta["entries"], meta["depth"], meta["root"], meta["txnid"]

    # ---------------------------------------------------------------------------------------------- meta pages
    def _read_meta(self, off):
        mm = self._mm
        if off + PAGEHDR + 136 > len(mm):
            return None
        flags, = struct.unpack_from("<H", mm, off + 10)
        magic, version = struct.unpack_from("<II", mm, off + PAGEHDR)
        if not (flags & P_META) or magic != MAGIC:
            return None
        if version != DATA_VERSION:
            raise LmdbError(f"{self.path}: LMDB data format version {version} (this reader knows version 1)")
        psize, = struct.unpack_from("<I", mm, off + PAGEHDR + 24)                  # free-list record's pad = page size
        dflags, depth, _b, _l, _o, entries, root = struct.unpack_from("<HHQQQQQ", mm, off + PAGEHDR + 24 + 48 + 4)
        _last, txnid = struct.unpack_from("<QQ", mm, off + PAGEHDR + 24 + 96)
        if psize < 512 or psize & (psize - 1):
            raise LmdbError(f"{self.path}: implausible page size {psize}")
        return {"psize": psize, "flags": dflags, "depth": depth, "entries": entries, "root": root, "txnid": txnid}

    # ---------------------------------------------------------------------------------------------- pages / nodes
    def _page(self, pgno):
        off = pgno * self.page_size
        if off + PAGEHDR > len(self._mm):
            raise LmdbError(f"{self.path}: page {pgno} lies beyond the end of the file")
        flags, lower = struct.unpack_from("<HH", self._mm, off + 10)
        n = (lower - PAGEHDR) >> 1
        if flags & P_LEAF2:
            raise LmdbError(f"{self.path}: fixed-size duplicate pages are not supported")
        if n < 0 or PAGEHDR + 2 * n > self.page_size:
            raise LmdbError(f"{self.path}: page {pgno} has a corrupt header")
        return off, flags, struct.unpack_from(f"<{n}H", self._mm, off + PAGEHDR)

    def _key(self, node):
        ksize, = struct.unpack_from("<H", self._mm, node + 6)
        return self._mm[node + 8:node + 8 + ksize]

    def _child(self, node):
        lo, hi, top = struct.unpack_from("<HHH", self._mm, node)
        return lo | (hi << 16) | (top << 32)

    def _data(self, node):
        lo, hi, flags, ksize = struct.unpack_from("<HHHH", self._mm, node)
        size, start = lo | (hi << 16), node + 8 + ksize
        if flags & (F_SUBDATA | F_DUPDATA):
            raise LmdbError(f"{self.path}: named sub-databases / duplicate values are not supported")
        if flags & F_BIGDATA:
            pgno, = struct.unpack_from("<Q", self._mm, start)
            off = pgno * self.page_size
            oflags, = struct.unpack_from("<H", self._mm, off + 10)
            if not (oflags & P_OVERFLOW):
                raise LmdbError(f"{self.path}: page {pgno} should be an overflow page")
            start = off + PAGEHDR
        if start + size > len(self._mm):
            raise LmdbError(f"{self.path}: value runs beyond the end of the file")
        return self._mm[start:start + size]

    # ---------------------------------------------------------------------------------------------- queries
    def __len__(self):
        return self.entries

    def get(self, key: bytes, default=None):
        if self._root == P_INVALID:
            return default
        pgno = self._root
        for _ in range(64):     # a B+tree this deep cannot exist; guards against cycles in a corrupt file
            off, flags, ptrs = self._page(pgno)
            if flags & P_BRANCH:
                lo, hi, at = 1, len(ptrs) - 1, 0     # node 0 has no key: it covers everything below node 1's key
                while lo <= hi:
                    mid = (lo + hi) >> 1
                    if self._key(off + ptrs[mid]) <= key:
                        at, lo = mid, mid + 1
                    else:
                        hi = mid - 1
                pgno = self._child(off + ptrs[at])
            elif flags & P_LEAF:
                lo, hi = 0, len(ptrs) - 1
                while lo <= hi:
                    mid = (lo + hi) >> 1
                    k = self._key(off + ptrs[mid])
                    if k == key:
                        return self._data(off + ptrs[mid])
                    if k < key:
                        lo = mid + 1
                    else:
                        hi = mid - 1
                return default
            else:
                raise LmdbError(f"{self.path}: page {pgno} is neither branch nor leaf (flags {flags:#x})")
        raise LmdbError(f"{self.path}: B+tree deeper than 64 levels (corrupt file)")

    def items(self):
        """(key, value) pairs in key order."""
        if self._root == P_INVALID:
            return
        stack = [(self._root, 0)]
        while stack:
            pgno, i = stack.pop()
            off, flags, ptrs = self._page(pgno)
            if flags & P_LEAF:
                for p in ptrs:
                    yield self._key(off + p), self._data(off + p)
            elif flags & P_BRANCH:
                if len(stack) > 64:
                    raise LmdbError(f"{self.path}: B+tree deeper than 64 levels (corrupt file)")
                if i + 1 < len(ptrs):
                    stack.append((pgno, i + 1))
                stack.append((self._child(off + ptrs[i]), 0))
            else:
                raise LmdbError(f"{self.path}: page {pgno} is neither branch nor leaf (flags {flags:#x})")

    def close(self):
        self._mm.close()
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
