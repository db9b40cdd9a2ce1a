"""Lays out a small LMDB data file (format version 1, 64-bit little-endian) for the reader tests: two meta pages, leaf
pages, as many branch levels as needed and overflow pages for large values -- following liblmdb's published page / node
description (see pixelrec_amd/data/lmdb_read.py).  Test infrastructure only: the product never writes LMDB files.
No liblmdb exists offline, so files made here are NOT verified against the real library."""
import os
import struct

PAGEHDR = 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
P_INVALID = (1 << 64) - 1


def _even(n):
    return (n + 1) & ~1


def _pack_page(pgno, flags, nodes, psize):
    """nodes: list of bytes; offsets grow up from the header, node bodies are stacked down from the page end."""
    page = bytearray(psize)
    upper = psize
    ptrs = []
    for nd in nodes:
        upper -= _even(len(nd))
        page[upper:upper + len(nd)] = nd
        ptrs.append(upper)
    lower = PAGEHDR + 2 * len(nodes)
    assert lower <= upper, "page overfull"
    struct.pack_into("<QHHHH", page, 0, pgno, 0, flags, lower, upper)
    struct.pack_into(f"<{len(ptrs)}H", page, PAGEHDR, *ptrs)
    return bytes(page)


def _fits(nodes, extra, psize):
    used = PAGEHDR + sum(2 + _even(len(n)) for n in nodes) + 2 + _even(len(extra))
    return used <= psize


def write_lmdb(path, items, psize=4096, max_nodes=None, current_meta=1, subdir=False):
    """items: {bytes key: bytes value}.  max_nodes caps the nodes per page (forces deep trees from little data).
    Returns the number of B+tree levels written."""
    if subdir:
        os.makedirs(path, exist_ok=True)
        path = os.path.join(path, "data.mdb")
    pages = {}
    nxt = [2]

    def alloc(n=1):
        p = nxt[0]
        nxt[0] += n
        return p

    nodemax = (((psize - PAGEHDR) // 2) & -2) - 2
    level = []          # (first key, pgno) of the pages of the level being built
    n_leaf = n_branch = n_over = 0

    cur, first = [], None

    def flush(flags):
        nonlocal cur, first, n_leaf, n_branch
        if not cur:
            return
        pg = alloc()
        pages[pg] = _pack_page(pg, flags, cur, psize)
        level.append((first, pg))
        if flags == P_LEAF:
            n_leaf += 1
        else:
            n_branch += 1
        cur, first = [], None

    for key in sorted(items):
        val = items[key]
        if 8 + len(key) + len(val) > nodemax:                     # value goes to overflow pages
            npg = (PAGEHDR - 1 + len(val)) // psize + 1
            pg = alloc(npg)
            blob = bytearray(npg * psize)
            struct.pack_into("<QHHI", blob, 0, pg, 0, P_OVERFLOW, npg)
            blob[PAGEHDR:PAGEHDR + len(val)] = val
            for i in range(npg):
                pages[pg + i] = bytes(blob[i * psize:(i + 1) * psize])
            n_over += npg
            node = struct.pack("<HHHH", len(val) & 0xFFFF, len(val) >> 16, F_BIGDATA, len(key)) + key + struct.pack("<Q", pg)
        else:
            node = struct.pack("<HHHH", len(val) & 0xFFFF, len(val) >> 16, 0, len(key)) + key + val
        if cur and (not _fits(cur, node, psize) or (max_nodes and len(cur) >= max_nodes)):
            flush(P_LEAF)
        if not cur:
            first = key
        cur.append(node)
    flush(P_LEAF)

    depth = 1 if level else 0
    while len(level) > 1:
        below, level = level, []
        for i, (k, pg) in enumerate(below):
            def mk(key):
                return struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, len(key)) + key
            node = mk(k)
            if cur and (not _fits(cur, node, psize) or (max_nodes and len(cur) >= max_nodes)):
                flush(P_BRANCH)
            if not cur:
                first = k
                node = mk(b"")                                     # node 0 of a branch page carries no key
            cur.append(node)
        flush(P_BRANCH)
        depth += 1
    root = level[0][1] if level else P_INVALID
    last = nxt[0] - 1

    def meta(pgno, txnid, live):
        page = bytearray(psize)
        struct.pack_into("<QHHHH", page, 0, pgno, 0, P_META, 0, 0)
        struct.pack_into("<IIQQ", page, PAGEHDR, 0xBEEFC0DE, 1, 0, 1 << 30)
        # free-list record: pad = page size, empty tree
        struct.pack_into("<IHHQQQQQ", page, PAGEHDR + 24, psize, 0, 0, 0, 0, 0, 0, P_INVALID)
        if live:
            struct.pack_into("<IHHQQQQQ", page, PAGEHDR + 72, 0, 0, depth, n_branch, n_leaf, n_over, len(items), root)
        else:
            struct.pack_into("<IHHQQQQQ", page, PAGEHDR + 72, 0, 0, 0, 0, 0, 0, 0, P_INVALID)
        struct.pack_into("<QQ", page, PAGEHDR + 120, last if live else 1, txnid)
        return bytes(page)

    pages[0] = meta(0, 2 if current_meta == 0 else 1, current_meta == 0)
    pages[1] = meta(1, 2 if current_meta == 1 else 1, current_meta == 1)
    with open(path, "wb") as f:
        for pg in range(last + 1):
            f.write(pages.get(pg, bytes(psize)))
    return depth
