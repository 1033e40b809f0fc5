"""The LMDB on-disk format (data.mdb), read and written without the `lmdb` package.

The reference stores every recorded CARLA episode as one LMDB environment (data_collector.py:234-252: keys `len`,
`rgb_%04d`, `birdview_%04d`, `measurements_%04d`, `control_%04d`) and reads it back through python-lmdb==0.94
(bird_view/utils/datasets/image_lmdb.py:113-145).  This image ships no `lmdb`, so the B+tree file format is restated
here from the published layout of LMDB 0.9.x (mdb.c: MDB_meta / MDB_page / MDB_node):

    page 0, 1   meta pages: 16-byte page header, then magic 0xBEEFC0DE, version 1, address, mapsize, two MDB_db records
                (free list, main: pad/flags/depth, branch/leaf/overflow page counts, entries, root), last page, txnid;
                the meta with the larger txnid is current; the free DB's md_pad holds the page size
    data pages  header = pgno u64 | pad u16 | flags u16 | lower u16 | upper u16 (overflow pages: page count u32);
                u16 node offsets from byte 16 up to `lower`, nodes packed downwards from the page end
    node        lo u16 | hi u16 | flags u16 | ksize u16 | key | data.  Leaf: data size = lo | hi << 16; F_BIGDATA (0x01):
                the data field is the u64 number of the first overflow page.  Branch: child page = lo | hi << 16 | flags << 32,
                the key of the first node of a branch page is ignored (it stands for -infinity)
    keys        compared as byte strings (memcmp, shorter first on a common prefix)

`LmdbReader` mmaps the file and serves `get(key)` as zero-copy memoryviews (a frame's 184 KB / 717 KB value is never
copied on the host before it is staged for the H2D copy).  `write_lmdb` bulk-builds a valid single-transaction
environment (used for synthetic datasets and the reader's tests; python-lmdb is not available here to cross-check it, the
layout follows mdb.c's mdb_node_add / mdb_page_split conventions: even node sizes, overflow for values that do not fit
half a page).
"""
import mmap
import os
import struct

MAGIC = 0xBEEFC0DE
PAGEHDRSZ = 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
NODESIZE = 8
P_INVALID = 0xFFFFFFFFFFFFFFFF


class LmdbError(RuntimeError):
    pass


class LmdbReader:
    """read-only view of one LMDB environment (a directory holding data.mdb, or the file itself)"""

    def __init__(self, path):
        f = os.path.join(path, "data.mdb") if os.path.isdir(path) else path
        if not os.path.exists(f):
            raise LmdbError("no LMDB environment at %s" % path)
        self._file = open(f, "rb")
        self._map = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ)
        self._mv = memoryview(self._map)
        best = None
        for pg in (0, 1):
            # the page size is not known before a meta page is parsed: meta 1 sits at one page size, try the common ones
            for psize in ((0,) if pg == 0 else (4096, 8192, 16384, 65536, 512, 1024, 2048, 32768)):
                off = pg * psize
                if off + PAGEHDRSZ + 136 > len(self._mv):
                    continue
                m = self._parse_meta(off)
                if m is not None and (pg == 0 or m["psize"] == psize):
                    if best is None or m["txnid"] > best["txnid"]:
                        best = m
                    break
        if best is None:
            raise LmdbError("%s: no valid LMDB meta page" % f)
        self.psize, self.root, self.depth, self.entries = best["psize"], best["root"], best["depth"], best["entries"]

    def _parse_meta(self, off):
        flags = struct.unpack_from("<H", self._mv, off + 10)[0]
        magic, version = struct.unpack_from("<II", self._mv, off + PAGEHDRSZ)
        if not (flags & P_META) or magic != MAGIC or version != 1:
            return None
        base = off + PAGEHDRSZ + 8 + 8 + 8       # magic+version, address, mapsize
        free_pad = struct.unpack_from("<I", self._mv, base)[0]
        main = base + 48
        _pad, mflags, depth, _br, _lf, _ov, entries, root = struct.unpack_from("<IHHQQQQQ", self._mv, main)
        last_pg, txnid = struct.unpack_from("<QQ", self._mv, main + 48)
        return {"psize": free_pad, "root": root, "depth": depth, "entries": entries, "txnid": txnid, "last_pg": last_pg}

    def close(self):
        self._mv.release()
        self._map.close()
        self._file.close()

    def __len__(self):
        return self.entries

    # ---- B+tree walk -------------------------------------------------------------------------------------------
    def _page(self, pgno):
        off = pgno * self.psize
        flags, lower, upper = struct.unpack_from("<HHH", self._mv, off + 10)
        return off, flags, (lower - PAGEHDRSZ) // 2

    def _node(self, off, i):
        noff = off + struct.unpack_from("<H", self._mv, off + PAGEHDRSZ + 2 * i)[0]
        lo, hi, flags, ksize = struct.unpack_from("<HHHH", self._mv, noff)
        return noff, lo, hi, flags, ksize

    def _key(self, noff, ksize):
        return bytes(self._mv[noff + NODESIZE:noff + NODESIZE + ksize])

    def get(self, key, default=None):
        """value of `key` as a zero-copy memoryview into the mapped file (valid until close())"""
        if isinstance(key, str):
            key = key.encode()
        if self.root == P_INVALID:
            return default
        pgno = self.root
        while True:
            off, flags, n = self._page(pgno)
            if flags & P_BRANCH:
                lo_i, hi_i = 1, n - 1         # node 0's key is -infinity
                idx = 0
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) // 2
                    noff, lo, hi, nf, ks = self._node(off, mid)
                    if self._key(noff, ks) <= key:
                        idx = mid
                        lo_i = mid + 1
                    else:
                        hi_i = mid - 1
                noff, lo, hi, nf, ks = self._node(off, idx)
                pgno = lo | (hi << 16) | (nf << 32)
                continue
            if not (flags & P_LEAF):
                raise LmdbError("page %d: unexpected flags 0x%x" % (pgno, flags))
            lo_i, hi_i = 0, n - 1
            while lo_i <= hi_i:
                mid = (lo_i + hi_i) // 2
                noff, lo, hi, nf, ks = self._node(off, mid)
                k = self._key(noff, ks)
                if k == key:
                    dsize = lo | (hi << 16)
                    dstart = noff + NODESIZE + ks
                    if nf & F_BIGDATA:
                        ov = struct.unpack_from("<Q", self._mv, dstart)[0]
                        s = ov * self.psize + PAGEHDRSZ
                        return self._mv[s:s + dsize]
                    return self._mv[dstart:dstart + dsize]
                if k < key:
                    lo_i = mid + 1
                else:
                    hi_i = mid - 1
            return default

    def keys(self):
        """all keys in order (walks the leaves)"""
        out = []

        def walk(pgno):
            off, flags, n = self._page(pgno)
            for i in range(n):
                noff, lo, hi, nf, ks = self._node(off, i)
                if flags & P_BRANCH:
                    walk(lo | (hi << 16) | (nf << 32))
                else:
                    out.append(self._key(noff, ks))
        if self.root != P_INVALID:
            walk(self.root)
        return out


# ----------------------------------------------------------------------------------------------------------------
def _even(n):
    return (n + 1) & ~1


def write_lmdb(path, items, psize=4096, mapsize=None):
    """Create the environment `path`/data.mdb holding `items` ({key bytes/str: value bytes-like}) in one transaction."""
    os.makedirs(path, exist_ok=True)
    kv = sorted(((k.encode() if isinstance(k, str) else bytes(k)), v) for k, v in items.items())
    nodemax = ((psize - PAGEHDRSZ) // 2) & ~1         # mdb.c: me_nodemax = (((psize - PAGEHDRSZ) / MDB_MINKEYS) & -2) - sizeof(indx_t)
    nodemax -= 2
    pages = {}                                       # pgno -> bytes (data pages; overflow runs stored under their first pgno)
    next_pg = 2
    n_overflow = 0

    def new_page():
        nonlocal next_pg
        p = next_pg
        next_pg += 1
        return p

    def build_page(pgno, flags, nodes):
        """nodes: list of raw node bytes (even sized); returns the page image"""
        buf = bytearray(psize)
        upper = psize
        ptrs = []
        for nb in nodes:
            upper -= len(nb)
            buf[upper:upper + len(nb)] = nb
            ptrs.append(upper)
        lower = PAGEHDRSZ + 2 * len(nodes)
        assert lower <= upper, "page overflow"
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
        for i, p in enumerate(ptrs):
            struct.pack_into("<H", buf, PAGEHDRSZ + 2 * i, p)
        return bytes(buf)

    # ---- leaves ----
    level = []                                       # (first key, pgno) of every page of the level under construction
    cur_nodes, cur_first, cur_used = [], None, PAGEHDRSZ

    def flush_leaf():
        nonlocal cur_nodes, cur_first, cur_used
        if cur_nodes:
            pg = new_page()
            pages[pg] = build_page(pg, P_LEAF, cur_nodes)
            level.append((cur_first, pg))
        cur_nodes, cur_first, cur_used = [], None, PAGEHDRSZ

    for k, v in kv:
        v = memoryview(v).cast("B") if not isinstance(v, (bytes, bytearray)) else v
        dsize = len(v)
        if len(k) > 511:
            raise LmdbError("key longer than 511 bytes")
        if NODESIZE + len(k) + dsize > nodemax:      # the value goes to overflow pages
            npg = (PAGEHDRSZ + dsize + psize - 1) // psize
            ov = next_pg
            next_pg += npg
            n_overflow += npg
            img = bytearray(npg * psize)
            struct.pack_into("<QHHI", img, 0, ov, 0, P_OVERFLOW, npg)
            img[PAGEHDRSZ:PAGEHDRSZ + dsize] = v
            pages[ov] = bytes(img)
            node = struct.pack("<HHHH", dsize & 0xFFFF, dsize >> 16, F_BIGDATA, len(k)) + k + struct.pack("<Q", ov)
        else:
            node = struct.pack("<HHHH", dsize & 0xFFFF, dsize >> 16, 0, len(k)) + k + bytes(v)
        if len(node) & 1:
            node += b"\0"
        if cur_used + len(node) + 2 > psize:
            flush_leaf()
        if cur_first is None:
            cur_first = k
        cur_nodes.append(node)
        cur_used += len(node) + 2
    flush_leaf()
    n_leaf = len(level)
    n_branch = 0
    depth = 1 if level else 0
    # ---- branch levels ----
    while len(level) > 1:
        up, nodes, first, used = [], [], None, PAGEHDRSZ

        def flush_branch():
            nonlocal nodes, first, used, n_branch
            if nodes:
                pg = new_page()
                pages[pg] = build_page(pg, P_BRANCH, nodes)
                up.append((first, pg))
                n_branch += 1
            nodes, first, used = [], None, PAGEHDRSZ
        for k, child in level:
            key = b"" if not nodes else k            # the first key of a branch page is never compared
            node = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, len(key)) + key
            if len(node) & 1:
                node += b"\0"
            if used + len(node) + 2 > psize:
                flush_branch()
                node = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, 0)
            if first is None:
                first = k
            nodes.append(node)
            used += len(node) + 2
        flush_branch()
        level = up
        depth += 1
    root = level[0][1] if level else P_INVALID
    last_pg = next_pg - 1
    if mapsize is None:
        mapsize = max(next_pg * psize, 1 << 20)

    def meta(pgno, txnid):
        buf = bytearray(psize)
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, P_META, 0, 0)
        o = PAGEHDRSZ
        struct.pack_into("<IIQQ", buf, o, MAGIC, 1, 0, mapsize)
        o += 24
        struct.pack_into("<IHHQQQQQ", buf, o, psize, 0, 0, 0, 0, 0, 0, P_INVALID)                 # free DB (md_pad = page size)
        o += 48
        struct.pack_into("<IHHQQQQQ", buf, o, 0, 0, depth, n_branch, n_leaf, n_overflow, len(kv), root)
        o += 48
        struct.pack_into("<QQ", buf, o, last_pg if last_pg >= 1 else 1, txnid)
        return bytes(buf)

    with open(os.path.join(path, "data.mdb"), "wb") as f:
        f.write(meta(0, 0))
        f.write(meta(1, 1))
        pg = 2
        for p in sorted(pages):
            assert p == pg, "pages must be dense"
            f.write(pages[p])
            pg += len(pages[p]) // psize
    return {"pages": next_pg, "depth": depth, "leaf_pages": n_leaf, "branch_pages": n_branch, "overflow_pages": n_overflow}
