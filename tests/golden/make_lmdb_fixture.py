"""Hand-assembled LMDB environment for the reader's parity test -- NOT produced by lmdb_format.write_lmdb.

python-lmdb / liblmdb are not installed in this image and /root/reference holds no recorded episode, so a file written by
the reference's own writer (data_collector.py:234-252 through py-lmdb 0.94) cannot be had.  This script lays the bytes of a
small environment out by hand from the LMDB 0.9.x structure definitions (lmdb.h / mdb.c: MDB_meta, MDB_db, MDB_page, MDB_node,
the overflow-page header, the free-list record), with the features a real two-transaction environment has and
`write_lmdb` never produces:

  * TWO committed transactions: meta page 0 carries txnid 2 (current, root = the branch page), meta page 1 carries txnid 1 and
    still points at the first transaction's root, a stale leaf holding `len` = b"0" -- a reader must pick the larger txnid;
  * a populated free DB (MDB_INTEGERKEY, one record: txnid 2 -> ID list [1, 9], the page the second transaction freed);
  * a root BRANCH page whose first node has key size 0 (mdb.c never compares it) over three leaves;
  * leaves whose node BODIES lie in insertion order (not key order) below the sorted pointer array, as mdb_node_add leaves them;
  * one value on OVERFLOW pages (9000 bytes -> 3 pages, F_BIGDATA node holding the first page number), every other value inline;
  * odd-sized nodes padded to even sizes.

Keys and value shapes are those of data_collector.py:234-252 (`len`, `rgb_%04d`, `birdview_%04d`, `measurements_%04d` = 17 float32,
`control_%04d` = 3 float32).  Values are a deterministic byte pattern (value_bytes below) so the test can regenerate them.

    python tests/golden/make_lmdb_fixture.py            # rewrites tests/golden/lmdb_handmade/data.mdb (45,056 bytes)
"""
import os
import struct

PSIZE = 4096
HDR = 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
MDB_INTEGERKEY = 0x08
INVALID = 0xFFFFFFFFFFFFFFFF


def value_bytes(key, n):
    seed = sum(key) * 131 + len(key)
    return bytes((seed + 7 * i + (i >> 8)) % 251 for i in range(n))


def records():
    """{key: value} of the current (second) transaction"""
    rec = {b"len": b"3"}
    for i in range(3):
        rec[b"rgb_%04d" % i] = value_bytes(b"rgb_%04d" % i, 600)
        rec[b"birdview_%04d" % i] = value_bytes(b"birdview_%04d" % i, 9000 if i == 0 else 1500)
        rec[b"measurements_%04d" % i] = struct.pack("<17f", *[i + 0.25 * j for j in range(17)])
        rec[b"control_%04d" % i] = struct.pack("<3f", 0.1 * i, 0.5, 0.0)
    return rec


def page_header(pgno, flags, lower=0, upper=0, pages=None):
    if pages is not None:       # overflow: the lower/upper pair is the 32-bit page count
        return struct.pack("<QHHI", pgno, 0, flags, pages)
    return struct.pack("<QHHHH", pgno, 0, flags, lower, upper)


def leaf_node(key, value, overflow_pgno=None):
    if overflow_pgno is None:
        body = struct.pack("<HHHH", len(value) & 0xFFFF, len(value) >> 16, 0, len(key)) + key + value
    else:
        body = struct.pack("<HHHH", len(value) & 0xFFFF, len(value) >> 16, F_BIGDATA, len(key)) + key + struct.pack("<Q", overflow_pgno)
    return body + (b"\0" if len(body) & 1 else b"")


def branch_node(key, child):
    body = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, len(key)) + key
    return body + (b"\0" if len(body) & 1 else b"")


def data_page(pgno, flags, nodes_in_key_order, insertion_order):
    """nodes_in_key_order: node images sorted by key; insertion_order: permutation giving the order in which the bodies were added
    (the first added body sits at the very end of the page)"""
    page = bytearray(PSIZE)
    upper = PSIZE
    where = {}
    for idx in insertion_order:
        body = nodes_in_key_order[idx]
        upper -= len(body)
        page[upper:upper + len(body)] = body
        where[idx] = upper
    lower = HDR + 2 * len(nodes_in_key_order)
    assert lower <= upper
    page[0:HDR] = page_header(pgno, flags, lower, upper)
    for i in range(len(nodes_in_key_order)):
        struct.pack_into("<H", page, HDR + 2 * i, where[i])
    return bytes(page)


def db_record(pad, flags, depth, branch, leaf, overflow, entries, root):
    return struct.pack("<IHHQQQQQ", pad, flags, depth, branch, leaf, overflow, entries, root)


def meta_page(pgno, txnid, free_db, main_db, last_pg):
    page = bytearray(PSIZE)
    page[0:HDR] = page_header(pgno, P_META)
    body = struct.pack("<IIQQ", 0xBEEFC0DE, 1, 0, 1 << 30) + free_db + main_db + struct.pack("<QQ", last_pg, txnid)
    page[HDR:HDR + len(body)] = body
    return bytes(page)


def build():
    rec = records()
    keys = sorted(rec)                                   # memcmp order
    groups = [[k for k in keys if k.startswith(b"birdview")],
              [k for k in keys if not k.startswith(b"birdview") and not k.startswith(b"rgb")],
              [k for k in keys if k.startswith(b"rgb")]]
    assert sum(len(g) for g in groups) == len(keys) and [k for g in groups for k in g] == keys
    LEAF_A, OV0, LEAF_B, LEAF_C, ROOT, OLD_ROOT, FREE_LEAF = 2, 3, 6, 7, 8, 9, 10
    pages = {}
    big = rec[b"birdview_0000"]
    ov_pages = (HDR + len(big) + PSIZE - 1) // PSIZE
    assert ov_pages == 3
    ov = bytearray(ov_pages * PSIZE)
    ov[0:HDR] = page_header(OV0, P_OVERFLOW, pages=ov_pages)
    ov[HDR:HDR + len(big)] = big
    pages[OV0] = bytes(ov)
    for pg, group, order in ((LEAF_A, groups[0], [2, 0, 1]), (LEAF_B, groups[1], [3, 0, 6, 1, 5, 2, 4]), (LEAF_C, groups[2], [1, 2, 0])):
        nodes = [leaf_node(k, rec[k], OV0 if k == b"birdview_0000" else None) for k in group]
        assert sorted(order) == list(range(len(nodes)))
        pages[pg] = data_page(pg, P_LEAF, nodes, order)
    pages[ROOT] = data_page(ROOT, P_BRANCH, [branch_node(b"", LEAF_A), branch_node(groups[1][0], LEAF_B), branch_node(groups[2][0], LEAF_C)], [0, 1, 2])
    pages[OLD_ROOT] = data_page(OLD_ROOT, P_LEAF, [leaf_node(b"len", b"0")], [0])
    # free DB: key = txnid (8 bytes, MDB_INTEGERKEY), data = ID list [count, pgno...]
    pages[FREE_LEAF] = data_page(FREE_LEAF, P_LEAF, [leaf_node(struct.pack("<Q", 2), struct.pack("<QQ", 1, OLD_ROOT))], [0])
    last = FREE_LEAF
    meta0 = meta_page(0, 2, db_record(PSIZE, MDB_INTEGERKEY, 1, 0, 1, 0, 1, FREE_LEAF), db_record(0, 0, 2, 1, 3, ov_pages, len(keys), ROOT), last)
    meta1 = meta_page(1, 1, db_record(PSIZE, MDB_INTEGERKEY, 0, 0, 0, 0, 0, INVALID), db_record(0, 0, 1, 0, 1, 0, 1, OLD_ROOT), OLD_ROOT)
    out = bytearray(meta0 + meta1)
    pg = 2
    while pg <= last:
        img = pages[pg]
        out += img
        pg += len(img) // PSIZE
    return bytes(out)


if __name__ == "__main__":
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lmdb_handmade")
    os.makedirs(d, exist_ok=True)
    data = build()
    with open(os.path.join(d, "data.mdb"), "wb") as f:
        f.write(data)
    print("wrote %d bytes" % len(data))
