"""Data side of the hot path: the LMDB on-disk format, the dataset's sample contract, the
device-side bird-view crop and the GPU colour augmentation.  Unmarked cases run the kernels under the CPU emulator."""
import ctypes
import os

import numpy as np
import pytest
import torch

from learningbycheating_amd import _lib
from learningbycheating_amd.bird_view import augmenter as A
from learningbycheating_amd.bird_view.utils.datasets import image_lmdb as D
from learningbycheating_amd.bird_view.utils.datasets.lmdb_format import LmdbReader, write_lmdb

gpu = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- LMDB format ---------------------------------------------------------------------------------------------------
def test_lmdb_round_trip_inline_overflow_and_branch_pages(tmp_path):
    rng = np.random.RandomState(0)
    items = {"len": b"12"}
    for i in range(12):                                   # the reference's keys: two of them far larger than a page
        items["rgb_%04d" % i] = rng.randint(0, 256, 160 * 384 * 3, dtype=np.uint8).tobytes()
        items["birdview_%04d" % i] = rng.randint(0, 256, 320 * 320 * 7, dtype=np.uint8).tobytes()
        items["measurements_%04d" % i] = rng.randn(17).astype(np.float32).tobytes()
        items["control_%04d" % i] = rng.randn(3).astype(np.float32).tobytes()
    info = write_lmdb(str(tmp_path / "ep"), items)
    assert info["overflow_pages"] == 12 * (-(-(16 + 184320) // 4096) + -(-(16 + 716800) // 4096))
    r = LmdbReader(str(tmp_path / "ep"))
    assert r.entries == len(items) and r.keys() == sorted(k.encode() for k in items)
    for k, v in items.items():
        assert bytes(r.get(k)) == v, k
    assert r.get("rgb_0012") is None and r.get("") is None and r.get("zzz") is None and r.get(b"a") is None
    assert int(bytes(r.get("len"))) == 12
    r.close()
    # three tree levels: enough small keys for more than one branch page
    many = {"k%07d" % i: ("v%d" % i).encode() * (1 + i % 5) for i in range(60000)}
    info = write_lmdb(str(tmp_path / "big"), many)
    assert info["depth"] == 3 and info["branch_pages"] > 1
    r = LmdbReader(str(tmp_path / "big"))
    for i in list(range(0, 60000, 997)) + [0, 59999]:
        k = "k%07d" % i
        assert bytes(r.get(k)) == many[k]
    assert r.get("k0060000") is None and len(r.keys()) == 60000


def test_lmdb_meta_page_layout(tmp_path):
    """byte-level checks of what an LMDB 0.9 reader parses first (mdb.c MDB_meta): magic, version, page size, root, entries"""
    import struct
    write_lmdb(str(tmp_path / "e"), {"a": b"1", "b": b"22"})
    raw = open(tmp_path / "e" / "data.mdb", "rb").read()
    for pg in (0, 1):
        off = pg * 4096
        pgno, pad, flags = struct.unpack_from("<QHH", raw, off)
        assert pgno == pg and flags == 0x08
        magic, version = struct.unpack_from("<II", raw, off + 16)
        assert magic == 0xBEEFC0DE and version == 1
        assert struct.unpack_from("<I", raw, off + 16 + 24)[0] == 4096                 # free DB md_pad = page size
        depth, = struct.unpack_from("<H", raw, off + 16 + 24 + 48 + 6)
        entries, root = struct.unpack_from("<QQ", raw, off + 16 + 24 + 48 + 32)
        assert depth == 1 and entries == 2 and root == 2
    assert struct.unpack_from("<Q", raw, 4096 + 16 + 24 + 96 + 8)[0] == 1              # meta 1 carries the newer txnid
    flags, lower, upper = struct.unpack_from("<HHH", raw, 2 * 4096 + 10)
    assert flags == 0x02 and lower == 16 + 4                                         # a leaf with two node pointers


def test_lmdb_reader_on_hand_assembled_environment():
    """tests/golden/lmdb_handmade/data.mdb was NOT written by write_lmdb: tests/golden/make_lmdb_fixture.py lays its bytes out from the
    LMDB 0.9 structure definitions with what a real two-transaction environment has and our writer never produces -- the current meta
    page is page 0 (txnid 2; page 1 still points at the first transaction's stale root), a populated free DB, a branch root whose first
    key is empty, leaf bodies in insertion order under sorted pointers, one value on three overflow pages.  (python-lmdb is not in this
    image and the reference ships no recorded episode, so a file from data_collector.py:234-252 itself cannot be had.)"""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_lmdb_fixture", os.path.join(here, "golden", "make_lmdb_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    path = os.path.join(here, "golden", "lmdb_handmade")
    assert open(os.path.join(path, "data.mdb"), "rb").read() == mk.build()       # the committed bytes are what the script assembles
    rec = mk.records()
    r = LmdbReader(path)
    assert (r.psize, r.root, r.depth, r.entries) == (4096, 8, 2, 13)             # meta page 0 (txnid 2) wins over page 1 (txnid 1)
    assert r.keys() == sorted(rec)
    for k, v in rec.items():
        assert bytes(r.get(k)) == v, k
    assert bytes(r.get("len")) == b"3"                                           # not the stale b"0" of the first transaction
    assert len(r.get("birdview_0000")) == 9000                                   # overflow pages
    assert np.allclose(np.frombuffer(r.get("measurements_0002"), np.float32), [2 + 0.25 * j for j in range(17)])
    for missing in ("", "a", "birdview_0003", "control", "lem", "rgb_0003", "zzz"):     # before / between / after every leaf
        assert r.get(missing) is None, missing
    r.close()


# ---- dataset -------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dataset_dir(tmp_path_factory):
    root = tmp_path_factory.mktemp("lbc_data")
    D.write_synthetic_dataset(str(root), episodes=2, frames=40, seed=3)
    return str(root)


def test_world_to_pixel_matches_reference_fixture():
    """fixture = the reference's own function (AST-extracted from bird_view/utils/datasets/image_lmdb.py:22-30 by oracle/make_golden.py)"""
    g = torch.load(os.path.join(GOLD, "reference_outputs.pt"))["world_to_pixel"]
    for args, want in zip(g["args"], g["out"]):
        got = D.world_to_pixel(*[float(a) for a in args])
        assert np.allclose(got, want.numpy(), rtol=1e-6, atol=1e-5)


def test_image_dataset_sample_contract(dataset_dir):
    ds = D.ImageDataset(os.path.join(dataset_dir, "train"))
    assert len(ds) == 2 * (40 - 25)                                # every episode loses gap * n_step frames (image_lmdb.py:113)
    rgb, bv, loc, cmd, speed = ds[7]
    assert rgb.shape == (3, 160, 384) and rgb.dtype == torch.float32 and 0 <= rgb.min() and rgb.max() <= 1
    assert bv.shape == (7, 192, 192) and set(np.unique(bv.numpy()).tolist()) <= {0.0, 1.0}
    assert loc.shape == (5, 2) and loc.dtype == np.float64 and 1 <= cmd <= 4 and 0 < speed <= 10.5
    # the sample is the stored bytes: rgb / 255 in CHW, the bird-view window rows 58:250, cols 64:256
    r_u8, b_u8, loc2, cmd2, speed2 = ds.raw(7)
    assert torch.equal(rgb, torch.from_numpy(r_u8.copy()).permute(2, 0, 1).float() / 255)
    assert torch.equal(bv, torch.from_numpy(b_u8[58:250, 64:256].copy()).permute(2, 0, 1).float() / 255)
    # waypoints: the ego vehicle drives forward, so future positions lie ahead (above the ego pixel row 260 - 58 - 10 = 192 of
    # the crop ... in crop coordinates y decreases with distance) and get further away step by step
    assert np.all(np.diff(loc2[:, 1]) < 0) and np.all(np.abs(loc2[:, 0] - 96) < 40)
    env = ds.envs[ds.file_map[7]]
    m = np.frombuffer(env.get("measurements_%04d" % ds.idx_map[7]), np.float32)
    assert abs(speed2 - np.linalg.norm(m[5:8])) < 1e-6 and cmd2 == m[11]


@pytest.mark.parametrize("where", ["emulator", pytest.param("mi355x", marks=gpu)])
def test_device_loader_batches_crop_and_batch_aug(env, dataset_dir, where):
    """frames at the reference's sizes (160 x 384 RGB, 320 x 320 x 7 bird-view cropped to 192 x 192 on the device): the crop kernel
    and the staging path run on the CPU emulator and, marked gpu, on the MI355X"""
    dev, _ = env
    assert (dev.type == "cuda") == (where == "mi355x")
    ds = D.ImageDataset(os.path.join(dataset_dir, "train"))
    ld = D.DeviceLoader(ds, batch_size=3, samples=2, device=dev, seed=5)
    ref_rng = np.random.RandomState(5 * 9973)
    n = 0
    for rgb, bv, loc, cmd, speed in ld:
        idx = ref_rng.randint(len(ds), size=3)
        assert rgb.shape == (3, 160, 384, 3) and rgb.dtype == torch.uint8 and bv.shape == (3, 192, 192, 7) and bv.dtype == torch.uint8
        assert loc.shape == (3, 5, 2) and cmd.shape == (3,) and not cmd.is_cuda and speed.shape == (3,)
        for i, j in enumerate(idx):
            r_u8, b_u8, l, c, s = ds.raw(int(j))
            assert torch.equal(rgb[i].cpu(), torch.from_numpy(r_u8.copy()))
            assert torch.equal(bv[i].cpu(), torch.from_numpy(b_u8[58:250, 64:256].copy()))           # the device-side crop
            assert torch.allclose(loc[i].cpu(), torch.from_numpy(l).float()) and float(cmd[i]) == float(c)
        n += 1
    assert n == 2
    ld = D.DeviceLoader(ds, batch_size=2, samples=1, device=dev, batch_aug=3, seed=6)
    rgb, bv, loc, cmd, speed = next(iter(ld))
    assert rgb.shape[0] == 6 and torch.equal(rgb[0], rgb[2]) and torch.equal(bv[3], bv[5]) and not torch.equal(rgb[0], rgb[3])
    assert torch.equal(cmd[:3], cmd[:1].expand(3))


# ---- the privileged agent's loader: rotation + window jitter (reference birdview_lmdb.py:33-199) --------------------------------
def _warp_affine_u8(src, im, y0, x0, H, W):
    """numpy restatement of OpenCV's 8-bit bilinear cv2.warpAffine (imgwarp.cpp: 1/1024 fixed-point coordinates from the inverted matrix,
    cvRound, + 16, >> 5; 32 x 32 table of 15-bit weights forced to sum to 32768; (sum + 16384) >> 15; constant zero border) on the window
    [y0, y0 + H) x [x0, x0 + W) of the destination.  cv2 is not installed here: this twin, like the kernel, follows the published
    algorithm -- PARITY UNPINNED against cv2 itself."""
    SH, SW, C = src.shape
    ys, xs = np.arange(y0, y0 + H)[:, None].astype(np.float64), np.arange(x0, x0 + W)[None, :].astype(np.float64)
    adelta, bdelta = np.rint(im[0] * xs * 1024.0).astype(np.int64), np.rint(im[3] * xs * 1024.0).astype(np.int64)
    X0 = np.rint((im[1] * ys + im[2]) * 1024.0).astype(np.int64) + 16
    Y0 = np.rint((im[4] * ys + im[5]) * 1024.0).astype(np.int64) + 16
    X, Y = (X0 + adelta) >> 5, (Y0 + bdelta) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    ax, ay = fx.astype(np.float32) * np.float32(1 / 32), fy.astype(np.float32) * np.float32(1 / 32)
    wf = np.stack([(1 - ay) * (1 - ax), (1 - ay) * ax, ay * (1 - ax), ay * ax], -1).astype(np.float32)
    w = np.rint(wf * np.float32(32768)).astype(np.int64)
    diff = w.sum(-1) - 32768
    mn, mx = w.argmin(-1), w.argmax(-1)                # (first occurrence, as the scan with strict comparisons)
    fix = np.where(diff < 0, mx, mn)
    np.put_along_axis(w, fix[..., None], np.take_along_axis(w, fix[..., None], -1) - diff[..., None], -1)
    out = np.zeros((H, W, C), np.int64)
    for k, (oy, ox) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + oy, sx + ox
        ok = (yy >= 0) & (yy < SH) & (xx >= 0) & (xx < SW)
        v = src[np.clip(yy, 0, SH - 1), np.clip(xx, 0, SW - 1)].astype(np.int64) * ok[..., None]
        out += v * w[..., k:k + 1]
    return np.clip((out + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("where", ["emulator", pytest.param("mi355x", marks=gpu)])
def test_birdview_warp_crop_kernel(env, where):
    """lbc_birdview_warp_crop_u8 = the numpy restatement bit for bit (rotations up to +-15 degrees about the ego pixel, windows that reach
    the zero border); angle 0 = the plain window"""
    import ctypes
    from learningbycheating_amd import _lib
    dev, _ = env
    rng = np.random.RandomState(11)
    N = 6
    src = rng.randint(0, 256, size=(N, 320, 320, 7)).astype(np.uint8)
    cases = [(0, 0, -10), (5, 3, -8), (-5, -5, -10), (15, 0, -5), (-15, 5, -10), (3, -60, 40)]      # the last: the window leaves the image
    params = np.stack([D.warp_params(a, dx, dy) for a, dx, dy in cases])
    d_src, d_par = torch.from_numpy(src).to(dev), torch.from_numpy(params).to(dev)
    out = torch.empty((N, 192, 192, 7), dtype=torch.uint8, device=dev)
    _lib.check(_lib.get().lbc_birdview_warp_crop_u8(_lib.ptr(d_src), _lib.ptr(out), _lib.ptr(d_par), N, 320, 320, 7, 192, 192, _lib.stream_for(out)))
    out = out.cpu().numpy()
    for n in range(N):
        y0, x0 = params[n, 6:7].view(np.int32)
        want = _warp_affine_u8(src[n], params[n, :6], int(y0), int(x0), 192, 192)
        assert np.array_equal(out[n], want), (cases[n], np.abs(out[n].astype(int) - want.astype(int)).max())
    assert np.array_equal(out[0], src[0, 58:250, 64:256])


def test_birdview_dataset_jitter_cap_and_biased_sampling(dataset_dir):
    from learningbycheating_amd.bird_view.utils.datasets import birdview_lmdb as B
    ds = B.BirdViewDataset(os.path.join(dataset_dir, "train"), crop_x_jitter=5, crop_y_jitter=4, angle_jitter=5)
    base = D.ImageDataset(os.path.join(dataset_dir, "train"))
    assert len(ds) == len(base)
    # window jitter alone moves the waypoints with the window (birdview_lmdb.py:139-140)
    _, _, l0, c0, s0 = ds.raw(3)
    _, _, l1, _, _ = ds.raw(3, 0, 4, -7)
    assert np.allclose(l1, l0 - np.array([4, 3]))
    # rotation: a marker painted at a waypoint of the stored map lands on the rotated sample's waypoint (the map turns about the ego pixel
    # (160, 260), the waypoints with the ego orientation: birdview_lmdb.py:107-125 -- our warp direction must be the one that agrees)
    for theta in (5, -5):
        _, _, lr, _, _ = ds.raw(3, theta, 0, -D.PIXEL_OFFSET)
        full = np.zeros((320, 320, 1), np.uint8)
        px, py = l0[2, 0] + 64, l0[2, 1] + 58                 # third waypoint in stored-map pixels (x right, y down)
        full[int(round(py)) - 1:int(round(py)) + 2, int(round(px)) - 1:int(round(px)) + 2] = 255
        p = D.warp_params(theta, 0, -D.PIXEL_OFFSET)
        y0, x0 = p[6:7].view(np.int32)
        w = _warp_affine_u8(full, p[:6], int(y0), int(x0), 192, 192)[..., 0].astype(np.float64)
        cy, cx = (w * np.arange(192)[:, None]).sum() / w.sum(), (w * np.arange(192)[None, :]).sum() / w.sum()
        assert abs(cx - lr[2, 0]) < 1.5 and abs(cy - lr[2, 1]) < 1.5, (theta, cx, cy, lr[2])
    rng = np.random.RandomState(3)
    js = np.array([ds.draw_jitter(rng) for _ in range(400)])
    assert js[:, 0].min() == -5 and js[:, 0].max() == 5 and js[:, 1].min() == -5 and js[:, 1].max() == 5 and js[:, 2].min() == -10 and js[:, 2].max() == -6
    # frame cap: episodes in reverse-sorted order until max_frames frames are listed (birdview_lmdb.py:64-83)
    capped = B.BirdViewDataset(os.path.join(dataset_dir, "train"), max_frames=20)
    assert len(capped) == 20 and len(capped.envs) == 2 and capped.file_map[0] == 0
    # command-biased sampling: the command is drawn with the given ratios, then a frame of it
    biased = B.BiasedBirdViewDataset(os.path.join(dataset_dir, "train"), left_ratio=0.0, right_ratio=0.0, straight_ratio=0.0)
    assert sum(len(v) for v in biased.cmd_map.values()) == len(biased)
    picks = [biased.sample_index(rng) for _ in range(50)]
    assert all(p in biased.cmd_map[4] for p in picks)


@pytest.mark.parametrize("where", ["emulator", pytest.param("mi355x", marks=gpu)])
def test_device_loader_with_birdview_jitter(env, dataset_dir, where):
    """the loader draws (angle, dx, dy) per sample, shifts / rotates the waypoints on the host and warps the stored map on the device"""
    from learningbycheating_amd.bird_view.utils.datasets import birdview_lmdb as B
    dev, _ = env
    ds = B.BirdViewDataset(os.path.join(dataset_dir, "train"), crop_x_jitter=5, crop_y_jitter=0, angle_jitter=5)
    ld = D.DeviceLoader(ds, batch_size=3, samples=2, device=dev, seed=7)
    ref_rng = np.random.RandomState(7 * 9973)
    for rgb, bv, loc, cmd, speed in ld:
        idx = ref_rng.randint(len(ds), size=3)
        for i, j in enumerate(idx):
            a, dx, dy = ds.draw_jitter(ref_rng)
            _, b_u8, l, c, s = ds.raw(int(j), a, dx, dy)
            p = D.warp_params(a, dx, dy)
            y0, x0 = p[6:7].view(np.int32)
            assert np.array_equal(bv[i].cpu().numpy(), _warp_affine_u8(b_u8, p[:6], int(y0), int(x0), 192, 192))
            assert torch.allclose(loc[i].cpu(), torch.from_numpy(l).float()) and float(cmd[i]) == float(c)


# ---- augmentation: numpy twin of csrc/data.hip ---------------------------------------------------------------------
def _h32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF; x ^= x >> 16
    return x


def _hash3(seed, a, b):
    return _h32(np.uint64(seed) ^ _h32((np.asarray(a, np.uint64) * 0x9E3779B9 + _h32(np.asarray(b, np.uint64) + 0x85EBCA6B)) & 0xFFFFFFFF))


def _u01(h):
    return (h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)


def _clip(v):
    return np.clip(np.rint(v), 0, 255).astype(np.float32)


def numpy_augment(img, p):
    """img (H,W,3) uint8, p = one lbc_aug_params: the operator sequence of csrc/data.hip in numpy"""
    H, W, _ = img.shape
    v = img.astype(np.float32)
    pix = np.arange(H * W, dtype=np.uint64).reshape(H, W)
    yy, xx = np.divmod(np.arange(H * W).reshape(H, W), W)
    for k in range(p.n_ops):
        op = p.order[k]
        if op == A.BLUR:
            if p.blur_sigma <= 1e-3:
                continue
            rad = min(int(4.0 * p.blur_sigma + 0.5), 16)
            d = np.arange(-rad, rad + 1)
            w = np.exp(-(d * d).astype(np.float32) * np.float32(0.5 / (p.blur_sigma * p.blur_sigma))).astype(np.float32)
            w /= w.sum()

            def refl(i, n):
                i = np.where(i < 0, -i - 1, i)
                i = np.where(i >= n, 2 * n - 1 - i, i)
                return np.clip(i, 0, n - 1)
            tmp = sum(w[j] * v[:, refl(np.arange(W) + d[j], W)] for j in range(len(d)))
            v = _clip(sum(w[j] * tmp[refl(np.arange(H) + d[j], H)] for j in range(len(d))))
        elif op == A.NOISE:
            for c in range(3):
                cc = c if p.noise_pc else 0
                u1 = _u01(_hash3(p.seed, 0x100 + cc, pix)) + np.float32(0.5 / 16777216.0)
                u2 = _u01(_hash3(p.seed, 0x110 + cc, pix))
                z = np.sqrt(-2.0 * np.log(u1)) * np.cos(np.float32(6.28318530718) * u2)
                v[..., c] = _clip(v[..., c] + np.float32(p.noise_scale) * z.astype(np.float32))
        elif op == A.COARSE_DROPOUT:
            cell = (yy * p.coarse_h // H) * p.coarse_w + xx * p.coarse_w // W
            for c in range(3):
                cc = c if p.coarse_pc else 0
                v[..., c] = np.where(_u01(_hash3(p.seed, 0x200 + cc, cell)) < np.float32(p.coarse_p), 0, v[..., c])
        elif op == A.DROPOUT:
            for c in range(3):
                cc = c if p.dropout_pc else 0
                v[..., c] = np.where(_u01(_hash3(p.seed, 0x300 + cc, pix)) < np.float32(p.dropout_p), 0, v[..., c])
        elif op == A.ADD:
            v = _clip(v + np.array(list(p.add), np.float32))
        elif op == A.MULTIPLY:
            v = _clip(v * np.array(list(p.multiply), np.float32))
        elif op == A.CONTRAST:
            v = _clip(np.float32(128.0) + np.array(list(p.contrast), np.float32) * (v - np.float32(128.0)))
    return v.astype(np.uint8)


@pytest.mark.parametrize("shape", [(3, 20, 36), pytest.param((16, 160, 384), marks=gpu)])
def test_augmentation_kernels_match_the_numpy_twin(env, shape):
    dev, _ = env
    N, H, W = shape
    rng = np.random.RandomState(11)
    imgs = rng.randint(0, 256, (N, H, W, 3), dtype=np.uint8)
    # a late-training recipe (every operator fires, large magnitudes) so that all code paths run
    recipe = A.super_hard(40_000_000)
    assert recipe.frequency == 1.0 and recipe.color == 1.0
    recipe.color = 0.5
    params, any_blur = recipe.sample(N, np.random.RandomState(12), H, W)
    assert any_blur and all(params[i].n_ops == 7 for i in range(N))
    got = A.BatchAugmenter(recipe).augment_batch(torch.from_numpy(imgs.copy()).to(dev), params=(params, any_blur)).cpu().numpy()
    worst, nbad = 0, 0
    for i in range(N):
        want = numpy_augment(imgs[i], params[i])
        diff = np.abs(got[i].astype(np.int32) - want.astype(np.int32))
        # the blur's float summation order and the device's logf / cosf differ from numpy's by an ulp: a value at x.5 before
        # rounding may land on the other side (+-1), and a +-1 that then passes through Multiply / Contrast grows by their factor
        worst = max(worst, int(diff.max()))
        nbad += int((diff > 0).sum())
    # measured on MI355X (16 frames of 160 x 384): 21 of 2.9 M values differ, by at most 6
    assert worst <= 12 and nbad <= 1e-3 * imgs.size, (worst, nbad)
    # the operators did something, on every image
    assert all(np.mean(got[i] != imgs[i]) > 0.5 for i in range(N))


@pytest.mark.parametrize("hw", [(12, 20), pytest.param((160, 384), marks=gpu)])
def test_augmentation_single_operators_are_exact(env, hw):
    dev, _ = env
    rng = np.random.RandomState(13)
    img = rng.randint(0, 256, (2, hw[0], hw[1], 3), dtype=np.uint8)
    for op in (A.COARSE_DROPOUT, A.DROPOUT, A.ADD, A.MULTIPLY, A.CONTRAST):
        arr = (_lib.AugParams * 2)()
        for i in range(2):
            p = arr[i]
            p.n_ops, p.blur_pos, p.seed = 1, 1, 1000 + i
            p.order[0] = op
            p.coarse_p, p.coarse_h, p.coarse_w, p.coarse_pc = 0.3, 4, 5, i
            p.dropout_p, p.dropout_pc = 0.25, i
            for c in range(3):
                p.add[c], p.multiply[c], p.contrast[c] = (-40.0, 13.0, 90.0)[c], (0.4, 1.0, 2.7)[c], (0.5, 1.5, 1.0)[c]
        got = A.BatchAugmenter(None).augment_batch(torch.from_numpy(img.copy()).to(dev), params=(arr, False)).cpu().numpy()
        for i in range(2):
            assert np.array_equal(got[i], numpy_augment(img[i], arr[i])), op
    # hand-checked values: Add / Multiply / Contrast saturate and round to nearest
    one = np.array([[[[200, 100, 7]]]], dtype=np.uint8)
    arr = (_lib.AugParams * 1)()
    arr[0].n_ops, arr[0].blur_pos = 3, 3
    arr[0].order[0], arr[0].order[1], arr[0].order[2] = A.ADD, A.MULTIPLY, A.CONTRAST
    for c in range(3):
        arr[0].add[c], arr[0].multiply[c], arr[0].contrast[c] = 60.0, 0.5, 2.0
    got = A.BatchAugmenter(None).augment_batch(torch.from_numpy(one.copy()).to(dev), params=(arr, False)).cpu().numpy()
    # (200+60 -> 255, 100+60 = 160, 67) * 0.5 -> (127.5 -> 128, 80, 33.5 -> 34) ; 128 + 2 (v - 128) -> (128, 32, 0)
    assert got.reshape(-1).tolist() == [128, 32, 0]


def test_augmentation_schedule_follows_the_reference_formulas():
    """bird_view/augmenter.py:227-246 at two counters (hand-evaluated)"""
    r0 = A.super_hard(0)
    assert abs(r0.frequency - 0.05) < 1e-12 and r0.color == 0.0 and r0.ops[A.BLUR] == (0.0, 0.5) and r0.ops[A.ADD] == (-10.0, 10.0)
    it = 3_200_000 / 32.0                                  # iteration 100000
    r = A.super_hard(3_200_000)
    assert r.frequency == 1.0 and abs(r.color - 1.0) < 1e-12 and abs(r.ops[A.BLUR][1] - 1.0) < 1e-12 and abs(r.ops[A.ADD][1] - 20.0) < 1e-12
    assert abs(r.ops[A.MULTIPLY][1] - (1 + 2.5 * it / 200000.0)) < 1e-12 and abs(r.ops[A.MULTIPLY][0] - (1 - 0.91 * it / 500000.0)) < 1e-12
    d = 0.198667 + (0.03856658 - 0.198667) / (1 + (it / 196416.6) ** 1.863486)
    assert abs(r.ops[A.DROPOUT][1] - d) < 1e-12 and r.ops[A.COARSE_DROPOUT]["size_percent"] == (0.08, 0.2)
    # sampling statistics: operators fire with probability `frequency`, in random order
    rng = np.random.RandomState(0)
    rec = A.super_hard(32 * 20000)                         # frequency 0.45
    params, _ = rec.sample(4000, rng)
    assert abs(np.mean([params[i].n_ops for i in range(4000)]) - 7 * rec.frequency) < 0.1
    full, _ = A.super_hard(40_000_000).sample(400, rng)    # frequency 1: all seven fire, random_order shuffles them
    assert all(sorted(full[i].order[k] for k in range(7)) == list(range(7)) for i in range(400))
    assert len({full[i].order[0] for i in range(400)}) == 7 and len({full[i].blur_pos for i in range(400)}) == 7
