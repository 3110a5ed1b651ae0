"""CPU tests of the data path: the HDF5 reader/writer against files written by real h5py
(tests/golden/hdf5 — generator: oracle/gen_hdf5_golden.py), the drop-in CaptionDataset / TagDataset against
the oracle restatement, the value table of the normalisation kernel, and the loader's epoch bookkeeping."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import data_ref as DR
from scnattn import data as SD
from scnattn import h5lite

G = os.path.join(os.path.dirname(__file__), "golden", "hdf5")
BASE = "tiny_2_cap_per_img_0_min_word_freq"
CONDA_PY = "/opt/conda/bin/python3.9"     # the image's second interpreter (has h5py); optional cross-check


def _expected():
    with open(os.path.join(G, "expected.json")) as fh:
        return json.load(fh)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", sorted(k for k in _expected()["files"] if "gzip" not in k))
def test_h5lite_reads_what_h5py_wrote(name):
    e = _expected()["files"][name]
    with h5lite.File(os.path.join(G, name)) as f:
        assert e["dataset"] in f and e["dataset"] in f.keys()
        d = f[e["dataset"]]
        assert list(d.shape) == e["shape"] and str(d.dtype) == e["dtype"] and len(d) == e["shape"][0]
        for k, v in e["attrs"].items():
            assert int(f.attrs[k]) == v
        if "row_sha256" in e:
            assert [_sha(d[i]) for i in range(d.shape[0])] == e["row_sha256"]
            assert _sha(d[:]) == _sha(np.stack([d[i] for i in range(d.shape[0])]))
        for i, c, y, x, v in e.get("probe", []):
            assert int(d[i][c, y, x]) == v and int(d.array[i, c, y, x]) == v
        if "values" in e:
            assert np.array_equal(d[:], np.asarray(e["values"], dtype=np.float32))
        with pytest.raises(KeyError):
            f["no_such_dataset"]


def test_h5lite_other_datasets_and_indexing():
    with h5lite.File(os.path.join(G, "small_contiguous.hdf5")) as f:
        assert f.keys() == ["aaa_other", "images", "zzz_f32"]
        assert f["aaa_other"].dtype == np.int64 and f["aaa_other"][:].tolist() == list(range(10))
        assert np.array_equal(f["zzz_f32"][:], np.linspace(0, 1, 7).astype("float32"))
        d = f["images"]
        assert d[[3, 1, 1]].shape == (3, 3, 8, 8) and np.array_equal(d[[3, 1, 1]][1], d[1])
        assert d[1:4].shape == (3, 3, 8, 8) and d.layout == "contiguous"
        a = d[0]
        a[...] = 0                       # a copy, like h5py: the file view is untouched
        assert d[0].any()
    with h5lite.File(os.path.join(G, "small_many_links.hdf5")) as f:
        assert len(f.keys()) == 41 and f["d17"][:].tolist() == [17, 17, 17]


def test_h5lite_refuses_what_it_cannot_read(tmp_path):
    with h5lite.File(os.path.join(G, "small_gzip.hdf5")) as f:
        with pytest.raises(h5lite.H5Error, match="compressed"):
            f["images"]
    p = tmp_path / "not_hdf5.hdf5"
    p.write_bytes(b"x" * 4096)
    with pytest.raises(h5lite.H5Error, match="signature"):
        h5lite.File(str(p))
    src = open(os.path.join(G, "VAL_IMAGES_" + BASE + ".hdf5"), "rb").read()
    q = tmp_path / "truncated.hdf5"
    q.write_bytes(src[:len(src) // 2])
    with h5lite.File(str(q)) as f:
        with pytest.raises(h5lite.H5Error, match="truncated"):
            f["images"]
    with pytest.raises(ValueError):
        h5lite.File(os.path.join(G, "small_contiguous.hdf5"), "w")


def test_h5lite_writer_roundtrip_and_real_h5py(tmp_path):
    rng = np.random.RandomState(5)
    imgs = rng.randint(0, 256, size=(6, 3, 16, 16)).astype("uint8")
    tags = rng.rand(6, 9).astype("float32")
    pi, pt = str(tmp_path / "w_images.hdf5"), str(tmp_path / "w_tags.hdf5")
    h5lite.write_arrays(pi, {"images": imgs}, {"captions_per_image": 5})
    h5lite.write_arrays(pt, {"tags": tags, "extra": np.arange(6, dtype="int32").reshape(2, 3)}, {"tag_size": 9})
    with h5lite.File(pi) as f:
        assert int(f.attrs["captions_per_image"]) == 5 and np.array_equal(f["images"][:], imgs)
    with h5lite.File(pt) as f:
        assert int(f.attrs["tag_size"]) == 9 and np.array_equal(f["tags"][:], tags)
        assert f["extra"][:].tolist() == [[0, 1, 2], [3, 4, 5]]
    with pytest.raises(ValueError):
        h5lite.write_arrays(pi, {})
    if not os.path.exists(CONDA_PY):
        return                                        # cross-check with libhdf5 only where the image has it
    code = ("import h5py, hashlib, json, numpy as np\n"
            "h = h5py.File(%r, 'r'); t = h5py.File(%r, 'r')\n"
            "print(json.dumps([int(h.attrs['captions_per_image']), list(h['images'].shape), str(h['images'].dtype),"
            " hashlib.sha256(h['images'][:].tobytes()).hexdigest(), int(t.attrs['tag_size']), sorted(t.keys()),"
            " hashlib.sha256(t['tags'][:].tobytes()).hexdigest(), t['extra'][:].tolist()]))\n" % (pi, pt))
    out = subprocess.run([CONDA_PY, "-c", code], capture_output=True, text=True, timeout=120,
                         env={"PATH": "/usr/bin:/bin"})
    if out.returncode != 0 and "No module named" in out.stderr:
        return
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got == [5, [6, 3, 16, 16], "uint8", _sha(imgs), 9, ["extra", "tags"], _sha(tags), [[0, 1, 2], [3, 4, 5]]]


def _load_json(split, kind):
    with open(os.path.join(G, "%s_%s_%s.json" % (split, kind, BASE))) as fh:
        return json.load(fh)


@pytest.mark.parametrize("split", ["TRAIN", "VAL"])
def test_caption_dataset_items_match_oracle(split):
    from datasets.caption import CaptionDataset, Normalize
    ds = CaptionDataset(G, BASE, split, transform=Normalize(DR.MEAN, DR.STD), cpi=2)
    with h5lite.File(os.path.join(G, split + "_IMAGES_" + BASE + ".hdf5")) as f:
        imgs = f["images"][:]
    caps, lens = _load_json(split, "CAPTIONS"), _load_json(split, "CAPLENS")
    assert len(ds) == len(caps) == 2 * imgs.shape[0]
    for i in range(len(ds)):
        got = ds[i]
        want = DR.caption_item(imgs, caps, lens, i, 2, split)
        assert len(got) == len(want) == (3 if split == "TRAIN" else 4)
        for a, b in zip(got, want):
            assert a.dtype == b.dtype and torch.equal(a, b)
    raw = CaptionDataset(G, BASE, split, cpi=None)            # cpi falsy -> the file attribute (caption.py:32)
    assert raw.cpi == 2 and torch.equal(raw[1][0], torch.FloatTensor(imgs[0] / 255.))
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=2)   # > 1 worker works here
    first = next(iter(loader))
    want = DR.caption_batch(imgs, caps, lens, [0, 1, 2], 2, split)
    for a, b in zip(first, want):
        assert torch.equal(a, b)


def test_tag_dataset_items_match_oracle():
    from datasets.tag import TagDataset
    ds = TagDataset(G, BASE, "TRAIN")
    e = _expected()["files"]["TRAIN_TAGS_" + BASE + ".hdf5"]
    assert len(ds) == 3
    with h5lite.File(os.path.join(G, "TRAIN_IMAGES_" + BASE + ".hdf5")) as f:
        for i in range(3):
            img, tags = ds[i]
            assert torch.equal(img, DR.image_item(f["images"][i], None))
            assert torch.equal(tags, torch.tensor(e["values"][i], dtype=torch.float32))


def test_normalize_lut_is_the_reference_arithmetic():
    lut = SD.normalize_lut()
    assert lut.shape == (3, 256) and lut.dtype == np.float32
    every = np.repeat(np.arange(256, dtype=np.uint8)[None, None, :], 3, axis=0)       # (3, 1, 256): all byte values
    want = DR.image_item(every).numpy()[:, 0, :]
    assert np.array_equal(lut.view(np.uint32), want.view(np.uint32))                    # bit for bit
    ident = SD.identity_lut(3)
    assert np.array_equal(ident[1], (np.arange(256) / 255.).astype(np.float32))
    with pytest.raises(ValueError):
        SD.normalize_lut((0.5, 0.5, 0.5), (0.2, 0.0, 0.2))


def test_epoch_order_and_rank_sharding():
    n = 103
    a = SD.epoch_order(n, 0, 7, True)
    b = SD.epoch_order(n, 1, 7, True)
    assert sorted(a.tolist()) == list(range(n)) and not np.array_equal(a, b)
    assert np.array_equal(a, SD.epoch_order(n, 0, 7, True))
    assert SD.epoch_order(n, 3, 7, False).tolist() == list(range(n))
    parts = [SD.epoch_order(n, 0, 7, True, r, 4) for r in range(4)]
    assert len({len(p) for p in parts}) == 1 and len(parts[0]) == 26
    assert set(np.concatenate(parts).tolist()) == set(range(n))
    inter = np.stack(parts, axis=1).reshape(-1)[:n]
    assert np.array_equal(inter, a)                  # rank r takes positions r, r+world, ... of the same permutation


def test_device_loader_has_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SD.DeviceBatchLoader(G, BASE, "TRAIN", 2, "cpu", cpi=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SD.gather_normalize(torch.zeros(1, 3, 4, 4, dtype=torch.uint8), None, torch.zeros(3, 256))


def test_h5lite_corrupt_files_fail_cleanly(tmp_path):
    """Bounded fuzz: random byte flips in the metadata region and random truncation of valid files must end in
    H5Error / KeyError (or succeed), never in another exception, a hang or a giant allocation."""
    import random
    import signal
    rng = random.Random(4)

    def on_alarm(*_):
        raise AssertionError("h5lite hung on a corrupt file")
    old = signal.signal(signal.SIGALRM, on_alarm)
    try:
        for name in ("small_contiguous.hdf5", "small_latest.hdf5", "small_chunked.hdf5", "small_many_links.hdf5"):
            data = open(os.path.join(G, name), "rb").read()
            for trial in range(120):
                d = bytearray(data)
                for _ in range(rng.randint(1, 4)):
                    d[rng.randrange(min(len(d), 4096))] = rng.randrange(256)
                if rng.random() < 0.2:
                    d = d[:rng.randrange(32, len(d))]
                p = tmp_path / "fuzz.hdf5"
                p.write_bytes(bytes(d))
                signal.alarm(10)
                try:
                    with h5lite.File(str(p)) as f:
                        for k in f.keys()[:50]:
                            try:
                                f[k][:]
                            except (h5lite.H5Error, KeyError):
                                pass
                except (h5lite.H5Error, KeyError):
                    pass
                finally:
                    signal.alarm(0)
    finally:
        signal.signal(signal.SIGALRM, old)
