"""Test infrastructure: genuine HDF5 fixtures in the on-disk format the reference's dataset builder
writes (utils/dataset.py:332-414) and its datasets read (datasets/caption.py:25-41, datasets/tag.py:23-34).
h5py is not importable by the project interpreter; the image's second interpreter has it:

    /opt/conda/bin/python3.9 oracle/gen_hdf5_golden.py        # h5py 3.3.0 / HDF5 1.10.6

Written exactly the way the reference does: `h5py.File(name, 'w')`, `h.attrs['captions_per_image'] = cpi`,
`h.create_dataset('images', (N, 3, 256, 256), dtype='uint8')` then row-by-row assignment; tags file with
`t.attrs['tag_size']` and a float32 `tags` dataset; captions / caplens / word map as JSON.
expected.json holds what h5py reads back (sha256 per image row, attribute values, shapes), so the
project's own reader is checked against h5py's view of the same bytes.  Extra small files cover layouts
the reference never writes (chunked, compressed, libver='latest') for the reader's error paths."""
import hashlib
import json
import os

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "hdf5")
BASE = "tiny_2_cap_per_img_0_min_word_freq"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.RandomState(7)
    expected = {"h5py": h5py.__version__, "hdf5": h5py.version.hdf5_version, "files": {}}
    cpi, tag_size, max_len = 2, 12, 6
    word_map = {"w%d" % i: i + 1 for i in range(20)}
    word_map["<unk>"] = 21
    word_map["<start>"] = 22
    word_map["<end>"] = 23
    word_map["<pad>"] = 0
    with open(os.path.join(OUT, "WORDMAP_" + BASE + ".json"), "w") as j:
        json.dump(word_map, j)
    for split, n in (("TRAIN", 3), ("VAL", 2)):
        ipath = os.path.join(OUT, split + "_IMAGES_" + BASE + ".hdf5")
        tpath = os.path.join(OUT, split + "_TAGS_" + BASE + ".hdf5")
        with h5py.File(ipath, "w") as h:
            with h5py.File(tpath, "w") as t:
                h.attrs["captions_per_image"] = cpi
                t.attrs["tag_size"] = tag_size
                images = h.create_dataset("images", (n, 3, 256, 256), dtype="uint8")
                tags = t.create_dataset("tags", (n, tag_size), dtype="float32")
                caps, lens = [], []
                for i in range(n):
                    img = rng.randint(0, 256, size=(3, 256, 256)).astype("uint8")
                    img[:, 0, :8] = [0, 1, 2, 127, 128, 254, 255, 255]       # every LUT corner appears
                    images[i] = img
                    tags[i] = (rng.rand(tag_size) > 0.7).astype("float32")
                    for _ in range(cpi):
                        k = rng.randint(1, max_len + 1)
                        words = rng.randint(1, 21, size=k).tolist()
                        caps.append([22] + words + [23] + [0] * (max_len - k))
                        lens.append(k + 2)
        with open(os.path.join(OUT, split + "_CAPTIONS_" + BASE + ".json"), "w") as j:
            json.dump(caps, j)
        with open(os.path.join(OUT, split + "_CAPLENS_" + BASE + ".json"), "w") as j:
            json.dump(lens, j)
        with h5py.File(ipath, "r") as h, h5py.File(tpath, "r") as t:
            d = h["images"]
            expected["files"][os.path.basename(ipath)] = {
                "dataset": "images", "shape": list(d.shape), "dtype": str(d.dtype),
                "attrs": {"captions_per_image": int(h.attrs["captions_per_image"])},
                "row_sha256": [sha(d[i]) for i in range(d.shape[0])],
                "probe": [[int(i), int(c), int(y), int(x), int(d[i, c, y, x])] for i, c, y, x in
                          zip(rng.randint(0, n, 16), rng.randint(0, 3, 16), rng.randint(0, 256, 16),
                              rng.randint(0, 256, 16))]}
            g = t["tags"]
            expected["files"][os.path.basename(tpath)] = {
                "dataset": "tags", "shape": list(g.shape), "dtype": str(g.dtype),
                "attrs": {"tag_size": int(t.attrs["tag_size"])}, "values": np.asarray(g).tolist()}

    # ---- small files for reader coverage beyond what the reference writes --------------------------------
    small = rng.randint(0, 256, size=(5, 3, 8, 8)).astype("uint8")

    def record(name, note):
        expected["files"][name] = {"dataset": "images", "shape": list(small.shape), "dtype": "uint8",
                                   "attrs": {"captions_per_image": 5}, "row_sha256": [sha(r) for r in small],
                                   "note": note}

    with h5py.File(os.path.join(OUT, "small_contiguous.hdf5"), "w") as h:
        h.attrs["captions_per_image"] = 5
        h.create_dataset("images", data=small)
        h.create_dataset("aaa_other", data=np.arange(10, dtype="int64"))      # more than one link in the root group
        h.create_dataset("zzz_f32", data=np.linspace(0, 1, 7).astype("float32"))
    record("small_contiguous.hdf5", "contiguous, three datasets in the root group")
    with h5py.File(os.path.join(OUT, "small_latest.hdf5"), "w", libver="latest") as h:
        h.attrs["captions_per_image"] = 5
        h.create_dataset("images", data=small)
    record("small_latest.hdf5", "libver=latest: superblock v3, version-2 object headers, link messages")
    with h5py.File(os.path.join(OUT, "small_chunked.hdf5"), "w") as h:
        h.attrs["captions_per_image"] = 5
        h.create_dataset("images", data=small, chunks=(1, 3, 8, 8))
    record("small_chunked.hdf5", "chunked layout, no filter")
    with h5py.File(os.path.join(OUT, "small_gzip.hdf5"), "w") as h:
        h.attrs["captions_per_image"] = 5
        h.create_dataset("images", data=small, chunks=(2, 3, 8, 8), compression="gzip")
    record("small_gzip.hdf5", "chunked + deflate: the reader must refuse it with a clear message")
    with h5py.File(os.path.join(OUT, "small_many_links.hdf5"), "w") as h:
        h.attrs["captions_per_image"] = 5
        for i in range(40):                                                    # > one symbol-table node (2K = 8 entries/leaf)
            h.create_dataset("d%02d" % i, data=np.full(3, i, dtype="int32"))
        h.create_dataset("images", data=small)
    record("small_many_links.hdf5", "41 links: multi-node group B-tree")

    with open(os.path.join(OUT, "expected.json"), "w") as j:
        json.dump(expected, j)
    for f in sorted(os.listdir(OUT)):
        print("%8d  %s" % (os.path.getsize(os.path.join(OUT, f)), f))


if __name__ == "__main__":
    main()
